#!/usr/bin/env python3
"""Round 6 fixtures: the reference's zero-parameter baseline `DudDecoder` (embedding_decoder.py:454-610) on seeded targets -- `forward` (single target, B x M and M x B
multi-target with and without weights, duplicated prefixes so that the vote has something to decide, zero-weighted trailing targets, only_pred, num_end_loss 2),
`generate` (with and without loss / logits, tau, alpha, label smoothing), `generate_beam`, `generate_all`.
Runs ONLY in the build container (imports /root/reference through make_golden.py's set-up):  python tests/golden/make_golden_r6.py
Inputs and outputs only; the decoder has no weights."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402,F401  (sets up sys.path for the reference)
from make_golden import ref_decoder, FakeEmbedder, make_target_config, make_data_config, t2l  # noqa: E402

V, CMAX, F = 23, 6, 8
DUD = {"unknown": [7, 3, 0], "": [0]}


class DudEmbedder(FakeEmbedder):
	"""+ tokenize_target for the two nouns DudDecoder asks for (reference :466-468); `known` False makes 'unknown' untokenisable (-1 ids) so that the fallback runs."""

	def __init__(self, known=True):
		super().__init__(F, make_target_config(V, CMAX))
		self.known = known

	def tokenize_target(self, text):
		ids = DUD[text] if (self.known or text == "") else [-1, -1, 0]
		t = torch.tensor([ids], dtype=torch.int64)
		return t, torch.zeros_like(t, dtype=torch.bool)


CFG = dict(vocab_quant=False, label_smoothing=0.0, hidden_dim=64, feedfwd_scale="1/4", mlp_hidden_layer="none", mlp_hidden_bias=False, mlp_hidden_norm=False, mlp_hidden_activation="gelu",
           input_dropout=0.0, num_layers=2, num_heads=4, layer_dropout=0.0, layer_activation="gelu", layer_norm_first=True, layer_bias=False, logits_bias=False, init_bias_zero=True,
           init_mlp_mode="balanced", init_mlp_unit_norm=False, init_tfrm_mode="balanced", init_tfrm_unit_norm=False, init_tfrm_unit_postnorm=True, init_tfrm_proj_layers=True,
           init_zero_norm=False, init_rezero_mode="none", mlp_seq_len=4)


def targets(B, M, seed, weights, drop_last):
	"""Targets with many shared prefixes: a small alphabet, and every other target copies a prefix of its sample's first target."""
	g = torch.Generator().manual_seed(seed)
	n = B * (M or 1)
	C = CMAX
	lens = torch.randint(1, C, (n,), generator=g)
	tgt = torch.zeros(n, C, dtype=torch.int64)
	pad = torch.zeros(n, C, dtype=torch.bool)
	for i, ln in enumerate(lens.tolist()):
		tgt[i, :ln] = torch.randint(1, 5, (ln,), generator=g)
		pad[i, ln + 1:] = True
	w = None
	if M:
		tgt, pad = tgt.view(B, M, C), pad.view(B, M, C)
		for b in range(B):
			for m in range(1, M):
				if torch.rand((), generator=g) < 0.6:
					k = int(torch.randint(1, C - 1, (), generator=g))
					tgt[b, m, :k] = tgt[b, 0, :k]
					pad[b, m, :k + 1] = False  # (a token behind a copied prefix exists or is END: not padding)
		if weights:
			w = torch.rand(B, M, generator=g).sort(dim=1, descending=True)[0]
			w[:, 1] = w[:, 0] * (torch.rand(B, generator=g) < 0.5)  + w[:, 1] * 0  # ties and zeros between the first two targets
			if drop_last:
				dead = torch.rand(B, generator=g) < 0.5
				w[dead, -1] = 0
				pad[dead, -1, :] = True
				tgt[dead, -1, :] = 0
	elif weights:
		w = (torch.rand(n, generator=g) > 0.2).float() * torch.rand(n, generator=g)
	return tgt, pad, w


def main():
	out = dict(V=V, CMAX=CMAX, F=F, dud=DUD, cfg=dict(CFG), forward=[], generate=[], beam=[], all=[])
	cases = [("single", None, False, False, False, 1, False), ("single_w", None, True, False, False, 1, False), ("single_only_pred", None, False, False, False, 1, True),
	         ("multi", 4, False, False, False, 1, False), ("multi_w", 4, True, False, False, 1, False), ("multi_w_drop", 3, True, True, False, 1, False),
	         ("multi_first", 4, False, False, True, 1, False), ("multi_first_w", 3, True, True, True, 1, False), ("multi_end2", 4, True, False, False, 2, False),
	         ("multi_only_pred", 4, True, False, False, 1, True), ("single_end3", None, False, False, False, 3, False)]
	for idx, (name, M, weights, drop, mf, nel, only_pred) in enumerate(cases):
		dc = make_data_config(multi_target=bool(M), multi_first=mf, use_weights=weights and bool(M), multi_length=M or 1)
		model = ref_decoder.DudDecoder(embedder=DudEmbedder(), data_config=dc, num_end_loss=nel, **CFG)
		B = 7
		tgt, pad, w = targets(B, M, 100 + idx, weights, drop)
		if mf:
			tgt, pad = tgt.transpose(0, 1).contiguous(), pad.transpose(0, 1).contiguous()
			w = None if w is None else w.transpose(0, 1).contiguous()
		embed = torch.zeros(B, F)
		for use_pad in (True, False):
			res = model(embed, tgt, pad if use_pad else None, w, True, True, only_pred, None)
			out["forward"].append(dict(name=f"{name}{'' if use_pad else '_nopad'}", M=M, multi_first=mf, weights=weights, num_end_loss=nel, only_pred=only_pred, target=tgt,
			                           padding=pad if use_pad else None, weight=w, logits_argmax=res[0].argmax(dim=-1), logits_sum=res[0].sum(dim=-1), out_padding=t2l(res[1]),
			                           loss_sum=t2l(res[2]), loss_basis=t2l(res[3]), correct=t2l(res[4])))
	for known in (True, False):
		for ls, tau, alpha in ((0.0, 1.0, 0.0), (0.1, 2.0, 0.5)):
			model = ref_decoder.DudDecoder(embedder=DudEmbedder(known), data_config=make_data_config(), num_end_loss=1, **dict(CFG, label_smoothing=ls))
			embed = torch.zeros(5, F)
			for collect, loss in ((False, True), (True, False), (False, False)):
				res = model.generate(embed, collect, loss, tau, alpha, None, None, False)
				out["generate"].append(dict(known=known, label_smoothing=ls, tau=tau, alpha=alpha, collect=collect, loss=loss, out=tuple(None if t is None else t2l(t) for t in res)))
			out["beam"].append(dict(known=known, out=tuple(t2l(t) for t in model.generate_beam(embed, 3, tau, alpha, None, False, 0.0, None, False))))
			guide = torch.zeros(9, 5, dtype=torch.int64)
			assert model.precompute_generate_all(alpha, None, False, 0.0, guide, False) is None
			out["all"].append(dict(known=known, guide_shape=tuple(guide.shape), out=tuple(t2l(t) for t in model.generate_all(embed, 4, tau, alpha, None, False, 0.0, guide, False))))
		total, parts = model.get_num_params()
		out["num_params"] = (total.total, list(parts))
	torch.save(out, os.path.join(HERE, "dud_decoder.pt"))
	n_changed = sum(int((c["logits_argmax"] != c["target"][..., -c["logits_argmax"].shape[-1]:]).sum()) for c in out["forward"])
	print(f"wrote dud_decoder.pt: {len(out['forward'])} forward cases ({n_changed} predictions differ from their own target: the vote decided), "
	      f"{len(out['generate'])} generate, {len(out['beam'])} beam, {len(out['all'])} generate_all")


if __name__ == "__main__":
	main()
