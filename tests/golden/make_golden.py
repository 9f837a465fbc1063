#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the reference's own modules.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

It imports the reference's ``embedding_decoder`` / ``embedding_noise`` / ``embedders`` /
``embedding_dataset`` / ``infer`` modules as they are (with an empty stand-in for the missing,
unused ``unidecode`` package), drives them on seeded synthetic inputs and stores *inputs and
outputs only* (data, no source) as ``*.pt`` fixtures.  Model weights are not stored: they are
re-created from ``oracle.decoder_oracle.init_state_dict(spec, seed)`` and loaded into the
reference module with ``load_state_dict(strict=True)``, which also pins the state-dict key names.

While generating, every fixture is cross-checked against the repo's CPU oracle so a drift between
oracle and reference fails here, loudly, before anything is written.
"""
from __future__ import annotations

import dataclasses
import math
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.modules.setdefault("unidecode", types.ModuleType("unidecode"))  # only used by utils.get_canon (off the hot path)
sys.modules["unidecode"].unidecode = lambda s: s

import embedders as ref_embedders  # noqa: E402
import embedding_dataset as ref_dataset  # noqa: E402
import embedding_decoder as ref_decoder  # noqa: E402
import embedding_noise as ref_noise  # noqa: E402
import infer as ref_infer  # noqa: E402

from oracle import decoder_oracle as O  # noqa: E402
from oracle import noise_oracle as NO  # noqa: E402

torch.set_num_threads(8)
torch.use_deterministic_algorithms(False)


# ---------------------------------------------------------------------------------------------
# reference model construction
# ---------------------------------------------------------------------------------------------

class FakeEmbedder:
	"""Duck-typed stand-in: PrefixedIterDecoder reads only these fields (embedding_decoder.py:77-86)."""

	def __init__(self, embed_dim, target_config):
		self.embed_dtype = torch.float32
		self.embed_dim = embed_dim
		self.target_config = target_config
		self.target_vocab = ()


def make_target_config(V, Cmax, token_dtype=torch.int64):
	return ref_embedders.TargetConfig(
		vocab_size=V, token_dtype=token_dtype, mask_dtype=torch.bool, start_token_id=None, end_token_id=0, pad_token_id=0, compact_ids=True,
		compact_map=None, compact_unmap=None, fixed_token_length=False, token_length=Cmax, use_masks=True,
	)


def make_data_config(multi_target=False, multi_first=False, use_weights=False, multi_length=1):
	return ref_dataset.DataConfig.create(dict(use_weights=use_weights, unit_weights=True, multi_target=multi_target, multi_first=multi_first, full_targets=True, fixed_multi_length=True, multi_length=multi_length))


def ref_model(spec: O.DecoderSpec, seed: int, data_config=None, dropout=0.0, token_dtype=torch.int64):
	cfg = dict(
		vocab_quant=False, num_end_loss=spec.num_end_loss, label_smoothing=spec.label_smoothing, hidden_dim=spec.hidden_dim,
		feedfwd_scale=f"{spec.feedfwd_dim}/{spec.hidden_dim}", mlp_hidden_layer="none", mlp_hidden_bias=False, mlp_hidden_norm=False,
		mlp_hidden_activation="gelu", input_dropout=dropout, num_layers=spec.num_layers, num_heads=spec.num_heads, layer_dropout=dropout,
		layer_activation="gelu", layer_norm_first=True, layer_bias=False, logits_bias=False, init_bias_zero=True, init_mlp_mode="balanced",
		init_mlp_unit_norm=False, init_tfrm_mode="balanced", init_tfrm_unit_norm=False, init_tfrm_unit_postnorm=True, init_tfrm_proj_layers=True,
		init_zero_norm=False, init_rezero_mode="none", mlp_seq_len=spec.mlp_seq_len, weight_tying=True, strictly_causal=spec.strictly_causal,
		enable_nested=False,
	)
	embedder = FakeEmbedder(spec.embed_dim, make_target_config(spec.vocab_size, spec.token_length, token_dtype))
	data_config = data_config or make_data_config()
	torch.manual_seed(seed + 1000)
	model = ref_decoder.PrefixedIterDecoder(embedder=embedder, data_config=data_config, **cfg)
	init_stats = {k: (float(v.float().mean()), float(v.float().std()) if v.numel() > 1 else 0.0) for k, v in model.state_dict().items() if k != "causality_mask"}
	sd = O.init_state_dict(spec, seed=seed)
	model.load_state_dict(sd, strict=True)  # pins key names/shapes incl. the causality_mask buffer
	assert torch.equal(model.causality_mask, sd["causality_mask"])
	model.eval()
	return model, sd, init_stats


def synth_batch(spec, B, seed, M=None, max_len=None, weights=False, token_dtype=torch.int64, full_targets=True):
	"""Unit-norm Gaussian embeddings + targets of random content length followed by END(0) and padding."""
	g = torch.Generator().manual_seed(seed)
	embed = torch.nn.functional.normalize(torch.randn(B, spec.embed_dim, generator=g), dim=-1)
	max_len = max_len or (spec.token_length - 1)
	n = B * (M or 1)
	lens = torch.randint(1, max_len + 1, (n,), generator=g)
	C = int(lens.max()) + 1
	target = torch.zeros(n, C, dtype=token_dtype)
	pad = torch.zeros(n, C, dtype=torch.bool)
	for i, ln in enumerate(lens.tolist()):
		target[i, :ln] = torch.randint(1, spec.vocab_size, (ln,), generator=g).to(token_dtype)
		pad[i, ln + 1:] = True
	weight = None
	if M is not None:
		target, pad = target.view(B, M, C), pad.view(B, M, C)
		if weights:
			w = torch.rand(B, M, generator=g).sort(dim=1, descending=True)[0]
			if not full_targets:  # zero-weighted, fully padded trailing targets for some samples
				drop = torch.rand(B, generator=g) < 0.4
				w[drop, -1] = 0
				pad[drop, -1, :] = True
				target[drop, -1, :] = 0
			weight = w / w.sum(dim=1, keepdim=True)
	elif weights:
		weight = torch.rand(B, generator=g) + 0.1
	return embed, target, pad, weight


def t2l(x):
	return None if x is None else x.detach().clone()


def check(name, a, b, atol=2e-5, rtol=1e-5, exact=False):
	if a is None or b is None:
		assert a is None and b is None, name
		return
	a, b = torch.as_tensor(a), torch.as_tensor(b)
	assert a.shape == b.shape, (name, a.shape, b.shape)
	if exact or a.dtype in (torch.bool, torch.int32, torch.int64):
		assert torch.equal(a, b.to(a.dtype)), name
	else:
		finite = torch.isfinite(a)
		assert torch.equal(finite, torch.isfinite(b)), name
		torch.testing.assert_close(a[finite].float(), b[finite].float(), atol=atol, rtol=rtol, msg=lambda m: f"{name}: {m}")


# ---------------------------------------------------------------------------------------------
# fixtures
# ---------------------------------------------------------------------------------------------

SMALL = O.DecoderSpec(embed_dim=32, vocab_size=53, token_length=6, hidden_dim=64, feedfwd_dim=16, num_layers=2, num_heads=4, mlp_seq_len=4)
DEFAULT = O.DecoderSpec(embed_dim=512, vocab_size=307, token_length=8)  # config/train.yaml defaults, small vocab to keep fixtures small


def forward_cases():
	cases = []
	variants = [
		("small_pad", SMALL, dict(B=7), dict()),
		("small_nopad", SMALL, dict(B=5), dict(no_pad=True)),
		("small_onlypred", SMALL, dict(B=5), dict(only_pred=True)),
		("small_weighted", SMALL, dict(B=6, weights=True), dict()),
		("small_weighted_nopad", SMALL, dict(B=6, weights=True), dict(no_pad=True)),
		("small_multi", SMALL, dict(B=4, M=3, weights=True), dict(data=dict(multi_target=True, use_weights=True, multi_length=3))),
		("small_multi_ragged", SMALL, dict(B=6, M=3, weights=True, full_targets=False), dict(data=dict(multi_target=True, use_weights=True, multi_length=3))),
		("small_multi_first", dataclasses.replace(SMALL, multi_first=True), dict(B=4, M=2, weights=True), dict(data=dict(multi_target=True, multi_first=True, use_weights=True, multi_length=2))),
		("small_smooth", dataclasses.replace(SMALL, label_smoothing=0.1), dict(B=5), dict()),
		("small_endloss2", dataclasses.replace(SMALL, num_end_loss=2), dict(B=5), dict()),
		("small_strict", dataclasses.replace(SMALL, strictly_causal=True), dict(B=5), dict()),
		("small_p1", dataclasses.replace(SMALL, mlp_seq_len=1), dict(B=5), dict()),
		("small_int32", SMALL, dict(B=5, token_dtype=torch.int32), dict(token_dtype=torch.int32, no_loss=True)),  # F.cross_entropy rejects int32 targets
		("default_pad", DEFAULT, dict(B=6), dict()),
		("default_multi", DEFAULT, dict(B=3, M=3, weights=True), dict(data=dict(multi_target=True, use_weights=True, multi_length=3))),
	]
	for idx, (name, spec, bk, opt) in enumerate(variants):
		seed = 100 + idx
		dc = make_data_config(**opt.get("data", {}))
		model, sd, init_stats = ref_model(spec, seed, data_config=dc, token_dtype=opt.get("token_dtype", torch.int64))
		embed, target, pad, weight = synth_batch(spec, seed=seed, **bk)
		if spec.multi_first and target.ndim == 3:
			target, pad, weight = target.transpose(0, 1).contiguous(), pad.transpose(0, 1).contiguous(), weight.transpose(0, 1).contiguous()
		if opt.get("no_pad"):
			pad = None
		only_pred = bool(opt.get("only_pred"))
		calc_loss = not opt.get("no_loss")
		with torch.no_grad():
			out = model(embed=embed, target=target, target_padding=pad, target_weight=weight, calc_loss=calc_loss, calc_correct=True, only_pred=only_pred, guide_targets=None)
		mine = O.forward(sd, spec, embed, target, pad, weight, calc_loss, True, only_pred)
		for nm, a, b in zip(("logits", "padding", "loss_sum", "loss_basis", "correct"), out, mine):
			check(f"{name}.{nm}", a, b)
		case = dict(name=name, spec=dataclasses.asdict(spec), seed=seed, embed=embed, target=target, padding=pad, weight=weight, only_pred=only_pred,
		            calc_loss=calc_loss,
		            logits=t2l(out[0]), out_padding=t2l(out[1]), loss_sum=t2l(out[2]), loss_basis=(None if out[3] is None else t2l(torch.as_tensor(out[3]))), correct=t2l(out[4]))
		if name in ("small_pad", "default_pad", "small_multi"):
			# reference under CPU bf16 autocast: pins the oracle's bf16 emulation loosely
			with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
				ob = model(embed=embed, target=target, target_padding=pad, target_weight=weight, calc_loss=True, calc_correct=False, only_pred=only_pred, guide_targets=None)
			mb = O.forward(sd, spec, embed, target, pad, weight, True, False, only_pred, bf16=True)
			err = (ob[0].float() - mb[0]).abs().max().item()
			scale = ob[0].float().abs().max().item()
			assert err <= 0.04 * max(scale, 1.0), (name, err, scale)
			case.update(bf16_logits=ob[0].float().clone(), bf16_loss_sum=ob[2].float().clone())
		if name == "default_pad":
			case.update(init_stats=init_stats)
			# gradients of mean loss wrt a few tensors (pins the backward targets)
			model.zero_grad()
			o2 = model(embed=embed, target=target, target_padding=pad, target_weight=weight, calc_loss=True, calc_correct=False, only_pred=False, guide_targets=None)
			(o2[2] / o2[3]).backward()
			grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
			sdg = {k: (v.clone().requires_grad_(True) if k != "causality_mask" else v) for k, v in sd.items()}
			m2 = O.forward(sdg, spec, embed, target, pad, weight, True, False, False)
			(m2[2] / m2[3]).backward()
			for k, gv in grads.items():
				check(f"{name}.grad.{k}", gv, sdg[k].grad, atol=1e-5, rtol=1e-4)
			case.update(grad_norms={k: float(gv.norm()) for k, gv in grads.items()},
			            grad_samples={k: gv.flatten()[:: max(1, gv.numel() // 64)][:64].clone() for k, gv in grads.items()})
		cases.append(case)
	return cases


def generate_cases():
	cases = []
	spec_small = dataclasses.replace(SMALL, vocab_size=61, token_length=7)
	for idx, (name, spec, B, zero_end, tau, alpha, dtype) in enumerate([
		("greedy_small", spec_small, 9, False, 1.0, 0.0, torch.int64),
		("greedy_small_full", spec_small, 9, True, 2.0, 0.5, torch.int64),
		("greedy_small_int32", spec_small, 5, True, 1.0, 1.0, torch.int32),
		("greedy_default", DEFAULT, 6, False, 1.0, 0.0, torch.int64),
		("greedy_default_full", DEFAULT, 6, True, 1.0, 0.5, torch.int64),
	]):
		seed = 200 + idx
		model, sd, _ = ref_model(spec, seed, token_dtype=dtype)
		if zero_end:  # END logit exactly 0 => generation runs the full G steps on a random-init model
			sd["logits_linear.weight"][0].zero_()
			model.load_state_dict(sd)
		embed = synth_batch(spec, B, seed)[0]
		with torch.no_grad():
			calc_loss = dtype == torch.int64  # F.cross_entropy rejects int32 targets, so the int32 case is ids/padding only
			out = model.generate(embed=embed, collect_logits=True, calc_loss=calc_loss, temperature=tau, length_alpha=alpha, sample_weight=None, guide_targets=None, guide_renorm=False)
		mine = O.generate(sd, spec, embed, True, calc_loss, tau, alpha, None, token_dtype=dtype)
		for nm, a, b in zip(("ids", "padding", "logits", "loss_sum", "loss_basis", "score"), out, mine):
			if nm == "logits":  # unspecified at padded positions (embedding_decoder.py:798)
				keep = ~out[1]
				check(f"{name}.{nm}", a[keep], b[keep], atol=5e-5)
			else:
				check(f"{name}.{nm}", a, b, atol=5e-5, rtol=1e-5)
		cases.append(dict(name=name, kind="greedy", spec=dataclasses.asdict(spec), seed=seed, zero_end=zero_end, temperature=tau, length_alpha=alpha, token_dtype=dtype,
		                  calc_loss=calc_loss,
		                  embed=embed, ids=t2l(out[0]), padding=t2l(out[1]), logits=t2l(out[2]), loss_sum=t2l(out[3]), loss_basis=(None if out[4] is None else t2l(torch.as_tensor(out[4]))), score=t2l(out[5])))
	for idx, (name, spec, B, H, zero_end, tau, alpha, dtype) in enumerate([
		("beam4_small", spec_small, 7, 4, False, 1.0, 0.0, torch.int64),
		("beam4_small_full", spec_small, 7, 4, True, 1.0, 0.0, torch.int64),
		("beam10_small_full_alpha", spec_small, 5, 10, True, 2.0, 0.5, torch.int64),
		("beam3_small", spec_small, 5, 3, True, 1.0, 1.0, torch.int64),  # (int32 ids: the reference's topk(out=) rejects them, so beam is int64-only)
		("beam4_default", DEFAULT, 5, 4, False, 1.0, 0.0, torch.int64),
		("beam4_default_full", DEFAULT, 5, 4, True, 1.0, 0.0, torch.int64),
		("beam10_default_full_alpha", DEFAULT, 4, 10, True, 1.0, 0.5, torch.int64),
	]):
		seed = 300 + idx
		model, sd, _ = ref_model(spec, seed, token_dtype=dtype)
		if zero_end:
			sd["logits_linear.weight"][0].zero_()
			model.load_state_dict(sd)
		embed = synth_batch(spec, B, seed)[0]
		with torch.no_grad():
			out = model.generate_beam(embed=embed, topk=H, temperature=tau, length_alpha=alpha, vocab_targets=None, vocab_per_token=False, vocab_scaler=0.0, guide_targets=None, guide_renorm=False)
		mine = O.generate_beam(sd, spec, embed, H, tau, alpha, token_dtype=dtype)
		for nm, a, b in zip(("ids", "padding", "score"), out, mine):
			check(f"{name}.{nm}", a, b, atol=5e-5, rtol=1e-5)
		# top-2 margin of the final ranking: tells consumers how tie-free this vector is
		sc = out[2]
		margin = float((sc[:, :-1] - sc[:, 1:]).abs().min()) if H > 1 else math.inf
		cases.append(dict(name=name, kind="beam", spec=dataclasses.asdict(spec), seed=seed, zero_end=zero_end, topk=H, temperature=tau, length_alpha=alpha, token_dtype=dtype,
		                  embed=embed, ids=t2l(out[0]), padding=t2l(out[1]), score=t2l(out[2]), min_rank_margin=margin))
	return cases


def random_guide_targets(spec, W, seed, max_len=None):
	"""W distinct noun tokenisations [W][Cmax]: 1..max_len content tokens (shared prefixes on purpose), END (0), zero padding."""
	g = torch.Generator().manual_seed(seed)
	max_len = max_len or (spec.token_length - 1)
	first = torch.randint(1, spec.vocab_size, (max(3, W // 3),), generator=g)  # few distinct first tokens => branching tries
	rows = set()
	while len(rows) < W:
		ln = int(torch.randint(1, max_len + 1, (1,), generator=g))
		toks = [int(first[int(torch.randint(0, len(first), (1,), generator=g))])] + [int(t) for t in torch.randint(1, min(spec.vocab_size, 12), (ln - 1,), generator=g)]
		rows.add(tuple(toks))
	out = torch.zeros(W, spec.token_length, dtype=torch.int64)
	for i, r in enumerate(sorted(rows)):
		out[i, :len(r)] = torch.tensor(r)
	return out


def guided_cases():
	cases = []
	spec_small = dataclasses.replace(SMALL, vocab_size=61, token_length=7)
	for idx, (name, spec, B, W, kind, kw) in enumerate([
		("greedy_gp_small", spec_small, 8, 25, "greedy", dict(guide_renorm=False, temperature=1.0, length_alpha=0.0)),
		("greedy_gr_small", spec_small, 8, 25, "greedy", dict(guide_renorm=True, temperature=2.0, length_alpha=0.5)),
		("greedy_gp_default", DEFAULT, 5, 40, "greedy", dict(guide_renorm=False, temperature=1.0, length_alpha=0.0)),
		("beam4_gp_small", spec_small, 6, 25, "beam", dict(topk=4, guide_renorm=False, temperature=1.0, length_alpha=0.0, prior=None)),
		("beam4_gr_small", spec_small, 6, 25, "beam", dict(topk=4, guide_renorm=True, temperature=1.0, length_alpha=0.5, prior=None)),
		("beam10_gp_default", DEFAULT, 4, 60, "beam", dict(topk=10, guide_renorm=False, temperature=1.0, length_alpha=0.0, prior=None)),
		("beam5_gp_few_targets", spec_small, 5, 3, "beam", dict(topk=5, guide_renorm=False, temperature=1.0, length_alpha=0.0, prior=None)),  # fewer candidates than beams
		("beam4_gp_prior_tgt_small", spec_small, 6, 25, "beam", dict(topk=4, guide_renorm=False, temperature=1.0, length_alpha=0.0, prior=("guide", False, 1.0))),
		("beam4_gr_prior_tok_small", spec_small, 6, 25, "beam", dict(topk=4, guide_renorm=True, temperature=1.0, length_alpha=0.0, prior=("guide", True, 0.5))),
		("beam4_gn_prior_tgt_small", spec_small, 6, 25, "beam", dict(topk=4, guide_renorm=False, temperature=1.0, length_alpha=0.0, prior=("only", False, 1.0))),
		# the vocabulary nouns of the prior are NOT the guide nouns (a superset minus a few guide nouns: those become unreachable)
		("beam4_gp_prior_tgt_diffvocab_small", spec_small, 6, 25, "beam", dict(topk=4, guide_renorm=False, temperature=1.0, length_alpha=0.0, prior=("diff", False, 1.0))),
		("beam5_gr_prior_tok_diffvocab_small", spec_small, 5, 25, "beam", dict(topk=5, guide_renorm=True, temperature=1.0, length_alpha=0.3, prior=("diff", True, 0.5))),
	]):
		seed = 600 + idx
		model, sd, _ = ref_model(spec, seed)
		embed = synth_batch(spec, B, seed)[0]
		guide = random_guide_targets(spec, W, seed, max_len=4)
		case = dict(name=name, kind=kind, spec=dataclasses.asdict(spec), seed=seed, embed=embed, guide_targets=guide, **{k: v for k, v in kw.items() if k != "prior"})
		if kind == "greedy":
			with torch.no_grad():
				out = model.generate(embed=embed, collect_logits=True, calc_loss=True, temperature=kw["temperature"], length_alpha=kw["length_alpha"], sample_weight=None,
				                     guide_targets=guide, guide_renorm=kw["guide_renorm"])
			mine = O.generate(sd, spec, embed, True, True, kw["temperature"], kw["length_alpha"], None, guide_targets=guide, guide_renorm=kw["guide_renorm"])
			for nm, a, b in zip(("ids", "padding", "logits", "loss_sum", "loss_basis", "score"), out, mine):
				if nm == "logits":
					keep = ~out[1]
					check(f"{name}.{nm}", a[keep], b[keep], atol=5e-5)
				else:
					check(f"{name}.{nm}", a, b, atol=5e-5, rtol=1e-5)
			case.update(ids=t2l(out[0]), padding=t2l(out[1]), logits=t2l(out[2]), loss_sum=t2l(out[3]), loss_basis=t2l(torch.as_tensor(out[4])), score=t2l(out[5]))
		else:
			prior = kw["prior"]
			g_arg = None if (prior and prior[0] == "only") else guide
			v_arg, per_tok, scaler = (guide, prior[1], prior[2]) if prior else (None, False, 0.0)
			if prior and prior[0] == "diff":
				extra = random_guide_targets(spec, 30, seed + 1000, max_len=4)
				v_arg = torch.unique(torch.cat((guide[3:], extra), dim=0), dim=0)   # drops three guide nouns, adds others
				case["vocab_targets"] = v_arg
			with torch.no_grad():
				out = model.generate_beam(embed=embed, topk=kw["topk"], temperature=kw["temperature"], length_alpha=kw["length_alpha"], vocab_targets=v_arg, vocab_per_token=per_tok,
				                          vocab_scaler=scaler, guide_targets=g_arg, guide_renorm=kw["guide_renorm"])
			mine = O.generate_beam(sd, spec, embed, kw["topk"], kw["temperature"], kw["length_alpha"], guide_targets=g_arg, guide_renorm=kw["guide_renorm"], vocab_targets=v_arg,
			                       vocab_per_token=per_tok, vocab_scaler=scaler)
			fin = torch.isfinite(out[2])
			assert torch.equal(fin, torch.isfinite(mine[2])), name
			check(f"{name}.score", out[2][fin], mine[2][fin], atol=5e-5, rtol=1e-5)
			assert torch.equal(out[0][fin], mine[0][fin]) and torch.equal(out[1][fin], mine[1][fin]), name   # -inf beams carry unspecified tokens
			case.update(ids=t2l(out[0]), padding=t2l(out[1]), score=t2l(out[2]), guided=g_arg is not None, vocab_prior=prior is not None, vocab_per_token=per_tok, vocab_scaler=scaler)
		cases.append(case)
	# teacher-forced guided correctness of forward() (embedding_decoder.py:756-763): targets drawn from the guide set, some corrupted so that they leave it
	for idx, (name, spec, B, W) in enumerate([("forward_guided_small", spec_small, 12, 25), ("forward_guided_default", DEFAULT, 6, 40)]):
		seed = 650 + idx
		model, sd, _ = ref_model(spec, seed)
		embed = synth_batch(spec, B, seed)[0]
		guide = random_guide_targets(spec, W, seed, max_len=4)
		g = torch.Generator().manual_seed(seed)
		tgt = guide[torch.randint(0, W, (B,), generator=g)].clone()
		C = int((tgt != 0).sum(dim=1).max()) + 1
		tgt = tgt[:, :C]
		pad = torch.zeros_like(tgt, dtype=torch.bool)
		pad[:, 1:] = (tgt[:, :-1] == 0).cummax(dim=1).values
		tgt[1, 0] = 1 + (int(tgt[1, 0]) % (spec.vocab_size - 1))          # a target that leaves the guide set at its first token
		with torch.no_grad():
			out = model(embed=embed, target=tgt, target_padding=pad, target_weight=None, calc_loss=True, calc_correct=True, only_pred=False, guide_targets=guide)
		mine = O.forward(sd, spec, embed, tgt, pad, None, True, True, False, guide_targets=guide)
		assert torch.equal(out[4], mine[4]), name
		check(f"{name}.loss", out[2], mine[2], atol=1e-4, rtol=1e-5)
		cases.append(dict(name=name, kind="forward", spec=dataclasses.asdict(spec), seed=seed, embed=embed, guide_targets=guide, target=tgt, padding=pad, correct=t2l(out[4]),
		                  loss_sum=t2l(out[2]), logits=t2l(out[0])))
	return cases


def generate_all_cases():
	"""reference generate_all (embedding_decoder.py:986-1079): every guide target scored by teacher forcing, top-k per sample."""
	cases = []
	spec_small = dataclasses.replace(SMALL, vocab_size=61, token_length=7)
	for idx, (name, spec, B, W, kw) in enumerate([
		("all_k5_gp_small", spec_small, 4, 25, dict(topk=5, guide_renorm=False, temperature=1.0, length_alpha=0.0, prior=None)),
		("all_k5_gr_a05_small", spec_small, 4, 25, dict(topk=5, guide_renorm=True, temperature=2.0, length_alpha=0.5, prior=None)),
		("all_k3_gp_prior_tgt_small", spec_small, 3, 25, dict(topk=3, guide_renorm=False, temperature=1.0, length_alpha=0.0, prior=(False, 1.0))),
		("all_k3_gr_prior_tok_small", spec_small, 3, 25, dict(topk=3, guide_renorm=True, temperature=1.0, length_alpha=0.3, prior=(True, 0.5))),
		("all_k10_gp_default", DEFAULT, 2, 30, dict(topk=10, guide_renorm=False, temperature=1.0, length_alpha=0.0, prior=None)),
		("all_k5_gp_prior_tgt_diffvocab_small", spec_small, 3, 25, dict(topk=5, guide_renorm=False, temperature=1.0, length_alpha=0.0, prior=(False, 1.0, "diff"))),
	]):
		seed = 700 + idx
		model, sd, _ = ref_model(spec, seed)
		embed = synth_batch(spec, B, seed)[0]
		guide = random_guide_targets(spec, W, seed, max_len=4)
		prior = kw["prior"]
		v_arg, per_tok, scaler = (guide, prior[0], prior[1]) if prior else (None, False, 0.0)
		diff = bool(prior) and len(prior) > 2
		if diff:
			v_arg = torch.unique(torch.cat((guide[3:], random_guide_targets(spec, 30, seed + 1000, max_len=4)), dim=0), dim=0)
		with torch.no_grad():
			out = model.generate_all(embed=embed, topk=kw["topk"], temperature=kw["temperature"], length_alpha=kw["length_alpha"], vocab_targets=v_arg, vocab_per_token=per_tok,
			                         vocab_scaler=scaler, guide_targets=guide, guide_renorm=kw["guide_renorm"], precompute=None)
		mine = O.generate_all(sd, spec, embed, kw["topk"], kw["temperature"], kw["length_alpha"], guide, kw["guide_renorm"], v_arg, per_tok, scaler)
		fin = torch.isfinite(out[2])
		assert torch.equal(fin, torch.isfinite(mine[2])), name
		check(f"{name}.score", out[2][fin], mine[2][fin], atol=5e-5, rtol=1e-5)
		assert torch.equal(out[0][fin], mine[0][fin]) and torch.equal(out[1][fin], mine[1][fin]), name
		cases.append(dict(name=name, spec=dataclasses.asdict(spec), seed=seed, embed=embed, guide_targets=guide, vocab_targets=v_arg if diff else None, topk=kw["topk"], temperature=kw["temperature"],
		                  length_alpha=kw["length_alpha"], guide_renorm=kw["guide_renorm"], vocab_prior=prior is not None, vocab_per_token=per_tok, vocab_scaler=scaler,
		                  ids=t2l(out[0]), padding=t2l(out[1]), score=t2l(out[2])))
	return cases


def noise_cases():
	cases = []
	B, F = 33, 48
	g = torch.Generator().manual_seed(7)
	base = torch.nn.functional.normalize(torch.randn(B, F, generator=g), dim=-1)
	shift = torch.randn(1, F, generator=g) * 0.05

	# GaussElem
	mod = ref_noise.EmbeddingNoise.create("GaussElem", F, 3.25, 0, 0, 0, 0)
	torch.manual_seed(11)
	ref_out = mod(base.clone())
	torch.manual_seed(11)
	z = torch.randn_like(base)
	check("noise.gauss_elem", ref_out, NO.gauss_elem(base, z, 3.25), atol=1e-6)
	cases.append(dict(name="gauss_elem", embed=base, z=z, vec_norm=3.25, out=ref_out.clone()))

	# GaussVec
	mod = ref_noise.EmbeddingNoise.create("GaussVec", F, 0.7, 0, 0, 0, 0)
	torch.manual_seed(12)
	ref_out = mod(base.clone())
	torch.manual_seed(12)
	z = torch.randn_like(base)
	r = torch.randn(B, 1)
	check("noise.gauss_vec", ref_out, NO.gauss_vec(base, z, r, 0.7), atol=1e-6)
	cases.append(dict(name="gauss_vec", embed=base, z=z, r=r, vec_norm=0.7, out=ref_out.clone()))

	# UniformAngle
	mod = ref_noise.EmbeddingNoise.create("UniformAngle", F, 0, 45.0, 75.0, 0, 0)
	torch.manual_seed(13)
	ref_out = mod(base.clone())
	torch.manual_seed(13)
	z = torch.randn_like(base)
	angle = torch.empty(B, 1).uniform_(math.radians(45.0), math.radians(75.0))
	check("noise.uniform_angle", ref_out, NO.rotate(base, z, angle), atol=1e-6)
	cosang = (ref_out * base).sum(dim=1)
	assert torch.all(cosang <= math.cos(math.radians(45.0)) + 1e-5) and torch.all(cosang >= math.cos(math.radians(75.0)) - 1e-5)
	cases.append(dict(name="uniform_angle", embed=base, z=z, angle=angle, angle_min=45.0, angle_max=75.0, out=ref_out.clone()))

	# GaussAngle
	mod = ref_noise.EmbeddingNoise.create("GaussAngle", F, 0, 0, 40.0, 25.0, 0)
	torch.manual_seed(14)
	ref_out = mod(base.clone())
	torch.manual_seed(14)
	z = torch.randn_like(base)
	r = torch.randn(B, 1)
	check("noise.gauss_angle", ref_out, NO.rotate(base, z, NO.gauss_angle_draw(r, 25.0, 40.0)), atol=1e-6)
	cases.append(dict(name="gauss_angle", embed=base, z=z, r=r, angle_std=25.0, angle_max=40.0, out=ref_out.clone()))

	# GaussElemUniformAngle (released-model recipe: 3.25 / 45-75 deg / 0.15, README.md:322)
	mod = ref_noise.EmbeddingNoise.create("GaussElemUniformAngle", F, 3.25, 45.0, 75.0, 0, 0.15)
	torch.manual_seed(15)
	ref_out = mod(base.clone())
	torch.manual_seed(15)
	z_angle = torch.randn_like(base)
	angle = torch.empty(B, 1).uniform_(math.radians(45.0), math.radians(75.0))
	z_gauss = torch.randn_like(base)
	u_mix = torch.rand(B, 1)
	u_angle = (angle - math.radians(45.0)) / (math.radians(75.0) - math.radians(45.0))
	check("noise.mix", ref_out, NO.gauss_elem_uniform_angle(base, z_gauss, z_angle, u_angle, u_mix, 3.25, 45.0, 75.0, 0.15), atol=2e-6)
	cases.append(dict(name="gauss_elem_uniform_angle", embed=base, z_gauss=z_gauss, z_angle=z_angle, u_angle=u_angle, u_mix=u_mix, vec_norm=3.25,
	                  angle_min=45.0, angle_max=75.0, mix_ratio=0.15, out=ref_out.clone()))

	# mean shift (train.py:1263-1265)
	e = base.clone()
	e.add_(shift)
	torch.nn.functional.normalize(e, dim=-1, out=e)
	check("noise.mean_shift", e, NO.mean_shift(base, shift), atol=1e-7)
	cases.append(dict(name="mean_shift", embed=base, shift=shift, out=e.clone()))
	return cases


def train_case():
	"""Short fixed-seed trajectory: the reference decoder driven by a loop restating train.py:1252-1286
	(dropout 0, noise off, accum 2, clip 1.0, AdamW(0.9,0.95) wd 0.1 on >=2-D params, constant lr)."""
	spec = SMALL
	seed = 400
	accum, steps, lr = 2, 6, 1.5e-3
	model, sd, _ = ref_model(spec, seed)
	model.train()
	params = [p for p in model.parameters() if p.requires_grad]
	groups = [{"params": [p for p in params if p.dim() < 2], "weight_decay": 0.0}, {"params": [p for p in params if p.dim() >= 2], "weight_decay": 0.1}]
	opt = torch.optim.AdamW(groups, lr=lr, betas=(0.9, 0.95), weight_decay=0.1)
	my_params = {k: v.clone() for k, v in sd.items() if k != "causality_mask"}
	my_state = {}
	batches, losses, norms, my_losses = [], [], [], []
	for step in range(1, steps + 1):
		mbs = [synth_batch(spec, 8, seed * 10 + step * accum + j) for j in range(accum)]
		batches.append(mbs)
		opt.zero_grad(set_to_none=True)
		step_loss = 0.0
		for embed, target, pad, weight in mbs:
			out = model(embed=embed, target=target, target_padding=pad, target_weight=weight, calc_loss=True, calc_correct=True, only_pred=False, guide_targets=None)
			loss = out[2] / out[3] / accum
			loss.backward()
			step_loss += float(loss)
		norms.append(float(torch.nn.utils.clip_grad_norm_(params, max_norm=1.0, error_if_nonfinite=True)))
		opt.step()
		losses.append(step_loss)
		# oracle side
		req = {k: v.clone().requires_grad_(True) for k, v in my_params.items()}
		req_sd = dict(req, causality_mask=sd["causality_mask"])
		total, _ = O.loss_for_step(req_sd, spec, mbs)
		total.backward()
		gn = O.clip_and_adamw(my_params, {k: v.grad for k, v in req.items()}, my_state, step, lr)
		my_losses.append(float(total))
		assert abs(float(gn) - norms[-1]) <= 1e-4 * max(1.0, norms[-1]), (float(gn), norms[-1])
	check("train.losses", torch.tensor(losses), torch.tensor(my_losses), atol=1e-5)
	final = {k: v.detach().clone() for k, v in model.state_dict().items() if k != "causality_mask"}
	for k, v in final.items():
		check(f"train.final.{k}", v, my_params[k], atol=2e-5, rtol=1e-4)
	return dict(spec=dataclasses.asdict(spec), seed=seed, accum=accum, lr=lr, batches=batches, losses=losses, grad_norms=norms,
	            final_checksum={k: (float(v.double().sum()), float(v.double().square().sum())) for k, v in final.items()},
	            final_samples={k: v.flatten()[:: max(1, v.numel() // 32)][:32].clone() for k, v in final.items()})


def gencfg_cases():
	names = ["greedy_k1_vnone_gn_t1_a0", "beam_k4_vnone_gn_t1_a0", "beam_k10_vnone_gp_t1_a0", "all_k5_vtok0.5_gr_t2_a0.5", "beam_k3_vtgt1_gn_t0.25_a1"]
	return [dict(name=n, fields={k: v for k, v in dataclasses.asdict(ref_infer.GenerationConfig.from_name(n)).items()}) for n in names]


def main():
	out = {
		"decoder_forward.pt": forward_cases(),
		"decoder_generate.pt": generate_cases(),
		"decoder_guided.pt": guided_cases(),
		"decoder_generate_all.pt": generate_all_cases(),
		"noise.pt": noise_cases(),
		"train_trajectory.pt": train_case(),
		"gencfg.pt": gencfg_cases(),
	}
	for fname, obj in out.items():
		path = os.path.join(HERE, fname)
		torch.save(obj, path)
		print(f"wrote {fname}: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
	main()
