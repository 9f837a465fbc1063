#!/usr/bin/env python3
"""Round 5 fixtures from the reference's own PrefixedIterDecoder for two of its non-default switches: an UNTIED token embedding (weight_tying=False, embedding_decoder.py:251-254)
and a logits bias (logits_bias=True, :239-245), alone and together -- forward (logits, loss, basis, correct), parameter gradients of the mean loss, greedy and beam-4 decoding.
Runs ONLY in the build container (imports /root/reference through make_golden.py's set-up):  python tests/golden/make_golden_r5.py
Weights are not stored: oracle.decoder_oracle.init_state_dict(spec, seed) + the extra tensors drawn below from the same seed."""
import dataclasses
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (sets up sys.path for the reference and the oracle)
from make_golden import O, ref_decoder, FakeEmbedder, make_target_config, make_data_config, synth_batch, check, t2l  # noqa: E402

SPEC = O.DecoderSpec(embed_dim=32, vocab_size=53, token_length=6, hidden_dim=64, feedfwd_dim=16, num_layers=2, num_heads=4, mlp_seq_len=4)
WIDE = O.DecoderSpec(embed_dim=64, vocab_size=307, token_length=7, hidden_dim=512, feedfwd_dim=128, num_layers=2, num_heads=8)  # the released layer shape: the fused kernels' sizes


def extra_tensors(spec, seed, untied, bias):
	"""The tensors of the variant that init_state_dict does not know, from a generator of their own (helpers.variant_state_dict on the test side draws the same)."""
	g = torch.Generator().manual_seed(seed + 777)
	out = {}
	if untied:
		out["token_embedding.weight"] = torch.randn(spec.vocab_size, spec.hidden_dim, generator=g) / 2 ** 0.5
	if bias:
		out["logits_linear.bias"] = torch.randn(spec.vocab_size, generator=g) * 0.3
	return out


def ref_variant(spec, seed, untied, bias):
	cfg = dict(
		vocab_quant=False, num_end_loss=spec.num_end_loss, label_smoothing=spec.label_smoothing, hidden_dim=spec.hidden_dim, feedfwd_scale=f"{spec.feedfwd_dim}/{spec.hidden_dim}",
		mlp_hidden_layer="none", mlp_hidden_bias=False, mlp_hidden_norm=False, mlp_hidden_activation="gelu", input_dropout=0.0, num_layers=spec.num_layers, num_heads=spec.num_heads,
		layer_dropout=0.0, layer_activation="gelu", layer_norm_first=True, layer_bias=False, logits_bias=bias, init_bias_zero=False, init_mlp_mode="balanced", init_mlp_unit_norm=False,
		init_tfrm_mode="balanced", init_tfrm_unit_norm=False, init_tfrm_unit_postnorm=True, init_tfrm_proj_layers=True, init_zero_norm=False, init_rezero_mode="none",
		mlp_seq_len=spec.mlp_seq_len, weight_tying=not untied, strictly_causal=spec.strictly_causal, enable_nested=False)
	torch.manual_seed(seed + 1000)
	model = ref_decoder.PrefixedIterDecoder(embedder=FakeEmbedder(spec.embed_dim, make_target_config(spec.vocab_size, spec.token_length)), data_config=make_data_config(), **cfg)
	init_stats = {k: (float(v.float().mean()), float(v.float().std()) if v.numel() > 1 else 0.0, tuple(v.shape)) for k, v in model.state_dict().items() if k != "causality_mask"}
	sd = O.init_state_dict(spec, seed=seed)
	sd.update(extra_tensors(spec, seed, untied, bias))
	if untied:
		sd["embed_tokens.weight"] = sd["token_embedding.weight"]  # (the reference registers the table under both names: embedding_decoder.py:252-253)
	model.load_state_dict(sd, strict=True)  # pins the key names: token_embedding.weight (+ embed_tokens.weight), logits_linear.bias
	model.eval()
	return model, sd, init_stats


def main():
	out = []
	for idx, (name, spec, untied, bias) in enumerate([("untied", SPEC, True, False), ("bias", SPEC, False, True), ("untied_bias", SPEC, True, True), ("wide_untied_bias", WIDE, True, True)]):
		seed = 900 + idx
		model, sd, init_stats = ref_variant(spec, seed, untied, bias)
		embed, target, pad, weight = synth_batch(spec, B=9, seed=seed)
		res = model(embed=embed, target=target, target_padding=pad, target_weight=None, calc_loss=True, calc_correct=True, only_pred=False, guide_targets=None)
		(res[2] / res[3]).backward()
		grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
		sdg = {k: (v.clone().requires_grad_(True) if k != "causality_mask" else v) for k, v in sd.items()}
		mine = O.forward(sdg, spec, embed, target, pad, None, True, True, False)
		(mine[2] / mine[3]).backward()
		for nm, a, b in zip(("logits", "padding", "loss_sum", "loss_basis", "correct"), res, mine):
			check(f"{name}.{nm}", a, b)
		for k, gr in grads.items():
			check(f"{name}.grad.{k}", gr, sdg[k].grad, atol=2e-5, rtol=1e-4)
		with torch.no_grad():
			gen = model.generate(embed, False, True, 1.0, 0.0, None, None, False)
			beam = model.generate_beam(embed, 4, 1.0, 0.0, None, False, 0.0, None, False)
			mg = O.generate(sd, spec, embed, False, True, 1.0, 0.0)
			mb = O.generate_beam(sd, spec, embed, 4, 1.0, 0.0)
		for nm, a, b in zip(("ids", "padding"), gen[:2], mg[:2]):
			check(f"{name}.greedy.{nm}", a, b)
		check(f"{name}.greedy.score", gen[5], mg[5], atol=1e-4)
		for nm, a, b in zip(("ids", "padding", "score"), beam, mb):
			check(f"{name}.beam.{nm}", a, b, atol=1e-4)
		out.append(dict(name=name, spec=dataclasses.asdict(spec), seed=seed, untied=untied, bias=bias, init_stats=init_stats, embed=embed, target=target, padding=pad,
		                logits=t2l(res[0]), out_padding=t2l(res[1]), loss_sum=t2l(res[2]), loss_basis=t2l(torch.as_tensor(res[3])), correct=t2l(res[4]),
		                grads=grads if spec is SPEC else None, grad_norms={k: float(v.double().norm()) for k, v in grads.items()},  # (the wide case's 11 MB of gradients are checked against the oracle here and kept as norms)
		                greedy=tuple(None if t is None else t2l(t) for t in gen), beam=tuple(t2l(t) for t in beam)))
		print(name, "ok: keys", sorted(k for k in sd if k.startswith(("token_", "logits_"))))
	path = os.path.join(HERE, "decoder_variants_r5.pt")
	torch.save(out, path)
	print(f"wrote decoder_variants_r5.pt: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
	main()
