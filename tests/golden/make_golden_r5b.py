#!/usr/bin/env python3
"""Round 5 (second set) fixtures from the reference's own PrefixedIterDecoder for three more of its non-default switches: `layer_activation` relu / tanh
(embedding_decoder.py:306, utils.py:100-110), `layer_bias=True` (:317-325: biases on every linear layer and LayerNorm of the transformer) and a HIDDEN LAYER in the prefix
MLP (`mlp_hidden_layer` min / max / gmean with `mlp_hidden_bias`, `mlp_hidden_norm`, `mlp_hidden_activation`, :1243-1267) -- forward (logits, loss, basis, correct),
parameter gradients of the mean loss, greedy and beam-4 decoding, and the statistics of the reference's own initialisation.
Runs ONLY in the build container (imports /root/reference through make_golden.py's set-up):  python tests/golden/make_golden_r5b.py
Weights are not stored: oracle.decoder_oracle.init_state_dict(spec, seed) + helpers.arch_variant_tensors(spec, seed, ...)."""
import dataclasses
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import make_golden as MG  # noqa: E402,F401  (sets up sys.path for the reference and the oracle)
from make_golden import O, ref_decoder, FakeEmbedder, make_target_config, make_data_config, synth_batch, check, t2l  # noqa: E402
from helpers import arch_variant_tensors, apply_extra  # noqa: E402

SMALL = dict(embed_dim=32, vocab_size=53, token_length=6, hidden_dim=64, feedfwd_dim=16, num_layers=2, num_heads=4, mlp_seq_len=4)
WIDE = dict(embed_dim=64, vocab_size=307, token_length=7, hidden_dim=512, feedfwd_dim=128, num_layers=2, num_heads=8)  # the released layer shape on the general kernels

CASES = [  # name, spec kwargs, constructor switches, init_bias_zero
	("relu", dict(SMALL, layer_activation="relu"), dict(), True),
	("tanh_bias", dict(SMALL, layer_activation="tanh"), dict(layer_bias=True), False),
	("wide_bias", dict(WIDE), dict(layer_bias=True), False),
	("wide_relu", dict(WIDE, layer_activation="relu"), dict(), True),
	("mlp_gmean", dict(SMALL), dict(mlp_hidden_layer="gmean"), True),
	("mlp_max_norm_bias_tanh", dict(SMALL, mlp_hidden_activation="tanh"), dict(mlp_hidden_layer="max", mlp_hidden_bias=True, mlp_hidden_norm=True), False),
	("mlp_min_bias_relu", dict(SMALL, mlp_hidden_activation="relu"), dict(mlp_hidden_layer="min", mlp_hidden_bias=True), False),
	("mlp_amean_norm", dict(WIDE), dict(mlp_hidden_layer="amean", mlp_hidden_norm=True), True),
	# post-LN layers (layer_norm_first = False: x = norm(x + block(x)), no final norm, :315 / :325) and ReZero (:1086-1117), alone and together
	("postln", dict(SMALL, layer_norm_first=False), dict(), True),
	("postln_bias_relu", dict(SMALL, layer_norm_first=False, layer_activation="relu"), dict(layer_bias=True), False),
	("wide_postln", dict(WIDE, layer_norm_first=False), dict(), True),
	("rezero_perskip", dict(SMALL), dict(init_rezero_mode="perskip"), True),
	("rezero_perlayer_postln_bias", dict(SMALL, layer_norm_first=False), dict(init_rezero_mode="perlayer", layer_bias=True), False),
	("wide_rezero", dict(WIDE), dict(init_rezero_mode="perskip"), True),
]


def ref_variant(spec, seed, switches, bias_zero):
	cfg = dict(
		vocab_quant=False, num_end_loss=spec.num_end_loss, label_smoothing=spec.label_smoothing, hidden_dim=spec.hidden_dim, feedfwd_scale=f"{spec.feedfwd_dim}/{spec.hidden_dim}",
		mlp_hidden_layer="none", mlp_hidden_bias=False, mlp_hidden_norm=False, mlp_hidden_activation=spec.mlp_hidden_activation, input_dropout=0.0, num_layers=spec.num_layers,
		num_heads=spec.num_heads, layer_dropout=0.0, layer_activation=spec.layer_activation, layer_norm_first=spec.layer_norm_first, layer_bias=False, logits_bias=False, init_bias_zero=bias_zero,
		init_mlp_mode="balanced", init_mlp_unit_norm=False, init_tfrm_mode="balanced", init_tfrm_unit_norm=False, init_tfrm_unit_postnorm=True, init_tfrm_proj_layers=True,
		init_zero_norm=False, init_rezero_mode="none", mlp_seq_len=spec.mlp_seq_len, weight_tying=True, strictly_causal=spec.strictly_causal, enable_nested=False)
	cfg.update(switches)
	torch.manual_seed(seed + 1000)
	model = ref_decoder.PrefixedIterDecoder(embedder=FakeEmbedder(spec.embed_dim, make_target_config(spec.vocab_size, spec.token_length)), data_config=make_data_config(), **cfg)
	init_stats = {k: (float(v.float().mean()), float(v.float().std()) if v.numel() > 1 else 0.0, tuple(v.shape)) for k, v in model.state_dict().items() if k != "causality_mask"}
	hidden = model.embed_mlp.hidden_size or 0
	sd = O.init_state_dict(spec, seed=seed)
	apply_extra(sd, arch_variant_tensors(spec, seed, layer_bias=cfg["layer_bias"], mlp_hidden=hidden, mlp_bias=cfg["mlp_hidden_bias"], mlp_norm=cfg["mlp_hidden_norm"],
	                                     rezero=cfg["init_rezero_mode"]))
	model.load_state_dict(sd, strict=True)  # pins the key names: *.in_proj_bias, *.bias, embed_mlp.mlp.{0,1,2,3}.*, scale1 / scale2, no transformer.norm.* behind post-LN layers
	model.eval()
	return model, sd, init_stats, hidden


def main():
	out = []
	for idx, (name, spec_kw, switches, bias_zero) in enumerate(CASES):
		spec = O.DecoderSpec(**spec_kw)
		seed = 950 + idx
		model, sd, init_stats, hidden = ref_variant(spec, seed, switches, bias_zero)
		embed, target, pad, weight = synth_batch(spec, B=9, seed=seed)
		res = model(embed=embed, target=target, target_padding=pad, target_weight=None, calc_loss=True, calc_correct=True, only_pred=False, guide_targets=None)
		(res[2] / res[3]).backward()
		grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
		sdg = {k: (v.clone().requires_grad_(True) if k != "causality_mask" else v) for k, v in sd.items()}
		mine = O.forward(sdg, spec, embed, target, pad, None, True, True, False)
		(mine[2] / mine[3]).backward()
		for nm, a, b in zip(("logits", "padding", "loss_sum", "loss_basis", "correct"), res, mine):
			check(f"{name}.{nm}", a, b)
		shared = {k for k in sdg if k.endswith(".scale2")} if switches.get("init_rezero_mode") == "perlayer" else set()  # (perlayer: scale2 IS scale1 -- one parameter, two state-dict keys)
		assert set(grads) == {k for k in sdg if k != "causality_mask"} - shared, (sorted(grads), sorted(sdg))
		for k, gr in grads.items():
			want = sdg[k].grad if not (shared and k.endswith(".scale1")) else sdg[k].grad + sdg[k[:-1] + "2"].grad
			check(f"{name}.grad.{k}", gr, want, atol=2e-5, rtol=1e-4)
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
		small = name in ("tanh_bias", "mlp_max_norm_bias_tanh", "postln_bias_relu", "rezero_perlayer_postln_bias")  # full gradients for these; every case keeps the norms (all were checked against the oracle above)
		out.append(dict(name=name, spec=dataclasses.asdict(spec), seed=seed, switches=switches, init_bias_zero=bias_zero, mlp_hidden=hidden, init_stats=init_stats, embed=embed,
		                target=target, padding=pad, logits=t2l(res[0]), out_padding=t2l(res[1]), loss_sum=t2l(res[2]), loss_basis=t2l(torch.as_tensor(res[3])), correct=t2l(res[4]),
		                grads=grads if small else None, grad_norms={k: float(v.double().norm()) for k, v in grads.items()},
		                greedy=tuple(None if t is None else t2l(t) for t in gen), beam=tuple(t2l(t) for t in beam)))
		print(name, "ok: hidden", hidden, "keys", sorted(k for k in sd if "bias" in k or k.startswith("embed_mlp"))[:8])
	path = os.path.join(HERE, "decoder_variants_r5b.pt")
	torch.save(out, path)
	print(f"wrote decoder_variants_r5b.pt: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
	main()
