"""The CPU oracle against the fixtures produced by the reference's own modules (tests/golden/make_golden.py)."""
import dataclasses

import pytest
import torch

from oracle import decoder_oracle as O
from oracle import noise_oracle as NO
from conftest import load_golden

FWD = load_golden("decoder_forward.pt")
GEN = load_golden("decoder_generate.pt")
NOISE = {c["name"]: c for c in load_golden("noise.pt")}


def close(a, b, atol=2e-5, rtol=1e-5):
	a, b = torch.as_tensor(a).float(), torch.as_tensor(b).float()
	fin = torch.isfinite(a)
	assert torch.equal(fin, torch.isfinite(b))
	torch.testing.assert_close(a[fin], b[fin], atol=atol, rtol=rtol)


@pytest.mark.parametrize("case", FWD, ids=[c["name"] for c in FWD])
def test_forward_matches_reference(case):
	spec = O.DecoderSpec(**case["spec"])
	sd = O.init_state_dict(spec, seed=case["seed"])
	out = O.forward(sd, spec, case["embed"], case["target"], case["padding"], case["weight"], case["calc_loss"], True, case["only_pred"])
	close(out[0], case["logits"])
	if case["out_padding"] is None:
		assert out[1] is None
	else:
		assert torch.equal(out[1], case["out_padding"])
	if case["calc_loss"]:
		close(out[2], case["loss_sum"], atol=1e-4)
		close(out[3], case["loss_basis"])
	assert torch.equal(out[4], case["correct"])
	if "bf16_logits" in case:  # reference under CPU bf16 autocast vs the oracle's bf16 emulation (loose: different rounding of SDPA internals)
		ob = O.forward(sd, spec, case["embed"], case["target"], case["padding"], case["weight"], True, False, case["only_pred"], bf16=True)
		scale = case["bf16_logits"].abs().max().item()
		assert (ob[0] - case["bf16_logits"]).abs().max().item() <= 0.04 * max(scale, 1.0)
		assert abs(float(ob[2]) - float(case["bf16_loss_sum"])) <= 0.02 * abs(float(case["bf16_loss_sum"]))


def test_init_statistics_match_reference_init():
	case = next(c for c in FWD if c["name"] == "default_pad")
	spec = O.DecoderSpec(**case["spec"])
	sd = O.init_state_dict(spec, seed=3)
	for k, (mean, std) in case["init_stats"].items():
		v = sd[k].float()
		if v.ndim == 1:  # norm weights are constants
			assert abs(float(v.mean()) - mean) < 1e-6 and float(v.std()) < 1e-6
		else:
			assert abs(float(v.mean()) - mean) < 5 * std / (v.numel() ** 0.5) + 1e-4
			assert abs(float(v.std()) / std - 1) < 0.03, k


def test_gradients_match_reference():
	case = next(c for c in FWD if c["name"] == "default_pad")
	spec = O.DecoderSpec(**case["spec"])
	sd = O.init_state_dict(spec, seed=case["seed"])
	sdg = {k: (v.clone().requires_grad_(True) if k != "causality_mask" else v) for k, v in sd.items()}
	out = O.forward(sdg, spec, case["embed"], case["target"], case["padding"], case["weight"], True, False, False)
	(out[2] / out[3]).backward()
	for k, n in case["grad_norms"].items():
		g = sdg[k].grad
		assert abs(float(g.norm()) - n) <= 1e-4 * max(n, 1e-3), k
		close(g.flatten()[:: max(1, g.numel() // 64)][:64], case["grad_samples"][k], atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("case", GEN, ids=[c["name"] for c in GEN])
def test_generate_matches_reference(case):
	spec = O.DecoderSpec(**case["spec"])
	sd = O.init_state_dict(spec, seed=case["seed"])
	if case["zero_end"]:
		sd["logits_linear.weight"][0].zero_()
	if case["kind"] == "greedy":
		ids, pad, logits, ls, lb, score = O.generate(sd, spec, case["embed"], True, case["calc_loss"], case["temperature"], case["length_alpha"], None, token_dtype=case["token_dtype"])
		assert ids.dtype == case["token_dtype"] and torch.equal(ids, case["ids"]) and torch.equal(pad, case["padding"])
		keep = ~pad
		close(logits[keep], case["logits"][keep], atol=5e-5)
		if case["calc_loss"]:
			close(ls, case["loss_sum"], atol=1e-4)
			close(lb, case["loss_basis"])
			close(score, case["score"], atol=5e-5)
		if case["zero_end"]:
			assert ids.shape[1] == spec.token_length - 1  # forced full-length decode
	else:
		ids, pad, score = O.generate_beam(sd, spec, case["embed"], case["topk"], case["temperature"], case["length_alpha"], token_dtype=case["token_dtype"])
		assert torch.equal(ids, case["ids"]) and torch.equal(pad, case["padding"])
		close(score, case["score"], atol=5e-5)
		assert torch.all(score[:, :-1] >= score[:, 1:])


def test_noise_matches_reference():
	c = NOISE["gauss_elem"]
	close(NO.gauss_elem(c["embed"], c["z"], c["vec_norm"]), c["out"], atol=1e-6)
	c = NOISE["gauss_vec"]
	close(NO.gauss_vec(c["embed"], c["z"], c["r"], c["vec_norm"]), c["out"], atol=1e-6)
	c = NOISE["uniform_angle"]
	close(NO.rotate(c["embed"], c["z"], c["angle"]), c["out"], atol=1e-6)
	c = NOISE["gauss_angle"]
	close(NO.rotate(c["embed"], c["z"], NO.gauss_angle_draw(c["r"], c["angle_std"], c["angle_max"])), c["out"], atol=1e-6)
	c = NOISE["gauss_elem_uniform_angle"]
	close(NO.gauss_elem_uniform_angle(c["embed"], c["z_gauss"], c["z_angle"], c["u_angle"], c["u_mix"], c["vec_norm"], c["angle_min"], c["angle_max"], c["mix_ratio"]), c["out"], atol=2e-6)
	c = NOISE["mean_shift"]
	close(NO.mean_shift(c["embed"], c["shift"]), c["out"], atol=1e-7)
	for c in NOISE.values():  # outputs are unit rows
		assert torch.allclose(c["out"].norm(dim=1), torch.ones(c["out"].shape[0]), atol=1e-5)


def test_training_trajectory_matches_reference():
	tr = load_golden("train_trajectory.pt")
	spec = O.DecoderSpec(**tr["spec"])
	sd = O.init_state_dict(spec, seed=tr["seed"])
	params = {k: v.clone() for k, v in sd.items() if k != "causality_mask"}
	state = {}
	for step, mbs in enumerate(tr["batches"], start=1):
		req = {k: v.clone().requires_grad_(True) for k, v in params.items()}
		total, _ = O.loss_for_step(dict(req, causality_mask=sd["causality_mask"]), spec, mbs)
		total.backward()
		gn = O.clip_and_adamw(params, {k: v.grad for k, v in req.items()}, state, step, tr["lr"])
		assert abs(float(total) - tr["losses"][step - 1]) < 1e-5
		assert abs(float(gn) - tr["grad_norms"][step - 1]) < 1e-4 * max(1.0, tr["grad_norms"][step - 1])
	for k, (s1, s2) in tr["final_checksum"].items():
		assert abs(float(params[k].double().sum()) - s1) < 1e-3 + 1e-5 * abs(s1), k
		assert abs(float(params[k].double().square().sum()) - s2) < 1e-3 + 1e-5 * abs(s2), k
		close(params[k].flatten()[:: max(1, params[k].numel() // 32)][:32], tr["final_samples"][k], atol=2e-5, rtol=1e-4)


GUIDED = load_golden("decoder_guided.pt")


@pytest.mark.parametrize("case", GUIDED, ids=[c["name"] for c in GUIDED])
def test_guided_generation_matches_reference(case):
	spec = O.DecoderSpec(**case["spec"])
	sd = O.init_state_dict(spec, seed=case["seed"])
	if case["kind"] == "forward":
		out = O.forward(sd, spec, case["embed"], case["target"], case["padding"], None, True, True, False, guide_targets=case["guide_targets"])
		assert torch.equal(out[4], case["correct"]) and not bool(out[4][1, 0])
		close(out[2], case["loss_sum"], atol=1e-4)
		return
	if case["kind"] == "greedy":
		ids, pad, logits, ls, lb, score = O.generate(sd, spec, case["embed"], True, True, case["temperature"], case["length_alpha"], None, guide_targets=case["guide_targets"],
		                                             guide_renorm=case["guide_renorm"])
		assert torch.equal(ids, case["ids"]) and torch.equal(pad, case["padding"])
		close(score, case["score"], atol=5e-5)
		close(ls, case["loss_sum"], atol=1e-4)
		# every decoded sequence is (a prefix of) one of the guide targets
		gt = {tuple(r.tolist()) for r in case["guide_targets"]}
		for row in ids.tolist():
			assert any(tuple(row) == g[:len(row)] for g in gt)
	else:
		g_arg = case["guide_targets"] if case["guided"] else None
		v_arg = (case["vocab_targets"] if case.get("vocab_targets") is not None else case["guide_targets"]) if case["vocab_prior"] else None
		ids, pad, score = O.generate_beam(sd, spec, case["embed"], case["topk"], case["temperature"], case["length_alpha"], guide_targets=g_arg, guide_renorm=case["guide_renorm"],
		                                  vocab_targets=v_arg, vocab_per_token=case["vocab_per_token"], vocab_scaler=case["vocab_scaler"])
		fin = torch.isfinite(case["score"])
		assert torch.equal(fin, torch.isfinite(score))
		close(score[fin], case["score"][fin], atol=5e-5)
		assert torch.equal(ids[fin], case["ids"][fin]) and torch.equal(pad[fin], case["padding"][fin])


ALL = load_golden("decoder_generate_all.pt")


@pytest.mark.parametrize("case", ALL, ids=[c["name"] for c in ALL])
def test_generate_all_matches_reference(case):
	spec = O.DecoderSpec(**case["spec"])
	sd = O.init_state_dict(spec, seed=case["seed"])
	v_arg = (case["vocab_targets"] if case.get("vocab_targets") is not None else case["guide_targets"]) if case["vocab_prior"] else None
	ids, pad, score = O.generate_all(sd, spec, case["embed"], case["topk"], case["temperature"], case["length_alpha"], case["guide_targets"], case["guide_renorm"], v_arg,
	                                 case["vocab_per_token"], case["vocab_scaler"])
	fin = torch.isfinite(case["score"])
	assert torch.equal(fin, torch.isfinite(score))
	close(score[fin], case["score"][fin], atol=5e-5)
	assert torch.equal(ids[fin], case["ids"][fin]) and torch.equal(pad[fin], case["padding"][fin])


TEXT = load_golden("text_forward.pt")


@pytest.mark.parametrize("case", TEXT, ids=[c["name"] for c in TEXT])
def test_text_oracle_matches_hf_fixture(case):
	from oracle import text_oracle as TO
	spec = TO.TextSpec(**case["spec"])
	sd = TO.init_state_dict(spec, seed=case["seed"])
	out = TO.encode_text(sd, spec, case["token_ids"], normalize=False)
	close(out, case["embeds_raw"], atol=2e-4 * max(1.0, float(case["embeds_raw"].abs().max())))
	close(TO.encode_text(sd, spec, case["token_ids"]), case["embeds"], atol=1e-5)


VIT_FULL = load_golden("vit_forward_full.pt")


def vit_full_images(spec, seed, B):
	"""The seeded image batch of tests/golden/make_golden_vit.py's full cases (one generator call per image, so a longer batch starts with the fixture's images)."""
	g = torch.Generator().manual_seed(seed)
	return torch.stack([torch.randn(3, spec.image_size, spec.image_size, generator=g) for _ in range(B)])


@pytest.mark.parametrize("case", VIT_FULL, ids=[c["name"] for c in VIT_FULL])
def test_vit_oracle_matches_hf_fixture_at_full_depth(case):
	"""The oracle tower at the depth and dims bench.py runs (ViT-B/32, all 12 layers; ViT-L/14 dims at depth 2) against transformers' CLIPVisionModelWithProjection."""
	from oracle import vit_oracle as VO
	spec = VO.ViTSpec(**case["spec"])
	sd = VO.init_state_dict(spec, case["seed"])
	images = vit_full_images(spec, case["seed"], case["batch"])
	assert abs(float(images.double().sum()) - case["image_checksum"]) < 1e-6  # the seeded generator reproduces the generator script's images
	out = VO.encode_image(sd, spec, images, normalize=False)
	close(out, case["embeds_raw"], atol=2e-4 * max(1.0, float(case["embeds_raw"].abs().max())))


def test_untied_embedding_and_logits_bias_variants_match_the_reference():
	"""Round 5 (tests/golden/make_golden_r5.py): the reference decoder with weight_tying=False and / or logits_bias=True -- its logits, loss, correct flags, parameter gradients,
	greedy and beam-4 outputs against the oracle, and the product class's constructor, parameter shapes, initial distributions and state-dict keys (incl. the reference's
	second name of the untied table, `embed_tokens.weight`) against the reference's."""
	from helpers import make_decoder, variant_extra_tensors
	for case in load_golden("decoder_variants_r5.pt"):
		spec = O.DecoderSpec(**case["spec"])
		sd = O.init_state_dict(spec, seed=case["seed"])
		sd.update(variant_extra_tensors(spec, case["seed"], case["untied"], case["bias"]))
		sdg = {k: (v.clone().requires_grad_(True) if k != "causality_mask" else v) for k, v in sd.items()}
		out = O.forward(sdg, spec, case["embed"], case["target"], case["padding"], None, True, True, False)
		torch.testing.assert_close(out[0], case["logits"], atol=2e-5, rtol=1e-5)
		assert torch.equal(out[1], case["out_padding"]) and torch.equal(out[4], case["correct"])
		torch.testing.assert_close(out[2], case["loss_sum"], atol=1e-4, rtol=1e-5)
		(out[2] / out[3]).backward()
		for k, n in case["grad_norms"].items():
			assert abs(float(sdg[k].grad.double().norm()) - n) <= 1e-4 * max(n, 1e-3), (case["name"], k)
			if case["grads"] is not None:
				torch.testing.assert_close(sdg[k].grad, case["grads"][k], atol=2e-5, rtol=1e-4)
		g = O.generate(sd, spec, case["embed"], False, True, 1.0, 0.0)
		assert torch.equal(g[0], case["greedy"][0]) and torch.equal(g[1], case["greedy"][1])
		b = O.generate_beam(sd, spec, case["embed"], 4, 1.0, 0.0)
		assert torch.equal(b[0], case["beam"][0]) and torch.equal(b[1], case["beam"][1])
		torch.testing.assert_close(b[2], case["beam"][2], atol=1e-4, rtol=1e-5)
		# the product class: same keys and shapes as the reference's state_dict, initial statistics of the new tensors as the reference draws them
		model, _ = make_decoder(spec, seed=None, untied=case["untied"], logits_bias=case["bias"])
		mine = {k: v for k, v in model.state_dict().items() if k != "causality_mask"}
		assert set(mine) == set(case["init_stats"]), (case["name"], set(mine) ^ set(case["init_stats"]))
		for k, (mean, std, shape) in case["init_stats"].items():
			assert tuple(mine[k].shape) == tuple(shape), (case["name"], k)
			if k in ("token_embedding.weight", "embed_tokens.weight", "logits_linear.bias") and mine[k].numel() >= 3000:
				assert abs(float(mine[k].std()) - std) <= 0.08 * std and abs(float(mine[k].mean()) - mean) <= 0.05 * std, (case["name"], k, float(mine[k].std()), std)
		full = dict(sd)
		if case["untied"]:
			full["embed_tokens.weight"] = sd["token_embedding.weight"]
		model.load_state_dict(full, strict=True)


def _arch_variant(case):
	"""(spec, state dict, constructor overrides) of a tests/golden/decoder_variants_r5b.pt case."""
	from helpers import arch_variant_tensors
	spec = O.DecoderSpec(**case["spec"])
	sw = case["switches"]
	from helpers import apply_extra
	extra = arch_variant_tensors(spec, case["seed"], layer_bias=sw.get("layer_bias", False), mlp_hidden=case["mlp_hidden"], mlp_bias=sw.get("mlp_hidden_bias", False),
	                             mlp_norm=sw.get("mlp_hidden_norm", False), rezero=sw.get("init_rezero_mode", "none"))
	sd = apply_extra(O.init_state_dict(spec, seed=case["seed"]), extra)
	return spec, sd, dict(sw, init_bias_zero=case["init_bias_zero"]), extra


def oracle_grad(sdg: dict, k: str, overrides: dict) -> torch.Tensor:
	"""Gradient of parameter k out of the oracle's state dict; ReZero 'perlayer' has ONE scalar under the names scale1 and scale2 (reference :1102-1103): their sum."""
	if overrides.get("init_rezero_mode") == "perlayer" and k.endswith(".scale1"):
		return sdg[k].grad + sdg[k[:-1] + "2"].grad
	return sdg[k].grad


def test_activation_bias_and_mlp_hidden_layer_variants_match_the_reference():
	"""Round 5 (tests/golden/make_golden_r5b.py): the reference decoder with layer_activation relu / tanh, layer_bias=True and a hidden layer in the prefix MLP (min / max /
	amean / gmean; bias, LayerNorm, relu / tanh) -- logits, loss, correct flags, parameter gradients, greedy and beam-4 outputs against the oracle; the product class's
	state-dict keys, shapes and the statistics of its initialisation against the reference's own (biases split the std with their weights when init_bias_zero is off)."""
	from helpers import make_decoder
	torch.manual_seed(20250)  # (the product's initial draws below: a fixed stream, so the statistical gates cannot flake)
	for case in load_golden("decoder_variants_r5b.pt"):
		spec, sd, overrides, extra = _arch_variant(case)
		sdg = {k: (v.clone().requires_grad_(True) if k != "causality_mask" else v) for k, v in sd.items()}
		out = O.forward(sdg, spec, case["embed"], case["target"], case["padding"], None, True, True, False)
		torch.testing.assert_close(out[0], case["logits"], atol=2e-5, rtol=1e-5)
		assert torch.equal(out[1], case["out_padding"]) and torch.equal(out[4], case["correct"])
		torch.testing.assert_close(out[2], case["loss_sum"], atol=1e-4, rtol=1e-5)
		(out[2] / out[3]).backward()
		for k, n in case["grad_norms"].items():
			mine = oracle_grad(sdg, k, overrides)
			assert abs(float(mine.double().norm()) - n) <= 1e-4 * max(n, 1e-3), (case["name"], k)
			if case["grads"] is not None:
				torch.testing.assert_close(mine, case["grads"][k], atol=2e-5, rtol=1e-4)
		g = O.generate(sd, spec, case["embed"], False, True, 1.0, 0.0)
		assert torch.equal(g[0], case["greedy"][0]) and torch.equal(g[1], case["greedy"][1])
		b = O.generate_beam(sd, spec, case["embed"], 4, 1.0, 0.0)
		assert torch.equal(b[0], case["beam"][0]) and torch.equal(b[1], case["beam"][1])
		torch.testing.assert_close(b[2], case["beam"][2], atol=1e-4, rtol=1e-5)
		# the product class
		model, _ = make_decoder(spec, seed=None, overrides=overrides)
		mine = {k: v for k, v in model.state_dict().items() if k != "causality_mask"}
		assert set(mine) == set(case["init_stats"]), (case["name"], set(mine) ^ set(case["init_stats"]))
		assert model.mlp_hidden_size == (case["mlp_hidden"] or None)
		for k, (mean, std, shape) in case["init_stats"].items():
			assert tuple(mine[k].shape) == tuple(shape), (case["name"], k)
			n = mine[k].numel()
			if std == 0.0 or n < 2:  # constants (LayerNorm weights / zero biases)
				assert float((mine[k] - mean).abs().max()) <= 1e-6 * max(1.0, abs(mean)), (case["name"], k, mean)
			elif n >= 1000:  # two independent draws: the means differ by ~ std sqrt(2 / n), the stds by ~ std / sqrt(n)
				assert abs(float(mine[k].std()) - std) <= max(0.05, 5 / n ** 0.5) * std and abs(float(mine[k].mean()) - mean) <= 7 * std / n ** 0.5, (case["name"], k, float(mine[k].std()), std)
			elif n >= 64:  # short bias vectors: a loose check that the scale is right
				assert 0.6 * std <= float(mine[k].std()) <= 1.5 * std, (case["name"], k, float(mine[k].std()), std)
		model.load_state_dict(sd, strict=True)
		assert (model.transformer.norm is None) == (not spec.layer_norm_first)
		n1 = sum(p.numel() for p in model.parameters() if p.ndim < 2)
		assert model.flat_parameters().numel() - model.num_decay_elements >= n1  # every 1-D tensor sits behind the weight-decayed ones (reference train.py:1103-1114)
		for k, p in model.named_parameters():
			o, _ = model._offsets[k]
			assert (o >= model.num_decay_elements) == (p.ndim < 2), k


def test_half_precision_tower_oracle_matches_transformers_in_float16():
	"""oracle.vit_oracle.encode_image_half restates clip's half-precision model (what the reference runs for 'openai:' embedders, embedders.py:488-489); pinned to
	transformers' CLIP vision tower cast to torch.float16 and run on the CPU (tests/golden/make_golden_vit_half.py).  The emulation of the HIP tower's own rounding points
	(`encode_image(bf16=True, half_stream=True)`: bf16 GEMM operands, half residual stream) stays within the tower tolerance of both."""
	from oracle import vit_oracle as VO
	for case in load_golden("vit_forward_half.pt"):
		if case["spec"]["layers"] > 2:
			continue  # (the 12-layer case: checked by the generator and on the GPU; a minute of CPU time here)
		spec = VO.ViTSpec(**case["spec"])
		sd = VO.init_state_dict(spec, case["seed"])
		g = torch.Generator().manual_seed(case["seed"])
		images = torch.stack([torch.randn(3, spec.image_size, spec.image_size, generator=g) for _ in range(case["batch"])])
		assert abs(float(images.double().sum()) - case["image_checksum"]) < 1e-6
		with torch.no_grad():
			half = VO.encode_image_half(sd, spec, images)
			emu = VO.encode_image(sd, spec, images, bf16=True, half_stream=True)
			full = VO.encode_image(sd, spec, images)
		assert float((half * case["embeds_half"]).sum(-1).min()) >= 0.99999 and float((half - case["embeds_half"]).norm(dim=-1).max()) <= 3e-3
		assert float((full * case["embeds_fp32"]).sum(-1).min()) >= 0.999999
		for ref in (half, full):
			assert float((emu * ref).sum(-1).min()) >= 0.9995 and float((emu - ref).norm(dim=-1).max()) <= 2e-2


def test_half_precision_text_tower_oracle_matches_transformers_in_float16():
	"""oracle.text_oracle.encode_text_half restates clip's half-precision text tower (the reference's 'openai:' embedders, embedders.py:488-489, :582-583); pinned to
	transformers' CLIP text tower cast to torch.float16 and run on the CPU (tests/golden/make_golden_text_half.py).  The emulation of the HIP tower's own rounding points
	(`encode_text(bf16=True, half_stream=True)`) stays within the tower tolerance of both precisions."""
	from oracle import text_oracle as TO
	for case in load_golden("text_forward_half.pt"):
		if case["spec"]["layers"] > 2:
			continue  # (the 12-layer case: checked by the generator and on the GPU)
		spec = TO.TextSpec(**case["spec"])
		sd = TO.init_state_dict(spec, case["seed"])
		ids = case["token_ids"]
		with torch.no_grad():
			half = TO.encode_text_half(sd, spec, ids)
			emu = TO.encode_text(sd, spec, ids, bf16=True, half_stream=True)
			full = TO.encode_text(sd, spec, ids)
		assert float((half * case["embeds_half"]).sum(-1).min()) >= 0.99999 and float((half - case["embeds_half"]).norm(dim=-1).max()) <= 4e-3
		assert float((full * case["embeds_fp32"]).sum(-1).min()) >= 0.999999
		for ref in (half, full):
			assert float((emu * ref).sum(-1).min()) >= 0.999 and float((emu - ref).norm(dim=-1).max()) <= 3e-2
