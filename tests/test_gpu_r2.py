"""Round-2 parity cases on the GPU (fixtures of tests/golden/make_golden_r2.py, generated from the reference's own modules):
configs[4] in small (F = 1024, M = 3 weighted targets: forward + parameter gradients), vocab_quant=True, the cfg-driven action_train with stop / resume,
and configs[0]'s path from image FILES through the image transform to label strings."""
import dataclasses
import json
import os

import pytest
import torch

from conftest import GOLDEN, load_golden
from helpers import clip_preprocess_restated as _torch_transform, make_decoder, to_dev
from oracle import decoder_oracle as O

pytestmark = pytest.mark.gpu
R2 = {c["name"]: c for c in load_golden("decoder_forward_r2.pt")}


def rel_l2(a, b):
	return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


def _check_forward(model, case, sd_for_oracle, spec):
	embed, target, pad, weight = to_dev(case["embed"], case["target"], case["padding"], case["weight"])
	with torch.no_grad():
		logits, out_pad, loss_sum, loss_basis, correct = model(embed=embed, target=target, target_padding=pad, target_weight=weight, calc_loss=True, calc_correct=True,
		                                                       only_pred=False, guide_targets=None)
	ref = case["logits"]
	assert logits.shape == ref.shape
	scale = max(1.0, float(ref.abs().max()))
	valid = ~case["out_padding"]
	assert float((logits.cpu() - ref)[valid].abs().max()) <= 3e-2 * scale
	ob = O.forward(sd_for_oracle, spec, case["embed"], case["target"], case["padding"], case["weight"], True, False, False, bf16=True)
	assert float((logits.cpu() - ob[0])[valid].abs().max()) <= 1.5e-2 * scale
	assert torch.equal(out_pad.cpu(), case["out_padding"])
	assert abs(float(loss_basis) - float(case["loss_basis"])) <= 1e-4 * max(1.0, float(case["loss_basis"]))
	assert abs(float(loss_sum) - float(case["loss_sum"])) <= 1e-2 * abs(float(case["loss_sum"]))
	top2 = ref.topk(2, dim=-1).values
	safe = valid & ((top2[..., 0] - top2[..., 1]) > 6e-2 * scale)
	assert torch.equal(correct.cpu()[safe], case["correct"][safe])


def test_multiset_f1024_weighted_forward_and_gradients():
	"""configs[4] against the reference: ViT-H/14 embedding width (F = 1024), 3 targets per embedding with descending weights, some of them zero-weighted and fully
	padded -- logits, loss statistics, correct flags and every parameter gradient (reference autograd, fp32) of the mean loss."""
	case = R2["multiset_f1024_m3_weighted"]
	spec = O.DecoderSpec(**case["spec"])
	model, sd = make_decoder(spec, seed=case["seed"], multi_target=True, use_weights=True, multi_length=3, device="cuda")
	model.eval()
	_check_forward(model, case, sd, spec)
	model.flat_grad().zero_()
	stats = model.forward_backward(*to_dev(case["embed"], case["target"], case["padding"], case["weight"]))
	torch.cuda.synchronize()
	# forward_backward's loss is loss_sum / loss_basis of the (single) group: the gradient of the reference's mean loss
	assert abs(float(stats[1, 0]) - float(case["loss_sum"])) <= 1e-2 * abs(float(case["loss_sum"]))
	for k, p in model.named_parameters():
		got = p.grad.cpu()
		assert abs(float(got.norm()) - case["grad_norms"][k]) <= 6e-2 * case["grad_norms"][k] + 1e-7, k
		samp = got.flatten()[:: max(1, got.numel() // 64)][:64]
		ref = case["grad_samples"][k]
		assert float((samp - ref).norm()) <= 6e-2 * float(ref.norm()) + 2e-2 * case["grad_norms"][k] * (64 / got.numel()) ** 0.5, k


def test_vocab_quant_forward():
	"""vocab_quant=True (reference embedding_decoder.py:228-278): the tied embedding has ceil(V / 64) * 64 rows, the unused ones are zero, never become logits and a
	checkpoint with non-zero values there is refused."""
	case = R2["small_vocab_quant"]
	spec = O.DecoderSpec(**case["spec"])
	model, _ = make_decoder(spec, sd=case["state_dict"], vocab_quant=True, device="cuda")
	assert model.logits_linear.weight.shape[0] == 64 and model.vocab_size_quant == 64
	model.eval()
	osd = dict(case["state_dict"])
	osd["logits_linear.weight"] = osd["logits_linear.weight"][:spec.vocab_size]
	_check_forward(model, case, osd, spec)
	bad = {k: v.clone() for k, v in case["state_dict"].items()}
	bad["logits_linear.weight"][spec.vocab_size + 1, 3] = 1.0
	with pytest.raises(ValueError):
		model.load_state_dict(bad, strict=True)
	# decode through the quantised table as well: ids stay below V
	with torch.no_grad():
		ids = model.generate(case["embed"].cuda(), False, False, 1.0, 0.0, None, None, False)[0]
	assert int(ids.max()) < spec.vocab_size


# ------------------------------------------------------------------------------------------------------------------------------
# action_train: cache file -> loader -> GradAccum -> noise -> model -> optimizer -> schedule -> loop, stop after 2 chunks, resume from the .train file
# ------------------------------------------------------------------------------------------------------------------------------

class _Stop(Exception):
	pass


def _train_cfg(tmp_path, **over):
	from novic_amd import train as T
	gold = load_golden("cache_batches.pt")
	spec_path = tmp_path / "embedder.json"
	if not spec_path.exists():
		spec_path.write_text(json.dumps(dict(tokens=gold["tokens"], embed_dim=gold["embed_dim"])))
	kw = dict(embedder_spec=f"local:{spec_path}", embedding_dataset=os.path.join(GOLDEN, "cache_single.bin"), strict_embedder=False, batch_size=4, accum_factor=2,
	          chunk_scale=1.6, max_chunks=4, max_epochs=0, hidden_dim=64, feedfwd_scale="1/4", num_layers=2, num_heads=4, input_dropout=0.1, layer_dropout=0.1,
	          noise_scheme="GaussElem", noise_vec_norm=0.5, save_every_min=1, save_every_max=2, save_top1_min=0.0, determ=True, determ_seed=3, lr_warmup=1, init_lr=2e-3)
	kw.update(over)
	return T.default_train_config(**kw)


def test_action_train_stops_and_resumes_on_the_same_trajectory(tmp_path):
	from novic_amd import train as T
	torch.manual_seed(5)
	full_dir, part_dir = tmp_path / "full", tmp_path / "part"
	infos_full = []
	full = T.action_train(_train_cfg(tmp_path), str(full_dir), False, log=lambda m: None, on_chunk=infos_full.append)
	C, S = full["train_loop_config"], full["train_loop_state"]
	assert C.chunk_batches == 4 and C.epoch_batches == 8 and C.max_chunks == 4 and S.chunk_id == 4 and S.batch_id == 16 and len(infos_full) == 4
	assert full["schedule"].chunks_done == 4 and infos_full[0]["lr"] < 2e-3 and infos_full[1]["lr"] == pytest.approx(2e-3 * 0.5 * (1 + __import__("math").cos(__import__("math").pi / 4)), rel=1e-6)

	# the same run, interrupted after its second chunk (a checkpoint is due there: save_every_max = 2) ...
	def stop_after_two(info):
		if info["chunk"] == 2:
			raise _Stop
	torch.manual_seed(5)
	with pytest.raises(_Stop):
		T.action_train(_train_cfg(tmp_path), str(part_dir), False, log=lambda m: None, on_chunk=stop_after_two)
	files = sorted(f for f in os.listdir(part_dir) if f.endswith(".train"))
	assert len(files) == 1 and "ovod_chunk0002_" in files[0]
	ckpt = torch.load(os.path.join(part_dir, files[0]), weights_only=False)
	rng = ckpt["novic_rng_state"]  # 4 optimizer steps so far: one forward/backward + one noise call each when the step's two micro-batches share a width (merged), two otherwise
	assert ckpt["train_loop_state"]["chunk_id"] == 3 and 4 <= rng["dropout_calls"] <= 8 and rng["noise_calls"] == rng["dropout_calls"] and "loader" in rng
	# ... and resumed from the .train file (weights, AdamW moments and step count, schedule position, loop / EWA state, dropout / noise / shuffle streams)
	infos_res = []
	res = T.action_train(_train_cfg(tmp_path, load_model=os.path.join(str(part_dir), files[0])), str(part_dir), False, log=lambda m: None, on_chunk=infos_res.append)
	assert [i["chunk"] for i in infos_res] == [3, 4] and res["optimizer"].step_count == full["optimizer"].step_count == 8
	for a, b in zip(infos_res, infos_full[2:]):
		assert a["lr"] == pytest.approx(b["lr"], rel=1e-9)
		assert a["loss"] == pytest.approx(b["loss"], rel=2e-3) and a["top1"] == pytest.approx(b["top1"], abs=0.02)
	S2 = res["train_loop_state"]
	assert (S2.chunk_id, S2.batch_id, S2.sample_id, S2.epoch_id) == (S.chunk_id, S.batch_id, S.sample_id, S.epoch_id)
	# same weights as the uninterrupted run, up to the order of the fp32 atomics in the embedding / LayerNorm-gain gradients (AdamW turns an element whose
	# gradient is at that noise level into an lr-sized step either way: bounded by 2 * steps * lr, and rare)
	wa, wb = res["model"].flat_parameters().cpu(), full["model"].flat_parameters().cpu()
	diff = (wa - wb).abs()
	assert float((diff > 2e-4).float().mean()) < 2e-3 and float(diff.max()) <= 2 * 4 * 2e-3 * 1.01, (float((diff > 2e-4).float().mean()), float(diff.max()))
	# a different seed really is a different trajectory (the equality above is not vacuous)
	other = T.action_train(_train_cfg(tmp_path, determ_seed=4, max_chunks=2), str(tmp_path / "other"), False, log=lambda m: None)
	assert float((other["model"].flat_parameters().cpu() - wb).abs().max()) > 1e-3


def test_action_train_mean_shift_and_config_checks(tmp_path):
	from novic_amd import train as T
	gold = load_golden("cache_batches.pt")
	shift = tmp_path / "modality_gap_x.json"
	cfg = _train_cfg(tmp_path, mean_shift=True, mean_shift_path=str(shift), max_chunks=1, noise_scheme="")
	shift.write_text(json.dumps(dict(cfg_embedder=dict(embedder_spec=cfg.embedder_spec), mean_shift=[0.01] * gold["embed_dim"])))
	out = T.action_train(cfg, str(tmp_path / "ms"), False, log=lambda m: None)
	assert out["train_loop_state"].chunk_id == 1 and out["noise"] is None
	shift.write_text(json.dumps(dict(cfg_embedder=dict(embedder_spec="other:thing"), mean_shift=[0.01] * gold["embed_dim"])))
	with pytest.raises(ValueError, match="embedder_spec"):
		T.action_train(cfg, str(tmp_path / "ms2"), False, log=lambda m: None)
	with pytest.raises(ValueError):
		T.default_train_config(no_such_key=1)
	msgs = []
	assert not T.check_loaded_config("hydra config", dict(a=1, b=2.0), dict(a=1, b=3.0, c=0), log=msgs.append) and any("'b'" in m for m in msgs) and any("'c'" in m for m in msgs)
	assert T.check_loaded_config("x", dict(a=torch.ones(2)), dict(a=torch.ones(2)), log=msgs.append)


# ------------------------------------------------------------------------------------------------------------------------------
# configs[0]: infer.py on image files (PIL -> get_image_transform -> tower -> decoder -> strings), CLI included
# ------------------------------------------------------------------------------------------------------------------------------

def test_image_files_to_labels_and_cli(tmp_path, capsys, monkeypatch):
	import numpy as np
	from PIL import Image
	from novic_amd import clip_vit, embedders, embedding_dataset, embedding_decoder, infer, train, utils
	from test_gpu_infer_e2e import NOUNS, TOKENS, _cfg_flat
	# three small image files of different shapes / modes (RGB landscape, RGB portrait, greyscale PNG)
	g = torch.Generator().manual_seed(4)
	paths = []
	for i, (h, w, mode) in enumerate(((80, 120, "RGB"), (150, 90, "RGB"), (64, 64, "L"))):
		arr = (torch.rand(h, w, 3 if mode == "RGB" else 1, generator=g) * 255).to(torch.uint8).numpy()
		# smooth the noise a little so that resampling differences between filters stay small relative to the content
		img = Image.fromarray(arr if mode == "RGB" else arr[:, :, 0], mode=mode).resize((w, h), Image.BILINEAR)
		p = tmp_path / f"img{i}.{'jpg' if i == 0 else 'png'}"
		img.save(p)
		paths.append(str(p))
	spec_path = tmp_path / "embedder.json"
	spec_path.write_text(json.dumps(dict(tokens=TOKENS, embed_dim=64)))
	emb = embedders.Embedder.create(f"local:{spec_path}", device="cuda")
	tc = emb.create_target_config(NOUNS, **embedding_decoder.PrefixedIterDecoder.get_target_config_kwargs(
		with_start_token=False, with_end_token=True, compact_ids=True, fixed_token_length=False, auto_fixed_token_length=True, use_masks=True))
	emb.configure_target(tc, NOUNS)
	cfg_flat = _cfg_flat(f"local:{spec_path}")
	torch.manual_seed(0)
	model = infer.load_decoder_model(utils.AttrDict.from_dict(cfg_flat), emb, embedding_dataset.DataConfig.single(), None).cuda()
	ckpt = train.save_train_checkpoint(cfg_flat, model, None, None, ("",) + NOUNS, 1, None, None, model_only=True, run_dir=str(tmp_path), chunk_id=1)
	vit = clip_vit.NativeViT(clip_vit.ViTConfig(image_size=64, patch_size=16, width=128, layers=2, heads=4, embed_dim=64), seed=2).cuda()
	emb2 = embedders.Embedder.create(f"local:{spec_path}", device="cuda")
	emb2.attach_image_tower(vit)
	nm = infer.NOVICModel(ckpt, gencfg="beam_k3_vnone_gp_t1_a0", batch_size=2, device="cuda", embedder=emb2)
	# static loaders + batching (reference infer.py:272-288)
	images = nm.load_images([os.path.basename(p) for p in paths], image_dir=str(tmp_path))
	assert [im.mode for im in images] == ["RGB"] * 3 and [im.size for im in images] == [(120, 80), (90, 150), (64, 64)]
	batches = nm.load_image_batches(paths)
	assert [len(b) for b in batches] == [2, 1]
	# the transform against a tensor restatement of resize-bicubic / centre-crop / normalise
	tf = nm.get_image_transform()
	for im in images:
		got = tf(im)
		want = _torch_transform(torch.from_numpy(np.asarray(im)), 64)
		assert got.shape == (3, 64, 64) and got.dtype == torch.float32
		assert float((got - want).abs().max()) <= 1.01 / 255 / 0.26 and float((got - want).abs().mean()) <= 0.05 / 255 / 0.26  # at most one grey level (Pillow's fixed-point coefficients)
	stacked = nm.transform_images(images)
	assert stacked.shape == (3, 3, 64, 64) and torch.equal(nm.transform_images(images[0])[0], stacked[0])
	with nm:
		from_files = nm.classify_images(images)
		from_tensor = nm.classify_images(stacked)
		one = nm.classify_image(images[1])
	assert from_files.preds == from_tensor.preds and from_files.logprobs == from_tensor.logprobs and one.preds[0] == from_files.preds[1]
	assert all(len(p) == 3 and set(p) <= set(NOUNS) for p in from_files.preds)  # guided over the model's nouns: every beam is a noun
	# (round 5) the same files as uint8 pixel batches, normalised by the tower's first kernel, and through classify_image_batches -- where the two caller batches of
	# equal shape share a tower launch and a decode call: the same embeddings, labels and scores per image
	with nm:
		by_batch = list(nm.classify_image_batches([images[:1], images[1:2], images[2:]]))
		nm.uint8_images = True
		try:
			assert nm.transform_images(images).dtype == torch.uint8
			as_u8 = nm.classify_images(images)
			by_batch_u8 = list(nm.classify_image_batches([images[:1], images[1:2], images[2:]]))
		finally:
			nm.uint8_images = False
	assert len(by_batch) == len(by_batch_u8) == 3
	for outs in (by_batch, by_batch_u8):
		assert tuple(p for o in outs for p in o.preds) == from_files.preds and tuple(p for o in outs for p in o.logprobs) == from_files.logprobs
		assert torch.equal(torch.cat([o.embeds for o in outs]), from_files.embeds)
	assert as_u8.preds == from_files.preds and as_u8.logprobs == from_files.logprobs and torch.equal(as_u8.embeds, from_files.embeds)
	# the CLI (reference infer.py:785-835) with the same checkpoint: NOVICModel is built from the flags; the embedder spec of the checkpoint has no image tower
	# of its own (local vocabulary), so the CLI's model gets the tower through the same hook a local deployment would use
	orig_init = infer.NOVICModel.__init__

	def init_with_tower(self, *a, **kw):
		kw.setdefault("embedder", emb2)
		orig_init(self, *a, **kw)
	monkeypatch.setattr(infer.NOVICModel, "__init__", init_with_tower)
	monkeypatch.setattr("sys.argv", ["infer.py", "--checkpoint", ckpt, "--image_dir", str(tmp_path), "--images"] + [os.path.basename(p) for p in paths] +
	                    ["--gencfg", "beam_k3_vnone_gp_t1_a0", "--batch_size", "2"])
	infer.main()
	lines = [ln for ln in capsys.readouterr().out.splitlines() if "-->" in ln]
	assert len(lines) == 3
	for ln, path, preds in zip(lines, paths, from_files.preds):
		assert ln.startswith(os.path.basename(path) + " --> ") and all(p in ln for p in preds) and "%" in ln
