"""Host-side logic that needs no GPU: data contracts, target tokenisation, generation-config names, schedules, decoder construction."""
import dataclasses
import math

import pytest
import torch

from conftest import load_golden
from helpers import make_decoder
from oracle import decoder_oracle as O


def test_generation_config_names_match_reference():
	from novic_amd import infer
	for c in load_golden("gencfg.pt"):
		cfg = infer.GenerationConfig.from_name(c["name"])
		assert dataclasses.asdict(cfg) == c["fields"]
	with pytest.raises(ValueError):
		infer.GenerationConfig.from_name("sample_k1")
	with pytest.raises(ValueError):
		infer.GenerationConfig.from_name("beam_k0_vnone_gn_t1_a0")
	with pytest.raises(ValueError):
		infer.GenerationConfig.from_name("beam_k4_vnone_gn_t0_a0")


def test_grad_accum_arithmetic():
	"""Loss weights of one optimizer step sum to 1 (the assertion of the reference's test_data_loader action, train.py:444-481)."""
	from novic_amd.embedding_dataset import GradAccum, LoaderInfo
	loader = list(range(11))
	info = LoaderInfo(num_workers=0, prefetch_factor=0, pin_memory=False, on_device=True, batch_size=8, batch_size_last=5, complete_batches=10, incomplete_batch=True,
	                  epoch_batches=11, epoch_samples=85, available_samples=85)
	ga = GradAccum(loader, info, accum_size=4, drop_last=True)
	assert (ga.loader_batches, ga.loader_steps, ga.complete_steps, ga.incomplete_step) == (8, 2, 2, False) and len(list(ga.loader())) == 8
	ga = GradAccum(loader, info, accum_size=4, drop_last=False)
	assert (ga.loader_batches, ga.loader_steps, ga.incomplete_batches, ga.incomplete_samples) == (11, 3, 3, 21)
	total, steps = 0.0, 0
	for i, _ in enumerate(ga.loader()):
		scale, step = ga.loss_scale(8 if i < 10 else 5)
		total += scale
		if step:
			assert abs(total - 1.0) < 1e-12
			total, steps = 0.0, steps + 1
	assert steps == 3


def test_data_config_normalisation():
	from novic_amd.embedding_dataset import DataConfig
	c = DataConfig.create(dict(use_weights=False, unit_weights=False, multi_target=False, multi_first=True, full_targets=False, fixed_multi_length=False, multi_length=3))
	assert c == DataConfig(False, True, False, False, True, True, 1)
	with pytest.raises(ValueError):
		DataConfig.create(dict(use_weights=True, unit_weights=True, multi_target=True, multi_first=False, full_targets=True, fixed_multi_length=True, multi_length=0))


def test_target_tokenisation_round_trip():
	from novic_amd import embedders
	toks = ["red", "panda", "fire", "truck", "sea", "lion", "ice", "cream", "cone", "unused"]
	emb = embedders.LocalVocabEmbedder(tokens=toks, embed_dim=32, check=True)
	nouns = ("red panda", "fire truck", "sea lion", "ice cream cone", "panda")
	from novic_amd.embedding_decoder import PrefixedIterDecoder
	kw = PrefixedIterDecoder.get_target_config_kwargs(with_start_token=True, with_end_token=False, compact_ids=False, fixed_token_length=False, auto_fixed_token_length=True, use_masks=True)
	tc = emb.create_target_config(nouns, **kw)
	assert tc.compact_ids and tc.end_token_id == 0 and tc.pad_token_id == 0 and tc.start_token_id is None
	assert tc.vocab_size == 1 + 9 and tc.token_length == 4  # 9 used content tokens + shared pad/end; longest noun = 3 tokens + END
	assert tc.compact_unmap.tolist() == [0] + sorted(emb.stoi[t] for t in toks[:-1])
	emb.configure_target(tc, nouns)
	ids, mask = emb.tokenize_target(nouns)
	assert ids.shape == (5, 4) and ids.dtype == torch.int64 and mask.dtype == torch.bool
	assert ids[0, 2] == 0 and not mask[0, 2] and mask[0, 3]  # END is not padding, what follows is
	assert emb.detokenize_target(ids) == list(nouns)
	assert emb.detokenize_target(ids.unsqueeze(1).repeat(1, 2, 1)) == [[n, n] for n in nouns]
	one, _ = emb.tokenize_target("sea lion")
	assert emb.detokenize_target(one.squeeze(0)) == "sea lion"
	with pytest.raises(ValueError):
		embedders.Embedder.create("openclip:apple/DFN5B-CLIP-ViT-H-14-378")


def test_chunk_schedule_matches_torch_schedulers():
	from novic_amd.train import ChunkSchedule

	class Opt:
		param_groups = [dict(lr=0.0)]
	for warm, kind, tmax, final in [(0, "cosine", 10, 0.0), (3, "cosine", 12, 1e-5), (5, "const", 1, 0.0)]:
		p = torch.nn.Parameter(torch.zeros(1))
		topt = torch.optim.SGD([p], lr=1.5e-3)
		tw = torch.optim.lr_scheduler.LinearLR(topt, start_factor=1 / (warm + 1), end_factor=1, total_iters=warm) if warm else None
		tc = torch.optim.lr_scheduler.CosineAnnealingLR(topt, T_max=tmax, eta_min=final) if kind == "cosine" else None
		mine = ChunkSchedule(Opt(), 1.5e-3, warm, kind, tmax, final)
		for chunk in range(tmax):
			assert abs(Opt.param_groups[0]["lr"] - topt.param_groups[0]["lr"]) < 1e-9 * 1.5e-3 + 1e-12, (warm, kind, chunk)
			topt.step()
			if tw: tw.step()
			if tc: tc.step()
			mine.step()


def test_decoder_construction_state_dict_and_flat_views():
	spec = O.DecoderSpec(embed_dim=32, vocab_size=53, token_length=6, hidden_dim=64, feedfwd_dim=16, num_layers=2, num_heads=4)
	model, sd = make_decoder(spec, seed=5)
	keys = list(model.state_dict().keys())
	assert set(keys) == set(sd.keys()) and "causality_mask" in keys and "transformer.layers.1.self_attn.in_proj_weight" in keys
	assert torch.equal(model.causality_mask, sd["causality_mask"])
	flat = model.flat_parameters()
	for k, p in model.named_parameters():
		assert torch.equal(p.data, sd[k])
		assert p.data.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr()  # every parameter is a view of the flat buffer
	assert model.num_decay_elements <= flat.numel()
	total, parts = model.get_num_params()
	assert total.used == sum(v.numel() for k, v in sd.items() if k != "causality_mask")
	assert parts["Transformer"].used == 2 * (3 * 64 * 64 + 64 * 64 + 2 * 16 * 64 + 2 * 64) + 64
	# fresh init follows the reference's balanced init statistics
	big = O.DecoderSpec(embed_dim=512, vocab_size=307, token_length=8)
	torch.manual_seed(20240511)  # the 3 % band below is ~3 sigma for the smallest tensors: pin the draw
	fresh, _ = make_decoder(big, seed=None)
	case = next(c for c in load_golden("decoder_forward.pt") if c["name"] == "default_pad")
	for k, (mean, std) in case["init_stats"].items():
		v = dict(fresh.named_parameters())[k].data
		if v.ndim == 1:
			assert abs(float(v.mean()) - mean) < 1e-6
		else:
			assert abs(float(v.std()) / std - 1) < 0.03, k
	with pytest.raises(Exception):
		fresh(torch.zeros(2, 512), None, None, None, False, False, True, None)  # CPU tensors: no fallback


def test_fused_adamw_state_travels_with_its_layout_table():
	"""Advisor, round 5: the flat moment buffers follow the decoder's flat parameter layout, which changed between versions of the package (storage rows of the logits
	matrix, appended bias rows) -- a state written for another layout must be placed by NAME or refused with a clear message, never copied raw."""
	from novic_amd import train as T
	spec = O.DecoderSpec(embed_dim=32, vocab_size=53, token_length=6, hidden_dim=64, feedfwd_dim=16, num_layers=2, num_heads=4)
	model, _ = make_decoder(spec, seed=5)
	opt = T.FusedAdamW(model, lr=1e-3)
	g = torch.Generator().manual_seed(1)
	opt.exp_avg.copy_(torch.randn(opt.exp_avg.shape, generator=g))
	opt.exp_avg_sq.copy_(torch.rand(opt.exp_avg_sq.shape, generator=g))
	opt.step_count = 7
	state = opt.state_dict()
	layout = state["layout"]
	assert [t[0] for t in layout] == list(model._offsets) and sum(t[3] for t in layout) == model._n_flat
	# (1) the same layout: a raw copy
	other = T.FusedAdamW(model, lr=1e-3)
	other.load_state_dict(state)
	assert torch.equal(other.exp_avg, opt.exp_avg) and torch.equal(other.exp_avg_sq, opt.exp_avg_sq) and other.step_count == 7
	# (2) another layout of the same tensors (the pre-round-5 one: the logits matrix with V rows of storage, everything behind it shifted; lists instead of tuples, as a
	# checkpoint that went through another serialiser would carry them): placed by name
	pos, old_layout, old_m, old_v = 0, [], [], []
	for name, o, shape, _ in layout:
		n = math.prod(shape)
		store = (n + 7) // 8 * 8
		old_layout.append([name, pos, list(shape), store])
		for src, dst in ((opt.exp_avg, old_m), (opt.exp_avg_sq, old_v)):
			dst.append(torch.cat([src[o:o + n], torch.zeros(store - n)]))
		pos += store
	old = dict(state, layout=old_layout, exp_avg=torch.cat(old_m), exp_avg_sq=torch.cat(old_v))
	assert old["exp_avg"].numel() != opt.exp_avg.numel()  # (53 rows of storage instead of 64)
	other = T.FusedAdamW(model, lr=1e-3)
	other.load_state_dict(old)
	for name, o, shape, _ in layout:
		n = math.prod(shape)
		assert torch.equal(other.exp_avg[o:o + n], opt.exp_avg[o:o + n]) and torch.equal(other.exp_avg_sq[o:o + n], opt.exp_avg_sq[o:o + n]), name
	# (3) no table and another size: refused with the reason, not an opaque copy_ error
	legacy = {k: v for k, v in old.items() if k != "layout"}
	with pytest.raises(ValueError, match="another parameter layout"):
		T.FusedAdamW(model, lr=1e-3).load_state_dict(legacy)
	# (4) a table of another model: refused
	bad = dict(state, layout=[("nonsense.weight", 0, (4,), 8)] + [tuple(t) for t in layout[1:]])
	with pytest.raises(ValueError, match="another model"):
		T.FusedAdamW(model, lr=1e-3).load_state_dict(bad)


def test_train_loop_config_arithmetic():
	from novic_amd.train import make_train_loop_config
	c = make_train_loop_config(run_dir="", batch_size=512, epoch_batches=1000, num_valid_targets=42919, accum_size=16, chunk_scale=50, max_epochs=18)
	assert c.chunk_batches == math.ceil(42919 * 50 / 512) and c.chunk_samples == c.chunk_batches * 512
	assert c.max_chunks == (18 * 1000) // c.chunk_batches
	assert abs(c.ewa_factor ** (4 * c.chunk_batches) - 0.5) < 1e-12


def test_image_transform_is_the_clip_preprocess():
	"""a3 (embedders.py:755-757, infer.py:293-299): get_image_transform on PIL images -- resize the shortest side with Pillow's bicubic resampling, centre crop,
	scale to [0, 1], CLIP mean / std -- against a tensor restatement of that pipeline (tests/helpers.py); host code, no GPU."""
	import numpy as np
	from PIL import Image
	from helpers import clip_preprocess_restated
	from novic_amd import clip_vit
	g = torch.Generator().manual_seed(4)
	for R, sizes in ((64, ((80, 120), (150, 90), (64, 64), (33, 200))), (224, ((300, 500), (224, 224), (180, 260)))):
		tf = clip_vit.NativeViT(clip_vit.ViTConfig(image_size=R, patch_size=32 if R == 224 else 16, width=128, layers=1, heads=4, embed_dim=64)).get_image_transform()
		for h, w in sizes:
			im = Image.fromarray((torch.rand(h, w, 3, generator=g) * 255).to(torch.uint8).numpy(), "RGB")
			got, want = tf(im), clip_preprocess_restated(torch.from_numpy(np.asarray(im).copy()), R)
			assert got.shape == (3, R, R) and got.dtype == torch.float32
			d = (got - want).abs()
			assert float(d.max()) <= 1.01 / 255 / 0.26 and float(d.mean()) <= 0.01 / 255 / 0.26, (R, h, w)  # Pillow's fixed-point coefficients: a grey level in rare pixels
	grey = Image.fromarray((torch.rand(70, 70, generator=g) * 255).to(torch.uint8).numpy(), "L")
	out = tf(grey)
	raw = [out[c] * sd + mu for c, (mu, sd) in enumerate(zip(clip_vit.CLIP_MEAN, clip_vit.CLIP_STD))]  # greyscale -> RGB: the same pixels under three normalisations
	assert out.shape == (3, 224, 224) and torch.allclose(raw[0], raw[1], atol=1e-6) and torch.allclose(raw[1], raw[2], atol=1e-6) and not torch.allclose(out[0], out[2], atol=1e-3)


def test_uint8_image_transform_is_the_fp32_transform_before_its_last_two_steps():
	"""get_image_transform(uint8=True) stops in front of ToTensor / Normalize: 3 x R x R uint8 pixels whose (u / 255 - mean) / std in fp32 -- the arithmetic the tower's first
	kernel applies on the device (novic_vit_im2col_u8) -- is the fp32 transform's output bit for bit."""
	from PIL import Image
	from novic_amd import clip_vit
	g = torch.Generator().manual_seed(14)
	vit = clip_vit.NativeViT(clip_vit.ViTConfig(image_size=64, patch_size=16, width=128, layers=1, heads=4, embed_dim=64))
	for pp in (None, dict(mean=(0.5, 0.5, 0.5), std=(0.5, 0.25, 0.125), interpolation="bilinear")):
		vit.preprocess = pp
		tf, tf8 = vit.get_image_transform(), vit.get_image_transform(uint8=True)
		mean, std = vit._pixel_norm()
		for h, w in ((80, 120), (64, 64), (33, 200)):
			im = Image.fromarray((torch.rand(h, w, 3, generator=g) * 255).to(torch.uint8).numpy(), "RGB")
			f, u = tf(im), tf8(im)
			assert u.dtype == torch.uint8 and u.shape == f.shape == (3, 64, 64) and u.is_contiguous()
			again = (u.float() / 255.0 - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)
			assert torch.equal(again, f), (pp, h, w)


def test_bench_refuses_what_it_cannot_measure():
	"""bench.py --gpus N started directly is a GPU-free parent that launches the ranks itself; on a node with fewer GPUs it must refuse (exit code 2) instead of
	measuring fewer GPUs under the wrong n_gpus, and a rank started without a GPU must fail loudly (no CPU fallback)."""
	import os
	import subprocess
	import sys
	import torch
	if torch.cuda.device_count() >= 2:
		import pytest
		pytest.skip("a multi-GPU node: the refusal path is for nodes with fewer GPUs than asked for")
	root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
	env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NOVIC_BENCH_REHEARSE")}
	r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=120)
	assert r.returncode == 2 and "GPU(s)" in r.stderr and '"metric"' not in r.stdout
	if not torch.cuda.is_available():
		r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=120)
		assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


def test_bench_takes_a_list_of_workgroup_budgets():
	"""`bench.py --persistent-cus A,B` (round 6): the first budget is the line's, one more timed region + instrumented pass runs per further one; 0 = no reservation."""
	import importlib.util
	import os
	import sys as _sys
	spec = importlib.util.spec_from_file_location("bench_for_args", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
	bench = importlib.util.module_from_spec(spec)
	_sys.modules["bench_for_args"] = bench  # (its dataclasses look their module up by name)
	old = _sys.argv
	try:
		spec.loader.exec_module(bench)
		_sys.argv = ["bench.py", "--gpus", "8", "--persistent-cus", "240,0,224"]
		a = bench.parse()
		assert a.persistent_cus == 240 and a.persistent_cus_list == [240, None, 224]
		_sys.argv = ["bench.py"]
		a = bench.parse()
		assert a.persistent_cus is None and a.persistent_cus_list == [] and a.gpus == 1 and a.repeats == 7
		_sys.argv = ["bench.py", "--persistent-cus", "208"]
		a = bench.parse()
		assert a.persistent_cus == 208 and a.persistent_cus_list == [208]
	finally:
		_sys.argv = old
		_sys.modules.pop("bench_for_args", None)
