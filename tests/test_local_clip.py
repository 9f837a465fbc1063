"""OpenCLIPEmbedder / OpenAIEmbedder over local model files (reference embedders.py:438-594, :597-764; novic_amd/local_clip.py).

Fixtures (tests/golden/make_golden_openclip.py): two open_clip hub-repository directories under tests/golden/openclip_tiny/ (a CLIP BPE tokenizer; a BERT-style tokenizer
with strip_sep_token + 'canonicalize'), OpenAI's download layout under tests/golden/openai_tiny/, and the ids / embeddings transformers' own tokenizer and CLIP towers give
for the same vocabularies and weights.  Host side without a GPU; towers and a checkpoint whose `embedder_spec` is `openclip:...` through `NOVICModel` on the GPU.
Tolerance on the unit-norm embeddings (bf16 MFMA towers vs fp32 transformers): cosine >= 0.9995, per-row L2 error <= 2e-2."""
import dataclasses
import json
import os
import shutil

import pytest
import torch

from conftest import load_golden

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT_OC = os.path.join(HERE, "golden", "openclip_tiny")
ROOT_OA = os.path.join(HERE, "golden", "openai_tiny")
EXP = load_golden("openclip_tiny_expected.pt")


@pytest.fixture
def model_root(monkeypatch, tmp_path):
	"""$NOVIC_MODEL_ROOT with both layouts: ORG/NAME directories and openai/<file>.pt"""
	root = tmp_path / "models"
	shutil.copytree(ROOT_OC, root)
	shutil.copytree(ROOT_OA, root / "openai")
	monkeypatch.setenv("NOVIC_MODEL_ROOT", str(root))
	monkeypatch.setenv("HF_HOME", str(tmp_path / "no_hf_cache"))
	monkeypatch.setenv("OPENAI_HOME", str(tmp_path / "no_clip_cache"))
	return root


def test_openclip_spec_resolves_against_local_storage_only(model_root, tmp_path, monkeypatch):
	from novic_amd import embedders, local_clip
	c = EXP["clip"]
	e = embedders.Embedder.create("openclip:" + c["model_id"], load_model=False, device="cpu")
	assert isinstance(e, embedders.OpenCLIPEmbedder) and e.model_dir == str(model_root / c["model_id"]) and not e.is_model_loaded()
	# the configuration (and with it the hash embedding caches are keyed by, reference :262-278) is the reference's: {'model_id', 'model_config' = open_clip_config.json}
	cfg = e.get_configuration()
	assert cfg["model_id"] == c["model_id"] and cfg["model_config"] == c["config"] and cfg["class"] == "OpenCLIPEmbedder"
	# a directory path works as the name; the Hugging Face hub cache layout is searched too
	assert embedders.Embedder.create("openclip:" + e.model_dir, load_model=False, device="cpu").model_dir == e.model_dir
	snap = tmp_path / "hf" / "hub" / "models--someorg--somemodel" / "snapshots" / "abc123"
	shutil.copytree(e.model_dir, snap)
	monkeypatch.setenv("HF_HOME", str(tmp_path / "hf"))
	assert local_clip.resolve_model_dir("someorg/somemodel") == str(snap)
	with pytest.raises(ValueError, match="no network"):
		embedders.Embedder.create("openclip:apple/DFN5B-CLIP-ViT-H-14-378", load_model=False, device="cpu")


def test_openclip_tokenizer_and_special_tokens(model_root):
	from novic_amd import embedders
	c = EXP["clip"]
	e = embedders.Embedder.create("openclip:" + c["model_id"], load_model=False, device="cpu", check=True)
	sp = c["special"]
	assert (e.start_token_id, e.end_token_id, e.pad_token_id, e.vocab_size, e.context_length) == (sp["start"], sp["end"], sp["pad"], sp["vocab"], sp["context"])
	assert e.embed_dim == 64 and e.token_dtype == torch.int64 and not e.cased_tokens and not e.strip_sep_token and not e.tokenizer_clean
	d = e.tokenize(c["texts"], output_dict=True)
	assert torch.equal(d["input_ids"], c["input_ids"]) and torch.equal(d["attention_mask"], c["attention_mask"]) and d["text"] == c["texts"]
	assert torch.equal(e.tokenize(c["texts"]), c["input_ids"])
	assert e.detokenize(c["input_ids"]) == c["decoded"] and e.detokenize(c["input_ids"][1]) == c["decoded"][1]
	assert e.tokenize(c["texts"], max_tokens=4).shape[1] == 4
	# target configuration over this tokenizer (what NOVICModel does with the checkpoint's nouns)
	nouns = ["cat", "dog", "bird house", "the photo", "starling"]
	tc = e.create_target_config(nouns, with_start_token=False, with_end_token=True, compact_ids=True, fixed_token_length=False, auto_fixed_token_length=True, use_masks=True)
	e.configure_target(tc, nouns)
	ids, mask = e.tokenize_target(nouns)  # check=True: round-trips through detokenize_target inside
	assert e.detokenize_target(ids) == nouns and bool((ids[mask] == 0).all())


def test_openclip_bert_style_tokenizer_rules(model_root):
	"""embedders.py:633-645: no bos / eos -> start = [CLS], end = [SEP]; strip_sep_token -> every [SEP] becomes the pad id and the END id IS the pad id; pad aliases are
	checked; tokenizer_kwargs.clean = 'canonicalize' is applied before tokenising (:708-709)."""
	from novic_amd import embedders
	b = EXP["bert"]
	e = embedders.Embedder.create("openclip:" + b["model_id"], load_model=False, device="cpu")
	sp = b["special"]
	assert (e.start_token_id, e.end_token_id, e.pad_token_id, e.vocab_size, e.context_length) == (sp["start"], sp["end"], sp["pad"], sp["vocab"], sp["context"])
	assert e.strip_sep_token and e.tokenizer_clean and e.end_token_id == e.pad_token_id
	d = e.tokenize(b["texts"], output_dict=True)
	assert torch.equal(d["input_ids"], b["input_ids"]) and torch.equal(d["attention_mask"], b["attention_mask"])
	assert int((d["input_ids"] == e.tokenizer.sep_token_id).sum()) == 0
	assert e.detokenize(d["input_ids"]) == b["clean"]
	assert e.get_configuration()["model_config"]["preprocess_cfg"]["mean"] == [0.5, 0.5, 0.5]


def test_openai_tokenizer_is_clips_bpe(model_root):
	"""SimpleBPE (CLIP's simple_tokenizer.py restated) against transformers' CLIPTokenizer over the same merges: ids, int32, pad = END (reference :484), decode."""
	from novic_amd import embedders, local_clip
	o = EXP["openai"]
	e = embedders.Embedder.create("openai:" + o["model_name"], load_model=False, device="cpu")
	assert isinstance(e, embedders.OpenAIEmbedder)
	sp = o["special"]
	assert (e.start_token_id, e.end_token_id, e.pad_token_id, e.vocab_size, e.context_length) == (sp["start"], sp["end"], sp["pad"], sp["vocab"], sp["context"])
	assert e.token_dtype == torch.int32 and e.embed_dim == 512 and e.amp_mode is False and e.manual_amp_dtype == torch.float16
	ids = e.tokenize(o["texts"])
	assert ids.dtype == torch.int32 and torch.equal(ids, o["input_ids"])
	d = e.tokenize(o["texts"], output_dict=True)
	want_mask = torch.ones_like(ids)
	want_mask[:, 1:] = (ids[:, :-1] != e.pad_token_id).to(ids.dtype)
	assert torch.equal(d["attention_mask"], want_mask)
	assert e.detokenize(ids) == o["decoded"] and e.detokenize(ids[1]) == o["decoded"][1]
	assert e.tokenize("a photo of the cat and of the dog", max_tokens=5).tolist()[0][-1] == e.end_token_id and e.tokenize("a photo of the cat", max_tokens=5).shape[1] == 5
	cfg = e.get_configuration()
	assert cfg["model_name"] == "ViT-B/32" and cfg["model_checkpoint"].endswith("/ViT-B-32.pt")
	# byte-level pieces: punctuation, digits, non-ASCII (each digit is its own piece; unknown merges fall back to byte symbols) -- ids agree with transformers' tokenizer
	import transformers
	tk = transformers.AutoTokenizer.from_pretrained(os.path.join(ROOT_OC, "testorg", "CLIP-ViT-tiny-quickgelu"), local_files_only=True)
	bpe = local_clip.SimpleBPE.from_file(os.path.join(ROOT_OA, "merges.txt"))
	for text in ("the cat's 12 dogs!", "café photo -- of a house...", "  Bird   house\tstar "):
		assert bpe.encode(text) == tk(text, add_special_tokens=False)["input_ids"], text
	with pytest.raises(ValueError, match="no network"):
		embedders.Embedder.create("openai:ViT-L/14", load_model=False, device="cpu")
	with pytest.raises(NotImplementedError):
		embedders.Embedder.create("openai:RN50", load_model=False, device="cpu")


def test_preprocess_follows_the_models_preprocess_cfg():
	"""get_image_transform = open_clip's inference transform with the repository's mean / std (the BERT fixture says 0.5 / 0.5, as SigLIP repositories do)."""
	import numpy as np
	from PIL import Image
	from novic_amd import clip_vit
	g = torch.Generator().manual_seed(3)
	im = Image.fromarray((torch.rand(70, 100, 3, generator=g) * 255).to(torch.uint8).numpy(), "RGB")
	a = clip_vit.make_image_transform(64)(im)
	b = clip_vit.make_image_transform(64, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))(im)
	raw_a = a * torch.tensor(clip_vit.CLIP_STD).view(3, 1, 1) + torch.tensor(clip_vit.CLIP_MEAN).view(3, 1, 1)
	assert torch.allclose(raw_a, b * 0.5 + 0.5, atol=1e-6) and float(b.min()) >= -1 and float(b.max()) <= 1
	# Resize(64) on 100 x 70 gives int(64 * 100 / 70) = 91 columns (torchvision truncates), centre crop offset round((91 - 64) / 2) = 14 (banker's rounding of 13.5)
	wide = Image.fromarray(np.tile(np.arange(100, dtype=np.uint8)[None, :, None] * 2, (70, 1, 3)), "RGB")
	out = clip_vit.make_image_transform(64, (0, 0, 0), (1, 1, 1), "bilinear")(wide)[0, 0] * 255
	resized = np.asarray(wide.resize((91, 64), Image.BILINEAR))[0, :, 0].astype(np.float32)
	assert np.allclose(out.numpy(), resized[14:14 + 64], atol=1e-4)


@pytest.mark.gpu
def test_openclip_and_openai_towers_match_transformers(model_root):
	from novic_amd import embedders
	for spec, exp in (("openclip:" + EXP["clip"]["model_id"], EXP["clip"]), ("openai:ViT-B/32", EXP["openai"])):
		e = embedders.Embedder.create(spec, device="cuda", check=True)
		assert e.is_model_loaded()
		with e.inference_mode():
			txt = e.inference_text(exp["texts"]).cpu()
			img = e.inference_image(exp["images"]).cpu()
		for got, ref in ((txt, exp["text_embeds"]), (img, exp["image_embeds"])):
			assert got.shape == ref.shape and got.dtype == torch.float32 and torch.allclose(got.norm(dim=1), torch.ones(got.shape[0]), atol=1e-5)
			assert float((got * ref).sum(dim=1).min()) >= 0.9995, spec
			assert float((got - ref).norm(dim=1).max()) <= 2e-2, spec
		assert e.unload_model() and not e.is_model_loaded() and e.load_model() and e.is_model_loaded()
	b = embedders.Embedder.create("openclip:" + EXP["bert"]["model_id"], device="cuda")  # .bin weights, pooling at the arg-max id, 0.5 / 0.5 preprocessing
	with b.inference_mode():
		out = b.inference_text(EXP["bert"]["texts"])
	assert out.shape == (3, 64) and bool(torch.isfinite(out).all())
	assert b.image_tower.preprocess["mean"] == [0.5, 0.5, 0.5]


@pytest.mark.gpu
def test_checkpoint_with_an_openclip_embedder_spec_loads_without_an_override(model_root, tmp_path):
	"""What fails for every released checkpoint in round 2: `NOVICModel(checkpoint)` where cfg_flat.embedder_spec is 'openclip:ORG/NAME' -- no `embedder=` argument, the
	embedder (tokenizer, towers, preprocessing) comes from the spec through $NOVIC_MODEL_ROOT, images go in as PIL images."""
	from PIL import Image
	from novic_amd import embedders, embedding_dataset, embedding_decoder, infer, train, utils
	from test_gpu_infer_e2e import _cfg_flat
	spec = "openclip:" + EXP["clip"]["model_id"]
	nouns = ("cat", "dog", "bird house", "the photo", "starling", "house of the dog", "ant")
	emb = embedders.Embedder.create(spec, load_model=False, device="cuda")
	tc = emb.create_target_config(nouns, **embedding_decoder.PrefixedIterDecoder.get_target_config_kwargs(
		with_start_token=False, with_end_token=True, compact_ids=True, fixed_token_length=False, auto_fixed_token_length=True, use_masks=True))
	emb.configure_target(tc, nouns)
	cfg_flat = _cfg_flat(spec)
	torch.manual_seed(0)
	model = infer.load_decoder_model(utils.AttrDict.from_dict(cfg_flat), emb, embedding_dataset.DataConfig.single(), None).cuda()
	ckpt = train.save_train_checkpoint(cfg_flat, model, None, None, ("",) + nouns, 1, None, None, model_only=True, run_dir=str(tmp_path), chunk_id=1)
	nm = infer.NOVICModel(ckpt, batch_size=4, device="cuda")  # default gencfg: guided beam-10 over the checkpoint's nouns
	assert isinstance(nm.embedder, embedders.OpenCLIPEmbedder) and nm.gencfg.name == "beam_k10_vnone_gp_t1_a0"
	g = torch.Generator().manual_seed(5)
	images = [Image.fromarray((torch.rand(h, w, 3, generator=g) * 255).to(torch.uint8).numpy(), "RGB") for h, w in ((80, 120), (150, 90), (64, 64))]
	with nm:
		out = nm.classify_images(images)
		embeds = nm.embed_images(images)
	assert embeds.shape == (3, 64) and torch.allclose(embeds.norm(dim=1).cpu(), torch.ones(3), atol=1e-5)
	import math
	for preds, lps in zip(out.preds, out.logprobs):
		live = [p for p, l in zip(preds, lps) if math.isfinite(l)]
		assert len(live) == len(nouns) and set(live) == set(nouns)  # 7 nouns < 10 beams: guided search enumerates the noun set, the tail is dead
	# the embedding is the tower's on the reference preprocessing of the same PIL images
	tf = nm.embedder.get_image_transform  # (needs the loaded model: called inside `with nm`)
	with nm:
		with nm.embedder.inference_mode():
			direct = nm.embedder.inference_image(torch.stack([tf()(im) for im in images]))
		assert torch.equal(direct, nm.embed_images(images))


# ---- SigLIP (timm trunk + attention-pool head, non-causal text tower pooled at the last position; tests/golden/make_golden_siglip.py) ----

SIG = load_golden("siglip_expected.pt")


def test_siglip_oracle_matches_the_transformers_fixture():
	"""oracle/siglip_oracle.py against the embeddings transformers.SiglipModel gave for the fixture's weights (regenerated from the seeds: the oracle's init is seeded)."""
	from oracle import siglip_oracle as SO
	vs, ts = SO.SigLIPVisionSpec(**SIG["vision_spec"]), SO.SigLIPTextSpec(**SIG["text_spec"])
	sd = SO.init_vision_state_dict(vs, SIG["seeds"][0])
	sd.update(SO.init_text_state_dict(ts, SIG["seeds"][1]))
	img = SO.encode_image(sd, vs, SIG["images"])
	txt = SO.encode_text(sd, ts, SIG["input_ids_full"])
	assert float((img - SIG["image_embeds"]).abs().max()) <= 1e-5 and float((txt - SIG["text_embeds"]).abs().max()) <= 1e-5
	raw = SO.encode_image(sd, vs, SIG["images"], normalize=False)
	assert float((raw - SIG["image_embeds_raw"]).abs().max()) <= 2e-4 * float(SIG["image_embeds_raw"].abs().max())


def test_siglip_tokenizer_conventions(model_root):
	"""A sentencepiece-style tokenizer without a start token whose pad token IS the END token (the SigLIP tokenizer): start None, end = pad = </s>, 'canonicalize' cleaning,
	target configuration with compact ids and no start token to strip (embedders.py:185-192, :633-645)."""
	from novic_amd import embedders
	e = embedders.Embedder.create("openclip:" + SIG["model_id"], load_model=False, device="cpu")
	sp = SIG["special"]
	assert (e.start_token_id, e.end_token_id, e.pad_token_id, e.vocab_size, e.context_length) == (sp["start"], sp["end"], sp["pad"], sp["vocab"], sp["context"])
	assert e.tokenizer_clean and not e.strip_sep_token and e.embed_dim == 64
	d = e.tokenize(SIG["texts"], output_dict=True)
	assert torch.equal(d["input_ids"], SIG["input_ids"]) and torch.equal(d["attention_mask"], SIG["attention_mask"])
	assert e.detokenize(d["input_ids"]) == SIG["clean"]
	nouns = ["cat", "dog", "bird house", "the photo", "starling"]
	tc = e.create_target_config(nouns, with_start_token=False, with_end_token=True, compact_ids=True, fixed_token_length=False, auto_fixed_token_length=True, use_masks=True)
	e.configure_target(tc, nouns)
	ids, mask = e.tokenize_target(nouns)
	assert tc.start_token_id is None and tc.end_token_id == 0 and e.detokenize_target(ids) == nouns and bool((ids[mask] == 0).all())


def test_unsupported_siglip_variants_are_named(model_root, tmp_path):
	"""What the SigLIP towers do not build is refused by name when the model is loaded (the host side -- tokenizer, target configuration -- still works): a timm trunk with
	another pooling than the attention-pool head, heads wider than the attention kernels' 80 columns."""
	import json
	from safetensors.torch import load_file
	from novic_amd import siglip
	src = model_root / SIG["model_id"]
	sd = load_file(str(src / "open_clip_model.safetensors"))
	cfg = json.loads((src / "open_clip_config.json").read_text())
	bad = dict(cfg["model_cfg"], vision_cfg=dict(cfg["model_cfg"]["vision_cfg"], timm_pool="avg"))
	with pytest.raises(NotImplementedError, match="timm_pool"):
		siglip.build_towers(bad, sd)
	with pytest.raises(NotImplementedError, match="head_dim"):
		siglip.NativeSigLIPViT(siglip.SigLIPVisionConfig(image_size=224, patch_size=14, width=1152, layers=1, heads=8, mlp_dim=4304))
	assert siglip.vision_config_from(dict(timm_model_name="vit_base_patch16_siglip_224", image_size=64), sd).heads == 12  # the head count comes from the timm name when the config does not say
	# ViT-SO400M-14-SigLIP's dimensions are accepted (72-wide heads run zero-padded to 80, the MLP to the next multiple of 64) and the tanh GELU is read off the config
	so = siglip.SigLIPVisionConfig(image_size=224, patch_size=14, width=1152, layers=1, heads=16, mlp_dim=4304)
	from novic_amd import clip_text
	assert (clip_text.padded_head_dim(so.width // so.heads), clip_text.padded_mlp_dim(so.mlp_dim)) == (80, 4352)
	tanh = dict(cfg["model_cfg"]["vision_cfg"], act_kwargs={"approximate": "tanh"})
	assert siglip.vision_config_from(tanh, sd).gelu_tanh and not siglip.vision_config_from(cfg["model_cfg"]["vision_cfg"], sd).gelu_tanh


def test_zero_padding_of_heads_is_exact():
	"""clip_text.pad_rows_per_head / pad_cols_per_head: attention through heads zero-padded from 72 to 80 columns with the scale of 72 equals the unpadded attention (fp32, torch)."""
	from novic_amd import clip_text
	g = torch.Generator().manual_seed(5)
	H, D, Dp, W, N = 2, 72, 80, 144, 7
	x = torch.randn(N, W, generator=g)
	wqkv, bqkv, wo = torch.randn(3 * W, W, generator=g) * 0.1, torch.randn(3 * W, generator=g), torch.randn(W, W, generator=g) * 0.1

	def attn(qkv, d, scale):
		q, k, v = (qkv[:, c * H * d:(c + 1) * H * d].view(N, H, d).transpose(0, 1) for c in range(3))
		return (torch.softmax(q @ k.transpose(1, 2) * scale, dim=-1) @ v).transpose(0, 1).reshape(N, H * d)
	want = attn(x @ wqkv.T + bqkv, D, D ** -0.5) @ wo.T
	wp, bp, wop = clip_text.pad_rows_per_head(wqkv, 3, H, D, Dp), clip_text.pad_rows_per_head(bqkv, 3, H, D, Dp), clip_text.pad_cols_per_head(wo, H, D, Dp)
	assert wp.shape == (3 * H * Dp, W) and bp.shape == (3 * H * Dp,) and wop.shape == (W, H * Dp)
	got = attn(x @ wp.T + bp, Dp, D ** -0.5) @ wop.T
	assert float((got - want).abs().max()) <= 1e-5
	assert clip_text.pad_dim(wo, 1, 192).shape == (W, 192) and float(clip_text.pad_dim(wo, 1, 192)[:, W:].abs().max()) == 0.0


@pytest.mark.gpu
def test_siglip_b16_at_its_released_dimensions_against_the_oracle():
	"""`timm/ViT-B-16-SigLIP` as released (224 pixels, 196 tokens, width 768, 12 layers, 12 heads, MLP 3072, attention-pool head) at full depth, 16 images, seeded
	weights: the native tower against the oracle tower (pinned to transformers.SiglipModel by the small fixtures above) in fp32 and in its bf16 emulation of the
	kernels' rounding points."""
	from novic_amd import siglip
	from oracle import siglip_oracle as SO
	dims = dict(image_size=224, patch_size=16, width=768, layers=12, heads=12, mlp_dim=3072)
	vs = SO.SigLIPVisionSpec(**dims)
	sd = SO.init_vision_state_dict(vs, 71)
	tower = siglip.NativeSigLIPViT(siglip.SigLIPVisionConfig(**dims))
	tower.load_state_dict({k: v for k, v in sd.items() if k.startswith("visual.trunk.")})
	tower.cuda()
	g = torch.Generator().manual_seed(72)
	images = torch.randn(16, 3, 224, 224, generator=g)
	got = tower(images.cuda()).cpu()
	ref = SO.encode_image(sd, vs, images)
	assert got.shape == ref.shape == (16, 768)
	assert float((got * ref).sum(dim=1).min()) >= 0.9995 and float((got - ref).norm(dim=1).max()) <= 2e-2   # (the tolerances of the other tower tests)
	emu = SO.encode_image(sd, vs, images, bf16=True)
	assert float((got - emu).norm(dim=1).max()) <= 8e-3


@pytest.mark.gpu
def test_siglip_b16_at_the_bench_batch_against_the_oracle():
	"""The shape `bench.py` times (`infer_siglip_b16_images_per_s`): 256 images = 50 176 token rows, full depth.  At that row count every GEMM of a layer -- QKV, fc1 and
	the two fp32-residual ones (proj, fc2: 588 tiles each, whose 256-tile RESID_F32 epilogue carried the dropped-bias bug of round 3) -- runs on the 256-wide persistent
	tile, asserted through the launch counters; all 256 embeddings against the oracle tower (fp32; chunks of 32 images) and 64 of them against its bf16 emulation."""
	from novic_amd import ops, siglip
	from oracle import siglip_oracle as SO
	dims = dict(image_size=224, patch_size=16, width=768, layers=12, heads=12, mlp_dim=3072)
	vs = SO.SigLIPVisionSpec(**dims)
	sd = SO.init_vision_state_dict(vs, 71)
	tower = siglip.NativeSigLIPViT(siglip.SigLIPVisionConfig(**dims))
	tower.load_state_dict({k: v for k, v in sd.items() if k.startswith("visual.trunk.")})
	tower.cuda()
	g = torch.Generator().manual_seed(73)
	images = torch.randn(256, 3, 224, 224, generator=g)
	dev_images = images.cuda()
	ops.gemm_tile_counts(reset=True)
	got = tower(dev_images).cpu()
	counts = ops.gemm_tile_counts()
	assert counts["t256"] >= 4 * 12 + 1, counts  # QKV, proj, fc1, fc2 of every layer + the pooling head's K / V projection
	again = tower(dev_images).cpu()               # second call: capture + replay of the trunk's graph (round 4)
	third = tower(dev_images).cpu()
	assert torch.equal(got, again) and torch.equal(got, third)
	# every fourth image against the oracle tower in fp32 (an image is 196 rows, a 256-row tile holds rows of two: every tile of every launch is sampled; round 6: all 256
	# images through the CPU oracle were 25 of this test's 33 seconds, and the review asked for the GPU suite to stay under 300 s), 16 of them against its bf16 emulation
	pick = torch.arange(0, 256, 4)
	ref = torch.cat([SO.encode_image(sd, vs, images[pick[i:i + 32]]) for i in range(0, len(pick), 32)])
	assert got.shape == (256, 768) and ref.shape == (64, 768)
	assert float((got[pick] * ref).sum(dim=1).min()) >= 0.9995 and float((got[pick] - ref).norm(dim=1).max()) <= 2e-2
	pick16 = torch.cat((torch.arange(0, 8), torch.arange(248, 256)))
	emu = SO.encode_image(sd, vs, images[pick16], bf16=True)
	assert float((got[pick16] - emu).norm(dim=1).max()) <= 8e-3


@pytest.mark.gpu
def test_siglip_so400m_at_its_released_width_against_the_oracle():
	"""`timm/ViT-SO400M-14-SigLIP` at its RELEASED width -- 1152, 16 heads of 72 (zero-padded to 80 in the tower), MLP 4304 (padded to 4352), patch 14 at 224 pixels = 256
	tokens -- at depth 1 (plus the attention-pool head, which has the same dimensions), 64 images = 16 384 rows so that the 256-wide tiles run: the odd dimensions at the
	size where their padding, leading dimensions and tile edges (1152 = 4.5 tiles, 4352 = 17) are the released ones.  (Full depth is 27 such layers.)"""
	from novic_amd import ops, siglip
	from oracle import siglip_oracle as SO
	dims = dict(image_size=224, patch_size=14, width=1152, layers=1, heads=16, mlp_dim=4304)
	vs = SO.SigLIPVisionSpec(**dims)
	sd = SO.init_vision_state_dict(vs, 75)
	tower = siglip.NativeSigLIPViT(siglip.SigLIPVisionConfig(**dims))
	tower.load_state_dict({k: v for k, v in sd.items() if k.startswith("visual.trunk.")})
	tower.cuda()
	g = torch.Generator().manual_seed(76)
	images = torch.randn(64, 3, 224, 224, generator=g)
	ops.gemm_tile_counts(reset=True)
	got = tower(images.cuda()).cpu()
	counts = ops.gemm_tile_counts()
	assert counts["t256"] >= 4, counts
	ref = torch.cat([SO.encode_image(sd, vs, images[i:i + 32]) for i in range(0, 64, 32)])
	assert got.shape == ref.shape == (64, 1152)
	assert float((got * ref).sum(dim=1).min()) >= 0.9995 and float((got - ref).norm(dim=1).max()) <= 2e-2
	emu = SO.encode_image(sd, vs, images[:32], bf16=True)
	assert float((got[:32] - emu).norm(dim=1).max()) <= 8e-3


SIG_SO = load_golden("siglip_so_expected.pt")


def _so_specs():
	from oracle import siglip_oracle as SO
	vs, ts = SO.SigLIPVisionSpec(**SIG_SO["vision_spec"]), SO.SigLIPTextSpec(**SIG_SO["text_spec"])
	sd = SO.init_vision_state_dict(vs, SIG_SO["seeds"][0])
	sd.update(SO.init_text_state_dict(ts, SIG_SO["seeds"][1]))
	return SO, vs, ts, sd


def test_siglip_oracle_with_so400m_dimensions_matches_transformers():
	"""72-wide heads, MLP width 200, tanh GELU: the oracle against transformers.SiglipModel (hidden_act 'gelu_pytorch_tanh') on the fixture's seeded weights."""
	SO, vs, ts, sd = _so_specs()
	assert vs.width // vs.heads == 72 and vs.mlp_dim % 64 and vs.gelu_tanh and ts.gelu_tanh
	img, txt = SO.encode_image(sd, vs, SIG_SO["images"]), SO.encode_text(sd, ts, SIG_SO["input_ids_full"])
	assert float((img - SIG_SO["image_embeds"]).abs().max()) <= 1e-5 and float((txt - SIG_SO["text_embeds"]).abs().max()) <= 1e-5
	erf = SO.encode_image(sd, dataclasses.replace(vs, gelu_tanh=False), SIG_SO["images"])
	assert float((erf - SIG_SO["image_embeds"]).abs().max()) > 1e-5  # the activation is visible at this tolerance: the fixture does pin the tanh form


@pytest.mark.gpu
def test_siglip_so400m_dimensions_match_transformers(model_root, tmp_path):
	"""The SO400M-shaped model through `Embedder.create('openclip:...')`: the directory is rebuilt from the fixture's seeds and config (weights are not stored), the towers run
	with zero-padded 80-wide heads and a 256-wide MLP, and land on transformers' embeddings like the other towers."""
	import json
	import shutil
	from safetensors.torch import save_file
	from novic_amd import embedders, siglip
	SO, vs, ts, sd = _so_specs()
	d = model_root / "testorg" / "ViT-so-tiny-SigLIP"
	shutil.copytree(model_root / SIG["model_id"], d)  # the tokenizer files
	(d / "open_clip_config.json").write_text(json.dumps(SIG_SO["config"]))
	sd["logit_scale"], sd["logit_bias"] = torch.tensor(2.3), torch.tensor(-10.0)
	save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "open_clip_model.safetensors"))
	e = embedders.Embedder.create("openclip:" + SIG_SO["model_id"], device="cuda", check=True)
	assert isinstance(e.image_tower, siglip.NativeSigLIPViT) and e.image_tower.cfg.gelu_tanh and e.text_tower.cfg.gelu_tanh and e.text_tower.cfg.mlp_dim == 200
	with e.inference_mode():
		txt = e.inference_text(SIG_SO["texts"]).cpu()
		img = e.inference_image(SIG_SO["images"]).cpu()
	for got, ref in ((txt, SIG_SO["text_embeds"]), (img, SIG_SO["image_embeds"])):
		assert got.shape == ref.shape and float((got * ref).sum(dim=1).min()) >= 0.9995 and float((got - ref).norm(dim=1).max()) <= 2e-2
	emu_i, emu_t = SO.encode_image(sd, vs, SIG_SO["images"], bf16=True), SO.encode_text(sd, ts, SIG_SO["input_ids_full"], bf16=True)
	assert float((img - emu_i).norm(dim=1).max()) <= 8e-3 and float((txt - emu_t).norm(dim=1).max()) <= 8e-3


@pytest.mark.gpu
@pytest.mark.parametrize("config_pad_id", [True, False])
def test_siglip_towers_match_transformers(model_root, config_pad_id):
	"""config_pad_id False: the repository's text_cfg carries NO pad_id, as open_clip's own SigLIP configs -- the context padding of the non-causal text tower must then
	come from the tokenizer's pad id (1 = '</s>' here), not from a default of 0 (the texts are shorter than the context: padding takes part in the attention)."""
	from novic_amd import embedders, siglip
	if not config_pad_id:
		cfg_path = os.path.join(str(model_root), *SIG["model_id"].split("/"), "open_clip_config.json")
		cfg = json.load(open(cfg_path))
		assert cfg["model_cfg"]["text_cfg"].pop("pad_id") == 1
		json.dump(cfg, open(cfg_path, "w"))
	e = embedders.Embedder.create("openclip:" + SIG["model_id"], device="cuda", check=True)
	assert e.pad_token_id == 1 and e.text_tower.cfg.pad_id == 1
	assert isinstance(e.image_tower, siglip.NativeSigLIPViT) and e.is_model_loaded()
	with e.inference_mode():
		txt = e.inference_text(SIG["texts"]).cpu()
		img = e.inference_image(SIG["images"]).cpu()
		img2 = e.inference_image(SIG["images"]).cpu()
	for got, ref in ((txt, SIG["text_embeds"]), (img, SIG["image_embeds"])):
		assert got.shape == ref.shape and torch.allclose(got.norm(dim=1), torch.ones(got.shape[0]), atol=1e-5)
		assert float((got * ref).sum(dim=1).min()) >= 0.9995
		assert float((got - ref).norm(dim=1).max()) <= 2e-2
	assert torch.equal(img, img2)
	# against the oracle's bf16 emulation of the same rounding points
	from oracle import siglip_oracle as SO
	vs = SO.SigLIPVisionSpec(**SIG["vision_spec"])
	sd = SO.init_vision_state_dict(vs, SIG["seeds"][0])
	emu = SO.encode_image(sd, vs, SIG["images"], bf16=True)
	assert float((img - emu).norm(dim=1).max()) <= 8e-3
	tf = e.get_image_transform()
	assert e.image_tower.preprocess["mean"] == [0.5, 0.5, 0.5] and callable(tf)
