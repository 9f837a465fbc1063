"""TransformersEmbedder over a LOCAL Hugging Face CLIP directory (tests/golden/hf_clip_tiny, written by tests/golden/make_golden_hfclip.py with
transformers' own CLIPModel / CLIPTokenizer): the host side (tokenizer, special tokens, target configuration: reference embedders.py:767-907,
:169-254, :331-406) without a GPU, and both native towers against transformers' get_text_features / get_image_features embeddings on the GPU.
Tolerance on the unit-norm embeddings (bf16 MFMA towers vs fp32 transformers): cosine >= 0.9995, per-row L2 error <= 2e-2."""
import os

import pytest
import torch

from conftest import load_golden

HERE = os.path.dirname(os.path.abspath(__file__))
DIR = os.path.join(HERE, "golden", "hf_clip_tiny")
EXP = load_golden("hf_clip_tiny_expected.pt")


def _create(**kw):
	from novic_amd import embedders
	return embedders.Embedder.create("transformers:" + DIR, **kw)


def test_tokenizer_and_special_tokens_follow_the_directory():
	e = _create(load_model=False, device="cpu")
	sp = EXP["special"]
	assert (e.start_token_id, e.end_token_id, e.pad_token_id, e.vocab_size, e.context_length) == (sp["bos"], sp["eos"], sp["pad"], sp["vocab"], sp["context"])
	assert e.embed_dim == 64 and e.token_dtype == torch.int64 and e.embed_dtype == torch.float32 and not e.cased_tokens and not e.is_model_loaded()
	ids = e.tokenize(EXP["texts"])
	assert torch.equal(ids, EXP["input_ids"])
	d = e.tokenize(EXP["texts"], output_dict=True)
	assert torch.equal(d["input_ids"], EXP["input_ids"]) and torch.equal(d["attention_mask"], EXP["attention_mask"])
	assert e.detokenize(ids) == EXP["decoded"] and e.detokenize(ids[1]) == EXP["decoded"][1]
	assert e.tokenize(EXP["texts"], max_tokens=4).shape[1] == 4  # truncated to max_tokens, END kept (transformers' truncation)
	assert e.get_configuration()["model_id"] == DIR


def test_target_configuration_over_a_real_bpe_tokenizer():
	"""create_target_config / tokenize_target / detokenize_target (reference :169-254, :331-406) with compact ids over the BPE vocabulary."""
	e = _create(load_model=False, device="cpu")
	nouns = ["cat", "dog", "bird house", "the photo", "starling"]
	tc = e.create_target_config(nouns, with_start_token=False, with_end_token=True, compact_ids=True, fixed_token_length=False, auto_fixed_token_length=False, use_masks=True)
	e.configure_target(tc, nouns)
	assert tc.compact_ids and tc.end_token_id == 0 and tc.pad_token_id == 0 and tc.start_token_id is None and tc.vocab_size < e.vocab_size
	ids, mask = e.tokenize_target(nouns)
	assert ids.shape == mask.shape and ids.dtype == tc.token_dtype and int(ids.max()) < tc.vocab_size
	assert bool((ids[mask] == 0).all())  # everything behind END is padding = id 0
	assert e.detokenize_target(ids) == nouns
	# uncompacted: the tokenizer's own ids minus the start token
	raw = e.tokenize(nouns)
	assert torch.equal(tc.compact_unmap[ids[0][:2]], raw[0][1:3])


def test_hub_names_are_refused():
	from novic_amd import embedders
	with pytest.raises(ValueError):
		embedders.Embedder.create("transformers:openai/clip-vit-base-patch32", load_model=False, device="cpu")


@pytest.mark.gpu
def test_native_towers_match_transformers_embeddings():
	e = _create(device="cuda")
	assert e.is_model_loaded()
	with e.inference_mode():
		txt = e.inference_text(EXP["texts"]).cpu()
		img = e.inference_image(EXP["images"]).cpu()
	for got, ref in ((txt, EXP["text_embeds"]), (img, EXP["image_embeds"])):
		assert got.shape == ref.shape and got.dtype == torch.float32 and torch.allclose(got.norm(dim=1), torch.ones(got.shape[0]), atol=1e-5)
		assert float((got * ref).sum(dim=1).min()) >= 0.9995
		assert float((got - ref).norm(dim=1).max()) <= 2e-2
	assert e.unload_model() and not e.is_model_loaded() and e.load_model() and e.is_model_loaded()
