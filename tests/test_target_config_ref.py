"""SURVEY 8 a4 against the REFERENCE: tests/golden/target_config_ref.pt holds what the reference's own TransformersEmbedder produced on the local Hugging
Face fixture directory (tests/golden/make_golden_r2.py: create_target_config / tokenize_target / detokenize_target, embedders.py:169-254, :331-406);
the product's embedder must reproduce every field, map, id and mask exactly.  Plus the checkpoint-interop manifest the same generator wrote after the
reference's infer.load_decoder_model(strict) loaded a product-written checkpoint (infer.py:713-778, train.py:1450-1473).  CPU only."""
import dataclasses
import json
import os

import pytest
import torch

from conftest import GOLDEN, load_golden

CASES = load_golden("target_config_ref.pt")
DIR = os.path.join(GOLDEN, "hf_clip_tiny")


def _same(a, b):
	if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor):
		return isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor) and a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b)
	return type(a) is type(b) and a == b


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_target_config_and_tokenisation_equal_the_references(case):
	from novic_amd import embedders
	e = embedders.Embedder.create("transformers:" + DIR, load_model=False, device="cpu")
	tc = e.create_target_config(targets=case["nouns"], **case["kwargs"])
	e.configure_target(target_config=tc, target_vocab=case["nouns"])
	mine, ref = dataclasses.asdict(tc), case["target_config"]
	assert set(mine) == set(ref)
	for k in ref:
		assert _same(mine[k], ref[k]), (k, mine[k], ref[k])
	for nouns, ids_key, mask_key in ((case["nouns"], "ids_all", "mask_all"), (case["sub"], "ids_sub", "mask_sub")):
		ids, mask = e.tokenize_target(nouns)
		assert _same(ids, case[ids_key]), ids_key
		assert (mask is None and case[mask_key] is None) or _same(mask, case[mask_key]), mask_key
	ids_all = case["ids_all"]
	assert list(e.detokenize_target(ids_all)) == case["detok_all"]
	assert e.detokenize_target(ids_all[2]) == case["detok_row"]
	assert [list(r) for r in e.detokenize_target(torch.stack((ids_all[:4], ids_all[4:8]), dim=0))] == case["detok_nested"]


def test_product_checkpoint_was_loaded_strictly_by_the_reference():
	"""The generator's report (reference side ran in the build container) + the product side re-checked here: a checkpoint written by
	train.save_train_checkpoint carries exactly the dict keys / state-dict names and shapes the reference accepted."""
	import sys
	import tempfile
	sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
	from helpers import make_decoder
	from oracle import decoder_oracle as O
	from novic_amd import train as T
	rep = json.load(open(os.path.join(GOLDEN, "interop_report.json")))
	assert rep["reference_strict_load"] == "ok" and rep["product_strict_load_of_reference_state_dict"] == "ok"
	assert max(rep["reference_forward_vs_oracle_max_abs_err"].values()) < 1e-4
	spec = O.DecoderSpec(embed_dim=32, vocab_size=61, token_length=7, hidden_dim=64, feedfwd_dim=16, num_layers=2, num_heads=4)
	model, _ = make_decoder(spec, seed=None)
	with tempfile.TemporaryDirectory() as d:
		path = T.save_train_checkpoint(dict(T.default_train_config()), model, None, None, ("", "a", "b"), 1, None, None, model_only=True, run_dir=d, chunk_id=3)
		assert path.endswith(".model") and "ovod_chunk0003_" in os.path.basename(path)
		ckpt = torch.load(path, map_location="cpu", weights_only=False)
	assert sorted(ckpt) == rep["checkpoint_keys"]
	assert sorted(ckpt["model_state_dict"]) == rep["state_dict_keys"]
	assert {k: list(v.shape) for k, v in ckpt["model_state_dict"].items()} == rep["state_dict_shapes"]
	assert all(v.dtype == torch.float32 and v.device.type == "cpu" for v in ckpt["model_state_dict"].values())
