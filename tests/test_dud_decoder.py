"""`DudDecoder` (reference embedding_decoder.py:454-610, loader branch infer.py:759-762): the zero-parameter baseline against outputs of the REFERENCE's class
(tests/golden/dud_decoder.pt, written by tests/golden/make_golden_r6.py in the build container).  Host code: runs on the CPU, exact for ids / flags, 1e-6 for the scores."""
import types

import pytest
import torch

from conftest import load_golden
from helpers import StubEmbedder, target_config

G = load_golden("dud_decoder.pt")


class _Embedder(StubEmbedder):
	def __init__(self, known=True):
		super().__init__(G["F"], target_config(G["V"], G["CMAX"]))
		self.known = known

	def tokenize_target(self, text):
		ids = G["dud"][text] if (self.known or text == "") else [-1, -1, 0]
		t = torch.tensor([ids], dtype=torch.int64)
		return t, torch.zeros_like(t, dtype=torch.bool)


def _data_config(multi_target=False, multi_first=False, use_weights=False, multi_length=1):
	from novic_amd import embedding_dataset
	return embedding_dataset.DataConfig.create(dict(use_weights=use_weights, unit_weights=True, multi_target=multi_target, multi_first=multi_first, full_targets=True,
	                                                fixed_multi_length=True, multi_length=multi_length))


def _model(known=True, dc=None, **over):
	from novic_amd import embedding_decoder
	return embedding_decoder.DudDecoder(embedder=_Embedder(known), data_config=dc or _data_config(), **dict(dict(G["cfg"], num_end_loss=1), **over))


def _same(a, b, tol=0.0):
	if a is None or b is None:
		assert a is None and b is None
		return
	assert a.shape == b.shape and a.dtype == b.dtype
	if tol:
		fin = torch.isfinite(b)
		assert torch.equal(torch.isfinite(a), fin) and torch.allclose(a[fin], b[fin], atol=tol, rtol=tol)
	else:
		assert torch.equal(a, b)


@pytest.mark.parametrize("case", G["forward"], ids=[c["name"] for c in G["forward"]])
def test_forward_cheats_exactly_as_the_reference(case):
	M = case["M"]
	dc = _data_config(multi_target=bool(M), multi_first=case["multi_first"], use_weights=case["weights"] and bool(M), multi_length=M or 1)
	model = _model(dc=dc, num_end_loss=case["num_end_loss"])
	B = case["target"].shape[1] if case["multi_first"] else case["target"].shape[0]
	embed = torch.zeros(B, G["F"])
	tgt = case["target"].clone()
	logits, padding, loss_sum, loss_basis, correct = model(embed, tgt, case["padding"], case["weight"], True, True, case["only_pred"], None)
	assert torch.equal(tgt, case["target"])  # the targets are read, never written
	assert logits.dtype == torch.float32 and logits.shape[-1] == G["V"]
	_same(logits.argmax(dim=-1), case["logits_argmax"])
	_same(logits.sum(dim=-1), case["logits_sum"])  # one-hot rows
	_same(padding, case["out_padding"])
	_same(loss_sum, case["loss_sum"]); _same(loss_basis, case["loss_basis"])
	_same(correct, case["correct"])
	off = model(embed, tgt, case["padding"], case["weight"], False, False, case["only_pred"], None)
	assert off[2] is None and off[3] is None and off[4] is None


def test_forward_needs_targets():
	with pytest.raises(ValueError, match="can only cheat"):
		_model()(torch.zeros(2, G["F"]), None, None, None, False, False, False, None)


@pytest.mark.parametrize("case", G["generate"], ids=lambda c: f"known{int(c['known'])}_ls{c['label_smoothing']}_c{int(c['collect'])}_l{int(c['loss'])}")
def test_generate_answers_unknown(case):
	model = _model(case["known"], num_end_loss=1, label_smoothing=case["label_smoothing"])
	got = model.generate(torch.zeros(5, G["F"]), case["collect"], case["loss"], case["tau"], case["alpha"], None, None, False)
	assert len(got) == 6
	for i, (a, b) in enumerate(zip(got, case["out"])):
		_same(a, b, tol=2e-6 if i in (3, 5) else 0.0)  # loss_sum / score: closed form here, a B x C x V log-softmax there


def test_beams_and_generate_all_carry_one_valid_result():
	for known, beam, all_ in zip((True, True, False, False), G["beam"], G["all"]):
		assert beam["known"] == known and all_["known"] == known
	for beam, all_ in zip(G["beam"], G["all"]):
		model = _model(beam["known"], num_end_loss=1)
		embed = torch.zeros(5, G["F"])
		for a, b in zip(model.generate_beam(embed, 3, 1.0, 0.0, None, False, 0.0, None, False), beam["out"]):
			_same(a, b)
		guide = torch.zeros(all_["guide_shape"], dtype=torch.int64)
		assert model.precompute_generate_all(0.0, None, False, 0.0, guide, False) is None
		for a, b in zip(model.generate_all(embed, 4, 1.0, 0.0, None, False, 0.0, guide, False), all_["out"]):
			_same(a, b)
	total, parts = _model().get_num_params()
	assert (total.total, list(parts)) == tuple(G["num_params"]) and total.to_str() == "0 params"


def test_loader_builds_it_from_a_checkpoint_configuration():
	"""infer.load_decoder_model dispatches on cfg.model (reference infer.py:716, :759-762); a DudDecoder checkpoint has an empty model_state_dict."""
	from novic_amd import embedding_decoder, infer
	cfg = types.SimpleNamespace(model="DudDecoder", num_end_loss=1, weight_tying=True, strictly_causal=False, enable_nested=False, **G["cfg"])
	model = infer.load_decoder_model(cfg, _Embedder(), _data_config(), dict(model_state_dict={}))
	assert type(model) is embedding_decoder.DudDecoder and len(list(model.parameters())) == 0 and model.state_dict() == {}
	with pytest.raises(ValueError, match="Unrecognised model class"):
		infer.load_decoder_model(types.SimpleNamespace(**dict(vars(cfg), model="EmbeddingVectorMLP")), _Embedder(), _data_config(), None)
