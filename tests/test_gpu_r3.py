"""Round-3 gates: oracle parity at the sizes bench.py measures, side-stream weight gradients, decode concurrency."""
import pytest
import torch

from helpers import make_decoder, synth_batch, to_dev
from oracle import decoder_oracle as O

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
	return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


def test_side_stream_weight_gradients_equal_the_main_stream_ones():
	"""overlap_wgrad=True runs the 256-wide weight-gradient launches (partial sums through scratch) on a side stream beside the main stream's K-split-tail GEMM, which
	takes scratch too: each stream must own its scratch (ops._splitk_ws is keyed by stream), else the partial sums of one overwrite the other's.  Every gradient of the
	overlapped pass must equal the serial pass bit for bit -- the fixed-order reductions make both deterministic."""
	spec = O.DecoderSpec(embed_dim=512, vocab_size=512, token_length=8, num_layers=2)
	model, _ = make_decoder(spec, seed=5, device="cuda")
	model.eval()
	batch = to_dev(*synth_batch(spec, 2560, seed=3))
	cls = type(model)
	prev = cls.overlap_wgrad
	res = {}
	try:
		for overlap in (False, True, True):
			cls.overlap_wgrad = overlap
			model.flat_grad().zero_()
			stats = model.forward_backward(*batch).clone()
			torch.cuda.synchronize()
			res.setdefault(overlap, []).append((stats, model.flat_grad().clone()))
	finally:
		cls.overlap_wgrad = prev
	base = res[False][0]
	assert float(base[1].abs().max()) > 0
	for stats, grad in res[True]:
		assert torch.equal(stats, base[0])
		assert torch.equal(grad, base[1])
