"""Size-independent properties of the decode path at the sizes `bench.py` measures (6 layers, d = 512, V = 6912, C_max = 12, 256 embeddings per batch,
greedy and beam-4, 11 forced steps) -- sizes at which the CPU oracle would take minutes, so the checks are the ones the domain offers without it:

  * determinism and batch-composition invariance: a sample's ids / score do not depend on which other samples share its batch (rows of the small-tile GEMMs,
    the per-sequence attention and the per-sample step kernels never mix), bit for bit, with and without the captured hipGraphs;
  * every emitted token is the arg-max of the logits that the ORDINARY (uncached, training-path) forward computes when teacher-forced on the emitted sequence,
    wherever that forward's top-2 margin exceeds the bf16 tolerance; the returned score is the sum of those log-probabilities (reference
    embedding_decoder.py:779-850: greedy = arg-max of log_softmax per step, score = sum of the chosen log-probs);
  * beam search (embedding_decoder.py:852-984) returns scores in descending order, its beams are distinct, every beam's score is the teacher-forced sum of its
    tokens' log-probabilities, and the best beam is at least as good as the greedy sequence (which is one of the candidates it keeps or beats)."""
import pytest
import torch

from oracle import decoder_oracle as O
from tests.helpers import make_decoder

pytestmark = pytest.mark.gpu
SPEC = O.DecoderSpec(embed_dim=512, vocab_size=6912, token_length=12)
B = 256


@pytest.fixture(scope="module")
def model():
	m, _ = make_decoder(SPEC, seed=4, device="cuda")
	with torch.no_grad():
		m.logits_linear.weight[0].zero_()   # END never wins: every sequence runs the full 11 steps, as in bench.py
	m.eval()
	return m


def _embeds(n, seed):
	g = torch.Generator().manual_seed(seed)
	return torch.nn.functional.normalize(torch.randn(n, SPEC.embed_dim, generator=g), dim=-1).cuda()


def _teacher_forced_logprobs(model, embed, ids):
	"""log_softmax of the uncached forward's logits at every step, fp32: [N][T][V]."""
	tgt = torch.cat((ids, torch.zeros(ids.shape[0], 1, dtype=ids.dtype, device=ids.device)), dim=1)
	with torch.no_grad():
		logits = model(embed=embed, target=tgt, target_padding=None, target_weight=None, calc_loss=False, calc_correct=False, only_pred=False, guide_targets=None)[0]
	return torch.log_softmax(logits[:, :ids.shape[1]].float(), dim=-1)


def test_greedy_full_size(model):
	e = _embeds(B, 1)
	with torch.no_grad():
		runs = [model.generate(e, False, True, 1.0, 0.0, None, None, False) for _ in range(3)]   # eager, capture, replay
		part = model.generate(e[64:160].contiguous(), False, True, 1.0, 0.0, None, None, False)
	ids, pad, score = runs[0][0], runs[0][1], runs[0][5]
	assert ids.shape == (B, SPEC.token_length - 1) and not bool(pad.any()) and bool((ids != 0).all())
	for r in runs[1:]:
		assert torch.equal(r[0], ids) and torch.equal(r[5], score)
	assert torch.equal(part[0], ids[64:160]) and torch.equal(part[5], score[64:160])   # batch composition does not matter
	lp = _teacher_forced_logprobs(model, e, ids)
	top2 = lp.topk(2, dim=-1).values
	margin = top2[..., 0] - top2[..., 1]
	clear = margin > 0.1
	assert float(clear.float().mean()) > 0.3   # (a random-init model sits near ties often; the check needs a population)
	assert torch.equal(lp.argmax(dim=-1)[clear], ids[clear])
	chosen = lp.gather(2, ids.unsqueeze(-1)).squeeze(-1)
	assert float((chosen - top2[..., 0]).abs().max()) <= 0.1 + 1e-3   # where the arg-max differs it is a near-tie: the emitted token is within the margin of the best
	torch.testing.assert_close(score, chosen.sum(dim=1), atol=6e-2, rtol=1e-2)


def test_beam4_full_size(model):
	e = _embeds(B, 2)
	with torch.no_grad():
		runs = [model.generate_beam(e, 4, 1.0, 0.0, None, False, 0.0, None, False) for _ in range(3)]
		part = model.generate_beam(e[:160].contiguous(), 4, 1.0, 0.0, None, False, 0.0, None, False)   # 640 rows: the same kernel choices as 1 024 (above 512 rows
		# the LayerNorm runs as its own launch instead of as a GEMM prologue -- equal up to bf16 ties, not bit for bit, so the comparison stays inside one regime)
		greedy = model.generate(e, False, True, 1.0, 0.0, None, None, False)
	ids, pad, score = runs[0]
	T = SPEC.token_length - 1
	assert ids.shape == (B, 4, T) and not bool(pad.any())
	for r in runs[1:]:
		assert torch.equal(r[0], ids) and torch.equal(r[2], score)
	assert torch.equal(part[0], ids[:160]) and torch.equal(part[2], score[:160])
	assert bool((score[:, :-1] >= score[:, 1:]).all())                                      # descending
	flat = ids.view(B, 4, T)
	for a in range(4):
		for b in range(a + 1, 4):
			assert bool((flat[:, a] != flat[:, b]).any(dim=1).all())                        # distinct beams
	lp = _teacher_forced_logprobs(model, e.repeat_interleave(4, dim=0), ids.reshape(B * 4, T))
	seq = lp.gather(2, ids.reshape(B * 4, T).unsqueeze(-1)).squeeze(-1).sum(dim=1).view(B, 4)
	torch.testing.assert_close(score, seq, atol=6e-2, rtol=1e-2)                            # a beam's score is the sum of its tokens' log-probabilities
	assert bool((score[:, 0] >= greedy[5] - 6e-2).all())                                    # the best beam is at least as good as the greedy sequence
