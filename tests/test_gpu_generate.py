"""GPU parity of greedy / beam decoding.

Sequence-level decisions of a random-init model sit on near-ties (top-2 logit gaps of a few 1e-2), so token-for-token equality with a
reference of different rounding is not a meaningful gate.  Parity is therefore established in two layers:
  1. exactness given identical logits: the step kernels are checked bit-for-bit against a torch restatement of the reference step on the
     SAME bf16 logits (tests/test_gpu_decode_steps.py), and here every chosen token is the arg-max of the logits the GPU itself produced;
  2. tolerance on the logits: teacher-forcing the CPU oracle (bf16 rounding points) on the GPU's own output must reproduce the per-step
     logits within 1.5e-2*scale and the sequence scores within 3e-2, and the GPU's best beam must be within tolerance of the oracle's own
     beam search optimum.
"""
import math
import os

import pytest
import torch

from conftest import load_golden
from helpers import make_decoder
from oracle import decoder_oracle as O

pytestmark = pytest.mark.gpu
GEN = load_golden("decoder_generate.pt")


def _model(case):
	spec = O.DecoderSpec(**case["spec"])
	sd = O.init_state_dict(spec, seed=case["seed"])
	if case["zero_end"]:
		sd["logits_linear.weight"][0].zero_()
	model, _ = make_decoder(spec, token_dtype=case["token_dtype"], sd=sd, device="cuda")
	model.eval()
	return spec, sd, model


def _teacher_forced_logits(sd, spec, embed, ids):
	"""Oracle logits for every position of the given sequences (B x T ids): position t is predicted from ids[:, :t]."""
	B, T = ids.shape
	tgt = torch.cat((ids.long(), torch.zeros(B, 1, dtype=torch.long)), dim=1)[:, :T]  # the last column is never fed, only its slot is predicted
	tgt = torch.cat((ids.long()[:, :T], torch.zeros(B, 1, dtype=torch.long)), dim=1)
	return O.forward(sd, spec, embed, tgt, None, None, False, False, False, bf16=True)[0][:, :T]


@pytest.mark.parametrize("case", [c for c in GEN if c["kind"] == "greedy"], ids=[c["name"] for c in GEN if c["kind"] == "greedy"])
def test_greedy(case):
	spec, sd, model = _model(case)
	tau, alpha = case["temperature"], case["length_alpha"]
	with torch.no_grad():
		ids, pad, logits, loss_sum, loss_basis, score = model.generate(embed=case["embed"].cuda(), collect_logits=True, calc_loss=True, temperature=tau, length_alpha=alpha,
		                                                              sample_weight=None, guide_targets=None, guide_renorm=False)
	ids, pad, logits, score = ids.cpu(), pad.cpu(), logits.cpu(), score.cpu()
	B, T = ids.shape
	G = spec.token_length - 1
	assert ids.dtype == case["token_dtype"] and pad.dtype == torch.bool and T <= G
	# (1) exact self-consistency with the GPU's own logits
	lg = logits.clone()
	lg[:, 0, 0] = float("-inf")
	am = lg.argmax(dim=2)
	keep = ~pad
	assert torch.equal(ids[keep].long(), am[keep]) and torch.all(ids[pad] == 0)
	done = torch.zeros(B, dtype=torch.bool)
	for t in range(T):  # a position is padded iff an END was produced before it
		assert torch.equal(pad[:, t], done)
		done = done | (am[:, t] == 0)
	assert T == G or bool(done.all())
	if T > 1:
		d2 = torch.zeros(B, dtype=torch.bool)
		for t in range(T - 1):
			d2 = d2 | (am[:, t] == 0)
		assert not bool(d2.all())  # no earlier exit was possible
	lsm = torch.log_softmax(logits / tau, dim=2).gather(2, ids.long().unsqueeze(2)).squeeze(2).masked_fill(pad, 0).sum(dim=1)
	n_tok = (T - pad.sum(dim=1)).clamp(min=1).float()
	torch.testing.assert_close(score, lsm * n_tok.pow(-alpha) if alpha != 0 else lsm, atol=2e-4, rtol=1e-4)
	nll = -torch.log_softmax(logits, dim=2).gather(2, ids.long().unsqueeze(2)).squeeze(2).masked_fill(pad, 0).sum()
	assert abs(float(loss_sum) - float(nll)) <= 2e-4 * max(1.0, float(nll)) and float(loss_basis) == float(keep.sum())
	# (2) oracle teacher-forced on the GPU's tokens reproduces the logits
	o_logits = _teacher_forced_logits(sd, spec, case["embed"], ids)
	scale = max(1.0, float(o_logits[keep].abs().max()))
	assert float((logits[keep] - o_logits[keep]).abs().max()) <= 1.5e-2 * scale
	# every GPU choice is within tolerance of the oracle's own optimum at that step
	ol = o_logits.clone()
	ol[:, 0, 0] = float("-inf")
	chosen = ol.gather(2, ids.long().unsqueeze(2)).squeeze(2)
	assert float((ol.max(dim=2).values - chosen)[keep].max()) <= 3e-2 * scale
	# informational agreement with the fp32 reference fixture
	if case["zero_end"] and case["ids"].shape == ids.shape:
		assert (ids == case["ids"]).float().mean().item() >= 0.7


@pytest.mark.parametrize("case", [c for c in GEN if c["kind"] == "beam"], ids=[c["name"] for c in GEN if c["kind"] == "beam"])
def test_beam(case):
	spec, sd, model = _model(case)
	tau, alpha, H = case["temperature"], case["length_alpha"], case["topk"]
	with torch.no_grad():
		ids, pad, score = model.generate_beam(embed=case["embed"].cuda(), topk=H, temperature=tau, length_alpha=alpha, vocab_targets=None, vocab_per_token=False,
		                                      vocab_scaler=0.0, guide_targets=None, guide_renorm=False)
	ids, pad, score = ids.cpu(), pad.cpu(), score.cpu()
	B, _, T = ids.shape
	assert ids.shape == (B, H, T) and pad.shape == ids.shape and score.shape == (B, H)
	assert torch.all(score[:, :-1] >= score[:, 1:]) and torch.all(torch.isfinite(score))
	assert torch.all(ids[pad] == 0) and not bool(pad[:, :, 0].any()) and torch.all(ids[:, :, 0] != 0)
	# padding starts right after the first END of a beam
	for t in range(1, T):
		ended = (ids[:, :, :t] == 0).any(dim=2)
		assert torch.equal(pad[:, :, t], ended)
	# beams of a sample are distinct sequences
	for b in range(B):
		assert len({tuple(r.tolist()) for r in ids[b]}) == H
	# teacher-forced oracle score of every returned beam
	flat = ids.view(B * H, T)
	o_logits = _teacher_forced_logits(sd, spec, case["embed"].repeat_interleave(H, dim=0), flat)
	lp = torch.log_softmax(o_logits / tau, dim=2).gather(2, flat.long().unsqueeze(2)).squeeze(2).masked_fill(pad.view(B * H, T), 0).sum(dim=1).view(B, H)
	n_tok = (T - pad.sum(dim=2)).clamp(min=1).float()
	ref_score = lp * n_tok.pow(-alpha) if alpha != 0 else lp
	torch.testing.assert_close(score, ref_score, atol=4e-2, rtol=1e-2)
	# the oracle's own beam search: same early-exit length class and an optimum within tolerance of ours
	margins = []
	o_ids, o_pad, o_score = O.generate_beam(sd, spec, case["embed"], H, tau, alpha, token_dtype=case["token_dtype"], bf16=True, margins=margins)
	assert float((o_score[:, 0] - score[:, 0]).abs().max()) <= 6e-2
	# MAX bound wherever the oracle's own search never came within 0.1 of a tie (its decisions are then forced for any kernel within the score tolerance): the
	# same beams in the same order, every score within 4e-2.  Random-init models leave few such samples (most candidates are near-ties by construction): the
	# trained fixtures of tests/test_gpu_generate_trained.py carry the exactness gate; for the rest the two searches must agree on average.
	safe = torch.stack(margins, dim=1).min(dim=1).values > 0.1
	if bool(safe.any()) and o_ids.shape[2] == T:
		assert torch.equal(ids[safe], o_ids[safe]) and torch.equal(pad[safe], o_pad[safe])
		assert float((o_score - score)[safe].abs().max()) <= 4e-2
	assert float((o_score - score).abs().mean()) <= 5e-2
	same = sum(len({tuple(r.tolist()) for r in ids[b]} & {tuple(r.tolist()) for r in o_ids[b][:, :T]}) for b in range(B)) / (B * H) if o_ids.shape[2] >= T else 1.0
	assert same >= 0.6, same


def test_graph_replay_matches_eager_and_uncached_forward():
	"""Call 1 runs the KV-cached steps eagerly, call 2 captures them into a hipGraph, call 3 replays it (new inputs): all must agree with each
	other and with logits recomputed by the uncached full forward (the reference's way of decoding)."""
	case = next(c for c in GEN if c["name"] == "beam4_default_full")
	spec, sd, model = _model(case)
	g = torch.Generator().manual_seed(5)
	e1 = torch.nn.functional.normalize(torch.randn(6, spec.embed_dim, generator=g), dim=-1).cuda()
	e2 = torch.nn.functional.normalize(torch.randn(6, spec.embed_dim, generator=g), dim=-1).cuda()
	with torch.no_grad():
		a1 = model.generate_beam(e1, 4, 1.0, 0.0, None, False, 0.0, None, False)
		a2 = model.generate_beam(e2, 4, 1.0, 0.0, None, False, 0.0, None, False)   # capture
		b1 = model.generate_beam(e1, 4, 1.0, 0.0, None, False, 0.0, None, False)   # replay
		b2 = model.generate_beam(e2, 4, 1.0, 0.0, None, False, 0.0, None, False)
		for x, y in ((a1, b1), (a2, b2)):
			assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) and torch.equal(x[2], y[2])
		g1 = model.generate(e1, True, True, 1.0, 0.0, None, None, False)
		g2 = model.generate(e2, True, True, 1.0, 0.0, None, None, False)
		h1 = model.generate(e1, True, True, 1.0, 0.0, None, None, False)
		assert torch.equal(g1[0], h1[0]) and torch.equal(g1[5], h1[5]) and torch.equal(g1[2], h1[2])
		# uncached: teacher-force the greedy output through the ordinary forward kernels; the per-step logits must match the cached ones
		ids = g1[0]
		tgt = torch.cat((ids, torch.zeros(ids.shape[0], 1, dtype=ids.dtype, device=ids.device)), dim=1)
		full = model(embed=e1, target=tgt, target_padding=None, target_weight=None, calc_loss=False, calc_correct=False, only_pred=False, guide_targets=None)[0]
	keep = ~g1[1]
	scale = max(1.0, float(full.abs().max()))
	assert float((full[:, :ids.shape[1]][keep] - g1[2][keep]).abs().max()) <= 1e-2 * scale


# ---- guided decoding / vocabulary priors (embedding_decoder.py:788-813, :877-879, :915-943, :969-975) ----
GUIDED = load_golden("decoder_guided.pt")


def _guided_seq_scores(logits, ids, pad, guide, tau, alpha, renorm, guided, prior, first_end_ban, vocab=None):
	"""Score of fully specified sequences under the reference's guided step rule, from teacher-forced logits (N x T x V): independent of any
	trie (per-sequence consistent-noun masks).  Returns (score N, on_guide N bool)."""
	N, T, V = logits.shape
	ok = torch.ones(N, guide.shape[0], dtype=torch.bool)
	vocab = guide if vocab is None else vocab
	vok = torch.ones(N, vocab.shape[0], dtype=torch.bool)
	score, on = torch.zeros(N), torch.ones(N, dtype=torch.bool)
	for t in range(T):
		live = ~pad[:, t]
		allowed = O.allowed_token_mask(guide, ok, t, V)
		lg = logits[:, t].float() / tau
		if guided and renorm:
			lg = lg.masked_fill(~allowed, float("-inf"))
		lp = torch.log_softmax(lg, dim=1)
		tok = ids[:, t].long()
		term = lp.gather(1, tok.unsqueeze(1)).squeeze(1)
		if prior is not None:
			col = vocab[:, t]
			for n in range(N):
				if live[n]:
					cons = col[vok[n]]
					if len(cons) == 0:
						term[n] = float("-inf")
						continue
					p = (1.0 / len(set(cons.tolist()))) if prior[0] else float((cons == tok[n]).sum()) / len(cons)
					term[n] = term[n] - prior[1] * math.log(p) if p > 0 else float("-inf")
		on &= ~live | allowed.gather(1, tok.unsqueeze(1)).squeeze(1)
		score += torch.where(live, term, torch.zeros(N))
		ok = ok & (guide[:, t].unsqueeze(0) == tok.unsqueeze(1))
		vok = vok & (vocab[:, t].unsqueeze(0) == tok.unsqueeze(1))
	n_tok = (T - pad.sum(dim=1)).clamp(min=1).float()
	return (score * n_tok.pow(-alpha) if alpha != 0 else score), on


@pytest.mark.parametrize("case", GUIDED, ids=[c["name"] for c in GUIDED])
def test_guided(case):
	spec = O.DecoderSpec(**case["spec"])
	sd = O.init_state_dict(spec, seed=case["seed"])
	model, _ = make_decoder(spec, token_dtype=torch.int64, sd=sd, device="cuda")
	model.eval()
	if case["kind"] == "forward":  # teacher-forced guided correctness (reference :756-763)
		with torch.no_grad():
			lg, pad, ls, lb, cor = model(embed=case["embed"].cuda(), target=case["target"].cuda(), target_padding=case["padding"].cuda(), target_weight=None, calc_loss=True,
			                             calc_correct=True, only_pred=False, guide_targets=case["guide_targets"].cuda())
			_, _, _, _, cor_free = model(embed=case["embed"].cuda(), target=case["target"].cuda(), target_padding=case["padding"].cuda(), target_weight=None, calc_loss=True,
			                             calc_correct=True, only_pred=False, guide_targets=None)
		cor, lg = cor.cpu(), lg.cpu()
		assert not bool(cor[case["padding"]].any()) and not bool(cor[1, 0])
		# exact given the GPU's own logits: the mask-based restatement on them
		V = lg.shape[2]
		tgt = case["target"].masked_fill(case["padding"], -1)
		for a in range(tgt.shape[0]):
			ok = torch.ones(case["guide_targets"].shape[0], dtype=torch.bool)
			for c in range(tgt.shape[1]):
				m = torch.full((V,), float("-inf"))
				m[case["guide_targets"][ok, c]] = 0
				assert bool(cor[a, c]) == (int((lg[a, c] + m).argmax()) == int(tgt[a, c])), (a, c)
				ok &= case["guide_targets"][:, c] == tgt[a, c]
		assert (cor == case["correct"]).float().mean().item() >= 0.9   # vs the fp32 fixture: only near-ties may differ
		return
	guide, tau, alpha, renorm = case["guide_targets"], case["temperature"], case["length_alpha"], case["guide_renorm"]
	embed = case["embed"]
	if case["kind"] == "greedy":
		with torch.no_grad():
			ids, pad, logits, loss_sum, loss_basis, score = model.generate(embed=embed.cuda(), collect_logits=True, calc_loss=True, temperature=tau, length_alpha=alpha,
			                                                              sample_weight=None, guide_targets=guide.cuda(), guide_renorm=renorm)
		ids, pad, logits, score = ids.cpu(), pad.cpu(), logits.cpu(), score.cpu()
		B, T = ids.shape
		keep = ~pad
		# exact self-consistency: every token is the arg-max of the GPU's own logits over what the still-consistent nouns allow
		ok = torch.ones(B, guide.shape[0], dtype=torch.bool)
		for t in range(T):
			allowed = O.allowed_token_mask(guide, ok, t, spec.vocab_size)
			am = logits[:, t].masked_fill(~allowed, float("-inf")).argmax(dim=1)
			assert torch.equal(ids[:, t][keep[:, t]], am[keep[:, t]]), t
			ok = ok & (guide[:, t].unsqueeze(0) == am.unsqueeze(1))
		my_score, on = _guided_seq_scores(logits, ids, pad, guide, tau, alpha, renorm, True, None, False)
		assert bool(on.all())
		torch.testing.assert_close(score, my_score, atol=2e-4, rtol=1e-4)
		nll = -torch.log_softmax(logits, dim=2).gather(2, ids.long().unsqueeze(2)).squeeze(2).masked_fill(pad, 0).sum()
		assert abs(float(loss_sum) - float(nll)) <= 2e-4 * max(1.0, float(nll)) and float(loss_basis) == float(keep.sum())
		o_logits = _teacher_forced_logits(sd, spec, embed, ids)
		scale = max(1.0, float(o_logits[keep].abs().max()))
		assert float((logits[keep] - o_logits[keep]).abs().max()) <= 1.5e-2 * scale
		# agreement with the fp32 reference fixture: same sequences wherever no near-tie is involved, scores within bf16 tolerance
		if case["ids"].shape == ids.shape:
			same = (ids == case["ids"]).all(dim=1)
			assert same.float().mean().item() >= 0.7
			torch.testing.assert_close(score[same], case["score"][same], atol=4e-2, rtol=1e-2)
		return
	H = case["topk"]
	g_arg = guide.cuda() if case["guided"] else None
	diff = case.get("vocab_targets") is not None
	vocab = case["vocab_targets"] if diff else guide
	v_arg = vocab.cuda() if diff else (g_arg if (case["vocab_prior"] and case["guided"]) else (guide.cuda() if case["vocab_prior"] else None))
	prior = (case["vocab_per_token"], case["vocab_scaler"]) if case["vocab_prior"] else None
	with torch.no_grad():
		ids, pad, score = model.generate_beam(embed=embed.cuda(), topk=H, temperature=tau, length_alpha=alpha, vocab_targets=v_arg, vocab_per_token=case["vocab_per_token"],
		                                      vocab_scaler=case["vocab_scaler"], guide_targets=g_arg, guide_renorm=renorm)
	ids, pad, score = ids.cpu(), pad.cpu(), score.cpu()
	B, _, T = ids.shape
	fin = torch.isfinite(score)
	assert torch.equal(fin, torch.isfinite(case["score"]))       # same number of live candidates per sample (fewer nouns than beams => -inf tail)
	assert torch.all(score[:, :-1] >= score[:, 1:]) and torch.all(ids[pad] == 0)
	for b in range(B):
		rows = [tuple(r.tolist()) for r, f in zip(ids[b], fin[b]) if f]
		assert len(set(rows)) == len(rows)
	flat, fpad = ids.view(B * H, T), pad.view(B * H, T)
	o_logits = _teacher_forced_logits(sd, spec, embed.repeat_interleave(H, dim=0), flat)
	ref_score, on = _guided_seq_scores(o_logits, flat, fpad, guide, tau, alpha, renorm, case["guided"], prior, True, vocab=vocab)
	ref_score, on = ref_score.view(B, H), on.view(B, H)
	assert bool(on[fin].all())                                       # every live beam spells (a prefix of) a noun of the set
	torch.testing.assert_close(score[fin], ref_score[fin], atol=4e-2, rtol=1e-2)
	# the oracle's own search at the same rounding points (a random-init model sits on near-ties: pruning may legitimately differ from fp32)
	o_ids, o_pad, o_score = O.generate_beam(sd, spec, embed, H, tau, alpha, bf16=True, guide_targets=guide if case["guided"] else None, guide_renorm=renorm,
	                                        vocab_targets=vocab if prior else None, vocab_per_token=case["vocab_per_token"], vocab_scaler=case["vocab_scaler"])
	assert torch.equal(torch.isfinite(o_score), fin)
	assert float((o_score[:, 0] - score[:, 0]).abs().max()) <= 6e-2 and float((o_score[fin] - score[fin]).abs().mean()) <= 5e-2
	gold = case["score"]
	assert float((gold[:, 0] - score[:, 0]).abs().max()) <= 6e-2
	assert float(((gold[fin] - score[fin]).abs() <= 6e-2).float().mean()) >= 0.8
	if case["ids"].shape == ids.shape:
		same = sum(len({tuple(r.tolist()) for r, f in zip(ids[b], fin[b]) if f} & {tuple(r.tolist()) for r, f in zip(case["ids"][b], fin[b]) if f}) for b in range(B)) / int(fin.sum())
		assert same >= 0.7, same


def test_guided_graph_replay_and_second_trie():
	case = next(c for c in GUIDED if c["name"] == "beam4_gp_small")
	spec = O.DecoderSpec(**case["spec"])
	model, _ = make_decoder(spec, token_dtype=torch.int64, sd=O.init_state_dict(spec, seed=case["seed"]), device="cuda")
	model.eval()
	guide = case["guide_targets"].cuda()
	e = case["embed"].cuda()
	with torch.no_grad():
		runs = [model.generate_beam(e, 4, 1.0, 0.0, None, False, 0.0, guide, False) for _ in range(3)]   # eager, capture, replay
		for r in runs[1:]:
			assert all(torch.equal(x, y) for x, y in zip(runs[0], r))
		gr = [model.generate(e, False, True, 1.0, 0.0, None, guide, True) for _ in range(3)]
		for r in gr[1:]:
			assert torch.equal(gr[0][0], r[0]) and torch.equal(gr[0][5], r[5])
		# a vocabulary prior over other nouns than the guide set runs too (second trie); replayed results are stable
		sub = guide[:10].clone()
		two = [model.generate_beam(e, 4, 1.0, 0.0, sub, False, 1.0, guide, False) for _ in range(3)]
		for r in two[1:]:
			assert all(torch.equal(x, y) for x, y in zip(two[0], r))
		# only nouns of the 10-noun vocabulary survive the prior: every live beam spells one of them
		allowed = {tuple(r.tolist()) for r in sub[:, :two[0][0].shape[2]].cpu()}
		live = torch.isfinite(two[0][2]).cpu()
		assert all(tuple(r.tolist()) in allowed for r in two[0][0].cpu()[live])


def test_generate_beam_returns_tensors_of_its_own():
	"""The outputs of generate_beam must not alias the decode session's buffers: a second call (same batch size and settings, other embeddings) used to rewrite the ids /
	padding a caller was still holding whenever the search ran its full length (the slice [:, :, :G] of the session buffer is the buffer)."""
	import dataclasses
	from helpers import make_decoder
	from oracle import decoder_oracle as O
	spec = O.DecoderSpec(embed_dim=32, vocab_size=53, token_length=6, hidden_dim=64, feedfwd_dim=16, num_layers=2, num_heads=4)
	model, _ = make_decoder(spec, seed=3, device="cuda")
	with torch.no_grad():
		model.logits_linear.weight[0].zero_()  # END never wins: full length
	model.eval()
	g = torch.Generator().manual_seed(0)
	e0, e1 = (torch.nn.functional.normalize(torch.randn(8, 32, generator=g), dim=-1).cuda() for _ in range(2))
	with torch.no_grad():
		a = model.generate_beam(e0, 3, 1.0, 0.0, None, False, 0.0, None, False)
		keep = tuple(t.clone() for t in a)
		b = model.generate_beam(e1, 3, 1.0, 0.0, None, False, 0.0, None, False)
		torch.cuda.synchronize()
	assert a[0].shape[2] == spec.token_length - 1
	assert all(torch.equal(x, y) for x, y in zip(a, keep)) and not torch.equal(a[0], b[0])


def test_graph_capture_keeps_the_garbage_collector_out():
	"""ops.graph_capture (round 5): a cyclic-garbage collection that starts in the middle of a hipGraph capture runs the finalisers of whatever cycles earlier work left behind
	-- decode sessions hang in a cycle with their model, holding graphs, events and pinned buffers -- inside the capture, where a forbidden HIP call aborts the process (seen
	once in a full suite run).  With every collector threshold at 1, unreachable cycles created DURING a capture are finalised only after it has ended."""
	import gc
	from novic_amd import ops
	seen = []

	class Trash:
		def __init__(self):
			self.me = self

		def __del__(self):
			seen.append(torch.cuda.is_current_stream_capturing())

	dev = torch.device("cuda", torch.cuda.current_device())
	x = torch.zeros(8, device=dev)
	g = torch.cuda.CUDAGraph()
	side = ops.capture_stream(dev)
	side.wait_stream(torch.cuda.current_stream(dev))
	old = gc.get_threshold()
	gc.set_threshold(1, 1, 1)
	try:
		with torch.cuda.stream(side):
			with ops.graph_capture(g, side):
				for _ in range(64):
					Trash()
				x += 1
		torch.cuda.current_stream(dev).wait_stream(side)
		assert gc.isenabled()
		gc.collect()
		assert len(seen) == 64 and not any(seen)  # all finalised, none of them inside the capture (with thresholds of 1 the collector runs at the first allocation behind it)
	finally:
		gc.set_threshold(*old)
	g.replay()
	torch.cuda.synchronize()
	assert float(x[0]) == 1.0  # (a capture records, it does not run: one replay = one increment)


@pytest.mark.gpu
def test_another_thread_may_free_hip_objects_while_a_capture_is_open():
	"""Round 6 (review of round 5: the abort was excluded by habit, not by construction).  What a second host thread may and may not do while a capture is open was MEASURED
	(tools/capture_free_probe.py): freeing pinned buffers, device tensors, events and streams is harmless in every capture mode; a page-locked ALLOCATION is an error under
	torch's default mode and fine under "thread_local"; DESTROYING A CAPTURED GRAPH aborts the process in every mode.  So `ops.graph_capture` captures thread-locally, the
	holders of captured graphs (decode sessions, tower slots) hand them to `ops.retire_graphs` -- parked while a capture is open, destroyed by whoever closes it -- and a
	collection another thread starts meanwhile waits (`ops._gc_guard`).  Here the other thread does all of it while this one holds a capture open."""
	import gc
	import threading
	import time
	from novic_amd import ops, tower_runtime
	dev = torch.device("cuda", torch.cuda.current_device())
	x = torch.zeros(8, device=dev)
	side = ops.capture_stream(dev)

	def captured(n):
		g = torch.cuda.CUDAGraph()
		side.wait_stream(torch.cuda.current_stream(dev))
		with torch.cuda.stream(side):
			with ops.graph_capture(g, side):
				y = x + n  # (an allocation inside the capture: the graph owns a private memory pool)
		torch.cuda.current_stream(dev).wait_stream(side)
		return g

	slot = tower_runtime._Slot()
	slot.graph = captured(1)
	spec = O.DecoderSpec(embed_dim=32, vocab_size=53, token_length=6, hidden_dim=64, feedfwd_dim=16, num_layers=2, num_heads=4)
	model, _ = make_decoder(spec, seed=3, device="cuda")
	model.eval()
	embed = torch.nn.functional.normalize(torch.randn(4, 32), dim=-1).cuda()
	with torch.no_grad():
		for _ in range(3):
			ref = model.generate(embed, False, True, 1.0, 0.0, None, None, False)  # (the third call replays the session's captured step graphs)
	sessions = list(model._decode_sessions.values())
	assert sessions and sessions[0].graphs
	n_graphs = len(sessions[0].graphs) + 1
	victim = dict(pinned=torch.empty(1 << 20, dtype=torch.float32).pin_memory(), dev=torch.empty(1 << 20, device=dev), ev=torch.cuda.Event(), slot=slot, sessions=sessions)
	model._decode_sessions.clear()
	del slot, sessions
	victim["ev"].record()
	torch.cuda.synchronize()
	opened, freed, errors, collected_while_open = threading.Event(), threading.Event(), [], []

	class Trash:
		def __init__(self):
			self.me = self

		def __del__(self):
			collected_while_open.append(ops.capture_open())

	def other():
		try:
			assert opened.wait(30)
			assert ops.capture_open()
			victim.clear()  # last references to a tower slot and a decode session (captured graphs), a pinned buffer, a device tensor, an event: dropped on THIS thread, now
			assert len(ops._parked) == n_graphs  # the graphs live on; everything else is gone
			torch.empty(1 << 18, dtype=torch.float32).pin_memory()  # a page-locked allocation: legal beside a thread-local capture
			Trash()
			freed.set()
			gc.collect()  # an explicit collection on this thread: parked by ops._gc_guard until the capture has ended
		except BaseException as e:  # noqa: BLE001 -- reported by the main thread
			errors.append(e)
		finally:
			freed.set()

	t = threading.Thread(target=other, name="test-freeing-thread")
	t.start()
	g = torch.cuda.CUDAGraph()
	side.wait_stream(torch.cuda.current_stream(dev))
	with torch.cuda.stream(side):
		with ops.graph_capture(g, side):
			x += 1
			opened.set()
			assert freed.wait(60)
			time.sleep(0.2)  # (the other thread is now inside gc.collect(), parked by the guard)
			x += 1
			assert not gc.isenabled() and collected_while_open == []
		assert gc.isenabled() and not ops.capture_open() and ops._parked == []  # the park was emptied by the thread that closed the capture
	t.join()
	torch.cuda.current_stream(dev).wait_stream(side)
	assert not errors, errors
	assert collected_while_open == [False]  # the other thread's collection ran, and only after the capture had ended
	g.replay()
	torch.cuda.synchronize()
	assert float(x[0]) == 2.0
	with torch.no_grad():  # the decoder works on (new session, new graphs) and gives what it gave before
		for _ in range(3):
			out = model.generate(embed, False, True, 1.0, 0.0, None, None, False)
	assert torch.equal(out[0], ref[0]) and torch.equal(out[5], ref[5])


# ---- the early-exit look without a copy or an event per step (round 6: ops.step_done, novic_step_done / novic_host_mapped_ptr) ----
def test_step_done_writes_its_word_into_mapped_host_memory():
	from novic_amd import ops, _lib
	host = torch.zeros(8, dtype=torch.int32).pin_memory()
	dev = ops.host_mapped_ptr(host)
	assert dev != 0
	active = torch.tensor([0, 3, 0, 1, -1, 0, 7, 0], dtype=torch.int32, device="cuda")
	for i in range(6):  # (words 6 and 7 stay untouched)
		ops.step_done(active[i:i + 1], dev + 4 * i)
	torch.cuda.synchronize()
	assert host.tolist() == [1, 2, 1, 2, 2, 1, 0, 0]
	host.zero_()
	st = torch.cuda.Stream()
	with torch.cuda.stream(st):  # visible to a polling host while the stream is still busy behind it (a system-scope release store, not an end-of-stream flush)
		ops.step_done(active[1:2], dev + 4)
		spin = torch.zeros(1 << 26, device="cuda")
		for _ in range(40):
			spin.add_(1.0)
	import time
	t0 = time.perf_counter()
	while int(host[1]) == 0 and time.perf_counter() - t0 < 10.0:
		pass
	seen_after, busy = time.perf_counter() - t0, not st.query()
	st.synchronize()
	assert int(host[1]) == 2 and seen_after < 10.0
	assert busy, "the stream had drained before the word was seen: the check above proved nothing -- lengthen the spin"
	with pytest.raises(_lib.NovicHipError):
		ops.host_mapped_ptr(torch.zeros(4, dtype=torch.int32))  # pageable memory
	with pytest.raises(_lib.NovicHipError):
		ops.host_mapped_ptr(active)


@pytest.mark.parametrize("beam", [False, True], ids=["greedy", "beam4"])
def test_the_host_stops_one_step_behind_the_last_active_one(beam):
	"""A model whose END logit wins every step from the second on (logits bias): every sequence is one token + END, the call returns two columns, and the host -- which
	enqueues step C before it looks at step C - 1's word -- enqueues exactly three of the ten steps: on the eager first call, on the capturing second one and on replays."""
	from novic_amd import embedding_decoder as ED
	spec = O.DecoderSpec(embed_dim=64, vocab_size=97, token_length=11, hidden_dim=64, feedfwd_dim=128, num_layers=2, num_heads=4)
	model, sd = make_decoder(spec, seed=21, logits_bias=True, device="cuda")
	with torch.no_grad():
		model.logits_linear.bias[0] = 1000.0
	model.eval()
	g = torch.Generator().manual_seed(8)
	calls = []
	orig = ED._DecodeSession.advance
	ED._DecodeSession.advance = lambda self, C: (calls.append(C), orig(self, C))[1]
	try:
		with torch.no_grad():
			for rep in range(4):
				embed = torch.nn.functional.normalize(torch.randn(5, spec.embed_dim, generator=g), dim=-1).cuda()
				calls.clear()
				out = model.generate_beam(embed, 4, 1.0, 0.0, None, False, 0.0, None, False) if beam else model.generate(embed, False, True, 1.0, 0.0, None, None, False)
				ids = out[0]
				assert ids.shape[-1] == 2, ids.shape
				first = ids[:, 0, 0] if beam else ids[:, 0]
				assert bool((first != 0).all()) and bool(((ids[:, 0, 1] if beam else ids[:, 1]) == 0).all())
				assert calls == [1, 2, 3], (rep, calls)
	finally:
		ED._DecodeSession.advance = orig
	# ... and a batch that never finishes runs all of them (END banned by the same bias)
	with torch.no_grad():
		model.logits_linear.bias[0] = -1000.0
		embed = torch.nn.functional.normalize(torch.randn(5, spec.embed_dim, generator=g), dim=-1).cuda()
		out = model.generate_beam(embed, 4, 1.0, 0.0, None, False, 0.0, None, False) if beam else model.generate(embed, False, True, 1.0, 0.0, None, None, False)
	assert out[0].shape[-1] == spec.token_length - 1  # (G generation steps: the guaranteed END of a full-length target is not generated)
