"""Beam-10 -- the released default decode mode, `beam_k10_vnone_gp_t1_a0` (reference infer.py:55) -- gated on EVERY sample and EVERY step.

Trained reference decoders cannot separate ten beams by a bf16-proof margin (tests/golden/make_golden_r2.py tried; the margin-gated tests of
test_gpu_generate_trained.py therefore pass on an almost empty set for H = 10).  This file closes the gap without margins, by splitting "the GPU search equals
the reference search" into two statements that together imply it up to the stated logit tolerance, and checking both for all samples, beams and steps:

  (1) BOOKKEEPING IS EXACT.  The oracle's restatement of one reference beam step (oracle.decoder_oracle.beam_step = embedding_decoder.py:911-978, pinned through
      O.generate_beam by the reference-generated fixtures) is applied to the GPU's OWN logits of step t and the GPU's state after step t - 1; the GPU's state after
      step t must be identical: finite pattern, ids, padding, source beam (= the K/V origin), lengths; scores to fp32 rounding.  Ties (bf16 logits do tie) are
      ranked by ascending flat index h * V + v, the tie-break the product defines.
  (2) THE LOGITS ARE THE MODEL'S.  Every live beam's step-t logits, produced through the prefix cache + per-beam K/V cache + origin table, equal the oracle's
      UNCACHED teacher-forced forward on that beam's history (bf16 emulation) within the forward tolerance -- a wrong cache row for any of the ten beams shows here.
"""
import math

import pytest
import torch

from conftest import load_golden
from helpers import make_decoder
from oracle import decoder_oracle as O

pytestmark = pytest.mark.gpu

TR = load_golden("decoder_trained.pt")


def _trained(name):
	m = TR["models"][name]
	spec = O.DecoderSpec(**m["spec"])
	sd = O.init_state_dict(spec, seed=0)
	sd.update({k: v.float() for k, v in m["weights"].items()})
	return spec, sd


def _guide(spec, W, seed, max_len=4):
	"""W distinct guide nouns of 1..max_len tokens sharing first tokens (so the ten beams branch inside the trie), END-terminated, zero-padded to token_length."""
	g = torch.Generator().manual_seed(seed)
	first = torch.randint(1, spec.vocab_size, (max(4, W // 6),), generator=g)
	rows = set()
	while len(rows) < W:
		ln = int(torch.randint(1, max_len + 1, (1,), generator=g))
		rows.add(tuple([int(first[int(torch.randint(0, len(first), (1,), generator=g))])] + [int(t) for t in torch.randint(1, min(spec.vocab_size, 12), (ln - 1,), generator=g)]))
	out = torch.zeros(W, spec.token_length, dtype=torch.int64)
	for i, r in enumerate(sorted(rows)):
		out[i, :len(r)] = torch.tensor(r)
	return out


def _case(name):
	if name == "trained_wide_gp":       # the fixture's own guided beam-10 case: trained reference decoder (d = 512, V = 131), 12 prototype embeddings
		spec, sd = _trained("wide")
		c = next(c for c in TR["cases"] if c["name"] == "wide_beam10_gp")
		return spec, sd, c["embed"], dict(topk=10, temperature=1.0, length_alpha=0.0, guide=c["guide_targets"], renorm=False)
	if name == "trained_wide_unguided":  # same decoder, no guide: ten free beams over V = 131, 64 embeddings (the 12 prototypes + 52 random directions)
		spec, sd = _trained("wide")
		c = next(c for c in TR["cases"] if c["name"] == "wide_beam10_gp")
		g = torch.Generator().manual_seed(5)
		extra = torch.nn.functional.normalize(torch.randn(52, spec.embed_dim, generator=g), dim=-1)
		return spec, sd, torch.cat((c["embed"], extra)), dict(topk=10, temperature=1.0, length_alpha=0.0, guide=None, renorm=False)
	if name == "default6_gp_a05":        # the released depth (6 layers), V = 307, random init, 300 guide nouns, length normalisation and a temperature
		spec = O.DecoderSpec(embed_dim=64, vocab_size=307, token_length=8)
		sd = O.init_state_dict(spec, seed=11)
		g = torch.Generator().manual_seed(6)
		embed = torch.nn.functional.normalize(torch.randn(32, spec.embed_dim, generator=g), dim=-1)
		return spec, sd, embed, dict(topk=10, temperature=1.5, length_alpha=0.5, guide=_guide(spec, 300, 7), renorm=False)
	if name == "default6_gr":            # guided with renormalisation over the allowed tokens (gr)
		spec = O.DecoderSpec(embed_dim=64, vocab_size=307, token_length=8)
		sd = O.init_state_dict(spec, seed=12)
		g = torch.Generator().manual_seed(8)
		embed = torch.nn.functional.normalize(torch.randn(24, spec.embed_dim, generator=g), dim=-1)
		return spec, sd, embed, dict(topk=10, temperature=1.0, length_alpha=0.0, guide=_guide(spec, 200, 9), renorm=True)
	raise KeyError(name)


@pytest.mark.parametrize("graphs", [False, True], ids=["eager", "graph_replay"])
@pytest.mark.parametrize("name", ["trained_wide_gp", "trained_wide_unguided", "default6_gp_a05", "default6_gr"])
def test_beam10_every_sample_every_step(name, graphs):
	spec, sd, embed, a = _case(name)
	model, _ = make_decoder(spec, sd=sd, device="cuda")
	model.eval()
	B, H, V, G = embed.shape[0], a["topk"], spec.vocab_size, spec.token_length - 1
	gt = a["guide"]
	gt_dev = None if gt is None else gt.cuda()
	run = lambda: model.generate_beam(embed.cuda(), H, a["temperature"], a["length_alpha"], None, False, 0.0, gt_dev, a["renorm"])
	with torch.no_grad():
		if graphs:  # call 1 eager, call 2 captures the per-step graphs, the traced call replays them
			run(); run()
		model.decode_trace, model.decode_trace_logits = [], []
		try:
			ids_out, pad_out, score_out = (t.cpu() for t in run())
			steps = [{k: v.cpu() for k, v in d.items()} for d in model.decode_trace_logits]
		finally:
			model.decode_trace = model.decode_trace_logits = None
	assert len(steps) >= 2
	ids, pad, score, seq_len, g_ok, _ = O.beam_start(B, H, G, n_guide=None if gt is None else gt.shape[0])
	checked = live_rows = 0
	for t, st in enumerate(steps):
		C = t + 1
		lg = st["logits"][:, :, :V].float()
		lg_pad = pad[:, :, C - 1:C].clone()
		# (2) the logits of every live, unfinished beam against the uncached forward on its history
		live = torch.isfinite(score) & ~pad[:, :, C - 1]
		ref = O.forward(sd, spec, embed, ids[:, :, :C], pad[:, :, :C], None, False, False, True, bf16=True)[0].squeeze(2)  # B x H x V
		if bool(live.any()):  # (a surplus step behind the early exit has only finished beams)
			scale = max(1.0, float(ref[live].abs().max()))
			err = float((lg - ref)[live].abs().max())
			assert err <= 1.5e-2 * scale, (name, C, err, scale)
		live_rows += int(live.sum())
		# (1) one reference step on the GPU's logits from the GPU's previous state
		if gt is not None:  # the guide nouns still consistent with each candidate's history (what the reference carries as a B x H x W mask, :969-971)
			g_ok = torch.isfinite(score).unsqueeze(2) & (ids[:, :, None, :C - 1] == gt[None, None, :, :C - 1]).all(dim=3)
		n_ids, n_pad = ids.clone(), pad.clone()
		n_score, n_normed, n_len, _, _, src, done = O.beam_step(lg, lg_pad, C, G, n_ids, n_pad, score, seq_len, a["temperature"], a["length_alpha"], g_ok, gt, a["renorm"],
		                                                        stable_ties=True)
		fin = torch.isfinite(n_score)
		assert torch.equal(torch.isfinite(st["score"]), fin), (name, C)
		assert torch.equal(st["ids"][:, :, :C][fin], n_ids[:, :, :C][fin]), (name, C)
		upto = min(C + 1, G)
		assert torch.equal(st["pad"].bool()[:, :, :upto][fin], n_pad[:, :, :upto][fin]), (name, C)
		assert torch.equal(st["src"].long()[fin], src[fin]), (name, C)
		torch.testing.assert_close(st["score"][fin], n_score[fin], atol=5e-5, rtol=1e-5)
		torch.testing.assert_close(st["normed"][fin], n_normed[fin], atol=5e-5, rtol=1e-5)
		if C < G and not done:
			assert torch.equal(st["lens"][fin], n_len[fin]), (name, C)
		checked += int(fin.sum())
		# continue from the GPU's state (dead beams: keep the oracle's, their contents are undefined on both sides)
		ids, pad = torch.where(fin.unsqueeze(2), st["ids"], n_ids), torch.where(fin.unsqueeze(2), st["pad"].bool(), n_pad)
		score, seq_len = st["score"], torch.where(fin, st["lens"], n_len)
	T = len(steps)
	# every sample took part at every step, with (after the first steps) all ten beams alive
	assert checked >= B * (T - 1) * H * 0.6, (checked, B, T)
	assert live_rows >= B * T, live_rows
	# what generate_beam returned is the last state (padded ids zeroed, reference :980), trimmed to the early-exit length
	Tc = ids_out.shape[2]
	last = steps[-1]
	fin = torch.isfinite(last["score"])
	want_ids = last["ids"][:, :, :Tc].masked_fill(last["pad"].bool()[:, :, :Tc], 0)
	assert torch.equal(ids_out[fin], want_ids[fin]) and torch.equal(pad_out[fin], last["pad"].bool()[:, :, :Tc][fin])
	torch.testing.assert_close(score_out[fin], last["normed"][fin] if a["length_alpha"] != 0 else last["score"][fin], atol=1e-6, rtol=0)
