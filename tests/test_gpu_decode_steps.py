"""Bit-exactness of the decode step kernels given identical logits: a torch restatement of one reference step
(embedding_decoder.py:905-978 for beams, :798-816 for greedy) applied to the SAME bf16 logits must give identical ids / padding and
scores to 1e-5, including the tie-break (lowest flat index) on deliberately tied logits."""
import pytest
import torch

pytestmark = pytest.mark.gpu
NEG = float("-inf")


def ref_beam_step(logits, C, G, ids, pad, score, lens, tau, alpha):
	B, H, V = logits.shape
	lg = (logits.float() / tau).clone()
	fin = pad[:, :, C - 1].bool()
	lg[:, :, 1:] = lg[:, :, 1:].masked_fill(fin.unsqueeze(2), NEG)
	cand = torch.log_softmax(lg, dim=2) + score.unsqueeze(2)
	if C == 1:
		cand[:, 0, 0] = NEG
	rank = cand * lens.clamp(min=1).pow(-alpha).unsqueeze(2) if alpha != 0 else cand
	flat = rank.view(B, -1)
	# stable descending sort = ties resolved towards the lowest flat index
	order = torch.sort(flat, dim=1, descending=True, stable=True).indices[:, :H]
	src, tok = order // V, order % V
	bidx = torch.arange(B).unsqueeze(1)
	n_ids, n_pad = torch.zeros_like(ids), torch.ones_like(pad)
	n_ids[:, :, :C - 1] = ids[bidx, src, :C - 1]
	n_ids[:, :, C - 1] = tok
	n_pad[:, :, :C] = pad[bidx, src, :C]
	nxt = (tok == 0) | pad[bidx, src, C - 1].bool()
	if C < G:
		n_pad[:, :, C] = nxt.to(pad.dtype)
	n_score = cand.view(B, -1).gather(1, order)
	n_rank = flat.gather(1, order)
	n_len = lens.gather(1, src) + ((~nxt).float() if C < G else 0)
	return n_ids, n_pad, n_score, n_rank, n_len, int((~nxt).sum())


@pytest.mark.parametrize("H,V,alpha,tau,ties", [(4, 53, 0.0, 1.0, False), (10, 307, 0.5, 2.0, False), (3, 61, 1.0, 0.7, True), (4, 6912, 0.0, 1.0, True)])
def test_beam_step_exact(H, V, alpha, tau, ties):
	from novic_amd import ops
	B, G = 5, 6
	g = torch.Generator().manual_seed(H * 1000 + V)
	Vp = (V + 7) // 8 * 8
	ids = torch.zeros(B, H, G, dtype=torch.int64)
	pad = torch.ones(B, H, G, dtype=torch.uint8)
	pad[:, 0, 0] = 0
	score = torch.full((B, H), NEG)
	score[:, 0] = 0
	lens = torch.zeros(B, H)
	lens[:, 0] = 1
	for C in range(1, G + 1):
		lg = torch.randn(B, H, Vp, generator=g)
		if ties:
			lg = (lg * 2).round() / 2  # many exact ties
		lg[:, :, 0] += 1.0  # make END competitive so beams finish at different steps
		lg16 = lg.to(torch.bfloat16)
		r = ref_beam_step(lg16[:, :, :V], C, G, ids, pad, score, lens, tau, alpha)
		d = lambda t: t.cuda().contiguous()
		o_ids, o_pad = torch.empty_like(ids).cuda(), torch.empty_like(pad).cuda()
		o_score, o_rank, o_len = torch.empty(B, H).cuda(), torch.empty(B, H).cuda(), torch.empty(B, H).cuda()
		active = torch.zeros(G, dtype=torch.int32).cuda()
		ops.beam_step(d(lg16.view(B * H, Vp)), Vp, V, B, H, G, C, d(ids), o_ids, d(pad), o_pad, d(score), o_score, o_rank, d(lens), o_len, active, tau, alpha)
		torch.cuda.synchronize()
		# compare only what the reference defines: columns < C (+ padding column C), finite-score bookkeeping
		assert torch.equal(o_ids.cpu()[:, :, :C], r[0][:, :, :C]), C
		assert torch.equal(o_pad.cpu()[:, :, :min(C + 1, G)], r[1][:, :, :min(C + 1, G)]), C
		fin = torch.isfinite(r[2])
		assert torch.equal(torch.isfinite(o_score.cpu()), fin)
		torch.testing.assert_close(o_score.cpu()[fin], r[2][fin], atol=2e-5, rtol=1e-5)
		torch.testing.assert_close(o_rank.cpu()[fin], r[3][fin], atol=2e-5, rtol=1e-5)
		assert torch.equal(o_len.cpu(), r[4]) and int(active[C - 1]) == r[5]
		ids, pad, score, lens = r[0], r[1], r[2], r[4]


@pytest.mark.parametrize("V,tau,ties", [(53, 1.0, False), (6912, 2.0, True), (307, 0.5, False)])
def test_greedy_step_exact(V, tau, ties):
	from novic_amd import ops
	B, G = 37, 5
	g = torch.Generator().manual_seed(V)
	Vp = (V + 7) // 8 * 8
	ids = torch.zeros(B, G, dtype=torch.int32).cuda()
	pad = torch.zeros(B, G, dtype=torch.uint8).cuda()
	alive = torch.ones(B).cuda()
	score, nll, count = torch.zeros(B).cuda(), torch.zeros(B).cuda(), torch.zeros(B).cuda()
	active = torch.zeros(G, dtype=torch.int32).cuda()
	r_alive = torch.ones(B, dtype=torch.bool)
	r_score, r_nll, r_cnt = torch.zeros(B), torch.zeros(B), torch.zeros(B)
	for C in range(1, G + 1):
		lg = torch.randn(B, Vp, generator=g)
		if ties:
			lg = (lg * 2).round() / 2
		lg[:, 0] += 2.0
		lg16 = lg.to(torch.bfloat16)
		ops.greedy_step(lg16.cuda(), Vp, V, B, G, C, ids, pad, alive, score, nll, count, active, None, tau, 0.0)
		x = lg16[:, :V].float()
		xa = x.clone()
		if C == 1:
			xa[:, 0] = NEG
		tok = xa.argmax(dim=1)  # torch arg-max returns the first maximal index
		r_pad = ~r_alive
		lp_t = torch.log_softmax(x / tau, dim=1).gather(1, tok.unsqueeze(1)).squeeze(1)
		lp = torch.log_softmax(x, dim=1).gather(1, tok.unsqueeze(1)).squeeze(1)
		r_score += torch.where(r_alive, lp_t, torch.zeros(B))
		r_nll += torch.where(r_alive, -lp, torch.zeros(B))
		r_cnt += r_alive.float()
		torch.cuda.synchronize()
		assert torch.equal(ids[:, C - 1].cpu().long(), tok) and torch.equal(pad[:, C - 1].cpu().bool(), r_pad)
		r_alive = r_alive & (tok != 0)
		assert torch.equal(alive.cpu().bool(), r_alive) and int(active[C - 1]) == int(r_alive.sum())
	torch.testing.assert_close(score.cpu(), r_score, atol=5e-5, rtol=1e-5)
	torch.testing.assert_close(nll.cpu(), r_nll, atol=5e-5, rtol=1e-5)
	assert torch.equal(count.cpu(), r_cnt)
