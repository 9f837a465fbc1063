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


@pytest.mark.parametrize("generic", [0, 1], ids=["wave_per_row", "workgroup_per_sample"])
@pytest.mark.parametrize("H,V,alpha,tau,ties", [(4, 53, 0.0, 1.0, False), (10, 307, 0.5, 2.0, False), (3, 61, 1.0, 0.7, True), (4, 6912, 0.0, 1.0, True), (5, 8192, 0.3, 1.0, True),
                                                (32, 40, 0.0, 1.0, False), (1, 9, 0.0, 1.0, True), (4, 8200, 0.0, 1.0, False)])
def test_beam_step_exact(H, V, alpha, tau, ties, generic):
	"""Both selection kernels (one wave per beam row, V <= 8192: the default; one workgroup per sample: any V) against the restatement."""
	from novic_amd import ops
	prev = ops.beam_step_policy(generic)
	try:
		_beam_step_exact(H, V, alpha, tau, ties)
	finally:
		ops.beam_step_policy(prev)


def _beam_step_exact(H, V, alpha, tau, ties):
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


# ---- guided steps: the kernels walk a token trie; the restatement below keeps the reference's per-beam "still consistent" noun masks ----

def _guide_targets(W, V, G, seed):
	g = torch.Generator().manual_seed(seed)
	first = torch.randint(1, V, (max(3, W // 3),), generator=g)
	rows = set()
	while len(rows) < W:
		ln = int(torch.randint(1, G + 1, (1,), generator=g))
		rows.add(tuple([int(first[int(torch.randint(0, len(first), (1,), generator=g))])] + [int(t) for t in torch.randint(1, min(V, 9), (ln - 1,), generator=g)]))
	out = torch.zeros(W, G + 1, dtype=torch.int64)
	for i, r in enumerate(sorted(rows)):
		out[i, :len(r)] = torch.tensor(r)
	return out


def ref_beam_step_guided(logits, C, G, ids, pad, score, lens, ok, guide, tau, alpha, renorm, guided, prior):
	"""ok: B x H x W bool.  prior: None | (per_token, scaler).  guided=False with a prior = vocabulary prior only (embedding_decoder.py:924-936)."""
	from oracle.decoder_oracle import allowed_token_mask
	B, H, V = logits.shape
	lg = (logits.float() / tau).clone()
	fin = pad[:, :, C - 1].bool()
	lg[:, :, 1:] = lg[:, :, 1:].masked_fill(fin.unsqueeze(2), NEG)
	allowed = allowed_token_mask(guide, ok, C - 1, V)
	gs = torch.zeros(B, H, V).masked_fill(~allowed, NEG)
	gs[:, :, 0] = gs[:, :, 0].masked_fill(fin, 0)
	if guided and renorm:
		lg = lg + gs
	cand = torch.log_softmax(lg, dim=2)
	if prior is not None:
		toks = guide[:, C - 1]
		pr = torch.zeros(B, H, V)
		for b in range(B):
			for h in range(H):
				t = toks[ok[b, h]]
				if prior[0]:
					pr[b, h, t] = 1.0
				else:
					pr[b, h].index_add_(0, t, torch.ones(t.shape[0]))
		adj = (pr / pr.sum(dim=2, keepdim=True)).log().nan_to_num(nan=float("inf"), neginf=float("inf"), posinf=float("inf"))
		adj[:, :, 0] = adj[:, :, 0].masked_fill(fin, 0)
		cand = cand - prior[1] * adj
	cand = cand + score.unsqueeze(2)
	if C == 1:
		cand[:, 0, 0] = NEG
	if guided and not renorm:
		cand = cand + gs
	cand = torch.where(torch.isnan(cand), torch.full_like(cand, NEG), cand)  # dead beams: -inf - inf etc.
	rank = cand * lens.clamp(min=1).pow(-alpha).unsqueeze(2) if alpha != 0 else cand
	flat = rank.view(B, -1)
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
	n_len = lens.gather(1, src) + ((~nxt).float() if C < G else 0)
	n_ok = ok[bidx, src] & (guide[:, C - 1].view(1, 1, -1) == tok.unsqueeze(2))
	return n_ids, n_pad, n_score, flat.gather(1, order), n_len, n_ok


@pytest.mark.parametrize("H,V,W,alpha,tau,renorm,guided,prior", [
	(4, 53, 20, 0.0, 1.0, False, True, None),
	(4, 53, 20, 0.5, 2.0, True, True, None),
	(6, 61, 3, 0.0, 1.0, False, True, None),          # fewer continuations than beams
	(4, 307, 40, 0.0, 1.0, False, True, (False, 1.0)),
	(5, 307, 40, 0.7, 1.0, True, True, (True, 0.5)),
	(4, 53, 20, 0.0, 1.0, False, False, (False, 1.0)),  # prior only
	(10, 6912, 200, 0.0, 1.0, False, True, None),
])
def test_beam_step_guided_exact(H, V, W, alpha, tau, renorm, guided, prior):
	from novic_amd import ops
	from novic_amd.guide_trie import TokenTrie
	B, G = 5, 6
	g = torch.Generator().manual_seed(H * 1000 + V + W)
	Vp = (V + 7) // 8 * 8
	guide = _guide_targets(W, V, G, seed=V + W)
	trie = TokenTrie(guide, torch.device("cuda"))
	logprior = None if prior is None else (trie.logprior_token if prior[0] else trie.logprior_target)
	ids = torch.zeros(B, H, G, dtype=torch.int64)
	pad = torch.ones(B, H, G, dtype=torch.uint8)
	pad[:, 0, 0] = 0
	score = torch.full((B, H), NEG)
	score[:, 0] = 0
	lens = torch.zeros(B, H)
	lens[:, 0] = 1
	ok = torch.zeros(B, H, W, dtype=torch.bool)
	ok[:, 0] = True
	node = torch.full((B, H), -2, dtype=torch.int32)
	node[:, 0] = 0
	node = node.cuda()
	for C in range(1, G + 1):
		lg = torch.randn(B, H, Vp, generator=g)
		lg[:, :, 0] += 2.5  # END competitive: beams finish at different steps and finished beams compete with live ones
		lg16 = lg.to(torch.bfloat16)
		r = ref_beam_step_guided(lg16[:, :, :V], C, G, ids, pad, score, lens, ok, guide, tau, alpha, renorm, guided, prior)
		d = lambda t: t.cuda().contiguous()
		o_ids, o_pad = torch.empty_like(ids).cuda(), torch.empty_like(pad).cuda()
		o_score, o_rank, o_len = torch.empty(B, H).cuda(), torch.empty(B, H).cuda(), torch.empty(B, H).cuda()
		active, src, o_node = torch.zeros(G, dtype=torch.int32).cuda(), torch.zeros(B, H, dtype=torch.int32).cuda(), torch.empty_like(node)
		ops.beam_step_guided(d(lg16.view(B * H, Vp)), Vp, V, B, H, G, C, d(ids), o_ids, d(pad), o_pad, d(score), o_score, o_rank, d(lens), o_len, active, src, node, o_node, trie,
		                     logprior, 0.0 if prior is None else prior[1], renorm and guided, tau, alpha)
		torch.cuda.synchronize()
		fin = torch.isfinite(r[2])
		assert torch.equal(torch.isfinite(o_score.cpu()), fin), C
		assert torch.equal(o_ids.cpu()[:, :, :C][fin], r[0][:, :, :C][fin]), C
		assert torch.equal(o_pad.cpu()[:, :, :min(C + 1, G)][fin], r[1][:, :, :min(C + 1, G)][fin]), C
		torch.testing.assert_close(o_score.cpu()[fin], r[2][fin], atol=3e-5, rtol=1e-5)
		torch.testing.assert_close(o_rank.cpu()[fin], r[3][fin], atol=3e-5, rtol=1e-5)
		assert torch.equal(o_len.cpu()[fin], r[4][fin])
		# dead picks stay dead: padded everywhere from the current column on
		assert bool((o_pad.cpu()[:, :, C - 1:][~fin] == 1).all()) and bool((o_node.cpu()[~fin] == -2).all())
		ids, pad, score, lens, ok, node = r[0], r[1], r[2], r[4], r[5], o_node
		ids[~fin] = 0
		pad[~fin] = 1


@pytest.mark.parametrize("V,W,tau,renorm", [(53, 20, 1.0, False), (307, 60, 2.0, True), (6912, 300, 1.0, True)])
def test_greedy_step_guided_exact(V, W, tau, renorm):
	from novic_amd import ops
	from novic_amd.guide_trie import TokenTrie
	from oracle.decoder_oracle import allowed_token_mask
	B, G = 37, 5
	g = torch.Generator().manual_seed(V + W)
	Vp = (V + 7) // 8 * 8
	guide = _guide_targets(W, V, G, seed=V)
	trie = TokenTrie(guide, torch.device("cuda"))
	ids = torch.zeros(B, G, dtype=torch.int32).cuda()
	pad = torch.zeros(B, G, dtype=torch.uint8).cuda()
	alive = torch.ones(B).cuda()
	score, nll, count = torch.zeros(B).cuda(), torch.zeros(B).cuda(), torch.zeros(B).cuda()
	active, node = torch.zeros(G, dtype=torch.int32).cuda(), torch.zeros(B, dtype=torch.int32).cuda()
	r_alive, ok = torch.ones(B, dtype=torch.bool), torch.ones(B, W, dtype=torch.bool)
	r_score, r_nll, r_cnt = torch.zeros(B), torch.zeros(B), torch.zeros(B)
	for C in range(1, G + 1):
		lg16 = torch.randn(B, Vp, generator=g).to(torch.bfloat16)
		ops.greedy_step_guided(lg16.cuda(), Vp, V, B, G, C, ids, pad, alive, score, nll, count, active, None, node, trie, renorm, tau, 0.0)
		x = lg16[:, :V].float()
		gs = torch.zeros(B, V).masked_fill(~allowed_token_mask(guide, ok, C - 1, V), NEG)
		tok = (x + gs).argmax(dim=1)
		ok = ok & (guide[:, C - 1].unsqueeze(0) == tok.unsqueeze(1))
		lp_t = torch.log_softmax(x / tau + (gs if renorm else 0), dim=1).gather(1, tok.unsqueeze(1)).squeeze(1)
		lp = torch.log_softmax(x, dim=1).gather(1, tok.unsqueeze(1)).squeeze(1)
		r_pad = ~r_alive
		r_score += torch.where(r_alive, lp_t, torch.zeros(B))
		r_nll += torch.where(r_alive, -lp, torch.zeros(B))
		r_cnt += r_alive.float()
		torch.cuda.synchronize()
		assert torch.equal(ids[:, C - 1].cpu().long()[r_alive], tok[r_alive]) and bool((ids[:, C - 1].cpu()[~r_alive] == 0).all())
		assert torch.equal(pad[:, C - 1].cpu().bool(), r_pad)
		r_alive = r_alive & (tok != 0)
		assert torch.equal(alive.cpu().bool(), r_alive) and int(active[C - 1]) == int(r_alive.sum())
	torch.testing.assert_close(score.cpu(), r_score, atol=5e-5, rtol=1e-5)
	torch.testing.assert_close(nll.cpu(), r_nll, atol=5e-5, rtol=1e-5)
	assert torch.equal(count.cpu(), r_cnt)


@pytest.mark.parametrize("B,beams,H,D,P,G,pos,use_origin", [(37, 1, 8, 64, 4, 11, 0, False), (37, 1, 8, 64, 4, 11, 10, False), (9, 4, 8, 64, 4, 11, 5, False),
                                                           (5, 3, 4, 64, 4, 12, 11, False), (6, 2, 4, 64, 4, 20, 15, False), (7, 2, 4, 32, 4, 11, 6, False),
                                                           (3, 1, 2, 64, 1, 11, 3, False), (9, 4, 8, 64, 4, 11, 7, True), (6, 2, 4, 64, 4, 20, 15, True), (7, 3, 4, 32, 4, 11, 6, True)])
def test_decode_attention_step(B, beams, H, D, P, G, pos, use_origin):
	"""novic_decode_attn (one new position per sequence against the prefix K/V of its sample and its label cache): the <= 16-key head_dim-64
	kernel and the general one, against a torch fp32 soft-max attention over the same bf16 rows; the new position's k, v must land in the
	sequence's OWN cache row and nothing else in the cache may change.  use_origin: label position g of sequence a is read from cache row
	origin[a][g] (beam search without moving K/V).  Tolerance: bf16 output (2^-8 relative) + fp32 summation order."""
	import math
	from novic_amd import ops
	A, E = B * beams, H * D
	g = torch.Generator().manual_seed(B * 100 + pos)
	qkv_new = torch.randn(A, 3 * E, generator=g).to(torch.bfloat16)
	prefix = torch.randn(B * P, 3 * E, generator=g).to(torch.bfloat16)
	ck = torch.randn(A, G, E, generator=g).to(torch.bfloat16)
	cv = torch.randn(A, G, E, generator=g).to(torch.bfloat16)
	origin = None
	if use_origin:  # any row of the same sample
		origin = (torch.arange(A).view(A, 1) // beams) * beams + torch.randint(0, beams, (A, G), generator=g)
	dk, dv = ck.clone().cuda(), cv.clone().cuda()
	o = torch.full((A, E), float("nan"), dtype=torch.bfloat16, device="cuda")
	ops.decode_attn(qkv_new.cuda(), prefix.cuda(), dk, dv, o, A, H, D, P, G, pos, beams, origin=None if origin is None else origin.to(torch.int32).cuda())
	want_k, want_v = ck.clone(), cv.clone()
	want_k[:, pos], want_v[:, pos] = qkv_new[:, E:2 * E], qkv_new[:, 2 * E:]
	assert torch.equal(dk.cpu(), want_k) and torch.equal(dv.cpu(), want_v)
	pk = prefix[:, E:2 * E].view(B, 1, P, E).expand(B, beams, P, E).reshape(A, P, E)
	pv = prefix[:, 2 * E:].view(B, 1, P, E).expand(B, beams, P, E).reshape(A, P, E)
	rows = torch.arange(A).view(A, 1).expand(A, G) if origin is None else origin
	lk, lv = ck[rows, torch.arange(G).view(1, G)].clone(), cv[rows, torch.arange(G).view(1, G)].clone()  # what each sequence sees as its label cache
	lk[:, pos], lv[:, pos] = qkv_new[:, E:2 * E], qkv_new[:, 2 * E:]
	K = torch.cat([pk, lk[:, :pos + 1]], dim=1).float().view(A, P + pos + 1, H, D).transpose(1, 2)  # A x H x keys x D
	Vv = torch.cat([pv, lv[:, :pos + 1]], dim=1).float().view(A, P + pos + 1, H, D).transpose(1, 2)
	q = qkv_new[:, :E].float().view(A, H, 1, D)
	ref = (torch.softmax(q @ K.transpose(-1, -2) / math.sqrt(D), dim=-1) @ Vv).reshape(A, E)
	assert float((o.cpu().float() - ref).abs().max()) <= 2 ** -7 * float(ref.abs().max())


def test_kv_origin_update():
	"""origin_out[a][g] = origin_in[sa][g] for g < npos - 1, sa for g = npos - 1 (sa = the old sequence a continues); columns >= npos untouched."""
	from novic_amd import ops
	B, beams, G, npos = 5, 4, 11, 6
	A = B * beams
	g = torch.Generator().manual_seed(3)
	src = torch.randint(0, beams, (A,), generator=g).to(torch.int32)
	oin = torch.randint(0, A, (A, G), generator=g).to(torch.int32)
	out = torch.full((A, G), -7, dtype=torch.int32, device="cuda")
	ops.kv_origin_update(src.cuda(), oin.cuda(), out, A, beams, G, npos)
	sa = (torch.arange(A) // beams) * beams + src.long()
	want = torch.full((A, G), -7, dtype=torch.int32)
	want[:, :npos - 1] = oin[sa, :npos - 1]
	want[:, npos - 1] = sa.to(torch.int32)
	assert torch.equal(out.cpu(), want)


@pytest.mark.parametrize("generic", [0, 1], ids=["wave_per_row", "workgroup_per_sample"])
def test_step_kernels_survive_rows_without_a_candidate(generic):
	"""A sample whose logits are all NaN offers no arg-max / fewer than H candidates.  The fused next-input path of the step kernels must stay inside W_tok and
	the beam histories exactly as the separate novic_decode_embed launch (which clamps the token) does: no memory fault, finite rows for the healthy samples,
	and for the NaN sample the embedding of an in-range token."""
	from novic_amd import ops
	prev = ops.beam_step_policy(generic)
	try:
		B, H, G, V, E = 3, 4, 5, 53, 64
		Vp = (V + 7) // 8 * 8
		g = torch.Generator().manual_seed(7)
		wtok, pos_row = torch.randn(V, E, generator=g).cuda(), torch.randn(E, generator=g).cuda()
		lg = torch.randn(B, Vp, generator=g)
		lg[1] = float("nan")
		# greedy
		ids = torch.zeros(B, G, dtype=torch.int64).cuda()
		pad = torch.zeros(B, G, dtype=torch.uint8).cuda()
		alive, score, nll, count = torch.ones(B).cuda(), torch.zeros(B).cuda(), torch.zeros(B).cuda(), torch.zeros(B).cuda()
		active = torch.zeros(G, dtype=torch.int32).cuda()
		x_next = torch.full((B, E), 7.0).cuda()
		ops.greedy_step(lg.to(torch.bfloat16).cuda(), Vp, V, B, G, 1, ids, pad, alive, score, nll, count, active, None, 1.0, 0.0, x_next=x_next, wtok=wtok, pos_row=pos_row)
		torch.cuda.synchronize()
		tok = ids[:, 0].cpu()
		assert int(tok.min()) >= 0 and int(tok.max()) < V and int(tok[1]) == 0 and float(alive[1]) == 0.0
		x_ref = torch.empty(B, E).cuda()
		ops.decode_embed(ids, G, 0, wtok, pos_row, x_ref, B, E, V)
		torch.cuda.synchronize()
		assert torch.equal(x_next, x_ref)
		# beam: sample 1 offers no candidate at all
		lgb = torch.randn(B, H, Vp, generator=g)
		lgb[1] = float("nan")
		bids, bpad = torch.zeros(B, H, G, dtype=torch.int64), torch.ones(B, H, G, dtype=torch.uint8)
		bpad[:, 0, 0] = 0
		bscore, lens = torch.full((B, H), NEG), torch.zeros(B, H)
		bscore[:, 0] = 0
		lens[:, 0] = 1
		d = lambda t: t.cuda().contiguous()
		o_ids, o_pad = torch.empty_like(bids).cuda(), torch.empty_like(bpad).cuda()
		o_score, o_rank, o_len = torch.empty(B, H).cuda(), torch.empty(B, H).cuda(), torch.empty(B, H).cuda()
		src = torch.full((B, H), -1, dtype=torch.int32).cuda()
		xb = torch.full((B * H, E), 7.0).cuda()
		origin_in = torch.arange(B * H, dtype=torch.int32).repeat_interleave(G).view(B * H, G).cuda().contiguous()
		origin_out = torch.full((B * H, G), -1, dtype=torch.int32).cuda()
		ops.beam_step(d(lgb.to(torch.bfloat16).view(B * H, Vp)), Vp, V, B, H, G, 1, d(bids), o_ids, d(bpad), o_pad, d(bscore), o_score, o_rank, d(lens), o_len, active, 1.0, 0.0,
		              src_out=src, x_next=xb, wtok=wtok, pos_row=pos_row, origin_in=origin_in, origin_out=origin_out, npos=1)
		torch.cuda.synchronize()
		assert int(o_ids.min()) >= 0 and int(o_ids.max()) < V
		assert int(src.min()) >= 0 and int(src.max()) < H
		assert int(origin_out[:, 0].min()) >= 0 and int(origin_out[:, 0].max()) < B * H
		x_ref = torch.empty(B * H, E).cuda()
		ops.decode_embed(o_ids.view(B * H, G), G, 0, wtok, pos_row, x_ref, B * H, E, V)
		torch.cuda.synchronize()
		assert torch.equal(xb, x_ref) and torch.isfinite(xb).all()
	finally:
		ops.beam_step_policy(prev)
