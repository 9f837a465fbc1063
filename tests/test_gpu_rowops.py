"""Direct parity of the row kernels (novic_layernorm_fwd / novic_layernorm_bwd) against torch fp32 LayerNorm (bias-free, the decoder's layer_bias=False
norms of embedding_decoder.py:1289-1300) and its autograd, including the row-selection window the final norm uses (only the label positions of each
sequence reach the logits, embedding_decoder.py:690) and row widths that are not multiples of the 256-element lane chunks.
Tolerances: the forward output is bf16 (rel 2^-8); the backward works in fp32 on a bf16 upstream gradient: |err| <= 2e-5 * scale."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ln(x, gamma, eps=1e-5):
	return torch.nn.functional.layer_norm(x, (x.shape[-1],), gamma, None, eps)


@pytest.mark.parametrize("rows,E", [(1000, 512), (77, 320), (4099, 768), (5, 2048), (13, 4)])
def test_layernorm_forward(rows, E):
	from novic_amd import ops
	g = torch.Generator().manual_seed(rows + E)
	x = torch.randn(rows, E, generator=g) * 3 + 0.5
	gamma = torch.randn(E, generator=g)
	y = torch.empty(rows, E, dtype=torch.bfloat16, device="cuda")
	ops.layernorm_fwd(x.cuda(), gamma.cuda(), y, rows, E)
	want = _ln(x, gamma)
	err = (y.float().cpu() - want).abs()
	assert float((err - want.abs() * 2 ** -8).max()) <= 1e-6


@pytest.mark.parametrize("A,seq_in,seq_out,seq_off,E,has_dx,g_out,drop", [
	(700, 1, 1, 0, 512, True, True, 0.0), (700, 1, 1, 0, 512, False, True, 0.0), (123, 10, 7, 3, 512, False, True, 0.0), (123, 10, 7, 3, 512, True, False, 0.0),
	(31, 5, 2, 1, 320, True, True, 0.0), (1, 1, 1, 0, 4, True, True, 0.0), (9000, 1, 1, 0, 768, True, True, 0.0), (6000, 10, 7, 3, 512, True, True, 0.1),
	(3, 1, 1, 0, 2048, True, True, 0.0)])
def test_layernorm_backward(A, seq_in, seq_out, seq_off, E, has_dx, g_out, drop):
	from novic_amd import ops
	rows = A * seq_in
	g = torch.Generator().manual_seed(A * 7 + E + seq_in)
	x = (torch.randn(rows, E, generator=g) * 2 + 0.25).requires_grad_(True)
	gamma = torch.randn(E, generator=g).requires_grad_(True)
	dy = torch.randn(A * seq_out, E, generator=g).bfloat16()
	dx_in = torch.randn(rows, E, generator=g) if has_dx else None
	y = _ln(x, gamma).view(A, seq_in, E)[:, seq_off:seq_off + seq_out].reshape(A * seq_out, E)
	y.backward(dy.float())
	want_dx = x.grad + (dx_in if has_dx else 0)
	dgamma0 = torch.randn(E, generator=g)

	dxo = torch.full((rows, E), float("nan"), device="cuda")
	gb = torch.empty(rows, E, dtype=torch.bfloat16, device="cuda") if g_out else None
	dgam = dgamma0.cuda()
	dr = ops.Dropout(drop, 99, 3)
	ops.layernorm_bwd(dy.cuda(), x.detach().cuda(), gamma.detach().cuda(), dx_in.cuda() if has_dx else None, dxo, gb, dgam, rows, E, seq_in=seq_in, seq_out=seq_out,
	                  seq_off=seq_off, dropout=dr)
	scale = float(want_dx.abs().max())
	assert float((dxo.cpu() - want_dx).abs().max()) <= 2e-5 * scale
	gs = float(gamma.grad.abs().max()) + 1.0
	assert float((dgam.cpu() - dgamma0 - gamma.grad).abs().max()) <= 1e-4 * gs   # fp32 atomics in arbitrary order over up to 1024 blocks
	if g_out:
		got = gb.float().cpu()
		if drop == 0.0:
			assert float((got - want_dx).abs().max()) <= scale * 2 ** -8
		else:  # every element is either dropped or scaled by 1/(1-p); the kept fraction matches p
			kept = got != 0
			assert float((got - want_dx / (1 - drop))[kept].abs().max()) <= scale / (1 - drop) * 2 ** -8
			frac = float(kept.float().mean())
			assert abs(frac - (1 - drop)) < 0.01
			# same site, same seed => the same mask again (the forward of the layer below draws it with identical arguments)
			gb2 = torch.empty_like(gb)
			ops.layernorm_bwd(dy.cuda(), x.detach().cuda(), gamma.detach().cuda(), dx_in.cuda() if has_dx else None, dxo, gb2, None, rows, E, seq_in=seq_in,
			                  seq_out=seq_out, seq_off=seq_off, dropout=dr)
			assert torch.equal(gb2 != 0, gb != 0)


def test_layernorm_backward_in_place():
	"""dx_out may alias dx_in (the training step accumulates the residual-stream gradient in one buffer)."""
	from novic_amd import ops
	rows, E = 5000, 512
	g = torch.Generator().manual_seed(5)
	x, gamma, dx = torch.randn(rows, E, generator=g).cuda(), torch.randn(E, generator=g).cuda(), torch.randn(rows, E, generator=g).cuda()
	dy = torch.randn(rows, E, generator=g).bfloat16().cuda()
	out = torch.empty_like(dx)
	ops.layernorm_bwd(dy, x, gamma, dx, out, None, None, rows, E)
	ops.layernorm_bwd(dy, x, gamma, dx, dx, None, None, rows, E)
	assert torch.equal(out, dx)


@pytest.mark.parametrize("V,ldl,A,T,C,col0,smoothing,argmax_from,tok_dtype", [
	(6912, 6912, 40, 7, 7, 0, 0.0, 0, torch.int64), (1000, 1008, 33, 3, 5, 1, 0.1, 1, torch.int32), (50, 56, 9, 2, 2, 0, 0.0, 0, torch.int64),
	(8192, 8192, 5, 4, 4, 0, 0.05, 0, torch.int64), (10000, 10000, 6, 2, 3, 1, 0.0, 1, torch.int64), (2048, 2048, 17, 1, 1, 0, 0.0, 0, torch.int32), (3, 8, 4, 1, 1, 0, 0.0, 0, torch.int64)])
def test_cross_entropy_rows(V, ldl, A, T, C, col0, smoothing, argmax_from, tok_dtype):
	"""novic_cross_entropy (register-resident rows up to 8192 columns, looping kernel beyond) against F.cross_entropy on the same bf16 logits
	(embedding_decoder.py:729-761): per-row loss within 1e-5 relative (fp32 log-sum-exp in another order), arg-max exact (lowest index on ties), the
	in-place gradient within bf16 rounding of scale * (softmax - smoothed one-hot), zero for padded / zero-weight rows and for the ldl - V pad columns."""
	from novic_amd import ops
	g = torch.Generator().manual_seed(V + A)
	logits = (torch.randn(A * T, ldl, generator=g) * 3).bfloat16()
	logits[0, :min(V, 5)] = 7.0  # ties: lowest eligible index wins
	target = torch.randint(0, V, (A, C), generator=g).to(tok_dtype)
	out_pad = (torch.rand(A, C, generator=g) < 0.2).to(torch.uint8)
	weight = torch.rand(A, generator=g) + 0.5
	weight[A // 2] = 0.0
	group_rows = max(1, A // 2)
	groups = (A + group_rows - 1) // group_rows
	basis = torch.rand(groups, generator=g) * 10 + 1
	grad_scale = 0.37
	dev = lambda t: t.cuda()
	lg = dev(logits.clone())
	row_loss = torch.empty(A * T, device="cuda")
	row_arg = torch.empty(A * T, dtype=torch.int32, device="cuda")
	row_cor = torch.empty(A * T, dtype=torch.uint8, device="cuda")
	ops.cross_entropy(lg, ldl, V, A, T, C, col0, dev(target), dev(out_pad), dev(weight), dev(basis), group_rows, grad_scale, smoothing, True, row_loss, row_arg, row_cor,
	                  argmax_from=argmax_from)
	x = logits[:, :V].float().requires_grad_(True)
	tg = target[:, col0:col0 + T].reshape(-1).long()
	ignored = (out_pad[:, col0:col0 + T].reshape(-1) != 0) | (weight.repeat_interleave(T) == 0)
	per = torch.nn.functional.cross_entropy(x, tg, reduction="none", label_smoothing=smoothing)
	want_loss = torch.where(ignored, torch.zeros_like(per), per)
	assert float((row_loss.cpu() - want_loss.detach()).abs().max()) <= 1e-5 * float(want_loss.abs().max() + 1)
	masked = x.detach().clone()
	masked[:, :argmax_from] = float("-inf")
	want_arg = masked.argmax(dim=1)  # torch returns the first maximal index
	assert torch.equal(row_arg.cpu().long(), want_arg)
	assert torch.equal(row_cor.cpu().bool(), (~ignored) & (want_arg == tg))
	sc = grad_scale * weight.repeat_interleave(T) / basis[torch.arange(A) // group_rows].repeat_interleave(T)
	sc = torch.where(ignored, torch.zeros_like(sc), sc)
	(per * sc).sum().backward()
	got = lg.float().cpu()
	tol = float(x.grad.abs().max()) * 2 ** -8 + 1e-8
	assert float((got[:, :V] - x.grad).abs().max()) <= tol
	assert float(got[:, V:].abs().max() if ldl > V else 0.0) == 0.0


@pytest.mark.parametrize("B,mrep,multi_first,S,P,E,V,tok_dtype", [
	(300, 1, False, 10, 4, 512, 50, torch.int64), (64, 3, False, 9, 4, 320, 1000, torch.int32), (64, 3, True, 9, 4, 320, 1000, torch.int64),
	(1, 1, False, 5, 4, 8, 3, torch.int64), (33, 1, False, 4, 4, 64, 10, torch.int64), (2000, 1, False, 10, 4, 512, 6912, torch.int64),
	(700, 1, False, 6, 1, 768, 7, torch.int64), (900, 1, False, 8, 4, 2048, 40000, torch.int64), (40, 1, False, 32, 4, 64, 100, torch.int64)])
def test_embed_backward(B, mrep, multi_first, S, P, E, V, tok_dtype):
	"""novic_embed_bwd against autograd of the layer-0 input assembly (embedding_decoder.py:665-675, :692-693): tied token-embedding gradient (fp32
	atomics), position gradient, bf16 prefix gradient summed over the sample's targets.  The first two ACCUMULATE into existing values.
	fp32 sums in another order: |err| <= 1e-5 * scale * sqrt(count)."""
	from novic_amd import ops
	A, L = B * mrep, S - P
	g = torch.Generator().manual_seed(B + S + V)
	dx0 = torch.randn(A, S, E, generator=g)
	tokens = torch.randint(0, V, (A, max(L, 1)), generator=g)
	if L > 0 and A > 4:
		tokens[:, L - 1] = 0  # every sequence ends in the END token
	dw0, dp0 = torch.randn(V, E, generator=g), torch.randn(S, E, generator=g)
	dw, dp = dw0.cuda(), dp0.cuda()
	dprefix = torch.empty(B, P * E, dtype=torch.bfloat16, device="cuda")
	ops.embed_bwd(dx0.cuda(), tokens.to(tok_dtype).cuda() if L > 0 else None, tokens.shape[1], dw, dp, dprefix, A, S, P, E, V, B, mrep, multi_first)
	want_dw = dw0.clone()
	if L > 0:
		want_dw.index_add_(0, tokens[:, :L].reshape(-1), dx0[:, P:].reshape(-1, E))
	want_dp = dp0 + dx0.sum(dim=0)
	sample_of = (torch.arange(A) % B) if multi_first else (torch.arange(A) // mrep)
	want_pre = torch.zeros(B, P, E).index_add_(0, sample_of, dx0[:, :P])
	tol = 1e-5 * float(dx0.abs().max()) * (A * max(L, 1)) ** 0.5 + 1e-6
	assert float((dw.cpu() - want_dw).abs().max()) <= tol
	assert float((dp.cpu() - want_dp).abs().max()) <= tol
	assert float((dprefix.float().cpu().view(B, P, E) - want_pre).abs().max()) <= float(want_pre.abs().max()) * 2 ** -8 + 1e-6


def test_transpose_bf16_batched():
	"""novic_transpose_bf16_batched: several matrices of one flat buffer -> their transposes in another (the W^T weight shadows), more than 32 per call,
	ragged tile counts; exact."""
	from novic_amd import ops
	g = torch.Generator().manual_seed(11)
	shapes = [(1536, 512), (512, 512), (128, 512), (6912, 512), (8, 8), (72, 200), (50, 64), (3, 5), (64, 37)] * 5  # 45 matrices, some with odd sizes
	desc, pos, dpos = [], 0, 0
	for k, (r, c) in enumerate(shapes):
		ld = r if k % 2 == 0 else (r + 7) // 8 * 8  # every other destination with a padded leading dimension
		desc.append((pos, dpos, r, c, ld))
		pos += (r * c + 7) // 8 * 8
		dpos += (c * ld + 7) // 8 * 8
	src = torch.randn(pos, generator=g).bfloat16().cuda()
	dst = torch.zeros(dpos, dtype=torch.bfloat16, device="cuda")
	ops.transpose_bf16_batched(src, dst, desc)
	for (o, d, r, c, ld) in desc:
		assert torch.equal(dst[d:d + c * ld].view(c, ld)[:, :r], src[o:o + r * c].view(r, c).t())


def test_compacted_loss_block_entry_points():
	"""novic_compact_rows + novic_layernorm_fwd_rows + novic_cross_entropy(row_map, row_limit) + novic_layernorm_bwd(dy_row) against the dense entry points:
	the rows that count get bit-identical results, the others zero loss / no gradient."""
	from novic_amd import ops
	A, S, T, C, E, V = 337, 10, 7, 7, 512, 1000  # 2359 token rows: three workgroups of the compaction
	g = torch.Generator().manual_seed(3)
	out_pad = (torch.rand(A, C, generator=g) < 0.35).to(torch.uint8)
	weight = torch.rand(A, generator=g) + 0.5
	weight[5] = 0.0
	R, M = A * T, A * S
	rows, src = torch.full((R,), -7, dtype=torch.int32, device="cuda"), torch.full((R,), -7, dtype=torch.int32, device="cuda")
	dst, cbuf = torch.full((M,), -7, dtype=torch.int32, device="cuda"), torch.zeros(1 + (R + 1023) // 1024, dtype=torch.int32, device="cuda")
	count = cbuf[:1]
	rl, ra, rc = torch.full((R,), 9.0, device="cuda"), torch.full((R,), 9, dtype=torch.int32, device="cuda"), torch.full((R,), 9, dtype=torch.uint8, device="cuda")
	ops.compact_rows(out_pad.cuda(), weight.cuda(), A, T, C, C - T, S, rows, src, dst, cbuf, rl, ra, rc)
	counts = ((out_pad == 0) & (weight != 0).unsqueeze(1)).reshape(-1)
	want_rows = counts.nonzero().flatten()
	n = int(count)
	assert n == len(want_rows) and torch.equal(rows[:n].cpu().long(), want_rows)
	a_of, t_of = want_rows // T, want_rows % T
	assert torch.equal(src[:n].cpu().long(), a_of * S + (S - T) + t_of)
	want_dst = torch.full((M,), -1, dtype=torch.long)
	want_dst[a_of * S + (S - T) + t_of] = torch.arange(n)
	assert torch.equal(dst.cpu().long(), want_dst)
	assert bool((rl.cpu()[~counts] == 0).all()) and bool((rl.cpu()[counts] == 9).all()) and bool((rc.cpu()[~counts] == 0).all())

	# final norm: gathered rows == the windowed dense rows
	x, gamma = torch.randn(M, E, generator=g).cuda(), torch.randn(E, generator=g).cuda()
	dense = torch.empty(R, E, dtype=torch.bfloat16, device="cuda")
	ops.layernorm_fwd(x, gamma, dense, R, E, seq_in=S, seq_out=T, seq_off=S - T)
	packed = torch.full((R, E), float("nan"), dtype=torch.bfloat16, device="cuda")
	ops.layernorm_fwd_rows(x, gamma, packed, src, count, R, E)
	assert torch.equal(packed[:n], dense[want_rows.cuda()]) and bool(torch.isnan(packed[n:].float()).all())

	# cross-entropy on compacted logits
	logits = (torch.randn(R, V, generator=g) * 2).bfloat16().cuda()
	target = torch.randint(0, V, (A, C), generator=g).cuda()
	basis = torch.ones(1, device="cuda")
	ld, la, lc = torch.zeros(R, device="cuda"), torch.zeros(R, dtype=torch.int32, device="cuda"), torch.zeros(R, dtype=torch.uint8, device="cuda")
	gd = logits.clone()
	ops.cross_entropy(gd, V, V, A, T, C, C - T, target, out_pad.cuda(), weight.cuda(), basis, A, 0.5, 0.0, True, ld, la, lc)
	gp = logits[want_rows.cuda()].clone().contiguous()
	pl, pa, pc = torch.zeros(R, device="cuda"), torch.zeros(R, dtype=torch.int32, device="cuda"), torch.zeros(R, dtype=torch.uint8, device="cuda")
	ops.cross_entropy(gp, V, V, A, T, C, C - T, target, out_pad.cuda(), weight.cuda(), basis, A, 0.5, 0.0, True, pl, pa, pc, row_map=rows, row_limit=count)
	assert torch.equal(gp, gd[want_rows.cuda()]) and torch.equal(pl, ld) and torch.equal(pc, lc)
	assert torch.equal(pa[want_rows.cuda()], la[want_rows.cuda()])

	# final norm backward: mapped upstream rows == windowed dense rows with zeros where nothing counts
	dy_dense = torch.zeros(R, E, dtype=torch.bfloat16, device="cuda")
	dy_dense[want_rows.cuda()] = torch.randn(n, E, generator=g).bfloat16().cuda()
	dy_packed = dy_dense[want_rows.cuda()].contiguous()
	dx_d, dx_p = torch.empty(M, E, device="cuda"), torch.empty(M, E, device="cuda")
	gb_d, gb_p = torch.empty(M, E, dtype=torch.bfloat16, device="cuda"), torch.empty(M, E, dtype=torch.bfloat16, device="cuda")
	dg_d, dg_p = torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda")
	ops.layernorm_bwd(dy_dense, x, gamma, None, dx_d, gb_d, dg_d, M, E, seq_in=S, seq_out=T, seq_off=S - T)
	ops.layernorm_bwd(dy_packed, x, gamma, None, dx_p, gb_p, dg_p, M, E, dy_row=dst)
	# (a row whose upstream gradient is all zero gives dx = 0 either way)
	assert torch.equal(dx_p, dx_d) and torch.equal(gb_p, gb_d)
	assert float((dg_p - dg_d).abs().max()) <= 1e-4 * float(dg_d.abs().max() + 1)


def test_packed_row_layout_entry_points():
	"""novic_seq_layout and the packed-row forms of embed / decoder attention against the dense [A][S] forms: a sequence keeps the positions in front of its
	padding suffix, lives at rows seq_start[a] .. + seq_len[a] - 1, and every kept row is bit-identical to its dense counterpart -- except the attention
	rows of sequences that share a merged tile (neighbours 2k, 2k + 1 whose rows fit 16 together): their soft-max sums run over the same numbers in
	another lane order, so they agree to the bf16 rounding of the outputs (<= 2^-7 of the largest value, relative L2 <= 2e-3)."""
	from novic_amd import ops
	A, S, P, E, H, V = 1500, 10, 4, 512, 8, 300
	D = E // H
	g = torch.Generator().manual_seed(8)
	lens = torch.randint(P, S + 1, (A,), generator=g)  # kept positions per sequence (prefix always kept)
	key_pad = (torch.arange(S).unsqueeze(0) >= lens.unsqueeze(1)).to(torch.uint8)
	key_pad[7, 5] = 0  # a hole in a padding suffix: everything up to the last unpadded position is kept
	lens[7] = max(int(lens[7]), 6)
	start, ln = torch.zeros(A, dtype=torch.int32, device="cuda"), torch.zeros(A, dtype=torch.int32, device="cuda")
	total = torch.zeros(1 + (A + 1023) // 1024, dtype=torch.int32, device="cuda")
	ops.seq_layout(key_pad.cuda(), A, S, start, ln, total)
	assert torch.equal(ln.cpu().long(), lens) and torch.equal(start.cpu().long(), torch.cumsum(lens, 0) - lens) and int(total[0]) == int(lens.sum())
	Mc = int(lens.sum())
	keep = (torch.arange(S).unsqueeze(0) < lens.unsqueeze(1)).reshape(-1).cuda()  # dense rows that exist in the packed layout, in packed order

	# embed forward
	prefix = torch.randn(A, P * E, generator=g).bfloat16().cuda()
	tokens = torch.randint(0, V, (A, S - P), generator=g).cuda()
	wtok, pos = torch.randn(V, E, generator=g).cuda(), torch.randn(S, E, generator=g).cuda()
	xd = torch.empty(A * S, E, device="cuda")
	xp = torch.full((A * S, E), float("nan"), device="cuda")
	ops.embed_fwd(prefix, tokens, S - P, wtok, pos, xd, A, S, P, E, V, A, 1, False)
	ops.embed_fwd(prefix, tokens, S - P, wtok, pos, xp, A, S, P, E, V, A, 1, False, seq=(start, ln))
	assert torch.equal(xp[:Mc], xd[keep]) and bool(torch.isnan(xp[Mc:]).all())

	# attention forward / backward
	qkv_d = (torch.randn(A * S, 3 * E, generator=g) * 0.5).bfloat16().cuda()
	do_d = torch.randn(A * S, E, generator=g).bfloat16().cuda()
	do_d[~keep] = 0  # nothing flows back into padded positions
	qkv_p, do_p = torch.zeros_like(qkv_d), torch.zeros_like(do_d)
	qkv_p[:Mc], do_p[:Mc] = qkv_d[keep], do_d[keep]
	kp = key_pad.cuda()
	od, op = torch.empty(A * S, E, dtype=torch.bfloat16, device="cuda"), torch.full((A * S, E), float("nan"), dtype=torch.bfloat16, device="cuda")
	ops.dec_attn_fwd(qkv_d, kp, od, A, S, H, D, P, False)
	ops.dec_attn_fwd(qkv_p, kp, op, A, S, H, D, P, False, seq=(start, ln))
	pair_sum = lens[0::2][:A // 2] + lens[1::2][:A // 2]
	merged_seq = torch.zeros(A, dtype=torch.bool)
	merged_seq[0:2 * (A // 2):2] = merged_seq[1:2 * (A // 2):2] = pair_sum <= 16
	assert 0.3 < float(merged_seq.float().mean()) < 1.0  # both kinds of tile occur
	single_row = (~merged_seq).repeat_interleave(lens).cuda()  # per packed row: does its sequence have a tile of its own?

	def same(packed, dense):
		assert torch.equal(packed[:Mc][single_row], dense[keep][single_row])
		a, b = packed[:Mc][~single_row].float(), dense[keep][~single_row].float()
		assert float((a - b).abs().max()) <= 2 ** -7 * float(b.abs().max()) and float((a - b).norm() / b.norm()) <= 2e-3

	same(op, od)
	assert bool(torch.isnan(op[Mc:].float()).all())
	gd, gp = torch.empty_like(qkv_d), torch.full_like(qkv_d, float("nan"))
	ops.dec_attn_bwd(qkv_d, kp, do_d, gd, A, S, H, D, P, False)
	ops.dec_attn_bwd(qkv_p, kp, do_p, gp, A, S, H, D, P, False, seq=(start, ln))
	same(gp, gd)


@pytest.mark.parametrize("rows,H,act,bias", [(513, 1024, "gelu", True), (64, 2048, "tanh", True), (1000, 1280, "relu", False), (7, 32, "gelu", False), (300, 256, "none", True)])
def test_hidden_norm_act_forward_and_backward(rows, H, act, bias):
	"""novic_hidden_norm_act_fwd / _bwd (round 5: the normalised hidden layer of the prefix MLP, reference embedding_decoder.py:1247-1253) against torch:
	y = act(LayerNorm(h0.float(); gamma, beta)) on a bf16 input, output bf16 (rel 2^-8); backward in fp32 on a bf16 upstream gradient, dh0 rounded to bf16 once."""
	from novic_amd import ops
	g = torch.Generator().manual_seed(rows + H)
	h0 = (torch.randn(rows, H, generator=g) * 1.5 + 0.2).bfloat16()
	gamma = (1 + 0.2 * torch.randn(H, generator=g)).requires_grad_(True)
	beta = (0.3 * torch.randn(H, generator=g)).requires_grad_(True) if bias else None
	dy = torch.randn(rows, H, generator=g).bfloat16()
	fn = {"gelu": torch.nn.functional.gelu, "tanh": torch.tanh, "relu": torch.relu, "none": lambda t: t}[act]
	code = {"gelu": ops.ACT_GELU, "tanh": ops.ACT_TANH, "relu": ops.ACT_RELU, "none": ops.ACT_NONE}[act]
	x = h0.float().requires_grad_(True)
	want = fn(torch.nn.functional.layer_norm(x, (H,), gamma, beta, 1e-5))
	want.backward(dy.float())
	y = torch.empty(rows, H, dtype=torch.bfloat16, device="cuda")
	ops.hidden_norm_act_fwd(h0.cuda(), gamma.detach().cuda(), beta.detach().cuda() if bias else None, y, rows, H, code)
	err = (y.float().cpu() - want.detach()).abs()
	assert float((err - want.detach().abs() * 2 ** -8).max()) <= 2e-6
	dh0 = torch.empty(rows, H, dtype=torch.bfloat16, device="cuda")
	dg0, db0 = torch.randn(H, generator=g), torch.randn(H, generator=g)
	dg, db = dg0.cuda(), (db0.cuda() if bias else None)
	ops.hidden_norm_act_bwd(dy.cuda(), h0.cuda(), gamma.detach().cuda(), beta.detach().cuda() if bias else None, dh0, dg, db, rows, H, code)
	scale = float(x.grad.abs().max())
	if act == "relu":  # the step at 0: an element whose normalised value is within rounding of zero may fall on either side
		z = torch.nn.functional.layer_norm(h0.float(), (H,), gamma.detach(), beta.detach() if bias else None, 1e-5)
		assert float((z.abs() < 1e-5).float().mean()) < 1e-3
	assert float((dh0.float().cpu() - x.grad).abs().max()) <= scale * 2 ** -7
	assert float((dg.cpu() - dg0 - gamma.grad).abs().max()) <= 2e-4 * (float(gamma.grad.abs().max()) + 1.0)
	if bias:
		assert float((db.cpu() - db0 - beta.grad).abs().max()) <= 2e-4 * (float(beta.grad.abs().max()) + 1.0)


@pytest.mark.parametrize("rows,E,has16,has32,bias,drop", [(700, 512, True, True, True, 0.0), (123, 320, True, False, False, 0.0), (4099, 64, False, True, True, 0.0),
                                                           (6000, 512, True, True, True, 0.1), (5, 2048, True, True, False, 0.0)])
def test_layernorm_backward_of_summed_gradients(rows, E, has16, has32, bias, drop):
	"""novic_layernorm_bwd_sum (round 5: the norms of post-LN layers, reference layer_norm_first = False): upstream gradient = bf16 part + fp32 part, against torch autograd of
	layer_norm; dgamma / dbeta accumulate on top of what is there; g_out = dx under the dropout mask of the site."""
	from novic_amd import ops
	g = torch.Generator().manual_seed(rows + E)
	x = (torch.randn(rows, E, generator=g) * 2 + 0.25).requires_grad_(True)
	gamma = torch.randn(E, generator=g).requires_grad_(True)
	beta = torch.randn(E, generator=g).requires_grad_(True)
	dy16 = torch.randn(rows, E, generator=g).bfloat16() if has16 else None
	dy32 = torch.randn(rows, E, generator=g) if has32 else None
	dy = (dy16.float() if has16 else 0) + (dy32 if has32 else 0)
	torch.nn.functional.layer_norm(x, (E,), gamma, beta, 1e-5).backward(dy)
	dg0, db0 = torch.randn(E, generator=g), torch.randn(E, generator=g)
	dxo = torch.full((rows, E), float("nan"), device="cuda")
	gb = torch.empty(rows, E, dtype=torch.bfloat16, device="cuda")
	dg, db = dg0.cuda(), (db0.cuda() if bias else None)
	ops.layernorm_bwd_sum(dy16.cuda() if has16 else None, dy32.cuda() if has32 else None, x.detach().cuda(), gamma.detach().cuda(), dxo, gb, dg, db, rows, E,
	                      dropout=ops.Dropout(drop, 77, 2))
	scale = float(x.grad.abs().max())
	assert float((dxo.cpu() - x.grad).abs().max()) <= 2e-5 * scale
	assert float((dg.cpu() - dg0 - gamma.grad).abs().max()) <= 1e-4 * (float(gamma.grad.abs().max()) + 1.0)
	if bias:
		assert float((db.cpu() - db0 - beta.grad).abs().max()) <= 1e-4 * (float(beta.grad.abs().max()) + 1.0)
	got = gb.float().cpu()
	if drop == 0.0:
		assert float((got - x.grad).abs().max()) <= scale * 2 ** -8
	else:
		kept = got != 0
		assert abs(float(kept.float().mean()) - (1 - drop)) < 0.01
		assert float((got - x.grad / (1 - drop))[kept].abs().max()) <= scale / (1 - drop) * 2 ** -8


@pytest.mark.parametrize("rows,E,drop", [(1000, 512, 0.0), (37, 64, 0.0), (5000, 512, 0.1)])
def test_rezero_scaling_forward_and_backward(rows, E, drop):
	"""novic_rezero_fwd / _bwd (round 5: reference TransformerEncoderLayer(rezero=...), embedding_decoder.py:1106-1116): out = resid + bf16(scale * branch) with the scalar on the
	device; backward g = bf16(dx), dscale += sum g * branch, g_out = bf16(bf16(g * scale) * mask) -- against the same arithmetic in torch; novic_add_bf16 on the way."""
	from novic_amd import ops
	g = torch.Generator().manual_seed(rows + E)
	resid, branch = torch.randn(rows, E, generator=g), torch.randn(rows, E, generator=g).bfloat16()
	scale = torch.tensor(0.73)
	out = torch.empty(rows, E, device="cuda")
	ops.rezero_fwd(resid.cuda(), branch.cuda(), scale.cuda(), out, rows, E)
	want = resid + (scale * branch.float()).bfloat16().float()
	assert torch.equal(out.cpu(), want)
	dx = torch.randn(rows, E, generator=g)
	dsc = torch.tensor(1.5).cuda()
	gout = torch.empty(rows, E, dtype=torch.bfloat16, device="cuda")
	ops.rezero_bwd(dx.cuda(), branch.cuda(), scale.cuda(), dsc, gout, rows, E, ops.Dropout(drop, 5, 9))
	gq = dx.bfloat16().float()
	ref_ds = float((gq.double() * branch.double()).sum())
	assert abs(float(dsc) - 1.5 - ref_ds) <= 1e-4 * (abs(ref_ds) + float((gq * branch.float()).abs().sum()) * 1e-2 + 1.0)
	wantg = (gq * scale).bfloat16().float()
	got = gout.float().cpu()
	if drop == 0.0:
		assert torch.equal(got, wantg)
	else:
		kept = got != 0
		assert abs(float(kept.float().mean()) - (1 - drop)) < 0.01
		assert float((got - (wantg / (1 - drop)).bfloat16().float())[kept].abs().max()) <= float(wantg.abs().max()) * 2 ** -7
	acc = torch.randn(rows, E, generator=g)
	accd = acc.cuda()
	ops.add_bf16(accd, branch.cuda())
	assert torch.equal(accd.cpu(), acc + branch.float())


@pytest.mark.parametrize("A,mrep,multi_first,S,P,E,V,packed,drop", [(1500, 1, False, 10, 4, 512, 300, True, 0.1), (96, 3, False, 9, 4, 512, 77, True, 0.0), (96, 3, True, 9, 4, 256, 77, False, 0.2),
                                                                     (40, 1, False, 5, 1, 64, 11, False, 0.0), (700, 2, False, 12, 4, 1024, 500, True, 0.1)])
def test_layer0_norm_fused_with_the_embedding_launches(A, mrep, multi_first, S, P, E, V, packed, drop):
	"""Round 6: novic_embed_fwd_ln = novic_embed_fwd + novic_layernorm_fwd (bit-identical: x0 and ln1), novic_ln_embed_bwd = novic_layernorm_bwd in front of novic_embed_bwd
	with the layer-0 input gradient never written -- the same gradients up to the order of the fp32 atomics and of one contraction (dprefix: a bf16 rounding step)."""
	from novic_amd import ops
	from novic_amd.ops import Dropout
	g = torch.Generator().manual_seed(A + S + E)
	B = A // mrep
	lens = torch.randint(P, S + 1, (A,), generator=g)
	key_pad = (torch.arange(S).unsqueeze(0) >= lens.unsqueeze(1)).to(torch.uint8)
	seq = None
	rows = A * S
	if packed:
		start, ln = torch.zeros(A, dtype=torch.int32, device="cuda"), torch.zeros(A, dtype=torch.int32, device="cuda")
		total = torch.zeros(1 + (A + 1023) // 1024, dtype=torch.int32, device="cuda")
		ops.seq_layout(key_pad.cuda(), A, S, start, ln, total)
		seq, rows = (start, ln), int(lens.sum())
	lim = torch.tensor([rows], dtype=torch.int32, device="cuda")
	prefix = torch.randn(B, P * E, generator=g).bfloat16().cuda()
	tokens = torch.randint(0, V, (A, S - P), generator=g).cuda()
	wtok, pos = torch.randn(V, E, generator=g).cuda(), torch.randn(S, E, generator=g).cuda()
	gamma = (1 + 0.2 * torch.randn(E, generator=g)).cuda()
	d_in = Dropout(drop, 0xABCDEF12345, 0)
	# forward
	x_a, x_b = torch.zeros(A * S, E, device="cuda"), torch.zeros(A * S, E, device="cuda")
	ln_a = torch.zeros(A * S, E, dtype=torch.bfloat16, device="cuda")
	ln_b = torch.zeros_like(ln_a)
	ops.embed_fwd(prefix, tokens, S - P, wtok, pos, x_a, A, S, P, E, V, B, mrep, multi_first, d_in, seq=seq)
	if packed:
		ops.layernorm_fwd_rows(x_a, gamma, ln_a, None, lim, A * S, E)
	else:
		ops.layernorm_fwd(x_a, gamma, ln_a, A * S, E)
	ops.embed_fwd_ln(prefix, tokens, S - P, wtok, pos, x_b, A, S, P, E, V, B, mrep, multi_first, gamma, ln_b, d_in, seq=seq)
	assert torch.equal(x_a[:rows], x_b[:rows]) and torch.equal(ln_a[:rows], ln_b[:rows])
	ref = torch.nn.functional.layer_norm(x_a[:rows], (E,), gamma)
	assert float((ln_b[:rows].float() - ref).abs().max()) <= 2 ** -7 * float(ref.abs().max())
	# backward: upstream = dln (bf16, through norm1) + dx (fp32, the residual path)
	dln = torch.randn(A * S, E, generator=g).bfloat16().cuda()
	dx = torch.randn(A * S, E, generator=g).cuda()
	if not packed:  # dense rows: the padded positions carry zero gradients (the reference masks them; here every row exists)
		pass
	outs = []
	for fused in (False, True):
		dgamma, dwtok, dpos = torch.zeros(E, device="cuda"), torch.zeros(V, E, device="cuda"), torch.zeros(S, E, device="cuda")
		dprefix = torch.zeros(B, P * E, dtype=torch.bfloat16, device="cuda")
		if fused:
			ops.ln_embed_bwd(dln, x_a, gamma, dx, dgamma, tokens, S - P, dwtok, dpos, dprefix, A, S, P, E, V, B, mrep, multi_first, d_in, seq=seq)
		else:
			dx0 = torch.zeros(A * S, E, device="cuda")
			ops.layernorm_bwd(dln, x_a, gamma, dx, dx0, None, dgamma, A * S, E, row_limit=lim if packed else None)
			ops.embed_bwd(dx0, tokens, S - P, dwtok, dpos, dprefix, A, S, P, E, V, B, mrep, multi_first, d_in, seq=seq)
		outs.append((dgamma, dwtok, dpos, dprefix.float()))
	for name, a, b in zip(("dgamma", "dwtok", "dpos", "dprefix"), *outs):
		scale = float(a.abs().max())
		tol = (2 ** -7 if name == "dprefix" else 2e-5) * scale
		assert float((a - b).abs().max()) <= tol, (name, float((a - b).abs().max()), scale)
		assert scale > 0
