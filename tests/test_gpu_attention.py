"""Direct parity of the decoder self-attention kernels (novic_dec_attn_fwd / _bwd) against a torch fp32 restatement of
F.scaled_dot_product_attention with the reference's mask (embedding_decoder.py:651-654 prefix block + causal, :696-712 key padding), on the
same bf16-rounded inputs.  Tolerances: outputs are bf16 (rel 2^-8), probabilities are rounded to bf16 before PV like autocast's bmm operands:
|err| <= 2e-2 * scale.  Dropout: the mask the kernel drew is recovered exactly through one-hot values and fed to the torch restatement, which
checks that forward and BOTH backward layouts use the same mask."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _mask(S, P, strict, key_pad):
	i = torch.arange(S).view(S, 1)
	j = torch.arange(S).view(1, S)
	vis = (j <= i) | ((i < P) & (j < P) & (not strict))
	m = vis.unsqueeze(0) & ~(key_pad.bool() & (torch.arange(S) > 0)).unsqueeze(1)  # A x S x S
	return m


def _ref(qkv, mask, H, D, keep=None, p=0.0):
	"""qkv: A x S x 3E fp32 (requires_grad), mask A x S x S, keep A x H x S x S or None -> o A x S x E"""
	A, S, _ = qkv.shape
	E = H * D
	q, k, v = (qkv[..., x * E:(x + 1) * E].view(A, S, H, D).transpose(1, 2) for x in range(3))
	s = (q @ k.transpose(-1, -2)) / math.sqrt(D)
	s = s.masked_fill(~mask.unsqueeze(1), float("-inf"))
	pr = torch.softmax(s, dim=-1)
	if keep is not None:
		pr = pr * keep / (1 - p)
	return (pr @ v).transpose(1, 2).reshape(A, S, E)


@pytest.mark.parametrize("A,S,H,D,P,strict,drop", [(37, 10, 8, 64, 4, False, 0.0), (16, 13, 4, 32, 4, False, 0.0), (9, 20, 2, 64, 4, True, 0.0), (21, 7, 3, 16, 2, False, 0.0),
                                                 (300, 10, 8, 64, 4, False, 0.1), (40, 20, 4, 64, 4, False, 0.25), (33, 12, 4, 32, 3, False, 0.5)])
def test_attention_forward_backward(A, S, H, D, P, strict, drop):
	_run_case(A, S, H, D, P, strict, drop, packed=False)


@pytest.mark.parametrize("A,S,H,D,P,strict,drop", [(301, 10, 8, 64, 4, False, 0.0), (300, 10, 8, 64, 4, False, 0.1), (64, 13, 4, 32, 3, True, 0.25), (50, 20, 2, 64, 4, False, 0.1)])
def test_attention_packed_rows(A, S, H, D, P, strict, drop):
	"""The packed-row layout (sequence a at rows seq_start[a] .. + seq_len[a] - 1; for S <= 16 neighbouring sequences that fit 16 rows together share
	one tile): same checks as the dense layout on the rows that exist, the dropout mask recovered from the packed kernels themselves -- forward
	and both backward orientations of a merged tile must draw the same mask."""
	_run_case(A, S, H, D, P, strict, drop, packed=True)


def _run_case(A, S, H, D, P, strict, drop, packed):
	from novic_amd import ops
	E = H * D
	g = torch.Generator().manual_seed(A * 100 + S)
	qkv = (torch.randn(A, S, 3 * E, generator=g) * 0.7).to(torch.bfloat16)
	d_o = torch.randn(A, S, E, generator=g).to(torch.bfloat16)
	lens = torch.randint(P + 1, S + 1, (A,), generator=g)
	key_pad = (torch.arange(S).unsqueeze(0) >= lens.unsqueeze(1)).to(torch.uint8)
	mask = _mask(S, P, strict, key_pad)
	dd = ops.Dropout(drop, seed=4242, site=7)
	dq, dkp = qkv.cuda().view(A * S, 3 * E), key_pad.cuda()
	rows = torch.ones(A * S, dtype=torch.bool, device="cuda")  # dense rows that exist in the layout under test
	seq = None
	if packed:
		start, ln = torch.zeros(A, dtype=torch.int32, device="cuda"), torch.zeros(A, dtype=torch.int32, device="cuda")
		total = torch.zeros(1 + (A + 1023) // 1024, dtype=torch.int32, device="cuda")
		ops.seq_layout(dkp, A, S, start, ln, total)
		assert torch.equal(ln.cpu().long(), lens)
		rows = (torch.arange(S).unsqueeze(0) < lens.unsqueeze(1)).reshape(-1).cuda()
		seq = (start, ln)
	Mc = int(rows.sum())

	def to_layout(t):  # dense [A*S][*] -> the layout under test
		if not packed:
			return t
		out = torch.zeros_like(t)
		out[:Mc] = t[rows]
		return out

	def from_layout(t):  # back to dense [A*S][*]; rows that do not exist read 0
		if not packed:
			return t
		out = torch.zeros_like(t)
		out[rows] = t[:Mc]
		return out

	def fwd(x, out, **kw):
		tmp = torch.empty_like(out)
		ops.dec_attn_fwd(to_layout(x), dkp, tmp, A, S, H, D, P, strict, seq=seq, **kw)
		out.copy_(from_layout(tmp))

	keep = None
	if drop > 0:
		# values = one-hot of the key index: o[a, i, h*D + j] = P[a,h,i,j] * keep / (1-p); compared with the dropout-free run the mask falls out
		assert S <= D
		hot = qkv.clone().view(A, S, 3, H, D)
		hot[:, :, 2] = 0
		for j in range(S):
			hot[:, j, 2, :, j] = 1
		hot = hot.view(A * S, 3 * E).cuda()
		o0, o1 = torch.empty(A * S, E, dtype=torch.bfloat16, device="cuda"), torch.empty(A * S, E, dtype=torch.bfloat16, device="cuda")
		fwd(hot, o0)
		fwd(hot, o1, dropout=dd)
		p0 = o0.float().cpu().view(A, S, H, D)[..., :S].permute(0, 2, 1, 3)  # A x H x S(i) x S(j)
		p1 = o1.float().cpu().view(A, S, H, D)[..., :S].permute(0, 2, 1, 3)
		sure = (p0 > 1e-3) & rows.cpu().view(A, 1, S, 1)
		keep = torch.where(sure, (p1 > 0).float(), torch.ones_like(p0))
		frac = keep[sure].mean().item()
		assert abs(frac - (1 - drop)) < 0.03, frac
		torch.testing.assert_close(p1[sure & (p1 > 0)], (p0 / (1 - drop))[sure & (p1 > 0)], atol=2e-2, rtol=3e-2)
		# where the probability was too small to see the mask, make the reference agree with whatever the kernel did by zeroing those probabilities' effect
		mask = mask.unsqueeze(1).expand(A, H, S, S) & sure
		mask_for_ref = None
	o = torch.empty(A * S, E, dtype=torch.bfloat16, device="cuda")
	fwd(dq, o, dropout=dd)
	if packed:
		d_o = d_o * rows.cpu().view(A, S, 1)  # nothing flows back into positions that do not exist
	dqkv = torch.full((A * S, 3 * E), float("nan"), dtype=torch.bfloat16, device="cuda")
	ops.dec_attn_bwd(to_layout(dq), dkp, to_layout(d_o.cuda().view(A * S, E)), dqkv, A, S, H, D, P, strict, dropout=dd, seq=seq)
	dqkv = from_layout(dqkv) if packed else dqkv
	x = qkv.float().requires_grad_(True)
	if drop > 0:
		# reference with the recovered mask; entries with invisible probabilities (< 1e-3) contribute < 1e-3 * |v| either way
		q, k, v = (x[..., t * E:(t + 1) * E].view(A, S, H, D).transpose(1, 2) for t in range(3))
		s = (q @ k.transpose(-1, -2)) / math.sqrt(D)
		full = _mask(S, P, strict, key_pad).unsqueeze(1)
		pr = torch.softmax(s.masked_fill(~full, float("-inf")), dim=-1) * keep / (1 - drop)
		ref = (pr @ v).transpose(1, 2).reshape(A, S, E)
		tol = 4e-2
	else:
		ref = _ref(x, mask, H, D)
		tol = 2e-2
	ref.backward(d_o.float())
	valid = rows.cpu().view(A, S)  # dense: padded query rows are computed too (the reference does the same), all rows comparable
	got_o = o.float().cpu().view(A, S, E)
	assert torch.isfinite(got_o).all()
	assert float((got_o - ref.detach())[valid].abs().max()) <= tol * max(1.0, float(ref.abs().max()))
	got_g = dqkv.float().cpu().view(A, S, 3 * E)
	assert torch.isfinite(got_g).all()
	assert float((got_g - x.grad).abs().max()) <= tol * max(1.0, float(x.grad.abs().max())) * 1.5
	rel = float((got_g - x.grad).norm() / x.grad.norm())
	assert rel <= (2.5e-2 if drop > 0 else 1e-2), rel
