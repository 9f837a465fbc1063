"""The fused feed-forward launch (csrc/ffn.hip: norm2 + linear1 + GELU + dropout + linear2 + dropout + residual + the next layer's norm1) against the chain of
kernels it replaces -- bit for bit: the same MFMA, K accumulated in the same order, the same epilogue and LayerNorm helpers, the same dropout masks --
and against a torch fp32 restatement of the reference's norm_first feed-forward block (embedding_decoder.py:309-327)."""
import pytest
import torch

from novic_amd import ops
from novic_amd.ops import Dropout

pytestmark = pytest.mark.gpu
E, K = 512, 128


def _inputs(M, seed):
	g = torch.Generator().manual_seed(seed)
	xmid = torch.randn(M, E, generator=g).cuda()
	g2 = (1 + 0.1 * torch.randn(E, generator=g)).cuda()
	gn = (1 + 0.1 * torch.randn(E, generator=g)).cuda()
	w1 = (torch.randn(K, E, generator=g) * 0.05).to(torch.bfloat16).cuda()
	w2 = (torch.randn(E, K, generator=g) * 0.08).to(torch.bfloat16).cuda()
	return xmid, g2, gn, w1, w2


def _unfused(xmid, g2, gn, w1, w2, M, p, seed, lim):
	ln2 = torch.zeros(M, E, dtype=torch.bfloat16, device="cuda")
	hact, hpre = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda"), torch.zeros(M, K, dtype=torch.bfloat16, device="cuda")
	x, lnn = torch.zeros(M, E, device="cuda"), torch.zeros(M, E, dtype=torch.bfloat16, device="cuda")
	if lim is None:
		ops.layernorm_fwd(xmid, g2, ln2, M, E)
	else:
		ops.layernorm_fwd_rows(xmid, g2, ln2, None, lim, M, E)
	ops.gemm(ln2, w1, M, K, E, kind=ops.EPI_GELU_BF16, out=hact, out2=hpre, dropout=Dropout(p, seed, 7), row_limit=lim)
	ops.gemm(hact, w2, M, E, K, kind=ops.EPI_RESID_F32, out=x, resid=xmid, dropout=Dropout(p, seed, 8), row_limit=lim)
	if lim is None:
		ops.layernorm_fwd(x, gn, lnn, M, E)
	else:
		ops.layernorm_fwd_rows(x, gn, lnn, None, lim, M, E)
	return x, ln2, hpre, hact, lnn


def _ulp_close(got, want, what):
	"""bf16 LayerNorm outputs of two kernels that run the same operation sequence: equal except where the fp32 value sits within an ulp of a bf16 rounding
	tie and the two compilations of the statistics differ in its last bit (measured: 1-2 elements per million) -- there, one bf16 ulp."""
	g, w = got.float(), want.float()
	diff = (g - w).abs()
	bad = diff > 0
	assert int(bad.sum()) <= max(2, int(2e-5 * bad.numel())), (what, int(bad.sum()))
	assert bool((diff <= w.abs() * 2.0 ** -7 + 1e-30).all()), what


@pytest.mark.parametrize("M,p,limit", [(8192, 0.1, None), (8192, 0.0, None), (6000, 0.1, 4321), (61, 0.1, None), (16, 0.0, None), (5000, 0.1, 0)])
def test_fused_ffn_is_bit_identical_to_the_unfused_chain(M, p, limit):
	"""Stage by stage, each stage of the unfused chain fed with the FUSED kernel's own input to that stage: the two GEMMs with their epilogues (GELU, dropout,
	residual) must agree bit for bit; the two LayerNorms to the last fp32 bit of their statistics (_ulp_close)."""
	assert ops.ffn_fused_supported(E, K) and not ops.ffn_fused_supported(256, 64)
	xmid, g2, gn, w1, w2 = _inputs(M, seed=M + 3)
	lim = None if limit is None else torch.tensor([limit], dtype=torch.int32, device="cuda")
	seed = 0x1234567887654321
	x = torch.zeros(M, E, device="cuda")
	ln2, lnn = (torch.zeros(M, E, dtype=torch.bfloat16, device="cuda") for _ in range(2))
	hpre, hact = (torch.zeros(M, K, dtype=torch.bfloat16, device="cuda") for _ in range(2))
	ops.ffn_fwd(xmid, g2, w1, w2, x, M, E, K, gamma_next=gn, ln_next=lnn, ln2=ln2, hpre=hpre, hact=hact, dropout=Dropout(p, seed, 0), site_gelu=7, site_out=8, row_limit=lim)
	torch.cuda.synchronize()
	rows = M if limit is None else min(M, limit)
	for name, t in (("x", x), ("ln2", ln2), ("hpre", hpre), ("hact", hact), ("ln_next", lnn)):
		assert not bool(t[rows:].any()), name + " written beyond the row limit"
	if rows == 0:
		return
	ln = lambda src, gamma, dst: ops.layernorm_fwd(src, gamma, dst, M, E) if lim is None else ops.layernorm_fwd_rows(src, gamma, dst, None, lim, M, E)
	r_ln2, r_lnn = (torch.zeros(M, E, dtype=torch.bfloat16, device="cuda") for _ in range(2))
	ln(xmid, g2, r_ln2)
	_ulp_close(ln2[:rows], r_ln2[:rows], "ln2")
	r_hact, r_hpre = (torch.zeros(M, K, dtype=torch.bfloat16, device="cuda") for _ in range(2))
	ops.gemm(ln2, w1, M, K, E, kind=ops.EPI_GELU_BF16, out=r_hact, out2=r_hpre, dropout=Dropout(p, seed, 7), row_limit=lim)   # from the fused kernel's own ln2
	assert torch.equal(hpre[:rows], r_hpre[:rows]) and torch.equal(hact[:rows], r_hact[:rows])
	r_x = torch.zeros(M, E, device="cuda")
	ops.gemm(hact, w2, M, E, K, kind=ops.EPI_RESID_F32, out=r_x, resid=xmid, dropout=Dropout(p, seed, 8), row_limit=lim)
	assert torch.equal(x[:rows], r_x[:rows])
	ln(x, gn, r_lnn)
	_ulp_close(lnn[:rows], r_lnn[:rows], "ln_next")
	if p > 0:
		assert 0.05 < float((hact[:rows] == 0).float().mean()) < 0.2  # dropout really zeroes about p of the hidden units
	# end to end against the fully unfused chain: the rare LayerNorm ties move a handful of rows by a bf16 ulp of one hidden input
	full = _unfused(xmid, g2, gn, w1, w2, M, p, seed, lim)
	assert float((x[:rows] - full[0][:rows]).abs().max()) <= 5e-2 and float(((x[:rows] != full[0][:rows]).any(dim=1)).float().mean()) <= 5e-3
	# optional outputs: inference stores nothing but the residual stream
	x2 = torch.zeros(M, E, device="cuda")
	ops.ffn_fwd(xmid, g2, w1, w2, x2, M, E, K, dropout=Dropout(p, seed, 0), site_gelu=7, site_out=8, row_limit=lim)
	assert torch.equal(x2[:rows], x[:rows])


def test_fused_ffn_matches_a_torch_restatement_of_the_reference_block():
	M = 777
	xmid, g2, gn, w1, w2 = _inputs(M, seed=5)
	x = torch.zeros(M, E, device="cuda")
	lnn = torch.zeros(M, E, dtype=torch.bfloat16, device="cuda")
	ops.ffn_fwd(xmid, g2, w1, w2, x, M, E, K, gamma_next=gn, ln_next=lnn)
	xm = xmid.cpu()
	ln = torch.nn.functional.layer_norm(xm, (E,), g2.cpu(), None, 1e-5)
	h = torch.nn.functional.gelu(ln @ w1.float().cpu().T)
	want = xm + h @ w2.float().cpu().T
	assert float((x.cpu() - want).abs().max()) <= 2e-2 * float(want.abs().max())
	want_ln = torch.nn.functional.layer_norm(want, (E,), gn.cpu(), None, 1e-5)
	assert float((lnn.float().cpu() - want_ln).abs().max()) <= 3e-2 * float(want_ln.abs().max())


@pytest.mark.parametrize("M,p,limit", [(8192, 0.1, None), (6000, 0.0, 4321), (61, 0.1, None), (5000, 0.1, 0)])
def test_fused_ffn_backward_matches_the_unfused_chain(M, p, limit):
	"""novic_ffn_bwd (linear2 dX + GELU' + linear1 dX + norm2 backward) against novic_gemm_bf16(GELU_BWD) + novic_gemm_bf16(STORE) + novic_layernorm_bwd:
	dh bit for bit (a handful of a * g'(h) products on bf16 ties aside, as between the two unfused GEMM kernels); dx to fp32 rounding, g to one bf16 ulp of it,
	dgamma to the order of its atomics."""
	g = torch.Generator().manual_seed(M + 11)
	gb = (torch.randn(M, E, generator=g) * 0.1).to(torch.bfloat16).cuda()
	hpre = (torch.randn(M, K, generator=g)).to(torch.bfloat16).cuda()
	xmid = torch.randn(M, E, generator=g).cuda()
	dx_in = (torch.randn(M, E, generator=g) * 0.1).cuda()
	g2 = (1 + 0.1 * torch.randn(E, generator=g)).cuda()
	w1 = (torch.randn(K, E, generator=g) * 0.05).to(torch.bfloat16).cuda()
	w2 = (torch.randn(E, K, generator=g) * 0.08).to(torch.bfloat16).cuda()
	w2t, w1t = w2.T.contiguous(), w1.T.contiguous()   # [K][E], [E][K]
	lim = None if limit is None else torch.tensor([limit], dtype=torch.int32, device="cuda")
	rows = M if limit is None else min(M, limit)
	seed = 0x0BADC0DE12345678
	# unfused
	r_dh = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda")
	ops.gemm(gb, w2t, M, K, E, kind=ops.EPI_GELU_BWD_BF16, out=r_dh, resid=hpre, dropout=Dropout(p, seed, 5), row_limit=lim)
	r_dln = torch.zeros(M, E, dtype=torch.bfloat16, device="cuda")
	ops.gemm(r_dh, w1t, M, E, K, out=r_dln, row_limit=lim)
	r_dx, r_g, r_dg = torch.zeros(M, E, device="cuda"), torch.zeros(M, E, dtype=torch.bfloat16, device="cuda"), torch.zeros(E, device="cuda")
	ops.layernorm_bwd(r_dln, xmid, g2, dx_in, r_dx, r_g, r_dg, M, E, dropout=Dropout(p, seed, 3), row_limit=lim)
	# fused (dx written in place over a copy of dx_in, as the backward pass does)
	dh, gout, dg = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda"), torch.zeros(M, E, dtype=torch.bfloat16, device="cuda"), torch.zeros(E, device="cuda")
	dx = dx_in.clone()
	ops.ffn_bwd(gb, hpre, xmid, dx, g2, w2t, w1t, dh, dx, gout, dg, M, E, K, dropout=Dropout(p, seed, 0), site_gelu=5, site_g=3, row_limit=lim)
	torch.cuda.synchronize()
	assert torch.equal(dx[rows:], dx_in[rows:]) and not bool(dh[rows:].any()) and not bool(gout[rows:].any())
	if rows == 0:
		assert not bool(dg.any())
		return
	bad = dh[:rows] != r_dh[:rows]
	assert int(bad.sum()) <= max(2, int(1e-5 * bad.numel())), int(bad.sum())
	scale = float(r_dx[:rows].abs().max())
	assert float((dx[:rows] - r_dx[:rows]).abs().max()) <= 2e-2 * scale * (1 if bool(bad.any()) else 1e-3) + 1e-6   # fp32 rounding only, unless a dh tie moved
	assert float((dx[:rows] - r_dx[:rows]).abs().mean()) <= 1e-6 * scale + 1e-9
	gd = (gout[:rows].float() - r_g[:rows].float()).abs()
	assert float((gd > 0).float().mean()) <= 1e-3 and bool((gd <= r_g[:rows].float().abs() * 2.0 ** -7 + 1e-6).all())
	assert float((dg - r_dg).abs().max()) <= 1e-3 * float(r_dg.abs().max()) + 1e-6
	if p > 0:
		assert 0.05 < float((gout[:rows] == 0).float().mean()) < 0.2


@pytest.mark.parametrize("M,p,limit", [(8192, 0.1, None), (6000, 0.0, 4321), (61, 0.1, None), (4097, 0.1, None), (5000, 0.1, 0)])
def test_ffn_backward_with_the_norm1_prologue_matches_the_two_launches(M, p, limit):
	"""novic_ffn_bwd_ln = novic_layernorm_bwd (norm1 of the layer above) followed by novic_ffn_bwd, dx kept on chip in between: gb bit for bit up to a bf16 ulp where the
	fp32 dx differs in its last bits (two compilations of the same LayerNorm arithmetic), dh / dx / g as close as the unfused pair is to itself, both dgamma to atomics order."""
	g = torch.Generator().manual_seed(M + 23)
	dln_up = (torch.randn(M, E, generator=g) * 0.1).to(torch.bfloat16).cuda()
	x_up = torch.randn(M, E, generator=g).cuda()          # the upper norm's input = this block's output
	g1 = (1 + 0.1 * torch.randn(E, generator=g)).cuda()
	hpre = (torch.randn(M, K, generator=g)).to(torch.bfloat16).cuda()
	xmid = torch.randn(M, E, generator=g).cuda()
	dx_in = (torch.randn(M, E, generator=g) * 0.1).cuda()
	g2 = (1 + 0.1 * torch.randn(E, generator=g)).cuda()
	w1 = (torch.randn(K, E, generator=g) * 0.05).to(torch.bfloat16).cuda()
	w2 = (torch.randn(E, K, generator=g) * 0.08).to(torch.bfloat16).cuda()
	w2t, w1t = w2.T.contiguous(), w1.T.contiguous()
	lim = None if limit is None else torch.tensor([limit], dtype=torch.int32, device="cuda")
	rows = M if limit is None else min(M, limit)
	seed = 0x0123456776543210
	z = lambda *shape, dtype=torch.float32: torch.zeros(shape, dtype=dtype, device="cuda")
	# two launches
	r_dx1, r_gb, r_dg1 = z(M, E), z(M, E, dtype=torch.bfloat16), z(E)
	ops.layernorm_bwd(dln_up, x_up, g1, dx_in, r_dx1, r_gb, r_dg1, M, E, dropout=Dropout(p, seed, 9), row_limit=lim)
	r_dh, r_g, r_dg2 = z(M, K, dtype=torch.bfloat16), z(M, E, dtype=torch.bfloat16), z(E)
	r_dx = r_dx1.clone()
	ops.ffn_bwd(r_gb, hpre, xmid, r_dx, g2, w2t, w1t, r_dh, r_dx, r_g, r_dg2, M, E, K, dropout=Dropout(p, seed, 0), site_gelu=5, site_g=3, row_limit=lim)
	# one launch (dx in place)
	gb, dh, gout, dg1, dg2 = z(M, E, dtype=torch.bfloat16), z(M, K, dtype=torch.bfloat16), z(M, E, dtype=torch.bfloat16), z(E), z(E)
	dx = dx_in.clone()
	ops.ffn_bwd_ln(dln_up, x_up, g1, dg1, gb, hpre, xmid, dx, g2, w2t, w1t, dh, dx, gout, dg2, M, E, K, dropout=Dropout(p, seed, 0), site_pre=9, site_gelu=5, site_g=3, row_limit=lim)
	torch.cuda.synchronize()
	assert torch.equal(dx[rows:], dx_in[rows:]) and not bool(dh[rows:].any()) and not bool(gout[rows:].any()) and not bool(gb[rows:].any())
	if rows == 0:
		assert not bool(dg1.any()) and not bool(dg2.any())
		return
	gbd = (gb[:rows].float() - r_gb[:rows].float()).abs()
	assert float((gbd > 0).float().mean()) <= 1e-3 and bool((gbd <= r_gb[:rows].float().abs() * 2.0 ** -7 + 1e-6).all())
	assert torch.equal((gb[:rows] == 0), (r_gb[:rows] == 0)) or p == 0.0   # the same dropout mask
	dhd = (dh[:rows].float() - r_dh[:rows].float()).abs()
	assert float((dhd > 0).float().mean()) <= 2e-3 and float(dhd.max()) <= 2e-2 * float(r_dh[:rows].float().abs().max())
	scale = float(r_dx[:rows].abs().max())
	assert float((dx[:rows] - r_dx[:rows]).abs().max()) <= 2e-2 * scale and float((dx[:rows] - r_dx[:rows]).abs().mean()) <= 2e-5 * scale
	gd = (gout[:rows].float() - r_g[:rows].float()).abs()
	assert float((gd > 0).float().mean()) <= 5e-3 and float(gd.max()) <= 2e-2 * scale
	assert float((dg1 - r_dg1).abs().max()) <= 1e-3 * float(r_dg1.abs().max()) + 1e-6
	assert float((dg2 - r_dg2).abs().max()) <= 2e-3 * float(r_dg2.abs().max()) + 1e-6


@pytest.mark.parametrize("M,R,p", [(8192, 5000, 0.1), (300, 64, 0.0), (4097, 4097, 0.1)])
def test_ffn_backward_with_the_final_norm_prologue_over_mapped_rows(M, R, p):
	"""The same launch in front of the TOP layer: the final norm's backward, whose upstream gradient exists only for the compacted output rows (row m takes row map[m] of
	dy, none where map[m] < 0) and whose input receives nothing else (dx_in = None) -- against novic_layernorm_bwd(dy_row=map, dx_in=None) + novic_ffn_bwd."""
	g = torch.Generator().manual_seed(M + R)
	dy = (torch.randn(R, E, generator=g) * 0.1).to(torch.bfloat16).cuda()
	perm = torch.randperm(M, generator=g)[:min(R, M)]
	row_map = torch.full((M,), -1, dtype=torch.int32)
	row_map[perm] = torch.arange(perm.numel(), dtype=torch.int32)
	row_map = row_map.cuda()
	x_up = torch.randn(M, E, generator=g).cuda()
	gf = (1 + 0.1 * torch.randn(E, generator=g)).cuda()
	hpre = (torch.randn(M, K, generator=g)).to(torch.bfloat16).cuda()
	xmid = torch.randn(M, E, generator=g).cuda()
	g2 = (1 + 0.1 * torch.randn(E, generator=g)).cuda()
	w1 = (torch.randn(K, E, generator=g) * 0.05).to(torch.bfloat16).cuda()
	w2 = (torch.randn(E, K, generator=g) * 0.08).to(torch.bfloat16).cuda()
	w2t, w1t = w2.T.contiguous(), w1.T.contiguous()
	seed = 0x0F0F0F0F12121212
	z = lambda *shape, dtype=torch.float32: torch.zeros(shape, dtype=dtype, device="cuda")
	r_dx, r_gb, r_dgf = z(M, E), z(M, E, dtype=torch.bfloat16), z(E)
	ops.layernorm_bwd(dy, x_up, gf, None, r_dx, r_gb, r_dgf, M, E, dropout=Dropout(p, seed, 9), dy_row=row_map)
	r_dh, r_g, r_dg2 = z(M, K, dtype=torch.bfloat16), z(M, E, dtype=torch.bfloat16), z(E)
	ops.ffn_bwd(r_gb, hpre, xmid, r_dx, g2, w2t, w1t, r_dh, r_dx, r_g, r_dg2, M, E, K, dropout=Dropout(p, seed, 0), site_gelu=5, site_g=3)
	gb, dh, gout, dgf, dg2 = z(M, E, dtype=torch.bfloat16), z(M, K, dtype=torch.bfloat16), z(M, E, dtype=torch.bfloat16), z(E), z(E)
	dx = torch.full((M, E), float("nan"), device="cuda")   # written, never read: dx_in is None
	ops.ffn_bwd_ln(dy, x_up, gf, dgf, gb, hpre, xmid, None, g2, w2t, w1t, dh, dx, gout, dg2, M, E, K, dropout=Dropout(p, seed, 0), site_pre=9, site_gelu=5, site_g=3,
	               pre_row_map=row_map)
	torch.cuda.synchronize()
	none = row_map < 0
	assert not bool(gb[none].any())   # rows without an upstream gradient: the norm passes nothing on
	gbd = (gb.float() - r_gb.float()).abs()
	assert float((gbd > 0).float().mean()) <= 1e-3 and bool((gbd <= r_gb.float().abs() * 2.0 ** -7 + 1e-6).all())
	scale = float(r_dx.abs().max())
	assert bool(torch.isfinite(dx).all()) and float((dx - r_dx).abs().max()) <= 2e-2 * scale and float((dx - r_dx).abs().mean()) <= 2e-5 * scale
	assert float((dh.float() - r_dh.float()).abs().max()) <= 2e-2 * float(r_dh.float().abs().max()) + 1e-6
	assert float((gout.float() - r_g.float()).abs().max()) <= 2e-2 * scale
	assert float((dgf - r_dgf).abs().max()) <= 1e-3 * float(r_dgf.abs().max()) + 1e-6
	assert float((dg2 - r_dg2).abs().max()) <= 2e-3 * float(r_dg2.abs().max()) + 1e-6
