"""Statistics of the dropout masks themselves (csrc/common.hpp: a 32-bit avalanche hash of (seed, site, element pair), 16-bit uniforms), recovered bit for bit
from kernels whose output is zero exactly where the mask is:

  * keep rate of every site within 4 sigma of 1 - p_eff (p_eff = round(65536 p) / 65536, the threshold the kernels compare against),
  * no correlation between sites, between seeds, between neighbouring rows, between neighbouring elements (the two halves of one hash word) beyond 4.5 sigma,
  * the backward kernels draw the SAME mask as the forward kernels of the same (seed, site): feed-forward GELU site (novic_ffn_fwd vs novic_ffn_bwd), the
    residual-branch site (out-projection epilogue vs the gradient leaving the LayerNorm backward, fused and unfused) -- bitwise.
reference: nn.Dropout inside nn.TransformerEncoderLayer as called from embedding_decoder.py:309-327 (an i.i.d. Bernoulli(1 - p) mask, scaled by 1 / (1 - p))."""
import math

import pytest
import torch

from novic_amd import ops
from novic_amd.ops import Dropout

pytestmark = pytest.mark.gpu
E, K = 512, 128


def _site_mask(M, N, drop):
	"""Keep-mask of site `drop` over an [M x N] activation, as the residual epilogue of novic_gemm_bf16 draws it: out = 0 + dropout(bf16(32)) is 32 / (1 - p) or 0."""
	a = torch.ones(M, 32, dtype=torch.bfloat16, device="cuda")
	w = torch.ones(N, 32, dtype=torch.bfloat16, device="cuda")
	out = torch.full((M, N), -1.0, device="cuda")
	ops.gemm(a, w, M, N, 32, kind=ops.EPI_RESID_F32, out=out, resid=torch.zeros(M, N, device="cuda"), dropout=drop)
	vals = torch.unique(out)
	assert vals.numel() <= 2 and abs(float(vals[-1]) - 32 / (1 - drop.p)) <= 0.26 and (vals.numel() == 1 or float(vals[0]) == 0.0), vals   # 32 / (1 - p), or exactly zero
	return out != 0


def _corr(a, b):
	a, b = a.double().flatten(), b.double().flatten()
	a, b = a - a.mean(), b - b.mean()
	return float((a * b).sum() / (a.norm() * b.norm()))


@pytest.mark.parametrize("p", [0.1, 0.5])
def test_keep_rate_and_independence(p):
	M, N = 4096, 512
	n = M * N
	p_eff = round(p * 65536) / 65536
	sigma_rate = math.sqrt(p_eff * (1 - p_eff) / n)
	seeds = (0x0123456789ABCDEF, 0x0123456789ABCDF0, 0xFFFFFFFF00000001)
	masks = {}
	for seed in seeds:
		for site in (0, 1, 2, 7, 8, 23):
			m = masks[(seed, site)] = _site_mask(M, N, Dropout(p, seed, site))
			keep = float(m.double().mean())
			assert abs(keep - (1 - p_eff)) <= 4 * sigma_rate, (hex(seed), site, keep)
			# per-row and per-column keep counts: no row or column is systematically favoured (binomial tails, 6 sigma over 4096 + 512 tests)
			rows, cols = m.double().mean(dim=1), m.double().mean(dim=0)
			assert float((rows - (1 - p_eff)).abs().max()) <= 6 * math.sqrt(p_eff * (1 - p_eff) / N)
			assert float((cols - (1 - p_eff)).abs().max()) <= 6 * math.sqrt(p_eff * (1 - p_eff) / M)
	tol = 4.5 / math.sqrt(n)
	keys = list(masks)
	for i in range(len(keys)):
		for j in range(i + 1, len(keys)):
			assert abs(_corr(masks[keys[i]], masks[keys[j]])) <= tol, (keys[i], keys[j])
	m = masks[(seeds[0], 0)]
	assert abs(_corr(m[:, 0::2], m[:, 1::2])) <= 4.5 / math.sqrt(n / 2)   # the two 16-bit halves of one hash word
	assert abs(_corr(m[:, 0:-2:2], m[:, 2::2])) <= 4.5 / math.sqrt(n / 2)  # neighbouring pairs (consecutive hash inputs)
	assert abs(_corr(m[:-1], m[1:])) <= tol                                 # neighbouring rows
	# the mask is a function of (seed, site, element index) only: the same site over a taller activation starts with the same rows
	assert torch.equal(_site_mask(2 * M, N, Dropout(p, seeds[0], 0))[:M], m)


def test_p_zero_and_tiny_p():
	assert bool(_site_mask(256, 512, Dropout(0.0, 5, 1)).all())
	m = _site_mask(4096, 512, Dropout(1.0 / 65536, 5, 1))  # the smallest non-zero threshold: about 32 of 2 M elements dropped
	dropped = int((~m).sum())
	assert 8 <= dropped <= 72, dropped


@pytest.mark.parametrize("M", [4096, 61])
def test_backward_draws_the_forward_mask(M):
	p, seed = 0.25, 0x5EED5EED12345678
	g = torch.Generator().manual_seed(M)
	# ---- feed-forward GELU site: forward mask from hact (zero where dropped, GELU(hpre) != 0 elsewhere), backward mask from dh ----
	xmid = torch.randn(M, E, generator=g).cuda()
	g2 = torch.ones(E, device="cuda")
	w1 = (torch.randn(K, E, generator=g) * 0.05).to(torch.bfloat16).cuda()
	w2 = (torch.randn(E, K, generator=g) * 0.08).to(torch.bfloat16).cuda()
	x = torch.zeros(M, E, device="cuda")
	hpre, hact = (torch.zeros(M, K, dtype=torch.bfloat16, device="cuda") for _ in range(2))
	ops.ffn_fwd(xmid, g2, w1, w2, x, M, E, K, hpre=hpre, hact=hact, dropout=Dropout(p, seed, 0), site_gelu=5, site_out=6)
	hact0 = torch.zeros_like(hact)
	ops.ffn_fwd(xmid, g2, w1, w2, torch.zeros(M, E, device="cuda"), M, E, K, hact=hact0)   # the same block without dropout
	visible = hact0 != 0
	fwd_gelu = hact != 0
	assert not bool((fwd_gelu & ~visible).any())
	want = _site_mask(M, K, Dropout(p, seed, 5))
	assert torch.equal(fwd_gelu[visible], want[visible])
	# residual-branch site of the same launch: x - xmid is zero where dropped
	x0 = torch.zeros(M, E, device="cuda")
	ops.ffn_fwd(xmid, g2, w1, w2, x0, M, E, K, hact=torch.zeros_like(hact), dropout=Dropout(p, seed, 0), site_gelu=5, site_out=999)  # (other out-site: only to get a second draw)
	delta, vis_out = x - xmid, None
	ref_out = _site_mask(M, E, Dropout(p, seed, 6))
	assert not bool(((delta != 0) & ~ref_out).any())                                  # nothing survives where the site's mask says dropped
	assert float(((delta != 0) == ref_out).double().mean()) >= 0.995                   # and (a branch value that rounds to zero aside) everything else does
	# backward: dh = bf16(gb W2) * mask * gelu'(hpre); g = bf16(dx * mask_g)
	gb = (torch.randn(M, E, generator=g) * 0.1).to(torch.bfloat16).cuda()
	dx_in = (torch.randn(M, E, generator=g) * 0.1).cuda()
	w2t, w1t = w2.T.contiguous(), w1.T.contiguous()
	dh, gout, dg = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda"), torch.zeros(M, E, dtype=torch.bfloat16, device="cuda"), torch.zeros(E, device="cuda")
	dx = dx_in.clone()
	ops.ffn_bwd(gb, hpre, xmid, dx, g2, w2t, w1t, dh, dx, gout, dg, M, E, K, dropout=Dropout(p, seed, 0), site_gelu=5, site_g=3)
	dh0, gout0 = torch.zeros_like(dh), torch.zeros_like(gout)
	dx0 = dx_in.clone()
	ops.ffn_bwd(gb, hpre, xmid, dx0, g2, w2t, w1t, dh0, dx0, gout0, torch.zeros(E, device="cuda"), M, E, K)
	vis = dh0 != 0
	assert torch.equal((dh != 0)[vis], want[vis]) and not bool(((dh != 0) & ~want).any())   # backward GELU-site mask == forward GELU-site mask, bitwise
	# the gradient that leaves the block towards the attention out-projection carries THAT site's forward mask (site 3 here: the out-projection's epilogue)
	fwd_site3 = _site_mask(M, E, Dropout(p, seed, 3))
	visg = gout0 != 0
	assert torch.equal((gout != 0)[visg], fwd_site3[visg]) and not bool(((gout != 0) & ~fwd_site3).any())
	# ... and so does the unfused LayerNorm backward
	r_g = torch.zeros(M, E, dtype=torch.bfloat16, device="cuda")
	dln = (torch.randn(M, E, generator=g) * 0.1).to(torch.bfloat16).cuda()
	ops.layernorm_bwd(dln, xmid, g2, dx_in, torch.zeros(M, E, device="cuda"), r_g, torch.zeros(E, device="cuda"), M, E, dropout=Dropout(p, seed, 3))
	r_g0 = torch.zeros_like(r_g)
	ops.layernorm_bwd(dln, xmid, g2, dx_in, torch.zeros(M, E, device="cuda"), r_g0, torch.zeros(E, device="cuda"), M, E)
	visr = r_g0 != 0
	assert torch.equal((r_g != 0)[visr], fwd_site3[visr]) and not bool(((r_g != 0) & ~fwd_site3).any())
