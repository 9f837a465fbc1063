"""GPU parity of the MFMA GEMM (all operand storages and epilogues) against fp32 torch matmul of the same bf16 operands."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(shape, seed, scale=1.0):
	g = torch.Generator(device="cpu").manual_seed(seed)
	return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).cuda()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 72), (517, 1536, 512), (1030, 128, 512), (640, 512, 128), (96, 307, 512), (33, 64, 16)])
def test_gemm_forward_layout(M, N, K):
	from novic_amd import ops
	a, b = _mk((M, K), 1), _mk((N, K), 2)
	out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda") if N % 8 == 0 else torch.full((M, (N + 7) // 8 * 8), float("nan"), dtype=torch.bfloat16, device="cuda")
	ops.gemm(a, b, M, N, K, out=out)
	ref = a.float() @ b.float().T
	got = out[:, :N].float()
	assert torch.isfinite(got).all()
	torch.testing.assert_close(got, ref.to(torch.bfloat16).float(), atol=2e-2 * math.sqrt(K / 64), rtol=2e-2)
	# exactness check with small integers (no rounding anywhere): catches any fragment/layout mix-up
	ai = torch.randint(-3, 4, (M, K), generator=torch.Generator().manual_seed(3)).to(torch.bfloat16).cuda()
	bi = torch.randint(-3, 4, (N, K), generator=torch.Generator().manual_seed(4)).to(torch.bfloat16).cuda()
	o32 = torch.zeros(M, (N + 3) // 4 * 4, dtype=torch.float32, device="cuda")
	ops.gemm(ai, bi, M, N, K, kind=ops.EPI_STORE_F32, out=o32)
	assert torch.equal(o32[:, :N], ai.float() @ bi.float().T)


@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (517, 512, 1536), (300, 64, 307), (130, 512, 128)])
def test_gemm_input_grad_layout(M, N, K):
	"""dX[M,N] = dY[M,K] * W[K,N]  (B stored k-strided: [K][N])."""
	from novic_amd import ops
	Kp = (K + 7) // 8 * 8
	ai = torch.zeros(M, Kp, dtype=torch.bfloat16)
	ai[:, :K] = torch.randint(-3, 4, (M, K), generator=torch.Generator().manual_seed(5)).to(torch.bfloat16)
	ai = ai.cuda()
	bi = torch.randint(-3, 4, (K, N), generator=torch.Generator().manual_seed(6)).to(torch.bfloat16).cuda()
	o32 = torch.zeros(M, N, dtype=torch.float32, device="cuda")
	ops.gemm(ai, bi, M, N, K, b_kstrided=True, kind=ops.EPI_STORE_F32, out=o32)
	assert torch.equal(o32, ai[:, :K].float() @ bi.float())


@pytest.mark.parametrize("M,N,K,split", [(128, 128, 256, 1), (1536, 512, 1000, 4), (128, 512, 5000, 8), (307, 512, 777, 3), (64, 16, 100, 2)])
def test_gemm_weight_grad_layout(M, N, K, split):
	"""dW[M,N] += dY^T * X with dY stored [K][M] and X stored [K][N] (both k-strided), split-K with fp32 atomics."""
	from novic_amd import ops
	Mp = (M + 7) // 8 * 8
	ai = torch.zeros(K, Mp, dtype=torch.bfloat16)
	ai[:, :M] = torch.randint(-2, 3, (K, M), generator=torch.Generator().manual_seed(7)).to(torch.bfloat16)
	ai = ai.cuda()
	bi = torch.randint(-2, 3, (K, N), generator=torch.Generator().manual_seed(8)).to(torch.bfloat16).cuda()
	o32 = torch.ones(M, N, dtype=torch.float32, device="cuda")
	ops.gemm(ai, bi, M, N, K, a_kstrided=True, b_kstrided=True, kind=ops.EPI_ATOMIC_F32, out=o32, split_k=split, alpha=0.5)
	assert torch.equal(o32, 1 + 0.5 * (ai[:, :M].float().T @ bi.float()))


def test_gemm_epilogues():
	from novic_amd import ops
	M, N, K = 260, 512, 128
	a, b = _mk((M, K), 11, 0.5), _mk((N, K), 12, 0.2)
	ref = (a.float() @ b.float().T).to(torch.bfloat16).float()
	resid = torch.randn(M, N, device="cuda")
	out = torch.empty(M, N, device="cuda")
	ops.gemm(a, b, M, N, K, kind=ops.EPI_RESID_F32, out=out, resid=resid)
	torch.testing.assert_close(out, resid + ref, atol=2e-2, rtol=2e-2)
	# dropout: mask is deterministic in (seed, site, index), keeps ~1-p, scales by 1/(1-p)
	d = ops.Dropout(0.25, seed=1234, site=3)
	out_d = torch.empty(M, N, device="cuda")
	ops.gemm(a, b, M, N, K, kind=ops.EPI_RESID_F32, out=out_d, resid=torch.zeros_like(resid), dropout=d)
	out_d2 = torch.empty(M, N, device="cuda")
	ops.gemm(a, b, M, N, K, kind=ops.EPI_RESID_F32, out=out_d2, resid=torch.zeros_like(resid), dropout=d)
	assert torch.equal(out_d, out_d2)
	kept = out_d != 0
	assert abs(kept.float().mean().item() - 0.75) < 0.01
	torch.testing.assert_close(out_d[kept], (ref / 0.75)[kept], atol=3e-2, rtol=3e-2)
	# GELU forward + its backward epilogue
	hact = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
	hpre = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
	ops.gemm(a, b, M, N, K, kind=ops.EPI_GELU_BF16, out=hact, out2=hpre)
	torch.testing.assert_close(hpre.float(), ref, atol=2e-2, rtol=2e-2)
	torch.testing.assert_close(hact.float(), torch.nn.functional.gelu(hpre.float()).to(torch.bfloat16).float(), atol=1e-2, rtol=1e-2)
	dh = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
	ops.gemm(a, b, M, N, K, kind=ops.EPI_GELU_BWD_BF16, out=dh, resid=hpre)
	x = hpre.float().requires_grad_(True)
	torch.nn.functional.gelu(x).backward(ref)
	torch.testing.assert_close(dh.float(), x.grad, atol=3e-2, rtol=3e-2)


@pytest.mark.parametrize("M,N,K,mode", [
	(61519, 1536, 512, "bf16"),       # the step's QKV GEMM at the packed row count: 241 x 6 = 1446 tiles, 5.6 rounds, ragged last row tile
	(36943, 6912, 512, "bf16"),       # logits GEMM on the compacted rows: 145 x 27 tiles
	(61519, 512, 1536, "bf16"),       # in-projection input gradient: two tile columns, 24 K-tiles
	(16384, 1536, 128, "bf16"),       # TWO K-tiles per tile: every K-tile of the stream is a tile's first or last
	(16384, 1536, 192, "bf16"),       # three (odd: the buffer parity flips from tile to tile)
	(16384, 1040, 320, "bf16"),       # five K-tiles, ragged N edge (1040 = 4 x 256 + 16)
	(8192, 2048, 512, "gelu"),        # GELU epilogue with saved pre-activations + dropout
	(65792, 1024, 1024, "resid"),     # ViT-L/14 proj at batch 256: fp32 residual epilogue, 1028 tiles -> host-planned K-split tail (4 tiles x 4 parts)
	(16448, 1024, 4096, "resid"),     # ViT-L/14 fc2 at batch 64: 260 tiles, K-split tail of 4 tiles x 16 parts
	(12800, 3072, 768, "bias_qgelu"), # ViT-B/32 fc1 at batch 256: bias + QuickGELU, 600 tiles
	(57344, 512, 6912, "rowlimit"),   # logits input gradient with a DEVICE row count (36 943 of 57 344) and the K-split tail planned on the device
	(700, 2048, 512, "bf16_forced"),  # 3 x 8 = 24 tiles on 24 workgroups (forced 256 tile): one tile per workgroup, prologue + tail only
])
def test_8phase_gemm_kernel_is_bit_identical_to_the_one_barrier_kernel(M, N, K, mode):
	"""gemm256p_kernel (8-phase K loop: staggered wave groups, half-tile LDS-DMA six half-tiles ahead across output tiles, counted vmcnt, the epilogue's stores left in
	flight) against gemm256_kernel<EPI, 4> (one barrier + vmcnt(0) per K-tile): same LDS image, fragment addresses and MFMA order per accumulator, so BIT-identical
	outputs for every epilogue.  Repeated: a read that overtakes its LDS-DMA or a DMA that lands on fragments still being read comes and goes with timing."""
	from novic_amd import ops
	a, b = _mk((M, K), 31, 0.5), _mk((N, K), 32, 0.2)
	kw, pol = {}, 2  # the 256 x 256 tile whatever the tile count (the policy's choice is not what this test is about)
	if mode == "gelu":
		kw = dict(kind=ops.EPI_GELU_BF16, dropout=ops.Dropout(0.1, seed=7, site=3))
	elif mode == "resid":
		kw = dict(kind=ops.EPI_RESID_F32, resid=torch.randn(M, N, device="cuda"), bias=torch.randn(N, device="cuda"), split_tail=True)
	elif mode == "bias_qgelu":
		kw = dict(bias=torch.randn(N, device="cuda"), act=ops.ACT_QUICKGELU)
	elif mode == "rowlimit":
		kw, pol = dict(row_limit=torch.tensor([36943], dtype=torch.int32, device="cuda"), split_tail=True), 1  # (the device-planned tail is the policy's own choice)

	def run():
		dt = torch.float32 if mode == "resid" else torch.bfloat16
		o = torch.zeros((M, N), dtype=dt, device="cuda")
		extra = {}
		if mode == "gelu":
			extra["out2"] = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
		ops.gemm(a, b, M, N, K, out=o, **kw, **extra)
		return [o] + list(extra.values())

	prev_pol, prev_pipe = ops.gemm_tile_policy(pol), ops.gemm256_pipeline(0)
	try:
		ref = run()
		assert ops.gemm_last_tile() == 256
		ops.gemm256_pipeline(1)
		outs = [run() for _ in range(8)]
		assert ops.gemm_last_tile() == 256
		torch.cuda.synchronize()
	finally:
		ops.gemm_tile_policy(prev_pol)
		ops.gemm256_pipeline(prev_pipe)
	assert float(ref[0].float().abs().max()) > 0
	for rep, out in enumerate(outs):
		for x, y in zip(out, ref):
			assert torch.equal(x, y), (rep, float((x.float() - y.float()).abs().max()))


@pytest.mark.parametrize("M,N,K", [(8192, 2048, 192), (8000, 2052, 128), (16384, 1536, 512), (20000, 1024, 64), (2600, 6912, 512)])
def test_large_tile_kernel_is_bit_identical(M, N, K):
	"""Problems with >= 256 tiles of 256x256 in >= 4 tile columns run on the 256^2-tile LDS-DMA kernel; it accumulates K in the same order with the same MFMA and shares
	the epilogue code, so every epilogue must match the 128^2-tile kernel bit for bit (ragged M / N edges, dropout masks, saved pre-activations)."""
	from novic_amd import ops
	a, b = _mk((M, K), 21, 0.5), _mk((N, K), 22, 0.2)
	resid = torch.randn(M, N, device="cuda")
	hpre = _mk((M, N), 23)
	d = ops.Dropout(0.1, seed=77, site=5)

	def run_all():
		outs = []
		o = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
		ops.gemm(a, b, M, N, K, out=o)
		outs.append(o)
		o = torch.full((M, N), float("nan"), device="cuda")
		ops.gemm(a, b, M, N, K, kind=ops.EPI_STORE_F32, out=o)
		outs.append(o)
		o = torch.full((M, N), float("nan"), device="cuda")
		ops.gemm(a, b, M, N, K, kind=ops.EPI_RESID_F32, out=o, resid=resid, dropout=d)
		outs.append(o)
		o, o2 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda"), torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
		ops.gemm(a, b, M, N, K, kind=ops.EPI_GELU_BF16, out=o, out2=o2, dropout=d)
		outs += [o, o2]
		o = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
		ops.gemm(a, b, M, N, K, kind=ops.EPI_GELU_BWD_BF16, out=o, resid=hpre, dropout=d)
		outs.append(o)
		return outs

	prev = ops.gemm_tile_policy(0)
	try:
		small = run_all()
		assert ops.gemm_last_tile() == 128
		ops.gemm_tile_policy(1)
		large = run_all()
		assert ops.gemm_last_tile() == 256
	finally:
		ops.gemm_tile_policy(prev)
	for x, y in zip(small, large):
		assert torch.isfinite(x.float()).all() and torch.equal(x, y)
	ref = a.float() @ b.float().T
	torch.testing.assert_close(large[1], ref, atol=2e-2 * math.sqrt(K / 64), rtol=2e-2)
	ai = torch.randint(-3, 4, (M, K), generator=torch.Generator().manual_seed(3)).to(torch.bfloat16).cuda()
	bi = torch.randint(-3, 4, (N, K), generator=torch.Generator().manual_seed(4)).to(torch.bfloat16).cuda()
	o32 = torch.zeros(M, N, device="cuda")
	ops.gemm(ai, bi, M, N, K, kind=ops.EPI_STORE_F32, out=o32)
	assert torch.equal(o32, ai.float() @ bi.float().T)


@pytest.mark.parametrize("M,N,K", [(12800, 768, 256), (700, 580, 128), (256, 192, 64), (1000, 2304, 192)])
def test_wide192_tile_bit_identical(M, N, K):
	"""The 256 x 192 variant of the LDS-DMA kernel (three MFMA column tiles per wave, B rows in natural order, epilogue straight from the accumulators):
	forced through the tile policy it must reproduce the 128^2 kernel bit for bit on every epilogue, ragged edges included; left to choose, the
	fp32-residual GEMM of ViT-B/32 at batch 256 ([12800 x 768 x K]: 200 tiles in one round) takes it by itself."""
	from novic_amd import ops
	a, b = _mk((M, K), 31, 0.5), _mk((N, K), 32, 0.2)
	resid = torch.randn(M, N, device="cuda")
	hpre = _mk((M, N), 33)
	bias = torch.randn(N, device="cuda")
	d = ops.Dropout(0.1, seed=78, site=6)

	def run_all():
		outs = []
		o = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
		ops.gemm(a, b, M, N, K, out=o, bias=bias, act=ops.ACT_QUICKGELU)
		outs.append(o)
		o = torch.full((M, N), float("nan"), device="cuda")
		ops.gemm(a, b, M, N, K, kind=ops.EPI_STORE_F32, out=o)
		outs.append(o)
		o = torch.full((M, N), float("nan"), device="cuda")
		ops.gemm(a, b, M, N, K, kind=ops.EPI_RESID_F32, out=o, resid=resid, bias=bias, dropout=d)
		outs.append(o)
		o, o2 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda"), torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
		ops.gemm(a, b, M, N, K, kind=ops.EPI_GELU_BF16, out=o, out2=o2, dropout=d)
		outs += [o, o2]
		o = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
		ops.gemm(a, b, M, N, K, kind=ops.EPI_GELU_BWD_BF16, out=o, resid=hpre, dropout=d)
		outs.append(o)
		return outs

	prev = ops.gemm_tile_policy(0)
	try:
		small = run_all()
		assert ops.gemm_last_tile() == 128
		ops.gemm_tile_policy(3)
		wide = run_all()
		assert ops.gemm_last_tile() == 192
		ops.gemm_tile_policy(1)
		o = torch.full((M, N), float("nan"), device="cuda")
		ops.gemm(a, b, M, N, K, kind=ops.EPI_RESID_F32, out=o, resid=resid, bias=bias, dropout=d)
		# (left to choose, ViT-B/32's fp32-residual GEMM now takes the 256 x 256 tile on the 8-phase K loop; the 192-wide tile when that schedule is switched off)
		assert ops.gemm_last_tile() == (256 if (M, N) == (12800, 768) else 128 if N < 1024 else ops.gemm_last_tile())
		if (M, N) == (12800, 768):
			pp = ops.gemm256_pipeline(0)
			ops.gemm(a, b, M, N, K, kind=ops.EPI_RESID_F32, out=o, resid=resid, bias=bias, dropout=d)
			ops.gemm256_pipeline(pp)
			assert ops.gemm_last_tile() == 192
		assert torch.equal(o, small[2])
	finally:
		ops.gemm_tile_policy(prev)
	for x, y in zip(small, wide):
		assert torch.isfinite(x.float()).all() and torch.equal(x, y)


@pytest.mark.parametrize("M,N,K", [(700, 580, 128), (8000, 2052, 128), (520, 1024, 64)])
def test_forced_256_tile_with_bias_and_activation_on_ragged_edges(M, N, K):
	"""Edge tiles of the 256 x 256 kernel leave through the per-element epilogue: bias and GELU / QuickGELU must be applied there too (the whole-line
	store path only serves interior tiles)."""
	from novic_amd import ops
	a, b = _mk((M, K), 41, 0.5), _mk((N, K), 42, 0.2)
	bias = torch.randn(N, device="cuda")
	prev = ops.gemm_tile_policy(0)
	try:
		outs = {}
		for pol in (0, 2):
			ops.gemm_tile_policy(pol)
			res = []
			for act, bb in ((ops.ACT_NONE, None), (ops.ACT_NONE, bias), (ops.ACT_GELU, bias), (ops.ACT_QUICKGELU, bias), (ops.ACT_QUICKGELU, None)):
				o = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
				ops.gemm(a, b, M, N, K, out=o, bias=bb, act=act)
				res.append(o)
			assert ops.gemm_last_tile() == (128 if pol == 0 else 256)
			outs[pol] = res
	finally:
		ops.gemm_tile_policy(prev)
	for x, y in zip(outs[0], outs[2]):
		assert torch.isfinite(x.float()).all() and torch.equal(x, y)
	ref = torch.nn.functional.gelu(a.float() @ b.float().T + bias)
	torch.testing.assert_close(outs[2][2].float(), ref, atol=3e-2, rtol=3e-2)


@pytest.mark.parametrize("M,N,K", [(700, 580, 128), (1024, 512, 256), (300, 200, 72)])
def test_tanh_gelu_epilogue(M, N, K):
	"""NOVIC_ACT_GELU_TANH (the SigLIP configs with act_kwargs.approximate = 'tanh'): 128^2 and 256-wide kernels, interior tiles (whole-line store path) and edge tiles,
	against torch's gelu(approximate='tanh') of the fp32 product; the two kernels bit for bit."""
	from novic_amd import ops
	a, b = _mk((M, K), 43, 0.5), _mk((N, K), 44, 0.2)
	bias = torch.randn(N, device="cuda")
	ref = torch.nn.functional.gelu(a.float() @ b.float().T + bias, approximate="tanh")
	prev = ops.gemm_tile_policy(0)
	outs = []
	try:
		for pol in (0, 2):
			ops.gemm_tile_policy(pol)
			o = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
			ops.gemm(a, b, M, N, K, out=o, bias=bias, act=ops.ACT_GELU_TANH)
			outs.append(o)
	finally:
		ops.gemm_tile_policy(prev)
	assert torch.equal(outs[0], outs[1])
	torch.testing.assert_close(outs[0].float(), ref, atol=3e-2, rtol=3e-2)
	erf = torch.nn.functional.gelu(a.float() @ b.float().T + bias)
	assert float((outs[0].float() - ref).abs().mean()) < float((outs[0].float() - erf).abs().mean()) or float((ref - erf).abs().max()) < 1e-3


@pytest.mark.parametrize("M", [81920, 4096, 5000, 12345])
def test_skinny_n128_kernel_bit_identical(M):
	"""C[M][128] = epilogue(A[M][512] W[128][512]^T) on the resident-weight streaming kernel (skinny.hip; tile policy 1 picks it for M >= 4096) against the
	128^2 kernel (policy 0): plain bf16 store, GELU with saved pre-activation and dropout (linear1 forward), GELU' with dropout (linear2 input gradient);
	ragged last tile; bit for bit."""
	from novic_amd import ops
	N, K = 128, 512
	a, b = _mk((M, K), 51, 0.5), _mk((N, K), 52, 0.2)
	hpre = _mk((M, N), 53)
	d = ops.Dropout(0.1, seed=79, site=7)

	def run_all():
		outs = []
		o = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
		ops.gemm(a, b, M, N, K, out=o)
		outs.append(o)
		o, o2 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda"), torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
		ops.gemm(a, b, M, N, K, kind=ops.EPI_GELU_BF16, out=o, out2=o2, dropout=d)
		outs += [o, o2]
		o = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
		ops.gemm(a, b, M, N, K, kind=ops.EPI_GELU_BWD_BF16, out=o, resid=hpre, dropout=d)
		outs.append(o)
		o = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
		ops.gemm(a, b, M, N, K, kind=ops.EPI_GELU_BWD_BF16, out=o, resid=hpre)
		outs.append(o)
		return outs

	prev = ops.gemm_tile_policy(0)
	try:
		small = run_all()
		assert ops.gemm_last_tile() == 128
		ops.gemm_tile_policy(1)
		tall = run_all()
		assert ops.gemm_last_tile() == 64
	finally:
		ops.gemm_tile_policy(prev)
	for k, (x, y) in enumerate(zip(small, tall)):
		assert torch.isfinite(x.float()).all()
		if k == 3:
			# GELU' with dropout: a * (1 / (1 - p)) * g'(h) -- g' = cdf + h pdf cancels to ~1e-3 around h = -0.75 and the product then sits on bf16
			# rounding ties; the two kernels land on different sides for a few elements per ten million (25 of 10.5 M): one bf16 ulp, rarely
			ne = x != y
			assert float(ne.float().mean()) <= 1e-5
			assert float(((x.float() - y.float()).abs() / x.float().abs().clamp_min(1e-30))[ne].max() if ne.any() else 0.0) <= 2 ** -7
		else:
			assert torch.equal(x, y)


@pytest.mark.parametrize("M", [81920, 4096, 5000, 12321])
def test_skinny_k128_residual_kernel_bit_identical(M):
	"""C[M][512] = resid + dropout(bf16(A[M][128] W[512][128]^T + bias)) on the resident-weight streaming kernel (skinny.hip) against the 128^2 kernel:
	with and without bias / dropout, in place (out = resid, as the decoder calls it) and out of place, ragged last tile; bit for bit."""
	from novic_amd import ops
	N, K = 512, 128
	a, b = _mk((M, K), 61, 0.5), _mk((N, K), 62, 0.2)
	resid = torch.randn(M, N, device="cuda")
	bias = torch.randn(N, device="cuda")
	d = ops.Dropout(0.1, seed=80, site=8)

	def run_all():
		outs = []
		for bb, dd, inplace in ((None, ops.NO_DROPOUT, False), (bias, d, False), (None, d, True)):
			r = resid.clone()
			o = r if inplace else torch.full((M, N), float("nan"), device="cuda")
			ops.gemm(a, b, M, N, K, kind=ops.EPI_RESID_F32, out=o, resid=r, bias=bb, dropout=dd)
			outs.append(o)
		return outs

	prev = ops.gemm_tile_policy(0)
	try:
		small = run_all()
		assert ops.gemm_last_tile() == 128
		ops.gemm_tile_policy(1)
		tall = run_all()
		assert ops.gemm_last_tile() == 64
	finally:
		ops.gemm_tile_policy(prev)
	for x, y in zip(small, tall):
		assert torch.isfinite(x).all() and torch.equal(x, y)


@pytest.mark.parametrize("lim", [0, 1, 777, 4096, 6000])
def test_row_limit_from_device_memory(lim):
	"""ep.row_limit: a device int clamps the token-row dimension without a host read-back -- M for the row-major-A forms (both tile sizes: rows beyond
	the limit are not written), K for the weight-gradient form (equal to the GEMM over the first `lim` rows; fp32 atomics in another order)."""
	from novic_amd import ops
	M, N, K = 6000, 1024, 256
	a, b = _mk((M, K), 71, 0.5), _mk((N, K), 72, 0.2)
	limit = torch.tensor([lim], dtype=torch.int32, device="cuda")
	prev = ops.gemm_tile_policy(0)
	try:
		for pol in (0, 2):
			ops.gemm_tile_policy(pol)
			full = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
			ops.gemm(a, b, M, N, K, out=full)
			o = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
			ops.gemm(a, b, M, N, K, out=o, row_limit=limit)
			assert torch.equal(o[:lim], full[:lim]) and bool(torch.isnan(o[lim:].float()).all())
		# input-gradient form (B K-strided) on the 128^2 kernel
		ops.gemm_tile_policy(0)
		bt = b.t().contiguous()
		o = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
		ops.gemm(a, bt, M, N, K, b_kstrided=True, out=o, row_limit=limit)
		assert torch.equal(o[:lim], full[:lim]) and bool(torch.isnan(o[lim:].float()).all())
	finally:
		ops.gemm_tile_policy(prev)
	# weight-gradient form: dW[n1][n2] = sum over the first lim rows of dy[row][n1] * x[row][n2]
	dy, x = _mk((M, 384), 73, 0.5), _mk((M, 256), 74, 0.5)
	want = dy[:lim].float().t() @ x[:lim].float()
	for splits in (1, 8, 24):
		got = torch.zeros(384, 256, device="cuda")
		ops.gemm(dy, x, 384, 256, M, a_kstrided=True, b_kstrided=True, kind=ops.EPI_ATOMIC_F32, out=got, split_k=splits, ldc=256, row_limit=limit)
		torch.testing.assert_close(got, want, atol=2e-3 * max(1.0, float(want.abs().max())), rtol=0)


@pytest.mark.parametrize("M,N,K,resid", [(65792, 1024, 1024, True), (32896, 1024, 4096, True), (65792, 3072, 2048, False), (16640, 4096, 2048, False)])
def test_k_split_tail_tiles(M, N, K, resid):
	"""split_tail: the output tiles behind the last whole round of 256 are cut along K and finished by gemm256_tail_kernel (ViT towers: 257 row
	tiles).  Same result as the unsplit kernel up to fp32 summation order (bf16 outputs may flip one rounding: <= 2^-7 relative on a few elements),
	identical from call to call (fixed summation order, no atomics), and rows outside the tail tiles are untouched by the split."""
	from novic_amd import ops
	g = torch.Generator().manual_seed(M + K)
	a = (torch.rand(M, K, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
	b = (torch.rand(N, K, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
	bias = torch.randn(N, generator=g).cuda()
	rs = torch.randn(M, N, generator=g).cuda() if resid else None
	kw = dict(kind=ops.EPI_RESID_F32, resid=rs, bias=bias) if resid else dict(bias=bias, act=ops.ACT_QUICKGELU)
	outs = []
	for split in (False, True, True):
		out = torch.full((M, N), float("nan"), dtype=torch.float32 if resid else torch.bfloat16, device="cuda")
		ops.gemm(a, b, M, N, K, out=out, split_tail=split, **kw)
		assert ops.gemm_last_tile() == 256
		outs.append(out.float())
	assert torch.equal(outs[1], outs[2])
	assert not torch.isnan(outs[1]).any()
	diff = (outs[0] - outs[1]).abs()
	scale = float(outs[0].abs().max())
	assert float(diff.max()) <= 2 ** -7 * scale  # the GEMM term is rounded to bf16 before the residual add as well
	changed = (diff > 0).any(dim=1).nonzero().flatten()
	assert changed.numel() > 0, "the split did not engage"
	assert int(changed.min()) >= (M // 256 - 1) * 256 - 256 * 4  # only rows of the last few row tiles can belong to tail tiles


@pytest.mark.parametrize("M,N,K,policy,split", [(12800, 768, 768, 1, True), (12800, 768, 3072, 1, True), (65792, 1024, 1024, 1, True), (700, 580, 128, 2, False), (1000, 768, 256, 3, False),
                                                 (517, 512, 128, 0, False), (81920, 512, 128, 1, False), (4099, 512, 512, 1, False), (300, 200, 72, 1, False)])
def test_residual_epilogue_in_place(M, N, K, policy, split):
	"""RESID_F32 with c == resid (include/novic_hip.h: allowed -- the towers update their fp32 residual stream in place): every element is read and written by one lane, once,
	on every kernel the call can land on (256 / 192 / 128 tiles, interior and edge paths, K-split tail tiles, the skinny and out-projection kernels): bit-identical to
	writing a second buffer."""
	from novic_amd import ops
	g = torch.Generator().manual_seed(M * 7 + N + K)
	a = (torch.rand(M, K, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
	b = (torch.rand(N, K, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
	bias = torch.randn(N, generator=g).cuda()
	rs = torch.randn(M, N, generator=g).cuda()
	ops.gemm_tile_policy(policy)
	try:
		ref = torch.full((M, N), float("nan"), device="cuda")
		ops.gemm(a, b, M, N, K, kind=ops.EPI_RESID_F32, out=ref, resid=rs, bias=bias, split_tail=split)
		tile = ops.gemm_last_tile()
		x = rs.clone()
		ops.gemm(a, b, M, N, K, kind=ops.EPI_RESID_F32, out=x, resid=x, bias=bias, split_tail=split)
		assert ops.gemm_last_tile() == tile
	finally:
		ops.gemm_tile_policy(1)
	assert not torch.isnan(ref).any()
	assert torch.equal(x, ref), f"tile {tile}: {int((x != ref).sum())} elements differ"


@pytest.mark.parametrize("limit", [None, 120000, 158000, 70])
def test_a_operand_beyond_2_gib_runs_on_the_256_wide_tile(limit):
	"""[160 000 x 512 x 6912] (the multiset step's logits input gradient has 172 032 rows): A is 2.2 GB, more than a buffer descriptor spans, so the 256-wide kernels run it as
	two launches over row ranges (round 4; until then such a call fell back to the 128 x 128 kernel: 896 us of a 17.5 ms step).  Without a row count: bit-identical to the
	128 x 128 kernel.  With a device row count (the K-split tail planned on the device, per range): the rows in front of it agree with it to the split's summation-order
	tolerance, on both sides of the range boundary, and no row behind it is written."""
	from novic_amd import ops
	M, N, K = 160000, 512, 6912
	g = torch.Generator(device="cuda").manual_seed(5)
	a = torch.empty(M, K, dtype=torch.bfloat16, device="cuda")
	for r0 in range(0, M, 20000):  # (generated in slices: no fp32 copy of the whole operand)
		a[r0:r0 + 20000] = (torch.rand(20000, K, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
	b = (torch.rand(N, K, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
	ops.gemm_tile_policy(0)
	try:
		ref = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
		ops.gemm(a, b, M, N, K, out=ref)
		assert ops.gemm_last_tile() == 128
	finally:
		ops.gemm_tile_policy(1)
	out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
	if limit is None:
		ops.gemm(a, b, M, N, K, out=out)
		assert ops.gemm_last_tile() == 256
		assert torch.equal(out, ref)
		return
	lim = torch.tensor([limit], dtype=torch.int32, device="cuda")
	ops.gemm_tile_counts(reset=True)
	ops.gemm(a, b, M, N, K, out=out, row_limit=lim, split_tail=True)
	assert ops.gemm_last_tile() == 256 and ops.gemm_tile_counts()["ksplit_tail_device"] >= 1
	assert torch.isnan(out[limit:].float()).all()  # (whole rows behind the count are never written)
	got, want = out[:limit].float(), ref[:limit].float()
	assert not torch.isnan(got).any()
	assert float((got - want).abs().max()) <= 2 ** -7 * float(want.abs().max())


@pytest.mark.parametrize("M,bias,drop,limit", [(61500, False, 0.1, None), (8192, True, 0.0, None), (20000, True, 0.25, 12345), (4100, False, 0.0, 4097)])
def test_outproj_streaming_kernel_is_bit_identical(M, bias, drop, limit):
	"""[M x 512 x 512] with the fp32 residual epilogue (the decoder's out-proj) runs as four 128-column blocks of the resident-weight streaming
	kernel under the default policy (from 49 152 rows on: the 256 x 256 tile on the 8-phase K loop, round 4): same bits as the 128^2 kernel (policy 0), with bias, dropout and a device-side row limit; rows behind the
	limit untouched."""
	from novic_amd import ops
	g = torch.Generator().manual_seed(M)
	a = (torch.rand(M, 512, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
	w = (torch.rand(512, 512, generator=g) * 0.1).to(torch.bfloat16).cuda()
	rs = torch.randn(M, 512, generator=g).cuda()
	b = torch.randn(512, generator=g).cuda() if bias else None
	lim = None if limit is None else torch.tensor([limit], dtype=torch.int32, device="cuda")
	outs = []
	prev = ops.gemm_tile_policy(-1)
	try:
		for pol in (0, 1):
			ops.gemm_tile_policy(pol)
			out = torch.full((M, 512), -7.0, device="cuda")
			ops.gemm(a, w, M, 512, 512, kind=ops.EPI_RESID_F32, out=out, resid=rs, bias=b, dropout=ops.Dropout(drop, 99, 3), row_limit=lim)
			outs.append(out)
	finally:
		ops.gemm_tile_policy(prev)
	assert torch.equal(outs[0], outs[1])
	n = M if limit is None else limit
	assert bool((outs[1][n:] == -7.0).all()) and not bool((outs[1][:n] == -7.0).all())


@pytest.mark.parametrize("wide", [1, 0], ids=["two_256_column_blocks", "four_128_column_blocks"])
@pytest.mark.parametrize("M,bias,limit", [(61500, False, None), (9000, True, 8001), (4099, False, None)])
def test_outproj_dgrad_streaming_kernel_is_bit_identical(M, bias, limit, wide):
	"""[M x 512 x 512] with the bf16 store (the out-proj input gradient) on the streaming kernel -- four 128-column blocks per row stream (default) or two 256-column
	blocks: same bits as the 128^2 kernel."""
	from novic_amd import ops
	prev_wide = ops.skinny_wide_policy(wide)
	try:
		_outproj_dgrad(M, bias, limit)
	finally:
		ops.skinny_wide_policy(prev_wide)


def _outproj_dgrad(M, bias, limit):
	from novic_amd import ops
	g = torch.Generator().manual_seed(M + 1)
	a = (torch.rand(M, 512, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
	w = (torch.rand(512, 512, generator=g) * 0.1).to(torch.bfloat16).cuda()
	b = torch.randn(512, generator=g).cuda() if bias else None
	lim = None if limit is None else torch.tensor([limit], dtype=torch.int32, device="cuda")
	outs = []
	prev = ops.gemm_tile_policy(-1)
	try:
		for pol in (0, 1):
			ops.gemm_tile_policy(pol)
			out = torch.full((M, 512), -7.0, dtype=torch.bfloat16, device="cuda")
			ops.gemm(a, w, M, 512, 512, out=out, bias=b, row_limit=lim)
			outs.append(out)
	finally:
		ops.gemm_tile_policy(prev)
	assert torch.equal(outs[0], outs[1])
	n = M if limit is None else limit
	assert bool((outs[1][n:] == -7.0).all()) and not bool((outs[1][:n] == -7.0).all())


@pytest.mark.parametrize("M,N,K,lda,ldb,limit,splits", [
	(1536, 512, 20000, 1536, 512, None, 0),      # in-proj dW, one round of 12 tiles x 21 parts
	(1536, 512, 20000, 1536, 512, 13333, 0),     # token count clamped by a device int (packed rows)
	(6912, 512, 9000, 6912, 512, 7001, 0),       # logits dW: 54 tiles x 4 parts
	(264, 520, 5000, 272, 528, None, 5),         # ragged edges in both output dimensions, padded leading dimensions, 2 x 3 tiles x 5 parts
	(512, 512, 130, 512, 512, None, 64),         # more parts asked for than K-tiles exist (3): clamped
	(256, 256, 64, 256, 256, 0, 0),              # row limit 0: nothing to add
	(128, 512, 20000, 128, 512, 17001, 0),       # linear1 dW: 128 x 256 tiles (the narrow dimension as tile rows)
	(512, 128, 20000, 512, 128, None, 0),        # linear2 dW: computed as its transpose, written back transposed
	(520, 72, 3000, 528, 72, None, 7),           # narrow N with ragged edges, transposed path, padded ld
	(64, 264, 2000, 64, 272, None, 3),           # narrow M below the tile height
])
def test_wgrad256_matches_fp32_matmul(M, N, K, lda, ldb, limit, splits):
	from novic_amd import ops
	"""novic_wgrad_bf16 (wgrad.hip): dW += alpha * dY^T X with both operands row-major over the token dimension.  bf16 operands, fp32 accumulation in another
	order than torch's: 2e-3 relative to the largest element of the product; accumulation into dW, alpha, and run-to-run determinism (fixed-order partial sums)."""
	g = torch.Generator().manual_seed(M + N + K)
	dy = (torch.randn(K, lda, generator=g) * 0.5).to(torch.bfloat16).cuda()
	x = (torch.randn(K, ldb, generator=g) * 0.5).to(torch.bfloat16).cuda()
	base = torch.randn(M, N, generator=g).cuda()
	lim = None if limit is None else torch.tensor([limit], dtype=torch.int32, device="cuda")
	Ke = K if limit is None else min(K, limit)
	want = base.double() + 0.25 * (dy[:Ke, :M].double().T @ x[:Ke, :N].double())
	outs = []
	for _ in range(2):
		out = base.clone()
		ops.wgrad(dy, x, M, N, K, out, alpha=0.25, row_limit=lim, splits=splits)
		outs.append(out)
	torch.cuda.synchronize()
	scale = float((want - base.double()).abs().max()) + 1e-6
	assert float((outs[0].double() - want).abs().max()) <= 2e-3 * scale + 1e-5
	assert torch.equal(outs[0], outs[1])
	if limit == 0:
		assert torch.equal(outs[0], base)


def test_wgrad256_operand_beyond_2_gib():
	"""The multiset step's logits gradient (configs[4]: 172 032 output rows x 6912 bf16 = 2.4 GB): the buffer descriptors are based at each part's first row, so
	only a PART's rows must fit 32-bit offsets.  Compared with torch's matmul of the same operands on the device (a host fp64 product of this size takes minutes)."""
	from novic_amd import ops
	K, M, N = 172032, 6912, 512
	g = torch.Generator(device="cuda").manual_seed(5)
	dy = (torch.randn(K, M, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
	x = (torch.randn(K, N, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
	assert dy.numel() * 2 > 2 ** 31
	out = torch.zeros(M, N, device="cuda")
	ops.wgrad(dy, x, M, N, K, out)
	want = torch.zeros(M, N, device="cuda")
	for k0 in range(0, K, 21504):  # fp32 matmul in slices: the last rows (beyond 2 GiB from the base) carry the same weight as the first
		want += dy[k0:k0 + 21504].float().T @ x[k0:k0 + 21504].float()
	torch.cuda.synchronize()
	assert float((out - want).abs().max()) <= 2e-3 * float(want.abs().max())
	tail = torch.zeros(M, N, device="cuda")
	lim = torch.tensor([K - 1000], dtype=torch.int32, device="cuda")
	ops.wgrad(dy, x, M, N, K, tail, row_limit=lim)
	torch.cuda.synchronize()
	diff = dy[K - 1000:].float().T @ x[K - 1000:].float()
	assert float((out - tail - diff).abs().max()) <= 2e-3 * float(want.abs().max())   # the clamp lands inside the last part, beyond 2 GiB from the base


def test_wgrad256_agrees_with_the_split_k_atomic_kernel():
	"""The same weight gradient on the 128^2 split-K kernel (fp32 atomics, gemm.hip) and on the 256-wide kernel: equal up to fp32 summation order."""
	from novic_amd import ops
	g = torch.Generator().manual_seed(12)
	K, M, N = 30000, 1536, 512
	dy = (torch.randn(K, M, generator=g) * 0.3).to(torch.bfloat16).cuda()
	x = (torch.randn(K, N, generator=g) * 0.3).to(torch.bfloat16).cuda()
	a = torch.zeros(M, N, device="cuda")
	b = torch.zeros(M, N, device="cuda")
	ops.gemm(dy, x, M, N, K, a_kstrided=True, b_kstrided=True, kind=ops.EPI_ATOMIC_F32, out=a, split_k=10, ldc=N)
	ops.wgrad(dy, x, M, N, K, b)
	torch.cuda.synchronize()
	assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max())


@pytest.mark.parametrize("M,N,K,limit,splits", [
	(1536, 512, 61519, None, 0),     # in-projection dW at the bench's packed row count: 12 tiles x 21 parts of 45-46 K-tiles, ragged last K-tile
	(512, 512, 61519, 60001, 0),     # out-projection dW, clamped by a device int inside a part
	(6912, 512, 36943, None, 0),     # logits dW: 54 tiles x 4 parts of 144-145 K-tiles
	(1536, 512, 20000, None, 21),    # 313 K-tiles over 21 parts: 15 K-tiles each, the last part 13
	(512, 512, 1000, None, 16),      # 16 K-tiles over 16 parts: ONE K-tile per part (prologue + tail only)
	(512, 512, 2000, None, 16),      # two K-tiles per part (32 K-tiles), the last part ragged
	(512, 512, 3000, None, 16),      # three
	(512, 512, 4000, None, 16),      # four: one steady trip is impossible (kt + 3 < ke fails), all through the tail path
	(512, 512, 5100, None, 16),      # five: one steady trip + a one-tile tail
	(128, 512, 20000, 17001, 0),     # 128 x 256 tiles (A pieces of four k rows, one per wave and half-tile)
	(512, 128, 20000, None, 0),      # ... the transposed narrow product
	(128, 512, 700, None, 8),        # narrow tiles, 11 K-tiles over 8 parts: 2 / 1 K-tiles per part
	(264, 520, 5000, None, 5),       # ragged output edges
	(512, 512, 18476, None, 2),      # 289 K-tiles over 2 parts: 145 (odd: the tail runs three K-tiles) and 144 (even: two), > 70 steady trips each, ragged last K-tile (44 rows)
	(512, 512, 18575, None, 2),      # 291 K-tiles: 146 (even) and 145 (odd), the ragged K-tile (15 rows, as the logits shape's) closing the ODD part
	(768, 512, 37100, 36943, 4),     # the logits shape's K geometry (145 / 145 / 145 / 143 K-tiles, 15 rows in the last) through the DEVICE row limit
])
def test_wgrad_8phase_kernel_is_bit_identical_to_the_one_barrier_kernel(M, N, K, limit, splits):
	"""wgrad256p_kernel (8-phase schedule: staggered wave groups, half-tile LDS-DMA six half-tiles ahead behind counted vmcnt waits, raw barriers) reads the same LDS
	image through the same fragment addresses and issues the same MFMAs per accumulator in the same order as wgrad256_kernel, so the partial sums -- and after the
	fixed-order reduction the gradients -- must be BIT-identical.  A read that overtakes its LDS-DMA, or a DMA that overwrites fragments still being read, shows as a
	difference in some runs: repeated, because such races come and go with timing (cdna_hip_programming.md section 5, 'Read a staged buffer one phase AFTER ...')."""
	from novic_amd import ops
	g = torch.Generator().manual_seed(M * 7 + N + K)
	dy = (torch.randn(K, M, generator=g) * 0.5).to(torch.bfloat16).cuda()
	x = (torch.randn(K, N, generator=g) * 0.5).to(torch.bfloat16).cuda()
	lim = None if limit is None else torch.tensor([limit], dtype=torch.int32, device="cuda")
	prev = ops.wgrad_policy(0)
	try:
		ref = torch.zeros(M, N, device="cuda")
		ops.wgrad(dy, x, M, N, K, ref, alpha=1.0, row_limit=lim, splits=splits)
		ops.wgrad_policy(1)
		outs = []
		for rep in range(12):
			out = torch.zeros(M, N, device="cuda")
			ops.wgrad(dy, x, M, N, K, out, alpha=1.0, row_limit=lim, splits=splits)
			outs.append(out)
		torch.cuda.synchronize()
	finally:
		ops.wgrad_policy(prev)
	assert float(ref.abs().max()) > 0
	for rep, out in enumerate(outs):
		if not torch.equal(out, ref):  # (round 4 saw ONE such mismatch in eight runs of the suite -- DESIGN.md section 4 "One unexplained event": say WHERE it is, should it come back)
			d = (out - ref)
			nz = d.nonzero()
			raise AssertionError(f"repetition {rep}: {nz.shape[0]} elements differ, rows {int(nz[:, 0].min())}..{int(nz[:, 0].max())}, columns {int(nz[:, 1].min())}..{int(nz[:, 1].max())}, "
			                     f"distinct rows {nz[:, 0].unique().numel()}, distinct columns {nz[:, 1].unique().numel()}, max |d| {float(d.abs().max()):.4g}, "
			                     f"mean |d| over them {float(d[d != 0].abs().mean()):.4g}; other repetitions that differ: {[r for r, o in enumerate(outs) if not torch.equal(o, ref)]}")


@pytest.mark.parametrize("K,limit", [(20000, None), (9000, 7001), (300, None), (5000, 0)])
def test_wgrad_pair_matches_two_calls(K, limit):
	"""novic_wgrad2_bf16: a layer's in-projection [1536 x 512] and out-projection [512 x 512] gradients over the same token rows in one launch pair -- against fp64 matmuls
	and against the two single calls (equal up to the fp32 summation order of a different part count); deterministic; accumulates into both outputs."""
	from novic_amd import ops
	g = torch.Generator().manual_seed(K)
	dy1 = (torch.randn(K, 1536, generator=g) * 0.3).to(torch.bfloat16).cuda()
	x1 = (torch.randn(K, 512, generator=g) * 0.3).to(torch.bfloat16).cuda()
	dy2 = (torch.randn(K, 512, generator=g) * 0.3).to(torch.bfloat16).cuda()
	x2 = (torch.randn(K, 512, generator=g) * 0.3).to(torch.bfloat16).cuda()
	b1, b2 = torch.randn(1536, 512, generator=g).cuda(), torch.randn(512, 512, generator=g).cuda()
	lim = None if limit is None else torch.tensor([limit], dtype=torch.int32, device="cuda")
	Ke = K if limit is None else min(K, limit)
	outs = []
	for _ in range(2):
		o1, o2 = b1.clone(), b2.clone()
		ops.wgrad2(dy1, x1, 1536, 512, o1, dy2, x2, 512, 512, o2, K, alpha=0.5, row_limit=lim)
		outs.append((o1, o2))
	torch.cuda.synchronize()
	assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
	w1 = b1.double() + 0.5 * (dy1[:Ke].double().T @ x1[:Ke].double())
	w2 = b2.double() + 0.5 * (dy2[:Ke].double().T @ x2[:Ke].double())
	for got, want, base in ((outs[0][0], w1, b1), (outs[0][1], w2, b2)):
		scale = float((want - base.double()).abs().max()) + 1e-6
		assert float((got.double() - want).abs().max()) <= 2e-3 * scale + 1e-5
	if limit == 0:
		assert torch.equal(outs[0][0], b1) and torch.equal(outs[0][1], b2)
	if K >= 16384:  # the single-problem entry takes these shapes: same sums in another order
		s1, s2 = b1.clone(), b2.clone()
		ops.wgrad(dy1, x1, 1536, 512, K, s1, alpha=0.5, row_limit=lim)
		ops.wgrad(dy2, x2, 512, 512, K, s2, alpha=0.5, row_limit=lim)
		assert float((s1 - outs[0][0]).abs().max()) <= 1e-4 * float(s1.abs().max()) and float((s2 - outs[0][1]).abs().max()) <= 1e-4 * float(s2.abs().max())


@pytest.mark.parametrize("K,limit", [(20000, None), (9000, 7001), (200, None)])
def test_wgrad_pair_of_narrow_outputs(K, limit):
	"""The feed-forward pair: linear2's gradient [512 x 128] (computed as its transpose, written back transposed) and linear1's [128 x 512] in one launch pair."""
	from novic_amd import ops
	g = torch.Generator().manual_seed(K + 5)
	gb = (torch.randn(K, 512, generator=g) * 0.3).to(torch.bfloat16).cuda()
	hact = (torch.randn(K, 128, generator=g) * 0.3).to(torch.bfloat16).cuda()
	dh = (torch.randn(K, 128, generator=g) * 0.3).to(torch.bfloat16).cuda()
	ln2 = (torch.randn(K, 512, generator=g) * 0.3).to(torch.bfloat16).cuda()
	b2, b1 = torch.randn(512, 128, generator=g).cuda(), torch.randn(128, 512, generator=g).cuda()
	lim = None if limit is None else torch.tensor([limit], dtype=torch.int32, device="cuda")
	Ke = K if limit is None else min(K, limit)
	outs = []
	for _ in range(2):
		o2, o1 = b2.clone(), b1.clone()
		ops.wgrad2(gb, hact, 512, 128, o2, dh, ln2, 128, 512, o1, K, row_limit=lim)
		outs.append((o2, o1))
	torch.cuda.synchronize()
	assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
	w2 = b2.double() + gb[:Ke].double().T @ hact[:Ke].double()
	w1 = b1.double() + dh[:Ke].double().T @ ln2[:Ke].double()
	for got, want, base in ((outs[0][0], w2, b2), (outs[0][1], w1, b1)):
		scale = float((want - base.double()).abs().max()) + 1e-6
		assert float((got.double() - want).abs().max()) <= 2e-3 * scale + 1e-5


@pytest.mark.parametrize("kind,K,limit,n", [("wide", 61553, None, 4), ("wide", 20000, 12345, 4), ("wide", 9000, 0, 3), ("narrow", 61553, 50001, 4), ("narrow", 700, None, 3), ("wide", 300, None, 1)])
def test_weight_gradients_of_two_layers_in_one_launch(kind, K, limit, n):
	"""novic_wgradn_bf16 (round 6): up to four weight gradients over the same token rows in one launch pair -- the attention pairs of two layers ([1536 x 512] + [512 x 512],
	twice: 32 tiles x 8 parts) or their narrow feed-forward pairs ([512 x 128] as its transpose + [128 x 512], twice: 8 tiles x 32 parts) -- against fp64 matmuls and
	against one novic_wgrad2_bf16 call per layer (the same sums in another fp32 order); deterministic; accumulates; a device row count clamps K."""
	from novic_amd import ops
	g = torch.Generator().manual_seed(K + n)
	shapes = [(1536, 512), (512, 512), (1536, 512), (512, 512)] if kind == "wide" else [(512, 128), (128, 512), (512, 128), (128, 512)]
	shapes = shapes[:n]
	dys = [(torch.randn(K, M, generator=g) * 0.3).to(torch.bfloat16).cuda() for M, _ in shapes]
	xs = [(torch.randn(K, N, generator=g) * 0.3).to(torch.bfloat16).cuda() for _, N in shapes]
	base = [torch.randn(M, N, generator=g).cuda() for M, N in shapes]
	lim = None if limit is None else torch.tensor([limit], dtype=torch.int32, device="cuda")
	Ke = K if limit is None else min(K, limit)
	runs = []
	for _ in range(2):
		outs = [b.clone() for b in base]
		ops.wgradn([(dy, x, M, N, o) for dy, x, (M, N), o in zip(dys, xs, shapes, outs)], K, alpha=0.5, row_limit=lim)
		runs.append(outs)
	torch.cuda.synchronize()
	for a, b in zip(*runs):
		assert torch.equal(a, b)
	for got, dy, x, b in zip(runs[0], dys, xs, base):
		want = b.double() + 0.5 * (dy[:Ke].double().T @ x[:Ke].double())
		scale = float((want - b.double()).abs().max()) + 1e-6
		assert float((got.double() - want).abs().max()) <= 2e-3 * scale + 1e-5
		if limit == 0:
			assert torch.equal(got, b)
	if n == 4:  # one pair launch per layer: the same sums in another order
		pairs = [b.clone() for b in base]
		for i in (0, 2):
			ops.wgrad2(dys[i], xs[i], shapes[i][0], shapes[i][1], pairs[i], dys[i + 1], xs[i + 1], shapes[i + 1][0], shapes[i + 1][1], pairs[i + 1], K, alpha=0.5, row_limit=lim)
		for a, b in zip(runs[0], pairs):
			assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max())


@pytest.mark.parametrize("M,limit", [(57344, 36943), (57344, 57344), (57344, 65536), (57344, 66000 - 256 * 30), (20480, 17000), (57344, 300), (57344, 0)])
def test_device_row_count_with_k_split_tail(M, limit):
	"""The logits input gradient [rows x 512 x 6912] with a DEVICE row count and scratch: the 256-wide kernel plans the K-split of its tail tiles on the device (290 tiles
	at 36 943 rows: 34 tail tiles x 7 parts; 448 at all rows: no split; one round or less: none).  Against torch's fp32 matmul of the same bf16 operands; rows at or
	beyond the limit untouched; run-to-run deterministic."""
	from novic_amd import ops
	N, K = 512, 6912
	g = torch.Generator(device="cuda").manual_seed(limit + 1)
	a = (torch.randn(M, K, generator=g, device="cuda") * 0.1).to(torch.bfloat16)
	w = (torch.randn(N, K, generator=g, device="cuda") * 0.1).to(torch.bfloat16)
	lim = torch.tensor([limit], dtype=torch.int32, device="cuda")
	rows = min(M, max(limit, 0))
	outs = []
	ops.gemm_tile_counts(reset=True)
	for _ in range(2):
		out = torch.full((M, N), -7.0, dtype=torch.bfloat16, device="cuda")
		ops.gemm(a, w, M, N, K, out=out, row_limit=lim, split_tail=True)
		outs.append(out)
	torch.cuda.synchronize()
	assert ops.gemm_last_tile() == 256  # (also the 160-tile case: on the 8-phase K loop the 256-wide tile is chosen from 144 tiles on)
	# the launch must be the one that plans a K-split tail on the device wherever the allocated size has more than a round of tiles (a more general tile rule in
	# front of this one took these shapes without the tail for a while in round 3: correct results, a whole extra round of tiles)
	assert ops.gemm_tile_counts()["ksplit_tail_device"] == (2 if M >= 32768 else 0)
	assert torch.equal(outs[0], outs[1])
	assert bool((outs[0][rows:] == -7.0).all())
	if rows:
		want = a[:rows].float() @ w.float().T
		err = (outs[0][:rows].float() - want).abs().max()
		assert float(err) <= 2e-2 * float(want.abs().max()), float(err)


@pytest.mark.parametrize("M,N,K,act,bias,drop", [(300, 128, 512, "relu", True, 0.0), (1000, 96, 64, "tanh", True, 0.1), (4096, 128, 512, "gelu", True, 0.0), (8192, 128, 512, "relu", False, 0.1),
                                                 (70000, 128, 512, "tanh", False, 0.0)])
def test_activation_epilogues_with_bias(M, N, K, act, bias, drop):
	"""ABI 10: the GELU_BF16 / GELU_BWD_BF16 epilogue pair with the reference's other activations (relu, tanh: utils.py:100-105) and a bias in front of the activation
	(linear1 of a layer_bias decoder, the prefix MLP's hidden layer), against torch on the same bf16 operands: c2 = bf16(acc + bias), c = dropout(bf16(act(c2))),
	backward c = bf16(bf16(acc) * mask * act'(pre)).  Shapes the skinny / 256-wide kernels would take for the erf GELU are declined by them and run on the 128 x 128 kernel."""
	from novic_amd import ops
	g = torch.Generator().manual_seed(M + N + K)
	a = torch.randn(M, K, generator=g).bfloat16()
	w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
	b = torch.randn(N, generator=g) * 0.5 if bias else None
	code = ops.ACT_BY_NAME[act]
	fn = {"gelu": torch.nn.functional.gelu, "tanh": torch.tanh, "relu": torch.relu}[act]
	hact = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
	hpre = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
	dr = ops.Dropout(drop, 1234, 5)
	ops.gemm(a.cuda(), w.cuda(), M, N, K, kind=ops.EPI_GELU_BF16, act=code, bias=b.cuda() if bias else None, out=hact, out2=hpre, dropout=dr)
	assert ops.gemm_last_tile() == 128
	acc = a.float() @ w.float().T + (b if bias else 0)
	pre = acc.bfloat16()
	assert float((hpre.float().cpu() - pre.float()).abs().max()) <= 2 ** -7 * float(pre.float().abs().max())  # (fp32 accumulation order: one bf16 ulp at rounding ties)
	want = fn(hpre.float().cpu()).bfloat16().float()  # from the kernel's own pre-activation: exact up to the activation's fp32 arithmetic
	got = hact.float().cpu()
	kept = got != 0 if drop > 0 else torch.ones_like(got, dtype=torch.bool)
	sc = 1 / (1 - drop)
	assert float((got - want * sc)[kept].abs().max()) <= 2 ** -7 * sc * max(1.0, float(want.abs().max()))
	if drop > 0:
		frac = float(((got == 0) & (want != 0)).float().sum() / (want != 0).float().sum())
		assert abs(frac - drop) < 0.02
	# backward: dY [M x N2] against W2^T -> [M x N], times the mask and act'(pre)
	N2 = 64
	dy = torch.randn(M, N2, generator=g).bfloat16()
	w2t = (torch.randn(N, N2, generator=g) / N2 ** 0.5).bfloat16()  # linear2.weight^T: [N][N2], K-contiguous
	dh = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
	ops.gemm(dy.cuda(), w2t.cuda(), M, N, N2, kind=ops.EPI_GELU_BWD_BF16, act=code, resid=hpre, out=dh, dropout=dr)
	x = hpre.float().cpu().requires_grad_(True)
	y = fn(x)
	if act == "tanh":
		y = y.bfloat16().float() + (y - y.detach())  # (torch's tanh_backward uses the bf16 result under autocast: 1 - y_bf16^2)
		d = 1 - fn(x.detach()).bfloat16().float() ** 2
	else:
		y.backward(torch.ones_like(y))
		d = x.grad
	mask = (got != 0).float() * sc if drop > 0 else torch.ones_like(got)
	if drop > 0:  # where act(pre) is exactly zero the mask cannot be read off the forward output: compare the kept elements only
		sel = want != 0
	else:
		sel = torch.ones_like(got, dtype=torch.bool)
	wantb = (dy.float() @ w2t.float().T).bfloat16().float() * mask * d
	assert float((dh.float().cpu() - wantb)[sel].abs().max()) <= 2 ** -6 * max(1.0, float(wantb.abs().max()))
