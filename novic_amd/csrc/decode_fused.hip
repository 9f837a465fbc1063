// Fused per-layer kernels of a KV-cached decode step (one new position per beam, M = beams x samples rows of a few hundred to a few thousand).
//
// At these sizes the unfused layer -- LayerNorm, QKV GEMM, attention, out-proj GEMM (+residual), LayerNorm, linear1 GEMM (+GELU), linear2 GEMM
// (+residual) -- is seven launches of 2..96 workgroups of the 128x128-tile GEMM.  What bounds such a launch is how fast ONE CU can pull bytes from
// beyond its L2 (the layer's 2.2 MB of weights are re-read every step and do not stay in a 4 MiB L2 across layers): measured ~12 B/clk/CU, so a
// workgroup that streams 128..768 KB of weights takes 9..28 us whatever the MFMA work (a fused out-proj + LayerNorm + MLP kernel with 16-row
// workgroups was tried: 768 KB per workgroup, 29 us).  The kernels here therefore cut the problem the other way: many small workgroups (16 rows
// x 64 columns, one 16 x 16 MFMA tile per wave, 16..64 KB of weights each) so that every CU streams a little, with the LayerNorm fused in as a
// prologue (recomputed per column block: 16 rows x 2 KB from L2) and the residual / GELU as epilogues.  A layer is five launches:
//
//   novic_decode_ln_gemm     qkv  = LayerNorm(x; norm1) Wqkv^T
//   novic_decode_attn        (decode.hip)
//   novic_decode_gemm_resid  xm   = x + att Wo^T
//   novic_decode_ln_gemm     h    = GELU(LayerNorm(xm; norm2) W1^T)
//   novic_decode_gemm_resid  x    = xm + h W2^T
//
// Weights (and the A rows of the residual GEMMs) are never staged in LDS: every wave reads its MFMA fragments straight from memory in fragment
// layout (lane (r, q) <- W[n0 + r][32 ks + 8 q .. +7], 16 bytes) -- nothing is shared between the waves of a workgroup.
// Arithmetic and rounding points are those of the unfused kernels (same MFMA, K accumulated in the same order, LayerNorm with the same lane
// layout and shuffles, bf16 rounding of GEMM outputs / GELU as in the gemm.hip epilogues): results are bit-identical to the unfused path
// (tests/test_gpu_decode_fused.py).  reference: nn.TransformerEncoderLayer (norm_first) as called from embedding_decoder.py:714.
#include "common.hpp"
#include "novic_hip.h"

namespace {

// ---- LDS images of a 16-row bf16 panel [16][K]: 16-byte chunks XOR-swizzled by the row so that the 16 lanes of a fragment read spread over banks ----
__device__ __forceinline__ int panel_off(int row, int chunk, int k_elems) {
	const int cpr = k_elems >> 3;  // 16-byte chunks per row (a power of two for the supported sizes); the XOR must stay inside the row
	return row * (k_elems * 2) + ((chunk ^ (row & 7 & (cpr - 1))) << 4);
}

__device__ __forceinline__ bf16x8 panel_frag(const char* lds, int ks, int lane, int k_elems) {
	return *reinterpret_cast<const bf16x8*>(lds + panel_off(lane & 15, ks * 4 + (lane >> 4), k_elems));
}

// B fragment of W [N][ldw] (K-contiguous) straight from memory: rows n0 + (lane & 15), k = 32 ks + 8 (lane >> 4) .. +7; rows >= N read as zero.
// SRD buffer loads: the range check supplies the zeros, so the (many) loads of a phase stay branch-free and all in flight together -- a
// predicated `n < N ? *p : 0` wraps every load in a branch and makes the compiler drain vmcnt between them.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t weight_rsrc(const bf16* W, int N, int ldw) {
	return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(W), 0, N * ldw * 2, 0x00020000);
}
__device__ __forceinline__ bf16x8 weight_frag(__amdgpu_buffer_rsrc_t W, int ldw, int N, int n0, int ks, int lane) {
	const int n = n0 + (lane & 15);
	const unsigned off = n < N ? (unsigned)((n * ldw + ks * 32 + (lane >> 4) * 8) * 2) : 0xFFFFFFF0u;
	const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(W, off, 0, 0);
	return __builtin_bit_cast(bf16x8, t);
}

// A wave's 16 x K weight tile through LDS instead: LDS-DMA (buffer_load_dwordx4 ... lds) moves whole rows -- one instruction = 1 KiB = 8 full
// 128-byte lines (a row at K = 512, 4 rows at K = 128) -- where the fragment-shaped register loads above touch 16 lines for the same 1 KiB
// (64 B of each), and the texture-address unit's rate per line is what bounds these kernels (~12 B/clk/CU measured with fragment loads).
// The image is the swizzled panel layout of panel_off(): lane L of an instruction lands in slot L of the instruction's 1 KiB, so it fetches
// the global chunk that belongs there (slot ^ (row & 7), the XOR being its own inverse).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
// first / step: instruction i = first, first + step, ... (a wave staging its own tile: 0, 1; four waves sharing one tile: w, 4)
template <int NKS>
__device__ __forceinline__ void stage_tile_dma(__amdgpu_buffer_rsrc_t W, int ldw, int N, int n0, char* tile, int lane, int first = 0, int step = 1) {
	constexpr int K = NKS * 32, CPR = K / 8;          // 16-byte chunks per row
	constexpr int RPI = 64 / CPR > 0 ? 64 / CPR : 1;  // rows per instruction (CPR <= 64)
	static_assert(CPR <= 64, "row longer than one LDS-DMA instruction");
#pragma unroll
	for (int i = first; i < 16 / RPI; i += step) {
		const int row = i * RPI + lane / CPR, slot = lane % CPR;
		const int n = n0 + row;
		const unsigned off = n < N ? (unsigned)((n * ldw + ((slot ^ (row & 7 & (CPR - 1))) << 3)) * 2) : 0xFFFFFFF0u;
		__builtin_amdgcn_raw_ptr_buffer_load_lds(W, (lds_ptr_t)(tile + i * 1024), 16, off, 0, 0, 0);
	}
}

// LayerNorm of one row held by one wave (element 256 c + 4 lane + i), identical arithmetic to layernorm_fwd_kernel (norm.hip); writes bf16 into
// the swizzled LDS panel.  src may be global or LDS memory.
template <int NC>
__device__ __forceinline__ void ln_row_to_panel(const float* src, const float* gamma, char* panel, int row, int E, int lane, float eps, bool valid) {
	float v[NC][4];
#pragma unroll
	for (int c = 0; c < NC; ++c) {
		const int e = c * 256 + lane * 4;
		if (valid && e < E) {
			const f32x4 t = *reinterpret_cast<const f32x4*>(src + e);
			v[c][0] = t[0]; v[c][1] = t[1]; v[c][2] = t[2]; v[c][3] = t[3];
		} else {
			v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f;
		}
	}
	float mean, rstd;
	ln_row_stats<NC>(v, E, lane, eps, mean, rstd);
#pragma unroll
	for (int c = 0; c < NC; ++c) {
		const int e = c * 256 + lane * 4;
		if (e < E) {
			const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + e);
			bf16x4 o = {(bf16)ln_apply(v[c][0], mean, rstd, gm[0]), (bf16)ln_apply(v[c][1], mean, rstd, gm[1]), (bf16)ln_apply(v[c][2], mean, rstd, gm[2]),
			            (bf16)ln_apply(v[c][3], mean, rstd, gm[3])};
			if (!valid) o = (bf16x4){0, 0, 0, 0};
			*reinterpret_cast<bf16x4*>(panel + panel_off(row, e >> 3, E) + ((e & 7) << 1)) = o;
		}
	}
}

template <int NKS>
__device__ __forceinline__ void load_tile(bf16x8 (&wf)[NKS], __amdgpu_buffer_rsrc_t W, int ldw, int N, int n0, int lane) {
#pragma unroll
	for (int ks = 0; ks < NKS; ++ks) wf[ks] = weight_frag(W, ldw, N, n0, ks, lane);
}
template <int NKS>
__device__ __forceinline__ f32x4 mma_tile(const bf16x8 (&wf)[NKS], const bf16x8 (&af)[NKS]) {
	f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
	for (int ks = 0; ks < NKS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], af[ks], acc, 0, 0, 0);
	return acc;
}

// ---------------------------------------------------------------------------------------------------------
// y[M][ldy] (bf16) = act(LayerNorm(x[M][E]; gamma) W[N][E]^T),  act = identity | GELU (with the bf16 rounding points of the gemm.hip epilogues).
// grid (ceil(M/16), ceil(N/64)); the workgroup's 4 waves normalise 4 rows each, then own one 16-column tile each (four tiles per wave = fewer
// LayerNorm recomputations was slower at every size tried: fewer, fatter workgroups).
// ---------------------------------------------------------------------------------------------------------
// NW = waves per workgroup (4: 64 output columns; 8: 128 -- round 5: the QKV projection at 256 rows is 16 x 24 = 384 workgroups of four waves, a round and a half of the chip
// for a launch that is one dependent chain per workgroup; 192 workgroups of eight waves are one round: greedy decoding 119.6 k -> 123.7 k labels/s at 256 rows, 202 k -> 212 k at
// 512, two builds in alternating processes, tools/decode_wide_ab.sh.  Same tiles, same arithmetic: bit-identical.)
#ifndef DECODE_LN_WIDE
#define DECODE_LN_WIDE 1  // (0: two-build A/B of the eight-wave form, tools/lib_ab.sh style)
#endif
template <int NKS, bool GELU, int NW = 4>
__global__ __launch_bounds__(NW * 64) void decode_ln_gemm_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const bf16* __restrict__ W, bf16* __restrict__ y,
                                                                 int M, int N, int ldy, float eps) {
	constexpr int E = NKS * 32, NC = (E + 255) / 256, TILE = 16 * E * 2;
	extern __shared__ __attribute__((aligned(16))) char smem[];  // [activation panel | NW weight tiles]
	char* panel = smem;
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int fr = lane & 15, fq = lane >> 4;
	const int m0 = blockIdx.x * 16, n0 = (blockIdx.y * NW + w) * 16;
	char* wtile = smem + TILE + w * TILE;
	stage_tile_dma<NKS>(weight_rsrc(W, N, E), E, N, n0, wtile, lane);  // the weights do not depend on the activations: they fly under the LayerNorm
#pragma unroll
	for (int rr = 0; rr < 16 / NW; ++rr) {  // unrolled: the rows' loads go out together (one at a time each row paid its own global round trip)
		const int row = w * (16 / NW) + rr, m = m0 + row;
		ln_row_to_panel<NC>(x + (size_t)(m < M ? m : 0) * E, gamma, panel, row, E, lane, eps, m < M);
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's own weight tile has landed (nobody else reads it)
	__syncthreads();                                   // everybody's rows of the activation panel are written
	f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
	for (int ks = 0; ks < NKS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(panel_frag(wtile, ks, lane, E), panel_frag(panel, ks, lane, E), acc, 0, 0, 0);
	const int m = m0 + fr, n = n0 + fq * 4;
	if (m < M && n < N) {
		bf16x4 o;
#pragma unroll
		for (int i = 0; i < 4; ++i) o[i] = GELU ? (bf16)gelu_erf(bf16_round(acc[i])) : (bf16)acc[i];
		*reinterpret_cast<bf16x4*>(y + (size_t)m * ldy + n) = o;
	}
}

// (Several 16-row tiles per workgroup -- a wave multiplying its weight tile against 2 or 4 activation tiles, picked to minimise rounds x bytes
// pulled per CU -- was measured again with LDS-DMA staging: LayerNorm + QKV at 256 rows 8.1 -> 11.3 us with two tiles, the plain QKV GEMM at 1024
// rows 10.3 -> 9.3 us with four, out-proj / linear2 unchanged or slower.  The launches are bound by the dependent chain inside a workgroup, not
// by the bytes a CU pulls: one tile per workgroup stays.)
// ---------------------------------------------------------------------------------------------------------
// Small-tile GEMM a[M][K] W[N][K]^T with the decode epilogues: residual add in fp32 (out-proj / linear2; out may alias resid), bf16 store, bf16 GELU
// (for row counts where a separate LayerNorm launch beats recomputing it in every column block).  grid (ceil(M/16), ceil(N/64)), one 16 x 16
// tile per wave, both operands through LDS-DMA.
// ---------------------------------------------------------------------------------------------------------
// EPI 0: out f32 = resid + bf16(acc);  1: y bf16 = acc;  2: y bf16 = GELU(bf16(acc))
// NW = waves per workgroup (4: 64 output columns, 8: 128 -- as decode_ln_gemm_kernel: fewer, fatter workgroups where four waves would need several rounds of the chip)
template <int NKS, int EPI, int NW = 4>
__global__ __launch_bounds__(NW * 64) void decode_gemm_kernel(const bf16* __restrict__ a, const bf16* __restrict__ W, const float* __restrict__ resid, void* __restrict__ outp,
                                                              int M, int N, int ldo) {
	constexpr int K = NKS * 32, TILE = 16 * K * 2;
	extern __shared__ __attribute__((aligned(16))) char smem[];  // [A tile (shared, each wave stages its share) | NW weight tiles]
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int fr = lane & 15, fq = lane >> 4;
	const int m0 = blockIdx.x * 16, n0 = (blockIdx.y * NW + w) * 16;
	char* atile = smem;
	char* wtile = smem + TILE + w * TILE;
	stage_tile_dma<NKS>(weight_rsrc(W, N, K), K, N, n0, wtile, lane);
	stage_tile_dma<NKS>(weight_rsrc(a, M, K), K, M, m0, atile, lane, w, NW);
	const int m = m0 + fr, n = n0 + fq * 4;
	const bool ok = m < M && n < N;
	f32x4 r = {0.f, 0.f, 0.f, 0.f};
	if (EPI == 0 && ok) r = *reinterpret_cast<const f32x4*>(resid + (size_t)m * ldo + n);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
	for (int ks = 0; ks < NKS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(panel_frag(wtile, ks, lane, K), panel_frag(atile, ks, lane, K), acc, 0, 0, 0);
	if (!ok) return;
	if (EPI == 0) {
		*reinterpret_cast<f32x4*>((float*)outp + (size_t)m * ldo + n) = (f32x4){r[0] + bf16_round(acc[0]), r[1] + bf16_round(acc[1]), r[2] + bf16_round(acc[2]), r[3] + bf16_round(acc[3])};
	} else {
		bf16x4 o;
#pragma unroll
		for (int i = 0; i < 4; ++i) o[i] = EPI == 2 ? (bf16)gelu_erf(bf16_round(acc[i])) : (bf16)acc[i];
		*reinterpret_cast<bf16x4*>((bf16*)outp + (size_t)m * ldo + n) = o;
	}
}

// ---------------------------------------------------------------------------------------------------------
// The feed-forward half of a decode layer as ONE launch (round 5): out[M][E] (f32) = x + GELU(LayerNorm(x; gamma) W1^T) W2^T, hidden width 128.
// It was two (decode_ln_gemm<GELU> on 16 x 64 tiles: 5.8 us, decode_gemm<resid>: 4.4 us at 256 rows), each a dependent chain of weight fetch -> MFMA -> store that the
// next launch waits for.  Here a workgroup of eight waves owns 16 rows x 128 output columns: every wave stages ONE 16-column tile of W1 by LDS-DMA (16 x E: the whole
// matrix per workgroup, 128 KiB at E = 512), takes its 16 x 128 slice of W2 straight into registers as MFMA fragments (four 16-byte loads per lane) and the residual
// quad it will add -- all of it in flight while the eight waves normalise two rows each --, multiplies its hidden tile, parks GELU(bf16(.)) in a 16 x 128 panel,
// and after one barrier multiplies that panel with its W2 fragments.  The hidden activations never leave the CU; linear1 is recomputed by the E / 128 workgroups of
// a row block (W1 comes out of L2).  Arithmetic and rounding points are the two kernels': same LayerNorm sequence, same MFMA order per accumulator, bf16 rounding of
// the hidden pre-activation and of the linear2 product -- bit-identical (tests/test_gpu_decode_fused.py).
// ---------------------------------------------------------------------------------------------------------
template <int NKS>
__global__ __launch_bounds__(512) void decode_ffn_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const bf16* __restrict__ W1, const bf16* __restrict__ W2,
                                                         float* __restrict__ out, int M, float eps) {
	constexpr int E = NKS * 32, KF = 128, NC = (E + 255) / 256, TILE = 16 * E * 2;
	extern __shared__ __attribute__((aligned(16))) char smem[];  // [activation panel 16 x E | 8 W1 tiles 16 x E | hidden panel 16 x 128]
	char* panel = smem;
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int fr = lane & 15, fq = lane >> 4;
	const int m0 = blockIdx.x * 16, n0 = blockIdx.y * 128 + w * 16;  // this wave's 16 output columns
	char* wtile = smem + TILE + w * TILE;
	char* hpanel = smem + 9 * TILE;
	stage_tile_dma<NKS>(weight_rsrc(W1, KF, E), E, KF, w * 16, wtile, lane);  // hidden columns 16 w .. + 15
	bf16x8 wf2[4];
	load_tile<4>(wf2, weight_rsrc(W2, E, KF), KF, E, n0, lane);
	const int m = m0 + fr, n = n0 + fq * 4;
	const bool ok = m < M;
	f32x4 r = {0.f, 0.f, 0.f, 0.f};
	if (ok) r = *reinterpret_cast<const f32x4*>(x + (size_t)m * E + n);
#pragma unroll
	for (int rr = 0; rr < 2; ++rr) {
		const int row = w * 2 + rr, mr = m0 + row;
		ln_row_to_panel<NC>(x + (size_t)(mr < M ? mr : 0) * E, gamma, panel, row, E, lane, eps, mr < M);
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
	for (int ks = 0; ks < NKS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(panel_frag(wtile, ks, lane, E), panel_frag(panel, ks, lane, E), acc, 0, 0, 0);
	{  // hidden element (row fr, column 16 w + 4 fq + i): GELU of the bf16-rounded pre-activation, as the GELU epilogue of decode_ln_gemm_kernel
		bf16x4 h;
#pragma unroll
		for (int i = 0; i < 4; ++i) h[i] = (bf16)gelu_erf(bf16_round(acc[i]));
		const int col = w * 16 + fq * 4;
		*reinterpret_cast<bf16x4*>(hpanel + panel_off(fr, col >> 3, KF) + ((col & 7) << 1)) = h;
	}
	__syncthreads();
	f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
	for (int ks = 0; ks < 4; ++ks) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf2[ks], panel_frag(hpanel, ks, lane, KF), acc2, 0, 0, 0);
	if (ok) *reinterpret_cast<f32x4*>(out + (size_t)m * E + n) = (f32x4){r[0] + bf16_round(acc2[0]), r[1] + bf16_round(acc2[1]), r[2] + bf16_round(acc2[2]), r[3] + bf16_round(acc2[3])};
}

template <int EPI>
int launch_decode_gemm(const void* a, const void* w, const float* resid, void* out, int M, int N, int K, int ldo, hipStream_t stream) {
	// (the eight-wave form, NW = 8, is instantiated but not chosen here: measured -1 ... -1.7 % on greedy at 1 024 rows and beam-4 at 1 024 / 4 096 rows -- these launches run
	// several rounds of workgroups either way, and four-wave workgroups pack them better; tools/decode_wide_ab.sh.  decode_ln_gemm_kernel, one-and-a-half rounds at 256 rows, gains)
	const bool wide = false;
	const dim3 grid((M + 15) / 16, wide ? N / 128 : (N + 63) / 64), block(wide ? 512 : 256);
	const size_t shm = (size_t)(wide ? 9 : 5) * 16 * K * 2;
#define NOVIC_DG_CASE(NKS)                                                                                                                              \
	case NKS * 32: {                                                                                                                                    \
		static std::atomic<bool> attr{false};                                                                                                                       \
		if (!attr) {                                                                                                                                    \
			(void)hipFuncSetAttribute((const void*)decode_gemm_kernel<NKS, EPI, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 16 * NKS * 32 * 2);    \
			(void)hipFuncSetAttribute((const void*)decode_gemm_kernel<NKS, EPI, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 16 * NKS * 32 * 2);    \
			attr = true;                                                                                                                                \
		}                                                                                                                                               \
		if (wide) hipLaunchKernelGGL((decode_gemm_kernel<NKS, EPI, 8>), grid, block, shm, stream, (const bf16*)a, (const bf16*)w, resid, out, M, N, ldo); \
		else hipLaunchKernelGGL((decode_gemm_kernel<NKS, EPI, 4>), grid, block, shm, stream, (const bf16*)a, (const bf16*)w, resid, out, M, N, ldo);   \
		break;                                                                                                                                          \
	}
	switch (K) {
		NOVIC_DG_CASE(1)
		NOVIC_DG_CASE(2)
		NOVIC_DG_CASE(4)
		NOVIC_DG_CASE(8)
		NOVIC_DG_CASE(16)
		default:
			novic_set_error("decode GEMM: K must be 32, 64, 128, 256 or 512");
			return -22;
	}
#undef NOVIC_DG_CASE
	NOVIC_LAUNCH_CHECK();
	return 0;
}

}  // namespace

extern "C" int novic_decode_fused_supported(int E, int Kf) {
	const bool e_ok = E == 64 || E == 128 || E == 256 || E == 512;
	const bool k_ok = Kf == 32 || Kf == 64 || Kf == 128 || Kf == 256 || Kf == 512;
	return (e_ok && k_ok) ? 1 : 0;
}

extern "C" int novic_decode_ln_gemm(const float* x, const float* gamma, const void* w_bf16, void* y_bf16, int M, int N, int E, int ldy, int gelu, float eps, hipStream_t stream) {
	NOVIC_CHECK(x && gamma && w_bf16 && y_bf16, "novic_decode_ln_gemm: null pointer");
	NOVIC_CHECK(M >= 0 && N >= 4 && N % 4 == 0 && ldy >= N && ldy % 4 == 0, "novic_decode_ln_gemm: bad shape (N and ldy multiples of 4)");
	NOVIC_CHECK(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_bf16 & 15) == 0 && ((uintptr_t)y_bf16 & 7) == 0, "novic_decode_ln_gemm: misaligned operand");
	if (M == 0) return 0;
	// eight waves per workgroup (128 columns) where four would need more than one round of the chip and the column count allows it (N a multiple of 128 keeps every wave's
	// tile inside N: no partial workgroups)
	const bool wide = DECODE_LN_WIDE && ((M + 15) / 16) * ((N + 63) / 64) > 256 && N % 128 == 0 && E <= 512;
	const dim3 grid((M + 15) / 16, wide ? N / 128 : (N + 63) / 64), block(wide ? 512 : 256);
	const size_t shm = (size_t)(wide ? 9 : 5) * 16 * E * 2;
#define NOVIC_LNG_LAUNCH(NKS, G)                                                                                                                  \
	{                                                                                                                                             \
		static std::atomic<bool> attr{false};                                                                                                                 \
		if (!attr) {                                                                                                                              \
			(void)hipFuncSetAttribute((const void*)decode_ln_gemm_kernel<NKS, G, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 16 * NKS * 32 * 2); \
			(void)hipFuncSetAttribute((const void*)decode_ln_gemm_kernel<NKS, G, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 16 * NKS * 32 * 2); \
			attr = true;                                                                                                                          \
		}                                                                                                                                         \
		if (wide) hipLaunchKernelGGL((decode_ln_gemm_kernel<NKS, G, 8>), grid, block, shm, stream, x, gamma, (const bf16*)w_bf16, (bf16*)y_bf16, M, N, ldy, eps); \
		else hipLaunchKernelGGL((decode_ln_gemm_kernel<NKS, G, 4>), grid, block, shm, stream, x, gamma, (const bf16*)w_bf16, (bf16*)y_bf16, M, N, ldy, eps); \
	}
#define NOVIC_LNG_CASE(NKS)                    \
	case NKS * 32:                             \
		if (gelu) NOVIC_LNG_LAUNCH(NKS, true)  \
		else NOVIC_LNG_LAUNCH(NKS, false)      \
		break;
	switch (E) {
		NOVIC_LNG_CASE(2)
		NOVIC_LNG_CASE(4)
		NOVIC_LNG_CASE(8)
		NOVIC_LNG_CASE(16)
		default:
			novic_set_error("novic_decode_ln_gemm: hidden size must be 64, 128, 256 or 512 (novic_decode_fused_supported)");
			return -22;
	}
#undef NOVIC_LNG_LAUNCH
#undef NOVIC_LNG_CASE
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_decode_ffn_supported(int E, int Kf) { return (Kf == 128 && (E == 128 || E == 256 || E == 512)) ? 1 : 0; }

extern "C" int novic_decode_ffn(const float* x, const float* gamma, const void* w1_bf16, const void* w2_bf16, float* out, int M, int E, int Kf, float eps, hipStream_t stream) {
	NOVIC_CHECK(x && gamma && w1_bf16 && w2_bf16 && out && out != x, "novic_decode_ffn: null pointer (or out == x: every column block of a row block reads all of x's rows)");
	NOVIC_CHECK(novic_decode_ffn_supported(E, Kf), "novic_decode_ffn: hidden width 128, model width 128 / 256 / 512 (novic_decode_ffn_supported)");
	NOVIC_CHECK((((uintptr_t)x | (uintptr_t)out | (uintptr_t)w1_bf16 | (uintptr_t)w2_bf16) & 15) == 0, "novic_decode_ffn: operands must be 16-byte aligned");
	if (M <= 0) return 0;
	const dim3 grid((M + 15) / 16, E / 128), block(512);
#define NOVIC_DFFN_CASE(NKS)                                                                                                                                    \
	case NKS * 32: {                                                                                                                                            \
		constexpr int LDS = 9 * 16 * NKS * 32 * 2 + 16 * 128 * 2;                                                                                               \
		static std::atomic<bool> attr{false};                                                                                                                   \
		if (!attr) {                                                                                                                                            \
			(void)hipFuncSetAttribute((const void*)decode_ffn_kernel<NKS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);                                    \
			attr = true;                                                                                                                                        \
		}                                                                                                                                                       \
		hipLaunchKernelGGL((decode_ffn_kernel<NKS>), grid, block, LDS, stream, x, gamma, (const bf16*)w1_bf16, (const bf16*)w2_bf16, out, M, eps);              \
		break;                                                                                                                                                  \
	}
	switch (E) {
		NOVIC_DFFN_CASE(4)
		NOVIC_DFFN_CASE(8)
		NOVIC_DFFN_CASE(16)
		default: return -22;
	}
#undef NOVIC_DFFN_CASE
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_decode_gemm_resid(const void* a_bf16, const void* w_bf16, const float* resid, float* out, int M, int N, int K, hipStream_t stream) {
	NOVIC_CHECK(a_bf16 && w_bf16 && resid && out, "novic_decode_gemm_resid: null pointer");
	NOVIC_CHECK(M >= 0 && N >= 4 && N % 4 == 0, "novic_decode_gemm_resid: N must be a multiple of 4");
	NOVIC_CHECK((((uintptr_t)a_bf16 | (uintptr_t)w_bf16 | (uintptr_t)resid | (uintptr_t)out) & 15) == 0, "novic_decode_gemm_resid: operands must be 16-byte aligned");
	if (M == 0) return 0;
	return launch_decode_gemm<0>(a_bf16, w_bf16, resid, out, M, N, K, N, stream);
}

extern "C" int novic_decode_gemm(const void* a_bf16, const void* w_bf16, void* y_bf16, int M, int N, int K, int ldy, int gelu, hipStream_t stream) {
	NOVIC_CHECK(a_bf16 && w_bf16 && y_bf16, "novic_decode_gemm: null pointer");
	NOVIC_CHECK(M >= 0 && N >= 4 && N % 4 == 0 && ldy >= N && ldy % 4 == 0, "novic_decode_gemm: N and ldy must be multiples of 4");
	NOVIC_CHECK((((uintptr_t)a_bf16 | (uintptr_t)w_bf16) & 15) == 0 && ((uintptr_t)y_bf16 & 7) == 0, "novic_decode_gemm: misaligned operand");
	if (M == 0) return 0;
	return gelu ? launch_decode_gemm<2>(a_bf16, w_bf16, nullptr, y_bf16, M, N, K, ldy, stream) : launch_decode_gemm<1>(a_bf16, w_bf16, nullptr, y_bf16, M, N, K, ldy, stream);
}
