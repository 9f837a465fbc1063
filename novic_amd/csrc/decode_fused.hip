// Fused per-layer kernels of a KV-cached decode step (one new position per beam, M = beams x samples rows of a few hundred to a few thousand).
//
// At these sizes the unfused layer -- LayerNorm, QKV GEMM, attention, out-proj GEMM (+residual), LayerNorm, linear1 GEMM (+GELU), linear2 GEMM
// (+residual) -- is seven launches of 2..96 workgroups whose time is launch / dependency latency, not work.  Here a layer is three launches:
//
//   novic_decode_ln_gemm    y = LayerNorm(x) W^T                      (QKV; also usable for any LayerNorm + bias-free linear)
//   novic_decode_attn       (decode.hip, unchanged)
//   novic_decode_post_attn  x' = x + att Wo^T;  x'' = x' + GELU(LayerNorm(x') W1^T) W2^T
//
// A workgroup owns 16 rows (one MFMA row tile).  Activations of the 16 rows live in LDS / registers for the whole chain; the weights are never
// staged: every wave reads its MFMA B fragments straight from L2 in fragment layout (lane (r, q) <- W[n0 + r][32 ks + 8 q .. +7], 16 bytes), because
// no two waves of a workgroup share a weight element and the 16-row A panel is reused from registers across all of a wave's column tiles.
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

// B fragment of W [N][ldw] (K-contiguous) straight from memory: rows n0 + (lane & 15), k = 32 ks + 8 (lane >> 4) .. +7; rows >= N read as zero
__device__ __forceinline__ bf16x8 weight_frag(const bf16* W, int ldw, int N, int n0, int ks, int lane) {
	const int n = n0 + (lane & 15);
	bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
	if (n >= N) return z;
	return *reinterpret_cast<const bf16x8*>(W + (size_t)n * ldw + ks * 32 + (lane >> 4) * 8);
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

// ---------------------------------------------------------------------------------------------------------
// y[M][ldy] (bf16) = LayerNorm(x[M][E]; gamma) W[N][E]^T.  grid (ceil(M/16), ceil(N/256)), 4 waves x 64 output columns.
// ---------------------------------------------------------------------------------------------------------
template <int NKS>
__global__ __launch_bounds__(256) void decode_ln_gemm_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const bf16* __restrict__ W, bf16* __restrict__ y,
                                                             int M, int N, int ldy, float eps) {
	constexpr int E = NKS * 32, NC = (E + 255) / 256;
	__shared__ __attribute__((aligned(16))) char panel[16 * E * 2];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int m0 = blockIdx.x * 16;
#pragma unroll
	for (int rr = 0; rr < 4; ++rr) {
		const int row = w * 4 + rr, m = m0 + row;
		ln_row_to_panel<NC>(x + (size_t)(m < M ? m : 0) * E, gamma, panel, row, E, lane, eps, m < M);
	}
	__syncthreads();
	bf16x8 af[NKS];
#pragma unroll
	for (int ks = 0; ks < NKS; ++ks) af[ks] = panel_frag(panel, ks, lane, E);
	const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
	for (int nt = 0; nt < 4; ++nt) {
		const int n0 = blockIdx.y * 256 + w * 64 + nt * 16;
		if (n0 >= N) break;  // wave-uniform
		bf16x8 wf[NKS];
#pragma unroll
		for (int ks = 0; ks < NKS; ++ks) wf[ks] = weight_frag(W, E, N, n0, ks, lane);
		f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
		for (int ks = 0; ks < NKS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], af[ks], acc, 0, 0, 0);
		const int m = m0 + fr, n = n0 + fq * 4;
		if (m < M && n < N) {
			bf16x4 o = {(bf16)acc[0], (bf16)acc[1], (bf16)acc[2], (bf16)acc[3]};
			*reinterpret_cast<bf16x4*>(y + (size_t)m * ldy + n) = o;
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// x_out = xm + bf16(GELU-MLP(LayerNorm(xm))),  xm = x + bf16(att Wo^T)          (E = 32 NKS, feed-forward width Kf = 32 NKF; one workgroup = 16 rows)
// ---------------------------------------------------------------------------------------------------------
template <int NKS, int NKF>
__global__ __launch_bounds__(256) void decode_post_attn_kernel(const bf16* __restrict__ att, const bf16* __restrict__ Wo, const float* __restrict__ x, const float* __restrict__ gamma2,
                                                               const bf16* __restrict__ W1, const bf16* __restrict__ W2, float* __restrict__ x_out, int M, float eps) {
	constexpr int E = NKS * 32, KF = NKF * 32, NC = (E + 255) / 256;
	constexpr int NT = (E / 16 + 3) / 4, NTF = (KF / 16 + 3) / 4;  // 16-column tiles per wave (E-wide and Kf-wide outputs)
	__shared__ __attribute__((aligned(16))) float xm[16][E];        // residual stream after the attention block (fp32)
	__shared__ __attribute__((aligned(16))) char panel[16 * E * 2];  // LayerNorm(xm) as the linear1 A panel
	__shared__ __attribute__((aligned(16))) char hpanel[16 * KF * 2];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int fr = lane & 15, fq = lane >> 4;
	const int m0 = blockIdx.x * 16, m = m0 + fr;
	const bool mv = m < M;

	// ---- out-proj + residual ----
	{
		bf16x8 af[NKS];
		const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
		for (int ks = 0; ks < NKS; ++ks) af[ks] = mv ? *reinterpret_cast<const bf16x8*>(att + (size_t)m * E + ks * 32 + fq * 8) : z;
#pragma unroll
		for (int t = 0; t < NT; ++t) {
			const int n0 = (w * NT + t) * 16;
			if (n0 >= E) break;
			bf16x8 wf[NKS];
#pragma unroll
			for (int ks = 0; ks < NKS; ++ks) wf[ks] = weight_frag(Wo, E, E, n0, ks, lane);
			f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
			for (int ks = 0; ks < NKS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], af[ks], acc, 0, 0, 0);
			f32x4 r = {0.f, 0.f, 0.f, 0.f};
			if (mv) r = *reinterpret_cast<const f32x4*>(x + (size_t)m * E + n0 + fq * 4);
			*reinterpret_cast<f32x4*>(&xm[fr][n0 + fq * 4]) = (f32x4){r[0] + bf16_round(acc[0]), r[1] + bf16_round(acc[1]), r[2] + bf16_round(acc[2]), r[3] + bf16_round(acc[3])};
		}
	}
	__syncthreads();
	// ---- LayerNorm 2 ----
#pragma unroll
	for (int rr = 0; rr < 4; ++rr) {
		const int row = w * 4 + rr;
		ln_row_to_panel<NC>(&xm[row][0], gamma2, panel, row, E, lane, eps, m0 + row < M);
	}
	__syncthreads();
	// ---- linear1 + GELU ----
	{
		bf16x8 af[NKS];
#pragma unroll
		for (int ks = 0; ks < NKS; ++ks) af[ks] = panel_frag(panel, ks, lane, E);
#pragma unroll
		for (int t = 0; t < NTF; ++t) {
			const int n0 = (w * NTF + t) * 16;
			if (n0 >= KF) break;
			bf16x8 wf[NKS];
#pragma unroll
			for (int ks = 0; ks < NKS; ++ks) wf[ks] = weight_frag(W1, E, KF, n0, ks, lane);
			f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
			for (int ks = 0; ks < NKS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], af[ks], acc, 0, 0, 0);
			bf16x4 o;
#pragma unroll
			for (int i = 0; i < 4; ++i) o[i] = (bf16)gelu_erf(bf16_round(acc[i]));
			const int n = n0 + fq * 4;
			*reinterpret_cast<bf16x4*>(hpanel + panel_off(fr, n >> 3, KF) + ((n & 7) << 1)) = o;
		}
	}
	__syncthreads();
	// ---- linear2 + residual ----
	{
		bf16x8 af[NKF];
#pragma unroll
		for (int ks = 0; ks < NKF; ++ks) af[ks] = panel_frag(hpanel, ks, lane, KF);
#pragma unroll
		for (int t = 0; t < NT; ++t) {
			const int n0 = (w * NT + t) * 16;
			if (n0 >= E) break;
			bf16x8 wf[NKF];
#pragma unroll
			for (int ks = 0; ks < NKF; ++ks) wf[ks] = weight_frag(W2, KF, E, n0, ks, lane);
			f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
			for (int ks = 0; ks < NKF; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], af[ks], acc, 0, 0, 0);
			if (mv) {
				const f32x4 r = *reinterpret_cast<const f32x4*>(&xm[fr][n0 + fq * 4]);
				*reinterpret_cast<f32x4*>(x_out + (size_t)m * E + n0 + fq * 4) =
					(f32x4){r[0] + bf16_round(acc[0]), r[1] + bf16_round(acc[1]), r[2] + bf16_round(acc[2]), r[3] + bf16_round(acc[3])};
			}
		}
	}
}

}  // namespace

extern "C" int novic_decode_fused_supported(int E, int Kf) {
	const bool e_ok = E == 64 || E == 128 || E == 256 || E == 512;
	const bool k_ok = Kf == 32 || Kf == 64 || Kf == 128 || Kf == 256;
	return (e_ok && k_ok) ? 1 : 0;
}

extern "C" int novic_decode_ln_gemm(const float* x, const float* gamma, const void* w_bf16, void* y_bf16, int M, int N, int E, int ldy, float eps, hipStream_t stream) {
	NOVIC_CHECK(x && gamma && w_bf16 && y_bf16, "novic_decode_ln_gemm: null pointer");
	NOVIC_CHECK(M >= 0 && N >= 4 && N % 4 == 0 && ldy >= N && ldy % 4 == 0, "novic_decode_ln_gemm: bad shape (N and ldy multiples of 4)");
	NOVIC_CHECK(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_bf16 & 15) == 0 && ((uintptr_t)y_bf16 & 7) == 0, "novic_decode_ln_gemm: misaligned operand");
	if (M == 0) return 0;
	const dim3 grid((M + 15) / 16, (N + 255) / 256), block(256);
#define NOVIC_LNG_CASE(NKS)                                                                                                                             \
	case NKS * 32:                                                                                                                                      \
		hipLaunchKernelGGL((decode_ln_gemm_kernel<NKS>), grid, block, 0, stream, x, gamma, (const bf16*)w_bf16, (bf16*)y_bf16, M, N, ldy, eps);         \
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
#undef NOVIC_LNG_CASE
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_decode_post_attn(const void* att_bf16, const void* wo_bf16, const float* x, const float* gamma2, const void* w1_bf16, const void* w2_bf16, float* x_out, int M,
                                      int E, int Kf, float eps, hipStream_t stream) {
	NOVIC_CHECK(att_bf16 && wo_bf16 && x && gamma2 && w1_bf16 && w2_bf16 && x_out, "novic_decode_post_attn: null pointer");
	NOVIC_CHECK(novic_decode_fused_supported(E, Kf), "novic_decode_post_attn: unsupported hidden / feed-forward size (novic_decode_fused_supported)");
	NOVIC_CHECK((((uintptr_t)att_bf16 | (uintptr_t)wo_bf16 | (uintptr_t)x | (uintptr_t)w1_bf16 | (uintptr_t)w2_bf16 | (uintptr_t)x_out) & 15) == 0,
	            "novic_decode_post_attn: operands must be 16-byte aligned");
	if (M <= 0) return 0;
	const dim3 grid((M + 15) / 16), block(256);
#define NOVIC_PA_CASE(NKS, NKF)                                                                                                                        \
	if (E == NKS * 32 && Kf == NKF * 32) {                                                                                                             \
		hipLaunchKernelGGL((decode_post_attn_kernel<NKS, NKF>), grid, block, 0, stream, (const bf16*)att_bf16, (const bf16*)wo_bf16, x, gamma2,        \
		                   (const bf16*)w1_bf16, (const bf16*)w2_bf16, x_out, M, eps);                                                                 \
		NOVIC_LAUNCH_CHECK();                                                                                                                          \
		return 0;                                                                                                                                      \
	}
#define NOVIC_PA_ROW(NKS) NOVIC_PA_CASE(NKS, 1) NOVIC_PA_CASE(NKS, 2) NOVIC_PA_CASE(NKS, 4) NOVIC_PA_CASE(NKS, 8)
	NOVIC_PA_ROW(2)
	NOVIC_PA_ROW(4)
	NOVIC_PA_ROW(8)
	NOVIC_PA_ROW(16)
#undef NOVIC_PA_ROW
#undef NOVIC_PA_CASE
	novic_set_error("novic_decode_post_attn: unsupported size");
	return -22;
}
