// Shared device helpers for the gfx950 (CDNA4) kernels: bf16 vectors, wave-64 reductions, Philox dropout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

typedef __bf16 bf16;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16;    // IEEE half: the residual stream of the OpenAI-family image towers (the reference runs that family in fp16 end to end, embedders.py:488-489)
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define NOVIC_WAVE 64

// Error plumbing shared by every entry point (api.cpp owns the storage).
extern "C" void novic_set_error(const char* msg);
#define NOVIC_CHECK(cond, msg)                 \
	do {                                       \
		if (!(cond)) {                         \
			novic_set_error(msg);              \
			return -22; /* -EINVAL */          \
		}                                      \
	} while (0)
#define NOVIC_LAUNCH_CHECK()                                   \
	do {                                                       \
		hipError_t e_ = hipGetLastError();                     \
		if (e_ != hipSuccess) {                                \
			novic_set_error(hipGetErrorString(e_));            \
			return -5; /* -EIO */                              \
		}                                                      \
	} while (0)

__device__ __forceinline__ float bf16_round(float x) { return (float)(bf16)x; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
	return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
	return v;
}

// ---- Philox4x32-10 (counter-based): the Gaussian / uniform draws of the embedding noise (noise.hip) ----
struct Philox4 {
	uint32_t x, y, z, w;
};
__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
	const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
	for (int i = 0; i < 10; ++i) {
		uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
		uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
		uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
		c0 = n0; c1 = n1; c2 = n2; c3 = n3;
		k0 += W0; k1 += W1;
	}
	return {c0, c1, c2, c3};
}
// uniform in [0,1) with 24 bits
__device__ __forceinline__ float u01(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

// Dropout descriptor: keep-probability scaling of 4 consecutive elements whose first flat index is idx (idx % 4 == 0).
// The mask is a pure function of (seed, site, element index), regenerated wherever it is needed (forward epilogue, the matching backward kernel).
// One 32-bit avalanche hash (two multiplies) yields the 16-bit uniforms of TWO elements: 4 integer multiplies per 4 elements where Philox4x32-10
// took 40 quarter-rate ones -- the Philox rounds were most of the GEMM epilogues of the training step (~100 us per launch) and a good part of
// the VALU-bound attention kernels.  Drop probability is quantised to 1/65536 (p = 0.1 -> 0.100006).  The Gaussian embedding noise keeps Philox.
struct DropoutDesc {
	float p;            // drop probability (0 => identity)
	uint32_t seed_lo, seed_hi;
	uint32_t site;      // distinguishes the dropout sites of one step
};
__device__ __forceinline__ uint32_t mix32(uint32_t x) {  // "lowbias32" finaliser: full avalanche in two multiplies
	x ^= x >> 16;
	x *= 0x7feb352du;
	x ^= x >> 15;
	x *= 0x846ca68bu;
	x ^= x >> 16;
	return x;
}
__device__ __forceinline__ void dropout_scale4(const DropoutDesc& d, uint64_t idx, float (&s)[4]) {
	if (d.p <= 0.f) {
		s[0] = s[1] = s[2] = s[3] = 1.f;
		return;
	}
	const uint32_t key = mix32(d.seed_lo ^ mix32(d.seed_hi + 0x9E3779B9u * (d.site + 1u)));  // uniform: hoisted out of the element loops
	const uint64_t q = idx >> 1;                                                               // pair index of elements (idx, idx + 1); q + 1 = the next pair
	const uint32_t hi = (uint32_t)(q >> 32) * 0x85EBCA6Bu;
	const uint32_t a = mix32(((uint32_t)q ^ hi) + key), b = mix32((((uint32_t)q + 1u) ^ hi) + key);
	const uint32_t thr = (uint32_t)(d.p * 65536.f + 0.5f);
	const float inv = 1.f / (1.f - d.p);
	s[0] = (a & 0xffffu) >= thr ? inv : 0.f;
	s[1] = (a >> 16) >= thr ? inv : 0.f;
	s[2] = (b & 0xffffu) >= thr ? inv : 0.f;
	s[3] = (b >> 16) >= thr ? inv : 0.f;
}
// The same masks without the early return for p = 0 (threshold 0: every element kept, scale 1 / (1 - 0) = 1): for kernels whose vector-memory wait counts
// must stay countable by the compiler -- a branch, even a uniform one, makes its s_waitcnt pass merge scoreboards and fall back to vmcnt(0) (csrc/ffn.hip).
__device__ __forceinline__ void dropout_scale4_branchless(const DropoutDesc& d, uint64_t idx, float (&s)[4]) {
	const uint32_t key = mix32(d.seed_lo ^ mix32(d.seed_hi + 0x9E3779B9u * (d.site + 1u)));
	const uint64_t q = idx >> 1;
	const uint32_t hi = (uint32_t)(q >> 32) * 0x85EBCA6Bu;
	const uint32_t a = mix32(((uint32_t)q ^ hi) + key), b = mix32((((uint32_t)q + 1u) ^ hi) + key);
	const uint32_t thr = (uint32_t)(d.p * 65536.f + 0.5f);
	const float inv = 1.f / (1.f - d.p);
	s[0] = (a & 0xffffu) >= thr ? inv : 0.f;
	s[1] = (a >> 16) >= thr ? inv : 0.f;
	s[2] = (b & 0xffffu) >= thr ? inv : 0.f;
	s[3] = (b >> 16) >= thr ? inv : 0.f;
}
__device__ __forceinline__ float dropout_scale1(const DropoutDesc& d, uint64_t idx) {  // the same mask, one element (any idx): one hash
	if (d.p <= 0.f) return 1.f;
	const uint32_t key = mix32(d.seed_lo ^ mix32(d.seed_hi + 0x9E3779B9u * (d.site + 1u)));
	const uint64_t q = idx >> 1;
	const uint32_t a = mix32(((uint32_t)q ^ ((uint32_t)(q >> 32) * 0x85EBCA6Bu)) + key);
	const uint32_t u = (idx & 1) ? (a >> 16) : (a & 0xffffu);
	return u >= (uint32_t)(d.p * 65536.f + 0.5f) ? 1.f / (1.f - d.p) : 0.f;
}

// LayerNorm row arithmetic shared by layernorm_fwd_kernel (norm.hip), the fused decode kernels (decode_fused.hip) and the fused feed-forward kernels (ffn.hip).  The row
// is held by one wave as v[c][i] = element 256 c + 4 lane + i.  Every kernel that inlines these helpers must execute the SAME IEEE operation sequence, so that fused and
// unfused paths agree bit for bit -- and `#pragma clang fp contract(off)` does not give that under -ffp-contract=fast: it only leaves the `contract` flag off the
// IR operations, while that build mode lets the backend fuse any multiply with any add.  Until round 5 layernorm_fwd_kernel accumulated the variance with v_fmac_f32
// and decode_ln_gemm_kernel with v_pk_mul_f32 + v_add_f32 (two compilations of this one function): statistics one fp32 ulp apart in some rows, a bf16 output off by an ulp
// where the value sat on a rounding midpoint, and with it greedy decoding of <= 512 rows (LayerNorm as a GEMM prologue) and of more (its own launch) a few logits apart.
// Every product that an addition follows now passes through `unfused()`: an empty asm on the value, which no combiner sees through (no instruction is emitted).
__device__ __forceinline__ float unfused(float x) {
	asm("" : "+v"(x));
	return x;
}
template <int NC>
__device__ __forceinline__ void ln_row_stats(const float (&v)[NC][4], int E, int lane, float eps, float& mean, float& rstd) {
	float s = 0.f;
#pragma unroll
	for (int c = 0; c < NC; ++c)
#pragma unroll
		for (int i = 0; i < 4; ++i) s += v[c][i];
	mean = wave_sum(s) / (float)E;
	float q = 0.f;
#pragma unroll
	for (int c = 0; c < NC; ++c) {
		const int e = c * 256 + lane * 4;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const float d = (e < E) ? v[c][i] - mean : 0.f;
			q += unfused(d * d);
		}
	}
	rstd = rsqrtf(unfused(wave_sum(q) / (float)E) + eps);
}
__device__ __forceinline__ float ln_apply(float v, float mean, float rstd, float gamma) {
	return unfused(unfused((v - mean) * rstd) * gamma);  // (the callers add a bias, if they have one, to this value: an addition of its own)
}

// Workgroup id -> problem id in XCD-MAJOR order.  Consecutive workgroups go to the eight XCDs in turn; with id = blockIdx.x neighbouring problems -- the heads of one
// image or sequence, whose operands are 128-byte pieces of the same 3-6 KiB token rows -- would be spread over all eight L2s, each fetching its pieces on its own.
// Here XCD x owns a contiguous range of ids and its workgroups walk it in dispatch order, so a token row's pieces are asked for together through one L2 (the
// tower attention kernels are bound by exactly this strided fetch: ViT-L/14 174 -> 164 us, ViT-H/14 at 378 pixels 789 -> 724 us).  NOT a rule: the decoder's
// attention kernels (one wave per 16-row sequence pair and head, eight concurrent streams instead of one) LOST 8-9 % with it (57 -> 63, 103 -> 112 us) and keep id = blockIdx.x.
__device__ __forceinline__ int xcd_major_id(int id, int total) {
	const int q8 = total >> 3, rm = total & 7, x = id & 7, slot = id >> 3;
	return (x < rm ? x * (q8 + 1) : rm * (q8 + 1) + (x - rm) * q8) + slot;
}

// erf-GELU (nn.GELU default, embedding_decoder.py:311 layer_activation "gelu") without libm's erff (two-range polynomial with branches, ~40
// instructions -- 20 us of the 47 us linear1 GEMM went into it): erfc(z) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2), t = 1 / (1 + p z), z >= 0
// (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 -- 1/30000 of a bf16 ulp of the result).  The negative tail uses erfc directly, 0.5 x erfc(|x|/sqrt 2),
// so it keeps its relative accuracy instead of cancelling in 1 + erf(x).  Returns erfc(|x| / sqrt 2) and exp(-x^2 / 2) for the gradient.
__device__ __forceinline__ float gelu_erfc_half(float x, float& gauss) {
#pragma clang fp contract(off)
	const float z = fabsf(x) * 0.70710678118654752f;
	const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
	const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
	gauss = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);  // exp(-z^2) = exp(-x^2 / 2)
	return poly * gauss;
}
__device__ __forceinline__ float gelu_erf(float x) {
#pragma clang fp contract(off)
	float gauss;
	const float e = gelu_erfc_half(x, gauss);
	return 0.5f * x * (x > 0.f ? 2.f - e : e);
}
// tanh-approximated GELU (nn.GELU(approximate='tanh'): the SigLIP towers of open_clip configs with act_kwargs.approximate = 'tanh'):
// 0.5 x (1 + tanh u) = x / (1 + exp(-2u)), u = sqrt(2 / pi) (x + 0.044715 x^3)
// (x * rcp(1 + 2^t), not x / (1 + exp(t)): the IEEE division is a ten-instruction sequence -- v_div_scale x 2, v_rcp, four fmas, v_div_fmas, v_div_fixup -- and the
// QuickGELU store phase of a 256 x 256 tile took 6.2-7.5 us against 2.4-3.0 us with a bias alone (tools/gemm_timeline.py 12800,3072,768 qgelu); v_rcp_f32 is within 1 ulp,
// 1/250 of a bf16 ulp of the result)
__device__ __forceinline__ float sigmoid_mul(float x, float neg_t_log2e) {
#pragma clang fp contract(off)
	return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(neg_t_log2e));
}
__device__ __forceinline__ float gelu_tanh(float x) {
#pragma clang fp contract(off)
	const float u2 = 1.5957691216057308f * (x + unfused(0.044715f * x * x * x));
	return sigmoid_mul(x, -1.4426950408889634f * u2);
}
// QuickGELU of OpenAI CLIP (x sigmoid(1.702 x)): the activation of ViT-B/32 and its text tower
__device__ __forceinline__ float quick_gelu(float x) {
#pragma clang fp contract(off)
	return sigmoid_mul(x, -2.4554669595930157f * x);  // 1.702 log2(e)
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
#pragma clang fp contract(off)
	float gauss;
	const float e = gelu_erfc_half(x, gauss);
	const float cdf = 0.5f * (x > 0.f ? 2.f - e : e);
	return cdf + unfused(x * (0.39894228040143268f * gauss));  // (unfused: see the LayerNorm helpers above -- the pragma alone does not keep this addition apart under -ffp-contract=fast)
}
