// Decoder self-attention over short prefix+label sequences (S <= 32), forward and backward, one wave per (sequence, head).
//
// reference: nn.TransformerEncoderLayer._sa_block -> F.scaled_dot_product_attention with the additive mask built in
// embedding_decoder.py:651-654 (prefix block bidirectional, rest causal) + :696-712 (per-row key padding, column 0 never masked).
// Here the mask is never materialised: allowed(i,j) = (j <= i or (i < P and j < P)) and not (key_pad[a][j] and j > 0).
//
// The whole S x S problem of one head sits in one wave's registers: the Q/K/V (/dO) rows of the head are staged once into
// LDS with 16-byte coalesced loads, QK^T is issued "swapped" (keys on MFMA rows, queries on columns) so each lane owns one
// query column and its softmax reduces in-lane + 2 shuffles, and P goes straight back into the PV MFMA as the B operand --
// the MFMA k index is permuted (k = 8g+j <-> key 4g+j | 16+4g+j-4) so that no lane movement is needed.  Operands that must be
// read "down the rows" (V, and in backward K / Q / dO) come from the same LDS image through ds_read_b64_tr_b16.
// Algorithmic bytes (HBM): forward 4*S*3E + 2*S*E per sequence... i.e. read qkv once, write o once; the kernel is HBM/issue-bound.
#include "common.hpp"
#include "novic_hip.h"

namespace {

template <int D>
__device__ __forceinline__ int lds_off(int row, int col) {
	constexpr int CPR = D / 8;  // 16-byte chunks per row
	return row * (D * 2) + ((((col >> 3) ^ (row & (CPR - 1) & 7))) << 4) + ((col & 7) << 1);
}

// Stage rows [0, S) x D of one head into LDS (zero rows up to ROWS).
template <int D, int ROWS>
__device__ __forceinline__ void stage_head(char* lds, const bf16* src, int row_stride, int S, int lane) {
	constexpr int CPR = D / 8;
	for (int c = lane; c < ROWS * CPR; c += 64) {
		const int row = c / CPR, ch = c - row * CPR;
		uint4 v = {0, 0, 0, 0};
		if (row < S) v = *reinterpret_cast<const uint4*>(src + (size_t)row * row_stride + ch * 8);
		*reinterpret_cast<uint4*>(lds + lds_off<D>(row, ch * 8)) = v;
	}
}

// lane l -> X[tile*16 + (l&15)][ks*32 + 8*(l>>4) + 0..7]   (zero beyond D)
template <int D>
__device__ __forceinline__ bf16x8 row_frag(const char* lds, int tile, int ks, int lane) {
	const int col = ks * 32 + 8 * (lane >> 4);
	bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
	if (col >= D) return z;
	return *reinterpret_cast<const bf16x8*>(lds + lds_off<D>(tile * 16 + (lane & 15), col));
}

// lane l -> X[idx(kk)][dt*16 + (l&15)] for kk = 8g + j: j < 4 -> row 4g+j, j >= 4 -> row 16+4g+(j-4) (only if NTS == 2)
template <int D, int NTS>
__device__ __forceinline__ bf16x8 tr_frag(const char* lds, int dt, int lane) {
	typedef bf16x4 __attribute__((address_space(3))) * lds4_t;
	const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
	const int col = dt * 16 + 4 * p;
	bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(lds + lds_off<D>(4 * g + q, col)));
	bf16x4 hi = {0, 0, 0, 0};
	if (NTS == 2) hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(lds + lds_off<D>(16 + 4 * g + q, col)));
	return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ float group_sum(float v) {  // across the 4 lane groups that share lane&15
	v += __shfl_xor(v, 16, 64);
	v += __shfl_xor(v, 32, 64);
	return v;
}
__device__ __forceinline__ float group_max(float v) {
	v = fmaxf(v, __shfl_xor(v, 16, 64));
	v = fmaxf(v, __shfl_xor(v, 32, 64));
	return v;
}

struct AttnArgs {
	const bf16* qkv;       // [A*S][3E]
	const uint8_t* keypad; // [A][S] or null
	bf16* o;               // fwd: [A*S][E]
	const bf16* d_o;       // bwd: [A*S][E]
	bf16* dqkv;            // bwd: [A*S][3E]
	int A, S, H, P, strict;
	float scale;
	DropoutDesc drop;
};

__device__ __forceinline__ bool allowed(const AttnArgs& g, const uint8_t* kp, int i, int j) {
	if (i >= g.S || j >= g.S) return false;
	const bool vis = (j <= i) || (!g.strict && i < g.P && j < g.P);
	return vis && !(kp && j > 0 && kp[j]);
}

template <int D, int NTS>
__global__ __launch_bounds__(256) void dec_attn_fwd_kernel(const AttnArgs g) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	constexpr int ROWS = NTS * 16, TILE = ROWS * D * 2, KS = (D + 31) / 32, DT = D / 16;
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	char* lq = smem + w * 3 * TILE;
	char* lk = lq + TILE;
	char* lv = lk + TILE;
	const int E = g.H * D;
	const int pair = blockIdx.x * 4 + w;
	const bool live = pair < g.A * g.H;
	const int a = live ? pair / g.H : 0, h = live ? pair - a * g.H : 0;
	const bf16* base = g.qkv + (size_t)a * g.S * 3 * E + h * D;
	stage_head<D, ROWS>(lq, base, 3 * E, g.S, lane);
	stage_head<D, ROWS>(lk, base + E, 3 * E, g.S, lane);
	stage_head<D, ROWS>(lv, base + 2 * E, 3 * E, g.S, lane);
	__syncthreads();
	const uint8_t* kp = g.keypad ? g.keypad + (size_t)a * g.S : nullptr;
	const int gq = lane >> 4;

#pragma unroll
	for (int qt = 0; qt < NTS; ++qt) {
		const int i = qt * 16 + (lane & 15);
		float p[NTS][4];
		float mx = -1e30f;
#pragma unroll
		for (int kt = 0; kt < NTS; ++kt) {
			f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
			for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<D>(lk, kt, ks, lane), row_frag<D>(lq, qt, ks, lane), acc, 0, 0, 0);
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const int j = kt * 16 + 4 * gq + r;
				p[kt][r] = allowed(g, kp, i, j) ? acc[r] * g.scale : -1e30f;
				mx = fmaxf(mx, p[kt][r]);
			}
		}
		mx = group_max(mx);
		float sum = 0.f;
#pragma unroll
		for (int kt = 0; kt < NTS; ++kt)
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				p[kt][r] = (p[kt][r] > -1e29f) ? __expf(p[kt][r] - mx) : 0.f;
				sum += p[kt][r];
			}
		sum = group_sum(sum);
		const float inv = sum > 0.f ? 1.f / sum : 0.f;
		bf16x8 pf;
#pragma unroll
		for (int kt = 0; kt < 2; ++kt)
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				float v = 0.f;
				if (kt < NTS) {
					const int j = kt * 16 + 4 * gq + r;
					v = p[kt][r] * inv;
					if (g.drop.p > 0.f) v *= dropout_scale1(g.drop, ((uint64_t)(a * g.H + h) * g.S + i) * g.S + j);
				}
				pf[kt * 4 + r] = (bf16)v;
			}
#pragma unroll
		for (int dt = 0; dt < DT; ++dt) {
			f32x4 acc = {0.f, 0.f, 0.f, 0.f};
			acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<D, NTS>(lv, dt, lane), pf, acc, 0, 0, 0);
			if (live && i < g.S) {
				bf16x4 ov = {(bf16)acc[0], (bf16)acc[1], (bf16)acc[2], (bf16)acc[3]};
				*reinterpret_cast<bf16x4*>(g.o + ((size_t)a * g.S + i) * E + h * D + dt * 16 + 4 * gq) = ov;
			}
		}
	}
}

template <int D, int NTS>
__global__ __launch_bounds__(256) void dec_attn_bwd_kernel(const AttnArgs g) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	constexpr int ROWS = NTS * 16, TILE = ROWS * D * 2, KS = (D + 31) / 32, DT = D / 16;
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	char* lq = smem + w * 4 * TILE;
	char* lk = lq + TILE;
	char* lv = lk + TILE;
	char* ld = lv + TILE;
	const int E = g.H * D;
	const int pair = blockIdx.x * 4 + w;
	const bool live = pair < g.A * g.H;
	const int a = live ? pair / g.H : 0, h = live ? pair - a * g.H : 0;
	const bf16* base = g.qkv + (size_t)a * g.S * 3 * E + h * D;
	stage_head<D, ROWS>(lq, base, 3 * E, g.S, lane);
	stage_head<D, ROWS>(lk, base + E, 3 * E, g.S, lane);
	stage_head<D, ROWS>(lv, base + 2 * E, 3 * E, g.S, lane);
	stage_head<D, ROWS>(ld, g.d_o + (size_t)a * g.S * E + h * D, E, g.S, lane);
	__syncthreads();
	const uint8_t* kp = g.keypad ? g.keypad + (size_t)a * g.S : nullptr;
	const int gq = lane >> 4;
	const uint64_t didx = (uint64_t)(a * g.H + h) * g.S;
	bf16* dq_base = g.dqkv + (size_t)a * g.S * 3 * E + h * D;

	// ---- layout 1: lane owns query column i, key rows j = 4g+r: softmax stats, delta, dS -> dQ ----
	float mx1[NTS], inv1[NTS], dl1[NTS];
#pragma unroll
	for (int qt = 0; qt < NTS; ++qt) {
		const int i = qt * 16 + (lane & 15);
		float p[NTS][4], dp[NTS][4];
		float mx = -1e30f;
#pragma unroll
		for (int kt = 0; kt < NTS; ++kt) {
			f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
			for (int ks = 0; ks < KS; ++ks) {
				acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<D>(lk, kt, ks, lane), row_frag<D>(lq, qt, ks, lane), acc, 0, 0, 0);
				acd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<D>(lv, kt, ks, lane), row_frag<D>(ld, qt, ks, lane), acd, 0, 0, 0);
			}
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const int j = kt * 16 + 4 * gq + r;
				p[kt][r] = allowed(g, kp, i, j) ? acc[r] * g.scale : -1e30f;
				dp[kt][r] = acd[r];
				mx = fmaxf(mx, p[kt][r]);
			}
		}
		mx = group_max(mx);
		float sum = 0.f;
#pragma unroll
		for (int kt = 0; kt < NTS; ++kt)
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				p[kt][r] = (p[kt][r] > -1e29f) ? __expf(p[kt][r] - mx) : 0.f;
				sum += p[kt][r];
			}
		sum = group_sum(sum);
		const float inv = sum > 0.f ? 1.f / sum : 0.f;
		float delta = 0.f;
#pragma unroll
		for (int kt = 0; kt < NTS; ++kt)
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const int j = kt * 16 + 4 * gq + r;
				p[kt][r] *= inv;
				if (g.drop.p > 0.f) dp[kt][r] *= dropout_scale1(g.drop, (didx + i) * g.S + j);
				delta += dp[kt][r] * p[kt][r];
			}
		delta = group_sum(delta);
		mx1[qt] = mx; inv1[qt] = inv; dl1[qt] = delta;
		bf16x8 dsf;
#pragma unroll
		for (int kt = 0; kt < 2; ++kt)
#pragma unroll
			for (int r = 0; r < 4; ++r) dsf[kt * 4 + r] = (bf16)(kt < NTS ? p[kt][r] * (dp[kt][r] - delta) * g.scale : 0.f);
#pragma unroll
		for (int dt = 0; dt < DT; ++dt) {
			f32x4 acc = {0.f, 0.f, 0.f, 0.f};
			acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<D, NTS>(lk, dt, lane), dsf, acc, 0, 0, 0);
			if (live && i < g.S) {
				bf16x4 ov = {(bf16)acc[0], (bf16)acc[1], (bf16)acc[2], (bf16)acc[3]};
				*reinterpret_cast<bf16x4*>(dq_base + (size_t)i * 3 * E + dt * 16 + 4 * gq) = ov;
			}
		}
	}

	// ---- layout 2: lane owns key column j, query rows i = 4g+r: Pd and dS -> dV, dK ----
#pragma unroll
	for (int kt = 0; kt < NTS; ++kt) {
		const int j = kt * 16 + (lane & 15);
		bf16x8 pdf, dsf;
#pragma unroll
		for (int qt = 0; qt < 2; ++qt) {
			if (qt < NTS) {
				f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
				for (int ks = 0; ks < KS; ++ks) {
					acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<D>(lq, qt, ks, lane), row_frag<D>(lk, kt, ks, lane), acc, 0, 0, 0);
					acd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<D>(ld, qt, ks, lane), row_frag<D>(lv, kt, ks, lane), acd, 0, 0, 0);
				}
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const int il = 4 * gq + r, i = qt * 16 + il;
					const float mx = __shfl(mx1[qt < NTS ? qt : 0], il, 64);
					const float inv = __shfl(inv1[qt < NTS ? qt : 0], il, 64);
					const float delta = __shfl(dl1[qt < NTS ? qt : 0], il, 64);
					float p = allowed(g, kp, i, j) ? __expf(acc[r] * g.scale - mx) * inv : 0.f;
					const float dm = (g.drop.p > 0.f) ? dropout_scale1(g.drop, (didx + i) * g.S + j) : 1.f;
					pdf[qt * 4 + r] = (bf16)(p * dm);
					dsf[qt * 4 + r] = (bf16)(p * (acd[r] * dm - delta) * g.scale);
				}
			} else {
#pragma unroll
				for (int r = 0; r < 4; ++r) { pdf[qt * 4 + r] = (bf16)0.f; dsf[qt * 4 + r] = (bf16)0.f; }
			}
		}
#pragma unroll
		for (int dt = 0; dt < DT; ++dt) {
			f32x4 av = {0.f, 0.f, 0.f, 0.f}, ak = {0.f, 0.f, 0.f, 0.f};
			av = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<D, NTS>(ld, dt, lane), pdf, av, 0, 0, 0);
			ak = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<D, NTS>(lq, dt, lane), dsf, ak, 0, 0, 0);
			if (live && j < g.S) {
				bf16x4 ov = {(bf16)av[0], (bf16)av[1], (bf16)av[2], (bf16)av[3]};
				bf16x4 ok = {(bf16)ak[0], (bf16)ak[1], (bf16)ak[2], (bf16)ak[3]};
				*reinterpret_cast<bf16x4*>(dq_base + (size_t)j * 3 * E + 2 * E + dt * 16 + 4 * gq) = ov;
				*reinterpret_cast<bf16x4*>(dq_base + (size_t)j * 3 * E + E + dt * 16 + 4 * gq) = ok;
			}
		}
	}
}

template <int D, int NTS>
int launch_attn(const AttnArgs& g, bool bwd, hipStream_t stream) {
	const int pairs = g.A * g.H;
	const int grid = (pairs + 3) / 4;
	const size_t shm = (size_t)4 * (bwd ? 4 : 3) * NTS * 16 * D * 2;
	if (bwd) hipLaunchKernelGGL((dec_attn_bwd_kernel<D, NTS>), dim3(grid), dim3(256), shm, stream, g);
	else hipLaunchKernelGGL((dec_attn_fwd_kernel<D, NTS>), dim3(grid), dim3(256), shm, stream, g);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

int dispatch_attn(const AttnArgs& g, int D, bool bwd, hipStream_t stream) {
	const int nts = (g.S + 15) / 16;
#define NOVIC_ATTN_CASE(DD)                                                         \
	case DD:                                                                        \
		return nts == 1 ? launch_attn<DD, 1>(g, bwd, stream) : launch_attn<DD, 2>(g, bwd, stream);
	switch (D) {
		NOVIC_ATTN_CASE(16)
		NOVIC_ATTN_CASE(32)
		NOVIC_ATTN_CASE(64)
		default:
			novic_set_error("decoder attention supports head_dim 16, 32 or 64");
			return -22;
	}
#undef NOVIC_ATTN_CASE
}

}  // namespace

extern "C" int novic_dec_attn_fwd(const void* qkv_bf16, const uint8_t* key_pad, void* o_bf16, int A, int S, int H, int D, int P, int strictly_causal, float drop_p,
                                  uint64_t seed, uint32_t drop_site, hipStream_t stream) {
	NOVIC_CHECK(qkv_bf16 && o_bf16, "novic_dec_attn_fwd: null pointer");
	NOVIC_CHECK(S >= 1 && S <= 32, "novic_dec_attn_fwd: sequence length must be in [1, 32]");
	NOVIC_CHECK(A >= 0 && H >= 1 && P >= 1, "novic_dec_attn_fwd: bad shape");
	if (A == 0) return 0;
	AttnArgs g = {(const bf16*)qkv_bf16, key_pad, (bf16*)o_bf16, nullptr, nullptr, A, S, H, P, strictly_causal, 1.f / sqrtf((float)D),
	              {drop_p, (uint32_t)seed, (uint32_t)(seed >> 32), drop_site}};
	return dispatch_attn(g, D, false, stream);
}

extern "C" int novic_dec_attn_bwd(const void* qkv_bf16, const uint8_t* key_pad, const void* do_bf16, void* dqkv_bf16, int A, int S, int H, int D, int P,
                                  int strictly_causal, float drop_p, uint64_t seed, uint32_t drop_site, hipStream_t stream) {
	NOVIC_CHECK(qkv_bf16 && do_bf16 && dqkv_bf16, "novic_dec_attn_bwd: null pointer");
	NOVIC_CHECK(S >= 1 && S <= 32, "novic_dec_attn_bwd: sequence length must be in [1, 32]");
	NOVIC_CHECK(A >= 0 && H >= 1 && P >= 1, "novic_dec_attn_bwd: bad shape");
	if (A == 0) return 0;
	AttnArgs g = {(const bf16*)qkv_bf16, key_pad, nullptr, (const bf16*)do_bf16, (bf16*)dqkv_bf16, A, S, H, P, strictly_causal, 1.f / sqrtf((float)D),
	              {drop_p, (uint32_t)seed, (uint32_t)(seed >> 32), drop_site}};
	return dispatch_attn(g, D, true, stream);
}
