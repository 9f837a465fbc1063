// Layer-0 input assembly and its backward.
//   x0[a][s][:] = dropout( (s < P ? prefix[sample(a)][s] : W_tok[token[a][s-P]]) + pos[s] )
// reference: embedding_decoder.py:665-675 (multi-target repeat), :692 (tied token embedding, utils.py:65-68), :693 / :1297 (positions + dropout).
// One wave per (a, s) row; HBM-bound: 4*E bytes written + (2*E | 4*E) read per row.
#include "common.hpp"
#include "novic_hip.h"

namespace {

__device__ __forceinline__ long long load_token(const void* tok, int tok_bytes, size_t i) {
	return tok_bytes == 8 ? ((const long long*)tok)[i] : (long long)((const int*)tok)[i];
}

__global__ __launch_bounds__(256) void embed_fwd_kernel(const bf16* __restrict__ prefix, const void* __restrict__ tokens, int tok_bytes, int tok_ld,
                                                        const float* __restrict__ wtok, const float* __restrict__ pos, float* __restrict__ x0, int A, int S, int P,
                                                        int E, int V, int B, int mrep, int multi_first, DropoutDesc drop, const int* __restrict__ seq_start,
                                                        const int* __restrict__ seq_len) {
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int rows = A * S;
	for (int drow = blockIdx.x * 4 + w; drow < rows; drow += gridDim.x * 4) {
		const int a = drow / S, s = drow - a * S;
		// packed rows: position s of sequence a is row seq_start[a] + s, and only the first seq_len[a] positions exist
		if (seq_len && s >= seq_len[a]) continue;
		const int row = seq_start ? seq_start[a] + s : drow;
		const float* src32 = nullptr;
		const bf16* src16 = nullptr;
		if (s < P) {
			const int b = multi_first ? (a % B) : (a / mrep);
			src16 = prefix + ((size_t)b * P + s) * E;
		} else {
			long long t = load_token(tokens, tok_bytes, (size_t)a * tok_ld + (s - P));
			t = t < 0 ? 0 : (t >= V ? V - 1 : t);
			src32 = wtok + (size_t)t * E;
		}
		for (int e = lane * 4; e < E; e += 256) {
			float v[4];
			if (src16) {
				const bf16x4 t = *reinterpret_cast<const bf16x4*>(src16 + e);
				v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
			} else {
				const f32x4 t = *reinterpret_cast<const f32x4*>(src32 + e);
				v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
			}
			const f32x4 pe = *reinterpret_cast<const f32x4*>(pos + (size_t)s * E + e);
			float sc[4];
			dropout_scale4(drop, (uint64_t)row * E + e, sc);
			*reinterpret_cast<f32x4*>(x0 + (size_t)row * E + e) = (f32x4){(v[0] + pe[0]) * sc[0], (v[1] + pe[1]) * sc[1], (v[2] + pe[2]) * sc[2], (v[3] + pe[3]) * sc[3]};
		}
	}
}

// Backward.  blockIdx.y = sequence position s; blockIdx.x strides over samples (s < P) or sequences (s >= P).
//   dpos[s]            += sum_a dx0[a][s]                    (per-block partial -> fp32 atomics)
//   dprefix[b][s] (bf16) = sum over the sample's mrep targets  (operand of the prefix-MLP weight-gradient GEMM)
//   dW_tok[token]      += dx0[a][s]                          (fp32 atomics; rows are 4*E contiguous bytes)
// 63 us per step of the default model (168 MB of reads, 60 MB of atomic adds at the memory-side units' ~1.3 TB/s) once every atomic instruction
// covers 256 contiguous bytes; the rows of padded positions are exactly zero and skipped.  Before that (4 consecutive elements per lane) it took
// 205 us, and two ways around the atomics were measured and dropped: bucketing the label positions by token with a counting sort and adding whole
// buckets with plain stores (sort 75 us -- scattered slot writes from one CU, 150 ns per same-address global atomic when spread over CUs -- + 35 us
// + 48 us); dealing the vocabulary out to workgroups that sum their rows in LDS (latency-bound scan of the token array per workgroup: 118 us
// + 65 us, and the END / padding id needs a route of its own or its owner reads half of dx0 alone).
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ dx0, const void* __restrict__ tokens, int tok_bytes, int tok_ld,
                                                        float* __restrict__ dwtok, float* __restrict__ dpos, bf16* __restrict__ dprefix, int A, int S, int P, int E,
                                                        int V, int B, int mrep, int multi_first, DropoutDesc drop, const int* __restrict__ seq_start,
                                                        const int* __restrict__ seq_len) {
	extern __shared__ float red[];  // [4][256]
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int s = blockIdx.y;
	const int items = (s < P) ? B : A;
	for (int e0 = 0; e0 < E; e0 += 256) {  // E <= 256 per pass keeps the per-lane accumulator at 4 floats
		float acc[4] = {0.f, 0.f, 0.f, 0.f};
		if (s < P) {
			const int e = e0 + lane * 4;  // lane owns 4 consecutive elements: 16-byte loads, 8-byte bf16 stores
			if (e < E) {
				for (int it = blockIdx.x * 4 + w; it < items; it += gridDim.x * 4) {
					float sum[4] = {0.f, 0.f, 0.f, 0.f};
					for (int r = 0; r < mrep; ++r) {
						const int a = multi_first ? (r * B + it) : (it * mrep + r);
						if (seq_len && s >= seq_len[a]) continue;  // (prefix positions always exist; kept for safety)
						const size_t row = seq_start ? (size_t)seq_start[a] + s : (size_t)a * S + s;
						const f32x4 g = *reinterpret_cast<const f32x4*>(dx0 + row * E + e);
						float sc[4];
						dropout_scale4(drop, (uint64_t)row * E + e, sc);
#pragma unroll
						for (int i = 0; i < 4; ++i) sum[i] += g[i] * sc[i];
					}
					bf16x4 o = {(bf16)sum[0], (bf16)sum[1], (bf16)sum[2], (bf16)sum[3]};
					*reinterpret_cast<bf16x4*>(dprefix + ((size_t)it * P + s) * E + e) = o;
#pragma unroll
					for (int i = 0; i < 4; ++i) acc[i] += sum[i];
				}
			}
		} else {
			// Token rows: lane owns elements e0 + 64 i + lane, so that every atomic wave-instruction adds into 256 CONTIGUOUS bytes of the gradient
			// row -- the shape the memory-side atomic units take at full rate (with 4 consecutive elements per lane an instruction touched 64 separate
			// dwords 16 bytes apart: 205 us per step for 60 MB of adds).
			for (int it0 = blockIdx.x * 4 + w; it0 < items; it0 += 2 * gridDim.x * 4) {
				float g[2][4];
				long long tk[2];
				size_t rowj[2];
				bool have[2];
#pragma unroll
				for (int j = 0; j < 2; ++j) {  // two rows in flight
					const int it = it0 + j * gridDim.x * 4 < items ? it0 + j * gridDim.x * 4 : items - 1;
					tk[j] = load_token(tokens, tok_bytes, (size_t)it * tok_ld + (s - P));
					have[j] = !(seq_len && s >= seq_len[it]);  // packed rows: a padded position has no row (its gradient is zero anyway)
					rowj[j] = seq_start ? (size_t)seq_start[it] + (have[j] ? s : 0) : (size_t)it * S + s;
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						const int e = e0 + 64 * i + lane;
						g[j][i] = (e < E && have[j]) ? dx0[rowj[j] * E + e] : 0.f;
					}
				}
#pragma unroll
				for (int j = 0; j < 2; ++j) {
					const int it = it0 + j * gridDim.x * 4;
					if (it < items && have[j]) {
						const size_t row = rowj[j];
						const long long t = tk[j] < 0 ? 0 : (tk[j] >= V ? V - 1 : tk[j]);
						float* dst = dwtok + (size_t)t * E + e0 + lane;
#pragma unroll
						for (int i = 0; i < 4; ++i) {
							const int e = e0 + 64 * i + lane;
							const float v = e < E ? g[j][i] * dropout_scale1(drop, (uint64_t)row * E + e) : 0.f;
							acc[i] += v;
							if (v != 0.f) atomicAdd(dst + 64 * i, v);
						}
					}
				}
			}
		}
		__syncthreads();
		// per-block partial of the position gradient; column of acc[i]: 4 lane + i (prefix rows) or 64 i + lane (token rows)
#pragma unroll
		for (int i = 0; i < 4; ++i) red[w * 256 + (s < P ? lane * 4 + i : 64 * i + lane)] = acc[i];
		__syncthreads();
		if (threadIdx.x + e0 < E) {
			const int t = threadIdx.x;
			const float v = red[t] + red[256 + t] + red[512 + t] + red[768 + t];
			if (v != 0.f) atomicAdd(dpos + (size_t)s * E + e0 + t, v);
		}
	}
}

}  // namespace

extern "C" int novic_embed_fwd(const void* prefix_bf16, const void* tokens, int tok_bytes, int tok_ld, const float* wtok, const float* pos, float* x0, int A, int S, int P,
                               int E, int V, int B, int mrep, int multi_first, float drop_p, uint64_t seed, uint32_t drop_site, const int* seq_start, const int* seq_len,
                               hipStream_t stream) {
	NOVIC_CHECK(prefix_bf16 && wtok && pos && x0, "novic_embed_fwd: null pointer");
	NOVIC_CHECK((seq_start == nullptr) == (seq_len == nullptr), "novic_embed_fwd: seq_start and seq_len go together");
	NOVIC_CHECK(tokens || S <= P, "novic_embed_fwd: tokens required when S > P");
	NOVIC_CHECK(tok_bytes == 4 || tok_bytes == 8, "novic_embed_fwd: tok_bytes must be 4 or 8");
	NOVIC_CHECK(E % 4 == 0 && S >= P && P >= 1 && mrep >= 1 && A == B * mrep, "novic_embed_fwd: bad shape");
	if (A <= 0) return 0;
	DropoutDesc d = {drop_p, (uint32_t)seed, (uint32_t)(seed >> 32), drop_site};
	int grid = (A * S + 3) / 4;
	if (grid > 8192) grid = 8192;
	hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid), dim3(256), 0, stream, (const bf16*)prefix_bf16, tokens, tok_bytes, tok_ld, wtok, pos, x0, A, S, P, E, V, B, mrep,
	                   multi_first, d, seq_start, seq_len);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_embed_bwd(const float* dx0, const void* tokens, int tok_bytes, int tok_ld, float* dwtok, float* dpos, void* dprefix_bf16, int A, int S, int P, int E,
                               int V, int B, int mrep, int multi_first, float drop_p, uint64_t seed, uint32_t drop_site, const int* seq_start, const int* seq_len,
                               hipStream_t stream) {
	NOVIC_CHECK(dx0 && dwtok && dpos && dprefix_bf16, "novic_embed_bwd: null pointer");
	NOVIC_CHECK((seq_start == nullptr) == (seq_len == nullptr), "novic_embed_bwd: seq_start and seq_len go together");
	NOVIC_CHECK(tokens || S <= P, "novic_embed_bwd: tokens required when S > P");
	NOVIC_CHECK(tok_bytes == 4 || tok_bytes == 8, "novic_embed_bwd: tok_bytes must be 4 or 8");
	NOVIC_CHECK(E % 4 == 0 && S >= P && P >= 1 && mrep >= 1 && A == B * mrep, "novic_embed_bwd: bad shape");
	if (A <= 0) return 0;
	DropoutDesc d = {drop_p, (uint32_t)seed, (uint32_t)(seed >> 32), drop_site};
	int gx = (A + 3) / 4;
	if (gx > 128) gx = 128;
	hipLaunchKernelGGL(embed_bwd_kernel, dim3(gx, S), dim3(256), 1024 * sizeof(float), stream, dx0, tokens, tok_bytes, tok_ld, dwtok, dpos, (bf16*)dprefix_bf16, A, S, P, E, V,
	                   B, mrep, multi_first, d, seq_start, seq_len);
	NOVIC_LAUNCH_CHECK();
	return 0;
}
