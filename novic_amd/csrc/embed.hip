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

// The same with layer 0's norm1 behind it (round 6): the row is in the wave's registers when it has been assembled, so its LayerNorm -- the statistics and the affine map
// of layernorm_fwd_kernel, the shared helpers of common.hpp on the same register layout: bit-identical -- is formed there and ln1 (bf16, the QKV GEMM's operand) leaves with
// x0, instead of a second launch reading x0 back (126 MB per step at the bench batch, 30 us).  No LayerNorm bias (the released recipe; layer_bias models keep two launches).
template <int NC>
__global__ __launch_bounds__(256) void embed_fwd_ln_kernel(const bf16* __restrict__ prefix, const void* __restrict__ tokens, int tok_bytes, int tok_ld,
                                                           const float* __restrict__ wtok, const float* __restrict__ pos, float* __restrict__ x0, int A, int S, int P,
                                                           int E, int V, int B, int mrep, int multi_first, DropoutDesc drop, const int* __restrict__ seq_start,
                                                           const int* __restrict__ seq_len, const float* __restrict__ gamma, bf16* __restrict__ ln_out, float eps) {
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int rows = A * S;
	f32x4 gm[NC];
#pragma unroll
	for (int c = 0; c < NC; ++c) {
		const int e = c * 256 + lane * 4;
		gm[c] = e < E ? *reinterpret_cast<const f32x4*>(gamma + e) : (f32x4){0.f, 0.f, 0.f, 0.f};
	}
	for (int drow = blockIdx.x * 4 + w; drow < rows; drow += gridDim.x * 4) {
		const int a = drow / S, s = drow - a * S;
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
		float r[NC][4];
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int e = c * 256 + lane * 4;
			if (e < E) {
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
#pragma unroll
				for (int i = 0; i < 4; ++i) r[c][i] = (v[i] + pe[i]) * sc[i];  // (exactly embed_fwd_kernel's expression)
				*reinterpret_cast<f32x4*>(x0 + (size_t)row * E + e) = (f32x4){r[c][0], r[c][1], r[c][2], r[c][3]};
			} else {
				r[c][0] = r[c][1] = r[c][2] = r[c][3] = 0.f;
			}
		}
		float mean, rstd;
		ln_row_stats<NC>(r, E, lane, eps, mean, rstd);
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int e = c * 256 + lane * 4;
			if (e < E) {
				bf16x4 ob = {(bf16)ln_apply(r[c][0], mean, rstd, gm[c][0]), (bf16)ln_apply(r[c][1], mean, rstd, gm[c][1]), (bf16)ln_apply(r[c][2], mean, rstd, gm[c][2]),
				             (bf16)ln_apply(r[c][3], mean, rstd, gm[c][3])};
				*reinterpret_cast<bf16x4*>(ln_out + (size_t)row * E + e) = ob;
			}
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

// ---------------------------------------------------------------------------------------------------------------------------------------------------------------
// Layer 0's norm1 backward IN FRONT of the embedding backward, one launch (round 6).  The separate launches were layernorm_bwd_kernel -- dx0 = dx + LN'(dln), written as
// fp32 rows (126 MB at the bench batch) -- and embed_bwd_kernel, which read those rows back and scattered them: here a wave forms a row's dx0 in its registers (the
// arithmetic of layernorm_bwd_kernel, expression for expression) and scatters it straight away; dx0 never exists in memory.  Work layout = embed_bwd_kernel's
// (blockIdx.y = position s, blockIdx.x strides over samples / sequences); the token rows' atomics keep the lane-contiguous shape that kernel found necessary (every
// atomic wave-instruction adds into 256 contiguous bytes: 63 against 205 us) by turning each 256-element chunk of the row through 1 KiB of wave-private LDS.
//   dgamma[e] += sum_rows dln * xhat   (norm1 of layer 0; per-block partials -> fp32 atomics, as layernorm_bwd_kernel)
// ---------------------------------------------------------------------------------------------------------------------------------------------------------------
template <int NC>
struct LnRowRaw {
	f32x4 dx[NC], x[NC];
	bf16x4 dy[NC];
};

template <int NC>
__global__ __launch_bounds__(256) void ln_embed_bwd_kernel(const bf16* __restrict__ dy, const float* __restrict__ x0, const float* __restrict__ gamma, const float* __restrict__ dx_in,
                                                           float* __restrict__ dgamma, const void* __restrict__ tokens, int tok_bytes, int tok_ld, float* __restrict__ dwtok,
                                                           float* __restrict__ dpos, bf16* __restrict__ dprefix, int A, int S, int P, int E, int V, int B, int mrep,
                                                           int multi_first, DropoutDesc drop, const int* __restrict__ seq_start, const int* __restrict__ seq_len, float eps) {
	extern __shared__ float sm[];                  // red[4][NC * 256] | tr[4][256]
	float* red = sm;
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	float* tr = sm + 4 * NC * 256 + w * 256;      // wave-private
	const int s = blockIdx.y;
	const int items = (s < P) ? B : A;
	f32x4 gm[NC];
	float dg[NC][4], acc[NC][4];
#pragma unroll
	for (int c = 0; c < NC; ++c) {
		const int e = c * 256 + lane * 4;
		gm[c] = e < E ? *reinterpret_cast<const f32x4*>(gamma + e) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
		for (int i = 0; i < 4; ++i) dg[c][i] = acc[c][i] = 0.f;
	}
	// row of (sequence a, position s) in the (packed) layout, or -1: the position does not exist
	auto row_of = [&](int a) -> long long {
		if (seq_len && s >= seq_len[a]) return -1;
		return seq_start ? (long long)seq_start[a] + s : (long long)a * S + s;
	};
	auto seq_of = [&](int it, int r) { return s < P ? (multi_first ? (r * B + it) : (it * mrep + r)) : it; };
	auto load = [&](LnRowRaw<NC>& raw, long long row) {  // (a row that does not exist: row 0 is read and never used -- no branch around the loads)
		const size_t o = (size_t)(row < 0 ? 0 : row) * E;
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			int e = c * 256 + lane * 4;
			e = e < E ? e : E - 4;
			raw.dx[c] = *reinterpret_cast<const f32x4*>(dx_in + o + e);
			raw.x[c] = *reinterpret_cast<const f32x4*>(x0 + o + e);
			raw.dy[c] = *reinterpret_cast<const bf16x4*>(dy + o + e);
		}
	};
	// dx0 of one row = dx_in + LN'(dy) -- layernorm_bwd_kernel's arithmetic -- times the INPUT dropout's mask (the forward multiplied x0 by it); dgamma partials on the way
	auto row_grad = [&](const LnRowRaw<NC>& raw, long long row, float (&g)[NC][4]) {
		float xr[NC][4], dyr[NC][4];
		float sum = 0.f;
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const bool in = c * 256 + lane * 4 < E;
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				xr[c][i] = in ? raw.x[c][i] : 0.f;
				dyr[c][i] = in ? (float)raw.dy[c][i] : 0.f;
				sum += xr[c][i];
			}
		}
		const float mean = wave_sum(sum) / (float)E;
		float q = 0.f;
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int e = c * 256 + lane * 4;
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const float d = (e < E) ? xr[c][i] - mean : 0.f;
				q += d * d;
			}
		}
		const float rstd = rsqrtf(wave_sum(q) / (float)E + eps);
		float s1 = 0.f, s2 = 0.f;
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int e = c * 256 + lane * 4;
			if (e < E) {
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const float xhat = (xr[c][i] - mean) * rstd;
					const float dxh = dyr[c][i] * gm[c][i];
					dg[c][i] += dyr[c][i] * xhat;
					s1 += dxh;
					s2 += dxh * xhat;
					xr[c][i] = xhat;
					dyr[c][i] = dxh;
				}
			}
		}
		s1 = wave_sum(s1) / (float)E;
		s2 = wave_sum(s2) / (float)E;
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int e = c * 256 + lane * 4;
			float sc[4];
			dropout_scale4(drop, (uint64_t)row * E + e, sc);
#pragma unroll
			for (int i = 0; i < 4; ++i) g[c][i] = e < E ? (raw.dx[c][i] + rstd * (dyr[c][i] - s1 - xr[c][i] * s2)) * sc[i] : 0.f;
		}
	};
	const int stride = gridDim.x * 4;
	int it = blockIdx.x * 4 + w, r = 0;
	const int reps = s < P ? mrep : 1;
	LnRowRaw<NC> cur;
	long long row = it < items ? row_of(seq_of(it, 0)) : -1;
	if (it < items) load(cur, row);
	float sum[NC][4];
#pragma unroll
	for (int c = 0; c < NC; ++c) sum[c][0] = sum[c][1] = sum[c][2] = sum[c][3] = 0.f;
	while (it < items) {
		// the next row's operands are requested before this one is reduced
		int nit = it, nr = r + 1;
		if (nr == reps) { nit = it + stride; nr = 0; }
		LnRowRaw<NC> nxt;
		const long long nrow = nit < items ? row_of(seq_of(nit, nr)) : -1;
		load(nxt, nrow);
		float g[NC][4];
		if (row >= 0) {
			row_grad(cur, row, g);
#pragma unroll
			for (int c = 0; c < NC; ++c)
#pragma unroll
				for (int i = 0; i < 4; ++i) acc[c][i] += g[c][i];
		}
		if (s < P) {
			if (row >= 0) {
#pragma unroll
				for (int c = 0; c < NC; ++c)
#pragma unroll
					for (int i = 0; i < 4; ++i) sum[c][i] += g[c][i];
			}
			if (r == reps - 1) {  // the sample's targets are summed: the row of the prefix-MLP weight gradient's operand
#pragma unroll
				for (int c = 0; c < NC; ++c) {
					const int e = c * 256 + lane * 4;
					if (e < E) *reinterpret_cast<bf16x4*>(dprefix + ((size_t)it * P + s) * E + e) = (bf16x4){(bf16)sum[c][0], (bf16)sum[c][1], (bf16)sum[c][2], (bf16)sum[c][3]};
					sum[c][0] = sum[c][1] = sum[c][2] = sum[c][3] = 0.f;
				}
			}
		} else if (row >= 0) {
			long long t = load_token(tokens, tok_bytes, (size_t)it * tok_ld + (s - P));
			t = t < 0 ? 0 : (t >= V ? V - 1 : t);
			float* dst = dwtok + (size_t)t * E;
#pragma unroll
			for (int c = 0; c < NC; ++c) {
				// 4 consecutive elements per lane -> element 64 i + lane per lane: every atomic instruction then adds into 256 contiguous bytes
				*reinterpret_cast<f32x4*>(tr + lane * 4) = (f32x4){g[c][0], g[c][1], g[c][2], g[c][3]};
				__builtin_amdgcn_wave_barrier();
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const int e = c * 256 + 64 * i + lane;
					const float v = tr[64 * i + lane];
					if (e < E && v != 0.f) atomicAdd(dst + e, v);
				}
				__builtin_amdgcn_wave_barrier();
			}
		}
		cur = nxt;
		row = nrow;
		it = nit;
		r = nr;
	}
	// per-block partials: the position gradient of this block's position, then layer 0's norm1 weight gradient
#pragma unroll
	for (int c = 0; c < NC; ++c)
#pragma unroll
		for (int i = 0; i < 4; ++i) red[w * (NC * 256) + c * 256 + lane * 4 + i] = acc[c][i];
	__syncthreads();
	for (int e = threadIdx.x; e < E; e += 256) {
		const float v = red[e] + red[NC * 256 + e] + red[2 * NC * 256 + e] + red[3 * NC * 256 + e];
		if (v != 0.f) atomicAdd(dpos + (size_t)s * E + e, v);
	}
	__syncthreads();
#pragma unroll
	for (int c = 0; c < NC; ++c)
#pragma unroll
		for (int i = 0; i < 4; ++i) red[w * (NC * 256) + c * 256 + lane * 4 + i] = dg[c][i];
	__syncthreads();
	for (int e = threadIdx.x; e < E; e += 256) {
		const float v = red[e] + red[NC * 256 + e] + red[2 * NC * 256 + e] + red[3 * NC * 256 + e];
		if (v != 0.f) atomicAdd(dgamma + e, v);
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

extern "C" int novic_embed_fwd_ln(const void* prefix_bf16, const void* tokens, int tok_bytes, int tok_ld, const float* wtok, const float* pos, float* x0, int A, int S, int P,
                                  int E, int V, int B, int mrep, int multi_first, float drop_p, uint64_t seed, uint32_t drop_site, const int* seq_start, const int* seq_len,
                                  const float* ln_gamma, void* ln_out_bf16, float eps, hipStream_t stream) {
	NOVIC_CHECK(prefix_bf16 && wtok && pos && x0 && ln_gamma && ln_out_bf16, "novic_embed_fwd_ln: null pointer");
	NOVIC_CHECK((seq_start == nullptr) == (seq_len == nullptr), "novic_embed_fwd_ln: seq_start and seq_len go together");
	NOVIC_CHECK(tokens || S <= P, "novic_embed_fwd_ln: tokens required when S > P");
	NOVIC_CHECK(tok_bytes == 4 || tok_bytes == 8, "novic_embed_fwd_ln: tok_bytes must be 4 or 8");
	NOVIC_CHECK(E % 4 == 0 && E <= 2048 && S >= P && P >= 1 && mrep >= 1 && A == B * mrep, "novic_embed_fwd_ln: bad shape (E a multiple of 4, at most 2048)");
	if (A <= 0) return 0;
	DropoutDesc d = {drop_p, (uint32_t)seed, (uint32_t)(seed >> 32), drop_site};
	int grid = (A * S + 3) / 4;
	if (grid > 8192) grid = 8192;
#define NOVIC_EMBED_LN(NCV)                                                                                                                                                    \
	case NCV:                                                                                                                                                                  \
		hipLaunchKernelGGL((embed_fwd_ln_kernel<NCV>), dim3(grid), dim3(256), 0, stream, (const bf16*)prefix_bf16, tokens, tok_bytes, tok_ld, wtok, pos, x0, A, S, P, E, V, B, \
		                   mrep, multi_first, d, seq_start, seq_len, ln_gamma, (bf16*)ln_out_bf16, eps);                                                                      \
		break;
	switch ((E + 255) / 256) {
		NOVIC_EMBED_LN(1) NOVIC_EMBED_LN(2) NOVIC_EMBED_LN(3) NOVIC_EMBED_LN(4) NOVIC_EMBED_LN(5) NOVIC_EMBED_LN(6) NOVIC_EMBED_LN(7) NOVIC_EMBED_LN(8)
	}
#undef NOVIC_EMBED_LN
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

extern "C" int novic_ln_embed_bwd(const void* dy_bf16, const float* x0, const float* ln_gamma, const float* dx_in, float* dgamma, const void* tokens, int tok_bytes, int tok_ld,
                                  float* dwtok, float* dpos, void* dprefix_bf16, int A, int S, int P, int E, int V, int B, int mrep, int multi_first, float drop_p, uint64_t seed,
                                  uint32_t drop_site, const int* seq_start, const int* seq_len, float eps, hipStream_t stream) {
	NOVIC_CHECK(dy_bf16 && x0 && ln_gamma && dx_in && dgamma && dwtok && dpos && dprefix_bf16, "novic_ln_embed_bwd: null pointer");
	NOVIC_CHECK((seq_start == nullptr) == (seq_len == nullptr), "novic_ln_embed_bwd: seq_start and seq_len go together");
	NOVIC_CHECK(tokens || S <= P, "novic_ln_embed_bwd: tokens required when S > P");
	NOVIC_CHECK(tok_bytes == 4 || tok_bytes == 8, "novic_ln_embed_bwd: tok_bytes must be 4 or 8");
	NOVIC_CHECK(E % 4 == 0 && E <= 1024 && S >= P && P >= 1 && mrep >= 1 && A == B * mrep, "novic_ln_embed_bwd: bad shape (E a multiple of 4, at most 1024)");
	if (A <= 0) return 0;
	DropoutDesc d = {drop_p, (uint32_t)seed, (uint32_t)(seed >> 32), drop_site};
	int gx = (A + 3) / 4;
	if (gx > 128) gx = 128;
#define NOVIC_LN_EMBED(NCV)                                                                                                                                                      \
	case NCV:                                                                                                                                                                    \
		hipLaunchKernelGGL((ln_embed_bwd_kernel<NCV>), dim3(gx, S), dim3(256), (4 * NCV * 256 + 4 * 256) * sizeof(float), stream, (const bf16*)dy_bf16, x0, ln_gamma, dx_in, dgamma, \
		                   tokens, tok_bytes, tok_ld, dwtok, dpos, (bf16*)dprefix_bf16, A, S, P, E, V, B, mrep, multi_first, d, seq_start, seq_len, eps);                      \
		break;
	switch ((E + 255) / 256) {
		NOVIC_LN_EMBED(1) NOVIC_LN_EMBED(2) NOVIC_LN_EMBED(3) NOVIC_LN_EMBED(4)
	}
#undef NOVIC_LN_EMBED
	NOVIC_LAUNCH_CHECK();
	return 0;
}
