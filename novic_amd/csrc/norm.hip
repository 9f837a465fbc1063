// Row-wise kernels: L2 row normalisation -> bf16, LayerNorm forward (f32 -> bf16) and backward.
// One 64-lane wave per row; the row lives in registers (E <= 2048), statistics by wave shuffles (no LDS, no re-read).
// All of these are HBM-bound: algorithmic bytes are listed at each entry point.
#include "common.hpp"
#include "novic_hip.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;  // 4 waves

// lane-owned chunks of a row: element index = 256*c + 4*lane + {0..3}
template <int NC>
struct RowRegs {
	float v[NC][4];
};

template <int NC>
__device__ __forceinline__ void load_row_f32(RowRegs<NC>& r, const float* x, int E, int lane) {
#pragma unroll
	for (int c = 0; c < NC; ++c) {
		const int e = c * 256 + lane * 4;
		if (e < E) {
			const f32x4 t = *reinterpret_cast<const f32x4*>(x + e);
			r.v[c][0] = t[0]; r.v[c][1] = t[1]; r.v[c][2] = t[2]; r.v[c][3] = t[3];
		} else {
			r.v[c][0] = r.v[c][1] = r.v[c][2] = r.v[c][3] = 0.f;
		}
	}
}
template <int NC>
__device__ __forceinline__ void load_row_bf16(RowRegs<NC>& r, const bf16* x, int E, int lane) {
#pragma unroll
	for (int c = 0; c < NC; ++c) {
		const int e = c * 256 + lane * 4;
		if (e < E) {
			const bf16x4 t = *reinterpret_cast<const bf16x4*>(x + e);
			r.v[c][0] = (float)t[0]; r.v[c][1] = (float)t[1]; r.v[c][2] = (float)t[2]; r.v[c][3] = (float)t[3];
		} else {
			r.v[c][0] = r.v[c][1] = r.v[c][2] = r.v[c][3] = 0.f;
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// y(bf16) = x / max(||x||, 1e-12)           (F.normalize prologue of the prefix MLP, embedding_decoder.py:1276)
// ---------------------------------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(256) void rownorm_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ y, int rows, int E, int ldy) {
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (int row = blockIdx.x * ROWS_PER_BLOCK + w; row < rows; row += gridDim.x * ROWS_PER_BLOCK) {
		RowRegs<NC> r;
		load_row_f32<NC>(r, x + (size_t)row * E, E, lane);
		float ss = 0.f;
#pragma unroll
		for (int c = 0; c < NC; ++c)
#pragma unroll
			for (int i = 0; i < 4; ++i) ss += r.v[c][i] * r.v[c][i];
		ss = wave_sum(ss);
		const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int e = c * 256 + lane * 4;
			if (e < E) {
				bf16x4 o = {(bf16)(r.v[c][0] * inv), (bf16)(r.v[c][1] * inv), (bf16)(r.v[c][2] * inv), (bf16)(r.v[c][3] * inv)};
				*reinterpret_cast<bf16x4*>(y + (size_t)row * ldy + e) = o;
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm forward: y[r] (bf16) = (x[src(r)] - mean) * rstd * gamma (+ beta), src(r) = (r / seq_out) * seq_in + seq_off + r % seq_out
// ---------------------------------------------------------------------------------------------------------
// XT: the element type of x -- float, or f16 for a tower whose residual stream is IEEE half (novic_layernorm_fwd_f16: the statistics and the affine map in fp32 all the same,
// as clip's LayerNorm subclass computes them: cast up, normalise, cast back)
template <int NC, typename XT = float>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const XT* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            bf16* __restrict__ y, float* __restrict__ y32, int rows_out, int E, int seq_in, int seq_out, int seq_off, float eps,
                                                            const int* __restrict__ src_rows, const int* __restrict__ row_count) {
	if (row_count) rows_out = min(rows_out, max(*row_count, 0));  // gathered form: output row j <- input row src_rows[j], j < *row_count
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	f32x4 gm[NC], bt[NC];
#pragma unroll
	for (int c = 0; c < NC; ++c) {
		const int e = c * 256 + lane * 4;
		gm[c] = e < E ? *reinterpret_cast<const f32x4*>(gamma + e) : (f32x4){0.f, 0.f, 0.f, 0.f};
		bt[c] = (beta && e < E) ? *reinterpret_cast<const f32x4*>(beta + e) : (f32x4){0.f, 0.f, 0.f, 0.f};
	}
	// the next row is requested before the current one is reduced (see layernorm_bwd_kernel); clamped indices, no branch around the loads
	auto load = [&](f32x4 (&raw)[NC], int row) {
		row = row < rows_out ? row : rows_out - 1;
		const XT* xr = x + (size_t)(src_rows ? src_rows[row] : (row / seq_out) * seq_in + seq_off + row % seq_out) * E;
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			int e = c * 256 + lane * 4;
			e = e < E ? e : E - 4;
			if constexpr (sizeof(XT) == 2) {
				const f16x4 h = *reinterpret_cast<const f16x4*>(xr + e);
				raw[c] = (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
			} else {
				raw[c] = *reinterpret_cast<const f32x4*>(xr + e);
			}
		}
	};
	const int stride = gridDim.x * ROWS_PER_BLOCK;
	int row = blockIdx.x * ROWS_PER_BLOCK + w;
	f32x4 cur[NC];
	if (rows_out <= 0) return;
	if (row < rows_out) load(cur, row);
	for (; row < rows_out; row += stride) {
		f32x4 nxt[NC];
		load(nxt, row + stride);
		RowRegs<NC> r;
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const bool in = c * 256 + lane * 4 < E;
#pragma unroll
			for (int i = 0; i < 4; ++i) r.v[c][i] = in ? cur[c][i] : 0.f;
		}
		float mean, rstd;
		ln_row_stats<NC>(r.v, E, lane, eps, mean, rstd);
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int e = c * 256 + lane * 4;
			if (e < E) {
				float o[4];
#pragma unroll
				for (int i = 0; i < 4; ++i) o[i] = ln_apply(r.v[c][i], mean, rstd, gm[c][i]);
				if (beta) {
#pragma unroll
					for (int i = 0; i < 4; ++i) o[i] += bt[c][i];
				}
				if (y) {
					bf16x4 ob = {(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
					*reinterpret_cast<bf16x4*>(y + (size_t)row * E + e) = ob;
				}
				if (y32) *reinterpret_cast<f32x4*>(y32 + (size_t)row * E + e) = (f32x4){o[0], o[1], o[2], o[3]};
			}
		}
#pragma unroll
		for (int c = 0; c < NC; ++c) cur[c] = nxt[c];
	}
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm backward (no bias).  For every INPUT row m (rows_in of them):
//   selected(m): (m % seq_in) >= seq_off && < seq_off + seq_out   -> r = (m / seq_in) * seq_out + m % seq_in - seq_off
//   dxhat = dy[r] * gamma ; dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat))      (0 if not selected)
//   dx_out[m] = (dx_in ? dx_in[m] : 0) + dx
//   g_out[m] (bf16, optional) = dx_out[m] * dropmask(site, m*E+e)      -- operand of the next backward GEMM
//   dgamma[e] += sum_rows dy * xhat
// ---------------------------------------------------------------------------------------------------------
// raw operands of one row, loaded one loop iteration ahead of their use so that a wave always has a row's worth of loads in flight while it
// reduces the previous row (a wave that loads, reduces 3 times and stores strictly in turn leaves HBM idle half the time: 136 -> ~75 us on
// [57344 x 512]).  Indices are clamped instead of predicated: no branch around the loads, nothing forces an early s_waitcnt.
template <int NC>
struct LnBwdRow {
	f32x4 dx[NC], x[NC];
	bf16x4 dy[NC];
};

template <int NC, bool HAS_DX>
__device__ __forceinline__ void ln_bwd_load(LnBwdRow<NC>& r, const bf16* dy, const float* x, const float* dx_in, int m, int rows_in, int E, int seq_in, int seq_out,
                                            int seq_off, int lane, const int* dy_row) {
	m = m < rows_in ? m : rows_in - 1;
	int s = m % seq_in - seq_off;
	s = s < 0 ? 0 : (s >= seq_out ? seq_out - 1 : s);
	size_t yo = ((size_t)(m / seq_in) * seq_out + s) * E;
	if (dy_row) yo = (size_t)max(dy_row[m], 0) * E;  // mapped form: the upstream gradient of input row m is row dy_row[m] (< 0: none)
	const size_t xo = (size_t)m * E;
#pragma unroll
	for (int c = 0; c < NC; ++c) {
		int e = c * 256 + lane * 4;
		e = e < E ? e : E - 4;
		if (HAS_DX) r.dx[c] = *reinterpret_cast<const f32x4*>(dx_in + xo + e);
		r.x[c] = *reinterpret_cast<const f32x4*>(x + xo + e);
		r.dy[c] = *reinterpret_cast<const bf16x4*>(dy + yo + e);
	}
}

template <int NC, bool HAS_DX>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ dx_in, float* __restrict__ dx_out, bf16* __restrict__ g_out,
                                                            float* __restrict__ dgamma, int rows_in, int E, int seq_in, int seq_out, int seq_off, float eps,
                                                            DropoutDesc drop, const int* __restrict__ dy_row, const int* __restrict__ row_limit) {
	__shared__ float red[ROWS_PER_BLOCK][NC * 256];
	if (row_limit) rows_in = min(rows_in, max(*row_limit, 0));  // packed rows: only the first *row_limit exist
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	float dg[NC][4];
	f32x4 gm[NC];
#pragma unroll
	for (int c = 0; c < NC; ++c) {
		dg[c][0] = dg[c][1] = dg[c][2] = dg[c][3] = 0.f;
		const int e = c * 256 + lane * 4;
		gm[c] = e < E ? *reinterpret_cast<const f32x4*>(gamma + e) : (f32x4){0.f, 0.f, 0.f, 0.f};
	}
	const int stride = gridDim.x * ROWS_PER_BLOCK;
	int m = blockIdx.x * ROWS_PER_BLOCK + w;
	LnBwdRow<NC> cur;
	if (m < rows_in) ln_bwd_load<NC, HAS_DX>(cur, dy, x, dx_in, m, rows_in, E, seq_in, seq_out, seq_off, lane, dy_row);  // (rows_in = 0: nothing to load, the loop below does not run)
	for (; m < rows_in; m += stride) {
		LnBwdRow<NC> nxt;
		ln_bwd_load<NC, HAS_DX>(nxt, dy, x, dx_in, m + stride, rows_in, E, seq_in, seq_out, seq_off, lane, dy_row);
		const int s = m % seq_in;
		const bool sel = dy_row ? dy_row[m] >= 0 : (s >= seq_off) && (s < seq_off + seq_out);
		float dxr[NC][4];
#pragma unroll
		for (int c = 0; c < NC; ++c)
#pragma unroll
			for (int i = 0; i < 4; ++i) dxr[c][i] = HAS_DX ? cur.dx[c][i] : 0.f;
		if (sel) {
			float xr[NC][4], dyr[NC][4];
			float sum = 0.f;
#pragma unroll
			for (int c = 0; c < NC; ++c) {
				const bool in = c * 256 + lane * 4 < E;
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					xr[c][i] = in ? cur.x[c][i] : 0.f;
					dyr[c][i] = in ? (float)cur.dy[c][i] : 0.f;
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
			for (int c = 0; c < NC; ++c)
#pragma unroll
				for (int i = 0; i < 4; ++i) dxr[c][i] += rstd * (dyr[c][i] - s1 - xr[c][i] * s2);
		}
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int e = c * 256 + lane * 4;
			if (e < E) {
				__builtin_nontemporal_store((f32x4){dxr[c][0], dxr[c][1], dxr[c][2], dxr[c][3]}, reinterpret_cast<f32x4*>(dx_out + (size_t)m * E + e));
				if (g_out) {
					float sc[4];
					dropout_scale4(drop, (uint64_t)m * E + e, sc);
					bf16x4 o = {(bf16)(dxr[c][0] * sc[0]), (bf16)(dxr[c][1] * sc[1]), (bf16)(dxr[c][2] * sc[2]), (bf16)(dxr[c][3] * sc[3])};
					*reinterpret_cast<bf16x4*>(g_out + (size_t)m * E + e) = o;
				}
			}
		}
		cur = nxt;
	}
	if (dgamma) {
#pragma unroll
		for (int c = 0; c < NC; ++c)
#pragma unroll
			for (int i = 0; i < 4; ++i) red[w][c * 256 + lane * 4 + i] = dg[c][i];
		__syncthreads();
		for (int e = threadIdx.x; e < E; e += 256) {
			const float t = red[0][e] + red[1][e] + red[2][e] + red[3][e];
			if (t != 0.f) atomicAdd(dgamma + e, t);
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// Hidden layer of the prefix MLP (reference EmbeddingVectorMLP with mlp_hidden_layer != 'none' and mlp_hidden_norm, embedding_decoder.py:1247-1253):
//   z = LayerNorm(h0; gamma, beta) in fp32 on the bf16 output of linear1 (autocast runs nn.LayerNorm in fp32), y = bf16(act(z)) = the operand of linear2.
// Backward: dz = dy * act'(z), LayerNorm backward, dh0 = bf16(.), dgamma / dbeta by atomics (B rows x H <= 2048: microseconds either way).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float hidden_act(int act, float z) {
	if (act == NOVIC_ACT_RELU) return fmaxf(z, 0.f);
	if (act == NOVIC_ACT_TANH) return tanhf(z);
	if (act == NOVIC_ACT_NONE) return z;
	return gelu_erf(z);
}
__device__ __forceinline__ float hidden_act_grad(int act, float z) {
	if (act == NOVIC_ACT_RELU) return z > 0.f ? 1.f : 0.f;
	if (act == NOVIC_ACT_TANH) {
		const float y = tanhf(z);
		return 1.f - y * y;
	}
	if (act == NOVIC_ACT_NONE) return 1.f;
	return gelu_erf_grad(z);
}

template <int NC>
__global__ __launch_bounds__(256) void hidden_norm_act_fwd_kernel(const bf16* __restrict__ h0, const float* __restrict__ gamma, const float* __restrict__ beta, bf16* __restrict__ y,
                                                                  int rows, int H, int ldh, int ldy, int act, float eps) {
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (int row = blockIdx.x * ROWS_PER_BLOCK + w; row < rows; row += gridDim.x * ROWS_PER_BLOCK) {
		RowRegs<NC> r;
		load_row_bf16<NC>(r, h0 + (size_t)row * ldh, H, lane);
		float mean, rstd;
		ln_row_stats<NC>(r.v, H, lane, eps, mean, rstd);
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int e = c * 256 + lane * 4;
			if (e < H) {
				const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + e);
				const f32x4 bt = beta ? *reinterpret_cast<const f32x4*>(beta + e) : (f32x4){0.f, 0.f, 0.f, 0.f};
				bf16x4 o;
#pragma unroll
				for (int i = 0; i < 4; ++i) o[i] = (bf16)hidden_act(act, ln_apply(r.v[c][i], mean, rstd, gm[i]) + bt[i]);
				*reinterpret_cast<bf16x4*>(y + (size_t)row * ldy + e) = o;
			}
		}
	}
}

template <int NC>
__global__ __launch_bounds__(256) void hidden_norm_act_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ h0, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, bf16* __restrict__ dh0, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                  int rows, int H, int ldy, int ldh, int ldd, int act, float eps) {
	__shared__ float red[2][ROWS_PER_BLOCK][NC * 256];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	float dg[NC][4], db[NC][4];
#pragma unroll
	for (int c = 0; c < NC; ++c)
#pragma unroll
		for (int i = 0; i < 4; ++i) dg[c][i] = db[c][i] = 0.f;
	for (int row = blockIdx.x * ROWS_PER_BLOCK + w; row < rows; row += gridDim.x * ROWS_PER_BLOCK) {
		RowRegs<NC> r, g;
		load_row_bf16<NC>(r, h0 + (size_t)row * ldh, H, lane);
		load_row_bf16<NC>(g, dy + (size_t)row * ldy, H, lane);
		float mean, rstd;
		ln_row_stats<NC>(r.v, H, lane, eps, mean, rstd);
		float s1 = 0.f, s2 = 0.f;
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int e = c * 256 + lane * 4;
			if (e < H) {
				const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + e);
				const f32x4 bt = beta ? *reinterpret_cast<const f32x4*>(beta + e) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const float xhat = (r.v[c][i] - mean) * rstd;
					const float dz = g.v[c][i] * hidden_act_grad(act, ln_apply(r.v[c][i], mean, rstd, gm[i]) + bt[i]);
					const float dxh = dz * gm[i];
					dg[c][i] += dz * xhat;
					db[c][i] += dz;
					s1 += dxh;
					s2 += dxh * xhat;
					r.v[c][i] = xhat;
					g.v[c][i] = dxh;
				}
			}
		}
		s1 = wave_sum(s1) / (float)H;
		s2 = wave_sum(s2) / (float)H;
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int e = c * 256 + lane * 4;
			if (e < H) {
				bf16x4 o;
#pragma unroll
				for (int i = 0; i < 4; ++i) o[i] = (bf16)(rstd * (g.v[c][i] - s1 - r.v[c][i] * s2));
				*reinterpret_cast<bf16x4*>(dh0 + (size_t)row * ldd + e) = o;
			}
		}
	}
#pragma unroll
	for (int c = 0; c < NC; ++c)
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			red[0][w][c * 256 + lane * 4 + i] = dg[c][i];
			red[1][w][c * 256 + lane * 4 + i] = db[c][i];
		}
	__syncthreads();
	for (int e = threadIdx.x; e < H; e += 256) {
		const float tg = red[0][0][e] + red[0][1][e] + red[0][2][e] + red[0][3][e];
		const float tb = red[1][0][e] + red[1][1][e] + red[1][2][e] + red[1][3][e];
		if (tg != 0.f) atomicAdd(dgamma + e, tg);
		if (dbeta && tb != 0.f) atomicAdd(dbeta + e, tb);
	}
}

// ---------------------------------------------------------------------------------------------------------
// Post-LN layers (reference layer_norm_first = False, nn.TransformerEncoderLayer: x = norm(x + block(x))): the norm's upstream gradient is the SUM of what flows into the
// fp32 stream behind it (dy_f32: the residual path of the block above) and into its bf16 copy (dy_bf16: the input gradient of that block's first linear) -- both optional.
//   dx[m] = LayerNorm'(dy_bf16[m] + dy_f32[m]; x[m], gamma);  g_out[m] (bf16, optional) = dx[m] * dropmask(site, m E + e);  dgamma += sum dy xhat;  dbeta += sum dy
// One wave per row, no prefetch (a variant path: correctness first); fp32 atomics of per-block partials like layernorm_bwd_kernel.
// ---------------------------------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(256) void layernorm_bwd_sum_kernel(const bf16* __restrict__ dy16, const float* __restrict__ dy32, const float* __restrict__ x,
                                                                const float* __restrict__ gamma, float* __restrict__ dx_out, bf16* __restrict__ g_out, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, int rows, int E, float eps, DropoutDesc drop, const int* __restrict__ row_limit) {
	__shared__ float red[2][ROWS_PER_BLOCK][NC * 256];
	if (row_limit) rows = min(rows, max(*row_limit, 0));
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	float dg[NC][4], db[NC][4];
#pragma unroll
	for (int c = 0; c < NC; ++c)
#pragma unroll
		for (int i = 0; i < 4; ++i) dg[c][i] = db[c][i] = 0.f;
	for (int m = blockIdx.x * ROWS_PER_BLOCK + w; m < rows; m += gridDim.x * ROWS_PER_BLOCK) {
		RowRegs<NC> xr, dy;
		load_row_f32<NC>(xr, x + (size_t)m * E, E, lane);
		if (dy32) load_row_f32<NC>(dy, dy32 + (size_t)m * E, E, lane);
		else {
#pragma unroll
			for (int c = 0; c < NC; ++c) dy.v[c][0] = dy.v[c][1] = dy.v[c][2] = dy.v[c][3] = 0.f;
		}
		if (dy16) {
			RowRegs<NC> t;
			load_row_bf16<NC>(t, dy16 + (size_t)m * E, E, lane);
#pragma unroll
			for (int c = 0; c < NC; ++c)
#pragma unroll
				for (int i = 0; i < 4; ++i) dy.v[c][i] += t.v[c][i];
		}
		float mean, rstd;
		ln_row_stats<NC>(xr.v, E, lane, eps, mean, rstd);
		float s1 = 0.f, s2 = 0.f;
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int e = c * 256 + lane * 4;
			if (e < E) {
				const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + e);
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const float xhat = (xr.v[c][i] - mean) * rstd;
					const float dxh = dy.v[c][i] * gm[i];
					dg[c][i] += dy.v[c][i] * xhat;
					db[c][i] += dy.v[c][i];
					s1 += dxh;
					s2 += dxh * xhat;
					xr.v[c][i] = xhat;
					dy.v[c][i] = dxh;
				}
			}
		}
		s1 = wave_sum(s1) / (float)E;
		s2 = wave_sum(s2) / (float)E;
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int e = c * 256 + lane * 4;
			if (e < E) {
				float d[4];
#pragma unroll
				for (int i = 0; i < 4; ++i) d[i] = rstd * (dy.v[c][i] - s1 - xr.v[c][i] * s2);
				*reinterpret_cast<f32x4*>(dx_out + (size_t)m * E + e) = (f32x4){d[0], d[1], d[2], d[3]};
				if (g_out) {
					float sc[4];
					dropout_scale4(drop, (uint64_t)m * E + e, sc);
					*reinterpret_cast<bf16x4*>(g_out + (size_t)m * E + e) = (bf16x4){(bf16)(d[0] * sc[0]), (bf16)(d[1] * sc[1]), (bf16)(d[2] * sc[2]), (bf16)(d[3] * sc[3])};
				}
			}
		}
	}
#pragma unroll
	for (int c = 0; c < NC; ++c)
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			red[0][w][c * 256 + lane * 4 + i] = dg[c][i];
			red[1][w][c * 256 + lane * 4 + i] = db[c][i];
		}
	__syncthreads();
	for (int e = threadIdx.x; e < E; e += 256) {
		const float tg = red[0][0][e] + red[0][1][e] + red[0][2][e] + red[0][3][e];
		const float tb = red[1][0][e] + red[1][1][e] + red[1][2][e] + red[1][3][e];
		if (dgamma && tg != 0.f) atomicAdd(dgamma + e, tg);
		if (dbeta && tb != 0.f) atomicAdd(dbeta + e, tb);
	}
}

// dst (f32) += src (bf16), elementwise: the two parts of a post-LN layer-0 input gradient meeting in front of novic_embed_bwd
__global__ __launch_bounds__(256) void add_bf16_kernel(float* __restrict__ dst, const bf16* __restrict__ src, size_t n4) {
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
		f32x4 d = reinterpret_cast<f32x4*>(dst)[i];
		const bf16x4 s = reinterpret_cast<const bf16x4*>(src)[i];
		d[0] += (float)s[0]; d[1] += (float)s[1]; d[2] += (float)s[2]; d[3] += (float)s[3];
		reinterpret_cast<f32x4*>(dst)[i] = d;
	}
}

// ---------------------------------------------------------------------------------------------------------
// ReZero (reference TransformerEncoderLayer(rezero = 'perskip' | 'perlayer'), embedding_decoder.py:1086-1117): the block's output -- after its dropout, a bf16 tensor under
// autocast -- is multiplied in place by a learned scalar before the residual add:   out = resid + bf16(scale * branch).
// Backward, as autograd's: g = bf16(d out) is the gradient of the scaled branch; d scale += sum g * branch; the gradient of the block's last linear (in front of its
// dropout) is bf16(bf16(g * scale) * dropmask).  `scale` is a DEVICE scalar (a parameter): nothing is read back.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rezero_fwd_kernel(const float* __restrict__ resid, const bf16* __restrict__ branch, const float* __restrict__ scale, float* __restrict__ out,
                                                         size_t n4, int E4, const int* __restrict__ row_limit) {
	if (row_limit) n4 = min(n4, (size_t)max(*row_limit, 0) * E4);
	const float sc = *scale;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
		const f32x4 r = reinterpret_cast<const f32x4*>(resid)[i];
		const bf16x4 b = reinterpret_cast<const bf16x4*>(branch)[i];
		f32x4 o;
#pragma unroll
		for (int k = 0; k < 4; ++k) o[k] = r[k] + bf16_round(unfused(sc * (float)b[k]));
		reinterpret_cast<f32x4*>(out)[i] = o;
	}
}

__global__ __launch_bounds__(256) void rezero_bwd_kernel(const float* __restrict__ dx, const bf16* __restrict__ branch, const float* __restrict__ scale, float* __restrict__ dscale,
                                                         bf16* __restrict__ g_out, size_t n4, int E4, DropoutDesc drop, const int* __restrict__ row_limit) {
	__shared__ float red[4];
	if (row_limit) n4 = min(n4, (size_t)max(*row_limit, 0) * E4);
	const float sc = *scale;
	float acc = 0.f;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
		const f32x4 d = reinterpret_cast<const f32x4*>(dx)[i];
		const bf16x4 b = reinterpret_cast<const bf16x4*>(branch)[i];
		float ds[4];
		dropout_scale4(drop, (uint64_t)i * 4, ds);
		bf16x4 o;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const float g = bf16_round(d[k]);
			acc += g * (float)b[k];
			o[k] = (bf16)(bf16_round(unfused(g * sc)) * ds[k]);
		}
		reinterpret_cast<bf16x4*>(g_out)[i] = o;
	}
	acc = wave_sum(acc);
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) {
		const float t = red[0] + red[1] + red[2] + red[3];
		if (t != 0.f) atomicAdd(dscale, t);
	}
}

inline int grid_for_rows(int rows) {
	int blocks = (rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
	return blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
}

}  // namespace

#define NOVIC_NC_DISPATCH(E, CALL)                 \
	switch (((E) + 255) / 256) {                   \
		case 1: { constexpr int NC = 1; CALL; break; } \
		case 2: { constexpr int NC = 2; CALL; break; } \
		case 3: { constexpr int NC = 3; CALL; break; } \
		case 4: { constexpr int NC = 4; CALL; break; } \
		case 5: { constexpr int NC = 5; CALL; break; } \
		case 6: { constexpr int NC = 6; CALL; break; } \
		case 7: { constexpr int NC = 7; CALL; break; } \
		case 8: { constexpr int NC = 8; CALL; break; } \
		default: novic_set_error("row kernels support E <= 2048"); return -22; \
	}

extern "C" int novic_rownorm_bf16(const float* x, void* y, int rows, int E, int ldy, hipStream_t stream) {
	NOVIC_CHECK(x && y, "novic_rownorm_bf16: null pointer");
	NOVIC_CHECK(E % 4 == 0 && E > 0 && ldy % 4 == 0, "novic_rownorm_bf16: E and ldy must be multiples of 4");
	if (rows <= 0) return 0;
	NOVIC_NC_DISPATCH(E, hipLaunchKernelGGL((rownorm_bf16_kernel<NC>), dim3(grid_for_rows(rows)), dim3(256), 0, stream, x, (bf16*)y, rows, E, ldy));
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* y_f32, int rows_out, int E, int seq_in, int seq_out,
                                   int seq_off, float eps, hipStream_t stream) {
	NOVIC_CHECK(x && gamma && (y_bf16 || y_f32), "novic_layernorm_fwd: null pointer");
	NOVIC_CHECK(E % 4 == 0 && E > 0, "novic_layernorm_fwd: E must be a multiple of 4");
	NOVIC_CHECK(seq_in >= 1 && seq_out >= 1 && seq_off >= 0 && seq_off + seq_out <= seq_in, "novic_layernorm_fwd: bad row-selection window");
	if (rows_out <= 0) return 0;
	NOVIC_NC_DISPATCH(E, hipLaunchKernelGGL((layernorm_fwd_kernel<NC>), dim3(grid_for_rows(rows_out)), dim3(256), 0, stream, x, gamma, beta, (bf16*)y_bf16, y_f32,
	                                        rows_out, E, seq_in, seq_out, seq_off, eps, (const int*)nullptr, (const int*)nullptr));
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_layernorm_fwd_f16(const void* x_f16, const float* gamma, const float* beta, void* y_bf16, int rows_out, int E, int seq_in, int seq_out, int seq_off, float eps,
                                       hipStream_t stream) {
	NOVIC_CHECK(x_f16 && gamma && y_bf16, "novic_layernorm_fwd_f16: null pointer");
	NOVIC_CHECK(E % 4 == 0 && E > 0, "novic_layernorm_fwd_f16: E must be a multiple of 4");
	NOVIC_CHECK(seq_in >= 1 && seq_out >= 1 && seq_off >= 0 && seq_off + seq_out <= seq_in, "novic_layernorm_fwd_f16: bad row-selection window");
	if (rows_out <= 0) return 0;
	NOVIC_NC_DISPATCH(E, hipLaunchKernelGGL((layernorm_fwd_kernel<NC, f16>), dim3(grid_for_rows(rows_out)), dim3(256), 0, stream, (const f16*)x_f16, gamma, beta, (bf16*)y_bf16,
	                                        (float*)nullptr, rows_out, E, seq_in, seq_out, seq_off, eps, (const int*)nullptr, (const int*)nullptr));
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_layernorm_fwd_rows(const float* x, const float* gamma, const float* beta, void* y_bf16, const int* src_rows, const int* row_count, int rows_max,
                                        int E, float eps, hipStream_t stream) {
	NOVIC_CHECK(x && gamma && y_bf16 && row_count, "novic_layernorm_fwd_rows: null pointer");  // src_rows may be null: rows 0 .. *row_count - 1 in place
	NOVIC_CHECK(E % 4 == 0 && E > 0, "novic_layernorm_fwd_rows: E must be a multiple of 4");
	if (rows_max <= 0) return 0;
	NOVIC_NC_DISPATCH(E, hipLaunchKernelGGL((layernorm_fwd_kernel<NC>), dim3(grid_for_rows(rows_max)), dim3(256), 0, stream, x, gamma, beta, (bf16*)y_bf16, (float*)nullptr,
	                                        rows_max, E, 1, 1, 0, eps, src_rows, row_count));
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_layernorm_bwd(const void* dy_bf16, const float* x, const float* gamma, const float* dx_in, float* dx_out, void* g_out_bf16, float* dgamma,
                                   int rows_in, int E, int seq_in, int seq_out, int seq_off, float eps, float drop_p, uint64_t seed, uint32_t drop_site,
                                   const int* dy_row, const int* row_limit, hipStream_t stream) {
	NOVIC_CHECK(dy_bf16 && x && gamma && dx_out, "novic_layernorm_bwd: null pointer");
	NOVIC_CHECK(E % 4 == 0 && E > 0, "novic_layernorm_bwd: E must be a multiple of 4");
	NOVIC_CHECK(seq_in >= 1 && seq_out >= 1 && seq_off >= 0 && seq_off + seq_out <= seq_in, "novic_layernorm_bwd: bad row-selection window");
	if (rows_in <= 0) return 0;
	DropoutDesc d = {drop_p, (uint32_t)seed, (uint32_t)(seed >> 32), drop_site};
	int grid = grid_for_rows(rows_in);
	if (grid > 1024) grid = 1024;  // bounds the dgamma atomics (E per block)
	if (dx_in) {
		NOVIC_NC_DISPATCH(E, hipLaunchKernelGGL((layernorm_bwd_kernel<NC, true>), dim3(grid), dim3(256), 0, stream, (const bf16*)dy_bf16, x, gamma, dx_in, dx_out,
		                                        (bf16*)g_out_bf16, dgamma, rows_in, E, seq_in, seq_out, seq_off, eps, d, dy_row, row_limit));
	} else {
		NOVIC_NC_DISPATCH(E, hipLaunchKernelGGL((layernorm_bwd_kernel<NC, false>), dim3(grid), dim3(256), 0, stream, (const bf16*)dy_bf16, x, gamma, dx_in, dx_out,
		                                        (bf16*)g_out_bf16, dgamma, rows_in, E, seq_in, seq_out, seq_off, eps, d, dy_row, row_limit));
	}
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_hidden_norm_act_fwd(const void* h0_bf16, const float* gamma, const float* beta, void* y_bf16, int rows, int H, int ldh, int ldy, int act, float eps,
                                         hipStream_t stream) {
	NOVIC_CHECK(h0_bf16 && gamma && y_bf16, "novic_hidden_norm_act_fwd: null pointer");
	NOVIC_CHECK(H % 4 == 0 && H > 0 && ldh % 4 == 0 && ldy % 4 == 0 && ldh >= H && ldy >= H, "novic_hidden_norm_act_fwd: H and the leading dimensions must be multiples of 4");
	NOVIC_CHECK(act == NOVIC_ACT_NONE || act == NOVIC_ACT_GELU || act == NOVIC_ACT_RELU || act == NOVIC_ACT_TANH, "novic_hidden_norm_act_fwd: activation must be none, gelu, relu or tanh");
	if (rows <= 0) return 0;
	NOVIC_NC_DISPATCH(H, hipLaunchKernelGGL((hidden_norm_act_fwd_kernel<NC>), dim3(grid_for_rows(rows)), dim3(256), 0, stream, (const bf16*)h0_bf16, gamma, beta, (bf16*)y_bf16, rows,
	                                        H, ldh, ldy, act, eps));
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_hidden_norm_act_bwd(const void* dy_bf16, const void* h0_bf16, const float* gamma, const float* beta, void* dh0_bf16, float* dgamma, float* dbeta, int rows,
                                         int H, int ldy, int ldh, int ldd, int act, float eps, hipStream_t stream) {
	NOVIC_CHECK(dy_bf16 && h0_bf16 && gamma && dh0_bf16 && dgamma, "novic_hidden_norm_act_bwd: null pointer");
	NOVIC_CHECK(H % 4 == 0 && H > 0 && ldh % 4 == 0 && ldy % 4 == 0 && ldd % 4 == 0 && ldh >= H && ldy >= H && ldd >= H,
	            "novic_hidden_norm_act_bwd: H and the leading dimensions must be multiples of 4");
	NOVIC_CHECK(act == NOVIC_ACT_NONE || act == NOVIC_ACT_GELU || act == NOVIC_ACT_RELU || act == NOVIC_ACT_TANH, "novic_hidden_norm_act_bwd: activation must be none, gelu, relu or tanh");
	if (rows <= 0) return 0;
	int grid = grid_for_rows(rows);
	if (grid > 256) grid = 256;  // bounds the dgamma / dbeta atomics (2 H per block)
	NOVIC_NC_DISPATCH(H, hipLaunchKernelGGL((hidden_norm_act_bwd_kernel<NC>), dim3(grid), dim3(256), 0, stream, (const bf16*)dy_bf16, (const bf16*)h0_bf16, gamma, beta,
	                                        (bf16*)dh0_bf16, dgamma, dbeta, rows, H, ldy, ldh, ldd, act, eps));
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_layernorm_bwd_sum(const void* dy_bf16, const float* dy_f32, const float* x, const float* gamma, float* dx_out, void* g_out_bf16, float* dgamma, float* dbeta,
                                       int rows, int E, float eps, float drop_p, uint64_t seed, uint32_t drop_site, const int* row_limit, hipStream_t stream) {
	NOVIC_CHECK((dy_bf16 || dy_f32) && x && gamma && dx_out, "novic_layernorm_bwd_sum: null pointer");
	NOVIC_CHECK(E % 4 == 0 && E > 0, "novic_layernorm_bwd_sum: E must be a multiple of 4");
	if (rows <= 0) return 0;
	DropoutDesc d = {drop_p, (uint32_t)seed, (uint32_t)(seed >> 32), drop_site};
	int grid = grid_for_rows(rows);
	if (grid > 512) grid = 512;  // bounds the dgamma / dbeta atomics (2 E per block)
	NOVIC_NC_DISPATCH(E, hipLaunchKernelGGL((layernorm_bwd_sum_kernel<NC>), dim3(grid), dim3(256), 0, stream, (const bf16*)dy_bf16, dy_f32, x, gamma, dx_out, (bf16*)g_out_bf16,
	                                        dgamma, dbeta, rows, E, eps, d, row_limit));
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_add_bf16(float* dst, const void* src_bf16, uint64_t n, hipStream_t stream) {
	NOVIC_CHECK(dst && src_bf16, "novic_add_bf16: null pointer");
	NOVIC_CHECK(n % 4 == 0, "novic_add_bf16: n must be a multiple of 4");
	if (n == 0) return 0;
	const uint64_t n4 = n / 4;
	const int grid = (int)(n4 / 256 + 1 < 2048 ? n4 / 256 + 1 : 2048);
	hipLaunchKernelGGL(add_bf16_kernel, dim3(grid), dim3(256), 0, stream, dst, (const bf16*)src_bf16, (size_t)n4);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_rezero_fwd(const float* resid, const void* branch_bf16, const float* scale, float* out, int rows, int E, const int* row_limit, hipStream_t stream) {
	NOVIC_CHECK(resid && branch_bf16 && scale && out, "novic_rezero_fwd: null pointer");
	NOVIC_CHECK(E % 4 == 0 && E > 0, "novic_rezero_fwd: E must be a multiple of 4");
	if (rows <= 0) return 0;
	const uint64_t n4 = (uint64_t)rows * E / 4;
	const int grid = (int)(n4 / 256 + 1 < 2048 ? n4 / 256 + 1 : 2048);
	hipLaunchKernelGGL(rezero_fwd_kernel, dim3(grid), dim3(256), 0, stream, resid, (const bf16*)branch_bf16, scale, out, (size_t)n4, E / 4, row_limit);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_rezero_bwd(const float* dx, const void* branch_bf16, const float* scale, float* dscale, void* g_out_bf16, int rows, int E, float drop_p, uint64_t seed,
                                uint32_t drop_site, const int* row_limit, hipStream_t stream) {
	NOVIC_CHECK(dx && branch_bf16 && scale && dscale && g_out_bf16, "novic_rezero_bwd: null pointer");
	NOVIC_CHECK(E % 4 == 0 && E > 0, "novic_rezero_bwd: E must be a multiple of 4");
	if (rows <= 0) return 0;
	DropoutDesc d = {drop_p, (uint32_t)seed, (uint32_t)(seed >> 32), drop_site};
	const uint64_t n4 = (uint64_t)rows * E / 4;
	const int grid = (int)(n4 / 256 + 1 < 1024 ? n4 / 256 + 1 : 1024);
	hipLaunchKernelGGL(rezero_bwd_kernel, dim3(grid), dim3(256), 0, stream, dx, (const bf16*)branch_bf16, scale, dscale, (bf16*)g_out_bf16, (size_t)n4, E / 4, d, row_limit);
	NOVIC_LAUNCH_CHECK();
	return 0;
}
