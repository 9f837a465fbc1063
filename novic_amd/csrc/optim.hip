// Optimizer-side kernels over the flat parameter buffer: global gradient norm, clip + decoupled AdamW + bf16 shadow refresh.
// reference: train.py:1280-1286 (clip_grad_norm_ to a global 2-norm, then step), train.py:1103-1119 (AdamW, weight decay on >= 2-D params only).
// HBM-bound: 28 bytes per parameter (+2 for the bf16 shadow the GEMMs read) -- read p,g,m,v, write p,m,v,w16.
#include "common.hpp"
#include "novic_hip.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, size_t n, double* __restrict__ partial) {
	__shared__ double red[4];
	double s = 0;
	const size_t n4 = n / 4;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
		const f32x4 t = reinterpret_cast<const f32x4*>(g)[i];
		s += (double)(t[0] * t[0] + t[1] * t[1]) + (double)(t[2] * t[2] + t[3] * t[3]);
	}
	if (blockIdx.x == 0)
		for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) s += (double)g[i] * g[i];
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void sumsq_final_kernel(const double* __restrict__ partial, int n, float* __restrict__ out_norm) {
	__shared__ double red[4];
	double s = 0;
	for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) out_norm[0] = (float)sqrt(red[0] + red[1] + red[2] + red[3]);
}

// the hyper-parameters travel in the kernel arguments (by value): whatever the host does to its copy after the launch call cannot reach this launch
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    bf16* __restrict__ w16, size_t n, size_t n_decay, const novic_adamw_hyper_t hyper,
                                                    const float* __restrict__ grad_norm) {
	const float lr = hyper.lr, b1 = hyper.beta1, b2 = hyper.beta2, eps = hyper.eps, wd = hyper.weight_decay, bc1 = hyper.bias_corr1, bc2 = hyper.bias_corr2,
	            max_norm = hyper.max_norm;
	float coef = 1.f;
	if (grad_norm && max_norm > 0.f) coef = fminf(1.f, max_norm / (grad_norm[0] + 1e-6f));
	const float rs2 = rsqrtf(bc2), step = lr / bc1;
	for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) {
		if (i + 4 <= n) {
			f32x4 pp = *reinterpret_cast<f32x4*>(p + i);
			const f32x4 gg = *reinterpret_cast<const f32x4*>(g + i);
			f32x4 mm = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
			bf16x4 ww;
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const float gk = gg[k] * coef;
				float x = pp[k];
				if (i + k < n_decay) x *= (1.f - lr * wd);
				mm[k] = b1 * mm[k] + (1.f - b1) * gk;
				vv[k] = b2 * vv[k] + (1.f - b2) * gk * gk;
				x -= step * mm[k] / (sqrtf(vv[k]) * rs2 + eps);
				pp[k] = x;
				ww[k] = (bf16)x;
			}
			*reinterpret_cast<f32x4*>(p + i) = pp;
			*reinterpret_cast<f32x4*>(m + i) = mm;
			*reinterpret_cast<f32x4*>(v + i) = vv;
			if (w16) *reinterpret_cast<bf16x4*>(w16 + i) = ww;
		} else {
			for (size_t k = i; k < n; ++k) {
				const float gk = g[k] * coef;
				float x = p[k];
				if (k < n_decay) x *= (1.f - lr * wd);
				m[k] = b1 * m[k] + (1.f - b1) * gk;
				v[k] = b2 * v[k] + (1.f - b2) * gk * gk;
				x -= step * m[k] / (sqrtf(v[k]) * rs2 + eps);
				p[k] = x;
				if (w16) w16[k] = (bf16)x;
			}
		}
	}
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ y, size_t n) {
	for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) {
		if (i + 4 <= n) {
			const f32x4 t = *reinterpret_cast<const f32x4*>(x + i);
			bf16x4 o = {(bf16)t[0], (bf16)t[1], (bf16)t[2], (bf16)t[3]};
			*reinterpret_cast<bf16x4*>(y + i) = o;
		} else {
			for (size_t k = i; k < n; ++k) y[k] = (bf16)x[k];
		}
	}
}

// Transposed bf16 copies of up to 32 matrices in one launch (the weight shadows W^T [cols][rows] that turn the input-gradient GEMMs dX = dY W
// into K-contiguous x K-contiguous problems for the 256-wide LDS-DMA kernel).  One 64 x 64 tile per workgroup through LDS: 16-byte reads along
// the source rows, 16-byte writes along the destination rows.
struct TransposeBatch {
	long long src_off[32], dst_off[32];
	int rows[32], cols[32], dst_ld[32];
};
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, const TransposeBatch b) {
	__shared__ unsigned short tile[64][72];
	const int i = blockIdx.y, R = b.rows[i], C = b.cols[i], LD = b.dst_ld[i];
	const int tiles_c = (C + 63) / 64, ntiles = ((R + 63) / 64) * tiles_c;
	const unsigned short* S = reinterpret_cast<const unsigned short*>(src) + b.src_off[i];
	unsigned short* D = reinterpret_cast<unsigned short*>(dst) + b.dst_off[i];
	typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
	const bool vsrc = (C & 7) == 0 && (b.src_off[i] & 7) == 0, vdst = (LD & 7) == 0 && (b.dst_off[i] & 7) == 0;  // 16-byte accesses where the rows allow them
	for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
		const int r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
		const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
#pragma unroll
		for (int pass = 0; pass < 2; ++pass) {
			const int r = r0 + ty + pass * 32, c = c0 + tx * 8;
			u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
			if (vsrc && r < R && c + 8 <= C) v = *reinterpret_cast<const u16x8*>(S + (size_t)r * C + c);
			else if (r < R)
				for (int k = 0; k < 8; ++k) v[k] = c + k < C ? S[(size_t)r * C + c + k] : (unsigned short)0;
#pragma unroll
			for (int k = 0; k < 8; ++k) tile[ty + pass * 32][tx * 8 + k] = v[k];
		}
		__syncthreads();
#pragma unroll
		for (int pass = 0; pass < 2; ++pass) {
			const int c = c0 + ty + pass * 32, r = r0 + tx * 8;  // destination row c, columns r .. r+7
			u16x8 v;
#pragma unroll
			for (int k = 0; k < 8; ++k) v[k] = tile[tx * 8 + k][ty + pass * 32];
			if (vdst && c < C && r + 8 <= R) *reinterpret_cast<u16x8*>(D + (size_t)c * LD + r) = v;
			else if (c < C)
				for (int k = 0; k < 8; ++k)
					if (r + k < R) D[(size_t)c * LD + r + k] = v[k];
		}
		__syncthreads();
	}
}

}  // namespace

extern "C" int novic_transpose_bf16_batched(const void* src_base, void* dst_base, const long long* desc, int n, hipStream_t stream) {
	NOVIC_CHECK(src_base && dst_base && (desc || n == 0), "novic_transpose_bf16_batched: null pointer");
	NOVIC_CHECK(n >= 0, "novic_transpose_bf16_batched: negative count");
	for (int base = 0; base < n; base += 32) {
		TransposeBatch b;
		const int m = n - base < 32 ? n - base : 32;
		int max_tiles = 1;
		for (int i = 0; i < m; ++i) {
			const long long* d = desc + (size_t)(base + i) * 5;
			NOVIC_CHECK(d[0] >= 0 && d[1] >= 0 && d[2] >= 1 && d[3] >= 1 && d[2] < (1ll << 30) && d[3] < (1ll << 30) && d[4] >= d[2] && d[4] < (1ll << 30),
			            "novic_transpose_bf16_batched: bad descriptor");
			NOVIC_CHECK(((uintptr_t)src_base & 15) == 0 && ((uintptr_t)dst_base & 15) == 0, "novic_transpose_bf16_batched: bases must be 16-byte aligned");
			b.src_off[i] = d[0]; b.dst_off[i] = d[1]; b.rows[i] = (int)d[2]; b.cols[i] = (int)d[3]; b.dst_ld[i] = (int)d[4];
			const int t = (int)(((d[2] + 63) / 64) * ((d[3] + 63) / 64));
			if (t > max_tiles) max_tiles = t;
		}
		if (max_tiles > 1024) max_tiles = 1024;
		hipLaunchKernelGGL(transpose_bf16_kernel, dim3(max_tiles, m), dim3(256), 0, stream, (const bf16*)src_base, (bf16*)dst_base, b);
		NOVIC_LAUNCH_CHECK();
	}
	return 0;
}

extern "C" int novic_grad_norm(const float* grads, uint64_t n, double* partial_ws, int ws_len, float* out_norm, hipStream_t stream) {
	NOVIC_CHECK(grads && partial_ws && out_norm, "novic_grad_norm: null pointer");
	NOVIC_CHECK(ws_len >= 1 && ((uintptr_t)grads & 15) == 0, "novic_grad_norm: need a 16-byte aligned buffer and a workspace");
	int blocks = (int)((n / 4 + 255) / 256);
	if (blocks < 1) blocks = 1;
	if (blocks > ws_len) blocks = ws_len;
	if (blocks > 1024) blocks = 1024;
	hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, stream, grads, (size_t)n, partial_ws);
	hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, stream, partial_ws, blocks, out_norm);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, uint64_t n, uint64_t n_decay,
                                const novic_adamw_hyper_t* hyper, const float* grad_norm, hipStream_t stream) {
	NOVIC_CHECK(params && grads && exp_avg && exp_avg_sq && hyper, "novic_adamw_step: null pointer");
	NOVIC_CHECK(hyper->bias_corr1 > 0.f && hyper->bias_corr2 > 0.f, "novic_adamw_step: bias corrections must be positive (step counts from 1)");
	NOVIC_CHECK((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0, "novic_adamw_step: buffers must be 16-byte aligned");
	NOVIC_CHECK(n_decay <= n, "novic_adamw_step: n_decay > n");
	if (n == 0) return 0;
	int blocks = (int)((n / 4 + 255) / 256);
	if (blocks < 1) blocks = 1;
	if (blocks > 4096) blocks = 4096;
	hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, stream, params, grads, exp_avg, exp_avg_sq, (bf16*)shadow_bf16, (size_t)n, (size_t)n_decay, *hyper,
	                   grad_norm);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Column sums of a bf16 matrix: out[c] += alpha * sum over rows r < min(rows, *row_limit) of x[r][c].  The bias gradient of a linear layer (autograd of
// `nn.Linear(bias=True)`: grad_bias = grad_output.sum(0); reference embedding_decoder.py:239 logits_bias).  HBM-bound, one read of x.  Deterministic: a workgroup
// sums a slab of rows for a strip of 512 columns (a wave = rows w, w + 4, ..., a lane = 8 neighbouring columns = one 16-byte load), the four waves are added
// through LDS in wave order, the slabs' partial sums go to `ws` [slabs][cols] and a second launch adds them in slab order.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16* __restrict__ x, int rows, int cols, int ld, const int* __restrict__ row_limit, float* __restrict__ ws, int slabs) {
	__shared__ float red[4][512];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	if (row_limit) rows = min(rows, max(*row_limit, 0));
	const int strip = blockIdx.x, slab = blockIdx.y;
	const int c0 = strip * 512 + lane * 8;
	const int per = (rows + slabs - 1) / slabs, r0 = slab * per, r1 = min(rows, r0 + per);
	float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	if (c0 + 8 <= cols) {
		for (int r = r0 + w; r < r1; r += 4) {
			const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (size_t)r * ld + c0);
#pragma unroll
			for (int i = 0; i < 8; ++i) acc[i] += (float)v[i];
		}
	} else {
		for (int r = r0 + w; r < r1; r += 4)
#pragma unroll
			for (int i = 0; i < 8; ++i)
				if (c0 + i < cols) acc[i] += (float)x[(size_t)r * ld + c0 + i];
	}
#pragma unroll
	for (int i = 0; i < 8; ++i) red[w][lane * 8 + i] = acc[i];
	__syncthreads();
	for (int c = threadIdx.x; c < 512; c += 256) {
		const int col = strip * 512 + c;
		if (col < cols) ws[(size_t)slab * cols + col] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
	}
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ ws, int cols, int slabs, float alpha, float* __restrict__ out) {
	const int c = blockIdx.x * 256 + threadIdx.x;
	if (c >= cols) return;
	float s = 0.f;
	for (int k = 0; k < slabs; ++k) s += ws[(size_t)k * cols + c];
	out[c] += alpha * s;
}

extern "C" int novic_colsum_bf16(const void* x_bf16, int rows, int cols, int ld, const int32_t* row_limit, float* out, float alpha, float* ws, uint64_t ws_bytes,
                                 hipStream_t stream) {
	NOVIC_CHECK(x_bf16 && out && ws, "novic_colsum_bf16: null pointer");
	NOVIC_CHECK(rows >= 0 && cols >= 1 && ld >= cols && ld % 8 == 0 && (((uintptr_t)x_bf16) & 15) == 0, "novic_colsum_bf16: rows of x must be 16-byte aligned (ld a multiple of 8)");
	if (rows == 0) return 0;
	const int strips = (cols + 511) / 512;
	int slabs = 256 / strips;  // one round of the chip
	if (slabs < 1) slabs = 1;
	if (slabs > (rows + 63) / 64) slabs = (rows + 63) / 64;
	NOVIC_CHECK((uint64_t)slabs * (uint64_t)cols * 4ull <= ws_bytes, "novic_colsum_bf16: scratch too small (slabs x cols floats; 256 x cols / ceil(cols / 512) covers every case)");
	hipLaunchKernelGGL(colsum_partial_kernel, dim3(strips, slabs), dim3(256), 0, stream, (const bf16*)x_bf16, rows, cols, ld, row_limit, ws, slabs);
	hipLaunchKernelGGL(colsum_final_kernel, dim3((cols + 255) / 256), dim3(256), 0, stream, ws, cols, slabs, alpha, out);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_cast_bf16(const float* x, void* y_bf16, uint64_t n, hipStream_t stream) {
	NOVIC_CHECK(x && y_bf16, "novic_cast_bf16: null pointer");
	NOVIC_CHECK((((uintptr_t)x) & 15) == 0 && (((uintptr_t)y_bf16) & 7) == 0, "novic_cast_bf16: misaligned buffer");
	if (n == 0) return 0;
	int blocks = (int)((n / 4 + 255) / 256);
	if (blocks < 1) blocks = 1;
	if (blocks > 4096) blocks = 4096;
	hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks), dim3(256), 0, stream, x, (bf16*)y_bf16, (size_t)n);
	NOVIC_LAUNCH_CHECK();
	return 0;
}
