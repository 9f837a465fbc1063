// CLIP ViT image tower pieces that are not plain GEMM / LayerNorm launches: patch extraction (im2col), token assembly + ln_pre,
// and non-causal multi-head attention over N = 50 / 257 / 730 tokens.
//
// reference call sites: embedders.py:593-594, :763-764, :906-907 (`model.encode_image` of open_clip / clip / transformers -- third-party
// packages absent from the reference tree; architecture restated from their public model definitions: conv1 (stride = patch, no bias)
// -> [cls; patches] + positional embedding -> ln_pre -> L x [x + attn(ln_1 x); x + mlp(ln_2 x)] -> ln_post(cls) -> @ proj).
//
// Attention: one workgroup per (image, head, 64-query slab); K/V stream through LDS in 32-key chunks with an online softmax.
// Same register choreography as the decoder kernel: swapped QK^T (lane = query column), P fed back as the MFMA B operand with the
// permuted k index, V read "down the rows" with ds_read_b64_tr_b16.  HBM traffic = qkv read once per 64-query slab (K/V re-read
// ceil(N/64) times out of L2), o written once; MFMA-bound only for N >= 257.
#include <type_traits>

#include "common.hpp"
#include "novic_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// im2col: images [B][3][R][R] f32 -> patches [B*g*g][Kp] bf16, k = c*p*p + y*p + x (the layout of conv1.weight.view(W, -1)), zero pad to Kp
// U8 form (novic_vit_im2col_u8): the images are the transform's uint8 pixels BEFORE ToTensor / Normalize, and the thread that moves a pixel applies
// (float(u) / 255 - mean[c]) / std[c] in fp32 -- the two IEEE divisions and the subtraction of torchvision's ToTensor + Normalize, in their order -- so the
// bf16 patch matrix is bit-identical to the one the fp32 images give, with a quarter of the bytes over PCIe and out of HBM.
// ---------------------------------------------------------------------------------------------------------
struct PixNorm { float mean[3], std[3]; };
__device__ __forceinline__ float pix_value(float v, const PixNorm&, int) { return v; }
__device__ __forceinline__ float pix_value(uint8_t u, const PixNorm& nm, int c) { return __fdiv_rn(__fsub_rn(__fdiv_rn((float)u, 255.0f), nm.mean[c]), nm.std[c]); }

template <typename PIX>
__global__ __launch_bounds__(256) void im2col_kernel(const PIX* __restrict__ img, bf16* __restrict__ out, int B, int R, int p, int g, int Kp, const PixNorm nm) {
	const int K = 3 * p * p;
	const size_t total = (size_t)B * g * g * (Kp / 4);
	for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
		const int k4 = (int)(idx % (Kp / 4)) * 4;
		const size_t row = idx / (Kp / 4);
		const int b = (int)(row / (g * g)), pi = (int)(row % (g * g));
		const int py = pi / g, px = pi % g;
		float v[4];
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int k = k4 + i;
			if (k < K) {
				const int c = k / (p * p), rem = k % (p * p), y = rem / p, x = rem % p;
				v[i] = pix_value(img[(((size_t)b * 3 + c) * R + (py * p + y)) * R + px * p + x], nm, c);
			} else {
				v[i] = 0.f;
			}
		}
		bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
		*reinterpret_cast<bf16x4*>(out + row * Kp + k4) = o;
	}
}

// patch sizes that are multiples of 4 (32, 16): a thread moves 4 consecutive pixels of an image row -- one 16-byte (uint8: 4-byte) load, reads in image order (fully
// coalesced), 32-bit index arithmetic; the patch rows come out as p/4 neighbouring 8-byte stores.  85 -> ~45 us for 256 images of 224^2 (231 MB).
template <typename PIX>
__global__ __launch_bounds__(256) void im2col_vec4_kernel(const PIX* __restrict__ img, bf16* __restrict__ out, int B, int R, int p, int g, int Kp, const PixNorm nm) {
	const int K = 3 * p * p, r4 = R / 4;
	const unsigned total = (unsigned)B * 3u * (unsigned)R * (unsigned)r4;
	for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
		const unsigned x4 = idx % r4, t = idx / r4;      // t = (b*3 + c)*R + Y
		const unsigned Y = t % R, bc = t / R;
		const unsigned c = bc % 3u, b = bc / 3u;
		const unsigned X = x4 * 4u;
		float v[4];
		if constexpr (std::is_same<PIX, float>::value) {
			const f32x4 q = *reinterpret_cast<const f32x4*>(img + (size_t)idx * 4);
			v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
		} else {
			const unsigned q = *reinterpret_cast<const unsigned*>(img + (size_t)idx * 4);
#pragma unroll
			for (int i = 0; i < 4; ++i) v[i] = pix_value((uint8_t)((q >> (8 * i)) & 0xffu), nm, (int)c);
		}
		const unsigned row = (b * g + Y / p) * g + X / p;
		const unsigned k = c * p * p + (Y % p) * p + X % p;
		bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
		*reinterpret_cast<bf16x4*>(out + (size_t)row * Kp + k) = o;
	}
	if (Kp > K) {  // zero padding of the patch rows
		const unsigned pad4 = (Kp - K) / 4, rows = (unsigned)B * g * g;
		for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < rows * pad4; idx += gridDim.x * 256u) {
			const unsigned row = idx / pad4, j = idx % pad4;
			*reinterpret_cast<bf16x4*>(out + (size_t)row * Kp + K + j * 4) = (bf16x4){(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// x[b][t] = ln_pre( (t == 0 ? cls : patch[b][t-1]) + pos[t] )     one wave per token row, W <= 2048
// ---------------------------------------------------------------------------------------------------------
// XT: float, or f16 -- the residual stream of a half-precision tower (novic_vit_embed_f16); the sum and ln_pre in fp32 either way, rounded once on the way out
template <int NC, typename XT = float>
__global__ __launch_bounds__(256) void vit_embed_kernel(const bf16* __restrict__ patches, const float* __restrict__ cls, const float* __restrict__ pos,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, XT* __restrict__ x, int B, int N, int W, float eps,
                                                        int has_ln) {
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (int row = blockIdx.x * 4 + w; row < B * N; row += gridDim.x * 4) {
		const int b = row / N, t = row - b * N;
		float v[NC][4];
		float s = 0.f;
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int e = c * 256 + lane * 4;
			if (e < W) {
				const f32x4 pe = *reinterpret_cast<const f32x4*>(pos + (size_t)t * W + e);
				if (t == 0) {
					const f32x4 ce = *reinterpret_cast<const f32x4*>(cls + e);
#pragma unroll
					for (int i = 0; i < 4; ++i) v[c][i] = ce[i] + pe[i];
				} else {
					const bf16x4 pv = *reinterpret_cast<const bf16x4*>(patches + ((size_t)b * (N - 1) + (t - 1)) * W + e);
#pragma unroll
					for (int i = 0; i < 4; ++i) v[c][i] = (float)pv[i] + pe[i];
				}
#pragma unroll
				for (int i = 0; i < 4; ++i) s += v[c][i];
			} else {
				v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f;
			}
		}
		float mean = 0.f, rstd = 1.f;
		if (has_ln) {
			mean = wave_sum(s) / (float)W;
			float q = 0.f;
#pragma unroll
			for (int c = 0; c < NC; ++c) {
				const int e = c * 256 + lane * 4;
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const float d = (e < W) ? v[c][i] - mean : 0.f;
					q += d * d;
				}
			}
			rstd = rsqrtf(wave_sum(q) / (float)W + eps);
		}
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int e = c * 256 + lane * 4;
			if (e < W) {
				float o[4];
				if (has_ln) {
					const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + e);
					const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + e);
#pragma unroll
					for (int i = 0; i < 4; ++i) o[i] = (v[c][i] - mean) * rstd * gm[i] + bt[i];
				} else {
#pragma unroll
					for (int i = 0; i < 4; ++i) o[i] = v[c][i];
				}
				if constexpr (sizeof(XT) == 2) *reinterpret_cast<f16x4*>(x + (size_t)row * W + e) = (f16x4){(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]};
				else *reinterpret_cast<f32x4*>(x + (size_t)row * W + e) = (f32x4){o[0], o[1], o[2], o[3]};
			}
		}
	}
}

// y(f32)[r] = x[r] / max(||x[r]||, 1e-12)
template <int NC>
__global__ __launch_bounds__(256) void rownorm_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int E) {
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (int row = blockIdx.x * 4 + w; row < rows; row += gridDim.x * 4) {
		float v[NC][4];
		float ss = 0.f;
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int e = c * 256 + lane * 4;
			if (e < E) {
				const f32x4 t = *reinterpret_cast<const f32x4*>(x + (size_t)row * E + e);
#pragma unroll
				for (int i = 0; i < 4; ++i) { v[c][i] = t[i]; ss += t[i] * t[i]; }
			} else {
				v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f;
			}
		}
		const float inv = 1.f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int e = c * 256 + lane * 4;
			if (e < E) *reinterpret_cast<f32x4*>(y + (size_t)row * E + e) = (f32x4){v[c][0] * inv, v[c][1] * inv, v[c][2] * inv, v[c][3] * inv};
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// attention
// ---------------------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ int voff(int row, int col) {  // LDS byte offset in a [rows][D] bf16 tile; 16-byte chunks XOR-swizzled when D/8 is a power of two
	constexpr int CPR = D / 8;
	constexpr bool POW2 = (CPR & (CPR - 1)) == 0;
	const int ch = POW2 ? ((col >> 3) ^ (row & (CPR - 1) & 7)) : (col >> 3);
	return row * (D * 2) + (ch << 4) + ((col & 7) << 1);
}

// the same for a fragment that may start at a column >= D (head_dim 80, k-step 2: columns 64 .. 95): rows are contiguous there (no swizzle when D / 8 is not a
// power of two), so the address runs on into the next row -- the caller multiplies those elements by zeros
template <int D>
__device__ __forceinline__ int voff_lin(int row, int col) {
	constexpr int CPR = D / 8;
	constexpr bool POW2 = (CPR & (CPR - 1)) == 0;
	if constexpr (POW2) return voff<D>(row, col);
	else return row * (D * 2) + (col << 1);
}

template <int D>
__global__ __launch_bounds__(256) void vit_attn_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o, int B, int N, int H, float scale, int causal) {
	constexpr int KS = (D + 31) / 32, DT = D / 16, CPR = D / 8, KC = 32;  // 32 keys per chunk
	__shared__ __attribute__((aligned(16))) char sk[KC * D * 2];
	__shared__ __attribute__((aligned(16))) char sv[KC * D * 2];
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4;
	const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
	const int W = H * D;
	const int q0 = blockIdx.y * 64 + w * 16;
	const int qi = q0 + (lane & 15);
	const bf16* base = qkv + (size_t)b * N * 3 * W + h * D;

	// Q fragments straight from global: lane -> Q[q0 + (l&15)][ks*32 + 8g .. +7]
	bf16x8 qf[KS];
#pragma unroll
	for (int ks = 0; ks < KS; ++ks) {
		const int col = ks * 32 + 8 * g;
		bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
		qf[ks] = z;
		if (col < D) {
			const int row = qi < N ? qi : N - 1;
			qf[ks] = *reinterpret_cast<const bf16x8*>(base + (size_t)row * 3 * W + col);
		}
	}
	f32x4 acc[DT];
#pragma unroll
	for (int dt = 0; dt < DT; ++dt) acc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
	float m_run = -1e30f, l_run = 0.f;

	// causal (text tower): keys after the workgroup's last query never matter -> stop there (workgroup-uniform bound)
	const int kend = causal ? min(N, (int)blockIdx.y * 64 + 64) : N;
	for (int k0 = 0; k0 < kend; k0 += KC) {
		__syncthreads();  // previous chunk fully consumed
		for (int c = tid; c < KC * CPR; c += 256) {
			const int row = c / CPR, ch = c - row * CPR;
			uint4 kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
			if (k0 + row < N) {
				const bf16* src = base + (size_t)(k0 + row) * 3 * W + ch * 8;
				kv = *reinterpret_cast<const uint4*>(src + W);
				vv = *reinterpret_cast<const uint4*>(src + 2 * W);
			}
			*reinterpret_cast<uint4*>(sk + voff<D>(row, ch * 8)) = kv;
			*reinterpret_cast<uint4*>(sv + voff<D>(row, ch * 8)) = vv;
		}
		__syncthreads();
		float p[2][4];
		float mx = m_run;
#pragma unroll
		for (int kt = 0; kt < 2; ++kt) {
			f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
			for (int ks = 0; ks < KS; ++ks) {
				const int col = ks * 32 + 8 * g;
				bf16x8 kf = {0, 0, 0, 0, 0, 0, 0, 0};
				if (col < D) kf = *reinterpret_cast<const bf16x8*>(sk + voff<D>(kt * 16 + (lane & 15), col));
				s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s, 0, 0, 0);
			}
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const int j = k0 + kt * 16 + 4 * g + r;
				p[kt][r] = (j < N && (!causal || j <= qi)) ? s[r] * scale : -1e30f;
				mx = fmaxf(mx, p[kt][r]);
			}
		}
		mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
		mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
		const float alpha = __expf(m_run - mx);
		float sum = 0.f;
		bf16x8 pf;
#pragma unroll
		for (int kt = 0; kt < 2; ++kt)
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const float e = (p[kt][r] > -1e29f) ? __expf(p[kt][r] - mx) : 0.f;
				sum += e;
				pf[kt * 4 + r] = (bf16)e;
			}
		sum += __shfl_xor(sum, 16, 64);
		sum += __shfl_xor(sum, 32, 64);
		l_run = l_run * alpha + sum;
		m_run = mx;
		typedef bf16x4 __attribute__((address_space(3))) * lds4_t;
		const int q = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
		for (int dt = 0; dt < DT; ++dt) {
			const int col = dt * 16 + 4 * pp;
			bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(sv + voff<D>(4 * g + q, col)));
			bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(sv + voff<D>(16 + 4 * g + q, col)));
			bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
			acc[dt] = acc[dt] * alpha;
			acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, acc[dt], 0, 0, 0);
		}
	}
	if (qi < N) {
		const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
#pragma unroll
		for (int dt = 0; dt < DT; ++dt) {
			bf16x4 ov = {(bf16)(acc[dt][0] * inv), (bf16)(acc[dt][1] * inv), (bf16)(acc[dt][2] * inv), (bf16)(acc[dt][3] * inv)};
			*reinterpret_cast<bf16x4*>(o + ((size_t)b * N + qi) * W + h * D + dt * 16 + 4 * g) = ov;
		}
	}
}

// Attention with the head's WHOLE K and V resident in LDS and an EXACT (two-pass) soft-max, for N <= 16 NKT keys (ViT-L/14, H/14: N = 257; B/16: 197;
// the text tower: 77): one workgroup per (image, head) stages K and V once, its four waves then walk the 16-query tiles without further
// synchronisation.  Per tile: all score tiles first (2 NKT independent MFMAs back to back, raw scores stay in registers), ONE row maximum and ONE
// sum for the whole row (two shuffles each), probabilities exp2((s - max) * scale * log2 e) rounded to bf16 in the registers the scores came from,
// then the PV MFMAs back to back.  The streaming kernel above runs a dependent chain QK -> max -> shuffles -> exp -> sum -> rescale -> PV per
// 32-key chunk with two waves per SIMD to hide it: 345 us per layer for ViT-L/14 at batch 256 (and the same with K/V resident but the online
// soft-max kept: the chain, not the staging, was the cost).  Not bit-identical to the streaming kernel: probabilities are rounded to bf16
// relative to the final row maximum instead of the running one (both within 2^-8 of the fp32 soft-max).
template <int D, int NKT, bool CAUSAL, int NW>
__global__ __launch_bounds__(NW * 64, 2) void vit_attn_full_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o, int B, int N, int H, float scale) {
	constexpr int KS = (D + 31) / 32, DT = D / 16, CPR = D / 8;
	constexpr bool POW2 = (CPR & (CPR - 1)) == 0;
	static_assert(NKT % 2 == 0, "key tiles come in pairs (32-key PV chunks)");
	constexpr int NP = NKT * 16;
	static_assert((NP * D * 2) % 1024 == 0, "the K / V images are whole LDS-DMA instructions");
	extern __shared__ __attribute__((aligned(16))) char smem_attn[];  // K [NP][D] | V [NP][D], NP = 16 NKT >= N rows (padding rows zero)
	char* sk = smem_attn;
	char* sv = smem_attn + (size_t)NP * D * 2;
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4;
	const int bh = xcd_major_id(blockIdx.x, gridDim.x), b = bh / H, h = bh - b * H;
	const int W = H * D;
	const bf16* base = qkv + (size_t)b * N * 3 * W + h * D;
	{
		// K and V of the head by LDS-DMA, all of it in flight at once (18 instructions per wave at N = 257): an instruction writes 1 KiB of the
		// image lane-linearly, so lane L of instruction i fills 16-byte slot c = 64 i + L -- row c / CPR, and it fetches the chunk that the
		// read-side XOR swizzle (voff) expects there.  Rows >= N: an out-of-range offset, which the buffer load returns as zeros.
		typedef __attribute__((address_space(3))) void* lds_ptr_t;
		const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(base), 0, (int)(((size_t)N * 3 * W - h * D) * 2), 0x00020000);
		constexpr int NI = NP * D * 2 / 1024;  // instructions per operand
#pragma unroll
		for (int i0 = 0; i0 < (2 * NI + NW - 1) / NW; ++i0) {
			const int i = i0 * NW + w;  // wave-uniform
			if (i < 2 * NI) {
				const int isv = i >= NI ? 1 : 0, ii = i - isv * NI;
				const int c = ii * 64 + lane, row = c / CPR, slot = c - row * CPR;
				const int ch = POW2 ? (slot ^ (row & (CPR - 1) & 7)) : slot;
				const unsigned off = row < N ? (unsigned)(((size_t)row * 3 * W + (1 + isv) * W + ch * 8) * 2) : 0xFFFFFFF0u;
				__builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem_attn + (size_t)i * 1024), 16, off, 0, 0, 0);
			}
		}
	}
	const int ntiles = (N + 15) / 16;
	auto load_q = [&](int qt, bf16x8 (&qf)[KS]) {
		const int qrow = min(qt * 16 + (lane & 15), N - 1);
#pragma unroll
		for (int ks = 0; ks < KS; ++ks) {
			const int col = ks * 32 + 8 * g;
			bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
			qf[ks] = z;
			if (col < D) qf[ks] = *reinterpret_cast<const bf16x8*>(base + (size_t)qrow * 3 * W + col);
		}
	};
	bf16x8 qf[KS], qn[KS];
	if (w < ntiles) load_q(w, qf);
#pragma unroll
	for (int ks = 0; ks < KS; ++ks) qn[ks] = qf[ks];
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of K / V has landed ...
	__syncthreads();                                   // ... and so has everybody else's: the only barrier
	const float c2 = scale * 1.4426950408889634f;  // exp(x * scale) = exp2(x * c2)
	const int q = (lane >> 2) & 3, pp = lane & 3;
	typedef bf16x4 __attribute__((address_space(3))) * lds4_t;
	const bf16x8 ones8 = {(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f};
	for (int qt = w; qt < ntiles; qt += NW) {
		if (qt + NW < ntiles) load_q(qt + NW, qn);  // the next tile's queries fly under this tile's work
		const int qi = qt * 16 + (lane & 15);
		// pass 1: raw scores s[kt][r] = q(lane & 15) . k(kt*16 + 4g + r); key rows >= N are zero in LDS
		f32x4 s[NKT];
#pragma unroll
		for (int kt = 0; kt < NKT; ++kt) {
			s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
			for (int ks = 0; ks < KS; ++ks) {
				// (columns >= D -- head_dim 80, k-step 2 -- are read all the same, the bytes behind the row inside the K | V image, and meet the query fragment's zeros: voff_lin)
				const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sk + voff_lin<D>(kt * 16 + (lane & 15), ks * 32 + 8 * g));
				s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[kt], 0, 0, 0);
			}
		}
		// mask: only the key tiles that reach past N, or (causal) past the tile's first query, test anything (wave-uniform); a masked score is
		// -1e30, whose exp2 below is exactly 0.  Then the row maximum.
		float mx = -1e30f;
#pragma unroll
		for (int kt = 0; kt < NKT; ++kt) {
			if ((kt + 1) * 16 > N || (CAUSAL && (kt + 1) * 16 > qt * 16)) {
				asm volatile("" ::: "memory");  // keeps this a real wave-uniform branch: if-converted, all 4 NKT scores get two compares and a select each
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const int j = kt * 16 + 4 * g + r;
					if (!(j < N && (!CAUSAL || j <= qi))) s[kt][r] = -1e30f;
				}
			}
#pragma unroll
			for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
		}
		mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
		mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
		const float mc = mx * c2;
		// probabilities (bf16, in chunk order for the PV operand).  The row sum is not added up on the VALU (72 adds and two shuffles per tile of a kernel that the VALU
		// bounds): one more MFMA per 32-key chunk multiplies the probabilities by a fragment of ones, which leaves every lane the sum of ITS query row -- of the bf16
		// probabilities, i.e. of exactly what the PV product used
		bf16x8 pf[NKT / 2];
#pragma unroll
		for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
			for (int r = 0; r < 4; ++r) pf[kt >> 1][(kt & 1) * 4 + r] = (bf16)__builtin_amdgcn_exp2f(s[kt][r] * c2 - mc);
		}
		// pass 2: o = P V
		f32x4 acc[DT], accl = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
		for (int dt = 0; dt < DT; ++dt) acc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
		for (int c = 0; c < NKT / 2; ++c) {
#pragma unroll
			for (int dt = 0; dt < DT; ++dt) {
				const int col = dt * 16 + 4 * pp;
				bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(sv + voff<D>(c * 32 + 4 * g + q, col)));
				bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(sv + voff<D>(c * 32 + 16 + 4 * g + q, col)));
				bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
				acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[c], acc[dt], 0, 0, 0);
			}
			accl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones8, pf[c], accl, 0, 0, 0);
		}
		const float sum = accl[0];
		if (qi < N) {
			const float inv = sum > 0.f ? 1.f / sum : 0.f;
#pragma unroll
			for (int dt = 0; dt < DT; ++dt) {
				bf16x4 ov = {(bf16)(acc[dt][0] * inv), (bf16)(acc[dt][1] * inv), (bf16)(acc[dt][2] * inv), (bf16)(acc[dt][3] * inv)};
				*reinterpret_cast<bf16x4*>(o + ((size_t)b * N + qi) * W + h * D + dt * 16 + 4 * g) = ov;
			}
		}
#pragma unroll
		for (int ks = 0; ks < KS; ++ks) qf[ks] = qn[ks];
	}
}

// More keys than the K/V-resident kernel holds (the DFN5B ViT-H/14 at 378 pixels: 730 tokens; the 384-pixel SigLIP towers: 729): the same per-tile arithmetic -- all score
// tiles of a key block back to back, one maximum / sum per row, bf16 probabilities in the score registers, the PV MFMAs back to back -- over BLOCKS of 128 keys with the
// online soft-max ONCE PER BLOCK (a running maximum, the accumulators rescaled by exp2(old - new)), instead of vit_attn_kernel's dependent chain per 32 keys.  A workgroup
// owns 128 queries (NW waves x QT 16-query tiles) of one (image, head) and streams the head's K and V through two LDS buffers by LDS-DMA: block kb + 1 is in flight
// while block kb is multiplied, one barrier per block; 80 KiB of LDS at head_dim 80, so two workgroups share a CU.  Eight waves of one 16-query tile each (four waves per
// SIMD) beat four waves of two tiles: ViT-H/14 at 378 pixels, 128 images: 1 469 us per layer on the streaming kernel -> 858 (4 x 2) -> 790-800 (8 x 1; 437-442 TFLOP/s;
// 1 024 tokens at head_dim 64: 604; the K/V-resident kernel reaches 390-400 at ViT-L/14: tools/attn_bench.py).  Rounding: probabilities relative to the running maximum of the blocks so far (as the streaming kernel, per 32 keys there).
template <int D, bool CAUSAL, int QT, int NW>
__global__ __launch_bounds__(NW * 64, 2) void vit_attn_blocked_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o, int B, int N, int H, float scale) {
	constexpr int KS = (D + 31) / 32, DT = D / 16, CPR = D / 8, KB = 128, NKT = KB / 16;
	constexpr bool POW2 = (CPR & (CPR - 1)) == 0;
	constexpr int OPB = KB * D * 2;   // one operand block
	constexpr int NI = OPB / 1024;    // LDS-DMA instructions per operand block
	static_assert(OPB % 1024 == 0 && (2 * NI) % NW == 0, "a block is whole LDS-DMA instructions, dealt out evenly to the waves");
	extern __shared__ __attribute__((aligned(16))) char smem_attn[];  // [2 buffers][K block [KB][D] | V block [KB][D]]
	const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4;
	const int bh = xcd_major_id(blockIdx.x, gridDim.x), b = bh / H, h = bh - b * H;
	const int W = H * D;
	const bf16* base = qkv + (size_t)b * N * 3 * W + h * D;
	const int q_wg0 = blockIdx.y * (NW * QT * 16);
	typedef __attribute__((address_space(3))) void* lds_ptr_t;
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(base), 0, (int)(((size_t)N * 3 * W - h * D) * 2), 0x00020000);
	// lane L of instruction i fills 16-byte slot c = 64 i + L of the operand block -- row c / CPR -- with the chunk the read-side XOR swizzle (voff) expects there;
	// key rows >= N: an out-of-range offset, which the buffer load returns as zeros.  The lane's row and byte offset inside a block do not change from block to block:
	// worked out once (the division by CPR = 10 at head_dim 80 costs a multiply-high chain per instruction otherwise, inside the loop).
	constexpr int NSTG = 2 * NI / NW;
	int st_row[NSTG];
	unsigned st_off[NSTG];
#pragma unroll
	for (int i0 = 0; i0 < NSTG; ++i0) {
		const int i = i0 * NW + w;  // wave-uniform
		const int isv = i >= NI ? 1 : 0, ii = i - isv * NI;
		const int c = ii * 64 + lane, row = c / CPR, slot = c - row * CPR;
		const int ch = POW2 ? (slot ^ (row & (CPR - 1) & 7)) : slot;
		st_row[i0] = row;
		st_off[i0] = (unsigned)(((size_t)row * 3 * W + (1 + isv) * W + ch * 8) * 2);
	}
	const unsigned blk_bytes = (unsigned)((size_t)KB * 3 * W * 2);
	auto stage = [&](int buf, int kb) {
		char* dst = smem_attn + buf * 2 * OPB;
#pragma unroll
		for (int i0 = 0; i0 < NSTG; ++i0) {
			const int i = i0 * NW + w;
			const unsigned off = kb * KB + st_row[i0] < N ? st_off[i0] + (unsigned)kb * blk_bytes : 0xFFFFFFF0u;
			__builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + (size_t)i * 1024), 16, off, 0, 0, 0);
		}
	};
	const int nkb_all = (N + KB - 1) / KB;
	const int nkb = CAUSAL ? min(nkb_all, (min(N, q_wg0 + NW * QT * 16) + KB - 1) / KB) : nkb_all;  // causal: keys behind the workgroup's last query never matter
	stage(0, 0);
	bf16x8 qf[QT][KS];
	int qt[QT];
#pragma unroll
	for (int t = 0; t < QT; ++t) {
		qt[t] = q_wg0 / 16 + w * QT + t;
		const int qrow = min(qt[t] * 16 + (lane & 15), N - 1);
#pragma unroll
		for (int ks = 0; ks < KS; ++ks) {
			const int col = ks * 32 + 8 * g;
			bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
			qf[t][ks] = z;
			if (col < D) qf[t][ks] = *reinterpret_cast<const bf16x8*>(base + (size_t)qrow * 3 * W + col);
		}
	}
	f32x4 acc[QT][DT], accl[QT];
	float m_run[QT];
	const bf16x8 ones8 = {(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f};
#pragma unroll
	for (int t = 0; t < QT; ++t) {
		m_run[t] = -1e30f;
		accl[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
		for (int dt = 0; dt < DT; ++dt) acc[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
	}
	const float c2 = scale * 1.4426950408889634f;  // exp(x * scale) = exp2(x * c2)
	const int q = (lane >> 2) & 3, pp = lane & 3;
	typedef bf16x4 __attribute__((address_space(3))) * lds4_t;
	for (int kb = 0; kb < nkb; ++kb) {
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of block kb (and, the first time, its queries) have landed ...
		__syncthreads();                                   // ... everybody's have, and everybody has finished reading the other buffer (block kb - 1)
		if (kb + 1 < nkb) stage((kb + 1) & 1, kb + 1);
		const char* sk = smem_attn + (kb & 1) * 2 * OPB;
		const char* sv = sk + OPB;
		const int k0 = kb * KB;
		// (one tile after the other: sharing every K / V fragment between the wave's two tiles -- half the LDS reads, 214 instead of 161 registers -- measured the same,
		// 887 against 858 us per layer at ViT-H/14-378: the MFMA -> soft-max -> MFMA chain of a tile, not the LDS pipe, is what two waves per SIMD do not hide)
#pragma unroll
		for (int t = 0; t < QT; ++t) {
			if (qt[t] * 16 >= N) continue;  // (wave-uniform: a tile behind the last query)
			const int qi = qt[t] * 16 + (lane & 15);
			f32x4 s[NKT];
#pragma unroll
			for (int kt = 0; kt < NKT; ++kt) {
				s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
				for (int ks = 0; ks < KS; ++ks) {
					// columns >= D (head_dim 80: the last k-step covers 64 .. 95) are read all the same -- the bytes behind the row, finite values inside the buffer -- and
					// multiplied by the zeros the query fragment holds there: no predicated load, no zero fill in the loop
					const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sk + voff_lin<D>(kt * 16 + (lane & 15), ks * 32 + 8 * g));
					s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[t][ks], s[kt], 0, 0, 0);
				}
			}
			float mx = -1e30f;
#pragma unroll
			for (int kt = 0; kt < NKT; ++kt) {
				if (k0 + (kt + 1) * 16 > N || (CAUSAL && k0 + (kt + 1) * 16 > qt[t] * 16)) {
					asm volatile("" ::: "memory");  // a real wave-uniform branch (see vit_attn_full_kernel)
#pragma unroll
					for (int r = 0; r < 4; ++r) {
						const int j = k0 + kt * 16 + 4 * g + r;
						if (!(j < N && (!CAUSAL || j <= qi))) s[kt][r] = -1e30f;
					}
				}
#pragma unroll
				for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
			}
			mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
			mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
			const float m_new = fmaxf(m_run[t], mx);
			const float alpha = __builtin_amdgcn_exp2f((m_run[t] - m_new) * c2);
			const float mc = m_new * c2;
			bf16x8 pf[NKT / 2];
#pragma unroll
			for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
				for (int r = 0; r < 4; ++r) pf[kt >> 1][(kt & 1) * 4 + r] = (bf16)__builtin_amdgcn_exp2f(s[kt][r] * c2 - mc);
			}
			m_run[t] = m_new;
			// the running row sum is one more accumulator tile: probabilities x a fragment of ones (see vit_attn_full_kernel), rescaled with the others
			accl[t] = accl[t] * alpha;
#pragma unroll
			for (int dt = 0; dt < DT; ++dt) acc[t][dt] = acc[t][dt] * alpha;
#pragma unroll
			for (int c = 0; c < NKT / 2; ++c) {
#pragma unroll
				for (int dt = 0; dt < DT; ++dt) {
					const int col = dt * 16 + 4 * pp;
					bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(sv + voff<D>(c * 32 + 4 * g + q, col)));
					bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(sv + voff<D>(c * 32 + 16 + 4 * g + q, col)));
					bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
					acc[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[c], acc[t][dt], 0, 0, 0);
				}
				accl[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones8, pf[c], accl[t], 0, 0, 0);
			}
		}
	}
#pragma unroll
	for (int t = 0; t < QT; ++t) {
		const int qi = qt[t] * 16 + (lane & 15);
		if (qi < N) {
			const float inv = accl[t][0] > 0.f ? 1.f / accl[t][0] : 0.f;
#pragma unroll
			for (int dt = 0; dt < DT; ++dt) {
				bf16x4 ov = {(bf16)(acc[t][dt][0] * inv), (bf16)(acc[t][dt][1] * inv), (bf16)(acc[t][dt][2] * inv), (bf16)(acc[t][dt][3] * inv)};
				*reinterpret_cast<bf16x4*>(o + ((size_t)b * N + qi) * W + h * D + dt * 16 + 4 * g) = ov;
			}
		}
	}
}

std::atomic<int> g_attn_policy{1};  // 0: streaming kernel only, 1: K/V-resident two-pass kernel where it fits, the blocked kernel beyond 288 keys (novic_vit_attn_policy)

inline int rows_grid(int rows) {
	int b = (rows + 3) / 4;
	return b < 1 ? 1 : (b > 8192 ? 8192 : b);
}

}  // namespace

#define NOVIC_VIT_NC(E, CALL)                          \
	switch (((E) + 255) / 256) {                       \
		case 1: { constexpr int NC = 1; CALL; break; } \
		case 2: { constexpr int NC = 2; CALL; break; } \
		case 3: { constexpr int NC = 3; CALL; break; } \
		case 4: { constexpr int NC = 4; CALL; break; } \
		case 5: { constexpr int NC = 5; CALL; break; } \
		case 6: { constexpr int NC = 6; CALL; break; } \
		case 7: { constexpr int NC = 7; CALL; break; } \
		case 8: { constexpr int NC = 8; CALL; break; } \
		default: novic_set_error("row kernels support widths <= 2048"); return -22; \
	}

template <typename PIX>
static int launch_im2col(const PIX* images, void* patches_bf16, int B, int R, int patch, int k_padded, const PixNorm& nm, hipStream_t stream) {
	const int g = R / patch;
	size_t total = (size_t)B * g * g * (k_padded / 4);
	int grid = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
	const bool vec4 = patch % 4 == 0 && R % 4 == 0 && (3 * patch * patch) % 4 == 0 && (((uintptr_t)images & (4 * sizeof(PIX) - 1)) == 0) && (uint64_t)B * 3 * R * R < 0xFFFFFFFFull;
	if (vec4) {
		const size_t n4 = (size_t)B * 3 * R * (R / 4);
		const int grid4 = (int)((n4 + 255) / 256 > 32768 ? 32768 : (n4 + 255) / 256);
		hipLaunchKernelGGL(im2col_vec4_kernel<PIX>, dim3(grid4), dim3(256), 0, stream, images, (bf16*)patches_bf16, B, R, patch, g, k_padded, nm);
	} else {
		hipLaunchKernelGGL(im2col_kernel<PIX>, dim3(grid), dim3(256), 0, stream, images, (bf16*)patches_bf16, B, R, patch, g, k_padded, nm);
	}
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_vit_im2col(const float* images, void* patches_bf16, int B, int R, int patch, int k_padded, hipStream_t stream) {
	NOVIC_CHECK(images && patches_bf16, "novic_vit_im2col: null pointer");
	NOVIC_CHECK(patch >= 1 && R % patch == 0 && k_padded % 8 == 0 && k_padded >= 3 * patch * patch, "novic_vit_im2col: bad patch geometry");
	if (B <= 0) return 0;
	return launch_im2col<float>(images, patches_bf16, B, R, patch, k_padded, PixNorm{}, stream);
}

extern "C" int novic_vit_im2col_u8(const uint8_t* images, void* patches_bf16, int B, int R, int patch, int k_padded, novic_pixel_norm_t norm, hipStream_t stream) {
	NOVIC_CHECK(images && patches_bf16, "novic_vit_im2col_u8: null pointer");
	NOVIC_CHECK(patch >= 1 && R % patch == 0 && k_padded % 8 == 0 && k_padded >= 3 * patch * patch, "novic_vit_im2col_u8: bad patch geometry");
	NOVIC_CHECK(norm.std[0] != 0.f && norm.std[1] != 0.f && norm.std[2] != 0.f, "novic_vit_im2col_u8: zero std");
	if (B <= 0) return 0;
	PixNorm nm;
	for (int c = 0; c < 3; ++c) { nm.mean[c] = norm.mean[c]; nm.std[c] = norm.std[c]; }
	return launch_im2col<uint8_t>(images, patches_bf16, B, R, patch, k_padded, nm, stream);
}

extern "C" int novic_vit_embed(const void* patches_bf16, const float* cls, const float* pos, const float* ln_gamma, const float* ln_beta, float* x, int B, int N, int W,
                               float eps, hipStream_t stream) {
	NOVIC_CHECK(patches_bf16 && cls && pos && x, "novic_vit_embed: null pointer");
	NOVIC_CHECK(W % 4 == 0 && N >= 2, "novic_vit_embed: bad shape");
	NOVIC_CHECK((ln_gamma == nullptr) == (ln_beta == nullptr), "novic_vit_embed: ln_pre needs both weight and bias (or neither)");
	if (B <= 0) return 0;
	const int has_ln = ln_gamma != nullptr;
	NOVIC_VIT_NC(W, hipLaunchKernelGGL((vit_embed_kernel<NC>), dim3(rows_grid(B * N)), dim3(256), 0, stream, (const bf16*)patches_bf16, cls, pos, ln_gamma, ln_beta, x, B, N,
	                                   W, eps, has_ln));
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_vit_embed_f16(const void* patches_bf16, const float* cls, const float* pos, const float* ln_gamma, const float* ln_beta, void* x_f16, int B, int N, int W,
                                   float eps, hipStream_t stream) {
	NOVIC_CHECK(patches_bf16 && cls && pos && x_f16, "novic_vit_embed_f16: null pointer");
	NOVIC_CHECK(W % 4 == 0 && N >= 2, "novic_vit_embed_f16: bad shape");
	NOVIC_CHECK((ln_gamma == nullptr) == (ln_beta == nullptr), "novic_vit_embed_f16: ln_pre needs both weight and bias (or neither)");
	if (B <= 0) return 0;
	const int has_ln = ln_gamma != nullptr;
	NOVIC_VIT_NC(W, hipLaunchKernelGGL((vit_embed_kernel<NC, f16>), dim3(rows_grid(B * N)), dim3(256), 0, stream, (const bf16*)patches_bf16, cls, pos, ln_gamma, ln_beta,
	                                   (f16*)x_f16, B, N, W, eps, has_ln));
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_rownorm_f32(const float* x, float* y, int rows, int E, hipStream_t stream) {
	NOVIC_CHECK(x && y, "novic_rownorm_f32: null pointer");
	NOVIC_CHECK(E % 4 == 0 && E > 0, "novic_rownorm_f32: E must be a multiple of 4");
	if (rows <= 0) return 0;
	NOVIC_VIT_NC(E, hipLaunchKernelGGL((rownorm_f32_kernel<NC>), dim3(rows_grid(rows)), dim3(256), 0, stream, x, y, rows, E));
	NOVIC_LAUNCH_CHECK();
	return 0;
}

template <int D, int NKT, bool CAUSAL, int NW>
static void launch_full(const void* qkv_bf16, void* o_bf16, int B, int N, int H, float scale, hipStream_t stream) {
	constexpr int lds = 2 * NKT * 16 * D * 2;
	static std::atomic<bool> attr{false};
	if (!attr) {
		(void)hipFuncSetAttribute((const void*)vit_attn_full_kernel<D, NKT, CAUSAL, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
		attr = true;
	}
	hipLaunchKernelGGL((vit_attn_full_kernel<D, NKT, CAUSAL, NW>), dim3(B * H), dim3(NW * 64), lds, stream, (const bf16*)qkv_bf16, (bf16*)o_bf16, B, N, H, scale);
}
// waves per workgroup: the count among 4 / 6 / 7 that wastes the fewest tile slots (17 query tiles at N = 257: 6 waves x 3 rounds; 13 at N = 197:
// 7 x 2; 5 at N = 77: 6 x 1; 4 at N = 50: 4 x 1)
template <int D, int NKT, bool CAUSAL>
static void launch_full_w(const void* qkv_bf16, void* o_bf16, int B, int N, int H, float scale, hipStream_t stream) {
	// waves per workgroup: the fewest idle waves in the last round of 16-query tiles.  (Eight waves -- four per SIMD with two workgroups per CU -- measured the same or
	// 2-3 % slower here, ViT-L/14 173 -> 178 us, SO400M 307 -> 308: unlike the blocked kernel below, this one is not short of waves.)
	const int nt = (N + 15) / 16;
	int best = 4, waste = (nt + 3) / 4 * 4 - nt;
	for (int nw : {6, 7}) {
		const int ws = (nt + nw - 1) / nw * nw - nt;
		if (ws < waste) { waste = ws; best = nw; }
	}
	if (best == 4) launch_full<D, NKT, CAUSAL, 4>(qkv_bf16, o_bf16, B, N, H, scale, stream);
	else if (best == 6) launch_full<D, NKT, CAUSAL, 6>(qkv_bf16, o_bf16, B, N, H, scale, stream);
	else launch_full<D, NKT, CAUSAL, 7>(qkv_bf16, o_bf16, B, N, H, scale, stream);
}
template <int D, bool CAUSAL>
static void launch_full_d(const void* qkv_bf16, void* o_bf16, int B, int N, int H, float scale, int NP, hipStream_t stream) {
	if (NP <= 96) launch_full_w<D, 6, CAUSAL>(qkv_bf16, o_bf16, B, N, H, scale, stream);
	else if (NP <= 224) launch_full_w<D, 14, CAUSAL>(qkv_bf16, o_bf16, B, N, H, scale, stream);
	else if (NP <= 256) launch_full_w<D, 16, CAUSAL>(qkv_bf16, o_bf16, B, N, H, scale, stream);  // (SigLIP SO400M/14: exactly 256 tokens)
	else launch_full_w<D, 18, CAUSAL>(qkv_bf16, o_bf16, B, N, H, scale, stream);
}

template <int D, bool CAUSAL, int QT, int NW>
static void launch_blocked_f(const void* qkv_bf16, void* o_bf16, int B, int N, int H, float scale, hipStream_t stream) {
	constexpr int LDS = 4 * 128 * D * 2;
	static std::atomic<bool> attr_done{false};
	if (!attr_done) {
		(void)hipFuncSetAttribute((const void*)vit_attn_blocked_kernel<D, CAUSAL, QT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
		attr_done = true;
	}
	constexpr int QW = NW * QT * 16;
	hipLaunchKernelGGL((vit_attn_blocked_kernel<D, CAUSAL, QT, NW>), dim3(B * H, (N + QW - 1) / QW), dim3(NW * 64), LDS, stream, (const bf16*)qkv_bf16, (bf16*)o_bf16, B, N, H, scale);
}
template <int D, bool CAUSAL>
static void launch_blocked(const void* qkv_bf16, void* o_bf16, int B, int N, int H, float scale, hipStream_t stream) {
	launch_blocked_f<D, CAUSAL, 1, 8>(qkv_bf16, o_bf16, B, N, H, scale, stream);  // eight waves x one query tile (round 3's four waves x two tiles measured slower and went in round 5)
}

extern "C" int novic_vit_attn_policy(int policy) {
	const int prev = g_attn_policy;
	if (policy == 0 || policy == 1) g_attn_policy = policy;
	return prev;
}

static int clip_attn_launch(const void* qkv_bf16, void* o_bf16, int B, int N, int H, int D, int causal, hipStream_t stream, float scale_in = 0.f) {
	dim3 grid(B * H, (N + 63) / 64), block(256);
	const float scale = scale_in > 0.f ? scale_in : 1.f / sqrtf((float)D);
	// up to 288 keys (18 score tiles in registers), head_dim 64 / 80: K and V of a head resident in LDS, exact soft-max
	const int NP = (N + 31) / 32 * 32;
	if (g_attn_policy == 1 && N > 16 && NP <= 288 && (D == 64 || (D == 80 && !causal)) && (size_t)N * 3 * H * D * 2 < 0x7FFFFFF0ull) {
		if (D == 64 && causal) launch_full_d<64, true>(qkv_bf16, o_bf16, B, N, H, scale, NP, stream);
		else if (D == 64) launch_full_d<64, false>(qkv_bf16, o_bf16, B, N, H, scale, NP, stream);
		else launch_full_d<80, false>(qkv_bf16, o_bf16, B, N, H, scale, NP, stream);
		NOVIC_LAUNCH_CHECK();
		return 0;
	}
	// more keys than that: 128-key blocks through two LDS buffers, the online soft-max once per block
	if (g_attn_policy == 1 && NP > 288 && (D == 64 || D == 80) && (size_t)N * 3 * H * D * 2 < 0x7FFFFFF0ull) {
		if (D == 64 && causal) launch_blocked<64, true>(qkv_bf16, o_bf16, B, N, H, scale, stream);
		else if (D == 64) launch_blocked<64, false>(qkv_bf16, o_bf16, B, N, H, scale, stream);
		else if (causal) launch_blocked<80, true>(qkv_bf16, o_bf16, B, N, H, scale, stream);
		else launch_blocked<80, false>(qkv_bf16, o_bf16, B, N, H, scale, stream);
		NOVIC_LAUNCH_CHECK();
		return 0;
	}
	switch (D) {
		case 32: hipLaunchKernelGGL((vit_attn_kernel<32>), grid, block, 0, stream, (const bf16*)qkv_bf16, (bf16*)o_bf16, B, N, H, scale, causal); break;
		case 64: hipLaunchKernelGGL((vit_attn_kernel<64>), grid, block, 0, stream, (const bf16*)qkv_bf16, (bf16*)o_bf16, B, N, H, scale, causal); break;
		case 80: hipLaunchKernelGGL((vit_attn_kernel<80>), grid, block, 0, stream, (const bf16*)qkv_bf16, (bf16*)o_bf16, B, N, H, scale, causal); break;
		default: novic_set_error("CLIP attention: head_dim must be 32, 64 or 80"); return -22;
	}
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_vit_attn_fwd(const void* qkv_bf16, void* o_bf16, int B, int N, int H, int D, hipStream_t stream) {
	NOVIC_CHECK(qkv_bf16 && o_bf16, "novic_vit_attn_fwd: null pointer");
	NOVIC_CHECK(B >= 0 && N >= 1 && H >= 1, "novic_vit_attn_fwd: bad shape");
	if (B == 0) return 0;
	return clip_attn_launch(qkv_bf16, o_bf16, B, N, H, D, 0, stream);
}

// ---------------------------------------------------------------------------------------------------------
// CLIP text tower helpers (embedders.py:423-426, :557-583, :728-753 -> third-party encode_text): token + positional embedding, causal attention
// (the kernel above with its causal bound), END-OF-TEXT pooling.  Linear layers / LayerNorms are novic_gemm_bf16 / novic_layernorm_fwd launches.
// ---------------------------------------------------------------------------------------------------------
namespace {

template <typename XT = float>  // (XT = f16: the half-precision residual stream of clip's fp16 text tower, novic_text_embed_f16 / novic_text_pool_f16)
__global__ __launch_bounds__(256) void text_embed_kernel(const void* __restrict__ ids, int tok_bytes, const float* __restrict__ tok_emb, const float* __restrict__ pos,
                                                         XT* __restrict__ x, int rows, int S, int W, int V) {
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (int r = blockIdx.x * 4 + w; r < rows; r += gridDim.x * 4) {
		long long t = tok_bytes == 8 ? ((const long long*)ids)[r] : (long long)((const int*)ids)[r];
		t = t < 0 ? 0 : (t >= V ? V - 1 : t);
		const int s = r % S;
		for (int e = lane * 4; e < W; e += 256) {
			const f32x4 a = *reinterpret_cast<const f32x4*>(tok_emb + (size_t)t * W + e), p = *reinterpret_cast<const f32x4*>(pos + (size_t)s * W + e);
			if constexpr (sizeof(XT) == 2) *reinterpret_cast<f16x4*>(x + (size_t)r * W + e) = (f16x4){(f16)(a[0] + p[0]), (f16)(a[1] + p[1]), (f16)(a[2] + p[2]), (f16)(a[3] + p[3])};
			else *reinterpret_cast<f32x4*>(x + (size_t)r * W + e) = (f32x4){a[0] + p[0], a[1] + p[1], a[2] + p[2], a[3] + p[3]};
		}
	}
}

// out[b] = x[b][s*], s* = argmax_s ids[b][s] (first maximum; eot < 0: CLIP's own vocabulary, where END-OF-TEXT is the largest id) or the first s with
// ids[b][s] == eot (any other vocabulary; position 0 if absent): one wave per sample
template <typename XT = float>
__global__ __launch_bounds__(256) void text_pool_kernel(const void* __restrict__ ids, int tok_bytes, const XT* __restrict__ x, float* __restrict__ out, int B, int S, int W,
                                                        long long eot) {
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (int b = blockIdx.x * 4 + w; b < B; b += gridDim.x * 4) {
		long long best = -1;
		int bi = 0x7fffffff;
		for (int s = lane; s < S; s += 64) {
			long long t = tok_bytes == 8 ? ((const long long*)ids)[(size_t)b * S + s] : (long long)((const int*)ids)[(size_t)b * S + s];
			if (eot >= 0) t = (t == eot) ? 1 : 0;
			if (t > best) { best = t; bi = s; }
		}
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			const long long ob = __shfl_xor(best, o, 64);
			const int oi = __shfl_xor(bi, o, 64);
			if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
		}
		for (int e = lane * 4; e < W; e += 256) {
			if constexpr (sizeof(XT) == 2) {
				const f16x4 h = *reinterpret_cast<const f16x4*>(x + ((size_t)b * S + bi) * W + e);
				*reinterpret_cast<f32x4*>(out + (size_t)b * W + e) = (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
			} else {
				*reinterpret_cast<f32x4*>(out + (size_t)b * W + e) = *reinterpret_cast<const f32x4*>(x + ((size_t)b * S + bi) * W + e);
			}
		}
	}
}

}  // namespace

extern "C" int novic_clip_attn_fwd(const void* qkv_bf16, void* o_bf16, int B, int N, int H, int D, int causal, hipStream_t stream) {
	NOVIC_CHECK(qkv_bf16 && o_bf16, "novic_clip_attn_fwd: null pointer");
	NOVIC_CHECK(B >= 0 && N >= 1 && H >= 1, "novic_clip_attn_fwd: bad shape");
	if (B == 0) return 0;
	return clip_attn_launch(qkv_bf16, o_bf16, B, N, H, D, causal ? 1 : 0, stream);
}

extern "C" int novic_clip_attn_fwd_scaled(const void* qkv_bf16, void* o_bf16, int B, int N, int H, int D, int causal, float scale, hipStream_t stream) {
	NOVIC_CHECK(qkv_bf16 && o_bf16, "novic_clip_attn_fwd_scaled: null pointer");
	NOVIC_CHECK(B >= 0 && N >= 1 && H >= 1 && scale == scale, "novic_clip_attn_fwd_scaled: bad shape / scale");
	if (B == 0) return 0;
	return clip_attn_launch(qkv_bf16, o_bf16, B, N, H, D, causal ? 1 : 0, stream, scale);
}

extern "C" int novic_text_embed(const void* ids, int tok_bytes, const float* tok_emb, const float* pos, float* x, int B, int S, int W, int V, hipStream_t stream) {
	NOVIC_CHECK(ids && tok_emb && pos && x, "novic_text_embed: null pointer");
	NOVIC_CHECK((tok_bytes == 4 || tok_bytes == 8) && W % 4 == 0 && S >= 1 && V >= 1, "novic_text_embed: bad arguments");
	if (B <= 0) return 0;
	hipLaunchKernelGGL(text_embed_kernel<float>, dim3(rows_grid(B * S)), dim3(256), 0, stream, ids, tok_bytes, tok_emb, pos, x, B * S, S, W, V);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_text_pool(const void* ids, int tok_bytes, const float* x, float* out, int B, int S, int W, long long eot_id, hipStream_t stream) {
	NOVIC_CHECK(ids && x && out, "novic_text_pool: null pointer");
	NOVIC_CHECK((tok_bytes == 4 || tok_bytes == 8) && W % 4 == 0 && S >= 1, "novic_text_pool: bad arguments");
	if (B <= 0) return 0;
	hipLaunchKernelGGL(text_pool_kernel<float>, dim3(rows_grid(B)), dim3(256), 0, stream, ids, tok_bytes, x, out, B, S, W, eot_id);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_text_embed_f16(const void* ids, int tok_bytes, const float* tok_emb, const float* pos, void* x_f16, int B, int S, int W, int V, hipStream_t stream) {
	NOVIC_CHECK(ids && tok_emb && pos && x_f16, "novic_text_embed_f16: null pointer");
	NOVIC_CHECK((tok_bytes == 4 || tok_bytes == 8) && W % 4 == 0 && S >= 1 && V >= 1, "novic_text_embed_f16: bad arguments");
	if (B <= 0) return 0;
	hipLaunchKernelGGL(text_embed_kernel<f16>, dim3(rows_grid(B * S)), dim3(256), 0, stream, ids, tok_bytes, tok_emb, pos, (f16*)x_f16, B * S, S, W, V);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_text_pool_f16(const void* ids, int tok_bytes, const void* x_f16, float* out, int B, int S, int W, long long eot_id, hipStream_t stream) {
	NOVIC_CHECK(ids && x_f16 && out, "novic_text_pool_f16: null pointer");
	NOVIC_CHECK((tok_bytes == 4 || tok_bytes == 8) && W % 4 == 0 && S >= 1, "novic_text_pool_f16: bad arguments");
	if (B <= 0) return 0;
	hipLaunchKernelGGL(text_pool_kernel<f16>, dim3(rows_grid(B)), dim3(256), 0, stream, ids, tok_bytes, (const f16*)x_f16, out, B, S, W, eot_id);
	NOVIC_LAUNCH_CHECK();
	return 0;
}
