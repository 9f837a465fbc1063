// The feed-forward half of a decoder layer as ONE launch over row tiles (reference: nn.TransformerEncoderLayer built in embedding_decoder.py:309-327,
// norm_first: x + dropout(linear2(dropout(gelu(linear1(norm2(x)))))), followed by the NEXT layer's norm1):
//
//     ln2 = LayerNorm(xmid) -> hpre = bf16(ln2 W1^T) -> hact = dropout(gelu(hpre)) -> x = xmid + dropout(bf16(hact W2^T)) -> ln1' = LayerNorm'(x)
//
// Unfused this is four launches (LayerNorm, the two narrow GEMMs of skinny.hip, the next LayerNorm) that move 23.5 E bytes per row (E = 512) through
// HBM -- the normalised rows written and read back twice, the residual stream read three times; here a row tile's xmid is read ONCE (4 E), the fp32
// residual stream written once (4 E), and only what the backward pass needs is stored beside it (ln2 2 E, hpre + hact 1 E, the next layer's ln1 2 E):
// 13 E.  Both weight matrices live in REGISTERS as MFMA fragments for the whole launch (linear1: a wave's 16 output columns x K = 512, linear2: its 64
// output columns x K = 128: 64 VGPRs each), the tile's rows in LDS (bf16 images for the two GEMMs, an fp32 image of the residual rows).
//
// Arithmetic, operation for operation, is that of the kernels it replaces -- layernorm_fwd_kernel (norm.hip: one wave per row, ln_row_stats / ln_apply),
// skinny_n128_kernel<GELU_BF16> and skinny_k128_resid_kernel (skinny.hip: v_mfma_f32_16x16x32_bf16, K accumulated in the same order, the shared
// epilogue helpers, dropout masks keyed by (site, row * N + column)).  Fed the same inputs, the two GEMM stages are BIT-IDENTICAL to the kernels they replace;
// so are the LayerNorm stages since round 5 (common.hpp `unfused`: until then two compilations of ln_row_stats could differ in whether the variance was accumulated
// with fused multiply-adds -- a bf16 output off by one ulp where its fp32 value sat on a rounding tie, 1-2 elements per million; tests/test_gpu_ffn.py checks each
// stage on the fused kernel's own input to it).
#include "gemm_epilogue.hpp"

namespace {

constexpr int FF_E = 512, FF_K = 128, FF_ROWS = 16, FF_NT = 512;  // 16-row tiles: with 32 the prefetched rows + both weight matrices spill
constexpr int FF_RPW = FF_ROWS / 8, FF_MT = FF_ROWS / 16;  // rows per wave in the LayerNorm phases, 16-row MFMA tiles per tile
constexpr int FF_A1 = FF_ROWS * FF_E * 2;    // ln2 image, bf16: 16 KiB
constexpr int FF_A2 = FF_ROWS * FF_K * 2;    // hact image, bf16: 4 KiB
constexpr int FF_X = FF_ROWS * FF_E * 4;     // residual rows, fp32: 32 KiB
constexpr int FF_G = 2 * FF_E * 4;          // the two LayerNorm gain vectors, fp32: 4 KiB
constexpr int FF_LDS = FF_A1 + FF_A2 + FF_X + FF_G;
constexpr int FF_BWD_LDS = FF_A1 + FF_A2 + FF_G + FF_E * FF_K * 2;  // backward: gb/dln image, dh image, gamma2 | the prologue's gamma, W1^T resident = 152 KiB of the CU's 160

struct FfnArgs {
	const float* xmid;      // [M][512] fp32
	const float* gamma2;    // norm2.weight
	const bf16* w1;         // linear1.weight [128][512]
	const bf16* w2;         // linear2.weight [512][128]
	const float* gamma_next;  // the next layer's norm1.weight, or null
	float* x_out;           // [M][512] fp32
	bf16* ln2;              // [M][512] or null
	bf16* hpre;             // [M][128] or null
	bf16* hact;             // [M][128] or null
	bf16* ln_next;          // [M][512] (with gamma_next)
	int M;
	float eps;
	DropoutDesc drop_gelu, drop_out;
	const int* row_limit;
};

// Barrier between the phases of a tile: only LDS traffic crosses it.  __syncthreads() would also drain the vector-memory counter (its workgroup-scope fence
// waits for every outstanding global access): the next tile's prefetched rows and this tile's stores -- a full HBM round trip three times per tile
// (measured: 7.4 us per 16-row tile, 111 us per launch, against ~3 us of dependent work).
__device__ __forceinline__ void lds_barrier() {
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	__builtin_amdgcn_s_barrier();
	asm volatile("" ::: "memory");
}

template <bool HAS_NEXT>
__global__ __launch_bounds__(FF_NT) void ffn_fwd_kernel(const FfnArgs gin) {
	FfnArgs g = gin;
	if (g.row_limit) g.M = min(g.M, max(*g.row_limit, 0));
	extern __shared__ __attribute__((aligned(16))) char smem[];
	char* a1 = smem;                  // [rows][1 KiB], 16-byte chunks XOR-swizzled by row & 15
	char* a2 = smem + FF_A1;          // [rows][256 B], the same swizzle
	char* xs = smem + FF_A1 + FF_A2;  // [rows][2 KiB] fp32, 16-byte chunks XOR-swizzled by row & 15
	float* gs = reinterpret_cast<float*>(smem + FF_A1 + FF_A2 + FF_X);  // gamma2 | gamma_next: read from LDS per tile (held in registers they spill; re-read
	                                                                     // from global they sit in the vector-memory queue behind the prefetch and stall on it)
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const int fr = lane & 15, fq = lane >> 4;
	const int ntiles = (g.M + FF_ROWS - 1) / FF_ROWS;
	int t = blockIdx.x;
	if (t >= ntiles) return;

	// weights as "first operand" fragments of the swapped MFMA (a lane then owns 4 consecutive output columns of its row):
	// linear1: wave w computes hidden columns 16 w .. 16 w + 15; fragment ks holds W1[16 w + fr][32 ks + 8 fq .. + 7]
	// linear2: wave w computes output columns 64 w .. 64 w + 63; fragment (nt, ks) holds W2[64 w + 16 nt + fr][32 ks + 8 fq .. + 7]
	bf16x8 w1f[16], w2f[4][4];
#pragma unroll
	for (int ks = 0; ks < 16; ++ks) w1f[ks] = *reinterpret_cast<const bf16x8*>(g.w1 + (size_t)(16 * w + fr) * FF_E + ks * 32 + fq * 8);
#pragma unroll
	for (int nt = 0; nt < 4; ++nt)
#pragma unroll
		for (int ks = 0; ks < 4; ++ks) w2f[nt][ks] = *reinterpret_cast<const bf16x8*>(g.w2 + (size_t)(64 * w + 16 * nt + fr) * FF_K + ks * 32 + fq * 8);

	for (int i = tid; i < FF_E; i += FF_NT) {
		gs[i] = g.gamma2[i];
		gs[FF_E + i] = HAS_NEXT ? g.gamma_next[i] : 0.f;
	}
	__syncthreads();

	// LayerNorm phases: wave w owns rows RPW w .. RPW w + RPW - 1 of the tile, a row as v[c][i] = element 256 c + 4 lane + i (the layout of layernorm_fwd_kernel)
	// (the two gain vectors are re-read per tile, 2 KiB from L1: held in registers beside both weight matrices they spill)
	auto load_rows = [&](f32x4 (&p)[FF_RPW][2], int tile) {  // clamped row index: no branch around the loads
#pragma unroll
		for (int i = 0; i < FF_RPW; ++i) {
			int m = tile * FF_ROWS + FF_RPW * w + i;
			m = m < g.M ? m : g.M - 1;
#pragma unroll
			for (int c = 0; c < 2; ++c) p[i][c] = *reinterpret_cast<const f32x4*>(g.xmid + (size_t)m * FF_E + c * 256 + lane * 4);
		}
	};
	f32x4 cur[FF_RPW][2], nxt[FF_RPW][2];
	load_rows(cur, t);

	// Every global store is a buffer store through a descriptor sized to the rows that exist (0 for an output that is not wanted): a row at or beyond M is
	// dropped by the range check, so NO store sits behind a branch.  With `if (m < M)` around them hipcc can no longer count the vector-memory operations
	// between the prefetch loads and their use and falls back to `s_waitcnt vmcnt(0)` right behind the prefetch: a full HBM round trip per tile (7.4 us per
	// 16-row tile where the dependent work takes ~3).
	typedef unsigned ff_u32x2 __attribute__((ext_vector_type(2)));
	typedef unsigned ff_u32x4 __attribute__((ext_vector_type(4)));
	auto srd = [&](void* p, unsigned row_bytes) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, p ? (unsigned)g.M * row_bytes : 0u, 0x00020000); };
	const __amdgpu_buffer_rsrc_t s_ln2 = srd(g.ln2, FF_E * 2), s_hpre = srd(g.hpre, FF_K * 2), s_hact = srd(g.hact, FF_K * 2), s_x = srd(g.x_out, FF_E * 4),
	                             s_lnn = srd(g.ln_next, FF_E * 2);
	auto st8 = [](__amdgpu_buffer_rsrc_t r, bf16x4 v, unsigned off) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ff_u32x2, v), r, off, 0, 0); };

	for (; t < ntiles; t += gridDim.x) {
		const int tn = t + (int)gridDim.x;
		load_rows(nxt, tn < ntiles ? tn : ntiles - 1);  // the next tile's rows fly under this tile's phases (unconditional: a load behind a branch is uncountable too)
		const int m0 = t * FF_ROWS;

		// ---- norm2: statistics and normalised row as layernorm_fwd_kernel computes them; bf16 row -> A1 image (+ global ln2), fp32 row -> X image ----
		f32x4 gm2[2];
#pragma unroll
		for (int c = 0; c < 2; ++c) gm2[c] = *reinterpret_cast<const f32x4*>(gs + c * 256 + lane * 4);
#pragma unroll
		for (int i = 0; i < FF_RPW; ++i) {
			const int row = FF_RPW * w + i, m = m0 + row;
			float v[2][4];
#pragma unroll
			for (int c = 0; c < 2; ++c)
#pragma unroll
				for (int e = 0; e < 4; ++e) v[c][e] = cur[i][c][e];
			float mean, rstd;
			ln_row_stats<2>(v, FF_E, lane, g.eps, mean, rstd);
#pragma unroll
			for (int c = 0; c < 2; ++c) {
				const bf16x4 ob = {(bf16)ln_apply(v[c][0], mean, rstd, gm2[c][0]), (bf16)ln_apply(v[c][1], mean, rstd, gm2[c][1]), (bf16)ln_apply(v[c][2], mean, rstd, gm2[c][2]),
				                   (bf16)ln_apply(v[c][3], mean, rstd, gm2[c][3])};
				const int chunk = 32 * c + (lane >> 1);  // 8 bf16 per 16-byte chunk; this lane's four are one half of it
				*reinterpret_cast<bf16x4*>(a1 + row * 1024 + ((chunk ^ (row & 15)) << 4) + (lane & 1) * 8) = ob;
				st8(s_ln2, ob, ((unsigned)m * FF_E + c * 256 + lane * 4) * 2u);
				*reinterpret_cast<f32x4*>(xs + row * 2048 + (((64 * c + lane) ^ (row & 15)) << 4)) = cur[i][c];
			}
		}
		lds_barrier();

		// ---- linear1 + GELU (+ dropout): the tile's rows x this wave's 16 hidden columns, K = 512 in the order of skinny_n128_kernel ----
#pragma unroll
		for (int mt = 0; mt < FF_MT; ++mt) {
			const int row = mt * 16 + fr, m = m0 + row;
			const char* rowp = a1 + row * 1024;
			f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
			for (int kh = 0; kh < 2; ++kh) {  // the row's fragments eight at a time (all sixteen at once, with both weight matrices resident, spills)
				bf16x8 af[8];
#pragma unroll
				for (int ks = 0; ks < 8; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(rowp + ((((kh * 8 + ks) * 4 + fq) ^ fr) << 4));
#pragma unroll
				for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[kh * 8 + ks], af[ks], acc, 0, 0, 0);
			}
			// lane (fr, fq): row `row`, hidden columns 16 w + 4 fq .. + 3
			const int n = 16 * w + 4 * fq;
			float sc[4];
			dropout_scale4_branchless(g.drop_gelu, (uint64_t)m * FF_K + n, sc);
			bf16x4 pre, act;
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const float pr = bf16_round(acc[r]);
				pre[r] = (bf16)pr;
				act[r] = (bf16)gelu_fwd_elem(pr, sc[r]);
			}
			const int chunk = 2 * w + (fq >> 1);
			*reinterpret_cast<bf16x4*>(a2 + row * 256 + ((chunk ^ fr) << 4) + (fq & 1) * 8) = act;
			st8(s_hpre, pre, ((unsigned)m * FF_K + n) * 2u);
			st8(s_hact, act, ((unsigned)m * FF_K + n) * 2u);
		}
		lds_barrier();

		// ---- linear2 + dropout + residual: the tile's rows x this wave's 64 output columns, K = 128 in the order of skinny_k128_resid_kernel ----
#pragma unroll
		for (int mt = 0; mt < FF_MT; ++mt) {
			const int row = mt * 16 + fr, m = m0 + row;
			const char* rowp = a2 + row * 256;
			bf16x8 af[4];
#pragma unroll
			for (int ks = 0; ks < 4; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(rowp + (((ks * 4 + fq) ^ fr) << 4));
#pragma unroll
			for (int nt = 0; nt < 4; ++nt) {
				f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
				for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[nt][ks], af[ks], acc, 0, 0, 0);
				const int n = 64 * w + 16 * nt + 4 * fq;
				float sc[4];
				dropout_scale4_branchless(g.drop_out, (uint64_t)m * FF_E + n, sc);
				char* xp = xs + row * 2048 + ((((n >> 2)) ^ fr) << 4);
				const f32x4 res = *reinterpret_cast<const f32x4*>(xp);
				float v[4];
#pragma unroll
				for (int r = 0; r < 4; ++r) v[r] = res[r] + bf16_round(acc[r]) * sc[r];  // as epilogue4<RESID_F32> (no bias)
				*reinterpret_cast<f32x4*>(xp) = (f32x4){v[0], v[1], v[2], v[3]};
				__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ff_u32x4, (f32x4){v[0], v[1], v[2], v[3]}), s_x, ((unsigned)m * FF_E + n) * 4u, 0, 2);  // aux 2 = nt, as st_f32x4
			}
		}
		lds_barrier();

		// ---- the next layer's norm1 on the finished rows (wave w: its RPW rows, read back in the LayerNorm layout) ----
		if constexpr (HAS_NEXT) {
			f32x4 gmn[2];
#pragma unroll
			for (int c = 0; c < 2; ++c) gmn[c] = *reinterpret_cast<const f32x4*>(gs + FF_E + c * 256 + lane * 4);
#pragma unroll
			for (int i = 0; i < FF_RPW; ++i) {
				const int row = FF_RPW * w + i, m = m0 + row;
				float v[2][4];
#pragma unroll
				for (int c = 0; c < 2; ++c) {
					const f32x4 x4 = *reinterpret_cast<const f32x4*>(xs + row * 2048 + (((64 * c + lane) ^ (row & 15)) << 4));
#pragma unroll
					for (int e = 0; e < 4; ++e) v[c][e] = x4[e];
				}
				float mean, rstd;
				ln_row_stats<2>(v, FF_E, lane, g.eps, mean, rstd);
#pragma unroll
				for (int c = 0; c < 2; ++c) {
					const bf16x4 ob = {(bf16)ln_apply(v[c][0], mean, rstd, gmn[c][0]), (bf16)ln_apply(v[c][1], mean, rstd, gmn[c][1]),
					                   (bf16)ln_apply(v[c][2], mean, rstd, gmn[c][2]), (bf16)ln_apply(v[c][3], mean, rstd, gmn[c][3])};
					st8(s_lnn, ob, ((unsigned)m * FF_E + c * 256 + lane * 4) * 2u);
				}
			}
		}
		// the rows this wave rewrites next (its own X image / A1 rows) were last read by other waves before the barrier above; A2 is
		// rewritten only behind the next tile's first barrier
#pragma unroll
		for (int i = 0; i < FF_RPW; ++i)
#pragma unroll
			for (int c = 0; c < 2; ++c) cur[i][c] = nxt[i][c];
	}
}


// ---------------------------------------------------------------------------------------------------------------------------------------------
// Backward of the same block up to (not including) the weight gradients, one launch over row tiles:
//
//     dh = bf16(bf16(gb W2) * dropmask_gelu * gelu'(hpre))          (input gradient of linear2, skinny_n128_kernel<GELU_BWD> against W2^T [128][512])
//     dln = bf16(dh W1)                                              (input gradient of linear1, against W1^T [512][128])
//     dx  = dx_in + LayerNorm'(dln; xmid, gamma2) ;  g = bf16(dx * dropmask_attn_out) ;  dgamma2 += sum_rows dln * xhat      (layernorm_bwd_kernel)
//
// gb = the masked upstream gradient of the block's output (bf16 [M][512]), dx_in = the fp32 residual-stream gradient (updated in place is fine: a tile's
// rows are read before they are written).  dh is stored for the linear1 weight gradient; dln never leaves the CU (unfused: written and read back, 2 x 2 E
// bytes per row), the two narrow GEMMs and the LayerNorm backward share one read of the tile: 17 E bytes per row instead of 21.5 E and one launch
// instead of three.  Same MFMA / epilogue helpers / reduction helpers as the kernels it replaces; the LayerNorm statistics are recomputed per row in
// the layout of layernorm_bwd_kernel (one wave per row).
struct FfnBwdArgs {
	const bf16* gb;       // [M][512]  (PRE: written -- the masked gradient the prologue forms, for the linear2 weight gradient)
	const bf16* hpre;     // [M][128]
	const float* xmid;    // [M][512]
	const float* dx_in;   // [M][512]
	const float* gamma2;
	const bf16* w2t;      // linear2.weight^T [128][512]
	const bf16* w1t;      // linear1.weight^T [512][128]
	bf16* dh;             // [M][128]
	float* dx_out;        // [M][512]
	bf16* g_out;          // [M][512]
	float* dgamma2;       // [512], accumulated into
	// PRE: the LayerNorm backward of the layer ABOVE (its norm1, whose input is this block's output) as a prologue
	const bf16* pre_dln;  // [M][512] gradient w.r.t. that norm's output (the in-projection input gradient)
	const float* pre_x;   // [M][512] that norm's input = this block's output
	const float* pre_gamma;
	float* pre_dgamma;    // [512], accumulated into
	const int* pre_row_map;  // MODE 2: row m's upstream gradient is pre_dln row pre_row_map[m] (< 0: none -- the final norm over compacted output rows)
	bf16* gb_out;         // [M][512]
	int M;
	float eps;
	DropoutDesc drop_gelu, drop_g, drop_pre;
	const int* row_limit;
};

// PRE = false: gb and dx_in are inputs (the block below the final norm).  PRE = true: the kernel first runs, per row, the norm1 backward of the layer above
//     dx = dx_in + LayerNorm'(pre_dln; pre_x, pre_gamma) ;  gb = bf16(dx * dropmask_pre) ;  pre_dgamma += sum_rows pre_dln * xhat
// -- what layernorm_bwd_kernel did in a launch of its own, writing dx (4 E bytes per row) and gb (2 E) for this kernel to read back: here dx stays in the
// wave's registers until the norm2 backward at the end of the tile and gb goes straight into the LDS image (and to memory once, for the weight gradient).
// The prologue of tile t+1 runs at the END of tile t (its three rows requested before the norm2 phase of tile t, which hides their latency).
template <int MODE>  // 0: no prologue; 1: prologue; 2: prologue whose upstream gradient rows are picked through pre_row_map (dx_in may be null = zero in 1 and 2)
__global__ __launch_bounds__(FF_NT) void ffn_bwd_kernel(const FfnBwdArgs gin) {
	constexpr bool PRE = MODE != 0;
	FfnBwdArgs g = gin;
	if (g.row_limit) g.M = min(g.M, max(*g.row_limit, 0));
	extern __shared__ __attribute__((aligned(16))) char smem[];
	char* a1 = smem;                  // gb image [rows][1 KiB] bf16, 16-byte chunks XOR-swizzled by row & 15
	char* dl = smem;                  // dln image, same shape and swizzle, OVER the gb image: GEMM 1 has read it (barrier) before GEMM 2 writes
	char* a2 = smem + FF_A1;          // dh image [rows][256 B]
	float* gs = reinterpret_cast<float*>(smem + FF_A1 + FF_A2);          // gamma2 | pre_gamma
	char* w1s = smem + FF_A1 + FF_A2 + FF_G;  // W1^T [512 output columns][256 B], chunks XOR-swizzled by column & 15: 128 KiB, resident for the launch
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const int fr = lane & 15, fq = lane >> 4;
	const int ntiles = (g.M + FF_ROWS - 1) / FF_ROWS;
	int t = blockIdx.x;
	if (t >= ntiles) return;

	// GEMM 1's weights (this wave's 16 hidden columns x K = 512 of W2^T) stay in registers, GEMM 2's (W1^T) in LDS: both in registers (128 of the
	// 256 a wave has at two waves per SIMD) leave no room for a tile of prefetched rows -- the kernel spilled, and every scratch reload drains the
	// whole vector-memory queue.  W1^T costs 128 KiB of LDS reads per tile and workgroup (~0.4 us at 128 B/clk) against ~10 us per tile before.
	bf16x8 w2f[16];
#pragma unroll
	for (int ks = 0; ks < 16; ++ks) w2f[ks] = *reinterpret_cast<const bf16x8*>(g.w2t + (size_t)(16 * w + fr) * FF_E + ks * 32 + fq * 8);
	for (int q = tid; q < FF_E * FF_K / 8; q += FF_NT) {
		const int n = q >> 4, c = q & 15;
		*reinterpret_cast<bf16x8*>(w1s + n * 256 + ((c ^ (n & 15)) << 4)) = *reinterpret_cast<const bf16x8*>(g.w1t + (size_t)q * 8);
	}
	for (int i = tid; i < FF_E; i += FF_NT) {
		gs[i] = g.gamma2[i];
		if (PRE) gs[FF_E + i] = g.pre_gamma[i];
	}
	__syncthreads();

	typedef unsigned ff_u32x2 __attribute__((ext_vector_type(2)));
	typedef unsigned ff_u32x4 __attribute__((ext_vector_type(4)));
	auto srd = [&](void* p, unsigned row_bytes) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, p ? (unsigned)g.M * row_bytes : 0u, 0x00020000); };
	const __amdgpu_buffer_rsrc_t s_dh = srd(g.dh, FF_K * 2), s_dx = srd(g.dx_out, FF_E * 4), s_g = srd(g.g_out, FF_E * 2), s_gb = srd(g.gb_out, FF_E * 2);
	const __amdgpu_buffer_rsrc_t s_dxin = srd(const_cast<float*>(g.dx_in), FF_E * 4);  // a null dx_in is an empty buffer: every load returns zeros, no branch
	auto st8 = [](__amdgpu_buffer_rsrc_t r, bf16x4 v, unsigned off) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ff_u32x2, v), r, off, 0, 0); };

	// Clamped row indices, no branch around any load (the compiler's wait counts stay countable).
	auto row_of = [&](int tile, int i) {
		const int m = tile * FF_ROWS + FF_RPW * w + i;
		return m < g.M ? m : g.M - 1;
	};
	auto load_hpre = [&](bf16x4 (&hp)[FF_MT], int tile) {  // in the layout of GEMM 1's epilogue: row mt*16 + fr, hidden columns 16 w + 4 fq ..
#pragma unroll
		for (int mt = 0; mt < FF_MT; ++mt) {
			int m = tile * FF_ROWS + mt * 16 + fr;
			m = m < g.M ? m : g.M - 1;
			hp[mt] = *reinterpret_cast<const bf16x4*>(g.hpre + (size_t)m * FF_K + 16 * w + 4 * fq);
		}
	};
	auto load_bf16_rows = [&](bf16x4 (&r)[FF_RPW][2], const bf16* src, int tile) {  // LayerNorm layout: wave w rows RPW w .., element 256 c + 4 lane + e
#pragma unroll
		for (int i = 0; i < FF_RPW; ++i)
#pragma unroll
			for (int c = 0; c < 2; ++c) r[i][c] = *reinterpret_cast<const bf16x4*>(src + (size_t)row_of(tile, i) * FF_E + c * 256 + lane * 4);
	};
	auto load_f32_rows = [&](f32x4 (&r)[FF_RPW][2], const float* src, int tile) {
#pragma unroll
		for (int i = 0; i < FF_RPW; ++i)
#pragma unroll
			for (int c = 0; c < 2; ++c) r[i][c] = *reinterpret_cast<const f32x4*>(src + (size_t)row_of(tile, i) * FF_E + c * 256 + lane * 4);
	};
	// LayerNorm backward of one row held by one wave, the arithmetic of layernorm_bwd_kernel: dxr += rstd * (dy*gamma - mean(dy*gamma) - xhat * mean(dy*gamma*xhat)),
	// dgacc += dy * xhat (if the row counts).  dyr comes in as dy, xr as x; both are overwritten.
	auto ln_bwd_row = [&](float (&xr)[2][4], float (&dyr)[2][4], float (&dxr)[2][4], const float* gamma_lds, float (&dgacc)[2][4], bool counts) {
		float sum = 0.f;
#pragma unroll
		for (int c = 0; c < 2; ++c)
#pragma unroll
			for (int e = 0; e < 4; ++e) sum += xr[c][e];
		const float mean = wave_sum(sum) / (float)FF_E;
		float q = 0.f;
#pragma unroll
		for (int c = 0; c < 2; ++c)
#pragma unroll
			for (int e = 0; e < 4; ++e) {
				const float d = xr[c][e] - mean;
				q += d * d;
			}
		const float rstd = rsqrtf(wave_sum(q) / (float)FF_E + g.eps);
		float s1 = 0.f, s2 = 0.f;
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			const f32x4 gmc = *reinterpret_cast<const f32x4*>(gamma_lds + c * 256 + lane * 4);  // (from LDS at the point of use: held across the tile, registers run out)
#pragma unroll
			for (int e = 0; e < 4; ++e) {
				const float xhat = (xr[c][e] - mean) * rstd;
				const float dxh = dyr[c][e] * gmc[e];
				dgacc[c][e] += counts ? dyr[c][e] * xhat : 0.f;  // (LDS atomics instead of these 2 x 8 registers: ds_add_f32 ran the launch at 260 / 430 us)
				s1 += dxh;
				s2 += dxh * xhat;
				xr[c][e] = xhat;
				dyr[c][e] = dxh;
			}
		}
		s1 = wave_sum(s1) / (float)FF_E;
		s2 = wave_sum(s2) / (float)FF_E;
#pragma unroll
		for (int c = 0; c < 2; ++c)
#pragma unroll
			for (int e = 0; e < 4; ++e) dxr[c][e] += rstd * (dyr[c][e] - s1 - xr[c][e] * s2);
	};

	bf16x4 gcur[FF_RPW][2], gnxt[FF_RPW][2];  // gb rows of this / the next tile (PRE: gnxt unused, gcur formed by the prologue)
	bf16x4 hcur[FF_MT], hnxt[FF_MT];
	f32x4 dxt[FF_RPW][2];                     // the residual-stream gradient of this tile's rows: loaded (PRE = false) or formed by the prologue
	float dg[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dgp[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};

	// PRE prologue of one tile from its three requested rows: dxt, gcur (registers) and the gb rows in memory
	auto prologue = [&](f32x4 (&px)[FF_RPW][2], int tile, bool counts) {
		// px (the norm's input rows, needed first) was requested before the norm2 phase of the previous tile; the other two rows only here: beside the resident weights,
		// the norm2 phase's rows and two sets of dgamma accumulators there are no registers to park all three (spills: every scratch reload drains the vector-memory
		// queue; accumulators in LDS: ds_add_f32 ran the launch at 430 us)
		bf16x4 pd[FF_RPW][2];
		f32x4 pdx[FF_RPW][2];
		int has_dy[FF_RPW];
#pragma unroll
		for (int i = 0; i < FF_RPW; ++i) {
			const int m = row_of(tile, i);
			int r = m;
			if constexpr (MODE == 2) r = g.pre_row_map[m];
			has_dy[i] = r >= 0;
			r = r < 0 ? 0 : r;
#pragma unroll
			for (int c = 0; c < 2; ++c) {
				pd[i][c] = *reinterpret_cast<const bf16x4*>(g.pre_dln + (size_t)r * FF_E + c * 256 + lane * 4);
				pdx[i][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s_dxin, ((unsigned)m * FF_E + c * 256 + lane * 4) * 4u, 0, 0));
			}
		}
#pragma unroll
		for (int i = 0; i < FF_RPW; ++i) {
			const int m = tile * FF_ROWS + FF_RPW * w + i;
			float xr[2][4], dyr[2][4], dxr[2][4];
#pragma unroll
			for (int c = 0; c < 2; ++c)
#pragma unroll
				for (int e = 0; e < 4; ++e) {
					xr[c][e] = px[i][c][e];
					dyr[c][e] = has_dy[i] ? (float)pd[i][c][e] : 0.f;
					dxr[c][e] = pdx[i][c][e];
				}
			ln_bwd_row(xr, dyr, dxr, gs + FF_E, dgp, counts && m < g.M);
#pragma unroll
			for (int c = 0; c < 2; ++c) {
				dxt[i][c] = (f32x4){dxr[c][0], dxr[c][1], dxr[c][2], dxr[c][3]};
				float sc[4];
				dropout_scale4_branchless(g.drop_pre, (uint64_t)m * FF_E + c * 256 + lane * 4, sc);
				gcur[i][c] = (bf16x4){(bf16)(dxr[c][0] * sc[0]), (bf16)(dxr[c][1] * sc[1]), (bf16)(dxr[c][2] * sc[2]), (bf16)(dxr[c][3] * sc[3])};
				st8(s_gb, gcur[i][c], counts ? ((unsigned)m * FF_E + c * 256 + lane * 4) * 2u : 0xFFFFFFF0u);  // the surplus prologue behind the last tile stores out of range: dropped, no branch
			}
			__builtin_amdgcn_sched_barrier(0);
		}
	};

	load_hpre(hcur, t);
	if constexpr (PRE) {
		f32x4 px0[FF_RPW][2];
		load_f32_rows(px0, g.pre_x, t);
		prologue(px0, t, true);
	} else {
		load_bf16_rows(gcur, g.gb, t);
	}

	for (; t < ntiles; t += gridDim.x) {
		const int tn = t + (int)gridDim.x;
		const int tnc = tn < ntiles ? tn : ntiles - 1;
		load_hpre(hnxt, tnc);
		if constexpr (!PRE) {
			load_bf16_rows(gnxt, g.gb, tnc);
			load_f32_rows(dxt, g.dx_in, t);
		}
		f32x4 xm[FF_RPW][2];  // the fp32 rows of this tile's norm2 phase: both GEMM phases to arrive
		load_f32_rows(xm, g.xmid, t);
		const int m0 = t * FF_ROWS;

		// ---- gb rows -> A1 image ----
#pragma unroll
		for (int i = 0; i < FF_RPW; ++i) {
			const int row = FF_RPW * w + i;
#pragma unroll
			for (int c = 0; c < 2; ++c) {
				const int chunk = 32 * c + (lane >> 1);
				*reinterpret_cast<bf16x4*>(a1 + row * 1024 + ((chunk ^ (row & 15)) << 4) + (lane & 1) * 8) = gcur[i][c];
			}
		}
		lds_barrier();

		// ---- GEMM 1 + GELU' (+ dropout mask of the forward GELU output): dh, this wave's 16 hidden columns ----
#pragma unroll
		for (int mt = 0; mt < FF_MT; ++mt) {
			const int row = mt * 16 + fr, m = m0 + row;
			const char* rowp = a1 + row * 1024;
			f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
			for (int kh = 0; kh < 2; ++kh) {
				bf16x8 af[8];
#pragma unroll
				for (int ks = 0; ks < 8; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(rowp + ((((kh * 8 + ks) * 4 + fq) ^ fr) << 4));
#pragma unroll
				for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[kh * 8 + ks], af[ks], acc, 0, 0, 0);
			}
			const int n = 16 * w + 4 * fq;
			float sc[4];
			dropout_scale4_branchless(g.drop_gelu, (uint64_t)m * FF_K + n, sc);
			bf16x4 d;
#pragma unroll
			for (int r = 0; r < 4; ++r) d[r] = (bf16)gelu_bwd_elem(acc[r], sc[r], (float)hcur[mt][r]);  // as epilogue4<GELU_BWD_BF16>
			const int chunk = 2 * w + (fq >> 1);
			*reinterpret_cast<bf16x4*>(a2 + row * 256 + ((chunk ^ fr) << 4) + (fq & 1) * 8) = d;
			st8(s_dh, d, ((unsigned)m * FF_K + n) * 2u);
		}
		lds_barrier();

		// ---- GEMM 2: dln = bf16(dh W1), this wave's 64 columns -> dln image ----
#pragma unroll
		for (int mt = 0; mt < FF_MT; ++mt) {
			const int row = mt * 16 + fr;
			const char* rowp = a2 + row * 256;
			bf16x8 af[4];
#pragma unroll
			for (int ks = 0; ks < 4; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(rowp + (((ks * 4 + fq) ^ fr) << 4));
#pragma unroll
			for (int nt = 0; nt < 4; ++nt) {
				f32x4 acc = {0.f, 0.f, 0.f, 0.f};
				const char* wrow = w1s + (64 * w + 16 * nt + fr) * 256;
				bf16x8 wf[4];
#pragma unroll
				for (int ks = 0; ks < 4; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(wrow + (((ks * 4 + fq) ^ fr) << 4));
#pragma unroll
				for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], af[ks], acc, 0, 0, 0);
				const int n = 64 * w + 16 * nt + 4 * fq;
				const bf16x4 o = {(bf16)acc[0], (bf16)acc[1], (bf16)acc[2], (bf16)acc[3]};  // as epilogue4<STORE_BF16>
				*reinterpret_cast<bf16x4*>(dl + row * 1024 + (((n >> 3) ^ fr) << 4) + ((n >> 2) & 1) * 8) = o;
			}
		}
		f32x4 px[FF_RPW][2];
		if constexpr (PRE) load_f32_rows(px, g.pre_x, tnc);  // the next tile's prologue: its first operand, with the whole norm2 phase to arrive
		lds_barrier();

		// ---- norm2 backward on the tile's rows (wave w: rows RPW w ..), every row selected ----
#pragma unroll
		for (int i = 0; i < FF_RPW; ++i) {
			const int row = FF_RPW * w + i, m = m0 + row;
			float xr[2][4], dyr[2][4], dxr[2][4];
#pragma unroll
			for (int c = 0; c < 2; ++c) {
				const int chunk = 32 * c + (lane >> 1);
				const bf16x4 dy4 = *reinterpret_cast<const bf16x4*>(dl + row * 1024 + ((chunk ^ (row & 15)) << 4) + (lane & 1) * 8);
#pragma unroll
				for (int e = 0; e < 4; ++e) {
					xr[c][e] = xm[i][c][e];
					dyr[c][e] = (float)dy4[e];
					dxr[c][e] = dxt[i][c][e];
				}
			}
			ln_bwd_row(xr, dyr, dxr, gs, dg, m < g.M);
#pragma unroll
			for (int c = 0; c < 2; ++c) {
				const unsigned off = (unsigned)m * FF_E + c * 256 + lane * 4;
				__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ff_u32x4, (f32x4){dxr[c][0], dxr[c][1], dxr[c][2], dxr[c][3]}), s_dx, off * 4u, 0, 2);
				float sc[4];
				dropout_scale4_branchless(g.drop_g, (uint64_t)m * FF_E + c * 256 + lane * 4, sc);
				const bf16x4 o = {(bf16)(dxr[c][0] * sc[0]), (bf16)(dxr[c][1] * sc[1]), (bf16)(dxr[c][2] * sc[2]), (bf16)(dxr[c][3] * sc[3])};
				st8(s_g, o, off * 2u);
			}
			__builtin_amdgcn_sched_barrier(0);  // one row at a time: interleaved, the rows' temporaries spill beside the resident weights
		}
		if constexpr (PRE) {
			// dx_in rows of the NEXT tile were read above, before this tile's dx_out stores: other rows, and (dx_out == dx_in in place) only this workgroup
			// ever touches them.  The image is free for the next tile's gb rows once every wave has read its dln rows: the barrier at the top covers it.
			prologue(px, tnc, tn < ntiles);
		} else {
#pragma unroll
			for (int i = 0; i < FF_RPW; ++i)
#pragma unroll
				for (int c = 0; c < 2; ++c) gcur[i][c] = gnxt[i][c];
		}
#pragma unroll
		for (int mt = 0; mt < FF_MT; ++mt) hcur[mt] = hnxt[mt];
	}

	// dgamma2 (and the prologue's dgamma): the eight waves' column sums through LDS, one atomic per column and workgroup (as layernorm_bwd_kernel)
	float* red = reinterpret_cast<float*>(smem);  // [8][512] fp32 = 16 KiB over the gb / dln image
#pragma unroll
	for (int pass = 0; pass < (PRE ? 2 : 1); ++pass) {
		lds_barrier();
#pragma unroll
		for (int c = 0; c < 2; ++c)
#pragma unroll
			for (int e = 0; e < 4; ++e) red[w * FF_E + c * 256 + lane * 4 + e] = pass == 0 ? dg[c][e] : dgp[c][e];
		lds_barrier();
		float* dst = pass == 0 ? g.dgamma2 : g.pre_dgamma;
		for (int e = tid; e < FF_E; e += FF_NT) {
			float tsum = 0.f;
#pragma unroll
			for (int ww = 0; ww < 8; ++ww) tsum += red[ww * FF_E + e];
			if (tsum != 0.f) atomicAdd(dst + e, tsum);
		}
	}
}

}  // namespace

extern "C" int novic_ffn_fused_supported(int E, int Kf) { return (E == FF_E && Kf == FF_K) ? 1 : 0; }

extern "C" int novic_ffn_fwd(const float* xmid, const float* gamma2, const void* w1_bf16, const void* w2_bf16, const float* gamma_next, float* x_out, void* ln2_bf16,
                             void* hpre_bf16, void* hact_bf16, void* ln_next_bf16, int M, int E, int Kf, float eps, float drop_p, uint64_t seed, uint32_t site_gelu,
                             uint32_t site_out, const int32_t* row_limit, hipStream_t stream) {
	NOVIC_CHECK(xmid && gamma2 && w1_bf16 && w2_bf16 && x_out, "novic_ffn_fwd: null pointer");
	NOVIC_CHECK(E == FF_E && Kf == FF_K, "novic_ffn_fwd: built for hidden 512 / feed-forward 128 (novic_ffn_fused_supported)");
	NOVIC_CHECK(!gamma_next == !ln_next_bf16, "novic_ffn_fwd: gamma_next and ln_next go together");
	NOVIC_CHECK(M >= 0, "novic_ffn_fwd: negative row count");
	NOVIC_CHECK((uint64_t)M * FF_E * 4 < 0xFFFFFFF0ull, "novic_ffn_fwd: M * 512 * 4 bytes must stay below 4 GiB (32-bit buffer descriptors and row offsets)");
	NOVIC_CHECK((((uintptr_t)xmid | (uintptr_t)gamma2 | (uintptr_t)w1_bf16 | (uintptr_t)w2_bf16 | (uintptr_t)x_out | (uintptr_t)gamma_next) & 15) == 0 &&
	            (((uintptr_t)ln2_bf16 | (uintptr_t)hpre_bf16 | (uintptr_t)hact_bf16 | (uintptr_t)ln_next_bf16) & 7) == 0, "novic_ffn_fwd: misaligned operand");
	if (M == 0) return 0;
	FfnArgs g;
	g.xmid = xmid; g.gamma2 = gamma2; g.w1 = (const bf16*)w1_bf16; g.w2 = (const bf16*)w2_bf16; g.gamma_next = gamma_next;
	g.x_out = x_out; g.ln2 = (bf16*)ln2_bf16; g.hpre = (bf16*)hpre_bf16; g.hact = (bf16*)hact_bf16; g.ln_next = (bf16*)ln_next_bf16;
	g.M = M; g.eps = eps;
	g.drop_gelu = {drop_p, (uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32), site_gelu};
	g.drop_out = {drop_p, (uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32), site_out};
	g.row_limit = row_limit;
	static std::atomic<bool> attr_done{false};
	if (!attr_done) {
		(void)hipFuncSetAttribute((const void*)ffn_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS);
		(void)hipFuncSetAttribute((const void*)ffn_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS);
		attr_done = true;
	}
	const int ntiles = (M + FF_ROWS - 1) / FF_ROWS;
	if (gamma_next) hipLaunchKernelGGL(ffn_fwd_kernel<true>, dim3(ntiles < 256 ? ntiles : 256), dim3(FF_NT), FF_LDS, stream, g);
	else hipLaunchKernelGGL(ffn_fwd_kernel<false>, dim3(ntiles < 256 ? ntiles : 256), dim3(FF_NT), FF_LDS, stream, g);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

static int ffn_bwd_launch(FfnBwdArgs& g, int mode, int M, hipStream_t stream) {
	static std::atomic<bool> attr_done{false};
	if (!attr_done) {
		(void)hipFuncSetAttribute((const void*)ffn_bwd_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_BWD_LDS);
		(void)hipFuncSetAttribute((const void*)ffn_bwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_BWD_LDS);
		(void)hipFuncSetAttribute((const void*)ffn_bwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_BWD_LDS);
		attr_done = true;
	}
	const int ntiles = (M + FF_ROWS - 1) / FF_ROWS;
	const dim3 grid(ntiles < 256 ? ntiles : 256), block(FF_NT);
	if (mode == 2) hipLaunchKernelGGL(ffn_bwd_kernel<2>, grid, block, FF_BWD_LDS, stream, g);
	else if (mode == 1) hipLaunchKernelGGL(ffn_bwd_kernel<1>, grid, block, FF_BWD_LDS, stream, g);
	else hipLaunchKernelGGL(ffn_bwd_kernel<0>, grid, block, FF_BWD_LDS, stream, g);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_ffn_bwd(const void* gb_bf16, const void* hpre_bf16, const float* xmid, const float* dx_in, const float* gamma2, const void* w2t_bf16, const void* w1t_bf16,
                             void* dh_bf16, float* dx_out, void* g_out_bf16, float* dgamma2, int M, int E, int Kf, float eps, float drop_p, uint64_t seed, uint32_t site_gelu,
                             uint32_t site_g, const int32_t* row_limit, hipStream_t stream) {
	NOVIC_CHECK(gb_bf16 && hpre_bf16 && xmid && dx_in && gamma2 && w2t_bf16 && w1t_bf16 && dh_bf16 && dx_out && g_out_bf16 && dgamma2, "novic_ffn_bwd: null pointer");
	NOVIC_CHECK(E == FF_E && Kf == FF_K, "novic_ffn_bwd: built for hidden 512 / feed-forward 128 (novic_ffn_fused_supported)");
	NOVIC_CHECK(M >= 0, "novic_ffn_bwd: negative row count");
	NOVIC_CHECK((uint64_t)M * FF_E * 4 < 0xFFFFFFF0ull, "novic_ffn_bwd: M * 512 * 4 bytes must stay below 4 GiB (32-bit buffer descriptors and row offsets)");
	NOVIC_CHECK((((uintptr_t)xmid | (uintptr_t)dx_in | (uintptr_t)gamma2 | (uintptr_t)w2t_bf16 | (uintptr_t)w1t_bf16 | (uintptr_t)dx_out) & 15) == 0 &&
	            (((uintptr_t)gb_bf16 | (uintptr_t)hpre_bf16 | (uintptr_t)dh_bf16 | (uintptr_t)g_out_bf16) & 7) == 0, "novic_ffn_bwd: misaligned operand");
	NOVIC_CHECK(gb_bf16 != g_out_bf16, "novic_ffn_bwd: g_out must not alias gb (other tiles' rows of gb are still being read)");
	if (M == 0) return 0;
	FfnBwdArgs g = {};
	g.gb = (const bf16*)gb_bf16; g.hpre = (const bf16*)hpre_bf16; g.xmid = xmid; g.dx_in = dx_in; g.gamma2 = gamma2; g.w2t = (const bf16*)w2t_bf16; g.w1t = (const bf16*)w1t_bf16;
	g.dh = (bf16*)dh_bf16; g.dx_out = dx_out; g.g_out = (bf16*)g_out_bf16; g.dgamma2 = dgamma2;
	g.M = M; g.eps = eps;
	g.drop_gelu = {drop_p, (uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32), site_gelu};
	g.drop_g = {drop_p, (uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32), site_g};
	g.drop_pre = g.drop_g;
	g.row_limit = row_limit;
	return ffn_bwd_launch(g, 0, M, stream);
}

extern "C" int novic_ffn_bwd_ln(const void* pre_dln_bf16, const int32_t* pre_row_map, const float* pre_x, const float* pre_gamma, float* pre_dgamma, void* gb_out_bf16, uint32_t site_pre,
                                const void* hpre_bf16, const float* xmid, const float* dx_in, const float* gamma2, const void* w2t_bf16, const void* w1t_bf16, void* dh_bf16,
                                float* dx_out, void* g_out_bf16, float* dgamma2, int M, int E, int Kf, float eps, float drop_p, uint64_t seed, uint32_t site_gelu, uint32_t site_g,
                                const int32_t* row_limit, hipStream_t stream) {
	NOVIC_CHECK(pre_dln_bf16 && pre_x && pre_gamma && pre_dgamma && gb_out_bf16, "novic_ffn_bwd_ln: null pointer (prologue operands)");
	NOVIC_CHECK(hpre_bf16 && xmid && gamma2 && w2t_bf16 && w1t_bf16 && dh_bf16 && dx_out && g_out_bf16 && dgamma2, "novic_ffn_bwd_ln: null pointer");
	NOVIC_CHECK(E == FF_E && Kf == FF_K, "novic_ffn_bwd_ln: built for hidden 512 / feed-forward 128 (novic_ffn_fused_supported)");
	NOVIC_CHECK(M >= 0, "novic_ffn_bwd_ln: negative row count");
	NOVIC_CHECK((uint64_t)M * FF_E * 4 < 0xFFFFFFF0ull, "novic_ffn_bwd_ln: M * 512 * 4 bytes must stay below 4 GiB (32-bit buffer descriptors and row offsets)");
	NOVIC_CHECK((((uintptr_t)xmid | (uintptr_t)dx_in | (uintptr_t)gamma2 | (uintptr_t)w2t_bf16 | (uintptr_t)w1t_bf16 | (uintptr_t)dx_out | (uintptr_t)pre_x | (uintptr_t)pre_gamma) & 15) == 0 &&
	            (((uintptr_t)pre_dln_bf16 | (uintptr_t)gb_out_bf16 | (uintptr_t)hpre_bf16 | (uintptr_t)dh_bf16 | (uintptr_t)g_out_bf16) & 7) == 0, "novic_ffn_bwd_ln: misaligned operand");
	NOVIC_CHECK(gb_out_bf16 != g_out_bf16 && pre_dln_bf16 != g_out_bf16 && pre_dln_bf16 != gb_out_bf16, "novic_ffn_bwd_ln: pre_dln, gb_out and g_out must be three buffers");
	if (M == 0) return 0;
	FfnBwdArgs g = {};
	g.pre_dln = (const bf16*)pre_dln_bf16; g.pre_row_map = pre_row_map; g.pre_x = pre_x; g.pre_gamma = pre_gamma; g.pre_dgamma = pre_dgamma; g.gb_out = (bf16*)gb_out_bf16;
	g.hpre = (const bf16*)hpre_bf16; g.xmid = xmid; g.dx_in = dx_in; g.gamma2 = gamma2; g.w2t = (const bf16*)w2t_bf16; g.w1t = (const bf16*)w1t_bf16;
	g.dh = (bf16*)dh_bf16; g.dx_out = dx_out; g.g_out = (bf16*)g_out_bf16; g.dgamma2 = dgamma2;
	g.M = M; g.eps = eps;
	g.drop_gelu = {drop_p, (uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32), site_gelu};
	g.drop_g = {drop_p, (uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32), site_g};
	g.drop_pre = {drop_p, (uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32), site_pre};
	g.row_limit = row_limit;
	return ffn_bwd_launch(g, pre_row_map ? 2 : 1, M, stream);
}
