// Tall GEMMs with a narrow output: C[M][128] = epilogue(A[M][512] x W[128][512]^T), M in the tens of thousands -- linear1 of the decoder's
// feed-forward block (512 -> 128, GELU epilogue) and the input gradient of linear2 (GELU' epilogue).  10.7 GFLOP against 126 MB of operands:
// HBM-bound (18 us at 6.8 TB/s), yet the 128^2-tile kernel takes 47-69 us on them -- 640 workgroups in 1.25 rounds, each paying 1.7 us of
// prologue and 4-9 us of epilogue around 8 us of K loop, every one re-staging its 128 KB weight tile.
//
// Here the weight matrix never moves: a persistent workgroup (8 waves = 2 row halves x 4 column quarters) loads W once, as MFMA fragments, into
// registers (a wave's 32 columns x K = 512: 32 fragments = 128 VGPRs) and streams 64-row tiles of A through a double-buffered LDS image filled by
// LDS-DMA (one instruction = one whole 1 KiB row; 16-byte chunks XOR-swizzled by row & 15 ON THE SOURCE SIDE so that the 16 rows x 4 k-chunks of
// a fragment read hit 16 different bank groups).  Per tile a CU fetches 64 KiB and issues 2 x 64 MFMAs per SIMD (1 us): the fetch is what takes
// the time, as it should.  Same MFMA, K accumulated in the same order, the shared epilogue code: results are bit-identical to the 128^2 kernel.
#include "gemm_epilogue.hpp"

namespace {

constexpr int SK_N = 128, SK_K = 512, SK_NKS = SK_K / 32, SK_ROWS = 64, SK_ROWB = SK_K * 2, SK_TILE = SK_ROWS * SK_ROWB;  // 64 KiB per A tile
constexpr unsigned SK_OOB = 0xFFFFFFF0u;

struct SkinnyArgs {
	const bf16* A;
	const bf16* W;
	int M, lda, ldw;  // (the kernels clamp M to *ep.row_limit when that is set)
	unsigned a_bytes, w_bytes;
	int n_blocks, n_total;  // skinny_n128_kernel as a column block of a wider GEMM (RESID_F32: out-proj [M x 512 x 512] = 4 blocks of 128 columns); 1, 128 otherwise
	novic_epilogue_t ep;
};

typedef __attribute__((address_space(3))) void* sk_lds_ptr_t;
typedef unsigned int sk_u32x4 __attribute__((ext_vector_type(4)));

// NMT = 16-row groups per wave: 2 = the workgroup's 8 waves are 2 row halves x 4 column quarters of a 128-column block; 4 = 1 x 8: every wave takes all 64 rows of
// the tile and the block is 256 columns wide (as a column block of a 512-wide GEMM: TWO blocks share a row stream instead of four, so an A tile is pulled into
// LDS by two CUs instead of four -- what a CU can pull is what bounds these launches).
template <int EPI, int NMT = 2>
__global__ __launch_bounds__(512) void skinny_n128_kernel(const SkinnyArgs gin) {
	SkinnyArgs g = gin;
	if (g.ep.row_limit) g.M = min(g.M, max(*g.ep.row_limit, 0));
	extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][64 rows][1 KiB]
	constexpr int BW = NMT == 2 ? SK_N : 2 * SK_N;  // columns of this workgroup's block
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const int wm = NMT == 2 ? w >> 2 : 0, wn = NMT == 2 ? w & 3 : w, fr = lane & 15, fq = lane >> 4;
	const __amdgpu_buffer_rsrc_t sa = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.A), 0, g.a_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t sw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.W), 0, g.w_bytes, 0x00020000);
	// Column blocks (n_blocks > 1): workgroups b, b + 8, b + 16, ... share a row stream -- and, with it, an XCD (b & 7), so that the A tiles the
	// blocks of a stream all read come out of that XCD's L2 after the first of them fetched them.
	const int nb = g.n_blocks, cb = nb > 1 ? ((int)blockIdx.x >> 3) % nb : 0;
	const int stream = nb > 1 ? ((int)blockIdx.x & 7) + 8 * ((int)blockIdx.x / (8 * nb)) : (int)blockIdx.x;
	const int nstreams = nb > 1 ? (int)gridDim.x / nb : (int)gridDim.x;
	const int n_off = cb * BW;

	// the wave's 32 columns of W, all of K, as "first operand" fragments.  Fragment row j of column tile nt holds column (j / 4) * 8 + nt * 4 + j % 4
	// of the wave's 32, so that after the (swapped) MFMA a lane owns, of its output row, the 8 CONSECUTIVE columns 8 fq .. 8 fq + 7 (first four in tile 0,
	// last four in tile 1): 16-byte epilogue accesses, four lanes = 64 contiguous bytes of a row.
	bf16x8 wf[2][SK_NKS];
#pragma unroll
	for (int nt = 0; nt < 2; ++nt)
#pragma unroll
		for (int ks = 0; ks < SK_NKS; ++ks) {
			const unsigned off = (unsigned)(((n_off + wn * 32 + (fr >> 2) * 8 + nt * 4 + (fr & 3)) * g.ldw + ks * 32 + fq * 8) * 2);
			wf[nt][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(sw, off, 0, 0));
		}

	const int ntiles = (g.M + SK_ROWS - 1) / SK_ROWS;
	// wave w stages rows 8 w .. 8 w + 7 of a tile, one LDS-DMA instruction per row: lane L fills slot L, i.e. fetches chunk L ^ (row & 15)
	auto stage = [&](int tile, int buf) {
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const int row = w * 8 + i, m = tile * SK_ROWS + row;
			const unsigned off = m < g.M ? ((unsigned)m * (unsigned)g.lda + (unsigned)((lane ^ (row & 15)) << 3)) * 2u : SK_OOB;
			__builtin_amdgcn_raw_ptr_buffer_load_lds(sa, (sk_lds_ptr_t)(smem + buf * SK_TILE + row * SK_ROWB), 16, off, 0, 0, 0);
		}
	};

	int t = stream;
	if (t >= ntiles) return;
	stage(t, 0);
	int buf = 0;
	for (; t < ntiles; t += nstreams, buf ^= 1) {
		const bool has_next = t + nstreams < ntiles;
		if (has_next) {
			stage(t + nstreams, buf ^ 1);  // that buffer was last read two barriers ago
			asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // everything older than the 8 rows just requested: this tile's rows (and the last epilogue)
		} else {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}
		__builtin_amdgcn_s_barrier();
		asm volatile("" ::: "memory");

		// RESID_F32: this tile's residual, requested before its MFMAs (8 consecutive columns per lane: four lanes = one 128-byte line of a row).  Requesting
		// it one tile ahead instead (right behind the next tile's 8 rows, then a counted vmcnt(12)) is sound by the in-order rule and was re-built in round 2:
		// bit-identical on the GPU, but 77 us instead of 72 -- the rnext -> rres register rotation makes hipcc wait vmcnt(0) before the back-edge, which drains
		// the next tile's LDS-DMA as well.  (Round 1's wrong result with "the same" change did not reproduce; DESIGN.md section 4, tools/audit_vmcnt.py.)
		f32x4 rres[NMT][2];
		if (EPI == NOVIC_EPI_RESID_F32) {
#pragma unroll
			for (int mt = 0; mt < NMT; ++mt) {
				const int m = t * SK_ROWS + wm * 32 + mt * 16 + fr;
				const float* R = (const float*)g.ep.resid + (size_t)(m < g.M ? m : 0) * g.ep.ldr + n_off + wn * 32 + fq * 8;
				rres[mt][0] = *reinterpret_cast<const f32x4*>(R);
				rres[mt][1] = *reinterpret_cast<const f32x4*>(R + 4);
			}
		}
		f32x4 acc[NMT][2];
#pragma unroll
		for (int mt = 0; mt < NMT; ++mt) {
			acc[mt][0] = acc[mt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
			const char* rowp = smem + buf * SK_TILE + (wm * 32 + mt * 16 + fr) * SK_ROWB;
			bf16x8 af[SK_NKS];  // the row group's 16 fragments requested together (read one at a time, every MFMA pair waited out a full LDS latency)
#pragma unroll
			for (int ks = 0; ks < SK_NKS; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(rowp + (((ks * 4 + fq) ^ fr) << 4));
#pragma unroll
			for (int ks = 0; ks < SK_NKS; ++ks) {
				acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][ks], af[ks], acc[mt][0], 0, 0, 0);
				acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][ks], af[ks], acc[mt][1], 0, 0, 0);
			}
			__builtin_amdgcn_sched_barrier(0);
		}
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();  // every wave is done reading this buffer: the next iteration may refill it
		asm volatile("" ::: "memory");

		// lane (fr, fq) holds rows mt*16 + fr, columns 8 fq .. 8 fq + 7 of the wave's 32 x 32 block: acc[mt][0] the first four, acc[mt][1] the last four.
		// Arithmetic and rounding points are those of epilogue4 (gemm_epilogue.hpp); plain (cached) 16-byte accesses -- the 64-byte row pieces of the
		// four column quarters merge into whole lines in L2, and the next GEMM reads the result straight away.
		epilogue_dispatch<EPI>(g.ep, [&](auto act_c, auto drop_c) {
			constexpr int DROP = decltype(drop_c)::value;
			const DropoutDesc d = {g.ep.drop_p, g.ep.seed_lo, g.ep.seed_hi, g.ep.drop_site};
#pragma unroll
			for (int mt = 0; mt < NMT; ++mt) {
				const int m = t * SK_ROWS + wm * 32 + mt * 16 + fr, n = n_off + wn * 32 + fq * 8;
				if (m >= g.M) continue;
				float v[8] = {acc[mt][0][0], acc[mt][0][1], acc[mt][0][2], acc[mt][0][3], acc[mt][1][0], acc[mt][1][1], acc[mt][1][2], acc[mt][1][3]};
				float sc[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
				if (DROP != 0 && EPI != NOVIC_EPI_STORE_BF16) {
					float s0[4], s1[4];
					dropout_scale4(d, (uint64_t)m * g.n_total + n, s0);
					dropout_scale4(d, (uint64_t)m * g.n_total + n + 4, s1);
#pragma unroll
					for (int i = 0; i < 4; ++i) { sc[i] = s0[i]; sc[4 + i] = s1[i]; }
				}
				const size_t o = (size_t)m * g.ep.ldc + n;
				if (EPI == NOVIC_EPI_RESID_F32) {  // out = resid + dropout(bf16(acc + bias)), fp32: as epilogue4<RESID_F32>
					if (g.ep.bias) {
						const f32x4 b0 = *reinterpret_cast<const f32x4*>((const float*)g.ep.bias + n), b1 = *reinterpret_cast<const f32x4*>((const float*)g.ep.bias + n + 4);
#pragma unroll
						for (int i = 0; i < 4; ++i) { v[i] += b0[i]; v[4 + i] += b1[i]; }
					}
					float o0[4], o1[4];
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						o0[i] = rres[mt][0][i] + bf16_round(v[i]) * sc[i];
						o1[i] = rres[mt][1][i] + bf16_round(v[4 + i]) * sc[4 + i];
					}
					st_f32x4((float*)g.ep.c + o, o0, true, 4);
					st_f32x4((float*)g.ep.c + o + 4, o1, true, 4);
					continue;
				}
				bf16x8 out;
				if (EPI == NOVIC_EPI_STORE_BF16) {
					if (g.ep.bias) {
						const f32x4 b0 = *reinterpret_cast<const f32x4*>((const float*)g.ep.bias + n), b1 = *reinterpret_cast<const f32x4*>((const float*)g.ep.bias + n + 4);
#pragma unroll
						for (int i = 0; i < 4; ++i) { v[i] += b0[i]; v[4 + i] += b1[i]; }
					}
#pragma unroll
					for (int i = 0; i < 8; ++i) out[i] = (bf16)v[i];
				} else if (EPI == NOVIC_EPI_GELU_BF16) {
					bf16x8 pre;
#pragma unroll
					for (int i = 0; i < 8; ++i) {
						const float pr = bf16_round(v[i]);
						pre[i] = (bf16)pr;
						out[i] = (bf16)gelu_fwd_elem(pr, sc[i]);
					}
					if (g.ep.c2) *reinterpret_cast<bf16x8*>((bf16*)g.ep.c2 + o) = pre;
				} else {  // GELU_BWD: c = bf16( bf16(acc) * dropmask * gelu'(hpre) )
					const bf16x8 h = *reinterpret_cast<const bf16x8*>((const bf16*)g.ep.resid + (size_t)m * g.ep.ldr + n);
#pragma unroll
					for (int i = 0; i < 8; ++i) out[i] = (bf16)gelu_bwd_elem(v[i], sc[i], (float)h[i]);
				}
				*reinterpret_cast<bf16x8*>((bf16*)g.ep.c + o) = out;
			}
		});
	}
}

template <int EPI, int NMT = 2>
void launch_skinny(const SkinnyArgs& g, int grid, hipStream_t stream) {
	static std::atomic<bool> attr_done{false};
	if (!attr_done) {
		(void)hipFuncSetAttribute((const void*)skinny_n128_kernel<EPI, NMT>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SK_TILE);
		attr_done = true;
	}
	hipLaunchKernelGGL((skinny_n128_kernel<EPI, NMT>), dim3(grid), dim3(512), 2 * SK_TILE, stream, g);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The other narrow GEMM of the feed-forward block: C[M][512] (fp32) = resid + dropout(bf16(A[M][128] x W[512][128]^T + bias)) -- linear2 with the
// residual add.  2 x 128 K-steps of MFMA against 8 bytes of HBM traffic per output element: a streaming kernel (357 MB, 52 us at 6.8 TB/s) that the
// 128^2 kernel runs in 70-95 us (2560 workgroups, each 4.5 us of prologue and 5 us of epilogue around 2.4 us of K loop).  Here W (512 x 128,
// 128 KiB) is parked in LDS once per persistent workgroup, 32-row tiles of A follow it through a small double buffer (LDS-DMA), and the residual
// of tile i+1 (32 rows x 2 KiB, 8 x 16 bytes per lane) is requested into registers while tile i is multiplied and written -- with the weights in
// registers instead (as in the kernel above) there was no room for that second register set and the residual's HBM latency stayed exposed
// (75 us).  Natural column order: for fp32 output the four lanes of a row cover 64 contiguous bytes per access (the 8-consecutive-columns
// permutation of the bf16 kernels leaves 16-byte pieces 32 bytes apart: 104 us).
constexpr int SR_N = 512, SR_K = 128, SR_NKS = SR_K / 32, SR_ROWS = 32, SR_ROWB = SR_K * 2, SR_TILE = SR_ROWS * SR_ROWB;  // 8 KiB per A tile
constexpr int SR_WBYTES = SR_N * SR_ROWB;                                                                               // 128 KiB of weights

__global__ __launch_bounds__(512) void skinny_k128_resid_kernel(const SkinnyArgs gin) {
	SkinnyArgs g = gin;
	if (g.ep.row_limit) g.M = min(g.M, max(*g.ep.row_limit, 0));
	extern __shared__ __attribute__((aligned(16))) char smem[];  // [512 rows][256 B] W | [2][32 rows][256 B] A
	char* atile = smem + SR_WBYTES;
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const int wm = w >> 2, wn = w & 3, fr = lane & 15, fq = lane >> 4;
	const __amdgpu_buffer_rsrc_t sa = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.A), 0, g.a_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t sw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.W), 0, g.w_bytes, 0x00020000);
	const int ntiles = (g.M + SR_ROWS - 1) / SR_ROWS;
	int t = blockIdx.x;
	if (t >= ntiles) return;

	// one LDS-DMA instruction = 1 KiB = 4 rows of 256 B (16 chunks each); slot c of row r holds chunk c ^ (r & 15), for W and for A alike
	const int sub = lane >> 4, slot = lane & 15;
#pragma unroll
	for (int i = 0; i < 16; ++i) {  // W: 128 instructions, 16 per wave
		const int row = (w * 16 + i) * 4 + sub;
		__builtin_amdgcn_raw_ptr_buffer_load_lds(sw, (sk_lds_ptr_t)(smem + (w * 16 + i) * 1024), 16, (unsigned)((row * g.ldw + ((slot ^ (row & 15)) << 3)) * 2), 0, 0, 0);
	}
	auto stage = [&](int tile, int buf) {  // A: wave w stages rows 4 w .. 4 w + 3
		const int row = w * 4 + sub, m = tile * SR_ROWS + row;
		const unsigned off = m < g.M ? ((unsigned)m * (unsigned)g.lda + (unsigned)((slot ^ (row & 15)) << 3)) * 2u : SK_OOB;
		__builtin_amdgcn_raw_ptr_buffer_load_lds(sa, (sk_lds_ptr_t)(atile + buf * SR_TILE + w * 1024), 16, off, 0, 0, 0);
	};
	// lane (fr, fq) owns, of tile row wm*16 + fr, columns wn*128 + nt*16 + fq*4 .. +3 (nt = 0..7)
	auto load_resid = [&](f32x4 (&rv)[8], int tile) {
		const int m = tile * SR_ROWS + wm * 16 + fr;
		const float* R = (const float*)g.ep.resid + (size_t)(m < g.M ? m : g.M - 1) * g.ep.ldr + wn * 128 + fq * 4;
#pragma unroll
		for (int nt = 0; nt < 8; ++nt) rv[nt] = *reinterpret_cast<const f32x4*>(R + nt * 16);
	};
	const DropoutDesc d = {g.ep.drop_p, g.ep.seed_lo, g.ep.seed_hi, g.ep.drop_site};
	const bool drop = g.ep.drop_p > 0.f;

	stage(t, 0);
	f32x4 rcur[8], rnext[8];
	load_resid(rcur, t);
	asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // W and the first A tile have landed (the 8 residual loads may still fly)
	__builtin_amdgcn_s_barrier();
	asm volatile("" ::: "memory");
	int buf = 0;
	for (; t < ntiles; t += gridDim.x, buf ^= 1) {
		const int tn = t + gridDim.x;
		const bool has_next = tn < ntiles;
		if (has_next) {
			stage(tn, buf ^ 1);  // that buffer was last read before the barrier that closed the previous iteration
			load_resid(rnext, tn);
		}

		f32x4 acc[8];
		{
			const char* rowp = atile + buf * SR_TILE + (wm * 16 + fr) * SR_ROWB;
			bf16x8 af[SR_NKS];
#pragma unroll
			for (int ks = 0; ks < SR_NKS; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(rowp + (((ks * 4 + fq) ^ fr) << 4));
#pragma unroll
			for (int nt = 0; nt < 8; ++nt) {
				const char* wrow = smem + (wn * 128 + nt * 16 + fr) * SR_ROWB;
				acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
				for (int ks = 0; ks < SR_NKS; ++ks)
					acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(wrow + (((ks * 4 + fq) ^ fr) << 4)), af[ks], acc[nt], 0, 0, 0);
			}
		}

		const int m = t * SR_ROWS + wm * 16 + fr;
		if (m < g.M) {
			float* C = (float*)g.ep.c + (size_t)m * g.ep.ldc + wn * 128 + fq * 4;
			const float* B = (const float*)g.ep.bias;
#pragma unroll
			for (int nt = 0; nt < 8; ++nt) {
				float sc[4] = {1.f, 1.f, 1.f, 1.f};
				if (drop) dropout_scale4(d, (uint64_t)m * SR_N + wn * 128 + nt * 16 + fq * 4, sc);
				const f32x4 bb = B ? *reinterpret_cast<const f32x4*>(B + wn * 128 + nt * 16 + fq * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
				float v[4];
#pragma unroll
				for (int r = 0; r < 4; ++r) v[r] = rcur[nt][r] + bf16_round(acc[nt][r] + bb[r]) * sc[r];  // as epilogue4<RESID_F32>
				st_f32x4(C + nt * 16, v, true, 4);
			}
		}
		if (has_next) {
			// The next tile's A rows were requested before this iteration's 8 residual loads and 8 stores (an edge tile that skips stores is the last
			// tile and has no next; bias loads, if the compiler keeps them in the loop, only make this stricter): they need not wait for the stores.
			asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#pragma unroll
			for (int nt = 0; nt < 8; ++nt) rcur[nt] = rnext[nt];
		}
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();  // A(next) visible to every wave; every wave is done reading A(this)
		asm volatile("" ::: "memory");
	}
}

}  // namespace

static std::atomic<int> g_skinny_wide{0};
// Diagnostic: 0 = the [M x 512 x 512] bf16-store GEMM as four 128-column blocks (two row halves x four column quarters per workgroup; default), 1 = two 256-column
// blocks; < 0 queries.  Returns the previous setting.  Results are bit-identical either way -- and so is the time (52.6 vs 53.1 us at 61.5 k rows: the launch is
// not bound by how many CUs pull the same A tile).
extern "C" int novic_skinny_wide_policy(int wide) {
	const int prev = g_skinny_wide;
	if (wide >= 0) g_skinny_wide = wide;
	return prev;
}

// Called by novic_gemm_bf16 (gemm.hip) for K-contiguous x K-contiguous problems; returns 1 if the problem is not one this kernel takes.
int novic_gemm_skinny_try(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const novic_epilogue_t* ep, hipStream_t stream) {
	if (N == SR_N && K == SR_K && M >= 4096 && ep->kind == NOVIC_EPI_RESID_F32) {
		const uint64_t ab = (uint64_t)M * lda * 2, wb = (uint64_t)N * ldb * 2;
		if (ab >= 0xFFFFFFF0ull || wb >= 0xFFFFFFF0ull) return 1;
		if ((ep->ldc & 3) || (ep->ldr & 3) || ((uintptr_t)ep->c & 15) || ((uintptr_t)ep->resid & 15) || (ep->bias && ((uintptr_t)ep->bias & 15))) return 1;
		SkinnyArgs g;
		g.A = (const bf16*)A; g.W = (const bf16*)B;
		g.M = M; g.lda = lda; g.ldw = ldb;
		g.a_bytes = (unsigned)ab; g.w_bytes = (unsigned)wb;
		g.ep = *ep;
		const int ntiles = (M + SR_ROWS - 1) / SR_ROWS;
		static std::atomic<bool> attr_done{false};
		if (!attr_done) {
			(void)hipFuncSetAttribute((const void*)skinny_k128_resid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SR_WBYTES + 2 * SR_TILE);
			attr_done = true;
		}
		hipLaunchKernelGGL(skinny_k128_resid_kernel, dim3(ntiles < 256 ? ntiles : 256), dim3(512), SR_WBYTES + 2 * SR_TILE, stream, g);
		return 0;
	}
	if (N == 4 * SK_N && K == SK_K && M >= 4096 && ep->kind == NOVIC_EPI_RESID_F32) {
		// [M x 512 x 512] + fp32 residual (the decoder's out-proj): four 128-column blocks of the resident-weight streaming kernel -- 315 MB of HBM
		// traffic (55 us) that the 128^2 kernel runs in 72-96 us
		const uint64_t ab = (uint64_t)M * lda * 2, wb = (uint64_t)N * ldb * 2;
		if (ab >= 0xFFFFFFF0ull || wb >= 0xFFFFFFF0ull) return 1;
		if ((ep->ldc & 3) || (ep->ldr & 3) || ((uintptr_t)ep->c & 15) || ((uintptr_t)ep->resid & 15) || (ep->bias && ((uintptr_t)ep->bias & 15))) return 1;
		SkinnyArgs g;
		g.A = (const bf16*)A; g.W = (const bf16*)B;
		g.M = M; g.lda = lda; g.ldw = ldb;
		g.a_bytes = (unsigned)ab; g.w_bytes = (unsigned)wb;
		g.n_blocks = 4; g.n_total = N;
		g.ep = *ep;
		launch_skinny<NOVIC_EPI_RESID_F32>(g, 256, stream);
		return 0;
	}
	// (round 3: with the 8-phase K loop the 256 x 256 tile runs this shape in 36-38 us where the four column blocks take 49-52 -- tools/outproj_ab.py -- so the
	// streaming form is only taken when that schedule is switched off)
	if (N == 4 * SK_N && K == SK_K && M >= 4096 && ep->kind == NOVIC_EPI_STORE_BF16 && ep->act == NOVIC_ACT_NONE && novic_gemm256_pipeline(-1) == 0) {
		// [M x 512 x 512] with the bf16 store (the out-proj input gradient against the transposed weight shadow): 126 MB of traffic, 55-64 us on the
		// 256-wide tile kernel -- the same four column blocks
		const uint64_t ab = (uint64_t)M * lda * 2, wb = (uint64_t)N * ldb * 2;
		if (ab >= 0xFFFFFFF0ull || wb >= 0xFFFFFFF0ull) return 1;
		if ((ep->ldc & 7) || ((uintptr_t)ep->c & 15) || (ep->bias && ((uintptr_t)ep->bias & 15))) return 1;
		SkinnyArgs g;
		g.A = (const bf16*)A; g.W = (const bf16*)B;
		g.M = M; g.lda = lda; g.ldw = ldb;
		g.a_bytes = (unsigned)ab; g.w_bytes = (unsigned)wb;
		g.n_blocks = g_skinny_wide ? 2 : 4; g.n_total = N;
		g.ep = *ep;
		if (g_skinny_wide) launch_skinny<NOVIC_EPI_STORE_BF16, 4>(g, 256, stream);  // two 256-column blocks per row stream
		else launch_skinny<NOVIC_EPI_STORE_BF16>(g, 256, stream);
		return 0;
	}
	if (N != SK_N || K != SK_K || M < 4096) return 1;
	if (ep->kind != NOVIC_EPI_STORE_BF16 && ep->kind != NOVIC_EPI_GELU_BF16 && ep->kind != NOVIC_EPI_GELU_BWD_BF16) return 1;
	if (epilogue_is_act_variant(ep)) return 1;  // (relu / tanh, a bias in front of the activation: the 128 x 128 kernel)
	if (ep->kind == NOVIC_EPI_STORE_BF16 && ep->act != NOVIC_ACT_NONE) return 1;
	// 16-byte epilogue accesses
	if ((ep->ldc & 7) || ((uintptr_t)ep->c & 15) || (ep->c2 && ((uintptr_t)ep->c2 & 15)) || (ep->bias && ((uintptr_t)ep->bias & 15))) return 1;
	if (ep->kind == NOVIC_EPI_GELU_BWD_BF16 && ((ep->ldr & 7) || ((uintptr_t)ep->resid & 15))) return 1;
	const uint64_t ab = (uint64_t)M * lda * 2, wb = (uint64_t)N * ldb * 2;
	if (ab >= 0xFFFFFFF0ull || wb >= 0xFFFFFFF0ull) return 1;
	SkinnyArgs g;
	g.A = (const bf16*)A; g.W = (const bf16*)B;
	g.M = M; g.lda = lda; g.ldw = ldb;
	g.a_bytes = (unsigned)ab; g.w_bytes = (unsigned)wb;
	g.n_blocks = 1; g.n_total = SK_N;
	g.ep = *ep;
	const int ntiles = (M + SK_ROWS - 1) / SK_ROWS;
	const int grid = ntiles < 256 ? ntiles : 256;
	switch (ep->kind) {
		case NOVIC_EPI_STORE_BF16: launch_skinny<NOVIC_EPI_STORE_BF16>(g, grid, stream); break;
		case NOVIC_EPI_GELU_BF16: launch_skinny<NOVIC_EPI_GELU_BF16>(g, grid, stream); break;
		default: launch_skinny<NOVIC_EPI_GELU_BWD_BF16>(g, grid, stream); break;
	}
	return 0;
}
