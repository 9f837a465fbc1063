// bf16 MFMA GEMM for gfx950 with fused epilogues.
//
//   C[m][n] = sum_k A(m,k) * B(k,n)      fp32 accumulate on v_mfma_f32_16x16x32_bf16
//
// Operand storage (both operands independently):
//   K-contiguous ("KC"):  X[row][k]  -- forward linears: A = activations [M][K], B = weight [N][K]
//   K-strided    ("KS"):  X[k][col]  -- backward: dX = dY * W (B = W[k=N_out][n]) and dW = dY^T * X (A = dY[k=rows][m], B = X[k=rows][n])
// KS operands are staged row-major into LDS exactly as they lie in HBM (coalesced along the contiguous dim)
// and turned into MFMA fragments by ds_read_b64_tr_b16 (the CDNA4 transposing LDS read), so no
// transposed copy of any activation or weight is ever materialised.
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA tiles; double-buffered LDS (64 KiB),
// global->register->LDS staging with TWO register sets so the loads of tiles k+1 and k+2 are in flight while tile k is multiplied
// (one barrier per K-tile).  The MFMA is issued "swapped" (first operand = B rows) so a lane holds 4 consecutive n of one output row;
// the accumulators then go through LDS once so that the epilogue touches global memory in whole 128/256-byte row segments.
// LDS swizzles: KC tiles (128-B rows) XOR the 16-B chunk with row&7 -> ds_read_b128 conflict-free;
//               KS tiles (256-B rows) XOR the 8-B granule with ((k&3)<<2 | (k>>3&1)<<4) -> tr reads conflict-free.
// Workgroup order is XCD-aware: the blocks an XCD receives (blockIdx % 8 round-robin) cover a contiguous range of a chunk-major tile order
// (chunks of column tiles whose B panel fits the XCD's 4 MiB L2), so B is fetched from beyond L2 once per chunk and A once per chunk.
#include <cstdlib>

#include "gemm_epilogue.hpp"

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NT = 256;
constexpr int TILE_BYTES = 128 * 64 * 2;  // one operand tile (either storage)

struct GemmArgs {
	const bf16* A;
	const bf16* B;
	int M, N, K;
	int lda, ldb;
	unsigned a_bytes, b_bytes, r_bytes;  // extents of the operand / residual buffers for the SRD range check
	int k_chunk;  // K range per split (multiple of BK)
	int tiles_m, tiles_n;
	int group_n;  // column tiles per L2-resident B chunk (tile order inside an XCD: chunk-major, then row panel, then column)
	int splits;   // > 1: 1-D grid of tiles * splits workgroups, K ranges dealt out per XCD (see the kernel)
	unsigned long long* trace;  // diagnostic: [workgroup][4] wall-clock stamps (start, first K-tile in LDS, K loop done, epilogue done); null = off
	novic_epilogue_t ep;
};

// Operand tiles are fetched with SRD buffer loads (buffer_load_dwordx4 ... offen): the hardware range check returns zeros for an offset past
// the buffer, so an out-of-range chunk (row >= rows or k >= K) only needs its OFFSET forced out of range before the load.  Two ways not to
// do it: a predicated `ok ? *p : 0` makes hipcc split the access into four branch-wrapped dword loads; a select on the loaded VALUE pins the
// s_waitcnt right behind the load and exposes the full memory latency every K-tile (cdna_hip_programming.md T8, "three .s-level traps" (c)).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0xFFFFFFF0u;

// ---- staging: each thread moves 4 x 16 B per operand per K-tile ----
template <bool KS>
__device__ __forceinline__ void stage_load(u32x4 (&r)[4], __amdgpu_buffer_rsrc_t rs, int ld, int row0, int nrows, int k0, int kend, int tid) {
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const int id = tid + NT * i;
		unsigned off;
		if (!KS) {  // tile [128 rows][64 k]: 8 chunks per row
			const int rr = id >> 3, c = id & 7;
			const int row = row0 + rr, k = k0 + c * 8;
			off = (row < nrows && k < kend) ? ((unsigned)row * (unsigned)ld + (unsigned)k) * 2u : OOB;
		} else {  // tile [64 k][128 cols]: 16 chunks per k-row
			const int kk = id >> 4, c = id & 15;
			const int k = k0 + kk, col = row0 + c * 8;
			off = (k < kend && col < nrows) ? ((unsigned)k * (unsigned)ld + (unsigned)col) * 2u : OOB;
		}
		r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
	}
}
template <bool KS>
__device__ __forceinline__ void stage_store(const u32x4 (&r)[4], char* lds, int tid) {
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const int id = tid + NT * i;
		int off;
		if (!KS) {
			const int rr = id >> 3, c = id & 7;
			off = rr * 128 + ((c ^ (rr & 7)) << 4);
		} else {
			const int kk = id >> 4, c = id & 15;
			const int x = ((kk & 3) << 2) | (((kk >> 3) & 1) << 4);
			off = kk * 256 + (((2 * c) ^ x) << 3);
		}
		*reinterpret_cast<u32x4*>(lds + off) = r[i];
	}
}

// ---- fragment reads: lane l gets X[row = base + (l&15)][k = ks*32 + 8*(l>>4) + 0..7] ----
template <bool KS>
__device__ __forceinline__ bf16x8 frag_read(const char* lds, int base, int ks, int lane) {
	if (!KS) {
		const int row = base + (lane & 15);
		const int chunk = ks * 4 + (lane >> 4);
		return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((chunk ^ (row & 7)) << 4));
	} else {
		const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
		const int gran = (base >> 2) + p;
		const int ka = ks * 32 + 8 * g + q, kb = ka + 4;
		const int xa = ((ka & 3) << 2) | (((ka >> 3) & 1) << 4);
		const int xb = ((kb & 3) << 2) | (((kb >> 3) & 1) << 4);
		typedef bf16x4 __attribute__((address_space(3))) * lds4_t;
		bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(lds + ka * 256 + ((gran ^ xa) << 3)));
		bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(lds + kb * 256 + ((gran ^ xb) << 3)));
		return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
	}
}

// RESID_F32 with the residual already in registers (full 4-column case)
template <int DROP>
__device__ __forceinline__ void epilogue_resid_pre(const novic_epilogue_t& ep, int m, int n, int N, float (&v)[4], const float (&rv)[4]) {
	float s[4] = {1.f, 1.f, 1.f, 1.f};
	if (DROP != 0) {
		DropoutDesc d = {ep.drop_p, ep.seed_lo, ep.seed_hi, ep.drop_site};
		dropout_scale4(d, (uint64_t)m * N + n, s);
	}
	if (ep.bias) {
		const f32x4 b = *reinterpret_cast<const f32x4*>((const float*)ep.bias + n);
#pragma unroll
		for (int r = 0; r < 4; ++r) v[r] += b[r];
	}
#pragma unroll
	for (int r = 0; r < 4; ++r) v[r] = rv[r] + bf16_round(v[r]) * s[r];
	st_f32x4((float*)ep.c + (size_t)m * ep.ldc + n, v, (ep.ldc & 3) == 0, 4);
}

template <bool A_KS, bool B_KS>
__device__ __forceinline__ void compute_tile(const char* la, const char* lb, int wm, int wn, int lane, f32x4 (&acc)[4][4]) {
	// all 16 fragment reads of the K-tile are issued up front (the second k-step's reads fly while the first k-step's MFMAs run)
	bf16x8 fa[2][4], fb[2][4];
#pragma unroll
	for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
		for (int i = 0; i < 4; ++i) fa[ks][i] = frag_read<A_KS>(la, wm * 64 + i * 16, ks, lane);
#pragma unroll
		for (int j = 0; j < 4; ++j) fb[ks][j] = frag_read<B_KS>(lb, wn * 64 + j * 16, ks, lane);
	}
#pragma unroll
	for (int ks = 0; ks < 2; ++ks)
#pragma unroll
		for (int i = 0; i < 4; ++i)
#pragma unroll
			for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);
}

template <bool A_KS, bool B_KS, int EPI>
__global__ __launch_bounds__(NT, 2) void gemm_kernel(const GemmArgs g) {
	extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][A tile | B tile]; reused by the epilogue
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int wm = wave >> 1, wn = wave & 1;

	// ep.row_limit (device int): only the first *row_limit token rows take part.  Token rows are M for the row-major-A forms -- the tile grid shrinks to
	// the row tiles that exist BEFORE the workgroups are dealt out over the XCDs (with the full grid the surviving tiles would all sit on the first XCDs)
	// -- and K for the weight-gradient form, whose K ranges are then dealt out over the clamped K so that the splits stay balanced.
	int Mlim = g.M, Klim = g.K, kchunk = g.k_chunk, tiles_m = g.tiles_m;
	if (g.ep.row_limit) {
		const int lim = max(*g.ep.row_limit, 0);
		if (A_KS && B_KS) {
			Klim = min(g.K, lim);
			kchunk = max(((Klim + BK - 1) / BK + g.splits - 1) / g.splits, 1) * BK;
		} else {
			Mlim = min(g.M, lim);
			tiles_m = (Mlim + BM - 1) / BM;
		}
	}
	// XCD-aware bijective remap of the linear block id (blocks b and b+8 share an XCD).
	const int nwg = tiles_m * g.tiles_n;
	const int bid = blockIdx.x;
	if (g.splits <= 1 && bid >= nwg) return;
	const int xcd = bid & 7, q = nwg >> 3, rm = nwg & 7;
	int lid, ksplit;
	if (g.splits > 1) {
		// split-K (weight gradients: K = all tokens, few output tiles): XCD x owns the K ranges x, x+8, ... and runs ALL output tiles of a range
		// together, so each K slice of A and B is fetched from HBM by exactly one XCD and shared by its workgroups through L2 -- dealing the
		// ranges over blockIdx.z instead let every XCD read (nearly) every slice (measured 3.6 GB fetched for 0.85 GB of operands).
		const int local = bid >> 3;
		ksplit = xcd + 8 * (local / nwg);
		lid = local % nwg;
	} else {
		ksplit = 0;
		lid = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + (bid >> 3);
	}
	// chunk-major tile order: a chunk of group_n column tiles keeps its B panel (group_n x 128 x K bf16 <= ~2 MiB) in the XCD's L2 while the
	// row panels stream past it, and each A panel is fetched once per chunk instead of once per column tile.
	const int per_chunk = tiles_m * g.group_n;
	const int full = (g.tiles_n / g.group_n) * per_chunk;
	int tm, tn;
	if (lid < full) {
		const int c = lid / per_chunk, rem = lid - c * per_chunk;
		tm = rem / g.group_n;
		tn = c * g.group_n + (rem - tm * g.group_n);
	} else {
		const int wt = g.tiles_n % g.group_n, rem = lid - full;
		tm = rem / wt;
		tn = (g.tiles_n / g.group_n) * g.group_n + (rem - tm * wt);
	}
	const int m0 = tm * BM, n0 = tn * BN;
	auto stamp = [&](int ev) {
		if (g.trace && tid == 0 && bid < 16384) g.trace[(size_t)bid * 4 + ev] = wall_clock64();
	};
	stamp(0);
	if (m0 >= Mlim) return;
	const int kbeg = ksplit * kchunk;
	if (kbeg >= Klim && (g.splits > 1 || Klim == 0)) return;  // empty K range (split count rounded up to a multiple of 8; nothing valid at all)
	const int kend = min(Klim, kbeg + kchunk);
	const int nk = (kend - kbeg + BK - 1) / BK;

	f32x4 acc[4][4];
#pragma unroll
	for (int i = 0; i < 4; ++i)
#pragma unroll
		for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

	// Two register sets keep the loads of tiles k+1 and k+2 in flight while tile k is multiplied (prefetch distance 2),
	// two LDS buffers; one barrier per K-tile.
	// The residual epilogue's fp32 input (the tile's 64 KiB of the residual stream) is requested before the K loop so its HBM latency hides
	// behind the MFMAs; lane mapping = the epilogue's (row = 4p + lane/16, 4 columns at 4*(lane%16)).
	f32x4 rpre[EPI == NOVIC_EPI_RESID_F32 ? 16 : 1];
	if (EPI == NOVIC_EPI_RESID_F32) {
		const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.ep.resid), 0, g.r_bytes, 0x00020000);
		const int n = n0 + wn * 64 + 4 * (lane & 15);
#pragma unroll
		for (int p = 0; p < 16; ++p) {
			const int m = m0 + wm * 64 + p * 4 + (lane >> 4);
			const unsigned off = (m < Mlim && n + 3 < g.N) ? ((unsigned)m * (unsigned)g.ep.ldr + (unsigned)n) * 4u : OOB;
			const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(sr, off, 0, 0);
			rpre[p] = (f32x4){__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3])};
		}
	}

	// Branch-free software pipeline: the loop body has NO conditionals (tiles past the end are fetched with out-of-range offsets = zeros, and an
	// odd tile count is rounded up), because the s_waitcnt pass merges scoreboards conservatively at control-flow joins and would otherwise wait
	// for the loads it has just issued (observed: vmcnt(7..0) instead of vmcnt(15..8) in front of the LDS writes).
	u32x4 ra0[4], rb0[4], ra1[4], rb1[4];
	const __amdgpu_buffer_rsrc_t sa = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.A), 0, g.a_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t sb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.B), 0, g.b_bytes, 0x00020000);
	char* buf0 = smem;
	char* buf1 = smem + 2 * TILE_BYTES;
	stage_load<A_KS>(ra0, sa, g.lda, m0, Mlim, kbeg, kend, tid);
	stage_load<B_KS>(rb0, sb, g.ldb, n0, g.N, kbeg, kend, tid);
	stage_load<A_KS>(ra1, sa, g.lda, m0, Mlim, kbeg + BK, kend, tid);
	stage_load<B_KS>(rb1, sb, g.ldb, n0, g.N, kbeg + BK, kend, tid);
	stage_store<A_KS>(ra0, buf0, tid);
	stage_store<B_KS>(rb0, buf0 + TILE_BYTES, tid);
	__syncthreads();
	stamp(1);

	const int nk2 = (nk + 1) & ~1;
	for (int kt = 0; kt < nk2; kt += 2) {
		stage_load<A_KS>(ra0, sa, g.lda, m0, Mlim, kbeg + (kt + 2) * BK, kend, tid);
		stage_load<B_KS>(rb0, sb, g.ldb, n0, g.N, kbeg + (kt + 2) * BK, kend, tid);
		compute_tile<A_KS, B_KS>(buf0, buf0 + TILE_BYTES, wm, wn, lane, acc);
		stage_store<A_KS>(ra1, buf1, tid);
		stage_store<B_KS>(rb1, buf1 + TILE_BYTES, tid);
		__syncthreads();
		stage_load<A_KS>(ra1, sa, g.lda, m0, Mlim, kbeg + (kt + 3) * BK, kend, tid);
		stage_load<B_KS>(rb1, sb, g.ldb, n0, g.N, kbeg + (kt + 3) * BK, kend, tid);
		compute_tile<A_KS, B_KS>(buf1, buf1 + TILE_BYTES, wm, wn, lane, acc);
		stage_store<A_KS>(ra0, buf0, tid);
		stage_store<B_KS>(rb0, buf0 + TILE_BYTES, tid);
		__syncthreads();
	}

	stamp(2);
	// ---- epilogue through LDS: each wave parks its 64x64 fp32 sub-tile in its own 16 KiB (16-byte chunks XOR-swizzled by row) and reads it
	// back row-wise, so 16 consecutive lanes own one 64-column row segment: 128-B (bf16) / 256-B (fp32) contiguous global accesses per row
	// instead of 32-B pieces, and for the atomic epilogue 64 lanes add into 256 contiguous bytes (the full-rate shape).
	char* wl = smem + wave * (64 * 256);
#pragma unroll
	for (int i = 0; i < 4; ++i)
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const int rr = i * 16 + (lane & 15), ch = j * 4 + (lane >> 4);
			*reinterpret_cast<f32x4*>(wl + rr * 256 + ((ch ^ (rr & 15)) << 4)) = acc[i][j];
		}
	__syncthreads();
	const int mw = m0 + wm * 64, nw = n0 + wn * 64;
	if (EPI == NOVIC_EPI_ATOMIC_F32) {
		float* C = (float*)g.ep.c;
		const int n = nw + lane;
#pragma unroll 4
		for (int rr = 0; rr < 64; ++rr) {
			const float v = *reinterpret_cast<const float*>(wl + rr * 256 + (((lane >> 2) ^ (rr & 15)) << 4) + ((lane & 3) << 2));
			const int m = mw + rr;
			if (m < Mlim && n < g.N) atomicAdd(C + (size_t)m * g.ep.ldc + n, v * g.ep.alpha);
		}
	} else {
		// (relu / tanh / identity in place of the erf GELU: forward row-major only, backward with a row-major A -- the forms the callers have; novic_gemm_bf16 refuses the others)
		epilogue_dispatch<EPI, (EPI == NOVIC_EPI_GELU_BF16 ? (!A_KS && !B_KS) : !A_KS)>(g.ep, [&](auto act_c, auto drop_c) {
			constexpr int ACT = decltype(act_c)::value, DROP = decltype(drop_c)::value;
#pragma unroll
			for (int p = 0; p < 16; ++p) {
				const int rr = p * 4 + (lane >> 4), ch = lane & 15;
				const f32x4 t = *reinterpret_cast<const f32x4*>(wl + rr * 256 + ((ch ^ (rr & 15)) << 4));
				const int m = mw + rr, n = nw + 4 * ch;
				if (m < Mlim && n < g.N) {
					float v[4] = {t[0], t[1], t[2], t[3]};
					if (EPI == NOVIC_EPI_RESID_F32 && n + 3 < g.N) {
						const f32x4 rr4 = rpre[p];
						float rv[4] = {rr4[0], rr4[1], rr4[2], rr4[3]};
						epilogue_resid_pre<DROP>(g.ep, m, n, g.N, v, rv);
					} else {
						epilogue4<EPI, ACT, DROP>(g.ep, m, n, g.N, v);
					}
				}
			}
		});
	}
	stamp(3);
}

template <bool A_KS, bool B_KS>
int launch_epi(const GemmArgs& g, int splits, hipStream_t stream) {
	dim3 grid(g.tiles_m * g.tiles_n * splits, 1, 1), block(NT);
	const size_t shm = 4 * TILE_BYTES;
#define NOVIC_GEMM_CASE(E)                                                                  \
	case E: {                                                                               \
		static std::atomic<bool> attr_done{false};                                                      \
		if (!attr_done) {                                                                   \
			(void)hipFuncSetAttribute((const void*)gemm_kernel<A_KS, B_KS, E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
			attr_done = true;                                                               \
		}                                                                                   \
		hipLaunchKernelGGL((gemm_kernel<A_KS, B_KS, E>), grid, block, shm, stream, g);      \
		break;                                                                              \
	}
	switch (g.ep.kind) {
		NOVIC_GEMM_CASE(NOVIC_EPI_STORE_BF16)
		NOVIC_GEMM_CASE(NOVIC_EPI_STORE_F32)
		NOVIC_GEMM_CASE(NOVIC_EPI_ATOMIC_F32)
		NOVIC_GEMM_CASE(NOVIC_EPI_RESID_F32)
		NOVIC_GEMM_CASE(NOVIC_EPI_GELU_BF16)
		NOVIC_GEMM_CASE(NOVIC_EPI_GELU_BWD_BF16)
		case NOVIC_EPI_RESID_F16:  // (row-major operands only: novic_gemm_bf16 checks; one instantiation instead of three)
			if constexpr (!A_KS && !B_KS) {
				static std::atomic<bool> attr_done{false};
				if (!attr_done) {
					(void)hipFuncSetAttribute((const void*)gemm_kernel<false, false, NOVIC_EPI_RESID_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
					attr_done = true;
				}
				hipLaunchKernelGGL((gemm_kernel<false, false, NOVIC_EPI_RESID_F16>), grid, block, shm, stream, g);
				break;
			}
			novic_set_error("novic_gemm_bf16: RESID_F16 takes row-major operands");
			return -22;
		default:
			novic_set_error("novic_gemm_bf16: unknown epilogue kind");
			return -22;
	}
#undef NOVIC_GEMM_CASE
	NOVIC_LAUNCH_CHECK();
	return 0;
}

}  // namespace

int novic_gemm256_try(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const novic_epilogue_t* ep, int force, int* tile_n,
                      hipStream_t stream);  // gemm256.hip

int novic_gemm_skinny_try(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const novic_epilogue_t* ep, hipStream_t stream);  // skinny.hip

static std::atomic<unsigned long long*> g_trace128{nullptr};
extern "C" int novic_gemm128_trace(unsigned long long* buf) {  // diagnostic, see include/novic_hip.h
	g_trace128 = buf;
	return 0;
}

static std::atomic<int> g_tile_policy{[] { const char* e = getenv("NOVIC_GEMM256"); return (e && e[0] == '0') ? 0 : 1; }()};

static thread_local int g_last_tile = 0;  // per host thread: the tile of THIS thread's last novic_gemm_bf16 call
extern "C" int novic_gemm_last_tile(void) { return g_last_tile; }

// launches per kernel since the last reset: [0] 128^2, [1] streaming 128-column kernel, [2] 256 x 256, [3] 256 x 192, [4] of those with a host-planned K-split
// tail, [5] with a device-planned one, [6] unused since round 5 (was: 128-row tiles).  Diagnostic (tests assert that a model-level check really ran through the persistent tiles); relaxed atomic counters, process-wide.
static std::atomic<unsigned long long> g_tile_counts[7];  // (zero-initialised: static storage)
extern "C" int novic_gemm_tile_counts(unsigned long long* out7, int reset) {
	if (out7) for (int i = 0; i < 7; ++i) out7[i] = g_tile_counts[i];
	if (reset) for (int i = 0; i < 7; ++i) g_tile_counts[i] = 0;
	return 0;
}

extern "C" int novic_gemm_tile_policy(int policy) {
	const int prev = g_tile_policy;
	if (policy >= 0 && policy <= 3) g_tile_policy = policy;  // 0: 128^2 only, 1: choose, 2 / 3: force the 256- / 192-wide LDS-DMA tile (benchmarks)
	return prev;
}

extern "C" int novic_gemm_bf16(const void* A, const void* B, int M, int N, int K, int lda, int ldb, int a_kstrided, int b_kstrided, int split_k,
                               const novic_epilogue_t* ep, hipStream_t stream) {
	NOVIC_CHECK(A && B && ep, "novic_gemm_bf16: null pointer");
	NOVIC_CHECK(ep->struct_bytes == (uint32_t)sizeof(novic_epilogue_t),
	            "novic_gemm_bf16: novic_epilogue_t.struct_bytes does not match this library's layout (binding built against another NOVIC_ABI_VERSION?)");
	NOVIC_CHECK(ep->c, "novic_gemm_bf16: null output pointer");
	NOVIC_CHECK(M >= 0 && N >= 0 && K >= 0, "novic_gemm_bf16: negative dimension");
	if (M == 0 || N == 0) return 0;
	NOVIC_CHECK(lda % 8 == 0 && ldb % 8 == 0, "novic_gemm_bf16: leading dimensions must be multiples of 8 elements (16-byte rows)");
	NOVIC_CHECK(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "novic_gemm_bf16: operands must be 16-byte aligned");
	{  // the epilogues store 4 columns per access whenever ldc allows it
		const bool f32_out = ep->kind == NOVIC_EPI_STORE_F32 || ep->kind == NOVIC_EPI_RESID_F32;
		NOVIC_CHECK(ep->ldc % 4 != 0 || ((uintptr_t)ep->c & (f32_out ? 15 : 7)) == 0, "novic_gemm_bf16: output must be aligned to 4 elements when ldc is a multiple of 4");
		NOVIC_CHECK(ep->ldc % 4 != 0 || !ep->c2 || ((uintptr_t)ep->c2 & 7) == 0, "novic_gemm_bf16: second output must be 8-byte aligned when ldc is a multiple of 4");
	}
	NOVIC_CHECK(split_k >= 1, "novic_gemm_bf16: split_k must be >= 1");
	NOVIC_CHECK(!(ep->kind == NOVIC_EPI_GELU_BF16 || ep->kind == NOVIC_EPI_GELU_BWD_BF16) || ep->act == NOVIC_ACT_NONE || ep->act == NOVIC_ACT_GELU || ep->act == NOVIC_ACT_RELU ||
	            ep->act == NOVIC_ACT_TANH || (ep->act == NOVIC_ACT_IDENTITY && ep->kind == NOVIC_EPI_GELU_BF16), "novic_gemm_bf16: unknown activation for the GELU epilogues");
	NOVIC_CHECK(!epilogue_is_act_variant(ep) || ep->act == NOVIC_ACT_NONE || ep->act == NOVIC_ACT_GELU || (!a_kstrided && !(b_kstrided && ep->kind == NOVIC_EPI_GELU_BF16)),
	            "novic_gemm_bf16: the relu / tanh / identity epilogues take a row-major A (and a row-major B in the forward form)");
	NOVIC_CHECK(split_k == 1 || ep->kind == NOVIC_EPI_ATOMIC_F32, "novic_gemm_bf16: split_k > 1 needs the atomic epilogue");
	GemmArgs g;
	g.A = (const bf16*)A;
	g.B = (const bf16*)B;
	g.M = M; g.N = N; g.K = K;
	g.lda = lda; g.ldb = ldb;
	{
		const uint64_t ab = (uint64_t)(a_kstrided ? K : M) * lda * 2, bb = (uint64_t)(b_kstrided ? K : N) * ldb * 2;
		NOVIC_CHECK(ab < 0xFFFFFFF0ull && bb < 0xFFFFFFF0ull, "novic_gemm_bf16: operands must be smaller than 4 GiB (32-bit buffer offsets)");
		g.a_bytes = (unsigned)ab; g.b_bytes = (unsigned)bb;
		const uint64_t rb = (uint64_t)M * (ep->ldr > 0 ? ep->ldr : 0) * 4;
		NOVIC_CHECK(ep->kind != NOVIC_EPI_RESID_F32 || (ep->resid && rb < 0xFFFFFFF0ull && ep->ldr % 4 == 0 && ((uintptr_t)ep->resid & 15) == 0),
		            "novic_gemm_bf16: residual must be 16-byte aligned, ldr a multiple of 4, smaller than 4 GiB");
		NOVIC_CHECK(ep->kind != NOVIC_EPI_RESID_F16 || (ep->resid && ep->ldr % 4 == 0 && ep->ldc % 4 == 0 && ((uintptr_t)ep->resid & 7) == 0 && ((uintptr_t)ep->c & 7) == 0 &&
		                                                !a_kstrided && !b_kstrided && !ep->row_limit && ep->drop_p == 0.f && !ep->c2),
		            "novic_gemm_bf16: RESID_F16 takes row-major operands, an 8-byte aligned half residual / output with leading dimensions that are multiples of 4, no dropout, no row_limit, no c2");
		g.r_bytes = (unsigned)rb;
	}
	g.tiles_m = (M + BM - 1) / BM;
	g.tiles_n = (N + BN - 1) / BN;
	// Tile order inside an XCD.  Short K: a chunk of column tiles whose whole B panel (group_n * 128 * K bf16 <= 2 MiB) stays in the 4 MiB L2.
	// Long K (nothing stays resident): only the K-slices the ~64 concurrently running blocks are working on can be shared, so make that set
	// as square as the shape allows (up to 8 column tiles side by side) -- with group_n = 1 every column tile re-streams A from HBM/MALL
	// (measured 3.3 GB instead of 0.86 GB per launch on the [57344 x 512 x 6912] input-gradient GEMM).
	g.group_n = 8192 / (K > 0 ? K : 1);
	if (g.group_n < 8) g.group_n = 8;
	if (g.group_n > g.tiles_n) g.group_n = g.tiles_n;
	int ktiles = (K + BK - 1) / BK;
	if (ktiles < 1) ktiles = 1;
	if (split_k > 1) {  // the request is a hint: K ranges are dealt out per XCD, so their number is a multiple of 8 (empty ranges add nothing)
		split_k = (split_k + 7) / 8 * 8;
		while (split_k > 8 && ktiles / split_k < 2) split_k -= 8;
	}
	g.k_chunk = ((ktiles + split_k - 1) / split_k) * BK;
	g.splits = split_k;
	g.ep = *ep;
	g.trace = g_trace128;
	const int policy = g_tile_policy;  // (the process-wide switch, read once per call)
	g_last_tile = 128;
	if (a_kstrided) {
		++g_tile_counts[0];
		if (b_kstrided) return launch_epi<true, true>(g, split_k, stream);
		novic_set_error("novic_gemm_bf16: (A k-strided, B k-contiguous) is not used by this path");
		return -22;
	}
	if (b_kstrided) {
		++g_tile_counts[0];
		return launch_epi<false, true>(g, split_k, stream);
	}
	// (The decoder's out-projection + residual [rows x 512 x 512] at training size stays on the streaming four-column-block kernel: on the 256 x 256 tile it was 70.4 -> 62.9 us
	// back to back in isolation, bit-identical, and 6.641 -> 6.694 ms per optimizer step INSIDE the step -- there the attention output and the residual stream arrive cold and the
	// streaming kernel's residual prefetch wins.  Round 4's switch for it was removed in round 5.)
	// Mid-size out-projections with a HOST row count (the text tower: [19 712 x 512 x 512]) on the 256 x 256 tile as well: in the tower 94.9 k -> 96.8 k texts/s, bit-identical
	// (tools/text_proj_ab.py); isolated [16 384 .. 39 424 rows] 24.9-47.9 -> 23.3-44.4 us (tools/vit_b32_gemm_ab.py ROWS 512).  Not the training step's (device row count,
	// dropout, 61 k rows: the streaming kernel wins there, above) and not below 16 k rows (the 256-wide plan declines; the streaming kernel keeps them).
	const bool midproj256 = policy == 1 && split_k == 1 && ep->kind == NOVIC_EPI_RESID_F32 && N == 512 && K == 512 && M >= 16384 && M < 40960 && !ep->row_limit && ep->drop_p == 0.f;
	if (split_k == 1 && policy == 1 && !midproj256 && novic_gemm_skinny_try(A, B, M, N, K, lda, ldb, ep, stream) == 0) {  // tall, 128 columns wide, K = 512
		g_last_tile = 64;
		++g_tile_counts[1];
		NOVIC_LAUNCH_CHECK();
		return 0;
	}
	if (split_k == 1 && policy != 0) {  // large problems: 256^2-tile LDS-DMA kernel (bit-identical results)
		int tn = 0;
		const int r = novic_gemm256_try(A, B, M, N, K, lda, ldb, ep, (policy == 2 || midproj256) ? 256 : (policy == 3 ? 192 : 0), &tn, stream);
		if (r <= 0) {
			if (r == 0) {
				g_last_tile = tn & 0xFFF;
				++g_tile_counts[g_last_tile == 192 ? 3 : 2];
				if (tn & 0x1000) ++g_tile_counts[4];
				if (tn & 0x2000) ++g_tile_counts[5];
				NOVIC_LAUNCH_CHECK();
			}
			return r;
		}
	}
	if (midproj256 && novic_gemm_skinny_try(A, B, M, N, K, lda, ldb, ep, stream) == 0) {  // (the 256-wide kernels declined after all: the streaming kernel, as before)
		g_last_tile = 64;
		++g_tile_counts[1];
		NOVIC_LAUNCH_CHECK();
		return 0;
	}
	g_last_tile = 128;
	++g_tile_counts[0];
	return launch_epi<false, false>(g, split_k, stream);
}
