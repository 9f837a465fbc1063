// Weight-gradient GEMMs: dW[m][n] += alpha * sum_k dY[k][m] * X[k][n], k = token rows (tens of thousands), both operands stored row-major as the
// forward / backward passes left them ([rows][features]: "K-strided" for this product), fp32 accumulation.
// reference: autograd of every nn.Linear of the decoder (embedding_decoder.py:309-327 layers, :725 logits_linear, :1276 prefix MLP) -- grad_weight = grad_output^T @ input.
//
// wgrad256_kernel: 256 x 256 output tile per workgroup (8 waves = 2 x 4, each 128 x 64 = 8 x 4 MFMA tiles of v_mfma_f32_16x16x32_bf16), the token range
// cut into S parts so that tiles x S workgroups fill the chip in ONE round.  What is different from the 128^2 split-K kernel of gemm.hip that these
// problems ran on (in-proj dW [1536 x 512] over 61 k rows: 140 us = 0.28 of the MFMA peak; logits dW [6912 x 512] over 37 k rows: 350 us = 0.30):
//   * half the operand bytes per FLOP (64 KiB per 8.4 MFLOP K-tile instead of 2 x 32 KiB per 2 x 2.1): a CU pulls operands at a fixed ~45-50 GB/s,
//     and that, not the MFMA pipe, bounded the 128^2 kernel;
//   * LDS-DMA staging (buffer_load_dwordx4 ... lds, no VGPR round trip) of the operands AS THEY LIE in memory -- [k][columns] slabs, 512-byte rows;
//     the fragments come out of LDS through ds_read_b64_tr_b16 (the transposing read), 8-byte granules XOR-swizzled by the k row -- applied on the
//     SOURCE side of the DMA, chunk-wise (the DMA writes LDS lane-linearly) -- so the reads are bank-conflict free;
//   * no fp32 atomics: every workgroup parks its raw accumulators (1 KiB per store instruction) in caller-provided scratch and wgrad_reduce_kernel
//     adds the S parts of every element in a FIXED order into dW -- deterministic, and the 64 MiB of partials move at the HBM rate (~5 TB/s) where
//     memory-side atomics run at 1.3 TB/s.
//   * the token count may be clamped by a DEVICE int (packed rows / compacted loss block): the buffer descriptor is built in the kernel with the
//     clamped size, rows beyond it read as zeros, and the parts are re-dealt over the clamped count.
#include <type_traits>

#include "common.hpp"
#include "novic_hip.h"

namespace {

constexpr int WG_TN = 256, WG_TK = 64, WG_NT = 512;  // (tile rows: 32 NMF, template parameter of the kernel)
constexpr int WG_ROWB = 512;                  // bytes per k row of an operand slab in LDS (256 columns)
constexpr int WG_OP = WG_TK * WG_ROWB;        // 32 KiB per operand per K-tile
constexpr int WG_BUF = 2 * WG_OP;             // A slab | B slab
constexpr unsigned WG_OOB = 0x80000000u;      // a part's rows of an operand are < 2 GiB: this offset is past any descriptor -> the load returns zeros

struct WgradArgs {
	const bf16* A;   // dY [K][lda], columns = M
	const bf16* B;   // X  [K][ldb], columns = N
	float* C;        // dW [M][ldc] fp32, accumulated into ([N][ldc] when transpose_out)
	int M, N, K, lda, ldb, ldc;
	int tiles_m, tiles_n, splits;
	int transpose_out;     // element (m, n) of the product goes to C[n][m]: lets a [512 x 128] gradient run as its transpose, a [128 x 512] product
	float alpha;
	const int* row_limit;  // null, or device int: only the first *row_limit token rows exist
	float* ws;             // [tiles * splits][8 waves][4 NMF fragments][64 lanes][4] fp32
};

// FURTHER problems over the same token rows (nextra of them, tiles2 = their tiles together): their tiles follow the first problem's in the tile sequence, so ONE launch fills
// the chip with all of them (a layer's in-projection [1536 x 512] and out-projection [512 x 512] gradients: 12 + 4 tiles x 16 parts instead of 12 x 21 and 4 x 64 -- half
// the partial-sum traffic and one launch pair instead of two; round 6: the pairs of TWO layers, 32 tiles x 8 parts -- half of it again).  A kernel argument of its own,
// read-only: as part of WgradArgs, which the kernels copy and rewrite, the array sent the whole struct to scratch (272 bytes per lane: tools/audit_scratch.py).
struct WgradExtra {
	const bf16* A;
	const bf16* B;
	float* C;
	int M, N, lda, ldb, ldc, tiles_n, tiles, transpose_out;
};
struct WgradExtras {
	WgradExtra e[3];
	int nextra, tiles2;
};

// tile (global index over all problems) -> the problem's own operands (written into g) and its local tile index
__device__ __forceinline__ void take_problem(WgradArgs& g, const WgradExtra& e) {
	g.A = e.A; g.B = e.B; g.C = e.C;
	g.M = e.M; g.N = e.N; g.lda = e.lda; g.ldb = e.ldb; g.ldc = e.ldc; g.tiles_n = e.tiles_n; g.transpose_out = e.transpose_out;
}
__device__ __forceinline__ int select_problem(WgradArgs& g, const WgradExtras& ex, int tile) {
	int t = tile - g.tiles_m * g.tiles_n;
	if (t < 0 || ex.nextra <= 0) return tile;
	if (t < ex.e[0].tiles || ex.nextra == 1) { take_problem(g, ex.e[0]); return t; }
	t -= ex.e[0].tiles;
	if (t < ex.e[1].tiles || ex.nextra == 2) { take_problem(g, ex.e[1]); return t; }
	t -= ex.e[1].tiles;
	take_problem(g, ex.e[2]);
	return t;
}

typedef __attribute__((address_space(3))) void* wg_lds_ptr_t;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned wg_u32x2 __attribute__((ext_vector_type(2)));

// K-tile range of part s of `splits` over nkt K-tiles (the same arithmetic in the kernel and in the reduction)
__device__ __forceinline__ void part_range(int nkt, int splits, int s, int& kb, int& ke) {
	const int per = (nkt + splits - 1) / splits;
	kb = s * per;
	ke = min(nkt, kb + per);
}

// The transposing reads are issued as inline assembly: through the builtin, hipcc puts an `s_waitcnt vmcnt(0)` in front of the first read of every
// K-tile (an LDS read behind a pending LDS-DMA whose destination it cannot tell apart), which drains the NEXT tile's DMA before the current tile is
// multiplied -- 2.0 us per K-tile, DMA latency + MFMA time, instead of their maximum.  The asm reads are invisible to that pass; their own ordering is
// by hand: a batch of reads, `s_waitcnt lgkmcnt(0)`, a sched_barrier (the MFMAs must not be hoisted above the wait: cdna_hip_programming.md rule 18).
template <int OFF>
__device__ __forceinline__ void tr_read(wg_u32x2& dst, unsigned addr) {
	asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

// NMF = 16-row fragments of the output tile per wave along M: 8 -> 256 x 256 tile (the wave's sub-tile 128 x 64), 4 -> 128 x 256 (64 x 64: the
// feed-forward gradients, whose output is 128 wide in one dimension)
template <int NMF>
__global__ __launch_bounds__(WG_NT) void wgrad256_kernel(const WgradArgs gin, const WgradExtras ex) {
	WgradArgs g = gin;
	constexpr int TM = 32 * NMF;               // output rows per tile
	constexpr int RA = TM * 2;                 // bytes per k row of the A slab (512 / 256); the B slab has 512
	constexpr int RPI = 1024 / RA;             // k rows per DMA instruction of the A slab (2 / 4)
	constexpr int NAI = 8 / RPI;               // A DMA instructions per wave and K-tile (4 / 2)
	constexpr int A_BYTES = WG_TK * RA;
	extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A slab | B slab 32 KiB], 64 KiB apart
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const int wr = w >> 2, wc = w & 3;
	int Klim = g.K;
	if (g.row_limit) Klim = min(g.K, max(*g.row_limit, 0));
	const int nkt = (Klim + WG_TK - 1) / WG_TK;

	// item -> workgroup: the workgroups of one XCD (blockIdx % 8 round-robin) take a CONTIGUOUS run of the part-major item sequence, so the tiles
	// of a token range sit (mostly) on one XCD and share its operand slabs through L2
	const int per_xcd = gridDim.x >> 3;
	const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
	const int ntiles = g.tiles_m * g.tiles_n + ex.tiles2;
	if (item >= ntiles * g.splits) return;
	const int s = item / ntiles, tile = select_problem(g, ex, item - s * ntiles);
	const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
	int kb, ke;
	part_range(nkt, g.splits, s, kb, ke);
	if (kb >= ke) return;  // empty part (the reduction skips it by the same arithmetic)

	// descriptors based at THIS PART's first k row (64-bit pointer arithmetic: the operands themselves may exceed the 2 GiB a 32-bit buffer offset spans --
	// the multiset step's logits gradient is 2.4 GB -- a part's rows never do, checked by the host) and sized to the rows that exist from there: a k row at
	// or beyond Klim is out of range and reads as zeros
	const size_t k0 = (size_t)kb * WG_TK;
	const uint64_t left = (uint64_t)(Klim - (int)k0);
	const uint64_t ra_bytes = left * (uint64_t)g.lda * 2ull, rb_bytes = left * (uint64_t)g.ldb * 2ull;
	const __amdgpu_buffer_rsrc_t sa = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.A) + k0 * (size_t)g.lda, 0, (unsigned)(ra_bytes < 0x7FFFFFF0ull ? ra_bytes : 0x7FFFFFF0ull), 0x00020000);
	const __amdgpu_buffer_rsrc_t sb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.B) + k0 * (size_t)g.ldb, 0, (unsigned)(rb_bytes < 0x7FFFFFF0ull ? rb_bytes : 0x7FFFFFF0ull), 0x00020000);

	// staging: wave w fills k rows 8w .. 8w+7 of each slab, 1 KiB (two 512-byte rows / four 256-byte rows) per instruction; the lane that writes slot s
	// of row k fetches the 16-byte chunk s ^ cx(k) of that row.  cx = the granule swizzle of the transposing reads below, taken chunk-wise.
	auto cx = [](int kk) { return ((kk & 3) << 1) | (((kk >> 3) & 1) << 3); };
	unsigned va[4], vb[4];  // NAI used (a template-dependent array size captured by the lambdas below makes hipcc drop the kernel's host stub)
#pragma unroll
	for (int i = 0; i < NAI; ++i) {
		const int kk = w * 8 + RPI * i + lane / (64 / RPI);
		const int ca = tm * TM + (((lane % (64 / RPI)) ^ cx(kk)) * 8);
		va[i] = ca < g.M ? ((unsigned)kk * (unsigned)g.lda + (unsigned)ca) * 2u : WG_OOB;
	}
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const int kk = w * 8 + 2 * i + (lane >> 5);
		const int cb = tn * WG_TN + (((lane & 31) ^ cx(kk)) * 8);
		vb[i] = cb < g.N ? ((unsigned)kk * (unsigned)g.ldb + (unsigned)cb) * 2u : WG_OOB;
	}
	const unsigned ka_step = (unsigned)WG_TK * (unsigned)g.lda * 2u, kb_step = (unsigned)WG_TK * (unsigned)g.ldb * 2u;
	auto stage = [&](int buf, int kt) {
		char* base = smem + buf * WG_BUF;
		const unsigned oa = (unsigned)(kt - kb) * ka_step, ob = (unsigned)(kt - kb) * kb_step;  // relative to the part's first row (the descriptors' base)
#pragma unroll
		for (int i = 0; i < NAI; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(sa, (wg_lds_ptr_t)(base + (w * NAI + i) * 1024), 16, va[i] == WG_OOB ? WG_OOB : va[i] + oa, 0, 0, 0);
#pragma unroll
		for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(sb, (wg_lds_ptr_t)(base + A_BYTES + (w * 4 + i) * 1024), 16, vb[i] == WG_OOB ? WG_OOB : vb[i] + ob, 0, 0, 0);
	};

	// fragment reads (as gemm.hip frag_read<KS = true>): lane l gets X[column = base + (l & 15)][k = ks*32 + 8*(l >> 4) + 0..7] from two transposing
	// 8-byte reads (k rows ka = ks*32 + 8*(l>>4) + (l>>2 & 3) and ka + 4) of granule (base >> 2) + (l & 3), stored at granule ^ x(k).
	// x(k) = ((k & 3) << 2) | ((k >> 3 & 1) << 4) depends on the LANE only (k & 3 = l>>2 & 3, k>>3 & 1 = l>>4 & 1 for every ks and both halves), and it
	// touches granule bits 2..4 = the low three bits c_lo of the 16-column fragment index c = base / 16: so the address is
	//     [lane part + ((c_lo ^ y) << 5)] + immediate(ks, half, c_hi, slab),   y = x >> 2
	// -- one per-lane register per fragment and buffer, and NO address arithmetic in the K loop (computed per read, the XOR / shift / add pairs
	// competed with the MFMAs for issue slots).
	const int fg = lane >> 4, fq4 = (lane >> 2) & 3, fp = lane & 3;
	const int ylane = fq4 | ((fg & 1) << 2);
	unsigned pa[2][8], pb[2][4];  // [buffer][fragment]: LDS byte addresses (A: c = wr*NMF + i; B: c = wc*4 + j)
	const unsigned smem_base = (unsigned)(uintptr_t)(wg_lds_ptr_t)smem;  // LDS addresses are 32-bit offsets
#pragma unroll
	for (int bf = 0; bf < 2; ++bf) {
#pragma unroll
		for (int i = 0; i < NMF; ++i) {
			const int c = wr * NMF + i;
			pa[bf][i] = smem_base + bf * WG_BUF + (unsigned)((8 * fg + fq4) * RA + fp * 8 + (((c & 7) ^ ylane) << 5) + (c >> 3) * 256);
		}
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const int c = wc * 4 + j;
			pb[bf][j] = smem_base + bf * WG_BUF + A_BYTES + (unsigned)((8 * fg + fq4) * WG_ROWB + fp * 8 + (((c & 7) ^ ylane) << 5) + (c >> 3) * 256);
		}
	}

	f32x4 acc[NMF][4];
#pragma unroll
	for (int i = 0; i < NMF; ++i)
#pragma unroll
		for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

	constexpr int HM = NMF / 2;  // row fragments per half
	// a quarter step = one k-step (32 k) x one half of the wave's row fragments: 4 x HM MFMAs
	auto mul = [&](const wg_u32x2 (&bl)[4], const wg_u32x2 (&bh)[4], const wg_u32x2 (&al)[HM], const wg_u32x2 (&ah)[HM], int h) {
		bf16x8 fb[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) fb[j] = __builtin_bit_cast(bf16x8, (u32x4_t){bl[j][0], bl[j][1], bh[j][0], bh[j][1]});
#pragma unroll
		for (int i = 0; i < HM; ++i) {
			const bf16x8 fa = __builtin_bit_cast(bf16x8, (u32x4_t){al[i][0], al[i][1], ah[i][0], ah[i][1]});
#pragma unroll
			for (int j = 0; j < 4; ++j) acc[h * HM + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa, acc[h * HM + i][j], 0, 0, 0);
		}
	};
	// One K-tile as four quarter steps, the reads running one quarter step ahead of the MFMAs that consume them -- issued all at once per k-step and
	// waited for in full, both waves of a SIMD (released by the same barrier) sat out the LDS latency together twice per K-tile (1.6 us per K-tile
	// instead of 2.0; the 64 KiB of operands take ~1.45 us to arrive).  Two B register sets (one per k-step), two A sets (one per row half).
	auto compute = [&](const unsigned (&qa)[8], const unsigned (&qb)[4]) {
		wg_u32x2 bl[2][4], bh[2][4], al[2][HM], ah[2][HM];
#pragma unroll
		for (int j = 0; j < 4; ++j) { tr_read<0>(bl[0][j], qb[j]); tr_read<4 * WG_ROWB>(bh[0][j], qb[j]); }
#pragma unroll
		for (int i = 0; i < HM; ++i) { tr_read<0>(al[0][i], qa[i]); tr_read<4 * RA>(ah[0][i], qa[i]); }
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int i = 0; i < HM; ++i) { tr_read<0>(al[1][i], qa[HM + i]); tr_read<4 * RA>(ah[1][i], qa[HM + i]); }
		__builtin_amdgcn_sched_barrier(0);
		mul(bl[0], bh[0], al[0], ah[0], 0);
		__builtin_amdgcn_sched_barrier(0);
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int j = 0; j < 4; ++j) { tr_read<32 * WG_ROWB>(bl[1][j], qb[j]); tr_read<36 * WG_ROWB>(bh[1][j], qb[j]); }
#pragma unroll
		for (int i = 0; i < HM; ++i) { tr_read<32 * RA>(al[0][i], qa[i]); tr_read<36 * RA>(ah[0][i], qa[i]); }
		__builtin_amdgcn_sched_barrier(0);
		mul(bl[0], bh[0], al[1], ah[1], 1);
		__builtin_amdgcn_sched_barrier(0);
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int i = 0; i < HM; ++i) { tr_read<32 * RA>(al[1][i], qa[HM + i]); tr_read<36 * RA>(ah[1][i], qa[HM + i]); }
		__builtin_amdgcn_sched_barrier(0);
		mul(bl[1], bh[1], al[0], ah[0], 0);
		__builtin_amdgcn_sched_barrier(0);
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		__builtin_amdgcn_sched_barrier(0);
		mul(bl[1], bh[1], al[1], ah[1], 1);
		__builtin_amdgcn_sched_barrier(0);
	};

	// two LDS buffers, one barrier per K-tile: the DMA of K-tile k+1 flies while K-tile k is multiplied (vmcnt(0): nothing else is outstanding)
	stage(0, kb);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	for (int kt = kb; kt < ke; kt += 2) {  // two K-tiles per trip: the buffer each half reads is fixed at compile time
		if (kt + 1 < ke) stage(1, kt + 1);
		compute(pa[0], pb[0]);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		asm volatile("" ::: "memory");
		if (kt + 1 >= ke) break;
		if (kt + 2 < ke) stage(0, kt + 2);
		compute(pa[1], pb[1]);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		asm volatile("" ::: "memory");
	}

	// raw accumulators out, 1 KiB per instruction (acc[mt][j][r] = element (row wr*16*NMF + mt*16 + lane%16, column wc*64 + j*16 + 4*(lane/16) + r) of the tile)
	float* wp = g.ws + ((size_t)item * 8 + w) * (NMF * 4 * 256) + lane * 4;
#pragma unroll
	for (int mt = 0; mt < NMF; ++mt)
#pragma unroll
		for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(acc[mt][j], reinterpret_cast<f32x4*>(wp + (mt * 4 + j) * 256));
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// The same product on the 8-phase schedule of cdna_hip_programming.md section 5 ("256^2 8-phase template", T3 + T4 + T5), round 3.
//
// wgrad256_kernel above multiplies a K-tile behind ONE barrier with the next K-tile's LDS-DMA behind a vmcnt(0): both waves of a SIMD read LDS together, then share
// the matrix pipe together, and every K-tile ends by draining the vector-memory queue -- 1.5 us per K-tile for 0.98 us of MFMA.  Here:
//   * a K-tile is four PHASES (quarter steps: one k-step of 32 x one half of the wave's row fragments = 4 x HM MFMAs), each phase a LOAD segment (this quarter's
//     transposing reads + one LDS-DMA half-tile of a LATER K-tile) and a COMPUTE segment (lgkmcnt(0), the MFMAs under s_setprio 1), a raw s_barrier after each;
//   * waves 4-7 -- the SIMD partners of waves 0-3 (MI355X_MICROARCH.md: a workgroup's waves go to the SIMDs cyclically) -- run ONE BARRIER BEHIND: while one wave of
//     a SIMD multiplies, its partner reads and stages (the ping-pong of the template's `if (wr == 1) s_barrier`);
//   * the operand slabs are staged as HALF-TILES in the order B(k rows 0-31), A(0-31), B(32-63), A(32-63), each wave two (A of the 128-row tile: one) 1 KiB pieces per
//     half-tile, SIX half-tiles ahead of the phase that runs; the vector-memory queue is never drained in the loop: a counted vmcnt at the odd phases leaves the four
//     youngest half-tiles in flight.
// Hazards (s = phase number, all waves; derived for the staggered groups in DESIGN.md section 4, "8-phase K loop"):
//   RAW  a half-tile is read no earlier than ONE PHASE AFTER the phase whose LOAD segment ended with the wait that retires it (every wave has then passed a barrier
//        behind every other wave's wait): quarters 4k / 4k+2 read the k-halves waited for at phases 4k-1 / 4k+1;
//   WAR  a half-tile is re-staged no earlier than TWO PHASES AFTER the last phase that reads it (B 0-31: read at 4k, restaged at 4k+2; A 0-31: 4k+1 -> 4k+3;
//        B 32-63: 4k+2 -> 4k+4; A 32-63: 4k+3 -> 4k+5): the reading waves' lgkmcnt(0) sits behind the barrier that follows their LOAD segment.
// Same LDS image, same fragment addresses, same MFMA order per accumulator as wgrad256_kernel: bit-identical partial sums (tests/test_gpu_gemm.py).
// ---------------------------------------------------------------------------------------------------------------------------------------------------
// Diagnostic builds (tools/wgrad_diag.sh; never the shipped library).  The schedule's claim is that its result does not depend on WHEN a wave reaches a segment, only on
// the barrier / wait counts between segments -- so a build that delays single waves by pseudo-random amounts at every segment boundary must stay bit-identical:
//   -DWGRAD_DIAG=1   every wave draws, at each of the five segment boundaries of a phase (before its reads, before its staging, before its counted wait, after the
//                    opening barrier, before the closing barrier), an s_sleep of 0 / 64 / 256 / ~1000 cycles from a per-wave LCG seeded by workgroup, wave and the
//                    launch's s_memtime: waves of one group, and the two groups, drift against each other by up to several phase lengths (a phase is ~310 cycles) --
//                    as far as the barriers let them;
//   -DWGRAD_DIAG=2   the same, and the steady-state counted waits are REMOVED (vmcnt(63)): a deliberately opened RAW hazard -- reads may overtake their LDS-DMA --
//                    which the bit-identity test could then report;
//   -DWGRAD_DIAG=3   the steady waits removed and NO delays: the loop at full speed with nothing between a read and its LDS-DMA but the staging distance (six
//                    half-tiles = 1.5 K-tiles ahead): measures how much margin the distance alone leaves.
// Round 5, one run each on an MI355X: 1 bit-identical (16 cases); 2 ALSO bit-identical -- with the waits gone the data still always lands before it is read, i.e. the
// counted waits of the steady loop are a guarantee that this kernel's timing never comes near needing ("Round 5").
#ifndef WGRAD_DIAG
#define WGRAD_DIAG 0
#endif
__device__ __forceinline__ void diag_jitter(unsigned& state) {
#if WGRAD_DIAG == 1 || WGRAD_DIAG == 2
	state = __builtin_amdgcn_readfirstlane(state * 1664525u + 1013904223u);
	const unsigned r = state >> 27;  // 0..31
	if (r == 0) __builtin_amdgcn_s_sleep(15);
	else if (r < 3) __builtin_amdgcn_s_sleep(4);
	else if (r < 7) __builtin_amdgcn_s_sleep(1);
#else
	(void)state;
#endif
}

template <int N> __device__ __forceinline__ void vm_wait_imm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// at most `n` vector-memory operations of this wave stay outstanding (n = what was issued BEHIND the piece that must have landed; fewer is always safe)
__device__ __forceinline__ void vm_wait_dyn(int n) {
	if (n >= 8) vm_wait_imm<8>();
	else if (n >= 6) vm_wait_imm<6>();
	else if (n >= 5) vm_wait_imm<5>();
	else if (n >= 4) vm_wait_imm<4>();
	else if (n >= 3) vm_wait_imm<3>();
	else if (n >= 2) vm_wait_imm<2>();
	else vm_wait_imm<0>();
}

template <int NMF>
__global__ __launch_bounds__(WG_NT) void wgrad256p_kernel(const WgradArgs gin, const WgradExtras ex) {
	WgradArgs g = gin;
	constexpr int TM = 32 * NMF;               // output rows per tile
	constexpr int RA = TM * 2;                 // bytes per k row of the A slab (512 / 256); the B slab has 512
	constexpr int RPA = 1024 / RA;             // k rows per DMA piece of the A slab (2 / 4)
	constexpr int NA = 4 / RPA;                // A pieces per wave and half-tile (2 / 1): 32 k rows = 32 / RPA pieces over 8 waves
	constexpr int A_BYTES = WG_TK * RA;
	constexpr int HM = NMF / 2;
	extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A slab | B slab 32 KiB], 64 KiB apart
	const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wr = w >> 2, wc = w & 3;
	unsigned jit = 0;  // (WGRAD_DIAG builds only)
#if WGRAD_DIAG
	jit = __builtin_amdgcn_readfirstlane((unsigned)__builtin_amdgcn_s_memtime() * 2654435761u + (blockIdx.x * 8u + (unsigned)w) * 40503u);
#endif
	int Klim = g.K;
	if (g.row_limit) Klim = min(g.K, max(*g.row_limit, 0));
	const int nkt = (Klim + WG_TK - 1) / WG_TK;
	const int per_xcd = gridDim.x >> 3;
	const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
	const int ntiles = g.tiles_m * g.tiles_n + ex.tiles2;
	if (item >= ntiles * g.splits) return;
	const int s = item / ntiles, tile = select_problem(g, ex, item - s * ntiles);
	const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
	int kb, ke;
	part_range(nkt, g.splits, s, kb, ke);
	if (kb >= ke) return;

	const size_t k0 = (size_t)kb * WG_TK;
	const uint64_t left = (uint64_t)(Klim - (int)k0);
	const uint64_t ra_bytes = left * (uint64_t)g.lda * 2ull, rb_bytes = left * (uint64_t)g.ldb * 2ull;
	const __amdgpu_buffer_rsrc_t sa = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.A) + k0 * (size_t)g.lda, 0, (unsigned)(ra_bytes < 0x7FFFFFF0ull ? ra_bytes : 0x7FFFFFF0ull), 0x00020000);
	const __amdgpu_buffer_rsrc_t sb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.B) + k0 * (size_t)g.ldb, 0, (unsigned)(rb_bytes < 0x7FFFFFF0ull ? rb_bytes : 0x7FFFFFF0ull), 0x00020000);

	// staging: half-tile (operand, kh) = k rows kh*32 .. +31 of the operand's slab, cut into 1 KiB pieces of RPA (A) / 2 (B) rows; wave w takes pieces w*NA + i (A) /
	// w*2 + i (B).  Source chunk = slot ^ cx(k row): the granule swizzle of the transposing reads, taken chunk-wise (as wgrad256_kernel).
	auto cx = [](int kk) { return ((kk & 3) << 1) | (((kk >> 3) & 1) << 3); };
	unsigned va[2][2], vb[2][2];  // [kh][piece]: byte offsets relative to the K-tile's first row (NA pieces used for A)
#pragma unroll
	for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
		for (int i = 0; i < NA; ++i) {
			const int kk = kh * 32 + (w * NA + i) * RPA + lane / (64 / RPA);
			const int ca = tm * TM + (((lane % (64 / RPA)) ^ cx(kk)) * 8);
			va[kh][i] = ca < g.M ? ((unsigned)kk * (unsigned)g.lda + (unsigned)ca) * 2u : WG_OOB;
		}
#pragma unroll
		for (int i = 0; i < 2; ++i) {
			const int kk = kh * 32 + (w * 2 + i) * 2 + (lane >> 5);
			const int cb = tn * WG_TN + (((lane & 31) ^ cx(kk)) * 8);
			vb[kh][i] = cb < g.N ? ((unsigned)kk * (unsigned)g.ldb + (unsigned)cb) * 2u : WG_OOB;
		}
	}
	const unsigned ka_step = (unsigned)WG_TK * (unsigned)g.lda * 2u, kb_step = (unsigned)WG_TK * (unsigned)g.ldb * 2u;
	// half-tile q of K-tile kt into buffer buf: q = 0 B rows 0-31, 1 A rows 0-31, 2 B rows 32-63, 3 A rows 32-63 (the order the quarters need them)
	auto stage_half = [&](int buf, int kt, auto qc) {
		constexpr int q = decltype(qc)::value, kh = q >> 1;
		char* base = smem + buf * WG_BUF;
		if constexpr ((q & 1) == 0) {
			const unsigned ob = (unsigned)(kt - kb) * kb_step;
#pragma unroll
			for (int i = 0; i < 2; ++i)
				__builtin_amdgcn_raw_ptr_buffer_load_lds(sb, (wg_lds_ptr_t)(base + A_BYTES + kh * 32 * WG_ROWB + (w * 2 + i) * 1024), 16, vb[kh][i] == WG_OOB ? WG_OOB : vb[kh][i] + ob, 0, 0, 0);
		} else {
			const unsigned oa = (unsigned)(kt - kb) * ka_step;
#pragma unroll
			for (int i = 0; i < NA; ++i)
				__builtin_amdgcn_raw_ptr_buffer_load_lds(sa, (wg_lds_ptr_t)(base + kh * 32 * RA + (w * NA + i) * 1024), 16, va[kh][i] == WG_OOB ? WG_OOB : va[kh][i] + oa, 0, 0, 0);
		}
	};
	// fragment addresses: exactly wgrad256_kernel's
	const int fg = lane >> 4, fq4 = (lane >> 2) & 3, fp = lane & 3;
	const int ylane = fq4 | ((fg & 1) << 2);
	unsigned pa[2][8], pb[2][4];
	const unsigned smem_base = (unsigned)(uintptr_t)(wg_lds_ptr_t)smem;
#pragma unroll
	for (int bf = 0; bf < 2; ++bf) {
#pragma unroll
		for (int i = 0; i < NMF; ++i) {
			const int c = wr * NMF + i;
			pa[bf][i] = smem_base + bf * WG_BUF + (unsigned)((8 * fg + fq4) * RA + fp * 8 + (((c & 7) ^ ylane) << 5) + (c >> 3) * 256);
		}
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const int c = wc * 4 + j;
			pb[bf][j] = smem_base + bf * WG_BUF + A_BYTES + (unsigned)((8 * fg + fq4) * WG_ROWB + fp * 8 + (((c & 7) ^ ylane) << 5) + (c >> 3) * 256);
		}
	}

	f32x4 acc[NMF][4];
#pragma unroll
	for (int i = 0; i < NMF; ++i)
#pragma unroll
		for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

	wg_u32x2 bl[4], bh[4], al[HM], ah[HM];  // the B fragments of the current k-step (two quarters), the A fragments of the current quarter
	auto read_b = [&](const unsigned (&qb)[4], auto ksc) {
		constexpr int ks = decltype(ksc)::value;
#pragma unroll
		for (int j = 0; j < 4; ++j) { tr_read<ks * 32 * WG_ROWB>(bl[j], qb[j]); tr_read<(ks * 32 + 4) * WG_ROWB>(bh[j], qb[j]); }
	};
	auto read_a = [&](const unsigned (&qa)[8], auto ksc, auto hc) {
		constexpr int ks = decltype(ksc)::value, h = decltype(hc)::value;
#pragma unroll
		for (int i = 0; i < HM; ++i) { tr_read<ks * 32 * RA>(al[i], qa[h * HM + i]); tr_read<(ks * 32 + 4) * RA>(ah[i], qa[h * HM + i]); }
	};
	auto mul = [&](auto hc) {
		constexpr int h = decltype(hc)::value;
		bf16x8 fb[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) fb[j] = __builtin_bit_cast(bf16x8, (u32x4_t){bl[j][0], bl[j][1], bh[j][0], bh[j][1]});
#pragma unroll
		for (int i = 0; i < HM; ++i) {
			const bf16x8 fa = __builtin_bit_cast(bf16x8, (u32x4_t){al[i][0], al[i][1], ah[i][0], ah[i][1]});
#pragma unroll
			for (int j = 0; j < 4; ++j) acc[h * HM + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa, acc[h * HM + i][j], 0, 0, 0);
		}
	};
	auto bar = [&]() {
		__builtin_amdgcn_sched_barrier(0);
		__builtin_amdgcn_s_barrier();
		__builtin_amdgcn_sched_barrier(0);
	};
	// COMPUTE segment of a phase: the reads of its LOAD segment have been issued; multiply, then the phase's closing barrier
	auto compute = [&](auto hc) {
		bar();
		diag_jitter(jit);
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		__builtin_amdgcn_sched_barrier(0);
		__builtin_amdgcn_s_setprio(1);
		mul(hc);
		__builtin_amdgcn_s_setprio(0);
		diag_jitter(jit);
		bar();
	};
	using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>; using C2 = std::integral_constant<int, 2>; using C3 = std::integral_constant<int, 3>;

	// One K-tile kt (buffer bf, compile time) = phases 4k .. 4k+3.  Phase s stages half-tile s + 6 of the stream: phases 0, 1 the B / A k-half 32-63 of K-tile kt + 1
	// (the other buffer), phases 2, 3 the B / A k-half 0-31 of K-tile kt + 2 (this buffer: read for the last time at phases 0 / 1).  Waits at phases 1 and 3: the
	// half-tiles the NEXT phase reads (k rows 32-63 of kt; k rows 0-31 of kt + 1) have landed, what was staged behind them stays in flight.
	auto ktile = [&](auto bfc, int kt, auto steady) {
		constexpr int bf = decltype(bfc)::value;
		constexpr bool STEADY = decltype(steady)::value;  // kt + 2 < ke: every half-tile this K-tile stages exists (counts are compile-time)
		const int rem = ke - kt;                            // K-tiles left including this one
		constexpr int STEADY_WAIT = WGRAD_DIAG >= 2 ? 63 : 2 * (2 + NA);  // (WGRAD_DIAG = 2, 3: the deliberately opened hazard)
		// phase 0
		diag_jitter(jit);
		read_b(pb[bf], C0{});
		read_a(pa[bf], C0{}, C0{});
		diag_jitter(jit);
		if (STEADY || rem > 1) stage_half(bf ^ 1, kt + 1, C2{});
		compute(C0{});
		// phase 1
		diag_jitter(jit);
		read_a(pa[bf], C0{}, C1{});
		diag_jitter(jit);
		if (STEADY || rem > 1) stage_half(bf ^ 1, kt + 1, C3{});
		diag_jitter(jit);
		// k rows 32-63 of this K-tile (q = 2, 3 of kt) must have landed; behind q = 3 of kt: all four half-tiles of kt + 1 (when it exists)
		if (STEADY) vm_wait_imm<STEADY_WAIT>();
		else vm_wait_dyn(rem > 1 ? 2 * (2 + NA) : 0);
		compute(C1{});
		// phase 2
		diag_jitter(jit);
		read_b(pb[bf], C1{});
		read_a(pa[bf], C1{}, C0{});
		diag_jitter(jit);
		if (STEADY || rem > 2) stage_half(bf, kt + 2, C0{});
		compute(C0{});
		// phase 3
		diag_jitter(jit);
		read_a(pa[bf], C1{}, C1{});
		diag_jitter(jit);
		if (STEADY || rem > 2) stage_half(bf, kt + 2, C1{});
		diag_jitter(jit);
		// k rows 0-31 of kt + 1 (q = 0, 1) must have landed; behind q = 1 of kt + 1: its q = 2, 3 and q = 0, 1 of kt + 2
		if (STEADY) vm_wait_imm<STEADY_WAIT>();
		else vm_wait_dyn(rem > 2 ? 2 * (2 + NA) : (rem > 1 ? (2 + NA) : 0));
		compute(C1{});
	};

	// prologue: K-tile kb whole, k rows 0-31 of kb + 1; k rows 0-31 of kb landed before anybody reads
	stage_half(0, kb, C0{}); stage_half(0, kb, C1{}); stage_half(0, kb, C2{}); stage_half(0, kb, C3{});
	if (kb + 1 < ke) { stage_half(1, kb + 1, C0{}); stage_half(1, kb + 1, C1{}); }
	vm_wait_dyn((2 + NA) + (kb + 1 < ke ? (2 + NA) : 0));
	bar();
	if (wr == 1) bar();  // waves 4-7 run one barrier behind their SIMD partners from here on
	int kt = kb;
	for (; kt + 3 < ke; kt += 2) {  // both K-tiles of the trip stage K-tiles that exist
		ktile(C0{}, kt, std::true_type{});
		ktile(C1{}, kt + 1, std::true_type{});
	}
	for (; kt < ke; kt += 2) {
		ktile(C0{}, kt, std::false_type{});
		if (kt + 1 < ke) ktile(C1{}, kt + 1, std::false_type{});
	}
	if (wr == 0) bar();  // the barrier waves 4-7 still owe

	float* wp = g.ws + ((size_t)item * 8 + w) * (NMF * 4 * 256) + lane * 4;
#pragma unroll
	for (int mt = 0; mt < NMF; ++mt)
#pragma unroll
		for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(acc[mt][j], reinterpret_cast<f32x4*>(wp + (mt * 4 + j) * 256));
}

// dW += alpha * (sum of the parts, in part order).  One thread per accumulator quad: grid = tiles x (8 NMF) workgroups of 256 threads.
template <int NMF>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradArgs gin, const WgradExtras ex) {
	WgradArgs g = gin;
	constexpr int TM = 32 * NMF, QUADS = 8 * NMF * 4 * 64, PER_TILE = QUADS / 256;
	int Klim = g.K;
	if (g.row_limit) Klim = min(g.K, max(*g.row_limit, 0));
	const int nkt = (Klim + WG_TK - 1) / WG_TK;
	const int tile = blockIdx.x / PER_TILE, idx = (blockIdx.x % PER_TILE) * 256 + threadIdx.x;  // idx = ((w * NMF + mt) * 4 + j) * 64 + lane
	const int lane = idx & 63, j = (idx >> 6) & 3, mt = (idx >> 8) % NMF, w = idx / (256 * NMF);
	const int wr = w >> 2, wc = w & 3;
	const int ntiles = g.tiles_m * g.tiles_n + ex.tiles2;  // (before select_problem rewrites tiles_n)
	const int tl = select_problem(g, ex, tile);
	const int tm = tl / g.tiles_n, tn = tl - tm * g.tiles_n;
	const int m = tm * TM + wr * (16 * NMF) + mt * 16 + (lane & 15), n = tn * WG_TN + wc * 64 + j * 16 + (lane >> 4) * 4;
	if (m >= g.M || n >= g.N) return;
	const int per = (nkt + g.splits - 1) / g.splits;
	const int nparts = per > 0 ? min(g.splits, (nkt + per - 1) / per) : 0;  // parts are non-empty up to the first empty one (part_range)
	const float* wp = g.ws + (size_t)tile * (QUADS * 4) + (size_t)idx * 4;
	const size_t pstride = (size_t)ntiles * (QUADS * 4);
	f32x4 sum = {0.f, 0.f, 0.f, 0.f};
	int s = 0;
	const int deep = nparts >= 32 ? (nparts & ~7) : 0;  // many parts (the narrow pair: 64): eight loads in flight -- with four, 16 dependent round trips; few parts: four (eight measured slower)
	for (; s < deep; s += 8) {
		f32x4 t[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const f32x4*>(wp + (size_t)(s + u) * pstride);
#pragma unroll
		for (int u = 0; u < 8; ++u)
#pragma unroll
			for (int r = 0; r < 4; ++r) sum[r] += t[u][r];
	}
	for (; s + 4 <= nparts; s += 4) {  // four loads in flight, added in part order
		const f32x4 t0 = *reinterpret_cast<const f32x4*>(wp + (size_t)s * pstride), t1 = *reinterpret_cast<const f32x4*>(wp + (size_t)(s + 1) * pstride);
		const f32x4 t2 = *reinterpret_cast<const f32x4*>(wp + (size_t)(s + 2) * pstride), t3 = *reinterpret_cast<const f32x4*>(wp + (size_t)(s + 3) * pstride);
#pragma unroll
		for (int r = 0; r < 4; ++r) sum[r] = (((sum[r] + t0[r]) + t1[r]) + t2[r]) + t3[r];
	}
	for (; s < nparts; ++s) {
		const f32x4 t = *reinterpret_cast<const f32x4*>(wp + (size_t)s * pstride);
		sum[0] += t[0]; sum[1] += t[1]; sum[2] += t[2]; sum[3] += t[3];
	}
	if (g.transpose_out) {
		for (int r = 0; r < 4 && n + r < g.N; ++r) g.C[(size_t)(n + r) * g.ldc + m] += g.alpha * sum[r];
		return;
	}
	float* c = g.C + (size_t)m * g.ldc + n;
	if (n + 4 <= g.N && (g.ldc & 3) == 0) {
		f32x4 o = *reinterpret_cast<f32x4*>(c);
		o[0] += g.alpha * sum[0]; o[1] += g.alpha * sum[1]; o[2] += g.alpha * sum[2]; o[3] += g.alpha * sum[3];
		*reinterpret_cast<f32x4*>(c) = o;
	} else {
		for (int r = 0; r < 4 && n + r < g.N; ++r) c[r] += g.alpha * sum[r];
	}
}

std::atomic<int> g_wgrad_pipelined{1};  // 1: wgrad256p_kernel (8-phase schedule), 0: wgrad256_kernel (one barrier per K-tile) -- novic_wgrad_policy, A/B measurements and tests

template <int NMF>
void launch_wgrad(const WgradArgs& g, const WgradExtras& ex, hipStream_t stream) {
	static std::atomic<bool> attr_done{false};
	if (!attr_done) {
		(void)hipFuncSetAttribute((const void*)wgrad256_kernel<NMF>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WG_BUF);
		(void)hipFuncSetAttribute((const void*)wgrad256p_kernel<NMF>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WG_BUF);
		attr_done = true;
	}
	const int ntiles = g.tiles_m * g.tiles_n + ex.tiles2;
	const int grid = ((ntiles * g.splits + 7) / 8) * 8;
	if (g_wgrad_pipelined) hipLaunchKernelGGL(wgrad256p_kernel<NMF>, dim3(grid), dim3(WG_NT), 2 * WG_BUF, stream, g, ex);
	else hipLaunchKernelGGL(wgrad256_kernel<NMF>, dim3(grid), dim3(WG_NT), 2 * WG_BUF, stream, g, ex);
	hipLaunchKernelGGL(wgrad_reduce_kernel<NMF>, dim3(ntiles * (8 * NMF * 4 * 64 / 256)), dim3(256), 0, stream, g, ex);
}

}  // namespace

extern "C" int novic_wgrad_policy(int policy) {  // see include/novic_hip.h
	const int prev = g_wgrad_pipelined;
	if (policy == 0 || policy == 1) g_wgrad_pipelined = policy;
	return prev;
}

// workgroups one launch may have: max_workgroups (0 = the whole chip, 256), rounded down to a multiple of 8
static int wgrad_budget(int max_workgroups) {
	if (max_workgroups <= 0 || max_workgroups > 256) return 256;
	return max_workgroups < 8 ? 8 : max_workgroups / 8 * 8;
}

extern "C" int novic_wgrad_bf16(const void* dY, const void* X, int M, int N, int K, int ldy, int ldx, float* dW, int ldw, float alpha, const int32_t* row_limit, void* ws,
                                uint64_t ws_bytes, int splits_hint, int max_workgroups, hipStream_t stream) {
	NOVIC_CHECK(dY && X && dW && ws, "novic_wgrad_bf16: null pointer");
	NOVIC_CHECK(M >= 1 && N >= 1 && K >= 0, "novic_wgrad_bf16: bad dimensions");
	NOVIC_CHECK(M % 8 == 0 && N % 8 == 0 && ldy % 8 == 0 && ldx % 8 == 0 && ldy >= M && ldx >= N, "novic_wgrad_bf16: M, N and the leading dimensions must be multiples of 8 (16-byte chunks)");
	NOVIC_CHECK((((uintptr_t)dY | (uintptr_t)X | (uintptr_t)dW | (uintptr_t)ws) & 15) == 0, "novic_wgrad_bf16: operands must be 16-byte aligned");
	if (K == 0) return 0;
	WgradArgs g = {};
	g.A = (const bf16*)dY; g.B = (const bf16*)X; g.C = dW;
	g.M = M; g.N = N; g.K = K; g.lda = ldy; g.ldb = ldx; g.ldc = ldw;
	g.transpose_out = 0;
	// an output that is at most 128 wide in one dimension (the feed-forward gradients [128 x 512] / [512 x 128]) runs on 128 x 256 tiles, the narrow
	// dimension as the tile's rows -- for a narrow N that is the transposed product X^T dY, written back transposed by the reduction
	int nmf = 8;
	if (M <= 128 || N <= 128) {
		nmf = 4;
		if (M > 128) {
			g.A = (const bf16*)X; g.B = (const bf16*)dY;
			g.M = N; g.N = M; g.lda = ldx; g.ldb = ldy;
			g.transpose_out = 1;
		}
	}
	NOVIC_CHECK(g.transpose_out ? ldw >= g.M : ldw >= g.N, "novic_wgrad_bf16: ldw smaller than the output's row length");
	const int TMc = 32 * nmf;
	g.tiles_m = (g.M + TMc - 1) / TMc;
	g.tiles_n = (g.N + WG_TN - 1) / WG_TN;
	const int ntiles = g.tiles_m * g.tiles_n;
	NOVIC_CHECK(ntiles <= 256, "novic_wgrad_bf16: more than 256 output tiles (this entry point is for weight-shaped outputs)");
	const int nkt = (K + WG_TK - 1) / WG_TK;
	// one round of the chip; the narrow feed-forward gradients on half of it (their 128 KiB partials per workgroup outweigh the K loop beyond 64 parts:
	// linear2 dW 36.4 -> 31.1 us with 64 parts, 38.1 us with 128)
	const int budget = wgrad_budget(max_workgroups);
	int S = splits_hint > 0 ? splits_hint : (nmf == 4 ? (budget / 2) / ntiles : budget / ntiles);
	if (S > nkt) S = nkt;
	if (S < 1) S = 1;
	NOVIC_CHECK((uint64_t)ntiles * S * (uint64_t)TMc * 256ull * 4ull <= ws_bytes, "novic_wgrad_bf16: scratch too small (tiles x parts x tile bytes)");
	{  // 32-bit buffer offsets span ONE PART's rows of an operand (the descriptors are based at the part's first row)
		const uint64_t part_rows = (uint64_t)((nkt + S - 1) / S) * WG_TK;
		NOVIC_CHECK(part_rows * (uint64_t)ldy * 2 < 0x7FFFFFF0ull && part_rows * (uint64_t)ldx * 2 < 0x7FFFFFF0ull,
		            "novic_wgrad_bf16: one part's rows of an operand must be smaller than 2 GiB (32-bit buffer offsets); raise splits_hint");
	}
	g.splits = S;
	g.alpha = alpha;
	g.row_limit = row_limit;
	g.ws = (float*)ws;
	const WgradExtras none = {};
	if (nmf == 8) launch_wgrad<8>(g, none, stream);
	else launch_wgrad<4>(g, none, stream);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

// One problem as the kernel sees it: an output that is at most 128 wide in one dimension runs on 128 x 256 tiles with the narrow dimension as the tile's rows -- for a
// narrow N that is the transposed product X^T dY, written back transposed by the reduction.
struct WgradProblem { const bf16* A; const bf16* B; float* C; int M, N, lda, ldb, ldc, transpose, narrow; };
static WgradProblem wgrad_problem(const void* dY, const void* X, int M, int N, int ldy, int ldx, float* dW, int ldw) {
	WgradProblem p = {(const bf16*)dY, (const bf16*)X, dW, M, N, ldy, ldx, ldw, 0, (M <= 128 || N <= 128) ? 1 : 0};
	if (p.narrow && M > 128) { p.A = (const bf16*)X; p.B = (const bf16*)dY; p.M = N; p.N = M; p.lda = ldx; p.ldb = ldy; p.transpose = 1; }
	return p;
}

// n problems (1 <= n <= 4) over the same K token rows in one launch pair: shared by novic_wgrad2_bf16 and novic_wgradn_bf16
static int wgrad_many(const char* who, const WgradProblem* ps, int n, int K, float alpha, const int32_t* row_limit, void* ws, uint64_t ws_bytes, int max_workgroups, hipStream_t stream) {
	const int nmf = ps[0].narrow ? 4 : 8, TMc = 32 * nmf;
	WgradArgs g = {};
	g.A = ps[0].A; g.B = ps[0].B; g.C = ps[0].C;
	g.M = ps[0].M; g.N = ps[0].N; g.K = K; g.lda = ps[0].lda; g.ldb = ps[0].ldb; g.ldc = ps[0].ldc; g.transpose_out = ps[0].transpose;
	g.tiles_m = (ps[0].M + TMc - 1) / TMc; g.tiles_n = (ps[0].N + WG_TN - 1) / WG_TN;
	WgradExtras ex = {};
	ex.nextra = n - 1;
	ex.tiles2 = 0;
	uint64_t ldmax = (uint64_t)max(ps[0].lda, ps[0].ldb);
	for (int i = 1; i < n; ++i) {
		WgradExtra& e = ex.e[i - 1];
		e.A = ps[i].A; e.B = ps[i].B; e.C = ps[i].C;
		e.M = ps[i].M; e.N = ps[i].N; e.lda = ps[i].lda; e.ldb = ps[i].ldb; e.ldc = ps[i].ldc; e.transpose_out = ps[i].transpose;
		e.tiles_n = (ps[i].N + WG_TN - 1) / WG_TN;
		e.tiles = ((ps[i].M + TMc - 1) / TMc) * e.tiles_n;
		ex.tiles2 += e.tiles;
		ldmax = max(ldmax, (uint64_t)max(ps[i].lda, ps[i].ldb));
	}
	const int ntiles = g.tiles_m * g.tiles_n + ex.tiles2;
	if (ntiles > 256) { novic_set_error("novic_wgrad*_bf16: more than 256 output tiles"); return -22; }
	const int nkt = (K + WG_TK - 1) / WG_TK;
	int S = wgrad_budget(max_workgroups) / ntiles;  // one round of the chip (or of the caller's share of it) over all problems
	if (nmf == 4 && n <= 2 && S > 64) S = 64;       // (the narrow feed-forward pair: beyond 64 parts its 128 KiB partials per workgroup outweigh the K loop -- 4 tiles x 64, as ever)
	if (S > nkt) S = nkt;
	if (S < 1) S = 1;
	if ((uint64_t)ntiles * S * (uint64_t)TMc * 256ull * 4ull > ws_bytes) { novic_set_error("novic_wgrad*_bf16: scratch too small (tiles x parts x tile bytes)"); return -22; }
	{
		const uint64_t part_rows = (uint64_t)((nkt + S - 1) / S) * WG_TK;
		if (part_rows * ldmax * 2 >= 0x7FFFFFF0ull) { novic_set_error("novic_wgrad*_bf16: one part's rows of an operand must be smaller than 2 GiB (32-bit buffer offsets)"); return -22; }
	}
	(void)who;
	g.splits = S;
	g.alpha = alpha;
	g.row_limit = row_limit;
	g.ws = (float*)ws;
	if (nmf == 8) launch_wgrad<8>(g, ex, stream);
	else launch_wgrad<4>(g, ex, stream);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_wgrad2_bf16(const void* dY1, const void* X1, int M1, int N1, int ldy1, int ldx1, float* dW1, int ldw1, const void* dY2, const void* X2, int M2, int N2,
                                 int ldy2, int ldx2, float* dW2, int ldw2, int K, float alpha, const int32_t* row_limit, void* ws, uint64_t ws_bytes, int max_workgroups,
                                 hipStream_t stream) {
	NOVIC_CHECK(dY1 && X1 && dW1 && dY2 && X2 && dW2 && ws, "novic_wgrad2_bf16: null pointer");
	NOVIC_CHECK(M1 >= 1 && N1 >= 1 && M2 >= 1 && N2 >= 1 && K >= 0, "novic_wgrad2_bf16: bad dimensions");
	NOVIC_CHECK(((M1 | N1 | M2 | N2 | ldy1 | ldx1 | ldy2 | ldx2) & 7) == 0 && ldy1 >= M1 && ldx1 >= N1 && ldy2 >= M2 && ldx2 >= N2,
	            "novic_wgrad2_bf16: dimensions and leading dimensions must be multiples of 8 (16-byte chunks)");
	NOVIC_CHECK((((uintptr_t)dY1 | (uintptr_t)X1 | (uintptr_t)dW1 | (uintptr_t)dY2 | (uintptr_t)X2 | (uintptr_t)dW2 | (uintptr_t)ws) & 15) == 0,
	            "novic_wgrad2_bf16: operands must be 16-byte aligned");
	const WgradProblem ps[2] = {wgrad_problem(dY1, X1, M1, N1, ldy1, ldx1, dW1, ldw1), wgrad_problem(dY2, X2, M2, N2, ldy2, ldx2, dW2, ldw2)};
	NOVIC_CHECK(ps[0].narrow == ps[1].narrow, "novic_wgrad2_bf16: both outputs at most 128 wide in one dimension (128 x 256 tiles), or neither (256 x 256 tiles)");
	NOVIC_CHECK((ps[0].transpose ? ldw1 >= ps[0].M : ldw1 >= ps[0].N) && (ps[1].transpose ? ldw2 >= ps[1].M : ldw2 >= ps[1].N), "novic_wgrad2_bf16: ldw smaller than the output's row length");
	if (K == 0) return 0;
	return wgrad_many("novic_wgrad2_bf16", ps, 2, K, alpha, row_limit, ws, ws_bytes, max_workgroups, stream);
}

extern "C" int novic_wgradn_bf16(const novic_wgrad_problem_t* problems, int n, int K, float alpha, const int32_t* row_limit, void* ws, uint64_t ws_bytes, int max_workgroups,
                                 hipStream_t stream) {
	NOVIC_CHECK(problems && ws, "novic_wgradn_bf16: null pointer");
	NOVIC_CHECK(n >= 1 && n <= 4 && K >= 0, "novic_wgradn_bf16: 1 to 4 problems");
	WgradProblem ps[4];
	for (int i = 0; i < n; ++i) {
		const novic_wgrad_problem_t& q = problems[i];
		NOVIC_CHECK(q.dY && q.X && q.dW && q.M >= 1 && q.N >= 1, "novic_wgradn_bf16: null pointer or empty problem");
		NOVIC_CHECK(((q.M | q.N | q.ldy | q.ldx) & 7) == 0 && q.ldy >= q.M && q.ldx >= q.N, "novic_wgradn_bf16: dimensions and leading dimensions must be multiples of 8 (16-byte chunks)");
		NOVIC_CHECK((((uintptr_t)q.dY | (uintptr_t)q.X | (uintptr_t)q.dW) & 15) == 0, "novic_wgradn_bf16: operands must be 16-byte aligned");
		ps[i] = wgrad_problem(q.dY, q.X, q.M, q.N, q.ldy, q.ldx, q.dW, q.ldw);
		NOVIC_CHECK(ps[i].narrow == ps[0].narrow, "novic_wgradn_bf16: all outputs at most 128 wide in one dimension (128 x 256 tiles), or none (256 x 256 tiles)");
		NOVIC_CHECK(ps[i].transpose ? q.ldw >= ps[i].M : q.ldw >= ps[i].N, "novic_wgradn_bf16: ldw smaller than the output's row length");
	}
	NOVIC_CHECK(((uintptr_t)ws & 15) == 0, "novic_wgradn_bf16: scratch must be 16-byte aligned");
	if (K == 0) return 0;
	return wgrad_many("novic_wgradn_bf16", ps, n, K, alpha, row_limit, ws, ws_bytes, max_workgroups, stream);
}
