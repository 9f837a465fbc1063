// bf16 MFMA GEMM, 256x256x64 tile, for large K-contiguous x K-contiguous problems (A [M][K], B [N][K]: forward linears, and input-gradient GEMMs
// against a transposed weight shadow).  Same arithmetic as gemm.hip (v_mfma_f32_16x16x32_bf16, K accumulated in the same order, the same fused
// epilogues) -- the results are bit-identical; what changes is how the chip is fed:
//
//  * 512 threads = 8 waves (2 along M x 4 along N), each wave a 128x64 sub-tile = 8x4 MFMA tiles: 64 MFMAs per 24 ds_read_b128 per K-tile, i.e.
//    1.5x fewer LDS read bytes per FLOP than the 128^2 / 64x64-per-wave kernel, whose LDS pipe co-limited the MFMA pipe.
//  * LDS-DMA staging (buffer_load_dwordx4 ... lds): the operand tiles go HBM/L2 -> LDS without passing through VGPRs (no ds_write pass, 64
//    staging VGPRs freed for accumulators).  An LDS-DMA wave instruction writes 1 KiB lane-linearly, so the XOR swizzle that keeps ds_read_b128
//    conflict-free (16-B chunk ^= row & 7 in a 128-B row) is applied on the SOURCE address: the lane that fills LDS slot s of row r fetches
//    global chunk s ^ (r & 7) of that row.  Every LDS row is still one whole 128-B global line.
//  * The B tile is laid out in LDS in a PERMUTED row order so that, with the MFMA issued "swapped" (first operand = B rows), a lane ends up with
//    twice 8 consecutive output columns of a row and four neighbouring lanes with 32: the epilogue stores straight from the accumulators in
//    whole 64-B sectors, no LDS round trip, so the LDS buffers can already hold the next tile's operands.
//  * Persistent workgroups (one per CU, 128 KiB LDS): the K pipeline runs across output tiles -- the first K-tile of the next output tile is
//    requested during the last K-tile of the current one, and its epilogue stores are issued after the next tile's loads.
//  * XCD-aware tile order as in gemm.hip: the 32 workgroups of an XCD walk a contiguous range of a chunk-major tile sequence together.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "gemm_epilogue.hpp"

namespace {

constexpr int TM = 256, TK = 64, NT2 = 512;
constexpr int OP_BYTES = 256 * TK * 2;   // the A tile: 256 rows x 128 B
// NTW = MFMA column tiles per wave: 4 -> 256 x 256 output tile, 3 -> 256 x 192 (N = 768 = 4 x 192 gives 200 tiles on 256 CUs where 256-wide tiles give 150)
template <int NTW> constexpr int tn_of() { return 64 * NTW; }
template <int NTW> constexpr int buf_bytes() { return OP_BYTES + tn_of<NTW>() * TK * 2; }  // A tile | B tile
// B rows in NATURAL order in LDS (a lane then holds columns j*16 + fq*4 .. +3 of its row in acc[mt][j]) for the 192-wide tile and for the fp32
// residual epilogue of the 256-wide one: four neighbouring lanes write 64 contiguous bytes of a row.  The permuted order (8 consecutive bf16
// columns per lane) would leave fp32 stores as 16-byte pieces 32 bytes apart: ViT-L/14 proj [65792 x 1024 x 1024] 300 us, fc2 [.. x 4096] 704 us.
template <int EPI> constexpr bool is_resid() { return EPI == NOVIC_EPI_RESID_F32 || EPI == NOVIC_EPI_RESID_F16; }  // out = resid + linear, in fp32 or (ABI 11) in half
template <int EPI, int NTW> constexpr bool natural_b() { return NTW != 4 || is_resid<EPI>(); }
// element / 4-vector type of the residual stream an epilogue reads and writes
template <int EPI> struct ResidT { typedef float elem; typedef f32x4 vec4; };
template <> struct ResidT<NOVIC_EPI_RESID_F16> { typedef f16 elem; typedef f16x4 vec4; };
constexpr unsigned OOB2 = 0x80000000u;   // operands are < 2 GiB, so this offset (+ any K offset) is out of range -> the load returns zeros

struct Gemm256Args {
	const bf16* A;
	const bf16* B;
	int M, N, K;
	int lda, ldb;
	unsigned a_bytes, b_bytes;
	int tiles_m, tiles_n, group_n, nk;
	int tail_first, tail_split;  // tail_split > 1: tiles [tail_first, tiles) are not run whole -- workgroup b < (tiles - tail_first) * tail_split multiplies
	float* ws;                   // K range b % tail_split of tile tail_first + b / tail_split into ws[b][wave][8][4][64 lanes][4] (gemm256_tail_kernel finishes them)
	int row_base;                // rows of the caller's problem in front of this launch's (an A operand beyond the 2 GiB a buffer descriptor spans runs as several launches): *ep.row_limit counts from there
	int tail_dyn;                // 1: the row count is a DEVICE int (ep.row_limit), so tail_first / tail_split are worked out by every workgroup from the clamped
	unsigned long long ws_bytes; //    tile count (plan_tail: the host's rule) instead of by the host
	unsigned long long* trace;  // diagnostic: [workgroup][32 tiles][4] wall-clock stamps (100 MHz), null = off (novic_gemm256_trace)
	int ncu;                    // workgroups the persistent grid may have (novic_epilogue_t.max_workgroups, else novic_persistent_cus: 256 = the whole chip): a round of tiles is this many
	int pipelined;              // host only: the 8-phase kernel (the process-wide switch, read ONCE per call by plan256)
	novic_epilogue_t ep;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void tile_coords(const Gemm256Args& g, int lid, int& tm, int& tn) {
	const int per_chunk = g.tiles_m * g.group_n;
	const int full = (g.tiles_n / g.group_n) * per_chunk;
	if (lid < full) {
		const int c = lid / per_chunk, rem = lid - c * per_chunk;
		tm = rem / g.group_n;
		tn = c * g.group_n + (rem - tm * g.group_n);
	} else {
		const int wt = g.tiles_n % g.group_n, rem = lid - full;
		tm = rem / wt;
		tn = (g.tiles_n / g.group_n) * g.group_n + (rem - tm * wt);
	}
}

// Interior 256 x 256 tile, bf16 output: see store_tile.  ACT and HAS_BIAS are compile-time so that the loop body holds exactly one activation.
// MT = MFMA row tiles per wave: 8 -> the wave's 128 rows of a 256-row tile, 4 -> 64 rows of a 128-row tile (gemm256p_kernel<EPI, 4>)
template <int ACT, bool HAS_BIAS, int MT = 8>
__device__ __forceinline__ void store_plain(const Gemm256Args& g, int m0, int n0, int wr, int wc, int fr, int fq, f32x4 (&acc)[MT][4], char* scratch) {
	const int lane = fq * 16 + fr;
	bf16* p = (bf16*)g.ep.c + (size_t)(m0 + wr * (MT * 16) + (lane >> 3)) * g.ep.ldc + (n0 + wc * 64 + (lane & 7) * 8);
	const size_t step = (size_t)8 * g.ep.ldc;
	float bias[2][8];
#pragma unroll
	for (int hp = 0; hp < 2; ++hp) {
		const float* bp = (HAS_BIAS && g.ep.bias) ? (const float*)g.ep.bias + n0 + wc * 64 + hp * 32 + fq * 8 : nullptr;
		const f32x4 b0 = bp ? *reinterpret_cast<const f32x4*>(bp) : (f32x4){0.f, 0.f, 0.f, 0.f}, b1 = bp ? *reinterpret_cast<const f32x4*>(bp + 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
		for (int i = 0; i < 4; ++i) { bias[hp][i] = b0[i]; bias[hp][4 + i] = b1[i]; }
	}
	auto fin = [&](float v, float b) -> bf16 {
		if (HAS_BIAS) v += b;
		if (ACT == NOVIC_ACT_GELU) v = gelu_erf(v);
		else if (ACT == NOVIC_ACT_QUICKGELU) v = quick_gelu(v);
		else if (ACT == NOVIC_ACT_GELU_TANH) v = gelu_tanh(v);
		return (bf16)v;
	};
#pragma unroll
	for (int q = 0; q < MT / 2; ++q) {
#pragma unroll
		for (int mtl = 0; mtl < 2; ++mtl)
#pragma unroll
			for (int hp = 0; hp < 2; ++hp) {
				const f32x4 lo = acc[2 * q + mtl][2 * hp], hi = acc[2 * q + mtl][2 * hp + 1];
				bf16x8 o = {fin(lo[0], bias[hp][0]), fin(lo[1], bias[hp][1]), fin(lo[2], bias[hp][2]), fin(lo[3], bias[hp][3]),
				            fin(hi[0], bias[hp][4]), fin(hi[1], bias[hp][5]), fin(hi[2], bias[hp][6]), fin(hi[3], bias[hp][7])};
				const int r = mtl * 16 + fr, sl = hp * 4 + fq;
				*reinterpret_cast<bf16x8*>(scratch + r * 128 + ((sl ^ (r & 7)) << 4)) = o;
			}
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const int r = j * 8 + (lane >> 3), sl = lane & 7;
			const bf16x8 o = *reinterpret_cast<const bf16x8*>(scratch + r * 128 + ((sl ^ (r & 7)) << 4));
			// Always non-temporal (a write-back variant was measured in round 4 -- towers equal to 0.3 %, the training step 6.51 -> 6.57 ms -- and removed in round 5).
			// (Never select the policy by a run-time `if`: hipcc merges the two stores into ONE plain store -- same address, same value, the hint is only metadata -- and
			// every interior tile silently loses its `nt`: logits GEMM fetch 201 -> 427 MB per launch, round 4's PMC pass.  tools/audit_vmcnt.py counts the `nt` stores.)
#if GEMM256_DIAG_NO_STORES  // diagnostic build only (tools/gemm_timeline.py under $NOVIC_HIP_LIB): everything but the stores themselves -- is the epilogue bound by the CU's store path?
			if (g.ep.ldc == -12345) *reinterpret_cast<bf16x8*>(p) = o;
#else
			__builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(p));
#endif
			p += step;
		}
	}
}

// Column ownership after the swapped MFMA with the permuted B rows: lane (fr, fq) holds, of output row mt*16 + fr, the 8 consecutive columns
// hp*32 + fq*8 .. +7 in acc[mt][2*hp] (first four) and acc[mt][2*hp + 1] (last four), hp = 0, 1.  One store instruction therefore writes, per
// row, the 64 contiguous bytes (bf16) of four neighbouring lanes -- whole 64-B sectors instead of 8-B pieces 32 B apart.
// Returns how many vector-memory instructions at the END of the wave's issue order are this epilogue's stores with nothing younger behind them (16 on
// the two interior fast paths, 0 = unknown): the next tile's first K-tile wait may leave that many outstanding -- they are in-order behind the
// LDS-DMA it actually waits for -- instead of draining the stores (256 KiB per tile with the fp32 residual epilogue) before its first barrier.
template <int EPI, int NTW, int MT = 8>
__device__ __forceinline__ int store_tile(const Gemm256Args& g, int m0, int n0, int wr, int wc, int fr, int fq, f32x4 (&acc)[MT][NTW], char* scratch) {
	constexpr int TN = tn_of<NTW>(), TMR = MT * 32, WROWS = MT * 16;  // rows of the tile / of a wave's share of it
	if constexpr (is_resid<EPI>() && (NTW == 4 || EPI == NOVIC_EPI_RESID_F32)) {  // (the half stream runs on 256-wide tiles only: plan256)
		typedef typename ResidT<EPI>::elem RT;   // fp32, or half: then a lane's four columns are 8 bytes and an instruction covers 4 rows x 128 contiguous bytes -- whole lines still
		typedef typename ResidT<EPI>::vec4 RT4;
		constexpr bool HALF = EPI == NOVIC_EPI_RESID_F16;
		if (m0 + TMR <= g.M && n0 + TN <= g.N && (g.ep.ldc & 3) == 0 && (g.ep.ldr & 3) == 0 && (!g.ep.bias || (((uintptr_t)g.ep.bias & 15) == 0))) {
			// interior tile: the bias once, the residual of four row groups at a time requested before any of it is used (one memory round trip per
			// half tile instead of one per row group: 82 -> ~60 us per launch inside the ViT, where nothing else hides them).  Same arithmetic, in
			// the same order, as epilogue4<RESID_F32>.
			if constexpr (NTW == 4) {
				// 256-wide tile: whole lines.  Straight from the accumulators a load / store instruction covers 16 rows x 64 bytes -- half of each
				// line it touches, and the CU's memory path works per line: the 512 KiB of residual in + out of a tile took 20 us (26 GB/s per CU,
				// against 44 GB/s in the K loop), ViT-L/14 proj [65792 x 1024 x 1024] 260 us with 23 us of K loop per tile.  So the accumulators of
				// one 16-row group at a time are re-laid through the wave's 4 KiB LDS corner (16-byte chunks XOR-swizzled by row: conflict-free both
				// ways) into the layout of the lines -- lane L holds, of rows L/16 + 4i, the four columns 4 (L % 16) -- in which the residual is
				// read and the result written: every instruction 4 rows x 256 contiguous bytes, the residual of the next row group in flight
				// while this one is finished.  Same arithmetic per element, in the same order, as epilogue4<RESID_F32>.
				const int lane = fq * 16 + fr, lr = lane >> 4, lc = lane & 15;
				const int mw = m0 + wr * WROWS, nw = n0 + wc * 64 + lc * 4;
				const RT* R = (const RT*)g.ep.resid + (size_t)(mw + lr) * g.ep.ldr + nw;
				RT* C = (RT*)g.ep.c + (size_t)(mw + lr) * g.ep.ldc + nw;
				const f32x4 bb = g.ep.bias ? *reinterpret_cast<const f32x4*>((const float*)g.ep.bias + nw) : (f32x4){0.f, 0.f, 0.f, 0.f};
				const DropoutDesc d = {g.ep.drop_p, g.ep.seed_lo, g.ep.seed_hi, g.ep.drop_site};
				const bool drop = !HALF && g.ep.drop_p > 0.f;
				constexpr int PD = 1;  // row groups of residual in flight ahead of the one being finished (16 VGPRs each; 3 ahead measured the same: 249 us)
				RT4 rv[PD + 1][4];
#pragma unroll
				for (int p = 0; p < PD; ++p)
#pragma unroll
					for (int i = 0; i < 4; ++i) rv[p][i] = *reinterpret_cast<const RT4*>(R + (size_t)(p * 16 + 4 * i) * g.ep.ldr);
#pragma unroll
				for (int mt = 0; mt < MT; ++mt) {
					if (mt + PD < MT) {
#pragma unroll
						for (int i = 0; i < 4; ++i) rv[(mt + PD) % (PD + 1)][i] = *reinterpret_cast<const RT4*>(R + (size_t)((mt + PD) * 16 + 4 * i) * g.ep.ldr);
					}
#pragma unroll
					for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(scratch + fr * 256 + (((j * 4 + fq) ^ fr) << 4)) = acc[mt][j];
					f32x4 a4[4];
#pragma unroll
					for (int i = 0; i < 4; ++i) a4[i] = *reinterpret_cast<const f32x4*>(scratch + (lr + 4 * i) * 256 + ((lc ^ (lr + 4 * i)) << 4));
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						const int row = mt * 16 + 4 * i;  // + lr: in R / C already
						if constexpr (HALF) {  // as epilogue4<RESID_F16>
							f16x4 o;
#pragma unroll
							for (int r = 0; r < 4; ++r) o[r] = resid_f16_elem(rv[mt % (PD + 1)][i][r], a4[i][r] + bb[r]);
							__builtin_nontemporal_store(o, reinterpret_cast<f16x4*>(C + (size_t)row * g.ep.ldc));
						} else {
						float sc[4] = {1.f, 1.f, 1.f, 1.f};
						if (drop) dropout_scale4(d, (uint64_t)(mw + lr + row) * g.N + nw, sc);
						float v[4];
#pragma unroll
						for (int r = 0; r < 4; ++r) v[r] = (float)rv[mt % (PD + 1)][i][r] + bf16_round(a4[i][r] + bb[r]) * sc[r];
						st_f32x4((float*)(C + (size_t)row * g.ep.ldc), v, true, 4);
						}
					}
					__builtin_amdgcn_sched_barrier(0);
				}
				return 4 * (PD + 1);  // behind the last residual loads: the stores of the last PD + 1 row groups
			}
			if constexpr (NTW != 4) {  // (the 256 x 192 tile: MT = 8 only)
			const int mb = m0 + wr * WROWS + fr, nb = n0 + wc * (16 * NTW);
			// columns of acc[mt][j] inside the wave's strip: natural B order (natural_b)
			auto col = [&](int j) { return j * 16 + fq * 4; };
			const float* R = (const float*)g.ep.resid + (size_t)mb * g.ep.ldr + nb;
			float* C = (float*)g.ep.c + (size_t)mb * g.ep.ldc + nb;
			f32x4 bj[NTW];
#pragma unroll
			for (int j = 0; j < NTW; ++j) bj[j] = g.ep.bias ? *reinterpret_cast<const f32x4*>((const float*)g.ep.bias + nb + col(j)) : (f32x4){0.f, 0.f, 0.f, 0.f};
			const DropoutDesc d = {g.ep.drop_p, g.ep.seed_lo, g.ep.seed_hi, g.ep.drop_site};
			const bool drop = g.ep.drop_p > 0.f;  // the mask hash sits behind ONE uniform branch per row group, not inside every element group
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				f32x4 rv[4][NTW];
#pragma unroll
				for (int i = 0; i < 4; ++i)
#pragma unroll
					for (int j = 0; j < NTW; ++j) rv[i][j] = *reinterpret_cast<const f32x4*>(R + (size_t)((h * 4 + i) * 16) * g.ep.ldr + col(j));
#pragma unroll
				for (int i = 0; i < 4; ++i)
#pragma unroll
					for (int j = 0; j < NTW; ++j) {
						float sc[4] = {1.f, 1.f, 1.f, 1.f};
						if (drop) dropout_scale4(d, (uint64_t)(mb + (h * 4 + i) * 16) * g.N + nb + col(j), sc);
						const f32x4 a4 = acc[h * 4 + i][j];
						float v[4];
#pragma unroll
						for (int r = 0; r < 4; ++r) v[r] = rv[i][j][r] + bf16_round(a4[r] + bj[j][r]) * sc[r];
						st_f32x4(C + (size_t)((h * 4 + i) * 16) * g.ep.ldc + col(j), v, true, 4);
					}
				__builtin_amdgcn_sched_barrier(0);
			}
			return 4 * NTW;
			}
		}
	}
	if constexpr (natural_b<EPI, NTW>()) {  // B rows in natural order: a lane holds columns j*16 + fq*4 .. +3 of its row in acc[mt][j]
		epilogue_dispatch<EPI>(g.ep, [&](auto act_c, auto drop_c) {
			constexpr int ACT = decltype(act_c)::value, DROP = decltype(drop_c)::value;
#pragma unroll
			for (int mt = 0; mt < MT; ++mt) {
				const int m = m0 + wr * WROWS + mt * 16 + fr;
#pragma unroll
				for (int j = 0; j < NTW; ++j) {
					const int n = n0 + wc * (16 * NTW) + j * 16 + fq * 4;
					if (m >= g.M || n >= g.N) continue;
					float v[4] = {acc[mt][j][0], acc[mt][j][1], acc[mt][j][2], acc[mt][j][3]};
					epilogue4<EPI, ACT, DROP>(g.ep, m, n, g.N, v);
				}
				__builtin_amdgcn_sched_barrier(0);
			}
		});
		return 0;
	} else {
	const bool plain = EPI == NOVIC_EPI_STORE_BF16 && (g.ep.ldc & 7) == 0 && (!g.ep.bias || (((uintptr_t)g.ep.bias & 15) == 0));
	if (plain && m0 + TMR <= g.M && n0 + TN <= g.N) {
		// Interior tile, bf16 output (+ bias, + GELU / QuickGELU): the wave's 128 x 64 sub-tile goes out 32 rows at a time through a wave-private
		// 4 KiB corner of LDS (16-B slots XOR-swizzled by row), so that every store instruction writes 8 rows x 128 contiguous bytes -- whole
		// lines, which is what makes the non-temporal policy cheap: streamed out without displacing the B chunk / A panels from L2 (L2 fetch
		// 0.60 -> 0.28 GB on the logits GEMM) and without the masked partial-line write requests that 64-B pieces turn into.
		// ONE branch on (activation, bias) around the whole sub-tile: tested per element, the three activation bodies were inlined 128 times
		// (25 k instructions, 1.5 k branches per kernel) and the store phase took 6.6 us per tile -- a third of the kernel -- fetching instructions.
		if (g.ep.act == NOVIC_ACT_NONE) {
			if (g.ep.bias) store_plain<NOVIC_ACT_NONE, true, MT>(g, m0, n0, wr, wc, fr, fq, acc, scratch);
			else store_plain<NOVIC_ACT_NONE, false, MT>(g, m0, n0, wr, wc, fr, fq, acc, scratch);
		} else if (g.ep.act == NOVIC_ACT_GELU) {
			store_plain<NOVIC_ACT_GELU, true, MT>(g, m0, n0, wr, wc, fr, fq, acc, scratch);
		} else if (g.ep.act == NOVIC_ACT_GELU_TANH) {
			store_plain<NOVIC_ACT_GELU_TANH, true, MT>(g, m0, n0, wr, wc, fr, fq, acc, scratch);
		} else {
			store_plain<NOVIC_ACT_QUICKGELU, true, MT>(g, m0, n0, wr, wc, fr, fq, acc, scratch);
		}
		return 2 * MT;  // (four stores per 32-row round)
	}
	// edge tiles / the other epilogues: straight from the accumulators, 8 consecutive columns per lane and row
	const bool raw8 = plain && !g.ep.bias && g.ep.act == NOVIC_ACT_NONE;
	epilogue_dispatch<EPI>(g.ep, [&](auto act_c, auto drop_c) {
		constexpr int ACT = decltype(act_c)::value, DROP = decltype(drop_c)::value;
#pragma unroll
		for (int mt = 0; mt < MT; ++mt) {
			const int m = m0 + wr * WROWS + mt * 16 + fr;
#pragma unroll
			for (int hp = 0; hp < 2; ++hp) {
				const int n = n0 + wc * 64 + hp * 32 + fq * 8;
				if (m >= g.M || n >= g.N) continue;
				const f32x4 lo = acc[mt][2 * hp], hi = acc[mt][2 * hp + 1];
				if (raw8 && n + 8 <= g.N) {
					bf16x8 o = {(bf16)lo[0], (bf16)lo[1], (bf16)lo[2], (bf16)lo[3], (bf16)hi[0], (bf16)hi[1], (bf16)hi[2], (bf16)hi[3]};
					*reinterpret_cast<bf16x8*>((bf16*)g.ep.c + (size_t)m * g.ep.ldc + n) = o;
				} else {
					float v0[4] = {lo[0], lo[1], lo[2], lo[3]}, v1[4] = {hi[0], hi[1], hi[2], hi[3]};
					epilogue4<EPI, ACT, DROP>(g.ep, m, n, g.N, v0);
					if (n + 4 < g.N) epilogue4<EPI, ACT, DROP>(g.ep, m, n + 4, g.N, v1);
				}
			}
			__builtin_amdgcn_sched_barrier(0);  // one row group at a time: hoisting every group's loads / mask state to the top spills
		}
	});
	return 0;
	}
}

// The K-split of the tiles behind the last whole round, decided from the tile count (novic_gemm256_try applies the same rule on the host when the row count is
// a host number): up to 64 tail tiles, each cut into S = min(256 / tail, nk / 4) non-empty parts, if the scratch holds them.
__device__ __forceinline__ void plan_tail(Gemm256Args& g) {
	const int ntiles = g.tiles_m * g.tiles_n;
	g.tail_first = ntiles;
	g.tail_split = 0;
	const int tail = ntiles % g.ncu;
	if (ntiles <= g.ncu || tail == 0 || tail > 64) return;
	int S = g.ncu / tail;
	if (S > g.nk / 4) S = g.nk / 4;
	if (S < 2) return;
	const int per = (g.nk + S - 1) / S;
	S = (g.nk + per - 1) / per;
	if (S < 2 || (unsigned long long)tail * S * 65536ull * 4ull > g.ws_bytes) return;  // (256-row tiles only: the device-planned tail belongs to the training step's GEMMs)
	g.tail_first = ntiles - tail;
	g.tail_split = S;
}

template <int EPI, int NTW>
__global__ __launch_bounds__(NT2) void gemm256_kernel(const Gemm256Args gin) {
	constexpr int TN = tn_of<NTW>(), BUF_BYTES = buf_bytes<NTW>();
	Gemm256Args g = gin;
	if (g.ep.row_limit) {  // only the first *row_limit rows of A / C take part: fewer row tiles, for every workgroup alike
		const int lim = *g.ep.row_limit - g.row_base;
		g.M = lim < g.M ? (lim > 0 ? lim : 0) : g.M;
		g.tiles_m = (g.M + TM - 1) / TM;
		if (g.tail_dyn) plan_tail(g);
	}
	extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][A tile | B tile] 128 KiB + 8 x 4 KiB epilogue staging = the CU's whole 160 KiB
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const int wr = w >> 2, wc = w & 3, fr = lane & 15, fq = lane >> 4;

	// this workgroup's tiles: XCD x owns a contiguous range of the tile sequence, its workgroups take the range round-robin.  With a K-split tail
	// (host: no row_limit, tail_first a multiple of 256) the whole tiles stop at tail_first and the workgroup may own one partial item behind them.
	const int ntiles = g.tiles_m * g.tiles_n;
	const bool split = g.tail_split > 1 && (!g.ep.row_limit || g.tail_dyn);
	const int nwhole = split ? g.tail_first : ntiles;
	const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
	const int q = nwhole >> 3, rm = nwhole & 7;
	const int xbeg = xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q;
	const int xcnt = q + (xcd < rm ? 1 : 0);
	const int nmain = slot < xcnt ? (xcnt - slot + nslots - 1) / nslots : 0;
	int ptile = -1, pkb = 0, pke = 0;
	if (split && (int)blockIdx.x < (ntiles - nwhole) * g.tail_split) {
		const int per = (g.nk + g.tail_split - 1) / g.tail_split;  // the host made every part non-empty
		ptile = nwhole + (int)blockIdx.x / g.tail_split;
		pkb = ((int)blockIdx.x % g.tail_split) * per;
		pke = min(g.nk, pkb + per);
	}
	if (nmain == 0) return;  // (a K-split tail exists only behind whole rounds: every workgroup then has whole tiles)

	const __amdgpu_buffer_rsrc_t sa = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.A), 0, g.a_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t sb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.B), 0, g.b_bytes, 0x00020000);

	// staging: thread fills LDS rows i*64 + (tid>>3), slot tid&7 (i = 0..3) of each operand tile; source chunk = slot ^ (row & 7)
	const int srow = tid >> 3;
	const unsigned gch_bytes = (unsigned)(((tid & 7) ^ (srow & 7)) * 16);
	// B: LDS row i*64 + nt*16 + j holds global column i*64 + (nt>>1)*32 + (j>>2)*8 + (nt&1)*4 + (j&3) of the tile (see store_tile's column ownership)
	const int bperm = natural_b<EPI, NTW>() ? srow : ((srow >> 5) & 1) * 32 + ((srow & 15) >> 2) * 8 + ((srow >> 4) & 1) * 4 + (srow & 3);
	unsigned va[4];    // byte offsets of this thread's chunks for the tile being fetched (K offset added per K-tile)
	unsigned vb[4];    // NTW used (a dependent-size `vb[NTW]` captured by the lambdas below makes hipcc silently drop the kernel's host stub)
	auto set_tile = [&](int m0, int n0) {
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int ra = m0 + i * 64 + srow;
			va[i] = ra < g.M ? ((unsigned)ra * (unsigned)g.lda) * 2u + gch_bytes : OOB2;
		}
#pragma unroll
		for (int i = 0; i < NTW; ++i) {
			const int rb = n0 + i * 64 + bperm;
			vb[i] = rb < g.N ? ((unsigned)rb * (unsigned)g.ldb) * 2u + gch_bytes : OOB2;
		}
	};
	auto stage = [&](int buf, int kt) {
		char* base = smem + buf * BUF_BYTES + w * 1024;
		const unsigned kb = (unsigned)kt * (TK * 2);
#pragma unroll
		for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(sa, (lds_ptr_t)(base + i * 8192), 16, va[i] + kb, 0, 0, 0);
#pragma unroll
		for (int i = 0; i < NTW; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(sb, (lds_ptr_t)(base + OP_BYTES + i * 8192), 16, vb[i] + kb, 0, 0, 0);
	};

	// fragment reads: lane (fr, fq) reads row base + fr, k-chunk ks*4 + fq (swizzled by row & 7 = fr & 7)
	const int sw0 = ((0 * 4 + fq) ^ (fr & 7)) * 16, sw1 = ((1 * 4 + fq) ^ (fr & 7)) * 16;
	const int a_off = (wr * 128 + fr) * 128, b_off = OP_BYTES + (wc * (16 * NTW) + fr) * 128;

	f32x4 acc[8][NTW];
	auto zero_acc = [&]() {
#pragma unroll
		for (int i = 0; i < 8; ++i)
#pragma unroll
			for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
	};
	auto compute = [&](int buf) {
		const char* l = smem + buf * BUF_BYTES;
		bf16x8 fb[2][NTW];
#pragma unroll
		for (int j = 0; j < NTW; ++j) {
			fb[0][j] = *reinterpret_cast<const bf16x8*>(l + b_off + j * 2048 + sw0);
			fb[1][j] = *reinterpret_cast<const bf16x8*>(l + b_off + j * 2048 + sw1);
		}
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			bf16x8 fa[2][4];
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				fa[0][i] = *reinterpret_cast<const bf16x8*>(l + a_off + (h * 4 + i) * 2048 + sw0);
				fa[1][i] = *reinterpret_cast<const bf16x8*>(l + a_off + (h * 4 + i) * 2048 + sw1);
			}
#pragma unroll
			for (int ks = 0; ks < 2; ++ks)
#pragma unroll
				for (int i = 0; i < 4; ++i)
#pragma unroll
					for (int j = 0; j < NTW; ++j) acc[h * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], fa[ks][i], acc[h * 4 + i][j], 0, 0, 0);
		}
	};

	int tm, tn;
	tile_coords(g, xbeg + slot, tm, tn);
	int m0 = tm * TM, n0 = tn * TN;
	set_tile(m0, n0);
	stage(0, 0);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (g.nk > 1) stage(1, 1);

	// Invariant at the top of an output tile: buf[cur] holds its K-tile 0 (landed, barrier passed) and, if nk > 1, K-tile 1 is already
	// requested into buf[cur ^ 1].  One barrier per K-tile; the LDS-DMA of K-tile k+1 flies while K-tile k is multiplied.
	int cur = 0, pend = 0;
	for (int t = slot; t < xcnt; t += nslots) {
		const bool has_next = t + nslots < xcnt;
		int nm0 = 0, nn0 = 0;
		if (has_next) {
			tile_coords(g, xbeg + t + nslots, tm, tn);
			nm0 = tm * TM;
			nn0 = tn * TN;
		}
		zero_acc();
		const int tix = (t - slot) / nslots;
		auto stamp = [&](int ev) {
			if (g.trace && tid == 0 && tix < 32) g.trace[((size_t)blockIdx.x * 32 + tix) * 4 + ev] = wall_clock64();
		};
		stamp(0);
		for (int kt = 0; kt < g.nk; ++kt) {
			if (kt + 1 < g.nk) {
				if (kt > 0) stage(cur ^ 1, kt + 1);
			} else if (has_next) {  // last K-tile: request the first K-tile of the next output tile
				set_tile(nm0, nn0);
				stage(cur ^ 1, 0);
			}
			compute(cur);
			// this wave's LDS-DMA has landed ...  (first K-tile after an interior epilogue: K-tile 1 was requested BEFORE the epilogue's stores, the
			// counter retires in issue order, so the last `pend` operations -- those stores -- may stay in flight under the next K-tile)
			if (kt == 0 && pend == 16 && g.nk > 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
			else if (kt == 0 && pend == 12 && g.nk > 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
			else if (kt == 0 && pend == 8 && g.nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
			else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();                      // ... and so has everybody else's; all reads of buf[cur] are done
			asm volatile("" ::: "memory");
			cur ^= 1;
			if (kt == 0) stamp(1);
		}
		stamp(2);
		// the next tile's second K-tile goes out before this tile's stores, which then drain behind the next tile's first MFMAs
		if (has_next && g.nk > 1) stage(cur ^ 1, 1);
		pend = store_tile<EPI, NTW>(g, m0, n0, wr, wc, fr, fq, acc, smem + 2 * BUF_BYTES + w * 4096);
		stamp(3);
		m0 = nm0;
		n0 = nn0;
	}

	// K-split tail: this workgroup's K range of one of the tiles behind the last whole round, as a separate cold-started pass (folding it into the
	// tile loop above as one more item cost 31 VGPRs -- spills in the K loop, QKV 362 -> 413 us); the partial sums leave as they are, 1 KiB per
	// instruction, for gemm256_tail_kernel.
	if (ptile >= 0) {
		tile_coords(g, ptile, tm, tn);
		set_tile(tm * TM, tn * TN);
		stage(0, pkb);  // every wave is past the last barrier of the loop above: both operand buffers are free
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		zero_acc();
		int pc = 0;
		for (int kt = pkb; kt < pke; ++kt) {
			if (kt + 1 < pke) stage(pc ^ 1, kt + 1);
			compute(pc);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
			asm volatile("" ::: "memory");
			pc ^= 1;
		}
		float* wp = g.ws + ((size_t)blockIdx.x * 8 + w) * (8 * NTW * 64 * 4) + lane * 4;
#pragma unroll
		for (int mt = 0; mt < 8; ++mt)
#pragma unroll
			for (int j = 0; j < NTW; ++j) __builtin_nontemporal_store(acc[mt][j], reinterpret_cast<f32x4*>(wp + (mt * NTW + j) * 256));
	}
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// gemm256p_kernel: the 256 x 256 tile of gemm256_kernel<EPI, 4> on the 8-phase schedule of cdna_hip_programming.md section 5 (T3 + T4 + T5), round 3.
//
// Same LDS image, fragment addresses, MFMA order per accumulator, tile sequence, epilogues and K-split tail as gemm256_kernel -- bit-identical results
// (tests/test_gpu_gemm.py) -- but the K loop no longer ends every K-tile by draining the vector-memory queue behind one barrier:
//   * a K-tile is four PHASES, one per quadrant of the wave's 128 x 64 sub-tile (64 rows x 32 columns x K 64 = 16 MFMAs): a LOAD segment (the quadrant's new
//     fragments: A rows 0-63 + B columns 0-31 | B columns 32-63 | A rows 64-127 | nothing, the first B fragments are still in registers; one half-tile of LDS-DMA
//     for a LATER K-tile; at three of the four phases a counted vmcnt) and a COMPUTE segment (lgkmcnt(0), 16 MFMAs under s_setprio 1), a raw s_barrier after each;
//   * waves 4-7, the SIMD partners of waves 0-3, run ONE BARRIER BEHIND: one wave of a SIMD multiplies while the other reads and stages;
//   * operands are staged as HALF-TILES of 16 KiB (two 1 KiB pieces per wave) in the order the quadrants need them -- A rows {0-63, 128-191} (what the eight
//     waves' first row halves read), B columns 0-31 of every wave column, B columns 32-63, A rows {64-127, 192-255} -- five to seven half-tiles ahead of the phase
//     that runs, across output tiles (the stream simply continues with the next tile's first K-tiles), four or five half-tiles in flight behind every wait.
//     Round 4: phase 0 (12 ds_read_b128 per lane) stages nothing and phase 3 (no reads) stages both B half-tiles: -2 ... -3.4 % on every shape (tools/lib_ab.sh);
//   * the epilogue's stores are younger than everything staged before them, so the waits of the next tile's first K-tile leave them in flight as well.
// Hazard rules as wgrad256p_kernel (wgrad.hip): read >= 1 phase after the wait that retires a half-tile, restage >= 2 phases after its last read.
// Late round 4: the 256-row tiles run a FOUR-phase form of this loop (ktile4 below: two phases of 32 MFMAs per wave group and K-tile, lgkmcnt(0) in front of the opening
// barrier, restage one phase after the last read) -- same buffers, half-tiles, fragments and MFMA order, bit-identical; ViT-B/32 74.4 k -> 76.0 k images/s, ViT-L/14 5 740 ->
// 5 870, the training step 6.584 -> 6.559 ms (tools/lib_ab_towers.sh against a build with -DGEMM256P_FOUR_PHASE=0, three alternating rounds).  Generic in the tile height (the
// 128- / 192-row tiles run it too and stay slower than 256 rows: off).  The 8-phase form stays for two-build A/Bs and the per-phase diagnostic (-DGEMM256_DIAG_PHASES=1 needs
// -DGEMM256P_FOUR_PHASE=0).
// ---------------------------------------------------------------------------------------------------------------------------------------------------
#ifndef GEMM256P_FOUR_PHASE
#define GEMM256P_FOUR_PHASE 1  // 256-row tiles: two phases of 32 MFMAs per wave group and K-tile instead of four of 16 (late round 4; 0 = the 8-phase K loop, for two-build A/Bs)
#endif
// Diagnostic build -DGEMM256_DIAG_JITTER=1 (tools/wgrad_diag.sh; never the shipped library): every wave draws pseudo-random s_sleep delays (0 / 64 / 256 / ~1000 cycles) at the
// segment boundaries of the four-phase K loop -- before its reads, before its staging, before its counted wait, after the opening barrier, before the closing one.  The
// schedule's claim is that its result depends on barrier / wait counts only, never on when a wave gets where: such a build must stay bit-identical (wgrad.hip has the same).
#ifndef GEMM256_DIAG_JITTER
#define GEMM256_DIAG_JITTER 0
#endif
__device__ __forceinline__ void diag_jitter256(unsigned& state) {
#if GEMM256_DIAG_JITTER
	state = __builtin_amdgcn_readfirstlane(state * 1664525u + 1013904223u);
	const unsigned r = state >> 27;
	if (r == 0) __builtin_amdgcn_s_sleep(15);
	else if (r < 3) __builtin_amdgcn_s_sleep(4);
	else if (r < 7) __builtin_amdgcn_s_sleep(1);
#else
	(void)state;
#endif
}
template <int N> __device__ __forceinline__ void vm_wait_imm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// at most n vector-memory operations of this wave stay outstanding (n counts what was issued BEHIND the piece that must have landed; fewer is always safe)
__device__ __forceinline__ void vm_wait_dyn(int n) {
	if (n >= 24) vm_wait_imm<24>();
	else if (n >= 16) vm_wait_imm<16>();
	else if (n >= 12) vm_wait_imm<12>();
	else if (n >= 10) vm_wait_imm<10>();
	else if (n >= 8) vm_wait_imm<8>();
	else if (n >= 6) vm_wait_imm<6>();
	else if (n >= 4) vm_wait_imm<4>();
	else if (n >= 3) vm_wait_imm<3>();
	else if (n >= 2) vm_wait_imm<2>();
	else vm_wait_imm<0>();
}

// the steady-state wait: the four youngest half-tiles (8 pieces) stay in flight, plus the `bonus` stores of the previous tile's epilogue when they are younger still
template <int PIECES>
__device__ __forceinline__ void vm_wait_steady(int bonus) {
	if (bonus >= 24) vm_wait_imm<PIECES + 24>();
	else if (bonus >= 16) vm_wait_imm<PIECES + 16>();
	else if (bonus >= 8) vm_wait_imm<PIECES + 8>();
	else vm_wait_imm<PIECES>();
}

// MT = MFMA row tiles per wave: 8 -> 256 x 256 output tiles (wave sub-tile 128 x 64, 16 MFMAs per phase); 4 -> 128 x 256 tiles (round 4: wave sub-tile 64 x 64, 8 MFMAs per
// phase, A half-tiles of 8 KiB = ONE piece per wave): the same phases, hazards and waits with a = MT / 4 pieces per A half-tile and 2 per B half-tile -- four consecutive
// half-tiles are 2 a + 4 pieces (8 / 6), which is what every steady wait leaves in flight.  For problems whose 256-row tiles fill less than a round of the chip (the towers'
// fc2 / out-projection at batch 256: 150 tiles on 256 CUs): twice the tiles at half the work each, the tiles behind the first round cut along K.
template <int EPI, int MT = 8>
__global__ __launch_bounds__(NT2) void gemm256p_kernel(const Gemm256Args gin) {
	constexpr int NTW = 4, TN = 256, TMR = MT * 32, WROWS = MT * 16, OPA = TMR * TK * 2, BUF_BYTES = OPA + 256 * TK * 2;
	// The two A half-tiles of a wave row: rows 0 .. HR0-1 and HR0 .. WROWS-1 of its WROWS rows.  MT = 8 / 4: equal halves (64 / 32 rows).  MT = 6 (192-row tiles, late round 4):
	// 64 + 32 rows -- two pieces and one piece per wave, four and two row tiles: phases 0 / 1 carry 16 MFMAs, phases 2 / 3 eight -- so that every half-tile is still a whole
	// number of 1 KiB pieces per wave and every wait count a constant of the schedule.
	constexpr int HR0 = MT == 6 ? 64 : WROWS / 2, HR1 = WROWS - HR0;
	constexpr int PA0 = HR0 / 32, PA1 = HR1 / 32, RT0 = HR0 / 16, RT1 = HR1 / 16;  // LDS-DMA pieces per wave / MFMA row tiles of the two A halves
	constexpr int PA = PA1, STEADY_PIECES = PA0 + PA1 + 4;  // PA: the A half staged LAST in a K-tile (the tail waits count it); pieces of four consecutive half-tiles
	Gemm256Args g = gin;
	if (g.ep.row_limit) {
		const int lim = *g.ep.row_limit - g.row_base;
		g.M = lim < g.M ? (lim > 0 ? lim : 0) : g.M;
		g.tiles_m = (g.M + TMR - 1) / TMR;
		if (g.tail_dyn) plan_tail(g);
	}
	extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][A tile | B tile] 128 KiB + 8 x 4 KiB epilogue staging
	const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wr = w >> 2, wc = w & 3, fr = lane & 15, fq = lane >> 4;
	unsigned jit = 0;  // (GEMM256_DIAG_JITTER builds only)
#if GEMM256_DIAG_JITTER
	jit = __builtin_amdgcn_readfirstlane((unsigned)__builtin_amdgcn_s_memtime() * 2654435761u + (blockIdx.x * 8u + (unsigned)w) * 40503u);
#endif

	const int ntiles = g.tiles_m * g.tiles_n;
	const bool split = g.tail_split > 1 && (!g.ep.row_limit || g.tail_dyn);
	const int nwhole = split ? g.tail_first : ntiles;
	const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
	const int q8 = nwhole >> 3, rm = nwhole & 7;
	const int xbeg = xcd < rm ? xcd * (q8 + 1) : rm * (q8 + 1) + (xcd - rm) * q8;
	const int xcnt = q8 + (xcd < rm ? 1 : 0);
	const int nmain = slot < xcnt ? (xcnt - slot + nslots - 1) / nslots : 0;
	int ptile = -1, pkb = 0, pke = 0;
	if (split && (int)blockIdx.x < (ntiles - nwhole) * g.tail_split) {
		const int per = (g.nk + g.tail_split - 1) / g.tail_split;
		ptile = nwhole + (int)blockIdx.x / g.tail_split;
		pkb = ((int)blockIdx.x % g.tail_split) * per;
		pke = min(g.nk, pkb + per);
	}
	if (nmain == 0) return;

	const __amdgpu_buffer_rsrc_t sa = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.A), 0, g.a_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t sb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.B), 0, g.b_bytes, 0x00020000);

	// ---- staging.  A piece = 8 LDS rows x 128 B = one DMA instruction; lane -> (row prow of the piece, slot lane & 7), source chunk slot ^ (row & 7) = slot ^ prow.
	// A half ah: LDS rows (w >> 2) * 128 + ah * 64 + (2 (w & 3) + i) * 8 + prow  (= the tile's rows: A is stored in natural order)
	// B half bh: LDS rows (w >> 1) * 64 + bh * 32 + (2 (w & 1) + i) * 8 + prow, holding -- permuted image, see gemm256_kernel -- the tile's column
	//            (w >> 1) * 64 + bh * 32 + 16 i + (prow >> 2) * 8 + (w & 1) * 4 + (prow & 3); natural image: the column of the same number as the row.
	// Rows / columns beyond M / N have offsets beyond the descriptor's size (row * ld * 2 >= rows * ld * 2) and read as zeros.
	constexpr bool NAT = natural_b<EPI, 4>();
	const int prow = lane >> 3;
	const unsigned gch = (unsigned)(((lane & 7) ^ prow) * 16);
	const int arow0 = (w >> 2) * WROWS + (w & 3) * (8 * PA0) + prow, arow1 = (w >> 2) * WROWS + HR0 + (w & 3) * (8 * PA1) + prow;  // this lane's first row of A half 0 / 1
	const int bcol0 = NAT ? (w >> 1) * 64 + (w & 1) * 16 + prow : (w >> 1) * 64 + (prow >> 2) * 8 + (w & 1) * 4 + (prow & 3);
	const unsigned a_i = 8u * (unsigned)g.lda * 2u, a_h = (unsigned)(arow1 - arow0) * (unsigned)g.lda * 2u;
	const unsigned b_i = (NAT ? 8u : 16u) * (unsigned)g.ldb * 2u, b_h = 32u * (unsigned)g.ldb * 2u;
	const unsigned lds_a0 = (unsigned)(((w >> 2) * WROWS + (w & 3) * (8 * PA0)) * 128), lds_a1 = (unsigned)(((w >> 2) * WROWS + HR0 + (w & 3) * (8 * PA1)) * 128);
	const unsigned lds_b = (unsigned)(OPA + ((w >> 1) * 64 + (w & 1) * 16) * 128);
	auto tile_base = [&](int m0, int n0, unsigned& ba, unsigned& bb) {
		ba = (unsigned)(m0 + arow0) * (unsigned)g.lda * 2u + gch;
		bb = (unsigned)(n0 + bcol0) * (unsigned)g.ldb * 2u + gch;
	};
	// half-tile q of K-tile kt of the tile whose bases are (ba, bb) into buffer buf: q = 0 A rows 0-63 (+128), 1 B columns 0-31, 2 B columns 32-63, 3 A rows 64-127 (+128)
	auto stage_half = [&](int buf, unsigned ba, unsigned bb, int kt, auto qc) {
		constexpr int q = decltype(qc)::value;
#ifdef GEMM256_DIAG_NO_DMA  // diagnostic build (timing only, results are garbage): 1 = no operand is staged at all, 2 = A only, 3 = B only -- what the K loop costs without (part of) its memory side
		if constexpr (GEMM256_DIAG_NO_DMA == 1 || (GEMM256_DIAG_NO_DMA == 2 && (q == 1 || q == 2)) || (GEMM256_DIAG_NO_DMA == 3 && (q == 0 || q == 3))) return;
#endif
		char* base = smem + buf * BUF_BYTES;
		const unsigned kof = (unsigned)kt * (TK * 2);
		if constexpr (q == 0 || q == 3) {
			constexpr int ah = q == 3;
#pragma unroll
			for (int i = 0; i < (ah ? PA1 : PA0); ++i)
				__builtin_amdgcn_raw_ptr_buffer_load_lds(sa, (lds_ptr_t)(base + (ah ? lds_a1 : lds_a0) + i * 8 * 128), 16, ba + (ah ? a_h : 0u) + (i ? a_i : 0u) + kof, 0, 0, 0);
		} else {
			constexpr int bh = q == 2;
#pragma unroll
			for (int i = 0; i < 2; ++i)
				__builtin_amdgcn_raw_ptr_buffer_load_lds(sb, (lds_ptr_t)(base + lds_b + (bh * 32 + i * 8) * 128), 16, bb + (bh ? b_h : 0u) + (i ? b_i : 0u) + kof, 0, 0, 0);
		}
	};

	// ---- fragments (gemm256_kernel's addresses): lane (fr, fq) reads row base + fr, k-chunk ks * 4 + fq, swizzled by row & 7 = fr & 7
	const int sw0 = ((0 * 4 + fq) ^ (fr & 7)) * 16, sw1 = ((1 * 4 + fq) ^ (fr & 7)) * 16;
	const int a_off = (wr * WROWS + fr) * 128, b_off = OPA + (wc * 64 + fr) * 128;
	f32x4 acc[MT][4];
	auto zero_acc = [&]() {
#pragma unroll
		for (int i = 0; i < MT; ++i)
#pragma unroll
			for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
	};
	bf16x8 fa[2][RT0], fb0[2][2], fb1[2][2];
	auto read_a = [&](const char* l, auto ahc) {
		constexpr int ah = decltype(ahc)::value;
#pragma unroll
		for (int i = 0; i < (ah ? RT1 : RT0); ++i) {
			fa[0][i] = *reinterpret_cast<const bf16x8*>(l + a_off + ((ah ? RT0 : 0) + i) * 2048 + sw0);
			fa[1][i] = *reinterpret_cast<const bf16x8*>(l + a_off + ((ah ? RT0 : 0) + i) * 2048 + sw1);
		}
	};
	auto read_b = [&](const char* l, bf16x8 (&fb)[2][2], auto bhc) {
		constexpr int bh = decltype(bhc)::value;
#pragma unroll
		for (int j = 0; j < 2; ++j) {
			fb[0][j] = *reinterpret_cast<const bf16x8*>(l + b_off + (bh * 2 + j) * 2048 + sw0);
			fb[1][j] = *reinterpret_cast<const bf16x8*>(l + b_off + (bh * 2 + j) * 2048 + sw1);
		}
	};
	auto mul = [&](const bf16x8 (&fb)[2][2], auto ahc, auto bhc) {
		constexpr int ah = decltype(ahc)::value, bh = decltype(bhc)::value;
#pragma unroll
		for (int ks = 0; ks < 2; ++ks)
#pragma unroll
			for (int i = 0; i < (ah ? RT1 : RT0); ++i)
#pragma unroll
				for (int j = 0; j < 2; ++j) acc[(ah ? RT0 : 0) + i][bh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], fa[ks][i], acc[(ah ? RT0 : 0) + i][bh * 2 + j], 0, 0, 0);
	};
	auto bar = [&]() {
		__builtin_amdgcn_sched_barrier(0);
		__builtin_amdgcn_s_barrier();
		__builtin_amdgcn_sched_barrier(0);
	};
	using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>; using C2 = std::integral_constant<int, 2>; using C3 = std::integral_constant<int, 3>;

	// ---- the stream of K-tiles: this workgroup's tiles one after the other, nk K-tiles each; global K-tile number gk -> buffer gk & 1
	int tm, tn;
	tile_coords(g, xbeg + slot, tm, tn);
	int m0 = tm * TMR, n0 = tn * TN;
	unsigned cba, cbb, nba = 0, nbb = 0;  // staging bases of the current tile and of the next one
	tile_base(m0, n0, cba, cbb);
	const int nk = g.nk;  // >= 2 (host)

	// prologue: K-tile 0 whole and the first three half-tiles of K-tile 1 (what the phases of K-tiles -2, -1 would have staged); the half-tiles phase 0 reads have landed
	stage_half(0, cba, cbb, 0, C0{}); stage_half(0, cba, cbb, 0, C1{}); stage_half(0, cba, cbb, 0, C2{}); stage_half(0, cba, cbb, 0, C3{});
	stage_half(1, cba, cbb, 1, C0{}); stage_half(1, cba, cbb, 1, C1{});
	stage_half(1, cba, cbb, 1, C2{});
	constexpr bool FOUR = GEMM256P_FOUR_PHASE != 0;
	vm_wait_imm<FOUR ? STEADY_PIECES : STEADY_PIECES + 2>();  // 8-phase: A0 / B0 of K-tile 0 have landed (B1 is waited for in phase 0); four-phase: B1 as well -- behind it A1(0), A0 / B0 / B1 (1)
	bar();
	if (wr == 1) bar();  // waves 4-7 run one barrier behind their SIMD partners from here on

	int buf = 0, pend = 0;
	for (int t = slot; t < xcnt; t += nslots) {
		const bool has_next = t + nslots < xcnt;
		int nm0 = 0, nn0 = 0;
		if (has_next) {
			tile_coords(g, xbeg + t + nslots, tm, tn);
			nm0 = tm * TMR;
			nn0 = tn * TN;
			tile_base(nm0, nn0, nba, nbb);
		}
		zero_acc();
		const int tix = (t - slot) / nslots;
		auto stamp = [&](int ev) {
			if (g.trace && tid == 0 && tix < 32) g.trace[((size_t)blockIdx.x * 32 + tix) * 4 + ev] = wall_clock64();
		};
		stamp(0);
		// One K-tile.  STEADY: two more K-tiles follow in the stream (every half-tile the phases stage exists): the waits leave 8 (phase 3: 10) pieces + the store bonus in flight.
		// The two K-tiles at the very end of the stream (rem = 2, 1) take the general form, peeled behind the steady loop of the last tile.
		auto ktile = [&](int kt, auto steady_c, auto diag_c) __attribute__((always_inline)) {
			constexpr bool STEADY = decltype(steady_c)::value;
			// diagnostic build (-DGEMM256_DIAG_PHASES=1, tools/gemm_phases.py): ONE K-tile of the first tile takes 16 shader-clock stamps per wave group -- per phase: LOAD
			// segment starts / ends, opening barrier passed, MFMAs issued -- via s_memtime (returns under the lgkmcnt(0) every phase has anyway)
			constexpr bool DG = decltype(diag_c)::value;
			unsigned long long ts[16];
			auto ps = [&](int idx) __attribute__((always_inline)) {
				if constexpr (DG) {
					__builtin_amdgcn_sched_barrier(0);
					ts[idx] = __builtin_amdgcn_s_memtime();
					__builtin_amdgcn_sched_barrier(0);
				}
			};
			auto compute = [&](const bf16x8 (&fb)[2][2], auto ahc, auto bhc, auto phc) __attribute__((always_inline)) {
				constexpr int ph = decltype(phc)::value;
				ps(4 * ph + 1);
				bar();
				asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
				ps(4 * ph + 2);
				__builtin_amdgcn_sched_barrier(0);
				__builtin_amdgcn_s_setprio(1);
				mul(fb, ahc, bhc);
				__builtin_amdgcn_s_setprio(0);
				ps(4 * ph + 3);
				bar();
			};
			// the K-tiles the phases of this one stage for: kt + 1 (phase 1) and kt + 2 (phases 2, 3), in this tile or at the head of the next
			const int rem = has_next ? 1 << 20 : nk - kt;  // K-tiles left in the stream including this one
			const bool in1 = kt + 1 < nk, in2 = kt + 2 < nk;
			const unsigned ba1 = in1 ? cba : nba, bb1 = in1 ? cbb : nbb, ba2 = in2 ? cba : nba, bb2 = in2 ? cbb : nbb;
			const int k1 = in1 ? kt + 1 : kt + 1 - nk, k2 = in2 ? kt + 2 : kt + 2 - nk;
			const char* l = smem + buf * BUF_BYTES;
			const int bonus = kt == 0 ? pend : 0;  // the previous tile's stores were issued behind everything this K-tile waits for
			// Round 4: B columns 32-63 are staged at phase 3 TWO K-tiles ahead (next to B columns 0-31), not at phase 0 one K-tile ahead: phase 0's LOAD segment carries
			// the most LDS reads (A rows 0-63 + B columns 0-31, 12 ds_read_b128 per lane) and phase 3's none, and a LOAD segment longer than the other group's 16 MFMAs
			// stretches the interval.  Stream order per K-tile k: A0(k) [phase 2 of k-2], B0(k), B1(k) [phase 3 of k-2], A1(k) [phase 1 of k-1].
			const int bonus1 = kt <= 1 ? pend : 0;  // (phase 0 of the tile's SECOND K-tile waits for a half-tile staged before the previous tile's stores as well)
			// phase 0: quadrant (A rows 0-63, B columns 0-31); B columns 32-63 of kt must have landed for phase 1: behind it A1(kt), A0 / B0 / B1 (kt + 1)
			ps(0);
			read_b(l, fb0, C0{});
			read_a(l, C0{});
			if constexpr (STEADY) vm_wait_steady<STEADY_PIECES>(bonus1);
			else vm_wait_dyn((rem > 1 ? STEADY_PIECES : PA) + bonus1);
			compute(fb0, C0{}, C0{}, C0{});
			// phase 1: (A 0-63, B 32-63); stages A rows 64-127 of kt + 1; A rows 64-127 of kt must have landed for phase 2: behind it A0 / B0 / B1 / A1 (kt + 1)
			ps(4);
			read_b(l, fb1, C1{});
			if (STEADY || rem > 1) stage_half(buf ^ 1, ba1, bb1, k1, C3{});
			if constexpr (STEADY) vm_wait_steady<STEADY_PIECES>(bonus);
			else vm_wait_dyn((rem > 1 ? STEADY_PIECES : 0) + bonus);
			compute(fb1, C0{}, C1{}, C1{});
			// phase 2: (A 64-127, B 32-63); stages A rows 0-63 of kt + 2 into THIS buffer (last read at phase 0)
			ps(8);
			read_a(l, C1{});
			if (STEADY || rem > 2) stage_half(buf, ba2, bb2, k2, C0{});
			compute(fb1, C1{}, C1{}, C2{});
			// phase 3: (A 64-127, B 0-31: the fragments of phase 0); stages BOTH B half-tiles of kt + 2 (columns 32-63 of this buffer were last read at phase 1);
			// A0 and B0 of kt + 1 must have landed for its phase 0: behind B0(kt + 1) are B1 / A1 (kt + 1) and A0 / B0 / B1 (kt + 2)
			ps(12);
			if (STEADY || rem > 2) {
				stage_half(buf, ba2, bb2, k2, C1{});
				stage_half(buf, ba2, bb2, k2, C2{});
			}
			if constexpr (STEADY) vm_wait_steady<STEADY_PIECES + 2>(bonus);
			else vm_wait_dyn((rem > 2 ? STEADY_PIECES + 2 : (rem > 1 ? PA + 2 : 0)) + bonus);
			compute(fb0, C1{}, C0{}, C3{});
			if constexpr (DG) {
				if (g.trace && lane == 0 && (w & 3) == 0) {
#pragma unroll
					for (int e = 0; e < 16; ++e) g.trace[((size_t)blockIdx.x * 32 + 16) * 4 + (w >> 2) * 16 + e] = ts[e];
				}
			}
			buf ^= 1;
		};
		// ---- Four-phase K-tile (late round 4).  The phase diagnostic above showed what an interval between two barriers costs whatever the MFMA block in it: LOAD issue + LDS latency
		// + hand-over, ~300 cycles -- the 16-MFMA block of the 8-phase loop (256 + ~45 cycles) only just covers it, and 128- / 192-row tiles with 8-MFMA blocks took the SAME
		// time per K-tile.  So: fewer, longer intervals.  A wave group's K-tile is TWO phases of 32 MFMAs -- A rows 0-63 against all 64 columns, then rows 64-127 -- with the same
		// fragment registers (B is read once per K-tile and kept, A re-read per phase), the same two operand buffers and the same half-tiles:
		//   P0: reads B0, B1, A0 of kt (16 ds_read_b128); stages A1 of kt + 1 (other buffer: last read at P1 of kt - 1); waits for A1 of kt
		//   P1: reads A1 of kt (8);                       stages A0, B0, B1 of kt + 2 (THIS buffer: last read at P0 of kt); waits for A0 / B0 / B1 of kt + 1
		// Stream order per wave: ... A1(kt+1) | A0 B0 B1(kt+2) | A1(kt+2) | A0 B0 B1(kt+3) ...: every wait leaves 8 pieces (four half-tiles) in flight, ~6 half-intervals of
		// flight time.  WAR at a distance of ONE phase: lgkmcnt(0) sits BEFORE the opening barrier here (the wave waits there for the other group's MFMAs anyway), so when any
		// wave passes a barrier every wave's reads of the segment in front of it are complete; RAW as before (a wait, then a barrier, then the read).  Same MFMA order per
		// accumulator as the 8-phase loop: bit-identical (tests/test_gpu_gemm.py).
		auto compute4 = [&](auto ahc) __attribute__((always_inline)) {
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			bar();
			diag_jitter256(jit);
			__builtin_amdgcn_s_setprio(1);
			mul(fb0, ahc, C0{});
			mul(fb1, ahc, C1{});
			__builtin_amdgcn_s_setprio(0);
			diag_jitter256(jit);
			bar();
		};
		auto ktile4 = [&](int kt, auto steady_c) __attribute__((always_inline)) {
			constexpr bool STEADY = decltype(steady_c)::value;
			const int rem = has_next ? 1 << 20 : nk - kt;  // K-tiles left in the stream including this one
			const bool in1 = kt + 1 < nk, in2 = kt + 2 < nk;
			const unsigned ba1 = in1 ? cba : nba, bb1 = in1 ? cbb : nbb, ba2 = in2 ? cba : nba, bb2 = in2 ? cbb : nbb;
			const int k1 = in1 ? kt + 1 : kt + 1 - nk, k2 = in2 ? kt + 2 : kt + 2 - nk;
			const char* l = smem + buf * BUF_BYTES;
			const int bonus = kt == 0 ? pend : 0;  // (both waits of a tile's first K-tile retire half-tiles staged before the previous tile's stores; from its second K-tile on none does)
			diag_jitter256(jit);
			read_b(l, fb0, C0{});
			read_b(l, fb1, C1{});
			read_a(l, C0{});
			diag_jitter256(jit);
			if (STEADY || rem > 1) stage_half(buf ^ 1, ba1, bb1, k1, C3{});
			diag_jitter256(jit);
#ifdef GEMM256_DIAG_VMCNT  // diagnostic build: STRICTER steady waits (fewer LDS-DMA pieces left in flight; always safe) -- how much does the K loop depend on its prefetch depth?
			if constexpr (STEADY) vm_wait_imm<GEMM256_DIAG_VMCNT>();
#else
			if constexpr (STEADY) vm_wait_steady<STEADY_PIECES>(bonus);
#endif
			else vm_wait_dyn((rem > 1 ? STEADY_PIECES : 0) + bonus);
			compute4(C0{});
			diag_jitter256(jit);
			read_a(l, C1{});
			diag_jitter256(jit);
			if (STEADY || rem > 2) {
				stage_half(buf, ba2, bb2, k2, C0{});
				stage_half(buf, ba2, bb2, k2, C1{});
				stage_half(buf, ba2, bb2, k2, C2{});
			}
#ifdef GEMM256_DIAG_VMCNT
			if constexpr (STEADY) vm_wait_imm<GEMM256_DIAG_VMCNT>();
#else
			if constexpr (STEADY) vm_wait_steady<STEADY_PIECES>(bonus);
#endif
			else vm_wait_dyn((rem > 2 ? STEADY_PIECES : (rem > 1 ? PA : 0)) + bonus);
			compute4(C1{});
			buf ^= 1;
		};
		auto ktile_any = [&](int kt, auto steady_c, auto diag_c) __attribute__((always_inline)) {
			if constexpr (FOUR) ktile4(kt, steady_c);
			else ktile(kt, steady_c, diag_c);
		};
		const int ksteady = has_next ? nk : nk - 2;
		if (ksteady > 0) {
			ktile_any(0, std::true_type{}, std::false_type{});  // (the first K-tile on its own: its waits carry the store bonus, and the timeline's stamp stays out of the steady loop)
			stamp(1);
		}
#if GEMM256_DIAG_PHASES
		for (int kt = 1; kt < ksteady; ++kt) {
			if (kt == 4 && tix == 0) ktile_any(kt, std::true_type{}, std::true_type{});
			else ktile_any(kt, std::true_type{}, std::false_type{});
		}
#else
		for (int kt = 1; kt < ksteady; ++kt) ktile_any(kt, std::true_type{}, std::false_type{});
#endif
		if (!has_next) {
			ktile_any(nk - 2, std::false_type{}, std::false_type{});
			ktile_any(nk - 1, std::false_type{}, std::false_type{});
		}
		stamp(2);
		// The epilogue runs LEVEL: staggered, the store phases of the two wave groups would follow each other (each group waits at its next barrier for the other's
		// stores to issue: 2 x 1.2 us per tile measured), level they share the CU's store path (1.8 us).  Waves 0-3 take the barrier waves 4-7 still owe ...
		if (wr == 0) bar();
		pend = store_tile<EPI, 4, MT>(g, m0, n0, wr, wc, fr, fq, acc, smem + 2 * BUF_BYTES + w * 4096);
#if GEMM256_DIAG_NO_STORES
		pend = 0;  // (no stores were issued: the next tile's waits must not count on them)
#endif
		stamp(3);
		if (has_next && wr == 1) bar();  // ... and waves 4-7 fall one barrier behind again for the next tile's K loop
		m0 = nm0; n0 = nn0;
		cba = nba; cbb = nbb;
	}

	// K-split tail (as gemm256_kernel): this workgroup's K range of one of the tiles behind the last whole round, a cold-started pass with one barrier per K-tile
	if (ptile >= 0) {
		tile_coords(g, ptile, tm, tn);
		tile_base(tm * TMR, tn * TN, cba, cbb);
		auto stage_all = [&](int b, int kt) { stage_half(b, cba, cbb, kt, C0{}); stage_half(b, cba, cbb, kt, C1{}); stage_half(b, cba, cbb, kt, C2{}); stage_half(b, cba, cbb, kt, C3{}); };
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		stage_all(0, pkb);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		zero_acc();
		int pc = 0;
		for (int kt = pkb; kt < pke; ++kt) {
			if (kt + 1 < pke) stage_all(pc ^ 1, kt + 1);
			const char* l = smem + pc * BUF_BYTES;
			read_b(l, fb0, C0{}); read_b(l, fb1, C1{});
			read_a(l, C0{});
			mul(fb0, C0{}, C0{}); mul(fb1, C0{}, C1{});
			read_a(l, C1{});
			mul(fb0, C1{}, C0{}); mul(fb1, C1{}, C1{});
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
			asm volatile("" ::: "memory");
			pc ^= 1;
		}
		float* wp = g.ws + ((size_t)blockIdx.x * 8 + w) * (MT * NTW * 64 * 4) + lane * 4;
#pragma unroll
		for (int mt = 0; mt < MT; ++mt)
#pragma unroll
			for (int j = 0; j < NTW; ++j) __builtin_nontemporal_store(acc[mt][j], reinterpret_cast<f32x4*>(wp + (mt * NTW + j) * 256));
	}
}

// Finishes the K-split tail tiles: sums the tail_split partial accumulators of every element in a fixed order (deterministic, unlike atomics) and runs
// the ordinary per-element epilogue.  One thread per accumulator quad; grid = tail tiles x 64 workgroups of 256 threads.
template <int EPI, int MT = 8>
__global__ __launch_bounds__(256) void gemm256_tail_kernel(const Gemm256Args gin) {
	constexpr int TMR = MT * 32, WROWS = MT * 16, WGS = 8 * MT;  // rows of a tile / of a wave's share; workgroups per tail tile
	Gemm256Args g = gin;
	const int r = blockIdx.x / WGS, idx = (blockIdx.x % WGS) * 256 + threadIdx.x;  // idx = ((w * MT + mt) * 4 + j) * 64 + lane
	if (g.tail_dyn) {  // device row count: the plan of gemm256_kernel, recomputed; the launch covers the largest tail there can be (64 tiles)
		const int lim = *g.ep.row_limit - g.row_base;
		g.M = lim < g.M ? (lim > 0 ? lim : 0) : g.M;
		g.tiles_m = (g.M + TM - 1) / TM;
		plan_tail(g);
		if (g.tail_split <= 1 || r >= g.tiles_m * g.tiles_n - g.tail_first) return;
	}
	const int lane = idx & 63, j = (idx >> 6) & 3, mt = (idx >> 8) & (MT - 1), w = idx / (MT * 256);
	const int wr = w >> 2, wc = w & 3, fr = lane & 15, fq = lane >> 4;
	int tm, tn;
	tile_coords(g, g.tail_first + r, tm, tn);
	const float* wp = g.ws + (size_t)r * g.tail_split * (MT * 8192) + idx * 4;
	f32x4 sum = *reinterpret_cast<const f32x4*>(wp);
	for (int sidx = 1; sidx < g.tail_split; ++sidx) {
		const f32x4 t = *reinterpret_cast<const f32x4*>(wp + (size_t)sidx * (MT * 8192));
		sum[0] += t[0]; sum[1] += t[1]; sum[2] += t[2]; sum[3] += t[3];
	}
	const int m = tm * TMR + wr * WROWS + mt * 16 + fr;
	const int n = tn * 256 + wc * 64 + (natural_b<EPI, 4>() ? j * 16 + fq * 4 : (j >> 1) * 32 + fq * 8 + (j & 1) * 4);
	const bool inside = m < g.M && n < g.N;
	float v[4] = {sum[0], sum[1], sum[2], sum[3]};
	if (inside) epilogue4<EPI>(g.ep, m, n, g.N, v);
}

// Process-wide settings (include/novic_hip.h, "Process-wide settings"): relaxed atomics -- a call reads each of them once, a concurrent setter can never tear a launch.
std::atomic<unsigned long long*> g_trace{nullptr};
// novic_persistent_cus: the DEFAULT workgroup budget of the persistent grids, for calls whose novic_epilogue_t.max_workgroups is 0; the start value can come from the
// environment (NOVIC_PERSISTENT_CUS: a node whose collectives need CUs of their own beside the backward pass)
std::atomic<int> g_ncu{[] { const char* e = getenv("NOVIC_PERSISTENT_CUS"); const int n = e ? atoi(e) : 0; return (n >= 8 && n <= 256) ? n / 8 * 8 : 256; }()};
// (Round 5 removed the switches whose losing side no call ever took -- K = 1024 tails off, write-back output stores, 128- / 192-row tiles: built in round 4, bit-identical,
// measured slower, DESIGN.md section 4 "Round 4" -- together with the kernel instantiations behind them; gemm256p_kernel stays generic in its tile height, MT = 8 is what ships.)
std::atomic<int> g_pipelined{1};  // 1: 256 x 256 tiles on gemm256p_kernel (8-phase schedule), 0: gemm256_kernel<EPI, 4> (one barrier per K-tile) -- novic_gemm256_pipeline

template <int EPI, int NTW>
void launch256(const Gemm256Args& g, int grid, hipStream_t stream) {
	constexpr int LDS = 2 * buf_bytes<NTW>() + 8 * 4096;
	static std::atomic<bool> attr_done{false};  // (hipFuncSetAttribute is idempotent: two threads racing here both set the same value)
	if (!attr_done.load(std::memory_order_acquire)) {
		(void)hipFuncSetAttribute((const void*)gemm256_kernel<EPI, NTW>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
		attr_done.store(true, std::memory_order_release);
	}
	if constexpr (NTW == 4) {
		if (g.pipelined && g.nk >= 2) {
			static std::atomic<bool> attr_p{false};
			if (!attr_p.load(std::memory_order_acquire)) {
				(void)hipFuncSetAttribute((const void*)gemm256p_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
				attr_p.store(true, std::memory_order_release);
			}
			hipLaunchKernelGGL((gemm256p_kernel<EPI>), dim3(grid), dim3(NT2), LDS, stream, g);
		} else {
			hipLaunchKernelGGL((gemm256_kernel<EPI, NTW>), dim3(grid), dim3(NT2), LDS, stream, g);
		}
	} else {
		hipLaunchKernelGGL((gemm256_kernel<EPI, NTW>), dim3(grid), dim3(NT2), LDS, stream, g);
	}
	if constexpr (NTW == 4) {
		if (g.tail_dyn) hipLaunchKernelGGL((gemm256_tail_kernel<EPI>), dim3(64 * 64), dim3(256), 0, stream, g);
		else if (g.tail_split > 1) hipLaunchKernelGGL((gemm256_tail_kernel<EPI>), dim3((g.tiles_m * g.tiles_n - g.tail_first) * 64), dim3(256), 0, stream, g);
	}
}

template <int NTW>
int launch256_epi(const Gemm256Args& g, int grid, hipStream_t stream) {
	switch (g.ep.kind) {
		case NOVIC_EPI_STORE_BF16: launch256<NOVIC_EPI_STORE_BF16, NTW>(g, grid, stream); break;
		case NOVIC_EPI_STORE_F32: launch256<NOVIC_EPI_STORE_F32, NTW>(g, grid, stream); break;
		case NOVIC_EPI_RESID_F32: launch256<NOVIC_EPI_RESID_F32, NTW>(g, grid, stream); break;
		case NOVIC_EPI_GELU_BF16: launch256<NOVIC_EPI_GELU_BF16, NTW>(g, grid, stream); break;
		case NOVIC_EPI_GELU_BWD_BF16: launch256<NOVIC_EPI_GELU_BWD_BF16, NTW>(g, grid, stream); break;
		case NOVIC_EPI_RESID_F16:
			if constexpr (NTW == 4) { launch256<NOVIC_EPI_RESID_F16, 4>(g, grid, stream); break; }
			return 1;  // (plan256 never picks the 192-wide tile for the half stream)
		default: return 1;
	}
	return 0;
}

}  // namespace

// Diagnostic (tools/gemm_timeline.py): subsequent launches of the LDS-DMA kernel stamp, per workgroup and for its first 32 tiles, the 100 MHz wall
// clock at tile start / after the first K-tile / after the K loop / after the stores are issued, into buf[256][32][4]; null switches it off.
extern "C" int novic_gemm256_pipeline(int on) {  // see include/novic_hip.h
	const int prev = g_pipelined.load(std::memory_order_relaxed);
	if (on == 0 || on == 1) g_pipelined.store(on, std::memory_order_relaxed);
	return prev;
}

extern "C" int novic_persistent_cus(int n) {  // see include/novic_hip.h
	if (n >= 8 && n <= 256) return g_ncu.exchange(n / 8 * 8, std::memory_order_relaxed);
	return g_ncu.load(std::memory_order_relaxed);
}

extern "C" int novic_gemm256_trace(unsigned long long* buf) {
	g_trace.store(buf, std::memory_order_relaxed);
	return 0;
}

// Called by novic_gemm_bf16 (gemm.hip) for K-contiguous x K-contiguous problems; returns 1 if the problem is not one this kernel takes, else 0 with
// *tile_n = the tile width used.  force: 0 = choose, 256 / 192 = that tile width whenever the kernel can run at all (benchmarks).
// The decision alone (host arithmetic, no HIP call): which tile, how many workgroups, whether and how the tiles behind the last whole round are cut along K.
static int plan256(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const novic_epilogue_t* ep, int force, Gemm256Args& g, int& tn_out, int& grid_out) {
	if (K % TK != 0 || K < TK || N % 4 != 0 || ep->kind == NOVIC_EPI_ATOMIC_F32) return 1;
	if (epilogue_is_act_variant(ep)) return 1;  // relu / tanh in place of the GELU, a bias in front of it: the 128 x 128 kernel's epilogue (gemm_epilogue.hpp)
	// the workgroup budget of THIS call (novic_epilogue_t.max_workgroups; 0: the process default) and the process-wide switches, each read once
	const int ncu = ep->max_workgroups ? (int)((ep->max_workgroups < 8 ? 8u : (ep->max_workgroups > 256 ? 256u : ep->max_workgroups)) / 8 * 8) : g_ncu.load(std::memory_order_relaxed);
	const int pipelined = g_pipelined.load(std::memory_order_relaxed);
	const uint64_t ab = (uint64_t)M * lda * 2, bb = (uint64_t)N * ldb * 2;
	if (ab >= 0x7FFFFFF0ull || bb >= 0x7FFFFFF0ull) return 1;
	g.A = (const bf16*)A; g.B = (const bf16*)B;
	g.row_base = 0;
	g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb;
	g.a_bytes = (unsigned)ab; g.b_bytes = (unsigned)bb;
	g.tiles_m = (M + TM - 1) / TM;
	// Measured on MI355X (tools/gemm_sweep.py): the 256-wide tile wins where there are several rounds of tiles per CU and >= 4 column tiles
	// ([57344 x 6912 x 512] 740 -> 520 us, [81920 x 1536 x 512] 212 -> 167 us); narrow outputs (N = 512: 2.5 rounds of 640 tiles) and few-tile
	// problems are as fast or faster on the 128^2 kernel, which also prefetches the residual operand of the RESID epilogue.
	const int t256 = g.tiles_m * ((N + 255) / 256), t192 = g.tiles_m * ((N + 191) / 192);
	int tn = 0;
	bool dyn_tail = false;
	if (force == 256 || force == 192) tn = (ep->kind == NOVIC_EPI_RESID_F16) ? 256 : force;
	else if (t256 >= 256 && (N + 255) / 256 >= 4) tn = 256;
	// Tall two-column problems with a DEVICE row count and scratch for a K-split tail -- BEFORE the general two-column rule below, which would take them without the tail
	// (it did for a while in round 3: 278 us instead of ~200) -- (the logits input gradient on the compacted rows: [36.9 k of 57.3 k x 512 x
	// 6912] = 290 of 448 tiles): the tiles behind the last whole round are cut along K by a plan every workgroup works out from the clamped tile count
	// (plan_tail), so 34 tail tiles cost a seventh of a round instead of a whole one -- 329 us on the 128^2 kernel -> ~190 us
	else if (ep->kind == NOVIC_EPI_STORE_BF16 && ep->act == NOVIC_ACT_NONE && !ep->bias && ep->row_limit && ep->splitk_ws && ((uintptr_t)ep->splitk_ws & 15) == 0 &&
	         (N + 255) / 256 >= 2 && K / TK >= 32 && t256 >= 256) { tn = 256; dyn_tail = true; }
	// Tall problems with only two tile columns (the input-gradient GEMMs of the decoder against the transposed weight shadows, N = 512): with
	// the bf16 whole-line store path [81920 x 512 x 1536] 180 -> 134 us, [.. x 512] 85 -> 55 us, [.. x 128] 41 -> 23 us against the 128^2 kernel.
	// Not the fp32-residual epilogue: its 8 bytes of HBM traffic per output element want the second resident workgroup of the 128^2 kernel.
	// (not [57344 x 512 x 6912], 448 tiles: 450-540 us against 456 us)
	else if (ep->kind == NOVIC_EPI_STORE_BF16 && (N + 255) / 256 >= 2 && t256 >= 256) tn = 256;
	// One round of 192-wide tiles that fills most of the chip, fp32 residual epilogue (ViT-B/32 at batch 256: [12800 x 768 x 3072] 96 -> 82 us,
	// [12800 x 768 x 768] 35.5 -> 33.7 us against the 128^2 kernel; with the bf16 epilogues the 192-wide tile's 8-byte stores lose).
	// (round 3: on the 8-phase K loop the 256-wide tile wins these too -- [12800 x 768 x 3072] 75.8 -> 67.9 us, [12800 x 768 x 768] 34.9 -> 30.5 us with 150 tiles
	// against 200 of the 192-wide one-barrier kernel, tools/vit_b32_gemm_ab.py -- so the 192-wide tile is only chosen when that schedule is switched off)
	else if (ep->kind == NOVIC_EPI_RESID_F32 && N % 192 == 0 && t192 >= 160 && t192 <= 256) tn = (pipelined && K / TK >= 2) ? 256 : 192;
	else if (ep->kind == NOVIC_EPI_RESID_F16 && N % 192 == 0 && t192 >= 160 && t192 <= 256) tn = 256;  // (the half stream: 256-wide tiles on either K loop)
	// Round 3: on the 8-phase K loop the 256-wide tile wins from a bit more than half a round of tiles on, whatever the epilogue and however few tile columns
	// (tools/vit_b32_gemm_ab.py ROWS WIDTH, against the 128^2 kernel): fp32-residual [50176 x 768 x 3072] 328 -> 252 us, [50176 x 768 x 768] 107 -> 93 (SigLIP B/16 at batch
	// 256: 588 tiles), [19712 x 768 x 3072] 121 -> 83, [19712 x 768 x 768] 48 -> 38 (231 tiles), [19712 x 512 x 2048] 66 -> 51 (154 tiles); bf16 [6400 x 2304 x 768]
	// 39 -> 26 us (225 tiles).  Below ~100 tiles the 128^2 kernel's two resident workgroups win by 3-5 % ([8192 x 768 x 3072] 96 tiles: 62 against 64 us).
	else if (pipelined && K / TK >= 2 && (N + 255) / 256 >= 2 && t256 >= 144) tn = 256;
	if (tn == 0) return 1;
	g.tiles_n = (N + tn - 1) / tn;
	const int ntiles = g.tiles_m * g.tiles_n;
	g.group_n = 4096 / K;        // B chunk = group_n * 256 rows * K * 2 B <= 2 MiB of the XCD's 4 MiB L2
	if (g.group_n < 4) g.group_n = 4;
	if (g.group_n > g.tiles_n) g.group_n = g.tiles_n;
	g.nk = K / TK;
	g.trace = g_trace.load(std::memory_order_relaxed);
	g.ncu = ncu;
	g.pipelined = pipelined;
	g.ep = *ep;
	// K-split tail (callers that hand over scratch: the ViT / text towers).  A few tiles more than whole rounds of 256 cost a whole extra round on 1-64
	// CUs (ViT-L/14 at batch 256: 257 x 4 = 1028 tiles for proj / fc2 -- five rounds for 4.02 rounds of work): the tiles behind the last full round
	// are cut along K into up to 256 / tail parts, one per workgroup, and finished by gemm256_tail_kernel.  Different summation order: not
	// bit-identical to the unsplit kernels (deterministic, though), so only where the caller asks for it.
	g.tail_first = ntiles;
	g.tail_split = 0;
	g.ws = nullptr;
	g.tail_dyn = 0;
	g.ws_bytes = 0;
	if (dyn_tail) {
		g.tail_dyn = 1;
		g.ws = (float*)ep->splitk_ws;
		g.ws_bytes = ep->splitk_ws_bytes;
	}
	// Worth it where the extra round is long: K >= 2048, or K >= 1024 with the fp32 residual epilogue (measured at ViT-L/14, batch 256: proj 249 ->
	// 211 us, fc2 665 -> 556 us; QKV and fc1, K = 1024 with the bf16 epilogue, 362 -> 371 and 547 -> 555 us: left unsplit).
	const int tail_probe = ntiles % ncu;
	// (round 3, 8-phase K loop: K = 1024 with the bf16 epilogues pays as well when the tail is a handful of tiles -- ViT-L/14 at batch 256: QKV 3084 tiles = 12 rounds + 12
	// tiles, fc1 4112 = 16 rounds + 16 -- see tools/vit_l14_tail_ab.py)
	if (tn == 256 && !ep->row_limit && ep->splitk_ws && ntiles > ncu &&
	    (g.nk >= 32 || (g.nk >= 16 && (ep->kind == NOVIC_EPI_RESID_F32 || ep->kind == NOVIC_EPI_RESID_F16 || tail_probe <= 32)))) {
		const int tail = ntiles % ncu;
		if (tail > 0 && tail <= 64) {
			int S = ncu / tail;
			if (S > g.nk / 4) S = g.nk / 4;
			if (S >= 2) {
				const int per = (g.nk + S - 1) / S;
				S = (g.nk + per - 1) / per;  // every part non-empty
				if (S >= 2 && (uint64_t)tail * S * 65536ull * 4ull <= ep->splitk_ws_bytes && ((uintptr_t)ep->splitk_ws & 15) == 0) {
					g.tail_first = ntiles - tail;
					g.tail_split = S;
					g.ws = (float*)ep->splitk_ws;
				}
			}
		}
	}
	int grid = ntiles < ncu ? ((ntiles + 7) / 8) * 8 : ncu;
	// No more workgroups than the rounds need: 450 tiles take two rounds on 256 CUs and on 232 alike (57 tiles per XCD over 29 slots) -- the same time, and 24 CUs stay
	// free for whatever runs on other streams meanwhile (the decode steps beside a tower, another lane).  Host row counts only: with a device row count the tiles that
	// really run are fewer than planned here, and a smaller grid could cost them a round.
	if (ntiles > ncu && !ep->row_limit && g.tail_split <= 1 && !g.tail_dyn) {
		const int rounds = (ntiles + ncu - 1) / ncu, per_xcd = (ntiles + 7) / 8;
		const int slots = (per_xcd + rounds - 1) / rounds;
		if (slots * 8 < grid) grid = slots * 8;
	}
	tn_out = tn;
	grid_out = grid;
	return 0;
}

int novic_gemm256_try(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const novic_epilogue_t* ep, int force, int* tile_n, hipStream_t stream) {
	Gemm256Args g;
	int tn = 0, grid = 0;
	// An A operand beyond the 2 GiB of a buffer descriptor (the multiset step's logits gradient: 172 032 x 6912 bf16 = 2.4 GB) runs as several launches over row ranges, each
	// planned like a problem of its own (same tiles, same K order per output element: bit-identical to one launch); a device row count is taken relative to the range's first
	// row.  Plain bf16 stores only (bias / activation are per element; the residual and dropout epilogues index rows globally and stay with the 128 x 128 kernel).
	const uint64_t a_bytes = (uint64_t)M * lda * 2;
	if (a_bytes >= 0x7FFFFFF0ull && ep->kind == NOVIC_EPI_STORE_BF16 && !ep->c2 && lda > 0) {
		const int per = (int)(0x7FFFFFF0ull / ((uint64_t)lda * 2)) / 256 * 256;
		if (per < 256 * 64) return 1;
		// EVERY range is planned before the first one is launched: a range the plan declines sends the whole call to the 128 x 128 kernel with nothing written yet
		// (round 4 launched as it planned -- a decline behind the first range returned an error with C partly written: advisor, round 4)
		constexpr int MAXR = 16;
		Gemm256Args gs[MAXR];
		int grids[MAXR], nr = 0, flags = 0;
		for (int row0 = 0; row0 < M; row0 += per, ++nr) {
			if (nr == MAXR) return 1;
			const int mc = M - row0 < per ? M - row0 : per;
			novic_epilogue_t e2 = *ep;
			e2.c = (char*)ep->c + (size_t)row0 * ep->ldc * 2;
			const char* a0 = (const char*)A + (size_t)row0 * lda * 2;
			int bad = plan256(a0, B, mc, N, K, lda, ldb, &e2, force, gs[nr], tn, grids[nr]);
			if (bad || tn != 256) bad = plan256(a0, B, mc, N, K, lda, ldb, &e2, 256, gs[nr], tn, grids[nr]);  // (a short last range is below the rules' tile counts: the same tile all the same)
			if (bad || tn != 256) return 1;
			gs[nr].row_base = row0;
			flags |= tn | (gs[nr].tail_split > 1 ? 0x1000 : 0) | (gs[nr].tail_dyn ? 0x2000 : 0);
		}
		for (int i = 0; i < nr; ++i) {
			const int rc = launch256_epi<4>(gs[i], grids[i], stream);
			if (rc) {
				novic_set_error("novic_gemm_bf16: a row range of an A operand beyond 2 GiB could not be launched on the 256-wide tile");
				return -22;
			}
		}
		if (tile_n) *tile_n = flags;
		return 0;
	}
	if (plan256(A, B, M, N, K, lda, ldb, ep, force, g, tn, grid)) return 1;
	if (tile_n) *tile_n = tn | (g.tail_split > 1 ? 0x1000 : 0) | (g.tail_dyn ? 0x2000 : 0);  // + whether a K-split tail runs (host-planned / planned on the device)
	return tn == 256 ? launch256_epi<4>(g, grid, stream) : launch256_epi<3>(g, grid, stream);
}

extern "C" int novic_gemm256_plan(int M, int N, int K, const novic_epilogue_t* ep, int* out4) {  // see include/novic_hip.h
	if (!ep || !out4) return -22;
	Gemm256Args g;
	int tn = 0, grid = 0;
	out4[0] = out4[1] = out4[2] = out4[3] = 0;
	if (plan256(nullptr, nullptr, M, N, K, K, K, ep, 0, g, tn, grid)) return 0;
	out4[0] = tn;
	out4[1] = grid;
	out4[2] = g.tail_dyn ? -1 : (g.tail_split > 1 ? g.tail_split : 0);
	out4[3] = g.tail_split > 1 ? g.tiles_m * g.tiles_n - g.tail_first : 0;
	return 0;
}
