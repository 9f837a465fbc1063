// Decoder self-attention over short prefix+label sequences (S <= 32), forward and backward, one wave per (sequence, head).
//
// reference: nn.TransformerEncoderLayer._sa_block -> F.scaled_dot_product_attention with the additive mask built in
// embedding_decoder.py:651-654 (prefix block bidirectional, rest causal) + :696-712 (per-row key padding, column 0 never masked).
// Here the mask is never materialised: allowed(i,j) = (j <= i or (i < P and j < P)) and not (key_pad[a][j] and j > 0).
//
// The whole S x S problem of one head sits in one wave's registers: the Q/K/V (/dO) rows of the head are staged once into
// LDS with 16-byte coalesced loads, QK^T is issued "swapped" (keys on MFMA rows, queries on columns) so each lane owns one
// query column and its softmax reduces in-lane + 2 shuffles, and P goes straight back into the PV MFMA as the B operand --
// the MFMA k index is permuted (k = 8g+j <-> key 4g+j | 16+4g+j-4) so that no lane movement is needed.  Operands that must be
// read "down the rows" (V, and in backward K / Q / dO) come from the same LDS image through ds_read_b64_tr_b16.
// Algorithmic bytes (HBM): forward 4*S*3E + 2*S*E per sequence... i.e. read qkv once, write o once; the kernel is HBM/issue-bound.
#include "common.hpp"
#include "novic_hip.h"

#ifndef DEC_ATTN_DIAG_EXTRA_LDS
#define DEC_ATTN_DIAG_EXTRA_LDS 0
#endif

namespace {

template <int D>
__device__ __forceinline__ int lds_off(int row, int col) {
	constexpr int CPR = D / 8;  // 16-byte chunks per row
	return row * (D * 2) + ((((col >> 3) ^ (row & (CPR - 1) & 7))) << 4) + ((col & 7) << 1);
}

// One head's rows [0, S) x D travel HBM -> registers (16-byte coalesced loads, issued one (sequence, head) pair AHEAD of their use so that they
// fly while the previous pair is computed) -> LDS (zero rows up to ROWS).
template <int D, int ROWS>
struct HeadRegs {
	static constexpr int CPR = D / 8, N = (ROWS * CPR + 63) / 64;
	uint4 v[N];
	// rows [0, S) of the head segment at element offset `base` of the buffer behind `srd` (row stride in elements): range-checked buffer loads, a row that
	// does not exist gets an out-of-range offset and comes back as zeros -- no predicate, no branch (the kernels are bound by instruction issue: every
	// predicated load was an s_and_saveexec / s_cbranch pair plus 64-bit address arithmetic)
	__device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t srd, unsigned base, int row_stride, int S, int lane) {
		typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
		for (int k = 0; k < N; ++k) {
			const int c = lane + 64 * k, row = c / CPR, ch = c - row * CPR;
			const unsigned off = (c < ROWS * CPR && row < S) ? (base + (unsigned)(row * row_stride + ch * 8)) * 2u : 0xFFFFFFF0u;
			const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(srd, off, 0, 0);
			v[k] = (uint4){t[0], t[1], t[2], t[3]};
		}
	}
	__device__ __forceinline__ void to_lds(char* lds, int lane) const {
#pragma unroll
		for (int k = 0; k < N; ++k) {
			const int c = lane + 64 * k, row = c / CPR, ch = c - row * CPR;
			if (c < ROWS * CPR) *reinterpret_cast<uint4*>(lds + lds_off<D>(row, ch * 8)) = v[k];
		}
	}
};

// lane l -> X[tile*16 + (l&15)][ks*32 + 8*(l>>4) + 0..7]   (zero beyond D)
template <int D>
__device__ __forceinline__ bf16x8 row_frag(const char* lds, int tile, int ks, int lane) {
	const int col = ks * 32 + 8 * (lane >> 4);
	bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
	if (col >= D) return z;
	return *reinterpret_cast<const bf16x8*>(lds + lds_off<D>(tile * 16 + (lane & 15), col));
}

// Column of X that MFMA output column c (= 4p + r of tile dt) stands for.  For D >= 32 the columns of two neighbouring tiles are interleaved so
// that a lane's accumulators of tiles 2u and 2u+1 are 8 CONSECUTIVE columns (u*32 + 8*(c>>2) + 0..7): outputs leave as 16-byte stores, four
// neighbouring lanes writing 64 contiguous bytes of a row instead of 8-byte pieces 32 bytes apart.
template <int D>
__device__ __forceinline__ int out_col(int dt, int p) {
	return D >= 32 ? (dt >> 1) * 32 + p * 8 + (dt & 1) * 4 : dt * 16 + 4 * p;
}

// lane l -> X[idx(kk)][out_col(dt, l&15 ...)] for kk = 8g + j: j < 4 -> row 4g+j, j >= 4 -> row 16+4g+(j-4) (only if NTS == 2)
template <int D, int NTS>
__device__ __forceinline__ bf16x8 tr_frag(const char* lds, int dt, int lane) {
	typedef bf16x4 __attribute__((address_space(3))) * lds4_t;
	const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
	const int col = out_col<D>(dt, p);
	bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(lds + lds_off<D>(4 * g + q, col)));
	bf16x4 hi = {0, 0, 0, 0};
	if (NTS == 2) hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(lds + lds_off<D>(16 + 4 * g + q, col)));
	return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ float group_sum(float v) {  // across the 4 lane groups that share lane&15
	v += __shfl_xor(v, 16, 64);
	v += __shfl_xor(v, 32, 64);
	return v;
}
__device__ __forceinline__ float group_max(float v) {
	v = fmaxf(v, __shfl_xor(v, 16, 64));
	v = fmaxf(v, __shfl_xor(v, 32, 64));
	return v;
}

// acc[dt] = 4 output columns out_col(dt, gq) + 0..3 of one row: write them to dst (the row's head segment)
// (the row's head segment starts at element offset `off` of the buffer behind `srd`; valid = false: the row does not exist, the stores go out of range and are dropped)
template <int D>
__device__ __forceinline__ void store_row(__amdgpu_buffer_rsrc_t srd, unsigned off, bool valid, const f32x4 (&acc)[D / 16], int gq) {
	typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
	typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
	if (D >= 32) {
#pragma unroll
		for (int u = 0; u < D / 32; ++u) {
			const f32x4 lo = acc[2 * u], hi = acc[2 * u + 1];
			bf16x8 o = {(bf16)lo[0], (bf16)lo[1], (bf16)lo[2], (bf16)lo[3], (bf16)hi[0], (bf16)hi[1], (bf16)hi[2], (bf16)hi[3]};
			__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), srd, valid ? (off + (unsigned)(u * 32 + gq * 8)) * 2u : 0xFFFFFFF0u, 0, 0);
		}
	} else {
		bf16x4 o = {(bf16)acc[0][0], (bf16)acc[0][1], (bf16)acc[0][2], (bf16)acc[0][3]};
		__builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, o), srd, valid ? (off + (unsigned)(4 * gq)) * 2u : 0xFFFFFFF8u, 0, 0);
	}
}

struct AttnArgs {
	const bf16* qkv;       // [A*S][3E]
	const uint8_t* keypad; // [A][S] or null
	bf16* o;               // fwd: [A*S][E]
	const bf16* d_o;       // bwd: [A*S][E]
	bf16* dqkv;            // bwd: [A*S][3E]
	int A, S, H, P, strict;
	float scale;
	DropoutDesc drop;
	// packed rows (optional): sequence a occupies rows seq_start[a] .. seq_start[a] + seq_len[a] - 1 of qkv / o / d_o / dqkv instead of a*S .. a*S + S - 1;
	// the positions from seq_len[a] on are padding (key-padded in `keypad`, which stays [A][S]) and simply do not exist
	const int* seq_start;
	const int* seq_len;
};
__device__ __forceinline__ int seq_row0(const AttnArgs& g, int a) { return g.seq_start ? g.seq_start[a] : a * g.S; }
__device__ __forceinline__ int seq_rows(const AttnArgs& g, int a) { return g.seq_len ? min(g.seq_len[a], g.S) : g.S; }

// padmask: bit j set <=> key j of this sequence is padding (j > 0); built once per (sequence, head) with one byte load per lane + a ballot
__device__ __forceinline__ uint32_t pad_bits(const uint8_t* kp, int S, int lane) {
	if (!kp) return 0u;
	const bool in = lane > 0 && lane < S;
	return (uint32_t)__ballot(in && kp[in ? lane : 0] != 0);  // (clamped index: an unconditional byte load instead of a predicated one)
}
__device__ __forceinline__ bool allowed(const AttnArgs& g, uint32_t padmask, int i, int j) {
	if (i >= g.S || j >= g.S) return false;  // (rows beyond a packed sequence's length are key-padded: padmask covers them)
	const bool vis = (j <= i) || (!g.strict && i < g.P && j < g.P);
	return vis && !((padmask >> j) & 1u);
}
// Dropout on the attention probabilities: element (pair, i, j) has mask index (pair*32 + i)*32 + j, so the four keys j = 4g..4g+3 a lane owns
// in the "query column" layout come out of ONE dropout_scale4 call (the kernels are VALU-bound: a call per element tripled their time).
__device__ __forceinline__ void drop4(const DropoutDesc& d, int pair, int i, int j0, float (&s)[4]) { dropout_scale4(d, ((uint64_t)pair * 32 + i) * 32 + j0, s); }

// What a wave multiplies in one pass: the rows of ONE sequence -- or, in the packed layout with S <= 16, of TWO neighbouring sequences whose rows
// are contiguous and fit one 16-row tile together (lengths 5..10 in the training step: 72 % of the neighbour pairs).  The kernels are bound by the
// instructions per tile, not by its rows, so a merged tile does two sequences for the price of one: the scores between rows of different
// sequences are masked like padding, everything else is the single-sequence arithmetic on tile-local indices.
struct Tile {
	int row0;    // first row in qkv / o / d_o / dqkv
	int nrows;   // rows that exist (loads, stores)
	int n0;      // rows of the first sequence; >= 32: the tile is a single sequence
	int lim;     // index bound of allowed(): S for a single sequence (rows behind its length are key-padded), nrows for a merged tile
	int a0, a1;  // the sequences (a1 < 0: none)
	int key;     // dropout key: mask index = (key * 32 + i) * 32 + j with tile-local i, j (a single sequence a: key = a * H + h, as ever)
	int h;
};
__device__ __forceinline__ bool allowed_t(const AttnArgs& g, const Tile& t, uint32_t kp0, uint32_t kp1, int i, int j) {
	if (i >= t.lim || j >= t.lim) return false;
	const int oi = i >= t.n0 ? t.n0 : 0, oj = j >= t.n0 ? t.n0 : 0;
	if (oi != oj) return false;  // rows of different sequences never see each other
	const int li = i - oi, lj = j - oj;
	const bool vis = (lj <= li) || (!g.strict && li < g.P && lj < g.P);
	return vis && !(((oj ? kp1 : kp0) >> lj) & 1u);
}
// allowed_t for a whole row / column at once, as one word per lane and tile (the kernels are bound by instruction issue, and four allowed_t calls per lane and
// MFMA tile were ~60 of their instructions):
//   query_mask(x): bit j set <=> allowed_t(i = x, j)   (layout "lane owns query column i")
//   key_mask(x)  : bit i set <=> allowed_t(i, j = x)   (layout "lane owns key column j" of the backward pass)
__device__ __forceinline__ uint32_t lowbits(int n) { return n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u); }
__device__ __forceinline__ uint32_t query_mask(const AttnArgs& g, const Tile& t, uint32_t kp0, uint32_t kp1, int x) {
	const bool second = x >= t.n0;
	const int off = second ? t.n0 : 0, lx = x - off;
	const int seglen = second ? t.lim - t.n0 : min(t.n0, t.lim);
	uint32_t m = lowbits(lx + 1) | ((!g.strict && lx < g.P) ? lowbits(g.P) : 0u);
	m &= ~(second ? kp1 : kp0) & lowbits(seglen);
	return x < t.lim ? m << off : 0u;
}
__device__ __forceinline__ uint32_t key_mask(const AttnArgs& g, const Tile& t, uint32_t kp0, uint32_t kp1, int x) {
	const bool second = x >= t.n0;
	const int off = second ? t.n0 : 0, lx = x - off;
	const int seglen = second ? t.lim - t.n0 : min(t.n0, t.lim);
	uint32_t m = ~lowbits(lx) | ((!g.strict && lx < g.P) ? lowbits(g.P) : 0u);
	m &= lowbits(seglen);
	const bool padded = (((second ? kp1 : kp0) >> lx) & 1u) != 0u;
	return (x < t.lim && !padded) ? m << off : 0u;
}

// Problems: (sequence, head) -- or (pair of neighbouring sequences, head) when tiles may merge; problem q has one tile, or two when its
// sequences do not merge.  Returns true if tile (q, 1) follows tile (q, 0).
template <int NTS>
__device__ __forceinline__ bool make_tile(const AttnArgs& g, int q, int sub, Tile& t) {
	if (NTS == 1 && g.seq_start) {
		const int ap = q / g.H, h = q - ap * g.H, a0 = 2 * ap, a1 = a0 + 1 < g.A ? a0 + 1 : -1;
		const int s0 = g.seq_start[a0], l0 = min(g.seq_len[a0], g.S);
		int s1 = 0, l1 = 0;
		if (a1 >= 0) { s1 = g.seq_start[a1]; l1 = min(g.seq_len[a1], g.S); }
		const bool merged = a1 >= 0 && s1 == s0 + l0 && l0 + l1 <= 16;
		if (sub == 0) {
			if (merged) { t = {s0, l0 + l1, l0, l0 + l1, a0, a1, a0 * g.H + h, h}; return false; }
			t = {s0, l0, 32, g.S, a0, -1, a0 * g.H + h, h};
			return a1 >= 0;
		}
		t = {s1, l1, 32, g.S, a1, -1, a1 * g.H + h, h};
		return false;
	}
	const int a = q / g.H, h = q - a * g.H;
	t = {seq_row0(g, a), seq_rows(g, a), 32, g.S, a, -1, q, h};
	return false;
}
template <int NTS>
__device__ __forceinline__ int num_problems(const AttnArgs& g) { return (NTS == 1 && g.seq_start) ? ((g.A + 1) / 2) * g.H : g.A * g.H; }

template <int D, int NTS>
__global__ __launch_bounds__(256) void dec_attn_fwd_kernel(const AttnArgs g) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	constexpr int ROWS = NTS * 16, TILE = ROWS * D * 2, KS = (D + 31) / 32, DT = D / 16;
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	char* lq = smem + w * 3 * TILE;  // wave-private: no workgroup barrier anywhere
	char* lk = lq + TILE;
	char* lv = lk + TILE;
	const int E = g.H * D, total = num_problems<NTS>(g), stride = gridDim.x * 4;
	const int gq = lane >> 4;
	HeadRegs<D, ROWS> rq, rk, rv;
	const unsigned rows_all = (unsigned)g.A * (unsigned)g.S;
	const __amdgpu_buffer_rsrc_t s_qkv = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.qkv), 0, rows_all * (unsigned)(3 * E) * 2u, 0x00020000);
	const __amdgpu_buffer_rsrc_t s_o = __builtin_amdgcn_make_buffer_rsrc(g.o, 0, rows_all * (unsigned)E * 2u, 0x00020000);
	auto fetch = [&](const Tile& t) {
		const unsigned base = (unsigned)t.row0 * (unsigned)(3 * E) + (unsigned)(t.h * D);
		rq.load(s_qkv, base, 3 * E, t.nrows, lane);
		rk.load(s_qkv, base + E, 3 * E, t.nrows, lane);
		rv.load(s_qkv, base + 2 * E, 3 * E, t.nrows, lane);
	};
	// the tile sequence of a wave is wave-uniform: kept in scalar registers (readfirstlane), so make_tile's index arithmetic and its seq_start / seq_len loads run
	// on the scalar unit beside the vector instructions instead of among them
	int q = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + w);
	Tile t, tn;
	bool more = false;
	if (q < total) { more = make_tile<NTS>(g, q, 0, t); fetch(t); }
	while (q < total) {
		const int nq = more ? q : q + stride;
		bool nmore = false;
		rq.to_lds(lq, lane);
		rk.to_lds(lk, lane);
		rv.to_lds(lv, lane);
		if (nq < total) { nmore = make_tile<NTS>(g, nq, more ? 1 : 0, tn); fetch(tn); }  // the next tile's rows fly while this one is computed
		const uint32_t kp = pad_bits(g.keypad ? g.keypad + (size_t)t.a0 * g.S : nullptr, g.S, lane);
		const uint32_t kp1 = (t.a1 >= 0) ? pad_bits(g.keypad ? g.keypad + (size_t)t.a1 * g.S : nullptr, g.S, lane) : 0u;
		const int pair = t.key;

#pragma unroll
		for (int qt = 0; qt < NTS; ++qt) {
			const int i = qt * 16 + (lane & 15);
			const uint32_t qmask = query_mask(g, t, kp, kp1, i) >> (4 * gq);  // bit kt * 16 + r: key kt * 16 + 4 gq + r is visible to query i
			float p[NTS][4];
			float mx = -1e30f;
#pragma unroll
			for (int kt = 0; kt < NTS; ++kt) {
				f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
				for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<D>(lk, kt, ks, lane), row_frag<D>(lq, qt, ks, lane), acc, 0, 0, 0);
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					p[kt][r] = ((qmask >> (kt * 16 + r)) & 1u) ? acc[r] * g.scale : -1e30f;
					mx = fmaxf(mx, p[kt][r]);
				}
			}
			mx = group_max(mx);
			float sum = 0.f;
#pragma unroll
			for (int kt = 0; kt < NTS; ++kt)
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					p[kt][r] = (p[kt][r] > -1e29f) ? __expf(p[kt][r] - mx) : 0.f;
					sum += p[kt][r];
				}
			sum = group_sum(sum);
			const float inv = sum > 0.f ? 1.f / sum : 0.f;
			bf16x8 pf;
#pragma unroll
			for (int kt = 0; kt < 2; ++kt) {
				float dm[4] = {1.f, 1.f, 1.f, 1.f};
				if (kt < NTS && g.drop.p > 0.f) drop4(g.drop, pair, i, kt * 16 + 4 * gq, dm);
#pragma unroll
				for (int r = 0; r < 4; ++r) pf[kt * 4 + r] = (bf16)(kt < NTS ? p[kt][r] * inv * dm[r] : 0.f);
			}
			f32x4 oacc[DT];
#pragma unroll
			for (int dt = 0; dt < DT; ++dt) {
				oacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
				oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<D, NTS>(lv, dt, lane), pf, oacc[dt], 0, 0, 0);
			}
			store_row<D>(s_o, (unsigned)(t.row0 + i) * (unsigned)E + (unsigned)(t.h * D), i < t.nrows, oacc, gq);
		}
		t = tn;
		q = nq;
		more = nmore;
	}
}

template <int D, int NTS>
__global__ __launch_bounds__(256, 3) void dec_attn_bwd_kernel(const AttnArgs g) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	constexpr int ROWS = NTS * 16, TILE = ROWS * D * 2, KS = (D + 31) / 32, DT = D / 16;
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	char* lq = smem + w * 4 * TILE;  // wave-private: no workgroup barrier anywhere
	char* lk = lq + TILE;
	char* lv = lk + TILE;
	char* ld = lv + TILE;
	const int E = g.H * D, total = num_problems<NTS>(g), stride = gridDim.x * 4;
	const int gq = lane >> 4;
	HeadRegs<D, ROWS> rq, rk, rv, rd;
	const unsigned rows_all = (unsigned)g.A * (unsigned)g.S;
	const __amdgpu_buffer_rsrc_t s_qkv = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.qkv), 0, rows_all * (unsigned)(3 * E) * 2u, 0x00020000);
	const __amdgpu_buffer_rsrc_t s_do = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.d_o), 0, rows_all * (unsigned)E * 2u, 0x00020000);
	const __amdgpu_buffer_rsrc_t s_dqkv = __builtin_amdgcn_make_buffer_rsrc(g.dqkv, 0, rows_all * (unsigned)(3 * E) * 2u, 0x00020000);
	auto fetch = [&](const Tile& t) {
		const unsigned base = (unsigned)t.row0 * (unsigned)(3 * E) + (unsigned)(t.h * D);
		rq.load(s_qkv, base, 3 * E, t.nrows, lane);
		rk.load(s_qkv, base + E, 3 * E, t.nrows, lane);
		rv.load(s_qkv, base + 2 * E, 3 * E, t.nrows, lane);
		rd.load(s_do, (unsigned)t.row0 * (unsigned)E + (unsigned)(t.h * D), E, t.nrows, lane);
	};
	int q = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + w);  // (wave-uniform tile sequence on the scalar unit: see the forward kernel)
	Tile t, tn;
	bool more = false;
	if (q < total) { more = make_tile<NTS>(g, q, 0, t); fetch(t); }
	while (q < total) {
	const int nq = more ? q : q + stride;
	bool nmore = false;
	rq.to_lds(lq, lane);
	rk.to_lds(lk, lane);
	rv.to_lds(lv, lane);
	rd.to_lds(ld, lane);
	if (nq < total) { nmore = make_tile<NTS>(g, nq, more ? 1 : 0, tn); fetch(tn); }  // the next tile's rows fly while this one is computed
	const uint32_t kp = pad_bits(g.keypad ? g.keypad + (size_t)t.a0 * g.S : nullptr, g.S, lane);
	const uint32_t kp1 = (t.a1 >= 0) ? pad_bits(g.keypad ? g.keypad + (size_t)t.a1 * g.S : nullptr, g.S, lane) : 0u;
	const int pair = t.key;
	const float drop_inv = g.drop.p > 0.f ? 1.f / (1.f - g.drop.p) : 1.f;
	const unsigned dq_base = (unsigned)t.row0 * (unsigned)(3 * E) + (unsigned)(t.h * D);  // element offset of the tile's dQ segment in dqkv
	const int Sa = t.nrows;

	// ---- layout 1: lane owns query column i, key rows j = 4g+r: softmax stats, delta, dS -> dQ ----
	float mx1[NTS], inv1[NTS], dl1[NTS];
	// keep[qt][kt][r]: wave-wide ballot of "probability (i = qt*16 + lane&15, j = kt*16 + 4*(lane>>4) + r) survives dropout": layout 2 reads its
	// masks out of these words instead of hashing again
	uint64_t keep[NTS][NTS][4];
#pragma unroll
	for (int qt = 0; qt < NTS; ++qt) {
		const int i = qt * 16 + (lane & 15);
		const uint32_t qmask = query_mask(g, t, kp, kp1, i) >> (4 * gq);
		float p[NTS][4], dp[NTS][4];
		float mx = -1e30f;
#pragma unroll
		for (int kt = 0; kt < NTS; ++kt) {
			f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
			for (int ks = 0; ks < KS; ++ks) {
				acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<D>(lk, kt, ks, lane), row_frag<D>(lq, qt, ks, lane), acc, 0, 0, 0);
				acd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<D>(lv, kt, ks, lane), row_frag<D>(ld, qt, ks, lane), acd, 0, 0, 0);
			}
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				p[kt][r] = ((qmask >> (kt * 16 + r)) & 1u) ? acc[r] * g.scale : -1e30f;
				dp[kt][r] = acd[r];
				mx = fmaxf(mx, p[kt][r]);
			}
		}
		mx = group_max(mx);
		float sum = 0.f;
#pragma unroll
		for (int kt = 0; kt < NTS; ++kt)
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				p[kt][r] = (p[kt][r] > -1e29f) ? __expf(p[kt][r] - mx) : 0.f;
				sum += p[kt][r];
			}
		sum = group_sum(sum);
		const float inv = sum > 0.f ? 1.f / sum : 0.f;
		float delta = 0.f;
#pragma unroll
		for (int kt = 0; kt < NTS; ++kt) {
			float dm[4] = {1.f, 1.f, 1.f, 1.f};
			if (g.drop.p > 0.f) drop4(g.drop, pair, i, kt * 16 + 4 * gq, dm);
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				keep[qt][kt][r] = __ballot(dm[r] != 0.f);
				p[kt][r] *= inv;
				dp[kt][r] *= dm[r];
				delta += dp[kt][r] * p[kt][r];
			}
		}
		delta = group_sum(delta);
		mx1[qt] = mx; inv1[qt] = inv; dl1[qt] = delta;
		bf16x8 dsf;
#pragma unroll
		for (int kt = 0; kt < 2; ++kt)
#pragma unroll
			for (int r = 0; r < 4; ++r) dsf[kt * 4 + r] = (bf16)(kt < NTS ? p[kt][r] * (dp[kt][r] - delta) * g.scale : 0.f);
		f32x4 qacc[DT];
#pragma unroll
		for (int dt = 0; dt < DT; ++dt) {
			qacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
			qacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<D, NTS>(lk, dt, lane), dsf, qacc[dt], 0, 0, 0);
		}
		store_row<D>(s_dqkv, dq_base + (unsigned)(i * 3 * E), i < Sa, qacc, gq);
	}

	// ---- layout 2: lane owns key column j, query rows i = 4g+r: Pd and dS -> dV, dK ----
#pragma unroll
	for (int kt = 0; kt < NTS; ++kt) {
		const int j = kt * 16 + (lane & 15);
		const uint32_t kmask = key_mask(g, t, kp, kp1, j) >> (4 * gq);  // bit qt * 16 + r: query qt * 16 + 4 gq + r sees key j
		bf16x8 pdf, dsf;
#pragma unroll
		for (int qt = 0; qt < 2; ++qt) {
			if (qt < NTS) {
				f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
				for (int ks = 0; ks < KS; ++ks) {
					acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<D>(lq, qt, ks, lane), row_frag<D>(lk, kt, ks, lane), acc, 0, 0, 0);
					acd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<D>(ld, qt, ks, lane), row_frag<D>(lv, kt, ks, lane), acd, 0, 0, 0);
				}
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const int il = 4 * gq + r;
					const float mx = __shfl(mx1[qt < NTS ? qt : 0], il, 64);
					const float inv = __shfl(inv1[qt < NTS ? qt : 0], il, 64);
					const float delta = __shfl(dl1[qt < NTS ? qt : 0], il, 64);
					float p = ((kmask >> (qt * 16 + r)) & 1u) ? __expf(acc[r] * g.scale - mx) * inv : 0.f;
					// (i, j) was lane (i & 15) + 16 * ((j & 15) >> 2) of ballot keep[qt][kt][j & 3]
					const int jl = lane & 15;
					const int qs = qt < NTS ? qt : 0;
					// (selected by masks, not by ?: -- hipcc turned the ternaries over the four ballot words into a per-lane SCRATCH array indexed by jl & 3: 32 bytes stored and
					// 8 loaded per lane and tile, 48 bytes of private segment per lane, the write traffic of the kernel 1.4 x its output: round 4, tools/audit_vmcnt.py checks)
					const uint64_t k0 = keep[qs][kt][0], k1 = keep[qs][kt][1], k2 = keep[qs][kt][2], k3 = keep[qs][kt][3];
					const uint64_t m1 = 0ull - (uint64_t)(jl & 1), m2 = 0ull - (uint64_t)((jl >> 1) & 1);
					const uint64_t w01 = k0 ^ ((k0 ^ k1) & m1), w23 = k2 ^ ((k2 ^ k3) & m1);
					const uint64_t wsel = w01 ^ ((w01 ^ w23) & m2);
					const float dm = ((wsel >> (il + 16 * (jl >> 2))) & 1ull) ? drop_inv : 0.f;
					pdf[qt * 4 + r] = (bf16)(p * dm);
					dsf[qt * 4 + r] = (bf16)(p * (acd[r] * dm - delta) * g.scale);
				}
			} else {
#pragma unroll
				for (int r = 0; r < 4; ++r) { pdf[qt * 4 + r] = (bf16)0.f; dsf[qt * 4 + r] = (bf16)0.f; }
			}
		}
		f32x4 av[DT], ak[DT];
#pragma unroll
		for (int dt = 0; dt < DT; ++dt) {
			av[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
			ak[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
			av[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<D, NTS>(ld, dt, lane), pdf, av[dt], 0, 0, 0);
			ak[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<D, NTS>(lq, dt, lane), dsf, ak[dt], 0, 0, 0);
		}
		store_row<D>(s_dqkv, dq_base + (unsigned)(j * 3 * E + 2 * E), j < Sa, av, gq);
		store_row<D>(s_dqkv, dq_base + (unsigned)(j * 3 * E + E), j < Sa, ak, gq);
	}
	t = tn;
	q = nq;
	more = nmore;
	}  // tiles
}

template <int D, int NTS>
int launch_attn(const AttnArgs& g, bool bwd, hipStream_t stream) {
	const int pairs = (NTS == 1 && g.seq_start) ? ((g.A + 1) / 2) * g.H : g.A * g.H;  // problems (num_problems)
	// every wave walks over several (sequence, head) pairs, fetching the next pair's rows while it computes the current one; enough workgroups
	// to fill every CU's LDS / wave slots a few times over, few enough that each wave still sees a pipeline of ~8 pairs
	int grid = (pairs + 3) / 4;
	// (DEC_ATTN_DIAG_EXTRA_LDS: diagnostic builds only -- tools/attn_occupancy_probe.sh -- bytes of dynamic LDS a workgroup asks for beyond what it uses, to measure these
	// kernels at the residency a fused prologue's LDS footprint would leave them: DESIGN.md section 4, "Round 6")
	const size_t shm = (size_t)4 * (bwd ? 4 : 3) * NTS * 16 * D * 2 + DEC_ATTN_DIAG_EXTRA_LDS;
	static std::atomic<int> resident[2];  // workgroups the chip holds at once (per template instance: function-local static, zero-initialised; two threads racing here compute the same value)
	if (!resident[bwd]) {
		int per_cu = 0, dev = 0, cus = 256;
		hipDeviceProp_t prop;
		if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
		const hipError_t e = bwd ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dec_attn_bwd_kernel<D, NTS>, 256, shm)
		                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dec_attn_fwd_kernel<D, NTS>, 256, shm);
		resident[bwd] = cus * ((e == hipSuccess && per_cu > 0) ? per_cu : 4);
	}
	if (grid > resident[bwd]) grid = resident[bwd];  // exactly one resident round: no tail round of partially filled CUs
	if (bwd) hipLaunchKernelGGL((dec_attn_bwd_kernel<D, NTS>), dim3(grid), dim3(256), shm, stream, g);
	else hipLaunchKernelGGL((dec_attn_fwd_kernel<D, NTS>), dim3(grid), dim3(256), shm, stream, g);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

int dispatch_attn(const AttnArgs& g, int D, bool bwd, hipStream_t stream) {
	const int nts = (g.S + 15) / 16;
#define NOVIC_ATTN_CASE(DD)                                                         \
	case DD:                                                                        \
		return nts == 1 ? launch_attn<DD, 1>(g, bwd, stream) : launch_attn<DD, 2>(g, bwd, stream);
	switch (D) {
		NOVIC_ATTN_CASE(16)
		NOVIC_ATTN_CASE(32)
		NOVIC_ATTN_CASE(64)
		default:
			novic_set_error("decoder attention supports head_dim 16, 32 or 64");
			return -22;
	}
#undef NOVIC_ATTN_CASE
}

}  // namespace

extern "C" int novic_dec_attn_fwd(const void* qkv_bf16, const uint8_t* key_pad, void* o_bf16, int A, int S, int H, int D, int P, int strictly_causal, float drop_p,
                                  uint64_t seed, uint32_t drop_site, const int* seq_start, const int* seq_len, hipStream_t stream) {
	NOVIC_CHECK((seq_start == nullptr) == (seq_len == nullptr), "novic_dec_attn_fwd: seq_start and seq_len go together");
	NOVIC_CHECK(qkv_bf16 && o_bf16, "novic_dec_attn_fwd: null pointer");
	NOVIC_CHECK(S >= 1 && S <= 32, "novic_dec_attn_fwd: sequence length must be in [1, 32]");
	NOVIC_CHECK(A >= 0 && H >= 1 && P >= 1, "novic_dec_attn_fwd: bad shape");
	NOVIC_CHECK((uint64_t)A * S * 3 * H * D * 2 < 0xFFFFFFF0ull, "novic_dec_attn_fwd: qkv must be smaller than 4 GiB (32-bit buffer offsets)");
	if (A == 0) return 0;
	AttnArgs g = {(const bf16*)qkv_bf16, key_pad, (bf16*)o_bf16, nullptr, nullptr, A, S, H, P, strictly_causal, 1.f / sqrtf((float)D),
	              {drop_p, (uint32_t)seed, (uint32_t)(seed >> 32), drop_site}, seq_start, seq_len};
	return dispatch_attn(g, D, false, stream);
}

extern "C" int novic_dec_attn_bwd(const void* qkv_bf16, const uint8_t* key_pad, const void* do_bf16, void* dqkv_bf16, int A, int S, int H, int D, int P,
                                  int strictly_causal, float drop_p, uint64_t seed, uint32_t drop_site, const int* seq_start, const int* seq_len,
                                  hipStream_t stream) {
	NOVIC_CHECK(qkv_bf16 && do_bf16 && dqkv_bf16, "novic_dec_attn_bwd: null pointer");
	NOVIC_CHECK((seq_start == nullptr) == (seq_len == nullptr), "novic_dec_attn_bwd: seq_start and seq_len go together");
	NOVIC_CHECK(S >= 1 && S <= 32, "novic_dec_attn_bwd: sequence length must be in [1, 32]");
	NOVIC_CHECK(A >= 0 && H >= 1 && P >= 1, "novic_dec_attn_bwd: bad shape");
	NOVIC_CHECK((uint64_t)A * S * 3 * H * D * 2 < 0xFFFFFFF0ull, "novic_dec_attn_bwd: qkv must be smaller than 4 GiB (32-bit buffer offsets)");
	if (A == 0) return 0;
	AttnArgs g = {(const bf16*)qkv_bf16, key_pad, nullptr, (const bf16*)do_bf16, (bf16*)dqkv_bf16, A, S, H, P, strictly_causal, 1.f / sqrtf((float)D),
	              {drop_p, (uint32_t)seed, (uint32_t)(seed >> 32), drop_site}, seq_start, seq_len};
	return dispatch_attn(g, D, true, stream);
}
