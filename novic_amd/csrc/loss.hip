// Padding bookkeeping, token cross-entropy (+ arg-max + in-place gradient) and the per-micro-batch loss reductions.
//
// reference: embedding_decoder.py:681-685 (zero weights fold into padding), :696-712 (sequence key padding), :729-761 (loss / basis / correct).
// The CE kernel is HBM-bound: per token row it reads V bf16 logits (twice, the second time out of L2) and, in training,
// writes V bf16 gradients in place: algorithmic bytes = 2*V (eval) or 4*V (train) per token.
#include "common.hpp"
#include "novic_hip.h"

namespace {

__device__ __forceinline__ long long load_tok(const void* tok, int tok_bytes, size_t i) {
	return tok_bytes == 8 ? ((const long long*)tok)[i] : (long long)((const int*)tok)[i];
}

// key_pad[a][s] (S = P + C - 1) and out_pad[a][c] from the caller's target padding (+ zero weights).
__global__ void build_padding_kernel(const uint8_t* __restrict__ tpad, const float* __restrict__ weight, uint8_t* __restrict__ key_pad, uint8_t* __restrict__ out_pad, int A,
                                     int C, int P, int N, int tpad_ld) {
	const int S = P + C - 1;
	const int n_expand = P + N - 2, n_keep = C - N + 1;
	for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < A * S; idx += gridDim.x * blockDim.x) {
		const int a = idx / S, s = idx - a * S;
		const bool zero_w = weight && weight[a] == 0.f;
		auto padv = [&](int c) -> bool { return zero_w || (tpad && tpad[(size_t)a * tpad_ld + c]); };
		bool kp;
		if (n_expand < 1) kp = padv(s);
		else if (n_keep <= 1 || s < n_expand) kp = padv(0);
		else kp = padv(s - n_expand);
		key_pad[idx] = (s > 0 && kp) ? 1 : 0;
		if (s >= S - C) {  // the padding the outputs use = the last C columns of the (un-forced) sequence padding
			out_pad[(size_t)a * C + (s - (S - C))] = kp ? 1 : 0;
		}
	}
}

// One wave per token row r = a*T + t.
struct CeArgs {
	bf16* logits;       // [R][ldl], overwritten with the gradient when dlogits != 0
	int ldl, V, A, T, C, col0;
	const void* target;  // [A][C] ids; row r uses column col0 + t
	int tok_bytes, tok_ld;
	const uint8_t* out_pad;  // [A][C] or null
	const float* weight;     // [A] or null
	const float* basis;      // [groups] (training) or null
	int group_rows;          // sequences per micro-batch group
	float grad_scale;        // upstream gradient * 1/accum
	const float* grad_scale_dev;  // optional device scalar multiplied in (autograd's incoming gradient)
	float smoothing;
	int write_grad;
	float* row_loss;         // [R]
	int* row_argmax;         // [R]
	uint8_t* row_correct;    // [R] or null
	int argmax_from;         // first vocabulary id eligible for the arg-max (1 => END excluded)
	const int* row_map;      // compacted form: logits row j belongs to token row row_map[j] (target / outputs indexed by it); null = identity
	const int* row_limit;    // compacted form: only the first *row_limit logits rows exist (device int); null = all A*T
};

__global__ __launch_bounds__(256) void ce_kernel(const CeArgs g) {
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int R = g.row_limit ? min(g.A * g.T, max(*g.row_limit, 0)) : g.A * g.T;
	for (int rj = blockIdx.x * 4 + w; rj < R; rj += gridDim.x * 4) {
		const int r = g.row_map ? g.row_map[rj] : rj;
		const int a = r / g.T, t = r - a * g.T;
		bf16* row = g.logits + (size_t)rj * g.ldl;
		long long tgt = g.target ? load_tok(g.target, g.tok_bytes, (size_t)a * g.tok_ld + g.col0 + t) : -1;
		const bool ignored = (g.out_pad && g.out_pad[(size_t)a * g.C + g.col0 + t]) || (g.weight && g.weight[a] == 0.f) || tgt < 0 || tgt >= g.V;
		// pass 1: online max / sum-exp, arg-max (lowest index on ties), sum of logits (label smoothing)
		float mx = -INFINITY, se = 0.f, sl = 0.f, bestv = -INFINITY;
		int besti = 0x7fffffff;
		for (int v0 = lane * 8; v0 < g.V; v0 += 512) {
			float x[8];
			if (v0 + 8 <= g.V) {
				const bf16x8 q = *reinterpret_cast<const bf16x8*>(row + v0);
#pragma unroll
				for (int i = 0; i < 8; ++i) x[i] = (float)q[i];
			} else {
#pragma unroll
				for (int i = 0; i < 8; ++i) x[i] = (v0 + i < g.V) ? (float)row[v0 + i] : -INFINITY;
			}
			float cm = x[0];
#pragma unroll
			for (int i = 1; i < 8; ++i) cm = fmaxf(cm, x[i]);
			const float nm = fmaxf(mx, cm);
			float cs = 0.f;
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				if (v0 + i < g.V) {
					cs += __expf(x[i] - nm);
					sl += x[i];
					if (v0 + i >= g.argmax_from && x[i] > bestv) { bestv = x[i]; besti = v0 + i; }
				}
			}
			se = se * __expf(mx - nm) + cs;
			mx = nm;
		}
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			const float omx = __shfl_xor(mx, o, 64), ose = __shfl_xor(se, o, 64);
			const float obv = __shfl_xor(bestv, o, 64);
			const int obi = __shfl_xor(besti, o, 64);
			const float nm = fmaxf(mx, omx);
			se = (nm == -INFINITY) ? 0.f : se * __expf(mx - nm) + ose * __expf(omx - nm);
			mx = nm;
			if (obv > bestv || (obv == bestv && obi < besti)) { bestv = obv; besti = obi; }
			sl += __shfl_xor(sl, o, 64);
		}
		const float lse = mx + __logf(se);
		float loss = 0.f;
		if (!ignored) {
			const float lt = (float)row[tgt];
			loss = lse - lt;
			if (g.smoothing > 0.f) loss = (1.f - g.smoothing) * loss + g.smoothing * (lse - sl / (float)g.V);
		}
		if (lane == 0) {
			g.row_loss[r] = loss;
			g.row_argmax[r] = besti;
			if (g.row_correct) g.row_correct[r] = (!ignored && besti == (int)tgt) ? 1 : 0;
		}
		if (g.write_grad) {
			float sc = 0.f;
			if (!ignored) {
				sc = g.grad_scale;
				if (g.grad_scale_dev) sc *= g.grad_scale_dev[0];
				if (g.weight) sc *= g.weight[a];
				if (g.basis) sc /= g.basis[a / g.group_rows];
			}
			const float us = g.smoothing / (float)g.V;
			for (int v0 = lane * 8; v0 < g.ldl; v0 += 512) {
				float x[8];
				const bf16x8 q = *reinterpret_cast<const bf16x8*>(row + v0);  // ldl is a multiple of 8
#pragma unroll
				for (int i = 0; i < 8; ++i) x[i] = (float)q[i];
				bf16x8 o;
#pragma unroll
				for (int i = 0; i < 8; ++i) {
					float gr = 0.f;
					if (v0 + i < g.V && sc != 0.f) {
						gr = __expf(x[i] - lse) - us;
						if (v0 + i == (int)tgt) gr -= (1.f - g.smoothing);
						gr *= sc;
					}
					o[i] = (bf16)gr;
				}
				*reinterpret_cast<bf16x8*>(row + v0) = o;
			}
		}
	}
}

// Register-resident variant for rows of at most NCH * 512 columns (ldl <= 8192: every vocabulary of the reference's models).  The looping kernel
// above is bound by the vector ALU, not by HBM (two exponentials per logit in the online pass, one more for the gradient, per-element arg-max
// and range bookkeeping: ~30 VALU operations per logit).  Here the row's NCH 16-byte pieces per lane are requested up front and stay in registers
// as packed bf16, with everything at or beyond column V overwritten by -inf once, so that no later pass needs a range test; the maximum is a
// max chain with the arg-max tracked per 8-logit piece (the winning piece is re-read from L1/L2 to find the element); each logit is exponentiated
// ONCE, summed in fp32 and parked as packed bf16 in the registers the logit came from; the gradient pass is one fma per logit.
// ~13 VALU operations per logit; the row is read once.
template <int NCH, bool SMOOTH>
__global__ __launch_bounds__(256) void ce_rows_kernel(const CeArgs g) {
	typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
	typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const float LOG2E = 1.4426950408889634f;
	const unsigned NINF2 = 0xFF80FF80u;  // two bf16 -inf
	const int R = g.row_limit ? min(g.A * g.T, max(*g.row_limit, 0)) : g.A * g.T;
	auto lo = [](unsigned u) { return __uint_as_float(u << 16); };
	auto hi = [](unsigned u) { return __uint_as_float(u & 0xffff0000u); };
	auto max3 = [](float a0, float a1, float a2) {  // the logits hold no NaN: skip fmaxf's canonicalising v_max x, x per operand
		float d;
		asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a0), "v"(a1), "v"(a2));
		return d;
	};
	for (int rj = blockIdx.x * 4 + w; rj < R; rj += gridDim.x * 4) {
		const int r = g.row_map ? g.row_map[rj] : rj;
		const int a = r / g.T, t = r - a * g.T;
		bf16* row = g.logits + (size_t)rj * g.ldl;
		// laundered per row: otherwise every lane's element indices and range masks are hoisted out of the row loop (100+ VGPRs, spills)
		int V = g.V, ldl = g.ldl, amin = g.argmax_from, lane8 = lane * 8;
		asm volatile("" : "+s"(V), "+s"(ldl), "+s"(amin), "+v"(lane8));
		V = __builtin_amdgcn_readfirstlane(V);
		ldl = __builtin_amdgcn_readfirstlane(ldl);
		amin = __builtin_amdgcn_readfirstlane(amin);
		unsigned q[NCH][4];
#pragma unroll
		for (int c = 0; c < NCH; ++c) {
			int v0 = c * 512 + lane8;
			v0 = v0 < ldl ? v0 : ldl - 8;  // clamped, not predicated: no branch around the loads
			const u32x4 v = *reinterpret_cast<const u32x4*>(row + v0);
			q[c][0] = v[0]; q[c][1] = v[1]; q[c][2] = v[2]; q[c][3] = v[3];
		}
		long long tgt = g.target ? load_tok(g.target, g.tok_bytes, (size_t)a * g.tok_ld + g.col0 + t) : -1;
		const bool ignored = (g.out_pad && g.out_pad[(size_t)a * g.C + g.col0 + t]) || (g.weight && g.weight[a] == 0.f) || tgt < 0 || tgt >= V;
		const float lt = ignored ? 0.f : (float)row[tgt];
#pragma unroll
		for (int c = 0; c < NCH; ++c) {
			if ((c + 1) * 512 > V) {  // uniform: only the pieces at or beyond V
				const int v0 = c * 512 + lane8;
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					if (v0 + 2 * k >= V) q[c][k] = NINF2;
					else if (v0 + 2 * k + 1 >= V) q[c][k] = (q[c][k] & 0xffffu) | 0xFF800000u;
				}
			}
		}
		// pass 1: maximum, first piece that attains the eligible maximum, sum of logits (label smoothing only)
		float mx = -INFINITY, bestv = -INFINITY, sl = 0.f;
		int bestc = 0;
#pragma unroll
		for (int c = 0; c < NCH; ++c) {
			float x[8];
#pragma unroll
			for (int k = 0; k < 4; ++k) { x[2 * k] = lo(q[c][k]); x[2 * k + 1] = hi(q[c][k]); }
			if (SMOOTH) {
#pragma unroll
				for (int i = 0; i < 8; ++i) sl += x[i] == -INFINITY ? 0.f : x[i];
			}
			const float cm = max3(max3(x[0], x[1], x[2]), max3(x[3], x[4], x[5]), fmaxf(x[6], x[7]));
			mx = fmaxf(mx, cm);
			float ce = cm;
			if (c == 0 && amin > 0) {  // uniform: ids below argmax_from (END) take no part in the arg-max
				ce = -INFINITY;
#pragma unroll
				for (int i = 0; i < 8; ++i) ce = lane8 + i >= amin ? fmaxf(ce, x[i]) : ce;
			}
			if (ce > bestv) { bestv = ce; bestc = c; }
			asm volatile("" : "+v"(q[c][0]), "+v"(q[c][1]), "+v"(q[c][2]), "+v"(q[c][3]));  // keep the row packed: re-convert in the next pass
			__builtin_amdgcn_sched_barrier(0);  // one piece at a time: the scheduler otherwise converts / exponentiates all 112 logits up front (200+ VGPRs)
		}
		int besti = 0x7fffffff;
		if (bestv > -INFINITY) {  // the element inside the winning piece: lowest eligible index that holds the lane's maximum
			const int v0 = bestc * 512 + lane8;
			const bf16x8 qq = *reinterpret_cast<const bf16x8*>(row + (v0 < ldl ? v0 : ldl - 8));
#pragma unroll
			for (int i = 7; i >= 0; --i)
				if (v0 + i < V && v0 + i >= amin && (float)qq[i] == bestv) besti = v0 + i;
		}
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			mx = fmaxf(mx, __shfl_xor(mx, o, 64));
			const float obv = __shfl_xor(bestv, o, 64);
			const int obi = __shfl_xor(besti, o, 64);
			if (obv > bestv || (obv == bestv && obi < besti)) { bestv = obv; besti = obi; }
			if (SMOOTH) sl += __shfl_xor(sl, o, 64);
		}
		// pass 2: p = exp(x - max) (exactly 0 beyond V), summed in fp32, parked as bf16 where the logit was
		const float nb = -mx * LOG2E;
		float se = 0.f;
#pragma unroll
		for (int c = 0; c < NCH; ++c) {
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const float p0 = __builtin_amdgcn_exp2f(fmaf(lo(q[c][k]), LOG2E, nb)), p1 = __builtin_amdgcn_exp2f(fmaf(hi(q[c][k]), LOG2E, nb));
				se += p0;
				se += p1;
				const bf16x2 pk = {(bf16)p0, (bf16)p1};
				q[c][k] = __builtin_bit_cast(unsigned, pk);
			}
			__builtin_amdgcn_sched_barrier(0);
		}
		se = wave_sum(se);
		const float lse = mx + __logf(se);
		float loss = 0.f;
		if (!ignored) {
			loss = lse - lt;
			if (SMOOTH) loss = (1.f - g.smoothing) * loss + g.smoothing * (lse - sl / (float)V);
		}
		if (lane == 0) {
			g.row_loss[r] = loss;
			g.row_argmax[r] = besti;
			if (g.row_correct) g.row_correct[r] = (!ignored && besti == (int)tgt) ? 1 : 0;
		}
		if (g.write_grad) {
			float sc = 0.f;
			if (!ignored) {
				sc = g.grad_scale;
				if (g.grad_scale_dev) sc *= g.grad_scale_dev[0];
				if (g.weight) sc *= g.weight[a];
				if (g.basis) sc /= g.basis[a / g.group_rows];
			}
			// gradient = (p / sum - smoothing / V - [v == target] (1 - smoothing)) * sc
			const float k1 = sc / se, k0 = -(g.smoothing / (float)V) * sc, kt = (1.f - g.smoothing) * sc;
#pragma unroll
			for (int c = 0; c < NCH; ++c) {
				const int v0 = c * 512 + lane8;
				if (v0 < ldl) {
					float gr[8];
#pragma unroll
					for (int k = 0; k < 4; ++k) { gr[2 * k] = fmaf(lo(q[c][k]), k1, k0); gr[2 * k + 1] = fmaf(hi(q[c][k]), k1, k0); }
					if (SMOOTH && (c + 1) * 512 > V) {  // uniform; the pad columns get exact zeros
#pragma unroll
						for (int i = 0; i < 8; ++i) gr[i] = v0 + i < V ? gr[i] : 0.f;
					}
					const unsigned j = (unsigned)((int)tgt - v0);
					if (j < 8u && !ignored) {
#pragma unroll
						for (int i = 0; i < 8; ++i) gr[i] -= (unsigned)i == j ? kt : 0.f;
					}
					bf16x8 o = {(bf16)gr[0], (bf16)gr[1], (bf16)gr[2], (bf16)gr[3], (bf16)gr[4], (bf16)gr[5], (bf16)gr[6], (bf16)gr[7]};
					*reinterpret_cast<bf16x8*>(row + v0) = o;
				}
				__builtin_amdgcn_sched_barrier(0);
			}
		}
	}
}

// One block per micro-batch group: deterministic sums.
//   basis[g]   = sum_a w_a * (#unpadded tokens of a)            (w_a = 1 without weights)
//   loss[g]    = sum_a w_a * sum_t row_loss[a][t]
//   correct[g] = #correct tokens,  tokens[g] = #unpadded tokens
__global__ __launch_bounds__(256) void group_reduce_kernel(const float* __restrict__ row_loss, const uint8_t* __restrict__ row_correct, const uint8_t* __restrict__ out_pad,
                                                           const float* __restrict__ weight, float* __restrict__ basis, float* __restrict__ loss, float* __restrict__ correct,
                                                           float* __restrict__ tokens, int A, int T, int C, int col0, int group_rows) {
	__shared__ double red[4][4];
	const int gidx = blockIdx.x;
	const int a0 = gidx * group_rows, a1 = min(A, a0 + group_rows);
	double sb = 0, sl = 0, sc = 0, st = 0;
	for (int a = a0 + threadIdx.x; a < a1; a += blockDim.x) {
		const float wa = weight ? weight[a] : 1.f;
		int cnt = 0, cor = 0;
		float ls = 0.f;
		for (int t = 0; t < T; ++t) {
			const bool pad = (out_pad && out_pad[(size_t)a * C + col0 + t]) || (weight && wa == 0.f);
			if (!pad) ++cnt;
			if (row_loss) ls += row_loss[(size_t)a * T + t];
			if (row_correct) cor += row_correct[(size_t)a * T + t];
		}
		sb += (double)wa * cnt;
		sl += (double)wa * ls;
		sc += cor;
		st += cnt;
	}
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		sb += __shfl_xor(sb, o, 64); sl += __shfl_xor(sl, o, 64); sc += __shfl_xor(sc, o, 64); st += __shfl_xor(st, o, 64);
	}
	if (lane == 0) { red[w][0] = sb; red[w][1] = sl; red[w][2] = sc; red[w][3] = st; }
	__syncthreads();
	if (threadIdx.x == 0) {
		double r0 = 0, r1 = 0, r2 = 0, r3 = 0;
		for (int i = 0; i < 4; ++i) { r0 += red[i][0]; r1 += red[i][1]; r2 += red[i][2]; r3 += red[i][3]; }
		if (basis) basis[gidx] = (float)r0;
		if (loss) loss[gidx] = (float)r1;
		if (correct) correct[gidx] = (float)r2;
		if (tokens) tokens[gidx] = (float)r3;
	}
}

// Compaction of the output positions that count, in two small launches of ceil(R / 1024) workgroups (one token row r = a * T + t per thread; a
// single workgroup walking all rows spent 150 us waiting on its own serial loads).  Row r counts if neither its target position is padding nor
// its sample's weight zero.  Pass 1: per-workgroup counts.  Pass 2: rows[j] = the j-th such r (ascending), src_rows[j] = its row
// a * S + (S - T) + t of the [A * S] hidden-state matrix, dst_of[m] (all A * S input rows) = j or -1, *count; the per-row loss outputs of the rows that
// do NOT count are zeroed (the compacted cross-entropy never visits them).
__device__ __forceinline__ bool row_counts(const uint8_t* out_pad, const float* weight, int T, int C, int col0, int r) {
	const int a = r / T, t = r - a * T;
	return !((out_pad && out_pad[(size_t)a * C + col0 + t]) || (weight && weight[a] == 0.f));
}
__device__ __forceinline__ int block_scan_1024(int v, int* part, int& total) {  // exclusive prefix of v over the workgroup's 1024 threads
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	int incl = v;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const int n = __shfl_up(incl, o, 64);
		if (lane >= o) incl += n;
	}
	if (lane == 63) part[w] = incl;
	__syncthreads();
	int base = 0;
	total = 0;
	for (int i = 0; i < 16; ++i) {
		const int c = part[i];
		if (i < w) base += c;
		total += c;
	}
	__syncthreads();
	return base + incl - v;
}
__global__ __launch_bounds__(1024) void compact_count_kernel(const uint8_t* __restrict__ out_pad, const float* __restrict__ weight, int A, int T, int C, int col0,
                                                             int* __restrict__ blk_count) {
	__shared__ int part[16];
	const int r = blockIdx.x * 1024 + threadIdx.x;
	const int f = (r < A * T && row_counts(out_pad, weight, T, C, col0, r)) ? 1 : 0;
	int total;
	(void)block_scan_1024(f, part, total);
	if (threadIdx.x == 0) blk_count[blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void compact_write_kernel(const uint8_t* __restrict__ out_pad, const float* __restrict__ weight, int A, int T, int C, int col0, int S,
                                                             const int* __restrict__ blk_count, int* __restrict__ rows, int* __restrict__ src_rows,
                                                             int* __restrict__ dst_of, int* __restrict__ count, float* __restrict__ row_loss, int* __restrict__ row_argmax,
                                                             uint8_t* __restrict__ row_correct, const int* __restrict__ seq_start) {
	__shared__ int part[16];
	__shared__ int base_s;
	const int tid = threadIdx.x, R = A * T, off = S - T;
	if (tid < 64) {  // rows that counted in the workgroups before this one
		int sum = 0;
		for (int b = tid; b < (int)blockIdx.x; b += 64) sum += blk_count[b];
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
		if (tid == 0) base_s = sum;
	}
	const int r = blockIdx.x * 1024 + tid;
	const bool in = r < R;
	const bool f = in && row_counts(out_pad, weight, T, C, col0, r);
	int total;
	const int pre = block_scan_1024(f ? 1 : 0, part, total);  // (its barriers also publish base_s)
	if (in) {
		const int a = r / T, t = r - a * T;
		const int row0 = seq_start ? seq_start[a] : a * S;  // packed hidden-state rows: sequence a starts at seq_start[a]
		const int m = row0 + off + t;
		if (f) {
			const int j = base_s + pre;
			rows[j] = r;
			src_rows[j] = m;
			dst_of[m] = j;
		} else {
			if (!seq_start) dst_of[m] = -1;  // (packed: that row may not exist, or belong to the next sequence -- dst_of was preset to -1 by the caller)
			if (row_loss) row_loss[r] = 0.f;
			if (row_argmax) row_argmax[r] = 0;
			if (row_correct) row_correct[r] = 0;
		}
		if (t == 0 && !seq_start)
			for (int s = 0; s < off; ++s) dst_of[a * S + s] = -1;  // the positions in front of the output window have no upstream row
	}
	if (blockIdx.x == gridDim.x - 1 && tid == 0) *count = base_s + total;
}

// Packed-row layout of a batch: sequence a keeps its positions 0 .. len[a] - 1, len[a] = 1 + the last position that is not key-padded (padding is a
// suffix of every sequence: embedding_decoder.py:696-712), and starts at row start[a] = sum of the lengths before it.  Two launches of
// ceil(A / 1024) workgroups, one sequence per thread; total[0] = the number of rows, total[1 ..] scratch.
__device__ __forceinline__ int seq_valid_len(const uint8_t* key_pad, int S, int a) {
	int n = 1;
	for (int s = 1; s < S; ++s)
		if (!key_pad[(size_t)a * S + s]) n = s + 1;
	return n;
}
__global__ __launch_bounds__(1024) void seq_count_kernel(const uint8_t* __restrict__ key_pad, int A, int S, int* __restrict__ blk_count) {
	__shared__ int part[16];
	const int a = blockIdx.x * 1024 + threadIdx.x;
	int total;
	(void)block_scan_1024(a < A ? seq_valid_len(key_pad, S, a) : 0, part, total);
	if (threadIdx.x == 0) blk_count[blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void seq_layout_kernel(const uint8_t* __restrict__ key_pad, int A, int S, const int* __restrict__ blk_count, int* __restrict__ start,
                                                          int* __restrict__ len, int* __restrict__ total_out) {
	__shared__ int part[16];
	__shared__ int base_s;
	const int tid = threadIdx.x;
	if (tid < 64) {
		int sum = 0;
		for (int b = tid; b < (int)blockIdx.x; b += 64) sum += blk_count[b];
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
		if (tid == 0) base_s = sum;
	}
	const int a = blockIdx.x * 1024 + tid;
	const int n = a < A ? seq_valid_len(key_pad, S, a) : 0;
	int total;
	const int pre = block_scan_1024(n, part, total);
	if (a < A) {
		start[a] = base_s + pre;
		len[a] = n;
	}
	if (blockIdx.x == gridDim.x - 1 && tid == 0) *total_out = base_s + total;
}

}  // namespace

extern "C" int novic_seq_layout(const uint8_t* key_pad, int A, int S, int* seq_start, int* seq_len, int* total, hipStream_t stream) {
	NOVIC_CHECK(key_pad && seq_start && seq_len && total, "novic_seq_layout: null pointer");
	NOVIC_CHECK(S >= 1 && (uint64_t)(A > 0 ? A : 0) * S < 0x7FFFFFFFull, "novic_seq_layout: bad shape");
	if (A <= 0) return 0;
	const int nblk = (A + 1023) / 1024;
	hipLaunchKernelGGL(seq_count_kernel, dim3(nblk), dim3(1024), 0, stream, key_pad, A, S, total + 1);
	hipLaunchKernelGGL(seq_layout_kernel, dim3(nblk), dim3(1024), 0, stream, key_pad, A, S, total + 1, seq_start, seq_len, total);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_compact_rows(const uint8_t* out_pad, const float* weight, int A, int T, int C, int col0, int S, int* rows, int* src_rows, int* dst_of, int* count,
                                  float* row_loss, int* row_argmax, uint8_t* row_correct, const int* seq_start, hipStream_t stream) {
	NOVIC_CHECK(rows && src_rows && dst_of && count, "novic_compact_rows: null output");
	NOVIC_CHECK(T >= 1 && S >= T && col0 >= 0 && col0 + T <= C, "novic_compact_rows: bad shape");
	NOVIC_CHECK((uint64_t)(A > 0 ? A : 0) * S < 0x7FFFFFFFull, "novic_compact_rows: A * S must fit 31 bits");
	if (A <= 0) return 0;
	// count[0] = the number of rows that count; count[1 .. nblk] = scratch (per-workgroup counts of pass 1)
	const int R = A * T, nblk = (R + 1023) / 1024;
	int* scratch = count + 1;
	if (seq_start) NOVIC_CHECK(hipMemsetAsync(dst_of, 0xFF, (size_t)A * S * sizeof(int), stream) == hipSuccess, "novic_compact_rows: memset failed");  // all -1
	hipLaunchKernelGGL(compact_count_kernel, dim3(nblk), dim3(1024), 0, stream, out_pad, weight, A, T, C, col0, scratch);
	hipLaunchKernelGGL(compact_write_kernel, dim3(nblk), dim3(1024), 0, stream, out_pad, weight, A, T, C, col0, S, scratch, rows, src_rows, dst_of, count, row_loss,
	                   row_argmax, row_correct, seq_start);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_build_padding(const uint8_t* target_padding, int tpad_ld, const float* weight, uint8_t* key_pad, uint8_t* out_pad, int A, int C, int P,
                                   int num_end_loss, hipStream_t stream) {
	NOVIC_CHECK(key_pad && out_pad, "novic_build_padding: null output");
	NOVIC_CHECK(C >= 1 && P >= 1 && num_end_loss >= 1, "novic_build_padding: bad shape");
	if (A <= 0) return 0;
	const int n = A * (P + C - 1);
	int grid = (n + 255) / 256;
	if (grid > 2048) grid = 2048;
	hipLaunchKernelGGL(build_padding_kernel, dim3(grid), dim3(256), 0, stream, target_padding, weight, key_pad, out_pad, A, C, P, num_end_loss, tpad_ld);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_cross_entropy(void* logits_bf16, int ldl, int V, int A, int T, int C, int col0, const void* target, int tok_bytes, int tok_ld, const uint8_t* out_pad,
                                   const float* weight, const float* basis, int group_rows, float grad_scale, const float* grad_scale_dev, float label_smoothing, int write_grad,
                                   float* row_loss, int* row_argmax, uint8_t* row_correct, int argmax_from, const int* row_map, const int* row_limit,
                                   hipStream_t stream) {
	NOVIC_CHECK(logits_bf16 && row_loss && row_argmax, "novic_cross_entropy: null pointer");
	NOVIC_CHECK(ldl % 8 == 0 && ldl >= V && V >= 1, "novic_cross_entropy: ldl must be a multiple of 8 and >= V");
	NOVIC_CHECK(tok_bytes == 4 || tok_bytes == 8, "novic_cross_entropy: tok_bytes must be 4 or 8");
	NOVIC_CHECK(T >= 1 && col0 >= 0 && col0 + T <= C && group_rows >= 1, "novic_cross_entropy: bad column window");
	if (A <= 0) return 0;
	CeArgs g = {(bf16*)logits_bf16, ldl, V, A, T, C, col0, target, tok_bytes, tok_ld, out_pad, weight, basis, group_rows, grad_scale, grad_scale_dev, label_smoothing, write_grad,
	            row_loss, row_argmax, row_correct, argmax_from, row_map, row_limit};
	int grid = (A * T + 3) / 4;
	if (grid > 16384) grid = 16384;
	const int nch = (ldl + 511) / 512;
#define NOVIC_CE_ROWS(NCH)                                                                                       \
	do {                                                                                                        \
		if (label_smoothing > 0.f) hipLaunchKernelGGL((ce_rows_kernel<NCH, true>), dim3(grid), dim3(256), 0, stream, g); \
		else hipLaunchKernelGGL((ce_rows_kernel<NCH, false>), dim3(grid), dim3(256), 0, stream, g);              \
	} while (0)
	if (nch <= 4) NOVIC_CE_ROWS(4);
	else if (nch <= 8) NOVIC_CE_ROWS(8);
	else if (nch <= 14) NOVIC_CE_ROWS(14);
	else if (nch <= 16) NOVIC_CE_ROWS(16);
	else hipLaunchKernelGGL(ce_kernel, dim3(grid), dim3(256), 0, stream, g);
#undef NOVIC_CE_ROWS
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_loss_group_reduce(const float* row_loss, const uint8_t* row_correct, const uint8_t* out_pad, const float* weight, float* basis, float* loss,
                                       float* correct, float* tokens, int A, int T, int C, int col0, int group_rows, hipStream_t stream) {
	NOVIC_CHECK(T >= 1 && col0 >= 0 && col0 + T <= C && group_rows >= 1, "novic_loss_group_reduce: bad column window");
	if (A <= 0) return 0;
	const int groups = (A + group_rows - 1) / group_rows;
	hipLaunchKernelGGL(group_reduce_kernel, dim3(groups), dim3(256), 0, stream, row_loss, row_correct, out_pad, weight, basis, loss, correct, tokens, A, T, C, col0,
	                   group_rows);
	NOVIC_LAUNCH_CHECK();
	return 0;
}
