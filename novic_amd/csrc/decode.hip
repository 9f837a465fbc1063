// Per-step decode kernels: greedy token selection with running scores, and the beam-search step
// (temperature, finished-beam rule, log-softmax, running score, first-step END ban, length normalisation, top-H with a defined
// tie-break, history / padding / length reordering) -- reference embedding_decoder.py:792-820 (greedy), :905-978 (beam).
// No host synchronisation per step: "all finished" is recorded per step in a device counter that the host reads once at the end.
// Tie-break (the reference leaves torch.topk's unspecified): lowest flat candidate index h*V + v first.
#include "common.hpp"
#include "novic_hip.h"

namespace {

__device__ __forceinline__ void store_tok(void* tok, int tok_bytes, size_t i, long long v) {
	if (tok_bytes == 8) ((long long*)tok)[i] = v;
	else ((int*)tok)[i] = (int)v;
}
__device__ __forceinline__ long long load_tok(const void* tok, int tok_bytes, size_t i) {
	return tok_bytes == 8 ? ((const long long*)tok)[i] : (long long)((const int*)tok)[i];
}

// ---------------------------------------------------------------------------------------------------------
// greedy: one workgroup per sample
// ---------------------------------------------------------------------------------------------------------
struct GreedyArgs {
	const bf16* logits;  // [B][ldl] logits of the position being predicted
	int ldl, V, B, G, step;  // step = C (1-based)
	void* ids; int tok_bytes;  // [B][G]
	uint8_t* pad;       // [B][G]
	float* alive;       // [B] 1 = still generating (doubles as the next forward's target_weight), 0 = finished
	float* score;       // [B] running sum of log-softmax(logits / temperature)[chosen] over unpadded positions
	float* nll;         // [B] running sum of -log-softmax(logits)[chosen] (+ label smoothing)
	float* count;       // [B] number of unpadded positions
	int* active;        // [G] number of samples still alive after each step
	float* step_logits; // optional [B][G][V] fp32 copy of the logits used (collect_logits)
	float inv_temp, smoothing;
	novic_next_embed_t next;  // optional (x_next != null): the NEXT step's input row, written here instead of by a launch of its own
};

// one workgroup (256 threads) per sample: 16-byte loads, in-thread online soft-max statistics, one block reduction
__global__ __launch_bounds__(256) void greedy_step_kernel(const GreedyArgs g) {
	__shared__ float s_mx[4], s_se[4], s_set[4], s_sl[4], s_bv[4];
	__shared__ int s_bi[4];
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const int c = g.step - 1;
	for (int b = blockIdx.x; b < g.B; b += gridDim.x) {
		const bf16* row = g.logits + (size_t)b * g.ldl;
		const bool was_alive = g.alive[b] != 0.f;
		const int from = (g.step == 1) ? 1 : 0;  // the first token may not be END (:803-804)
		float mx = -INFINITY, se = 0.f, set = 0.f, sl = 0.f, bestv = -INFINITY;
		int besti = 0x7fffffff;
		if (g.V <= 8192) {
			// the thread's (up to four) 16-byte chunks of the row stay in registers: the maximum first, then both exponential sums against it -- no
			// running-maximum chain (27 dependent compare / exp steps per thread in the online form below: 11.3 us per launch at 6912 logits)
			bf16x8 xs[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const int v0 = (tid + (k << 8)) * 8;
				if (v0 < g.V) xs[k] = *reinterpret_cast<const bf16x8*>(row + v0);
			}
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const int v0 = (tid + (k << 8)) * 8;
				if (v0 < g.V) {
#pragma unroll
					for (int e = 0; e < 8; ++e) {
						const int v = v0 + e;
						if (v < g.V) {
							const float x = (float)xs[k][e];
							if (g.step_logits) g.step_logits[((size_t)b * g.G + c) * g.V + v] = x;
							if (v >= from && x > bestv) { bestv = x; besti = v; }
							mx = fmaxf(mx, x);
							sl += x;
						}
					}
				}
			}
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const int v0 = (tid + (k << 8)) * 8;
				if (v0 < g.V) {
#pragma unroll
					for (int e = 0; e < 8; ++e)
						if (v0 + e < g.V) {
							const float d = (float)xs[k][e] - mx;
							se += __expf(d);
							set += __expf(d * g.inv_temp);
						}
				}
			}
		} else
		for (int v0 = tid * 8; v0 < g.V; v0 += 256 * 8) {
			const bf16x8 xs = *reinterpret_cast<const bf16x8*>(row + v0);  // ldl is a multiple of 8: the row's padding makes this load legal
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				const int v = v0 + k;
				if (v >= g.V) break;
				const float x = (float)xs[k];
				if (g.step_logits) g.step_logits[((size_t)b * g.G + c) * g.V + v] = x;
				if (v >= from && x > bestv) { bestv = x; besti = v; }
				if (x > mx) {
					const float sc = __expf(mx - x), sct = __expf((mx - x) * g.inv_temp);
					se = se * sc + 1.f; set = set * sct + 1.f; mx = x;
				} else {
					se += __expf(x - mx); set += __expf((x - mx) * g.inv_temp);
				}
				sl += x;
			}
		}
		auto merge = [&](float omx, float ose, float oset, float osl, float obv, int obi) {
			const float nm = fmaxf(mx, omx);
			if (nm != -INFINITY) {
				se = se * __expf(mx - nm) + ose * __expf(omx - nm);
				set = set * __expf((mx - nm) * g.inv_temp) + oset * __expf((omx - nm) * g.inv_temp);
			}
			mx = nm;
			sl += osl;
			if (obv > bestv || (obv == bestv && obi < besti)) { bestv = obv; besti = obi; }
		};
#pragma unroll
		for (int o = 32; o > 0; o >>= 1)
			merge(__shfl_xor(mx, o, 64), __shfl_xor(se, o, 64), __shfl_xor(set, o, 64), __shfl_xor(sl, o, 64), __shfl_xor(bestv, o, 64), __shfl_xor(besti, o, 64));
		__syncthreads();  // previous sample's shared values are consumed
		if (lane == 0) { s_mx[w] = mx; s_se[w] = se; s_set[w] = set; s_sl[w] = sl; s_bv[w] = bestv; s_bi[w] = besti; }
		__syncthreads();
		if (tid == 0) {
			for (int k = 1; k < 4; ++k) merge(s_mx[k], s_se[k], s_set[k], s_sl[k], s_bv[k], s_bi[k]);
			const float lse = mx + __logf(se), lse_t = mx * g.inv_temp + __logf(set);
			if (besti == 0x7fffffff) besti = 0;  // no logit of the row compared greater than -inf (all NaN / -inf): the sample emits END
			g.pad[(size_t)b * g.G + c] = was_alive ? 0 : 1;
			store_tok(g.ids, g.tok_bytes, (size_t)b * g.G + c, besti);
			if (was_alive) {
				g.score[b] += bestv * g.inv_temp - lse_t;
				float l = lse - bestv;
				if (g.smoothing > 0.f) l = (1.f - g.smoothing) * l + g.smoothing * (lse - sl / (float)g.V);
				g.nll[b] += l;
				g.count[b] += 1.f;
			}
			const bool still = was_alive && besti != 0;
			g.alive[b] = still ? 1.f : 0.f;
			if (still) atomicAdd(g.active + c, 1);
			s_bi[0] = besti;
		}
		if (g.next.x_next) {  // the next step's input row of this sample (novic_decode_embed's arithmetic): W_tok[token] + pos_row
			__syncthreads();
			// a row of NaN / -inf logits offers no arg-max (besti stays 0x7fffffff): clamp as novic_decode_embed does, the read must stay inside W_tok
			const int tok = min(max(s_bi[0], 0), g.V - 1), E = g.next.E;
			for (int e = tid * 4; e < E; e += 256 * 4) {
				const f32x4 tv = *reinterpret_cast<const f32x4*>(g.next.wtok + (size_t)tok * E + e);
				const f32x4 pv = *reinterpret_cast<const f32x4*>(g.next.pos_row + e);
				*reinterpret_cast<f32x4*>(g.next.x_next + (size_t)b * E + e) = (f32x4){tv[0] + pv[0], tv[1] + pv[1], tv[2] + pv[2], tv[3] + pv[3]};
			}
		}
	}
}

__global__ void greedy_finalize_kernel(void* ids, int tok_bytes, const uint8_t* pad, float* score, const float* count, int B, int G, float alpha) {
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * G; i += gridDim.x * blockDim.x) {
		if (pad[i]) store_tok(ids, tok_bytes, i, 0);
		if (alpha != 0.f && i < B) score[i] *= powf(fmaxf(count[i], 1.f), -alpha);
	}
}

// ---------------------------------------------------------------------------------------------------------
// beam: one workgroup (256 threads) per sample
// ---------------------------------------------------------------------------------------------------------
struct BeamArgs {
	const bf16* logits;  // [B*H][ldl]
	int ldl, V, B, H, G, step;
	const void* ids_in; void* ids_out; int tok_bytes;  // [B][H][G]
	const uint8_t* pad_in; uint8_t* pad_out;           // [B][H][G]
	const float* score_in; float* score_out;           // [B][H] raw running scores
	float* score_normed;                               // [B][H] ranking scores (== raw when alpha == 0)
	const float* len_in; float* len_out;               // [B][H]
	int* active;                                       // [G] beams still unfinished after each step (all samples)
	int* src_out;                                      // optional [B][H]: which old beam each new beam continues (KV-cache reorder)
	float inv_temp, alpha;
	novic_next_embed_t next;  // optional (x_next != null): the NEXT step's input rows (and, origin_out != null, the K/V origin table) written here
};

// A sample whose logits are all NaN / -inf offers fewer than H candidates: the unfilled picks stay 0x7fffffff and would index ids / W_tok / the origin table far out
// of range.  Such a pick becomes (beam h, END): memory-safe, and the beam ends.  (Called by every thread after the selection's last barrier.)
__device__ __forceinline__ void sanitize_picks(int* pick_idx, int H, int V, int tid) {
	if (tid < H && (unsigned)pick_idx[tid] >= (unsigned)(H * V)) pick_idx[tid] = tid * V;
	__syncthreads();
}

// What novic_decode_embed and novic_kv_origin_update would do in two launches of their own after a beam step, done by the step's workgroup (everything it
// needs is per sample): x_next[(b, h')] = W_tok[token chosen for beam h'] + pos_row, and the origin rows of the sample's new beams.
__device__ __forceinline__ void beam_next_inputs(const BeamArgs& g, int b, const int* pick_idx, int tid) {
	const novic_next_embed_t& n = g.next;
	if (!n.x_next) return;
	const int H = g.H, V = g.V, E = n.E;
	for (int i = tid * 4; i < H * E; i += 256 * 4) {
		const int hn = i / E, e = i - hn * E;
		const int tok = pick_idx[hn] % V;
		const f32x4 tv = *reinterpret_cast<const f32x4*>(n.wtok + (size_t)tok * E + e);
		const f32x4 pv = *reinterpret_cast<const f32x4*>(n.pos_row + e);
		*reinterpret_cast<f32x4*>(n.x_next + ((size_t)b * H + hn) * E + e) = (f32x4){tv[0] + pv[0], tv[1] + pv[1], tv[2] + pv[2], tv[3] + pv[3]};
	}
	if (n.origin_out) {
		for (int i = tid; i < H * n.npos; i += 256) {
			const int hn = i / n.npos, gp = i - hn * n.npos;
			const int a = b * H + hn, sa = b * H + pick_idx[hn] / V;
			n.origin_out[(size_t)a * g.G + gp] = gp == n.npos - 1 ? sa : n.origin_in[(size_t)sa * g.G + gp];
		}
	}
}

// Top-H of the H*V candidates WITHOUT H full scans: every thread owns a fixed subset of the candidates (16-byte chunks tid, tid + 256, ... of
// every beam's row) and caches the best one it still has to offer; a selection round is a block arg-max over the 256 cached offers, after which
// only the winner's wave re-evaluates the winner's subset (one candidate per lane) for its next offer.  Order: value descending, flat index
// h*V + v ascending (the defined tie-break), exactly as a stable sort would give.
__global__ __launch_bounds__(256) void beam_step_kernel(const BeamArgs g) {
	constexpr int MAXH = 32;
	__shared__ float s_lse[MAXH], s_add[MAXH], s_scale[MAXH];
	__shared__ uint8_t s_fin[MAXH];
	__shared__ float s_val[4], s_raw[4];
	__shared__ int s_idx[4], s_who[4];
	__shared__ float s_pick_val[MAXH], s_pick_raw[MAXH];
	__shared__ int s_pick_idx[MAXH], s_pick_who[MAXH];
	const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const int C = g.step, c = C - 1, H = g.H, V = g.V;
	const bf16* lg = g.logits + (size_t)b * H * g.ldl;
	const int nchunk = (V + 7) >> 3;            // 16-byte chunks per row (ldl is a multiple of 8: the last chunk may run into the row padding)
	const int per_thread = (nchunk + 255) >> 8;  // chunks of one row a thread owns

	auto cand = [&](int h, int v, float x, float& raw) -> float {  // ranking value of candidate (h, v) with logit x
		if (s_fin[h] && v > 0) { raw = -INFINITY; return -INFINITY; }
		if (C == 1 && h == 0 && v == 0) { raw = -INFINITY; return -INFINITY; }
		raw = (x * g.inv_temp - s_lse[h]) + s_add[h];
		return raw * s_scale[h];
	};
	auto better = [](float v, int i, float bv, int bi) { return bi == 0x7fffffff || v > bv || (v == bv && i < bi); };
	float my_val = -INFINITY, my_raw = -INFINITY;
	int my_idx = 0x7fffffff;

	if (per_thread <= 4) {
		// Rows of up to 8192 logits: a thread keeps its (up to four) 16-byte chunks of the row in registers -- ONE pass over memory per row.  All 256
		// threads work on every row: maximum, then the exponentials (no running-maximum chain: the one-wave-per-beam online form below spent 108
		// dependent compare / exp steps per lane), one block merge per row, and the row's candidates are ranked from the same registers.
		__shared__ float s_pm[MAXH][4], s_ps[MAXH][4];
		// per-beam state first, by H threads in parallel (one global round trip for the whole kernel instead of one per row on thread 0)
		if (tid < H) {
			s_fin[tid] = g.pad_in[((size_t)b * H + tid) * g.G + c] != 0;
			s_add[tid] = g.score_in[b * H + tid];
			s_scale[tid] = (g.alpha != 0.f) ? powf(fmaxf(g.len_in[b * H + tid], 1.f), -g.alpha) : 1.f;
		}
		bf16x8 nx[4];  // the next row's chunks, requested one row ahead
		auto load_row = [&](int h, bf16x8 (&dst)[4]) {
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const int ch = tid + (k << 8);
				if (ch < nchunk) dst[k] = *reinterpret_cast<const bf16x8*>(lg + (size_t)h * g.ldl + ch * 8);
			}
		};
		load_row(0, nx);
		__syncthreads();
		for (int h = 0; h < H; ++h) {
			const bool fin = s_fin[h] != 0;
			bf16x8 xs[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) xs[k] = nx[k];
			if (h + 1 < H) load_row(h + 1, nx);
			float mx = -INFINITY, se = 0.f;
			if (!fin) {
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					const int ch = tid + (k << 8);
					if (ch < nchunk) {
#pragma unroll
						for (int e = 0; e < 8; ++e)
							if (ch * 8 + e < V) mx = fmaxf(mx, (float)xs[k][e] * g.inv_temp);
					}
				}
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					const int ch = tid + (k << 8);
					if (ch < nchunk) {
#pragma unroll
						for (int e = 0; e < 8; ++e)
							if (ch * 8 + e < V) se += __expf((float)xs[k][e] * g.inv_temp - mx);
					}
				}
#pragma unroll
				for (int o = 32; o > 0; o >>= 1) {
					const float omx = __shfl_xor(mx, o, 64), ose = __shfl_xor(se, o, 64);
					const float nm = fmaxf(mx, omx);
					if (nm != -INFINITY) se = se * __expf(mx - nm) + ose * __expf(omx - nm);
					mx = nm;
				}
			}
			if (lane == 0) { s_pm[h][w] = mx; s_ps[h][w] = se; }
			__syncthreads();
			if (tid == 0) {
				float m = s_pm[h][0], sacc = s_ps[h][0];
				for (int k = 1; k < 4; ++k) {
					const float om = s_pm[h][k], os = s_ps[h][k];
					const float nm = fmaxf(m, om);
					if (nm != -INFINITY) sacc = sacc * __expf(m - nm) + os * __expf(om - nm);
					m = nm;
				}
				s_lse[h] = fin ? (float)lg[(size_t)h * g.ldl] * g.inv_temp : m + __logf(sacc);
			}
			__syncthreads();
			// first offers: best candidate of the thread's own chunks of this row
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const int ch = tid + (k << 8);
				if (ch < nchunk) {
#pragma unroll
					for (int e = 0; e < 8; ++e) {
						const int v = ch * 8 + e;
						if (v < V) {
							float raw;
							const float val = cand(h, v, (float)xs[k][e], raw);
							if (better(val, h * V + v, my_val, my_idx)) { my_val = val; my_idx = h * V + v; my_raw = raw; }
						}
					}
				}
			}
		}
	} else {
	// per-beam log-sum-exp of logits / temperature (finished beams: only END survives, log-prob 0): one wave per beam, 16-byte loads
	for (int h = w; h < H; h += 4) {
		const bool fin = g.pad_in[((size_t)b * H + h) * g.G + c] != 0;
		float mx = -INFINITY, se = 0.f;
		if (!fin) {
			for (int ch = lane; ch < nchunk; ch += 64) {
				const bf16x8 xs = *reinterpret_cast<const bf16x8*>(lg + (size_t)h * g.ldl + ch * 8);
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					if (ch * 8 + k >= V) break;
					const float x = (float)xs[k] * g.inv_temp;
					if (x > mx) { se = se * __expf(mx - x) + 1.f; mx = x; }
					else se += __expf(x - mx);
				}
			}
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) {
				const float omx = __shfl_xor(mx, o, 64), ose = __shfl_xor(se, o, 64);
				const float nm = fmaxf(mx, omx);
				if (nm != -INFINITY) se = se * __expf(mx - nm) + ose * __expf(omx - nm);
				mx = nm;
			}
		}
		if (lane == 0) {
			s_fin[h] = fin;
			s_lse[h] = fin ? (float)lg[(size_t)h * g.ldl] * g.inv_temp : mx + __logf(se);
			s_add[h] = g.score_in[b * H + h];
			s_scale[h] = (g.alpha != 0.f) ? powf(fmaxf(g.len_in[b * H + h], 1.f), -g.alpha) : 1.f;
		}
	}
	__syncthreads();

	// first offers: best candidate of the thread's own subset
	for (int h = 0; h < H; ++h)
		for (int k = 0; k < per_thread; ++k) {
			const int ch = tid + (k << 8);
			if (ch >= nchunk) break;
			const bf16x8 xs = *reinterpret_cast<const bf16x8*>(lg + (size_t)h * g.ldl + ch * 8);
#pragma unroll
			for (int e = 0; e < 8; ++e) {
				const int v = ch * 8 + e;
				if (v >= V) break;
				float raw;
				const float val = cand(h, v, (float)xs[e], raw);
				if (better(val, h * V + v, my_val, my_idx)) { my_val = val; my_idx = h * V + v; my_raw = raw; }
			}
		}

	}

	for (int r = 0; r < H; ++r) {
		// block arg-max over the cached offers
		float bv = my_val, braw = my_raw;
		int bi = my_idx, who = tid;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			const float ov = __shfl_xor(bv, o, 64), orw = __shfl_xor(braw, o, 64);
			const int oi = __shfl_xor(bi, o, 64), ow = __shfl_xor(who, o, 64);
			if (oi != 0x7fffffff && better(ov, oi, bv, bi)) { bv = ov; bi = oi; braw = orw; who = ow; }
		}
		if (lane == 0) { s_val[w] = bv; s_idx[w] = bi; s_raw[w] = braw; s_who[w] = who; }
		__syncthreads();
		if (tid == 0) {
			float fv = s_val[0], fr = s_raw[0];
			int fi = s_idx[0], fw = s_who[0];
			for (int k = 1; k < 4; ++k)
				if (s_idx[k] != 0x7fffffff && better(s_val[k], s_idx[k], fv, fi)) { fv = s_val[k]; fi = s_idx[k]; fr = s_raw[k]; fw = s_who[k]; }
			s_pick_val[r] = fv; s_pick_idx[r] = fi; s_pick_raw[r] = fr; s_pick_who[r] = fw;
		}
		__syncthreads();
		// the winner's wave rebuilds the winner's offer: best of its subset strictly AFTER the pick in (value desc, index asc) order
		const int winner = s_pick_who[r];
		if (r + 1 < H && (winner >> 6) == w) {
			const float pv = s_pick_val[r];
			const int pi = s_pick_idx[r], wt = winner;
			float nv = -INFINITY, nraw = -INFINITY;
			int ni = 0x7fffffff;
			const int per_row = per_thread * 8;
			for (int e = lane; e < H * per_row; e += 64) {
				const int h = e / per_row, rem = e - h * per_row;
				const int ch = wt + ((rem >> 3) << 8), v = ch * 8 + (rem & 7);
				if (ch >= nchunk || v >= V) continue;
				float raw;
				const float val = cand(h, v, (float)lg[(size_t)h * g.ldl + v], raw);
				const int i = h * V + v;
				const bool after = (val < pv) || (val == pv && i > pi);
				if (after && better(val, i, nv, ni)) { nv = val; ni = i; nraw = raw; }
			}
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) {
				const float ov = __shfl_xor(nv, o, 64), orw = __shfl_xor(nraw, o, 64);
				const int oi = __shfl_xor(ni, o, 64);
				if (oi != 0x7fffffff && better(ov, oi, nv, ni)) { nv = ov; ni = oi; nraw = orw; }
			}
			if (tid == wt) { my_val = nv; my_idx = ni; my_raw = nraw; }
		}
	}
	__syncthreads();
	sanitize_picks(s_pick_idx, H, V, tid);

	// reorder histories (ping-pong buffers), append tokens, update padding / scores / lengths
	for (int i = tid; i < H * g.G; i += 256) {
		const int hn = i / g.G, col = i - hn * g.G;
		const int src = s_pick_idx[hn] / V, tok = s_pick_idx[hn] - src * V;
		const size_t o = ((size_t)b * H + hn) * g.G + col, s = ((size_t)b * H + src) * g.G + col;
		long long idv;
		uint8_t pv;
		if (col < c) { idv = load_tok(g.ids_in, g.tok_bytes, s); pv = g.pad_in[s]; }
		else if (col == c) { idv = tok; pv = g.pad_in[s]; }
		else if (col == C) { idv = 0; pv = (tok == 0 || g.pad_in[((size_t)b * H + src) * g.G + c]) ? 1 : 0; }
		else { idv = 0; pv = 1; }
		store_tok(g.ids_out, g.tok_bytes, o, idv);
		g.pad_out[o] = pv;
	}
	if (tid < H) {
		const int src = s_pick_idx[tid] / V, tok = s_pick_idx[tid] - src * V;
		g.score_out[b * H + tid] = s_pick_raw[tid];
		if (g.src_out) g.src_out[b * H + tid] = src;
		g.score_normed[b * H + tid] = s_pick_val[tid];
		const bool nxt_pad = (tok == 0) || g.pad_in[((size_t)b * H + src) * g.G + c] != 0;
		g.len_out[b * H + tid] = g.len_in[b * H + src] + ((C < g.G && !nxt_pad) ? 1.f : 0.f);
		if (!nxt_pad) atomicAdd(g.active + c, 1);
	}
	beam_next_inputs(g, b, s_pick_idx, tid);
}

// The same step with ONE WAVE PER BEAM ROW (V <= 8192).  The kernel above serialises a sample's H rows through the whole workgroup (two block barriers and a
// thread-0 merge per row) and then runs H selection rounds that each cost two more barriers and a rescan of the winner's candidates through 2-byte global
// loads: 40 us per step at H = 4, V = 6912 -- 13 % of a beam-4 decode.  Here wave w owns rows w, w + 4, ...: the row's 16-byte chunks sit in the lanes'
// registers after one round trip (lane l: chunks l, l + 64, ...), the soft-max statistics are one wave reduction, and the row's own top-H comes out of H rounds
// of a wave-wide maximum over PACKED KEYS -- (order-preserving 16-bit image of the bf16 logit) << 16 | (0xFFFF - v): within a row the ranking value
// ((x / T - lse) + score) * scale rises with the logit, so "largest key" is "best candidate, lowest token id first on equal logits", one v_max_u32 per
// element and no index bookkeeping.  The H sorted lists (at most H x H candidates) are then merged by wave 0 in H rounds of a 32-lane arg-max on (ranking
// value descending, flat index h * V + v ascending) -- the defined tie-break.  One block barrier in all.
// (Inside ONE row two DIFFERENT logits whose fp32 ranking values coincide -- tiny logits under a large running score -- are ranked by logit here, by token id in
// the kernel above: both are orders the reference's unspecified torch.topk tie handling allows; equal logits are ranked by token id in both.)
__global__ __launch_bounds__(256) void beam_step_rows_kernel(const BeamArgs g) {
	constexpr int MAXH = 32, NCH = 16;  // 16-byte chunks per lane: 64 lanes x 16 x 8 = 8192 logits
	__shared__ float s_add[MAXH], s_scale[MAXH];
	__shared__ uint8_t s_fin[MAXH];
	__shared__ float s_cval[MAXH][MAXH], s_craw[MAXH][MAXH];
	__shared__ int s_cidx[MAXH][MAXH], s_cnt[MAXH];
	__shared__ float s_pick_val[MAXH], s_pick_raw[MAXH];
	__shared__ int s_pick_idx[MAXH];
	const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const int C = g.step, c = C - 1, H = g.H, V = g.V;
	const bf16* lg = g.logits + (size_t)b * H * g.ldl;
	const int nchunk = (V + 7) >> 3;
	if (tid < H) {
		s_fin[tid] = g.pad_in[((size_t)b * H + tid) * g.G + c] != 0;
		s_add[tid] = g.score_in[b * H + tid];
		s_scale[tid] = (g.alpha != 0.f) ? powf(fmaxf(g.len_in[b * H + tid], 1.f), -g.alpha) : 1.f;
	}
	__syncthreads();
	for (int h = w; h < H; h += 4) {
		const bool fin = s_fin[h] != 0;
		int Vl = V;
		asm volatile("" : "+s"(Vl));  // laundered per row: otherwise hipcc hoists the 128 range masks out of the row loop and keeps them in (spilled) SGPRs
		typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
		u16x8 xs[NCH];
#pragma unroll
		for (int k = 0; k < NCH; ++k) {
			const int ch = lane + (k << 6);
			xs[k] = (u16x8){0, 0, 0, 0, 0, 0, 0, 0};
			if (ch < nchunk) xs[k] = *reinterpret_cast<const u16x8*>(lg + (size_t)h * g.ldl + ch * 8);
		}
		auto f32_of = [](unsigned u) { return __builtin_bit_cast(float, u << 16); };
		// everything at or beyond V (row padding, the tail of the last chunk, chunks past the row) becomes -inf ONCE, so that no later pass tests a range
		// (128 loop-invariant lane masks kept alive across the passes were 500 spilled SGPRs)
#pragma unroll
		for (int k = 0; k < NCH; ++k)
#pragma unroll
			for (int e = 0; e < 8; ++e)
				if (((lane + (k << 6)) << 3) + e >= Vl) xs[k][e] = 0xFF80;
		// log-sum-exp of logits / T over the row (a finished beam: only END survives, with log-probability 0)
		float mx = -INFINITY;
#pragma unroll
		for (int k = 0; k < NCH; ++k)
#pragma unroll
			for (int e = 0; e < 8; ++e) mx = fmaxf(mx, f32_of(xs[k][e]) * g.inv_temp);
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
		float se = 0.f;
#pragma unroll
		for (int k = 0; k < NCH; ++k)
#pragma unroll
			for (int e = 0; e < 8; ++e) se += __expf(f32_of(xs[k][e]) * g.inv_temp - mx);
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
		const float lse = fin ? (float)lg[(size_t)h * g.ldl] * g.inv_temp : mx + __logf(se);
		// packed keys of the candidates this row may offer (0: none)
		unsigned key[NCH * 8];
#pragma unroll
		for (int k = 0; k < NCH; ++k)
#pragma unroll
			for (int e = 0; e < 8; ++e) {
				const int v = ((lane + (k << 6)) << 3) + e;
				const unsigned u = xs[k][e];
				const unsigned ord = (u & 0x8000u) ? (~u & 0xFFFFu) : (u | 0x8000u);
				key[k * 8 + e] = (ord << 16) | (0xFFFFu - (unsigned)(v & 0xFFFF));   // (beyond V: ord = 0x007F, the image of -inf -- dropped below)
			}
		// what may not be offered: everything beyond V (the -inf image), every token but END of a finished beam, END itself in the first step
		const unsigned floor_key = 0x007FFFFFu;  // largest key of a -inf logit
		if (fin) {
#pragma unroll
			for (int i = 0; i < NCH * 8; ++i) key[i] = 0u;
			if (lane == 0) key[0] = (0x8000u << 16) | 0xFFFFu;  // END (v = 0) of a finished beam: its logit is irrelevant (log-probability 0 by construction of lse)
		}
		if (C == 1 && h == 0 && lane == 0) key[0] = 0u;
		unsigned lim = 0xFFFFFFFFu;
		int cnt = 0;
		for (int r = 0; r < H; ++r) {
			unsigned m = 0u;
#pragma unroll
			for (int i = 0; i < NCH * 8; ++i) m = max(m, key[i] < lim ? key[i] : 0u);
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
			if (m <= floor_key) break;  // nothing (real) left to offer (wave-uniform)
			const unsigned ord = m >> 16;
			const unsigned u = (ord & 0x8000u) ? (ord & 0x7FFFu) : (~ord & 0xFFFFu);
			const int v = (int)(0xFFFFu - (m & 0xFFFFu));
			const float x = fin ? (float)lg[(size_t)h * g.ldl] : f32_of(u);
			const float raw = (x * g.inv_temp - lse) + s_add[h];
			if (lane == 0) { s_cval[h][r] = raw * s_scale[h]; s_craw[h][r] = raw; s_cidx[h][r] = h * V + v; }
			lim = m;
			cnt = r + 1;
		}
		if (lane == 0) s_cnt[h] = cnt;
	}
	__syncthreads();
	if (w == 0) {
		// H-way merge of the sorted row lists: lane h offers the head of row h's list
		int ptr = 0;
		const int myc = lane < H ? s_cnt[lane] : 0;
		for (int r = 0; r < H; ++r) {
			const bool have = lane < H && ptr < myc;
			float bv = have ? s_cval[lane][ptr] : -INFINITY, braw = have ? s_craw[lane][ptr] : -INFINITY;
			int bi = have ? s_cidx[lane][ptr] : 0x7fffffff;
#pragma unroll
			for (int o = 16; o > 0; o >>= 1) {
				const float ov = __shfl_xor(bv, o, 64), orw = __shfl_xor(braw, o, 64);
				const int oi = __shfl_xor(bi, o, 64);
				const bool take = oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi));
				if (take) { bv = ov; bi = oi; braw = orw; }
			}
			if (have && s_cidx[lane][ptr] == bi) ++ptr;
			if (lane == 0) { s_pick_val[r] = bv; s_pick_idx[r] = bi; s_pick_raw[r] = braw; }
		}
	}
	__syncthreads();
	sanitize_picks(s_pick_idx, H, V, tid);

	// reorder histories (ping-pong buffers), append tokens, update padding / scores / lengths: as beam_step_kernel
	for (int i = tid; i < H * g.G; i += 256) {
		const int hn = i / g.G, col = i - hn * g.G;
		const int src = s_pick_idx[hn] / V, tok = s_pick_idx[hn] - src * V;
		const size_t o = ((size_t)b * H + hn) * g.G + col, s = ((size_t)b * H + src) * g.G + col;
		long long idv;
		uint8_t pv;
		if (col < c) { idv = load_tok(g.ids_in, g.tok_bytes, s); pv = g.pad_in[s]; }
		else if (col == c) { idv = tok; pv = g.pad_in[s]; }
		else if (col == C) { idv = 0; pv = (tok == 0 || g.pad_in[((size_t)b * H + src) * g.G + c]) ? 1 : 0; }
		else { idv = 0; pv = 1; }
		store_tok(g.ids_out, g.tok_bytes, o, idv);
		g.pad_out[o] = pv;
	}
	if (tid < H) {
		const int src = s_pick_idx[tid] / V, tok = s_pick_idx[tid] - src * V;
		g.score_out[b * H + tid] = s_pick_raw[tid];
		if (g.src_out) g.src_out[b * H + tid] = src;
		g.score_normed[b * H + tid] = s_pick_val[tid];
		const bool nxt_pad = (tok == 0) || g.pad_in[((size_t)b * H + src) * g.G + c] != 0;
		g.len_out[b * H + tid] = g.len_in[b * H + src] + ((C < g.G && !nxt_pad) ? 1.f : 0.f);
		if (!nxt_pad) atomicAdd(g.active + c, 1);
	}
	beam_next_inputs(g, b, s_pick_idx, tid);
}

__global__ void mask_ids_kernel(void* ids, int tok_bytes, const uint8_t* pad, int n) {
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
		if (pad[i]) store_tok(ids, tok_bytes, i, 0);
}

}  // namespace

static int check_next(const novic_next_embed_t* next, novic_next_embed_t& out, bool with_origin) {
	out = novic_next_embed_t{};
	if (!next || !next->x_next) return 0;
	NOVIC_CHECK(next->struct_bytes == sizeof(novic_next_embed_t), "novic_next_embed_t: struct_bytes does not match this library's layout");
	NOVIC_CHECK(next->wtok && next->pos_row && next->E >= 4 && next->E % 4 == 0, "novic_next_embed_t: wtok / pos_row / E (a multiple of 4)");
	NOVIC_CHECK((((uintptr_t)next->wtok | (uintptr_t)next->pos_row | (uintptr_t)next->x_next) & 15) == 0, "novic_next_embed_t: operands must be 16-byte aligned");
	NOVIC_CHECK(with_origin || !next->origin_out, "novic_next_embed_t: an origin table only goes with a beam step");
	NOVIC_CHECK(!next->origin_out || (next->origin_in && next->npos >= 1), "novic_next_embed_t: origin_in / npos go with origin_out");
	out = *next;
	return 0;
}

extern "C" int novic_greedy_step(const void* logits_bf16, int ldl, int V, int B, int G, int step, void* ids, int tok_bytes, uint8_t* pad, float* alive, float* score,
                                 float* nll, float* count, int* active, float* step_logits, float temperature, float label_smoothing, hipStream_t stream) {
	return novic_greedy_step_next(logits_bf16, ldl, V, B, G, step, ids, tok_bytes, pad, alive, score, nll, count, active, step_logits, temperature, label_smoothing, nullptr, stream);
}

extern "C" int novic_greedy_step_next(const void* logits_bf16, int ldl, int V, int B, int G, int step, void* ids, int tok_bytes, uint8_t* pad, float* alive, float* score,
                                      float* nll, float* count, int* active, float* step_logits, float temperature, float label_smoothing, const novic_next_embed_t* next,
                                      hipStream_t stream) {
	NOVIC_CHECK(logits_bf16 && ids && pad && alive && score && nll && count && active, "novic_greedy_step: null pointer");
	NOVIC_CHECK(step >= 1 && step <= G && V >= 2 && temperature > 0.f, "novic_greedy_step: bad step / vocabulary / temperature");
	NOVIC_CHECK(tok_bytes == 4 || tok_bytes == 8, "novic_greedy_step: tok_bytes must be 4 or 8");
	if (B <= 0) return 0;
	GreedyArgs g = {(const bf16*)logits_bf16, ldl, V, B, G, step, ids, tok_bytes, pad, alive, score, nll, count, active, step_logits, 1.f / temperature, label_smoothing, {}};
	if (int rc = check_next(next, g.next, false)) return rc;
	NOVIC_CHECK(ldl % 8 == 0 && ((uintptr_t)logits_bf16 & 15) == 0, "novic_greedy_step: logits rows must be 16-byte aligned (ldl a multiple of 8)");
	int grid = B;
	if (grid > 8192) grid = 8192;
	hipLaunchKernelGGL(greedy_step_kernel, dim3(grid), dim3(256), 0, stream, g);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_greedy_finalize(void* ids, int tok_bytes, const uint8_t* pad, float* score, const float* count, int B, int G, float length_alpha, hipStream_t stream) {
	NOVIC_CHECK(ids && pad && score && count, "novic_greedy_finalize: null pointer");
	if (B <= 0) return 0;
	int grid = (B * G + 255) / 256;
	if (grid > 1024) grid = 1024;
	hipLaunchKernelGGL(greedy_finalize_kernel, dim3(grid), dim3(256), 0, stream, ids, tok_bytes, pad, score, count, B, G, length_alpha);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

static std::atomic<int> g_beam_step_generic{0};  // process-wide A/B switch (novic_beam_step_policy), read once per call
// Diagnostic: 1 forces the workgroup-per-sample kernel for every vocabulary size (tests compare the two); returns the previous setting.
extern "C" int novic_beam_step_policy(int generic) {
	const int prev = g_beam_step_generic;
	if (generic >= 0) g_beam_step_generic = generic;
	return prev;
}

extern "C" int novic_beam_step(const void* logits_bf16, int ldl, int V, int B, int H, int G, int step, const void* ids_in, void* ids_out, int tok_bytes, const uint8_t* pad_in,
                               uint8_t* pad_out, const float* score_in, float* score_out, float* score_normed, const float* len_in, float* len_out, int* active,
                               int* src_out, float temperature, float length_alpha, hipStream_t stream) {
	return novic_beam_step_next(logits_bf16, ldl, V, B, H, G, step, ids_in, ids_out, tok_bytes, pad_in, pad_out, score_in, score_out, score_normed, len_in, len_out, active, src_out,
	                            temperature, length_alpha, nullptr, stream);
}

extern "C" int novic_beam_step_next(const void* logits_bf16, int ldl, int V, int B, int H, int G, int step, const void* ids_in, void* ids_out, int tok_bytes,
                                    const uint8_t* pad_in, uint8_t* pad_out, const float* score_in, float* score_out, float* score_normed, const float* len_in, float* len_out,
                                    int* active, int* src_out, float temperature, float length_alpha, const novic_next_embed_t* next, hipStream_t stream) {
	NOVIC_CHECK(logits_bf16 && ids_in && ids_out && pad_in && pad_out && score_in && score_out && score_normed && len_in && len_out && active, "novic_beam_step: null pointer");
	NOVIC_CHECK(H >= 1 && H <= 32, "novic_beam_step: beam width must be in [1, 32]");
	NOVIC_CHECK(step >= 1 && step <= G && V >= 2 && temperature > 0.f, "novic_beam_step: bad step / vocabulary / temperature");
	NOVIC_CHECK(V - 1 >= H, "novic_beam_step: need at least H non-END tokens");
	NOVIC_CHECK(tok_bytes == 4 || tok_bytes == 8, "novic_beam_step: tok_bytes must be 4 or 8");
	NOVIC_CHECK(ldl % 8 == 0 && ((uintptr_t)logits_bf16 & 15) == 0, "novic_beam_step: logits rows must be 16-byte aligned (ldl a multiple of 8)");
	if (B <= 0) return 0;
	BeamArgs g = {(const bf16*)logits_bf16, ldl, V, B, H, G, step, ids_in, ids_out, tok_bytes, pad_in, pad_out, score_in, score_out, score_normed, len_in, len_out, active,
	              src_out, 1.f / temperature, length_alpha, {}};
	if (int rc = check_next(next, g.next, true)) return rc;
	if (V <= 8192 && !g_beam_step_generic) hipLaunchKernelGGL(beam_step_rows_kernel, dim3(B), dim3(256), 0, stream, g);  // one wave per beam row
	else hipLaunchKernelGGL(beam_step_kernel, dim3(B), dim3(256), 0, stream, g);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

// ---------------------------------------------------------------------------------------------------------
// "Is anybody still generating?" without a device -> host copy (round 6).  A decode call looks at the selection kernels' counter of step C - 1 after it has enqueued step C
// (early exit, embedding_decoder.py:819-820 / :965-967).  That look used to be a 4-byte copy into page-locked memory + an event behind every step, BETWEEN the steps' graphs:
// 7.4 us of a 170 us greedy step at 256 rows (tools/decode_copy_probe.py: +4.5 % greedy, +2.6 % beam-4 with neither).  This one-thread launch -- the last node of the step's graph --
// writes the answer straight into page-locked host memory that is mapped into the device's address space (fine-grained: hipHostMalloc's default, which is what torch's pinned
// allocator uses): 1 = the step is done and nothing is active, 2 = done and something is; the host clears the word before the call and polls it.  A system-scope store to
// fine-grained memory: visible to the host without waiting for the end of the graph or of the stream.
// ---------------------------------------------------------------------------------------------------------
__global__ void step_done_kernel(const int* __restrict__ active, int* __restrict__ flag) {
	const int a = __hip_atomic_load(active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	// RELAXED: the host reads nothing but this word behind it, so nothing has to be made visible WITH it -- a release at system scope writes the L2's dirty lines back first,
	// which beside a tower that streams its outputs through the L2s is the tower's data, at every step of every lane
	__hip_atomic_store(flag, a != 0 ? 2 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int novic_step_done(const int* active, int* done_flag_dev, hipStream_t stream) {
	NOVIC_CHECK(active && done_flag_dev, "novic_step_done: null pointer");
	NOVIC_CHECK((((uintptr_t)active | (uintptr_t)done_flag_dev) & 3) == 0, "novic_step_done: misaligned word");
	hipLaunchKernelGGL(step_done_kernel, dim3(1), dim3(1), 0, stream, active, done_flag_dev);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_host_mapped_ptr(void* host, void** dev) {
	NOVIC_CHECK(host && dev, "novic_host_mapped_ptr: null pointer");
	*dev = nullptr;
	hipPointerAttribute_t at;
	if (hipPointerGetAttributes(&at, host) != hipSuccess || at.type != hipMemoryTypeHost) {
		(void)hipGetLastError();
		novic_set_error("novic_host_mapped_ptr: not page-locked host memory (hipHostMalloc / a pinned torch tensor)");
		return -22;
	}
	void* d = nullptr;
	if (hipHostGetDevicePointer(&d, host, 0) != hipSuccess || !d) {
		(void)hipGetLastError();
		novic_set_error("novic_host_mapped_ptr: the allocation is not mapped into the device's address space");
		return -22;
	}
	*dev = d;
	return 0;
}

extern "C" int novic_mask_ids(void* ids, int tok_bytes, const uint8_t* pad, int n, hipStream_t stream) {
	NOVIC_CHECK(ids && pad, "novic_mask_ids: null pointer");
	if (n <= 0) return 0;
	int grid = (n + 255) / 256;
	if (grid > 1024) grid = 1024;
	hipLaunchKernelGGL(mask_ids_kernel, dim3(grid), dim3(256), 0, stream, ids, tok_bytes, pad, n);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

// ---------------------------------------------------------------------------------------------------------
// KV-cached decode step (legal because outputs are identical: SURVEY.md A.6): the prefix positions are computed once per SAMPLE and
// shared by its H beams, every later step processes ONE new position per beam.
//   novic_decode_embed : x[a] = W_tok[ids[a][step-2]] + pos[P + step - 2]                       (f32 [A][E])
//   novic_decode_attn  : append this position's k,v to the beam cache, attend the single query over prefix keys (per-sample cache =
//                        the step-1 qkv buffer) + label keys (per-beam cache); one wave per (beam, head); keys <= 32
//   novic_kv_reorder   : cache_out[h'] = cache_in[src[h']] for the label positions written so far (beam reordering)
// All three are HBM/latency-bound helpers around the small-M GEMMs of a step.
// ---------------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void decode_embed_kernel(const void* __restrict__ ids, int tok_bytes, int G, int col, const float* __restrict__ wtok,
                                                           const float* __restrict__ pos_row, float* __restrict__ x, int A, int E, int V) {
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (int a = blockIdx.x * 4 + w; a < A; a += gridDim.x * 4) {
		long long t = load_tok(ids, tok_bytes, (size_t)a * G + col);
		t = t < 0 ? 0 : (t >= V ? V - 1 : t);
		for (int e = lane * 4; e < E; e += 256) {
			const f32x4 tv = *reinterpret_cast<const f32x4*>(wtok + (size_t)t * E + e);
			const f32x4 pv = *reinterpret_cast<const f32x4*>(pos_row + e);
			*reinterpret_cast<f32x4*>(x + (size_t)a * E + e) = (f32x4){tv[0] + pv[0], tv[1] + pv[1], tv[2] + pv[2], tv[3] + pv[3]};
		}
	}
}

struct DecAttnArgs {
	const bf16* qkv_new;     // [A][3E] this position's q,k,v
	const bf16* prefix_qkv;  // [B*P][3E] step-1 qkv of the prefix positions (k at +E, v at +2E)
	bf16* cache_k;           // [A][G][E] label positions (seq position P + g)
	bf16* cache_v;
	bf16* o;                 // [A][E]
	int A, H, D, P, G, pos, beams;  // pos = label position index being written (0-based), beams = sequences per sample
	float scale;
	const int* origin;       // NULL, or [A][G]: the cache ROW that holds label position g of sequence a (beam search without moving K/V: novic_kv_origin_update)
};

__global__ __launch_bounds__(256) void decode_attn_kernel(const DecAttnArgs g) {
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int E = g.H * g.D;
	const int pair = blockIdx.x * 4 + w;
	if (pair >= g.A * g.H) return;
	const int a = pair / g.H, h = pair - a * g.H;
	const int b = a / g.beams;
	const bf16* qn = g.qkv_new + (size_t)a * 3 * E + h * g.D;
	// append k, v of the new position (lanes <-> d)
	if (lane < g.D) {
		g.cache_k[((size_t)a * g.G + g.pos) * E + h * g.D + lane] = qn[E + lane];
		g.cache_v[((size_t)a * g.G + g.pos) * E + h * g.D + lane] = qn[2 * E + lane];
	}
	const int nkeys = g.P + g.pos + 1;  // <= 32
	auto crow = [&](int gp) -> size_t { return (size_t)(g.origin ? g.origin[(size_t)a * g.G + gp] : a) * g.G + gp; };  // cache row of label position gp
	auto krow = [&](int j) -> const bf16* {
		return (j < g.P) ? g.prefix_qkv + ((size_t)b * g.P + j) * 3 * E + E + h * g.D : ((j - g.P == g.pos) ? qn + E : g.cache_k + crow(j - g.P) * E + h * g.D);
	};
	auto vrow = [&](int j) -> const bf16* {
		return (j < g.P) ? g.prefix_qkv + ((size_t)b * g.P + j) * 3 * E + 2 * E + h * g.D : ((j - g.P == g.pos) ? qn + 2 * E : g.cache_v + crow(j - g.P) * E + h * g.D);
	};
	// The value rows do not depend on the scores: request them first (lanes <-> d, one 2-byte element per key) so they fly under the QK part.
	float vv[32];
#pragma unroll
	for (int jj = 0; jj < 32; ++jj) vv[jj] = (jj < nkeys && lane < g.D) ? (float)vrow(jj)[lane] : 0.f;
	// scores: lane = key j (0..31) x half of the head dimension; 16-byte loads when the half is a multiple of 8 elements (D = 32, 64)
	const int j = lane & 31, half = lane >> 5;
	const int dh = g.D >> 1;
	float s = 0.f;
	if (j < nkeys) {
		const bf16* kr = krow(j) + half * dh;
		const bf16* qr = qn + half * dh;
		if ((dh & 7) == 0) {
			for (int d = 0; d < dh; d += 8) {
				const bf16x8 kq = *reinterpret_cast<const bf16x8*>(kr + d), qq = *reinterpret_cast<const bf16x8*>(qr + d);
#pragma unroll
				for (int k = 0; k < 8; ++k) s += (float)qq[k] * (float)kq[k];
			}
		} else {
			for (int d = 0; d < dh; ++d) s += (float)qr[d] * (float)kr[d];
		}
	}
	s += __shfl_xor(s, 32, 64);
	s = (j < nkeys) ? s * g.scale : -1e30f;
	float mx = s;
#pragma unroll
	for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
	float e = (j < nkeys) ? __expf(s - mx) : 0.f;
	float sum = e;
#pragma unroll
	for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
	const float p = e / sum;  // lanes l and l+32 hold the same p_j
	// out[d] = sum_j p_j v[j][d], lanes <-> d
	float acc = 0.f;
#pragma unroll
	for (int jj = 0; jj < 32; ++jj) acc += __shfl(p, jj, 64) * vv[jj];
	if (lane < g.D) g.o[(size_t)a * E + h * g.D + lane] = (bf16)acc;
}

// The same step for head_dim 64 and <= 16 keys (prefix 4 + up to 12 label positions: every released configuration): lane (j, c) = key j x 16-dim
// quarter c.  Each lane picks its key's row ONCE (prefix / cache / the new position), so a wave issues six 16-byte loads per lane -- q, k and v
// quarters -- in one round trip, where the kernel above runs 32 predicated 2-byte value loads with a three-way address select each plus a key
// loop (8.4 us per launch at 256 rows, 16 us at 1024: a quarter of a beam-4 step).  Scores: 16-term partial dot + two lane exchanges; soft-max over
// the key lanes; PV as a reduce-scatter over the key lanes (15 exchanges: every step halves the values a lane still carries), which leaves output
// dimension 16 c + bits(j) on lane (j, c).  fp32 sums in a different order than the kernel above (equal to ~1e-7 relative before the bf16 store).
__global__ __launch_bounds__(256) void decode_attn16_kernel(const DecAttnArgs g) {
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int E = g.H * 64;
	const int pair = blockIdx.x * 4 + w;
	if (pair >= g.A * g.H) return;
	const int a = pair / g.H, h = pair - a * g.H;
	const int b = a / g.beams;
	const int j = lane >> 2, c = lane & 3;
	const int nkeys = g.P + g.pos + 1;  // <= 16
	const bf16* qn = g.qkv_new + (size_t)a * 3 * E + h * 64 + c * 16;
	const bool valid = j < nkeys, isnew = j == nkeys - 1;
	const bf16 *kr = qn + E, *vr = qn + 2 * E;  // the new position (lanes beyond the last key read it too: masked below)
	if (j < g.P) {
		kr = g.prefix_qkv + ((size_t)b * g.P + j) * 3 * E + E + h * 64 + c * 16;
		vr = kr + E;
	} else if (valid && !isnew) {
		const int row = g.origin ? g.origin[(size_t)a * g.G + (j - g.P)] : a;  // whose cache row holds this position (beams: the ancestor that computed it)
		const size_t off = ((size_t)row * g.G + (j - g.P)) * E + h * 64 + c * 16;
		kr = g.cache_k + off;
		vr = g.cache_v + off;
	}
	const bf16x8 q0 = *reinterpret_cast<const bf16x8*>(qn), q1 = *reinterpret_cast<const bf16x8*>(qn + 8);
	const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(kr), k1 = *reinterpret_cast<const bf16x8*>(kr + 8);
	const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(vr), v1 = *reinterpret_cast<const bf16x8*>(vr + 8);
	if (isnew) {  // append k, v of the new position
		const size_t off = ((size_t)a * g.G + g.pos) * E + h * 64 + c * 16;
		*reinterpret_cast<bf16x8*>(g.cache_k + off) = k0;
		*reinterpret_cast<bf16x8*>(g.cache_k + off + 8) = k1;
		*reinterpret_cast<bf16x8*>(g.cache_v + off) = v0;
		*reinterpret_cast<bf16x8*>(g.cache_v + off + 8) = v1;
	}
	float s = 0.f;
#pragma unroll
	for (int i = 0; i < 8; ++i) s += (float)q0[i] * (float)k0[i];
#pragma unroll
	for (int i = 0; i < 8; ++i) s += (float)q1[i] * (float)k1[i];
	s += __shfl_xor(s, 1, 64);
	s += __shfl_xor(s, 2, 64);
	s = valid ? s * g.scale : -1e30f;
	float mx = s;
#pragma unroll
	for (int o = 4; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
	const float e = valid ? __expf(s - mx) : 0.f;
	float sum = e;
#pragma unroll
	for (int o = 4; o < 64; o <<= 1) sum += __shfl_xor(sum, o, 64);
	const float p = e / sum;
	float t[16];
#pragma unroll
	for (int i = 0; i < 8; ++i) { t[i] = p * (float)v0[i]; t[8 + i] = p * (float)v1[i]; }
	// reduce-scatter over the key lanes: after the exchange with lane ^ 32 a lane keeps 8 of its 16 dimensions, then 4, 2, 1
	float r8[8], r4[4], r2[2];
	{
		const bool hi = (lane & 32) != 0;
#pragma unroll
		for (int i = 0; i < 8; ++i) r8[i] = (hi ? t[8 + i] : t[i]) + __shfl_xor(hi ? t[i] : t[8 + i], 32, 64);
	}
	{
		const bool hi = (lane & 16) != 0;
#pragma unroll
		for (int i = 0; i < 4; ++i) r4[i] = (hi ? r8[4 + i] : r8[i]) + __shfl_xor(hi ? r8[i] : r8[4 + i], 16, 64);
	}
	{
		const bool hi = (lane & 8) != 0;
#pragma unroll
		for (int i = 0; i < 2; ++i) r2[i] = (hi ? r4[2 + i] : r4[i]) + __shfl_xor(hi ? r4[i] : r4[2 + i], 8, 64);
	}
	const bool hi4 = (lane & 4) != 0;
	const float out = (hi4 ? r2[1] : r2[0]) + __shfl_xor(hi4 ? r2[0] : r2[1], 4, 64);
	const int d = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
	g.o[(size_t)a * E + h * 64 + c * 16 + d] = (bf16)out;
}

// new sequence a = (b, h') copies the label-position rows [0, npos) of old sequence (b, src) where src = first column's provenance:
// the beam step records it in src_idx[a] (old beam index within the sample).
__global__ __launch_bounds__(256) void kv_reorder_kernel(const bf16* __restrict__ k_in, const bf16* __restrict__ v_in, bf16* __restrict__ k_out, bf16* __restrict__ v_out,
                                                         const int* __restrict__ src_idx, int layers, int A, int beams, int G, int E, int npos) {
	const int chunks = npos * E / 8;
	for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < (size_t)layers * A * chunks; idx += (size_t)gridDim.x * 256) {
		const int al = (int)(idx / chunks), c = (int)(idx % chunks);
		const int l = al / A, a0 = al - l * A;
		const int a = al;
		const int srca = l * A + (a0 / beams) * beams + src_idx[a0];
		const size_t so = (size_t)srca * G * E + (size_t)c * 8, dof = (size_t)a * G * E + (size_t)c * 8;
		*reinterpret_cast<uint4*>(k_out + dof) = *reinterpret_cast<const uint4*>(k_in + so);
		*reinterpret_cast<uint4*>(v_out + dof) = *reinterpret_cast<const uint4*>(v_in + so);
	}
}

}  // namespace

extern "C" int novic_decode_embed(const void* ids, int tok_bytes, int G, int col, const float* wtok, const float* pos_row, float* x, int A, int E, int V, hipStream_t stream) {
	NOVIC_CHECK(ids && wtok && pos_row && x, "novic_decode_embed: null pointer");
	NOVIC_CHECK((tok_bytes == 4 || tok_bytes == 8) && E % 4 == 0 && col >= 0 && col < G, "novic_decode_embed: bad arguments");
	if (A <= 0) return 0;
	int grid = (A + 3) / 4;
	if (grid > 4096) grid = 4096;
	hipLaunchKernelGGL(decode_embed_kernel, dim3(grid), dim3(256), 0, stream, ids, tok_bytes, G, col, wtok, pos_row, x, A, E, V);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_decode_attn(const void* qkv_new_bf16, const void* prefix_qkv_bf16, void* cache_k_bf16, void* cache_v_bf16, void* o_bf16, int A, int H, int D, int P, int G,
                                 int pos, int beams, const int* origin, hipStream_t stream) {
	NOVIC_CHECK(qkv_new_bf16 && prefix_qkv_bf16 && cache_k_bf16 && cache_v_bf16 && o_bf16, "novic_decode_attn: null pointer");
	NOVIC_CHECK(D <= 64 && D % 2 == 0 && pos >= 0 && pos < G && P + pos + 1 <= 32 && beams >= 1 && A % beams == 0, "novic_decode_attn: bad shape (head_dim <= 64, <= 32 keys)");
	if (A <= 0) return 0;
	DecAttnArgs g = {(const bf16*)qkv_new_bf16, (const bf16*)prefix_qkv_bf16, (bf16*)cache_k_bf16, (bf16*)cache_v_bf16, (bf16*)o_bf16, A, H, D, P, G, pos, beams,
	                 1.f / sqrtf((float)D), origin};
	const bool aligned = ((((uintptr_t)qkv_new_bf16 | (uintptr_t)prefix_qkv_bf16 | (uintptr_t)cache_k_bf16 | (uintptr_t)cache_v_bf16) & 15) == 0);
	if (D == 64 && P + pos + 1 <= 16 && aligned) hipLaunchKernelGGL(decode_attn16_kernel, dim3((A * H + 3) / 4), dim3(256), 0, stream, g);
	else hipLaunchKernelGGL(decode_attn_kernel, dim3((A * H + 3) / 4), dim3(256), 0, stream, g);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

// Beam search without moving K/V: sequence a = (b, h') continues old sequence sa = (b, src[a]); its label positions [0, npos - 1) live where sa's
// did, and position npos - 1 -- computed in the step that just ran -- lives in sa's own cache row.  A x G ints per step instead of the caches of
// every layer (kv_reorder: 138 MB at 1024 sequences x 11 positions, 29 us per step).
__global__ void kv_origin_update_kernel(const int* __restrict__ src, const int* __restrict__ in, int* __restrict__ out, int A, int beams, int G, int npos) {
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < A * npos; i += gridDim.x * blockDim.x) {
		const int a = i / npos, gp = i - a * npos;
		const int sa = (a / beams) * beams + src[a];
		out[(size_t)a * G + gp] = gp == npos - 1 ? sa : in[(size_t)sa * G + gp];
	}
}

extern "C" int novic_kv_origin_update(const int* src_idx, const int* origin_in, int* origin_out, int A, int beams, int G, int npos, hipStream_t stream) {
	NOVIC_CHECK(src_idx && origin_in && origin_out && origin_in != origin_out, "novic_kv_origin_update: null pointer / in-place update");
	NOVIC_CHECK(beams >= 1 && A % beams == 0 && npos >= 0 && npos <= G, "novic_kv_origin_update: bad shape");
	if (A <= 0 || npos == 0) return 0;
	int grid = (A * npos + 255) / 256;
	if (grid > 1024) grid = 1024;
	hipLaunchKernelGGL(kv_origin_update_kernel, dim3(grid), dim3(256), 0, stream, src_idx, origin_in, origin_out, A, beams, G, npos);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_kv_reorder(const void* k_in, const void* v_in, void* k_out, void* v_out, const int* src_idx, int layers, int A, int beams, int G, int E, int npos,
                                hipStream_t stream) {
	NOVIC_CHECK(k_in && v_in && k_out && v_out && src_idx, "novic_kv_reorder: null pointer");
	NOVIC_CHECK(E % 8 == 0 && npos >= 0 && npos <= G && beams >= 1, "novic_kv_reorder: bad shape");
	if (A <= 0 || npos == 0) return 0;
	size_t total = (size_t)layers * A * (npos * E / 8);
	int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
	hipLaunchKernelGGL(kv_reorder_kernel, dim3(grid), dim3(256), 0, stream, (const bf16*)k_in, (const bf16*)v_in, (bf16*)k_out, (bf16*)v_out, src_idx, layers, A, beams, G, E,
	                   npos);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Guided decoding over a token trie of the guide / vocabulary nouns (reference embedding_decoder.py:788, :808-813 greedy; :915-943, :969-975
// beams).  The reference keeps a B x H x W "still consistent" mask over all W nouns and scatters their next tokens into a (V+1)-wide mask
// every step (W = 42 919 for the released vocabulary); here a beam carries ONE integer -- its trie node -- and the tokens it may emit next
// are that node's children (CSR arrays, children sorted by token id).  node >= 0: on the trie; -1: finished (emitted END); -2: dead.
//   renorm: probabilities are renormalised over the allowed tokens (guide_renorm=True), otherwise the full-vocabulary log-softmax is kept.
//   prior : child_logprior[edge] = log P(token | prefix) among the vocabulary nouns (per target: count ratio, per token: 1 / #children);
//           scores -= prior_scale * log P (the reference's vocab_targets / vocab_scaler correction).
// ---------------------------------------------------------------------------------------------------------
namespace {

struct Trie {
	const int* start;      // [nodes + 1]
	const int* tok;        // [edges]
	const int* next;       // [edges]  child node, -1 = END edge
	const float* logprior; // [edges] or null
};

struct GuidedBeamArgs {
	BeamArgs b;
	Trie t;
	const int* node_in;  // [B][H]
	int* node_out;
	int renorm;
	float prior_scale;
	// optional SECOND trie: the vocabulary nouns when they are not the guide nouns (prior = v.logprior of the candidate token under the beam's vocabulary
	// node; a token no consistent vocabulary noun continues with has prior 0 -> the candidate is banned, reference :924-936)
	Trie v;
	const int* vnode_in;
	int* vnode_out;
};

constexpr int GB_CACHE = 7168;  // candidates whose (ranking value, flat index) fit the kernel's 56 KiB of dynamic LDS (the root of a V = 6912 trie has < 6912 children)

__global__ __launch_bounds__(256) void beam_step_guided_kernel(const GuidedBeamArgs a) {
	constexpr int MAXH = 32;
	const BeamArgs& g = a.b;
	__shared__ float s_lse[MAXH], s_add[MAXH], s_scale[MAXH];
	__shared__ int s_node[MAXH], s_off[MAXH + 1], s_first[MAXH], s_vnode[MAXH], s_pick_vnext[MAXH];
	const bool two = a.v.start != nullptr;
	__shared__ uint8_t s_fin[MAXH];
	__shared__ float s_val[4], s_raw[4];
	__shared__ int s_idx[4], s_cand[4];
	__shared__ float s_pick_val[MAXH], s_pick_raw[MAXH];
	__shared__ int s_pick_flat[MAXH], s_pick_next[MAXH];
	const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const int C = g.step, c = C - 1, H = g.H, V = g.V;
	const bf16* lg = g.logits + (size_t)b * H * g.ldl;

	for (int h = w; h < H; h += 4) {
		const bool fin = g.pad_in[((size_t)b * H + h) * g.G + c] != 0;
		const int node = a.node_in[b * H + h];
		const int e0 = (!fin && node >= 0) ? a.t.start[node] : 0, e1 = (!fin && node >= 0) ? a.t.start[node + 1] : 0;
		float mx = -INFINITY, se = 0.f;
		if (!fin) {
			if (a.renorm) {
				for (int e = e0 + lane; e < e1; e += 64) {
					const float x = (float)lg[(size_t)h * g.ldl + a.t.tok[e]] * g.inv_temp;
					if (x > mx) { se = se * __expf(mx - x) + 1.f; mx = x; } else se += __expf(x - mx);
				}
			} else {
				if (V <= 8192) {
					// the lane's chunks of the row in registers after ONE round trip, the maximum first, then the exponentials: the online form below is a
					// dependent compare / exp chain of 8 * chunks steps per lane (beam-10 guided: 78 -> ~40 us per step, three rows per wave)
					bf16x8 xs[16];
					const int nchunk = (V + 7) >> 3;
#pragma unroll
					for (int k = 0; k < 16; ++k) {
						const int ch = lane + (k << 6);
						if (ch < nchunk) xs[k] = *reinterpret_cast<const bf16x8*>(lg + (size_t)h * g.ldl + ch * 8);  // ldl is a multiple of 8 (checked by the launcher)
					}
#pragma unroll
					for (int k = 0; k < 16; ++k)
#pragma unroll
						for (int e = 0; e < 8; ++e)
							if (((lane + (k << 6)) << 3) + e < V) mx = fmaxf(mx, (float)xs[k][e] * g.inv_temp);
#pragma unroll
					for (int k = 0; k < 16; ++k)
#pragma unroll
						for (int e = 0; e < 8; ++e)
							if (((lane + (k << 6)) << 3) + e < V) se += __expf((float)xs[k][e] * g.inv_temp - mx);
				} else {
				for (int v0 = lane * 8; v0 < V; v0 += 64 * 8) {
					const bf16x8 xs = *reinterpret_cast<const bf16x8*>(lg + (size_t)h * g.ldl + v0);
#pragma unroll
					for (int k = 0; k < 8; ++k) {
						if (v0 + k >= V) break;
						const float x = (float)xs[k] * g.inv_temp;
						if (x > mx) { se = se * __expf(mx - x) + 1.f; mx = x; } else se += __expf(x - mx);
					}
				}
				}
			}
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) {
				const float omx = __shfl_xor(mx, o, 64), ose = __shfl_xor(se, o, 64);
				const float nm = fmaxf(mx, omx);
				if (nm != -INFINITY) se = se * __expf(mx - nm) + ose * __expf(omx - nm);
				mx = nm;
			}
		}
		if (lane == 0) {
			s_fin[h] = fin;
			s_node[h] = node;
			s_vnode[h] = two ? a.vnode_in[b * H + h] : -2;
			s_first[h] = e0;
			s_off[h + 1] = fin ? 1 : (e1 - e0);
			s_lse[h] = fin ? (float)lg[(size_t)h * g.ldl] * g.inv_temp : mx + __logf(se);
			s_add[h] = g.score_in[b * H + h];
			s_scale[h] = (g.alpha != 0.f) ? powf(fmaxf(g.len_in[b * H + h], 1.f), -g.alpha) : 1.f;
		}
	}
	__syncthreads();
	if (tid == 0) {
		s_off[0] = 0;
		for (int h = 0; h < H; ++h) s_off[h + 1] += s_off[h];
	}
	__syncthreads();
	const int total = s_off[H];

	auto cand = [&](int i, float& raw, int& flat, int& nxt, int& vnxt) -> float {
		int h = 0;
		while (i >= s_off[h + 1]) ++h;
		int tok = 0;
		nxt = vnxt = -1;
		float prior = 0.f;
		if (!s_fin[h]) {
			const int e = s_first[h] + (i - s_off[h]);
			tok = a.t.tok[e];
			nxt = a.t.next[e];
			if (a.t.logprior) prior = a.t.logprior[e];
			if (two) {  // the token's edge under the beam's vocabulary node (children sorted by token: binary search)
				const int vn = s_vnode[h];
				int lo = vn >= 0 ? a.v.start[vn] : 0, hi = vn >= 0 ? a.v.start[vn + 1] : 0;
				const int end = hi;
				while (lo < hi) {
					const int mid = (lo + hi) >> 1;
					if (a.v.tok[mid] < tok) lo = mid + 1; else hi = mid;
				}
				flat = h * V + tok;
				if (lo >= end || a.v.tok[lo] != tok) { raw = -INFINITY; return -INFINITY; }  // no consistent vocabulary noun has this token here
				prior = a.v.logprior[lo];
				vnxt = a.v.next[lo];
			}
		}
		flat = h * V + tok;
		if (C == 1 && h == 0 && tok == 0) { raw = -INFINITY; return -INFINITY; }
		raw = ((float)lg[(size_t)h * g.ldl + tok] * g.inv_temp - s_lse[h]) - a.prior_scale * prior + s_add[h];
		if (s_fin[h]) raw = s_add[h];
		return raw * s_scale[h];
	};

	// Selection as in beam_step_kernel: every thread caches the best candidate of its strided subset (i = tid, tid + 256, ...); a round is a block
	// arg-max over the cached offers; only the winning thread re-scans its subset for its next offer.  Order: value descending, flat index h*V + tok
	// ascending; -inf candidates (dead parents, the first-step END ban, tokens without vocabulary mass) are no candidates.
	// A candidate's ranking value costs a chain of dependent global loads (trie edge -> token -> logit, a binary search in the vocabulary trie): the FIRST scan
	// parks (value, flat index) of every candidate in LDS (GB_CACHE of them: a trie node has at most V children, only the root has thousands), the re-scans of the
	// selection rounds read them back from there (the winning thread alone re-evaluated ~20 candidates per round through memory: 78 us per step at H = 10).
	extern __shared__ __attribute__((aligned(8))) char gb_cache[];
	float* c_val = reinterpret_cast<float*>(gb_cache);
	int* c_flat = reinterpret_cast<int*>(gb_cache + GB_CACHE * sizeof(float));
	const bool cached = total <= GB_CACHE;
	auto scan = [&](bool first, float pv, int pf, float& bv, float& braw, int& bflat, int& bi) {
		bv = braw = -INFINITY; bflat = 0x7fffffff; bi = -1;
		for (int i = tid; i < total; i += 256) {
			float raw = 0.f, val;
			int flat, nxt, vnxt;
			if (first || !cached) {
				val = cand(i, raw, flat, nxt, vnxt);
				if (cached) { c_val[i] = val; c_flat[i] = flat; }
			} else {
				val = c_val[i];
				flat = c_flat[i];
			}
			const bool after = pf < 0 || (val < pv) || (val == pv && flat > pf);
			if (!after || !(val > -INFINITY)) continue;
			if (bi < 0 || val > bv || (val == bv && flat < bflat)) { bv = val; bflat = flat; braw = raw; bi = i; }
		}
	};
	float my_v, my_raw;
	int my_flat, my_i;
	scan(true, 0.f, -1, my_v, my_raw, my_flat, my_i);
	// Few candidates (every step but the first: a beam's trie node has a handful of children): ONE wave holds them all -- 16 per lane out of the LDS cache -- and
	// runs the H selection rounds as wave-wide arg-max reductions without a single block barrier (the general rounds below cost three barriers, a thread-0 merge and
	// a one-thread re-scan each: ~3.5 us per round, 35 of the step's 50 us at H = 10); the picks' trie edges are looked up afterwards by H threads in parallel.
	__shared__ int s_pick_cand[MAXH];
	const bool small = cached && total <= 1024;
	if (small) {
		__syncthreads();  // the cache is complete
		if (w == 0) {
			// NK candidates per lane (1, 4 or 16: a step with a handful of candidates should not pay for 16 compare chains per round)
			auto select = [&](auto nk_c) {
				constexpr int NK = decltype(nk_c)::value;
				float cv[NK];
				int cf[NK];
#pragma unroll
				for (int k = 0; k < NK; ++k) {
					const int i = lane + (k << 6);
					cv[k] = i < total ? c_val[i] : -INFINITY;
					cf[k] = i < total ? c_flat[i] : 0x7fffffff;
				}
				float pv = 0.f;
				int pf = -1;
				for (int r = 0; r < H; ++r) {
					float bv = -INFINITY;
					int bflat = 0x7fffffff, bi = -1;
#pragma unroll
					for (int k = 0; k < NK; ++k) {
						const bool after = pf < 0 || (cv[k] < pv) || (cv[k] == pv && cf[k] > pf);
						const bool ok = after && cv[k] > -INFINITY;
						if (ok && (bi < 0 || cv[k] > bv || (cv[k] == bv && cf[k] < bflat))) { bv = cv[k]; bflat = cf[k]; bi = lane + (k << 6); }
					}
#pragma unroll
					for (int o = 32; o > 0; o >>= 1) {
						const float ov = __shfl_xor(bv, o, 64);
						const int of = __shfl_xor(bflat, o, 64), oi = __shfl_xor(bi, o, 64);
						if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && of < bflat))) { bv = ov; bflat = of; bi = oi; }
					}
					if (lane == 0) { s_pick_val[r] = bi >= 0 ? bv : -INFINITY; s_pick_flat[r] = bi >= 0 ? bflat : -1; s_pick_cand[r] = bi; }
					if (bi < 0) {  // nothing left: the remaining beams are dead (wave-uniform)
						for (int rr = r + 1; rr < H; ++rr)
							if (lane == 0) { s_pick_val[rr] = -INFINITY; s_pick_flat[rr] = -1; s_pick_cand[rr] = -1; }
						break;
					}
					pv = bv;
					pf = bflat;
				}
			};
			if (total <= 64) select(std::integral_constant<int, 1>{});
			else if (total <= 256) select(std::integral_constant<int, 4>{});
			else select(std::integral_constant<int, 16>{});
		}
		__syncthreads();
		if (tid < H) {
			const int fi = s_pick_cand[tid];
			float raw = -INFINITY;
			int flat, nxt = -2, vnxt = -2;
			if (fi >= 0) cand(fi, raw, flat, nxt, vnxt);
			s_pick_raw[tid] = fi >= 0 ? raw : -INFINITY;
			s_pick_next[tid] = nxt;
			s_pick_vnext[tid] = vnxt;
		}
		__syncthreads();
	} else
	for (int r = 0; r < H; ++r) {
		float bv = my_v, braw = my_raw;
		int bflat = my_flat, bi = my_i;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			const float ov = __shfl_xor(bv, o, 64), orw = __shfl_xor(braw, o, 64);
			const int of = __shfl_xor(bflat, o, 64), oi = __shfl_xor(bi, o, 64);
			if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && of < bflat))) { bv = ov; bflat = of; braw = orw; bi = oi; }
		}
		if (lane == 0) { s_val[w] = bv; s_idx[w] = bflat; s_raw[w] = braw; s_cand[w] = bi; }
		__syncthreads();
		if (tid == 0) {
			float fv = s_val[0], fr = s_raw[0];
			int ff = s_idx[0], fi = s_cand[0];
			for (int k = 1; k < 4; ++k)
				if (s_cand[k] >= 0 && (fi < 0 || s_val[k] > fv || (s_val[k] == fv && s_idx[k] < ff))) { fv = s_val[k]; ff = s_idx[k]; fr = s_raw[k]; fi = s_cand[k]; }
			int nxt = -2, vnxt = -2;
			if (fi >= 0) { float raw; int flat; cand(fi, raw, flat, nxt, vnxt); fr = raw; }  // (the raw score comes from here: cached re-scans carry only the ranking value)
			s_pick_val[r] = fi >= 0 ? fv : -INFINITY;
			s_pick_raw[r] = fi >= 0 ? fr : -INFINITY;
			s_pick_flat[r] = fi >= 0 ? ff : -1;   // -1: no candidate left (fewer allowed continuations than beams) -> dead beam
			s_pick_next[r] = nxt;
			s_pick_vnext[r] = vnxt;
			s_cand[0] = fi;
		}
		__syncthreads();
		const int won = s_cand[0];
		if (won >= 0 && (won & 255) == tid) scan(false, s_pick_val[r], s_pick_flat[r], my_v, my_raw, my_flat, my_i);  // candidate i belongs to thread i % 256
		__syncthreads();
	}

	for (int i = tid; i < H * g.G; i += 256) {
		const int hn = i / g.G, col = i - hn * g.G;
		const bool dead = s_pick_flat[hn] < 0;
		const int src = dead ? 0 : s_pick_flat[hn] / V, tok = dead ? 0 : s_pick_flat[hn] - src * V;
		const size_t o = ((size_t)b * H + hn) * g.G + col, s = ((size_t)b * H + src) * g.G + col;
		long long idv;
		uint8_t pv;
		if (col < c) { idv = load_tok(g.ids_in, g.tok_bytes, s); pv = g.pad_in[s]; }
		else if (col == c) { idv = tok; pv = dead ? 1 : g.pad_in[s]; }
		else if (col == C) { idv = 0; pv = (dead || tok == 0 || g.pad_in[((size_t)b * H + src) * g.G + c]) ? 1 : 0; }
		else { idv = 0; pv = 1; }
		store_tok(g.ids_out, g.tok_bytes, o, idv);
		g.pad_out[o] = pv;
	}
	if (tid < H) {
		const bool dead = s_pick_flat[tid] < 0;
		const int src = dead ? 0 : s_pick_flat[tid] / V, tok = dead ? 0 : s_pick_flat[tid] - src * V;
		g.score_out[b * H + tid] = s_pick_raw[tid];
		g.score_normed[b * H + tid] = s_pick_val[tid];
		if (g.src_out) g.src_out[b * H + tid] = src;
		const bool nxt_pad = dead || (tok == 0) || g.pad_in[((size_t)b * H + src) * g.G + c] != 0;
		g.len_out[b * H + tid] = g.len_in[b * H + src] + ((C < g.G && !nxt_pad) ? 1.f : 0.f);
		a.node_out[b * H + tid] = dead ? -2 : (nxt_pad ? -1 : s_pick_next[tid]);
		if (two) a.vnode_out[b * H + tid] = dead ? -2 : (nxt_pad ? -1 : s_pick_vnext[tid]);
		if (!nxt_pad) atomicAdd(g.active + c, 1);
	}
}

struct GuidedGreedyArgs {
	GreedyArgs g;
	Trie t;
	int* node;  // [B] in/out
	int renorm;
};

__global__ __launch_bounds__(256) void greedy_step_guided_kernel(const GuidedGreedyArgs a) {
	const GreedyArgs& g = a.g;
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int c = g.step - 1;
	for (int b = blockIdx.x * 4 + w; b < g.B; b += gridDim.x * 4) {
		const bf16* row = g.logits + (size_t)b * g.ldl;
		const bool was_alive = g.alive[b] != 0.f;
		const int node = a.node[b];
		const int e0 = node >= 0 ? a.t.start[node] : 0, e1 = node >= 0 ? a.t.start[node + 1] : 0;
		// full-vocabulary log-sum-exp (tau = 1 for the loss; tau for the score unless renormalised over the allowed tokens)
		float mx = -INFINITY, se = 0.f, set = 0.f, sl = 0.f;
		for (int v = lane; v < g.V; v += 64) {
			const float x = (float)row[v];
			if (g.step_logits) g.step_logits[((size_t)b * g.G + c) * g.V + v] = x;
			if (x > mx) { se = se * __expf(mx - x) + 1.f; set = set * __expf((mx - x) * g.inv_temp) + 1.f; mx = x; }
			else { se += __expf(x - mx); set += __expf((x - mx) * g.inv_temp); }
			sl += x;
		}
		// allowed tokens: arg-max (children are sorted by token id, so the first maximum is the lowest id) and their own log-sum-exp
		float bestv = -INFINITY, amx = -INFINITY, ase = 0.f;
		int beste = 0x7fffffff;
		for (int e = e0 + lane; e < e1; e += 64) {
			const float x = (float)row[a.t.tok[e]];
			if (x > bestv) { bestv = x; beste = e; }
			const float xt = x * g.inv_temp;
			if (xt > amx) { ase = ase * __expf(amx - xt) + 1.f; amx = xt; } else ase += __expf(xt - amx);
		}
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			const float omx = __shfl_xor(mx, o, 64), ose = __shfl_xor(se, o, 64), oset = __shfl_xor(set, o, 64);
			const float nm = fmaxf(mx, omx);
			if (nm != -INFINITY) {
				se = se * __expf(mx - nm) + ose * __expf(omx - nm);
				set = set * __expf((mx - nm) * g.inv_temp) + oset * __expf((omx - nm) * g.inv_temp);
			}
			mx = nm;
			sl += __shfl_xor(sl, o, 64);
			const float obv = __shfl_xor(bestv, o, 64);
			const int obe = __shfl_xor(beste, o, 64);
			if (obv > bestv || (obv == bestv && obe < beste)) { bestv = obv; beste = obe; }
			const float oamx = __shfl_xor(amx, o, 64), oase = __shfl_xor(ase, o, 64);
			const float na = fmaxf(amx, oamx);
			if (na != -INFINITY) ase = ase * __expf(amx - na) + oase * __expf(oamx - na);
			amx = na;
		}
		if (lane == 0) {
			const bool has = beste != 0x7fffffff;
			const int tok = has ? a.t.tok[beste] : 0;
			g.pad[(size_t)b * g.G + c] = was_alive ? 0 : 1;
			store_tok(g.ids, g.tok_bytes, (size_t)b * g.G + c, tok);
			if (was_alive) {
				const float lse = mx + __logf(se);
				const float lse_t = a.renorm ? amx + __logf(ase) : mx * g.inv_temp + __logf(set);
				g.score[b] += bestv * g.inv_temp - lse_t;
				float l = lse - bestv;
				if (g.smoothing > 0.f) l = (1.f - g.smoothing) * l + g.smoothing * (lse - sl / (float)g.V);
				g.nll[b] += l;
				g.count[b] += 1.f;
			}
			const bool still = was_alive && has && tok != 0;
			g.alive[b] = still ? 1.f : 0.f;
			a.node[b] = still ? a.t.next[beste] : -1;
			if (still) atomicAdd(g.active + c, 1);
		}
	}
}

}  // namespace

static int beam_step_guided_launch(const void* logits_bf16, int ldl, int V, int B, int H, int G, int step, const void* ids_in, void* ids_out, int tok_bytes, const uint8_t* pad_in,
                                   uint8_t* pad_out, const float* score_in, float* score_out, float* score_normed, const float* len_in, float* len_out, int* active, int* src_out,
                                   const int* node_in, int* node_out, Trie t, float prior_scale, int renorm, Trie v, const int* vnode_in, int* vnode_out, float temperature,
                                   float length_alpha, hipStream_t stream) {
	NOVIC_CHECK(logits_bf16 && ids_in && ids_out && pad_in && pad_out && score_in && score_out && score_normed && len_in && len_out && active && node_in && node_out && t.start &&
	            t.tok && t.next, "novic_beam_step_guided: null pointer");
	NOVIC_CHECK(!v.start || (v.tok && v.next && v.logprior && vnode_in && vnode_out), "novic_beam_step_guided_vocab: incomplete vocabulary trie");
	NOVIC_CHECK(H >= 1 && H <= 32 && step >= 1 && step <= G && V >= 2 && temperature > 0.f, "novic_beam_step_guided: bad beam width / step / vocabulary / temperature");
	NOVIC_CHECK(tok_bytes == 4 || tok_bytes == 8, "novic_beam_step_guided: tok_bytes must be 4 or 8");
	NOVIC_CHECK(ldl % 8 == 0 && ((uintptr_t)logits_bf16 & 15) == 0, "novic_beam_step_guided: logits rows must be 16-byte aligned (ldl a multiple of 8)");
	if (B <= 0) return 0;
	GuidedBeamArgs a = {{(const bf16*)logits_bf16, ldl, V, B, H, G, step, ids_in, ids_out, tok_bytes, pad_in, pad_out, score_in, score_out, score_normed, len_in, len_out, active,
	                     src_out, 1.f / temperature, length_alpha},
	                    t, node_in, node_out, renorm, (t.logprior || v.start) ? prior_scale : 0.f, v, vnode_in, vnode_out};
	hipLaunchKernelGGL(beam_step_guided_kernel, dim3(B), dim3(256), GB_CACHE * 8, stream, a);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_beam_step_guided(const void* logits_bf16, int ldl, int V, int B, int H, int G, int step, const void* ids_in, void* ids_out, int tok_bytes,
                                      const uint8_t* pad_in, uint8_t* pad_out, const float* score_in, float* score_out, float* score_normed, const float* len_in, float* len_out,
                                      int* active, int* src_out, const int* node_in, int* node_out, const int* trie_start, const int* trie_tok, const int* trie_next,
                                      const float* trie_logprior, float prior_scale, int renorm, float temperature, float length_alpha, hipStream_t stream) {
	return beam_step_guided_launch(logits_bf16, ldl, V, B, H, G, step, ids_in, ids_out, tok_bytes, pad_in, pad_out, score_in, score_out, score_normed, len_in, len_out, active, src_out,
	                               node_in, node_out, Trie{trie_start, trie_tok, trie_next, trie_logprior}, prior_scale, renorm, Trie{nullptr, nullptr, nullptr, nullptr}, nullptr, nullptr,
	                               temperature, length_alpha, stream);
}

extern "C" int novic_beam_step_guided_vocab(const void* logits_bf16, int ldl, int V, int B, int H, int G, int step, const void* ids_in, void* ids_out, int tok_bytes,
                                            const uint8_t* pad_in, uint8_t* pad_out, const float* score_in, float* score_out, float* score_normed, const float* len_in, float* len_out,
                                            int* active, int* src_out, const int* node_in, int* node_out, const int* trie_start, const int* trie_tok, const int* trie_next,
                                            const int* vnode_in, int* vnode_out, const int* vocab_start, const int* vocab_tok, const int* vocab_next, const float* vocab_logprior,
                                            float prior_scale, int renorm, float temperature, float length_alpha, hipStream_t stream) {
	return beam_step_guided_launch(logits_bf16, ldl, V, B, H, G, step, ids_in, ids_out, tok_bytes, pad_in, pad_out, score_in, score_out, score_normed, len_in, len_out, active, src_out,
	                               node_in, node_out, Trie{trie_start, trie_tok, trie_next, nullptr}, prior_scale, renorm, Trie{vocab_start, vocab_tok, vocab_next, vocab_logprior},
	                               vnode_in, vnode_out, temperature, length_alpha, stream);
}

extern "C" int novic_greedy_step_guided(const void* logits_bf16, int ldl, int V, int B, int G, int step, void* ids, int tok_bytes, uint8_t* pad, float* alive, float* score,
                                        float* nll, float* count, int* active, float* step_logits, int* node, const int* trie_start, const int* trie_tok, const int* trie_next,
                                        int renorm, float temperature, float label_smoothing, hipStream_t stream) {
	NOVIC_CHECK(logits_bf16 && ids && pad && alive && score && nll && count && active && node && trie_start && trie_tok && trie_next, "novic_greedy_step_guided: null pointer");
	NOVIC_CHECK(step >= 1 && step <= G && V >= 2 && temperature > 0.f && (tok_bytes == 4 || tok_bytes == 8), "novic_greedy_step_guided: bad arguments");
	if (B <= 0) return 0;
	GuidedGreedyArgs a = {{(const bf16*)logits_bf16, ldl, V, B, G, step, ids, tok_bytes, pad, alive, score, nll, count, active, step_logits, 1.f / temperature, label_smoothing},
	                      {trie_start, trie_tok, trie_next, nullptr}, node, renorm};
	int grid = (B + 3) / 4;
	if (grid > 4096) grid = 4096;
	hipLaunchKernelGGL(greedy_step_guided_kernel, dim3(grid), dim3(256), 0, stream, a);
	NOVIC_LAUNCH_CHECK();
	return 0;
}
