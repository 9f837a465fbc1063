// generate_all (reference embedding_decoder.py:1043-1079): teacher-forced log-probability of every guide target for every sample, then the top-k targets per
// sample.  The forward pass is the ordinary decoder forward over (sample x target-chunk) sequences; these two kernels are what follows it:
//
//   novic_score_targets  score[b][w0 + h] = sum_t log-softmax(logits[(b, h), t] / tau)[target[h][t]]   over the unpadded positions of target h,
//                        the soft-max taken over the whole vocabulary or (guide_renorm) over the tokens that targets sharing the prefix allow
//                        = the children of the target's trie node at that position (:1000-1007, :1062-1068)
//   novic_topk_rows      (value, index) of the k largest (score - adjust) * scale per row, ties towards the lower index (:1070-1079; adjust = vocabulary
//                        prior sums, scale = length normalisation, both per target)
//
// Both are HBM-bound: the first reads every logit once (2 V bytes per sequence position), the second reads B x W floats k-independently.
#include "common.hpp"
#include "novic_hip.h"

namespace {

struct ScoreArgs {
	const bf16* logits;     // [A * T][ldl], A = B * Hc sequences, sequence a = b * Hc + h
	const void* targets;    // [Hc][T] token ids of the chunk's targets
	int tok_bytes;
	const uint8_t* pad;     // [Hc][T] 1 = position after the target's END
	const int* node;        // [Hc][T] trie node of the prefix before position t (guide_renorm) or null
	const int* trie_start;  // CSR children of the guide trie (guide_renorm)
	const int* trie_tok;
	float* out;             // [B][ldo]; element (b, w0 + h)
	int B, Hc, T, V, ldl, ldo, w0;
	float inv_temp;
};

// one workgroup per sequence; wave w takes positions w, w + 4, ...
__global__ __launch_bounds__(256) void score_targets_kernel(const ScoreArgs g) {
	__shared__ float part[4];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int a = blockIdx.x, b = a / g.Hc, h = a - b * g.Hc;
	float total = 0.f;
	for (int t = w; t < g.T; t += 4) {
		if (g.pad[h * g.T + t]) continue;  // wave-uniform
		const bf16* row = g.logits + ((size_t)a * g.T + t) * g.ldl;
		const long long tgt = g.tok_bytes == 8 ? ((const long long*)g.targets)[h * g.T + t] : (long long)((const int*)g.targets)[h * g.T + t];
		float mx = -INFINITY, se = 0.f;
		auto add = [&](float x) {
			if (x > mx) { se = se * __expf(mx - x) + 1.f; mx = x; }
			else se += __expf(x - mx);
		};
		if (g.node) {
			const int nd = g.node[h * g.T + t];
			const int e0 = g.trie_start[nd], e1 = g.trie_start[nd + 1];
			for (int e = e0 + lane; e < e1; e += 64) add((float)row[g.trie_tok[e]] * g.inv_temp);
		} else {
			for (int v0 = lane * 8; v0 < g.V; v0 += 64 * 8) {
				const bf16x8 xs = *reinterpret_cast<const bf16x8*>(row + v0);  // ldl is a multiple of 8: the row padding makes the last load legal
#pragma unroll
				for (int k = 0; k < 8; ++k)
					if (v0 + k < g.V) add((float)xs[k] * g.inv_temp);
			}
		}
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			const float omx = __shfl_xor(mx, o, 64), ose = __shfl_xor(se, o, 64);
			const float nm = fmaxf(mx, omx);
			if (nm != -INFINITY) se = se * __expf(mx - nm) + ose * __expf(omx - nm);
			mx = nm;
		}
		total += (float)row[tgt] * g.inv_temp - (mx + __logf(se));
	}
	if (lane == 0) part[w] = total;
	__syncthreads();
	if (threadIdx.x == 0) g.out[(size_t)b * g.ldo + g.w0 + h] = (part[0] + part[1]) + (part[2] + part[3]);
}

// one workgroup per row: every thread caches the best element of its strided subset; k rounds of block arg-max, after each of which only the winner
// re-scans its own subset for its next offer (same scheme as beam_step_kernel)
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ scores, int W, int lds, const float* __restrict__ adjust, float adjust_scale,
                                                        const float* __restrict__ scale, int K, float* __restrict__ out_val, int* __restrict__ out_idx) {
	__shared__ float s_val[4];
	__shared__ int s_idx[4];
	__shared__ float s_pv;
	__shared__ int s_pi;
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const float* row = scores + (size_t)blockIdx.x * lds;
	auto value = [&](int i) -> float {
		float v = row[i];
		if (adjust) v -= adjust_scale * adjust[i];
		if (scale) v *= scale[i];
		return v;
	};
	auto better = [](float v, int i, float bv, int bi) { return bi < 0 || v > bv || (v == bv && i < bi); };
	auto scan = [&](float pv, int pi, float& bv, int& bi) {  // best of the thread's subset strictly after (pv, pi) in (value desc, index asc) order
		bv = -INFINITY; bi = -1;
		for (int i = tid; i < W; i += 256) {
			const float v = value(i);
			const bool after = pi < 0 || v < pv || (v == pv && i > pi);
			if (after && !(v != v) && better(v, i, bv, bi)) { bv = v; bi = i; }
		}
	};
	float my_v;
	int my_i;
	scan(0.f, -1, my_v, my_i);
	for (int r = 0; r < K; ++r) {
		float bv = my_v;
		int bi = my_i;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			const float ov = __shfl_xor(bv, o, 64);
			const int oi = __shfl_xor(bi, o, 64);
			if (oi >= 0 && better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
		}
		if (lane == 0) { s_val[w] = bv; s_idx[w] = bi; }
		__syncthreads();
		if (tid == 0) {
			float fv = s_val[0];
			int fi = s_idx[0];
			for (int k = 1; k < 4; ++k)
				if (s_idx[k] >= 0 && better(s_val[k], s_idx[k], fv, fi)) { fv = s_val[k]; fi = s_idx[k]; }
			s_pv = fv; s_pi = fi;
			out_val[(size_t)blockIdx.x * K + r] = fi >= 0 ? fv : -INFINITY;
			out_idx[(size_t)blockIdx.x * K + r] = fi >= 0 ? fi : 0;
		}
		__syncthreads();
		const int pi = s_pi;
		if (pi >= 0 && (pi & 255) == tid) scan(s_pv, pi, my_v, my_i);  // element i belongs to thread i % 256
		__syncthreads();
	}
}

// Guided teacher-forced correctness (reference embedding_decoder.py:756-763): at position t of sequence a the prediction is the arg-max of the
// logits over the tokens that guide nouns consistent with the TARGET's own prefix target[a][:t] have at column t = the children of the trie
// node reached by walking the target (dead once the target leaves the trie: the reference then arg-maxes an all -inf row, i.e. predicts 0).
// One wave per sequence walks its target; correct[a][t] = (prediction == target[a][t]) and the position is not padding.
struct GuidedCorrectArgs {
	const bf16* logits;      // [A * T][ldl]
	const void* targets;     // [A][tok_ld]
	int tok_bytes, tok_ld;
	const uint8_t* out_pad;  // [A][C] or null (column c of the target <-> out_pad[a][c])
	const int* trie_start;
	const int* trie_tok;
	const int* trie_next;
	uint8_t* correct;        // [A * T]
	int A, T, ldl;
};

__global__ __launch_bounds__(256) void guided_correct_kernel(const GuidedCorrectArgs g) {
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (int a = blockIdx.x * 4 + w; a < g.A; a += gridDim.x * 4) {
		int node = 0;  // root
		for (int t = 0; t < g.T; ++t) {
			const long long tgt = g.tok_bytes == 8 ? ((const long long*)g.targets)[(size_t)a * g.tok_ld + t] : (long long)((const int*)g.targets)[(size_t)a * g.tok_ld + t];
			const bool padded = g.out_pad && g.out_pad[(size_t)a * g.T + t];
			const bf16* row = g.logits + ((size_t)a * g.T + t) * g.ldl;
			float bv = -INFINITY;
			int be = 0x7fffffff, child = -2;
			if (node >= 0) {
				const int e0 = g.trie_start[node], e1 = g.trie_start[node + 1];
				for (int e = e0 + lane; e < e1; e += 64) {
					const int tk = g.trie_tok[e];
					const float x = (float)row[tk];
					if (x > bv) { bv = x; be = e; }  // children are sorted by token: the first maximum within a lane is the lowest token
					if (tk == (int)tgt) child = g.trie_next[e];
				}
			}
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) {
				const float ov = __shfl_xor(bv, o, 64);
				const int oe = __shfl_xor(be, o, 64), oc = __shfl_xor(child, o, 64);
				if (ov > bv || (ov == bv && oe < be)) { bv = ov; be = oe; }
				child = max(child, oc);  // at most one lane found the edge (>= -1), the others hold -2
			}
			if (lane == 0) {
				const int pred = be != 0x7fffffff ? g.trie_tok[be] : 0;
				g.correct[(size_t)a * g.T + t] = (!padded && pred == (int)tgt) ? 1 : 0;
			}
			node = (node >= 0 && child >= 0) ? child : -2;  // END edge (-1) or no edge: nothing consistent is left
		}
	}
}

}  // namespace

extern "C" int novic_score_targets(const void* logits_bf16, int ldl, int V, const void* targets, int tok_bytes, const uint8_t* pad, const int* node, const int* trie_start,
                                   const int* trie_tok, float* out, int ldo, int w0, int B, int Hc, int T, float temperature, hipStream_t stream) {
	NOVIC_CHECK(logits_bf16 && targets && pad && out, "novic_score_targets: null pointer");
	NOVIC_CHECK(!node || (trie_start && trie_tok), "novic_score_targets: renormalised scoring needs the trie arrays");
	NOVIC_CHECK((tok_bytes == 4 || tok_bytes == 8) && V >= 2 && T >= 1 && temperature > 0.f, "novic_score_targets: bad arguments");
	NOVIC_CHECK(ldl % 8 == 0 && ((uintptr_t)logits_bf16 & 15) == 0, "novic_score_targets: logits rows must be 16-byte aligned (ldl a multiple of 8)");
	if (B <= 0 || Hc <= 0) return 0;
	ScoreArgs g = {(const bf16*)logits_bf16, targets, tok_bytes, pad, node, trie_start, trie_tok, out, B, Hc, T, V, ldl, ldo, w0, 1.f / temperature};
	hipLaunchKernelGGL(score_targets_kernel, dim3(B * Hc), dim3(256), 0, stream, g);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_topk_rows(const float* scores, int B, int W, int lds, const float* adjust, float adjust_scale, const float* scale, int K, float* out_val, int* out_idx,
                               hipStream_t stream) {
	NOVIC_CHECK(scores && out_val && out_idx, "novic_topk_rows: null pointer");
	NOVIC_CHECK(K >= 1 && K <= W && lds >= W, "novic_topk_rows: need 1 <= K <= W <= lds");
	if (B <= 0) return 0;
	hipLaunchKernelGGL(topk_rows_kernel, dim3(B), dim3(256), 0, stream, scores, W, lds, adjust, adjust_scale, scale, K, out_val, out_idx);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_guided_correct(const void* logits_bf16, int ldl, const void* targets, int tok_bytes, int tok_ld, const uint8_t* out_pad, const int* trie_start,
                                    const int* trie_tok, const int* trie_next, uint8_t* correct, int A, int T, hipStream_t stream) {
	NOVIC_CHECK(logits_bf16 && targets && trie_start && trie_tok && trie_next && correct, "novic_guided_correct: null pointer");
	NOVIC_CHECK((tok_bytes == 4 || tok_bytes == 8) && T >= 1 && tok_ld >= T, "novic_guided_correct: bad arguments");
	if (A <= 0) return 0;
	GuidedCorrectArgs g = {(const bf16*)logits_bf16, targets, tok_bytes, tok_ld, out_pad, trie_start, trie_tok, trie_next, correct, A, T, ldl};
	int grid = (A + 3) / 4;
	if (grid > 16384) grid = 16384;
	hipLaunchKernelGGL(guided_correct_kernel, dim3(grid), dim3(256), 0, stream, g);
	NOVIC_LAUNCH_CHECK();
	return 0;
}
