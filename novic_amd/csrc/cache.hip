// On-device batch assembly from an HBM-resident embedding cache (reference embedding_cache.py:690-723 get_samples + :832-895 batch rules):
//   row(b) = (start + b) % N                         contiguous slice with per-epoch rotation and wrap
//   embed[b]      = embeds[row]                      F floats
//   target[b][m]  = token_table[ids[row][m]][:C]     (mask likewise), m < M
//   weight[b][m]  = weights[row][m] | L1-normalised over the kept M | 1
// One wave per batch row; HBM-bound: 8*F + M*C*(tok_bytes+1)*2 + 8*M bytes per row.
#include "common.hpp"
#include "novic_hip.h"

namespace {

// the batches of one launch: batch g's row 0 is cache row s[g]; one batch for novic_cache_gather, an optimizer step's micro-batches for novic_cache_gather_group
constexpr int GATHER_MAX_GROUPS = 32;
struct GatherStarts {
	long long s[GATHER_MAX_GROUPS];
};

__global__ __launch_bounds__(256) void cache_gather_kernel(const float* __restrict__ embeds, const int* __restrict__ ids, const void* __restrict__ tok, int tok_bytes,
                                                           const uint8_t* __restrict__ msk, const float* __restrict__ wts, const GatherStarts starts, int Bg, int B, long long N, int F, int Mf,
                                                           int Cf, int M, int C, float* __restrict__ o_embed, void* __restrict__ o_target, uint8_t* __restrict__ o_mask,
                                                           float* __restrict__ o_weight, int weight_mode, long long staged_row0) {
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (int b = blockIdx.x * 4 + w; b < B; b += gridDim.x * 4) {
		const int grp = b / Bg;  // (output row b = row b - grp * Bg of batch grp: the batches' outputs lie one behind the other)
		const long long row = (starts.s[grp] + (b - grp * Bg)) % N;
		const long long erow = staged_row0 >= 0 ? staged_row0 + b : row;  // streaming loader: `embeds` is a staged slab that already holds the batch's rows in order
		for (int e = lane * 4; e < F; e += 256) {
			if (e + 4 <= F) *reinterpret_cast<f32x4*>(o_embed + (size_t)b * F + e) = *reinterpret_cast<const f32x4*>(embeds + (size_t)erow * F + e);
			else
				for (int i = e; i < F; ++i) o_embed[(size_t)b * F + i] = embeds[(size_t)erow * F + i];
		}
		if (!ids) continue;
		for (int i = lane; i < M * C; i += 64) {
			const int m = i / C, c = i - m * C;
			const int noun = ids[(size_t)row * Mf + m];
			const size_t src = (size_t)noun * Cf + c, dst = ((size_t)b * M + m) * C + c;
			if (tok_bytes == 8) ((long long*)o_target)[dst] = ((const long long*)tok)[src];
			else ((int*)o_target)[dst] = ((const int*)tok)[src];
			if (o_mask) o_mask[dst] = msk[src];
		}
		if (o_weight) {
			float v = (lane < M) ? wts[(size_t)row * Mf + lane] : 0.f;
			if (weight_mode == 1) {
				const float s = wave_sum(fabsf(v));
				v = v / fmaxf(s, 1e-12f);
			} else if (weight_mode == 2) {
				v = 1.f;
			}
			if (lane < M) o_weight[(size_t)b * M + lane] = v;
		}
	}
}

}  // namespace

extern "C" int novic_cache_gather(const float* embeds, const int* target_ids, const void* token_table, int tok_bytes, const uint8_t* mask_table, const float* weights,
                                  int64_t start, int B, int64_t N, int F, int M_file, int C_file, int M, int C, float* out_embed, void* out_target, uint8_t* out_mask,
                                  float* out_weight, int weight_mode, int64_t staged_row0, hipStream_t stream) {
	NOVIC_CHECK(embeds && out_embed, "novic_cache_gather: null pointer");
	NOVIC_CHECK(B >= 0 && N >= 1 && start >= 0 && F >= 1, "novic_cache_gather: bad shape");
	NOVIC_CHECK(!target_ids || (token_table && out_target && M >= 0 && M <= M_file && M <= 64 && C >= 0 && C <= C_file && (tok_bytes == 4 || tok_bytes == 8)),
	            "novic_cache_gather: bad target arguments");
	NOVIC_CHECK(!out_mask || mask_table, "novic_cache_gather: mask output without mask table");
	NOVIC_CHECK(!out_weight || weights, "novic_cache_gather: weight output without weights");
	NOVIC_CHECK(weight_mode >= 0 && weight_mode <= 2, "novic_cache_gather: weight_mode must be 0 (copy), 1 (L1 normalise) or 2 (ones)");
	if (B == 0) return 0;
	int grid = (B + 3) / 4;
	if (grid > 4096) grid = 4096;
	GatherStarts st = {};
	st.s[0] = (long long)start;
	hipLaunchKernelGGL(cache_gather_kernel, dim3(grid), dim3(256), 0, stream, embeds, target_ids, token_table, tok_bytes, mask_table, weights, st, B, B, (long long)N, F,
	                   M_file, C_file, M, C, out_embed, out_target, out_mask, out_weight, weight_mode, (long long)staged_row0);
	NOVIC_LAUNCH_CHECK();
	return 0;
}

extern "C" int novic_cache_gather_group(const float* embeds, const int* target_ids, const void* token_table, int tok_bytes, const uint8_t* mask_table, const float* weights,
                                        const int64_t* starts, int groups, int B_each, int64_t N, int F, int M_file, int C_file, int M, int C, float* out_embed, void* out_target,
                                        uint8_t* out_mask, float* out_weight, int weight_mode, int64_t staged_row0, hipStream_t stream) {
	NOVIC_CHECK(embeds && out_embed && starts, "novic_cache_gather_group: null pointer");
	NOVIC_CHECK(groups >= 1 && groups <= GATHER_MAX_GROUPS, "novic_cache_gather_group: 1 .. 32 batches per launch");
	NOVIC_CHECK(B_each >= 1 && N >= 1 && F >= 1 && (long long)groups * B_each <= 0x7fffffffLL, "novic_cache_gather_group: bad shape");
	NOVIC_CHECK(!target_ids || (token_table && out_target && M >= 0 && M <= M_file && M <= 64 && C >= 0 && C <= C_file && (tok_bytes == 4 || tok_bytes == 8)),
	            "novic_cache_gather_group: bad target arguments");
	NOVIC_CHECK(!out_mask || mask_table, "novic_cache_gather_group: mask output without mask table");
	NOVIC_CHECK(!out_weight || weights, "novic_cache_gather_group: weight output without weights");
	NOVIC_CHECK(weight_mode >= 0 && weight_mode <= 2, "novic_cache_gather_group: weight_mode must be 0 (copy), 1 (L1 normalise) or 2 (ones)");
	GatherStarts st = {};
	for (int g = 0; g < groups; ++g) {
		NOVIC_CHECK(starts[g] >= 0, "novic_cache_gather_group: negative start row");
		st.s[g] = (long long)starts[g];
	}
	const int B = groups * B_each;
	int grid = (B + 3) / 4;
	if (grid > 4096) grid = 4096;
	hipLaunchKernelGGL(cache_gather_kernel, dim3(grid), dim3(256), 0, stream, embeds, target_ids, token_table, tok_bytes, mask_table, weights, st, B_each, B, (long long)N, F,
	                   M_file, C_file, M, C, out_embed, out_target, out_mask, out_weight, weight_mode, (long long)staged_row0);
	NOVIC_LAUNCH_CHECK();
	return 0;
}
