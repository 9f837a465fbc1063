// Fused train-time embedding noise: (mean shift + renorm) -> Gaussian add / random rotation / mixture -> L2 renorm, in place.
// reference: embedding_noise.py:72-75 (GaussElem), :78-95 (GaussVec), :105-112 (rotation), :131-132 / :151-152 (angle draws),
//            :169-172 (mixture), train.py:1263-1265 (mean shift).
// One wave per row, the row stays in registers: 1 read + 1 write of B x F fp32 = 8*B*F algorithmic bytes (HBM-bound);
// the reference issues ~12 elementwise/reduce kernels and 4 RNG launches for the same result.
// Randomness: Philox4x32-10 keyed by (seed), counter (row, chunk, stream, offset) + Box-Muller, or caller-injected tensors
// (z1/z2/r/u_mix), which make the kernel comparable value-for-value with the oracle.
#include "common.hpp"
#include "novic_hip.h"

namespace {

struct NoiseArgs {
	float* embed;
	int B, F, mode;
	float vec_norm, angle_lo, angle_hi, angle_std, mix_ratio;  // radians
	uint32_t seed_lo, seed_hi, offset;
	const float* z1;  // [B][F] injected N(0,1): Gaussian-add noise / rotation direction
	const float* z2;  // [B][F] injected N(0,1): rotation direction of the mixture
	const float* r;   // [B] injected per-row draw: N(0,1) (GaussVec, GaussAngle) or U[0,1) (UniformAngle, mixture angle)
	const float* u_mix;  // [B] injected U[0,1)
	const float* shift;  // [F] or null
};

template <int NC>
__device__ __forceinline__ float row_dot(const float (&a)[NC][4], const float (&b)[NC][4]) {
	float s = 0.f;
#pragma unroll
	for (int c = 0; c < NC; ++c)
#pragma unroll
		for (int i = 0; i < 4; ++i) s += a[c][i] * b[c][i];
	return wave_sum(s);
}
template <int NC>
__device__ __forceinline__ void row_normalize(float (&a)[NC][4]) {
	const float inv = 1.f / fmaxf(sqrtf(row_dot<NC>(a, a)), 1e-12f);
#pragma unroll
	for (int c = 0; c < NC; ++c)
#pragma unroll
		for (int i = 0; i < 4; ++i) a[c][i] *= inv;
}

__device__ __forceinline__ void normal4(const NoiseArgs& g, uint32_t row, uint32_t chunk, uint32_t stream, float (&z)[4]) {
	const Philox4 p = philox4x32_10(row, chunk, stream, g.offset, g.seed_lo, g.seed_hi);
	const float u0 = fmaxf(u01(p.x), 5.96e-8f), u1 = u01(p.y), u2 = fmaxf(u01(p.z), 5.96e-8f), u3 = u01(p.w);
	const float r0 = sqrtf(-2.f * __logf(u0)), r1 = sqrtf(-2.f * __logf(u2));
	float s0, c0, s1, c1;
	__sincosf(6.283185307179586f * u1, &s0, &c0);
	__sincosf(6.283185307179586f * u3, &s1, &c1);
	z[0] = r0 * c0; z[1] = r0 * s0; z[2] = r1 * c1; z[3] = r1 * s1;
}
__device__ __forceinline__ void row_scalars(const NoiseArgs& g, uint32_t row, float& n0, float& u0, float& u1) {
	const Philox4 p = philox4x32_10(row, 0xffffffffu, 7u, g.offset, g.seed_lo, g.seed_hi);
	const float a = fmaxf(u01(p.x), 5.96e-8f);
	n0 = sqrtf(-2.f * __logf(a)) * __cosf(6.283185307179586f * u01(p.y));
	u0 = u01(p.z);
	u1 = u01(p.w);
}

template <int NC>
__device__ __forceinline__ void load_z(const NoiseArgs& g, const float* inj, int row, uint32_t stream, int lane, float (&z)[NC][4]) {
#pragma unroll
	for (int c = 0; c < NC; ++c) {
		const int e = c * 256 + lane * 4;
		if (e < g.F) {
			if (inj) {
				const f32x4 t = *reinterpret_cast<const f32x4*>(inj + (size_t)row * g.F + e);
				z[c][0] = t[0]; z[c][1] = t[1]; z[c][2] = t[2]; z[c][3] = t[3];
			} else {
				normal4(g, (uint32_t)row, (uint32_t)(e >> 2), stream, z[c]);
			}
		} else {
			z[c][0] = z[c][1] = z[c][2] = z[c][3] = 0.f;
		}
	}
}

template <int NC>
__device__ __forceinline__ void rotate_row(float (&x)[NC][4], float (&z)[NC][4], float angle) {
	const float d = row_dot<NC>(x, z);
#pragma unroll
	for (int c = 0; c < NC; ++c)
#pragma unroll
		for (int i = 0; i < 4; ++i) z[c][i] -= x[c][i] * d;
	row_normalize<NC>(z);
	float sn, cs;
	sincosf(angle, &sn, &cs);
#pragma unroll
	for (int c = 0; c < NC; ++c)
#pragma unroll
		for (int i = 0; i < 4; ++i) x[c][i] = x[c][i] * cs + z[c][i] * sn;
	row_normalize<NC>(x);
}

template <int NC>
__global__ __launch_bounds__(256) void noise_kernel(const NoiseArgs g) {
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (int row = blockIdx.x * 4 + w; row < g.B; row += gridDim.x * 4) {
		float x[NC][4], z[NC][4];
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int e = c * 256 + lane * 4;
			if (e < g.F) {
				const f32x4 t = *reinterpret_cast<const f32x4*>(g.embed + (size_t)row * g.F + e);
				x[c][0] = t[0]; x[c][1] = t[1]; x[c][2] = t[2]; x[c][3] = t[3];
			} else {
				x[c][0] = x[c][1] = x[c][2] = x[c][3] = 0.f;
			}
		}
		if (g.shift) {
#pragma unroll
			for (int c = 0; c < NC; ++c) {
				const int e = c * 256 + lane * 4;
				if (e < g.F) {
					const f32x4 t = *reinterpret_cast<const f32x4*>(g.shift + e);
#pragma unroll
					for (int i = 0; i < 4; ++i) x[c][i] += t[i];
				}
			}
			row_normalize<NC>(x);
		}
		float n0, u0, u1;
		row_scalars(g, (uint32_t)row, n0, u0, u1);
		if (g.r) { n0 = g.r[row]; u0 = g.r[row]; }
		if (g.u_mix) u1 = g.u_mix[row];
		int mode = g.mode;
		if (mode == NOVIC_NOISE_GAUSS_ELEM_UNIFORM_ANGLE) mode = (u1 < g.mix_ratio) ? -NOVIC_NOISE_UNIFORM_ANGLE : NOVIC_NOISE_GAUSS_ELEM;
		if (mode == NOVIC_NOISE_GAUSS_ELEM) {
			load_z<NC>(g, g.z1, row, 1u, lane, z);
			const float sd = g.vec_norm / sqrtf((float)g.F);
#pragma unroll
			for (int c = 0; c < NC; ++c)
#pragma unroll
				for (int i = 0; i < 4; ++i) x[c][i] += z[c][i] * sd;
			row_normalize<NC>(x);
		} else if (mode == NOVIC_NOISE_GAUSS_VEC) {
			load_z<NC>(g, g.z1, row, 1u, lane, z);
			row_normalize<NC>(z);
			const float k = n0 * g.vec_norm;
#pragma unroll
			for (int c = 0; c < NC; ++c)
#pragma unroll
				for (int i = 0; i < 4; ++i) x[c][i] += z[c][i] * k;
			row_normalize<NC>(x);
		} else if (mode == NOVIC_NOISE_GAUSS_ANGLE) {
			load_z<NC>(g, g.z1, row, 1u, lane, z);
			rotate_row<NC>(x, z, fminf(fmaxf(n0 * g.angle_std, -g.angle_hi), g.angle_hi));
		} else if (mode == NOVIC_NOISE_UNIFORM_ANGLE) {
			load_z<NC>(g, g.z1, row, 1u, lane, z);
			rotate_row<NC>(x, z, g.angle_lo + u0 * (g.angle_hi - g.angle_lo));
		} else if (mode == -NOVIC_NOISE_UNIFORM_ANGLE) {  // rotation branch of the mixture: its own direction stream
			load_z<NC>(g, g.z2, row, 2u, lane, z);
			rotate_row<NC>(x, z, g.angle_lo + u0 * (g.angle_hi - g.angle_lo));
		}
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int e = c * 256 + lane * 4;
			if (e < g.F) *reinterpret_cast<f32x4*>(g.embed + (size_t)row * g.F + e) = (f32x4){x[c][0], x[c][1], x[c][2], x[c][3]};
		}
	}
}

}  // namespace

extern "C" int novic_noise_fused(float* embed, int B, int F, int mode, float vec_norm, float angle_min_rad, float angle_max_rad, float angle_std_rad, float mix_ratio,
                                 uint64_t seed, uint32_t offset, const float* inj_z1, const float* inj_z2, const float* inj_row, const float* inj_mix,
                                 const float* mean_shift, hipStream_t stream) {
	NOVIC_CHECK(embed, "novic_noise_fused: null pointer");
	NOVIC_CHECK(F % 4 == 0 && F > 0 && F <= 2048, "novic_noise_fused: F must be a multiple of 4 and <= 2048");
	NOVIC_CHECK(mode >= NOVIC_NOISE_NONE && mode <= NOVIC_NOISE_GAUSS_ELEM_UNIFORM_ANGLE, "novic_noise_fused: unknown mode");
	if (B <= 0) return 0;
	NoiseArgs g = {embed, B, F, mode, vec_norm, angle_min_rad, angle_max_rad, angle_std_rad, mix_ratio, (uint32_t)seed, (uint32_t)(seed >> 32), offset,
	               inj_z1, inj_z2, inj_row, inj_mix, mean_shift};
	int grid = (B + 3) / 4;
	if (grid > 4096) grid = 4096;
	switch ((F + 255) / 256) {
#define NOVIC_NOISE_CASE(N) case N: hipLaunchKernelGGL((noise_kernel<N>), dim3(grid), dim3(256), 0, stream, g); break;
		NOVIC_NOISE_CASE(1) NOVIC_NOISE_CASE(2) NOVIC_NOISE_CASE(3) NOVIC_NOISE_CASE(4) NOVIC_NOISE_CASE(5) NOVIC_NOISE_CASE(6) NOVIC_NOISE_CASE(7) NOVIC_NOISE_CASE(8)
#undef NOVIC_NOISE_CASE
	}
	NOVIC_LAUNCH_CHECK();
	return 0;
}
