// Per-element GEMM epilogues shared by the GEMM kernels: v = 4 consecutive columns n..n+3 of output row m.
#pragma once
#include "common.hpp"
#include "novic_hip.h"

namespace {

// ---- epilogues: v = 4 consecutive n of row m; one vector access per operand when the 4 columns are in range ----
__device__ __forceinline__ void st_bf16x4(bf16* p, const float (&v)[4], bool full, int nrem) {
	if (full) {
		bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
		__builtin_nontemporal_store(o, reinterpret_cast<bf16x4*>(p));  // GEMM outputs stream out: keep the operand panels in L2 instead
	} else {
		for (int r = 0; r < nrem; ++r) p[r] = (bf16)v[r];
	}
}
__device__ __forceinline__ void st_f32x4(float* p, const float (&v)[4], bool full, int nrem) {
	if (full) __builtin_nontemporal_store((f32x4){v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(p));
	else
		for (int r = 0; r < nrem; ++r) p[r] = v[r];
}
__device__ __forceinline__ void ld_f32x4(const float* p, float (&v)[4], bool full, int nrem) {
	if (full) {
		const f32x4 t = *reinterpret_cast<const f32x4*>(p);
		v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
	} else {
		for (int r = 0; r < 4; ++r) v[r] = r < nrem ? p[r] : 0.f;
	}
}
__device__ __forceinline__ void ld_bf16x4(const bf16* p, float (&v)[4], bool full, int nrem) {
	if (full) {
		const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
		v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
	} else {
		for (int r = 0; r < 4; ++r) v[r] = r < nrem ? (float)p[r] : 0.f;
	}
}

// RESID_F16, one element: the GEMM output (fp32 accumulator + bias) rounded to half as a half-precision linear returns it, added to the half residual, the sum rounded to
// half -- one definition for every kernel that has this epilogue (contraction off: two roundings, in this order)
__device__ __forceinline__ f16 resid_f16_elem(f16 resid, float acc_plus_bias) {
#pragma clang fp contract(off)
	const float lin = (float)(f16)acc_plus_bias;
	return (f16)((float)resid + lin);
}

// Element arithmetic of the GELU epilogues, shared by every kernel that implements them (gemm.hip, gemm256.hip, skinny.hip).  Contraction is off
// inside: a * s * g'(h) sits near bf16 rounding ties often enough that two kernels which fuse or order it differently disagree in the last bit
// of a few elements per ten million.
__device__ __forceinline__ float gelu_fwd_elem(float pre_bf16, float s) {
#pragma clang fp contract(off)
	return bf16_round(gelu_erf(pre_bf16)) * s;
}
__device__ __forceinline__ float gelu_bwd_elem(float acc, float s, float h) {
#pragma clang fp contract(off)
	const float a = bf16_round(acc) * s;
	return a * gelu_erf_grad(h);
}
// The same two epilogues for the reference's other layer / hidden activations (utils.get_activation_gain, utils.py:100-105: 'relu', 'tanh'; novic_epilogue_t.act =
// NOVIC_ACT_RELU / NOVIC_ACT_TANH, ABI 10).  Backward as torch's: relu'(x) = [x > 0] (threshold_backward on the result), tanh' = 1 - y^2 with y the bf16 result the
// forward stored (tanh_backward) -- recomputed here from the saved bf16 pre-activation.  ACT <= NOVIC_ACT_GELU: the erf GELU above.
template <int ACT>
__device__ __forceinline__ float act_fwd_elem(float pre_bf16, float s) {
	if (ACT == NOVIC_ACT_RELU) return fmaxf(pre_bf16, 0.f) * s;
	if (ACT == NOVIC_ACT_TANH) return bf16_round(tanhf(pre_bf16)) * s;
	if (ACT == NOVIC_ACT_IDENTITY) return pre_bf16 * s;  // (a block's output behind its dropout as a bf16 tensor of its own: ReZero)
	return gelu_fwd_elem(pre_bf16, s);
}
template <int ACT>
__device__ __forceinline__ float act_bwd_elem(float acc, float s, float h) {
#pragma clang fp contract(off)
	if (ACT == NOVIC_ACT_RELU) return h > 0.f ? bf16_round(acc) * s : 0.f;
	if (ACT == NOVIC_ACT_TANH) {
		const float y = bf16_round(tanhf(h));
		return unfused(bf16_round(acc) * s) * unfused(1.f - unfused(y * y));
	}
	return gelu_bwd_elem(acc, s, h);
}

// ACT / DROP >= 0 fix the activation (STORE_BF16; the GELU kinds: NOVIC_ACT_RELU / NOVIC_ACT_TANH, anything else = erf GELU) / whether dropout is on at compile time;
// -1 = read it from `ep` per call (STORE_BF16 only).  The kernels call this
// 16-32 times per thread in unrolled loops: with the choice made per call, every copy carries the erf GELU, the QuickGELU and the dropout-mask hash
// (7-25 k instructions per kernel, the epilogue then runs at the speed of the instruction cache) -- so they branch ONCE around the whole loop.
template <int EPI, int ACT = -1, int DROP = -1>
__device__ __forceinline__ void epilogue4(const novic_epilogue_t& ep, int m, int n, int N, float (&v)[4]) {
	const int nrem = N - n;                    // >= 1
	const bool full = nrem >= 4 && (ep.ldc & 3) == 0;
	const size_t o = (size_t)m * ep.ldc + n;
	float s[4] = {1.f, 1.f, 1.f, 1.f};
	if ((EPI == NOVIC_EPI_RESID_F32 || EPI == NOVIC_EPI_GELU_BF16 || EPI == NOVIC_EPI_GELU_BWD_BF16) && DROP != 0) {
		DropoutDesc d = {ep.drop_p, ep.seed_lo, ep.seed_hi, ep.drop_site};
		dropout_scale4(d, (uint64_t)m * N + n, s);
	}
	if ((EPI == NOVIC_EPI_STORE_BF16 || EPI == NOVIC_EPI_RESID_F32 || EPI == NOVIC_EPI_RESID_F16 || EPI == NOVIC_EPI_STORE_F32 || EPI == NOVIC_EPI_GELU_BF16) && ep.bias) {  // (fp32 store: the biased projections of the SigLIP towers; GELU_BF16: linear1 of a layer_bias decoder, ABI 10)
		float b[4];
		ld_f32x4((const float*)ep.bias + n, b, nrem >= 4, nrem);
#pragma unroll
		for (int r = 0; r < 4; ++r) v[r] += b[r];
	}
	if (EPI == NOVIC_EPI_STORE_BF16) {
		const int act = ACT >= 0 ? ACT : ep.act;
		if (act == NOVIC_ACT_GELU) {
#pragma unroll
			for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
		} else if (act == NOVIC_ACT_QUICKGELU) {
#pragma unroll
			for (int r = 0; r < 4; ++r) v[r] = quick_gelu(v[r]);
		} else if (act == NOVIC_ACT_GELU_TANH) {
#pragma unroll
			for (int r = 0; r < 4; ++r) v[r] = gelu_tanh(v[r]);
		}
		st_bf16x4((bf16*)ep.c + o, v, full, nrem);
	} else if (EPI == NOVIC_EPI_STORE_F32) {
		st_f32x4((float*)ep.c + o, v, full, nrem);
	} else if (EPI == NOVIC_EPI_ATOMIC_F32) {
		float* C = (float*)ep.c;
#pragma unroll
		for (int r = 0; r < 4; ++r) if (r < nrem) atomicAdd(C + o + r, v[r] * ep.alpha);
	} else if (EPI == NOVIC_EPI_RESID_F32) {
		// out = resid + dropout(bf16(acc [+ bias]))     (pre-LN residual add; GEMM output rounded to bf16 like autocast's linear)
		float rr[4];
		ld_f32x4((const float*)ep.resid + (size_t)m * ep.ldr + n, rr, nrem >= 4 && (ep.ldr & 3) == 0, nrem);
#pragma unroll
		for (int r = 0; r < 4; ++r) v[r] = rr[r] + bf16_round(v[r]) * s[r];
		st_f32x4((float*)ep.c + o, v, full, nrem);
		if (ep.c2) {  // the bf16 copy the GEMM behind the next LayerNorm multiplies (stays in L2: an ordinary store)
			if (full) *reinterpret_cast<bf16x4*>((bf16*)ep.c2 + o) = (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
			else for (int r = 0; r < nrem; ++r) ((bf16*)ep.c2)[o + r] = (bf16)v[r];
		}
	} else if (EPI == NOVIC_EPI_RESID_F16) {
		// out(f16) = f16(resid(f16) + f16(acc [+ bias])): the residual add of clip's half-precision model -- the linear's output rounded to half, the sum rounded to half
		const f16* R = (const f16*)ep.resid + (size_t)m * ep.ldr + n;
		f16* C = (f16*)ep.c + o;
		if (nrem >= 4 && (ep.ldr & 3) == 0 && full) {
			const f16x4 rr = *reinterpret_cast<const f16x4*>(R);
			f16x4 out;
#pragma unroll
			for (int r = 0; r < 4; ++r) out[r] = resid_f16_elem(rr[r], v[r]);
			__builtin_nontemporal_store(out, reinterpret_cast<f16x4*>(C));
		} else {
			for (int r = 0; r < 4 && r < nrem; ++r) C[r] = resid_f16_elem(R[r], v[r]);
		}
	} else if (EPI == NOVIC_EPI_GELU_BF16) {
		// c2 = bf16(acc [+ bias]) (pre-activation, saved for backward); c = dropout(bf16(act(bf16(acc [+ bias]))))
		constexpr int GA = (ACT == NOVIC_ACT_RELU || ACT == NOVIC_ACT_TANH || ACT == NOVIC_ACT_IDENTITY) ? ACT : NOVIC_ACT_GELU;
		float pre[4], act[4];
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			pre[r] = bf16_round(v[r]);
			act[r] = act_fwd_elem<GA>(pre[r], s[r]);
		}
		if (ep.c2) st_bf16x4((bf16*)ep.c2 + o, pre, full, nrem);
		st_bf16x4((bf16*)ep.c + o, act, full, nrem);
	} else if (EPI == NOVIC_EPI_GELU_BWD_BF16) {
		// c = bf16( bf16(acc) * dropmask * act'(hpre) )
		constexpr int GA = (ACT == NOVIC_ACT_RELU || ACT == NOVIC_ACT_TANH) ? ACT : NOVIC_ACT_GELU;
		float h[4];
		ld_bf16x4((const bf16*)ep.resid + (size_t)m * ep.ldr + n, h, nrem >= 4 && (ep.ldr & 3) == 0, nrem);
#pragma unroll
		for (int r = 0; r < 4; ++r) v[r] = act_bwd_elem<GA>(v[r], s[r], h[r]);
		st_bf16x4((bf16*)ep.c + o, v, full, nrem);
	}
}

// Calls f(integral_constant<ACT>, integral_constant<DROP>) with the epilogue's activation / dropout switch resolved: one uniform branch here
// instead of one per epilogue4 call.  VARIANTS: the caller also runs the GELU kinds with the reference's other activations (relu / tanh: the 128 x 128 kernel only --
// the kernels that decline such calls keep one copy of their store loop per dropout setting).
template <int V> struct epi_const { static constexpr int value = V; };
// (host) true for an epilogue only the 128 x 128 kernel implements: a GELU kind with another activation than the erf GELU, or GELU_BF16 with a bias
inline bool epilogue_is_act_variant(const novic_epilogue_t* ep) {
	if (ep->kind != NOVIC_EPI_GELU_BF16 && ep->kind != NOVIC_EPI_GELU_BWD_BF16) return false;
	return ep->act == NOVIC_ACT_RELU || ep->act == NOVIC_ACT_TANH || ep->act == NOVIC_ACT_IDENTITY || (ep->kind == NOVIC_EPI_GELU_BF16 && ep->bias);
}
template <int EPI, bool VARIANTS = false, class F>
__device__ __forceinline__ void epilogue_dispatch(const novic_epilogue_t& ep, F&& f) {
	if constexpr (EPI == NOVIC_EPI_STORE_BF16) {
		if (ep.act == NOVIC_ACT_GELU) f(epi_const<NOVIC_ACT_GELU>{}, epi_const<0>{});
		else if (ep.act == NOVIC_ACT_QUICKGELU) f(epi_const<NOVIC_ACT_QUICKGELU>{}, epi_const<0>{});
		else if (ep.act == NOVIC_ACT_GELU_TANH) f(epi_const<NOVIC_ACT_GELU_TANH>{}, epi_const<0>{});
		else f(epi_const<NOVIC_ACT_NONE>{}, epi_const<0>{});
	} else if constexpr (VARIANTS && (EPI == NOVIC_EPI_GELU_BF16 || EPI == NOVIC_EPI_GELU_BWD_BF16)) {
		if (ep.act == NOVIC_ACT_RELU) {
			if (ep.drop_p > 0.f) f(epi_const<NOVIC_ACT_RELU>{}, epi_const<1>{});
			else f(epi_const<NOVIC_ACT_RELU>{}, epi_const<0>{});
		} else if (ep.act == NOVIC_ACT_TANH) {
			if (ep.drop_p > 0.f) f(epi_const<NOVIC_ACT_TANH>{}, epi_const<1>{});
			else f(epi_const<NOVIC_ACT_TANH>{}, epi_const<0>{});
		} else if (EPI == NOVIC_EPI_GELU_BF16 && ep.act == NOVIC_ACT_IDENTITY) {
			if (ep.drop_p > 0.f) f(epi_const<NOVIC_ACT_IDENTITY>{}, epi_const<1>{});
			else f(epi_const<NOVIC_ACT_IDENTITY>{}, epi_const<0>{});
		} else if (ep.drop_p > 0.f) f(epi_const<0>{}, epi_const<1>{});
		else f(epi_const<0>{}, epi_const<0>{});
	} else if constexpr (EPI == NOVIC_EPI_RESID_F32 || EPI == NOVIC_EPI_GELU_BF16 || EPI == NOVIC_EPI_GELU_BWD_BF16) {
		if (ep.drop_p > 0.f) f(epi_const<0>{}, epi_const<1>{});
		else f(epi_const<0>{}, epi_const<0>{});
	} else {
		f(epi_const<0>{}, epi_const<0>{});
	}
}

}  // namespace
