/*
 * novic_hip.h -- C ABI of libnovic_hip.so: the MI355X (gfx950) kernels behind the NOVIC hot path.
 *
 * The reference (pallgeuer/novic) has no FFI of its own: its hot path is a sequence of torch ops issued by
 * embedding_decoder.py / embedding_noise.py / embedders.py.  Every entry point below replaces one such op
 * sequence (cited as reference file:line) and is what a reference-side binding (ctypes) calls.
 *
 * Conventions
 *   - plain pointers + sizes only; all pointers are DEVICE pointers owned by the caller (torch allocates);
 *   - no allocation, no device synchronisation, no global state except the last-error string;
 *   - every call enqueues on the hipStream_t it is given (graph-capture safe) and returns 0 or a negative errno-style
 *     code, with novic_last_error() describing the failure;
 *   - "bf16" buffers are 16-bit bfloat16, row-major, 16-byte aligned, leading dimensions multiples of 8 elements;
 *   - token ids are int64 or int32 (tok_bytes = 8 | 4), masks are 1-byte bool.
 */
#ifndef NOVIC_HIP_H
#define NOVIC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t; /* identical to hip_runtime_api.h's typedef, so C callers need no HIP headers */

#define NOVIC_ABI_VERSION 1

int novic_abi_version(void);
const char* novic_last_error(void);

/* ------------------------------------------------------------------------------------------------------------
 * GEMM with fused epilogue (replaces torch linear / matmul + the elementwise ops around it:
 * embedding_decoder.py:1276 prefix MLP, :714 -> nn.TransformerEncoderLayer in_proj/out_proj/linear1/linear2,
 * :725 logits_linear, and their autograd backward).
 * ------------------------------------------------------------------------------------------------------------ */
enum {
	NOVIC_EPI_STORE_BF16 = 0,     /* c(bf16) = act(acc + bias)                                            */
	NOVIC_EPI_STORE_F32 = 1,      /* c(f32)  = acc                                                        */
	NOVIC_EPI_ATOMIC_F32 = 2,     /* c(f32) += alpha * acc   (split-K weight gradients)                   */
	NOVIC_EPI_RESID_F32 = 3,      /* c(f32)  = resid(f32) + dropout(bf16(acc + bias))                     */
	NOVIC_EPI_GELU_BF16 = 4,      /* c2(bf16) = bf16(acc); c(bf16) = dropout(gelu(bf16(acc)))             */
	NOVIC_EPI_GELU_BWD_BF16 = 5,  /* c(bf16) = bf16(acc) * dropmask * gelu'(resid(bf16 pre-activation))   */
};
enum { NOVIC_ACT_NONE = 0, NOVIC_ACT_GELU = 1, NOVIC_ACT_QUICKGELU = 2 };

typedef struct novic_epilogue_t {
	int32_t kind;        /* NOVIC_EPI_*                                                     */
	int32_t act;         /* NOVIC_ACT_* (STORE_BF16 only)                                   */
	void* c;             /* primary output, leading dimension ldc                           */
	void* c2;            /* secondary output (GELU_BF16: pre-activation), may be NULL       */
	const void* resid;   /* RESID_F32: f32 residual; GELU_BWD_BF16: bf16 pre-activation     */
	const void* bias;    /* f32 [N] or NULL                                                 */
	int32_t ldc, ldr;    /* leading dimensions of c/c2 and of resid (elements)              */
	float alpha;         /* ATOMIC_F32 scale                                                */
	float drop_p;        /* dropout probability (0 = off)                                   */
	uint32_t seed_lo, seed_hi, drop_site;  /* Philox key + site id; mask index = m*N + n    */
	uint32_t _pad;
} novic_epilogue_t;

/* C[M][N] = A * B.  a_kstrided = 0: A stored [M][K] (lda >= K); 1: A stored [K][M] (lda >= M).
 *                   b_kstrided = 0: B stored [N][K] (ldb >= K); 1: B stored [K][N] (ldb >= N).
 * Supported: (0,0) forward, (0,1) input gradients, (1,1) weight gradients.  split_k > 1 requires ATOMIC_F32. */
int novic_gemm_bf16(const void* A, const void* B, int M, int N, int K, int lda, int ldb, int a_kstrided, int b_kstrided, int split_k,
                    const novic_epilogue_t* ep, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NOVIC_HIP_H */
