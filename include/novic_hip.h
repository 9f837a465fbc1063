/*
 * novic_hip.h -- C ABI of libnovic_hip.so: the MI355X (gfx950) kernels behind the NOVIC hot path.
 *
 * The reference (pallgeuer/novic) has no FFI of its own: its hot path is a sequence of torch ops issued by
 * embedding_decoder.py / embedding_noise.py / embedders.py.  Every entry point below replaces one such op
 * sequence (cited as reference file:line) and is what a reference-side binding (ctypes) calls.
 *
 * Conventions
 *   - plain pointers + sizes only; all pointers are DEVICE pointers owned by the caller (torch allocates);
 *   - no allocation, no device synchronisation; data-path state lives in the caller's buffers only.  What the library keeps for itself is listed under
 *     "Process-wide settings" below: the (thread-local) last-error string and a handful of tuning / diagnostic switches, each ONE relaxed std::atomic that a call
 *     reads once -- calls from several host threads on different streams are safe, a setter racing a call changes which kernel that call picks, never its result;
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

#define NOVIC_ABI_VERSION 12

/* Process-wide settings (everything the library keeps outside the caller's buffers; ABI 8 moved the one knob a PRODUCT path changed between launches -- the
 * workgroup budget of the persistent GEMM grids -- into the call: novic_epilogue_t.max_workgroups; ABI 9 dropped the switches of rejected experiments: SIX remain):
 *   - novic_last_error(): thread-local string of the calling thread's last failure;
 *   - novic_gemm_last_tile(): thread-local, the calling thread's last novic_gemm_bf16;
 *   - tuning switches, std::atomic<int>, set by tools / tests for A/B runs only (every setting computes the same numbers unless its comment says otherwise):
 *     novic_gemm_tile_policy, novic_gemm256_pipeline, novic_wgrad_policy, novic_skinny_wide_policy, novic_vit_attn_policy, novic_beam_step_policy;
 *   - novic_persistent_cus: std::atomic<int>, the DEFAULT of max_workgroups for callers that pass 0 (start value from $NOVIC_PERSISTENT_CUS);
 *   - diagnostics: novic_gemm_tile_counts (atomic counters), novic_gemm256_trace / novic_gemm128_trace (atomic pointers to caller buffers);
 *   - one-time hipFuncSetAttribute flags per kernel instantiation (atomic<bool>; the call is idempotent) and the cached occupancy of the decoder attention kernels (atomic<int>).
 * tests/test_cabi.py greps csrc/ for file-scope or function-local mutable statics outside exactly this list. */

int novic_abi_version(void);
const char* novic_last_error(void);

/* ------------------------------------------------------------------------------------------------------------
 * GEMM with fused epilogue (replaces torch linear / matmul + the elementwise ops around it:
 * embedding_decoder.py:1276 prefix MLP, :714 -> nn.TransformerEncoderLayer in_proj/out_proj/linear1/linear2,
 * :725 logits_linear, and their autograd backward).
 * ------------------------------------------------------------------------------------------------------------ */
enum {
	NOVIC_EPI_STORE_BF16 = 0,     /* c(bf16) = act(acc + bias)                                            */
	NOVIC_EPI_STORE_F32 = 1,      /* c(f32)  = acc + bias                                                 */
	NOVIC_EPI_ATOMIC_F32 = 2,     /* c(f32) += alpha * acc   (split-K weight gradients)                   */
	NOVIC_EPI_RESID_F32 = 3,      /* c(f32)  = resid(f32) + dropout(bf16(acc + bias))                     */
	NOVIC_EPI_GELU_BF16 = 4,      /* c2(bf16) = p = bf16(acc + bias); c(bf16) = dropout(act(p)): linear1 + activation of a layer / of the prefix MLP's hidden layer */
	NOVIC_EPI_GELU_BWD_BF16 = 5,  /* c(bf16) = bf16(acc) * dropmask * act'(resid(bf16 pre-activation))    */
	NOVIC_EPI_RESID_F16 = 6,      /* c(f16)  = f16(resid(f16) + f16(acc + bias)): the residual add of a tower whose stream is IEEE half (ABI 11) -- clip's fp16 model, which the   */
	                              /* reference runs for 'openai:' embedders (embedders.py:488-489: manual_amp_dtype = float16); no dropout; row-major operands, no row_limit       */
};
/* `act` of STORE_BF16: NONE / GELU / QUICKGELU / GELU_TANH.  Of the two GELU kinds (ABI 10): NONE or GELU = the erf GELU (the reference's default layer_activation), RELU, TANH =
 * its other choices (utils.get_activation_gain, utils.py:100-105); RELU / TANH and a bias in front of the activation run on the 128 x 128 kernel. */
enum { NOVIC_ACT_NONE = 0, NOVIC_ACT_GELU = 1, NOVIC_ACT_QUICKGELU = 2, NOVIC_ACT_GELU_TANH = 3, NOVIC_ACT_RELU = 4, NOVIC_ACT_TANH = 5,
       NOVIC_ACT_IDENTITY = 6 /* GELU_BF16 only: c = dropout(bf16(acc + bias)) -- a block's output as a bf16 tensor of its own (ReZero scales it before the residual add) */ };

typedef struct novic_epilogue_t {
	uint32_t struct_bytes;  /* = sizeof(novic_epilogue_t) of the header the CALLER was built against: novic_gemm_bf16 refuses any other value, so a    */
	                        /* binding written for an older, shorter layout fails with -EINVAL instead of having fields read past its struct             */
	int32_t kind;        /* NOVIC_EPI_*                                                     */
	int32_t act;         /* NOVIC_ACT_* (STORE_BF16; the GELU kinds: which activation)      */
	uint32_t max_workgroups;  /* 0: the library default (novic_persistent_cus), else the workgroups THIS call's persistent 256-wide GEMM grid may have (rounded down to */
	                          /* a multiple of 8 in 8..256): the other CUs stay free for kernels on other streams; per call, so concurrent callers cannot disturb each other */
	void* c;             /* primary output, leading dimension ldc                           */
	void* c2;            /* secondary output (GELU_BF16: pre-activation), may be NULL       */
	const void* resid;   /* RESID_F32 / RESID_F16: f32 / f16 residual (may be c itself, ldr == ldc: in place); GELU_BWD_BF16: bf16 pre-activation */
	const void* bias;    /* f32 [N] or NULL                                                 */
	int32_t ldc, ldr;    /* leading dimensions of c/c2 and of resid (elements)              */
	float alpha;         /* ATOMIC_F32 scale                                                */
	float drop_p;        /* dropout probability (0 = off)                                   */
	uint32_t seed_lo, seed_hi, drop_site;  /* dropout key + site id; mask index = m*N + n   */
	uint32_t reserved0;    /* 0 (ABI 8: store_policy of the bf16 tiles -- the write-back variant measured no gain and went in ABI 9; the tiles' stores are non-temporal) */
	const int32_t* row_limit;  /* NULL, or a DEVICE int: only the first *row_limit token rows take part -- M is clamped to it (row-major A), or K
	                            * for the weight-gradient form (both operands K-strided: the K ranges of the splits are dealt out over the
	                            * clamped K).  Lets a caller compact the non-padded rows to the front without reading the count back. */
	void* splitk_ws;           /* NULL, or caller-owned DEVICE scratch (16-byte aligned, splitk_ws_bytes long; 64 MiB covers every case) that lets the  */
	uint64_t splitk_ws_bytes;  /* 256-wide-tile kernel cut the few output tiles behind the last whole round of 256 along K (ViT towers: 257 row tiles).  */
	                           /* Deterministic, but a different summation order than the unsplit kernels: leave NULL where bit-identity matters.       */
	                           /* The scratch is in use until the call's kernels have run: calls sharing it must be ordered on one stream.              */
} novic_epilogue_t;

/* C[M][N] = A * B.  a_kstrided = 0: A stored [M][K] (lda >= K); 1: A stored [K][M] (lda >= M).
 *                   b_kstrided = 0: B stored [N][K] (ldb >= K); 1: B stored [K][N] (ldb >= N).
 * Supported: (0,0) forward, (0,1) input gradients, (1,1) weight gradients.  split_k > 1 requires ATOMIC_F32. */
int novic_gemm_bf16(const void* A, const void* B, int M, int N, int K, int lda, int ldb, int a_kstrided, int b_kstrided, int split_k,
                    const novic_epilogue_t* ep, hipStream_t stream);
/* Weight gradient dW[M][N] (fp32, ldw) += alpha * dY^T X over the K token rows, dY [K][ldy >= M] and X [K][ldx >= N] bf16 row-major as the passes left them
 * (autograd's grad_output^T @ input of every nn.Linear: embedding_decoder.py:309-327, :725, :1276).  256 x 256 output tiles, the token range cut into parts
 * so that tiles x parts workgroups fill the chip in one round (splits_hint > 0 overrides the count); the parts' fp32 partial sums go through `ws`
 * (caller-owned DEVICE scratch, tiles x parts x 256 KiB <= ws_bytes; 64 MiB covers every case) and are added in a fixed order: deterministic, no atomics.
 * row_limit: NULL, or a DEVICE int clamping K (packed rows).  M, N, ldy, ldx multiples of 8; at most 256 output tiles.  Calls sharing `ws` must be
 * ordered on one stream.  max_workgroups (ABI 9): 0 = the whole chip, else the workgroups this launch may have (as novic_epilogue_t.max_workgroups: the part count is
 * chosen to fit it -- a different part count is a different fp32 summation order, deterministic for a given value). */
int novic_wgrad_bf16(const void* dY, const void* X, int M, int N, int K, int ldy, int ldx, float* dW, int ldw, float alpha, const int32_t* row_limit, void* ws,
                     uint64_t ws_bytes, int splits_hint, int max_workgroups, hipStream_t stream);
/* Two weight gradients over the SAME token rows in one launch pair (a layer's in-projection and out-projection gradients): the tiles of both problems share the
 * chip's one round of workgroups, so each is cut into fewer parts -- half the partial-sum traffic of two separate calls.  Either both outputs wider than 128 in both
 * dimensions (256 x 256 tiles) or both at most 128 wide in one (128 x 256 tiles: the feed-forward pair [128 x 512] / [512 x 128], the latter computed as its transpose);
 * otherwise as novic_wgrad_bf16 (fixed-order partial sums, run-to-run deterministic; a different part count, hence a different fp32 summation order). */
int novic_wgrad2_bf16(const void* dY1, const void* X1, int M1, int N1, int ldy1, int ldx1, float* dW1, int ldw1, const void* dY2, const void* X2, int M2, int N2, int ldy2,
                      int ldx2, float* dW2, int ldw2, int K, float alpha, const int32_t* row_limit, void* ws, uint64_t ws_bytes, int max_workgroups, hipStream_t stream);
/* ABI 11: up to FOUR weight gradients over the same token rows in one launch pair -- the in-projection + out-projection gradients of TWO layers (32 tiles x 8 parts: half the
 * partial-sum traffic of two novic_wgrad2_bf16 calls, one reduction instead of two), or the narrow feed-forward pairs of two layers (8 tiles of 128 x 256 x 32 parts).  All
 * outputs wide (256 x 256 tiles) or all narrow (at most 128 in one dimension); otherwise as novic_wgrad2_bf16.  dW_i[M_i][ldw_i] (fp32) += alpha * dY_i^T X_i. */
typedef struct novic_wgrad_problem { const void* dY; const void* X; float* dW; int32_t M, N, ldy, ldx, ldw, reserved0; } novic_wgrad_problem_t;
int novic_wgradn_bf16(const novic_wgrad_problem_t* problems, int n, int K, float alpha, const int32_t* row_limit, void* ws, uint64_t ws_bytes, int max_workgroups,
                      hipStream_t stream);
/* Kernel selection for novic_wgrad_bf16 / novic_wgrad2_bf16 (A/B measurements and tests: both kernels produce bit-identical partial sums).  1 (default): the 8-phase
 * schedule (wgrad256p_kernel); 0: one barrier per K-tile (wgrad256_kernel).  Any other value only queries.  Returns the previous policy. */
int novic_wgrad_policy(int policy);
/* Kernel selection knob for novic_gemm_bf16 (tuning / A-B measurements only: both kernels give bit-identical results).  policy 1 (default): large
 * K-contiguous x K-contiguous problems run on the 256x256-tile LDS-DMA kernel; policy 0: always the 128x128-tile kernel.  Any other value only
 * queries.  Returns the previous policy. */
int novic_gemm_tile_policy(int policy);
/* Tile edge (128 or 256) of the kernel the CALLING THREAD's most recent novic_gemm_bf16 call launched (0 before its first call): for tests / profiling. */
int novic_gemm_last_tile(void);
/* K-loop schedule of the 256 x 256 tile (A/B measurements and tests: bit-identical results either way).  1 (default): the 8-phase schedule (gemm256p_kernel: staggered
 * wave groups, half-tile LDS-DMA six half-tiles ahead, counted vmcnt); 0: one barrier per K-tile (gemm256_kernel).  Any other value only queries.  Returns the previous schedule.
 * (ABI 9 removed the values 2-11: knobs of round-4 experiments whose losing side no call took -- K = 1024 tails off, write-back stores, 128- / 192-row tiles.) */
int novic_gemm256_pipeline(int on);
/* Process-wide DEFAULT of novic_epilogue_t.max_workgroups: how many workgroups the persistent 256-wide GEMM grids may have when a call passes 0 (a multiple of 8 in
 * 8..256; default 256 = one per CU, or $NOVIC_PERSISTENT_CUS; a negative value only queries; returns the previous value; atomic).  Below 256 the remaining CUs stay free for kernels of other streams -- a decode step beside an image tower, a collective beside the backward pass --
 * which otherwise wait for a whole persistent grid to end; K-split tails are planned for rounds of this many tiles, so sums may differ in the last bits from 256. */
int novic_persistent_cus(int n);
/* What novic_gemm_bf16 would choose for a K-contiguous [M x N x K] problem with this epilogue once it reaches the 256-wide kernels -- the decision alone, no launch, no
 * HIP call (tests pin the tile policy with it): out4 = {tile width 256 | 192 (256 rows) | 0 = left to the 128 x 128 kernel, workgroups, K-split of the tail tiles: parts | -1 = planned
 * on the device from the row count | 0 = none, tail tiles}.  Only the null-ness / alignment of the epilogue's pointers is looked at.  lda = ldb = K is assumed. */
int novic_gemm256_plan(int M, int N, int K, const novic_epilogue_t* ep, int* out4);
/* Launch counters of novic_gemm_bf16 since the last reset, for tests that must prove a model-level check ran through the large tiles: out7 = {128x128 kernel,
 * streaming 128-column kernel, 256-wide tile kernels (256x256 and 128x256), 256x192 tile, launches with a host-planned K-split tail, launches with a device-planned one,
 * of the 256-wide launches those on 128x256 or 192x256 tiles}.  reset != 0 zeroes them after the copy; out7 may be null.  Diagnostic only (no reference counterpart). */
int novic_gemm_tile_counts(unsigned long long* out7, int reset);
/* diagnostic: per-workgroup timeline of the LDS-DMA GEMM kernel into buf[256][32][4] (100 MHz wall-clock stamps: tile start, first K-tile done,
 * K loop done, stores issued); NULL = off (tools/gemm_timeline.py) */
int novic_gemm256_trace(unsigned long long* buf);
/* the same for the 128^2 kernel: buf[workgroup < 16384][4] = start, first K-tile in LDS, K loop done, epilogue done */
int novic_gemm128_trace(unsigned long long* buf);

/* ------------------------------------------------------------------------------------------------------------
 * Row kernels (one wave per row, statistics by wave shuffles).
 * ------------------------------------------------------------------------------------------------------------ */
/* y(bf16)[r] = x[r] / max(||x[r]||, 1e-12): the F.normalize prologue of the prefix MLP (embedding_decoder.py:1276)
 * and of inference_image (embedders.py:764). */
int novic_rownorm_bf16(const float* x, void* y_bf16, int rows, int E, int ldy, hipStream_t stream);

/* LayerNorm forward, f32 in -> bf16 (and/or f32) out, optional beta.  Output row r reads input row
 * (r / seq_out) * seq_in + seq_off + r % seq_out, which lets the final norm run on the predicted positions only
 * (embedding_decoder.py:714-723 with nn.LayerNorm eps; plain LayerNorm is seq_in = seq_out = 1, seq_off = 0). */
int novic_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* y_f32, int rows_out, int E, int seq_in, int seq_out,
                        int seq_off, float eps, hipStream_t stream);
/* ABI 11: the same over rows stored as IEEE half (x_f16 [rows][E]) -> bf16: the LayerNorm of a tower whose residual stream is half precision -- clip's fp16 model, which
 * the reference runs for 'openai:' embedders (embedders.py:488-489); statistics and affine map in fp32, as clip's LayerNorm subclass computes them (cast up, normalise). */
int novic_layernorm_fwd_f16(const void* x_f16, const float* gamma, const float* beta, void* y_bf16, int rows_out, int E, int seq_in, int seq_out, int seq_off, float eps,
                            hipStream_t stream);
/* gathered form: y[j] = LayerNorm(x[src_rows[j]]) for j < *row_count (a DEVICE int, clamped to rows_max): the final norm over the non-padded output
 * positions only (embedding_decoder.py:690, :721), which a compaction kernel (novic_compact_rows) lists first. */
int novic_layernorm_fwd_rows(const float* x, const float* gamma, const float* beta, void* y_bf16, const int* src_rows, const int* row_count, int rows_max, int E,
                             float eps, hipStream_t stream);

/* LayerNorm backward (no bias) over every input row m: dx_out[m] = (dx_in ? dx_in[m] : 0) + LN'(dy[r(m)]) (0 for unselected rows),
 * g_out(bf16)[m] = dx_out[m] * dropout mask of `drop_site` (index m*E+e) -- the operand of the next backward GEMM --
 * and dgamma += sum_rows dy * xhat (fp32 atomics of per-block partials). */
int novic_layernorm_bwd(const void* dy_bf16, const float* x, const float* gamma, const float* dx_in, float* dx_out, void* g_out_bf16, float* dgamma, int rows_in,
                        int E, int seq_in, int seq_out, int seq_off, float eps, float drop_p, uint64_t seed, uint32_t drop_site,
                        const int* dy_row, const int* row_limit, hipStream_t stream);
/* The normalised hidden layer of the prefix MLP (ABI 10; reference EmbeddingVectorMLP with mlp_hidden_layer != 'none' and mlp_hidden_norm, embedding_decoder.py:1247-1253;
 * without the norm the hidden layer is novic_gemm_bf16's GELU_BF16 / GELU_BWD_BF16 epilogue pair): y(bf16)[r] = act(LayerNorm(h0(bf16)[r]; gamma, beta)) -- the
 * LayerNorm in fp32 on linear1's bf16 output, as under autocast -- and its backward dh0(bf16) = LayerNorm'(dy * act'(z)), dgamma += sum_rows dz * xhat, dbeta += sum_rows dz
 * (fp32 atomics of per-block partials; dbeta / beta may be NULL).  act: NOVIC_ACT_NONE | GELU (erf) | RELU | TANH.  H <= 2048, a multiple of 4 like the leading dimensions. */
int novic_hidden_norm_act_fwd(const void* h0_bf16, const float* gamma, const float* beta, void* y_bf16, int rows, int H, int ldh, int ldy, int act, float eps, hipStream_t stream);
int novic_hidden_norm_act_bwd(const void* dy_bf16, const void* h0_bf16, const float* gamma, const float* beta, void* dh0_bf16, float* dgamma, float* dbeta, int rows, int H,
                              int ldy, int ldh, int ldd, int act, float eps, hipStream_t stream);
/* Post-LN layers (ABI 10; reference layer_norm_first = False: x = norm(x + block(x)), nn.TransformerEncoderLayer built in embedding_decoder.py:309-327): the backward of a
 * norm whose upstream gradient is the SUM of an fp32 part (dy_f32: what flows into the stream behind the norm through the next block's residual path) and a bf16 part (dy_bf16:
 * the input gradient of that block's first linear); either may be NULL.  dx_out = LayerNorm'(dy; x, gamma), g_out(bf16, optional) = dx_out * dropout mask of `drop_site`
 * (index m*E+e), dgamma += sum_rows dy * xhat, dbeta += sum_rows dy (fp32 atomics of per-block partials; either may be NULL).  row_limit as elsewhere. */
int novic_layernorm_bwd_sum(const void* dy_bf16, const float* dy_f32, const float* x, const float* gamma, float* dx_out, void* g_out_bf16, float* dgamma, float* dbeta, int rows,
                            int E, float eps, float drop_p, uint64_t seed, uint32_t drop_site, const int* row_limit, hipStream_t stream);
/* dst(f32)[i] += src(bf16)[i], n a multiple of 4: the two parts of a post-LN layer-0 input gradient in front of novic_embed_bwd. */
int novic_add_bf16(float* dst, const void* src_bf16, uint64_t n, hipStream_t stream);
/* ReZero (ABI 10; reference TransformerEncoderLayer(rezero = 'perskip' | 'perlayer'), embedding_decoder.py:1086-1117: `x *= scale` on a block's bf16 output behind its dropout):
 * out = resid + bf16(scale * branch), `scale` a DEVICE scalar (the parameter).  Backward: g = bf16(dx) is the gradient of the scaled branch; dscale += sum g * branch (fp32
 * atomics of per-block partials); g_out = bf16(bf16(g * scale) * dropout mask of `drop_site`), the gradient of the block's last linear output. */
int novic_rezero_fwd(const float* resid, const void* branch_bf16, const float* scale, float* out, int rows, int E, const int* row_limit, hipStream_t stream);
int novic_rezero_bwd(const float* dx, const void* branch_bf16, const float* scale, float* dscale, void* g_out_bf16, int rows, int E, float drop_p, uint64_t seed, uint32_t drop_site,
                     const int* row_limit, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * The feed-forward half of a decoder layer in one launch (nn.TransformerEncoderLayer with norm_first, embedding_decoder.py:309-327:
 * x_out = xmid + dropout(linear2(dropout(gelu(linear1(LayerNorm(xmid; gamma2)))))), then the NEXT layer's LayerNorm(x_out; gamma_next)).
 * Outputs the backward pass needs are optional (NULL = not stored): ln2 (bf16 [M][E]), hpre / hact (bf16 [M][Kf]); ln_next (bf16 [M][E]) goes with
 * gamma_next.  Stage for stage the arithmetic of novic_layernorm_fwd + novic_gemm_bf16(GELU_BF16) + novic_gemm_bf16(RESID_F32) + novic_layernorm_fwd (the GEMM
 * stages bit-identical given the same inputs, the LayerNorm stages to one bf16 ulp on fp32 rounding ties).
 * Built for E = 512, Kf = 128 (novic_ffn_fused_supported); dropout masks: site_gelu keyed by row * Kf + column, site_out by row * E + column.
 * ------------------------------------------------------------------------------------------------------------ */
int novic_ffn_fused_supported(int E, int Kf);
int novic_ffn_fwd(const float* xmid, const float* gamma2, const void* w1_bf16, const void* w2_bf16, const float* gamma_next, float* x_out, void* ln2_bf16,
                  void* hpre_bf16, void* hact_bf16, void* ln_next_bf16, int M, int E, int Kf, float eps, float drop_p, uint64_t seed, uint32_t site_gelu,
                  uint32_t site_out, const int32_t* row_limit, hipStream_t stream);

/* Backward of the same block up to the weight gradients, one launch: dh = bf16(bf16(gb W2) * dropmask(site_gelu) * gelu'(hpre)) (stored: operand of the
 * linear1 weight gradient), dln = bf16(dh W1) (never stored), dx_out = dx_in + LayerNorm'(dln; xmid, gamma2), g_out = bf16(dx_out * dropmask(site_g)),
 * dgamma2 += sum_rows dln * xhat.  w2t / w1t: the TRANSPOSED bf16 weights, linear2.weight^T [Kf][E] and linear1.weight^T [E][Kf].  dx_out may be dx_in;
 * g_out must not alias gb.  The arithmetic of novic_gemm_bf16(GELU_BWD_BF16) + novic_gemm_bf16(STORE_BF16) + novic_layernorm_bwd. */
int novic_ffn_bwd(const void* gb_bf16, const void* hpre_bf16, const float* xmid, const float* dx_in, const float* gamma2, const void* w2t_bf16, const void* w1t_bf16,
                  void* dh_bf16, float* dx_out, void* g_out_bf16, float* dgamma2, int M, int E, int Kf, float eps, float drop_p, uint64_t seed, uint32_t site_gelu,
                  uint32_t site_g, const int32_t* row_limit, hipStream_t stream);

/* The same launch with the LayerNorm backward of the layer ABOVE as its prologue (norm1 of layer l+1, whose input pre_x is this block's output: autograd's
 * backward of `x = x + sa_block(norm1(x))` meeting `x = x + ff_block(norm2(x))`, embedding_decoder.py:309-327):
 *     dx = dx_in + LayerNorm'(pre_dln; pre_x, pre_gamma);  gb_out = bf16(dx * dropmask(site_pre));  pre_dgamma += sum_rows pre_dln * xhat
 * and then novic_ffn_bwd on (gb_out, dx) -- dx stays on chip between the two, gb_out is stored once for the linear2 weight gradient.
 * pre_row_map (optional, device int32 [M]): row m's upstream gradient is pre_dln row pre_row_map[m], none if < 0 (the FINAL norm over the compacted
 * output rows, novic_layernorm_bwd's dy_row); dx_in may be null (= zero: nothing flows into the final norm's input from elsewhere).
 * The arithmetic of novic_layernorm_bwd followed by novic_ffn_bwd; pre_dln, gb_out and g_out must be three different buffers. */
int novic_ffn_bwd_ln(const void* pre_dln_bf16, const int32_t* pre_row_map, const float* pre_x, const float* pre_gamma, float* pre_dgamma, void* gb_out_bf16, uint32_t site_pre,
                     const void* hpre_bf16, const float* xmid, const float* dx_in, const float* gamma2, const void* w2t_bf16, const void* w1t_bf16, void* dh_bf16,
                     float* dx_out, void* g_out_bf16, float* dgamma2, int M, int E, int Kf, float eps, float drop_p, uint64_t seed, uint32_t site_gelu, uint32_t site_g,
                     const int32_t* row_limit, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Layer-0 input: prefix tokens + tied token embedding + learned positions + dropout
 * (embedding_decoder.py:665-675, :692-693, :1297; utils.py:65-68) and its backward.
 * Sequence a belongs to sample a / mrep (or a % B when multi_first).
 * ------------------------------------------------------------------------------------------------------------ */
int novic_embed_fwd(const void* prefix_bf16, const void* tokens, int tok_bytes, int tok_ld, const float* wtok, const float* pos, float* x0, int A, int S, int P, int E,
                    int V, int B, int mrep, int multi_first, float drop_p, uint64_t seed, uint32_t drop_site,
                    const int* seq_start, const int* seq_len, hipStream_t stream);
int novic_embed_bwd(const float* dx0, const void* tokens, int tok_bytes, int tok_ld, float* dwtok, float* dpos, void* dprefix_bf16, int A, int S, int P, int E, int V,
                    int B, int mrep, int multi_first, float drop_p, uint64_t seed, uint32_t drop_site,
                    const int* seq_start, const int* seq_len, hipStream_t stream);

/* ABI 11: the two ends of the layer stack fused with layer 0's norm1 (bias-free LayerNorm: the released recipe; embedding_decoder.py:309-327 builds norm_first layers).
 * novic_embed_fwd_ln = novic_embed_fwd + novic_layernorm_fwd of the rows it has just assembled (x0 AND ln_out = bf16(LayerNorm(x0; ln_gamma)), bit-identical to the two
 * launches).  novic_ln_embed_bwd = novic_layernorm_bwd (dx0 = dx_in + LN'(dy; x0, ln_gamma), dgamma += ...) in front of novic_embed_bwd: dx0 is scattered from registers and
 * never written (the arithmetic of the two kernels; atomics as novic_embed_bwd).  E <= 2048 / 1024. */
int novic_embed_fwd_ln(const void* prefix_bf16, const void* tokens, int tok_bytes, int tok_ld, const float* wtok, const float* pos, float* x0, int A, int S, int P, int E,
                       int V, int B, int mrep, int multi_first, float drop_p, uint64_t seed, uint32_t drop_site, const int* seq_start, const int* seq_len,
                       const float* ln_gamma, void* ln_out_bf16, float eps, hipStream_t stream);
int novic_ln_embed_bwd(const void* dy_bf16, const float* x0, const float* ln_gamma, const float* dx_in, float* dgamma, const void* tokens, int tok_bytes, int tok_ld,
                       float* dwtok, float* dpos, void* dprefix_bf16, int A, int S, int P, int E, int V, int B, int mrep, int multi_first, float drop_p, uint64_t seed,
                       uint32_t drop_site, const int* seq_start, const int* seq_len, float eps, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Decoder self-attention, S <= 32, head_dim in {16, 32, 64}; mask from integers:
 * allowed(i,j) = (j <= i or (!strict and i < P and j < P)) and not (key_pad[a][j] and j > 0)
 * (embedding_decoder.py:651-654, :696-712 + torch's _sa_block / SDPA).  qkv is [A*S][3*H*D] bf16, o is [A*S][H*D] bf16.
 * ------------------------------------------------------------------------------------------------------------ */
int novic_dec_attn_fwd(const void* qkv_bf16, const uint8_t* key_pad, void* o_bf16, int A, int S, int H, int D, int P, int strictly_causal, float drop_p, uint64_t seed,
                       uint32_t drop_site,
                       const int* seq_start, const int* seq_len, hipStream_t stream);
int novic_dec_attn_bwd(const void* qkv_bf16, const uint8_t* key_pad, const void* do_bf16, void* dqkv_bf16, int A, int S, int H, int D, int P, int strictly_causal,
                       float drop_p, uint64_t seed, uint32_t drop_site,
                       const int* seq_start, const int* seq_len, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Padding, loss, accuracy (embedding_decoder.py:681-685, :696-712, :729-761).
 * ------------------------------------------------------------------------------------------------------------ */
/* key_pad[A][P+C-1] (column 0 never set) and out_pad[A][C] from target padding [A][C] with row stride tpad_ld (may be NULL) and weights (may be NULL). */
int novic_build_padding(const uint8_t* target_padding, int tpad_ld, const float* weight, uint8_t* key_pad, uint8_t* out_pad, int A, int C, int P,
                        int num_end_loss, hipStream_t stream);
/* Token cross entropy over bf16 logits [A*T][ldl] (row a*T+t is scored against target[a*tok_ld + col0+t], padding out_pad[a*C + col0+t]); writes the unweighted per-token loss,
 * the arg-max (lowest index on ties, ids < argmax_from excluded) and the correct flag; with write_grad != 0 the logits are
 * overwritten by d(loss)/d(logits) = (softmax - onehot) * weight[a] * grad_scale * (grad_scale_dev ? *grad_scale_dev : 1) / basis[a / group_rows]. */
int novic_cross_entropy(void* logits_bf16, int ldl, int V, int A, int T, int C, int col0, const void* target, int tok_bytes, int tok_ld, const uint8_t* out_pad,
                        const float* weight, const float* basis, int group_rows, float grad_scale, const float* grad_scale_dev, float label_smoothing, int write_grad,
                        float* row_loss, int* row_argmax, uint8_t* row_correct, int argmax_from,
                        const int* row_map, const int* row_limit, hipStream_t stream);
/* Compacted form of the loss block: row_map / row_limit of novic_cross_entropy name the logits rows that exist (logits row j = token row row_map[j], j <
 * *row_limit); novic_compact_rows produces them from the output padding and the sample weights (one small launch): rows[] for the cross-entropy, src_rows[]
 * for novic_layernorm_fwd_rows, dst_of[] for novic_layernorm_bwd's dy_row, count[0] as the row_limit of the three logits GEMMs (count must hold
 * 1 + ceil(A * T / 1024) ints: the rest is scratch of the two-pass compaction).  The reference computes and
 * then masks the padded positions (embedding_decoder.py:729-745); their loss and gradient are exactly zero either way. */
int novic_compact_rows(const uint8_t* out_pad, const float* weight, int A, int T, int C, int col0, int S, int* rows, int* src_rows, int* dst_of, int* count,
                       float* row_loss, int* row_argmax, uint8_t* row_correct, const int* seq_start, hipStream_t stream);

/* Packed rows ("variable-length batch"): a sequence keeps only its positions in front of its padding suffix; sequence a then lives in rows
 * seq_start[a] .. seq_start[a] + seq_len[a] - 1 of every [rows][*] activation of the decoder instead of a*S .. a*S + S - 1.  novic_seq_layout derives
 * start / len from the key padding (embedding_decoder.py:696-712) and leaves the row count in total[0] (total must hold 1 + ceil(A / 1024) ints),
 * all on the device: the embed, attention and compaction entry points take seq_start / seq_len, every row-wise kernel and GEMM takes total as its
 * row_limit.  NULL everywhere = the dense [A][S] layout.  Padded positions produce no loss and no gradient in the reference either. */
int novic_seq_layout(const uint8_t* key_pad, int A, int S, int* seq_start, int* seq_len, int* total, hipStream_t stream);
/* Per micro-batch group (group_rows sequences): basis, weighted loss sum, #correct, #unpadded tokens (deterministic block reductions). */
int novic_loss_group_reduce(const float* row_loss, const uint8_t* row_correct, const uint8_t* out_pad, const float* weight, float* basis, float* loss,
                            float* correct, float* tokens, int A, int T, int C, int col0, int group_rows, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fused embedding noise, in place on B x F fp32 unit rows (embedding_noise.py:72-172, train.py:1263-1265).
 * Angles are radians.  inj_* (may be NULL) replace the Philox draws: inj_z1/inj_z2 [B][F] N(0,1), inj_row [B] (N(0,1) for
 * GAUSS_VEC / GAUSS_ANGLE, U[0,1) for the uniform angle), inj_mix [B] U[0,1).
 * ------------------------------------------------------------------------------------------------------------ */
enum {
	NOVIC_NOISE_NONE = 0,
	NOVIC_NOISE_GAUSS_ELEM = 1,
	NOVIC_NOISE_GAUSS_VEC = 2,
	NOVIC_NOISE_GAUSS_ANGLE = 3,
	NOVIC_NOISE_UNIFORM_ANGLE = 4,
	NOVIC_NOISE_GAUSS_ELEM_UNIFORM_ANGLE = 5,
};
int novic_noise_fused(float* embed, int B, int F, int mode, float vec_norm, float angle_min_rad, float angle_max_rad, float angle_std_rad, float mix_ratio,
                      uint64_t seed, uint32_t offset, const float* inj_z1, const float* inj_z2, const float* inj_row, const float* inj_mix, const float* mean_shift,
                      hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Optimizer over the flat parameter buffer (train.py:1103-1119, :1280-1286).
 * ------------------------------------------------------------------------------------------------------------ */
/* out_norm[0] = ||grads||_2 (two deterministic passes; partial_ws holds >= ws_len doubles). */
int novic_grad_norm(const float* grads, uint64_t n, double* partial_ws, int ws_len, float* out_norm, hipStream_t stream);
/* clip (coef = min(1, max_norm / (grad_norm + 1e-6))) + decoupled AdamW on params[0..n), weight decay on [0..n_decay) only,
 * refreshing the bf16 shadow the GEMMs read.  `hyper` is a HOST struct read during the call and passed to the kernel BY VALUE: a later step's
 * values can never reach an earlier step's launch, however far the host runs ahead of the device (no staging buffer to race on). */
typedef struct novic_adamw_hyper_t {
	float lr, beta1, beta2, eps, weight_decay;
	float bias_corr1, bias_corr2;  /* 1 - beta1^t, 1 - beta2^t of this step t (1-based) */
	float max_norm;                /* <= 0: no clipping */
} novic_adamw_hyper_t;
int novic_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, uint64_t n, uint64_t n_decay,
                     const novic_adamw_hyper_t* hyper, const float* grad_norm, hipStream_t stream);
int novic_cast_bf16(const float* x, void* y_bf16, uint64_t n, hipStream_t stream);
/* out[c] (fp32) += alpha * sum over the rows r < min(rows, *row_limit) of x[r][c], x bf16 [rows][ld >= cols]: the bias gradient of a linear layer (grad_output.sum(0); the
 * decoder's optional logits bias, embedding_decoder.py:239).  One read of x; deterministic (row slabs summed in a fixed order through `ws`: caller-owned DEVICE scratch of
 * at least 256 x cols fp32).  ld a multiple of 8, x 16-byte aligned; row_limit NULL or a device int. */
int novic_colsum_bf16(const void* x_bf16, int rows, int cols, int ld, const int32_t* row_limit, float* out, float alpha, float* ws, uint64_t ws_bytes, hipStream_t stream);
/* n transposed bf16 copies in one launch: desc (HOST array) [n][5] = source offset, destination offset (elements from the bases), rows, columns of the
 * source, leading dimension of the destination (>= rows); destination i = [columns][ld].  The W^T weight shadows of the input-gradient GEMMs (torch autograd's `grad @ weight`, e.g.
 * embedding_decoder.py:1291-1296 linears in backward).  Bases 16-byte aligned; 16-byte accesses where offsets / rows / columns are multiples of 8. */
int novic_transpose_bf16_batched(const void* src_base, void* dst_base, const long long* desc, int n, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Decode steps (no host synchronisation; active[step-1] counts sequences/beams still unfinished after the step).
 * ------------------------------------------------------------------------------------------------------------ */
/* Optional extra output of a greedy / beam step: the NEXT decode step's inputs, so that novic_decode_embed (and, for beams, novic_kv_origin_update) need
 * no launches of their own -- x_next[row] = wtok[token chosen for that row] + pos_row (fp32 [rows][E]); with origin_out, the K/V origin rows of the sample's new
 * beams as novic_kv_origin_update(src, origin_in, origin_out, ..., npos) writes them.  struct_bytes = sizeof(novic_next_embed_t), checked. */
typedef struct novic_next_embed_t {
	uint32_t struct_bytes;
	int32_t E;               /* hidden size (a multiple of 4) */
	const float* wtok;       /* tied token embedding [V][E] fp32 */
	const float* pos_row;    /* the next position's row of the learned positional embedding [E] */
	float* x_next;           /* [B (x H)][E] */
	const int32_t* origin_in; /* beams only, optional: [B*H][G] */
	int32_t* origin_out;
	int32_t npos;
	int32_t _pad0;
} novic_next_embed_t;

/* Greedy (embedding_decoder.py:792-820, :826-845): arg-max of the step's logits (END excluded at step 1), writes ids[:, step-1] and
 * pad[:, step-1] (= finished before this step), accumulates score / nll / count over unpadded positions, updates alive (1/0). */
int novic_greedy_step(const void* logits_bf16, int ldl, int V, int B, int G, int step, void* ids, int tok_bytes, uint8_t* pad, float* alive, float* score, float* nll,
                      float* count, int* active, float* step_logits, float temperature, float label_smoothing, hipStream_t stream);
int novic_greedy_step_next(const void* logits_bf16, int ldl, int V, int B, int G, int step, void* ids, int tok_bytes, uint8_t* pad, float* alive, float* score, float* nll,
                           float* count, int* active, float* step_logits, float temperature, float label_smoothing, const novic_next_embed_t* next, hipStream_t stream);
/* ids[pad] = 0 and score *= max(count, 1)^-alpha (embedding_decoder.py:824, :835-836). */
int novic_greedy_finalize(void* ids, int tok_bytes, const uint8_t* pad, float* score, const float* count, int B, int G, float length_alpha, hipStream_t stream);
/* Beam step (embedding_decoder.py:905-978), one workgroup per sample: temperature, finished beams emit END with log-prob 0, log-softmax,
 * + running score, END banned for beam 0 at step 1, length-normalised ranking, top-H over H*V (ties: lowest h*V+v), reorder of ids / padding /
 * scores / lengths from the *_in to the *_out buffers (ping-pong); src_out (may be NULL) records the old beam each new beam continues. */
int novic_beam_step(const void* logits_bf16, int ldl, int V, int B, int H, int G, int step, const void* ids_in, void* ids_out, int tok_bytes, const uint8_t* pad_in,
                    uint8_t* pad_out, const float* score_in, float* score_out, float* score_normed, const float* len_in, float* len_out, int* active, int* src_out,
                    float temperature, float length_alpha, hipStream_t stream);
int novic_beam_step_next(const void* logits_bf16, int ldl, int V, int B, int H, int G, int step, const void* ids_in, void* ids_out, int tok_bytes, const uint8_t* pad_in,
                         uint8_t* pad_out, const float* score_in, float* score_out, float* score_normed, const float* len_in, float* len_out, int* active, int* src_out,
                         float temperature, float length_alpha, const novic_next_embed_t* next, hipStream_t stream);
/* Diagnostic: 1 = always the workgroup-per-sample selection kernel, 0 = one wave per beam row where the vocabulary allows (V <= 8192; default), < 0 = query.
 * Returns the previous setting. */
int novic_beam_step_policy(int generic);
/* Diagnostic: the [M x 512 x 512] bf16-store GEMM on the resident-weight streaming kernel as four 128-column blocks (0, default) or two 256-column blocks (1);
 * < 0 queries.  Returns the previous setting; results are bit-identical. */
int novic_skinny_wide_policy(int wide);
int novic_mask_ids(void* ids, int tok_bytes, const uint8_t* pad, int n, hipStream_t stream);
/* ABI 12.  The early-exit check of a decode loop (embedding_decoder.py:819-820: `if not alive.any(): break`; :965-967 for beams) without a device -> host copy or an event per
 * step: novic_step_done -- one thread, enqueued behind a step's selection kernel (the last node of the step's hipGraph) -- writes *done_flag = 2 if *active != 0 else 1 with a
 * system-scope store (relaxed: the host reads nothing but the word); done_flag is the DEVICE address (novic_host_mapped_ptr) of a word of page-locked host memory that the host set to 0 before the call and polls.
 * novic_host_mapped_ptr: the device address of page-locked, device-mapped host memory (hipHostMalloc; a pinned torch tensor) -- an error for anything else.  No stream operation:
 * call it when the buffer is made, not inside a capture. */
int novic_step_done(const int* active, int* done_flag_dev, hipStream_t stream);
int novic_host_mapped_ptr(void* host, void** dev);
/* Guided variants (embedding_decoder.py:788, :808-813; :915-943, :969-975): the set of nouns a beam may still spell is a node of a token trie
 * (CSR: trie_start[nodes+1], trie_tok / trie_next[edges], children sorted by token, next = -1 on END edges); node state: >= 0 on the trie,
 * -1 finished, -2 dead.  renorm: probabilities renormalised over the allowed tokens.  trie_logprior (may be NULL): log P(token | prefix) among
 * the vocabulary nouns, subtracted prior_scale times (vocab_targets / vocab_scaler).  Beams beyond the number of allowed continuations come
 * out dead with score -inf. */
int novic_beam_step_guided(const void* logits_bf16, int ldl, int V, int B, int H, int G, int step, const void* ids_in, void* ids_out, int tok_bytes, const uint8_t* pad_in,
                           uint8_t* pad_out, const float* score_in, float* score_out, float* score_normed, const float* len_in, float* len_out, int* active, int* src_out,
                           const int* node_in, int* node_out, const int* trie_start, const int* trie_tok, const int* trie_next, const float* trie_logprior, float prior_scale,
                           int renorm, float temperature, float length_alpha, hipStream_t stream);
/* The same beam step when the vocabulary nouns of the prior are NOT the guide nouns: a beam carries a second node (vnode, same state encoding) on the vocabulary
 * trie; a candidate token's prior is vocab_logprior of its edge under that node, and a token without such an edge is banned (prior probability 0). */
int novic_beam_step_guided_vocab(const void* logits_bf16, int ldl, int V, int B, int H, int G, int step, const void* ids_in, void* ids_out, int tok_bytes, const uint8_t* pad_in,
                                 uint8_t* pad_out, const float* score_in, float* score_out, float* score_normed, const float* len_in, float* len_out, int* active, int* src_out,
                                 const int* node_in, int* node_out, const int* trie_start, const int* trie_tok, const int* trie_next, const int* vnode_in, int* vnode_out,
                                 const int* vocab_start, const int* vocab_tok, const int* vocab_next, const float* vocab_logprior, float prior_scale, int renorm, float temperature,
                                 float length_alpha, hipStream_t stream);
int novic_greedy_step_guided(const void* logits_bf16, int ldl, int V, int B, int G, int step, void* ids, int tok_bytes, uint8_t* pad, float* alive, float* score, float* nll,
                             float* count, int* active, float* step_logits, int* node, const int* trie_start, const int* trie_tok, const int* trie_next, int renorm,
                             float temperature, float label_smoothing, hipStream_t stream);
/* KV-cached decode step helpers (outputs identical to re-running the whole sequence): prefix keys/values come from the step-1
 * qkv buffer [B*P][3E] shared by a sample's `beams` sequences, label keys/values from the per-sequence cache [A][G][E].
 * decode_embed: x[a] = W_tok[ids[a*G + col]] + pos_row;  decode_attn: append k,v at label position `pos`, attend (P + pos + 1 <= 32 keys);
 * kv_reorder: for every layer l, sequence a: cache_out[l][a][0..npos) = cache_in[l][(a / beams) * beams + src_idx[a]][0..npos). */
int novic_decode_embed(const void* ids, int tok_bytes, int G, int col, const float* wtok, const float* pos_row, float* x, int A, int E, int V, hipStream_t stream);
int novic_decode_attn(const void* qkv_new_bf16, const void* prefix_qkv_bf16, void* cache_k_bf16, void* cache_v_bf16, void* o_bf16, int A, int H, int D, int P, int G, int pos,
                      int beams, const int* origin, hipStream_t stream);
/* origin (decode_attn; NULL = every sequence reads its own cache row): [A][G] ints, origin[a][g] = the cache ROW holding label position g of sequence a.
 * Beam search then never moves K/V: after a beam step, origin_out[a][g] = g == npos - 1 ? sa : origin_in[sa][g] for g < npos, sa = (a / beams) * beams +
 * src_idx[a] (the old sequence that a continues; position npos - 1 was computed -- and written into ITS row -- by the step that just ran). */
int novic_kv_origin_update(const int* src_idx, const int* origin_in, int* origin_out, int A, int beams, int G, int npos, hipStream_t stream);
int novic_kv_reorder(const void* k_in, const void* v_in, void* k_out, void* v_out, const int* src_idx, int layers, int A, int beams, int G, int E, int npos,
                     hipStream_t stream);

/* generate_all (embedding_decoder.py:1043-1079): after the ordinary forward over (sample x target-chunk) sequences a = b * Hc + h,
 * out[b][w0 + h] = sum over the unpadded positions t of target h of log-softmax(logits[a * T + t] / temperature)[targets[h][t]]; node != NULL
 * (guide_renorm): the soft-max runs over the children of trie node node[h][t] only (CSR trie_start / trie_tok, see novic_beam_step_guided). */
int novic_score_targets(const void* logits_bf16, int ldl, int V, const void* targets, int tok_bytes, const uint8_t* pad, const int* node, const int* trie_start,
                        const int* trie_tok, float* out, int ldo, int w0, int B, int Hc, int T, float temperature, hipStream_t stream);
/* out_val / out_idx [B][K]: the K largest (scores[b][i] - adjust_scale * adjust[i]) * scale[i] of every row in descending order, ties towards the lower
 * index (torch.topk(sorted=True) of :1076 with a defined tie-break); adjust / scale may be NULL. */
int novic_topk_rows(const float* scores, int B, int W, int lds, const float* adjust, float adjust_scale, const float* scale, int K, float* out_val, int* out_idx,
                    hipStream_t stream);

/* Guided teacher-forced correctness of forward(..., calc_correct=True, guide_targets=...) (embedding_decoder.py:756-763): correct[a * T + t] = 1 iff position t of
 * sequence a is not padding (out_pad [A][T], may be NULL) and the arg-max of its logits over the children of the trie node reached by walking targets[a][:t]
 * equals targets[a][t] (prediction 0 once the target has left the trie). */
int novic_guided_correct(const void* logits_bf16, int ldl, const void* targets, int tok_bytes, int tok_ld, const uint8_t* out_pad, const int* trie_start, const int* trie_tok,
                         const int* trie_next, uint8_t* correct, int A, int T, hipStream_t stream);

/* Small-tile layer kernels of a KV-cached decode step (decode_fused.hip): 16-row x 64-column workgroups so that every CU streams a little of the
 * weights; replace novic_layernorm_fwd + novic_gemm_bf16 of nn.TransformerEncoderLayer (norm_first, bias-free, embedding_decoder.py:309-327, called
 * :714) for one new position per sequence, same arithmetic and rounding points.  Supported: novic_decode_fused_supported(E, Kf). */
int novic_decode_fused_supported(int E, int Kf);
/* y[M][ldy] (bf16) = act(LayerNorm(x[M][E]; gamma, eps) W[N][E]^T), act = identity (gelu = 0) or bf16(GELU(bf16(.))) (gelu = 1) */
int novic_decode_ln_gemm(const float* x, const float* gamma, const void* w_bf16, void* y_bf16, int M, int N, int E, int ldy, int gelu, float eps, hipStream_t stream);
/* out[M][N] (f32) = resid[M][N] + bf16(a[M][K] W[N][K]^T); K in {32, 64, 128, 256, 512}; out may alias resid */
int novic_decode_gemm_resid(const void* a_bf16, const void* w_bf16, const float* resid, float* out, int M, int N, int K, hipStream_t stream);
/* The feed-forward half of a decode layer as one launch: out[M][E] (f32) = x + GELU(LayerNorm(x; gamma, eps) W1^T) W2^T with W1 bf16 [Kf][E], W2 bf16 [E][Kf], the arithmetic and
 * rounding points of novic_decode_ln_gemm(gelu = 1) followed by novic_decode_gemm_resid (bit-identical); out must not be x.  Supported: novic_decode_ffn_supported(E, Kf). */
int novic_decode_ffn_supported(int E, int Kf);
int novic_decode_ffn(const float* x, const float* gamma, const void* w1_bf16, const void* w2_bf16, float* out, int M, int E, int Kf, float eps, hipStream_t stream);
/* y[M][ldy] (bf16) = act(a[M][K] W[N][K]^T), same small-tile kernel with a bf16 store / GELU epilogue (the LayerNorm done by novic_layernorm_fwd) */
int novic_decode_gemm(const void* a_bf16, const void* w_bf16, void* y_bf16, int M, int N, int K, int ldy, int gelu, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * CLIP ViT image tower (embedders.py:593-594, :763-764, :906-907 -> third-party encode_image).  Linear layers and LayerNorms are
 * novic_gemm_bf16 (bias / GELU / QuickGELU / residual epilogues) and novic_layernorm_fwd launches; these are the remaining pieces.
 * ------------------------------------------------------------------------------------------------------------ */
/* images [B][3][R][R] f32 -> patches [B*(R/patch)^2][k_padded] bf16, k = c*patch^2 + y*patch + x (conv1.weight.view(W,-1) order), zero padded. */
int novic_vit_im2col(const float* images, void* patches_bf16, int B, int R, int patch, int k_padded, hipStream_t stream);
/* The same from the transform's uint8 pixels (the image BEFORE torchvision's ToTensor / Normalize: embedders.py:755-757 returns open_clip's transform, whose last two steps
 * are x / 255 and (x - mean) / std in fp32): the kernel applies exactly those operations per pixel, so the patch matrix is bit-identical to novic_vit_im2col's on the
 * normalised fp32 images, for a quarter of the bytes over PCIe (38.5 instead of 154 MB per 256 images of 224 x 224).  images [B][3][R][R] uint8. */
typedef struct novic_pixel_norm { float mean[3]; float std[3]; } novic_pixel_norm_t;
int novic_vit_im2col_u8(const uint8_t* images, void* patches_bf16, int B, int R, int patch, int k_padded, novic_pixel_norm_t norm, hipStream_t stream);
/* x[b][t] = ln_pre((t == 0 ? cls : patches[b][t-1]) + pos[t]), f32 [B*N][W]; ln_gamma/ln_beta may both be NULL (no ln_pre). */
int novic_vit_embed(const void* patches_bf16, const float* cls, const float* pos, const float* ln_gamma, const float* ln_beta, float* x, int B, int N, int W, float eps,
                    hipStream_t stream);
/* ABI 11: the same with the stream written as IEEE half, x_f16 [B*N][W] (sum and ln_pre in fp32, rounded once): the first rows of a half-precision tower's residual stream. */
int novic_vit_embed_f16(const void* patches_bf16, const float* cls, const float* pos, const float* ln_gamma, const float* ln_beta, void* x_f16, int B, int N, int W, float eps,
                        hipStream_t stream);
/* Non-causal softmax(QK^T / sqrt(D)) V per (image, head); qkv [B*N][3*H*D] bf16 -> o [B*N][H*D] bf16; head_dim 32 | 64 | 80, any N. */
int novic_vit_attn_fwd(const void* qkv_bf16, void* o_bf16, int B, int N, int H, int D, hipStream_t stream);
/* Diagnostic / tests: 0 = streaming attention kernel only, 1 = the K/V-resident kernel where a head's K and V fit into half a CU's LDS and there
 * is more than one 64-query block, the blocked kernel (128-key blocks through two LDS buffers) beyond 288 keys at head_dim 64 / 80 (default; results
 * within the bf16 rounding of the probabilities of each other; the blocked kernel runs as eight waves x one query tile -- the four-wave form of round 3 went in ABI 9);
 * any other value only queries.  Returns the kernel policy (0 | 1) that was in force. */
int novic_vit_attn_policy(int policy);
/* The same attention with an optional causal mask (query i sees keys j <= i): CLIP text tower. */
int novic_clip_attn_fwd(const void* qkv_bf16, void* o_bf16, int B, int N, int H, int D, int causal, hipStream_t stream);
/* ... and with the soft-max scale given by the caller (scale <= 0: 1 / sqrt(D)): heads whose real width is not one of 32 | 64 | 80 run zero-padded to
 * the next one (ViT-SO400M-14-SigLIP: 1152 / 16 = 72 -> 80, scale 72^-1/2; the padding columns of Q, K, V are zeros of the padded weights). */
int novic_clip_attn_fwd_scaled(const void* qkv_bf16, void* o_bf16, int B, int N, int H, int D, int causal, float scale, hipStream_t stream);
/* CLIP text tower (embedders.py:423-426, :557-583, :728-753 -> third-party encode_text): x[b*S + s] = tok_emb[ids[b][s]] + pos[s] (f32);
 * out[b] = x[b][s*] with s* = argmax_s ids[b][s] (eot_id < 0: CLIP's vocabulary, END-OF-TEXT has the largest id; first maximum) or the first s
 * whose id equals eot_id (position 0 if none). */
int novic_text_embed(const void* ids, int tok_bytes, const float* tok_emb, const float* pos, float* x, int B, int S, int W, int V, hipStream_t stream);
int novic_text_pool(const void* ids, int tok_bytes, const float* x, float* out, int B, int S, int W, long long eot_id, hipStream_t stream);
/* ABI 11: the same two with the residual stream as IEEE half (x_f16 [B*S][W]): clip's fp16 text tower, which the reference runs for 'openai:' embedders (embedders.py:488-489).
 * The pooled row leaves as fp32 (an exact conversion) for the final LayerNorm. */
int novic_text_embed_f16(const void* ids, int tok_bytes, const float* tok_emb, const float* pos, void* x_f16, int B, int S, int W, int V, hipStream_t stream);
int novic_text_pool_f16(const void* ids, int tok_bytes, const void* x_f16, float* out, int B, int S, int W, long long eot_id, hipStream_t stream);
/* y[r] = x[r] / max(||x[r]||, 1e-12) in f32 (the final F.normalize of inference_image, embedders.py:764). */
int novic_rownorm_f32(const float* x, float* y, int rows, int E, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Batch assembly from an HBM-resident embedding cache (embedding_cache.py:690-723 get_samples, :832-895 batch rules): batch row b reads
 * cache row (start + b) % N; target/mask rows are gathered from the noun token table through the per-embedding target ids (first M of
 * M_file targets, first C of C_file token columns); weight_mode 0 = copy, 1 = L1-normalise over the kept M, 2 = ones.
 * target_ids == NULL: embeddings only.
 * ------------------------------------------------------------------------------------------------------------ */
int novic_cache_gather(const float* embeds, const int* target_ids, const void* token_table, int tok_bytes, const uint8_t* mask_table, const float* weights, int64_t start,
                       int B, int64_t N, int F, int M_file, int C_file, int M, int C, float* out_embed, void* out_target, uint8_t* out_mask, float* out_weight,
                       int weight_mode, int64_t staged_row0, hipStream_t stream);
/* staged_row0 (novic_cache_gather): -1 = `embeds` is the whole [N][F] table; >= 0 = `embeds` is a staged slab whose rows staged_row0 .. + B - 1 already hold the
 * batch's embedding rows in order (streaming loader for caches larger than the HBM budget); target ids / weights are still indexed by (start + b) % N. */
/* The micro-batches of ONE optimizer step in one launch (the train action's loader: GradAccum hands `accum` loader batches to a step, train.py:1247-1263; sixteen launches of
 * 512 rows were sixteen ctypes calls and as many host-side index computations per step): `groups` (1..32) batches of B_each rows, batch g reading cache rows
 * (starts[g] + b) % N; the outputs are [groups * B_each] rows, batch after batch -- each batch exactly what novic_cache_gather(start = starts[g], B = B_each) writes.
 * `starts` is HOST memory (copied into the launch).  staged_row0 as above: >= 0 = `embeds` is a staged buffer whose rows staged_row0 + g * B_each + b hold batch g's rows. */
int novic_cache_gather_group(const float* embeds, const int* target_ids, const void* token_table, int tok_bytes, const uint8_t* mask_table, const float* weights,
                             const int64_t* starts, int groups, int B_each, int64_t N, int F, int M_file, int C_file, int M, int C, float* out_embed, void* out_target,
                             uint8_t* out_mask, float* out_weight, int weight_mode, int64_t staged_row0, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NOVIC_HIP_H */
