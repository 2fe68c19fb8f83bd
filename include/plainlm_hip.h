/* plainlm_hip.h — C ABI of libplainlm_hip.so (MI355X / gfx950 only).
 *
 * plainLM has no FFI of its own: its hot path is Python calling torch ops
 * (SURVEY.md §8b).  Each entry point below therefore cites the torch call site
 * in the reference that it replaces (paths relative to the reference root).
 *
 * Conventions (identical for every function):
 *   - all pointers are DEVICE pointers owned by the caller (workspaces too);
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on
 *     that stream, never synchronises, never allocates;
 *   - bf16 tensors are passed as uint16_t* (raw bfloat16 bits), row-major;
 *   - return value: 0 = ok, <0 = error (PLM_E_*); plm_last_error_string()
 *     returns a thread-local description of the last failure on this thread;
 *   - re-entrant and thread-safe per stream (autograd's backward thread differs
 *     from the forward thread).
 */
#ifndef PLAINLM_HIP_H
#define PLAINLM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLM_OK 0
#define PLM_E_INVALID (-1)     /* bad argument / unsupported shape */
#define PLM_E_HIP (-2)         /* a HIP runtime call or launch failed */
#define PLM_E_COMM (-3)        /* RCCL failure */
#define PLM_E_WORKSPACE (-4)   /* caller-provided workspace too small */

int plm_version(void);
const char* plm_last_error_string(void);
/* The library reads its environment switches (PLM_GEMM_V1, PLM_TN_NO_BIG, PLM_NT_NO_HYBRID, PLM_NT_HYBRID_MIN_K: tests and A/B
 * runs; none is needed in production) ONCE, at its first call - no launch path calls getenv.  A process that changes one of them
 * afterwards (the test-suite does) calls this to have them read again. */
void plm_reload_env(void);

/* ---- parameter casts --------------------------------------------------
 * Replaces autocast's per-forward fp32->bf16 weight casts
 * (engine/engine.py:75,108-109).  The `_t` form also writes the transposed
 * copy dst_t[cols, ld_t] (ld_t >= rows; columns rows..ld_t are left untouched) used by the dX GEMMs. */
int plm_cast_f32_bf16(const float* src, uint16_t* dst, int64_t n, void* stream);
int plm_cast_f32_bf16_t(const float* src, uint16_t* dst, uint16_t* dst_t, int64_t rows, int64_t cols, int64_t ld_t, void* stream);
/* the `_t` cast for a whole list of weights in one launch (every Linear of the model at the top of a step) */
typedef struct plm_cast_item {
  const float* src; /* fp32 [rows, cols] */
  uint16_t* dst;    /* bf16 [rows, cols] */
  uint16_t* dst_t;  /* bf16 [cols, ld_t], columns 0..rows-1 written */
  int64_t rows, cols, ld_t;
} plm_cast_item;
int plm_cast_f32_bf16_t_multi(const plm_cast_item* items, int count, void* stream);

/* ---- embedding (models/transformer.py:94,110) --------------------------
 * fwd: out[m,:] = W[ids[m],:]           ids int64[M], W fp32[V,d], out fp32[M,d]
 * bwd: dW[ids[m],:] += dout[m,:]        (atomic fp32 adds; dW must be initialised) */
int plm_embed_fwd(const int64_t* ids, const float* W, float* out, int64_t M, int64_t d, int64_t V, void* stream);
int plm_embed_bwd(const int64_t* ids, const float* dout, float* dW, int64_t M, int64_t d, int64_t V, void* stream);
/* Deterministic variant without atomics: stable radix sort of the ids, then every vocabulary row is written exactly once as
 * the sum of its tokens' rows in increasing token order (rows without tokens: zeros when accumulate == 0, untouched
 * otherwise).  Needs M <= 65536 and V < 65536 and a workspace of plm_embed_bwd_workspace_bytes(M, V) bytes (0 = shape not
 * supported, use plm_embed_bwd).  With accumulate == 0 dW does not have to be cleared first. */
size_t plm_embed_bwd_workspace_bytes(int64_t M, int64_t V);
int plm_embed_bwd_sorted(const int64_t* ids, const float* dout, float* dW, int64_t M, int64_t d, int64_t V, int accumulate,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ---- RMSNorm (+ residual add) (models/components.py:16-28, transformer.py:81-82)
 * fwd:  r = x (+ branch if branch != NULL)          x fp32[M,d], branch bf16[M,d]
 *       if xout != NULL: xout = r                    fp32[M,d]  (may alias x)
 *       rstd[m] = rsqrt(mean(r^2) + eps)             fp32[M]
 *       y = bf16((r * rstd) * w)                     bf16[M,d]
 * bwd:  a = dy * w ; dxn = rstd*a - x*rstd^3*mean(a*x)
 *       dx = (gin ? gin : 0) + dxn                   fp32[M,d]  (may alias gin)
 *       if dx_bf16 != NULL: dx_bf16 = bf16(dx)
 *       dw_partial[blk,:] = sum over the block's rows of dy*x*rstd   fp32[nblk,d]
 *       nblk = plm_rmsnorm_bwd_blocks(M); reduce with plm_colsum_f32. */
int plm_rmsnorm_fwd(const float* x, const uint16_t* branch, float* xout, const float* w, uint16_t* y, float* rstd,
                    int64_t M, int64_t d, float eps, void* stream);
int64_t plm_rmsnorm_bwd_blocks(int64_t M);
int plm_rmsnorm_bwd(const uint16_t* dy, const float* x, const float* w, const float* rstd, const float* gin,
                    float* dx, uint16_t* dx_bf16, float* dw_partial, int64_t M, int64_t d, void* stream);
/* out[j] (+)= sum_r part[r, j]; accumulate != 0 adds into out. */
int plm_colsum_f32(const float* part, float* out, int64_t rows, int64_t cols, int accumulate, void* stream);
/* the same reduction for a whole list of partial buffers in ONE launch (all RMSNorm weight gradients of a backward pass:
 * 25 at the 160M size, each a few microseconds of launch-bound work); every item has the same rows x cols */
typedef struct plm_colsum_item {
  const float* part; /* fp32 [rows, cols] */
  float* out;        /* fp32 [cols] */
  int accumulate;
} plm_colsum_item;
int plm_colsum_f32_multi(const plm_colsum_item* items, int count, int64_t rows, int64_t cols, void* stream);

/* ---- SwiGLU gate (models/components.py:55-56) ---------------------------
 * u bf16[M,2h] = fc1 output, x = u[:, :h], z = u[:, h:]
 * fwd: out = bf16(bf16(silu(x)) * z)                          bf16[M,h]
 * bwd: du[:, :h] = d/dx, du[:, h:] = d/dz (bf16 autograd chain of the reference) */
int plm_swiglu_fwd(const uint16_t* u, uint16_t* out, int64_t M, int64_t h, void* stream);
int plm_swiglu_bwd(const uint16_t* dout, const uint16_t* u, uint16_t* du, int64_t M, int64_t h, void* stream);
/* The activations of the reference's two plain MLP classes (models/components.py:31-40 `MLP` = fc2(silu(fc1 x)); :59-70 `MLPReluSquared` =
 * fc2(relu(fc1 x)^2); models/transformer.py:26 MLP_CLASSES), element-wise over n = M * h bf16 values (n % 8 == 0, 16-byte aligned):
 * kind 0 = silu, 1 = relu squared; fp32 math, bf16 results in the autocast rounding order.  plm_act_bwd: du = dout * act'(u). */
int plm_act_fwd(const uint16_t* u, uint16_t* out, int64_t n, int kind, void* stream);
int plm_act_bwd(const uint16_t* dout, const uint16_t* u, uint16_t* du, int64_t n, int kind, void* stream);

/* ---- bf16 MFMA GEMMs (nn.Linear: transformer.py:36-37,42,67,97,114; components.py:50-51)
 * nt: C[M,N] = alpha * A[M,K] · B[N,K]^T        A,B bf16 K-contiguous (y = x W^T, and dX = dY (W^T)^T)
 * tn: C[M,N] (+)= alpha * A[K,M]^T · B[K,N]     A,B bf16, contraction over rows (dW = dY^T X), C fp32
 * c_dtype: 0 = bf16, 1 = fp32.  accumulate (fp32 C only): C += result.
 * alpha_dev: optional device pointer to one fp32 scale (NULL = 1).
 * Requirements: K % 8 == 0 (nt) ; M % 8 == 0 and N % 8 == 0 (tn); ld* % 8 == 0; 16-byte aligned bases.
 * tn splits the contraction over `splits` slabs when M*N is too small to fill the chip; it needs a
 * fp32 workspace of plm_gemm_tn_workspace_bytes(M,N,K) bytes (0 when no split is used). */
int plm_gemm_bf16_nt(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, void* C, int64_t ldc,
                     int64_t M, int64_t N, int64_t K, int c_dtype, int accumulate, const float* alpha_dev, void* stream);
/* same as plm_gemm_bf16_nt with an explicit kernel choice (autotuning / A-B measurements / tests):
 * variant 0 = automatic, 1 = 128x128 register-staged, 2 = 128x128 LDS-DMA double-buffered,
 * 4 / 5 / 6 / 7 = persistent 256x256 / 256x192 / 256x128 / 128x192 with the deep-prefetch ring (half-tile slots refilled two
 *             K-tiles ahead) and offset wave groups - the kernels the automatic policy chooses from (128x192, round 5: the short
 *             batches, M = 8192 of config_doc_mask.yaml:35); 3 = 4 (kept for callers of round 1)
 * (3..7: bf16 C, no accumulate; 2..7: K % 64 == 0, N % 8 == 0, ldc % 8 == 0). */
int plm_gemm_bf16_nt_ex(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, void* C, int64_t ldc,
                        int64_t M, int64_t N, int64_t K, int c_dtype, int accumulate, const float* alpha_dev, int variant,
                        void* stream);
/* nt with an optional caller-owned fp32 workspace (plm_gemm_nt_workspace_bytes; 0 = none needed): with it the automatic
 * choice may use the hybrid whole-K + split-K schedule on 256x256 tiles, which removes the round quantisation of shapes
 * such as N = 768 (384 tiles on 256 CUs).  Without a workspace the call behaves exactly like plm_gemm_bf16_nt_ex. */
size_t plm_gemm_nt_workspace_bytes(int64_t M, int64_t N, int64_t K);
int plm_gemm_bf16_nt_ws(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, void* C, int64_t ldc,
                        int64_t M, int64_t N, int64_t K, int c_dtype, int accumulate, const float* alpha_dev, int variant,
                        void* workspace, size_t workspace_bytes, void* stream);
size_t plm_gemm_tn_workspace_bytes(int64_t M, int64_t N, int64_t K);
/* Several tn problems with the SAME contraction length K (the dW GEMMs of one or more transformer blocks: engine/engine.py's
 * backward reaches them one after the other, none of them has a consumer inside backward) as ONE launch over the union of
 * their 256x256 output tiles - whole-K tiles written directly for the full rounds of the persistent grid, split-K + one
 * reduce for the remaining tiles: C_p[M_p, N_p] (+)= alpha_p * A_p[K, M_p]^T B_p[K, N_p].  count <= 48; row strides < 2^31;
 * workspace from the query below (0 = shapes cannot be grouped). */
typedef struct plm_tn_problem {
  const uint16_t* A; int64_t lda;
  const uint16_t* B; int64_t ldb;
  float* C; int64_t ldc;
  int64_t M, N;
  int accumulate;
  const float* alpha_dev;
} plm_tn_problem;
size_t plm_gemm_tn_grouped_workspace_bytes(const int64_t* Ms, const int64_t* Ns, int count, int64_t K);
int plm_gemm_bf16_tn_grouped(const plm_tn_problem* probs, int count, int64_t K, void* workspace, size_t workspace_bytes,
                             void* stream);
int plm_gemm_bf16_tn(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc,
                     int64_t M, int64_t N, int64_t K, int accumulate, const float* alpha_dev,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ---- RoPE + causal / document-masked attention
 * (models/transformer.py:43-65, models/embeddings.py:15-30, data/datasets/data_prep_utils.py:7-23)
 * qkv bf16[B*T, 3*nh*hd] straight from w_qkv (q | k | v column blocks, head h = cols h*hd..), T % 4 == 0; hd == 64 takes the tuned kernels
 * described below (every shipped config), hd == 32 or 128 (models/transformer.py:34 allows any dim // n_heads) a plain kernel family
 * (csrc/attn_generic.hip: element-wise masks, no plan needed, the inverse rotation as a separate in-place pass) behind the same entry points.
 * rope_cos/sin fp32[T, hd/2] (interleaved-pair convention).  doc_start int32[B,T] or NULL (pure causal):
 * query i attends key j iff doc_start[i] <= j <= i (doc_start non-decreasing in i).
 * plm_rope_qk : rotates the q and k blocks of qkv IN PLACE (fp32 math, bf16 result).  The training step does not launch it: the
 *               rotation is fused into plm_qkv_rope_bf16's GEMM epilogue; this is that entry point's fallback and the tests' yardstick.
 * Two kernel families behind the same two entry points: doc_start == NULL (plain causal batches) takes the 256-query-tile kernels of
 * csrc/attn_causal.hip, a document mask the 128-row-tile kernels of csrc/attn_doc.hip (+ the DOC mode of the causal dK/dV kernel); lse / delta are only meaningful within one
 * forward / backward pair of the same family (pass the same doc_start to both).
 * plm_attn_fwd: qkv with q,k ALREADY rotated -> out bf16[B*T, nh*hd], lse fp32[B, nh, T] (BASE-2 log-sum-exp of the
 *               scaled scores, = LSE / ln 2: the form plm_attn_bwd's exp2 consumes; opaque to the caller otherwise).  No transposed / contiguous copies of q, k, v are made anywhere.
 * plm_attn_bwd: same rotated qkv; dqkv bf16[B*T, 3*nh*hd] = gradient w.r.t. the PRE-rotation q, k (the inverse
 *               rotation is applied in the kernels' epilogues) and v; delta fp32[B,nh,T] scratch (+-rowsum(dO * O), written by the dQ
 *               kernel and read by the dK/dV kernel that follows it; stored negated: opaque to the caller).
 * qkv, out, dout, dqkv and the RoPE tables must be 16-byte aligned (LDS-DMA sources, whole-row 16-byte epilogue stores); misaligned
 * pointers are refused with PLM_E_INVALID. */
int plm_rope_qk(uint16_t* qkv, const float* rope_cos, const float* rope_sin, int64_t B, int64_t T, int64_t nh, int64_t hd,
                void* stream);
/* w_qkv projection + RoPE (transformer.py:42-47): QKV[M, 3*nh*hd] = X[M,K] W[3*nh*hd, K]^T with the q | k column blocks rotated
 * (ldq must be 3*nh*hd).  Default path: ONE launch - the rotation happens on the store side of the GEMM's epilogue (the bf16 values
 * the GEMM would have written are rotated on their way from the epilogue's LDS transposition scratch to memory; K % 64 == 0,
 * M >= 512, 16-byte aligned operands).  Other shapes, and PLM_GEMM_V1, take the fallback: the plain GEMM followed by the in-place
 * plm_rope_qk pass.  Both paths produce the same bits. */
int plm_qkv_rope_bf16(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, uint16_t* QKV, int64_t ldq, int64_t M,
                      int64_t K, const float* rope_cos, const float* rope_sin, int64_t B, int64_t T, int64_t nh, int64_t hd,
                      void* stream);
/* fc1 of the SwiGLU MLP with the activation in the GEMM epilogue (models/components.py:50-56):
 * U[M, 2h] = X[M,K] W[2h,K]^T (gate | up columns, dense, kept for backward) and ACT[M, h] = bf16(bf16(silu(gate)) * up), the
 * reference's autocast rounding order.  One launch when 2h % 256 == 0, K % 64 == 0, M >= 512; otherwise GEMM + plm_swiglu_fwd
 * (same bits). */
int plm_fc1_swiglu_bf16(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, uint16_t* U, uint16_t* ACT, int64_t M,
                        int64_t h, int64_t K, void* stream);
/* dX of fc2 with the SwiGLU backward in the GEMM epilogue: DU[M, 2h] = swiglu_bwd(dY[M,K] W2T[h,K]^T, U[M,2h]) where W2T is the
 * transposed bf16 copy of fc2's weight and U the saved fc1 output; d(act) is never stored.  One launch when h % 256 == 0,
 * K % 64 == 0, M >= 512; otherwise GEMM + plm_swiglu_bwd through `scratch` (M*h bf16).  With scratch == NULL the call either takes the
 * one-launch path or returns PLM_E_WORKSPACE without launching anything. */
int plm_fc2_dx_swiglu_bwd_bf16(const uint16_t* dY, int64_t lddy, const uint16_t* W2T, int64_t ldw, const uint16_t* U, uint16_t* DU,
                               uint16_t* scratch, int64_t M, int64_t h, int64_t K, void* stream);
/* Document masks (models/transformer.py:52-61 with the masks of data/datasets/data_prep_utils.py:7-23, stacked at engine/engine.py:21-23): a batch's
 * doc_start[B,T] comes with a PLAN, built once per batch by plm_attn_doc_plan and shared by every layer's forward and backward launches:
 * doc_end[B,T] (the first query that no longer sees a key: the dK/dV kernel's mask becomes two thresholds per key) and the batch's 128-row
 * query / key tiles sorted by the number of 64-row tiles they actually walk, heaviest first - the launch order of the kernels; when the
 * grid of B * ceil(T/128) * nh workgroups is resident at once, the heaviest tiles enter the lists as two 64-row halves (half the chain each).
 * plan is caller-owned device memory of plm_attn_doc_plan_bytes(B, T) bytes, 16-byte aligned; doc_start must be non-decreasing along T with
 * doc_start[b][i] <= i.  plm_attn_fwd / plm_attn_bwd: doc_start == NULL -> causal (doc_plan ignored); otherwise doc_plan is required
 * (PLM_E_INVALID without it). */
/* The reference's own mask format -> doc_start: mask is the bool [B, T, T] tensor of engine/engine.py:21-23 (one byte per element, nonzero =
 * may attend), doc_start[b][i] = first allowed key of row i.  Every row is checked to be exactly [doc_start, i] (the block-diagonal causal
 * masks of data_prep_utils.py:7-23 are); status[0] (caller-zeroed int32) is set to 1 when some row is not - such a mask cannot be expressed
 * as doc_start and the caller must refuse it. */
int plm_attn_doc_start_from_mask(const uint8_t* mask, int32_t* doc_start, int32_t* status, int64_t B, int64_t T, void* stream);
int64_t plm_attn_doc_plan_bytes(int64_t B, int64_t T);
int plm_attn_doc_plan(const int32_t* doc_start, int32_t* plan, int64_t B, int64_t T, int64_t nh, void* stream);
int plm_attn_fwd(const uint16_t* qkv, const int32_t* doc_start, const int32_t* doc_plan, uint16_t* out, float* lse, int64_t B, int64_t T,
                 int64_t nh, int64_t hd, void* stream);
int plm_attn_bwd(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* lse,
                 const float* rope_cos, const float* rope_sin, const int32_t* doc_start, const int32_t* doc_plan,
                 uint16_t* dqkv, float* delta, int64_t B, int64_t T, int64_t nh, int64_t hd, void* stream);

/* ---- fused cross-entropy forward+backward (engine/engine.py:81,111) -----
 * logits bf16[M, ld] (row stride ld >= V) are OVERWRITTEN with dlogits = (softmax - onehot) * grad_scale
 * (grad_scale = g/M); pad columns V..ld are set to zero so the buffer can feed a GEMM with K = ld.
 * loss_rows fp32[M] = logsumexp(l) - l[target].  Then plm_mean_f32 reduces to the scalar mean. */
int plm_ce_fwd_bwd(uint16_t* logits, const int64_t* targets, float* loss_rows, int64_t M, int64_t V, int64_t ld,
                   float grad_scale, void* stream);
int plm_mean_f32(const float* x, float* out, int64_t n, void* stream);

/* ---- device-scalar scaling (chunked lm_head + cross-entropy, models/transformer.py:114 + engine/engine.py:111,118)
 * The chunked head runs its dX / dW GEMMs inside forward, before autograd hands over the upstream gradient g
 * (engine.py:118 `(loss / accum).backward()`); backward applies g with
 *   plm_scale_bf16: x[i] <- bf16(x[i] * *alpha_dev)                       (n elements, in place)
 *   plm_axpy_f32:   out[i] <- (accumulate ? out[i] : 0) + *alpha_dev * x[i]  (alpha_dev NULL = 1.0) */
int plm_scale_bf16(uint16_t* x, int64_t n, const float* alpha_dev, void* stream);
int plm_axpy_f32(float* out, const float* x, int64_t n, const float* alpha_dev, int accumulate, void* stream);

/* ---- optimizer tail (engine/engine.py:126-135, optim/init_optim.py:14-21)
 * sumsq: out[0] = sum(x^2) (single-launch deterministic two-stage reduce; scratch >= 4096 floats)
 * adamw: decoupled-weight-decay Adam on a flat fp32 span, matching torch.optim.AdamW:
 *   p *= 1 - lr*wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
 *   p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps), with g pre-multiplied by *clip_coef_dev (NULL = 1). */
int plm_sumsq_f32(const float* x, int64_t n, float* scratch, float* out, void* stream);
int plm_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, float bc1, float bc2, const float* clip_coef_dev, void* stream);
/* The same update for a list of Linear weights [rows, cols] that ALSO writes the bf16 shadows the next step's GEMMs read
 * (dst = bf16(W) [rows, cols], dst_t = bf16(W)^T [cols, ld_t], columns 0..rows-1), i.e. plm_adamw_f32 followed by
 * plm_cast_f32_bf16_t_multi in one pass over the parameters ("bf16 weight shadow copy emitted by the
 * optimizer"; replaces the per-call fp32 -> bf16 weight casts of autocast, engine/engine.py:75).  Same bits as the two calls. */
typedef struct plm_adamw_item {
  float* p;         /* fp32 [rows, cols], updated in place */
  const float* g;   /* fp32 gradient */
  float* m;         /* exp_avg */
  float* v;         /* exp_avg_sq */
  uint16_t* dst;    /* bf16 [rows, cols] */
  uint16_t* dst_t;  /* bf16 [cols, ld_t] */
  int64_t rows, cols, ld_t;
} plm_adamw_item;
int plm_adamw_cast_multi(const plm_adamw_item* items, int count, float lr, float beta1, float beta2, float eps,
                         float weight_decay, float bc1, float bc2, const float* clip_coef_dev, void* stream);

/* Leave `n` CUs free when sizing the persistent GEMM grids (one workgroup per CU, static tile schedule), so that
 * concurrently running RCCL collectives do not push GEMM workgroups into a second round.  Process-wide; 0 = whole chip. */
int plm_set_cu_reserve(int n);

/* ---- RCCL data-parallel gradient exchange (replaces DDP: engine/engine.py:64-65,104-105)
 * One communicator per process (one process per GPU).  uid is the 128-byte ncclUniqueId
 * produced by plm_comm_unique_id on rank 0 and shipped to the other ranks by the host. */
typedef struct plm_comm plm_comm_t;
int plm_comm_unique_id(uint8_t uid[128]);
int plm_comm_init(plm_comm_t** comm, const uint8_t uid[128], int rank, int world_size, int device);
/* Same, with a per-communicator cap on the workgroups (= CUs) RCCL may use for one collective (ncclConfig_t.maxCTAs;
 * max_ctas <= 0: RCCL's default).  Gradient buckets reduced WHILE backward is still running go through a capped
 * communicator (the persistent GEMMs leave exactly that many CUs free, plm_set_cu_reserve), the buckets that become ready
 * when backward has ended - embed_tokens, 154 MB, ready last: engine/engine.py:104-105 leaves it exposed too - go through
 * an uncapped one: with nothing left to overlap with, the tail should use every xGMI link.  The host side (plainlm_amd/ddp.py,
 * round 5) creates the UNCAPPED communicator first (plm_comm_init) and splits the capped ones off it (plm_comm_split). */
int plm_comm_init_capped(plm_comm_t** comm, const uint8_t uid[128], int rank, int world_size, int device, int max_ctas);
/* ncclCommSplit of `parent` into a communicator over the same ranks with its own CTA cap (collective over parent's ranks). */
int plm_comm_split(plm_comm_t* parent, plm_comm_t** child, int max_ctas);
int plm_comm_destroy(plm_comm_t* comm);
/* in-place all-reduce-mean of a fp32 span on `stream` (the side stream owned by the caller) */
int plm_comm_allreduce_avg_f32(plm_comm_t* comm, float* buf, int64_t count, void* stream);
/* the same mean as reduce-scatter + all-gather in place (one-hop collectives over all xGMI links; the reducer's `algo` choice -
 * PLM_COMM_ALGO=rsag, or bench.py's autotune) */
int plm_comm_rsag_avg_f32(plm_comm_t* comm, float* buf, int64_t count, void* stream);
int plm_comm_broadcast_f32(plm_comm_t* comm, float* buf, int64_t count, int root, void* stream);

/* ---- hardware probes used by the GPU tests (semantics of gfx950 instructions the kernels rely on)
 * out int32[64*4]: element [lane*4+j] = value lane received in slot j from ds_read_b64_tr_b16 when
 * lane l supplied address base + 8*l bytes and LDS held the uint16 sequence 0,1,2,... */
int plm_probe_ds_read_tr16(int32_t* out, void* stream);
/* out fp32[32*32] = A·B for A[32,16], B[16,32] given as fp32 row-major (rounded to bf16), one wave */
int plm_probe_mfma32(const float* A, const float* B, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PLAINLM_HIP_H */
