// Attention for gfx950: the C ABI and the stand-alone RoPE pass.  The kernels live in attn_causal.hip (causal batches) and attn_doc.hip
// (batches with document masks); both replace models/transformer.py:43-65 (split, RoPE, transposes, SDPA, transpose back) and
// models/embeddings.py:15-30.  The mask of data/datasets/data_prep_utils.py:7-23 is expressed as doc_start[B,T] (query i sees key j iff
// doc_start[i] <= j <= i; doc_start is non-decreasing in i) plus the plan plm_attn_doc_plan derives from it.
//
// Common ground of all attention kernels: q, k, v are read strided straight out of the w_qkv output [B*T, 3*nh*64] (no transposed copies); all
// matrix products are v_mfma_f32_32x32x16_bf16; scores are computed TRANSPOSED (S^T[kv][q] = K Q^T) so a lane owns one query column - row max /
// row sum are in-lane plus one cross-half shuffle and the probabilities are already in the B-operand layout of the following P V product; the
// operands whose contraction index is the token row are fetched with ds_read_b64_tr_b16 from the same LDS image that serves the ds_read_b128
// operands; K / V (Q / dO) tiles are staged global -> LDS by LDS-DMA.  q and k arrive ROTATED (the w_qkv GEMM's epilogue applies RoPE,
// plm_qkv_rope_bf16; rope_qk_kernel below is that entry point's fallback and the tests' yardstick); the backward kernels apply the inverse
// rotation to dQ / dK on their way out, so dqkv is the gradient w.r.t. the PRE-rotation projection.
#include "plm_device.h"

#include "attn_common.h"

// =============================================================================================
// RoPE on the q and k column blocks of the w_qkv output, in place (models/embeddings.py:15-30):
// interleaved pairs (x[2i], x[2i+1]) -> (a cos - b sin, b cos + a sin), fp32 math, bf16 result.
// The fallback of plm_qkv_rope_bf16 (shapes its fused epilogue does not take) and the yardstick of the epilogue's bit-equality tests.
// =============================================================================================
__global__ __launch_bounds__(256) void rope_qk_kernel(uint16_t* __restrict__ qkv, const float* __restrict__ rcos,
                                                      const float* __restrict__ rsin, int64_t BT, int T, int nh, int hd, float sgn) {
  const int dm = nh * hd, ld = 3 * dm;
  const int cpr = 2 * dm / 8;  // 16-byte chunks of q|k per token row
  const int64_t total = BT * cpr;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = PLM_REV_BLOCK() * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t row = i / cpr;
    const int c = (int)(i - row * cpr);
    const int t = (int)(row % T);
    const int pair0 = (c * 8 % hd) / 2;
    uint16_t* p = qkv + row * ld + c * 8;
    const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(rcos + t * (hd / 2) + pair0);
    const f32x4_t sn = *reinterpret_cast<const f32x4_t*>(rsin + t * (hd / 2) + pair0);
    st_bf16x8(p, rope8(ld_bf16x8(p), cs, sn, sgn));  // sgn = -1: the inverse rotation (the generic backward path, attn_generic.hip)
  }
}

// =============================================================================================
// C ABI
// =============================================================================================
// causal batches (no document mask): attn_causal.hip
void plm_attn_fwd_causal(const uint16_t* qkv, uint16_t* out, float* lse, int64_t B, int64_t T, int64_t nh, hipStream_t s);
void plm_attn_bwd_causal(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* lse, float* delta, const float* rc,
                         const float* rs, uint16_t* dqkv, int64_t B, int64_t T, int64_t nh, hipStream_t s);
// document masks: attn_doc.hip
void plm_attn_doc_plan_launch(const int32_t* doc_start, int32_t* plan, int64_t B, int64_t T, int64_t nh, int split_min_q, hipStream_t s);
void plm_attn_doc_start_from_mask_launch(const uint8_t* mask, int32_t* doc_start, int32_t* status, int64_t B, int64_t T, hipStream_t s);
void plm_attn_fwd_doc(const uint16_t* qkv, const int32_t* doc_start, const int32_t* plan, uint16_t* out, float* lse, int64_t B, int64_t T,
                      int64_t nh, hipStream_t s);
void plm_attn_bwd_doc(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* lse, float* delta, const float* rc,
                      const float* rs, const int32_t* doc_start, const int32_t* plan, uint16_t* dqkv, int64_t B, int64_t T, int64_t nh,
                      hipStream_t s);

// head dims other than 64: attn_generic.hip
bool plm_attn_generic_supported(int64_t hd);
void plm_attn_fwd_generic(const uint16_t* qkv, const int32_t* doc_start, uint16_t* out, float* lse, int64_t B, int64_t T, int64_t nh, int64_t hd,
                          hipStream_t s);
void plm_attn_bwd_generic(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* lse, float* delta,
                          const int32_t* doc_start, uint16_t* dqkv, int64_t B, int64_t T, int64_t nh, int64_t hd, hipStream_t s);

static int check_attn_shape(const char* name, int64_t B, int64_t T, int64_t nh, int64_t hd) {
  PLM_REQUIRE(hd == HD || plm_attn_generic_supported(hd), "%s: head_dim %ld unsupported (64: the tuned kernels; 32, 128: the generic ones)", name, (long)hd);
  PLM_REQUIRE(B > 0 && T > 0 && nh > 0 && B < 65536 && nh < 65536 && T < (1 << 24), "%s: bad shape B=%ld T=%ld nh=%ld", name, (long)B,
              (long)T, (long)nh);
  PLM_REQUIRE(T % 4 == 0, "%s: T=%ld must be a multiple of 4", name, (long)T);
  return PLM_OK;
}

static void rope_qk_launch(uint16_t* qkv, const float* rope_cos, const float* rope_sin, int64_t B, int64_t T, int64_t nh, int64_t hd, float sgn,
                           hipStream_t s) {
  const int64_t items = B * T * (2 * nh * hd / 8);
  int64_t blocks = plm_cdiv(items, 256);
  if (blocks > ((int64_t)1 << 20)) blocks = (int64_t)1 << 20;  // one item per thread in memory order (see elementwise_grid)
  hipLaunchKernelGGL(rope_qk_kernel, dim3((unsigned)blocks), dim3(256), 0, s, qkv, rope_cos, rope_sin, B * T, (int)T, (int)nh, (int)hd, sgn);
}

extern "C" int plm_rope_qk(uint16_t* qkv, const float* rope_cos, const float* rope_sin, int64_t B, int64_t T, int64_t nh, int64_t hd,
                           void* stream) {
  PLM_REQUIRE(qkv && rope_cos && rope_sin, "plm_rope_qk: null pointer");
  if (int rc = check_attn_shape("plm_rope_qk", B, T, nh, hd)) return rc;
  rope_qk_launch(qkv, rope_cos, rope_sin, B, T, nh, hd, 1.f, (hipStream_t)stream);
  PLM_CHECK_LAUNCH("plm_rope_qk");
  return PLM_OK;
}

extern "C" int64_t plm_attn_doc_plan_bytes(int64_t B, int64_t T) { return (B > 0 && T > 0) ? 4 * doc_plan_ints(B, T) : 0; }

extern "C" int plm_attn_doc_plan(const int32_t* doc_start, int32_t* plan, int64_t B, int64_t T, int64_t nh, void* stream) {
  PLM_REQUIRE(doc_start && plan, "plm_attn_doc_plan: null pointer");
  PLM_REQUIRE((reinterpret_cast<uintptr_t>(plan) & 15) == 0, "plm_attn_doc_plan: plan must be 16-byte aligned");
  if (int rc = check_attn_shape("plm_attn_doc_plan", B, T, nh, HD)) return rc;
  plm_attn_doc_plan_launch(doc_start, plan, B, T, nh, plm_env().attn_doc_split_min, (hipStream_t)stream);
  PLM_CHECK_LAUNCH("plm_attn_doc_plan");
  return PLM_OK;
}

extern "C" int plm_attn_doc_start_from_mask(const uint8_t* mask, int32_t* doc_start, int32_t* status, int64_t B, int64_t T, void* stream) {
  PLM_REQUIRE(mask && doc_start && status, "plm_attn_doc_start_from_mask: null pointer");
  PLM_REQUIRE((reinterpret_cast<uintptr_t>(mask) & 3) == 0, "plm_attn_doc_start_from_mask: mask must be 4-byte aligned");
  if (int rc = check_attn_shape("plm_attn_doc_start_from_mask", B, T, 1, HD)) return rc;
  PLM_REQUIRE(B * T < ((int64_t)1 << 31), "plm_attn_doc_start_from_mask: B x T too large");
  plm_attn_doc_start_from_mask_launch(mask, doc_start, status, B, T, (hipStream_t)stream);
  PLM_CHECK_LAUNCH("plm_attn_doc_start_from_mask");
  return PLM_OK;
}

extern "C" int plm_attn_fwd(const uint16_t* qkv, const int32_t* doc_start, const int32_t* doc_plan, uint16_t* out, float* lse, int64_t B,
                            int64_t T, int64_t nh, int64_t hd, void* stream) {
  PLM_REQUIRE(qkv && out && lse, "plm_attn_fwd: null pointer");
  // LDS-DMA sources and the whole-row epilogue stores (RowStage::flush) are 16-byte accesses
  PLM_REQUIRE(((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, "plm_attn_fwd: qkv and out must be 16-byte aligned");
  if (int rc = check_attn_shape("plm_attn_fwd", B, T, nh, hd)) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (hd != HD) {
    plm_attn_fwd_generic(qkv, doc_start, out, lse, B, T, nh, hd, s);  // (no plan: element-wise masks)
  } else if (!doc_start) {
    plm_attn_fwd_causal(qkv, out, lse, B, T, nh, s);
  } else {
    PLM_REQUIRE(doc_plan, "plm_attn_fwd: a document mask needs its plan (plm_attn_doc_plan)");
    PLM_REQUIRE((reinterpret_cast<uintptr_t>(doc_plan) & 15) == 0, "plm_attn_fwd: doc_plan must be 16-byte aligned");
    plm_attn_fwd_doc(qkv, doc_start, doc_plan, out, lse, B, T, nh, s);
  }
  PLM_CHECK_LAUNCH("plm_attn_fwd");
  return PLM_OK;
}

extern "C" int plm_attn_bwd(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* lse, const float* rope_cos,
                            const float* rope_sin, const int32_t* doc_start, const int32_t* doc_plan, uint16_t* dqkv, float* delta,
                            int64_t B, int64_t T, int64_t nh, int64_t hd, void* stream) {
  PLM_REQUIRE(qkv && out && dout && lse && rope_cos && rope_sin && dqkv && delta, "plm_attn_bwd: null pointer");
  PLM_REQUIRE(((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(dqkv) |
                reinterpret_cast<uintptr_t>(rope_cos) | reinterpret_cast<uintptr_t>(rope_sin)) & 15) == 0,
              "plm_attn_bwd: qkv, out, dout, dqkv and the RoPE tables must be 16-byte aligned");
  if (int rc = check_attn_shape("plm_attn_bwd", B, T, nh, hd)) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (hd != HD) {
    // the generic kernels return the gradient w.r.t. the ROTATED q, k; the rotation is orthogonal, so its backward is the inverse rotation
    plm_attn_bwd_generic(qkv, out, dout, lse, delta, doc_start, dqkv, B, T, nh, hd, s);
    rope_qk_launch(dqkv, rope_cos, rope_sin, B, T, nh, hd, -1.f, s);
  } else if (!doc_start) {
    plm_attn_bwd_causal(qkv, out, dout, lse, delta, rope_cos, rope_sin, dqkv, B, T, nh, s);
  } else {
    PLM_REQUIRE(doc_plan, "plm_attn_bwd: a document mask needs its plan (plm_attn_doc_plan)");
    PLM_REQUIRE((reinterpret_cast<uintptr_t>(doc_plan) & 15) == 0, "plm_attn_bwd: doc_plan must be 16-byte aligned");
    plm_attn_bwd_doc(qkv, out, dout, lse, delta, rope_cos, rope_sin, doc_start, doc_plan, dqkv, B, T, nh, s);
  }
  PLM_CHECK_LAUNCH("plm_attn_bwd");
  return PLM_OK;
}
