// Error plumbing, version, and the hardware-semantics probes used by the GPU tests.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "plm_device.h"

static thread_local char g_err[512] = "";

void plm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* plm_last_error_string(void) { return g_err; }
extern "C" int plm_version(void) { return 107; }  // 107: round 6 (document-mask plan: plm_attn_doc_plan, plm_attn_fwd / plm_attn_bwd take doc_plan); 106: round 5 (NT variant 7 re-purposed: the 128 x 192 tile; the PLM_NT_DUO / PLM_DUO_* switches removed); 105: round 4 (+ plm_adamw_cast_multi); 104: round 3 (+ plm_reload_env); 103 / 102: round 2 (see include/plainlm_hip.h); 101: round 1

static PlmEnv g_env;
static void load_env() {
  auto num = [](const char* name) -> long long {
    const char* e = getenv(name);
    return e ? atoll(e) : -1;
  };
  g_env.gemm_v1 = getenv("PLM_GEMM_V1") != nullptr;
  g_env.tn_no_big = getenv("PLM_TN_NO_BIG") != nullptr;
  g_env.nt_no_hybrid = getenv("PLM_NT_NO_HYBRID") != nullptr;
  g_env.nt_hybrid_min_k = num("PLM_NT_HYBRID_MIN_K");
  g_env.attn_doc_split_min = num("PLM_ATTN_DOC_SPLIT_MIN") >= 0 ? (int)num("PLM_ATTN_DOC_SPLIT_MIN") : 8;
}
const PlmEnv& plm_env() {
  static const bool once = (load_env(), true);
  (void)once;
  return g_env;
}
extern "C" void plm_reload_env(void) {
  (void)plm_env();
  load_env();
}

// ---------------------------------------------------------------------------
// probe: what does ds_read_b64_tr_b16 deliver?  LDS holds 0,1,2,...; lane l reads at byte 8*l.
// ---------------------------------------------------------------------------
__global__ void probe_tr16_kernel(int32_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[256];
  const int l = threadIdx.x;
  for (int i = l; i < 256; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const s16x4_t v = lds_read_tr16(reinterpret_cast<const char*>(lds) + 8 * l);
#pragma unroll
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (int32_t)(uint16_t)v[j];
}

extern "C" int plm_probe_ds_read_tr16(int32_t* out, void* stream) {
  PLM_REQUIRE(out, "plm_probe_ds_read_tr16: null pointer");
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
  PLM_CHECK_LAUNCH("plm_probe_ds_read_tr16");
  return PLM_OK;
}

// ---------------------------------------------------------------------------
// probe: operand / accumulator layout of v_mfma_f32_32x32x16_bf16 as mfma32() documents it
// ---------------------------------------------------------------------------
__global__ void probe_mfma32_kernel(const float* A, const float* B, float* out) {
  const int l = threadIdx.x, l31 = l & 31, hi = l >> 5;
  bf16x8_t a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a[e] = f2bf(A[l31 * 16 + hi * 8 + e]);    // A[i][k], row-major [32][16]
    b[e] = f2bf(B[(hi * 8 + e) * 32 + l31]);  // B[k][j], row-major [16][32]
  }
  f32x16_t c;
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = mfma32(a, b, c);
#pragma unroll
  for (int r = 0; r < 16; ++r) out[mfma32_row(r, hi) * 32 + l31] = c[r];
}

extern "C" int plm_probe_mfma32(const float* A, const float* B, float* out, void* stream) {
  PLM_REQUIRE(A && B && out, "plm_probe_mfma32: null pointer");
  hipLaunchKernelGGL(probe_mfma32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, out);
  PLM_CHECK_LAUNCH("plm_probe_mfma32");
  return PLM_OK;
}
