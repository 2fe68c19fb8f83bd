// Shared device/host helpers for the gfx950 kernels of libplainlm_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/plainlm_hip.h"

// ---------------------------------------------------------------------------
// host-side error plumbing
// ---------------------------------------------------------------------------
void plm_set_error(const char* fmt, ...);

#define PLM_REQUIRE(cond, ...)       \
  do {                               \
    if (!(cond)) {                   \
      plm_set_error(__VA_ARGS__);    \
      return PLM_E_INVALID;          \
    }                                \
  } while (0)

#define PLM_CHECK_LAUNCH(name)                                                   \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      plm_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
      return PLM_E_HIP;                                                          \
    }                                                                            \
  } while (0)

static inline int64_t plm_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Environment switches of the library (tests / A-B runs), read ONCE at the first call and again only on plm_reload_env():
// no getenv on any launch path.
struct PlmEnv {
  bool gemm_v1;               // PLM_GEMM_V1: register-staged 128x128 GEMMs everywhere (the fused entry points take their two-launch paths)
  bool tn_no_big;             // PLM_TN_NO_BIG: no persistent 256x256 TN kernel
  bool nt_no_hybrid;          // PLM_NT_NO_HYBRID: no whole-K + stream-K NT schedule
  long long nt_hybrid_min_k;  // PLM_NT_HYBRID_MIN_K: lowers the hybrid schedule's thresholds (-1: defaults)
  int attn_doc_split_min;     // PLM_ATTN_DOC_SPLIT_MIN: tile steps from which a heavy document-mask QUERY tile is split into two 64-row items (default 8; 0: never)
};
const PlmEnv& plm_env();

// ---------------------------------------------------------------------------
// device-side types
// ---------------------------------------------------------------------------
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#define PLM_WAVE 64

// Row order of the HBM-bound kernels: DESCENDING.  The persistent GEMMs walk their tiles in ascending row order, so the rows they
// wrote last (the ones still in the 256 MB Infinity Cache) are the highest; a consumer that starts there, and that leaves ITS
// newest output at the low rows where the next GEMM starts, turns the cache into a LIFO between producer and consumer
// (run 36: swiglu_fwd behind the fc1 GEMM 83 -> 70 us in the step; isolated timings do not change).  -DPLM_EW_FORWARD restores
// ascending order for A/B builds.
#ifdef PLM_EW_FORWARD
#define PLM_REV_BLOCK() ((int64_t)blockIdx.x)
#define PLM_REV_ROW(row, M) (row)
#else
#define PLM_REV_BLOCK() ((int64_t)(gridDim.x - 1 - blockIdx.x))
#define PLM_REV_ROW(row, M) ((M) - 1 - (row))
#endif

__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }  // v_cvt_pk_bf16_f32, RNE

// 16-byte global load / store of 8 bf16
__device__ __forceinline__ bf16x8_t ld_bf16x8(const uint16_t* p) { return *reinterpret_cast<const bf16x8_t*>(p); }
__device__ __forceinline__ void st_bf16x8(uint16_t* p, bf16x8_t v) { *reinterpret_cast<bf16x8_t*>(p) = v; }
// Store of a GEMM output row segment.  PLM_C_STORE_MODE: 0 plain | 1 `sc1` (the line does not stay in the XCD's L2:
// MI355X_MICROARCH.md, stores of each flavour) | 2 `nt` (default).  Same-box A/Bs of the whole step (profiles/r04_ab_stores.txt):
// sc1 -0.6 %, nt +0.6 % (the outputs are 50 MB - 3.3 GB streams that only push operand panels out of the 4 MB L2), nt on the fused
// epilogues' extra outputs as well (st_c2_bf16x8) -0.5 % against nt on C alone.
#ifndef PLM_C_STORE_MODE
#define PLM_C_STORE_MODE 2
#endif
__device__ __forceinline__ void st_c_bf16x8(uint16_t* p, bf16x8_t v) {
#if PLM_C_STORE_MODE == 1
  // an asm dwordx4 store ends with s_nop 1 INSIDE the string: hipcc pads nothing around an asm statement, and its next instruction
  // may overwrite the data registers before the store has read them (cdna_hip_programming.md section 5.7; the first build of the nt
  // variant lacked it and stored garbage in the 256x192 kernel - caught by test_gemm_nt_variants)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#elif PLM_C_STORE_MODE == 2
  __builtin_nontemporal_store(v, reinterpret_cast<bf16x8_t*>(p));  // global_store_dwordx4 ... nt, counted and padded by hipcc
#else
  *reinterpret_cast<bf16x8_t*>(p) = v;
#endif
}
// the fused epilogues' extra outputs (d(gate) | d(up), rotated q | k, the SwiGLU activation): PLM_C_STORE_ALL = 1 gives them the mode above
#ifndef PLM_C_STORE_ALL
#define PLM_C_STORE_ALL 0
#endif
__device__ __forceinline__ void st_c2_bf16x8(uint16_t* p, bf16x8_t v) {
#if PLM_C_STORE_ALL
  st_c_bf16x8(p, v);
#else
  *reinterpret_cast<bf16x8_t*>(p) = v;
#endif
}
__device__ __forceinline__ bf16x4_t ld_bf16x4(const uint16_t* p) { return *reinterpret_cast<const bf16x4_t*>(p); }
__device__ __forceinline__ void st_bf16x4(uint16_t* p, bf16x4_t v) { *reinterpret_cast<bf16x4_t*>(p) = v; }

__device__ __forceinline__ bf16x8_t zero_bf16x8() {
  u32x4_t z = {0u, 0u, 0u, 0u};
  return __builtin_bit_cast(bf16x8_t, z);
}

// Two bf16 packed in one dword -> two floats with ONE VALU op each (shift / mask), and back with one
// v_cvt_pk_bf16_f32 per pair.  Element-wise access to bf16 ext-vectors makes hipcc shuffle lanes with v_perm_b32.
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  bf16x2_t t;
  t[0] = (bf16_t)lo;
  t[1] = (bf16_t)hi;
  return __builtin_bit_cast(unsigned, t);
}

// full-wave (64 lanes) reductions; every lane receives the result
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// v_mfma_f32_32x32x16_bf16:  D[i][j] += sum_k A[i][k] B[k][j]
//   a: lane l holds A[i = l&31][k-slots (l>>5)*8 + 0..7]
//   b: lane l holds B[k-slots (l>>5)*8 + 0..7][j = l&31]
//   d: lane l reg r holds D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31]
__device__ __forceinline__ f32x16_t mfma32(bf16x8_t a, bf16x8_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// sigmoid by v_exp_f32 + v_rcp_f32 (1 ulp each): every user rounds the result to bf16, a full-precision division buys nothing.
// Shared by the stand-alone SwiGLU kernels and the fc1 GEMM epilogue so that both produce the same bits.
__device__ __forceinline__ float plm_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
// out = bf16(bf16(silu(x)) * z) for bf16 x, z: the reference's autocast rounding order (models/components.py:55-56)
__device__ __forceinline__ bf16_t plm_swiglu_bf16(bf16_t x, bf16_t z) {
  const float xf = bf2f(x);
  const bf16_t s = f2bf(xf * plm_sigmoid(xf));
  return f2bf(bf2f(s) * bf2f(z));
}

// RoPE on four interleaved pairs (models/embeddings.py:15-30): (a, b) -> (a cos - b sin, b cos + a sin), fp32 math, bf16 result;
// sgn = -1 is the inverse rotation.  Shared by rope_qk_kernel, the attention backward epilogues and the w_qkv GEMM epilogue.
__device__ __forceinline__ bf16x8_t rope8(bf16x8_t v, f32x4_t c, f32x4_t s, float sgn) {
  bf16x8_t o;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a = bf2f(v[2 * p]), b = bf2f(v[2 * p + 1]);
    const float sn = s[p] * sgn;
    o[2 * p] = f2bf(a * c[p] - b * sn);
    o[2 * p + 1] = f2bf(b * c[p] + a * sn);
  }
  return o;
}

// v_mfma_f32_16x16x32_bf16:  D[i][j] += sum_k A[i][k] B[k][j]   (same flops per cycle; a quarter of the accumulator registers
// per instruction - measured 7-12 % more flops at the board's power limit on random operands, tools/ubench/mfma_power.hip)
//   a: lane l holds A[i = l&15][k-slots (l>>4)*8 + 0..7]
//   b: lane l holds B[k-slots (l>>4)*8 + 0..7][j = l&15]
//   d: lane l reg r holds D[i = 4*(l>>4) + r][j = l&15]
__device__ __forceinline__ f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// row (i) index of accumulator register r for lane-half hi in a 32x32 tile
__device__ __forceinline__ int mfma32_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ds_read_b64_tr_b16: within each 16-lane group, lane t supplies the address of
// 4 consecutive bf16 of row (t>>2), column chunk (t&3) of a [4 rows][16 cols] block
// (any row stride) and receives column t of that block: out[j] = block[row j][col t].
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
__device__ __forceinline__ s16x4_t lds_read_tr16(const void* lds_ptr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(lds_ptr));
}
__device__ __forceinline__ bf16x8_t join_tr(s16x4_t lo, s16x4_t hi) {
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// LDS-DMA (global_load_lds_dwordx4) issued through inline asm: 16 bytes per lane from `gsrc` land at
// lds_wave_base + lane*16 (wave-uniform base).  Unlike the builtin, hipcc does not see an LDS write here, so it does
// not put a draining s_waitcnt vmcnt(0) in front of later LDS reads it cannot disambiguate (it did so before every
// ds_read_b64_tr_b16).  The caller owns the ordering: counted s_waitcnt vmcnt(N), then a barrier, then the reads.
// No VGPR destination, so the statement is register-safe; M0 is saved and restored inside the statement.
__device__ __forceinline__ unsigned lds_addr_u32(const void* p) {
  return (unsigned)(uintptr_t)((const __attribute__((address_space(3))) void*)p);
}
__device__ __forceinline__ void dma16_asm(const void* gsrc, const void* lds_wave_base) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr_u32(lds_wave_base));
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}

// Same LDS-DMA with the address split into a wave-uniform 64-bit base (SGPR pair) and a per-lane unsigned 32-bit byte
// offset: no 64-bit VALU arithmetic per instruction when only the base moves between issues.
__device__ __forceinline__ void dma16_saddr_asm(const void* uniform_base, unsigned lane_byte_off, const void* lds_wave_base) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr_u32(lds_wave_base));
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(lane_byte_off), "s"(uniform_base), "s"(dst)
               : "memory");
}

// The same with the LDS destination as a byte address (wave-uniform integer): no generic -> LDS pointer cast per issue (hipcc turns
// that cast into a compare against the shared aperture, and mis-selects it when it believes the pointer divergent).
__device__ __forceinline__ void dma16_saddr_u32(const void* uniform_base, unsigned lane_byte_off, unsigned lds_wave_addr) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_wave_addr);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(lane_byte_off), "s"(uniform_base), "s"(dst)
               : "memory");
}

// Values of lane l and lane l^32 as a pair (lower-half value, upper-half value) in every lane, by one
// v_permlane32_swap (gfx950): a cross-half reduction without the LDS round trip of ds_bpermute.
__device__ __forceinline__ void half_pair(float v, float& lo, float& hi) {
  // inline asm: hipcc's __builtin_amdgcn_permlane32_swap folds its two results into one register when both
  // operands carry the same value (observed: v_max_f32 v35, v35, v35 after the swap)
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  lo = a;  // {v[0..31], v[0..31]}
  hi = b;  // {v[32..63], v[32..63]}
}

// 4-byte global store to (wave-uniform 64-bit base in SGPRs) + (per-lane unsigned 32-bit byte offset): no address VGPRs beyond
// the one lane offset, however many different bases a fully unrolled epilogue uses.
__device__ __forceinline__ void st_f32_saddr(float* uniform_base, unsigned lane_byte_off, float v) {
  asm volatile("global_store_dword %0, %1, %2" ::"v"(lane_byte_off), "v"(v), "s"(uniform_base) : "memory");
}

// XCD-aware bijective remap of a linear workgroup id: consecutive ids on one XCD
// (hardware places workgroup b on XCD b % 8) so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}
