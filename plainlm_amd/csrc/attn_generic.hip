// Attention for head dims OTHER than 64 (models/transformer.py:34 allows any dim // n_heads; every shipped config has 64, which the tuned
// kernels of attn_causal.hip / attn_doc.hip serve).  One plain flash-style kernel family, templated on HD in {32, 128}: 128-row query tiles
// (4 waves x 32 rows), 64-key K / V tiles copied to LDS by the whole workgroup (padded rows, two barriers per tile), S^T = K Q^T so a lane
// owns a query, element-wise causal / document masks (doc_start[B,T] or none), the exact running maximum, V^T / Q^T / dO^T operands through
// ds_read_b64_tr_b16.  Same numerics contract as the tuned family (fp32 statistics, bf16 P / dS, base-2 LSE), deterministic (no atomics), and
// correctness first: no LDS-DMA ring, no plan, no split items.  q and k arrive ROTATED (plm_rope_qk, the stand-alone pass, for these head dims);
// the backward kernels return the gradient w.r.t. the ROTATED q, k and plm_attn_bwd applies the inverse rotation in place afterwards
// (rope_qk_kernel with sign -1).
#include "plm_device.h"

#define GLOG2E 1.4426950408889634f

template <int HD>
struct GenTile {
  static constexpr int ROWB = HD * 2 + 16;  // bytes per tile row in LDS: 16 bytes of padding spread the rows over the banks
  static constexpr int BYTES = 64 * ROWB;
  // the whole workgroup (256 threads) copies rows row0 .. row0 + 63 (clamped to last_row) of a [*, ld] bf16 matrix at column col0
  static __device__ __forceinline__ void load(char* tile, const uint16_t* src, int64_t ld, int row0, int last_row, int tid) {
    constexpr int CPR = HD / 8;  // 16-byte chunks per row
    for (int c = tid; c < 64 * CPR; c += 256) {
      const int r = c / CPR, k = c - r * CPR;
      *reinterpret_cast<bf16x8_t*>(tile + r * ROWB + k * 16) = ld_bf16x8(src + (int64_t)min(row0 + r, last_row) * ld + k * 8);
    }
  }
  // A operand, i = tile row (lane & 31), k = head dims ks*16 + hi*8 .. + 7
  static __device__ __forceinline__ bf16x8_t rows(const char* tile, int row, int ks, int hi) {
    return *reinterpret_cast<const bf16x8_t*>(tile + row * ROWB + (ks * 16 + hi * 8) * 2);
  }
  // A operand, i = head dim db*32 + (lane & 31), k-slot e of lane half hi = tile row rbase + (e & 3) + 8 (e >> 2)  (see frag_cols, attn_common.h)
  static __device__ __forceinline__ bf16x8_t cols(const char* tile, int db, int rbase, int lane) {
    const int ib = (lane >> 4) & 1, t16 = lane & 15;
    const int col = db * 32 + ib * 16 + (t16 & 3) * 4;
    const int row = rbase + (t16 >> 2);
    return join_tr(lds_read_tr16(tile + row * ROWB + col * 2), lds_read_tr16(tile + (row + 8) * ROWB + col * 2));
  }
};

__device__ __forceinline__ void gzero16(f32x16_t& v) {
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = 0.f;
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256, 2) void attn_fwd_generic_kernel(const uint16_t* __restrict__ qkv, const int32_t* __restrict__ doc_start,
                                                                  uint16_t* __restrict__ out, float* __restrict__ lse, int T, int nh,
                                                                  float scale) {
  using TL = GenTile<HD>;
  __shared__ __attribute__((aligned(16))) char smem[2 * TL::BYTES];  // K | V
  constexpr int NKS = HD / 16, NDB = HD / 32;
  const int ntile = (T + 127) / 128;
  const int tile = ntile - 1 - (int)(blockIdx.x / (gridDim.x / ntile));  // heaviest (latest) query tiles first
  const int bh = blockIdx.x % (gridDim.x / ntile), h = bh % nh, b = bh / nh;
  const int dm = nh * HD, ld = 3 * dm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int q0 = tile * 128, qrow = q0 + wave * 32 + l31;
  const bool qvalid = qrow < T;
  const uint16_t* base = qkv + (int64_t)b * T * ld + h * HD;
  const float c2 = scale * GLOG2E;
  bf16x8_t qf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) qf[ks] = qvalid ? ld_bf16x8(base + (int64_t)qrow * ld + ks * 16 + hi * 8) : zero_bf16x8();
  const int dsq = (doc_start && qvalid) ? doc_start[(int64_t)b * T + qrow] : 0;
  f32x16_t o[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db) gzero16(o[db]);
  float m = -INFINITY, lsum = 0.f;
  const int jt_hi = (min(T, q0 + 128) + 63) / 64;
  const int jt_lo = doc_start ? doc_start[(int64_t)b * T + q0] / 64 : 0;  // doc_start is non-decreasing: tiles before the first row's document are invisible to all
  char* sK = smem;
  char* sV = smem + TL::BYTES;
  for (int jt = jt_lo; jt < jt_hi; ++jt) {
    const int kv0 = jt * 64;
    __syncthreads();  // everyone is done with the previous tile
    TL::load(sK, base + dm, ld, kv0, T - 1, tid);
    TL::load(sV, base + 2 * dm, ld, kv0, T - 1, tid);
    __syncthreads();
    if (kv0 > q0 + wave * 32 + 31) continue;  // entirely above this wave's diagonal (wave-uniform; barriers are outside the skipped part)
    f32x16_t s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      gzero16(s[kb]);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) s[kb] = mfma32(TL::rows(sK, kb * 32 + l31, ks, hi), qf[ks], s[kb]);
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kvg = kv0 + kb * 32 + mfma32_row(r, hi);
        if (!(kvg <= qrow && kvg >= dsq)) s[kb][r] = -INFINITY;
        tmax = fmaxf(tmax, s[kb][r]);
      }
    {
      float t_lo, t_hi;
      half_pair(tmax, t_lo, t_hi);
      tmax = fmaxf(t_lo, t_hi);
    }
    const float m_new = fmaxf(m, tmax * c2);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m - m_safe);  // m = -inf: 0
    float psum = 0.f;
    bf16x8_t pf[4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], c2, -m_safe));
        psum += p;
        pf[kb * 2 + (r >> 3)][r & 7] = f2bf(p);
      }
    lsum = lsum * alpha + psum;
    m = m_new;
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) o[db] = mfma32(TL::cols(sV, db, (sp >> 1) * 32 + (sp & 1) * 16 + 4 * hi, lane), pf[sp], o[db]);
    }
  }
  float l_lo, l_hi;
  half_pair(lsum, l_lo, l_hi);
  const float ltot = l_lo + l_hi, inv = 1.f / ltot;
  if (qvalid) {
    if (hi == 0) lse[((int64_t)b * nh + h) * T + qrow] = m + __builtin_amdgcn_logf(ltot);  // base-2 LSE
    uint16_t* orow = out + ((int64_t)b * T + qrow) * dm + h * HD;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // O^T[d][q]: this lane holds head dims db*32 + 8g + 4hi .. + 3 of its query
        bf16x4_t v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = f2bf(o[db][4 * g + e] * inv);
        st_bf16x4(orow + db * 32 + 8 * g + 4 * hi, v);
      }
  }
}

// ---------------------------------------------------------------------------------------------
// backward dQ (w.r.t. the ROTATED q); publishes delta[q] = sum_d dO O
// ---------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_generic_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out,
                                                                     const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                                                                     float* __restrict__ delta, const int32_t* __restrict__ doc_start,
                                                                     uint16_t* __restrict__ dqkv, int T, int nh, float scale) {
  using TL = GenTile<HD>;
  __shared__ __attribute__((aligned(16))) char smem[2 * TL::BYTES];
  constexpr int NKS = HD / 16, NDB = HD / 32;
  const int ntile = (T + 127) / 128;
  const int tile = ntile - 1 - (int)(blockIdx.x / (gridDim.x / ntile));
  const int bh = blockIdx.x % (gridDim.x / ntile), h = bh % nh, b = bh / nh;
  const int dm = nh * HD, ld = 3 * dm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int q0 = tile * 128, qrow = q0 + wave * 32 + l31;
  const bool qvalid = qrow < T;
  const uint16_t* base = qkv + (int64_t)b * T * ld + h * HD;
  const float c2 = scale * GLOG2E;
  bf16x8_t qf[NKS], dof[NKS];
  float part = 0.f;
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int d0 = ks * 16 + hi * 8;
    qf[ks] = qvalid ? ld_bf16x8(base + (int64_t)qrow * ld + d0) : zero_bf16x8();
    dof[ks] = qvalid ? ld_bf16x8(dout + ((int64_t)b * T + qrow) * dm + h * HD + d0) : zero_bf16x8();
    if (qvalid) {
      const bf16x8_t o8 = ld_bf16x8(out + ((int64_t)b * T + qrow) * dm + h * HD + d0);
#pragma unroll
      for (int e = 0; e < 8; ++e) part += bf2f(o8[e]) * bf2f(dof[ks][e]);
    }
  }
  float d_lo, d_hi;
  half_pair(part, d_lo, d_hi);
  const float Dq = d_lo + d_hi;
  const float Lq = qvalid ? lse[((int64_t)b * nh + h) * T + qrow] : 0.f;
  const int dsq = (doc_start && qvalid) ? doc_start[(int64_t)b * T + qrow] : 0;
  if (qvalid && hi == 0) delta[((int64_t)b * nh + h) * T + qrow] = Dq;
  f32x16_t dq[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db) gzero16(dq[db]);
  const int jt_hi = (min(T, q0 + 128) + 63) / 64;
  const int jt_lo = doc_start ? doc_start[(int64_t)b * T + q0] / 64 : 0;
  char* sK = smem;
  char* sV = smem + TL::BYTES;
  for (int jt = jt_lo; jt < jt_hi; ++jt) {
    const int kv0 = jt * 64;
    __syncthreads();
    TL::load(sK, base + dm, ld, kv0, T - 1, tid);
    TL::load(sV, base + 2 * dm, ld, kv0, T - 1, tid);
    __syncthreads();
    if (kv0 > q0 + wave * 32 + 31) continue;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16_t s, dp;
      gzero16(s);
      gzero16(dp);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        s = mfma32(TL::rows(sK, kb * 32 + l31, ks, hi), qf[ks], s);      // S^T[kv][q]
        dp = mfma32(TL::rows(sV, kb * 32 + l31, ks, hi), dof[ks], dp);   // dP^T[kv][q]
      }
      bf16x8_t dsf[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kvg = kv0 + kb * 32 + mfma32_row(r, hi);
        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c2, -Lq));
        p = (kvg <= qrow && kvg >= dsq) ? p : 0.f;
        dsf[r >> 3][r & 7] = f2bf(p * (dp[r] - Dq));
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int db = 0; db < NDB; ++db) dq[db] = mfma32(TL::cols(sK, db, kb * 32 + s2 * 16 + 4 * hi, lane), dsf[s2], dq[db]);  // dQ^T[d][q]
    }
  }
  if (qvalid) {
    uint16_t* orow = dqkv + ((int64_t)b * T + qrow) * ld + h * HD;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4_t v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = f2bf(dq[db][4 * g + e] * scale);
        st_bf16x4(orow + db * 32 + 8 * g + 4 * hi, v);
      }
  }
}

// ---------------------------------------------------------------------------------------------
// backward dK / dV (dK w.r.t. the ROTATED k): one workgroup per 128 keys (4 waves x 32 keys in registers), loops over 64-query tiles
// ---------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkdv_generic_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout,
                                                                       const float* __restrict__ lse, const float* __restrict__ delta,
                                                                       const int32_t* __restrict__ doc_start, uint16_t* __restrict__ dqkv,
                                                                       int T, int nh, float scale) {
  using TL = GenTile<HD>;
  __shared__ __attribute__((aligned(16))) char smem[2 * TL::BYTES + 3 * 64 * 4];  // Q | dO | lse[64] | delta[64] | doc_start[64]
  constexpr int NKS = HD / 16, NDB = HD / 32;
  const int ntile = (T + 127) / 128;
  const int kt = (int)(blockIdx.x / (gridDim.x / ntile));  // key tile 0 meets every query tile: heaviest first
  const int bh = blockIdx.x % (gridDim.x / ntile), h = bh % nh, b = bh / nh;
  const int dm = nh * HD, ld = 3 * dm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int kv0 = kt * 128, kvrow = kv0 + wave * 32 + l31;
  const bool kvalid = kvrow < T;
  const uint16_t* base = qkv + (int64_t)b * T * ld + h * HD;
  const uint16_t* dobase = dout + (int64_t)b * T * dm + h * HD;
  const float c2 = scale * GLOG2E;
  bf16x8_t kf[NKS], vf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const uint16_t* p = base + (int64_t)kvrow * ld + ks * 16 + hi * 8;
    kf[ks] = kvalid ? ld_bf16x8(p + dm) : zero_bf16x8();
    vf[ks] = kvalid ? ld_bf16x8(p + 2 * dm) : zero_bf16x8();
  }
  f32x16_t dk[NDB], dv[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db) {
    gzero16(dk[db]);
    gzero16(dv[db]);
  }
  char* sQ = smem;
  char* sDO = smem + TL::BYTES;
  float* sL = reinterpret_cast<float*>(smem + 2 * TL::BYTES);
  float* sD = sL + 64;
  int* sDS = reinterpret_cast<int*>(sD + 64);
  const int nqt = (T + 63) / 64;
  for (int jq = kv0 / 64; jq < nqt; ++jq) {
    const int qt0 = jq * 64;
    // doc_start is non-decreasing: once a query tile's first row starts behind this key tile, every later one does
    if (doc_start && doc_start[(int64_t)b * T + qt0] > kv0 + 127) break;  // (uniform over the workgroup)
    __syncthreads();
    TL::load(sQ, base, ld, qt0, T - 1, tid);
    TL::load(sDO, dobase, dm, qt0, T - 1, tid);
    if (tid < 64) {
      const int q = min(qt0 + tid, T - 1);
      sL[tid] = lse[((int64_t)b * nh + h) * T + q];
      sD[tid] = delta[((int64_t)b * nh + h) * T + q];
      sDS[tid] = doc_start ? doc_start[(int64_t)b * T + q] : 0;
    }
    __syncthreads();
    if (qt0 + 63 < kv0 + wave * 32) continue;  // every query precedes this wave's first key
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16_t s, dp;
      gzero16(s);
      gzero16(dp);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        s = mfma32(TL::rows(sQ, qb * 32 + l31, ks, hi), kf[ks], s);       // S[q][kv]
        dp = mfma32(TL::rows(sDO, qb * 32 + l31, ks, hi), vf[ks], dp);    // dP[q][kv]
      }
      bf16x8_t pf[2], dsf[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = qb * 32 + mfma32_row(r, hi), qg = qt0 + ql;
        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c2, -sL[ql]));
        p = (kvrow <= qg && qg < T && kvrow >= sDS[ql]) ? p : 0.f;
        pf[r >> 3][r & 7] = f2bf(p);
        dsf[r >> 3][r & 7] = f2bf(p * (dp[r] - sD[ql]));
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
          dv[db] = mfma32(TL::cols(sDO, db, qb * 32 + s2 * 16 + 4 * hi, lane), pf[s2], dv[db]);   // dV^T[d][kv]
          dk[db] = mfma32(TL::cols(sQ, db, qb * 32 + s2 * 16 + 4 * hi, lane), dsf[s2], dk[db]);   // dK^T[d][kv]
        }
    }
  }
  if (kvalid) {
    uint16_t* krow = dqkv + ((int64_t)b * T + kvrow) * ld + dm + h * HD;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4_t a, c;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[e] = f2bf(dk[db][4 * g + e] * scale);
          c[e] = f2bf(dv[db][4 * g + e]);
        }
        st_bf16x4(krow + db * 32 + 8 * g + 4 * hi, a);
        st_bf16x4(krow + dm + db * 32 + 8 * g + 4 * hi, c);
      }
  }
}

// ---------------------------------------------------------------------------------------------
// launchers (called from the C ABI entry points in attn.hip when hd != 64)
// ---------------------------------------------------------------------------------------------
bool plm_attn_generic_supported(int64_t hd) { return hd == 32 || hd == 128; }

void plm_attn_fwd_generic(const uint16_t* qkv, const int32_t* doc_start, uint16_t* out, float* lse, int64_t B, int64_t T, int64_t nh, int64_t hd,
                          hipStream_t s) {
  const dim3 grid((unsigned)(plm_cdiv(T, 128) * nh * B)), block(256);
  const float scale = 1.f / sqrtf((float)hd);
  if (hd == 32) hipLaunchKernelGGL(attn_fwd_generic_kernel<32>, grid, block, 0, s, qkv, doc_start, out, lse, (int)T, (int)nh, scale);
  else hipLaunchKernelGGL(attn_fwd_generic_kernel<128>, grid, block, 0, s, qkv, doc_start, out, lse, (int)T, (int)nh, scale);
}

void plm_attn_bwd_generic(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* lse, float* delta,
                          const int32_t* doc_start, uint16_t* dqkv, int64_t B, int64_t T, int64_t nh, int64_t hd, hipStream_t s) {
  const dim3 grid((unsigned)(plm_cdiv(T, 128) * nh * B)), block(256);
  const float scale = 1.f / sqrtf((float)hd);
  if (hd == 32) {
    hipLaunchKernelGGL(attn_bwd_dq_generic_kernel<32>, grid, block, 0, s, qkv, out, dout, lse, delta, doc_start, dqkv, (int)T, (int)nh, scale);
    hipLaunchKernelGGL(attn_bwd_dkdv_generic_kernel<32>, grid, block, 0, s, qkv, dout, lse, delta, doc_start, dqkv, (int)T, (int)nh, scale);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_generic_kernel<128>, grid, block, 0, s, qkv, out, dout, lse, delta, doc_start, dqkv, (int)T, (int)nh, scale);
    hipLaunchKernelGGL(attn_bwd_dkdv_generic_kernel<128>, grid, block, 0, s, qkv, dout, lse, delta, doc_start, dqkv, (int)T, (int)nh, scale);
  }
}
