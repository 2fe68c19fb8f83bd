// Flash-style attention for gfx950 with RoPE applied in-kernel: the C ABI, the stand-alone RoPE pass, and the kernels that serve
// batches WITH document masks (causal batches without a mask take the kernels of attn_causal.hip).
// Replaces models/transformer.py:43-65 (split, RoPE, transposes, SDPA, transpose back) and models/embeddings.py:15-30; the mask of
// data/datasets/data_prep_utils.py:7-23 is expressed as doc_start[B,T] (query i sees key j iff doc_start[i] <= j <= i; doc_start is
// non-decreasing in i).
//
// q, k, v are read strided straight out of the w_qkv output [B*T, 3*nh*64]; no transposed copies.  All matrix products are
// v_mfma_f32_32x32x16_bf16.  Scores are computed TRANSPOSED (S^T[kv][q] = K Q^T) so a lane owns one query column: row max / row sum are
// in-lane plus one cross-half shuffle, and the probabilities are already in the B-operand layout of the following P V product (no LDS
// round trip for P).  The V / K / Q / dO operands whose contraction index is the token row are fetched with ds_read_b64_tr_b16 (hardware
// transpose) from the same LDS image that serves the ds_read_b128 operands.
//
// q and k arrive ROTATED (the w_qkv GEMM's epilogue applies RoPE, plm_qkv_rope_bf16; rope_qk_kernel below is that entry point's fallback
// and the tests' yardstick), so no inner loop rotates anything; K / V (fwd, dQ) and Q / dO (dK/dV) tiles are staged global -> LDS by LDS-DMA
// (no VGPR round trip, no ds_write); the backward kernels apply the inverse rotation to dQ / dK in their epilogues, so dqkv is the gradient
// w.r.t. the PRE-rotation projection.
//
// Kernels for document masks: 128-row tiles (4 waves x 32 rows, up to 4 workgroups per CU), 64-row K / V (Q / dO) tiles through two LDS
// stages; tiles that lie entirely before the first document of a workgroup's rows (or after its last) are skipped, every processed tile
// takes the masked softmax.  With documents of a few hundred tokens most tiles are skipped, which is why these kernels - not the
// 256-row causal ones - serve this case (profiles/r03_attn_ablation.txt: 62 vs 86 us forward at mean document length 256).
#include "plm_device.h"

#include "attn_common.h"

// =============================================================================================
// RoPE on the q and k column blocks of the w_qkv output, in place (models/embeddings.py:15-30):
// interleaved pairs (x[2i], x[2i+1]) -> (a cos - b sin, b cos + a sin), fp32 math, bf16 result.
// The fallback of plm_qkv_rope_bf16 (shapes its fused epilogue does not take) and the yardstick of the epilogue's bit-equality tests.
// =============================================================================================
__global__ __launch_bounds__(256) void rope_qk_kernel(uint16_t* __restrict__ qkv, const float* __restrict__ rcos,
                                                      const float* __restrict__ rsin, int64_t BT, int T, int nh) {
  const int dm = nh * HD, ld = 3 * dm;
  const int cpr = 2 * dm / 8;  // 16-byte chunks of q|k per token row
  const int64_t total = BT * cpr;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = PLM_REV_BLOCK() * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t row = i / cpr;
    const int c = (int)(i - row * cpr);
    const int t = (int)(row % T);
    const int pair0 = (c * 8 % HD) / 2;
    uint16_t* p = qkv + row * ld + c * 8;
    const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(rcos + t * (HD / 2) + pair0);
    const f32x4_t sn = *reinterpret_cast<const f32x4_t*>(rsin + t * (HD / 2) + pair0);
    st_bf16x8(p, rope8(ld_bf16x8(p), cs, sn, 1.f));
  }
}

// =============================================================================================
// forward with document masks (q, k already rotated)
// =============================================================================================
// Block -> (128-row tile, head, batch).  A causal tile's work grows linearly with its index (2 .. 2*T/128 key tiles), and the
// hardware hands blocks out in blockIdx order, so the order is tile-major: ALL blocks of the heaviest tile index first,
// the lightest last (longest-processing-time-first; with the (tile, h, b) 3-D grid every (h, b) group ended on its own
// light tiles but the last groups' heavy blocks ran on into a ~40 % tail at falling occupancy - 2.1 of 5 possible waves
// per SIMD on average, run 30 counters).
__device__ __forceinline__ void attn_block(int T, int nh, int& tile, int& h, int& b) {
  const int ntile = (T + 127) / 128;
  const int nbh = gridDim.x / ntile;
  const int bh = blockIdx.x % nbh;
  tile = blockIdx.x / nbh;
  h = bh % nh;
  b = bh / nh;
}

__global__ __launch_bounds__(256, 4) void attn_fwd_doc_kernel(const uint16_t* __restrict__ qkv, const int32_t* __restrict__ doc_start,
                                                          uint16_t* __restrict__ out, float* __restrict__ lse, int T, int nh) {
  constexpr int KT = 64;            // kv rows per tile
  constexpr int TILE = KT * 128;    // 8 KiB
  __shared__ __attribute__((aligned(1024))) char smem[2 * 2 * TILE];  // [stage][K|V]

  int tile_, h, b;
  attn_block(T, nh, tile_, h, b);
  const int qt = (T + 127) / 128 - 1 - tile_;  // heaviest (latest) query tiles first
  const int dm = nh * HD, ld = 3 * dm;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int q0 = qt * 128;
  const int qw0 = q0 + wave * 32;
  const int qrow = qw0 + l31;
  const bool qvalid = qrow < T;
  const uint16_t* base = qkv + (int64_t)b * T * ld + h * HD;
  const float scale = 0.125f;  // 1/sqrt(64)
  const float c2 = scale * LOG2E;

  bf16x8_t qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    qf[ks] = qvalid ? ld_bf16x8(base + (int64_t)qrow * ld + ks * 16 + hi * 8) : zero_bf16x8();
  int dsq = 0;
  if (qvalid) dsq = doc_start[(int64_t)b * T + qrow];
  asm volatile("; q fragments resident" ::"v"(qf[0]), "v"(qf[1]), "v"(qf[2]), "v"(qf[3]), "v"(dsq));  // consumed before any DMA is in flight

  f32x16_t o[2];
  zero16(o[0]);
  zero16(o[1]);
  float m = -INFINITY, lsum = 0.f;

  const int kv_hi = min(T, q0 + 128);
  const int jt_hi = (kv_hi + KT - 1) / KT;
  const int jt_lo = __builtin_amdgcn_readfirstlane(doc_start[(int64_t)b * T + q0]) / KT;  // tiles before the first row's document: skipped

  TileDma dma;
  dma.init(wave, lane, ld);
  auto stage = [&](int st, int jt) {
    const int kv0 = jt * KT;
    const uint16_t* src = base + (int64_t)kv0 * ld;
    if (kv0 + KT <= T) {  // whole tile inside the sequence (always, when T % 64 == 0): no per-lane address arithmetic
      dma.issue_full(smem + st * 2 * TILE, src + dm, wave);
      dma.issue_full(smem + st * 2 * TILE + TILE, src + 2 * dm, wave);
    } else {
      dma.issue(smem + st * 2 * TILE, src + dm, ld, T - 1 - kv0, wave);
      dma.issue(smem + st * 2 * TILE + TILE, src + 2 * dm, ld, T - 1 - kv0, wave);
    }
  };

  // one KV tile: S^T = K Q^T, masked online softmax, O^T += V^T P^T
  auto tile_body = [&](int jt, int st) {
    const int kv0 = jt * KT;
    if (jt + 1 < jt_hi) stage(st ^ 1, jt + 1);
    const char* sK = smem + st * 2 * TILE;
    const char* sV = sK + TILE;
    if (kv0 <= qw0 + 31) {  // tile not entirely above this wave's diagonal
      f32x16_t s[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        zero16(s[kb]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s[kb] = mfma32(frag_rows(sK, kb * 32 + l31, ks, hi), qf[ks], s[kb]);
      }
      float tmax = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kvg = kv0 + kb * 32 + mfma32_row(r, hi);
          if (!((kvg <= qrow) && (kvg >= dsq))) s[kb][r] = -INFINITY;
          tmax = fmaxf(tmax, s[kb][r]);
        }
      }
      {
        float t_lo, t_hi;
        half_pair(tmax, t_lo, t_hi);
        tmax = fmaxf(t_lo, t_hi);
      }
      const float m_new = fmaxf(m, tmax);
      const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = fast_exp2((m - m_safe) * c2);
      const float mc = m_safe * c2;
      float psum = 0.f;
      bf16x8_t pf[4];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = fast_exp2(s[kb][r] * c2 - mc);
          psum += p;
          pf[kb * 2 + (r >> 3)][r & 7] = f2bf(p);
        }
      }
      lsum = lsum * alpha + psum;
      const bool grew = m_new > m;
      m = m_new;
      if (__builtin_amdgcn_ballot_w64(grew) != 0ull) {  // wave-uniform: skip the O rescale when no row max moved
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      }
#pragma unroll
      for (int db = 0; db < 2; ++db) {
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
          const int rbase = (sp >> 1) * 32 + (sp & 1) * 16 + 4 * hi;
          o[db] = mfma32(frag_cols(sV, db, rbase, lane), pf[sp], o[db]);
        }
      }
    }
    attn_wait_vm<0>();  // next tile landed (this wave's pieces) ...
    attn_barrier();     // ... everyone's; and every wave is done reading the current stage
  };

  if (jt_lo < jt_hi) stage(0, jt_lo);
  attn_wait_vm<0>();
  attn_barrier();
  int st = 0;
  for (int jt = jt_lo; jt < jt_hi; ++jt, st ^= 1) tile_body(jt, st);

  float l_lo, l_hi;
  half_pair(lsum, l_lo, l_hi);
  const float ltot = l_lo + l_hi;
  const float inv = 1.f / ltot;
  const RowStage rs{smem + wave * 4096, lane};  // every wave is past the last tile's barrier: the stages are free (see attn_common.h)
#pragma unroll
  for (int db = 0; db < 2; ++db) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4_t v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = f2bf(o[db][4 * g + e] * inv);
      rs.put(l31, hi, db * 4 + g, v);
    }
  }
  if (qvalid && hi == 0) lse[((int64_t)b * nh + h) * T + qrow] = m * c2 + __builtin_amdgcn_logf(ltot);  // base-2 LSE: the backward's exp2 argument directly
  rs.flush(out + (int64_t)b * T * dm, dm, qw0, T, h * HD);
}

// =============================================================================================
// backward with document masks: dK, dV  (one workgroup per 128 key rows; loops over query tiles of 64 rows; q, k rotated)
//
// Why the backward stays two passes of 4-wave workgroups (round 2, profiles/r02_ubench_overlap.txt, r02_pmc_sq.txt):
// the kernels are bound by instruction issue and LDS reads, not by the matrix pipe (27 % busy) - per 32x32 block a wave
// issues 16 MFMAs next to ~100 VALU + 16 exp2 + 16 cvt + 28-32 LDS reads, and on this chip a ds_read_b128 costs its wave
// 24-33 cycles of issue, a wave doing VALU / LDS work beside an MFMA-streaming partner on the same SIMD slows down 2.5-4x.
// Measured against this kernel (177 us / layer at the 160M shape): an 8-wave form that alternates matrix and vector slots
// between the two waves of a SIMD 268 us; a one-wave-per-SIMD three-stage software pipeline (MFMAs of blocks b-1 / b+1
// interleaved with the softmax of block b) 269-285 us - with one wave per SIMD the LDS reads alone take ~1200 cycles per
// block.  A single-pass kernel (dQ with dK / dV from one recomputation) saves 8 of 28 MFMAs per block but has to move
// ~0.44 GB of fp32 dQ partials per layer through HBM twice to stay deterministic - no gain while the MFMAs are not the limit.
// =============================================================================================
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_doc_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               const float* __restrict__ rcos, const float* __restrict__ rsin,
                                                               const int32_t* __restrict__ doc_start, uint16_t* __restrict__ dqkv,
                                                               int T, int nh) {
  constexpr int QT = 64;
  constexpr int TILE = QT * 128;           // 8 KiB
  constexpr int STAGE = 2 * TILE + 1024;   // Q | dO | statistics (lse[64], delta[64], doc_start[64])
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];

  int kt, h, b;  // key tile 0 meets every query tile: heaviest first
  attn_block(T, nh, kt, h, b);
  const int dm = nh * HD, ld = 3 * dm;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int kv0 = kt * 128;
  const int kvw0 = kv0 + wave * 32;
  const int kvrow = kvw0 + l31;
  const bool kvalid = kvrow < T;
  const uint16_t* base = qkv + (int64_t)b * T * ld + h * HD;
  const uint16_t* dobase = dout + (int64_t)b * T * dm + h * HD;
  const float* lrow = lse + ((int64_t)b * nh + h) * T;
  const float* drow = delta + ((int64_t)b * nh + h) * T;
  const int32_t* dsrow = doc_start + (int64_t)b * T;
  const float scale = 0.125f, c2 = scale * LOG2E;

  bf16x8_t kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const uint16_t* p = base + (int64_t)kvrow * ld + ks * 16 + hi * 8;
    kf[ks] = kvalid ? ld_bf16x8(p + dm) : zero_bf16x8();
    vf[ks] = kvalid ? ld_bf16x8(p + 2 * dm) : zero_bf16x8();
  }
  // query-tile range: from the diagonal down; with document masks stop once a tile's first row starts
  // after this key block (doc_start is non-decreasing)
  const int nqt = (T + QT - 1) / QT;
  const int jq_lo = kv0 / QT;
  int jq_hi = jq_lo;
  while (jq_hi < nqt && __builtin_amdgcn_readfirstlane(dsrow[jq_hi * QT]) <= kv0 + 127) ++jq_hi;
  asm volatile("; k/v fragments resident" ::"v"(kf[0]), "v"(kf[1]), "v"(kf[2]), "v"(kf[3]), "v"(vf[0]), "v"(vf[1]), "v"(vf[2]),
               "v"(vf[3]));  // every ordinary load is consumed before the first DMA is in flight

  f32x16_t dk[2], dv[2];
  zero16(dk[0]); zero16(dk[1]); zero16(dv[0]); zero16(dv[1]);

  TileDma dma, dmad;
  dma.init(wave, lane, ld);
  dmad.init(wave, lane, dm);
  auto stage = [&](int st, int jq) {
    const int qt0 = jq * QT;
    char* dst = smem + st * STAGE;
    if (qt0 + QT <= T) {
      dma.issue_full(dst, base + (int64_t)qt0 * ld, wave);
      dmad.issue_full(dst + TILE, dobase + (int64_t)qt0 * dm, wave);
    } else {
      dma.issue(dst, base + (int64_t)qt0 * ld, ld, T - 1 - qt0, wave);
      dma.issue(dst + TILE, dobase + (int64_t)qt0 * dm, dm, T - 1 - qt0, wave);
    }
    if (wave == 0 && lane < 16) {  // 64 floats = 16 lanes x 16 bytes per statistic (T % 4 == 0 is checked on the host)
      const int q = min(qt0 + lane * 4, T - 4);
      dma16_asm(lrow + q, dst + 2 * TILE);
      dma16_asm(drow + q, dst + 2 * TILE + 256);
      dma16_asm(dsrow + q, dst + 2 * TILE + 512);
    }
  };

  auto tile_body = [&](int jq, int st) {
    const int qt0 = jq * QT;
    if (jq + 1 < jq_hi) stage(st ^ 1, jq + 1);
    const char* sQ = smem + st * STAGE;
    const char* sDO = sQ + TILE;
    const float* sL = reinterpret_cast<const float*>(sQ + 2 * TILE);
    const float* sD = sL + 64;
    const int* sDS = reinterpret_cast<const int*>(sL + 128);
    if (qt0 + QT - 1 >= kvw0) {  // some query at or below this wave's first key
#pragma unroll 1
      for (int qb = 0; qb < 2; ++qb) {
        f32x16_t s, dp;
        zero16(s);
        zero16(dp);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s = mfma32(frag_rows(sQ, qb * 32 + l31, ks, hi), kf[ks], s);       // S[q][kv]
          dp = mfma32(frag_rows(sDO, qb * 32 + l31, ks, hi), vf[ks], dp);    // dP[q][kv]
        }
        bf16x8_t pf[2], dsf[2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          // this lane's query rows for registers 4g..4g+3 are consecutive: one 16-byte read per statistic
          const int ql0 = qb * 32 + 8 * g + 4 * hi;
          const f32x4_t L4 = *reinterpret_cast<const f32x4_t*>(sL + ql0);  // base-2 LSE
          const f32x4_t D4 = *reinterpret_cast<const f32x4_t*>(sD + ql0);
          int ds4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) ds4[e] = sDS[ql0 + e];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            float p = fast_exp2(__builtin_fmaf(s[r], c2, -L4[e]));  // explicit fma: hipcc otherwise pairs s*c2 with lse*LOG2E in a v_pk_mul (16 v_mov per block)
            const int qg = qt0 + ql0 + e;
            p = ((kvrow <= qg) && (qg < T) && (kvrow >= ds4[e])) ? p : 0.f;
            const float dsv = p * (dp[r] - D4[e]);  // the 1/sqrt(hd) factor (a power of two: exact) is applied once, to dK, in the epilogue
            pf[r >> 3][r & 7] = f2bf(p);
            dsf[r >> 3][r & 7] = f2bf(dsv);
          }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int rbase = qb * 32 + s2 * 16 + 4 * hi;
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            dv[db] = mfma32(frag_cols(sDO, db, rbase, lane), pf[s2], dv[db]);   // dV^T[d][kv]
            dk[db] = mfma32(frag_cols(sQ, db, rbase, lane), dsf[s2], dk[db]);   // dK^T[d][kv]
          }
        }
      }
    }
    attn_wait_vm<0>();
    attn_barrier();
  };

  if (jq_lo < jq_hi) stage(0, jq_lo);
  attn_wait_vm<0>();
  attn_barrier();
  int st = 0;
  for (int jq = jq_lo; jq < jq_hi; ++jq, st ^= 1) tile_body(jq, st);

  const RowStage rs{smem + wave * 4096, lane};  // every wave is past the last tile's barrier: the stages are free
  const int trow = min(kvrow, T - 1) * 32;
#pragma unroll
  for (int db = 0; db < 2; ++db) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4_t ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[e] = f2bf(dv[db][4 * g + e]);
      rs.put(l31, hi, db * 4 + g, ov);
    }
  }
  rs.flush(dqkv + (int64_t)b * T * ld, ld, kvw0, T, 2 * dm + h * HD);
#pragma unroll
  for (int db = 0; db < 2; ++db) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = db * 32 + 8 * g + 4 * hi;
      // inverse rotation of the two (even, odd) pairs of dK: gradient w.r.t. the PRE-rotation k
      const float c0 = rcos[trow + d0 / 2], c1 = rcos[trow + d0 / 2 + 1];
      const float s0 = rsin[trow + d0 / 2], s1 = rsin[trow + d0 / 2 + 1];
      const float a0 = dk[db][4 * g + 0] * scale, b0 = dk[db][4 * g + 1] * scale, a1 = dk[db][4 * g + 2] * scale, b1 = dk[db][4 * g + 3] * scale;
      bf16x4_t ok;
      ok[0] = f2bf(a0 * c0 + b0 * s0);
      ok[1] = f2bf(b0 * c0 - a0 * s0);
      ok[2] = f2bf(a1 * c1 + b1 * s1);
      ok[3] = f2bf(b1 * c1 - a1 * s1);
      rs.put(l31, hi, db * 4 + g, ok);
    }
  }
  rs.flush(dqkv + (int64_t)b * T * ld, ld, kvw0, T, dm + h * HD);
}

// =============================================================================================
// backward with document masks: dQ  (one workgroup per 128 query rows; loops over key tiles of 64 rows; q, k rotated)
// =============================================================================================
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_doc_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out,
                                                             const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                                                             float* __restrict__ delta,
                                                             const float* __restrict__ rcos, const float* __restrict__ rsin,
                                                             const int32_t* __restrict__ doc_start, uint16_t* __restrict__ dqkv, int T,
                                                             int nh) {
  constexpr int KT = 64;
  constexpr int TILE = KT * 128;
  __shared__ __attribute__((aligned(1024))) char smem[2 * 2 * TILE];  // [stage][K|V]

  int tile_, h, b;
  attn_block(T, nh, tile_, h, b);
  const int qt = (T + 127) / 128 - 1 - tile_;
  const int dm = nh * HD, ld = 3 * dm;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int q0 = qt * 128, qw0 = q0 + wave * 32, qrow = qw0 + l31;
  const bool qvalid = qrow < T;
  const uint16_t* base = qkv + (int64_t)b * T * ld + h * HD;
  const float scale = 0.125f, c2 = scale * LOG2E;

  bf16x8_t qf[4], dof[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int d0 = ks * 16 + hi * 8;
    qf[ks] = qvalid ? ld_bf16x8(base + (int64_t)qrow * ld + d0) : zero_bf16x8();
    dof[ks] = qvalid ? ld_bf16x8(dout + ((int64_t)b * T + qrow) * dm + h * HD + d0) : zero_bf16x8();
  }
  // delta[q] = sum_d dO[q][d] * O[q][d] (the softmax-backward row term) is computed HERE - the lane pair (hi = 0, 1) of a
  // query holds all 64 dims of its dO row in the fragments above - and published for the dK/dV kernel, which runs after
  // this one: no separate pre-pass over O and dO.
  float Lq = 0.f, Dq = 0.f;
  int dsq = 0;
  if (qvalid) {
    Lq = lse[((int64_t)b * nh + h) * T + qrow];  // base-2 LSE
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8_t o8 = ld_bf16x8(out + ((int64_t)b * T + qrow) * dm + h * HD + ks * 16 + hi * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) part += bf2f(o8[e]) * bf2f(dof[ks][e]);
    }
    Dq = part;
    dsq = doc_start[(int64_t)b * T + qrow];
  }
  {
    float d_lo, d_hi;
    half_pair(Dq, d_lo, d_hi);  // rows beyond T hold zeros in both halves
    Dq = d_lo + d_hi;
    if (qvalid && hi == 0) delta[((int64_t)b * nh + h) * T + qrow] = Dq;
  }
  const int kv_hi = min(T, q0 + 128);
  const int jt_hi = (kv_hi + KT - 1) / KT;
  const int jt_lo = __builtin_amdgcn_readfirstlane(doc_start[(int64_t)b * T + q0]) / KT;
  asm volatile("; q/dO fragments resident" ::"v"(qf[0]), "v"(qf[1]), "v"(qf[2]), "v"(qf[3]), "v"(dof[0]), "v"(dof[1]), "v"(dof[2]),
               "v"(dof[3]), "v"(Lq), "v"(Dq), "v"(dsq));

  f32x16_t dq[2];
  zero16(dq[0]);
  zero16(dq[1]);

  TileDma dma;
  dma.init(wave, lane, ld);
  auto stage = [&](int st, int jt) {
    const int kv0 = jt * KT;
    const uint16_t* src = base + (int64_t)kv0 * ld;
    if (kv0 + KT <= T) {
      dma.issue_full(smem + st * 2 * TILE, src + dm, wave);
      dma.issue_full(smem + st * 2 * TILE + TILE, src + 2 * dm, wave);
    } else {
      dma.issue(smem + st * 2 * TILE, src + dm, ld, T - 1 - kv0, wave);
      dma.issue(smem + st * 2 * TILE + TILE, src + 2 * dm, ld, T - 1 - kv0, wave);
    }
  };

  auto tile_body = [&](int jt, int st) {
    const int kv0 = jt * KT;
    if (jt + 1 < jt_hi) stage(st ^ 1, jt + 1);
    const char* sK = smem + st * 2 * TILE;
    const char* sV = sK + TILE;
    if (kv0 <= qw0 + 31) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        f32x16_t s, dp;
        zero16(s);
        zero16(dp);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s = mfma32(frag_rows(sK, kb * 32 + l31, ks, hi), qf[ks], s);      // S^T[kv][q]
          dp = mfma32(frag_rows(sV, kb * 32 + l31, ks, hi), dof[ks], dp);   // dP^T[kv][q]
        }
        bf16x8_t dsf[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float p = fast_exp2(__builtin_fmaf(s[r], c2, -Lq));
          const int kvg = kv0 + kb * 32 + mfma32_row(r, hi);
          p = ((kvg <= qrow) && (kvg >= dsq)) ? p : 0.f;
          dsf[r >> 3][r & 7] = f2bf(p * (dp[r] - Dq));  // x 1/sqrt(hd) once, in the epilogue
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int rbase = kb * 32 + s2 * 16 + 4 * hi;
#pragma unroll
          for (int db = 0; db < 2; ++db) dq[db] = mfma32(frag_cols(sK, db, rbase, lane), dsf[s2], dq[db]);  // dQ^T[d][q]
        }
      }
    }
    attn_wait_vm<0>();
    attn_barrier();
  };

  if (jt_lo < jt_hi) stage(0, jt_lo);
  attn_wait_vm<0>();
  attn_barrier();
  int st = 0;
  for (int jt = jt_lo; jt < jt_hi; ++jt, st ^= 1) tile_body(jt, st);

  const RowStage rs{smem + wave * 4096, lane};  // every wave is past the last tile's barrier: the stages are free
  const int trow = min(qrow, T - 1) * 32;
#pragma unroll
  for (int db = 0; db < 2; ++db) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = db * 32 + 8 * g + 4 * hi;
      const float c0 = rcos[trow + d0 / 2], c1 = rcos[trow + d0 / 2 + 1];
      const float s0 = rsin[trow + d0 / 2], s1 = rsin[trow + d0 / 2 + 1];
      const float a0 = dq[db][4 * g + 0] * scale, b0 = dq[db][4 * g + 1] * scale, a1 = dq[db][4 * g + 2] * scale, b1 = dq[db][4 * g + 3] * scale;
      bf16x4_t ov;
      ov[0] = f2bf(a0 * c0 + b0 * s0);
      ov[1] = f2bf(b0 * c0 - a0 * s0);
      ov[2] = f2bf(a1 * c1 + b1 * s1);
      ov[3] = f2bf(b1 * c1 - a1 * s1);
      rs.put(l31, hi, db * 4 + g, ov);
    }
  }
  rs.flush(dqkv + (int64_t)b * T * ld, ld, qw0, T, h * HD);
}

// =============================================================================================
// C ABI
// =============================================================================================
// causal batches (no document mask): attn_causal.hip
void plm_attn_fwd_causal(const uint16_t* qkv, uint16_t* out, float* lse, int64_t B, int64_t T, int64_t nh, hipStream_t s);
void plm_attn_bwd_causal(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* lse, float* delta, const float* rc,
                         const float* rs, uint16_t* dqkv, int64_t B, int64_t T, int64_t nh, hipStream_t s);

static int check_attn_shape(const char* name, int64_t B, int64_t T, int64_t nh, int64_t hd) {
  PLM_REQUIRE(hd == HD, "%s: head_dim %ld unsupported (this build implements head_dim 64)", name, (long)hd);
  PLM_REQUIRE(B > 0 && T > 0 && nh > 0 && B < 65536 && nh < 65536 && T < (1 << 24), "%s: bad shape B=%ld T=%ld nh=%ld", name, (long)B,
              (long)T, (long)nh);
  PLM_REQUIRE(T % 4 == 0, "%s: T=%ld must be a multiple of 4", name, (long)T);
  return PLM_OK;
}

extern "C" int plm_rope_qk(uint16_t* qkv, const float* rope_cos, const float* rope_sin, int64_t B, int64_t T, int64_t nh, int64_t hd,
                           void* stream) {
  PLM_REQUIRE(qkv && rope_cos && rope_sin, "plm_rope_qk: null pointer");
  if (int rc = check_attn_shape("plm_rope_qk", B, T, nh, hd)) return rc;
  const int64_t items = B * T * (2 * nh * hd / 8);
  int64_t blocks = plm_cdiv(items, 256);
  if (blocks > ((int64_t)1 << 20)) blocks = (int64_t)1 << 20;  // one item per thread in memory order (see elementwise_grid)
  hipLaunchKernelGGL(rope_qk_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, qkv, rope_cos, rope_sin, B * T, (int)T, (int)nh);
  PLM_CHECK_LAUNCH("plm_rope_qk");
  return PLM_OK;
}

extern "C" int plm_attn_fwd(const uint16_t* qkv, const int32_t* doc_start, uint16_t* out, float* lse, int64_t B, int64_t T, int64_t nh,
                            int64_t hd, void* stream) {
  PLM_REQUIRE(qkv && out && lse, "plm_attn_fwd: null pointer");
  // LDS-DMA sources and the whole-row epilogue stores (RowStage::flush) are 16-byte accesses
  PLM_REQUIRE(((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, "plm_attn_fwd: qkv and out must be 16-byte aligned");
  if (int rc = check_attn_shape("plm_attn_fwd", B, T, nh, hd)) return rc;
  const dim3 grid((unsigned)(plm_cdiv(T, 128) * nh * B)), block(256);  // see attn_block
  hipStream_t s = (hipStream_t)stream;
  if (!doc_start) {
    plm_attn_fwd_causal(qkv, out, lse, B, T, nh, s);
    PLM_CHECK_LAUNCH("plm_attn_fwd");
    return PLM_OK;
  }
  hipLaunchKernelGGL(attn_fwd_doc_kernel, grid, block, 0, s, qkv, doc_start, out, lse, (int)T, (int)nh);
  PLM_CHECK_LAUNCH("plm_attn_fwd");
  return PLM_OK;
}

extern "C" int plm_attn_bwd(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* lse, const float* rope_cos,
                            const float* rope_sin, const int32_t* doc_start, uint16_t* dqkv, float* delta, int64_t B, int64_t T,
                            int64_t nh, int64_t hd, void* stream) {
  PLM_REQUIRE(qkv && out && dout && lse && rope_cos && rope_sin && dqkv && delta, "plm_attn_bwd: null pointer");
  PLM_REQUIRE(((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(dqkv) |
                reinterpret_cast<uintptr_t>(rope_cos) | reinterpret_cast<uintptr_t>(rope_sin)) & 15) == 0,
              "plm_attn_bwd: qkv, out, dout, dqkv and the RoPE tables must be 16-byte aligned");
  if (int rc = check_attn_shape("plm_attn_bwd", B, T, nh, hd)) return rc;
  hipStream_t s = (hipStream_t)stream;
  const dim3 block(256);
  // dQ first: it computes delta[b,h,q] for its queries and publishes it for the dK/dV kernel
  const dim3 gkv((unsigned)(plm_cdiv(T, 128) * nh * B));
  if (!doc_start) {
    plm_attn_bwd_causal(qkv, out, dout, lse, delta, rope_cos, rope_sin, dqkv, B, T, nh, s);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_doc_kernel, gkv, block, 0, s, qkv, out, dout, lse, delta, rope_cos, rope_sin, doc_start, dqkv, (int)T, (int)nh);
    hipLaunchKernelGGL(attn_bwd_dkdv_doc_kernel, gkv, block, 0, s, qkv, dout, lse, delta, rope_cos, rope_sin, doc_start, dqkv, (int)T, (int)nh);
  }
  PLM_CHECK_LAUNCH("plm_attn_bwd");
  return PLM_OK;
}
