// Attention forward / backward, second generation: 64 rows per wave.
//
// Why (profiles/r02_pmc.txt, r02_ubench_overlap.txt): the first-generation kernels (attn.hip: 4 waves x 32 rows, 3-4 waves per SIMD)
// ran the matrix pipe 25 % busy.  Their waves are serial chains  ds_read -> s_waitcnt -> MFMA  (every operand fragment read from LDS
// right in front of the MFMA that consumes it), and every wave re-reads the WHOLE K / V (Q / dO) tile from LDS for 32 rows of its own:
// 1 KiB of LDS reads per MFMA, and an LDS read costs its SIMD 24-33 cycles of issue whether the wave itself or its neighbour streams
// MFMAs.  Here a wave owns 64 rows (two 32-row blocks): every fragment read from LDS feeds two MFMAs, all fragments of a tile are
// requested up front, and the softmax arithmetic of one row block sits next to the MFMAs of the other in the same basic block.
// The forward softmax defers the running-max update (the O rescale) until a row's maximum has grown by more than 2^8: with the exact
// maximum some row of a wave grows in almost every tile, so the "skip when nothing moved" test of generation one never skipped.
//
// Same LDS image, LDS-DMA staging, masks and C ABI as attn.hip (see there and attn_common.h).
#include "plm_device.h"

#include <type_traits>

#include "attn_common.h"

// Block -> (row tile of RB rows, head, batch), heaviest (latest) tiles first: see attn_block() in attn.hip.
template <int RB>
__device__ __forceinline__ void attn_block2(int T, int nh, int& tile, int& h, int& b) {
  const int ntile = (T + RB - 1) / RB;
  const int nbh = gridDim.x / ntile;
  const int bh = blockIdx.x % nbh;
  tile = blockIdx.x / nbh;
  h = bh % nh;
  b = bh / nh;
}

#define ATTN_DEFER_LOG2 8.0f  // the running maximum is updated when a row's new maximum exceeds it by more than 2^8 (P <= 256)

// =============================================================================================
// forward: one workgroup = 4 waves x (32 * NQB) query rows; key tiles of 64 rows through two LDS stages
// =============================================================================================
// ABL: timing-only ablations (results are garbage): 1 no exp, 2 no Q K^T MFMAs, 4 no P V MFMAs, 8 no LDS fragment reads, 16 no LDS-DMA,
// 32 no barriers, 64 no softmax arithmetic at all
template <bool HAS_DOC, int NQB, int MINW, int ABL = 0>
__global__ __launch_bounds__(256, MINW) void attn_fwd2_kernel(const uint16_t* __restrict__ qkv, const int32_t* __restrict__ doc_start,
                                                              uint16_t* __restrict__ out, float* __restrict__ lse, int T, int nh) {
  constexpr int KT = 64;
  constexpr int TILE = KT * 128;  // 8 KiB
  constexpr int QW = 32 * NQB, QB = 4 * QW;
  __shared__ __attribute__((aligned(1024))) char smem[2 * 2 * TILE];  // [stage][K|V]

  int tile_, h, b;
  attn_block2<QB>(T, nh, tile_, h, b);
  const int qt = (T + QB - 1) / QB - 1 - tile_;
  const int dm = nh * HD, ld = 3 * dm;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int q0 = qt * QB, qw0 = q0 + wave * QW;
  const uint16_t* base = qkv + (int64_t)b * T * ld + h * HD;
  const float c2 = 0.125f * LOG2E;  // 1/sqrt(64) and the base-2 exponent in one factor

  bf16x8_t qf[NQB][4];
  int dsq[NQB];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    const int qrow = qw0 + qb * 32 + l31;
    const bool qvalid = qrow < T;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = qvalid ? ld_bf16x8(base + (int64_t)qrow * ld + ks * 16 + hi * 8) : zero_bf16x8();
    dsq[qb] = 0;
    if (HAS_DOC && qvalid) dsq[qb] = doc_start[(int64_t)b * T + qrow];
  }
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb)
    asm volatile("; q fragments resident" ::"v"(qf[qb][0]), "v"(qf[qb][1]), "v"(qf[qb][2]), "v"(qf[qb][3]), "v"(dsq[qb]));  // consumed before any DMA is in flight

  f32x16_t o[NQB][2];
  float mc[NQB], lsum[NQB];  // running reference maximum in log2 units (s * c2), running sum
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    zero16(o[qb][0]);
    zero16(o[qb][1]);
    mc[qb] = -INFINITY;
    lsum[qb] = 0.f;
  }

  const int kv_hi = min(T, q0 + QB);
  const int jt_hi = (kv_hi + KT - 1) / KT;
  int jt_lo = 0;
  if (HAS_DOC) jt_lo = __builtin_amdgcn_readfirstlane(doc_start[(int64_t)b * T + q0]) / KT;

  TileDma dma;
  dma.init(wave, lane, ld);
  auto stage = [&](int st, int jt) {
    if (ABL & 16) return;
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

  // one KV tile for this wave's NQB row blocks: S^T = K Q^T, online softmax, O^T += V^T P^T
  auto compute = [&](int kv0, const char* sK, const char* sV, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    bf16x8_t kfr[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        kfr[kb][ks] = (ABL & 8) ? qf[0][ks] : frag_rows(sK, kb * 32 + l31, ks, hi);
        if (ABL & 8) asm volatile("" : "+v"(kfr[kb][ks]));
      }
    f32x16_t s[NQB][2];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        zero16(s[qb][kb]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ABL & 2)
            asm volatile("" : "+v"(s[qb][kb]) : "v"(kfr[kb][ks]));
          else
            s[qb][kb] = mfma32(kfr[kb][ks], qf[qb][ks], s[qb][kb]);
        }
      }
    if (MINW > 1) __builtin_amdgcn_sched_barrier(0);  // 256 registers: keep the V fragments out of the Q K^T phase
    bf16x8_t vfr[2][4];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        vfr[db][sp] = (ABL & 8) ? qf[0][sp] : frag_cols(sV, db, (sp >> 1) * 32 + (sp & 1) * 16 + 4 * hi, lane);
        if (ABL & 8) asm volatile("" : "+v"(vfr[db][sp]));
      }
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      if (ABL & 64) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int sp = 0; sp < 4; ++sp) {
            f32x4_t t4 = {s[qb][sp >> 1][(sp & 1) * 8], s[qb][sp >> 1][(sp & 1) * 8 + 1], s[qb][sp >> 1][(sp & 1) * 8 + 2], s[qb][sp >> 1][(sp & 1) * 8 + 3]};
            const bf16x8_t pq = __builtin_bit_cast(bf16x8_t, t4);
            if (ABL & 4)
              asm volatile("" : "+v"(o[qb][db]) : "v"(vfr[db][sp]), "v"(pq));
            else
              o[qb][db] = mfma32(vfr[db][sp], pq, o[qb][db]);
          }
        continue;
      }
      // key (kb, r) of this lane is tile row kb*32 + (r&3) + 8*(r>>2) + 4*hi: visible iff  c_lo <= kb*32 + (r&3) + 8*(r>>2) <= c_hi
      const int c_hi = qw0 + qb * 32 + l31 - kv0 - 4 * hi;
      const int c_lo = HAS_DOC ? dsq[qb] - kv0 - 4 * hi : 0;
      float tmax = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (MASK) {
            const int c = kb * 32 + (r & 3) + 8 * (r >> 2);
            const bool ok = (c <= c_hi) && (!HAS_DOC || c >= c_lo);
            if (!ok) s[qb][kb][r] = -INFINITY;
          }
          tmax = fmaxf(tmax, s[qb][kb][r]);
        }
      {
        float t_lo, t_hi;
        half_pair(tmax, t_lo, t_hi);
        tmax = fmaxf(t_lo, t_hi);
      }
      const float tm = tmax * c2;
      const bool need = tm > mc[qb] + ATTN_DEFER_LOG2;  // both -inf (nothing visible yet): false
      if (__builtin_amdgcn_ballot_w64(need) != 0ull) {   // wave-uniform and rare after the first tile
        const float mn = fmaxf(mc[qb], tm);
        const float alpha = fast_exp2(mc[qb] - ((mn == -INFINITY) ? 0.f : mn));
        mc[qb] = mn;
        lsum[qb] *= alpha;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[qb][db][r] *= alpha;
      }
      const float mref = (MASK && mc[qb] == -INFINITY) ? 0.f : mc[qb];
      float ps[4] = {0.f, 0.f, 0.f, 0.f};
      bf16x8_t pf[4];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float px = __builtin_fmaf(s[qb][kb][r], c2, -mref);
          const float p = (ABL & 1) ? px : fast_exp2(px);
          ps[r & 3] += p;
          pf[kb * 2 + (r >> 3)][r & 7] = f2bf(p);
        }
      lsum[qb] += (ps[0] + ps[1]) + (ps[2] + ps[3]);
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
          if (ABL & 4)
            asm volatile("" : "+v"(o[qb][db]) : "v"(vfr[db][sp]), "v"(pf[sp]));
          else
            o[qb][db] = mfma32(vfr[db][sp], pf[sp], o[qb][db]);
        }
    }
  };

  // Every wave walks all tiles of the workgroup (DMA issue, one barrier per tile) in three consecutive loops of its own: tiles entirely
  // below its first query row (no mask), the tiles its diagonal crosses (masked; with document masks every tile), and the tiles above
  // its last row (nothing to compute).  Separate loops keep the mask code out of the hot loop's register allocation.
  const bool wave_rows = qw0 < T;
  const int jt_act = wave_rows ? min(jt_hi, (qw0 + QW - 1) / KT + 1) : jt_lo;
  const int jt_um = HAS_DOC ? jt_lo : min(jt_act, max(jt_lo, (qw0 + 1) / KT));
  if (jt_lo < jt_hi) stage(0, jt_lo);
  attn_wait_vm<0>();
  attn_barrier();
  int st = 0, jt = jt_lo;
  auto run = [&](int jt_end, auto mask_tag, bool active) {
    for (; jt < jt_end; ++jt, st ^= 1) {
      if (jt + 1 < jt_hi) stage(st ^ 1, jt + 1);
      if (active) compute(jt * KT, smem + st * 2 * TILE, smem + st * 2 * TILE + TILE, mask_tag);
      attn_wait_vm<0>();  // next tile landed (this wave's pieces) ...
      if (!(ABL & 32)) attn_barrier();     // ... everyone's; and every wave is done reading the current stage
    }
  };
  run(jt_um, std::false_type{}, true);
  run(jt_act, std::true_type{}, true);
  run(jt_hi, std::true_type{}, false);

#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    const int qrow = qw0 + qb * 32 + l31;
    float l_lo, l_hi;
    half_pair(lsum[qb], l_lo, l_hi);
    const float ltot = l_lo + l_hi;
    if (qrow < T) {
      const float inv = 1.f / ltot;
      uint16_t* op = out + ((int64_t)b * T + qrow) * dm + h * HD;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4_t v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = f2bf(o[qb][db][4 * g + e] * inv);
          st_bf16x4(op + db * 32 + 8 * g + 4 * hi, v);
        }
      if (hi == 0) lse[((int64_t)b * nh + h) * T + qrow] = mc[qb] + __builtin_amdgcn_logf(ltot);  // base-2 LSE: the backward's exp2 argument directly
    }
  }
}

// =============================================================================================
// forward, third form: one workgroup = 4 waves x 64 query rows (256-row tiles), wave w owns the 32-row blocks w and 7 - w of the tile -
// every wave of a causal tile then has the same amount of work (block w ends 7 - 2w blocks before block 7 - w) and no wave idles through
// the diagonal region; K / V tiles of 64 rows go through an NST-deep LDS ring filled NST - 1 tiles ahead (counted vmcnt waits: the wait
// in front of tile i only covers tile i), one barrier per tile.
// Per tile a row block is OFF (tile above its diagonal), UM (no mask needed) or MASK.
// =============================================================================================
enum { QB_OFF = 0, QB_UM = 1, QB_MASK = 2 };

template <bool HAS_DOC, int NST, int MINW, int ABL = 0>
__global__ __launch_bounds__(256, MINW) void attn_fwd3_kernel(const uint16_t* __restrict__ qkv, const int32_t* __restrict__ doc_start,
                                                              uint16_t* __restrict__ out, float* __restrict__ lse, int T, int nh) {
  constexpr int KT = 64;
  constexpr int TILE = KT * 128;  // 8 KiB
  constexpr int QB = 256;
  __shared__ __attribute__((aligned(1024))) char smem[NST * 2 * TILE];  // [stage][K|V]

  int tile_, h, b;
  attn_block2<QB>(T, nh, tile_, h, b);
  const int qt = (T + QB - 1) / QB - 1 - tile_;
  const int dm = nh * HD, ld = 3 * dm;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int q0 = qt * QB;
  const int r0[2] = {q0 + 32 * wave, q0 + 32 * (7 - wave)};  // first rows of this wave's two blocks
  const uint16_t* base = qkv + (int64_t)b * T * ld + h * HD;
  const float c2 = 0.125f * LOG2E;  // 1/sqrt(64) and the base-2 exponent in one factor

  bf16x8_t qf[2][4];
  int dsq[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = r0[qb] + l31;
    const bool qvalid = qrow < T;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = qvalid ? ld_bf16x8(base + (int64_t)qrow * ld + ks * 16 + hi * 8) : zero_bf16x8();
    dsq[qb] = 0;
    if (HAS_DOC && qvalid) dsq[qb] = doc_start[(int64_t)b * T + qrow];
  }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
    asm volatile("; q fragments resident" ::"v"(qf[qb][0]), "v"(qf[qb][1]), "v"(qf[qb][2]), "v"(qf[qb][3]), "v"(dsq[qb]));  // consumed before any DMA is in flight

  f32x16_t o[2][2];
  float mc[2], lsum[2];  // running reference maximum in log2 units (s * c2), running sum
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    zero16(o[qb][0]);
    zero16(o[qb][1]);
    mc[qb] = -INFINITY;
    lsum[qb] = 0.f;
  }

  const int kv_hi = min(T, q0 + QB);
  const int jt_hi = (kv_hi + KT - 1) / KT;
  int jt_lo = 0;
  if (HAS_DOC) jt_lo = __builtin_amdgcn_readfirstlane(doc_start[(int64_t)b * T + q0]) / KT;
  const int n = jt_hi - jt_lo;

  TileDma dma;
  dma.init(wave, lane, ld);
  auto stage = [&](int slot, int jt) {  // 4 LDS-DMA instructions per wave
    if (ABL & 16) return;
    const int kv0 = jt * KT;
    const uint16_t* src = base + (int64_t)kv0 * ld;
    if (kv0 + KT <= T) {  // whole tile inside the sequence (always, when T % 64 == 0): no per-lane address arithmetic
      dma.issue_full(smem + slot * 2 * TILE, src + dm, wave);
      dma.issue_full(smem + slot * 2 * TILE + TILE, src + 2 * dm, wave);
    } else {
      dma.issue(smem + slot * 2 * TILE, src + dm, ld, T - 1 - kv0, wave);
      dma.issue(smem + slot * 2 * TILE + TILE, src + 2 * dm, ld, T - 1 - kv0, wave);
    }
  };

  // one row block's softmax + P V for one tile (s: its S^T accumulators, vfr: the tile's V fragments)
  auto soft_pv = [&](int qb_, f32x16_t (&s)[2], const bf16x8_t (&vfr)[2][4], int kv0, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const int qb = qb_;
    if (ABL & 64) {
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
          f32x4_t t4 = {s[sp >> 1][(sp & 1) * 8], s[sp >> 1][(sp & 1) * 8 + 1], s[sp >> 1][(sp & 1) * 8 + 2], s[sp >> 1][(sp & 1) * 8 + 3]};
          const bf16x8_t pq = __builtin_bit_cast(bf16x8_t, t4);
          if (ABL & 4)
            asm volatile("" : "+v"(o[qb][db]) : "v"(vfr[db][sp]), "v"(pq));
          else
            o[qb][db] = mfma32(vfr[db][sp], pq, o[qb][db]);
        }
      return;
    }
    // key (kb, r) of this lane is tile row kb*32 + (r&3) + 8*(r>>2) + 4*hi: visible iff  c_lo <= kb*32 + (r&3) + 8*(r>>2) <= c_hi
    const int c_hi = r0[qb] + l31 - kv0 - 4 * hi;
    const int c_lo = HAS_DOC ? dsq[qb] - kv0 - 4 * hi : 0;
    float tmax = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (MASK) {
          const int c = kb * 32 + (r & 3) + 8 * (r >> 2);
          const bool ok = (c <= c_hi) && (!HAS_DOC || c >= c_lo);
          if (!ok) s[kb][r] = -INFINITY;
        }
        tmax = fmaxf(tmax, s[kb][r]);
      }
    {
      float t_lo, t_hi;
      half_pair(tmax, t_lo, t_hi);
      tmax = fmaxf(t_lo, t_hi);
    }
    const float tm = tmax * c2;
    const bool need = tm > mc[qb] + ATTN_DEFER_LOG2;  // both -inf (nothing visible yet): false
    if (__builtin_amdgcn_ballot_w64(need) != 0ull) {   // wave-uniform and rare after the first tile
      const float mn = fmaxf(mc[qb], tm);
      const float alpha = fast_exp2(mc[qb] - ((mn == -INFINITY) ? 0.f : mn));
      mc[qb] = mn;
      lsum[qb] *= alpha;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[qb][db][r] *= alpha;
    }
    const float mref = (MASK && mc[qb] == -INFINITY) ? 0.f : mc[qb];
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
    bf16x8_t pf[4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float px = __builtin_fmaf(s[kb][r], c2, -mref);
        const float p = (ABL & 1) ? px : fast_exp2(px);
        ps[r & 3] += p;
        pf[kb * 2 + (r >> 3)][r & 7] = f2bf(p);
      }
    lsum[qb] += (ps[0] + ps[1]) + (ps[2] + ps[3]);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        if (ABL & 4)
          asm volatile("" : "+v"(o[qb][db]) : "v"(vfr[db][sp]), "v"(pf[sp]));
        else
          o[qb][db] = mfma32(vfr[db][sp], pf[sp], o[qb][db]);
      }
  };

  // one KV tile: S^T = K Q^T for the active row blocks, then softmax + P V per block.  M0 / M1: mode of block 0 / 1.
  auto compute = [&](int kv0, const char* sK, const char* sV, auto m0_tag, auto m1_tag) {
    constexpr int M0 = decltype(m0_tag)::value, M1 = decltype(m1_tag)::value;
    bf16x8_t kfr[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        kfr[kb][ks] = (ABL & 8) ? qf[0][ks] : frag_rows(sK, kb * 32 + l31, ks, hi);
        if (ABL & 8) asm volatile("" : "+v"(kfr[kb][ks]));
      }
    f32x16_t s[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      if ((qb == 0 ? M0 : M1) == QB_OFF) continue;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        zero16(s[qb][kb]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ABL & 2)
            asm volatile("" : "+v"(s[qb][kb]) : "v"(kfr[kb][ks]));
          else
            s[qb][kb] = mfma32(kfr[kb][ks], qf[qb][ks], s[qb][kb]);
        }
      }
    }
    if (MINW > 1) __builtin_amdgcn_sched_barrier(0);  // 256 registers: keep the V fragments out of the Q K^T phase
    bf16x8_t vfr[2][4];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        vfr[db][sp] = (ABL & 8) ? qf[0][sp] : frag_cols(sV, db, (sp >> 1) * 32 + (sp & 1) * 16 + 4 * hi, lane);
        if (ABL & 8) asm volatile("" : "+v"(vfr[db][sp]));
      }
    if (M0 == QB_UM) soft_pv(0, s[0], vfr, kv0, std::false_type{});
    if (M0 == QB_MASK) soft_pv(0, s[0], vfr, kv0, std::true_type{});
    if (M1 == QB_UM) soft_pv(1, s[1], vfr, kv0, std::false_type{});
    if (M1 == QB_MASK) soft_pv(1, s[1], vfr, kv0, std::true_type{});
  };

  // ring: tile i = jt - jt_lo lives in slot i % NST; tiles are issued NST - 1 ahead
#pragma unroll
  for (int i = 0; i < NST - 1; ++i)
    if (i < n) stage(i, jt_lo + i);
  int i = 0, slot = 0;
  auto run = [&](int jt_end, auto m0_tag, auto m1_tag, bool active) {
    for (; jt_lo + i < jt_end; ++i) {
      // wait for this wave's pieces of tile i: the tiles issued after it (at most NST - 2, fewer at the end) may stay in flight
      const int rem = min(NST - 2, n - 1 - i);
      if (NST >= 4 && rem >= 2) attn_wait_vm<8>();
      else if (NST >= 3 && rem == 1) attn_wait_vm<4>();
      else attn_wait_vm<0>();
      if (!(ABL & 32)) attn_barrier();  // everyone's pieces landed; and every wave is done reading tile i - 1, whose slot is refilled now
      if (i + NST - 1 < n) stage(slot == 0 ? NST - 1 : slot - 1, jt_lo + i + NST - 1);
      if (active) compute((jt_lo + i) * KT, smem + slot * 2 * TILE, smem + slot * 2 * TILE + TILE, m0_tag, m1_tag);
      slot = (slot + 1 == NST) ? 0 : slot + 1;
    }
  };
  using OFF_ = std::integral_constant<int, QB_OFF>;
  using UM_ = std::integral_constant<int, QB_UM>;
  using MK_ = std::integral_constant<int, QB_MASK>;
  // block qb: tiles [0, r0/64) need no mask, tile r0/64 holds its diagonal, later tiles are above it; rows beyond T compute on zeros
  const int a0 = min(jt_hi, r0[0] / KT), e0 = min(jt_hi, r0[0] / KT + 1);
  const int a1 = min(jt_hi, r0[1] / KT), e1 = min(jt_hi, r0[1] / KT + 1);
  if (HAS_DOC) {
    run(e0, MK_{}, MK_{}, true);
    run(e1, OFF_{}, MK_{}, true);
  } else {
    run(a0, UM_{}, UM_{}, true);
    run(e0, MK_{}, UM_{}, true);
    run(a1, OFF_{}, UM_{}, true);
    run(e1, OFF_{}, MK_{}, true);
  }
  run(jt_hi, OFF_{}, OFF_{}, false);

#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = r0[qb] + l31;
    float l_lo, l_hi;
    half_pair(lsum[qb], l_lo, l_hi);
    const float ltot = l_lo + l_hi;
    if (qrow < T) {
      const float inv = 1.f / ltot;
      uint16_t* op = out + ((int64_t)b * T + qrow) * dm + h * HD;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4_t v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = f2bf(o[qb][db][4 * g + e] * inv);
          st_bf16x4(op + db * 32 + 8 * g + 4 * hi, v);
        }
      if (hi == 0) lse[((int64_t)b * nh + h) * T + qrow] = mc[qb] + __builtin_amdgcn_logf(ltot);  // base-2 LSE: the backward's exp2 argument directly
    }
  }
}

// =============================================================================================
// forward, persistent form.  At T = 1024 / head_dim 64 the forward pass moves 200 MB through HBM (q, k, v in, o out: ~37 us at the
// achievable rate) for 23 us of MFMA work at peak: a workgroup that loads its Q rows, computes, then stores O leaves the memory system
// idle while it computes and the matrix pipe idle while it loads and stores (measured on the non-persistent kernel above: 31 us of
// the 87 are the Q loads + O stores of the workgroups, 15 us the K / V DMA).  Here 2 workgroups per CU stay resident and walk a list of
// (256-query tile, head, batch) items, heaviest first; EVERYTHING an item reads arrives through one LDS ring that never drains:
//   step 0, 1 of an item: its Q rows (2 x 128 rows = 2 x 16 KiB), steps 2..: its K | V tiles of 64 keys (16 KiB each),
// issued NST - 1 steps ahead of their consumption, across item boundaries - the next item's Q and first K / V tiles are in flight while
// the current item finishes.  O leaves through a per-wave LDS transposition as whole 128-byte rows (16-byte stores, 8 lanes per row)
// and nobody waits for those stores.  Wave w owns the 32-row blocks w and 7 - w of the tile (equal causal work for every wave).
// =============================================================================================
struct Fwd4Item {
  int q0, b, h, jt_lo, jt_hi;
  bool valid;
};

template <bool HAS_DOC, int NST, int ABL = 0>
__global__ __launch_bounds__(256, 2) void attn_fwd4_kernel(const uint16_t* __restrict__ qkv, const int32_t* __restrict__ doc_start,
                                                           uint16_t* __restrict__ out, float* __restrict__ lse, int T, int nh, int nbh, int n_items,
                                                           unsigned* __restrict__ sched) {
  constexpr int KT = 64;
  constexpr int TILE = KT * 128;  // 8 KiB
  constexpr int QB = 256;
  constexpr int OST = 4096;       // per-wave O staging: 32 rows x 128 bytes
  __shared__ __attribute__((aligned(1024))) char smem[NST * 2 * TILE + 4 * OST + (HAS_DOC ? 2048 : 0)];
  __shared__ int ids[4];  // ids[k & 3]: list index of this workgroup's k-th item (k >= 1), fetched two items ahead by wave 0
  char* const ostage = smem + NST * 2 * TILE;
  int* const dstage = reinterpret_cast<int*>(smem + NST * 2 * TILE + 4 * OST);  // [2][256] doc starts of the current / next item's rows

  const int dm = nh * HD, ld = 3 * dm;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int ntile = (T + QB - 1) / QB;
  const int G = gridDim.x, g = blockIdx.x;
  const float c2 = 0.125f * LOG2E;  // 1/sqrt(64) and the base-2 exponent in one factor

  // The item list is sorted heaviest (latest query tile) first.  Workgroup g starts with item g; every further item is the next one
  // nobody has taken (one device-wide counter, bumped by wave 0 two items ahead of the consumer and handed to the other waves through
  // LDS): whichever workgroup is free takes the next-heaviest item, so all of them end within one light item of each other.  (A static
  // deal of the 1536 items of the 160M shape left the average SIMD 1.46 of 2 waves busy.)
  auto fetch_id = [&](int k) {  // wave 0 only; the compiler's vmcnt(0) in front of the LDS write drains this wave's DMA once per item
    const int v = (lane == 0) ? (int)atomicAdd(sched, 1u) : 0;
    if (lane == 0) ids[k & 3] = min(G + v, n_items);
  };
  auto load_item = [&](int k) {
    Fwd4Item it;
    const int idx = (k == 0) ? g : __builtin_amdgcn_readfirstlane(ids[k & 3]);
    it.valid = idx < n_items;
    const int rank = it.valid ? idx / nbh : 0, bh = it.valid ? idx % nbh : 0;
    it.q0 = (ntile - 1 - rank) * QB;
    it.h = bh % nh;
    it.b = bh / nh;
    it.jt_hi = (min(T, it.q0 + QB) + KT - 1) / KT;
    it.jt_lo = 0;
    if (HAS_DOC && it.valid) it.jt_lo = doc_start[(int64_t)it.b * T + it.q0] / KT;  // wave-uniform address: a scalar load
    return it;
  };

  if (wave == 0) fetch_id(1);
  __syncthreads();
  TileDma dma;
  dma.init(wave, lane, ld);
  // ---- producer: step ps of item P goes into slot pslot (4 LDS-DMA instructions per wave and step) ----
  int pk = 0, ps = 0, pslot = 0, inflight = 0;  // inflight: steps issued and not yet waited for
  Fwd4Item P = load_item(0);
  auto produce = [&]() {
    if (!P.valid) return;
    const uint16_t* base = (ABL & 128) ? qkv : qkv + (int64_t)P.b * T * ld + P.h * HD;  // 128: every step from the same (cache-resident) rows
    char* dst = smem + pslot * 2 * TILE;
    if (!(ABL & 16)) {
      int row0, row1, col;
      if (ps < 2) {  // Q rows [128 ps, 128 ps + 128) of the tile
        row0 = (ABL & 128) ? 0 : P.q0 + 128 * ps;
        row1 = row0 + 64;
        col = 0;
      } else {       // K | V tile
        row0 = row1 = (ABL & 128) ? 0 : (P.jt_lo + ps - 2) * KT;
        col = dm;
      }
      const uint16_t* s0 = base + (int64_t)row0 * ld + col;
      const uint16_t* s1 = base + (int64_t)row1 * ld + (ps < 2 ? 0 : 2 * dm);
      if (row1 + KT <= T) {
        dma.issue_full(dst, s0, wave);
        dma.issue_full(dst + TILE, s1, wave);
      } else {  // rows beyond T are clamped to the last row (their products are masked or never stored)
        dma.issue(dst, base + col, ld, T - 1, wave, row0);
        dma.issue(dst + TILE, base + (ps < 2 ? 0 : 2 * dm), ld, T - 1, wave, row1);
      }
      if (HAS_DOC && ps == 0 && wave == 0) {  // the item's 256 doc starts (one 1 KiB piece; T % 4 == 0 is checked on the host)
        const int q = min(P.q0 + lane * 4, T - 4);
        dma16_asm(doc_start + (int64_t)P.b * T + q, reinterpret_cast<char*>(dstage + (pk & 1) * 256));
      }
    }
    ++inflight;
    pslot = (pslot + 1 == NST) ? 0 : pslot + 1;
    const int ns = 2 + P.jt_hi - P.jt_lo;
    if (++ps == ns) {
      ps = 0;
      P = load_item(++pk);
    }
  };
#pragma unroll
  for (int i = 0; i < NST - 1; ++i) produce();

  // ---- consumer ----
  int cslot = 0, store_credit = 0;
  auto sync_step = [&]() {
    // Wait for this wave's pieces of the oldest step in flight; the younger steps (at most NST - 2) stay in flight, and so do the 10
    // stores of the previous item's epilogue while they are younger than the awaited step (vmcnt counts loads and stores in issue order;
    // without the allowance every item would wait for its predecessor's O rows to reach memory).
    const int younger = inflight - 1;
    if (store_credit > 0) {
      --store_credit;
      if (NST >= 4 && younger >= 2) attn_wait_vm<18>();
      else if (NST >= 3 && younger == 1) attn_wait_vm<14>();
      else attn_wait_vm<10>();
    } else {
      if (NST >= 4 && younger >= 2) attn_wait_vm<8>();
      else if (NST >= 3 && younger == 1) attn_wait_vm<4>();
      else attn_wait_vm<0>();
    }
    --inflight;
    if (!(ABL & 32)) attn_barrier();  // everyone's pieces landed; and every wave is done with the previous step, whose slot is refilled now
    produce();
  };
  auto next_slot = [&]() { cslot = (cslot + 1 == NST) ? 0 : cslot + 1; };

  bf16x8_t qf[2][4];
  int dsq[2];
  f32x16_t o[2][2];
  float mc[2], lsum[2];  // running reference maximum in log2 units (s * c2), running sum
  int r0[2];             // first rows of this wave's two blocks (wave-uniform)

  // one row block's softmax + P V for one tile (s: its S^T accumulators, vfr: the tile's V fragments)
  auto soft_pv = [&](int qb_, f32x16_t (&s)[2], const bf16x8_t (&vfr)[2][4], int kv0, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const int qb = qb_;
    if (ABL & 64) {
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
          f32x4_t t4 = {s[sp >> 1][(sp & 1) * 8], s[sp >> 1][(sp & 1) * 8 + 1], s[sp >> 1][(sp & 1) * 8 + 2], s[sp >> 1][(sp & 1) * 8 + 3]};
          const bf16x8_t pq = __builtin_bit_cast(bf16x8_t, t4);
          if (ABL & 4)
            asm volatile("" : "+v"(o[qb][db]) : "v"(vfr[db][sp]), "v"(pq));
          else
            o[qb][db] = mfma32(vfr[db][sp], pq, o[qb][db]);
        }
      return;
    }
    // key (kb, r) of this lane is tile row kb*32 + (r&3) + 8*(r>>2) + 4*hi: visible iff  c_lo <= kb*32 + (r&3) + 8*(r>>2) <= c_hi
    const int c_hi = r0[qb] + l31 - kv0 - 4 * hi;
    const int c_lo = HAS_DOC ? dsq[qb] - kv0 - 4 * hi : 0;
    float tmax = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (MASK) {
          const int c = kb * 32 + (r & 3) + 8 * (r >> 2);
          const bool ok = (c <= c_hi) && (!HAS_DOC || c >= c_lo);
          if (!ok) s[kb][r] = -INFINITY;
        }
        tmax = fmaxf(tmax, s[kb][r]);
      }
    {
      float t_lo, t_hi;
      half_pair(tmax, t_lo, t_hi);
      tmax = fmaxf(t_lo, t_hi);
    }
    const float tm = tmax * c2;
    const bool need = tm > mc[qb] + ATTN_DEFER_LOG2;  // both -inf (nothing visible yet): false
    if (__builtin_amdgcn_ballot_w64(need) != 0ull) {   // wave-uniform and rare after the first tile
      const float mn = fmaxf(mc[qb], tm);
      const float alpha = fast_exp2(mc[qb] - ((mn == -INFINITY) ? 0.f : mn));
      mc[qb] = mn;
      lsum[qb] *= alpha;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[qb][db][r] *= alpha;
    }
    const float mref = (MASK && mc[qb] == -INFINITY) ? 0.f : mc[qb];
    float ps4[4] = {0.f, 0.f, 0.f, 0.f};
    bf16x8_t pf[4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float px = __builtin_fmaf(s[kb][r], c2, -mref);
        const float p = (ABL & 1) ? px : fast_exp2(px);
        ps4[r & 3] += p;
        pf[kb * 2 + (r >> 3)][r & 7] = f2bf(p);
      }
    lsum[qb] += (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        if (ABL & 4)
          asm volatile("" : "+v"(o[qb][db]) : "v"(vfr[db][sp]), "v"(pf[sp]));
        else
          o[qb][db] = mfma32(vfr[db][sp], pf[sp], o[qb][db]);
      }
  };

  // one KV tile: S^T = K Q^T for the active row blocks, then softmax + P V per block.  M0 / M1: mode of block 0 / 1.
  auto compute = [&](int kv0, const char* sK, const char* sV, auto m0_tag, auto m1_tag) {
    constexpr int M0 = decltype(m0_tag)::value, M1 = decltype(m1_tag)::value;
    bf16x8_t kfr[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        kfr[kb][ks] = (ABL & 8) ? qf[0][ks] : frag_rows(sK, kb * 32 + l31, ks, hi);
        if (ABL & 8) asm volatile("" : "+v"(kfr[kb][ks]));
      }
    f32x16_t s[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      if ((qb == 0 ? M0 : M1) == QB_OFF) continue;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        zero16(s[qb][kb]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ABL & 2)
            asm volatile("" : "+v"(s[qb][kb]) : "v"(kfr[kb][ks]));
          else
            s[qb][kb] = mfma32(kfr[kb][ks], qf[qb][ks], s[qb][kb]);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // 256 registers: keep the V fragments out of the Q K^T phase
    bf16x8_t vfr[2][4];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        vfr[db][sp] = (ABL & 8) ? qf[0][sp] : frag_cols(sV, db, (sp >> 1) * 32 + (sp & 1) * 16 + 4 * hi, lane);
        if (ABL & 8) asm volatile("" : "+v"(vfr[db][sp]));
      }
    if (M0 == QB_UM) soft_pv(0, s[0], vfr, kv0, std::false_type{});
    if (M0 == QB_MASK) soft_pv(0, s[0], vfr, kv0, std::true_type{});
    if (M1 == QB_UM) soft_pv(1, s[1], vfr, kv0, std::false_type{});
    if (M1 == QB_MASK) soft_pv(1, s[1], vfr, kv0, std::true_type{});
  };

  using OFF_ = std::integral_constant<int, QB_OFF>;
  using UM_ = std::integral_constant<int, QB_UM>;
  using MK_ = std::integral_constant<int, QB_MASK>;

  for (int ck = 0;; ++ck) {
    const Fwd4Item C = load_item(ck);
    if (!C.valid) break;
    r0[0] = C.q0 + 32 * wave;
    r0[1] = C.q0 + 32 * (7 - wave);
    if (wave == 0) {  // the id of item ck + 2 (ids[(ck + 1) & 3] was fetched during the previous item)
      if (ck == 0 || __builtin_amdgcn_readfirstlane(ids[(ck + 1) & 3]) < n_items) {
        store_credit = 0;  // the atomic's own wait drains everything
        fetch_id(ck + 2);
      } else if (lane == 0) {
        ids[(ck + 2) & 3] = n_items;
      }
    }
    // ---- steps 0, 1: this wave's Q blocks out of the ring (block bi of the tile: half bi / 4, 64-row tile (bi & 3) / 2, rows 32 (bi & 1)..) ----
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      sync_step();
      const int bi = (qb == 0 ? wave : 7 - wave) & 3;
      const char* tq = smem + cslot * 2 * TILE + (bi >> 1) * TILE;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = frag_rows(tq, (bi & 1) * 32 + l31, ks, hi);
      dsq[qb] = 0;
      if (HAS_DOC) dsq[qb] = dstage[(ck & 1) * 256 + (r0[qb] - C.q0) + l31];  // landed with step 0 (wave 0 waited for it, then the barrier)
      zero16(o[qb][0]);
      zero16(o[qb][1]);
      mc[qb] = -INFINITY;
      lsum[qb] = 0.f;
      next_slot();
    }
    // ---- K | V tiles ----
    int jt = C.jt_lo;
    auto run = [&](int jt_end, auto m0_tag, auto m1_tag, bool active) {
      for (; jt < jt_end; ++jt) {
        sync_step();
        if (active) compute(jt * KT, smem + cslot * 2 * TILE, smem + cslot * 2 * TILE + TILE, m0_tag, m1_tag);
        next_slot();
      }
    };
    // block qb: tiles [0, r0/64) need no mask, tile r0/64 holds its diagonal, later tiles are above it; rows beyond T compute on clamped rows
    const int a0 = min(C.jt_hi, r0[0] / KT), e0 = min(C.jt_hi, r0[0] / KT + 1);
    const int a1 = min(C.jt_hi, r0[1] / KT), e1 = min(C.jt_hi, r0[1] / KT + 1);
    if (HAS_DOC) {
      run(e0, MK_{}, MK_{}, true);
      run(e1, OFF_{}, MK_{}, true);
    } else {
      run(a0, UM_{}, UM_{}, true);
      run(e0, MK_{}, UM_{}, true);
      run(a1, OFF_{}, UM_{}, true);
      run(e1, OFF_{}, MK_{}, true);
    }
    run(C.jt_hi, OFF_{}, OFF_{}, false);
    // ---- epilogue: O rows through the wave's LDS staging area as whole 128-byte rows; nobody waits for the stores ----
    char* const ost = ostage + wave * OST;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float l_lo, l_hi;
      half_pair(lsum[qb], l_lo, l_hi);
      const float ltot = l_lo + l_hi;
      const float inv = 1.f / ltot;
      // lane (l31, hi) holds row l31, head dims db*32 + 8g + 4hi + e: 8-byte piece `hi` of the 16-byte chunk c16 = 4 db + g, stored at
      // chunk position c16 ^ (row & 7) (spreads a column of chunks over the banks)
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          bf16x4_t v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = f2bf(o[qb][db][4 * gq + e] * inv);
          *reinterpret_cast<bf16x4_t*>(ost + l31 * 128 + (((db * 4 + gq) ^ (l31 & 7)) << 4) + hi * 8) = v;
        }
      const int qrow = r0[qb] + l31;
      if (ABL & 256) continue;
      if (qrow < T && hi == 0) lse[((int64_t)C.b * nh + C.h) * T + qrow] = mc[qb] + __builtin_amdgcn_logf(ltot);  // base-2 LSE
      // read back: lane L takes chunk L & 7 of rows L >> 3, + 8, + 16, + 24 (same wave wrote them: no barrier, the compiler's lgkmcnt orders it)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + (lane >> 3), c16 = lane & 7;
        const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(ost + row * 128 + ((c16 ^ (row & 7)) << 4));
        const int orow = r0[qb] + row;
        if (orow < T) st_bf16x8(out + ((int64_t)C.b * T + orow) * dm + C.h * HD + c16 * 8, v);
      }
    }
    store_credit = (C.q0 + QB <= T && !(ABL & 256)) ? NST - 1 : 0;  // all 10 store instructions were issued by every wave
  }
  // the last workgroup out resets the scheduler for the next launch (launches of this kernel on one device must not overlap)
  if (t == 0 && atomicAdd(sched + 1, 1u) == (unsigned)(G - 1)) {
    sched[0] = 0;
    sched[1] = 0;
  }
}

// =============================================================================================
// backward: dQ  (one workgroup = 4 waves x (32 * NQB) query rows; key tiles of 64 rows; q, k rotated)
// Also computes delta[q] = sum_d dO[q][d] O[q][d] for its rows and publishes it for the dK/dV kernel, which runs after it.
// =============================================================================================
template <bool HAS_DOC, int NQB, int MINW>
__global__ __launch_bounds__(256, MINW) void attn_bwd_dq2_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out,
                                                                 const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                                                                 float* __restrict__ delta, const float* __restrict__ rcos,
                                                                 const float* __restrict__ rsin, const int32_t* __restrict__ doc_start,
                                                                 uint16_t* __restrict__ dqkv, int T, int nh) {
  constexpr int KT = 64;
  constexpr int TILE = KT * 128;
  constexpr int QW = 32 * NQB, QB = 4 * QW;
  __shared__ __attribute__((aligned(1024))) char smem[2 * 2 * TILE];  // [stage][K|V]

  int tile_, h, b;
  attn_block2<QB>(T, nh, tile_, h, b);
  const int qt = (T + QB - 1) / QB - 1 - tile_;
  const int dm = nh * HD, ld = 3 * dm;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int q0 = qt * QB, qw0 = q0 + wave * QW;
  const uint16_t* base = qkv + (int64_t)b * T * ld + h * HD;
  const float scale = 0.125f, c2 = scale * LOG2E;

  bf16x8_t qf[NQB][4], dof[NQB][4];
  float Lq[NQB], Dq[NQB];
  int dsq[NQB];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    const int qrow = qw0 + qb * 32 + l31;
    const bool qvalid = qrow < T;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int d0 = ks * 16 + hi * 8;
      qf[qb][ks] = qvalid ? ld_bf16x8(base + (int64_t)qrow * ld + d0) : zero_bf16x8();
      dof[qb][ks] = qvalid ? ld_bf16x8(dout + ((int64_t)b * T + qrow) * dm + h * HD + d0) : zero_bf16x8();
    }
    Lq[qb] = 0.f;
    dsq[qb] = 0;
    float part = 0.f;
    if (qvalid) {
      Lq[qb] = lse[((int64_t)b * nh + h) * T + qrow];  // base-2 LSE
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8_t o8 = ld_bf16x8(out + ((int64_t)b * T + qrow) * dm + h * HD + ks * 16 + hi * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) part += bf2f(o8[e]) * bf2f(dof[qb][ks][e]);
      }
      if (HAS_DOC) dsq[qb] = doc_start[(int64_t)b * T + qrow];
    }
    float d_lo, d_hi;
    half_pair(part, d_lo, d_hi);  // rows beyond T hold zeros in both halves
    Dq[qb] = d_lo + d_hi;
    if (qvalid && hi == 0) delta[((int64_t)b * nh + h) * T + qrow] = Dq[qb];
  }
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb)
    asm volatile("; q/dO fragments resident" ::"v"(qf[qb][0]), "v"(qf[qb][1]), "v"(qf[qb][2]), "v"(qf[qb][3]), "v"(dof[qb][0]), "v"(dof[qb][1]),
                 "v"(dof[qb][2]), "v"(dof[qb][3]), "v"(Lq[qb]), "v"(Dq[qb]), "v"(dsq[qb]));

  f32x16_t dq[NQB][2];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    zero16(dq[qb][0]);
    zero16(dq[qb][1]);
  }

  const int kv_hi = min(T, q0 + QB);
  const int jt_hi = (kv_hi + KT - 1) / KT;
  int jt_lo = 0;
  if (HAS_DOC) jt_lo = __builtin_amdgcn_readfirstlane(doc_start[(int64_t)b * T + q0]) / KT;

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

  auto compute = [&](int kv0, const char* sK, const char* sV, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (MINW > 1) __builtin_amdgcn_sched_barrier(0);  // 256 registers: one key block's fragments at a time
      bf16x8_t kfr[4], vfr[4], ktr[2][2];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        kfr[ks] = frag_rows(sK, kb * 32 + l31, ks, hi);
        vfr[ks] = frag_rows(sV, kb * 32 + l31, ks, hi);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int db = 0; db < 2; ++db) ktr[db][s2] = frag_cols(sK, db, kb * 32 + s2 * 16 + 4 * hi, lane);
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) {
        // key r of this lane is tile row kb*32 + (r&3) + 8*(r>>2) + 4*hi: visible iff  c_lo <= kb*32 + (r&3) + 8*(r>>2) <= c_hi
        const int c_hi = qw0 + qb * 32 + l31 - kv0 - 4 * hi;
        const int c_lo = HAS_DOC ? dsq[qb] - kv0 - 4 * hi : 0;
        f32x16_t s, dp;
        zero16(s);
        zero16(dp);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s = mfma32(kfr[ks], qf[qb][ks], s);      // S^T[kv][q]
          dp = mfma32(vfr[ks], dof[qb][ks], dp);   // dP^T[kv][q]
        }
        bf16x8_t dsf[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float p = fast_exp2(__builtin_fmaf(s[r], c2, -Lq[qb]));
          if (MASK) {
            const int c = kb * 32 + (r & 3) + 8 * (r >> 2);
            const bool ok = (c <= c_hi) && (!HAS_DOC || c >= c_lo);
            p = ok ? p : 0.f;
          }
          dsf[r >> 3][r & 7] = f2bf(p * (dp[r] - Dq[qb]));  // x 1/sqrt(hd) once, in the epilogue
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int db = 0; db < 2; ++db) dq[qb][db] = mfma32(ktr[db][s2], dsf[s2], dq[qb][db]);  // dQ^T[d][q]
      }
    }
  };

  // three loops per wave (no mask / masked / idle), as in the forward kernel
  const bool wave_rows = qw0 < T;
  const int jt_act = wave_rows ? min(jt_hi, (qw0 + QW - 1) / KT + 1) : jt_lo;
  const int jt_um = HAS_DOC ? jt_lo : min(jt_act, max(jt_lo, (qw0 + 1) / KT));
  if (jt_lo < jt_hi) stage(0, jt_lo);
  attn_wait_vm<0>();
  attn_barrier();
  int st = 0, jt = jt_lo;
  auto run = [&](int jt_end, auto mask_tag, bool active) {
    for (; jt < jt_end; ++jt, st ^= 1) {
      if (jt + 1 < jt_hi) stage(st ^ 1, jt + 1);
      if (active) compute(jt * KT, smem + st * 2 * TILE, smem + st * 2 * TILE + TILE, mask_tag);
      attn_wait_vm<0>();
      attn_barrier();
    }
  };
  run(jt_um, std::false_type{}, true);
  run(jt_act, std::true_type{}, true);
  run(jt_hi, std::true_type{}, false);

#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    const int qrow = qw0 + qb * 32 + l31;
    if (qrow < T) {
      uint16_t* dqp = dqkv + ((int64_t)b * T + qrow) * ld + h * HD;
      const int trow = qrow * 32;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d0 = db * 32 + 8 * g + 4 * hi;
          const float c0 = rcos[trow + d0 / 2], c1 = rcos[trow + d0 / 2 + 1];
          const float s0 = rsin[trow + d0 / 2], s1 = rsin[trow + d0 / 2 + 1];
          const float a0 = dq[qb][db][4 * g + 0] * scale, b0 = dq[qb][db][4 * g + 1] * scale, a1 = dq[qb][db][4 * g + 2] * scale,
                      b1 = dq[qb][db][4 * g + 3] * scale;
          bf16x4_t ov;  // inverse rotation: gradient w.r.t. the PRE-rotation q
          ov[0] = f2bf(a0 * c0 + b0 * s0);
          ov[1] = f2bf(b0 * c0 - a0 * s0);
          ov[2] = f2bf(a1 * c1 + b1 * s1);
          ov[3] = f2bf(b1 * c1 - a1 * s1);
          st_bf16x4(dqp + d0, ov);
        }
    }
  }
}

// =============================================================================================
// backward: dK, dV  (one workgroup = 4 waves x (32 * NKB) key rows; query tiles of 64 rows; q, k rotated)
// K / V of a wave's rows stay in registers as B operands; Q / dO tiles go through LDS and are read once as row fragments (S, dP) and
// once transposed (dV, dK) per 32-query block - for NKB key blocks at a time.
// =============================================================================================
template <bool HAS_DOC, int NKB, int MINW>
__global__ __launch_bounds__(256, MINW) void attn_bwd_dkdv2_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout,
                                                                   const float* __restrict__ lse, const float* __restrict__ delta,
                                                                   const float* __restrict__ rcos, const float* __restrict__ rsin,
                                                                   const int32_t* __restrict__ doc_start, uint16_t* __restrict__ dqkv,
                                                                   int T, int nh) {
  constexpr int QT = 64;
  constexpr int TILE = QT * 128;          // 8 KiB
  constexpr int STAGE = 2 * TILE + 1024;  // Q | dO | statistics (lse[64], delta[64], doc_start[64])
  constexpr int KW = 32 * NKB, KB = 4 * KW;
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];

  int kt, h, b;  // key tile 0 meets every query tile: heaviest first
  attn_block2<KB>(T, nh, kt, h, b);
  const int dm = nh * HD, ld = 3 * dm;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int kv0 = kt * KB, kvw0 = kv0 + wave * KW;
  const uint16_t* base = qkv + (int64_t)b * T * ld + h * HD;
  const uint16_t* dobase = dout + (int64_t)b * T * dm + h * HD;
  const float* lrow = lse + ((int64_t)b * nh + h) * T;
  const float* drow = delta + ((int64_t)b * nh + h) * T;
  const int32_t* dsrow = doc_start + (HAS_DOC ? (int64_t)b * T : 0);
  const float scale = 0.125f, c2 = scale * LOG2E;

  bf16x8_t kf[NKB][4], vf[NKB][4];
#pragma unroll
  for (int kvb = 0; kvb < NKB; ++kvb) {
    const int kvrow = kvw0 + kvb * 32 + l31;
    const bool kvalid = kvrow < T;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const uint16_t* p = base + (int64_t)kvrow * ld + ks * 16 + hi * 8;
      kf[kvb][ks] = kvalid ? ld_bf16x8(p + dm) : zero_bf16x8();
      vf[kvb][ks] = kvalid ? ld_bf16x8(p + 2 * dm) : zero_bf16x8();
    }
  }
  // query-tile range: from the block's diagonal down; with document masks stop once a tile's first row starts after this key block
  const int nqt = (T + QT - 1) / QT;
  const int jq_lo = kv0 / QT;
  int jq_hi = nqt;
  if (HAS_DOC) {
    jq_hi = jq_lo;
    while (jq_hi < nqt && __builtin_amdgcn_readfirstlane(dsrow[jq_hi * QT]) <= kv0 + KB - 1) ++jq_hi;
  }
#pragma unroll
  for (int kvb = 0; kvb < NKB; ++kvb)
    asm volatile("; k/v fragments resident" ::"v"(kf[kvb][0]), "v"(kf[kvb][1]), "v"(kf[kvb][2]), "v"(kf[kvb][3]), "v"(vf[kvb][0]),
                 "v"(vf[kvb][1]), "v"(vf[kvb][2]), "v"(vf[kvb][3]));  // every ordinary load is consumed before the first DMA is in flight

  f32x16_t dk[NKB][2], dv[NKB][2];
#pragma unroll
  for (int kvb = 0; kvb < NKB; ++kvb) {
    zero16(dk[kvb][0]); zero16(dk[kvb][1]); zero16(dv[kvb][0]); zero16(dv[kvb][1]);
  }

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
      if (HAS_DOC) dma16_asm(dsrow + q, dst + 2 * TILE + 512);
    }
  };

  auto compute = [&](int qt0, const char* sQ, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const char* sDO = sQ + TILE;
    const float* sL = reinterpret_cast<const float*>(sQ + 2 * TILE);
    const float* sD = sL + 64;
    const int* sDS = reinterpret_cast<const int*>(sL + 128);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      __builtin_amdgcn_sched_barrier(0);  // one query block's fragments at a time
      bf16x8_t qfr[4], dofr[4], dotr[2][2], qtr[2][2];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        qfr[ks] = frag_rows(sQ, qb * 32 + l31, ks, hi);
        dofr[ks] = frag_rows(sDO, qb * 32 + l31, ks, hi);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dotr[db][s2] = frag_cols(sDO, db, qb * 32 + s2 * 16 + 4 * hi, lane);
          qtr[db][s2] = frag_cols(sQ, db, qb * 32 + s2 * 16 + 4 * hi, lane);
        }
      // this lane's query rows for registers 4g..4g+3 are consecutive: one 16-byte read per statistic
      f32x4_t L4[4], D4[4];
      int ds4[4][4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ql0 = qb * 32 + 8 * g + 4 * hi;
        L4[g] = *reinterpret_cast<const f32x4_t*>(sL + ql0);  // base-2 LSE
        D4[g] = *reinterpret_cast<const f32x4_t*>(sD + ql0);
        if (MASK && HAS_DOC) {
#pragma unroll
          for (int e = 0; e < 4; ++e) ds4[g][e] = sDS[ql0 + e];
        }
      }
#pragma unroll
      for (int kvb = 0; kvb < NKB; ++kvb) {
        const int kvrow = kvw0 + kvb * 32 + l31;
        // query r = 4g + e of this lane is tile row qb*32 + 8g + e + 4*hi: visible iff  c_lo <= qb*32 + 8g + e < c_end
        const int c_lo = kvrow - qt0 - 4 * hi, c_end = T - qt0 - 4 * hi;
        f32x16_t s, dp;
        zero16(s);
        zero16(dp);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s = mfma32(qfr[ks], kf[kvb][ks], s);       // S[q][kv]
          dp = mfma32(dofr[ks], vf[kvb][ks], dp);    // dP[q][kv]
        }
        bf16x8_t pf[2], dsf[2];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            float p = fast_exp2(__builtin_fmaf(s[r], c2, -L4[g][e]));
            if (MASK) {
              const int c = qb * 32 + 8 * g + e;
              bool ok = (c >= c_lo) && (c < c_end);
              if (HAS_DOC) ok = ok && (kvrow >= ds4[g][e]);
              p = ok ? p : 0.f;
            }
            const float dsv = p * (dp[r] - D4[g][e]);  // the 1/sqrt(hd) factor (a power of two: exact) is applied once, to dK, in the epilogue
            pf[r >> 3][r & 7] = f2bf(p);
            dsf[r >> 3][r & 7] = f2bf(dsv);
          }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            dv[kvb][db] = mfma32(dotr[db][s2], pf[s2], dv[kvb][db]);   // dV^T[d][kv]
            dk[kvb][db] = mfma32(qtr[db][s2], dsf[s2], dk[kvb][db]);   // dK^T[d][kv]
          }
      }
    }
  };

  // Every wave walks all query tiles of the workgroup (DMA issue, one barrier per tile) in four consecutive loops of its own: tiles
  // entirely above its first key (idle), the tiles its diagonal crosses (masked), the tiles entirely below its last key (no mask), and a
  // partial last tile when T % 64 != 0 (masked).  With document masks every active tile takes the masked body.
  const bool wave_rows = kvw0 < T;
  const int jq_act = wave_rows ? min(jq_hi, max(jq_lo, kvw0 / QT)) : jq_hi;             // first tile with a query at or below this wave's first key
  const int jq_m = HAS_DOC ? jq_hi : min(jq_hi, max(jq_act, (kvw0 + KW - 1 + QT - 1) / QT));  // first tile whose every query sees every key of the wave
  const int jq_u = HAS_DOC ? jq_hi : min(jq_hi, max(jq_m, T / QT));                     // first partial tile
  if (jq_lo < jq_hi) stage(0, jq_lo);
  attn_wait_vm<0>();
  attn_barrier();
  int st = 0, jq = jq_lo;
  auto run = [&](int jq_end, auto mask_tag, bool active) {
    for (; jq < jq_end; ++jq, st ^= 1) {
      if (jq + 1 < jq_hi) stage(st ^ 1, jq + 1);
      if (active) compute(jq * QT, smem + st * STAGE, mask_tag);
      attn_wait_vm<0>();
      attn_barrier();
    }
  };
  run(jq_act, std::true_type{}, false);
  run(jq_m, std::true_type{}, true);
  run(jq_u, std::false_type{}, true);
  run(jq_hi, std::true_type{}, true);

#pragma unroll
  for (int kvb = 0; kvb < NKB; ++kvb) {
    const int kvrow = kvw0 + kvb * 32 + l31;
    if (kvrow < T) {
      uint16_t* dkp = dqkv + ((int64_t)b * T + kvrow) * ld + dm + h * HD;
      uint16_t* dvp = dkp + dm;
      const int trow = kvrow * 32;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d0 = db * 32 + 8 * g + 4 * hi;
          bf16x4_t ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov[e] = f2bf(dv[kvb][db][4 * g + e]);
          st_bf16x4(dvp + d0, ov);
          // inverse rotation of the two (even, odd) pairs of dK: gradient w.r.t. the PRE-rotation k
          const float c0 = rcos[trow + d0 / 2], c1 = rcos[trow + d0 / 2 + 1];
          const float s0 = rsin[trow + d0 / 2], s1 = rsin[trow + d0 / 2 + 1];
          const float a0 = dk[kvb][db][4 * g + 0] * scale, b0 = dk[kvb][db][4 * g + 1] * scale, a1 = dk[kvb][db][4 * g + 2] * scale,
                      b1 = dk[kvb][db][4 * g + 3] * scale;
          bf16x4_t ok;
          ok[0] = f2bf(a0 * c0 + b0 * s0);
          ok[1] = f2bf(b0 * c0 - a0 * s0);
          ok[2] = f2bf(a1 * c1 + b1 * s1);
          ok[3] = f2bf(b1 * c1 - a1 * s1);
          st_bf16x4(dkp + d0, ok);
        }
    }
  }
}

// =============================================================================================
// launchers (called from the C ABI entry points in attn.hip)
// =============================================================================================
template <int NB, int MINW, int ABL = 0>
static void launch_fwd2(const uint16_t* qkv, const int32_t* doc_start, uint16_t* out, float* lse, int64_t B, int64_t T, int64_t nh, hipStream_t s) {
  const dim3 grid((unsigned)(plm_cdiv(T, 128 * NB) * nh * B)), block(256);
  if (doc_start && ABL == 0)
    hipLaunchKernelGGL((attn_fwd2_kernel<true, NB, MINW, 0>), grid, block, 0, s, qkv, doc_start, out, lse, (int)T, (int)nh);
  else
    hipLaunchKernelGGL((attn_fwd2_kernel<false, NB, MINW, ABL>), grid, block, 0, s, qkv, doc_start, out, lse, (int)T, (int)nh);
}
template <int NST, int MINW, int ABL = 0>
static void launch_fwd3(const uint16_t* qkv, const int32_t* doc_start, uint16_t* out, float* lse, int64_t B, int64_t T, int64_t nh, hipStream_t s) {
  const dim3 grid((unsigned)(plm_cdiv(T, 256) * nh * B)), block(256);
  if (doc_start && ABL == 0)
    hipLaunchKernelGGL((attn_fwd3_kernel<true, NST, MINW, 0>), grid, block, 0, s, qkv, doc_start, out, lse, (int)T, (int)nh);
  else
    hipLaunchKernelGGL((attn_fwd3_kernel<false, NST, MINW, ABL>), grid, block, 0, s, qkv, doc_start, out, lse, (int)T, (int)nh);
}
static int attn_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return n;
}
__device__ unsigned g_attn_sched[8];  // {next item, finished workgroups} per kernel family: forward, dQ, dK/dV
static unsigned* attn_sched(int family) {
  static unsigned* base = nullptr;
  if (!base && hipGetSymbolAddress(reinterpret_cast<void**>(&base), HIP_SYMBOL(g_attn_sched)) != hipSuccess) base = nullptr;
  return base + 2 * family;
}
template <int NST, int ABL = 0>
static void launch_fwd4(const uint16_t* qkv, const int32_t* doc_start, uint16_t* out, float* lse, int64_t B, int64_t T, int64_t nh, hipStream_t s) {
  const int n_items = (int)(plm_cdiv(T, 256) * nh * B);
  const int G = n_items < 2 * attn_num_cus() ? n_items : 2 * attn_num_cus();
  const dim3 grid((unsigned)G), block(256);
  if (doc_start && ABL == 0)
    hipLaunchKernelGGL((attn_fwd4_kernel<true, NST, 0>), grid, block, 0, s, qkv, doc_start, out, lse, (int)T, (int)nh, (int)(nh * B), n_items, attn_sched(0));
  else
    hipLaunchKernelGGL((attn_fwd4_kernel<false, NST, ABL>), grid, block, 0, s, qkv, doc_start, out, lse, (int)T, (int)nh, (int)(nh * B), n_items, attn_sched(0));
}
template <int NB, int MINW>
static void launch_dq2(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* lse, float* delta, const float* rc,
                       const float* rs, const int32_t* doc_start, uint16_t* dqkv, int64_t B, int64_t T, int64_t nh, hipStream_t s) {
  const dim3 grid((unsigned)(plm_cdiv(T, 128 * NB) * nh * B)), block(256);
  if (doc_start)
    hipLaunchKernelGGL((attn_bwd_dq2_kernel<true, NB, MINW>), grid, block, 0, s, qkv, out, dout, lse, delta, rc, rs, doc_start, dqkv, (int)T, (int)nh);
  else
    hipLaunchKernelGGL((attn_bwd_dq2_kernel<false, NB, MINW>), grid, block, 0, s, qkv, out, dout, lse, delta, rc, rs, doc_start, dqkv, (int)T, (int)nh);
}
template <int NB, int MINW>
static void launch_dkdv2(const uint16_t* qkv, const uint16_t* dout, const float* lse, const float* delta, const float* rc, const float* rs,
                         const int32_t* doc_start, uint16_t* dqkv, int64_t B, int64_t T, int64_t nh, hipStream_t s) {
  const dim3 grid((unsigned)(plm_cdiv(T, 128 * NB) * nh * B)), block(256);
  if (doc_start)
    hipLaunchKernelGGL((attn_bwd_dkdv2_kernel<true, NB, MINW>), grid, block, 0, s, qkv, dout, lse, delta, rc, rs, doc_start, dqkv, (int)T, (int)nh);
  else
    hipLaunchKernelGGL((attn_bwd_dkdv2_kernel<false, NB, MINW>), grid, block, 0, s, qkv, dout, lse, delta, rc, rs, doc_start, dqkv, (int)T, (int)nh);
}

// variant codes (PLM_ATTN_FWD / PLM_ATTN_DQ / PLM_ATTN_DKDV, A/B runs): 21 = 64 rows per wave at 1 wave per SIMD, 22 = at 2 waves per
// SIMD, 12 = 32 rows per wave at 2 waves per SIMD
void plm_attn_fwd2(int variant, const uint16_t* qkv, const int32_t* doc_start, uint16_t* out, float* lse, int64_t B, int64_t T, int64_t nh,
                   hipStream_t s) {
  switch (variant) {
    case 21: launch_fwd2<2, 1>(qkv, doc_start, out, lse, B, T, nh, s); break;
    case 12: launch_fwd2<1, 2>(qkv, doc_start, out, lse, B, T, nh, s); break;
    case 43: launch_fwd4<3>(qkv, doc_start, out, lse, B, T, nh, s); break;
    case 44: launch_fwd4<4>(qkv, doc_start, out, lse, B, T, nh, s); break;
    case 3243: launch_fwd4<3, 32>(qkv, doc_start, out, lse, B, T, nh, s); break;
    case 1643: launch_fwd4<3, 16>(qkv, doc_start, out, lse, B, T, nh, s); break;
    case 6443: launch_fwd4<3, 64>(qkv, doc_start, out, lse, B, T, nh, s); break;
    case 12643: launch_fwd4<3, 126>(qkv, doc_start, out, lse, B, T, nh, s); break;
    case 12843: launch_fwd4<3, 128>(qkv, doc_start, out, lse, B, T, nh, s); break;
    case 25643: launch_fwd4<3, 256>(qkv, doc_start, out, lse, B, T, nh, s); break;
    case 38443: launch_fwd4<3, 384>(qkv, doc_start, out, lse, B, T, nh, s); break;
    case 32: launch_fwd3<2, 2>(qkv, doc_start, out, lse, B, T, nh, s); break;
    case 33: launch_fwd3<3, 2>(qkv, doc_start, out, lse, B, T, nh, s); break;
    case 34: launch_fwd3<4, 2>(qkv, doc_start, out, lse, B, T, nh, s); break;
    case 3234: launch_fwd3<4, 2, 32>(qkv, doc_start, out, lse, B, T, nh, s); break;
    case 1634: launch_fwd3<4, 2, 16>(qkv, doc_start, out, lse, B, T, nh, s); break;
    case 6434: launch_fwd3<4, 2, 64>(qkv, doc_start, out, lse, B, T, nh, s); break;
    case 12634: launch_fwd3<4, 2, 126>(qkv, doc_start, out, lse, B, T, nh, s); break;
#define ABLV(a)                                                                         \
  case 22 + 100 * a: launch_fwd2<2, 2, a>(qkv, doc_start, out, lse, B, T, nh, s); break; \
  case 12 + 100 * a: launch_fwd2<1, 2, a>(qkv, doc_start, out, lse, B, T, nh, s); break;
      ABLV(1) ABLV(6) ABLV(8) ABLV(16) ABLV(24) ABLV(32) ABLV(64) ABLV(65) ABLV(70) ABLV(78) ABLV(126)
#undef ABLV
    default: launch_fwd2<2, 2>(qkv, doc_start, out, lse, B, T, nh, s); break;
  }
}
void plm_attn_dq2(int variant, const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* lse, float* delta, const float* rc,
                  const float* rs, const int32_t* doc_start, uint16_t* dqkv, int64_t B, int64_t T, int64_t nh, hipStream_t s) {
  switch (variant) {
    case 21: launch_dq2<2, 1>(qkv, out, dout, lse, delta, rc, rs, doc_start, dqkv, B, T, nh, s); break;
    case 12: launch_dq2<1, 2>(qkv, out, dout, lse, delta, rc, rs, doc_start, dqkv, B, T, nh, s); break;
    default: launch_dq2<2, 2>(qkv, out, dout, lse, delta, rc, rs, doc_start, dqkv, B, T, nh, s); break;
  }
}
void plm_attn_dkdv2(int variant, const uint16_t* qkv, const uint16_t* dout, const float* lse, const float* delta, const float* rc,
                    const float* rs, const int32_t* doc_start, uint16_t* dqkv, int64_t B, int64_t T, int64_t nh, hipStream_t s) {
  switch (variant) {
    case 22: launch_dkdv2<2, 2>(qkv, dout, lse, delta, rc, rs, doc_start, dqkv, B, T, nh, s); break;
    case 12: launch_dkdv2<1, 2>(qkv, dout, lse, delta, rc, rs, doc_start, dqkv, B, T, nh, s); break;
    default: launch_dkdv2<2, 1>(qkv, dout, lse, delta, rc, rs, doc_start, dqkv, B, T, nh, s); break;
  }
}
