// Causal attention (no document masks) for gfx950, second generation of kernels: forward, backward dQ, backward dK / dV - and, as its DOC mode,
// the dK / dV kernel of document-masked batches (forward / dQ of those: attn_doc.hip, which shares the LDS image, the LDS-DMA staging and the
// C ABI of attn.hip with this file); q, k are already rotated (RoPE lives in the w_qkv GEMM's epilogue).
//
// What round 3 measured on the first-generation kernels (profiles/r03_attn_ablation.txt, r03_ubench_mfma_gap.txt: timing-only ablations, SQ counters,
// tools/ubench/mfma_gap.hip) and what these kernels do about it:
//   * Nothing overlapped: removing any one of {softmax arithmetic, MFMAs, LDS fragment reads, K / V DMA, barriers, Q loads + O stores} from the
//     forward kernel saved that part's own time (35 / 21 / 16 / 33 / 15 / 30 of 106 us).  The waves are dependency chains  ds_read -> wait -> MFMA
//     -> softmax -> MFMA, and with 2-4 waves per SIMD whose MFMA and VALU instructions share one issue port the SIMD sits at ~50 % of its
//     MFMA + VALU time.  A wave here owns 64 rows (two 32-row blocks): every K / V (Q / dO) fragment read from LDS feeds two MFMAs, all
//     fragments of a tile are requested up front, and one block's softmax arithmetic is scheduled beside the other block's MFMAs.
//   * In a causal 256-row tile the waves of a workgroup had unequal work (rows 0-63 stop 3 key tiles before rows 192-255) and idled at the
//     per-tile barrier: wave w now owns the 32-row blocks w and 7 - w - every wave does the same number of 32 x 64 blocks (forward and dQ).
//   * K / V (Q / dO) tiles go through an NST-deep LDS ring filled NST - 1 tiles ahead with counted vmcnt waits instead of a double buffer.
//   * The forward softmax defers the running-max update (the O rescale) until a row's maximum has grown by more than 2^8: with the exact
//     maximum SOME row of a wave grows in almost every tile, so the "skip when nothing moved" test of generation one never skipped.
//   * Mask code (diagonal tiles only) lives in separate per-wave loops; masks are two integer thresholds per lane and tile.
//   * Output blocks (O, dQ, dK, dV) leave as whole 128-byte rows through a wave-private LDS transposition (RowStage, attn_common.h) instead
//     of 8-byte row-per-lane stores that touched 32 cache lines per instruction: backward 300 -> 273 us per layer - more than everything
//     above together.
// Measured (B = 32, T = 1024, 12 heads, same-box A/Bs): forward 95 -> 86 us, backward (dQ + dK/dV) 314 -> 273 us per layer.  What did NOT work is recorded in
// DESIGN.md section 5.3 (a persistent forward with one Q/K/V ring and a dynamic item queue; 64 key rows per wave at one wave per SIMD).
#include "plm_device.h"

#include <type_traits>

#include "attn_common.h"

// Block -> (row tile of RB rows, head, batch), heaviest (latest) tiles first: the hardware hands workgroups out in blockIdx order, so ALL blocks
// of the heaviest tile index come first, the lightest last (longest-processing-time-first).  (Head-major orders that let the
// tiles of a head meet in their XCD's L2 are slower - load balance is worth more than L2 hits: profiles/r04_attn_pingpong.txt section 3.)
template <int RB>
__device__ __forceinline__ void attn_block2(int T, int nh, int& tile, int& h, int& b) {
  const int ntile = (T + RB - 1) / RB;
  const int nbh = gridDim.x / ntile;
  const int bh = blockIdx.x % nbh;
  tile = blockIdx.x / nbh;
  h = bh % nh;
  b = bh / nh;
}

// LDS-DMA instructions one stage() issues per wave: a K | V (or Q | dO) tile pair = 2 x TileDma's two pieces.  The counted waits of the
// tile loops are multiples of it (the dK/dV kernel's wave 0 adds its two statistics pieces): a stage() that issued a different number, or
// ANY other vector-memory instruction between a kernel's prologue and its epilogue, would make those waits release tiles that have not
// landed - nothing else would notice.  Keep the loops free of loads and stores; change this constant with stage().
constexpr int ATTN_DMA_PER_STAGE = 4;
constexpr int ATTN_DMA_STATS = 2;  // dK/dV kernel, wave 0: the LSE and delta rows of the query tile
static_assert(ATTN_DMA_PER_STAGE == 2 * 2, "stage() = two TileDma::issue calls of two instructions each");

#define ATTN_DEFER_LOG2 8.0f
#define PLM_ATTN_TRACE_SETTER plm_dbg_attn_trace_causal
ATTN_TRACE_DECL()
#ifdef PLM_ATTN_ONEPASS_ABLATION  // tools/attn_onepass_ablation.py: never in the shipped library
static __device__ float* g_onepass_buf = nullptr;
static __device__ int g_onepass_mode = 0;  // 0 off | 1 dS through LDS + the 4 extra MFMAs | 2 + fp32 atomics of the partial dQ
extern "C" int plm_dbg_attn_onepass(float* buf, int mode) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_onepass_buf), &buf, sizeof(buf)) != hipSuccess) return -1;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_onepass_mode), &mode, sizeof(mode));
}
#define ONEPASS_LDS 8192
#else
#define ONEPASS_LDS 0
#endif  // the running maximum is updated when a row's new maximum exceeds it by more than 2^8 (P <= 256)

enum { QB_OFF = 0, QB_UM = 1, QB_MASK = 2 };  // a 32-row block on a key tile: above its diagonal / no mask needed / masked

// =============================================================================================
// forward: one workgroup = 4 waves x 64 query rows (256-row tiles), wave w owns the 32-row blocks w and 7 - w of the tile -
// every wave of a causal tile then has the same amount of work (block w ends 7 - 2w blocks before block 7 - w) and no wave idles through
// the diagonal region; K / V tiles of 64 rows go through an NST-deep LDS ring filled NST - 1 tiles ahead (counted vmcnt waits: the wait
// in front of tile i only covers tile i), one barrier per tile.
// Per tile a row block is OFF (tile above its diagonal), UM (no mask needed) or MASK.
// =============================================================================================
template <int NST>
__global__ __launch_bounds__(256, 2) void attn_fwd_causal_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out,
                                                                 float* __restrict__ lse, int T, int nh) {
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

  const int kv_hi = min(T, q0 + QB);
  const int jt_hi = (kv_hi + KT - 1) / KT;
  constexpr int jt_lo = 0;
  const int n = jt_hi;

  TileDma dma;
  dma.init(wave, lane, ld);
  auto stage = [&](int slot, int jt) {  // 4 LDS-DMA instructions per wave
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

  // ring: tile i = jt - jt_lo lives in slot i % NST; tiles are issued NST - 1 ahead.  Tile 0 goes out BEFORE the Q rows are asked for (round 6:
  // the two latencies overlap; the ordinary loads are younger than the DMA, so waiting for them covers it), the others right behind them
  if (n > 0) stage(0, jt_lo);

  bf16x8_t qf[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = r0[qb] + l31;
    const bool qvalid = qrow < T;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = qvalid ? ld_bf16x8(base + (int64_t)qrow * ld + ks * 16 + hi * 8) : zero_bf16x8();
  }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) asm volatile("; q fragments resident" ::"v"(qf[qb][0]), "v"(qf[qb][1]), "v"(qf[qb][2]), "v"(qf[qb][3]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every ordinary load has landed (and tile 0 with them): nothing but LDS-DMA is counted from here on
#pragma unroll
  for (int i = 1; i < NST - 1; ++i)
    if (i < n) stage(i, jt_lo + i);

  f32x16_t o[2][2];
  float mc[2], lsum[2];  // running reference maximum in log2 units (s * c2), running sum
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    zero16(o[qb][0]);
    zero16(o[qb][1]);
    mc[qb] = -INFINITY;
    lsum[qb] = 0.f;
  }

  // one row block's softmax + P V for one tile (s: its S^T accumulators, vfr: the tile's V fragments)
  auto soft_pv = [&](int qb_, f32x16_t (&s)[2], const bf16x8_t (&vfr)[2][4], int kv0, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const int qb = qb_;
    // key (kb, r) of this lane is tile row kb*32 + (r&3) + 8*(r>>2) + 4*hi: visible iff  kb*32 + (r&3) + 8*(r>>2) <= c_hi
    const int c_hi = r0[qb] + l31 - kv0 - 4 * hi;
    float tmax = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (MASK) {
          const int c = kb * 32 + (r & 3) + 8 * (r >> 2);
          if (c > c_hi) s[kb][r] = -INFINITY;
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
        const float p = fast_exp2(__builtin_fmaf(s[kb][r], c2, -mref));
        ps[r & 3] += p;
        pf[kb * 2 + (r >> 3)][r & 7] = f2bf(p);
      }
    lsum[qb] += (ps[0] + ps[1]) + (ps[2] + ps[3]);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) o[qb][db] = mfma32(vfr[db][sp], pf[sp], o[qb][db]);
  };

  // one KV tile: S^T = K Q^T for the active row blocks, then softmax + P V per block.  M0 / M1: mode of block 0 / 1.
  auto compute = [&](int kv0, const char* sK, const char* sV, auto m0_tag, auto m1_tag) {
    constexpr int M0 = decltype(m0_tag)::value, M1 = decltype(m1_tag)::value;
    bf16x8_t kfr[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kfr[kb][ks] = frag_rows(sK, kb * 32 + l31, ks, hi);
    f32x16_t s[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      if ((qb == 0 ? M0 : M1) == QB_OFF) continue;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        zero16(s[qb][kb]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s[qb][kb] = mfma32(kfr[kb][ks], qf[qb][ks], s[qb][kb]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // 256 registers: keep the V fragments out of the Q K^T phase
    bf16x8_t vfr[2][4];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) vfr[db][sp] = frag_cols(sV, db, (sp >> 1) * 32 + (sp & 1) * 16 + 4 * hi, lane);
    if (M0 == QB_UM) soft_pv(0, s[0], vfr, kv0, std::false_type{});
    if (M0 == QB_MASK) soft_pv(0, s[0], vfr, kv0, std::true_type{});
    if (M1 == QB_UM) soft_pv(1, s[1], vfr, kv0, std::false_type{});
    if (M1 == QB_MASK) soft_pv(1, s[1], vfr, kv0, std::true_type{});
  };

  // (tile 0 was issued in front of the Q loads, the others behind them)
  int i = 0, slot = 0;
  auto run = [&](int jt_end, auto m0_tag, auto m1_tag, bool active) {
    for (; jt_lo + i < jt_end; ++i) {
      // wait for this wave's pieces of tile i: the tiles issued after it (at most NST - 2, fewer at the end) may stay in flight
      const int rem = min(NST - 2, n - 1 - i);
      if (NST >= 4 && rem >= 2) attn_wait_vm<2 * ATTN_DMA_PER_STAGE>();
      else if (NST >= 3 && rem == 1) attn_wait_vm<ATTN_DMA_PER_STAGE>();
      else attn_wait_vm<0>();
      attn_barrier();  // everyone's pieces landed; and every wave is done reading tile i - 1, whose slot is refilled now
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
  run(a0, UM_{}, UM_{}, true);
  run(e0, MK_{}, UM_{}, true);
  run(a1, OFF_{}, UM_{}, true);
  run(e1, OFF_{}, MK_{}, true);
  run(jt_hi, OFF_{}, OFF_{}, false);

  // every wave has passed the last tile's barrier: all slots but the last tile's are free; slot n % NST holds this workgroup's O staging
  const RowStage rs{smem + (n % NST) * 2 * TILE + wave * 4096, lane};
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = r0[qb] + l31;
    float l_lo, l_hi;
    half_pair(lsum[qb], l_lo, l_hi);
    const float ltot = l_lo + l_hi;
    const float inv = 1.f / ltot;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4_t v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = f2bf(o[qb][db][4 * g + e] * inv);
        rs.put(l31, hi, db * 4 + g, v);
      }
    if (qrow < T && hi == 0) lse[((int64_t)b * nh + h) * T + qrow] = mc[qb] + __builtin_amdgcn_logf(ltot);  // base-2 LSE: the backward's exp2 argument directly
    rs.flush(out + (int64_t)b * T * dm, dm, r0[qb], T, h * HD);
  }
}

// =============================================================================================
// backward dQ: 256-query tiles, wave w owns the 32-row blocks w and 7 - w (equal causal work), K | V tiles through an NST-deep
// ring (see attn_fwd_causal_kernel).  Also computes delta[q] = sum_d dO[q][d] O[q][d] for its rows and publishes it for the dK/dV kernel.
// =============================================================================================
template <int NST>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_causal_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out,
                                                              const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                                                              float* __restrict__ delta, const float* __restrict__ rcos,
                                                              const float* __restrict__ rsin, uint16_t* __restrict__ dqkv, int T, int nh,
                                                              float dsign) {  // delta is published as dsign * delta (-1: what the dK/dV kernel reads into its dP accumulators)
  constexpr int KT = 64;
  constexpr int TILE = KT * 128;
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
  const int r0[2] = {q0 + 32 * wave, q0 + 32 * (7 - wave)};
  const uint16_t* base = qkv + (int64_t)b * T * ld + h * HD;
  const float scale = 0.125f, c2 = scale * LOG2E;

  const int kv_hi = min(T, q0 + QB);
  const int jt_hi = (kv_hi + KT - 1) / KT;
  constexpr int jt_lo = 0;
  const int n = jt_hi;

  TileDma dma;
  dma.init(wave, lane, ld);
  auto stage = [&](int slot, int jt) {  // 4 LDS-DMA instructions per wave
    const int kv0 = jt * KT;
    const uint16_t* src = base + (int64_t)kv0 * ld;
    if (kv0 + KT <= T) {
      dma.issue_full(smem + slot * 2 * TILE, src + dm, wave);
      dma.issue_full(smem + slot * 2 * TILE + TILE, src + 2 * dm, wave);
    } else {
      dma.issue(smem + slot * 2 * TILE, src + dm, ld, T - 1 - kv0, wave);
      dma.issue(smem + slot * 2 * TILE + TILE, src + 2 * dm, ld, T - 1 - kv0, wave);
    }
  };

  if (n > 0) stage(0, jt_lo);  // in front of the row loads: the two latencies overlap (see the forward kernel)

  bf16x8_t qf[2][4], dof[2][4];
  float Lq[2], Dq[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = r0[qb] + l31;
    const bool qvalid = qrow < T;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int d0 = ks * 16 + hi * 8;
      qf[qb][ks] = qvalid ? ld_bf16x8(base + (int64_t)qrow * ld + d0) : zero_bf16x8();
      dof[qb][ks] = qvalid ? ld_bf16x8(dout + ((int64_t)b * T + qrow) * dm + h * HD + d0) : zero_bf16x8();
    }
    Lq[qb] = 0.f;
    float part = 0.f;
    if (qvalid) {
      Lq[qb] = lse[((int64_t)b * nh + h) * T + qrow];  // base-2 LSE
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8_t o8 = ld_bf16x8(out + ((int64_t)b * T + qrow) * dm + h * HD + ks * 16 + hi * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) part += bf2f(o8[e]) * bf2f(dof[qb][ks][e]);
      }
    }
    float d_lo, d_hi;
    half_pair(part, d_lo, d_hi);  // rows beyond T hold zeros in both halves
    Dq[qb] = d_lo + d_hi;         // published in the epilogue (round 6): a store here had to be waited for before the counted DMA waits could start
  }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
    asm volatile("; q/dO fragments resident" ::"v"(qf[qb][0]), "v"(qf[qb][1]), "v"(qf[qb][2]), "v"(qf[qb][3]), "v"(dof[qb][0]), "v"(dof[qb][1]),
                 "v"(dof[qb][2]), "v"(dof[qb][3]), "v"(Lq[qb]), "v"(Dq[qb]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every ordinary load has landed (and tile 0 with them): nothing but LDS-DMA is counted from here on
#pragma unroll
  for (int i = 1; i < NST - 1; ++i)
    if (i < n) stage(i, jt_lo + i);

  f32x16_t dq[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    zero16(dq[qb][0]);
    zero16(dq[qb][1]);
  }

  // one 32-key block x one row block: S^T, dP^T, dS^T, dQ^T += K^T dS^T
  auto block = [&](int qb_, int kb, const bf16x8_t (&kfr)[4], const bf16x8_t (&vfr)[4], const bf16x8_t (&ktr)[2][2], int kv0, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const int qb = qb_;
    // key r of this lane is tile row kb*32 + (r&3) + 8*(r>>2) + 4*hi: visible iff  kb*32 + (r&3) + 8*(r>>2) <= c_hi
    const int c_hi = r0[qb] + l31 - kv0 - 4 * hi;
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
        p = (c <= c_hi) ? p : 0.f;
      }
      dsf[r >> 3][r & 7] = f2bf(p * (dp[r] - Dq[qb]));  // x 1/sqrt(hd) once, in the epilogue
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int db = 0; db < 2; ++db) dq[qb][db] = mfma32(ktr[db][s2], dsf[s2], dq[qb][db]);  // dQ^T[d][q]
  };

  auto compute = [&](int kv0, const char* sK, const char* sV, auto m0_tag, auto m1_tag) {
    constexpr int M0 = decltype(m0_tag)::value, M1 = decltype(m1_tag)::value;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      __builtin_amdgcn_sched_barrier(0);  // 256 registers: one key block's fragments at a time
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
      // (skipping the key block of a diagonal tile that lies entirely above a row block - one wave-uniform compare - costs this kernel
      // its last registers: 6 spills; the dK/dV kernel does skip its dead query blocks)
      if (M0 == QB_UM) block(0, kb, kfr, vfr, ktr, kv0, std::false_type{});
      if (M0 == QB_MASK) block(0, kb, kfr, vfr, ktr, kv0, std::true_type{});
      if (M1 == QB_UM) block(1, kb, kfr, vfr, ktr, kv0, std::false_type{});
      if (M1 == QB_MASK) block(1, kb, kfr, vfr, ktr, kv0, std::true_type{});
    }
  };

  int i = 0, slot = 0;  // (tile 0 was issued in front of the row loads, tile 1 behind them)
  auto run = [&](int jt_end, auto m0_tag, auto m1_tag, bool active) {
    for (; jt_lo + i < jt_end; ++i) {
      const int rem = min(NST - 2, n - 1 - i);
      if (NST >= 4 && rem >= 2) attn_wait_vm<2 * ATTN_DMA_PER_STAGE>();
      else if (NST >= 3 && rem == 1) attn_wait_vm<ATTN_DMA_PER_STAGE>();
      else attn_wait_vm<0>();
      attn_barrier();
      if (i + NST - 1 < n) stage(slot == 0 ? NST - 1 : slot - 1, jt_lo + i + NST - 1);
      if (active) compute((jt_lo + i) * KT, smem + slot * 2 * TILE, smem + slot * 2 * TILE + TILE, m0_tag, m1_tag);
      slot = (slot + 1 == NST) ? 0 : slot + 1;
    }
  };
  using OFF_ = std::integral_constant<int, QB_OFF>;
  using UM_ = std::integral_constant<int, QB_UM>;
  using MK_ = std::integral_constant<int, QB_MASK>;
  const int a0 = min(jt_hi, r0[0] / KT), e0 = min(jt_hi, r0[0] / KT + 1);
  const int a1 = min(jt_hi, r0[1] / KT), e1 = min(jt_hi, r0[1] / KT + 1);
  run(a0, UM_{}, UM_{}, true);
  run(e0, MK_{}, UM_{}, true);
  run(a1, OFF_{}, UM_{}, true);
  run(e1, OFF_{}, MK_{}, true);
  run(jt_hi, OFF_{}, OFF_{}, false);

  const RowStage rs{smem + (n % NST) * 2 * TILE + wave * 4096, lane};  // a slot nobody reads any more (see the forward kernel)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    if (r0[qb] + l31 < T && hi == 0) delta[((int64_t)b * nh + h) * T + r0[qb] + l31] = Dq[qb] * dsign;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4_t ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = f2bf(dq[qb][db][4 * g + e] * scale);
        rs.put(l31, hi, db * 4 + g, ov);
      }
    rs.flush_rot(dqkv + (int64_t)b * T * ld, ld, r0[qb], T, h * HD, rcos, rsin);  // inverse rotation: gradient w.r.t. the PRE-rotation q
  }
}

// =============================================================================================
// backward dK / dV: 4 waves x 32 key rows (128-key tiles, K / V of a wave's rows in registers), Q | dO | statistics tiles of 64
// queries through an NST-deep ring filled NST - 1 tiles ahead; every fragment of a 32-query block is requested before the block's first MFMA.
// =============================================================================================
// DOC (document masks, doc_start[B,T]): key j is seen by the queries j .. doc_end[j] - 1 (doc_start is non-decreasing, so the set is contiguous;
// doc_end[] comes from the plan kernel of attn_doc.hip) - in this kernel's S[q][kv] layout a lane owns a KEY, so the mask stays what it is in the
// causal case: two integer thresholds per lane (c_lo from the key's own index, c_end from its doc_end instead of T).  The workgroup takes its
// item (batch, first key, one past the last query tile, kind) from the plan's sorted list instead of the tile-major causal order.
// (Key tiles are never split into 64-key items the way the forward / dQ kernels split their heavy query tiles: with two workgroups per CU this
// kernel's grid is not resident at once at the reference's micro-batch, its launch lasts two rounds of fixed costs rather than one longest
// chain, and halving the chains changed nothing - tools/attn_trace.py, profiles/r06_attn_trace_*.)
template <int NST, bool DOC>
__device__ __forceinline__ void attn_bwd_dkdv_body(char* smem, const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout,
                                                   const float* __restrict__ lse, const float* __restrict__ ndelta,
                                                   const float* __restrict__ rcos, const float* __restrict__ rsin,
                                                   const int32_t* __restrict__ doc_end, uint16_t* __restrict__ dqkv, int T, int nh, int b, int h,
                                                   int kv0, int jq_plan, unsigned long long* trace_t = nullptr) {
  constexpr int QT = 64;
  constexpr int TILE = QT * 128;          // 8 KiB
  constexpr int STAGE = 2 * TILE + 512;   // Q | dO | statistics (lse[64], delta[64])
  constexpr int NQB = 2;                  // 32-query blocks of a query tile
  const int dm = nh * HD, ld = 3 * dm;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  constexpr int qoff = 0;
  const int kvw0 = kv0 + 32 * wave;
  const int kvrow = kvw0 + l31;
  const bool kvalid = kvrow < T;
  const uint16_t* base = qkv + (int64_t)b * T * ld + h * HD;
  const uint16_t* dobase = dout + (int64_t)b * T * dm + h * HD;
  const float* lrow = lse + ((int64_t)b * nh + h) * T;
  const float* drow = ndelta + ((int64_t)b * nh + h) * T;
  const float scale = 0.125f, c2 = scale * LOG2E;

  const int nqt = (T + QT - 1) / QT;
  const int jq_lo = kv0 / QT;
  const int jq_hi = DOC ? jq_plan : nqt;
  const int n = jq_hi - jq_lo;

  TileDma dma, dmad;
  dma.init(wave, lane, ld);
  dmad.init(wave, lane, dm);
  auto stage = [&](int slot, int jq) {  // 4 LDS-DMA instructions per wave, + 2 sixteen-lane ones on wave 0
    const int qt0 = jq * QT;
    char* dst = smem + slot * STAGE;
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
    }
  };
  // the first Q | dO tiles are on their way before the wave's K / V rows are asked for (the two latencies overlap; the ordinary loads are
  // younger than the DMA, so waiting for them covers it)
  bf16x8_t kf[4], vf[4];
  int de = T;  // first query that does not see this lane's key
  if (DOC) {
    // document masks (small grids, every workgroup starts in the same microsecond): the wave's K and V rows come by LDS-DMA into its quarters
    // of slots 1 and 2 (whole 128-byte rows: rows_dma, attn_common.h) instead of 32-byte fragment loads; tile 1 is staged when every wave
    // has its fragments (the barrier below), tile 2 at the loop's first step as always.  Rows beyond T repeat row T - 1 (never stored).
    static_assert(!DOC || NST == 3, "the prologue parks the wave's rows in slots 1 and 2");
    if (n > 0) stage(0, jq_lo);
    char* kreg = smem + 1 * STAGE + wave * 4096;
    char* vreg = smem + 2 * STAGE + wave * 4096;
    rows_dma(kreg, base + dm, ld, kvw0, T - 1, lane);
    rows_dma(vreg, base + 2 * dm, ld, kvw0, T - 1, lane);
    de = doc_end[(int64_t)b * T + min(kvrow, T - 1)];
    asm volatile("; row data requested" ::"v"(de));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // rows and tile 0 have landed: nothing but LDS-DMA is counted from here on
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = rows_frag(kreg, l31, ks, hi);
      vf[ks] = rows_frag(vreg, l31, ks, hi);
    }
    attn_barrier();
    if (n > 1) stage(1, jq_lo + 1);
  } else {
#pragma unroll
    for (int i = 0; i < NST - 1; ++i)
      if (i < n) stage(i, jq_lo + i);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const uint16_t* p = base + (int64_t)kvrow * ld + ks * 16 + hi * 8;
      kf[ks] = kvalid ? ld_bf16x8(p + dm) : zero_bf16x8();
      vf[ks] = kvalid ? ld_bf16x8(p + 2 * dm) : zero_bf16x8();
    }
    asm volatile("; k/v fragments resident" ::"v"(kf[0]), "v"(kf[1]), "v"(kf[2]), "v"(kf[3]), "v"(vf[0]), "v"(vf[1]), "v"(vf[2]), "v"(vf[3]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every ordinary load has landed: nothing but LDS-DMA is counted from here on
  }
  const int de_lo = DOC ? __builtin_amdgcn_readlane(de, 0) : T, de_hi = DOC ? __builtin_amdgcn_readlane(de, 31) : T;  // the wave's first / last key

  f32x16_t dk[2], dv[2];
  zero16(dk[0]); zero16(dk[1]); zero16(dv[0]); zero16(dv[1]);

  auto compute = [&](int qt0, const char* sQ, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const char* sDO = sQ + TILE;
    const float* sL = reinterpret_cast<const float*>(sQ + 2 * TILE);
    const float* sD = sL + 64;
    // query r = 4g + e of this lane is tile row qb*32 + 8g + e + 4*hi: visible iff  c_lo <= qb*32 + 8g + e < c_end
    const int c_lo = kvrow - qt0 - qoff - 4 * hi, c_end = de - qt0 - qoff - 4 * hi;
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      __builtin_amdgcn_sched_barrier(0);  // one query block's fragments at a time
      if (MASK && qt0 + qoff + qb * 32 + 31 < kvw0) continue;  // every query of the block precedes this wave's first key: P = 0 (wave-uniform)
      if (DOC && MASK && qt0 + qoff + qb * 32 >= de_hi) continue;  // ... or lies behind the document of its last key
      bf16x8_t qfr[4], dofr[4], dotr[2][2], qtr[2][2];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        qfr[ks] = frag_rows(sQ, qoff + qb * 32 + l31, ks, hi);
        dofr[ks] = frag_rows(sDO, qoff + qb * 32 + l31, ks, hi);
      }
      f32x4_t L4[4];
      f32x16_t s, dp;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ql0 = qoff + qb * 32 + 8 * g + 4 * hi;
        L4[g] = *reinterpret_cast<const f32x4_t*>(sL + ql0);  // base-2 LSE
        const f32x4_t nd = *reinterpret_cast<const f32x4_t*>(sD + ql0);  // -delta: the dP accumulators start from it (dP' = dP - delta)
#pragma unroll
        for (int e = 0; e < 4; ++e) dp[4 * g + e] = nd[e];
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int db = 0; db < 2; ++db) {  // every fragment of the block is requested before its first MFMA
          dotr[db][s2] = frag_cols(sDO, db, qoff + qb * 32 + s2 * 16 + 4 * hi, lane);
          qtr[db][s2] = frag_cols(sQ, db, qoff + qb * 32 + s2 * 16 + 4 * hi, lane);
        }
      zero16(s);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = mfma32(qfr[ks], kf[ks], s);       // S[q][kv]
        dp = mfma32(dofr[ks], vf[ks], dp);    // dP'[q][kv]
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
            p = ((c >= c_lo) && (c < c_end)) ? p : 0.f;
          }
          const float dsv = p * dp[r];  // the 1/sqrt(hd) factor (a power of two: exact) is applied once, to dK, in the epilogue
          pf[r >> 3][r & 7] = f2bf(p);
          dsf[r >> 3][r & 7] = f2bf(dsv);
        }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dv[db] = mfma32(dotr[db][s2], pf[s2], dv[db]);   // dV^T[d][kv]
          dk[db] = mfma32(qtr[db][s2], dsf[s2], dk[db]);   // dK^T[d][kv]
        }
#ifdef PLM_ATTN_ONEPASS_ABLATION
      if (g_onepass_mode) {
        // TIMING ONLY (tools/attn_onepass_ablation.py; results are garbage): what a one-pass backward would add to this block.  dS is in
        // this kernel's layout (a lane owns a KEY); dQ contracts over keys, so dS has to change hands - through LDS into the A-operand
        // layout (a lane owns a query) - then 4 MFMAs with K fragments from LDS (stood in for by the Q tile: same instructions) give the
        // block's 32 x 64 partial dQ, which mode 2 adds to an fp32 [M, d] buffer with atomics (whole 128-byte row segments per half wave).
        char* scr = smem + NST * STAGE + wave * 2048;
        *reinterpret_cast<bf16x8_t*>(scr + lane * 32) = dsf[0];
        *reinterpret_cast<bf16x8_t*>(scr + lane * 32 + 16) = dsf[1];
        bf16x8_t at[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          at[s2] = join_tr(lds_read_tr16(scr + s2 * 1024 + (lane & 15) * 64 + (lane >> 4) * 8), lds_read_tr16(scr + s2 * 1024 + 512 + (lane & 15) * 32 + (lane >> 4) * 8));
        f32x16_t dqp[2];
        zero16(dqp[0]);
        zero16(dqp[1]);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int db = 0; db < 2; ++db) dqp[db] = mfma32(at[s2], frag_cols(sQ, db, qoff + qb * 32 + s2 * 16 + 4 * hi, lane), dqp[db]);
        if (g_onepass_mode == 2) {
#pragma unroll
          for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int q = min(qt0 + qb * 32 + mfma32_row(r, hi), T - 1);
              atomicAdd(g_onepass_buf + ((int64_t)b * T + q) * dm + h * HD + db * 32 + l31, dqp[db][r]);
            }
        } else {
          asm volatile("; partial dQ" ::"v"(dqp[0]), "v"(dqp[1]));
        }
      }
#endif
    }
  };

  // per wave: tiles entirely above its first key (idle), the tiles its diagonal crosses (masked), the tiles below its last key (no mask),
  // a partial last tile when T % 64 != 0 (masked)
  // (DOC: ... the tiles behind the document of its first key (masked again), the tiles behind the document of its last key (idle))
  const bool wave_rows = kvw0 < T;
  const int jq_act = wave_rows ? min(jq_hi, max(jq_lo, kvw0 / QT)) : jq_hi;
  const int jq_m = min(jq_hi, max(jq_act, (kvw0 + 31 + QT - 1) / QT));
  const int jq_u = min(jq_hi, max(jq_m, de_lo / QT));
  const int jq_e = min(jq_hi, max(jq_u, (de_hi + QT - 1) / QT));
#ifdef PLM_ATTN_TRACE
  if (trace_t) trace_t[0] = __builtin_amdgcn_s_memrealtime();
#endif
  int i = 0, slot = 0;
  auto run = [&](int jq_end, auto mask_tag, bool active) {
    for (; jq_lo + i < jq_end; ++i) {
      // wait for this wave's pieces of tile i; the tiles issued after it (at most NST - 2) stay in flight: 4 instructions each, plus
      // the statistics pieces on wave 0
      const int rem = min(NST - 2, n - 1 - i);
      constexpr int W0 = ATTN_DMA_PER_STAGE + ATTN_DMA_STATS;
      if (wave == 0) {
        if (NST >= 4 && rem >= 2) attn_wait_vm<2 * W0>();
        else if (NST >= 3 && rem == 1) attn_wait_vm<W0>();
        else attn_wait_vm<0>();
      } else {
        if (NST >= 4 && rem >= 2) attn_wait_vm<2 * ATTN_DMA_PER_STAGE>();
        else if (NST >= 3 && rem == 1) attn_wait_vm<ATTN_DMA_PER_STAGE>();
        else attn_wait_vm<0>();
      }
      attn_barrier();
      if (i + NST - 1 < n) stage(slot == 0 ? NST - 1 : slot - 1, jq_lo + i + NST - 1);
      if (active) compute((jq_lo + i) * QT, smem + slot * STAGE, mask_tag);
      slot = (slot + 1 == NST) ? 0 : slot + 1;
    }
  };
  run(jq_act, std::true_type{}, false);
  run(jq_m, std::true_type{}, true);
  run(jq_u, std::false_type{}, true);
  run(jq_e, std::true_type{}, true);
  if (DOC) run(jq_hi, std::true_type{}, false);
#ifdef PLM_ATTN_TRACE
  if (trace_t) trace_t[1] = __builtin_amdgcn_s_memrealtime();
#endif

  const RowStage rs{smem + (n % NST) * STAGE + wave * 4096, lane};  // a slot nobody reads any more (see the forward kernel)
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4_t ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[e] = f2bf(dv[db][4 * g + e]);
      rs.put(l31, hi, db * 4 + g, ov);
    }
  rs.flush(dqkv + (int64_t)b * T * ld, ld, kvw0, T, 2 * dm + h * HD);
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4_t ok;
#pragma unroll
      for (int e = 0; e < 4; ++e) ok[e] = f2bf(dk[db][4 * g + e] * scale);
      rs.put(l31, hi, db * 4 + g, ok);
    }
  rs.flush_rot(dqkv + (int64_t)b * T * ld, ld, kvw0, T, dm + h * HD, rcos, rsin);  // inverse rotation: gradient w.r.t. the PRE-rotation k
}


template <int NST>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_causal_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout,
                                                                      const float* __restrict__ lse, const float* __restrict__ ndelta,
                                                                      const float* __restrict__ rcos, const float* __restrict__ rsin,
                                                                      uint16_t* __restrict__ dqkv, int T, int nh) {
  __shared__ __attribute__((aligned(1024))) char smem[NST * (2 * 8192 + 512) + ONEPASS_LDS];
  int kt, h, b;  // key tile 0 meets every query tile: heaviest first
  attn_block2<128>(T, nh, kt, h, b);
  attn_bwd_dkdv_body<NST, false>(smem, qkv, dout, lse, ndelta, rcos, rsin, nullptr, dqkv, T, nh, b, h, kt * 128, 0);
}

template <int NST>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_doc_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout,
                                                                   const float* __restrict__ lse, const float* __restrict__ ndelta,
                                                                   const float* __restrict__ rcos, const float* __restrict__ rsin,
                                                                   const int32_t* __restrict__ header, const int32_t* __restrict__ doc_end,
                                                                   const int4* __restrict__ items, uint16_t* __restrict__ dqkv, int T, int nh) {
  __shared__ __attribute__((aligned(1024))) char smem[NST * (2 * 8192 + 512) + ONEPASS_LDS];
  ATTN_TRACE_T(tr0);
  const int idx = blockIdx.x / nh, h = blockIdx.x - idx * nh;
  if (idx >= __builtin_amdgcn_readfirstlane(header[0])) return;
  const int4 it = items[idx];
  const int b = __builtin_amdgcn_readfirstlane(it.x), kv0 = __builtin_amdgcn_readfirstlane(it.y), jq_hi = __builtin_amdgcn_readfirstlane(it.z);
  const int kc = __builtin_amdgcn_readfirstlane(it.w);
  (void)kc;
#ifdef PLM_ATTN_TRACE
  unsigned long long tt[2] = {0, 0};
#else
  unsigned long long* tt = nullptr;
#endif
  attn_bwd_dkdv_body<NST, true>(smem, qkv, dout, lse, ndelta, rcos, rsin, doc_end, dqkv, T, nh, b, h, kv0, jq_hi, tt);
  ATTN_TRACE_END(2, idx, kc, tr0, tt[0], tt[1]);
}

// =============================================================================================
// launchers (called from the C ABI entry points in attn.hip when no document mask is given)
// =============================================================================================
void plm_attn_fwd_causal(const uint16_t* qkv, uint16_t* out, float* lse, int64_t B, int64_t T, int64_t nh, hipStream_t s) {
  const dim3 grid((unsigned)(plm_cdiv(T, 256) * nh * B)), block(256);
  hipLaunchKernelGGL((attn_fwd_causal_kernel<4>), grid, block, 0, s, qkv, out, lse, (int)T, (int)nh);
}
void plm_attn_bwd_causal(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* lse, float* delta, const float* rc,
                         const float* rs, uint16_t* dqkv, int64_t B, int64_t T, int64_t nh, hipStream_t s) {
  const dim3 block(256);
  // dQ first: it computes delta[b,h,q] for its queries and publishes -delta for the dK/dV kernel (which reads it straight into its dP accumulators)
  hipLaunchKernelGGL((attn_bwd_dq_causal_kernel<3>), dim3((unsigned)(plm_cdiv(T, 256) * nh * B)), block, 0, s, qkv, out, dout, lse, delta, rc, rs, dqkv, (int)T, (int)nh, -1.f);
  hipLaunchKernelGGL((attn_bwd_dkdv_causal_kernel<3>), dim3((unsigned)(plm_cdiv(T, 128) * nh * B)), block, 0, s, qkv, dout, lse, delta, rc, rs, dqkv,
                     (int)T, (int)nh);
}
// dK / dV of a document-masked batch (the dQ kernel of attn_doc.hip runs first and publishes -delta)
void plm_attn_bwd_dkdv_doc(const uint16_t* qkv, const uint16_t* dout, const float* lse, const float* ndelta, const float* rc, const float* rs,
                           const int32_t* header, const int32_t* doc_end, const int4* items_k, uint16_t* dqkv, int64_t B, int64_t T, int64_t nh,
                           hipStream_t s) {
  hipLaunchKernelGGL((attn_bwd_dkdv_doc_kernel<3>), dim3((unsigned)(doc_plan_cap(B, T) * nh)), dim3(256), 0, s, qkv, dout, lse, ndelta, rc, rs,
                     header, doc_end, items_k, dqkv, (int)T, (int)nh);
}
