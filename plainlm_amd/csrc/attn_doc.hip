// Attention with document masks (doc_start[B,T]: query i sees key j iff doc_start[i] <= j <= i), third generation: the plan kernel, the
// bool-mask conversion, forward and backward dQ.  (Backward dK / dV is the DOC mode of attn_bwd_dkdv_body in attn_causal.hip.)
// Replaces F.scaled_dot_product_attention(q, k, v, attn_mask=...) of models/transformer.py:52-61 with the masks of
// data/datasets/data_prep_utils.py:7-23.
//
// What a per-workgroup trace (tools/attn_trace.py) showed about the second-generation kernels at the reference's own micro-batch
// (config_doc_mask.yaml:35, B = 8: 768 workgroups, every one of them resident at once; MFMA utilisation 0.09-0.12) and what this file does about it:
//   * The work of a 128-row tile is the number of 64-key tiles between the first document of its rows and its diagonal: 2 ... 16 at T = 1024 with a
//     mean of 5 - and a launch whose whole grid is resident lasts as long as its LONGEST workgroup (most ended after 11-13 us, the twelve with 14
//     tile steps after 22.5).  The grid was ordered by tile INDEX (the causal kernels' heaviest-first order), which says nothing about cost here.  A
//     small PLAN kernel (once per batch: every layer's forward, dQ and dK/dV launches share it) sorts the tiles by their actual cost, and - when the
//     grid is resident at once - enters the heaviest query tiles as two 64-row items whose four waves are 2 row blocks x 2 halves of every key tile:
//     half the chain of dependent tile steps, partial results merged once through LDS (attn_common.h has the plan's layout).
//   * The dK/dV kernel found its last query tile with a loop of DEPENDENT scalar loads (one per query tile: up to 16 x ~0.7 us in front of the first DMA);
//     the plan carries it, and a per-key doc_end[] turns that kernel's mask into two integer thresholds per lane - the causal kernel's own form.
//   * A workgroup cost ~10 us before its first and after its last tile step.  Two LDS stages with a draining vmcnt(0) per tile -> the causal kernels'
//     NST-deep ring with counted waits, tile 0 issued in front of the row loads; the wave's own rows by LDS-DMA (whole 128-byte rows instead of
//     fragment loads that touch 32 lines per instruction); the inverse RoPE rotation on the store side of the epilogue (RowStage::flush_rot).
//   * Every tile took the element-wise mask (five VALU instructions per score) -> per WAVE the tiles fall into idle / masked / unmasked / masked / idle
//     segments (doc_start is non-decreasing, so a wave's first and last rows bound all of them): static loops like the causal kernels', masks are two
//     integer thresholds per lane, the forward softmax defers its rescale.
// Waves stay at 32 rows: at B = 8 the kernels are bound by the longest chain of dependent tile steps and by fixed costs, not by LDS reads per MFMA.
#include "plm_device.h"

#include <type_traits>

#include "attn_common.h"

// LDS-DMA instructions one stage() issues per wave (see ATTN_DMA_PER_STAGE in attn_causal.hip: the counted waits are multiples of it).
constexpr int DOC_DMA_PER_STAGE = 4;
#define DOC_DEFER_LOG2 8.0f
#define PLM_ATTN_TRACE_SETTER plm_dbg_attn_trace_doc
ATTN_TRACE_DECL()

// =============================================================================================
// plan: doc_end[] + the two sorted item lists (layout: attn_common.h)
// =============================================================================================
constexpr int DOC_PLAN_SORT_MAX = 768;  // tiles per list the single sorting workgroup ranks in LDS; longer lists stay in tile order, unsplit

// first query q > j that does not see key j (doc_start[q] > j), T if every later query does: binary search on the non-decreasing row
__device__ __forceinline__ int doc_end_of(const int32_t* __restrict__ ds_row, int j, int T) {
  int lo = j + 1, hi = T;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (ds_row[mid] > j) hi = mid;
    else lo = mid + 1;
  }
  return lo;
}
// expected duration of an item in tile steps of a whole 128-row workgroup: a split half walks its tiles at about twice the rate and pays a combine
__device__ __forceinline__ int doc_item_est(int cost, bool split) { return split ? (cost + 1) / 2 + 1 : cost; }

__global__ __launch_bounds__(1024) void attn_doc_plan_kernel(const int32_t* __restrict__ ds, int32_t* __restrict__ plan, int B, int T, int nh,
                                                            int split_min_q) {
  // per list (0: query items, 1: key items) and tile: cost | cost of half 0 | cost of half 1 | bound 0 | bound 1 | items (1 or 2); rk: ranks
  __shared__ int sh[2][6][DOC_PLAN_SORT_MAX];
  __shared__ int rk[2][2 * DOC_PLAN_SORT_MAX];
  __shared__ int n_split[2];
  const int nt = (T + 127) / 128, n = B * nt;
  const int64_t BT = (int64_t)B * T;
  const int tid = threadIdx.x;
  if (blockIdx.x + 1 < gridDim.x) {  // doc_end[]
    const int64_t i = (int64_t)blockIdx.x * 1024 + tid;
    if (i < BT) {
      const int b = (int)(i / T), j = (int)(i - (int64_t)b * T);
      plan[8 + i] = doc_end_of(ds + (int64_t)b * T, j, T);
    }
    return;
  }
  const int cap = (int)doc_plan_cap(B, T);
  int4* iq = reinterpret_cast<int4*>(plan + doc_plan_pq(B, T));
  int4* ik = iq + cap;
  const bool sorted = n <= DOC_PLAN_SORT_MAX;
  // halves of heavy tiles become items of their own only when the launch lasts as long as its longest item: a grid that is resident (nearly) at once
  const bool small = sorted && (int64_t)n * nh <= 1024;
  const int split_min[2] = {small ? split_min_q : 0, 0};  // key tiles are never split (attn_causal.hip, dK/dV)
  if (tid < 2) n_split[tid] = 0;
  if (tid < 8) plan[tid] = tid < 2 ? n : 0;
  for (int i = tid; i < n; i += 1024) {
    const int b = i / nt, t = i - b * nt, r0 = t * 128, r1 = min(r0 + 64, T - 1);
    const int32_t* row = ds + (int64_t)b * T;
    const int lo0 = row[r0] / 64, lo1 = row[r1] / 64;
    const int hi0 = (min(T, r0 + 64) + 63) / 64, hi1 = (min(T, r0 + 128) + 63) / 64;
    const int e0 = (doc_end_of(row, min(r0 + 63, T - 1), T) + 63) / 64, e1 = (doc_end_of(row, min(r0 + 127, T - 1), T) + 63) / 64;
    if (sorted) {
      sh[0][0][i] = hi1 - lo0;  sh[0][1][i] = hi0 - lo0;       sh[0][2][i] = hi1 - lo1;             sh[0][3][i] = lo0;  sh[0][4][i] = lo1;
      sh[1][0][i] = e1 - r0 / 64;  sh[1][1][i] = e0 - r0 / 64;  sh[1][2][i] = e1 - (r0 + 64) / 64;  sh[1][3][i] = e0;  sh[1][4][i] = e1;
      sh[0][5][i] = sh[1][5][i] = 1;
      rk[0][i] = rk[1][i] = 0;
    } else {
      iq[i] = make_int4(b, r0, lo0, hi1 - lo0);
      ik[i] = make_int4(b, r0, e1, e1 - r0 / 64);
    }
  }
  if (!sorted) return;
  __syncthreads();
  // Ranks are counted by whole WAVES (one item per wave at a time, the lanes walk the list 64 entries per step, ballot + popcount): no
  // atomics, deterministic, ~4 steps per item at n = 256.  (One thread per tile walking the list alone took 60 us; pair-parallel LDS atomics
  // serialised on the item's counter.)
  const int wave = tid >> 6, lane = tid & 63;
  // which tiles are split: the n / 4 heaviest, if their cost reaches split_min (and their second half exists)
  for (int l = 0; l < 2; ++l) {
    if (split_min[l] <= 0) continue;  // uniform
    for (int i = wave; i < n; i += 16) {
      const int c = sh[l][0][i];
      int rank = 0;
      if (c >= split_min[l])
        for (int j0 = 0; j0 < n; j0 += 64) {
          const int j = j0 + lane;
          const int cj = j < n ? sh[l][0][j] : -1;
          rank += __builtin_popcountll(__builtin_amdgcn_ballot_w64((cj > c) || (cj == c && j < i)));
        }
      if (lane == 0) rk[l][i] = rank;
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += 1024)
    for (int l = 0; l < 2; ++l) {
      const bool two = split_min[l] > 0 && sh[l][0][i] >= split_min[l] && rk[l][i] < n / 4 && (i % nt) * 128 + 64 < T;
      sh[l][5][i] = two ? 2 : 1;
      if (two) atomicAdd(&n_split[l], 1);
    }
  __syncthreads();
  if (tid < 2) plan[tid] = n + n_split[tid];
  // stable rank of every item by descending expected duration (ties in tile order); slot 2 i + a = half a of tile i
  for (int l = 0; l < 2; ++l) {
    const int step = n_split[l] ? 1 : 2;  // no split tile: only the even slots exist
    for (int si = wave * step; si < 2 * n; si += 16 * step) {
      const int i = si >> 1, a = si & 1, iti = sh[l][5][i];
      if (a >= iti) continue;  // wave-uniform
      const int est = doc_item_est(iti == 2 ? sh[l][1 + a][i] : sh[l][0][i], iti == 2);
      int rank = 0;
      for (int j0 = 0; j0 < 2 * n; j0 += 64 * step) {
        const int sj = j0 + lane * step, j = sj >> 1, a2 = sj & 1;
        bool before = false;
        if (sj < 2 * n) {
          const int itj = sh[l][5][j];
          if (a2 < itj) {
            const int ej = doc_item_est(itj == 2 ? sh[l][1 + a2][j] : sh[l][0][j], itj == 2);
            before = (ej > est) || (ej == est && sj < si);
          }
        }
        rank += __builtin_popcountll(__builtin_amdgcn_ballot_w64(before));
      }
      if (lane == 0) rk[l][si] = rank;
    }
  }
  __syncthreads();
  for (int si = tid; si < 2 * n; si += 1024)
    for (int l = 0; l < 2; ++l) {
      const int i = si >> 1, a = si & 1, items = sh[l][5][i];
      if (a >= items) continue;
      const int b = i / nt, r0 = (i - b * nt) * 128;
      const int cost = items == 2 ? sh[l][1 + a][i] : sh[l][0][i];
      // query items: bound = the first key tile of the item's first row; key items: one past the last query tile that sees the item's last key
      const int bound = l == 0 ? sh[0][3 + a][i] : (items == 2 ? sh[1][3 + a][i] : sh[1][4][i]);
      const int pos = rk[l][si];
      (l == 0 ? iq : ik)[pos] = make_int4(b, r0 + 64 * a, bound, ((items == 2) << DOC_KIND_SHIFT) | cost);
    }
}

// =============================================================================================
// The reference's own calling convention: a bool [B, T, T] mask (True = may attend; data_prep_utils.py:7-23 stacked at engine/engine.py:21-23).
// One wave per query row: doc_start = the row's first True, and the row is CHECKED to be exactly True on [doc_start, i] - the only masks
// doc_start[B, T] can express; any other row raises status[0] (the caller reports it: a mask that is silently mis-read would train on the
// wrong attention pattern).  Replaces an aten cast + argmax over B T^2 elements.
// =============================================================================================
__global__ __launch_bounds__(256) void doc_start_from_mask_kernel(const uint8_t* __restrict__ mask, int32_t* __restrict__ doc_start,
                                                                  int32_t* __restrict__ status, int64_t rows, int T) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int i = (int)(row % T);
  const uint32_t* w = reinterpret_cast<const uint32_t*>(mask + row * T);  // T % 4 == 0: every row starts on a word
  const int nw = T >> 2;
  int first = T;
  for (int k = lane; k < nw; k += 64) {
    const uint32_t v = w[k];
    if (v && first == T) first = 4 * k + (__builtin_ctz(v) >> 3);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
  bool bad = first > i;  // a query sees at least itself
  const int ds = bad ? i : first;
  for (int k = lane; k < nw; k += 64) {
    const uint32_t v = w[k];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = 4 * k + e;
      bad |= (((v >> (8 * e)) & 0xffu) != 0u) != (j >= ds && j <= i);
    }
  }
  if (__builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) atomicOr(status, 1);
  if (lane == 0) doc_start[row] = ds;
}

// =============================================================================================
// forward.  A 128-row item (SPLIT = false): 4 waves x 32 query rows; a 64-row item (SPLIT = true): wave = (row block rb = wave & 1, part = wave >> 1),
// the two parts of a row block take the two 32-key halves of every tile and merge their (reference maximum, sum, O) once, at the end, through LDS.
// K | V tiles of 64 keys go through an NST-deep LDS ring filled NST - 1 tiles ahead (counted vmcnt waits, one barrier per tile).
// S^T = K Q^T (a lane owns a query), deferred running-max update, masks = two integer thresholds per lane.
// Per wave (rows qw0 .. qw0 + 31, first documents ds_lo <= ds_hi of its first / last row, diagonal tile jd = qw0 / 64) the tiles are
//   [jt_lo, ja) idle (before every row's document) | [ja, jm) masked | [jm, jd) no mask (every key visible to every row) | jd masked | (jd, jt_hi) idle.
// =============================================================================================
template <int NST, bool SPLIT>
__device__ __forceinline__ void attn_fwd_doc_body(char* smem, const uint16_t* __restrict__ qkv, const int32_t* __restrict__ doc_start,
                                                  uint16_t* __restrict__ out, float* __restrict__ lse, int T, int nh, int b, int h, int q0,
                                                  int jt_lo, unsigned long long* trace_t = nullptr) {
  constexpr int KT = 64;
  constexpr int TILE = KT * 128;  // 8 KiB
  constexpr int NKB = SPLIT ? 1 : 2;  // 32-key blocks of a tile this wave takes
  const int dm = nh * HD, ld = 3 * dm;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int part = SPLIT ? (wave >> 1) : 0, koff = part * 32;
  const int qw0 = q0 + 32 * (SPLIT ? (wave & 1) : wave), qrow = qw0 + l31;
  const bool qvalid = qrow < T;
  const uint16_t* base = qkv + (int64_t)b * T * ld + h * HD;
  const float c2 = 0.125f * LOG2E;  // 1/sqrt(64) and the base-2 exponent in one factor

  const int jt_hi = (min(T, q0 + (SPLIT ? 64 : 128)) + KT - 1) / KT;
  const int n = jt_hi - jt_lo;

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
  // the first K | V tiles are on their way BEFORE the Q rows are asked for: the item carries everything their addresses need, and the two
  // latencies overlap (the ordinary loads below are younger than the DMA, so waiting for them covers the DMA as well; from the loop on,
  // nothing but LDS-DMA is in flight - see the note on counted waits in attn_causal.hip)
#pragma unroll
  for (int i = 0; i < NST - 1; ++i)
    if (i < n) stage(i, jt_lo + i);
  // the wave's own Q rows: LDS-DMA into its quarter of the ring's last slot (nobody stages into it before the first tile's barrier), whole
  // 128-byte rows instead of 32-byte fragments (rows_dma, attn_common.h); rows beyond T repeat row T - 1 (never stored)
  char* qreg = smem + (NST - 1) * 2 * TILE + wave * 4096;
  rows_dma(qreg, base, ld, qw0, T - 1, lane);
  const int dsq = doc_start[(int64_t)b * T + min(qrow, T - 1)];  // rows beyond T repeat the last row (keeps the wave's bounds monotone)
  asm volatile("; row data requested" ::"v"(dsq));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the wave's rows and the first tiles have landed: nothing but LDS-DMA is counted from here on
  bf16x8_t qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = rows_frag(qreg, l31, ks, hi);

  f32x16_t o[2];
  zero16(o[0]);
  zero16(o[1]);
  float mc = -INFINITY, lsum = 0.f;  // running reference maximum in log2 units (s * c2), running sum

  // the wave's segment bounds (wave-uniform)
  const int ds_lo = __builtin_amdgcn_readlane(dsq, 0), ds_hi = __builtin_amdgcn_readlane(dsq, 31);
  const int jd = min(jt_hi - 1, qw0 / KT);                       // the wave's diagonal tile (a wave entirely beyond T computes on zeros, stores nothing)
  const int ja = min(jd, max(jt_lo, ds_lo / KT));                // first tile with a visible key
  const int jm = min(jd, max(ja, (ds_hi + KT - 1) / KT));        // first tile whose every key is at or behind the last row's document start

  // one K | V tile (this wave's 32-key blocks of it): S^T = K Q^T, softmax with deferred rescale, O^T += V^T P^T
  auto compute = [&](int kv0, const char* sK, const char* sV, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    bf16x8_t kfr[NKB][4];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kfr[kb][ks] = frag_rows(sK, koff + kb * 32 + l31, ks, hi);
    f32x16_t s[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      zero16(s[kb]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s[kb] = mfma32(kfr[kb][ks], qf[ks], s[kb]);
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the V fragments out of the Q K^T phase (register budget: three workgroups per CU)
    bf16x8_t vfr[2][2 * NKB];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int sp = 0; sp < 2 * NKB; ++sp) vfr[db][sp] = frag_cols(sV, db, koff + (sp >> 1) * 32 + (sp & 1) * 16 + 4 * hi, lane);
    // key (kb, r) of this lane is tile row koff + c + 4*hi with c = kb*32 + (r&3) + 8*(r>>2): visible iff c_lo <= c <= c_hi
    const int c_hi = qrow - kv0 - koff - 4 * hi, c_lo = dsq - kv0 - koff - 4 * hi;
    float tmax = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (MASK) {
          const int c = kb * 32 + (r & 3) + 8 * (r >> 2);
          if (c > c_hi || c < c_lo) s[kb][r] = -INFINITY;
        }
        tmax = fmaxf(tmax, s[kb][r]);
      }
    {
      float t_lo, t_hi;
      half_pair(tmax, t_lo, t_hi);
      tmax = fmaxf(t_lo, t_hi);
    }
    const float tm = tmax * c2;
    const bool need = tm > mc + DOC_DEFER_LOG2;  // both -inf (nothing visible yet): false
    if (__builtin_amdgcn_ballot_w64(need) != 0ull) {  // wave-uniform and rare after a row's first visible tile
      const float mn = fmaxf(mc, tm);
      const float alpha = fast_exp2(mc - ((mn == -INFINITY) ? 0.f : mn));
      mc = mn;
      lsum *= alpha;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    }
    const float mref = (MASK && mc == -INFINITY) ? 0.f : mc;
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
    bf16x8_t pf[2 * NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(__builtin_fmaf(s[kb][r], c2, -mref));
        ps[r & 3] += p;
        pf[kb * 2 + (r >> 3)][r & 7] = f2bf(p);
      }
    lsum += (ps[0] + ps[1]) + (ps[2] + ps[3]);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int sp = 0; sp < 2 * NKB; ++sp) o[db] = mfma32(vfr[db][sp], pf[sp], o[db]);
  };

  int i = 0, slot = 0;
  auto run = [&](int jt_end, auto mask_tag, bool active) {
    for (; jt_lo + i < jt_end; ++i) {
      // wait for this wave's pieces of tile i: the tiles issued after it (at most NST - 2, fewer at the end) may stay in flight
      const int rem = min(NST - 2, n - 1 - i);
      if (NST >= 4 && rem >= 2) attn_wait_vm<2 * DOC_DMA_PER_STAGE>();
      else if (NST >= 3 && rem == 1) attn_wait_vm<DOC_DMA_PER_STAGE>();
      else attn_wait_vm<0>();
      attn_barrier();  // everyone's pieces landed; and every wave is done reading tile i - 1, whose slot is refilled now
      if (i + NST - 1 < n) stage(slot == 0 ? NST - 1 : slot - 1, jt_lo + i + NST - 1);
      if (active) compute((jt_lo + i) * KT, smem + slot * 2 * TILE, smem + slot * 2 * TILE + TILE, mask_tag);
      slot = (slot + 1 == NST) ? 0 : slot + 1;
    }
  };
#ifdef PLM_ATTN_TRACE
  if (trace_t) trace_t[0] = __builtin_amdgcn_s_memrealtime();
#endif
  run(ja, std::true_type{}, false);
  run(jm, std::true_type{}, true);
  run(jd, std::false_type{}, true);
  run(jd + 1, std::true_type{}, true);
  run(jt_hi, std::true_type{}, false);
#ifdef PLM_ATTN_TRACE
  if (trace_t) trace_t[1] = __builtin_amdgcn_s_memrealtime();
#endif

  char* stage_base = smem + (n % NST) * 2 * TILE;  // every wave has passed the last tile's barrier: slot n % NST is read by nobody any more
  if (SPLIT) {
    // merge the two parts of a row block: part 1 leaves (O, reference maximum, sum) in LDS, part 0 rescales both to the common maximum and adds.
    // The ring is free once everyone is past the last tile (barrier); [0, 17 KiB) exchange, [32 KiB, ...) the output staging.
    attn_barrier();
    float* xch = reinterpret_cast<float*>(smem) + (wave & 1) * (34 * 64);
    if (part == 1) {
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[(db * 16 + r) * 64 + lane] = o[db][r];
      xch[32 * 64 + lane] = mc;
      xch[33 * 64 + lane] = lsum;
    }
    attn_barrier();
    if (part == 1) return;
    const float m1 = xch[32 * 64 + lane], l1 = xch[33 * 64 + lane];
    const float mn = fmaxf(mc, m1);  // never -inf for both parts: a row sees at least itself
    const float a0 = fast_exp2(mc - mn), a1 = fast_exp2(m1 - mn);
    lsum = lsum * a0 + l1 * a1;
    mc = mn;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] = o[db][r] * a0 + xch[(db * 16 + r) * 64 + lane] * a1;
    stage_base = smem + 32768;
  }
  float l_lo, l_hi;
  half_pair(lsum, l_lo, l_hi);
  const float ltot = l_lo + l_hi;
  const float inv = 1.f / ltot;
  const RowStage rs{stage_base + wave * 4096, lane};
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4_t v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = f2bf(o[db][4 * g + e] * inv);
      rs.put(l31, hi, db * 4 + g, v);
    }
  if (qvalid && hi == 0) lse[((int64_t)b * nh + h) * T + qrow] = mc + __builtin_amdgcn_logf(ltot);  // base-2 LSE: the backward's exp2 argument directly
  rs.flush(out + (int64_t)b * T * dm, dm, qw0, T, h * HD);
}

template <int NST>
__global__ __launch_bounds__(256, 3) void attn_fwd_doc_kernel(const uint16_t* __restrict__ qkv, const int32_t* __restrict__ doc_start,
                                                              const int32_t* __restrict__ header, const int4* __restrict__ items,
                                                              uint16_t* __restrict__ out, float* __restrict__ lse, int T, int nh) {
  __shared__ __attribute__((aligned(1024))) char smem[NST * 2 * 8192];  // [stage][K|V]
  ATTN_TRACE_T(tr0);
  const int idx = blockIdx.x / nh, h = blockIdx.x - idx * nh;
  if (idx >= __builtin_amdgcn_readfirstlane(header[0])) return;
  const int4 it = items[idx];
  const int b = __builtin_amdgcn_readfirstlane(it.x), q0 = __builtin_amdgcn_readfirstlane(it.y), jt_lo = __builtin_amdgcn_readfirstlane(it.z);
  const int kc = __builtin_amdgcn_readfirstlane(it.w);
#ifdef PLM_ATTN_TRACE
  unsigned long long tt[2] = {0, 0};
#else
  unsigned long long* tt = nullptr;
#endif
  if (kc >> DOC_KIND_SHIFT) attn_fwd_doc_body<NST, true>(smem, qkv, doc_start, out, lse, T, nh, b, h, q0, jt_lo, tt);
  else attn_fwd_doc_body<NST, false>(smem, qkv, doc_start, out, lse, T, nh, b, h, q0, jt_lo, tt);
  ATTN_TRACE_END(0, idx, kc, tr0, tt[0], tt[1]);
}

// =============================================================================================
// backward dQ: the forward's tiling, items and segments; also computes delta[q] = sum_d dO[q][d] O[q][d] for its rows and publishes -delta
// (what the dK/dV kernel reads straight into its dP accumulators).  SPLIT: the two parts of a row block sum their dQ through LDS.
// =============================================================================================
template <int NST, bool SPLIT>
__device__ __forceinline__ void attn_bwd_dq_doc_body(char* smem, const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out,
                                                     const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                                                     float* __restrict__ ndelta, const float* __restrict__ rcos,
                                                     const float* __restrict__ rsin, const int32_t* __restrict__ doc_start,
                                                     uint16_t* __restrict__ dqkv, int T, int nh, int b, int h, int q0, int jt_lo,
                                                     unsigned long long* trace_t = nullptr) {
  constexpr int KT = 64;
  constexpr int TILE = KT * 128;
  constexpr int NKB = SPLIT ? 1 : 2;
  const int dm = nh * HD, ld = 3 * dm;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int part = SPLIT ? (wave >> 1) : 0, koff = part * 32;
  const int qw0 = q0 + 32 * (SPLIT ? (wave & 1) : wave), qrow = qw0 + l31;
  const bool qvalid = qrow < T;
  const uint16_t* base = qkv + (int64_t)b * T * ld + h * HD;
  const float scale = 0.125f, c2 = scale * LOG2E;

  const int jt_hi = (min(T, q0 + (SPLIT ? 64 : 128)) + KT - 1) / KT;
  const int n = jt_hi - jt_lo;

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
  // Prologue: tile 0 goes to slot 0; the wave's own Q and O rows come by LDS-DMA (whole 128-byte rows: rows_dma, attn_common.h) into its
  // quarters of slots 1 and 2 (three 4 KiB regions per wave do not fit beside tile 0: dO stays a fragment load); tiles 1 and 2 are staged
  // behind the first barrier of the loop, when every wave has its fragments in registers.  Rows beyond T repeat row T - 1 (never stored).
  if (n > 0) stage(0, jt_lo);
  char* qreg = smem + 1 * 2 * TILE + wave * 4096;
  char* oreg = smem + 2 * 2 * TILE + wave * 4096;
  rows_dma(qreg, base, ld, qw0, T - 1, lane);
  rows_dma(oreg, out + (int64_t)b * T * dm + h * HD, dm, qw0, T - 1, lane);
  bf16x8_t qf[4], dof[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) dof[ks] = qvalid ? ld_bf16x8(dout + ((int64_t)b * T + qrow) * dm + h * HD + ks * 16 + hi * 8) : zero_bf16x8();
  const int dsq = doc_start[(int64_t)b * T + min(qrow, T - 1)];
  float Lq = qvalid ? lse[((int64_t)b * nh + h) * T + qrow] : 0.f;  // base-2 LSE
  asm volatile("; row data requested" ::"v"(dof[0]), "v"(dof[1]), "v"(dof[2]), "v"(dof[3]), "v"(Lq), "v"(dsq));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // rows and tile 0 have landed: nothing but LDS-DMA is counted from here on
  float Dq;
  {
    float part_sum = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = rows_frag(qreg, l31, ks, hi);
      const bf16x8_t o8 = rows_frag(oreg, l31, ks, hi);
#pragma unroll
      for (int e = 0; e < 8; ++e) part_sum += bf2f(o8[e]) * bf2f(dof[ks][e]);  // dO is zero for rows beyond T
    }
    float d_lo, d_hi;
    half_pair(part_sum, d_lo, d_hi);
    Dq = d_lo + d_hi;  // published (negated) in the epilogue: a store here would have to be waited for before the counted DMA waits
  }

  f32x16_t dq[2];
  zero16(dq[0]);
  zero16(dq[1]);

  const int ds_lo = __builtin_amdgcn_readlane(dsq, 0), ds_hi = __builtin_amdgcn_readlane(dsq, 31);
  const int jd = min(jt_hi - 1, qw0 / KT);
  const int ja = min(jd, max(jt_lo, ds_lo / KT));
  const int jm = min(jd, max(ja, (ds_hi + KT - 1) / KT));

  // one K | V tile, one 32-key block at a time: S^T, dP^T, dS^T, dQ^T += K^T dS^T
  auto compute = [&](int kv0, const char* sK, const char* sV, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const int c_hi = qrow - kv0 - koff - 4 * hi, c_lo = dsq - kv0 - koff - 4 * hi;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      __builtin_amdgcn_sched_barrier(0);  // one key block's fragments at a time
      bf16x8_t kfr[4], vfr[4], ktr[2][2];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        kfr[ks] = frag_rows(sK, koff + kb * 32 + l31, ks, hi);
        vfr[ks] = frag_rows(sV, koff + kb * 32 + l31, ks, hi);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int db = 0; db < 2; ++db) ktr[db][s2] = frag_cols(sK, db, koff + kb * 32 + s2 * 16 + 4 * hi, lane);
      f32x16_t s, dp;
      zero16(s);
      zero16(dp);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = mfma32(kfr[ks], qf[ks], s);      // S^T[kv][q]
        dp = mfma32(vfr[ks], dof[ks], dp);   // dP^T[kv][q]
      }
      bf16x8_t dsf[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = fast_exp2(__builtin_fmaf(s[r], c2, -Lq));
        if (MASK) {
          const int c = kb * 32 + (r & 3) + 8 * (r >> 2);
          p = (c <= c_hi && c >= c_lo) ? p : 0.f;
        }
        dsf[r >> 3][r & 7] = f2bf(p * (dp[r] - Dq));  // x 1/sqrt(hd) once, in the epilogue
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int db = 0; db < 2; ++db) dq[db] = mfma32(ktr[db][s2], dsf[s2], dq[db]);  // dQ^T[d][q]
    }
  };

  static_assert(NST == 3, "the prologue parks the wave's rows in slots 1 and 2");
  // every wave has its row fragments in registers: slot 1 is free for tile 1 (tile 2 follows at the loop's first step, as always).  The loop's
  // first wait then finds tile 0 landed (the prologue's vmcnt(0)) and at most tile 1's four instructions in flight - its usual count.
  attn_barrier();
  if (n > 1) stage(1, jt_lo + 1);
  int i = 0, slot = 0;
  auto run = [&](int jt_end, auto mask_tag, bool active) {
    for (; jt_lo + i < jt_end; ++i) {
      const int rem = min(NST - 2, n - 1 - i);
      if (NST >= 4 && rem >= 2) attn_wait_vm<2 * DOC_DMA_PER_STAGE>();
      else if (NST >= 3 && rem == 1) attn_wait_vm<DOC_DMA_PER_STAGE>();
      else attn_wait_vm<0>();
      attn_barrier();
      if (i + NST - 1 < n) stage(slot == 0 ? NST - 1 : slot - 1, jt_lo + i + NST - 1);
      if (active) compute((jt_lo + i) * KT, smem + slot * 2 * TILE, smem + slot * 2 * TILE + TILE, mask_tag);
      slot = (slot + 1 == NST) ? 0 : slot + 1;
    }
  };
#ifdef PLM_ATTN_TRACE
  if (trace_t) trace_t[0] = __builtin_amdgcn_s_memrealtime();
#endif
  run(ja, std::true_type{}, false);
  run(jm, std::true_type{}, true);
  run(jd, std::false_type{}, true);
  run(jd + 1, std::true_type{}, true);
  run(jt_hi, std::true_type{}, false);
#ifdef PLM_ATTN_TRACE
  if (trace_t) trace_t[1] = __builtin_amdgcn_s_memrealtime();
#endif

  char* stage_base = smem + (n % NST) * 2 * TILE;  // a slot nobody reads any more (see the forward kernel)
  if (SPLIT) {  // dQ = part 0's sum + part 1's, always in this order
    attn_barrier();
    float* xch = reinterpret_cast<float*>(smem) + (wave & 1) * (32 * 64);
    if (part == 1) {
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[(db * 16 + r) * 64 + lane] = dq[db][r];
    }
    attn_barrier();
    if (part == 1) return;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[db][r] += xch[(db * 16 + r) * 64 + lane];
    stage_base = smem + 32768;
  }
  const RowStage rs{stage_base + wave * 4096, lane};
  if (qvalid && hi == 0) ndelta[((int64_t)b * nh + h) * T + qrow] = -Dq;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4_t ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[e] = f2bf(dq[db][4 * g + e] * scale);
      rs.put(l31, hi, db * 4 + g, ov);
    }
  rs.flush_rot(dqkv + (int64_t)b * T * ld, ld, qw0, T, h * HD, rcos, rsin);  // inverse rotation: gradient w.r.t. the PRE-rotation q
}

template <int NST>
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_doc_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out,
                                                                 const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                                                                 float* __restrict__ ndelta, const float* __restrict__ rcos,
                                                                 const float* __restrict__ rsin, const int32_t* __restrict__ doc_start,
                                                                 const int32_t* __restrict__ header, const int4* __restrict__ items,
                                                                 uint16_t* __restrict__ dqkv, int T, int nh) {
  __shared__ __attribute__((aligned(1024))) char smem[NST * 2 * 8192];  // [stage][K|V]
  ATTN_TRACE_T(tr0);
  const int idx = blockIdx.x / nh, h = blockIdx.x - idx * nh;
  if (idx >= __builtin_amdgcn_readfirstlane(header[0])) return;
  const int4 it = items[idx];
  const int b = __builtin_amdgcn_readfirstlane(it.x), q0 = __builtin_amdgcn_readfirstlane(it.y), jt_lo = __builtin_amdgcn_readfirstlane(it.z);
  const int kc = __builtin_amdgcn_readfirstlane(it.w);
#ifdef PLM_ATTN_TRACE
  unsigned long long tt[2] = {0, 0};
#else
  unsigned long long* tt = nullptr;
#endif
  if (kc >> DOC_KIND_SHIFT) attn_bwd_dq_doc_body<NST, true>(smem, qkv, out, dout, lse, ndelta, rcos, rsin, doc_start, dqkv, T, nh, b, h, q0, jt_lo, tt);
  else attn_bwd_dq_doc_body<NST, false>(smem, qkv, out, dout, lse, ndelta, rcos, rsin, doc_start, dqkv, T, nh, b, h, q0, jt_lo, tt);
  ATTN_TRACE_END(1, idx, kc, tr0, tt[0], tt[1]);
}

// =============================================================================================
// launchers (called from the C ABI entry points in attn.hip when a document mask is given)
// =============================================================================================
void plm_attn_bwd_dkdv_doc(const uint16_t* qkv, const uint16_t* dout, const float* lse, const float* ndelta, const float* rc, const float* rs,
                           const int32_t* header, const int32_t* doc_end, const int4* items_k, uint16_t* dqkv, int64_t B, int64_t T, int64_t nh,
                           hipStream_t s);

void plm_attn_doc_start_from_mask_launch(const uint8_t* mask, int32_t* doc_start, int32_t* status, int64_t B, int64_t T, hipStream_t s) {
  const int64_t rows = B * T;
  hipLaunchKernelGGL(doc_start_from_mask_kernel, dim3((unsigned)plm_cdiv(rows, 4)), dim3(256), 0, s, mask, doc_start, status, rows, (int)T);
}

void plm_attn_doc_plan_launch(const int32_t* doc_start, int32_t* plan, int64_t B, int64_t T, int64_t nh, int split_min_q, hipStream_t s) {
  const unsigned blocks = (unsigned)plm_cdiv(B * T, 1024) + 1;  // doc_end[] blocks + the one block that builds and sorts the item lists
  hipLaunchKernelGGL(attn_doc_plan_kernel, dim3(blocks), dim3(1024), 0, s, doc_start, plan, (int)B, (int)T, (int)nh, split_min_q);
}

void plm_attn_fwd_doc(const uint16_t* qkv, const int32_t* doc_start, const int32_t* plan, uint16_t* out, float* lse, int64_t B, int64_t T,
                      int64_t nh, hipStream_t s) {
  const DocPlan p = doc_plan_view(plan, B, T);
  const dim3 grid((unsigned)(doc_plan_cap(B, T) * nh)), block(256);
  hipLaunchKernelGGL((attn_fwd_doc_kernel<3>), grid, block, 0, s, qkv, doc_start, p.header, p.items_q, out, lse, (int)T, (int)nh);
}

void plm_attn_bwd_doc(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* lse, float* delta, const float* rc,
                      const float* rs, const int32_t* doc_start, const int32_t* plan, uint16_t* dqkv, int64_t B, int64_t T, int64_t nh,
                      hipStream_t s) {
  const DocPlan p = doc_plan_view(plan, B, T);
  const dim3 grid((unsigned)(doc_plan_cap(B, T) * nh)), block(256);
  // dQ first: it computes delta[b,h,q] for its queries and publishes -delta for the dK/dV kernel
  hipLaunchKernelGGL((attn_bwd_dq_doc_kernel<3>), grid, block, 0, s, qkv, out, dout, lse, delta, rc, rs, doc_start, p.header, p.items_q, dqkv,
                     (int)T, (int)nh);
  plm_attn_bwd_dkdv_doc(qkv, dout, lse, delta, rc, rs, p.header + 1, p.doc_end, p.items_k, dqkv, B, T, nh, s);
}
