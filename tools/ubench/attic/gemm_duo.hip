// Two persistent 4-wave workgroups per CU on 256x128 output tiles:  C[M,N] (bf16) = alpha * A[M,K] · B[N,K]^T  with the
// fused epilogues of gemm_nt_big_kernel (SwiGLU forward, SwiGLU backward, RoPE).
//
// Why a second persistent NT family (round 4).  gemm_nt_big_kernel owns a CU with ONE 8-wave workgroup, so while that workgroup
// is in a tile's epilogue nothing on the CU issues MFMAs, and while it is in the K loop the CU's load / store path idles.  A CU
// pulls ~25 GB/s from HBM whatever the rest of the chip does (MI355X_MICROARCH.md: ~10-11 B/cycle/CU), so an epilogue that moves
// 512 KB per 256x256 tile (SwiGLU backward: 256 KB of gate / up in, 256 KB of d(gate) / d(up) out) takes 21 us per tile on every
// CU at once and cannot be hidden by staggering CUs (measured).  With K = 768 the K loop of that tile is
// 20 us: the launch alternates between an MFMA-bound and a memory-bound half.  Here each CU holds TWO independent workgroups,
// one wave per SIMD each, started half a tile period apart: one workgroup's epilogue (loads, sigmoid arithmetic, LDS
// transposition, stores) runs under the other's K loop.  The same holds, at a smaller scale, for every short-K launch of the
// step (K = 768: 12 K-tiles per tile).
//
//   * 256x128 tile, 256 threads = 4 waves as 2 (M) x 2 (N); wave tile 128x64 of v_mfma_f32_16x16x32_bf16 accumulators - the
//     wave tile, fragment reads and MFMA order of gemm_nt_big_kernel<256,256,2,4>, so a wave's K-loop instruction stream is the same.
//   * LDS: 72 KiB per workgroup (two fit a CU's 160 KiB): a 3-slot ring of A half-tiles (128 rows x 128 B = 16 KiB) and a
//     3-slot ring of B half-tiles (64 rows = 8 KiB).  Every half-tile is read from LDS exactly once (phase 1: A0, B0; phase 2:
//     B1; phase 3: A1; the fragments stay in registers for the second quadrant that needs them), so a slot is refilled as soon
//     as the barrier behind its read phase has been passed:
//         phase 1(g): A0(g+1)      phase 2(g): A1(g+1), B1(g+1)      phase 3(g): B0(g+2)      phase 4(g): -
//     Each half-tile has 4-6 phases to land (gemm_nt_big_kernel: 6-7); the partner workgroup covers the rest.  Two counted waits
//     per K-tile (vmcnt(8) before phase 1, vmcnt(6) before phase 2) and three barriers, as in the 8-wave kernel.
//   * The epilogue's transposition scratch is the ring slot that is free between the last K-tile of a tile and phase 1 of the
//     next one (the slot of A1(last), refilled by A0 of the next tile's second K-tile).  A wave's DMA instructions for an A
//     half-tile write exactly the 4 KiB it uses as scratch, so no barrier is needed between the epilogue and that refill.
//   * Epilogue stores are counted into the first K-tile's waits of the next tile (vmcnt retires in order), as in the 8-wave kernel.
//
// Requirements (checked by the launchers): K % 64 == 0, 16-byte aligned operands and C rows, N % 8 == 0 (fused epilogues:
// N % 128 == 0).
#include <stdlib.h>

#include <initializer_list>
#include <type_traits>

#include "plm_device.h"

typedef __attribute__((address_space(3))) void duo_lds_void_t;
typedef const __attribute__((address_space(1))) void duo_gbl_void_t;

__device__ __forceinline__ void duo_dma16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((duo_gbl_void_t*)gsrc, (duo_lds_void_t*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int duo_swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
template <int N>
__device__ __forceinline__ void duo_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// vmcnt(flag ? N1 : N0), flag wave-uniform (see gemm_big.hip)
template <int N0, int N1>
__device__ __forceinline__ void duo_wait_vm_sel(int flag) {
  asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_waitcnt vmcnt(%1)\n\ts_branch 2f\n1:\n\ts_waitcnt vmcnt(%2)\n2:" ::"s"(flag), "n"(N0), "n"(N1)
               : "memory", "scc");
}
__device__ __forceinline__ void duo_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define DUO_GROUP_M 4
#ifndef PLM_DUO_GLUB_DEPTH
#define PLM_DUO_GLUB_DEPTH 4
#endif

enum { DUO_PLAIN = 0, DUO_GLU = 1, DUO_GLUB = 2, DUO_ROPE = 3 };

struct DuoEpi {
  uint16_t* act;      // GLU: activation output [M, N/2];  GLUB: the saved fc1 output [M, 2N] (read-only)
  int64_t ldact;
  const float* rcos;  // ROPE: fp32 [T, 32] tables
  const float* rsin;
  int T, rope_cols;
};

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_duo_kernel(const uint16_t* __restrict__ A, int64_t lda, const uint16_t* __restrict__ B,
                                                             int64_t ldb, uint16_t* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                             const float* __restrict__ alpha_dev, int tiles_m, int tiles_n, DuoEpi ea,
                                                             int stagger_ticks, int dbg) {
  constexpr bool GLU = EPI == DUO_GLU, GLUB = EPI == DUO_GLUB, ROPE = EPI == DUO_ROPE;
  constexpr int BM = 256, BN = 128, WN = 2, TM = 128, TN = 64, AH = 64;
  constexpr int A_HT = 128 * 128, B_HT = 64 * 128;  // bytes per half-tile
  constexpr int A_DMA = 4, B_DMA = 2;               // LDS-DMA instructions per wave and half-tile
  constexpr int OFF_B = 3 * A_HT;
  // LDS-DMA instructions younger than the half-tile a wait retires (issue order: ph1 A0', ph2 A1' B1', ph3 B0'')
  constexpr int W_P4 = A_DMA + 2 * B_DMA;  // end of phase 4 -> A0 (and the older B0) of the next K-tile: A1', B1', B0'' in flight
  constexpr int W_P1 = B_DMA + A_DMA;      // end of phase 1 -> B1 (and the older A1) of this K-tile: B0', A0' in flight
  // stores of one tile's epilogue per wave (interior tile): counted into the next tile's first waits
  constexpr int NS = GLUB ? 32 : (GLU ? 24 : 16);
  static_assert(W_P4 + NS < 64, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(1024))) char smem[3 * A_HT + 3 * B_HT];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l15 = lane & 15, q = lane >> 4;
  const int nkt = K / 64;
  const int ntiles = tiles_m * tiles_n;
  uint16_t* const act = ea.act;
  const int64_t ldact = ea.ldact;

  auto tile_origin = [&](int item, int& m0, int& n0) {
    const int group_size = DUO_GROUP_M * tiles_n;
    const int group = item / group_size;
    const int first_m = group * DUO_GROUP_M;
    const int gm = min(tiles_m - first_m, DUO_GROUP_M);
    const int in_group = item - group * group_size;
    m0 = (first_m + in_group % gm) * BM;
    n0 = (in_group / gm) * BN;
  };

  const int first = xcd_remap(blockIdx.x, gridDim.x);
  if (first >= ntiles) return;

  // ---- staging cursors: the A stream (A0, A1 of K-tile g+1) and the B stream (B1 of g+1, B0 of g+2) run at different positions ----
  unsigned oa[2][A_DMA], ob[2][B_DMA];
  const uint16_t* s_ak = A;  // running pointers: tile's row panel + k of the K-tile under the cursor
  const uint16_t* s_bk = B;
  int a_item = first, a_k = 0, b_item = first, b_k = 0;
  auto open_a = [&]() {
    int m0, n0;
    tile_origin(a_item, m0, n0);
    s_ak = A + (int64_t)m0 * lda;
    a_k = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < A_DMA; ++i) {
        const int r = (wave * A_DMA + i) * 8 + (lane >> 3);  // row of the half-tile: this wave fills rows wave*32 .. +31 = its own 4 KiB
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        const int m = min((r / AH) * TM + h * AH + (r % AH), M - 1 - m0);
        oa[h][i] = (unsigned)(((int64_t)m * lda + chunk * 8) * 2);
      }
  };
  auto open_b = [&]() {
    int m0, n0;
    tile_origin(b_item, m0, n0);
    s_bk = B + (int64_t)(GLU ? n0 / 2 : n0) * ldb;  // GLU: the tile's first gate row
    b_k = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < B_DMA; ++i) {
        const int r = (wave * B_DMA + i) * 8 + (lane >> 3);  // 0..63
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        // GLU: half-tile row r = gate row n0/2 + r of the weight for h = 0, up row N/2 + n0/2 + r for h = 1
        const int n = GLU ? h * (N / 2) + r : min((r / 32) * TN + h * 32 + (r % 32), N - 1 - n0);
        ob[h][i] = (unsigned)(((int64_t)n * ldb + chunk * 8) * 2);
      }
  };
  // The cursors never run dry: past the last tile they stay on its last K-tile (re-staged into slots nobody reads again, a few KiB
  // of L2 hits), so the K loop carries no conditionals around its DMA issue and waits; everything is drained before the kernel ends.
  auto advance_a = [&]() {
    a_k += 64;
    s_ak += 64;
    if (a_k >= K) {
      if (a_item + (int)gridDim.x < ntiles) {
        a_item += gridDim.x;
        open_a();
      } else {
        a_k -= 64;
        s_ak -= 64;
      }
    }
  };
  auto advance_b = [&]() {
    b_k += 64;
    s_bk += 64;
    if (b_k >= K) {
      if (b_item + (int)gridDim.x < ntiles) {
        b_item += gridDim.x;
        open_b();
      } else {
        b_k -= 64;
        s_bk -= 64;
      }
    }
  };
  auto issue_a = [&](int h, int slot) {
    char* dst = smem + slot * A_HT + wave * (A_DMA * 1024);
#pragma unroll
    for (int i = 0; i < A_DMA; ++i) duo_dma16(reinterpret_cast<const char*>(s_ak) + oa[h][i], dst + i * 1024);
  };
  auto issue_b = [&](int h, int slot) {
    char* dst = smem + OFF_B + slot * B_HT + wave * (B_DMA * 1024);
#pragma unroll
    for (int i = 0; i < B_DMA; ++i) duo_dma16(reinterpret_cast<const char*>(s_bk) + ob[h][i], dst + i * 1024);
  };
  auto frag16 = [&](const char* ht, int row, int ks) -> bf16x8_t {  // row = 16-row block base + l15; ks = K-step of 32
    return *reinterpret_cast<const bf16x8_t*>(ht + duo_swz(row, ks * 4 + q));
  };

  open_a();
  open_b();
  float alpha = alpha_dev ? *alpha_dev : 1.f;
  asm volatile("; alpha pinned" : "+v"(alpha));  // consumed now: no ordinary load stays pending in hipcc's bookkeeping (see gemm_big.hip)

  // Ring slots of K-tile g (same numbers in the A ring and the B ring): half 0 in s0 = 2g % 3, half 1 in s1 = (2g + 1) % 3;
  // sp = (2g + 2) % 3 is the third one.
  int s0 = 0, s1 = 1, sp = 2;
  // prologue: B0(0), A0(0), A1(0), B1(0), B0(1) - what the steady state has issued before phase 1 of K-tile 0
  issue_b(0, 0);
  issue_a(0, 0);
  issue_a(1, 1);
  advance_a();
  issue_b(1, 1);
  advance_b();
  issue_b(0, 2);
  // The second workgroup of every CU starts half a tile period late, so that one workgroup's epilogue meets the other's K loop.
  // Dispatch order fills every CU once before any CU gets its second workgroup (observed, tools/ubench/duo_census.hip; a wrong
  // guess costs speed only).
  if (stagger_ticks > 0 && blockIdx.x >= (gridDim.x >> 1)) {
    const uint64_t t0 = __builtin_readcyclecounter();  // s_memtime: shader clock
    while (__builtin_readcyclecounter() - t0 < (uint64_t)stagger_ticks * 100) __builtin_amdgcn_s_sleep(32);
  }
  duo_wait_vm<W_P4>();
  duo_barrier();

  int credit_i = 0;  // NS stores of the previous tile's epilogue are still counted by vmcnt (scalar operand of duo_wait_vm_sel)
  for (int item = first; item < ntiles; item += gridDim.x) {
    f32x4_t acc4[8][4];  // [16-row block of the wave's 128 rows: A half f / 4, block f % 4][16-column block of its 64 columns: B half j / 2]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc4[i][j][r] = 0.f;

    for (int kt = 0; kt < nkt; ++kt) {
      const char* a0 = smem + s0 * A_HT;
      const char* a1 = smem + s1 * A_HT;
      const char* b0 = smem + OFF_B + s0 * B_HT;
      const char* b1 = smem + OFF_B + s1 * B_HT;
      bf16x8_t a6[4][2], b06[2][2], b16[2][2];  // fragments: [16-row block][K-step of 32]

      // ---- phase 1: quadrant (A0, B0); stage A0 of the next K-tile into the third slot
      issue_a(0, sp);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < 2; ++j) b06[j][ks] = frag16(b0, wn * 32 + j * 16 + l15, ks);
#pragma unroll
        for (int f = 0; f < 4; ++f) a6[f][ks] = frag16(a0, wm * AH + f * 16 + l15, ks);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc4[f][j] = mfma16(b06[j][ks], a6[f][ks], acc4[f][j]);
      // The previous tile's epilogue stores are younger than everything this wait retires and OLDER than the A0 issued a few lines up,
      // which the phase-4 wait retires: they can be counted in here only (gemm_nt_big_kernel prefetches two K-tiles ahead and carries
      // the credit through the whole first K-tile).
      duo_wait_vm_sel<W_P1, W_P1 + NS>(credit_i);
      credit_i = 0;
      duo_barrier();

      // ---- phase 2: quadrant (A0, B1); stage A1, B1 of the next K-tile into the slots phase 1 read
      issue_a(1, s0);
      advance_a();
      issue_b(1, s0);
      advance_b();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int c = 0; c < 2; ++c) b16[c][ks] = frag16(b1, wn * 32 + c * 16 + l15, ks);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc4[f][2 + c] = mfma16(b16[c][ks], a6[f][ks], acc4[f][2 + c]);
      duo_barrier();  // A1 of this K-tile is older than the B1 the previous wait retired: no vmcnt wait due here

      // ---- phase 3: quadrant (A1, B1); stage B0 two K-tiles ahead into the slot phase 2 read.  Phase 4 reads nothing from LDS:
      // no barrier between the two.
      issue_b(0, s1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < 4; ++f) a6[f][ks] = frag16(a1, wm * AH + f * 16 + l15, ks);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc4[4 + f][2 + c] = mfma16(b16[c][ks], a6[f][ks], acc4[4 + f][2 + c]);

      // ---- phase 4: quadrant (A1, B0), everything in registers
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc4[4 + f][j] = mfma16(b06[j][ks], a6[f][ks], acc4[4 + f][j]);
      duo_wait_vm<W_P4>();
      duo_barrier();
      // next K-tile: 2 (g + 1) % 3 = the old third slot
      const int o0 = s0;
      s0 = sp;
      sp = s1;
      s1 = o0;
    }

    // ---- epilogue: 32-row x 64-col pieces through this wave's 4 KiB of the free A slot (sp: refilled in phase 1 of the next K-tile
    // by THIS wave's DMA instructions only) ----
    // Everything the epilogue needs is derived from values pinned HERE: hipcc otherwise hoists the tile origin and the per-lane
    // address arithmetic above the K loop, where every live register beyond the accumulators and fragments is a spill.
    int item_e = item, lane_e = lane;
    asm volatile("; epilogue starts" : "+s"(item_e), "+v"(lane_e));
    const int lane = lane_e, l15 = lane_e & 15, q = lane_e >> 4;
    int m0, n0;
    tile_origin(item_e, m0, n0);
    char* epi = smem + sp * A_HT + wave * 4096;
    if constexpr (GLUB) {
      // SwiGLU backward on the wave's 128 x 64 block of d(act) (see gemm_nt_big_kernel), 16 rows ("batch") at a time.  vmcnt retires in
      // issue order, so a batch written as `loads -> wait -> arithmetic -> stores` makes every batch's loads wait for the previous
      // batch's stores to be acknowledged: eight serial HBM round trips per tile with 4 KiB per wave in flight (what
      // gemm_nt_big_kernel's GLUB epilogue does: 21 us per 256x256 tile whatever the rest of the chip is doing).  Here the epilogue
      // is two passes.  Pass 1: the gate / up values of FOUR batches are in flight (16 KiB per wave; the loads of batch b + 4 are
      // issued when batch b has been consumed) and the results d(gate) / d(up) replace the accumulators they were computed from, in
      // registers - no store is issued, so no load ever waits for one.  Pass 2: transposition through the scratch and all 32 stores.
      const uint16_t* u = act;
      bf16x4_t xv[8][4], zv[8][4];
      // `after`: a value the loads must not be scheduled above (their row index is routed through an asm statement that consumes it:
      // hipcc otherwise hoists all 64 loads to the top of the pass - sched_barrier does not stop its IR-level motion - and spills)
      auto ld_batch = [&](int b, int after) {  // b is a constant after unrolling; after: batch whose results must exist first (-1: none)
        int gm_l = min(m0 + wm * TM + (b >> 2) * AH + ((b >> 1) & 1) * 32 + (b & 1) * 16 + l15, M - 1);
        if (dbg & 1) gm_l &= 255;  // ablation: cache-resident rows
        if (after >= 0)
          asm volatile("; loads of batch %1 from here" : "+v"(gm_l) : "n"(b), "v"(acc4[after][0]), "v"(acc4[after][1]), "v"(acc4[after][2]), "v"(acc4[after][3]));
        const uint16_t* row = u + (int64_t)gm_l * ldact + n0 + wn * TN + 4 * q;  // N % 128 == 0: the columns are always in range
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {  // pos = bq * 2 + sc: hidden units bq * 32 + sc * 16 + 4 q .. + 3 of the wave's 64
          xv[b][pos] = ld_bf16x4(row + pos * 16);
          zv[b][pos] = ld_bf16x4(row + N + pos * 16);
        }
      };
      constexpr int DEPTH = PLM_DUO_GLUB_DEPTH;  // batches of gate / up values in flight
#pragma unroll
      for (int b = 0; b < DEPTH; ++b) ld_batch(b, -1);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
          bf16x4_t dxo, dzo;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float gf = bf2f(f2bf(acc4[b][pos][e] * alpha));  // d(act) as the GEMM would have stored it
            const float xf = bf2f(xv[b][pos][e]), zf = bf2f(zv[b][pos][e]);
            const float sig = (dbg & 8) ? xf : plm_sigmoid(xf);  // ablation: no exp / rcp
            const bf16_t sv = f2bf(xf * sig);
            const bf16_t ds = f2bf(gf * zf);
            dzo[e] = f2bf(gf * bf2f(sv));
            dxo[e] = f2bf(bf2f(ds) * (sig * (1.f + xf * (1.f - sig))));
          }
          // the results take the place of the accumulators they came from: dwords 0-1 = d(gate), 2-3 = d(up)
          const u32x2_t dxw = __builtin_bit_cast(u32x2_t, dxo), dzw = __builtin_bit_cast(u32x2_t, dzo);
          acc4[b][pos][0] = __uint_as_float(dxw[0]);
          acc4[b][pos][1] = __uint_as_float(dxw[1]);
          acc4[b][pos][2] = __uint_as_float(dzw[0]);
          acc4[b][pos][3] = __uint_as_float(dzw[1]);
        }
        if (b + DEPTH < 8) ld_batch(b + DEPTH, b);
      }
#pragma unroll
      for (int b = 0; b < 8; ++b) {  // d(gate) in scratch rows 0-15, d(up) in rows 16-31
        // (store addresses derive from a value pinned here: computed up front - as hipcc does, unasked - they are 64 registers)
        int lane = lane_e;
        asm volatile("; stores of batch %1 from here" : "+v"(lane) : "n"(b));
        const int l15 = lane & 15, q = lane >> 4;
        const int mrow = m0 + wm * TM + (b >> 2) * AH + ((b >> 1) & 1) * 32 + (b & 1) * 16;
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
          const int c = pos * 2 + (q >> 1);
          const u32x2_t dxw = {__float_as_uint(acc4[b][pos][0]), __float_as_uint(acc4[b][pos][1])};
          const u32x2_t dzw = {__float_as_uint(acc4[b][pos][2]), __float_as_uint(acc4[b][pos][3])};
          *reinterpret_cast<u32x2_t*>(epi + l15 * 128 + ((c ^ (l15 & 7)) << 4) + (q & 1) * 8) = dxw;
          *reinterpret_cast<u32x2_t*>(epi + (16 + l15) * 128 + ((c ^ (l15 & 7)) << 4) + (q & 1) * 8) = dzw;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int c = it * 64 + lane;
          const int row = c >> 3, ch = c & 7;  // rows 0-15: d(gate), 16-31: d(up) of output row row & 15
          const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(epi + row * 128 + ((ch ^ (row & 7)) << 4));
          const int gm = mrow + (row & 15);
          if (gm < M && !(dbg & 2)) st_bf16x8(C + (int64_t)((dbg & 4) ? (gm & 255) : gm) * ldc + (row >> 4) * N + n0 + wn * TN + ch * 8, v);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    // RoPE: the cos / sin chunks of piece mf + 1 are requested before the stores of piece mf are issued, for the same reason (one
    // table buffer: a piece's rotation is done before the next request overwrites it)
    f32x4_t rc[4], rs[4];
    auto ld_tables = [&](int mf) {  // mf is a constant after unrolling
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int c = it * 64 + lane;
        const int gm = min(m0 + wm * TM + (mf >> 1) * AH + (mf & 1) * 32 + (c >> 3), M - 1);
        const int gn = n0 + wn * TN + (c & 7) * 8;
        const int tab = (gm % ea.T) * 32 + ((gn & 63) >> 1);  // four pairs of head dims (gn % 64) / 2 .. + 3 at position gm % T
        rc[it] = *reinterpret_cast<const f32x4_t*>(ea.rcos + tab);
        rs[it] = *reinterpret_cast<const f32x4_t*>(ea.rsin + tab);
      }
    };
#ifndef PLM_DUO_ROPE_MODE
#define PLM_DUO_ROPE_MODE 0
#endif
    if constexpr (ROPE && PLM_DUO_ROPE_MODE == 0) {
      __builtin_amdgcn_sched_barrier(0);  // not into the K loop
      ld_tables(0);
    }
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) {
      if constexpr (GLUB) continue;
      const int mrow0 = m0 + wm * TM + (mf >> 1) * AH + (mf & 1) * 32;
      const int fb = (mf >> 1) * 4 + (mf & 1) * 2;  // first 16-row accumulator block of this piece
#pragma unroll
      for (int bq = 0; bq < 2; ++bq)
#pragma unroll
        for (int sr = 0; sr < 2; ++sr)
#pragma unroll
          for (int sc = 0; sc < 2; ++sc) {  // this lane holds columns 4 q .. 4 q + 3 of the 16-column block
            bf16x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(acc4[fb + sr][bq * 2 + sc][e] * alpha);
            const int row = sr * 16 + l15, c = bq * 4 + sc * 2 + (q >> 1);
            *reinterpret_cast<bf16x4_t*>(epi + row * 128 + ((c ^ (row & 7)) << 4) + (q & 1) * 8) = o;
          }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      bf16x8_t vv[4];
      if constexpr (ROPE && PLM_DUO_ROPE_MODE == 1) ld_tables(mf);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int c = it * 64 + lane;
        const int row = c >> 3, ch = c & 7;
        vv[it] = *reinterpret_cast<const bf16x8_t*>(epi + row * 128 + ((ch ^ (row & 7)) << 4));
        if constexpr (ROPE) {
          // Rotated on every path and selected bit-wise (v columns pass through): a table load that stays "pending" in hipcc's
          // bookkeeping on a not-taken branch becomes a draining s_waitcnt vmcnt(0) where the next tile zeroes its accumulators.
          // (Consuming the tables with an empty asm statement instead left the youngest load's last lanes un-waited: sporadic wrong
          // chunks at lanes 48-63, tools/_dbg notes in profiles/r04_duo_notes.txt.)
          const u32x4_t rot = __builtin_bit_cast(u32x4_t, rope8(vv[it], rc[it], rs[it], 1.f)), raw = __builtin_bit_cast(u32x4_t, vv[it]);
          const unsigned keep = (n0 + wn * TN + ch * 8 < ea.rope_cols) ? 0xffffffffu : 0u;
          u32x4_t sel;
#pragma unroll
          for (int e = 0; e < 4; ++e) sel[e] = (rot[e] & keep) | (raw[e] & ~keep);
          // ... and materialised HERE: sunk into the `gm < M` branch of the store below, the rotation (and the wait for its tables) is
          // skipped on the other path and the tables stay pending there
          asm volatile("" : "+v"(sel));
          vv[it] = __builtin_bit_cast(bf16x8_t, sel);
        }
      }
      if constexpr (ROPE) {  // the tables of this piece are consumed: request the next piece's before this piece's stores
        __builtin_amdgcn_sched_barrier(0);  // not above the rotation that consumes the current tables (one buffer)
        if (mf < 3 && PLM_DUO_ROPE_MODE == 0) ld_tables(mf + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int c = it * 64 + lane;
        const int row = c >> 3, ch = c & 7;
        const int gm = mrow0 + row;
        const int gn = GLU ? (ch >> 2) * (N / 2) + n0 / 2 + wn * 32 + (ch & 3) * 8 : n0 + wn * TN + ch * 8;
        if (gm < M && gn < N) st_bf16x8(C + (int64_t)gm * ldc + gn, vv[it]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (GLU) {  // act for the 32 rows x 32 hidden units of this piece, through the same scratch (64-byte rows)
#pragma unroll
        for (int sr = 0; sr < 2; ++sr)
#pragma unroll
          for (int sc = 0; sc < 2; ++sc) {
            bf16x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = plm_swiglu_bf16(f2bf(acc4[fb + sr][sc][e] * alpha), f2bf(acc4[fb + sr][2 + sc][e] * alpha));
            const int row = sr * 16 + l15, c = sc * 2 + (q >> 1);
            *reinterpret_cast<bf16x4_t*>(epi + row * 128 + ((c ^ (row & 7)) << 4) + (q & 1) * 8) = o;
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int c = it * 64 + lane;
          const int row = c >> 2, ch = c & 3;
          const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(epi + row * 128 + ((ch ^ (row & 7)) << 4));
          const int gm = mrow0 + row;
          if (gm < M) st_bf16x8(act + (int64_t)gm * ldact + n0 / 2 + wn * 32 + ch * 8, v);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    const bool credit = m0 + BM <= M && n0 + BN <= N;  // interior tile: every wave issued exactly NS stores
    credit_i = __builtin_amdgcn_readfirstlane(credit ? 1 : 0);
  }
  duo_wait_vm<0>();  // the re-staged tail loads write this workgroup's LDS: they must have landed before the last wave leaves
}

// ---------------------------------------------------------------------------------------------
// launchers (called from gemm_big.hip's fused launchers and from plm_gemm_bf16_nt_ws, variant 7)
// ---------------------------------------------------------------------------------------------
int plm_persistent_slots();  // gemm_big.hip: CUs minus the reserve

static bool duo_aligned16(std::initializer_list<const void*> ptrs) {
  uintptr_t v = 0;
  for (const void* p : ptrs) v |= reinterpret_cast<uintptr_t>(p);
  return (v & 15) == 0;
}

// Start offset of every CU's second workgroup, in units of 100 shader cycles: half of one tile's K loop at ~5 TFLOP/s per CU and
// ~2 GHz, i.e. 256 * 128 * K * 2 flop / 5e12 / 2 -> K / 153 us; PLM_DUO_STAGGER_US overrides (tests / A-B runs; 0 = no offset).
static int duo_stagger_ticks(int64_t K) {
  const PlmEnv& e = plm_env();
  const double us = e.duo_stagger_us >= 0 ? e.duo_stagger_us : (double)K / 153.0;
  return (int)(us * 20.0);  // 100 cycles at ~2 GHz = 0.05 us
}

// epi: DUO_PLAIN / DUO_GLU / DUO_GLUB / DUO_ROPE.  Returns false when the shape does not qualify.
bool plm_launch_gemm_nt_duo(int epi, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, uint16_t* C, int64_t ldc, int64_t M,
                            int64_t N, int64_t K, const float* alpha_dev, uint16_t* act, int64_t ldact, const float* rcos,
                            const float* rsin, int64_t T, int64_t rope_cols, hipStream_t s) {
  if (K % 64 != 0 || N % 8 != 0 || M < 512 || N < 128 || lda % 8 != 0 || ldb % 8 != 0 || ldc % 8 != 0) return false;
  if (!duo_aligned16({A, B, C})) return false;
  if (epi == DUO_GLU && (N % 128 != 0 || ldact % 8 != 0 || !duo_aligned16({act}))) return false;
  if (epi == DUO_GLUB && (N % 128 != 0 || ldact % 4 != 0 || (reinterpret_cast<uintptr_t>(act) & 7) != 0)) return false;
  if (epi == DUO_ROPE && !duo_aligned16({rcos, rsin})) return false;
  const int tm = (int)plm_cdiv(M, 256), tn = (int)plm_cdiv(N, 128);
  const int slots = 2 * plm_persistent_slots();
  const int nt_ = tm * tn;
  const dim3 grid(nt_ < slots ? nt_ : slots), block(256);
  const DuoEpi ea{act, ldact, rcos, rsin, (int)T, (int)rope_cols};
  const int stg = nt_ > slots / 2 ? duo_stagger_ticks(K) : 0;
  const int dbg = plm_env().duo_dbg;  // timing-only ablations of the SwiGLU-backward epilogue (PLM_DUO_DBG, see the kernel; 0 in any real run)
#define DUO_LAUNCH(E) \
  hipLaunchKernelGGL((gemm_nt_duo_kernel<E>), grid, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, alpha_dev, tm, tn, ea, stg, dbg)
  if (epi == DUO_GLU) DUO_LAUNCH(DUO_GLU);
  else if (epi == DUO_GLUB) DUO_LAUNCH(DUO_GLUB);
  else if (epi == DUO_ROPE) DUO_LAUNCH(DUO_ROPE);
  else DUO_LAUNCH(DUO_PLAIN);
#undef DUO_LAUNCH
  return true;
}
