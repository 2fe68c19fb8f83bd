// Persistent, deep-pipelined bf16 NT GEMM for gfx950:  C[M,N] (bf16) = alpha * A[M,K] · B[N,K]^T.
//
// Why a second NT kernel: with 128x128 tiles one K-step moves 32 KB into LDS per 0.25 us of MFMA work
// (~33 TB/s chip-wide at full rate = the whole L2 bandwidth), so gemm_nt_dma_kernel is L2-latency /
// bandwidth bound at 600-1000 TFLOP/s.  This kernel doubles the flops per staged byte and keeps
// ~1.5 K-tiles of LDS-DMA in flight:
//
//   * 256x256 (or 256x128) output tile, 512 threads = 8 waves, wave tile 128x64 (or 64x64) of
//     v_mfma_f32_32x32x16_bf16 accumulators; ONE workgroup per CU, persistent over tiles
//     (tile = step * gridDim + xcd_remap(block)), so the DMA stream never drains between tiles.
//   * The K-tile (BK = 64) of each operand is split into two HALF-TILES chosen so that every wave needs
//     half h of A and half h of B for quadrant (h_a, h_b) of its accumulators.  A K-tile is consumed in
//     four phases  (A0,B0) (A0,B1) (A1,B1) (A1,B0);  phase j also issues the LDS-DMA of half-tile j
//     (order A0,B0,B1,A1) of the NEXT K-tile into the other stage.  Each half-tile therefore has >= 3
//     phases (>= 24 MFMAs per wave) to land, and a phase ends with a COUNTED s_waitcnt vmcnt(N) that only
//     waits for the half-tile the next phase reads, then one s_barrier.
//   * LDS: 2 stages x 4 half-tiles (128-byte rows, XOR swizzle applied on the DMA source address, as in
//     gemm.hip) + a 4 KiB per-wave epilogue scratch: C leaves as full 128-byte row segments.
//
// Requirements (checked by the dispatcher in gemm.hip): bf16 C, no accumulate, K % 64 == 0, N % 8 == 0,
// 16-byte aligned C rows.
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "plm_device.h"

#include <initializer_list>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

__device__ __forceinline__ void big_dma16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_void_t*)gsrc, (lds_void_t*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int big_swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int N>
__device__ __forceinline__ void wait_vm() {
#ifdef PLM_DBG_DRAIN  // the self-check build of tests/test_perf_guard_gpu.py (tools/perf_guard_selfcheck.sh): every counted wait drains the DMA ring
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
// vmcnt(flag ? N1 : N0) with a wave-uniform flag: s_waitcnt only takes an immediate, and the if / else form costs hipcc six scalar
// instructions and two branches per wait (it routes the arms through a mask register)
template <int N0, int N1>
__device__ __forceinline__ void wait_vm_sel(int flag) {
#ifdef PLM_DBG_DRAIN
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  return;
#endif
  asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_waitcnt vmcnt(%1)\n\ts_branch 2f\n1:\n\ts_waitcnt vmcnt(%2)\n2:" ::"s"(flag), "n"(N0), "n"(N1)
               : "memory", "scc");
}
// three-way form: flag 0 -> N0, 1 -> N1, anything else -> N2
template <int N0, int N1, int N2>
__device__ __forceinline__ void wait_vm_sel3(int flag) {
#ifdef PLM_DBG_DRAIN
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  return;
#endif
  asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_waitcnt vmcnt(%1)\n\ts_branch 3f\n1:\n\ts_cmp_lg_u32 %0, 1\n\ts_cbranch_scc1 2f\n\ts_waitcnt vmcnt(%2)\n\t"
               "s_branch 3f\n2:\n\ts_waitcnt vmcnt(%3)\n3:" ::"s"(flag), "n"(N0), "n"(N1), "n"(N2)
               : "memory", "scc");
}
// all of this wave's LDS reads have returned, then the workgroup barrier (never drains VMEM)
__device__ __forceinline__ void phase_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define BIG_GROUP_M 4

static int g_num_cus = 0;
static int g_cu_reserve = 0;  // CUs left free for concurrent kernels (RCCL collectives during backward)

// number of persistent workgroups to launch: one per CU minus the reserve
static int persistent_slots() {
  const int n = g_num_cus - g_cu_reserve;
  return n < 8 ? 8 : n;
}

// The persistent GEMMs launch one workgroup per CU with a static tile schedule.  A concurrent kernel that occupies
// some CUs (RCCL's all-reduce on the side stream) would push the displaced workgroups into a second round; leaving
// `n` CUs free avoids that.  Process-wide setting; 0 restores the full chip.
extern "C" int plm_set_cu_reserve(int n) {
  if (n < 0 || n > 128) {
    plm_set_error("plm_set_cu_reserve: n=%d out of range 0..128", n);
    return PLM_E_INVALID;
  }
  g_cu_reserve = n;
  return PLM_OK;
}

static bool ensure_num_cus() {
  if (g_num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return true;
}

static double round_efficiency(int64_t tiles, int slots) {
  const int64_t rounds = (tiles + slots - 1) / slots;
  return (double)tiles / (double)(rounds * slots);
}

// Hybrid work items (HYB): the tiles of the first `rfull` tile rows take the whole contraction and write bf16 C (whole
// rounds of the persistent grid); the K-tiles of the remaining tile rows form ONE stream (tile-major, then k) that is
// cut into `nchunks` equal runs of `L` K-tiles, one run per workgroup ("stream-K" for the last, partial round).  A run
// may cross a tile boundary, so it is executed as up to ceil(L/nkt)+1 PIECES; every piece writes raw fp32
// accumulators to slabs[sidx][row - rfull*BM][N] with sidx = chunk - first chunk touching that tile, and
// nt_streamk_reduce_kernel sums a tile's pieces.  nchunks = 0 is the plain schedule.  384 tiles on 256 CUs: 2 rounds
// become 1 + ~0.52.
struct HybridArgs {
  int rfull;
  int nchunks;
  int L;
  float* slabs;
};

// Schedule (one form; the plain ring, the staggered-barrier form and the one-barrier form of round 1 are gone): a half-tile slot
// is refilled as soon as the phase that read it has ended, with the half-tile of the K-tile TWO ahead (phase 1: A1 of kt+1,
// phases 2/3/4: A0/B0/B1 of kt+2), so five LDS-DMA groups (80 KiB for 256x256) are in flight behind every counted wait.
// Measured (profiles/r01_kbench_run19*): global->LDS fill and LDS->MFMA compute each take ~70 % of the kernel alone; the deep
// queue lets them overlap.
// MFMA shape: v_mfma_f32_16x16x32_bf16 (16-row fragments, two K-steps of 32 per K-tile; lane = row l15 of a 16-row block, K-chunk
// q = lane >> 4).  The GEMMs of the step run at the board's power limit, not at an issue limit (tools/power_probe.py: the same launch
// on zero operands is 17-41 % faster); per flop this shape moves a quarter of the accumulator registers of 32x32x16 per instruction
// and measured 7-12 % more flops under the cap (tools/ubench/mfma_power.hip), +4-10 % on every NT shape of the step, for the same
// ds_read_b128 count and bytes and twice the MFMA instructions.
// GLU (256x256 tile only): B is the fused fc1 weight [2h, K] of a SwiGLU MLP (gate rows 0..h-1, up rows h..2h-1, N = 2h).  A tile's
// 256 columns are 128 gate columns + the 128 up columns of the same hidden units: half-tile B0 = gate rows, B1 = up rows, so a wave
// holds gate and up of its 32 hidden units at the same lane / register positions.  The epilogue writes C = fc1 output in its
// ordinary [gate | up] layout (backward needs it) AND act[M, h] = bf16(bf16(silu(gate)) * up) computed from the rounded values -
// the bits plm_swiglu_fwd produces - which removes one read of C and one launch per layer (swiglu_fwd: 69 us, 402 MB).
// GLUB (256x256 tile only): the dX GEMM of fc2 with the SwiGLU backward in its epilogue.  The accumulators are d(act)[M, h]
// (N = h); the epilogue rounds them to bf16 (what the stand-alone GEMM would have stored), loads gate / up of the same rows and
// hidden units from the saved fc1 output `act` (= u [M, 2h], read-only here) and writes C = du [M, 2h]: d(gate) in columns
// 0..h-1, d(up) in columns h..2h-1 - the bits of plm_swiglu_bwd.  d(act) never reaches memory (swiglu_bwd: 113 us, 670 MB).
// ROPE: C is the w_qkv projection [M, 3*nh*64] of M = B*T token rows; the 16-byte chunks of its q | k column blocks (columns
// < rope_cols) are rotated on their way from the transposition scratch to memory - the same rope8() on the same bf16 values as
// rope_qk_kernel, i.e. the same bits, without the extra pass over 2/3 of the projection (31 us, 200 MB per layer).
struct EpiArgs {
  uint16_t* act;        // GLU: activation output [M, N/2];  GLUB: the saved fc1 output [M, 2N] (read-only)
  int64_t ldact;
  const float* rcos;    // ROPE: fp32 [T, 32] tables
  const float* rsin;
  int T, rope_cols;
};

template <class F, int... U>
__device__ __forceinline__ void nt_for_units(F& f, std::integer_sequence<int, U...>) {
  (f(std::integral_constant<int, U>{}), ...);
}
// Epilogue units of a wave tile and where each one leaves (see OVL in the kernel): NA 32-row pieces x NPASS column passes.
template <int NA, int NBF, int BF0>
struct NtOvlPlan {
  static constexpr int NPASS = (NBF + 1) / 2, NUNIT = NA * NPASS;
  static constexpr int pass_blocks(int u) { const int p0 = (u % NPASS) * 2; return NBF - p0 < 2 ? NBF - p0 : 2; }
  static constexpr int stores(int u) { return 2 * pass_blocks(u); }
  // phase of the last K-tile after which unit u is final: (A0,B0) 1, (A0,B1) 2, (A1,B1) 3, (A1,B0) 4
  static constexpr int ready(int u) {
    const int mf = u / NPASS, p0 = (u % NPASS) * 2, nb = pass_blocks(u);
    const int h[2] = {(2 * mf) / NA, (2 * mf + 1) / NA};       // A halves of its two 16-row blocks
    const bool b[2] = {p0 < BF0, p0 + nb - 1 >= BF0};           // B halves its 32-column blocks belong to
    int r = 0;
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j)
        if (b[j]) {
          const int ph = h[i] == 0 ? (j == 0 ? 1 : 2) : (j == 1 ? 3 : 4);
          r = ph > r ? ph : r;
        }
    return r;
  }
  // slot of unit u: 2 / 3 / 4 = inside that phase of the last K-tile, 5 = behind the loop; units in order of readiness take the earliest
  // free phase behind their ready phase (one unit per phase: they share the wave's 4 KiB scratch)
  static constexpr int slot(int u) {
    bool taken[6] = {false, false, false, false, false, false};
    int slot_u = 5;
    for (int r = 1; r <= 4; ++r)
      for (int v = 0; v < NUNIT; ++v)
        if (ready(v) == r) {
          int sl = 5;
          for (int c = r + 1; c <= 4; ++c)
            if (!taken[c]) {
              sl = c;
              break;
            }
          if (sl < 5) taken[sl] = true;
          if (v == u) slot_u = sl;
        }
    return slot_u;
  }
  static constexpr int stores_in_slot(int sl) {
    int n = 0;
    for (int u = 0; u < NUNIT; ++u)
      if (slot(u) == sl) n += stores(u);
    return n;
  }
  static constexpr int unit_in_slot(int sl) {  // -1: none
    for (int u = 0; u < NUNIT; ++u)
      if (slot(u) == sl) return u;
    return -1;
  }
};
static_assert(NtOvlPlan<2, 3, 2>::slot(0) == 2 && NtOvlPlan<2, 3, 2>::slot(1) == 3 && NtOvlPlan<2, 3, 2>::slot(3) == 4 && NtOvlPlan<2, 3, 2>::slot(2) == 5,
              "256x192: (A0,B0) (A0,B1) (A1,B1) leave in phases 2 3 4, (A1,B0) behind the loop");
static_assert(NtOvlPlan<4, 2, 1>::slot(0) == 3 && NtOvlPlan<4, 2, 1>::slot(1) == 4 && NtOvlPlan<4, 2, 1>::slot(2) == 5 && NtOvlPlan<4, 2, 1>::slot(3) == 5,
              "256x256: the two pieces of A half 0 leave in phases 3 and 4");
static_assert(NtOvlPlan<1, 3, 2>::slot(1) == 4 && NtOvlPlan<1, 3, 2>::slot(0) == 5, "128x192: the B1 pass leaves in phase 4");

template <int BM, int BN, int WM, int WN, bool HYB = false, bool GLU = false, bool GLUB = false, bool ROPE = false>
__global__ __launch_bounds__(512, 2) void gemm_nt_big_kernel(const uint16_t* __restrict__ A, int64_t lda,
                                                             const uint16_t* __restrict__ B, int64_t ldb,
                                                             uint16_t* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                             const float* __restrict__ alpha_dev, int tiles_m, int tiles_n,
                                                             HybridArgs hyb, EpiArgs ea) {
  uint16_t* const act = ea.act;
  const int64_t ldact = ea.ldact;
  static_assert(!(GLU || GLUB) || (BM == 256 && BN == 256 && WN == 4 && !HYB), "GLU epilogues: 256x256 tiles, whole-K items");
  static_assert(!(GLU && GLUB), "one epilogue at a time");
  constexpr int TM = BM / WM, TN = BN / WN;  // wave tile
  static_assert(WM * WN == 8 && (TN == 64 || TN == 96) && (TM == 128 || TM == 64 || TM == 32), "unsupported geometry");
  constexpr int AH = TM / 2;                        // rows of one wave's A half
  constexpr int NA = AH / 16;                       // 16-row MFMA blocks per A half (= 32-row pieces of the wave tile: TM / 32)
  // A wave's B columns split into "half" 0 = its first BF0 32-column fragments and half 1 = its last fragment
  // (TN = 64: 1 + 1; TN = 96: 2 + 1 - the 256x192 tile, whose phases 1 and 4 carry twice the MFMAs of 2 and 3).
  constexpr int BF0 = TN / 32 - 1, NBF = BF0 + 1;
  constexpr int A_ROWS = BM / 2, B0_ROWS = WN * 32 * BF0, B1_ROWS = WN * 32;  // rows per half-tile
  constexpr int A_HT = A_ROWS * 128, B0_HT = B0_ROWS * 128, B1_HT = B1_ROWS * 128;
  constexpr int STAGE = 2 * A_HT + B0_HT + B1_HT;
  constexpr int A_DMA = A_ROWS / 64, B0_DMA = B0_ROWS / 64, B1_DMA = B1_ROWS / 64;  // LDS-DMA instructions per thread per half-tile
  static_assert(B0_DMA >= B1_DMA && B1_DMA >= 1, "half-tiles are whole DMA rounds of the workgroup");
  constexpr int OFF_A0 = 0, OFF_B0 = A_HT, OFF_B1 = A_HT + B0_HT, OFF_A1 = A_HT + B0_HT + B1_HT;
  // LDS-DMA instructions younger than the group a wait retires (see the schedule above)
  constexpr int D_P4 = 2 * A_DMA + B0_DMA + 2 * B1_DMA;  // end of phase 4 -> A0, B0 of kt+1: B1', A1', A0'', B0'', B1'' in flight
  constexpr int D_P1 = 3 * A_DMA + B0_DMA + B1_DMA;      // end of phase 1 -> B1 of kt: A1, A0', B0', B1', A1' in flight
  constexpr int D_P2 = 3 * A_DMA + B0_DMA + B1_DMA;      // end of phase 2 -> A1 of kt: A0', B0', B1', A1', A0'' in flight
  // Offset wave groups: waves w and w+4 share a SIMD.  With every wave on the same schedule both do their
  // DMA issue + LDS reads at the same time and then queue for the matrix pipe: a phase costs overhead + 2 x MFMA time.  Here the
  // second group (waves 4-7) takes each barrier BETWEEN the reads and the MFMAs of a phase instead of behind the MFMAs - same
  // code, same three barriers per K-tile and wave, same slots and waits - so inside every barrier interval group 0 runs
  // reads(p), MFMAs(p) while group 1 runs MFMAs(p-1), reads(p): one wave's reads and DMA issue sit under the other's MFMAs.
  // The C stores of an epilogue (NS per wave when the tile is interior) are YOUNGER than the loads the first
  // K-tile of the next tile waits for; counting them in lets them drain under that K-tile's MFMAs instead of in front
  // of them (vmcnt retires in order, so a plain count would wait for every store).
  constexpr int NS = NA * 2 * NBF * (GLUB ? 2 : 1) + (GLU ? NA * 2 : 0);
  static_assert(D_P1 + NS < 64, "vmcnt is a 6-bit counter");
  // Overlapped epilogue (round 5; PLM_NT_OVL, plain epilogue only).  An epilogue UNIT = (32-row piece mf, column pass p0) of the wave tile.  In the
  // LAST K-tile of a tile a quadrant of the accumulators is final as soon as its phase has run - (A0,B0) after phase 1, (A0,B1) after 2,
  // (A1,B1) after 3, (A1,B0) after 4 - so a unit whose quadrants are all final leaves DURING the following phase: its packed bf16 values
  // go into the wave's transposition scratch before the phase's MFMAs are issued, and are read back and stored behind them, under the
  // matrix pipe's work instead of after the K loop.  256x192: units final after phases 1 / 2 / 3 / 4 -> three of four overlap; 256x256 and
  // 256x128: the two A halves -> half overlaps; 128x192: the B1 pass.  The stores issued inside the K-tile are counted into every later
  // vmcnt wait exactly (OVS_*: cumulative stores in front of each wait, per wave group - group 1 waits BEFORE its phase's MFMAs and
  // stores), and into the next tile's first K-tile through credit value 2 (the B0 half-tile of the next tile's SECOND K-tile is issued in
  // phase 3, after the phase-2 unit's stores: the waits for it may leave NS - OVS_2 stores outstanding, not NS).
#ifndef PLM_NT_OVL
#define PLM_NT_OVL 1
#endif
  constexpr bool OVL = PLM_NT_OVL && !HYB && !GLU && !GLUB && !ROPE;
  using OP = NtOvlPlan<NA, NBF, BF0>;  // unit u = mf * NPASS + pass
  constexpr int NPASS = OP::NPASS, NUNIT = OP::NUNIT;
  constexpr int OVS_2 = OVL ? OP::stores_in_slot(2) : 0, OVS_3 = OVL ? OP::stores_in_slot(3) : 0, OVS_4 = OVL ? OP::stores_in_slot(4) : 0;
  static_assert(OVS_2 + OVS_3 + OVS_4 + (OVL ? OP::stores_in_slot(5) : NS) == NS, "every unit has a slot");
  static_assert(D_P4 + NS < 64, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE + 8 * 4096];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const bool grp1 = (wave >> 2) != 0;  // wave-uniform
  // Waves w and w+4 share a SIMD, and instruction issue is arbitrated by priority, then age: at equal priority the second-dispatched
  // half (waves 4-7) loses every arbitration.  ONE static s_setprio for that half, no per-phase flips: +1.6 % end to end (run 40).
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  const int l15 = lane & 15, q = lane >> 4;  // fragment coordinates: row inside a 16-row block, K-chunk
  // HYB = false instantiations keep the plain schedule free of the stream-K bookkeeping
  const int rfull = HYB ? hyb.rfull : tiles_m;
  const int n_full = rfull * tiles_n;
  const int nkt = K / 64;
  const int g_total = HYB ? (tiles_m - rfull) * tiles_n * nkt : 0;  // K-tiles in the stream-K remainder
  const int ntiles = n_full + (HYB ? hyb.nchunks : 0);              // work items
  char* epi = smem + 2 * STAGE + wave * 4096;

  // A cursor walks this workgroup's PIECES: item (stride gridDim.x), and inside a chunk item the stream position.
  struct Cur {
    int item, g, gend;
  };
  auto cur_init = [&](Cur& c, int item) {
    c.item = item;
    c.g = c.gend = 0;
    if (HYB && item >= n_full && item < ntiles) {
      c.g = (item - n_full) * hyb.L;
      c.gend = min(g_total, c.g + hyb.L);
    }
  };
  // the piece under the cursor (tile origin, K range, slab index; sidx = -1: whole-K tile written to C); advances
  auto take = [&](Cur& c, int& m0, int& n0, int& kbeg, int& kend, int& sidx) {
    if (!HYB || c.item < n_full) {
      const int group_size = BIG_GROUP_M * tiles_n;
      const int group = c.item / group_size;
      const int first_m = group * BIG_GROUP_M;
      const int gm = min(rfull - first_m, BIG_GROUP_M);
      const int in_group = c.item - group * group_size;
      m0 = (first_m + in_group % gm) * BM;
      n0 = (in_group / gm) * BN;
      kbeg = 0;
      kend = K;
      sidx = -1;
      cur_init(c, c.item + gridDim.x);
    } else {
      const int t = c.g / nkt;
      const int k0 = c.g - t * nkt;
      const int k1 = min(nkt, k0 + (c.gend - c.g));
      m0 = (rfull + t / tiles_n) * BM;
      n0 = (t % tiles_n) * BN;
      kbeg = k0 * 64;
      kend = k1 * 64;
      sidx = (c.item - n_full) - (t * nkt) / hyb.L;
      c.g += k1 - k0;
      if (c.g >= c.gend) cur_init(c, c.item + gridDim.x);
    }
  };

  // ---- source pointers of the item being staged (one K-tile of one output tile) ----
  // LDS-DMA addresses: wave-uniform base of the staged K-tile (SGPR pair, running pointers) + per-lane unsigned byte offsets
  // inside the tile's row panel, issued through the builtin (the inline-asm form measured 1 % slower in the 4-phase kernels:
  // two instructions per phase leave hipcc nothing to schedule around).
  unsigned oa[2][A_DMA], ob[2][B0_DMA];
  const uint16_t* s_ab = A;
  const uint16_t* s_bb = B;
  // s_ak / s_bk = s_ab / s_bb + s_k: running pointers of the K-tile under the staging cursor (one 64-bit scalar add per operand
  // and K-tile instead of a sign-extend + shift + add in front of every DMA pair)
  const uint16_t* s_ak = A;
  const uint16_t* s_bk = B;
  auto set_ptrs = [&](int m0, int n0) {
    s_ab = A + (int64_t)m0 * lda;
    s_bb = B + (int64_t)(GLU ? n0 / 2 : n0) * ldb;  // GLU: the tile's first gate row
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < A_DMA; ++i) {
        const int r = (i * 8 + wave) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        const int m = min((r / AH) * TM + h * AH + (r % AH), M - 1 - m0);
        oa[h][i] = (unsigned)(((int64_t)m * lda + chunk * 8) * 2);
      }
#pragma unroll
      for (int i = 0; i < (h ? B1_DMA : B0_DMA); ++i) {
        const int r = (i * 8 + wave) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        const int rw = h ? 32 : 32 * BF0;  // rows of one wave inside this half-tile
        // GLU: half-tile row r (wave r / 32) = gate row n0/2 + r of the weight for h = 0, up row N/2 + n0/2 + r for h = 1
        const int n = GLU ? h * (N / 2) + r : min((r / rw) * TN + h * 32 * BF0 + (r % rw), N - 1 - n0);
        ob[h][i] = (unsigned)(((int64_t)n * ldb + chunk * 8) * 2);
      }
    }
  };
  auto issue_a = [&](int h, char* stage, int k0) {
    char* dst = stage + (h ? OFF_A1 : OFF_A0);
#pragma unroll
    for (int i = 0; i < A_DMA; ++i) {
      (void)k0;
      big_dma16(reinterpret_cast<const char*>(s_ak) + oa[h][i], dst + (i * 8 + wave) * 1024);  // uniform base + zext(lane offset)
    }
  };
  auto issue_b = [&](int h, char* stage, int k0) {
    char* dst = stage + (h ? OFF_B1 : OFF_B0);
#pragma unroll
    for (int i = 0; i < B0_DMA; ++i) {
      if (i >= (h ? B1_DMA : B0_DMA)) continue;
      (void)k0;
      big_dma16(reinterpret_cast<const char*>(s_bk) + ob[h][i], dst + (i * 8 + wave) * 1024);
    }
  };
  auto frag16 = [&](const char* ht, int row, int ks) -> bf16x8_t {  // row = 16-row block base + l15; ks = K-step of 32
    return *reinterpret_cast<const bf16x8_t*>(ht + big_swz(row, ks * 4 + q));
  };

  const int first = xcd_remap(blockIdx.x, gridDim.x);
  if (first >= ntiles) return;
  // staging side: sc = cursor past the piece being staged, (s_k, s_kend) = next K-tile / end of that piece
  Cur sc;
  cur_init(sc, first);
  int s_k = 0, s_kend = 0;
  int s_item = first;  // item of the piece being staged; >= ntiles: nothing left to stage (workgroup-uniform)
  auto open_piece = [&]() {
    s_item = sc.item;
    if (s_item < ntiles) {
      int m0, n0, sp_;
      take(sc, m0, n0, s_k, s_kend, sp_);
      set_ptrs(m0, n0);
      s_ak = s_ab + s_k;
      s_bk = s_bb + s_k;
    } else {  // nothing left: stay on the last K-tile (the ring keeps issuing - see the K loop)
      s_k -= 64;
      s_ak -= 64;
      s_bk -= 64;
    }
  };
  int s_st = 0;  // stage buffer of the K-tile under the staging cursor
  auto advance_staged = [&]() {
    s_k += 64;
    s_ak += 64;
    s_bk += 64;
    s_st ^= 1;
    if (s_k >= s_kend) open_piece();
  };
  open_piece();

  // read alpha and make the compiler consume it NOW: an ordinary global load still "pending" in hipcc's own
  // bookkeeping would make it emit a draining s_waitcnt vmcnt(0) at the first use inside the tile loop
  float alpha = alpha_dev ? *alpha_dev : 1.f;
  asm volatile("; alpha pinned" : "+v"(alpha));

  // prologue: the first K-tile entirely, into stage 0
  {
    issue_a(0, smem, s_k);
    issue_b(0, smem, s_k);
    issue_b(1, smem, s_k);
    issue_a(1, smem, s_k);
    advance_staged();
  }
  {  // plus A0, B0, B1 of the second K-tile (its A1 follows in phase 1 of the first)
    issue_a(0, smem + STAGE, s_k);
    issue_b(0, smem + STAGE, s_k);
    issue_b(1, smem + STAGE, s_k);
    wait_vm<D_P4>();
  }
  phase_barrier();

  int st = 0;
  bool credit = false;  // NS stores of the previous tile's epilogue are still counted by vmcnt
  int credit_i = 0;     // the same flag as a scalar register operand of wait_vm_sel
  Cur cc;
  cur_init(cc, first);
  while (cc.item < ntiles) {
    // plain schedule: the tile origin is only needed by the epilogue - computing it there keeps it out of the K loop's
    // live registers (measured: 5 % on the 256x128 K <= 2304 shapes)
    int m0 = 0, n0 = 0, kbeg = 0, kend = K, split = -1;
    if (HYB) take(cc, m0, n0, kbeg, kend, split);
    const int nk = HYB ? (kend - kbeg) / 64 : nkt;
    f32x4_t acc4[2 * NA][2 * NBF];  // [16-row block of the wave's 2 * AH rows][16-column block of its TN columns]
#pragma unroll
    for (int i = 0; i < 2 * NA; ++i)
#pragma unroll
      for (int j = 0; j < 2 * NBF; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc4[i][j][r] = 0.f;

    // ---- epilogue units (plain epilogue): 32-row piece mf x column pass p0 through this wave's private 4 KiB scratch ----
    auto unit_write = [&](auto utag) {  // accumulators -> packed bf16 -> scratch (lane owns 4 consecutive columns of a row)
      constexpr int u = decltype(utag)::value, mf = u / NPASS, p0 = (u % NPASS) * 2, nb = OP::pass_blocks(u);
#pragma unroll
      for (int bq = 0; bq < nb; ++bq) {
        const int bh = p0 + bq;
#pragma unroll
        for (int sr = 0; sr < 2; ++sr)  // 16-row half of the 32-row piece
#pragma unroll
          for (int sc = 0; sc < 2; ++sc) {  // 16-column half of the 32-column block: this lane holds columns 4 q .. 4 q + 3 of it
            bf16x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(acc4[2 * mf + sr][bh * 2 + sc][e]);
            const int row = sr * 16 + l15, c = bq * 4 + sc * 2 + (q >> 1);
            *reinterpret_cast<bf16x4_t*>(epi + row * 128 + ((c ^ (row & 7)) << 4) + (q & 1) * 8) = o;
          }
      }
    };
    auto unit_store = [&](auto utag, auto interior, int m0, int n0) {  // scratch -> whole row segments (16 bytes per lane, 8 or 4 lanes per row)
      constexpr int u = decltype(utag)::value, mf = u / NPASS, p0 = (u % NPASS) * 2, nb = OP::pass_blocks(u);
      const int mrow0 = m0 + wm * TM + mf * 32;
      bf16x8_t v[2 * nb];
#pragma unroll
      for (int it = 0; it < 2 * nb; ++it) {
        const int c = it * 64 + lane;
        const int row = nb == 2 ? c >> 3 : c >> 2, ch = nb == 2 ? c & 7 : c & 3;
        v[it] = *reinterpret_cast<const bf16x8_t*>(epi + row * 128 + ((ch ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int it = 0; it < 2 * nb; ++it) {
        const int c = it * 64 + lane;
        const int row = nb == 2 ? c >> 3 : c >> 2, ch = nb == 2 ? c & 7 : c & 3;
        const int gm = mrow0 + row;
        const int gn = n0 + wn * TN + p0 * 32 + ch * 8;
        if (decltype(interior)::value || (gm < M && gn < N)) st_c_bf16x8(C + (int64_t)gm * ldc + gn, v[it]);  // interior tile: no predicate, no branch
      }
    };

    // One K-tile.  LASTOVL = the last K-tile of an interior tile with the overlapped epilogue (see OVL): the unit of slot p is written
    // into the scratch in front of phase p's MFMAs and stored behind them; its stores are counted into the waits behind them.
    auto ktile = [&](auto lasttag, int m0, int n0) {
      constexpr bool LASTOVL = decltype(lasttag)::value;
      // The staging cursor never runs dry: when a workgroup is out of K-tiles it re-stages its last one into slots nobody reads
      // again (a few KiB of L2 hits per workgroup), so the DMA issue and the counted waits of the loop carry no conditionals;
      // the loads are drained before the kernel ends.
      const char* cur = smem + st * STAGE;
      bf16x8_t a6[NA][2], b06[2 * BF0][2], b16[2][2];  // fragments: [16-row block][K-step of 32]
      // end of a phase: counted wait for the half-tile the next phase reads (+ the previous epilogue's stores while they are
      // still counted: credit 1 = all NS of them are younger than every load of this K-tile, credit 2 = the previous tile's epilogue
      // was overlapped - `w2` says how many of its stores are younger than THIS wait's target), then the workgroup barrier
      auto end_phase = [&](auto wtag, auto w2tag) {
        constexpr int W = decltype(wtag)::value, W2 = decltype(w2tag)::value;
        if (LASTOVL) wait_vm<W>();  // never the first K-tile of a tile (nk >= 2): no credit; W includes this K-tile's own unit stores
        else if (OVL) wait_vm_sel3<W, W + NS, W + W2>(credit_i);
        else wait_vm_sel<W, W + NS>(credit_i);
        phase_barrier();
      };
      using std::integral_constant;
      auto slot_write = [&](auto sltag) {
        constexpr int u = OP::unit_in_slot(decltype(sltag)::value);
        if constexpr (LASTOVL && u >= 0) unit_write(integral_constant<int, (u >= 0 ? u : 0)>{});
      };
      auto slot_store = [&](auto sltag) {
        constexpr int u = OP::unit_in_slot(decltype(sltag)::value);
        if constexpr (LASTOVL && u >= 0) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          unit_store(integral_constant<int, (u >= 0 ? u : 0)>{}, std::true_type{}, m0, n0);
        }
      };
      // cumulative unit stores in front of a wait of the overlapped K-tile: group 0 waits behind its phase's stores, group 1 in front
      constexpr int S2 = LASTOVL ? OVS_2 : 0, S4 = LASTOVL ? OVS_2 + OVS_3 + OVS_4 : 0;

      // ---- phase 1: quadrant (A0, B0); stage A1 of the next K-tile, then move the cursor on
      issue_a(1, smem + s_st * STAGE, s_k);
      advance_staged();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < 2 * BF0; ++j) b06[j][ks] = frag16(cur + OFF_B0, wn * BF0 * 32 + j * 16 + l15, ks);
#pragma unroll
        for (int f = 0; f < NA; ++f) a6[f][ks] = frag16(cur + OFF_A0, wm * AH + f * 16 + l15, ks);
      }
      if (grp1) end_phase(integral_constant<int, D_P1>{}, integral_constant<int, NS>{});
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < NA; ++f)
#pragma unroll
          for (int j = 0; j < 2 * BF0; ++j) acc4[f][j] = mfma16(b06[j][ks], a6[f][ks], acc4[f][j]);
      if (!grp1) end_phase(integral_constant<int, D_P1>{}, integral_constant<int, NS>{});

      // ---- phase 2: quadrant (A0, B1); stage A0 two K-tiles ahead, into the slot phase 1 just read
      issue_a(0, smem + s_st * STAGE, s_k);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int c = 0; c < 2; ++c) b16[c][ks] = frag16(cur + OFF_B1, wn * 32 + c * 16 + l15, ks);
      slot_write(integral_constant<int, 2>{});
      if (grp1) end_phase(integral_constant<int, D_P2>{}, integral_constant<int, NS>{});
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < NA; ++f)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc4[f][2 * BF0 + c] = mfma16(b16[c][ks], a6[f][ks], acc4[f][2 * BF0 + c]);
      slot_store(integral_constant<int, 2>{});
      if (!grp1) end_phase(integral_constant<int, D_P2 + S2>{}, integral_constant<int, NS>{});

      // ---- phase 3: quadrant (A1, B1); stage B0.  Phase 4 reads nothing new from LDS, so no vmcnt wait is due here.
      issue_b(0, smem + s_st * STAGE, s_k);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < NA; ++f) a6[f][ks] = frag16(cur + OFF_A1, wm * AH + f * 16 + l15, ks);
      slot_write(integral_constant<int, 3>{});
      // B1'' is only issued in phase 4.  First K-tile behind an overlapped epilogue: the B0 half-tile this wait is for was issued behind the
      // phase-2 unit's stores of that tile
      if (grp1) end_phase(integral_constant<int, D_P4 - B1_DMA + S2>{}, integral_constant<int, NS - OVS_2>{});
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < NA; ++f)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc4[NA + f][2 * BF0 + c] = mfma16(b16[c][ks], a6[f][ks], acc4[NA + f][2 * BF0 + c]);
      slot_store(integral_constant<int, 3>{});

      // ---- phase 4: quadrant (A1, B0) (B0 fragments still in registers); stage B1
      issue_b(1, smem + s_st * STAGE, s_k);
      slot_write(integral_constant<int, 4>{});
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < NA; ++f)
#pragma unroll
          for (int j = 0; j < 2 * BF0; ++j) acc4[NA + f][j] = mfma16(b06[j][ks], a6[f][ks], acc4[NA + f][j]);
      slot_store(integral_constant<int, 4>{});
      if (!grp1) end_phase(integral_constant<int, D_P4 + S4>{}, integral_constant<int, NS - OVS_2>{});
      credit = false;
      credit_i = 0;
      st ^= 1;
    };

    // plain schedule: an interior tile with at least two K-tiles runs its last K-tile in the overlapped form; the tile origin is
    // computed where it is needed (and again for the epilogue) so that it stays out of the K loop's live registers
    bool ovl = false;
    if (OVL) {
      Cur peek = cc;
      int pm0, pn0, pk0, pk1, psp;
      take(peek, pm0, pn0, pk0, pk1, psp);
      ovl = __builtin_amdgcn_readfirstlane(nk >= 2 && alpha_dev == nullptr && pm0 + BM <= M && pn0 + BN <= N);  // every wave issues exactly the counted stores
    }
    {
      const int nloop = ovl ? nk - 1 : nk;
      for (int kt = 0; kt < nloop; ++kt) ktile(std::false_type{}, 0, 0);
    }
    if (OVL && ovl) {
      Cur peek = cc;
      int pm0, pn0, pk0, pk1, psp;
      take(peek, pm0, pn0, pk0, pk1, psp);
      ktile(std::true_type{}, pm0, pn0);
    }

    if (!HYB) take(cc, m0, n0, kbeg, kend, split);
    if (HYB && split >= 0) {
      // ---- partial piece: raw fp32 accumulators into slab[sidx] (lane owns a row, 4 consecutive columns) ----
      const int mrem = M - rfull * BM;
      float* slab = hyb.slabs + ((int64_t)split * mrem - (int64_t)rfull * BM) * N;
#pragma unroll
      for (int f = 0; f < 2 * NA; ++f) {  // 16-row block f of the wave tile (A half f / NA, block f % NA inside it: rows f * 16 ...)
        const int gm = m0 + wm * TM + f * 16 + l15;
        if (gm >= M) continue;
#pragma unroll
        for (int j = 0; j < 2 * NBF; ++j) {
          const int gn = n0 + wn * TN + j * 16 + 4 * q;
          if (gn >= N) continue;
          *reinterpret_cast<f32x4_t*>(slab + (int64_t)gm * N + gn) = acc4[f][j];
        }
      }
      continue;
    }
    // ---- epilogue: 32-row x 64-col pieces through this wave's private 4 KiB scratch ----
    // alpha is 1 for every launch of the step but lm_head's dX: the 4 * AF * 2 * NBF * 4 multiplies are taken only when a device scalar
    // was passed (x * 1.0f is exact, so the bits do not depend on which way the branch goes)
    // (HYB: always - with the branch hipcc keeps two copies of the accumulators alive across the slab / tile paths of that instantiation:
    // 256 VGPRs + 59 spills since round 4's conditional, 236 and none without it; found in round 5 by tools/isa_scan.py)
    if (HYB || alpha_dev != nullptr) {
#pragma unroll
      for (int i = 0; i < 2 * NA; ++i)
#pragma unroll
        for (int j = 0; j < 2 * NBF; ++j) acc4[i][j] *= alpha;
    }
    if constexpr (OVL) {
      // plain epilogue, unit by unit; after an overlapped last K-tile only the units that became final in its fourth phase are left
      auto drain = [&](auto utag) {
        constexpr int u = decltype(utag)::value;
        if (ovl && OP::slot(u) != 5) return;
        unit_write(utag);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        unit_store(utag, std::false_type{}, m0, n0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      };
      nt_for_units(drain, std::make_integer_sequence<int, NUNIT>{});
      credit = m0 + BM <= M && n0 + BN <= N;  // interior tile: every wave issued exactly NS stores
      credit_i = __builtin_amdgcn_readfirstlane(credit ? (ovl ? 2 : 1) : 0);
      continue;
    }
#pragma unroll
    for (int mf = 0; mf < NA; ++mf) {  // 32-row pieces = the 16-row blocks 2 mf, 2 mf + 1 of acc4 (block f holds rows f * 16 ... of the wave tile)
      const int mrow0 = m0 + wm * TM + mf * 32;
      if (GLUB) {
        // SwiGLU backward on this 32 x 64 piece of d(act): lane (l15, q) holds, per 16 x 16 block (sr, bq, sc), row sr*16 + l15 and
        // hidden units bq*32 + sc*16 + 4 q .. + 3 of the wave's 64
        const uint16_t* u = act;
#pragma unroll
        for (int sr = 0; sr < 2; ++sr) {  // 16 rows at a time: d(gate) in scratch rows 0-15, d(up) in rows 16-31
          const int gm_l = min(mrow0 + sr * 16 + l15, M - 1);
          const int fi = 2 * mf + sr;
#pragma unroll
          for (int bq = 0; bq < 2; ++bq)
#pragma unroll
            for (int sc = 0; sc < 2; ++sc) {
              const int gn = n0 + wn * TN + bq * 32 + sc * 16 + 4 * q;  // N % 256 == 0: always in range
              const bf16x4_t xv = ld_bf16x4(u + (int64_t)gm_l * ldact + gn);
              const bf16x4_t zv = ld_bf16x4(u + (int64_t)gm_l * ldact + N + gn);
              bf16x4_t dxo, dzo;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float gf = bf2f(f2bf(acc4[fi][bq * 2 + sc][e]));  // d(act) as the GEMM would have stored it
                const float xf = bf2f(xv[e]), zf = bf2f(zv[e]);
                const float sig = plm_sigmoid(xf);
                const bf16_t sv = f2bf(xf * sig);
                const bf16_t ds = f2bf(gf * zf);
                dzo[e] = f2bf(gf * bf2f(sv));
                dxo[e] = f2bf(bf2f(ds) * (sig * (1.f + xf * (1.f - sig))));
              }
              const int c = bq * 4 + sc * 2 + (q >> 1);
              *reinterpret_cast<bf16x4_t*>(epi + l15 * 128 + ((c ^ (l15 & 7)) << 4) + (q & 1) * 8) = dxo;
              *reinterpret_cast<bf16x4_t*>(epi + (16 + l15) * 128 + ((c ^ (l15 & 7)) << 4) + (q & 1) * 8) = dzo;
            }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int c = it * 64 + lane;
            const int row = c >> 3, ch = c & 7;  // rows 0-15: d(gate), 16-31: d(up) of output row row & 15
            const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(epi + row * 128 + ((ch ^ (row & 7)) << 4));
            const int gm = mrow0 + sr * 16 + (row & 15);
            if (gm < M) st_c2_bf16x8(C + (int64_t)gm * ldc + (row >> 4) * N + n0 + wn * TN + ch * 8, v);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        continue;
      }
      // the wave's TN columns leave in passes of up to 64 (TN = 96: 64 + 32)
#pragma unroll
      for (int p0 = 0; p0 < NBF; p0 += 2) {
        const int nb = NBF - p0 < 2 ? NBF - p0 : 2;  // 32-column blocks of this pass
#pragma unroll
        for (int bq = 0; bq < 2; ++bq) {
          if (bq >= nb) continue;
          const int bh = p0 + bq;
#pragma unroll
          for (int sr = 0; sr < 2; ++sr)  // 16-row half of the 32-row piece
#pragma unroll
            for (int sc = 0; sc < 2; ++sc) {  // 16-column half of the 32-column block: this lane holds columns 4 q .. 4 q + 3 of it
              bf16x4_t o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = f2bf(acc4[2 * mf + sr][bh * 2 + sc][e]);
              const int row = sr * 16 + l15, c = bq * 4 + sc * 2 + (q >> 1);
              *reinterpret_cast<bf16x4_t*>(epi + row * 128 + ((c ^ (row & 7)) << 4) + (q & 1) * 8) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          if (it >= 2 * nb) continue;
          const int c = it * 64 + lane;
          const int row = nb == 2 ? c >> 3 : c >> 2, ch = nb == 2 ? c & 7 : c & 3;
          const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(epi + row * 128 + ((ch ^ (row & 7)) << 4));
          const int gm = mrow0 + row;
          const int gn = GLU ? (ch >> 2) * (N / 2) + n0 / 2 + wn * 32 + (ch & 3) * 8 : n0 + wn * TN + p0 * 32 + ch * 8;
          if (ROPE) {
            if (gm < M && gn < ea.rope_cols) {  // a q / k chunk: four pairs of head dims (gn % 64) / 2 .. + 3 at position gm % T
              const int tab = (gm % ea.T) * 32 + ((gn & 63) >> 1);
              st_c2_bf16x8(C + (int64_t)gm * ldc + gn, rope8(v, *reinterpret_cast<const f32x4_t*>(ea.rcos + tab),
                                                          *reinterpret_cast<const f32x4_t*>(ea.rsin + tab), 1.f));
              continue;
            }
          }
          if (gm < M && gn < N) st_c_bf16x8(C + (int64_t)gm * ldc + gn, v);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      if (GLU) {  // act for the 32 rows x 32 hidden units of this piece, through the same scratch (64-byte rows)
#pragma unroll
        for (int sr = 0; sr < 2; ++sr)
#pragma unroll
          for (int sc = 0; sc < 2; ++sc) {
            const int fi = 2 * mf + sr;
            bf16x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = plm_swiglu_bf16(f2bf(acc4[fi][sc][e]), f2bf(acc4[fi][2 + sc][e]));
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
          if (gm < M) st_c2_bf16x8(act + (int64_t)gm * ldact + n0 / 2 + wn * 32 + ch * 8, v);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    credit = m0 + BM <= M && n0 + BN <= N;  // interior tile: every wave issued exactly NS stores
    credit_i = __builtin_amdgcn_readfirstlane(credit ? 1 : 0);
  }
  wait_vm<0>();  // the re-staged tail loads write this workgroup's LDS: they must have landed before the last wave leaves
}

// =============================================================================================
// TN big tile:  slab[split][i][j] (or C) = sum_{k in split} A[k][i] * B[k][j]   (dW = dY^T X)
// Same persistent 4-phase pipeline; a work ITEM is (split, 256x256 tile) with its own K range, items of
// one split are adjacent so the workgroups of an XCD share the split's A/B panels in L2.
// Half-tile LDS image: [64 k-rows][128 cols] bf16 (256-byte rows, one DMA wave-instruction = 4 rows);
// the 32-byte column pairs of row k are rotated by 2*(k&3) on the DMA source address so that the
// ds_read_b64_tr_b16 operand reads (k = token row) are bank-conflict free (see gemm_tn_dma_kernel).
// =============================================================================================
// Work items: the first n_full tiles (tile rows 0 .. rfull-1) take the whole contraction and write C directly
// (C = / += alpha*acc); the tiles of the remaining rows are split `splits` ways over K and write raw fp32 partials
// into slab[split][row - rfull*256][N], summed by splitk_reduce_kernel.  rfull = 0 is plain split-K (small
// outputs), splits = 1 with no remainder is plain tiling; the hybrid keeps every CU busy for a whole number of rounds.
// Schedule: see gemm_nt_big_kernel - half-tile slots are refilled two K-tiles ahead (phase 1: A1 of kt+1, phases 2/3/4: A0/B0/B1
// of kt+2), five LDS-DMA groups in flight behind every counted wait.
// GROUPED: up to PLM_TN_GROUP_MAX independent problems with the same contraction length (the dW GEMMs of one or more
// transformer blocks: same token rows, different projections) run as ONE launch over the union of their 256x256 output tiles,
// with the same hybrid as above: the first n_full tiles (whole rounds of the persistent grid) take the whole contraction and
// write (C = / += alpha*acc) themselves, the remaining tiles are split over K into dense fp32 blocks
// ws[split * n_rem + r][256][256] that tn_grouped_reduce_kernel sums into C.  The more problems a launch carries, the smaller the
// split remainder: one block (108 tiles) is all remainder (7 pieces per tile), twelve blocks (1296 tiles) leave 16 tiles to split.
#define PLM_TN_GROUP_MAX 48
// (kernel arguments: 48 problems x (operands + outputs) = 3.2 KB of the 4 KB kernarg segment)
struct TnGroup {
  const uint16_t* A[PLM_TN_GROUP_MAX];
  const uint16_t* B[PLM_TN_GROUP_MAX];
  int lda[PLM_TN_GROUP_MAX], ldb[PLM_TN_GROUP_MAX];
  int M[PLM_TN_GROUP_MAX], N[PLM_TN_GROUP_MAX], tiles_n[PLM_TN_GROUP_MAX];
  int tile_base[PLM_TN_GROUP_MAX + 1];  // first global tile of each problem; [count] = number of tiles
  int count;
  int n_full;  // tiles 0 .. n_full-1 take the whole contraction and write C directly (whole rounds of the persistent grid)
  int splits;  // the remaining tiles are cut `splits` ways over K (L K-tiles each): items n_full + split * n_rem + r, split-major so
  int L;       // that the workgroups of an XCD share a split's A / B panels in L2; pieces go to ws[split * n_rem + r][256][256]
};
struct TnGroupOut {
  float* C[PLM_TN_GROUP_MAX];
  const float* alpha[PLM_TN_GROUP_MAX];
  int ldc[PLM_TN_GROUP_MAX];
  int accumulate[PLM_TN_GROUP_MAX];
};

// v_mfma_f32_16x16x32_bf16 (see gemm_nt_big_kernel): a lane group of 16 reads K-chunk q = lane >> 4 of a 16-column block, so the
// four groups of one transpose read touch k-rows 8 apart - the pair rotation takes bit 3 of the k-row as well as its low two bits.
__device__ __forceinline__ int tn_rot(int k) { return 2 * (k & 3) + ((k >> 3) & 1); }

template <bool GROUPED = false>
__global__ __launch_bounds__(512, 2) void gemm_tn_big_kernel(const uint16_t* __restrict__ A, int64_t lda,
                                                             const uint16_t* __restrict__ B, int64_t ldb, float* __restrict__ C,
                                                             int64_t ldc, float* __restrict__ slabs, int M, int N, int K, int kchunk,
                                                             int splits, int rfull, int accumulate,
                                                             const float* __restrict__ alpha_dev, int tiles_m, int tiles_n, TnGroup grp,
                                                             TnGroupOut gout) {
  constexpr int BM = 256, BN = 256, WN = 4;
  constexpr int TM = 128, TN = 64, AH = 64, AF = 2, NA = 2 * AF;
  constexpr int HT = 64 * 256;  // one half-tile: 64 k-rows x 128 cols bf16
  constexpr int STAGE = 4 * HT;
  constexpr int OFF_A0 = 0, OFF_B0 = HT, OFF_B1 = 2 * HT, OFF_A1 = 3 * HT;
  constexpr int W_ALL = 10;  // A_DMA = B_DMA = 2: five DMA groups stay in flight behind every counted wait
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const bool grp1 = (wave >> 2) != 0;  // wave-uniform; waves w and w+4 share a SIMD
#ifndef PLM_NO_PRIO_HALF_TN
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);  // static priority for the second-dispatched half (see gemm_nt_big_kernel)
#endif
  const int t16 = lane & 15;
  const int q = lane >> 4;  // K-chunk of the lane group
  const int n_full = rfull * tiles_n;
  const int n_rem = tiles_m * tiles_n - n_full;
  const int nkt = K / 64;
  const int g_ntl = GROUPED ? grp.tile_base[grp.count] : 0, g_nrem = GROUPED ? g_ntl - grp.n_full : 0;
  const int nitems = GROUPED ? grp.n_full + g_nrem * grp.splits : n_full + n_rem * splits;
  int c_prob = 0, c_tile = 0;  // GROUPED: problem / global tile of the item coords() looked at last

  // split = -1: whole-K tile written to C; split >= 0: partial into slab[split]
  auto coords = [&](int item, int& i0, int& j0, int& kbeg, int& kend, int& split) {
    int tile;
    if (GROUPED) {
      int t, lo, hi2;
      if (item < grp.n_full) {
        split = -1;
        t = item;
        lo = 0;
        hi2 = nkt;
      } else {
        const int j = item - grp.n_full;
        split = j / g_nrem;
        t = grp.n_full + (j - split * g_nrem);
        lo = min(nkt, split * grp.L);
        hi2 = min(nkt, (split + 1) * grp.L);
      }
      kbeg = kend = 0;
      i0 = j0 = 0;
      if (hi2 <= lo) return;  // an empty split (L does not divide the contraction)
      // everything here is wave-uniform; saying so keeps the table reads on the scalar unit.  (A VECTOR load of a table entry is
      // tracked by vmcnt, and hipcc then drains the LDS-DMA ring with s_waitcnt vmcnt(0) at the top of every K-tile: -20 %.)
      t = __builtin_amdgcn_readfirstlane(t);
      int pidx = 0;
#pragma unroll
      for (int q = 1; q < PLM_TN_GROUP_MAX; ++q)
        if (q < grp.count && t >= grp.tile_base[q]) pidx = q;
      pidx = __builtin_amdgcn_readfirstlane(pidx);
      const int local = t - grp.tile_base[pidx];
      i0 = (local / grp.tiles_n[pidx]) * BM;
      j0 = (local % grp.tiles_n[pidx]) * BN;
      kbeg = lo * 64;
      kend = hi2 * 64;
      c_prob = pidx;
      c_tile = t;
      return;
    }
    if (item < n_full) {
      split = -1;
      tile = item;
      kbeg = 0;
      kend = K;
    } else {
      const int j = item - n_full;
      split = j / n_rem;
      tile = n_full + (j - split * n_rem);
      kbeg = min(K, split * kchunk);
      kend = min(K, kbeg + kchunk);
    }
    i0 = (tile / tiles_n) * BM;
    j0 = (tile % tiles_n) * BN;
  };

  // DMA lane map: instruction q = i*8 + wave covers k-rows q*4 .. q*4+3; lane -> (row, physical 16-byte chunk)
  // LDS-DMA addresses = wave-uniform operand base (+ k0 rows, added on the scalar side) + per-lane unsigned byte offsets
  unsigned pa[2][2], pb[2][2];
  const uint16_t* s_ap = A;
  const uint16_t* s_bp = B;
  int64_t s_lda = lda, s_ldb = ldb;  // row strides of the problem being staged
  auto set_ptrs = [&](int item) {
    int i0, j0, kbeg, kend, split;
    coords(item, i0, j0, kbeg, kend, split);
    const uint16_t* Ap = GROUPED ? grp.A[c_prob] : A;
    const uint16_t* Bp = GROUPED ? grp.B[c_prob] : B;
    const int Mp = GROUPED ? grp.M[c_prob] : M, Np = GROUPED ? grp.N[c_prob] : N;
    if (GROUPED) {
      s_lda = (int64_t)grp.lda[c_prob];
      s_ldb = (int64_t)grp.ldb[c_prob];
    }
    s_ap = Ap;
    s_bp = Bp;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (i * 8 + wave) * 4 + (lane >> 4);
      const int pp = (lane & 15) >> 1, half16 = lane & 1;
      const int cb = (pp - tn_rot(row)) & 7;     // logical 32-byte pair held at physical position pp
      const int c = cb * 16 + half16 * 8;        // logical column inside the half-tile (0..127)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ci = i0 + (c / AH) * TM + h * AH + (c % AH);  // A: wave-row group, half, local column
        const int cj = j0 + (c / 32) * TN + h * 32 + (c % 32);  // B: wave-col group, half, local column
        pa[h][i] = (unsigned)(((int64_t)row * s_lda + min(ci, Mp - 8)) * 2);
        pb[h][i] = (unsigned)(((int64_t)row * s_ldb + min(cj, Np - 8)) * 2);
      }
    }
  };
  const unsigned smem_u = __builtin_amdgcn_readfirstlane(lds_addr_u32(smem));  // LDS byte address of the stage ring
  auto issue = [&](const uint16_t* base, const unsigned (&off)[2], int64_t ld, unsigned dst, int k0) {
    const uint16_t* kb = base + (int64_t)k0 * ld;  // wave-uniform
#pragma unroll
    for (int i = 0; i < 2; ++i) dma16_saddr_u32(kb, off[i], dst + (i * 8 + wave) * 1024);
  };
  // operand fragment: 16 logical columns starting at c0 of a half-tile, K-step of 32 `ks` (lane group q holds k = ks*32 + q*8 .. +7),
  // by two transpose reads
  auto tr_frag16 = [&](const char* ht, int c0, int ks) -> bf16x8_t {
    const int row = ks * 32 + q * 8 + (t16 >> 2);
    const char* p = ht + row * 256 + (((c0 >> 4) + tn_rot(row)) & 7) * 32 + (t16 & 3) * 8;
    return join_tr(lds_read_tr16(p), lds_read_tr16(p + 4 * 256));
  };

  const int first = xcd_remap(blockIdx.x, gridDim.x);
  if (first >= nitems) return;
  float alpha = alpha_dev ? *alpha_dev : 1.f;
  asm volatile("; alpha pinned" : "+v"(alpha));

  // staging cursor: (s_item, s_k) = next K-tile to stage; s_kend = end of that item's K range
  int s_item = first, s_k = 0, s_kend = 0;
  auto open_item = [&]() {   // position the cursor on the first K-tile of s_item, skipping empty items
    while (s_item < nitems) {
      int i0, j0, kbeg, kend, split;
      coords(s_item, i0, j0, kbeg, kend, split);
      if (kend > kbeg) {
        s_k = kbeg;
        s_kend = kend;
        set_ptrs(s_item);
        return;
      }
      s_item += gridDim.x;
    }
  };
  int s_st = 0;  // stage buffer of the K-tile under the staging cursor
  auto advance_staged = [&]() {
    s_k += 64;
    s_st ^= 1;
    if (s_k >= s_kend) {
      s_item += gridDim.x;
      open_item();
    }
  };
  open_item();
  if (s_item < nitems) {
    issue(s_ap, pa[0], s_lda, smem_u + OFF_A0, s_k);
    issue(s_bp, pb[0], s_ldb, smem_u + OFF_B0, s_k);
    issue(s_bp, pb[1], s_ldb, smem_u + OFF_B1, s_k);
    issue(s_ap, pa[1], s_lda, smem_u + OFF_A1, s_k);
    advance_staged();
  }
  if (s_item < nitems) {  // plus A0, B0, B1 of the second K-tile
    issue(s_ap, pa[0], s_lda, smem_u + STAGE + OFF_A0, s_k);
    issue(s_bp, pb[0], s_ldb, smem_u + STAGE + OFF_B0, s_k);
    issue(s_bp, pb[1], s_ldb, smem_u + STAGE + OFF_B1, s_k);
    wait_vm<W_ALL>();
  } else {
    wait_vm<0>();
  }
  phase_barrier();

  int st = 0;
  for (int item = first; item < nitems; item += gridDim.x) {
    int nk;
    {
      int i0_, j0_, kbeg_, kend_, split_;
      coords(item, i0_, j0_, kbeg_, kend_, split_);
      if (GROUPED && kend_ <= kbeg_) continue;  // empty split
      nk = (kend_ - kbeg_) / 64;
    }
    f32x4_t acc4[4 * AF][4];  // [16-row block of the wave's 128 rows][16-column block of its 64 columns]
#pragma unroll
    for (int i = 0; i < 2 * NA; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc4[i][j][r] = 0.f;

    for (int kt = 0; kt < nk; ++kt) {
      bool more = s_item < nitems;  // workgroup-uniform: the staging cursor still points at a K-tile
      unsigned sst = smem_u + s_st * STAGE;
      const char* cur = smem + st * STAGE;
      bf16x8_t a6[2 * AF][2], b06[2][2], b16[2][2];  // fragments: [16-column block][K-step of 32]

      if (more) {
        issue(s_ap, pa[1], s_lda, sst + OFF_A1, s_k);
        advance_staged();
        more = s_item < nitems;
        sst = smem_u + s_st * STAGE;
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int c = 0; c < 2; ++c) b06[c][ks] = tr_frag16(cur + OFF_B0, wn * 32 + c * 16, ks);
#pragma unroll
        for (int f = 0; f < NA; ++f) a6[f][ks] = tr_frag16(cur + OFF_A0, wm * AH + f * 16, ks);
      }
      auto sync = [&](auto wtag) {
        if (more) wait_vm<decltype(wtag)::value>(); else wait_vm<0>();
        phase_barrier();
      };
      using WAll = std::integral_constant<int, W_ALL>;
      if (grp1) sync(WAll{});  // the second wave group's barrier sits between the reads and the MFMAs (see gemm_nt_big_kernel)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < NA; ++f)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc4[f][c] = mfma16(a6[f][ks], b06[c][ks], acc4[f][c]);
      if (!grp1) sync(WAll{});

      if (more) {
        issue(s_ap, pa[0], s_lda, sst + OFF_A0, s_k);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int c = 0; c < 2; ++c) b16[c][ks] = tr_frag16(cur + OFF_B1, wn * 32 + c * 16, ks);
      if (grp1) sync(WAll{});
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < NA; ++f)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc4[f][2 + c] = mfma16(a6[f][ks], b16[c][ks], acc4[f][2 + c]);
      if (!grp1) sync(WAll{});

      if (more) {
        issue(s_bp, pb[0], s_ldb, sst + OFF_B0, s_k);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < NA; ++f) a6[f][ks] = tr_frag16(cur + OFF_A1, wm * AH + f * 16, ks);
      if (grp1) sync(std::integral_constant<int, W_ALL - 2>{});  // the fourth DMA group of this K-tile is only issued in phase 4
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < NA; ++f)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc4[NA + f][2 + c] = mfma16(a6[f][ks], b16[c][ks], acc4[NA + f][2 + c]);

      if (more) {
        issue(s_bp, pb[1], s_ldb, sst + OFF_B1, s_k);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < NA; ++f)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc4[NA + f][c] = mfma16(a6[f][ks], b06[c][ks], acc4[NA + f][c]);
      if (!grp1) {
        if (more) wait_vm<W_ALL>(); else wait_vm<0>();
        phase_barrier();
      }
      st ^= 1;
    }

    // epilogue: D block (f, j) of 16x16: lane owns column l15 = lane & 15 and rows 4 q .. 4 q + 3; 64-byte row segments per lane group.
    // The item's coordinates are looked up again here instead of being carried through the K loop: every scalar that lives across
    // the loop costs the staging pointers their registers (SGPR spills in the loop: -20 % on the grouped launch).
    int i0, j0, kbeg, kend, split;
    coords(item, i0, j0, kbeg, kend, split);
    const int e_tile = c_tile, e_prob = c_prob;
    if (GROUPED && split >= 0) {  // raw accumulators into this piece's dense 256x256 block (edge rows / columns hold clamped duplicates: ignored)
      float* blk = slabs + ((int64_t)split * g_nrem + (e_tile - grp.n_full)) * (BM * BN);
      // one lane offset + a wave-uniform base per store (128 precomputed per-lane addresses spilled to scratch)
      const unsigned lane_off = (unsigned)(((wm * TM + 4 * q) * BN + wn * TN + (lane & 15)) * 4);
#pragma unroll
      for (int f = 0; f < 4 * AF; ++f) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int r = 0; r < 4; ++r) st_f32_saddr(blk + (f * 16 + r) * BN + j * 16, lane_off, acc4[f][j][r]);
        }
      }
      continue;
    }
    const bool direct = split < 0;  // workgroup-uniform
    const int mrem = M - rfull * BM;
    // GROUPED reaches this point with whole-K tiles only: the problem's own output, alpha and accumulate flag
    int Mo = M, No = N;
    float* out;
    int64_t ld;
    float scl;
    bool rmw;
    if constexpr (GROUPED) {
      Mo = grp.M[e_prob];
      No = grp.N[e_prob];
      out = gout.C[e_prob];
      ld = (int64_t)gout.ldc[e_prob];
      // alpha through the scalar unit: a vector load here is hoisted above the slab / direct branch by hipcc, stays "pending" on
      // the slab path and costs a draining s_waitcnt vmcnt(0) at the top of every K-tile of the next item (-20 %)
      const float* ap = gout.alpha[e_prob];
      scl = 1.f;
      if (ap != nullptr) {
        unsigned bits;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(bits) : "s"(ap) : "memory");
        scl = __builtin_bit_cast(float, bits);
      }
      rmw = gout.accumulate[e_prob] != 0;
    } else {
      out = direct ? C : slabs + ((int64_t)split * mrem - (int64_t)rfull * BM) * N;
      ld = direct ? ldc : N;
      scl = direct ? alpha : 1.f;
      rmw = direct && accumulate;
    }
#pragma unroll
    for (int f = 0; f < 4 * AF; ++f) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = j0 + wn * TN + j * 16 + (lane & 15);
        if (col >= No) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = i0 + wm * TM + f * 16 + 4 * q + r;
          if (row >= Mo) continue;
          float* dst = out + (int64_t)row * ld + col;
          const float v = acc4[f][j][r] * scl;
          *dst = rmw ? *dst + v : v;  // (`nt` on these stores: -0.1 % in the step, profiles/r04_ab_stores.txt)
        }
      }
    }
  }
}

// Plan for the big TN kernel: returns false when it should not be used.  rfull = tile rows done without split.
bool plm_tn_big_plan(int64_t M, int64_t N, int64_t K, int* splits, int* rfull) {
  if (g_num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  if (K % 64 != 0 || M < 256 || N < 256) return false;
  const int64_t R = plm_cdiv(M, 256), Cn = plm_cdiv(N, 256), tiles = R * Cn;
  const int64_t max_by_k = K / 512 > 0 ? K / 512 : 1;  // >= 8 K-tiles per item
  const int64_t slots = persistent_slots();
  int64_t rf = 0, s = 1;
  if (tiles < slots) {
    s = slots / tiles;
  } else {
    rf = ((tiles / slots) * slots) / Cn;
    const int64_t rem = (R - rf) * Cn;
    s = rem > 0 ? slots / rem : 1;
  }
  if (s > max_by_k) s = max_by_k;
  if (s < 1) s = 1;
  *splits = (int)s;
  *rfull = (int)rf;
  return true;
}

void plm_launch_gemm_tn_big(int splits, int rfull, int accumulate, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb,
                            float* C, int64_t ldc, float* slabs, int64_t M, int64_t N, int64_t K, const float* alpha_dev,
                            hipStream_t s) {
  const int tm = (int)plm_cdiv(M, 256), tn = (int)plm_cdiv(N, 256);
  const int kchunk = (int)(plm_cdiv(plm_cdiv(K, splits), 64) * 64);
  const int nitems = rfull * tn + (tm - rfull) * tn * splits;
  const int slots = persistent_slots();
  const dim3 grid(nitems < slots ? nitems : slots), block(512);
  // deep-prefetch ring + offset wave groups (see gemm_nt_big_kernel): each alone measured equal to the plain ring in the step, together
  // +0.8 % end to end (round 1, run 34)
  hipLaunchKernelGGL((gemm_tn_big_kernel<false>), grid, block, 0, s, A, lda, B, ldb, C, ldc, slabs, (int)M, (int)N, (int)K, kchunk, splits,
                     rfull, accumulate, alpha_dev, tm, tn, TnGroup{}, TnGroupOut{});
}

// ---- grouped TN (dW of one transformer block in one stream-K launch) ----------------------------------------------

// C_p[tile rows/cols] (+)= alpha_p * sum of the tile's pieces, for the split tiles n_full .. ntiles-1; one workgroup per
// (tile, 16-row group)
__global__ __launch_bounds__(256) void tn_grouped_reduce_kernel(const float* __restrict__ ws, TnGroup grp, TnGroupOut out) {
  const int r_ = blockIdx.x >> 4, rg = blockIdx.x & 15;
  const int t = grp.n_full + r_;
  int pidx = 0;
#pragma unroll
  for (int q = 1; q < PLM_TN_GROUP_MAX; ++q)
    if (q < grp.count && t >= grp.tile_base[q]) pidx = q;
  const int local = t - grp.tile_base[pidx];
  const int i0 = (local / grp.tiles_n[pidx]) * 256, j0 = (local % grp.tiles_n[pidx]) * 256;
  const int nrem = grp.tile_base[grp.count] - grp.n_full;
  const float a = out.alpha[pidx] ? *out.alpha[pidx] : 1.f;
  const int M = grp.M[pidx], N = grp.N[pidx];
  float* C = out.C[pidx];
  const int64_t ldc = out.ldc[pidx];
  const bool acc = out.accumulate[pidx] != 0;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int e = it * 256 + threadIdx.x;      // 16 rows x 64 float4
    const int r = rg * 16 + (e >> 6), c = (e & 63) * 4;
    const int row = i0 + r, col = j0 + c;
    if (row >= M || col >= N) continue;        // N % 8 == 0: the four columns are in range together
    const float* src = ws + (int64_t)r_ * 65536 + r * 256 + c;
    f32x4_t v = *reinterpret_cast<const f32x4_t*>(src);
    for (int k = 1; k < grp.splits; ++k) v += *reinterpret_cast<const f32x4_t*>(src + (int64_t)k * nrem * 65536);
    v *= a;
    float* dst = C + (int64_t)row * ldc + col;
    if (acc) v += *reinterpret_cast<const f32x4_t*>(dst);
    *reinterpret_cast<f32x4_t*>(dst) = v;
  }
}

static bool tn_group_plan(const int64_t* Ms, const int64_t* Ns, int count, int64_t K, TnGroup* g, int* nslabs) {
  if (g_num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  if (count < 1 || count > PLM_TN_GROUP_MAX || K % 64 != 0 || K < 64) return false;
  int base = 0;
  for (int p = 0; p < count; ++p) {
    if (Ms[p] < 8 || Ns[p] < 8 || Ms[p] % 8 != 0 || Ns[p] % 8 != 0) return false;
    g->M[p] = (int)Ms[p];
    g->N[p] = (int)Ns[p];
    g->tiles_n[p] = (int)plm_cdiv(Ns[p], 256);
    g->tile_base[p] = base;
    base += (int)(plm_cdiv(Ms[p], 256) * plm_cdiv(Ns[p], 256));
  }
  g->tile_base[count] = base;
  g->count = count;
  const int64_t nkt = K / 64;
  const int slots = persistent_slots();
  // whole-K tiles for the full rounds; the remainder is split over K with the count that fills its rounds best (>= 8 K-tiles per
  // piece, mild bias against slab traffic).  Fewer than `slots` tiles: everything is remainder (plain split-K).
  const int nfull = (base / slots) * slots, nrem = base - nfull;
  int best = 1;
  if (nrem > 0) {
    double best_cost = 1e30;
    for (int sp = 1; sp <= 32 && (sp == 1 || nkt / sp >= 8); ++sp) {
      const int64_t L = plm_cdiv(nkt, sp);
      const int64_t rounds = plm_cdiv((int64_t)sp * nrem, slots);
      const double cost = (double)(rounds * L) * (1.0 + 0.01 * (sp - 1));  // K-tiles of wall time for the remainder
      if (cost < best_cost - 1e-9) {
        best_cost = cost;
        best = sp;
      }
    }
  }
  g->n_full = nfull;
  g->splits = nrem > 0 ? best : 0;
  g->L = (int)plm_cdiv(nkt, best);
  *nslabs = nrem > 0 ? best * nrem : 0;  // dense 256x256 fp32 blocks
  return (int64_t)base * nkt < (1ll << 30);
}

extern "C" size_t plm_gemm_tn_grouped_workspace_bytes(const int64_t* Ms, const int64_t* Ns, int count, int64_t K) {
  TnGroup g{};
  int ns = 0;
  if (!Ms || !Ns || !tn_group_plan(Ms, Ns, count, K, &g, &ns)) return 0;
  return (size_t)ns * 65536 * sizeof(float) + 16;  // never zero: 0 means "unsupported shapes"
}

extern "C" int plm_gemm_bf16_tn_grouped(const plm_tn_problem* probs, int count, int64_t K, void* workspace, size_t workspace_bytes,
                                        void* stream) {
  PLM_REQUIRE(probs && workspace, "plm_gemm_bf16_tn_grouped: null pointer");
  PLM_REQUIRE(count >= 1 && count <= PLM_TN_GROUP_MAX, "plm_gemm_bf16_tn_grouped: count=%d must be 1..%d", count, PLM_TN_GROUP_MAX);
  int64_t Ms[PLM_TN_GROUP_MAX], Ns[PLM_TN_GROUP_MAX];
  TnGroup g{};
  TnGroupOut o{};
  for (int p = 0; p < count; ++p) {
    const plm_tn_problem& q = probs[p];
    PLM_REQUIRE(q.A && q.B && q.C, "plm_gemm_bf16_tn_grouped: null pointer in problem %d", p);
    PLM_REQUIRE(q.M % 8 == 0 && q.N % 8 == 0 && q.lda % 8 == 0 && q.ldb % 8 == 0 && q.ldc % 4 == 0 && q.lda >= q.M && q.ldb >= q.N && q.ldc >= q.N,
                "plm_gemm_bf16_tn_grouped: problem %d: M, N, lda, ldb must be multiples of 8, ldc of 4", p);
    PLM_REQUIRE(q.lda < (1ll << 31) && q.ldb < (1ll << 31) && q.ldc < (1ll << 31), "plm_gemm_bf16_tn_grouped: problem %d: row strides must fit 31 bits", p);
    PLM_REQUIRE(((reinterpret_cast<uintptr_t>(q.A) | reinterpret_cast<uintptr_t>(q.B) | reinterpret_cast<uintptr_t>(q.C)) & 15) == 0,
                "plm_gemm_bf16_tn_grouped: problem %d: base pointers must be 16-byte aligned", p);
    Ms[p] = q.M;
    Ns[p] = q.N;
    g.A[p] = q.A;
    g.B[p] = q.B;
    g.lda[p] = (int)q.lda;
    g.ldb[p] = (int)q.ldb;
    o.C[p] = q.C;
    o.ldc[p] = (int)q.ldc;
    o.alpha[p] = q.alpha_dev;
    o.accumulate[p] = q.accumulate;
  }
  int ns = 0;
  PLM_REQUIRE(tn_group_plan(Ms, Ns, count, K, &g, &ns), "plm_gemm_bf16_tn_grouped: unsupported shapes (K %% 64 == 0, M, N multiples of 8)");
  const size_t need = (size_t)ns * 65536 * sizeof(float);
  if (workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15) != 0) {
    plm_set_error("plm_gemm_bf16_tn_grouped: workspace of %zu bytes (16-byte aligned) required, %zu given", need, workspace_bytes);
    return PLM_E_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int nrem = g.tile_base[count] - g.n_full;
  const int64_t nitems = g.n_full + (int64_t)nrem * g.splits;
  const int slots = persistent_slots();
  hipLaunchKernelGGL((gemm_tn_big_kernel<true>), dim3((unsigned)(nitems < slots ? nitems : slots)), dim3(512), 0, s, nullptr, 0, nullptr, 0,
                     nullptr, 0, (float*)workspace, 0, 0, (int)K, 0, 1, 0, 0, nullptr, 0, 0, g, o);
  if (nrem > 0)
    hipLaunchKernelGGL(tn_grouped_reduce_kernel, dim3((unsigned)(nrem * 16)), dim3(256), 0, s, (const float*)workspace, g, o);
  PLM_CHECK_LAUNCH("plm_gemm_bf16_tn_grouped");
  return PLM_OK;
}

// C[row0 + r][c] = bf16(alpha * sum of the pieces of (r, c)'s tile) for the stream-K rows of a hybrid NT GEMM.
// The pieces of remainder tile t come from chunks floor(t*nkt/L) .. floor(((t+1)*nkt-1)/L), slab index = chunk - first.
__global__ __launch_bounds__(256) void nt_streamk_reduce_kernel(const float* __restrict__ ws, uint16_t* __restrict__ C, int64_t ldc, int rows,
                                                                int N, int tiles_n, int nkt, int L, int nchunks,
                                                                const float* __restrict__ alpha_dev) {
  const float alpha = alpha_dev ? *alpha_dev : 1.f;
  const int nq = N >> 3;
  const int64_t nv = (int64_t)rows * nq;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    const int64_t row = i / nq;
    const int col = (int)(i - row * nq) * 8;
    const int t = (int)(row >> 8) * tiles_n + (col >> 8);
    const int cf = (t * nkt) / L;
    const int cl = min(((t + 1) * nkt - 1) / L, nchunks - 1);
    f32x4_t a = *reinterpret_cast<const f32x4_t*>(ws + row * N + col);
    f32x4_t b = *reinterpret_cast<const f32x4_t*>(ws + row * N + col + 4);
    for (int k = 1; k <= cl - cf; ++k) {
      a += *reinterpret_cast<const f32x4_t*>(ws + ((int64_t)k * rows + row) * N + col);
      b += *reinterpret_cast<const f32x4_t*>(ws + ((int64_t)k * rows + row) * N + col + 4);
    }
    u32x4_t o = {pack_bf2(a[0] * alpha, a[1] * alpha), pack_bf2(a[2] * alpha, a[3] * alpha), pack_bf2(b[0] * alpha, b[1] * alpha),
                 pack_bf2(b[2] * alpha, b[3] * alpha)};
    *reinterpret_cast<u32x4_t*>(C + row * ldc + col) = o;
  }
}

// Tile shapes of the persistent NT kernel and the automatic policy that picks one per launch: round efficiency of the tile count on the
// persistent grid x the useful fraction of the (ragged) edge tiles x a measured per-tile rate relative to 256x256.  256x192 has 22 % fewer
// LDS-DMA bytes and 17 % fewer LDS reads per MFMA than 256x128, and N = 768 is four 192-column tiles = exactly two rounds at M = 32768;
// 256x256 at 0.90 round efficiency beats 256x128 at 1.0 on the qkv shape.  128x192 (round 5; 8 waves of 32x96) is the shape of the
// short batches: M = 8192 (the reference's document-mask config, config_doc_mask.yaml:35) makes N = 768 exactly ONE round of 256 tiles
// and N = 2304 exactly three, where every 256-row tile leaves 25-62 % of the chip idle.
// Rates fitted on gpurun_out/r05b kbench --variants tables at M = 8192 / 16384 / 32768 (profiles/r05_kbench_variants.txt).
// Order = preference on ties (the first strictly greater wins).
struct NtTileShape {
  int bm, bn;
  double rate;
};
static const NtTileShape kNtTiles[4] = {{256, 256, 1.0}, {256, 128, 0.88}, {256, 192, 0.94}, {128, 192, 0.72}};
static const int kNtTileVariant[4] = {4, 6, 5, 7};  // the explicit variant number of each (plm_gemm_bf16_nt_ex)
static double nt_tile_eff(int i, int64_t M, int64_t N, int slots) {
  const NtTileShape& t = kNtTiles[i];
  const int64_t tm = plm_cdiv(M, t.bm), tn = plm_cdiv(N, t.bn);
  return round_efficiency(tm * tn, slots) * ((double)N / (double)(tn * t.bn)) * ((double)M / (double)(tm * t.bm)) * t.rate;
}
// index into kNtTiles of the best shape, its efficiency in *eff
static int nt_pick_tile(int64_t M, int64_t N, int slots, double* eff) {
  int best = 0;
  double be = -1.0;
  for (int i = 0; i < 4; ++i) {
    const double e = nt_tile_eff(i, M, N, slots);
    if (e > be) {
      be = e;
      best = i;
    }
  }
  *eff = be;
  return best;
}
// the hardware-scheduled 128x128 LDS-DMA kernel of gemm.hip (two 4-wave workgroups per CU, ~0.8 of the persistent 256x256 per-tile rate)
// on the same scale: what the persistent kernels have to beat
static double nt_dma128_eff(int64_t M, int64_t N, int slots) {
  const int64_t tm = plm_cdiv(M, 128), tn = plm_cdiv(N, 128);
  return round_efficiency(tm * tn, 2 * slots) * ((double)N / (double)(tn * 128)) * ((double)M / (double)(tm * 128)) * 0.80;
}

// Hybrid plan for the 256x256 NT kernel: whole-K tiles for the full rounds, stream-K over the remaining tile rows.
// Returns false when the plain schedules are at least as good (or the shape does not qualify).
struct NtHybridPlan {
  int rfull, nchunks, L, nslabs;
};
bool plm_nt_hybrid_plan(int64_t M, int64_t N, int64_t K, NtHybridPlan* p) {
  if (g_num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  if (plm_env().nt_no_hybrid) return false;
  const bool mk = plm_env().nt_hybrid_min_k >= 0;  // tests / A-B runs lower the thresholds
  // measured (profiles/r01_kbench_run18*): the fp32 slab traffic (~40 us) only pays off for long K.  Round 3 lowered the threshold to 2048
  // while CUs are reserved for RCCL because the plain alternative was then the 128x128 kernel (the 0.85 cliff); with the shared tile policy
  // the alternative is the persistent 256x256 kernel on two ragged rounds, which beats the hybrid at K = 2048 ... 4096 under an 8- and a
  // 16-CU reserve (profiles/r05_kbench_variants.txt: fc2 fwd 97 vs 115 us, dX qkv 108 vs 134, dX fc1 187 vs 197) - only lm_head's dX
  // (K = 50304) still gains (2004 vs 2089 us under 16 reserved CUs)
  const int64_t min_k = mk ? plm_env().nt_hybrid_min_k : 8192, min_l = mk ? 2 : 8;
  if (K % 64 != 0 || N % 8 != 0 || M < 2048 || N < 256 || K < min_k) return false;
  const int slots = persistent_slots();
  const int64_t R = plm_cdiv(M, 256), Cn = plm_cdiv(N, 256), tiles = R * Cn, nkt = K / 64;
  if (tiles <= slots) return false;                          // single partial round: nothing to balance
  if (round_efficiency(tiles, slots) >= 0.9) return false;  // plain 256x256 is already well packed
  // ... or a narrower plain tile is (lm_head dX on the whole chip: 4 x 192 columns = exactly two rounds; in the step that beats the
  // hybrid's slab traffic by 0.5 % end to end, round 2).  The same per-tile rates as the automatic policy below.
  double e_plain;
  nt_pick_tile(M, N, slots, &e_plain);
  if (!mk && e_plain >= 0.9) return false;
  const int64_t rf = ((tiles / slots) * slots) / Cn;        // whole tile rows inside the full rounds
  const int64_t rem = (R - rf) * Cn;
  if (rem <= 0 || rf <= 0) return false;
  const int64_t total = rem * nkt;
  const int64_t L = plm_cdiv(total, slots);
  if (L < min_l || L * 10 > nkt * 9) return false;  // too short to amortise a prologue / no round saved
  p->rfull = (int)rf;
  p->L = (int)L;
  p->nchunks = (int)plm_cdiv(total, L);
  p->nslabs = (int)((nkt - 1) / L + 2);
  return true;
}

extern "C" size_t plm_gemm_nt_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  NtHybridPlan p;
  if (M <= 0 || N <= 0 || K <= 0 || !plm_nt_hybrid_plan(M, N, K, &p)) return 0;
  return (size_t)p.nslabs * (size_t)(M - (int64_t)p.rfull * 256) * (size_t)N * sizeof(float);
}

// 16-byte alignment of every pointer a fused launch touches with 16-byte vector accesses (LDS-DMA sources, row stores, the saved fc1 output,
// the RoPE tables): a caller of the C ABI with a misaligned view gets the two-launch fallback (whose GEMM checks its own operands), not a
// misaligned global_load_lds_dwordx4
static bool aligned16(std::initializer_list<const void*> ptrs) {
  uintptr_t v = 0;
  for (const void* p : ptrs) v |= reinterpret_cast<uintptr_t>(p);
  return (v & 15) == 0;
}

// fc1 + SwiGLU in one launch (see GLU above).  Returns false when the shape does not qualify (the caller then runs the GEMM and
// plm_swiglu_fwd separately - same bits).
bool plm_launch_gemm_nt_glu(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, uint16_t* C, int64_t ldc, uint16_t* act,
                            int64_t ldact, int64_t M, int64_t N, int64_t K, hipStream_t s) {
  if (g_num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  if (K % 64 != 0 || N % 256 != 0 || M < 512 || lda % 8 != 0 || ldb % 8 != 0 || ldc % 8 != 0 || ldact % 8 != 0) return false;
  if (!aligned16({A, B, C, act})) return false;
  const int tm = (int)plm_cdiv(M, 256), tn = (int)(N / 256);
  const int slots = persistent_slots();
  const int nt_ = tm * tn;
  const HybridArgs hyb{tm, 0, 1, nullptr};
  hipLaunchKernelGGL((gemm_nt_big_kernel<256, 256, 2, 4, false, true>), dim3(nt_ < slots ? nt_ : slots), dim3(512), 0, s, A, lda, B, ldb, C, ldc,
                     (int)M, (int)N, (int)K, nullptr, tm, tn, hyb, EpiArgs{act, ldact, nullptr, nullptr, 0, 0});
  return true;
}

// dX of fc2 + SwiGLU backward in one launch (see GLUB above): DU[M, 2h] from dY[M, K] , W2^T[h, K] and the saved fc1 output U[M, 2h].
bool plm_launch_gemm_nt_glub(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint16_t* U, int64_t ldu, uint16_t* DU,
                             int64_t lddu, int64_t M, int64_t h, int64_t K, hipStream_t s) {
  if (g_num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  if (K % 64 != 0 || h % 256 != 0 || M < 512 || lda % 8 != 0 || ldb % 8 != 0 || ldu % 4 != 0 || lddu % 8 != 0) return false;
  if (!aligned16({A, B, DU}) || (reinterpret_cast<uintptr_t>(U) & 7) != 0) return false;  // U is read in 8-byte pieces
  const int tm = (int)plm_cdiv(M, 256), tn = (int)(h / 256);
  const int slots = persistent_slots();
  const int nt_ = tm * tn;
  const HybridArgs hyb{tm, 0, 1, nullptr};
  hipLaunchKernelGGL((gemm_nt_big_kernel<256, 256, 2, 4, false, false, true>), dim3(nt_ < slots ? nt_ : slots), dim3(512), 0, s, A, lda, B, ldb, DU,
                     lddu, (int)M, (int)h, (int)K, nullptr, tm, tn, hyb, EpiArgs{const_cast<uint16_t*>(U), ldu, nullptr, nullptr, 0, 0});
  return true;
}

// w_qkv projection with RoPE in the epilogue (see ROPE above): the automatic tile policy of plm_launch_gemm_nt_big, no hybrid.
bool plm_launch_gemm_nt_rope(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, uint16_t* C, int64_t ldc, int64_t M, int64_t N,
                             int64_t K, const float* rcos, const float* rsin, int64_t T, int64_t rope_cols, hipStream_t s) {
  if (g_num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  if (K % 64 != 0 || N % 8 != 0 || M < 512 || N < 128 || lda % 8 != 0 || ldb % 8 != 0 || ldc % 8 != 0) return false;
  if (!aligned16({A, B, C, rcos, rsin})) return false;
  const int slots = persistent_slots();
  double eff;
  const int which = nt_pick_tile(M, N, slots, &eff);
  // badly quantised on every tile shape (odd CU reserves): the caller takes the plain GEMM (which may prefer the 128x128 kernel) + the rope pass
  if (eff < nt_dma128_eff(M, N, slots)) return false;
  const int tm = (int)plm_cdiv(M, kNtTiles[which].bm), tn_ = (int)plm_cdiv(N, kNtTiles[which].bn);
  const int nt_ = tm * tn_;
  const dim3 g(nt_ < slots ? nt_ : slots), block(512);
  const HybridArgs hyb{tm, 0, 1, nullptr};
  const EpiArgs ea{nullptr, 0, rcos, rsin, (int)T, (int)rope_cols};
  if (which == 0)
    hipLaunchKernelGGL((gemm_nt_big_kernel<256, 256, 2, 4, false, false, false, true>), g, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K,
                       nullptr, tm, tn_, hyb, ea);
  else if (which == 1)
    hipLaunchKernelGGL((gemm_nt_big_kernel<256, 128, 4, 2, false, false, false, true>), g, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K,
                       nullptr, tm, tn_, hyb, ea);
  else if (which == 2)
    hipLaunchKernelGGL((gemm_nt_big_kernel<256, 192, 4, 2, false, false, false, true>), g, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K,
                       nullptr, tm, tn_, hyb, ea);
  else
    hipLaunchKernelGGL((gemm_nt_big_kernel<128, 192, 4, 2, false, false, false, true>), g, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K,
                       nullptr, tm, tn_, hyb, ea);
  return true;
}

// Host-side launcher used by plm_gemm_bf16_nt (gemm.hip). Returns false when no big-tile variant fits.

// variant: 0 automatic | 4 / 5 / 6 / 7 the persistent kernel on 256x256 / 256x192 / 256x128 / 128x192 tiles (what the automatic policy picks
// from; explicit numbers exist for the tests and tools/kbench.py) | 3 = 4.
// Returns false when the shape is better served by the 128x128 kernels of gemm.hip.
bool plm_launch_gemm_nt_big(int variant, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, uint16_t* C, int64_t ldc,
                            int64_t M, int64_t N, int64_t K, const float* alpha_dev, void* workspace, size_t workspace_bytes,
                            hipStream_t s) {
  if (!ensure_num_cus()) return false;
  const int slots = persistent_slots();
  const dim3 block(512);
  // long K with a badly quantised tile count (lm_head dX: 384 tiles on 256 CUs): whole-K tiles for the full rounds + stream-K
  // over the remaining tile rows, on the staggered schedule (needs the caller's fp32 workspace)
  if (variant == 0 && workspace) {
    NtHybridPlan p;
    if (plm_nt_hybrid_plan(M, N, K, &p)) {
      const int tm = (int)plm_cdiv(M, 256), tn256 = (int)plm_cdiv(N, 256);
      const int64_t rem_rows = M - (int64_t)p.rfull * 256;
      const size_t need = (size_t)p.nslabs * (size_t)rem_rows * (size_t)N * sizeof(float);
      if (workspace_bytes >= need) {
        const HybridArgs h{p.rfull, p.nchunks, p.L, (float*)workspace};
        const int nitems = p.rfull * tn256 + p.nchunks;
        const dim3 g2(nitems < slots ? nitems : slots);
        hipLaunchKernelGGL((gemm_nt_big_kernel<256, 256, 2, 4, true>), g2, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, alpha_dev,
                           tm, tn256, h, EpiArgs{});
        const int64_t nv = rem_rows * (N / 8);
        int64_t rb = plm_cdiv(nv, 256);
        if (rb > 4096) rb = 4096;
        hipLaunchKernelGGL(nt_streamk_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, s, (const float*)workspace,
                           C + (int64_t)p.rfull * 256 * ldc, ldc, (int)rem_rows, (int)N, tn256, (int)(K / 64), p.L, p.nchunks, alpha_dev);
        return true;
      }
    }
  }
  // one schedule (deep-prefetch 4-phase ring with offset wave groups), four tile shapes; variant 3 is kept as an alias of 4
  int which = 0;
  if (variant == 0) {
    double eff;
    which = nt_pick_tile(M, N, slots, &eff);
    // when every persistent shape quantises badly (e.g. odd slot counts while CUs are reserved for RCCL) the hardware-scheduled 128x128
    // LDS-DMA kernel is the better choice
    if (M < 512 || N < 128 || eff < nt_dma128_eff(M, N, slots)) return false;
  } else {
    for (int i = 0; i < 4; ++i)
      if (kNtTileVariant[i] == (variant == 3 ? 4 : variant)) which = i;
  }
  const int tm = (int)plm_cdiv(M, kNtTiles[which].bm), tn_ = (int)plm_cdiv(N, kNtTiles[which].bn);
  const int nt_ = tm * tn_;
  const dim3 g(nt_ < slots ? nt_ : slots);
  const HybridArgs hyb{tm, 0, 1, nullptr};  // plain schedule
  if (which == 0)
    hipLaunchKernelGGL((gemm_nt_big_kernel<256, 256, 2, 4>), g, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, alpha_dev, tm, tn_, hyb, EpiArgs{});
  else if (which == 1)
    hipLaunchKernelGGL((gemm_nt_big_kernel<256, 128, 4, 2>), g, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, alpha_dev, tm, tn_, hyb, EpiArgs{});
  else if (which == 2)
    hipLaunchKernelGGL((gemm_nt_big_kernel<256, 192, 4, 2>), g, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, alpha_dev, tm, tn_, hyb, EpiArgs{});
  else
    hipLaunchKernelGGL((gemm_nt_big_kernel<128, 192, 4, 2>), g, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, alpha_dev, tm, tn_, hyb, EpiArgs{});
  return true;
}
