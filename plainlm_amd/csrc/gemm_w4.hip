// Compiled WITHOUT -amdgpu-mfma-vgpr-form: the 256 accumulator registers of a wave live in the AGPR half of the
// unified register file, the MFMA A/B operands and everything else in the VGPR half (with the VGPR-form flag hipcc
// shuffles accumulators between the two halves inside the loop).
#include "plm_device.h"

#define BIG_GROUP_M 4

__device__ __forceinline__ int big_swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void phase_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// =============================================================================================
// NT "w4": 256x256 tile, FOUR waves (one per SIMD, each with the whole 512-register file), wave tile 128x128 =
// 4x4 MFMA 32x32 accumulators (16 independent chains, one ds_read_b128 per two MFMAs).  Fragments are double-
// buffered in registers and the loop is software-pipelined ACROSS K-tiles: the single barrier of a K-tile sits after
// the last LDS reads of the current stage and before its last 16 MFMAs, and the first fragments of the next K-tile
// are read right behind it, so the MFMA stream has no bubble at the K-tile boundary.  Persistent, LDS-DMA into two
// 64 KiB stages, whole next K-tile issued at the top of the current one, cross-tile prefetch, same LDS image
// (128-byte rows, XOR swizzle on the DMA source) and the same LDS-staged epilogue as gemm_nt_big_kernel.
// =============================================================================================
__global__ __launch_bounds__(256, 1) void gemm_nt_w4_kernel(const uint16_t* __restrict__ A, int64_t lda,
                                                            const uint16_t* __restrict__ B, int64_t ldb,
                                                            uint16_t* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                            const float* __restrict__ alpha_dev, int tiles_m, int tiles_n) {
  constexpr int BM = 256, BN = 256;
  constexpr int OPND = 256 * 128;     // one operand K-tile: 256 rows x 128 B
  constexpr int STAGE = 2 * OPND;     // A | B
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE + 4 * 4096];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  const int ntiles = tiles_m * tiles_n;
  const int nk = K / 64;
  char* epi = smem + 2 * STAGE + wave * 4096;

  auto coords = [&](int tile, int& m0, int& n0) {
    const int group_size = BIG_GROUP_M * tiles_n;
    const int group = tile / group_size;
    const int first_m = group * BIG_GROUP_M;
    const int gm = min(tiles_m - first_m, BIG_GROUP_M);
    const int in_group = tile - group * group_size;
    m0 = (first_m + in_group % gm) * BM;
    n0 = (in_group / gm) * BN;
  };

  // DMA: operand tile = 32 wave-instructions (8 rows each); wave w issues q = i*4 + w, i = 0..7, i.e. rows
  // r0 + 32*i with r0 = wave*8 + lane/8.  The swizzled chunk (lane&7) ^ ((r>>1)&7) does not depend on i, so a lane
  // needs just (r0, column offset) — no per-instruction pointer array (which hipcc spilled to scratch).
  const int r0 = wave * 8 + (lane >> 3);
  const int coff = ((lane & 7) ^ ((r0 >> 1) & 7)) * 8;
  // LDS-DMA addresses = wave-uniform base of the staged tile (SGPRs, moves by 128 bytes per K-tile) + per-lane byte offsets
  // that only change with the tile (row clamp at the matrix edge): no 64-bit VALU arithmetic in the K loop
  // (the launcher only takes shapes with M % 256 == 0 and N % 256 == 0, so no row needs clamping: ONE lane offset per operand,
  // the 32-row step between a wave's pieces goes into the uniform base)
  const unsigned offa = (unsigned)(((int64_t)r0 * lda + coff) * 2), offb = (unsigned)(((int64_t)r0 * ldb + coff) * 2);
  const uint16_t* s_a = A;
  const uint16_t* s_b = B;
  auto set_ptrs = [&](int tile) {
    int m0, n0;
    coords(tile, m0, n0);
    s_a = A + (int64_t)m0 * lda;
    s_b = B + (int64_t)n0 * ldb;
  };
  // instruction e (0..15) of the next K-tile: even = the A piece e/2, odd = the B piece e/2
  auto issue_one = [&](char* stage, int k0, int e) {
    const int i = e >> 1;
    if (e & 1)
      dma16_saddr_asm(s_b + (int64_t)(32 * i) * ldb + k0, offb, stage + OPND + (i * 4 + wave) * 1024);
    else
      dma16_saddr_asm(s_a + (int64_t)(32 * i) * lda + k0, offa, stage + (i * 4 + wave) * 1024);
  };
  auto issue = [&](char* stage, int k0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) issue_one(stage, k0, e);
  };
  struct Frags {
    bf16x8_t a[4], b[4];
  };
  auto read_frags = [&](Frags& f, const char* stage, int ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f.a[i] = *reinterpret_cast<const bf16x8_t*>(stage + big_swz(wm * 128 + i * 32 + l31, ks * 2 + hi));
      f.b[i] = *reinterpret_cast<const bf16x8_t*>(stage + OPND + big_swz(wn * 128 + i * 32 + l31, ks * 2 + hi));
    }
  };

  const int first = xcd_remap(blockIdx.x, gridDim.x);
  if (first >= ntiles) return;
  int s_tile = first, s_k = 0;
  set_ptrs(s_tile);
  auto advance_staged = [&]() {
    s_k += 64;
    if (s_k >= K) {
      s_k = 0;
      s_tile += gridDim.x;
      if (s_tile < ntiles) set_ptrs(s_tile);
    }
  };
  float alpha = alpha_dev ? *alpha_dev : 1.f;
  asm volatile("; alpha pinned" : "+v"(alpha));

  issue(smem, s_k);
  advance_staged();
  wait_vm<0>();
  phase_barrier();
  Frags fc, fn;
  read_frags(fc, smem, 0);

  int st = 0;
  for (int tile = first; tile < ntiles; tile += gridDim.x) {
    f32x16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto mfma4 = [&](const Frags& f, int i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma32(f.b[j], f.a[i], acc[i][j]);  // D'[n][m]: lane owns a C row
    };
    // 16 MFMAs with one LDS-DMA instruction of the next K-tile behind every second one (an LDS-DMA occupies the wave's issue
    // port for ~60 cycles = two MFMAs of queued matrix work; four in a row drained the pipe: run 26)
    auto mfma16_dma = [&](const Frags& f, char* stage, int k0, int e0, bool on) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = mfma32(f.b[j], f.a[i], acc[i][j]);
          if ((j & 1) && on) issue_one(stage, k0, e0 + i * 2 + (j >> 1));
        }
      }
    };
    auto mfma16 = [&](const Frags& f) {
#pragma unroll
      for (int i = 0; i < 4; ++i) mfma4(f, i);
    };

    for (int kt = 0; kt < nk; ++kt) {
      const bool more = s_tile < ntiles;  // workgroup-uniform: another K-tile (this or a later tile) to stage
      const char* cur = smem + st * STAGE;
      char* nxt = smem + (st ^ 1) * STAGE;
      // k-step 0: the next k-step's fragments, then 16 MFMAs with the next K-tile's 16 DMA instructions spread between them
      read_frags(fn, cur, 1);
      __builtin_amdgcn_sched_barrier(0);  // keep the reads AHEAD of this k-step's MFMAs (hipcc sinks them next to their use)
      const bool dma_on = more;
      mfma16_dma(fc, nxt, s_k, 0, dma_on);   // k-step 0: instructions 0..7
      read_frags(fc, cur, 2);
      __builtin_amdgcn_sched_barrier(0);
      mfma16_dma(fn, nxt, s_k, 8, dma_on);   // k-step 1: instructions 8..15
      read_frags(fn, cur, 3);
      __builtin_amdgcn_sched_barrier(0);
      mfma16(fc);
      // K-tile boundary: all reads of `cur` are issued; once they are back and the next stage has landed, everybody
      // may move on — the last 16 MFMAs of this K-tile then run while the next K-tile's first fragments arrive
      if (more) advance_staged();
      wait_vm<0>();
      phase_barrier();
      if (more) read_frags(fc, nxt, 0);
      __builtin_amdgcn_sched_barrier(0);
      mfma16(fn);
      st ^= 1;
    }

    int m0, n0;
    coords(tile, m0, n0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int mrow0 = m0 + wm * 128 + i * 32;
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {  // 64-column halves of the wave's 128 columns
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            bf16x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[i][nh * 2 + jj][4 * g + e] * alpha);
            const int c = jj * 4 + g;
            *reinterpret_cast<bf16x4_t*>(epi + l31 * 128 + ((c ^ (l31 & 7)) << 4) + hi * 8) = o;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int c = it * 64 + lane;
          const int row = c >> 3, ch = c & 7;
          const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(epi + row * 128 + ((ch ^ (row & 7)) << 4));
          const int gm = mrow0 + row, gn = n0 + wn * 128 + nh * 64 + ch * 8;
          if (gm < M && gn < N) st_bf16x8(C + (int64_t)gm * ldc + gn, v);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
  }
}

void plm_launch_gemm_nt_w4(int slots, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, uint16_t* C, int64_t ldc, int64_t M,
                           int64_t N, int64_t K, const float* alpha_dev, hipStream_t s) {
  const int tm = (int)plm_cdiv(M, 256), tn = (int)plm_cdiv(N, 256);
  const int ntiles = tm * tn;
  hipLaunchKernelGGL(gemm_nt_w4_kernel, dim3(ntiles < slots ? ntiles : slots), dim3(256), 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K,
                     alpha_dev, tm, tn);
}
