// bf16 MFMA GEMMs for gfx950 (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
//   nt : C[M,N] = alpha * A[M,K] · B[N,K]^T      (y = x W^T ; dX = dY (W^T)^T with the transposed weight copy)
//   tn : C[M,N] (+)= alpha * A[K,M]^T · B[K,N]    (dW = dY^T X, contraction over the token rows)
//
// Structure (both): 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each wave a 64x64
// sub-tile = 2x2 MFMA 32x32 accumulators), BK = 64, operands staged global -> registers -> LDS with the
// next tile's global loads in flight during the MFMAs of the current one (one LDS buffer, two barriers
// per K-step, ~3 workgroups per CU so other workgroups cover the barrier bubbles).
//
// nt LDS image: [128 rows][64 k] bf16 (128-byte rows); the 16-byte chunk index is XOR-ed with
//   (row>>1)&7 so the sixteen lanes of a ds_read_b128 service group (16 distinct rows mod 16, same
//   k-chunk) hit sixteen different 16-byte slots of the 256-byte bank row: conflict-free.
// tn LDS image: [8 column-subtiles][64 k-rows][16 cols] (32-byte rows, sub-tile stride 2048+128 B) read
//   with ds_read_b64_tr_b16, which hands lane t column t of a [4 k][16 col] block: the MFMA operand
//   with k = token row.  A 32-lane half covers two sub-tiles 128 B apart mod 256: conflict-free.
//
// Workgroup -> tile map: XCD-aware remap (each XCD gets a contiguous id range) and grouped
// rasterisation (GROUP_M row-panels x all column tiles) so A panels and B tiles are reused from L2.
#include <stdlib.h>

#include "plm_device.h"

#define GBM 128
#define GBN 128
#define GBK 64
#define GROUP_M 8
#define TN_SUB_STRIDE 2176  // 64 rows * 32 B + 128 B pad

__device__ __forceinline__ void tile_coords(int bid, int nwg, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int pid = xcd_remap(bid, nwg);
  const int group_size = GROUP_M * tiles_n;
  const int group = pid / group_size;
  const int first_m = group * GROUP_M;
  const int gm = min(tiles_m - first_m, GROUP_M);
  const int in_group = pid - group * group_size;
  tm = first_m + in_group % gm;
  tn = in_group / gm;
}

// ---------------------------------------------------------------------------------------------
// NT
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int nt_lds_off(int row, int chunk) {  // byte offset of a 16-byte chunk
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <bool OUT_F32, bool ACCUM>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const uint16_t* __restrict__ A, int64_t lda,
                                                      const uint16_t* __restrict__ B, int64_t ldb, void* __restrict__ Cv,
                                                      int64_t ldc, int M, int N, int K, const float* __restrict__ alpha_dev,
                                                      int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(16))) char smem[2 * GBM * GBK * 2];
  char* sA = smem;
  char* sB = smem + GBM * GBK * 2;

  int tm, tn;
  tile_coords(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * GBM, n0 = tn * GBN;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;

  // staging assignment: 4 chunks of A and 4 of B per thread
  int ld_row[4], ld_chunk[4];
  const uint16_t* a_ptr[4];
  const uint16_t* b_ptr[4];
  bool a_ok[4], b_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = i * 256 + t;
    ld_row[i] = idx >> 3;
    ld_chunk[i] = idx & 7;
    a_ok[i] = (m0 + ld_row[i]) < M;
    b_ok[i] = (n0 + ld_row[i]) < N;
    a_ptr[i] = A + (int64_t)(m0 + ld_row[i]) * lda + ld_chunk[i] * 8;
    b_ptr[i] = B + (int64_t)(n0 + ld_row[i]) * ldb + ld_chunk[i] * 8;
  }

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bf16x8_t ra[4], rb[4];
  auto g_load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool kin = (k0 + ld_chunk[i] * 8) < K;
      ra[i] = (a_ok[i] && kin) ? ld_bf16x8(a_ptr[i] + k0) : zero_bf16x8();
      rb[i] = (b_ok[i] && kin) ? ld_bf16x8(b_ptr[i] + k0) : zero_bf16x8();
    }
  };
  auto s_store = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int off = nt_lds_off(ld_row[i], ld_chunk[i]);
      *reinterpret_cast<bf16x8_t*>(sA + off) = ra[i];
      *reinterpret_cast<bf16x8_t*>(sB + off) = rb[i];
    }
  };

  const int nk = (K + GBK - 1) / GBK;
  g_load(0);
  s_store();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) g_load((kt + 1) * GBK);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8_t af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *reinterpret_cast<const bf16x8_t*>(sA + nt_lds_off(wm * 64 + i * 32 + l31, ks * 2 + hi));
        bfr[i] = *reinterpret_cast<const bf16x8_t*>(sB + nt_lds_off(wn * 64 + i * 32 + l31, ks * 2 + hi));
      }
      // D'[n][m] = sum_k B[n][k] A[m][k]: the lane ends up owning one C row (m) and runs of 4 consecutive n
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(bfr[j], af[i], acc[i][j]);
    }
    __syncthreads();
    if (kt + 1 < nk) {
      s_store();
      __syncthreads();
    }
  }

  const float alpha = alpha_dev ? *alpha_dev : 1.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 64 + i * 32 + l31;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 64 + j * 32 + 8 * g + 4 * hi;
        if (n >= N) continue;
        f32x4_t v = {acc[i][j][4 * g + 0] * alpha, acc[i][j][4 * g + 1] * alpha, acc[i][j][4 * g + 2] * alpha,
                     acc[i][j][4 * g + 3] * alpha};
        if (OUT_F32) {
          float* dst = reinterpret_cast<float*>(Cv) + (int64_t)m * ldc + n;
          if (n + 3 < N) {
            if (ACCUM) v += *reinterpret_cast<const f32x4_t*>(dst);
            *reinterpret_cast<f32x4_t*>(dst) = v;
          } else {
            for (int e = 0; e < 4 && n + e < N; ++e) dst[e] = ACCUM ? dst[e] + v[e] : v[e];
          }
        } else {
          uint16_t* dst = reinterpret_cast<uint16_t*>(Cv) + (int64_t)m * ldc + n;
          bf16x4_t o;
          o[0] = f2bf(v[0]); o[1] = f2bf(v[1]); o[2] = f2bf(v[2]); o[3] = f2bf(v[3]);
          if (n + 3 < N) {
            st_bf16x4(dst, o);
          } else {
            for (int e = 0; e < 4 && n + e < N; ++e) reinterpret_cast<bf16_t*>(dst)[e] = o[e];
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// NT v2: operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write),
// two LDS buffers, ONE barrier per K-step; C leaves through LDS as full 128-byte row segments.
// The DMA writes lane-linear (wave-uniform base + lane*16 B), so the XOR swizzle of nt_lds_off() is applied
// to the per-lane SOURCE address: LDS slot s of row r receives global chunk s ^ ((r>>1)&7).
// Needs K % 64 == 0 (no K tail), N % 8 == 0 and 16-byte aligned C rows; other shapes use gemm_nt_kernel.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

__device__ __forceinline__ void dma16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_void_t*)gsrc, (lds_void_t*)lds_wave_base, 16, 0, 0);
}

#define EPI_STRIDE_BF16 144   // 64 bf16 + 16 B pad: 16-byte aligned rows, 2-way at worst on the 8-byte writes
#define EPI_STRIDE_F32 272    // 64 fp32 + 16 B pad

template <bool OUT_F32, bool ACCUM>
__global__ __launch_bounds__(256) void gemm_nt_dma_kernel(const uint16_t* __restrict__ A, int64_t lda,
                                                          const uint16_t* __restrict__ B, int64_t ldb, void* __restrict__ Cv,
                                                          int64_t ldc, int M, int N, int K, const float* __restrict__ alpha_dev,
                                                          int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * 2 * GBM * GBK * 2];  // [buf][A|B][16 KiB]
  int tm, tn;
  tile_coords(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * GBM, n0 = tn * GBN;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;

  // DMA assignment: instruction i of wave w fills LDS rows (i*4+w)*8 .. +7 (1 KiB); lane -> (row, slot)
  const uint16_t* a_src[4];
  const uint16_t* b_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * 4 + wave) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    const int ar = min(m0 + row, M - 1), br = min(n0 + row, N - 1);  // tail rows: any valid address, results unused
    a_src[i] = A + (int64_t)ar * lda + chunk * 8;
    b_src[i] = B + (int64_t)br * ldb + chunk * 8;
  }
  auto dma_tile = [&](int buf, int k0) {
    char* dA = smem + buf * (2 * GBM * GBK * 2);
    char* dB = dA + GBM * GBK * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dma16(a_src[i] + k0, dA + (i * 4 + wave) * 1024);
      dma16(b_src[i] + k0, dB + (i * 4 + wave) * 1024);
    }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = K / GBK;
  dma_tile(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) dma_tile(cur ^ 1, (kt + 1) * GBK);
    const char* sA = smem + cur * (2 * GBM * GBK * 2);
    const char* sB = sA + GBM * GBK * 2;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8_t af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *reinterpret_cast<const bf16x8_t*>(sA + nt_lds_off(wm * 64 + i * 32 + l31, ks * 2 + hi));
        bfr[i] = *reinterpret_cast<const bf16x8_t*>(sB + nt_lds_off(wn * 64 + i * 32 + l31, ks * 2 + hi));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(bfr[j], af[i], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces of the next tile have landed
    __syncthreads();                                   // ... and everyone's; also: all reads of buf[cur] are done
  }

  // ---- epilogue: accumulators -> LDS (per-wave region) -> 16-byte row-contiguous global stores ----
  const float alpha = alpha_dev ? *alpha_dev : 1.f;
  constexpr int STRIDE = OUT_F32 ? EPI_STRIDE_F32 : EPI_STRIDE_BF16;
  char* epi = smem + wave * (32 * STRIDE);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    // 32 rows (m) x 64 cols (n) of this wave: lane owns row l31, 4-column runs
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = j * 32 + 8 * g + 4 * hi;
        f32x4_t v = {acc[i][j][4 * g + 0] * alpha, acc[i][j][4 * g + 1] * alpha, acc[i][j][4 * g + 2] * alpha,
                     acc[i][j][4 * g + 3] * alpha};
        if (OUT_F32) {
          *reinterpret_cast<f32x4_t*>(epi + l31 * STRIDE + col * 4) = v;
        } else {
          bf16x4_t o;
          o[0] = f2bf(v[0]); o[1] = f2bf(v[1]); o[2] = f2bf(v[2]); o[3] = f2bf(v[3]);
          *reinterpret_cast<bf16x4_t*>(epi + l31 * STRIDE + col * 2) = o;
        }
      }
    }
    __syncthreads();
    constexpr int CPR = OUT_F32 ? 16 : 8;            // 16-byte chunks per 64-column row
    constexpr int ITERS = 32 * CPR / 64;             // per-lane chunks for 32 rows
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int c = it * 64 + lane;
      const int row = c / CPR, cc = c % CPR;
      const int gm = m0 + wm * 64 + i * 32 + row;
      const int gn = n0 + wn * 64 + cc * (OUT_F32 ? 4 : 8);
      if (gm < M && gn < N) {
        if (OUT_F32) {
          f32x4_t v = *reinterpret_cast<const f32x4_t*>(epi + row * STRIDE + cc * 16);
          float* dst = reinterpret_cast<float*>(Cv) + (int64_t)gm * ldc + gn;
          if (ACCUM) v += *reinterpret_cast<const f32x4_t*>(dst);
          *reinterpret_cast<f32x4_t*>(dst) = v;
        } else {
          const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(epi + row * STRIDE + cc * 16);
          st_bf16x8(reinterpret_cast<uint16_t*>(Cv) + (int64_t)gm * ldc + gn, v);
        }
      }
    }
    if (i == 0) __syncthreads();
  }
}

bool plm_launch_gemm_nt_big(int variant, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, uint16_t* C, int64_t ldc,
                            int64_t M, int64_t N, int64_t K, const float* alpha_dev, void* workspace, size_t workspace_bytes,
                            hipStream_t s);  // gemm_big.hip
extern "C" int plm_rope_qk(uint16_t* qkv, const float* rope_cos, const float* rope_sin, int64_t B, int64_t T, int64_t nh, int64_t hd,
                           void* stream);

extern "C" int plm_gemm_bf16_nt_ws(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                                   int64_t N, int64_t K, int c_dtype, int accumulate, const float* alpha_dev, int variant,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  PLM_REQUIRE(A && B && C, "plm_gemm_bf16_nt: null pointer");
  PLM_REQUIRE(variant >= 0 && variant <= 7, "plm_gemm_bf16_nt_ex: variant must be 0..7");
  PLM_REQUIRE(M > 0 && N > 0 && K > 0 && M < (1 << 30) && N < (1 << 30) && K < (1 << 30), "plm_gemm_bf16_nt: bad shape M=%ld N=%ld K=%ld",
              (long)M, (long)N, (long)K);
  PLM_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, "plm_gemm_bf16_nt: K, lda, ldb must be multiples of 8 and ldc of 4 (K=%ld lda=%ld ldb=%ld ldc=%ld)",
              (long)K, (long)lda, (long)ldb, (long)ldc);
  PLM_REQUIRE(((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C)) & 15) == 0,
              "plm_gemm_bf16_nt: base pointers must be 16-byte aligned");
  PLM_REQUIRE(c_dtype == 0 || c_dtype == 1, "plm_gemm_bf16_nt: c_dtype must be 0 (bf16) or 1 (fp32)");
  PLM_REQUIRE(!(accumulate && c_dtype == 0), "plm_gemm_bf16_nt: accumulate needs an fp32 C");
  const int tiles_m = (int)plm_cdiv(M, GBM), tiles_n = (int)plm_cdiv(N, GBN);
  const dim3 grid((unsigned)(tiles_m * tiles_n)), block(256);
  hipStream_t s = (hipStream_t)stream;
  const bool force_v1 = plm_env().gemm_v1;
  const bool dma_shape = (K % GBK == 0) && (N % 8 == 0) && (ldc % 8 == 0);
  PLM_REQUIRE(variant <= 1 || dma_shape, "plm_gemm_bf16_nt_ex: variant %d needs K %% 64 == 0, N %% 8 == 0, ldc %% 8 == 0", variant);
  PLM_REQUIRE(variant <= 2 || c_dtype == 0, "plm_gemm_bf16_nt_ex: the big-tile variants write bf16 C only");
  const bool dma_ok = variant >= 2 || (variant == 0 && !force_v1 && dma_shape);
  if ((variant == 0 && dma_ok && c_dtype == 0) || variant >= 3) {
    if (plm_launch_gemm_nt_big(variant, A, lda, B, ldb, (uint16_t*)C, ldc, M, N, K, alpha_dev, workspace, workspace_bytes, s)) {
      PLM_CHECK_LAUNCH("plm_gemm_bf16_nt (big tile)");
      return PLM_OK;
    }
  }
#define PLM_NT_LAUNCH(KERN)                                                                                              \
  hipLaunchKernelGGL(KERN, grid, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, alpha_dev, tiles_m, tiles_n)
  if (dma_ok) {
    if (c_dtype == 0) PLM_NT_LAUNCH((gemm_nt_dma_kernel<false, false>));
    else if (accumulate) PLM_NT_LAUNCH((gemm_nt_dma_kernel<true, true>));
    else PLM_NT_LAUNCH((gemm_nt_dma_kernel<true, false>));
  } else {
    if (c_dtype == 0) PLM_NT_LAUNCH((gemm_nt_kernel<false, false>));
    else if (accumulate) PLM_NT_LAUNCH((gemm_nt_kernel<true, true>));
    else PLM_NT_LAUNCH((gemm_nt_kernel<true, false>));
  }
#undef PLM_NT_LAUNCH
  PLM_CHECK_LAUNCH("plm_gemm_bf16_nt");
  return PLM_OK;
}

extern "C" int plm_gemm_bf16_nt_ex(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                                   int64_t N, int64_t K, int c_dtype, int accumulate, const float* alpha_dev, int variant,
                                   void* stream) {
  return plm_gemm_bf16_nt_ws(A, lda, B, ldb, C, ldc, M, N, K, c_dtype, accumulate, alpha_dev, variant, nullptr, 0, stream);
}

extern "C" int plm_gemm_bf16_nt(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                                int64_t N, int64_t K, int c_dtype, int accumulate, const float* alpha_dev, void* stream) {
  return plm_gemm_bf16_nt_ws(A, lda, B, ldb, C, ldc, M, N, K, c_dtype, accumulate, alpha_dev, 0, nullptr, 0, stream);
}

// w_qkv projection with RoPE: qkv[M, 3*nh*hd] = x W^T with the q | k column blocks rotated (row m = position m % T) in the GEMM
// epilogue: the 16-byte chunks are rotated on their way from the transposition scratch to memory, one 16-byte table read per
// table and chunk (round 1 rotated accumulator fragments - per-lane table gathers, 58 us per call - and lost to the 31 us
// stand-alone pass, which remains the fallback: same bits).
bool plm_launch_gemm_nt_rope(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, uint16_t* C, int64_t ldc, int64_t M, int64_t N,
                             int64_t K, const float* rcos, const float* rsin, int64_t T, int64_t rope_cols, hipStream_t s);
extern "C" int plm_qkv_rope_bf16(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, uint16_t* QKV, int64_t ldq, int64_t M,
                                 int64_t K, const float* rope_cos, const float* rope_sin, int64_t B, int64_t T, int64_t nh, int64_t hd,
                                 void* stream) {
  PLM_REQUIRE(X && W && QKV && rope_cos && rope_sin, "plm_qkv_rope_bf16: null pointer");
  PLM_REQUIRE((hd == 64 || hd == 32 || hd == 128) && B > 0 && T > 0 && nh > 0 && M == B * T, "plm_qkv_rope_bf16: bad shape (hd 32 / 64 / 128, M == B*T)");
  const int64_t N = 3 * nh * hd;
  PLM_REQUIRE(ldq == N, "plm_qkv_rope_bf16: needs a dense output (ldq == 3*nh*hd)");
  if (hd == 64 && !plm_env().gemm_v1 &&  // the store-side rotation is built for 64-wide heads; other head dims take GEMM + the stand-alone pass
      plm_launch_gemm_nt_rope(X, ldx, W, ldw, QKV, ldq, M, N, K, rope_cos, rope_sin, T, 2 * nh * hd, (hipStream_t)stream)) {
    PLM_CHECK_LAUNCH("plm_qkv_rope_bf16");
    return PLM_OK;
  }
  if (int rc = plm_gemm_bf16_nt_ex(X, ldx, W, ldw, QKV, ldq, M, N, K, 0, 0, nullptr, 0, stream)) return rc;
  return plm_rope_qk(QKV, rope_cos, rope_sin, B, T, nh, hd, stream);
}

// fc1 of the SwiGLU MLP with the activation in the GEMM epilogue (models/components.py:50-56):
//   U[M, 2h] = X[M, K] W[2h, K]^T  (gate | up, kept for backward)  and  ACT[M, h] = bf16(bf16(silu(gate)) * up).
// One launch on the persistent 256x256 kernel when the shape qualifies (2h % 256 == 0, K % 64 == 0, M >= 512); otherwise the GEMM
// followed by plm_swiglu_fwd - the two paths produce the same bits.
bool plm_launch_gemm_nt_glu(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, uint16_t* C, int64_t ldc, uint16_t* act,
                            int64_t ldact, int64_t M, int64_t N, int64_t K, hipStream_t s);
extern "C" int plm_fc1_swiglu_bf16(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, uint16_t* U, uint16_t* ACT, int64_t M,
                                   int64_t h, int64_t K, void* stream) {
  PLM_REQUIRE(X && W && U && ACT, "plm_fc1_swiglu_bf16: null pointer");
  PLM_REQUIRE(M > 0 && h > 0 && K > 0 && h % 8 == 0, "plm_fc1_swiglu_bf16: bad shape (h %% 8 == 0)");
  const int64_t N = 2 * h;
  if (!plm_env().gemm_v1 && plm_launch_gemm_nt_glu(X, ldx, W, ldw, U, N, ACT, h, M, N, K, (hipStream_t)stream)) {
    PLM_CHECK_LAUNCH("plm_fc1_swiglu_bf16");
    return PLM_OK;
  }
  if (int rc = plm_gemm_bf16_nt_ex(X, ldx, W, ldw, U, N, M, N, K, 0, 0, nullptr, 0, stream)) return rc;
  return plm_swiglu_fwd(U, ACT, M, h, stream);
}

// Backward of the SwiGLU MLP's second half (models/components.py:55-57): d(act)[M, h] = dY[M, K] W2T[h, K]^T never reaches memory -
// the epilogue of that GEMM applies the SwiGLU backward with the saved fc1 output U[M, 2h] and writes DU[M, 2h] (d(gate) | d(up)).
// One launch when h % 256 == 0, K % 64 == 0, M >= 512; otherwise the GEMM followed by plm_swiglu_bwd (same bits; `scratch` must then
// hold M*h bf16 values for d(act), it is not touched by the fused path and may be NULL when the shape qualifies).
bool plm_launch_gemm_nt_glub(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint16_t* U, int64_t ldu, uint16_t* DU,
                             int64_t lddu, int64_t M, int64_t h, int64_t K, hipStream_t s);
extern "C" int plm_fc2_dx_swiglu_bwd_bf16(const uint16_t* dY, int64_t lddy, const uint16_t* W2T, int64_t ldw, const uint16_t* U, uint16_t* DU,
                                          uint16_t* scratch, int64_t M, int64_t h, int64_t K, void* stream) {
  PLM_REQUIRE(dY && W2T && U && DU, "plm_fc2_dx_swiglu_bwd_bf16: null pointer");
  PLM_REQUIRE(M > 0 && h > 0 && K > 0 && h % 8 == 0, "plm_fc2_dx_swiglu_bwd_bf16: bad shape (h %% 8 == 0)");
  if (!plm_env().gemm_v1 && plm_launch_gemm_nt_glub(dY, lddy, W2T, ldw, U, 2 * h, DU, 2 * h, M, h, K, (hipStream_t)stream)) {
    PLM_CHECK_LAUNCH("plm_fc2_dx_swiglu_bwd_bf16");
    return PLM_OK;
  }
  if (!scratch) {  // not an error of the shape: the caller retries with the buffer
    plm_set_error("plm_fc2_dx_swiglu_bwd_bf16: this shape (or PLM_GEMM_V1) takes the two-launch path and needs the M*h bf16 d(act) scratch");
    return PLM_E_WORKSPACE;
  }
  if (int rc = plm_gemm_bf16_nt_ex(dY, lddy, W2T, ldw, scratch, h, M, h, K, 0, 0, nullptr, 0, stream)) return rc;
  return plm_swiglu_bwd(scratch, U, DU, M, h, stream);
}

// ---------------------------------------------------------------------------------------------
// TN   C[i][j] = sum_k A[k][i] B[k][j]
// ---------------------------------------------------------------------------------------------
// MODE 0: C = alpha*acc   MODE 1: C += alpha*acc   MODE 2: raw partial into a split-K slab (ld = N)
template <int MODE>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const uint16_t* __restrict__ A, int64_t lda,
                                                      const uint16_t* __restrict__ B, int64_t ldb, float* __restrict__ C,
                                                      int64_t ldc, int M, int N, int K, int kchunk,
                                                      const float* __restrict__ alpha_dev, int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 8 * TN_SUB_STRIDE];
  char* sA = smem;
  char* sB = smem + 8 * TN_SUB_STRIDE;

  int tm, tn;
  tile_coords(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn);
  const int i0 = tm * GBM, j0 = tn * GBN;
  const int kbeg = blockIdx.y * kchunk;
  const int kend = min(K, kbeg + kchunk);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5, ib = (lane >> 4) & 1, t16 = lane & 15;

  int ld_krow[4], ld_off[4];
  const uint16_t* a_ptr[4];
  const uint16_t* b_ptr[4];
  bool a_ok[4], b_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = i * 256 + t;
    const int u = idx & 7, cl = u & 3, rl = u >> 2, rest = idx >> 3;
    const int c = (rest & 3) * 4 + cl;           // 16-byte chunk along the 128 columns
    ld_krow[i] = ((rest >> 2) << 1) | rl;        // 0..63
    ld_off[i] = (c >> 1) * TN_SUB_STRIDE + ld_krow[i] * 32 + (c & 1) * 16;
    a_ok[i] = (i0 + c * 8) < M;
    b_ok[i] = (j0 + c * 8) < N;
    a_ptr[i] = A + (int64_t)ld_krow[i] * lda + i0 + c * 8;
    b_ptr[i] = B + (int64_t)ld_krow[i] * ldb + j0 + c * 8;
  }

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bf16x8_t ra[4], rb[4];
  auto g_load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool kin = (k0 + ld_krow[i]) < kend;
      ra[i] = (a_ok[i] && kin) ? ld_bf16x8(a_ptr[i] + (int64_t)k0 * lda) : zero_bf16x8();
      rb[i] = (b_ok[i] && kin) ? ld_bf16x8(b_ptr[i] + (int64_t)k0 * ldb) : zero_bf16x8();
    }
  };
  auto s_store = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<bf16x8_t*>(sA + ld_off[i]) = ra[i];
      *reinterpret_cast<bf16x8_t*>(sB + ld_off[i]) = rb[i];
    }
  };
  // transpose-read of one 32-column operand block starting at tile column cb, k-step ks
  auto tr_frag = [&](const char* base, int cb, int ks) -> bf16x8_t {
    const char* p = base + ((cb >> 4) + ib) * TN_SUB_STRIDE + (ks * 16 + hi * 8 + (t16 >> 2)) * 32 + (t16 & 3) * 8;
    const s16x4_t lo = lds_read_tr16(p);
    const s16x4_t hh = lds_read_tr16(p + 4 * 32);
    return join_tr(lo, hh);
  };

  const int nk = (kend - kbeg + GBK - 1) / GBK;
  if (nk > 0) {
    g_load(kbeg);
    s_store();
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) g_load(kbeg + (kt + 1) * GBK);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8_t af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = tr_frag(sA, wm * 64 + i * 32, ks);
        bfr[i] = tr_frag(sB, wn * 64 + i * 32, ks);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(af[i], bfr[j], acc[i][j]);
    }
    __syncthreads();
    if (kt + 1 < nk) {
      s_store();
      __syncthreads();
    }
  }

  const float alpha = (MODE != 2 && alpha_dev) ? *alpha_dev : 1.f;
  float* out = (MODE == 2) ? C + (int64_t)blockIdx.y * M * N : C;
  const int64_t ld = (MODE == 2) ? N : ldc;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = j0 + wn * 64 + j * 32 + l31;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + wm * 64 + i * 32 + mfma32_row(r, hi);
        if (row >= M) continue;
        float* dst = out + (int64_t)row * ld + col;
        const float v = acc[i][j][r] * alpha;
        *dst = (MODE == 1) ? *dst + v : v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// TN v2: LDS-DMA + two LDS buffers + one barrier per K-step (needs K and the split chunk % 64 == 0).
// LDS image per operand: [64 k-rows][128 cols] bf16, 256-byte rows, written lane-linear by the DMA
// (one wave-instruction = 4 full rows: perfectly coalesced source reads).  The 32-byte column pairs of
// row k are ROTATED by 2*(k&3) pair positions (on the source address): the four rows of a transpose-read
// block then sit on four different 32-byte bank segments and the two 16-lane groups of a half-wave use
// the even/odd segments: ds_read_b64_tr_b16 is conflict-free.
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void gemm_tn_dma_kernel(const uint16_t* __restrict__ A, int64_t lda,
                                                          const uint16_t* __restrict__ B, int64_t ldb, float* __restrict__ C,
                                                          int64_t ldc, int M, int N, int K, int kchunk,
                                                          const float* __restrict__ alpha_dev, int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * 2 * GBK * GBM * 2];  // [buf][A|B][16 KiB]
  int tm, tn;
  tile_coords(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn);
  const int i0 = tm * GBM, j0 = tn * GBN;
  const int kbeg = blockIdx.y * kchunk;
  const int kend = min(K, kbeg + kchunk);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5, ib = (lane >> 4) & 1, t16 = lane & 15;

  const uint16_t* a_src[4];
  const uint16_t* b_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * 4 + wave) * 4 + (lane >> 4);      // k-row inside the tile
    const int pp = (lane & 15) >> 1, half = lane & 1;       // physical 32-byte pair / 16-byte half
    const int cb = (pp - 2 * (row & 3)) & 7;                // logical pair stored there
    const int col = cb * 16 + half * 8;
    a_src[i] = A + (int64_t)row * lda + min(i0 + col, M - 8);  // column tails: any valid address, results unused
    b_src[i] = B + (int64_t)row * ldb + min(j0 + col, N - 8);
  }
  auto dma_tile = [&](int buf, int k0) {
    char* dA = smem + buf * (2 * GBK * GBM * 2);
    char* dB = dA + GBK * GBM * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dma16_asm(a_src[i] + (int64_t)k0 * lda, dA + (i * 4 + wave) * 1024);
      dma16_asm(b_src[i] + (int64_t)k0 * ldb, dB + (i * 4 + wave) * 1024);
    }
  };
  auto tr_frag = [&](const char* base, int cb0, int ks) -> bf16x8_t {
    const int row = ks * 16 + hi * 8 + (t16 >> 2);
    const char* p = base + row * 256 + (((cb0 + ib) + 2 * (row & 3)) & 7) * 32 + (t16 & 3) * 8;
    return join_tr(lds_read_tr16(p), lds_read_tr16(p + 4 * 256));
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (kend - kbeg) / GBK;
  if (nk > 0) dma_tile(0, kbeg);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) dma_tile(cur ^ 1, kbeg + (kt + 1) * GBK);
    const char* sA = smem + cur * (2 * GBK * GBM * 2);
    const char* sB = sA + GBK * GBM * 2;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8_t af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = tr_frag(sA, (wm * 64 + i * 32) >> 4, ks);
        bfr[i] = tr_frag(sB, (wn * 64 + i * 32) >> 4, ks);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(af[i], bfr[j], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  const float alpha = (MODE != 2 && alpha_dev) ? *alpha_dev : 1.f;
  float* out = (MODE == 2) ? C + (int64_t)blockIdx.y * M * N : C;
  const int64_t ld = (MODE == 2) ? N : ldc;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = j0 + wn * 64 + j * 32 + l31;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + wm * 64 + i * 32 + mfma32_row(r, hi);
        if (row >= M) continue;
        float* dst = out + (int64_t)row * ld + col;
        const float v = acc[i][j][r] * alpha;
        *dst = (MODE == 1) ? *dst + v : v;
      }
    }
  }
}

// C (+)= alpha * sum_s slab[s]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, int64_t ldc, int M,
                                                            int N, int splits, int accumulate,
                                                            const float* __restrict__ alpha_dev) {
  const float alpha = alpha_dev ? *alpha_dev : 1.f;
  const int64_t nv = (int64_t)M * (N >> 2);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int nq = N >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    const int64_t row = i / nq;
    const int col = (int)(i - row * nq) * 4;
    f32x4_t s = *reinterpret_cast<const f32x4_t*>(ws + row * N + col);
    for (int k = 1; k < splits; ++k) s += *reinterpret_cast<const f32x4_t*>(ws + (int64_t)k * M * N + row * N + col);
    s *= alpha;
    float* dst = C + row * ldc + col;
    if (accumulate) s += *reinterpret_cast<const f32x4_t*>(dst);
    *reinterpret_cast<f32x4_t*>(dst) = s;
  }
}

static int tn_splits(int64_t M, int64_t N, int64_t K) {
  // The DMA kernels run 2 workgroups per CU (512 slots on 256 CUs): aim for just under 2 full rounds.
  const int64_t tiles = plm_cdiv(M, GBM) * plm_cdiv(N, GBN);
  if (tiles >= 512) return 1;
  int64_t s = 1024 / tiles;
  const int64_t max_by_k = K / 512 > 0 ? K / 512 : 1;  // keep >= 512 contraction rows per slab
  if (s > max_by_k) s = max_by_k;
  if (s > 32) s = 32;
  return (int)(s < 1 ? 1 : s);
}

bool plm_tn_big_plan(int64_t M, int64_t N, int64_t K, int* splits, int* rfull);  // gemm_big.hip
void plm_launch_gemm_tn_big(int splits, int rfull, int accumulate, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb,
                            float* C, int64_t ldc, float* slabs, int64_t M, int64_t N, int64_t K, const float* alpha_dev,
                            hipStream_t s);

// one place decides kernel + split layout, so the workspace query and the launch always agree.
// rfull: tile rows (of 256) that the big kernel computes without split (0 for the 128x128 kernels).
static int tn_plan(int64_t M, int64_t N, int64_t K, bool& big, int& rfull) {
  big = false;
  rfull = 0;
  if (!plm_env().tn_no_big && !plm_env().gemm_v1) {
    int sb = 1;
    if (plm_tn_big_plan(M, N, K, &sb, &rfull)) {
      big = true;
      return sb;
    }
  }
  rfull = 0;
  return tn_splits(M, N, K);
}

extern "C" size_t plm_gemm_tn_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  bool big;
  int rfull;
  const int s = tn_plan(M, N, K, big, rfull);
  const int64_t rows = M - (int64_t)rfull * 256;
  return (s > 1 && rows > 0) ? (size_t)s * (size_t)rows * (size_t)N * sizeof(float) : 0;
}

extern "C" int plm_gemm_bf16_tn(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                                int64_t N, int64_t K, int accumulate, const float* alpha_dev, void* workspace,
                                size_t workspace_bytes, void* stream) {
  PLM_REQUIRE(A && B && C, "plm_gemm_bf16_tn: null pointer");
  PLM_REQUIRE(M > 0 && N > 0 && K > 0 && M < (1 << 30) && N < (1 << 30) && K < (1 << 30), "plm_gemm_bf16_tn: bad shape M=%ld N=%ld K=%ld",
              (long)M, (long)N, (long)K);
  PLM_REQUIRE(M % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0,
              "plm_gemm_bf16_tn: M, N, lda, ldb must be multiples of 8 and ldc of 4 (M=%ld N=%ld lda=%ld ldb=%ld ldc=%ld)", (long)M, (long)N,
              (long)lda, (long)ldb, (long)ldc);
  PLM_REQUIRE(((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C)) & 15) == 0,
              "plm_gemm_bf16_tn: base pointers must be 16-byte aligned");
  const int tiles_m = (int)plm_cdiv(M, GBM), tiles_n = (int)plm_cdiv(N, GBN);
  bool big;
  int rfull;
  const int splits = tn_plan(M, N, K, big, rfull);
  hipStream_t s = (hipStream_t)stream;
  const dim3 block(256);
  if (big) {
    const int64_t rem_rows = M - (int64_t)rfull * 256;  // rows whose tiles are split over K
    const bool split_part = splits > 1 && rem_rows > 0;
    if (split_part) {
      const size_t need = (size_t)splits * (size_t)rem_rows * (size_t)N * sizeof(float);
      if (!workspace || workspace_bytes < need) {
        plm_set_error("plm_gemm_bf16_tn: workspace of %zu bytes required, %zu given", need, workspace_bytes);
        return PLM_E_WORKSPACE;
      }
      PLM_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "plm_gemm_bf16_tn: workspace must be 16-byte aligned");
    }
    plm_launch_gemm_tn_big(split_part ? splits : 1, split_part ? rfull : (int)plm_cdiv(M, 256), accumulate, A, lda, B, ldb, C, ldc,
                           (float*)workspace, M, N, K, alpha_dev, s);
    if (split_part) {
      const int64_t nv = rem_rows * (N / 4);
      int64_t rb = plm_cdiv(nv, 256);
      if (rb > 4096) rb = 4096;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)rb), block, 0, s, (const float*)workspace, C + (int64_t)rfull * 256 * ldc, ldc,
                         (int)rem_rows, (int)N, splits, accumulate, alpha_dev);
    }
    PLM_CHECK_LAUNCH("plm_gemm_bf16_tn (big tile)");
    return PLM_OK;
  }
  const bool force_v1 = plm_env().gemm_v1;
  const bool dma_ok = !force_v1 && (K % GBK == 0);
  if (splits == 1) {
    const dim3 grid((unsigned)(tiles_m * tiles_n), 1);
    if (dma_ok) {
      if (accumulate)
        hipLaunchKernelGGL(gemm_tn_dma_kernel<1>, grid, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, (int)K, alpha_dev, tiles_m, tiles_n);
      else
        hipLaunchKernelGGL(gemm_tn_dma_kernel<0>, grid, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, (int)K, alpha_dev, tiles_m, tiles_n);
    } else if (accumulate)
      hipLaunchKernelGGL(gemm_tn_kernel<1>, grid, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, (int)K, alpha_dev, tiles_m, tiles_n);
    else
      hipLaunchKernelGGL(gemm_tn_kernel<0>, grid, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, (int)K, alpha_dev, tiles_m, tiles_n);
    PLM_CHECK_LAUNCH("plm_gemm_bf16_tn");
    return PLM_OK;
  }
  const size_t need = (size_t)splits * (size_t)M * (size_t)N * sizeof(float);
  if (!workspace || workspace_bytes < need) {
    plm_set_error("plm_gemm_bf16_tn: workspace of %zu bytes required, %zu given", need, workspace_bytes);
    return PLM_E_WORKSPACE;
  }
  PLM_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "plm_gemm_bf16_tn: workspace must be 16-byte aligned");
  const int kchunk = (int)(plm_cdiv(plm_cdiv(K, splits), GBK) * GBK);
  const dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)splits);
  if (dma_ok)
    hipLaunchKernelGGL(gemm_tn_dma_kernel<2>, grid, block, 0, s, A, lda, B, ldb, (float*)workspace, (int64_t)N, (int)M, (int)N, (int)K, kchunk,
                       (const float*)nullptr, tiles_m, tiles_n);
  else
    hipLaunchKernelGGL(gemm_tn_kernel<2>, grid, block, 0, s, A, lda, B, ldb, (float*)workspace, (int64_t)N, (int)M, (int)N, (int)K, kchunk,
                       (const float*)nullptr, tiles_m, tiles_n);
  const int64_t nv = M * (N / 4);
  int64_t rb = plm_cdiv(nv, 256);
  if (rb > 4096) rb = 4096;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)rb), block, 0, s, (const float*)workspace, C, ldc, (int)M, (int)N, splits, accumulate,
                     alpha_dev);
  PLM_CHECK_LAUNCH("plm_gemm_bf16_tn");
  return PLM_OK;
}
