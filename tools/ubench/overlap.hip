// Micro-benchmark: what do the two waves of one SIMD (waves w and w+4 of a 512-thread workgroup) share?
// One workgroup per CU (100 KB of LDS), every wave runs `iters` rounds of its role's instruction block between two
// s_memtime reads; roles are chosen per wave group (waves 0-3 / waves 4-7):
//   M  16 back-to-back v_mfma_f32_32x32x16_bf16 on 4 independent accumulators          (512 cycles of matrix pipe)
//   V  64 independent v_fma_f32                                                        E  32 v_exp_f32
//   C  32 v_cvt_pk_bf16_f32 + 32 v_fma                                                 L  16 ds_read_b128 + 16 v_fma
//   X  one wave interleaving: per MFMA, F filler VALU (F = 4 / 6 / 8) - "fillers hidden per MFMA gap"
//   -  idle (the group exits at once)
// Prints cycles per round for every configuration: T(M,-), T(-,V), T(M,V) ... so that "max" (overlap) vs "sum" (one issue
// port) can be read off directly.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

enum Role { IDLE = 0, MFMA = 1, VFMA = 2, VEXP = 3, VCVT = 4, LDS = 5, MIX4 = 6, MIX6 = 7, MIX8 = 8, MIX8E = 9 };

template <int F, bool EXPS>
__device__ __forceinline__ void mix_round(f32x16 (&acc)[4], bf16x8 a, bf16x8 b, float (&v)[16]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
#pragma unroll
    for (int f = 0; f < F; ++f) {
      const int j = (i * F + f) & 15;
      if (EXPS && (f & 3) == 3) v[j] = __builtin_amdgcn_exp2f(v[j]);
      else v[j] = __builtin_fmaf(v[j], 1.0001f, 0.5f);
    }
  }
}

__global__ __launch_bounds__(512, 2) void k(int role0, int role1, int iters, long long* cycles, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int role = wave < 4 ? role0 : role1;
  for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<float*>(smem)[i] = (float)i;
  __syncthreads();
  if (role == IDLE) return;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * (lane - e)); }
  float v[16];
  for (int j = 0; j < 16; ++j) v[j] = 0.01f * (lane + j);
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (role == MFMA) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
    } else if (role == VFMA) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(v[j], 1.0001f, 0.5f);
    } else if (role == VEXP) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_amdgcn_exp2f(v[j]);
    } else if (role == VCVT) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
          typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
          bf16x2 p;
          p[0] = (__bf16)v[j];
          p[1] = (__bf16)v[j + 1];
          v[j] = __builtin_fmaf((float)p[0], 1.0001f, 0.5f);
          v[j + 1] = __builtin_fmaf((float)p[1], 1.0001f, 0.25f);
        }
    } else if (role == LDS) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(smem + ((lane * 16 + j * 1024 + it * 64) & 65535));
        v[j] = __builtin_fmaf(v[j], 1.0001f, q[0] + q[3]);
      }
    } else if (role == MIX4) {
      mix_round<4, false>(acc, a, b, v);
    } else if (role == MIX6) {
      mix_round<6, false>(acc, a, b, v);
    } else if (role == MIX8) {
      mix_round<8, false>(acc, a, b, v);
    } else if (role == MIX8E) {
      mix_round<8, true>(acc, a, b, v);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int j = 0; j < 16; ++j) s += v[j];
  if (s == 12345.678f) sink[0] = s;
  if (lane == 0 && blockIdx.x == 0) cycles[wave] = t1 - t0;
}

int main() {
  long long* cyc;
  float* sink;
  hipMalloc(&cyc, 8 * sizeof(long long));
  hipMalloc(&sink, 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  const char* names[] = {"-", "M", "V", "E", "C", "L", "X4", "X6", "X8", "X8e"};
  const int cfgs[][2] = {{MFMA, IDLE}, {IDLE, MFMA}, {MFMA, MFMA}, {VFMA, IDLE}, {VFMA, VFMA}, {MFMA, VFMA}, {VEXP, IDLE}, {VEXP, VEXP},
                         {MFMA, VEXP}, {VCVT, IDLE}, {MFMA, VCVT}, {LDS, IDLE}, {LDS, LDS}, {MFMA, LDS}, {MIX4, IDLE}, {MIX6, IDLE},
                         {MIX8, IDLE}, {MIX8E, IDLE}, {MIX4, MIX4}, {MIX8, MIX8}, {MIX8E, MIX8E}, {VFMA, VEXP}, {VFMA, LDS}};
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("# cycles per round (s_memtime, wave 0 / wave 4 of workgroup 0); wall = chip-wide ms for %d rounds on 256 workgroups\n", iters);
  for (auto& c : cfgs) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(cyc, 0, 8 * sizeof(long long));
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 100 * 1024, 0, c[0], c[1], iters, cyc, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      long long h[8];
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      if (rep) printf("waves0-3=%-3s waves4-7=%-3s  cycles/round: w0 %7.1f  w4 %7.1f   wall %.3f ms\n", names[c[0]], names[c[1]], (double)h[0] / iters,
                      (double)h[4] / iters, ms);
    }
  }
  return 0;
}
