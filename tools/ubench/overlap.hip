// Micro-benchmark: what do the two waves of one SIMD (waves w and w+4 of a 512-thread workgroup) share?
// One workgroup per CU (100 KB of LDS), every wave runs `iters` rounds of its role's instruction block between two
// s_memtime reads; roles are chosen per wave group (waves 0-3 / waves 4-7):
//   M  16 back-to-back v_mfma_f32_32x32x16_bf16 on 4 independent accumulators          (512 cycles of matrix pipe)
//   V  64 independent v_fma_f32                                                        E  32 v_exp_f32
//   C  32 v_cvt_pk_bf16_f32 + 32 v_fma                                                 L  16 ds_read_b128 + 16 v_fma
//   X  one wave interleaving: per MFMA, F filler VALU (F = 4 / 6 / 8) - "fillers hidden per MFMA gap"
//   XL1 / XL2  one wave: 16 MFMAs with a ds_read_b128 behind every / every second one (results consumed a round later)
//   XG   one wave, the k-step of a 128x128-per-wave GEMM: 16 MFMAs + 8 ds_read_b128 + 4 LDS-DMA (global_load_lds_dwordx4, L2-resident source)
//   XGS  as XG, but the LDS-DMA source is a 64 MB stream that EVERY workgroup walks in the same order (the operand sharing of a
//        GEMM: one HBM / Infinity-Cache fetch per XCD, L2 hits for its other 31 CUs);  XGP: the same stream, private per workgroup pair of rows
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

enum Role { IDLE = 0, MFMA = 1, VFMA = 2, VEXP = 3, VCVT = 4, LDS = 5, MIX4 = 6, MIX6 = 7, MIX8 = 8, MIX8E = 9, MFMA8 = 10, MFMA16S = 11, MFMA1 = 12, MFMA2 = 13, MFMA_A = 14, XL1 = 15, XL2 = 16, XG = 17, XGV = 18, XGS = 19, XGP = 20, XR = 21 };

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

template <int role0, int role1>
__global__ __launch_bounds__(512, 2) void k(int iters, long long* cycles, float* sink, const char* gbuf) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int role = wave < 4 ? role0 : role1;
  constexpr bool U4 = role0 == MFMA || role1 == MFMA || role0 == MFMA1 || role1 == MFMA1 || role0 == MFMA2 || role1 == MFMA2 || role0 >= MIX4 && role0 <= MIX8E || role1 >= MIX4 && role1 <= MIX8E || role0 == MFMA8 || role1 == MFMA8 || role0 == MFMA_A || role1 == MFMA_A || role0 >= XL1 || role1 >= XL1;
  constexpr bool U8 = role0 == MFMA8 || role1 == MFMA8;
  constexpr bool U16 = role0 == MFMA16S || role1 == MFMA16S;
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
  f32x16 acc8[4];  // four more accumulators for the 8-chain role
  typedef __attribute__((ext_vector_type(4))) float f32x4v;
  f32x4v accs[16];  // 16x16x32: sixteen 4-register accumulators
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc8[i][r] = 0.f;
  for (int i = 0; i < 16; ++i)
    for (int r = 0; r < 4; ++r) accs[i][r] = 0.f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if ((role0 == MFMA8 || role1 == MFMA8) && role == MFMA8) {  // 16 MFMAs on 8 independent accumulators
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i & 4) acc8[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc8[i & 3], 0, 0, 0);
        else acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
      }
    } else if ((role0 == XL1 || role1 == XL1) && role == XL1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
        const f32x4 q = *reinterpret_cast<const f32x4*>(smem + ((lane * 16 + i * 1024 + it * 64) & 65535));
        v[i] += q[0];
      }
    } else if ((role0 == XL2 || role1 == XL2) && role == XL2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
        if (i & 1) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(smem + ((lane * 16 + i * 1024 + it * 64) & 65535));
          v[i] += q[0];
        }
      }
    } else if ((role0 == XG || role1 == XG) && role == XG) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
        if (i & 1) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(smem + ((lane * 16 + i * 1024 + it * 64) & 65535));
          v[i] += q[0];
        }
        if ((i & 3) == 2) {  // LDS-DMA: 1 KiB per wave-instruction into the upper 32 KiB of the LDS block
          unsigned keep;
          const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)((const __attribute__((address_space(3))) void*)(smem + 65536 + wave * 4096 + (i >> 2) * 1024)));
          const char* src = gbuf + ((blockIdx.x * 8 + wave) * 4 + (i >> 2)) * 1024 + lane * 16;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
      }
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else if ((role0 == XGS || role1 == XGS || role0 == XGP || role1 == XGP) && (role == XGS || role == XGP)) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
        if (i & 1) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(smem + ((lane * 16 + i * 1024 + it * 64) & 65535));
          v[i] += q[0];
        }
        if ((i & 3) == 2) {
          unsigned keep;
          const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)((const __attribute__((address_space(3))) void*)(smem + 65536 + wave * 4096 + (i >> 2) * 1024)));
          // a [rows][64 k] bf16 panel walk: 8 rows x 128 B per instruction, row stride 1536 B (K = 768), 32 KiB per round and workgroup
          const long pos = ((long)it * 8 + wave) * 4 + (i >> 2);
          const long wgoff = role == XGP ? (long)(blockIdx.x & 31) * (2l << 20) : 0;
          const char* src = gbuf + (wgoff + (pos * 8 + (lane >> 3)) * 1536 + (lane & 7) * 16) % (60l << 20);
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
      }
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if ((role0 == XR || role1 == XR) && role == XR) {  // the ratio of the shipped 8-wave GEMM: 12 reads + 4 DMA per 16 MFMAs, shared stream
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
        if ((i & 3) != 3) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(smem + ((lane * 16 + i * 1024 + it * 64) & 65535));
          v[i] += q[0];
        }
        if ((i & 3) == 2) {
          unsigned keep;
          const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)((const __attribute__((address_space(3))) void*)(smem + 65536 + wave * 4096 + (i >> 2) * 1024)));
          const long pos = ((long)it * 8 + wave) * 4 + (i >> 2);
          const char* src = gbuf + ((pos * 8 + (lane >> 3)) * 1536 + (lane & 7) * 16) % (60l << 20);
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
      }
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if ((role0 == XGV || role1 == XGV) && role == XGV) {  // the same reads + DMA, no MFMAs
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i & 1) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(smem + ((lane * 16 + i * 1024 + it * 64) & 65535));
          v[i] += q[0];
        }
        if ((i & 3) == 2) {
          unsigned keep;
          const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)((const __attribute__((address_space(3))) void*)(smem + 65536 + wave * 4096 + (i >> 2) * 1024)));
          const char* src = gbuf + ((blockIdx.x * 8 + wave) * 4 + (i >> 2)) * 1024 + lane * 16;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
      }
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else if ((role0 == MFMA_A || role1 == MFMA_A) && role == MFMA_A) {  // 4 accumulators held in AGPRs (inline asm: hipcc prefers VGPRs)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i & 3]) : "v"(a), "v"(b));
    } else if ((role0 == MFMA1 || role1 == MFMA1) && role == MFMA1) {  // 16 MFMAs on ONE accumulator (dependent chain)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
    } else if ((role0 == MFMA2 || role1 == MFMA2) && role == MFMA2) {  // two chains
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 1], 0, 0, 0);
    } else if ((role0 == MFMA16S || role1 == MFMA16S) && role == MFMA16S) {  // 32 x v_mfma_f32_16x16x32_bf16 (same flops as 16 x 32x32x16) on 16 accumulators
#pragma unroll
      for (int i = 0; i < 32; ++i) accs[i & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, accs[i & 15], 0, 0, 0);
    } else if ((role0 == MFMA || role1 == MFMA) && role == MFMA) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
    } else if ((role0 == VFMA || role1 == VFMA) && role == VFMA) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(v[j], 1.0001f, 0.5f);
    } else if ((role0 == VEXP || role1 == VEXP) && role == VEXP) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_amdgcn_exp2f(v[j]);
    } else if ((role0 == VCVT || role1 == VCVT) && role == VCVT) {
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
    } else if ((role0 == LDS || role1 == LDS) && role == LDS) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(smem + ((lane * 16 + j * 1024 + it * 64) & 65535));
        v[j] = __builtin_fmaf(v[j], 1.0001f, q[0] + q[3]);
      }
    } else if ((role0 == MIX4 || role1 == MIX4) && role == MIX4) {
      mix_round<4, false>(acc, a, b, v);
    } else if ((role0 == MIX6 || role1 == MIX6) && role == MIX6) {
      mix_round<6, false>(acc, a, b, v);
    } else if ((role0 == MIX8 || role1 == MIX8) && role == MIX8) {
      mix_round<8, false>(acc, a, b, v);
    } else if ((role0 == MIX8E || role1 == MIX8E) && role == MIX8E) {
      mix_round<8, true>(acc, a, b, v);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  if (U4)
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int j = 0; j < 16; ++j) s += v[j];
  if (U8)
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) s += acc8[i][r];
  if (U16)
    for (int i = 0; i < 16; ++i)
      for (int r = 0; r < 4; ++r) s += accs[i][r];
  if (s == 12345.678f) sink[0] = s;
  if (lane == 0 && blockIdx.x == 0) cycles[wave] = t1 - t0;
}

template <int R0, int R1>
void run(const char* n0, const char* n1, long long* cyc, float* sink, hipEvent_t e0, hipEvent_t e1, const char* gbuf) {
  const int iters = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<R0, R1>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(cyc, 0, 8 * sizeof(long long));
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<R0, R1>), dim3(256), dim3(512), 100 * 1024, 0, iters, cyc, sink, gbuf);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    if (rep) printf("waves0-3=%-4s waves4-7=%-4s  cycles/round: w0 %7.1f  w4 %7.1f   wall %.3f ms\n", n0, n1, (double)h[0] / iters, (double)h[4] / iters, ms);
  }
}

int main() {
  long long* cyc;
  float* sink;
  (void)hipMalloc(&cyc, 8 * sizeof(long long));
  (void)hipMalloc(&sink, 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  printf("# cycles per round (s_memtime, wave 0 / wave 4 of workgroup 0); a round = 16 MFMA 32x32x16 (or 32 MFMA 16x16x32) = 512 cycles of matrix pipe at peak;\n# wall = chip-wide ms for 2000 rounds on 256 workgroups\n");
char* gbuf;
  (void)hipMalloc(&gbuf, 64l << 20);
  (void)hipMemset(gbuf, 0, 64l << 20);
#define RUN(A, B) run<A, B>(#A, #B, cyc, sink, e0, e1, gbuf)
  RUN(MFMA1, IDLE); RUN(MFMA2, IDLE); RUN(MFMA, IDLE); RUN(MFMA8, IDLE); RUN(MFMA16S, IDLE);
  RUN(MFMA_A, IDLE); RUN(MFMA_A, MFMA_A); RUN(MFMA_A, VFMA);
  RUN(MFMA1, MFMA1); RUN(MFMA2, MFMA2); RUN(MFMA, MFMA); RUN(MFMA8, MFMA8); RUN(MFMA16S, MFMA16S);
  RUN(VFMA, IDLE); RUN(VFMA, VFMA); RUN(MFMA, VFMA); RUN(MFMA8, VFMA);
  RUN(VEXP, IDLE); RUN(VEXP, VEXP); RUN(MFMA, VEXP);
  RUN(VCVT, IDLE); RUN(MFMA, VCVT);
  RUN(LDS, IDLE); RUN(LDS, LDS); RUN(MFMA, LDS);
  RUN(MIX4, IDLE); RUN(MIX6, IDLE); RUN(MIX8, IDLE); RUN(MIX8E, IDLE); RUN(MIX4, MIX4); RUN(MIX8, MIX8); RUN(MIX8E, MIX8E);
  RUN(VFMA, VEXP); RUN(VFMA, LDS);
  RUN(XL1, IDLE); RUN(XL2, IDLE); RUN(XG, IDLE); RUN(XGV, IDLE); RUN(XL1, XL1); RUN(XL2, XL2); RUN(XG, XG); RUN(XGV, XGV);
  RUN(XGS, IDLE); RUN(XGS, XGS); RUN(XGP, XGP); RUN(XR, IDLE); RUN(XR, XR);
  return 0;
}
