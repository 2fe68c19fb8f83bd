// Micro-benchmark: LDS-DMA (global_load_lds_dwordx4) fill rate per CU with nothing else going on.
// One 512-thread workgroup per CU; every wave streams 1 KiB pieces (8 rows x 128 B of a row-major matrix, like a GEMM
// operand tile) into its own LDS ring with at most `depth` instructions in flight.  argv: rows-stride-bytes, total MB, depth.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ void dma16(const void* g, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
template <int DEPTH>
__global__ __launch_bounds__(512) void k(const char* base, long ld, long rows_total, int iters, int wgs) {
  __shared__ __attribute__((aligned(1024))) char smem[8 * 16 * 1024];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + wave * 16 * 1024;
  // wave w of workgroup b walks K (128 B per step) over an 8-row strip; strips are distinct per wave
  const long strip = ((long)blockIdx.x * 8 + wave) % (rows_total / 8);
  const char* p = base + (strip * 8 + (lane >> 3)) * ld + (lane & 7) * 16;
  const long kmax = ld / 128;
  long kk = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      dma16(p + kk * 128, lds0 + u * 1024);
      kk = (kk + 1 == kmax) ? 0 : kk + 1;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
int main(int argc, char** argv) {
  const long ld = argc > 1 ? atol(argv[1]) : 8192;
  const long mb = argc > 2 ? atol(argv[2]) : 64;
  const long rows = mb * 1024 * 1024 / ld;
  char* buf;
  hipMalloc(&buf, rows * ld);
  hipMemset(buf, 1, rows * ld);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 2000;
  for (int depth : {2, 4, 8, 16}) {
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (depth == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, buf, ld, rows, iters, 256);
      if (depth == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, buf, ld, rows, iters, 256);
      if (depth == 8) hipLaunchKernelGGL(k<8>, dim3(256), dim3(512), 0, 0, buf, ld, rows, iters, 256);
      if (depth == 16) hipLaunchKernelGGL(k<16>, dim3(256), dim3(512), 0, 0, buf, ld, rows, iters, 256);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double bytes = 256.0 * 8 * iters * 16 * 1024;
    printf("ld=%ld buf=%ldMB depth=%d  %.3f ms  %.2f TB/s  %.1f GB/s/CU\n", ld, mb, depth, best, bytes / best / 1e9, bytes / best / 1e6 / 256);
  }
  return 0;
}
