// Micro-benchmark: per-CU store rate of global_store_dwordx4 for different footprints of one wave-instruction.
// mode 0: 8 rows x 128 B (row stride ld bytes)   mode 1: 4 rows x 256 B   mode 2: 2 rows x 512 B   mode 3: 1 KiB contiguous
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__global__ __launch_bounds__(512) void k(char* base, long ld, int mode, int iters, int nt) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seg = mode == 0 ? 8 : mode == 1 ? 16 : mode == 2 ? 32 : 64;  // lanes per row
  const int rows_per_inst = 64 / seg;
  u32x4 v = {1u, 2u, 3u, (unsigned)lane};
  // each workgroup owns a 256-row x 512-byte output tile per iteration (like the GEMM epilogue), tiles striped over the buffer
  for (int it = 0; it < iters; ++it) {
    const long tile = (long)blockIdx.x + (long)it * gridDim.x;
    char* t0 = base + (tile % nt) * 512;           // column block
    t0 += (tile / nt) * 256 * ld;                  // row block
    // wave handles 16 stores = 16 KB: rows [wave*32, wave*32+32) x 512 B
    for (int s = 0; s < 16; ++s) {
      const int r = (lane / seg) + rows_per_inst * (s % (32 / rows_per_inst));  // row within the wave's 32 rows
      const int cb = (lane % seg) * 16 + (s / (32 / rows_per_inst)) * seg * 16;  // byte column within 512
      if (mode == 3 && s >= 8) continue;
      char* p = t0 + (long)(wave * 32 + r) * ld + (cb % 512);
      *reinterpret_cast<u32x4*>(p) = v;
    }
  }
}
int main(int argc, char** argv) {
  const long ld = argc > 1 ? atol(argv[1]) : 8192;   // bytes per output row
  const int nt = (int)(ld / 512);
  const long rows = 32768;
  char* buf; hipMalloc(&buf, rows * ld);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int nwg = argc > 2 ? atoi(argv[2]) : 256;
  const int iters = (int)((rows / 256) * nt / 256);
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(nwg), dim3(512), 0, 0, buf, ld, mode, iters, nt);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)nwg * iters * 8 * 16 * 1024;
      if (rep) printf("ld=%ld nwg=%d mode=%d  %.3f ms  %.2f TB/s  %.1f GB/s/CU\n", ld, nwg, mode, ms, bytes / ms / 1e9, bytes / ms / 1e6 / nwg);
    }
  }
  return 0;
}
