// Which two workgroups share a CU when a grid of 2 x (CU count) workgroups with 72 KiB of LDS each is launched?
// gemm_duo.hip starts the second workgroup of every CU half a tile period late and guesses "second" = blockIdx >= grid / 2.
// Build: hipcc -O2 --offload-arch=gfx950 duo_census.hip -o duo_census ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>

__global__ __launch_bounds__(256, 2) void census(unsigned* out, int spin) {
  __shared__ char lds[73728];
  lds[threadIdx.x] = (char)threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    const unsigned long long t = __builtin_readcyclecounter();
    out[blockIdx.x * 4 + 0] = hw;
    out[blockIdx.x * 4 + 1] = xcc;
    out[blockIdx.x * 4 + 2] = (unsigned)t;
    out[blockIdx.x * 4 + 3] = lds[5];
  }
  // stay resident so that every block of the grid is co-resident with its partner
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, grid = 2 * cus;
  unsigned* d;
  hipMalloc(&d, grid * 16);
  std::vector<unsigned> h(grid * 4);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(census, dim3(grid), dim3(256), 0, 0, d, 400000);
    hipMemcpy(h.data(), d, grid * 16, hipMemcpyDeviceToHost);
    std::map<unsigned long long, std::vector<int>> by_cu;
    for (int b = 0; b < grid; ++b) {
      const unsigned hw = h[b * 4], xcc = h[b * 4 + 1] & 15;
      const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      by_cu[((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(b);
    }
    int pairs_ok = 0, two = 0;
    for (auto& kv : by_cu) {
      if (kv.second.size() == 2) {
        ++two;
        if ((kv.second[0] < cus) != (kv.second[1] < cus)) ++pairs_ok;
      }
    }
    printf("rep %d: %d CUs, grid %d: distinct CU ids %zu, CUs with exactly two workgroups %d, of those one from each half of the grid %d\n",
           rep, cus, grid, by_cu.size(), two, pairs_ok);
    if (rep == 0) {
      int shown = 0;
      for (auto& kv : by_cu) {
        if (shown++ >= 6) break;
        printf("  cu key %llx:", kv.first);
        for (int b : kv.second) printf(" %d", b);
        printf("\n");
      }
    }
  }
  return 0;
}
