// Can a CU stream HBM and feed its matrix pipes at the same time under this board's power cap?
// Two 4-wave workgroups per CU (72 KiB of LDS each forces exactly two): role A = back-to-back v_mfma_f32_16x16x32_bf16 on random
// register operands (no memory traffic at all), role B = a streaming copy (16-byte loads / stores, 16 loads in flight per lane).
// Timed: A alone, B alone, both in one launch.  If t(A+B) ~ max(tA, tB) the two overlap; if ~ tA + tB they do not (energy or fabric).
// This is the question behind gemm_duo.hip (one workgroup's epilogue under the other's K loop).
// Build: hipcc -O3 --offload-arch=gfx950 mfma_hbm_overlap.hip -o mfma_hbm_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__global__ __launch_bounds__(256, 2) void k(const u32x4_t* __restrict__ rnd, float* __restrict__ sink, const u32x4_t* __restrict__ src,
                                            u32x4_t* __restrict__ dst, long n16, int mfma_iters, int do_a, int do_b, int b_waves_only) {
  __shared__ char lds[73728];
  lds[threadIdx.x] = 0;
  const int half = gridDim.x >> 1;
  if ((int)blockIdx.x < half) {
    if (!do_a) return;
    // role A: 8 independent accumulators, operands from a random buffer (the power draw of an MFMA depends on its data)
    const int lane = threadIdx.x;
    bf16x8_t a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(bf16x8_t, rnd[(blockIdx.x * 256 + lane) * 6 + i]);
    for (int i = 0; i < 2; ++i) b[i] = __builtin_bit_cast(bf16x8_t, rnd[(blockIdx.x * 256 + lane) * 6 + 4 + i]);
    f32x4_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i * 2 + j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) sink[0] = s;
  } else {
    if (!do_b) return;
    const int wg = blockIdx.x - half;
    const long stride = (long)half * 256;
    long i = (long)wg * 256 + threadIdx.x;
    // 16 loads in flight per lane
    for (; i + 15 * stride < n16; i += 16 * stride) {
      u32x4_t v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
      for (int u = 0; u < 16; ++u) __builtin_nontemporal_store(v[u], dst + i + u * stride);
    }
  }
  if (lds[threadIdx.x] == 77) sink[1] = 1.f;
}

static float run(hipStream_t s, int cus, const u32x4_t* rnd, float* sink, const u32x4_t* src, u32x4_t* dst, long n16, int iters, int a, int b) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(2 * cus), dim3(256), 0, s, rnd, sink, src, dst, n16, iters, a, b, 0);
  hipEventRecord(e0, s);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(2 * cus), dim3(256), 0, s, rnd, sink, src, dst, n16, iters, a, b, 0);
  hipEventRecord(e1, s);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const long bytes = 1l << 30;  // 1 GiB read + 1 GiB written per launch
  const long n16 = bytes / 16;
  u32x4_t *src, *dst, *rnd;
  float* sink;
  hipMalloc(&src, bytes);
  hipMalloc(&dst, bytes);
  hipMalloc(&rnd, (size_t)cus * 256 * 6 * 16);
  hipMalloc(&sink, 64);
  std::vector<unsigned> h((size_t)cus * 256 * 6 * 4);
  srand(1);
  for (auto& x : h) {
    // two random bf16 in [-1, 1): sign random, exponent 0x3f0..0x3f7 region, random mantissa
    unsigned lo = (rand() & 0x807f) | (0x3e00 + ((rand() & 3) << 7)), hi = (rand() & 0x807f) | (0x3e00 + ((rand() & 3) << 7));
    x = lo | (hi << 16);
  }
  hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemset(src, 1, bytes);
  hipStream_t s;
  hipStreamCreate(&s);
  // role B alone first, to size role A to the same duration
  const float tb = run(s, cus, rnd, sink, src, dst, n16, 0, 0, 1);
  printf("B alone (copy 1 GiB -> 1 GiB on %d workgroups, one per CU): %.3f ms = %.2f TB/s (read + write)\n", cus, tb, 2.0 * bytes / tb / 1e9);
  int iters = 20000;
  float ta = run(s, cus, rnd, sink, src, dst, n16, iters, 1, 0);
  iters = (int)(iters * tb / ta);
  for (int rep = 0; rep < 3; ++rep) {
    ta = run(s, cus, rnd, sink, src, dst, n16, iters, 1, 0);
    const float tb2 = run(s, cus, rnd, sink, src, dst, n16, 0, 0, 1);
    const float tab = run(s, cus, rnd, sink, src, dst, n16, iters, 1, 1);
    const double fl = (double)cus * 4 * iters * 8 * 16384.0;
    printf("rep %d: A alone %.3f ms (%.0f TFLOP/s)   B alone %.3f ms (%.2f TB/s)   A + B in one launch %.3f ms  -> %.2f x max, %.2f x sum\n", rep, ta,
           fl / ta / 1e9, tb2, 2.0 * bytes / tb2 / 1e9, tab, tab / (ta > tb2 ? ta : tb2), tab / (ta + tb2));
  }
  return 0;
}
