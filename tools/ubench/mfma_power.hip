// Power-limited MFMA rate: what does a real operand stream cost at the board's power cap?
// tools/power_probe.py showed the shipped GEMMs sit at the 1400 W cap on random operands and run 17-41 % faster on zeros (same
// instruction stream).  This probe prices the pieces under that cap: one workgroup of 8 waves per CU (two per SIMD, a 128x64
// accumulator tile per wave, as the shipped 256x256 kernels), operands random or zero, and per k32 step of the wave
//   mode 0  16 x v_mfma_f32_32x32x16_bf16, fragments resident in registers (two alternating sets)
//   mode 1  32 x v_mfma_f32_16x16x32_bf16 (same flops, a quarter of the accumulator registers per instruction)
//   mode 2  mode 0 + 12 ds_read_b128 that refresh the fragments (the shipped kernels' read ratio)
//   mode 3  mode 0 +  8 ds_read_b128 (the ratio of a 128x128-per-wave tile)
//   mode 4  mode 2 +  4 global_load_lds_dwordx4 per wave from a 60 MB stream every workgroup walks in the same order
//   mode 5  mode 1 + 12 ds_read_b128
//   mode 6 / 7  modes 1 / 0 with the loops swapped (the B fragment is the one reused by consecutive MFMAs)
// Prints TFLOP/s (wall clock) and the average socket power sampled from hwmon while the mode runs.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip -lpthread
#include <hip/hip_runtime.h>
#include <glob.h>
#include <limits.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define CHECK(x)                                                                         \
  do {                                                                                   \
    hipError_t e__ = (x);                                                                \
    if (e__ != hipSuccess) {                                                             \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e__)); \
      exit(1);                                                                           \
    }                                                                                    \
  } while (0)

__device__ __forceinline__ bf16x8 as_frag(u32x4 v) {
  union {
    u32x4 u;
    bf16x8 b;
  } c;
  c.u = v;
  return c.b;
}

__device__ __forceinline__ u32x4 lds_read(const char* smem, unsigned off) {
  return *reinterpret_cast<const u32x4*>(smem + off);
}

template <int MODE>
__global__ __launch_bounds__(512) void k(const u32x4* __restrict__ rnd, const char* __restrict__ stream, long stream_bytes, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // 64 KB of LDS filled with the operand pattern (random or zero)
  for (int i = threadIdx.x; i < 4096; i += 512) reinterpret_cast<u32x4*>(smem)[i] = rnd[(blockIdx.x * 4096 + i) & 65535];
  __syncthreads();
  constexpr bool M16 = MODE == 1 || MODE == 5 || MODE == 6;
  constexpr bool SWAP = MODE == 6 || MODE == 7;
  constexpr int READS = MODE == 2 || MODE == 4 || MODE == 5 ? 12 : MODE == 3 ? 8 : 0;
  constexpr bool DMA = MODE == 4;
  u32x4 fr[2][12];  // two sets of 12 fragments (8 bf16 per lane each): a0..a7, b0..b3 for 16x16x32;  a0..a3 x 2 k16 steps, b0..b1 x 2 for 32x32x16
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int f = 0; f < 12; ++f) fr[s][f] = lds_read(smem, ((s * 12 + f) * 1024 + lane * 16 + wave * 64) & 65535);
  f32x16 acc32[8];
  f32x4 acc16[32];
  if (!M16) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc16[i][r] = 0.f;
  }
  long spos = (long)wave * 4096 + lane * 16;
  const unsigned dma_lds = 65536 + wave * 4096;  // 32 KB DMA landing zone behind the operand pattern
#pragma unroll 1
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // refresh the other set for the next half-iteration
      if (READS > 0) {
        const unsigned base = (unsigned)((it + s) * 1040 + lane * 16 + wave * 64);
#pragma unroll
        for (int f = 0; f < READS; ++f) fr[s ^ 1][f] = lds_read(smem, (base + f * 4096) & 65535 & ~15u);
      }
      if (DMA) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const char* src = stream + spos;
          const unsigned dst = __builtin_amdgcn_readfirstlane(dma_lds + g * 1024);
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
          spos += 32768;
          if (spos >= stream_bytes) spos -= stream_bytes;
        }
      }
      if (!M16) {
        // two k16 steps: fragments 0-3 (A) x 4-5 (B), then 6-9 (A) x 10-11 (B)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int x = 0; x < 8; ++x) {
            const int i = SWAP ? x % 4 : x / 2, j = SWAP ? x / 4 : x % 2;
            acc32[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(fr[s][h * 6 + i]), as_frag(fr[s][h * 6 + 4 + j]), acc32[i * 2 + j], 0, 0, 0);
          }
      } else {
#pragma unroll
        for (int x = 0; x < 32; ++x) {
          const int i = SWAP ? x % 8 : x / 4, j = SWAP ? x / 8 : x % 4;
          acc16[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(fr[s][i]), as_frag(fr[s][8 + j]), acc16[i * 4 + j], 0, 0, 0);
        }
      }
    }
    if (DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float t = 0.f;
  if (!M16) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc32[i][r];
  } else {
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) t += acc16[i][r];
  }
  if (t == 123.456f) sink[threadIdx.x] = t;
}

// hwmon power file of the HIP device in use (the box has eight GPUs; other tenants' cards show up in /sys as well)
static std::string power_file() {
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, sizeof bus, 0) != hipSuccess) return "";
  for (char* c = bus; *c; ++c) *c = (char)tolower(*c);
  glob_t g;
  std::string best;
  if (glob("/sys/class/drm/card*/device", 0, nullptr, &g) == 0) {
    for (size_t i = 0; i < g.gl_pathc && best.empty(); ++i) {
      char real[512];
      if (!realpath(g.gl_pathv[i], real)) continue;
      std::string r(real);
      for (auto& c : r) c = (char)tolower(c);
      if (r.size() < strlen(bus) || r.compare(r.size() - strlen(bus), strlen(bus), bus) != 0) continue;
      for (const char* leaf : {"power1_input", "power1_average"}) {
        glob_t h;
        const std::string pat = std::string(g.gl_pathv[i]) + "/hwmon/hwmon*/" + leaf;
        if (glob(pat.c_str(), 0, nullptr, &h) == 0 && h.gl_pathc > 0) best = h.gl_pathv[0];
        globfree(&h);
        if (!best.empty()) break;
      }
    }
    globfree(&g);
  }
  return best;
}

struct Sampler {
  std::string path;
  std::atomic<bool> stop{false};
  double sum = 0;
  long n = 0;
  std::thread th;
  void start() {
    stop = false, sum = 0, n = 0;
    th = std::thread([this] {
      while (!stop) {
        FILE* f = fopen(path.c_str(), "r");
        long v;
        if (f) {
          if (fscanf(f, "%ld", &v) == 1) sum += v / 1e6, ++n;
          fclose(f);
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
      }
    });
  }
  double finish() {
    stop = true;
    th.join();
    return n ? sum / n : 0.0;
  }
};

template <int MODE>
static void run(const char* name, const u32x4* pattern, const char* tag, const char* stream, long stream_bytes, float* sink, Sampler& smp, double seconds) {
  const int iters = 4000, lds = 100 * 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto launch = [&] { hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), lds, 0, pattern, stream, stream_bytes, iters, sink); };
  launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float one;
  CHECK(hipEventElapsedTime(&one, e0, e1));
  const int warm = (int)(0.6e3 / one) + 1, n = (int)(seconds * 1e3 / one) + 1;
  for (int i = 0; i < warm; ++i) launch();  // let the power controller settle
  CHECK(hipDeviceSynchronize());
  smp.start();
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < n; ++i) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  const double watts = smp.finish();
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = 2.0 * 32 * 32 * 16 * 16 * (double)iters * 8 * 256 * n;
  const double cyc_ideal = 512.0 * iters;  // matrix-pipe cycles per launch at 100 % utilisation
  printf("%-44s %-6s %8.1f TFLOP/s  %7.1f W  %6.3f ms/launch  (matrix pipe needs %.3f ms at 2.4 GHz)\n", name, tag, flops / ms / 1e9, watts, ms / n, 2 * cyc_ideal / 2.4e6);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
  const long stream_bytes = 60l << 20;
  std::vector<unsigned> h(65536 * 4);
  unsigned x = 12345u;
  for (auto& w : h) {  // bf16 pairs ~ uniform in (-2, 2): random sign, exponent 126..127, random mantissa
    unsigned lo, hi;
    x = x * 1664525u + 1013904223u;
    lo = ((x >> 16) & 0x80ffu) | (0x3f00u + ((x >> 3) & 0x80u));
    x = x * 1664525u + 1013904223u;
    hi = ((x >> 16) & 0x80ffu) | (0x3f00u + ((x >> 3) & 0x80u));
    w = lo | (hi << 16);
  }
  u32x4 *rnd, *zer;
  char* stream_r;
  char* stream_z;
  float* sink;
  CHECK(hipMalloc(&rnd, 65536 * 16));
  CHECK(hipMalloc(&zer, 65536 * 16));
  CHECK(hipMalloc(&stream_r, stream_bytes));
  CHECK(hipMalloc(&stream_z, stream_bytes));
  CHECK(hipMalloc(&sink, 4096));
  CHECK(hipMemcpy(rnd, h.data(), 65536 * 16, hipMemcpyHostToDevice));
  CHECK(hipMemset(zer, 0, 65536 * 16));
  CHECK(hipMemset(stream_z, 0, stream_bytes));
  for (long o = 0; o < stream_bytes; o += 65536 * 16) CHECK(hipMemcpy(stream_r + o, rnd, 65536 * 16, hipMemcpyDeviceToDevice));
  Sampler smp;
  smp.path = power_file();
  printf("# power file: %s\n", smp.path.empty() ? "(none found)" : smp.path.c_str());
  for (int pass = 0; pass < 2; ++pass) {
    const u32x4* p = pass == 0 ? rnd : zer;
    const char* st = pass == 0 ? stream_r : stream_z;
    const char* tag = pass == 0 ? "random" : "zeros";
    run<0>("16 x mfma 32x32x16, registers only", p, tag, st, stream_bytes, sink, smp, seconds);
    run<1>("32 x mfma 16x16x32, registers only", p, tag, st, stream_bytes, sink, smp, seconds);
    run<7>("16 x mfma 32x32x16, B-major order", p, tag, st, stream_bytes, sink, smp, seconds);
    run<6>("32 x mfma 16x16x32, B-major order", p, tag, st, stream_bytes, sink, smp, seconds);
    run<3>("16 x mfma 32x32x16 +  8 ds_read_b128", p, tag, st, stream_bytes, sink, smp, seconds);
    run<2>("16 x mfma 32x32x16 + 12 ds_read_b128", p, tag, st, stream_bytes, sink, smp, seconds);
    run<5>("32 x mfma 16x16x32 + 12 ds_read_b128", p, tag, st, stream_bytes, sink, smp, seconds);
    run<4>("16 x mfma 32x32x16 + 12 reads + 4 LDS-DMA", p, tag, st, stream_bytes, sink, smp, seconds);
  }
  return 0;
}
