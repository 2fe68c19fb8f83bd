// How many independent VALU instructions hide in the shadow of one v_mfma_f32_32x32x16_bf16 when they are PLACED between the MFMAs?
// The round (16 MFMAs on 4 rotating accumulators, F fillers after each) is one hand-written asm block, so the placement is exact.
// One or two waves per SIMD, operands random or zero.  Fillers: v_fma_f32 (kind 0), every 4th a v_exp_f32 (kind 1),
// v_cvt_pk_bf16_f32 (kind 2), ds_read_b128 as every 4th (kind 3).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_gap mfma_gap.hip ; run on an MI355X:  ./mfma_gap 1   (0 = zero operands)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>

#define STR2(x) #x
#define STR(x) STR2(x)
#define MF(c) "v_mfma_f32_32x32x16_bf16 v[" c "], v[64:67], v[68:71], v[" c "]\n\t"
#define FMA(r) "v_fma_f32 v" #r ", v" #r ", v88, v89\n\t"
#define EXP(r) "v_exp_f32 v" #r ", v" #r "\n\t"
#define CVT(r) "v_cvt_pk_bf16_f32 v" #r ", v" #r ", v88\n\t"
#define DSR(r) "ds_read_b128 v[92:95], v90\n\t"

template <int KIND, int R>
__device__ __forceinline__ void filler() {
#define CASE(r)                                                                     \
  if constexpr (R == r) {                                                           \
    if constexpr (KIND == 0) asm volatile("v_fma_f32 v" #r ", v" #r ", v88, v89"); \
    if constexpr (KIND == 1) asm volatile("v_exp_f32 v" #r ", v" #r);              \
    if constexpr (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 v" #r ", v" #r ", v88"); \
    if constexpr (KIND == 3) asm volatile("ds_read_b128 v[92:95], v90");           \
  }
  CASE(72) CASE(73) CASE(74) CASE(75) CASE(76) CASE(77) CASE(78) CASE(79) CASE(80) CASE(81) CASE(82) CASE(83) CASE(84) CASE(85) CASE(86) CASE(87)
#undef CASE
}

template <int F, int KIND>
__device__ __forceinline__ void round_asm() {
  // filler j uses register 72 + (j & 15)
#define FILL(j)                                                     \
  if (F > (j)) {                                                    \
    if (KIND == 1 && ((j) & 3) == 3) filler<1, 72 + ((j) & 15)>();  \
    else if (KIND == 2) filler<2, 72 + ((j) & 15)>();               \
    else if (KIND == 3 && ((j) & 3) == 3) filler<3, 72>();          \
    else filler<0, 72 + ((j) & 15)>();                              \
  }
#define STEP(c, base)                                                                            \
  asm volatile("v_mfma_f32_32x32x16_bf16 v[" c "], v[64:67], v[68:71], v[" c "]");             \
  FILL(0) FILL(1) FILL(2) FILL(3) FILL(4) FILL(5) FILL(6) FILL(7) FILL(8) FILL(9) FILL(10) FILL(11)
  STEP("0:15", 0) STEP("16:31", 0) STEP("32:47", 0) STEP("48:63", 0)
  STEP("0:15", 0) STEP("16:31", 0) STEP("32:47", 0) STEP("48:63", 0)
  STEP("0:15", 0) STEP("16:31", 0) STEP("32:47", 0) STEP("48:63", 0)
  STEP("0:15", 0) STEP("16:31", 0) STEP("32:47", 0) STEP("48:63", 0)
}

template <int F, int KIND>
__global__ __launch_bounds__(512, 1) void k(const uint32_t* in, float* out, long long* cyc, int rounds) {
  __shared__ float lds[4096];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 1.0f;
  __syncthreads();
  uint32_t w[8];
  for (int i = 0; i < 8; ++i) w[i] = in[lane * 8 + i];
  // every register the round touches is set up and torn down inside asm; the compiler is told about all of v0..v95
  asm volatile(
      "v_mov_b32 v64, %0\n\tv_mov_b32 v65, %1\n\tv_mov_b32 v66, %2\n\tv_mov_b32 v67, %3\n\t"
      "v_mov_b32 v68, %4\n\tv_mov_b32 v69, %5\n\tv_mov_b32 v70, %6\n\tv_mov_b32 v71, %7\n\t"
      "v_mov_b32 v88, 0x3f800347\n\tv_mov_b32 v89, 0.5\n\tv_mov_b32 v90, %8\n\t" ::"v"(w[0]),
      "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(lane * 16)
      : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v88", "v89", "v90");
#define Z(r) asm volatile("v_mov_b32 v" #r ", 0" ::: "v" #r);
  Z(0) Z(1) Z(2) Z(3) Z(4) Z(5) Z(6) Z(7) Z(8) Z(9) Z(10) Z(11) Z(12) Z(13) Z(14) Z(15) Z(16) Z(17) Z(18) Z(19) Z(20) Z(21) Z(22) Z(23) Z(24) Z(25) Z(26) Z(27) Z(28) Z(29) Z(30) Z(31)
  Z(32) Z(33) Z(34) Z(35) Z(36) Z(37) Z(38) Z(39) Z(40) Z(41) Z(42) Z(43) Z(44) Z(45) Z(46) Z(47) Z(48) Z(49) Z(50) Z(51) Z(52) Z(53) Z(54) Z(55) Z(56) Z(57) Z(58) Z(59) Z(60) Z(61) Z(62) Z(63)
  Z(72) Z(73) Z(74) Z(75) Z(76) Z(77) Z(78) Z(79) Z(80) Z(81) Z(82) Z(83) Z(84) Z(85) Z(86) Z(87) Z(92) Z(93) Z(94) Z(95)
  const long long t0 = clock64();
  for (int it = 0; it < rounds; ++it) {
    round_asm<F, KIND>();
    if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  const long long t1 = clock64();
  float s;
  asm volatile("v_add_f32 %0, v0, v72\n\tv_add_f32 %0, %0, v16\n\tv_add_f32 %0, %0, v92" : "=v"(s));
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int F, int KIND>
static void run(int threads, const uint32_t* din, float* dout, long long* dcyc) {
  const int rounds = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<F, KIND>), dim3(256), dim3(threads), 0, 0, din, dout, dcyc, 50);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<F, KIND>), dim3(256), dim3(threads), 0, 0, din, dout, dcyc, rounds);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
  const double mf = 16.0 * rounds;
  const double tf = 256.0 * (threads / 64) * mf * 32768.0 / (ms * 1e-3) / 1e12;
  printf("%s  F=%2d kind=%d: clock64 ticks/MFMA(wave 0) %7.2f   wall %.3f ms  %6.0f TFLOP/s  -> SIMD time per MFMA %.1f ns\n", threads == 256 ? "1 wave/SIMD " : "2 waves/SIMD", F, KIND, (double)c / mf, ms, tf,
         ms * 1e6 / (mf * (threads / 256)));
}

int main(int argc, char** argv) {
  const bool zero = argc > 1 && atoi(argv[1]) == 0;
  uint32_t* h = (uint32_t*)malloc(64 * 8 * 4);
  for (int i = 0; i < 64 * 8; ++i) {
    const uint32_t lo = 0x3f80u ^ ((rand() & 0x7f)) ^ ((rand() & 1) << 15), hi = 0x3f80u ^ ((rand() & 0x7f)) ^ ((rand() & 1) << 15);
    h[i] = zero ? 0u : (lo | (hi << 16));
  }
  uint32_t* din; float* dout; long long* dcyc;
  hipMalloc(&din, 64 * 8 * 4); hipMalloc(&dout, 4 * 512 * 256); hipMalloc(&dcyc, 8 * 256);
  hipMemcpy(din, h, 64 * 8 * 4, hipMemcpyHostToDevice);
  printf("operands: %s\n", zero ? "zeros" : "random bf16 in +-[1,2)");
  for (int threads : {256, 512}) {
#define R(F, K) run<F, K>(threads, din, dout, dcyc)
    R(0, 0); R(2, 0); R(3, 0); R(4, 0); R(5, 0); R(6, 0); R(8, 0); R(10, 0); R(12, 0); R(4, 1); R(8, 1); R(12, 1); R(4, 2); R(8, 2); R(4, 3); R(8, 3);
  }
  return 0;
}
