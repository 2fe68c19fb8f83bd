// Fused cross-entropy forward + backward over bf16 logits (engine/engine.py:81,111).
//
// One 1024-thread workgroup per token row.  The row (V bf16 values) is read from HBM
// exactly once into registers (NCH 16-byte chunks per lane), reduced with an online
// (max, sum-exp) pair, and dlogits = (softmax - onehot) * grad_scale is written back
// in place: algorithmic traffic = 2*V bytes read + 2*V bytes written per row.
#include "plm_device.h"

#define LOG2E_F 1.4426950408889634f

struct MaxSum {
  float m, s;
};

__device__ __forceinline__ MaxSum ms_combine(MaxSum a, MaxSum b) {
  const float m = fmaxf(a.m, b.m);
  // exp(-inf - m) = 0 for the empty element (m = -inf, s = 0); guard the (-inf) - (-inf) case
  const float sa = (a.m == -INFINITY) ? 0.f : a.s * __expf(a.m - m);
  const float sb = (b.m == -INFINITY) ? 0.f : b.s * __expf(b.m - m);
  return MaxSum{m, sa + sb};
}

__device__ __forceinline__ MaxSum block_maxsum(MaxSum v, MaxSum* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    MaxSum other{__shfl_xor(v.m, o, 64), __shfl_xor(v.s, o, 64)};
    v = ms_combine(v, other);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  MaxSum t = (lane < nw) ? sh[lane] : MaxSum{-INFINITY, 0.f};
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    MaxSum other{__shfl_xor(t.m, o, 64), __shfl_xor(t.s, o, 64)};
    t = ms_combine(t, other);
  }
  // lanes 0..15 of every wave now hold the block result in lane 0's group
  t.m = __shfl(t.m, 0, 64);
  t.s = __shfl(t.s, 0, 64);
  return t;
}

// Fast path: V % 8 == 0 and V <= 1024 * 8 * NCH; row resident in registers.
template <int NCH>
__global__ __launch_bounds__(1024) void ce_fwd_bwd_kernel(uint16_t* __restrict__ logits, const int64_t* __restrict__ targets,
                                                          float* __restrict__ loss_rows, int64_t V, int64_t ld,
                                                          float grad_scale) {
  __shared__ MaxSum sh[16];
  const int64_t row = PLM_REV_BLOCK();
  uint16_t* lr = logits + row * ld;
  const int nvec = (int)(V >> 3);
  const int64_t tgt = targets[row];
  const bool tgt_ok = tgt >= 0 && tgt < V;
  float xt = 0.f;
  if (tgt_ok) xt = bf2f(reinterpret_cast<const bf16_t*>(lr)[tgt]);  // read before anything is overwritten

  u32x4_t v[NCH];  // the row stays in registers as packed bf16 pairs
  MaxSum acc{-INFINITY, 0.f};
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = threadIdx.x + 1024 * i;
    if (c < nvec) {
      v[i] = *reinterpret_cast<const u32x4_t*>(lr + c * 8);
      float f[8];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        f[2 * w] = bf_lo(v[i][w]);
        f[2 * w + 1] = bf_hi(v[i][w]);
      }
      float mx = fmaxf(fmaxf(fmaxf(f[0], f[1]), fmaxf(f[2], f[3])), fmaxf(fmaxf(f[4], f[5]), fmaxf(f[6], f[7])));
      const float mxl = mx * LOG2E_F;
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += __builtin_amdgcn_exp2f(f[e] * LOG2E_F - mxl);
      acc = ms_combine(acc, MaxSum{mx, s});
    }
  }
  const MaxSum tot = block_maxsum(acc, sh);
  const float lse = tot.m + __logf(tot.s);
  if (threadIdx.x == 0) loss_rows[row] = tgt_ok ? (lse - xt) : 0.f;
  const float gs = tgt_ok ? grad_scale : 0.f;  // ignored rows get zero gradient
  const float lsel = lse * LOG2E_F;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = threadIdx.x + 1024 * i;
    if (c < nvec) {
      float p[8];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        p[2 * w] = __builtin_amdgcn_exp2f(bf_lo(v[i][w]) * LOG2E_F - lsel);
        p[2 * w + 1] = __builtin_amdgcn_exp2f(bf_hi(v[i][w]) * LOG2E_F - lsel);
      }
      u32x4_t o;
#pragma unroll
      for (int w = 0; w < 4; ++w) o[w] = pack_bf2(p[2 * w] * gs, p[2 * w + 1] * gs);
      *reinterpret_cast<u32x4_t*>(lr + c * 8) = o;
    }
  }
  // the target column gets softmax - 1: one 2-byte fix-up after the vector stores (ordered by the barrier) instead of
  // a compare + select on every element of every chunk
  __syncthreads();
  if (threadIdx.x == 0 && tgt_ok)
    reinterpret_cast<bf16_t*>(lr)[tgt] = f2bf((__builtin_amdgcn_exp2f(xt * LOG2E_F - lsel) - 1.f) * gs);
  // zero the pad columns V..ld (ld - V < 64)
  for (int64_t c = V + threadIdx.x; c < ld; c += 1024) lr[c] = 0;
}

// Generic path (any V): three passes over the row with 2-byte accesses.
__global__ __launch_bounds__(1024) void ce_fwd_bwd_generic_kernel(uint16_t* __restrict__ logits,
                                                                  const int64_t* __restrict__ targets,
                                                                  float* __restrict__ loss_rows, int64_t V, int64_t ld,
                                                                  float grad_scale) {
  __shared__ MaxSum sh[16];
  const int64_t row = PLM_REV_BLOCK();
  bf16_t* lr = reinterpret_cast<bf16_t*>(logits) + row * ld;
  const int64_t tgt = targets[row];
  const bool tgt_ok = tgt >= 0 && tgt < V;
  const float xt = tgt_ok ? bf2f(lr[tgt]) : 0.f;
  MaxSum acc{-INFINITY, 0.f};
  for (int64_t c = threadIdx.x; c < V; c += 1024) acc = ms_combine(acc, MaxSum{bf2f(lr[c]), 1.f});
  const MaxSum tot = block_maxsum(acc, sh);
  const float lse = tot.m + __logf(tot.s);
  if (threadIdx.x == 0) loss_rows[row] = tgt_ok ? (lse - xt) : 0.f;
  const float gs = tgt_ok ? grad_scale : 0.f;
  __syncthreads();
  for (int64_t c = threadIdx.x; c < V; c += 1024) {
    float p = __expf(bf2f(lr[c]) - lse);
    if (c == tgt) p -= 1.f;
    lr[c] = f2bf(p * gs);
  }
  for (int64_t c = V + threadIdx.x; c < ld; c += 1024) lr[c] = f2bf(0.f);
}

extern "C" int plm_ce_fwd_bwd(uint16_t* logits, const int64_t* targets, float* loss_rows, int64_t M, int64_t V, int64_t ld,
                              float grad_scale, void* stream) {
  PLM_REQUIRE(logits && targets && loss_rows, "plm_ce_fwd_bwd: null pointer");
  PLM_REQUIRE(M > 0 && V > 0 && ld >= V && ld - V < 1024, "plm_ce_fwd_bwd: bad shape M=%ld V=%ld ld=%ld", (long)M, (long)V, (long)ld);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)M), block(1024);
  const bool fast = (V % 8 == 0) && (ld % 8 == 0) && ((reinterpret_cast<uintptr_t>(logits) & 15) == 0) && V <= 1024 * 8 * 8;
  if (!fast) {
    hipLaunchKernelGGL(ce_fwd_bwd_generic_kernel, grid, block, 0, s, logits, targets, loss_rows, V, ld, grad_scale);
  } else {
    const int64_t nch = plm_cdiv(V / 8, 1024);
    if (nch <= 1) hipLaunchKernelGGL(ce_fwd_bwd_kernel<1>, grid, block, 0, s, logits, targets, loss_rows, V, ld, grad_scale);
    else if (nch <= 2) hipLaunchKernelGGL(ce_fwd_bwd_kernel<2>, grid, block, 0, s, logits, targets, loss_rows, V, ld, grad_scale);
    else if (nch <= 4) hipLaunchKernelGGL(ce_fwd_bwd_kernel<4>, grid, block, 0, s, logits, targets, loss_rows, V, ld, grad_scale);
    else if (nch <= 7) hipLaunchKernelGGL(ce_fwd_bwd_kernel<7>, grid, block, 0, s, logits, targets, loss_rows, V, ld, grad_scale);
    else hipLaunchKernelGGL(ce_fwd_bwd_kernel<8>, grid, block, 0, s, logits, targets, loss_rows, V, ld, grad_scale);
  }
  PLM_CHECK_LAUNCH("plm_ce_fwd_bwd");
  return PLM_OK;
}
