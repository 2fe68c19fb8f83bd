// HBM-bound kernels of the plainLM hot path for gfx950: parameter casts, embedding,
// RMSNorm (+residual), SwiGLU gate, small reductions and the AdamW tail.
// One wave (64 lanes) owns one row wherever a row reduction is needed; every global
// access is 8 or 16 bytes per lane and coalesced along the row.
#include "plm_device.h"

// ===========================================================================
// fp32 -> bf16 casts   (autocast weight casts, engine/engine.py:75)
// ===========================================================================
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst,
                                                            int64_t n) {
  const int64_t nvec = n >> 3;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(src + i * 8);
    const f32x4_t b = *reinterpret_cast<const f32x4_t*>(src + i * 8 + 4);
    bf16x8_t o;
    o[0] = f2bf(a[0]); o[1] = f2bf(a[1]); o[2] = f2bf(a[2]); o[3] = f2bf(a[3]);
    o[4] = f2bf(b[0]); o[5] = f2bf(b[1]); o[6] = f2bf(b[2]); o[7] = f2bf(b[3]);
    st_bf16x8(dst + i * 8, o);
  }
  // tail (n not a multiple of 8)
  const int64_t tail0 = nvec << 3;
  const int64_t t = tail0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) reinterpret_cast<bf16_t*>(dst)[t] = f2bf(src[t]);
}

// 64x64 tile: write dst (same layout) and dst_t (transposed) from one read of src.
__global__ __launch_bounds__(256) void cast_f32_bf16_t_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst,
                                                              uint16_t* __restrict__ dst_t, int64_t rows, int64_t cols,
                                                              int64_t ld_t) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];  // [col][row], 144-byte rows (16B aligned)
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (t >> 4) + 16 * i, c = (t & 15) * 4;
    const int64_t gr = r0 + r, gc = c0 + c;
    f32x4_t v = {0.f, 0.f, 0.f, 0.f};
    if (gr < rows && gc < cols) v = *reinterpret_cast<const f32x4_t*>(src + gr * cols + gc);
    bf16x4_t o;
    o[0] = f2bf(v[0]); o[1] = f2bf(v[1]); o[2] = f2bf(v[2]); o[3] = f2bf(v[3]);
    if (gr < rows && gc < cols) st_bf16x4(dst + gr * cols + gc, o);
    tile[c + 0][r] = o[0]; tile[c + 1][r] = o[1]; tile[c + 2][r] = o[2]; tile[c + 3][r] = o[3];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = (t >> 3) + 32 * i, rch = (t & 7) * 8;
    const int64_t gc = c0 + c, gr = r0 + rch;
    if (gc < cols && gr < rows) {
      const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(&tile[c][rch]);
      st_bf16x8(dst_t + gc * ld_t + gr, v);
    }
  }
}

// All stale weight shadows of a model in one launch: the 49 per-Linear casts of the 160M step are launch-latency bound
// (1 - 6 MB each, 5 - 7 us per launch against ~1 us of traffic).  The item table travels in the kernel arguments.
#define PLM_CAST_MULTI_MAX 56
struct CastGroup {
  const float* src[PLM_CAST_MULTI_MAX];
  uint16_t* dst[PLM_CAST_MULTI_MAX];
  uint16_t* dst_t[PLM_CAST_MULTI_MAX];
  int rows[PLM_CAST_MULTI_MAX], cols[PLM_CAST_MULTI_MAX], ld_t[PLM_CAST_MULTI_MAX];
  int block_base[PLM_CAST_MULTI_MAX + 1];  // first block of each item; [count] = number of blocks
  int count;
};

__global__ __launch_bounds__(256) void cast_f32_bf16_t_multi_kernel(CastGroup g) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];
  int it = 0;
  for (int q = 1; q < g.count; ++q)
    if ((int)blockIdx.x >= g.block_base[q]) it = q;  // block-uniform scalar search
  const int local = blockIdx.x - g.block_base[it];
  const int tiles_x = (g.cols[it] + 63) / 64;
  const int64_t rows = g.rows[it], cols = g.cols[it], ld_t = g.ld_t[it];
  const float* __restrict__ src = g.src[it];
  uint16_t* __restrict__ dst = g.dst[it];
  uint16_t* __restrict__ dst_t = g.dst_t[it];
  const int64_t r0 = (int64_t)(local / tiles_x) * 64, c0 = (int64_t)(local % tiles_x) * 64;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (t >> 4) + 16 * i, c = (t & 15) * 4;
    const int64_t gr = r0 + r, gc = c0 + c;
    f32x4_t v = {0.f, 0.f, 0.f, 0.f};
    if (gr < rows && gc < cols) v = *reinterpret_cast<const f32x4_t*>(src + gr * cols + gc);
    bf16x4_t o;
    o[0] = f2bf(v[0]); o[1] = f2bf(v[1]); o[2] = f2bf(v[2]); o[3] = f2bf(v[3]);
    if (gr < rows && gc < cols) st_bf16x4(dst + gr * cols + gc, o);
    tile[c + 0][r] = o[0]; tile[c + 1][r] = o[1]; tile[c + 2][r] = o[2]; tile[c + 3][r] = o[3];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = (t >> 3) + 32 * i, rch = (t & 7) * 8;
    const int64_t gc = c0 + c, gr = r0 + rch;
    if (gc < cols && gr < rows) {
      const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(&tile[c][rch]);
      st_bf16x8(dst_t + gc * ld_t + gr, v);
    }
  }
}

extern "C" int plm_cast_f32_bf16_t_multi(const plm_cast_item* items, int count, void* stream) {
  PLM_REQUIRE(items && count >= 1, "plm_cast_f32_bf16_t_multi: null pointer or empty list");
  for (int first = 0; first < count; first += PLM_CAST_MULTI_MAX) {
    const int n = count - first < PLM_CAST_MULTI_MAX ? count - first : PLM_CAST_MULTI_MAX;
    CastGroup g{};
    int base = 0;
    for (int i = 0; i < n; ++i) {
      const plm_cast_item& q = items[first + i];
      PLM_REQUIRE(q.src && q.dst && q.dst_t, "plm_cast_f32_bf16_t_multi: null pointer in item %d", first + i);
      PLM_REQUIRE(q.rows > 0 && q.cols > 0 && q.rows % 8 == 0 && q.cols % 8 == 0 && q.rows < (1ll << 31) && q.cols < (1ll << 31),
                  "plm_cast_f32_bf16_t_multi: item %d: rows=%ld cols=%ld must be positive multiples of 8", first + i, (long)q.rows, (long)q.cols);
      PLM_REQUIRE(q.ld_t >= q.rows && q.ld_t % 8 == 0 && q.ld_t < (1ll << 31), "plm_cast_f32_bf16_t_multi: item %d: ld_t=%ld must be >= rows and a multiple of 8",
                  first + i, (long)q.ld_t);
      g.src[i] = q.src;
      g.dst[i] = q.dst;
      g.dst_t[i] = q.dst_t;
      g.rows[i] = (int)q.rows;
      g.cols[i] = (int)q.cols;
      g.ld_t[i] = (int)q.ld_t;
      g.block_base[i] = base;
      const int64_t nb = plm_cdiv(q.rows, 64) * plm_cdiv(q.cols, 64);
      PLM_REQUIRE(base + nb < (1ll << 31), "plm_cast_f32_bf16_t_multi: too many tiles");
      base += (int)nb;
    }
    g.block_base[n] = base;
    g.count = n;
    hipLaunchKernelGGL(cast_f32_bf16_t_multi_kernel, dim3((unsigned)base), dim3(256), 0, (hipStream_t)stream, g);
    PLM_CHECK_LAUNCH("plm_cast_f32_bf16_t_multi");
  }
  return PLM_OK;
}

extern "C" int plm_cast_f32_bf16(const float* src, uint16_t* dst, int64_t n, void* stream) {
  PLM_REQUIRE(src && dst && n >= 0, "plm_cast_f32_bf16: null pointer or negative n");
  if (n == 0) return PLM_OK;
  const int64_t work = plm_cdiv(plm_cdiv(n, 8), 256);
  const int grid = (int)(work < ((int64_t)1 << 20) ? (work < 1 ? 1 : work) : ((int64_t)1 << 20));
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, n);
  PLM_CHECK_LAUNCH("plm_cast_f32_bf16");
  return PLM_OK;
}

// ===========================================================================
// device-scalar scaling (chunked lm_head + cross-entropy: the chunk GEMMs run in forward, before autograd knows the
// upstream gradient g; backward applies g with these two passes)
// ===========================================================================
__global__ __launch_bounds__(256) void scale_bf16_kernel(uint16_t* __restrict__ x, int64_t n, const float* __restrict__ alpha_dev) {
  const float alpha = *alpha_dev;
  const int64_t nvec = n >> 3;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(x + i * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) * alpha);
    st_bf16x8(x + i * 8, v);
  }
  const int64_t t = (nvec << 3) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) reinterpret_cast<bf16_t*>(x)[t] = f2bf(bf2f(reinterpret_cast<bf16_t*>(x)[t]) * alpha);
}

__global__ __launch_bounds__(256) void axpy_f32_kernel(float* __restrict__ out, const float* __restrict__ x, int64_t n,
                                                       const float* __restrict__ alpha_dev, int accumulate) {
  const float alpha = alpha_dev ? *alpha_dev : 1.f;
  const int64_t nvec = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(x + i * 4);
    f32x4_t o = {0.f, 0.f, 0.f, 0.f};
    if (accumulate) o = *reinterpret_cast<const f32x4_t*>(out + i * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = __builtin_fmaf(alpha, a[j], o[j]);
    *reinterpret_cast<f32x4_t*>(out + i * 4) = o;
  }
  const int64_t t = (nvec << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) out[t] = __builtin_fmaf(alpha, x[t], accumulate ? out[t] : 0.f);
}

static int plm_stream_grid(int64_t nvec) {
  const int64_t work = plm_cdiv(nvec, 256);
  return (int)(work < ((int64_t)1 << 20) ? (work < 1 ? 1 : work) : ((int64_t)1 << 20));
}

extern "C" int plm_scale_bf16(uint16_t* x, int64_t n, const float* alpha_dev, void* stream) {
  PLM_REQUIRE(x && alpha_dev && n >= 0, "plm_scale_bf16: null pointer or negative n");
  PLM_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "plm_scale_bf16: x must be 16-byte aligned");
  if (n == 0) return PLM_OK;
  hipLaunchKernelGGL(scale_bf16_kernel, dim3(plm_stream_grid(plm_cdiv(n, 8))), dim3(256), 0, (hipStream_t)stream, x, n, alpha_dev);
  PLM_CHECK_LAUNCH("plm_scale_bf16");
  return PLM_OK;
}

extern "C" int plm_axpy_f32(float* out, const float* x, int64_t n, const float* alpha_dev, int accumulate, void* stream) {
  PLM_REQUIRE(out && x && n >= 0, "plm_axpy_f32: null pointer or negative n");
  PLM_REQUIRE(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(x)) & 15) == 0, "plm_axpy_f32: pointers must be 16-byte aligned");
  if (n == 0) return PLM_OK;
  hipLaunchKernelGGL(axpy_f32_kernel, dim3(plm_stream_grid(plm_cdiv(n, 4))), dim3(256), 0, (hipStream_t)stream, out, x, n, alpha_dev,
                     accumulate);
  PLM_CHECK_LAUNCH("plm_axpy_f32");
  return PLM_OK;
}

extern "C" int plm_cast_f32_bf16_t(const float* src, uint16_t* dst, uint16_t* dst_t, int64_t rows, int64_t cols,
                                   int64_t ld_t, void* stream) {
  PLM_REQUIRE(src && dst && dst_t, "plm_cast_f32_bf16_t: null pointer");
  PLM_REQUIRE(rows > 0 && cols > 0 && rows % 8 == 0 && cols % 8 == 0, "plm_cast_f32_bf16_t: rows=%ld cols=%ld must be positive multiples of 8",
              (long)rows, (long)cols);
  PLM_REQUIRE(ld_t >= rows && ld_t % 8 == 0, "plm_cast_f32_bf16_t: ld_t=%ld must be >= rows and a multiple of 8", (long)ld_t);
  dim3 grid((unsigned)plm_cdiv(cols, 64), (unsigned)plm_cdiv(rows, 64));
  hipLaunchKernelGGL(cast_f32_bf16_t_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, dst_t, rows, cols, ld_t);
  PLM_CHECK_LAUNCH("plm_cast_f32_bf16_t");
  return PLM_OK;
}

// ===========================================================================
// embedding  (models/transformer.py:94,110)
// ===========================================================================
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ W,
                                                        float* __restrict__ out, int64_t M, int d, int64_t V) {
  const int lane = threadIdx.x & 63;
  const int64_t row_ = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row_ >= M) return;
  const int64_t row = PLM_REV_ROW(row_, M);
  int64_t id = ids[row];
  id = id < 0 ? 0 : (id >= V ? V - 1 : id);  // host validates ids; clamp keeps the kernel memory-safe
  const float* src = W + id * d;
  float* dst = out + row * d;
  for (int c = lane * 4; c < d; c += 256) *reinterpret_cast<f32x4_t*>(dst + c) = *reinterpret_cast<const f32x4_t*>(src + c);
}

__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dout,
                                                        float* __restrict__ dW, int64_t M, int d, int64_t V) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int64_t id = ids[row];
  if (id < 0 || id >= V) return;
  const float* src = dout + row * d;
  float* dst = dW + id * d;
  for (int c = lane * 4; c < d; c += 256) {
    const f32x4_t v = *reinterpret_cast<const f32x4_t*>(src + c);
    unsafeAtomicAdd(dst + c + 0, v[0]);
    unsafeAtomicAdd(dst + c + 1, v[1]);
    unsafeAtomicAdd(dst + c + 2, v[2]);
    unsafeAtomicAdd(dst + c + 3, v[3]);
  }
}

// ---------------------------------------------------------------------------------------------
// Deterministic embedding backward (no atomics on the gradient): a stable LSD radix sort of the token ids (two passes of
// 8-bit digits over 1024-token blocks: per-block histogram -> one-block scan -> stable scatter; M <= 65536 tokens and
// V < 65536 so that (id, token index) packs into 32 bits), segment bounds per vocabulary row, then one wave per row adds
// its tokens' gradient rows in increasing token order and writes the row ONCE (zeros for rows without tokens, so the
// caller does not have to clear dW first).  The sort moves 128 KiB; its cost is the ten dependent launches (~45 us).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void embed_keys_kernel(const int64_t* __restrict__ ids, unsigned* __restrict__ buf, int M, int V) {
  const int e = blockIdx.x * 1024 + threadIdx.x;
  if (e >= M) return;
  const int64_t id = ids[e];
  const unsigned key = (id < 0 || id >= V) ? (unsigned)V : (unsigned)id;  // out-of-range ids sort behind every row and are ignored
  buf[e] = (key << 16) | (unsigned)e;
}

// hist[digit * nb + block] = number of elements of `block` with that digit
__global__ __launch_bounds__(1024) void embed_hist_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ hist, int M, int shift) {
  __shared__ unsigned cnt[256];
  const int t = threadIdx.x, e = blockIdx.x * 1024 + t;
  if (t < 256) cnt[t] = 0;
  __syncthreads();
  if (e < M) atomicAdd(&cnt[(src[e] >> shift) & 255], 1u);  // integer counts: order-independent
  __syncthreads();
  if (t < 256) hist[t * gridDim.x + blockIdx.x] = cnt[t];
}

// in-place exclusive scan of n <= 16384 counters (digit-major, block-minor = the order of a stable sort)
__global__ __launch_bounds__(1024) void embed_scan_kernel(unsigned* __restrict__ hist, int n) {
  __shared__ unsigned wsum[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  unsigned loc[16], sum = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    loc[j] = (t * 16 + j < n) ? hist[t * 16 + j] : 0u;
    sum += loc[j];
  }
  unsigned incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  unsigned base = incl - sum;
  for (int w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (t * 16 + j < n) hist[t * 16 + j] = base;
    base += loc[j];
  }
}

// stable scatter: position = scanned base of (digit, block) + elements of the same digit earlier in the block
__global__ __launch_bounds__(1024) void embed_scatter_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ dst,
                                                             const unsigned* __restrict__ base, int M, int shift) {
  __shared__ unsigned wcnt[16 * 256];  // [wave][digit]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, e = blockIdx.x * 1024 + t;
#pragma unroll
  for (int j = 0; j < 4; ++j) wcnt[t * 4 + j] = 0;
  const bool valid = e < M;
  const unsigned v = valid ? src[e] : 0u;
  const unsigned d = (v >> shift) & 255;
  unsigned long long mask = __ballot(valid);  // lanes of this wave holding the same digit
#pragma unroll
  for (int bit = 0; bit < 8; ++bit) {
    const bool on = (d >> bit) & 1;
    const unsigned long long bal = __ballot(on);
    mask &= on ? bal : ~bal;
  }
  const unsigned rank = (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
  __syncthreads();
  if (valid && rank == 0) wcnt[wave * 256 + d] = (unsigned)__popcll(mask);
  __syncthreads();
  if (t < 256) {
    unsigned run = base[t * gridDim.x + blockIdx.x];
    for (int w = 0; w < 16; ++w) {
      const unsigned c = wcnt[w * 256 + t];
      wcnt[w * 256 + t] = run;
      run += c;
    }
  }
  __syncthreads();
  if (valid) dst[wcnt[wave * 256 + d] + rank] = v;
}

__global__ __launch_bounds__(1024) void embed_bounds_kernel(const unsigned* __restrict__ sorted, int* __restrict__ seg_lo,
                                                            int* __restrict__ seg_hi, int M) {
  const int e = blockIdx.x * 1024 + threadIdx.x;
  if (e >= M) return;
  const unsigned k = sorted[e] >> 16;
  if (e == 0 || (sorted[e - 1] >> 16) != k) seg_lo[k] = e;
  if (e == M - 1 || (sorted[e + 1] >> 16) != k) seg_hi[k] = e + 1;
}

__global__ __launch_bounds__(256) void embed_bwd_sorted_kernel(const unsigned* __restrict__ sorted, const int* __restrict__ seg_lo,
                                                               const int* __restrict__ seg_hi, const float* __restrict__ dout,
                                                               float* __restrict__ dW, int d, int V, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (v >= V) return;
  const int lo = seg_lo[v], hi = seg_hi[v];
  if (hi <= lo && accumulate) return;  // nothing to add
  float* dst = dW + (int64_t)v * d;
  for (int c = lane * 4; c < d; c += 256) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    if (accumulate) acc = *reinterpret_cast<const f32x4_t*>(dst + c);
    for (int p = lo; p < hi; ++p) {
      const int tok = (int)(sorted[p] & 0xffffu);
      acc += *reinterpret_cast<const f32x4_t*>(dout + (int64_t)tok * d + c);
    }
    *reinterpret_cast<f32x4_t*>(dst + c) = acc;
  }
}

extern "C" size_t plm_embed_bwd_workspace_bytes(int64_t M, int64_t V) {
  if (M <= 0 || V <= 0 || M > 65536 || V >= 65536) return 0;  // 0: use plm_embed_bwd (atomic scatter-add)
  const int64_t nb = plm_cdiv(M, 1024);
  return (size_t)(2 * M + 2 * (V + 1) + 256 * nb) * 4;
}

extern "C" int plm_embed_bwd_sorted(const int64_t* ids, const float* dout, float* dW, int64_t M, int64_t d, int64_t V, int accumulate,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  PLM_REQUIRE(ids && dout && dW && workspace, "plm_embed_bwd_sorted: null pointer");
  PLM_REQUIRE(M > 0 && d > 0 && d % 4 == 0 && V > 0, "plm_embed_bwd_sorted: bad shape M=%ld d=%ld V=%ld", (long)M, (long)d, (long)V);
  const size_t need = plm_embed_bwd_workspace_bytes(M, V);
  PLM_REQUIRE(need != 0, "plm_embed_bwd_sorted: needs M <= 65536 and V < 65536 (M=%ld V=%ld); use plm_embed_bwd", (long)M, (long)V);
  PLM_REQUIRE(workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
              "plm_embed_bwd_sorted: workspace of %zu bytes (16-byte aligned) required, %zu given", need, workspace_bytes);
  hipStream_t s = (hipStream_t)stream;
  const int nb = (int)plm_cdiv(M, 1024);
  unsigned* buf0 = (unsigned*)workspace;
  unsigned* buf1 = buf0 + M;
  int* seg_lo = (int*)(buf1 + M);
  int* seg_hi = seg_lo + (V + 1);
  unsigned* hist = (unsigned*)(seg_hi + (V + 1));
  if (hipMemsetAsync(seg_lo, 0, (size_t)2 * (V + 1) * 4, s) != hipSuccess) {
    plm_set_error("plm_embed_bwd_sorted: hipMemsetAsync failed");
    return PLM_E_HIP;
  }
  const dim3 grid((unsigned)nb), block(1024);
  hipLaunchKernelGGL(embed_keys_kernel, grid, block, 0, s, ids, buf0, (int)M, (int)V);
  unsigned* src = buf0;
  unsigned* dst = buf1;
  const int passes = V < 256 ? 1 : 2;  // keys are 0..V (V = the out-of-range sentinel)
  for (int pass = 0; pass < passes; ++pass) {
    const int shift = 16 + 8 * pass;
    hipLaunchKernelGGL(embed_hist_kernel, grid, block, 0, s, src, hist, (int)M, shift);
    hipLaunchKernelGGL(embed_scan_kernel, dim3(1), block, 0, s, hist, 256 * nb);
    hipLaunchKernelGGL(embed_scatter_kernel, grid, block, 0, s, src, dst, hist, (int)M, shift);
    unsigned* tmp = src;
    src = dst;
    dst = tmp;
  }
  hipLaunchKernelGGL(embed_bounds_kernel, grid, block, 0, s, src, seg_lo, seg_hi, (int)M);
  hipLaunchKernelGGL(embed_bwd_sorted_kernel, dim3((unsigned)plm_cdiv(V, 4)), dim3(256), 0, s, src, seg_lo, seg_hi, dout, dW, (int)d, (int)V,
                     accumulate);
  PLM_CHECK_LAUNCH("plm_embed_bwd_sorted");
  return PLM_OK;
}

extern "C" int plm_embed_fwd(const int64_t* ids, const float* W, float* out, int64_t M, int64_t d, int64_t V, void* stream) {
  PLM_REQUIRE(ids && W && out, "plm_embed_fwd: null pointer");
  PLM_REQUIRE(M > 0 && d > 0 && d % 4 == 0 && V > 0, "plm_embed_fwd: bad shape M=%ld d=%ld V=%ld", (long)M, (long)d, (long)V);
  hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)plm_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, ids, W, out, M, (int)d, V);
  PLM_CHECK_LAUNCH("plm_embed_fwd");
  return PLM_OK;
}

extern "C" int plm_embed_bwd(const int64_t* ids, const float* dout, float* dW, int64_t M, int64_t d, int64_t V, void* stream) {
  PLM_REQUIRE(ids && dout && dW, "plm_embed_bwd: null pointer");
  PLM_REQUIRE(M > 0 && d > 0 && d % 4 == 0 && V > 0, "plm_embed_bwd: bad shape M=%ld d=%ld V=%ld", (long)M, (long)d, (long)V);
  hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)plm_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, ids, dout, dW, M, (int)d, V);
  PLM_CHECK_LAUNCH("plm_embed_bwd");
  return PLM_OK;
}

// ===========================================================================
// RMSNorm (+ residual add)   (models/components.py:16-28, transformer.py:81-82)
// One wave per row; the row lives in registers (NCH float4 per lane, d <= 256*NCH).
// ===========================================================================
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const float* __restrict__ x, const uint16_t* __restrict__ branch,
                                                          float* __restrict__ xout, const float* __restrict__ w,
                                                          uint16_t* __restrict__ y, float* __restrict__ rstd, int64_t M,
                                                          int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row_ = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row_ >= M) return;
  const int64_t row = PLM_REV_ROW(row_, M);
  const int nvec = d >> 2;
  const float* xr = x + row * d;
  f32x4_t v[NCH];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    v[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (c < nvec) {
      v[i] = *reinterpret_cast<const f32x4_t*>(xr + c * 4);
      if (branch) {
        const bf16x4_t b = ld_bf16x4(branch + row * d + c * 4);
        v[i][0] += bf2f(b[0]); v[i][1] += bf2f(b[1]); v[i][2] += bf2f(b[2]); v[i][3] += bf2f(b[3]);
      }
      if (xout) *reinterpret_cast<f32x4_t*>(xout + row * d + c * 4) = v[i];
      ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    }
  }
  ss = wave_sum(ss);
  const float r = rsqrtf(ss / (float)d + eps);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nvec) {
      const f32x4_t wv = *reinterpret_cast<const f32x4_t*>(w + c * 4);
      bf16x4_t o;
      o[0] = f2bf((v[i][0] * r) * wv[0]);
      o[1] = f2bf((v[i][1] * r) * wv[1]);
      o[2] = f2bf((v[i][2] * r) * wv[2]);
      o[3] = f2bf((v[i][3] * r) * wv[3]);
      st_bf16x4(y + row * d + c * 4, o);
    }
  }
  if (lane == 0) rstd[row] = r;
}

#define PLM_RMS_BWD_MAX_BLOCKS 1024

template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const uint16_t* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ w, const float* __restrict__ rstd,
                                                          const float* __restrict__ gin, float* __restrict__ dx,
                                                          uint16_t* __restrict__ dx_bf16, float* __restrict__ dw_partial,
                                                          int64_t M, int d) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = d >> 2;
  const float inv_d = 1.f / (float)d;
  f32x4_t wv[NCH], dwacc[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    wv[i] = (c < nvec) ? *reinterpret_cast<const f32x4_t*>(w + c * 4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    dwacc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  for (int64_t row_ = (int64_t)blockIdx.x * 4 + wave; row_ < M; row_ += (int64_t)gridDim.x * 4) {
    const int64_t row = PLM_REV_ROW(row_, M);
    f32x4_t a[NCH], xv[NCH];
    float dot = 0.f;
    const float r = rstd[row];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      a[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      xv[i] = a[i];
      if (c < nvec) {
        const bf16x4_t g = ld_bf16x4(dy + row * d + c * 4);
        xv[i] = *reinterpret_cast<const f32x4_t*>(x + row * d + c * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float gf = bf2f(g[e]);
          dwacc[i][e] += gf * (xv[i][e] * r);
          a[i][e] = gf * wv[i][e];
          dot += a[i][e] * xv[i][e];
        }
      }
    }
    dot = wave_sum(dot);
    const float coef = dot * r * r * r * inv_d;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        f32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = r * a[i][e] - xv[i][e] * coef;
        if (gin) {
          const f32x4_t gi = *reinterpret_cast<const f32x4_t*>(gin + row * d + c * 4);
          o += gi;
        }
        *reinterpret_cast<f32x4_t*>(dx + row * d + c * 4) = o;
        if (dx_bf16) {
          bf16x4_t ob;
          ob[0] = f2bf(o[0]); ob[1] = f2bf(o[1]); ob[2] = f2bf(o[2]); ob[3] = f2bf(o[3]);
          st_bf16x4(dx_bf16 + row * d + c * 4, ob);
        }
      }
    }
  }
  // block-level reduction of the 4 waves' dw partials
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nvec) *reinterpret_cast<f32x4_t*>(red + wave * d + c * 4) = dwacc[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += 256) {
    dw_partial[(int64_t)blockIdx.x * d + c] = red[c] + red[d + c] + red[2 * d + c] + red[3 * d + c];
  }
}

extern "C" int64_t plm_rmsnorm_bwd_blocks(int64_t M) {
  const int64_t b = plm_cdiv(M, 4);
  return b < PLM_RMS_BWD_MAX_BLOCKS ? b : PLM_RMS_BWD_MAX_BLOCKS;
}

extern "C" int plm_rmsnorm_fwd(const float* x, const uint16_t* branch, float* xout, const float* w, uint16_t* y, float* rstd,
                               int64_t M, int64_t d, float eps, void* stream) {
  PLM_REQUIRE(x && w && y && rstd, "plm_rmsnorm_fwd: null pointer");
  PLM_REQUIRE(M > 0 && d > 0 && d % 4 == 0 && d <= 2048, "plm_rmsnorm_fwd: unsupported shape M=%ld d=%ld (d %% 4 == 0, d <= 2048)", (long)M, (long)d);
  const dim3 grid((unsigned)plm_cdiv(M, 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (d <= 256) hipLaunchKernelGGL(rmsnorm_fwd_kernel<1>, grid, block, 0, s, x, branch, xout, w, y, rstd, M, (int)d, eps);
  else if (d <= 768) hipLaunchKernelGGL(rmsnorm_fwd_kernel<3>, grid, block, 0, s, x, branch, xout, w, y, rstd, M, (int)d, eps);
  else if (d <= 1024) hipLaunchKernelGGL(rmsnorm_fwd_kernel<4>, grid, block, 0, s, x, branch, xout, w, y, rstd, M, (int)d, eps);
  else hipLaunchKernelGGL(rmsnorm_fwd_kernel<8>, grid, block, 0, s, x, branch, xout, w, y, rstd, M, (int)d, eps);
  PLM_CHECK_LAUNCH("plm_rmsnorm_fwd");
  return PLM_OK;
}

extern "C" int plm_rmsnorm_bwd(const uint16_t* dy, const float* x, const float* w, const float* rstd, const float* gin,
                               float* dx, uint16_t* dx_bf16, float* dw_partial, int64_t M, int64_t d, void* stream) {
  PLM_REQUIRE(dy && x && w && rstd && dx && dw_partial, "plm_rmsnorm_bwd: null pointer");
  PLM_REQUIRE(M > 0 && d > 0 && d % 4 == 0 && d <= 2048, "plm_rmsnorm_bwd: unsupported shape M=%ld d=%ld", (long)M, (long)d);
  const dim3 grid((unsigned)plm_rmsnorm_bwd_blocks(M)), block(256);
  const size_t smem = (size_t)4 * d * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  if (d <= 256) hipLaunchKernelGGL(rmsnorm_bwd_kernel<1>, grid, block, smem, s, dy, x, w, rstd, gin, dx, dx_bf16, dw_partial, M, (int)d);
  else if (d <= 768) hipLaunchKernelGGL(rmsnorm_bwd_kernel<3>, grid, block, smem, s, dy, x, w, rstd, gin, dx, dx_bf16, dw_partial, M, (int)d);
  else if (d <= 1024) hipLaunchKernelGGL(rmsnorm_bwd_kernel<4>, grid, block, smem, s, dy, x, w, rstd, gin, dx, dx_bf16, dw_partial, M, (int)d);
  else hipLaunchKernelGGL(rmsnorm_bwd_kernel<8>, grid, block, smem, s, dy, x, w, rstd, gin, dx, dx_bf16, dw_partial, M, (int)d);
  PLM_CHECK_LAUNCH("plm_rmsnorm_bwd");
  return PLM_OK;
}

// out[j] (+)= sum_r part[r][j]: one 1024-thread block per 64 columns; thread (g, c) sums rows g, g+16, ... with four
// independent accumulators (loads stay in flight), then the 16 row groups are combined in LDS in a fixed order.
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ part, float* __restrict__ out, int64_t rows,
                                                      int64_t cols, int accumulate) {
  __shared__ float red[16][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * 64 + c;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < cols) {
    int64_t r = g;
    for (; r + 48 < rows; r += 64) {
      s0 += part[r * cols + col];
      s1 += part[(r + 16) * cols + col];
      s2 += part[(r + 32) * cols + col];
      s3 += part[(r + 48) * cols + col];
    }
    for (; r < rows; r += 16) s0 += part[r * cols + col];
  }
  red[g][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && col < cols) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[i][c];
    out[col] = accumulate ? out[col] + s : s;
  }
}

#define PLM_COLSUM_MAX 64
struct ColsumGroup {
  const float* part[PLM_COLSUM_MAX];
  float* out[PLM_COLSUM_MAX];
  unsigned long long accumulate_mask;
};
// blockIdx.y = item: the single-item kernel's body on that item's buffers
__global__ __launch_bounds__(1024) void colsum_multi_kernel(ColsumGroup g, int64_t rows, int64_t cols) {
  __shared__ float red[16][64];
  const float* __restrict__ part = g.part[blockIdx.y];
  float* __restrict__ out = g.out[blockIdx.y];
  const int accumulate = (int)((g.accumulate_mask >> blockIdx.y) & 1ull);
  const int c = threadIdx.x & 63, gi = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * 64 + c;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < cols) {
    int64_t r = gi;
    for (; r + 48 < rows; r += 64) {
      s0 += part[r * cols + col];
      s1 += part[(r + 16) * cols + col];
      s2 += part[(r + 32) * cols + col];
      s3 += part[(r + 48) * cols + col];
    }
    for (; r < rows; r += 16) s0 += part[r * cols + col];
  }
  red[gi][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (gi == 0 && col < cols) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[i][c];
    out[col] = accumulate ? out[col] + s : s;
  }
}

extern "C" int plm_colsum_f32_multi(const plm_colsum_item* items, int count, int64_t rows, int64_t cols, void* stream) {
  PLM_REQUIRE(items && count >= 1 && rows > 0 && cols > 0, "plm_colsum_f32_multi: bad arguments");
  for (int first = 0; first < count; first += PLM_COLSUM_MAX) {
    const int n = count - first < PLM_COLSUM_MAX ? count - first : PLM_COLSUM_MAX;
    ColsumGroup g;
    g.accumulate_mask = 0ull;
    for (int i = 0; i < n; ++i) {
      PLM_REQUIRE(items[first + i].part && items[first + i].out, "plm_colsum_f32_multi: null pointer in item %d", first + i);
      g.part[i] = items[first + i].part;
      g.out[i] = items[first + i].out;
      if (items[first + i].accumulate) g.accumulate_mask |= 1ull << i;
    }
    hipLaunchKernelGGL(colsum_multi_kernel, dim3((unsigned)plm_cdiv(cols, 64), (unsigned)n), dim3(1024), 0, (hipStream_t)stream, g, rows, cols);
    PLM_CHECK_LAUNCH("plm_colsum_f32_multi");
  }
  return PLM_OK;
}

extern "C" int plm_colsum_f32(const float* part, float* out, int64_t rows, int64_t cols, int accumulate, void* stream) {
  PLM_REQUIRE(part && out && rows > 0 && cols > 0, "plm_colsum_f32: bad arguments");
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)plm_cdiv(cols, 64)), dim3(1024), 0, (hipStream_t)stream, part, out, rows, cols, accumulate);
  PLM_CHECK_LAUNCH("plm_colsum_f32");
  return PLM_OK;
}

// ===========================================================================
// SwiGLU gate   (models/components.py:55-56) — bf16 autograd chain of the reference:
//   s = bf16(silu(x)) ; out = bf16(s * z)
//   ds = bf16(dout * z) ; dz = bf16(dout * s) ; dx = bf16(ds * sig * (1 + x * (1 - sig)))
// ===========================================================================
__device__ __forceinline__ float sigmoidf_(float x) { return plm_sigmoid(x); }  // plm_device.h (shared with the fc1 GEMM epilogue)

// One 16-byte item per thread; a block covers 256 consecutive items of ONE row (grid = M x blocks-per-row), so the row / column split is a
// wave-uniform 32-bit division on the block index instead of a 64-bit division per thread (~45 % of the instructions of the previous form;
// the kernels were and are HBM-bound at 5.8 TB/s of algorithmic bytes, so the timing did not move: 69 / 116 us at M = 32768, h = 2048).
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const uint16_t* __restrict__ u, uint16_t* __restrict__ out, int64_t M,
                                                         int64_t h, unsigned bpr) {
  const unsigned bid = (unsigned)PLM_REV_BLOCK();
  const unsigned mrow = bid / bpr;
  const int64_t m = mrow, c = (int64_t)((bid - mrow * bpr) * 256u + threadIdx.x) * 8;
  if (c >= h) return;
  const bf16x8_t xv = ld_bf16x8(u + m * 2 * h + c);
  const bf16x8_t zv = ld_bf16x8(u + m * 2 * h + h + c);
  bf16x8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = plm_swiglu_bf16(xv[e], zv[e]);
  st_bf16x8(out + m * h + c, o);
}

__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const uint16_t* __restrict__ dout, const uint16_t* __restrict__ u,
                                                         uint16_t* __restrict__ du, int64_t M, int64_t h, unsigned bpr) {
  const unsigned bid = (unsigned)PLM_REV_BLOCK();
  const unsigned mrow = bid / bpr;
  const int64_t m = mrow, c = (int64_t)((bid - mrow * bpr) * 256u + threadIdx.x) * 8;
  if (c >= h) return;
  const bf16x8_t xv = ld_bf16x8(u + m * 2 * h + c);
  const bf16x8_t zv = ld_bf16x8(u + m * 2 * h + h + c);
  const bf16x8_t gv = ld_bf16x8(dout + m * h + c);
  bf16x8_t dxo, dzo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float xf = bf2f(xv[e]), zf = bf2f(zv[e]), gf = bf2f(gv[e]);
    const float sig = sigmoidf_(xf);
    const bf16_t s = f2bf(xf * sig);
    const bf16_t ds = f2bf(gf * zf);
    dzo[e] = f2bf(gf * bf2f(s));
    dxo[e] = f2bf(bf2f(ds) * (sig * (1.f + xf * (1.f - sig))));
  }
  st_bf16x8(du + m * 2 * h + c, dxo);
  st_bf16x8(du + m * 2 * h + h + c, dzo);
}

// The activations of the reference's two plain MLPs (models/components.py:31-40 `MLP`: fc2(silu(fc1 x)); :59-70 `MLPReluSquared`:
// fc2(relu(fc1 x)^2)), forward and backward, bf16 in / out with fp32 math and the autocast rounding order (every torch op rounds its bf16
// result: relu(x)^2 is bf16(r * r) of the bf16 r).  kind 0 = silu, 1 = relu squared.  Same launch shape as the SwiGLU kernels above.
template <int KIND>
__global__ __launch_bounds__(256) void act_fwd_kernel(const uint16_t* __restrict__ u, uint16_t* __restrict__ out, int64_t n8) {
  const int64_t i = PLM_REV_BLOCK() * 256 + threadIdx.x;
  if (i >= n8) return;
  const bf16x8_t xv = ld_bf16x8(u + i * 8);
  bf16x8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float xf = bf2f(xv[e]);
    if (KIND == 0) o[e] = f2bf(xf * plm_sigmoid(xf));
    else {
      const float r = fmaxf(xf, 0.f);
      o[e] = f2bf(r * r);
    }
  }
  st_bf16x8(out + i * 8, o);
}
template <int KIND>
__global__ __launch_bounds__(256) void act_bwd_kernel(const uint16_t* __restrict__ dout, const uint16_t* __restrict__ u,
                                                      uint16_t* __restrict__ du, int64_t n8) {
  const int64_t i = PLM_REV_BLOCK() * 256 + threadIdx.x;
  if (i >= n8) return;
  const bf16x8_t xv = ld_bf16x8(u + i * 8), gv = ld_bf16x8(dout + i * 8);
  bf16x8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float xf = bf2f(xv[e]), gf = bf2f(gv[e]);
    if (KIND == 0) {
      const float sig = plm_sigmoid(xf);
      o[e] = f2bf(gf * (sig * (1.f + xf * (1.f - sig))));
    } else {
      o[e] = f2bf(bf2f(f2bf(gf * 2.f * fmaxf(xf, 0.f))));  // pow's backward (bf16), then relu's mask (already zero where x <= 0)
    }
  }
  st_bf16x8(du + i * 8, o);
}

extern "C" int plm_act_fwd(const uint16_t* u, uint16_t* out, int64_t n, int kind, void* stream) {
  PLM_REQUIRE(u && out && n > 0 && n % 8 == 0 && (kind == 0 || kind == 1), "plm_act_fwd: bad arguments (n %% 8 == 0, kind 0 = silu | 1 = relu^2)");
  PLM_REQUIRE(((reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, "plm_act_fwd: pointers must be 16-byte aligned");
  const int64_t n8 = n / 8, blocks = plm_cdiv(n8, 256);
  PLM_REQUIRE(blocks < ((int64_t)1 << 31), "plm_act_fwd: too large for one launch");
  if (kind == 0) hipLaunchKernelGGL(act_fwd_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, u, out, n8);
  else hipLaunchKernelGGL(act_fwd_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, u, out, n8);
  PLM_CHECK_LAUNCH("plm_act_fwd");
  return PLM_OK;
}
extern "C" int plm_act_bwd(const uint16_t* dout, const uint16_t* u, uint16_t* du, int64_t n, int kind, void* stream) {
  PLM_REQUIRE(dout && u && du && n > 0 && n % 8 == 0 && (kind == 0 || kind == 1), "plm_act_bwd: bad arguments (n %% 8 == 0, kind 0 = silu | 1 = relu^2)");
  PLM_REQUIRE(((reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(du)) & 15) == 0,
              "plm_act_bwd: pointers must be 16-byte aligned");
  const int64_t n8 = n / 8, blocks = plm_cdiv(n8, 256);
  PLM_REQUIRE(blocks < ((int64_t)1 << 31), "plm_act_bwd: too large for one launch");
  if (kind == 0) hipLaunchKernelGGL(act_bwd_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dout, u, du, n8);
  else hipLaunchKernelGGL(act_bwd_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dout, u, du, n8);
  PLM_CHECK_LAUNCH("plm_act_bwd");
  return PLM_OK;
}

// One item per thread, blocks in memory order: a grid capped at 8192 blocks with a grid-stride loop (every resident block
// a stride apart) measured 4-5 % slower on the SwiGLU kernels and 29 % slower on AdamW (0.95 -> 0.74 ms for 162M parameters)
// than letting the dispatcher walk memory linearly (run 32); the stride loop only remains for > 2^20 blocks.
static int elementwise_grid(int64_t items) {
  const int64_t cap = (int64_t)1 << 20;
  const int64_t b = plm_cdiv(items, 256);
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

extern "C" int plm_swiglu_fwd(const uint16_t* u, uint16_t* out, int64_t M, int64_t h, void* stream) {
  PLM_REQUIRE(u && out && M > 0 && h > 0 && h % 8 == 0, "plm_swiglu_fwd: bad arguments (h %% 8 == 0)");
  const int64_t bpr = plm_cdiv(h / 8, 256);
  PLM_REQUIRE(M * bpr < ((int64_t)1 << 31), "plm_swiglu_fwd: M x h too large for one launch");
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3((unsigned)(M * bpr)), dim3(256), 0, (hipStream_t)stream, u, out, M, h, (unsigned)bpr);
  PLM_CHECK_LAUNCH("plm_swiglu_fwd");
  return PLM_OK;
}

extern "C" int plm_swiglu_bwd(const uint16_t* dout, const uint16_t* u, uint16_t* du, int64_t M, int64_t h, void* stream) {
  PLM_REQUIRE(dout && u && du && M > 0 && h > 0 && h % 8 == 0, "plm_swiglu_bwd: bad arguments (h %% 8 == 0)");
  const int64_t bpr = plm_cdiv(h / 8, 256);
  PLM_REQUIRE(M * bpr < ((int64_t)1 << 31), "plm_swiglu_bwd: M x h too large for one launch");
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3((unsigned)(M * bpr)), dim3(256), 0, (hipStream_t)stream, dout, u, du, M, h, (unsigned)bpr);
  PLM_CHECK_LAUNCH("plm_swiglu_bwd");
  return PLM_OK;
}

// ===========================================================================
// small deterministic reductions
// ===========================================================================
__device__ __forceinline__ float block_sum_1024(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  const int nw = blockDim.x >> 6;
  float t = (threadIdx.x < nw) ? sh[threadIdx.x] : 0.f;
  if (wave == 0) t = wave_sum(t);
  __syncthreads();
  if (threadIdx.x == 0) sh[0] = t;
  __syncthreads();
  const float r = sh[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(1024) void mean_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n) {
  __shared__ float sh[16];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) s += x[i];
  s = block_sum_1024(s, sh);
  if (threadIdx.x == 0) out[0] = s / (float)n;
}

extern "C" int plm_mean_f32(const float* x, float* out, int64_t n, void* stream) {
  PLM_REQUIRE(x && out && n > 0, "plm_mean_f32: bad arguments");
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, out, n);
  PLM_CHECK_LAUNCH("plm_mean_f32");
  return PLM_OK;
}

__global__ __launch_bounds__(1024) void sumsq_stage1_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ scratch) {
  __shared__ float sh[16];
  const int64_t nvec = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * 1024;
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < nvec; i += stride) {
    const f32x4_t v = *reinterpret_cast<const f32x4_t*>(x + i * 4);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0) {
    for (int64_t i = (nvec << 2) + threadIdx.x; i < n; i += 1024) s += x[i] * x[i];
  }
  s = block_sum_1024(s, sh);
  if (threadIdx.x == 0) scratch[blockIdx.x] = s;
}

__global__ __launch_bounds__(1024) void sum_stage2_kernel(const float* __restrict__ scratch, int n, float* __restrict__ out) {
  __shared__ float sh[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) s += scratch[i];
  s = block_sum_1024(s, sh);
  if (threadIdx.x == 0) out[0] = s;
}

extern "C" int plm_sumsq_f32(const float* x, int64_t n, float* scratch, float* out, void* stream) {
  PLM_REQUIRE(x && scratch && out && n > 0, "plm_sumsq_f32: bad arguments");
  PLM_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "plm_sumsq_f32: x must be 16-byte aligned");
  int64_t blocks = plm_cdiv(plm_cdiv(n, 4), 1024);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sumsq_stage1_kernel, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, x, n, scratch);
  hipLaunchKernelGGL(sum_stage2_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, scratch, (int)blocks, out);
  PLM_CHECK_LAUNCH("plm_sumsq_f32");
  return PLM_OK;
}

// ===========================================================================
// AdamW on a flat fp32 span  (torch.optim.AdamW semantics; optim/init_optim.py:14-21)
// ===========================================================================
// One element of torch.optim.AdamW (decoupled decay; bias corrections folded into step / bc2_sqrt), with every rounding spelled out so that
// the flat kernel and the shadow-emitting one produce the same bits whatever the compiler would contract around them.
__device__ __forceinline__ float adamw_elem(float p, float g, float& m, float& v, float cs, float b1, float b2, float eps, float decay, float step,
                                            float bc2_sqrt) {
  const float gi = __fmul_rn(g, cs);
  const float mi = __fmaf_rn(b1, m, __fmul_rn(1.f - b1, gi));
  const float vi = __fmaf_rn(b2, v, __fmul_rn(__fmul_rn(1.f - b2, gi), gi));
  m = mi;
  v = vi;
  const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vi), bc2_sqrt), eps);
  return __fmaf_rn(-step, __fdiv_rn(mi, denom), __fmul_rn(p, decay));
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2_sqrt, const float* __restrict__ clip) {
  const float cs = clip ? *clip : 1.f;
  const float step = lr / bc1;
  const float decay = 1.f - lr * wd;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // one element per thread, blocks in memory order (see elementwise_grid: measured 29 % faster than a capped grid with a stride loop;
  // a float4-per-thread form of this kernel measured 0.94 ms against 0.73 ms for 162 M parameters, round 4)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float mi = m[i], vi = v[i];
    p[i] = adamw_elem(p[i], g[i], mi, vi, cs, b1, b2, eps, decay, step, bc2_sqrt);
    m[i] = mi;
    v[i] = vi;
  }
}

// AdamW for a list of Linear weights that also emits the bf16 shadows the next forward / backward consume (W [rows, cols] and
// W^T [cols, ld_t]): the arithmetic of adamw_kernel per element, the tile walk and LDS transposition of
// cast_f32_bf16_t_multi_kernel.  One read of p / g / m / v, one write of p / m / v, plus 4 bytes per parameter of shadows - the
// stand-alone cast (one more read of p, the same shadow writes) disappears from the training step.
#define PLM_ADAMW_MULTI_MAX 56
struct AdamwGroup {
  float* p[PLM_ADAMW_MULTI_MAX];
  const float* g[PLM_ADAMW_MULTI_MAX];
  float* m[PLM_ADAMW_MULTI_MAX];
  float* v[PLM_ADAMW_MULTI_MAX];
  uint16_t* dst[PLM_ADAMW_MULTI_MAX];
  uint16_t* dst_t[PLM_ADAMW_MULTI_MAX];
  int rows[PLM_ADAMW_MULTI_MAX], cols[PLM_ADAMW_MULTI_MAX], ld_t[PLM_ADAMW_MULTI_MAX];
  int block_base[PLM_ADAMW_MULTI_MAX + 1];
  int count;
};

__global__ __launch_bounds__(256) void adamw_cast_multi_kernel(AdamwGroup g, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                               float bc2_sqrt, const float* __restrict__ clip) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];
  int it = 0;
  for (int q = 1; q < g.count; ++q)
    if ((int)blockIdx.x >= g.block_base[q]) it = q;  // block-uniform scalar search
  const int local = blockIdx.x - g.block_base[it];
  const int tiles_x = (g.cols[it] + 63) / 64;
  const int64_t rows = g.rows[it], cols = g.cols[it], ld_t = g.ld_t[it];
  float* __restrict__ P = g.p[it];
  const float* __restrict__ G = g.g[it];
  float* __restrict__ Mm = g.m[it];
  float* __restrict__ V = g.v[it];
  uint16_t* __restrict__ dst = g.dst[it];
  uint16_t* __restrict__ dst_t = g.dst_t[it];
  const int64_t r0 = (int64_t)(local / tiles_x) * 64, c0 = (int64_t)(local % tiles_x) * 64;
  const float cs = clip ? *clip : 1.f;
  const float step = lr / bc1;
  const float decay = 1.f - lr * wd;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (t >> 4) + 16 * i, c = (t & 15) * 4;
    const int64_t gr = r0 + r, gc = c0 + c;
    const bool in = gr < rows && gc < cols;  // cols % 8 == 0: the four columns are in range together
    f32x4_t pn = {0.f, 0.f, 0.f, 0.f};
    if (in) {
      const int64_t o = gr * cols + gc;
      const f32x4_t pv = *reinterpret_cast<const f32x4_t*>(P + o), gv = *reinterpret_cast<const f32x4_t*>(G + o);
      f32x4_t mv = *reinterpret_cast<const f32x4_t*>(Mm + o), vv = *reinterpret_cast<const f32x4_t*>(V + o);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float me = mv[e], ve = vv[e];
        pn[e] = adamw_elem(pv[e], gv[e], me, ve, cs, b1, b2, eps, decay, step, bc2_sqrt);
        mv[e] = me;
        vv[e] = ve;
      }
      *reinterpret_cast<f32x4_t*>(Mm + o) = mv;
      *reinterpret_cast<f32x4_t*>(V + o) = vv;
      *reinterpret_cast<f32x4_t*>(P + o) = pn;
    }
    bf16x4_t o4;
    o4[0] = f2bf(pn[0]); o4[1] = f2bf(pn[1]); o4[2] = f2bf(pn[2]); o4[3] = f2bf(pn[3]);
    if (in) st_bf16x4(dst + gr * cols + gc, o4);
    tile[c + 0][r] = o4[0]; tile[c + 1][r] = o4[1]; tile[c + 2][r] = o4[2]; tile[c + 3][r] = o4[3];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = (t >> 3) + 32 * i, rch = (t & 7) * 8;
    const int64_t gc = c0 + c, gr = r0 + rch;
    if (gc < cols && gr < rows) {
      const bf16x8_t v8 = *reinterpret_cast<const bf16x8_t*>(&tile[c][rch]);
      st_bf16x8(dst_t + gc * ld_t + gr, v8);
    }
  }
}

extern "C" int plm_adamw_cast_multi(const plm_adamw_item* items, int count, float lr, float beta1, float beta2, float eps,
                                    float weight_decay, float bc1, float bc2, const float* clip_coef_dev, void* stream) {
  PLM_REQUIRE(items && count >= 1, "plm_adamw_cast_multi: null pointer or empty list");
  for (int first = 0; first < count; first += PLM_ADAMW_MULTI_MAX) {
    const int n = count - first < PLM_ADAMW_MULTI_MAX ? count - first : PLM_ADAMW_MULTI_MAX;
    AdamwGroup g{};
    int base = 0;
    for (int i = 0; i < n; ++i) {
      const plm_adamw_item& q = items[first + i];
      PLM_REQUIRE(q.p && q.g && q.m && q.v && q.dst && q.dst_t, "plm_adamw_cast_multi: null pointer in item %d", first + i);
      PLM_REQUIRE(q.rows > 0 && q.cols > 0 && q.rows % 8 == 0 && q.cols % 8 == 0 && q.rows < (1ll << 31) && q.cols < (1ll << 31),
                  "plm_adamw_cast_multi: item %d: rows=%ld cols=%ld must be positive multiples of 8", first + i, (long)q.rows, (long)q.cols);
      PLM_REQUIRE(q.ld_t >= q.rows && q.ld_t % 8 == 0 && q.ld_t < (1ll << 31), "plm_adamw_cast_multi: item %d: ld_t=%ld must be >= rows and a multiple of 8",
                  first + i, (long)q.ld_t);
      PLM_REQUIRE(((reinterpret_cast<uintptr_t>(q.p) | reinterpret_cast<uintptr_t>(q.g) | reinterpret_cast<uintptr_t>(q.m) | reinterpret_cast<uintptr_t>(q.v) |
                    reinterpret_cast<uintptr_t>(q.dst) | reinterpret_cast<uintptr_t>(q.dst_t)) & 15) == 0,
                  "plm_adamw_cast_multi: item %d: pointers must be 16-byte aligned", first + i);
      g.p[i] = q.p; g.g[i] = q.g; g.m[i] = q.m; g.v[i] = q.v; g.dst[i] = q.dst; g.dst_t[i] = q.dst_t;
      g.rows[i] = (int)q.rows; g.cols[i] = (int)q.cols; g.ld_t[i] = (int)q.ld_t;
      g.block_base[i] = base;
      const int64_t nb = plm_cdiv(q.rows, 64) * plm_cdiv(q.cols, 64);
      PLM_REQUIRE(base + nb < (1ll << 31), "plm_adamw_cast_multi: too many tiles");
      base += (int)nb;
    }
    g.block_base[n] = base;
    g.count = n;
    hipLaunchKernelGGL(adamw_cast_multi_kernel, dim3((unsigned)base), dim3(256), 0, (hipStream_t)stream, g, lr, beta1, beta2, eps, weight_decay, bc1,
                       sqrtf(bc2), clip_coef_dev);
    PLM_CHECK_LAUNCH("plm_adamw_cast_multi");
  }
  return PLM_OK;
}

extern "C" int plm_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, float bc1, float bc2, const float* clip_coef_dev, void* stream) {
  PLM_REQUIRE(p && g && m && v && n > 0, "plm_adamw_f32: bad arguments");
  hipLaunchKernelGGL(adamw_kernel, dim3(elementwise_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, bc1, sqrtf(bc2), clip_coef_dev);
  PLM_CHECK_LAUNCH("plm_adamw_f32");
  return PLM_OK;
}
