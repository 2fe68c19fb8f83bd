// Attention backward, software-pipelined form: ONE wave per SIMD (256-thread workgroups, up to 512 registers per lane),
// three blocks in flight per wave.
//
// What tools/ubench/overlap.hip measured on MI355X about the two waves of a SIMD (profiles/r02_ubench_overlap.txt):
//   * a wave that streams v_mfma_f32_32x32x16_bf16 runs at the pipe's 32 cycles per instruction, alone or beside anything;
//   * a partner wave doing VALU work beside it issues one instruction per ~12 cycles instead of one per ~3 (64 v_fma:
//     200 cycles alone, 769 beside the MFMA stream), LDS reads 2.5x slower - so "one wave computes while the other does
//     the softmax" (the 8-wave ping-pong form, attn_bwd_dkdv8_kernel) loses: 2234 cycles per 32x32 block against 1877 for
//     the plain 4-wave kernel, whose serial chain  S/dP MFMAs -> softmax -> dV/dK MFMAs  leaves the pipe 27 % busy;
//   * ONE wave that interleaves its own MFMAs with independent VALU work hides most of it: 16 MFMAs + 128 v_fma take 773
//     cycles (512 + 400 issued back to back), and two such waves per SIMD are no faster than one.
// Hence: one wave per SIMD, and inside it the MFMAs of blocks b+1 (S, dP) and b-1 (dV, dK) interleaved with the softmax
// arithmetic of block b and the LDS reads of later blocks - a three-stage software pipeline over 32-query blocks with two
// register sets that swap roles every block (no copies), the matrix accumulators in the accumulator half of the file.
#include "plm_device.h"

#include <stdlib.h>

#include <type_traits>

#include "attn_common.h"

// What one block iteration hands to the next (two sets that swap roles: no register copies).
struct PipeRegs {
  f32x16_t s, dp;          // S, dP of the block whose softmax is due next
  bf16x8_t pf[2], dsf[2];  // P, dS of the block whose dV / dK MFMAs are due next (B operands)
};

// =============================================================================================
// dK, dV: one workgroup per 128 key rows (4 waves x 32 keys), causal, T % 128 == 0.
// Iteration b of a wave (b = 0 .. nb):   B: softmax of block b      C: dV/dK MFMAs of block b-1      A: S/dP MFMAs of block b+1
//   LDS reads at its top: row fragments of block b+1 (A), transposed fragments of block b-1 (C); at its end: lse / delta of
//   block b+1.  B's inputs are in registers when the iteration starts, so its arithmetic covers the latency of those reads.
// The first 4 blocks (the two query tiles on the diagonal of the key block) take the masked softmax.  Blocks outside
// [0, nb) run on whatever finite or not the ring holds and are never consumed (iteration nb only matters for its C stage; P
// and dS start as zeros and block -1 reads tile 0, so C of block -1 adds zeros).  Q / dO tiles of 64 queries go through a
// 6-stage LDS ring: a trip of two iterations touches three tiles, and a tile is staged FOUR trips before its first reader -
// with one workgroup per CU nothing else hides the ~1 us an LDS-DMA takes under load (staged two trips ahead the kernel
// waited for every tile: 269 us against 119 us of arithmetic).  A trip ends with a COUNTED vmcnt that only waits for the
// tile the next trip needs (the two younger tiles stay in flight) and one barrier.
// =============================================================================================
template <int ABL>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkdv_pipe_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout,
                                                                    const float* __restrict__ lse, const float* __restrict__ delta,
                                                                    const float* __restrict__ rcos, const float* __restrict__ rsin,
                                                                    uint16_t* __restrict__ dqkv, int T, int nh) {
  constexpr int QT = 64, PD = 4, NST = PD + 2;  // tiles staged PD trips ahead; a trip touches three tiles
  constexpr int TILE = QT * 128;           // 8 KiB
  constexpr int STAGE = 2 * TILE + 1024;   // Q | dO | lse[64], delta[64]
  __shared__ __attribute__((aligned(1024))) char smem[NST * STAGE];

  const int ntile_k = T / 128;
  const int nbh = gridDim.x / ntile_k;
  const int bh = blockIdx.x % nbh;
  const int kt = blockIdx.x / nbh;  // key tile 0 meets every query tile: heaviest first
  const int h = bh % nh, b = bh / nh;
  const int dm = nh * HD, ld = 3 * dm;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int kv0 = kt * 128;
  const int kvrow = kv0 + wave * 32 + l31;
  const uint16_t* base = qkv + (int64_t)b * T * ld + h * HD;
  const uint16_t* dobase = dout + (int64_t)b * T * dm + h * HD;
  const float* lrow = lse + ((int64_t)b * nh + h) * T;
  const float* drow = delta + ((int64_t)b * nh + h) * T;
  const float scale = 0.125f, c2 = scale * LOG2E;

  bf16x8_t kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const uint16_t* p = base + (int64_t)kvrow * ld + ks * 16 + hi * 8;
    kf[ks] = ld_bf16x8(p + dm);
    vf[ks] = ld_bf16x8(p + 2 * dm);
  }
  const int jq_lo = kv0 / QT;
  const int ntile = T / QT - jq_lo;  // query tiles from the diagonal down
  const int nb = 2 * ntile;
  asm volatile("; k/v fragments resident" ::"v"(kf[0]), "v"(kf[1]), "v"(kf[2]), "v"(kf[3]), "v"(vf[0]), "v"(vf[1]), "v"(vf[2]),
               "v"(vf[3]));  // every ordinary load is consumed before the first DMA is in flight

  f32x16_t dk[2], dv[2];
  zero16(dk[0]); zero16(dk[1]); zero16(dv[0]); zero16(dv[1]);
  const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  TileDma dma, dmad;
  dma.init(wave, lane, ld);
  dmad.init(wave, lane, dm);
  auto stage_tile = [&](int u) {  // query tile u of this workgroup into ring slot u % NST (tiles past the end: nothing)
    if (u >= ntile) return;
    const int qt0 = (jq_lo + u) * QT;
    char* dst = smem + (u % NST) * STAGE;
    dma.issue_full(dst, base + (int64_t)qt0 * ld, wave);
    dmad.issue_full(dst + TILE, dobase + (int64_t)qt0 * dm, wave);
    if (wave < 2 && lane < 16)  // 64 floats = 16 lanes x 16 bytes per statistic: wave 0 brings lse, wave 1 delta
      dma16_asm((wave == 0 ? lrow : drow) + qt0 + lane * 4, dst + 2 * TILE + wave * 256);
  };
  // end of a trip: the tile staged PD - 2 trips ago (needed by the next trip) has landed for this wave; younger ones may fly
  auto wait_tile = [&](bool counted) {
    if (ABL == 4) return;
    if (!counted) attn_wait_vm<0>();
    else if (wave < 2) attn_wait_vm<(PD - 2) * 5>();  // TileDma: 2 + 2 instructions per tile, + 1 statistic
    else attn_wait_vm<(PD - 2) * 4>();
  };
  auto tile_of = [&](int bi) { return smem + ((max(bi, 0) >> 1) % NST) * STAGE; };
  bf16x8_t rq[4], rdo[4];          // row fragments of Q / dO (A stage)
  bf16x8_t tq[2][2], tdo[2][2];    // transposed fragments of Q / dO (C stage)
  f32x4_t L[4], D[4];              // lse (base 2), delta of the block whose softmax is due: 16 queries per lane
  auto load_rows = [&](int bi) {
    const char* sQ = tile_of(bi);
    const int r = (bi & 1) * 32 + l31;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      rq[ks] = frag_rows(sQ, r, ks, hi);
      rdo[ks] = frag_rows(sQ + TILE, r, ks, hi);
    }
  };
  auto load_stats = [&](int bi) {
    const float* sL = reinterpret_cast<const float*>(tile_of(bi) + 2 * TILE);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ql0 = (bi & 1) * 32 + 8 * g + 4 * hi;
      L[g] = *reinterpret_cast<const f32x4_t*>(sL + ql0);
      D[g] = *reinterpret_cast<const f32x4_t*>(sL + 64 + ql0);
    }
  };
  auto load_tr = [&](int bi) {
    const char* sQ = tile_of(bi);
    const int qb = max(bi, 0) & 1;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int rbase = qb * 32 + s2 * 16 + 4 * hi;
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        tdo[s2][db] = frag_cols(sQ + TILE, db, rbase, lane);
        tq[s2][db] = frag_cols(sQ, db, rbase, lane);
      }
    }
  };

  // One block iteration.  P: produced by the previous iteration (consumed here); N: produced here.
  auto iteration = [&](int bi, PipeRegs& P, PipeRegs& N, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    load_tr(bi - 1);
    load_rows(bi + 1);
    const int q0 = (jq_lo + (bi >> 1)) * QT + (bi & 1) * 32;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      // B: softmax of block bi, queries 8g + 4hi .. +3 of the block
      if (ABL != 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          float p = fast_exp2(__builtin_fmaf(P.s[r], c2, -L[g][e]));
          if (MASK) {
            const int qg = q0 + 8 * g + 4 * hi + e;
            p = (kvrow <= qg) ? p : 0.f;
          }
          const float dsv = p * (P.dp[r] - D[g][e]);  // x 1/sqrt(hd) once, on dK, in the epilogue
          N.pf[r >> 3][r & 7] = f2bf(p);
          N.dsf[r >> 3][r & 7] = f2bf(dsv);
        }
      }
      if (ABL != 3) {
        // C: dV^T / dK^T of block bi-1 (two of its eight MFMAs per group)
        const int s2 = g >> 1, db = g & 1;
        dv[db] = mfma32(tdo[s2][db], P.pf[s2], dv[db]);
        dk[db] = mfma32(tq[s2][db], P.dsf[s2], dk[db]);
        // A: S / dP of block bi+1, k-slice g
        N.s = mfma32(rq[g], kf[g], g == 0 ? zero : N.s);
        N.dp = mfma32(rdo[g], vf[g], g == 0 ? zero : N.dp);
      }
    }
    if (ABL == 3) asm volatile("" ::"v"(rq[0]), "v"(rdo[0]), "v"(tq[0][0]), "v"(tdo[1][1]), "v"(P.pf[0]), "v"(P.dsf[1]));
    if (ABL == 2) asm volatile("" ::"v"(P.s), "v"(P.dp), "v"(L[0]), "v"(D[3]));
    load_stats(bi + 1);
  };

  // ---- prologue: tiles 0, 1 land; S / dP and the statistics of block 0 are prepared
  PipeRegs X, Y;
#pragma unroll
  for (int u = 0; u < PD; ++u) stage_tile(u);
  wait_tile(ntile >= PD);  // tiles 0 and 1
  attn_barrier();
  load_rows(0);
  load_stats(0);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    X.s = mfma32(rq[ks], kf[ks], ks == 0 ? zero : X.s);
    X.dp = mfma32(rdo[ks], vf[ks], ks == 0 ? zero : X.dp);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    X.pf[i] = zero_bf16x8();
    X.dsf[i] = zero_bf16x8();
  }
  // ---- main loop: two blocks (one query tile) per trip, X and Y swap roles
  for (int bi = 0; bi <= nb; bi += 2) {
    stage_tile((bi >> 1) + PD);  // ring slot of tile (bi >> 1) - 2: every wave finished reading it before the last barrier
    const bool two = bi + 1 <= nb;  // the last trip only owes the C stage of block nb - 1
    if (bi < 4) {
      iteration(bi, X, Y, std::true_type{});
      if (two) iteration(bi + 1, Y, X, std::true_type{});
    } else {
      iteration(bi, X, Y, std::false_type{});
      if (two) iteration(bi + 1, Y, X, std::false_type{});
    }
    wait_tile((bi >> 1) + PD < ntile);
    attn_barrier();
  }
  attn_wait_vm<0>();

  {
    uint16_t* dkp = dqkv + ((int64_t)b * T + kvrow) * ld + dm + h * HD;
    uint16_t* dvp = dkp + dm;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = db * 32 + 8 * g + 4 * hi;
        bf16x4_t ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = f2bf(dv[db][4 * g + e]);
        st_bf16x4(dvp + d0, ov);
        // inverse rotation of the two (even, odd) pairs of dK: gradient w.r.t. the PRE-rotation k
        const float c0 = rcos[kvrow * 32 + d0 / 2], c1 = rcos[kvrow * 32 + d0 / 2 + 1];
        const float s0 = rsin[kvrow * 32 + d0 / 2], s1 = rsin[kvrow * 32 + d0 / 2 + 1];
        const float a0 = dk[db][4 * g + 0] * scale, b0 = dk[db][4 * g + 1] * scale, a1 = dk[db][4 * g + 2] * scale, b1 = dk[db][4 * g + 3] * scale;
        bf16x4_t ok;
        ok[0] = f2bf(a0 * c0 + b0 * s0);
        ok[1] = f2bf(b0 * c0 - a0 * s0);
        ok[2] = f2bf(a1 * c1 + b1 * s1);
        ok[3] = f2bf(b1 * c1 - a1 * s1);
        st_bf16x4(dkp + d0, ok);
      }
    }
  }
}

// host side: launched by plm_attn_bwd (attn.hip) for causal sequences with T % 128 == 0
void plm_launch_attn_bwd_dkdv_pipe(const uint16_t* qkv, const uint16_t* dout, const float* lse, const float* delta, const float* rope_cos,
                                   const float* rope_sin, uint16_t* dqkv, int64_t B, int64_t T, int64_t nh, hipStream_t s) {
  static const int abl = getenv("PLM_ATTN_ABL") ? atoi(getenv("PLM_ATTN_ABL")) : 0;
  const dim3 grid((unsigned)((T / 128) * nh * B)), block(256);
  if (abl == 4)
    hipLaunchKernelGGL(attn_bwd_dkdv_pipe_kernel<4>, grid, block, 0, s, qkv, dout, lse, delta, rope_cos, rope_sin, dqkv, (int)T, (int)nh);
  else if (abl == 2)
    hipLaunchKernelGGL(attn_bwd_dkdv_pipe_kernel<2>, grid, block, 0, s, qkv, dout, lse, delta, rope_cos, rope_sin, dqkv, (int)T, (int)nh);
  else if (abl == 3)
    hipLaunchKernelGGL(attn_bwd_dkdv_pipe_kernel<3>, grid, block, 0, s, qkv, dout, lse, delta, rope_cos, rope_sin, dqkv, (int)T, (int)nh);
  else
    hipLaunchKernelGGL(attn_bwd_dkdv_pipe_kernel<0>, grid, block, 0, s, qkv, dout, lse, delta, rope_cos, rope_sin, dqkv, (int)T, (int)nh);
}
