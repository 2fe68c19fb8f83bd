// Device helpers shared by the attention kernels (attn.hip, attn_pipe.hip): LDS tile image, fragment reads, LDS-DMA of tiles.
#pragma once
#include "plm_device.h"

#define HD 64
#define LOG2E 1.4426950408889634f

// raw v_exp_f32: arguments here are <= 0 (or -inf), so no denormal-range fix-up is needed
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// ---------------------------------------------------------------------------------------------
// LDS image of a [rows][64 d] bf16 tile: row-major, 128-byte rows (so one LDS-DMA wave-instruction =
// 8 whole rows = 8 fully used 128-byte global segments).  Inside row r (bits b0..b3) the logical 16-byte
// chunk c = 2*pair + half is stored at pair' = (pair + 2*b1 + b3) & 3, half' = half ^ b2:
//   * ds_read_b128 of one chunk across 16 rows (r mod 16 distinct) hits 16 different 16-byte slots of
//     the 256-byte bank row (b0 picks the 128-byte half, (b1,b3) the pair, b2 the half): conflict-free;
//   * ds_read_b64_tr_b16 of a [4 rows][16 cols] block (rows R..R+3, R % 4 == 0) puts the four rows on
//     four different 32-byte segments, and the neighbouring 16-column block on the other four.
// The permutation is applied on the DMA SOURCE address (the DMA itself writes lane-linear).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int rs_off(int row, int c) {
  const int pp = ((c >> 1) + 2 * ((row >> 1) & 1) + ((row >> 3) & 1)) & 3;
  const int ph = (c & 1) ^ ((row >> 2) & 1);
  return row * 128 + pp * 32 + ph * 16;
}
__device__ __forceinline__ int rs_logical_chunk(int row, int pc) {  // inverse: physical chunk pc of row -> logical chunk
  const int cb = ((pc >> 1) - 2 * ((row >> 1) & 1) - ((row >> 3) & 1)) & 3;
  return cb * 2 + ((pc & 1) ^ ((row >> 2) & 1));
}

// A-operand fragment (rows i = tile rows, k = head dims ks*16 + hi*8 ..) by ds_read_b128
__device__ __forceinline__ bf16x8_t frag_rows(const char* tile, int row, int ks, int hi) {
  return *reinterpret_cast<const bf16x8_t*>(tile + rs_off(row, ks * 2 + hi));
}
// A-operand fragment (rows i = head dims db*32 + (lane&31), k = tile rows) by two transpose reads.
// k-slot e of lane-half hi maps to tile row  rbase + (e&3) + 8*(e>>2)  — the row order in which a lane
// holds the matching B operand after a transposed-score MFMA (see mfma32_row()).
__device__ __forceinline__ bf16x8_t frag_cols(const char* tile, int db, int rbase, int lane) {
  const int ib = (lane >> 4) & 1, t16 = lane & 15;
  const int sub = t16 & 3;                 // 8-byte piece of the 32-byte (16-column) block
  const int c = (db * 2 + ib) * 2 + (sub >> 1);
  const int row = rbase + (t16 >> 2);
  const char* p0 = tile + rs_off(row, c) + (sub & 1) * 8;
  const char* p1 = tile + rs_off(row + 8, c) + (sub & 1) * 8;
  return join_tr(lds_read_tr16(p0), lds_read_tr16(p1));
}

// LDS-DMA of a [64 rows][64 d] tile (8 KiB = 8 wave-instructions, two per wave).  `src` points at (row 0, d 0) of
// the tile in global memory, `ld` is the row stride in elements; rows above `last_row` are clamped (their values are
// masked out by the caller).
struct TileDma {
  int row[2], coff[2];
  unsigned boff[2];  // byte offset of this lane's 16 bytes inside a full tile, for the row stride given to init()
  __device__ __forceinline__ void init(int wave, int lane, int64_t ld = 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      row[i] = (i * 4 + wave) * 8 + (lane >> 3);
      coff[i] = rs_logical_chunk(row[i], lane & 7) * 8;
      boff[i] = (unsigned)((row[i] * ld + coff[i]) * 2);
    }
  }
  __device__ __forceinline__ void issue(char* dst_tile, const uint16_t* src, int64_t ld, int last_row, int wave, int row0 = 0) const {
#pragma unroll
    for (int i = 0; i < 2; ++i)  // tile row r is source row row0 + r, clamped to last_row
      dma16_asm(src + (int64_t)min(row0 + row[i], last_row) * ld + coff[i], dst_tile + (i * 4 + wave) * 1024);
  }
  // full tile (no row clamp), row stride = the one given to init(): wave-uniform base in SGPRs + constant lane offsets
  __device__ __forceinline__ void issue_full(char* dst_tile, const uint16_t* src, int wave) const {
#pragma unroll
    for (int i = 0; i < 2; ++i) dma16_saddr_asm(src, boff[i], dst_tile + (i * 4 + wave) * 1024);
  }
};

template <int N>
__device__ __forceinline__ void attn_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void attn_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void zero16(f32x16_t& v) {
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = 0.f;
}

// A wave's 32-row x 64-column bf16 block leaves as whole 128-byte rows: lane (l31, hi) holds row l31 in 8-byte pieces (head dims
// db*32 + 8g + 4hi .. + 3 = piece `hi` of the 16-byte chunk c16 = 4 db + g); the pieces go through a 4 KiB LDS region of the wave's own
// (chunk c16 of row r at position c16 ^ (r & 7): spreads a column of chunks over the banks) and come back as 16 bytes per lane, 8 lanes
// per row - 4 store instructions of 8 full lines each instead of 16 that touch 32 lines each (the per-block cost the timing-only builds
// showed: launch + Q loads + O stores alone were 30 of the forward kernel's 88 us).
struct RowStage {
  char* base;  // 4 KiB, private to the wave, not read by anyone else any more
  int lane;
  __device__ __forceinline__ void put(int l31, int hi, int c16, bf16x4_t v) const {
    *reinterpret_cast<bf16x4_t*>(base + l31 * 128 + ((c16 ^ (l31 & 7)) << 4) + hi * 8) = v;
  }
  // rows row0 .. row0 + 31 of a [*, ld] bf16 matrix at column col0; rows >= row_end are not stored
  __device__ __forceinline__ void flush(uint16_t* dst, int64_t ld, int row0, int row_end, int col0) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 8 * i + (lane >> 3), c16 = lane & 7;
      const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(base + row * 128 + ((c16 ^ (row & 7)) << 4));
      if (row0 + row < row_end) st_bf16x8(dst + (int64_t)(row0 + row) * ld + col0 + c16 * 8, v);
    }
  }
};

