// Device helpers shared by the attention kernels (attn.hip, attn_causal.hip, attn_doc.hip): LDS tile image, fragment reads, LDS-DMA of tiles,
// and the layout of the document-mask plan.
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

// LDS-DMA of a WAVE's own 32 rows x 64 d (4 KiB = 4 wave-instructions of 8 whole 128-byte rows) into a wave-private region with the tile
// image's swizzle, and the fragment read that goes with it (row = lane & 31, chunk = 2 ks + hi).  The row fragments of the document-mask
// kernels' prologues come this way: a fragment load straight from global memory is 32 rows x 32 bytes per instruction - four times the cache-line
// requests for the same bytes, and at the reference's micro-batch all 3072 waves of the grid issue them in the same microsecond.
__device__ __forceinline__ void rows_dma(char* region, const uint16_t* src, int64_t ld, int row0, int last_row, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * i + (lane >> 3);
    dma16_asm(src + (int64_t)min(row0 + row, last_row) * ld + rs_logical_chunk(row, lane & 7) * 8, region + i * 1024);
  }
}
__device__ __forceinline__ bf16x8_t rows_frag(const char* region, int l31, int ks, int hi) {
  return *reinterpret_cast<const bf16x8_t*>(region + rs_off(l31, ks * 2 + hi));
}

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
  // The same with the INVERSE RoPE rotation applied to the staged bf16 values on their way out (dQ, dK: the gradient w.r.t. the
  // un-rotated projection; models/embeddings.py:15-30 differentiated).  After the transposition a lane holds 8 consecutive head dims = 4
  // pairs of one row, so its cos / sin values are ONE 16-byte load per table, 8 lanes cover a table row: 8 fully used lines per
  // instruction.  (Rotating the fp32 accumulators in their MFMA layout instead took 32 four-byte loads per lane that touched 32 lines
  // each: ~3 us of every backward workgroup's life, tools/attn_trace.py.)  The value is rounded to bf16 before the rotation and again after
  // it - the reference's own order (SDPA's backward returns bf16, the rotation's backward runs in fp32 and casts back).
  __device__ __forceinline__ void flush_rot(uint16_t* dst, int64_t ld, int row0, int row_end, int col0, const float* __restrict__ rcos,
                                            const float* __restrict__ rsin) const {
    f32x4_t cs[4], sn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // issued before the LDS round trip is waited for
      const int trow = min(row0 + 8 * i + (lane >> 3), row_end - 1);
      cs[i] = *reinterpret_cast<const f32x4_t*>(rcos + trow * (HD / 2) + (lane & 7) * 4);
      sn[i] = *reinterpret_cast<const f32x4_t*>(rsin + trow * (HD / 2) + (lane & 7) * 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 8 * i + (lane >> 3), c16 = lane & 7;
      const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(base + row * 128 + ((c16 ^ (row & 7)) << 4));
      if (row0 + row < row_end) st_bf16x8(dst + (int64_t)(row0 + row) * ld + col0 + c16 * 8, rope8(v, cs[i], sn[i], -1.f));
    }
  }
};


// ---------------------------------------------------------------------------------------------
// Document-mask plan (plm_attn_doc_plan, attn_doc.hip), int32 units, for a batch of doc_start[B][T] and nh heads:
//   [0, 8)                        header {number of query items, number of key items, 0 ...}
//   [8, 8 + B*T)                  doc_end[b][j]: first query that does NOT see key j (queries j .. doc_end - 1 do; doc_start is non-decreasing)
//   [pq, pq + 4 cap)              query items {b, first row, first key tile (64 keys) of the item's first row, kind << 30 | cost}
//   [pq + 4 cap, pq + 8 cap)      key items   {b, first key, one past the last query tile (64 rows) that sees a key of the item, kind << 30 | cost}
// with pq = 8 + B*T rounded up to a multiple of 4, n = B * ceil(T / 128) tiles and cap = n + n / 4 items per list.
// An item is a 128-row tile (kind 0: four waves x 32 rows) or - when the whole grid is resident at once and the launch therefore lasts as long as
// its longest chain of tile steps - one 64-row half of a HEAVY tile (kind 1: 2 row blocks x 2 halves of every streamed tile, partial results
// combined in LDS; half the chain).  Both lists are sorted by their expected duration, longest first: the hardware hands workgroups out in
// blockIdx order, so the grid is a longest-processing-time-first schedule of the ACTUAL work - with document masks the work of a tile no
// longer follows from its index.  The kernels are launched with cap * nh workgroups; those beyond the header's count leave at once.
// ---------------------------------------------------------------------------------------------
struct DocPlan {
  const int32_t* header;
  const int32_t* doc_end;
  const int4* items_q;
  const int4* items_k;
};
constexpr int DOC_KIND_SHIFT = 30;
__host__ __device__ __forceinline__ int64_t doc_plan_tiles(int64_t B, int64_t T) { return B * ((T + 127) / 128); }
__host__ __device__ __forceinline__ int64_t doc_plan_cap(int64_t B, int64_t T) { return doc_plan_tiles(B, T) + doc_plan_tiles(B, T) / 4; }
__host__ __device__ __forceinline__ int64_t doc_plan_pq(int64_t B, int64_t T) { return (8 + B * T + 3) & ~(int64_t)3; }
__host__ __device__ __forceinline__ int64_t doc_plan_ints(int64_t B, int64_t T) { return doc_plan_pq(B, T) + 8 * doc_plan_cap(B, T); }
__host__ __device__ __forceinline__ DocPlan doc_plan_view(const int32_t* plan, int64_t B, int64_t T) {
  DocPlan p;
  p.header = plan;
  p.doc_end = plan + 8;
  p.items_q = reinterpret_cast<const int4*>(plan + doc_plan_pq(B, T));
  p.items_k = p.items_q + doc_plan_cap(B, T);
  return p;
}

// ---------------------------------------------------------------------------------------------
// -DPLM_ATTN_TRACE (tools/attn_trace.py builds such a copy of the library; never the shipped one): every workgroup of a document-mask kernel
// leaves {start, loop start, loop end, end} on the 100 MHz wall clock, its CU (HW_ID / XCC_ID), rank and cost in a host-provided buffer.
// ---------------------------------------------------------------------------------------------
#ifdef PLM_ATTN_TRACE
#define ATTN_TRACE_DECL()                                                                    \
  static __device__ unsigned long long* g_attn_trace = nullptr;                               \
  extern "C" int PLM_ATTN_TRACE_SETTER(unsigned long long* buf) {                             \
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_trace), &buf, sizeof(buf));               \
  }
#define ATTN_TRACE_T(var) const unsigned long long var = __builtin_amdgcn_s_memrealtime()
#define ATTN_TRACE_END(kid, rank_, cost_, t0, t1, t2)                                                                              \
  if (threadIdx.x == 0 && g_attn_trace) {                                                                                          \
    unsigned long long* r_ = g_attn_trace + ((size_t)(kid) * 65536 + blockIdx.x) * 8;                                             \
    r_[0] = t0; r_[1] = t1; r_[2] = t2; r_[3] = __builtin_amdgcn_s_memrealtime();                                                 \
    r_[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4); r_[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);                         \
    r_[6] = (unsigned long long)(rank_); r_[7] = (unsigned long long)(cost_);                                                      \
  }
#else
#define ATTN_TRACE_DECL()
#define ATTN_TRACE_T(var)
#define ATTN_TRACE_END(kid, rank_, cost_, t0, t1, t2)
#endif
