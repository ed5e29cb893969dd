// Counter-based dropout decisions shared by the dropout kernels (k7_dropout.hip) and the sparse-feature projection
// (k9_sparse.hip): Philox-4x32-10 keyed by a seed in DEVICE memory.  Two counter layouts, neither part of the C ABI:
//   draw4       one mask over a [rows x cols] table: decision (row, col) = word (col & 3) of
//               Philox(counter = (col >> 2, row, stream_id, tag), key = seed) < keep * 2^32  -- four COLUMNS per call;
//   draw_heads4 the per-head input masks (layers.py:34,132: every head drops its own copy of x): decision (row, col,
//               head h) = word (h & 3) of Philox(counter = (col, row, stream_id, h >> 2)) -- four HEADS per call, so
//               that the sparse kernels, which meet the non-zeros of x one (row, col) at a time, spend one call per
//               non-zero and four heads instead of one per head (the first layout cost them 4 x the Philox work).
#pragma once
#include "common.h"

namespace pygat {

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
  }
  return c;
}

struct DropRng {
  const uint64_t* seed;  // device memory, [1]
  uint32_t stream_id;    // separates the masks drawn from one seed (input / Wh / attention, level)
  uint32_t thresh;       // keep iff word < thresh  (keep = 1 - p)
  float scale;           // 1 / keep
};

__device__ __forceinline__ uint32_t word_of(const uint4& w, int q) {
  return q == 0 ? w.x : q == 1 ? w.y : q == 2 ? w.z : w.w;
}

// 4 decisions for (row, columns 4*c4 .. 4*c4+3)
__device__ __forceinline__ uint4 draw4(const DropRng& g, uint64_t seed, uint32_t row_lo, uint32_t row_hi, uint32_t c4) {
  return philox4x32_10(make_uint4(c4, row_lo, g.stream_id, row_hi), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
}

// 4 decisions for (row, col, heads 4*hq .. 4*hq+3)
__device__ __forceinline__ uint4 draw_heads4(const DropRng& g, uint64_t seed, uint32_t row, uint32_t col, uint32_t hq) {
  return philox4x32_10(make_uint4(col, row, g.stream_id, hq), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
}
__device__ __forceinline__ uint32_t keep_nibble(const DropRng& g, const uint4& w) {
  return (w.x < g.thresh ? 1u : 0u) | (w.y < g.thresh ? 2u : 0u) | (w.z < g.thresh ? 4u : 0u) | (w.w < g.thresh ? 8u : 0u);
}

static inline bool make_rng(float p, const void* seed, uint32_t stream_id, DropRng* g) {
  if (!(p >= 0.f && p <= 1.f)) return false;   // p = 1 (F.dropout accepts it): nothing is kept, everything becomes 0
  const double keep = 1.0 - (double)p;
  double t = keep * 4294967296.0;
  g->seed = (const uint64_t*)seed;
  g->stream_id = stream_id;
  g->thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
  g->scale = keep > 0.0 ? (float)(1.0 / keep) : 0.f;
  return true;
}


}  // namespace pygat
