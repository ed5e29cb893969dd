// Shared host/device helpers for the gfx950 GAT kernels (not part of the C ABI).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/pygat_amd.h"

namespace pygat {

void set_error(const char* fmt, ...);

#define PYGAT_REQUIRE(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      pygat::set_error(__VA_ARGS__);        \
      return PYGAT_EINVAL;                  \
    }                                       \
  } while (0)

// call after a kernel launch; never synchronises
#define PYGAT_CHECK_LAUNCH(what)                                          \
  do {                                                                    \
    hipError_t e__ = hipGetLastError();                                   \
    if (e__ != hipSuccess) {                                              \
      pygat::set_error("%s: %s", what, hipGetErrorString(e__));           \
      return PYGAT_EHIP;                                                  \
    }                                                                     \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int ilog2(int x) { int l = 0; while ((1 << l) < x) ++l; return l; }
#ifdef __HIPCC__
__host__ __device__
#endif
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Column-block addressing of a pygat_col_blocks matrix inside the kernels: lw = log2(w), bs = block stride in floats; an
// ordinary matrix is the one-block case (lw = 62, bs = 0), so kernels carry no second code path for it.
struct ColBlocks {
  int lw;
  int64_t bs;
};
static inline ColBlocks col_blocks_of(const pygat_col_blocks* b) {
  ColBlocks c;
  c.lw = 62; c.bs = 0;
  if (b && b->w > 0) { c.lw = 0; while ((1 << c.lw) < b->w) ++c.lw; c.bs = b->stride; }
  return c;
}
static inline bool col_blocks_on(const pygat_col_blocks* b) { return b && b->w > 0; }
// w a power of two >= 16 that divides `cols`; stride and row stride 16-byte multiples
static inline bool col_blocks_ok(const pygat_col_blocks* b, int64_t cols, int64_t ld) {
  if (!col_blocks_on(b)) return true;
  return b->w >= 16 && (b->w & (b->w - 1)) == 0 && cols % b->w == 0 && (b->stride % 4) == 0 && ld >= b->w && (ld % 4) == 0;
}
#ifdef __HIPCC__
// float offset of column c of row 0 (c and c + 3 lie in one block when c % 4 == 0)
__host__ __device__ __forceinline__ int64_t blk_off(int64_t c, const ColBlocks& b) {
  return (c >> b.lw) * b.bs + (c & ((int64_t(1) << b.lw) - 1));
}
#endif

// width of one head inside a padded row: power of two in [4, 256]
static inline int padded_width(int f) {
  if (f <= 0 || f > 256) return 0;
  int p = 4;
  while (p < f) p <<= 1;
  return p;
}

#ifdef __HIPCC__
// ---- wave64 cross-lane helpers -------------------------------------------------
// DPP controls (gfx9): quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E,
// row_half_mirror = 0x141, row_mirror = 0x140.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  int v = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true);
  return __int_as_float(v);
}

// Sum over aligned groups of G consecutive lanes (G power of two <= 64); every lane
// of a group ends with the group total.  Steps 1,2 use quad_perm, 4 and 8 use the
// mirror controls (valid because after the previous steps all lanes of a
// sub-group already hold equal values), 16 and 32 go through ds_bpermute.
template <int G>
__device__ __forceinline__ float group_sum(float x) {
  if constexpr (G >= 2) x += dpp_mov<0xB1>(x);
  if constexpr (G >= 4) x += dpp_mov<0x4E>(x);
  if constexpr (G >= 8) x += dpp_mov<0x141>(x);
  if constexpr (G >= 16) x += dpp_mov<0x140>(x);
  if constexpr (G >= 32) x += __shfl_xor(x, 16);
  if constexpr (G >= 64) x += __shfl_xor(x, 32);
  return x;
}

// runtime group size (power of two <= 64)
__device__ __forceinline__ float group_sum_rt(float x, int g) {
  if (g >= 2) x += dpp_mov<0xB1>(x);
  if (g >= 4) x += dpp_mov<0x4E>(x);
  if (g >= 8) x += dpp_mov<0x141>(x);
  if (g >= 16) x += dpp_mov<0x140>(x);
  if (g >= 32) x += __shfl_xor(x, 16);
  if (g >= 64) x += __shfl_xor(x, 32);
  return x;
}

__device__ __forceinline__ int ilog2_dev(int x) { return 31 - __clz(x); }   // x a power of two
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float dot4(float4 a, float4 b) {
  return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}
#endif

}  // namespace pygat
