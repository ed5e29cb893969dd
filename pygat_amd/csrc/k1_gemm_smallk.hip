// K1 fast path -- tall-skinny fp32 MFMA GEMM with a SMALL inner dimension (gfx950).
//
// C[M x N] = A[M x K] * op(B), K <= 256, M huge (one row per graph node): the projection
// Wh = h W of the reference (layers.py:35,134, all heads and the s/t columns at once) and the
// input gradient dX = dWh W^T.  These shapes are MFMA-bound in fp32 (157 TF peak is only
// ~20 flop/B), so the structure removes everything but MFMAs from the steady state:
//
//   * op(B) (K x BN, <= 84 KB) is staged ONCE per work-group into LDS, k-major, and is then
//     read-only: no barrier in the main loop, waves run free.
//   * A never touches LDS: lane (i = l&31, h = l>>5) of a wave owns row i of the wave's 32-row
//     tile and streams it in 32-float chunks straight into registers with 16-byte loads (the
//     two half-waves fetch the same addresses and pick the even / odd k of each float4, which
//     is exactly the A fragment of v_mfma_f32_32x32x2_f32: A[i][k = 2q + h]).
//   * 8 waves per work-group (2 per SIMD) share the LDS image; work-groups are persistent and
//     walk the row tiles with the next chunk prefetched during the current chunk's MFMAs.
#include "common.h"
#include "gemm_fast.h"
#include <stdlib.h>

namespace pygat {


template <bool TB, int NT, int NBUF, bool SV>
__global__ __launch_bounds__(512) void gemm_smallk_kernel(SmallKArgs g) {
  constexpr int BN = 32 * NT;
  // chunks of the A row stream in flight per wave: a chunk's MFMAs take 16 * NT * 64 cycles, which for one or two
  // column tiles (a head-parallel rank with 1-2 heads of 16: 0.4-0.9 us) does not cover an HBM access -- narrow
  // outputs prefetch three chunks ahead instead of one (the accumulators are small there, the registers are free):
  // NBUF = 4 for NT <= 2, else 2 (3 selectable: launch_smallk)
  constexpr int LDB = BN + 4;
  extern __shared__ __attribute__((aligned(16))) float Bs[];  // [K][LDB]
  const int n0 = blockIdx.y * BN;
  // ---- stage op(B) once
  if constexpr (!TB) {
#pragma unroll 8
    for (int idx = threadIdx.x; idx < g.K * BN; idx += 512) {
      const int k = idx / BN, n = idx % BN;
      Bs[k * LDB + n] = (n0 + n < g.N) ? g.B[(int64_t)k * g.ldb + n0 + n] : 0.f;
    }
  } else {
#pragma unroll 8
    for (int idx = threadIdx.x; idx < g.K * BN; idx += 512) {
      const int n = idx / g.K, k = idx % g.K;
      Bs[k * LDB + n] = (n0 + n < g.N) ? g.B[(int64_t)(n0 + n) * g.ldb + k] : 0.f;
    }
  }
  float* Us = Bs + g.K * LDB;   // [K][8]: svec, zero padded
  if constexpr (SV) {
    for (int idx = threadIdx.x; idx < g.K * 8; idx += 512) {
      const int k = idx >> 3, h = idx & 7;
      Us[idx] = (h < g.sv_n) ? g.svec[(int64_t)k * g.sv_ld + h] : 0.f;
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int nchunks = g.K / 32;
  if ((int)blockIdx.x >= g.tiles_m) return;
  const int my_tiles = (g.tiles_m - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_tiles * nchunks;  // chunks this wave streams, tile after tile

  // address of the wave's c-th chunk (clamped: every load below is unconditional, a conditionally
  // defined register array ends up in scratch memory)
  auto chunk_ptr = [&](int c) -> const float* {
    if (c > total - 1) c = total - 1;
    const int t = c / nchunks, kc = c - t * nchunks;
    int64_t row = ((int64_t)blockIdx.x + (int64_t)t * gridDim.x) * 256 + 32 * w + fr;
    if (row > g.M - 1) row = g.M - 1;
    return g.A + row * g.lda + kc * 32;
  };

  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

#define PYGAT_LOAD8(R, P)                                                                     \
  {                                                                                           \
    const float* p__ = (P);                                                                   \
    R##0 = ld4(p__); R##1 = ld4(p__ + 4); R##2 = ld4(p__ + 8); R##3 = ld4(p__ + 12);          \
    R##4 = ld4(p__ + 16); R##5 = ld4(p__ + 20); R##6 = ld4(p__ + 24); R##7 = ld4(p__ + 28);   \
  }
#define PYGAT_BREAD(B0, B1, C)                                                                \
  _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                         \
    B0[nt] = bs[(4 * (C)) * LDB + 32 * nt];                                                   \
    B1[nt] = bs[(4 * (C) + 2) * LDB + 32 * nt];                                               \
  }
// B fragments of float4 C+1 are read from LDS BEFORE the 2*NT MFMAs of float4 C are issued; the
// sched_barriers keep hipcc from sinking the ds_reads back to their first use (it otherwise
// serialises ds_read -> lgkmcnt(0) -> 2 MFMAs through one register pair).
#define PYGAT_MMA4(V, C, BC0, BC1, BN0, BN1)                                                  \
  {                                                                                           \
    const float a0 = fh ? (V).y : (V).x; /* k = 4C + fh */                                    \
    const float a1 = fh ? (V).w : (V).z; /* k = 4C + 2 + fh */                                \
    if ((C) < 7) PYGAT_BREAD(BN0, BN1, (C) + 1)                                               \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                         \
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, BC0[nt], acc[nt], 0, 0, 0);        \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                         \
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, BC1[nt], acc[nt], 0, 0, 0);        \
    __builtin_amdgcn_sched_barrier(0);                                                        \
  }
#define PYGAT_SACC(V, C)                                                                      \
  {                                                                                           \
    const float4 u0 = ld4(us + (4 * (C) + 0) * 8), u1 = ld4(us + (4 * (C) + 1) * 8);          \
    const float4 u2 = ld4(us + (4 * (C) + 2) * 8), u3 = ld4(us + (4 * (C) + 3) * 8);          \
    sacc.x = fmaf((V).w, u3.x, fmaf((V).z, u2.x, fmaf((V).y, u1.x, fmaf((V).x, u0.x, sacc.x)))); \
    sacc.y = fmaf((V).w, u3.y, fmaf((V).z, u2.y, fmaf((V).y, u1.y, fmaf((V).x, u0.y, sacc.y)))); \
    sacc.z = fmaf((V).w, u3.z, fmaf((V).z, u2.z, fmaf((V).y, u1.z, fmaf((V).x, u0.z, sacc.z)))); \
    sacc.w = fmaf((V).w, u3.w, fmaf((V).z, u2.w, fmaf((V).y, u1.w, fmaf((V).x, u0.w, sacc.w)))); \
  }
#define PYGAT_STEP(R, CIDX)                                                                   \
  {                                                                                           \
    const int c__ = (CIDX);                                                                   \
    const int t__ = c__ / nchunks, kc__ = c__ - t__ * nchunks;                                \
    const float* bs = Bs + (kc__ * 32 + fh) * LDB + fr;                                       \
    if constexpr (SV) if (sv_on) {                                                            \
      const float* us = Us + kc__ * 32 * 8 + 4 * fh;                                          \
      PYGAT_SACC(R##0, 0) PYGAT_SACC(R##1, 1) PYGAT_SACC(R##2, 2) PYGAT_SACC(R##3, 3)         \
      PYGAT_SACC(R##4, 4) PYGAT_SACC(R##5, 5) PYGAT_SACC(R##6, 6) PYGAT_SACC(R##7, 7)         \
    }                                                                                         \
    float bx0[NT], bx1[NT], by0[NT], by1[NT];                                                 \
    PYGAT_BREAD(bx0, bx1, 0)                                                                  \
    PYGAT_MMA4(R##0, 0, bx0, bx1, by0, by1) PYGAT_MMA4(R##1, 1, by0, by1, bx0, bx1)           \
    PYGAT_MMA4(R##2, 2, bx0, bx1, by0, by1) PYGAT_MMA4(R##3, 3, by0, by1, bx0, bx1)           \
    PYGAT_MMA4(R##4, 4, bx0, bx1, by0, by1) PYGAT_MMA4(R##5, 5, by0, by1, bx0, bx1)           \
    PYGAT_MMA4(R##6, 6, bx0, bx1, by0, by1) PYGAT_MMA4(R##7, 7, by0, by1, bx0, bx1)           \
    if (kc__ == nchunks - 1) store_tile(t__);                                                 \
  }

  // epilogue of one 32 x BN wave tile: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // The common case (no accumulate, tile fully inside M) is a straight run of 16 stores per column
  // block: per-element branches make hipcc wait vmcnt(0) around every store, which also drains the
  // prefetched A chunk.
  float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);   // SV: columns 4*fh .. 4*fh+3 of row (tile row0 + fr)
  const bool sv_on = SV && blockIdx.y == 0;        // one column tile of work-groups takes the extra columns
  auto store_tile = [&](int t) {
    const int64_t row0 = ((int64_t)blockIdx.x + (int64_t)t * gridDim.x) * 256 + 32 * w;
    const bool full = row0 + 32 <= g.M;  // wave-uniform
    if constexpr (SV) {
      if (sv_on && row0 + fr < g.M) {
        float* so = g.s_out + (row0 + fr) * g.s_ld + 4 * fh;
        if (4 * fh + 0 < g.sv_n) so[0] = sacc.x;
        if (4 * fh + 1 < g.sv_n) so[1] = sacc.y;
        if (4 * fh + 2 < g.sv_n) so[2] = sacc.z;
        if (4 * fh + 3 < g.sv_n) so[3] = sacc.w;
      }
      sacc = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = n0 + 32 * nt + fr;
      int64_t ld;
      float* base = out_segment(g.out, col, ld);
      base += (row0 + 4 * fh) * ld;
      if (col < g.N) {
        if (full && !g.accumulate) {
#pragma unroll
          for (int r = 0; r < 16; ++r) base[((r & 3) + 8 * (r >> 2)) * ld] = acc[nt][r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2);
            if (row0 + 4 * fh + rr < g.M) {
              if (g.accumulate) base[rr * ld] += acc[nt][r]; else base[rr * ld] = acc[nt][r];
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    }
  };

  // (named registers, not arrays: a register array that is conditionally consumed ends up in scratch memory)
  float4 ra0, ra1, ra2, ra3, ra4, ra5, ra6, ra7;
  float4 rb0, rb1, rb2, rb3, rb4, rb5, rb6, rb7;
  if constexpr (NBUF == 2) {
    PYGAT_LOAD8(ra, chunk_ptr(0))
    for (int c = 0; c < total; c += 2) {
      PYGAT_LOAD8(rb, chunk_ptr(c + 1))
      PYGAT_STEP(ra, c)
      PYGAT_LOAD8(ra, chunk_ptr(c + 2))
      if (c + 1 < total) PYGAT_STEP(rb, c + 1)
    }
  } else if constexpr (NBUF == 3) {
    float4 rc0, rc1, rc2, rc3, rc4, rc5, rc6, rc7;
    PYGAT_LOAD8(ra, chunk_ptr(0))
    PYGAT_LOAD8(rb, chunk_ptr(1))
    for (int c = 0; c < total; c += 3) {
      PYGAT_LOAD8(rc, chunk_ptr(c + 2))
      PYGAT_STEP(ra, c)
      PYGAT_LOAD8(ra, chunk_ptr(c + 3))
      if (c + 1 < total) PYGAT_STEP(rb, c + 1)
      PYGAT_LOAD8(rb, chunk_ptr(c + 4))
      if (c + 2 < total) PYGAT_STEP(rc, c + 2)
    }
  } else {
    float4 rc0, rc1, rc2, rc3, rc4, rc5, rc6, rc7;
    float4 rd0, rd1, rd2, rd3, rd4, rd5, rd6, rd7;
    PYGAT_LOAD8(ra, chunk_ptr(0))
    PYGAT_LOAD8(rb, chunk_ptr(1))
    PYGAT_LOAD8(rc, chunk_ptr(2))
    for (int c = 0; c < total; c += 4) {
      PYGAT_LOAD8(rd, chunk_ptr(c + 3))
      PYGAT_STEP(ra, c)
      PYGAT_LOAD8(ra, chunk_ptr(c + 4))
      if (c + 1 < total) PYGAT_STEP(rb, c + 1)
      PYGAT_LOAD8(rb, chunk_ptr(c + 5))
      if (c + 2 < total) PYGAT_STEP(rc, c + 2)
      PYGAT_LOAD8(rc, chunk_ptr(c + 6))
      if (c + 3 < total) PYGAT_STEP(rd, c + 3)
    }
  }
#undef PYGAT_LOAD8
#undef PYGAT_SACC
#undef PYGAT_BREAD
#undef PYGAT_MMA4
#undef PYGAT_STEP
}

template <bool TB, bool SV>
static hipError_t launch_smallk(const SmallKArgs& g, int NT, dim3 grid, size_t lds, hipStream_t st) {
  int dev = -1;
  (void)hipGetDevice(&dev);
  // (three A chunks in flight at 3-5 column tiles were measured: 0.434 -> 0.437 ms at 5 tiles, DESIGN.md section 8: two)
#define PYGAT_SMALLK_LAUNCH(n, nb)                                                                        \
  {                                                                                                       \
    static bool attr_set[64] = {};   /* per device: the attribute belongs to the device's code object */ \
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_smallk_kernel<TB, n, nb, SV>),        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                  \
      if (dev >= 0 && dev < 64) attr_set[dev] = true;                                                     \
    }                                                                                                     \
    hipLaunchKernelGGL((gemm_smallk_kernel<TB, n, nb, SV>), grid, dim3(512), lds, st, g);                 \
  }
#define PYGAT_SMALLK_CASE(n)                                                                              \
  case n:                                                                                                 \
    PYGAT_SMALLK_LAUNCH(n, 2)                                                                             \
    break;
  switch (NT) {
    case 1: PYGAT_SMALLK_LAUNCH(1, 4) break;
    case 2: PYGAT_SMALLK_LAUNCH(2, 4) break;
    PYGAT_SMALLK_CASE(3)
    PYGAT_SMALLK_CASE(4)
    default: PYGAT_SMALLK_CASE(5)
  }
#undef PYGAT_SMALLK_CASE
#undef PYGAT_SMALLK_LAUNCH
  return hipGetLastError();
}

// returns 1 if the fast path took the call, 0 if the shape does not qualify, <0 on launch error
int try_gemm_smallk(int transB, int M, int N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                    const pygat_out_segments* out, int accumulate, bool split, hipStream_t st, const float* svec, int64_t sv_ld,
                    int sv_n, float* s_out, int64_t s_ld) {
  if (K < 32 || K > 256 || (K % 32) != 0 || M < 8192) return 0;
  if (!aligned16(A) || (lda % 4) != 0) return 0;
  if (svec && (sv_n < 1 || sv_n > 8 || !s_out || transB)) return 0;
  const int nt_needed = (int)cdiv(N, 32);
  // up to 160 columns in one tile; wider outputs in balanced 128-column tiles
  const int NT = nt_needed <= 5 ? nt_needed : 4;
  const size_t lds = (size_t)K * (32 * NT + 4) * sizeof(float) + (svec ? (size_t)K * 8 * sizeof(float) : 0);
  if (lds > 150 * 1024) return 0;
  SmallKArgs g;
  g.M = M; g.N = N; g.K = (int)K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.out = *out; g.accumulate = accumulate;
  g.svec = svec; g.sv_ld = sv_ld; g.sv_n = sv_n; g.s_out = s_out; g.s_ld = s_ld;
  g.tiles_m = (int)cdiv(M, 256);
  g.transB = transB; g.sr_a = nullptr; g.sr_fp = 0;
  const int tiles_n = (int)cdiv(N, 32 * NT);
  int gx = 256 / tiles_n;
  if (gx < 1) gx = 1;
  if (gx > g.tiles_m) gx = g.tiles_m;
  dim3 grid((unsigned)gx, (unsigned)tiles_n, 1);
  if (split) {   // the same GEMM on the bf16 MFMA pipe from exactly split operands (k1_gemm_x3.hip)
    const int r = try_gemm_smallk_x3(g, NT, grid, st);
    if (r != 0) return r;
  }
  hipError_t e = transB ? launch_smallk<true, false>(g, NT, grid, lds, st)
                        : (svec ? launch_smallk<false, true>(g, NT, grid, lds, st) : launch_smallk<false, false>(g, NT, grid, lds, st));
  if (e != hipSuccess) {
    set_error("gemm_smallk: %s", hipGetErrorString(e));
    return PYGAT_EHIP;
  }
  return 1;
}


// ---------------------------------------------------------------------------------------------
// Weight-gradient fast path: C[M x N] = A^T B with A [K x M], B [K x N], K huge (one row per node),
// M and N small: dW = X^T dWh (autograd of layers.py:35,134).  Both operands are k-strided, i.e. a
// fragment of v_mfma_f32_32x32x2_f32 (lane (i = l&31, h = l>>5) -> A[k = 2q+h][m0+i], B[2q+h][n0+i])
// is two coalesced 128-B segments of one wave-wide dword load: no LDS, no barrier.  Each wave owns a
// 32 x (32*NT) block of C, the work-group a 128 x (32*NT) tile of one K slab; slabs go to the
// split-K workspace and are summed in slab order by gemm_splitk_reduce_kernel.

template <int NT, int UK>
__global__ __launch_bounds__(256) void gemm_tn_stream_kernel(TnArgs g) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int m = blockIdx.y * 128 + 32 * w + fr;
  const int n0 = blockIdx.z * 32 * NT;
  const int64_t kbeg = (int64_t)blockIdx.x * g.k_per_split;
  const int64_t kend = (kbeg + g.k_per_split < g.K) ? kbeg + g.k_per_split : g.K;
  const float* ap = g.A + (m < g.M ? m : 0);
  const float* bp[NT];
  int64_t ldt[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + 32 * nt + fr;
    const bool second = g.B2 != nullptr && n0 + 32 * nt >= g.N1;   // wave-uniform; tiles past N read a clamped column
    const int nn = second ? n - g.N1 : n, lim = second ? g.N - g.N1 : (g.B2 ? g.N1 : g.N);
    bp[nt] = (second ? g.B2 : g.B) + (nn < lim ? nn : 0);
    ldt[nt] = second ? g.ldb2 : g.ldb;
  }
  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // Lanes whose m / n fall outside C read a clamped (valid) column: their products only reach rows /
  // columns of C that are never stored, so the steady state needs no masking at all -- a select on
  // a loaded value makes hipcc branch around the load and wait vmcnt(0) after each one.
  // double-buffered: the 8 k2-steps of block i+1 are in flight while block i's MFMAs issue
#define PYGAT_TN_LOAD(AX, BX, KBASE)                                                          \
  _Pragma("unroll") for (int u = 0; u < UK; ++u) {                                            \
    const int64_t k = (KBASE) + 2 * u + fh;                                                   \
    AX[u] = ap[k * g.lda];                                                                    \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) BX[u][nt] = bp[nt][k * ldt[nt]];       \
  }
#define PYGAT_TN_MMA(AX, BX)                                                                  \
  _Pragma("unroll") for (int u = 0; u < UK; ++u)                                              \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                       \
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(AX[u], BX[u][nt], acc[nt], 0, 0, 0);
  int64_t k0 = kbeg;
  const int64_t nfull = (kend - kbeg) / (2 * UK);  // full blocks of 16 k rows
  // Two-level sum: an MFMA accumulator rounds every addition at the size of the running sum, so the error of a slab
  // grows with the root of its length -- every TN_FLUSH_ROWS rows the running tile is added into a second set of
  // registers and restarted (8192-row slabs of a 262144-node level sat at 3.5-4.2 x the error of the fp32 CPU port,
  // tests/test_gpu_fullsize.py; the kernel runs one wave per SIMD, the registers are free).
  constexpr int TN_FLUSH_ROWS = 1024;
  f32x16 acc2[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[i][r] = 0.f;
  if (nfull > 0) {
    float a0[UK], b0[UK][NT], a1[UK], b1[UK][NT];
    PYGAT_TN_LOAD(a0, b0, k0)
    for (int64_t i = 0; i < nfull; i += 2) {
      if (i > 0 && (i % (TN_FLUSH_ROWS / (2 * UK))) == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) { acc2[t][r] += acc[t][r]; acc[t][r] = 0.f; }
      }
      // clamp the prefetch address to the last full block: loads stay unconditional
      const int64_t kn1 = kbeg + ((i + 1 < nfull) ? i + 1 : nfull - 1) * 2 * UK;
      PYGAT_TN_LOAD(a1, b1, kn1)
      __builtin_amdgcn_sched_barrier(0);
      PYGAT_TN_MMA(a0, b0)
      __builtin_amdgcn_sched_barrier(0);
      const int64_t kn2 = kbeg + ((i + 2 < nfull) ? i + 2 : nfull - 1) * 2 * UK;
      PYGAT_TN_LOAD(a0, b0, kn2)
      __builtin_amdgcn_sched_barrier(0);
      if (i + 1 < nfull) PYGAT_TN_MMA(a1, b1)
      __builtin_amdgcn_sched_barrier(0);
    }
    k0 = kbeg + nfull * 2 * UK;
  }
#undef PYGAT_TN_LOAD
#undef PYGAT_TN_MMA
  for (; k0 < kend; k0 += 2) {  // K tail of the last slab: rows past kend contribute a zero A operand
    const int64_t k = k0 + fh;
    const float keep = k < kend ? 1.f : 0.f;
    const int64_t kk = k < kend ? k : kend - 1;
    const float av = ap[kk * g.lda] * keep;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[nt][kk * ldt[nt]], acc[nt], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] += acc2[t][r];
  float* base = g.ws + (int64_t)blockIdx.x * g.M * g.N;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = n0 + 32 * nt + fr;
    if (col >= g.N) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = blockIdx.y * 128 + 32 * w + (r & 3) + 8 * (r >> 2) + 4 * fh;
      if (row < g.M) base[(int64_t)row * g.N + col] = acc[nt][r];
    }
  }
}

// (A wide-load variant -- every lane fetching four consecutive columns of a k row, the tile dealt out by column residue -- was
// built in round 1 and measured slower: it re-fetches A once per wave, 3.5 GB of HBM reads for a 1 GB problem.  Removed in round 4.)

// picks the slab count, launches; returns the number of slabs written to ws (>= 1), 0 if the shape
// does not qualify, < 0 on error.  ws must hold max_splits * M * N floats.
int try_gemm_tn_stream(int M, int N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                       int max_splits, float* ws, bool split, hipStream_t st, int N1, const float* B2, int64_t ldb2) {
  if (K < 4096 || (int64_t)M * N > 512 * 512 || max_splits < 1 || !ws) return 0;
  if (!B2) N1 = N;
  if (B2 && (N1 <= 0 || N1 >= N || (N1 % 32) != 0)) return 0;
  const int nt_needed = (int)cdiv(N, 32);
  // a fifth tile for the few extra columns of a second operand: one pass over A instead of two column tiles
  const int NT = nt_needed < 4 ? nt_needed : ((B2 && nt_needed == 5) ? 5 : 4);
  const int tiles_m = (int)cdiv(M, 128), tiles_n = (int)cdiv(N, 32 * NT);
  int splits = max_splits;
  // narrow outputs (one or two 32-column tiles: a head-parallel rank with 1-2 heads of 16) run one 4-wave work-group
  // per CU with few registers: 16 k-pairs per buffer instead of 8 keep twice the bytes in flight
  // measured (1M x 128 operand, 256 slabs): 16 columns 0.162 / 0.127 / 0.121 ms at 8 / 16 / 32 k-pairs per buffer,
  // 64 columns 0.229 / 0.201 / 0.213, 128 columns 0.414 / 0.433 / 0.408 (MFMA-bound there): narrow outputs (a
  // head-parallel rank with 1-2 heads of 16) keep more bytes in flight
  const int uk = NT <= 2 ? 16 : 8;
  int64_t kps = cdiv(cdiv(K, splits), 2 * uk) * 2 * uk;
  splits = (int)cdiv(K, kps);
  TnArgs g;
  g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.k_per_split = kps; g.ws = ws;
  g.N1 = N1; g.B2 = B2; g.ldb2 = ldb2;
  if (split) {
    const int64_t kps16 = cdiv(cdiv(K, max_splits), 48) * 48;
    TnArgs gx = g;
    gx.k_per_split = kps16;
    const int r = try_gemm_tn_x3(gx, (int)cdiv(K, kps16), st);
    if (r != 0) return r;
  }
  dim3 grid((unsigned)splits, (unsigned)tiles_m, (unsigned)tiles_n);
#define PYGAT_TN_CASE(NTV)                                                                                      \
  case NTV:                                                                                                     \
    if (uk == 16) hipLaunchKernelGGL((gemm_tn_stream_kernel<NTV, 16>), grid, dim3(256), 0, st, g);              \
    else hipLaunchKernelGGL((gemm_tn_stream_kernel<NTV, 8>), grid, dim3(256), 0, st, g);                        \
    break;
  switch (NT) {
    PYGAT_TN_CASE(1)
    PYGAT_TN_CASE(2)
    PYGAT_TN_CASE(3)
    PYGAT_TN_CASE(5)
    default:
    PYGAT_TN_CASE(4)
  }
#undef PYGAT_TN_CASE
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("gemm_tn_stream: %s", hipGetErrorString(e));
    return PYGAT_EHIP;
  }
  return splits;
}

}  // namespace pygat
