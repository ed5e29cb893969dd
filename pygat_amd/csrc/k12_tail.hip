// Self-loop-only nodes (round 5).  A node whose only neighbour is itself -- 55 % of the R-MAT workload's nodes once self loops
// are added (utils.py:52 adds them to every node) -- has a one-element softmax row: alpha_ii = 1 exactly, so
//   forward   h'_i = ELU(Wh_i (+ skip_i))                     (layers.py:146-170 with one edge; m_i, Z_i only shift / scale alpha)
//   backward  de_ii = alpha (dp - sum alpha dp) = 0  =>  dz = ds = dt = 0,  dWh_i = Gp_i      (layers.py:81-90 and its autograd)
// In a degree-ordered pattern (CSRGraph.degree_ordered) these nodes are a contiguous TAIL of the row range, and their slots a
// suffix of the slot list: the fused kernels run on the slot prefix (pygat_graph.slot_count) and the two streams below take the
// tail at the memory system's streaming rate instead of 64 row-finish rounds per slot -- measured at config 5: the tail cost the
// fused forward 0.16-0.18 of its 0.95 ms against 0.12 as a stream (tools/tail_cost.py).
// Exactness: the forward values are bit-identical to the fused kernel's (acc = 1 * Wh_i, 1 / Z = 1); the backward's exact zeros
// replace rounding residue of the order 1e-7 |dp| (the fused kernel forms dp - D from two differently rounded dot products).
#include "attn_common.h"

namespace pygat {

__global__ __launch_bounds__(256) void fwd_tail_kernel(int row_first, int n_rows, int H, int Fo, int Fp, int flags,
                                                      const float* __restrict__ Wh, int64_t ldwh, const float* __restrict__ sk,
                                                      float* __restrict__ out, const int32_t* __restrict__ urow,
                                                      float* __restrict__ m, float* __restrict__ Z, float* __restrict__ qneg) {
  const int R4 = H * Fp / 4;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)n_rows * R4) return;
  const int64_t i = row_first + idx / R4;
  const int co = 4 * (int)(idx % R4), h = co / Fp, f0 = co % Fp;
  float4 v = ld4(Wh + i * ldwh + co);
  if (flags & PYGAT_F_SKIP) {
    const float4 k4 = ld4(sk + i * (int64_t)(H * Fp) + co);
    v.x += k4.x; v.y += k4.y; v.z += k4.z; v.w += k4.w;
  }
  if (flags & PYGAT_F_ELU) { v.x = elu1(v.x); v.y = elu1(v.y); v.z = elu1(v.z); v.w = elu1(v.w); }
  const int64_t io = urow ? (int64_t)urow[i] : i;
  if (Fo == Fp) {
    st4(out + io * (int64_t)(H * Fo) + co, v);
  } else {
    float* o = out + io * (int64_t)(H * Fo) + (int64_t)h * Fo + f0;
    if (f0 + 0 < Fo) o[0] = v.x;
    if (f0 + 1 < Fo) o[1] = v.y;
    if (f0 + 2 < Fo) o[2] = v.z;
    if (f0 + 3 < Fo) o[3] = v.w;
  }
  if (f0 == 0 && m) {
    m[i * H + h] = 0.f; Z[i * H + h] = 1.f;     // alpha = exp(e - m) / Z with a single edge is 1 whatever the shift: never read back
    if (qneg) qneg[i * H + h] = 0.f;            // not a mixed-branch row: K3a leaves ds_i = 0 and does not read aneg_i
  }
}

__global__ __launch_bounds__(256) void col_tail_kernel(int row_first, int n_rows, int H, int Fp, const float* __restrict__ GR,
                                                      int64_t ldgr, float* __restrict__ dWh, float* __restrict__ dt) {
  const int R4 = H * Fp / 4;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)n_rows * R4) return;
  const int64_t j = row_first + idx / R4;
  const int co = 4 * (int)(idx % R4);
  st4(dWh + j * (int64_t)(H * Fp) + co, ld4(GR + j * ldgr + co));      // dWh_j = alpha_jj Gp_j + ds_j a_src + dt_j a_dst = Gp_j
  if (co % Fp == 0) dt[j * H + co / Fp] = 0.f;
}

// The whole backward of the tail in one stream (levels without a skip projection, whose weight gradient is the only other reader
// of Gp): dWh_j = Gp_j = G_u ELU'(out_u) straight from the caller's rows u = user_row[j] -- the tail's rows of GR are neither
// written (pygat_gat_backward_prepare runs on the rows before the tail) nor read.  ELU' is recovered from the output exactly as
// K3a does (out > 0 ? 1 : out + 1).
__global__ __launch_bounds__(256) void bwd_tail_kernel(int row_first, int n_rows, int H, int Fo, int Fp, int flags,
                                                      const float* __restrict__ G, const float* __restrict__ y,
                                                      const int32_t* __restrict__ urow, float* __restrict__ dWh, int64_t ld_dwh,
                                                      int zero_cols, float* __restrict__ ds, float* __restrict__ dt) {
  const int R4 = H * Fp / 4;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)n_rows * R4) return;
  const int64_t j = row_first + idx / R4;
  const int co = 4 * (int)(idx % R4), h = co / Fp, f0 = co % Fp;
  const int64_t ju = urow ? (int64_t)urow[j] : j;
  float g[4] = {0.f, 0.f, 0.f, 0.f}, o[4] = {1.f, 1.f, 1.f, 1.f};
  if (Fo == Fp) {
    const float4 g4 = ld4(G + ju * (int64_t)(H * Fo) + co), y4 = ld4(y + ju * (int64_t)(H * Fo) + co);
    g[0] = g4.x; g[1] = g4.y; g[2] = g4.z; g[3] = g4.w; o[0] = y4.x; o[1] = y4.y; o[2] = y4.z; o[3] = y4.w;
  } else {
    const int64_t b = ju * (int64_t)(H * Fo) + (int64_t)h * Fo + f0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (f0 + q < Fo) { g[q] = G[b + q]; o[q] = y[b + q]; }
  }
  if (flags & PYGAT_F_ELU) {
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] *= o[q] > 0.f ? 1.f : o[q] + 1.f;
  }
  st4(dWh + j * ld_dwh + co, make_float4(g[0], g[1], g[2], g[3]));
  if (co < zero_cols) st4(dWh + j * ld_dwh + H * Fp + co, make_float4(0.f, 0.f, 0.f, 0.f));   // (GATv2: the dWhj half of the row)
  if (f0 == 0) { if (dt) dt[j * H + h] = 0.f; if (ds) ds[j * H + h] = 0.f; }
}

}  // namespace pygat

using namespace pygat;

extern "C" int pygat_gat_forward_tail(int row_first, int n_rows, int H, int Fo, int flags, const float* Wh, int64_t ldwh,
                                      const float* sk, float* out, const int32_t* user_row, float* m, float* Z, float* qneg,
                                      void* stream) {
  const int Fp = padded_width(Fo);
  if (ldwh == 0) ldwh = (int64_t)H * Fp;
  PYGAT_REQUIRE(row_first >= 0 && n_rows > 0 && H > 0 && Fp > 0 && Wh && out && ldwh >= (int64_t)H * Fp && (ldwh % 4) == 0,
                "gat_forward_tail: bad arguments");
  PYGAT_REQUIRE(!(flags & PYGAT_F_SKIP) || sk, "gat_forward_tail: PYGAT_F_SKIP without sk");
  PYGAT_REQUIRE((m == nullptr) == (Z == nullptr) && (!qneg || m), "gat_forward_tail: m and Z come together (qneg with them)");
  PYGAT_REQUIRE(aligned16(Wh) && (!sk || aligned16(sk)) && (Fo != Fp || aligned16(out)), "gat_forward_tail: row tables must be 16-byte aligned");
  const int64_t items = (int64_t)n_rows * (H * Fp / 4);
  hipLaunchKernelGGL(fwd_tail_kernel, dim3((unsigned)cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, row_first, n_rows, H, Fo, Fp,
                     flags & (PYGAT_F_ELU | PYGAT_F_SKIP), Wh, ldwh, sk, out, user_row, m, Z, qneg);
  PYGAT_CHECK_LAUNCH("gat_forward_tail");
  return PYGAT_OK;
}

extern "C" int pygat_gat_backward_col_tail(int row_first, int n_rows, int H, int Fo, const float* GR, float* dWh, float* dt,
                                           void* stream) {
  const int Fp = padded_width(Fo);
  PYGAT_REQUIRE(row_first >= 0 && n_rows > 0 && H > 0 && Fp > 0 && GR && dWh && dt, "gat_backward_col_tail: bad arguments");
  PYGAT_REQUIRE(aligned16(GR) && aligned16(dWh), "gat_backward_col_tail: row tables must be 16-byte aligned");
  const int64_t items = (int64_t)n_rows * (H * Fp / 4);
  hipLaunchKernelGGL(col_tail_kernel, dim3((unsigned)cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, row_first, n_rows, H, Fp, GR,
                     (int64_t)H * Fp + 4 * H, dWh, dt);
  PYGAT_CHECK_LAUNCH("gat_backward_col_tail");
  return PYGAT_OK;
}

extern "C" int pygat_gat_backward_tail(int row_first, int n_rows, int H, int Fo, int flags, const float* G, const float* y,
                                       const int32_t* user_row, float* dWh, int64_t ld_dwh, int zero_cols, float* ds, float* dt,
                                       void* stream) {
  const int Fp = padded_width(Fo);
  if (ld_dwh == 0) ld_dwh = (int64_t)H * Fp;
  PYGAT_REQUIRE(row_first >= 0 && n_rows > 0 && H > 0 && Fp > 0 && G && y && dWh, "gat_backward_tail: bad arguments");
  PYGAT_REQUIRE(zero_cols >= 0 && zero_cols <= H * Fp && (zero_cols % 4) == 0 && ld_dwh >= (int64_t)H * Fp + zero_cols && (ld_dwh % 4) == 0,
                "gat_backward_tail: bad row stride / zero columns");
  PYGAT_REQUIRE(aligned16(dWh) && (Fo != Fp || (aligned16(G) && aligned16(y))), "gat_backward_tail: row tables must be 16-byte aligned");
  const int64_t items = (int64_t)n_rows * (H * Fp / 4);
  hipLaunchKernelGGL(bwd_tail_kernel, dim3((unsigned)cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, row_first, n_rows, H, Fo, Fp,
                     flags & PYGAT_F_ELU, G, y, user_row, dWh, ld_dwh, zero_cols, ds, dt);
  PYGAT_CHECK_LAUNCH("gat_backward_tail");
  return PYGAT_OK;
}
