// K4 -- backward, column pass: dWh, dt, ds (gfx950, wave64).
//
// Replaces `grad_b = a.t().matmul(grad_output)` of SpecialSpmmFunction.backward (reference
// layers.py:89) -- the transposed SpMM -- plus the autograd of the two a-halves matmuls
// (layers.py:60-61):
//     dWh_j = sum_i alpha_ij Gp_i + ds_j a_src + dt_j a_dst
//     dt_j  = sum_i dz_ij          (column sums of dz)
//     ds_j  = sum_k dz_jk          (row sums of dz)
// nnz split over the TRANSPOSED pattern: slot walks transposed edges (j <- i), gathers Gp_i (one
// head-interleaved row, 16 B per lane); the (alpha, dz) pairs are either gathered through perm_t
// (wide rows) or were scattered by K3b into this order and stream in sequentially (narrow rows).  For a symmetric pattern the transposed CSR has the
// forward layout, so position k is also the forward edge (j, col[k]) and ds_j is accumulated from
// dz_f[k] on the way -- no atomics, no second pass.  Rows cut by a slot border go through `part` + a fix-up launch (fixed order).
#include "attn_common.h"

namespace pygat {

struct ColArgs {
  GraphDev g;  // transposed pattern
  RowShape rs;
  const int32_t* perm;  // nullptr: ebuf is in this (transposed) order; else forward edge of each edge here
  int symmetric;
  const float* Gp;
  const float* ebuf;  // [nnz][2][H] (alpha, dz)
  const float* dzf;   // [nnz][H] dz in forward edge order (== this order's positions if symmetric)
  const float* a_pad;
  float* dWh;
  float* ds;
  float* dt;
  float* part;  // [2 * nslots][R + 2H]: acc[R], dt[H], ds[H]
};

template <int VEC>
__device__ __forceinline__ void col_finish(const ColArgs& a, const LaneCols<VEC>& lc, int j,
                                           const float4 (&acc)[VEC], const float (&dt)[VEC],
                                           const float (&ds)[VEC]) {
  const int H = a.rs.H, R = a.rs.R, Fp = a.rs.Fp;
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    if (!lc.valid[v]) continue;
    const int co = lc.cofs[v], h = lc.head[v], f0 = co & (Fp - 1);
    const float dsj = a.symmetric ? ds[v] : a.ds[(int64_t)j * H + h];
    const float4 as = ld4(a.a_pad + (int64_t)h * 2 * Fp + f0);
    const float4 ad = ld4(a.a_pad + (int64_t)h * 2 * Fp + Fp + f0);
    float4 o;
    o.x = acc[v].x + dsj * as.x + dt[v] * ad.x;
    o.y = acc[v].y + dsj * as.y + dt[v] * ad.y;
    o.z = acc[v].z + dsj * as.z + dt[v] * ad.z;
    o.w = acc[v].w + dsj * as.w + dt[v] * ad.w;
    st4(a.dWh + (int64_t)j * R + co, o);
    if (((co >> 2) & (a.rs.lph - 1)) == 0) {
      a.dt[(int64_t)j * H + h] = dt[v];
      if (a.symmetric) a.ds[(int64_t)j * H + h] = ds[v];
    }
  }
}

template <int VEC>
__device__ __forceinline__ void col_flush(const ColArgs& a, const LaneCols<VEC>& lc, int64_t k, int j,
                                          bool is_head, bool is_tail, const float4 (&acc)[VEC],
                                          const float (&dt)[VEC], const float (&ds)[VEC]) {
  if (is_head || is_tail) {
    float* p = a.part + (2 * k + (is_head ? 0 : 1)) * (int64_t)(a.rs.R + 2 * a.rs.H);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      if (!lc.valid[v]) continue;
      st4(p + lc.cofs[v], acc[v]);
      if (((lc.cofs[v] >> 2) & (a.rs.lph - 1)) == 0) {
        p[a.rs.R + lc.head[v]] = dt[v];
        p[a.rs.R + a.rs.H + lc.head[v]] = ds[v];
      }
    }
  } else {
    col_finish<VEC>(a, lc, j, acc, dt, ds);
  }
}

template <int LPR, int VEC>
__global__ __launch_bounds__(256) void gat_bwd_col_kernel(ColArgs a) {
  constexpr int EPW = 64 / LPR;
  constexpr int U = (VEC == 1) ? 4 : 2;
  const int lane = threadIdx.x & 63;
  const int64_t k = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * EPW + lane / LPR;
  if (k >= num_slots(a.g)) return;
  int64_t e0, e1;
  slot_range(a.g, k, &e0, &e1);
  const LaneCols<VEC> lc = lane_cols<LPR, VEC>(a.rs);
  const int H = a.rs.H, R = a.rs.R;
  const int2* __restrict__ rc = a.g.rc;
  const int r_first = rc[e0].x;
  const bool head_partial = a.g.rowptr[r_first] < e0;
  int cur = r_first;
  float4 acc[VEC];
  float dt[VEC], ds[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) { acc[v] = make_float4(0.f, 0.f, 0.f, 0.f); dt[v] = 0.f; ds[v] = 0.f; }

  for (int64_t e = e0; e < e1; e += U) {
    int2 p[U];
    int64_t pe[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t ee = (e + u < e1) ? e + u : e1 - 1;
      p[u] = rc[ee];
      pe[u] = a.perm ? (int64_t)a.perm[ee] : ee;
    }
    float al[U][VEC], dz[U][VEC], dzo[U][VEC];
    float4 gv[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const int64_t ee = (e + u < e1) ? e + u : e1 - 1;
        const float* eb = a.ebuf + pe[u] * 2 * H + lc.head[v];
        al[u][v] = eb[0];
        dz[u][v] = eb[H];
        // ds_j: dz of the forward edge at this same position (symmetric patterns only)
        dzo[u][v] = !a.symmetric ? 0.f : (a.perm ? a.ebuf[ee * 2 * H + H + lc.head[v]] : a.dzf[ee * H + lc.head[v]]);
        gv[u][v] = ld4(a.Gp + (int64_t)p[u].y * R + lc.cofs[v]);
      }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (e + u < e1) {
        if (p[u].x != cur) {
          col_flush<VEC>(a, lc, k, cur, cur == r_first && head_partial, false, acc, dt, ds);
          cur = p[u].x;
#pragma unroll
          for (int v = 0; v < VEC; ++v) { acc[v] = make_float4(0.f, 0.f, 0.f, 0.f); dt[v] = 0.f; ds[v] = 0.f; }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          acc[v].x = fmaf(al[u][v], gv[u][v].x, acc[v].x); acc[v].y = fmaf(al[u][v], gv[u][v].y, acc[v].y);
          acc[v].z = fmaf(al[u][v], gv[u][v].z, acc[v].z); acc[v].w = fmaf(al[u][v], gv[u][v].w, acc[v].w);
          dt[v] += dz[u][v];
          ds[v] += dzo[u][v];
        }
      }
    }
  }
  const bool tail_partial = a.g.rowptr[cur + 1] > e1;
  col_flush<VEC>(a, lc, k, cur, cur == r_first && head_partial, tail_partial, acc, dt, ds);
}

// Fix-up of cut rows (same scheme as gat_fwd_fixup_kernel): a work-group screens FIX_SCREEN slots, the
// owner slot of a cut row is the one where the row starts; its pieces tail(k), head(k+1), ..., head(k_e)
// are summed by the 4 waves x EPW lane groups and combined through LDS in a fixed order.
template <int LPR, int VEC>
__global__ __launch_bounds__(256) void gat_bwd_col_fixup_kernel(ColArgs a) {
  constexpr int EPW = 64 / LPR;
  constexpr int PF = (VEC == 1) ? 4 : 2;
  extern __shared__ __attribute__((aligned(16))) float fix_sm[];  // [4][R + 2H]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t kbase = (int64_t)blockIdx.x * FIX_SCREEN;
  const int64_t nslots = num_slots(a.g);
  int my_r = 0, my_end = 0;
  bool owner = false;
  if (lane < FIX_SCREEN && kbase + lane < nslots) {
    int64_t e0, e1;
    slot_range(a.g, kbase + lane, &e0, &e1);
    my_r = a.g.rc[e1 - 1].x;
    my_end = a.g.rowptr[my_r + 1];
    owner = (int64_t)my_end > e1 && (int64_t)a.g.rowptr[my_r] >= e0;
  }
  unsigned long long todo = __ballot(owner);
  const LaneCols<VEC> lc = lane_cols<LPR, VEC>(a.rs);
  const int slot = lane / LPR;
  const int64_t PS = a.rs.R + 2 * a.rs.H;
  while (todo) {
    const int src = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int64_t k = kbase + src;
    const int r = __shfl(my_r, src);
    const int64_t row_end = __shfl(my_end, src);
    const int64_t k_e = slot_of(a.g, row_end - 1);
    const int npieces = (int)(k_e - k) + 1;
    const bool wide = npieces > EPW * PF;
    float4 acc[VEC];
    float dt[VEC], ds[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { acc[v] = make_float4(0.f, 0.f, 0.f, 0.f); dt[v] = 0.f; ds[v] = 0.f; }
    if (wide || w == 0) {
      const int nw = wide ? 4 : 1;
      for (int q = (wide ? w : 0) * EPW * PF + slot; q < npieces; q += nw * EPW * PF) {
        float4 xp[PF][VEC];
        float tp[PF][VEC], sp[PF][VEC];
#pragma unroll
        for (int f = 0; f < PF; ++f) {
          const int qq = q + f * EPW;
          const int qc = qq < npieces ? qq : q;
          const float* p = a.part + (qc == 0 ? 2 * k + 1 : 2 * (k + qc)) * PS;
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            xp[f][v] = ld4(p + lc.cofs[v]); tp[f][v] = p[a.rs.R + lc.head[v]]; sp[f][v] = p[a.rs.R + a.rs.H + lc.head[v]];
          }
        }
#pragma unroll
        for (int f = 0; f < PF; ++f)
          if (q + f * EPW < npieces) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
              acc[v].x += xp[f][v].x; acc[v].y += xp[f][v].y; acc[v].z += xp[f][v].z; acc[v].w += xp[f][v].w;
              dt[v] += tp[f][v];
              ds[v] += sp[f][v];
            }
          }
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        acc[v] = slot_sum4<LPR>(acc[v]);
        dt[v] = slot_sum<LPR>(dt[v]);
        ds[v] = slot_sum<LPR>(ds[v]);
      }
    }
    if (wide) {
      if (slot == 0) {
        float* p = fix_sm + w * PS;
#pragma unroll
        for (int v = 0; v < VEC; ++v)
          if (lc.valid[v]) {
            st4(p + lc.cofs[v], acc[v]);
            if (((lc.cofs[v] >> 2) & (a.rs.lph - 1)) == 0) { p[a.rs.R + lc.head[v]] = dt[v]; p[a.rs.R + a.rs.H + lc.head[v]] = ds[v]; }
          }
      }
      __syncthreads();
      if (w == 0 && slot == 0) {
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) {
          const float* p = fix_sm + ww * PS;
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const float4 x = ld4(p + lc.cofs[v]);
            acc[v].x += x.x; acc[v].y += x.y; acc[v].z += x.z; acc[v].w += x.w;
            dt[v] += p[a.rs.R + lc.head[v]];
            ds[v] += p[a.rs.R + a.rs.H + lc.head[v]];
          }
        }
      }
      __syncthreads();
    }
    if (w == 0 && slot == 0) col_finish<VEC>(a, lc, r, acc, dt, ds);
  }
}

}  // namespace pygat

using namespace pygat;

extern "C" int pygat_gat_backward_col(const pygat_graph* gT, const int32_t* perm_t, int symmetric, int H, int Fo,
                                      const float* Gp, const float* ebuf, const float* dz_f, const float* a_pad, float* dWh,
                                      float* ds, float* dt, void* part, void* stream) {
  ColArgs a;
  int rc = check_graph(gT, &a.g);
  if (rc) return rc;
  PYGAT_REQUIRE(make_row_shape(H, Fo, &a.rs), "gat_backward_col: unsupported H=%d F'=%d", H, Fo);
  PYGAT_REQUIRE(Gp && ebuf && (dz_f || perm_t || !symmetric) && ds && a_pad && dWh && dt && part, "gat_backward_col: null pointer");
  PYGAT_REQUIRE(aligned16(Gp) && aligned16(dWh) && aligned16(a_pad) && aligned16(part),
                "gat_backward_col: row tables must be 16-byte aligned");
  a.perm = perm_t; a.symmetric = symmetric; a.Gp = Gp; a.ebuf = ebuf; a.dzf = dz_f; a.a_pad = a_pad; a.dWh = dWh; a.ds = ds;
  a.dt = dt; a.part = (float*)part;
  int lpr, vec;
  pick_lanes(a.rs, &lpr, &vec);
  hipStream_t st = (hipStream_t)stream;
  const int64_t nslots = num_slots(a.g);
  const unsigned blocks = (unsigned)cdiv(cdiv(nslots, 64 / lpr), 4);
  PYGAT_DISPATCH_LANES(lpr, vec,
                       hipLaunchKernelGGL((gat_bwd_col_kernel<LPR, VEC>), dim3(blocks), dim3(256), 0, st, a));
  PYGAT_CHECK_LAUNCH("gat_backward_col");
  const size_t fix_lds = 4 * (size_t)(a.rs.R + 2 * a.rs.H) * sizeof(float);
  PYGAT_DISPATCH_LANES(lpr, vec, hipLaunchKernelGGL((gat_bwd_col_fixup_kernel<LPR, VEC>),
                                                    dim3((unsigned)cdiv(nslots, FIX_SCREEN)), dim3(256), fix_lds, st, a));
  PYGAT_CHECK_LAUNCH("gat_backward_col_fixup");
  return PYGAT_OK;
}
