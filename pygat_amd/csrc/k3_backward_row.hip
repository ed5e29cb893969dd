// K3 -- backward, row side (gfx950, wave64).
//
// Replaces SpecialSpmmFunction.backward's grad_values (reference layers.py:84-87), which the
// reference obtains from a DENSE N x N product `grad_output.matmul(b.t())` gathered at the edges,
// and the ATen autograd of layers.py:144-170 (LeakyReLU, exp, row normalisation, ELU).
//
//   K3a (one lane group per row)  Gp_i = G_i * ELU'(pre_i),  D_i = Gp_i . hattn_i  (= sum_j alpha_ij dp_ij)
//                                 GR_i = [ Gp_i (R floats) | (s_i, m_i, 1/Z_i, D_i) per head (4H floats) ]
//   K3b (nnz split over CSR rows) dp_ij = Gp_i . Wh_j   (the (i,j) entry of layers.py:85)
//                                 alpha_ij = exp(e_ij - m_i) / Z_i,  e_ij = LeakyReLU(s_i + t_j), t_j = Wh_j . a_dst
//                                 dz_ij = alpha_ij (mask_ij dp_ij - D_i) * LeakyReLU'(s_i + t_j)
//                                 ds_i  = sum_j dz_ij
// Nothing per-edge is stored: the column pass K4 recomputes alpha and dz from the same two tables
// (GR gathered, Wh row-local there), which costs two dot products per edge but removes the
// [nnz x 2H] hand-off buffer, its scattered re-read and the edge permutation from the hot path.
// In concat mode hattn is not stored by the forward: it is recovered from the output,
// pre = out > 0 ? out : log1p(out), hattn = pre - sk, ELU'(pre) = out > 0 ? 1 : out + 1.
#include "attn_common.h"

namespace pygat {


struct PrepArgs {
  int n;
  RowShape rs;
  int flags, mean_mode;
  const float* G;
  const float* y;   // forward output (concat) or hattn (mean)
  const float* sk;
  const float* s;
  const float* m;
  const float* Z;
  float* GR;        // [n][ldgr]: per backward head window [Gp | 4-float records]  [| copy of Whi (R), GATv2]
  int64_t ldgr;
  int h0p;          // first head covered by this pass, counted inside the head range of the call
  int gr_hg;        // heads per backward window: fixes where a head's Gp and record live inside a GR row
  int gr_heads;     // heads of the range (GR is laid out for them alone)
  const float* whi; // GATv2: table whose first R floats per row are copied behind the rowtab, else nullptr
  int64_t ld_whi;
  // row sums from the forward's alpha-branch shares (k2_forward.hip, AUX): ds_i = -(1 - slope)(Gp_i . aneg_i - D_i qneg_i)
  const float* aneg;  // [n][ldr] or nullptr
  const float* qneg;  // [n][ldh]
  float* ds;          // [n][ldh]
  float slope;        // LeakyReLU alpha
  const int32_t* urow;  // caller's row of internal node i for G (and y in concat mode), or nullptr
};

// float offsets inside a GR row (include/pygat_amd.h, K3a): window w0 = first head of the backward window
__device__ __forceinline__ int64_t gr_gp_off(const PrepArgs& a, int co, int h) {
  const int gh = a.h0p + h;
  return (int64_t)a.h0p * a.rs.Fp + co + 4 * ((gh / a.gr_hg) * a.gr_hg);
}
__device__ __forceinline__ int64_t gr_rt_off(const PrepArgs& a, int h) {
  const int gh = a.h0p + h, w0 = (gh / a.gr_hg) * a.gr_hg;
  const int hcw = (a.gr_heads - w0 < a.gr_hg) ? a.gr_heads - w0 : a.gr_hg;
  return (int64_t)w0 * (a.rs.Fp + 4) + (int64_t)hcw * a.rs.Fp + 4 * (gh - w0);
}

template <int LPR, int VEC>
__global__ __launch_bounds__(256) void gat_bwd_prepare_kernel(PrepArgs a) {
  constexpr int EPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int64_t i = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * EPW + lane / LPR;
  if (i >= a.n) return;  // whole lane groups leave together: the DPP sums below stay inside a group
  const int64_t iu = a.urow ? (int64_t)a.urow[i] : i;   // row of the caller's arrays (G; y in concat mode)
  const LaneCols<VEC> lc = lane_cols<LPR, VEC>(a.rs);
  const int H = a.rs.H, R = a.rs.R, Fo = a.rs.Fo, Fp = a.rs.Fp;
  const int64_t RW = a.ldgr, ldr = a.rs.ldr, ldh = a.rs.ldh, ldo = a.rs.ldo;
  const int lph = a.rs.lph < 64 ? a.rs.lph : 64;
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const int co = lc.cofs[v], h = lc.head[v], f0 = co & (Fp - 1);
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f), hat = g4;
    if (lc.valid[v]) {
      if (a.mean_mode) {
        hat = ld4(a.y + i * ldr + co);
        const float* gr = a.G + iu * Fo + f0;
        const float inv = 1.0f / (float)a.rs.Htot;
        if (f0 + 0 < Fo) g4.x = gr[0] * inv;
        if (f0 + 1 < Fo) g4.y = gr[1] * inv;
        if (f0 + 2 < Fo) g4.z = gr[2] * inv;
        if (f0 + 3 < Fo) g4.w = gr[3] * inv;
      } else {
        float4 y4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (Fo == Fp) {
          g4 = ld4(a.G + iu * ldo + co);
          y4 = ld4(a.y + iu * ldo + co);
        } else {
          const int64_t o = iu * ldo + (int64_t)h * Fo + f0;
          if (f0 + 0 < Fo) { g4.x = a.G[o + 0]; y4.x = a.y[o + 0]; }
          if (f0 + 1 < Fo) { g4.y = a.G[o + 1]; y4.y = a.y[o + 1]; }
          if (f0 + 2 < Fo) { g4.z = a.G[o + 2]; y4.z = a.y[o + 2]; }
          if (f0 + 3 < Fo) { g4.w = a.G[o + 3]; y4.w = a.y[o + 3]; }
        }
        hat = y4;
        if (a.flags & PYGAT_F_ELU) {
          // out = ELU(pre): pre = out > 0 ? out : log1p(out); ELU'(pre) = out > 0 ? 1 : out + 1.
          // out == -1 (pre < -17): the gradient factor is exactly 0, keep hattn finite.
          const float o4[4] = {y4.x, y4.y, y4.z, y4.w};
          float gq[4] = {g4.x, g4.y, g4.z, g4.w}, pq[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float d = o4[q] > 0.f ? 1.f : o4[q] + 1.f;
            pq[q] = o4[q] > 0.f ? o4[q] : (d > 0.f ? __logf(d) : 0.f);  // log(1+out): 1+out is exact near -1, abs error <= 1e-7 near 0
            gq[q] *= d;
          }
          g4 = make_float4(gq[0], gq[1], gq[2], gq[3]);
          hat = make_float4(pq[0], pq[1], pq[2], pq[3]);
        }
        if (a.flags & PYGAT_F_SKIP) {
          const float4 k4 = ld4(a.sk + i * ldr + co);
          hat.x -= k4.x; hat.y -= k4.y; hat.z -= k4.z; hat.w -= k4.w;
        }
      }
      st4(a.GR + i * RW + gr_gp_off(a, co, h), g4);
      if (a.whi) st4(a.GR + i * RW + R + 4 * H + co, ld4(a.whi + i * a.ld_whi + co));
    }
    const float D = group_sum_rt(dot4(g4, hat), lph);
    // (qneg == 0: no edge of the head, or every edge, on the alpha branch -- ds = 0, and K2 may not have written the
    // row of aneg at all: its value is selected away, never multiplied in)
    const float qn = (a.aneg && lc.valid[v]) ? a.qneg[i * ldh + h] : 0.f;
    float gn = 0.f;
    if (a.aneg) gn = group_sum_rt((lc.valid[v] && qn != 0.f) ? dot4(g4, ld4(a.aneg + i * ldr + co)) : 0.f, lph);
    if (lc.valid[v] && ((co >> 2) & (a.rs.lph - 1)) == 0) {
      const int64_t q = i * ldh + h;
      st4(a.GR + i * RW + gr_rt_off(a, h), make_float4(a.s[q], a.m[q], 1.0f / a.Z[q], D));
      if (a.aneg) a.ds[q] = (qn != 0.f) ? (a.slope - 1.f) * (gn - D * qn) : 0.f;
    }
  }
}

// Fast variant for the common layout (concat mode, F' == Fp, one chunk per lane): RB rows per
// lane group with all loads issued up front (the kernel is a pure stream: 2 reads + 1 write).
template <int LPR, int LPH = 0>   // LPH > 0: lanes per head known at compile time
__global__ __launch_bounds__(256) void gat_bwd_prepare_fast_kernel(PrepArgs a) {
  constexpr int EPW = 64 / LPR;
  constexpr int RB = 4;
  const int lane = threadIdx.x & 63;
  const int64_t i0 = (((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * EPW + lane / LPR) * RB;
  if (i0 >= a.n) return;
  const LaneCols<1> lc = lane_cols<LPR, 1>(a.rs);
  const int H = a.rs.H, R = a.rs.R;
  const int64_t RW = a.ldgr, ldr = a.rs.ldr, ldh = a.rs.ldh, ldo = a.rs.ldo;
  const int lph = LPH > 0 ? LPH : (a.rs.lph < 64 ? a.rs.lph : 64);
  const int co = lc.cofs[0], h = lc.head[0];
  const bool valid = lc.valid[0];
  const bool lead = valid && (((co >> 2) & (a.rs.lph - 1)) == 0);
  const int64_t gp_off = gr_gp_off(a, co, h), rt_off = gr_rt_off(a, h);
  float4 g4[RB], y4[RB], k4[RB], n4[RB];
  float sv[RB], mv[RB], zv[RB], qv[RB];
  // qneg goes first (vmcnt counts in issue order): a row whose heads all have qneg == 0 -- no edge, or every edge, on
  // the alpha branch: K2 did not write its aneg row -- skips the 4R-byte read, and the rows that need it issue theirs
  // while G and y are still in flight
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int64_t i = (i0 + r < a.n) ? i0 + r : a.n - 1;
    qv[r] = (a.aneg && valid) ? a.qneg[i * ldh + h] : 0.f;
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int64_t i = (i0 + r < a.n) ? i0 + r : a.n - 1;
    // (non-temporal loads of G / y and stores of Gp were measured, gpurun_out r4j: 0.374-0.379 -> 0.352-0.373 ms, inside the
    // run-to-run spread, and K4 behind it 1.23 -> 1.25: not used)
    const int64_t iu = a.urow ? (int64_t)a.urow[i] : i;
    g4[r] = ld4(a.G + iu * ldo + co);
    y4[r] = ld4(a.y + iu * ldo + co);
    k4[r] = (a.flags & PYGAT_F_SKIP) ? ld4(a.sk + i * ldr + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    sv[r] = a.s[i * ldh + h]; mv[r] = a.m[i * ldh + h]; zv[r] = a.Z[i * ldh + h];
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int64_t i = (i0 + r < a.n) ? i0 + r : a.n - 1;
    n4[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row_any<LPR>(qv[r] != 0.f)) n4[r] = ld4(a.aneg + i * ldr + co);
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int64_t i = i0 + r;
    float gq[4] = {g4[r].x, g4[r].y, g4[r].z, g4[r].w};
    float pq[4] = {y4[r].x, y4[r].y, y4[r].z, y4[r].w};
    if (a.flags & PYGAT_F_ELU) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float o = pq[q];
        const float d = o > 0.f ? 1.f : o + 1.f;
        pq[q] = o > 0.f ? o : (d > 0.f ? __logf(d) : 0.f);
        gq[q] *= d;
      }
    }
    float4 g = valid ? make_float4(gq[0], gq[1], gq[2], gq[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 hat = make_float4(pq[0] - k4[r].x, pq[1] - k4[r].y, pq[2] - k4[r].z, pq[3] - k4[r].w);
    const float D = group_sum_rt(dot4(g, hat), lph);
    const float gn = a.aneg ? group_sum_rt(qv[r] != 0.f ? dot4(g, n4[r]) : 0.f, lph) : 0.f;
    if (i < a.n) {
      if (valid) {
        st4(a.GR + i * RW + gp_off, g);
        if (a.whi) st4(a.GR + i * RW + R + 4 * H + co, ld4(a.whi + i * a.ld_whi + co));
      }
      if (lead) st4(a.GR + i * RW + rt_off, make_float4(sv[r], mv[r], 1.0f / zv[r], D));
      if (lead) {
        if (a.aneg) a.ds[i * ldh + h] = (qv[r] != 0.f) ? (a.slope - 1.f) * (gn - D * qv[r]) : 0.f;
      }
    }
  }
}

// ------------------------------------------------------------------ K3b: ds_i = sum_j dz_ij
struct RowArgs {
  GraphDev g;
  RowShape rs;
  float alpha;
  const float* Wh;
  const float* a_pad;  // t_j = Wh_j . a_dst is recomputed from the gathered row
  const float* GR;
  int64_t ldgr;        // row stride of GR
  const float* mask;   // [nnz][Htot] attention dropout mask (forward edge order) or nullptr
  float* ds;           // [n][Htot]
  float* part;         // [2 * nslots][H]
  // row-sum mode (pygat_gat_backward_rowsum): dz comes from the column pass instead of being recomputed
  const float* dz_t;   // [nnz][Htot], per TRANSPOSED edge (written by the column pass)
  const int32_t* perm_f;  // forward edge -> transposed position
};

template <int VEC>
__device__ __forceinline__ void row_flush(const RowArgs& a, const LaneCols<VEC>& lc, int64_t k, int i,
                                          bool is_head, bool is_tail, const float (&acc)[VEC]) {
  float* dst = (is_head || is_tail) ? a.part + (2 * k + (is_head ? 0 : 1)) * (int64_t)a.rs.H
                                    : a.ds + (int64_t)i * a.rs.ldh;
#pragma unroll
  for (int v = 0; v < VEC; ++v)
    if (lc.valid[v] && (((lc.cofs[v] >> 2) & (a.rs.lph - 1)) == 0)) dst[lc.head[v]] = acc[v];
}

template <int LPR, int VEC>
__global__ __launch_bounds__(256) void gat_bwd_row_kernel(RowArgs a) {
  constexpr int EPW = 64 / LPR;
  constexpr int U = (VEC == 1) ? 4 : 2;
  const int lane = threadIdx.x & 63;
  const int64_t k = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * EPW + lane / LPR;
  if (k >= num_slots(a.g)) return;
  int64_t e0, e1;
  slot_range(a.g, k, &e0, &e1);
  const LaneCols<VEC> lc = lane_cols<LPR, VEC>(a.rs);
  const int R = a.rs.R;
  const int64_t RW = a.ldgr, ldr = a.rs.ldr, ldh = a.rs.ldh;
  const int lph = a.rs.lph < 64 ? a.rs.lph : 64;
  const int2* __restrict__ rc = a.g.rc;
  float4 adst[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    adst[v] = ld4(a.a_pad + (int64_t)lc.head[v] * 2 * a.rs.Fp + a.rs.Fp + (lc.cofs[v] & (a.rs.Fp - 1)));
    if (!lc.valid[v]) adst[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int r_first = rc[e0].x;
  const bool head_partial = a.g.rowptr[r_first] < e0;
  int cur = r_first;
  int rprev = -1;
  float4 gcur[VEC], rtcur[VEC];
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) { acc[v] = 0.f; gcur[v] = make_float4(0.f, 0.f, 0.f, 0.f); rtcur[v] = gcur[v]; }

  for (int64_t e = e0; e < e1; e += U) {
    int2 p[U];
#pragma unroll
    for (int u = 0; u < U; ++u) p[u] = rc[(e + u < e1) ? e + u : e1 - 1];
    float4 rt[U][VEC], wv[U][VEC], gv[U][VEC];
    float mk[U][VEC];
    // the gathers first; the row-local GR_i chunk is only re-fetched when the row id changes inside the
    // batch (edge-weighted, most consecutive edges share their row: the kernel is load-issue bound)
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        wv[u][v] = ld4(a.Wh + (int64_t)p[u].y * ldr + lc.cofs[v]);
        mk[u][v] = a.mask ? a.mask[((e + u < e1) ? e + u : e1 - 1) * ldh + lc.head[v]] : 1.f;
      }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int prev = (u == 0) ? rprev : p[u - 1].x;
      if (p[u].x != prev) {
        const float* gr = a.GR + (int64_t)p[u].x * RW;
#pragma unroll
        for (int v = 0; v < VEC; ++v) { gv[u][v] = ld4(gr + lc.cofs[v]); rt[u][v] = ld4(gr + R + 4 * lc.head[v]); }
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          gv[u][v] = (u == 0) ? gcur[v] : gv[u > 0 ? u - 1 : 0][v];
          rt[u][v] = (u == 0) ? rtcur[v] : rt[u > 0 ? u - 1 : 0][v];
        }
      }
    }
    rprev = p[U - 1].x;
#pragma unroll
    for (int v = 0; v < VEC; ++v) { gcur[v] = gv[U - 1][v]; rtcur[v] = rt[U - 1][v]; }
    // all lanes of the group are active here: the per-head DPP sums are safe
    float dz[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const float dp = group_sum_rt(lc.valid[v] ? dot4(gv[u][v], wv[u][v]) : 0.f, lph);
        const float tj = group_sum_rt(dot4(wv[u][v], adst[v]), lph);
        const float zz = rt[u][v].x + tj;
        const float ev = zz > 0.f ? zz : a.alpha * zz;
        const float al = __expf(ev - rt[u][v].y) * rt[u][v].z;
        dz[u][v] = al * (mk[u][v] * dp - rt[u][v].w) * (zz > 0.f ? 1.f : a.alpha);
      }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (e + u < e1) {
        if (p[u].x != cur) {
          row_flush<VEC>(a, lc, k, cur, cur == r_first && head_partial, false, acc);
          cur = p[u].x;
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += dz[u][v];
      }
    }
  }
  const bool tail_partial = a.g.rowptr[cur + 1] > e1;
  row_flush<VEC>(a, lc, k, cur, cur == r_first && head_partial, tail_partial, acc);
}

// Row-sum pass: ds_i = sum over the forward edges k of row i of dz_t[perm_f[k]] -- the light replacement of the
// kernel above when the column pass has already written every dz (4H-byte records instead of a gathered Wh row).
// Same slots, same partial records and fix-up kernels as the row pass.
template <int LPR, int VEC>
__global__ __launch_bounds__(256) void gat_bwd_rowsum_kernel(RowArgs a) {
  constexpr int EPW = 64 / LPR;
  constexpr int U = 16;
  const int lane = threadIdx.x & 63;
  const int64_t k = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * EPW + lane / LPR;
  if (k >= num_slots(a.g)) return;
  int64_t e0, e1;
  slot_range(a.g, k, &e0, &e1);
  const LaneCols<VEC> lc = lane_cols<LPR, VEC>(a.rs);
  const int64_t ldh = a.rs.ldh;
  const int2* __restrict__ rc = a.g.rc;
  const int r_first = rc[e0].x;
  const bool head_partial = a.g.rowptr[r_first] < e0;
  int cur = r_first;
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
  for (int64_t e = e0; e < e1; e += U) {
    int row[U];
    float dz[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t ee = (e + u < e1) ? e + u : e1 - 1;
      row[u] = rc[ee].x;
      const int64_t q = a.perm_f[ee];
#pragma unroll
      for (int v = 0; v < VEC; ++v) dz[u][v] = a.dz_t[q * ldh + lc.head[v]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (e + u < e1) {
        if (row[u] != cur) {
          row_flush<VEC>(a, lc, k, cur, cur == r_first && head_partial, false, acc);
          cur = row[u];
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += dz[u][v];
      }
    }
  }
  const bool tail_partial = a.g.rowptr[cur + 1] > e1;
  row_flush<VEC>(a, lc, k, cur, cur == r_first && head_partial, tail_partial, acc);
}

// cut rows: one thread per (slot, head) screens ownership and sums the pieces in slot order
__global__ __launch_bounds__(256) void gat_bwd_row_fixup_kernel(RowArgs a) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int H = a.rs.H;
  const int64_t k = idx / H;
  const int h = (int)(idx % H);
  if (k >= num_slots(a.g)) return;
  int64_t e0, e1;
  slot_range(a.g, k, &e0, &e1);
  const int r = a.g.rc[e1 - 1].x;
  const int64_t row_end = a.g.rowptr[r + 1];
  if (row_end <= e1 || (int64_t)a.g.rowptr[r] < e0) return;
  const int64_t k_e = slot_of(a.g, row_end - 1);
  float acc = a.part[(2 * k + 1) * H + h];
  for (int64_t kk = k + 1; kk <= k_e; ++kk) acc += a.part[(2 * kk) * H + h];
  a.ds[(int64_t)r * a.rs.ldh + h] = acc;
}

// list-driven variant: one wave per cut row; lane l sums pieces l, l+64, ... of each head, then a fixed
// butterfly adds the 64 partial sums (a 26k-edge row has 400+ pieces: a single thread would walk them serially)
__global__ __launch_bounds__(256) void gat_bwd_row_fixup_list_kernel(RowArgs a) {
  const int q0 = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q0 >= a.g.n_cut) return;
  const int lane = threadIdx.x & 63;
  const int64_t k = a.g.cut[3 * q0];
  const int r = a.g.cut[3 * q0 + 1];
  const int npieces = a.g.cut[3 * q0 + 2];
  const int H = a.rs.H;
  for (int h = 0; h < H; ++h) {
    float acc = 0.f;
    for (int q = lane; q < npieces; q += 64) acc += a.part[(q == 0 ? 2 * k + 1 : 2 * (k + q)) * (int64_t)H + h];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) a.ds[(int64_t)r * a.rs.ldh + h] = acc;
  }
}

}  // namespace pygat

using namespace pygat;

static int launch_prepare(int n, int H, int Fo, int flags, int mean_mode, const float* G, const float* y,
                          const float* sk, const float* s, const float* m, const float* Z, float* GR,
                          const float* whi, int64_t ld_whi, const float* aneg, const float* qneg, float slope, float* ds,
                          int h_first, int h_count, int head_group, void* stream, const int32_t* user_row = nullptr);

extern "C" int pygat_gat_backward_prepare(int n, int H, int Fo, int flags, int mean_mode, const float* G,
                                          const float* y, const float* sk, const float* s, const float* m,
                                          const float* Z, float* GR, const float* aneg, const float* qneg, float alpha,
                                          float* ds, int h_first, int h_count, int head_group, const int32_t* user_row,
                                          void* stream) {
  return launch_prepare(n, H, Fo, flags, mean_mode, G, y, sk, s, m, Z, GR, nullptr, 0, aneg, qneg, alpha, ds, h_first,
                        h_count, head_group, stream, user_row);
}

/* GATv2: GRW [n x (2R + 4H)] = [Gp | (., m, 1/Z, D) | Whi], Whi copied from WW [n x 2R] */
extern "C" int pygat_gatv2_backward_prepare(int n, int H, int Fo, int flags, int mean_mode, const float* G,
                                            const float* y, const float* sk, const float* m, const float* Z,
                                            const float* WW, float* GRW, const int32_t* user_row, void* stream) {
  if (!WW) { pygat::set_error("gatv2_backward_prepare: null WW"); return PYGAT_EINVAL; }
  int Fp = pygat::padded_width(Fo);
  return launch_prepare(n, H, Fo, flags, mean_mode, G, y, sk, m /* s slot unused in V2 */, m, Z, GRW, WW,
                        2 * (int64_t)H * Fp, nullptr, nullptr, 0.f, nullptr, 0, 0, 0, stream, user_row);
}

static int launch_prepare(int n, int H, int Fo, int flags, int mean_mode, const float* G, const float* y,
                          const float* sk, const float* s, const float* m, const float* Z, float* GR,
                          const float* whi, int64_t ld_whi, const float* aneg, const float* qneg, float slope, float* ds,
                          int h_first, int h_count, int head_group, void* stream, const int32_t* user_row) {
  PrepArgs a;
  a.urow = user_row;
  const int Fp = padded_width(Fo);
  HeadRange rg;
  PYGAT_REQUIRE(H > 0 && Fp > 0, "gat_backward_prepare: unsupported H=%d F'=%d", H, Fo);
  PYGAT_REQUIRE(make_head_range(H, h_first, h_count, &rg) && (!whi || rg.hr == H),
                "gat_backward_prepare: bad head range [%d, +%d) of %d", h_first, h_count, H);
  PYGAT_REQUIRE(n > 0 && G && y && s && m && Z && GR, "gat_backward_prepare: null pointer");
  PYGAT_REQUIRE((aneg == nullptr) == (qneg == nullptr) && (aneg == nullptr) == (ds == nullptr) && (!aneg || aligned16(aneg)),
                "gat_backward_prepare: aneg, qneg and ds come together (aneg 16-byte aligned)");
  PYGAT_REQUIRE(!(flags & PYGAT_F_SKIP) || sk, "gat_backward_prepare: PYGAT_F_SKIP without sk");
  PYGAT_REQUIRE(!(mean_mode && (flags & PYGAT_F_ELU)), "gat_backward_prepare: the head mean never carries an ELU (models.py:23)");
  PYGAT_REQUIRE(aligned16(GR) && (!sk || aligned16(sk)) &&
                    (mean_mode ? aligned16(y) : (Fo != Fp || (aligned16(G) && aligned16(y)))),
                "gat_backward_prepare: row tables must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int hg = whi ? H : head_group_fwd(rg.hr, Fp);   // kernel passes (GATv2 is not windowed)
  const int gr_hg = head_group_arg(head_group, n, rg.hr, Fp);   // heads per window of the GR layout
  PYGAT_REQUIRE(gr_hg > 0, "gat_backward_prepare: head_group=%d gives rows of more than 1024 floats per pass", head_group);
  for (int h0 = 0; h0 < rg.hr; h0 += hg) {
    const int hc = (rg.hr - h0 < hg) ? rg.hr - h0 : hg;
    const int gh = rg.hb + h0;                             // first head of the pass inside the level
    PYGAT_REQUIRE(make_window_shape(H, Fo, hc, &a.rs), "gat_backward_prepare: unsupported H=%d F'=%d", hc, Fo);
    a.n = n; a.flags = flags; a.mean_mode = mean_mode;
    // mean mode: G is [n, F'] (shared by the heads), y = hattn [n, R]; concat: G and y = out are [n, H*F']
    a.G = mean_mode ? G : G + (int64_t)gh * Fo;
    a.y = mean_mode ? y + (int64_t)gh * Fp : y + (int64_t)gh * Fo;
    a.sk = sk ? sk + (int64_t)gh * Fp : nullptr;
    a.s = s + gh; a.m = m + gh; a.Z = Z + gh;
    a.ldgr = (int64_t)rg.hr * Fp * (whi ? 2 : 1) + 4 * rg.hr;
    a.GR = GR; a.h0p = h0; a.gr_hg = whi ? H : gr_hg; a.gr_heads = rg.hr;
    a.whi = whi; a.ld_whi = ld_whi;
    a.aneg = aneg ? aneg + (int64_t)gh * Fp : nullptr; a.qneg = qneg ? qneg + gh : nullptr;
    a.ds = ds ? ds + gh : nullptr; a.slope = slope;
    int lpr, vec;
    pick_lanes(a.rs, &lpr, &vec);
    if (!mean_mode && a.rs.Fo == a.rs.Fp && vec == 1) {
      const unsigned fb = (unsigned)cdiv(cdiv(cdiv(n, 4), 64 / lpr), 4);
      switch (lpr) {
        case 1: hipLaunchKernelGGL((gat_bwd_prepare_fast_kernel<1>), dim3(fb), dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL((gat_bwd_prepare_fast_kernel<2>), dim3(fb), dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL((gat_bwd_prepare_fast_kernel<4>), dim3(fb), dim3(256), 0, st, a); break;
        case 8: hipLaunchKernelGGL((gat_bwd_prepare_fast_kernel<8>), dim3(fb), dim3(256), 0, st, a); break;
        case 16: hipLaunchKernelGGL((gat_bwd_prepare_fast_kernel<16>), dim3(fb), dim3(256), 0, st, a); break;
        case 32:
          if (a.rs.lph == 4) hipLaunchKernelGGL((gat_bwd_prepare_fast_kernel<32, 4>), dim3(fb), dim3(256), 0, st, a);
          else hipLaunchKernelGGL((gat_bwd_prepare_fast_kernel<32>), dim3(fb), dim3(256), 0, st, a);
          break;
        default: hipLaunchKernelGGL((gat_bwd_prepare_fast_kernel<64>), dim3(fb), dim3(256), 0, st, a); break;
      }
    } else {
      const unsigned blocks = (unsigned)cdiv(cdiv(n, 64 / lpr), 4);
      PYGAT_DISPATCH_LANES(lpr, vec, hipLaunchKernelGGL((gat_bwd_prepare_kernel<LPR, VEC>), dim3(blocks), dim3(256), 0,
                                                        st, a));
    }
    PYGAT_CHECK_LAUNCH("gat_backward_prepare");
  }
  return PYGAT_OK;
}

extern "C" int pygat_gat_backward_row(const pygat_graph* g, int H, int Fo, float alpha, const float* Wh,
                                      const float* a_pad, const float* GR, const float* att_mask, float* ds,
                                      void* part, int h_first, int h_count, int head_group, void* stream) {
  RowArgs a;
  int rc = check_graph(g, &a.g);
  if (rc) return rc;
  const int Fp = padded_width(Fo);
  HeadRange rg;
  PYGAT_REQUIRE(H > 0 && Fp > 0, "gat_backward_row: unsupported H=%d F'=%d", H, Fo);
  PYGAT_REQUIRE(make_head_range(H, h_first, h_count, &rg), "gat_backward_row: bad head range [%d, +%d) of %d", h_first, h_count, H);
  PYGAT_REQUIRE(Wh && a_pad && GR && ds && part, "gat_backward_row: null pointer");
  PYGAT_REQUIRE(aligned16(Wh) && aligned16(GR) && aligned16(a_pad), "gat_backward_row: row tables must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t nslots = num_slots(a.g);
  const int hg = head_group_arg(head_group, a.g.n, rg.hr, Fp);
  PYGAT_REQUIRE(hg > 0, "gat_backward_row: head_group=%d gives rows of more than 1024 floats per pass", head_group);
  for (int h0 = 0; h0 < rg.hr; h0 += hg) {
    const int hc = (rg.hr - h0 < hg) ? rg.hr - h0 : hg;
    const int gh = rg.hb + h0;
    PYGAT_REQUIRE(make_window_shape(H, Fo, hc, &a.rs), "gat_backward_row: unsupported H=%d F'=%d", hc, Fo);
    a.alpha = alpha; a.Wh = Wh + (int64_t)gh * Fp; a.a_pad = a_pad + (int64_t)gh * 2 * Fp;
    a.GR = GR + gr_window_offset(h0, Fp); a.ldgr = (int64_t)rg.hr * Fp + 4 * rg.hr;
    a.mask = att_mask ? att_mask + gh : nullptr; a.ds = ds + gh; a.part = (float*)part;
    int lpr, vec;
    pick_lanes(a.rs, &lpr, &vec);
    const unsigned blocks = (unsigned)cdiv(cdiv(nslots, 64 / lpr), 4);
    PYGAT_DISPATCH_LANES(lpr, vec, hipLaunchKernelGGL((gat_bwd_row_kernel<LPR, VEC>), dim3(blocks), dim3(256), 0, st, a));
    PYGAT_CHECK_LAUNCH("gat_backward_row");
    if (a.g.cut) {
      if (a.g.n_cut > 0)
        hipLaunchKernelGGL(gat_bwd_row_fixup_list_kernel, dim3((unsigned)cdiv(a.g.n_cut, 4)), dim3(256), 0, st, a);
    } else {
      hipLaunchKernelGGL(gat_bwd_row_fixup_kernel, dim3((unsigned)cdiv(nslots * hc, 256)), dim3(256), 0, st, a);
    }
    PYGAT_CHECK_LAUNCH("gat_backward_row_fixup");
  }
  return PYGAT_OK;
}

/* ds_i = sum_j dz_ij from the per-edge records of the column pass.  g = forward pattern, perm_f[k] = transposed
 * position of forward edge k.  (The ds_i a_src term that pass left out of dWh_i is added by pygat_a_grad.) */
extern "C" int pygat_gat_backward_rowsum(const pygat_graph* g, const int32_t* perm_f, int H, int Fo, const float* dz_t,
                                         float* ds, void* part, int h_first, int h_count, void* stream) {
  RowArgs a;
  int rc = check_graph(g, &a.g);
  if (rc) return rc;
  const int Fp = padded_width(Fo);
  HeadRange rg;
  PYGAT_REQUIRE(H > 0 && Fp > 0, "gat_backward_rowsum: unsupported H=%d F'=%d", H, Fo);
  PYGAT_REQUIRE(make_head_range(H, h_first, h_count, &rg), "gat_backward_rowsum: bad head range [%d, +%d) of %d", h_first, h_count, H);
  PYGAT_REQUIRE(perm_f && dz_t && ds && part, "gat_backward_rowsum: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int64_t nslots = num_slots(a.g);
  // the records are 4 bytes per head: walk all heads of a row with one lane per head (padded width 4)
  PYGAT_REQUIRE(make_row_shape(rg.hr, 4, &a.rs), "gat_backward_rowsum: too many heads (%d)", rg.hr);
  a.rs.Htot = H; a.rs.ldh = H;          // ds and the records keep the level's width
  a.alpha = 0.f; a.Wh = nullptr; a.a_pad = nullptr; a.GR = nullptr; a.ldgr = 0; a.mask = nullptr;
  a.ds = ds + rg.hb; a.part = (float*)part; a.dz_t = dz_t + rg.hb; a.perm_f = perm_f;
  int lpr, vec;
  pick_lanes(a.rs, &lpr, &vec);
  const unsigned blocks = (unsigned)cdiv(cdiv(nslots, 64 / lpr), 4);
  PYGAT_DISPATCH_LANES(lpr, vec, hipLaunchKernelGGL((gat_bwd_rowsum_kernel<LPR, VEC>), dim3(blocks), dim3(256), 0, st, a));
  PYGAT_CHECK_LAUNCH("gat_backward_rowsum");
  if (a.g.cut) {
    if (a.g.n_cut > 0)
      hipLaunchKernelGGL(gat_bwd_row_fixup_list_kernel, dim3((unsigned)cdiv(a.g.n_cut, 4)), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(gat_bwd_row_fixup_kernel, dim3((unsigned)cdiv(nslots * rg.hr, 256)), dim3(256), 0, st, a);
  }
  PYGAT_CHECK_LAUNCH("gat_backward_rowsum_fixup");
  return PYGAT_OK;
}
