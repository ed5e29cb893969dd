// K3 / K4 -- backward of the fused edge-softmax + aggregation (gfx950, wave64).
//
// Replaces SpecialSpmmFunction.backward (reference layers.py:81-90), whose
// grad_values come from a DENSE N x N product `grad_output.matmul(b.t())`
// (layers.py:85) gathered at the edges (layers.py:86-87), and the ATen autograd
// of layers.py:141-170.  Per edge instead, no N x N, no float atomics:
//
//   K3 (row pass, CSR of A):     Gp_i  = G_i * ELU'(hattn_i + sk_i)
//                                D_i   = Gp_i . hattn_i            (= sum_j alpha_ij dp_ij)
//                                dp_ij = Gp_i . Wh_j               (the (i,j) entry of layers.py:85)
//                                alpha_ij = exp(e_ij - m_i) / Z_i
//                                dz_ij = alpha_ij (dp_ij - D_i) * LeakyReLU'(s_i + t_j)
//                                ds_i  = sum_j dz_ij
//   K4 (column pass, CSR of A^T with the edge permutation; layers.py:89 `a.t().matmul(grad)`):
//                                dWh_j = sum_i alpha_ij Gp_i + ds_j a_src + dt_j a_dst
//                                dt_j  = sum_i dz_ij
// alpha and dz travel from K3 to K4 through ebuf [nnz][2][H].
#include "attn_common.h"

namespace pygat {

struct BwdRowArgs {
  GraphDev g;
  RowShape rs;
  float alpha;
  int flags;
  int mean_mode;
  const float* G;
  const float* Wh;
  const float* s;
  const float* t;
  const float* sk;
  const float* hattn;
  const float* m;
  const float* Z;
  float* Gp;
  float* ebuf;
  float* ds;
  float* part;  // [n_items][R + 2H]; K3 uses the first H floats of a record
};

struct BwdColArgs {
  GraphDev g;  // transposed pattern
  RowShape rs;
  const int32_t* perm;
  const float* Gp;
  const float* ebuf;
  const float* ds;
  const float* a_pad;
  float* dWh;
  float* dt;
  float* part;  // [n_items][R + 2H]: acc[R], dt[H]
};

// ------------------------------------------------------------------------- K3
template <int LPR, int VEC>
__device__ __forceinline__ void bwd_row_range(const BwdRowArgs& a, int i, int e0, int e1, bool first_item,
                                              float* ds_dst /* where ds of this range goes, indexed by head */) {
  constexpr int EPW = 64 / LPR;
  constexpr int U = 2;
  const LaneCols<VEC> lc = lane_cols<LPR, VEC>(a.rs);
  const int slot = (threadIdx.x & 63) / LPR;
  const int H = a.rs.H, R = a.rs.R, Fo = a.rs.Fo, Fp = a.rs.Fp;
  const int lph = a.rs.lph < 64 ? a.rs.lph : 64;

  float4 gp[VEC];
  float D[VEC], si[VEC], mi[VEC], rz[VEC], dsacc[VEC];
  bool head_lead[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const int co = lc.cofs[v], h = lc.head[v];
    float4 hat = ld4(a.hattn + (int64_t)i * R + co);
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int f0 = co & (Fp - 1);
    if (a.mean_mode) {
      const float* gr = a.G + (int64_t)i * Fo + f0;
      const float inv = 1.0f / (float)H;
      if (f0 + 0 < Fo) g4.x = gr[0] * inv;
      if (f0 + 1 < Fo) g4.y = gr[1] * inv;
      if (f0 + 2 < Fo) g4.z = gr[2] * inv;
      if (f0 + 3 < Fo) g4.w = gr[3] * inv;
    } else if (Fo == Fp) {
      g4 = ld4(a.G + (int64_t)i * R + co);
    } else {
      const float* gr = a.G + (int64_t)i * H * Fo + (int64_t)h * Fo + f0;
      if (f0 + 0 < Fo) g4.x = gr[0];
      if (f0 + 1 < Fo) g4.y = gr[1];
      if (f0 + 2 < Fo) g4.z = gr[2];
      if (f0 + 3 < Fo) g4.w = gr[3];
    }
    if (a.flags & PYGAT_F_ELU) {
      float4 pre = hat;
      if (a.flags & PYGAT_F_SKIP) {
        float4 k4 = ld4(a.sk + (int64_t)i * R + co);
        pre.x += k4.x; pre.y += k4.y; pre.z += k4.z; pre.w += k4.w;
      }
      g4.x *= pre.x > 0.f ? 1.f : expf(pre.x);
      g4.y *= pre.y > 0.f ? 1.f : expf(pre.y);
      g4.z *= pre.z > 0.f ? 1.f : expf(pre.z);
      g4.w *= pre.w > 0.f ? 1.f : expf(pre.w);
    }
    if (!lc.valid[v]) g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    gp[v] = g4;
    if (first_item && slot == 0 && lc.valid[v]) st4(a.Gp + (int64_t)i * R + co, g4);
    D[v] = group_sum_rt(dot4(g4, hat), lph);
    si[v] = a.s[(int64_t)i * H + h];
    mi[v] = a.m[(int64_t)i * H + h];
    rz[v] = 1.0f / a.Z[(int64_t)i * H + h];
    dsacc[v] = 0.f;
    head_lead[v] = lc.valid[v] && (((co >> 2) & (a.rs.lph - 1)) == 0);
  }
  // Fp = 256 with several chunks per head cannot happen (lph = 64 = one chunk per lane per head).

  // wave-uniform trip count: every lane stays active so the DPP reductions see their whole group
  const int iters = (e1 - e0 + U * EPW - 1) / (U * EPW);
  for (int it = 0; it < iters; ++it) {
    int e[U], j[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      e[u] = e0 + (it * U + u) * EPW + slot;
      ok[u] = e[u] < e1;
      j[u] = a.g.col[ok[u] ? e[u] : e1 - 1];
    }
    float tv[U][VEC];
    float4 wv[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        tv[u][v] = a.t[(int64_t)j[u] * H + lc.head[v]];
        wv[u][v] = ld4(a.Wh + (int64_t)j[u] * R + lc.cofs[v]);
      }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const float zz = si[v] + tv[u][v];
        const float ev = zz > 0.f ? zz : a.alpha * zz;
        const float al = __expf(ev - mi[v]) * rz[v];
        const float dp = group_sum_rt(dot4(gp[v], wv[u][v]), lph);
        float dz = al * (dp - D[v]) * (zz > 0.f ? 1.f : a.alpha);
        if (ok[u]) {
          dsacc[v] += dz;
          if (head_lead[v]) {
            float* eb = a.ebuf + (int64_t)e[u] * 2 * H + lc.head[v];
            eb[0] = al;
            eb[H] = dz;
          }
        }
      }
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    float tot = slot_sum<LPR>(dsacc[v]);
    if (slot == 0 && head_lead[v]) ds_dst[lc.head[v]] = tot;
  }
}

template <int LPR, int VEC>
__global__ __launch_bounds__(256) void gat_bwd_row_kernel(BwdRowArgs a) {
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gw < a.g.n) {
    const int e0 = a.g.rowptr[gw], e1 = a.g.rowptr[gw + 1];
    if (e1 - e0 > a.g.chunk || e1 == e0) return;
    bwd_row_range<LPR, VEC>(a, gw, e0, e1, true, a.ds + (int64_t)gw * a.rs.H);
  } else {
    const int it = gw - a.g.n;
    if (it >= a.g.n_items) return;
    const int i = a.g.heavy_row[a.g.item_row_slot[it]];
    const int e0 = a.g.item_begin[it];
    bwd_row_range<LPR, VEC>(a, i, e0, a.g.item_end[it], e0 == a.g.rowptr[i],
                            a.part + (int64_t)it * (a.rs.R + 2 * a.rs.H));
  }
}

// ds of heavy rows: sum item partials in item order; one thread per (heavy row, head)
__global__ __launch_bounds__(256) void gat_bwd_row_combine_kernel(BwdRowArgs a) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.g.n_heavy * a.rs.H) return;
  const int hr = idx / a.rs.H, h = idx % a.rs.H;
  const int PS = a.rs.R + 2 * a.rs.H;
  float acc = 0.f;
  for (int it = a.g.heavy_item_ptr[hr]; it < a.g.heavy_item_ptr[hr + 1]; ++it) acc += a.part[(int64_t)it * PS + h];
  a.ds[(int64_t)a.g.heavy_row[hr] * a.rs.H + h] = acc;
}

// ------------------------------------------------------------------------- K4
template <int LPR, int VEC>
__device__ __forceinline__ void bwd_col_range(const BwdColArgs& a, const LaneCols<VEC>& lc, int e0, int e1,
                                              float4 (&acc)[VEC], float (&dt)[VEC]) {
  constexpr int EPW = 64 / LPR;
  constexpr int U = (VEC == 1) ? 4 : 2;
  const int slot = (threadIdx.x & 63) / LPR;
  const int H = a.rs.H, R = a.rs.R;
#pragma unroll
  for (int v = 0; v < VEC; ++v) { acc[v] = make_float4(0.f, 0.f, 0.f, 0.f); dt[v] = 0.f; }
  int e = e0 + slot;
  for (; e + (U - 1) * EPW < e1; e += U * EPW) {
    int i[U], pe[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { i[u] = a.g.col[e + u * EPW]; pe[u] = a.perm[e + u * EPW]; }
    float al[U][VEC], dz[U][VEC];
    float4 gv[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const float* eb = a.ebuf + (int64_t)pe[u] * 2 * H + lc.head[v];
        al[u][v] = eb[0];
        dz[u][v] = eb[H];
        gv[u][v] = ld4(a.Gp + (int64_t)i[u] * R + lc.cofs[v]);
      }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        acc[v].x = fmaf(al[u][v], gv[u][v].x, acc[v].x); acc[v].y = fmaf(al[u][v], gv[u][v].y, acc[v].y);
        acc[v].z = fmaf(al[u][v], gv[u][v].z, acc[v].z); acc[v].w = fmaf(al[u][v], gv[u][v].w, acc[v].w);
        dt[v] += dz[u][v];
      }
  }
  for (; e < e1; e += EPW) {
    const int i = a.g.col[e], pe = a.perm[e];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const float* eb = a.ebuf + (int64_t)pe * 2 * H + lc.head[v];
      const float al = eb[0], dz = eb[H];
      const float4 g = ld4(a.Gp + (int64_t)i * R + lc.cofs[v]);
      acc[v].x = fmaf(al, g.x, acc[v].x); acc[v].y = fmaf(al, g.y, acc[v].y);
      acc[v].z = fmaf(al, g.z, acc[v].z); acc[v].w = fmaf(al, g.w, acc[v].w);
      dt[v] += dz;
    }
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    acc[v] = slot_sum4<LPR>(acc[v]);
    dt[v] = slot_sum<LPR>(dt[v]);
  }
}

template <int LPR, int VEC>
__device__ __forceinline__ void bwd_col_finish(const BwdColArgs& a, const LaneCols<VEC>& lc, int j,
                                               const float4 (&acc)[VEC], const float (&dt)[VEC]) {
  if ((threadIdx.x & 63) / LPR != 0) return;
  const int H = a.rs.H, R = a.rs.R, Fp = a.rs.Fp;
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    if (!lc.valid[v]) continue;
    const int co = lc.cofs[v], h = lc.head[v], f0 = co & (Fp - 1);
    const float dsj = a.ds[(int64_t)j * H + h];
    const float4 as = ld4(a.a_pad + (int64_t)h * 2 * Fp + f0);
    const float4 ad = ld4(a.a_pad + (int64_t)h * 2 * Fp + Fp + f0);
    float4 o;
    o.x = acc[v].x + dsj * as.x + dt[v] * ad.x;
    o.y = acc[v].y + dsj * as.y + dt[v] * ad.y;
    o.z = acc[v].z + dsj * as.z + dt[v] * ad.z;
    o.w = acc[v].w + dsj * as.w + dt[v] * ad.w;
    st4(a.dWh + (int64_t)j * R + co, o);
    if (((co >> 2) & (a.rs.lph - 1)) == 0) a.dt[(int64_t)j * H + h] = dt[v];
  }
}

template <int LPR, int VEC>
__global__ __launch_bounds__(256) void gat_bwd_col_kernel(BwdColArgs a) {
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const LaneCols<VEC> lc = lane_cols<LPR, VEC>(a.rs);
  float4 acc[VEC];
  float dt[VEC];
  if (gw < a.g.n) {
    const int e0 = a.g.rowptr[gw], e1 = a.g.rowptr[gw + 1];
    if (e1 - e0 > a.g.chunk) return;
    bwd_col_range<LPR, VEC>(a, lc, e0, e1, acc, dt);
    bwd_col_finish<LPR, VEC>(a, lc, gw, acc, dt);
  } else {
    const int it = gw - a.g.n;
    if (it >= a.g.n_items) return;
    bwd_col_range<LPR, VEC>(a, lc, a.g.item_begin[it], a.g.item_end[it], acc, dt);
    if ((threadIdx.x & 63) / LPR != 0) return;
    float* p = a.part + (int64_t)it * (a.rs.R + 2 * a.rs.H);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      if (!lc.valid[v]) continue;
      st4(p + lc.cofs[v], acc[v]);
      if (((lc.cofs[v] >> 2) & (a.rs.lph - 1)) == 0) p[a.rs.R + lc.head[v]] = dt[v];
    }
  }
}

template <int LPR, int VEC>
__global__ __launch_bounds__(256) void gat_bwd_col_combine_kernel(BwdColArgs a) {
  const int hr = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (hr >= a.g.n_heavy) return;
  const LaneCols<VEC> lc = lane_cols<LPR, VEC>(a.rs);
  float4 acc[VEC];
  float dt[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) { acc[v] = make_float4(0.f, 0.f, 0.f, 0.f); dt[v] = 0.f; }
  const int PS = a.rs.R + 2 * a.rs.H;
  for (int it = a.g.heavy_item_ptr[hr]; it < a.g.heavy_item_ptr[hr + 1]; ++it) {
    const float* p = a.part + (int64_t)it * PS;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float4 q = ld4(p + lc.cofs[v]);
      acc[v].x += q.x; acc[v].y += q.y; acc[v].z += q.z; acc[v].w += q.w;
      dt[v] += p[a.rs.R + lc.head[v]];
    }
  }
  bwd_col_finish<LPR, VEC>(a, lc, a.g.heavy_row[hr], acc, dt);
}

// ------------------------------------------------------------------- da reduction
// da_src[h][f] = sum_i ds[i][h] Wh[i][h*Fp+f], da_dst likewise with dt (layers.py:60-61 autograd).
constexpr int AG_BLOCKS = 512;

__global__ __launch_bounds__(256) void a_grad_partial_kernel(int n, RowShape rs, const float* __restrict__ Wh,
                                                             const float* __restrict__ ds,
                                                             const float* __restrict__ dt,
                                                             float* __restrict__ ws) {
  // TPR threads per row (power of two >= NCH, <= 256), 256/TPR rows in flight per block
  int tpr = 1;
  while (tpr < rs.NCH) tpr <<= 1;
  const int rpb = 256 / tpr;
  const int c = threadIdx.x % tpr, rg = threadIdx.x / tpr;
  const bool valid = c < rs.NCH;
  const int co = valid ? 4 * c : 0, h = co >> rs.fp_shift;
  const int64_t rows_per_block = cdiv(n, AG_BLOCKS);
  const int64_t r0 = blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < n) ? r0 + rows_per_block : n;
  float4 as = make_float4(0.f, 0.f, 0.f, 0.f), ad = as;
  for (int64_t i = r0 + rg; i < r1; i += rpb) {
    const float4 w = ld4(Wh + i * rs.R + co);
    const float a1 = ds[i * rs.H + h], a2 = dt[i * rs.H + h];
    as.x = fmaf(a1, w.x, as.x); as.y = fmaf(a1, w.y, as.y); as.z = fmaf(a1, w.z, as.z); as.w = fmaf(a1, w.w, as.w);
    ad.x = fmaf(a2, w.x, ad.x); ad.y = fmaf(a2, w.y, ad.y); ad.z = fmaf(a2, w.z, ad.z); ad.w = fmaf(a2, w.w, ad.w);
  }
  __shared__ float4 sm[2][256];
  sm[0][threadIdx.x] = as;
  sm[1][threadIdx.x] = ad;
  __syncthreads();
  if (rg == 0 && valid) {
    for (int g = 1; g < rpb; ++g) {  // fixed order: deterministic
      float4 q = sm[0][g * tpr + c], r = sm[1][g * tpr + c];
      as.x += q.x; as.y += q.y; as.z += q.z; as.w += q.w;
      ad.x += r.x; ad.y += r.y; ad.z += r.z; ad.w += r.w;
    }
    float* o = ws + (int64_t)blockIdx.x * 2 * rs.R;
    st4(o + co, as);
    st4(o + rs.R + co, ad);
  }
}

__global__ __launch_bounds__(256) void a_grad_final_kernel(RowShape rs, const float* __restrict__ ws,
                                                           float* __restrict__ da) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // over H * 2 * Fo
  if (idx >= rs.H * 2 * rs.Fo) return;
  const int h = idx / (2 * rs.Fo), r = idx % (2 * rs.Fo);
  const int which = r / rs.Fo, f = r % rs.Fo;
  float acc = 0.f;
  for (int b = 0; b < AG_BLOCKS; ++b) acc += ws[(int64_t)b * 2 * rs.R + which * rs.R + h * rs.Fp + f];
  da[idx] = acc;
}

// ------------------------------------------------------------- parameter packing
__global__ __launch_bounds__(256) void pack_params_kernel(int H, int Fin, int Fo, int Fp,
                                                          const float* __restrict__ W,
                                                          const float* __restrict__ a,
                                                          const float* __restrict__ w_skip,
                                                          float* __restrict__ Wcat, int64_t ldw,
                                                          float* __restrict__ a_pad) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int R = H * Fp, Rs = w_skip ? R : 0;
  if (idx < (int64_t)H * 2 * Fp) {  // a_pad[h][which][f]
    const int h = (int)(idx / (2 * Fp)), r = (int)(idx % (2 * Fp)), which = r / Fp, f = r % Fp;
    a_pad[idx] = f < Fo ? a[(int64_t)h * 2 * Fo + which * Fo + f] : 0.f;
  }
  if (idx >= (int64_t)Fin * ldw) return;
  const int k = (int)(idx / ldw), col = (int)(idx % ldw);
  float v = 0.f;
  if (col < R) {
    const int h = col / Fp, f = col % Fp;
    if (f < Fo) v = W[((int64_t)h * Fin + k) * Fo + f];
  } else if (col < R + Rs) {
    const int h = (col - R) / Fp, f = (col - R) % Fp;
    if (f < Fo) v = w_skip[((int64_t)h * Fin + k) * Fo + f];
  } else if (col < R + Rs + 2 * H) {
    const int c = col - R - Rs, h = c % H, which = c / H;
    const float* wr = W + ((int64_t)h * Fin + k) * Fo;
    const float* ar = a + (int64_t)h * 2 * Fo + which * Fo;
    for (int f = 0; f < Fo; ++f) v = fmaf(wr[f], ar[f], v);
  }
  Wcat[idx] = v;
}

__global__ __launch_bounds__(256) void unpack_wgrad_kernel(int H, int Fin, int Fo, int Fp,
                                                           const float* __restrict__ dWcat, int64_t ld,
                                                           int col_offset, float* __restrict__ dW) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)H * Fin * Fo) return;
  const int f = (int)(idx % Fo);
  const int k = (int)((idx / Fo) % Fin);
  const int h = (int)(idx / ((int64_t)Fo * Fin));
  dW[idx] = dWcat[(int64_t)k * ld + col_offset + h * Fp + f];
}

}  // namespace pygat

using namespace pygat;

extern "C" int pygat_gat_backward_row(const pygat_graph* g, int H, int Fo, float alpha, int flags, int mean_mode,
                                      const float* G, const float* Wh, const float* s, const float* t,
                                      const float* sk, const float* hattn, const float* m, const float* Z,
                                      float* Gp, float* ebuf, float* ds, void* part, void* stream) {
  BwdRowArgs a;
  int rc = check_graph(g, &a.g);
  if (rc) return rc;
  PYGAT_REQUIRE(make_row_shape(H, Fo, &a.rs), "gat_backward_row: unsupported H=%d F'=%d", H, Fo);
  PYGAT_REQUIRE(G && Wh && s && t && hattn && m && Z && Gp && ebuf && ds, "gat_backward_row: null pointer");
  PYGAT_REQUIRE(!(flags & PYGAT_F_SKIP) || sk, "gat_backward_row: PYGAT_F_SKIP without sk");
  PYGAT_REQUIRE(!(mean_mode && (flags & PYGAT_F_ELU)), "gat_backward_row: head mean never carries an ELU (models.py:23)");
  PYGAT_REQUIRE(aligned16(Wh) && aligned16(hattn) && aligned16(Gp) && (!sk || aligned16(sk)) &&
                    (mean_mode || a.rs.Fo != a.rs.Fp || aligned16(G)),
                "gat_backward_row: row tables must be 16-byte aligned");
  PYGAT_REQUIRE(a.g.n_items == 0 || part, "gat_backward_row: heavy rows present but no partials workspace");
  a.alpha = alpha; a.flags = flags; a.mean_mode = mean_mode; a.G = G; a.Wh = Wh; a.s = s; a.t = t; a.sk = sk;
  a.hattn = hattn; a.m = m; a.Z = Z; a.Gp = Gp; a.ebuf = ebuf; a.ds = ds; a.part = (float*)part;
  int lpr, vec;
  pick_lanes(a.rs, &lpr, &vec);
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = (unsigned)cdiv((int64_t)a.g.n + a.g.n_items, 4);
  PYGAT_DISPATCH_LANES(lpr, vec,
                       hipLaunchKernelGGL((gat_bwd_row_kernel<LPR, VEC>), dim3(blocks), dim3(256), 0, st, a));
  PYGAT_CHECK_LAUNCH("gat_backward_row");
  if (a.g.n_heavy > 0) {
    hipLaunchKernelGGL(gat_bwd_row_combine_kernel, dim3((unsigned)cdiv((int64_t)a.g.n_heavy * H, 256)), dim3(256), 0,
                       st, a);
    PYGAT_CHECK_LAUNCH("gat_backward_row_combine");
  }
  return PYGAT_OK;
}

extern "C" int pygat_gat_backward_col(const pygat_graph* gT, const int32_t* perm_t, int H, int Fo, const float* Gp,
                                      const float* ebuf, const float* ds, const float* a_pad, float* dWh,
                                      float* dt, void* part, void* stream) {
  BwdColArgs a;
  int rc = check_graph(gT, &a.g);
  if (rc) return rc;
  PYGAT_REQUIRE(make_row_shape(H, Fo, &a.rs), "gat_backward_col: unsupported H=%d F'=%d", H, Fo);
  PYGAT_REQUIRE(perm_t && Gp && ebuf && ds && a_pad && dWh && dt, "gat_backward_col: null pointer");
  PYGAT_REQUIRE(aligned16(Gp) && aligned16(dWh) && aligned16(a_pad), "gat_backward_col: row tables must be 16-byte aligned");
  PYGAT_REQUIRE(a.g.n_items == 0 || part, "gat_backward_col: heavy rows present but no partials workspace");
  a.perm = perm_t; a.Gp = Gp; a.ebuf = ebuf; a.ds = ds; a.a_pad = a_pad; a.dWh = dWh; a.dt = dt;
  a.part = (float*)part;
  int lpr, vec;
  pick_lanes(a.rs, &lpr, &vec);
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = (unsigned)cdiv((int64_t)a.g.n + a.g.n_items, 4);
  PYGAT_DISPATCH_LANES(lpr, vec,
                       hipLaunchKernelGGL((gat_bwd_col_kernel<LPR, VEC>), dim3(blocks), dim3(256), 0, st, a));
  PYGAT_CHECK_LAUNCH("gat_backward_col");
  if (a.g.n_heavy > 0) {
    const unsigned cb = (unsigned)cdiv(a.g.n_heavy, 4);
    PYGAT_DISPATCH_LANES(lpr, vec,
                         hipLaunchKernelGGL((gat_bwd_col_combine_kernel<LPR, VEC>), dim3(cb), dim3(256), 0, st, a));
    PYGAT_CHECK_LAUNCH("gat_backward_col_combine");
  }
  return PYGAT_OK;
}

extern "C" size_t pygat_agrad_workspace_bytes(int H, int Fo) {
  int Fp = padded_width(Fo);
  if (H <= 0 || Fp == 0) return 0;
  return (size_t)AG_BLOCKS * 2 * (size_t)(H * Fp) * sizeof(float);
}

extern "C" int pygat_a_grad(int n, int H, int Fo, const float* Wh, const float* ds, const float* dt, float* da,
                            void* ws, void* stream) {
  RowShape rs;
  PYGAT_REQUIRE(make_row_shape(H, Fo, &rs), "a_grad: unsupported H=%d F'=%d", H, Fo);
  PYGAT_REQUIRE(n > 0 && Wh && ds && dt && da && ws && aligned16(Wh) && aligned16(ws), "a_grad: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(a_grad_partial_kernel, dim3(AG_BLOCKS), dim3(256), 0, st, n, rs, Wh, ds, dt, (float*)ws);
  hipLaunchKernelGGL(a_grad_final_kernel, dim3((unsigned)cdiv(rs.H * 2 * rs.Fo, 256)), dim3(256), 0, st, rs,
                     (const float*)ws, da);
  PYGAT_CHECK_LAUNCH("a_grad");
  return PYGAT_OK;
}

extern "C" int pygat_pack_params(int H, int Fin, int Fo, const float* W, const float* a, const float* w_skip,
                                 float* Wcat, int64_t ldw, float* a_pad, void* stream) {
  int Fp = padded_width(Fo);
  PYGAT_REQUIRE(H > 0 && Fin > 0 && Fp > 0 && W && a && Wcat && a_pad, "pack_params: bad arguments");
  const int need = H * Fp * (w_skip ? 2 : 1) + 2 * H;
  PYGAT_REQUIRE(ldw >= need && ldw % 4 == 0, "pack_params: ldw=%lld must be a multiple of 4 and >= %d", (long long)ldw, need);
  int64_t tot = (int64_t)Fin * ldw;
  if (tot < (int64_t)H * 2 * Fp) tot = (int64_t)H * 2 * Fp;
  hipLaunchKernelGGL(pack_params_kernel, dim3((unsigned)cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, H, Fin, Fo,
                     Fp, W, a, w_skip, Wcat, ldw, a_pad);
  PYGAT_CHECK_LAUNCH("pack_params");
  return PYGAT_OK;
}

extern "C" int pygat_unpack_wgrad(int H, int Fin, int Fo, const float* dWcat, int64_t ld, int col_offset, float* dW,
                                  void* stream) {
  int Fp = padded_width(Fo);
  PYGAT_REQUIRE(H > 0 && Fin > 0 && Fp > 0 && dWcat && dW && col_offset >= 0 && ld >= col_offset + H * Fp,
                "unpack_wgrad: bad arguments");
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3((unsigned)cdiv((int64_t)H * Fin * Fo, 256)), dim3(256), 0,
                     (hipStream_t)stream, H, Fin, Fo, Fp, dWcat, ld, col_offset, dW);
  PYGAT_CHECK_LAUNCH("unpack_wgrad");
  return PYGAT_OK;
}
