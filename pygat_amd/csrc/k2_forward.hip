// K2 -- fused edge-softmax + neighbour aggregation, forward (gfx950, wave64).
//
// Replaces the per-head ATen sequence of the reference
//   layers.py:141     edge_h = cat(Wh[edge[0]], Wh[edge[1]])      (never materialised here)
//   layers.py:144     edge_e = LeakyReLU(a . edge_h)              = LeakyReLU(s_i + t_j), layers.py:60-64
//   layers.py:145-146 scatter_max + exp(edge_e - max)             row max / numerators
//   layers.py:150     special_spmm(edge, edge_e, ones)            Z_i
//   layers.py:156-160 special_spmm(edge, edge_e, Wh) / Z          hattn_i
//   layers.py:165-170 (+ skip) and ELU                            epilogue
// and the dense equivalents layers.py:40-51, for all local heads in one pass.
//
// One wave per CSR row: the wave's EPW edge slots each gather one whole
// head-interleaved Wh row (16 B per lane, coalesced) per instruction, keep
// (m, Z, acc) in registers with the online-softmax recurrence, and merge the
// slots at the end with a shuffle.  No LDS, no atomics, no [2F',E] or [N,N]
// temporaries.  Rows longer than g.chunk are cut into items handled by extra
// waves of the same launch; a second tiny launch merges their partials in item
// order (deterministic).
#include "attn_common.h"

namespace pygat {

struct FwdArgs {
  GraphDev g;
  RowShape rs;
  float alpha;
  int flags;
  const float* Wh;
  const float* s;
  const float* t;
  const float* sk;
  float* out;
  float* hattn;
  float* m;
  float* Z;
  float* part;  // [n_items][R + 2H]
};

__device__ __forceinline__ float lrelu(float z, float alpha) { return z > 0.f ? z : alpha * z; }

// online-softmax merge of (m2,z2,a2) into (m,z,a)
__device__ __forceinline__ void merge_state(float& m, float& z, float4& a, float m2, float z2, float4 a2) {
  float mn = fmaxf(m, m2);
  float sa = __expf(m - mn), sb = __expf(m2 - mn);
  z = z * sa + z2 * sb;
  a.x = a.x * sa + a2.x * sb; a.y = a.y * sa + a2.y * sb;
  a.z = a.z * sa + a2.z * sb; a.w = a.w * sa + a2.w * sb;
  m = mn;
}

// accumulate the edges [e0,e1) of row i into the lane state; on return every lane of
// slot 0 (and all other slots) holds the totals for its chunk(s).
template <int LPR, int VEC>
__device__ __forceinline__ void fwd_range(const FwdArgs& a, const LaneCols<VEC>& lc, int i, int e0, int e1,
                                          float (&m)[VEC], float (&z)[VEC], float4 (&acc)[VEC]) {
  constexpr int EPW = 64 / LPR;
  constexpr int U = (VEC == 1) ? 4 : 2;
  const int slot = (threadIdx.x & 63) / LPR;
  const int H = a.rs.H, R = a.rs.R;
  float si[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    si[v] = a.s[(int64_t)i * H + lc.head[v]];
    m[v] = NEG_BIG; z[v] = 0.f; acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  int e = e0 + slot;
  for (; e + (U - 1) * EPW < e1; e += U * EPW) {
    int j[U];
#pragma unroll
    for (int u = 0; u < U; ++u) j[u] = a.g.col[e + u * EPW];
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
    for (int v = 0; v < VEC; ++v) {
      float ev[U];
      float mn = m[v];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        ev[u] = lrelu(si[v] + tv[u][v], a.alpha);
        mn = fmaxf(mn, ev[u]);
      }
      float sc = __expf(m[v] - mn);
      float zz = z[v] * sc;
      float4 ac = make_float4(acc[v].x * sc, acc[v].y * sc, acc[v].z * sc, acc[v].w * sc);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float p = __expf(ev[u] - mn);
        zz += p;
        ac.x = fmaf(p, wv[u][v].x, ac.x); ac.y = fmaf(p, wv[u][v].y, ac.y);
        ac.z = fmaf(p, wv[u][v].z, ac.z); ac.w = fmaf(p, wv[u][v].w, ac.w);
      }
      m[v] = mn; z[v] = zz; acc[v] = ac;
    }
  }
  for (; e < e1; e += EPW) {
    const int j = a.g.col[e];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float tj = a.t[(int64_t)j * H + lc.head[v]];
      float4 w = ld4(a.Wh + (int64_t)j * R + lc.cofs[v]);
      float ev = lrelu(si[v] + tj, a.alpha);
      float mn = fmaxf(m[v], ev);
      float sc = __expf(m[v] - mn), p = __expf(ev - mn);
      z[v] = z[v] * sc + p;
      acc[v].x = fmaf(p, w.x, acc[v].x * sc); acc[v].y = fmaf(p, w.y, acc[v].y * sc);
      acc[v].z = fmaf(p, w.z, acc[v].z * sc); acc[v].w = fmaf(p, w.w, acc[v].w * sc);
      m[v] = mn;
    }
  }
  // merge the edge slots (all lanes reconverged here)
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float m2 = __shfl_xor(m[v], off), z2 = __shfl_xor(z[v], off);
      float4 a2;
      a2.x = __shfl_xor(acc[v].x, off); a2.y = __shfl_xor(acc[v].y, off);
      a2.z = __shfl_xor(acc[v].z, off); a2.w = __shfl_xor(acc[v].w, off);
      merge_state(m[v], z[v], acc[v], m2, z2, a2);
    }
  }
}

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1f(x); }

// normalise, epilogue (skip, ELU) and stores for row i; executed by the slot-0 lanes
template <int LPR, int VEC>
__device__ __forceinline__ void fwd_finish(const FwdArgs& a, const LaneCols<VEC>& lc, int i,
                                           const float (&m)[VEC], const float (&z)[VEC],
                                           const float4 (&acc)[VEC]) {
  const int slot = (threadIdx.x & 63) / LPR;
  if (slot != 0) return;
  const int H = a.rs.H, R = a.rs.R, Fo = a.rs.Fo, Fp = a.rs.Fp;
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    if (!lc.valid[v]) continue;
    const int co = lc.cofs[v], h = lc.head[v];
    float4 hat = make_float4(acc[v].x / z[v], acc[v].y / z[v], acc[v].z / z[v], acc[v].w / z[v]);
    if (a.hattn) st4(a.hattn + (int64_t)i * R + co, hat);
    if (a.out) {
      float4 pre = hat;
      if (a.flags & PYGAT_F_SKIP) {
        float4 k4 = ld4(a.sk + (int64_t)i * R + co);
        pre.x += k4.x; pre.y += k4.y; pre.z += k4.z; pre.w += k4.w;
      }
      if (a.flags & PYGAT_F_ELU) { pre.x = elu1(pre.x); pre.y = elu1(pre.y); pre.z = elu1(pre.z); pre.w = elu1(pre.w); }
      if (Fo == Fp) {
        st4(a.out + (int64_t)i * R + co, pre);
      } else {
        const int f0 = co & (Fp - 1);
        float* o = a.out + (int64_t)i * H * Fo + (int64_t)h * Fo + f0;
        if (f0 + 0 < Fo) o[0] = pre.x;
        if (f0 + 1 < Fo) o[1] = pre.y;
        if (f0 + 2 < Fo) o[2] = pre.z;
        if (f0 + 3 < Fo) o[3] = pre.w;
      }
    }
    if (a.m && ((co >> 2) & (a.rs.lph - 1)) == 0) {
      a.m[(int64_t)i * H + h] = m[v];
      a.Z[(int64_t)i * H + h] = z[v];
    }
  }
}

template <int LPR, int VEC>
__global__ __launch_bounds__(256) void gat_fwd_kernel(FwdArgs a) {
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const LaneCols<VEC> lc = lane_cols<LPR, VEC>(a.rs);
  float m[VEC], z[VEC];
  float4 acc[VEC];
  if (gw < a.g.n) {
    const int e0 = a.g.rowptr[gw], e1 = a.g.rowptr[gw + 1];
    if (e1 - e0 > a.g.chunk || e1 == e0) return;  // heavy rows: items below; empty rows are rejected on the host
    fwd_range<LPR, VEC>(a, lc, gw, e0, e1, m, z, acc);
    fwd_finish<LPR, VEC>(a, lc, gw, m, z, acc);
  } else {
    const int it = gw - a.g.n;
    if (it >= a.g.n_items) return;
    const int i = a.g.heavy_row[a.g.item_row_slot[it]];
    fwd_range<LPR, VEC>(a, lc, i, a.g.item_begin[it], a.g.item_end[it], m, z, acc);
    if ((threadIdx.x & 63) / LPR != 0) return;
    float* p = a.part + (int64_t)it * (a.rs.R + 2 * a.rs.H);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      if (!lc.valid[v]) continue;
      st4(p + lc.cofs[v], acc[v]);
      if (((lc.cofs[v] >> 2) & (a.rs.lph - 1)) == 0) {
        p[a.rs.R + lc.head[v]] = m[v];
        p[a.rs.R + a.rs.H + lc.head[v]] = z[v];
      }
    }
  }
}

// one wave per heavy row: merge its item partials in item order, then the normal epilogue
template <int LPR, int VEC>
__global__ __launch_bounds__(256) void gat_fwd_combine_kernel(FwdArgs a) {
  const int hr = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (hr >= a.g.n_heavy) return;
  const LaneCols<VEC> lc = lane_cols<LPR, VEC>(a.rs);
  float m[VEC], z[VEC];
  float4 acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) { m[v] = NEG_BIG; z[v] = 0.f; acc[v] = make_float4(0.f, 0.f, 0.f, 0.f); }
  const int PS = a.rs.R + 2 * a.rs.H;
  for (int it = a.g.heavy_item_ptr[hr]; it < a.g.heavy_item_ptr[hr + 1]; ++it) {
    const float* p = a.part + (int64_t)it * PS;
#pragma unroll
    for (int v = 0; v < VEC; ++v)
      merge_state(m[v], z[v], acc[v], p[a.rs.R + lc.head[v]], p[a.rs.R + a.rs.H + lc.head[v]], ld4(p + lc.cofs[v]));
  }
  fwd_finish<LPR, VEC>(a, lc, a.g.heavy_row[hr], m, z, acc);
}

// models.py:34 -- mean over heads of (hattn [+ sk]); one thread per output element
__global__ __launch_bounds__(256) void head_mean_kernel(int n, int H, int Fo, int Fp,
                                                        const float* __restrict__ hattn,
                                                        const float* __restrict__ sk,
                                                        float* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)n * Fo) return;
  int64_t i = idx / Fo;
  int f = (int)(idx % Fo);
  const float* r = hattn + i * (int64_t)H * Fp + f;
  float acc = 0.f;
  // same association as torch.mean over the stacked heads: sum in head order, then divide
  for (int h = 0; h < H; ++h) {
    float v = r[h * Fp];
    if (sk) v += sk[i * (int64_t)H * Fp + h * Fp + f];
    acc += v;
  }
  out[idx] = acc / (float)H;
}

}  // namespace pygat

using namespace pygat;

extern "C" size_t pygat_partials_bytes(int n_items, int H, int Fp) {
  if (n_items <= 0) return 0;
  return (size_t)n_items * (size_t)(H * Fp + 2 * H) * sizeof(float);
}

extern "C" int pygat_gat_forward(const pygat_graph* g, int H, int Fo, float alpha, int flags, const float* Wh,
                                 const float* s, const float* t, const float* sk, float* out, float* hattn,
                                 float* m, float* Z, void* part, void* stream) {
  FwdArgs a;
  int rc = check_graph(g, &a.g);
  if (rc) return rc;
  PYGAT_REQUIRE(make_row_shape(H, Fo, &a.rs), "gat_forward: unsupported H=%d F'=%d (need H*pad(F') <= 1024, F' <= 256)", H, Fo);
  PYGAT_REQUIRE(Wh && s && t, "gat_forward: null Wh/s/t");
  PYGAT_REQUIRE(out || hattn, "gat_forward: need out and/or hattn");
  PYGAT_REQUIRE(!(flags & PYGAT_F_SKIP) || sk, "gat_forward: PYGAT_F_SKIP without sk");
  PYGAT_REQUIRE((m == nullptr) == (Z == nullptr), "gat_forward: m and Z must be given together");
  PYGAT_REQUIRE(aligned16(Wh) && (!sk || aligned16(sk)) && (!hattn || aligned16(hattn)) &&
                    (!out || a.rs.Fo != a.rs.Fp || aligned16(out)),
                "gat_forward: row tables must be 16-byte aligned");
  PYGAT_REQUIRE(a.g.n_items == 0 || part, "gat_forward: heavy rows present but no partials workspace");
  a.alpha = alpha; a.flags = flags; a.Wh = Wh; a.s = s; a.t = t; a.sk = sk;
  a.out = out; a.hattn = hattn; a.m = m; a.Z = Z; a.part = (float*)part;
  int lpr, vec;
  pick_lanes(a.rs, &lpr, &vec);
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = (unsigned)cdiv((int64_t)a.g.n + a.g.n_items, 4);
  PYGAT_DISPATCH_LANES(lpr, vec,
                       hipLaunchKernelGGL((gat_fwd_kernel<LPR, VEC>), dim3(blocks), dim3(256), 0, st, a));
  PYGAT_CHECK_LAUNCH("gat_forward");
  if (a.g.n_heavy > 0) {
    const unsigned cb = (unsigned)cdiv(a.g.n_heavy, 4);
    PYGAT_DISPATCH_LANES(lpr, vec,
                         hipLaunchKernelGGL((gat_fwd_combine_kernel<LPR, VEC>), dim3(cb), dim3(256), 0, st, a));
    PYGAT_CHECK_LAUNCH("gat_forward_combine");
  }
  return PYGAT_OK;
}

extern "C" int pygat_head_mean(int n, int H, int Fo, const float* hattn, const float* sk, float* out,
                               void* stream) {
  int Fp = padded_width(Fo);
  PYGAT_REQUIRE(n > 0 && H > 0 && Fp > 0 && hattn && out, "head_mean: bad arguments");
  hipLaunchKernelGGL(head_mean_kernel, dim3((unsigned)cdiv((int64_t)n * Fo, 256)), dim3(256), 0,
                     (hipStream_t)stream, n, H, Fo, Fp, hattn, sk, out);
  PYGAT_CHECK_LAUNCH("head_mean");
  return PYGAT_OK;
}
