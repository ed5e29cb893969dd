// K6 -- backward of the reference's sparse GATv2 layer (SpGraphAttentionLayerV2, layers.py:258-313).
//
//   e_ij = a . LeakyReLU(Whi_i + Whj_j)  per head          (layers.py:280-283)
//   alpha_ij = softmax_j e_ij,  h'_i = sum_j alpha~_ij Whi_j (layers.py:285-300; note Whi, not Whj, is aggregated)
// With Gp, D as in K3a:  dp_ij = Gp_i . Whi_j,  de_ij = alpha_ij (mask_ij dp_ij - D_i),
//   q_ij = de_ij * a (.) LeakyReLU'(Whi_i + Whj_j)          (an F'-vector per edge, never stored)
//   dWhi_i += sum_j q_ij (row sums)     dWhj_j += sum_i q_ij (column sums)     dWhi_j += sum_i alpha~_ij Gp_i
//   da = sum_ij de_ij LeakyReLU(Whi_i + Whj_j)
// Same nnz-split structure as K3b/K4.  The COLUMN pass runs first, over the transposed pattern (GRW_i = [Gp | m,1/Z,D | Whi]
// gathered as one contiguous row, WW_j = [Whi_j | Whj_j] row-local): it recomputes e, alpha and de per edge, sums dWhj_j
// and the aggregation part of dWhi_j, and leaves de_ij per transposed edge (H floats, sequential).  The ROW pass (forward
// pattern) then needs neither Gp nor Whi_j: it fetches de_ij through perm_f and gathers only the Whj HALF of WW_j to form
// q_ij = de_ij a (.) LeakyReLU'(Whi_i + Whj_j) -- 0.6 KB per edge instead of the 1 KB [Whi_j | Whj_j] row it used to
// gather to recompute de (round 3: 1.97 -> see DESIGN.md) -- adds the row sums into dWhi_i and forms da.
#include "attn_common.h"

namespace pygat {

struct V2Args {
  GraphDev g;
  RowShape rs;
  float alpha;
  const int32_t* perm;  // column pass: transposed position -> forward edge (mask index); row pass: forward edge -> transposed position
  float* de_t;          // [nnz][H] de_ij per TRANSPOSED edge: written by the column pass, read by the row pass
  const float* mask;    // [nnz][H] forward order or nullptr
  const float* WW;      // [n][2R]
  const float* GRW;     // [n][2R + 4H]
  const float* a2;      // [H][Fp]
  const float* dwhi_row;  // column pass: row-side part of dWhi to add in
  float* out;           // row pass: dWhi_row [n][R]; column pass: dWW [n][2R]
  float* part;          // [2 * nslots][R] (row) / [2 * nslots][2R] (col)
  float* da_part;       // row pass: [blocks][R]
};

__device__ __forceinline__ float lrelu2(float z, float alpha) { return z > 0.f ? z : alpha * z; }

// per-edge recomputation shared by both passes
struct EdgeOut { float al, de; float4 q, l; };
__device__ __forceinline__ EdgeOut v2_edge(float4 wi_i, float4 wi_j, float4 wj_j, float4 gp, float4 rt, float4 a4,
                                           float mk, float alpha, bool valid, int lph) {
  const float4 h = make_float4(wi_i.x + wj_j.x, wi_i.y + wj_j.y, wi_i.z + wj_j.z, wi_i.w + wj_j.w);
  EdgeOut o;
  o.l = make_float4(lrelu2(h.x, alpha), lrelu2(h.y, alpha), lrelu2(h.z, alpha), lrelu2(h.w, alpha));
  const float e = group_sum_rt(dot4(o.l, a4), lph);          // a4 is zero on padded / invalid chunks
  const float dp = group_sum_rt(valid ? dot4(gp, wi_j) : 0.f, lph);
  const float a0 = __expf(e - rt.y) * rt.z;
  o.de = a0 * (mk * dp - rt.w);
  o.al = a0 * mk;
  o.q = make_float4(o.de * a4.x * (h.x > 0.f ? 1.f : alpha), o.de * a4.y * (h.y > 0.f ? 1.f : alpha),
                    o.de * a4.z * (h.z > 0.f ? 1.f : alpha), o.de * a4.w * (h.w > 0.f ? 1.f : alpha));
  return o;
}

// ---------------------------------------------------------------------------- row pass
template <int LPR, int VEC>
__global__ __launch_bounds__(256) void gat2_bwd_row_kernel(V2Args a) {
  constexpr int EPW = 64 / LPR;
  constexpr int U = (VEC == 1) ? 4 : 2;
  __shared__ __attribute__((aligned(16))) float sm_da[4][1024];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t k = ((int64_t)blockIdx.x * 4 + w) * EPW + lane / LPR;
  const bool active = k < a.g.kn;   // (kn: all slots, or the prefix before a self-loop-only tail) no early return: every lane joins the da reduction below
  const LaneCols<VEC> lc = lane_cols<LPR, VEC>(a.rs);
  const int H = a.rs.H, R = a.rs.R;
  const int64_t LW = 2 * (int64_t)R, LG = 2 * (int64_t)R + 4 * H;
  float4 a4[VEC], dacc[VEC], acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    a4[v] = ld4(a.a2 + (int64_t)lc.head[v] * a.rs.Fp + (lc.cofs[v] & (a.rs.Fp - 1)));
    if (!lc.valid[v]) a4[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    dacc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (active) {
    int64_t e0, e1;
    slot_range(a.g, k, &e0, &e1);
    const int2* __restrict__ rc = a.g.rc;
    const int r_first = rc[e0].x;
    const bool head_partial = a.g.rowptr[r_first] < e0;
    int cur = r_first;
    auto flush = [&](int i, bool is_head, bool is_tail) {   // a whole row: added into dWhi_i = dWW[i, :R] (the column pass wrote it)
      if (is_head || is_tail) {
        float* dst = a.part + (2 * k + (is_head ? 0 : 1)) * (int64_t)R;
#pragma unroll
        for (int v = 0; v < VEC; ++v)
          if (lc.valid[v]) st4(dst + lc.cofs[v], acc[v]);
      } else {
        float* dst = a.out + (int64_t)i * LW;
#pragma unroll
        for (int v = 0; v < VEC; ++v)
          if (lc.valid[v]) {
            const float4 o = ld4(dst + lc.cofs[v]);
            st4(dst + lc.cofs[v], make_float4(o.x + acc[v].x, o.y + acc[v].y, o.z + acc[v].z, o.w + acc[v].w));
          }
      }
    };
    for (int64_t e = e0; e < e1; e += U) {
      int2 p[U];
#pragma unroll
      for (int u = 0; u < U; ++u) p[u] = rc[(e + u < e1) ? e + u : e1 - 1];
      // de_ij comes from the column pass (transposed position of this edge: perm); only the Whj half of row j is gathered
      float dev[U][VEC];
      float4 qv[U][VEC], lv[U][VEC];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t ee = (e + u < e1) ? e + u : e1 - 1;
        const int64_t pt = a.perm[ee];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const float4 wi = ld4(a.GRW + (int64_t)p[u].x * LG + R + 4 * H + lc.cofs[v]);   // Whi_i, row-local
          const float4 wj = ld4(a.WW + (int64_t)p[u].y * LW + R + lc.cofs[v]);            // Whj_j, gathered
          const float de = a.de_t[pt * H + lc.head[v]];
          const float4 h = make_float4(wi.x + wj.x, wi.y + wj.y, wi.z + wj.z, wi.w + wj.w);
          dev[u][v] = de;
          lv[u][v] = make_float4(lrelu2(h.x, a.alpha), lrelu2(h.y, a.alpha), lrelu2(h.z, a.alpha), lrelu2(h.w, a.alpha));
          qv[u][v] = make_float4(de * a4[v].x * (h.x > 0.f ? 1.f : a.alpha), de * a4[v].y * (h.y > 0.f ? 1.f : a.alpha),
                                 de * a4[v].z * (h.z > 0.f ? 1.f : a.alpha), de * a4[v].w * (h.w > 0.f ? 1.f : a.alpha));
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (e + u < e1) {
          if (p[u].x != cur) {
            flush(cur, cur == r_first && head_partial, false);
            cur = p[u].x;
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            acc[v].x += qv[u][v].x; acc[v].y += qv[u][v].y; acc[v].z += qv[u][v].z; acc[v].w += qv[u][v].w;
            dacc[v].x = fmaf(dev[u][v], lv[u][v].x, dacc[v].x); dacc[v].y = fmaf(dev[u][v], lv[u][v].y, dacc[v].y);
            dacc[v].z = fmaf(dev[u][v], lv[u][v].z, dacc[v].z); dacc[v].w = fmaf(dev[u][v], lv[u][v].w, dacc[v].w);
          }
        }
      }
    }
    flush(cur, cur == r_first && head_partial, a.g.rowptr[cur + 1] > e1);
  }
  // da: lane groups -> wave (shuffles) -> work-group (LDS) -> one record per work-group, fixed order
#pragma unroll
  for (int v = 0; v < VEC; ++v) dacc[v] = slot_sum4<LPR>(dacc[v]);
  if (lane / LPR == 0)
#pragma unroll
    for (int v = 0; v < VEC; ++v)
      if (lc.valid[v]) st4(&sm_da[w][lc.cofs[v]], dacc[v]);
  __syncthreads();
  for (int c = threadIdx.x; c < R; c += 256)
    a.da_part[(int64_t)blockIdx.x * R + c] = sm_da[0][c] + sm_da[1][c] + sm_da[2][c] + sm_da[3][c];
}

// sums the R-float pieces of the rows cut by a slot border (thread per (slot, float))
__global__ __launch_bounds__(256) void gat2_rowsum_fixup_kernel(V2Args a, int width, float* dst, int64_t ld_dst,
                                                                 const float* add, int col_finish) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t k = idx / width;
  const int c = (int)(idx % width);
  if (k >= num_slots(a.g)) return;
  int64_t e0, e1;
  slot_range(a.g, k, &e0, &e1);
  const int r = a.g.rc[e1 - 1].x;
  const int64_t row_end = a.g.rowptr[r + 1];
  if (row_end <= e1 || (int64_t)a.g.rowptr[r] < e0) return;
  const int64_t k_e = slot_of(a.g, row_end - 1);
  float acc = a.part[(2 * k + 1) * (int64_t)width + c];
  for (int64_t kk = k + 1; kk <= k_e; ++kk) acc += a.part[(2 * kk) * (int64_t)width + c];
  if (col_finish == 2) acc += dst[(int64_t)r * ld_dst + c];             // row pass: added into dWhi_i
  dst[(int64_t)r * ld_dst + c] = acc;
}

// stage 1 of the da reduction: DA_STAGE work-groups, each sums every DA_STAGE-th record (coalesced R-float rows)
constexpr int DA_STAGE = 128;
// list-driven variant: one work-group per cut row; thread c sums float c of its pieces, 4 in flight
__global__ __launch_bounds__(256) void gat2_rowsum_fixup_list_kernel(V2Args a, int width, float* dst, int64_t ld_dst,
                                                                      const float* add, int col_finish) {
  const int q0 = blockIdx.x;
  const int64_t k = a.g.cut[3 * q0];
  const int r = a.g.cut[3 * q0 + 1];
  const int npieces = a.g.cut[3 * q0 + 2];
  for (int c = threadIdx.x; c < width; c += 256) {
    float acc = a.part[(2 * k + 1) * (int64_t)width + c];
    int q = 1;
    for (; q + 3 < npieces; q += 4) {
      const float x0 = a.part[(2 * (k + q)) * (int64_t)width + c], x1 = a.part[(2 * (k + q + 1)) * (int64_t)width + c];
      const float x2 = a.part[(2 * (k + q + 2)) * (int64_t)width + c], x3 = a.part[(2 * (k + q + 3)) * (int64_t)width + c];
      acc += x0; acc += x1; acc += x2; acc += x3;
    }
    for (; q < npieces; ++q) acc += a.part[(2 * (k + q)) * (int64_t)width + c];
    if (col_finish == 2) acc += dst[(int64_t)r * ld_dst + c];           // row pass: added into dWhi_i
    dst[(int64_t)r * ld_dst + c] = acc;
  }
}

__global__ __launch_bounds__(256) void gat2_da_stage_kernel(int R, int nrec, const float* __restrict__ da_part,
                                                            float* __restrict__ out2) {
  for (int c = threadIdx.x; c < R; c += 256) {
    float acc = 0.f;
    int b = blockIdx.x;
    for (; b + 3 * DA_STAGE < nrec; b += 4 * DA_STAGE) {
      const float x0 = da_part[(int64_t)b * R + c], x1 = da_part[(int64_t)(b + DA_STAGE) * R + c];
      const float x2 = da_part[(int64_t)(b + 2 * DA_STAGE) * R + c], x3 = da_part[(int64_t)(b + 3 * DA_STAGE) * R + c];
      acc += x0; acc += x1; acc += x2; acc += x3;
    }
    for (; b < nrec; b += DA_STAGE) acc += da_part[(int64_t)b * R + c];
    out2[(int64_t)blockIdx.x * R + c] = acc;
  }
}

__global__ __launch_bounds__(256) void gat2_da_final_kernel(RowShape rs, int nblocks, const float* __restrict__ da_part,
                                                            float* __restrict__ da) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // over H * Fo
  if (idx >= rs.H * rs.Fo) return;
  const int h = idx / rs.Fo, f = idx % rs.Fo;
  float acc = 0.f;
  for (int b = 0; b < nblocks; ++b) acc += da_part[(int64_t)b * rs.R + h * rs.Fp + f];
  da[idx] = acc;
}

// ------------------------------------------------------------------------- column pass
template <int LPR, int VEC>
__global__ __launch_bounds__(256) void gat2_bwd_col_kernel(V2Args a) {
  constexpr int EPW = 64 / LPR;
  constexpr int U = 2;
  const int lane = threadIdx.x & 63;
  const int64_t k = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * EPW + lane / LPR;
  if (k >= a.g.kn) return;
  int64_t e0, e1;
  slot_range(a.g, k, &e0, &e1);
  const LaneCols<VEC> lc = lane_cols<LPR, VEC>(a.rs);
  const int H = a.rs.H, R = a.rs.R;
  const int64_t LW = 2 * (int64_t)R, LG = 2 * (int64_t)R + 4 * H;
  const int lph = a.rs.lph < 64 ? a.rs.lph : 64;
  const int2* __restrict__ rc = a.g.rc;
  float4 a4[VEC], accA[VEC], accJ[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    a4[v] = ld4(a.a2 + (int64_t)lc.head[v] * a.rs.Fp + (lc.cofs[v] & (a.rs.Fp - 1)));
    if (!lc.valid[v]) a4[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    accA[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    accJ[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int r_first = rc[e0].x;
  const bool head_partial = a.g.rowptr[r_first] < e0;
  int cur = r_first;
  auto flush = [&](int j, bool is_head, bool is_tail) {
    if (is_head || is_tail) {
      float* dst = a.part + (2 * k + (is_head ? 0 : 1)) * LW;
#pragma unroll
      for (int v = 0; v < VEC; ++v)
        if (lc.valid[v]) { st4(dst + lc.cofs[v], accA[v]); st4(dst + R + lc.cofs[v], accJ[v]); }
    } else {
      float* dst = a.out + (int64_t)j * LW;
#pragma unroll
      for (int v = 0; v < VEC; ++v)
        if (lc.valid[v]) {
          st4(dst + lc.cofs[v], accA[v]);          // the row pass adds its row sums to this half afterwards
          st4(dst + R + lc.cofs[v], accJ[v]);
        }
    }
  };
  for (int64_t e = e0; e < e1; e += U) {
    int2 p[U];  // (j, i)
#pragma unroll
    for (int u = 0; u < U; ++u) p[u] = rc[(e + u < e1) ? e + u : e1 - 1];
    EdgeOut eo[U][VEC];
    float4 gv[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const float* gi = a.GRW + (int64_t)p[u].y * LG;   // gathered: [Gp_i | rowtab_i | Whi_i]
        const float* wj = a.WW + (int64_t)p[u].x * LW;    // row-local: [Whi_j | Whj_j]
        float mk = 1.f;
        if (a.mask) mk = a.mask[(int64_t)a.perm[(e + u < e1) ? e + u : e1 - 1] * H + lc.head[v]];
        gv[u][v] = ld4(gi + lc.cofs[v]);
        eo[u][v] = v2_edge(ld4(gi + R + 4 * H + lc.cofs[v]), ld4(wj + lc.cofs[v]), ld4(wj + R + lc.cofs[v]), gv[u][v],
                           ld4(gi + R + 4 * lc.head[v]), a4[v], mk, a.alpha, lc.valid[v], lph);
        // de_ij for the row pass: the first lane of a head stores it (every lane of the head holds the same value)
        if (lc.valid[v] && e + u < e1 && ((lc.cofs[v] >> 2) & (lph - 1)) == 0) a.de_t[(e + u) * H + lc.head[v]] = eo[u][v].de;
      }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (e + u < e1) {
        if (p[u].x != cur) {
          flush(cur, cur == r_first && head_partial, false);
          cur = p[u].x;
#pragma unroll
          for (int v = 0; v < VEC; ++v) { accA[v] = make_float4(0.f, 0.f, 0.f, 0.f); accJ[v] = make_float4(0.f, 0.f, 0.f, 0.f); }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const float al = eo[u][v].al;
          accA[v].x = fmaf(al, gv[u][v].x, accA[v].x); accA[v].y = fmaf(al, gv[u][v].y, accA[v].y);
          accA[v].z = fmaf(al, gv[u][v].z, accA[v].z); accA[v].w = fmaf(al, gv[u][v].w, accA[v].w);
          accJ[v].x += eo[u][v].q.x; accJ[v].y += eo[u][v].q.y; accJ[v].z += eo[u][v].q.z; accJ[v].w += eo[u][v].q.w;
        }
      }
    }
  }
  flush(cur, cur == r_first && head_partial, a.g.rowptr[cur + 1] > e1);
}

}  // namespace pygat

using namespace pygat;

extern "C" size_t pygat_gatv2_workspace_bytes(int64_t nnz, int slot_edges, int H, int Fo) {
  const int Fp = padded_width(Fo);
  if (nnz <= 0 || slot_edges <= 0 || Fp == 0) return 0;
  const int64_t nslots = (nnz + slot_edges - 1) / slot_edges;
  const int64_t R = (int64_t)H * Fp;
  // partial records of the column pass (2R floats each) + one da record per row-pass work-group + de per transposed edge
  return (size_t)(2 * nslots * 2 * R + (nslots / 4 + 2) * R + DA_STAGE * R + nnz * H + 4) * sizeof(float);
}

extern "C" int pygat_gatv2_backward(const pygat_graph* g, const pygat_graph* gT, const int32_t* perm_t, const int32_t* perm_f,
                                    int H, int Fo, float alpha, const float* WW, const float* a2, const float* GRW,
                                    const float* att_mask, float* dWW, float* da, void* ws, void* stream) {
  V2Args a;
  int rc = check_graph(g, &a.g, 2);     // (a slot PREFIX: the slots before a self-loop-only tail, pygat_gat_backward_tail)
  if (rc) return rc;
  PYGAT_REQUIRE(make_row_shape(H, Fo, &a.rs), "gatv2_backward: unsupported H=%d F'=%d", H, Fo);
  PYGAT_REQUIRE(a.rs.R <= 1024, "gatv2_backward: row too wide");
  PYGAT_REQUIRE(WW && a2 && GRW && dWW && da && ws && perm_f, "gatv2_backward: null pointer");
  PYGAT_REQUIRE(!att_mask || perm_t, "gatv2_backward: an attention mask needs perm_t");
  PYGAT_REQUIRE(aligned16(WW) && aligned16(a2) && aligned16(GRW) && aligned16(dWW) && aligned16(ws),
                "gatv2_backward: row tables must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int R = a.rs.R;
  a.alpha = alpha; a.mask = att_mask; a.WW = WW; a.GRW = GRW; a.a2 = a2; a.dwhi_row = nullptr;
  int lpr, vec;
  pick_lanes(a.rs, &lpr, &vec);
  const int64_t nslots = num_slots(a.g);     // (the workspace is laid out for all slots)
  const unsigned blocks = (unsigned)cdiv(cdiv(a.g.kn, 64 / lpr), 4);
  PYGAT_REQUIRE(a.g.kn == nslots || a.g.cut, "gatv2_backward: a slot prefix needs the cut-row list");
  float* part = (float*)ws;
  float* da_part = part + 2 * nslots * 2 * (int64_t)R;
  float* da_stage = da_part + (nslots / 4 + 2) * (int64_t)R;
  float* de_t = da_stage + (int64_t)DA_STAGE * R;
  // column pass over the transposed pattern: dWW = [aggregation part of dWhi | dWhj], de per transposed edge
  V2Args b = a;
  rc = check_graph(gT, &b.g, 2);
  if (rc) return rc;
  PYGAT_REQUIRE(b.g.nnz == a.g.nnz && b.g.ts == a.g.ts && b.g.kn == a.g.kn, "gatv2_backward: g and gT differ in size / slot length / prefix");
  b.perm = perm_t; b.out = dWW; b.part = part; b.de_t = de_t; b.da_part = nullptr;
  PYGAT_DISPATCH_LANES(lpr, vec, hipLaunchKernelGGL((gat2_bwd_col_kernel<LPR, VEC>), dim3(blocks), dim3(256), 0, st, b));
  PYGAT_CHECK_LAUNCH("gatv2_backward_col");
  if (b.g.cut) {
    if (b.g.n_cut > 0)
      hipLaunchKernelGGL(gat2_rowsum_fixup_list_kernel, dim3((unsigned)b.g.n_cut), dim3(256), 0, st, b, 2 * R, dWW,
                         2 * (int64_t)R, (const float*)nullptr, 1);
  } else {
    hipLaunchKernelGGL(gat2_rowsum_fixup_kernel, dim3((unsigned)cdiv(nslots * 2 * R, 256)), dim3(256), 0, st, b, 2 * R, dWW,
                       2 * (int64_t)R, (const float*)nullptr, 1);
  }
  PYGAT_CHECK_LAUNCH("gatv2_backward_col_fixup");
  // row pass over the forward pattern: dWhi_i += sum_j q_ij (de through perm_f, only Whj_j gathered), da
  a.perm = perm_f; a.out = dWW; a.part = part; a.da_part = da_part; a.de_t = de_t;
  PYGAT_DISPATCH_LANES(lpr, vec, hipLaunchKernelGGL((gat2_bwd_row_kernel<LPR, VEC>), dim3(blocks), dim3(256), 0, st, a));
  PYGAT_CHECK_LAUNCH("gatv2_backward_row");
  if (a.g.cut) {
    if (a.g.n_cut > 0)
      hipLaunchKernelGGL(gat2_rowsum_fixup_list_kernel, dim3((unsigned)a.g.n_cut), dim3(256), 0, st, a, R, dWW, 2 * (int64_t)R,
                         (const float*)nullptr, 2);
  } else {
    hipLaunchKernelGGL(gat2_rowsum_fixup_kernel, dim3((unsigned)cdiv(nslots * R, 256)), dim3(256), 0, st, a, R, dWW,
                       2 * (int64_t)R, (const float*)nullptr, 2);
  }
  hipLaunchKernelGGL(gat2_da_stage_kernel, dim3(DA_STAGE), dim3(256), 0, st, R, (int)blocks, (const float*)da_part, da_stage);
  hipLaunchKernelGGL(gat2_da_final_kernel, dim3((unsigned)cdiv(a.rs.H * a.rs.Fo, 256)), dim3(256), 0, st, a.rs, DA_STAGE,
                     (const float*)da_stage, da);
  PYGAT_CHECK_LAUNCH("gatv2_backward_row_fixup");
  return PYGAT_OK;
}
