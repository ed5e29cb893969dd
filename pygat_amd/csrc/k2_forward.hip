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
// nnz split: the edge list (row,col pairs in CSR order) is cut into slots of `ts`
// edges.  A lane group (LPR lanes = one head-interleaved row of 16 B per lane) owns
// a slot: it issues U independent (s_i, t_j, Wh_j) gathers at a time -- the memory
// parallelism does not depend on the degree of the rows it happens to cross --
// and folds them into the online-softmax state (m, Z, acc) of the current row.
// When the row id changes the finished row is normalised and stored; the first
// / last row of a slot may continue in a neighbouring slot, then its state goes
// to `part` and gat_fwd_fixup_kernel merges the pieces in slot order.
// No LDS, no atomics, no [2F',E] or [N,N] temporaries, bitwise reproducible.
#include "attn_common.h"

namespace pygat {

struct FwdArgs {
  GraphDev g;
  RowShape rs;
  float alpha;
  int flags;
  const float* Wh;     // gathered table; GATv2: [Whi | Whj] rows of 2R floats
  int64_t ldwh;        // its row stride in floats (R, or 2R for GATv2)
  const float* s;
  const float* a_pad;  // [H][2][Fp]; the a_dst halves give t_j = Wh_j . a_dst on the fly (GATv2: a [H][Fp])
  const float* sk;
  const float* mask;  // [nnz][H] attention dropout mask (pre-scaled) or nullptr
  float* out;
  float* hattn;
  float* m;
  float* Z;
  float* part;  // [2 * nslots][PS]: record 2k = head piece of slot k, 2k+1 = tail piece; PS = R + 2H, or 2R + 3H (AUX)
  // AUX (training forward): the part of the row sum that went through the alpha branch of the LeakyReLU,
  //   aneg_i = sum_{j: z_ij <= 0} alpha~_ij Wh_j,   qneg_i = sum_{j: z_ij <= 0} alpha_ij          (alpha~ = alpha * mask)
  // With them the backward gets ds_i = sum_j dz_ij row-locally: sum_j de_ij = 0, hence
  //   ds_i = -(1 - slope) (Gp_i . aneg_i - D_i qneg_i)
  // and neither a second gather pass nor per-edge dz records are needed (k3_backward_row.hip, K3a).
  float* aneg;  // [n][ldr] or nullptr
  float* qneg;  // [n][ldh] or nullptr
};

__device__ __forceinline__ float lrelu(float z, float alpha) { return z > 0.f ? z : alpha * z; }

// fold one edge (logit ev, row w) into the running softmax state; one exp per edge
// (mk = dropout mask on alpha: scales the aggregated term only, Z is taken before it, layers.py:150-153)
template <bool AUX>
__device__ __forceinline__ void fold_edge(float& m, float& z, float4& a, float& zn, float4& an, float ev, bool neg,
                                          float4 w, float mk) {
  const float d = ev - m;
  const float ex = __expf(-fabsf(d));
  const bool up = d > 0.f;
  const float sc = up ? ex : 1.f, p = up ? 1.f : ex;
  z = fmaf(z, sc, p);
  const float pm = p * mk;
  a.x = fmaf(a.x, sc, pm * w.x); a.y = fmaf(a.y, sc, pm * w.y);
  a.z = fmaf(a.z, sc, pm * w.z); a.w = fmaf(a.w, sc, pm * w.w);
  if constexpr (AUX) {   // the same sums restricted to the edges on the alpha branch
    const float pn = neg ? p : 0.f, pmn = neg ? pm : 0.f;
    zn = fmaf(zn, sc, pn);
    an.x = fmaf(an.x, sc, pmn * w.x); an.y = fmaf(an.y, sc, pmn * w.y);
    an.z = fmaf(an.z, sc, pmn * w.z); an.w = fmaf(an.w, sc, pmn * w.w);
  }
  m = up ? ev : m;
}

// online-softmax merge of (m2,z2,a2) into (m,z,a)
template <bool AUX>
__device__ __forceinline__ void merge_state(float& m, float& z, float4& a, float& zn, float4& an, float m2, float z2,
                                            float4 a2, float zn2, float4 an2) {
  const float mn = fmaxf(m, m2);
  const float sa = __expf(m - mn), sb = __expf(m2 - mn);
  z = z * sa + z2 * sb;
  a.x = a.x * sa + a2.x * sb; a.y = a.y * sa + a2.y * sb;
  a.z = a.z * sa + a2.z * sb; a.w = a.w * sa + a2.w * sb;
  if constexpr (AUX) {
    zn = zn * sa + zn2 * sb;
    an.x = an.x * sa + an2.x * sb; an.y = an.y * sa + an2.y * sb;
    an.z = an.z * sa + an2.z * sb; an.w = an.w * sa + an2.w * sb;
  }
  m = mn;
}

// per-row state of the online softmax; the AUX half exists only in the training forward
template <int VEC, bool AUX>
struct RowState {
  float m[VEC], z[VEC];
  float4 acc[VEC];
  float zn[AUX ? VEC : 1];
  float4 accn[AUX ? VEC : 1];
  __device__ __forceinline__ void reset() {
#pragma unroll
    for (int v = 0; v < VEC; ++v) { m[v] = NEG_BIG; z[v] = 0.f; acc[v] = make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
    for (int v = 0; v < (AUX ? VEC : 1); ++v) { zn[v] = 0.f; accn[v] = make_float4(0.f, 0.f, 0.f, 0.f); }
  }
};
template <bool AUX>
__host__ __device__ __forceinline__ int64_t part_stride(const RowShape& rs) { return AUX ? 2 * (int64_t)rs.R + 3 * rs.H : (int64_t)rs.R + 2 * rs.H; }


// normalise, epilogue (skip, ELU) and stores of a finished row i (all lanes of the group)
// CR > 0 (with LPH > 0, VEC == 1): the row tables are dense with CR floats per row, heads of 4 LPH columns, F' == Fp --
// strides become constants (shifts instead of 32/64-bit multiplies, which run at a quarter of the VALU rate)
template <int LPR, int VEC, bool AUX, int LPH = 0, int CR = 0>
__device__ __forceinline__ void fwd_finish(const FwdArgs& a, const LaneCols<VEC>& lc, int i, const RowState<VEC, AUX>& st) {
  const int Fo = CR ? 4 * LPH : a.rs.Fo, Fp = CR ? 4 * LPH : a.rs.Fp;
  const int64_t ldr = CR ? CR : a.rs.ldr, ldh = CR ? CR / (4 * (LPH ? LPH : 1)) : a.rs.ldh, ldo = CR ? CR : a.rs.ldo;
  // AUX: a (row, head) needs aneg / qneg only if its edges lie on BOTH branches of the LeakyReLU -- all on the
  // identity branch: both are 0; all on the alpha branch: zn went through the very operations of z (bitwise equal)
  // and ds_i = -(1 - slope) sum_neg de = -(1 - slope) sum_all de = 0.  Such heads get qneg = 0 (K3a reads that as
  // "ds = 0, ignore aneg"), and a row none of whose heads is mixed does not write its aneg row at all: 56 % of the
  // rows of the R-MAT workload (tools/diag/mixed_rows.py; 55 % of its nodes have nothing but their self loop).
  bool mixed[VEC];
  bool row_mixed = false;
  if constexpr (AUX) {
    bool any = false;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      mixed[v] = lc.valid[v] && st.zn[v] != 0.f && st.zn[v] != st.z[v];
      any = any || mixed[v];
    }
    row_mixed = row_any<LPR>(any);
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    if (!lc.valid[v]) continue;
    const int co = lc.cofs[v], h = lc.head[v];
    const float rz = 1.0f / st.z[v];
    float4 hat = make_float4(st.acc[v].x * rz, st.acc[v].y * rz, st.acc[v].z * rz, st.acc[v].w * rz);
    if (a.hattn) st4(a.hattn + (int64_t)i * ldr + co, hat);
    if constexpr (AUX)
      if (row_mixed)
        st4(a.aneg + (int64_t)i * ldr + co, make_float4(st.accn[v].x * rz, st.accn[v].y * rz, st.accn[v].z * rz, st.accn[v].w * rz));
    if (a.out) {
      const int64_t io = a.g.urow ? (int64_t)a.g.urow[i] : (int64_t)i;   // `out` is the caller's: its row of internal node i
      float4 pre = hat;
      if (a.flags & PYGAT_F_SKIP) {
        float4 k4 = ld4(a.sk + (int64_t)i * ldr + co);
        pre.x += k4.x; pre.y += k4.y; pre.z += k4.z; pre.w += k4.w;
      }
      if (a.flags & PYGAT_F_ELU) { pre.x = elu1(pre.x); pre.y = elu1(pre.y); pre.z = elu1(pre.z); pre.w = elu1(pre.w); }
      if (Fo == Fp) {
        st4(a.out + io * ldo + co, pre);
      } else {
        const int f0 = co & (Fp - 1);
        float* o = a.out + io * ldo + (int64_t)h * Fo + f0;
        if (f0 + 0 < Fo) o[0] = pre.x;
        if (f0 + 1 < Fo) o[1] = pre.y;
        if (f0 + 2 < Fo) o[2] = pre.z;
        if (f0 + 3 < Fo) o[3] = pre.w;
      }
    }
    if (a.m && ((co >> 2) & (a.rs.lph - 1)) == 0) {
      a.m[(int64_t)i * ldh + h] = st.m[v];
      a.Z[(int64_t)i * ldh + h] = st.z[v];
      if constexpr (AUX) a.qneg[(int64_t)i * ldh + h] = mixed[v] ? st.zn[v] * rz : 0.f;
    }
  }
}

// partial record: [acc R | m H | z H] and, AUX, [accn R | zn H]
template <int VEC, bool AUX>
__device__ __forceinline__ void part_store(const FwdArgs& a, const LaneCols<VEC>& lc, float* p, const RowState<VEC, AUX>& st) {
  const int R = a.rs.R, H = a.rs.H;
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    if (!lc.valid[v]) continue;
    st4(p + lc.cofs[v], st.acc[v]);
    if constexpr (AUX) st4(p + R + 2 * H + lc.cofs[v], st.accn[v]);
    if (((lc.cofs[v] >> 2) & (a.rs.lph - 1)) == 0) {
      p[R + lc.head[v]] = st.m[v];
      p[R + H + lc.head[v]] = st.z[v];
      if constexpr (AUX) p[2 * R + 2 * H + lc.head[v]] = st.zn[v];
    }
  }
}
// a partial record in registers (loads first, merges afterwards: PF records are in flight together)
template <int VEC, bool AUX>
struct PartRec {
  float m[VEC], z[VEC];
  float4 acc[VEC];
  float zn[AUX ? VEC : 1];
  float4 accn[AUX ? VEC : 1];
};
template <int VEC, bool AUX>
__device__ __forceinline__ void part_load(const FwdArgs& a, const LaneCols<VEC>& lc, const float* p, PartRec<VEC, AUX>& r) {
  const int R = a.rs.R, H = a.rs.H;
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    r.m[v] = p[R + lc.head[v]]; r.z[v] = p[R + H + lc.head[v]]; r.acc[v] = ld4(p + lc.cofs[v]);
    if constexpr (AUX) { r.zn[v] = p[2 * R + 2 * H + lc.head[v]]; r.accn[v] = ld4(p + R + 2 * H + lc.cofs[v]); }
  }
}
template <int VEC, bool AUX>
__device__ __forceinline__ void part_merge(RowState<VEC, AUX>& st, const PartRec<VEC, AUX>& r) {
#pragma unroll
  for (int v = 0; v < VEC; ++v)
    merge_state<AUX>(st.m[v], st.z[v], st.acc[v], st.zn[AUX ? v : 0], st.accn[AUX ? v : 0], r.m[v], r.z[v], r.acc[v],
                     r.zn[AUX ? v : 0], r.accn[AUX ? v : 0]);
}

// row finished inside the slot -> final stores; row continuing in a neighbour slot -> partial record
template <int LPR, int VEC, bool AUX, int LPH = 0, int CR = 0>
__device__ __forceinline__ void fwd_flush(const FwdArgs& a, const LaneCols<VEC>& lc, int64_t k, int i,
                                          bool is_head, bool is_tail, const RowState<VEC, AUX>& st) {
  if (is_head || is_tail) {
    part_store<VEC, AUX>(a, lc, a.part + (2 * k + (is_head ? 0 : 1)) * part_stride<AUX>(a.rs), st);
  } else {
    fwd_finish<LPR, VEC, AUX, LPH, CR>(a, lc, i, st);
  }
}

// V2 = the reference's SpGraphAttentionLayerV2 scoring (layers.py:280-283): e_ij = a . LeakyReLU(Whi_i +
// Whj_j) per head, aggregation of Whi_j (layers.py:296); the gathered row is [Whi_j | Whj_j].
// FAST = no attention mask and every gathered table below 4 GiB: element offsets are 32-bit (a scalar base plus
// a 32-bit lane offset per load instead of a 64-bit multiply-add per address) and the mask registers vanish --
// 9-10 VGPRs less, which is what keeps the training forward (AUX) at 5 waves per SIMD.
// LPH > 0: lanes per head known at compile time (0: read from the shape) -- the per-edge head sums are DPP chains whose
// length otherwise costs a scalar branch per step, six per edge.
#ifndef PYGAT_K2_PREFETCH_ALL
#define PYGAT_K2_PREFETCH_ALL 0   // experiment (tools/build_variant.sh): edge-record prefetch on EVERY one-chunk instantiation
#endif
#ifndef PYGAT_K2_PREFETCH_TRAIN
#define PYGAT_K2_PREFETCH_TRAIN 1 // edge-record prefetch on the training forward (AUX && FAST, one chunk per lane)
#endif
#ifndef PYGAT_K2_HEADLINE_WAVES
#define PYGAT_K2_HEADLINE_WAVES 4 // (round 4: 5 waves at 96 VGPRs + 12 bytes of scratch without the prefetch, 1.04 ms; 4 waves at 104
#endif                            //  VGPRs with it, 1.00 ms -- same box, gpurun_out r4a; the prefetch at 5 waves spills 24 bytes: 1.025)
#ifndef PYGAT_K2_DEEP
#define PYGAT_K2_DEEP 0   // (experiment, tools/build_variant.sh: rows of round r + 1 gathered while round r folds, 16- / 8-lane rows)
#endif
#ifndef PYGAT_K2_NARROW_U
#define PYGAT_K2_NARROW_U 4   // (8 was measured: 0.32 -> 0.30 ms at one head of 16, 140 registers)
#endif
template <int LPR, int VEC, bool V2, bool AUX, bool FAST, int LPH = 0, int CR = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((AUX && FAST && VEC == 1 && LPR > 8) ? PYGAT_K2_HEADLINE_WAVES : 1))) void gat_fwd_kernel(FwdArgs a) {
  constexpr int EPW = 64 / LPR;
  // Narrow rows (LPR <= 8: at most 32 floats -- one or two 16-float heads, the shard of an 8- or 4-GPU head-parallel run):
  // a gather is one or two sectors, a wave carries 8-16 slots and the whole grid is ~20 waves per SIMD, so the kernel is
  // bound by the chain of dependent memory round trips a slot walks through (SQ counters at one head: 12 % VALU-active,
  // 60 % waiting, 3.4 waves per SIMD).  There: 8 edges per round instead of 4, and the edge records of the NEXT round
  // are fetched while this round's rows are in flight -- 1 serial round trip per 8 edges instead of 2 per 4.
  constexpr bool NARROW = (VEC == 1 && LPR <= 8);
  constexpr bool PREF = NARROW || (PYGAT_K2_PREFETCH_ALL && VEC == 1) || (PYGAT_K2_PREFETCH_TRAIN && AUX && FAST && VEC == 1);   // edge records of the next round fetched a round ahead
  constexpr int U = (VEC == 1) ? (NARROW ? PYGAT_K2_NARROW_U : 4) : 2;
  const int lane = threadIdx.x & 63;
  const int64_t kl = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * EPW + lane / LPR;
  if (kl >= a.g.kn) return;  // lane groups are independent: no cross-lane op below
  const int64_t k = a.g.order ? a.g.order[kl] : a.g.k0 + kl;  // slot id (slot_order: the whole range, k0 = 0)
  int64_t e0, e1;
  const int2* __restrict__ rc = a.g.rc;
  int r_first;
  bool head_partial, tail_known = false, tail_flag = false;
  if (a.g.meta) {   // (wave-uniform) one record instead of the chain slot_begin -> edge_rc -> rowptr
    const int4 mt = a.g.meta[k];
    e0 = mt.x; e1 = mt.y; r_first = mt.z;
    head_partial = (mt.w & 1) != 0; tail_known = true; tail_flag = (mt.w & 2) != 0;
  } else {
    slot_range(a.g, k, &e0, &e1);
    r_first = rc[e0].x;
    head_partial = a.g.rowptr[r_first] < e0;
  }
  const LaneCols<VEC> lc = lane_cols<LPR, VEC>(a.rs);
  const int R = CR ? CR : a.rs.R;
  const int64_t ldh = CR ? CR / (4 * (LPH ? LPH : 1)) : a.rs.ldh;

  int cur = r_first;
  const int lph = LPH > 0 ? LPH : (a.rs.lph < 64 ? a.rs.lph : 64);
  float4 adst[VEC];  // this lane's slice of a_dst (zero on padded / invalid chunks)
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    adst[v] = V2 ? ld4(a.a_pad + (int64_t)lc.head[v] * a.rs.Fp + (lc.cofs[v] & (a.rs.Fp - 1)))
                 : ld4(a.a_pad + (int64_t)lc.head[v] * 2 * a.rs.Fp + a.rs.Fp + (lc.cofs[v] & (a.rs.Fp - 1)));
    if (!lc.valid[v]) adst[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int64_t ldw = CR ? CR : a.ldwh;
  RowState<VEC, AUX> st;
  st.reset();

  // DEEP (rows of 16 / 8 lanes: the 2- and 4-head shards of a head-parallel run): the rows of round r + 1 are gathered while
  // round r folds -- a wave carries 4 / 8 lane groups whose rounds otherwise end with the slowest of 16 / 32 gathers, and the
  // line-request rate sat at 35-39 G/s against 47-49 at one and at eight heads (DESIGN_HISTORY.md section 8)
  constexpr bool DEEP = PYGAT_K2_DEEP && !V2 && AUX && FAST && VEC == 1 && (LPR == 16 || LPR == 8);
  if constexpr (DEEP) {
    int2 pc[U], px[U];
    float svn[U];
    float4 wvn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) pc[u] = rc[(e0 + u < e1) ? e0 + u : e1 - 1];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      wvn[u] = ld4(a.Wh + (uint32_t)((uint32_t)pc[u].y * (uint32_t)ldw + (uint32_t)lc.cofs[0]));
      svn[u] = a.s[(uint32_t)((uint32_t)pc[u].x * (uint32_t)ldh + (uint32_t)lc.head[0])];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) px[u] = rc[(e0 + U + u < e1) ? e0 + U + u : e1 - 1];
    for (int64_t e = e0; e < e1; e += U) {
      int2 p[U];
      float sv[U], tv[U];
      float4 wv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { p[u] = pc[u]; wv[u] = wvn[u]; sv[u] = svn[u]; pc[u] = px[u]; }
#pragma unroll
      for (int u = 0; u < U; ++u) {   // round r + 1's rows: their records were fetched a round ago
        wvn[u] = ld4(a.Wh + (uint32_t)((uint32_t)pc[u].y * (uint32_t)ldw + (uint32_t)lc.cofs[0]));
        svn[u] = a.s[(uint32_t)((uint32_t)pc[u].x * (uint32_t)ldh + (uint32_t)lc.head[0])];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) px[u] = rc[(e + 2 * U + u < e1) ? e + 2 * U + u : e1 - 1];
#pragma unroll
      for (int u = 0; u < U; ++u) tv[u] = group_sum_rt(dot4(wv[u], adst[0]), lph);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (e + u < e1) {
          if (p[u].x != cur) {
            fwd_flush<LPR, VEC, AUX, LPH, CR>(a, lc, k, cur, cur == r_first && head_partial, false, st);
            cur = p[u].x;
            st.reset();
          }
          const float zz = sv[u] + tv[u];
          fold_edge<AUX>(st.m[0], st.z[0], st.acc[0], st.zn[0], st.accn[0], lrelu(zz, a.alpha), !(zz > 0.f), wv[u], 1.f);
        }
      }
    }
    const bool tail_partial_d = tail_known ? tail_flag : a.g.rowptr[cur + 1] > e1;
    fwd_flush<LPR, VEC, AUX, LPH, CR>(a, lc, k, cur, cur == r_first && head_partial, tail_partial_d, st);
    return;
  }
  int2 pn[PREF ? U : 1];
  if constexpr (PREF) {
#pragma unroll
    for (int u = 0; u < U; ++u) pn[u] = rc[(e0 + u < e1) ? e0 + u : e1 - 1];
  }
  for (int64_t e = e0; e < e1; e += U) {
    int2 p[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (PREF) p[u] = pn[u];
      else p[u] = rc[(e + u < e1) ? e + u : e1 - 1];
    }
    // All U edge records are ISSUED before the first one is used.  Round 3's build of the headline instantiation had lost
    // this: hipcc sank record 0's first use above the loads of records 1..3 and waited vmcnt(0) for it -- one more dependent
    // memory round trip per 4 edges, K2 1.04 -> 1.11 ms on the same box (same-lease A/B, profiles/r4ab_*).
    if constexpr (!PREF) __builtin_amdgcn_sched_barrier(0);
    float sv[U][VEC], tv[U][VEC], mk[U][VEC];
    float4 wv[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        if constexpr (FAST) {
          mk[u][v] = 1.f;
          wv[u][v] = ld4(a.Wh + (uint32_t)((uint32_t)p[u].y * (uint32_t)ldw + (uint32_t)lc.cofs[v]));
        } else {
          mk[u][v] = a.mask ? a.mask[((e + u < e1) ? e + u : e1 - 1) * ldh + lc.head[v]] : 1.f;
          wv[u][v] = ld4(a.Wh + (int64_t)p[u].y * ldw + lc.cofs[v]);
        }
        if constexpr (V2) {
          const float4 wi = ld4(a.Wh + (int64_t)p[u].x * ldw + lc.cofs[v]);      // Whi_i (row-local)
          const float4 wj = ld4(a.Wh + (int64_t)p[u].y * ldw + R + lc.cofs[v]);  // Whj_j (same gathered row)
          const float4 hh = make_float4(wi.x + wj.x, wi.y + wj.y, wi.z + wj.z, wi.w + wj.w);
          const float4 ll = make_float4(lrelu(hh.x, a.alpha), lrelu(hh.y, a.alpha), lrelu(hh.z, a.alpha), lrelu(hh.w, a.alpha));
          sv[u][v] = dot4(ll, adst[v]);   // partial of e_ij over this lane's 4 features
        } else if constexpr (FAST) {
          sv[u][v] = a.s[(uint32_t)((uint32_t)p[u].x * (uint32_t)ldh + (uint32_t)lc.head[v])];
        } else {
          sv[u][v] = a.s[(int64_t)p[u].x * ldh + lc.head[v]];
        }
      }
    if constexpr (PREF) {   // next round's edge records: in flight together with this round's rows
#pragma unroll
      for (int u = 0; u < U; ++u) pn[u] = rc[(e + U + u < e1) ? e + U + u : e1 - 1];
    }
    // per-head sums over the lanes of a head (all lanes of the group are active here):
    // V1: t_j = Wh_j . a_dst from the row just gathered; V2: the logit e_ij itself
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        if constexpr (V2) { tv[u][v] = group_sum_rt(sv[u][v], lph); }
        else { tv[u][v] = group_sum_rt(dot4(wv[u][v], adst[v]), lph); }
      }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (e + u < e1) {
        if (p[u].x != cur) {
          fwd_flush<LPR, VEC, AUX, LPH, CR>(a, lc, k, cur, cur == r_first && head_partial, false, st);
          cur = p[u].x;
          st.reset();
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const float zz = sv[u][v] + tv[u][v];        // V1: the logit before the LeakyReLU
          fold_edge<AUX>(st.m[v], st.z[v], st.acc[v], st.zn[AUX ? v : 0], st.accn[AUX ? v : 0],
                         V2 ? tv[u][v] : lrelu(zz, a.alpha), !(zz > 0.f), wv[u][v], mk[u][v]);
        }
      }
    }
  }
  const bool tail_partial = tail_known ? tail_flag : a.g.rowptr[cur + 1] > e1;
  fwd_flush<LPR, VEC, AUX, LPH, CR>(a, lc, k, cur, cur == r_first && head_partial, tail_partial, st);
}

// Merge of one cut row: its pieces tail(k), head(k+1), ..., head(k + npieces - 1) are dealt round-robin to
// nw (1, or the NW waves of the work-group) waves x EPW lane groups (PF pieces in flight each -- the chain of a 26k-edge row has 400+ pieces), combined
// inside a wave with shuffles and across waves through LDS, always in the same order (reproducible).
// G: lane groups that share one row (64 / LPR = the whole wave: one row per wave; fewer: 64 / LPR / G rows side by side, each
// lane group with its own k, r, npieces -- the packed entries of the list-driven kernel, never `wide`).
template <int LPR, int VEC, bool AUX, int NW, int G = 64 / LPR>
__device__ __forceinline__ void fwd_merge_row(const FwdArgs& a, const LaneCols<VEC>& lc, float* fix_sm, int64_t k, int r,
                                              int npieces, bool wide, int w) {
  constexpr int PF = (VEC == 1) ? 4 : 2;
  const int lane = threadIdx.x & 63, slot = lane / LPR, g = slot % G;
  const int64_t PS = part_stride<AUX>(a.rs);
  RowState<VEC, AUX> st;
  st.reset();
  if (wide || w == 0) {
    const int nw = wide ? NW : 1;
    for (int q = (wide ? w : 0) * G * PF + g; q < npieces; q += nw * G * PF) {
      PartRec<VEC, AUX> rec[PF];
#pragma unroll
      for (int f = 0; f < PF; ++f) {
        const int qq = q + f * G;
        const int qc = qq < npieces ? qq : q;   // clamped: the loads stay unconditional
        part_load<VEC, AUX>(a, lc, a.part + (qc == 0 ? 2 * k + 1 : 2 * (k + qc)) * PS, rec[f]);
      }
#pragma unroll
      for (int f = 0; f < PF; ++f)
        if (q + f * G < npieces) part_merge<VEC, AUX>(st, rec[f]);
    }
#pragma unroll
    for (int off = LPR; off < LPR * G; off <<= 1) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const float mo = __shfl_xor(st.m[v], off), zo = __shfl_xor(st.z[v], off);
        float4 ao, no = make_float4(0.f, 0.f, 0.f, 0.f);
        float zno = 0.f;
        ao.x = __shfl_xor(st.acc[v].x, off); ao.y = __shfl_xor(st.acc[v].y, off);
        ao.z = __shfl_xor(st.acc[v].z, off); ao.w = __shfl_xor(st.acc[v].w, off);
        if constexpr (AUX) {
          zno = __shfl_xor(st.zn[v], off);
          no.x = __shfl_xor(st.accn[v].x, off); no.y = __shfl_xor(st.accn[v].y, off);
          no.z = __shfl_xor(st.accn[v].z, off); no.w = __shfl_xor(st.accn[v].w, off);
        }
        merge_state<AUX>(st.m[v], st.z[v], st.acc[v], st.zn[AUX ? v : 0], st.accn[AUX ? v : 0], mo, zo, ao, zno, no);
      }
    }
  }
  if (wide) {  // wave partials -> LDS -> wave 0 (uniform branch: `wide` is the same in all waves)
    if (slot == 0) part_store<VEC, AUX>(a, lc, fix_sm + w * PS, st);
    __syncthreads();
    if (w == 0 && slot == 0) {
#pragma unroll
      for (int ww = 1; ww < NW; ++ww) {
        PartRec<VEC, AUX> rec;
        part_load<VEC, AUX>(a, lc, fix_sm + ww * PS, rec);
        part_merge<VEC, AUX>(st, rec);
      }
    }
    __syncthreads();
  }
  if (w == 0 && g == 0 && npieces > 0) fwd_finish<LPR, VEC, AUX>(a, lc, r, st);
}

// Fix-up of the rows cut by a slot border.  A work-group screens FIX_SCREEN consecutive slots: slot k
// OWNS a cut row if its last row starts inside k and continues beyond.  Every owned row is merged in turn.
template <int LPR, int VEC, bool AUX>
__global__ __launch_bounds__(256) void gat_fwd_fixup_kernel(FwdArgs a) {
  constexpr int EPW = 64 / LPR;
  constexpr int PF = (VEC == 1) ? 4 : 2;
  extern __shared__ __attribute__((aligned(16))) float fix_sm[];  // [4][PS]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t kbase = a.g.k0 + (int64_t)blockIdx.x * FIX_SCREEN;
  const int64_t nslots = a.g.k0 + a.g.kn;
  int my_r = 0, my_end = 0;
  bool owner = false;
  if (lane < FIX_SCREEN && kbase + lane < nslots) {
    int64_t e0, e1;
    slot_range(a.g, kbase + lane, &e0, &e1);
    my_r = a.g.rc[e1 - 1].x;
    my_end = a.g.rowptr[my_r + 1];
    owner = (int64_t)my_end > e1 && (int64_t)a.g.rowptr[my_r] >= e0;
  }
  unsigned long long todo = __ballot(owner);  // identical in the 4 waves
  const LaneCols<VEC> lc = lane_cols<LPR, VEC>(a.rs);
  while (todo) {
    const int src = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int64_t k = kbase + src;
    const int r = __shfl(my_r, src);
    const int64_t row_end = __shfl(my_end, src);
    const int64_t k_e = slot_of(a.g, row_end - 1);
    const int npieces = (int)(k_e - k) + 1;
    const bool wide = npieces > EPW * PF;  // more pieces than one wave takes in a single round
    fwd_merge_row<LPR, VEC, AUX, 4>(a, lc, fix_sm, k, r, npieces, wide, w);
  }
}

// list-driven variant: entry q of g.cut = (owner slot k, row, pieces); the first n_cut_wide entries (long
// chains) get a whole work-group of FIX_LIST_WAVES waves each (the 26 779-edge hub of config 5 is a chain of 420
// pieces and the critical path of this launch: 16 waves take it in 4 rounds instead of 13), the others one wave each.
template <int LPR, int VEC, bool AUX>
__global__ __launch_bounds__(64 * FIX_LIST_WAVES) void gat_fwd_fixup_list_kernel(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float fix_sm[];  // [FIX_LIST_WAVES][PS]
  constexpr int RPW = fix_rows_per_wave(LPR), G = 64 / LPR / RPW;
  const int w = threadIdx.x >> 6;
  const bool wide = (int)blockIdx.x < a.g.n_cut_wide;
  const LaneCols<VEC> lc = lane_cols<LPR, VEC>(a.rs);
  if (wide) {
    const int q0 = (int)blockIdx.x;
    fwd_merge_row<LPR, VEC, AUX, FIX_LIST_WAVES>(a, lc, fix_sm, a.g.cut[3 * q0], a.g.cut[3 * q0 + 1], a.g.cut[3 * q0 + 2], true, w);
    return;
  }
  // RPW rows per wave, G lane groups each; a lone wave plays wave 0 of the merge
  const int q0 = a.g.n_cut_wide + (((int)blockIdx.x - a.g.n_cut_wide) * FIX_LIST_WAVES + w) * RPW + ((threadIdx.x & 63) / LPR) / G;
  const bool have = q0 < a.g.n_cut;
  const int qc = have ? q0 : a.g.n_cut - 1;
  fwd_merge_row<LPR, VEC, AUX, FIX_LIST_WAVES, G>(a, lc, fix_sm, a.g.cut[3 * qc], a.g.cut[3 * qc + 1], have ? a.g.cut[3 * qc + 2] : 0,
                                                  false, 0);
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

// K0 helper: (row, col) pair per edge
__global__ __launch_bounds__(256) void edge_pairs_kernel(int n, const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ col, int2* __restrict__ rc) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  for (int k = rowptr[i] + lane; k < rowptr[i + 1]; k += 64) rc[k] = make_int2(i, col[k]);
}

}  // namespace pygat

using namespace pygat;

extern "C" int pygat_edge_pairs(int n, const int32_t* rowptr, const int32_t* col, int32_t* edge_rc, void* stream) {
  PYGAT_REQUIRE(n > 0 && rowptr && col && edge_rc, "edge_pairs: bad arguments");
  hipLaunchKernelGGL(edge_pairs_kernel, dim3((unsigned)cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, n, rowptr, col,
                     reinterpret_cast<int2*>(edge_rc));
  PYGAT_CHECK_LAUNCH("edge_pairs");
  return PYGAT_OK;
}

extern "C" size_t pygat_partials_bytes(int64_t nnz, int slot_edges, int H, int Fp) {
  if (nnz <= 0 || slot_edges <= 0) return 0;
  const int64_t nslots = (nnz + slot_edges - 1) / slot_edges;
  // sized for the widest record any pass writes: [acc R | m H | z H | accn R | zn H] (training forward)
  return (size_t)(2 * nslots) * (size_t)(2 * H * Fp + 3 * H) * sizeof(float);
}

static int launch_forward(const pygat_graph* g, int H, int Fo, float alpha, int flags, const float* Wh, int v2,
                          const float* s, const float* a_pad, const float* sk, const float* att_mask, float* out,
                          float* hattn, float* m, float* Z, float* aneg, float* qneg, void* part, void* stream) {
  FwdArgs a;
  int rc = check_graph(g, &a.g, /*allow_slot_range=*/v2 ? 2 : 1);   // (GATv2: all slots, or the prefix before a self-loop-only tail)
  if (rc) return rc;
  const int Fp = padded_width(Fo);
  PYGAT_REQUIRE(H > 0 && Fp > 0, "gat_forward: unsupported H=%d F'=%d", H, Fo);
  PYGAT_REQUIRE(Wh && (s || v2) && a_pad && part, "gat_forward: null Wh/s/a_pad/part");
  PYGAT_REQUIRE(out || hattn, "gat_forward: need out and/or hattn");
  PYGAT_REQUIRE(!(flags & PYGAT_F_SKIP) || sk, "gat_forward: PYGAT_F_SKIP without sk");
  PYGAT_REQUIRE((m == nullptr) == (Z == nullptr), "gat_forward: m and Z must be given together");
  PYGAT_REQUIRE((aneg == nullptr) == (qneg == nullptr) && (!aneg || (m && !v2 && aligned16(aneg))),
                "gat_forward: aneg and qneg come together, need m/Z, 16-byte alignment and the v1 layer");
  PYGAT_REQUIRE(aligned16(Wh) && aligned16(part) && aligned16(a_pad) && (!sk || aligned16(sk)) && (!hattn || aligned16(hattn)) &&
                    (!out || Fo != Fp || aligned16(out)),
                "gat_forward: row tables must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t nslots = a.g.kn;   // slots of this call (a row range, or all)
  // GATv2 gathers [Whi|Whj] rows and is not windowed
  const int hg = v2 ? H : head_group_fwd_n(a.g.n, H, Fp);
  // one phase of the pass only (PYGAT_F_MAIN_ONLY / PYGAT_F_FIXUP_ONLY: a row-chunk pipeline runs chunk c's fix-up on a second
  // stream beside chunk c + 1's main launch): the partial records are indexed by slot, so chunks never share one -- head
  // WINDOWS do (the launches of a window reuse `part` in stream order), hence one window only
  const bool do_main = !(flags & PYGAT_F_FIXUP_ONLY), do_fix = !(flags & PYGAT_F_MAIN_ONLY);
  PYGAT_REQUIRE(do_main || do_fix, "gat_forward: PYGAT_F_MAIN_ONLY and PYGAT_F_FIXUP_ONLY exclude each other");
  PYGAT_REQUIRE((do_main && do_fix) || hg >= H, "gat_forward: a single phase needs the level in one head window (pygat_gat_forward_phases_ok)");
  flags &= ~(PYGAT_F_MAIN_ONLY | PYGAT_F_FIXUP_ONLY);
  for (int h0 = 0; h0 < H; h0 += hg) {
    const int hc = (H - h0 < hg) ? H - h0 : hg;
    PYGAT_REQUIRE(make_window_shape(H, Fo, hc, &a.rs),
                  "gat_forward: unsupported H=%d F'=%d (a pass takes rows of at most 1024 floats)", hc, Fo);
    a.alpha = alpha; a.flags = flags;
    a.Wh = Wh + (int64_t)h0 * Fp; a.ldwh = v2 ? 2 * a.rs.ldr : a.rs.ldr;
    a.s = s ? s + h0 : nullptr;
    a.a_pad = a_pad + (int64_t)h0 * (v2 ? 1 : 2) * Fp;
    a.sk = sk ? sk + (int64_t)h0 * Fp : nullptr;
    a.mask = att_mask ? att_mask + h0 : nullptr;
    a.out = out ? out + (int64_t)h0 * Fo : nullptr;
    a.hattn = hattn ? hattn + (int64_t)h0 * Fp : nullptr;
    a.m = m ? m + h0 : nullptr; a.Z = Z ? Z + h0 : nullptr;
    a.aneg = aneg ? aneg + (int64_t)h0 * Fp : nullptr; a.qneg = qneg ? qneg + h0 : nullptr;
    a.part = (float*)part;   // reused by the windows: the launches are ordered on the stream
    int lpr, vec;
    pick_lanes(a.rs, &lpr, &vec);
    // narrow rows (a wave carries 8-64 slots, the whole grid is a few ten waves per SIMD): one-wave work-groups, so that a
    // SIMD slot is refilled as soon as ITS wave ends instead of when the slowest of four does
    const unsigned bt = (vec == 1 && lpr <= 8) ? narrow_block() : 256u;
    const unsigned blocks = (unsigned)cdiv(cdiv(nslots, 64 / lpr), bt / 64);
    const bool aux = aneg != nullptr;
    // 32-bit element offsets: the gathered table and s below 2^32 bytes
    // measured (config 5, same box): the training forward 1.147 -> 1.114 ms (5 instead of 4 waves per SIMD), the plain
    // forward 0.93 -> 0.98 ms (same occupancy, the 32-bit form is the slower one): FAST only where it buys occupancy
    const bool fast = aux && !att_mask && (int64_t)a.g.n * a.ldwh * 4 < ((int64_t)1 << 32);
#define PYGAT_FWD(V2V, AUXV, FASTV)                                                                                   \
    PYGAT_DISPATCH_LANES(lpr, vec, hipLaunchKernelGGL((gat_fwd_kernel<LPR, VEC, V2V, AUXV, FASTV>), dim3(blocks),    \
                                                      dim3(bt), 0, st, a))
    if (!do_main) {}
    else if (v2) PYGAT_FWD(true, false, false);
    else if (aux && fast && lpr == 32 && vec == 1 && a.rs.lph == 4)   // 8 heads x 16: the headline shape
      if (a.rs.R == 128 && a.rs.ldr == 128 && a.rs.ldh == 8 && a.rs.ldo == 128 && a.rs.Fo == 16 && a.ldwh == 128 && !(flags & PYGAT_F_SKIP))
        hipLaunchKernelGGL((gat_fwd_kernel<32, 1, false, true, true, 4, 128>), dim3(blocks), dim3(bt), 0, st, a);
      else
        hipLaunchKernelGGL((gat_fwd_kernel<32, 1, false, true, true, 4>), dim3(blocks), dim3(bt), 0, st, a);
    else if (aux && fast) PYGAT_FWD(false, true, true);
    else if (aux) PYGAT_FWD(false, true, false);
    else PYGAT_FWD(false, false, false);
#undef PYGAT_FWD
    PYGAT_CHECK_LAUNCH("gat_forward");
    const bool listed = a.g.cut != nullptr;   // the caller listed the cut rows: go straight to them
    const int fix_waves = listed ? FIX_LIST_WAVES : 4;
    const size_t fix_lds = fix_waves * (size_t)(aux ? part_stride<true>(a.rs) : part_stride<false>(a.rs)) * sizeof(float);
    const unsigned fb = listed ? (unsigned)(a.g.n_cut_wide + cdiv(a.g.n_cut - a.g.n_cut_wide, FIX_LIST_WAVES * fix_rows_per_wave(lpr)))
                               : (unsigned)cdiv(nslots, FIX_SCREEN);
    if ((listed && a.g.n_cut == 0) || !do_fix) continue;
#define PYGAT_FIX(AUXV)                                                                                               \
    do {                                                                                                              \
      if (listed) {                                                                                                   \
        PYGAT_DISPATCH_LANES(lpr, vec, hipLaunchKernelGGL((gat_fwd_fixup_list_kernel<LPR, VEC, AUXV>), dim3(fb),     \
                                                          dim3(64 * FIX_LIST_WAVES), fix_lds, st, a));                \
      } else {                                                                                                        \
        PYGAT_DISPATCH_LANES(lpr, vec, hipLaunchKernelGGL((gat_fwd_fixup_kernel<LPR, VEC, AUXV>), dim3(fb), dim3(256), \
                                                          fix_lds, st, a));                                           \
      }                                                                                                               \
    } while (0)
    if (aux) PYGAT_FIX(true); else PYGAT_FIX(false);
#undef PYGAT_FIX
    PYGAT_CHECK_LAUNCH("gat_forward_fixup");
  }
  return PYGAT_OK;
}

extern "C" int pygat_head_group(int n, int H, int Fo) {
  const int Fp = padded_width(Fo);
  if (n <= 0 || H <= 0 || Fp <= 0) return 0;
  return head_group_bwd(n, H, Fp);
}


// register / scratch footprint of a kernel of this file as the loaded code object reports it (pygat_kernel_footprint)
static int footprint_of(const void* fn, int* regs, int* scratch) {
  hipFuncAttributes at;
  const hipError_t e = hipFuncGetAttributes(&at, fn);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    pygat::set_error("kernel_footprint: %s", hipGetErrorString(e));
    return PYGAT_EHIP;
  }
  *regs = at.numRegs; *scratch = (int)at.localSizeBytes;
  return PYGAT_OK;
}
namespace pygat {
int footprint_k2_headline(int* regs, int* scratch) {
  return footprint_of(reinterpret_cast<const void*>(&gat_fwd_kernel<32, 1, false, true, true, 4, 128>), regs, scratch);
}
}  // namespace pygat

extern "C" int pygat_gat_forward_phases_ok(int n, int H, int Fo) {
  const int Fp = padded_width(Fo);
  if (n <= 0 || H <= 0 || Fp <= 0) return 0;
  return head_group_fwd_n(n, H, Fp) >= H ? 1 : 0;
}

extern "C" int pygat_gat_forward(const pygat_graph* g, int H, int Fo, float alpha, int flags, const float* Wh,
                                 const float* s, const float* a_pad, const float* sk, const float* att_mask,
                                 float* out, float* hattn, float* m, float* Z, float* aneg, float* qneg, void* part,
                                 void* stream) {
  return launch_forward(g, H, Fo, alpha, flags, Wh, 0, s, a_pad, sk, att_mask, out, hattn, m, Z, aneg, qneg, part, stream);
}

extern "C" int pygat_gatv2_forward(const pygat_graph* g, int H, int Fo, float alpha, int flags, const float* WW,
                                   const float* a2, const float* sk, const float* att_mask, float* out,
                                   float* hattn, float* m, float* Z, void* part, void* stream) {
  return launch_forward(g, H, Fo, alpha, flags, WW, 1, nullptr, a2, sk, att_mask, out, hattn, m, Z, nullptr, nullptr, part, stream);
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
