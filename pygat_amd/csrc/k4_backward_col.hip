// K4 -- backward, column pass: dWh, dt (gfx950, wave64).
//
// Replaces `grad_b = a.t().matmul(grad_output)` of SpecialSpmmFunction.backward (reference
// layers.py:89) -- the transposed SpMM -- plus the autograd of the two a-halves matmuls
// (layers.py:60-61):
//     dWh_j = sum_i alpha~_ij Gp_i + ds_j a_src + dt_j a_dst        alpha~ = alpha * dropout mask
//     dt_j  = sum_i dz_ij                                            (column sums of dz)
// nnz split over the TRANSPOSED pattern: a slot walks transposed edges (j <- i) and gathers ONE
// contiguous row per edge, GR_i = [Gp_i | (s_i, m_i, 1/Z_i, D_i) per head]; with the row-local Wh_j
// (and t_j = Wh_j . a_dst) it recomputes alpha_ij and dz_ij in registers instead of reading a
// per-edge buffer through the edge permutation.  ds comes from K3b.  Rows cut by a slot border go
// through `part` + a fix-up launch (fixed order).  No atomics.
#include "attn_common.h"

namespace pygat {

struct ColArgs {
  GraphDev g;  // transposed pattern: rc[k] = (j, i) for the forward edge (i, j)
  RowShape rs;
  float alpha;
  const int32_t* perm;  // transposed position -> forward edge (only to index the dropout mask)
  const float* mask;    // [nnz][H] attention dropout mask in FORWARD edge order, or nullptr
  const float* Wh;
  const float* GR;
  int64_t ldgr;         // row stride of GR
  const float* a_pad;
  const float* ds;      // row sums from the row pass K3b, or nullptr: then the ds_j a_src term is left out here
  float* dz_t;          // [nnz][Htot] or nullptr: dz of every TRANSPOSED edge, for the row-sum pass that follows
  float* dWh;
  float* dt;
  float* part;  // [2 * nslots][R + 2H]: acc[R], dt[H]
  float* da_part;  // DA instantiations: [work-groups][2 R] = (sum ds_j Wh_j | sum dt_j Wh_j) over the rows the work-group finished
};

#ifndef PYGAT_K4_DA_ATOMIC
#define PYGAT_K4_DA_ATOMIC 0
#endif
#ifndef PYGAT_K4_DA_WROW
#define PYGAT_K4_DA_WROW 0   // 1: the finished row's Wh from the registers of the round's previous edge where there is one --
                             // measured: the longer-lived rows cost 36 bytes of scratch at four waves (K4 1.26 -> 1.39 ms) or
                             // the fourth wave at 144 VGPRs (1.38 ms); the fetch in the flush stays (gpurun_out r4g)
#endif
#ifndef PYGAT_DIAG_K4
#define PYGAT_DIAG_K4 0   // tools/build_variant.sh only: bit 0 no LDS sums, bit 1 no Wh_j load either (then da is wrong, the time is the point)
#endif
__device__ __forceinline__ void lds_add(float* p, float v) {   // ds_add_f32 without a return value
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// CR > 0 (with LPH > 0, VEC == 1): dense row tables of CR floats, heads of 4 LPH columns -- constant strides
// DA (round 4): the attention-vector gradient da_src = sum_j ds_j Wh_j, da_dst = sum_j dt_j Wh_j (autograd of layers.py:60-61,
// a.mm(edge_h) layers.py:144) taken HERE, where row j finishes with dt_j just formed, ds_j loaded for the dWh term and Wh_j in
// L1 from the row's edges -- instead of by a pass of its own that streams Wh, ds and dt again (pygat_a_grad: 0.6 GB, 0.12 ms
// at config 5).  The 8 running sums of a lane live in LDS (da_lds[0], da_lds[da_stride]: its own two float4s, no other lane
// touches them), not in registers: the kernel sits at 122 of the 128 VGPRs that keep four waves per SIMD.
template <int VEC, int LPH = 0, int CR = 0, bool DA = false>
__device__ __forceinline__ void col_finish(const ColArgs& a, const LaneCols<VEC>& lc, int j,
                                           const float4 (&acc)[VEC], const float (&dt)[VEC], float4* da_lds = nullptr,
                                           int da_stride = 0, const float4* wrow = nullptr) {
  const int Fp = CR ? 4 * LPH : a.rs.Fp;
  const int64_t ldr = CR ? CR : a.rs.ldr, ldh = CR ? CR / (4 * (LPH ? LPH : 1)) : a.rs.ldh;
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    if (!lc.valid[v]) continue;
    const int co = lc.cofs[v], h = lc.head[v], f0 = co & (Fp - 1);
    const float dsj = a.ds ? a.ds[(int64_t)j * ldh + h] : 0.f;
    const float4 as = ld4(a.a_pad + (int64_t)h * 2 * Fp + f0);
    const float4 ad = ld4(a.a_pad + (int64_t)h * 2 * Fp + Fp + f0);
    // DA: Wh_j.  wrow: the caller still holds it in registers (the row's last edge was gathered in this round of U edges:
    // three flushes of four); else it is fetched HERE, with the other loads of the flush and before its stores -- vmcnt
    // counts loads and stores together in issue order (issued behind the dWh / dt stores, the wait for this load was a wait
    // for their acknowledgement from HBM: K4 1.21 -> 2.12 ms in the first build of this path, gpurun_out r4b).  The fetch is
    // not free even so: the row comes from L2, not L1 (16 waves of 640-byte gathers turn the 32 KB L1 over every round), and
    // one more 512-byte read per finished row cost K4 0.05 ms (diagnostic builds, gpurun_out r4f).
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (DA && !(PYGAT_DIAG_K4 & 2)) w = wrow ? wrow[v] : ld4(a.Wh + (int64_t)j * ldr + co);
    float4 o;
    o.x = acc[v].x + dsj * as.x + dt[v] * ad.x;
    o.y = acc[v].y + dsj * as.y + dt[v] * ad.y;
    o.z = acc[v].z + dsj * as.z + dt[v] * ad.z;
    o.w = acc[v].w + dsj * as.w + dt[v] * ad.w;
    if constexpr (DA) {
      static_assert(VEC == 1, "da accumulation: one chunk per lane");
      // Read-modify-write of the lane's own two float4s, ONE AFTER THE OTHER (the fence keeps hipcc from fetching both up
      // front: with both in flight the kernel's peak is 130 VGPRs = three waves per SIMD instead of four).  Only this lane
      // touches these eight words, in program order: the sums are as reproducible as register accumulators.
      // (ds_add_f32 instead -- one product register at a time, 128 VGPRs -- was measured: LDS float atomics retire about
      // one lane per clock, K4 1.20 -> 2.14 ms; gpurun_out r4b / r4c.)
#if PYGAT_DIAG_K4 & 1     /* diagnostic builds only: the Wh_j load without the LDS sums */
      asm volatile("" :: "v"(w.x), "v"(w.y), "v"(w.z), "v"(w.w));
#elif PYGAT_K4_DA_ATOMIC
      float* xs = reinterpret_cast<float*>(da_lds);
      float* ys = reinterpret_cast<float*>(da_lds + da_stride);
      lds_add(xs + 0, dsj * w.x); lds_add(xs + 1, dsj * w.y); lds_add(xs + 2, dsj * w.z); lds_add(xs + 3, dsj * w.w);
      lds_add(ys + 0, dt[v] * w.x); lds_add(ys + 1, dt[v] * w.y); lds_add(ys + 2, dt[v] * w.z); lds_add(ys + 3, dt[v] * w.w);
#else
      {
        float4 x = da_lds[0];
        x.x = fmaf(dsj, w.x, x.x); x.y = fmaf(dsj, w.y, x.y); x.z = fmaf(dsj, w.z, x.z); x.w = fmaf(dsj, w.w, x.w);
        da_lds[0] = x;
      }
      {
        float4 y = da_lds[da_stride];
        y.x = fmaf(dt[v], w.x, y.x); y.y = fmaf(dt[v], w.y, y.y); y.z = fmaf(dt[v], w.z, y.z); y.w = fmaf(dt[v], w.w, y.w);
        da_lds[da_stride] = y;
      }
#endif
    }
    st4(a.dWh + (int64_t)j * ldr + co, o);
    if (((co >> 2) & (a.rs.lph - 1)) == 0) a.dt[(int64_t)j * ldh + h] = dt[v];
  }
}

template <int VEC, int LPH = 0, int CR = 0, bool DA = false>
__device__ __forceinline__ void col_flush(const ColArgs& a, const LaneCols<VEC>& lc, int64_t q, int j,
                                          bool is_head, bool is_tail, const float4 (&acc)[VEC],
                                          const float (&dt)[VEC], float4* da_lds = nullptr, int da_stride = 0,
                                          const float4* wrow = nullptr) {
  if (is_head || is_tail) {
    // (the slot id is looked up again HERE, in the rare cut-row branch: kept live through the walk -- it no longer follows
    // from the block and thread ids alone once a slot_order is allowed -- it cost the 8 x 16 da instantiation 12 bytes of scratch)
    const int64_t k = slot_at(a.g, q);
    float* p = a.part + (2 * k + (is_head ? 0 : 1)) * (int64_t)(a.rs.R + 2 * a.rs.H);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      if (!lc.valid[v]) continue;
      st4(p + lc.cofs[v], acc[v]);
      if (((lc.cofs[v] >> 2) & (a.rs.lph - 1)) == 0) p[a.rs.R + lc.head[v]] = dt[v];
    }
  } else {
    col_finish<VEC, LPH, CR, DA>(a, lc, j, acc, dt, da_lds, da_stride, wrow);
  }
}

// one of U registers by a runtime index (U = 2 or 4)
template <int U, typename T>
__device__ __forceinline__ T pick(const T (&x)[U], int u) {
  T r = x[0];
#pragma unroll
  for (int q = 1; q < U; ++q) r = (u == q) ? x[q] : r;
  return r;
}

// WRITE_DZ: this pass computes every dz_ij anyway (for dt_j); written out per transposed edge (sequential
// 4H-byte records) they let the row sums ds_i = sum_j dz_ij be taken by a light pass that fetches the
// records through perm_f (pygat_gat_backward_rowsum) instead of the row pass K3b, which gathers a whole
// Wh_j row per forward edge just to recompute them.  (Scattering the records to their forward positions
// here, so that the row sums become a pure stream, was measured too: the random 32-byte stores cost K4
// 0.48 ms and saved 0.10 ms there.)
// The lanes of a head all hold dz for the U edges of a round: lane (u mod S) of the head stores edge u.
// LPH > 0: lanes per head known at compile time (0: read from the shape): the two head sums per edge are DPP chains
// whose length otherwise costs a scalar branch per step.
#ifndef PYGAT_K4_DA_WAVES
#define PYGAT_K4_DA_WAVES 4         // minimum waves per SIMD asked of the dense 8 x 16 instantiation that takes da along
#endif
#ifndef PYGAT_K4_HEADLINE_WAVES
#define PYGAT_K4_HEADLINE_WAVES 1   // (experiment: minimum waves per SIMD asked of the CR > 0 instantiation)
#endif
template <int LPR, int VEC, bool WRITE_DZ, int LPH, int CR, bool DA>
__device__ __forceinline__ void col_walk(const ColArgs& a, const int64_t q, float4* da_lds, int da_stride) {
  const int64_t k = slot_at(a.g, q);   // grid position -> slot (pygat_graph.slot_order; identity without one)
  // narrow rows: next round's edge records prefetched (see gat_fwd_kernel; 8 edges per round were measured too: two rows
  // per edge in registers leave 2 waves per SIMD at U = 8, 0.30 -> 0.35 ms at one head of 16)
#ifndef PYGAT_K4_PREFETCH_ALL
#define PYGAT_K4_PREFETCH_ALL 0   // experiment (tools/build_variant.sh): edge-record prefetch on one-chunk rows of any width
#endif
  constexpr bool NARROW = (VEC == 1 && LPR <= 8) || (PYGAT_K4_PREFETCH_ALL && VEC == 1);
  constexpr int U = (VEC == 1) ? 4 : 2;
  int64_t e0, e1;
  int r_first;
  bool head_partial, tail_known = false, tail_flag = false;
  const int2* __restrict__ rc = a.g.rc;
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
  constexpr int HC = CR / (4 * (LPH ? LPH : 1));   // heads, when CR > 0
  const int64_t RW = CR ? CR + 4 * HC : a.ldgr, ldr = CR ? CR : a.rs.ldr, ldh = CR ? HC : a.rs.ldh;
  const int lph = LPH > 0 ? LPH : (a.rs.lph < 64 ? a.rs.lph : 64);
  float4 adst[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    adst[v] = ld4(a.a_pad + (int64_t)lc.head[v] * 2 * a.rs.Fp + a.rs.Fp + (lc.cofs[v] & (a.rs.Fp - 1)));
    if (!lc.valid[v]) adst[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  int cur = r_first;
  float4 acc[VEC];
  float dt[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) { acc[v] = make_float4(0.f, 0.f, 0.f, 0.f); dt[v] = 0.f; }

  int2 pn[NARROW ? U : 1];
  if constexpr (NARROW) {
#pragma unroll
    for (int u = 0; u < U; ++u) pn[u] = rc[(e0 + u < e1) ? e0 + u : e1 - 1];
  }
  for (int64_t e = e0; e < e1; e += U) {
    int2 p[U];  // (j, i): j = this (transposed) row, i = the forward row that attends to j
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (NARROW) p[u] = pn[u];
      else p[u] = rc[(e + u < e1) ? e + u : e1 - 1];
    }
    float4 rt[U][VEC], gv[U][VEC], wv[U][VEC];
    float mk[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        if constexpr (CR > 0) {   // both tables below 4 GiB (checked at launch): 32-bit element offsets from a scalar base
          // (measured and dropped, gpurun_out r4h / r4i: without the record read -- 128 of the row's 640 bytes, its fifth line --
          // K4 runs 1.24 -> 1.07 ms; from a table of its own (128 MB, Infinity-Cache sized) the records cost the same 0.17 ms:
          // the price is the gathered line, wherever it lies)
          const uint32_t go = (uint32_t)p[u].y * (uint32_t)RW;
          gv[u][v] = ld4(a.GR + (go + (uint32_t)lc.cofs[v]));
          rt[u][v] = (PYGAT_DIAG_K4 & 8) ? gv[u][v] : ld4(a.GR + (go + (uint32_t)(R + 4 * lc.head[v])));   // (bit 3: diagnostic, no scalar-record read)
          wv[u][v] = ld4(a.Wh + ((uint32_t)((PYGAT_DIAG_K4 & 4) ? (p[u].x & 63) : p[u].x) * (uint32_t)ldr + (uint32_t)lc.cofs[v]));   // (bit 2: diagnostic, row-local reads from 64 cached rows)
        } else {
        const float* gr = a.GR + (int64_t)p[u].y * RW;           // gathered: one contiguous row
        gv[u][v] = ld4(gr + lc.cofs[v]);
        rt[u][v] = ld4(gr + R + 4 * lc.head[v]);
        wv[u][v] = ld4(a.Wh + (int64_t)p[u].x * ldr + lc.cofs[v]);  // row-local (L1 after the first edge)
        }
        mk[u][v] = 1.f;
        if (a.mask) mk[u][v] = a.mask[(int64_t)a.perm[(e + u < e1) ? e + u : e1 - 1] * ldh + lc.head[v]];
      }
    if constexpr (NARROW) {
#pragma unroll
      for (int u = 0; u < U; ++u) pn[u] = rc[(e + U + u < e1) ? e + U + u : e1 - 1];
    }
    float al[U][VEC], dz[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const float dp = group_sum_rt(lc.valid[v] ? dot4(gv[u][v], wv[u][v]) : 0.f, lph);
        const float tj = group_sum_rt(dot4(wv[u][v], adst[v]), lph);
        const float zz = rt[u][v].x + tj;
        const float ev = zz > 0.f ? zz : a.alpha * zz;
        const float a0 = __expf(ev - rt[u][v].y) * rt[u][v].z;
        dz[u][v] = a0 * (mk[u][v] * dp - rt[u][v].w) * (zz > 0.f ? 1.f : a.alpha);
        al[u][v] = a0 * mk[u][v];
      }
    if constexpr (WRITE_DZ) {
      const int S = lph < U ? lph : U;                  // lanes of a head that store (1, 2 or 4)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const int lih = (lc.cofs[v] >> 2) & (lph - 1);  // lane inside its head
#pragma unroll
        for (int u = 0; u < U; ++u)                      // lane (u mod S) of the head stores edge u
          if (lih == (u & (S - 1)) && lc.valid[v] && e + u < e1) a.dz_t[(e + u) * ldh + lc.head[v]] = dz[u][v];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (e + u < e1) {
        if (p[u].x != cur) {
          // (u > 0: the finished row's last edge is edge u - 1 of this round, its Wh row is still in wv[u - 1])
          col_flush<VEC, LPH, CR, DA>(a, lc, q, cur, cur == r_first && head_partial, false, acc, dt, da_lds, da_stride,
                                      (DA && PYGAT_K4_DA_WROW && u > 0) ? wv[u > 0 ? u - 1 : 0] : nullptr);
          cur = p[u].x;
#pragma unroll
          for (int v = 0; v < VEC; ++v) { acc[v] = make_float4(0.f, 0.f, 0.f, 0.f); dt[v] = 0.f; }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          acc[v].x = fmaf(al[u][v], gv[u][v].x, acc[v].x); acc[v].y = fmaf(al[u][v], gv[u][v].y, acc[v].y);
          acc[v].z = fmaf(al[u][v], gv[u][v].z, acc[v].z); acc[v].w = fmaf(al[u][v], gv[u][v].w, acc[v].w);
          dt[v] += dz[u][v];
        }
      }
    }
  }
  const bool tail_partial = tail_known ? tail_flag : a.g.rowptr[cur + 1] > e1;
  col_flush<VEC, LPH, CR, DA>(a, lc, q, cur, cur == r_first && head_partial, tail_partial, acc, dt, da_lds, da_stride);
}

template <int LPR, int VEC, bool WRITE_DZ, int LPH = 0, int CR = 0, bool DA = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CR > 0 ? (DA ? PYGAT_K4_DA_WAVES : PYGAT_K4_HEADLINE_WAVES) : 1))) void gat_bwd_col_kernel(ColArgs a) {
  constexpr int EPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int64_t k = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * EPW + lane / LPR;   // grid position (-> slot: col_walk)
  if constexpr (!DA) {
    if (k >= a.g.kn) return;
    col_walk<LPR, VEC, WRITE_DZ, LPH, CR, false>(a, k, nullptr, 0);
  } else {
    extern __shared__ __attribute__((aligned(16))) float4 da_sm[];   // [2][blockDim.x]: (src | dst) running sums, one pair per lane
    __shared__ int da_arrived;
    const int nt = (int)blockDim.x;
    float4* mine = da_sm + threadIdx.x;
    mine[0] = make_float4(0.f, 0.f, 0.f, 0.f); mine[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (threadIdx.x == 0) da_arrived = 0;
    __syncthreads();   // (at the START, where the waves stand together anyway: the counter below must be zero before its first add)
    if (k < a.g.kn) col_walk<LPR, VEC, WRITE_DZ, LPH, CR, true>(a, k, mine, nt);
    // No barrier at the end: a wave that is done leaves its SIMD slot; the LAST wave of the work-group to arrive adds the lane
    // groups, always in the same order (so the record does not depend on which wave that is).  Its workgroup-scope
    // acquire-release add comes after every other wave's LDS sums (theirs precede their own add in program order).
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(&da_arrived, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old == (nt >> 6) - 1) {
      const int R = CR ? CR : a.rs.R;
      for (int idx = lane; idx < 2 * LPR; idx += 64) {     // (which, c): float4 c of the (src | dst) half
        const int which = idx / LPR, c = idx % LPR;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int g = 0; g < nt / LPR; ++g) {
          const float4 x = da_sm[which * nt + g * LPR + c];
          t.x += x.x; t.y += x.y; t.z += x.z; t.w += x.w;
        }
        if (4 * c < R) st4(a.da_part + (int64_t)blockIdx.x * 2 * R + which * R + 4 * c, t);
      }
    }
  }
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
    float dt[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { acc[v] = make_float4(0.f, 0.f, 0.f, 0.f); dt[v] = 0.f; }
    if (wide || w == 0) {
      const int nw = wide ? 4 : 1;
      for (int q = (wide ? w : 0) * EPW * PF + slot; q < npieces; q += nw * EPW * PF) {
        float4 xp[PF][VEC];
        float tp[PF][VEC];
#pragma unroll
        for (int f = 0; f < PF; ++f) {
          const int qq = q + f * EPW;
          const int qc = qq < npieces ? qq : q;
          const float* p = a.part + (qc == 0 ? 2 * k + 1 : 2 * (k + qc)) * PS;
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            xp[f][v] = ld4(p + lc.cofs[v]); tp[f][v] = p[a.rs.R + lc.head[v]];
          }
        }
#pragma unroll
        for (int f = 0; f < PF; ++f)
          if (q + f * EPW < npieces) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
              acc[v].x += xp[f][v].x; acc[v].y += xp[f][v].y; acc[v].z += xp[f][v].z; acc[v].w += xp[f][v].w;
              dt[v] += tp[f][v];
            }
          }
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        acc[v] = slot_sum4<LPR>(acc[v]);
        dt[v] = slot_sum<LPR>(dt[v]);
      }
    }
    if (wide) {
      if (slot == 0) {
        float* p = fix_sm + w * PS;
#pragma unroll
        for (int v = 0; v < VEC; ++v)
          if (lc.valid[v]) {
            st4(p + lc.cofs[v], acc[v]);
            if (((lc.cofs[v] >> 2) & (a.rs.lph - 1)) == 0) p[a.rs.R + lc.head[v]] = dt[v];
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
          }
        }
      }
      __syncthreads();
    }
    if (w == 0 && slot == 0) col_finish<VEC>(a, lc, r, acc, dt);
  }
}

// list-driven variant: entry q of g.cut = (owner slot k, row, pieces); the first n_cut_wide entries (long
// chains) get a whole work-group each, the others one wave each.
template <int LPR, int VEC>
__global__ __launch_bounds__(64 * FIX_LIST_WAVES) void gat_bwd_col_fixup_list_kernel(ColArgs a) {
  constexpr int EPW = 64 / LPR;
  constexpr int PF = (VEC == 1) ? 4 : 2;
  constexpr int RPW = fix_rows_per_wave(LPR), GP = EPW / RPW;   // packed entries: RPW rows per wave, GP lane groups each
  extern __shared__ __attribute__((aligned(16))) float fix_sm[];  // [FIX_LIST_WAVES][R + 2H]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const bool wide = (int)blockIdx.x < a.g.n_cut_wide;   // (block-uniform)
  const int slot = lane / LPR;
  const int G = wide ? EPW : GP, g = slot % G;
  const int q0 = wide ? (int)blockIdx.x
                      : a.g.n_cut_wide + (((int)blockIdx.x - a.g.n_cut_wide) * FIX_LIST_WAVES + w) * RPW + slot / GP;
  const bool have = q0 < a.g.n_cut;
  const int qe = have ? q0 : a.g.n_cut - 1;
  const int64_t k = a.g.cut[3 * qe];
  const int r = a.g.cut[3 * qe + 1];
  const int npieces = have ? a.g.cut[3 * qe + 2] : 0;
  const LaneCols<VEC> lc = lane_cols<LPR, VEC>(a.rs);
  const int64_t PS = a.rs.R + 2 * a.rs.H;
  const int wsel = wide ? w : 0;   // wave index inside the merge; a lone wave plays wave 0
  float4 acc[VEC];
  float dt[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) { acc[v] = make_float4(0.f, 0.f, 0.f, 0.f); dt[v] = 0.f; }
  {
    const int nw = wide ? FIX_LIST_WAVES : 1;
    for (int q = (wide ? w : 0) * G * PF + g; q < npieces; q += nw * G * PF) {
      float4 xp[PF][VEC];
      float tp[PF][VEC];
#pragma unroll
      for (int f = 0; f < PF; ++f) {
        const int qq = q + f * G;
        const int qc = qq < npieces ? qq : q;
        const float* p = a.part + (qc == 0 ? 2 * k + 1 : 2 * (k + qc)) * PS;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          xp[f][v] = ld4(p + lc.cofs[v]); tp[f][v] = p[a.rs.R + lc.head[v]];
        }
      }
#pragma unroll
      for (int f = 0; f < PF; ++f)
        if (q + f * G < npieces) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            acc[v].x += xp[f][v].x; acc[v].y += xp[f][v].y; acc[v].z += xp[f][v].z; acc[v].w += xp[f][v].w;
            dt[v] += tp[f][v];
          }
        }
    }
    if (wide) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        acc[v] = slot_sum4<LPR>(acc[v]);
        dt[v] = slot_sum<LPR>(dt[v]);
      }
    } else {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        acc[v] = slot_sum4<LPR, LPR * GP>(acc[v]);
        dt[v] = slot_sum<LPR, LPR * GP>(dt[v]);
      }
    }
  }
  if (wide) {
    if (slot == 0) {
      float* p = fix_sm + w * PS;
#pragma unroll
      for (int v = 0; v < VEC; ++v)
        if (lc.valid[v]) {
          st4(p + lc.cofs[v], acc[v]);
          if (((lc.cofs[v] >> 2) & (a.rs.lph - 1)) == 0) p[a.rs.R + lc.head[v]] = dt[v];
        }
    }
    __syncthreads();
    if (w == 0 && slot == 0) {
#pragma unroll
      for (int ww = 1; ww < FIX_LIST_WAVES; ++ww) {
        const float* p = fix_sm + ww * PS;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const float4 x = ld4(p + lc.cofs[v]);
          acc[v].x += x.x; acc[v].y += x.y; acc[v].z += x.z; acc[v].w += x.w;
          dt[v] += p[a.rs.R + lc.head[v]];
        }
      }
    }
    __syncthreads();
  }
  if (wsel == 0 && g == 0 && npieces > 0) col_finish<VEC>(a, lc, r, acc, dt);
}


}  // namespace pygat

using namespace pygat;


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
int footprint_k4_headline_da(int* regs, int* scratch) {
  return footprint_of(reinterpret_cast<const void*>(&gat_bwd_col_kernel<32, 1, false, 4, 128, true>), regs, scratch);
}
}  // namespace pygat

extern "C" size_t pygat_gat_backward_col_da_bytes(const pygat_graph* gT, int H, int Fo, int head_group) {
  GraphDev g;
  if (!gT || check_graph(gT, &g, 2) != PYGAT_OK) return 0;
  const int Fp = padded_width(Fo);
  if (H <= 0 || Fp <= 0) return 0;
  return (size_t)col_da_blocks(g, H, Fp, head_group_arg(head_group, g.n, H, Fp), nullptr) * 2 * (size_t)H * Fp * sizeof(float);
}

extern "C" int pygat_gat_backward_col(const pygat_graph* gT, const int32_t* perm_t, int H, int Fo, float alpha,
                                      const float* Wh, const float* a_pad, const float* GR,
                                      const float* att_mask, const float* ds, float* dWh, float* dt, float* dz_t,
                                      void* part, float* da_part, int h_first, int h_count, int head_group, void* stream) {

  ColArgs a;
  int rc = check_graph(gT, &a.g, 2);
  if (rc) return rc;
  const int Fp = padded_width(Fo);
  HeadRange rg;
  PYGAT_REQUIRE(H > 0 && Fp > 0, "gat_backward_col: unsupported H=%d F'=%d", H, Fo);
  PYGAT_REQUIRE(make_head_range(H, h_first, h_count, &rg), "gat_backward_col: bad head range [%d, +%d) of %d", h_first, h_count, H);
  PYGAT_REQUIRE(Wh && a_pad && GR && dWh && dt && part, "gat_backward_col: null pointer");
  PYGAT_REQUIRE((ds != nullptr) != (dz_t != nullptr),
                "gat_backward_col: pass either ds (row sums known) or dz_t (row sums taken afterwards), not both");
  PYGAT_REQUIRE(!att_mask || perm_t, "gat_backward_col: an attention mask needs perm_t (mask is in forward edge order)");
  PYGAT_REQUIRE(aligned16(Wh) && aligned16(GR) && aligned16(dWh) && aligned16(a_pad) && aligned16(part),
                "gat_backward_col: row tables must be 16-byte aligned");
  const int hg = head_group_arg(head_group, a.g.n, rg.hr, Fp);
  PYGAT_REQUIRE(hg > 0, "gat_backward_col: head_group=%d gives rows of more than 1024 floats per pass", head_group);
  if (da_part) {
    PYGAT_REQUIRE(ds && rg.hr == H && aligned16(da_part) && col_da_blocks(a.g, H, Fp, hg, nullptr) > 0,
                  "gat_backward_col: da_part needs ds, 8 heads x 16 in one pass and a cut-row list "
                  "(pygat_gat_backward_col_da_bytes() == 0 otherwise)");
  }
  a.da_part = da_part;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nslots = a.g.kn;   // all slots, or the prefix before a self-loop-only tail (pygat_gat_backward_col_tail)
  PYGAT_REQUIRE(a.g.kn == num_slots(a.g) || (a.g.cut && !a.g.order), "gat_backward_col: a slot prefix needs the cut-row list and no slot_order");
  for (int h0 = 0; h0 < rg.hr; h0 += hg) {
    const int hc = (rg.hr - h0 < hg) ? rg.hr - h0 : hg;
    const int gh = rg.hb + h0;
    PYGAT_REQUIRE(make_window_shape(H, Fo, hc, &a.rs), "gat_backward_col: unsupported H=%d F'=%d", hc, Fo);
    a.alpha = alpha; a.perm = perm_t; a.mask = att_mask ? att_mask + gh : nullptr;
    a.Wh = Wh + (int64_t)gh * Fp; a.GR = GR + gr_window_offset(h0, Fp); a.ldgr = (int64_t)rg.hr * Fp + 4 * rg.hr;
    a.a_pad = a_pad + (int64_t)gh * 2 * Fp; a.ds = ds ? ds + gh : nullptr; a.dz_t = dz_t ? dz_t + gh : nullptr;
    a.dWh = dWh + (int64_t)gh * Fp; a.dt = dt + gh; a.part = (float*)part;
    int lpr, vec;
    pick_lanes(a.rs, &lpr, &vec);
    // narrow rows (a wave carries 8-64 slots, the whole grid is a few ten waves per SIMD): one-wave work-groups, so that a
    // SIMD slot is refilled as soon as ITS wave ends instead of when the slowest of four does
    const unsigned bt = (vec == 1 && lpr <= 8) ? (unsigned)narrow_block() : 256u;
    const unsigned blocks = (unsigned)cdiv(cdiv(nslots, 64 / lpr), bt / 64);
    const size_t da_lds = da_part ? 2 * (size_t)bt * sizeof(float4) : 0;
    if (dz_t) {
      PYGAT_DISPATCH_LANES(lpr, vec,
                           hipLaunchKernelGGL((gat_bwd_col_kernel<LPR, VEC, true>), dim3(blocks), dim3(bt), 0, st, a));
    } else if (lpr == 32 && vec == 1 && a.rs.lph == 4) {   // 8 heads x 16: the headline shape
      const bool dense = a.rs.R == 128 && a.rs.ldr == 128 && a.rs.ldh == 8 && a.ldgr == 160 && a.rs.H == 8 && !att_mask &&
                         (int64_t)a.g.n * 160 * 4 < ((int64_t)1 << 32);
      if (dense && da_part)
        hipLaunchKernelGGL((gat_bwd_col_kernel<32, 1, false, 4, 128, true>), dim3(blocks), dim3(bt), da_lds, st, a);
      else if (dense)
        hipLaunchKernelGGL((gat_bwd_col_kernel<32, 1, false, 4, 128>), dim3(blocks), dim3(bt), 0, st, a);
      else if (da_part)   // (8 x 16 with an attention mask or a table of 4 GiB and more: run-time strides, 130 VGPRs)
        hipLaunchKernelGGL((gat_bwd_col_kernel<32, 1, false, 4, 0, true>), dim3(blocks), dim3(bt), da_lds, st, a);
      else
        hipLaunchKernelGGL((gat_bwd_col_kernel<32, 1, false, 4>), dim3(blocks), dim3(bt), 0, st, a);
    } else {
      PYGAT_DISPATCH_LANES(lpr, vec,
                           hipLaunchKernelGGL((gat_bwd_col_kernel<LPR, VEC, false>), dim3(blocks), dim3(bt), 0, st, a));
    }
    PYGAT_CHECK_LAUNCH("gat_backward_col");
    const size_t fix_lds = (a.g.cut ? FIX_LIST_WAVES : 4) * (size_t)(a.rs.R + 2 * a.rs.H) * sizeof(float);
    if (a.g.cut) {
      if (a.g.n_cut > 0) {
        const unsigned fb = (unsigned)(a.g.n_cut_wide + cdiv(a.g.n_cut - a.g.n_cut_wide, FIX_LIST_WAVES * fix_rows_per_wave(lpr)));
        PYGAT_DISPATCH_LANES(lpr, vec, hipLaunchKernelGGL((gat_bwd_col_fixup_list_kernel<LPR, VEC>), dim3(fb),
                                                          dim3(64 * FIX_LIST_WAVES), fix_lds, st, a));
      }
    } else {
      PYGAT_DISPATCH_LANES(lpr, vec, hipLaunchKernelGGL((gat_bwd_col_fixup_kernel<LPR, VEC>),
                                                        dim3((unsigned)cdiv(nslots, FIX_SCREEN)), dim3(256), fix_lds, st, a));
    }
    PYGAT_CHECK_LAUNCH("gat_backward_col_fixup");
  }
  return PYGAT_OK;
}
