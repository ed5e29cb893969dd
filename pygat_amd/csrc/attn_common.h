// Lane mapping shared by the fused attention kernels K2 (forward), K3 (row
// backward) and K4 (column backward).
//
// A node row is R = H*Fp floats (all local heads interleaved, each padded to
// Fp = power of two), i.e. NCH = R/4 float4 "chunks".  The CSR edge list is cut
// into SLOTS of `ts` consecutive edges; a lane group of LPR lanes owns one slot
// and walks its edges (nnz split: perfect balance whatever the degree skew):
//   LPR  lanes per gathered row  = pow2 >= NCH, capped at 64
//   EPW  = 64 / LPR slots per wave (independent lane groups, no shuffles between them)
//   VEC  chunks per lane (only when NCH > 64, LPR = 64): chunk c = c0 + 64*v
// Lane l: slot-in-wave = l / LPR, c0 = l % LPR.  A float4 never straddles heads
// (Fp % 4 == 0); the head of chunk c is (4c) / Fp and the LPH = Fp/4 lanes of a
// head are consecutive and aligned, so per-head reductions are DPP butterflies.
#pragma once
#include "common.h"

namespace pygat {

constexpr int FIX_LIST_WAVES = 16;  // waves per work-group of the list-driven fix-up kernels
// Cut rows one wave of a list-driven fix-up merges side by side (the entries behind the wide ones): a wave of 64 / LPR lane
// groups per row piece had most of them idle -- the config-5 graph has 21.7 k cut rows of 5 pieces on average at 64-edge slots,
// 57.9 k of 4.8 at 32 -- and the launch is latency per wave, not bytes.  The list is sorted by piece count, so the rows of a
// wave are alike.
__host__ __device__ constexpr int fix_rows_per_wave(int lpr) { return 64 / lpr >= 8 ? 4 : (64 / lpr >= 2 ? 2 : 1); }
constexpr int FIX_WIDE = 32;   // cut rows with more pieces are merged by a whole work-group
constexpr int FIX_SCREEN = 8;  // slots screened per wave by the fix-up kernels (owned rows are merged serially)
constexpr float NEG_BIG = -1.0e30f;  // running-max seed: exp(NEG_BIG - x) == 0, exp(NEG_BIG - NEG_BIG) == 1

struct RowShape {
  int H, Fo, Fp, R, NCH;  // heads of THIS kernel pass, true width, padded width, H*Fp, R/4
  int fp_shift;           // log2(Fp)
  int lph;                // lanes per head = Fp/4 (power of two, may exceed 64 only if Fp > 256: rejected)
  // A pass may cover a WINDOW of the level's heads (see head_group): the per-node tables keep the
  // level's full width, so rows are addressed with the level's strides and pointers pre-offset to the window.
  int Htot;               // heads of the level (mean mode divides by it)
  int64_t ldr;            // row stride of the R-wide tables (Wh, Sk, hattn, dWh) = Htot*Fp
  int64_t ldh;            // row stride of the per-head tables (s, m, Z, ds, dt, attention mask) = Htot
  int64_t ldo;            // row stride of out / G = Htot*Fo
};

static inline bool make_row_shape(int H, int Fo, RowShape* rs) {
  int Fp = padded_width(Fo);
  if (H <= 0 || Fp == 0) return false;
  rs->H = H; rs->Fo = Fo; rs->Fp = Fp; rs->R = H * Fp; rs->NCH = rs->R / 4;
  rs->fp_shift = ilog2(Fp); rs->lph = Fp / 4;
  rs->Htot = H; rs->ldr = rs->R; rs->ldh = H; rs->ldo = (int64_t)H * Fo;
  return rs->NCH <= 256;  // VEC <= 4
}

// Heads per kernel pass ("head window").
// A pass takes rows of at most 1024 floats (VEC <= 4 chunks of 16 B per lane); wider levels are walked
// window by window.  That is the only reason to window the forward: measured on MI355X (RMAT 1M/10.8M,
// 8 heads x 128) K2 is fastest on the full 4-KB rows (9.1 ms; 9.7 ms as 4 windows of 1 KB, whose
// gathers hit only a quarter of each row and load the HBM channels unevenly).
static inline int head_group_fwd(int H, int Fp) {
  if (H * Fp <= 1024) return H;
  const int g = 1024 / Fp;
  return g < 1 ? 1 : g;
}
// K2 on a cache-resident table (PPI: 3144 x 1024 floats): at VEC = 4 it holds 205 VGPRs = 2 waves per SIMD; two windows of
// 512 floats run 4 and finish sooner despite the second launch (PPI epoch 2.43 -> 2.40 ms).  Large tables keep whole rows
// (the gathered row is the HBM transaction; a window re-reads the edge structure: measured).
static inline int head_group_fwd_n(int64_t n, int H, int Fp) {
  constexpr int fwd_w = 512;   // floats per window on a cache-resident table
  const int64_t R = (int64_t)H * Fp;
  if (R > fwd_w && R <= 1024 && n * R * 4 < ((int64_t)256 << 20)) { const int g = fwd_w / Fp; return g < 1 ? 1 : g; }
  return head_group_fwd(H, Fp);
}
// The two backward passes hold two gathered/row-local rows per edge: at VEC >= 3 they need 170-250
// VGPRs, run 2 waves per SIMD and stop covering the HBM latency (same workload: K3b 11.0 -> 8.5 ms,
// K4 11.6 -> 9.9 ms as windows of 256 floats = one chunk per lane).  Windows cost extra launches, so
// they are used where the gathered table is far beyond the caches and rows are wider than 256 floats (8 heads x 64 on the
// config-5 graph: K4 5.14 -> 4.72 ms as two windows of 256, the step 13.6 -> 13.0); a cache-resident table wider than 512
// floats takes windows of 512 (VEC = 2, 150 VGPRs) instead of whole rows -- K4 at VEC = 4 holds 256 VGPRs + 16 AGPRs, ONE
// wave per SIMD (PPI epoch 2.55 -> 2.43 ms; windows of 256 there: 2.66).
// This is the DEFAULT (pygat_head_group): the backward entry points take the heads per window as an argument (0 = this
// default), because it is also the layout of a GR row, which the caller allocates and reads -- no environment variable
// decides it (rounds 1-3 read PYGAT_BWD_WINDOW_BYTES and three more on every call).
static inline int head_group_bwd(int64_t n, int H, int Fp) {
  const int64_t R = (int64_t)H * Fp;
  constexpr int64_t min_bytes = (int64_t)256 << 20;   // gathered table beyond the Infinity Cache
  if (R <= 256) return head_group_fwd(H, Fp);
  if (n * R * 4 < min_bytes) {   // cache-resident table: windows of 512 floats
    if (512 >= R) return head_group_fwd(H, Fp);
    const int g = 512 / Fp;
    return g < 1 ? 1 : g;
  }
  const int g = 256 / Fp;
  return g < 1 ? 1 : g;
}
// heads per window of a backward call: the caller's choice when it gave one (any window a kernel pass can take), else the default
static inline int head_group_arg(int head_group, int64_t n, int H, int Fp) {
  if (head_group <= 0) return head_group_bwd(n, H, Fp);
  const int cap = head_group_fwd(H, Fp);
  if (head_group > cap) return -1;                // rows of more than 1024 floats per pass
  return head_group < H ? head_group : H;
}
// narrow rows (a wave carries 8-64 slots, the whole grid is a few ten waves per SIMD): one-wave work-groups, so that a SIMD
// slot is refilled as soon as ITS wave ends instead of when the slowest of four does (measured against 128 / 256: DESIGN.md)
static inline unsigned narrow_block() { return 64u; }

// shape of the window [h0, h0+hc) of a level with Htot heads
static inline bool make_window_shape(int Htot, int Fo, int hc, RowShape* rs) {
  if (!make_row_shape(hc, Fo, rs)) return false;
  rs->Htot = Htot; rs->ldr = (int64_t)Htot * rs->Fp; rs->ldh = Htot; rs->ldo = (int64_t)Htot * Fo;
  return true;
}
// A backward call may cover only the heads [hb, hb+hr) of a level with H heads (h_count == 0: all of them):
// the level's tables (Wh, G, out, s, m, Z, ds, dt, dWh, dz_t, masks) keep their full width and are addressed with
// the level's strides, while GR -- written and read only by the backward -- is compact for the range.
struct HeadRange { int hb, hr; };
static inline bool make_head_range(int H, int h_first, int h_count, HeadRange* r) {
  if (h_count == 0 && h_first == 0) { r->hb = 0; r->hr = H; return H > 0; }
  if (h_first < 0 || h_count <= 0 || h_first + h_count > H) return false;
  r->hb = h_first; r->hr = h_count;
  return true;
}
// column offset of a window inside a GR row [Gp(window) | rowtab(window)] ... : h0 * (Fp + 4)
static inline int64_t gr_window_offset(int h0, int Fp) { return (int64_t)h0 * (Fp + 4); }

// (LPR, VEC) for a row shape
static inline void pick_lanes(const RowShape& rs, int* lpr, int* vec) {
  if (rs.NCH <= 64) {
    int l = 1;
    while (l < rs.NCH) l <<= 1;
    *lpr = l; *vec = 1;
  } else {
    *lpr = 64; *vec = (rs.NCH + 63) / 64;
  }
}

#define PYGAT_DISPATCH_LANES(LPRV, VECV, CALL)                      \
  do {                                                              \
    if ((VECV) == 1) {                                              \
      switch (LPRV) {                                               \
        case 1: { constexpr int LPR = 1, VEC = 1; CALL; } break;    \
        case 2: { constexpr int LPR = 2, VEC = 1; CALL; } break;    \
        case 4: { constexpr int LPR = 4, VEC = 1; CALL; } break;    \
        case 8: { constexpr int LPR = 8, VEC = 1; CALL; } break;    \
        case 16: { constexpr int LPR = 16, VEC = 1; CALL; } break;  \
        case 32: { constexpr int LPR = 32, VEC = 1; CALL; } break;  \
        default: { constexpr int LPR = 64, VEC = 1; CALL; } break;  \
      }                                                             \
    } else if ((VECV) == 2) { constexpr int LPR = 64, VEC = 2; CALL; \
    } else if ((VECV) == 3) { constexpr int LPR = 64, VEC = 3; CALL; \
    } else { constexpr int LPR = 64, VEC = 4; CALL; }               \
  } while (0)

struct GraphDev {  // device view of pygat_graph
  int n;
  int64_t nnz;
  const int32_t* rowptr;
  const int2* rc;   // (row, col) per edge
  int ts;           // nominal edges per slot
  const int32_t* sb;  // row-snapped slot borders [nslots+1] or nullptr (uniform slots)
  const int4* meta;   // [nslots] (e0, e1, first row, flags) or nullptr (pygat_graph.slot_meta)
  const int32_t* cut;  // [n_cut][3] (slot, row, pieces) or nullptr
  int n_cut, n_cut_wide;
  int64_t k0, kn;   // active slot range [k0, k0 + kn) (the forward can work on a range of whole rows)
  const int32_t* order;   // slot handed to grid position q (pygat_graph.slot_order) or nullptr: q itself
  const int32_t* urow;    // caller's row of internal node i (pygat_graph.user_row) or nullptr: i itself
};

// allow_slot_range: 0 none, 1 any range of whole rows (pygat_gat_forward), 2 a PREFIX [0, slot_count) (the column pass and the fold of
// its da records: the slots behind it are the self-loop-only tail of a degree-ordered pattern, pygat_gat_backward_col_tail)
static inline int check_graph(const pygat_graph* g, GraphDev* d, int allow_slot_range = 0) {
  if (!g || !g->rowptr || !g->edge_rc || g->n <= 0 || g->nnz <= 0) {
    set_error("graph: null or empty (n=%d nnz=%lld)", g ? g->n : -1, g ? (long long)g->nnz : -1LL);
    return PYGAT_EINVAL;
  }
  if (g->nnz >= (int64_t)1 << 31) {
    set_error("graph: nnz %lld exceeds int32 edge indexing", (long long)g->nnz);
    return PYGAT_EINVAL;
  }
  if (g->slot_edges < 4 || (g->slot_edges & 3)) {
    set_error("graph: slot_edges=%d must be a multiple of 4, >= 4", g->slot_edges);
    return PYGAT_EINVAL;
  }
  d->n = g->n; d->nnz = g->nnz; d->rowptr = g->rowptr; d->rc = reinterpret_cast<const int2*>(g->edge_rc);
  d->ts = g->slot_edges; d->sb = g->slot_begin; d->meta = reinterpret_cast<const int4*>(g->slot_meta);
  if (g->slot_meta && !aligned16(g->slot_meta)) {
    set_error("graph: slot_meta must be 16-byte aligned");
    return PYGAT_EINVAL;
  }
  d->cut = g->cut_rows; d->n_cut = g->cut_rows ? g->n_cut : 0; d->n_cut_wide = g->cut_rows ? g->n_cut_wide : 0;
  if (d->n_cut < 0 || d->n_cut_wide < 0 || d->n_cut_wide > d->n_cut) {
    set_error("graph: bad cut-row list (n_cut=%d n_cut_wide=%d)", g->n_cut, g->n_cut_wide);
    return PYGAT_EINVAL;
  }
  const int64_t total = (g->nnz + g->slot_edges - 1) / g->slot_edges;
  d->k0 = 0; d->kn = total;
  d->order = g->slot_order;
  d->urow = g->user_row;
  if (g->slot_order && g->slot_count != 0) {
    set_error("graph: slot_order needs the whole slot range (slot_count = 0)");
    return PYGAT_EINVAL;
  }
  if (g->slot_count != 0) {
    if (!allow_slot_range || g->slot_first < 0 || g->slot_count < 0 || g->slot_first + g->slot_count > total ||
        (allow_slot_range == 2 && g->slot_first != 0)) {
      set_error("graph: slot range [%lld, +%lld) of %lld %s", (long long)g->slot_first, (long long)g->slot_count,
                (long long)total, allow_slot_range ? "is out of bounds (or not a prefix, for the column pass)"
                                                   : "is not supported by this entry point");
      return PYGAT_EINVAL;
    }
    d->k0 = g->slot_first; d->kn = g->slot_count;
  }
  return PYGAT_OK;
}

#ifdef __HIPCC__
__host__ __device__
#endif
static inline int64_t num_slots(const GraphDev& g) { return (g.nnz + g.ts - 1) / g.ts; }
#ifdef __HIPCC__
// ELU: expm1 by a short series near 0 (where exp(x)-1 cancels), fast exp elsewhere (K2's epilogue and the self-loop-only tail)
__device__ __forceinline__ float elu1(float x) {
  if (x > 0.f) return x;
  if (x > -0.03125f) return x * (1.f + x * (0.5f + x * (0.16666667f + x * 0.041666668f)));
  return __expf(x) - 1.f;
}
// slot handed to grid position q of a main launch (q < num_slots): pygat_graph.slot_order, or q itself
__device__ __forceinline__ int64_t slot_at(const GraphDev& g, int64_t q) { return g.order ? (int64_t)g.order[q] : q; }
#endif

// Work-groups of the column pass (pygat_gat_backward_col) over gT for a level of H heads = its da_part records, or 0 when the
// pass does not take the attention-vector gradient along: no cut-row list (the rows its fix-up finishes are folded in by
// pygat_a_grad_fold from that list), head windows (hg < H), or a row shape other than 8 heads x 16 -- the one
// instantiation that keeps its four waves per SIMD with the sums (128 VGPRs).  The others hold 130-138 VGPRs with them,
// three waves, and lose more than the a-gradient pass costs: measured at 1 / 2 / 4 heads of 16 on the config-5 graph, K4
// 0.275 -> 0.291, 0.413 -> 0.444, 0.71 -> 0.83 ms (profiles/r4m_as_rank_of.jsonl of the first build), so they are not built.
static inline int64_t col_da_blocks(const GraphDev& g, int H, int Fp, int hg, int* lpr_out) {
  RowShape rs;
  if (!g.cut || !make_window_shape(H, Fp, H, &rs)) return 0;   // (Fp as F': padded_width(Fp) == Fp)
  if (hg < H) return 0;
  int lpr, vec;
  pick_lanes(rs, &lpr, &vec);
  if (vec != 1 || H != 8 || Fp != 16) return 0;
  const unsigned bt = (lpr <= 8) ? narrow_block() : 256u;
  if (lpr_out) *lpr_out = lpr;
  return cdiv(cdiv(g.kn, 64 / lpr), bt / 64);   // (kn: the active slots -- all of them, or the prefix of the call)
}

#ifdef __HIPCC__
// edge range of slot k
__device__ __forceinline__ void slot_range(const GraphDev& g, int64_t k, int64_t* e0, int64_t* e1) {
  if (g.sb) {
    *e0 = g.sb[k];
    *e1 = g.sb[k + 1];
  } else {
    *e0 = k * g.ts;
    *e1 = (*e0 + g.ts < g.nnz) ? *e0 + g.ts : g.nnz;
  }
}
// slot that contains edge position pos
__device__ __forceinline__ int64_t slot_of(const GraphDev& g, int64_t pos) {
  int64_t k = pos / g.ts;
  if (g.sb && pos < g.sb[k]) --k;
  return k;
}

template <int VEC>
struct LaneCols {
  int cofs[VEC];   // float offset of the lane's chunk inside a padded row (clamped when invalid)
  int head[VEC];
  bool valid[VEC];
};

template <int LPR, int VEC>
__device__ __forceinline__ LaneCols<VEC> lane_cols(const RowShape& rs) {
  LaneCols<VEC> lc;
  const int c0 = (threadIdx.x & 63) & (LPR - 1);
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    int c = c0 + 64 * v;
    lc.valid[v] = c < rs.NCH;
    int cc = lc.valid[v] ? c : 0;
    lc.cofs[v] = 4 * cc;
    lc.head[v] = (4 * cc) >> rs.fp_shift;
  }
  return lc;
}

// sum / online-softmax merge across the EPW edge slots of a wave (lanes l and l^off, off >= LPR)
// "does any lane of my LPR-lane group say yes": a ballot, usable in divergent code as long as whole groups diverge
template <int LPR>
__device__ __forceinline__ bool row_any(bool p) {
  const unsigned long long b = __ballot(p);
  if constexpr (LPR >= 64) {
    return b != 0ull;
  } else if constexpr (LPR == 32) {   // one select between the two halves of the (scalar) ballot
    return (((threadIdx.x & 32) ? (uint32_t)(b >> 32) : (uint32_t)b)) != 0u;
  } else {
    const int sh = (threadIdx.x & 63) & ~(LPR - 1);
    return ((b >> sh) & ((1ull << LPR) - 1ull)) != 0ull;
  }
}

template <int LPR, int SPAN = 64>   // sum over the lane groups of every aligned SPAN-lane block
__device__ __forceinline__ float slot_sum(float x) {
#pragma unroll
  for (int off = LPR; off < SPAN; off <<= 1) x += __shfl_xor(x, off);
  return x;
}
template <int LPR, int SPAN = 64>
__device__ __forceinline__ float4 slot_sum4(float4 v) {
#pragma unroll
  for (int off = LPR; off < SPAN; off <<= 1) {
    v.x += __shfl_xor(v.x, off); v.y += __shfl_xor(v.y, off);
    v.z += __shfl_xor(v.z, off); v.w += __shfl_xor(v.w, off);
  }
  return v;
}
#endif

}  // namespace pygat
