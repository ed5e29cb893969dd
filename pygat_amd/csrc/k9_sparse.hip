// K9 -- the level-1 projection and its weight gradient on SPARSE input features (gfx950).
//
// The reference hands the first level a dense [N, Fin] feature matrix (utils.py:38-41,60: `features.todense()`) and
// multiplies it with every head's W (layers.py:35,134; under per-head dropout, layers.py:34,132).  On its datasets that
// matrix is a row-normalised bag of words -- Cora 1.27 % non-zero (18 of 1433 per row), Citeseer 0.85 %, Pubmed 10 % --
// so the dense products (Cora: 2 x 0.5 GFLOP per training step through 32-column MFMA tiles of which a head of 8 uses a
// quarter) spend 40 % of an epoch multiplying zeros.  The pattern of x never changes between epochs: it is extracted
// once (pygat_amd/features.py, cached like the adjacency), and
//   forward   [Wh | Sk | s][i, :] = scale * sum_{k in nz(i)} x_ik m_h(i,k) Wcat[k, :]        one WAVE per row
//   backward  dW_h[k, :] = scale * sum_{i in nz^T(k)} x_ik m_h(i,k) dWh_h[i, :]  (dWskip: Gp)   one wave per feature k
// with m_h the SAME per-head input-dropout decisions as the dense path (rng.h: a function of (i, k, h) and the seed, or the
// explicit bit bytes of the tests) -- the two paths differ in summation order only.  The lanes of a wave fetch up to 64
// (index, value) pairs of the row / column at once and hand them round with readlane-style shuffles, four gathered rows
// of Wcat / dWh in flight; fixed order, no atomics.
#include "rng.h"

namespace pygat {

constexpr int SP_CPL = 8;   // columns per lane: up to 512 output columns

struct SpArgs {
  int n, Fin, H, Fo, Fp, R, ncols, has_skip, with_s;
  const int32_t* ptr;       // rowptr [n+1] (forward) / colptr [Fin+1] (backward)
  const int32_t* idx;       // column index per non-zero (forward) / row index (backward)
  const float* val;
  int masked;               // 0: no dropout
  int hq_shift;             // log2 of (quads of heads, rounded up to a power of two) (masked)
  DropRng g;                // masked, bits == nullptr
  const unsigned char* bits;  // [n x Fin] explicit decisions or nullptr
};

// head of output column c: [0, R) W heads, [R, 2R) skip heads (has_skip), behind them the H score columns (never masked)
__device__ __forceinline__ int sp_head(const SpArgs& a, int c, int fp_shift) { return (c < a.R ? c : c - a.R) >> fp_shift; }

// keep decisions of heads 4*hq .. 4*hq+3 for x[i, k], as a nibble
__device__ __forceinline__ uint32_t sp_keep4(const SpArgs& a, uint64_t seed, int i, int k, int hq) {
  if (a.bits) return ((uint32_t)a.bits[(int64_t)i * a.Fin + k] >> (4 * hq)) & 15u;
  return keep_nibble(a.g, draw_heads4(a.g, seed, (uint32_t)i, (uint32_t)k, (uint32_t)hq));
}

// The walk both kernels share.  `fixed` is the row (forward) / feature column (BWD) this wave owns, the non-zeros
// [e0, e1) carry the other coordinate.  The 64 lanes fetch 64 (index, value) pairs at once; each pair is then handed to all
// lanes through a scalar register (v_readlane with a uniform lane number: no LDS crossbar), eight gathered table rows in
// flight.  Dropout decisions depend on (row, column, HEAD) only, not on the output column, and one Philox call yields four
// heads (rng.h draw_heads4): a batch of 64 / Q pairs (Q = quads of heads, rounded up to a power of two) gets its decisions
// from ONE call per lane -- lane = (pair, quad) -- collected with four ballots; every output column then picks the ballot
// of its head's word, shifted by its quad, once per batch, and finds pair j's decision at the uniform bit j * Q.  (One call
// per lane per pair, as the first version had it, made Pubmed's 50-word rows 25 x more Philox than needed: 158 us.)
template <int CPL, bool BWD, bool MASKED>
__device__ __forceinline__ void sp_walk(const SpArgs& a, uint64_t seed, int fixed, int e0, int e1, int lane, const int (&cc)[CPL],
                                        const int (&hh)[CPL], float (&acc)[CPL], const float* __restrict__ T1, int64_t ld1,
                                        const float* __restrict__ T2, int64_t ld2) {
  constexpr int U = 8;
  const int B = MASKED ? (64 >> a.hq_shift) : 64;                    // 64, 32, 16 or 8: a multiple of U
  // byte offsets in 32 bits (sp_setup checks the tables stay below 4 GiB): one scalar multiply per pair, the gathered row's
  // address is a uniform base plus a 32-bit lane offset
  const char* __restrict__ B1 = (const char*)T1;
  const char* __restrict__ B2 = (const char*)T2;
  const uint32_t s1 = (uint32_t)ld1 * 4u, s2 = (uint32_t)ld2 * 4u;
  uint32_t c4[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) c4[q] = 4u * (uint32_t)((BWD && cc[q] >= a.R) ? cc[q] - a.R : cc[q]);
  for (int eb = e0; eb < e1; eb += 64) {
    const int cnt = (e1 - eb < 64) ? e1 - eb : 64;
    const int li = eb + (lane < cnt ? lane : cnt - 1);
    const int kk = a.idx[li];
    const int vv = lane < cnt ? __float_as_int(a.val[li]) : 0;       // lanes past the end: a valid index with value 0
    for (int jb = 0; jb < cnt; jb += B) {
      uint32_t lo[CPL], hi[CPL];
      if (MASKED) {
        const int jl = jb + (lane >> a.hq_shift), ql = lane & ((1 << a.hq_shift) - 1);
        const int other = __shfl(kk, jl < cnt ? jl : cnt - 1);
        uint32_t nib = 0;
        if (jl < cnt && 4 * ql < a.H) nib = BWD ? sp_keep4(a, seed, other, fixed, ql) : sp_keep4(a, seed, fixed, other, ql);
        const uint64_t m0 = __ballot(nib & 1u), m1 = __ballot(nib & 2u), m2 = __ballot(nib & 4u), m3 = __ballot(nib & 8u);
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          const int wd = hh[q] & 3;
          const uint64_t sel = (wd == 0 ? m0 : wd == 1 ? m1 : wd == 2 ? m2 : m3) >> (hh[q] >> 2);
          lo[q] = (uint32_t)sel; hi[q] = (uint32_t)(sel >> 32);
        }
      }
      const int nb = (cnt - jb < B) ? cnt - jb : B;
      // pairs j + u >= nb of a round exist only in the last batch (B is a multiple of U): lanes >= cnt, value 0
      for (int j = 0; j < nb; j += U) {
        float v[U], w[U][CPL];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t id = (uint32_t)__builtin_amdgcn_readlane(kk, jb + j + u);
          v[u] = __int_as_float(__builtin_amdgcn_readlane(vv, jb + j + u));
          const uint32_t o1 = id * s1, o2 = id * s2;
#pragma unroll
          for (int q = 0; q < CPL; ++q)
            w[u][q] = (BWD && cc[q] >= a.R) ? *(const float*)(B2 + (o2 + c4[q])) : *(const float*)(B1 + (o1 + c4[q]));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int pos = ((j + u) << a.hq_shift) & 63;
#pragma unroll
          for (int q = 0; q < CPL; ++q) {
            float wq = w[u][q];
            if (MASKED) {
              const uint32_t mw = (pos & 32) ? hi[q] : lo[q];
              wq = __int_as_float(__float_as_int(wq) & -(int)((mw >> (pos & 31)) & 1u));
            }
            acc[q] = fmaf(v[u], wq, acc[q]);
          }
        }
      }
    }
  }
}

template <int CPL, bool MASKED>
__global__ __launch_bounds__(256) void sparse_project_kernel(SpArgs a, const float* __restrict__ Wcat, int64_t ldw,
                                                             float* __restrict__ Wh, float* __restrict__ Sk, float* __restrict__ s) {
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
  if (i >= a.n) return;
  const int fp_shift = ilog2_dev(a.Fp);
  const uint64_t seed = (MASKED && !a.bits) ? *a.g.seed : 0ull;
  float acc[CPL];
  int cc[CPL], hh[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    acc[q] = 0.f;
    const int c = lane + 64 * q;
    cc[q] = c < a.ncols ? c : a.ncols - 1;           // clamped: loads stay unconditional, the value is not stored
    hh[q] = MASKED ? sp_head(a, cc[q], fp_shift) : 0;   // < H under a mask: no score columns then (sp_setup)
  }
  const int e0 = __builtin_amdgcn_readfirstlane(a.ptr[i]), e1 = __builtin_amdgcn_readfirstlane(a.ptr[i + 1]);
  sp_walk<CPL, false, MASKED>(a, seed, __builtin_amdgcn_readfirstlane(i), e0, e1, lane, cc, hh, acc, Wcat, ldw, nullptr, 0);
  const float scale = MASKED ? a.g.scale : 1.f;
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    const int c = lane + 64 * q;
    if (c >= a.ncols) continue;
    const float v = acc[q] * scale;
    if (c < a.R) Wh[(int64_t)i * a.R + c] = v;
    else if (a.has_skip && c < 2 * a.R) Sk[(int64_t)i * a.R + (c - a.R)] = v;
    else s[(int64_t)i * a.H + (c - a.R * (a.has_skip ? 2 : 1))] = v;
  }
}

// D1 = dWh [n x R]; D2 = Gp rows (the skip projection's upstream gradient) with row stride ld2, or nullptr.
// A feature column's non-zeros are cut into SEGMENTS of at most SP_SEG entries (features.py): one wave per segment leaves its
// partial sums, a second launch adds a column's segments in order -- a word that occurs in a third of the documents (a
// 900-entry column of Cora, 6000 of Pubmed) is no longer one wave's serial walk.
[[maybe_unused]] constexpr int SP_SEG = 128;   // features.py SEGMENT
template <int CPL, bool MASKED>
__global__ __launch_bounds__(256) void sparse_wgrad_kernel(SpArgs a, int nseg, const int32_t* __restrict__ seg_col,
                                                           const int32_t* __restrict__ seg_begin, const int32_t* __restrict__ seg_end,
                                                           const float* __restrict__ D1, const float* __restrict__ D2, int64_t ld2,
                                                           float* __restrict__ part) {
  const int sg = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
  if (sg >= nseg) return;
  const int k = __builtin_amdgcn_readfirstlane(seg_col[sg]);
  const int fp_shift = ilog2_dev(a.Fp);
  const uint64_t seed = (MASKED && !a.bits) ? *a.g.seed : 0ull;
  float acc[CPL];
  int cc[CPL], hh[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    acc[q] = 0.f;
    const int c = lane + 64 * q;
    cc[q] = c < a.ncols ? c : a.ncols - 1;
    hh[q] = MASKED ? sp_head(a, cc[q], fp_shift) : 0;   // < H under a mask: no score columns then (sp_setup)
  }
  const int e0 = __builtin_amdgcn_readfirstlane(seg_begin[sg]), e1 = __builtin_amdgcn_readfirstlane(seg_end[sg]);
  sp_walk<CPL, true, MASKED>(a, seed, k, e0, e1, lane, cc, hh, acc, D1, (int64_t)a.R, D2, ld2);
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    const int c = lane + 64 * q;
    if (c < a.ncols) part[(int64_t)sg * a.ncols + c] = acc[q];
  }
}

// column k = the sum of its segments' partial rows, in segment order; written straight into the [H x Fin x F'] layout
__global__ __launch_bounds__(256) void sparse_wgrad_reduce_kernel(SpArgs a, const int32_t* __restrict__ colseg, const float* __restrict__ part,
                                                                  float* __restrict__ dW, float* __restrict__ dWs) {
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (k >= a.Fin) return;
  const int fp_shift = ilog2_dev(a.Fp);
  const float scale = a.masked ? a.g.scale : 1.f;
  const int s0 = colseg[k], s1 = colseg[k + 1];
  for (int c = lane; c < a.ncols; c += 64) {
    float t = 0.f;
    for (int sg = s0; sg < s1; ++sg) t += part[(int64_t)sg * a.ncols + c];
    const int cr = c < a.R ? c : c - a.R, h = cr >> fp_shift, f = cr & (a.Fp - 1);
    if (f < a.Fo) (c < a.R ? dW : dWs)[((int64_t)h * a.Fin + k) * a.Fo + f] = t * scale;
  }
}

}  // namespace pygat

using namespace pygat;

static int sp_setup(SpArgs* a, const char* what, int n, int Fin, int H, int Fo, const int32_t* ptr, const int32_t* idx, const float* val,
                    float p, const void* seed, int stream_id, const unsigned char* bits, bool skip, bool with_s) {
  const int Fp = padded_width(Fo);
  if (!(n > 0 && Fin > 0 && H > 0 && Fp > 0 && ptr && idx && val)) { set_error("%s: bad arguments", what); return PYGAT_EINVAL; }
  a->n = n; a->Fin = Fin; a->H = H; a->Fo = Fo; a->Fp = Fp; a->R = H * Fp; a->has_skip = skip ? 1 : 0; a->with_s = with_s ? 1 : 0;
  a->ncols = a->R * (skip ? 2 : 1) + (with_s ? H : 0);
  a->ptr = ptr; a->idx = idx; a->val = val; a->bits = bits;
  a->masked = p > 0.f ? 1 : 0;
  if (a->ncols > 64 * SP_CPL) { set_error("%s: %d output columns exceed %d", what, a->ncols, 64 * SP_CPL); return PYGAT_EINVAL; }
  if (a->masked) {
    if (!seed && !bits) { set_error("%s: dropout needs a seed or explicit mask bits", what); return PYGAT_EINVAL; }
    if (!make_rng(p, seed, (uint32_t)stream_id, &a->g)) { set_error("%s: p=%g outside [0,1]", what, (double)p); return PYGAT_EINVAL; }
    if (H > 8 && bits) { set_error("%s: explicit mask bits hold 8 heads", what); return PYGAT_EINVAL; }
    if (H > 32) { set_error("%s: dropout decisions are drawn for at most 32 heads", what); return PYGAT_EINVAL; }
    a->hq_shift = 0;
    while ((4 << a->hq_shift) < H) ++a->hq_shift;
  } else {
    a->g.seed = nullptr; a->g.stream_id = 0; a->g.thresh = 0xFFFFFFFFu; a->g.scale = 1.f; a->hq_shift = 0;
  }
  return PYGAT_OK;
}

#define PYGAT_SP_DISPATCH_M(KERNEL, M, GRID, ...)                                                        \
  do {                                                                                                   \
    const int cpl__ = (int)cdiv(a.ncols, 64);                                                            \
    if (cpl__ <= 1) hipLaunchKernelGGL((KERNEL<1, M>), GRID, dim3(256), 0, st, a, __VA_ARGS__);          \
    else if (cpl__ <= 2) hipLaunchKernelGGL((KERNEL<2, M>), GRID, dim3(256), 0, st, a, __VA_ARGS__);     \
    else if (cpl__ <= 4) hipLaunchKernelGGL((KERNEL<4, M>), GRID, dim3(256), 0, st, a, __VA_ARGS__);     \
    else hipLaunchKernelGGL((KERNEL<SP_CPL, M>), GRID, dim3(256), 0, st, a, __VA_ARGS__);                \
  } while (0)
#define PYGAT_SP_DISPATCH(KERNEL, GRID, ...)                               \
  do {                                                                     \
    if (a.masked) PYGAT_SP_DISPATCH_M(KERNEL, true, GRID, __VA_ARGS__);    \
    else PYGAT_SP_DISPATCH_M(KERNEL, false, GRID, __VA_ARGS__);            \
  } while (0)

static bool sp_fits_32bit(int64_t rows, int64_t ld) { return rows * ld * 4 < ((int64_t)1 << 32); }

extern "C" int pygat_project_sparse(int n, int Fin, int H, int Fo, const int32_t* rowptr, const int32_t* col, const float* val,
                                    const float* Wcat, int64_t ldw, float p, const void* seed, int stream_id,
                                    const unsigned char* bits, float* Wh, float* Sk, float* s, void* stream) {
  SpArgs a;
  int rc = sp_setup(&a, "project_sparse", n, Fin, H, Fo, rowptr, col, val, p, seed, stream_id, bits, Sk != nullptr, s != nullptr);
  if (rc) return rc;
  PYGAT_REQUIRE(Wcat && Wh && ldw >= a.ncols, "project_sparse: bad arguments (ldw=%lld, %d columns)", (long long)ldw, a.ncols);
  PYGAT_REQUIRE(sp_fits_32bit(Fin, ldw), "project_sparse: the weight table exceeds 4 GiB");
  PYGAT_REQUIRE(!(a.masked && s), "project_sparse: the score columns are not formed under dropout (pygat_attn_scores does, on the masked Wh)");
  hipStream_t st = (hipStream_t)stream;
  PYGAT_SP_DISPATCH(sparse_project_kernel, dim3((unsigned)cdiv(n, 4)), Wcat, ldw, Wh, Sk, s);
  PYGAT_CHECK_LAUNCH("project_sparse");
  return PYGAT_OK;
}

extern "C" size_t pygat_wgrad_sparse_workspace_bytes(int nseg, int H, int Fo, int skip) {
  const int Fp = padded_width(Fo);
  if (nseg <= 0 || H <= 0 || Fp == 0) return 0;
  return (size_t)nseg * (size_t)(H * Fp * (skip ? 2 : 1)) * sizeof(float);
}

extern "C" int pygat_wgrad_sparse(int n, int Fin, int H, int Fo, int nseg, const int32_t* colseg, const int32_t* seg_col,
                                  const int32_t* seg_begin, const int32_t* seg_end, const int32_t* row, const float* val, float p,
                                  const void* seed, int stream_id, const unsigned char* bits, const float* dWh, const float* Gp,
                                  int64_t ldg, void* ws, float* dW, float* dWskip, void* stream) {
  SpArgs a;
  int rc = sp_setup(&a, "wgrad_sparse", n, Fin, H, Fo, colseg, row, val, p, seed, stream_id, bits, Gp != nullptr, false);
  if (rc) return rc;
  PYGAT_REQUIRE(nseg >= Fin && seg_col && seg_begin && seg_end && ws && dWh && dW && (!Gp || (dWskip && ldg >= a.R)),
                "wgrad_sparse: bad arguments");
  PYGAT_REQUIRE(sp_fits_32bit(n, a.R) && (!Gp || sp_fits_32bit(n, ldg)), "wgrad_sparse: the gradient tables exceed 4 GiB (n=%d)", n);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)ws;
  PYGAT_SP_DISPATCH(sparse_wgrad_kernel, dim3((unsigned)cdiv(nseg, 4)), nseg, seg_col, seg_begin, seg_end, dWh, Gp, ldg, part);
  PYGAT_CHECK_LAUNCH("wgrad_sparse");
  hipLaunchKernelGGL(sparse_wgrad_reduce_kernel, dim3((unsigned)cdiv(Fin, 4)), dim3(256), 0, st, a, colseg, (const float*)part, dW, dWskip);
  PYGAT_CHECK_LAUNCH("wgrad_sparse(reduce)");
  return PYGAT_OK;
}
