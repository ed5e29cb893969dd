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
  DropRng g;                // masked, bits == nullptr
  const unsigned char* bits;  // [n x Fin] explicit decisions or nullptr
};

// head of output column c: [0, R) W heads, [R, 2R) skip heads (has_skip), behind them the H score columns (never masked)
__device__ __forceinline__ int sp_head(const SpArgs& a, int c, int fp_shift) { return (c < a.R ? c : c - a.R) >> fp_shift; }

__device__ __forceinline__ bool sp_keep(const SpArgs& a, uint64_t seed, int i, int k, int h) {
  if (a.bits) return (a.bits[(int64_t)i * a.Fin + k] >> h) & 1u;
  const uint4 w = draw4(a.g, seed, (uint32_t)i, (uint32_t)h, (uint32_t)(k >> 2));
  return word_of(w, k & 3) < a.g.thresh;
}

template <int CPL>
__global__ __launch_bounds__(256) void sparse_project_kernel(SpArgs a, const float* __restrict__ Wcat, int64_t ldw,
                                                             float* __restrict__ Wh, float* __restrict__ Sk, float* __restrict__ s) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= a.n) return;
  const int fp_shift = ilog2_dev(a.Fp);
  const uint64_t seed = (a.masked && !a.bits) ? *a.g.seed : 0ull;
  float acc[CPL];
  int cc[CPL], hh[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    acc[q] = 0.f;
    const int c = lane + 64 * q;
    cc[q] = c < a.ncols ? c : a.ncols - 1;           // clamped: loads stay unconditional, the value is not stored
    hh[q] = sp_head(a, cc[q], fp_shift);
  }
  const int e0 = a.ptr[i], e1 = a.ptr[i + 1];
  for (int eb = e0; eb < e1; eb += 64) {
    const int cnt = (e1 - eb < 64) ? e1 - eb : 64;
    const int kk = a.idx[eb + (lane < cnt ? lane : cnt - 1)];
    const float vv = a.val[eb + (lane < cnt ? lane : cnt - 1)];
    for (int j = 0; j < cnt; j += 4) {
      int k4[4];
      float v4[4], w4[4][CPL];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int jj = (j + u < cnt) ? j + u : cnt - 1;
        k4[u] = __shfl(kk, jj);
        v4[u] = (j + u < cnt) ? __shfl(vv, jj) : 0.f;
#pragma unroll
        for (int q = 0; q < CPL; ++q) w4[u][q] = Wcat[(int64_t)k4[u] * ldw + cc[q]];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          bool keep = true;
          if (a.masked) keep = sp_keep(a, seed, i, k4[u], hh[q]);
          acc[q] = keep ? fmaf(v4[u], w4[u][q], acc[q]) : acc[q];
        }
    }
  }
  const float scale = a.masked ? a.g.scale : 1.f;
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

// D1 = dWh [n x R]; D2 = Gp rows (the skip projection's upstream gradient) with row stride ld2, or nullptr
template <int CPL>
__global__ __launch_bounds__(256) void sparse_wgrad_kernel(SpArgs a, const float* __restrict__ D1, const float* __restrict__ D2,
                                                           int64_t ld2, float* __restrict__ dW, float* __restrict__ dWs) {
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (k >= a.Fin) return;
  const int fp_shift = ilog2_dev(a.Fp);
  const uint64_t seed = (a.masked && !a.bits) ? *a.g.seed : 0ull;
  float acc[CPL];
  int cc[CPL], hh[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    acc[q] = 0.f;
    const int c = lane + 64 * q;
    cc[q] = c < a.ncols ? c : a.ncols - 1;
    hh[q] = sp_head(a, cc[q], fp_shift);
  }
  const int e0 = a.ptr[k], e1 = a.ptr[k + 1];
  for (int eb = e0; eb < e1; eb += 64) {
    const int cnt = (e1 - eb < 64) ? e1 - eb : 64;
    const int ii = a.idx[eb + (lane < cnt ? lane : cnt - 1)];
    const float vv = a.val[eb + (lane < cnt ? lane : cnt - 1)];
    for (int j = 0; j < cnt; j += 4) {
      int i4[4];
      float v4[4], d4[4][CPL];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int jj = (j + u < cnt) ? j + u : cnt - 1;
        i4[u] = __shfl(ii, jj);
        v4[u] = (j + u < cnt) ? __shfl(vv, jj) : 0.f;
#pragma unroll
        for (int q = 0; q < CPL; ++q)
          d4[u][q] = cc[q] < a.R ? D1[(int64_t)i4[u] * a.R + cc[q]] : D2[(int64_t)i4[u] * ld2 + (cc[q] - a.R)];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          bool keep = true;
          if (a.masked) keep = sp_keep(a, seed, i4[u], k, hh[q]);
          acc[q] = keep ? fmaf(v4[u], d4[u][q], acc[q]) : acc[q];
        }
    }
  }
  const float scale = a.masked ? a.g.scale : 1.f;
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    const int c = lane + 64 * q;
    if (c >= a.ncols) continue;
    const int cr = c < a.R ? c : c - a.R, h = cr >> fp_shift, f = cr & (a.Fp - 1);
    if (f >= a.Fo) continue;                                           // padding columns of a head
    float* o = (c < a.R ? dW : dWs) + ((int64_t)h * a.Fin + k) * a.Fo + f;   // straight into the [H x Fin x F'] layout
    *o = acc[q] * scale;
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
  } else {
    a->g.seed = nullptr; a->g.stream_id = 0; a->g.thresh = 0xFFFFFFFFu; a->g.scale = 1.f;
  }
  return PYGAT_OK;
}

#define PYGAT_SP_DISPATCH(KERNEL, GRID, ...)                                                             \
  do {                                                                                                   \
    const int cpl__ = (int)cdiv(a.ncols, 64);                                                            \
    if (cpl__ <= 1) hipLaunchKernelGGL((KERNEL<1>), GRID, dim3(256), 0, st, a, __VA_ARGS__);             \
    else if (cpl__ <= 2) hipLaunchKernelGGL((KERNEL<2>), GRID, dim3(256), 0, st, a, __VA_ARGS__);        \
    else if (cpl__ <= 4) hipLaunchKernelGGL((KERNEL<4>), GRID, dim3(256), 0, st, a, __VA_ARGS__);        \
    else hipLaunchKernelGGL((KERNEL<SP_CPL>), GRID, dim3(256), 0, st, a, __VA_ARGS__);                   \
  } while (0)

extern "C" int pygat_project_sparse(int n, int Fin, int H, int Fo, const int32_t* rowptr, const int32_t* col, const float* val,
                                    const float* Wcat, int64_t ldw, float p, const void* seed, int stream_id,
                                    const unsigned char* bits, float* Wh, float* Sk, float* s, void* stream) {
  SpArgs a;
  int rc = sp_setup(&a, "project_sparse", n, Fin, H, Fo, rowptr, col, val, p, seed, stream_id, bits, Sk != nullptr, s != nullptr);
  if (rc) return rc;
  PYGAT_REQUIRE(Wcat && Wh && ldw >= a.ncols, "project_sparse: bad arguments (ldw=%lld, %d columns)", (long long)ldw, a.ncols);
  PYGAT_REQUIRE(!(a.masked && s), "project_sparse: the score columns are not formed under dropout (pygat_attn_scores does, on the masked Wh)");
  hipStream_t st = (hipStream_t)stream;
  PYGAT_SP_DISPATCH(sparse_project_kernel, dim3((unsigned)cdiv(n, 4)), Wcat, ldw, Wh, Sk, s);
  PYGAT_CHECK_LAUNCH("project_sparse");
  return PYGAT_OK;
}

extern "C" int pygat_wgrad_sparse(int n, int Fin, int H, int Fo, const int32_t* colptr, const int32_t* row, const float* val, float p,
                                  const void* seed, int stream_id, const unsigned char* bits, const float* dWh, const float* Gp,
                                  int64_t ldg, float* dW, float* dWskip, void* stream) {
  SpArgs a;
  int rc = sp_setup(&a, "wgrad_sparse", n, Fin, H, Fo, colptr, row, val, p, seed, stream_id, bits, Gp != nullptr, false);
  if (rc) return rc;
  PYGAT_REQUIRE(dWh && dW && (!Gp || (dWskip && ldg >= a.R)), "wgrad_sparse: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  PYGAT_SP_DISPATCH(sparse_wgrad_kernel, dim3((unsigned)cdiv(Fin, 4)), dWh, Gp, ldg, dW, dWskip);
  PYGAT_CHECK_LAUNCH("wgrad_sparse");
  return PYGAT_OK;
}
