// K0 -- graph intake for the GAT hot path (gfx950).
//
// Replaces `edge = adj.nonzero().t()` (reference layers.py:129), which the
// reference re-derives from the dense N x N adjacency in every forward of every
// head, and the `adj > 0` mask of layers.py:41.  Here the pattern is extracted
// ONCE into CSR and cached by the caller.  Also builds the mirror permutation
// used by the atomics-free backward (K4).
#include "common.h"

namespace pygat {

// ---- dense -> row counts: one wave per row, coalesced 256-B sweeps -------------
__global__ __launch_bounds__(256) void dense_row_counts_kernel(const float* __restrict__ adj, int n,
                                                               int64_t ld, int mode,
                                                               int32_t* __restrict__ counts) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float* r = adj + (int64_t)row * ld;
  int cnt = 0;
  for (int c0 = 0; c0 < n; c0 += 64) {
    int c = c0 + lane;
    float v = (c < n) ? r[c] : 0.f;
    bool nz = mode ? (v > 0.f) : (v != 0.f);
    cnt += __popcll(__ballot(nz));
  }
  if (lane == 0) counts[row] = cnt;
}

// ---- dense -> sorted column lists: ordered compaction with ballot + mbcnt -------
__global__ __launch_bounds__(256) void dense_fill_cols_kernel(const float* __restrict__ adj, int n,
                                                              int64_t ld, int mode,
                                                              const int32_t* __restrict__ rowptr,
                                                              int32_t* __restrict__ col) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float* r = adj + (int64_t)row * ld;
  int64_t base = rowptr[row];
  for (int c0 = 0; c0 < n; c0 += 64) {
    int c = c0 + lane;
    float v = (c < n) ? r[c] : 0.f;
    bool nz = mode ? (v > 0.f) : (v != 0.f);
    unsigned long long b = __ballot(nz);
    int before = __popcll(b & ((1ull << lane) - 1ull));
    if (nz) col[base + before] = c;
    base += __popcll(b);
  }
}

// ---- exclusive scan (three phases, deterministic) -------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;                       // per thread
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 4096 per block

__device__ __forceinline__ int block_exclusive_scan(int v, int* total) {
  __shared__ int wsum[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int y = __shfl_up(x, off);
    if (lane >= off) x += y;
  }
  if (lane == 63) wsum[w] = x;
  __syncthreads();
  int pre = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k < w) pre += wsum[k];
    tot += wsum[k];
  }
  __syncthreads();
  *total = tot;
  return pre + x - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_sums_kernel(const int32_t* __restrict__ in,
                                                                       int64_t n,
                                                                       int32_t* __restrict__ tile_sums) {
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k)
    if (base + k < n) s += in[base + k];
  int tot;
  block_exclusive_scan(s, &tot);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// one block scans all tile sums in place (sequential over chunks of 256 with a carry)
__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_offsets_kernel(int32_t* __restrict__ tile_sums,
                                                                          int64_t ntiles) {
  int carry = 0;
  for (int64_t b = 0; b < ntiles; b += SCAN_THREADS) {
    int64_t i = b + threadIdx.x;
    int v = (i < ntiles) ? tile_sums[i] : 0;
    int tot;
    int ex = block_exclusive_scan(v, &tot);
    if (i < ntiles) tile_sums[i] = carry + ex;
    carry += tot;
  }
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(const int32_t* __restrict__ in, int64_t n,
                                                                   const int32_t* __restrict__ tile_off,
                                                                   int32_t* __restrict__ out) {
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  int s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    v[k] = (base + k < n) ? in[base + k] : 0;
    s += v[k];
  }
  int tot;
  int run = block_exclusive_scan(s, &tot) + tile_off[blockIdx.x];
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    run += v[k];
    if (base + k < n) out[base + k + 1] = run;  // out[i+1] = inclusive sum
  }
}

// ---- mirror permutation of a symmetric CSR with sorted rows ----------------------
// one wave per row i; lane handles edge k = (i,j): binary-search i inside row j.
__global__ __launch_bounds__(256) void csr_symmetric_perm_kernel(int n, const int32_t* __restrict__ rowptr,
                                                                 const int32_t* __restrict__ col,
                                                                 int32_t* __restrict__ perm,
                                                                 int32_t* __restrict__ flags) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int b = rowptr[i], e = rowptr[i + 1];
  if (e == b && lane == 0) flags[1] = 1;
  for (int k = b + lane; k < e; k += 64) {
    const int j = col[k];
    int lo = rowptr[j], hi = rowptr[j + 1];
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (col[mid] < i) lo = mid + 1; else hi = mid;
    }
    if (lo < rowptr[j + 1] && col[lo] == i) perm[k] = lo;
    else { perm[k] = k; flags[0] = 1; }
  }
}

// ---- row-snapped slot borders for the nnz-split kernels ---------------------------
__global__ __launch_bounds__(256) void slot_bounds_kernel(int64_t nnz, int ts, int64_t nslots,
                                                          const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ edge_rc,
                                                          int32_t* __restrict__ sb) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k > nslots) return;
  if (k == nslots) { sb[k] = (int32_t)nnz; return; }
  const int64_t pos = k * ts;
  const int r = edge_rc[2 * pos];        // row that contains edge `pos`
  int64_t b = pos;
  if (rowptr[r] != pos) {                // inside a row: move to its end if that is close
    const int64_t nxt = rowptr[r + 1];
    if (nxt - pos < ts / 2 && nxt < nnz) b = nxt;
  }
  sb[k] = (int32_t)b;
}

// (first edge, end edge, first row, flags) of every slot: what a slot's start-up otherwise fetches through the chain
// slot_begin -> edge_rc -> rowptr, and the rowptr load at its end, in one 16-byte record
__global__ __launch_bounds__(256) void slot_meta_kernel(int64_t nnz, int ts, int64_t nslots, const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ edge_rc, const int32_t* __restrict__ sb,
                                                        int4* __restrict__ meta) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= nslots) return;
  int64_t e0, e1;
  if (sb) { e0 = sb[k]; e1 = sb[k + 1]; }
  else { e0 = k * ts; e1 = (e0 + ts < nnz) ? e0 + ts : nnz; }
  const int rf = edge_rc[2 * e0], rl = edge_rc[2 * (e1 - 1)];
  const int flags = (rowptr[rf] < e0 ? 1 : 0) | (rowptr[rl + 1] > e1 ? 2 : 0);
  meta[k] = make_int4((int)e0, (int)e1, rf, flags);
}

}  // namespace pygat

using namespace pygat;

extern "C" int pygat_slot_meta(int n, int64_t nnz, const int32_t* rowptr, const int32_t* edge_rc, int slot_edges,
                               const int32_t* slot_begin, int32_t* slot_meta, void* stream) {
  PYGAT_REQUIRE(n > 0 && nnz > 0 && nnz < ((int64_t)1 << 31) && rowptr && edge_rc && slot_meta && slot_edges >= 4 &&
                    (slot_edges & 3) == 0 && aligned16(slot_meta), "slot_meta: bad arguments");
  const int64_t nslots = cdiv(nnz, slot_edges);
  hipLaunchKernelGGL(slot_meta_kernel, dim3((unsigned)cdiv(nslots, 256)), dim3(256), 0, (hipStream_t)stream, nnz, slot_edges,
                     nslots, rowptr, edge_rc, slot_begin, reinterpret_cast<int4*>(slot_meta));
  PYGAT_CHECK_LAUNCH("slot_meta");
  return PYGAT_OK;
}

extern "C" int pygat_dense_row_counts(const float* adj, int n, int64_t ld, int mode, int32_t* counts,
                                      void* stream) {
  PYGAT_REQUIRE(adj && counts && n > 0 && ld >= n, "dense_row_counts: bad arguments (n=%d ld=%lld)", n,
                (long long)ld);
  hipLaunchKernelGGL(dense_row_counts_kernel, dim3((unsigned)cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream,
                     adj, n, ld, mode, counts);
  PYGAT_CHECK_LAUNCH("dense_row_counts");
  return PYGAT_OK;
}

extern "C" int pygat_dense_fill_cols(const float* adj, int n, int64_t ld, int mode, const int32_t* rowptr,
                                     int32_t* col, void* stream) {
  PYGAT_REQUIRE(adj && rowptr && col && n > 0 && ld >= n, "dense_fill_cols: bad arguments");
  hipLaunchKernelGGL(dense_fill_cols_kernel, dim3((unsigned)cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream,
                     adj, n, ld, mode, rowptr, col);
  PYGAT_CHECK_LAUNCH("dense_fill_cols");
  return PYGAT_OK;
}

extern "C" size_t pygat_scan_workspace_bytes(int64_t n) {
  return (size_t)(cdiv(n > 0 ? n : 1, SCAN_TILE) + 1) * sizeof(int32_t);
}

extern "C" int pygat_exclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out, void* ws, void* stream) {
  PYGAT_REQUIRE(in && out && ws && n > 0, "exclusive_scan: bad arguments");
  const int64_t ntiles = cdiv(n, SCAN_TILE);
  int32_t* tiles = (int32_t*)ws;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, st, in, n, tiles);
  hipLaunchKernelGGL(scan_tile_offsets_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, tiles, ntiles);
  hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, st, in, n, tiles, out);
  PYGAT_CHECK_LAUNCH("exclusive_scan");
  return PYGAT_OK;
}

extern "C" int pygat_csr_symmetric_perm(int n, const int32_t* rowptr, const int32_t* col, int32_t* perm,
                                        int32_t* flags, void* stream) {
  PYGAT_REQUIRE(rowptr && col && perm && flags && n > 0, "csr_symmetric_perm: bad arguments");
  hipLaunchKernelGGL(csr_symmetric_perm_kernel, dim3((unsigned)cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream,
                     n, rowptr, col, perm, flags);
  PYGAT_CHECK_LAUNCH("csr_symmetric_perm");
  return PYGAT_OK;
}

extern "C" int pygat_slot_bounds(int n, int64_t nnz, const int32_t* rowptr, const int32_t* edge_rc, int slot_edges,
                                 int32_t* slot_begin, void* stream) {
  PYGAT_REQUIRE(n > 0 && nnz > 0 && rowptr && edge_rc && slot_begin && slot_edges >= 4 && (slot_edges & 3) == 0,
                "slot_bounds: bad arguments");
  const int64_t nslots = cdiv(nnz, slot_edges);
  hipLaunchKernelGGL(slot_bounds_kernel, dim3((unsigned)cdiv(nslots + 1, 256)), dim3(256), 0, (hipStream_t)stream, nnz,
                     slot_edges, nslots, rowptr, edge_rc, slot_begin);
  PYGAT_CHECK_LAUNCH("slot_bounds");
  return PYGAT_OK;
}
