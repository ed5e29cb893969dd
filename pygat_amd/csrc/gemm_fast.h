// Argument blocks of the two streamed fp32 GEMM fast paths (k1_gemm_smallk.hip: fp32 MFMA; k1_gemm_x3.hip: the same
// products on the bf16 MFMA pipe from an exact three-way split of every fp32 operand).  Not part of the C ABI.
#pragma once
#include "common.h"

namespace pygat {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct SmallKArgs {
  int M, N, K;
  const float* A;
  int64_t lda;
  const float* B;
  int64_t ldb;
  pygat_out_segments out;
  int accumulate;
  int tiles_m;  // 256-row tiles
  int transB;   // B given as [N x K] (dX = dWh W^T)
  // SR (k1_gemm_x3.hip): s[i, h] = sum_f C[i, h Fp + f] a_pad[h, 0, f] taken from the ACCUMULATORS in the epilogue --
  // Wh_i . a_src as the reference forms it (layers.py:60) -- for head widths of 8 / 16 columns (a head lies inside
  // one 32-column MFMA tile, whose columns are lanes: a DPP reduction over 8 / 16 lanes).  N == H * sr_fp.
  const float* sr_a;   // a_pad [H][2][Fp]
  int sr_fp;
  // SV: s[i, h] = sum_k A[i,k] svec[k, h] for up to 8 extra "columns" WITHOUT MFMA tiles for them.  The lane that
  // streams row i has its k values in registers anyway: each half-wave takes 4 of the columns, 4 FMAs per loaded
  // float against an LDS broadcast of svec -- VALU work that issues in the shadow of the MFMAs, no cross-lane
  // reduction (the projection's s_i = x_i . (W_h a_src), layers.py:60: a fifth 32-column tile for 8 columns costs
  // 20 % of the kernel; reducing the accumulators across lanes in the epilogue cost more than that tile).
  const float* svec;   // [K x sv_ld], columns 0 .. sv_n-1 used (sv_n <= 8)
  int64_t sv_ld;
  int sv_n;
  float* s_out;        // [M x s_ld]
  int64_t s_ld;
  // (k1_gemm_x3.hip, set by try_gemm_smallk_x3) every output segment starts at a multiple of 32 columns with 16-byte aligned
  // rows: a lane's four consecutive columns of a row go out as one 16-byte store; s_vec: the same for the lane's s values
  int vec_out, s_vec;
  int lds_rows;   // 1: the epilogue turns every 32 x 32 tile through the wave's own LDS patch and stores WHOLE 128-byte lines
};

struct TnArgs {
  int M, N;
  int64_t K;
  const float* A;
  int64_t lda;
  const float* B;
  int64_t ldb;
  int64_t k_per_split;
  float* ws;  // [splits][M][N]
  // optional second B operand: columns [N1, N) come from B2 (its own leading dimension); N1 % 32 == 0 so a
  // 32-column tile never straddles the two.  N1 == N: none.  (dW = X^T [dWh | ds], pygat_wgrad)
  int N1;
  const float* B2;
  int64_t ldb2;
};

#ifdef __HIPCC__
// Output segment of column `col`: leading dimension and address of (row 0, col).  Written as selects over CONSTANT
// indices: `out.ld[sgm]` with a per-lane sgm makes hipcc read the kernel arguments with a vector load -- and the
// s_waitcnt vmcnt(0) in front of its use drains every prefetched A chunk and every store of the wave, per column tile.
__device__ __forceinline__ float* out_segment(const pygat_out_segments& out, int col, int64_t& ld) {
  ld = out.ld[0];
  float* p = out.ptr[0];
  int c0 = 0;
#pragma unroll
  for (int q = 1; q < PYGAT_MAX_SEGMENTS; ++q)
    if (q < out.nseg && col >= out.col_start[q]) { ld = out.ld[q]; p = out.ptr[q]; c0 = out.col_start[q]; }
  return p + (col - c0);
}
#endif

// product mode of a call (include/pygat_amd.h, PYGAT_GEMM_*) -> true: the streamed fast paths split each operand exactly
// into three bf16 pieces and sum all nine piece products in fp32; false: every product on v_mfma_f32_32x32x2_f32
bool gemm_split(int mode);

// each returns 1 (or the slab count) if it took the call, 0 if the shape does not qualify, < 0 on a launch error
int try_gemm_smallk_x3(const SmallKArgs& g, int NT, dim3 grid, hipStream_t st);
// [Wh | s] of a projection whose heads are 8 or 16 columns wide: s from the accumulators (SmallKArgs::sr_a)
int try_project_x3(int n, int Fin, int H, int Fp, const float* X, int64_t ldx, const float* Wcat, int64_t ldw, float* Wh,
                   float* s, const float* a_pad, bool split, hipStream_t st);
int try_gemm_tn_x3(const TnArgs& g, int splits, hipStream_t st);
// any operand layout, K % 4 == 0, aligned rows, M, N > 64 (k1_gemm_x3.hip, gemm_x3g_kernel): 1 if it took the call
int try_gemm_x3g(int transA, int transB, int M, int N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                 const pygat_out_segments* out, int accumulate, int splits, int64_t k_per_split, float* ws, hipStream_t st,
                 ColBlocks ab, ColBlocks cb, int* splits_used);   // *splits_used: the slabs written (<= splits)

}  // namespace pygat
