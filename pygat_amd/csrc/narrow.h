// Internal: the narrow-level kernels of k10_narrow.hip, as pygat_project_dropout / pygat_wgrad_dropout (k1_gemm.hip) reach them.
#pragma once
#include "common.h"

namespace pygat {

bool narrow_takes(int Fin, int H, int Fo, bool skip);
int narrow_project(int n, int Fin, int H, int Fo, const float* X, int64_t ldx, const unsigned char* bits, float p, const float* Wcat,
                   int64_t ldw, float* Wh, float* Sk, hipStream_t st);
// split_k = slabs of rows, one wave each; ws >= split_k * Fin * ntot floats (pygat_wgrad_dropout_workspace_bytes)
int narrow_wgrad(int n, int Fin, int H, int Fo, const float* X, int64_t ldx, const unsigned char* bits, float p, const float* dWh,
                 const float* Gp, int64_t ldgp, float* dWc, int split_k, void* ws, hipStream_t st);

}  // namespace pygat
