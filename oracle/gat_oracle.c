/*
 * gat_oracle.c -- plain C (OpenMP) restatement of one GAT level, forward + backward.
 *
 * TEST INFRASTRUCTURE ONLY: linked/called from tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  The product (pygat_amd/) never touches it.
 *
 * PARITY UNPINNED (same status as oracle/gat_oracle.py): the reference's layers.py cannot be
 * imported in the build container (layers.py:5 needs torch_scatter, absent) and the reference
 * ships no fixtures.  This file is pinned against gat_oracle.py (tests/test_oracle_c.py), which
 * restates the reference line by line.
 *
 * It is the only CPU path that can run the 1M-node / 10M-edge configuration: the reference
 * itself needs a dense N x N adjacency (utils.py:55) and a dense N x N gradient (layers.py:85).
 *
 * Algorithm (per head h; see SURVEY.md 8(a) a11 for the gradient derivation):
 *   Wh = X W_h                                     layers.py:134
 *   s = Wh a[:F], t = Wh a[F:]                     layers.py:60-61 (== a . [Wh_i ; Wh_j], layers.py:141-144)
 *   e_ij = LeakyReLU(s_i + t_j), m_i = max_j e_ij  layers.py:144-145
 *   p_ij = exp(e_ij - m_i), Z_i = sum_j p_ij       layers.py:146,150
 *   hp_i = sum_j p_ij Wh_j / Z_i                   layers.py:156-160
 *   out  = ELU(hp) if concat else hp               layers.py:168-173; heads concatenated
 *          (models.py:32) or averaged (models.py:34)
 *   backward = chain rule through the above, per edge (no N x N as in layers.py:85).
 *
 * gat_oracle_level_v2 (round 4): the same for SpGraphAttentionLayerV2 (layers.py:234-316), per head
 *   Whi = X W[:Fin], Whj = X W[Fin:]               layers.py:268-269
 *   e_ij = a . LeakyReLU(Whi_i + Whj_j)            layers.py:280-283
 *   softmax over row i as above                    layers.py:285-290
 *   hp_i = sum_j p_ij Whi_j / Z_i                  layers.py:296-300 (Whi is what is aggregated)
 * pinned against gat_oracle.py's sparse_head_forward_v2 + torch autograd (tests/test_oracle_c.py).
 *
 * Built twice from this one source (oracle/Makefile): REAL = float -> gat_oracle_level (fp32 like the
 * reference; long sums already run in double), and -DORACLE_F64 -> gat_oracle_level_f64, the fp64 ground
 * truth the full-size GPU tests price BOTH fp32 implementations against (forward atol 1e-5,
 * gradients <= max(1e-5, 4 x the fp32 oracle's own error against fp64)).
 */
#include <math.h>
#ifdef ORACLE_F64
typedef double REAL;
#define FN(name) name##_f64
#define EXPR(x) exp(x)
#define EXPM1R(x) expm1(x)
#else
typedef float REAL;
#define FN(name) name
#define EXPR(x) expf(x)
#define EXPM1R(x) expm1f(x)
#endif
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int FN(gat_oracle_threads)(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* Optional capture of the near-kink edges of the next level call (tests/parity.py, flip-aware comparison): every
 * (head, edge) with |z_ij| <= tau (|s_i| + |t_j|), its logit z and de_ij = alpha_ij (dp_ij - D_i) (dL/de before the
 * LeakyReLU slope).  cap = 0 switches it off.  The order of the entries depends on thread timing. */
static struct { double tau; int64_t cap; int32_t* h; int64_t* e; double* z; double* de; int64_t* count; } kink = {0, 0, 0, 0, 0, 0, 0};
void FN(gat_oracle_capture_kinks)(double tau, int64_t cap, int32_t* h, int64_t* e, double* z, double* de, int64_t* count) {
  kink.tau = tau; kink.cap = cap; kink.h = h; kink.e = e; kink.z = z; kink.de = de; kink.count = count;
  if (count) *count = 0;
}

/* C[M x N] = A[M x K] * B[K x N] (row-major), or with A transposed: C[M x N] = A[K x M]^T B[K x N] */
static void gemm_nn(int64_t M, int N, int K, const REAL* A, const REAL* B, REAL* C) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < M; ++i) {
    REAL* c = C + i * N;
    for (int n = 0; n < N; ++n) c[n] = (REAL)0;
    for (int k = 0; k < K; ++k) {
      const REAL a = A[i * K + k];
      const REAL* b = B + (int64_t)k * N;
      for (int n = 0; n < N; ++n) c[n] += a * b[n];
    }
  }
}

/* dW[K x N] = X[M x K]^T dWh[M x N]: per-thread private accumulators, reduced in thread order */
static void gemm_tn(int64_t M, int N, int K, const REAL* X, const REAL* D, REAL* out) {
  int nt = FN(gat_oracle_threads)();
  double* acc = (double*)calloc((size_t)nt * K * N, sizeof(double));
#pragma omp parallel
  {
#ifdef _OPENMP
    int tid = omp_get_thread_num();
#else
    int tid = 0;
#endif
    double* a = acc + (size_t)tid * K * N;
#pragma omp for schedule(static)
    for (int64_t i = 0; i < M; ++i)
      for (int k = 0; k < K; ++k) {
        const double x = X[i * K + k];
        if (x == 0.0) continue;
        const REAL* d = D + i * N;
        double* r = a + (size_t)k * N;
        for (int n = 0; n < N; ++n) r[n] += x * d[n];
      }
  }
  for (int64_t q = 0; q < (int64_t)K * N; ++q) {
    double s = 0;
    for (int t = 0; t < nt; ++t) s += acc[(size_t)t * K * N + q];
    out[q] = (REAL)s;
  }
  free(acc);
}

/*
 * One level, H heads, eval mode / dropout 0.
 *   X [N x Fin], W [H x Fin x F], a [H x 2F], G [N x H*F] (concat) or [N x F] (mean)
 *   rowptr [N+1], col [E]; rowptr_t/col_t/perm_t: transposed pattern and, per transposed
 *   edge, the index of its forward edge (for a symmetric pattern rowptr_t == rowptr).
 * outputs: out (same shape as G), dW [H x Fin x F], da [H x 2F], dX [N x Fin] (may be NULL).
 * returns 0, or -1 on allocation failure.
 */
int FN(gat_oracle_level)(int64_t N, int64_t E, const int32_t* rowptr, const int32_t* col,
                     const int32_t* rowptr_t, const int32_t* col_t, const int32_t* perm_t,
                     int Fin, int H, int F, REAL alpha, int concat,
                     const REAL* X, const REAL* W, const REAL* a, const REAL* G,
                     REAL* out, REAL* dW, REAL* da, REAL* dX) {
  REAL* Wh = (REAL*)malloc((size_t)N * F * sizeof(REAL));
  REAL* s = (REAL*)malloc((size_t)N * sizeof(REAL));
  REAL* t = (REAL*)malloc((size_t)N * sizeof(REAL));
  REAL* hp = (REAL*)malloc((size_t)N * F * sizeof(REAL));
  REAL* Gp = (REAL*)malloc((size_t)N * F * sizeof(REAL));
  REAL* al = (REAL*)malloc((size_t)E * sizeof(REAL));
  REAL* dz = (REAL*)malloc((size_t)E * sizeof(REAL));
  REAL* ds = (REAL*)malloc((size_t)N * sizeof(REAL));
  REAL* dt = (REAL*)malloc((size_t)N * sizeof(REAL));
  REAL* dWh = (REAL*)malloc((size_t)N * F * sizeof(REAL));
  if (!Wh || !s || !t || !hp || !Gp || !al || !dz || !ds || !dt || !dWh) return -1;
  const int OC = concat ? H * F : F;
  if (!concat) memset(out, 0, (size_t)N * F * sizeof(REAL));
  if (dX) memset(dX, 0, (size_t)N * Fin * sizeof(REAL));

  for (int h = 0; h < H; ++h) {
    const REAL* Wm = W + (size_t)h * Fin * F;
    const REAL* as = a + (size_t)h * 2 * F;
    const REAL* ad = as + F;
    gemm_nn(N, F, Fin, X, Wm, Wh);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
      REAL x = (REAL)0, y = (REAL)0;
      for (int f = 0; f < F; ++f) { x += Wh[i * F + f] * as[f]; y += Wh[i * F + f] * ad[f]; }
      s[i] = x; t[i] = y;
    }
    /* forward row pass + row part of the backward */
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < N; ++i) {
      const int b = rowptr[i], e = rowptr[i + 1];
      REAL m = -INFINITY;
      for (int k = b; k < e; ++k) {
        REAL z = s[i] + t[col[k]];
        REAL ev = z > (REAL)0 ? z : alpha * z;
        al[k] = ev;
        if (ev > m) m = ev;
      }
      /* row sums in double: a hub row has 10^4+ terms and fp32 running sums would make the ORACLE the
         less accurate side of the comparison */
      double Zd = 0.0;
      for (int k = b; k < e; ++k) { al[k] = EXPR(al[k] - m); Zd += al[k]; }
      const REAL Z = (REAL)Zd;
      REAL* hr = hp + i * F;
      for (int f = 0; f < F; ++f) {
        double acc = 0.0;
        for (int k = b; k < e; ++k) acc += (double)al[k] * Wh[(int64_t)col[k] * F + f];
        hr[f] = (REAL)acc;
      }
      double D = 0.0;
      for (int f = 0; f < F; ++f) {
        hr[f] /= Z;
        REAL g, o;
        if (concat) {
          o = hr[f] > (REAL)0 ? hr[f] : EXPM1R(hr[f]);
          out[i * OC + h * F + f] = o;
          g = G[i * OC + h * F + f] * (hr[f] > (REAL)0 ? (REAL)1 : EXPR(hr[f]));
        } else {
          out[i * OC + f] += hr[f] / (REAL)H;
          g = G[i * OC + f] / (REAL)H;
        }
        Gp[i * F + f] = g;
        D += (double)g * hr[f];
      }
      double dsi = 0.0;
      for (int k = b; k < e; ++k) {
        const REAL* wj = Wh + (int64_t)col[k] * F;
        REAL dp = (REAL)0;
        for (int f = 0; f < F; ++f) dp += Gp[i * F + f] * wj[f];
        al[k] /= Z;
        REAL z = s[i] + t[col[k]];
        const REAL de = al[k] * (dp - (REAL)D);
        if (kink.cap > 0 && fabs((double)z) <= kink.tau * (fabs((double)s[i]) + fabs((double)t[col[k]]))) {
          int64_t q;
#pragma omp atomic capture
          q = (*kink.count)++;
          if (q < kink.cap) { kink.h[q] = h; kink.e[q] = k; kink.z[q] = (double)z; kink.de[q] = (double)de; }
        }
        dz[k] = de * (z > (REAL)0 ? (REAL)1 : alpha);
        dsi += dz[k];
      }
      ds[i] = (REAL)dsi;
    }
    /* column pass over the transposed pattern */
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t j = 0; j < N; ++j) {
      REAL* dr = dWh + j * F;
      double dtd = 0.0;
      for (int k = rowptr_t[j]; k < rowptr_t[j + 1]; ++k) dtd += dz[perm_t[k]];
      const REAL dtj = (REAL)dtd;
      dt[j] = dtj;
      for (int f = 0; f < F; ++f) {
        double acc = 0.0;
        for (int k = rowptr_t[j]; k < rowptr_t[j + 1]; ++k) acc += (double)al[perm_t[k]] * Gp[(int64_t)col_t[k] * F + f];
        dr[f] = (REAL)acc + ds[j] * as[f] + dtj * ad[f];
      }
    }
    /* da */
    for (int f = 0; f < F; ++f) {
      double x = 0, y = 0;
#pragma omp parallel for reduction(+ : x, y) schedule(static)
      for (int64_t i = 0; i < N; ++i) { x += (double)ds[i] * Wh[i * F + f]; y += (double)dt[i] * Wh[i * F + f]; }
      da[(size_t)h * 2 * F + f] = (REAL)x;
      da[(size_t)h * 2 * F + F + f] = (REAL)y;
    }
    gemm_tn(N, F, Fin, X, dWh, dW + (size_t)h * Fin * F);
    if (dX) {
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < N; ++i)
        for (int k = 0; k < Fin; ++k) {
          REAL acc = (REAL)0;
          for (int f = 0; f < F; ++f) acc += dWh[i * F + f] * Wm[(size_t)k * F + f];
          dX[i * Fin + k] += acc;
        }
    }
  }
  free(Wh); free(s); free(t); free(hp); free(Gp); free(al); free(dz); free(ds); free(dt); free(dWh);
  return 0;
}


/*
 * One SpGraphAttentionLayerV2 level (layers.py:258-313), H heads, eval mode / dropout 0, no skip projection.
 *   X [N x Fin], W [H x 2Fin x F] (rows 0..Fin-1: the Whi half, layers.py:268; rows Fin..: the Whj half, 269), a [H x F]
 *   (layers.py:249), G as in gat_oracle_level.  outputs: out, dW [H x 2Fin x F], da [H x F], dX [N x Fin] (may be NULL).
 * Gradients, with u_ijf = Whi_if + Whj_jf, L = LeakyReLU, alpha_ij = p_ij / Z_i, Gp = dL/dhp:
 *   de_ij = alpha_ij (Gp_i . Whi_j - Gp_i . hp_i)                     (softmax Jacobian; the max shift cancels)
 *   da_f  = sum_ij de_ij L(u_ijf),     q_ijf = de_ij a_f L'(u_ijf)
 *   dWhi_i = sum_j q_ij (row sums) + sum_k alpha_ki Gp_k (column sums: Whi_i is aggregated by the rows k that see i)
 *   dWhj_j = sum_i q_ij (column sums)
 *   dW[:Fin] = X^T dWhi, dW[Fin:] = X^T dWhj, dX = dWhi W[:Fin]^T + dWhj W[Fin:]^T
 */
int FN(gat_oracle_level_v2)(int64_t N, int64_t E, const int32_t* rowptr, const int32_t* col,
                        const int32_t* rowptr_t, const int32_t* col_t, const int32_t* perm_t,
                        int Fin, int H, int F, REAL alpha, int concat,
                        const REAL* X, const REAL* W, const REAL* a, const REAL* G,
                        REAL* out, REAL* dW, REAL* da, REAL* dX) {
  REAL* Whi = (REAL*)malloc((size_t)N * F * sizeof(REAL));
  REAL* Whj = (REAL*)malloc((size_t)N * F * sizeof(REAL));
  REAL* hp = (REAL*)malloc((size_t)N * F * sizeof(REAL));
  REAL* Gp = (REAL*)malloc((size_t)N * F * sizeof(REAL));
  REAL* al = (REAL*)malloc((size_t)E * sizeof(REAL));
  REAL* de = (REAL*)malloc((size_t)E * sizeof(REAL));
  REAL* dWhi = (REAL*)malloc((size_t)N * F * sizeof(REAL));
  REAL* dWhj = (REAL*)malloc((size_t)N * F * sizeof(REAL));
  const int nt = FN(gat_oracle_threads)();
  double* dacc = (double*)malloc((size_t)nt * F * sizeof(double));
  if (!Whi || !Whj || !hp || !Gp || !al || !de || !dWhi || !dWhj || !dacc) return -1;
  const int OC = concat ? H * F : F;
  if (!concat) memset(out, 0, (size_t)N * F * sizeof(REAL));
  if (dX) memset(dX, 0, (size_t)N * Fin * sizeof(REAL));

  for (int h = 0; h < H; ++h) {
    const REAL* Wi = W + (size_t)h * 2 * Fin * F;
    const REAL* Wj = Wi + (size_t)Fin * F;
    const REAL* av = a + (size_t)h * F;
    gemm_nn(N, F, Fin, X, Wi, Whi);
    gemm_nn(N, F, Fin, X, Wj, Whj);
    memset(dacc, 0, (size_t)nt * F * sizeof(double));
    /* row pass: forward, de, the row sums of q and da */
#pragma omp parallel
    {
#ifdef _OPENMP
      double* dap = dacc + (size_t)omp_get_thread_num() * F;
#else
      double* dap = dacc;
#endif
#pragma omp for schedule(dynamic, 256)
      for (int64_t i = 0; i < N; ++i) {
        const int b = rowptr[i], e = rowptr[i + 1];
        const REAL* wi = Whi + i * F;
        REAL m = -INFINITY;
        for (int k = b; k < e; ++k) {
          const REAL* wj = Whj + (int64_t)col[k] * F;
          REAL ev = (REAL)0;
          for (int f = 0; f < F; ++f) {
            const REAL u = wi[f] + wj[f];
            ev += av[f] * (u > (REAL)0 ? u : alpha * u);
          }
          al[k] = ev;
          if (ev > m) m = ev;
        }
        double Zd = 0.0;
        for (int k = b; k < e; ++k) { al[k] = EXPR(al[k] - m); Zd += al[k]; }
        const REAL Z = (REAL)Zd;
        REAL* hr = hp + i * F;
        for (int f = 0; f < F; ++f) {
          double acc = 0.0;
          for (int k = b; k < e; ++k) acc += (double)al[k] * Whi[(int64_t)col[k] * F + f];
          hr[f] = (REAL)acc;
        }
        double D = 0.0;
        for (int f = 0; f < F; ++f) {
          hr[f] /= Z;
          REAL g;
          if (concat) {
            out[i * OC + h * F + f] = hr[f] > (REAL)0 ? hr[f] : EXPM1R(hr[f]);
            g = G[i * OC + h * F + f] * (hr[f] > (REAL)0 ? (REAL)1 : EXPR(hr[f]));
          } else {
            out[i * OC + f] += hr[f] / (REAL)H;
            g = G[i * OC + f] / (REAL)H;
          }
          Gp[i * F + f] = g;
          D += (double)g * hr[f];
        }
        REAL* dri = dWhi + i * F;
        for (int f = 0; f < F; ++f) dri[f] = (REAL)0;
        for (int k = b; k < e; ++k) {
          const REAL* whi_j = Whi + (int64_t)col[k] * F;
          const REAL* whj_j = Whj + (int64_t)col[k] * F;
          REAL dp = (REAL)0;
          for (int f = 0; f < F; ++f) dp += Gp[i * F + f] * whi_j[f];
          al[k] /= Z;
          const REAL d = al[k] * (dp - (REAL)D);
          de[k] = d;
          for (int f = 0; f < F; ++f) {
            const REAL u = wi[f] + whj_j[f];
            dap[f] += (double)d * (u > (REAL)0 ? u : alpha * u);
            dri[f] += d * av[f] * (u > (REAL)0 ? (REAL)1 : alpha);
          }
        }
      }
    }
    for (int f = 0; f < F; ++f) {
      double x = 0;
      for (int t = 0; t < nt; ++t) x += dacc[(size_t)t * F + f];
      da[(size_t)h * F + f] = (REAL)x;
    }
    /* column pass over the transposed pattern: node j as the gathered node of the rows i = col_t[k] */
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t j = 0; j < N; ++j) {
      const REAL* wj = Whj + j * F;
      for (int f = 0; f < F; ++f) {
        double qs = 0.0, ag = 0.0;
        for (int k = rowptr_t[j]; k < rowptr_t[j + 1]; ++k) {
          const int64_t i = col_t[k];
          const int32_t fe = perm_t[k];
          const REAL u = Whi[i * F + f] + wj[f];
          qs += (double)de[fe] * av[f] * (u > (REAL)0 ? (REAL)1 : alpha);
          ag += (double)al[fe] * Gp[i * F + f];
        }
        dWhj[j * F + f] = (REAL)qs;
        dWhi[j * F + f] += (REAL)ag;
      }
    }
    gemm_tn(N, F, Fin, X, dWhi, dW + (size_t)h * 2 * Fin * F);
    gemm_tn(N, F, Fin, X, dWhj, dW + (size_t)h * 2 * Fin * F + (size_t)Fin * F);
    if (dX) {
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < N; ++i)
        for (int k = 0; k < Fin; ++k) {
          REAL acc = (REAL)0;
          for (int f = 0; f < F; ++f) acc += dWhi[i * F + f] * Wi[(size_t)k * F + f] + dWhj[i * F + f] * Wj[(size_t)k * F + f];
          dX[i * Fin + k] += acc;
        }
    }
  }
  free(Whi); free(Whj); free(hp); free(Gp); free(al); free(de); free(dWhi); free(dWhj); free(dacc);
  return 0;
}
