// K5 helpers -- attention-vector gradient and parameter (un)packing (gfx950).
//
// da = autograd of the two a-halves matmuls `Wh a[:F']`, `Wh a[F':]` (reference layers.py:60-61 /
// `a.mm(edge_h)` layers.py:144); pack/unpack move between the reference's per-head parameters
// (W [Fin,F'], a [2F'], skip_projection [Fin,F'], layers.py:21-28,111-119) and the padded,
// head-interleaved operand of the fused projection.
#include "attn_common.h"

namespace pygat {

// ------------------------------------------------------------------- da reduction
// da_src[h][f] = sum_i ds[i][h] Wh[i][h*Fp+f], da_dst likewise with dt (layers.py:60-61 autograd).
constexpr int AG_BLOCKS = 1024;

// With dWh != nullptr the same stream also finishes dWh_i += ds_i a_src -- the term a column pass that ran
// before the row sums were known (pygat_gat_backward_col with dz_t) had to leave out.
__global__ __launch_bounds__(256) void a_grad_partial_kernel(int n, int nblocks, RowShape rs, const float* __restrict__ Wh,
                                                             const float* __restrict__ ds,
                                                             const float* __restrict__ dt,
                                                             float* __restrict__ ws, const float* __restrict__ a_pad,
                                                             float* __restrict__ dWh, const float* __restrict__ dmask) {
  // dWh != nullptr: the rows of dWh are rewritten in the same pass -- finished (+= ds_i a_src, when a_pad is given) and / or
  // taken back through the Wh dropout (x dmask: layers.py:37,136's mask, pre-scaled), which used to be a launch of its own
  // TPR threads per row (power of two >= NCH, <= 256), 256/TPR rows in flight per block
  int tpr = 1;
  while (tpr < rs.NCH) tpr <<= 1;
  const int rpb = 256 / tpr;
  const int c = threadIdx.x % tpr, rg = threadIdx.x / tpr;
  const bool valid = c < rs.NCH;
  const int co = valid ? 4 * c : 0, h = co >> rs.fp_shift;
  const int64_t rows_per_block = cdiv(n, nblocks);
  const int64_t r0 = blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < n) ? r0 + rows_per_block : n;
  float4 as = make_float4(0.f, 0.f, 0.f, 0.f), ad = as;
  float4 asrc = as;     // (zero without a_pad: the finishing term vanishes)
  if (dWh && a_pad) asrc = ld4(a_pad + (int64_t)h * 2 * rs.Fp + (co & (rs.Fp - 1)));
  int64_t i = r0 + rg;
  for (; i + 3 * rpb < r1; i += 4 * rpb) {  // 4 rows in flight per thread: the kernel is a pure stream
    float4 w[4], d[4];
    float a1[4], a2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t r = i + q * rpb;
      w[q] = ld4(Wh + r * rs.ldr + co);
      a1[q] = ds[r * rs.ldh + h];
      a2[q] = dt[r * rs.ldh + h];
      if (dWh) d[q] = ld4(dWh + r * rs.ldr + co);
    }
    if (dWh && valid) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        d[q].x = fmaf(a1[q], asrc.x, d[q].x); d[q].y = fmaf(a1[q], asrc.y, d[q].y);
        d[q].z = fmaf(a1[q], asrc.z, d[q].z); d[q].w = fmaf(a1[q], asrc.w, d[q].w);
        if (dmask) {
          const float4 mk = ld4(dmask + (i + q * rpb) * rs.ldr + co);
          d[q].x *= mk.x; d[q].y *= mk.y; d[q].z *= mk.z; d[q].w *= mk.w;
        }
        st4(dWh + (i + q * rpb) * rs.ldr + co, d[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      as.x = fmaf(a1[q], w[q].x, as.x); as.y = fmaf(a1[q], w[q].y, as.y); as.z = fmaf(a1[q], w[q].z, as.z); as.w = fmaf(a1[q], w[q].w, as.w);
      ad.x = fmaf(a2[q], w[q].x, ad.x); ad.y = fmaf(a2[q], w[q].y, ad.y); ad.z = fmaf(a2[q], w[q].z, ad.z); ad.w = fmaf(a2[q], w[q].w, ad.w);
    }
  }
  for (; i < r1; i += rpb) {
    const float4 w = ld4(Wh + i * rs.ldr + co);
    const float a1 = ds[i * rs.ldh + h], a2 = dt[i * rs.ldh + h];
    if (dWh && valid) {
      float4 d = ld4(dWh + i * rs.ldr + co);
      d.x = fmaf(a1, asrc.x, d.x); d.y = fmaf(a1, asrc.y, d.y); d.z = fmaf(a1, asrc.z, d.z); d.w = fmaf(a1, asrc.w, d.w);
      if (dmask) {
        const float4 mk = ld4(dmask + i * rs.ldr + co);
        d.x *= mk.x; d.y *= mk.y; d.z *= mk.z; d.w *= mk.w;
      }
      st4(dWh + i * rs.ldr + co, d);
    }
    as.x = fmaf(a1, w.x, as.x); as.y = fmaf(a1, w.y, as.y); as.z = fmaf(a1, w.z, as.z); as.w = fmaf(a1, w.w, as.w);
    ad.x = fmaf(a2, w.x, ad.x); ad.y = fmaf(a2, w.y, ad.y); ad.z = fmaf(a2, w.z, ad.z); ad.w = fmaf(a2, w.w, ad.w);
  }
  __shared__ float4 sm[2][256];
  sm[0][threadIdx.x] = as;
  sm[1][threadIdx.x] = ad;
  __syncthreads();
  if (rg == 0 && valid) {
    for (int g = 1; g < rpb; ++g) {  // fixed order: deterministic
      float4 q = sm[0][g * tpr + c], r = sm[1][g * tpr + c];
      as.x += q.x; as.y += q.y; as.z += q.z; as.w += q.w;
      ad.x += r.x; ad.y += r.y; ad.z += r.z; ad.w += r.w;
    }
    float* o = ws + (int64_t)blockIdx.x * 2 * rs.R;
    st4(o + co, as);
    st4(o + rs.R + co, ad);
  }
}

// 8 lanes share one output: lane q sums slabs q, q+8, ... (8 loads in flight each), then a fixed butterfly
__global__ __launch_bounds__(256) void a_grad_final_kernel(RowShape rs, int nblocks, const float* __restrict__ ws,
                                                           float* __restrict__ da) {
  const int idx = (blockIdx.x * 256 + threadIdx.x) >> 3;  // over H * 2 * Fo
  const int q = threadIdx.x & 7;
  const bool ok = idx < rs.H * 2 * rs.Fo;
  const int ii = ok ? idx : 0;
  const int h = ii / (2 * rs.Fo), r = ii % (2 * rs.Fo);
  const int which = r / rs.Fo, f = r % rs.Fo;
  const float* p = ws + which * rs.R + h * rs.Fp + f;
  const int64_t st = 2 * (int64_t)rs.R;
  float acc = 0.f;
  int b = q;
  for (; b + 56 < nblocks; b += 64) {
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = p[(b + 8 * u) * st];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += x[u];
  }
  for (; b < nblocks; b += 8) acc += p[b * st];
  acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
  if (ok && q == 0) da[idx] = acc;
}

// da when the column pass took the sums along (pygat_gat_backward_col with da_part): its per-work-group records [2 R] and the
// rows it left to its fix-up launch (the cut-row list: their ds / dt / Wh rows are read here, a few ten thousand rows) are
// folded into gridDim.x slabs in a fixed order; a_grad_final_kernel adds the slabs.  TPR threads (power of two >= R / 2)
// own the R / 2 float4s of a record, 256 / TPR records or rows are in flight per round.
__global__ __launch_bounds__(256) void a_grad_fold_kernel(RowShape rs, int64_t nrec, const float* __restrict__ da_part, int n_cut,
                                                          const int32_t* __restrict__ cut, const float* __restrict__ Wh,
                                                          const float* __restrict__ ds, const float* __restrict__ dt,
                                                          float* __restrict__ ws) {
  const int q4n = rs.R / 2;   // float4s per record
  int tpr = 1;
  while (tpr < q4n) tpr <<= 1;
  const int nsub = 256 / tpr, q4 = threadIdx.x % tpr, sub = threadIdx.x / tpr;
  const bool valid = q4 < q4n;
  const int qq = valid ? q4 : 0;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int64_t stride = (int64_t)gridDim.x * nsub;
  int64_t r = (int64_t)blockIdx.x * nsub + sub;
  for (; r + 3 * stride < nrec; r += 4 * stride) {
    float4 x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = ld4(da_part + (r + u * stride) * 2 * rs.R + 4 * qq);
#pragma unroll
    for (int u = 0; u < 4; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
  }
  for (; r < nrec; r += stride) {
    const float4 x = ld4(da_part + r * 2 * rs.R + 4 * qq);
    acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
  }
  const int which = qq >= rs.R / 4, col = 4 * (qq - which * (rs.R / 4)), h = col >> rs.fp_shift;
  const float* sc = which ? dt : ds;
  for (int64_t c = (int64_t)blockIdx.x * nsub + sub; c < n_cut; c += stride) {
    const int64_t row = cut[3 * c + 1];
    const float4 w = ld4(Wh + row * rs.ldr + col);
    const float v = sc[row * rs.ldh + h];
    acc.x = fmaf(v, w.x, acc.x); acc.y = fmaf(v, w.y, acc.y); acc.z = fmaf(v, w.z, acc.z); acc.w = fmaf(v, w.w, acc.w);
  }
  __shared__ float4 sm[256];
  sm[threadIdx.x] = acc;
  __syncthreads();
  if (sub == 0 && valid) {
    for (int g = 1; g < nsub; ++g) {   // fixed order: deterministic
      const float4 x = sm[g * tpr + q4];
      acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
    }
    st4(ws + (int64_t)blockIdx.x * 2 * rs.R + 4 * q4, acc);
  }
}

// s, t from a (masked) Wh table: one thread per (node, head).  wh_mask != nullptr: the Wh dropout (layers.py:37,136) is applied
// here, in place, instead of by a launch of its own
__global__ __launch_bounds__(256) void attn_scores_kernel(int n, int H, int Fp, float* __restrict__ Wh,
                                                          const float* __restrict__ wh_mask, const float* __restrict__ a_pad,
                                                          float* __restrict__ s, float* __restrict__ t) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)n * H) return;
  const int h = (int)(idx % H);
  float* w = Wh + idx * Fp;               // row i, head h: offset (i*H + h)*Fp
  const float* as = a_pad + (int64_t)h * 2 * Fp;
  float x = 0.f, y = 0.f;
  for (int f = 0; f < Fp; f += 4) {
    float4 w4 = ld4(w + f);
    if (wh_mask) {
      const float4 mk = ld4(wh_mask + idx * Fp + f);
      w4.x *= mk.x; w4.y *= mk.y; w4.z *= mk.z; w4.w *= mk.w;
      st4(w + f, w4);
    }
    const float4 a4 = ld4(as + f), d4 = ld4(as + Fp + f);
    x += dot4(w4, a4);
    y += dot4(w4, d4);
  }
  s[idx] = x;
  if (t) t[idx] = y;
}

// ------------------------------------------------------------- parameter packing
// Blocks [0, copy_blocks): one element of Wcat (or of a_pad) per thread -- the padded head columns of W and W_skip.
// Blocks behind them: one WAVE per (k, h) forms the two columns (W_h a_src_h)[k], (W_h a_dst_h)[k]: coalesced reads of the
// F' weights of row k, two DPP sums.  (The first version let one thread loop over F' for each of those 2 H columns: 8
// lanes per row walking 256 strided floats each -- 25 us for the PPI levels, a fifth of their projection GEMM.)
// TABLE: the parameters of head h come from their own tensors (pygat_pack_params_heads: pointers as kernel arguments), else
// from the stacked arrays W [H x Fin x F'], a [H x 2F'], w_skip [H x Fin x F']
constexpr int MAX_HEADS_TABLE = PYGAT_MAX_HEADS_TABLE;
struct HeadPtrs {
  const float* w[MAX_HEADS_TABLE];
  const float* a[MAX_HEADS_TABLE];
  const float* s[MAX_HEADS_TABLE];
};
template <bool TABLE>
__global__ __launch_bounds__(256) void pack_params_kernel(int H, int Fin, int Fo, int Fp,
                                                          const float* __restrict__ W_,
                                                          const float* __restrict__ a_,
                                                          const float* __restrict__ w_skip_, HeadPtrs tab, int has_skip,
                                                          float* __restrict__ Wcat, int64_t ldw,
                                                          float* __restrict__ a_pad, int copy_blocks) {
  const int R = H * Fp, Rs = has_skip ? R : 0;
  auto Wp = [&](int h) { return TABLE ? tab.w[h] : W_ + (int64_t)h * Fin * Fo; };        // [Fin x F'] of head h
  auto Ap = [&](int h) { return TABLE ? tab.a[h] : a_ + (int64_t)h * 2 * Fo; };          // [2F']
  auto Sp = [&](int h) { return TABLE ? tab.s[h] : w_skip_ + (int64_t)h * Fin * Fo; };
  if ((int)blockIdx.x >= copy_blocks) {
    const int64_t wv = ((int64_t)blockIdx.x - copy_blocks) * 4 + (threadIdx.x >> 6);
    if (wv >= (int64_t)Fin * H) return;
    const int k = (int)(wv / H), h = (int)(wv % H), lane = threadIdx.x & 63;
    const float* wr = Wp(h) + (int64_t)k * Fo;
    const float* ar = Ap(h);
    float vs = 0.f, vd = 0.f;
    for (int f = lane; f < Fo; f += 64) {
      const float w = wr[f];
      vs = fmaf(w, ar[f], vs);
      vd = fmaf(w, ar[Fo + f], vd);
    }
    vs = group_sum<64>(vs);
    vd = group_sum<64>(vd);
    if (lane == 0) {
      float* o = Wcat + (int64_t)k * ldw + R + Rs;
      o[h] = vs;
      o[H + h] = vd;
    }
    return;
  }
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < (int64_t)H * 2 * Fp) {  // a_pad[h][which][f]
    const int h = (int)(idx / (2 * Fp)), r = (int)(idx % (2 * Fp)), which = r / Fp, f = r % Fp;
    a_pad[idx] = f < Fo ? Ap(h)[which * Fo + f] : 0.f;
  }
  if (idx >= (int64_t)Fin * ldw) return;
  const int k = (int)(idx / ldw), col = (int)(idx % ldw);
  float v = 0.f;
  if (col < R) {
    const int h = col / Fp, f = col % Fp;
    if (f < Fo) v = Wp(h)[(int64_t)k * Fo + f];
  } else if (col < R + Rs) {
    const int h = (col - R) / Fp, f = (col - R) % Fp;
    if (f < Fo) v = Sp(h)[(int64_t)k * Fo + f];
  } else if (col < R + Rs + 2 * H) {
    return;                         // the W_h a columns: written by the waves of the second block range
  }
  Wcat[idx] = v;
}

__global__ __launch_bounds__(256) void unpack_wgrad_kernel(int H, int Fin, int Fo, int Fp,
                                                           const float* __restrict__ dWcat, int64_t ld,
                                                           int col_offset, float* __restrict__ dW) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)H * Fin * Fo) return;
  const int f = (int)(idx % Fo);
  const int k = (int)((idx / Fo) % Fin);
  const int h = (int)(idx / ((int64_t)Fo * Fin));
  dW[idx] = dWcat[(int64_t)k * ld + col_offset + h * Fp + f];
}

}  // namespace pygat

using namespace pygat;

extern "C" size_t pygat_agrad_workspace_bytes(int H, int Fo) {
  int Fp = padded_width(Fo);
  if (H <= 0 || Fp == 0) return 0;
  return (size_t)AG_BLOCKS * 2 * (size_t)(H * Fp) * sizeof(float);
}

extern "C" int pygat_a_grad(int n, int H, int Fo, const float* Wh, const float* ds, const float* dt, float* da,
                            void* ws, const float* a_pad, float* dWh, const float* dwh_mask, int h_first, int h_count,
                            void* stream) {
  RowShape rs;
  const int Fp = padded_width(Fo);
  HeadRange rg;
  PYGAT_REQUIRE(H > 0 && Fp > 0 && Fp <= 1024, "a_grad: unsupported H=%d F'=%d", H, Fo);
  PYGAT_REQUIRE(make_head_range(H, h_first, h_count, &rg), "a_grad: bad head range [%d, +%d) of %d", h_first, h_count, H);
  PYGAT_REQUIRE(n > 0 && Wh && ds && dt && da && ws && aligned16(Wh) && aligned16(ws), "a_grad: bad arguments");
  PYGAT_REQUIRE(!dWh || ((a_pad || dwh_mask) && aligned16(dWh) && (!a_pad || aligned16(a_pad)) && (!dwh_mask || aligned16(dwh_mask))),
                "a_grad: dWh needs a_pad and / or dwh_mask, 16-byte aligned");
  PYGAT_REQUIRE(!dwh_mask || dWh, "a_grad: dwh_mask needs dWh");
  hipStream_t st = (hipStream_t)stream;
  // a work-group streams rows of at most 1024 floats (256 threads x 16 B): wider levels go window by window
  const int hg = (rg.hr * Fp <= 1024) ? rg.hr : (1024 / Fp);
  // slabs: >= 64 rows per work-group on narrow rows (one round of four rows in flight per thread), but a work-group holds only
  // 256 / (threads per row) rows at a time -- ONE for rows of 1024 floats, where 256-row slabs meant 13 work-groups walking
  // the PPI batch (3144 x 4 KB) serially: 47 us for 13 MB.  At most 32 rounds of rows per work-group, at most AG_BLOCKS slabs.
  int tpr_h = 1;
  while (tpr_h < (hg * Fp) / 4) tpr_h <<= 1;
  const int64_t per_round = 256 / (tpr_h < 256 ? tpr_h : 256);
  int64_t nb = cdiv(n, 64);     // (256-row slabs left Cora with 11 work-groups of four dependent rounds each: 12 us for 0.7 MB)
  if (cdiv(n, 32 * per_round) > nb) nb = cdiv(n, 32 * per_round);
  int nblocks = nb > AG_BLOCKS ? AG_BLOCKS : (int)nb;
  for (int h0 = 0; h0 < rg.hr; h0 += hg) {
    const int hc = (rg.hr - h0 < hg) ? rg.hr - h0 : hg;
    const int gh = rg.hb + h0;
    PYGAT_REQUIRE(make_window_shape(H, Fo, hc, &rs), "a_grad: unsupported H=%d F'=%d", hc, Fo);
    hipLaunchKernelGGL(a_grad_partial_kernel, dim3(nblocks), dim3(256), 0, st, n, nblocks, rs, Wh + (int64_t)gh * Fp,
                       ds + gh, dt + gh, (float*)ws, a_pad ? a_pad + (int64_t)gh * 2 * Fp : nullptr,
                       dWh ? dWh + (int64_t)gh * Fp : nullptr, dwh_mask ? dwh_mask + (int64_t)gh * Fp : nullptr);
    hipLaunchKernelGGL(a_grad_final_kernel, dim3((unsigned)cdiv(rs.H * 2 * rs.Fo * 8, 256)), dim3(256), 0, st, rs, nblocks,
                       (const float*)ws, da + (int64_t)gh * 2 * Fo);
    PYGAT_CHECK_LAUNCH("a_grad");
  }
  return PYGAT_OK;
}

extern "C" int pygat_a_grad_fold(const pygat_graph* gT, int H, int Fo, const float* Wh, const float* ds, const float* dt,
                                 const float* da_part, float* da, void* ws, int head_group, void* stream) {
  GraphDev g;
  int rc = check_graph(gT, &g, 2);
  if (rc) return rc;
  const int Fp = padded_width(Fo);
  PYGAT_REQUIRE(H > 0 && Fp > 0, "a_grad_fold: unsupported H=%d F'=%d", H, Fo);
  const int hg = head_group_arg(head_group, g.n, H, Fp);
  const int64_t nrec = hg > 0 ? col_da_blocks(g, H, Fp, hg, nullptr) : 0;
  PYGAT_REQUIRE(nrec > 0, "a_grad_fold: the column pass leaves no da records for this level (pygat_gat_backward_col_da_bytes() == 0)");
  PYGAT_REQUIRE(Wh && ds && dt && da_part && da && ws && aligned16(Wh) && aligned16(da_part) && aligned16(ws), "a_grad_fold: bad arguments");
  RowShape rs;
  PYGAT_REQUIRE(make_window_shape(H, Fo, H, &rs), "a_grad_fold: unsupported H=%d F'=%d", H, Fo);
  hipStream_t st = (hipStream_t)stream;
  int tpr = 1;
  while (tpr < rs.R / 2) tpr <<= 1;
  const int nsub = 256 / tpr;
  int64_t nb = cdiv(nrec + g.n_cut, (int64_t)nsub * 8);
  const int nblocks = nb > AG_BLOCKS ? AG_BLOCKS : (nb < 1 ? 1 : (int)nb);
  hipLaunchKernelGGL(a_grad_fold_kernel, dim3(nblocks), dim3(256), 0, st, rs, nrec, da_part, g.n_cut, g.cut, Wh, ds, dt, (float*)ws);
  hipLaunchKernelGGL(a_grad_final_kernel, dim3((unsigned)cdiv(rs.H * 2 * rs.Fo * 8, 256)), dim3(256), 0, st, rs, nblocks,
                     (const float*)ws, da);
  PYGAT_CHECK_LAUNCH("a_grad_fold");
  return PYGAT_OK;
}

extern "C" int pygat_attn_scores(int n, int H, int Fo, float* Wh, const float* wh_mask, const float* a_pad, float* s, float* t,
                                 void* stream) {
  int Fp = padded_width(Fo);
  PYGAT_REQUIRE(n > 0 && H > 0 && Fp > 0 && Wh && a_pad && s && aligned16(Wh) && aligned16(a_pad) && (!wh_mask || aligned16(wh_mask)),
                "attn_scores: bad arguments");
  hipLaunchKernelGGL(attn_scores_kernel, dim3((unsigned)cdiv((int64_t)n * H, 256)), dim3(256), 0, (hipStream_t)stream, n,
                     H, Fp, Wh, wh_mask, a_pad, s, t);
  PYGAT_CHECK_LAUNCH("attn_scores");
  return PYGAT_OK;
}

extern "C" int pygat_pack_params(int H, int Fin, int Fo, const float* W, const float* a, const float* w_skip,
                                 float* Wcat, int64_t ldw, float* a_pad, void* stream) {
  int Fp = padded_width(Fo);
  PYGAT_REQUIRE(H > 0 && Fin > 0 && Fp > 0 && W && a && Wcat && a_pad, "pack_params: bad arguments");
  const int need = H * Fp * (w_skip ? 2 : 1) + 2 * H;
  PYGAT_REQUIRE(ldw >= need && ldw % 4 == 0, "pack_params: ldw=%lld must be a multiple of 4 and >= %d", (long long)ldw, need);
  int64_t tot = (int64_t)Fin * ldw;
  if (tot < (int64_t)H * 2 * Fp) tot = (int64_t)H * 2 * Fp;
  const int copy_blocks = (int)cdiv(tot, 256);
  HeadPtrs none = {};
  hipLaunchKernelGGL(pack_params_kernel<false>, dim3((unsigned)(copy_blocks + cdiv((int64_t)Fin * H, 4))), dim3(256), 0,
                     (hipStream_t)stream, H, Fin, Fo, Fp, W, a, w_skip, none, w_skip ? 1 : 0, Wcat, ldw, a_pad, copy_blocks);
  PYGAT_CHECK_LAUNCH("pack_params");
  return PYGAT_OK;
}

// The per-head parameter tensors of a level copied into stacked arrays W [H x nW], a [H x nA], w_skip [H x nW] in ONE
// launch (torch.stack: one cat launch per parameter kind).  For the level flavours that still take stacked parameters.
namespace pygat {
__global__ __launch_bounds__(256) void stack_heads_kernel(int H, int64_t nW, int nA, int64_t nS, int64_t nWo, int64_t nSo, HeadPtrs tab,
                                                          float* __restrict__ W, float* __restrict__ a, float* __restrict__ S) {
  // output blocks of nWo >= nW (nSo >= nS) elements per head: the tail of a block is zero (rows appended to [Fin, F'])
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per = nWo + nA + nSo;
  if (idx >= per * H) return;
  const int h = (int)(idx / per);
  const int64_t r = idx % per;
  if (r < nWo) W[h * nWo + r] = r < nW ? tab.w[h][r] : 0.f;
  else if (r < nWo + nA) a[(int64_t)h * nA + (r - nWo)] = tab.a[h][r - nWo];
  else S[h * nSo + (r - nWo - nA)] = (r - nWo - nA) < nS ? tab.s[h][r - nWo - nA] : 0.f;
}
}  // namespace pygat

extern "C" int pygat_stack_heads_padded(int H, int64_t nW, int64_t nW_out, int nA, int64_t nS, int64_t nS_out, const float* const* W,
                                        const float* const* a, const float* const* w_skip, float* W_out, float* a_out, float* skip_out,
                                        void* stream) {
  if (!w_skip) nS = nS_out = 0;
  PYGAT_REQUIRE(H > 0 && H <= MAX_HEADS_TABLE && nW > 0 && nW_out >= nW && nA > 0 && nS >= 0 && nS_out >= nS && W && a && W_out &&
                    a_out && (!w_skip || (skip_out && nS > 0)),
                "stack_heads: bad arguments (1 <= H <= %d)", MAX_HEADS_TABLE);
  HeadPtrs tab = {};
  for (int h = 0; h < H; ++h) {
    PYGAT_REQUIRE(W[h] && a[h] && (!w_skip || w_skip[h]), "stack_heads: null parameter pointer of head %d", h);
    tab.w[h] = W[h]; tab.a[h] = a[h]; tab.s[h] = w_skip ? w_skip[h] : nullptr;
  }
  const int64_t tot = (nW_out + nA + nS_out) * H;
  hipLaunchKernelGGL(stack_heads_kernel, dim3((unsigned)cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, H, nW, nA, nS, nW_out,
                     nS_out, tab, W_out, a_out, skip_out);
  PYGAT_CHECK_LAUNCH("stack_heads");
  return PYGAT_OK;
}

extern "C" int pygat_stack_heads(int H, int64_t nW, int nA, int64_t nS, const float* const* W, const float* const* a,
                                 const float* const* w_skip, float* W_out, float* a_out, float* skip_out, void* stream) {
  return pygat_stack_heads_padded(H, nW, nW, nA, nS, nS, W, a, w_skip, W_out, a_out, skip_out, stream);
}

extern "C" int pygat_pack_params_heads(int H, int Fin, int Fo, const float* const* W, const float* const* a,
                                       const float* const* w_skip, float* Wcat, int64_t ldw, float* a_pad, void* stream) {
  int Fp = padded_width(Fo);
  PYGAT_REQUIRE(H > 0 && H <= MAX_HEADS_TABLE && Fin > 0 && Fp > 0 && W && a && Wcat && a_pad,
                "pack_params_heads: bad arguments (1 <= H <= %d)", MAX_HEADS_TABLE);
  const int need = H * Fp * (w_skip ? 2 : 1) + 2 * H;
  PYGAT_REQUIRE(ldw >= need && ldw % 4 == 0, "pack_params_heads: ldw=%lld must be a multiple of 4 and >= %d", (long long)ldw, need);
  HeadPtrs tab = {};
  for (int h = 0; h < H; ++h) {
    PYGAT_REQUIRE(W[h] && a[h] && (!w_skip || w_skip[h]), "pack_params_heads: null parameter pointer of head %d", h);
    tab.w[h] = W[h]; tab.a[h] = a[h]; tab.s[h] = w_skip ? w_skip[h] : nullptr;
  }
  int64_t tot = (int64_t)Fin * ldw;
  if (tot < (int64_t)H * 2 * Fp) tot = (int64_t)H * 2 * Fp;
  const int copy_blocks = (int)cdiv(tot, 256);
  hipLaunchKernelGGL(pack_params_kernel<true>, dim3((unsigned)(copy_blocks + cdiv((int64_t)Fin * H, 4))), dim3(256), 0,
                     (hipStream_t)stream, H, Fin, Fo, Fp, nullptr, nullptr, nullptr, tab, w_skip ? 1 : 0, Wcat, ldw, a_pad,
                     copy_blocks);
  PYGAT_CHECK_LAUNCH("pack_params_heads");
  return PYGAT_OK;
}

extern "C" int pygat_unpack_wgrad(int H, int Fin, int Fo, const float* dWcat, int64_t ld, int col_offset, float* dW,
                                  void* stream) {
  int Fp = padded_width(Fo);
  PYGAT_REQUIRE(H > 0 && Fin > 0 && Fp > 0 && dWcat && dW && col_offset >= 0 && ld >= col_offset + H * Fp,
                "unpack_wgrad: bad arguments");
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3((unsigned)cdiv((int64_t)H * Fin * Fo, 256)), dim3(256), 0,
                     (hipStream_t)stream, H, Fin, Fo, Fp, dWcat, ld, col_offset, dW);
  PYGAT_CHECK_LAUNCH("unpack_wgrad");
  return PYGAT_OK;
}
