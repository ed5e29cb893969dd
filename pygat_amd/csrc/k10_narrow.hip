// K10 -- a NARROW level under per-head input dropout, on the vector ALUs (gfx950).
//
// The second level of the reference's citation models (models.py:24-29 SpGAT.out_att behind F.dropout, layers.py:132-136)
// multiplies N x 64 inputs with H heads of 3-8 output features: Cora 2708 x 64 -> 1 x 7, Pubmed 19717 x 64 -> 8 x 3.  Every
// head drops its own copy of x (layers.py:34,132), so the product is  Wh_h = 1/(1-p) (x o m_h) W_h  with the decisions of all
// heads in one byte per input element (pygat_dropout_bits).  The head-masked MFMA kernels (k1_gemm.hip gemm_headmask_kernel)
// are built for Fin in the thousands; on a 64 x 32 weight table they spend 21-35 us per product staging 128-row tiles of
// which the MFMAs use a sliver, and the gradient into x went through one GEMM PER HEAD plus a folding pass (Pubmed: 8 x 9 us
// + 13 us).  Here the weight table lives in REGISTERS -- lane = output column, one register per input feature (or lane =
// input feature, one register per output column for the gradient into x) -- and a wave walks its rows; a row's x under each
// head's mask is staged once in the wave's LDS (project, wgrad), a row of [dWh | Gp] reaches the lanes as scalar operands
// (dx: v_readlane with constant lane numbers):
//   project  [Wh | Sk][i, c]  = scale sum_k bit_h(c)(i,k) x[i,k] Wcat[k, c]                      1 FMA per (row, k)
//   wgrad    dWc[k, c]        = scale sum_i bit_h(c)(i,k) x[i,k] [dWh | Gp][i, c]                wave partials, three-level sum
//   dx       dx[i, k]         = scale sum_h bit_h(i,k) sum_f [dWh | Gp][i, h, f] Wcat[k, h, f]   1 FMA per (row, column)
// Taken when Fin <= 128, H <= 8 and (skip ? 2 : 1) H Fp <= 128 (pygat_dropout_narrow); pygat_project_dropout and
// pygat_wgrad_dropout route here by themselves, pygat_dx_dropout is new.  fp32 FMA chains over Fin <= 128 terms (project, dx)
// or over the rows of a slab (wgrad: slabs of 4-32 rows, then 16 slabs at a time, a lane's chunks, 16 lanes -- no chain beyond 32).
#include "narrow.h"

namespace pygat {

struct NarrowArgs {
  int n, Fin, H, Fp, R, ntot, rpw;   // rpw: rows per wave
  const float* X;
  int64_t ldx;
  const unsigned char* bits;         // [n x Fin], bit h = head h keeps x[i,k]
  float scale;
};

__device__ __forceinline__ float and_mask(float w, int msk) { return __int_as_float(__float_as_int(w) & msk); }

template <int NK>
__device__ __forceinline__ void load_row(const NarrowArgs& a, int i, int lane, float (&xv)[NK / 64], int (&bv)[NK / 64]) {
#pragma unroll
  for (int j = 0; j < NK / 64; ++j) {
    const int k = lane + 64 * j;
    const bool in = k < a.Fin;
    xv[j] = in ? a.X[(int64_t)i * a.ldx + k] : 0.f;
    bv[j] = in ? (int)a.bits[(int64_t)i * a.Fin + k] : 0;
  }
}

// A row's x under every head's mask, staged in the wave's own LDS region sm[h][k] = bit_h(i,k) ? x[i,k] : 0 (lane = k writes
// H values; row stride NK + 4 floats: the eight heads' b128 reads fall on different banks).  The product loops then read
// their head's masked row as float4s, same address for all lanes of a head: 1 FMA + a quarter LDS read per (row, k) instead
// of two v_readlane, a bit extract, an AND and the FMA.  LDS operations of one wave execute in order, so the wave needs no
// barrier between its own writes and reads -- only the compiler must keep them in program order (wave_sync).
constexpr int NARROW_MAX_H = 8;
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <int NK>
__device__ __forceinline__ void stage_masked(float* __restrict__ sm, int H, int lane, const float (&xv)[NK / 64], const int (&bv)[NK / 64]) {
#pragma unroll
  for (int h = 0; h < NARROW_MAX_H; ++h)
    if (h < H) {
#pragma unroll
      for (int j = 0; j < NK / 64; ++j) sm[h * (NK + 4) + lane + 64 * j] = ((bv[j] >> h) & 1) ? xv[j] : 0.f;
    }
}

// lane = output column c of [Wh | Sk]; w[k] = Wcat[k, c]
template <int NK>
__global__ __launch_bounds__(256) void narrow_project_kernel(NarrowArgs a, const float* __restrict__ Wcat, int64_t ldw,
                                                             float* __restrict__ Wh, float* __restrict__ Sk) {
  __shared__ __attribute__((aligned(16))) float smem[4][NARROW_MAX_H * (NK + 4)];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);   // uniform: scalar row loop
  const int r0 = wave * a.rpw;
  if (r0 >= a.n) return;
  const int r1 = (r0 + a.rpw < a.n) ? r0 + a.rpw : a.n;
  const int c = blockIdx.y * 64 + lane, cv = c < a.ntot ? c : a.ntot - 1;
  const int hl = (cv < a.R ? cv : cv - a.R) >> ilog2_dev(a.Fp);
  float* sm = smem[wv];
  const float4* mine = reinterpret_cast<const float4*>(sm + hl * (NK + 4));
  float w[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) w[k] = k < a.Fin ? Wcat[(int64_t)k * ldw + cv] : 0.f;
  float xv[NK / 64], xn[NK / 64];
  int bv[NK / 64], bn[NK / 64];
  load_row<NK>(a, r0, lane, xv, bv);
  for (int i = r0; i < r1; ++i) {
    load_row<NK>(a, (i + 1 < r1) ? i + 1 : i, lane, xn, bn);       // next row in flight
    stage_masked<NK>(sm, a.H, lane, xv, bv);
    wave_sync();
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < NK / 4; ++q) {
      const float4 v = mine[q];
      acc = fmaf(v.x, w[4 * q], acc); acc = fmaf(v.y, w[4 * q + 1], acc);
      acc = fmaf(v.z, w[4 * q + 2], acc); acc = fmaf(v.w, w[4 * q + 3], acc);
    }
    wave_sync();
    if (c < a.ntot) {
      if (c < a.R) Wh[(int64_t)i * a.R + c] = acc * a.scale;
      else Sk[(int64_t)i * a.R + (c - a.R)] = acc * a.scale;
    }
#pragma unroll
    for (int j = 0; j < NK / 64; ++j) { xv[j] = xn[j]; bv[j] = bn[j]; }
  }
}

// lane = column c of D = [dWh | Gp]; acc[k] = this wave's rows of dWc[k, c]; partial sums to part[wave][Fin][ntot]
template <int NK>
__global__ __launch_bounds__(256) void narrow_wgrad_kernel(NarrowArgs a, const float* __restrict__ D1, const float* __restrict__ D2,
                                                           int64_t ld2, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float smem[4][NARROW_MAX_H * (NK + 4)];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);
  const int r0 = wave * a.rpw;
  if (r0 >= a.n) return;
  const int r1 = (r0 + a.rpw < a.n) ? r0 + a.rpw : a.n;
  const int c = blockIdx.y * 64 + lane, cv = c < a.ntot ? c : a.ntot - 1;
  const int hl = (cv < a.R ? cv : cv - a.R) >> ilog2_dev(a.Fp);
  float* sm = smem[wv];
  const float4* mine = reinterpret_cast<const float4*>(sm + hl * (NK + 4));
  float acc[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) acc[k] = 0.f;
  float xv[NK / 64], xn[NK / 64];
  int bv[NK / 64], bn[NK / 64];
  load_row<NK>(a, r0, lane, xv, bv);
  auto dval = [&](int i) { return cv < a.R ? D1[(int64_t)i * a.R + cv] : D2[(int64_t)i * ld2 + (cv - a.R)]; };
  float d = dval(r0);
  for (int i = r0; i < r1; ++i) {
    const int in = (i + 1 < r1) ? i + 1 : i;
    load_row<NK>(a, in, lane, xn, bn);
    const float dn = dval(in);
    stage_masked<NK>(sm, a.H, lane, xv, bv);
    wave_sync();
#pragma unroll
    for (int q = 0; q < NK / 4; ++q) {
      const float4 v = mine[q];
      acc[4 * q] = fmaf(v.x, d, acc[4 * q]); acc[4 * q + 1] = fmaf(v.y, d, acc[4 * q + 1]);
      acc[4 * q + 2] = fmaf(v.z, d, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(v.w, d, acc[4 * q + 3]);
    }
    wave_sync();
    d = dn;
#pragma unroll
    for (int j = 0; j < NK / 64; ++j) { xv[j] = xn[j]; bv[j] = bn[j]; }
  }
  if (c < a.ntot) {
    float* o = part + (int64_t)wave * a.Fin * a.ntot + c;
#pragma unroll
    for (int k = 0; k < NK; ++k)
      if (k < a.Fin) o[(int64_t)k * a.ntot] = acc[k];
  }
}

// dWc[e] = scale * sum over the slabs, in three fixed levels: 16 consecutive slabs (their 16 loads in flight together), a
// lane's chunks in order, then the 16 lanes of an element in order.  16 elements x 16 lanes per work-group.
__global__ __launch_bounds__(256) void narrow_wgrad_reduce_kernel(int total, int splits, const float* __restrict__ part, float scale,
                                                                  float* __restrict__ out) {
  __shared__ float sm[16][17];
  const int el = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el, ev = e < total ? e : total - 1;
  float t = 0.f;
  for (int s0 = g * 16; s0 < splits; s0 += 256) {
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = (s0 + q < splits) ? part[(int64_t)(s0 + q) * total + ev] : 0.f;
    float u = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) u += v[q];
    t += u;
  }
  sm[g][el] = t;
  __syncthreads();
  if (g == 0 && e < total) {
    float r = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) r += sm[q][el];
    out[e] = r * scale;
  }
}

// lane = input feature k; w[c] = Wcat[k, c] over the NT (padded) columns of [W | Wskip]; heads are blocks of FP columns.
// Row i of [dWh | Gp] is the same for every lane: its values arrive as scalar loads (uniform addresses), FMA operands straight
// from scalar registers.
template <int NT, int FPS>
__global__ __launch_bounds__(256) void narrow_dx_kernel(NarrowArgs a, const float* __restrict__ D1, const float* __restrict__ D2,
                                                        int64_t ld2, const float* __restrict__ Wcat, int64_t ldw,
                                                        float* __restrict__ dx, int64_t lddx, int accumulate) {
  constexpr int FP = 1 << FPS;
  // the work-group's 64 rows of Wcat (its k chunk) staged in LDS with coalesced loads, row stride ntot + 1 (lane k then reads
  // row k without bank conflicts): read straight from global memory, every lane walking its own row, the set-up was 2048
  // cache lines per wave -- 24 us of Pubmed's 19717 x 64 level against 5 for the rows themselves
  extern __shared__ float dx_sm[];
  {
    const int k0 = blockIdx.y * 64, nk = (a.Fin - k0 < 64) ? a.Fin - k0 : 64;
    for (int e = threadIdx.x; e < 64 * a.ntot; e += 256) {
      const int kk = e / a.ntot, c = e - kk * a.ntot;
      dx_sm[kk * (a.ntot + 1) + c] = Wcat[(int64_t)(k0 + (kk < nk ? kk : nk - 1)) * ldw + c];
    }
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int r0 = wave * a.rpw;
  if (r0 >= a.n) return;
  const int r1 = (r0 + a.rpw < a.n) ? r0 + a.rpw : a.n;
  const int k = blockIdx.y * 64 + lane, kv = k < a.Fin ? k : a.Fin - 1;
  const int nblk = a.ntot >> FPS;
  float w[NT];
#pragma unroll
  for (int c = 0; c < NT; ++c) w[c] = dx_sm[lane * (a.ntot + 1) + (c < a.ntot ? c : 0)];
#pragma unroll
  for (int c = 0; c < NT; ++c) w[c] = c < a.ntot ? w[c] : 0.f;
  int b = (int)a.bits[(int64_t)r0 * a.Fin + kv];
  for (int i = r0; i < r1; ++i) {
    const int bnx = (int)a.bits[(int64_t)((i + 1 < r1) ? i + 1 : i) * a.Fin + kv];
    float acc = 0.f;
#pragma unroll
    for (int blk = 0; blk < NT / FP; ++blk) {
      if (blk < nblk) {                                  // uniform
        const int hb = blk < a.H ? blk : blk - a.H;      // W block, then the skip block of the same head
        const float* __restrict__ src = blk < a.H ? D1 + (int64_t)i * a.R + blk * FP : D2 + (int64_t)i * ld2 + hb * FP;
        float t = 0.f;
#pragma unroll
        for (int f = 0; f < FP; ++f) t = fmaf(src[f], w[blk * FP + f], t);
        acc += ((b >> hb) & 1) ? t : 0.f;
      }
    }
    if (k < a.Fin) {
      float* o = dx + (int64_t)i * lddx + k;
      *o = accumulate ? *o + acc * a.scale : acc * a.scale;
    }
    b = bnx;
  }
}

static const int g_narrow_on = [] { const char* e = getenv("PYGAT_NARROW"); return (e && atoi(e) == 0) ? 0 : 1; }();

bool narrow_takes(int Fin, int H, int Fo, bool skip) {
  const int Fp = padded_width(Fo);
  return g_narrow_on && Fin >= 1 && Fin <= 128 && H >= 1 && H <= 8 && Fp > 0 && Fp <= 64 && H * Fp * (skip ? 2 : 1) <= 128;
}

static int rows_per_wave(int n) {
  int r = n / 4096;
  return r < 4 ? 4 : (r > 32 ? 32 : r);
}

static int dx_rows(int n) {
  return rows_per_wave(n);
}

static NarrowArgs narrow_args(int n, int Fin, int H, int Fo, bool skip, const float* X, int64_t ldx, const unsigned char* bits, float p,
                              int rpw) {
  NarrowArgs a;
  a.n = n; a.Fin = Fin; a.H = H; a.Fp = padded_width(Fo); a.R = H * a.Fp; a.ntot = a.R * (skip ? 2 : 1); a.rpw = rpw;
  a.X = X; a.ldx = ldx; a.bits = bits; a.scale = p < 1.f ? 1.f / (1.f - p) : 0.f;
  return a;
}

int narrow_project(int n, int Fin, int H, int Fo, const float* X, int64_t ldx, const unsigned char* bits, float p, const float* Wcat,
                   int64_t ldw, float* Wh, float* Sk, hipStream_t st) {
  const NarrowArgs a = narrow_args(n, Fin, H, Fo, Sk != nullptr, X, ldx, bits, p, rows_per_wave(n));
  const dim3 grid((unsigned)cdiv(cdiv(n, a.rpw), 4), (unsigned)cdiv(a.ntot, 64));
  if (Fin <= 64) hipLaunchKernelGGL((narrow_project_kernel<64>), grid, dim3(256), 0, st, a, Wcat, ldw, Wh, Sk);
  else hipLaunchKernelGGL((narrow_project_kernel<128>), grid, dim3(256), 0, st, a, Wcat, ldw, Wh, Sk);
  PYGAT_CHECK_LAUNCH("project_dropout(narrow)");
  return PYGAT_OK;
}

int narrow_wgrad(int n, int Fin, int H, int Fo, const float* X, int64_t ldx, const unsigned char* bits, float p, const float* dWh,
                 const float* Gp, int64_t ldgp, float* dWc, int split_k, void* ws, hipStream_t st) {
  if (split_k < 1) split_k = 1;
  const int rpw = (int)cdiv(n, split_k), splits = (int)cdiv(n, rpw);
  const NarrowArgs a = narrow_args(n, Fin, H, Fo, Gp != nullptr, X, ldx, bits, p, rpw);
  const dim3 grid((unsigned)cdiv(splits, 4), (unsigned)cdiv(a.ntot, 64));
  float* part = (float*)ws;
  if (Fin <= 64) hipLaunchKernelGGL((narrow_wgrad_kernel<64>), grid, dim3(256), 0, st, a, dWh, Gp, ldgp, part);
  else hipLaunchKernelGGL((narrow_wgrad_kernel<128>), grid, dim3(256), 0, st, a, dWh, Gp, ldgp, part);
  PYGAT_CHECK_LAUNCH("wgrad_dropout(narrow)");
  const int total = Fin * a.ntot;
  hipLaunchKernelGGL(narrow_wgrad_reduce_kernel, dim3((unsigned)cdiv(total, 16)), dim3(256), 0, st, total, splits, (const float*)part,
                     a.scale, dWc);
  PYGAT_CHECK_LAUNCH("wgrad_dropout(narrow reduce)");
  return PYGAT_OK;
}

}  // namespace pygat

using namespace pygat;

extern "C" int pygat_dropout_narrow(int Fin, int H, int Fo, int skip) { return narrow_takes(Fin, H, Fo, skip != 0) ? 1 : 0; }

#define PYGAT_DX_CASE(NT, FPS)                                                                                          \
  hipLaunchKernelGGL((narrow_dx_kernel<NT, FPS>), grid, dim3(256), lds, st, a, dWh, Gp, ldgp, Wcat, ldw, dx, lddx, accumulate)

extern "C" int pygat_dx_dropout(int n, int Fin, int H, int Fo, const float* dWh, const float* Gp, int64_t ldgp,
                                const unsigned char* bits, float p, const float* Wcat, int64_t ldw, float* dx, int64_t lddx,
                                int accumulate, void* stream) {
  PYGAT_REQUIRE(n > 0 && dWh && bits && Wcat && dx && lddx >= Fin, "dx_dropout: bad arguments");
  PYGAT_REQUIRE(narrow_takes(Fin, H, Fo, Gp != nullptr), "dx_dropout: unsupported Fin=%d H=%d F'=%d (pygat_dropout_narrow)", Fin, H, Fo);
  PYGAT_REQUIRE(p >= 0.f && p <= 1.f, "dx_dropout: p=%g outside [0,1]", (double)p);
  const NarrowArgs a = narrow_args(n, Fin, H, Fo, Gp != nullptr, nullptr, 0, bits, p, dx_rows(n));
  PYGAT_REQUIRE(ldw >= a.ntot && (!Gp || ldgp >= a.R), "dx_dropout: bad leading dimensions");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)cdiv(cdiv(n, a.rpw), 4), (unsigned)cdiv(Fin, 64));
  const size_t lds = (size_t)64 * (a.ntot + 1) * sizeof(float);
  int fps = 0;
  while ((1 << fps) < a.Fp) ++fps;
  const int nt = a.ntot <= 32 ? 32 : (a.ntot <= 64 ? 64 : 128);
  switch (nt * 8 + fps) {
    case 32 * 8 + 2: PYGAT_DX_CASE(32, 2); break;
    case 32 * 8 + 3: PYGAT_DX_CASE(32, 3); break;
    case 32 * 8 + 4: PYGAT_DX_CASE(32, 4); break;
    case 32 * 8 + 5: PYGAT_DX_CASE(32, 5); break;
    case 64 * 8 + 2: PYGAT_DX_CASE(64, 2); break;
    case 64 * 8 + 3: PYGAT_DX_CASE(64, 3); break;
    case 64 * 8 + 4: PYGAT_DX_CASE(64, 4); break;
    case 64 * 8 + 5: PYGAT_DX_CASE(64, 5); break;
    case 64 * 8 + 6: PYGAT_DX_CASE(64, 6); break;
    case 128 * 8 + 2: PYGAT_DX_CASE(128, 2); break;
    case 128 * 8 + 3: PYGAT_DX_CASE(128, 3); break;
    case 128 * 8 + 4: PYGAT_DX_CASE(128, 4); break;
    case 128 * 8 + 5: PYGAT_DX_CASE(128, 5); break;
    case 128 * 8 + 6: PYGAT_DX_CASE(128, 6); break;
    default: set_error("dx_dropout: no kernel for %d columns of F'p=%d", a.ntot, a.Fp); return PYGAT_EINVAL;
  }
  PYGAT_CHECK_LAUNCH("dx_dropout");
  return PYGAT_OK;
}
