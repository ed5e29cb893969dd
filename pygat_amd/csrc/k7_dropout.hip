// K7 -- train-mode dropout around the projection (gfx950).
//
// The reference drops out inside every head, and every head draws its own masks because models.GAT
// calls the heads one after another (models.py:32,34):
//     layers.py:34 / :132   h = F.dropout(h, p)            per-head mask on the [N, Fin] input
//     layers.py:37 / :136   Wh = F.dropout(Wh, p)          per-head mask on [N, F']
//     layers.py:43 / :153   attention = F.dropout(...)     per-edge mask (applied in K2/K3b/K4)
// A per-head input mask breaks the "one GEMM for all heads" projection.  Instead of H small GEMMs on H
// masked copies, the masked input is written once as ONE wide operand
//     A'[i, h*Fin + k] = x[i,k] * m_h[i,k]                 (dropout_expand_kernel)
// and multiplied with the block-diagonal stack of the head weights (pack_blockdiag_kernel), so the
// level-1 projection of Cora (8 heads, Fin 1433) is one MFMA GEMM with K = 11 464 instead of 8
// launch-bound ones; dW comes back from one A'^T dWh product (unpack_blockdiag_kernel keeps the
// diagonal blocks) and dX from dWh B'^T folded over the heads under the same masks
// (dropout_head_sum_kernel).
//
// Masks are either given (tests: parity with the oracle is defined on explicit masks, no other
// implementation can reproduce torch's RNG stream) or drawn in-kernel with Philox-4x32-10 from a
// seed that lives in DEVICE memory (a captured HIP graph replays with fresh masks when the seed tensor
// is refreshed by a graph-safe generator).  Decision for element (row, col): word (col & 3) of
// Philox(counter = (col >> 2, row, stream_id), key = seed) < keep * 2^32; kept values are scaled by
// 1/keep as F.dropout does.
#include "rng.h"

namespace pygat {

// flat pre-scaled mask: out[e] = keep ? 1/keep : 0, e < count (layers.py:37,43 style masks)
__global__ __launch_bounds__(256) void dropout_mask_kernel(int64_t count, DropRng g, float* __restrict__ out) {
  const int64_t q4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t e0 = q4 * 4;
  if (e0 >= count) return;
  const uint64_t seed = *g.seed;
  const uint4 w = draw4(g, seed, (uint32_t)(q4 >> 32), 0xFFFFFFFFu, (uint32_t)q4);
  float v[4] = {w.x < g.thresh ? g.scale : 0.f, w.y < g.thresh ? g.scale : 0.f, w.z < g.thresh ? g.scale : 0.f,
                w.w < g.thresh ? g.scale : 0.f};
  if (e0 + 3 < count) {
    st4(out + e0, make_float4(v[0], v[1], v[2], v[3]));
  } else {
    for (int q = 0; e0 + q < count; ++q) out[e0 + q] = v[q];
  }
}

// Input-dropout decisions of all heads as ONE byte per input element: bit h of bits[i, k] = "head h keeps x[i,k]"
// (rng.h draw_heads4: word (h & 3) of Philox(counter = (k, i, stream_id, h >> 2), key = seed) < keep * 2^32).  Consumed by the
// head-masked GEMMs (k1_gemm.hip) and by dropout_head_sum_bits_kernel: 1 byte instead of 4H bytes per element.
__global__ __launch_bounds__(256) void dropout_bits_kernel(int n, int Fin, int H, DropRng g, unsigned char* __restrict__ bits) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)n * Fin) return;
  const int i = (int)(idx / Fin), k = (int)(idx - (int64_t)i * Fin);
  const uint64_t seed = *g.seed;
  uint32_t b = 0;
  for (int hq = 0; 4 * hq < H; ++hq) b |= keep_nibble(g, draw_heads4(g, seed, (uint32_t)i, (uint32_t)k, (uint32_t)hq)) << (4 * hq);
  bits[idx] = (unsigned char)(b & ((1u << H) - 1u));
}

// dx[i,k] (+)= scale * sum_h bit_h[i,k] * dxe[i, h*Fin + k]  -- back through the per-head input dropout (bits form)
__global__ __launch_bounds__(256) void dropout_head_sum_bits_kernel(int n, int Fin, int H, const float* __restrict__ dxe,
                                                                    int64_t lde, const unsigned char* __restrict__ bits,
                                                                    float scale, float* __restrict__ dx, int64_t ldx,
                                                                    int accumulate) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)n * Fin) return;
  const int i = (int)(idx / Fin), k = (int)(idx - (int64_t)i * Fin);
  const uint32_t b = bits[idx];
  float acc = 0.f;
  for (int h = 0; h < H; ++h)
    if ((b >> h) & 1u) acc += dxe[(int64_t)i * lde + (int64_t)h * Fin + k];
  acc *= scale;
  float* o = dx + (int64_t)i * ldx + k;
  *o = accumulate ? *o + acc : acc;
}

// A'[i, h*Fin + k] = x[i,k] * m_h[i,k]; one work-group per (row, 1024-column chunk), thread = 4 columns
__global__ __launch_bounds__(256) void dropout_expand_kernel(int n, int Fin, int H, int nchunks,
                                                             const float* __restrict__ x, int64_t ldx,
                                                             const float* __restrict__ mask,  // [H][n][Fin] or null
                                                             DropRng g, float* __restrict__ out, int64_t ldo) {
  const int64_t b = blockIdx.x;
  const int i = (int)(b / nchunks);
  const int c0 = ((int)(b % nchunks) * 256 + threadIdx.x) * 4;
  const int HF = H * Fin;
  if (c0 >= HF) return;
  uint4 w = make_uint4(0, 0, 0, 0);
  if (!mask) w = draw4(g, *g.seed, (uint32_t)i, 0u, (uint32_t)(c0 >> 2));
  float v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = c0 + q;
    v[q] = 0.f;
    if (c < HF) {
      const int h = c / Fin, k = c - h * Fin;
      const float m = mask ? mask[((int64_t)h * n + i) * Fin + k] : (word_of(w, q) < g.thresh ? g.scale : 0.f);
      v[q] = x[(int64_t)i * ldx + k] * m;
    }
  }
  float* o = out + (int64_t)i * ldo + c0;
  if (c0 + 3 < HF && ((ldo & 3) == 0)) {
    st4(o, make_float4(v[0], v[1], v[2], v[3]));
  } else {
    for (int q = 0; q < 4 && c0 + q < HF; ++q) o[q] = v[q];
  }
}

// dx[i,k] (+)= sum_h m_h[i,k] * dxe[i, h*Fin + k]  -- back through the per-head input dropout
__global__ __launch_bounds__(256) void dropout_head_sum_kernel(int n, int Fin, int H, const float* __restrict__ dxe,
                                                               int64_t lde, const float* __restrict__ mask, DropRng g,
                                                               float* __restrict__ dx, int64_t ldx, int accumulate) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)n * Fin) return;
  const int i = (int)(idx / Fin), k = (int)(idx - (int64_t)i * Fin);
  const uint64_t seed = mask ? 0 : *g.seed;
  float acc = 0.f;
  for (int h = 0; h < H; ++h) {
    const int c = h * Fin + k;
    float m;
    if (mask) {
      m = mask[((int64_t)h * n + i) * Fin + k];
    } else {
      const uint4 w = draw4(g, seed, (uint32_t)i, 0u, (uint32_t)(c >> 2));
      m = word_of(w, c & 3) < g.thresh ? g.scale : 0.f;
    }
    acc = fmaf(m, dxe[(int64_t)i * lde + c], acc);
  }
  float* o = dx + (int64_t)i * ldx + k;
  *o = accumulate ? *o + acc : acc;
}

// B'[h*Fin + k, :] = [ 0 .. W_h[k,:] (padded to Fp) .. 0 | 0 .. Wskip_h[k,:] .. 0 ]
__global__ __launch_bounds__(256) void pack_blockdiag_kernel(int H, int Fin, int Fo, int Fp, const float* __restrict__ W,
                                                             const float* __restrict__ w_skip,
                                                             float* __restrict__ Bp, int64_t ldb) {
  const int R = H * Fp, ncol = R * (w_skip ? 2 : 1);
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)H * Fin * ncol) return;
  const int c = (int)(idx % ncol);
  const int64_t row = idx / ncol;
  const int h = (int)(row / Fin), k = (int)(row - (int64_t)h * Fin);
  const int cc = c < R ? c : c - R;
  const int hc = cc / Fp, f = cc - hc * Fp;
  float v = 0.f;
  if (hc == h && f < Fo) v = (c < R ? W : w_skip)[((int64_t)h * Fin + k) * Fo + f];
  Bp[row * ldb + c] = v;
}

// dW[h,k,f] = dB'[h*Fin + k, col_offset + h*Fp + f]   (the diagonal blocks of A'^T dWh)
__global__ __launch_bounds__(256) void unpack_blockdiag_kernel(int H, int Fin, int Fo, int Fp,
                                                               const float* __restrict__ dBp, int64_t ldb,
                                                               int col_offset, float* __restrict__ dW) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)H * Fin * Fo) return;
  const int f = (int)(idx % Fo);
  const int64_t row = idx / Fo;          // h*Fin + k
  const int h = (int)(row / Fin);
  dW[idx] = dBp[row * ldb + col_offset + h * Fp + f];
}

}  // namespace pygat

using namespace pygat;

extern "C" int pygat_dropout_mask(int64_t count, float p, const void* seed, uint32_t stream_id, float* out,
                                  void* stream) {
  DropRng g;
  PYGAT_REQUIRE(count > 0 && seed && out && aligned16(out), "dropout_mask: bad arguments");
  PYGAT_REQUIRE(make_rng(p, seed, stream_id, &g), "dropout_mask: p=%g outside [0,1]", (double)p);
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)cdiv(cdiv(count, 4), 256)), dim3(256), 0, (hipStream_t)stream,
                     count, g, out);
  PYGAT_CHECK_LAUNCH("dropout_mask");
  return PYGAT_OK;
}

// Two flat masks from one seed in ONE launch (the Wh mask and the attention mask of a level: streams 2 and 3 of its seed)
namespace pygat {
__global__ __launch_bounds__(256) void dropout_mask2_kernel(int64_t count1, DropRng g1, float* __restrict__ out1, int64_t count2,
                                                            DropRng g2, float* __restrict__ out2, int64_t q1) {
  int64_t q4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool second = q4 >= q1;
  if (second) q4 -= q1;
  const int64_t count = second ? count2 : count1;
  const DropRng& g = second ? g2 : g1;
  float* out = second ? out2 : out1;
  const int64_t e0 = q4 * 4;
  if (e0 >= count) return;
  const uint64_t seed = *g.seed;
  const uint4 w = draw4(g, seed, (uint32_t)(q4 >> 32), 0xFFFFFFFFu, (uint32_t)q4);
  float v[4] = {w.x < g.thresh ? g.scale : 0.f, w.y < g.thresh ? g.scale : 0.f, w.z < g.thresh ? g.scale : 0.f,
                w.w < g.thresh ? g.scale : 0.f};
  if (e0 + 3 < count) {
    st4(out + e0, make_float4(v[0], v[1], v[2], v[3]));
  } else {
    for (int q = 0; e0 + q < count; ++q) out[e0 + q] = v[q];
  }
}
}  // namespace pygat

extern "C" int pygat_dropout_mask2(float p, const void* seed, int64_t count1, uint32_t stream1, float* out1, int64_t count2,
                                   uint32_t stream2, float* out2, void* stream) {
  DropRng g1, g2;
  PYGAT_REQUIRE(count1 > 0 && count2 > 0 && seed && out1 && out2 && aligned16(out1) && aligned16(out2), "dropout_mask2: bad arguments");
  PYGAT_REQUIRE(make_rng(p, seed, stream1, &g1) && make_rng(p, seed, stream2, &g2), "dropout_mask2: p=%g outside [0,1]", (double)p);
  const int64_t q1 = cdiv(cdiv(count1, 4), 256) * 256, q2 = cdiv(count2, 4);
  hipLaunchKernelGGL(dropout_mask2_kernel, dim3((unsigned)cdiv(q1 + q2, 256)), dim3(256), 0, (hipStream_t)stream, count1, g1, out1,
                     count2, g2, out2, q1);
  PYGAT_CHECK_LAUNCH("dropout_mask2");
  return PYGAT_OK;
}

extern "C" int pygat_dropout_expand(int n, int Fin, int H, const float* x, int64_t ldx, const float* mask, float p,
                                    const void* seed, uint32_t stream_id, float* out, int64_t ldo, void* stream) {
  DropRng g;
  PYGAT_REQUIRE(n > 0 && Fin > 0 && H > 0 && x && out && ldx >= Fin && ldo >= (int64_t)H * Fin && (mask || seed),
                "dropout_expand: bad arguments");
  PYGAT_REQUIRE((int64_t)H * Fin < ((int64_t)1 << 30), "dropout_expand: H*Fin too large");
  PYGAT_REQUIRE(make_rng(p, seed, stream_id, &g), "dropout_expand: p=%g outside [0,1]", (double)p);
  PYGAT_REQUIRE(aligned16(out), "dropout_expand: out must be 16-byte aligned");
  const int nchunks = (int)cdiv((int64_t)H * Fin, 1024);
  PYGAT_REQUIRE((int64_t)n * nchunks < ((int64_t)1 << 31), "dropout_expand: grid too large");
  hipLaunchKernelGGL(dropout_expand_kernel, dim3((unsigned)((int64_t)n * nchunks)), dim3(256), 0, (hipStream_t)stream, n,
                     Fin, H, nchunks, x, ldx, mask, g, out, ldo);
  PYGAT_CHECK_LAUNCH("dropout_expand");
  return PYGAT_OK;
}

extern "C" int pygat_dropout_head_sum(int n, int Fin, int H, const float* dxe, int64_t lde, const float* mask, float p,
                                      const void* seed, uint32_t stream_id, float* dx, int64_t ldx, int accumulate,
                                      void* stream) {
  DropRng g;
  PYGAT_REQUIRE(n > 0 && Fin > 0 && H > 0 && dxe && dx && ldx >= Fin && lde >= (int64_t)H * Fin && (mask || seed),
                "dropout_head_sum: bad arguments");
  PYGAT_REQUIRE(make_rng(p, seed, stream_id, &g), "dropout_head_sum: p=%g outside [0,1]", (double)p);
  hipLaunchKernelGGL(dropout_head_sum_kernel, dim3((unsigned)cdiv((int64_t)n * Fin, 256)), dim3(256), 0,
                     (hipStream_t)stream, n, Fin, H, dxe, lde, mask, g, dx, ldx, accumulate);
  PYGAT_CHECK_LAUNCH("dropout_head_sum");
  return PYGAT_OK;
}

extern "C" int pygat_pack_blockdiag(int H, int Fin, int Fo, const float* W, const float* w_skip, float* Bp,
                                    int64_t ldb, void* stream) {
  const int Fp = padded_width(Fo);
  PYGAT_REQUIRE(H > 0 && Fin > 0 && Fp > 0 && W && Bp, "pack_blockdiag: bad arguments");
  const int64_t ncol = (int64_t)H * Fp * (w_skip ? 2 : 1);
  PYGAT_REQUIRE(ldb >= ncol, "pack_blockdiag: ldb=%lld < %lld columns", (long long)ldb, (long long)ncol);
  const int64_t tot = (int64_t)H * Fin * ncol;
  PYGAT_REQUIRE(cdiv(tot, 256) < ((int64_t)1 << 31), "pack_blockdiag: too large");
  hipLaunchKernelGGL(pack_blockdiag_kernel, dim3((unsigned)cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, H, Fin,
                     Fo, Fp, W, w_skip, Bp, ldb);
  PYGAT_CHECK_LAUNCH("pack_blockdiag");
  return PYGAT_OK;
}

extern "C" int pygat_unpack_blockdiag(int H, int Fin, int Fo, const float* dBp, int64_t ldb, int col_offset, float* dW,
                                      void* stream) {
  const int Fp = padded_width(Fo);
  PYGAT_REQUIRE(H > 0 && Fin > 0 && Fp > 0 && dBp && dW && col_offset >= 0 && ldb >= (int64_t)col_offset + (int64_t)H * Fp,
                "unpack_blockdiag: bad arguments");
  hipLaunchKernelGGL(unpack_blockdiag_kernel, dim3((unsigned)cdiv((int64_t)H * Fin * Fo, 256)), dim3(256), 0,
                     (hipStream_t)stream, H, Fin, Fo, Fp, dBp, ldb, col_offset, dW);
  PYGAT_CHECK_LAUNCH("unpack_blockdiag");
  return PYGAT_OK;
}


/* bits [n x Fin] bytes, bit h = head h keeps x[i,k] (H <= 8); drawn from (seed, stream_id) like the other masks. */
extern "C" int pygat_dropout_bits(int n, int Fin, int H, float p, const void* seed, int stream_id, unsigned char* bits,
                                  void* stream) {
  DropRng g;
  PYGAT_REQUIRE(n > 0 && Fin > 0 && H >= 1 && H <= 8 && seed && bits, "dropout_bits: bad arguments (H <= 8)");
  PYGAT_REQUIRE(make_rng(p, seed, (uint32_t)stream_id, &g), "dropout_bits: p=%g outside [0,1]", (double)p);
  const int64_t tot = (int64_t)n * Fin;
  hipLaunchKernelGGL(dropout_bits_kernel, dim3((unsigned)cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, n, Fin, H, g, bits);
  PYGAT_CHECK_LAUNCH("dropout_bits");
  return PYGAT_OK;
}

extern "C" int pygat_dropout_head_sum_bits(int n, int Fin, int H, const float* dxe, int64_t lde, const unsigned char* bits,
                                           float p, float* dx, int64_t ldx, int accumulate, void* stream) {
  PYGAT_REQUIRE(n > 0 && Fin > 0 && H >= 1 && H <= 8 && dxe && bits && dx && p >= 0.f && p <= 1.f, "dropout_head_sum_bits: bad arguments");
  const float scale = p < 1.f ? 1.f / (1.f - p) : 0.f;
  hipLaunchKernelGGL(dropout_head_sum_bits_kernel, dim3((unsigned)cdiv((int64_t)n * Fin, 256)), dim3(256), 0,
                     (hipStream_t)stream, n, Fin, H, dxe, lde, bits, scale, dx, ldx, accumulate);
  PYGAT_CHECK_LAUNCH("dropout_head_sum_bits");
  return PYGAT_OK;
}
