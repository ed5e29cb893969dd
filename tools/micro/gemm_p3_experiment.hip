// EXPERIMENT, NOT BUILT INTO THE LIBRARY (round 4): fp32 GEMM from operands split into their bf16 planes
// ONCE, by a pass of its own, instead of once per work-group inside gemm_x3g_kernel.  Correct (bit-exactness and accuracy
// cases of tests/test_gpu_gemm_split.py passed when it was wired in as pygat_gemm_p3_f32) and no faster: on the PPI level-2
// shapes under rocprofv3 the plane GEMM takes 92.5 us on average against gemm_x3g_kernel's 94 us (139.5 / 80.1 / 62.6), plus
// 16 us of pre-split passes -- the in-kernel split is not what bounds the general split kernel (PPI epoch 2.30 -> 2.53 ms).
// Needs pygat_amd/csrc/gemm_fast.h and a host wrapper of gemm_splitk_reduce_kernel (launch_splitk_reduce) to build.
// Compute-bound fp32 GEMMs of the wide levels (PPI: 3144 x 1024 x 2056 projections, layers.py:35,134,48,166; their input and
// weight gradients; GATv2's projections) from operands split ONCE.
//
// gemm_x3g_kernel (k1_gemm_x3.hip) cuts every fp32 operand into its three bf16 pieces INSIDE the GEMM, once per work-group:
// a 128 x 128 tile re-splits its A rows for each of the N / 128 column tiles and its B columns for each of the M / 128 row
// tiles (PPI level 2: x17 and x25), and the split -- 5.5 VALU operations per element plus the LDS writes -- sits between the
// MFMAs of every step (MFMA pipe 0.55-0.6 busy, 100 TF fp32-equivalent).  Here the cut is a pass of its own:
//
//   presplit   op(A) [M x K] and op(B)^T [N x K] -> three bf16 planes each, K-CONTIGUOUS rows padded to tiles (zero filled),
//              whatever the source layout (a k-strided source -- B of a plain call, A of a transA one -- goes through a
//              32 x 32 LDS transpose): 4 bytes read and 6 written per element, once;
//   gemm_p3    C = sum over the nine piece pairs: 128 x 128 tile per 4-wave work-group, 16 k per step; a step's planes go
//              global -> registers (16-byte loads, issued before the step's MFMAs) -> LDS images [row][16 k] with 48-byte
//              rows (ds_write_b128; the 16 lanes of a ds_read_b128 group hit 16 different 16-byte slots of the bank row) ->
//              fragments by ds_read_b128 -> 36 MFMAs per wave and step with NO vector-ALU work between them but addresses
//              (32 k per step held 48 staging registers beside 128 of accumulators: 208 bytes of scratch).
//              Two accumulator levels as in gemm_x3g_kernel (the running tile is added into a second register set every
//              128 k: an MFMA accumulator is one fp32 summation chain).  Output through the segment table with optional
//              accumulate, or split-K slabs (weight gradients: few tiles, long K).
// Every product is still the exact sum of nine bf16 piece products in fp32: same arithmetic as the other split-bf16 kernels,
// only the order of additions differs (tests/test_gpu_gemm_split.py holds all of them to the same bit-exactness cases).
#include "gemm_fast.h"

namespace pygat {

typedef __bf16 bf16x8_p3 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, uint16_t& h, uint16_t& m, uint16_t& l) {
  const uint32_t u = __float_as_uint(x);
  h = (uint16_t)(u >> 16);
  const float r = x - __uint_as_float(u & 0xffff0000u);
  const uint32_t v = __float_as_uint(r);
  m = (uint16_t)(v >> 16);
  const float q = r - __uint_as_float(v & 0xffff0000u);
  l = (uint16_t)(__float_as_uint(q) >> 16);
}

// planes[p][outer_pad][Kp] (uint16), p = 0 (hi), 1 (mid), 2 (lo).
// CONTIG: src is [outer x K] with leading dimension ld (k contiguous).  One thread per 8 k of a row.
__global__ __launch_bounds__(256) void presplit_rows_kernel(const float* __restrict__ src, int64_t ld, int outer, int64_t K,
                                                            int outer_pad, int64_t Kp, uint16_t* __restrict__ planes) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t k8n = Kp / 8;
  if (idx >= (int64_t)outer_pad * k8n) return;
  const int64_t row = idx / k8n, k0 = (idx % k8n) * 8;
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = 0.f;
  if (row < outer) {
    const float* p = src + row * ld + k0;
    if (k0 + 8 <= K && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
      const float4 a = ld4(p), b = ld4(p + 4);
      x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) if (k0 + j < K) x[j] = p[j];
    }
  }
  uint16_t h[8], m[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) split3(x[j], h[j], m[j], l[j]);
  const int64_t plane = (int64_t)outer_pad * Kp, o = row * Kp + k0;
  auto pack = [](const uint16_t (&v)[8]) {
    return make_uint4((uint32_t)v[0] | ((uint32_t)v[1] << 16), (uint32_t)v[2] | ((uint32_t)v[3] << 16),
                      (uint32_t)v[4] | ((uint32_t)v[5] << 16), (uint32_t)v[6] | ((uint32_t)v[7] << 16));
  };
  *reinterpret_cast<uint4*>(planes + o) = pack(h);
  *reinterpret_cast<uint4*>(planes + plane + o) = pack(m);
  *reinterpret_cast<uint4*>(planes + 2 * plane + o) = pack(l);
}

// src is [K x outer] with leading dimension ld (k strided): 32 x 32 tiles transposed through LDS.
__global__ __launch_bounds__(256) void presplit_cols_kernel(const float* __restrict__ src, int64_t ld, int outer, int64_t K,
                                                            int outer_pad, int64_t Kp, uint16_t* __restrict__ planes) {
  __shared__ float tile[32][33];
  const int64_t k0 = (int64_t)blockIdx.x * 32;
  const int o0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 8 rows of 32 threads
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int64_t k = k0 + r;
    const int o = o0 + tx;
    tile[r][tx] = (k < K && o < outer) ? src[k * ld + o] : 0.f;
  }
  __syncthreads();
  const int64_t plane = (int64_t)outer_pad * Kp;
#pragma unroll
  for (int r = ty; r < 32; r += 8) {          // output row o0 + r, k = k0 + tx
    uint16_t h, m, l;
    split3(tile[tx][r], h, m, l);
    const int64_t o = (int64_t)(o0 + r) * Kp + k0 + tx;
    planes[o] = h; planes[plane + o] = m; planes[2 * plane + o] = l;
  }
}

struct P3Args {
  int M, N;
  int64_t Kp;            // padded K (multiple of 32)
  int Mp, Np;            // padded rows of the plane tables (multiples of 128)
  const uint16_t* A3;    // [3][Mp][Kp]
  const uint16_t* B3;    // [3][Np][Kp]
  int64_t k_per_split;   // multiple of 32
  float* ws;             // split-K slabs [splits][M][N] or nullptr (direct output)
  pygat_out_segments out;
  int accumulate;
};

constexpr int P3_FLUSH_K = 128;                // k between the two accumulator levels

// BK = 16 or 32 k per step.  Image rows are BK k + 8 of padding: 48 / 80 bytes -- the 16 lanes of a ds_read_b128 group
// (16 rows) then hit 16 different 16-byte slots of the 256-byte bank row.
template <int BK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_p3_kernel(P3Args g) {
  constexpr int RS = BK + 8;                   // uint16 per image row
  constexpr int IMG = 128 * RS;                // one piece image
  constexpr int SEGS = BK / 8;                 // 16-byte segments per row
  constexpr int RPP = 256 / SEGS;              // rows covered by one pass of the 256 loader threads
  constexpr int NL = 128 / RPP;                // passes (loads per thread, piece and operand)
  constexpr int FLUSH = P3_FLUSH_K / BK;
  extern __shared__ __attribute__((aligned(16))) uint16_t lds_p3[];   // A(h, m, l), B(h, m, l): 6 images
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.z * 128;
  const int64_t kbeg = (int64_t)blockIdx.x * g.k_per_split;
  const int64_t kend = (kbeg + g.k_per_split < g.Kp) ? kbeg + g.k_per_split : g.Kp;
  const int nsteps = (int)((kend - kbeg) / BK);
  // loader role: thread (row = tid / SEGS + RPP j, seg = tid % SEGS) moves the 16 bytes k = 8 seg .. + 7 of its rows, per piece
  const int lrow = tid / SEGS, lseg = tid % SEGS;
  const int64_t planeA = (int64_t)g.Mp * g.Kp, planeB = (int64_t)g.Np * g.Kp;
  const uint16_t* ga = g.A3 + (int64_t)(m0 + lrow) * g.Kp + kbeg + 8 * lseg;
  const uint16_t* gb = g.B3 + (int64_t)(n0 + lrow) * g.Kp + kbeg + 8 * lseg;
  const int64_t rowpp = (int64_t)RPP * g.Kp;
  uint16_t* la = lds_p3 + lrow * RS + 8 * lseg;
  uint16_t* lb = la + 3 * IMG;
  // staging registers as SCALARS moved by macros (as arrays -- through lambdas or unrolled loops -- hipcc keeps them in scratch)
  uint4 ra00, ra10, ra20, rb00, rb10, rb20, ra01, ra11, ra21, rb01, rb11, rb21;
#define PYGAT_P3_LD1(J, KO)                                                                        \
    ra0##J = *reinterpret_cast<const uint4*>(ga + (J) * rowpp + (KO));                             \
    ra1##J = *reinterpret_cast<const uint4*>(ga + planeA + (J) * rowpp + (KO));                    \
    ra2##J = *reinterpret_cast<const uint4*>(ga + 2 * planeA + (J) * rowpp + (KO));                \
    rb0##J = *reinterpret_cast<const uint4*>(gb + (J) * rowpp + (KO));                             \
    rb1##J = *reinterpret_cast<const uint4*>(gb + planeB + (J) * rowpp + (KO));                    \
    rb2##J = *reinterpret_cast<const uint4*>(gb + 2 * planeB + (J) * rowpp + (KO));
#define PYGAT_P3_ST1(J)                                                                            \
    *reinterpret_cast<uint4*>(la + (J) * RPP * RS) = ra0##J;                                       \
    *reinterpret_cast<uint4*>(la + IMG + (J) * RPP * RS) = ra1##J;                                 \
    *reinterpret_cast<uint4*>(la + 2 * IMG + (J) * RPP * RS) = ra2##J;                             \
    *reinterpret_cast<uint4*>(lb + (J) * RPP * RS) = rb0##J;                                       \
    *reinterpret_cast<uint4*>(lb + IMG + (J) * RPP * RS) = rb1##J;                                 \
    *reinterpret_cast<uint4*>(lb + 2 * IMG + (J) * RPP * RS) = rb2##J;
#define PYGAT_P3_LOAD(STEP)                                                                        \
  {                                                                                                \
    const int64_t ko__ = BK * (int64_t)(STEP);                                                     \
    PYGAT_P3_LD1(0, ko__)                                                                          \
    if constexpr (NL == 2) { PYGAT_P3_LD1(1, ko__) }                                               \
  }
#define PYGAT_P3_STAGE()                                                                           \
  {                                                                                                \
    PYGAT_P3_ST1(0)                                                                                \
    if constexpr (NL == 2) { PYGAT_P3_ST1(1) }                                                     \
  }
  if constexpr (NL == 1) { ra01 = ra11 = ra21 = rb01 = rb11 = rb21 = make_uint4(0, 0, 0, 0); }
  f32x16 acc[2][2], tot[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }
  // consumer role: wave (w >> 1, w & 1) owns rows 64 (w >> 1) .., columns 64 (w & 1) .. of the tile
  const uint16_t* fa = lds_p3 + (64 * (w >> 1) + fr) * RS + 8 * fh;
  const uint16_t* fb = lds_p3 + 3 * IMG + (64 * (w & 1) + fr) * RS + 8 * fh;

  if (nsteps > 0) { PYGAT_P3_LOAD(0) PYGAT_P3_STAGE() }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    if (s + 1 < nsteps) PYGAT_P3_LOAD(s + 1)    // in flight during this step's MFMAs
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      uint4 fbq[2][3];                         // B columns 0-31 / 32-63 of the wave's block, three pieces
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        fbq[0][p] = *reinterpret_cast<const uint4*>(fb + p * IMG + 16 * kk);
        fbq[1][p] = *reinterpret_cast<const uint4*>(fb + p * IMG + 32 * RS + 16 * kk);
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        uint4 faq[3];                          // A rows 32 tm .. + 31
#pragma unroll
        for (int p = 0; p < 3; ++p) faq[p] = *reinterpret_cast<const uint4*>(fa + p * IMG + tm * 32 * RS + 16 * kk);
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          f32x16 c = acc[tm][tn];
          // all nine piece products, small terms first (pieces: 0 hi, 1 mid, 2 lo)
#define PYGAT_P3_MMA(PA, PB) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_p3, faq[PA]), \
                                                                          __builtin_bit_cast(bf16x8_p3, fbq[tn][PB]), c, 0, 0, 0);
          PYGAT_P3_MMA(2, 2) PYGAT_P3_MMA(2, 1) PYGAT_P3_MMA(1, 2) PYGAT_P3_MMA(2, 0) PYGAT_P3_MMA(0, 2)
          PYGAT_P3_MMA(1, 1) PYGAT_P3_MMA(1, 0) PYGAT_P3_MMA(0, 1) PYGAT_P3_MMA(0, 0)
#undef PYGAT_P3_MMA
          acc[tm][tn] = c;
        }
      }
    }
    if ((s % FLUSH) == FLUSH - 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) { tot[i][j][r] += acc[i][j][r]; acc[i][j][r] = 0.f; }
    }
    __syncthreads();                           // every wave has read this step's images
    if (s + 1 < nsteps) {
      PYGAT_P3_STAGE()
      __syncthreads();
    }
  }
#undef PYGAT_P3_LOAD
#undef PYGAT_P3_STAGE
#undef PYGAT_P3_LD1
#undef PYGAT_P3_ST1
  const int wm0 = m0 + 64 * (w >> 1), wn0 = n0 + 64 * (w & 1);
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int col = wn0 + 32 * tn + fr;
      if (col >= g.N) continue;
      float* base;
      int64_t ld;
      if (g.ws) { base = g.ws + (int64_t)blockIdx.x * g.M * g.N + col; ld = g.N; }
      else base = out_segment(g.out, col, ld);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm0 + 32 * tm + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (row < g.M) {
          const float v = tot[tm][tn][r] + acc[tm][tn][r];
          float* p = base + (int64_t)row * ld;
          if (g.accumulate && !g.ws) *p += v; else *p = v;
        }
      }
    }
}

#ifndef PYGAT_P3_BK
#define PYGAT_P3_BK 16
#endif

static inline int64_t p3_pad(int64_t v, int64_t q) { return cdiv(v, q) * q; }

}  // namespace pygat

using namespace pygat;

// bytes of the workspace of pygat_gemm_p3_f32: the six bf16 planes (padded to tiles) + the split-K slabs
extern "C" size_t pygat_gemm_p3_workspace_bytes(int M, int N, int64_t K, int split_k) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (split_k < 1) split_k = 1;
  const int64_t kps = p3_pad(cdiv(K, split_k), 32), Kp = kps * cdiv(K, kps);
  const int64_t Mp = p3_pad(M, 128), Np = p3_pad(N, 128);
  const size_t planes = (size_t)3 * (size_t)(Mp + Np) * (size_t)Kp * sizeof(uint16_t);
  const int64_t splits = cdiv(K, kps);
  return planes + 256 + (splits > 1 ? (size_t)splits * (size_t)M * (size_t)N * sizeof(float) : 0);
}

extern "C" int pygat_gemm_p3_f32(int transA, int transB, int M, int N, int64_t K, const float* A, int64_t lda, const float* B,
                                 int64_t ldb, const pygat_out_segments* out, int accumulate, int split_k, void* ws, void* stream) {
  PYGAT_REQUIRE(A && B && out && ws && aligned16(ws), "gemm_p3: null pointer / unaligned workspace");
  PYGAT_REQUIRE(M > 0 && N > 0 && K > 0 && !(transA && transB), "gemm_p3: bad sizes M=%d N=%d K=%lld (or transA && transB)", M, N, (long long)K);
  PYGAT_REQUIRE(out->nseg >= 1 && out->nseg <= PYGAT_MAX_SEGMENTS && out->col_start[0] == 0 && out->col_start[out->nseg] == N,
                "gemm_p3: output segments must cover [0,N) (nseg=%d)", out->nseg);
  for (int s = 0; s < out->nseg; ++s)
    PYGAT_REQUIRE(out->ptr[s] && out->col_start[s + 1] > out->col_start[s] && out->ld[s] >= out->col_start[s + 1] - out->col_start[s],
                  "gemm_p3: bad output segment %d", s);
  PYGAT_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N), "gemm_p3: leading dimension too small");
  if (split_k < 1) split_k = 1;
  hipStream_t st = (hipStream_t)stream;
  const int64_t kps = p3_pad(cdiv(K, split_k), 32);
  const int splits = (int)cdiv(K, kps);
  const int64_t Kp = kps * splits;
  const int Mp = (int)p3_pad(M, 128), Np = (int)p3_pad(N, 128);
  uint16_t* A3 = reinterpret_cast<uint16_t*>(ws);
  uint16_t* B3 = A3 + (size_t)3 * Mp * Kp;
  float* slabs = splits > 1 ? reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + (((size_t)3 * (size_t)(Mp + Np) * (size_t)Kp * sizeof(uint16_t) + 255) / 256) * 256)
                            : nullptr;
  // op(A) [M x K]: plain A is k-contiguous, A of a transA call ([K x M]) k-strided; op(B)^T [N x K]: B of a transB call
  // ([N x K]) is k-contiguous, a plain B ([K x N]) k-strided
  auto presplit = [&](const float* src, int64_t ld, bool kcontig, int outer, int outer_pad, uint16_t* planes) {
    if (kcontig) {
      const int64_t n8 = (int64_t)outer_pad * (Kp / 8);
      hipLaunchKernelGGL(presplit_rows_kernel, dim3((unsigned)cdiv(n8, 256)), dim3(256), 0, st, src, ld, outer, K, outer_pad, Kp, planes);
    } else {
      hipLaunchKernelGGL(presplit_cols_kernel, dim3((unsigned)(Kp / 32), (unsigned)(outer_pad / 32)), dim3(256), 0, st, src, ld, outer, K,
                         outer_pad, Kp, planes);
    }
  };
  presplit(A, lda, !transA, M, Mp, A3);
  presplit(B, ldb, transB != 0, N, Np, B3);
  PYGAT_CHECK_LAUNCH("gemm_p3 presplit");
  P3Args g;
  g.M = M; g.N = N; g.Kp = Kp; g.Mp = Mp; g.Np = Np; g.A3 = A3; g.B3 = B3; g.k_per_split = kps; g.ws = slabs; g.out = *out;
  g.accumulate = accumulate;
  constexpr int BK = PYGAT_P3_BK;
  constexpr size_t lds = 6 * 128 * (BK + 8) * sizeof(uint16_t);
  int dev = -1;
  (void)hipGetDevice(&dev);
  static bool attr_set[64] = {};   // per device: the attribute belongs to the device's code object
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_p3_kernel<BK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL(gemm_p3_kernel<BK>, dim3((unsigned)splits, (unsigned)(Mp / 128), (unsigned)(Np / 128)), dim3(256), lds, st, g);
  PYGAT_CHECK_LAUNCH("gemm_p3");
  if (splits > 1) return launch_splitk_reduce(M, N, splits, slabs, *out, accumulate, st);
  return PYGAT_OK;
}
