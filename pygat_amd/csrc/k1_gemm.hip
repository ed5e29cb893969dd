// K1 / K5 -- fp32 MFMA GEMM for the dense projections of the GAT layer (gfx950).
//
// Replaces, per level and for all local heads at once:
//   Wh = h W              (reference layers.py:35,134)      C = X * Wcat
//   h W_skip              (layers.py:48,166)                extra columns of Wcat
//   Wh a[:F'], Wh a[F':]  (layers.py:60-61)                 extra columns W_h a_src_h, W_h a_dst_h
// and in the backward the weight / input gradients the reference gets from ATen autograd:
//   dW = X^T dWh (transA, split-K), dX = dWh Wcat^T (transB).
//
// v_mfma_f32_32x32x2_f32: exact fp32 (k-ordered fma chain), 64 FLOP/clk/SIMD.
// Work-group = 4 waves, tile 128 x (32*NT) x 16; wave w owns rows [32w, 32w+32)
// and all NT column blocks.  LDS images are k-major (As[k][m], Bs[k][n]) so a
// fragment read is 32 consecutive floats per half-wave: conflict-free ds_read_b32.
#include "common.h"
#include "narrow.h"
#include <type_traits>

namespace pygat {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;
constexpr int BK = 16;
constexpr int PAD = 4;  // (BM+PAD)*4 mod 32 == 16 -> the 4-scalar transposing LDS writes are 2-way at worst

struct GemmArgs {
  int M, N;
  int64_t K;
  const float* A;
  int64_t lda;
  const float* B;
  int64_t ldb;
  pygat_out_segments out;
  int accumulate;
  int64_t k_per_split;
  float* ws;        // != nullptr: partial tile results go to ws[z][M][N]
  int a_vec, b_vec; // 16-B vector loads allowed (base and ld aligned)
  ColBlocks ab, cb; // column blocks of the stored A / of C (common.h); one block = an ordinary matrix
};

constexpr ColBlocks kOneBlock = {62, 0};

// operand stored [rows x K] (K contiguous): tile ROWS x 16, staged transposed into S[k][row]
template <int ROWS, int NV, int BKT = BK>
__device__ __forceinline__ void load_kcontig(const float* __restrict__ P, int64_t ld, int vec, int row0,
                                             int nrows, int64_t k0, int64_t kend, float4 (&r)[NV], const ColBlocks& cbk = kOneBlock) {
  constexpr int QK = BKT / 4;   // float4 units per row of the tile
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int idx = threadIdx.x + 256 * i;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < ROWS * QK) {
      int row = row0 + (idx / QK);
      int64_t k = k0 + (idx % QK) * 4;
      if (row < nrows && k < kend) {
        const float* p = P + (int64_t)row * ld + blk_off(k, cbk);
        if (vec && k + 3 < kend) {
          v = ld4(p);
        } else {
          v.x = p[0];
          if (k + 1 < kend) v.y = p[1];
          if (k + 2 < kend) v.z = p[2];
          if (k + 3 < kend) v.w = p[3];
        }
      }
    }
    r[i] = v;
  }
}
template <int ROWS, int NV, int LDS_LD, int BKT = BK>
__device__ __forceinline__ void store_kcontig(float* S, const float4 (&r)[NV]) {
  constexpr int QK = BKT / 4;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int idx = threadIdx.x + 256 * i;
    if (idx < ROWS * QK) {
      int row = idx / QK, kq = (idx % QK) * 4;
      S[(kq + 0) * LDS_LD + row] = r[i].x;
      S[(kq + 1) * LDS_LD + row] = r[i].y;
      S[(kq + 2) * LDS_LD + row] = r[i].z;
      S[(kq + 3) * LDS_LD + row] = r[i].w;
    }
  }
}

// operand stored [K x cols] (K strided): tile 16 x COLS, staged as is into S[k][col]
template <int COLS, int NV, int BKT = BK>
__device__ __forceinline__ void load_kstrided(const float* __restrict__ P, int64_t ld, int vec, int col0,
                                              int ncols, int64_t k0, int64_t kend, float4 (&r)[NV], const ColBlocks& cbk = kOneBlock) {
  constexpr int Q = COLS / 4;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int idx = threadIdx.x + 256 * i;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < BKT * Q) {
      int64_t k = k0 + idx / Q;
      int c = col0 + (idx % Q) * 4;
      if (k < kend && c < ncols) {
        const float* p = P + k * ld + blk_off(c, cbk);
        if (vec && c + 3 < ncols) {
          v = ld4(p);
        } else {
          v.x = p[0];
          if (c + 1 < ncols) v.y = p[1];
          if (c + 2 < ncols) v.z = p[2];
          if (c + 3 < ncols) v.w = p[3];
        }
      }
    }
    r[i] = v;
  }
}
template <int COLS, int NV, int LDS_LD, int BKT = BK>
__device__ __forceinline__ void store_kstrided(float* S, const float4 (&r)[NV]) {
  constexpr int Q = COLS / 4;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int idx = threadIdx.x + 256 * i;
    if (idx < BKT * Q) st4(S + (idx / Q) * LDS_LD + (idx % Q) * 4, r[i]);
  }
}


// Branch-free variants (16-byte aligned rows; K slab, and the extent along which a float4 runs, multiples of 4): every
// load is unconditional -- rows / columns past the operand are CLAMPED to the last valid ones (their products only
// reach rows / columns of C that are never stored) and k positions past the slab are multiplied by 0.  In the guarded
// loaders above each value is `in range ? load : 0`; hipcc turns that select into a branch around the load and waits
// vmcnt(0) for it BEFORE the MFMA phase of the current k-tile -- the prefetch of the next tile then overlaps nothing.
template <int ROWS, int NV, int BKT>
__device__ __forceinline__ void load_kcontig_fast(const float* __restrict__ P, int64_t ld, int row0, int nrows, int64_t k0,
                                                  int64_t kend, float4 (&r)[NV], const ColBlocks& cbk = kOneBlock) {
  constexpr int QK = BKT / 4;
  static_assert(ROWS * QK == NV * 256, "exact cover");
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = threadIdx.x + 256 * i;
    int row = row0 + idx / QK;
    row = row < nrows ? row : nrows - 1;
    const int64_t k = k0 + (idx % QK) * 4;
    const float keep = k < kend ? 1.f : 0.f;
    const float4 v = ld4(P + (int64_t)row * ld + blk_off(k < kend ? k : kend - 4, cbk));
    r[i] = make_float4(v.x * keep, v.y * keep, v.z * keep, v.w * keep);
  }
}
template <int COLS, int NV, int BKT>
__device__ __forceinline__ void load_kstrided_fast(const float* __restrict__ P, int64_t ld, int col0, int ncols, int64_t k0,
                                                   int64_t kend, float4 (&r)[NV], const ColBlocks& cbk = kOneBlock) {
  constexpr int Q = COLS / 4;
  static_assert(BKT * Q == NV * 256, "exact cover");
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = threadIdx.x + 256 * i;
    const int64_t k = k0 + idx / Q;
    const float keep = k < kend ? 1.f : 0.f;
    int c = col0 + (idx % Q) * 4;
    c = c < ncols ? c : ncols - 4;
    const float4 v = ld4(P + (k < kend ? k : kend - 1) * ld + blk_off(c, cbk));
    r[i] = make_float4(v.x * keep, v.y * keep, v.z * keep, v.w * keep);
  }
}

// BKT = depth of a k-tile: 16, or 32 for long K (half the barriers and LDS hand-offs per MFMA; twice the LDS)
template <bool TA, bool TB, int NT, int BKT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  constexpr int BK = BKT;            // shadows the namespace-wide default inside this kernel
  constexpr int BN = 32 * NT;
  constexpr int LDAS = BM + PAD, LDBS = BN + PAD;
  constexpr int NVA = BM * BK / 4 / 256;
  constexpr int NVB = (BN * BK / 4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float gm_sm[];   // As[2][BK * LDAS] | Bs[2][BK * LDBS]
  float (*As)[BK * LDAS] = reinterpret_cast<float (*)[BK * LDAS]>(gm_sm);
  float (*Bs)[BK * LDBS] = reinterpret_cast<float (*)[BK * LDBS]>(gm_sm + 2 * BK * LDAS);

  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int64_t kbeg = (int64_t)blockIdx.z * g.k_per_split;
  const int64_t kend = (kbeg + g.k_per_split < g.K) ? kbeg + g.k_per_split : g.K;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;

  // Two-level accumulation.  An MFMA accumulator is ONE fp32 summation chain over k: after n additions its rounding
  // error has grown like n (every addition rounds at the size of the running sum), where a CPU BLAS -- 64 and more
  // independent partial sums -- stays near sqrt(n): a K = 384 input gradient sat at 4.5 x, a K = 9001 chain at 7.5 x
  // the error of the fp32 CPU product (tests/parity.py prices every result at <= 4 x).  So the chain is cut: every
  // FLUSH k the running tile is added into a second register set and restarted from zero (chains of FLUSH / 2 small
  // sums + K / FLUSH large ones; K = 1024: a third of the error).  Costs 16 NT registers, which is why the
  // 192- and 256-column tiles of the first version are gone (launch_gemm: at most 160 columns per tile).
  constexpr int FLUSH_TILES = 64 / BK;
  f32x16 acc[NT], acc2[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; acc2[i][r] = 0.f; }
  auto flush = [&]() {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc2[i][r] += acc[i][r]; acc[i][r] = 0.f; }
  };

  constexpr bool EXACT = (BN * BK / 4) % 256 == 0;      // the B tile is a whole number of float4 per thread
  // branch-free staging loads (see load_*_fast): uniform in the launch
  const bool interior = EXACT && g.a_vec && g.b_vec && kend > kbeg && (kbeg & 3) == 0 && (kend & 3) == 0 && kend - kbeg >= 4 &&
                        (!TA || ((g.M & 3) == 0 && g.M >= 4)) && (TB || ((g.N & 3) == 0 && g.N >= 4));
  const int fr = lane & 31, fk = lane >> 5;
  const int64_t nkt = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;
  // the whole k loop exists twice (FAST / guarded): one loop with a branch per load would merge the two loaders'
  // registers at every join and wait for the loads there
  auto kloop = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    float4 ra[NVA], rb[NVB];
    auto gload = [&](int64_t k0) {
      if constexpr (FAST && EXACT) {
        if constexpr (TA) load_kstrided_fast<BM, NVA, BK>(g.A, g.lda, m0, g.M, k0, kend, ra, g.ab);
        else load_kcontig_fast<BM, NVA, BK>(g.A, g.lda, m0, g.M, k0, kend, ra, g.ab);
        if constexpr (TB) load_kcontig_fast<BN, NVB, BK>(g.B, g.ldb, n0, g.N, k0, kend, rb);
        else load_kstrided_fast<BN, NVB, BK>(g.B, g.ldb, n0, g.N, k0, kend, rb);
      } else {
        if constexpr (TA) load_kstrided<BM, NVA, BK>(g.A, g.lda, g.a_vec, m0, g.M, k0, kend, ra, g.ab);
        else load_kcontig<BM, NVA, BK>(g.A, g.lda, g.a_vec, m0, g.M, k0, kend, ra, g.ab);
        if constexpr (TB) load_kcontig<BN, NVB, BK>(g.B, g.ldb, g.b_vec, n0, g.N, k0, kend, rb);
        else load_kstrided<BN, NVB, BK>(g.B, g.ldb, g.b_vec, n0, g.N, k0, kend, rb);
      }
    };
    auto sstore = [&](int buf) {
      if constexpr (TA) store_kstrided<BM, NVA, LDAS, BK>(As[buf], ra);
      else store_kcontig<BM, NVA, LDAS, BK>(As[buf], ra);
      if constexpr (TB) store_kcontig<BN, NVB, LDBS, BK>(Bs[buf], rb);
      else store_kstrided<BN, NVB, LDBS, BK>(Bs[buf], rb);
    };
    auto mma_tile = [&](int buf) {
      const float* as = As[buf] + fk * LDAS + 32 * w + fr;
      const float* bs = Bs[buf] + fk * LDBS + fr;
      // all fragments of a group of k-pairs are read from LDS BEFORE its MFMAs are issued: left alone, hipcc sinks
      // every ds_read to its first use through one register pair (ds_read -> lgkmcnt(0) -> 2 MFMAs -> ds_read ...),
      // which exposes an LDS latency per MFMA pair
      constexpr int KG = 8;   // k-pairs per group
#pragma unroll
      for (int k8 = 0; k8 < BK; k8 += 2 * KG) {
        float av[KG], bv[KG][NT];
#pragma unroll
        for (int q = 0; q < KG; ++q) {
          av[q] = as[(k8 + 2 * q) * LDAS];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bv[q][nt] = bs[(k8 + 2 * q) * LDBS + 32 * nt];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < KG; ++q)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q][nt], acc[nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if constexpr (FAST && EXACT) {
      // two register sets, two k-tiles of loads in flight: one tile of MFMAs (0.85 us at 4 column tiles) does not cover
      // an HBM / L2 round trip under load, and the grid of a PPI-sized projection leaves 1-2 work-groups per CU
      float4 ra2[NVA], rb2[NVB];
      auto gload2 = [&](int64_t k0) {
        if constexpr (TA) load_kstrided_fast<BM, NVA, BK>(g.A, g.lda, m0, g.M, k0, kend, ra2, g.ab);
        else load_kcontig_fast<BM, NVA, BK>(g.A, g.lda, m0, g.M, k0, kend, ra2, g.ab);
        if constexpr (TB) load_kcontig_fast<BN, NVB, BK>(g.B, g.ldb, n0, g.N, k0, kend, rb2);
        else load_kstrided_fast<BN, NVB, BK>(g.B, g.ldb, n0, g.N, k0, kend, rb2);
      };
      auto sstore2 = [&](int buf) {
        if constexpr (TA) store_kstrided<BM, NVA, LDAS, BK>(As[buf], ra2);
        else store_kcontig<BM, NVA, LDAS, BK>(As[buf], ra2);
        if constexpr (TB) store_kcontig<BN, NVB, LDBS, BK>(Bs[buf], rb2);
        else store_kstrided<BN, NVB, LDBS, BK>(Bs[buf], rb2);
      };
      // tile t is loaded into set (t & 1) and staged into LDS buffer (t & 1); loads past the last tile are clamped
      auto kof = [&](int64_t t) { return kbeg + (t < nkt ? t : nkt - 1) * BK; };
      gload(kof(0));
      gload2(kof(1));
      sstore(0);
      __syncthreads();
      for (int64_t kt = 0; kt < nkt; kt += 2) {
        gload(kof(kt + 2));            // set 0 is free: its tile kt is in LDS buffer 0
        mma_tile(0);
        if (kt + 1 < nkt) sstore2(1);  // tile kt+1, loaded one iteration ago
        __syncthreads();
        if (kt + 1 < nkt) {
          gload2(kof(kt + 3));
          mma_tile(1);
          if (kt + 2 < nkt) sstore(0);
          __syncthreads();
        }
        if (((kt >> 1) % (FLUSH_TILES / 2)) == FLUSH_TILES / 2 - 1) flush();
      }
    } else {
      if (nkt > 0) {
        gload(kbeg);
        sstore(0);
      }
      __syncthreads();
      for (int64_t kt = 0; kt < nkt; ++kt) {
        const int buf = (int)(kt & 1);
        if (kt + 1 < nkt) gload(kbeg + (kt + 1) * BK);
        mma_tile(buf);
        if (kt + 1 < nkt) sstore(buf ^ 1);
        __syncthreads();
        if ((kt % FLUSH_TILES) == FLUSH_TILES - 1) flush();
      }
    }
  };
  if (interior) kloop(std::true_type{});
  else kloop(std::false_type{});
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] += acc2[i][r];

  // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = n0 + 32 * nt + fr;
    if (col >= g.N) continue;
    float* base;
    int64_t ld;
    if (g.ws) {
      base = g.ws + ((int64_t)blockIdx.z * g.M) * g.N + col;
      ld = g.N;
    } else if (g.cb.lw < 62) {      // column-blocked C (one segment)
      base = g.out.ptr[0] + blk_off(col, g.cb);
      ld = g.out.ld[0];
    } else {
      int s = 0;
#pragma unroll
      for (int q = 1; q < PYGAT_MAX_SEGMENTS; ++q)
        if (q < g.out.nseg && col >= g.out.col_start[q]) s = q;
      base = g.out.ptr[s] + (col - g.out.col_start[s]);
      ld = g.out.ld[s];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + 32 * w + (r & 3) + 8 * (r >> 2) + 4 * fk;
      if (row < g.M) {
        float* p = base + (int64_t)row * ld;
        if (!g.ws && g.accumulate) *p += acc[nt][r];
        else *p = acc[nt][r];
      }
    }
  }
}

// deterministic reduction of split-K partials, routed to the segments.  A 512-thread work-group owns
// 64 consecutive output elements: wave q sums slabs q, q+8, ... in order (256-B coalesced reads,
// unrolled so several are in flight), then the 8 partial sums are added in wave order -- the same
// association on every run.
__global__ __launch_bounds__(512) void gemm_splitk_reduce_kernel(int M, int N, int splits,
                                                                 const float* __restrict__ ws,
                                                                 pygat_out_segments out, int accumulate) {
  const int64_t tot = (int64_t)M * N;
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  float v = 0.f;
  if (i < tot) {
    int z = q;
    for (; z + 24 < splits; z += 32) {
      const float x0 = ws[(int64_t)z * tot + i], x1 = ws[(int64_t)(z + 8) * tot + i];
      const float x2 = ws[(int64_t)(z + 16) * tot + i], x3 = ws[(int64_t)(z + 24) * tot + i];
      v += x0; v += x1; v += x2; v += x3;
    }
    for (; z < splits; z += 8) v += ws[(int64_t)z * tot + i];
  }
  __shared__ float sm[8][64];
  sm[q][lane] = v;
  __syncthreads();
  if (q == 0 && i < tot) {
    float acc = 0.f;
#pragma unroll
    for (int z = 0; z < 8; ++z) acc += sm[z][lane];
    const int row = (int)(i / N), col = (int)(i % N);
    int s = 0;
    for (int k = 1; k < out.nseg; ++k)
      if (col >= out.col_start[k]) s = k;
    float* p = out.ptr[s] + (int64_t)row * out.ld[s] + (col - out.col_start[s]);
    if (accumulate) *p += acc; else *p = acc;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Projection under per-head input dropout (layers.py:34,132: every head draws its own mask, models.py:32,34).
//
//   forward (TA = false)   C_h[M x cols] = scale * (X .* m_h)[M x K] * B_h[K x cols]            M = nodes, K = Fin
//   weight grad (TA = true) C_h[M x cols] = scale * (X .* m_h)^T[M x K] * D_h[K x cols]          M = Fin,  K = nodes
//
// for ALL heads in one launch, X read ONCE: the A tile is staged in LDS together with its mask bits (one byte
// per element, bit h = "head h keeps this input", written by dropout_bits_kernel), and the heads are an inner loop
// over the fragments -- a_h = bit_h ? x : 0 is one v_cndmask per MFMA.  Replaces round 1's wide operand
// A'[N, H*Fin] times a block-diagonal weight stack: no N*H*Fin table (Citeseer: 394 MB -> 12 MB of bits), no
// products with the H-1 zero blocks.  B_h / C_h are column slices: head h, column c < Fp of source 1
// (W or dWh), Fp <= c < 2Fp of source 2 (skip weights or Gp), each at column h*Fp of its table.
struct HeadMaskArgs {
  int M;
  int64_t K;
  const float* A;          // X: [M x K] (TA = false) or [K x M] (TA = true), row stride lda
  int64_t lda;
  const unsigned char* bits;  // same indexing and row stride as A
  const float* B1;         // [K x .], row stride ldb1, columns h*Fp + c
  int64_t ldb1;
  const float* B2;         // second source or nullptr
  int64_t ldb2;
  float* C1;               // [M x .], row stride ldc1, columns h*Fp + c
  int64_t ldc1;
  float* C2;
  int64_t ldc2;
  int64_t c_split_stride;  // TA: partial results of K slab z go to C + z * c_split_stride
  int H, Fp;
  float scale;
  int64_t k_per_split;
};

// 16-byte / 4-byte loads that are only 4-byte / 1-byte aligned (rows of Fin = 1433 floats, mask bytes): gfx950 takes
// them as one global_load_dwordx4 / global_load_dword (unaligned access mode of the amdhsa target)
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32_u __attribute__((aligned(1)));

template <bool TA, int HB, int NTH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gemm_headmask_kernel(HeadMaskArgs g) {
  constexpr int NTOT = HB * NTH, BN = 32 * NTOT;
  constexpr int LDAS = BM + PAD, LDBS = BN + PAD;
  constexpr int NVA = 2;
  constexpr int NU = (BK * HB * 8 * NTH + 255) / 256;   // float4 units of B per thread and k-tile (<= 32*NTH columns per head)
  extern __shared__ __attribute__((aligned(16))) float hm_sm[];
  float* As = hm_sm;                                   // [2][BK * LDAS]
  float* Bs = As + 2 * BK * LDAS;                      // [2][BK * LDBS]
  unsigned char* Ms = reinterpret_cast<unsigned char*>(Bs + 2 * BK * LDBS);   // [2][BK * LDAS] mask byte per element
  const int m0 = blockIdx.x * BM, hb0 = blockIdx.y * HB;
  const int64_t kbeg = (int64_t)blockIdx.z * g.k_per_split;
  const int64_t kend = (kbeg + g.k_per_split < g.K) ? kbeg + g.k_per_split : g.K;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int cols_h = g.B2 ? 2 * g.Fp : g.Fp;
  const int q_h = cols_h >> 2;                         // float4 units per head and k row (Fp % 4 == 0)
  const int units = BK * HB * q_h;

  f32x16 acc[NTOT];
#pragma unroll
  for (int i = 0; i < NTOT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  float4 ra[NVA];
  uint32_t rm[NVA];   // 4 mask bytes matching ra
  float4 rb[NU];
  auto gload = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
      const int idx = threadIdx.x + 256 * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      uint32_t mv = 0;
      int64_t off;     // element offset of the 4 values (same for x and its mask bytes)
      int nval;        // how many of them exist
      if constexpr (TA) {   // element (k, m .. m+3)
        const int64_t k = k0 + idx / (BM / 4);
        const int c = m0 + (idx % (BM / 4)) * 4;
        off = k * g.lda + c;
        nval = (k < kend) ? (g.M - c) : 0;
      } else {              // element (row, k .. k+3)
        const int row = m0 + (idx >> 2);
        const int64_t k = k0 + (idx & 3) * 4;
        off = (int64_t)row * g.lda + k;
        nval = (row < g.M) ? (int)(kend - k) : 0;
      }
      if (nval >= 4) {
        const f32x4_u t = *reinterpret_cast<const f32x4_u*>(g.A + off);
        v = make_float4(t.x, t.y, t.z, t.w);
        mv = *reinterpret_cast<const u32_u*>(g.bits + off);
      } else if (nval > 0) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < nval; ++q) { t[q] = g.A[off + q]; mv |= (uint32_t)g.bits[off + q] << (8 * q); }
        v = make_float4(t[0], t[1], t[2], t[3]);
      }
      ra[i] = v; rm[i] = mv;
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int u = threadIdx.x + 256 * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (u < units) {
        const int kk = u / (HB * q_h), r2 = u - kk * (HB * q_h), hl = r2 / q_h, c = (r2 - hl * q_h) * 4;
        const int64_t k = k0 + kk;
        const int h = hb0 + hl;
        if (k < kend && h < g.H)
          v = c < g.Fp ? ld4(g.B1 + k * g.ldb1 + (int64_t)h * g.Fp + c) : ld4(g.B2 + k * g.ldb2 + (int64_t)h * g.Fp + (c - g.Fp));
      }
      rb[i] = v;
    }
  };
  auto sstore = [&](int buf) {
    float* as = As + buf * BK * LDAS;
    unsigned char* ms = Ms + buf * BK * LDAS;
    if constexpr (TA) store_kstrided<BM, NVA, LDAS>(as, ra);
    else store_kcontig<BM, NVA, LDAS>(as, ra);
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
      const int idx = threadIdx.x + 256 * i;
      if constexpr (TA) {
        *reinterpret_cast<uint32_t*>(ms + (idx / (BM / 4)) * LDAS + (idx % (BM / 4)) * 4) = rm[i];   // LDAS % 4 == 0
      } else {
        const int row = idx >> 2, kq = (idx & 3) * 4;
        ms[(kq + 0) * LDAS + row] = (unsigned char)rm[i]; ms[(kq + 1) * LDAS + row] = (unsigned char)(rm[i] >> 8);
        ms[(kq + 2) * LDAS + row] = (unsigned char)(rm[i] >> 16); ms[(kq + 3) * LDAS + row] = (unsigned char)(rm[i] >> 24);
      }
    }
    float* bs = Bs + buf * BK * LDBS;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int u = threadIdx.x + 256 * i;
      if (u < units) {
        const int kk = u / (HB * q_h), r2 = u - kk * (HB * q_h), hl = r2 / q_h, c = (r2 - hl * q_h) * 4;
        st4(bs + kk * LDBS + hl * 32 * NTH + c, rb[i]);   // columns >= cols_h of a head stay unwritten: never stored
      }
    }
  };

  const int64_t nkt = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;
  if (nkt > 0) { gload(kbeg); sstore(0); }
  __syncthreads();
  const int fr = lane & 31, fk = lane >> 5;
  for (int64_t kt = 0; kt < nkt; ++kt) {
    const int buf = (int)(kt & 1);
    if (kt + 1 < nkt) gload(kbeg + (kt + 1) * BK);
    const float* as = As + buf * BK * LDAS + fk * LDAS + 32 * w + fr;
    const unsigned char* ms = Ms + buf * BK * LDAS + fk * LDAS + 32 * w + fr;
    const float* bs = Bs + buf * BK * LDBS + fk * LDBS + fr;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float xs = as[kk * LDAS];
      const uint32_t mb = (uint32_t)ms[kk * LDAS] >> hb0;
#pragma unroll
      for (int hl = 0; hl < HB; ++hl) {
        const float a = ((mb >> hl) & 1u) ? xs : 0.f;
#pragma unroll
        for (int nt = 0; nt < NTH; ++nt)
          acc[hl * NTH + nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bs[kk * LDBS + 32 * (hl * NTH + nt)], acc[hl * NTH + nt], 0, 0, 0);
      }
    }
    if (kt + 1 < nkt) sstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int64_t zoff = (int64_t)blockIdx.z * g.c_split_stride;
#pragma unroll
  for (int hl = 0; hl < HB; ++hl) {
    const int h = hb0 + hl;
    if (h >= g.H) continue;
#pragma unroll
    for (int nt = 0; nt < NTH; ++nt) {
      const int c = 32 * nt + fr;
      if (c >= cols_h) continue;
      float* base = c < g.Fp ? g.C1 + zoff + (int64_t)h * g.Fp + c : g.C2 + zoff + (int64_t)h * g.Fp + (c - g.Fp);
      const int64_t ld = c < g.Fp ? g.ldc1 : g.ldc2;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + 32 * w + (r & 3) + 8 * (r >> 2) + 4 * fk;
        if (row < g.M) base[(int64_t)row * ld] = acc[hl * NTH + nt][r] * g.scale;
      }
    }
  }
}

template <bool TA>
static int launch_headmask(const HeadMaskArgs& g, int splits, hipStream_t st) {
  const int cols_h = g.B2 ? 2 * g.Fp : g.Fp;
  const int nth = cols_h <= 32 ? 1 : cols_h <= 64 ? 2 : cols_h <= 128 ? 4 : 8;
  const int hb = 8 / nth;
  if (cols_h > 256 || g.H > 8) return 0;
  dim3 grid((unsigned)cdiv(g.M, BM), (unsigned)cdiv(g.H, hb), (unsigned)splits);
  const size_t lds = (size_t)(2 * BK * (BM + PAD) + 2 * BK * (32 * 8 + PAD)) * sizeof(float) + 2 * BK * (BM + PAD);
  switch (nth) {
    case 1: hipLaunchKernelGGL((gemm_headmask_kernel<TA, 8, 1>), grid, dim3(256), lds, st, g); break;
    case 2: hipLaunchKernelGGL((gemm_headmask_kernel<TA, 4, 2>), grid, dim3(256), lds, st, g); break;
    case 4: hipLaunchKernelGGL((gemm_headmask_kernel<TA, 2, 4>), grid, dim3(256), lds, st, g); break;
    default: hipLaunchKernelGGL((gemm_headmask_kernel<TA, 1, 8>), grid, dim3(256), lds, st, g); break;
  }
  return 1;
}

template <bool TA, bool TB>
static int launch_gemm(const GemmArgs& g, int splits, hipStream_t st) {
  const int nt_needed = (int)cdiv(g.N, 32);
  int NT = nt_needed <= 1 ? 1 : nt_needed <= 2 ? 2 : nt_needed <= 3 ? 3 : nt_needed <= 4 ? 4 : nt_needed <= 5 ? 5 : 4;
  dim3 grid((unsigned)cdiv(g.M, BM), (unsigned)cdiv(g.N, 32 * NT), (unsigned)splits);
  // (32-deep k-tiles for the 128-column tile were measured on the PPI level-2 projection 3144 x 2056 x 1024: 189 us with
  // 16-deep tiles, 219 us with 32-deep ones -- 66 KB of LDS leave two work-groups per CU where 34 KB leave three)
  auto lds = [](int nt, int bk) { return (size_t)2 * bk * ((BM + PAD) + (32 * nt + PAD)) * sizeof(float); };
  switch (NT) {
    case 1: hipLaunchKernelGGL((gemm_f32_kernel<TA, TB, 1, 16>), grid, dim3(256), lds(1, 16), st, g); break;
    case 2: hipLaunchKernelGGL((gemm_f32_kernel<TA, TB, 2, 16>), grid, dim3(256), lds(2, 16), st, g); break;
    case 3: hipLaunchKernelGGL((gemm_f32_kernel<TA, TB, 3, 16>), grid, dim3(256), lds(3, 16), st, g); break;
    case 4: hipLaunchKernelGGL((gemm_f32_kernel<TA, TB, 4, 16>), grid, dim3(256), lds(4, 16), st, g); break;
    default: hipLaunchKernelGGL((gemm_f32_kernel<TA, TB, 5, 16>), grid, dim3(256), lds(5, 16), st, g); break;
  }
  return 0;
}

int try_gemm_smallk(int transB, int M, int N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                    const pygat_out_segments* out, int accumulate, bool split, hipStream_t st, const float* svec = nullptr,
                    int64_t sv_ld = 0, int sv_n = 0, float* s_out = nullptr, int64_t s_ld = 0);
int try_gemm_tn_stream(int M, int N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                       int max_splits, float* ws, bool split, hipStream_t st, int N1, const float* B2, int64_t ldb2);
int try_project_x3(int n, int Fin, int H, int Fp, const float* X, int64_t ldx, const float* Wcat, int64_t ldw, float* Wh,
                   float* s, const float* a_pad, bool split, hipStream_t st);
int try_gemm_x3g(int transA, int transB, int M, int N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                 const pygat_out_segments* out, int accumulate, int splits, int64_t k_per_split, float* ws, hipStream_t st,
                 ColBlocks ab, ColBlocks cb, int* splits_used);
bool gemm_split(int mode);

}  // namespace pygat

using namespace pygat;

extern "C" size_t pygat_gemm_workspace_bytes(int M, int N, int split_k) {
  if (split_k <= 1) return 0;
  return (size_t)split_k * (size_t)M * (size_t)N * sizeof(float);
}

extern "C" int pygat_gemm_f32(int transA, int transB, int M, int N, int64_t K, const float* A, int64_t lda,
                              const float* B, int64_t ldb, const pygat_out_segments* out, int accumulate,
                              int split_k, void* ws, int gemm_mode, void* stream) {
  return pygat_gemm_f32_blocked(transA, transB, M, N, K, A, lda, nullptr, B, ldb, out, nullptr, accumulate, split_k, ws, gemm_mode,
                                stream);
}

extern "C" int pygat_gemm_f32_blocked(int transA, int transB, int M, int N, int64_t K, const float* A, int64_t lda,
                                      const pygat_col_blocks* a_blk, const float* B, int64_t ldb, const pygat_out_segments* out,
                                      const pygat_col_blocks* c_blk, int accumulate, int split_k, void* ws, int gemm_mode,
                                      void* stream) {
  PYGAT_REQUIRE(A && B && out, "gemm: null pointer");
  PYGAT_REQUIRE(gemm_mode >= PYGAT_GEMM_DEFAULT && gemm_mode <= PYGAT_GEMM_FP32_MFMA, "gemm: unknown product mode %d", gemm_mode);
  const bool split = gemm_split(gemm_mode);
  PYGAT_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: bad sizes M=%d N=%d K=%lld", M, N, (long long)K);
  PYGAT_REQUIRE(out->nseg >= 1 && out->nseg <= PYGAT_MAX_SEGMENTS && out->col_start[0] == 0 && out->col_start[out->nseg] == N,
                "gemm: output segments must cover [0,N) (nseg=%d)", out->nseg);
  const bool a_blocked = col_blocks_on(a_blk), c_blocked = col_blocks_on(c_blk);
  for (int s = 0; s < out->nseg; ++s)
    PYGAT_REQUIRE(out->ptr[s] && out->col_start[s + 1] > out->col_start[s] &&
                      (c_blocked || out->ld[s] >= out->col_start[s + 1] - out->col_start[s]),
                  "gemm: bad output segment %d", s);
  PYGAT_REQUIRE((a_blocked ? lda >= a_blk->w : lda >= (transA ? M : K)) && ldb >= (transB ? K : N), "gemm: leading dimension too small");
  PYGAT_REQUIRE(col_blocks_ok(a_blk, transA ? M : K, lda) && (!a_blocked || aligned16(A)),
                "gemm: bad column blocks of A (w a power of two >= 16 dividing the blocked extent; 16-byte aligned base, rows, blocks)");
  PYGAT_REQUIRE(!c_blocked || (col_blocks_ok(c_blk, N, out->ld[0]) && out->nseg == 1 && aligned16(out->ptr[0])),
                "gemm: bad column blocks of C (one segment; w a power of two >= 16 dividing N)");
  if (split_k < 1) split_k = 1;
  PYGAT_REQUIRE(split_k == 1 || ws, "gemm: split_k > 1 needs a workspace");
  PYGAT_REQUIRE(!c_blocked || split_k == 1, "gemm: a column-blocked C takes no split-K");
  const ColBlocks ab = col_blocks_of(a_blk), cb = col_blocks_of(c_blk);
  hipStream_t st = (hipStream_t)stream;
  if (a_blocked || c_blocked) {
    // blocked operands: the general kernels only (the streamed fast paths below address A / C as ordinary matrices)
  } else if (!transA && split_k == 1) {  // tall-skinny, small K: B resident in LDS, A streamed through registers
    int r = try_gemm_smallk(transB, M, N, K, A, lda, B, ldb, out, accumulate, split, st);
    if (r < 0) return r;
    if (r == 1) return PYGAT_OK;
  }
  if (!a_blocked && transA && !transB && split_k > 1) {  // weight gradient: huge K, small M x N, no LDS
    int r = try_gemm_tn_stream(M, N, K, A, lda, B, ldb, split_k, (float*)ws, split, st, N, nullptr, 0);
    if (r < 0) return r;
    if (r >= 1) {
      int64_t tot = (int64_t)M * N;
      hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)cdiv(tot, 64)), dim3(512), 0, st, M, N, r,
                         (const float*)ws, *out, accumulate);
      PYGAT_CHECK_LAUNCH("gemm_splitk_reduce");
      return PYGAT_OK;
    }
  }
  if (split) {   // any layout on the bf16 pipe from exactly split operands (gemm_x3g_kernel); odd shapes fall through
    const int64_t kps16 = cdiv(cdiv(K, split_k), 16) * 16;
    int sp = (int)cdiv(K, kps16);
    const int r = try_gemm_x3g(transA, transB, M, N, K, A, lda, B, ldb, out, accumulate, sp, kps16, (float*)ws, st, ab, cb, &sp);
    if (r < 0) return r;
    if (r == 1) {
      if (sp > 1) {
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)cdiv((int64_t)M * N, 64)), dim3(512), 0, st, M, N, sp,
                           (const float*)ws, *out, accumulate);
        PYGAT_CHECK_LAUNCH("gemm_splitk_reduce");
      }
      return PYGAT_OK;
    }
  }
  GemmArgs g;
  g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.out = *out;
  g.accumulate = accumulate;
  g.ab = ab; g.cb = cb;
  int64_t kps = cdiv(cdiv(K, split_k), BK) * BK;
  int splits = (int)cdiv(K, kps);
  g.k_per_split = kps;
  g.ws = (splits > 1) ? (float*)ws : nullptr;
  g.a_vec = aligned16(A) && (lda % 4 == 0);
  g.b_vec = aligned16(B) && (ldb % 4 == 0);
  if (!transA && !transB) launch_gemm<false, false>(g, splits, st);
  else if (transA && !transB) launch_gemm<true, false>(g, splits, st);
  else if (!transA && transB) launch_gemm<false, true>(g, splits, st);
  else { pygat::set_error("gemm: transA && transB is not built"); return PYGAT_EINVAL; }
  PYGAT_CHECK_LAUNCH("gemm_f32");
  if (splits > 1) {
    int64_t tot = (int64_t)M * N;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)cdiv(tot, 64)), dim3(512), 0, st, M, N, splits,
                       (const float*)ws, *out, accumulate);
    PYGAT_CHECK_LAUNCH("gemm_splitk_reduce");
  }
  return PYGAT_OK;
}

// Projection of one level in one GEMM: [Wh | Sk | s] = X * Wcat[:, :R (+R) + H] -- the H columns behind the
// heads are W_h a_src_h (written by pygat_pack_params), so s_i = Wh_i . a_src (layers.py:60) falls out of
// the same pass.  On the small-K fast path with H <= 8 the s columns do not get MFMA tiles of their own (8 columns
// would occupy a fifth 32-column tile, 20 % of the kernel): the lane that streams row i accumulates them on the VALU
// in the shadow of the MFMAs (gemm_smallk_kernel, SV).  With a_pad given, heads of 8 / 16 columns and Fin = 64 / 128
// (split-bf16 mode) s comes from the accumulators in the epilogue instead, Wh_i . a_src as the reference forms it:
// a head is 8 / 16 LANES of one MFMA tile, its sum four DPP adds per row register (round 1 tried this with
// ds_bpermute reductions over whole tiles, which cost more than the fifth tile they saved).
extern "C" int pygat_project(int n, int Fin, int H, int Fo, const float* X, int64_t ldx, const float* Wcat, int64_t ldw,
                             const float* a_pad, float* Wh, float* Sk, float* s, int split_k, void* ws, int gemm_mode,
                             void* stream) {
  return pygat_project_blocked(n, Fin, H, Fo, X, ldx, nullptr, Wcat, ldw, a_pad, Wh, Sk, s, split_k, ws, gemm_mode, stream);
}

// x_blk: X as the head exchange of the previous level left it (column blocks = ranks, include/pygat_amd.h): the general
// kernels read it in place, all output columns [Wh | Sk | s] from one GEMM
extern "C" int pygat_project_blocked(int n, int Fin, int H, int Fo, const float* X, int64_t ldx, const pygat_col_blocks* x_blk,
                                     const float* Wcat, int64_t ldw, const float* a_pad, float* Wh, float* Sk, float* s,
                                     int split_k, void* ws, int gemm_mode, void* stream) {
  const bool blocked = col_blocks_on(x_blk);
  const int Fp = padded_width(Fo);
  PYGAT_REQUIRE(n > 0 && Fin > 0 && H > 0 && Fp > 0 && X && Wcat && Wh && s, "project: bad arguments");
  PYGAT_REQUIRE(gemm_mode >= PYGAT_GEMM_DEFAULT && gemm_mode <= PYGAT_GEMM_FP32_MFMA, "project: unknown product mode %d", gemm_mode);
  const bool split = gemm_split(gemm_mode);
  const int R = H * Fp, nw = R * (Sk ? 2 : 1), ncols = nw + H;
  PYGAT_REQUIRE(ldw >= nw + 2 * H && (blocked ? ldx >= x_blk->w : ldx >= Fin), "project: leading dimension too small");
  pygat_out_segments seg;
  int k = 0;
  if (!blocked && a_pad && split_k <= 1) {   // heads of 8 / 16 columns, Fin 64 / 128: s = Wh . a_src from the accumulators (k1_gemm_x3.hip)
    const int r = try_project_x3(n, Fin, H, Fp, X, ldx, Wcat, ldw, Wh, s, a_pad, split, (hipStream_t)stream);
    if (r < 0) return r;
    if (r == 1) {
      if (!Sk) return PYGAT_OK;
      seg.nseg = 1; seg.col_start[0] = 0; seg.col_start[1] = R; seg.ptr[0] = Sk; seg.ld[0] = R;
      return pygat_gemm_f32(0, 0, n, R, Fin, X, ldx, Wcat + R, ldw, &seg, 0, 1, nullptr, gemm_mode, stream);
    }
  }
  seg.col_start[0] = 0; seg.ptr[k] = Wh; seg.ld[k] = R; ++k;
  if (Sk) { seg.col_start[k] = R; seg.ptr[k] = Sk; seg.ld[k] = R; ++k; }
  if (!blocked && H <= 8 && split_k <= 1) {   // s on the VALU of the small-K kernel, no MFMA tile for it
    seg.col_start[k] = nw; seg.nseg = k;
    const int r = try_gemm_smallk(0, n, nw, Fin, X, ldx, Wcat, ldw, &seg, 0, split, (hipStream_t)stream, Wcat + nw, ldw, H, s, H);
    if (r < 0) return r;
    if (r == 1) return PYGAT_OK;
  }
  seg.col_start[k] = nw; seg.ptr[k] = s; seg.ld[k] = H; ++k;
  seg.col_start[k] = ncols;
  seg.nseg = k;
  return pygat_gemm_f32_blocked(0, 0, n, ncols, Fin, X, ldx, x_blk, Wcat, ldw, &seg, nullptr, 0, split_k, ws, gemm_mode, stream);
}

// Weight gradient of one level (autograd of layers.py:35,134):  dW_h = X^T dWh_h, all heads in one GEMM.
// With ds != NULL the dWh passed in lacks its ds_i a_src term (pygat_gat_backward_col called with dz_t):
//   X^T (dWh' + ds (x) a_src) = X^T dWh' + (X^T ds) (x) a_src,
// so ds rides along as H extra B columns of the same GEMM (a fifth 32-column MFMA tile on the streamed-K
// path) and the rank-1 terms are added while unpacking -- dWh is never read-modified-written for it.
namespace pygat {
__global__ __launch_bounds__(256) void unpack_wgrad_rank1_kernel(int H, int Fin, int Fo, int Fp, const float* __restrict__ dWc,
                                                                 int64_t ld, int with_ds, const float* __restrict__ a_pad,
                                                                 float* __restrict__ dW) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)H * Fin * Fo) return;
  const int f = (int)(idx % Fo);
  const int k = (int)((idx / Fo) % Fin);
  const int h = (int)(idx / ((int64_t)Fo * Fin));
  float v = dWc[(int64_t)k * ld + h * Fp + f];
  if (with_ds) v = fmaf(dWc[(int64_t)k * ld + H * Fp + h], a_pad[(int64_t)h * 2 * Fp + f], v);
  dW[idx] = v;
}
}  // namespace pygat

extern "C" size_t pygat_wgrad_workspace_bytes(int Fin, int H, int Fo, int split_k) {
  const int Fp = padded_width(Fo);
  if (Fin <= 0 || H <= 0 || Fp <= 0) return 0;
  if (split_k < 1) split_k = 1;
  const size_t ncols = (size_t)H * Fp + (size_t)((H + 3) / 4) * 4;
  return (size_t)(split_k + 1) * (size_t)Fin * ncols * sizeof(float);
}

extern "C" int pygat_wgrad(int n, int Fin, int H, int Fo, const float* X, int64_t ldx, const float* dWh, const float* ds,
                           const float* a_pad, float* dW, int split_k, void* ws, int h_first, int h_count, int gemm_mode,
                           void* stream) {
  return pygat_wgrad_blocked(n, Fin, H, Fo, X, ldx, nullptr, dWh, ds, a_pad, dW, split_k, ws, h_first, h_count, gemm_mode, stream);
}

extern "C" int pygat_wgrad_blocked(int n, int Fin, int H, int Fo, const float* X, int64_t ldx, const pygat_col_blocks* x_blk,
                                   const float* dWh, const float* ds, const float* a_pad, float* dW, int split_k, void* ws,
                                   int h_first, int h_count, int gemm_mode, void* stream) {
  const bool blocked = col_blocks_on(x_blk);
  const int Fp = padded_width(Fo);
  PYGAT_REQUIRE(gemm_mode >= PYGAT_GEMM_DEFAULT && gemm_mode <= PYGAT_GEMM_FP32_MFMA, "wgrad: unknown product mode %d", gemm_mode);
  const bool split = gemm_split(gemm_mode);
  PYGAT_REQUIRE(n > 0 && Fin > 0 && H > 0 && Fp > 0 && X && dWh && dW && ws && (blocked ? ldx >= x_blk->w : ldx >= Fin),
                "wgrad: bad arguments");
  PYGAT_REQUIRE(!ds || a_pad, "wgrad: ds needs a_pad");
  if (h_count == 0 && h_first == 0) h_count = H;
  PYGAT_REQUIRE(h_first >= 0 && h_count > 0 && h_first + h_count <= H, "wgrad: bad head range [%d, +%d) of %d", h_first, h_count, H);
  // a head range: the operands keep the level's width (strides), the GEMM covers the range's columns only
  const int64_t ldd = (int64_t)H * Fp, lds = H;
  dWh += (int64_t)h_first * Fp;
  if (ds) ds += h_first;
  if (a_pad) a_pad += (int64_t)h_first * 2 * Fp;
  dW += (int64_t)h_first * Fin * Fo;
  H = h_count;
  if (split_k < 1) split_k = 1;
  hipStream_t st = (hipStream_t)stream;
  const int R = H * Fp;
  const int ldc = R + ((H + 3) / 4) * 4;            // [dWc (R) | X^T ds (H, padded)]
  float* dWc = (float*)ws;
  float* slabs = dWc + (size_t)Fin * ldc;
  pygat_out_segments seg;
  seg.nseg = 1; seg.col_start[0] = 0; seg.ptr[0] = dWc; seg.ld[0] = ldc;
  bool done = false;
  if (!blocked && ds && split_k > 1 && split && R > 64 && Fin > 64) {
    // split-bf16 mode: the wide part on the split kernel (it takes one B operand), the H columns of ds in a second,
    // narrow streamed-K pass over X -- 0.1 ms of extra reading against 45 % fewer MFMA cycles on the R columns
    seg.col_start[1] = R;
    int r = try_gemm_tn_stream(Fin, R, n, X, ldx, dWh, ldd, split_k, slabs, split, st, R, nullptr, 0);
    if (r < 0) return r;
    if (r >= 1) {
      hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)cdiv((int64_t)Fin * R, 64)), dim3(512), 0, st, Fin, R, r,
                         (const float*)slabs, seg, 0);
      PYGAT_CHECK_LAUNCH("wgrad(reduce)");
      pygat_out_segments seg2;
      seg2.nseg = 1; seg2.col_start[0] = 0; seg2.col_start[1] = H; seg2.ptr[0] = dWc + R; seg2.ld[0] = ldc;
      int64_t sk2 = (int64_t)split_k * (R + H) / H;   // slabs of the narrow pass that fit the same workspace
      if (sk2 > 256) sk2 = 256;
      const int r2 = try_gemm_tn_stream(Fin, H, n, X, ldx, ds, lds, (int)sk2, slabs, split, st, H, nullptr, 0);
      if (r2 < 0) return r2;
      if (r2 >= 1) {
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)cdiv((int64_t)Fin * H, 64)), dim3(512), 0, st, Fin, H, r2,
                           (const float*)slabs, seg2, 0);
        PYGAT_CHECK_LAUNCH("wgrad(reduce ds)");
      } else {
        const int rc = pygat_gemm_f32(1, 0, Fin, H, n, X, ldx, ds, lds, &seg2, 0, split_k, slabs, gemm_mode, stream);
        if (rc) return rc;
      }
      done = true;
    }
  }
  if (!blocked && !done && ds && split_k > 1) {     // one streamed-K GEMM over [dWh | ds]
    seg.col_start[1] = R + H;
    int r = try_gemm_tn_stream(Fin, R + H, n, X, ldx, dWh, ldd, split_k, slabs, split, st, R, ds, lds);
    if (r < 0) return r;
    if (r >= 1) {
      const int64_t tot = (int64_t)Fin * (R + H);
      hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)cdiv(tot, 64)), dim3(512), 0, st, Fin, R + H, r,
                         (const float*)slabs, seg, 0);
      PYGAT_CHECK_LAUNCH("wgrad(reduce)");
      done = true;
    }
  }
  if (!done) {                                      // any shape: the general path, once per operand
    seg.col_start[1] = R;
    int rc = pygat_gemm_f32_blocked(1, 0, Fin, R, n, X, ldx, x_blk, dWh, ldd, &seg, nullptr, 0, split_k, slabs, gemm_mode, stream);
    if (rc) return rc;
    if (ds) {
      seg.col_start[1] = H; seg.ptr[0] = dWc + R;
      rc = pygat_gemm_f32_blocked(1, 0, Fin, H, n, X, ldx, x_blk, ds, lds, &seg, nullptr, 0, split_k, slabs, gemm_mode, stream);
      if (rc) return rc;
    }
  }
  hipLaunchKernelGGL(unpack_wgrad_rank1_kernel, dim3((unsigned)cdiv((int64_t)H * Fin * Fo, 256)), dim3(256), 0, st, H, Fin,
                     Fo, Fp, (const float*)dWc, (int64_t)ldc, ds ? 1 : 0, a_pad, dW);
  PYGAT_CHECK_LAUNCH("wgrad(unpack)");
  return PYGAT_OK;
}


// ------------------------------------------------------------------------------------------------------------
// Projection / weight gradient under per-head input dropout, all heads in one launch, X read once (see
// gemm_headmask_kernel).  bits [n x Fin] bytes: bit h = head h keeps x[i,k] (pygat_dropout_bits); p = drop rate.
// Supported: H <= 8, (skip ? 2 : 1) * Fp <= 256 -- the callers fall back to the wide-operand path otherwise.
extern "C" int pygat_headmask_supported(int H, int Fo, int skip) {
  const int Fp = padded_width(Fo);
  return (H >= 1 && H <= 8 && Fp > 0 && Fp * (skip ? 2 : 1) <= 256) ? 1 : 0;
}

/* [Wh | Sk] = (1/(1-p)) (X .* m_h) [W_h | Wskip_h] per head; Wcat as written by pygat_pack_params.
 * split_k > 1: K slabs over Fin (a small graph has few 128-row tiles: Cora 22), partial sums in ws
 * (>= pygat_project_dropout_workspace_bytes), summed in slab order. */
extern "C" size_t pygat_project_dropout_workspace_bytes(int n, int H, int Fo, int skip, int split_k) {
  const int Fp = padded_width(Fo);
  if (n <= 0 || H <= 0 || Fp <= 0 || split_k <= 1) return 0;
  return (size_t)split_k * (size_t)n * (size_t)(H * Fp * (skip ? 2 : 1)) * sizeof(float);
}

extern "C" int pygat_project_dropout(int n, int Fin, int H, int Fo, const float* X, int64_t ldx, const unsigned char* bits,
                                     float p, const float* Wcat, int64_t ldw, float* Wh, float* Sk, int split_k, void* ws,
                                     void* stream) {
  const int Fp = padded_width(Fo);
  PYGAT_REQUIRE(n > 0 && Fin > 0 && X && bits && Wcat && Wh && ldx == Fin, "project_dropout: bad arguments (X must be dense: ldx == Fin)");
  PYGAT_REQUIRE(pygat_headmask_supported(H, Fo, Sk != nullptr), "project_dropout: unsupported H=%d F'=%d", H, Fo);
  PYGAT_REQUIRE(p >= 0.f && p <= 1.f, "project_dropout: p=%g outside [0,1]", (double)p);
  if (narrow_takes(Fin, H, Fo, Sk != nullptr))       // 64-feature levels: weights in registers (k10_narrow.hip), no slabs
    return narrow_project(n, Fin, H, Fo, X, ldx, bits, p, Wcat, ldw, Wh, Sk, (hipStream_t)stream);
  if (split_k < 1) split_k = 1;
  PYGAT_REQUIRE(split_k == 1 || ws, "project_dropout: split_k > 1 needs a workspace");
  const int R = H * Fp, ntot = R * (Sk ? 2 : 1);
  hipStream_t st = (hipStream_t)stream;
  const int64_t kps = cdiv(cdiv(Fin, split_k), BK) * BK;
  const int splits = (int)cdiv(Fin, kps);
  HeadMaskArgs g;
  g.M = n; g.K = Fin; g.A = X; g.lda = ldx; g.bits = bits;
  g.B1 = Wcat; g.ldb1 = ldw; g.B2 = Sk ? Wcat + R : nullptr; g.ldb2 = ldw;
  if (splits > 1) {
    g.C1 = (float*)ws; g.ldc1 = ntot; g.C2 = (float*)ws + R; g.ldc2 = ntot; g.c_split_stride = (int64_t)n * ntot;
  } else {
    g.C1 = Wh; g.ldc1 = R; g.C2 = Sk; g.ldc2 = R; g.c_split_stride = 0;
  }
  g.H = H; g.Fp = Fp; g.scale = p < 1.f ? 1.f / (1.f - p) : 0.f; g.k_per_split = kps;
  launch_headmask<false>(g, splits, st);
  PYGAT_CHECK_LAUNCH("project_dropout");
  if (splits > 1) {
    pygat_out_segments seg;
    seg.nseg = Sk ? 2 : 1; seg.col_start[0] = 0; seg.ptr[0] = Wh; seg.ld[0] = R; seg.col_start[1] = R;
    if (Sk) { seg.ptr[1] = Sk; seg.ld[1] = R; seg.col_start[2] = 2 * R; }
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)cdiv((int64_t)n * ntot, 64)), dim3(512), 0, st, n, ntot,
                       splits, (const float*)ws, seg, 0);
    PYGAT_CHECK_LAUNCH("project_dropout(reduce)");
  }
  return PYGAT_OK;
}

extern "C" size_t pygat_wgrad_dropout_workspace_bytes(int Fin, int H, int Fo, int skip, int split_k) {
  const int Fp = padded_width(Fo);
  if (Fin <= 0 || H <= 0 || Fp <= 0) return 0;
  if (split_k < 1) split_k = 1;
  return (size_t)split_k * (size_t)Fin * (size_t)(H * Fp * (skip ? 2 : 1)) * sizeof(float);
}

/* dWc [Fin x R (+R)] = (1/(1-p)) (X .* m_h)^T [dWh_h | Gp_h] per head (columns h*Fp.. of each half); Gp = NULL: no
 * skip half.  split_k K slabs over the nodes, partial sums in ws, summed in slab order (deterministic). */
extern "C" int pygat_wgrad_dropout(int n, int Fin, int H, int Fo, const float* X, int64_t ldx, const unsigned char* bits,
                                   float p, const float* dWh, const float* Gp, int64_t ldgp, float* dWc, int split_k,
                                   void* ws, void* stream) {
  const int Fp = padded_width(Fo);
  PYGAT_REQUIRE(n > 0 && Fin > 0 && X && bits && dWh && dWc && ldx == Fin, "wgrad_dropout: bad arguments");
  PYGAT_REQUIRE(pygat_headmask_supported(H, Fo, Gp != nullptr), "wgrad_dropout: unsupported H=%d F'=%d", H, Fo);
  if (split_k < 1) split_k = 1;
  PYGAT_REQUIRE(split_k == 1 || ws, "wgrad_dropout: split_k > 1 needs a workspace");
  if (narrow_takes(Fin, H, Fo, Gp != nullptr)) {     // k10_narrow.hip: the slabs are one wave's rows each, always through ws
    PYGAT_REQUIRE(ws, "wgrad_dropout: the narrow kernels need the workspace (pygat_wgrad_dropout_workspace_bytes)");
    return narrow_wgrad(n, Fin, H, Fo, X, ldx, bits, p, dWh, Gp, ldgp, dWc, split_k, ws, (hipStream_t)stream);
  }
  const int R = H * Fp, ntot = R * (Gp ? 2 : 1);
  hipStream_t st = (hipStream_t)stream;
  int64_t kps = cdiv(cdiv(n, split_k), BK) * BK;
  const int splits = (int)cdiv(n, kps);
  float* out = splits > 1 ? (float*)ws : dWc;
  HeadMaskArgs g;
  g.M = Fin; g.K = n; g.A = X; g.lda = ldx; g.bits = bits;
  g.B1 = dWh; g.ldb1 = R; g.B2 = Gp; g.ldb2 = ldgp;
  g.C1 = out; g.ldc1 = ntot; g.C2 = out + R; g.ldc2 = ntot; g.c_split_stride = (int64_t)Fin * ntot;
  g.H = H; g.Fp = Fp; g.scale = p < 1.f ? 1.f / (1.f - p) : 0.f; g.k_per_split = kps;
  launch_headmask<true>(g, splits, st);
  PYGAT_CHECK_LAUNCH("wgrad_dropout");
  if (splits > 1) {
    pygat_out_segments seg;
    seg.nseg = 1; seg.col_start[0] = 0; seg.col_start[1] = ntot; seg.ptr[0] = dWc; seg.ld[0] = ntot;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)cdiv((int64_t)Fin * ntot, 64)), dim3(512), 0, st, Fin, ntot,
                       splits, (const float*)ws, seg, 0);
    PYGAT_CHECK_LAUNCH("wgrad_dropout(reduce)");
  }
  return PYGAT_OK;
}
