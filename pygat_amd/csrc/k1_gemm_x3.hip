// The two streamed fp32 GEMMs of a GAT level (projection Wh = h W, layers.py:35,134; weight gradient dW = X^T dWh, its
// autograd) with their products moved from the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32: 64 flop/clk/SIMD) to the bf16
// one (v_mfma_f32_32x32x16_bf16: 1024 flop/clk/SIMD) WITHOUT giving up fp32 operands:
//
//   every fp32 value is cut exactly into three bf16 pieces, x = hi + mid + lo -- hi the top 8 significant bits of x
//   (truncation, so all pieces share x's sign), mid the top 8 of the remainder x - hi (a subtraction fp32 does
//   exactly), lo what is left (<= 8 bits): 8 + 8 + 8 = the 24 significant bits of an fp32.  A product x * y is then
//   the sum of the nine piece products, each of which an fp32 holds exactly (8 x 8 bits), and the MFMA adds them
//   into the same fp32 accumulators the fp32 MFMA would use.  No bit of an operand is dropped; only the order of the
//   fp32 additions differs from the fp32-MFMA kernels (tests/test_gpu_gemm_split.py prices both against fp64).
//   Nine bf16 MFMAs of 32 cycles replace eight fp32 MFMAs of 64 per 32 x 32 x 16 block: 288 cycles instead of 512,
//   which is what lets these two GEMMs (20 flop/B) run at the HBM rate instead of the fp32-MFMA rate.
//   (Non-finite inputs: inf - inf in the split gives NaN where the fp32 kernels give +-inf.)
//
// The split costs ~5.5 VALU operations per element and is done where the element is already in registers.
#include "gemm_fast.h"
#include <stdlib.h>
#include <type_traits>

namespace pygat {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct U4 { uint32_t v[4]; };
struct Frag3 { U4 h, m, l; };   // 8 bf16 per piece: one A or B operand of v_mfma_f32_32x32x16_bf16

// two fp32 -> their pieces, element 0 in the low half-word (v_perm_b32 picks bytes 2,3 of each source)
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  const uint32_t u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
  h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
  const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
  const uint32_t v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
  m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
  const float q0 = r0 - __uint_as_float(v0 & 0xffff0000u), q1 = r1 - __uint_as_float(v1 & 0xffff0000u);
  l = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
}
__device__ __forceinline__ Frag3 split8(float x0, float x1, float x2, float x3, float x4, float x5, float x6, float x7) {
  Frag3 f;
  split_pair(x0, x1, f.h.v[0], f.m.v[0], f.l.v[0]);
  split_pair(x2, x3, f.h.v[1], f.m.v[1], f.l.v[1]);
  split_pair(x4, x5, f.h.v[2], f.m.v[2], f.l.v[2]);
  split_pair(x6, x7, f.h.v[3], f.m.v[3], f.l.v[3]);
  return f;
}
__device__ __forceinline__ Frag3 split8v(const float4& a, const float4& b) { return split8(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w); }
__device__ __forceinline__ void split_one(float x, uint16_t& h, uint16_t& m, uint16_t& l) {
  const uint32_t u = __float_as_uint(x);
  h = (uint16_t)(u >> 16);
  const float r = x - __uint_as_float(u & 0xffff0000u);
  const uint32_t v = __float_as_uint(r);
  m = (uint16_t)(v >> 16);
  const float q = r - __uint_as_float(v & 0xffff0000u);
  l = (uint16_t)(__float_as_uint(q) >> 16);
}

template <class TA_, class TB_>
__device__ __forceinline__ f32x16 mfma_bf16(const TA_& a, const TB_& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// pair j (0..3) of the eight floats (a, b) of one step
__device__ __forceinline__ void pair_of(const float4& a, const float4& b, int j, float& x0, float& x1) {
  if (j == 0) { x0 = a.x; x1 = a.y; } else if (j == 1) { x0 = a.z; x1 = a.w; } else if (j == 2) { x0 = b.x; x1 = b.y; } else { x0 = b.z; x1 = b.w; }
}
// PYGAT_DIAG_K1 (tools/build_variant.sh only; never set in the shipped library): bit 0 no output stores, bit 1 no s
// reduction, bit 2 A rows folded onto 256 rows (cache-resident reads), bit 3 one piece product instead of nine --
// decomposes the projection kernel's time into its store / reduction / HBM-read / MFMA shares.
#ifndef PYGAT_DIAG_K1
#define PYGAT_DIAG_K1 0
#endif
#ifndef PYGAT_K1_LDS_ROWS
#define PYGAT_K1_LDS_ROWS 1   // the projection's epilogue stores whole lines through LDS patches (0: 16-byte stores of 32 rows each)
#endif
// all nine piece products of one 32 x 32 x 16 block, small terms first
template <class TB_>
__device__ __forceinline__ f32x16 mma9(const Frag3& a, const TB_& bh, const TB_& bm, const TB_& bl, f32x16 c) {
#if (PYGAT_DIAG_K1 & 8)
  return mfma_bf16(a.h, bh, c);
#endif
  c = mfma_bf16(a.l, bl, c);
  c = mfma_bf16(a.l, bm, c);
  c = mfma_bf16(a.m, bl, c);
  c = mfma_bf16(a.l, bh, c);
  c = mfma_bf16(a.h, bl, c);
  c = mfma_bf16(a.m, bm, c);
  c = mfma_bf16(a.m, bh, c);
  c = mfma_bf16(a.h, bm, c);
  c = mfma_bf16(a.h, bh, c);
  return c;
}

// The same nine products with the operand ROLES swapped: the weight fragment is the MFMA's A operand, the streamed row
// fragment its B operand (both are "lane (i, h) holds k = 8 h .. + 7 of row / column i": nothing is loaded differently), so
// the 32 x 32 tile comes out TRANSPOSED -- register q of lane (r, h) is C[row r][column (q & 3) + 8 (q >> 2) + 4 h]: a lane
// holds 16 columns of ONE row, four at a time consecutive.  What that buys the epilogue of gemm_smallk_x3_kernel (round 4):
// 16-byte stores (16 per 32 x 128 tile and wave instead of 64 dword stores), and the per-head sums s = Wh . a_src as eight
// in-lane FMAs + one half-wave swap per head instead of a four-step DPP reduction per accumulator register.
template <class TB_>
__device__ __forceinline__ f32x16 mma9t(const Frag3& x, const TB_& wh, const TB_& wm, const TB_& wl, f32x16 c) {
#if (PYGAT_DIAG_K1 & 8)
  return mfma_bf16(wh, x.h, c);
#endif
  c = mfma_bf16(wl, x.l, c);
  c = mfma_bf16(wm, x.l, c);
  c = mfma_bf16(wl, x.m, c);
  c = mfma_bf16(wh, x.l, c);
  c = mfma_bf16(wl, x.h, c);
  c = mfma_bf16(wm, x.m, c);
  c = mfma_bf16(wh, x.m, c);
  c = mfma_bf16(wm, x.h, c);
  c = mfma_bf16(wh, x.h, c);
  return c;
}
// sum of a value over the two half-waves (lanes l and l ^ 32), in every lane: one v_permlane32_swap + one add
__device__ __forceinline__ float half_wave_sum(float x) {
  const uint32_t u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// ---------------------------------------------------------------------------------------------------------------
// C[M x N] = A[M x K] op(B), K <= 256, M huge: same structure as gemm_smallk_kernel (persistent 512-thread
// work-groups, op(B) staged once into LDS, A streamed row-per-lane straight into registers, waves run free), with
//   * op(B) held as three bf16 images [column][k] (k contiguous: a B fragment = 8 consecutive k of one column is one
//     ds_read_b128; rows padded by 8 elements = 16 B, which spreads 16 consecutive columns over all banks);
//   * lane (i = l&31, h = l>>5) loading k = 16 q + 8 h .. + 7 of row i per 16-deep step: the A fragment of the bf16
//     MFMA, and no address is fetched by both half-waves any more;
//   * the s columns (SV) on the VALU as before, each lane over its half of k, the halves added once per row tile.
//   * SPC = 1 / 2 (K = 64 / 128: a row tile is exactly one turn of the four-chunk register ring, a chunk SPC steps of
//     16 k): the tile loop is straight-line code with an unconditional epilogue and a peeled first tile.  vmcnt counts
//     loads and stores together in issue order, and hipcc merges wait counts conservatively at every join: behind a
//     CONDITIONAL epilogue (SPC = 0, any K % 32 == 0) the first chunk of the next tile -- loaded long before -- waits
//     for the tile's 64 stores to be acknowledged, 22 % of the kernel.
//     In that loop (round 4) a wave takes its 32-row tiles from an LDS counter of the work-group, one tile ahead (with a fixed
//     16 tiles per wave the 8 waves ended 177 .. 248 us into a 261 us launch), and a full tile leaves through a per-wave LDS
//     patch as whole 128-byte lines; the products are formed with the weight fragment as the MFMA's A operand, so a lane holds
//     16 columns of one row (mma9t).
//   * SM = 0: no s columns; 1: SV; 8 / 16: SR, s from the accumulators for heads of 8 / 16 columns (SmallKArgs).
//   * PYGAT_DIAG_K1 (tools/build_variant.sh only): 1 no stores, 2 no s, 4 cache-resident rows, 8 one product of nine,
//     16 s_memtime / s_memrealtime stamps per wave (read back with pygat_diag_k1_stamps: tools/k1_stamps.py).
#if (PYGAT_DIAG_K1 & 16)
// diagnostic builds only (tools/build_variant.sh): the stamps of the last launch, 8 words per wave of at most 2048 work-groups x 8
__device__ unsigned long long k1_stamps[2048 * 8 * 8];
#endif
template <bool TB, int NT, int SM, int SPC>
__global__ __launch_bounds__(512) void gemm_smallk_x3_kernel(SmallKArgs g) {
  constexpr bool SV = (SM == 1);
  constexpr int SRF = (SM >= 4) ? SM : 0;
  constexpr int BN = 32 * NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_x3[];
  const int KP = g.K + 8;
  uint16_t* Bimg = reinterpret_cast<uint16_t*>(smem_x3);          // [3][BN][KP]
  const int IMG = BN * KP;
  float* Us = reinterpret_cast<float*>(Bimg + 3 * IMG);            // [K][8]: svec, zero padded (SV only)
  float* Asr = Us + (SV ? g.K * 8 : 0);                            // [BN]: a_src of every output column of this block (SR only)
  float* Stg = Asr + (SRF > 0 ? BN : 0) + 4;                       // [8 waves][32][36]: the epilogue's row patches (lds_rows), behind the 4 turn words
  const int n0 = blockIdx.y * BN;
  unsigned long long st_entry = 0; (void)st_entry;
  if ((PYGAT_DIAG_K1 & 16)) st_entry = __builtin_amdgcn_s_memrealtime();
  // op(B) into its three bf16 images: a thread takes 8 consecutive k of one column -- eight loads in flight (coalesced over the
  // columns of the lanes when B is [k][n]), one ds_write_b128 per image.  (Rounds 2-3: one element per thread and turn, 32
  // dependent load -> split -> three 2-byte LDS writes per thread: stamps showed a wave's tile loop spanning 236 of the
  // kernel's 293 us.)
  for (int item = threadIdx.x; item < (g.K >> 3) * BN; item += 512) {
    const int n = item % BN, k8 = item / BN;
    const bool ok = n0 + n < g.N;
    float v[8];
    if constexpr (!TB) {
      const float* bp = g.B + (int64_t)(8 * k8) * g.ldb + (ok ? n0 + n : 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = bp[(int64_t)j * g.ldb];
    } else {
      const float* bp = g.B + (int64_t)(ok ? n0 + n : 0) * g.ldb + 8 * k8;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = bp[j];
    }
    uint32_t ph[4], pm[4], pl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint16_t h0, m0, l0, h1, m1, l1;
      split_one(ok ? v[2 * j] : 0.f, h0, m0, l0);
      split_one(ok ? v[2 * j + 1] : 0.f, h1, m1, l1);
      ph[j] = (uint32_t)h0 | ((uint32_t)h1 << 16); pm[j] = (uint32_t)m0 | ((uint32_t)m1 << 16); pl[j] = (uint32_t)l0 | ((uint32_t)l1 << 16);
    }
    uint16_t* d = Bimg + n * KP + 8 * k8;
    *reinterpret_cast<uint4*>(d) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
    *reinterpret_cast<uint4*>(d + IMG) = make_uint4(pm[0], pm[1], pm[2], pm[3]);
    *reinterpret_cast<uint4*>(d + 2 * IMG) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
  }
  if constexpr (SV) {
    for (int idx = threadIdx.x; idx < g.K * 8; idx += 512) {
      const int k = idx >> 3, h = idx & 7;
      Us[idx] = (h < g.sv_n) ? g.svec[(int64_t)k * g.sv_ld + h] : 0.f;
    }
  }
  if constexpr (SRF > 0) {
    for (int n = threadIdx.x; n < BN; n += 512) {
      const int col = n0 + n;
      Asr[n] = (col < g.N) ? g.sr_a[(col / SRF) * 2 * SRF + (col % SRF)] : 0.f;
    }
  }
  if (threadIdx.x < 4) reinterpret_cast<int*>(Asr + (SRF > 0 ? BN : 0))[threadIdx.x] = 0;   // word 0: the wave-tile counter
  __syncthreads();

  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int nchunks = g.K / 32;
  if ((int)blockIdx.x >= g.tiles_m) return;
  const int my_tiles = (g.tiles_m - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_tiles * nchunks;  // 32-deep chunks this wave streams, tile after tile

  auto chunk_ptr = [&](int c) -> const float* {   // clamped: every load is unconditional
    if (c > total - 1) c = total - 1;
    const int t = c / nchunks, kc = c - t * nchunks;
    int64_t row = ((int64_t)blockIdx.x + (int64_t)t * gridDim.x) * 256 + 32 * w + fr;
    if (row > g.M - 1) row = g.M - 1;
    return g.A + row * g.lda + kc * 32 + 8 * fh;
  };

  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float sacc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) sacc[i] = 0.f;
  const bool sv_on = SV && blockIdx.y == 0;

  // The accumulators hold the tile TRANSPOSED (mma9t): register 4 g + j of lane (fr, fh) is row row0 + fr, column
  // colt + 8 g + 4 fh + j of the 32-column tile starting at colt -- four consecutive columns per register quad.
  // FAST: the caller knows the wave's 32 rows lie inside M, N is a multiple of 32, accumulate is off, the output takes
  // 16-byte stores (vec_out), s too (s_vec) and the LDS holds the row patches (lds_rows) -- no branch at all (a branch is a
  // join, and behind a join hipcc waits for the stores: SPC above)
  auto store_tile = [&](int64_t row0, auto fast_tag) {   // row0: the first of the wave's 32 rows
    constexpr bool FAST = decltype(fast_tag)::value;
    const int64_t row = row0 + fr;
    const bool row_ok = FAST || row < g.M;
    const bool full = FAST || row0 + 32 <= g.M;  // wave-uniform
    if constexpr (SV) {
      if (sv_on) {
#pragma unroll
        for (int i = 0; i < 8; ++i) sacc[i] += __shfl_xor(sacc[i], 32);
        if (fh == 0 && row_ok) {
          float* so = g.s_out + row * g.s_ld;
          if (FAST && g.sv_n == 8) {   // (wave-uniform)
#pragma unroll
            for (int i = 0; i < 8; ++i) so[i] = sacc[i];
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
              if (i < g.sv_n) so[i] = sacc[i];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) sacc[i] = 0.f;
    }
    // s[row, head] = sum over the head's SRF columns of C[row, col] a_src[col]: this lane holds 4 (SRF = 8) or 8 (SRF = 16)
    // of them, lane ^ 32 the others.  Taken tile by tile, in front of the tile's stores (its registers are free after them);
    // four finished heads go out as one 16-byte store of the half-wave (group index & 1).
    constexpr int SRD = SRF > 0 ? SRF : 32;                 // (no division by a zero template argument in the SRF = 0 instantiations)
    constexpr int HPT = 32 / SRD, GPH = SRF > 0 ? SRF / 8 : 1;   // heads per 32-column tile, register quads per head
    float hs[4] = {0.f, 0.f, 0.f, 0.f};
    float keep[4] = {0.f, 0.f, 0.f, 0.f};   // FAST: the even four-head group of a pair, held back for one store of the pair
    (void)keep;
    float* so = (SRF > 0) ? g.s_out + row * g.s_ld + n0 / SRD : nullptr;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if constexpr (SRF > 0 && !(PYGAT_DIAG_K1 & 2)) {
#pragma unroll
        for (int hh = 0; hh < HPT; ++hh) {
          float p = 0.f;
#pragma unroll
          for (int gg = 0; gg < GPH; ++gg) {
            const int gq = hh * GPH + gg;
            const float4 av = ld4(Asr + 32 * nt + 8 * gq + 4 * fh);   // (LDS broadcast: two addresses per wave)
            p = fmaf(acc[nt][4 * gq], av.x, p); p = fmaf(acc[nt][4 * gq + 1], av.y, p);
            p = fmaf(acc[nt][4 * gq + 2], av.z, p); p = fmaf(acc[nt][4 * gq + 3], av.w, p);
          }
          const int hi = nt * HPT + hh;          // head index inside this block column
          hs[hi & 3] = half_wave_sum(p);
          if constexpr (FAST) {
            // (FAST implies s_vec and NT HPT % 4 == 0.)  No store under a lane mask -- that is a branch around the store, and
            // a join for the wait counts: half-wave 0 keeps the even group of a pair, half-wave 1 the odd one, and every lane
            // stores 16 bytes once per pair (a last group without a partner: both half-waves store it, same bytes).
            if ((hi & 3) == 3) {
              const int grp = hi >> 2;
              if ((grp & 1) == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) keep[q] = hs[q];
                if (hi == NT * HPT - 1) st4(so + 4 * grp, make_float4(keep[0], keep[1], keep[2], keep[3]));
              } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) keep[q] = fh ? hs[q] : keep[q];
                st4(so + 4 * (grp - 1) + 4 * fh, make_float4(keep[0], keep[1], keep[2], keep[3]));
              }
            }
          } else if ((hi & 3) == 3) {                   // heads hi - 3 .. hi are complete
            if (g.s_vec) {                       // (wave-uniform)
              if (row_ok && fh == ((hi >> 2) & 1)) st4(so + (hi - 3), make_float4(hs[0], hs[1], hs[2], hs[3]));
            } else if (row_ok && fh == 0) {
              so[hi - 3] = hs[0]; so[hi - 2] = hs[1]; so[hi - 1] = hs[2]; so[hi] = hs[3];
            }
          } else if (hi == NT * HPT - 1) {        // a last group of fewer than four heads (NT * HPT not a multiple of 4)
            if (row_ok && fh == 0) {
#pragma unroll
              for (int q = 0; q <= (hi & 3); ++q) so[hi - (hi & 3) + q] = hs[q];
            }
          }
        }
      }
      const int colt = n0 + 32 * nt;   // (wave-uniform)
      if constexpr (FAST) {
        if (!(PYGAT_DIAG_K1 & 1)) {
          int64_t ld;
          float* base = out_segment(g.out, colt, ld);   // vec_out: no segment border inside a 32-column tile
          // (opaque to the optimiser: the row offsets of every column tile are loop invariants, and hoisted out of the
          // straight-line tile loop they cost more registers than the kernel has)
          asm volatile("" : "+v"(ld));
          // The tile through the wave's own LDS patch [32 rows][36 floats] (LDS operations of one wave execute in order: no
          // barrier), read back so that 8 lanes hold one row's 128 bytes: a store instruction then writes 8 WHOLE lines
          // instead of 32 bytes of each of 32 rows (same-lease A/B, gpurun_out r4u: projection 0.301-0.311 -> 0.274-0.277 ms).
          float* patch = Stg + w * (32 * 36);
#pragma unroll
          for (int gq = 0; gq < 4; ++gq)
            st4(patch + fr * 36 + 8 * gq + 4 * fh, make_float4(acc[nt][4 * gq], acc[nt][4 * gq + 1], acc[nt][4 * gq + 2], acc[nt][4 * gq + 3]));
          float* bq = base + (row0 + (lane >> 3)) * ld + 4 * (lane & 7);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            st4(bq + (int64_t)8 * q * ld, ld4(patch + ((lane >> 3) + 8 * q) * 36 + 4 * (lane & 7)));
        }
      } else {
      const bool vec = full && !g.accumulate && g.vec_out && colt + 32 <= g.N;
      if ((PYGAT_DIAG_K1 & 1) ? (g.M == -12345) : vec) {
        int64_t ld;
        float* base = out_segment(g.out, colt, ld) + row * ld + 4 * fh;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          st4(base + 8 * gq, make_float4(acc[nt][4 * gq], acc[nt][4 * gq + 1], acc[nt][4 * gq + 2], acc[nt][4 * gq + 3]));
      } else if (!(PYGAT_DIAG_K1 & 1)) {   // (every segment starts at a multiple of 4 columns: try_gemm_smallk_x3)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int col4 = colt + 8 * gq + 4 * fh;
          int64_t ld;
          float* p = out_segment(g.out, col4, ld) + row * ld;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (row_ok && col4 + j < g.N) {
              if (g.accumulate) p[j] += acc[nt][4 * gq + j]; else p[j] = acc[nt][4 * gq + j];
            }
        }
      }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    }
  };

  // one chunk = k 32 kc .. + 31 of the wave's rows: R0,R1 = k 8 fh .. + 7 (step 0), R2,R3 = k 16 + 8 fh .. (step 1)
#define PYGAT_X3_LOAD(R, P)                                                                   \
  {                                                                                           \
    const float* p__ = (P);                                                                   \
    R##0 = ld4(p__); R##1 = ld4(p__ + 4); R##2 = ld4(p__ + 16); R##3 = ld4(p__ + 20);         \
  }
#define PYGAT_X3_SACC(X, KK)                                                                  \
  {                                                                                           \
    const float4 ua = ld4(us + (KK) * 8), ub = ld4(us + (KK) * 8 + 4);                        \
    sacc[0] = fmaf((X), ua.x, sacc[0]); sacc[1] = fmaf((X), ua.y, sacc[1]);                   \
    sacc[2] = fmaf((X), ua.z, sacc[2]); sacc[3] = fmaf((X), ua.w, sacc[3]);                   \
    sacc[4] = fmaf((X), ub.x, sacc[4]); sacc[5] = fmaf((X), ub.y, sacc[5]);                   \
    sacc[6] = fmaf((X), ub.z, sacc[6]); sacc[7] = fmaf((X), ub.w, sacc[7]);                   \
  }
#define PYGAT_X3_SACC8(V0, V1, KK)                                                            \
  PYGAT_X3_SACC((V0).x, (KK) + 0) PYGAT_X3_SACC((V0).y, (KK) + 1) PYGAT_X3_SACC((V0).z, (KK) + 2) PYGAT_X3_SACC((V0).w, (KK) + 3) \
  PYGAT_X3_SACC((V1).x, (KK) + 4) PYGAT_X3_SACC((V1).y, (KK) + 5) PYGAT_X3_SACC((V1).z, (KK) + 6) PYGAT_X3_SACC((V1).w, (KK) + 7)
  // B fragments of the next (step, column tile) are read from LDS before the nine MFMAs of the current one issue;
  // the sched_barriers keep hipcc from sinking the reads to their first use.
#define PYGAT_X3_BREAD(DST, S, NTI)                                                           \
  {                                                                                           \
    const uint16_t* q__ = bb + (NTI) * 32 * KP + (S) * 16;                                    \
    DST[0] = *reinterpret_cast<const uint4*>(q__);                                            \
    DST[1] = *reinterpret_cast<const uint4*>(q__ + IMG);                                      \
    DST[2] = *reinterpret_cast<const uint4*>(q__ + 2 * IMG);                                  \
  }
#define PYGAT_X3_BODY(R, KOFF, NSTEP)                                                         \
  {                                                                                           \
    const uint16_t* bb = Bimg + fr * KP + (KOFF) + 8 * fh;                                    \
    if constexpr (SV) if (sv_on) {                                                            \
      const float* us = Us + ((KOFF) + 8 * fh) * 8;                                           \
      PYGAT_X3_SACC8(R##0, R##1, 0)                                                           \
      if constexpr ((NSTEP) == 2) { PYGAT_X3_SACC8(R##2, R##3, 16) }                          \
    }                                                                                         \
    Frag3 af[2];                                                                              \
    af[0] = split8v((R##0), (R##1));                                                          \
    if constexpr ((NSTEP) == 2) af[1] = split8v((R##2), (R##3));                              \
    uint4 bf[2][3];                                                                           \
    PYGAT_X3_BREAD(bf[0], 0, 0)                                                               \
    _Pragma("unroll") for (int i__ = 0; i__ < (NSTEP) * NT; ++i__) {                          \
      const int s__ = i__ / NT, nt__ = i__ % NT;                                              \
      if (i__ + 1 < (NSTEP) * NT) PYGAT_X3_BREAD(bf[(i__ + 1) & 1], (i__ + 1) / NT, (i__ + 1) % NT) \
      __builtin_amdgcn_sched_barrier(0);                                                      \
      acc[nt__] = mma9t(af[s__], bf[i__ & 1][0], bf[i__ & 1][1], bf[i__ & 1][2], acc[nt__]);  \
      __builtin_amdgcn_sched_barrier(0);                                                      \
    }                                                                                         \
  }
#define PYGAT_X3_STEP(R, CIDX)                                                                \
  {                                                                                           \
    const int c__ = (CIDX);                                                                   \
    const int t__ = c__ / nchunks, kc__ = c__ - t__ * nchunks;                                \
    PYGAT_X3_BODY(R, kc__ * 32, 2)                                                            \
    if (kc__ == nchunks - 1) store_tile(((int64_t)blockIdx.x + (int64_t)t__ * gridDim.x) * 256 + 32 * w, std::false_type{}); \
  }

  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3, rc0, rc1, rc2, rc3, rd0, rd1, rd2, rd3;
  if constexpr (SPC == 0) {
    // (the sched_barriers pin the ISSUE ORDER of the prologue: vmcnt counts in issue order, and hipcc, free to issue
    // chunk 2 first, then has to wait for all three chunks at the top of every iteration instead of the oldest one)
    PYGAT_X3_LOAD(ra, chunk_ptr(0))
    __builtin_amdgcn_sched_barrier(0);
    PYGAT_X3_LOAD(rb, chunk_ptr(1))
    __builtin_amdgcn_sched_barrier(0);
    PYGAT_X3_LOAD(rc, chunk_ptr(2))
    __builtin_amdgcn_sched_barrier(0);
    for (int c = 0; c < total; c += 4) {
      PYGAT_X3_LOAD(rd, chunk_ptr(c + 3))
      PYGAT_X3_STEP(ra, c)
      PYGAT_X3_LOAD(ra, chunk_ptr(c + 4))
      if (c + 1 < total) PYGAT_X3_STEP(rb, c + 1)
      PYGAT_X3_LOAD(rb, chunk_ptr(c + 5))
      if (c + 2 < total) PYGAT_X3_STEP(rc, c + 2)
      PYGAT_X3_LOAD(rc, chunk_ptr(c + 6))
      if (c + 3 < total) PYGAT_X3_STEP(rd, c + 3)
    }
  } else {
    constexpr int CW = 16 * SPC;   // k per chunk; K == 4 * CW
    // Wave tiles (32 rows) are handed out DYNAMICALLY inside the work-group: wave tile q = row tile q >> 3 of this work-group,
    // rows 32 (q & 7) .. + 31 of it, taken from an LDS counter one tile ahead.  With a fixed 16 tiles per wave the ends of the
    // 8 waves of a work-group lay 177 .. 248 us after the kernel's start (100 MHz stamps, kernel 261 us): of the two waves that
    // share a SIMD one is served first and runs ahead, and once it has finished the other issues its MFMAs alone, at ~70 %.
    const int total_q = 8 * my_tiles;
    auto q_row0 = [&](int q) -> int64_t {
      return ((int64_t)blockIdx.x + (int64_t)(q >> 3) * gridDim.x) * 256 + 32 * (q & 7);
    };
    auto tile_ptr = [&](int q, int chunk) -> const float* {   // chunk 0..3 of wave tile q, clamped to the work-group's last
      if (q > total_q - 1) q = total_q - 1;
      int64_t row = q_row0(q) + fr;
      if (row > g.M - 1) row = g.M - 1;
      if (PYGAT_DIAG_K1 & 4) row &= 255;
      return g.A + row * g.lda + chunk * CW + 8 * fh;
    };
    int* q_next = reinterpret_cast<int*>(Asr + (SRF > 0 ? BN : 0));   // (zeroed above)
    const uint32_t q_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int*)q_next;
    auto grab = [&]() -> int {   // lane 0 adds one, the others nothing: lane 0's return value is this wave's alone
      // (as one instruction: the compiler's atomic optimiser turns the builtin with a per-lane value into a loop over the lanes)
      int r;
      asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(q_addr), "v"(lane == 0 ? 1 : 0) : "memory");
      return r;
    };
#define PYGAT_X3_LOADC(R, P)                                                                  \
  {                                                                                           \
    const float* p__ = (P);                                                                   \
    R##0 = ld4(p__); R##1 = ld4(p__ + 4);                                                     \
    if constexpr (SPC == 2) { R##2 = ld4(p__ + 16); R##3 = ld4(p__ + 20); }                   \
  }
    // One 16-deep step, software-pipelined: while the 9 NT MFMAs of the current step (operand AC) issue, the NEXT step's
    // eight floats (SRC0, SRC1) are split into AN, a quarter per column tile, and the s columns take their FMAs of the
    // current step's floats (CUR0, CUR1) -- in program order BETWEEN the MFMAs (sched_group_barrier): an MFMA holds
    // the SIMD's issue port for 8 of its 32 cycles, and both waves of a SIMD run the same stream in step, so VALU work
    // left in a block of its own leaves the MFMA pipe idle for its whole length.
    constexpr int PPG = (4 + NT - 1) / NT;   // pairs of floats handled per column-tile group
    constexpr int VPM = ((11 + (SV ? 16 : 0)) * PPG + 8) / 9;
#define PYGAT_X3_PSTEP(SIDX, AC, AN, SRC0, SRC1, CUR0, CUR1, KOFF, KNEXT)                     \
  _Pragma("unroll") for (int i__ = 0; i__ < NT; ++i__) {                                      \
    constexpr int par__ = 0;                                                                  \
    (void)par__;                                                                              \
    const int cur__ = ((SIDX) * NT + i__) & 1;                                                \
    {                                                                                         \
      const int nk__ = (i__ + 1 < NT) ? (KOFF) : (KNEXT), nn__ = (i__ + 1 < NT) ? i__ + 1 : 0; \
      const uint16_t* q__ = Bimg + fr * KP + nk__ + 8 * fh + nn__ * 32 * KP;                  \
      bfP[cur__ ^ 1][0] = *reinterpret_cast<const uint4*>(q__);                               \
      bfP[cur__ ^ 1][1] = *reinterpret_cast<const uint4*>(q__ + IMG);                         \
      bfP[cur__ ^ 1][2] = *reinterpret_cast<const uint4*>(q__ + 2 * IMG);                     \
    }                                                                                         \
    _Pragma("unroll") for (int j__ = i__ * PPG; j__ < (i__ + 1) * PPG && j__ < 4; ++j__) {    \
      float x0__, x1__;                                                                       \
      pair_of((SRC0), (SRC1), j__, x0__, x1__);                                               \
      split_pair(x0__, x1__, AN.h.v[j__], AN.m.v[j__], AN.l.v[j__]);                          \
      if constexpr (SV) if (sv_on) {                                                          \
        pair_of((CUR0), (CUR1), j__, x0__, x1__);                                             \
        const float* us = Us + ((KOFF) + 8 * fh + 2 * j__) * 8;                               \
        PYGAT_X3_SACC(x0__, 0) PYGAT_X3_SACC(x1__, 1)                                         \
      }                                                                                       \
    }                                                                                         \
    acc[i__] = mma9t(AC, bfP[cur__][0], bfP[cur__][1], bfP[cur__][2], acc[i__]);              \
    _Pragma("unroll") for (int m__ = 0; m__ < 9; ++m__) {                                     \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                      \
      __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);                                    \
      if (m__ < (SV ? 7 : 3)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);              \
    }                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                        \
  }
#if (PYGAT_DIAG_K1 & 16)   /* diagnostic builds only: shader-clock stamps around a tile's MFMA phase and its epilogue */
#define PYGAT_X3_STAMP(V) { __builtin_amdgcn_sched_barrier(0); V = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define PYGAT_X3_STAMP(V)
#endif
    unsigned long long st_a = 0, st_b = 0, st_c = 0, st_mfma = 0, st_epi = 0, st_first = 0, st_real = 0, st_n = 0;
    (void)st_n; (void)st_a; (void)st_b; (void)st_c; (void)st_mfma; (void)st_epi; (void)st_first; (void)st_real;
    if ((PYGAT_DIAG_K1 & 16)) st_real = __builtin_amdgcn_s_memrealtime();   // the constant 100 MHz counter beside the shader clock
#define PYGAT_X3_TILE(FAST)                                                                   \
  {                                                                                           \
    PYGAT_X3_STAMP(st_a)                                                                      \
    PYGAT_X3_LOADC(rd, tile_ptr(qc, 3))                                                       \
    const int qn = __builtin_amdgcn_readfirstlane(qn_v);                                      \
    if constexpr (SPC == 2) {                                                                 \
      PYGAT_X3_PSTEP(0, afA, afB, ra2, ra3, ra0, ra1, 0, 16)                                  \
      PYGAT_X3_PSTEP(1, afB, afA, rb0, rb1, ra2, ra3, 16, 32)                                 \
      PYGAT_X3_LOADC(ra, tile_ptr(qn, 0))                                                     \
      PYGAT_X3_PSTEP(2, afA, afB, rb2, rb3, rb0, rb1, 32, 48)                                 \
      PYGAT_X3_PSTEP(3, afB, afA, rc0, rc1, rb2, rb3, 48, 64)                                 \
      PYGAT_X3_LOADC(rb, tile_ptr(qn, 1))                                                     \
      PYGAT_X3_PSTEP(4, afA, afB, rc2, rc3, rc0, rc1, 64, 80)                                 \
      PYGAT_X3_PSTEP(5, afB, afA, rd0, rd1, rc2, rc3, 80, 96)                                 \
      PYGAT_X3_LOADC(rc, tile_ptr(qn, 2))                                                     \
      PYGAT_X3_PSTEP(6, afA, afB, rd2, rd3, rd0, rd1, 96, 112)                                \
      PYGAT_X3_PSTEP(7, afB, afA, ra0, ra1, rd2, rd3, 112, 0)                                 \
    } else {                                                                                  \
      PYGAT_X3_PSTEP(0, afA, afB, rb0, rb1, ra0, ra1, 0, 16)                                  \
      PYGAT_X3_LOADC(ra, tile_ptr(qn, 0))                                                     \
      PYGAT_X3_PSTEP(1, afB, afA, rc0, rc1, rb0, rb1, 16, 32)                                 \
      PYGAT_X3_LOADC(rb, tile_ptr(qn, 1))                                                     \
      PYGAT_X3_PSTEP(2, afA, afB, rd0, rd1, rc0, rc1, 32, 48)                                 \
      PYGAT_X3_LOADC(rc, tile_ptr(qn, 2))                                                     \
      PYGAT_X3_PSTEP(3, afB, afA, ra0, ra1, rd0, rd1, 48, 0)                                  \
    }                                                                                         \
    PYGAT_X3_STAMP(st_b)                                                                      \
    store_tile(q_row0(qc), FAST);                                                             \
    qc = qn;                                                                                  \
    qn_v = grab();                                                                            \
    PYGAT_X3_STAMP(st_c)                                                                      \
    if ((PYGAT_DIAG_K1 & 16)) { if (!st_first) st_first = st_a; st_mfma += st_b - st_a; st_epi += st_c - st_b; ++st_n; }   \
  }
    int qc = __builtin_amdgcn_readfirstlane(grab());
    int qn_v = grab();
    PYGAT_X3_LOADC(ra, tile_ptr(qc, 0))
    __builtin_amdgcn_sched_barrier(0);
    PYGAT_X3_LOADC(rb, tile_ptr(qc, 1))
    __builtin_amdgcn_sched_barrier(0);
    PYGAT_X3_LOADC(rc, tile_ptr(qc, 2))
    __builtin_amdgcn_sched_barrier(0);
    Frag3 afA = split8v(ra0, ra1), afB;
    uint4 bfP[2][3];
    {
      const uint16_t* q__ = Bimg + fr * KP + 8 * fh;
      bfP[0][0] = *reinterpret_cast<const uint4*>(q__);
      bfP[0][1] = *reinterpret_cast<const uint4*>(q__ + IMG);
      bfP[0][2] = *reinterpret_cast<const uint4*>(q__ + 2 * IMG);
    }
    // only the work-group's last row tile can reach past M: its wave tiles with all 32 rows inside M, and every wave tile of
    // the row tiles before it, take the branch-free epilogue (accumulate calls use the any-K kernel: launch_smallk_x3).
    // (N % 32 != 0: a column tile is cut by N, every tile takes the guarded epilogue)
    const int64_t last_base = ((int64_t)blockIdx.x + (int64_t)(my_tiles - 1) * gridDim.x) * 256;
    int full_last = (int)((g.M - last_base) / 32);
    full_last = full_last < 0 ? 0 : (full_last > 8 ? 8 : full_last);
    constexpr bool FASTABLE = SRF == 0 || ((NT * 32 / (SRF > 0 ? SRF : 32)) % 4) == 0;   // whole four-head groups of s
    const int fast_q = (!FASTABLE || (g.N % 32) != 0 || !g.vec_out || !g.lds_rows || (SRF > 0 && !g.s_vec)) ? 0 : 8 * (my_tiles - 1) + full_last;
    if (qc < fast_q) {   // (peeled: the first tile's operands come from the loads above)
      PYGAT_X3_TILE(std::true_type{})
      while (qc < fast_q) PYGAT_X3_TILE(std::true_type{})
    }
    while (qc < total_q) PYGAT_X3_TILE(std::false_type{})
#if (PYGAT_DIAG_K1 & 16)
    if (lane == 0 && blockIdx.x < 2048) {
      unsigned long long* o = k1_stamps + ((int64_t)blockIdx.x * 8 + w) * 8;
      o[0] = st_c - st_first; o[1] = st_mfma; o[2] = st_epi; o[3] = st_n;
      o[4] = st_entry; o[5] = st_real; o[6] = __builtin_amdgcn_s_memrealtime(); o[7] = 0;   // 100 MHz stamps: entry, first tile, end
    }
#endif
#undef PYGAT_X3_STAMP
#undef PYGAT_X3_LOADC
#undef PYGAT_X3_TILE
#undef PYGAT_X3_PSTEP
  }
#undef PYGAT_X3_LOAD
#undef PYGAT_X3_SACC
#undef PYGAT_X3_SACC8
#undef PYGAT_X3_BREAD
#undef PYGAT_X3_STEP
#undef PYGAT_X3_BODY
}

template <bool TB, int SM, int SPC>
static hipError_t launch_smallk_x3(const SmallKArgs& g, int NT, dim3 grid, size_t lds, hipStream_t st) {
  int dev = -1;
  (void)hipGetDevice(&dev);
#define PYGAT_X3_LAUNCH(n)                                                                                \
  case n: {                                                                                               \
    static bool attr_set[64] = {};   /* per device: the attribute belongs to the device's code object */ \
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_smallk_x3_kernel<TB, n, SM, SPC>),    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                  \
      if (dev >= 0 && dev < 64) attr_set[dev] = true;                                                     \
    }                                                                                                     \
    hipLaunchKernelGGL((gemm_smallk_x3_kernel<TB, n, SM, SPC>), grid, dim3(512), lds, st, g);             \
  } break;
  switch (NT) {
    PYGAT_X3_LAUNCH(1)
    PYGAT_X3_LAUNCH(2)
    PYGAT_X3_LAUNCH(3)
    PYGAT_X3_LAUNCH(4)
    default:
    PYGAT_X3_LAUNCH(5)
  }
#undef PYGAT_X3_LAUNCH
  return hipGetLastError();
}

// s on the VALU (SV) in the pipelined loop: one or two column tiles only (with more the s accumulators spill)
template <int SPC>
static hipError_t launch_smallk_x3_sv(const SmallKArgs& g, int NT, dim3 grid, size_t lds, hipStream_t st) {
  int dev = -1;
  (void)hipGetDevice(&dev);
#define PYGAT_X3_LAUNCH(n)                                                                                \
  {                                                                                                       \
    static bool attr_set[64] = {};                                                                        \
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_smallk_x3_kernel<false, n, 1, SPC>),  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                  \
      if (dev >= 0 && dev < 64) attr_set[dev] = true;                                                     \
    }                                                                                                     \
    hipLaunchKernelGGL((gemm_smallk_x3_kernel<false, n, 1, SPC>), grid, dim3(512), lds, st, g);           \
  }
  if (NT == 1) PYGAT_X3_LAUNCH(1) else PYGAT_X3_LAUNCH(2)
#undef PYGAT_X3_LAUNCH
  return hipGetLastError();
}

int try_gemm_smallk_x3(const SmallKArgs& g_in, int NT, dim3 grid, hipStream_t st) {
  SmallKArgs g = g_in;
  const size_t lds = (size_t)3 * (32 * NT) * (g.K + 8) * sizeof(uint16_t) + (g.svec ? (size_t)g.K * 8 * sizeof(float) : 0) +
                     (g.sr_a ? (size_t)32 * NT * sizeof(float) : 0) + 16;   // (+ the four turn words of the MFMA token)
  const size_t lds_patches = (size_t)8 * 32 * 36 * sizeof(float);
  // 16-byte stores of four consecutive columns: every segment starts at a multiple of 32 columns (a 32-column tile lies in one
  // segment) and has 16-byte aligned rows
  g.vec_out = 1;
  for (int q = 0; q < g.out.nseg; ++q) {
    if ((g.out.col_start[q] % 4) != 0) return 0;   // a lane's four consecutive columns lie in one segment
    if ((g.out.col_start[q] % 32) != 0 || (g.out.ld[q] % 4) != 0 || !aligned16(g.out.ptr[q])) g.vec_out = 0;
  }
  // whole-line stores through LDS patches where the patches fit beside the B images
  g.lds_rows = (PYGAT_K1_LDS_ROWS && g.vec_out && lds + lds_patches <= 150 * 1024) ? 1 : 0;
  g.s_vec = (g.s_out && aligned16(g.s_out) && (g.s_ld % 4) == 0 && g.sr_fp > 0 && ((32 * NT / g.sr_fp) % 4) == 0) ? 1 : 0;
  if (lds > 150 * 1024 || (g.svec && NT == 5)) return 0;
  const size_t ldsz = lds + (g.lds_rows ? lds_patches : 0);   // (5 tiles + the s accumulators do not fit 256 registers)
  constexpr bool generic = false;
  // s on the VALU (svec) keeps the any-K loop: the pipelined one has no registers left for its accumulators
  // (five column tiles: the pipelined K = 128 loop spills)
  const int spc = (generic || g.accumulate || g.svec || (NT == 5 && g.K == 128)) ? 0 : (g.K == 128 ? 2 : (g.K == 64 ? 1 : 0));
  hipError_t e;
  if (g.sr_a) {
    if (spc == 0 || g.transB || g.svec || (g.sr_fp != 8 && g.sr_fp != 16) || (g.N % (32 * NT)) != 0) return 0;
    if (spc == 2) e = g.sr_fp == 16 ? launch_smallk_x3<false, 16, 2>(g, NT, grid, ldsz, st) : launch_smallk_x3<false, 8, 2>(g, NT, grid, ldsz, st);
    else e = g.sr_fp == 16 ? launch_smallk_x3<false, 16, 1>(g, NT, grid, ldsz, st) : launch_smallk_x3<false, 8, 1>(g, NT, grid, ldsz, st);
  } else if (g.svec) {
    // (one or two column tiles -- the one- and two-head shards of a head-parallel level, 16-wide hidden levels: the straight-line
    // pipelined loop has the registers for the s accumulators there, and the any-K loop's joins cost it its prefetch)
    const int spc_sv = (NT <= 2 && !g.accumulate) ? (g.K == 128 ? 2 : (g.K == 64 ? 1 : 0)) : 0;
    e = spc_sv == 2 ? launch_smallk_x3_sv<2>(g, NT, grid, ldsz, st)
                    : (spc_sv == 1 ? launch_smallk_x3_sv<1>(g, NT, grid, ldsz, st) : launch_smallk_x3<false, 1, 0>(g, NT, grid, ldsz, st));
  } else if (g.transB) {
    e = spc == 2 ? launch_smallk_x3<true, 0, 2>(g, NT, grid, ldsz, st)
                 : (spc == 1 ? launch_smallk_x3<true, 0, 1>(g, NT, grid, ldsz, st) : launch_smallk_x3<true, 0, 0>(g, NT, grid, ldsz, st));
  } else {
    e = spc == 2 ? launch_smallk_x3<false, 0, 2>(g, NT, grid, ldsz, st)
                 : (spc == 1 ? launch_smallk_x3<false, 0, 1>(g, NT, grid, ldsz, st) : launch_smallk_x3<false, 0, 0>(g, NT, grid, ldsz, st));
  }
  if (e != hipSuccess) {
    set_error("gemm_smallk_x3: %s", hipGetErrorString(e));
    return PYGAT_EHIP;
  }
  return 1;
}

int try_project_x3(int n, int Fin, int H, int Fp, const float* X, int64_t ldx, const float* Wcat, int64_t ldw, float* Wh,
                   float* s, const float* a_pad, bool split, hipStream_t st) {
  const int R = H * Fp;
  if (!split || !a_pad || (Fp != 8 && Fp != 16) || (R % 32) != 0 || (Fin != 64 && Fin != 128) || n < 8192) return 0;
  if (!aligned16(X) || (ldx % 4) != 0) return 0;
  const int nt_r = R / 32;
  if (nt_r > 4 && (nt_r % 4) != 0) return 0;   // (whole 128-column tiles only: narrower ones re-read X once per tile)
  const int NT = nt_r <= 4 ? nt_r : 4;
  SmallKArgs g;
  g.M = n; g.N = R; g.K = Fin; g.A = X; g.lda = ldx; g.B = Wcat; g.ldb = ldw; g.accumulate = 0;
  g.out.nseg = 1; g.out.col_start[0] = 0; g.out.col_start[1] = R; g.out.ptr[0] = Wh; g.out.ld[0] = R;
  g.svec = nullptr; g.sv_ld = 0; g.sv_n = 0; g.s_out = s; g.s_ld = H;
  g.tiles_m = (int)cdiv(n, 256); g.transB = 0; g.sr_a = a_pad; g.sr_fp = Fp;
  const int tiles_n = R / (32 * NT);
  int gx = 256 / tiles_n;
  if (gx < 1) gx = 1;
  if (gx > g.tiles_m) gx = g.tiles_m;
  return try_gemm_smallk_x3(g, NT, dim3((unsigned)gx, (unsigned)tiles_n, 1), st);
}


// ---------------------------------------------------------------------------------------------------------------
// C[M x N] = A^T B with A [K x M], B [K x N], K huge (one row per node): dW = X^T dWh.  Both operands are k-strided,
// the bf16 MFMA wants 8 consecutive k per lane: the work-group (4 waves, a 128 x 128 tile of C over one K slab) moves
// 16 k rows of both operands per step through LDS --
//   * thread (kp = tid & 7, c4 = tid >> 3) loads rows k0 + 2 kp, k0 + 2 kp + 1, columns 4 c4 .. + 3 of A and of B with
//     16-byte loads (a wave covers 128 contiguous bytes of 16 rows), three steps ahead, into a register ring;
//   * it splits the (k even, k odd) pair of every column ONCE for the whole work-group (the register-only version
//     split every element in two waves and was bound by those VALU instructions) and writes the three packed pieces
//     to images [column][16 k] with 40-byte rows: the dword writes of a half-wave (8 k pairs x 4 column groups) and
//     the 8-byte fragment reads of a half-wave (32 columns) both touch every bank once.  (48-byte rows read with
//     ds_read_b128 put two column groups of every write on the same banks: a third of the LDS cycles were conflicts);
//   * a wave owns a 64 x 64 block of C: 4 fragments x 3 pieces x 2 ds_read_b64 feed 36 MFMAs;
//   * two LDS stages, one barrier per step; the splits and LDS writes of step i + 1 sit between the MFMAs of step i.
// Needs 16-byte aligned rows (lda, ldb, M, N multiples of 4).
constexpr int TNX_RS = 10;                         // dwords per image row (8 of data + 2: see the kernel's comment)
constexpr int TNX_IMG = 128 * TNX_RS;              // dwords per piece image
constexpr int TNX_STAGE = 6 * TNX_IMG;             // A(h, m, l), B(h, m, l)
__device__ __forceinline__ uint4 ld_frag(const uint32_t* p) {   // 8-byte aligned: two ds_read_b64
  const uint2 a = *reinterpret_cast<const uint2*>(p), b = *reinterpret_cast<const uint2*>(p + 2);
  return make_uint4(a.x, a.y, b.x, b.y);
}

__global__ __launch_bounds__(256) void gemm_tn_x3_kernel(TnArgs g) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_tn[];   // [2 * TNX_STAGE]: 60 KB, two work-groups per CU
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.z * 128;
  const int64_t kbeg = (int64_t)blockIdx.x * g.k_per_split;
  const int64_t kend = (kbeg + g.k_per_split < g.K) ? kbeg + g.k_per_split : g.K;
  const int nsteps = (int)((kend - kbeg + 15) / 16);
  // loader role: columns past M / N read column 0 (their products only reach rows / columns of C never stored)
  const int kp = tid & 7, c4 = tid >> 3;
  const float* la = g.A + ((m0 + 4 * c4 + 3 < g.M) ? m0 + 4 * c4 : 0);
  const float* lb = g.B + ((n0 + 4 * c4 + 3 < g.N) ? n0 + 4 * c4 : 0);
  // rows past kend (the last step of the last slab) read row kend - 1, and A's copy is zeroed
#define PYGAT_TNL_LOAD(R, STEP)                                                               \
  {                                                                                           \
    const int64_t k__ = kbeg + 16 * (int64_t)(STEP) + 2 * kp;   /* (steps past the slab: row kend - 1, zeroed) */ \
    int64_t k0__ = k__ < kend ? k__ : kend - 1, k1__ = k__ + 1 < kend ? k__ + 1 : kend - 1;   \
    if (PYGAT_DIAG_K1 & 4) { k0__ &= 255; k1__ &= 255; }   /* diagnostic builds only: cache-resident operand rows */ \
    R##a0 = ld4(la + k0__ * g.lda); R##a1 = ld4(la + k1__ * g.lda);                           \
    R##b0 = ld4(lb + k0__ * g.ldb); R##b1 = ld4(lb + k1__ * g.ldb);                           \
    R##z0 = k__ < kend ? 1.f : 0.f; R##z1 = k__ + 1 < kend ? 1.f : 0.f;                       \
  }
  // split the pairs (row 2 kp, row 2 kp + 1) of the thread's four columns and store the pieces: dword kp of the
  // column's row in each piece image
#define PYGAT_TNL_PUT(IMGBASE, X0, X1, COL)                                                   \
  {                                                                                           \
    uint32_t h__, m__, l__;                                                                   \
    split_pair((X0), (X1), h__, m__, l__);                                                    \
    uint32_t* q__ = (IMGBASE) + (4 * c4 + (COL)) * TNX_RS + kp;                               \
    q__[0] = h__; q__[TNX_IMG] = m__; q__[2 * TNX_IMG] = l__;                                 \
  }
#define PYGAT_TNL_PUTA(R, STAGE, C, CMP)                                                       \
  PYGAT_TNL_PUT(lds_tn + (STAGE) * TNX_STAGE, R##a0.CMP * R##z0, R##a1.CMP * R##z1, C)
#define PYGAT_TNL_PUTB(R, STAGE, C, CMP)                                                       \
  PYGAT_TNL_PUT(lds_tn + (STAGE) * TNX_STAGE + 3 * TNX_IMG, R##b0.CMP, R##b1.CMP, C)
#define PYGAT_TNL_SPLIT(R, STAGE)                                                             \
  PYGAT_TNL_PUTA(R, STAGE, 0, x) PYGAT_TNL_PUTA(R, STAGE, 1, y) PYGAT_TNL_PUTA(R, STAGE, 2, z) PYGAT_TNL_PUTA(R, STAGE, 3, w) \
  PYGAT_TNL_PUTB(R, STAGE, 0, x) PYGAT_TNL_PUTB(R, STAGE, 1, y) PYGAT_TNL_PUTB(R, STAGE, 2, z) PYGAT_TNL_PUTB(R, STAGE, 3, w)
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // consumer role: wave (w >> 1, w & 1) owns rows 64 (w >> 1) .., columns 64 (w & 1) .. of the tile
  const uint32_t* fa = lds_tn + (64 * (w >> 1) + fr) * TNX_RS + 4 * fh;
  const uint32_t* fb = lds_tn + 3 * TNX_IMG + (64 * (w & 1) + fr) * TNX_RS + 4 * fh;
  // (the fragment reads come FIRST in program order: hipcc cannot tell the two stages apart and keeps LDS reads
  // behind every earlier LDS write -- with the split in front, no MFMA could start before its last write)
#define PYGAT_TNL_READ(STAGE)                                                                 \
  uint4 fq[4][3];                                                                             \
  {                                                                                           \
    const uint32_t* a__ = fa + (STAGE) * TNX_STAGE;                                           \
    const uint32_t* b__ = fb + (STAGE) * TNX_STAGE;                                           \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                           \
      fq[0][p] = ld_frag(a__ + p * TNX_IMG);                                                  \
      fq[1][p] = ld_frag(a__ + p * TNX_IMG + 32 * TNX_RS);                                    \
      fq[2][p] = ld_frag(b__ + p * TNX_IMG);                                                  \
      fq[3][p] = ld_frag(b__ + p * TNX_IMG + 32 * TNX_RS);                                    \
    }                                                                                         \
  }
  // one 32 x 32 tile of the wave's block: its nine MFMAs with a quarter of the next step's split (22 VALU
  // instructions, 6 LDS writes) between them -- small scheduling regions, the whole step in one is beyond what
  // hipcc's group scheduler arranges
#define PYGAT_TNL_QUARTER(TM, TN, PUT0, PUT1)                                                 \
  {                                                                                           \
    PUT0 PUT1                                                                                 \
    Frag3 af__;                                                                               \
    af__.h = __builtin_bit_cast(U4, fq[TM][0]); af__.m = __builtin_bit_cast(U4, fq[TM][1]);   \
    af__.l = __builtin_bit_cast(U4, fq[TM][2]);                                               \
    acc[TM][TN] = mma9(af__, fq[2 + TN][0], fq[2 + TN][1], fq[2 + TN][2], acc[TM][TN]);       \
    _Pragma("unroll") for (int m__ = 0; m__ < 9; ++m__) {                                     \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                      \
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                      \
      if (m__ < 6) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                         \
    }                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                        \
  }
#define PYGAT_TNL_STEP(RSPLIT, RLOAD, I)                                                      \
  {                                                                                           \
    PYGAT_TNL_LOAD(RLOAD, (I) + 3)                                                            \
    PYGAT_TNL_READ((I) & 1)                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    const int sn__ = ((I) + 1) & 1;   /* (past the last step: clamped data into the idle stage) */ \
    PYGAT_TNL_QUARTER(0, 0, PYGAT_TNL_PUTA(RSPLIT, sn__, 0, x), PYGAT_TNL_PUTA(RSPLIT, sn__, 1, y)) \
    PYGAT_TNL_QUARTER(0, 1, PYGAT_TNL_PUTA(RSPLIT, sn__, 2, z), PYGAT_TNL_PUTA(RSPLIT, sn__, 3, w)) \
    PYGAT_TNL_QUARTER(1, 0, PYGAT_TNL_PUTB(RSPLIT, sn__, 0, x), PYGAT_TNL_PUTB(RSPLIT, sn__, 1, y)) \
    PYGAT_TNL_QUARTER(1, 1, PYGAT_TNL_PUTB(RSPLIT, sn__, 2, z), PYGAT_TNL_PUTB(RSPLIT, sn__, 3, w)) \
    __syncthreads();                                                                          \
  }
  float4 r0a0, r0a1, r0b0, r0b1, r1a0, r1a1, r1b0, r1b1, r2a0, r2a1, r2b0, r2b1;
  float r0z0, r0z1, r1z0, r1z1, r2z0, r2z1;
  PYGAT_TNL_LOAD(r0, 0)
  __builtin_amdgcn_sched_barrier(0);   // (issue order = wait order, see gemm_smallk_x3_kernel)
  PYGAT_TNL_LOAD(r1, 1)
  __builtin_amdgcn_sched_barrier(0);
  PYGAT_TNL_LOAD(r2, 2)
  __builtin_amdgcn_sched_barrier(0);
  PYGAT_TNL_SPLIT(r0, 0)
  __syncthreads();
  // step i: MFMAs on stage i & 1; ring slot (i + 1) % 3 is split into stage (i + 1) & 1; slot i % 3 (split during
  // step i - 1) takes the loads of step i + 3
  // (three steps per turn, unconditionally: the one or two steps past the slab multiply zeroed rows of A -- with the two steps
  // under `if (i + k < nsteps)` the loop had joins, and behind them hipcc gave a load the registers of an address still in
  // use: `s_waitcnt vmcnt(0)` at the top of every turn, the loads issued three steps ahead waited for after one)
  for (int i = 0; i < nsteps; i += 3) {
    PYGAT_TNL_STEP(r1, r0, i)
    PYGAT_TNL_STEP(r2, r1, i + 1)
    PYGAT_TNL_STEP(r0, r2, i + 2)
  }
#undef PYGAT_TNL_LOAD
#undef PYGAT_TNL_PUT
#undef PYGAT_TNL_SPLIT
#undef PYGAT_TNL_QUARTER
#undef PYGAT_TNL_PUTA
#undef PYGAT_TNL_PUTB
#undef PYGAT_TNL_READ
#undef PYGAT_TNL_STEP
  float* base = g.ws + (int64_t)blockIdx.x * g.M * g.N;
  const int wm0 = m0 + 64 * (w >> 1), wn0 = n0 + 64 * (w & 1);
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int col = wn0 + 32 * tn + fr;
      if (col >= g.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm0 + 32 * tm + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (row < g.M) base[(int64_t)row * g.N + col] = acc[tm][tn][r];
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same GEMM on v_mfma_f32_16x16x32_bf16 (round 5).  MI355X_MICROARCH.md (DVFS give-back, item 7): on random data the chip
// holds a higher clock under the 16 x 16 x 32 shape than under 32 x 32 x 16 at equal cycles per flop; the nine-product loop of
// this file, LDS-fed, two waves per SIMD, measured 1.08-1.11 x (tools/micro/mfma_shape_x9.hip, profiles/r5*_mfma_shape_x9.txt).
// The 16 x 16 x 32 instruction sums 32 k of one row, so an LDS stage holds 32 k rows instead of 16:
//   * work-group = 8 waves (512 threads), ONE per CU, a 128 x 128 tile of C over one K slab; wave (w >> 1, w & 1) owns rows
//     32 (w >> 1) .., columns 64 (w & 1) ..: 2 x 4 tiles of 16 x 16, eight accumulators of four registers;
//   * stage = 32 k x (128 + 128) columns x three pieces, images [column][32 k] with NO row padding (64-byte rows): the four
//     8-k slices of a row sit rotated by (row >> 1) & 3, which makes the 16-byte fragment reads of every ds_read_b128 lane
//     group (16 lanes: 8 rows of one slice + 8 rows of the next) hit all 64 banks once, and the dword writes of a half-wave
//     2-way at worst (free for ds_write_b32); two stages = 96 KB;
//   * thread (kp = tid & 7, c4 = (tid >> 3) & 31, half = tid >> 8) loads rows k0 + 2 pr, + 1 (pr = kp + 8 half) x 4 columns of
//     both operands (a wave covers 128 contiguous bytes of 16 rows, as above), three steps ahead; one barrier per 32 k --
//     half as many per flop as the 16-k kernel -- and the split work of a step is spread over eight waves instead of four.
// Same-lease A/B at config 5 (gpurun_out r5h, tools/ab_variants.sh): k5_wgrad 0.285-0.302 -> 0.260-0.265 ms, step 3.18 -> 3.15 ms;
// parity tests unchanged (same nine products, fp32 accumulators).  -DPYGAT_TN_WIDE=0 builds the 32 x 32 x 16 kernel above.
#ifndef PYGAT_TN_WIDE
#define PYGAT_TN_WIDE 1
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TNW_RS = 16;                         // dwords per image row: 32 k of bf16
constexpr int TNW_IMG = 128 * TNW_RS;              // dwords per piece image (8 KB)
constexpr int TNW_STAGE = 6 * TNW_IMG;             // A(h, m, l), B(h, m, l): 48 KB
template <class TA_, class TB_>
__device__ __forceinline__ f32x4 mfma16_bf16(const TA_& a, const TB_& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(512) void gemm_tn_x3w_kernel(TnArgs g) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_tw[];   // [2 * TNW_STAGE]: 96 KB, one work-group per CU
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fi = lane & 15, fq = lane >> 4;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.z * 128;
  const int64_t kbeg = (int64_t)blockIdx.x * g.k_per_split;
  const int64_t kend = (kbeg + g.k_per_split < g.K) ? kbeg + g.k_per_split : g.K;
  const int nsteps = (int)((kend - kbeg + 31) / 32);
  // loader role: columns past M / N read column 0 (their products only reach rows / columns of C never stored)
  const int kp = tid & 7, c4 = (tid >> 3) & 31, pr = kp + 8 * (tid >> 8);
  const float* la = g.A + ((m0 + 4 * c4 + 3 < g.M) ? m0 + 4 * c4 : 0);
  const float* lb = g.B + ((n0 + 4 * c4 + 3 < g.N) ? n0 + 4 * c4 : 0);
  // image position of this thread's k pair in column 4 c4 + COL: slice pr >> 2 rotated by (column >> 1) & 3, dword pr & 3
  uint32_t* const wbase = lds_tw + (4 * c4) * TNW_RS + (pr & 3);
  const int wq = pr >> 2, wrot = 2 * (c4 & 1);      // ((4 c4 + COL) >> 1) & 3 = (2 (c4 & 1) + (COL >> 1)) & 3
#define PYGAT_TWL_LOAD(R, STEP)                                                               \
  {                                                                                           \
    const int64_t k__ = kbeg + 32 * (int64_t)(STEP) + 2 * pr;   /* (steps past the slab: row kend - 1, zeroed) */ \
    const int64_t k0__ = k__ < kend ? k__ : kend - 1, k1__ = k__ + 1 < kend ? k__ + 1 : kend - 1;   \
    R##a0 = ld4(la + k0__ * g.lda); R##a1 = ld4(la + k1__ * g.lda);                           \
    R##b0 = ld4(lb + k0__ * g.ldb); R##b1 = ld4(lb + k1__ * g.ldb);                           \
    R##z0 = k__ < kend ? 1.f : 0.f; R##z1 = k__ + 1 < kend ? 1.f : 0.f;                       \
  }
#define PYGAT_TWL_PUT(IMGBASE, X0, X1, COL)                                                   \
  {                                                                                           \
    uint32_t h__, m__, l__;                                                                   \
    split_pair((X0), (X1), h__, m__, l__);                                                    \
    uint32_t* q__ = (IMGBASE) + (wbase - lds_tw) + (COL) * TNW_RS + 4 * ((wq + wrot + ((COL) >> 1)) & 3);   \
    q__[0] = h__; q__[TNW_IMG] = m__; q__[2 * TNW_IMG] = l__;                                 \
  }
#define PYGAT_TWL_PUTA(R, STAGE, C, CMP)                                                       \
  PYGAT_TWL_PUT(lds_tw + (STAGE) * TNW_STAGE, R##a0.CMP * R##z0, R##a1.CMP * R##z1, C)
#define PYGAT_TWL_PUTB(R, STAGE, C, CMP)                                                       \
  PYGAT_TWL_PUT(lds_tw + (STAGE) * TNW_STAGE + 3 * TNW_IMG, R##b0.CMP, R##b1.CMP, C)
#define PYGAT_TWL_SPLIT(R, STAGE)                                                             \
  PYGAT_TWL_PUTA(R, STAGE, 0, x) PYGAT_TWL_PUTA(R, STAGE, 1, y) PYGAT_TWL_PUTA(R, STAGE, 2, z) PYGAT_TWL_PUTA(R, STAGE, 3, w) \
  PYGAT_TWL_PUTB(R, STAGE, 0, x) PYGAT_TWL_PUTB(R, STAGE, 1, y) PYGAT_TWL_PUTB(R, STAGE, 2, z) PYGAT_TWL_PUTB(R, STAGE, 3, w)
  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  // consumer role.  Fragment of row / column c = base + fi (base a multiple of 16): slice fq at dword 4 ((fq + (fi >> 1)) & 3)
  const int fso = 4 * ((fq + (fi >> 1)) & 3);
  const uint32_t* fa = lds_tw + (32 * (w >> 1) + fi) * TNW_RS + fso;
  const uint32_t* fb = lds_tw + 3 * TNW_IMG + (64 * (w & 1) + fi) * TNW_RS + fso;
  // (fragment reads FIRST in program order, as in the kernel above)
#define PYGAT_TWL_READ(STAGE)                                                                 \
  uint4 fqa[2][3], fqb[4][3];                                                                 \
  {                                                                                           \
    const uint32_t* a__ = fa + (STAGE) * TNW_STAGE;                                           \
    const uint32_t* b__ = fb + (STAGE) * TNW_STAGE;                                           \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                           \
      fqa[0][p] = *reinterpret_cast<const uint4*>(a__ + p * TNW_IMG);                         \
      fqa[1][p] = *reinterpret_cast<const uint4*>(a__ + p * TNW_IMG + 16 * TNW_RS);           \
      _Pragma("unroll") for (int t = 0; t < 4; ++t)                                           \
        fqb[t][p] = *reinterpret_cast<const uint4*>(b__ + p * TNW_IMG + 16 * t * TNW_RS);     \
    }                                                                                         \
  }
  // one 16 x 16 tile: its nine MFMAs (small terms first, as mma9) with an eighth of the next step's split (11 VALU
  // instructions, 3 LDS writes) between them
#define PYGAT_TWL_GROUP(TM, TN, PUT0)                                                         \
  {                                                                                           \
    PUT0                                                                                      \
    f32x4 c__ = acc[TM][TN];                                                                  \
    c__ = mfma16_bf16(fqa[TM][2], fqb[TN][2], c__); c__ = mfma16_bf16(fqa[TM][2], fqb[TN][1], c__);   \
    c__ = mfma16_bf16(fqa[TM][1], fqb[TN][2], c__); c__ = mfma16_bf16(fqa[TM][2], fqb[TN][0], c__);   \
    c__ = mfma16_bf16(fqa[TM][0], fqb[TN][2], c__); c__ = mfma16_bf16(fqa[TM][1], fqb[TN][1], c__);   \
    c__ = mfma16_bf16(fqa[TM][1], fqb[TN][0], c__); c__ = mfma16_bf16(fqa[TM][0], fqb[TN][1], c__);   \
    c__ = mfma16_bf16(fqa[TM][0], fqb[TN][0], c__);                                           \
    acc[TM][TN] = c__;                                                                        \
    _Pragma("unroll") for (int m__ = 0; m__ < 9; ++m__) {                                     \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                      \
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                      \
      if (m__ >= 3 && m__ < 6) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);             \
    }                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                        \
  }
#define PYGAT_TWL_STEP(RSPLIT, RLOAD, I)                                                      \
  {                                                                                           \
    PYGAT_TWL_LOAD(RLOAD, (I) + 3)                                                            \
    PYGAT_TWL_READ((I) & 1)                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    const int sn__ = ((I) + 1) & 1;   /* (past the last step: clamped data into the idle stage) */ \
    PYGAT_TWL_GROUP(0, 0, PYGAT_TWL_PUTA(RSPLIT, sn__, 0, x)) PYGAT_TWL_GROUP(0, 1, PYGAT_TWL_PUTA(RSPLIT, sn__, 1, y)) \
    PYGAT_TWL_GROUP(0, 2, PYGAT_TWL_PUTA(RSPLIT, sn__, 2, z)) PYGAT_TWL_GROUP(0, 3, PYGAT_TWL_PUTA(RSPLIT, sn__, 3, w)) \
    PYGAT_TWL_GROUP(1, 0, PYGAT_TWL_PUTB(RSPLIT, sn__, 0, x)) PYGAT_TWL_GROUP(1, 1, PYGAT_TWL_PUTB(RSPLIT, sn__, 1, y)) \
    PYGAT_TWL_GROUP(1, 2, PYGAT_TWL_PUTB(RSPLIT, sn__, 2, z)) PYGAT_TWL_GROUP(1, 3, PYGAT_TWL_PUTB(RSPLIT, sn__, 3, w)) \
    __syncthreads();                                                                          \
  }
  float4 r0a0, r0a1, r0b0, r0b1, r1a0, r1a1, r1b0, r1b1, r2a0, r2a1, r2b0, r2b1;
  float r0z0, r0z1, r1z0, r1z1, r2z0, r2z1;
  PYGAT_TWL_LOAD(r0, 0)
  __builtin_amdgcn_sched_barrier(0);   // (issue order = wait order)
  PYGAT_TWL_LOAD(r1, 1)
  __builtin_amdgcn_sched_barrier(0);
  PYGAT_TWL_LOAD(r2, 2)
  __builtin_amdgcn_sched_barrier(0);
  PYGAT_TWL_SPLIT(r0, 0)
  __syncthreads();
  for (int i = 0; i < nsteps; i += 3) {     // (three steps per turn, unconditionally: see the kernel above)
    PYGAT_TWL_STEP(r1, r0, i)
    PYGAT_TWL_STEP(r2, r1, i + 1)
    PYGAT_TWL_STEP(r0, r2, i + 2)
  }
#undef PYGAT_TWL_LOAD
#undef PYGAT_TWL_PUT
#undef PYGAT_TWL_SPLIT
#undef PYGAT_TWL_GROUP
#undef PYGAT_TWL_PUTA
#undef PYGAT_TWL_PUTB
#undef PYGAT_TWL_READ
#undef PYGAT_TWL_STEP
  // C/D layout of the 16 x 16 tile: column = lane & 15, row = 4 (lane >> 4) + register
  float* base = g.ws + (int64_t)blockIdx.x * g.M * g.N;
  const int wm0 = m0 + 32 * (w >> 1), wn0 = n0 + 64 * (w & 1);
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      const int col = wn0 + 16 * tn + fi;
      if (col >= g.N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wm0 + 16 * tm + 4 * fq + r;
        if (row < g.M) base[(int64_t)row * g.N + col] = acc[tm][tn][r];
      }
    }
}

int try_gemm_tn_x3(const TnArgs& g, int splits, hipStream_t st) {
  if (g.B2 || g.M <= 64 || g.N <= 64 || (g.k_per_split % 16) != 0) return 0;
  if (!aligned16(g.A) || !aligned16(g.B) || (g.lda % 4) != 0 || (g.ldb % 4) != 0 || (g.M % 4) != 0 || (g.N % 4) != 0) return 0;
  int dev = -1;
  (void)hipGetDevice(&dev);
#if PYGAT_TN_WIDE
  {   // the 16 x 16 x 32 kernel: one 8-wave work-group per CU, 32-k steps, three per loop turn
    const int tiles = (int)(cdiv(g.M, 128) * cdiv(g.N, 128));
    int sw = splits;
    if (sw * tiles > 256) sw = 256 / tiles > 0 ? 256 / tiles : 1;
    TnArgs gw = g;
    gw.k_per_split = cdiv(cdiv(g.K, sw), 96) * 96;
    sw = (int)cdiv(g.K, gw.k_per_split);
    constexpr size_t ldsw = 2 * TNW_STAGE * sizeof(uint32_t);
    static bool attr_w[64] = {};
    if (dev < 0 || dev >= 64 || !attr_w[dev]) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_x3w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw);
      if (dev >= 0 && dev < 64) attr_w[dev] = true;
    }
    hipLaunchKernelGGL(gemm_tn_x3w_kernel, dim3((unsigned)sw, (unsigned)cdiv(g.M, 128), (unsigned)cdiv(g.N, 128)), dim3(512), ldsw, st, gw);
    hipError_t ew = hipGetLastError();
    if (ew != hipSuccess) {
      set_error("gemm_tn_x3w: %s", hipGetErrorString(ew));
      return PYGAT_EHIP;
    }
    return sw;
  }
#endif
  dim3 grid((unsigned)splits, (unsigned)cdiv(g.M, 128), (unsigned)cdiv(g.N, 128));
  constexpr size_t lds = 2 * TNX_STAGE * sizeof(uint32_t);
  static bool attr_set[64] = {};   // per device: the attribute belongs to the device's code object
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL(gemm_tn_x3_kernel, grid, dim3(256), lds, st, g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("gemm_tn_x3: %s", hipGetErrorString(e));
    return PYGAT_EHIP;
  }
  return splits;
}

// ---------------------------------------------------------------------------------------------------------------
// The same pipeline for ANY operand layout: C[M x N] = op(A) op(B) with every product formed from the exact three-way
// bf16 split (projections with K > 256 or few rows: the PPI levels' 1024 -> 1024 GEMMs, layers.py:35,134,48,166; their
// input gradients dX = dWh W^T; GATv2's projections; weight gradients the streamed-K kernel above does not take).
// gemm_tn_x3_kernel is the (k-strided, k-strided) case of it and stays as tuned for the headline's dW.
//
// An operand is either K-STRIDED ([K x cols], cols contiguous: A of a transA call, B of a plain one) and staged as
// above -- thread (kp, c4) loads rows 2 kp, 2 kp + 1 x 4 columns and writes the packed (k even, k odd) pieces as dwords --
// or K-CONTIGUOUS ([rows x K]: A of a plain call, B of a transB one): thread (h = (tid >> 4) & 1, row = (tid & 15) +
// 16 (tid >> 5)) loads the 8 floats k0 + 8 h .. + 7 of its row (two 16-byte loads; a row's 16 k of a step are one
// 64-byte sector), splits the four (k, k + 1) pairs and writes each piece's four dwords as two ds_write_b64 into the
// SAME image layout ([row or column][16 k], 40-byte rows): the 16 lanes of a ds_write_b64 group hold 16 consecutive
// rows of one h, 16 x 40 bytes hit the 16 even banks, the pair fills the odd ones -- conflict-free like the dword
// writes of the k-strided stage.  Consumer side, scheduling (split + LDS writes of step i + 1 between the MFMAs of
// step i, 9-MFMA regions), the three-slot load ring and the two LDS stages are those of the kernel above.
//
// Two-level accumulation (see gemm_f32_kernel): every 96 k the running 64 x 64 block is added into a second register
// set.  DIRECT output (ws == nullptr): rows / columns routed to the output segments, optional accumulate; otherwise
// slab blockIdx.x of the split-K workspace.
// Needs 16-byte aligned rows: ld % 4 == 0 for both operands, K % 4 == 0 and 4-aligned slabs for a k-contiguous one,
// the contiguous extent (M or N) of a k-strided one a multiple of 4 or its rows padded to one.
struct X3gArgs {
  int M, N;
  int64_t K;
  const float* A;
  int64_t lda;
  const float* B;
  int64_t ldb;
  int64_t k_per_split;
  float* ws;
  pygat_out_segments out;
  int accumulate;
  ColBlocks ab, cb;   // column blocks of the stored A / of C (common.h; one block = an ordinary matrix)
};

struct X3gSlot {      // one step of both operands, in flight
  float4 a0, a1, b0, b1;
};

template <bool KCA, bool KCB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_x3g_kernel(X3gArgs g) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_xg[];   // [2 * TNX_STAGE]: 60 KB, two work-groups per CU
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.z * 128;
  const int64_t kbeg = (int64_t)blockIdx.x * g.k_per_split;
  const int64_t kend = (kbeg + g.k_per_split < g.K) ? kbeg + g.k_per_split : g.K;
  const int nsteps = (int)((kend - kbeg + 15) / 16);
  // loader roles
  const int kp = tid & 7, c4 = tid >> 3;                                  // k-strided
  const int lh = (tid >> 4) & 1, lrow = (tid & 15) + 16 * (tid >> 5);     // k-contiguous
  const float* la;
  const float* lb;
  if constexpr (KCA) la = g.A + (int64_t)((m0 + lrow < g.M) ? m0 + lrow : g.M - 1) * g.lda;   // rows past M: the last row
  else la = g.A + blk_off((m0 + 4 * c4 < g.M) ? m0 + 4 * c4 : 0, g.ab);   // columns past M: column 0 (a float4 may run up to 3 columns into the row's padding)
  if constexpr (KCB) lb = g.B + (int64_t)((n0 + lrow < g.N) ? n0 + lrow : g.N - 1) * g.ldb;
  else lb = g.B + ((n0 + 4 * c4 < g.N) ? n0 + 4 * c4 : 0);
  // (their products only reach rows / columns of C that are never stored)

  auto load = [&](X3gSlot& r, int step) {
    const int st = step < nsteps ? step : nsteps - 1;
    const int64_t ks = kbeg + 16 * (int64_t)st;
    {   // k-strided addressing (rows 2 kp, 2 kp + 1 of the step)
      const int64_t k = ks + 2 * kp;
      const int64_t k0 = k < kend ? k : kend - 1, k1 = k + 1 < kend ? k + 1 : kend - 1;
      if constexpr (!KCA) { r.a0 = ld4(la + k0 * g.lda); r.a1 = ld4(la + k1 * g.lda); }
      if constexpr (!KCB) { r.b0 = ld4(lb + k0 * g.ldb); r.b1 = ld4(lb + k1 * g.ldb); }
    }
    {   // k-contiguous addressing (floats 8 lh .. + 7 of the step)
      const int64_t k = ks + 8 * lh;
      const int64_t k0 = k < kend ? k : kend - 4, k1 = k + 4 < kend ? k + 4 : kend - 4;
      // (a column-blocked A: the step's 16 k -- and kend - 4 >= ks -- lie in ONE block, blocks being >= 16 wide and slabs
      // starting at multiples of 16: the block offset is a scalar of the step)
      if constexpr (KCA) { const float* las = la + (blk_off(ks, g.ab) - ks); r.a0 = ld4(las + k0); r.a1 = ld4(las + k1); }
      if constexpr (KCB) { r.b0 = ld4(lb + k0); r.b1 = ld4(lb + k1); }
    }
  };
  // half H (0 / 1) of one operand's split + LDS writes: two of the step's eight "put units"
  auto put_strided = [&](uint32_t* img, const float4& x0, const float4& x1, float z0, float z1, auto half_tag) {
    constexpr int HALF = decltype(half_tag)::value;
    float p0, p1, q0, q1;
    if constexpr (HALF == 0) { p0 = x0.x; p1 = x1.x; q0 = x0.y; q1 = x1.y; } else { p0 = x0.z; p1 = x1.z; q0 = x0.w; q1 = x1.w; }
    uint32_t h, m, l;
    split_pair(p0 * z0, p1 * z1, h, m, l);
    uint32_t* d = img + (4 * c4 + 2 * HALF) * TNX_RS + kp;
    d[0] = h; d[TNX_IMG] = m; d[2 * TNX_IMG] = l;
    split_pair(q0 * z0, q1 * z1, h, m, l);
    d += TNX_RS;
    d[0] = h; d[TNX_IMG] = m; d[2 * TNX_IMG] = l;
  };
  auto put_contig = [&](uint32_t* img, const float4& x, float z) {   // one float4 = two (k, k + 1) pairs
    uint32_t h0, m0_, l0, h1, m1, l1;
    split_pair(x.x * z, x.y * z, h0, m0_, l0);
    split_pair(x.z * z, x.w * z, h1, m1, l1);
    return [=](uint32_t* d) {
      *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(d + TNX_IMG) = make_uint2(m0_, m1);
      *reinterpret_cast<uint2*>(d + 2 * TNX_IMG) = make_uint2(l0, l1);
    }(img);
  };
  // A's copy of k positions past the slab is zeroed (only the last step of a slab has any): step index -> 1 / 0 factors
  // of the thread's two rows (k-strided) or two float4 (k-contiguous), recomputed here instead of riding in the ring
  auto put_a = [&](const X3gSlot& r, int stage, int step_of_r, auto half_tag) {
    constexpr int HALF = decltype(half_tag)::value;
    uint32_t* img = lds_xg + stage * TNX_STAGE;
    const int64_t ks = kbeg + 16 * (int64_t)step_of_r;
    if constexpr (KCA) {
      const int64_t k = ks + 8 * lh + 4 * HALF;
      put_contig(img + lrow * TNX_RS + 4 * lh + 2 * HALF, HALF == 0 ? r.a0 : r.a1, k < kend ? 1.f : 0.f);
    } else {
      const int64_t k = ks + 2 * kp;
      put_strided(img, r.a0, r.a1, k < kend ? 1.f : 0.f, k + 1 < kend ? 1.f : 0.f, half_tag);
    }
  };
  auto put_b = [&](const X3gSlot& r, int stage, auto half_tag) {
    constexpr int HALF = decltype(half_tag)::value;
    uint32_t* img = lds_xg + stage * TNX_STAGE + 3 * TNX_IMG;
    if constexpr (KCB) put_contig(img + lrow * TNX_RS + 4 * lh + 2 * HALF, HALF == 0 ? r.b0 : r.b1, 1.f);
    else put_strided(img, r.b0, r.b1, 1.f, 1.f, half_tag);
  };

  f32x16 acc[2][2], acc2[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acc2[i][j][r] = 0.f; }
  // consumer role: wave (w >> 1, w & 1) owns rows 64 (w >> 1) .., columns 64 (w & 1) .. of the tile
  const uint32_t* fa = lds_xg + (64 * (w >> 1) + fr) * TNX_RS + 4 * fh;
  const uint32_t* fb = lds_xg + 3 * TNX_IMG + (64 * (w & 1) + fr) * TNX_RS + 4 * fh;

  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  // one step: global loads of step i + 2, fragment reads of stage i & 1 (FIRST in program order, see the kernel
  // above), then per 32 x 32 tile of the wave's block its nine MFMAs with a quarter of step i + 1's split between them
  auto step = [&](const X3gSlot& rs, X3gSlot& rl, int i) {
    load(rl, i + 2);
    uint4 fq[4][3];
    {
      const uint32_t* a_ = fa + (i & 1) * TNX_STAGE;
      const uint32_t* b_ = fb + (i & 1) * TNX_STAGE;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        fq[0][p] = ld_frag(a_ + p * TNX_IMG);
        fq[1][p] = ld_frag(a_ + p * TNX_IMG + 32 * TNX_RS);
        fq[2][p] = ld_frag(b_ + p * TNX_IMG);
        fq[3][p] = ld_frag(b_ + p * TNX_IMG + 32 * TNX_RS);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    const int sn = (i + 1) & 1;   // (past the last step: clamped data into the idle stage)
    auto quarter = [&](auto tm_tag, auto tn_tag) {
      constexpr int TM = decltype(tm_tag)::value, TN = decltype(tn_tag)::value;
      Frag3 af;
      af.h = __builtin_bit_cast(U4, fq[TM][0]); af.m = __builtin_bit_cast(U4, fq[TM][1]); af.l = __builtin_bit_cast(U4, fq[TM][2]);
      acc[TM][TN] = mma9(af, fq[2 + TN][0], fq[2 + TN][1], fq[2 + TN][2], acc[TM][TN]);
#pragma unroll
      for (int m = 0; m < 9; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        if (m < 6) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    put_a(rs, sn, i + 1, H0{}); quarter(H0{}, H0{});
    put_a(rs, sn, i + 1, H1{}); quarter(H0{}, H1{});
    put_b(rs, sn, H0{}); quarter(H1{}, H0{});
    put_b(rs, sn, H1{}); quarter(H1{}, H1{});
    __syncthreads();
  };

  // two-slot load ring (the streamed-K kernel above keeps three: here the second accumulator set takes the registers,
  // and two co-resident work-groups x two steps x 16 KB are 64 KB in flight per CU): step i splits the slot loaded
  // during step i - 1 and refills the other one with step i + 2
  X3gSlot r0, r1;
  load(r0, 0);
  __builtin_amdgcn_sched_barrier(0);   // (issue order = wait order)
  load(r1, 1);
  __builtin_amdgcn_sched_barrier(0);
  put_a(r0, 0, 0, H0{}); put_a(r0, 0, 0, H1{}); put_b(r0, 0, H0{}); put_b(r0, 0, H1{});
  __syncthreads();
  for (int i = 0; i < nsteps; i += 2) {
    step(r1, r0, i);
    if (i + 1 < nsteps) step(r0, r1, i + 1);
    if (((i >> 1) % 3) == 2) {      // every 6 steps = 96 k
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) { acc2[a][b][r] += acc[a][b][r]; acc[a][b][r] = 0.f; }
    }
  }
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
      else if (g.cb.lw < 62) { base = g.out.ptr[0] + blk_off(col, g.cb); ld = g.out.ld[0]; }
      else base = out_segment(g.out, col, ld);
      const bool add = !g.ws && g.accumulate;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm0 + 32 * tm + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (row < g.M) {
          float* p = base + (int64_t)row * ld;
          const float v = acc[tm][tn][r] + acc2[tm][tn][r];
          *p = add ? *p + v : v;
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// gemm_x3g_kernel on v_mfma_f32_16x16x32_bf16 (round 5; see gemm_tn_x3w_kernel for the why and the LDS image): 8 waves, one
// work-group per CU, 32 k per stage, 128 x 128 tile, wave (w >> 1, w & 1) = rows 32 (w >> 1) .., columns 64 (w & 1) ...
// K-STRIDED operand: thread (kp, c4, half) as in gemm_tn_x3w_kernel.  K-CONTIGUOUS operand: thread (row = (tid & 15) + 16 (tid >> 6),
// slice q = (tid >> 4) & 3) loads the 8 floats k0 + 8 q .. + 7 of its row, splits the four pairs and writes each piece's slice with
// ONE ds_write_b128 (8 consecutive rows of one slice, rotated by (row >> 1) & 3: the 8-lane groups of a b128 write hit 32 banks once).
// Two-level accumulation every 96 k (three steps), output / slabs / column blocks as gemm_x3g_kernel.
// Same-lease A/B (gpurun_out r5i, tools/ab_x3gw.sh; profiles/r5i_x3g_16x16x32_same_lease.txt): PPI level-2 input gradient
// 3144 x 1024 x 1024 (200 tiles) 87.4 -> 69.5 us, level-3 65.4 -> 53.5, weight gradient 84.5 -> 78 (4 slabs instead of 8),
// projection (425 tiles = two rounds of one-per-CU work-groups) 146 -> 140; PPI epoch 2.35 -> 2.14 ms.
// -DPYGAT_X3G_WIDE=0 builds the 32 x 32 x 16 kernel above as the general kernel.
#ifndef PYGAT_X3G_WIDE
#define PYGAT_X3G_WIDE 1
#endif
struct X3gwSlot {      // one 32-k step of both operands, in flight
  float4 a0, a1, b0, b1;
};

template <bool KCA, bool KCB>
__global__ __launch_bounds__(512) void gemm_x3gw_kernel(X3gArgs g) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_xw[];   // [2 * TNW_STAGE]: 96 KB
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fi = lane & 15, fq = lane >> 4;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.z * 128;
  const int64_t kbeg = (int64_t)blockIdx.x * g.k_per_split;
  const int64_t kend = (kbeg + g.k_per_split < g.K) ? kbeg + g.k_per_split : g.K;
  const int nsteps = (int)((kend - kbeg + 31) / 32);
  // loader roles
  const int kp = tid & 7, c4 = (tid >> 3) & 31, pr = kp + 8 * (tid >> 8);   // k-strided: pair pr (rows 2 pr, 2 pr + 1) x columns 4 c4 ..
  const int lq = (tid >> 4) & 3, lrow = (tid & 15) + 16 * (tid >> 6);        // k-contiguous: slice lq of row lrow
  const float* la;
  const float* lb;
  if constexpr (KCA) la = g.A + (int64_t)((m0 + lrow < g.M) ? m0 + lrow : g.M - 1) * g.lda;   // rows past M: the last row
  else la = g.A + blk_off((m0 + 4 * c4 < g.M) ? m0 + 4 * c4 : 0, g.ab);
  if constexpr (KCB) lb = g.B + (int64_t)((n0 + lrow < g.N) ? n0 + lrow : g.N - 1) * g.ldb;
  else lb = g.B + ((n0 + 4 * c4 < g.N) ? n0 + 4 * c4 : 0);

  auto load = [&](X3gwSlot& r, int step) {
    const int st = step < nsteps ? step : nsteps - 1;
    const int64_t ks = kbeg + 32 * (int64_t)st;
    {   // k-strided addressing
      const int64_t k = ks + 2 * pr;
      const int64_t k0 = k < kend ? k : kend - 1, k1 = k + 1 < kend ? k + 1 : kend - 1;
      if constexpr (!KCA) { r.a0 = ld4(la + k0 * g.lda); r.a1 = ld4(la + k1 * g.lda); }
      if constexpr (!KCB) { r.b0 = ld4(lb + k0 * g.ldb); r.b1 = ld4(lb + k1 * g.ldb); }
    }
    {   // k-contiguous addressing (a column-blocked A: an 8-float slice never straddles a block, blocks being >= 16 wide)
      const int64_t k = ks + 8 * lq;
      const int64_t k0 = k < kend ? k : kend - 4, k1 = k + 4 < kend ? k + 4 : kend - 4;
      if constexpr (KCA) { r.a0 = ld4(la + blk_off(k0, g.ab)); r.a1 = ld4(la + blk_off(k1, g.ab)); }
      if constexpr (KCB) { r.b0 = ld4(lb + k0); r.b1 = ld4(lb + k1); }
    }
  };
  // ---- one EIGHTH of a step's split + LDS writes per operand quarter Q (0..3) ----
  // k-strided: column 4 c4 + Q of the thread's pair -> three dword writes
  auto put_strided = [&](uint32_t* img, const float4& x0, const float4& x1, float z0, float z1, auto q_tag) {
    constexpr int Q = decltype(q_tag)::value;
    const float p0 = Q == 0 ? x0.x : Q == 1 ? x0.y : Q == 2 ? x0.z : x0.w;
    const float p1 = Q == 0 ? x1.x : Q == 1 ? x1.y : Q == 2 ? x1.z : x1.w;
    uint32_t h, m, l;
    split_pair(p0 * z0, p1 * z1, h, m, l);
    uint32_t* d = img + (4 * c4 + Q) * TNW_RS + 4 * (((pr >> 2) + 2 * (c4 & 1) + (Q >> 1)) & 3) + (pr & 3);
    d[0] = h; d[TNW_IMG] = m; d[2 * TNW_IMG] = l;
  };
  // k-contiguous: pair Q of the thread's eight floats into the piece registers; the slice is written with pair 3
  auto put_contig = [&](uint32_t* img, const float4& x0, const float4& x1, float z0, float z1, uint32_t (&ph)[4], uint32_t (&pm)[4],
                        uint32_t (&pl)[4], auto q_tag) {
    constexpr int Q = decltype(q_tag)::value;
    const float p0 = Q == 0 ? x0.x * z0 : Q == 1 ? x0.z * z0 : Q == 2 ? x1.x * z1 : x1.z * z1;
    const float p1 = Q == 0 ? x0.y * z0 : Q == 1 ? x0.w * z0 : Q == 2 ? x1.y * z1 : x1.w * z1;
    split_pair(p0, p1, ph[Q], pm[Q], pl[Q]);
    if constexpr (Q == 3) {
      uint32_t* d = img + lrow * TNW_RS + 4 * ((lq + (lrow >> 1)) & 3);
      *reinterpret_cast<uint4*>(d) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
      *reinterpret_cast<uint4*>(d + TNW_IMG) = make_uint4(pm[0], pm[1], pm[2], pm[3]);
      *reinterpret_cast<uint4*>(d + 2 * TNW_IMG) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
    }
  };
  uint32_t pah[4], pam[4], pal[4], pbh[4], pbm[4], pbl[4];   // (k-contiguous operands only)
  // A's copy of k positions past the slab is zeroed (only the last step of a slab has any)
  auto put_a = [&](const X3gwSlot& r, int stage, int step_of_r, auto q_tag) {
    uint32_t* img = lds_xw + stage * TNW_STAGE;
    const int64_t ks = kbeg + 32 * (int64_t)step_of_r;
    if constexpr (KCA) {
      const int64_t k = ks + 8 * lq;
      put_contig(img, r.a0, r.a1, k < kend ? 1.f : 0.f, k + 4 < kend ? 1.f : 0.f, pah, pam, pal, q_tag);
    } else {
      const int64_t k = ks + 2 * pr;
      put_strided(img, r.a0, r.a1, k < kend ? 1.f : 0.f, k + 1 < kend ? 1.f : 0.f, q_tag);
    }
  };
  auto put_b = [&](const X3gwSlot& r, int stage, auto q_tag) {
    uint32_t* img = lds_xw + stage * TNW_STAGE + 3 * TNW_IMG;
    if constexpr (KCB) put_contig(img, r.b0, r.b1, 1.f, 1.f, pbh, pbm, pbl, q_tag);
    else put_strided(img, r.b0, r.b1, 1.f, 1.f, q_tag);
  };

  f32x4 acc[2][4], acc2[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc[i][j][r] = 0.f; acc2[i][j][r] = 0.f; }
  const int fso = 4 * ((fq + (fi >> 1)) & 3);
  const uint32_t* fa = lds_xw + (32 * (w >> 1) + fi) * TNW_RS + fso;
  const uint32_t* fb = lds_xw + 3 * TNW_IMG + (64 * (w & 1) + fi) * TNW_RS + fso;

  using Q0 = std::integral_constant<int, 0>;
  using Q1 = std::integral_constant<int, 1>;
  using Q2 = std::integral_constant<int, 2>;
  using Q3 = std::integral_constant<int, 3>;
  auto step = [&](const X3gwSlot& rs, X3gwSlot& rl, int i) {
    load(rl, i + 2);
    uint4 fqa[2][3], fqb[4][3];
    {
      const uint32_t* a_ = fa + (i & 1) * TNW_STAGE;
      const uint32_t* b_ = fb + (i & 1) * TNW_STAGE;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        fqa[0][p] = *reinterpret_cast<const uint4*>(a_ + p * TNW_IMG);
        fqa[1][p] = *reinterpret_cast<const uint4*>(a_ + p * TNW_IMG + 16 * TNW_RS);
#pragma unroll
        for (int t = 0; t < 4; ++t) fqb[t][p] = *reinterpret_cast<const uint4*>(b_ + p * TNW_IMG + 16 * t * TNW_RS);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    const int sn = (i + 1) & 1;   // (past the last step: clamped data into the idle stage)
    auto group = [&](auto tm_tag, auto tn_tag) {
      constexpr int TM = decltype(tm_tag)::value, TN = decltype(tn_tag)::value;
      f32x4 c = acc[TM][TN];
      c = mfma16_bf16(fqa[TM][2], fqb[TN][2], c); c = mfma16_bf16(fqa[TM][2], fqb[TN][1], c);
      c = mfma16_bf16(fqa[TM][1], fqb[TN][2], c); c = mfma16_bf16(fqa[TM][2], fqb[TN][0], c);
      c = mfma16_bf16(fqa[TM][0], fqb[TN][2], c); c = mfma16_bf16(fqa[TM][1], fqb[TN][1], c);
      c = mfma16_bf16(fqa[TM][1], fqb[TN][0], c); c = mfma16_bf16(fqa[TM][0], fqb[TN][1], c);
      c = mfma16_bf16(fqa[TM][0], fqb[TN][0], c);
      acc[TM][TN] = c;
#pragma unroll
      for (int m = 0; m < 9; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        if (m >= 3 && m < 6) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    put_a(rs, sn, i + 1, Q0{}); group(Q0{}, Q0{});
    put_a(rs, sn, i + 1, Q1{}); group(Q0{}, Q1{});
    put_a(rs, sn, i + 1, Q2{}); group(Q0{}, Q2{});
    put_a(rs, sn, i + 1, Q3{}); group(Q0{}, Q3{});
    put_b(rs, sn, Q0{}); group(Q1{}, Q0{});
    put_b(rs, sn, Q1{}); group(Q1{}, Q1{});
    put_b(rs, sn, Q2{}); group(Q1{}, Q2{});
    put_b(rs, sn, Q3{}); group(Q1{}, Q3{});
    __syncthreads();
  };

  X3gwSlot r0, r1;
  load(r0, 0);
  __builtin_amdgcn_sched_barrier(0);   // (issue order = wait order)
  load(r1, 1);
  __builtin_amdgcn_sched_barrier(0);
  put_a(r0, 0, 0, Q0{}); put_a(r0, 0, 0, Q1{}); put_a(r0, 0, 0, Q2{}); put_a(r0, 0, 0, Q3{});
  put_b(r0, 0, Q0{}); put_b(r0, 0, Q1{}); put_b(r0, 0, Q2{}); put_b(r0, 0, Q3{});
  __syncthreads();
  for (int i = 0; i < nsteps; i += 2) {
    step(r1, r0, i);
    if (i + 1 < nsteps) step(r0, r1, i + 1);
    if (((i >> 1) & 1) == 1) {      // every 4 steps = 128 k (the 16-k kernel: 96)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) { acc2[a][b][r] += acc[a][b][r]; acc[a][b][r] = 0.f; }
    }
  }
  const int wm0 = m0 + 32 * (w >> 1), wn0 = n0 + 64 * (w & 1);
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      const int col = wn0 + 16 * tn + fi;
      if (col >= g.N) continue;
      float* base;
      int64_t ld;
      if (g.ws) { base = g.ws + (int64_t)blockIdx.x * g.M * g.N + col; ld = g.N; }
      else if (g.cb.lw < 62) { base = g.out.ptr[0] + blk_off(col, g.cb); ld = g.out.ld[0]; }
      else base = out_segment(g.out, col, ld);
      const bool add = !g.ws && g.accumulate;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wm0 + 16 * tm + 4 * fq + r;
        if (row < g.M) {
          float* p = base + (int64_t)row * ld;
          const float v = acc[tm][tn][r] + acc2[tm][tn][r];
          *p = add ? *p + v : v;
        }
      }
    }
}

// 1: took the call (slabs, if any, are in ws: the caller reduces them); 0: shape does not qualify; < 0: launch error
int try_gemm_x3g(int transA, int transB, int M, int N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                 const pygat_out_segments* out, int accumulate, int splits, int64_t k_per_split, float* ws, hipStream_t st,
                 ColBlocks ab, ColBlocks cb, int* splits_used) {
  if (splits_used) *splits_used = splits;
  if (M < 64 || N <= 64 || K < 32) return 0;   // (a 64-row operand fills half a tile and still beats the fp32 kernel: PPI level-1 dW 38 -> 24 us)
  if (!aligned16(A) || !aligned16(B) || (lda % 4) != 0 || (ldb % 4) != 0) return 0;
  const bool kca = !transA, kcb = transB != 0;
  if ((kca || kcb) && ((K % 4) != 0 || (k_per_split % 4) != 0)) return 0;
  // a k-strided operand is read four columns at a time: its extent is a multiple of 4, or the rows are padded to one
  // (the projection's Wcat: 2 R + H columns in rows of ldw = the next multiple of 4)
  if ((!kca && (M % 4) != 0 && lda < (M + 3) / 4 * 4) || (!kcb && (N % 4) != 0 && ldb < (N + 3) / 4 * 4)) return 0;
  if (splits > 1 && (!ws || (k_per_split % 16) != 0)) return 0;
  X3gArgs g;
  g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb;
  g.k_per_split = splits > 1 ? k_per_split : K;
  g.ws = splits > 1 ? ws : nullptr;
  g.out = *out; g.accumulate = accumulate;
  g.ab = ab; g.cb = cb;
  if (cb.lw < 62 && splits > 1) return 0;
#if PYGAT_X3G_WIDE
  {   // the 16 x 16 x 32 kernel: ONE 8-wave work-group per CU.  Slabs: as many as fill the CUs once (the caller sized them for two
      // 4-wave work-groups per CU), in whole 32-k steps
    const int tiles = (int)(cdiv(M, 128) * cdiv(N, 128));
    int sw = splits;
    if (sw > 1) {
      if (sw * tiles > 256) sw = 256 / tiles > 1 ? 256 / tiles : 1;
      if (sw > 1) {
        g.k_per_split = cdiv(cdiv(K, sw), 32) * 32;
        sw = (int)cdiv(K, g.k_per_split);
      }
      if (sw <= 1) { sw = 1; g.k_per_split = K; g.ws = nullptr; }
    }
    if (splits_used) *splits_used = sw;
    constexpr size_t ldsw = 2 * TNW_STAGE * sizeof(uint32_t);
    dim3 gridw((unsigned)sw, (unsigned)cdiv(M, 128), (unsigned)cdiv(N, 128));
#define PYGAT_X3GW_LAUNCH(KA, KB)                                                                                         \
  do {                                                                                                                    \
    int dev = -1;                                                                                                         \
    (void)hipGetDevice(&dev);                                                                                             \
    static bool attr_set[64] = {};                                                                                        \
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3gw_kernel<KA, KB>),                                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw);                                   \
      if (dev >= 0 && dev < 64) attr_set[dev] = true;                                                                     \
    }                                                                                                                     \
    hipLaunchKernelGGL((gemm_x3gw_kernel<KA, KB>), gridw, dim3(512), ldsw, st, g);                                        \
  } while (0)
    if (kca && kcb) PYGAT_X3GW_LAUNCH(true, true);
    else if (kca) PYGAT_X3GW_LAUNCH(true, false);
    else if (kcb) PYGAT_X3GW_LAUNCH(false, true);
    else PYGAT_X3GW_LAUNCH(false, false);
#undef PYGAT_X3GW_LAUNCH
    hipError_t ew = hipGetLastError();
    if (ew != hipSuccess) {
      set_error("gemm_x3gw: %s", hipGetErrorString(ew));
      return PYGAT_EHIP;
    }
    return 1;
  }
#endif
  dim3 grid((unsigned)splits, (unsigned)cdiv(M, 128), (unsigned)cdiv(N, 128));
  constexpr size_t lds = 2 * TNX_STAGE * sizeof(uint32_t);
#define PYGAT_X3G_LAUNCH(KA, KB)                                                                                          \
  do {                                                                                                                    \
    int dev = -1;                                                                                                         \
    (void)hipGetDevice(&dev);                                                                                             \
    static bool attr_set[64] = {};                                                                                        \
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3g_kernel<KA, KB>),                                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                    \
      if (dev >= 0 && dev < 64) attr_set[dev] = true;                                                                     \
    }                                                                                                                     \
    hipLaunchKernelGGL((gemm_x3g_kernel<KA, KB>), grid, dim3(256), lds, st, g);                                           \
  } while (0)
  if (kca && kcb) PYGAT_X3G_LAUNCH(true, true);
  else if (kca) PYGAT_X3G_LAUNCH(true, false);
  else if (kcb) PYGAT_X3G_LAUNCH(false, true);
  else PYGAT_X3G_LAUNCH(false, false);
#undef PYGAT_X3G_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("gemm_x3g: %s", hipGetErrorString(e));
    return PYGAT_EHIP;
  }
  return 1;
}

int footprint_gemm_x3(int which, int* regs, int* scratch) {
  hipFuncAttributes at;
  const void* fn = which == 0 ? reinterpret_cast<const void*>(&gemm_tn_x3w_kernel)
                              : reinterpret_cast<const void*>(&gemm_x3gw_kernel<true, false>);
  const hipError_t e = hipFuncGetAttributes(&at, fn);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error("kernel_footprint: %s", hipGetErrorString(e));
    return PYGAT_EHIP;
  }
  *regs = at.numRegs; *scratch = (int)at.localSizeBytes;
  return PYGAT_OK;
}

}  // namespace pygat

#if (PYGAT_DIAG_K1 & 16)
// diagnostic builds only: copies the stamps the last streamed-A GEMM launch left (device-synchronising); tools/k1_stamps.py
extern "C" __attribute__((visibility("default"))) int pygat_diag_k1_stamps(unsigned long long* host, int words) {
  const int all = 2048 * 8 * 8;
  if (!host || words < all) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(pygat::k1_stamps), sizeof(unsigned long long) * all) == hipSuccess ? all : -3;
}
#endif
