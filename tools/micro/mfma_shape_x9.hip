// does v_mfma_f32_16x16x32_bf16 buy the nine-product (exact three-way bf16 split) GEMM loop the
// 1.12-1.15 x MI355X_MICROARCH.md (DVFS give-back, item 7) reports for bare loops?  Same structure as gemm_x3g_kernel's consumer
// side -- work-groups of 4 waves, each wave a 64 x 64 block of a 128 x 128 tile, three bf16 images of A and of B in LDS with
// 16-byte fragments read by ds_read_b128, all nine piece products of a block chained on one accumulator -- on random data, with
// FILL vector instructions per 32 k and wave (the split + LDS-write work of the real kernel: ~260) placed between the MFMAs:
//   shape 32: per 16 k   2 x 2 tiles of 32 x 32 x 16,  12 fragment reads,  36 MFMAs of 32 cycles
//   shape 16: per 32 k   4 x 4 tiles of 16 x 16 x 32,  24 fragment reads, 144 MFMAs of 16 cycles
// Equal FLOPs, LDS bytes and accumulator registers.   hipcc --offload-arch=gfx950 -O3 mfma_shape_x9.hip -o mfma_shape_x9
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
static uint32_t rand_bf16() { float x = (rand() / (float)RAND_MAX) * 4.f - 2.f; uint32_t u; ::memcpy(&u, &x, 4); return u >> 16; }
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int RS = 20;                 // dwords per image row: 32 k of bf16 (16 dwords) + 4 of padding
constexpr int IMG = 128 * RS;          // one piece image of one operand
__device__ __forceinline__ uint4 ldf(const uint32_t* p) { return *reinterpret_cast<const uint4*>(p); }

template <int SHAPE, int FILL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(const uint32_t* __restrict__ src, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];   // [6 images] (61 KB: two work-groups per CU)
  for (int i = threadIdx.x; i < 6 * IMG; i += 256) lds[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float fill = (float)lane;
  if constexpr (SHAPE == 32) {
    const int fr = lane & 31, fh = lane >> 5;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const uint32_t* fa = lds + (64 * (w >> 1) + fr) * RS + 4 * fh;
    const uint32_t* fb = lds + 3 * IMG + (64 * (w & 1) + fr) * RS + 4 * fh;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {     // two 16-k steps = one 32-k unit
        const uint32_t* a_ = fa + 0 + 8 * half;
        const uint32_t* b_ = fb + 0 + 8 * half;
        uint4 fq[4][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          fq[0][p] = ldf(a_ + p * IMG); fq[1][p] = ldf(a_ + p * IMG + 32 * RS);
          fq[2][p] = ldf(b_ + p * IMG); fq[3][p] = ldf(b_ + p * IMG + 32 * RS);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
            for (int pa = 2; pa >= 0; --pa)
#pragma unroll
              for (int pb = 2; pb >= 0; --pb)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fq[tm][pa]), __builtin_bit_cast(bf16x8, fq[2 + tn][pb]), acc[tm][tn], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < FILL / 8; ++f) fill = __builtin_fmaf(fill, 1.0001f, 0.5f);
          }
      }
      asm volatile("" ::: "memory");
    }
    float s = fill;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    const int fr = lane & 15, fq4 = lane >> 4;     // lane (row fr, k quarter fq4) holds k = 8 fq4 .. + 7 of the 32
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    const uint32_t* fa = lds + (64 * (w >> 1) + fr) * RS + 4 * fq4;
    const uint32_t* fb = lds + 3 * IMG + (64 * (w & 1) + fr) * RS + 4 * fq4;
    for (int it = 0; it < iters; ++it) {
      const uint32_t* a_ = fa + 0;
      const uint32_t* b_ = fb + 0;
      uint4 af[4][3], bf[2][3];
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int p = 0; p < 3; ++p) af[tm][p] = ldf(a_ + p * IMG + 16 * tm * RS);
#pragma unroll
      for (int p = 0; p < 3; ++p) bf[0][p] = ldf(b_ + p * IMG);
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        if (tn + 1 < 4) {
#pragma unroll
          for (int p = 0; p < 3; ++p) bf[(tn + 1) & 1][p] = ldf(b_ + p * IMG + 16 * (tn + 1) * RS);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
          for (int pa = 2; pa >= 0; --pa)
#pragma unroll
            for (int pb = 2; pb >= 0; --pb)
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[tm][pa]), __builtin_bit_cast(bf16x8, bf[tn & 1][pb]), acc[tm][tn], 0, 0, 0);
#pragma unroll
          for (int f = 0; f < FILL / 16; ++f) fill = __builtin_fmaf(fill, 1.0001f, 0.5f);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("" ::: "memory");
    }
    float s = fill;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}

template <int SHAPE, int FILL>
static double run(const uint32_t* src, float* out, int wgs, int iters, int reps) {
  const size_t ldsb = 6 * IMG * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<SHAPE, FILL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<SHAPE, FILL>), dim3(wgs), dim3(256), ldsb, 0, src, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<SHAPE, FILL>), dim3(wgs), dim3(256), ldsb, 0, src, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  // per wave and 32-k unit: 64 x 64 x 32 x 2 flops x 9 products
  const double flops = (double)wgs * 4 * iters * 64.0 * 64.0 * 32.0 * 2.0 * 9.0;
  printf("shape %2d fill %3d: %8.3f ms  %7.1f TF bf16 (= %6.1f TF fp32-equivalent)\n", SHAPE, FILL, ms, flops / ms / 1e9, flops / 9 / ms / 1e9);
  return ms;
}

int main() {
  const size_t n = 6 * IMG;
  uint32_t* h = (uint32_t*)malloc(n * 4);
  srand(1);
  for (size_t i = 0; i < n; ++i) {   // two random bf16 in [-2, 2) per dword
    h[i] = rand_bf16() | (rand_bf16() << 16);
  }
  uint32_t* src; float* out;
  hipMalloc(&src, n * 4); hipMalloc(&out, 4096 * 256 * 4);
  hipMemcpy(src, h, n * 4, hipMemcpyHostToDevice);
  const int wgs = 512, iters = 2000;    // two work-groups per CU, ~ the x3g kernels' residency
  for (int rep = 0; rep < 2; ++rep) {
    run<32, 0>(src, out, wgs, iters, 5);
    run<16, 0>(src, out, wgs, iters, 5);
    run<32, 128>(src, out, wgs, iters, 5);
    run<16, 128>(src, out, wgs, iters, 5);
    run<32, 256>(src, out, wgs, iters, 5);
    run<16, 256>(src, out, wgs, iters, 5);
  }
  return 0;
}
