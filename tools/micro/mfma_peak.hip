// fp32 MFMA peak probe (development tool): 8 waves/CU (2 per SIMD), NT independent accumulators,
// operands in registers (no memory).  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NT>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  f32x16 acc[NT];
  for (int i = 0; i < NT; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[nt], 0, 0, 0);
    a += 1e-6f;
  }
  float s = 0;
  for (int i = 0; i < NT; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NT> void run(const char* name, int wgs, int threads) {
  float* out; hipMalloc(&out, 4 * 4096 * 512);
  int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NT>), dim3(wgs), dim3(threads), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NT>), dim3(wgs), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)wgs * (threads / 64) * iters * 8.0 * NT * 4096.0;
  printf("%s NT=%d wgs=%d thr=%d: %.3f ms  %.1f TFLOP/s\n", name, NT, wgs, threads, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<5>("mfma32x32x2", 256, 512);
  run<4>("mfma32x32x2", 256, 512);
  run<5>("mfma32x32x2", 256, 256);
  run<4>("mfma32x32x2", 512, 256);
  run<1>("mfma32x32x2", 256, 512);
  return 0;
}
