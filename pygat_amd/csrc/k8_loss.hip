// K8 -- the training loss of the reference's citation-network script as two launches (gfx950).
//
// train.py:151-152,159:  output = F.log_softmax(F.elu(model(features, adj)), dim=1)
//                        loss   = F.nll_loss(output[idx_train], labels[idx_train])
// ATen runs that as elu, log_softmax, two index kernels, nll_loss and -- in the backward -- nll_loss_backward, three fills,
// an index_put with accumulate (five index-arithmetic kernels and a radix sort), log_softmax_backward and elu_backward:
// 17 launches of 2-5 us each in an epoch whose attention kernels take 7-10 us (tools/epoch_sequence.py).  Here:
//   forward   loss = sum_r w_r * (-log softmax(elu(out_r))[y_r]),   w_r = (multiplicity of r in the index set) / |index set|
//   backward  dOut[r, c] = g * w_r * (softmax(elu(out_r))[c] - [c == y_r]) * elu'(out[r, c])      (0 where w_r == 0)
// one thread per row (C is 3-7 classes), partial sums per block, the LAST block to finish adds them in block order
// (a counter in the workspace, left at zero again): deterministic, no second launch.
#include "common.h"

namespace pygat {

__device__ __forceinline__ float elu_f(float x) { return x > 0.f ? x : expm1f(x); }

__global__ __launch_bounds__(256) void nll_fwd_kernel(int n, int C, const float* __restrict__ out, int64_t ldo,
                                                      const int32_t* __restrict__ label, const float* __restrict__ weight,
                                                      float* __restrict__ ws, float* __restrict__ loss) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  float v = 0.f;
  if (r < n) {
    const float w = weight[r];
    if (w != 0.f) {
      const float* o = out + (int64_t)r * ldo;
      float mx = -INFINITY;
      for (int c = 0; c < C; ++c) mx = fmaxf(mx, elu_f(o[c]));
      float se = 0.f;
      for (int c = 0; c < C; ++c) se += expf(elu_f(o[c]) - mx);
      v = w * (mx + logf(se) - elu_f(o[label[r]]));
    }
  }
  // block sum in a fixed order: lanes by DPP / shuffles, waves through LDS
  v = group_sum<64>(v);
  __shared__ float sm[4];
  __shared__ int last;
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned int* counter = reinterpret_cast<unsigned int*>(ws);
  float* part = ws + 4;
  if (threadIdx.x == 0) {
    part[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    __threadfence();
    last = (atomicAdd(counter, 1u) == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (last) {   // the last block to arrive: every partial is visible (fence above, fence below)
    __threadfence();
    float t = 0.f;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += 256) t += __builtin_nontemporal_load(part + b);
    t = group_sum<64>(t);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
      loss[0] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
      *counter = 0u;   // ready for the next launch (a replayed HIP graph included)
    }
  }
}

// Small outputs (Cora 2708 rows, Citeseer 3327): ONE work-group of 1024 threads strides over the rows and reduces in LDS --
// no partials, no counter, no fences (a device-scope fence is an L2 write-back on this part: the multi-block form spends
// 6.5 us on Cora's 11 blocks, this one 3).  Same per-row values; the sum is taken in a different (fixed) order.
__global__ __launch_bounds__(1024) void nll_fwd_small_kernel(int n, int C, const float* __restrict__ out, int64_t ldo,
                                                             const int32_t* __restrict__ label, const float* __restrict__ weight,
                                                             float* __restrict__ loss) {
  float v = 0.f;
  for (int r = threadIdx.x; r < n; r += 1024) {
    const float w = weight[r];
    if (w != 0.f) {
      const float* o = out + (int64_t)r * ldo;
      float mx = -INFINITY;
      for (int c = 0; c < C; ++c) mx = fmaxf(mx, elu_f(o[c]));
      float se = 0.f;
      for (int c = 0; c < C; ++c) se += expf(elu_f(o[c]) - mx);
      v += w * (mx + logf(se) - elu_f(o[label[r]]));
    }
  }
  v = group_sum<64>(v);
  __shared__ float sm[16];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sm[q];
    loss[0] = t;
  }
}

__global__ __launch_bounds__(256) void nll_bwd_kernel(int n, int C, const float* __restrict__ out, int64_t ldo,
                                                      const int32_t* __restrict__ label, const float* __restrict__ weight,
                                                      const float* __restrict__ gscale, float* __restrict__ dout, int64_t ldd) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const float w = weight[r] * gscale[0];
  float* d = dout + (int64_t)r * ldd;
  if (w == 0.f) {
    for (int c = 0; c < C; ++c) d[c] = 0.f;
    return;
  }
  const float* o = out + (int64_t)r * ldo;
  float mx = -INFINITY;
  for (int c = 0; c < C; ++c) mx = fmaxf(mx, elu_f(o[c]));
  float se = 0.f;
  for (int c = 0; c < C; ++c) se += expf(elu_f(o[c]) - mx);
  const float rse = 1.f / se;
  const int y = label[r];
  for (int c = 0; c < C; ++c) {
    const float x = o[c];
    const float p = expf(elu_f(x) - mx) * rse;
    d[c] = w * (p - (c == y ? 1.f : 0.f)) * (x > 0.f ? 1.f : expf(x));
  }
}

// train_ppi.py:114,157 -- nn.BCEWithLogitsLoss(reduction='mean') over the [N, 121] logits of the PPI model: ATen runs it and its
// backward as a dozen element-wise launches and two reductions (60 us of a 2.6 ms epoch).  Here
//   forward   loss = 1/n sum ( max(x, 0) - x y + log1p(exp(-|x|)) )        (ATen's own stable form)
//   backward  dx   = g/n (sigmoid(x) - y)
// BCE_PER_BLOCK elements per work-group, partial sums per block, the last block adds them in block order (as above).
constexpr int BCE_BWD_PER_BLOCK = 256 * 8;
constexpr int BCE_PER_BLOCK = 256 * 8;   // 8 elements per thread (64 per thread -- an eighth of the work-groups and of their fences -- was measured: 27 us instead of 11, the launch is latency-bound)

__global__ __launch_bounds__(256) void bce_fwd_kernel(int64_t total, const float* __restrict__ x, const float* __restrict__ y,
                                                      float* __restrict__ ws, float* __restrict__ loss) {
  float v = 0.f;
  const int64_t base = (int64_t)blockIdx.x * BCE_PER_BLOCK + threadIdx.x;
  for (int q0 = 0; q0 < BCE_PER_BLOCK / 256; q0 += 8) {      // eight loads of each operand in flight
    float xs[8], ys[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int64_t i = base + (int64_t)(q0 + q) * 256;
      xs[q] = i < total ? x[i] : 0.f;
      ys[q] = i < total ? y[i] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (base + (int64_t)(q0 + q) * 256 < total) v += fmaxf(xs[q], 0.f) - xs[q] * ys[q] + log1pf(expf(-fabsf(xs[q])));
  }
  v = group_sum<64>(v);
  __shared__ float sm[4];
  __shared__ int last;
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned int* counter = reinterpret_cast<unsigned int*>(ws);
  float* part = ws + 4;
  if (threadIdx.x == 0) {
    part[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    __threadfence();
    last = (atomicAdd(counter, 1u) == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (last) {
    __threadfence();
    float t = 0.f;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += 256) t += __builtin_nontemporal_load(part + b);
    t = group_sum<64>(t);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
      loss[0] = ((sm[0] + sm[1]) + (sm[2] + sm[3])) / (float)total;
      *counter = 0u;
    }
  }
}

__global__ __launch_bounds__(256) void bce_bwd_kernel(int64_t total, const float* __restrict__ x, const float* __restrict__ y,
                                                      const float* __restrict__ gscale, float* __restrict__ dx) {
  const float g = gscale[0] / (float)total;
  const int64_t base = (int64_t)blockIdx.x * BCE_BWD_PER_BLOCK + threadIdx.x;
#pragma unroll
  for (int q = 0; q < BCE_BWD_PER_BLOCK / 256; ++q) {
    const int64_t i = base + q * 256;
    if (i < total) dx[i] = g * (1.f / (1.f + expf(-x[i])) - y[i]);
  }
}

}  // namespace pygat

using namespace pygat;

extern "C" size_t pygat_bce_workspace_bytes(int64_t total) {
  if (total <= 0) return 0;
  return (size_t)(4 + cdiv(total, BCE_PER_BLOCK)) * sizeof(float);
}

extern "C" int pygat_bce_with_logits(int64_t total, const float* logits, const float* targets, void* ws, float* loss, void* stream) {
  PYGAT_REQUIRE(total > 0 && total < ((int64_t)1 << 40) && logits && targets && ws && loss, "bce_with_logits: bad arguments");
  hipLaunchKernelGGL(bce_fwd_kernel, dim3((unsigned)cdiv(total, BCE_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, total, logits,
                     targets, (float*)ws, loss);
  PYGAT_CHECK_LAUNCH("bce_with_logits");
  return PYGAT_OK;
}

extern "C" int pygat_bce_with_logits_backward(int64_t total, const float* logits, const float* targets, const float* gscale,
                                              float* dlogits, void* stream) {
  PYGAT_REQUIRE(total > 0 && logits && targets && gscale && dlogits, "bce_with_logits_backward: bad arguments");
  hipLaunchKernelGGL(bce_bwd_kernel, dim3((unsigned)cdiv(total, BCE_BWD_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, total, logits,
                     targets, gscale, dlogits);
  PYGAT_CHECK_LAUNCH("bce_with_logits_backward");
  return PYGAT_OK;
}


extern "C" size_t pygat_nll_workspace_bytes(int n) {
  if (n <= 0) return 0;
  return (size_t)(4 + cdiv(n, 256)) * sizeof(float);
}

extern "C" int pygat_elu_logsoftmax_nll(int n, int C, const float* out, int64_t ldo, const int32_t* label, const float* weight,
                                        void* ws, float* loss, void* stream) {
  PYGAT_REQUIRE(n > 0 && C > 0 && out && label && weight && ws && loss && ldo >= C, "elu_logsoftmax_nll: bad arguments");
  if (n <= 8192)
    hipLaunchKernelGGL(nll_fwd_small_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, C, out, ldo, label, weight, loss);
  else
    hipLaunchKernelGGL(nll_fwd_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n, C, out, ldo, label,
                       weight, (float*)ws, loss);
  PYGAT_CHECK_LAUNCH("elu_logsoftmax_nll");
  return PYGAT_OK;
}

extern "C" int pygat_elu_logsoftmax_nll_backward(int n, int C, const float* out, int64_t ldo, const int32_t* label,
                                                 const float* weight, const float* gscale, float* dout, int64_t ldd, void* stream) {
  PYGAT_REQUIRE(n > 0 && C > 0 && out && label && weight && gscale && dout && ldo >= C && ldd >= C,
                "elu_logsoftmax_nll_backward: bad arguments");
  hipLaunchKernelGGL(nll_bwd_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n, C, out, ldo, label,
                     weight, gscale, dout, ldd);
  PYGAT_CHECK_LAUNCH("elu_logsoftmax_nll_backward");
  return PYGAT_OK;
}
