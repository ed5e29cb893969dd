// K11 -- the optimiser step of the reference's training loop (train.py:64-66,122 and train_ppi.py:58-60: optim.Adam with
// weight decay, `optimizer.step()` once per epoch / batch) as ONE launch over all parameter tensors (gfx950).
//
// torch's own single-kernel Adam (fused=True, capturable=True) is two launches -- the step counters, then the update --
// of 3 + 12 us on the citation models' 92 k parameters: a sixteenth of a Cora epoch.  Here the tensors are a table in the
// kernel arguments (up to PYGAT_ADAM_MAX_TENSORS per launch), a work-group owns a 4096-element chunk of one tensor, and the
// step counter lives in device memory so that a replayed HIP graph advances it: every work-group reads the counter when it
// starts, and the LAST one to finish (a done-counter) writes counter + 1 -- by then all have read.
// The update is torch.optim.Adam's (torch/optim/adam.py _single_tensor_adam, maximize = amsgrad = False):
//   g += wd p;  m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g g;
//   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include "common.h"

namespace pygat {

constexpr int ADAM_MAX = 48;      // == PYGAT_ADAM_MAX_TENSORS
constexpr int ADAM_CHUNK = 4096;  // elements per work-group

struct AdamTable {
  float* p[ADAM_MAX];
  const float* g[ADAM_MAX];
  float* m[ADAM_MAX];
  float* v[ADAM_MAX];
  int64_t n[ADAM_MAX];
  int first_block[ADAM_MAX + 1];  // work-groups [first_block[t], first_block[t+1]) own tensor t
  int count;
};

// hyper-parameters as the doubles torch.optim.Adam holds them: 1 - beta and the bias corrections are formed in double and
// rounded once (1.f - 0.999f is 4.7e-5 off 0.001)
__global__ __launch_bounds__(256) void adam_step_kernel(AdamTable tab, double lr_d, double b1_d, double b2_d, double eps_d, double wd_d,
                                                        int* __restrict__ step, unsigned* __restrict__ done, double* __restrict__ pows) {
  const int t_now = *step + 1;                 // read before this work-group can be counted as done
  // beta^t as running products in the state (a pow() per thread cost more than the update: 152 us on PPI's 3.7 M parameters)
  const double b1t = t_now == 1 ? b1_d : pows[0] * b1_d, b2t = t_now == 1 ? b2_d : pows[1] * b2_d;
  int t = 0;
  while (t + 1 < tab.count && (int)blockIdx.x >= tab.first_block[t + 1]) ++t;   // uniform
  const int64_t base = (int64_t)((int)blockIdx.x - tab.first_block[t]) * ADAM_CHUNK;
  const double bc1 = 1.0 - b1t, bc2 = 1.0 - b2t;
  const float step_size = (float)(lr_d / bc1), inv_sq_bc2 = (float)(1.0 / sqrt(bc2));
  const float b2 = (float)b2_d, omb1 = (float)(1.0 - b1_d), omb2 = (float)(1.0 - b2_d), eps = (float)eps_d, wd = (float)wd_d;
  float* __restrict__ p = tab.p[t];
  const float* __restrict__ g = tab.g[t];
  float* __restrict__ m = tab.m[t];
  float* __restrict__ v = tab.v[t];
  const int64_t n = tab.n[t];
  auto upd = [&](float& pi, float gi, float& mi, float& vi) {
    gi = fmaf(wd, pi, gi);
    mi = fmaf(gi - mi, omb1, mi);
    vi = fmaf(b2, vi, omb2 * gi * gi);
    pi = pi - step_size * (mi / (sqrtf(vi) * inv_sq_bc2 + eps));
  };
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                     reinterpret_cast<uintptr_t>(v)) & 15u) == 0;      // uniform
  if (vec && n >= 4) {                           // 16-byte aligned tensors: float4s, four in flight per thread (also in a
    float4 pp[4], gg[4], mm[4], vv[4];           // tensor's last, partial chunk: out-of-range threads load element 0 and drop it)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t i = base + (q * 256 + threadIdx.x) * 4, ii = (i + 3 < n) ? i : 0;
      pp[q] = ld4(p + ii); gg[q] = ld4(g + ii); mm[q] = ld4(m + ii); vv[q] = ld4(v + ii);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t i = base + (q * 256 + threadIdx.x) * 4;
      if (i + 3 < n) {
        upd(pp[q].x, gg[q].x, mm[q].x, vv[q].x); upd(pp[q].y, gg[q].y, mm[q].y, vv[q].y);
        upd(pp[q].z, gg[q].z, mm[q].z, vv[q].z); upd(pp[q].w, gg[q].w, mm[q].w, vv[q].w);
        st4(p + i, pp[q]); st4(m + i, mm[q]); st4(v + i, vv[q]);
      } else if (i < n) {                        // the tensor's last 1-3 elements
        for (int64_t j = i; j < n; ++j) {
          float pi = p[j], mi = m[j], vi = v[j];
          upd(pi, g[j], mi, vi);
          p[j] = pi; m[j] = mi; v[j] = vi;
        }
      }
    }
  } else {
#pragma unroll 4
    for (int q = 0; q < ADAM_CHUNK / 256; ++q) {
      const int64_t i = base + q * 256 + threadIdx.x;
      if (i < n) {
        float pi = p[i], mi = m[i], vi = v[i];
        upd(pi, g[i], mi, vi);
        p[i] = pi; m[i] = mi; v[i] = vi;
      }
    }
  }
  // No fence: the last work-group reads nothing the others wrote -- the only ordering needed is that every work-group has READ
  // `step` before the last one overwrites it, and each read precedes its own work-group's count (a device-scope atomic) in
  // program order.  (A __threadfence() here is an L2 write-back + invalidate per wave on this multi-L2 part: it made the
  // kernel 6 x slower than its memory traffic.)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = atomicAdd(done, 1u);
    if (prev == gridDim.x - 1) {               // the last work-group: every other one read `step` long ago
      *step = t_now;
      pows[0] = b1t; pows[1] = b2t;
      atomicExch(done, 0u);
    }
  }
}

}  // namespace pygat

using namespace pygat;

/* state = {int32 step, uint32 done, double beta1^t, double beta2^t}: 24 bytes of device memory (8-byte aligned), zero before
 * the first step. */
extern "C" int pygat_adam_step(int ntensors, float* const* params, const float* const* grads, float* const* exp_avg,
                               float* const* exp_avg_sq, const int64_t* numel, double lr, double beta1, double beta2, double eps,
                               double weight_decay, void* state, void* stream) {
  PYGAT_REQUIRE(ntensors >= 1 && ntensors <= ADAM_MAX && params && grads && exp_avg && exp_avg_sq && numel && state && (reinterpret_cast<uintptr_t>(state) & 7u) == 0,
                "adam_step: bad arguments (1 <= tensors <= %d per call)", ADAM_MAX);
  PYGAT_REQUIRE(lr >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0., "adam_step: bad hyper-parameters");
  AdamTable tab;
  int blocks = 0;
  for (int t = 0; t < ntensors; ++t) {
    PYGAT_REQUIRE(params[t] && grads[t] && exp_avg[t] && exp_avg_sq[t] && numel[t] > 0, "adam_step: null tensor %d", t);
    tab.p[t] = params[t]; tab.g[t] = grads[t]; tab.m[t] = exp_avg[t]; tab.v[t] = exp_avg_sq[t]; tab.n[t] = numel[t];
    tab.first_block[t] = blocks;
    const int64_t nb = cdiv(numel[t], ADAM_CHUNK);
    PYGAT_REQUIRE(nb + blocks < ((int64_t)1 << 30), "adam_step: too many elements");
    blocks += (int)nb;
  }
  tab.first_block[ntensors] = blocks;
  tab.count = ntensors;
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tab, lr, beta1, beta2, eps,
                     weight_decay, (int*)state, (unsigned*)state + 1, (double*)state + 1);
  PYGAT_CHECK_LAUNCH("adam_step");
  return PYGAT_OK;
}
