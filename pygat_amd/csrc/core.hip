// Error string, version and device queries of the C ABI (include/pygat_amd.h).
#include "common.h"
#include <string.h>

namespace pygat {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// PYGAT_GEMM_F32=1 in the environment starts the process on the fp32 MFMA kernels
static int g_gemm_split = [] { const char* e = getenv("PYGAT_GEMM_F32"); return (e && atoi(e) != 0) ? 0 : 1; }();
int gemm_split_mode() { return g_gemm_split; }
}  // namespace pygat

extern "C" int pygat_get_gemm_mode(void) { return pygat::g_gemm_split ? PYGAT_GEMM_SPLIT_BF16 : PYGAT_GEMM_FP32_MFMA; }
extern "C" int pygat_set_gemm_mode(int mode) {
  if (mode != PYGAT_GEMM_SPLIT_BF16 && mode != PYGAT_GEMM_FP32_MFMA) {
    pygat::set_error("set_gemm_mode: unknown mode %d", mode);
    return PYGAT_EINVAL;
  }
  pygat::g_gemm_split = (mode == PYGAT_GEMM_SPLIT_BF16);
  return PYGAT_OK;
}

extern "C" int pygat_abi_version(void) { return PYGAT_ABI_VERSION; }
extern "C" const char* pygat_last_error(void) { return pygat::g_err; }
extern "C" int pygat_padded_width(int f_out) { return pygat::padded_width(f_out); }

extern "C" int pygat_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

extern "C" int pygat_device_name(char* host_buf, int len) {
  if (!host_buf || len <= 0) return PYGAT_EINVAL;
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
    (void)hipGetLastError();
    pygat::set_error("no HIP device");
    return PYGAT_ENODEV;
  }
  snprintf(host_buf, (size_t)len, "%s (%s)", p.name, p.gcnArchName);
  return PYGAT_OK;
}
