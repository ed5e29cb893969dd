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

// PYGAT_GEMM_F32=1 in the environment makes fp32 MFMA the default product mode of the process.  Read once at load
// and never written again: the mode of a call is an ARGUMENT of the entry points that run GEMMs (include/pygat_amd.h).
static const int g_default_split = [] { const char* e = getenv("PYGAT_GEMM_F32"); return (e && atoi(e) != 0) ? 0 : 1; }();
bool gemm_split(int mode) { return mode == PYGAT_GEMM_DEFAULT ? g_default_split != 0 : mode == PYGAT_GEMM_SPLIT_BF16; }
}  // namespace pygat

extern "C" int pygat_default_gemm_mode(void) { return pygat::g_default_split ? PYGAT_GEMM_SPLIT_BF16 : PYGAT_GEMM_FP32_MFMA; }

extern "C" int pygat_abi_version(void) { return PYGAT_ABI_VERSION; }
extern "C" const char* pygat_last_error(void) { return pygat::g_err; }
extern "C" int pygat_padded_width(int f_out) { return pygat::padded_width(f_out); }

namespace pygat {
int footprint_k2_headline(int* regs, int* scratch);
int footprint_k4_headline_da(int* regs, int* scratch);
int footprint_gemm_x3(int which, int* regs, int* scratch);
}  // namespace pygat

extern "C" int pygat_kernel_footprint(const char* kernel, int* num_regs, int* scratch_bytes) {
  PYGAT_REQUIRE(kernel && num_regs && scratch_bytes, "kernel_footprint: null argument");
  if (!strcmp(kernel, "k2_headline")) return pygat::footprint_k2_headline(num_regs, scratch_bytes);
  if (!strcmp(kernel, "k4_headline_da")) return pygat::footprint_k4_headline_da(num_regs, scratch_bytes);
  if (!strcmp(kernel, "tn_x3w")) return pygat::footprint_gemm_x3(0, num_regs, scratch_bytes);
  if (!strcmp(kernel, "x3gw")) return pygat::footprint_gemm_x3(1, num_regs, scratch_bytes);
  pygat::set_error("kernel_footprint: unknown kernel '%s' (k2_headline, k4_headline_da, tn_x3w, x3gw)", kernel);
  return PYGAT_EINVAL;
}

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
