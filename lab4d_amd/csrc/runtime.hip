// Error plumbing + version of liblab4d_hip.so
#include <stdarg.h>

#include "common.hpp"

namespace lab4d {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

__global__ void __launch_bounds__(256) k_zero_words(uint32_t* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}

int zero_async(void* p, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return LAB4D_OK;
  if ((bytes & 3) || ((uintptr_t)p & 3)) {
    set_error("zero_async: %zu bytes at %p is not a whole number of aligned words", bytes, p);
    return LAB4D_EINVAL;
  }
  const size_t n = bytes / 4;
  size_t blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(k_zero_words, dim3((unsigned)blocks), dim3(256), 0, stream, (uint32_t*)p, n);
  return check_launch("zero_async");
}
}  // namespace lab4d

extern "C" const char* lab4d_last_error(void) { return lab4d::g_err; }
extern "C" int lab4d_version(void) { return 1; }
extern "C" const char* lab4d_arch(void) { return "gfx950"; }
