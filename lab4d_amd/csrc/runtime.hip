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

// Every kernel-experiment macro (timing ablations whose results are WRONG, alternative store / schedule forms) this library was compiled with:
// "" for the shipped build.  The experiment builds of tools/build_variants.sh pass their -D flags to every translation unit (LAB4D_HIPCC_EXTRA), so
// this unit sees them too; tests/test_gpu_ops.py and tests/test_abi.py refuse a library that reports any (a stray -D ships a fast, wrong library).
// tests/test_abi.py also checks that the list below names every LAB4D_* macro the sources test.
extern "C" const char* lab4d_build_flags(void) {
  return ""
#ifdef LAB4D_ABL_ACG14
  " ABL_ACG14"
#endif
#ifdef LAB4D_ABL_ACG7
  " ABL_ACG7"
#endif
#ifdef LAB4D_ABL_HNOA
  " ABL_HNOA"
#endif
#ifdef LAB4D_ABL_L2STORE
  " ABL_L2STORE"
#endif
#ifdef LAB4D_ABL_MASK1
  " ABL_MASK1"
#endif
#ifdef LAB4D_ABL_NOAFETCH
  " ABL_NOAFETCH"
#endif
#ifdef LAB4D_ABL_NOBAR
  " ABL_NOBAR"
#endif
#ifdef LAB4D_ABL_NOMASK
  " ABL_NOMASK"
#endif
#ifdef LAB4D_ABL_NOPROG
  " ABL_NOPROG"
#endif
#ifdef LAB4D_ABL_NOSTORE
  " ABL_NOSTORE"
#endif
#ifdef LAB4D_ABL_OCC1
  " ABL_OCC1"
#endif
#ifdef LAB4D_ABL_PLAINSTORE
  " ABL_PLAINSTORE"
#endif
#ifdef LAB4D_ABL_WGRAD4
  " ABL_WGRAD4"
#endif
#ifdef LAB4D_ABL_WGRAD_L2
  " ABL_WGRAD_L2"
#endif
#ifdef LAB4D_ACACHE_G
  " ACACHE_G"
#endif
#ifdef LAB4D_ADMA
  " ADMA"
#endif
#ifdef LAB4D_A_NT
  " A_NT"
#endif
#ifdef LAB4D_A_SC
  " A_SC"
#endif
#ifdef LAB4D_FENCE_ALWAYS
  " FENCE_ALWAYS"
#endif
#ifdef LAB4D_H_ACG16
  " H_ACG16"
#endif
#ifdef LAB4D_MASK_RING
  " MASK_RING"
#endif
#ifdef LAB4D_MFMA_VGPR_FORM
  " MFMA_VGPR_FORM"
#endif
#ifdef LAB4D_PROG_FWD
  " PROG_FWD"
#endif
#ifdef LAB4D_SCHED_FWD_ON
  " SCHED_FWD_ON"
#endif
#ifdef LAB4D_SCHED_IL
  " SCHED_IL"
#endif
#ifdef LAB4D_SCHED_NV
  " SCHED_NV"
#endif
#ifdef LAB4D_SCHED_NV_FWD
  " SCHED_NV_FWD"
#endif
#ifdef LAB4D_ST_AGPR
  " ST_AGPR"
#endif
#ifdef LAB4D_ST_BUF
  " ST_BUF"
#endif
#ifdef LAB4D_TRSPREAD
  " TRSPREAD"
#endif
#ifdef LAB4D_TRSTORE
  " TRSTORE"
#endif
#ifdef LAB4D_WSABL_HALFB
  " WSABL_HALFB"
#endif
#ifdef LAB4D_WSABL_NOAPF
  " WSABL_NOAPF"
#endif
#ifdef LAB4D_WSABL_NOBIAS
  " WSABL_NOBIAS"
#endif
#ifdef LAB4D_WSABL_NOFLUSH
  " WSABL_NOFLUSH"
#endif
#ifdef LAB4D_WSABL_NOPOSENC
  " WSABL_NOPOSENC"
#endif
#ifdef LAB4D_WSABL_NOST
  " WSABL_NOST"
#endif
#ifdef LAB4D_WSABL_NOTR
  " WSABL_NOTR"
#endif
#ifdef LAB4D_WS_BD
  " WS_BD"
#endif
#ifdef LAB4D_WS_HIDE_POSENC
  " WS_HIDE_POSENC"
#endif
#ifdef LAB4D_WS_LINEAR_STORE
  " WS_LINEAR_STORE"
#endif
#ifdef LAB4D_WS_SWZ
  " WS_SWZ"
#endif
#ifdef LAB4D_WS_SYNC
  " WS_SYNC"
#endif
#ifdef LAB4D_WS_TOKEN
  " WS_TOKEN"
#endif
#ifdef LAB4D_WS_TRACE
  " WS_TRACE"
#endif
      ;
}
