// Shared helpers for the libdiagan_hip.so kernels (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <mutex>

#define DIAGAN_OK 0
#define DIAGAN_EINVAL (-1)   // bad argument (shape / alignment / null pointer)
#define DIAGAN_EHIP (-2)     // a HIP runtime call or kernel launch failed
#define DIAGAN_EUNSUP (-3)   // configuration not supported by the kernels

#define DIAGAN_API extern "C" __attribute__((visibility("default")))

namespace diagan {

// thread-local last-error text, read through diagan_last_error()
char* err_buf();
int set_err(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_err(DIAGAN_EHIP, "%s: %s", what, hipGetErrorString(e));
  return DIAGAN_OK;
}

#define DG_REQUIRE(cond, ...)                                  \
  do {                                                         \
    if (!(cond)) return diagan::set_err(DIAGAN_EINVAL, __VA_ARGS__); \
  } while (0)

#define DG_HIP(call)                                                              \
  do {                                                                            \
    hipError_t _e = (call);                                                       \
    if (_e != hipSuccess)                                                         \
      return diagan::set_err(DIAGAN_EHIP, "%s: %s", #call, hipGetErrorString(_e)); \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE): one latch object per kernel, one entry per device,
// taken under a lock so that two host threads (or two devices of one process) launching the same kernel both find it set.
struct FuncAttrLatch {
  std::mutex m;
  size_t have[64] = {};
};
// `kernel` may be launched with `bytes` of dynamic LDS on the CURRENT device from here on; hipSuccess or the runtime's error
inline hipError_t ensure_dynamic_lds(FuncAttrLatch& l, const void* kernel, size_t bytes) {
  int d = 0;
  hipError_t e = hipGetDevice(&d);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> g(l.m);
  size_t& have = l.have[d & 63];
  if (have >= bytes) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) have = bytes;
  return e;
}
#define DG_LDS(latch, kern, bytes)                                                                                    \
  do {                                                                                                                \
    hipError_t _e = diagan::ensure_dynamic_lds(latch, (const void*)(kern), (bytes));                                  \
    if (_e != hipSuccess) return diagan::set_err(DIAGAN_EHIP, "%s: %zu bytes of dynamic LDS: %s", #kern, (size_t)(bytes), hipGetErrorString(_e)); \
  } while (0)

// wave64 reductions by shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace diagan
