// Shared helpers for the libflowtrack_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flowtrack_hip.h"

namespace ft {

// Records the failing HIP call for ft_last_hip_error() and maps to FT_ERR_HIP.
int record_hip_error(hipError_t e, const char* what);

#define FT_HIP_CHECK(expr)                                         \
  do {                                                             \
    hipError_t _e = (expr);                                        \
    if (_e != hipSuccess) return ::ft::record_hip_error(_e, #expr); \
  } while (0)

// Kernel launches report configuration errors through hipGetLastError().
#define FT_LAUNCH_CHECK(name)                                        \
  do {                                                               \
    hipError_t _e = hipGetLastError();                               \
    if (_e != hipSuccess) return ::ft::record_hip_error(_e, name);   \
  } while (0)

// More than 64 KiB of dynamic LDS is an opt-in per kernel AND per device: one flag per device id, set on the first launch
// there (a process that drives several GPUs from one thread must not inherit another device's flag).
#define FT_RAISE_LDS(kernel, bytes)                                                                              \
  do {                                                                                                           \
    static bool _raised[64] = {};                                                                                \
    int _dev = 0;                                                                                                \
    FT_HIP_CHECK(hipGetDevice(&_dev));                                                                           \
    if (_dev < 0 || _dev >= 64 || !_raised[_dev]) {                                                              \
      FT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
      if (_dev >= 0 && _dev < 64) _raised[_dev] = true;                                                          \
    }                                                                                                            \
  } while (0)

static inline hipStream_t as_stream(ft_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline int ceil_div(int x, int m) { return (x + m - 1) / m; }

static inline size_t dtype_size(int dtype) { return dtype == FT_F16 ? 2 : 4; }

typedef _Float16 half_t;
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));

}  // namespace ft
