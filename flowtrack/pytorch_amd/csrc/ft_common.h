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

// Activation-output stores.  FT_YSTORE_AUX = 16 (sc1) makes them write-through: nothing is left dirty in the XCD L2s for
// the end-of-kernel release to write back (the line is dropped from L2; the next launch reads it from the memory side).
#ifndef FT_YSTORE_AUX
#define FT_YSTORE_AUX 16
#endif
#define FT_YSTORE_BUF_AUX (FT_YSTORE_AUX == 2 ? 2 : ((FT_YSTORE_AUX & 16) ? FT_YSTORE_AUX : 0))   // raw_buffer_store aux: 2 = nt, 16 = sc1, 17 = sc0 sc1
__device__ __forceinline__ void store_out16(void* ptr, uint4_t v) {
#if FT_YSTORE_AUX == 16
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
#elif FT_YSTORE_AUX == 17
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
#elif FT_YSTORE_AUX == 2
  __builtin_nontemporal_store(v, reinterpret_cast<uint4_t*>(ptr));
#else
  *reinterpret_cast<uint4_t*>(ptr) = v;
#endif
}

}  // namespace ft
