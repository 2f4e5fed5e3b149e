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

typedef float float2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// act(v) for k = 0 (ReLU) / 0 < k <= 1 (LeakyReLU) / 1 (none) = `v > 0 ? v : k * v` with torch's value for EVERY input, three vector
// instructions: t = k (*) v with (*) = v_mul_legacy_f32 (0 * x = 0 for every x, so ReLU(-inf) = 0 where the IEEE product -inf * 0 = NaN
// made the two-instruction max(v, k * v) of rounds 2-5 return -inf), then `!(v <= 0) ? v : t`: the negated ordered compare keeps a NaN
// (F.relu(NaN) = NaN; v_max would drop it) and +inf, every other value takes t.  conv_common.h: apply_act.
__device__ __forceinline__ float act_mul(float v, float k) {
  float t;
  asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(t) : "v"(k), "v"(v));      // hipcc has no builtin for the legacy multiply
  return !(v <= 0.f) ? v : t;
}

// Packed epilogues.  The fused kernels are bound by instruction ISSUE (a SIMD issues about one instruction per four cycles; of
// the ~7000 instructions of a 128- / 256-plane block's workgroup 8-12 % are MFMAs, 30-40 % are the BatchNorm + ReLU + fp16
// epilogues: profiles/README.md, round 5), so two results per instruction where the ISA has it:
//   fp16(relu(a * k + b))      = v_pk_fma_f32, v_cvt_pk_f16_f32, v_pk_max_f16                     (scalar form: 2 fma, 2 max, 1 cvt_pk)
//   fp16(relu(a * k + b + r))  = v_pk_fma_f32, 2 v_cvt_f32_f16, v_pk_add_f32, v_cvt_pk_f16_f32, v_pk_max_f16
// v_pk_fma_f32 / v_pk_add_f32 are the same IEEE operations as the scalar forms, and ReLU commutes with the rounding to fp16
// (monotonic, sign-preserving): the same values as `(half_t)fmaxf(a * k + b, 0.f)`; a negative result that rounds to -0 comes out
// as a zero of either sign.
// (FT_PK_EPILOGUE=0: the scalar forms, A/B builds: tools/dev/build_variant.sh nopk -DFT_PK_EPILOGUE=0)
#ifndef FT_PK_EPILOGUE
#define FT_PK_EPILOGUE 1
#endif
__device__ __forceinline__ half2_t bn_relu_pk(float a0, float a1, float k0, float k1, float b0, float b1) {
#if FT_PK_EPILOGUE
  const float2_t v = __builtin_elementwise_fma(float2_t{a0, a1}, float2_t{k0, k1}, float2_t{b0, b1});
  return __builtin_elementwise_max(__builtin_convertvector(v, half2_t), half2_t{(half_t)0.f, (half_t)0.f});
#else
  return half2_t{(half_t)__builtin_fmaxf(a0 * k0 + b0, 0.f), (half_t)__builtin_fmaxf(a1 * k1 + b1, 0.f)};
#endif
}
__device__ __forceinline__ half2_t bn_res_relu_pk(float a0, float a1, float k0, float k1, float b0, float b1, half_t r0, half_t r1) {
#if FT_PK_EPILOGUE
  const float2_t v = __builtin_elementwise_fma(float2_t{a0, a1}, float2_t{k0, k1}, float2_t{b0, b1}) + __builtin_convertvector(half2_t{r0, r1}, float2_t);
  return __builtin_elementwise_max(__builtin_convertvector(v, half2_t), half2_t{(half_t)0.f, (half_t)0.f});
#else
  return half2_t{(half_t)__builtin_fmaxf(a0 * k0 + b0 + (float)r0, 0.f), (half_t)__builtin_fmaxf(a1 * k1 + b1 + (float)r1, 0.f)};
#endif
}
// the sixteen results of a 32 x 32 accumulator block of one lane (register r: scale / shift [r >> 2][r & 3]) as two 16-byte halves
__device__ __forceinline__ void bn_relu_acc16(const float16_t& acc, const float4_t (&sc)[4], const float4_t (&sh)[4], half8_t (&h8)[2]) {
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const half2_t o = bn_relu_pk(acc[r], acc[r + 1], sc[r >> 2][r & 3], sc[r >> 2][(r & 3) + 1], sh[r >> 2][r & 3], sh[r >> 2][(r & 3) + 1]);
    h8[r >> 3][r & 7] = o[0];
    h8[r >> 3][(r & 7) + 1] = o[1];
  }
}
// registers 8 h .. 8 h + 7 plus the residual's eight halves
__device__ __forceinline__ half8_t bn_res_relu_acc8(const float16_t& acc, int h, const float4_t (&sc)[4], const float4_t (&sh)[4], const half8_t& rs) {
  half8_t o;
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const int r = h * 8 + e;
    const half2_t v = bn_res_relu_pk(acc[r], acc[r + 1], sc[r >> 2][r & 3], sc[r >> 2][(r & 3) + 1], sh[r >> 2][r & 3], sh[r >> 2][(r & 3) + 1], rs[e], rs[e + 1]);
    o[e] = v[0];
    o[e + 1] = v[1];
  }
  return o;
}

// Folded form (round 6): the BatchNorm scale sits in the fp16 weights, the shift (and the identity residual) reach the accumulator as
// extra MFMA k-steps, so the epilogue of a 32 x 32 block is fp16(relu(acc)): v_cvt_pk_f16_f32 + v_pk_max_f16 per pair of values.
__device__ __forceinline__ void relu_acc16(const float16_t& acc, half8_t (&h8)[2]) {
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const half2_t o = __builtin_elementwise_max(__builtin_convertvector(float2_t{acc[r], acc[r + 1]}, half2_t), half2_t{(half_t)0.f, (half_t)0.f});
    h8[r >> 3][r & 7] = o[0];
    h8[r >> 3][(r & 7) + 1] = o[1];
  }
}

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
