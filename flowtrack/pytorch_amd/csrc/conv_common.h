// Shared pieces of the implicit-GEMM conv kernels (conv_igemm.hip: the tiled one-barrier-per-K-step kernels and the
// host dispatch; conv_igemm8.hip: the 256 x 256 8-phase kernel): kernel parameter block, MFMA slice helpers and the
// common epilogue (folded BN / bias, residual, activation, LDS-transposed 16-byte stores, fused tail 1x1 conv).
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "ft_common.h"

#ifndef FT_EPI_NT
#define FT_EPI_NT 0     // non-temporal stores in the fp16 epilogue (dev A/B)
#endif
// Workgroup barrier of the K-loops.  The builtin is IntrNoMem for LLVM: ds_reads that follow it in program order may be
// hoisted ABOVE it (measured: the stem kernel read patch rows other waves' LDS-DMA had not landed yet, ~0.1 % of the
// tiles wrong once workgroups are recycled on a CU).  The inline-asm form with a memory clobber pins the order.
#define FT_LDS_BARRIER() asm volatile("s_barrier" ::: "memory")

namespace ft {

struct ConvParams {
  const char* x;
  const char* w;
  const float* scale;
  const float* shift;
  const char* res;
  char* y;
  int M;            // pixels per phase = N * Hq * Wq
  int HqWq, Wq;     // pixel-grid decode
  int Hi, Wi;
  int sy;           // input step per pixel-grid step (conv stride; 1 for transposed)
  int x_cstride, x_coff;
  int kh, kw;       // taps per phase (2x2 for transposed)
  int dmul;         // +1 conv, -1 transposed
  int pad;          // conv padding, rows (unused for transposed)
  int pad_x;        // conv padding, columns (row-packed inputs: pad - x_lpad <= 0, the buffer holds the zeros)
  int transposed;
  int cin_groups;   // generic path: roundup8(Cin) / VEC
  int kc;           // dma path: K-steps per tap = cin_pad / BK
  unsigned x_bytes; // dma path: size of the activation buffer (buffer descriptor range)
  int nk;           // K-steps
  int Kpad;         // elements per packed weight row
  int Cout, Cout_pad;
  int Ho, Wo, omul; // output tensor size; 1 (conv) or 2 (transposed) output step per grid step
  int y_cstride, y_coff, out_layout;
  int res_cstride, res_coff;
  int act;
  float slope;
  int npt, nct, nph;  // pixel tiles, output-channel tiles, phases (grid = npt * nct * nph, 1-D)
  int epi_lds;        // fp16 NHWC, 8-channel aligned: transpose the tile through LDS for 16-byte coalesced stores
  int sk;             // split-K across workgroups (1: none): grid = tiles * sk, fp32 partial tiles go to `ws`
  float* ws;          // [sk][nph * M][Cout_pad] partial sums, reduced by conv_splitk_reduce_kernel
  const char* tail_w; // fused tail 1x1 conv: fp16 [32][Cout] weights (hi) + fp32 [32] bias + fp16 [32][Cout] (lo = w - hi), or nullptr
  int tail_cout;
  const char* x2;     // second input (K-concat), or nullptr
  int kc2;            // its K-steps (0: none)
  int x2_hi, x2_wi, x2_cstride, x2_coff, x2_stride;
  unsigned x2_bytes;
  int h_tx, h_ty;     // halo path: patch tiles per image row / column
  int h_pw, h_npix;   // halo path: input-patch width and pixel count
  int h_npww, h_pb;   // halo path: patch wave-loads per wave per chunk, bytes of one patch buffer
  unsigned y_bytes;   // conv_igemm8_kernel: size of the output buffer (buffer descriptor range of its direct stores)
  unsigned w_bytes;   // conv_igemm8_kernel: size of the packed weight set (all phases and channel tiles)
  int dbg;            // developer ablation (FT_CONV_DBG): 1 = no MFMA, 2 = no operand loads, 4 = no epilogue; 0 in production
  int shift_n;        // > 0: `shift` is per sample, [N][shift_n] floats (conv_stem_persist_kernel only)
  int x_planar;       // 1: x is the NCHW fp32 network input (conv_stem_pool_kernel only); x_lpad / x_w / x_c: the virtual view's left pad, the planes' width, their count
  int x_lpad, x_w, x_c;
};

template <typename T> struct Elem;
template <> struct Elem<half_t> { static constexpr int VEC = 8; };
template <> struct Elem<float> { static constexpr int VEC = 4; };

// act(v) = `v > 0 ? v : k * v` with k = 0 (ReLU), slope (LeakyReLU, 0 <= slope <= 1) or 1 (none), branch-free (ft_common.h: act_mul,
// three vector instructions; the branchy form compiled to two scalar compare-and-branch pairs PER ELEMENT inside the unrolled
// epilogues: 64 s_cbranch per 128-pixel tile in the stem kernel).  Non-finite inputs come out as torch has them: NaN stays NaN,
// +inf stays +inf, -inf becomes k * -inf and 0 under ReLU (rounds 2-5 returned -inf there: max(v, 0 * v) with the IEEE product
// NaN).  A slope > 1 takes the compare-and-select form (uniform branch).  conv_direct.hip / conv_wstat.hip call act_mul() in their
// own epilogues; the ReLU-only fused bottleneck kernels use max(v, 0) and conv_igemm8.hip max(v, 0) + k * min(v, 0) (NaN -> 0 in both,
// documented there).  tests/test_conv_gpu.py: test_epilogue_keeps_non_finite_values, test_relu_of_non_finite_values.
// (act_mul: ft_common.h)
__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  const float k = act == FT_ACT_RELU ? 0.f : (act == FT_ACT_LEAKY ? slope : 1.f);
  if (k <= 1.f) return act_mul(v, k);
  return v > 0.f ? v : v * k;
}

template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}
template <int B, int E, typename F>
__device__ __forceinline__ void static_for_from(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for_from<B + 1, E>(f);
  }
}

// One 32-byte-per-row K slice for every (i, j) MFMA tile of the wave.
template <int MT_C, int MT_P>
__device__ __forceinline__ void mma_slice(const uint4_t (&a)[MT_C], const uint4_t (&b)[MT_P],
                                          float16_t (&acc)[MT_C][MT_P], half_t*) {
#pragma unroll
  for (int i = 0; i < MT_C; ++i)
#pragma unroll
    for (int j = 0; j < MT_P; ++j)
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
          __builtin_bit_cast(half8_t, a[i]), __builtin_bit_cast(half8_t, b[j]), acc[i][j], 0, 0, 0);
}
template <int MT_C, int MT_P>
__device__ __forceinline__ void mma_slice(const uint4_t (&a)[MT_C], const uint4_t (&b)[MT_P],
                                          float16_t (&acc)[MT_C][MT_P], float*) {
  // whole-vector bit casts: __builtin_bit_cast on a single vector ELEMENT lvalue reads element 0
  float4_t af[MT_C], bf[MT_P];
#pragma unroll
  for (int i = 0; i < MT_C; ++i) af[i] = __builtin_bit_cast(float4_t, a[i]);
#pragma unroll
  for (int j = 0; j < MT_P; ++j) bf[j] = __builtin_bit_cast(float4_t, b[j]);
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int i = 0; i < MT_C; ++i)
#pragma unroll
      for (int j = 0; j < MT_P; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
}

__device__ __forceinline__ void store4(half_t* dst, const float (&v)[4]) {
  half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
  *reinterpret_cast<half4_t*>(dst) = h;
}
__device__ __forceinline__ void store4(float* dst, const float (&v)[4]) {
  float4_t f = {v[0], v[1], v[2], v[3]};
  *reinterpret_cast<float4_t*>(dst) = f;
}
__device__ __forceinline__ void load4(const half_t* src, float (&v)[4]) {
  half4_t h = *reinterpret_cast<const half4_t*>(src);
  v[0] = (float)h[0]; v[1] = (float)h[1]; v[2] = (float)h[2]; v[3] = (float)h[3];
}
__device__ __forceinline__ void load4(const float* src, float (&v)[4]) {
  float4_t f = *reinterpret_cast<const float4_t*>(src);
  v[0] = f[0]; v[1] = f[1]; v[2] = f[2]; v[3] = f[3];
}

// ---- shared epilogue -------------------------------------------------------------------------------
// acc[i][j][reg]: output channel co0 + wc*WT_C + i*32 + (reg&3) + 8*(reg>>2) + 4*(lane>>5),
//                 pixel m0 + wp*WT_P + j*32 + (lane&31)   (32x32 MFMA C/D layout, weights as operand A).
// smem: at least BP*BC*2 bytes reusable + BP*8 bytes at offset `opix_off` (all K-loop LDS traffic done).
// Output pixel index of every tile row (or -1 past the end), shared by the residual prefetch and the epilogue.
template <int BP>
__device__ __forceinline__ void conv_row_table(const ConvParams& p, long long* s_opix, int m0, int py, int px) {
  const int tid = threadIdx.x;
  if (tid < BP) {
    const int m = m0 + tid;
    long long o = -1;
    if (m < p.M) {
      const int n = m / p.HqWq;
      const int rem = m - n * p.HqWq;
      const int qy = rem / p.Wq;
      const int qx = rem - qy * p.Wq;
      o = ((long long)n * p.Ho + (qy * p.omul + py)) * p.Wo + (qx * p.omul + px);
    }
    s_opix[tid] = o;
  }
}

// PRE: the row table is already in LDS and `rpre` holds this thread's residual chunks (issued before the
// K-loop so their HBM latency overlaps the operand loads); otherwise both are produced here.
template <typename T, int BP, int BC, int WGP, int WGC, bool PRE = false, int NT = 256>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, float16_t (&acc)[BC / WGC / 32][BP / WGP / 32],
                                              char* smem, int opix_off, int m0, int co0, int py, int px,
                                              const uint4_t* rpre = nullptr) {
  constexpr int WT_P = BP / WGP, WT_C = BC / WGC;
  constexpr int MT_P = WT_P / 32, MT_C = WT_C / 32;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wp = wave % WGP, wc = wave / WGP;
  const int l31 = lane & 31, lhi = lane >> 5;
  // ---- epilogue A (fp16 NHWC fast path): transpose through LDS -> 16-byte coalesced HBM traffic ------
  if constexpr (sizeof(T) == 2) {
    if (p.epi_lds) {
      constexpr int NCH = BC / 8;         // 16-byte chunks per output-tile row
      constexpr int ROWB = BC * 2;        // bytes per output-tile row
      char* s_tile = smem;                // the K-loop stages are dead (last loop iteration ended on a barrier)
      long long* s_opix = reinterpret_cast<long long*>(smem + opix_off);
      if constexpr (!PRE) {
        conv_row_table<BP>(p, s_opix, m0, py, px);
        __syncthreads();
      }
      if (p.res) {
        if constexpr (PRE) {
#pragma unroll
          for (int k = 0; k < BP * NCH / NT; ++k) {
            const int idx = tid + k * NT;
            const int pl = idx / NCH, ch = idx % NCH;
            *reinterpret_cast<uint4_t*>(s_tile + pl * ROWB + ((ch ^ (pl & (NCH - 1))) << 4)) = rpre[k];
          }
        } else {
          const half_t* rbase = reinterpret_cast<const half_t*>(p.res) + p.res_coff + co0;
          for (int idx = tid; idx < BP * NCH; idx += NT) {
            const int pl = idx / NCH, ch = idx % NCH;
            const long long o = s_opix[pl];
            uint4_t v = {0u, 0u, 0u, 0u};
            if (o >= 0 && co0 + ch * 8 < p.Cout) v = *reinterpret_cast<const uint4_t*>(rbase + o * p.res_cstride + ch * 8);
            *reinterpret_cast<uint4_t*>(s_tile + pl * ROWB + ((ch ^ (pl & (NCH - 1))) << 4)) = v;
          }
        }
        __syncthreads();
      }
      {
        // channel groups outside, pixel tiles inside: the folded-BN vectors of a group are loaded (and the `scale` / `shift`
        // pointers tested) once per group instead of once per (group, pixel tile); an absent scale / shift is 1 / 0.  Scale,
        // then shift, each rounded (no contraction): the fused stem + pool kernel reproduces these bits.
#pragma clang fp contract(off)
        const float4_t one4 = {1.f, 1.f, 1.f, 1.f}, zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MT_C; ++i) {
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int cl = wc * WT_C + i * 32 + 8 * rg + 4 * lhi;  // channel inside the tile
            const int cb = co0 + cl;
            const float4_t sc = p.scale ? *reinterpret_cast<const float4_t*>(p.scale + cb) : one4;
            const float4_t sh = p.shift ? *reinterpret_cast<const float4_t*>(p.shift + cb) : zero4;
#pragma unroll
            for (int j = 0; j < MT_P; ++j) {
              const int pl = wp * WT_P + j * 32 + l31;
              float v[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = acc[i][j][rg * 4 + e] * sc[e];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = v[e] + sh[e];
              half_t* sp = reinterpret_cast<half_t*>(s_tile + pl * ROWB + (((cl >> 3) ^ (pl & (NCH - 1))) << 4) + lhi * 8);
              if (p.res) {
                float r4[4];
                load4(sp, r4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += r4[e];
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act, p.slope);
              store4(sp, v);
            }
          }
        }
      }
      __syncthreads();
      if (p.tail_w) {
        // fused tail 1x1 conv on the LDS-resident tile (BC == Cout: every channel of these pixels is here):
        // out2[co2][pix] = sum_c Wt[co2][c] * tile[pix][c]; one 32-pixel group per wave, K = BC in steps of 16
        // tail weights arrive as an fp16 hi / lo pair (w = hi + lo to ~22 bits): the heatmap conv runs on fp32-grade
        // weights, two MFMAs per K slice (the head's last step decides the arg-max)
        const half_t* wt = reinterpret_cast<const half_t*>(p.tail_w);
        const float* bt = reinterpret_cast<const float*>(p.tail_w + (size_t)32 * BC * 2);
        const half_t* wl = reinterpret_cast<const half_t*>(p.tail_w + (size_t)32 * BC * 2 + 128);
        for (int g = wave; g < BP / 32; g += NT / 64) {
          const int pl = g * 32 + l31;
          float16_t a2;
#pragma unroll
          for (int r = 0; r < 16; ++r) a2[r] = 0.f;
#pragma unroll 4
          for (int k0 = 0; k0 < BC; k0 += 16) {
            const uint4_t wa = *reinterpret_cast<const uint4_t*>(wt + l31 * BC + k0 + lhi * 8);
            const uint4_t tb = *reinterpret_cast<const uint4_t*>(s_tile + pl * ROWB + ((((k0 >> 3) + lhi) ^ (pl & (NCH - 1))) << 4));
            const uint4_t wb = *reinterpret_cast<const uint4_t*>(wl + l31 * BC + k0 + lhi * 8);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, wa), __builtin_bit_cast(half8_t, tb), a2, 0, 0, 0);
            if (!(p.dbg & 128))
              a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, wb), __builtin_bit_cast(half8_t, tb), a2, 0, 0, 0);
          }
          const long long o = s_opix[pl];
          if (o >= 0) {
            const long long hw = (long long)p.Ho * p.Wo;
            const long long n = o / hw, pix = o - n * hw;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int co2 = (r & 3) + 8 * (r >> 2) + 4 * lhi;
              if (co2 < p.tail_cout) {
                const float v = a2[r] + bt[co2];
                if (p.out_layout == FT_LAYOUT_NHWC)
                  reinterpret_cast<half_t*>(p.y)[o * p.y_cstride + p.y_coff + co2] = (half_t)v;
                else
                  reinterpret_cast<float*>(p.y)[(n * p.tail_cout + co2) * hw + pix] = v;
              }
            }
          }
        }
        return;
      }
      half_t* ybase = reinterpret_cast<half_t*>(p.y) + p.y_coff + co0;
      for (int idx = tid; idx < BP * NCH; idx += NT) {
        const int pl = idx / NCH, ch = idx % NCH;
        const long long o = s_opix[pl];
        if (o >= 0 && co0 + ch * 8 < p.Cout) {
          const uint4_t v = *reinterpret_cast<const uint4_t*>(s_tile + pl * ROWB + ((ch ^ (pl & (NCH - 1))) << 4));
#if FT_EPI_NT && FT_YSTORE_AUX == 0
          __builtin_nontemporal_store(v, reinterpret_cast<uint4_t*>(ybase + o * p.y_cstride + ch * 8));
#else
          store_out16(ybase + o * p.y_cstride + ch * 8, v);
#endif
        }
      }
      return;
    }
  }

  // ---- epilogue B (general): scale/shift (+residual) + activation, NHWC runs of 4 or NCHW fp32 ------
#pragma unroll
  for (int j = 0; j < MT_P; ++j) {
    const int m = m0 + wp * WT_P + j * 32 + l31;
    if (m >= p.M) continue;
    const int n = m / p.HqWq;
    const int rem = m - n * p.HqWq;
    const int qy = rem / p.Wq;
    const int qx = rem - qy * p.Wq;
    const int oy = qy * p.omul + py, ox = qx * p.omul + px;
    const size_t opix = ((size_t)n * p.Ho + oy) * p.Wo + ox;
#pragma unroll
    for (int i = 0; i < MT_C; ++i) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int cb = co0 + wc * WT_C + i * 32 + 8 * rg + 4 * lhi;
        if (cb >= p.Cout) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][rg * 4 + e];
        if (p.scale) {
          const float4_t s = *reinterpret_cast<const float4_t*>(p.scale + cb);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= s[e];
        }
        if (p.shift) {
          const float4_t s = *reinterpret_cast<const float4_t*>(p.shift + cb);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += s[e];
        }
        const bool full = cb + 3 < p.Cout;
        if (p.res) {
          const T* rp = reinterpret_cast<const T*>(p.res) + opix * p.res_cstride + p.res_coff + cb;
          if (full) {
            float r4[4];
            load4(rp, r4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += r4[e];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (cb + e < p.Cout) v[e] += (float)rp[e];
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act, p.slope);
        if (p.out_layout == FT_LAYOUT_NHWC) {
          T* yp = reinterpret_cast<T*>(p.y) + opix * p.y_cstride + p.y_coff + cb;
          if (full) {
            store4(yp, v);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (cb + e < p.Cout) yp[e] = (T)v[e];
          }
        } else {
          float* yp = reinterpret_cast<float*>(p.y);
          const size_t hw = (size_t)p.Ho * p.Wo;
          const size_t pix = (size_t)oy * p.Wo + ox;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (cb + e < p.Cout) yp[((size_t)n * p.Cout + cb + e) * hw + pix] = v[e];
        }
      }
    }
  }
}

// conv_igemm8.hip: the 256 x 256 tile on the 8-phase ping-pong schedule, persistent (fp16, Cin % 64 == 0, Cout_pad % 256 == 0,
// no second input, no residual; output = NHWC fp16 in 8-channel-aligned views, the fused tail's map, or split-K partials);
// p.kc / p.nk count 64-channel K-tiles, `tiles` = p.npt * p.nct * p.nph * p.sk; one workgroup of 512 threads per CU walks them.
int launch_igemm8(const ConvParams& p, unsigned tiles, hipStream_t s);

}  // namespace ft
