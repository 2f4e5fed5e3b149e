// FlowNet2's three custom operators, forward only (the reference ships them as CUDA-only cffi
// extensions; its CPU entry points are empty stubs, correlation_package/src/correlation.c:3-33):
//   Correlation   correlation_package/src/correlation_cuda_kernel.cu:10-106
//   Resample2d    resample2d_package/src/Resample2d_kernel.cu:20-66
//   ChannelNorm   channelnorm_package/src/ChannelNorm_kernel.cu:19-51
// plus the fused warp/diff/norm/concat stage that sits between stacked FlowNets (models.py:396-403).
// The reference's NCHW->padded-NHWC copy kernels (`channels_first`) do not exist here: padding is a
// bounds test, and the in-network form reads the NHWC activations the conv stack already produces.
#include <stdlib.h>

#include <type_traits>

#include "ft_common.h"

namespace ft {

// ---- Correlation, reference-compatible API: NCHW fp32, any parameters ---------------------------
// One thread per output element, lanes along x so both feature reads are coalesced; the sum runs over
// (j, i, c) of the kernel window exactly as correlation_cuda_kernel.cu:75-88 (fp32 accumulation,
// different association order: within fp32 round-off, tolerance stated in the tests).
__global__ __launch_bounds__(256) void correlation_nchw_kernel(const float* __restrict__ in1, const float* __restrict__ in2,
                                                               float* __restrict__ out, int C, int H, int W, int oc, int oh,
                                                               int ow, int pad, int ksize, int max_disp, int s1, int s2,
                                                               size_t total) {
  const int krad = (ksize - 1) / 2;
  const int drad = max_disp / s2;
  const int D = 2 * drad + 1;
  const float nelems = (float)(ksize * ksize * C);
  const size_t HW = (size_t)H * W;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % ow);
    size_t t = idx / ow;
    const int y = (int)(t % oh);
    t /= oh;
    const int tc = (int)(t % oc);
    const size_t n = t / oc;
    const int tj = tc / D - drad, ti = tc % D - drad;
    // coordinates in the zero-padded frame of the reference, shifted back by pad
    const int y1 = y * s1 + max_disp + krad - pad, x1 = x * s1 + max_disp + krad - pad;
    const int y2 = y1 + tj * s2, x2 = x1 + ti * s2;
    const float* p1 = in1 + n * C * HW;
    const float* p2 = in2 + n * C * HW;
    float acc = 0.f;
    for (int j = -krad; j <= krad; ++j) {
      const int ya = y1 + j, yb = y2 + j;
      if ((unsigned)ya >= (unsigned)H || (unsigned)yb >= (unsigned)H) continue;  // zero padding
      for (int i = -krad; i <= krad; ++i) {
        const int xa = x1 + i, xb = x2 + i;
        if ((unsigned)xa >= (unsigned)W || (unsigned)xb >= (unsigned)W) continue;
        const float* a = p1 + (size_t)ya * W + xa;
        const float* b = p2 + (size_t)yb * W + xb;
        for (int c = 0; c < C; ++c) acc += a[c * HW] * b[c * HW];
      }
    }
    out[idx] = acc / nelems;
  }
}

// ---- Correlation, in-network form: NHWC features, kernel_size 1, stride1 1, pad = max_disp --------
// out[pix, coff + (dy+r)*D + (dx+r)] = act( 1/C * sum_c f1[pix, c] * f2[pix + s2*(dy,dx), c] )
// Workgroup = TX consecutive pixels of one image row.  The f1 row tile stays in LDS for the whole
// workgroup; for each of the D displaced rows the (TX + 2*max_disp)-pixel f2 window is staged once in
// LDS and reused by all D horizontal displacements (the reference re-reads both operands from global
// for every displacement with one 32-thread block per pixel, correlation_cuda_kernel.cu:69-101).
// Thread = (pixel x, group of GX horizontal displacements): the f1 chunk is loaded once per GX dots.
// Results collect in an LDS tile and leave as contiguous D*D-channel runs per pixel.
template <typename T, int TX, int GX, int NT>
__global__ __launch_bounds__(NT) void correlation_nhwc_kernel(const T* __restrict__ f1, const T* __restrict__ f2,
                                                               T* __restrict__ y, int C, int H, int W, int max_disp,
                                                               int s2, int f_cstride, int y_cstride, int y_coff, int act,
                                                               float slope) {
  constexpr int VEC = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int drad = max_disp / s2;
  const int D = 2 * drad + 1;
  const int DD = D * D;
  const int win = TX + 2 * max_disp;           // f2 pixels per displaced row
  const int pstride = C * (int)sizeof(T) + 16;  // LDS bytes per pixel (+16: bank spread)
  char* s_f1 = smem;
  char* s_f2 = s_f1 + TX * pstride;
  float* s_out = reinterpret_cast<float*>(s_f2 + win * pstride);  // [TX][DD]

  const int tilesx = (W + TX - 1) / TX;
  const int bx = blockIdx.x % tilesx;
  const int yrow = blockIdx.x / tilesx;
  const int n = blockIdx.y;
  const int x0 = bx * TX;
  const int cvecs = C / VEC;
  const T* f1row = f1 + ((size_t)n * H + yrow) * W * f_cstride;

  for (int i = threadIdx.x; i < TX * cvecs; i += NT) {
    const int px = i / cvecs, cv = i - px * cvecs;
    uint4_t v = {0u, 0u, 0u, 0u};
    if (x0 + px < W) v = *reinterpret_cast<const uint4_t*>(f1row + (size_t)(x0 + px) * f_cstride + cv * VEC);
    *reinterpret_cast<uint4_t*>(s_f1 + px * pstride + cv * 16) = v;
  }

  const int ngx = (D + GX - 1) / GX;  // displacement groups per pixel
  const int nwork = TX * ngx;
  const float inv_c = 1.0f / (float)C;

  for (int dyi = 0; dyi < D; ++dyi) {
    const int y2 = yrow + (dyi - drad) * s2;
    const bool row_ok = (unsigned)y2 < (unsigned)H;
    __syncthreads();  // previous row's readers are done with s_f2 (and s_f1 is written, first time)
    if (row_ok) {
      const T* f2row = f2 + ((size_t)n * H + y2) * W * f_cstride;
      for (int i = threadIdx.x; i < win * cvecs; i += NT) {
        const int px = i / cvecs, cv = i - px * cvecs;
        const int x2 = x0 - max_disp + px;
        uint4_t v = {0u, 0u, 0u, 0u};
        if ((unsigned)x2 < (unsigned)W) v = *reinterpret_cast<const uint4_t*>(f2row + (size_t)x2 * f_cstride + cv * VEC);
        *reinterpret_cast<uint4_t*>(s_f2 + px * pstride + cv * 16) = v;
      }
    }
    __syncthreads();
    for (int wk = threadIdx.x; wk < nwork; wk += NT) {
      const int px = wk % TX, g = wk / TX;
      float acc[GX];
#pragma unroll
      for (int e = 0; e < GX; ++e) acc[e] = 0.f;
      if (row_ok) {
        const char* a = s_f1 + px * pstride;
        for (int cv = 0; cv < cvecs; ++cv) {
          const uint4_t av = *reinterpret_cast<const uint4_t*>(a + cv * 16);
#pragma unroll
          for (int e = 0; e < GX; ++e) {
            const int dxi = g * GX + e;
            if (dxi < D) {
              const int p2 = px + max_disp + (dxi - drad) * s2;  // index inside the staged window
              const uint4_t bv = *reinterpret_cast<const uint4_t*>(s_f2 + p2 * pstride + cv * 16);
              if constexpr (sizeof(T) == 2) {
                const half8_t ah = __builtin_bit_cast(half8_t, av), bh = __builtin_bit_cast(half8_t, bv);
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[e] += (float)ah[q] * (float)bh[q];
              } else {
                const float4_t af = __builtin_bit_cast(float4_t, av), bf = __builtin_bit_cast(float4_t, bv);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[e] += af[q] * bf[q];
              }
            }
          }
        }
      }
#pragma unroll
      for (int e = 0; e < GX; ++e) {
        const int dxi = g * GX + e;
        if (dxi < D) {
          float v = acc[e] * inv_c;
          if (act == FT_ACT_RELU) v = v > 0.f ? v : 0.f;
          else if (act == FT_ACT_LEAKY) v = v > 0.f ? v : v * slope;
          s_out[px * DD + dyi * D + dxi] = v;
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TX * DD; i += NT) {
    const int px = i / DD, ch = i - px * DD;
    if (x0 + px < W)
      y[(((size_t)n * H + yrow) * W + x0 + px) * y_cstride + y_coff + ch] = (T)s_out[i];
  }
}

// ---- Correlation on the matrix cores (fp16 features, kernel 1, stride1 1, stride2 2) ------------------
// stride2 = 2 means an output pixel only ever meets f2 pixels of ITS OWN column parity, so per image row y,
// displaced row y2 = y + 2*dy and parity p the whole cost-volume slice is one small GEMM
//     G[x', x2'] = sum_c f1[y, 2x'+p, c] * f2[y2, 2x2'+p, c],      out[y, 2x'+p, (dy, dx)] = G[x', x'+dx] / C
// of which the band |dx| <= drad is kept.  Workgroup = (image, row, 64-pixel column chunk); wave = (parity,
// 32-column tile of the f2 window).  The f1 fragments (32 pixels x C channels) live in REGISTERS for all D
// displaced rows; each f2 row window (64 + 4*drad pixels) is DMA'd into a 2-slot LDS ring
// (buffer_load ... lds, out-of-image pixels / rows are out-of-range offsets -> zeros) and consumed by
// KS = C/16 back-to-back v_mfma_f32_32x32x16_f16.  MFMA does 64/(2*drad+1) ~ 3x the useful MACs but at
// ~16x the fp32 VALU rate; the band leaves as runs of up to D contiguous channels per pixel.
template <int KS>
__global__ __launch_bounds__(256) void correlation_mfma_kernel(const half_t* __restrict__ f1, const half_t* __restrict__ f2,
                                                               half_t* __restrict__ y, int H, int W, int drad,
                                                               unsigned f2_bytes, int f_cstride, int y_cstride,
                                                               int y_coff, int act, float slope) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int C = KS * 16;
  constexpr int ROWB = C * 2;               // bytes per window pixel in LDS
  constexpr int RPI = 1024 / ROWB;          // window pixels per 1-KiB wave load
  constexpr int CHUNKS = ROWB / 16;
  static_assert(ROWB <= 1024 && CHUNKS >= 16, "C in {128, 256, 512}");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int par = wave & 1, jt = wave >> 1;
  const int c = lane & 31, h = lane >> 5;
  const int x0 = blockIdx.x * 64, yrow = blockIdx.y, n = blockIdx.z;
  const int D = 2 * drad + 1;
  const int wrows = 64 + 4 * drad;          // f2 window: pixels x0 - 2*drad ... x0 + 63 + 2*drad
  const int stage = ((wrows * ROWB + 1023) / 1024) * 1024;
  const float inv_c = 1.0f / (float)C;

  // f1 fragments (operand A: row = f1 pixel of this parity, k = channel)
  uint4_t a[KS];
  {
    const int x = x0 + 2 * c + par;
    const half_t* src = f1 + ((size_t)(n * H + yrow) * W + x) * f_cstride + h * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      uint4_t v = {0u, 0u, 0u, 0u};
      if (x < W) v = *reinterpret_cast<const uint4_t*>(src + s * 16);
      a[s] = v;
    }
  }
  // operand B: column = window pixel 64*jt + 2*c + par (clamped: columns past the window are never in the band)
  int wr = 64 * jt + 2 * c + par;
  wr = wr < wrows ? wr : wrows - 1;
  const int b_base = wr * ROWB;
  const int b_key = (wr >> 1) & 15;

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(f2), 0, f2_bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;
  const int ninstr = stage / 1024;
  auto issue = [&](int dyi, int slot) {
    const int y2 = yrow + 2 * (dyi - drad);
    const bool row_ok = dyi < D && (unsigned)y2 < (unsigned)H;
    for (int i = wave; i < ninstr; i += 4) {
      const int row = i * RPI + lane / CHUNKS, pos = lane % CHUNKS;
      const int lc = pos ^ ((row >> 1) & 15);
      const int x2 = x0 - 2 * drad + row;
      const bool ok = row_ok && row < wrows && (unsigned)x2 < (unsigned)W;
      const unsigned voff = ok ? (unsigned)((((n * H + y2) * W + x2) * f_cstride) * 2 + lc * 16) : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(smem + slot * stage + i * 1024), 16, voff, 0, 0, 0);
    }
  };

  issue(0, 0);
  for (int dyi = 0; dyi < D; ++dyi) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of row dyi has landed (and its last stores left)
    __builtin_amdgcn_s_barrier();                       // everyone's has; everyone is done reading the other slot
    issue(dyi + 1, (dyi + 1) & 1);
    const int y2 = yrow + 2 * (dyi - drad);
    float16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if ((unsigned)y2 < (unsigned)H) {
      const char* st = smem + (dyi & 1) * stage + b_base;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const uint4_t b = *reinterpret_cast<const uint4_t*>(st + (((2 * s + h) ^ b_key) << 4));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, a[s]), __builtin_bit_cast(half8_t, b), acc, 0, 0, 0);
      }
    }
    // band extraction: accumulator row r = f1 pixel x' (this parity), column c = window column -> dx index c + 32*jt - r
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int r = (g & 3) + 8 * (g >> 2) + 4 * h;
      const int x = x0 + 2 * r + par;
      const int dxi = c + 32 * jt - r;
      if (x < W && (unsigned)dxi < (unsigned)D) {
        float v = acc[g] * inv_c;
        if (act == FT_ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (act == FT_ACT_LEAKY) v = v > 0.f ? v : v * slope;
        y[((size_t)(n * H + yrow) * W + x) * y_cstride + y_coff + dyi * D + dxi] = (half_t)v;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the (all out-of-range) look-ahead load of the last iteration
#endif
}

// ---- the same band GEMM with R output rows per workgroup -------------------------------------------------------
// stride2 = 2 also ties the ROWS together: output row y only meets f2 rows of its own parity, and an f2 row is
// shared by the D output rows around it.  A workgroup therefore owns R consecutive output rows of one row-parity class
// (y = 2i + q, i = i0 .. i0+R-1) of a 64-pixel column chunk, keeps their f1 fragments in REGISTERS (R x C/16 x 4
// VGPRs) and streams the R + 2*drad f2 rows of that class through a 3-slot LDS ring ONCE: every f2 row feeds up to R
// band products.  At R = 3 that is 23 row loads for 63 products instead of 63 (3.6 x fewer L2 -> LDS bytes, and 2-3
// products of matrix work per barrier instead of one).  The ring runs two rows ahead on a counted vmcnt: the band
// leaves as unconditional 2-byte buffer stores (lanes outside the band / the image store to an out-of-range offset), so
// the number of vector-memory operations between a row load and its wait is a compile-time constant.
template <int KS, int R, int DRAD>
__global__ __launch_bounds__(256, 1) void correlation_mfma_rows_kernel(const half_t* __restrict__ f1, const half_t* __restrict__ f2,
                                                                        half_t* __restrict__ y, int H, int W, unsigned f2_bytes,
                                                                        unsigned y_bytes, int f_cstride, int y_cstride, int y_coff,
                                                                        int act, float slope, int ngx, int ngy) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int C = KS * 16, ROWB = C * 2, RPI = 1024 / ROWB, CHUNKS = ROWB / 16;
  constexpr int D = 2 * DRAD + 1, WROWS = 64 + 4 * DRAD;
  constexpr int STAGE = (WROWS * ROWB + 1023) / 1024 * 1024, NINSTR = STAGE / 1024, NL = NINSTR / 4;
  constexpr int NJ = R + 2 * DRAD;          // f2 rows of the class this workgroup walks
  static_assert(ROWB <= 1024 && CHUNKS >= 16 && NINSTR % 4 == 0 && 3 * STAGE <= 160 * 1024, "shape");
  static_assert(NL + 16 * R <= 63, "vmcnt immediate");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int par = wave & 1, jt = wave >> 1;
  const int c = lane & 31, h = lane >> 5;
  // workgroup b runs on XCD b % 8 (each with its own L2): hand every XCD a contiguous range of (image, row group) pairs,
  // so the f2 rows its workgroups share are fetched into ONE L2 instead of all eight (measured: 139 -> ~30 MB of f2 reads)
  int x0, n, q, i0;
  {
    const int total = gridDim.x, b = blockIdx.x;
    const int qq = total >> 3, rr = total & 7, xcd = b & 7, loc = b >> 3;
    int logical = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + loc;
    const int gy = logical % ngy;
    logical /= ngy;
    x0 = (logical % ngx) * 64;
    n = logical / ngx;
    q = gy & 1;
    i0 = (gy >> 1) * R;
  }
  const int Hq = (H - q + 1) >> 1;          // rows of this parity class
  const float inv_c = 1.0f / (float)C;

  // f1 fragments of the R rows (operand A: row = f1 pixel of this column parity, k = channel)
  constexpr unsigned kOOB = 0x80000000u;
  uint4_t a[R][KS];
  {
    // buffer loads: pixels / rows outside the image are out-of-range offsets (zeros), no branches, all in flight at once
    const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(f1), 0, f2_bytes, 0x00020000);
    const int x = x0 + 2 * c + par;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int yy = 2 * (i0 + r) + q;
      const unsigned voff = (x < W && yy < H) ? (unsigned)((((n * H + yy) * W + x) * f_cstride + h * 8) * 2) : kOOB;
#pragma unroll
      for (int s = 0; s < KS; ++s) a[r][s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc1, voff, s * 32, 0);
    }
  }
  // operand B: column = window pixel 64*jt + 2*c + par (clamped: columns past the window are never in the band)
  int wr = 64 * jt + 2 * c + par;
  wr = wr < WROWS ? wr : WROWS - 1;
  const int b_base = wr * ROWB;
  const int b_key = (wr >> 1) & 15;

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(f2), 0, f2_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(y, 0, y_bytes, 0x00020000);
  // loader lanes: wave-load i covers window rows i*RPI .. ; XOR swizzle on the source chunk as in correlation_mfma_kernel
  unsigned l_voff[NL];
#pragma unroll
  for (int t = 0; t < NL; ++t) {
    const int i = t * 4 + wave;
    const int row = i * RPI + lane / CHUNKS, pos = lane % CHUNKS;
    const int lc = pos ^ ((row >> 1) & 15);
    const int x2 = x0 - 2 * DRAD + row;
    l_voff[t] = (row < WROWS && (unsigned)x2 < (unsigned)W) ? (unsigned)(((n * H) * W + x2) * f_cstride * 2 + lc * 16) : kOOB;
  }
  const int row_bytes = W * f_cstride * 2;
  auto issue = [&](int jj, int slot) {       // always NL loads per wave: rows outside the image / past the walk are out of range
    const int j = i0 - DRAD + jj;
    const bool row_ok = jj < NJ && (unsigned)j < (unsigned)Hq;
    const int soff = row_ok ? (2 * j + q) * row_bytes : 0;
#pragma unroll
    for (int t = 0; t < NL; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(smem + slot * STAGE + (t * 4 + wave) * 1024), 16,
                                               row_ok ? l_voff[t] : kOOB, soff, 0, 0);
  };
  // band extraction: accumulator register g of lane (c, h) = f1 pixel rr(g) x window column c -> displacement c + 32*jt - rr
  unsigned s_voff[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int rr = (g & 3) + 8 * (g >> 2) + 4 * h;
    const int x = x0 + 2 * rr + par;
    const int dxi = c + 32 * jt - rr;
    s_voff[g] = (x < W && (unsigned)dxi < (unsigned)D) ? (unsigned)(((n * H) * W + x) * y_cstride + y_coff + dxi) * 2u : kOOB;
  }
  const int yrow_bytes = W * y_cstride * 2;
  // act(v) = max(v, s*v) for s in [0, 1] (relu: 0, leaky: slope, none: 1); the 1/C of the correlation rides along
  const float k_pos = inv_c, k_neg = inv_c * (act == FT_ACT_RELU ? 0.f : (act == FT_ACT_LEAKY ? slope : 1.f));

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the f1 fragments sit in registers before the ring starts counting ..
  // .. and the compiler has to know it: it cannot see the wait above, and with the products behind branches it kept "a[] may still be
  // in flight" alive at every join, i.e. s_waitcnt vmcnt(15) .. vmcnt(0) in front of the sixteen MFMAs of EVERY step — the last one
  // drains the whole queue (look-ahead ring rows and the band stores of the previous steps).  An empty asm that redefines the
  // registers ends that (round 6; found in the ISA of the DIRECT form: 23 x sixteen descending waits).
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(a[r][s]));
  issue(0, 0);
  issue(1, 1);
  int slot = 0;
  for (int jj = 0; jj < NJ; ++jj) {
    // row jj has landed (this wave's share): behind it only row jj+1 and the 16*R stores of the previous step may fly
    if (jj == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL + 16 * R) : "memory");
    asm volatile("s_barrier" ::: "memory");             // everyone's share has; everyone is done reading the slot refilled now
    issue(jj + 2, slot == 0 ? 2 : slot - 1);
    const int j = i0 - DRAD + jj;
    const bool row_ok = (unsigned)j < (unsigned)Hq;
    const char* st = smem + slot * STAGE + b_base;
    uint4_t b[KS];
    if (row_ok) {
#pragma unroll
      for (int s = 0; s < KS; ++s) b[s] = *reinterpret_cast<const uint4_t*>(st + (((2 * s + h) ^ b_key) << 4));
    }
    // all R band products first (independent accumulators keep the matrix pipe busy), then their stores
    float16_t acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
      const int dyi = jj - r;                           // f2 row j is displacement dyi - DRAD of output row i0 + r
      if ((unsigned)dyi < (unsigned)D && i0 + r < Hq && row_ok) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
          acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, a[r][s]), __builtin_bit_cast(half8_t, b[s]), acc[r], 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int dyi = jj - r;
      const bool live = (unsigned)dyi < (unsigned)D && i0 + r < Hq;
      if (live) {                                        // wave-uniform
        const int soff = (2 * (i0 + r) + q) * yrow_bytes + dyi * D * 2;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const float v = __builtin_fmaxf(acc[r][g] * k_pos, acc[r][g] * k_neg);   // act(v / C), slopes in [0, 1]
          const half_t hv = (half_t)v;
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hv), rsrc_y, s_voff[g], soff, 0);
        }
      } else {                                           // same count of vector-memory operations, nothing stored
#pragma unroll
        for (int g = 0; g < 16; ++g) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)0, rsrc_y, kOOB, 0, 0);
      }
    }
    slot = slot == 2 ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the two (all out-of-range) look-ahead rows of the last steps
#endif
}

// ---- the rows kernel for maps up to 64 pixels wide ---------------------------------------------------------------------
// Shipped form (STAGED = false, NRS = 2): 64-column ring slots, EIGHT waves — R = 4 output rows per workgroup, the upper
// two owned by waves 4..7 with their own f1 fragments on the same ring — and the rows kernel's 2-byte band stores:
// 61.9-63.5 us at [16,256,48,64] against 65.6-67.3 for the 104-column, four-wave rows kernel (same box).
// The STAGED form below was the attempt to get rid of the 1008 two-byte stores per lane; it is correct and slower:
// At W <= 64 (FlowNetC at 512x384: 48 x 64) only 64 of the 104 window columns of an f2 row exist, so a ring slot shrinks
// from 52 to 32 KiB (window columns outside the image read one shared zero row) and the 64 KiB that frees hold a TRANSPOSE
// tile per output row: the band products of GD = 8 consecutive dy displacements are written (ds_write_b16, band lanes
// only) into [64 pixels][8 x 21] fp16 = 336 contiguous, 16-byte-aligned bytes of each pixel's 441-channel run, and leave
// as whole 16-byte pieces (21 per pixel) when the group is complete: ~50 b128 stores per thread instead of 1008 two-byte
// stores per lane, at the price of one extra barrier per flush (9 per workgroup).  The last group holds 5 x 21 values =
// 210 bytes: thirteen pieces and one 2-byte store.  Same MFMA order and fp32 -> fp16 step as the rows kernel.
// Every step waits with vmcnt(NL): loads return in order among loads, so "at most NL vector-memory operations in flight"
// means row jj has landed whatever the stores of the previous flush are doing.
template <int N, int I = 0, typename F>
__device__ __forceinline__ void corr_unroll(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    corr_unroll<N, I + 1>(f);
  }
}

// STAGED = false keeps the rows kernel's 2-byte band stores (and its store-counting waits) on the narrow ring: measured at
// [16,256,48,64], the transpose costs more in ds_write_b16 issue than it saves in stores (75 us staged; 86 us with a branch
// per band register instead of the scratch-slot select; 63.5 us unstaged with four waves, 62 us with eight).  Pulling the f2
// window straight into registers instead of through the ring (each wave uses its own 32 window pixels) was also tried with
// the staged band: 87 us.
// NRS = 2: eight waves, the second four take the upper half of the R output rows with their own f1 fragments and read the
// same ring: two waves per SIMD, one's band stores issue while the other's products run on the matrix pipe.
typedef uint32_t corr_u2_t __attribute__((ext_vector_type(2)));
// TR (round 4): the products TRANSPOSED — f2 window columns as the MFMA's rows, f1 pixels as its columns — so that a lane owns ONE
// output pixel and its accumulator registers walk the displacement axis: the four registers of a group are four CONSECUTIVE
// x-displacements = 8 contiguous output bytes.  A group that lies wholly inside the band leaves as one 8-byte store (2-byte
// aligned: the 441-channel rows have no better alignment), a group cut by the band's end as up to three 2-byte stores: 7.5 lane
// stores per pixel and y-displacement on average instead of 21 — the kernel is bound by the issue of scattered lane stores.
// DIRECT (round 4, second form): the tile's columns are the IMAGE columns of this parity (32 of them at W <= 64) instead of two
// 32-column halves of the 104-column window — half the MFMAs and half the band store instructions per (f2 row, output row).
// The band of f1 pixel i then is f2 pixel i2 = i + dxi - DRAD in [0, 32); the displacements that fall off the image
// (i2 < 0 or >= 32: structural zeros the window form produced from its zero columns) are written by the lanes of the
// columns i2 mod 32, which are out of band for that row: every lane slot with (c - i + DRAD) mod 32 < D stores, a real product
// where the difference did not wrap, zero where it did.  NWV waves = NWV / 2 output rows x 2 column parities, one row per wave
// (six waves / R = 3 give 256 workgroups at 16 x 48 rows, eight / R = 4 give 192).  62.4 -> 50.2-51.2 us at [16,256,48,64] (same box).
// Ablations of this form (FT_CORR_DBG, same box): full 54.2 us; no band store instructions 37.5 (stores kept but all out of range:
// 40.8 of 50.2); also no MFMAs 27.2; also no fragment reads 26.9; also no ring loads 20.4; also no barriers 17.7 (launch + the f1
// fragments + 23 empty steps).  Three rewrites of the band's way out, each correct, none faster: (a) pairs of lanes packed into
// aligned 4-byte stores (ds_bpermute for the neighbour, the odd element carried to the next dy): 70.0 us; (b) the band through
// wave-private LDS tiles [32 px][8 dy x 21 dx] and out as 16-byte pieces (30 store instructions per wave instead of 336): 51.2-
// 51.7 us — so it is neither the address unit's instruction rate nor the 2-byte granularity; (c) the epilogue of step jj - 1
// woven between the MFMAs of step jj (one store per MFMA in the ISA): 56.8-57.9 vs 54.8-55.3 us.  What a step costs is spread
// over products, epilogue ALU work, stores and the ring, none of them alone; (a)-(c) were removed again.
template <int KS, int R, int DRAD, bool STAGED, int NRS, bool TR = false, bool DIRECT = false, int NWV = 4 * NRS>
__global__ __launch_bounds__(64 * NWV, 1) void correlation_mfma_rows64_kernel(const half_t* __restrict__ f1, const half_t* __restrict__ f2,
                                                                          half_t* __restrict__ y, int H, int W, unsigned f2_bytes,
                                                                          unsigned y_bytes, int f_cstride, int y_cstride, int y_coff,
                                                                          int act, float slope, int ngy) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int C = KS * 16, ROWB = C * 2;
  constexpr int D = 2 * DRAD + 1, WROWS = 64 + 4 * DRAD;
  constexpr int NJ = R + 2 * DRAD;          // f2 rows of the class this workgroup walks
  constexpr int NSETS = DIRECT ? NWV / 2 : NRS;   // row sets among the waves
  constexpr int RW = R / NSETS;             // output rows per wave
  constexpr int SLOT = 64 * ROWB, NL = (SLOT / 1024 + NWV - 1) / NWV;   // wave-loads per wave and ring row (the last ones may fall behind the slot: scratch)
  static_assert(R % NSETS == 0 && (NRS == 1 || !STAGED) && !(DIRECT && (STAGED || TR)) && (!DIRECT || D <= 32) && NWV % 2 == 0, "row sets");
  // DIRECT: four ring slots, rows issued THREE steps ahead.  vmcnt counts loads and stores in one queue, in order, so "row jj has
  // landed" can only be asked as "everything older than the ops issued after it is done"; at distance three the stores of three
  // steps may still be in flight (vmcnt(2 NL + 48) <= 63).  Measured: no gain over distance two (52.0 vs 51.6 us) — the store
  // round trip is not what a step waits for; kept because it costs nothing.
  constexpr int DEPTH = DIRECT ? 3 : 2, NSLOT = DEPTH + 1;
  constexpr int ZROW = NSLOT * SLOT, STG = ZROW + ROWB;
  constexpr int GD = 8;
  constexpr int PROW = GD * D * 2;          // bytes of a pixel's run per group
  constexpr int TILE = 64 * PROW;           // the transpose tile of one output row
  constexpr int DUMMY = STG + R * TILE;     // 128 bytes: where the lanes outside the band write
  static_assert(PROW % 16 == 0 && (!STAGED || DUMMY + 128 <= 160 * 1024) && ROWB == 512 && (DIRECT || NL * NWV * 1024 == SLOT), "shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int par = wave & 1, jt = DIRECT ? 0 : (wave >> 1) & 1, rs = DIRECT ? wave >> 1 : wave >> 2;
  const int c = lane & 31, h = lane >> 5;
  int n, q, i0;
  {
    const int total = gridDim.x, b = blockIdx.x;
    const int qq = total >> 3, rr = total & 7, xcd = b & 7, loc = b >> 3;
    const int logical = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + loc;
    const int gy = logical % ngy;
    n = logical / ngy;
    q = gy & 1;
    i0 = (gy >> 1) * R;
  }
  const int Hq = (H - q + 1) >> 1;          // rows of this parity class
  const float inv_c = 1.0f / (float)C;
  constexpr unsigned kOOB = 0x80000000u;
  // developer ablation (FT_CORR_DBG, DIRECT form, timing only): 1 = every band store out of range, 2 = no MFMAs, 4 = no fragment
  // reads, 8 = no ring loads, 16 = no barriers, 32 = no band store instructions at all; 0 in production
  const int cdbg = DIRECT ? act >> 8 : 0;
  if constexpr (DIRECT) act &= 0xff;

  // f1 fragments of the R rows (operand A: row = f1 pixel of this column parity, k = channel)
  uint4_t a[RW][KS];
  {
    const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(f1), 0, f2_bytes, 0x00020000);
    const int x = 2 * c + par;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int yy = 2 * (i0 + rs * RW + r) + q;
      const unsigned voff = (x < W && yy < H) ? (unsigned)((((n * H + yy) * W + x) * f_cstride + h * 8) * 2) : kOOB;
#pragma unroll
      for (int s = 0; s < KS; ++s) a[r][s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc1, voff, s * 32, 0);
    }
  }
  // operand B: column = window pixel 64*jt + 2*c + par = image column x2 (ring row x2), the zero row outside the image
  int wr = 64 * jt + 2 * c + par;
  wr = wr < WROWS ? wr : WROWS - 1;
  const int x2 = DIRECT ? 2 * c + par : wr - 2 * DRAD;
  const bool in_img = (unsigned)x2 < (unsigned)W;
  const int b_base = in_img ? x2 * ROWB : ZROW - 0;      // slot-relative for image columns; ZROW is absolute (see b_abs)
  const int b_key = in_img ? (x2 >> 1) & 15 : 0;

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(f2), 0, f2_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(y, 0, y_bytes, 0x00020000);
  // loader lanes: wave-load i = t*4 + wave covers ring rows 2i, 2i+1; XOR swizzle on the source chunk
  unsigned l_voff[NL];
#pragma unroll
  for (int t = 0; t < NL; ++t) {
    const int i = t * NWV + wave;
    const int row = i * 2 + (lane >> 5), pos = lane & 31;
    const int lc = pos ^ ((row >> 1) & 15);
    l_voff[t] = (row < W && i < SLOT / 1024) ? (unsigned)(((n * H) * W + row) * f_cstride * 2 + lc * 16) : kOOB;
  }
  const int row_bytes = W * f_cstride * 2;
  auto issue = [&](int jj, int slot) {       // always NL loads per wave: rows outside the image / past the walk are out of range
    const int j = i0 - DRAD + jj;
    const bool row_ok = jj < NJ && (unsigned)j < (unsigned)Hq;
    const int soff = row_ok ? (2 * j + q) * row_bytes : 0;
    if (cdbg & 8) return;
#pragma unroll
    for (int t = 0; t < NL; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(t * NWV + wave < SLOT / 1024 ? smem + slot * SLOT + (t * NWV + wave) * 1024
                                                                                           : smem + STG + 128), 16,
                                               row_ok ? l_voff[t] : kOOB, soff, 0, 0);
  };
  // band lanes: accumulator register g of lane (c, h) = f1 pixel rr = G(g) + 4h (of this column parity) x window column c
  // -> displacement index dxi = c + 32*jt - rr, kept when 0 <= dxi < D and the pixel lies inside the image
  const int dxi0 = DIRECT ? c - 4 * h + DRAD : c + 32 * jt - 4 * h;
  const int lane_base = STG + (8 * h + par) * PROW + dxi0 * 2;     // byte of (pixel 2*(4h) + par, dxi0) of tile 0
  int vmask = 0, realmask = 0;       // DIRECT: stores / stores that carry a product (the others are the off-image zeros)
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int G = (g & 3) + 8 * (g >> 2);
    const int dr = dxi0 - G, dd = DIRECT ? dr & 31 : dr;
    if ((unsigned)dd < (unsigned)D && 2 * (G + 4 * h) + par < W) vmask |= 1 << g;
    if (dd == dr) realmask |= 1 << g;
  }
  // act(v) = max(v, s*v) for s in [0, 1] (relu: 0, leaky: slope, none: 1); the 1/C of the correlation rides along
  const float k_pos = inv_c, k_neg = inv_c * (act == FT_ACT_RELU ? 0.f : (act == FT_ACT_LEAKY ? slope : 1.f));
  const int yrow_bytes = W * y_cstride * 2;
  unsigned s_voff[(STAGED || TR) ? 1 : 16];      // !STAGED: the band leaves as 2-byte stores in the accumulator layout
  // TR: lane (c, h) = f1 pixel x = 2c + par; register 4 gq + e = window column 8 gq + 4 h + e of this half: displacement index
  // dxi = 32 jt + 8 gq + 4 h + e - c.  t_base = byte offset of (pixel, dxi of gq = e = 0); group gq / element e ride in the
  // instruction's immediate offset.
  const int t_dxi0 = 32 * jt + 4 * h - c;
  const bool t_px = 2 * c + par < W;
  const unsigned t_base = (unsigned)((((n * H) * W + 2 * c + par) * y_cstride + y_coff + t_dxi0) * 2);
  if constexpr (!STAGED && !TR) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int G = (g & 3) + 8 * (g >> 2);
      const int x = 2 * (G + 4 * h) + par;
      const int dd = DIRECT ? (dxi0 - G) & 31 : dxi0 - G;
      s_voff[g] = (((vmask >> g) & 1) && !(cdbg & 1)) ? (unsigned)(((n * H) * W + x) * y_cstride + y_coff + dd) * 2u : kOOB;
    }
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the f1 fragments sit in registers before the ring starts counting ..
  // .. and the compiler has to know it: it cannot see the wait above, and with the products behind branches it kept "a[] may still be
  // in flight" alive at every join, i.e. s_waitcnt vmcnt(15) .. vmcnt(0) in front of the sixteen MFMAs of EVERY step — the last one
  // drains the whole queue (look-ahead ring rows and the band stores of the previous steps).  An empty asm that redefines the
  // registers ends that (round 6; found in the ISA of the DIRECT form: 23 x sixteen descending waits).
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(a[r][s]));
  if (tid < ROWB / 16) *reinterpret_cast<uint4_t*>(smem + ZROW + tid * 16) = uint4_t{0u, 0u, 0u, 0u};
  corr_unroll<DEPTH>([&](auto pc) { issue(decltype(pc)::value, decltype(pc)::value); });
  corr_unroll<NJ>([&](auto jc) {
    constexpr int jj = decltype(jc)::value;
    constexpr int slot = jj % NSLOT;
    // row jj has landed (this wave's share).  !STAGED: behind it row jj+1 and the 16*R band stores of the previous step may fly
    // (DIRECT: rows jj+1, jj+2 and the stores of up to three steps)
    if constexpr (DIRECT) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((DEPTH - 1) * NL + 16 * RW * (jj < DEPTH ? jj : DEPTH)) : "memory");
    else if constexpr (STAGED || jj == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NL) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NL + (TR ? 20 : 16) * RW) : "memory");
    if (!(cdbg & 16)) asm volatile("s_barrier" ::: "memory");   // everyone's has; everyone is done reading the slot refilled now
    issue(jj + DEPTH, (jj + DEPTH) % NSLOT);
    const int j = i0 - DRAD + jj;
    const bool row_ok = (unsigned)j < (unsigned)Hq;
    uint4_t b[KS];
    if (cdbg & 4) {
#pragma unroll
      for (int s = 0; s < KS; ++s) b[s] = uint4_t{(unsigned)jj, 0u, 0u, 0u};
    } else if (row_ok) {
      const char* st = smem + (in_img ? slot * SLOT : 0) + b_base;
#pragma unroll
      for (int s = 0; s < KS; ++s) b[s] = *reinterpret_cast<const uint4_t*>(st + (((2 * s + h) ^ b_key) << 4));
    }
    corr_unroll<RW>([&](auto rc) {
      constexpr int rw = decltype(rc)::value;
      constexpr int r = rw;                              // STAGED (NRS = 1): the row itself
      constexpr int dyi = jj - r;                        // f2 row j is displacement dyi - DRAD of output row i0 + r
      if constexpr (!STAGED) {
        // the rows kernel's form: always 16 stores per (step, row) so that the waits can count them
        const int r = rs * RW + rw;                      // wave-uniform
        const int dyi = jj - r;
        const bool live = dyi >= 0 && dyi < D && i0 + r < Hq;
        if constexpr (TR) {
          // always 4 x (one 8-byte + four 2-byte) stores per (step, row), most of them out of range, so that the waits can count
          if (live) {
            float16_t acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            if (row_ok) {
#pragma unroll
              for (int s = 0; s < KS; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, b[s]), __builtin_bit_cast(half8_t, a[rw][s]), acc, 0, 0, 0);
            }
            const int soff = (2 * (i0 + r) + q) * yrow_bytes + dyi * D * 2;
            corr_unroll<4>([&](auto gc) {
              constexpr int gq = decltype(gc)::value;
              const int d0 = t_dxi0 + 8 * gq;                          // displacement index of element 0
              const bool full = t_px && d0 >= 0 && d0 + 3 < D;
              half_t hv[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) hv[e] = (half_t)__builtin_fmaxf(acc[4 * gq + e] * k_pos, acc[4 * gq + e] * k_neg);
              const half4_t h4 = {hv[0], hv[1], hv[2], hv[3]};
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(corr_u2_t, h4), rsrc_y, full ? t_base + 16 * gq : kOOB, soff, 0);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const bool one = t_px && !full && (unsigned)(d0 + e) < (unsigned)D;
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hv[e]), rsrc_y, one ? t_base + 16 * gq + 2 * e : kOOB, soff, 0);
              }
            });
          } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              __builtin_amdgcn_raw_buffer_store_b64(corr_u2_t{0u, 0u}, rsrc_y, kOOB, 0, 0);
#pragma unroll
              for (int e = 0; e < 4; ++e) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)0, rsrc_y, kOOB, 0, 0);
            }
          }
        } else if (live) {
          float16_t acc;
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[e] = 0.f;
          if (row_ok && !(cdbg & 2)) {
#pragma unroll
            for (int s = 0; s < KS; ++s)
              acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, a[rw][s]), __builtin_bit_cast(half8_t, b[s]), acc, 0, 0, 0);
          }
          const int soff = (2 * (i0 + r) + q) * yrow_bytes + dyi * D * 2;
          if (cdbg & 32) {
            float keep = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) keep += acc[g];
            if (keep == 12345.678f) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)1, rsrc_y, s_voff[0], soff, 0);
          } else {
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            half_t hv = (half_t)__builtin_fmaxf(acc[g] * k_pos, acc[g] * k_neg);
            if constexpr (DIRECT) hv = ((realmask >> g) & 1) ? hv : (half_t)0.f;
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hv), rsrc_y, s_voff[g], soff, 0);
          }
          }
        } else if (!(cdbg & 32)) {
#pragma unroll
          for (int g = 0; g < 16; ++g) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)0, rsrc_y, kOOB, 0, 0);
        }
      } else if constexpr (dyi >= 0 && dyi < D) {
        if (i0 + r < Hq) {                               // workgroup-uniform
          constexpr int gg = dyi / GD, dslot = dyi % GD;
          float16_t acc;
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[e] = 0.f;
          if (row_ok) {
#pragma unroll
            for (int s = 0; s < KS; ++s)
              acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, a[r][s]), __builtin_bit_cast(half8_t, b[s]), acc, 0, 0, 0);
          }
          char* tile = smem + r * TILE + dslot * D * 2 + lane_base;
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            const int G = (g & 3) + 8 * (g >> 2);
            const float v = __builtin_fmaxf(acc[g] * k_pos, acc[g] * k_neg);   // act(v / C), slopes in [0, 1]
            // lanes outside the band write a scratch slot behind the tiles: straight-line code (a branch per register would
            // fence the next product's MFMAs off from these writes)
            char* dst = ((vmask >> g) & 1) ? tile + G * (2 * PROW - 2) : smem + DUMMY + lane * 2;
            *reinterpret_cast<half_t*>(dst) = (half_t)v;
          }
          constexpr bool last_of_group = dslot == GD - 1 || dyi == D - 1;
          if constexpr (last_of_group) {
            constexpr int gbytes = (dyi == D - 1 ? D - gg * GD : GD) * D * 2;   // bytes of the group per pixel
            constexpr int npc = gbytes / 16, tail = gbytes % 16;                 // whole 16-byte pieces, bytes behind them
            static_assert(tail == 0 || tail == 2, "tail");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");         // every wave's band writes of this group sit in the tile
            const char* src = smem + STG + r * TILE;
            const int soff = (2 * (i0 + r) + q) * yrow_bytes + (y_coff + gg * GD * D) * 2;
#pragma unroll
            for (int k = 0; k < (64 * npc + 255) / 256; ++k) {
              const int idx = tid + 256 * k;
              const int px = idx / npc, pc = idx - px * npc;
              const uint4_t v = *reinterpret_cast<const uint4_t*>(src + (px < 64 ? px : 63) * PROW + pc * 16);
              const unsigned vo = (px < 64 && px < W) ? (unsigned)((((n * H) * W + px) * y_cstride) * 2 + pc * 16) : kOOB;
              __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_y, vo + (unsigned)soff, 0, FT_YSTORE_BUF_AUX);
            }
            if constexpr (tail == 2) {
              const int px = tid;
              const unsigned short v = px < 64 ? *reinterpret_cast<const unsigned short*>(src + px * PROW + npc * 16) : (unsigned short)0;
              const unsigned vo = (px < 64 && px < W) ? (unsigned)((((n * H) * W + px) * y_cstride) * 2 + npc * 16) : kOOB;
              __builtin_amdgcn_raw_buffer_store_b16(v, rsrc_y, vo, soff, 0);
            }
          }
        }
      }
    });
  });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the (all out-of-range) look-ahead rows of the last steps
#endif
}

// Workgroup b of a 1-D grid runs on XCD b % 8 and every XCD has its own L2.  The gather kernels below read a 2 x 2
// neighbourhood around a displaced position: with the plain blockIdx order the rows one workgroup touches are also touched
// by its neighbours on seven other XCDs and every L2 fetches them again (PMC: 2.7-2.9 x the algorithmic read bytes).  This
// bijective remap hands each XCD one contiguous eighth of the blocks, i.e. whole images.
__device__ __forceinline__ size_t xcd_contiguous_block() {
  const unsigned total = gridDim.x, b = blockIdx.x;
  const unsigned q = total >> 3, r = total & 7, xcd = b & 7, loc = b >> 3;
  return (size_t)((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc);
}

// ---- Resample2d: backward bilinear warp with border clamp -----------------------------------------
// Weights from the UNclamped floor, neighbour indices clamped, no renormalisation
// (Resample2d_kernel.cu:42-59).  Thread = output pixel, all channels (flow read once).
__global__ __launch_bounds__(256) void resample2d_kernel(const float* __restrict__ in1, const float* __restrict__ flow,
                                                         float* __restrict__ out, int C, int H, int W, size_t total) {
  const size_t HW = (size_t)H * W;
  for (size_t i = xcd_contiguous_block() * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / HW, pix = i - b * HW;
    const int y = (int)(pix / W), x = (int)(pix - (size_t)y * W);
    const float dx = flow[(b * 2 + 0) * HW + pix], dy = flow[(b * 2 + 1) * HW + pix];
    const float xf = (float)x + dx, yf = (float)y + dy;
    const float fx = floorf(xf), fy = floorf(yf);
    const float alpha = xf - fx, beta = yf - fy;
    // clamp in float first so huge/inf flows cannot overflow the int conversion
    const int xL = (int)fminf(fmaxf(fx, 0.f), (float)(W - 1));
    const int xR = (int)fminf(fmaxf(fx + 1.f, 0.f), (float)(W - 1));
    const int yT = (int)fminf(fmaxf(fy, 0.f), (float)(H - 1));
    const int yB = (int)fminf(fmaxf(fy + 1.f, 0.f), (float)(H - 1));
    const float w00 = (1.f - alpha) * (1.f - beta), w01 = alpha * (1.f - beta);
    const float w10 = (1.f - alpha) * beta, w11 = alpha * beta;
    for (int c = 0; c < C; ++c) {
      const float* p = in1 + (b * C + c) * HW;
      float v = w00 * p[(size_t)yT * W + xL];
      v += w01 * p[(size_t)yT * W + xR];
      v += w10 * p[(size_t)yB * W + xL];
      v += w11 * p[(size_t)yB * W + xR];
      out[(b * C + c) * HW + pix] = v;
    }
  }
}

// The same with the two x-neighbours of a row fetched as ONE 8-byte load: the kernel is bound by cache-line requests (every
// lane gathers from its own line: 4 taps x C channels = 12 requests per pixel, 38 M per launch at BASELINE configs[3] shapes
// = the measured 58 us at ~1 request per clock per CU), not by bytes and not by the per-thread dependency chain (four pixels
// per thread with all 48 gathers independent measured 70 us).  xL and xR are adjacent except where the clamp folds them onto
// one column, so a pair starting at column min(xL, W - 2) always holds both: 2 requests per row pair instead of 4.
template <int C>
__global__ __launch_bounds__(256) void resample2d_pair_kernel(const float* __restrict__ in1, const float* __restrict__ flow,
                                                              float* __restrict__ out, int H, int W, size_t total) {
  typedef float float2_t __attribute__((ext_vector_type(2), aligned(4)));
  const size_t HW = (size_t)H * W;
  for (size_t i = xcd_contiguous_block() * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / HW, pix = i - b * HW;
    const int y = (int)(pix / W), x = (int)(pix - (size_t)y * W);
    const float dx = flow[(b * 2 + 0) * HW + pix], dy = flow[(b * 2 + 1) * HW + pix];
    const float xf = (float)x + dx, yf = (float)y + dy;
    const float fx = floorf(xf), fy = floorf(yf);
    const float alpha = xf - fx, beta = yf - fy;
    const int xL = (int)fminf(fmaxf(fx, 0.f), (float)(W - 1));
    const int xR = (int)fminf(fmaxf(fx + 1.f, 0.f), (float)(W - 1));
    const int yT = (int)fminf(fmaxf(fy, 0.f), (float)(H - 1));
    const int yB = (int)fminf(fmaxf(fy + 1.f, 0.f), (float)(H - 1));
    const float w00 = (1.f - alpha) * (1.f - beta), w01 = alpha * (1.f - beta);
    const float w10 = (1.f - alpha) * beta, w11 = alpha * beta;
    const int xb = xL < W - 2 ? xL : W - 2;       // pair [xb, xb + 1] holds columns xL and xR
    const bool l1 = xL != xb, r1 = xR != xb;
    float2_t t[C], u[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float* p = in1 + (b * C + c) * HW;
      t[c] = *reinterpret_cast<const float2_t*>(p + (size_t)yT * W + xb);
      u[c] = *reinterpret_cast<const float2_t*>(p + (size_t)yB * W + xb);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float v = w00 * (l1 ? t[c][1] : t[c][0]);
      v += w01 * (r1 ? t[c][1] : t[c][0]);
      v += w10 * (l1 ? u[c][1] : u[c][0]);
      v += w11 * (r1 ? u[c][1] : u[c][0]);
      out[(b * C + c) * HW + pix] = v;
    }
  }
}

// ---- Resample2d through an LDS window (round 4) ---------------------------------------------------------------------------
// The gather kernels above are bound by cache-line REQUESTS, not bytes: with incoherent per-pixel flows every lane's taps sit
// in their own lines (12 requests per pixel at C = 3), 38 M requests per launch at configs[3] shapes.  Here a workgroup owns a
// 16 x 64 tile of output pixels (4 per thread): it reads the tile's flow once, reduces the bounding box of every tap the tile
// touches (clamped indices, so the box is inside the image), loads that window of in1 — all C planes — into LDS with
// row-contiguous, fully coalesced 4-byte-per-lane loads, and serves the four taps of every pixel from LDS.  A tile whose
// window does not fit the LDS budget (a flow field with > ~20 px of spread inside 16 x 64 pixels) takes the direct gathers of
// resample2d_pair_kernel for that tile only (workgroup-uniform branch).  Same weights, same order of the four products as
// Resample2d_kernel.cu:42-59 restated above: results are bit-identical to the gather kernels.
constexpr int kRsTH = 16, kRsTW = 64, kRsPPT = 4;          // tile rows / columns, pixels per thread
#ifndef FT_RS_CLIP_PITCH
#define FT_RS_CLIP_PITCH 96
#endif
constexpr int kRsClipPitch = FT_RS_CLIP_PITCH;                           // row pitch (floats) of a window clipped around its tile: 16 + 2 x 8 rows x 64 + 2 x 15 columns at the default budget
constexpr int kRsMaxWindow = 6656;                         // window floats per plane (row pitch x rows): 3 planes x 26 KiB = 78 KiB, two workgroups per CU

template <int C>
__global__ __launch_bounds__(256) void resample2d_window_kernel(const float* __restrict__ in1, const float* __restrict__ flow,
                                                                float* __restrict__ out, int H, int W, int tiles_x, int tiles_y,
                                                                unsigned in_bytes, int wbudget) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) float rs_smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  __shared__ int s_box[4][4];                              // per wave: min x, max x, min y, max y
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles = tiles_x * tiles_y;
  const int b = blockIdx.x / tiles, trem = blockIdx.x - b * tiles;
  const int ty0 = (trem / tiles_x) * kRsTH, tx0 = (trem % tiles_x) * kRsTW;
  const size_t HW = (size_t)H * W;
  const int x = tx0 + lane;
  float w00[kRsPPT], w01[kRsPPT], w10[kRsPPT], w11[kRsPPT];
  int xL[kRsPPT], xR[kRsPPT], yT[kRsPPT], yB[kRsPPT];
  int bx0 = 0x7fffffff, bx1 = -1, by0 = 0x7fffffff, by1 = -1;
#pragma unroll
  for (int k = 0; k < kRsPPT; ++k) {
    const int y = ty0 + k * 4 + wave;
    const bool live = x < W && y < H;
    float dx = 0.f, dy = 0.f;
    if (live) {
      const size_t pix = (size_t)y * W + x;
      dx = flow[(b * 2 + 0) * HW + pix];
      dy = flow[(b * 2 + 1) * HW + pix];
    }
    const float xf = (float)x + dx, yf = (float)y + dy;
    const float fx = floorf(xf), fy = floorf(yf);
    const float alpha = xf - fx, beta = yf - fy;
    xL[k] = (int)fminf(fmaxf(fx, 0.f), (float)(W - 1));
    xR[k] = (int)fminf(fmaxf(fx + 1.f, 0.f), (float)(W - 1));
    yT[k] = (int)fminf(fmaxf(fy, 0.f), (float)(H - 1));
    yB[k] = (int)fminf(fmaxf(fy + 1.f, 0.f), (float)(H - 1));
    w00[k] = (1.f - alpha) * (1.f - beta); w01[k] = alpha * (1.f - beta);
    w10[k] = (1.f - alpha) * beta;         w11[k] = alpha * beta;
    if (live) {
      bx0 = min(bx0, xL[k]); bx1 = max(bx1, xR[k]);
      by0 = min(by0, yT[k]); by1 = max(by1, yB[k]);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    bx0 = min(bx0, __shfl_xor(bx0, off)); bx1 = max(bx1, __shfl_xor(bx1, off));
    by0 = min(by0, __shfl_xor(by0, off)); by1 = max(by1, __shfl_xor(by1, off));
  }
  if (lane == 0) { s_box[wave][0] = bx0; s_box[wave][1] = bx1; s_box[wave][2] = by0; s_box[wave][3] = by1; }
  __syncthreads();
  bx0 = min(min(s_box[0][0], s_box[1][0]), min(s_box[2][0], s_box[3][0]));
  bx1 = max(max(s_box[0][1], s_box[1][1]), max(s_box[2][1], s_box[3][1]));
  by0 = min(min(s_box[0][2], s_box[1][2]), min(s_box[2][2], s_box[3][2]));
  by1 = max(max(s_box[0][3], s_box[1][3]), max(s_box[2][3], s_box[3][3]));
  if (bx1 < 0) return;                                     // (a tile without a live pixel: cannot happen for ceil-divided grids)
  bx0 = __builtin_amdgcn_readfirstlane(bx0); bx1 = __builtin_amdgcn_readfirstlane(bx1);
  by0 = __builtin_amdgcn_readfirstlane(by0); by1 = __builtin_amdgcn_readfirstlane(by1);
  // 16-byte DMA pieces need 16-byte-aligned sources: the window starts at a multiple of four columns (W % 4 == 0: every row base
  // is aligned); otherwise 4-byte pieces, four times as many instructions
  const bool vec4 = (W & 3) == 0;
  const int pgran = vec4 ? 32 : 64;                        // floats per LDS row: a multiple of 32 (16-byte pieces) / 64 (4-byte pieces)
  int wx0 = vec4 ? bx0 & ~3 : bx0, wx1 = bx1, wy0 = by0, wy1 = by1;
  int ww = wx1 - wx0 + 1, wh = wy1 - wy0 + 1;
  int pitch = (ww + pgran - 1) / pgran * pgran;
  pitch = pitch < 64 ? 64 : pitch;
  bool window = (((long long)pitch * wh + 255) & ~255LL) <= wbudget;     // the whole box fits: every tap comes from LDS
  // Round 6: a box that does not fit no longer sends the whole tile to the gathers.  The window is CLIPPED around the tile (rows of
  // kRsClipPitch floats, as many as the budget holds, centred on the tile's own pixels), pixels whose four taps lie inside it take
  // them from LDS and only the others gather from memory (8-byte x-neighbour pairs, issued before the window's wait): with
  // per-pixel N(0, 4 px) flows about one pixel in eight gathers instead of all of them.  A field that spreads over more than four
  // such windows (the "wide" image of the tests) skips the window.
  const int clip_pitch = vec4 ? kRsClipPitch : 128;        // (4-byte pieces fill 64-float row segments)
  if (!window && ww <= 4 * clip_pitch && wh <= 4 * (wbudget / clip_pitch)) {
    const int cp = pitch < clip_pitch ? pitch : clip_pitch;
    const int rows = wbudget / cp;
    if (rows >= kRsTH + 2 && cp >= kRsTW + 2) {
      int cx0 = tx0 - ((cp - (kRsTW + 1)) >> 1);
      cx0 = cx0 > wx0 ? cx0 : wx0;
      if (vec4) cx0 &= ~3;                                  // wx0 is a multiple of four already: cx0 >= wx0 still
      int cy0 = ty0 - ((rows - (kRsTH + 1)) >> 1);
      cy0 = cy0 > wy0 ? cy0 : wy0;
      wx0 = cx0;
      wx1 = wx1 < cx0 + cp - 1 ? wx1 : cx0 + cp - 1;
      wy0 = cy0;
      wy1 = wy1 < cy0 + rows - 1 ? wy1 : cy0 + rows - 1;
      ww = wx1 - wx0 + 1;
      wh = wy1 - wy0 + 1;
      pitch = cp;
      window = ww > 0 && wh > 0;
    }
  }
  // pixels outside the window: their taps as 8-byte pairs straight from memory, in flight while the window loads
  typedef float float2_t __attribute__((ext_vector_type(2), aligned(4)));
  float2_t gt[kRsPPT][C], gu[kRsPPT][C];
  bool inw[kRsPPT];
#pragma unroll
  for (int k = 0; k < kRsPPT; ++k) {
    const int y = ty0 + k * 4 + wave;
    const bool live = x < W && y < H;
    inw[k] = window && xL[k] >= wx0 && xR[k] <= wx1 && yT[k] >= wy0 && yB[k] <= wy1;
#pragma unroll
    for (int c = 0; c < C; ++c) gt[k][c] = gu[k][c] = float2_t{0.f, 0.f};
    if (live && !inw[k]) {
      const int xb = xL[k] < W - 2 ? xL[k] : W - 2;         // pair [xb, xb + 1] holds columns xL and xR (W >= 2: the host's condition)
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float* p = in1 + ((size_t)b * C + c) * HW;
        gt[k][c] = *reinterpret_cast<const float2_t*>(p + (size_t)yT[k] * W + xb);
        gu[k][c] = *reinterpret_cast<const float2_t*>(p + (size_t)yB[k] * W + xb);
      }
    }
  }
  const int plane = (pitch * wh + 255) & ~255;             // whole 1-KiB pieces per plane: a piece's tail never reaches the next plane
  if (window) {
    // the window of every plane through LDS-DMA, straight into LDS (no VGPR round trip, every piece in flight at once: the tile's
    // latency is ONE memory round trip, not one per row).  A wave instruction moves 1 KiB = 256 consecutive LDS floats; lanes past
    // the window read out of range = 0
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in1), 0, in_bytes, 0x00020000);
    constexpr unsigned kOOB = 0x80000000u;
    if (vec4) {
      // lane -> (row, column) of its four floats in piece `wave`, then + 1024 floats per round (no division in the loop)
      const int f0 = wave * 256 + lane * 4;
      const int r0 = f0 / pitch, q0 = f0 - r0 * pitch;
      const int dr = 1024 / pitch, dq = 1024 - dr * pitch;
      const int npieces = plane >> 8;
#pragma unroll 1
      for (int c = 0; c < C; ++c) {
        const unsigned cbase = (unsigned)((((size_t)b * C + c) * HW + (size_t)wy0 * W + wx0) * 4);
        int r = r0, q = q0;
#pragma unroll 4
        for (int pc = wave; pc < npieces; pc += 4) {
          const unsigned voff = (r < wh && q < ww) ? cbase + (unsigned)((r * W + q) * 4) : kOOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(rs_smem + c * plane + pc * 256), 16, voff, 0, 0, 0);
          q += dq;
          r += dr;
          if (q >= pitch) { q -= pitch; ++r; }
        }
      }
    } else {
      const int nch = pitch >> 6;
      const int npieces = wh * nch;
#pragma unroll 1
      for (int c = 0; c < C; ++c) {
        const unsigned cbase = (unsigned)((((size_t)b * C + c) * HW + (size_t)wy0 * W + wx0) * 4);
#pragma unroll 4
        for (int pc = wave; pc < npieces; pc += 4) {
          const int r = pc / nch, j = pc - r * nch;
          const int q = 64 * j + lane;
          const unsigned voff = q < ww ? cbase + (unsigned)((r * W + q) * 4) : kOOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(rs_smem + c * plane + r * pitch + 64 * j), 4, voff, 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < kRsPPT; ++k) {
    const int y = ty0 + k * 4 + wave;
    if (x < W && y < H) {
      float tL[C], tR[C], uL[C], uR[C];
      if (inw[k]) {
        const int oT = (yT[k] - wy0) * pitch, oB = (yB[k] - wy0) * pitch, oL = xL[k] - wx0, oR = xR[k] - wx0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float* p = rs_smem + c * plane;
          tL[c] = p[oT + oL]; tR[c] = p[oT + oR]; uL[c] = p[oB + oL]; uR[c] = p[oB + oR];
        }
      } else {
        const int xb = xL[k] < W - 2 ? xL[k] : W - 2;
        const bool l1 = xL[k] != xb, r1 = xR[k] != xb;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          tL[c] = l1 ? gt[k][c][1] : gt[k][c][0]; tR[c] = r1 ? gt[k][c][1] : gt[k][c][0];
          uL[c] = l1 ? gu[k][c][1] : gu[k][c][0]; uR[c] = r1 ? gu[k][c][1] : gu[k][c][0];
        }
      }
#pragma unroll
      for (int c = 0; c < C; ++c) {
        float v = w00[k] * tL[c];
        v += w01[k] * tR[c];
        v += w10[k] * uL[c];
        v += w11[k] * uR[c];
        out[((size_t)b * C + c) * HW + (size_t)y * W + x] = v;
      }
    }
  }
#endif
}

// ---- ChannelNorm: sqrt(sum_c x^2) ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void channelnorm_kernel(const float* __restrict__ in, float* __restrict__ out, int C,
                                                          size_t HW, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / HW, pix = i - b * HW;
    float s = 0.f;
    for (int c = 0; c < C; ++c) {
      const float v = in[(b * C + c) * HW + pix];
      s += v * v;
    }
    out[i] = sqrtf(s);
  }
}

// Four pixels per thread, 16-byte loads and stores (round 5): HW % 4 == 0 and 16-byte aligned planes.  Same per-pixel
// arithmetic and channel order as channelnorm_kernel (bit-identical); the grid covers the tensor once (no grid-stride tail).
__global__ __launch_bounds__(256) void channelnorm_vec4_kernel(const float4_t* __restrict__ in, float4_t* __restrict__ out, int C,
                                                               size_t HW4, size_t total4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const size_t b = i / HW4, pix = i - b * HW4;
  float4_t s = {0.f, 0.f, 0.f, 0.f};
  if (C == 3) {          // (the FlowNet2 shape: all three loads in flight before the first use)
    const float4_t v0 = in[(b * 3 + 0) * HW4 + pix], v1 = in[(b * 3 + 1) * HW4 + pix], v2 = in[(b * 3 + 2) * HW4 + pix];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s[e] += v0[e] * v0[e];
      s[e] += v1[e] * v1[e];
      s[e] += v2[e] * v2[e];
    }
  } else {
    for (int c = 0; c < C; ++c) {
      const float4_t v = in[(b * C + c) * HW4 + pix];
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += v[e] * v[e];
    }
  }
  float4_t o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = sqrtf(s[e]);
  out[i] = o;     // (plain: non-temporal loads / stores measured 34.7 us against 11.7; two / four 16-byte pieces per thread: 11.4 / 12.4 against 11.2)
}

// ---- fused inter-network stage: warp img1 by flow, brightness error, 12-channel concat ------------------
template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void ld8<half_t>(const half_t* p, float (&v)[8]) {
  const half8_t h = *reinterpret_cast<const half8_t*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
}
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&v)[8]) {
  const float4_t a = reinterpret_cast<const float4_t*>(p)[0], b = reinterpret_cast<const float4_t*>(p)[1];
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void st8<half_t>(half_t* p, const float (&v)[8]) {
  half8_t h;
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = (half_t)v[i];
  *reinterpret_cast<half8_t*>(p) = h;
}
template <> __device__ __forceinline__ void st8<float>(float* p, const float (&v)[8]) {
  const float4_t a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
  reinterpret_cast<float4_t*>(p)[0] = a;
  reinterpret_cast<float4_t*>(p)[1] = b;
}

#ifndef FT_WARP_STORE_SC1
#define FT_WARP_STORE_SC1 0
#endif
#ifndef FT_WARP_STORE_LINES
#define FT_WARP_STORE_LINES 1     // fp16 output: whole-line stores after a lane exchange (-DFT_WARP_STORE_LINES=0: round 4's two strided pieces)
#endif
template <typename T>
__global__ __launch_bounds__(256) void flow_warp_concat_kernel(const T* __restrict__ x6, const float* __restrict__ flow,
                                                               float div_flow, T* __restrict__ y, int H, int W, int xl,
                                                               int xp, int yl, int yp, size_t total) {
  // one thread per PHYSICAL output pixel (b, yy, col): col in [yl, yl + W) carries data, the rest is zero
  const size_t HW = (size_t)H * W;
  for (size_t i = xcd_contiguous_block() * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % yp);
    const size_t row = i / yp;
    const size_t b = row / H;
    const int yy = (int)(row - b * H);
    const int xx = col - yl;
    float lo[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, hi[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if ((unsigned)xx < (unsigned)W) {
      const size_t pix = (size_t)yy * W + xx;
      const float dx = flow[(b * 2 + 0) * HW + pix], dy = flow[(b * 2 + 1) * HW + pix];
      const float xf = (float)xx + dx, yf = (float)yy + dy;
      const float fx = floorf(xf), fy = floorf(yf);
      const float alpha = xf - fx, beta = yf - fy;
      const int xL = (int)fminf(fmaxf(fx, 0.f), (float)(W - 1));
      const int xR = (int)fminf(fmaxf(fx + 1.f, 0.f), (float)(W - 1));
      const int yT = (int)fminf(fmaxf(fy, 0.f), (float)(H - 1));
      const int yB = (int)fminf(fmaxf(fy + 1.f, 0.f), (float)(H - 1));
      const float w00 = (1.f - alpha) * (1.f - beta), w01 = alpha * (1.f - beta);
      const float w10 = (1.f - alpha) * beta, w11 = alpha * beta;
      const T* img = x6 + b * (size_t)H * xp * 8;
      float c[8], tl[8], tr[8], bl[8], br[8];
      ld8<T>(img + ((size_t)yy * xp + xl + xx) * 8, c);
      ld8<T>(img + ((size_t)yT * xp + xl + xL) * 8, tl);
      ld8<T>(img + ((size_t)yT * xp + xl + xR) * 8, tr);
      ld8<T>(img + ((size_t)yB * xp + xl + xL) * 8, bl);
      ld8<T>(img + ((size_t)yB * xp + xl + xR) * 8, br);
      float warp[3], nrm = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float v = w00 * tl[3 + k];
        v += w01 * tr[3 + k];
        v += w10 * bl[3 + k];
        v += w11 * br[3 + k];
        warp[k] = v;
        const float d = c[k] - v;
        nrm += d * d;
      }
      lo[0] = c[0]; lo[1] = c[1]; lo[2] = c[2]; lo[3] = c[3]; lo[4] = c[4]; lo[5] = c[5]; lo[6] = warp[0]; lo[7] = warp[1];
      hi[0] = warp[2]; hi[1] = dx / div_flow; hi[2] = dy / div_flow; hi[3] = sqrtf(nrm);
    }
#if FT_WARP_STORE_LINES
    if constexpr (std::is_same<T, half_t>::value) {
      // A pixel's 32 output bytes leave as two 16-byte pieces; written straight from the pixel's lane a wave's store touches
      // every other 16 bytes of 2 KB (half of 16 lines, twice).  Round 5: the lanes trade pieces first (ds_bpermute), so that
      // store s writes ONE contiguous KiB = pixels 32 s .. 32 s + 31 whole (lane l: pixel 32 s + l / 2, piece l % 2).  Needs
      // the wave's 64 pixels consecutive in memory: true here (thread = physical pixel, whole wave inside `total`, else fallback).
      half8_t hl, hh;
#pragma unroll
      for (int e = 0; e < 8; ++e) { hl[e] = (half_t)lo[e]; hh[e] = (half_t)hi[e]; }
      const uint4_t vl = __builtin_bit_cast(uint4_t, hl), vh = __builtin_bit_cast(uint4_t, hh);
      const int lane = threadIdx.x & 63;
      const size_t wave_first = i - lane;
      if (wave_first + 64 <= total) {                      // wave-uniform
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
          const int src = (lane >> 1) + 32 * sidx;
          uint4_t a, b;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a[e] = (unsigned)__shfl((int)vl[e], src);
            b[e] = (unsigned)__shfl((int)vh[e], src);
          }
          const uint4_t v = (lane & 1) ? b : a;
#if FT_WARP_STORE_SC1
          store_out16(y + (wave_first + 32 * sidx) * 16 + lane * 8, v);
#else
          *reinterpret_cast<uint4_t*>(y + (wave_first + 32 * sidx) * 16 + lane * 8) = v;
#endif
        }
      } else {
        *reinterpret_cast<uint4_t*>(y + i * 16) = vl;
        *reinterpret_cast<uint4_t*>(y + i * 16 + 8) = vh;
      }
    } else
#endif
    {
      st8<T>(y + i * 16, lo);
      st8<T>(y + i * 16 + 8, hi);
    }
  }
}

// ---- FlowNet2 fusion input (models.py:140-168): 11 channels per pixel from img0/img1 and two flow fields -----
// (img0, sd_flow, s2_flow, |sd_flow|, |s2_flow|, |img0 - warp(img1, sd_flow)|, |img0 - warp(img1, s2_flow)|):
// two Resample2d, four ChannelNorm and the concat of the reference in one pass; NHWC [B,H,W,16] out.
template <typename T>
__device__ __forceinline__ float warp_err(const T* __restrict__ img, int H, int W, int xl, int xp, int yy, int xx,
                                          float dx, float dy, const float (&c)[8]) {
  const float xf = (float)xx + dx, yf = (float)yy + dy;
  const float fx = floorf(xf), fy = floorf(yf);
  const float alpha = xf - fx, beta = yf - fy;
  const int xL = (int)fminf(fmaxf(fx, 0.f), (float)(W - 1));
  const int xR = (int)fminf(fmaxf(fx + 1.f, 0.f), (float)(W - 1));
  const int yT = (int)fminf(fmaxf(fy, 0.f), (float)(H - 1));
  const int yB = (int)fminf(fmaxf(fy + 1.f, 0.f), (float)(H - 1));
  const float w00 = (1.f - alpha) * (1.f - beta), w01 = alpha * (1.f - beta), w10 = (1.f - alpha) * beta, w11 = alpha * beta;
  float tl[8], tr[8], bl[8], br[8];
  ld8<T>(img + ((size_t)yT * xp + xl + xL) * 8, tl);
  ld8<T>(img + ((size_t)yT * xp + xl + xR) * 8, tr);
  ld8<T>(img + ((size_t)yB * xp + xl + xL) * 8, bl);
  ld8<T>(img + ((size_t)yB * xp + xl + xR) * 8, br);
  float nrm = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float v = w00 * tl[3 + k];
    v += w01 * tr[3 + k];
    v += w10 * bl[3 + k];
    v += w11 * br[3 + k];
    const float d = c[k] - v;
    nrm += d * d;
  }
  return sqrtf(nrm);
}

template <typename T>
__global__ __launch_bounds__(256) void flow_fusion_concat_kernel(const T* __restrict__ x6, const float* __restrict__ fsd,
                                                                 const float* __restrict__ fs2, T* __restrict__ y, int H, int W,
                                                                 int xl, int xp, int yl, int yp, size_t total) {
  const size_t HW = (size_t)H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / HW, pix = i - b * HW;
    const int yy = (int)(pix / W), xx = (int)(pix - (size_t)yy * W);
    const T* img = x6 + b * (size_t)H * xp * 8;
    T* dst = y + (((size_t)b * H + yy) * yp + yl + xx) * 16;
    float c[8];
    ld8<T>(img + ((size_t)yy * xp + xl + xx) * 8, c);
    const float sdx = fsd[(b * 2 + 0) * HW + pix], sdy = fsd[(b * 2 + 1) * HW + pix];
    const float s2x = fs2[(b * 2 + 0) * HW + pix], s2y = fs2[(b * 2 + 1) * HW + pix];
    const float lo[8] = {c[0], c[1], c[2], sdx, sdy, s2x, s2y, sqrtf(sdx * sdx + sdy * sdy)};
    const float hi[8] = {sqrtf(s2x * s2x + s2y * s2y), warp_err<T>(img, H, W, xl, xp, yy, xx, sdx, sdy, c),
                         warp_err<T>(img, H, W, xl, xp, yy, xx, s2x, s2y, c), 0.f, 0.f, 0.f, 0.f, 0.f};
    st8<T>(dst, lo);
    st8<T>(dst + 8, hi);
  }
}

// nn.Upsample(scale_factor=4, mode='nearest') of (x * mul)  (models.py:59-60,448)
__global__ __launch_bounds__(256) void upsample_nearest4x_kernel(const float* __restrict__ x, float* __restrict__ y, int h, int w,
                                                                 size_t total, float mul) {
  const int H = 4 * h, W = 4 * w;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % W);
    const size_t t = i / W;
    const int oy = (int)(t % H);
    const size_t nc = t / H;
    y[i] = x[nc * (size_t)h * w + (size_t)(oy >> 2) * w + (ox >> 2)] * mul;
  }
}

static inline int grid_for(size_t total) {
  size_t g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

constexpr int kCorrGX = 3;  // horizontal displacements per thread

}  // namespace ft

using namespace ft;

extern "C" int ft_correlation_out_shape(int C, int H, int W, int pad_size, int kernel_size, int max_displacement,
                                        int stride1, int stride2, int* out_c, int* out_h, int* out_w) {
  if (C <= 0 || H <= 0 || W <= 0 || pad_size < 0 || kernel_size <= 0 || max_displacement < 0 || stride1 <= 0 ||
      stride2 <= 0)
    return FT_ERR_INVALID_ARG;
  // correlation_cuda.c:25-35
  const int krad = (kernel_size - 1) / 2;
  const int border = krad + max_displacement;
  const int ph = H + 2 * pad_size, pw = W + 2 * pad_size;
  const int D = (max_displacement / stride2) * 2 + 1;
  const int oh = (ph - 2 * border + stride1 - 1) / stride1;  // ceil
  const int ow = (pw - 2 * border + stride1 - 1) / stride1;
  if (oh <= 0 || ow <= 0) return FT_ERR_INVALID_ARG;
  if (out_c) *out_c = D * D;
  if (out_h) *out_h = oh;
  if (out_w) *out_w = ow;
  return FT_OK;
}

extern "C" int ft_correlation_fwd(const float* in1, const float* in2, float* out, int B, int C, int H, int W,
                                  int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                                  int corr_type_multiply, ft_stream_t stream) {
  if (!in1 || !in2 || !out || B <= 0) return FT_ERR_INVALID_ARG;
  if (corr_type_multiply != 1) return FT_ERR_UNSUPPORTED;
  int oc, oh, ow;
  int st = ft_correlation_out_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &oc, &oh, &ow);
  if (st != FT_OK) return st;
  const size_t total = (size_t)B * oc * oh * ow;
  hipLaunchKernelGGL(correlation_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), in1, in2, out, C, H,
                     W, oc, oh, ow, pad_size, kernel_size, max_displacement, stride1, stride2, total);
  FT_LAUNCH_CHECK("correlation_nchw_kernel");
  return FT_OK;
}

extern "C" int ft_correlation_nhwc_fwd(const void* f1, const void* f2, void* y, int B, int C, int H, int W,
                                       int max_displacement, int stride2, int f_cstride, int y_cstride, int y_coff,
                                       int act, float slope, int dtype, ft_stream_t stream) {
  if (!f1 || !f2 || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0 || max_displacement < 0 || stride2 <= 0)
    return FT_ERR_INVALID_ARG;
  if (dtype != FT_F16 && dtype != FT_F32) return FT_ERR_INVALID_ARG;
  if (C % 8 || f_cstride % 8 || f_cstride < C) return FT_ERR_INVALID_ARG;
  const int D = (max_displacement / stride2) * 2 + 1;
  if (y_coff < 0 || y_cstride < y_coff + D * D) return FT_ERR_INVALID_ARG;
  const size_t esz = dtype_size(dtype);
  const unsigned long long f_bytes = (unsigned long long)B * H * W * f_cstride * esz;
  if (dtype == FT_F16 && stride2 == 2 && max_displacement % 2 == 0 && max_displacement <= 32 && C == 256 &&
      f_bytes < (1ull << 31)) {
    const int drad = max_displacement / 2;
    const unsigned long long y_bytes = (unsigned long long)B * H * W * y_cstride * 2;
    static const bool no_rows = getenv("FT_CORR_ROWS") && atoi(getenv("FT_CORR_ROWS")) == 0;   // dev A/B: one row per workgroup
    static const bool no_t = getenv("FT_CORR_NARROW") && atoi(getenv("FT_CORR_NARROW")) == 0;   // dev A/B: the 104-column ring
    if (drad == 10 && y_bytes < (1ull << 31) && !no_rows && !no_t && W <= 64 && y_coff % 8 == 0 && y_cstride % 8 == 0) {
      // FlowNetC's shape on maps up to 64 wide: 64-column ring slots (FT_CORR_STAGED=1: band transposed through LDS)
      static const bool stg = getenv("FT_CORR_STAGED") && atoi(getenv("FT_CORR_STAGED")) == 1;   // dev A/B: band through LDS
      static const bool w8 = !(getenv("FT_CORR_WAVES") && atoi(getenv("FT_CORR_WAVES")) == 4);     // dev A/B: four waves, R = 3
      static const bool tr = getenv("FT_CORR_TR") && atoi(getenv("FT_CORR_TR")) == 1;               // dev A/B: 1 = the transposed band (slower)
      static const bool direct = !(getenv("FT_CORR_DIRECT") && atoi(getenv("FT_CORR_DIRECT")) == 0);   // dev A/B: 0 = the window-column tiles
      static const int dwaves = getenv("FT_CORR_DIRECT_WAVES") ? atoi(getenv("FT_CORR_DIRECT_WAVES")) : 6;   // dev A/B: 8 = R 4 (192 workgroups at 16 x 48 rows)
      static const int cdbg = getenv("FT_CORR_DBG") ? atoi(getenv("FT_CORR_DBG")) : 0;               // dev ablation (DIRECT form): 1 = no band stores, 2 = no MFMAs
      const bool dir = direct && !stg && w8 && !tr;
      const bool d6 = dir && dwaves == 6;
      const int R = (stg || !w8 || d6) ? 3 : 4;
      auto k = stg ? correlation_mfma_rows64_kernel<16, 3, 10, true, 1>
                   : (w8 ? (tr ? correlation_mfma_rows64_kernel<16, 4, 10, false, 2, true>
                               : (d6 ? correlation_mfma_rows64_kernel<16, 3, 10, false, 2, false, true, 6>
                                     : (direct ? correlation_mfma_rows64_kernel<16, 4, 10, false, 2, false, true>
                                               : correlation_mfma_rows64_kernel<16, 4, 10, false, 2>)))
                         : correlation_mfma_rows64_kernel<16, 3, 10, false, 1>);
      const size_t ldst = dir ? 4 * 64 * 512 + 512 + 128 + 1024 : 3 * 64 * 512 + 512 + 3 * 64 * (8 * 21 * 2) + 128;
      static bool raised[64] = {};
      int dev = 0;
      FT_HIP_CHECK(hipGetDevice(&dev));
      if (dev < 0 || dev >= 64 || !raised[dev]) {
        FT_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldst));
        if (dev >= 0 && dev < 64) raised[dev] = true;
      }
      const int ngy = 2 * ceil_div((H + 1) / 2, R);
      hipLaunchKernelGGL(k, dim3(ngy * B), dim3(d6 ? 384 : ((!stg && w8) ? 512 : 256)), ldst, as_stream(stream), static_cast<const half_t*>(f1),
                         static_cast<const half_t*>(f2), static_cast<half_t*>(y), H, W, (unsigned)f_bytes, (unsigned)y_bytes,
                         f_cstride, y_cstride, y_coff, act | (dir ? cdbg << 8 : 0), slope, ngy);
      FT_LAUNCH_CHECK("correlation_mfma_rows64_kernel");
      return FT_OK;
    }
    if (drad == 10 && y_bytes < (1ull << 31) && !no_rows) {     // FlowNetC's shape: 3 output rows per workgroup
      constexpr int R = 3;
      auto k = correlation_mfma_rows_kernel<16, R, 10>;
      constexpr size_t lds3 = 3 * (((size_t)(64 + 40) * 512 + 1023) / 1024 * 1024);
      static bool raised[64] = {};
      int dev = 0;
      FT_HIP_CHECK(hipGetDevice(&dev));
      if (dev < 0 || dev >= 64 || !raised[dev]) {
        FT_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
        if (dev >= 0 && dev < 64) raised[dev] = true;
      }
      const int groups = ceil_div((H + 1) / 2, R);
      const int ngx = ceil_div(W, 64), ngy = 2 * groups;
      hipLaunchKernelGGL(k, dim3(ngx * ngy * B), dim3(256), lds3, as_stream(stream), static_cast<const half_t*>(f1),
                         static_cast<const half_t*>(f2), static_cast<half_t*>(y), H, W, (unsigned)f_bytes, (unsigned)y_bytes,
                         f_cstride, y_cstride, y_coff, act, slope, ngx, ngy);
      FT_LAUNCH_CHECK("correlation_mfma_rows_kernel");
      return FT_OK;
    }
    const size_t stage = ((size_t)(64 + 4 * drad) * C * 2 + 1023) / 1024 * 1024;
    const size_t lds2 = 2 * stage;
    auto k = correlation_mfma_kernel<16>;
    FT_RAISE_LDS(k, 160 * 1024);
    hipLaunchKernelGGL(k, dim3(ceil_div(W, 64), H, B), dim3(256), lds2, as_stream(stream), static_cast<const half_t*>(f1),
                       static_cast<const half_t*>(f2), static_cast<half_t*>(y), H, W, drad, (unsigned)f_bytes, f_cstride,
                       y_cstride, y_coff, act, slope);
    FT_LAUNCH_CHECK("correlation_mfma_kernel");
    return FT_OK;
  }
  const size_t pstride = C * esz + 16;
  // fp16: 32-pixel tiles / 256 threads; fp32 (parity mode): 16-pixel tiles / 128 threads (LDS budget)
  const int tx = dtype == FT_F16 ? 32 : 16;
  const size_t lds = (size_t)(tx + tx + 2 * max_displacement) * pstride + (size_t)tx * D * D * 4;
  if (lds > 160 * 1024) return FT_ERR_UNSUPPORTED;
  dim3 grid(ceil_div(W, tx) * H, B);
  if (dtype == FT_F16) {
    auto k = correlation_nhwc_kernel<half_t, 32, kCorrGX, 256>;
    if (lds > 64 * 1024) FT_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, grid, dim3(256), lds, as_stream(stream), static_cast<const half_t*>(f1),
                       static_cast<const half_t*>(f2), static_cast<half_t*>(y), C, H, W, max_displacement, stride2,
                       f_cstride, y_cstride, y_coff, act, slope);
  } else {
    auto k = correlation_nhwc_kernel<float, 16, kCorrGX, 128>;
    if (lds > 64 * 1024) FT_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, grid, dim3(128), lds, as_stream(stream), static_cast<const float*>(f1),
                       static_cast<const float*>(f2), static_cast<float*>(y), C, H, W, max_displacement, stride2,
                       f_cstride, y_cstride, y_coff, act, slope);
  }
  FT_LAUNCH_CHECK("correlation_nhwc_kernel");
  return FT_OK;
}

extern "C" int ft_resample2d_fwd(const float* in1, const float* flow, float* out, int B, int C, int H, int W,
                                 ft_stream_t stream) {
  if (!in1 || !flow || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0) return FT_ERR_INVALID_ARG;
  const size_t total = (size_t)B * H * W;
  static const bool no_window = getenv("FT_RESAMPLE_WINDOW") && atoi(getenv("FT_RESAMPLE_WINDOW")) == 0;   // dev A/B: the gather kernels
  const unsigned long long in_bytes = (unsigned long long)B * C * H * W * 4ull;
  if (!no_window && C >= 1 && C <= 4 && W >= 2 && in_bytes < (1ull << 31)) {
    const int tiles_x = ceil_div(W, kRsTW), tiles_y = ceil_div(H, kRsTH);
    const long long nblk = (long long)B * tiles_x * tiles_y;
    if (nblk <= 0x7fffffffLL) {
      const dim3 grid((unsigned)nblk);
      // window budget per plane (floats): the LDS a workgroup reserves = C x budget x 4 bytes decides how many workgroups share a CU
      // (26 KiB per plane: two; 9 KiB: five); tiles whose window exceeds it take the pair gathers.  FT_RESAMPLE_WBUDGET (dev A/B)
      // Round 5, same box, configs[3] shapes (noise = per-pixel N(0, 4 px) flows, smooth = a 12 x 16 grid of N(0, 6 px) vectors upsampled):
      //   budget 6656 (26 KiB per plane, 2 workgroups per CU): noise 45.5 us, smooth 39.7 us
      //   budget 3328 (13 KiB, 4 per CU):                      noise 56.4 us (every tile falls back), smooth 32.3 us
      //   budget 2304 / 1536:                                   noise 55.8 / 55.1, smooth 35.6 / 37.0
      // Flow fields that reach this operator are network outputs (smooth): 3328 was the default of round 5.
      // Round 6 (clipped windows: a tile whose box does not fit keeps a window around itself and gathers only the pixels outside it),
      // same box, noise / smooth: budget 1536: 45.2 / 32.0 us, 1792: 37.5 / 29.9, 2048: 37.0 / 30.0, 2304: 36.3 / 29.4,
      // 2560: 35.3 / 29.4 (five workgroups per CU), 2816: 37.4 / 31.3, 3072: 36.8 / 31.9, 4096: 38.8 / 33.0, 6656: 47.2 / 40.2;
      // clip pitch 80 / 96 / 112 / 128 floats at 2560: 34.0 / 34.8-35.0 / 34.8 / 34.6 us noise, 28.0-28.7 smooth.  2560 is the default.
      static const int wb_env = getenv("FT_RESAMPLE_WBUDGET") ? atoi(getenv("FT_RESAMPLE_WBUDGET")) : 0;
      const int wbudget = wb_env >= 256 && wb_env <= kRsMaxWindow ? (wb_env & ~255) : 2560;
      const size_t lds = (size_t)C * wbudget * sizeof(float);
#define FT_RS_LAUNCH(CC)                                                                                                   \
  {                                                                                                                        \
    auto kw = resample2d_window_kernel<CC>;                                                                                \
    if (lds > 64 * 1024) FT_RAISE_LDS(kw, 112 * 1024);                                                                     \
    hipLaunchKernelGGL(kw, grid, dim3(256), lds, as_stream(stream), in1, flow, out, H, W, tiles_x, tiles_y, (unsigned)in_bytes, wbudget); \
  }
      switch (C) {
        case 1: FT_RS_LAUNCH(1) break;
        case 2: FT_RS_LAUNCH(2) break;
        case 3: FT_RS_LAUNCH(3) break;
        default: FT_RS_LAUNCH(4) break;
      }
#undef FT_RS_LAUNCH
      FT_LAUNCH_CHECK("resample2d_window_kernel");
      return FT_OK;
    }
  }
  static const bool no_pair = getenv("FT_RESAMPLE_PAIR") && atoi(getenv("FT_RESAMPLE_PAIR")) == 0;   // dev A/B
  if (!no_pair && W >= 2 && C >= 1 && C <= 4 && (long long)H * W < (1LL << 31)) {
    const dim3 grid(grid_for(total));
    switch (C) {
      case 1: hipLaunchKernelGGL(resample2d_pair_kernel<1>, grid, dim3(256), 0, as_stream(stream), in1, flow, out, H, W, total); break;
      case 2: hipLaunchKernelGGL(resample2d_pair_kernel<2>, grid, dim3(256), 0, as_stream(stream), in1, flow, out, H, W, total); break;
      case 3: hipLaunchKernelGGL(resample2d_pair_kernel<3>, grid, dim3(256), 0, as_stream(stream), in1, flow, out, H, W, total); break;
      default: hipLaunchKernelGGL(resample2d_pair_kernel<4>, grid, dim3(256), 0, as_stream(stream), in1, flow, out, H, W, total); break;
    }
    FT_LAUNCH_CHECK("resample2d_pair_kernel");
    return FT_OK;
  }
  hipLaunchKernelGGL(resample2d_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), in1, flow, out, C, H, W,
                     total);
  FT_LAUNCH_CHECK("resample2d_kernel");
  return FT_OK;
}

extern "C" int ft_channelnorm_fwd(const float* in1, float* out, int B, int C, int H, int W, ft_stream_t stream) {
  if (!in1 || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0) return FT_ERR_INVALID_ARG;
  const size_t HW = (size_t)H * W, total = (size_t)B * HW;
  if (HW % 4 == 0 && (reinterpret_cast<uintptr_t>(in1) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && total / 4 / 256 < (1u << 30)) {
    const size_t total4 = total / 4;
    hipLaunchKernelGGL(channelnorm_vec4_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4_t*>(in1), reinterpret_cast<float4_t*>(out), C, HW / 4, total4);
    FT_LAUNCH_CHECK("channelnorm_vec4_kernel");
    return FT_OK;
  }
  hipLaunchKernelGGL(channelnorm_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), in1, out, C, HW, total);
  FT_LAUNCH_CHECK("channelnorm_kernel");
  return FT_OK;
}

extern "C" int ft_flow_warp_concat(const void* x6, const float* flow, float div_flow, void* y, int B, int H, int W,
                                   int x_lpad, int x_wpitch, int y_lpad, int y_wpitch, int dtype, ft_stream_t stream) {
  if (!x6 || !flow || !y || B <= 0 || H <= 0 || W <= 0 || div_flow == 0.f) return FT_ERR_INVALID_ARG;
  if (x_lpad < 0 || y_lpad < 0 || x_wpitch < x_lpad + W || y_wpitch < y_lpad + W) return FT_ERR_INVALID_ARG;
  if (dtype != FT_F16 && dtype != FT_F32) return FT_ERR_INVALID_ARG;
  const size_t total = (size_t)B * H * y_wpitch;
  if (dtype == FT_F16)
    hipLaunchKernelGGL(flow_warp_concat_kernel<half_t>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream),
                       static_cast<const half_t*>(x6), flow, div_flow, static_cast<half_t*>(y), H, W, x_lpad, x_wpitch,
                       y_lpad, y_wpitch, total);
  else
    hipLaunchKernelGGL(flow_warp_concat_kernel<float>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream),
                       static_cast<const float*>(x6), flow, div_flow, static_cast<float*>(y), H, W, x_lpad, x_wpitch,
                       y_lpad, y_wpitch, total);
  FT_LAUNCH_CHECK("flow_warp_concat_kernel");
  return FT_OK;
}

extern "C" int ft_flow_fusion_concat(const void* x6, const float* flow_sd, const float* flow_s2, void* y, int B, int H, int W,
                                     int x_lpad, int x_wpitch, int y_lpad, int y_wpitch, int dtype, ft_stream_t stream) {
  if (!x6 || !flow_sd || !flow_s2 || !y || B <= 0 || H <= 0 || W <= 0) return FT_ERR_INVALID_ARG;
  if (x_lpad < 0 || x_wpitch < x_lpad + W || y_lpad < 0 || y_wpitch < y_lpad + W) return FT_ERR_INVALID_ARG;
  if (dtype != FT_F16 && dtype != FT_F32) return FT_ERR_INVALID_ARG;
  const size_t total = (size_t)B * H * W;
  if (dtype == FT_F16)
    hipLaunchKernelGGL(flow_fusion_concat_kernel<half_t>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream),
                       static_cast<const half_t*>(x6), flow_sd, flow_s2, static_cast<half_t*>(y), H, W, x_lpad, x_wpitch,
                       y_lpad, y_wpitch, total);
  else
    hipLaunchKernelGGL(flow_fusion_concat_kernel<float>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream),
                       static_cast<const float*>(x6), flow_sd, flow_s2, static_cast<float*>(y), H, W, x_lpad, x_wpitch,
                       y_lpad, y_wpitch, total);
  FT_LAUNCH_CHECK("flow_fusion_concat_kernel");
  return FT_OK;
}

extern "C" int ft_upsample_nearest4x(const float* x, float* y, int N, int C, int h, int w, float mul, ft_stream_t stream) {
  if (!x || !y || N <= 0 || C <= 0 || h <= 0 || w <= 0) return FT_ERR_INVALID_ARG;
  const size_t total = (size_t)N * C * 16 * h * w;
  hipLaunchKernelGGL(upsample_nearest4x_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x, y, h, w, total, mul);
  FT_LAUNCH_CHECK("upsample_nearest4x_kernel");
  return FT_OK;
}
