// HBM-bound helper kernels around the implicit-GEMM convs: layout packing, 3x3/s2 max-pool,
// heatmap arg-max (max_preds / final_preds nudge), FlowNet input normalisation, x4 bilinear
// upsampling.  All are one-pass, vectorised to 16 bytes per lane on the NHWC side.
#include "ft_common.h"

namespace ft {

template <typename T> struct Vec8;  // 8 channels of T
template <> struct Vec8<half_t> { typedef half8_t type; };
template <> struct Vec8<float> { struct type { float4_t lo, hi; }; };

template <typename T> __device__ __forceinline__ void store8(T* dst, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<half_t>(half_t* dst, const float (&v)[8]) {
  half8_t h;
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = (half_t)v[i];
  *reinterpret_cast<half8_t*>(dst) = h;
}
template <> __device__ __forceinline__ void store8<float>(float* dst, const float (&v)[8]) {
  float4_t a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
  reinterpret_cast<float4_t*>(dst)[0] = a;
  reinterpret_cast<float4_t*>(dst)[1] = b;
}
template <typename T> __device__ __forceinline__ void store4c(T* dst, const float (&v)[4]);
template <> __device__ __forceinline__ void store4c<half_t>(half_t* dst, const float (&v)[4]) {
  half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
  *reinterpret_cast<half4_t*>(dst) = h;
}
template <> __device__ __forceinline__ void store4c<float>(float* dst, const float (&v)[4]) {
  float4_t f = {v[0], v[1], v[2], v[3]};
  *reinterpret_cast<float4_t*>(dst) = f;
}
template <typename T> __device__ __forceinline__ void load8(const T* src, float (&v)[8]);
template <> __device__ __forceinline__ void load8<half_t>(const half_t* src, float (&v)[8]) {
  half8_t h = *reinterpret_cast<const half8_t*>(src);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
}
template <> __device__ __forceinline__ void load8<float>(const float* src, float (&v)[8]) {
  float4_t a = reinterpret_cast<const float4_t*>(src)[0], b = reinterpret_cast<const float4_t*>(src)[1];
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
  v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}

// ---- NCHW fp32 -> NHWC T (channel-padded, optionally column-padded "row-packed" layout) ----------------
// one thread per PHYSICAL pixel (n, y, xp): xp in [lpad, lpad + W) carries data, every other column is zero
template <typename T>
__global__ __launch_bounds__(256) void pack_nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y,
                                                                int C, int H, int W, int cpad, int lpad, int wpitch,
                                                                size_t total) {
  const size_t HW = (size_t)H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xp = (int)(i % wpitch);
    const size_t row = i / wpitch;          // n * H + yy
    const size_t n = row / H, yy = row - n * H;
    const int xx = xp - lpad;
    const bool live = (unsigned)xx < (unsigned)W;
    const float* xp_ = x + n * C * HW + yy * W + (live ? xx : 0);
    T* yp = y + i * cpad;
    for (int c0 = 0; c0 < cpad; c0 += 4) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (live && c0 + e < C) ? xp_[(size_t)(c0 + e) * HW] : 0.f;
      store4c<T>(yp + c0, v);
    }
  }
}

// fast path of the above for the 3-channel network inputs (fp16, 4 halves per pixel, W % 4 == 0): one thread = 4
// consecutive pixels = three 16-byte planar reads and 32 contiguous output bytes, 4x fewer index divisions; the first /
// last thread of a row also zero the pad columns
__global__ __launch_bounds__(256) void pack_nchw_rows4_kernel(const float* __restrict__ x, half_t* __restrict__ y, int C, int H,
                                                              int W, int lpad, int wpitch, size_t total) {
  const size_t HW = (size_t)H * W;
  const int W4 = W >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % W4);
    const size_t row = i / W4;              // n * H + yy
    const size_t n = row / H, yy = row - n * H;
    const float* xp_ = x + n * C * HW + yy * W + 4 * g;
    float4_t v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = c < C ? *reinterpret_cast<const float4_t*>(xp_ + (size_t)c * HW) : float4_t{0.f, 0.f, 0.f, 0.f};
    half_t* yrow = y + row * wpitch * 4;
    half4_t* yp = reinterpret_cast<half4_t*>(yrow + (size_t)(lpad + 4 * g) * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) yp[k] = half4_t{(half_t)v[0][k], (half_t)v[1][k], (half_t)v[2][k], (half_t)0.f};
    const half4_t z = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
    if (g == 0)
      for (int c = 0; c < lpad; ++c) reinterpret_cast<half4_t*>(yrow)[c] = z;
    if (g == W4 - 1)
      for (int c = lpad + W; c < wpitch; ++c) reinterpret_cast<half4_t*>(yrow)[c] = z;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void unpack_nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ y,
                                                                  int C, size_t HW, size_t total, int cstride, int coff) {
  // one thread per output element, lanes along the pixel axis (coalesced NCHW writes)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i % HW;
    const size_t nc = i / HW;
    const size_t n = nc / C, c = nc - n * C;
    y[i] = (float)x[(n * HW + pix) * cstride + coff + c];
  }
}

// ---- MaxPool2d(3, 2, 1), NHWC ----------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const T* __restrict__ x, T* __restrict__ y, int Hi, int Wi,
                                                           int Ho, int Wo, int C8, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    size_t t = i / C8;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const size_t n = t / Ho;
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if ((unsigned)iy >= (unsigned)Hi) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if ((unsigned)ix >= (unsigned)Wi) continue;
        float v[8];
        load8<T>(x + (((n * Hi + iy) * Wi + ix) * (size_t)C8 + c8) * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], v[e]);
      }
    }
    store8<T>(y + i * 8, m);
  }
}

// ---- max_preds (+ final_preds nudge): one workgroup per (n,k) map ------------------------------
__global__ __launch_bounds__(256) void heatmap_max_preds_kernel(const float* __restrict__ hm, int H, int W, int adjust,
                                                                int32_t* __restrict__ idx_out, float* __restrict__ score_out,
                                                                float* __restrict__ coords_out, int sstride, int cstride) {
  const int map = blockIdx.x;
  const int HW = H * W;
  const float* p = hm + (size_t)map * HW;
  float best = -INFINITY;
  int bidx = 0x7fffffff;
  for (int i = threadIdx.x; i < HW; i += 256) {
    const float v = p[i];
    if (v > best) { best = v; bidx = i; }  // strided ascending i: keeps the first occurrence per thread
  }
  // wave64 butterfly: larger value wins, ties go to the smaller index (first occurrence, row-major)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off);
    const int oi = __shfl_xor(bidx, off);
    if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
  }
  __shared__ float s_v[4];
  __shared__ int s_i[4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_v[wave] = best; s_i[wave] = bidx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (s_v[w] > best || (s_v[w] == best && s_i[w] < bidx)) { best = s_v[w]; bidx = s_i[w]; }
    if (bidx == 0x7fffffff) bidx = 0;  // all-NaN / empty map
    idx_out[map] = bidx;
    score_out[(size_t)map * sstride] = best;
    float cx = 0.f, cy = 0.f;
    if (best > 0.f) {  // coords.mul(mask), evaluation.py:17-19
      const int x = bidx % W, y = bidx / W;
      cx = (float)x;
      cy = (float)y;
      if (adjust && x > 0 && x < W - 1 && y > 0 && y < H - 1) {  // evaluation.py:31-33
        const float dx = p[y * W + x + 1] - p[y * W + x - 1];
        const float dy = p[(y + 1) * W + x] - p[(y - 1) * W + x];
        cx += dx > 0.f ? 0.25f : (dx < 0.f ? -0.25f : 0.f);
        cy += dy > 0.f ? 0.25f : (dy < 0.f ? -0.25f : 0.f);
      }
    }
    coords_out[(size_t)map * cstride] = cx;
    coords_out[(size_t)map * cstride + 1] = cy;
  }
}

// ---- arg-max margin screen: per crop, the smallest (top-1 - top-2) over its K maps ------------------------------------
// One workgroup per crop walks its K maps; a thread keeps the two largest values it has seen (at different pixels), waves
// and workgroup merge pairs: best = max(b1, b2), second = max(min(b1, b2), s1, s2).  Equal maxima at two pixels -> margin 0.
__global__ __launch_bounds__(256) void heatmap_min_margin_kernel(const float* __restrict__ hm, int K, int HW, float* __restrict__ out) {
  const int n = blockIdx.x;
  __shared__ float s_b[4], s_s[4];
  float crop_min = INFINITY;
  for (int k = 0; k < K; ++k) {
    const float* p = hm + ((size_t)n * K + k) * HW;
    float b = -INFINITY, s2 = -INFINITY;
    for (int i = threadIdx.x; i < HW; i += 256) {
      const float v = p[i];
      s2 = fmaxf(s2, fminf(b, v));          // (branch-free: see heatmap_argmax_screen_kernel)
      b = fmaxf(b, v);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ob = __shfl_xor(b, off), os = __shfl_xor(s2, off);
      const float nb = fmaxf(b, ob);
      s2 = fmaxf(fminf(b, ob), fmaxf(s2, os));
      b = nb;
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();                  // (the previous map's s_b / s_s have been read)
    if ((threadIdx.x & 63) == 0) { s_b[wave] = b; s_s[wave] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; ++w) {
        const float nb = fmaxf(b, s_b[w]);
        s2 = fmaxf(fminf(b, s_b[w]), fmaxf(s2, s_s[w]));
        b = nb;
      }
      crop_min = fminf(crop_min, b - s2);
    }
  }
  if (threadIdx.x == 0) out[n] = crop_min;
}

// ---- arg-max screen with a RELATIVE error bound: which crops may have another arg-max (or another `score > 0` mask) in fp32 ----
// Per crop: (top-1, top-2, min) of each of its K maps, then with R = (largest value - smallest value over the crop's maps) and
// E = rel_bound * R (the fp16 mode's heat-map error scales with the maps' range, not with an absolute constant):
//   flag = any map with !(top1 - top2 >= 2 E)   (two values closer than 2 E can swap under an error of E per element)
//          or !(|top1| >= E)                    (max_preds zeroes the coordinates when score <= 0, evaluation.py:17-19: the mask can flip)
//          or any non-finite statistic          (NaN compares false: written as negated >=)
// stats[n] = (smallest margin, R, smallest |top1|, E).  K <= 256.
// Round 5: a WAVE per map (the four waves of a crop's block walk its K maps in parallel, 16-byte loads when HW % 4 == 0, no
// block barrier inside the walk; before: the whole block per map with two __syncthreads each: ~50 us for 64 crops x 17 maps on
// 64 CUs, which is what the exact mode's "nothing flagged" path cost on top of the plain step).
__global__ __launch_bounds__(256) void heatmap_argmax_screen_kernel(const float* __restrict__ hm, int K, int HW, float rel_bound,
                                                                    int32_t* __restrict__ flags, float* __restrict__ stats) {
  const int n = blockIdx.x;
  __shared__ float s_t1[256], s_t2[256], s_mn[256];
  __shared__ int s_bad[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bool nf = false;
  for (int k = wave; k < K; k += 4) {
    const float* p = hm + ((size_t)n * K + k) * HW;
    float b = -INFINITY, s2 = -INFINITY, mn = INFINITY;
    auto take = [&](float v) {
      // branch-free top-2 update: the if / else-if form (`if (v > b) { s2 = b; b = v; } else if (v > s2) s2 = v;`) was compiled
      // into an indexed store to a two-element STACK array — a scratch load + store + vmcnt(0) per element, 10 us per map
      nf |= !(fabsf(v) <= 3.0e38f);
      s2 = fmaxf(s2, fminf(b, v));
      b = fmaxf(b, v);
      mn = fminf(mn, v);
    };
    if ((HW & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
      // eight 16-byte loads in flight per lane (one after the other the walk is load latency x loads: 73 us measured)
      const float4_t* p4 = reinterpret_cast<const float4_t*>(p);
      const int n4 = HW >> 2;
      for (int i0 = 0; i0 < n4; i0 += 8 * 64) {
        float4_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + u * 64 + lane;
          v[u] = p4[i < n4 ? i : lane % n4];           // (tail: re-read a valid element, folded away below)
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (i0 + u * 64 + lane < n4) { take(v[u][0]); take(v[u][1]); take(v[u][2]); take(v[u][3]); }
        }
      }
    } else {
      for (int i = lane; i < HW; i += 64) take(p[i]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ob = __shfl_xor(b, off), os = __shfl_xor(s2, off);
      const float nb = fmaxf(b, ob);
      s2 = fmaxf(fminf(b, ob), fmaxf(s2, os));
      b = nb;
      mn = fminf(mn, __shfl_xor(mn, off));
    }
    if (lane == 0) { s_t1[k] = b; s_t2[k] = s2; s_mn[k] = mn; }
  }
  const bool wbad = __any(nf) != 0;
  if (lane == 0) s_bad[wave] = wbad ? 1 : 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    bool flag = (s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3]) != 0;
    float crop_max = -INFINITY, crop_min = INFINITY;
    for (int k = 0; k < K; ++k) {
      crop_max = fmaxf(crop_max, s_t1[k]);
      crop_min = fminf(crop_min, s_mn[k]);
    }
    const float R = crop_max - crop_min, E = rel_bound * R;
    float mmin = INFINITY, amin = INFINITY;
    for (int k = 0; k < K; ++k) {
      const float mg = s_t1[k] - s_t2[k], at = fabsf(s_t1[k]);
      flag |= !(mg >= 2.f * E) || !(at >= E);
      mmin = fminf(mmin, mg);
      amin = fminf(amin, at);
    }
    flag |= !(R <= 3.0e38f) || !(E >= 0.f);
    flags[n] = flag ? 1 : 0;
    stats[4 * n] = mmin;
    stats[4 * n + 1] = R;
    stats[4 * n + 2] = amin;
    stats[4 * n + 3] = E;
  }
}

// Compaction + gather of the flagged crops ON THE DEVICE (round 5): block (k, part) finds the k-th set flag (every block
// recomputes the small prefix: N <= 1024 flags) and copies that crop's row to slot k of `dst`; block (0, 0) also writes the
// header = {count, idx[0], idx[1], ...}.  With nothing flagged every block exits after reading the flags.
__global__ __launch_bounds__(256) void gather_flagged_rows_kernel(const int32_t* __restrict__ flags, int N, const uint4_t* __restrict__ src,
                                                                  size_t row16, uint4_t* __restrict__ dst, int32_t* __restrict__ header) {
  __shared__ int s_cnt[4];
  __shared__ int s_idx;
  const int k = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // thread t owns flags [4 t, 4 t + 4): running count over threads by wave ballots of per-thread counts
  int f[4], mine = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = 4 * threadIdx.x + e;
    f[e] = (i < N && flags[i] != 0) ? 1 : 0;
    mine += f[e];
  }
  int incl = mine;                         // inclusive scan inside the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  if (lane == 63) s_cnt[wave] = incl;
  if (threadIdx.x == 0) s_idx = -1;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += s_cnt[w];
  const int total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
  int before = base + incl - mine;         // flagged crops in front of this thread's four
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (f[e]) {
      if (before == k) s_idx = 4 * threadIdx.x + e;
      if (blockIdx.x == 0 && blockIdx.y == 0) header[1 + before] = 4 * threadIdx.x + e;     // (block (0, 0) writes the whole list)
      ++before;
    }
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) header[0] = total;
  __syncthreads();
  const int idx = s_idx;
  if (idx < 0) return;                     // fewer than k + 1 flagged crops
  const uint4_t* s = src + (size_t)idx * row16;
  uint4_t* d = dst + (size_t)k * row16;
  for (size_t i = (size_t)blockIdx.y * 256 + threadIdx.x; i < row16; i += (size_t)gridDim.y * 256) d[i] = s[i];
}

// ---- FlowNet2* rgb mean: stage 1 partial sums, stage 2 finish -----------------------------------
__global__ __launch_bounds__(256) void rgb_partial_sum_kernel(const float* __restrict__ x, size_t L,
                                                              float* __restrict__ partial) {
  const int bc = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
  const size_t per = (L + nsplit - 1) / nsplit;
  const size_t lo = (size_t)split * per;
  const size_t hi = lo + per < L ? lo + per : L;
  const float* p = x + (size_t)bc * L;
  float s = 0.f;
  if ((L & 3) == 0 && (per & 3) == 0) {      // 16-byte loads, four independent partial sums per lane
    // four loads in flight per lane (the one-load loop ran at 3.3 TB/s: latency-bound), combined in a fixed order
    float4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    const float4_t* p4 = reinterpret_cast<const float4_t*>(p);
    const size_t e = hi / 4;
    size_t i = lo / 4 + threadIdx.x;
    for (; i + 768 < e; i += 1024) {
      a0 += p4[i];
      a1 += p4[i + 256];
      a2 += p4[i + 512];
      a3 += p4[i + 768];
    }
    for (; i < e; i += 256) a0 += p4[i];
    const float4_t a = (a0 + a1) + (a2 + a3);
    s = (a[0] + a[1]) + (a[2] + a[3]);
  } else {
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) s += p[i];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  __shared__ float sw[4];
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[bc * nsplit + split] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}

__global__ __launch_bounds__(64) void rgb_mean_finish_kernel(const float* __restrict__ partial, int nsplit, float inv_L,
                                                             float* __restrict__ mean) {
  const int bc = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < nsplit; i += 64) s += partial[bc * nsplit + i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (threadIdx.x == 0) mean[bc] = s * inv_L;
}

// ---- (x - mean) / rgb_max and NHWC packing of the frame pair (optionally row-packed) ---------------------
template <typename T>
__global__ __launch_bounds__(256) void flow_pack_pair_kernel(const float* __restrict__ in, const float* __restrict__ mean,
                                                             float rgb_max, T* __restrict__ y, int B, int H, int W,
                                                             int lpad, int wpitch, size_t total, int mode) {
  const size_t HW = (size_t)H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xp = (int)(i % wpitch);
    const size_t row = i / wpitch;          // b * H + yy
    const size_t b = row / H, yy = row - b * H;
    const int xx = xp - lpad;
    const bool live = (unsigned)xx < (unsigned)W;
    const size_t pix = yy * W + (live ? xx : 0);
    float v[2][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float m = mean[b * 3 + c];
#pragma unroll
      for (int f = 0; f < 2; ++f) v[f][c] = live ? (in[((b * 3 + c) * 2 + f) * HW + pix] - m) / rgb_max : 0.f;
    }
    if (mode == 0) {
      const float o[8] = {v[0][0], v[0][1], v[0][2], v[1][0], v[1][1], v[1][2], 0.f, 0.f};
      store8<T>(y + i * 8, o);
    } else {
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const float o[4] = {v[f][0], v[f][1], v[f][2], 0.f};
        store4c<T>(y + (((size_t)f * B + b) * H * wpitch + yy * wpitch + xp) * 4, o);
      }
    }
  }
}

// fast path of the above (W % 4 == 0, 16-byte aligned planes): one thread = 4 consecutive pixels = six 16-byte planar reads
// and 64 (mode 0) / 2 x 32 (mode 1) contiguous output bytes; the first / last thread of a row also zero the pad columns.
// ---- rgb mean + normalise + pack in ONE launch (round 4) ------------------------------------------------------------------
// FlowNet2S starts with two passes over the frame pair: the per-sample, per-colour mean over both frames (models.py:255) and
// (x - mean) / rgb_max into the stem's row-packed layout — 21.9 + 30.0 us at 16 x 512 x 384 for 75 MB read twice and 51 MB
// written.  Here a workgroup keeps its share of a sample (R rows of all six planes: up to 72 float4 per lane, ~290 VGPRs, one
// workgroup per CU) IN REGISTERS across the reduction: it sums its share, publishes the three partial sums as 8-byte
// {epoch, value} granules (one sc1 store each: the data is the flag, no fence: cdna_hip_programming.md Guideline 16, form R2),
// one wave sweeps the <= 21 x 3 granules of its sample until every tag carries this launch's epoch, every lane adds them in the
// same fixed order, and the held pixels leave normalised.  The input is read ONCE.
// Epoch without a memset node: every workgroup counts its own launches in a private word of `state` (all workgroups of a
// sample have run equally often, so they agree); state is zeroed once at allocation.  Workgroups of a sample have consecutive
// block ids and never wait for a later sample, so waiting cannot deadlock whatever else runs; the sweep is bounded (~0.1 s) and
// on time-out the means are NaN and state's last word is set.
typedef __attribute__((address_space(1))) unsigned long long gu64_t;
constexpr int kMpNit = 6;                  // 4-pixel groups per lane and plane: R * W / 4 <= 6 * 256 (36 float4 per lane: two workgroups per CU)

template <typename T>
__global__ __launch_bounds__(256, 2) void flow_mean_pack_pair_kernel(const float* __restrict__ in, float rgb_max, T* __restrict__ y,
                                                                     T* __restrict__ y3, float* __restrict__ mean_out,
                                                                     unsigned long long* state, int B, int H, int W, int R, int nW,
                                                                     int lpad, int wpitch, int lpad3, int wpitch3,
                                                                     unsigned long long* err) {
  __shared__ float s_w[4][3];
  __shared__ float s_val[128];
  __shared__ unsigned s_epoch;
  __shared__ int s_fail;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / nW, w = blockIdx.x - b * nW;
  const int r0 = w * R, rows = H - r0 < R ? H - r0 : R;
  const int W4 = W >> 2, nitems = rows * W4;
  const size_t HW = (size_t)H * W;
  gu64_t* st = (gu64_t*)state + (size_t)b * 4 * nW;     // [0, 3 nW): granules; [3 nW, 4 nW): launch counts
  if (tid == 0) {
    const unsigned e = (unsigned)__hip_atomic_load(st + 3 * nW + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    __hip_atomic_store(st + 3 * nW + w, (unsigned long long)(e == 0u ? 1u : e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_epoch = e == 0u ? 1u : e;
    s_fail = 0;
  }
  // ---- this workgroup's share, all six planes, into registers
  float4_t v[kMpNit][6];
#pragma unroll
  for (int it = 0; it < kMpNit; ++it) {
    const int idx = it * 256 + tid;
    const bool live = idx < nitems;
    const int row = live ? idx / W4 : 0, g = live ? idx - row * W4 : 0;
    const float* px = in + (size_t)b * 6 * HW + (size_t)(r0 + row) * W + 4 * g;
#pragma unroll
    for (int pl = 0; pl < 6; ++pl)
      v[it][pl] = live ? *reinterpret_cast<const float4_t*>(px + (size_t)pl * HW) : float4_t{0.f, 0.f, 0.f, 0.f};
  }
  // ---- partial sums per colour (planes 2c, 2c + 1 = the colour's two frames), fixed order
  float sc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < kMpNit; ++it)
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const float4_t q = v[it][2 * c + f];
        sc[c] += (q[0] + q[1]) + (q[2] + q[3]);
      }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sc[c] += __shfl_xor(sc[c], off);
    if (lane == 0) s_w[wave][c] = sc[c];
  }
  __syncthreads();
  const unsigned epoch = s_epoch;
  if (tid < 3) {
    const float part = (s_w[0][tid] + s_w[1][tid]) + (s_w[2][tid] + s_w[3][tid]);
    __hip_atomic_store(st + w * 3 + tid, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(part),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                      // ONE aligned 8-byte sc1 store per granule
  }
  // ---- one wave sweeps the sample's granules until every tag is this launch's
  if (wave == 0) {
    const int ng = 3 * nW;                           // <= 128: two granules per lane
    unsigned val[2] = {0u, 0u};
    for (unsigned spins = 0; ; ++spins) {
      bool ok = true;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int gi = lane + 64 * h;
        if (gi < ng) {
          const unsigned long long x = __hip_atomic_load(st + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          val[h] = (unsigned)x;
          ok = ok && (unsigned)(x >> 32) == epoch;
        }
      }
      if (__all(ok)) break;
      if (spins >= (1u << 16)) { if (lane == 0) s_fail = 1; break; }
      __builtin_amdgcn_s_sleep(32);
    }
    s_val[lane] = __uint_as_float(val[0]);
    s_val[lane + 64] = __uint_as_float(val[1]);
  }
  __syncthreads();
  float m[3];
  {
    const float inv_L = 1.0f / (float)(2 * HW);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float s = 0.f;
      for (int k = 0; k < nW; ++k) s += s_val[3 * k + c];
      m[c] = s_fail ? __uint_as_float(0x7fc00000u) : s * inv_L;
    }
  }
  if (s_fail && tid == 0) *err = 1ull;
  if (w == 0 && tid < 3) mean_out[b * 3 + tid] = m[tid];
  // ---- the held pixels leave normalised: (x - mean) / rgb_max in fp32, then the cast (flow_pack_pair4_kernel's arithmetic)
  // y: the 6-channel view [B, H, wpitch, 8] (frame 0's colours, then frame 1's); y3: the siamese view [2 B, H, wpitch3, 4] (frame f
  // of sample b = image f * B + b); either may be absent
  const float z8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, z4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < kMpNit; ++it) {
    const int idx = it * 256 + tid;
    if (idx < nitems) {
      const int row = idx / W4, g = idx - row * W4;
      float o[4][8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e][0] = (v[it][0][e] - m[0]) / rgb_max; o[e][1] = (v[it][2][e] - m[1]) / rgb_max; o[e][2] = (v[it][4][e] - m[2]) / rgb_max;
        o[e][3] = (v[it][1][e] - m[0]) / rgb_max; o[e][4] = (v[it][3][e] - m[1]) / rgb_max; o[e][5] = (v[it][5][e] - m[2]) / rgb_max;
        o[e][6] = 0.f; o[e][7] = 0.f;
      }
      if (y) {
        T* yr = y + ((size_t)b * H + r0 + row) * wpitch * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) store8<T>(yr + (size_t)(lpad + 4 * g + e) * 8, o[e]);
        if (g == 0)
          for (int xp = 0; xp < lpad; ++xp) store8<T>(yr + (size_t)xp * 8, z8);
        if (g == W4 - 1)
          for (int xp = lpad + W; xp < wpitch; ++xp) store8<T>(yr + (size_t)xp * 8, z8);
      }
      if (y3) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          T* yr = y3 + (((size_t)f * B + b) * H + r0 + row) * wpitch3 * 4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float q[4] = {o[e][3 * f], o[e][3 * f + 1], o[e][3 * f + 2], 0.f};
            store4c<T>(yr + (size_t)(lpad3 + 4 * g + e) * 4, q);
          }
          if (g == 0)
            for (int xp = 0; xp < lpad3; ++xp) store4c<T>(yr + (size_t)xp * 4, z4);
          if (g == W4 - 1)
            for (int xp = lpad3 + W; xp < wpitch3; ++xp) store4c<T>(yr + (size_t)xp * 4, z4);
        }
      }
    }
  }
}

// Same arithmetic per element as the scalar kernel ((x - mean) / rgb_max in fp32, then the cast).
template <typename T>
__global__ __launch_bounds__(256) void flow_pack_pair4_kernel(const float* __restrict__ in, const float* __restrict__ mean,
                                                              float rgb_max, T* __restrict__ y, int B, int H, int W,
                                                              int lpad, int wpitch, size_t total, int mode) {
  const size_t HW = (size_t)H * W;
  const int W4 = W >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % W4);
    const size_t row = i / W4;              // b * H + yy
    const size_t b = row / H, yy = row - b * H;
    float4_t v[2][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float m = mean[b * 3 + c];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const float4_t x4 = *reinterpret_cast<const float4_t*>(in + ((b * 3 + c) * 2 + f) * HW + yy * W + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[f][c][e] = (x4[e] - m) / rgb_max;
      }
    }
    const size_t rowbase = row * wpitch;    // pixel index of the row's first (pad) column
    const float z8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, z4[4] = {0.f, 0.f, 0.f, 0.f};
    if (mode == 0) {
      T* yr = y + rowbase * 8;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float o[8] = {v[0][0][e], v[0][1][e], v[0][2][e], v[1][0][e], v[1][1][e], v[1][2][e], 0.f, 0.f};
        store8<T>(yr + (size_t)(lpad + 4 * g + e) * 8, o);
      }
      if (g == 0)
        for (int xp = 0; xp < lpad; ++xp) store8<T>(yr + (size_t)xp * 8, z8);
      if (g == W4 - 1)
        for (int xp = lpad + W; xp < wpitch; ++xp) store8<T>(yr + (size_t)xp * 8, z8);
    } else {
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        T* yr = y + (((size_t)f * B + b) * H * wpitch + yy * wpitch) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float o[4] = {v[f][0][e], v[f][1][e], v[f][2][e], 0.f};
          store4c<T>(yr + (size_t)(lpad + 4 * g + e) * 4, o);
        }
        if (g == 0)
          for (int xp = 0; xp < lpad; ++xp) store4c<T>(yr + (size_t)xp * 4, z4);
        if (g == W4 - 1)
          for (int xp = lpad + W; xp < wpitch; ++xp) store4c<T>(yr + (size_t)xp * 4, z4);
      }
    }
  }
}

// ---- the rgb mean folded into conv1 (include/flowtrack_hip.h, ft_flow_pack_pair_sums / ft_flow_mean_fold) -------------------
// Pass 1: one workgroup = FT_PACK_SUMS_ROWS image rows of one sample.  x / rgb_max leaves as fp16 into the row-AND-column padded
// view (the padding pixels are pass 2's), the colour sums of the rows (both frames) leave as one partial per (sample, colour,
// chunk): fixed order inside the workgroup (lane-local over its groups, xor-butterfly, the four waves), fixed order in pass 2.
__global__ __launch_bounds__(256) void flow_pack_pair_sums_kernel(const float* __restrict__ in, float rgb_max, half_t* __restrict__ y,
                                                                  int H, int W, int pad, int wpitch, int nchunk,
                                                                  float* __restrict__ partial) {
  const int b = blockIdx.x / nchunk, chunk = blockIdx.x - b * nchunk;
  const int r0 = chunk * FT_PACK_SUMS_ROWS;
  const int rows = H - r0 < FT_PACK_SUMS_ROWS ? H - r0 : FT_PACK_SUMS_ROWS;
  const size_t HW = (size_t)H * W;
  const int W4 = W >> 2;
  float s[3] = {0.f, 0.f, 0.f};
  for (int idx = threadIdx.x; idx < rows * W4; idx += 256) {
    const int r = idx / W4, g = idx - r * W4;
    const int yy = r0 + r;
    float4_t v[2][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int f = 0; f < 2; ++f)
        v[f][c] = *reinterpret_cast<const float4_t*>(in + (((size_t)b * 3 + c) * 2 + f) * HW + (size_t)yy * W + 4 * g);
#pragma unroll
    for (int c = 0; c < 3; ++c)
      s[c] += ((v[0][c][0] + v[0][c][1]) + (v[0][c][2] + v[0][c][3])) + ((v[1][c][0] + v[1][c][1]) + (v[1][c][2] + v[1][c][3]));
    half_t* yr = y + (((size_t)b * (H + 2 * pad) + pad + yy) * wpitch + pad + 4 * g) * 8;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float o[8] = {v[0][0][e] / rgb_max, v[0][1][e] / rgb_max, v[0][2][e] / rgb_max,
                          v[1][0][e] / rgb_max, v[1][1][e] / rgb_max, v[1][2][e] / rgb_max, 0.f, 0.f};
      store8<half_t>(yr + e * 8, o);
    }
  }
  __shared__ float sw[4][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s[c] += __shfl_xor(s[c], off);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6][c] = s[c];
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int c = threadIdx.x;
    partial[((size_t)b * 3 + c) * nchunk + chunk] = (sw[0][c] + sw[1][c]) + (sw[2][c] + sw[3][c]);
  }
}

// Pass 2: grid (B, K).  Every workgroup re-derives its sample's means (3 x nchunk floats, the same order everywhere), then
// takes its share of the sample's padding pixels; workgroup (b, 0) also writes mean[b] and the sample's shift vector.
__global__ __launch_bounds__(256) void flow_mean_fold_kernel(const float* __restrict__ partial, int nchunk, float inv_L, float rgb_max,
                                                             half_t* __restrict__ y, int H, int W, int pad, int wpitch,
                                                             const float* __restrict__ wsum, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int Cout, float* __restrict__ shift_n,
                                                             float* __restrict__ mean) {
  const int b = blockIdx.x;
  float m[3] = {0.f, 0.f, 0.f}, mf[3];
  const float* pb = partial + (size_t)b * 3 * nchunk;
  for (int i = threadIdx.x & 63; i < nchunk; i += 64) {     // the three colours' loads of a round fly together
    m[0] += pb[i];
    m[1] += pb[nchunk + i];
    m[2] += pb[2 * nchunk + i];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m[c] += __shfl_xor(m[c], off);
    m[c] *= inv_L;
    mf[c] = (float)(half_t)(m[c] / rgb_max);
  }
  if (blockIdx.y == 0) {
    if (threadIdx.x < 3) mean[b * 3 + threadIdx.x] = m[threadIdx.x];
    for (int co = threadIdx.x; co < Cout; co += 256) {
      const float* w = wsum + co * 8;
      const float corr = ((mf[0] * w[0] + mf[1] * w[1]) + mf[2] * w[2]) + ((mf[0] * w[3] + mf[1] * w[4]) + mf[2] * w[5]);
      shift_n[(size_t)b * Cout + co] = (shift ? shift[co] : 0.f) - (scale ? scale[co] : 1.f) * corr;
    }
  }
  // padding pixels of the sample: `pad` full rows on top, `pad` at the bottom, then per image row the left `pad` and the right
  // wpitch - pad - W columns
  const float o[8] = {mf[0], mf[1], mf[2], mf[0], mf[1], mf[2], 0.f, 0.f};
  const int nfull = 2 * pad * wpitch, nside = wpitch - W;
  const int total = nfull + H * nside;
  half_t* yb = y + (size_t)b * (H + 2 * pad) * wpitch * 8;
  for (int i = blockIdx.y * 256 + threadIdx.x; i < total; i += gridDim.y * 256) {
    int row, col;
    if (i < nfull) {
      const int r = i / wpitch;
      col = i - r * wpitch;
      row = r < pad ? r : H + r;            // r in [pad, 2 pad): rows H + pad ...
    } else {
      const int j = i - nfull, r = j / nside, k = j - r * nside;
      row = pad + r;
      col = k < pad ? k : W + k;            // k in [pad, nside): columns pad + W ...
    }
    store8<half_t>(yb + ((size_t)row * wpitch + col) * 8, o);
  }
}

// ---- nn.Upsample(scale_factor=4, mode='bilinear'), align_corners=False, times `mul` -------------
__global__ __launch_bounds__(256) void upsample_bilinear4x_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                  int h, int w, size_t total, float mul) {
  // one thread = the 4 x 4 outputs over one source cell: output rows 4 iy + {0, 1} blend source rows (iy - 1, iy), rows
  // 4 iy + {2, 3} blend (iy, iy + 1) (clamped at the borders exactly as the per-output formulas below clamp them), all
  // four share three source columns: 12 loads for 16 outputs and four 16-byte stores, one per output row (lanes = consecutive
  // cells: 1 KiB per wave-store).  The per-output arithmetic is the one-output-row-per-thread kernel's, term for term.
  const int W = 4 * w;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int jx = (int)(i % w);                 // source column the 4 output columns straddle
    const size_t t = i / w;
    const int iy = (int)(t % h);
    const size_t nc = t / h;
    const float* px = x + nc * (size_t)h * w;
    int x0[4], x1[4];
    float lx[4], hx[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float sx = ((float)(4 * jx + e) + 0.5f) * 0.25f - 0.5f;
      sx = sx < 0.f ? 0.f : sx;
      x0[e] = (int)sx;
      x1[e] = x0[e] + (x0[e] < w - 1 ? 1 : 0);
      lx[e] = sx - (float)x0[e];
      hx[e] = 1.f - lx[e];
    }
    // the three source columns the four outputs of a row touch: x0[0] = x0[1], x1[0] = x1[1] = x0[2] = x0[3], x1[2] = x1[3]
    const int ca = x0[0], cb = x0[2], cc = x1[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int oy = 4 * iy + 2 * half;
      float sy = ((float)oy + 0.5f) * 0.25f - 0.5f;
      sy = sy < 0.f ? 0.f : sy;
      const int y0 = (int)sy;                     // the same for output row oy + 1
      const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
      const float* p0 = px + (size_t)y0 * w;
      const float* p1 = px + (size_t)y1 * w;
      const float r0[3] = {p0[ca] * mul, p0[cb] * mul, p0[cc] * mul};
      const float r1[3] = {p1[ca] * mul, p1[cb] * mul, p1[cc] * mul};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float syk = ((float)(oy + k) + 0.5f) * 0.25f - 0.5f;
        syk = syk < 0.f ? 0.f : syk;
        const float ly = syk - (float)y0, hy = 1.f - ly;
        float4_t out;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // x0[e] / x1[e] among (ca, cb, cc), picked by VALUE: at the left / right border the clamped columns repeat
          const float v00 = x0[e] == ca ? r0[0] : (x0[e] == cb ? r0[1] : r0[2]);
          const float v01 = x1[e] == cb ? r0[1] : (x1[e] == cc ? r0[2] : r0[0]);
          const float v10 = x0[e] == ca ? r1[0] : (x0[e] == cb ? r1[1] : r1[2]);
          const float v11 = x1[e] == cb ? r1[1] : (x1[e] == cc ? r1[2] : r1[0]);
          out[e] = hy * (hx[e] * v00 + lx[e] * v01) + ly * (hx[e] * v10 + lx[e] * v11);
        }
        *reinterpret_cast<float4_t*>(y + (nc * (size_t)(4 * h) + (size_t)(oy + k)) * W + 4 * jx) = out;
      }
    }
  }
}

// ---- BatchNorm2d batch statistics (training-mode forward / running-stat recalibration), NHWC -------------
// The inference path folds eval-mode BN into the conv epilogue and needs no reduction; this kernel covers the
// `model.train()`-style forward of nn.BatchNorm2d (tools/pose/main.py:207,342: per-channel mean and biased
// variance over N*H*W) for recalibrating running statistics.  Lanes own channels (16-byte vectors along C),
// rows of pixels are strided over the workgroup's waves, per-wave partials meet in LDS, workgroups combine
// with one fp32 atomicAdd pair per channel (sum, sum of squares; fp32, Welford is not needed at these counts:
// tolerance stated in the test).
template <typename T>
__global__ __launch_bounds__(256) void bn_batch_stats_kernel(const T* __restrict__ x, int C, int cstride, size_t npix,
                                                             float* __restrict__ sums /* [2*C], pre-zeroed */) {
  const int cvecs = C / 8;
  const int lane_v = threadIdx.x % 32, rowl = threadIdx.x / 32;   // 32 vector lanes x 8 pixel rows per pass
  __shared__ float s_part[8][32][16];
  for (int v0 = 0; v0 < cvecs; v0 += 32) {
    const int v = v0 + lane_v;
    float a[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = q[e] = 0.f;
    if (v < cvecs) {
      for (size_t pix = (size_t)blockIdx.x * 8 + rowl; pix < npix; pix += (size_t)gridDim.x * 8) {
        float f[8];
        load8<T>(x + pix * cstride + v * 8, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[e] += f[e]; q[e] += f[e] * f[e]; }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { s_part[rowl][lane_v][e] = a[e]; s_part[rowl][lane_v][8 + e] = q[e]; }
    __syncthreads();
    if (rowl == 0 && v < cvecs) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += s_part[r][lane_v][e];
        atomicAdd(&sums[(e < 8 ? 0 : C) + v * 8 + (e & 7)], t);
      }
    }
    __syncthreads();
  }
}

__global__ void bn_finish_stats_kernel(const float* __restrict__ sums, int C, float inv_n, float* __restrict__ mean,
                                       float* __restrict__ var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float m = sums[c] * inv_n;
    mean[c] = m;
    const float v = sums[C + c] * inv_n - m * m;
    var[c] = v > 0.f ? v : 0.f;
  }
}

static inline int grid_for(size_t total) {
  size_t g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace ft

using namespace ft;

extern "C" int ft_pack_nchw_to_nhwc(const float* x, void* y, int N, int C, int H, int W, int cpad, int lpad, int wpitch,
                                    int dtype, ft_stream_t stream) {
  if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || cpad < C || cpad % 4) return FT_ERR_INVALID_ARG;
  if (lpad < 0 || wpitch < lpad + W) return FT_ERR_INVALID_ARG;
  if (dtype != FT_F16 && dtype != FT_F32) return FT_ERR_INVALID_ARG;
  const size_t total = (size_t)N * H * wpitch;
  if (dtype == FT_F16 && C <= 3 && cpad == 4 && W % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const size_t groups = (size_t)N * H * (W / 4);
    hipLaunchKernelGGL(pack_nchw_rows4_kernel, dim3(grid_for(groups)), dim3(256), 0, as_stream(stream), x, static_cast<half_t*>(y),
                       C, H, W, lpad, wpitch, groups);
    FT_LAUNCH_CHECK("pack_nchw_rows4_kernel");
    return FT_OK;
  }
  if (dtype == FT_F16)
    hipLaunchKernelGGL(pack_nchw_to_nhwc_kernel<half_t>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x,
                       static_cast<half_t*>(y), C, H, W, cpad, lpad, wpitch, total);
  else
    hipLaunchKernelGGL(pack_nchw_to_nhwc_kernel<float>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x,
                       static_cast<float*>(y), C, H, W, cpad, lpad, wpitch, total);
  FT_LAUNCH_CHECK("pack_nchw_to_nhwc_kernel");
  return FT_OK;
}

extern "C" int ft_unpack_nhwc_to_nchw(const void* x, float* y, int N, int C, int H, int W, int x_cstride, int x_coff,
                                      int dtype, ft_stream_t stream) {
  if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || x_coff < 0 || x_cstride < x_coff + C) return FT_ERR_INVALID_ARG;
  if (dtype != FT_F16 && dtype != FT_F32) return FT_ERR_INVALID_ARG;
  const size_t HW = (size_t)H * W, total = (size_t)N * C * HW;
  if (dtype == FT_F16)
    hipLaunchKernelGGL(unpack_nhwc_to_nchw_kernel<half_t>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream),
                       static_cast<const half_t*>(x), y, C, HW, total, x_cstride, x_coff);
  else
    hipLaunchKernelGGL(unpack_nhwc_to_nchw_kernel<float>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream),
                       static_cast<const float*>(x), y, C, HW, total, x_cstride, x_coff);
  FT_LAUNCH_CHECK("unpack_nhwc_to_nchw_kernel");
  return FT_OK;
}

extern "C" int ft_maxpool3x3s2_fwd(const void* x, void* y, int N, int Hi, int Wi, int C, int dtype, ft_stream_t stream) {
  if (!x || !y || N <= 0 || Hi <= 0 || Wi <= 0 || C <= 0 || C % 8) return FT_ERR_INVALID_ARG;
  if (dtype != FT_F16 && dtype != FT_F32) return FT_ERR_INVALID_ARG;
  const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
  const size_t total = (size_t)N * Ho * Wo * (C / 8);
  if (dtype == FT_F16)
    hipLaunchKernelGGL(maxpool3x3s2_kernel<half_t>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream),
                       static_cast<const half_t*>(x), static_cast<half_t*>(y), Hi, Wi, Ho, Wo, C / 8, total);
  else
    hipLaunchKernelGGL(maxpool3x3s2_kernel<float>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream),
                       static_cast<const float*>(x), static_cast<float*>(y), Hi, Wi, Ho, Wo, C / 8, total);
  FT_LAUNCH_CHECK("maxpool3x3s2_kernel");
  return FT_OK;
}

extern "C" int ft_heatmap_max_preds(const float* heatmaps, int N, int K, int H, int W, int adjust_coords, int32_t* idx,
                                    float* score, float* coords, ft_stream_t stream) {
  if (!heatmaps || !idx || !score || !coords || N <= 0 || K <= 0 || H <= 0 || W <= 0) return FT_ERR_INVALID_ARG;
  hipLaunchKernelGGL(heatmap_max_preds_kernel, dim3(N * K), dim3(256), 0, as_stream(stream), heatmaps, H, W,
                     adjust_coords, idx, score, coords, 1, 2);
  FT_LAUNCH_CHECK("heatmap_max_preds_kernel");
  return FT_OK;
}

extern "C" int ft_heatmap_keypoint_rows(const float* heatmaps, int N, int K, int H, int W, int adjust_coords, int32_t* idx,
                                        float* rows, ft_stream_t stream) {
  if (!heatmaps || !idx || !rows || N <= 0 || K <= 0 || H <= 0 || W <= 0) return FT_ERR_INVALID_ARG;
  hipLaunchKernelGGL(heatmap_max_preds_kernel, dim3(N * K), dim3(256), 0, as_stream(stream), heatmaps, H, W,
                     adjust_coords, idx, rows + 2, rows, 3, 3);
  FT_LAUNCH_CHECK("heatmap_max_preds_kernel");
  return FT_OK;
}

extern "C" int ft_heatmap_min_margin(const float* heatmaps, int N, int K, int H, int W, float* min_margin, ft_stream_t stream) {
  if (!heatmaps || !min_margin || N <= 0 || K <= 0 || H <= 0 || W <= 0 || (long long)H * W < 2) return FT_ERR_INVALID_ARG;
  hipLaunchKernelGGL(heatmap_min_margin_kernel, dim3(N), dim3(256), 0, as_stream(stream), heatmaps, K, H * W, min_margin);
  FT_LAUNCH_CHECK("heatmap_min_margin_kernel");
  return FT_OK;
}

extern "C" int ft_heatmap_argmax_screen(const float* heatmaps, int N, int K, int H, int W, float rel_bound, int32_t* flags, float* stats,
                                        ft_stream_t stream) {
  if (!heatmaps || !flags || !stats || N <= 0 || K <= 0 || H <= 0 || W <= 0 || (long long)H * W < 2 || !(rel_bound >= 0.f))
    return FT_ERR_INVALID_ARG;
  if (K > 256) return FT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(heatmap_argmax_screen_kernel, dim3(N), dim3(256), 0, as_stream(stream), heatmaps, K, H * W, rel_bound, flags, stats);
  FT_LAUNCH_CHECK("heatmap_argmax_screen_kernel");
  return FT_OK;
}

extern "C" int ft_gather_flagged_rows(const int32_t* flags, int N, const void* src, long long row_bytes, void* dst, int32_t* header,
                                      ft_stream_t stream) {
  if (!flags || !src || !dst || !header || N <= 0 || row_bytes <= 0) return FT_ERR_INVALID_ARG;
  if (N > 1024 || row_bytes % 16 || (reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 15)) return FT_ERR_UNSUPPORTED;
  const size_t row16 = (size_t)row_bytes / 16;
  const unsigned parts = (unsigned)((row16 + 256 * 16 - 1) / (256 * 16));      // ~16 pieces of 16 bytes per thread
  hipLaunchKernelGGL(gather_flagged_rows_kernel, dim3(N, parts < 1 ? 1 : (parts > 64 ? 64 : parts)), dim3(256), 0, as_stream(stream), flags, N,
                     static_cast<const uint4_t*>(src), row16, static_cast<uint4_t*>(dst), header);
  FT_LAUNCH_CHECK("gather_flagged_rows_kernel");
  return FT_OK;
}

extern "C" int ft_flow_rgb_mean(const float* inputs, int B, int H, int W, float* partial, float* mean,
                                ft_stream_t stream) {
  if (!inputs || !partial || !mean || B <= 0 || H <= 0 || W <= 0) return FT_ERR_INVALID_ARG;
  const size_t L = (size_t)2 * H * W;
  hipLaunchKernelGGL(rgb_partial_sum_kernel, dim3(B * 3, FT_RGB_MEAN_SPLITS), dim3(256), 0, as_stream(stream), inputs, L,
                     partial);
  FT_LAUNCH_CHECK("rgb_partial_sum_kernel");
  hipLaunchKernelGGL(rgb_mean_finish_kernel, dim3(B * 3), dim3(64), 0, as_stream(stream), partial, FT_RGB_MEAN_SPLITS,
                     1.0f / (float)L, mean);
  FT_LAUNCH_CHECK("rgb_mean_finish_kernel");
  return FT_OK;
}

extern "C" int ft_flow_pack_pair(const float* inputs, const float* mean, float rgb_max, void* y, int B, int H, int W,
                                 int mode, int lpad, int wpitch, int dtype, ft_stream_t stream) {
  if (!inputs || !mean || !y || B <= 0 || H <= 0 || W <= 0 || (mode != 0 && mode != 1) || rgb_max == 0.f)
    return FT_ERR_INVALID_ARG;
  if (lpad < 0 || wpitch < lpad + W) return FT_ERR_INVALID_ARG;
  if (dtype != FT_F16 && dtype != FT_F32) return FT_ERR_INVALID_ARG;
  if (W % 4 == 0 && (reinterpret_cast<uintptr_t>(inputs) & 15) == 0) {
    const size_t groups = (size_t)B * H * (W / 4);
    if (dtype == FT_F16)
      hipLaunchKernelGGL(flow_pack_pair4_kernel<half_t>, dim3(grid_for(groups)), dim3(256), 0, as_stream(stream), inputs, mean,
                         rgb_max, static_cast<half_t*>(y), B, H, W, lpad, wpitch, groups, mode);
    else
      hipLaunchKernelGGL(flow_pack_pair4_kernel<float>, dim3(grid_for(groups)), dim3(256), 0, as_stream(stream), inputs, mean,
                         rgb_max, static_cast<float*>(y), B, H, W, lpad, wpitch, groups, mode);
    FT_LAUNCH_CHECK("flow_pack_pair4_kernel");
    return FT_OK;
  }
  const size_t total = (size_t)B * H * wpitch;
  if (dtype == FT_F16)
    hipLaunchKernelGGL(flow_pack_pair_kernel<half_t>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), inputs, mean,
                       rgb_max, static_cast<half_t*>(y), B, H, W, lpad, wpitch, total, mode);
  else
    hipLaunchKernelGGL(flow_pack_pair_kernel<float>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), inputs, mean,
                       rgb_max, static_cast<float*>(y), B, H, W, lpad, wpitch, total, mode);
  FT_LAUNCH_CHECK("flow_pack_pair_kernel");
  return FT_OK;
}

extern "C" long long ft_flow_pack_pair_sums_chunks(int H) { return H > 0 ? (H + FT_PACK_SUMS_ROWS - 1) / FT_PACK_SUMS_ROWS : 0; }

extern "C" int ft_flow_pack_pair_sums(const float* inputs, float rgb_max, void* y, int B, int H, int W, int pad, int wpitch, int dtype,
                                      float* partial, ft_stream_t stream) {
  if (!inputs || !y || !partial || B <= 0 || H <= 0 || W <= 0 || pad < 0 || wpitch < W + 2 * pad || rgb_max == 0.f)
    return FT_ERR_INVALID_ARG;
  if (dtype != FT_F16 || W % 4 != 0 || (reinterpret_cast<uintptr_t>(inputs) & 15) != 0) return FT_ERR_UNSUPPORTED;
  const int nchunk = (int)ft_flow_pack_pair_sums_chunks(H);
  if ((long long)B * nchunk > 0x7fffffffLL) return FT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(flow_pack_pair_sums_kernel, dim3(B * nchunk), dim3(256), 0, as_stream(stream), inputs, rgb_max,
                     static_cast<half_t*>(y), H, W, pad, wpitch, nchunk, partial);
  FT_LAUNCH_CHECK("flow_pack_pair_sums_kernel");
  return FT_OK;
}

extern "C" int ft_flow_mean_fold(const float* partial, float rgb_max, void* y, int B, int H, int W, int pad, int wpitch, int dtype,
                                 const float* wsum, const float* scale, const float* shift, int Cout, float* shift_n, float* mean,
                                 ft_stream_t stream) {
  if (!partial || !y || !wsum || !shift_n || !mean || B <= 0 || H <= 0 || W <= 0 || pad < 0 || wpitch < W + 2 * pad || Cout <= 0 ||
      rgb_max == 0.f)
    return FT_ERR_INVALID_ARG;
  if (dtype != FT_F16) return FT_ERR_UNSUPPORTED;
  const int nchunk = (int)ft_flow_pack_pair_sums_chunks(H);
  const int total = 2 * pad * wpitch + H * (wpitch - W);
  int k = (total + 1023) / 1024;            // ~4 padding pixels per thread
  k = k < 1 ? 1 : (k > 16 ? 16 : k);
  hipLaunchKernelGGL(flow_mean_fold_kernel, dim3(B, k), dim3(256), 0, as_stream(stream), partial, nchunk,
                     1.0f / ((float)2 * H * W), rgb_max, static_cast<half_t*>(y), H, W, pad, wpitch, wsum, scale, shift, Cout, shift_n,
                     mean);
  FT_LAUNCH_CHECK("flow_mean_fold_kernel");
  return FT_OK;
}

constexpr int kMpMaxGrid = 512;            // 256 CUs x two workgroups (__launch_bounds__(256, 2)): every workgroup of the grid is resident
// rows per workgroup / workgroups per sample of flow_mean_pack_pair_kernel, or false where it does not apply
static bool mean_pack_plan(int H, int W, int* R, int* nW) {
  if (H <= 0 || W < 4 || (W & 3)) return false;
  int r = (kMpNit * 256 * 4) / W;
  if (r < 1) return false;
  r = r < H ? r : H;
  const int n = (H + r - 1) / r;
  if (3 * n > 128) return false;                    // two granules per lane in the sweep
  *R = (H + n - 1) / n;                             // same workgroup count, balanced rows
  *nW = n;
  return true;
}

extern "C" long long ft_flow_mean_pack_pair_state_words(int B, int H, int W) {
  int R, nW;
  if (B <= 0 || !mean_pack_plan(H, W, &R, &nW) || (long long)B * nW > kMpMaxGrid) return 0;
  return (long long)B * 4 * nW + 1;
}

extern "C" int ft_flow_mean_pack_pair(const float* inputs, float rgb_max, void* y, int lpad, int wpitch, void* y3, int lpad3, int wpitch3,
                                      int B, int H, int W, int dtype, unsigned long long* state, float* mean, ft_stream_t stream) {
  if (!inputs || (!y && !y3) || !state || !mean || B <= 0 || rgb_max == 0.f) return FT_ERR_INVALID_ARG;
  if ((y && (lpad < 0 || wpitch < lpad + W)) || (y3 && (lpad3 < 0 || wpitch3 < lpad3 + W))) return FT_ERR_INVALID_ARG;
  if (dtype != FT_F16 && dtype != FT_F32) return FT_ERR_INVALID_ARG;
  int R, nW;
  if (!mean_pack_plan(H, W, &R, &nW) || (reinterpret_cast<uintptr_t>(inputs) & 15) != 0) return FT_ERR_UNSUPPORTED;
  if ((long long)B * nW > kMpMaxGrid) return FT_ERR_UNSUPPORTED;   // the sample's workgroups wait for each other: all resident
  unsigned long long* err = state + (size_t)B * 4 * nW;
  if (dtype == FT_F16)
    hipLaunchKernelGGL(flow_mean_pack_pair_kernel<half_t>, dim3(B * nW), dim3(256), 0, as_stream(stream), inputs, rgb_max,
                       static_cast<half_t*>(y), static_cast<half_t*>(y3), mean, state, B, H, W, R, nW, lpad, wpitch, lpad3, wpitch3, err);
  else
    hipLaunchKernelGGL(flow_mean_pack_pair_kernel<float>, dim3(B * nW), dim3(256), 0, as_stream(stream), inputs, rgb_max,
                       static_cast<float*>(y), static_cast<float*>(y3), mean, state, B, H, W, R, nW, lpad, wpitch, lpad3, wpitch3, err);
  FT_LAUNCH_CHECK("flow_mean_pack_pair_kernel");
  return FT_OK;
}

extern "C" int ft_upsample_bilinear4x(const float* x, float* y, int N, int C, int h, int w, float mul,
                                      ft_stream_t stream) {
  if (!x || !y || N <= 0 || C <= 0 || h <= 0 || w <= 0) return FT_ERR_INVALID_ARG;
  const size_t cells = (size_t)N * C * h * w;         // 16 outputs per source cell, one cell per thread
  hipLaunchKernelGGL(upsample_bilinear4x_kernel, dim3(grid_for(cells)), dim3(256), 0, as_stream(stream), x, y, h, w,
                     cells, mul);
  FT_LAUNCH_CHECK("upsample_bilinear4x_kernel");
  return FT_OK;
}

extern "C" int ft_bn_batch_stats(const void* x, int N, int H, int W, int C, int x_cstride, int dtype, float* workspace,
                                 float* mean, float* var, ft_stream_t stream) {
  if (!x || !workspace || !mean || !var || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || x_cstride % 8 || x_cstride < C)
    return FT_ERR_INVALID_ARG;
  if (dtype != FT_F16 && dtype != FT_F32) return FT_ERR_INVALID_ARG;
  const size_t npix = (size_t)N * H * W;
  FT_HIP_CHECK(hipMemsetAsync(workspace, 0, sizeof(float) * 2 * (size_t)C, as_stream(stream)));
  size_t g = (npix + 63) / 64;
  g = g < 1 ? 1 : (g > 2048 ? 2048 : g);
  if (dtype == FT_F16)
    hipLaunchKernelGGL(bn_batch_stats_kernel<half_t>, dim3((unsigned)g), dim3(256), 0, as_stream(stream),
                       static_cast<const half_t*>(x), C, x_cstride, npix, workspace);
  else
    hipLaunchKernelGGL(bn_batch_stats_kernel<float>, dim3((unsigned)g), dim3(256), 0, as_stream(stream),
                       static_cast<const float*>(x), C, x_cstride, npix, workspace);
  FT_LAUNCH_CHECK("bn_batch_stats_kernel");
  hipLaunchKernelGGL(bn_finish_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), workspace, C,
                     1.0f / (float)npix, mean, var);
  FT_LAUNCH_CHECK("bn_finish_stats_kernel");
  return FT_OK;
}
