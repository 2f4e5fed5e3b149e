// bottleneck_rstat_kernel — the REGISTER-STATIONARY, PERSISTENT-STRIP form of the 64-plane identity Bottleneck (fp16):
// conv1 1x1 + bn1 + relu -> conv2 3x3 + bn2 + relu -> conv3 1x1 + bn3 + identity residual + relu in ONE launch
// (reference: Bottleneck.forward, lib/pose/models/blocks.py:105-120; layer1.1 / layer1.2 of resnet.py:29-36).  gfx950 only.
//
// Why a second form beside bottleneck_fused_kernel (bottleneck.hip).  That kernel owns a 128-pixel patch per workgroup: every
// patch pulls its 136 KB of weights through L2 -> LDS again, computes conv1 on a 1.5x halo, has its x loads in flight only
// during one of its three phases and reads 1.5 KB of LDS per MFMA; with no memory traffic at all it still takes 41 us per
// block at batch 64 (FT_BNK_DBG ablations), against an HBM floor of ~34 us (100 MB in, 100 MB out).  Here:
//   * ONE 8-wave workgroup per CU owns a STRIP of SR full-width rows of one image (batch 64, 64 x 48 maps: 16 rows, 256 strips)
//     and walks it in steps of 64 consecutive pixels (row-major, flat: a 3x3 tap is a shift by dy * W + dx in a flat ring);
//     conv1's halo is the row above / below the strip, computed once (1.12x at 16 rows), nothing is recomputed between steps.
//   * The 139 KB of weights are loaded ONCE per workgroup and stay in registers as MFMA A operands: the waves are two GROUPS,
//     G0 (waves 0-3) holds W1 and W3 (its wave (nt, pt): channel tile nt of conv1 / channel tiles 4 nt .. 4 nt + 3 of conv3,
//     pixel tile pt), G1 (waves 4-7) holds W2 (channel tile nt, pixel tile pt: 36 fragments = 144 registers).  One wave of
//     each group per SIMD: while G0 runs its epilogues (80 results per lane and step) G1's 36 MFMAs own the matrix pipe.
//   * Software pipeline, ONE s_barrier per step.  Iteration i:  G0: conv3(step i - 1) from T2 + residual -> y; then the
//     residual fetch of step i; then conv1(step i + 2) from the x ring -> T1 ring.   G1: LDS-DMA of x(step i + 3) into the
//     ring (+ an L2 touch of step i + 6: the demand load one iteration ahead then hits L2), conv2(step i) from T1 -> T2.
//   * x ring: two 32-KiB step buffers (64 pixels x 512 B, 16-byte chunks XOR-ed with pixel & 15 on the SOURCE side of the
//     DMA: conflict-free ds_read_b128 fragments).  T1: a 256-pixel flat ring of 128-byte rows (live span 2 W + 130 pixels:
//     W <= 62).  T2: two 64-pixel tiles.  The residual does not wait in LDS for three steps: each G0 wave DMAs its own
//     32-pixel x 128-channel piece of x again (an L2 hit) into a wave-private 8-KiB tile, adds it in the accumulator layout,
//     writes y in place and stores the tile as whole 16-byte pieces (256 contiguous bytes per pixel).
//   * Every in-loop LDS access is inline asm with hand-placed lgkmcnt waits (hipcc puts `s_waitcnt vmcnt(0)` in front of
//     compiler-visible LDS accesses while an LDS-DMA is in flight: conv_wstat.hip).
// Same operand layouts, same K order, same epilogue expressions as bottleneck_fused_kernel: the results are bit-identical.
#include "bottleneck_rstat.h"

#include <stdlib.h>

#include "conv_common.h"

namespace ft {
namespace {

typedef uint32_t uint2_t __attribute__((ext_vector_type(2)));

struct BnrParams {
  const char* x;
  char* y;
  const char *w1, *w2, *w3;
  const float* tab;  // [s1 64 | b1 64 | s2 64 | b2 64 | s3 256 | b3 256]
  int N, H, W;
  int SR, S;         // rows per strip, strips per image
  int x_cstride, x_coff, y_cstride, y_coff;
  unsigned x_bytes, y_bytes;
  int total;
  int dbg;           // FT_BNR_DBG (dev): 1 no x / residual loads, 4 no stores, 8 no L2 touch, 32 per-wave cycle counts
};

constexpr int kXB = 32768;        // one step of x: 64 pixels x 512 B
constexpr int kOffX = 0;          // two step buffers
constexpr int kOffStg = 65536;    // four wave-private 8-KiB residual / output tiles (32 pixels x 256 B)
constexpr int kOffT1 = 98304;     // 256-slot flat ring x 128 B
constexpr int kOffT2 = 131072;    // two 64-pixel x 128-B tiles
constexpr int kOffTab = 147456;   // 3 KiB folded-BN table
constexpr int kOffZero = 150528;  // 64 B of zeros (x-border taps)
constexpr int kOffScr = 150592;   // 1 KiB scratch of the L2 touch loads
constexpr int kLds = 151616;

#define BNR_BARRIER() asm volatile("s_barrier" ::: "memory")

__device__ __forceinline__ void rd128(uint4_t& v, unsigned a) { asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); }
template <int OFF>
__device__ __forceinline__ void rd128o(uint4_t& v, unsigned a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF)); }
template <int OFF>
__device__ __forceinline__ void rd128fo(float4_t& v, unsigned a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF)); }
__device__ __forceinline__ void rd64(uint2_t& v, unsigned a) { asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a)); }
__device__ __forceinline__ void wr64(unsigned a, uint2_t v) { asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory"); }
// counted LDS waits that carry the registers they release (no MFMA / VALU use can move above them)
template <int N>
__device__ __forceinline__ void wait4(uint4_t (&f)[4]) { asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(N)); }
template <int N>
__device__ __forceinline__ void wait8(uint4_t (&f)[8]) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "n"(N));
}
__device__ __forceinline__ void wait_tab(float4_t (&sc)[4], float4_t (&sh)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sc[0]), "+v"(sc[1]), "+v"(sc[2]), "+v"(sc[3]), "+v"(sh[0]), "+v"(sh[1]), "+v"(sh[2]), "+v"(sh[3]));
}
__device__ __forceinline__ void wait_res(float4_t (&sc)[4], float4_t (&sh)[4], uint2_t (&rs)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(sc[0]), "+v"(sc[1]), "+v"(sc[2]), "+v"(sc[3]), "+v"(sh[0]), "+v"(sh[1]), "+v"(sh[2]), "+v"(sh[3]), "+v"(rs[0]), "+v"(rs[1]),
                 "+v"(rs[2]), "+v"(rs[3]));
}

typedef float float2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
// relu(fp16(v)) == fp16(relu(v)) (rounding is monotonic and keeps the sign); packed: v_cvt_pk_f16_f32 + v_pk_max_f16.  A negative
// value that rounds to -0 comes out as a zero of either sign: equal as a number.
__device__ __forceinline__ half2_t relu2(float2_t v) {
  const half2_t h = __builtin_convertvector(v, half2_t);
  return __builtin_elementwise_max(h, half2_t{(half_t)0.f, (half_t)0.f});
}
// accumulator registers 4 g .. 4 g + 3 -> relu(acc * scale + shift) as four fp16 (two v_pk_fma_f32)
__device__ __forceinline__ uint2_t epi4(const float16_t& acc, const float4_t& sc, const float4_t& sh, int g) {
  half4_t hv;
#pragma unroll
  for (int e2 = 0; e2 < 2; ++e2) {
    const float2_t a = {acc[g * 4 + e2 * 2], acc[g * 4 + e2 * 2 + 1]};
    const float2_t k = {sc[e2 * 2], sc[e2 * 2 + 1]}, b = {sh[e2 * 2], sh[e2 * 2 + 1]};
    const half2_t o = relu2(__builtin_elementwise_fma(a, k, b));
    hv[e2 * 2] = o[0];
    hv[e2 * 2 + 1] = o[1];
  }
  return __builtin_bit_cast(uint2_t, hv);
}

__global__ __launch_bounds__(512, 1) void bottleneck_rstat_kernel(const BnrParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, gw = wave & 3, nt = wave & 1, pt = (wave >> 1) & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;

  // XCD-aware order: block b runs on XCD b % 8; each XCD gets one contiguous range of strips (neighbouring strips share their
  // halo rows through that XCD's L2)
  int logical;
  {
    const int b = blockIdx.x;
    const int q = p.total >> 3, r = p.total & 7, xcd = b & 7, loc = b >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int n = logical / p.S, si = logical - n * p.S;
  const int W = p.W, HW = p.H * W;
  const int r0 = si * p.SR;
  const int sra = p.H - r0 < p.SR ? p.H - r0 : p.SR;
  const int npx = sra * W;                       // output pixels of the strip, flat index q in [0, npx)
  const int NS = (npx + 63) >> 6;                // output steps
  const int jlo = -((W + 1 + 63) >> 6);          // conv1 covers q in [-(W + 1), npx + W]
  const int jhi = (npx + W) >> 6;
  const int r0W = r0 * W, imgbase = n * HW;

  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.w1), 0, 64 * 256 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.w2), 0, 64 * 576 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.w3), 0, 256 * 64 * 2, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;

  // ---- this wave's weights, MFMA A fragments (lane (m = l31, lhi): 8 halves k = 16 s + 8 lhi .. of row m) ----------------
  // G0: wt[0..15] = W1 rows nt * 32 + m, k-steps 0..15; wt[16 + 4 mt + kk] = W3 rows (4 nt + mt) * 32 + m, k-step kk.
  // G1: wt[4 tap + kk] = W2 rows nt * 32 + m, k = tap * 64 + 16 kk ...
  uint4_t wt[36];
  if (grp == 0) {
#pragma unroll
    for (int k = 0; k < 16; ++k) wt[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w1, (unsigned)(((nt * 32 + l31) * 256 + k * 16 + lhi * 8) * 2), 0, 0);
#pragma unroll
    for (int k = 0; k < 16; ++k)
      wt[16 + k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w3, (unsigned)((((nt * 4 + (k >> 2)) * 32 + l31) * 64 + (k & 3) * 16 + lhi * 8) * 2), 0, 0);
#pragma unroll
    for (int k = 32; k < 36; ++k) wt[k] = uint4_t{0u, 0u, 0u, 0u};
  } else {
#pragma unroll
    for (int k = 0; k < 36; ++k)
      wt[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w2, (unsigned)(((nt * 32 + l31) * 576 + (k >> 2) * 64 + (k & 3) * 16 + lhi * 8) * 2), 0, 0);
  }
  // folded-BN table and the zero row (plain accesses: no LDS-DMA is in flight yet)
  if (tid < 192) *reinterpret_cast<float4_t*>(smem + kOffTab + tid * 16) = *reinterpret_cast<const float4_t*>(p.tab + tid * 4);
  if (tid >= 192 && tid < 196) *reinterpret_cast<uint4_t*>(smem + kOffZero + (tid - 192) * 16) = uint4_t{0u, 0u, 0u, 0u};
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  // ... and the compiler has to know the weights are there (it cannot see the wait above and would guard their first in-loop
  // use with a vmcnt(0) of its own: the whole x stream)
#pragma unroll
  for (int k = 0; k < 36; ++k) asm volatile("" : "+v"(wt[k]));
  BNR_BARRIER();

  // ---- per-lane constants -----------------------------------------------------------------------------------------------
  // Few bases, everything else derived from them by ONE VALU operation or an instruction offset at the point of use; each
  // iteration passes the bases through an empty asm so that hipcc does not hoist the ~120 derived addresses out of the loop
  // (it did, and spilled them: 167 scratch registers in the first version).
  int ptile = pt * 32 + l31;                                    // this lane's pixel inside a step (B-fragment row)
  unsigned tab_a = lds0 + kOffTab + lhi * 16 + nt * 128;        // folded-BN table, conv1 / conv2: float index X + g * 8 + lhi * 4 = base + offset
  // The two groups need different constants: ONE set of registers, named per group below (both sets live = 23 spilled registers)
  unsigned lc0, lc1, lc2, lc3, lc4, lc5, lc6, lc7, lc8;
  const int xpitch = p.x_cstride * 2, ypitch = p.y_cstride * 2;
  if (grp == 0) {
    // conv1: B fragment of k-step s from the x step buffer: pixel row ptile (512 B), chunk (2 s + lhi) ^ (pixel & 15)
    lc0 = (unsigned)(ptile * 512 + ((lhi ^ (l31 & 15)) << 4));                      // xrd_off, ^ (s << 5)
    // 128-byte rows (T2 tile / T1 slot r): chunk c of row r sits at (c ^ ((r >> 1) & 7)) << 4
    lc1 = (unsigned)(ptile * 128 + ((lhi ^ ((ptile >> 1) & 7)) << 4));              // t2rd_off, ^ (kk << 5)
    // staging tile (wave-private, 32 pixels x 256 B): chunk c of pixel r at (c ^ (r & 15)) << 4
    lc2 = (lds0 + kOffStg + gw * 8192 + l31 * 256 + lhi * 8) ^ (unsigned)((l31 & 15) << 4);   // stg_acc, ^ (chunk << 4)
    lc3 = lds0 + kOffStg + gw * 8192 + lane * 16;                                   // stg_row, + t * 1024: pixel 4 t + lane / 16
    // piece t of the tile: pixel 4 t + rp, 16-byte position lane & 15 = chunk (lane & 15) ^ (pixel & 15): byte offset from the tile's
    // first pixel = 4 t * pitch + rp * pitch + (c16 ^ ((4 t & 15) << 4))
    lc4 = (unsigned)(lane >> 4);                                                    // rp
    lc5 = (unsigned)((lane >> 4) * xpitch);                                         // rx
    lc6 = (unsigned)((lane >> 4) * ypitch);                                         // ry
    lc7 = (unsigned)(((lane & 15) ^ (lane >> 4)) << 4);                             // c16
    lc8 = lds0 + kOffTab + lhi * 16 + nt * 512;                                     // tab_c: conv3, channels nt * 128 ..
  } else {
    lc0 = (unsigned)(ptile * 128 + lhi * 8) ^ (unsigned)((((ptile >> 1) & 7)) << 4) ^ (unsigned)(nt << 6);   // t2wr_off, ^ (g << 4)
    // x DMA lane: piece t * 4 + gw = pixels 2 piece, 2 piece + 1; lane (lhi, l31) = 16-byte position l31 of pixel pp = 8 t + xpp, source
    // chunk l31 ^ (pp & 15): byte offset from the step's first pixel = 8 t * pitch + xpp * pitch + (xc16 ^ ((t & 1) << 7))
    lc1 = (unsigned)(2 * gw + lhi);                                                 // xpp
    lc2 = (unsigned)((2 * gw + lhi) * xpitch);                                      // xpx
    lc3 = (unsigned)((l31 ^ lhi ^ (2 * gw)) << 4);                                  // xc16
    lc4 = (unsigned)(ptile % W);                                                    // col: the lane's column (x border of the 3x3 taps)
    lc5 = lc6 = lc7 = lc8 = 0u;
  }
  unsigned &xrd_off = lc0, &t2rd_off = lc1, &stg_acc = lc2, &stg_row = lc3, &rp = lc4, &rx = lc5, &ry = lc6, &c16 = lc7, &tab_c = lc8;
  unsigned &t2wr_off = lc0, &xpp = lc1, &xpx = lc2, &xc16 = lc3, &col = lc4;
  const unsigned dcol = (unsigned)(64 % W);
  const unsigned zaddr = lds0 + kOffZero;

  // dev (FT_BNR_DBG & 32): per-section s_memtime sums of this wave. G0: 0 conv3 + tile out, 1 residual issue, 2 conv1, 3 barrier;
  // G1: 0 x issue, 1 conv2, 2 wait for x, 3 barrier
  unsigned long long tph[4] = {0, 0, 0, 0}, tprev = 0, tc0 = 0;
#define BNR_TS(k) do { if (p.dbg & 32) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tph[k] += t_ - tprev; tprev = t_; } } while (0)
  if (p.dbg & 32) tc0 = tprev = __builtin_amdgcn_s_memtime();

  for (int i = jlo - 3; i <= NS; ++i) {
    asm volatile("" : "+v"(ptile), "+v"(tab_a), "+v"(lc0), "+v"(lc1), "+v"(lc2), "+v"(lc3), "+v"(lc4), "+v"(lc5), "+v"(lc6), "+v"(lc7), "+v"(lc8));
    if (grp == 0) {
      // ================= G0: conv3 of step i - 1 ========================================================================
      const int ic = i - 1;
      if (ic >= 0 && ic < NS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's residual piece of step ic has landed
        const unsigned t2b = lds0 + kOffT2 + (ic & 1) * 8192;
        uint4_t fb[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) rd128(fb[kk], t2b + (t2rd_off ^ (unsigned)(kk << 5)));
        wait4<0>(fb);
        // channel tile mt + 1 is multiplied while tile mt's epilogue runs on the vector ALU
        float16_t acc[2];
        auto mul = [&](auto mc) {
          constexpr int mt = decltype(mc)::value;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt & 1][r] = 0.f;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            acc[mt & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, wt[16 + mt * 4 + kk]), __builtin_bit_cast(half8_t, fb[kk]),
                                                                 acc[mt & 1], 0, 0, 0);
        };
        mul(std::integral_constant<int, 0>{});
        static_for<4>([&](auto mc) {
          constexpr int mt = decltype(mc)::value;
          float4_t sc[4], sh[4];
          uint2_t rs[4];
          static_for<4>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            rd128fo<1024 + mt * 128 + g * 32>(sc[g], tab_c);
            rd128fo<2048 + mt * 128 + g * 32>(sh[g], tab_c);
            rd64(rs[g], stg_acc ^ (unsigned)((mt * 4 + g) << 4));
          });
          if constexpr (mt + 1 < 4) mul(std::integral_constant<int, mt + 1>{});
          wait_res(sc, sh, rs);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const half4_t res = __builtin_bit_cast(half4_t, rs[g]);
            half4_t hv;
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
              const float2_t a = {acc[mt & 1][g * 4 + e2 * 2], acc[mt & 1][g * 4 + e2 * 2 + 1]};
              const float2_t k = {sc[g][e2 * 2], sc[g][e2 * 2 + 1]}, b = {sh[g][e2 * 2], sh[g][e2 * 2 + 1]};
              const half2_t r2 = {res[e2 * 2], res[e2 * 2 + 1]};
              const half2_t o = relu2(__builtin_elementwise_fma(a, k, b) + __builtin_convertvector(r2, float2_t));
              hv[e2 * 2] = o[0];
              hv[e2 * 2 + 1] = o[1];
            }
            wr64(stg_acc ^ (unsigned)((mt * 4 + g) << 4), __builtin_bit_cast(uint2_t, hv));
          }
        });
        // the finished 32 x 128-channel tile leaves as whole 16-byte pieces (LDS operations of one wave execute in order)
        uint4_t rv[8];
        static_for<8>([&](auto tc) { rd128o<decltype(tc)::value * 1024>(rv[decltype(tc)::value], stg_row); });
        wait8<0>(rv);
        const int qb = ic * 64 + pt * 32;
        const unsigned yb = (unsigned)(((imgbase + r0W + qb) * p.y_cstride + p.y_coff + nt * 128) * 2);
        const int lim = (p.dbg & 4) ? 0 : npx - qb;                // pixels of the tile that exist
#pragma unroll
        for (int t = 0; t < 8; ++t)
          __builtin_amdgcn_raw_buffer_store_b128(rv[t], rsrc_y, t * 4 + (int)rp < lim ? yb + (unsigned)(t * 4 * ypitch) + ry + (c16 ^ (unsigned)((t & 3) << 6)) : kOOB,
                                                 0, FT_YSTORE_BUF_AUX);
      }
      BNR_TS(0);
      // ================= G0: residual piece of step i (x again, an L2 hit) into the wave's tile ===============================
      if (i >= 0 && i < NS) {
        const int qb = i * 64 + pt * 32;
        const unsigned xb = (unsigned)(((imgbase + r0W + qb) * p.x_cstride + p.x_coff + nt * 128) * 2);
        const int lim = (p.dbg & 1) ? 0 : npx - qb;
#pragma unroll
        for (int t = 0; t < 8; ++t)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr)(smem + kOffStg + gw * 8192 + t * 1024), 16,
                                                   t * 4 + (int)rp < lim ? xb + (unsigned)(t * 4 * xpitch) + rx + (c16 ^ (unsigned)((t & 3) << 6)) : kOOB, 0, 0, 0);
      }
      BNR_TS(1);
      // ================= G0: conv1 of step i + 2 -> T1 ring =================================================================
      const int j = i + 2;
      if (j >= jlo && j <= jhi) {
        const unsigned xbuf = lds0 + kOffX + (j & 1) * kXB;
        uint4_t f0[8], f1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) rd128(f0[k], xbuf + (xrd_off ^ (unsigned)(k << 5)));
#pragma unroll
        for (int k = 0; k < 8; ++k) rd128(f1[k], xbuf + (xrd_off ^ (unsigned)((8 + k) << 5)));
        float16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        wait8<8>(f0);
#pragma unroll
        for (int k = 0; k < 8; ++k)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, wt[k]), __builtin_bit_cast(half8_t, f0[k]), acc, 0, 0, 0);
        float4_t sc[4], sh[4];
        static_for<4>([&](auto gc) {
          constexpr int g = decltype(gc)::value;
          rd128fo<g * 32>(sc[g], tab_a);
          rd128fo<256 + g * 32>(sh[g], tab_a);
        });
        wait8<8>(f1);
#pragma unroll
        for (int k = 0; k < 8; ++k)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, wt[8 + k]), __builtin_bit_cast(half8_t, f1[k]), acc, 0, 0, 0);
        wait_tab(sc, sh);
        const int q = j * 64 + ptile;
        const bool inside = (unsigned)(r0W + q) < (unsigned)HW;     // out-of-image rows are conv2's zero padding, not relu(bn1(0))
        const int slot = q & 255;
        const unsigned t1w = (lds0 + kOffT1 + (unsigned)(slot * 128 + lhi * 8)) ^ (unsigned)(((slot >> 1) & 7) << 4) ^ (unsigned)(nt << 6);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2_t hb = epi4(acc, sc[g], sh[g], g);
          hb.x = inside ? hb.x : 0u;
          hb.y = inside ? hb.y : 0u;
          wr64(t1w ^ (unsigned)(g << 4), hb);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      BNR_TS(2);
    } else {
      // ================= G1: x of step i + 3 into the ring, L2 touch of step i + 6 ===========================================
      const int jx = i + 3;
      if (jx >= jlo && jx <= jhi) {
        // pixel pp of the step is wanted iff lo <= pp <= hi: inside the image and inside conv1's range of the strip
        const int fb0 = r0W + jx * 64;
        int lo = -fb0, hi = HW - 1 - fb0;
        lo = lo > -(W + 1) - jx * 64 ? lo : -(W + 1) - jx * 64;
        hi = hi < npx + W - jx * 64 ? hi : npx + W - jx * 64;
        if (p.dbg & 1) hi = lo - 1;
        const unsigned span = (unsigned)(hi - lo);                 // (hi < lo: nothing)
        const unsigned xb = (unsigned)(((imgbase + fb0) * p.x_cstride + p.x_coff) * 2);
#pragma unroll
        for (int t = 0; t < 8; ++t)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr)(smem + kOffX + (jx & 1) * kXB + (t * 4 + gw) * 1024), 16,
                                                   (hi >= lo && (unsigned)(t * 8 + (int)xpp - lo) <= span) ? xb + (unsigned)(t * 8 * xpitch) + xpx + (xc16 ^ (unsigned)((t & 1) << 7)) : kOOB,
                                                   0, 0, 0);
      }
      const int jt = i + 6;
      if (jt <= jhi && !(p.dbg & 8)) {
        const int idx = gw * 64 + lane;                            // line idx & 3 of pixel idx >> 2
        const int q = jt * 64 + (idx >> 2);
        const int f = r0W + q;
        const bool ok = (unsigned)f < (unsigned)HW && q <= npx + W && !(p.dbg & 1);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr)(smem + kOffScr + gw * 256), 4,
                                                 ok ? (unsigned)(((imgbase + f) * p.x_cstride + p.x_coff) * 2 + (idx & 3) * 128) : kOOB, 0, 0, 0);
      }
      BNR_TS(0);
      // ================= G1: conv2 of step i from the T1 ring -> T2 =========================================================
      if (i >= 0 && i < NS) {
        const int q = i * 64 + ptile;
        const unsigned t1b = lds0 + kOffT1;
        uint4_t fr[3][4];                      // fragments of taps t, t + 1, t + 2: two taps (256 MFMA cycles) of read-ahead
        auto rd_tap = [&](auto tc, uint4_t (&f)[4]) {
          constexpr int tap = decltype(tc)::value, dy = tap / 3 - 1, dx = tap % 3 - 1;
          const int slot = (q + dy * W + dx) & 255;
          const unsigned rel = (unsigned)(slot * 128 + ((lhi ^ ((slot >> 1) & 7)) << 4));
          bool edge = false;
          if constexpr (dx == -1) edge = col == 0u;
          if constexpr (dx == 1) edge = col == (unsigned)(W - 1);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) rd128(f[kk], edge ? zaddr : t1b + (rel ^ (unsigned)(kk << 5)));
        };
        float16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        rd_tap(std::integral_constant<int, 0>{}, fr[0]);
        rd_tap(std::integral_constant<int, 1>{}, fr[1]);
        static_for<9>([&](auto tc) {
          constexpr int tap = decltype(tc)::value;
          if constexpr (tap + 2 < 9) rd_tap(std::integral_constant<int, tap + 2>{}, fr[(tap + 2) % 3]);
          wait4<(8 - tap < 2 ? 8 - tap : 2) * 4>(fr[tap % 3]);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, wt[tap * 4 + kk]), __builtin_bit_cast(half8_t, fr[tap % 3][kk]),
                                                         acc, 0, 0, 0);
        });
        float4_t sc[4], sh[4];
        static_for<4>([&](auto gc) {
          constexpr int g = decltype(gc)::value;
          rd128fo<512 + g * 32>(sc[g], tab_a);
          rd128fo<768 + g * 32>(sh[g], tab_a);
        });
        wait_tab(sc, sh);
        const unsigned t2w = lds0 + kOffT2 + (i & 1) * 8192 + t2wr_off;
#pragma unroll
        for (int g = 0; g < 4; ++g) wr64(t2w ^ (unsigned)(g << 4), epi4(acc, sc[g], sh[g], g));
        col += dcol;
        col = col >= (unsigned)W ? col - (unsigned)W : col;
      }
      BNR_TS(1);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // x of step i + 3 has landed (this wave's share), T2 is written
      BNR_TS(2);
    }
    BNR_BARRIER();
    BNR_TS(3);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (p.dbg & 32) {      // dev: lifetime and x-wait cycles of every wave (the output is garbage then)
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.y) + (blockIdx.x * 8 + wave) * 8;
      o[0] = t1 - tc0;
      for (int k = 0; k < 4; ++k) o[1 + k] = tph[k];
      o[5] = tc0;
    }
  }
#endif
}

}  // namespace

int bnr_plan(const ft_bottleneck_desc* d, BnrPlan* out) {
  if (!d || !out) return FT_ERR_INVALID_ARG;
  // FT_BNK_RSTAT (read per call: dev / tests): 0 = off, 1 = where the cost rule below takes it (default), 2 = wherever the shape fits;
  // FT_BNR_SR = rows per strip
  const int mode = getenv("FT_BNK_RSTAT") ? atoi(getenv("FT_BNK_RSTAT")) : 1;
  if (mode <= 0) return FT_ERR_UNSUPPORTED;
  if (d->dtype != FT_F16 || d->C != 256 || d->P != 64 || d->stride > 1 || d->head_only || d->projection) return FT_ERR_UNSUPPORTED;
  if (d->W < 3 || d->W > 62 || d->H < 1 || d->N < 1) return FT_ERR_UNSUPPORTED;       // T1 ring: 2 W + 130 <= 256 pixels
  if ((long long)d->N * d->H * d->W * d->y_cstride * 2 >= (1LL << 31)) return FT_ERR_UNSUPPORTED;
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    static int cached[64] = {};
    if (dev >= 0 && dev < 64) {
      if (!cached[dev]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cached[dev] = v;
        else cached[dev] = 256;
      }
      ncu = cached[dev];
    }
  }
  // strips per image so that N * S fills the CUs once; a strip pays (SR + 2) / SR on conv1 and ~5 iterations of pipeline fill
  int S = ceil_div(ncu, d->N);
  if (S > d->H) S = d->H;
  int SR = ceil_div(d->H, S);
  const int force_sr = getenv("FT_BNR_SR") ? atoi(getenv("FT_BNR_SR")) : 0;
  if (force_sr > 0) SR = force_sr < d->H ? force_sr : d->H;
  S = ceil_div(d->H, SR);
  if (mode < 2 && SR * d->W < 512) return FT_ERR_UNSUPPORTED;   // fewer than eight steps per strip: the patch kernel is the better form
  out->SR = SR;
  out->S = S;
  return FT_OK;
}

int bnr_launch(const ft_bottleneck_desc* d, const BnrPlan& pl, const void* x, const void* w1, const void* w2, const void* w3,
               const float* scale_shift, void* y, hipStream_t stream) {
  BnrParams p{};
  p.x = static_cast<const char*>(x);
  p.y = static_cast<char*>(y);
  p.w1 = static_cast<const char*>(w1);
  p.w2 = static_cast<const char*>(w2);
  p.w3 = static_cast<const char*>(w3);
  p.tab = scale_shift;
  p.N = d->N; p.H = d->H; p.W = d->W;
  p.SR = pl.SR; p.S = pl.S;
  p.x_cstride = d->x_cstride; p.x_coff = d->x_coff; p.y_cstride = d->y_cstride; p.y_coff = d->y_coff;
  p.x_bytes = (unsigned)((size_t)d->N * d->H * d->W * d->x_cstride * 2);
  p.y_bytes = (unsigned)((size_t)d->N * d->H * d->W * d->y_cstride * 2);
  p.total = d->N * pl.S;
  static const int dbg = getenv("FT_BNR_DBG") ? atoi(getenv("FT_BNR_DBG")) : 0;
  p.dbg = dbg;
  FT_RAISE_LDS(bottleneck_rstat_kernel, kLds);
  hipLaunchKernelGGL(bottleneck_rstat_kernel, dim3(p.total), dim3(512), kLds, stream, p);
  FT_LAUNCH_CHECK("bottleneck_rstat_kernel");
  return FT_OK;
}

}  // namespace ft
