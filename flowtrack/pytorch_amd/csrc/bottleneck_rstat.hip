// bottleneck_rstat_kernel — the REGISTER-STATIONARY, PERSISTENT-STRIP form of the 64-plane identity Bottleneck (fp16):
// conv1 1x1 + bn1 + relu -> conv2 3x3 + bn2 + relu -> conv3 1x1 + bn3 + identity residual + relu in ONE launch
// (reference: Bottleneck.forward, lib/pose/models/blocks.py:105-120; layer1.1 / layer1.2 of resnet.py:29-36).  gfx950 only.
//
// Why a second form beside bottleneck_fused_kernel (bottleneck.hip).  That kernel owns a 128-pixel patch per workgroup: every
// patch pulls its 136 KB of weights through L2 -> LDS again (58 LDS-DMA pieces per wave and patch), computes conv1 on a 1.5x halo
// and has its x loads in flight only during one of its three phases; with no memory traffic at all it still takes 41 us per block at
// batch 64 (FT_BNK_DBG ablations) against an HBM floor of ~34 us (100 MB in, 100 MB out).  What bounds BOTH forms without memory
// is instruction ISSUE: a SIMD issues one instruction per ~4 cycles whatever the number of waves (measured here with per-section
// s_memtime sums against the ISA's instruction counts: 8 cycles per instruction per wave at two waves per SIMD) and of the ~1300
// instructions per 64 pixels and SIMD only 68 are MFMAs.  So this form is built around the instruction count:
//   * ONE 8-wave workgroup per CU owns a STRIP of SR full-width rows of one image (batch 64, 64 x 48 maps: 16 rows, 256 strips)
//     and walks it in steps of 64 consecutive pixels (row-major, flat: a 3x3 tap is a shift by dy * W + dx in a flat ring);
//     conv1's halo is the row above / below the strip, computed once (1.12x at 16 rows), nothing is recomputed between steps.
//   * The weights are loaded ONCE per workgroup and stay in registers as MFMA A operands: the waves are two GROUPS, G0 (waves 0-3)
//     holds W1 and W3 (its wave (nt, pt): channel tile nt of conv1 / channel tiles 4 nt .. 4 nt + 3 of conv3, pixel tile pt), G1
//     (waves 4-7) holds W2 (channel tile nt, pixel tile pt).  One wave of each group per SIMD.
//   * The epilogues are (almost) gone: the BatchNorm SCALE is folded into the fp16 weights (from the fp32 master weights: one
//     rounding, as before), the SHIFT enters through one extra MFMA k-step (A = [hi(shift) lo(shift) 0 ...], B = [1 1 0 ...]: exact to
//     22 bits) that also replaces the accumulator's zero fill, and the RESIDUAL is two more k-steps against an identity fragment
//     (x * 1.0 accumulates exactly).  What is left per result pair is v_cvt_pk_f16_f32 + v_pk_max_f16 (relu(fp16(v)) == fp16(relu(v))).
//     No scale / shift table, no fp16 -> fp32 conversions, no per-value VALU multiply.
//   * Software pipeline, ONE s_barrier per step.  Iteration i:  G0: conv3(step i - 1) from T2 + residual -> y; then the
//     residual fetch of step i; then conv1(step i + 2) from the x ring -> T1 ring.   G1: LDS-DMA of x(step i + 3) into the
//     ring, conv2(step i) from T1 -> T2.  (An L2 touch of step i + 6 — one dword per line, so that the demand load hits L2 —
//     is in the code behind FT_BNR_DBG & 8: 65 us with it, 54 without.)
//   * x ring: two 32-KiB step buffers (64 pixels x 512 B, 16-byte chunks XOR-ed with pixel & 15 on the SOURCE side of the
//     DMA: conflict-free ds_read_b128 fragments).  T1: a 256-pixel flat ring, T2: two 64-pixel tiles, both with 144-byte rows
//     (128 + 16: sixteen consecutive rows land on sixteen distinct 16-byte bank slots, and a fragment's k-step is an instruction
//     offset instead of an XOR).  T1's live span is 2 W + 130 pixels: W <= 62.  The residual does not wait in LDS for three steps:
//     each G0 wave DMAs its own 32-pixel x 128-channel piece of x again (an L2 hit) into a wave-private 8-KiB tile, multiplies it
//     in as B fragments, writes y in place and stores the tile as whole 16-byte pieces (256 contiguous bytes per pixel).
//   * Every in-loop LDS access is inline asm with hand-placed lgkmcnt waits (hipcc puts `s_waitcnt vmcnt(0)` in front of
//     compiler-visible LDS accesses while an LDS-DMA is in flight: conv_wstat.hip).
// Measured (batch 64, 64 x 48; tools/dev/bnk_bench.py, bnr_phases.py): 53.6 us against the patch kernel's 57.5 us on the same box
// (first version with the scale / shift table and the residual on the vector ALU: 71 us); inside the R50 network 54.3 / 57.0 against
// 55.6 / 54.8 us on one box, 60.8 / 60.8 against 59.5 / 57.9 on another (behind the cold L2 a network leaves this form loses 7-9 us,
// the patch form 4-6), so the Python host records it as an ALTERNATIVE form of the block only with FT_STRIP_KERNELS=1.
// With every load and store off it takes 41 us: G0 needs 5.9 k cycles per step, G1 3.9 k (it waits at the barrier for 30 % of its
// life); ~900 instructions per step and SIMD, 82 of them MFMAs.  Tried on top and not kept: wave priority for G0 (no change); one
// predicated instruction stream per group with every MFMA followed by its share of the rest (sched_barrier pins): the groups balance
// at 4.3 k cycles per step but the pipeline's fill and drain iterations then cost full steps: 62 us.
// Weight buffer (ft_bottleneck_rstat_fwd's `wpack`, built by the caller once per weight set; fp16):
//   [64][272]  conv1: w1[co][ci] * scale1[co], then 16 columns {hi(shift1[co]), lo(shift1[co]), 0 x 14}
//   [64][592]  conv2: w2[co][(ky * 3 + kx) * 64 + ci] * scale2[co], then the 16 shift columns
//   [256][80]  conv3: w3[co][ci] * scale3[co], then the 16 shift columns         (hi = fp16(shift), lo = fp16(shift - hi))
#include <stdlib.h>

#include "conv_common.h"

namespace ft {
namespace {

typedef uint32_t uint2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

struct BnrPlan {
  int SR;   // rows per strip
  int S;    // strips per image
};

struct BnrParams {
  const char* x;
  char* y;
  const char* wp;
  int N, H, W;
  int SR, S;         // rows per strip, strips per image
  int x_cstride, x_coff, y_cstride, y_coff;
  unsigned x_bytes, y_bytes;
  int total;
  int dbg;           // FT_BNR_DBG (dev): 1 no x / residual loads, 4 no stores, 8 no L2 touch, 32 per-wave cycle counts
};

constexpr int kK1 = 272, kK2 = 592, kK3 = 80;                       // packed row lengths (halves)
constexpr int kW1Bytes = 64 * kK1 * 2, kW2Bytes = 64 * kK2 * 2, kW3Bytes = 256 * kK3 * 2;
constexpr int kXB = 32768;        // one step of x: 64 pixels x 512 B
constexpr int kRow = 144;         // T1 / T2 row pitch
constexpr int kOffX = 0;          // two step buffers
constexpr int kOffStg = 65536;    // four wave-private 8-KiB residual / output tiles (32 pixels x 256 B)
constexpr int kOffT1 = 98304;     // 256-slot flat ring x 144 B
constexpr int kOffT2 = kOffT1 + 256 * kRow;    // two 64-pixel tiles x 144 B
constexpr int kT2B = 64 * kRow;
constexpr int kOffZero = kOffT2 + 2 * kT2B;    // 128 B of zeros (x-border taps: all four k-steps of a tap at 32-byte offsets)
constexpr int kOffScr = kOffZero + 128;        // 1 KiB scratch of the L2 touch loads
constexpr int kLds = kOffScr + 1024;
static_assert(kLds <= 163840, "LDS");

#define BNR_BARRIER() asm volatile("s_barrier" ::: "memory")

__device__ __forceinline__ void rd128(uint4_t& v, unsigned a) { asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); }
template <int OFF>
__device__ __forceinline__ void rd128o(uint4_t& v, unsigned a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF)); }
template <int OFF>
__device__ __forceinline__ void wr64o(unsigned a, uint2_t v) { asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a), "v"(v), "n"(OFF) : "memory"); }
__device__ __forceinline__ void wr64(unsigned a, uint2_t v) { asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory"); }
// counted LDS waits that carry the registers they release (no MFMA / VALU use can move above them)
template <int N>
__device__ __forceinline__ void wait2(uint4_t (&f)[2]) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f[0]), "+v"(f[1]) : "n"(N)); }
template <int N>
__device__ __forceinline__ void wait4(uint4_t (&f)[4]) { asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(N)); }
template <int N>
__device__ __forceinline__ void wait8(uint4_t (&f)[8]) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "n"(N));
}
__device__ __forceinline__ float16_t mfma(const uint4_t& a, const uint4_t& b, const float16_t& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
}
// accumulator registers 4 g .. 4 g + 3 -> relu as four fp16.  relu(fp16(v)) == fp16(relu(v)) (rounding is monotonic and keeps the
// sign); a negative value that rounds to -0 comes out as a zero of either sign: equal as a number.
__device__ __forceinline__ uint2_t relu4(const float16_t& acc, int g) {
  half4_t hv;
#pragma unroll
  for (int e2 = 0; e2 < 2; ++e2) {
    const float2_t a = {acc[g * 4 + e2 * 2], acc[g * 4 + e2 * 2 + 1]};
    const half2_t h = __builtin_elementwise_max(__builtin_convertvector(a, half2_t), half2_t{(half_t)0.f, (half_t)0.f});
    hv[e2 * 2] = h[0];
    hv[e2 * 2 + 1] = h[1];
  }
  return __builtin_bit_cast(uint2_t, hv);
}

// wt[] slots.  G0: 0..15 W1 k-steps, 16 its shift step; 17 + 5 mt + kk: W3 channel tile mt, k-step kk (kk = 4: shift step);
// 37, 38 the identity halves.  G1: 0..35 W2 k-steps (4 per tap), 36 its shift step.
constexpr int kNW = 39;

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
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.wp), 0, kW1Bytes + kW2Bytes + kW3Bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;

  // ---- this wave's weights, MFMA A fragments (lane (m = l31, lhi): 8 halves k = 16 s + 8 lhi .. of row m) ----------------
  uint4_t wt[kNW];
  if (grp == 0) {
#pragma unroll
    for (int k = 0; k < 17; ++k) wt[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (unsigned)(((nt * 32 + l31) * kK1 + k * 16 + lhi * 8) * 2), 0, 0);
#pragma unroll
    for (int k = 0; k < 20; ++k)
      wt[17 + k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (unsigned)(kW1Bytes + kW2Bytes + (((nt * 4 + k / 5) * 32 + l31) * kK3 + (k % 5) * 16 + lhi * 8) * 2), 0, 0);
    // identity halves: row m, k-step s holds 1.0 at k = m - 16 s
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int e = l31 - s2 * 16 - lhi * 8;                  // position 0..7 inside this lane's 8 halves, if any
      uint4_t id = {0u, 0u, 0u, 0u};
      const unsigned one = (e & 1) ? 0x3C000000u : 0x00003C00u;
      if (e >= 0 && e < 8) {
        id.x = (e >> 1) == 0 ? one : 0u;
        id.y = (e >> 1) == 1 ? one : 0u;
        id.z = (e >> 1) == 2 ? one : 0u;
        id.w = (e >> 1) == 3 ? one : 0u;
      }
      wt[37 + s2] = id;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 37; ++k) wt[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (unsigned)(kW1Bytes + ((nt * 32 + l31) * kK2 + k * 16 + lhi * 8) * 2), 0, 0);
    wt[37] = wt[38] = uint4_t{0u, 0u, 0u, 0u};
  }
  // B fragment of the shift step: k = 0 and 1 are 1.0 for every pixel
  uint4_t ones = {lhi == 0 ? 0x3C003C00u : 0u, 0u, 0u, 0u};
  if (tid < 8) *reinterpret_cast<uint4_t*>(smem + kOffZero + tid * 16) = uint4_t{0u, 0u, 0u, 0u};      // (no LDS-DMA is in flight yet)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  // ... and the compiler has to know the weights are there (it cannot see the wait above and would guard their first in-loop
  // use with a vmcnt(0) of its own: the whole x stream)
#pragma unroll
  for (int k = 0; k < kNW; ++k) asm volatile("" : "+v"(wt[k]));
  BNR_BARRIER();

  // ---- per-lane constants -----------------------------------------------------------------------------------------------
  // Few bases, everything else derived from them by ONE VALU operation or an instruction offset at the point of use; each
  // iteration passes the bases through an empty asm so that hipcc does not hoist the derived addresses out of the loop (it
  // did, and spilled them: 167 scratch registers in the first version).  The two groups need different constants: ONE set of
  // registers, named per group below.
  int ptile = pt * 32 + l31;                                    // this lane's pixel inside a step (B-fragment row)
  unsigned lc0, lc1, lc2, lc3, lc4, lc5, lc6, lc7;
  const int xpitch = p.x_cstride * 2, ypitch = p.y_cstride * 2;
  if (grp == 0) {
    // conv1: B fragment of k-step s from the x step buffer: pixel row ptile (512 B), chunk (2 s + lhi) ^ (pixel & 15)
    lc0 = (unsigned)(ptile * 512 + ((lhi ^ (l31 & 15)) << 4));                      // xrd_off, ^ (s << 5)
    lc1 = (unsigned)(ptile * kRow + lhi * 16);                                      // t2rd_off: T2 row ptile, + 32 kk
    // staging tile (wave-private, 32 pixels x 256 B): chunk c of pixel r at (c ^ (r & 15)) << 4
    lc2 = (lds0 + kOffStg + gw * 8192 + l31 * 256) ^ (unsigned)((l31 & 15) << 4);   // stg_px, ^ (chunk << 4)
    lc3 = lds0 + kOffStg + gw * 8192 + lane * 16;                                   // stg_row, + t * 1024: pixel 4 t + lane / 16
    // piece t of the tile: pixel 4 t + rp, 16-byte position lane & 15 = chunk (lane & 15) ^ (pixel & 15): byte offset from the tile's
    // first pixel = 4 t * pitch + rp * pitch + (c16 ^ ((4 t & 15) << 4))
    lc4 = (unsigned)(lane >> 4);                                                    // rp
    lc5 = (unsigned)((lane >> 4) * xpitch);                                         // rx
    lc6 = (unsigned)((lane >> 4) * ypitch);                                         // ry
    lc7 = (unsigned)(((lane & 15) ^ (lane >> 4)) << 4);                             // c16
  } else {
    lc0 = (unsigned)(ptile * kRow + nt * 64 + lhi * 8);                             // t2wr_off: T2 row ptile, channels nt * 32 + 4 lhi, + 16 g
    // x DMA lane: piece t * 4 + gw = pixels 2 piece, 2 piece + 1; lane (lhi, l31) = 16-byte position l31 of pixel pp = 8 t + xpp, source
    // chunk l31 ^ (pp & 15): byte offset from the step's first pixel = 8 t * pitch + xpp * pitch + (xc16 ^ ((t & 1) << 7))
    lc1 = (unsigned)(2 * gw + lhi);                                                 // xpp
    lc2 = (unsigned)((2 * gw + lhi) * xpitch);                                      // xpx
    lc3 = (unsigned)((l31 ^ lhi ^ (2 * gw)) << 4);                                  // xc16
    lc4 = (unsigned)(ptile % W);                                                    // col: the lane's column (x border of the 3x3 taps)
    lc5 = lc2 + lc3;                                                                // xl0: lane offset of even pieces
    lc6 = lc2 + (lc3 ^ 128u);                                                       // xl1: ... of odd pieces
    lc7 = 0u;
  }
  unsigned &xrd_off = lc0, &t2rd_off = lc1, &stg_px = lc2, &stg_row = lc3, &rp = lc4, &rx = lc5, &ry = lc6, &c16 = lc7;
  unsigned &t2wr_off = lc0, &xpp = lc1, &col = lc4, &xl0 = lc5, &xl1 = lc6;   // (lc2 = xpp * pitch, lc3 = the lane's source chunk * 16)
  const unsigned dcol = (unsigned)(64 % W);
  const unsigned zaddr = lds0 + kOffZero;

  // dev (FT_BNR_DBG & 32): per-section s_memtime sums of this wave. G0: 0 conv3 + tile out, 1 residual issue, 2 conv1, 3 barrier;
  // G1: 0 x issue, 1 conv2, 2 wait for x, 3 barrier
  unsigned long long tph[4] = {0, 0, 0, 0}, tprev = 0, tc0 = 0;
#define BNR_TS(k) do { if (p.dbg & 32) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tph[k] += t_ - tprev; tprev = t_; } } while (0)
  if (p.dbg & 32) tc0 = tprev = __builtin_amdgcn_s_memtime();
  const float16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  for (int i = jlo - 3; i <= NS; ++i) {
    asm volatile("" : "+v"(ptile), "+v"(lc0), "+v"(lc1), "+v"(lc2), "+v"(lc3), "+v"(lc4), "+v"(lc5), "+v"(lc6), "+v"(lc7), "+v"(ones));
    if (grp == 0) {
      // ================= G0: conv3 of step i - 1 ========================================================================
      const int ic = i - 1;
      if (ic >= 0 && ic < NS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's residual piece of step ic has landed
        const unsigned t2b = lds0 + kOffT2 + (ic & 1) * kT2B + t2rd_off;
        uint4_t fb[4], xr[2][2];
        static_for<4>([&](auto kc) { rd128o<decltype(kc)::value * 32>(fb[decltype(kc)::value], t2b); });
        // residual B fragments of channel tile mt: channels 32 mt + 16 s + 8 lhi .. = chunk 4 mt + 2 s + lhi of the lane's pixel
        auto rd_res = [&](auto mc, uint4_t (&f)[2]) {
          constexpr int mt = decltype(mc)::value;
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) rd128(f[s2], stg_px ^ (unsigned)((mt * 4 + s2 * 2 + lhi) << 4));
        };
        rd_res(std::integral_constant<int, 0>{}, xr[0]);
        rd_res(std::integral_constant<int, 1>{}, xr[1]);
        wait4<4>(fb);
        float16_t acc[2];
        // shift step (replaces the zero fill), four k-steps of t2, two identity steps of x.  `behind` = LDS operations issued after
        // this tile's residual reads
        auto mul = [&](auto mc, auto bc) {
          constexpr int mt = decltype(mc)::value, behind = decltype(bc)::value;
          float16_t a = mfma(wt[17 + mt * 5 + 4], ones, zero16);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) a = mfma(wt[17 + mt * 5 + kk], fb[kk], a);
          wait2<behind>(xr[mt & 1]);
          a = mfma(wt[37], xr[mt & 1][0], a);
          a = mfma(wt[38], xr[mt & 1][1], a);
          acc[mt & 1] = a;
        };
        mul(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
        static_for<4>([&](auto mc) {
          constexpr int mt = decltype(mc)::value;
          // tile mt + 1 goes to the matrix pipe before tile mt's results are touched; tile mt + 2's residual reads follow
          if constexpr (mt == 0) mul(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
          if constexpr (mt == 1 || mt == 2) mul(std::integral_constant<int, mt + 1>{}, std::integral_constant<int, 4>{});
          if constexpr (mt + 2 < 4) rd_res(std::integral_constant<int, mt + 2>{}, xr[mt & 1]);
#pragma unroll
          for (int g = 0; g < 4; ++g) wr64((stg_px + lhi * 8) ^ (unsigned)((mt * 4 + g) << 4), relu4(acc[mt & 1], g));
        });
        // the finished 32 x 128-channel tile leaves as whole 16-byte pieces (LDS operations of one wave execute in order)
        uint4_t rv[8];
        static_for<8>([&](auto tc) { rd128o<decltype(tc)::value * 1024>(rv[decltype(tc)::value], stg_row); });
        wait8<0>(rv);
        const int qb = ic * 64 + pt * 32;
        const unsigned yb = (unsigned)(((imgbase + r0W + qb) * p.y_cstride + p.y_coff + nt * 128) * 2);
        const int lim = (p.dbg & 4) ? 0 : npx - qb;                // pixels of the tile that exist
        // (16-byte buffer stores keep the literal 0 as scalar offset: conv_igemm8.hip's hazard note)
        if (lim >= 32) {
#pragma unroll
          for (int t = 0; t < 8; ++t)
            __builtin_amdgcn_raw_buffer_store_b128(rv[t], rsrc_y, (yb + (unsigned)(t * 4 * ypitch)) + ry + (c16 ^ (unsigned)((t & 3) << 6)), 0, FT_YSTORE_BUF_AUX);
        } else {
#pragma unroll
          for (int t = 0; t < 8; ++t)
            __builtin_amdgcn_raw_buffer_store_b128(rv[t], rsrc_y, t * 4 + (int)rp < lim ? (yb + (unsigned)(t * 4 * ypitch)) + ry + (c16 ^ (unsigned)((t & 3) << 6)) : kOOB,
                                                   0, FT_YSTORE_BUF_AUX);
        }
      }
      BNR_TS(0);
      // ================= G0: residual piece of step i (x again, an L2 hit) into the wave's tile ===============================
      if (i >= 0 && i < NS) {
        const int qb = i * 64 + pt * 32;
        const unsigned xb = (unsigned)(((imgbase + r0W + qb) * p.x_cstride + p.x_coff + nt * 128) * 2);
        const int lim = (p.dbg & 1) ? 0 : npx - qb;
        const int ldsb = kOffStg + gw * 8192;
        if (lim >= 32) {
#pragma unroll
          for (int t = 0; t < 8; ++t)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr)(smem + ldsb + t * 1024), 16, rx + (c16 ^ (unsigned)((t & 3) << 6)),
                                                     xb + (unsigned)(t * 4 * xpitch), 0, 0);
        } else {
#pragma unroll
          for (int t = 0; t < 8; ++t)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr)(smem + ldsb + t * 1024), 16,
                                                     t * 4 + (int)rp < lim ? rx + (c16 ^ (unsigned)((t & 3) << 6)) : kOOB, xb + (unsigned)(t * 4 * xpitch), 0, 0);
        }
      }
      BNR_TS(1);
      // ================= G0: conv1 of step i + 2 -> T1 ring =================================================================
      const int j = i + 2;
      if (j >= jlo && j <= jhi) {
        const unsigned xbuf = lds0 + kOffX + (j & 1) * kXB;
        uint4_t fr[3][4];                      // four k-steps per group, two groups (256 MFMA cycles) of read-ahead
        auto rd_grp = [&](auto gc, uint4_t (&f)[4]) {
          constexpr int g4 = decltype(gc)::value;
#pragma unroll
          for (int k = 0; k < 4; ++k) rd128(f[k], xbuf + (xrd_off ^ (unsigned)((g4 * 4 + k) << 5)));
        };
        rd_grp(std::integral_constant<int, 0>{}, fr[0]);
        rd_grp(std::integral_constant<int, 1>{}, fr[1]);
        float16_t acc = mfma(wt[16], ones, zero16);
        static_for<4>([&](auto gc) {
          constexpr int g4 = decltype(gc)::value;
          if constexpr (g4 + 2 < 4) rd_grp(std::integral_constant<int, g4 + 2>{}, fr[(g4 + 2) % 3]);
          wait4<(3 - g4 < 2 ? 3 - g4 : 2) * 4>(fr[g4 % 3]);
#pragma unroll
          for (int k = 0; k < 4; ++k) acc = mfma(wt[g4 * 4 + k], fr[g4 % 3][k], acc);
        });
        const int q = j * 64 + ptile;
        const bool inside = (unsigned)(r0W + q) < (unsigned)HW;     // out-of-image rows are conv2's zero padding, not relu(bn1(0))
        const unsigned t1w = lds0 + kOffT1 + (unsigned)((q & 255) * kRow + nt * 64 + lhi * 8);
        static_for<4>([&](auto gc) {
          constexpr int g = decltype(gc)::value;
          uint2_t hb = relu4(acc, g);
          hb.x = inside ? hb.x : 0u;
          hb.y = inside ? hb.y : 0u;
          wr64o<g * 16>(t1w, hb);
        });
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
        // the uniform part of the address rides in the instruction's scalar offset (not range-checked: an unwanted lane's vector
        // offset alone is out of range), the lane part is one of two constants: no vector ALU work per piece on whole steps
        const unsigned xb = (unsigned)(((imgbase + fb0) * p.x_cstride + p.x_coff) * 2);
        const int ldsb = kOffX + (jx & 1) * kXB + gw * 1024;
        if (lo <= 0 && hi >= 63) {
#pragma unroll
          for (int t = 0; t < 8; ++t)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr)(smem + ldsb + t * 4096), 16, (t & 1) ? xl1 : xl0, xb + (unsigned)(t * 8 * xpitch), 0, 0);
        } else {
          // (ragged steps: the step's first pixel may lie before the image, i.e. a NEGATIVE base: everything in the 32-bit vector offset)
#pragma unroll
          for (int t = 0; t < 8; ++t)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr)(smem + ldsb + t * 4096), 16,
                                                     (hi >= lo && (unsigned)(t * 8 + (int)xpp - lo) <= span) ? xb + (unsigned)(t * 8 * xpitch) + ((t & 1) ? xl1 : xl0) : kOOB,
                                                     0, 0, 0);
        }
      }
      const int jt = i + ((p.dbg >> 8) & 7 ? ((p.dbg >> 8) & 7) : 6);     // (dev: FT_BNR_DBG bits 8..10 = the touch's lead over the step the loop is in)
      if (jt <= jhi && (p.dbg & 8)) {        // (off: measured 65 us with the touch, 54 without)
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
        const unsigned t1b = lds0 + kOffT1 + lhi * 16;
        uint4_t fr[3][4];                      // fragments of taps t, t + 1, t + 2: two taps (256 MFMA cycles) of read-ahead
        auto rd_tap = [&](auto tc, uint4_t (&f)[4]) {
          constexpr int tap = decltype(tc)::value, dy = tap / 3 - 1, dx = tap % 3 - 1;
          unsigned a = t1b + (unsigned)(((q + dy * W + dx) & 255) * kRow);
          if constexpr (dx == -1) a = col == 0u ? zaddr : a;           // (the 128 zero bytes serve all four k-steps)
          if constexpr (dx == 1) a = col == (unsigned)(W - 1) ? zaddr : a;
          static_for<4>([&](auto kc) { rd128o<decltype(kc)::value * 32>(f[decltype(kc)::value], a); });
        };
        rd_tap(std::integral_constant<int, 0>{}, fr[0]);
        rd_tap(std::integral_constant<int, 1>{}, fr[1]);
        float16_t acc = mfma(wt[36], ones, zero16);
        static_for<9>([&](auto tc) {
          constexpr int tap = decltype(tc)::value;
          if constexpr (tap + 2 < 9) rd_tap(std::integral_constant<int, tap + 2>{}, fr[(tap + 2) % 3]);
          wait4<(8 - tap < 2 ? 8 - tap : 2) * 4>(fr[tap % 3]);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) acc = mfma(wt[tap * 4 + kk], fr[tap % 3][kk], acc);
        });
        const unsigned t2w = lds0 + kOffT2 + (i & 1) * kT2B + t2wr_off;
        static_for<4>([&](auto gc) { wr64o<decltype(gc)::value * 16>(t2w, relu4(acc, decltype(gc)::value)); });
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
  if (p.dbg & 32) {      // dev: lifetime and per-section cycles of every wave (the output is garbage then)
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

static int bnr_plan(const ft_bottleneck_desc* d, BnrPlan* out) {
  if (!d || !out) return FT_ERR_INVALID_ARG;
  // FT_BNK_RSTAT (read per call: dev / tests): 0 = off, 1 = where the cost rule below takes it (default), 2 = wherever the shape fits;
  // FT_BNR_SR = rows per strip
  const int mode = getenv("FT_BNK_RSTAT") ? atoi(getenv("FT_BNK_RSTAT")) : 1;
  if (mode <= 0) return FT_ERR_UNSUPPORTED;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0) return FT_ERR_INVALID_ARG;
  if (d->dtype != FT_F16 || d->C != 256 || d->P != 64 || d->stride > 1 || d->head_only || d->projection) return FT_ERR_UNSUPPORTED;
  if (d->W < 3 || d->W > 62) return FT_ERR_UNSUPPORTED;       // T1 ring: 2 W + 130 <= 256 pixels
  if (d->x_coff < 0 || d->y_coff < 0 || d->x_coff % 8 || d->y_coff % 8 || d->x_cstride % 8 || d->y_cstride % 8) return FT_ERR_UNSUPPORTED;
  if (d->x_cstride < d->x_coff + d->C || d->y_cstride < d->y_coff + d->C) return FT_ERR_INVALID_ARG;
  if ((long long)d->N * d->H * d->W * d->x_cstride * 2 >= (1LL << 31) || (long long)d->N * d->H * d->W * d->y_cstride * 2 >= (1LL << 31)) return FT_ERR_UNSUPPORTED;
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

}  // namespace
}  // namespace ft

extern "C" int ft_bottleneck_rstat_supported(const ft_bottleneck_desc* d) {
  ft::BnrPlan pl;
  return ft::bnr_plan(d, &pl);
}

extern "C" long long ft_bottleneck_rstat_weight_bytes(void) { return ft::kW1Bytes + ft::kW2Bytes + ft::kW3Bytes; }

extern "C" int ft_bottleneck_rstat_fwd(const ft_bottleneck_desc* d, const void* x, const void* wpack, void* y, ft_stream_t stream) {
  using namespace ft;
  BnrPlan pl;
  const int st = bnr_plan(d, &pl);
  if (st != FT_OK) return st;
  if (!x || !wpack || !y) return FT_ERR_INVALID_ARG;
  BnrParams p{};
  p.x = static_cast<const char*>(x);
  p.y = static_cast<char*>(y);
  p.wp = static_cast<const char*>(wpack);
  p.N = d->N; p.H = d->H; p.W = d->W;
  p.SR = pl.SR; p.S = pl.S;
  p.x_cstride = d->x_cstride; p.x_coff = d->x_coff; p.y_cstride = d->y_cstride; p.y_coff = d->y_coff;
  p.x_bytes = (unsigned)((size_t)d->N * d->H * d->W * d->x_cstride * 2);
  p.y_bytes = (unsigned)((size_t)d->N * d->H * d->W * d->y_cstride * 2);
  p.total = d->N * pl.S;
  static const int dbg = getenv("FT_BNR_DBG") ? atoi(getenv("FT_BNR_DBG")) : 0;
  p.dbg = dbg;
  FT_RAISE_LDS(bottleneck_rstat_kernel, kLds);
  hipLaunchKernelGGL(bottleneck_rstat_kernel, dim3(p.total), dim3(512), kLds, as_stream(stream), p);
  FT_LAUNCH_CHECK("bottleneck_rstat_kernel");
  return FT_OK;
}
