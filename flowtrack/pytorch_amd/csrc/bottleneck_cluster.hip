// Whole-bottleneck fusion for the 256-plane ResNet stage on SMALL maps (fp16), CLUSTER form (round 5): conv1 1x1 + bn1 + relu
// -> conv2 3x3 + bn2 + relu -> conv3 1x1 + bn3 + identity residual + relu in ONE launch, the work of ONE IMAGE shared by a
// cluster of FOUR workgroups (four CUs of one XCD) that exchange t1 and t2 through global memory inside the launch
// (reference: Bottleneck.forward, lib/pose/models/blocks.py:105-120; layer3.1-5 of ResNet-50, layer3.1-22 of ResNet-101,
// resnet.py:29-36,51-55).
//
// Why: bottleneck_stream_direct_kernel gives a workgroup a 48-pixel strip and streams the block's WHOLE weight set (2.2 MB)
// through every CU: 600 MB of L2 -> CU traffic per block at batch 64, which is what its 44 us are (31 of the 32 B/clk/CU an
// XCD's L2 delivers when all 32 CUs stream), and 48-pixel strips pad to two 32-pixel MFMA tiles and recompute conv1 on their
// halo rows (executed MFMA work 1.5x the algorithmic).  Here a 16 x 12 map is 192 pixels = six FULL tiles, nothing is
// recomputed, and a CU streams about half the bytes:
//   phase 1  PIXEL split: member m computes t1 = relu(bn1(W1 . x)) for ITS quarter of the rows (48 pixels, all 256 channels;
//            the whole W1, 512 KB, straight to registers; its x rows register-staged into two LDS buffers) and publishes them.
//   exchange every member gathers the image's whole t1 (96 KB) into LDS (zero halo row above and below).
//   phase 2  CHANNEL split: member m computes t2 for output channels [64 m, 64 m + 64) on ALL 192 pixels: W2's quarter only
//            (295 KB); wave = (channel tile, K half), the two K halves meet in LDS; publishes its channel slice.
//   exchange every member gathers the whole t2.
//   phase 3  CHANNEL split: output channels [256 m, 256 m + 256): W3's quarter (128 KB, held in registers for both pixel
//            passes), residual straight from x, y leaves from the accumulator layout (32 contiguous bytes per lane).
// Per CU: 0.94 MB of weights + 0.1 MB of x + 0.19 MB of exchange + 0.1 MB of residual = 1.3 MB instead of 2.35 MB.
//
// The hand-off is the placement-independent one of the hardware guide (cdna_hip_programming.md Guideline 16, form R1):
// payload stored WRITE-THROUGH (sc1) as 16-byte pieces, every storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier,
// ONE lane adds 1 to the cluster's counter (agent-scope atomic), ONE lane polls it with relaxed agent-scope loads (+ s_sleep),
// workgroup barrier, then the payload is read with sc1 LDS-DMA loads (they bypass this CU's L1, the only cache another CU's
// stores never refresh).  Members of a cluster are blocks b, b + 8, b + 16, b + 24 (same XCD under the observed b % 8
// placement: speed only, never correctness).  The counter is monotonic across launches: a launch adds exactly 8 per cluster
// (every member arrives twice, unconditionally), so a member derives its launch's base as (value at start) & ~7 — no per-call
// memset node, nothing frozen under graph replay.  Every spin is bounded; a timeout sets the workspace's status word and the
// member goes on (wrong output, no hang).  Needs the four members of a cluster co-resident: the grid is one workgroup per CU
// (126 KB of LDS) and blocks are dispatched in order per XCD, so at most one cluster per XCD and queue is ever partially
// resident while the complete ones ahead of it finish.
#include <stdlib.h>

#include <type_traits>

#include "ft_common.h"

namespace ft {
namespace {

struct BncParams {
  const char* x;
  char* y;
  const char* ws;    // packed weight stream of ft_bottleneck_stream_pack (P = 256)
  const char* tab;   // float [6][2P]: {s1 b1} {s2 b2} {s3 b3 of quarter 0} .. {quarter 3}
  char* t1x;         // exchange buffers [N][HW][256] fp16
  char* t2x;
  unsigned* cnt;     // per cluster: one counter per 64 bytes
  unsigned* status;  // != 0: a spin timed out
  int N, H, W, HW;
  int rows_m;        // rows of a member in phase 1
  int x_cstride, x_coff, y_cstride, y_coff;
  unsigned x_bytes, y_bytes, ws_bytes, tx_bytes;
  int dbg;           // FT_BNC_DBG (dev): 1 no cluster waits (wrong results: timing of the phases alone), 32 phase timestamps
};

template <int N, int I = 0, typename F>
__device__ __forceinline__ void bnc_unroll(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    bnc_unroll<N, I + 1>(f);
  }
}

#define BNC_BARRIER() asm volatile("s_barrier" ::: "memory")
#define BNC_XKEY(hp) (((hp) >> 1) & 7)      // x-chunk rows of 128 bytes: two rows share a 256-byte bank row
#ifndef FT_BNC_PIN
#define FT_BNC_PIN 1    // pinned issue order in phase 1 (as bottleneck_stream_direct_kernel)
#endif
#ifndef FT_BNC_SLOTS1
#define FT_BNC_SLOTS1 4   // weight-step register slots of phase 1 (8 KiB per wave and step)
#endif
#ifndef FT_BNC_SLOTS2
#define FT_BNC_SLOTS2 6   // of phase 2 (2 KiB per wave and step)
#endif

typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int kP = 256, kNCT = 8, kNC1 = 16, kKC = 4;
constexpr int kWSTEP = kNCT * 4096;                 // bytes of a weight step: 4 K16 slices x 8 channel tiles x 1 KiB
constexpr int kG2 = kNC1, kG3 = kG2 + 9 * kKC;      // first weight step of conv2 / conv3 in the stream
constexpr int kROWB = 2 * kP, kTABB = 8 * kP;
constexpr int kMT = 6;                              // pixel tiles of an image (<= 192 pixels)
[[maybe_unused]] constexpr int kXSTRIDE = 8192;     // phase-1 x-chunk buffer: 64 rows x 128 bytes
constexpr int kT1ROWS = 224;                        // T1 rows incl. the zero halo rows (HW + 2 W + 1 <= 224)
constexpr int kZROW = kT1ROWS * kROWB, kTABS = kZROW + 2048, kLdsBytes = kTABS + 6 * kTABB;
static_assert(kLdsBytes <= 163840, "LDS map");

// Cluster barrier: see the header.  Precondition: every wave that stored payload has executed s_waitcnt vmcnt(0).
// `prefetch` (register loads of the next phase's first operands) is issued between the arrival and the end of the wait: by
// waves 1..3 at once, by wave 0 only after its lane 0 has seen the counter (its polls must not queue behind those loads).
template <typename F>
__device__ __forceinline__ void bnc_arrive_wait(unsigned* cnt, unsigned target, unsigned* status, int tid, int wave, bool skip, F&& prefetch) {
  BNC_BARRIER();
  if (tid == 0) __hip_atomic_fetch_add((gu32*)(uintptr_t)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (wave != 0) prefetch();
  if (tid == 0 && !skip) {
    gu32* c = (gu32*)(uintptr_t)cnt;
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 19)) {
        __hip_atomic_store((gu32*)(uintptr_t)status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  if (wave == 0) prefetch();
  BNC_BARRIER();
}

__global__ __launch_bounds__(256, 1) void bottleneck_cluster_kernel(const BncParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  using c0 = std::integral_constant<int, 0>;
  using c1 = std::integral_constant<int, 1>;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  // block -> (image n, member m): the members of a cluster are four consecutive blocks of one XCD's dispatch order
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int m = loc & 3;
  const int n = (loc >> 2) * 8 + xcd;
  if (n >= p.N) return;
  const int W = p.W, HW = p.HW;

  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.ws), 0, p.ws_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.tab), 0, 6 * kTABB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_t1 = __builtin_amdgcn_make_buffer_rsrc(p.t1x, 0, p.tx_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_t2 = __builtin_amdgcn_make_buffer_rsrc(p.t2x, 0, p.tx_bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;
  const unsigned lane16 = (unsigned)lane * 16u;

  unsigned* cnt = p.cnt + (size_t)n * 16;
  unsigned base = 0;
  if (tid == 0) base = __hip_atomic_load((gu32*)(uintptr_t)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ~7u;
  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define BNC_TS(i) do { if (p.dbg & 32) ts[i] = __builtin_amdgcn_s_memtime(); } while (0)
  BNC_TS(0);

  // ================= phase 1: t1 rows of this member, all 256 channels ====================================================
  // wave -> output-channel tiles 2 wave, 2 wave + 1 (of 8) x two 32-pixel tiles; x chunk c = channels [64 c, 64 c + 64)
  const int r0 = m * p.rows_m;
  const int np1 = (p.H - r0 < p.rows_m ? (p.H - r0 > 0 ? p.H - r0 : 0) : p.rows_m) * W;   // pixels of this member (<= 64)
  const int hp0 = r0 * W;
  constexpr int LX = 2, NS = FT_BNC_SLOTS1, D = NS - 1;
  static_assert(NS >= 3, "prefetch distance");
  // The x rows go global -> REGISTERS -> LDS (two chunks ahead in registers, one ahead in LDS), not by LDS-DMA.  The first
  // version used the LDS-DMA ring of bottleneck_stream_direct_kernel with its hand-counted `s_waitcnt vmcnt(N)` (x chunk c+1
  // has landed while the weight loads of later steps stay in flight).  Measured with tools/dev/bnc_stress.py (x rows
  // cache-cold: torch kernels stream hundreds of MB between runs): 15 of 16 runs WRONG; with an extra `s_waitcnt vmcnt(2)` in
  // front of the barrier still 12 of 16; with the rows past the member's 48 pixels clamped to a valid address instead of an
  // out-of-range offset 0 of 16; with a full drain + barrier per chunk 0 of 16.  The mechanism was not isolated further (the
  // strip / patch kernels, which issue the same kind of fully out-of-range wave loads for their padding rows, pass the same
  // cold-cache stress: tools/dev/lds_dma_oob_stress.py, 0 of 20 runs each) — so this loop does not count across two kinds of
  // loads at all: every load in it is a register load and every vmcnt wait is hipcc's own.
  unsigned x_goff[LX];
  int x_loff[LX];
#pragma unroll
  for (int t = 0; t < LX; ++t) {
    const int pi = (t * 4 + wave) * 8 + (lane >> 3);
    const int pic = pi < np1 ? hp0 + pi : 0;       // rows past the member's pixels re-read pixel 0 (results never stored)
    x_goff[t] = (unsigned)(((n * HW + pic) * p.x_cstride + p.x_coff) * 2 + ((lane & 7) << 4));
    x_loff[t] = pi * 128 + (((lane & 7) ^ BNC_XKEY(pi)) << 4);
  }
  uint4_t xr[2][LX];
  auto gload_x = [&](auto sc, int c) {
    constexpr int S = decltype(sc)::value;
#pragma unroll
    for (int t = 0; t < LX; ++t) xr[S][t] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, x_goff[t], c * 128, 0);
  };
  auto swrite_x = [&](auto sc, int buf) {
    constexpr int S = decltype(sc)::value;
#pragma unroll
    for (int t = 0; t < LX; ++t) *reinterpret_cast<uint4_t*>(smem + buf * kXSTRIDE + x_loff[t]) = xr[S][t];
  };
  uint4_t areg[NS][4][2];
  // K16 slices {0, 1} or {2, 3} of step g.  The last D steps of the loop run past conv1's weights into conv2's (valid bytes of
  // the stream, never multiplied)
  auto load_a_half = [&](auto slotc, int g, auto halfc) {
    constexpr int SL = decltype(slotc)::value, HF = decltype(halfc)::value;
#pragma unroll
    for (int kk = 2 * HF; kk < 2 * HF + 2; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        areg[SL][kk][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16, g * kWSTEP + (kk * kNCT + 2 * wave + i) * 1024, 0);
  };
  // prologue: all six tables (12 KiB = 48 pieces of 256 bytes, LDS-DMA: first read behind a full drain at the end of the
  // phase), x chunks 0 and 1, the weights of steps 0 .. D-1
#pragma unroll
  for (int t = 0; t < 12; ++t)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_t, (lds_ptr)(smem + kTABS + (t * 4 + wave) * 256), 4, (unsigned)lane * 4u, (t * 4 + wave) * 256, 0, 0);
  gload_x(c0{}, 0);
  gload_x(c1{}, 1);
  bnc_unroll<D>([&](auto sc) {
    load_a_half(sc, decltype(sc)::value, c0{});
    load_a_half(sc, decltype(sc)::value, c1{});
  });

  {
    float16_t acc1[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[i][j][r] = 0.f;
    int b1_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pi = j * 32 + l31;
      b1_off[j] = pi * 128 + ((lhi ^ BNC_XKEY(pi)) << 4);
    }
    uint4_t fx[2][2];
    auto ldx = [&](auto setc, int buf, int kk) {
      constexpr int S = decltype(setc)::value;
      const char* xb = smem + buf * kXSTRIDE;
#pragma unroll
      for (int j = 0; j < 2; ++j) fx[S][j] = *reinterpret_cast<const uint4_t*>(xb + (b1_off[j] ^ (kk << 5)));
    };
    auto mma1 = [&](auto setc, auto slotc, auto kkc) {
      constexpr int S = decltype(setc)::value, SL = decltype(slotc)::value, kk = decltype(kkc)::value;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, areg[SL][kk][i]),
                                                              __builtin_bit_cast(half8_t, fx[S][j]), acc1[i][j], 0, 0, 0);
    };
    swrite_x(c0{}, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    BNC_BARRIER();
    bnc_unroll<kNC1>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      constexpr int buf = c & 1;
      using slot = std::integral_constant<int, c % NS>;
      if constexpr (c + 2 < kNC1) gload_x(std::integral_constant<int, c & 1>{}, c + 2);   // (its registers went to LDS one chunk ago)
      load_a_half(std::integral_constant<int, (c + D) % NS>{}, c + D, c0{});
      ldx(c0{}, buf, 0);
      ldx(c1{}, buf, 1);
      mma1(c0{}, slot{}, std::integral_constant<int, 0>{});
      ldx(c0{}, buf, 2);
      mma1(c1{}, slot{}, std::integral_constant<int, 1>{});
      load_a_half(std::integral_constant<int, (c + D) % NS>{}, c + D, c1{});
      ldx(c1{}, buf, 3);
      mma1(c0{}, slot{}, std::integral_constant<int, 2>{});
      mma1(c1{}, slot{}, std::integral_constant<int, 3>{});
      if constexpr (c + 1 < kNC1) {
        // chunk c+1 (in registers since the previous chunk) -> the other buffer, whose readers (chunk c-1) are a barrier behind
        swrite_x(std::integral_constant<int, (c + 1) & 1>{}, (c + 1) & 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        BNC_BARRIER();
      }
    });
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the tables' LDS-DMA loads of every wave have landed ...
    BNC_BARRIER();                                                //  ... behind this barrier)
    BNC_TS(1);
    // bn1 + relu -> fp16 -> the exchange buffer, write-through, straight from the accumulator layout (a lane owns 16
    // consecutive channels of its pixel: two 16-byte stores)
    const float* tb = reinterpret_cast<const float*>(smem + kTABS);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ch = (2 * wave + i) * 32 + 16 * lhi;
      float4_t sc[4], sh[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        sc[g4] = *reinterpret_cast<const float4_t*>(tb + ch + g4 * 4);
        sh[g4] = *reinterpret_cast<const float4_t*>(tb + kP + ch + g4 * 4);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int pi = j * 32 + l31;
        half8_t h8[2];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          h8[r >> 3][r & 7] = (half_t)__builtin_fmaxf(acc1[i][j][r] * sc[r >> 2][r & 3] + sh[r >> 2][r & 3], 0.f);
        const unsigned vo = pi < np1 ? (unsigned)(((n * HW + hp0 + pi) * kP + ch) * 2) : kOOB;
#pragma unroll
        for (int h = 0; h < 2; ++h)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, h8[h]), rsrc_t1, vo + (unsigned)(h * 16), 0, 16);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // every storing wave drains; also: past the last x-chunk read
  BNC_BARRIER();
  // zero halo rows of T1 (rows [0, W) and [W + HW, 2 W + HW]) and the shared zero row (the x-border taps read it)
  for (int i = tid; i < (2 * W + 1) * 32 + 32; i += 256) {
    const int r = i >> 5, cpos = i & 31;
    const int row = r < W ? r : (r < 2 * W + 1 ? W + HW + (r - W) : kT1ROWS);
    *reinterpret_cast<uint4_t*>(smem + row * kROWB + cpos * 16) = uint4_t{0u, 0u, 0u, 0u};
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int ct = wave & 1, kh = wave >> 1;
  constexpr int NS2 = FT_BNC_SLOTS2, D2 = NS2 - 1;
  uint4_t a2[NS2][2];
  auto load_a2 = [&](auto slotc, int s) {       // conv2 step s (tap s / 4, K chunk s % 4): this wave's two K16 slices of tile 2 m + ct
    constexpr int SL = decltype(slotc)::value;   // (past the last step: conv3's first steps, valid bytes, never multiplied)
#pragma unroll
    for (int kq = 0; kq < 2; ++kq)
      a2[SL][kq] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16, (kG2 + s) * kWSTEP + ((2 * kh + kq) * kNCT + 2 * m + ct) * 1024, 0);
  };
  int m_out[kMT], edge[kMT];
#pragma unroll
  for (int j = 0; j < kMT; ++j) {
    m_out[j] = j * 32 + l31;
    const int ox = m_out[j] % W;
    edge[j] = (ox == 0 ? 1 : 0) | (ox == W - 1 ? 2 : 0);
  }
  // the residual of phase 3 (this member's 256 channels of x at all pixels, in the accumulator layout: 16 consecutive channels
  // per lane, 96 registers) is fetched HERE, two phases ahead: loaded inside phase 3 its HBM latency was exposed once per
  // pixel pass (phase 3 took 21 k cycles for 6 k cycles of MFMA)
  uint4_t res[2][2][3][2];
  auto load_res = [&]() {
#pragma unroll
    for (int ps = 0; ps < 2; ++ps)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
          const int hp = m_out[3 * ps + jj] < HW ? m_out[3 * ps + jj] : HW - 1;     // (padding pixels: any valid address)
          const unsigned vo = (unsigned)(((n * HW + hp) * p.x_cstride + p.x_coff + m * kP + (2 * wave + i) * 32 + 16 * lhi) * 2);
#pragma unroll
          for (int h = 0; h < 2; ++h) res[ps][i][jj][h] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, vo + (unsigned)(h * 16), 0, 0);
        }
  };
  bnc_arrive_wait(cnt, base + 4u, p.status, tid, wave, p.dbg & 1, [&] {
    bnc_unroll<D2>([&](auto sc) { load_a2(sc, decltype(sc)::value); });    // phase 2's first weight steps
  });
  BNC_TS(2);
  // gather the image's t1: LDS row W + hp <- T1X row hp, 512 bytes each, two rows per wave load; sc1: not through this CU's L1
  {
    const int nload = HW >> 1;                  // (HW is even: every wave load is in range)
    for (int t = wave; t < nload; t += 4) {
      const int hp = 2 * t + (lane >> 5);
      const int row = W + hp;
      const unsigned vo = (unsigned)((n * HW + hp) * kROWB + (((lane & 31) ^ (row & 15)) << 4));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_t1, (lds_ptr)(smem + (W + 2 * t) * kROWB), 16, vo, 0, 0, 16);
    }
  }
  // the gather has landed (a full drain, not a counted wait with register loads in flight behind it: see phase 1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  BNC_BARRIER();
  load_res();
  BNC_TS(3);

  // ================= phase 2: t2 channels [64 m, 64 m + 64) on all pixels ====================================================
  {
    float16_t acc2[kMT];
#pragma unroll
    for (int j = 0; j < kMT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
    bnc_unroll<9 * kKC>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int tap = s / kKC, kc = s % kKC, ky = tap / 3, kx = tap % 3;
      constexpr int bad = kx == 0 ? 1 : (kx == 2 ? 2 : 0);
      using slot = std::integral_constant<int, s % NS2>;
      load_a2(std::integral_constant<int, (s + D2) % NS2>{}, s + D2);
      uint4_t fb[2][kMT];
#pragma unroll
      for (int j = 0; j < kMT; ++j) {
        const int row = m_out[j] + ky * W + kx - 1;          // T1 row of the tap (row 0 = the halo row above the image)
        const int v = (edge[j] & bad) ? kZROW + (lhi << 4) : row * kROWB + (((row & 15) ^ lhi) << 4);
#pragma unroll
        for (int kq = 0; kq < 2; ++kq)
          fb[kq][j] = *reinterpret_cast<const uint4_t*>(smem + (v ^ (kc << 7) ^ ((2 * kh + kq) << 5)));
      }
#pragma unroll
      for (int kq = 0; kq < 2; ++kq)
#pragma unroll
        for (int j = 0; j < kMT; ++j)
          acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, a2[slot::value][kq]),
                                                           __builtin_bit_cast(half8_t, fb[kq][j]), acc2[j], 0, 0, 0);
    });
    BNC_TS(4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    BNC_BARRIER();          // every wave is past its last T1 read: the region becomes the K-half exchange, then T2
    // the two K halves of a channel tile meet in LDS: [ct][tile j][register quad][lane] 16-byte pieces
    if (kh == 1) {
#pragma unroll
      for (int j = 0; j < kMT; ++j)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
          *reinterpret_cast<float4_t*>(smem + ((ct * kMT + j) * 4 + r4) * 1024 + lane * 16) =
              float4_t{acc2[j][4 * r4], acc2[j][4 * r4 + 1], acc2[j][4 * r4 + 2], acc2[j][4 * r4 + 3]};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    BNC_BARRIER();
    if (kh == 0) {
      const float* tb = reinterpret_cast<const float*>(smem + kTABS + kTABB);
      const int ch = (2 * m + ct) * 32 + 16 * lhi;
      float4_t sc[4], sh[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        sc[g4] = *reinterpret_cast<const float4_t*>(tb + ch + g4 * 4);
        sh[g4] = *reinterpret_cast<const float4_t*>(tb + kP + ch + g4 * 4);
      }
#pragma unroll
      for (int j = 0; j < kMT; ++j) {
        half8_t h8[2];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const float4_t o = *reinterpret_cast<const float4_t*>(smem + ((ct * kMT + j) * 4 + r4) * 1024 + lane * 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * r4 + e;
            h8[r >> 3][r & 7] = (half_t)__builtin_fmaxf((acc2[j][r] + o[e]) * sc[r4][e] + sh[r4][e], 0.f);
          }
        }
        const unsigned vo = m_out[j] < HW ? (unsigned)(((n * HW + m_out[j]) * kP + ch) * 2) : kOOB;
#pragma unroll
        for (int h = 0; h < 2; ++h)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, h8[h]), rsrc_t2, vo + (unsigned)(h * 16), 0, 16);
      }
    }
  }
  // phase 3's weights (this wave's two channel tiles of quarter m, all 16 K16 slices: 32 KiB) go to registers while the
  // exchange is in flight
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // every storing wave drains (and: the K-half exchange has been read)
  uint4_t a3[kKC][4][2];
  bnc_arrive_wait(cnt, base + 8u, p.status, tid, wave, p.dbg & 1, [&] {
#pragma unroll
    for (int kc = 0; kc < kKC; ++kc)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          a3[kc][kk][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16, (kG3 + m * kKC + kc) * kWSTEP + (kk * kNCT + 2 * wave + i) * 1024, 0);
  });
  BNC_TS(5);
  {
    const int nload = HW >> 1;
    for (int t = wave; t < nload; t += 4) {
      const int hp = 2 * t + (lane >> 5);
      const unsigned vo = (unsigned)((n * HW + hp) * kROWB + (((lane & 31) ^ (hp & 15)) << 4));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_t2, (lds_ptr)(smem + 2 * t * kROWB), 16, vo, 0, 0, 16);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  BNC_BARRIER();
  BNC_TS(6);

  // ================= phase 3: y channels [256 m, 256 m + 256) = relu(bn3(W3 . t2) + x), two passes of three pixel tiles ======
  {
    const float* tb = reinterpret_cast<const float*>(smem + kTABS + (2 + m) * kTABB);
    bnc_unroll<2>([&](auto pc) {
      constexpr int ps = decltype(pc)::value;
      float16_t acc3[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc3[i][jj][r] = 0.f;
#pragma unroll
      for (int kc = 0; kc < kKC; ++kc)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          uint4_t fb[3];
#pragma unroll
          for (int jj = 0; jj < 3; ++jj) {
            const int row = m_out[3 * ps + jj];
            fb[jj] = *reinterpret_cast<const uint4_t*>(smem + ((row * kROWB + (((row & 15) ^ lhi) << 4)) ^ (kc << 7) ^ (kk << 5)));
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 3; ++jj)
              acc3[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, a3[kc][kk][i]),
                                                                   __builtin_bit_cast(half8_t, fb[jj]), acc3[i][jj], 0, 0, 0);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ch = (2 * wave + i) * 32 + 16 * lhi;
        float4_t sc[4], sh[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          sc[g4] = *reinterpret_cast<const float4_t*>(tb + ch + g4 * 4);
          sh[g4] = *reinterpret_cast<const float4_t*>(tb + kP + ch + g4 * 4);
        }
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
          const int hp = m_out[3 * ps + jj];
          const unsigned vo = hp < HW ? (unsigned)(((n * HW + hp) * p.y_cstride + p.y_coff + m * kP + ch) * 2) : kOOB;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const half8_t rs = __builtin_bit_cast(half8_t, res[ps][i][jj][h]);
            half8_t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int r = h * 8 + e;
              o[e] = (half_t)__builtin_fmaxf(acc3[i][jj][r] * sc[r >> 2][r & 3] + sh[r >> 2][r & 3] + (float)rs[e], 0.f);
            }
            if (!(p.dbg & 4)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, o), rsrc_y, vo + (unsigned)(h * 16), 0, 0);
          }
        }
      }
    });
  }
  if (p.dbg & 32) {
    ts[7] = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.y + ((size_t)(n * HW + m * 8) * p.y_cstride + p.y_coff) * 2);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = ts[i];
    }
  }
#endif
}

struct BncPlan {
  int rows_m, npad;
  long long tx_bytes, cnt_off, status_off, total;
};

static int bnc_plan(const ft_bottleneck_desc* d, BncPlan* out) {
  if (!d) return FT_ERR_INVALID_ARG;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0) return FT_ERR_INVALID_ARG;
  static const bool off = getenv("FT_BNC") && atoi(getenv("FT_BNC")) == 0;
  if (off) return FT_ERR_UNSUPPORTED;
  if (d->dtype != FT_F16 || d->projection || d->head_only || d->P != kP || d->C != 4 * kP || d->stride > 1) return FT_ERR_UNSUPPORTED;
  if (d->x_coff < 0 || d->y_coff < 0 || d->x_coff % 8 || d->y_coff % 8 || d->x_cstride % 8 || d->y_cstride % 8) return FT_ERR_UNSUPPORTED;
  if (d->x_cstride < d->x_coff + d->C || d->y_cstride < d->y_coff + d->C) return FT_ERR_INVALID_ARG;
  const int hw = d->H * d->W;
  const int rows_m = ceil_div(d->H, 4);
  // a whole image = at most six 32-pixel tiles; a member's phase-1 rows at most two; the zero-halo rows fit the T1 region
  if (hw > 32 * kMT || (hw & 1) || rows_m * d->W > 64 || 32 * kMT + 2 * d->W + 1 > kT1ROWS || d->W < 2) return FT_ERR_UNSUPPORTED;
  if ((long long)d->N * hw * d->x_cstride * 2 >= (1LL << 31) || (long long)d->N * hw * d->y_cstride * 2 >= (1LL << 31)) return FT_ERR_UNSUPPORTED;
  BncPlan pl;
  pl.rows_m = rows_m;
  pl.npad = round_up(d->N, 8);
  pl.tx_bytes = (long long)d->N * hw * kROWB;
  pl.cnt_off = 2 * ((pl.tx_bytes + 255) / 256 * 256);
  pl.status_off = pl.cnt_off + (long long)pl.npad * 64;
  pl.total = pl.status_off + 64;
  if (out) *out = pl;
  return FT_OK;
}

}  // namespace
}  // namespace ft

extern "C" int ft_bottleneck_cluster_supported(const ft_bottleneck_desc* d) { return ft::bnc_plan(d, nullptr); }

extern "C" long long ft_bottleneck_cluster_workspace_bytes(const ft_bottleneck_desc* d) {
  ft::BncPlan pl;
  return ft::bnc_plan(d, &pl) == FT_OK ? pl.total : 0;
}

extern "C" long long ft_bottleneck_cluster_status_offset(const ft_bottleneck_desc* d) {
  ft::BncPlan pl;
  return ft::bnc_plan(d, &pl) == FT_OK ? pl.status_off : -1;
}

extern "C" int ft_bottleneck_cluster_fwd(const ft_bottleneck_desc* d, const void* x, const void* wstream, const float* tables, void* y,
                                         void* workspace, ft_stream_t stream) {
  using namespace ft;
  BncPlan pl;
  const int st = bnc_plan(d, &pl);
  if (st != FT_OK) return st;
  if (!x || !wstream || !tables || !y || !workspace || x == y) return FT_ERR_INVALID_ARG;
  BncParams p{};
  p.x = static_cast<const char*>(x);
  p.y = static_cast<char*>(y);
  p.ws = static_cast<const char*>(wstream);
  p.tab = reinterpret_cast<const char*>(tables);
  char* wsb = static_cast<char*>(workspace);
  p.t1x = wsb;
  p.t2x = wsb + pl.cnt_off / 2;
  p.cnt = reinterpret_cast<unsigned*>(wsb + pl.cnt_off);
  p.status = reinterpret_cast<unsigned*>(wsb + pl.status_off);
  p.N = d->N; p.H = d->H; p.W = d->W; p.HW = d->H * d->W;
  p.rows_m = pl.rows_m;
  p.x_cstride = d->x_cstride; p.x_coff = d->x_coff; p.y_cstride = d->y_cstride; p.y_coff = d->y_coff;
  p.x_bytes = (unsigned)((size_t)d->N * p.HW * d->x_cstride * 2);
  p.y_bytes = (unsigned)((size_t)d->N * p.HW * d->y_cstride * 2);
  p.ws_bytes = (unsigned)((kG3 + 4 * kKC) * kWSTEP);
  p.tx_bytes = (unsigned)pl.tx_bytes;
  static const int dbg = getenv("FT_BNC_DBG") ? atoi(getenv("FT_BNC_DBG")) : 0;
  p.dbg = dbg;
  hipStream_t s = as_stream(stream);
  auto k = bottleneck_cluster_kernel;
  FT_RAISE_LDS(k, kLdsBytes);
  hipLaunchKernelGGL(k, dim3(pl.npad * 4), dim3(256), kLdsBytes, s, p);
  FT_LAUNCH_CHECK("bottleneck_cluster_kernel");
  return FT_OK;
}
