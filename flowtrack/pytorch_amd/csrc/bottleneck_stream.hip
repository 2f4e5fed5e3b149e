// Whole-bottleneck fusion for the 128- and 256-plane ResNet stages (fp16): conv1 1x1 + bn1 + relu -> conv2 3x3 + bn2 +
// relu -> conv3 1x1 + bn3 + identity residual + relu in ONE launch, with every weight byte STREAMED through LDS
// (reference: Bottleneck.forward, lib/pose/models/blocks.py:105-120, the blocks without a projection shortcut:
// layer2.1-3 / layer3.1-5 of ResNet-50, layer3.1-22 of ResNet-101, resnet.py:29-36,51-55).
//
// Why a second fused kernel (bottleneck.hip covers the 64-plane stage): at 512 / 1024 channels the three launches of a
// block are no longer HBM-bound, they are bound by operand delivery L2 -> LDS: with 12 k .. 49 k pixels per layer the
// GEMM tiles that fill 256 CUs are small (64x64 .. 128x128), every operand byte is re-delivered M/BP or N/BC times
// (200 + 226 + 200 MB per layer3 block at batch 64) and the three launches take 82 (layer3) / 94 us (layer2) for 27.4
// GFLOP.  Here a workgroup owns a full-width strip of output rows of one image, keeps t1 and t2 in LDS and streams the
// block's weights ONCE through a ring (2.2 MB per workgroup at 256 planes, 0.55 MB at 128): the pixel operand never
// leaves the CU, the weight operand arrives as 1-KiB-contiguous DMA pieces (pre-packed in MFMA fragment order, so the
// LDS image needs no swizzle and ds_read_b128 is lane-linear).  tools/dev/ubench/stream_ring.hip is the micro-model of
// the inner loop: 1.24 PFLOP/s chip-wide at 256 workgroups (21 B/clk/CU of weights beside the matrix pipe).
//
// Geometry (P planes, C = 4P channels; 4 waves = WCOLS x WPG, wave = 2 output-channel tiles x MT pixel tiles of 32):
//   P = 256: WCOLS 4, WPG 1: strip <= 96 output pixels (MT2 = 3) on <= 120 halo pixels (MT1 = 4)   [or 64 on 96]
//   P = 128: WCOLS 2, WPG 2: strip <= 192 output pixels on <= 256 halo pixels
//   phase 1  t1 = relu(bn1(W1 . x)) on the strip + one halo row above and below: x streams in 64-channel chunks (whole
//            128-byte lines per pixel, XOR-swizzled on the source side), W1 in matching K-slices; the residual (the
//            strip's own pixels) is picked out of the chunks in phase 3's accumulator layout as they pass through LDS.
//   phase 2  t2 = relu(bn2(W2 * t1)): the pixel operand of tap (ky, kx) is T1 at row offset ky*W + kx - 1 (full-width
//            strips make the 3x3 neighbourhood a linear shift; the two x-border taps are lane-masked); only W2 streams.
//   phase 3  y = relu(bn3(W3 . t2) + x) in four quarters of P output channels; each lane owns 16 consecutive channels of
//            a pixel (the A-fragment rows are permuted at pack time), so y leaves as 32-byte runs per lane.
// LDS: T1 / T2 (aliased) <= 60 / 64 KiB, three x-chunk buffers (inside the T1 region during phase 1), three weight-step
// buffers, two folded-BN table buffers = 160 KiB, one workgroup per CU.
#include <stdlib.h>

#include <type_traits>

#include "ft_common.h"

namespace ft {
namespace {

struct BnsParams {
  const char* x;
  char* y;
  const char* ws;    // packed weight stream (ft_bottleneck_stream_pack)
  const char* tab;   // float [6][2P]: {s1 b1} {s2 b2} {s3 b3 of quarter 0} .. {quarter 3}
  int H, W, TH;      // image size, output rows per strip
  int ppi, total;    // strips per image (x column parts), workgroups
  int TWc, csplit;   // column-split form of the direct kernel: output columns per workgroup, parts per row of strips
  int Ho, Wo;        // stride-2 head form (conv1 + conv2 of a stage's entry block): output map; H, W are the INPUT map
  int x_cstride, x_coff, y_cstride, y_coff;
  unsigned x_bytes, y_bytes, ws_bytes;
  int dbg;
};

template <int N, int I = 0, typename F>
__device__ __forceinline__ void bns_unroll(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    bns_unroll<N, I + 1>(f);
  }
}

#define BNS_BARRIER() asm volatile("s_barrier" ::: "memory")
// XOR key of the x-chunk buffers' 128-byte rows: two rows share a 256-byte bank row (see bottleneck.hip: BNK_KEY)
#ifndef FT_BNS_XKEY_SHIFT
#define FT_BNS_XKEY_SHIFT 1
#endif
#define BNS_XKEY(hp) (((hp) >> FT_BNS_XKEY_SHIFT) & 7)
#ifndef FT_BNS_STG
#define FT_BNS_STG 1    // dev A/B: 0 = phase 3 of the direct kernel stores straight from the accumulator layout (no LDS staging tile)
#endif
#ifndef FT_BNS_TOUCH_FIRST
#define FT_BNS_TOUCH_FIRST 0   // 1 = the L2 touch and the table / shift loads in FRONT of x chunk 0 (rounds 3-5), 0 = behind it and the first weight step
#endif
#ifndef FT_BNS_WSTG
#define FT_BNS_WSTG 1   // round 6: phase 3 of the direct kernel (two channel tiles per wave) transposes a quarter's tile through a WAVE-PRIVATE
                        // piece of the staging tile: the wave's 64 channels of a pixel are one aligned 128-byte run of y, so it writes whole
                        // lines without meeting the other waves — no workgroup barrier per quarter (the LDS queue of a wave is in order)
#endif
#ifndef FT_BNS_DIRECT_AUX
#define FT_BNS_DIRECT_AUX 0   // cache policy of those stores (plain: the L2 merges the 16-byte pieces of a line)
#endif
#ifndef FT_BNS_OVL
#define FT_BNS_OVL 0    // dev A/B: phase 3 of the direct kernel hides a quarter's epilogue inside the next quarter's weight steps
#endif
// the ring kernel's phase-3 epilogue (table form) in packed form (ft_common.h: bn_res_relu_acc8): 0 = scalar (the packed form needs
// aligned register pairs and pushed <128,4,3> from 506 registers / no spill to 512 / 48 spilled)
#ifndef FT_BNS_PK_RES
#define FT_BNS_PK_RES 0
#endif
#ifndef FT_BNS_L2_TOUCH
#define FT_BNS_L2_TOUCH 1   // the direct kernel's first round of workgroups pulls the weight stream into its XCD's L2 (one touch per line)
#endif
#ifndef FT_BNSD_ABL
#define FT_BNSD_ABL 0   // dev ablations of the direct kernel (TIMING ONLY, results are wrong): 1 = no chunk barriers in phase 1, 2 = no residual
                        // pick-up, 16 / 32 = the weight loads of phase 1 / phases 2 + 3 are not issued at all (FT_BNS_DBG=64 still issues them)
#endif
#ifndef FT_BNS_PIN
#define FT_BNS_PIN 3    // dev A/B: bit 0 = pinned issue order in phase 1 of the direct kernel, bit 1 = in its weight steps (dstep)
#endif

// MFMA row r of an A fragment holds output channel sigma(r) of its 32-channel tile, so that accumulator register k of
// lane (pixel, half) is channel 16 * half + k: 16 consecutive channels per lane.
__host__ __device__ constexpr int bns_sigma(int r) { return 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3); }

template <int P>
struct BnsGeom {
  static constexpr int C = 4 * P;
  static constexpr int NCT = P / 32;          // output-channel tiles of a P-wide GEMM
  static constexpr int WCOLS = P / 64;        // wave columns (2 channel tiles each)
  static constexpr int WPG = 4 / WCOLS;       // pixel groups
  static constexpr int NC1 = C / 64;          // x chunks = weight steps of phase 1
  static constexpr int KC = P / 64;           // 64-wide K chunks of a P-deep GEMM
  static constexpr int WSTEP = NCT * 4096;    // bytes of a weight step: 4 K16 slices x NCT fragments x 1 KiB
  static constexpr int LW = NCT;              // 1-KiB weight loads per wave per step
  static constexpr int G2 = NC1, G3 = NC1 + 9 * KC, GEND = G3 + 4 * KC;
  static constexpr int ROWB = 2 * P;          // bytes of a T1 / T2 / staging row
  static constexpr int TABB = 8 * P;          // bytes of one table {scale[P], shift[P]}
  static constexpr int LT = P / 128;          // 256-byte table loads per wave
  // LDS map: [0, ..) three x-chunk buffers in phase 1, then T1, then T2 (+ the output staging tile behind it at 128
  // planes); one folded-BN table, one all-zero row (the x-border taps read it), three weight-step buffers
  static constexpr int XSTRIDE = P == 256 ? 16384 : 32768;
  static constexpr int T1_ROWS = P == 256 ? 120 : 256;
  static constexpr int TAB = P == 256 ? 61440 : 98304;
  static constexpr int ZROW = TAB + TABB;
  static constexpr int WBASE = P == 256 ? 65536 : 114688;
  static constexpr int STG = 49152;           // P = 128: staging tile [48 K, 96 K)
  static constexpr int LDS_BYTES = 163840;
  static_assert(ZROW % ROWB == 0 && ZROW + ROWB <= WBASE && WBASE + 3 * WSTEP == LDS_BYTES, "LDS map");
};

// FOLD: the folded operands (see bottleneck_stream_direct_kernel): scales in the weights, shifts as (hi, lo) pairs in p.tab, shift and
// residual added by MFMA, every epilogue = fp16(relu(acc)); no table ever travels through LDS.
template <int P, int MT1, int MT2, bool FOLD = false>
__global__ __launch_bounds__(256, 1) void bottleneck_stream_kernel(const BnsParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using G = BnsGeom<P>;
  constexpr int NCT = G::NCT, WCOLS = G::WCOLS, WPG = G::WPG, NC1 = G::NC1, KC = G::KC, WSTEP = G::WSTEP, LW = G::LW;
  constexpr int ROWB = G::ROWB;
  constexpr int XROWS = WPG * MT1 * 32, LX = WPG * MT1;     // rows of an x chunk buffer; 1-KiB x loads per wave per chunk
  constexpr int NOUT = WPG * MT2 * 32;                       // output pixels of the strip (padded)
  // the output tile of a quarter goes through LDS for whole-line stores: behind T2 at 128 planes, in the weight buffer the
  // quarter's last step has just released at 256 planes (32 KiB: the 64-pixel strips only)
  constexpr bool STAGED = P == 128 || NOUT * ROWB <= WSTEP;
  static_assert(XROWS * 128 <= G::XSTRIDE, "x chunk buffer");
  static_assert(NOUT * ROWB <= 49152, "T2 fits below the staging tile / inside the T1 region");
  static_assert(2 * (LX + LW) + 6 + 12 <= 63, "vmcnt immediate");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  using c0 = std::integral_constant<int, 0>;
  using c1 = std::integral_constant<int, 1>;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wcol = wave % WCOLS, pg = wave / WCOLS;
  const int l31 = lane & 31, lhi = lane >> 5;

  int logical;
  {
    const int b = blockIdx.x;
    const int q = p.total >> 3, r = p.total & 7, xcd = b & 7, loc = b >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int n = logical / p.ppi;
  const int y0 = (logical - n * p.ppi) * p.TH;
  const int W = p.W;
  const int rows_out = p.H - y0 < p.TH ? p.H - y0 : p.TH;
  const int npix_out = rows_out * W;
  const int npix_halo = (p.TH + 2) * W;

  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.ws), 0, p.ws_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.tab), 0, FOLD ? (2 * P + G::C) * 4 : 6 * G::TABB, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;

  // folded form: shift pairs of this wave's MFMA rows per (epilogue, tile) and the residual's 0/1 matrix (see the direct kernel)
  [[maybe_unused]] unsigned shp[6][2];
  [[maybe_unused]] uint4_t permA[2];
  [[maybe_unused]] const uint4_t onesB = uint4_t{0x3C003C00u, 0u, 0u, 0u};
  auto load_shp = [&]() {
    const unsigned so = lhi == 0 ? 4u * (unsigned)bns_sigma(l31) : kOOB;
#pragma unroll
    for (int e = 0; e < 6; ++e)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        shp[e][i] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_t, so, 4 * (e * P + (2 * wcol + i) * 32), 0);
  };
  if constexpr (FOLD) {
    const int sg = bns_sigma(l31);
    const bool on = lhi == (sg >> 4);
    const int ph = (sg >> 3) & 1, pe = sg & 7;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      unsigned w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = (on && ph == hh && (pe >> 1) == k) ? (0x3C00u << (16 * (pe & 1))) : 0u;
      permA[hh] = uint4_t{w[0], w[1], w[2], w[3]};
    }
  }
  [[maybe_unused]] auto add_shift = [&](int e, auto& A, auto mtc) {      // A[i][j] += shift of epilogue e (one MFMA per tile)
    constexpr int MT = decltype(mtc)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint4_t sa = uint4_t{shp[e][i], 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < MT; ++j)
        A[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, sa), __builtin_bit_cast(half8_t, onesB), A[i][j], 0, 0, 0);
    }
  };

  // the first round of workgroups on an XCD pulls the block's weight stream into that XCD's L2, each its own 1/n-th, one
  // dword per 128-byte line (see the direct kernel); the scratch corner sits between the zero row and the weight ring.
  // EXACTLY kTouch loads per thread (lines past the share: out of range, no traffic): the touch is issued BEHIND the first x
  // chunk (round 6: it used to sit in front of it in every wave's in-order load queue) and the first hand-counted wait counts it
  constexpr int kTouch = FT_BNS_L2_TOUCH ? 6 : 0;
  auto issue_touch = [&]() {
#if FT_BNS_L2_TOUCH
    constexpr int SCR = P == 256 ? 64000 : 102400;
    static_assert(SCR >= G::ZROW + ROWB && SCR + 1024 <= G::WBASE, "scratch of the L2 touch loads");
    const bool on = blockIdx.x < 256 && !(p.dbg & (512 | 1024));
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int first = p.total < 256 ? p.total : 256;
    const int nloc = (first - xcd + 7) >> 3;
    const unsigned lines = (p.ws_bytes + 127u) >> 7;
    const unsigned per = (lines + nloc - 1) / nloc;
    const unsigned lo = loc * per, hi = lo + per < lines ? lo + per : lines;
#pragma unroll
    for (int k = 0; k < kTouch; ++k) {
      const unsigned l = lo + tid + 256u * k;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr)(smem + SCR + wave * 256), 4, (on && l < hi) ? l << 7 : kOOB, 0, 0, 0);
    }
#endif
  };
  // ---- loaders -------------------------------------------------------------------------------------------------------
  // x chunk: row = halo pixel, 128 bytes (64 channels); a 1-KiB wave load covers 8 rows, lane -> (row = lane / 8,
  // 16-byte position lane % 8), the XOR swizzle (position ^= row & 7) is applied to the SOURCE position
  unsigned x_voff[LX];
#pragma unroll
  for (int t = 0; t < LX; ++t) {
    const int hp = (t * 4 + wave) * 8 + (lane >> 3);
    const int hr = hp / W, hc = hp - hr * W;
    const int iy = y0 - 1 + hr;
    unsigned v = kOOB;
    if (hp < npix_halo && (unsigned)iy < (unsigned)p.H)
      v = (unsigned)((((n * p.H + iy) * W + hc) * p.x_cstride + p.x_coff) * 2 + (((lane & 7) ^ BNS_XKEY(hp)) << 4));
    x_voff[t] = v;
  }
  const unsigned lane16 = (unsigned)lane * 16u;
  auto issue_x = [&](int c, int buf) {
    char* dst = smem + buf * G::XSTRIDE;
#pragma unroll
    for (int t = 0; t < LX; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr)(dst + (t * 4 + wave) * 1024), 16, x_voff[t], c * 128, 0, 0);
  };
  auto issue_w = [&](int g, int buf) {
    char* dst = smem + G::WBASE + buf * WSTEP;
#pragma unroll
    for (int t = 0; t < LW; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr)(dst + (t * 4 + wave) * 1024), 16, lane16,
                                               g * WSTEP + (t * 4 + wave) * 1024, 0, 0);
  };
  auto issue_tab = [&](int e) {       // {scale[P], shift[P]} of epilogue e; one buffer: issued once epilogue e-1 is behind a barrier
    if constexpr (FOLD) return;
#pragma unroll
    for (int t = 0; t < G::LT; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_t, (lds_ptr)(smem + G::TAB + (t * 4 + wave) * 256), 4, (unsigned)lane * 4u,
                                               e * G::TABB + (t * 4 + wave) * 256, 0, 0);
  };
  // per-lane A-fragment base inside a weight step: fragment (kk, channel tile 2*wcol + i) at (kk * NCT + 2*wcol + i) KiB
  const unsigned a_base = (unsigned)(G::WBASE + (2 * wcol) * 1024) + lane16;
  uint4_t fa[2][2];                   // A fragments, two register sets: slice k+1 is read while slice k multiplies

  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define BNS_TS(i) do { if (p.dbg & 32) ts[i] = __builtin_amdgcn_s_memtime(); } while (0)
  BNS_TS(0);
  // prologue: chunk 0 (x + W1 slice) leads every wave's load queue, then the L2 touch and table 0 / the shift pairs, chunks 1 and 2,
  // the zero row
#if FT_BNS_TOUCH_FIRST
  issue_touch();
  if constexpr (FOLD) load_shp();
  else issue_tab(0);
  asm volatile("" ::: "memory");
  issue_x(0, 0); issue_w(0, 0);
#else
  issue_x(0, 0); issue_w(0, 0);
  issue_touch();
  if constexpr (FOLD) load_shp();
  else issue_tab(0);
  asm volatile("" ::: "memory");
#endif
  issue_x(1, 1); issue_w(1, 1);
  issue_x(2, 2); issue_w(2, 2);
  if (tid < ROWB / 16) *reinterpret_cast<uint4_t*>(smem + G::ZROW + tid * 16) = uint4_t{0u, 0u, 0u, 0u};

  // residual = the block input at the strip's own pixels, in phase 3's accumulator layout: [quarter][tile i][pixel tile j][half]
  uint4_t res[4][2][MT2][2];
  int m_out[MT2];       // output pixel (strip-relative, row-major at width W) of this lane per pixel tile
#pragma unroll
  for (int j = 0; j < MT2; ++j) m_out[j] = (pg * MT2 + j) * 32 + l31;

  // ================= phase 1: t1 = relu(bn1(W1 . x)) on the halo strip =============================================
  float16_t acc1[2][MT1];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < MT1; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[i][j][r] = 0.f;
  int b1_off[MT1];      // x-chunk fragment offsets (buffer-relative): row hp, 16-byte position (kk*2 + lhi) ^ BNS_XKEY(hp)
#pragma unroll
  for (int j = 0; j < MT1; ++j) {
    const int hp = (pg * MT1 + j) * 32 + l31;
    b1_off[j] = hp * 128 + ((lhi ^ BNS_XKEY(hp)) << 4);
  }
  {
    uint4_t fx[2][MT1];
    auto ld1 = [&](auto setc, int buf, int kk) {
      constexpr int S = decltype(setc)::value;
      const unsigned ab = a_base + buf * WSTEP;
      const char* xb = smem + buf * G::XSTRIDE;
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[S][i] = *reinterpret_cast<const uint4_t*>(smem + ab + (kk * NCT + i) * 1024);
#pragma unroll
      for (int j = 0; j < MT1; ++j) fx[S][j] = *reinterpret_cast<const uint4_t*>(xb + (b1_off[j] ^ (kk << 5)));
    };
    auto mma1 = [&](auto setc) {
      constexpr int S = decltype(setc)::value;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < MT1; ++j)
          acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, fa[S][i]), __builtin_bit_cast(half8_t, fx[S][j]),
                                                              acc1[i][j], 0, 0, 0);
    };
    // chunk 0 has landed (this wave's share) while the touch, the table / shift loads and chunks 1 and 2 fly; after the barrier everyone's share has
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * (LX + LW) + (FT_BNS_TOUCH_FIRST ? 0 : kTouch + (FOLD ? 12 : G::LT))) : "memory");
    BNS_BARRIER();
    ld1(c0{}, 0, 0);
    bns_unroll<NC1>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      constexpr int buf = c % 3;
      __builtin_amdgcn_sched_barrier(0);
      ld1(c1{}, buf, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma1(c0{});
      __builtin_amdgcn_sched_barrier(0);
      ld1(c0{}, buf, 2);
      if (c % WCOLS == wcol) {       // this chunk holds the channels of this wave column for quarter c / WCOLS
        constexpr int q = c / WCOLS;
        const char* xb = smem + buf * G::XSTRIDE;
#pragma unroll
        for (int j = 0; j < MT2; ++j) {
          const int hp = m_out[j] + W;
          const char* rowp = xb + hp * 128;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
              res[q][i][j][h] = *reinterpret_cast<const uint4_t*>(rowp + (((4 * i + 2 * lhi + h) ^ BNS_XKEY(hp)) << 4));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      mma1(c1{});
      __builtin_amdgcn_sched_barrier(0);
      ld1(c1{}, buf, 3);
      __builtin_amdgcn_sched_barrier(0);
      mma1(c0{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (c + 1 < NC1) {
        // chunk c+1 has landed (chunk c+2 may fly), every read of chunk c's buffers is complete: refill them with chunk c+3
        if constexpr (c + 2 < NC1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LX + LW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LW) : "memory");
        BNS_BARRIER();
        if constexpr (c + 3 < NC1) issue_x(c + 3, buf);
        issue_w(c + 3, buf);
        ld1(c0{}, (c + 1) % 3, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      mma1(c1{});
    });
  }
  // every wave is past its last x-chunk read once it reaches this barrier: the x buffers become T1
  BNS_TS(1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  BNS_BARRIER();
  {
    if constexpr (FOLD) add_shift(0, acc1, std::integral_constant<int, MT1>{});
    [[maybe_unused]] const float* tb = reinterpret_cast<const float*>(smem + G::TAB);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      [[maybe_unused]] const int ch = (2 * wcol + i) * 32 + 16 * lhi;
      [[maybe_unused]] float4_t sc[4], sh[4];
      if constexpr (!FOLD) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          sc[g4] = *reinterpret_cast<const float4_t*>(tb + ch + g4 * 4);
          sh[g4] = *reinterpret_cast<const float4_t*>(tb + P + ch + g4 * 4);
        }
      }
#pragma unroll
      for (int j = 0; j < MT1; ++j) {
        const int hp = (pg * MT1 + j) * 32 + l31;
        const int hr = hp / W;
        const int iy = y0 - 1 + hr;
        const bool inside = (unsigned)iy < (unsigned)p.H;     // out-of-image halo rows are conv2's zero padding
        half8_t h8[2];
        if constexpr (FOLD) relu_acc16(acc1[i][j], h8);
        else bn_relu_acc16(acc1[i][j], sc, sh, h8);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          uint4_t u = __builtin_bit_cast(uint4_t, h8[h]);
          u.x = inside ? u.x : 0u; u.y = inside ? u.y : 0u; u.z = inside ? u.z : 0u; u.w = inside ? u.w : 0u;
          h8[h] = __builtin_bit_cast(half8_t, u);
        }
        if (hp < npix_halo) {
          char* rowp = smem + hp * ROWB;
          const int cb = (2 * wcol + i) * 4 + 2 * lhi;           // 16-byte chunk of channel `ch`
#pragma unroll
          for (int h = 0; h < 2; ++h)
            *reinterpret_cast<half8_t*>(rowp + (((cb + h) ^ (hp & 15)) << 4)) = h8[h];
        }
      }
    }
  }
  BNS_TS(2);

  // ================= phases 2 + 3: weight steps G2 .. GEND-1, pixel operand from T1 / T2 ===========================
  float16_t acc[2][MT2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < MT2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  zero_acc();
  uint4_t fb[2][MT2];
  auto ld2 = [&](auto setc, int buf, int kk, const int (&rb)[MT2]) {
    constexpr int S = decltype(setc)::value;
    const unsigned ab = a_base + buf * WSTEP;
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[S][i] = *reinterpret_cast<const uint4_t*>(smem + ab + (kk * NCT + i) * 1024);
#pragma unroll
    for (int j = 0; j < MT2; ++j) fb[S][j] = *reinterpret_cast<const uint4_t*>(smem + (rb[j] ^ (kk << 5)));
  };
  auto mma2 = [&](auto setc) {
    constexpr int S = decltype(setc)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < MT2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, fa[S][i]), __builtin_bit_cast(half8_t, fb[S][j]),
                                                           acc[i][j], 0, 0, 0);
  };
  // x-border flags of this lane's output pixels: bit 0 = first column (tap kx = 0 is padding), bit 1 = last column
  int edge[MT2];
#pragma unroll
  for (int j = 0; j < MT2; ++j) {
    const int ox = m_out[j] % W;
    edge[j] = (ox == 0 ? 1 : 0) | (ox == W - 1 ? 2 : 0);
  }
  // byte base of pixel row `m + off` of the T region for K chunk kc (slice kk is one more XOR); lanes whose tap is
  // x-padding read the zero row instead
  auto row_bases = [&](int off, int kc, int bad, int (&rb)[MT2]) {
#pragma unroll
    for (int j = 0; j < MT2; ++j) {
      const int row = m_out[j] + off;
      const int v = row * ROWB + (((row & 15) ^ lhi) << 4);
      rb[j] = ((edge[j] & bad) ? G::ZROW + (lhi << 4) : v) ^ (kc << 7);
    }
  };
  // one weight step (slices 1..3 of it, slice 0 already sits in register set 0) with the hand-over to the next one in front
  // of its last slice: the next step's weights have landed (this wave's share, then everyone's), every read of THIS step's
  // buffer is complete, so the buffer is refilled with step g+3 (unless `defer`: phase 3 stages its output tile there first)
  auto pstep = [&](int g, int buf, const int (&rb)[MT2], bool has_next, const int (&rb_next)[MT2], int tab_e, bool defer) {
    __builtin_amdgcn_sched_barrier(0);
    ld2(c1{}, buf, 1, rb);
    __builtin_amdgcn_sched_barrier(0);
    mma2(c0{});
    __builtin_amdgcn_sched_barrier(0);
    ld2(c0{}, buf, 2, rb);
    __builtin_amdgcn_sched_barrier(0);
    mma2(c1{});
    __builtin_amdgcn_sched_barrier(0);
    ld2(c1{}, buf, 3, rb);
    __builtin_amdgcn_sched_barrier(0);
    mma2(c0{});
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) {
      if (g + 2 < G::GEND) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      BNS_BARRIER();
      if (tab_e >= 0) issue_tab(tab_e);
      if (g + 3 < G::GEND && !defer) issue_w(g + 3, buf);
      ld2(c0{}, buf == 2 ? 0 : buf + 1, 0, rb_next);
      __builtin_amdgcn_sched_barrier(0);
    }
    mma2(c1{});
  };
  // start of a phase: the T region has just been written, weight step g has landed: slice 0 into register set 0
  auto pstart = [&](int g, int buf, const int (&rb)[MT2], int tab_e) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LW) : "memory");
    BNS_BARRIER();
    issue_tab(tab_e);
    issue_w(g + 2, buf == 0 ? 2 : buf - 1);          // (g + 2) % 3: the buffer of step g-1
    ld2(c0{}, buf, 0, rb);
  };

  // ---- phase 2: nine taps x KC chunks ---------------------------------------------------------------------------------
  {
    int rb[MT2], rbn[MT2];
    row_bases(-1, 0, 1, rb);                         // tap (0, 0), chunk 0
    pstart(G::G2, G::G2 % 3, rb, 1);
    int g = G::G2, buf = G::G2 % 3;
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const bool has_next = !(tap == 8 && kc == KC - 1);
        int nky = ky, nkx = kx, nkc = kc + 1;
        if (nkc == KC) { nkc = 0; if (++nkx == 3) { nkx = 0; ++nky; } }
        row_bases(nky * W + nkx - 1, nkc, nkx == 0 ? 1 : (nkx == 2 ? 2 : 0), rbn);
        pstep(g, buf, rb, has_next, rbn, -1, false);
#pragma unroll
        for (int j = 0; j < MT2; ++j) rb[j] = rbn[j];
        ++g;
        buf = buf == 2 ? 0 : buf + 1;
      }
    }
  }
  // every wave is past its last T1 read once it reaches this barrier: T2 overwrites T1
  BNS_TS(3);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  BNS_BARRIER();
  {
    if constexpr (FOLD) add_shift(1, acc, std::integral_constant<int, MT2>{});
    [[maybe_unused]] const float* tb = reinterpret_cast<const float*>(smem + G::TAB);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      [[maybe_unused]] const int ch = (2 * wcol + i) * 32 + 16 * lhi;
      [[maybe_unused]] float4_t sc[4], sh[4];
      if constexpr (!FOLD) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          sc[g4] = *reinterpret_cast<const float4_t*>(tb + ch + g4 * 4);
          sh[g4] = *reinterpret_cast<const float4_t*>(tb + P + ch + g4 * 4);
        }
      }
#pragma unroll
      for (int j = 0; j < MT2; ++j) {
        const int m = m_out[j];
        half8_t h8[2];
        if constexpr (FOLD) relu_acc16(acc[i][j], h8);
        else bn_relu_acc16(acc[i][j], sc, sh, h8);
        char* rowp = smem + m * ROWB;
        const int cb = (2 * wcol + i) * 4 + 2 * lhi;
#pragma unroll
        for (int h = 0; h < 2; ++h) *reinterpret_cast<half8_t*>(rowp + (((cb + h) ^ (m & 15)) << 4)) = h8[h];
      }
    }
  }
  zero_acc();
  BNS_TS(4);

  // ---- phase 3: four quarters of P output channels, K = P -------------------------------------------------------------
  {
    // direct form: lane -> 32-byte runs of its own pixel; staged form: thread -> 16-byte chunks of whole pixel rows
    unsigned y_voff[MT2];
#pragma unroll
    for (int j = 0; j < MT2; ++j) {
      const int m = m_out[j];
      y_voff[j] = m < npix_out ? (unsigned)((((n * p.H + y0) * W + m) * p.y_cstride + p.y_coff + (2 * wcol) * 32 + 16 * lhi) * 2) : kOOB;
    }
    constexpr int CPR = ROWB / 16;                   // 16-byte chunks per staging row
    constexpr int NSTG = NOUT * CPR / 256;           // chunks per thread
    unsigned s_voff[STAGED ? NSTG : 1];
    int s_off[STAGED ? NSTG : 1];
    if constexpr (STAGED) {
#pragma unroll
      for (int k = 0; k < NSTG; ++k) {
        const int idx = tid + 256 * k, m = idx / CPR, ch = idx % CPR;
        s_voff[k] = m < npix_out ? (unsigned)((((n * p.H + y0) * W + m) * p.y_cstride + p.y_coff + ch * 8) * 2) : kOOB;
        s_off[k] = m * ROWB + ((ch ^ (m & 15)) << 4);
      }
    }
    int rb[MT2];
    row_bases(0, 0, 0, rb);
    pstart(G::G3, G::G3 % 3, rb, 2);
    bns_unroll<4>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      bns_unroll<KC>([&](auto kcc) {
        constexpr int kc = decltype(kcc)::value;
        constexpr int g = G::G3 + q * KC + kc;
        constexpr bool last_of_q = kc == KC - 1;
        int rbn[MT2];
        row_bases(0, last_of_q ? 0 : kc + 1, 0, rbn);
        // the table of the NEXT quarter's epilogue follows the barrier of the quarter's first step
        pstep(g, g % 3, rb, g + 1 < G::GEND, rbn, (kc == 0 && q > 0) ? 2 + q : -1, STAGED && P == 256 && last_of_q);
#pragma unroll
        for (int j = 0; j < MT2; ++j) rb[j] = rbn[j];
      });
      if constexpr (q == 3) {
        // the last table: no later hand-over waits for it, and every wave loaded only its share of it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BNS_BARRIER();
      }
      constexpr int glast = G::G3 + q * KC + KC - 1;
      char* stg = smem + (P == 128 ? G::STG : G::WBASE + (glast % 3) * WSTEP);
      [[maybe_unused]] const float* tb = reinterpret_cast<const float*>(smem + G::TAB);
      if constexpr (FOLD) {
        // shift3 of the quarter and the residual join the accumulators as three MFMAs per tile
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint4_t sa = uint4_t{shp[2 + q][i], 0u, 0u, 0u};
#pragma unroll
          for (int j = 0; j < MT2; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, sa), __builtin_bit_cast(half8_t, onesB), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, permA[h]), __builtin_bit_cast(half8_t, res[q][i][j][h]),
                                                                 acc[i][j], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        [[maybe_unused]] const int ch = (2 * wcol + i) * 32 + 16 * lhi;
        [[maybe_unused]] float4_t sc[4], sh[4];
        if constexpr (!FOLD) {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            sc[g4] = *reinterpret_cast<const float4_t*>(tb + ch + g4 * 4);
            sh[g4] = *reinterpret_cast<const float4_t*>(tb + P + ch + g4 * 4);
          }
        }
#pragma unroll
        for (int j = 0; j < MT2; ++j) {
          [[maybe_unused]] half8_t o8[2];
          if constexpr (FOLD) relu_acc16(acc[i][j], o8);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            half8_t o;
            if constexpr (FOLD) {
              o = o8[h];
            } else {
              const half8_t rs = __builtin_bit_cast(half8_t, res[q][i][j][h]);
#if FT_BNS_PK_RES
              o = bn_res_relu_acc8(acc[i][j], h, sc, sh, rs);
#else
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const int r = h * 8 + e;
                o[e] = (half_t)__builtin_fmaxf(acc[i][j][r] * sc[r >> 2][r & 3] + sh[r >> 2][r & 3] + (float)rs[e], 0.f);
              }
#endif
            }
            if constexpr (STAGED) {
              const int m = m_out[j];
              *reinterpret_cast<half8_t*>(stg + m * ROWB + ((((2 * wcol + i) * 4 + 2 * lhi + h) ^ (m & 15)) << 4)) = o;
            } else if (!(p.dbg & 4)) {
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, o), rsrc_y, y_voff[j] + (unsigned)((q * P + i * 32 + h * 8) * 2), 0, FT_YSTORE_BUF_AUX);
            }
          }
        }
      }
      zero_acc();
      if constexpr (STAGED) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        BNS_BARRIER();
#pragma unroll
        for (int k = 0; k < NSTG; ++k) {
          const uint4_t v = *reinterpret_cast<const uint4_t*>(stg + s_off[k]);
          if (!(p.dbg & 4)) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_y, s_voff[k] + (unsigned)(q * P * 2), 0, FT_YSTORE_BUF_AUX);
        }
        if constexpr (P == 256 && q < 3) {
          // the staging tile sat in a weight buffer: hand it back (its refill with step glast + 3 was deferred)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          BNS_BARRIER();
          if constexpr (glast + 3 < G::GEND) issue_w(glast + 3, glast % 3);
        }
      }
    });
  }
  if (p.dbg & 32) {       // dev: phase timestamps of wave 0 over the strip's first output pixel (the output is garbage then)
    ts[5] = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts[6] = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.y + ((size_t)((n * p.H + y0) * W) * p.y_cstride + p.y_coff) * 2);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = ts[i];
    }
  }
#endif
}

// ---- 256 planes, weights straight to registers -----------------------------------------------------------------------
// At 256 planes every wave owns its own two output-channel tiles, so no weight fragment is shared between waves: the
// LDS round trip of the weight stream (DMA write + ds_read, competing with the pixel-operand reads for the LDS port) is
// pure overhead.  Here each wave loads ITS fragments of a step (8 x 1 KiB, contiguous in the fragment-ordered stream)
// straight into VGPRs two steps ahead; only the pixel operand (x chunks, T1, T2) lives in LDS.  No ring, no ring
// barriers: the waves only meet at the phase transitions.  tools/dev/ubench/stream_ring.hip: 447 vs ~680 ns per 32-KiB
// step on the 2 x 2 micro-tile at 256 workgroups (30 vs 20 B/clk/CU of weights beside the matrix pipe).
// LDS: [0, 48 K) x-chunk buffers / [0, 60 K) T1 then T2, zero row at 60 K, all six folded-BN tables at 64 K (loaded
// once), two output staging tiles from 80 K.
// XH = column-split form for small batches (R101 384x288 at 16 crops per GPU gives 128 full-width strips for 256 CUs): a
// workgroup takes TWc of the W columns of its rows and carries its own x-halo — the T1 patch is (TH + 2) x (TWc + 2) with
// out-of-image columns zeroed like out-of-image rows, so every 3x3 tap is a plain shift inside the patch (no lane masks)
// and twice as many, half as long workgroups fill the chip.  conv1 is recomputed on the halo columns as on the halo rows.
// NS = weight-step register slots (prefetch distance NS - 1 steps of 8 KiB per wave): inside a network every block's 2.2 MB
// of weights arrive cold (from the MALL, not the XCD's L2); the column-split form has the registers for a deeper ring.
// HEADC > 0 = the stride-2 HEAD of a stage's entry block (conv1 1x1 + bn1 + relu -> conv2 3x3 / stride 2 + bn2 + relu,
// blocks.py:105-112 with stride on conv2, resnet.py:44-49): x has 64 * HEADC channels, the strip's t1 is kept at full input
// resolution (2 TH + 1 rows), the taps of output pixel (r, c) sit at T1 row (2r + ky) W + 2c - 1 + kx (only the left border
// needs the zero row: W is even), and the kernel ends behind conv2: t2 [N, H/2, W/2, P] leaves through the T2 tile.
// NW = waves per workgroup (round 5).  4: a wave owns TWO output-channel tiles (one wave per SIMD).  8: a wave owns ONE tile and
// two waves share a SIMD: every weight byte still reaches the CU once (the waves split the CHANNEL tiles, not the pixels), but
// while one wave of a SIMD is held at its issue port by its weight loads (a buffer_load_b128 holds it ~40 cycles: a step of the
// 4-wave form costs its MFMA time + ~335 cycles for its eight loads, tools/dev/ubench/l2_burst.hip) the other one multiplies.
// FOLD (round 6) = the folded operands: the BatchNorm scales are in the fp16 weights (one rounding, like the plain weights),
// p.tab holds the shifts as (hi, lo) fp16 pairs, uint32 [P + P + C].  A shift enters its accumulator as ONE extra MFMA per tile (A row
// = {hi, lo, 0 ..}, B = ones at k = 0, 1), the identity residual as TWO (A = the 0/1 matrix that maps the registers the residual was
// picked up in — 16 consecutive channels per lane — onto the tile's MFMA rows), and every epilogue shrinks to fp16(relu(acc)):
// per quarter of phase 3 ~190 vector instructions + 32 table reads become 12 MFMAs + 64 (ft_common.h: relu_acc16).
template <int MT1, int MT2, bool XH, int NS, int HEADC = 0, int NW = 4, bool FOLD = false>
__global__ __launch_bounds__(64 * NW, NW / 4) void bottleneck_stream_direct_kernel(const BnsParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int P = 256;
  using G = BnsGeom<P>;
  constexpr bool HEAD = HEADC > 0;
  static_assert(!(HEAD && XH), "the head form uses full-width strips");
  constexpr int NCT = G::NCT, NC1 = HEAD ? HEADC : G::NC1, KC = G::KC, WSTEP = G::WSTEP, ROWB = G::ROWB;
  constexpr int G2 = NC1, G3 = G2 + 9 * KC, GEND = HEAD ? G3 : G3 + 4 * KC;
  static_assert(NW == 4 || NW == 8, "waves per workgroup");
  constexpr int CTW = 8 / NW, NT = 64 * NW;                  // output-channel tiles per wave, threads
  constexpr int XROWS = MT1 * 32, LX = (MT1 * 4 + NW - 1) / NW;   // x-chunk wave loads (8 rows each) per wave
  static_assert(LX * NW * 8 * 128 <= G::XSTRIDE, "x-chunk buffer");
  constexpr int NOUT = MT2 * 32;
  constexpr int ZROW = 61440, TABS = 65536, STG = 81920, STGB = NOUT * ROWB;   // [76 K, 77 K): scratch of the L2 touch loads
  constexpr int LDS_BYTES = STG + 2 * STGB;
  static_assert(XROWS * 128 <= G::XSTRIDE && NOUT * ROWB <= 61440 && LDS_BYTES <= 163840, "LDS map");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  using c0 = std::integral_constant<int, 0>;
  using c1 = std::integral_constant<int, 1>;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wcol = wave;
  const int l31 = lane & 31, lhi = lane >> 5;

  int logical;
  {
    const int b = blockIdx.x;
    const int q = p.total >> 3, r = p.total & 7, xcd = b & 7, loc = b >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int n = logical / p.ppi;
  const int W = p.W;
  const int TWc = XH ? p.TWc : (HEAD ? p.Wo : W);   // output columns of this workgroup; output pixel m = r * TWc + c
  const int PW = XH ? TWc + 2 : W;            // row pitch of the x-chunk / T1 patch
  int y0, x0 = 0;
  {
    const int li = logical - n * p.ppi;
    if constexpr (XH) {
      const int st = li / p.csplit;
      x0 = (li - st * p.csplit) * TWc;
      y0 = st * p.TH;
    } else {
      y0 = li * p.TH;
    }
  }
  const int Hout = HEAD ? p.Ho : p.H;
  const int rows_out = Hout - y0 < p.TH ? Hout - y0 : p.TH;
  const int cols_out = HEAD ? TWc : (W - x0 < TWc ? W - x0 : TWc);
  const int npix_out = rows_out * TWc;
  const int npix_halo = (HEAD ? 2 * p.TH + 1 : p.TH + 2) * PW;
  const int iy0 = HEAD ? 2 * y0 - 1 : y0 - 1;          // input row of the patch's first row

  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.ws), 0, p.ws_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.tab), 0, FOLD ? (2 * P + G::C) * 4 : 6 * G::TABB, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;

  unsigned x_voff[LX];
#pragma unroll
  for (int t = 0; t < LX; ++t) {
    const int hp = (t * NW + wave) * 8 + (lane >> 3);
    const int hr = hp / PW, hc = hp - hr * PW;
    const int iy = iy0 + hr, ix = XH ? x0 - 1 + hc : hc;
    unsigned v = kOOB;
    if (hp < npix_halo && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)W)
      v = (unsigned)((((n * p.H + iy) * W + ix) * p.x_cstride + p.x_coff) * 2 + (((lane & 7) ^ BNS_XKEY(hp)) << 4));
    x_voff[t] = v;
  }
  const unsigned lane16 = (unsigned)lane * 16u;
  auto issue_x = [&](int c, int buf) {
    char* dst = smem + buf * G::XSTRIDE;
#pragma unroll
    for (int t = 0; t < LX; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr)(dst + (t * NW + wave) * 1024), 16, x_voff[t], c * 128, 0, 0);
  };
  // the weight fragments of step g for this wave: (kk, tile 2*wcol + i) at g * WSTEP + (kk * NCT + 2*wcol + i) KiB
  static_assert(NS >= 3 && 12 % NS == 0, "ring phase of the unrolled phase-2 body (12 steps per kernel row)");
  constexpr int D = NS - 1;
  uint4_t areg[NS][4][CTW];
  auto load_a = [&](auto slotc, int g) {        // past the end of the stream: out of range, zeros, never multiplied
    constexpr int SL = decltype(slotc)::value;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < CTW; ++i) {
#if FT_BNSD_ABL & 32
        if (g >= NC1) { uint4_t z{0u, 0u, 0u, 0u}; asm volatile("" : "+v"(z)); areg[SL][kk][i] = z; continue; }
#endif
        areg[SL][kk][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16, g * WSTEP + (kk * NCT + CTW * wcol + i) * 1024, 0);
      }
  };

  auto load_a_half = [&](auto slotc, int g, auto halfc) {   // K16 slices {0, 1} or {2, 3} of the step
    constexpr int SL = decltype(slotc)::value, HF = decltype(halfc)::value;
#pragma unroll
    for (int kk = 2 * HF; kk < 2 * HF + 2; ++kk)
#pragma unroll
      for (int i = 0; i < CTW; ++i) {
#if FT_BNSD_ABL & 16
        { uint4_t z{0u, 0u, 0u, 0u}; asm volatile("" : "+v"(z)); areg[SL][kk][i] = z; continue; }
#endif
        areg[SL][kk][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16, g * WSTEP + (kk * NCT + CTW * wcol + i) * 1024, 0);
      }
  };
  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define BNSD_TS(i) do { if (p.dbg & 32) ts[i] = __builtin_amdgcn_s_memtime(); } while (0)
  BNSD_TS(0);
  // Inside a network the block's weights are not in this XCD's L2 when the kernel starts, and every workgroup of the XCD
  // asks for the same lines at the same moment: the stream then costs 17 us of a 46-us block (FT_BNS_DBG=64 in situ).  The
  // first round of workgroups on an XCD therefore TOUCHES the whole stream once, each its own 1/n-th (one dword per
  // 128-byte line, results discarded): the lines are on their way into the L2 before the lock-step demand loads reach them.
  // (as LDS-DMA into a scratch corner: no destination register whose reuse the compiler would have to guard.)  Round 6: EXACTLY
  // kTouch loads per thread (lines past the share: out of range, no traffic), issued BEHIND x chunk 0 and the weights of step 0 —
  // the touch used to lead every wave's in-order load queue, so the first chunk waited for 100 KB of line fetches per CU (start-up
  // 5.4 k cycles of a 71 k-cycle workgroup, tools/dev/bns_phases.py) — and counted by the first hand-counted wait.
  constexpr int kTouch = FT_BNS_L2_TOUCH ? 6 : 0;
  auto issue_touch = [&]() {
#if FT_BNS_L2_TOUCH
    const bool on = blockIdx.x < 256 && !(p.dbg & 512);
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int first = p.total < 256 ? p.total : 256;
    const int nloc = (first - xcd + 7) >> 3;                       // workgroups of the first round on this XCD
    const unsigned lines = (p.ws_bytes + 127u) >> 7;
    const unsigned per = (lines + nloc - 1) / nloc;
    const unsigned lo = loc * per, hi = lo + per < lines ? lo + per : lines;
#pragma unroll
    for (int k = 0; k < kTouch; ++k) {
      const unsigned l = lo + tid + (unsigned)(NT * k);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr)(smem + 77824 + wave * 256), 4, (on && l < hi) ? l << 7 : kOOB, 0, 0, 0);
    }
#endif
  };
  // prologue: all six tables (12 KiB, 3 x 256-byte pieces per wave ... 48 pieces), x chunks 0..2, weights of steps 0 and 1, zero row
  // folded form: no tables in LDS; the shift pair of MFMA row l31 of every (epilogue, channel tile) of this wave sits in a register
  // (lanes 32..63 hold k = 8..15 of the shift slice: zeros, fetched out of range), and the two residual slices' 0/1 matrix is built here:
  // accumulator register k of lane (pixel, half) is channel 16 half + k (bns_sigma), the residual registers hold channels
  // 16 half + 8 h + e as element e of slice h, which the MFMA reads as k = 8 half + e: row r takes slice (sigma(r) >> 3) & 1, k = 8 (sigma(r) >> 4) + (sigma(r) & 7)
  [[maybe_unused]] unsigned shp[6][CTW];
  [[maybe_unused]] uint4_t permA[2];
  [[maybe_unused]] const uint4_t onesB = uint4_t{0x3C003C00u, 0u, 0u, 0u};
  if constexpr (FOLD) {
    const int sg = bns_sigma(l31);
    const bool on = lhi == (sg >> 4);
    const int ph = (sg >> 3) & 1, pe = sg & 7;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      unsigned w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = (on && ph == hh && (pe >> 1) == k) ? (0x3C00u << (16 * (pe & 1))) : 0u;
      permA[hh] = uint4_t{w[0], w[1], w[2], w[3]};
    }
  }
  constexpr int kTabLoads = FOLD ? 6 * CTW : 48 / NW;
  auto load_tables = [&]() {
    if constexpr (FOLD) {
      const unsigned so = lhi == 0 ? 4u * (unsigned)bns_sigma(l31) : kOOB;
#pragma unroll
      for (int e = 0; e < 6; ++e)
#pragma unroll
        for (int i = 0; i < CTW; ++i)
          shp[e][i] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_t, so, 4 * (e * P + (CTW * wcol + i) * 32), 0);
    } else {
#pragma unroll
      for (int t = 0; t < 48 / NW; ++t)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_t, (lds_ptr)(smem + TABS + (t * NW + wave) * 256), 4, (unsigned)lane * 4u, (t * NW + wave) * 256, 0, 0);
    }
  };
  [[maybe_unused]] auto add_shift = [&](int e, auto& A, auto mtc) {      // A[i][j] += shift of epilogue e (one MFMA per tile)
    constexpr int MT = decltype(mtc)::value;
#pragma unroll
    for (int i = 0; i < CTW; ++i) {
      const uint4_t sa = uint4_t{shp[e][i], 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < MT; ++j)
        A[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, sa), __builtin_bit_cast(half8_t, onesB), A[i][j], 0, 0, 0);
    }
  };
  // order of every wave's (in-order) load queue: x chunk 0, the weights of step 0, the L2 touch, the tables / shift pairs, x chunks 1
  // and 2, the weights of steps 1 .. D-1
#if FT_BNS_TOUCH_FIRST
  issue_touch();
  load_tables();
  asm volatile("" ::: "memory");
  issue_x(0, 0);
  load_a(std::integral_constant<int, 0>{}, 0);
#else
  issue_x(0, 0);
  load_a(std::integral_constant<int, 0>{}, 0);
  issue_touch();
  load_tables();
  asm volatile("" ::: "memory");
#endif
  issue_x(1, 1);
  issue_x(2, 2);
  bns_unroll<D - 1>([&](auto sc) { load_a(std::integral_constant<int, decltype(sc)::value + 1>{}, decltype(sc)::value + 1); });
  if (tid < ROWB / 16) *reinterpret_cast<uint4_t*>(smem + ZROW + tid * 16) = uint4_t{0u, 0u, 0u, 0u};

  uint4_t res[4][CTW][MT2][2];
  int m_out[MT2], hp_out[MT2];          // output pixel of the lane in tile j, and its index in the halo patch
#pragma unroll
  for (int j = 0; j < MT2; ++j) {
    m_out[j] = j * 32 + l31;
    if constexpr (XH) {
      const int r = m_out[j] / TWc;
      hp_out[j] = (r + 1) * PW + (m_out[j] - r * TWc) + 1;
    } else if constexpr (HEAD) {
      const int r = m_out[j] / TWc;
      hp_out[j] = (2 * r + 1) * W + 2 * (m_out[j] - r * TWc);      // the pixel's centre tap in the T1 patch
    } else {
      hp_out[j] = m_out[j] + W;
    }
  }

  // ================= phase 1 ============================================================================================
  float16_t acc1[CTW][MT1];
#pragma unroll
  for (int i = 0; i < CTW; ++i)
#pragma unroll
    for (int j = 0; j < MT1; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[i][j][r] = 0.f;
  int b1_off[MT1];
#pragma unroll
  for (int j = 0; j < MT1; ++j) {
    const int hp = j * 32 + l31;
    b1_off[j] = hp * 128 + ((lhi ^ BNS_XKEY(hp)) << 4);
  }
  {
    uint4_t fx[2][MT1];
    auto ldx = [&](auto setc, int buf, int kk) {
      constexpr int S = decltype(setc)::value;
      const char* xb = smem + buf * G::XSTRIDE;
#pragma unroll
      for (int j = 0; j < MT1; ++j) {
#if FT_BNSD_ABL & 8
        { uint4_t z{0u, 0u, 0u, 0u}; asm volatile("" : "+v"(z)); fx[S][j] = z; continue; }
#endif
        fx[S][j] = *reinterpret_cast<const uint4_t*>(xb + (b1_off[j] ^ (kk << 5)));
      }
    };
    auto mma1 = [&](auto setc, auto slotc, auto kkc) {
      constexpr int S = decltype(setc)::value, SL = decltype(slotc)::value, kk = decltype(kkc)::value;
#pragma unroll
      for (int i = 0; i < CTW; ++i)
#pragma unroll
        for (int j = 0; j < MT1; ++j)
          acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, areg[SL][kk][i]),
                                                              __builtin_bit_cast(half8_t, fx[S][j]), acc1[i][j], 0, 0, 0);
    };
    // x chunk 0 has landed (this wave's share): behind it chunks 1, 2 and the two weight steps (8 loads each) may fly
    static_assert(2 * LX + 4 * CTW * D + kTouch + kTabLoads <= 63, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * LX + 4 * CTW * D + (FT_BNS_TOUCH_FIRST ? 0 : kTouch + kTabLoads)) : "memory");
    BNS_BARRIER();
    BNSD_TS(7);             // start-up: x chunk 0 of every wave has landed
    ldx(c0{}, 0, 0);
    bns_unroll<NC1>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      constexpr int buf = c % 3;
      using slot = std::integral_constant<int, c % NS>;
      // The wave-uniform branch below (residual pick-up) splits the chunk into two scheduling regions; each gets half of the
      // step's weight loads and a pinned issue order (see dstep): one vector-memory load and the LDS reads behind every two
      // MFMAs instead of hipcc's clusters of 6-8 loads with the matrix pipe drained behind them.
      load_a_half(std::integral_constant<int, (c + D) % NS>{}, c + D, c0{});
      ldx(c1{}, buf, 1);
      mma1(c0{}, slot{}, std::integral_constant<int, 0>{});
      ldx(c0{}, buf, 2);
      if constexpr ((FT_BNS_PIN & 1) && NW == 4) {
        // region: [x chunk c+2 DMA, slice-0 reads of this chunk, slice 3 of chunk c-1] (behind the last barrier) + the above
        __builtin_amdgcn_sched_group_barrier(0x020, LX, 0);
        bns_unroll<3>([&](auto) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, (2 * MT1 + 2) / 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        });
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (MT1 + 2) / 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        bns_unroll<MT1 - 1>([&](auto) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, (MT1 + 2) / 3, 0);
        });
      }
      if (!HEAD && !(FT_BNSD_ABL & 2) && c % 4 == (CTW * wcol) / 2) {  // this chunk holds the channels of this wave column for quarter c / 4
        constexpr int q = c / 4;
        const char* xb = smem + buf * G::XSTRIDE;
#pragma unroll
        for (int j = 0; j < MT2; ++j) {
          const int hp = hp_out[j];
          const char* rowp = xb + hp * 128;
#pragma unroll
          for (int i = 0; i < CTW; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
              res[q][i][j][h] = *reinterpret_cast<const uint4_t*>(rowp + (((4 * ((CTW * wcol + i) & 1) + 2 * lhi + h) ^ BNS_XKEY(hp)) << 4));
        }
      }
      mma1(c1{}, slot{}, std::integral_constant<int, 1>{});
      load_a_half(std::integral_constant<int, (c + D) % NS>{}, c + D, c1{});
      ldx(c1{}, buf, 3);
      mma1(c0{}, slot{}, std::integral_constant<int, 2>{});
      if constexpr ((FT_BNS_PIN & 1) && NW == 4) {
        bns_unroll<MT1>([&](auto) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        });
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 4 - MT1 > 0 ? 4 - MT1 : 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT1 - 2 * MT1 - 2, 0);
      }
      if constexpr (c + 1 < NC1) {
        // x chunk c+1 has landed and every read of chunk c's buffer is complete: refill it with chunk c+3.  Younger than x
        // chunk c+1 are the weights of step c+1 (needed next anyway), x chunk c+2 and the weights of step c+2.
        // (with a prefetch distance of three or more steps the weights of step c+1 are OLDER than x chunk c+1: two weight
        // steps may stay in flight)
#if FT_BNSD_ABL & 4
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#elif !(FT_BNSD_ABL & 16)
        if constexpr (c + 2 < NC1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LX + (D == 2 ? 4 * CTW : 8 * CTW)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(D == 2 ? 4 * CTW : 8 * CTW) : "memory");
#else
        if constexpr (c + 2 < NC1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LX) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
#if !(FT_BNSD_ABL & 1)
        BNS_BARRIER();
#endif
        if constexpr (c + 3 < NC1 && !(FT_BNSD_ABL & 4)) issue_x(c + 3, buf);
        ldx(c0{}, (c + 1) % 3, 0);
      }
      mma1(c1{}, slot{}, std::integral_constant<int, 3>{});
    });
  }
  BNSD_TS(1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  BNS_BARRIER();          // every wave is past its last x-chunk read: the x buffers become T1
  {
    if constexpr (FOLD) add_shift(0, acc1, std::integral_constant<int, MT1>{});
    [[maybe_unused]] const float* tb = reinterpret_cast<const float*>(smem + TABS);
#pragma unroll
    for (int i = 0; i < CTW; ++i) {
      [[maybe_unused]] const int ch = (CTW * wcol + i) * 32 + 16 * lhi;
      [[maybe_unused]] float4_t sc[4], sh[4];
      if constexpr (!FOLD) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          sc[g4] = *reinterpret_cast<const float4_t*>(tb + ch + g4 * 4);
          sh[g4] = *reinterpret_cast<const float4_t*>(tb + P + ch + g4 * 4);
        }
      }
#pragma unroll
      for (int j = 0; j < MT1; ++j) {
        const int hp = j * 32 + l31;
        const int hr = hp / PW;
        const int iy = iy0 + hr, ix = XH ? x0 - 1 + (hp - hr * PW) : 0;
        const bool inside = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)W;
        half8_t h8[2];
        if constexpr (FOLD) relu_acc16(acc1[i][j], h8);
        else bn_relu_acc16(acc1[i][j], sc, sh, h8);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          uint4_t u = __builtin_bit_cast(uint4_t, h8[h]);
          u.x = inside ? u.x : 0u; u.y = inside ? u.y : 0u; u.z = inside ? u.z : 0u; u.w = inside ? u.w : 0u;
          h8[h] = __builtin_bit_cast(half8_t, u);
        }
        if (hp < npix_halo) {
          char* rowp = smem + hp * ROWB;
          const int cb = (CTW * wcol + i) * 4 + 2 * lhi;
#pragma unroll
          for (int h = 0; h < 2; ++h)
            *reinterpret_cast<half8_t*>(rowp + (((cb + h) ^ (hp & 15)) << 4)) = h8[h];
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  BNS_BARRIER();          // T1 complete
  BNSD_TS(2);

  // ================= phases 2 + 3 =======================================================================================
  float16_t acc[CTW][MT2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < CTW; ++i)
#pragma unroll
      for (int j = 0; j < MT2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  zero_acc();
  int edge[MT2];
#pragma unroll
  for (int j = 0; j < MT2; ++j) {
    const int ox = m_out[j] % TWc;
    edge[j] = XH ? 0 : (HEAD ? (ox == 0 ? 1 : 0) : (ox == 0 ? 1 : 0) | (ox == W - 1 ? 2 : 0));
  }
  // T1 / T2 row of the pixel operand: `off` relative to the lane's own row; full-width strips mask the two x-border taps
  // (bad), the column-split form reads its zeroed halo columns instead
  auto row_bases = [&](int off, int kc, int bad, int (&rb)[MT2], bool t1) {
#pragma unroll
    for (int j = 0; j < MT2; ++j) {
      const int row = ((XH && t1) ? hp_out[j] - PW - 1 : ((HEAD && t1) ? hp_out[j] - W - 1 : m_out[j])) + off;
      const int v = row * ROWB + (((row & 15) ^ lhi) << 4);
      rb[j] = ((edge[j] & bad) ? ZROW + (lhi << 4) : v) ^ (kc << 7);
    }
  };
  auto tap_off = [&](int ky, int kx) { return (XH || HEAD) ? ky * PW + kx : ky * W + kx - 1; };
  uint4_t fb[2][MT2];
  auto ldb = [&](auto setc, int kk, const int (&rb)[MT2]) {
    constexpr int S = decltype(setc)::value;
#pragma unroll
    for (int j = 0; j < MT2; ++j) {
#if FT_BNSD_ABL & 64
      { uint4_t z{0u, 0u, 0u, 0u}; asm volatile("" : "+v"(z)); fb[S][j] = z; continue; }
#endif
      fb[S][j] = *reinterpret_cast<const uint4_t*>(smem + (rb[j] ^ (kk << 5)));
    }
  };
  auto mma2 = [&](auto setc, auto slotc, auto kkc, float16_t (&A)[CTW][MT2]) {
    constexpr int S = decltype(setc)::value, SL = decltype(slotc)::value, kk = decltype(kkc)::value;
#pragma unroll
    for (int i = 0; i < CTW; ++i)
#pragma unroll
      for (int j = 0; j < MT2; ++j)
        A[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, areg[SL][kk][i]),
                                                         __builtin_bit_cast(half8_t, fb[S][j]), A[i][j], 0, 0, 0);
  };
  // one weight step g (ring slot SL = g % 3): slice 0 of the pixel operand already sits in register set 0; `rbn` = row
  // bases of the next step; `extra` = independent vector work that rides along (phase 3: a piece of the previous
  // quarter's epilogue), NX = how many of its VALU instructions each of the eight issue groups takes
  auto dstep = [&](auto slotc, int g, const int (&rb)[MT2], bool has_next, const int (&rbn)[MT2], float16_t (&A)[CTW][MT2], auto nxc,
                   auto&& extra) {
    constexpr int SL = decltype(slotc)::value, NX = decltype(nxc)::value;
    using slot = std::integral_constant<int, SL>;
    load_a(std::integral_constant<int, (SL + D) % NS>{}, g + D);
    ldb(c1{}, 1, rb);
    mma2(c0{}, slot{}, std::integral_constant<int, 0>{}, A);
    extra();
    ldb(c0{}, 2, rb);
    mma2(c1{}, slot{}, std::integral_constant<int, 1>{}, A);
    ldb(c1{}, 3, rb);
    mma2(c0{}, slot{}, std::integral_constant<int, 2>{}, A);
    if (has_next) ldb(c0{}, 0, rbn);
    mma2(c1{}, slot{}, std::integral_constant<int, 3>{}, A);
    // Issue order of the step: one weight load and the pixel-operand reads after every MT2 MFMAs.  A buffer_load_b128 holds
    // the wave's issue port for ~40 cycles (tools/dev/ubench/l2_burst.hip: a step costs its compute time + 335 cycles for its
    // eight loads, whatever the prefetch depth); clustered as hipcc places them, the matrix pipe drains behind them.
    if constexpr ((FT_BNS_PIN & 1) && NW == 4) {
      bns_unroll<8>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        __builtin_amdgcn_sched_group_barrier(0x008, MT2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (i & 1) ? (MT2 + 1) / 2 : MT2 / 2, 0);
        if constexpr (NX > 0) {
          __builtin_amdgcn_sched_group_barrier(0x002, NX, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, FOLD ? 0 : 1, 0);      // the folded-BN table reads of the piece
          __builtin_amdgcn_sched_group_barrier(0x200, (i & 3) == 3 ? 1 : 0, 0);
        }
      });
    }
  };
  using nx0 = std::integral_constant<int, 0>;
  auto no_extra = [] {};

  // ---- phase 2: nine taps x KC chunks, three taps per loop trip (12 steps: a multiple of the register ring period) -------
  {
    static_assert((3 * KC) % NS == 0, "ring phase of the unrolled body");
    int rb[MT2], rbn[MT2];
    row_bases(tap_off(0, 0), 0, 1, rb, true);
    ldb(c0{}, 0, rb);
    for (int ky = 0; ky < 3; ++ky) {
      bns_unroll<3 * KC>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int kx = s / KC, kc = s % KC;
        constexpr int nkc = (kc + 1) % KC, nkx = kc + 1 == KC ? (kx + 1) % 3 : kx;
        const int nky = (kc + 1 == KC && kx == 2) ? ky + 1 : ky;
        row_bases(tap_off(nky, nkx), nkc, nkx == 0 ? 1 : (nkx == 2 ? 2 : 0), rbn, true);
        dstep(std::integral_constant<int, (G2 + s) % NS>{}, G2 + 3 * KC * ky + s, rb, !(ky == 2 && s == 3 * KC - 1), rbn, acc, nx0{}, no_extra);
#pragma unroll
        for (int j = 0; j < MT2; ++j) rb[j] = rbn[j];
      });
    }
  }
  BNSD_TS(3);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  BNS_BARRIER();          // every wave is past its last T1 read: T2 overwrites T1
  {
    if constexpr (FOLD) add_shift(1, acc, std::integral_constant<int, MT2>{});
    [[maybe_unused]] const float* tb = reinterpret_cast<const float*>(smem + TABS + G::TABB);
#pragma unroll
    for (int i = 0; i < CTW; ++i) {
      [[maybe_unused]] const int ch = (CTW * wcol + i) * 32 + 16 * lhi;
      [[maybe_unused]] float4_t sc[4], sh[4];
      if constexpr (!FOLD) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          sc[g4] = *reinterpret_cast<const float4_t*>(tb + ch + g4 * 4);
          sh[g4] = *reinterpret_cast<const float4_t*>(tb + P + ch + g4 * 4);
        }
      }
#pragma unroll
      for (int j = 0; j < MT2; ++j) {
        const int m = m_out[j];
        half8_t h8[2];
        if constexpr (FOLD) relu_acc16(acc[i][j], h8);
        else bn_relu_acc16(acc[i][j], sc, sh, h8);
        char* rowp = smem + m * ROWB;
        const int cb = (CTW * wcol + i) * 4 + 2 * lhi;
#pragma unroll
        for (int h = 0; h < 2; ++h) *reinterpret_cast<half8_t*>(rowp + (((cb + h) ^ (m & 15)) << 4)) = h8[h];
      }
    }
  }
  zero_acc();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  BNS_BARRIER();          // T2 complete
  BNSD_TS(4);
  if constexpr (HEAD) {
    // the head form ends here: the T2 tile (rows = output pixels, 16-byte chunk ^= row & 15) leaves as whole lines
    constexpr int CPRH = ROWB / 16, NSTH = NOUT * CPRH / NT;
#pragma unroll
    for (int k = 0; k < NSTH; ++k) {
      const int idx = tid + NT * k, m = idx / CPRH, ch = idx % CPRH;
      const int r = m / TWc, c = m - r * TWc;
      const unsigned vo = m < npix_out ? (unsigned)((((n * p.Ho + y0 + r) * p.Wo + c) * p.y_cstride + p.y_coff + ch * 8) * 2) : kOOB;
      const uint4_t v = *reinterpret_cast<const uint4_t*>(smem + m * ROWB + ((ch ^ (m & 15)) << 4));
      if (!(p.dbg & 4)) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_y, vo, 0, FT_YSTORE_BUF_AUX);
    }
    return;
  }

  // ---- phase 3: four quarters of P output channels; the tile of a quarter leaves through one of two LDS staging tiles ----
  {
    // WSTG: the wave's own [NOUT pixels][128 bytes = its 64 channels of the quarter] tile, 16-byte chunk c of pixel m at chunk
    // c ^ (m & 7): the eight lanes of a ds_write_b128 group (eight pixels, one chunk) and the sixteen of a ds_read_b128 group
    // (two pixels x eight chunks ...) cover distinct bank groups.  Read-out: lane -> pixel lane / 8 + 8 k, chunk lane % 8.
    constexpr bool WSTG = FT_BNS_WSTG && FT_BNS_STG && CTW == 2;
    constexpr int CPR = ROWB / 16, NSTG = WSTG ? NOUT / 8 : NOUT * CPR / NT;
    unsigned s_voff[NSTG];
    int s_off[NSTG];
#pragma unroll
    for (int k = 0; k < NSTG; ++k) {
      if constexpr (WSTG) {
        const int m = (lane >> 3) + 8 * k, ch = lane & 7;
        const int r = m / TWc, c = m - r * TWc;
        s_voff[k] = (m < npix_out && c < cols_out)
                        ? (unsigned)((((n * p.H + y0 + r) * W + x0 + c) * p.y_cstride + p.y_coff + wcol * 64 + ch * 8) * 2) : kOOB;
        s_off[k] = wave * (NOUT * 128) + m * 128 + ((ch ^ (m & 7)) << 4);
      } else {
        const int idx = tid + NT * k, m = idx / CPR, ch = idx % CPR;
        const int r = m / TWc, c = m - r * TWc;
        s_voff[k] = (m < npix_out && c < cols_out) ? (unsigned)((((n * p.H + y0 + r) * W + x0 + c) * p.y_cstride + p.y_coff + ch * 8) * 2) : kOOB;
        s_off[k] = m * ROWB + ((ch ^ (m & 15)) << 4);
      }
    }
    int rb[MT2], rbn[MT2];
    row_bases(0, 0, 0, rb, false);
    ldb(c0{}, 0, rb);
    // epilogue of quarter q, tile (i, j): folded BN + residual + ReLU -> fp16 -> staging tile q & 1
    auto epi_piece = [&](auto qc, int i, int j, float16_t (&A)[CTW][MT2]) {
      constexpr int q = decltype(qc)::value;
      char* stg = smem + STG + (q & 1) * STGB;
      [[maybe_unused]] const float* tb = reinterpret_cast<const float*>(smem + TABS + (2 + q) * G::TABB);
      [[maybe_unused]] const int ch = (CTW * wcol + i) * 32 + 16 * lhi;
      [[maybe_unused]] float4_t sc[4], sh[4];
      [[maybe_unused]] half8_t o8[2];
      if constexpr (FOLD) {
        relu_acc16(A[i][j], o8);
      } else {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          sc[g4] = *reinterpret_cast<const float4_t*>(tb + ch + g4 * 4);
          sh[g4] = *reinterpret_cast<const float4_t*>(tb + P + ch + g4 * 4);
        }
      }
      const int m = m_out[j];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        half8_t o;
        if constexpr (FOLD) o = o8[h];
        else o = bn_res_relu_acc8(A[i][j], h, sc, sh, __builtin_bit_cast(half8_t, res[q][i][j][h]));
#if FT_BNS_STG
        if constexpr (WSTG) *reinterpret_cast<half8_t*>(stg + wave * (NOUT * 128) + m * 128 + (((i * 4 + 2 * lhi + h) ^ (m & 7)) << 4)) = o;
        else *reinterpret_cast<half8_t*>(stg + m * ROWB + ((((CTW * wcol + i) * 4 + 2 * lhi + h) ^ (m & 15)) << 4)) = o;
#else
        // straight from the accumulator layout: the lane's 16 consecutive channels = two adjacent 16-byte stores
        const int orow = m / TWc, ocol = m - orow * TWc;
        const unsigned vo = (m < npix_out && ocol < cols_out) ? (unsigned)((((n * p.H + y0 + orow) * W + x0 + ocol) * p.y_cstride + p.y_coff + ch + 8 * h) * 2) : kOOB;
        if (!(p.dbg & 4)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, o), rsrc_y, vo + (unsigned)(q * P * 2), 0, FT_BNS_DIRECT_AUX);
#endif
      }
    };
    // folded form: shift3 of the quarter and the residual join tile (i, j)'s accumulator as three MFMAs
    [[maybe_unused]] auto fold_tail = [&](auto qc, int i, int j, float16_t (&A)[CTW][MT2]) {
      constexpr int q = decltype(qc)::value;
      const uint4_t sa = uint4_t{shp[2 + q][i], 0u, 0u, 0u};
      A[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, sa), __builtin_bit_cast(half8_t, onesB), A[i][j], 0, 0, 0);
#pragma unroll
      for (int h = 0; h < 2; ++h)
        A[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, permA[h]), __builtin_bit_cast(half8_t, res[q][i][j][h]), A[i][j], 0, 0, 0);
    };
    auto readout = [&](auto qc) {
      constexpr int q = decltype(qc)::value;
      const char* stg = smem + STG + (q & 1) * STGB;
#if !FT_BNS_STG
      return;
#endif
#pragma unroll
      for (int k = 0; k < NSTG; ++k) {
        const uint4_t v = *reinterpret_cast<const uint4_t*>(stg + s_off[k]);
        if (!(p.dbg & 4)) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_y, s_voff[k] + (unsigned)(q * P * 2), 0, FT_YSTORE_BUF_AUX);
      }
    };
    [[maybe_unused]] auto zero_set = [&](float16_t (&A)[CTW][MT2]) {
#pragma unroll
      for (int i = 0; i < CTW; ++i)
#pragma unroll
        for (int j = 0; j < MT2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) A[i][j][r] = 0.f;
    };
    constexpr bool OVL = FT_BNS_OVL && CTW * MT2 == KC;     // one epilogue tile per weight step
    if constexpr (OVL) {
    // Overlapped form: two accumulator sets alternate between the quarters; the epilogue of quarter q-1 (VALU + LDS writes,
    // ~3.4 k cycles when it ran alone behind its quarter) rides inside the weight steps of quarter q, one (i, j) tile per
    // step, pinned between the MFMAs by dstep's issue groups.  The staging tiles alternate as before: tile (q-1) & 1 is
    // written during quarter q, read out behind quarter q's barrier; its previous readers (quarter q-3) are two barriers back.
    float16_t acc_b[CTW][MT2];
    zero_set(acc_b);
    bns_unroll<4>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      float16_t (&A)[CTW][MT2] = (q & 1) ? acc_b : acc;
      float16_t (&Aprev)[CTW][MT2] = (q & 1) ? acc : acc_b;
      bns_unroll<KC>([&](auto kcc) {
        constexpr int kc = decltype(kcc)::value;
        constexpr int g = G3 + q * KC + kc;
        row_bases(0, (kc + 1) % KC, 0, rbn, false);
        if constexpr (q > 0) {
          dstep(std::integral_constant<int, g % NS>{}, g, rb, g + 1 < GEND, rbn, A, std::integral_constant<int, FOLD ? 4 : 12>{},
                [&] {
                  if constexpr (FOLD) fold_tail(std::integral_constant<int, q - 1>{}, kc / MT2, kc % MT2, Aprev);
                  epi_piece(std::integral_constant<int, q - 1>{}, kc / MT2, kc % MT2, Aprev);
                });
        } else {
          dstep(std::integral_constant<int, g % NS>{}, g, rb, g + 1 < GEND, rbn, A, nx0{}, no_extra);
        }
#pragma unroll
        for (int j = 0; j < MT2; ++j) rb[j] = rbn[j];
      });
      if constexpr (q > 0) {
        zero_set(Aprev);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        BNS_BARRIER();
        readout(std::integral_constant<int, q - 1>{});
      }
    });
    // quarter 3 has no successor to hide behind
#pragma unroll
    for (int i = 0; i < CTW; ++i)
#pragma unroll
      for (int j = 0; j < MT2; ++j) {
        if constexpr (FOLD) fold_tail(std::integral_constant<int, 3>{}, i, j, acc_b);
        epi_piece(std::integral_constant<int, 3>{}, i, j, acc_b);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    BNS_BARRIER();
    readout(std::integral_constant<int, 3>{});
    } else {
    bns_unroll<4>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      bns_unroll<KC>([&](auto kcc) {
        constexpr int kc = decltype(kcc)::value;
        constexpr int g = G3 + q * KC + kc;
        row_bases(0, (kc + 1) % KC, 0, rbn, false);
        dstep(std::integral_constant<int, g % NS>{}, g, rb, g + 1 < GEND, rbn, acc, nx0{}, no_extra);
#pragma unroll
        for (int j = 0; j < MT2; ++j) rb[j] = rbn[j];
      });
      if constexpr (FOLD) {
#pragma unroll
        for (int i = 0; i < CTW; ++i)
#pragma unroll
          for (int j = 0; j < MT2; ++j) fold_tail(qc, i, j, acc);
      }
#pragma unroll
      for (int i = 0; i < CTW; ++i)
#pragma unroll
        for (int j = 0; j < MT2; ++j) epi_piece(qc, i, j, acc);
      zero_acc();
#if FT_BNS_STG
      // the two staging tiles alternate: a tile's previous readers (quarter q-2) are two barriers behind
      // (wave-private pieces: a wave reads back only what it wrote itself, its LDS operations execute in order: no barrier)
      if constexpr (!WSTG) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        BNS_BARRIER();
      } else {
        asm volatile("" ::: "memory");       // compiler fence only: the read-out must stay behind the tile's writes in program order
      }
      readout(qc);
#endif
    });
    }
  }
  if (p.dbg & 32) {
    ts[5] = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts[6] = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.y + ((size_t)((n * p.H + y0) * W) * p.y_cstride + p.y_coff) * 2);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = ts[i];
    }
  }
#endif
}

// ---- weight stream packing -------------------------------------------------------------------------------------------
// w1 [P][C], w2 [P][9P] (k = tap * P + ci), w3 [C][P]: the K-major layouts of ft_conv_pack_geometry.  One thread per
// 16-byte piece of the stream: step g, slice kk, channel tile i, lane -> 8 consecutive k of one output channel.
template <int P>
__global__ __launch_bounds__(256) void bns_pack_kernel(const half_t* __restrict__ w1, const half_t* __restrict__ w2,
                                                       const half_t* __restrict__ w3, uint4_t* __restrict__ out) {
  using G = BnsGeom<P>;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= G::GEND * G::WSTEP / 16) return;
  const int lane = idx & 63, frag = idx >> 6;
  const int per_step = 4 * G::NCT;
  const int g = frag / per_step, f = frag - g * per_step;
  const int kk = f / G::NCT, i = f - kk * G::NCT;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int co_t = 32 * i + bns_sigma(l31);
  const int k_t = 16 * kk + 8 * lhi;
  const half_t* src;
  if (g < G::G2) {
    src = w1 + (size_t)co_t * G::C + 64 * g + k_t;
  } else if (g < G::G3) {
    const int s = g - G::G2, tap = s / G::KC, kc = s - tap * G::KC;
    src = w2 + (size_t)co_t * (9 * P) + tap * P + 64 * kc + k_t;
  } else {
    const int s = g - G::G3, q = s / G::KC, kc = s - q * G::KC;
    src = w3 + (size_t)(q * P + co_t) * P + 64 * kc + k_t;
  }
  out[idx] = *reinterpret_cast<const uint4_t*>(src);
}

// the head form's stream (P = 256): NC1 = C / 64 steps of w1 [P][C], then nine taps x KC steps of w2; same piece order
__global__ __launch_bounds__(256) void bns_pack_head_kernel(const half_t* __restrict__ w1, const half_t* __restrict__ w2,
                                                            uint4_t* __restrict__ out, int C) {
  using G = BnsGeom<256>;
  const int nc1 = C / 64, gend = nc1 + 9 * G::KC;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= gend * G::WSTEP / 16) return;
  const int lane = idx & 63, frag = idx >> 6;
  const int per_step = 4 * G::NCT;
  const int g = frag / per_step, f = frag - g * per_step;
  const int kk = f / G::NCT, i = f - kk * G::NCT;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int co_t = 32 * i + bns_sigma(l31);
  const int k_t = 16 * kk + 8 * lhi;
  const half_t* src;
  if (g < nc1) {
    src = w1 + (size_t)co_t * C + 64 * g + k_t;
  } else {
    const int sft = g - nc1, tap = sft / G::KC, kc = sft - tap * G::KC;
    src = w2 + (size_t)co_t * (9 * 256) + tap * 256 + 64 * kc + k_t;
  }
  out[idx] = *reinterpret_cast<const uint4_t*>(src);
}

struct BnsPlan {
  int variant;   // 0: <128,4,3>  1: <256,4,3>  2: <256,3,2>  3: <256,2,1> column-split (direct kernel only)  4: stride-2 head <4,1>
  int TH, ppi;
  int TWc, csplit;
};

static int bns_plan(const ft_bottleneck_desc* d, BnsPlan* out) {
  if (!d) return FT_ERR_INVALID_ARG;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0) return FT_ERR_INVALID_ARG;
  if (d->head_only) {
    // conv1 + conv2 / stride 2 of a 256-plane entry block (layer3.0): 512 input channels, even map, <= 120 T1 rows per strip
    if (d->dtype != FT_F16 || d->projection || d->P != 256 || d->C != 512 || d->stride != 2 || (d->H & 1) || (d->W & 1)) return FT_ERR_UNSUPPORTED;
    if (d->x_coff < 0 || d->y_coff < 0 || d->x_coff % 8 || d->y_coff % 8 || d->x_cstride % 8 || d->y_cstride % 8) return FT_ERR_UNSUPPORTED;
    const int Ho = d->H / 2, Wo = d->W / 2;
    if (d->x_cstride < d->x_coff + d->C || d->y_cstride < d->y_coff + d->P) return FT_ERR_INVALID_ARG;
    if ((long long)d->N * d->H * d->W * d->x_cstride * 2 >= (1LL << 31) || (long long)d->N * Ho * Wo * d->y_cstride * 2 >= (1LL << 31))
      return FT_ERR_UNSUPPORTED;
    int th = (120 / d->W - 1) / 2;
    const int th2 = 32 / Wo;
    th = th < th2 ? th : th2;
    if (th < 1) return FT_ERR_UNSUPPORTED;
    th = th < Ho ? th : Ho;
    th = ceil_div(Ho, ceil_div(Ho, th));
    *out = BnsPlan{4, th, ceil_div(Ho, th), d->W, 1};
    return FT_OK;
  }
  if (d->dtype != FT_F16 || d->projection || (d->P != 128 && d->P != 256) || d->C != 4 * d->P || d->stride > 1) return FT_ERR_UNSUPPORTED;
  if (d->x_coff < 0 || d->y_coff < 0 || d->x_coff % 8 || d->y_coff % 8 || d->x_cstride % 8 || d->y_cstride % 8) return FT_ERR_UNSUPPORTED;
  if (d->x_cstride < d->x_coff + d->C || d->y_cstride < d->y_coff + d->C) return FT_ERR_INVALID_ARG;
  if ((long long)d->N * d->H * d->W * d->x_cstride * 2 >= (1LL << 31) || (long long)d->N * d->H * d->W * d->y_cstride * 2 >= (1LL << 31))
    return FT_ERR_UNSUPPORTED;
  auto rows = [&](int outcap, int halocap) {
    int th = outcap / d->W;
    const int th2 = halocap / d->W - 2;
    th = th < th2 ? th : th2;
    if (th < 1) return 0;
    th = th < d->H ? th : d->H;
    return ceil_div(d->H, ceil_div(d->H, th));     // same strip count, balanced rows
  };
  const int force = getenv("FT_BNS_VARIANT") ? atoi(getenv("FT_BNS_VARIANT")) : -1;   // dev / tests: 1, 2 or 3 (read per call)
  if (d->P == 128) {
    // strips of <= 192 output pixels on <= 256 halo pixels (<128, 4, 3>), or of <= 128 on <= 192 (<128, 3, 2>, round 4) where the
    // large strips leave CUs idle: ResNet-101 at 384 x 288 with 16 crops per GPU has 160 large strips for 256 CUs (48 x 36 maps,
    // 5 rows each) and exactly 256 small ones (3 rows).  Cost = rounds of 256 workgroups x MFMAs per wave: 32 K16 steps of conv1
    // on MT1 halo tiles + (72 + 32) steps on MT2 output tiles, per pair of channel tiles.
    const int th_b = rows(192, 256), th_s = rows(128, 192);
    if (th_b < 1 && th_s < 1) return FT_ERR_UNSUPPORTED;
    const int force128 = getenv("FT_BNS_VARIANT128") ? atoi(getenv("FT_BNS_VARIANT128")) : 0;   // dev / tests: 1 large, 2 small (read per call)
    auto cost = [&](int th, int mt1, int mt2) {
      const long long wg = (long long)d->N * ceil_div(d->H, th);
      return ((wg + 255) / 256) * (long long)(32 * mt1 + 104 * mt2);
    };
    bool small = th_b < 1 || (th_s >= 1 && cost(th_s, 3, 2) < cost(th_b, 4, 3));
    if (force128 == 1 && th_b >= 1) small = false;
    if (force128 == 2 && th_s >= 1) small = true;
    const int th = small ? th_s : th_b;
    *out = BnsPlan{small ? 5 : 0, th, ceil_div(d->H, th), d->W, 1};
    return FT_OK;
  }
  const int th_big = rows(96, 120), th_small = rows(64, 96);
  // column-split form: <= 32 output pixels on a <= 64-pixel patch with its own x-halo; the fewest parts that fit
  int xs = 0, x_tw = 0, x_th = 0;
  for (int cs = 2; cs <= 4 && !xs; ++cs) {
    const int tw = ceil_div(d->W, cs);
    int th = 32 / tw;
    const int th2 = 64 / (tw + 2) - 2;
    th = th < th2 ? th : th2;
    if (th < 1 || tw * (cs - 1) >= d->W) continue;
    th = th < d->H ? th : d->H;
    xs = cs; x_tw = tw; x_th = ceil_div(d->H, ceil_div(d->H, th));
  }
  static const bool no_direct_k = getenv("FT_BNS_DIRECT") && atoi(getenv("FT_BNS_DIRECT")) == 0;
  if (no_direct_k) xs = 0;
  if (th_big < 1 && th_small < 1 && !xs) return FT_ERR_UNSUPPORTED;
  int pick;
  if (force == 1 || force == 2 || (force == 3 && xs)) pick = force;
  else {
    // Both variants run one workgroup per CU at the matrix pipe's pace (the FT_BNS_DBG=64/128 ablation: same phase times
    // with every load out of range), so the cost of a launch is rounds of 256 workgroups x MFMAs per workgroup:
    // 64 K-steps on MT1 halo tiles + (144 + 64) K-steps on MT2 output tiles, per wave and pair of channel tiles.
    auto cost = [&](int th, int mt1, int mt2) {
      const long long wg = (long long)d->N * ceil_div(d->H, th);
      return ((wg + 255) / 256) * (long long)(64 * mt1 + 208 * mt2);
    };
    if (th_big < 1) pick = 2;
    else if (th_small < 1) pick = 1;
    else pick = cost(th_big, 4, 3) <= cost(th_small, 3, 2) ? 1 : 2;
    if (pick == 1 && th_big < 1) pick = 2;
    if (pick == 2 && th_small < 1) pick = th_big >= 1 ? 1 : 3;
    if (xs && pick != 3) {
      // the column-split form only where it needs no more rounds of 256 workgroups than it saves in work per workgroup
      const long long wgx = (long long)d->N * ceil_div(d->H, x_th) * xs;
      const long long cx = ((wgx + 255) / 256) * (long long)(64 * 2 + 208 * 1);
      const long long cf = pick == 1 ? cost(th_big, 4, 3) : cost(th_small, 3, 2);
      if (cx < cf) pick = 3;
    }
  }
  if (pick == 1 && th_big < 1) pick = 2;
  if (pick == 2 && th_small < 1) pick = th_big >= 1 ? 1 : 3;
  if (pick == 3) {
    *out = BnsPlan{3, x_th, ceil_div(d->H, x_th) * xs, x_tw, xs};
    return FT_OK;
  }
  const int th = pick == 1 ? th_big : th_small;
  *out = BnsPlan{pick, th, ceil_div(d->H, th), d->W, 1};
  return FT_OK;
}

template <int P, int MT1, int MT2, bool FOLD = false>
static int bns_launch(const BnsParams& p, hipStream_t s) {
  auto k = bottleneck_stream_kernel<P, MT1, MT2, FOLD>;
  static bool attr_done[64] = {};          // the LDS opt-in is per device
  int dev = 0;
  FT_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_done[dev]) {
    FT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, BnsGeom<P>::LDS_BYTES));
    if (dev >= 0 && dev < 64) attr_done[dev] = true;
  }
  hipLaunchKernelGGL(k, dim3(p.total), dim3(256), BnsGeom<P>::LDS_BYTES, s, p);
  FT_LAUNCH_CHECK("bottleneck_stream_kernel");
  return FT_OK;
}

#ifndef FT_BNS_SLOTS
#define FT_BNS_SLOTS 4        // weight-step register slots of the full-width strips: 3 -> 4 = +0.9-1.2 % on the R50 batch-64 step (56.25 / 56.34 -> 56.74 / 57.01 k crops/s same box), 6: -4.5 %; the column-split form: no difference at 3 / 4 / 6 (tools/dev/ab/ns_ab.sh)
#endif
#ifndef FT_BNS_XH_SLOTS
#define FT_BNS_XH_SLOTS 3
#endif
#ifndef FT_BNS_WAVES_DEFAULT
#define FT_BNS_WAVES_DEFAULT 4      // waves per workgroup of the direct kernels unless FT_BNS_WAVES says otherwise
#endif
template <int MT1, int MT2, bool XH, int HEADC = 0, int NW = 4, bool FOLD = false>
static int bns_launch_direct(const BnsParams& p, hipStream_t s) {
  auto k = bottleneck_stream_direct_kernel<MT1, MT2, XH, XH ? FT_BNS_XH_SLOTS : FT_BNS_SLOTS, HEADC, NW, FOLD>;
  constexpr int lds = 81920 + 2 * MT2 * 32 * 512;
  static bool attr_done[64] = {};          // the LDS opt-in is per device
  int dev = 0;
  FT_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_done[dev]) {
    FT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if (dev >= 0 && dev < 64) attr_done[dev] = true;
  }
  hipLaunchKernelGGL(k, dim3(p.total), dim3(64 * NW), lds, s, p);
  FT_LAUNCH_CHECK("bottleneck_stream_direct_kernel");
  return FT_OK;
}

}  // namespace
}  // namespace ft

extern "C" int ft_bottleneck_stream_supported(const ft_bottleneck_desc* d) {
  ft::BnsPlan pl;
  return ft::bns_plan(d, &pl);
}

extern "C" int ft_bottleneck_stream_folds(const ft_bottleneck_desc* d) {
  ft::BnsPlan pl;
  if (ft::bns_plan(d, &pl) != FT_OK) return 0;
  return pl.variant != 4;
}

extern "C" long long ft_bottleneck_stream_weight_bytes(const ft_bottleneck_desc* d) {
  ft::BnsPlan pl;
  if (ft::bns_plan(d, &pl) != FT_OK) return 0;
  if (pl.variant == 4) return (long long)(d->C / 64 + 9 * ft::BnsGeom<256>::KC) * ft::BnsGeom<256>::WSTEP;
  return d->P == 128 ? (long long)ft::BnsGeom<128>::GEND * ft::BnsGeom<128>::WSTEP : (long long)ft::BnsGeom<256>::GEND * ft::BnsGeom<256>::WSTEP;
}

extern "C" int ft_bottleneck_stream_pack(const ft_bottleneck_desc* d, const void* w1, const void* w2, const void* w3, void* wstream,
                                         ft_stream_t stream) {
  using namespace ft;
  BnsPlan pl;
  const int st = bns_plan(d, &pl);
  if (st != FT_OK) return st;
  if (!w1 || !w2 || (!w3 && pl.variant != 4) || !wstream) return FT_ERR_INVALID_ARG;
  hipStream_t s = as_stream(stream);
  const half_t *a = static_cast<const half_t*>(w1), *b = static_cast<const half_t*>(w2), *c = static_cast<const half_t*>(w3);
  if (pl.variant == 4) {
    const int n16 = (int)(ft_bottleneck_stream_weight_bytes(d) / 16);
    hipLaunchKernelGGL(bns_pack_head_kernel, dim3(ceil_div(n16, 256)), dim3(256), 0, s, a, b, static_cast<uint4_t*>(wstream), d->C);
    FT_LAUNCH_CHECK("bns_pack_head_kernel");
    return FT_OK;
  }
  if (d->P == 128) {
    const int n16 = BnsGeom<128>::GEND * BnsGeom<128>::WSTEP / 16;
    hipLaunchKernelGGL(bns_pack_kernel<128>, dim3(ceil_div(n16, 256)), dim3(256), 0, s, a, b, c, static_cast<uint4_t*>(wstream));
  } else {
    const int n16 = BnsGeom<256>::GEND * BnsGeom<256>::WSTEP / 16;
    hipLaunchKernelGGL(bns_pack_kernel<256>, dim3(ceil_div(n16, 256)), dim3(256), 0, s, a, b, c, static_cast<uint4_t*>(wstream));
  }
  FT_LAUNCH_CHECK("bns_pack_kernel");
  return FT_OK;
}

extern "C" int ft_bottleneck_stream_fwd(const ft_bottleneck_desc* d, const void* x, const void* wstream, const float* tables, void* y,
                                        ft_stream_t stream) {
  using namespace ft;
  BnsPlan pl;
  const int st = bns_plan(d, &pl);
  if (st != FT_OK) return st;
  if (!x || !wstream || !tables || !y || x == y) return FT_ERR_INVALID_ARG;
  if (d->folded && pl.variant == 4) return FT_ERR_UNSUPPORTED;      // the stride-2 head keeps its tables (ft_bottleneck_stream_folds)
  BnsParams p{};
  p.x = static_cast<const char*>(x);
  p.y = static_cast<char*>(y);
  p.ws = static_cast<const char*>(wstream);
  p.tab = reinterpret_cast<const char*>(tables);
  p.H = d->H; p.W = d->W; p.TH = pl.TH; p.ppi = pl.ppi;
  p.TWc = pl.TWc; p.csplit = pl.csplit;
  if (pl.variant == 4) { p.Ho = d->H / 2; p.Wo = d->W / 2; }
  p.total = d->N * pl.ppi;
  p.x_cstride = d->x_cstride; p.x_coff = d->x_coff; p.y_cstride = d->y_cstride; p.y_coff = d->y_coff;
  p.x_bytes = (unsigned)((size_t)d->N * d->H * d->W * d->x_cstride * 2);
  p.y_bytes = (unsigned)((size_t)d->N * d->H * d->W * d->y_cstride * 2);
  p.ws_bytes = (unsigned)ft_bottleneck_stream_weight_bytes(d);
  if (pl.variant == 4) p.y_bytes = (unsigned)((size_t)d->N * p.Ho * p.Wo * d->y_cstride * 2);
  static const int dbg = getenv("FT_BNS_DBG") ? atoi(getenv("FT_BNS_DBG")) : 0;
  p.dbg = dbg;
  if (dbg & 64) p.ws_bytes = 0;     // dev: every weight load out of range (returns 0, no L2 access): the kernel's time without its weight stream
  if (dbg & 128) p.x_bytes = 0;     // dev: likewise the input
  hipStream_t s = as_stream(stream);
  // FT_BNS_WAVES=4|8 (read per call: dev A/B and tests): waves per workgroup of the direct (256-plane) kernels.  Default: eight
  // for the column-split form (R101 384x288 at 16 crops: 34.4 -> 33.3 us per block, R50 at 16 crops 30.0 -> 28.5), four for the
  // full-width strips (batch 64: 46.1 vs 46.0 us — the eight-wave form gains in phases 1 and 3 what its doubled pixel-operand
  // reads cost in phase 2; both forms sit on the CU's 64 B/clk weight path: 32 KiB of fragments per 16-MFMA step)
  const bool waves8 = getenv("FT_BNS_WAVES") ? atoi(getenv("FT_BNS_WAVES")) == 8 : (FT_BNS_WAVES_DEFAULT == 8 || pl.variant == 3);
  static const bool no_direct = getenv("FT_BNS_DIRECT") && atoi(getenv("FT_BNS_DIRECT")) == 0;
  if (d->folded) {
    switch (pl.variant) {
      case 0: return bns_launch<128, 4, 3, true>(p, s);
      case 5: return bns_launch<128, 3, 2, true>(p, s);
      case 1: return bns_launch<256, 4, 3, true>(p, s);
      case 3: return waves8 ? bns_launch_direct<2, 1, true, 0, 8, true>(p, s) : bns_launch_direct<2, 1, true, 0, 4, true>(p, s);
      default: return no_direct ? bns_launch<256, 3, 2, true>(p, s)
                                : (waves8 ? bns_launch_direct<3, 2, false, 0, 8, true>(p, s) : bns_launch_direct<3, 2, false, 0, 4, true>(p, s));
    }
  }
  switch (pl.variant) {
    case 0: return bns_launch<128, 4, 3>(p, s);
    case 5: return bns_launch<128, 3, 2>(p, s);
    case 1: return bns_launch<256, 4, 3>(p, s);
    case 3: return waves8 ? bns_launch_direct<2, 1, true, 0, 8>(p, s) : bns_launch_direct<2, 1, true>(p, s);
    case 4: return waves8 ? bns_launch_direct<4, 1, false, 8, 8>(p, s) : bns_launch_direct<4, 1, false, 8>(p, s);
    default: {
      // 64-pixel strips at 256 planes: weights straight to registers (FT_BNS_DIRECT=0: through the LDS ring, dev A/B)
      return no_direct ? bns_launch<256, 3, 2>(p, s) : (waves8 ? bns_launch_direct<3, 2, false, 0, 8>(p, s) : bns_launch_direct<3, 2, false>(p, s));
    }
  }
}
