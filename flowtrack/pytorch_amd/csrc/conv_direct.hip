// 1x1 convolutions of the deep, small-map stages (fp16) as a weight-streaming GEMM whose weight fragments never touch
// LDS (reference: the 1x1 convs of Bottleneck, lib/pose/models/blocks.py:89,95 + the projection shortcut :98-103, run
// there through torch.nn -> cuDNN).
//
// Why: at 16x12 / 8x6 maps a layer has 12 k / 3 k pixels at batch 64.  The GEMM tiles that fill 256 CUs are then so small
// (64x64) that every operand byte is re-delivered L2 -> LDS many times and a K-step is a barrier-bound ~0.5 us: layer4's
// 1x1 convs run at 260-290 TFLOP/s in conv_igemm_dma_kernel.  Here a workgroup owns 96 pixels; the pixel operand streams
// through a 3-slot LDS ring in K chunks (whole lines per pixel, DMA), and each wave pulls ITS weight fragments straight
// from the fragment-ordered stream into registers two chunks ahead (1-KiB contiguous loads, nothing shared between waves,
// so an LDS round trip would only add traffic on the LDS port; tools/dev/ubench/stream_ring.hip).  Two shapes:
//   KSPLIT = 1  N-tile 256: wave w owns output-channel tiles {2w, 2w+1}, all four K16 slices of a 64-channel chunk
//   KSPLIT = 4  N-tile  64: all waves own tiles {0, 1}; a chunk is 256 channels and wave w takes its w-th quarter; the four
//               partial accumulators meet in LDS (for N <= 512 layers, whose 96 x 256 tiles would leave CUs idle)
// Either way a wave runs 2 x 3 MFMA tiles over four K16 slices per chunk from 8 register-resident fragments.
// Epilogue: folded BN (+ identity residual) + ReLU, fp16, through an LDS staging tile, whole-line stores.
// Optional second input (K-concat, ft_conv_desc.x2_*: conv3 + projection shortcut as one GEMM): its chunks follow x's.
#include <stdlib.h>

#include <type_traits>

#include "ft_common.h"
#include "conv_wstat.h"

namespace ft {
namespace {

struct CdParams {
  const char* x;
  const char* x2;
  const char* res;
  char* y;
  const char* ws;
  const float* scale;
  const float* shift;
  int M;                       // pixels
  int nc1, nc2;                // chunks of x / of x2
  int x_cstride, x_coff;
  int HqWq, Wq;                // output pixel grid (for the strided second input)
  int x2_hi, x2_wi, x2_cstride, x2_coff, x2_stride;
  int y_cstride, y_coff, res_cstride, res_coff;
  int Cout, act;
  float slope;
  int npt, ncb;                // pixel tiles, output-channel blocks
  unsigned x_bytes, x2_bytes, y_bytes, res_bytes, ws_bytes;
  int cpt, kw, Hi, Wi, stride, pad;   // TAPS form (kh x kw, any stride): chunks per tap, kernel width, input map, geometry
  int dbg;                     // FT_CD_DBG (dev): 32 = phase timestamps of wave 0 into the tile's first output row
  int wmajor;                  // workgroup order: 0 = channel block fastest (an XCD shares pixel tiles), 1 = pixel tile fastest (an XCD shares weights)
};

// Which operand should an XCD's L2 share?  Block b runs on XCD b % 8 and the kernels hand each XCD one contiguous range of the
// (pixel tile, channel block) grid.  Channel block fastest: the range covers few pixel tiles and EVERY channel block, so each of
// the eight L2s pulls the whole weight stream.  Right for the ResNet stages (2-5 MB of weights against 3-100 MB of activations),
// wrong for FlowNet's deep end: conv6_1 has 18.9 MB of weights for 1.5 MB of input, and 8 x 18.9 MB through HBM is 30 us of a
// 36-us launch.  Pixel tile fastest gives an XCD 1 / 8 of the channel blocks and every pixel tile: 8 x the input + 1 x the weights.
static inline int cd_wmajor(long long x_bytes, long long w_bytes, int npt, int ncb) {
  // Measured (round 6, one call): alone and L2-warm conv6_1 36.2 -> 27.1 us, conv6 21.4 -> 20.2 us; behind cold caches 35.7 -> 32.8 us;
  // FlowNet2S with the rule below 18.93 / 19.00 / 19.00 k against 19.00 / 19.02 / 19.06 k pairs/s without: nothing in the network, so
  // the order is OFF unless FT_CD_WMAJOR says otherwise (1 = every layer, 2 = the byte rule).
  static const int mode = getenv("FT_CD_WMAJOR") ? atoi(getenv("FT_CD_WMAJOR")) : 0;
  if (mode != 2) return mode == 1;
  if (ncb < 8 || npt < 2) return 0;
  return 8 * x_bytes + w_bytes < x_bytes + 8 * w_bytes;
}

template <int N, int I = 0, typename F>
__device__ __forceinline__ void cd_unroll(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    cd_unroll<N, I + 1>(f);
  }
}

#define CD_BARRIER() asm volatile("s_barrier" ::: "memory")

__host__ __device__ constexpr int cd_sigma(int r) { return 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3); }   // as bns_sigma

// XOR key of a ring row's 16-byte chunk position.  128-byte rows (CPR = 8): two rows share a 256-byte bank row, so the key
// changes every second row (`row & 7` made rows r and r + 8 collide: 27-44 % bank-conflict cycles in the PMC); 512-byte rows: row & 15.
#ifndef FT_CD_KEY_SHIFT
#define FT_CD_KEY_SHIFT 1
#endif
template <int CPR>
__device__ __forceinline__ int cd_key(int row) { return CPR < 16 ? ((row >> FT_CD_KEY_SHIFT) & (CPR - 1)) : (row & 15); }

#ifndef FT_CD_WSLOTS
#define FT_CD_WSLOTS 3
#endif
#ifndef FT_CD_L2_TOUCH
#define FT_CD_L2_TOUCH 0   // conv_direct_kernel: every wave touches its weight stream once at kernel start (cold L2 inside a network).
                           // Measured in the R50 network (net_bench.py, NB_HOT=1, two runs each): the K-concat exits 27 -> 31-35 us, the step
                           // 1.147 -> 1.169 ms: up to 36 LDS-DMA issues per wave in front of the first x load cost more than the misses they hide
#endif
template <int KSPLIT>
struct CdGeom {
  static constexpr int MT = 3, BP = 96;                       // pixel tiles per wave, pixels per workgroup
  static constexpr int BN = KSPLIT == 1 ? 256 : 64;           // output channels per workgroup
  static constexpr int CHUNK_K = KSPLIT == 1 ? 64 : 256;      // channels per ring slot
  static constexpr int ROWB = CHUNK_K * 2;                    // bytes of a ring row
  static constexpr int XB = BP * ROWB;                        // bytes of a ring slot
  static constexpr int LX = XB / 1024 / 4;                    // 1-KiB x loads per wave per chunk
  static constexpr int WCHUNK = 4 * 8 * 1024;                 // weight bytes per chunk: 4 waves x 8 fragments
  static constexpr int STG_ROWB = BN * 2;
  static constexpr int RING = 3 * XB;
  static constexpr int PART = KSPLIT == 1 ? 0 : 4 * 6 * 4096; // fp32 partial tiles of the four waves
  static constexpr int LDS_BYTES = (RING > PART + BP * STG_ROWB ? RING : PART + BP * STG_ROWB) + 2 * BN * 4;
  static_assert(LDS_BYTES <= 163840, "LDS map");   // ring [0, RING) during the K walk; afterwards partials [0, PART) + staging tile behind them
};

// NCH > 0: the chunk walk is unrolled for exactly NCH chunks — every vector-memory wait is then a compile-time count.  (With
// a run-time trip count hipcc joins its wait-count states at the loop header and drains the whole queue, vmcnt(0), at the
// first use of a prefetched weight fragment: the look-ahead is lost, 22 instead of 12 us on layer4's conv3.)  NCH = 0 is
// the run-time-loop fallback for K sizes without an instantiation.
// TAPS: a kh x kw convolution (3x3 / stride 2 of the stages' entry blocks, blocks.py:92-95 with stride on conv2): chunk c of
// the K walk is channel chunk c % cpt of tap c / cpt; the pixel operand of a tap is a gather (input pixel oy*s - pad + ky,
// ox*s - pad + kx: one base offset per ring row + a scalar tap offset, taps that leave the image read out of range = zeros).
template <int KSPLIT, bool HAS_RES, int NCH, bool TAPS = false>
__global__ __launch_bounds__(256, 1) void conv_direct_kernel(const CdParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using G = CdGeom<KSPLIT>;
  constexpr int MT = G::MT, BP = G::BP, BN = G::BN, ROWB = G::ROWB, XB = G::XB, LX = G::LX;
  constexpr int TAB = G::LDS_BYTES - 2 * BN * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  using c0 = std::integral_constant<int, 0>;
  using c1 = std::integral_constant<int, 1>;
  using c2 = std::integral_constant<int, 2>;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;

  // XCD-aware order: output-channel block fastest, so the workgroups that share a pixel tile sit in one XCD's L2
  int logical;
  {
    const int total = p.npt * p.ncb;
    const int b = blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = b & 7, loc = b >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  // wmajor (round 6): the XCD's contiguous range walks the pixel tiles of ONE channel block first, so an XCD's L2 holds its own
  // 1 / 8 of the weight stream instead of all of it (layers whose weights outweigh their input: see cd_wmajor)
  const int cb = p.wmajor ? logical / p.npt : logical % p.ncb, pt = p.wmajor ? logical % p.npt : logical / p.ncb;
  const int m0 = pt * BP;
  const int nchunk = NCH > 0 ? NCH : p.nc1 + p.nc2;

  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_x2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.nc2 ? p.x2 : p.x), 0, p.nc2 ? p.x2_bytes : p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.ws), 0, p.ws_bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;

#if FT_CD_L2_TOUCH
  // Inside a network this layer's weights are not in the XCD's L2 when the kernel starts, and the register ring keeps only
  // WS - 1 chunks (16 KiB per wave) in flight: a cold stream is paced by the miss latency.  Every wave therefore asks for all
  // lines of ITS stream once, up front (one dword per 128-byte line, LDS-DMA into a scratch corner: the oldest vector-memory
  // operations of the wave, every counted wait below covers them).  FT_CD_DBG & 512 switches it off.
  if (!(p.dbg & 512)) {
    for (int c = (NCH > 0 ? FT_CD_WSLOTS : 3) - 1; c < nchunk; ++c)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr)(smem + G::LDS_BYTES + wave * 256), 4, (unsigned)lane * 128u,
                                               ((cb * nchunk + c) * 4 + wave) * 8192, 0, 0);
  }
#endif
  // ---- loaders ---------------------------------------------------------------------------------------------------------
  // ring row = pixel, ROWB bytes; a 1-KiB wave load covers 1024 / ROWB rows; XOR swizzle on the SOURCE 16-byte position
  constexpr int CPR = ROWB / 16;                  // 16-byte positions per row: 8 or 32
  constexpr int RPL = 64 / CPR;                   // rows per wave load: 8 or 2
  unsigned x_voff[LX], x2_voff[TAPS ? 1 : LX];
  int x_tmask[TAPS ? LX : 1];      // TAPS: bit t = tap t of this row's pixel lies inside the input map
#pragma unroll
  for (int t = 0; t < LX; ++t) {
    const int row = (t * 4 + wave) * RPL + lane / CPR, pos = lane % CPR;
    const int m = m0 + row;
    const unsigned swz = (unsigned)((pos ^ cd_key<CPR>(row)) << 4);
    unsigned v = kOOB, v2 = kOOB;
    if constexpr (TAPS) {
      int mk = 0;
      if (m < p.M) {
        const int n = m / p.HqWq, rem = m - n * p.HqWq, oy = rem / p.Wq, ox = rem - oy * p.Wq;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        v = (unsigned)((((n * p.Hi + iy0) * p.Wi + ix0) * p.x_cstride + p.x_coff) * 2) + swz;    // tap (0, 0); may wrap, only used when valid
        const int ntap = p.nc1 / p.cpt;
        for (int tp = 0; tp < ntap; ++tp) {
          const int ky = tp / p.kw, kx = tp - ky * p.kw;
          if ((unsigned)(iy0 + ky) < (unsigned)p.Hi && (unsigned)(ix0 + kx) < (unsigned)p.Wi) mk |= 1 << tp;
        }
      }
      x_tmask[t] = mk;
      x_voff[t] = v;
      continue;
    }
    if (m < p.M) {
      v = (unsigned)((m * p.x_cstride + p.x_coff) * 2) + swz;
      if (p.nc2) {
        const int n = m / p.HqWq, rem = m - n * p.HqWq, qy = rem / p.Wq, qx = rem - qy * p.Wq;
        v2 = (unsigned)((((n * p.x2_hi + qy * p.x2_stride) * p.x2_wi + qx * p.x2_stride) * p.x2_cstride + p.x2_coff) * 2) + swz;
      }
    }
    x_voff[t] = v;
    x2_voff[TAPS ? 0 : t] = v2;
  }
  auto issue_x = [&](int c, int buf) {            // chunk c of the K walk: x's chunks, then x2's
    char* dst = smem + buf * XB;
    if constexpr (TAPS) {
      const int tap = c / p.cpt, cc = c - tap * p.cpt;
      const int ky = tap / p.kw, kx = tap - ky * p.kw;
      const unsigned delta = (unsigned)(((ky * p.Wi + kx) * p.x_cstride) * 2 + cc * ROWB);
#pragma unroll
      for (int t = 0; t < LX; ++t)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr)(dst + (t * 4 + wave) * 1024), 16,
                                                 ((x_tmask[t] >> tap) & 1) ? x_voff[t] + delta : kOOB, 0, 0, 0);
    } else if (c < p.nc1) {
#pragma unroll
      for (int t = 0; t < LX; ++t)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr)(dst + (t * 4 + wave) * 1024), 16, x_voff[t], c * ROWB, 0, 0);
    } else {
#pragma unroll
      for (int t = 0; t < LX; ++t)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x2, (lds_ptr)(dst + (t * 4 + wave) * 1024), 16, x2_voff[t], (c - p.nc1) * ROWB, 0, 0);
    }
  };
  // weights of chunk c for this wave: 8 contiguous KiB at ((cb * nchunk + c) * 4 + wave) * 8 KiB; past the stream: zeros
  const unsigned lane16 = (unsigned)lane * 16u;
  // weight-step register slots: the K walk's look-ahead is WS - 1 chunks of 8 KiB per wave.  Three slots keep 64 KB per CU in
  // flight; the unrolled walks (every wait a compile-time count) can take FT_CD_WSLOTS (A/B builds: -DFT_CD_WSLOTS=4)
  constexpr int WS = NCH > 0 ? FT_CD_WSLOTS : 3;
  uint4_t areg[WS][4][2];
  auto load_a = [&](auto slotc, int c) {
    constexpr int SL = decltype(slotc)::value;
    const int base = c < nchunk ? ((cb * nchunk + c) * 4 + wave) * 8192 : 0x7fff0000;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        areg[SL][kk][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16, base + (kk * 2 + i) * 1024, 0);
  };

  // folded-BN table of this channel block -> LDS (scale may be absent: the K-concat form folds it into the weights)
  if (tid < BN / 4) {
    const float4_t one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
    const int ch = cb * BN + tid * 4;
    reinterpret_cast<float4_t*>(smem + TAB)[tid] = p.scale ? *reinterpret_cast<const float4_t*>(p.scale + ch) : one;
    reinterpret_cast<float4_t*>(smem + TAB + BN * 4)[tid] = p.shift ? *reinterpret_cast<const float4_t*>(p.shift + ch) : zero;
  }
  // residual rows of this wave's output tiles, accumulator layout (lane = pixel, 16 consecutive channels): two 16-byte loads per tile
  constexpr int NTILE = 2 * MT;                    // output tiles a wave finalises at most
  const int wtile0 = KSPLIT == 1 ? 2 * wave : 0;   // first channel tile (inside the block) of the MFMA phase
  uint4_t rres[HAS_RES ? (KSPLIT == 1 ? NTILE : 2) : 1][2];
  if constexpr (HAS_RES) {
    const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.res), 0, p.res_bytes, 0x00020000);
    if constexpr (KSPLIT == 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          const int m = m0 + j * 32 + l31;
          const unsigned v = m < p.M ? (unsigned)((m * p.res_cstride + p.res_coff + cb * BN + (wtile0 + i) * 32 + 16 * lhi) * 2) : kOOB;
#pragma unroll
          for (int h = 0; h < 2; ++h) rres[i * MT + j][h] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, v, h * 16, 0);
        }
    } else {
      // after the reduction wave w finalises tiles w and w + 4 (tile = i * MT + j over 2 channel tiles x 3 pixel tiles)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int tl = wave + 4 * k, i = tl / MT, j = tl - i * MT;
        const int m = m0 + j * 32 + l31;
        const unsigned v = (tl < NTILE && m < p.M) ? (unsigned)((m * p.res_cstride + p.res_coff + cb * BN + i * 32 + 16 * lhi) * 2) : kOOB;
#pragma unroll
        for (int h = 0; h < 2; ++h) rres[k][h] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, v, h * 16, 0);
      }
    }
  }

  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define CD_TS(i) do { if (p.dbg & 32) ts[i] = __builtin_amdgcn_s_memtime(); } while (0)
  CD_TS(0);
  issue_x(0, 0);
  issue_x(1, 1);
  issue_x(2, 2);
  cd_unroll<WS - 1>([&](auto sc) { load_a(sc, decltype(sc)::value); });

  float16_t acc[2][MT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // B fragment offsets (slot-relative): row j*32 + l31, slice (KSPLIT == 4 ? 4 * wave : 0) + kk
  int b_off[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int row = j * 32 + l31;
    const int s0 = KSPLIT == 1 ? 0 : 4 * wave;
    b_off[j] = row * ROWB + ((((s0 * 2 + lhi) ^ cd_key<CPR>(row))) << 4);
  }
  uint4_t fx[2][MT];
  auto ldx = [&](auto setc, int buf, int kk) {
    constexpr int S = decltype(setc)::value;
    const char* xb = smem + buf * XB;
#pragma unroll
    for (int j = 0; j < MT; ++j) fx[S][j] = *reinterpret_cast<const uint4_t*>(xb + (b_off[j] ^ (kk << 5)));
  };
  auto mma = [&](auto setc, auto slotc, auto kkc) {
    constexpr int S = decltype(setc)::value, SL = decltype(slotc)::value, kk = decltype(kkc)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, areg[SL][kk][i]), __builtin_bit_cast(half8_t, fx[S][j]),
                                                           acc[i][j], 0, 0, 0);
  };
  auto issue_x_dummy = [&](int buf) {               // keeps the count of vector-memory operations per chunk constant
    char* dst = smem + buf * XB;
#pragma unroll
    for (int t = 0; t < LX; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr)(dst + (t * 4 + wave) * 1024), 16, kOOB, 0, 0, 0);
  };
  // one chunk (ring slot = register slot = c % 3): slice 0 of the pixel operand already sits in register set 0
  auto chunk = [&](auto slotc, auto wslotc, int c) {
    constexpr int SL = decltype(slotc)::value, WL = decltype(wslotc)::value;
    using slot = std::integral_constant<int, WL>;          // register slot of this chunk's weights; SL = its ring slot
    load_a(std::integral_constant<int, (WL + WS - 1) % WS>{}, c + WS - 1);
    ldx(c1{}, SL, 1);
    mma(c0{}, slot{}, std::integral_constant<int, 0>{});
    ldx(c0{}, SL, 2);
    mma(c1{}, slot{}, std::integral_constant<int, 1>{});
    ldx(c1{}, SL, 3);
    mma(c0{}, slot{}, std::integral_constant<int, 2>{});
    if (c + 1 < nchunk) {
      // x chunk c+1 has landed and every read of chunk c's slot is complete: refill it with chunk c+3.  Younger than x chunk
      // c+1 are the weights of chunk c+1 (needed next anyway), x chunk c+2 and the weights of chunk c+2 (WS = 3; with more slots
      // the weights of chunk c+1 are OLDER than x chunk c+1 and those of chunks c+2 .. c+WS-1 younger).
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LX + 8 * (WS - 2)) : "memory");
      CD_BARRIER();
      if (c + 3 < nchunk) issue_x(c + 3, SL);
      else issue_x_dummy(SL);
      ldx(c0{}, (SL + 1) % 3, 0);
    }
    mma(c1{}, slot{}, std::integral_constant<int, 3>{});
  };

  // x chunk 0 has landed (this wave's share): behind it chunks 1, 2 and the two weight chunks may fly
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * LX + 8 * (WS - 1)) : "memory");
  CD_BARRIER();
  CD_TS(1);
  ldx(c0{}, 0, 0);
  if constexpr (NCH > 0) {
    cd_unroll<NCH>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      chunk(std::integral_constant<int, c % 3>{}, std::integral_constant<int, c % WS>{}, c);
    });
  } else {
    for (int c = 0; c < nchunk; c += 3) {
      chunk(c0{}, c0{}, c);
      if (c + 1 < nchunk) chunk(c1{}, c1{}, c + 1);
      if (c + 2 < nchunk) chunk(c2{}, c2{}, c + 2);
    }
  }
  CD_TS(2);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the look-ahead (dummy) loads must not outlive the ring
  CD_BARRIER();                                                    // every wave is past its last ring read
  CD_TS(3);

  // ---- epilogue ----------------------------------------------------------------------------------------------------------
  const float* tsc = reinterpret_cast<const float*>(smem + TAB);
  const float* tsh = tsc + BN;
  constexpr int SCPR = G::STG_ROWB / 16;            // 16-byte chunks per staging row: 32 or 8
  constexpr int SMASK = SCPR < 16 ? SCPR - 1 : 15;
  char* stg = smem + G::PART;
  // act(v) = max(v, k*v) with k = 0 (relu), slope (leaky, <= 1) or 1 (none): branch-free
  const float act_k = p.act == FT_ACT_RELU ? 0.f : (p.act == FT_ACT_LEAKY ? p.slope : 1.f);
  auto finish_tile = [&](const float16_t& a, int i, int j, const uint4_t* rr) {   // channel tile i (inside the block), pixel tile j
    const int ch = i * 32 + 16 * lhi;
    const int row = j * 32 + l31;
    float4_t sc[4], sh[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      sc[g4] = *reinterpret_cast<const float4_t*>(tsc + ch + g4 * 4);
      sh[g4] = *reinterpret_cast<const float4_t*>(tsh + ch + g4 * 4);
    }
    half8_t o[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      half8_t rs;
      if constexpr (HAS_RES) rs = __builtin_bit_cast(half8_t, rr[h]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int r = h * 8 + e;
        float v = a[r] * sc[r >> 2][r & 3] + sh[r >> 2][r & 3];
        if constexpr (HAS_RES) v += (float)rs[e];
        o[h][e] = (half_t)act_mul(v, act_k);
      }
      *reinterpret_cast<half8_t*>(stg + row * G::STG_ROWB + ((((ch >> 3) + h) ^ (row & SMASK)) << 4)) = o[h];
    }
  };
  if constexpr (KSPLIT == 1) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) finish_tile(acc[i][j], wtile0 + i, j, HAS_RES ? rres[i * MT + j] : nullptr);
  } else {
    // the four K quarters meet in LDS: partial tile (wave, tile, register group) as lane-contiguous float4 rows
    float4_t* part = reinterpret_cast<float4_t*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const float4_t v = {acc[i][j][4 * g4], acc[i][j][4 * g4 + 1], acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3]};
          part[((wave * NTILE + i * MT + j) * 4 + g4) * 64 + lane] = v;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CD_BARRIER();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int tl = wave + 4 * k;
      if (tl < NTILE) {
        float16_t sum;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          float4_t v = part[((0 * NTILE + tl) * 4 + g4) * 64 + lane];
#pragma unroll
          for (int w = 1; w < 4; ++w) v += part[((w * NTILE + tl) * 4 + g4) * 64 + lane];
          sum[4 * g4] = v[0]; sum[4 * g4 + 1] = v[1]; sum[4 * g4 + 2] = v[2]; sum[4 * g4 + 3] = v[3];
        }
        finish_tile(sum, tl / MT, tl % MT, HAS_RES ? rres[k] : nullptr);
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  CD_BARRIER();
  constexpr int NST = BP * SCPR / 256;              // 16-byte chunks per thread: 12 or 3
#pragma unroll
  for (int k = 0; k < NST; ++k) {
    const int idx = tid + 256 * k, row = idx / SCPR, ch = idx % SCPR;
    const int m = m0 + row;
    const uint4_t v = *reinterpret_cast<const uint4_t*>(stg + row * G::STG_ROWB + ((ch ^ (row & SMASK)) << 4));
    const unsigned voff = (m < p.M && cb * BN + ch * 8 < p.Cout) ? (unsigned)((m * p.y_cstride + p.y_coff + cb * BN + ch * 8) * 2) : kOOB;
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_y, voff, 0, FT_YSTORE_BUF_AUX);
  }
  if (p.dbg & 32) {
    ts[4] = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts[5] = __builtin_amdgcn_s_memtime();
    if (tid == 0 && m0 < p.M) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.y + ((size_t)m0 * p.y_cstride + p.y_coff + cb * BN) * 2);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = ts[i];
    }
  }
#endif
}

// ---- 3x3 / stride 1 / pad 1 on whole small maps (layer4's conv2: 8x6 at 256x192, 12x9 at 384x288) -----------------------
// A workgroup owns `ipw` whole images (<= MT*32 pixels) x 64 output channels.  The input tile (every channel of those images,
// <= 128 KiB) is loaded into LDS ONCE and serves all nine taps as a row shift (taps that leave the image read a zero row);
// the four waves split K by channel quarter (wave w: channels [w*C/4, (w+1)*C/4) of every tap), each streaming its own
// weight fragments straight into registers — no barrier inside the K walk at all.  The four partial accumulators meet in
// LDS as in the K-split 1x1 form.
struct C3Params {
  const char* x;
  char* y;
  const char* ws;
  const float* scale;
  const float* shift;
  int M, HW, W, H, ipw;        // pixels, map size, images per workgroup (stride-2 form: the INPUT map; M = output pixels)
  int Ho, Wo;                  // stride-2 form: output map (0 in the stride-1 form)
  int strip_rows, nstrips;     // stride-1 STRIP form (0: whole images): output rows per workgroup, strips per image
  int x_cstride, x_coff, y_cstride, y_coff;
  int Cout, act;
  float slope;
  int npt, ncb;
  unsigned x_bytes, y_bytes, ws_bytes;
  int wmajor;                  // see CdParams / cd_wmajor
};

// TT > MT: the STRIP form for maps too large to be resident whole (FlowNet's conv5_1 on 12 x 16, FlowNetS.py:30): a workgroup owns
// p.strip_rows output rows of one image; its tile holds those rows plus one halo row above and below ((strip_rows + 2) * W <=
// TT * 32 tile rows; rows outside the image are out-of-range loads = zeros, so only the x borders need the tap mask).
template <int MT, int SPT, int TT = MT>     // output pixel tiles per workgroup; 4-slice steps per tap per wave (= C / 256); tile row tiles
__global__ __launch_bounds__(256, 1) void conv3x3_direct_kernel(const C3Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int C = SPT * 256, ROWB = C * 2, BN = 64, NTILE = 2 * MT;
  constexpr int TROWS = MT * 32, TILE_ROWS = TT * 32;
  constexpr bool STRIP = TT != MT;
  constexpr int ZROW = 131072, TAB = ZROW + ROWB, STG = TAB + 2 * BN * 4, STG_ROWB = BN * 2;
  constexpr int PART = 4 * NTILE * 4096;
  constexpr int NSTEP = 9 * SPT;
  static_assert(TILE_ROWS * ROWB <= ZROW && PART <= ZROW && STG + TROWS * STG_ROWB <= 163840 && ZROW % ROWB == 0, "LDS map");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  using c0 = std::integral_constant<int, 0>;
  using c1 = std::integral_constant<int, 1>;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  int logical;
  {
    const int total = p.npt * p.ncb;
    const int b = blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = b & 7, loc = b >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  // wmajor (round 6): the XCD's contiguous range walks the pixel tiles of ONE channel block first, so an XCD's L2 holds its own
  // 1 / 8 of the weight stream instead of all of it (layers whose weights outweigh their input: see cd_wmajor)
  const int cb = p.wmajor ? logical / p.npt : logical % p.ncb, pt = p.wmajor ? logical % p.npt : logical / p.ncb;
  // whole images: tile rows = output pixels = the pixels of ipw images.  Strip: the tile starts one image row above the strip.
  int npix, m0, tile_m0, tile_rows;
  if constexpr (STRIP) {
    const int n = pt / p.nstrips, y0 = (pt - n * p.nstrips) * p.strip_rows;
    const int rows = p.H - y0 < p.strip_rows ? p.H - y0 : p.strip_rows;
    npix = rows * p.W;
    m0 = n * p.HW + y0 * p.W;
    tile_m0 = m0 - p.W;                             // (may be negative / run past the image: masked row by row below)
    tile_rows = (rows + 2) * p.W;
  } else {
    npix = p.ipw * p.HW;                            // pixels of this workgroup's images
    m0 = pt * npix;
    tile_m0 = m0;
    tile_rows = npix;
  }
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.ws), 0, p.ws_bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;

  // ---- the input tile: row = pixel, all C channels; 1 KiB per wave load = 1024 / ROWB rows; XOR swizzle (low 4 bits of the
  //      16-byte position ^= row & 15) on the source side
  constexpr int CPR = ROWB / 16, RPL = 1024 / ROWB > 0 ? 1024 / ROWB : 1;
  constexpr int LPR = ROWB > 1024 ? ROWB / 1024 : 1;   // wave loads per row (C = 1024: 2)
  constexpr int NLOAD = TILE_ROWS * ROWB / 1024 / 4;   // wave loads per wave
#pragma unroll
  for (int t = 0; t < NLOAD; ++t) {
    const int piece = t * 4 + wave;
    const int row = ROWB > 1024 ? piece / LPR : piece * RPL + lane / CPR;
    const int pos = ROWB > 1024 ? (piece % LPR) * 64 + lane : lane % CPR;
    const int m = tile_m0 + row;
    bool ok = row < tile_rows && m < p.M;
    if constexpr (STRIP) {                            // halo rows above / below the image: out of range = zeros
      const int img0 = (m0 / p.HW) * p.HW;
      ok = row < tile_rows && m >= img0 && m < img0 + p.HW;
    }
    const unsigned voff = ok ? (unsigned)((m * p.x_cstride + p.x_coff) * 2 + (((pos & ~15) | ((pos ^ row) & 15)) << 4)) : kOOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr)(smem + piece * 1024), 16, voff, 0, 0, 0);
  }
  const unsigned lane16 = (unsigned)lane * 16u;
  uint4_t areg[3][4][2];
  auto load_a = [&](auto slotc, int st) {            // step st of this wave's stream: 8 contiguous KiB; past the end: zeros
    constexpr int SL = decltype(slotc)::value;
    const int base = st < NSTEP ? ((cb * 4 + wave) * NSTEP + st) * 8192 : 0x7fff0000;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        areg[SL][kk][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16, base + (kk * 2 + i) * 1024, 0);
  };
  load_a(c0{}, 0);
  load_a(c1{}, 1);
  if (tid < ROWB / 16) *reinterpret_cast<uint4_t*>(smem + ZROW + tid * 16) = uint4_t{0u, 0u, 0u, 0u};
  if (tid < BN / 4) {
    const float4_t one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
    const int ch = cb * BN + tid * 4;
    reinterpret_cast<float4_t*>(smem + TAB)[tid] = p.scale ? *reinterpret_cast<const float4_t*>(p.scale + ch) : one;
    reinterpret_cast<float4_t*>(smem + TAB + BN * 4)[tid] = p.shift ? *reinterpret_cast<const float4_t*>(p.shift + ch) : zero;
  }
  // per-lane pixel geometry: 9-bit tap validity of the lane's pixel in each pixel tile
  int pix[MT], tmask[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int pp = j * 32 + l31;
    pix[j] = STRIP ? pp + p.W : pp;                  // tile row of the pixel itself (the strip's tile starts one image row higher)
    int mk = 0;
    if (pp < npix) {
      const int rem = pp % p.HW, yy = rem / p.W, xx = STRIP ? pp % p.W : rem - yy * p.W;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int ny = yy + t / 3 - 1, nx = xx + t % 3 - 1;
        // strip: rows above / below the image are zero rows OF THE TILE, only the x borders need the mask
        if ((STRIP || (unsigned)ny < (unsigned)p.H) && (unsigned)nx < (unsigned)p.W) mk |= 1 << t;
      }
    }
    tmask[j] = mk;
  }
  float16_t acc[2][MT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // byte base of the lane's operand row for tap t (slice s inside the wave's channel quarter is one more XOR, s << 5)
  constexpr int WCH = C / 32;                       // 16-byte positions per wave quarter
  auto row_bases = [&](int t, int (&rb)[MT]) {
    const int off = (t / 3 - 1) * p.W + (t % 3 - 1);
    const int wq = wave * WCH;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int row = pix[j] + off;
      const int v = row * ROWB + ((wq & ~15) << 4) + (((((wq & 15) + lhi) ^ row) & 15) << 4);
      rb[j] = ((tmask[j] >> t) & 1) ? v : ZROW + (lhi << 4);
    }
  };
  uint4_t fb[2][MT];
  auto ldb = [&](auto setc, int s, const int (&rb)[MT]) {
    constexpr int S = decltype(setc)::value;
#pragma unroll
    for (int j = 0; j < MT; ++j) fb[S][j] = *reinterpret_cast<const uint4_t*>(smem + (rb[j] ^ (s << 5)));
  };
  auto mma = [&](auto setc, auto slotc, auto kkc) {
    constexpr int S = decltype(setc)::value, SL = decltype(slotc)::value, kk = decltype(kkc)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, areg[SL][kk][i]), __builtin_bit_cast(half8_t, fb[S][j]),
                                                           acc[i][j], 0, 0, 0);
  };
  // the tile has landed (this wave's share; the two weight steps may fly), then everyone's
  asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
  CD_BARRIER();
  int rb[MT], rbn[MT];
  row_bases(0, rb);
  ldb(c0{}, 0, rb);
  cd_unroll<NSTEP>([&](auto sc) {
    constexpr int st = decltype(sc)::value;
    constexpr int tap = st / SPT, s0 = (st % SPT) * 4;       // first slice (inside the quarter) of this step
    using slot = std::integral_constant<int, st % 3>;
    load_a(std::integral_constant<int, (st + 2) % 3>{}, st + 2);
    ldb(c1{}, s0 + 1, rb);
    mma(c0{}, slot{}, std::integral_constant<int, 0>{});
    ldb(c0{}, s0 + 2, rb);
    mma(c1{}, slot{}, std::integral_constant<int, 1>{});
    ldb(c1{}, s0 + 3, rb);
    mma(c0{}, slot{}, std::integral_constant<int, 2>{});
    if constexpr (st + 1 < NSTEP) {
      constexpr int ntap = (st + 1) / SPT, ns0 = ((st + 1) % SPT) * 4;
      if constexpr (ntap != tap) {
        row_bases(ntap, rbn);
        ldb(c0{}, ns0, rbn);
      } else {
        ldb(c0{}, ns0, rb);
      }
    }
    mma(c1{}, slot{}, std::integral_constant<int, 3>{});
    if constexpr (st + 1 < NSTEP && (st + 1) / SPT != tap) {
#pragma unroll
      for (int j = 0; j < MT; ++j) rb[j] = rbn[j];
    }
  });
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  CD_BARRIER();                                     // every wave is past its last read of the tile: the partials overwrite it

  // ---- K quarters meet in LDS, then the epilogue (as the K-split 1x1 form) ---------------------------------------------------
  const float* tsc = reinterpret_cast<const float*>(smem + TAB);
  const float* tsh = tsc + BN;
  const float act_k = p.act == FT_ACT_RELU ? 0.f : (p.act == FT_ACT_LEAKY ? p.slope : 1.f);
  char* stg = smem + STG;
  float4_t* part = reinterpret_cast<float4_t*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4_t v = {acc[i][j][4 * g4], acc[i][j][4 * g4 + 1], acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3]};
        part[((wave * NTILE + i * MT + j) * 4 + g4) * 64 + lane] = v;
      }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  CD_BARRIER();
#pragma unroll
  for (int k = 0; k < (NTILE + 3) / 4; ++k) {
    const int tl = wave + 4 * k;
    if (tl < NTILE) {
      const int i = tl / MT, j = tl - i * MT;
      const int ch = i * 32 + 16 * lhi, row = j * 32 + l31;
      float4_t sc[4], sh[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        sc[g4] = *reinterpret_cast<const float4_t*>(tsc + ch + g4 * 4);
        sh[g4] = *reinterpret_cast<const float4_t*>(tsh + ch + g4 * 4);
      }
      half8_t o[2];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float4_t v = part[((0 * NTILE + tl) * 4 + g4) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) v += part[((w * NTILE + tl) * 4 + g4) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float u = v[e] * sc[g4][e] + sh[g4][e];
          o[g4 >> 1][(g4 & 1) * 4 + e] = (half_t)act_mul(u, act_k);
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
        *reinterpret_cast<half8_t*>(stg + row * STG_ROWB + ((((ch >> 3) + h) ^ (row & 7)) << 4)) = o[h];
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  CD_BARRIER();
  constexpr int NST = TROWS * 8 / 256;
#pragma unroll
  for (int k = 0; k < NST; ++k) {
    const int idx = tid + 256 * k, row = idx >> 3, ch = idx & 7;
    const int m = m0 + row;
    const uint4_t v = *reinterpret_cast<const uint4_t*>(stg + row * STG_ROWB + ((ch ^ (row & 7)) << 4));
    const unsigned voff = (row < npix && m < p.M && cb * BN + ch * 8 < p.Cout) ? (unsigned)((m * p.y_cstride + p.y_coff + cb * BN + ch * 8) * 2) : kOOB;
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_y, voff, 0, FT_YSTORE_BUF_AUX);
  }
#endif
}

// weight stream of the 3x3 form: [channel block][wave][step = tap * SPT + q][kk][i] fragments; K-major source, k = tap * C + ci
template <int SPT>
__global__ __launch_bounds__(256) void c3_pack_kernel(const half_t* __restrict__ w, uint4_t* __restrict__ out, int ncb, int kpad, int cout_pad) {
  constexpr int C = SPT * 256, NSTEP = 9 * SPT;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= ncb * 4 * NSTEP * 512) return;
  const int lane = idx & 63;
  int f = idx >> 6;
  const int i = f & 1; f >>= 1;
  const int kk = f & 3; f >>= 2;
  const int st = f % NSTEP; f /= NSTEP;
  const int wv = f & 3, cb = f >> 2;
  const int tap = st / SPT, s = (st % SPT) * 4 + kk;          // slice inside the wave's channel quarter
  const int l31 = lane & 31, lhi = lane >> 5;
  const int co = cb * 64 + i * 32 + cd_sigma(l31);
  const int k = tap * C + wv * (C / 4) + s * 16 + 8 * lhi;
  uint4_t v = {0u, 0u, 0u, 0u};
  if (co < cout_pad && k < kpad) v = *reinterpret_cast<const uint4_t*>(w + (size_t)co * kpad + k);
  out[idx] = v;
}


// ---- 3x3 / stride 2 / pad 1 on whole small maps (the entry block's conv2 of the 512-plane stage: 16x12 -> 8x6 at 256x192,
//      blocks.py:92-95 with stride on conv2; FlowNet conv6: 12x16 -> 6x8, FlowNetS.py:33) -----------------------------------
// The stride-1 form above with two differences.  (1) The input map of ONE image (<= 256 pixels) is four times the output map, so
// only 256 of its channels fit LDS at a time: the K walk is NH passes of 256 channels, each over all nine taps, the tile is
// reloaded between passes (the weight ring keeps running across the reload).  (2) The pixel operand of output pixel (oy, ox) for
// tap (ky, kx) is input row (2 oy + ky - 1) * Wi + 2 ox + kx - 1, again a per-lane row base (a zero row where the tap leaves the
// map).  A workgroup = one image x 64 output channels; the four waves split a pass's 256 channels (64 each = one 4-slice step
// per tap), partial accumulators meet in LDS as above.  Weight stream: [channel block][wave][step = pass * 9 + tap][kk][i].
template <int MT, int NH>      // output pixel tiles (Ho * Wo <= MT * 32); 256-channel passes (Cin = NH * 256)
__global__ __launch_bounds__(256, 1) void conv3x3s2_direct_kernel(const C3Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int ROWB = 512, BN = 64, NTILE = 2 * MT;
  constexpr int TROWS = MT * 32, INROWS = 256;
  constexpr int ZROW = 131072, TAB = ZROW + ROWB, STG = TAB + 2 * BN * 4, STG_ROWB = BN * 2;
  constexpr int PART = 4 * NTILE * 4096;
  constexpr int NSTEP = 9 * NH;
  static_assert(INROWS * ROWB <= ZROW && PART <= ZROW && STG + TROWS * STG_ROWB <= 163840, "LDS map");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  using c0 = std::integral_constant<int, 0>;
  using c1 = std::integral_constant<int, 1>;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  int logical;
  {
    const int total = p.npt * p.ncb;
    const int b = blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = b & 7, loc = b >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  // wmajor (round 6): the XCD's contiguous range walks the pixel tiles of ONE channel block first, so an XCD's L2 holds its own
  // 1 / 8 of the weight stream instead of all of it (layers whose weights outweigh their input: see cd_wmajor)
  const int cb = p.wmajor ? logical / p.npt : logical % p.ncb, pt = p.wmajor ? logical % p.npt : logical / p.ncb;
  // whole image: the tile is the image's input map.  STRIP form (p.strip_rows > 0: maps of more than 256 input pixels, FlowNet's
  // conv5 on 24 x 32): the workgroup owns strip_rows output rows; its tile = the 2 R + 1 input rows under them (rows outside the
  // image are out-of-range loads = zeros: only the x borders need the tap mask)
  const bool strip = p.strip_rows > 0;
  int n = pt, npix = p.Ho * p.Wo, m0, tile0 = 0, tile_rows = p.HW;
  if (strip) {
    n = pt / p.nstrips;
    const int y0 = (pt - n * p.nstrips) * p.strip_rows;
    const int rows = p.Ho - y0 < p.strip_rows ? p.Ho - y0 : p.strip_rows;
    npix = rows * p.Wo;
    m0 = n * p.Ho * p.Wo + y0 * p.Wo;
    tile0 = (2 * y0 - 1) * p.W;                     // first tile row as a pixel index of the image (negative for the top strip)
    tile_rows = (2 * rows + 1) * p.W;
  } else {
    m0 = n * npix;
  }
  const int npix_in = p.HW;                         // input pixels of an image
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.ws), 0, p.ws_bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;

  // ---- the input tile of one pass: row = input pixel, 256 channels = 32 16-byte positions; a 1-KiB wave load = 2 rows; XOR
  //      swizzle (low 4 bits of the position ^= row & 15) on the source side
  constexpr int NLOAD = INROWS * ROWB / 1024 / 4;      // 32 wave loads per wave
  unsigned x_voff[NLOAD];
#pragma unroll
  for (int t = 0; t < NLOAD; ++t) {
    const int piece = t * 4 + wave;
    const int row = piece * 2 + (lane >> 5), pos = lane & 31;
    const int ip = tile0 + row;                     // pixel of the image this tile row holds
    x_voff[t] = (row < tile_rows && (unsigned)ip < (unsigned)npix_in)
                    ? (unsigned)(((n * npix_in + ip) * p.x_cstride + p.x_coff) * 2 + (((pos & ~15) | ((pos ^ row) & 15)) << 4)) : kOOB;
  }
  // The tile travels THROUGH REGISTERS (32 x 16 bytes per lane), not by LDS-DMA: the DMA path delivers ~25 GB/s per CU (98 KiB =
  // 3.9 us, exposed once per pass: 33 us for the layer on 512 workgroups), the register path several times that, and — the point —
  // the NEXT pass's tile can be in flight in registers while this pass multiplies; at the pass boundary only the LDS writes remain.
  uint4_t treg[NLOAD];
  auto load_tile = [&](int pass) {
#pragma unroll
    for (int t = 0; t < NLOAD; ++t) treg[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, x_voff[t], pass * ROWB, 0);
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int t = 0; t < NLOAD; ++t) *reinterpret_cast<uint4_t*>(smem + (t * 4 + wave) * 1024 + lane * 16) = treg[t];
  };
  const unsigned lane16 = (unsigned)lane * 16u;
  uint4_t areg[3][4][2];
  auto load_a = [&](auto slotc, int st) {            // step st of this wave's stream: 8 contiguous KiB; past the end: zeros
    constexpr int SL = decltype(slotc)::value;
    const int base = st < NSTEP ? ((cb * 4 + wave) * NSTEP + st) * 8192 : 0x7fff0000;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        areg[SL][kk][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16, base + (kk * 2 + i) * 1024, 0);
  };
  load_tile(0);
  load_a(c0{}, 0);
  load_a(c1{}, 1);
  if (tid < ROWB / 16) *reinterpret_cast<uint4_t*>(smem + ZROW + tid * 16) = uint4_t{0u, 0u, 0u, 0u};
  if (tid < BN / 4) {
    const float4_t one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
    const int ch = cb * BN + tid * 4;
    reinterpret_cast<float4_t*>(smem + TAB)[tid] = p.scale ? *reinterpret_cast<const float4_t*>(p.scale + ch) : one;
    reinterpret_cast<float4_t*>(smem + TAB + BN * 4)[tid] = p.shift ? *reinterpret_cast<const float4_t*>(p.shift + ch) : zero;
  }
  store_tile();                                      // (hipcc waits for the tile's loads here; the two weight steps stay in flight)
  if constexpr (NH > 1) load_tile(1);                // the second pass's tile: in flight while the first pass multiplies
  // per-lane geometry: the input pixel of tap (0, 0) of the lane's output pixel in each tile, and its 9-bit tap validity
  int ibase[MT], tmask[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int pp = j * 32 + l31;
    int mk = 0, ib = 0;
    if (pp < npix) {
      const int oy = pp / p.Wo, ox = pp - oy * p.Wo;
      const int iy0 = strip ? 2 * oy : 2 * oy - 1, ix0 = 2 * ox - 1;    // (a strip's tile starts at input row 2 y0 - 1)
      ib = iy0 * p.W + ix0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int ny = iy0 + t / 3, nx = ix0 + t % 3;
        if ((strip || (unsigned)ny < (unsigned)p.H) && (unsigned)nx < (unsigned)p.W) mk |= 1 << t;
      }
    }
    ibase[j] = ib;
    tmask[j] = mk;
  }
  float16_t acc[2][MT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // byte base of the lane's operand row for tap t: this wave's 64 channels = positions [8 * wave, 8 * wave + 8); slice s is one
  // more XOR (s << 5)
  auto row_bases = [&](int t, int (&rb)[MT]) {
    const int off = (t / 3) * p.W + (t % 3);
    const int wq = wave * 8;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int row = ibase[j] + off;
      const int v = row * ROWB + ((wq & ~15) << 4) + (((((wq & 15) + lhi) ^ row) & 15) << 4);
      rb[j] = ((tmask[j] >> t) & 1) ? v : ZROW + (lhi << 4);
    }
  };
  uint4_t fb[2][MT];
  auto ldb = [&](auto setc, int s, const int (&rb)[MT]) {
    constexpr int S = decltype(setc)::value;
#pragma unroll
    for (int j = 0; j < MT; ++j) fb[S][j] = *reinterpret_cast<const uint4_t*>(smem + (rb[j] ^ (s << 5)));
  };
  auto mma = [&](auto setc, auto slotc, auto kkc) {
    constexpr int S = decltype(setc)::value, SL = decltype(slotc)::value, kk = decltype(kkc)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, areg[SL][kk][i]), __builtin_bit_cast(half8_t, fb[S][j]),
                                                           acc[i][j], 0, 0, 0);
  };
  // the first tile is in LDS (this wave's share), then everyone's
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  CD_BARRIER();
  int rb[MT], rbn[MT];
  row_bases(0, rb);
  ldb(c0{}, 0, rb);
  cd_unroll<NSTEP>([&](auto sc) {
    constexpr int st = decltype(sc)::value;
    constexpr int tap = st % 9;
    using slot = std::integral_constant<int, st % 3>;
    load_a(std::integral_constant<int, (st + 2) % 3>{}, st + 2);
    ldb(c1{}, 1, rb);
    mma(c0{}, slot{}, std::integral_constant<int, 0>{});
    ldb(c0{}, 2, rb);
    mma(c1{}, slot{}, std::integral_constant<int, 1>{});
    ldb(c1{}, 3, rb);
    mma(c0{}, slot{}, std::integral_constant<int, 2>{});
    if constexpr (st + 1 < NSTEP && tap != 8) {
      row_bases(tap + 1, rbn);
      ldb(c0{}, 0, rbn);
    }
    mma(c1{}, slot{}, std::integral_constant<int, 3>{});
    if constexpr (st + 1 < NSTEP) {
      if constexpr (tap != 8) {
#pragma unroll
        for (int j = 0; j < MT; ++j) rb[j] = rbn[j];
      } else {
        // pass boundary: every wave is past its last read of this pass's tile -> the next 256 channels, which have been
        // sitting in registers since the previous boundary, go to LDS; the tile after that starts its flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        CD_BARRIER();
        store_tile();
        if constexpr ((st + 1) / 9 + 1 < NH) load_tile((st + 1) / 9 + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        CD_BARRIER();
        row_bases(0, rb);
        ldb(c0{}, 0, rb);
      }
    }
  });
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  CD_BARRIER();                                     // every wave is past its last read of the tile: the partials overwrite it

  // ---- K quarters meet in LDS, then the epilogue (as the stride-1 form) ------------------------------------------------------
  const float* tsc = reinterpret_cast<const float*>(smem + TAB);
  const float* tsh = tsc + BN;
  const float act_k = p.act == FT_ACT_RELU ? 0.f : (p.act == FT_ACT_LEAKY ? p.slope : 1.f);
  char* stg = smem + STG;
  float4_t* part = reinterpret_cast<float4_t*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4_t v = {acc[i][j][4 * g4], acc[i][j][4 * g4 + 1], acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3]};
        part[((wave * NTILE + i * MT + j) * 4 + g4) * 64 + lane] = v;
      }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  CD_BARRIER();
#pragma unroll
  for (int k = 0; k < (NTILE + 3) / 4; ++k) {
    const int tl = wave + 4 * k;
    if (tl < NTILE) {
      const int i = tl / MT, j = tl - i * MT;
      const int ch = i * 32 + 16 * lhi, row = j * 32 + l31;
      float4_t sc[4], sh[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        sc[g4] = *reinterpret_cast<const float4_t*>(tsc + ch + g4 * 4);
        sh[g4] = *reinterpret_cast<const float4_t*>(tsh + ch + g4 * 4);
      }
      half8_t o[2];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float4_t v = part[((0 * NTILE + tl) * 4 + g4) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) v += part[((w * NTILE + tl) * 4 + g4) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float u = v[e] * sc[g4][e] + sh[g4][e];
          o[g4 >> 1][(g4 & 1) * 4 + e] = (half_t)act_mul(u, act_k);
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
        *reinterpret_cast<half8_t*>(stg + row * STG_ROWB + ((((ch >> 3) + h) ^ (row & 7)) << 4)) = o[h];
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  CD_BARRIER();
  constexpr int NST = TROWS * 8 / 256;
#pragma unroll
  for (int k = 0; k < NST; ++k) {
    const int idx = tid + 256 * k, row = idx >> 3, ch = idx & 7;
    const int m = m0 + row;
    const uint4_t v = *reinterpret_cast<const uint4_t*>(stg + row * STG_ROWB + ((ch ^ (row & 7)) << 4));
    const unsigned voff = (row < npix && m < p.M && cb * BN + ch * 8 < p.Cout) ? (unsigned)((m * p.y_cstride + p.y_coff + cb * BN + ch * 8) * 2) : kOOB;
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_y, voff, 0, FT_YSTORE_BUF_AUX);
  }
#endif
}

// weight stream of the stride-2 form: [channel block][wave][step = pass * 9 + tap][kk][i]; K-major source, k = tap * Cin + ci
template <int NH>
__global__ __launch_bounds__(256) void c3s2_pack_kernel(const half_t* __restrict__ w, uint4_t* __restrict__ out, int ncb, int kpad, int cout_pad) {
  constexpr int C = NH * 256, NSTEP = 9 * NH;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= ncb * 4 * NSTEP * 512) return;
  const int lane = idx & 63;
  int f = idx >> 6;
  const int i = f & 1; f >>= 1;
  const int kk = f & 3; f >>= 2;
  const int st = f % NSTEP; f /= NSTEP;
  const int wv = f & 3, cb = f >> 2;
  const int pass = st / 9, tap = st % 9;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int co = cb * 64 + i * 32 + cd_sigma(l31);
  const int k = tap * C + pass * 256 + wv * 64 + kk * 16 + 8 * lhi;
  uint4_t v = {0u, 0u, 0u, 0u};
  if (co < cout_pad && k < kpad) v = *reinterpret_cast<const uint4_t*>(w + (size_t)co * kpad + k);
  out[idx] = v;
}


// ---- the stride-2 form for TWO images per workgroup (round 4) ------------------------------------------------------------------
// conv3x3s2_direct_kernel streams a channel block's 590 KB of weights for 48 output pixels; with 64 crops that is 512 workgroups
// and the layer is bound by that stream (33 us; the stride-1 sibling, 96 pixels per workgroup, takes 20).  Here a workgroup takes
// TWO images (<= 96 output pixels, MT = 3): their input maps (<= 512 pixels) are resident 128 channels at a time — four passes, the
// next pass's tile in flight in registers — and the K split of a pass is (channel half) x (tap parity): wave w takes channels
// [64 (w & 1), +64) of the taps with parity ((w >> 1) + pass) & 1, i.e. 5 + 4 + 5 + 4 or 4 + 5 + 4 + 5 taps over the four passes,
// 18 four-slice steps per wave either way.  Every pass has five step slots; the slot a wave has no tap for multiplies zero weights
// (an out-of-range load: no traffic).  Weight stream: [channel block][wave][its 18 steps in walk order][kk][i].
template <int MT, int NLD>     // output pixel tiles (2 * Ho * Wo <= MT * 32); 1-KiB tile pieces per wave (2 * Hi * Wi * 256 B <= NLD * 4 KiB)
__global__ __launch_bounds__(256, 1) void conv3x3s2p_direct_kernel(const C3Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int ROWB = 256, BN = 64, NTILE = 2 * MT, NP = 4, NSL = 5, NSTEP = NP * NSL;
  constexpr int TROWS = MT * 32;
  constexpr int ZROW = 131072, TAB = ZROW + ROWB, STG = TAB + 2 * BN * 4, STG_ROWB = BN * 2;
  constexpr int PART = 4 * NTILE * 4096;
  static_assert(NLD * 4096 <= ZROW && PART <= ZROW && STG + TROWS * STG_ROWB <= 163840, "LDS map");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using c0 = std::integral_constant<int, 0>;
  using c1 = std::integral_constant<int, 1>;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int hch = wave & 1, qpar = wave >> 1;        // channel half of a pass, tap-parity group
  int logical;
  {
    const int total = p.npt * p.ncb;
    const int b = blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = b & 7, loc = b >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  // wmajor (round 6): the XCD's contiguous range walks the pixel tiles of ONE channel block first, so an XCD's L2 holds its own
  // 1 / 8 of the weight stream instead of all of it (layers whose weights outweigh their input: see cd_wmajor)
  const int cb = p.wmajor ? logical / p.npt : logical % p.ncb, pt = p.wmajor ? logical % p.npt : logical / p.ncb;
  const int npix_in = p.HW, npo = p.Ho * p.Wo;
  const int rows_in = 2 * npix_in;                  // input pixels of this workgroup's two images
  const int npix = 2 * npo;                         // its output pixels
  const int m0 = pt * npix;
  const int N_in = (p.M / npo) * npix_in;           // input pixels of the whole batch
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.ws), 0, p.ws_bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;

  // the input tile of one pass: row = input pixel (of either image), 128 channels = 16 16-byte positions, 4 rows per 1-KiB piece;
  // XOR swizzle (position ^= row & 15) on the source side; through registers as in the one-image form
  unsigned x_voff[NLD];
#pragma unroll
  for (int t = 0; t < NLD; ++t) {
    const int piece = t * 4 + wave;
    const int row = piece * 4 + (lane >> 4), pos = lane & 15;
    const int gp = pt * rows_in + row;
    x_voff[t] = (row < rows_in && gp < N_in) ? (unsigned)((gp * p.x_cstride + p.x_coff) * 2 + (((pos ^ row) & 15) << 4)) : kOOB;
  }
  uint4_t treg[NLD];
  auto load_tile = [&](int pass) {
#pragma unroll
    for (int t = 0; t < NLD; ++t) treg[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, x_voff[t], pass * ROWB, 0);
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int t = 0; t < NLD; ++t) *reinterpret_cast<uint4_t*>(smem + (t * 4 + wave) * 1024 + lane * 16) = treg[t];
  };
  // step slot st = pass * 5 + i -> tap 2 i + ((qpar + pass) & 1) (none when that is 9) and the wave's stream index
  auto slot_tap = [&](int st) { return 2 * (st % NSL) + ((qpar + st / NSL) & 1); };
  const unsigned lane16 = (unsigned)lane * 16u;
  uint4_t areg[3][4][2];
  auto load_a = [&](auto slotc, int st) {
    constexpr int SL = decltype(slotc)::value;
    int base = 0x7fff0000;
    if (st < NSTEP && slot_tap(st) < 9) {
      const int pass = st / NSL;
      const int sidx = (pass >> 1) * 9 + ((pass & 1) ? (qpar == 0 ? 5 : 4) : 0) + st % NSL;
      base = ((cb * 4 + wave) * 18 + sidx) * 8192;
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        areg[SL][kk][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16, base + (kk * 2 + i) * 1024, 0);
  };
  load_tile(0);
  load_a(c0{}, 0);
  load_a(c1{}, 1);
  if (tid < ROWB / 16) *reinterpret_cast<uint4_t*>(smem + ZROW + tid * 16) = uint4_t{0u, 0u, 0u, 0u};
  if (tid < BN / 4) {
    const float4_t one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
    const int ch = cb * BN + tid * 4;
    reinterpret_cast<float4_t*>(smem + TAB)[tid] = p.scale ? *reinterpret_cast<const float4_t*>(p.scale + ch) : one;
    reinterpret_cast<float4_t*>(smem + TAB + BN * 4)[tid] = p.shift ? *reinterpret_cast<const float4_t*>(p.shift + ch) : zero;
  }
  store_tile();
  load_tile(1);
  // per-lane geometry: tile row of tap (0, 0) of the lane's output pixel in each pixel tile, and its 9-bit tap validity
  int ibase[MT], tmask[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int pp = j * 32 + l31;
    int mk = 0, ib = 0;
    if (pp < npix) {
      const int img = pp >= npo ? 1 : 0, rem = pp - img * npo;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      const int iy0 = 2 * oy - 1, ix0 = 2 * ox - 1;
      ib = img * npix_in + iy0 * p.W + ix0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int ny = iy0 + t / 3, nx = ix0 + t % 3;
        if ((unsigned)ny < (unsigned)p.H && (unsigned)nx < (unsigned)p.W) mk |= 1 << t;
      }
    }
    ibase[j] = ib;
    tmask[j] = mk;
  }
  float16_t acc[2][MT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto row_bases = [&](int t, int (&rb)[MT]) {      // t = 9: the slot without a tap (zero weights): any readable row
    const int off = (t / 3) * p.W + (t % 3);
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int row = ibase[j] + off;
      const int v = row * ROWB + ((((hch * 8 + lhi) ^ row) & 15) << 4);
      rb[j] = (t < 9 && ((tmask[j] >> t) & 1)) ? v : ZROW + (lhi << 4);
    }
  };
  uint4_t fb[2][MT];
  auto ldb = [&](auto setc, int s, const int (&rb)[MT]) {
    constexpr int S = decltype(setc)::value;
#pragma unroll
    for (int j = 0; j < MT; ++j) fb[S][j] = *reinterpret_cast<const uint4_t*>(smem + (rb[j] ^ (s << 5)));
  };
  auto mma = [&](auto setc, auto slotc, auto kkc) {
    constexpr int S = decltype(setc)::value, SL = decltype(slotc)::value, kk = decltype(kkc)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, areg[SL][kk][i]), __builtin_bit_cast(half8_t, fb[S][j]),
                                                           acc[i][j], 0, 0, 0);
  };
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  CD_BARRIER();
  int rb[MT], rbn[MT];
  row_bases(slot_tap(0), rb);
  ldb(c0{}, 0, rb);
  cd_unroll<NSTEP>([&](auto sc) {
    constexpr int st = decltype(sc)::value;
    constexpr int pass = st / NSL, i5 = st % NSL;
    using slot = std::integral_constant<int, st % 3>;
    load_a(std::integral_constant<int, (st + 2) % 3>{}, st + 2);
    ldb(c1{}, 1, rb);
    mma(c0{}, slot{}, std::integral_constant<int, 0>{});
    ldb(c0{}, 2, rb);
    mma(c1{}, slot{}, std::integral_constant<int, 1>{});
    ldb(c1{}, 3, rb);
    mma(c0{}, slot{}, std::integral_constant<int, 2>{});
    if constexpr (i5 != NSL - 1) {
      row_bases(slot_tap(st + 1), rbn);
      ldb(c0{}, 0, rbn);
    }
    mma(c1{}, slot{}, std::integral_constant<int, 3>{});
    if constexpr (st + 1 < NSTEP) {
      if constexpr (i5 != NSL - 1) {
#pragma unroll
        for (int j = 0; j < MT; ++j) rb[j] = rbn[j];
      } else {
        // pass boundary: every wave is past its last read of this pass's tile -> the next 128 channels go from registers to LDS,
        // the tile after that starts its flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        CD_BARRIER();
        store_tile();
        if constexpr (pass + 2 < NP) load_tile(pass + 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        CD_BARRIER();
        row_bases(slot_tap(st + 1), rb);
        ldb(c0{}, 0, rb);
      }
    }
  });
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  CD_BARRIER();                                     // every wave is past its last read of the tile: the partials overwrite it

  const float* tsc = reinterpret_cast<const float*>(smem + TAB);
  const float* tsh = tsc + BN;
  const float act_k = p.act == FT_ACT_RELU ? 0.f : (p.act == FT_ACT_LEAKY ? p.slope : 1.f);
  char* stg = smem + STG;
  float4_t* part = reinterpret_cast<float4_t*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4_t v = {acc[i][j][4 * g4], acc[i][j][4 * g4 + 1], acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3]};
        part[((wave * NTILE + i * MT + j) * 4 + g4) * 64 + lane] = v;
      }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  CD_BARRIER();
#pragma unroll
  for (int k = 0; k < (NTILE + 3) / 4; ++k) {
    const int tl = wave + 4 * k;
    if (tl < NTILE) {
      const int i = tl / MT, j = tl - i * MT;
      const int ch = i * 32 + 16 * lhi, row = j * 32 + l31;
      float4_t sc[4], sh[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        sc[g4] = *reinterpret_cast<const float4_t*>(tsc + ch + g4 * 4);
        sh[g4] = *reinterpret_cast<const float4_t*>(tsh + ch + g4 * 4);
      }
      half8_t o[2];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float4_t v = part[((0 * NTILE + tl) * 4 + g4) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) v += part[((w * NTILE + tl) * 4 + g4) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float u = v[e] * sc[g4][e] + sh[g4][e];
          o[g4 >> 1][(g4 & 1) * 4 + e] = (half_t)act_mul(u, act_k);
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
        *reinterpret_cast<half8_t*>(stg + row * STG_ROWB + ((((ch >> 3) + h) ^ (row & 7)) << 4)) = o[h];
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  CD_BARRIER();
  constexpr int NST = TROWS * 8 / 256;
#pragma unroll
  for (int k = 0; k < NST; ++k) {
    const int idx = tid + 256 * k, row = idx >> 3, ch = idx & 7;
    const int m = m0 + row;
    const uint4_t v = *reinterpret_cast<const uint4_t*>(stg + row * STG_ROWB + ((ch ^ (row & 7)) << 4));
    const unsigned voff = (row < npix && m < p.M && cb * BN + ch * 8 < p.Cout) ? (unsigned)((m * p.y_cstride + p.y_coff + cb * BN + ch * 8) * 2) : kOOB;
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_y, voff, 0, FT_YSTORE_BUF_AUX);
  }
#endif
}

// weight stream of the two-image form: [channel block][wave][18 steps in the wave's walk order][kk][i]; K-major source, k = tap * 512 + ci
__global__ __launch_bounds__(256) void c3s2p_pack_kernel(const half_t* __restrict__ w, uint4_t* __restrict__ out, int ncb, int kpad, int cout_pad) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= ncb * 4 * 18 * 512) return;
  const int lane = idx & 63;
  int f = idx >> 6;
  const int i = f & 1; f >>= 1;
  const int kk = f & 3; f >>= 2;
  int sidx = f % 18; f /= 18;
  const int wv = f & 3, cb = f >> 2;
  const int hch = wv & 1, qpar = wv >> 1;
  int pass = 0;
  for (; pass < 4; ++pass) {                       // the wave's passes hold 5 or 4 steps, alternating
    const int np = ((qpar + pass) & 1) == 0 ? 5 : 4;
    if (sidx < np) break;
    sidx -= np;
  }
  const int tap = 2 * sidx + ((qpar + pass) & 1);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int co = cb * 64 + i * 32 + cd_sigma(l31);
  const int k = tap * 512 + pass * 128 + hch * 64 + kk * 16 + 8 * lhi;
  uint4_t v = {0u, 0u, 0u, 0u};
  if (co < cout_pad && k < kpad) v = *reinterpret_cast<const uint4_t*>(w + (size_t)co * kpad + k);
  out[idx] = v;
}

struct C3Plan {
  int mt, spt, ipw, npt, ncb, stride;
  int strip_rows = 0, nstrips = 0;
};

static int c3_plan(const ft_conv_desc* d, C3Plan* out) {
  if (!d) return FT_ERR_INVALID_ARG;
  if (d->act == FT_ACT_LEAKY && !(d->slope >= 0.f && d->slope <= 1.f)) return FT_ERR_UNSUPPORTED;   // epilogues use max(v, k * v): valid for 0 <= k <= 1 only
  if (d->dtype != FT_F16 || d->transposed || d->kh != 3 || d->kw != 3 || (d->stride != 1 && d->stride != 2) || d->pad != 1) return FT_ERR_UNSUPPORTED;
  if (d->tail_cout || d->pool || d->x_wpitch || d->x2_cin || d->has_residual || d->out_layout != FT_LAYOUT_NHWC) return FT_ERR_UNSUPPORTED;
  if (d->N <= 0 || d->Hi <= 0 || d->Wi <= 0) return FT_ERR_UNSUPPORTED;
  if (d->x_coff % 8 || d->x_cstride % 8 || d->y_coff % 8 || d->y_cstride % 8 || d->Cout % 64) return FT_ERR_UNSUPPORTED;
  if (d->x_cstride < d->x_coff + d->Cin || d->y_cstride < d->y_coff + d->Cout) return FT_ERR_INVALID_ARG;
  const int hw = d->Hi * d->Wi;
  if (d->stride == 2) {
    // one image per workgroup, its whole INPUT map (<= 256 pixels) resident 256 channels at a time (conv3x3s2_direct_kernel)
    static const bool no_s2 = getenv("FT_CD_NO_S2") != nullptr;                                   // dev A/B
    if (no_s2 || d->Ho != (d->Hi + 1) / 2 || d->Wo != (d->Wi + 1) / 2) return FT_ERR_UNSUPPORTED;
    if (d->Cin != 512) return FT_ERR_UNSUPPORTED;                // (instantiated for two 256-channel passes)
    if (hw > 256 || d->Ho * d->Wo > 64) {
      // strips of R output rows: (2 R + 1) * Wi <= 256 tile rows, R * Wo <= 64 output pixels (FlowNet's conv5 on 24 x 32: R = 3)
      static const bool no_strip = getenv("FT_CD_NO_STRIP") != nullptr;                            // dev A/B
      int R = (256 / d->Wi - 1) / 2;
      if (R * d->Wo > 64) R = 64 / d->Wo;
      if (no_strip || R < 1) return FT_ERR_UNSUPPORTED;
      const int nstrips = (d->Ho + R - 1) / R;
      const long long nwg = (long long)d->N * nstrips * (d->Cout / 64);
      if (nwg < 200 || nwg > 1024) return FT_ERR_UNSUPPORTED;
      if ((long long)d->N * hw * d->x_cstride * 2 >= (1LL << 31) || (long long)d->N * d->Ho * d->Wo * d->y_cstride * 2 >= (1LL << 31))
        return FT_ERR_UNSUPPORTED;
      *out = C3Plan{(R * d->Wo + 31) / 32, d->Cin / 256, 1, d->N * nstrips, d->Cout / 64, 2, R, nstrips};
      return FT_OK;
    }
    if ((long long)d->N * hw * d->x_cstride * 2 >= (1LL << 31) || (long long)d->N * d->Ho * d->Wo * d->y_cstride * 2 >= (1LL << 31))
      return FT_ERR_UNSUPPORTED;
    // two images per workgroup (conv3x3s2p_direct_kernel: half the weight stream per output pixel) where that still gives about one
    // workgroup per CU; else one image per workgroup
    static const bool no_pair = getenv("FT_CD_NO_S2P") != nullptr;                                 // dev A/B
    const int npo = d->Ho * d->Wo;
    if (!no_pair && 2 * npo <= 96 && 2 * hw <= 512 && (long long)((d->N + 1) / 2) * (d->Cout / 64) >= 200) {
      *out = C3Plan{(2 * npo + 31) / 32, d->Cin / 256, 2, (d->N + 1) / 2, d->Cout / 64, 2};
      return FT_OK;
    }
    *out = C3Plan{(npo + 31) / 32, d->Cin / 256, 1, d->N, d->Cout / 64, 2};
    return FT_OK;
  }
  if (d->Ho != d->Hi || d->Wo != d->Wi) return FT_ERR_UNSUPPORTED;
  if (d->Cin != 512 && d->Cin != 1024) return FT_ERR_UNSUPPORTED;   // (instantiated for the 512-plane stage and FlowNet's conv6_1)
  const long long M = (long long)d->N * hw;
  if (M * d->x_cstride * 2 >= (1LL << 31) || M * d->y_cstride * 2 >= (1LL << 31)) return FT_ERR_UNSUPPORTED;
  if (d->Cin == 512 && hw > 128) {
    // maps too large to be resident whole: strips of R output rows + a halo row on either side, (R + 2) * W <= 128 tile rows and
    // R * W <= 96 output pixels (FlowNet's conv5_1 on 12 x 16: R = 6), where that gives about one workgroup per CU or more
    static const bool no_strip = getenv("FT_CD_NO_STRIP") != nullptr;                              // dev A/B
    int R = 128 / d->Wi - 2;
    if (R * d->Wi > 96) R = 96 / d->Wi;
    if (no_strip || R < 2) return FT_ERR_UNSUPPORTED;
    const int nstrips = (d->Hi + R - 1) / R;
    const long long nwg = (long long)d->N * nstrips * (d->Cout / 64);
    if (nwg < 200 || nwg > 1024) return FT_ERR_UNSUPPORTED;
    *out = C3Plan{3, 2, 1, d->N * nstrips, d->Cout / 64, 1, R, nstrips};
    return FT_OK;
  }
  if (hw > (d->Cin == 512 ? 128 : 64)) return FT_ERR_UNSUPPORTED;  // the tile (every channel of the workgroup's images) must fit 128 KiB
  const int ipw = d->Cin == 512 ? (hw <= 96 ? 96 / hw : 1) : 1;
  const int mt = d->Cin == 512 ? (ipw * hw <= 96 ? 3 : 4) : 2;
  *out = C3Plan{mt, d->Cin / 256, ipw, (d->N + ipw - 1) / ipw, d->Cout / 64, 1};
  return FT_OK;
}

template <int MT, int NLD>
static int c3s2p_launch(const C3Params& p, hipStream_t s) {
  auto k = conv3x3s2p_direct_kernel<MT, NLD>;
  constexpr int lds = 131072 + 256 + 2 * 64 * 4 + MT * 32 * 128;
  FT_RAISE_LDS(k, lds);
  hipLaunchKernelGGL(k, dim3(p.npt * p.ncb), dim3(256), lds, s, p);
  FT_LAUNCH_CHECK("conv3x3s2p_direct_kernel");
  return FT_OK;
}

template <int MT, int NH>
static int c3s2_launch(const C3Params& p, hipStream_t s) {
  auto k = conv3x3s2_direct_kernel<MT, NH>;
  constexpr int lds = 131072 + 512 + 2 * 64 * 4 + MT * 32 * 128;
  FT_RAISE_LDS(k, lds);
  hipLaunchKernelGGL(k, dim3(p.npt * p.ncb), dim3(256), lds, s, p);
  FT_LAUNCH_CHECK("conv3x3s2_direct_kernel");
  return FT_OK;
}

template <int MT, int SPT, int TT = MT>
static int c3_launch(const C3Params& p, hipStream_t s) {
  auto k = conv3x3_direct_kernel<MT, SPT, TT>;
  constexpr int lds = 131072 + SPT * 512 + 2 * 64 * 4 + MT * 32 * 128;
  static bool attr_done[64] = {};
  int dev = 0;
  FT_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_done[dev]) {
    FT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if (dev >= 0 && dev < 64) attr_done[dev] = true;
  }
  hipLaunchKernelGGL(k, dim3(p.npt * p.ncb), dim3(256), lds, s, p);
  FT_LAUNCH_CHECK("conv3x3_direct_kernel");
  return FT_OK;
}

// ---- weight stream: [channel block][chunk][wave][kk][i] fragments of 1 KiB -----------------------------------------------
// from the K-major packed layout of ft_conv_pack_geometry ([Cout_pad][kpad], k over x's channels then x2's).
template <int KSPLIT>
__global__ __launch_bounds__(256) void cd_pack_kernel(const half_t* __restrict__ w, uint4_t* __restrict__ out, int nchunk, int ncb,
                                                      int kpad, int cout_pad) {
  using G = CdGeom<KSPLIT>;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int total = ncb * nchunk * (G::WCHUNK / 16);
  if (idx >= total) return;
  const int lane = idx & 63;
  int f = idx >> 6;                                  // fragment index
  const int i = f & 1; f >>= 1;
  const int kk = f & 3; f >>= 2;
  const int wv = f & 3; f >>= 2;
  const int c = f % nchunk, cb = f / nchunk;
  const int l31 = lane & 31, lhi = lane >> 5;
  int co, k;
  if (KSPLIT == 1) {
    co = cb * 256 + (2 * wv + i) * 32 + cd_sigma(l31);
    k = c * 64 + kk * 16 + 8 * lhi;
  } else {
    co = cb * 64 + i * 32 + cd_sigma(l31);
    k = c * 256 + (4 * wv + kk) * 16 + 8 * lhi;
  }
  uint4_t v = {0u, 0u, 0u, 0u};
  if (co < cout_pad && k < kpad) v = *reinterpret_cast<const uint4_t*>(w + (size_t)co * kpad + k);
  out[idx] = v;
}


// ---- 1x1, K = 256, MANY pixels (layer2.0.conv1: 256 -> 128 on 196 k pixels at batch 64): weight-stationary, persistent -----
// That layer is HBM-bound (150 MB) and ran at 3.1 TB/s on the tiled kernels: a tile's life there is load -> 4 K-steps ->
// store, and the per-tile ramps do not overlap well.  Here one workgroup per CU keeps its 128 x 256 weight block in
// REGISTERS for the whole launch (each wave: 2 channel tiles x 16 K16 slices = 128 VGPRs, fragment-ordered stream) and
// walks pixel tiles of 128: the next tile's 64 KiB stream into the other half of LDS (whole 512-byte pixel rows, DMA) while
// the current one is multiplied; the results leave straight from the accumulator layout (a lane owns 16 consecutive
// channels: two adjacent 16-byte stores; the L2 merges the four pieces of a line, so these are plain, not write-through).
struct CsParams {
  const char* x;
  char* y;
  const char* ws;
  const float* scale;
  const float* shift;
  int M, ntiles;
  int x_cstride, x_coff, y_cstride, y_coff;
  int Cout, act;
  float slope;
  unsigned x_bytes, y_bytes, ws_bytes;
};

// NS = K / 16 slices; WC x WP waves, wave (wc, wp) owns CT channel tiles x PT pixel tiles:
//   K = 256: 4 waves = 2 x 2, CT = PT = 2: 128 channels x 128 pixels per workgroup (blockIdx.y = channel block)
//   K = 512: 8 waves = 8 x 1, CT = 1, PT = 2: 256 channels x 64 pixels — half of the CU's register file holds the weights
template <int NS, int WC, int WP, int CT, int PT>
__global__ __launch_bounds__(64 * WC * WP, 1) void conv1x1_stream_kernel(const CsParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = WC * WP, BP = WP * PT * 32, BN = WC * CT * 32, ROWB = NS * 32, SLOTB = BP * ROWB, LX = SLOTB / 1024 / NW;
  constexpr int RPL = 1024 / ROWB, CPR = ROWB / 16;          // rows per 1-KiB wave load (2 or 1), 16-byte chunks per row
  constexpr int NST = CT * PT * 2;                           // stores per lane per tile
  static_assert(LX * NW * 1024 == SLOTB && 2 * SLOTB <= 160 * 1024 && RPL >= 1, "shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % WC, wp = wave / WC;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int cb = blockIdx.y;
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.ws), 0, p.ws_bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;

  // the weight block of this wave: fragment (channel tile wc*CT + i of the block, slice s) = 1 KiB, tile-major
  uint4_t a[CT][NS];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int s = 0; s < NS; ++s)
      a[i][s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (unsigned)lane * 16u, ((cb * (BN / 32) + wc * CT + i) * NS + s) * 1024, 0);
  // folded BN of the lane's 16 consecutive channels of each tile
  float4_t sc[CT][4], sh[CT][4];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int ch = cb * BN + (wc * CT + i) * 32 + 16 * lhi + 4 * g4;
      const float4_t one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
      sc[i][g4] = p.scale ? *reinterpret_cast<const float4_t*>(p.scale + ch) : one;
      sh[i][g4] = p.shift ? *reinterpret_cast<const float4_t*>(p.shift + ch) : zero;
    }
  const float act_k = p.act == FT_ACT_RELU ? 0.f : (p.act == FT_ACT_LEAKY ? p.slope : 1.f);

  // loader lanes: wave load (t*NW + wave) covers RPL tile rows; XOR swizzle of the low 4 chunk bits on the source
  int l_row[LX];
  unsigned l_off[LX];
#pragma unroll
  for (int t = 0; t < LX; ++t) {
    const int piece = t * NW + wave;
    const int row = RPL == 2 ? piece * 2 + (lane >> 5) : piece / (ROWB / 1024);
    const int pos = RPL == 2 ? (lane & 31) : (piece % (ROWB / 1024)) * 64 + lane;
    l_row[t] = row;
    l_off[t] = (unsigned)((row * p.x_cstride + p.x_coff) * 2 + (((pos & ~15) | ((pos ^ row) & 15)) << 4));
  }
  auto issue = [&](int tile, int slot) {            // always LX loads: tiles past the end / rows past M read out of range
    const int m0 = tile * BP;
    const bool tile_ok = tile < p.ntiles;
    const int soff = tile_ok ? m0 * p.x_cstride * 2 : 0;
#pragma unroll
    for (int t = 0; t < LX; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr)(smem + slot * SLOTB + (t * NW + wave) * 1024), 16,
                                               (tile_ok && m0 + l_row[t] < p.M) ? l_off[t] : kOOB, soff, 0, 0);
  };
  int b_off[PT];
#pragma unroll
  for (int j = 0; j < PT; ++j) {
    const int row = (wp * PT + j) * 32 + l31;
    b_off[j] = row * ROWB + ((lhi ^ (row & 15)) << 4);
  }
  static_assert(CPR >= 32, "chunk index bits");

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // weights and tables sit in registers before the tile loads are counted
  int tile = blockIdx.x;
  issue(tile, 0);
  for (int k = 0; tile < p.ntiles; ++k, tile += gridDim.x) {
    const int slot = k & 1;
    // this tile has landed (this wave's share): behind it only the NST stores of the previous tile may fly
    if (k == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
    asm volatile("s_barrier" ::: "memory");          // everyone's share has; everyone is done reading the other half
    issue(tile + gridDim.x, slot ^ 1);
    float16_t acc[CT][PT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int j = 0; j < PT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const char* xb = smem + slot * SLOTB;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      uint4_t fx[PT];
#pragma unroll
      for (int j = 0; j < PT; ++j) fx[j] = *reinterpret_cast<const uint4_t*>(xb + (b_off[j] ^ (s << 5)));
#pragma unroll
      for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, a[i][s]), __builtin_bit_cast(half8_t, fx[j]),
                                                             acc[i][j], 0, 0, 0);
    }
    // the loads of the next tile were issued before these stores: at the next wait they are the older operations
    const int m0 = tile * BP;
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int j = 0; j < PT; ++j) {
        const int m = m0 + (wp * PT + j) * 32 + l31;
        const unsigned vo = m < p.M ? (unsigned)((m * p.y_cstride + p.y_coff + cb * BN + (wc * CT + i) * 32 + 16 * lhi) * 2) : kOOB;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          half8_t o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int r = h * 8 + e;
            const float v = acc[i][j][r] * sc[i][r >> 2][r & 3] + sh[i][r >> 2][r & 3];
            o[e] = (half_t)act_mul(v, act_k);
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, o), rsrc_y, vo, h * 16, 0);
        }
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the look-ahead loads of the last trip must not outlive the workgroup's LDS
#endif
}

// fragment-ordered weights for conv1x1_stream_kernel: [channel tile Cout / 32][slice ns][lane] x 16 bytes
__global__ __launch_bounds__(256) void cs_pack_kernel(const half_t* __restrict__ w, uint4_t* __restrict__ out, int ntile, int ns, int kpad,
                                                      int cout_pad) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= ntile * ns * 64) return;
  const int lane = idx & 63;
  const int f = idx >> 6;
  const int sl = f % ns, tl = f / ns;
  const int co = tl * 32 + cd_sigma(lane & 31), k = sl * 16 + 8 * (lane >> 5);
  uint4_t v = {0u, 0u, 0u, 0u};
  if (co < cout_pad && k < kpad) v = *reinterpret_cast<const uint4_t*>(w + (size_t)co * kpad + k);
  out[idx] = v;
}

struct CdPlan {
  int ksplit;          // 1 or 4; 0 = the persistent weight-stationary form (conv1x1_stream_kernel)
  int nc1, nc2, npt, ncb;
};

static int cd_plan(const ft_conv_desc* d, CdPlan* out) {
  if (!d) return FT_ERR_INVALID_ARG;
  if (d->act == FT_ACT_LEAKY && !(d->slope >= 0.f && d->slope <= 1.f)) return FT_ERR_UNSUPPORTED;   // epilogues use max(v, k * v): valid for 0 <= k <= 1 only
  if (d->dtype != FT_F16 || d->transposed) return FT_ERR_UNSUPPORTED;
  const bool taps = d->kh != 1 || d->kw != 1;
  if (taps) {        // the gather form: 3x3, stride 1 or 2, pad 1, no second input, no residual
    if (d->kh != 3 || d->kw != 3 || d->pad != 1 || (d->stride != 1 && d->stride != 2) || d->x2_cin || d->has_residual) return FT_ERR_UNSUPPORTED;
    if (d->Ho != (d->Hi + 2 - 3) / d->stride + 1 || d->Wo != (d->Wi + 2 - 3) / d->stride + 1) return FT_ERR_INVALID_ARG;
  } else if (d->stride != 1 || d->pad != 0) return FT_ERR_UNSUPPORTED;
  if (d->tail_cout || d->pool || d->x_wpitch || d->out_layout != FT_LAYOUT_NHWC) return FT_ERR_UNSUPPORTED;
  if (d->N <= 0 || d->Hi <= 0 || d->Wi <= 0 || (!taps && (d->Ho != d->Hi || d->Wo != d->Wi))) return FT_ERR_UNSUPPORTED;
  if (d->x_coff % 8 || d->x_cstride % 8 || d->y_coff % 8 || d->y_cstride % 8 || d->Cout % 64) return FT_ERR_UNSUPPORTED;
  if (d->x_cstride < d->x_coff + d->Cin || d->y_cstride < d->y_coff + d->Cout) return FT_ERR_INVALID_ARG;
  if (d->has_residual && (d->x2_cin || d->res_coff % 8 || d->res_cstride % 8)) return FT_ERR_UNSUPPORTED;
  if (d->x2_cin && (d->x2_coff % 8 || d->x2_cstride % 8 || d->x2_stride < 1 || d->Ho != (d->x2_hi - 1) / d->x2_stride + 1 ||
                    d->Wo != (d->x2_wi - 1) / d->x2_stride + 1)) return FT_ERR_UNSUPPORTED;
  const long long M = (long long)d->N * d->Ho * d->Wo;
  const long long lim = 1LL << 31;
  if ((long long)d->N * d->Hi * d->Wi * d->x_cstride * 2 >= lim || M * d->y_cstride * 2 >= lim || (d->has_residual && M * d->res_cstride * 2 >= lim) ||
      (d->x2_cin && (long long)d->N * d->x2_hi * d->x2_wi * d->x2_cstride * 2 >= lim)) return FT_ERR_UNSUPPORTED;
  // many pixels, short K, plain epilogue: the weight-stationary persistent form (the HBM-bound ResNet layer2.0.conv1: K = 256
  // in 128-channel blocks).  FT_CD_STATIONARY=2 (dev) also sends K = 512 -> 256 layers with >= 32768 pixels (layer3.0.conv1)
  // to the 8-wave form: measured 24.7 us against 22.3 us for the K-split kernel, and 1.8 % slower over the whole pose step.
  static const int cs_mode = getenv("FT_CD_STATIONARY") ? atoi(getenv("FT_CD_STATIONARY")) : 1;
  const bool cs256 = d->Cin == 256 && d->Cout % 128 == 0 && M > 65536, cs512 = cs_mode == 2 && d->Cin == 512 && d->Cout == 256 && M >= 32768;
  if (!taps && cs_mode != 0 && !d->x2_cin && !d->has_residual && (cs256 || cs512)) {
    const int bp = cs256 ? 128 : 64;
    *out = CdPlan{0, d->Cin / 64, 0, (int)((M + bp - 1) / bp), cs256 ? d->Cout / 128 : 1};
    return FT_OK;
  }
  if (!taps && M > 65536) return FT_ERR_UNSUPPORTED;
  const int npt = (int)((M + 95) / 96);
  static const int force = getenv("FT_CD_KSPLIT") ? atoi(getenv("FT_CD_KSPLIT")) : 0;
  const bool a_ok = d->Cout % 256 == 0 && d->Cin % 64 == 0 && d->x2_cin % 64 == 0;
  const bool b_ok = d->Cin % 256 == 0 && d->x2_cin % 256 == 0;
  if (!a_ok && !b_ok) return FT_ERR_UNSUPPORTED;
  // N-tile 256 unless that leaves most CUs idle and the 64-wide, K-split form is available.  Only for narrow layers: with many
  // output-channel blocks every one of them re-reads the whole pixel tile (measured at 1728 pixels, 512 -> 2048: 26 us K-split
  // vs 18 us in conv_igemm_dma_kernel)
  int ks = a_ok ? 1 : 4;
  if (a_ok && b_ok && (long long)npt * (d->Cout / 256) < 160 && d->Cout <= 512) ks = 4;
  if (force == 1 && a_ok) ks = 1;
  if (force == 4 && b_ok) ks = 4;
  const int ck = ks == 1 ? 64 : 256;
  *out = CdPlan{ks, d->kh * d->kw * (d->Cin / ck), d->x2_cin / ck, npt, d->Cout / (ks == 1 ? 256 : 64)};
  return FT_OK;
}

template <int KSPLIT, bool HAS_RES, int NCH, bool TAPS = false>
static int cd_launch(const CdParams& p, hipStream_t s) {
  auto k = conv_direct_kernel<KSPLIT, HAS_RES, NCH, TAPS>;
  constexpr int lds = CdGeom<KSPLIT>::LDS_BYTES + (FT_CD_L2_TOUCH ? 1024 : 0);     // (+ the L2 touch's scratch)
  static_assert(lds <= 163840, "LDS map");
  static bool attr_done[64] = {};
  int dev = 0;
  FT_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_done[dev]) {
    FT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if (dev >= 0 && dev < 64) attr_done[dev] = true;
  }
  hipLaunchKernelGGL(k, dim3(p.npt * p.ncb), dim3(256), lds, s, p);
  FT_LAUNCH_CHECK("conv_direct_kernel");
  return FT_OK;
}

#ifndef FT_CD_TAPS72
#define FT_CD_TAPS72 1
#endif
template <int KSPLIT>
static int cd_dispatch_taps(const CdParams& p, hipStream_t s) {
  if (KSPLIT == 1) {
    switch (p.nc1) {   // 3x3 on 256 channels in 64-channel chunks; longer walks take the run-time loop
      case 36: return cd_launch<KSPLIT, false, 36, true>(p, s);
#if FT_CD_TAPS72
      case 72: return cd_launch<KSPLIT, false, 72, true>(p, s);     // 3x3 on 512 channels (FlowNet conv4_1 / conv5 / conv5_1)
#endif
      default: return cd_launch<KSPLIT, false, 0, true>(p, s);
    }
  }
  switch (p.nc1) {     // 3x3 on 256 / 512 channels in 256-channel chunks (ResNet layer3.0 / layer4.0 conv2, FlowNet conv4 .. conv5_1)
    case 9: return cd_launch<KSPLIT, false, 9, true>(p, s);
    case 18: return cd_launch<KSPLIT, false, 18, true>(p, s);
    default: return cd_launch<KSPLIT, false, 0, true>(p, s);
  }
}

template <int KSPLIT, bool HAS_RES>
static int cd_dispatch(const CdParams& p, hipStream_t s) {
  static const bool no_unroll = getenv("FT_CD_NO_UNROLL") != nullptr;     // dev A/B: the run-time-loop form
  const int n = no_unroll ? 0 : p.nc1 + p.nc2;
  if (KSPLIT == 1) {
    switch (n) {     // K = 256 / 384 / 512 / 768 / 1024 / 1536 (ResNet layer2-4 1x1 convs and their K-concatenated block exits)
      case 4: return cd_launch<KSPLIT, HAS_RES, 4>(p, s);
      case 6: return cd_launch<KSPLIT, HAS_RES, 6>(p, s);
      case 8: return cd_launch<KSPLIT, HAS_RES, 8>(p, s);
      case 12: return cd_launch<KSPLIT, HAS_RES, 12>(p, s);
      case 16: return cd_launch<KSPLIT, HAS_RES, 16>(p, s);
      case 24: return cd_launch<KSPLIT, HAS_RES, 24>(p, s);
      default: return cd_launch<KSPLIT, HAS_RES, 0>(p, s);
    }
  }
  switch (n) {       // K = 512 (ResNet layer3.0.conv1) / 1024 / 2048 in 256-channel chunks
    case 2: return cd_launch<KSPLIT, HAS_RES, 2>(p, s);
    case 4: return cd_launch<KSPLIT, HAS_RES, 4>(p, s);
    case 8: return cd_launch<KSPLIT, HAS_RES, 8>(p, s);
    default: return cd_launch<KSPLIT, HAS_RES, 0>(p, s);
  }
}

}  // namespace
}  // namespace ft

extern "C" int ft_conv_direct_supported(const ft_conv_desc* d) {
  ft::CdPlan pl;
  ft::C3Plan p3;
  ft::WsPlan pw;
  if (d && d->kh == 5) return ft::ws_plan(d, &pw);                        // 5x5 / stride 2 on 64 channels: register-stationary (conv_wstat.hip)
  if (d && d->kh == 3 && ft::c3_plan(d, &p3) == FT_OK) return FT_OK;     // whole small maps; other 3x3s: the gather form
  return ft::cd_plan(d, &pl);
}

// Which weight-stream layout ft_conv_direct_pack builds / ft_conv_direct_fwd expects for `d`: the form (whole-map 3x3, gather
// or 1x1 with K split 1 / 4, weight-stationary) and its block counts.  The form depends on the pixel count, so one layer can
// need more than one stream (one per id) when it runs at several batch sizes; all of them have the same byte count.
extern "C" int ft_conv_direct_stream_id(const ft_conv_desc* d) {
  ft::CdPlan pl;
  ft::C3Plan p3;
  ft::WsPlan pw;
  if (d && d->kh == 5) return ft::ws_plan(d, &pw) == FT_OK ? (0x50000000 | pw.ncg) : -1;
  if (d && d->kh == 3 && ft::c3_plan(d, &p3) == FT_OK) return 0x40000000 | (p3.stride == 2 ? (p3.ipw == 2 ? 0x30000000 : 0x10000000) : 0) | (p3.ncb << 8) | p3.spt;
  if (ft::cd_plan(d, &pl) != FT_OK) return -1;
  return ((pl.ksplit + 1) << 24) | ((pl.nc1 + pl.nc2) << 12) | pl.ncb;
}

extern "C" long long ft_conv_direct_weight_bytes(const ft_conv_desc* d) {
  ft::CdPlan pl;
  ft::C3Plan p3;
  ft::WsPlan pw;
  if (d && d->kh == 5) return ft::ws_plan(d, &pw) == FT_OK ? ft::ws_weight_bytes(pw) : 0;
  if (d && d->kh == 3 && ft::c3_plan(d, &p3) == FT_OK) return (long long)p3.ncb * 4 * 9 * p3.spt * 8192;
  if (ft::cd_plan(d, &pl) != FT_OK) return 0;
  if (pl.ksplit == 0) return (long long)d->Cout * d->Cin * 2;
  return (long long)pl.ncb * (pl.nc1 + pl.nc2) * 32768;
}

extern "C" int ft_conv_direct_pack(const ft_conv_desc* d, const void* w_packed, int kpad, int cout_pad, void* wstream, ft_stream_t stream) {
  using namespace ft;
  C3Plan p3;
  if (d && d->kh == 5) {
    WsPlan pw;
    const int st = ws_plan(d, &pw);
    return st != FT_OK ? st : ws_pack(d, pw, w_packed, kpad, cout_pad, wstream, as_stream(stream));
  }
  if (d && d->kh == 3 && c3_plan(d, &p3) == FT_OK) {
    if (!w_packed || !wstream || kpad < 9 * d->Cin || cout_pad < d->Cout) return FT_ERR_INVALID_ARG;
    const int total = p3.ncb * 4 * 9 * p3.spt * 512;
    const dim3 pg(ceil_div(total, 256));
    const half_t* wsrc = static_cast<const half_t*>(w_packed);
    uint4_t* wdst = static_cast<uint4_t*>(wstream);
    if (p3.stride == 2 && p3.ipw == 2) hipLaunchKernelGGL(c3s2p_pack_kernel, pg, dim3(256), 0, as_stream(stream), wsrc, wdst, p3.ncb, kpad, cout_pad);
    else if (p3.stride == 2) hipLaunchKernelGGL(c3s2_pack_kernel<2>, pg, dim3(256), 0, as_stream(stream), wsrc, wdst, p3.ncb, kpad, cout_pad);
    else if (p3.spt == 4) hipLaunchKernelGGL(c3_pack_kernel<4>, pg, dim3(256), 0, as_stream(stream), wsrc, wdst, p3.ncb, kpad, cout_pad);
    else hipLaunchKernelGGL(c3_pack_kernel<2>, pg, dim3(256), 0, as_stream(stream), wsrc, wdst, p3.ncb, kpad, cout_pad);
    FT_LAUNCH_CHECK("c3_pack_kernel");
    return FT_OK;
  }
  CdPlan pl;
  const int st = cd_plan(d, &pl);
  if (st != FT_OK) return st;
  if (!w_packed || !wstream || kpad < d->kh * d->kw * d->Cin + d->x2_cin || cout_pad < d->Cout) return FT_ERR_INVALID_ARG;
  const int nchunk = pl.nc1 + pl.nc2;
  const int total = pl.ncb * nchunk * 2048;
  hipStream_t s = as_stream(stream);
  if (pl.ksplit == 0) {
    const int ntile = d->Cout / 32, ns = d->Cin / 16;
    hipLaunchKernelGGL(cs_pack_kernel, dim3(ceil_div(ntile * ns * 64, 256)), dim3(256), 0, s, static_cast<const half_t*>(w_packed),
                       static_cast<uint4_t*>(wstream), ntile, ns, kpad, cout_pad);
    FT_LAUNCH_CHECK("cs_pack_kernel");
    return FT_OK;
  }
  if (pl.ksplit == 1)
    hipLaunchKernelGGL(cd_pack_kernel<1>, dim3(ceil_div(total, 256)), dim3(256), 0, s, static_cast<const half_t*>(w_packed),
                       static_cast<uint4_t*>(wstream), nchunk, pl.ncb, kpad, cout_pad);
  else
    hipLaunchKernelGGL(cd_pack_kernel<4>, dim3(ceil_div(total, 256)), dim3(256), 0, s, static_cast<const half_t*>(w_packed),
                       static_cast<uint4_t*>(wstream), nchunk, pl.ncb, kpad, cout_pad);
  FT_LAUNCH_CHECK("cd_pack_kernel");
  return FT_OK;
}

extern "C" int ft_conv_direct_fwd(const ft_conv_desc* d, const void* x, const void* wstream, const float* scale, const float* shift,
                                  const void* residual, void* y, ft_stream_t stream) {
  using namespace ft;
  C3Plan p3;
  if (d && d->kh == 5) {
    WsPlan pw;
    const int st = ws_plan(d, &pw);
    return st != FT_OK ? st : ws_launch(d, pw, x, wstream, scale, shift, y, as_stream(stream));
  }
  if (d && d->kh == 3 && c3_plan(d, &p3) == FT_OK) {
    if (!x || !wstream || !y) return FT_ERR_INVALID_ARG;
    C3Params q{};
    q.x = static_cast<const char*>(x);
    q.y = static_cast<char*>(y);
    q.ws = static_cast<const char*>(wstream);
    q.scale = scale;
    q.shift = shift;
    q.HW = d->Hi * d->Wi; q.W = d->Wi; q.H = d->Hi; q.ipw = p3.ipw;
    q.Ho = p3.stride == 2 ? d->Ho : 0; q.Wo = p3.stride == 2 ? d->Wo : 0;
    q.strip_rows = p3.strip_rows; q.nstrips = p3.nstrips;
    q.M = p3.stride == 2 ? d->N * d->Ho * d->Wo : d->N * q.HW;
    q.x_cstride = d->x_cstride; q.x_coff = d->x_coff; q.y_cstride = d->y_cstride; q.y_coff = d->y_coff;
    q.Cout = d->Cout; q.act = d->act; q.slope = d->slope;
    q.npt = p3.npt; q.ncb = p3.ncb;
    q.x_bytes = (unsigned)((size_t)d->N * q.HW * d->x_cstride * 2);
    q.y_bytes = (unsigned)((size_t)q.M * d->y_cstride * 2);
    q.ws_bytes = (unsigned)ft_conv_direct_weight_bytes(d);
    q.wmajor = cd_wmajor((long long)d->N * q.HW * d->Cin * 2, q.ws_bytes, q.npt, q.ncb);
    if (p3.stride == 2 && p3.ipw == 2) {
      const int nld = (2 * q.HW * 256 + 4095) / 4096;            // 1-KiB tile pieces per wave
      if (p3.mt <= 2) return nld <= 16 ? c3s2p_launch<2, 16>(q, as_stream(stream)) : c3s2p_launch<2, 32>(q, as_stream(stream));
      return nld <= 24 ? c3s2p_launch<3, 24>(q, as_stream(stream)) : c3s2p_launch<3, 32>(q, as_stream(stream));
    }
    if (p3.stride == 2) return p3.mt == 1 ? c3s2_launch<1, 2>(q, as_stream(stream)) : c3s2_launch<2, 2>(q, as_stream(stream));
    if (p3.strip_rows) return c3_launch<3, 2, 4>(q, as_stream(stream));
    if (p3.spt == 4) return c3_launch<2, 4>(q, as_stream(stream));
    return p3.mt == 3 ? c3_launch<3, 2>(q, as_stream(stream)) : c3_launch<4, 2>(q, as_stream(stream));
  }
  CdPlan pl;
  const int st = cd_plan(d, &pl);
  if (st != FT_OK) return st;
  if (!x || !wstream || !y || ((d->has_residual || d->x2_cin) && !residual)) return FT_ERR_INVALID_ARG;
  if (pl.ksplit == 0) {
    CsParams q{};
    q.x = static_cast<const char*>(x);
    q.y = static_cast<char*>(y);
    q.ws = static_cast<const char*>(wstream);
    q.scale = scale;
    q.shift = shift;
    q.M = d->N * d->Ho * d->Wo;
    q.ntiles = pl.npt;
    q.x_cstride = d->x_cstride; q.x_coff = d->x_coff; q.y_cstride = d->y_cstride; q.y_coff = d->y_coff;
    q.Cout = d->Cout; q.act = d->act; q.slope = d->slope;
    q.x_bytes = (unsigned)((size_t)q.M * d->x_cstride * 2);
    q.y_bytes = (unsigned)((size_t)q.M * d->y_cstride * 2);
    q.ws_bytes = (unsigned)ft_conv_direct_weight_bytes(d);
    constexpr int lds = 2 * 65536;
    int gx = 256 / pl.ncb;
    gx = gx < 1 ? 1 : (gx > pl.npt ? pl.npt : gx);
    if (d->Cin == 256) {
      auto k = conv1x1_stream_kernel<16, 2, 2, 2, 2>;
      FT_RAISE_LDS(k, lds);
      hipLaunchKernelGGL(k, dim3(gx, pl.ncb), dim3(256), lds, as_stream(stream), q);
    } else {
      auto k = conv1x1_stream_kernel<32, 8, 1, 1, 2>;
      FT_RAISE_LDS(k, lds);
      hipLaunchKernelGGL(k, dim3(gx, 1), dim3(512), lds, as_stream(stream), q);
    }
    FT_LAUNCH_CHECK("conv1x1_stream_kernel");
    return FT_OK;
  }
  CdParams p{};
  p.x = static_cast<const char*>(x);
  p.y = static_cast<char*>(y);
  p.ws = static_cast<const char*>(wstream);
  p.scale = scale;
  p.shift = shift;
  p.M = d->N * d->Ho * d->Wo;
  p.nc1 = pl.nc1; p.nc2 = pl.nc2; p.npt = pl.npt; p.ncb = pl.ncb;
  p.x_cstride = d->x_cstride; p.x_coff = d->x_coff;
  p.HqWq = d->Ho * d->Wo; p.Wq = d->Wo;
  p.y_cstride = d->y_cstride; p.y_coff = d->y_coff;
  p.Cout = d->Cout; p.act = d->act; p.slope = d->slope;
  p.x_bytes = (unsigned)((size_t)d->N * d->Hi * d->Wi * d->x_cstride * 2);
  p.y_bytes = (unsigned)((size_t)p.M * d->y_cstride * 2);
  p.ws_bytes = (unsigned)ft_conv_direct_weight_bytes(d);
  const bool taps = d->kh != 1;
  p.cpt = d->Cin / (pl.ksplit == 1 ? 64 : 256); p.kw = d->kw; p.Hi = d->Hi; p.Wi = d->Wi; p.stride = d->stride; p.pad = d->pad;
  if (d->x2_cin) {
    p.x2 = static_cast<const char*>(residual);
    p.x2_hi = d->x2_hi; p.x2_wi = d->x2_wi; p.x2_cstride = d->x2_cstride; p.x2_coff = d->x2_coff; p.x2_stride = d->x2_stride;
    p.x2_bytes = (unsigned)((size_t)d->N * d->x2_hi * d->x2_wi * d->x2_cstride * 2);
  }
  if (d->has_residual) {
    p.res = static_cast<const char*>(residual);
    p.res_cstride = d->res_cstride; p.res_coff = d->res_coff;
    p.res_bytes = (unsigned)((size_t)p.M * d->res_cstride * 2);
  }
  static const int dbg = getenv("FT_CD_DBG") ? atoi(getenv("FT_CD_DBG")) : 0;
  p.dbg = dbg;
  p.wmajor = cd_wmajor((long long)d->N * d->Hi * d->Wi * d->Cin * 2 + (d->x2_cin ? (long long)d->N * d->x2_hi * d->x2_wi * d->x2_cin * 2 : 0), p.ws_bytes,
                       p.npt, p.ncb);
  hipStream_t s = as_stream(stream);
  if (taps) return pl.ksplit == 1 ? cd_dispatch_taps<1>(p, s) : cd_dispatch_taps<4>(p, s);
  if (pl.ksplit == 1) return d->has_residual ? cd_dispatch<1, true>(p, s) : cd_dispatch<1, false>(p, s);
  return d->has_residual ? cd_dispatch<4, true>(p, s) : cd_dispatch<4, false>(p, s);
}
