// conv5x5s2_wstat_kernel — the REGISTER-STATIONARY form of a 5x5 / stride 2 / pad 2 convolution on 64 input channels
// (FlowNetS / FlowNetC `conv2`, lib/flownet/networks/FlowNetS.py:21 / FlowNetC.py:19: conv(64, 128, kernel_size=5, stride=2),
// submodules.py:7-29), fp16 storage / fp32 accumulate, folded scale / shift + activation fused.  gfx950 only.
//
// Why a form of its own.  The layer has K = 25 taps x 64 channels = 1600 per output and only 128 outputs per pixel: as an
// implicit GEMM every 256-pixel tile pulls 25 x (256 + 128) x 128 B through L2 -> LDS (940 MB per 16 pairs: the launch was
// L2 -> LDS bound at ~120 us), and a halo kernel that keeps the input patch resident still streams the 410 KB weight set per
// patch.  The weights are the operand worth keeping: 64 output channels x 1600 x 2 B = 205 KB is 40 % of a CU's register
// file.  So: ONE persistent 8-wave workgroup per CU holds the weights of one 64-channel output group in registers for its
// whole life (wave (c, q): channel tile c of 32, K quarter q = 25 of the 100 k-steps of 16 -> 100 VGPRs), walks 8 x 8 output
// patches whose 19 x 19 x 64 input patch is LDS-DMA'd (double-buffered, out-of-image pixels = out-of-range loads = zeros)
// one patch ahead, multiplies B fragments read from the patch against its A registers, and the four K quarters of a channel
// tile meet in a 64-KiB fp32 exchange buffer in LDS, from which all 512 threads finish the patch (8 channels of a pixel per
// thread: 16-byte stores, 128 contiguous bytes per 8 lanes).  L2 -> LDS traffic: the input once per output group (x 1.41 halo).
//
// LDS map: [patch 0: 48 KiB][patch 1: 48 KiB][exchange: 64 KiB] = 160 KiB.
//  * patch: see the layout note at ws_read (256-byte pixel pairs, 16 slots XOR-ed with a (column, row) key: conflict-free
//    ds_read_b128 fragments);
//  * exchange: [quarter][group of 4 channels G = 0..15][pixel ^ (G >> 1)][4 floats]: writes are lane-linear per (register
//    group, tile), the eight lanes that finish one pixel read eight distinct 16-byte columns.
//
// What was measured on the way (FlowNet2S conv2, 16 x 192 x 256 x 64 -> 96 x 128 x 128; implicit GEMM 122-125 us):
//   first correct version (reads left to the compiler's schedule, finish after a second barrier)      98-100 us
//   finish / next patch's loads / tile-0 exchange writes hung between the MFMAs, tiles one after the other   92-93 us
//   + conflict-free patch layout, every in-loop LDS access as inline asm (hipcc had put `s_waitcnt vmcnt(0)` — the whole
//     next patch — in front of compiler-visible LDS accesses and of the first in-loop use of preloaded registers)   86-87 us
//   dead ends: FOUR waves (one per SIMD, both channel tiles per wave: 37 % less LDS traffic) 100-110 us — 64-68 cycles per
//   MFMA whatever the number of accumulator chains; both pixel tiles interleaved per k-step in the 8-wave form: slower than the
//   implicit GEMM (8.6 k cycles per patch against 6.0 k).  tools/dev/ws_phases.py prints the per-phase cycle sums.
#include "conv_wstat.h"

#include <stdlib.h>

#include "conv_common.h"

namespace ft {

constexpr int kWsPH = 19;                        // input patch rows / used columns: (8 - 1) * 2 + 5
constexpr int kWsPairs = 10;                     // a patch row = 10 pixel PAIRS of 256 B (column 19 is padding)
constexpr int kWsPieces = 48;                    // 1-KiB wave loads per patch buffer (19 * 10 * 256 B = 47.5 KiB; 6 per wave)
constexpr int kWsPatchB = kWsPieces * 1024;
constexpr int kWsPartB = 4 * 16 * 64 * 16;
constexpr int kWsLds = 2 * kWsPatchB + kWsPartB;
constexpr int kWsNJ = 25;                        // k-steps per wave
constexpr int kWsTileOff = 4 * 2 * kWsPairs * 256;   // second pixel tile of the patch: four output rows further down

__host__ __device__ constexpr int ws_sigma(int r) { return 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3); }   // as cd_sigma: a lane owns 16 consecutive channels

struct WsParams {
  const char* x;
  char* y;
  const char* ws;
  const float* scale;
  const float* shift;
  int H, W, Ho, Wo;
  int x_cstride, x_coff, y_cstride, y_coff;
  int act;
  float slope;
  int ncg, tiles_x, tiles_y, npatches, ppp;      // ppp: patches per workgroup pair
  unsigned x_bytes, y_bytes, ws_bytes;
  int dbg;
};

// The MFMA walk of one patch = 50 half-steps: k-steps 0..24 of pixel tile 0, then of pixel tile 1 (the A registers serve both).
// B fragments are read kWsAhead half-steps before their MFMA, by hand: left to itself the compiler emitted read / wait /
// multiply on one fragment register set (the LDS latency in front of every MFMA).  The reads are inline asm, the counted
// lgkmcnt waits carry the fragment register as an in-out operand so no MFMA can move above its wait (LDS operations return in
// order: other LDS traffic between them only makes a count conservative).  `hook(hs)` runs after half-step hs: the kernel hangs
// everything else of the patch loop there — the next patch's LDS-DMA loads, the previous patch's finish, this patch's exchange
// writes of tile 0 — so that none of it waits in front of the matrix pipe.  `mid()` sits between the tiles.
//
// Patch layout (conflict-free for ds_read_b128's 16-lane groups {0-3, 12-15, 20-27}, ...: a B fragment's 32 pixels are 8
// columns two apart in 4 rows two apart, ALL of one column parity — with 128-byte pixels in row-major order every lane sat in
// the same 128-byte half of the 64 banks, a 2-way conflict on every read): input pixel (r, cc) lives in the 256-byte PAIR
// r * 10 + cc / 2, at 16-byte slot ((cc & 1) * 8 + chunk) ^ key, key = (cc / 2 & 3) | ((r / 2 & 3) << 2): the 16 lanes of a
// group differ in (cc / 2 & 3, r / 2 & 3), so they cover the 16 slots of a bank row.
constexpr int kWsAhead = 5, kWsHS = 2 * kWsNJ;
template <int Q, int HS>
__device__ __forceinline__ void ws_read(unsigned abase, const int (&hk)[3][3], uint4_t& b) {
  constexpr int J = HS % kWsNJ, s = Q * kWsNJ + J, tap = s >> 2, c16 = s & 3, ty = tap / 5, tx = tap % 5;
  constexpr int toff = (ty * kWsPairs + (tx >> 1)) * 256 + (HS / kWsNJ) * kWsTileOff;
  constexpr int lc = (((tx & 1) << 3) | (c16 << 1)) << 4;
  const unsigned a = abase + (unsigned)(lc ^ hk[tx >> 1][ty >> 1]);
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b) : "v"(a), "n"(toff));
}
// the finish's exchange reads (quarters 2H, 2H + 1 of the thread's octet) and their sum, NEWER LDS operations later
template <int H>
__device__ __forceinline__ void ws_ex_rd(unsigned addr, float4_t (&fr)[4]) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[0]) : "v"(addr), "n"(H * 32768));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[1]) : "v"(addr), "n"(H * 32768 + 1024));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[2]) : "v"(addr), "n"(H * 32768 + 16384));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[3]) : "v"(addr), "n"(H * 32768 + 16384 + 1024));
}
template <int H, int NEWER>
__device__ __forceinline__ void ws_ex_sum(float4_t (&fr)[4], float (&v)[8]) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fr[0]), "+v"(fr[1]), "+v"(fr[2]), "+v"(fr[3]) : "n"(NEWER));
#pragma unroll
  for (int e2 = 0; e2 < 2; ++e2)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if constexpr (H == 0) v[e2 * 4 + e] = fr[e2][e] + fr[2 + e2][e];
      else v[e2 * 4 + e] = (v[e2 * 4 + e] + fr[e2][e]) + fr[2 + e2][e];
    }
}
__device__ __forceinline__ void ws_ex_wr(unsigned addr, float4_t u) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(u) : "memory"); }

template <int Q, typename Hook, typename Mid>
__device__ __forceinline__ void ws_mfma(const uint4_t (&wt)[kWsNJ], unsigned abase, const int (&hk)[3][3], float16_t (&acc)[2], Hook&& hook, Mid&& mid) {
  uint4_t b[kWsAhead + 1];
  static_for<kWsAhead>([&](auto hc) { ws_read<Q, decltype(hc)::value>(abase, hk, b[decltype(hc)::value]); });
  static_for<kWsHS>([&](auto hc) {
    constexpr int hs = decltype(hc)::value, j = hs % kWsNJ, t = hs / kWsNJ;
    if constexpr (hs == kWsNJ) mid();
    if constexpr (hs + kWsAhead < kWsHS) ws_read<Q, hs + kWsAhead>(abase, hk, b[(hs + kWsAhead) % (kWsAhead + 1)]);
    constexpr int sl = hs % (kWsAhead + 1);
    constexpr int behind = kWsHS - 1 - hs < kWsAhead ? kWsHS - 1 - hs : kWsAhead;     // reads issued after half-step hs's
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(b[sl]) : "n"(behind));
    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, wt[j]), __builtin_bit_cast(half8_t, b[sl]), acc[t], 0, 0, 0);
    hook(hc);
  });
}

__global__ __launch_bounds__(512, 1) void conv5x5s2_wstat_kernel(const WsParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = wave & 1, q = wave >> 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  // workgroup -> (XCD, output group, pair): the ncg workgroups of a pair walk the same patches at the same time on one XCD
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ppx = 32 / p.ncg;
  if (slot >= ppx * p.ncg) return;
  const int cg = slot % p.ncg;
  const int pair = xcd * ppx + slot / p.ncg;
  const int p0 = pair * p.ppp;
  const int p1 = p0 + p.ppp < p.npatches ? p0 + p.ppp : p.npatches;
  if (p0 >= p1) return;
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.ws), 0, p.ws_bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;

  // this wave's weights: [group][channel tile][K quarter][k-step][lane] x 16 B
  uint4_t wt[kWsNJ];
  {
    const int wbase = (((cg * 2 + c) * 4 + q) * kWsNJ) * 1024;
#pragma unroll
    for (int j = 0; j < kWsNJ; ++j) wt[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (unsigned)lane * 16u, wbase + j * 1024, 0);
  }
  // finishing thread: pixel tid >> 3 of the patch, channels cg * 64 + (tid & 7) * 8 .., their folded scales / shifts in registers
  const int e_pix = tid >> 3, e_oct = tid & 7;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = p.scale ? p.scale[cg * 64 + e_oct * 8 + e] : 1.f;
    sh[e] = p.shift ? p.shift[cg * 64 + e_oct * 8 + e] : 0.f;
  }
  const float act_k = p.act == FT_ACT_RELU ? 0.f : (p.act == FT_ACT_LEAKY ? p.slope : 1.f);

  // loader lanes: wave load t * 8 + wave fills 16-byte slots S = piece * 64 + lane of the patch buffer
  constexpr int NLD = 6;
  int l_rel[NLD], l_rc[NLD];
#pragma unroll
  for (int t = 0; t < NLD; ++t) {
    const int S = (t * 8 + wave) * 64 + lane;
    const int pr = S >> 4, phys = S & 15;
    const int row = pr / kWsPairs, cp = pr - row * kWsPairs;
    const int logical = phys ^ ((cp & 3) | (((row >> 1) & 3) << 2));
    const int col = cp * 2 + (logical >> 3), chunk = logical & 7;
    l_rel[t] = ((row * p.W + col) * p.x_cstride + p.x_coff) * 2 + chunk * 16;
    l_rc[t] = ((row < kWsPH && col < kWsPH) ? row : 0x7fff) | (col << 16);
  }
  const int tiles = p.tiles_x * p.tiles_y;
  // LDS-DMA of patch `id` into buffer `bufi`: nx_* are set once per patch (nx_set), piece t is issued by issue_one(t) — six
  // loads per wave; patches past the end / pixels outside the image / the padding column read out of range (zeros)
  int nx_iy0 = 0, nx_ix0 = 0, nx_origin = 0, nx_buf = 0;
  bool nx_live = false;
  auto nx_set = [&](int id, int bufi) {
    nx_live = id < p1 && !(p.dbg & 2);
    const int img = id / tiles, rem = id - img * tiles;
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    nx_iy0 = ty * 16 - 2, nx_ix0 = tx * 16 - 2;
    nx_origin = ((img * p.H + nx_iy0) * p.W + nx_ix0) * p.x_cstride * 2;
    nx_buf = bufi * kWsPatchB;
  };
  auto issue_one = [&](auto tc) {
    constexpr int t = decltype(tc)::value;
    const bool ok = nx_live && (unsigned)(nx_iy0 + (l_rc[t] & 0xffff)) < (unsigned)p.H && (unsigned)(nx_ix0 + (l_rc[t] >> 16)) < (unsigned)p.W;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr)(smem + nx_buf + (t * 8 + wave) * 1024), 16,
                                             ok ? (unsigned)(nx_origin + l_rel[t]) : kOOB, 0, 0, 0);
  };
  // B fragment of pixel tile 0: lane (n, lhi) = output (oy, ox) = (n >> 3, n & 7) reads input (2 oy + ty, 2 ox + tx): pair
  // (2 oy + ty) * 10 + ox + tx / 2, slot ((tx & 1) * 8 + c16 * 2 + lhi) ^ key; hk[tx / 2][ty / 2] = (lhi ^ key) * 16
  const int oy_l = l31 >> 3, ox_l = l31 & 7;
  const int b_base = (2 * oy_l * kWsPairs + ox_l) * 256;
  int hk[3][3];
#pragma unroll
  for (int a2 = 0; a2 < 3; ++a2)
#pragma unroll
    for (int b2 = 0; b2 < 3; ++b2) hk[a2][b2] = (lhi ^ (((ox_l + a2) & 3) | (((oy_l + b2) & 3) << 2))) << 4;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;
  const unsigned ex0 = lds0 + 2u * kWsPatchB;

  // Finish of a patch, in pieces hung between the MFMAs (see ws_mfma).  Every LDS access inside the patch loop is inline asm:
  // hipcc orders a compiler-visible LDS access behind ALL earlier LDS-DMA loads (it put an `s_waitcnt vmcnt(0)` — the whole
  // next patch — in front of the finish, in the middle of the MFMA walk).  ex_rd issues the four reads of two K quarters,
  // ex_sum (two half-steps = two B reads later: lgkmcnt(2)) adds them up.
  float v[8];
  float4_t fr[4];
  unsigned vo_prev = kOOB;            // where the patch whose sums sit in the exchange buffer goes (none yet)
  const unsigned ex_addr = ex0 + (unsigned)((2 * e_oct * 64 + (e_pix ^ e_oct)) << 4);
  auto fin_c = [&]() {
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float u = v[e] * sc[e] + sh[e];
      o[e] = (half_t)act_mul(u, act_k);
    }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, o), rsrc_y, vo_prev, 0, FT_YSTORE_BUF_AUX);
  };

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // weights and tables are in registers before the patch loads are counted
  // ... and the compiler has to KNOW it: it cannot see the wait above, and would guard the first use of each of these registers
  // inside the loop with its own vmcnt — a vmcnt(0), i.e. the whole next patch, in front of the finish's scale / shift
#pragma unroll
  for (int j = 0; j < kWsNJ; ++j) asm volatile("" : "+v"(wt[j]));
#pragma unroll
  for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(sc[e]), "+v"(sh[e]));
  nx_set(p0, 0);
  static_for<NLD>([&](auto tc) { issue_one(tc); });
  int k = 0;
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;       // dev (FT_CD_DBG & 32): per-phase s_memtime sums of this wave
#define WS_TS(i) do { if (p.dbg & 32) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tph[i] += t_ - tprev; tprev = t_; } } while (0)
  if (p.dbg & 32) tprev = __builtin_amdgcn_s_memtime();
  const unsigned long long tstart = tprev;
  for (int id = p0; id < p1; ++id, ++k) {
    const int bufi = k & 1;
    // this patch has landed (this wave's share): behind it only the one store of the patch before the previous one may fly
    if (k == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    WS_TS(0);
    // everyone's share has, everyone's sums of the previous patch are in the exchange buffer, the other patch buffer is free
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    WS_TS(1);
    nx_set(id + 1, bufi ^ 1);
    float16_t acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const unsigned abase = lds0 + (unsigned)(bufi * kWsPatchB + b_base);
    unsigned vo_next = kOOB;
    // the K quarters meet in LDS: lane (n, lhi) holds channels c * 32 + 16 * lhi + r of pixel t * 32 + n
    auto ex_write = [&](auto tc, auto rc) {
      constexpr int t = decltype(tc)::value, r4 = decltype(rc)::value;
      const int G = c * 8 + lhi * 4 + r4, px = t * 32 + l31;
      const float4_t u = {acc[t][r4 * 4], acc[t][r4 * 4 + 1], acc[t][r4 * 4 + 2], acc[t][r4 * 4 + 3]};
      ws_ex_wr(ex0 + (unsigned)((((q * 16 + G) * 64 + (px ^ (G >> 1))) << 4)), u);
    };
    auto hook = [&](auto hc) {
      constexpr int hs = decltype(hc)::value;
      if constexpr (hs < 2 * NLD && hs % 2 == 0) issue_one(std::integral_constant<int, hs / 2>{});     // six loads, then the finish's store
      if constexpr (hs == 1) ws_ex_rd<0>(ex_addr, fr);
      if constexpr (hs == 3) { ws_ex_sum<0, 2>(fr, v); ws_ex_rd<1>(ex_addr, fr); }
      if constexpr (hs == 5) ws_ex_sum<1, 2>(fr, v);
      if constexpr (hs == 13) fin_c();
      if constexpr (hs >= 28 && hs < 44 && hs % 4 == 0) ex_write(std::integral_constant<int, 0>{}, std::integral_constant<int, (hs - 28) / 4>{});
      if constexpr (hs == 45) {
        const int img = id / tiles, rem = id - img * tiles;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int oy = ty * 8 + (e_pix >> 3), ox = tx * 8 + (e_pix & 7);
        vo_next = (oy < p.Ho && ox < p.Wo && !(p.dbg & 4))
                      ? (unsigned)((((img * p.Ho + oy) * p.Wo + ox) * p.y_cstride + p.y_coff + cg * 64 + e_oct * 8) * 2) : kOOB;
      }
    };
    // between the tiles: everyone has taken the previous patch's sums out of the exchange buffer
    auto mid = [&]() { asm volatile("s_barrier" ::: "memory"); };
    switch (q) {
      case 0: ws_mfma<0>(wt, abase, hk, acc, hook, mid); break;
      case 1: ws_mfma<1>(wt, abase, hk, acc, hook, mid); break;
      case 2: ws_mfma<2>(wt, abase, hk, acc, hook, mid); break;
      default: ws_mfma<3>(wt, abase, hk, acc, hook, mid); break;
    }
    WS_TS(3);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (the last MFMAs' results, read by inline asm: beyond any hazard window)
    static_for<4>([&](auto rc) { ex_write(std::integral_constant<int, 1>{}, rc); });
    vo_prev = vo_next;
    WS_TS(5);
  }
  if (p.dbg & 32) {                                      // (the output is garbage then)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if ((tid & 63) == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.y) + (blockIdx.x * 8 + wave) * 8;
      for (int i = 0; i < 6; ++i) o[i] = tph[i];
      o[6] = tprev - tstart;
      o[7] = (unsigned long long)k;
    }
    return;
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  ws_ex_rd<0>(ex_addr, fr); ws_ex_sum<0, 0>(fr, v);
  ws_ex_rd<1>(ex_addr, fr); ws_ex_sum<1, 0>(fr, v);
  fin_c();                                               // the last patch
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the look-ahead loads of the last trip must not outlive the workgroup's LDS
#endif
}

// fragment-ordered weights: [group][channel tile 2][K quarter 4][k-step 25][lane 64] x 16 bytes; k-step s = quarter * 25 + j
// holds channels (s & 3) * 16 + 8 * lhi .. of tap s >> 2, i.e. k = s * 16 + 8 * lhi of the packed row (cin_pad = 64)
__global__ __launch_bounds__(256) void ws_pack_kernel(const half_t* __restrict__ w, uint4_t* __restrict__ out, int ncg, int kpad, int cout_pad) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= ncg * 2 * 4 * kWsNJ * 64) return;
  const int lane = idx & 63;
  int f = idx >> 6;
  const int j = f % kWsNJ; f /= kWsNJ;
  const int q = f & 3; f >>= 2;
  const int c = f & 1, cg = f >> 1;
  const int co = cg * 64 + c * 32 + ws_sigma(lane & 31);
  const int k = (q * kWsNJ + j) * 16 + 8 * (lane >> 5);
  uint4_t v = {0u, 0u, 0u, 0u};
  if (co < cout_pad && k + 8 <= kpad) v = *reinterpret_cast<const uint4_t*>(w + (size_t)co * kpad + k);
  out[idx] = v;
}

int ws_plan(const ft_conv_desc* d, WsPlan* out) {
  if (!d || !out) return FT_ERR_INVALID_ARG;
  static const int off = getenv("FT_CD_NO_WSTAT") ? atoi(getenv("FT_CD_NO_WSTAT")) : 0;
  if (off) return FT_ERR_UNSUPPORTED;
  if (d->act == FT_ACT_LEAKY && !(d->slope >= 0.f && d->slope <= 1.f)) return FT_ERR_UNSUPPORTED;   // epilogues use max(v, k * v): valid for 0 <= k <= 1 only
  if (d->dtype != FT_F16 || d->transposed || d->kh != 5 || d->kw != 5 || d->stride != 2 || d->pad != 2 || d->Cin != 64) return FT_ERR_UNSUPPORTED;
  if (d->x2_cin || d->has_residual || d->tail_cout || d->pool || d->x_wpitch || d->out_layout != FT_LAYOUT_NHWC) return FT_ERR_UNSUPPORTED;
  if (d->N <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Cout <= 0 || d->Cout % 64 || d->Cout > 2048) return FT_ERR_UNSUPPORTED;
  if (d->Ho != (d->Hi + 4 - 5) / 2 + 1 || d->Wo != (d->Wi + 4 - 5) / 2 + 1 || d->Ho <= 0 || d->Wo <= 0) return FT_ERR_INVALID_ARG;
  if (d->x_coff % 8 || d->x_cstride % 8 || d->y_coff % 8 || d->y_cstride % 8) return FT_ERR_UNSUPPORTED;
  if (d->x_cstride < d->x_coff + d->Cin || d->y_cstride < d->y_coff + d->Cout) return FT_ERR_INVALID_ARG;
  const long long lim = 1LL << 31;
  // (the patch origin may sit two rows / columns before the image: its negative offset must not wrap either)
  if (((long long)d->N * d->Hi + 4) * (d->Wi + 4) * d->x_cstride * 2 >= lim || (long long)d->N * d->Ho * d->Wo * d->y_cstride * 2 >= lim) return FT_ERR_UNSUPPORTED;
  const int ncg = d->Cout / 64;
  if (ncg > 32) return FT_ERR_UNSUPPORTED;
  out->ncg = ncg;
  out->tiles_x = ceil_div(d->Wo, 8);
  out->tiles_y = ceil_div(d->Ho, 8);
  const long long np = (long long)d->N * out->tiles_x * out->tiles_y;
  if (np >= (1 << 24)) return FT_ERR_UNSUPPORTED;
  out->npatches = (int)np;
  return FT_OK;
}

long long ws_weight_bytes(const WsPlan& pl) { return (long long)pl.ncg * 2 * 4 * kWsNJ * 1024; }

int ws_pack(const ft_conv_desc* d, const WsPlan& pl, const void* w_packed, int kpad, int cout_pad, void* wstream, hipStream_t stream) {
  if (!w_packed || !wstream || kpad < 25 * 64 || cout_pad < d->Cout) return FT_ERR_INVALID_ARG;
  const int total = pl.ncg * 2 * 4 * kWsNJ * 64;
  hipLaunchKernelGGL(ws_pack_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, static_cast<const half_t*>(w_packed),
                     static_cast<uint4_t*>(wstream), pl.ncg, kpad, cout_pad);
  FT_LAUNCH_CHECK("ws_pack_kernel");
  return FT_OK;
}

int ws_launch(const ft_conv_desc* d, const WsPlan& pl, const void* x, const void* wstream, const float* scale, const float* shift,
              void* y, hipStream_t stream) {
  if (!x || !wstream || !y) return FT_ERR_INVALID_ARG;
  static const int dbg = getenv("FT_CD_DBG") ? atoi(getenv("FT_CD_DBG")) : 0;
  WsParams q{};
  q.x = static_cast<const char*>(x);
  q.y = static_cast<char*>(y);
  q.ws = static_cast<const char*>(wstream);
  q.scale = scale;
  q.shift = shift;
  q.H = d->Hi; q.W = d->Wi; q.Ho = d->Ho; q.Wo = d->Wo;
  q.x_cstride = d->x_cstride; q.x_coff = d->x_coff; q.y_cstride = d->y_cstride; q.y_coff = d->y_coff;
  q.act = d->act; q.slope = d->slope;
  q.ncg = pl.ncg; q.tiles_x = pl.tiles_x; q.tiles_y = pl.tiles_y; q.npatches = pl.npatches;
  const int npairs = 8 * (32 / pl.ncg);
  q.ppp = ceil_div(pl.npatches, npairs);
  q.x_bytes = (unsigned)((size_t)d->N * d->Hi * d->Wi * d->x_cstride * 2);
  q.y_bytes = (unsigned)((size_t)d->N * d->Ho * d->Wo * d->y_cstride * 2);
  q.ws_bytes = (unsigned)ws_weight_bytes(pl);
  q.dbg = dbg;
  FT_RAISE_LDS(conv5x5s2_wstat_kernel, kWsLds);
  hipLaunchKernelGGL(conv5x5s2_wstat_kernel, dim3(256), dim3(512), kWsLds, stream, q);
  FT_LAUNCH_CHECK("conv5x5s2_wstat_kernel");
  return FT_OK;
}

}  // namespace ft
