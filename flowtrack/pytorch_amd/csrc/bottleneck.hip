// Whole-bottleneck fusion for the HBM-bound first ResNet stage (fp16): conv1 1x1 + bn1 + relu -> conv2 3x3 + bn2 + relu
// -> conv3 1x1 + bn3 + identity residual + relu in ONE launch (reference: Bottleneck.forward, lib/pose/models/blocks.py:
// 105-120, the blocks without a projection shortcut, layer1.1 / layer1.2 of resnet.py:29-36).
//
// Why: at 64x48 x 256 channels the three launches move 6.1 MB per crop through HBM (read x, write/read t1, write/read
// t2, read the residual, write y) for 27 MFLOP/KB — they run at the HBM roof (134 us per block at batch 64, of which
// the matrix pipe needs ~45).  Fused, a block reads x once and writes y once (3.0 MB per crop): t1 and t2 only ever
// exist in LDS.
//
// One 256-thread workgroup owns an 8x16 (or 16x8) patch of output pixels of one image:
//   phase 1  t1 = relu(bn1(W1 . x)) on the 10x18 HALO patch (180 pixels padded to 192; out-of-image pixels forced to 0 =
//            conv2's zero padding): GEMM [64 co] x [192 px] x K=256.  x streams HBM -> LDS in four 64-channel chunks
//            (whole 128-byte lines per pixel) through two DMA stages together with the matching K-slice of W1; the
//            residual (the patch's own 128 pixels) is picked out of the same stages in phase 3's register layout;
//            result -> fp16 T1 in LDS (24 KiB).
//   phase 2  t2 = relu(bn2(W2 * t1)): GEMM [64 co] x [128 px] x K=9*64; the pixel operand of every tap is read from T1
//            at the tap's offset (the conv_halo_kernel idea), only W2 streams (one tap = 8 KiB per K-step, 4-slot ring,
//            W3's quarters follow the taps through the same ring) -> fp16 T2 (16 KiB).
//   phase 3  y = relu(bn3(W3 . t2) + x): four 64-channel quarters, GEMM [64 co] x [128 px] x K=64 each, nothing is loaded
//            from memory any more; the fp16 tile is transposed through two alternating LDS buffers for 16-byte
//            coalesced stores (one barrier per quarter, no wait on the stores).
// 75 KiB of LDS -> two workgroups per CU.  The halo makes phase 1 do 1.4x the 1x1's MACs (0.9 GMAC of 13.7 per block at
// batch 64): cheap next to 3 MB of HBM.
// Measured (batch 64, 64x48): 134 us for the three launches -> 64 us.  Per patch 232 KB go L2 -> LDS (96 x, 136 weights)
// and that delivery (~16 B/clk/CU, the same ceiling the conv kernels see) is what bounds it now, not HBM: tools/dev/
// bnk_phases.py (phase timestamps) and the FT_BNK_DBG ablations in bnk_bench.py are how that was established; deeper
// rings, 64- vs 128-byte rows, an L2 warm-up of the later chunks and staggering the co-resident workgroups changed nothing.
#include <stdlib.h>

#include <type_traits>

#include "ft_common.h"

// cache-policy bits of the block input's LDS-DMA loads (read once per workgroup + its halo neighbours): 0 = default, 2 = nt (A/B
// builds: tools/dev/build_variant.sh bnknt -DFT_BNK_XLOAD_AUX=2)
#ifndef FT_BNK_XLOAD_AUX
#define FT_BNK_XLOAD_AUX 0
#endif
namespace ft {
namespace {

struct BnkParams {
  const char* x;
  char* y;
  const char *w1, *w2, *w3;
  const char* tab;   // float [s1 64 | b1 64 | s2 64 | b2 64 | s3 256 | b3 256]
  int N, H, W;
  int x_cstride, x_coff, y_cstride, y_coff;
  unsigned x_bytes;
  int tx, ty;      // patches per image along x / y
  int total;       // workgroups
  int w3_pitch;    // bytes per output-channel row of w3: 128 (W3 [256][64]); 256 at a projection block ([256][W3' 64 | Wd' 64])
  int dbg;         // FT_BNK_DBG (dev): 1 no x loads, 4 no stores, 16 x loads from images 0..7 only (L2-resident), 32 phase timestamps
};

template <int N, int I = 0, typename F>
__device__ __forceinline__ void unroll_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    unroll_for<N, I + 1>(f);
  }
}

// s_barrier with a memory clobber: the builtin is IntrNoMem, LDS reads may be hoisted above it (see conv_igemm.hip)
#define BNK_BARRIER() asm volatile("s_barrier" ::: "memory")
// XOR key of the 16-byte chunk position inside a 128-byte LDS row.  A 256-byte bank row holds TWO such rows, so the sixteen
// consecutive rows a ds_read_b128 services together hit sixteen distinct 16-byte slots only if the key changes every second
// row: with `row & 7` rows r and r + 8 collided.  PMC (profiles/r02_lds_wait_counters.txt): bank-conflict cycles were 43-47 % of
// this kernel's LDS-active cycles before and are 30-33 % after — the fragment reads this key governs are fixed (the same change
// takes conv_direct's ring reads from 27-44 % to 0), but a third of the LDS time here is STILL conflicts.  Not tracked down;
// candidates are the 8-byte accesses (T1 / T2 / staging epilogue writes, the residual pick-up), where the two lanes of a pixel
// write the halves of one 16-byte chunk.  60.4 -> 57.3 us per block came from the part that is fixed.
#ifndef FT_BNK_KEY_SHIFT
#define FT_BNK_KEY_SHIFT 1
#endif
#define BNK_KEY(r) (((r) >> FT_BNK_KEY_SHIFT) & 7)
// T1 (the 18- or 10-pixel-wide halo patch conv2's taps read at a row / column shift): ds_read_b128 serves lanes {0-3, 12-15,
// 20-27} together = with 16-pixel tile rows: columns 0-3 and 12-15 of one patch row and 4-11 of the next.  Keyed by the linear
// index (r >> 1 = 9 * row + column / 2) columns 12-13 of a row and 10-11 of the next share key AND 128-byte half: a 2-way
// conflict in every such read (+1 LDS cycle on 4: the 30-33 % conflict cycles the PMC kept showing).  Keyed by the COLUMN pair
// alone the sixteen lanes cover the sixteen slots.  (8-pixel tile rows: four rows x four columns per group; the old key stays.)
#ifndef FT_BNK_T1_COLKEY
#define FT_BNK_T1_COLKEY 1
#endif
template <int TW> __device__ __forceinline__ int bnk_t1_key(int r, int pc) { return (FT_BNK_T1_COLKEY && TW == 16) ? ((pc >> 1) & 7) : BNK_KEY(r); }

constexpr int kC = 256, kP = 64;                 // block width / planes this kernel is written for
// LDS map (bytes).  Phase 1: two 32-KiB stages (64-channel x chunk 24 KiB + W1 K-slice 8 KiB) at 0 .. 64 Ki.
// Phase 2: T1 at 0 (24 KiB), W2 ring slots 1..3 at 24 / 32 / 40 Ki and slot 0 at 64 Ki, T2 at 48 Ki (16 KiB).
// Phase 3: W3 quarters in the ring slots, output staging A at 0 and B at 48 Ki (16 KiB each).  Folded-BN table at 72 Ki.
[[maybe_unused]] constexpr int kStage1 = 32768, kXChunk = 24576;
[[maybe_unused]] constexpr int kOffT1 = 0, kOffT2 = 49152, kOffOutA = 0, kOffOutB = 49152, kOffTab = 73728;
constexpr int kLdsBytes = 76800;
__device__ __forceinline__ constexpr int ring_slot(int i) { return i == 0 ? 65536 : 24576 + (i - 1) * 8192; }

// NCH = 64-channel chunks of the block input (4: the 256-wide identity blocks, 1: the stage's entry block, 64 in).
// FULL = false stops after phase 2 and writes t2 (the entry block's conv3 is K-concatenated with its projection shortcut
// in ft_conv2d_fwd's x2_* path; here only its conv1 + conv2 pair is fused).
// PROJ = the entry block WHOLE (64 -> 64 -> 64 -> 256 with its projection shortcut, blocks.py:104-119): phase 3 is the GEMM
// over K = [t2 | x] of FusedShortcutConv (both BatchNorms folded into the weights, the shifts added): W3' quarters come through
// the ring as before, the x fragments of the patch's own pixels are picked out of the phase-1 stage in the B-operand layout
// and the shortcut weights Wd' (32 KiB per workgroup, L2-resident) go straight to registers at kernel start.
template <int TW, int NCH, bool FULL, bool PROJ = false>
__global__ __launch_bounds__(256, 2) void bottleneck_fused_kernel(const BnkParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TH = 128 / TW, PW = TW + 2, PH = TH + 2, NPIX = PW * PH;
  static_assert(NPIX <= 192, "halo patch");
  static_assert(!PROJ || (FULL && NCH == 1), "projection form: 64-channel input, whole block");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;

  // XCD-aware order: block b runs on XCD b % 8; give each XCD one contiguous range of patches (neighbouring patches
  // share their halo rows through that XCD's L2)
  int logical;
  {
    const int b = blockIdx.x;
    const int q = p.total >> 3, r = p.total & 7, xcd = b & 7, loc = b >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tiles_per_img = p.tx * p.ty;
  const int n = logical / tiles_per_img;
  const int trem = logical - n * tiles_per_img;
  const int tyi = trem / p.tx, txi = trem - tyi * p.tx;
  const int qy0 = tyi * TH, qx0 = txi * TW;

  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  // (dbg & 64, dev: every weight load out of range = no traffic, zeros: the kernel's time without its 136 KB of weights per patch)
  const bool now = p.dbg & 64;
  const __amdgpu_buffer_rsrc_t rsrc_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.w1), 0, now ? 0 : kP * NCH * 128, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.w2), 0, now ? 0 : kP * 9 * kP * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.w3), 0, (FULL && !now) ? kC * kP * (PROJ ? 4 : 2) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_tab = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.tab), 0, FULL ? 3072 : 1024, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;

  // ---- loader lanes --------------------------------------------------------------------------------------------
  // every LDS tile here has 128-byte rows (64 fp16): a 1-KiB wave load covers 8 rows, lane -> (row = lane / 8, 16-byte
  // position = lane % 8) — whole 128-byte lines per row for the texture path (64-byte rows cost twice the requests per
  // byte: the phase-1 stream was request-bound with them).  The LDS image is lane-linear, so the XOR swizzle
  // (position ^= BNK_KEY(row)) is applied to the SOURCE position.
  const int lrow = lane >> 3, lpos = lane & 7;
  unsigned x_voff[6];
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    const int pp = (t * 4 + wave) * 8 + lrow;           // halo pixel
    const int pr = pp / PW, pc = pp - pr * PW;
    const int iy = qy0 - 1 + pr, ix = qx0 - 1 + pc;
    unsigned v = kOOB;
    if (pp < NPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && !(p.dbg & 1))
      v = (unsigned)(((((p.dbg & 16 ? (n & 7) : n) * p.H + iy) * p.W + ix) * p.x_cstride + p.x_coff) * 2 + ((lpos ^ BNK_KEY(pp)) << 4));
    x_voff[t] = v;
  }
  unsigned w1_voff[2], w2_voff[2], w3_voff[2];           // rows (t * 4 + wave) * 8 + lrow of a 64-row weight block
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int wr = (t * 4 + wave) * 8 + lrow;
    const unsigned lc = (unsigned)((lpos ^ BNK_KEY(wr)) << 4);
    w1_voff[t] = (unsigned)(wr * NCH * 128) + lc;        // W1 [64][64 * NCH]
    w2_voff[t] = (unsigned)(wr * 9 * kP * 2) + lc;       // W2 [64][576]
    w3_voff[t] = (unsigned)(wr * p.w3_pitch) + lc;       // W3 [256][64] (or the first half of [256][128]), + quarter * 64 rows
  }

  auto load_stage1 = [&](int slot, int c) {              // chunk c: channels 64c .. 64c+63 of the halo patch + W1's K-slice
    char* st = smem + slot * kStage1;
#pragma unroll
    for (int t = 0; t < 6; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr)(st + (t * 4 + wave) * 1024), 16, x_voff[t],
                                               x_voff[t] == kOOB ? 0 : c * 128, 0, FT_BNK_XLOAD_AUX);
#pragma unroll
    for (int t = 0; t < 2; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w1, (lds_ptr)(st + kXChunk + (t * 4 + wave) * 1024), 16, w1_voff[t], c * 128, 0, 0);
  };
  // ring items 0..8 = the nine taps of W2 ([64 co][64 ci] = 128-byte rows), items 9..12 = the quarters of W3
  auto load_item = [&](int item) {
    char* st = smem + ring_slot(item & 3);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (item < 9)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w2, (lds_ptr)(st + (t * 4 + wave) * 1024), 16, w2_voff[t], item * 128, 0, 0);
      else    // !FULL: rsrc_w3 is empty, every lane is out of range -> zero fill, no traffic, same load count for the waits
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w3, (lds_ptr)(st + (t * 4 + wave) * 1024), 16, w3_voff[t], (item - 9) * 64 * p.w3_pitch, 0, 0);
    }
  };

  // oldest loads of every wave: the folded-BN table (3 KiB, waves 0..2) and W2's first tap (its ring slot lies outside the
  // phase-1 stages) — both land long before they are needed and sit in front of every counted wait below
  // projection form: the shortcut weights of this wave's channel tile, all four quarters x four 16-channel K slices, straight
  // to registers (the oldest loads of the wave: every counted wait below covers them)
  uint4_t fad[PROJ ? 4 : 1][4];
  if constexpr (PROJ) {
    const int wc2_ = wave >> 1;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4)
        fad[q][k4] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w3, (unsigned)((q * 64 + wc2_ * 32 + l31) * 256 + 128 + (k4 * 2 + lhi) * 16), 0, 0);
  }
  if (wave < (FULL ? 3 : 1))
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_tab, (lds_ptr)(smem + kOffTab + wave * 1024), 16, (unsigned)(lane * 16), wave * 1024, 0, 0);
  load_item(0);
  const float* tab = reinterpret_cast<const float*>(smem + kOffTab);
  unsigned long long ts[8];
#define BNK_TS(i) do { if (p.dbg & 32) ts[i] = __builtin_amdgcn_s_memtime(); } while (0)
  BNK_TS(0);

  // ================= phase 1: t1 = relu(bn1(W1 . x)) on the halo patch ==========================================
  // wave -> output-channel tile (wave & 1) x three 32-pixel tiles (wave >> 1)
  const int wc1 = wave & 1, wp1 = wave >> 1;
  const int a1_row = wc1 * 32 + l31;
  const int a1_off = a1_row * 128 + ((lhi ^ BNK_KEY(a1_row)) << 4);       // + (k16 * 2) << 4 by XOR: 16-channel slice k16
  int b1_off[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int r = (wp1 * 3 + j) * 32 + l31;
    b1_off[j] = r * 128 + ((lhi ^ BNK_KEY(r)) << 4);
  }
  float16_t acc1[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;

  // residual = the block input at the patch's own 128 pixels, picked up in the ACCUMULATOR layout of phase 3 (lane ->
  // pixel wp2*64 + j*32 + l31, channels q*64 + wc2*32 + g*8 + lhi*4 .. +3) from the phase-1 stages as they pass through
  // LDS: chunk c = quarter c.  No second trip to L2 for them.
  const int wc2 = wave >> 1, wp2 = wave & 1;
  int rc_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = wp2 * 64 + j * 32 + l31;
    const int rc = (m / TW + 1) * PW + (m % TW + 1);
    rc_off[j] = rc * 128 + lhi * 8 + (((wc2 * 4) ^ BNK_KEY(rc)) << 4);    // 16-byte chunk wc2*4 + g sits at (.. ^ g) << 4
  }
  half4_t res[FULL && !PROJ ? 4 : 1][2][4];
  uint4_t fbx[PROJ ? 2 : 1][2][2];      // projection form: the block input at the patch's own pixels as phase 3's pixel operand

  // two 32-KiB stages: chunk c+1 streams while chunk c is multiplied; a stage is refilled (chunk c+2) once every wave is
  // past its reads — a second barrier per chunk, four chunks
  load_stage1(0, 0);
  if constexpr (NCH > 1) load_stage1(1, 1);
  unroll_for<NCH>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if constexpr (c < NCH - 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");   // chunk c landed; c+1 (8 loads) may fly
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    BNK_BARRIER();
    const char* st = smem + (c & 1) * kStage1;
    uint4_t fa[4], fb[4][3];
#pragma unroll
    for (int k16 = 0; k16 < 4; ++k16) {
      fa[k16] = *reinterpret_cast<const uint4_t*>(st + kXChunk + (a1_off ^ (k16 << 5)));
#pragma unroll
      for (int j = 0; j < 3; ++j) fb[k16][j] = *reinterpret_cast<const uint4_t*>(st + (b1_off[j] ^ (k16 << 5)));
    }
    if constexpr (FULL && !PROJ) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) res[c][j][g] = *reinterpret_cast<const half4_t*>(st + (rc_off[j] ^ (g << 4)));
    }
    if constexpr (PROJ) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m = wp2 * 64 + j * 32 + l31;
        const int rc = (m / TW + 1) * PW + (m % TW + 1);
        const int lsw = lhi ^ BNK_KEY(rc);
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
            fbx[sl][kk][j] = *reinterpret_cast<const uint4_t*>(st + rc * 128 + ((lsw ^ (sl * 4 + kk * 2)) << 4));
      }
    }
#pragma unroll
    for (int k16 = 0; k16 < 4; ++k16)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, fa[k16]),
                                                         __builtin_bit_cast(half8_t, fb[k16][j]), acc1[j], 0, 0, 0);
    if constexpr (c + 2 < NCH) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      BNK_BARRIER();
      load_stage1(c & 1, c + 2);
    }
  });
  // the last chunk had vmcnt(0): nothing of this wave is in flight.  Every wave is past its last stage read once it
  // reaches this barrier: the stage region becomes T1 + the W2 ring.
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  BNK_BARRIER();
  BNK_TS(1);
  load_item(1);
  load_item(2);
  {
    char* t1 = smem + kOffT1;
    float4_t sc[4], sh[4];               // read once: the compiler cannot hoist LDS reads over the T1 writes itself
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      sc[g] = *reinterpret_cast<const float4_t*>(tab + wc1 * 32 + g * 8 + lhi * 4);
      sh[g] = *reinterpret_cast<const float4_t*>(tab + 64 + wc1 * 32 + g * 8 + lhi * 4);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int r = (wp1 * 3 + j) * 32 + l31;
      const int pr = r / PW, pc = r - pr * PW;
      const int iy = qy0 - 1 + pr, ix = qx0 - 1 + pc;
      const bool inside = r < NPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      char* rowp = t1 + r * 128 + lhi * 8;
      const int rsw = bnk_t1_key<TW>(r, pc) << 4;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half4_t h;
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const half2_t o = bn_relu_pk(acc1[j][g * 4 + e], acc1[j][g * 4 + e + 1], sc[g][e], sc[g][e + 1], sh[g][e], sh[g][e + 1]);
          h[e] = o[0];
          h[e + 1] = o[1];
        }
        uint2 hb = __builtin_bit_cast(uint2, h);
        hb.x = inside ? hb.x : 0u;       // out-of-image halo pixels are conv2's zero padding, not relu(bn1(0))
        hb.y = inside ? hb.y : 0u;
        *reinterpret_cast<uint2*>(rowp + ((((wc1 * 4 + g) << 4)) ^ rsw)) = hb;
      }
    }
  }
  BNK_TS(2);
  // ================= phase 2: t2 = relu(bn2(W2 * t1)), pixel operand from T1 ======================================
  // wave -> output-channel tile wc2 = wave >> 1 x two 32-pixel tiles (wp2 = wave & 1)
  const int a2_row = wc2 * 32 + l31;
  const int a2_off = a2_row * 128 + ((lhi ^ BNK_KEY(a2_row)) << 4);
  int r0[2], c0[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = wp2 * 64 + j * 32 + l31;
    r0[j] = (m / TW) * PW + (m % TW);
    c0[j] = m % TW;
  }
  float16_t acc2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;

  unroll_for<9>([&](auto tc) {
    constexpr int tap = decltype(tc)::value;
    constexpr int ky = tap / 3, kx = tap % 3;
    // item `tap` has landed; the two younger items (2 loads per wave each) stay in flight.  (tap 0: also T1 written)
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    BNK_BARRIER();
    load_item(tap + 3);                                  // taps 3..8, then quarters 0..2 of W3
    const char* st = smem + ring_slot(tap & 3);
    const char* t1 = smem + kOffT1;
    uint4_t fa[2][2], fb[2][2][2];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fa[sl][kk] = *reinterpret_cast<const uint4_t*>(st + (a2_off ^ ((sl * 2 + kk) << 5)));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = r0[j] + ky * PW + kx;
      const int lsw = lhi ^ bnk_t1_key<TW>(r, c0[j] + kx);
      const char* rowp = t1 + r * 128;
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          fb[sl][kk][j] = *reinterpret_cast<const uint4_t*>(rowp + ((lsw ^ (sl * 4 + kk * 2)) << 4));
    }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, fa[sl][kk]),
                                                           __builtin_bit_cast(half8_t, fb[sl][kk][j]), acc2[j], 0, 0, 0);
  });

  BNK_TS(3);
  {
    char* t2 = smem + kOffT2;            // the T2 region held only a phase-1 stage, dead since the barrier after phase 1
    float4_t sc[4], sh[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      sc[g] = *reinterpret_cast<const float4_t*>(tab + 128 + wc2 * 32 + g * 8 + lhi * 4);
      sh[g] = *reinterpret_cast<const float4_t*>(tab + 192 + wc2 * 32 + g * 8 + lhi * 4);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = wp2 * 64 + j * 32 + l31;
      char* rowp = t2 + m * 128 + lhi * 8;
      const int msw = BNK_KEY(m) << 4;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half4_t h;
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const half2_t o = bn_relu_pk(acc2[j][g * 4 + e], acc2[j][g * 4 + e + 1], sc[g][e], sc[g][e + 1], sh[g][e], sh[g][e + 1]);
          h[e] = o[0];
          h[e + 1] = o[1];
        }
        *reinterpret_cast<half4_t*>(rowp + (((wc2 * 4 + g) << 4) ^ msw)) = h;
      }
    }
  }
  // T2 complete, T1 and the last tap's slot dead once every wave is here: the last quarter of W3 goes into that slot
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  BNK_BARRIER();
  BNK_TS(4);
  if constexpr (!FULL) {     // entry block: t2 is the result — 128 pixels x 128 bytes, 16 bytes per lane
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's trailing (empty) loads must not outlive the workgroup's LDS
    const char* t2 = smem + kOffT2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i, m = idx >> 3, ch = idx & 7;
      const int oy = qy0 + m / TW, ox = qx0 + m % TW;
      const uint4_t v = *reinterpret_cast<const uint4_t*>(t2 + m * 128 + ((ch ^ BNK_KEY(m)) << 4));
      if (oy < p.H && ox < p.W && !(p.dbg & 4))
        store_out16(p.y + ((((long long)n * p.H + oy) * p.W + ox) * p.y_cstride + p.y_coff + ch * 8) * 2, v);
    }
    return;
  }
  load_item(12);

  // ================= phase 3: y = relu(bn3(W3 . t2) + x), four quarters of 64 output channels ====================
  uint4_t fb3[2][2][2];
  {
    const char* t2 = smem + kOffT2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = wp2 * 64 + j * 32 + l31;
      const int lsw = lhi ^ BNK_KEY(m);
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          fb3[sl][kk][j] = *reinterpret_cast<const uint4_t*>(t2 + m * 128 + ((lsw ^ (sl * 4 + kk * 2)) << 4));
    }
  }
  long long spix[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = (tid + 256 * i) >> 3;
    const int oy = qy0 + m / TW, ox = qx0 + m % TW;
    spix[i] = (oy < p.H && ox < p.W) ? ((long long)n * p.H + oy) * p.W + ox : -1;
  }
  // every W3 quarter and the residuals have landed; every wave holds its T2 fragments (staging B reuses the T2 region)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  BNK_BARRIER();
  BNK_TS(5);

  unroll_for<4>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    const char* wq = smem + ring_slot((q + 1) & 3);
    float16_t acc3[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc3[j][r] = 0.f;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const uint4_t fa = *reinterpret_cast<const uint4_t*>(wq + (a2_off ^ ((sl * 2 + kk) << 5)));
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc3[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, fa),
                                                           __builtin_bit_cast(half8_t, fb3[sl][kk][j]), acc3[j], 0, 0, 0);
      }
    if constexpr (PROJ) {
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc3[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, fad[q][sl * 2 + kk]),
                                                             __builtin_bit_cast(half8_t, fbx[sl][kk][j]), acc3[j], 0, 0, 0);
    }
    char* so = smem + ((q & 1) ? kOffOutB : kOffOutA);
    float4_t sc[4], sh[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      sc[g] = *reinterpret_cast<const float4_t*>(tab + 256 + q * 64 + wc2 * 32 + g * 8 + lhi * 4);
      sh[g] = *reinterpret_cast<const float4_t*>(tab + 512 + q * 64 + wc2 * 32 + g * 8 + lhi * 4);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = wp2 * 64 + j * 32 + l31;
      char* rowp = so + m * 128 + lhi * 8;
      const int msw = BNK_KEY(m) << 4;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half4_t h;
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          half2_t o;
          if constexpr (PROJ) o = bn_relu_pk(acc3[j][g * 4 + e], acc3[j][g * 4 + e + 1], sc[g][e], sc[g][e + 1], sh[g][e], sh[g][e + 1]);
          else o = bn_res_relu_pk(acc3[j][g * 4 + e], acc3[j][g * 4 + e + 1], sc[g][e], sc[g][e + 1], sh[g][e], sh[g][e + 1],
                                  res[PROJ ? 0 : q][j][g][e], res[PROJ ? 0 : q][j][g][e + 1]);
          h[e] = o[0];
          h[e + 1] = o[1];
        }
        *reinterpret_cast<half4_t*>(rowp + (((wc2 * 4 + g) << 4) ^ msw)) = h;
      }
    }
    // one barrier per quarter: the two staging buffers alternate, a buffer's previous readers are two barriers behind
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    BNK_BARRIER();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i, m = idx >> 3, ch = idx & 7;
      const uint4_t v = *reinterpret_cast<const uint4_t*>(so + m * 128 + ((ch ^ BNK_KEY(m)) << 4));
      if (spix[i] >= 0 && !(p.dbg & 4))
        store_out16(p.y + (spix[i] * p.y_cstride + p.y_coff + q * 64 + ch * 8) * 2, v);
    }
  });
  if (p.dbg & 32) {       // dev: phase timestamps of wave 0 over the tile's first output pixel (output is garbage then)
    ts[6] = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts[7] = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.y + ((((long long)n * p.H + qy0) * p.W + qx0) * p.y_cstride + p.y_coff) * 2);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = ts[i];
    }
  }
#endif
}

static int supported(const ft_bottleneck_desc* d) {
  if (!d) return FT_ERR_INVALID_ARG;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0) return FT_ERR_INVALID_ARG;
  if (d->dtype != FT_F16 || d->P != kP || d->stride > 1) return FT_ERR_UNSUPPORTED;
  if (d->head_only && d->projection) return FT_ERR_INVALID_ARG;
  if (d->head_only || d->projection ? d->C != 64 : d->C != kC) return FT_ERR_UNSUPPORTED;   // (256-wide head-only: no caller, not instantiated)
  const int yc = d->head_only ? d->P : kC;
  if (d->x_coff < 0 || d->y_coff < 0 || d->x_coff % 8 || d->y_coff % 8 || d->x_cstride % 8 || d->y_cstride % 8) return FT_ERR_UNSUPPORTED;
  if (d->x_cstride < d->x_coff + d->C || d->y_cstride < d->y_coff + yc) return FT_ERR_INVALID_ARG;
  if ((long long)d->N * d->H * d->W * d->x_cstride * 2 >= (1LL << 31)) return FT_ERR_UNSUPPORTED;
  return FT_OK;
}

template <int TW, int NCH, bool FULL, bool PROJ = false>
static int launch(const BnkParams& p, hipStream_t s) {
  auto k = bottleneck_fused_kernel<TW, NCH, FULL, PROJ>;
  FT_RAISE_LDS(k, kLdsBytes);
  hipLaunchKernelGGL(k, dim3(p.total), dim3(256), kLdsBytes, s, p);
  FT_LAUNCH_CHECK("bottleneck_fused_kernel");
  return FT_OK;
}

}  // namespace
}  // namespace ft

extern "C" int ft_bottleneck_supported(const ft_bottleneck_desc* d) { return ft::supported(d); }

extern "C" double ft_bottleneck_flops(const ft_bottleneck_desc* d) {
  if (!d) return 0.0;
  const double cout = 4.0 * d->P;
  if (d->head_only && d->stride == 2)     // conv1 on the input map, conv2 on the halved one
    return 2.0 * d->N * ((double)d->H * d->W * d->C * d->P + (double)(d->H / 2) * (d->W / 2) * 9.0 * d->P * d->P);
  return 2.0 * d->N * d->H * d->W * ((double)d->C * d->P + 9.0 * d->P * d->P + (d->head_only ? 0.0 : (double)d->P * cout) +
                                     (d->projection ? (double)d->C * cout : 0.0));
}

extern "C" int ft_bottleneck_fwd(const ft_bottleneck_desc* d, const void* x, const void* w1, const void* w2, const void* w3,
                                 const float* scale_shift, void* y, ft_stream_t stream) {
  using namespace ft;
  const int st = supported(d);
  if (st != FT_OK) return st;
  if (!x || !w1 || !w2 || (!w3 && !d->head_only) || !scale_shift || !y) return FT_ERR_INVALID_ARG;
  BnkParams p{};
  p.x = static_cast<const char*>(x);
  p.y = static_cast<char*>(y);
  p.w1 = static_cast<const char*>(w1);
  p.w2 = static_cast<const char*>(w2);
  p.w3 = static_cast<const char*>(w3);
  p.tab = reinterpret_cast<const char*>(scale_shift);
  p.N = d->N; p.H = d->H; p.W = d->W;
  p.x_cstride = d->x_cstride; p.x_coff = d->x_coff; p.y_cstride = d->y_cstride; p.y_coff = d->y_coff;
  p.x_bytes = (unsigned)((size_t)d->N * d->H * d->W * d->x_cstride * 2);
  p.w3_pitch = d->projection ? 256 : 128;
  // 8 rows x 16 columns unless the width only divides by 8 (R101 at 384x288: 96x72 maps -> 16 rows x 8 columns)
  const bool tall = d->W % 16 != 0 && d->W % 8 == 0;
  const int tw = tall ? 8 : 16, th = 128 / tw;
  p.tx = ceil_div(d->W, tw);
  p.ty = ceil_div(d->H, th);
  p.total = d->N * p.tx * p.ty;
  static const int dbg = getenv("FT_BNK_DBG") ? atoi(getenv("FT_BNK_DBG")) : 0;
  p.dbg = dbg;
  hipStream_t s = as_stream(stream);
  if (d->head_only) return tall ? launch<8, 1, false>(p, s) : launch<16, 1, false>(p, s);
  if (d->projection) return tall ? launch<8, 1, true, true>(p, s) : launch<16, 1, true, true>(p, s);
  return tall ? launch<8, 4, true>(p, s) : launch<16, 4, true>(p, s);
}

