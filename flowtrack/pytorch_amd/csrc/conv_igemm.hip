// Fused Conv2d / ConvTranspose2d(4,2,1) forward as an implicit GEMM on the gfx950 matrix cores.
//
// What it replaces (reference, run there through torch.nn -> cuDNN, every op a separate pass):
//   Conv2d + BatchNorm2d(eval) + ReLU (+ residual add)   lib/pose/models/blocks.py:105-120, resnet.py:19-23
//   ConvTranspose2d(4,2,1) + BatchNorm2d + ReLU          lib/pose/models/pose_deconv.py:19-28
//   Conv2d + bias + LeakyReLU(0.1)                       lib/flownet/networks/submodules.py:7-18
//   ConvTranspose2d(4,2,1) + bias + LeakyReLU(0.1)       lib/flownet/networks/submodules.py:34-38
//
// Formulation (MI355X-first, not a cuDNN look-alike):
//   D[co][pix] = sum_k W[co][k] * X[pix][k],  k = (tap, ci), NHWC activations, K-major packed weights.
//   The weight tile is the MFMA "A" operand and the pixel tile the "B" operand, so in the
//   32x32 accumulator layout each lane owns ONE pixel and runs of 4 consecutive output channels:
//   the epilogue (folded-BN scale/shift, residual, activation) then stores 8-byte (fp16) / 16-byte
//   (fp32) channel runs into NHWC, or lane-coalesced rows into NCHW fp32.
//   A transposed 4x4/s2/p1 conv is 4 output-parity phases, each a 2x2-tap conv:
//   blockIdx.z = phase, same kernel.
//   One 256-thread workgroup (4 wave64) computes a BP x BC tile.  Two K-loops share the epilogue:
//   * conv_igemm_dma_kernel (main path, channel-aligned layers): both operand tiles go HBM/L2 -> LDS
//     with buffer_load_dwordx4 ... lds (1 KiB per wave instruction, no VGPR round trip), through a
//     STAGES-deep LDS ring with ONE raw s_barrier per K-step and counted s_waitcnt vmcnt(N) so
//     STAGES-1 tiles stay in flight.  Padding taps and ragged tiles are out-of-range buffer offsets
//     (hardware returns 0), the tap offset is a scalar, per-row tap validity is a precomputed bitmask:
//     ~4 VALU per 16-byte vector instead of a 64-bit address computation.  The LDS image is XOR-
//     swizzled on the SOURCE side (DMA writes lane-linear) so ds_read_b128 fragment reads are
//     bank-conflict free.
//   * conv_igemm_kernel (generic path: Cin = 3/6/12 stems, Cout <= 32 heads): register-staged
//     global -> VGPR -> LDS, 2 stages, K runs over (tap, channel-group) pairs that may straddle taps.
//   fp16 uses v_mfma_f32_32x32x16_f16 (fp32 accumulate); fp32 uses v_mfma_f32_32x32x2_f32 (exact fp32
//   FMA chain) so the parity mode and the fast mode share every line of index arithmetic.
#include <stdlib.h>

#include <type_traits>

#include "ft_common.h"

#ifndef FT_DMA_FRAGDB
#define FT_DMA_FRAGDB 0   // 1: double-buffer the MFMA fragments in registers across K-steps (+32 VGPRs)
#endif
#ifndef FT_DMA_BKB
#define FT_DMA_BKB 64
#endif
#ifndef FT_DMA_STAGES
#define FT_DMA_STAGES 3   // measured on R50 B=64: 2 -> 1.88 ms, 3 -> 1.90, 4 -> 1.98, 5 -> 2.22 (occupancy beats ring depth)
#endif
#include "conv_common.h"

namespace ft {

// BP pixels x BC output channels per workgroup, waves arranged WGP x WGC, BKB bytes of K per
// tile row per K-step (64 -> 32 fp16 / 16 fp32 elements).
template <typename T, int BP, int BC, int WGP, int WGC, int BKB>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int CH = BKB / 16;             // 16-byte chunks per tile row
  constexpr int RPP = 256 / CH;            // tile rows covered by one pass of the 256 threads
  constexpr int NB = BP / RPP;             // pixel-tile vectors per thread
  constexpr int NA = (BC + RPP - 1) / RPP; // weight-tile vectors per thread
  constexpr int WT_P = BP / WGP, WT_C = BC / WGC;
  constexpr int MT_P = WT_P / 32, MT_C = WT_C / 32;
  constexpr int KK = BKB / 32;
  constexpr int SWZ_DIV = 256 / BKB;       // tile rows per 256-byte LDS bank row
  constexpr int A_BYTES = BC * BKB, B_BYTES = BP * BKB, STAGE = A_BYTES + B_BYTES;
  static_assert(WGP * WGC == 4, "4 waves per workgroup");
  static_assert(BP % RPP == 0 && WT_P % 32 == 0 && WT_C % 32 == 0, "tile shape");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wp = wave % WGP, wc = wave / WGP;
  const int lrow = tid / CH, chunk = tid % CH;

  // 1-D grid, XCD-aware: the dispatcher places block b on XCD b % 8 (speed only, never correctness), so
  // give every XCD one contiguous range of logical tiles.  Logical order = output-channel tile fastest,
  // then phase, then pixel tile: all blocks that re-read one activation tile (every co tile, every
  // transposed-conv phase) and the neighbouring halo rows meet in ONE XCD's L2 at about the same time.
  int ctile, phase, ptile;
  {
    const int total = p.npt * p.nct * p.nph;
    const int b = blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = b & 7, loc = b >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    ctile = logical % p.nct;
    const int t = logical / p.nct;
    phase = t % p.nph;
    ptile = t / p.nph;
  }
  const int py = phase >> 1, px = phase & 1;
  const int m0 = ptile * BP;
  const int co0 = ctile * BC;
  const int dbase_y = p.transposed ? py : -p.pad;
  const int dbase_x = p.transposed ? px : -p.pad_x;

  // ---- per-thread loader state -------------------------------------------------------------
  int b_row[NB], b_iy0[NB], b_ix0[NB];
  bool b_ok[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int m = m0 + lrow + j * RPP;
    b_ok[j] = m < p.M;
    const int mm = b_ok[j] ? m : 0;
    const int n = mm / p.HqWq;
    const int rem = mm - n * p.HqWq;
    const int qy = rem / p.Wq;
    const int qx = rem - qy * p.Wq;
    b_row[j] = n * p.Hi;
    b_iy0[j] = qy * p.sy + dbase_y;
    b_ix0[j] = qx * p.sy + dbase_x;
  }
  // K position of this thread's chunk: group index g -> (tap = (ky,kx), channel group cg)
  int cg, ky, kx;
  {
    const int tap = chunk / p.cin_groups;
    cg = chunk - tap * p.cin_groups;
    ky = tap / p.kw;
    kx = tap - ky * p.kw;
  }
  const size_t esz = sizeof(T);
  const char* wrow[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int r = lrow + i * RPP;
    wrow[i] = p.w + ((size_t)(phase * p.Cout_pad + co0 + (r < BC ? r : 0)) * p.Kpad + (size_t)chunk * VEC) * esz;
  }

  uint4_t ra[NA], rb[NB];
  auto load_tiles = [&](int ks) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int r = lrow + i * RPP;
      if (r < BC) ra[i] = *reinterpret_cast<const uint4_t*>(wrow[i] + (size_t)ks * (BKB));
    }
    const bool kvalid = ky < p.kh;
    const int dy = p.dmul * ky, dx = p.dmul * kx;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int iy = b_iy0[j] + dy, ix = b_ix0[j] + dx;
      const bool ok = b_ok[j] && kvalid && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      uint4_t v = {0u, 0u, 0u, 0u};
      if (ok) {
        const size_t off = ((size_t)(b_row[j] + iy) * p.Wi + ix) * p.x_cstride + p.x_coff + cg * VEC;
        v = *reinterpret_cast<const uint4_t*>(p.x + off * esz);
      }
      rb[j] = v;
    }
  };
  auto advance_k = [&]() {
    cg += CH;
    while (cg >= p.cin_groups) {
      cg -= p.cin_groups;
      if (++kx == p.kw) { kx = 0; ++ky; }
    }
  };
  auto store_tiles = [&](int stage) {
    char* sA = smem + stage * STAGE;
    char* sB = sA + A_BYTES;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int r = lrow + i * RPP;
      if (r < BC)
        *reinterpret_cast<uint4_t*>(sA + r * BKB + ((chunk ^ ((r / SWZ_DIV) % CH)) << 4)) = ra[i];
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int r = lrow + j * RPP;
      *reinterpret_cast<uint4_t*>(sB + r * BKB + ((chunk ^ ((r / SWZ_DIV) % CH)) << 4)) = rb[j];
    }
  };

  float16_t acc[MT_C][MT_P];
#pragma unroll
  for (int i = 0; i < MT_C; ++i)
#pragma unroll
    for (int j = 0; j < MT_P; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_tiles(0);
  store_tiles(0);
  __syncthreads();

  const int l31 = lane & 31, lhi = lane >> 5;
  for (int ks = 0; ks < p.nk; ++ks) {
    const int cur = ks & 1;
    const bool more = ks + 1 < p.nk;
    if (more) {
      advance_k();
      load_tiles(ks + 1);
    }
    const char* sA = smem + cur * STAGE;
    const char* sB = sA + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      uint4_t a[MT_C], b[MT_P];
      const int c = kk * 2 + lhi;
#pragma unroll
      for (int i = 0; i < MT_C; ++i) {
        const int r = wc * WT_C + i * 32 + l31;
        a[i] = *reinterpret_cast<const uint4_t*>(sA + r * BKB + ((c ^ ((r / SWZ_DIV) % CH)) << 4));
      }
#pragma unroll
      for (int j = 0; j < MT_P; ++j) {
        const int r = wp * WT_P + j * 32 + l31;
        b[j] = *reinterpret_cast<const uint4_t*>(sB + r * BKB + ((c ^ ((r / SWZ_DIV) % CH)) << 4));
      }
      mma_slice<MT_C, MT_P>(a, b, acc, (T*)nullptr);
    }
    if (more) store_tiles(cur ^ 1);
    __syncthreads();
  }

  conv_epilogue<T, BP, BC, WGP, WGC>(p, acc, smem, 2 * STAGE, m0, co0, py, px);
}

// ---- main path: direct-to-LDS, STAGES-deep ring, counted vmcnt -----------------------------------------
// Requirements (checked on the host, else the generic kernel runs): kh*kw <= 32, BC >= 64, every tap's
// channel run padded to a multiple of BK = BKB/sizeof(T) in BOTH the packed weights and the activation
// pixel stride (x_cstride >= x_coff + cin_pad, padding channels zero), buffers < 2 GiB.
// Occupancy target (waves per SIMD) the register allocator must respect: the K-loop is latency-bound per
// workgroup (one barrier per K-step), so co-resident workgroups are what keeps the MFMA pipe fed.
// Workgroup barrier of the K-loops.  The builtin is IntrNoMem for LLVM: ds_reads that follow it in program order may be
// hoisted ABOVE it (measured: the stem kernel read patch rows other waves' LDS-DMA had not landed yet, ~0.1 % of the
// tiles wrong once workgroups are recycled on a CU).  The inline-asm form with a memory clobber pins the order.
#define FT_LDS_BARRIER() asm volatile("s_barrier" ::: "memory")
#ifndef FT_HALO_KEY_SHIFT
#define FT_HALO_KEY_SHIFT 1    // dev A/B: 0 = the old `r & 7` patch key of conv_halo_kernel's 128-byte rows
#endif
#ifndef FT_EPI_NT
#define FT_EPI_NT 0     // non-temporal stores in the fp16 epilogue (dev A/B)
#endif
#ifndef FT_DMA_INTERLEAVE
#define FT_DMA_INTERLEAVE 1
#endif
#ifndef FT_DMA_LEAN
#define FT_DMA_LEAN 1
#endif
#ifndef FT_DMA_WAVES_BIG
#define FT_DMA_WAVES_BIG 3   // 128x128 tile: 64 accumulator + <= 104 other registers
#endif
// KS > 1 = intra-workgroup split-K for layers with few tiles but a long K (layer3/4, deconv.0 at batch 64, every
// layer at batch 1): KS groups of 4 waves each run the K-loop over 1/KS of the K-steps with their own LDS ring
// (same tile, same barriers), then partial accumulators are summed through LDS and group 0 runs the epilogue.
// The K-loop of ONE wave is a ~600-cycle serial chain per K-step with 128-256 cycles of MFMA in it; what fills
// the matrix pipe is other waves, and a small layer has no other tiles to offer — so the extra waves come from K.
template <typename T, int BP, int BC, int WGP, int WGC, int BKB, int STAGES, bool HAS_RES, int KS>
__global__ __launch_bounds__(64 * WGP * WGC * KS,
                             (KS > 1 || WGP * WGC > 4 ? 1 : (BKB > 64 ? 2 : (BP * BC >= 128 * 128 ? FT_DMA_WAVES_BIG : 4))))
void conv_igemm_dma_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // the body uses gfx950-only types (__amdgpu_buffer_rsrc_t); the host pass only needs the stub
  constexpr int NW = WGP * WGC;             // waves per K-group: 4 (128x128 and smaller tiles) or 8 (256x128)
  constexpr int NT = 64 * NW;
  constexpr int CH = BKB / 16;              // 16-byte chunks per tile row
  constexpr int RPI = 64 / CH;              // tile rows filled by one 1-KiB wave load
  constexpr int NIA = BC / RPI / NW;        // weight-tile loads per wave per stage
  constexpr int NIB = BP / RPI / NW;        // pixel-tile loads per wave per stage
  constexpr int NL = NIA + NIB;
  constexpr int WT_P = BP / WGP, WT_C = BC / WGC;
  constexpr int MT_P = WT_P / 32, MT_C = WT_C / 32;
  constexpr int KK = BKB / 32;
  constexpr int SWZ_DIV = 256 / BKB;
  constexpr int A_BYTES = BC * BKB, B_BYTES = BP * BKB, STAGE = A_BYTES + B_BYTES;
  static_assert((NW == 4 || NW == 8) && NIA >= 1 && NIB >= 1 && (KS == 1 || NW == 4), "tile shape");
  static_assert(BC % (RPI * NW) == 0 && BP % (RPI * NW) == 0, "tile rows must split evenly over the waves");
  static_assert(NL * (STAGES - 1) <= 63, "vmcnt is a 6-bit counter");
  static_assert(STAGES >= (FT_DMA_FRAGDB ? 3 : 2), "ring depth");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave_all / NW;            // K-split group (0 when KS == 1)
  const int wave = wave_all % NW;           // wave inside the group
  const int wp = wave % WGP, wc = wave / WGP;
  char* const gsm = smem + grp * (STAGES * STAGE);   // this group's ring

  int ctile, phase, ptile, ksplit;
  {
    const int tiles = p.npt * p.nct * p.nph;
    const int total = tiles * p.sk;
    const int b = blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = b & 7, loc = b >> 3;
    int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    ksplit = logical / tiles;               // cross-workgroup K split: this workgroup's share of the K-steps
    logical -= ksplit * tiles;
    ctile = logical % p.nct;
    const int t = logical / p.nct;
    phase = t % p.nph;
    ptile = t / p.nph;
  }
  const int py = phase >> 1, px = phase & 1;
  const int m0 = ptile * BP;
  const int co0 = ctile * BC;
  const int dbase_y = p.transposed ? py : -p.pad;
  const int dbase_x = p.transposed ? px : -p.pad_x;
  constexpr int esz = (int)sizeof(T);
  // K-steps of this group: [ks_begin, ks_end); every group iterates nk_g times so the barriers line up.
  // With p.sk > 1 (KS == 1) the workgroup owns the ksplit-th slice of the K-steps instead.
  const int nk_sk = (p.nk + p.sk - 1) / p.sk;
  const int sk_lo = ksplit * nk_sk, sk_hi = sk_lo + nk_sk < p.nk ? sk_lo + nk_sk : p.nk;
  const int nk_g = KS > 1 ? (p.nk + KS - 1) / KS : (sk_hi > sk_lo ? sk_hi - sk_lo : 0);
  const int ks_begin = KS > 1 ? grp * nk_g : sk_lo;
  const int ks_end = KS > 1 ? (ks_begin + nk_g < p.nk ? ks_begin + nk_g : p.nk) : sk_hi;

  // buffer descriptors: weights of this (phase, co tile); the whole activation buffer
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(p.w) + (size_t)(phase * p.Cout_pad + co0) * p.Kpad * esz, 0, BC * p.Kpad * esz, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;    // >= any num_records we accept: the load returns zeros

  // ---- per-lane constants of the loader ---------------------------------------------------------------
  const int lrow = lane / CH, pos = lane % CH;
  unsigned a_voff[NIA];
#pragma unroll
  for (int t = 0; t < NIA; ++t) {
    const int r = (wave + NW * t) * RPI + lrow;
    const int lc = pos ^ ((r / SWZ_DIV) % CH);           // source chunk that lands at LDS position `pos`
    a_voff[t] = (p.dbg & 256) ? kOOB : (unsigned)(r * p.Kpad * esz + lc * 16);   // (dev probe 256: weight-tile loads fetch nothing)
  }
  int b_base[NIB];
  unsigned b_mask[NIB];
#pragma unroll
  for (int t = 0; t < NIB; ++t) {
    const int r = (wave + NW * t) * RPI + lrow;
    const int lc = pos ^ ((r / SWZ_DIV) % CH);
    const int m = m0 + r;
    unsigned mask = 0;
    int base = 0;
    if (m < p.M) {
      const int n = m / p.HqWq;
      const int rem = m - n * p.HqWq;
      const int qy = rem / p.Wq;
      const int qx = rem - qy * p.Wq;
      const int iy0 = qy * p.sy + dbase_y, ix0 = qx * p.sy + dbase_x;
      base = (((n * p.Hi + iy0) * p.Wi + ix0) * p.x_cstride + p.x_coff) * esz + lc * 16;
      for (int ky = 0; ky < p.kh; ++ky)
        for (int kx = 0; kx < p.kw; ++kx) {
          const int iy = iy0 + p.dmul * ky, ix = ix0 + p.dmul * kx;
          if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi) mask |= 1u << (ky * p.kw + kx);
        }
    }
    b_base[t] = base;
    b_mask[t] = mask;
  }

  // second input (K-concat): one more K-run per pixel row, read from another tensor at (n, qy * s2, qx * s2)
  const bool dual = p.kc2 > 0;
  const __amdgpu_buffer_rsrc_t rsrc_b2 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dual ? p.x2 : p.x), 0, dual ? p.x2_bytes : p.x_bytes, 0x00020000);
  unsigned b2_voff[NIB];
#pragma unroll
  for (int t = 0; t < NIB; ++t) {
    unsigned v = kOOB;
    if (dual) {
      const int r = (wave + NW * t) * RPI + lrow;
      const int lc = pos ^ ((r / SWZ_DIV) % CH);
      const int m = m0 + r;
      if (m < p.M) {
        const int n = m / p.HqWq;
        const int rem = m - n * p.HqWq;
        const int qy = rem / p.Wq;
        const int qx = rem - qy * p.Wq;
        v = (unsigned)((((n * p.x2_hi + qy * p.x2_stride) * p.x2_wi + qx * p.x2_stride) * p.x2_cstride + p.x2_coff) * esz + lc * 16);
      }
    }
    b2_voff[t] = v;
  }

  // ---- K position of the NEXT stage to issue (wave-uniform, lives in SGPRs) -----------------------------
  int i_ks = ks_begin, i_cc, i_ky, i_kx;
  {
    const int tap0 = ks_begin / p.kc;
    i_cc = ks_begin - tap0 * p.kc;
    i_ky = tap0 / p.kw;
    i_kx = tap0 - i_ky * p.kw;
  }
  const int cstride_b = p.x_cstride * esz;
#ifdef FT_CONV_TIMING
  unsigned long long tiss[3] = {0, 0, 0};
#define FT_TI(i) do { const unsigned long long _t = __builtin_readcyclecounter(); tiss[i] += _t - tip; tip = _t; } while (0)
#else
#define FT_TI(i) do { } while (0)
#endif
  auto issue = [&](int stage) {
#ifdef FT_CONV_TIMING
    unsigned long long tip = __builtin_readcyclecounter();
#endif
    char* sA = gsm + stage * STAGE;
    char* sB = sA + A_BYTES;
    const bool live = i_ks < ks_end;
    const int a_soff = i_ks * BKB;
    const int tap = i_ky * p.kw + i_kx;
    const int delta = ((p.dmul * i_ky) * p.Wi + p.dmul * i_kx) * cstride_b + i_cc * BKB;
    const unsigned tapbit = live ? (1u << tap) : 0u;
    FT_TI(0);
#pragma unroll
    for (int t = 0; t < NIA; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr)(sA + (wave + NW * t) * 1024), 16,
                                               live ? a_voff[t] : kOOB, live ? a_soff : 0, 0, 0);
    FT_TI(1);
#pragma unroll
    for (int t = 0; t < NIB; ++t) {
      const unsigned voff = (b_mask[t] & tapbit) ? (unsigned)(b_base[t] + delta) : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr)(sB + (wave + NW * t) * 1024), 16, voff, 0, 0, 0);
    }
    FT_TI(2);
    ++i_ks;
    if (++i_cc == p.kc) {
      i_cc = 0;
      if (++i_kx == p.kw) { i_kx = 0; ++i_ky; }
    }
  };

  // issue state of the stage being filled (wave-uniform scalars), split from the loads themselves so the
  // K-loop can interleave ONE load after each MFMA: a blocked vector-memory issue then hides under the
  // matrix instruction that is still executing instead of leaving the pipe empty (FT_DMA_INTERLEAVE)
  char* is_sA = gsm;
  bool is_live = false;
  int is_asoff = 0, is_delta = 0;
  unsigned is_tapbit = 0;
  auto issue_prep = [&](int stage) {
    is_sA = gsm + stage * STAGE;
    is_live = i_ks < ks_end;
    is_asoff = i_ks * BKB;
    const int tap = i_ky * p.kw + i_kx;
    is_delta = ((p.dmul * i_ky) * p.Wi + p.dmul * i_kx) * cstride_b + i_cc * BKB;
    is_tapbit = is_live ? (1u << tap) : 0u;
    ++i_ks;
    if (++i_cc == p.kc) {
      i_cc = 0;
      if (++i_kx == p.kw) { i_kx = 0; ++i_ky; }
    }
  };
  auto issue_one = [&](auto idx) {
    constexpr int t = decltype(idx)::value;
    if constexpr (t < NIA) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr)(is_sA + (wave + NW * t) * 1024), 16,
                                               is_live ? a_voff[t] : kOOB, is_live ? is_asoff : 0, 0, 0);
    } else {
      constexpr int u = t - NIA;
      const unsigned voff = (b_mask[u] & is_tapbit) ? (unsigned)(b_base[u] + is_delta) : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr)(is_sA + A_BYTES + (wave + NW * u) * 1024), 16, voff, 0, 0, 0);
    }
  };
  // ---- per-lane fragment read offsets (stage-relative); kk selects chunk pair via XOR (kk << 5) ----------
  const int l31 = lane & 31, lhi = lane >> 5;
  int a_off[MT_C], b_off[MT_P];
#pragma unroll
  for (int i = 0; i < MT_C; ++i) {
    const int r = wc * WT_C + i * 32 + l31;
    a_off[i] = r * BKB + ((lhi ^ ((r / SWZ_DIV) % CH)) << 4);
  }
#pragma unroll
  for (int j = 0; j < MT_P; ++j) {
    const int r = wp * WT_P + j * 32 + l31;
    b_off[j] = A_BYTES + r * BKB + ((lhi ^ ((r / SWZ_DIV) % CH)) << 4);
  }

  float16_t acc[MT_C][MT_P];
#pragma unroll
  for (int i = 0; i < MT_C; ++i)
#pragma unroll
    for (int j = 0; j < MT_P; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Residual tile: issue its (coalesced, 16-byte) loads NOW so their HBM latency hides under the operand
  // pipeline instead of sitting between the last MFMA and the store (1x1 bottleneck-exit layers are
  // bandwidth-bound: what matters is bytes in flight per CU).
  constexpr bool PRE = HAS_RES && sizeof(T) == 2;
  constexpr int NPRE = PRE ? BP * (BC / 8) / NT : 1;
  uint4_t rpre[NPRE];
  long long* s_opix = reinterpret_cast<long long*>(smem + KS * STAGES * STAGE);
  if constexpr (PRE) {
    conv_row_table<BP>(p, s_opix, m0, py, px);
    __syncthreads();
    constexpr int NCH = BC / 8;
    const half_t* rbase = reinterpret_cast<const half_t*>(p.res) + p.res_coff + co0;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      if (KS > 1 && grp != 0) { rpre[k] = uint4_t{0u, 0u, 0u, 0u}; continue; }
      const int idx = tid + k * NT;
      const int pl = idx / NCH, ch = idx % NCH;
      const long long o = s_opix[pl];
      uint4_t v = {0u, 0u, 0u, 0u};
      if (o >= 0 && co0 + ch * 8 < p.Cout) v = *reinterpret_cast<const uint4_t*>(rbase + o * p.res_cstride + ch * 8);
      rpre[k] = v;
    }
  }

  // experiment: break the phase lock between co-resident workgroups (all start together and would otherwise
  // hit their load / MFMA / store phases at the same time)
  if (p.dbg & 24) {
    const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 15u;   // HW_ID.WAVE_ID
    if (p.dbg & 8) {
      if ((slot % 3) == 1) __builtin_amdgcn_s_sleep(12);
      else if ((slot % 3) == 2) __builtin_amdgcn_s_sleep(24);
    }
    if (p.dbg & 16) {
      if (slot & 1) __builtin_amdgcn_s_setprio(1);
    }
  }

#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) issue(s);

  int cur = 0, nxt = STAGES - 1;   // ring slot of K-step ks; slot the next issue() fills
#if FT_DMA_FRAGDB
  // Fragment registers are double-buffered: while the MFMAs of K-step ks run from set P, the
  // ds_read_b128s of K-step ks+1 fill set P^1, so no MFMA ever waits on LDS latency right after a
  // barrier.  (Static set indices: the loop is unrolled by two through step<P>().)
  uint4_t fa[2][KK][MT_C], fb[2][KK][MT_P];
  auto load_frags = [&](auto set, int slot) {
    constexpr int P = decltype(set)::value;
    const char* st = gsm + slot * STAGE;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
      for (int i = 0; i < MT_C; ++i) fa[P][kk][i] = *reinterpret_cast<const uint4_t*>(st + (a_off[i] ^ (kk << 5)));
#pragma unroll
      for (int j = 0; j < MT_P; ++j) fb[P][kk][j] = *reinterpret_cast<const uint4_t*>(st + (b_off[j] ^ (kk << 5)));
    }
  };
  auto step = [&](auto set) {
    constexpr int P = decltype(set)::value;
    const int cur1 = cur + 1 == STAGES ? 0 : cur + 1;
    // this wave's loads of K-step ks+1 have landed (STAGES-3 younger stages may still be in flight) ...
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL * (STAGES - 3)) : "memory");
    // ... after the barrier everyone's have, and everyone is done reading the slot issue() refills
    FT_LDS_BARRIER();
    issue(nxt);
    load_frags(std::integral_constant<int, P ^ 1>{}, cur1);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) mma_slice<MT_C, MT_P>(fa[P][kk], fb[P][kk], acc, (T*)nullptr);
    cur = cur1;
    nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
  };

  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL * (STAGES - 2)) : "memory");
  FT_LDS_BARRIER();
  load_frags(std::integral_constant<int, 0>{}, 0);
  int ks = 0;
  for (; ks + 1 < nk_g; ks += 2) {
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
  }
  if (ks < nk_g) step(std::integral_constant<int, 0>{});
#else
#ifdef FT_CONV_TIMING   // developer build: per-phase s_memtime accounting of the K-loop, dumped through p.y
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long t_start = __builtin_readcyclecounter();
#define FT_T(i)                                           \
  do {                                                    \
    const unsigned long long _t = __builtin_readcyclecounter(); \
    tacc[i] += _t - tprev;                                \
    tprev = _t;                                           \
  } while (0)
#else
#define FT_T(i) do { } while (0)
#endif
#if FT_DMA_LEAN && !defined(FT_CONV_TIMING)
  constexpr bool kLean = (KS == 1);
#else
  constexpr bool kLean = false;
#endif
  if constexpr (kLean) {
    // ---- lean K-loop: unrolled by STAGES so every ring slot / LDS offset is an immediate, the pixel-tile
    // offsets are recomputed only when the TAP changes (the channel walk inside a tap is the scalar soffset of
    // the load), loads are interleaved one per MFMA pair, and the tail (no loads left to issue) is peeled so the
    // main loop carries no "live" predicate.  ~35 instructions per K-step instead of ~100: the K-loop is issue-
    // bound on its scalar / address arithmetic long before the matrix pipe is full (profiles/README.md).
    unsigned cur_voff[NIB];
    auto refresh = [&]() {
      const int tap = i_ky * p.kw + i_kx;
      const int delta = ((p.dmul * i_ky) * p.Wi + p.dmul * i_kx) * cstride_b;
      const unsigned tapbit = 1u << tap;
#pragma unroll
      for (int t = 0; t < NIB; ++t) cur_voff[t] = (b_mask[t] & tapbit) ? (unsigned)(b_base[t] + delta) : kOOB;
      if (p.dbg & 64) {   // dev probe: pixel-tile loads fetch nothing (upper bound of an LDS-resident halo patch)
#pragma unroll
        for (int t = 0; t < NIB; ++t) cur_voff[t] = kOOB;
      }
    };
    int src = 0, kc_cur = p.kc;              // K-run being walked: 0 = the taps of x, 1 = the second input
    if (i_ky < p.kh) refresh();
    else if (dual) {                          // the prologue already consumed all of x's K-run
      src = 1;
      kc_cur = p.kc2;
#pragma unroll
      for (int t = 0; t < NIB; ++t) cur_voff[t] = b2_voff[t];
    } else {
#pragma unroll
      for (int t = 0; t < NIB; ++t) cur_voff[t] = kOOB;
    }
    int soff_a = i_ks * BKB, soff_b = i_cc * BKB;
    auto lean_issue_one = [&](auto idx, auto slot_c) {
      constexpr int t = decltype(idx)::value;
      constexpr int slot = decltype(slot_c)::value;
      if constexpr (t < NIA) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr)(smem + slot * STAGE + (wave + NW * t) * 1024), 16, a_voff[t],
                                                 soff_a, 0, 0);
      } else {
        constexpr int u = t - NIA;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(src ? rsrc_b2 : rsrc_b, (lds_ptr)(smem + slot * STAGE + A_BYTES + (wave + NW * u) * 1024),
                                                 16, cur_voff[u], soff_b, 0, 0);
      }
    };
    auto advance = [&]() {
      soff_a += BKB;
      soff_b += BKB;
      if (++i_cc == kc_cur) {
        i_cc = 0;
        soff_b = 0;
        if (src == 0) {
          if (++i_kx == p.kw) { i_kx = 0; ++i_ky; }
          if (i_ky < p.kh) refresh();
          else if (dual) {
            src = 1;
            kc_cur = p.kc2;
#pragma unroll
            for (int t = 0; t < NIB; ++t) cur_voff[t] = b2_voff[t];
          }
        }
      }
    };
    constexpr int NM = (sizeof(T) == 2 ? 1 : 4) * KK * MT_C * MT_P;   // MFMAs per K-step
    constexpr int GAP = NM >= NL ? NM / NL : 1;                        // one load after every GAP-th MFMA
    auto lstep = [&](auto slot_c, auto issue_c, auto wait_c) {
      constexpr int slot = decltype(slot_c)::value;
      constexpr bool do_issue = decltype(issue_c)::value;
      constexpr int nslot = (slot + STAGES - 1) % STAGES;
      // this wave's loads of this K-step have landed (wait_c younger loads may stay in flight) ...
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(decltype(wait_c)::value) : "memory");
      // ... after the barrier everyone's have, and everyone is done reading the slot refilled below
      FT_LDS_BARRIER();
      const char* st = smem + slot * STAGE;
      // 8 accumulator tiles per wave (the 256 x 256 tile): fragments are read slice by slice, 24 instead of 96 registers
      // (two waves per SIMD share the 512-entry file: 128 accumulators + all four slices at once spilled 300 registers)
      constexpr bool kPerSlice = sizeof(T) == 2 && MT_C * MT_P >= 8 && KK > 2;
      if constexpr (kPerSlice) {
        static_for<KK>([&](auto kkc) {
          constexpr int kk = decltype(kkc)::value;
          uint4_t fa1[MT_C], fb1[MT_P];
#pragma unroll
          for (int i = 0; i < MT_C; ++i) fa1[i] = *reinterpret_cast<const uint4_t*>(st + (a_off[i] ^ (kk << 5)));
#pragma unroll
          for (int j = 0; j < MT_P; ++j) fb1[j] = *reinterpret_cast<const uint4_t*>(st + (b_off[j] ^ (kk << 5)));
          static_for<MT_C * MT_P>([&](auto mi) {
            constexpr int m = kk * MT_C * MT_P + decltype(mi)::value;
            constexpr int i = (m / MT_P) % MT_C, j = m % MT_P;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, fa1[i]),
                                                               __builtin_bit_cast(half8_t, fb1[j]), acc[i][j], 0, 0, 0);
            if constexpr (do_issue && (m % GAP) == GAP - 1 && m / GAP < NL) {
              lean_issue_one(std::integral_constant<int, m / GAP>{}, std::integral_constant<int, nslot>{});
              __builtin_amdgcn_sched_barrier(0);
            }
          });
        });
      }
      uint4_t fa[kPerSlice ? 1 : KK][MT_C], fb[kPerSlice ? 1 : KK][MT_P];
      if constexpr (!kPerSlice) {
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
        for (int i = 0; i < MT_C; ++i) fa[kk][i] = *reinterpret_cast<const uint4_t*>(st + (a_off[i] ^ (kk << 5)));
#pragma unroll
        for (int j = 0; j < MT_P; ++j) fb[kk][j] = *reinterpret_cast<const uint4_t*>(st + (b_off[j] ^ (kk << 5)));
      }
      }
      if constexpr (!kPerSlice)
      static_for<NM>([&](auto mi) {
        constexpr int m = decltype(mi)::value;
        constexpr int i = (m / MT_P) % MT_C, j = m % MT_P;
        if constexpr (sizeof(T) == 2) {
          constexpr int kk = m / (MT_C * MT_P);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, fa[kk][i]),
                                                             __builtin_bit_cast(half8_t, fb[kk][j]), acc[i][j], 0, 0, 0);
        } else {
          constexpr int kk = m / (4 * MT_C * MT_P), e = (m / (MT_C * MT_P)) % 4;
          const float4_t af = __builtin_bit_cast(float4_t, fa[kk][i]), bf = __builtin_bit_cast(float4_t, fb[kk][j]);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc[i][j], 0, 0, 0);
        }
        if constexpr (do_issue && (m % GAP) == GAP - 1 && m / GAP < NL) {
          lean_issue_one(std::integral_constant<int, m / GAP>{}, std::integral_constant<int, nslot>{});
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      if constexpr (do_issue) {
        if constexpr (NM < NL)
          static_for_from<NM, NL>([&](auto t) { lean_issue_one(t, std::integral_constant<int, nslot>{}); });
        advance();
      }
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    using wmain = std::integral_constant<int, NL * (STAGES - 2)>;
    const int n_main = nk_g > STAGES - 1 ? nk_g - (STAGES - 1) : 0;   // K-steps that still have a stage to issue
    const int tn = nk_g < STAGES - 1 ? nk_g : STAGES - 1;             // peeled tail steps
    int ks = 0;
    for (; ks + STAGES <= n_main; ks += STAGES)
      static_for<STAGES>([&](auto sc) { lstep(sc, yes{}, wmain{}); });
    const int rem = n_main - ks;
    static_for<STAGES>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      if (rem == r) {
        static_for<r>([&](auto sc) { lstep(sc, yes{}, wmain{}); });
        static_for<STAGES - 1>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          if (t < tn)
            lstep(std::integral_constant<int, (r + t) % STAGES>{}, no{}, std::integral_constant<int, NL * (STAGES - 2 - t)>{});
        });
      }
    });
  }
  if constexpr (!kLean)
  for (int ks = 0; ks < nk_g; ++ks) {
#ifdef FT_CONV_TIMING
    unsigned long long tprev = __builtin_readcyclecounter();
#endif
    // this wave's loads of K-step ks have landed (STAGES-2 younger stages may still be in flight) ...
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NL * (STAGES - 2)) : "memory");
    FT_T(0);
    // ... after the barrier everyone's have, and everyone is done reading the slot issue() refills
    FT_LDS_BARRIER();
    FT_T(1);
#if FT_DMA_INTERLEAVE && !defined(FT_CONV_TIMING)
    if constexpr (sizeof(T) == 2) {
      issue_prep(nxt);
      const char* st = gsm + cur * STAGE;
      uint4_t fa[KK][MT_C], fb[KK][MT_P];
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
        for (int i = 0; i < MT_C; ++i) fa[kk][i] = *reinterpret_cast<const uint4_t*>(st + (a_off[i] ^ (kk << 5)));
#pragma unroll
        for (int j = 0; j < MT_P; ++j) fb[kk][j] = *reinterpret_cast<const uint4_t*>(st + (b_off[j] ^ (kk << 5)));
      }
      constexpr int NM = KK * MT_C * MT_P;                      // MFMAs per K-step
      constexpr int GAP = NM >= NL ? NM / NL : 1;               // one load after every GAP-th MFMA
      auto mm = [&](auto mi) {
        constexpr int m = decltype(mi)::value;
        constexpr int kk = m / (MT_C * MT_P), i = (m / MT_P) % MT_C, j = m % MT_P;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, fa[kk][i]),
                                                           __builtin_bit_cast(half8_t, fb[kk][j]), acc[i][j], 0, 0, 0);
        if constexpr ((m % GAP) == GAP - 1 && m / GAP < NL) {
          issue_one(std::integral_constant<int, m / GAP>{});
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      static_for<NM>(mm);
      if constexpr (NM < NL) static_for_from<NM, NL>([&](auto t) { issue_one(t); });
      cur = cur + 1 == STAGES ? 0 : cur + 1;
      nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
      continue;
    }
#endif
    if (!(p.dbg & 2)) issue(nxt);
    FT_T(2);
    const char* st = gsm + cur * STAGE;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      uint4_t a[MT_C], b[MT_P];
#pragma unroll
      for (int i = 0; i < MT_C; ++i) a[i] = *reinterpret_cast<const uint4_t*>(st + (a_off[i] ^ (kk << 5)));
#pragma unroll
      for (int j = 0; j < MT_P; ++j) b[j] = *reinterpret_cast<const uint4_t*>(st + (b_off[j] ^ (kk << 5)));
#ifdef FT_CONV_TIMING
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      FT_T(3);
#endif
      if (p.dbg & 1) {
#pragma unroll
        for (int i = 0; i < MT_C; ++i) asm volatile("" ::"v"(a[i]));
#pragma unroll
        for (int j = 0; j < MT_P; ++j) asm volatile("" ::"v"(b[j]));
      } else {
        mma_slice<MT_C, MT_P>(a, b, acc, (T*)nullptr);
      }
      FT_T(4);
    }
    cur = cur + 1 == STAGES ? 0 : cur + 1;
    nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
    FT_T(5);
  }
#ifdef FT_CONV_TIMING
  if (p.dbg & 32) {
    const unsigned long long t_loop = __builtin_readcyclecounter() - t_start;
    if (lane == 0 && wave_all == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2)) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.y) + (blockIdx.x == 0 ? 0 : 8);
      for (int i = 0; i < 6; ++i) o[i] = tacc[i];
      o[6] = t_loop;
      o[7] = (unsigned long long)nk_g;
      unsigned long long* o2 = reinterpret_cast<unsigned long long*>(p.y) + 16 + (blockIdx.x == 0 ? 0 : 4);
      for (int i = 0; i < 3; ++i) o2[i] = tiss[i];
    }
    return;
  }
#endif
#endif
  // drain the (all out-of-range) tail loads before LDS is reused by the epilogue
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (p.dbg & 4) {
    if (acc[0][0][0] == 12345.678f) p.y[0] = 1;   // keep the accumulators live
    return;
  }
  if constexpr (KS > 1) {
    // sum the K-split partials through LDS (the rings are dead): group g > 0 parks its accumulators lane-
    // contiguously (16-byte stores, conflict-free), group 0 adds them; then only group 0 lives on (s_barrier
    // ignores terminated waves)
    constexpr int NV4 = MT_C * MT_P * 4;                       // float4s per lane
    float4_t* part = reinterpret_cast<float4_t*>(smem);
    if (grp != 0) {
      float4_t* dst = part + ((size_t)((grp - 1) * NW + wave) * NV4) * 64 + lane;
#pragma unroll
      for (int i = 0; i < MT_C; ++i)
#pragma unroll
        for (int j = 0; j < MT_P; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4_t v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            dst[((i * MT_P + j) * 4 + q) * 64] = v;
          }
    }
    __syncthreads();
    if (grp != 0) return;
#pragma unroll
    for (int g = 1; g < KS; ++g) {
      const float4_t* src = part + ((size_t)((g - 1) * NW + wave) * NV4) * 64 + lane;
#pragma unroll
      for (int i = 0; i < MT_C; ++i)
#pragma unroll
        for (int j = 0; j < MT_P; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4_t v = src[((i * MT_P + j) * 4 + q) * 64];
            acc[i][j][4 * q] += v[0]; acc[i][j][4 * q + 1] += v[1]; acc[i][j][4 * q + 2] += v[2]; acc[i][j][4 * q + 3] += v[3];
          }
    }
    __syncthreads();   // group 0 only: the partial region is about to be reused by the epilogue
  }
  if (p.sk > 1) {
    // cross-workgroup split-K: raw fp32 partial tile -> workspace [ksplit][phase * M + pixel][Cout_pad]; the scale /
    // shift / residual / activation epilogue runs in conv_splitk_reduce_kernel once every slice has landed
    const int l31 = lane & 31, lhi = lane >> 5;
    float* wsb = p.ws + ((size_t)ksplit * p.nph + phase) * (size_t)p.M * p.Cout_pad;
#pragma unroll
    for (int j = 0; j < MT_P; ++j) {
      const int m = m0 + wp * WT_P + j * 32 + l31;
      if (m >= p.M) continue;
#pragma unroll
      for (int i = 0; i < MT_C; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int cb = co0 + wc * WT_C + i * 32 + 8 * rg + 4 * lhi;
          const float4_t v = {acc[i][j][rg * 4], acc[i][j][rg * 4 + 1], acc[i][j][rg * 4 + 2], acc[i][j][rg * 4 + 3]};
          *reinterpret_cast<float4_t*>(wsb + (size_t)m * p.Cout_pad + cb) = v;
        }
    }
    return;
  }
  conv_epilogue<T, BP, BC, WGP, WGC, PRE, NT>(p, acc, smem, KS * STAGES * STAGE, m0, co0, py, px, rpre);
#endif
}

// Sum of the split-K partial tiles + the fused epilogue (folded BN / bias, residual, activation), 4 channels per lane.
template <typename T>
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvParams p) {
  const int c4n = p.Cout_pad / 4;
  const size_t total = (size_t)p.nph * p.M * c4n;
  const size_t slice = (size_t)p.nph * p.M * p.Cout_pad;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int c4 = (int)(idx % c4n);
    const size_t pm = idx / c4n;
    const int cb = c4 * 4;
    if (cb >= p.Cout) continue;
    float4_t a = *reinterpret_cast<const float4_t*>(p.ws + pm * p.Cout_pad + cb);
    for (int s = 1; s < p.sk; ++s) a += *reinterpret_cast<const float4_t*>(p.ws + s * slice + pm * p.Cout_pad + cb);
    const int phase = (int)(pm / p.M), m = (int)(pm - (size_t)phase * p.M);
    const int n = m / p.HqWq, rem = m - n * p.HqWq, qy = rem / p.Wq, qx = rem - qy * p.Wq;
    const int oy = qy * p.omul + (phase >> 1), ox = qx * p.omul + (phase & 1);
    const size_t opix = ((size_t)n * p.Ho + oy) * p.Wo + ox;
    float v[4] = {a[0], a[1], a[2], a[3]};
    if (p.scale) {
      const float4_t sc = *reinterpret_cast<const float4_t*>(p.scale + cb);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= sc[e];
    }
    if (p.shift) {
      const float4_t sh = *reinterpret_cast<const float4_t*>(p.shift + cb);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += sh[e];
    }
    const bool full = cb + 3 < p.Cout;
    if (p.res) {
      const T* rp = reinterpret_cast<const T*>(p.res) + opix * p.res_cstride + p.res_coff + cb;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (cb + e < p.Cout) v[e] += (float)rp[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act, p.slope);
    if (p.out_layout == FT_LAYOUT_NHWC) {
      T* yp = reinterpret_cast<T*>(p.y) + opix * p.y_cstride + p.y_coff + cb;
      if (full) store4(yp, v);
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (cb + e < p.Cout) yp[e] = (T)v[e];
      }
    } else {
      float* yp = reinterpret_cast<float*>(p.y);
      const size_t hw = (size_t)p.Ho * p.Wo, pix = (size_t)oy * p.Wo + ox;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (cb + e < p.Cout) yp[((size_t)n * p.Cout + cb + e) * hw + pix] = v[e];
    }
  }
}

// ---- halo path (fp16): 3x3/s1 convs and the 2x2-tap phases of ConvTranspose2d(4,2,1) -------------------
// The implicit-GEMM kernel above re-reads every input pixel once per tap through L2 -> LDS (9x for a 3x3), and
// that operand delivery, not the matrix pipe, bounds it.  Here a workgroup owns a TH x TW patch of ONE image,
// keeps the (TH+kh-1) x (TW+kw-1) input patch of a 64-channel chunk resident in LDS (double-buffered across
// chunks) and serves the pixel operand of every tap from it; only the weight tile still streams through the
// ring.  L2 -> LDS bytes per K-step drop from (BC + 128) x 64 to BC x 64 + ~1/6 of the pixel tile.
//   K order: chunk (64 channels) outermost, then tap, then 32-channel slice — the packed weight layout
//   k = tap * cin_pad + ci is unchanged, only the walk differs.
//   Every wave issues the same number of loads per K-step (weight loads + one patch "slot", filled with a load of
//   the next chunk's patch or a zero-traffic out-of-range load), so the counted vmcnt stays a constant.
template <int BC, int TW, int NTAPS, int KW, int S, int CCH>
__global__ __launch_bounds__(256, (CCH == 32 ? 3 : 2)) void conv_halo_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using T = half_t;
  constexpr int BP = 128, TH = BP / TW, NW = 4;
  constexpr int WGP = 2, WGC = 2;
  constexpr int WT_P = BP / WGP, WT_C = BC / WGC, MT_P = WT_P / 32, MT_C = WT_C / 32;
  constexpr int BKB = 64, A_STAGE = BC * BKB;
  constexpr int NIAW = BC / 64;             // weight wave-loads per wave per K-step
  constexpr int NLS = NIAW + 1;             // + the patch slot
  constexpr int NSLM = CCH / 32;            // 32-channel slices (K-steps per tap) in a regular chunk
  constexpr int ROWB = CCH * 2;             // bytes of one patch pixel row
  constexpr int LPP = ROWB / 16;            // lanes per patch pixel in a 1-KiB wave load
  constexpr int PPW = 64 / LPP;             // patch pixels per wave load
  constexpr int NPWW_MAX = NTAPS * NSLM - (S - 1) < 6 ? NTAPS * NSLM - (S - 1) : 6;   // patch wave-loads per wave per chunk (host checks)
  constexpr int CH = 4, SWZ_DIV = 4, RPI = 16;
  static_assert(BC == 64 || BC == 128, "channel tile");
  static_assert(CCH == 32 || CCH == 64, "channels per resident patch chunk");
  static_assert((NTAPS * NSLM) % S == 0, "a regular chunk must be a whole number of ring turns");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave % WGP, wc = wave / WGP;
  const int l31 = lane & 31, lhi = lane >> 5;

  int ctile, phase, ptile;
  {
    const int total = p.npt * p.nct * p.nph;
    const int b = blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = b & 7, loc = b >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    ctile = logical % p.nct;
    const int t = logical / p.nct;
    phase = t % p.nph;
    ptile = t / p.nph;
  }
  const int py = phase >> 1, px = phase & 1;
  const int co0 = ctile * BC;
  const int tiles_per_img = p.h_ty * p.h_tx;
  const int n = ptile / tiles_per_img;
  const int trem = ptile - n * tiles_per_img;
  const int tyi = trem / p.h_tx, txi = trem - tyi * p.h_tx;
  const int qy0 = tyi * TH, qx0 = txi * TW;
  const int Hq = p.HqWq / p.Wq;
  // input patch origin (top-left input pixel any tap of the patch touches)
  const int iy_org = p.transposed ? qy0 + py - (p.kh - 1) : qy0 - p.pad;
  const int ix_org = p.transposed ? qx0 + px - (p.kw - 1) : qx0 - p.pad_x;
  const int PW = p.h_pw;
  const int npww = p.h_npww;

  char* const ring = smem;
  char* const patch0 = smem + S * A_STAGE;
  const int PB = p.h_pb;
  char* const scratch = patch0 + 2 * PB;                       // 1 KiB: destination of the zero-traffic slot loads
  long long* s_opix = reinterpret_cast<long long*>(scratch + 1024);

  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(p.w) + (size_t)(phase * p.Cout_pad + co0) * p.Kpad * 2, 0, BC * p.Kpad * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;

  // ---- loader constants ---------------------------------------------------------------------------------
  unsigned a_voff[NIAW];
  {
    const int lrow = lane / CH, pos = lane % CH;
#pragma unroll
    for (int t = 0; t < NIAW; ++t) {
      const int r = (wave + NW * t) * RPI + lrow;
      const int lc = pos ^ ((r / SWZ_DIV) % CH);
      a_voff[t] = (unsigned)(r * p.Kpad * 2 + lc * 16);
    }
  }
  // patch swizzle: the 16-byte chunk q of patch pixel r sits at position q ^ pswz(r) of its row (bank-conflict-free
  // fragment reads: 16 consecutive pixels of a tile row hit 16 different bank groups)
  // 128-byte rows (CCH == 64): two rows share a 256-byte bank row, so the key changes every second row — with `r & 7` rows r
  // and r + 8 share a 16-byte slot (PMC: 33-40 % bank-conflict cycles on the <*,16,4,2,4,64> instantiations; the reasoning is
  // bottleneck.hip's BNK_KEY).  MEASURED EFFECT HERE: none — FlowNet2S 15.8-16.0 k pairs/s with this key against 16.0-16.1 k
  // with `r & 7` (same box, interleaved), deconv2..5 within 0.5 us of each other: these launches rarely WAIT on LDS
  // (SQ_WAIT_INST_LDS 4-8 % of wave cycles), so removing conflict cycles does not show.  Kept because it is the consistent key;
  // it is not a speed-up.  64-byte rows (CCH == 32) keep (r >> 2) & 3: their conflicts (40-56 %) come from the jump of PW - TW
  // rows between tile rows inside one 16-lane group, which no per-row key addresses.
  auto pswz = [](int r) { return CCH == 64 ? ((r >> FT_HALO_KEY_SHIFT) & 7) : ((r >> 2) & 3); };
  unsigned p_voff[NPWW_MAX];      // patch pixel rows of ROWB bytes: LPP lanes per pixel
  {
    const int ppl = lane / LPP, pos = lane % LPP;
#pragma unroll
    for (int t = 0; t < NPWW_MAX; ++t) {
      const int pp = (t * NW + wave) * PPW + ppl;
      unsigned v = kOOB;
      if (pp < p.h_npix) {
        const int pr = pp / PW, pc = pp - pr * PW;
        const int iy = iy_org + pr, ix = ix_org + pc;
        if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
          v = (unsigned)((((n * p.Hi + iy) * p.Wi + ix) * p.x_cstride + p.x_coff) * 2 + ((pos ^ pswz(pp)) << 4));
      }
      p_voff[t] = v;
    }
  }
  // pixel-operand fragment rows: tile pixel -> patch row of tap (0,0)
  int r0[MT_P];
#pragma unroll
  for (int j = 0; j < MT_P; ++j) {
    const int m = wp * WT_P + j * 32 + l31;
    r0[j] = (m / TW) * PW + (m % TW);
  }
  int a_off[MT_C];
#pragma unroll
  for (int i = 0; i < MT_C; ++i) {
    const int r = wc * WT_C + i * 32 + l31;
    a_off[i] = r * BKB + ((lhi ^ ((r / SWZ_DIV) % CH)) << 4);
  }

  // output pixel table for the epilogue (and -1 for the ragged part of the patch)
  if (tid < BP) {
    const int oy = qy0 + tid / TW, ox = qx0 + tid % TW;
    long long o = -1;
    if (oy < Hq && ox < p.Wq) o = ((long long)n * p.Ho + (oy * p.omul + py)) * p.Wo + (ox * p.omul + px);
    s_opix[tid] = o;
  }

  float16_t acc[MT_C][MT_P];
#pragma unroll
  for (int i = 0; i < MT_C; ++i)
#pragma unroll
    for (int j = 0; j < MT_P; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int kc = p.kc;                       // 32-channel slices per tap
  const int nfull = kc / NSLM, half = kc % NSLM;   // regular chunks; trailing 32-channel chunk (CCH == 64 only)
  const int cin_b = p.kc * BKB;              // bytes of one tap's K-run in the packed weight row

  auto load_a = [&](auto slot_c, bool live, int soff) {
    constexpr int slot = decltype(slot_c)::value;
#pragma unroll
    for (int t = 0; t < NIAW; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr)(ring + slot * A_STAGE + (wave + NW * t) * 1024), 16,
                                               live ? a_voff[t] : kOOB, live ? soff : 0, 0, 0);
  };
  auto load_patch = [&](auto t_c, bool live, int buf, int soff) {
    constexpr int t = decltype(t_c)::value;
    if constexpr (t < NPWW_MAX) {
      const bool on = live && t < npww;
      char* dst = on ? patch0 + buf * PB + (t * NW + wave) * 1024 : scratch;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr)dst, 16, on ? p_voff[t] : kOOB, on ? soff : 0, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr)scratch, 16, kOOB, 0, 0, 0);
    }
  };

  // ---- prologue: patch of chunk 0, then S-1 weight stages (each preceded by its patch slot) --------------
  static_for<NPWW_MAX>([&](auto tc) {
    if (decltype(tc)::value < npww) load_patch(tc, true, 0, 0);
  });
  const bool one_half_only = nfull == 0;     // cin_pad == 32: the only chunk is the half chunk
  // weight offset of global K-step (chunk c, step st) for chunk kind NSL
  auto a_soff_of = [&](int c, int st, int nsl) {
    const int tap = nsl == 2 ? st >> 1 : st, sl = nsl == 2 ? st & 1 : 0;
    return tap * cin_b + c * ROWB + sl * 64;
  };
  const int total_steps = NTAPS * kc;
  static_for<S - 1>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    load_patch(std::integral_constant<int, NPWW_MAX>{}, false, 0, 0);   // slot (nothing to prefetch yet)
    load_a(sc, s < total_steps, a_soff_of(0, s, one_half_only ? 1 : NSLM));
  });
  __syncthreads();                           // s_opix visible (also orders nothing else: LDS-DMA uses vmcnt)

  // ---- one chunk: NTAPS * NSL K-steps, fully unrolled ----------------------------------------------------
  auto chunk_body = [&](auto nsl_c, int c, bool has_next, int next_nsl) {
    constexpr int NSL = decltype(nsl_c)::value;
    constexpr int SPC = NTAPS * NSL;
    const char* pbuf = patch0 + (c & 1) * PB;
    static_for<SPC>([&](auto st_c) {
      constexpr int st = decltype(st_c)::value;
      constexpr int tap = st / NSL, sl = st % NSL;
      constexpr int slot = st % S, nslot = (st + S - 1) % S;
      if constexpr (st == 0) {
        // first K-step on a freshly prefetched patch buffer: wait for EVERY load of this wave (see conv_stem_kernel:
        // the counted form was not reliable for a resident patch that older loads fill while younger ones stream)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NLS * (S - 2)) : "memory");
      }
      FT_LDS_BARRIER();
      // fragments: weights from the ring, pixels from the resident patch at this tap's offset
      const char* sa = ring + slot * A_STAGE;
      constexpr int ky = tap / KW, kx = tap % KW;
      const int trow = p.transposed ? ((NTAPS / KW - 1 - ky) * PW + (KW - 1 - kx)) : (ky * PW + kx);
      uint4_t fa[2][MT_C], fb[2][MT_P];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < MT_C; ++i) fa[kk][i] = *reinterpret_cast<const uint4_t*>(sa + (a_off[i] ^ (kk << 5)));
#pragma unroll
      for (int j = 0; j < MT_P; ++j) {
        const int r = r0[j] + trow;
        const int lsw = lhi ^ pswz(r);
        const char* rowp = pbuf + r * ROWB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          fb[kk][j] = *reinterpret_cast<const uint4_t*>(rowp + ((lsw ^ (sl * 4 + kk * 2)) << 4));
      }
      // loads of K-step (st + S - 1): patch slot first, then the weight stage
      constexpr int la = st + S - 1;                     // lookahead step, may fall into the next chunk
      constexpr int NM = 2 * MT_C * MT_P;
      static_for<NM>([&](auto mi) {
        constexpr int m = decltype(mi)::value;
        constexpr int kk = m / (MT_C * MT_P), i = (m / MT_P) % MT_C, j = m % MT_P;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, fa[kk][i]),
                                                           __builtin_bit_cast(half8_t, fb[kk][j]), acc[i][j], 0, 0, 0);
        if constexpr (m == 0) {
          load_patch(std::integral_constant<int, (st < NPWW_MAX ? st : NPWW_MAX)>{}, has_next, (c + 1) & 1, (c + 1) * ROWB);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (m == 1 || (NM == 2 && m == 1)) {
          if constexpr (la < SPC) {
            load_a(std::integral_constant<int, nslot>{}, true, a_soff_of(c, la, NSL));
          } else {
            load_a(std::integral_constant<int, nslot>{}, has_next, a_soff_of(c + 1, la - SPC, next_nsl));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      });
    });
  };
  using regular = std::integral_constant<int, NSLM>;
  using one = std::integral_constant<int, 1>;
  for (int c = 0; c < nfull; ++c) {
    const bool last_full = c + 1 == nfull;
    chunk_body(regular{}, c, !last_full || half, last_full && half ? 1 : NSLM);
  }
  if constexpr (NSLM == 2) {
    if (half) chunk_body(one{}, nfull, false, 1);
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  conv_epilogue<T, BP, BC, WGP, WGC, true>(p, acc, smem, (int)(reinterpret_cast<char*>(s_opix) - smem), 0, co0, py, px, nullptr);
#endif
}

// ---- stem path (fp16): row-packed small-Cin convs (7x7/s2 on 3/6/12 channels, 3x3/s1 on 6/11) -> 64 channels ----
// The input (physically x-padded NHWC, 4/8/16 channels per pixel) is so thin that an 8x16 output patch needs only
// 6-26 KiB of it: the patch is loaded ONCE into LDS and every kernel row's pixel operand is a contiguous 16-byte read
// from it (the taps of one kernel row ARE contiguous in a row-packed row).  Only the weight rows stream (one K-step
// per kernel row).  The layer becomes HBM-bound (read the frame once, write the 64-channel map once) instead of
// re-reading each input row kh times through L2 -> LDS.
#ifndef FT_STEM_WGS
#define FT_STEM_WGS 2   // workgroups per CU the register budget is set for (dev A/B)
#endif
template <int KH, int STRIDE, int RUNB, int NT>
__global__ __launch_bounds__(256, FT_STEM_WGS) void conv_stem_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using T = half_t;
  constexpr int BP = 128, BC = 64, TW = 16, TH = 8, NW = 4, WGP = 2, WGC = 2;
  constexpr int WT_P = BP / WGP, MT_P = WT_P / 32, MT_C = 1;
  constexpr int S = 3, A_STAGE = BC * RUNB;
  constexpr int CH = RUNB / 16, SWZ_DIV = 256 / RUNB >= 1 ? 256 / RUNB : 1, RPI = 64 / CH;
  constexpr int NIA = BC / RPI / NW;          // weight wave-loads per wave per kernel row
  constexpr int G = RUNB / 32;                // MFMA k-groups (16 halfs) per kernel row
  constexpr int PH = (TH - 1) * STRIDE + KH;  // input rows of the patch
  constexpr int NPLW_MAX = 12;
  constexpr int TWS = TW * NT;                // output columns of the super-tile: NT tiles of 16 share one pass over the weights
  static_assert(RUNB == 64 || RUNB == 128 || RUNB == 256, "bytes of one packed kernel row");
  static_assert(NIA >= 1, "weight rows must split over the waves");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave % WGP, wc = wave / WGP;
  const int l31 = lane & 31, lhi = lane >> 5;

  int ptile;
  {
    const int total = p.npt;
    const int b = blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = b & 7, loc = b >> 3;
    ptile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tiles_per_img = p.h_ty * p.h_tx;
  const int n = ptile / tiles_per_img;
  const int trem = ptile - n * tiles_per_img;
  const int tyi = trem / p.h_tx, txi = trem - tyi * p.h_tx;
  const int oy0 = tyi * TH, ox0 = txi * TWS;
  const int cpb = p.x_cstride * 2;                       // bytes per input pixel
  const int CPR = p.h_pw;                                // 16-byte chunks per patch row
  const int RBp = CPR * 16;                              // patch row pitch in LDS
  const int iy_org = oy0 * STRIDE - p.pad;
  const int col0 = ox0 * STRIDE - p.pad_x;               // physical column of the patch's first pixel (>= 0)

  char* const ring = smem;
  char* const patch = smem + S * A_STAGE;
  long long* s_opix = reinterpret_cast<long long*>(patch + p.h_pb);

  const __amdgpu_buffer_rsrc_t rsrc_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.w), 0, BC * p.Kpad * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;

  // ---- the whole input patch: PH rows of CPR 16-byte chunks, lane-linear (row-major) in LDS ---------------
  const int nchunks = PH * CPR;
#pragma unroll
  for (int t = 0; t < NPLW_MAX; ++t) {
    if (t < p.h_npww) {
      const int gci = (t * NW + wave) * 64 + lane;
      const int row = gci / CPR, ch = gci - row * CPR;
      const int iy = iy_org + row;
      unsigned v = kOOB;
      if (gci < nchunks && (unsigned)iy < (unsigned)p.Hi)
        v = (unsigned)(((n * p.Hi + iy) * p.Wi + col0) * cpb + ch * 16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr)(patch + (t * NW + wave) * 1024), 16, v, 0, 0, 0);
    }
  }
  // ---- weight rows: [64 co][RUNB] tiles, XOR-swizzled on the source side as in the dma kernel ----------------
  unsigned a_voff[NIA];
  {
    const int lrow = lane / CH, pos = lane % CH;
#pragma unroll
    for (int t = 0; t < NIA; ++t) {
      const int r = (wave + NW * t) * RPI + lrow;
      const int lc = pos ^ ((r / SWZ_DIV) % CH);
      a_voff[t] = (unsigned)(r * p.Kpad * 2 + lc * 16);
    }
  }
  auto load_a = [&](auto slot_c, bool live, int ky) {
    constexpr int slot = decltype(slot_c)::value;
#pragma unroll
    for (int t = 0; t < NIA; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr)(ring + slot * A_STAGE + (wave + NW * t) * 1024), 16,
                                               live ? a_voff[t] : kOOB, live ? ky * RUNB : 0, 0, 0);
  };
  static_for<S - 1>([&](auto sc) { load_a(sc, decltype(sc)::value < KH, decltype(sc)::value); });

  // fragment offsets
  const int r_a = wc * 32 + l31;
  const int a_off = r_a * RUNB + ((lhi ^ ((r_a / SWZ_DIV) % CH)) << 4);
  int b_off[MT_P];              // tile 0; tile t adds t * 16 * STRIDE * cpb
#pragma unroll
  for (int j = 0; j < MT_P; ++j) {
    const int m = wp * WT_P + j * 32 + l31;
    b_off[j] = (m / TW) * STRIDE * RBp + (m % TW) * STRIDE * cpb + lhi * 16;
  }
  const int tile_step = TW * STRIDE * cpb;
  for (int idx = tid; idx < BP * NT; idx += 256) {
    const int t = idx / BP, m = idx % BP;
    const int oy = oy0 + m / TW, ox = ox0 + t * TW + m % TW;
    long long o = -1;
    if (oy < p.Ho && ox < p.Wo) o = ((long long)n * p.Ho + oy) * p.Wo + ox;
    s_opix[idx] = o;
  }

  float16_t acc[NT][MT_C][MT_P];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < MT_P; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][0][j][r] = 0.f;

  static_for<KH>([&](auto ky_c) {
    constexpr int ky = decltype(ky_c)::value;
    constexpr int slot = ky % S, nslot = (ky + S - 1) % S;
    // FULL wait (every load of this wave has landed, every ds_read returned), not the counted vmcnt(NIA * (S - 2)) of
    // the other kernels: with a counted wait ~0.1 % of the tiles came out wrong once workgroups were recycled on a CU.
    // Probes (full wait at step 0 only / at the later steps only / no dummy tail loads) all still failed and the
    // ordering micro-benchmarks under tools/dev/ubench pass, so the root cause is not pinned down; this form is the
    // most conservative DMA -> LDS -> ds_read sync there is and is what tests/ + tools/dev/determinism_stress.py
    // hold green.  Only 7 K-steps per workgroup: the lost weight prefetch depth is noise.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    FT_LDS_BARRIER();
    load_a(std::integral_constant<int, nslot>{}, ky + S - 1 < KH, ky + S - 1);
    const char* sa = ring + slot * A_STAGE;
    const char* pb = patch + ky * RBp;
    constexpr int GB = G > 2 ? 2 : G;       // k-groups per register batch
    static_for<G / GB>([&](auto gb_c) {
      constexpr int g0 = decltype(gb_c)::value * GB;
      uint4_t fa[GB];
#pragma unroll
      for (int g = 0; g < GB; ++g) fa[g] = *reinterpret_cast<const uint4_t*>(sa + (a_off ^ ((g0 + g) << 5)));
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        uint4_t fb[GB][MT_P];
#pragma unroll
        for (int g = 0; g < GB; ++g)
#pragma unroll
          for (int j = 0; j < MT_P; ++j)
            fb[g][j] = *reinterpret_cast<const uint4_t*>(pb + b_off[j] + t * tile_step + (g0 + g) * 32);
#pragma unroll
        for (int g = 0; g < GB; ++g)
#pragma unroll
          for (int j = 0; j < MT_P; ++j)
            acc[t][0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, fa[g]),
                                                                  __builtin_bit_cast(half8_t, fb[g][j]), acc[t][0][j], 0, 0, 0);
      }
    });
  });

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int opix_off = (int)(reinterpret_cast<char*>(s_opix) - smem);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    conv_epilogue<T, BP, BC, WGP, WGC, true>(p, acc[t], smem, opix_off + t * BP * 8, 0, 0, 0, 0, nullptr);
    if (t + 1 < NT) __syncthreads();          // the next tile reuses the LDS output tile
  }
#endif
}

// ---- persistent, weight-stationary form of the stem (round 3) -------------------------------------------------------
// conv_stem_kernel re-streams the layer's whole weight set (KH x 64 x RUNB bytes: 56 KiB for FlowNet's 7x7 on a frame pair)
// for every 128-pixel tile and pays a full vector-memory wait + barrier per kernel row: 79-91 us on FlowNet2S's conv1 for
// 150 MB of HBM traffic and 18 us of MFMA.  Here one workgroup per CU keeps ALL kernel rows' weights in LDS for its
// lifetime and walks a contiguous range of 8 x 16 output tiles: the next tile's input patch is DMA'd into the other patch
// buffer while the current one is multiplied (no wait, no barrier inside the 7-row K walk), the output tile leaves through
// its own LDS transposition buffer with a FIXED number of buffer stores per lane (out-of-image pixels are out-of-range
// stores), so the single counted wait per tile — `vmcnt(stores of the previous tile)` — is exact: vector-memory operations
// complete in issue order on this counter.
template <int KH, int STRIDE, int RUNB>
__global__ __launch_bounds__(512, 1) void conv_stem_persist_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BP = 128, BC = 64, TW = 16, TH = 8, NW = 4, WGP = 2;
  constexpr int WT_P = BP / WGP, MT_P = WT_P / 32;
  constexpr int A_STAGE = BC * RUNB;
  constexpr int CH = RUNB / 16, SWZ_DIV = 256 / RUNB >= 1 ? 256 / RUNB : 1, RPI = 64 / CH;
  constexpr int NIA = BC / RPI / NW;          // weight wave-loads per wave per kernel row
  constexpr int G = RUNB / 32;                // MFMA k-groups (16 halfs) per kernel row
  constexpr int PH = (TH - 1) * STRIDE + KH;  // input rows of the patch
  constexpr int NPLW_MAX = 12;
  constexpr int WTS = KH * A_STAGE;           // [0, WTS): the weights; then two patch buffers of p.h_pb bytes; then the output tile
  constexpr int NCHO = BC / 8, OROWB = BC * 2;
  constexpr int NST = BP * NCHO / 256;        // 16-byte stores per lane per tile (4)
  static_assert(RUNB == 64 || RUNB == 128 || RUNB == 256, "bytes of one packed kernel row");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  // eight waves = two independent quartets (`sub`), each walking its own tiles with its own patch buffers and output tile and
  // sharing the weights: two waves per SIMD hide each other's LDS latency (one quartet per CU ran at 4.2 us per tile)
  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = wave8 >> 2, wave = wave8 & 3;
  const int wp = wave % WGP, wc = wave / WGP;
  const int l31 = lane & 31, lhi = lane >> 5;

  // contiguous tile range of this workgroup, XCD by XCD (neighbouring tiles share halo rows / columns in that XCD's L2)
  int t_lo, t_hi;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, loc = b >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int per = p.npt / nwg, rem = p.npt - per * nwg;
    t_lo = logical * per + (logical < rem ? logical : rem);
    t_hi = t_lo + per + (logical < rem ? 1 : 0);
  }
  const int tiles_per_img = p.h_ty * p.h_tx;
  const int cpb = p.x_cstride * 2;                       // bytes per input pixel
  const int CPR = p.h_pw;                                // 16-byte chunks per patch row
  const int RBp = CPR * 16;
  const int nchunks = PH * CPR;
  // Stride 2 on 16-byte pixels: a fragment read takes every other chunk of a patch row, and the sixteen lanes one
  // ds_read_b128 cycle serves ({0-3, 12-15, 20-27} ...) sit on two output rows = patch rows pr and pr + 2, both on the same
  // chunk parity: 2-way conflicts on every pixel-operand read.  Chunk c of patch row pr therefore lives at c ^ ((pr >> 1) & 1)
  // (applied on the SOURCE side of the DMA; the host keeps CPR even): the two rows of a cycle take opposite parities.  (The
  // same swizzle changed nothing in conv_stem_kernel, which is not LDS-bound; this form is.)
  const int swz = (STRIDE == 2 && cpb == 16) ? 1 : 0;
  char* const patch0 = smem + WTS + sub * (2 * p.h_pb + BP * OROWB);
  char* const otile = patch0 + 2 * p.h_pb;

  const __amdgpu_buffer_rsrc_t rsrc_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.w), 0, BC * p.Kpad * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;

  // ---- all kernel rows' weights, once: [ky][64 co][RUNB] tiles, XOR-swizzled on the source side as in conv_stem_kernel ----
  if (sub == 0) {
    const int lrow = lane / CH, pos = lane % CH;
#pragma unroll
    for (int t = 0; t < NIA; ++t) {
      const int r = (wave + NW * t) * RPI + lrow;
      const int lc = pos ^ ((r / SWZ_DIV) % CH);
      const unsigned voff = (unsigned)(r * p.Kpad * 2 + lc * 16);
#pragma unroll
      for (int ky = 0; ky < KH; ++ky)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr)(smem + ky * A_STAGE + (wave + NW * t) * 1024), 16, voff, ky * RUNB, 0, 0);
    }
  }
  // the input patch of tile t -> patch buffer `buf`: PH rows of CPR 16-byte chunks, lane-linear (always NPLW wave-loads).
  // A lane's (row, chunk) inside the patch does not depend on the tile: the divisions are done once, here.
  int p_row[NPLW_MAX], p_off[NPLW_MAX];
#pragma unroll
  for (int u = 0; u < NPLW_MAX; ++u) {
    const int gci = (u * NW + wave) * 64 + lane;
    const int row = gci / CPR, ch = (gci - row * CPR) ^ (swz & (row >> 1));
    p_row[u] = (u < p.h_npww && gci < nchunks) ? row : 0x40000000;      // never inside the image
    p_off[u] = row * p.Wi * cpb + ch * 16;
  }
  auto load_patch = [&](int t, int buf) {
    const bool live = t < t_hi && !(p.dbg & 2);      // FT_CONV_DBG (dev ablation): 2 = no patch loads, 1 = no K walk, 4 = no stores
    const int n = t / tiles_per_img, trem = t - n * tiles_per_img;
    const int tyi = trem / p.h_tx, txi = trem - tyi * p.h_tx;
    const int iy_org = tyi * TH * STRIDE - p.pad, col0 = txi * TW * STRIDE - p.pad_x;
    const int base = ((n * p.Hi + iy_org) * p.Wi + col0) * cpb;
    char* dst = patch0 + buf * p.h_pb;
#pragma unroll
    for (int u = 0; u < NPLW_MAX; ++u) {
      if (u < p.h_npww) {
        const unsigned v = (live && (unsigned)(iy_org + p_row[u]) < (unsigned)p.Hi) ? (unsigned)(base + p_off[u]) : kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr)(dst + (u * NW + wave) * 1024), 16, v, 0, 0, 0);
      }
    }
  };
  load_patch(t_lo + sub, 0);

  // fragment offsets (tile-independent)
  const int r_a = wc * 32 + l31;
  const int a_off = r_a * RUNB + ((lhi ^ ((r_a / SWZ_DIV) % CH)) << 4);
  int b_off[2][MT_P];           // [kpar]: kernel rows with (ky >> 1) & 1 == kpar
#pragma unroll
  for (int j = 0; j < MT_P; ++j) {
    const int m = wp * WT_P + j * 32 + l31;
    const int base = (m / TW) * STRIDE * RBp + (m % TW) * STRIDE * cpb;
    const int key = swz & (m / TW);          // patch row = (m / TW) * 2 + ky: its chunk key is ((m / TW) + (ky >> 1)) & 1
    b_off[0][j] = base + ((lhi ^ key) << 4);
    b_off[1][j] = base + ((lhi ^ key ^ swz) << 4);
  }
  // folded BN / bias of this lane's channels: register r of lane (pixel, half) is channel 8 * (r / 4) + 4 * half + r % 4 of the wave's 32
  float4_t sc[4], sh[4];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const int co = wc * 32 + g4 * 8 + lhi * 4;
    sc[g4] = p.scale ? *reinterpret_cast<const float4_t*>(p.scale + co) : float4_t{1.f, 1.f, 1.f, 1.f};
    sh[g4] = p.shift ? *reinterpret_cast<const float4_t*>(p.shift + co) : float4_t{0.f, 0.f, 0.f, 0.f};
  }
  // act(v) = act_s * min(v, 0) + max(v, 0) with act_s = 0 (ReLU), slope (LeakyReLU) or 1 (none): the same values as the
  // branchy form, three instructions, no scalar branch per element (apply_act compiled to 64 s_cbranch per tile here)
  const float act_s = p.act == FT_ACT_RELU ? 0.f : (p.act == FT_ACT_LEAKY ? p.slope : 1.f);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // weights + first patch (this wave's pieces), the table loads

  const int nit = (t_hi - t_lo + 1) >> 1;     // both quartets run the same number of rounds (the last tile of an odd range is a dummy)
  // (Running the quartets half a round apart — one multiplies while the other transposes and stores — was measured: 100 vs 73
  // us.  In lock-step the two waves of a SIMD hide each other's LDS latency during the K walk; half a round apart each
  // quartet is alone on the matrix pipe again.)
  for (int it = 0; it < nit; ++it) {
    const int t = t_lo + 2 * it + sub;
    const int buf = it & 1;
    // patch(t) was issued before the previous tile's NST stores: it has landed once at most NST operations are outstanding
    if (it > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
    FT_LDS_BARRIER();            // everyone's pieces of patch(t); every read of the other patch buffer and of the output tile is done
    if (p.shift_n) {             // per-sample shift (FlowNet2S's rgb mean folded into conv1, ft_flow_mean_fold): issued AHEAD of the
                                 // look-ahead patch loads, so the wait in front of the epilogue leaves those in flight
      const int ns = t < t_hi ? t / tiles_per_img : 0;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) sh[g4] = *reinterpret_cast<const float4_t*>(p.shift + (size_t)ns * p.shift_n + wc * 32 + g4 * 8 + lhi * 4);
    }
    load_patch(t + 2, buf ^ 1);

    float16_t acc[MT_P];
#pragma unroll
    for (int j = 0; j < MT_P; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const char* pbase = patch0 + buf * p.h_pb;
    if (!(p.dbg & 1)) static_for<KH>([&](auto ky_c) {
      constexpr int ky = decltype(ky_c)::value;
      const char* sa = smem + ky * A_STAGE;
      const char* pb = pbase + ky * RBp;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const uint4_t fa = *reinterpret_cast<const uint4_t*>(sa + (a_off ^ (g << 5)));
        uint4_t fb[MT_P];
#pragma unroll
        for (int j = 0; j < MT_P; ++j) fb[j] = *reinterpret_cast<const uint4_t*>(pb + b_off[(ky >> 1) & 1][j] + g * 32);
#pragma unroll
        for (int j = 0; j < MT_P; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, fa), __builtin_bit_cast(half8_t, fb[j]), acc[j], 0, 0, 0);
      }
    });

    // ---- bn / bias + activation -> fp16 output tile [128 px][64 co] (128-byte rows, 16-byte chunk ^= row & 7) ----------
    {
#pragma clang fp contract(off)   // scale, then shift, each rounded — as conv_epilogue does
#pragma unroll
      for (int j = 0; j < MT_P; ++j) {
        const int pl = wp * WT_P + j * 32 + l31;
        char* rowp = otile + pl * OROWB + lhi * 8;
        const int msw = (pl & 7) << 4;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          half4_t hv;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = acc[j][g4 * 4 + e] * sc[g4][e] + sh[g4][e];
            hv[e] = (half_t)__builtin_fmaf(act_s, __builtin_fminf(v, 0.f), __builtin_fmaxf(v, 0.f));
          }
          *reinterpret_cast<half4_t*>(rowp + (((wc * 4 + g4) << 4) ^ msw)) = hv;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    FT_LDS_BARRIER();
    {
      const int n = t / tiles_per_img, trem = t - n * tiles_per_img;
      const int tyi = trem / p.h_tx, txi = trem - tyi * p.h_tx;
      const int oy0 = tyi * TH, ox0 = txi * TW;
#pragma unroll
      for (int k = 0; k < NST; ++k) {
        const int idx = tid + k * 256, pl = idx / NCHO, ch = idx % NCHO;
        const int oy = oy0 + pl / TW, ox = ox0 + pl % TW;
        const uint4_t v = *reinterpret_cast<const uint4_t*>(otile + pl * OROWB + ((ch ^ (pl & 7)) << 4));
        const unsigned vo = (t < t_hi && oy < p.Ho && ox < p.Wo && !(p.dbg & 4)) ? (unsigned)((((n * p.Ho + oy) * p.Wo + ox) * p.y_cstride + p.y_coff + ch * 8) * 2) : kOOB;
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_y, vo, 0, FT_YSTORE_BUF_AUX);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the look-ahead patch loads must not outlive the workgroup's LDS
#endif
}

// ---- stem + max-pool (fp16): 7x7/s2 row-packed stem conv + bn + relu with the 3x3/s2/p1 max-pool behind it -----
// (resnet.py:19-23 = conv1 -> bn1 -> relu -> maxpool).  The workgroup owns an 8x8 patch of POOLED pixels: it computes the
// 17x17 stem outputs those windows cover (1.13x recompute of a cheap layer) with the stem kernel's K-loop, parks them as
// fp16 in LDS, and writes only the pooled maxima: the [N, H/2, W/2, 64] stem map — the largest activation of the trunk,
// 100 MB at batch 64 — is never written or re-read.  Pooling happens on the same fp16 values the separate launches
// would pool, so the result is bit-identical to conv_stem_kernel + maxpool3x3s2.
// Round 6, tried and left OFF (FT_STEM_POOL_ALLW=1): all seven kernel rows' weights (28 KiB) DMA'd into LDS in the prologue and the K
// walk without a wait or a barrier (the three-slot ring puts a full vmcnt(0) + s_barrier in front of every ten MFMAs of a wave).
// Bit-identical, and SLOWER: 54.3 vs 50.3 us in the network (three interleaved runs each, one box) — a workgroup then waits for 28 KiB of
// weights before its first MFMA instead of 8, and the barriers were not what the patch gather leaves exposed.
#ifndef FT_STEM_POOL_ALLW
#define FT_STEM_POOL_ALLW 0
#endif
#ifndef FT_STEM_ABL
#define FT_STEM_ABL 0     // dev ablations (timing only, wrong results): 1 = no planar gather loads, 2 = no output stores, 4 = no weight loads, 8 = no MFMAs
#endif
// Round 6: the kernel is bound by instruction ISSUE, not by a memory or the matrix pipe (tools/dev/isa_phases.py: 70 MFMAs against
// 1461 vector + 506 scalar instructions per wave, four workgroups per CU: 33 per MFMA; ~1200 of them were the address arithmetic of
// the planar gather in the prologue).  CPRC > 0 = the 16-byte chunks per patch row as a COMPILE-TIME constant (the pose stem: 20)
// and a gather mapping without divisions: thread -> (chunk tid % CPRC, row tid / CPRC + (256 / CPRC) t): the column tests and the
// base offset are computed once, a round adds one constant and tests its row; the epilogue multiplies and adds in pairs
// (v_pk_mul_f32 / v_pk_add_f32: the same two roundings as the scalar pair), ReLU and the 3x3 pool are INTEGER maxima on the fp16
// bit patterns (v_pk_max_i16: exact for the non-negative values behind a ReLU, -0 and negatives order below +0, and no
// canonicalising v_pk_max in front of every maximum as with the fp16 form).  Bit-identical to the forms it replaces.
template <int RUNB, int CPRC = 0>
__global__ __launch_bounds__(256, 2) void conv_stem_pool_kernel(const ConvParams p, int Hp, int Wp) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int KH = 7, STRIDE = 2, BC = 64, NW = 4, WGP = 2;
  constexpr int PT = 8, TS = 2 * PT + 1, NPX = TS * TS;      // pooled tile edge, stem patch edge (17), stem pixels (289)
  constexpr int MT_P = 5, WT_P = MT_P * 32;                  // 2 x 160 >= 289 pixels
  constexpr int S = FT_STEM_POOL_ALLW ? KH : 3, A_STAGE = BC * RUNB;
  constexpr int CH = RUNB / 16, SWZ_DIV = 256 / RUNB >= 1 ? 256 / RUNB : 1, RPI = 64 / CH;
  constexpr int NIA = BC / RPI / NW;
  constexpr int G = RUNB / 32;
  constexpr int PH = (TS - 1) * STRIDE + KH;                 // 39 input rows
  constexpr int NPLW_MAX = 12;
  static_assert(RUNB == 64 || RUNB == 128, "bytes of one packed kernel row");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave % WGP, wc = wave / WGP;
  const int l31 = lane & 31, lhi = lane >> 5;

  int ptile;
  {
    const int total = p.npt;
    const int b = blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = b & 7, loc = b >> 3;
    ptile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tiles_per_img = p.h_ty * p.h_tx;
  const int n = ptile / tiles_per_img;
  const int trem = ptile - n * tiles_per_img;
  const int tyi = trem / p.h_tx, txi = trem - tyi * p.h_tx;
  const int py0 = tyi * PT, px0 = txi * PT;              // pooled origin
  const int oy0 = 2 * py0 - 1, ox0 = 2 * px0 - 1;        // stem-output origin (row / column -1 = the pool's padding)
  const int cpb = p.x_cstride * 2;
  const int CPR = CPRC > 0 ? CPRC : p.h_pw;
  const int RBp = CPR * 16;
  const int iy_org = oy0 * STRIDE - p.pad;
  const int col0 = ox0 * STRIDE - p.pad_x;               // >= 0: the host requires x_lpad >= pad + 2

  char* const ring = smem;
  char* const patch = smem + S * A_STAGE;

  const __amdgpu_buffer_rsrc_t rsrc_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.w), 0, BC * p.Kpad * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;

  const int nchunks = PH * CPR;
  constexpr int NR = 5;                        // x_planar: rounds of 256 chunks: 39 rows x <= 32 chunks (the pose stem: 39 x 20 = 780)
  float v[NR][2][3];
  if (p.x_planar) {
    // ft_conv_desc.x_nchw_f32: the patch straight from the network's NCHW fp32 input.  Chunk gci = two pixels (8 bytes each: up to
    // 4 channels as fp16) of patch row gci / CPR, at the LDS byte the packed view's chunk would have been DMA'd to; a lane reads
    // its pixels' planes (lanes = consecutive pixel pairs: 160 contiguous bytes per plane and patch row), casts as
    // pack_nchw_rows4_kernel does and writes 16 bytes.  Loads of all rounds first; the casts and LDS writes follow the weight prologue below.
    // (buffer loads with out-of-range offsets for everything outside the image: no branch per load — as predicated global loads
    // every one of the 30 sat in its own divergent region behind a full wait: 66 us instead of 42.5 + 18.4 for pack + stem)
    const __amdgpu_buffer_rsrc_t rsrc_f = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
    const int HWb = p.Hi * p.x_w * 4;
    if constexpr (CPRC > 0) {
      constexpr int RPR = 256 / CPRC;                     // patch rows per round
      static_assert((PH + RPR - 1) / RPR <= NR, "rounds");
      const int r0 = tid / CPRC, ch = tid - r0 * CPRC;
      const int ix0 = col0 + 2 * ch - p.x_lpad;
      const bool lane_on = r0 < RPR;
      const bool in0 = lane_on && (unsigned)ix0 < (unsigned)p.x_w, in1 = lane_on && (unsigned)(ix0 + 1) < (unsigned)p.x_w;
      const int rstep = RPR * p.x_w * 4;
      unsigned rbase = (unsigned)(((n * 3 * p.Hi + iy_org + r0) * p.x_w + ix0) * 4);
#pragma unroll
      for (int t = 0; t < NR; ++t) {
        if (t * RPR < PH) {
          const int row = r0 + t * RPR;
          const bool row_in = row < PH && (unsigned)(iy_org + row) < (unsigned)p.Hi;
          const unsigned vo0 = (row_in && in0 && !(FT_STEM_ABL & 1)) ? rbase : kOOB, vo1 = (row_in && in1 && !(FT_STEM_ABL & 1)) ? rbase + 4u : kOOB;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            v[t][0][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_f, vo0, c * HWb, 0));
            v[t][1][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_f, vo1, c * HWb, 0));
          }
          rbase += (unsigned)rstep;
        }
      }
    } else {
#pragma unroll
    for (int t = 0; t < NR; ++t) {
      const int gci = (t * NW + wave) * 64 + lane;
      const int row = gci / CPR, ch = gci - row * CPR;
      const int iy = iy_org + row, ix0 = col0 + 2 * ch - p.x_lpad;
      const bool row_in = t < p.h_npww && gci < nchunks && (unsigned)iy < (unsigned)p.Hi;
      const unsigned rbase = (unsigned)(((n * p.x_c * p.Hi + iy) * p.x_w + ix0) * 4);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const unsigned vo = (row_in && (unsigned)(ix0 + e) < (unsigned)p.x_w) ? rbase + 4u * e : kOOB;
#pragma unroll
        for (int c = 0; c < 3; ++c)
          v[t][e][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_f, c < p.x_c ? vo : kOOB, c * HWb, 0));
      }
    }
    }
  } else {
#pragma unroll
  for (int t = 0; t < NPLW_MAX; ++t) {
    if (t < p.h_npww) {
      const int gci = (t * NW + wave) * 64 + lane;
      const int row = gci / CPR, ch = gci - row * CPR;
      const int iy = iy_org + row;
      unsigned v = kOOB;
      if (gci < nchunks && (unsigned)iy < (unsigned)p.Hi)
        v = (unsigned)(((n * p.Hi + iy) * p.Wi + col0) * cpb + ch * 16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr)(patch + (t * NW + wave) * 1024), 16, v, 0, 0, 0);
    }
  }
  }
  unsigned a_voff[NIA];
  {
    const int lrow = lane / CH, pos = lane % CH;
#pragma unroll
    for (int t = 0; t < NIA; ++t) {
      const int r = (wave + NW * t) * RPI + lrow;
      const int lc = pos ^ ((r / SWZ_DIV) % CH);
      a_voff[t] = (unsigned)(r * p.Kpad * 2 + lc * 16);
    }
  }
  auto load_a = [&](auto slot_c, bool live, int ky) {
    constexpr int slot = decltype(slot_c)::value;
#pragma unroll
    for (int t = 0; t < NIA; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr)(ring + slot * A_STAGE + (wave + NW * t) * 1024), 16,
                                               (live && !(FT_STEM_ABL & 4)) ? a_voff[t] : kOOB, live ? ky * RUNB : 0, 0, 0);
  };
  static_for<FT_STEM_POOL_ALLW ? S : S - 1>([&](auto sc) { load_a(sc, decltype(sc)::value < KH, decltype(sc)::value); });
  if (p.x_planar) {                            // the gathered pixels land in the patch while the first weight rows are on their way
#pragma unroll
    for (int t = 0; t < NR; ++t) {
      const half8_t h8 = {(half_t)v[t][0][0], (half_t)v[t][0][1], (half_t)v[t][0][2], (half_t)0.f,
                          (half_t)v[t][1][0], (half_t)v[t][1][1], (half_t)v[t][1][2], (half_t)0.f};
      if constexpr (CPRC > 0) {
        constexpr int RPR = 256 / CPRC;
        if (t * RPR < PH && tid < RPR * CPRC && tid / CPRC + t * RPR < PH) *reinterpret_cast<half8_t*>(patch + (tid + t * RPR * CPRC) * 16) = h8;
      } else if (t < p.h_npww) {
        *reinterpret_cast<half8_t*>(patch + ((t * NW + wave) * 64 + lane) * 16) = h8;
      }
    }
  }

  const int r_a = wc * 32 + l31;
  const int a_off = r_a * RUNB + ((lhi ^ ((r_a / SWZ_DIV) % CH)) << 4);
  int b_off[MT_P];
#pragma unroll
  for (int j = 0; j < MT_P; ++j) {
    int m = wp * WT_P + j * 32 + l31;
    m = m < NPX ? m : NPX - 1;                           // the 31 surplus pixels recompute the last one (never stored)
    b_off[j] = (m / TS) * STRIDE * RBp + (m % TS) * STRIDE * cpb + lhi * 16;
  }

  float16_t acc[MT_P];
#pragma unroll
  for (int j = 0; j < MT_P; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  static_for<KH>([&](auto ky_c) {
    constexpr int ky = decltype(ky_c)::value;
    constexpr int slot = ky % S, nslot = (ky + S - 1) % S;
    if constexpr (!FT_STEM_POOL_ALLW || ky == 0) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // full wait: see conv_stem_kernel
      FT_LDS_BARRIER();
    }
    if constexpr (!FT_STEM_POOL_ALLW) load_a(std::integral_constant<int, nslot>{}, ky + S - 1 < KH, ky + S - 1);
    const char* sa = ring + slot * A_STAGE;
    const char* pb = patch + ky * RBp;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const uint4_t fa = *reinterpret_cast<const uint4_t*>(sa + (a_off ^ (g << 5)));
      uint4_t fb[MT_P];
#pragma unroll
      for (int j = 0; j < MT_P; ++j) fb[j] = *reinterpret_cast<const uint4_t*>(pb + b_off[j] + g * 32);
#pragma unroll
      for (int j = 0; j < MT_P; ++j) {
#if FT_STEM_ABL & 8
        acc[j][0] += __builtin_bit_cast(float, fa.x ^ fb[j].x);
        continue;
#endif
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, fa), __builtin_bit_cast(half8_t, fb[j]), acc[j], 0, 0, 0);
      }
    }
  });

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  FT_LDS_BARRIER();                                      // ring + patch are dead: the stem tile takes their place
  // ---- bn + relu -> fp16 stem tile [320 px][64 co] in LDS (128-byte rows, 16-byte chunk ^= row & 7) -----------
  char* const stage = smem;
  {
#pragma clang fp contract(off)   // scale, then shift, each rounded — as conv_epilogue does: the result must match it bit for bit
    float4_t sc[4], sh[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = wc * 32 + g * 8 + lhi * 4;
      sc[g] = p.scale ? *reinterpret_cast<const float4_t*>(p.scale + co) : float4_t{1.f, 1.f, 1.f, 1.f};
      sh[g] = p.shift ? *reinterpret_cast<const float4_t*>(p.shift + co) : float4_t{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < MT_P; ++j) {
      const int m = wp * WT_P + j * 32 + l31;
      const int oy = oy0 + m / TS, ox = ox0 + m % TS;
      // relu output >= 0, every window holds a real pixel: 0 is the identity for the pool's padding / ragged edge
      const bool inside = m < NPX && (unsigned)oy < (unsigned)p.Ho && (unsigned)ox < (unsigned)p.Wo;
      char* rowp = stage + m * 128 + lhi * 8;
      const int msw = (m & 7) << 4;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typedef short short2_t __attribute__((ext_vector_type(2)));
        uint2 hb;
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const float2_t prod = float2_t{acc[j][g * 4 + e], acc[j][g * 4 + e + 1]} * float2_t{sc[g][e], sc[g][e + 1]};   // rounded
          const float2_t sum = prod + float2_t{sh[g][e], sh[g][e + 1]};                                                      // rounded again
          const half2_t hh = __builtin_convertvector(sum, half2_t);
          // ReLU on the fp16 bit pattern: as 16-bit integers every negative half (and -0) is below +0
          const short2_t r = __builtin_elementwise_max(__builtin_bit_cast(short2_t, hh), short2_t{0, 0});
          (e == 0 ? hb.x : hb.y) = __builtin_bit_cast(unsigned, r);
        }
        hb.x = inside ? hb.x : 0u;
        hb.y = inside ? hb.y : 0u;
        *reinterpret_cast<uint2*>(rowp + (((wc * 4 + g) << 4) ^ msw)) = hb;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  FT_LDS_BARRIER();
  // ---- 3x3/s2 max over the tile: thread = (pooled pixel, 8-channel chunk), 16 bytes out ------------------------
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i, pp = idx >> 3, ch = idx & 7;
    const int ppy = pp >> 3, ppx = pp & 7;
    const int py = py0 + ppy, px = px0 + ppx;
    typedef short short8_t __attribute__((ext_vector_type(8)));
    short8_t best = {0, 0, 0, 0, 0, 0, 0, 0};            // the tile holds non-negative halves: their order is the order of their bit patterns
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int m = (2 * ppy + dy) * TS + 2 * ppx + dx;
        const uint4_t v = *reinterpret_cast<const uint4_t*>(stage + m * 128 + ((ch ^ (m & 7)) << 4));
        best = __builtin_elementwise_max(best, __builtin_bit_cast(short8_t, v));
      }
    if (py < Hp && px < Wp && !(FT_STEM_ABL & 2))
      store_out16(p.y + ((((long long)n * Hp + py) * Wp + px) * p.y_cstride + p.y_coff + ch * 8) * 2, __builtin_bit_cast(uint4_t, best));
  }
#endif
}

// Round 6, tried and removed: the <64, 20> form as a PERSISTENT kernel (three workgroups per CU walking four tiles each, the next
// tile's patch gather and the previous tile's stores issued right after the K loop so that they fly under the epilogue and the pool;
// patch behind the stem tile in LDS, scale / shift from LDS, 164 VGPRs).  Bit-identical, 49.6 / 50.0 us against 49.2 / 49.0 us for
// this kernel in the network (same call): the 15 us the gather costs (FT_STEM_ABL=1) are not exposed latency — four workgroups per
// CU already overlap one another's gathers — profiles/README.md, round 6.

// ---- few-output-channel conv (FlowNet predict_flow: Cout = 2, K up to 9 * 1026) -----------------------
// A GEMM tile would leave 30/32 MFMA columns idle and serialise a 9k-long K loop in a handful of
// workgroups.  Instead: one wave per group of PIX consecutive output pixels, the 64 lanes split the
// (tap, channel-group) axis with 16-byte coalesced loads, fp32 dot products (v_dot2_f32_f16 for fp16: no
// explicit converts), a transpose-reduce butterfly (PIX*NCO values -> one per lane group), the (tiny) weight
// set in LDS once per workgroup, each weight vector reused for PIX pixels from registers.
// Uses the generic packed layout [Cout_pad][Kpad].
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float dot8(const uint4_t& a, const uint4_t& b, float c, half_t*) {
  // whole-vector bit casts only (bit_cast of one vector element reads element 0, see mma_slice)
  const half8_t ah = __builtin_bit_cast(half8_t, a), bh = __builtin_bit_cast(half8_t, b);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const half2_t a2 = {ah[2 * q], ah[2 * q + 1]}, b2 = {bh[2 * q], bh[2 * q + 1]};
    c = __builtin_amdgcn_fdot2(a2, b2, c, false);
  }
  return c;
}
__device__ __forceinline__ float dot8(const uint4_t& a, const uint4_t& b, float c, float*) {
  const float4_t af = __builtin_bit_cast(float4_t, a), bf = __builtin_bit_cast(float4_t, b);
#pragma unroll
  for (int q = 0; q < 4; ++q) c += af[q] * bf[q];
  return c;
}

template <typename T, int NCO, int PIX>
__global__ __launch_bounds__(256) void conv_fewout_kernel(const ConvParams p, int ppb) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int NV = PIX * NCO;  // partial sums per lane
  static_assert(NV == 8, "the reduction below folds exactly 8 values over the 64 lanes");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nvec = p.kh * p.kw * p.cin_groups;   // 16-byte vectors along K per output pixel
  // Round 5: no weight staging and no branches around the loads.  A wave makes ONE pass over its PIX pixels (ppb = 16 for every
  // layer below 32 k pixels), so copying the weight set into LDS first bought nothing but a barrier; and the input loads sat
  // each in its own divergent region behind a full wait (18 K steps x 4 dependent L2 round trips: 20 us for 28 MFLOP at
  // FlowNet's predict_flow6).  Now: weights straight from L2 (every wave reads the same lines), inputs as buffer loads whose
  // offset is out of range where the tap leaves the image, two K vectors per lane in flight = 12 independent loads per wait.
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;
  const int m_begin = blockIdx.x * ppb;
  const int m_end = m_begin + ppb < p.M ? m_begin + ppb : p.M;
  const int tap0 = lane / p.cin_groups;
  const int cg0 = lane - tap0 * p.cin_groups;
  for (int m = m_begin + wave * PIX; m < m_end; m += 4 * PIX) {
    // decode PIX consecutive output pixels (one division pair, then carries)
    int pn[PIX], py_[PIX], px_[PIX];
    {
      int n = m / p.HqWq;
      const int rem = m - n * p.HqWq;
      int oy = rem / p.Wq;
      int ox = rem - oy * p.Wq;
#pragma unroll
      for (int i = 0; i < PIX; ++i) {
        pn[i] = n; py_[i] = oy; px_[i] = ox;
        if (++ox == p.Wq) { ox = 0; if (++oy * p.Wq == p.HqWq) { oy = 0; ++n; } }
      }
    }
    float acc[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) acc[c] = 0.f;
    int tap = tap0, cg = cg0;
    constexpr int U = 2;
    for (int v = lane; v < nvec; v += 64 * U) {
      uint4_t wv[U][NCO], xv[U][PIX];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int vv = v + 64 * u;
        const bool live = vv < nvec;
        const int vc = live ? vv : 0;
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
#pragma unroll
        for (int c = 0; c < NCO; ++c) wv[u][c] = reinterpret_cast<const uint4_t*>(p.w + (size_t)c * p.Kpad * sizeof(T))[vc];
#pragma unroll
        for (int i = 0; i < PIX; ++i) {
          const int iy = py_[i] * p.sy - p.pad + ky, ix = px_[i] * p.sy - p.pad_x + kx;
          const bool ok = live && m + i < m_end && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
          const unsigned off = (unsigned)((((pn[i] * p.Hi + iy) * p.Wi + ix) * p.x_cstride + p.x_coff + cg * VEC) * (int)sizeof(T));
          xv[u][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, ok ? off : kOOB, 0, 0);
        }
        cg += 64;
        while (cg >= p.cin_groups) { cg -= p.cin_groups; ++tap; }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int i = 0; i < PIX; ++i)
#pragma unroll
          for (int c = 0; c < NCO; ++c) acc[i * NCO + c] = dot8(xv[u][i], wv[u][c], acc[i * NCO + c], (T*)nullptr);
    }
    // transpose-reduce: 8 values x 64 lanes -> value j summed over all lanes, held by lanes with (lane>>3)==j
    float r4[4], r2[2], r1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // offset 32: lower half keeps values 0..3, upper half 4..7
      const float keep = lane & 32 ? acc[4 + j] : acc[j];
      const float send = lane & 32 ? acc[j] : acc[4 + j];
      r4[j] = keep + __shfl_xor(send, 32);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {   // offset 16
      const float keep = lane & 16 ? r4[2 + j] : r4[j];
      const float send = lane & 16 ? r4[j] : r4[2 + j];
      r2[j] = keep + __shfl_xor(send, 16);
    }
    {
      const float keep = lane & 8 ? r2[1] : r2[0];
      const float send = lane & 8 ? r2[0] : r2[1];
      r1 = keep + __shfl_xor(send, 8);
    }
    r1 += __shfl_xor(r1, 4);
    r1 += __shfl_xor(r1, 2);
    r1 += __shfl_xor(r1, 1);
    if ((lane & 7) == 0) {
      const int vi = lane >> 3;            // which of the 8 values this lane group holds
      const int i = vi / NCO, c = vi - i * NCO;
      if (m + i < m_end && c < p.Cout) {
        const size_t opix = ((size_t)pn[i] * p.Ho + py_[i]) * p.Wo + px_[i];
        float v = r1;
        if (p.scale) v *= p.scale[c];
        if (p.shift) v += p.shift[c];
        if (p.res) v += (float)reinterpret_cast<const T*>(p.res)[opix * p.res_cstride + p.res_coff + c];
        v = apply_act(v, p.act, p.slope);
        if (p.out_layout == FT_LAYOUT_NHWC)
          reinterpret_cast<T*>(p.y)[opix * p.y_cstride + p.y_coff + c] = (T)v;
        else
          reinterpret_cast<float*>(p.y)[((size_t)pn[i] * p.Cout + c) * p.Ho * p.Wo + (size_t)py_[i] * p.Wo + px_[i]] = v;
      }
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------
constexpr int kBKB = 64;       // generic kernel: bytes of K per tile row per step
constexpr int kBP = 128;       // generic kernel: pixel tile
// ---- predict_flow path (fp16): 3x3/s1/p1 conv to <= 2 output channels from an LDS-resident input patch ------
// The few-output kernel above re-reads each input pixel for its 9 taps from L2; at 96x128 x 224 channels that is
// ~0.8 GB of L2 traffic for 88 MB of input.  Here a workgroup owns an 8x16 (or 16x8) output patch: the 10x18 input
// patch of a 64-channel chunk is DMA'd once into LDS (double-buffered over chunks, XOR-swizzled like the halo
// kernel), thread (pixel, channel half) accumulates both outputs with v_dot2_f32_f16 against weights read as LDS
// broadcasts, and the two channel halves are summed through LDS.  Uses the generic packed layout [Cout_pad][Kpad].
template <int TW>
__global__ __launch_bounds__(256, 3) void conv_pflow_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BP = 128, TH = BP / TW, NW = 4;
  constexpr int PW = TW + 2, NPIX = (TH + 2) * PW;
  constexpr int NPWW = (NPIX + 31) / 32;                 // patch wave-loads per wave per chunk (8 pixels per load, 4 waves)
  constexpr int PB = NPWW * NW * 1024;
  constexpr int WB = 9 * 2 * 128;                        // weights of one chunk: [tap][co][64 ch] fp16
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* const patch0 = smem;
  char* const wts0 = smem + 2 * PB;
  float* const red = reinterpret_cast<float*>(wts0 + 2 * WB);

  int ptile;
  {
    const int total = p.npt;
    const int b = blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = b & 7, loc = b >> 3;
    ptile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tiles_per_img = p.h_ty * p.h_tx;
  const int n = ptile / tiles_per_img;
  const int trem = ptile - n * tiles_per_img;
  const int tyi = trem / p.h_tx, txi = trem - tyi * p.h_tx;
  const int qy0 = tyi * TH, qx0 = txi * TW;
  const int iy_org = qy0 - 1, ix_org = qx0 - 1;

  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;
  unsigned p_voff[NPWW];
  {
    const int ppl = lane >> 3, pos = lane & 7;
#pragma unroll
    for (int t = 0; t < NPWW; ++t) {
      const int pp = (t * NW + wave) * 8 + ppl;
      unsigned v = kOOB;
      if (pp < NPIX) {
        const int pr = pp / PW, pc = pp - pr * PW;
        const int iy = iy_org + pr, ix = ix_org + pc;
        if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
          v = (unsigned)((((n * p.Hi + iy) * p.Wi + ix) * p.x_cstride + p.x_coff) * 2 + ((pos ^ (pp & 7)) << 4));
      }
      p_voff[t] = v;
    }
  }
  const int nchunks = (p.cin_groups + 7) >> 3;           // 64-channel chunks (cin_groups = 8-channel groups)
  const int cin8 = p.cin_groups * 8;
  auto load_patch = [&](int c) {
#pragma unroll
    for (int t = 0; t < NPWW; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr)(patch0 + (c & 1) * PB + (t * NW + wave) * 1024), 16, p_voff[t],
                                               c * 128, 0, 0);
  };
  // weights of a chunk: 18 (tap, co) rows of 8 x 16 bytes; threads 0..143 carry one piece each
  const int w_tapco = tid >> 3, w_piece = tid & 7;
  const bool w_thread = tid < 144;
  auto fetch_w = [&](int c) {
    uint4_t v = {0u, 0u, 0u, 0u};
    const int cg = c * 8 + w_piece, tap = w_tapco >> 1, co = w_tapco & 1;
    if (w_thread && cg < p.cin_groups && co < p.Cout)
      v = *reinterpret_cast<const uint4_t*>(p.w + ((size_t)co * p.Kpad + (size_t)tap * cin8 + cg * 8) * 2);
    return v;
  };
  auto store_w = [&](int c, const uint4_t& v) {
    if (w_thread) *reinterpret_cast<uint4_t*>(wts0 + (c & 1) * WB + w_tapco * 128 + w_piece * 16) = v;
  };

  const int m = tid & (BP - 1), half = tid >> 7;        // pixel of the patch, channel half of the chunk (wave-uniform)
  const int r0 = (m / TW) * PW + (m % TW);
  float acc0 = 0.f, acc1 = 0.f;

  load_patch(0);
  store_w(0, fetch_w(0));
  for (int c = 0; c < nchunks; ++c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint4_t wnext = {0u, 0u, 0u, 0u};
    const bool more = c + 1 < nchunks;
    if (more) {
      load_patch(c + 1);
      wnext = fetch_w(c + 1);
    }
    const char* pb = patch0 + (c & 1) * PB;
    const char* wb = wts0 + (c & 1) * WB + half * 64;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {     // not unrolled: the scheduler would hoist all 108 fragment reads and spill
      const int r = r0 + (tap / 3) * PW + (tap % 3);
      const char* rowp = pb + r * 128;
      const int sw = r & 7;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint4_t xv = *reinterpret_cast<const uint4_t*>(rowp + (((half * 4 + k) ^ sw) << 4));
        const uint4_t w0 = *reinterpret_cast<const uint4_t*>(wb + (tap * 2 + 0) * 128 + k * 16);
        const uint4_t w1 = *reinterpret_cast<const uint4_t*>(wb + (tap * 2 + 1) * 128 + k * 16);
        acc0 = dot8(xv, w0, acc0, (half_t*)nullptr);
        acc1 = dot8(xv, w1, acc1, (half_t*)nullptr);
      }
    }
    if (more) store_w(c + 1, wnext);
  }
  // sum the two channel halves, then bias / activation and the store
  if (half == 1) {
    red[m * 2] = acc0;
    red[m * 2 + 1] = acc1;
  }
  __syncthreads();
  if (half == 0) {
    const int oy = qy0 + m / TW, ox = qx0 + m % TW;
    if (oy < p.Ho && ox < p.Wo) {
      float v[2] = {acc0 + red[m * 2], acc1 + red[m * 2 + 1]};
#pragma unroll
      for (int co = 0; co < 2; ++co) {
        if (co < p.Cout) {
          if (p.scale) v[co] *= p.scale[co];
          if (p.shift) v[co] += p.shift[co];
          v[co] = apply_act(v[co], p.act, p.slope);
        }
      }
      if (p.out_layout == FT_LAYOUT_NHWC) {
        half_t* yp = reinterpret_cast<half_t*>(p.y) + (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.y_cstride + p.y_coff;
        yp[0] = (half_t)v[0];
        if (p.Cout > 1) yp[1] = (half_t)v[1];
      } else {
        float* yp = reinterpret_cast<float*>(p.y);
        const size_t hw = (size_t)p.Ho * p.Wo, pix = (size_t)oy * p.Wo + ox;
        yp[((size_t)n * p.Cout) * hw + pix] = v[0];
        if (p.Cout > 1) yp[((size_t)n * p.Cout + 1) * hw + pix] = v[1];
      }
    }
  }
#endif
}

// ---- predict_flow on the matrix pipe (fp16): the 3x3 conv to <= 2 channels as ONE 1x1 GEMM to 18 (tap, co) rows per INPUT
// pixel + a 9-term shift-and-add.  conv_pflow_kernel above still reads every input fragment nine times from LDS and runs the
// dot products on the vector ALU (45 us at 96x128x16 x 194 channels against an 11-us byte floor); here a workgroup takes the
// 14 x 18 input patch of a 12 x 16 output tile (252 pixels = 8 MFMA pixel tiles, two per wave), streams it ONCE through a
// 4-deep ring of 32-channel chunks (LDS-DMA, 64-byte pixel rows, chunk key (pixel / 4) & 3: conflict-free fragment reads),
// multiplies each chunk by the [32 x 16] weight fragments (rows tap * 2 + co, 18 live; all chunks resident in LDS in fragment
// order), parks the fp32 [pixel][18] partials in LDS and lets 192 threads add the nine neighbours of their output pixel.
// MFMA work: 16 instructions per 32 channels per workgroup — the kernel is a stream over its input (1.31x with the halo).
// fp32 summation order differs from the other two kernels (channels inside a tap first), well inside the fp16 tolerance.
__global__ __launch_bounds__(256, 2) void conv_pflow_mfma_kernel(const ConvParams p, int nchunks) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TW = 16, TH = 12, PW = TW + 2, NPIX = (TH + 2) * PW;   // 252
  constexpr int NW = 4, RING = 4, CHB = 16384;                       // 256 pixel slots x 64 bytes per chunk
  constexpr int SP = 20;                                             // floats per pixel of the partial tile (18 + pad: 16-byte rows)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  char* const ring = smem;
  char* const wts = smem + RING * CHB;                               // nchunks x 2 slices x 1 KiB
  float* const part = reinterpret_cast<float*>(smem);                // aliases the ring once the last chunk has been read

  int ptile;
  {
    const int total = p.npt;
    const int b = blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = b & 7, loc = b >> 3;
    ptile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tiles_per_img = p.h_ty * p.h_tx;
  const int n = ptile / tiles_per_img;
  const int trem = ptile - n * tiles_per_img;
  const int tyi = trem / p.h_tx, txi = trem - tyi * p.h_tx;
  const int qy0 = tyi * TH, qx0 = txi * TW;

  // ---- weights: piece (slice s, k half, row co') = 8 consecutive channels of one (tap, co); ordinary loads, before any DMA --
  {
    const int cin8 = p.cin_groups * 8;
    const int npieces = nchunks * 2 * 64;
    for (int i = tid; i < npieces; i += 256) {
      const int sl = i >> 6, l = i & 63, row = l & 31, cg = sl * 2 + (l >> 5);
      const int tap = row >> 1, co = row & 1;
      uint4_t v = {0u, 0u, 0u, 0u};
      if (row < 18 && cg < p.cin_groups && co < p.Cout)
        v = *reinterpret_cast<const uint4_t*>(p.w + ((size_t)co * p.Kpad + (size_t)tap * cin8 + cg * 8) * 2);
      *reinterpret_cast<uint4_t*>(wts + i * 16) = v;
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;
  unsigned p_voff[4];            // 16 pixels per wave-load (4 lanes x 16 bytes each), 4 wave-loads per wave per chunk
  {
    const int ppl = lane >> 2, pos = lane & 3;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int pp = (t * NW + wave) * 16 + ppl;
      unsigned v = kOOB;
      if (pp < NPIX) {
        const int pr = pp / PW, pc = pp - pr * PW;
        const int iy = qy0 - 1 + pr, ix = qx0 - 1 + pc;
        if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
          v = (unsigned)((((n * p.Hi + iy) * p.Wi + ix) * p.x_cstride + p.x_coff) * 2 + ((pos ^ ((pp >> 2) & 3)) << 4));
      }
      p_voff[t] = v;
    }
  }
  auto load_chunk = [&](int c) {       // past the last chunk: out-of-range loads (zeros into a slot nobody reads) keep vmcnt uniform
    const bool live = c < nchunks;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr)(ring + (c & (RING - 1)) * CHB + (t * NW + wave) * 1024), 16,
                                               live ? p_voff[t] : kOOB, live ? c * 64 : 0, 0, 0);
  };
  load_chunk(0);
  load_chunk(1);
  load_chunk(2);

  float16_t acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  int b_off[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int pp = (wave * 2 + t) * 32 + l31;
    b_off[t] = pp * 64 + ((lhi ^ ((pp >> 2) & 3)) << 4);
  }
  const int a_off = lane * 16;

  for (int c = 0; c < nchunks; ++c) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // chunk c has landed (this wave's pieces); chunks c+1, c+2 may fly
    FT_LDS_BARRIER();                                       // ... everyone's pieces; and every read of chunk c-1 is done
    load_chunk(c + 3);
    const char* cb = ring + (c & (RING - 1)) * CHB;
    const char* wb = wts + c * 2048 + a_off;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const uint4_t fa = *reinterpret_cast<const uint4_t*>(wb + k * 1024);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const uint4_t fb = *reinterpret_cast<const uint4_t*>(cb + (b_off[t] ^ (k << 5)));
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, fa), __builtin_bit_cast(half8_t, fb), acc[t], 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the dummy tail loads too: the ring becomes the partial tile
  FT_LDS_BARRIER();
  // ---- partials [pixel][18]: accumulator register r of lane (pixel, half) is row 8 * (r / 4) + 4 * half + r % 4 -------------
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int pp = (wave * 2 + t) * 32 + l31;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int row0 = g * 8 + lhi * 4;
      if (row0 < 18)
        *reinterpret_cast<float4_t*>(part + pp * SP + row0) = float4_t{acc[t][g * 4], acc[t][g * 4 + 1], acc[t][g * 4 + 2], acc[t][g * 4 + 3]};
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  FT_LDS_BARRIER();
  if (tid < TW * TH) {
    const int oy_l = tid / TW, ox_l = tid - oy_l * TW;
    const int oy = qy0 + oy_l, ox = qx0 + ox_l;
    float v[2] = {0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float2 s2 = *reinterpret_cast<const float2*>(part + ((oy_l + tap / 3) * PW + ox_l + tap % 3) * SP + tap * 2);
      v[0] += s2.x;
      v[1] += s2.y;
    }
    if (oy < p.Ho && ox < p.Wo) {
#pragma unroll
      for (int co = 0; co < 2; ++co) {
        if (co < p.Cout) {
          if (p.scale) v[co] *= p.scale[co];
          if (p.shift) v[co] += p.shift[co];
          v[co] = apply_act(v[co], p.act, p.slope);
        }
      }
      if (p.out_layout == FT_LAYOUT_NHWC) {
        half_t* yp = reinterpret_cast<half_t*>(p.y) + (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.y_cstride + p.y_coff;
        yp[0] = (half_t)v[0];
        if (p.Cout > 1) yp[1] = (half_t)v[1];
      } else {
        float* yp = reinterpret_cast<float*>(p.y);
        const size_t hw = (size_t)p.Ho * p.Wo, pix = (size_t)oy * p.Wo + ox;
        yp[((size_t)n * p.Cout) * hw + pix] = v[0];
        if (p.Cout > 1) yp[((size_t)n * p.Cout + 1) * hw + pix] = v[1];
      }
    }
  }
#endif
}

constexpr int kDmaBKB = FT_DMA_BKB;        // dma kernel: bytes of K per tile row per step (64 -> 32 fp16 / 16 fp32 channels)
constexpr int kDmaStages = FT_DMA_STAGES;  // dma kernel: LDS ring depth

static int validate(const ft_conv_desc* d) {
  if (!d) return FT_ERR_INVALID_ARG;
  if (d->dtype != FT_F16 && d->dtype != FT_F32) return FT_ERR_INVALID_ARG;
  if (d->N <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Cin <= 0 || d->Cout <= 0) return FT_ERR_INVALID_ARG;
  if (d->x_wpitch > 0) {  // row-packed input
    if (d->transposed || d->x_coff != 0 || d->x_cstride % 4 || d->x_cstride < d->Cin) return FT_ERR_INVALID_ARG;
    if (d->x_lpad < d->pad || d->x_wpitch < d->x_lpad + d->Wi) return FT_ERR_INVALID_ARG;
  } else {
    if (d->x_wpitch < 0 || d->x_lpad != 0) return FT_ERR_INVALID_ARG;
    if (d->x_cstride % 8 || d->x_coff % 8 || d->x_coff < 0) return FT_ERR_INVALID_ARG;
    if (d->x_cstride < d->x_coff + round_up(d->Cin, 8)) return FT_ERR_INVALID_ARG;
  }
  if (d->transposed) {
    if (d->kh != 4 || d->kw != 4 || d->stride != 2 || d->pad != 1) return FT_ERR_UNSUPPORTED;
    if (d->Ho != 2 * d->Hi || d->Wo != 2 * d->Wi) return FT_ERR_INVALID_ARG;
  } else {
    if (d->kh <= 0 || d->kw <= 0 || d->stride <= 0 || d->pad < 0) return FT_ERR_INVALID_ARG;
    if (d->Ho != (d->Hi + 2 * d->pad - d->kh) / d->stride + 1) return FT_ERR_INVALID_ARG;
    if (d->Wo != (d->Wi + 2 * d->pad - d->kw) / d->stride + 1) return FT_ERR_INVALID_ARG;
  }
  if (d->out_layout == FT_LAYOUT_NHWC) {
    if (d->y_cstride % 4 || d->y_coff % 4 || d->y_coff < 0) return FT_ERR_INVALID_ARG;
    if (d->y_cstride < d->y_coff + (d->tail_cout > 0 ? d->tail_cout : d->Cout)) return FT_ERR_INVALID_ARG;
  } else if (d->out_layout != FT_LAYOUT_NCHW_F32) {
    return FT_ERR_INVALID_ARG;
  }
  if (d->has_residual) {
    if (d->res_cstride % 4 || d->res_coff % 4 || d->res_coff < 0) return FT_ERR_INVALID_ARG;
    if (d->res_cstride < d->res_coff + d->Cout) return FT_ERR_INVALID_ARG;
  }
  if (d->act < FT_ACT_NONE || d->act > FT_ACT_LEAKY) return FT_ERR_INVALID_ARG;
  if (d->tail_cout < 0 || d->tail_cout > 32) return FT_ERR_INVALID_ARG;
  if (d->tail_cout > 0) {   // fused tail 1x1 conv: the whole channel dimension of a pixel tile must sit in one workgroup
    if (d->dtype != FT_F16 || d->has_residual || d->x2_cin != 0 || !(d->Cout == 64 || d->Cout == 128 || d->Cout == 256))
      return FT_ERR_UNSUPPORTED;
  }
  if (d->pool != 0 && d->pool != 1) return FT_ERR_INVALID_ARG;
  if (d->x2_cin < 0) return FT_ERR_INVALID_ARG;
  if (d->x2_cin > 0) {   // second input (K-concat): 1x1 / stride 1 main conv, no residual, plain NHWC views
    if (d->transposed || d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0 || d->has_residual || d->x_wpitch > 0)
      return FT_ERR_UNSUPPORTED;
    if (d->x2_stride <= 0 || d->x2_hi <= 0 || d->x2_wi <= 0 || d->x2_cstride % 8 || d->x2_coff % 8 || d->x2_coff < 0)
      return FT_ERR_INVALID_ARG;
    if (d->x2_cstride < d->x2_coff + round_up(d->x2_cin, 8)) return FT_ERR_INVALID_ARG;
    if (d->Ho != (d->x2_hi - 1) / d->x2_stride + 1 || d->Wo != (d->x2_wi - 1) / d->x2_stride + 1) return FT_ERR_INVALID_ARG;
  }
  return FT_OK;
}

// Packed-weight layout + kernel choice. Everything here depends only on (dtype, Cin, Cout, kernel,
// transposed, x_cstride - x_coff): NOT on the batch / spatial size, so weights are packed once per layer.
struct Geometry {
  int nphases, ntaps, cin_pad, cout_pad, kpad;
  int run_taps, run_cpad;         // row-packed: kernel columns per K-run and their channel stride; else 1, cin_pad
  int dma;                        // 1: conv_igemm_dma_kernel, 0: generic conv_igemm_kernel
  int rowpack;
  int nk, cin_groups, vec, kc;    // K-loop bookkeeping of the chosen kernel
  int cin2_pad, kc2;              // second input (K-concat): its padded K-run and K-steps
};

static int geometry(const ft_conv_desc* d, Geometry* g) {
  int st = validate(d);
  if (st != FT_OK) return st;
  const int esz = d->dtype == FT_F16 ? 2 : 4;
  g->vec = 16 / esz;
  g->nphases = d->transposed ? 4 : 1;
  g->ntaps = d->transposed ? 4 : d->kh * d->kw;
  const int bk = kDmaBKB / esz;
  g->rowpack = 0;
  g->cin2_pad = g->kc2 = 0;
  if (d->x_wpitch > 0) {
    // one K-run = one kernel row: kw taps x x_cstride channels, padded to whole K-steps with zero weights
    if (d->Cout <= 32 || d->kh > 32) return FT_ERR_UNSUPPORTED;
    const int run = round_up(d->kw * d->x_cstride, bk);
    if ((long long)128 * d->kh * run * esz >= (1LL << 31)) return FT_ERR_UNSUPPORTED;
    g->rowpack = g->dma = 1;
    g->ntaps = d->kh;
    g->cin_pad = run;
    g->cout_pad = round_up(d->Cout, d->Cout % 128 == 0 ? 128 : 64);
    g->kc = run / bk;
    g->nk = g->ntaps * g->kc;
    g->kpad = g->ntaps * run;
    g->cin_groups = 0;
    g->run_taps = d->kw;
    g->run_cpad = d->x_cstride;
    return FT_OK;
  }
  const int cin_bk = round_up(d->Cin, bk);
  const long long wbytes = (long long)128 * g->ntaps * cin_bk * esz;  // one co tile of packed weights
  // Cout <= 32 runs on the generic kernel (a 64-wide channel tile would idle half of it) — except 3x3/s1 layers with
  // 16..32 outputs at high resolution (FlowNetFusion inter_conv0/1): channel-aligned layout + LDS-patch (halo) kernel
  // cut their operand traffic so much that the idle half does not matter (inter_conv1: 680 -> 324 us before the patch)
  // (the transposed 16..32-output layers were tried too: fusion deconv0 578 -> 514 us, deconv1 91 -> 103 us: left alone)
  const bool small_3x3 = d->Cout >= 16 && !d->transposed && d->kh == 3 && d->kw == 3 && d->stride == 1;
  g->dma = (d->Cout > 32 || small_3x3) && g->ntaps <= 32 && d->x_cstride >= d->x_coff + cin_bk && wbytes < (1LL << 31);
  if (d->x2_cin > 0) {     // K-concat needs the channel-aligned kernel for both inputs
    const int cin2_bk = round_up(d->x2_cin, bk);
    if (!g->dma || d->x2_cstride < d->x2_coff + cin2_bk || cin_bk / bk < 2) return FT_ERR_UNSUPPORTED;
    g->cin2_pad = cin2_bk;
    g->kc2 = cin2_bk / bk;
  }
  if (g->dma) {
    g->cin_pad = cin_bk;
    g->cout_pad = round_up(d->Cout, d->Cout % 128 == 0 ? 128 : 64);
    g->kc = cin_bk / bk;
    g->nk = g->ntaps * g->kc + g->kc2;
    g->kpad = g->ntaps * cin_bk + g->cin2_pad;
    g->cin_groups = 0;
  } else {
    g->cin_pad = round_up(d->Cin, 8);
    const int bc = d->Cout <= 32 ? 32 : (d->Cout % 128 == 0 ? 128 : 64);
    g->cout_pad = round_up(d->Cout, bc);
    g->cin_groups = g->cin_pad / g->vec;
    const int ch = kBKB / 16;
    g->nk = ceil_div(g->ntaps * g->cin_groups, ch);
    g->kpad = g->nk * ch * g->vec;
    g->kc = 0;
  }
  g->run_taps = 1;
  g->run_cpad = g->cin_pad;
  return FT_OK;
}

template <typename T, int BC, int WGP, int WGC>
static void launch_generic(const ConvParams& p, dim3 grid, hipStream_t s) {
  constexpr size_t lds = 2 * (size_t)(BC + kBP) * kBKB + (size_t)kBP * 8;  // 2 stages + per-row output offsets
  static_assert((size_t)kBP * BC * 2 <= 2 * (size_t)(BC + kBP) * kBKB, "fp16 output tile must fit in the stage buffers");
  hipLaunchKernelGGL((conv_igemm_kernel<T, kBP, BC, WGP, WGC, kBKB>), grid, dim3(256), lds, s, p);
}

template <typename T, int BP, int BC, int WGP, int WGC, bool HAS_RES, int KS = 1, int BKB = kDmaBKB, int S = kDmaStages>
static int launch_dma_r(const ConvParams& p, dim3 grid, hipStream_t s) {
  constexpr size_t ring = (size_t)KS * S * (BC + BP) * BKB;
  constexpr size_t lds = ring + (size_t)BP * 8;
  static_assert((size_t)BP * BC * 2 <= ring, "fp16 output tile must fit in the ring");
  static_assert((size_t)(KS - 1) * BP * BC * 4 <= ring, "K-split partials must fit in the rings");
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto k = conv_igemm_dma_kernel<T, BP, BC, WGP, WGC, BKB, S, HAS_RES, KS>;
  if (lds > 64 * 1024) FT_RAISE_LDS(k, lds);
  hipLaunchKernelGGL(k, grid, dim3(64 * WGP * WGC * KS), lds, s, p);
  return FT_OK;
}

template <typename T, int BP, int BC, int WGP, int WGC, int KS = 1, int BKB = kDmaBKB, int S = kDmaStages>
static int launch_dma(const ConvParams& p, dim3 grid, hipStream_t s) {
  // the residual-prefetch variant exists for the fp16 LDS-transposed epilogue only
  static const bool no_respre = getenv("FT_NO_RESPRE") != nullptr;   // dev A/B: load the residual in the epilogue instead
  if (sizeof(T) == 2 && p.res && p.epi_lds && !no_respre) return launch_dma_r<T, BP, BC, WGP, WGC, true, KS, BKB, S>(p, grid, s);
  return launch_dma_r<T, BP, BC, WGP, WGC, false, KS, BKB, S>(p, grid, s);
}

// "wide-K" variants (fp16): 128 bytes of K per tile row per step = whole 128-byte lines per row from L2, half the
// barriers and K-steps; twice the LDS per stage, so fewer stages / workgroups per CU.  Offered to the tile benchmark
// for the layers whose K-loop is latency-bound (few workgroups, long K).
constexpr int kHintWideShift = 28;   // tile_hint bits 28-29: wide-K level w, BKB = 64 << w

// Tile variants the dma kernel is instantiated for (ft_conv_tile_candidates / ft_conv_desc.tile_hint).
#ifndef FT_HALO_CCH
#define FT_HALO_CCH 32   // channels per resident patch chunk of the 3x3 variant (same-box A/B vs 64: +2.4 % R50, +0.8 % FlowNet2S)
#endif
template <int BC, int TW, int NTAPS, int KW, int S, int CCH>
static int launch_halo_k(const ConvParams& p, dim3 grid, size_t lds, hipStream_t s) {
  auto k = conv_halo_kernel<BC, TW, NTAPS, KW, S, CCH>;
  if (lds > 64 * 1024) FT_RAISE_LDS(k, 160 * 1024);
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, p);
  return FT_OK;
}

template <int KH, int STRIDE, int RUNB, int NT>
static int launch_stem_k(const ConvParams& p, dim3 grid, size_t lds, hipStream_t s) {
  auto k = conv_stem_kernel<KH, STRIDE, RUNB, NT>;
  if (lds > 64 * 1024) FT_RAISE_LDS(k, 160 * 1024);
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, p);
  return FT_OK;
}

template <int KH, int STRIDE, int RUNB>
static int launch_stem_nt(const ConvParams& p, int nt, dim3 grid, size_t lds, hipStream_t s) {
  if (nt == 2) return launch_stem_k<KH, STRIDE, RUNB, 2>(p, grid, lds, s);
  return launch_stem_k<KH, STRIDE, RUNB, 1>(p, grid, lds, s);
}

// geometry of the persistent stem form for `d` (pw / npww / lds), or false where it does not apply
static bool stem_persist_plan(const ft_conv_desc* d, const Geometry& g, int* pw_out, int* npww_out, size_t* lds_out, int* ntiles_out) {
  static const bool no_persist = getenv("FT_STEM_PERSIST") && atoi(getenv("FT_STEM_PERSIST")) == 0;
  const int runb = g.cin_pad * 2, cpb = d->x_cstride * 2;
  const int ph = 7 * d->stride + d->kh;
  const int rb1 = 15 * d->stride * cpb + runb;
  const int pw = (d->stride == 2 && cpb == 16) ? round_up(ceil_div(rb1, 16), 2) : ceil_div(rb1, 16);   // the kernel's parity swizzle pairs chunks
  const int npww = ceil_div(ceil_div(ph * pw, 64), 4);
  const size_t lds = (size_t)d->kh * 64 * runb + 2 * (2 * (size_t)npww * 4096 + 16384);   // weights + two quartets' patch pair and output tile
  const long long ybytes = (long long)d->N * d->Ho * d->Wo * d->y_cstride * 2;
  const int ntiles = d->N * ceil_div(d->Ho, 8) * ceil_div(d->Wo, 16);
  if (no_persist || d->kh != 7 || npww > 12 || lds > 160 * 1024 || ybytes >= (1LL << 31) || ntiles < 512) return false;
  *pw_out = pw; *npww_out = npww; *lds_out = lds; *ntiles_out = ntiles;
  return true;
}

static int launch_stem(ConvParams p, const ft_conv_desc* d, const Geometry& g, hipStream_t s) {
  const int runb = g.cin_pad * 2, cpb = d->x_cstride * 2;
  const int ph = 7 * d->stride + d->kh;
  // super-tile = nt tiles of 8x16 outputs side by side share one pass over the weights.  Measured (in situ, us):
  // pose stem (64-byte rows) nt 1/2/4 = 47.5 / 53.6 / 73; FlowNet stem (128-byte rows) 72.2 / 67.6 / 81.4
  static const int force_nt = getenv("FT_STEM_NT") ? atoi(getenv("FT_STEM_NT")) : 0;   // dev knob (1 or 2)
  int nt = force_nt == 1 || force_nt == 2 ? force_nt : (runb >= 128 ? 2 : 1);
  if (nt == 2 && ((ceil_div(d->Wo, 32) * 32 - d->Wo) * 100 / d->Wo >= 10 || ph * round_up(31 * d->stride * cpb + runb, 16) > 56 * 1024))
    nt = 1;
  {
    // persistent weight-stationary form (one workgroup per CU walks many 8 x 16 tiles): FT_STEM_PERSIST=0 keeps the per-tile kernel
    int pw, npww, ntiles;
    size_t lds;
    const long long ybytes = (long long)d->N * d->Ho * d->Wo * d->y_cstride * 2;
    if (stem_persist_plan(d, g, &pw, &npww, &lds, &ntiles)) {
      p.h_pw = pw;
      p.h_npww = npww;
      p.h_pb = npww * 4 * 1024;
      p.h_ty = ceil_div(d->Ho, 8);
      p.h_tx = ceil_div(d->Wo, 16);
      p.npt = ntiles;
      p.nct = 1;
      p.y_bytes = (unsigned)ybytes;
      int dev = 0, ncu = 256;
      FT_HIP_CHECK(hipGetDevice(&dev));
      FT_HIP_CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
      const int pairs = (ntiles + 1) / 2;
      dim3 grid(ncu < pairs ? ncu : pairs);
      if (runb == 64) { auto k = conv_stem_persist_kernel<7, 2, 64>; FT_RAISE_LDS(k, 160 * 1024); hipLaunchKernelGGL(k, grid, dim3(512), lds, s, p); }
      else if (runb == 128) { auto k = conv_stem_persist_kernel<7, 2, 128>; FT_RAISE_LDS(k, 160 * 1024); hipLaunchKernelGGL(k, grid, dim3(512), lds, s, p); }
      else { auto k = conv_stem_persist_kernel<7, 2, 256>; FT_RAISE_LDS(k, 160 * 1024); hipLaunchKernelGGL(k, grid, dim3(512), lds, s, p); }
      FT_LAUNCH_CHECK("conv_stem_persist_kernel");
      return FT_OK;
    }
  }
  if (p.shift_n) return FT_ERR_UNSUPPORTED;              // the per-tile kernel reads one shared shift vector
  const int tws = 16 * nt;
  const int rb = (tws - 1) * d->stride * cpb + runb;     // bytes of one patch row
  p.h_pw = ceil_div(rb, 16);
  p.h_npww = ceil_div(ceil_div(ph * p.h_pw, 64), 4);
  if (p.h_npww > 12) return FT_ERR_UNSUPPORTED;
  p.h_pb = p.h_npww * 4 * 1024;
  p.h_ty = ceil_div(d->Ho, 8);
  p.h_tx = ceil_div(d->Wo, tws);
  p.npt = d->N * p.h_ty * p.h_tx;
  p.nct = 1;
  const size_t ring = (size_t)3 * 64 * runb;
  if (ring + p.h_pb < 16384) p.h_pb = 16384 - (int)ring;  // the epilogue's 16 KiB output tile must not reach the row table
  const size_t lds = ring + p.h_pb + (size_t)nt * 128 * 8;
  dim3 grid(p.npt);
  int rc;
  if (d->kh == 7)
    rc = runb == 64 ? launch_stem_nt<7, 2, 64>(p, nt, grid, lds, s) : runb == 128 ? launch_stem_nt<7, 2, 128>(p, nt, grid, lds, s)
                                                                                  : launch_stem_nt<7, 2, 256>(p, nt, grid, lds, s);
  else
    rc = runb == 64 ? launch_stem_nt<3, 1, 64>(p, nt, grid, lds, s) : launch_stem_nt<3, 1, 128>(p, nt, grid, lds, s);
  if (rc != FT_OK) return rc;
  FT_LAUNCH_CHECK("conv_stem_kernel");
  return FT_OK;
}

static bool stem_ok(const ft_conv_desc* d, const Geometry& g);

// ft_conv_desc.pool: the pose stem with its max-pool fused (conv_stem_pool_kernel)
static int launch_stem_pool(ConvParams p, const ft_conv_desc* d, const Geometry& g, hipStream_t s) {
  const int runb = g.cin_pad * 2, cpb = d->x_cstride * 2;
  if (!stem_ok(d, g) || d->kh != 7 || runb != 64 || d->act != FT_ACT_RELU || d->Ho % 2 || d->Wo % 2 || d->x_lpad < d->pad + 2 ||
      ((d->x_lpad - d->pad - 2) * cpb) % 16 != 0)
    return FT_ERR_UNSUPPORTED;
  const int Hp = d->Ho / 2, Wp = d->Wo / 2;
  const int rb = 16 * d->stride * cpb + runb;            // bytes of one patch row (17 stem columns)
  p.h_pw = ceil_div(rb, 16);
  p.h_npww = ceil_div(ceil_div(39 * p.h_pw, 64), 4);
  if (p.h_npww > 12 || (p.x_planar && (p.h_npww > 5 || cpb != 8))) return FT_ERR_UNSUPPORTED;
  p.h_pb = p.h_npww * 4 * 1024;
  p.h_ty = ceil_div(Hp, 8);
  p.h_tx = ceil_div(Wp, 8);
  p.npt = d->N * p.h_ty * p.h_tx;
  p.nct = 1;
  size_t lds = (size_t)(FT_STEM_POOL_ALLW ? 7 : 3) * 64 * runb + p.h_pb;
  if (lds < 320 * 128) lds = 320 * 128;
  if (p.x_planar && p.h_pw == 20 && d->Cin == 3)      // the pose stem from the NCHW fp32 crop: compile-time patch pitch, division-free gather
    hipLaunchKernelGGL((conv_stem_pool_kernel<64, 20>), dim3(p.npt), dim3(256), lds, s, p, Hp, Wp);
  else
    hipLaunchKernelGGL(conv_stem_pool_kernel<64>, dim3(p.npt), dim3(256), lds, s, p, Hp, Wp);
  FT_LAUNCH_CHECK("conv_stem_pool_kernel");
  return FT_OK;
}

static int launch_halo(ConvParams p, const ft_conv_desc* d, const Geometry& g, int bc, hipStream_t s) {
  if (g.rowpack) return launch_stem(p, d, g, s);
  const int Hq = p.HqWq / p.Wq, Wq = p.Wq;
  const int kq = d->transposed ? 2 : 3;                      // taps per dimension of one phase
  // patch shape 8x16 or 16x8 output pixels: the one that wastes fewer pixels on this image size
  const long long t16 = (long long)ceil_div(Hq, 8) * ceil_div(Wq, 16), t8 = (long long)ceil_div(Hq, 16) * ceil_div(Wq, 8);
  const int tw = t16 <= t8 ? 16 : 8, th = 128 / tw;
  p.h_ty = ceil_div(Hq, th);
  p.h_tx = ceil_div(Wq, tw);
  p.h_pw = tw + kq - 1;
  p.h_npix = (th + kq - 1) * p.h_pw;
  // 3x3: 32-channel patch chunks (49 KiB of LDS = 3 workgroups per CU; 9 K-steps per chunk on a 3-slot weight ring);
  // transposed phases: 64-channel chunks (8 K-steps per chunk on a 4-slot ring)
  const int cch = d->transposed ? 64 : FT_HALO_CCH;
  p.h_npww = ceil_div(ceil_div(p.h_npix, 1024 / (cch * 2)), 4);
  if (p.h_npww > (d->transposed ? 5 : 6)) return FT_ERR_UNSUPPORTED;
  p.h_pb = p.h_npww * 4 * 1024;
  const int S = d->transposed ? 4 : 3;
  const size_t lds = (size_t)S * bc * 64 + 2 * (size_t)p.h_pb + 1024 + 128 * 8;
  const int N = p.M / p.HqWq;
  p.npt = N * p.h_ty * p.h_tx;
  p.nct = g.cout_pad / bc;
  if ((long long)p.npt * p.nct * p.nph > 0x7fffffffLL) return FT_ERR_UNSUPPORTED;
  dim3 grid(p.npt * p.nct * p.nph);
  int rc;
  if (d->transposed) {
    if (bc == 128) rc = tw == 16 ? launch_halo_k<128, 16, 4, 2, 4, 64>(p, grid, lds, s) : launch_halo_k<128, 8, 4, 2, 4, 64>(p, grid, lds, s);
    else rc = tw == 16 ? launch_halo_k<64, 16, 4, 2, 4, 64>(p, grid, lds, s) : launch_halo_k<64, 8, 4, 2, 4, 64>(p, grid, lds, s);
  } else {
    if (bc == 128) rc = tw == 16 ? launch_halo_k<128, 16, 9, 3, 3, FT_HALO_CCH>(p, grid, lds, s) : launch_halo_k<128, 8, 9, 3, 3, FT_HALO_CCH>(p, grid, lds, s);
    else rc = tw == 16 ? launch_halo_k<64, 16, 9, 3, 3, FT_HALO_CCH>(p, grid, lds, s) : launch_halo_k<64, 8, 9, 3, 3, FT_HALO_CCH>(p, grid, lds, s);
  }
  if (rc != FT_OK) return rc;
  FT_LAUNCH_CHECK("conv_halo_kernel");
  return FT_OK;
}

constexpr int kHintHalo = 1 << 30;   // tile_hint bit 30: LDS-resident input patch (conv_halo_kernel)

// conv_halo_kernel: fp16, 3x3 / stride 1 or the 2x2-tap phases of ConvTranspose2d(4,2,1), NHWC fp16 8-aligned output,
// no residual, channel-aligned (dma) weight layout
static bool stem_ok(const ft_conv_desc* d, const Geometry& g) {
  if (!g.rowpack || d->dtype != FT_F16 || d->has_residual || d->Cout != 64 || d->x_coff != 0) return false;
  if (!((d->kh == 7 && d->stride == 2) || (d->kh == 3 && d->stride == 1))) return false;
  const int runb = g.cin_pad * 2;
  if (!(runb == 64 || runb == 128 || (runb == 256 && d->kh == 7))) return false;
  if ((d->stride * d->x_cstride * 2) % 16 != 0 || d->pad - d->x_lpad > 0) return false;   // 16-byte aligned pixel steps
  return d->out_layout == FT_LAYOUT_NHWC && d->y_coff % 8 == 0 && d->y_cstride % 8 == 0;
}

static bool halo_ok(const ft_conv_desc* d, const Geometry& g) {
  if (g.rowpack) return stem_ok(d, g);
  if (!g.dma || d->dtype != FT_F16 || d->has_residual) return false;
  if (!(d->transposed || (d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1))) return false;
  return d->out_layout == FT_LAYOUT_NHWC && d->Cout % 8 == 0 && d->y_coff % 8 == 0 && d->y_cstride % 8 == 0;
}

// ft_conv_desc.shift_nstride: the persistent row-packed 7x7 / stride 2 fp16 stem only
static bool shift_n_ok(const ft_conv_desc* d, const Geometry& g) {
  if (d->shift_nstride < d->Cout || d->shift_nstride % 4 || d->pool || d->tail_cout > 0 || d->x2_cin > 0 || d->kh != 7 || d->stride != 2)
    return false;
  if (!stem_ok(d, g)) return false;
  if ((unsigned long long)d->N * d->Hi * d->x_wpitch * d->x_cstride * 2 >= (1ull << 31)) return false;
  int pw, npww, ntiles;
  size_t lds;
  return stem_persist_plan(d, g, &pw, &npww, &lds, &ntiles);
}

extern "C" int ft_conv_shift_nstride_supported(const ft_conv_desc* d) {
  Geometry g;
  const int st = geometry(d, &g);
  if (st != FT_OK) return st;
  return shift_n_ok(d, g) ? FT_OK : FT_ERR_UNSUPPORTED;
}

constexpr int kHintSkShift = 21;     // tile_hint bits 21-23: the cross-workgroup K split (needs a workspace): codes 0..3 = 1, 2, 4, 8
                                     // slices (log2, as before); 4..7 = 3, 5, 6, 7 slices (8-phase kernel only: 48 tiles x 5 = 240
                                     // workgroups fill 256 CUs where x 4 leaves a quarter of them idle)
__host__ __device__ constexpr int hint_sk(int code) { return code < 4 ? 1 << code : (code == 4 ? 3 : code + 0); }   // 5, 6, 7 are themselves
constexpr int sk_code(int sk) { return sk == 1 ? 0 : sk == 2 ? 1 : sk == 4 ? 2 : sk == 8 ? 3 : sk == 3 ? 4 : sk; }

constexpr int kWide8 = 3;           // tile_hint bits 28-29 == 3: the 256 x 256 tile on the 8-phase schedule (conv_igemm8.hip)

// conv_igemm8_kernel: fp16, channel-aligned layout, 64-channel K-tiles (cin_pad % 64 == 0), all-256-channel tiles, no second input
static bool igemm8_ok(const ft_conv_desc* d, const Geometry& g) {
  if (!(g.dma && !g.rowpack && d->dtype == FT_F16 && g.kc2 == 0 && g.kc % 2 == 0 && g.cout_pad % 256 == 0 && g.nk / 2 >= 2)) return false;
  if (d->has_residual) return false;     // results leave straight from the accumulator registers: no residual pick-up
  if ((unsigned long long)g.nphases * g.cout_pad * g.kpad * 2 >= (1ull << 31)) return false;   // one buffer descriptor over all weights
  if (g.cout_pad > 1024) return false;   // folded-BN scale / shift of every output channel sit in 8 KiB of LDS
  if (d->tail_cout > 0) return d->Cout == 256 && d->tail_cout <= 24;   // (the channel halves exchange 12 registers = 24 outputs)
  if (!(d->out_layout == FT_LAYOUT_NHWC && d->Cout % 8 == 0 && d->y_coff % 8 == 0 && d->y_cstride % 8 == 0)) return false;
  return (unsigned long long)d->N * d->Ho * d->Wo * d->y_cstride * 2 < (1ull << 31);
}

static bool tile_valid(const ft_conv_desc* d, const Geometry& g, int bp, int bc, int ks, int wide = 0, bool halo = false,
                       int sk = 1) {
  if (!g.dma) return false;
  if (wide == kWide8) {
    if (!igemm8_ok(d, g) || bp != 256 || bc != 256 || ks != 1 || halo) return false;
    if (sk != 1) {     // every K slice = ceil(K-tiles / sk) but the last, which must still hold two K-tiles (the kernel's tile hand-over)
      const int nk8 = g.nk / 2, per = (nk8 + sk - 1) / sk;
      if (sk < 2 || sk > 8 || d->tail_cout > 0 || per < 4 || nk8 - (sk - 1) * per < 2) return false;
    }
    return true;
  }
  if (sk != 1) {     // split-K across workgroups: fp32 partial tiles in a workspace + a reduce launch
    if (!(sk == 2 || sk == 4 || sk == 8) || ks != 1 || halo || g.kc2 > 0 || d->tail_cout > 0 || g.rowpack || bp > 128) return false;
    if ((g.nk >> wide) / sk < 4) return false;
  }
  if (g.kc2 > 0) {   // K-concat runs in the lean loop only: no split-K, no halo; wide-K needs whole wide steps of both runs
    if (ks != 1 || halo) return false;
    if (wide && (g.kc2 % 2 != 0 || (g.kc >> 1) < 2)) return false;
  }
  if (halo) return halo_ok(d, g) && bp == 128 && (bc == 64 || (bc == 128 && !g.rowpack)) && g.cout_pad % bc == 0 && ks == 1 && wide == 0;
  if (wide < 0 || wide > 1) return false;   // (BKB = 256 was benchmarked too: never the fastest on any layer)
  if (wide && !(d->dtype == FT_F16 && g.kc % (1 << wide) == 0)) return false;
  if (wide && ks > 1) return false;         // (wide + split-K likewise)
  if (bc == 256) {   // all-256-channel tiles (8 waves at 128 pixels, 4 at 64): the pixel tile is loaded once per 256 outputs
    if (!(d->dtype == FT_F16 && (bp == 64 || bp == 128 || bp == 256) && ks == 1 && sk == 1 && g.cout_pad % 256 == 0)) return false;
    if (bp == 256 && wide != 1) return false;   // 256 x 256 (8 waves, half the operand bytes per MAC of 128 x 128): wide-K form only
    return true;
  }
  if (!((bp == 64 || bp == 128 || bp == 256) && (bc == 64 || bc == 128) && (ks == 1 || ks == 2 || ks == 4))) return false;
  if (g.cout_pad % bc != 0) return false;
  if (bp == 256 && !(bc == 128 && d->dtype == FT_F16 && ks == 1)) return false;
  if (ks > 1) {
    if (d->dtype != FT_F16 || (bp == 128 && bc == 64) || (ks == 4 && bp == 128)) return false;
    if ((g.nk >> wide) < 2 * ks) return false;
    if ((size_t)ks * kDmaStages * (bc + bp) * (kDmaBKB << wide) + 2048 + (size_t)bp * 8 > 160 * 1024) return false;
  }
  return true;
}

static int env_int(const char* name) {  // developer tile overrides (FT_CONV_BP / FT_CONV_BC), 0 = heuristic
  const char* v = getenv(name);
  return v ? atoi(v) : 0;
}

}  // namespace ft

using namespace ft;

extern "C" int ft_conv_tile_candidates(const ft_conv_desc* d, int* hints, int max) {
  Geometry g;
  int st = geometry(d, &g);
  if (st != FT_OK) return -st;
  if (!hints || max <= 0) return -FT_ERR_INVALID_ARG;
  if (d->pool) return 0;            // one kernel
  if (d->tail_cout > 0) {           // pixel tile x ALL channels: the 128-pixel 8-wave tile (the default) or the 8-phase 256 x 256 tile
    if (max < 2 || d->Cout != 256 || !tile_valid(d, g, 256, 256, 1, kWide8)) return 0;
    hints[0] = 128 | (256 << 12) | (1 << 24);
    hints[1] = 256 | (256 << 12) | (1 << 24) | (kWide8 << kHintWideShift);
    return 2;
  }
  static const int kTiles[5][2] = {{256, 128}, {128, 128}, {128, 64}, {64, 128}, {64, 64}};
  int n = 0;
  for (const auto& t : kTiles)
    for (int ks = 1; ks <= 4; ks <<= 1)
      if (n < max && tile_valid(d, g, t[0], t[1], ks)) hints[n++] = t[0] | (t[1] << 12) | (ks << 24);
  for (const auto& t : kTiles)
    if (n < max && tile_valid(d, g, t[0], t[1], 1, 1)) hints[n++] = t[0] | (t[1] << 12) | (1 << 24) | (1 << kHintWideShift);
  for (int bp = 256; bp >= 64; bp >>= 1)
    for (int wide = 0; wide <= 1; ++wide)
      if (n < max && tile_valid(d, g, bp, 256, 1, wide)) hints[n++] = bp | (256 << 12) | (1 << 24) | (wide << kHintWideShift);
  // the 8-phase 256 x 256 tile, and its cross-workgroup split-K forms where the layer has fewer such tiles than CUs
  if (n < max && tile_valid(d, g, 256, 256, 1, kWide8)) {
    hints[n++] = 256 | (256 << 12) | (1 << 24) | (kWide8 << kHintWideShift);
    const int Hq8 = d->transposed ? d->Hi : d->Ho, Wq8 = d->transposed ? d->Wi : d->Wo;
    const long long nblk8 = (long long)ceil_div(d->N * Hq8 * Wq8, 256) * (g.cout_pad / 256) * g.nphases;
    for (int lg = 1; lg <= 3; ++lg)
      if (n < max && nblk8 <= 160 && (nblk8 << lg) <= 640 && tile_valid(d, g, 256, 256, 1, kWide8, false, 1 << lg))
        hints[n++] = 256 | (256 << 12) | (1 << 24) | (kWide8 << kHintWideShift) | (lg << kHintSkShift);
    // odd splits where they bring the workgroup count closer to one round of 256 CUs than the neighbouring powers of two
    // (deconv.0 of the pose head at batch 64: 48 tiles x 5 slices = 240 workgroups; x 4 = 192, x 8 = 384)
    for (int sk = 3; sk <= 7; ++sk) {
      if (sk == 4) continue;
      if (n < max && nblk8 <= 160 && nblk8 * sk > 160 && nblk8 * sk <= 272 && tile_valid(d, g, 256, 256, 1, kWide8, false, sk))
        hints[n++] = 256 | (256 << 12) | (1 << 24) | (kWide8 << kHintWideShift) | (sk_code(sk) << kHintSkShift);
    }
  }
  // cross-workgroup split-K where the layer has less than ~one workgroup per CU even on 64-wide tiles (long K, few
  // pixels: layer4 / FlowNet conv5..6 / every deep layer at small batch).  Needs ft_conv2d_fwd_ws.
  {
    const int Hq = d->transposed ? d->Hi : d->Ho, Wq = d->transposed ? d->Wi : d->Wo;
    const long long M = (long long)d->N * Hq * Wq;
    static const int kSkTiles[3][2] = {{128, 128}, {64, 128}, {64, 64}};
    static const int sk_max_blocks = getenv("FT_SK_MAX_BLOCKS") ? atoi(getenv("FT_SK_MAX_BLOCKS")) : 256;   // dev knob
    for (const auto& t : kSkTiles) {
      if (g.cout_pad % t[1] != 0) continue;
      const long long nblk = (long long)ceil_div((int)M, t[0]) * (g.cout_pad / t[1]) * g.nphases;
      const int wide = tile_valid(d, g, t[0], t[1], 1, 1) ? 1 : 0;
      for (int lg = 1; lg <= 3; ++lg)
        if (n < max && nblk <= sk_max_blocks && (nblk << lg) <= 4 * sk_max_blocks + 512 && tile_valid(d, g, t[0], t[1], 1, wide, false, 1 << lg))
          hints[n++] = t[0] | (t[1] << 12) | (1 << 24) | (wide << kHintWideShift) | (lg << kHintSkShift);
    }
  }
  // dev: A/B the tile benchmark without the halo variants ("1": none, "stem": no stem kernel, "conv": no conv_halo_kernel)
  static const char* nh = getenv("FT_CONV_NO_HALO");
  const bool no_halo = nh && (nh[0] == '1' || (nh[0] == 's' && g.rowpack) || (nh[0] == 'c' && !g.rowpack));
  for (int bc = 128; bc >= 64 && !no_halo; bc >>= 1)
    if (n < max && tile_valid(d, g, 128, bc, 1, 0, true)) hints[n++] = 128 | (bc << 12) | (1 << 24) | kHintHalo;
  return n;
}

extern "C" int ft_conv_pack_geometry(const ft_conv_desc* d, ft_conv_geometry* out) {
  Geometry g;
  int st = geometry(d, &g);
  if (st != FT_OK) return st;
  if (!out) return FT_ERR_INVALID_ARG;
  out->nphases = g.nphases;
  out->ntaps = g.ntaps;
  out->cin_pad = g.cin_pad;
  out->cout_pad = g.cout_pad;
  out->kpad = g.kpad;
  out->run_taps = g.run_taps;
  out->run_cpad = g.run_cpad;
  out->cin2_pad = g.cin2_pad;
  return FT_OK;
}

extern "C" int ft_conv_tap_source(const ft_conv_desc* d, int phase, int tap, int sub, int* ky, int* kx) {
  Geometry g;
  int st = geometry(d, &g);
  if (st != FT_OK) return st;
  if (phase < 0 || phase >= g.nphases || tap < 0 || tap >= g.ntaps || sub < 0 || sub >= g.run_taps || !ky || !kx)
    return FT_ERR_INVALID_ARG;
  if (g.rowpack) {
    *ky = tap;
    *kx = sub;
  } else if (d->transposed) {
    // out[2q+p] takes in[q + p - t] * W[k],  k = (p == 0) ? 1 + 2t : 2t   
    const int py = phase >> 1, px = phase & 1, ty = tap >> 1, tx = tap & 1;
    *ky = py == 0 ? 1 + 2 * ty : 2 * ty;
    *kx = px == 0 ? 1 + 2 * tx : 2 * tx;
  } else {
    *ky = tap / d->kw;
    *kx = tap % d->kw;
  }
  return FT_OK;
}

extern "C" double ft_conv_flops(const ft_conv_desc* d) {
  if (validate(d) != FT_OK) return 0.0;
  const double taps = d->transposed ? 4.0 : (double)d->kh * d->kw;  // per output pixel
  return 2.0 * d->N * (double)d->Ho * d->Wo * d->Cout * ((double)d->Cin * taps + (double)d->x2_cin + (double)d->tail_cout);
}

extern "C" size_t ft_conv_workspace_bytes(const ft_conv_desc* d) {
  int hints[64];     // every form the enumeration can produce (15 + 5 + 6 + 4 + 9 + 2 = 41): nothing truncated, max_sk is the true maximum
  const int n = ft_conv_tile_candidates(d, hints, 64);
  int max_sk = 1;
  for (int i = 0; i < n; ++i) {
    const int sk = hint_sk((hints[i] >> kHintSkShift) & 7);
    if (sk > max_sk) max_sk = sk;
  }
  if (max_sk == 1) return 0;                 // no split-K variant is offered for this layer
  Geometry g;
  if (geometry(d, &g) != FT_OK) return 0;
  const int Hq = d->transposed ? d->Hi : d->Ho, Wq = d->transposed ? d->Wi : d->Wo;
  return (size_t)max_sk * g.nphases * d->N * Hq * Wq * g.cout_pad * sizeof(float);
}

static int conv2d_fwd_impl(const ft_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift,
                           const void* residual, void* y, void* workspace, size_t workspace_bytes, ft_stream_t stream);

extern "C" int ft_conv2d_fwd(const ft_conv_desc* d, const void* x, const void* w_packed,
                             const float* scale, const float* shift, const void* residual, void* y,
                             ft_stream_t stream) {
  return conv2d_fwd_impl(d, x, w_packed, scale, shift, residual, y, nullptr, 0, stream);
}

extern "C" int ft_conv2d_fwd_ws(const ft_conv_desc* d, const void* x, const void* w_packed,
                                const float* scale, const float* shift, const void* residual, void* y,
                                void* workspace, size_t workspace_bytes, ft_stream_t stream) {
  if (workspace && (reinterpret_cast<uintptr_t>(workspace) & 15)) return FT_ERR_INVALID_ARG;
  return conv2d_fwd_impl(d, x, w_packed, scale, shift, residual, y, workspace, workspace_bytes, stream);
}

static int conv2d_fwd_impl(const ft_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift,
                           const void* residual, void* y, void* workspace, size_t workspace_bytes, ft_stream_t stream) {
  Geometry g;
  int st = geometry(d, &g);
  if (st != FT_OK) return st;
  if (!x || !w_packed || !y) return FT_ERR_INVALID_ARG;
  if (d->has_residual && !residual) return FT_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_packed) |
       reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual) |
       reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift)) & 15)
    return FT_ERR_INVALID_ARG;

  ConvParams p;
  p.x = static_cast<const char*>(x);
  p.w = static_cast<const char*>(w_packed);
  p.scale = scale;
  p.shift = shift;
  p.res = d->has_residual ? static_cast<const char*>(residual) : nullptr;
  p.y = static_cast<char*>(y);
  const int Hq = d->transposed ? d->Hi : d->Ho;
  const int Wq = d->transposed ? d->Wi : d->Wo;
  p.M = d->N * Hq * Wq;
  p.HqWq = Hq * Wq;
  p.Wq = Wq;
  p.Hi = d->Hi;
  p.Wi = d->Wi;
  p.sy = d->transposed ? 1 : d->stride;
  p.x_cstride = d->x_cstride;
  p.x_coff = d->x_coff;
  p.kh = d->transposed ? 2 : d->kh;
  p.kw = d->transposed ? 2 : d->kw;
  p.dmul = d->transposed ? -1 : 1;
  p.pad = d->pad;
  p.pad_x = d->pad;
  p.transposed = d->transposed;
  if (g.rowpack) {      // the kernel sees a (kh x 1)-tap conv over the physically padded rows
    p.kw = 1;
    p.Wi = d->x_wpitch;
    p.pad_x = d->pad - d->x_lpad;
  }
  p.cin_groups = g.cin_groups;
  p.kc = g.kc;
  p.nk = g.nk;
  p.sk = 1;
  p.ws = nullptr;
  p.tail_w = nullptr;
  p.tail_cout = 0;
  p.x2 = nullptr;
  p.kc2 = g.kc2;
  if (g.kc2 > 0) {
    if (!residual) return FT_ERR_INVALID_ARG;
    const unsigned long long x2b = (unsigned long long)d->N * d->x2_hi * d->x2_wi * d->x2_cstride * dtype_size(d->dtype);
    if (x2b >= (1ull << 31)) return FT_ERR_UNSUPPORTED;
    p.x2 = static_cast<const char*>(residual);
    p.x2_bytes = (unsigned)x2b;
    p.x2_hi = d->x2_hi; p.x2_wi = d->x2_wi; p.x2_cstride = d->x2_cstride; p.x2_coff = d->x2_coff; p.x2_stride = d->x2_stride;
  }
  p.Kpad = g.kpad;
  p.Cout = d->Cout;
  p.Cout_pad = g.cout_pad;
  p.Ho = d->Ho;
  p.Wo = d->Wo;
  p.omul = d->transposed ? 2 : 1;
  p.y_cstride = d->y_cstride;
  p.y_coff = d->y_coff;
  p.out_layout = d->out_layout;
  p.res_cstride = d->res_cstride;
  p.res_coff = d->res_coff;
  p.act = d->act;
  p.slope = d->slope;
  p.nph = g.nphases;
  static const int dbg = env_int("FT_CONV_DBG");
  p.dbg = dbg;
  p.shift_n = d->shift_nstride;
  p.x_planar = 0;
  p.x_lpad = d->x_lpad;
  p.x_w = d->Wi;
  p.x_c = d->Cin;
  if (d->x_nchw_f32 != 0 && (!d->pool || !g.rowpack || d->x_cstride != 4 || d->Cin > 3 || d->dtype != FT_F16)) return FT_ERR_UNSUPPORTED;
  p.epi_lds = d->dtype == FT_F16 && d->out_layout == FT_LAYOUT_NHWC && d->Cout % 8 == 0 && d->y_coff % 8 == 0 &&
              d->y_cstride % 8 == 0 && (!d->has_residual || (d->res_coff % 8 == 0 && d->res_cstride % 8 == 0));
  hipStream_t s = as_stream(stream);
  const size_t esz = dtype_size(d->dtype);
  const unsigned long long x_bytes =
      (unsigned long long)d->N * d->Hi * (d->x_wpitch > 0 ? d->x_wpitch : d->Wi) * d->x_cstride * esz;

  if (d->shift_nstride != 0) {      // per-sample shift: straight to the persistent stem, the one kernel that reads it
    if (!shift || !shift_n_ok(d, g)) return FT_ERR_UNSUPPORTED;
    p.x_bytes = (unsigned)x_bytes;
    return launch_stem(p, d, g, s);
  }
  if (d->pool) {            // stem + max-pool: y is the pooled [N, Ho/2, Wo/2, Cout] map
    if (!g.rowpack || x_bytes >= (1ull << 31) || d->tail_cout > 0 || d->x2_cin > 0 || d->has_residual) return FT_ERR_UNSUPPORTED;
    p.x_bytes = (unsigned)x_bytes;
    p.x_planar = d->x_nchw_f32 != 0;
    if (p.x_planar) {       // the buffer descriptor then covers the fp32 planes
      const unsigned long long pb = (unsigned long long)d->N * d->Cin * d->Hi * d->Wi * 4;
      if (pb >= (1ull << 31)) return FT_ERR_UNSUPPORTED;
      p.x_bytes = (unsigned)pb;
    }
    return launch_stem_pool(p, d, g, s);
  }
  if (d->tail_cout > 0) {   // conv + fused tail 1x1 conv: 128 pixels x all Cout channels per workgroup
    if (!g.dma || g.rowpack || x_bytes >= (1ull << 31) || g.cout_pad != d->Cout) return FT_ERR_UNSUPPORTED;
    if (!residual) return FT_ERR_INVALID_ARG;
    p.x_bytes = (unsigned)x_bytes;
    p.tail_w = static_cast<const char*>(residual);
    p.tail_cout = d->tail_cout;
    p.epi_lds = 1;
    // FT_TAIL_BP=256 (dev A/B): 256 pixels x 256 channels per 8-wave workgroup (wide-K form).  Half the weight-tile traffic
    // per pixel, but one workgroup per CU with its 8 waves in lock-step: measured 160 vs 146 us on deconv.6 + heatmap at
    // batch 64 — the 128-pixel tile (two workgroups per CU) stays the default.
    static const int tail_bp = env_int("FT_TAIL_BP");
    const bool big = d->Cout == 256 && g.kc % 2 == 0 && tail_bp == 256;
    static const int tail_8ph = env_int("FT_TAIL_8PH");   // dev: 1 forces the 8-phase tile, -1 forbids it
    const bool want8 = tail_8ph > 0 || (tail_8ph == 0 && ((d->tile_hint >> kHintWideShift) & 3) == kWide8);
    if (want8 && d->Cout == 256 && tile_valid(d, g, 256, 256, 1, kWide8)) {
      p.npt = ceil_div(p.M, 256);
      p.nct = 1;
      p.kc = g.kc >> 1;
      p.nk = g.nk >> 1;
      if ((long long)p.npt * p.nph > 0x7fffffffLL) return FT_ERR_UNSUPPORTED;
      p.y_bytes = 0;     // (the tail writes through plain pointers)
      p.w_bytes = (unsigned)((unsigned long long)g.nphases * g.cout_pad * g.kpad * 2);
      const int rc8 = launch_igemm8(p, (unsigned)(p.npt * p.nph), s);
      if (rc8 != FT_OK) return rc8;
      FT_LAUNCH_CHECK("conv_igemm8_kernel (tail)");
      return FT_OK;
    }
    static const bool tail_lo_off = getenv("FT_TAIL_LO") && atoi(getenv("FT_TAIL_LO")) == 0;   // dev A/B: hi weights only
    if (tail_lo_off) p.dbg |= 128;
    p.npt = ceil_div(p.M, big ? 256 : 128);
    p.nct = 1;
    if ((long long)p.npt * p.nph > 0x7fffffffLL) return FT_ERR_UNSUPPORTED;
    dim3 grid(p.npt * p.nph);
    if (big) {
      p.kc = g.kc >> 1;
      p.nk = g.nk >> 1;
    }
    const int rc = big ? launch_dma<half_t, 256, 256, 4, 2, 1, 128, 2>(p, grid, s)
                   : d->Cout == 256 ? launch_dma<half_t, 128, 256, 2, 4>(p, grid, s)
                   : d->Cout == 128 ? launch_dma<half_t, 128, 128, 2, 2>(p, grid, s) : launch_dma<half_t, 128, 64, 2, 2>(p, grid, s);
    if (rc != FT_OK) return rc;
    FT_LAUNCH_CHECK("conv_igemm_dma_kernel (tail)");
    return FT_OK;
  }
  if (g.dma && x_bytes < (1ull << 31)) {
    p.x_bytes = (unsigned)x_bytes;
    // tile choice (launch-time only; the packed layout does not depend on it): fill the 256 CUs
    int bc = d->Cout % 128 == 0 ? 128 : 64;
    int bp = 128;
    auto blocks = [&](int bp_, int bc_) { return (long long)ceil_div(p.M, bp_) * (g.cout_pad / bc_) * g.nphases; };
    // The K-loop is bound by operand delivery L2 -> LDS (~10 TB/s chip-wide measured, same as the best GEMMs of
    // the hardware guide): bytes per MAC = (1/BP + 1/BC) * esz, so the largest tile that still fills the chip wins.
    if (d->dtype == FT_F16 && bc == 128 && g.ntaps * g.cin_pad > 512 && blocks(256, 128) >= 2 * 256) bp = 256;
    if (blocks(bp, bc) < 384) bp = 64;
    if (blocks(bp, bc) < 384 && bc == 128) bc = 64;
    // short-K layers (1x1 bottleneck exits) are HBM-bound: smaller pixel tiles = more workgroups per CU =
    // more bytes in flight (measured on MI355X, R50 shapes: 64x128 beats 128x128 by 10-18 % for K <= 512)
    if (g.ntaps * g.cin_pad <= 512 && bc == 128) bp = 64;
    // intra-workgroup split-K (fp16): few tiles + long K => take the missing waves from K, as long as every
    // workgroup still fits on the chip in ONE round (each K-group brings its own LDS ring)
    int ks = 1;
    if (d->dtype == FT_F16 && bp <= 128 && !(bp == 128 && bc == 64)) {
      const long long nblk = blocks(bp, bc);
      const long long per_cu = (nblk + 255) / 256;
      const size_t ring = (size_t)kDmaStages * (bc + bp) * kDmaBKB;
      for (int cand = 4; cand >= 2; cand >>= 1) {
        if (cand == 4 && bp == 128) continue;
        if (per_cu * (cand * ring + 2048) <= 160 * 1024 && per_cu * cand * 4 <= 32 && g.nk >= 8 * cand && nblk <= 768) {
          ks = cand;
          break;
        }
      }
    }
    // explicit choice: the caller's benchmarked hint, or the developer override from the environment
    static const int force = (env_int("FT_CONV_BP") & 0xfff) | ((env_int("FT_CONV_BC") & 0xfff) << 12) | (env_int("FT_CONV_KS") << 24);
    // few workgroups + long K: the K-loop is latency-bound (one barrier per step, <= 2 waves per SIMD), so take
    // 128 bytes of K per step instead of splitting K (measured in situ on R50 / FlowNet2S: 15-25 % on those layers)
    int wide = 0, sk = 1;
    bool halo = false;
    if (d->dtype == FT_F16 && bp <= 128 && g.kc % 2 == 0 && g.ntaps * g.cin_pad >= 512 && blocks(bp, bc) <= 768) {
      wide = 1;
      ks = 1;
    }
    if (g.kc2 > 0) {   // K-concat: lean loop only
      ks = 1;
      if (wide && !tile_valid(d, g, bp, bc, 1, wide)) wide = 0;
    }
    static const int force_wide = env_int("FT_CONV_WIDE"), force_halo = env_int("FT_CONV_HALO");
    const int hint = force ? (force | ((force_wide & 3) << kHintWideShift) | (force_halo ? kHintHalo : 0)) : d->tile_hint;
    if (hint) {
      const int hbp = hint & 0xfff, hbc = (hint >> 12) & 0x1ff, hks = (hint >> 24) & 0xf;
      const int hwide = (hint >> kHintWideShift) & 3;
      const bool hhalo = (hint & kHintHalo) != 0;
      const int hsk = hint_sk((hint >> kHintSkShift) & 7);
      const int nbp = hbp ? hbp : bp, nbc = hbc ? hbc : bc, nks = hks ? hks : (hbp || hbc ? 1 : ks);
      if (tile_valid(d, g, nbp, nbc, nks, hwide, hhalo, hsk)) { bp = nbp; bc = nbc; ks = nks; wide = hwide; halo = hhalo; sk = hsk; }
    }
    if (sk > 1 && (!workspace || workspace_bytes < (size_t)sk * g.nphases * p.M * g.cout_pad * sizeof(float))) sk = 1;
    if (halo) return launch_halo(p, d, g, bc, s);
    if (wide == kWide8) {
      p.kc = g.kc >> 1;
      p.nk = g.nk >> 1;
      p.npt = ceil_div(p.M, 256);
      p.nct = g.cout_pad / 256;
      if ((long long)p.npt * p.nct * p.nph * sk > 0x7fffffffLL) return FT_ERR_UNSUPPORTED;
      p.y_bytes = (unsigned)((unsigned long long)d->N * d->Ho * d->Wo * d->y_cstride * 2);
      p.w_bytes = (unsigned)((unsigned long long)g.nphases * g.cout_pad * g.kpad * 2);
      if (sk > 1) {
        p.sk = sk;
        p.ws = static_cast<float*>(workspace);
      }
      const int rc8 = launch_igemm8(p, (unsigned)(p.npt * p.nct * p.nph * sk), s);
      if (rc8 != FT_OK) return rc8;
      FT_LAUNCH_CHECK("conv_igemm8_kernel");
      if (sk > 1) {
        const size_t total = (size_t)p.nph * p.M * (p.Cout_pad / 4);
        const unsigned rgrid = (unsigned)(total / 256 + 1 > 4096 ? 4096 : total / 256 + 1);
        hipLaunchKernelGGL(conv_splitk_reduce_kernel<half_t>, dim3(rgrid), dim3(256), 0, s, p);
        FT_LAUNCH_CHECK("conv_splitk_reduce_kernel");
      }
      return FT_OK;
    }
    if (wide) {
      p.kc = g.kc >> wide;
      p.kc2 = g.kc2 >> wide;
      p.nk = g.nk >> wide;
    }
    p.npt = ceil_div(p.M, bp);
    p.nct = g.cout_pad / bc;
    if ((long long)p.npt * p.nct * p.nph * sk > 0x7fffffffLL) return FT_ERR_UNSUPPORTED;
    dim3 grid(p.npt * p.nct * p.nph * sk);
    const char* const res_saved = p.res;
    if (sk > 1) {          // partial tiles only: the residual belongs to the reduce launch
      p.sk = sk;
      p.ws = static_cast<float*>(workspace);
      p.res = nullptr;
    }
    int rc;
    if (bc == 256) {
      if (wide == 1 && bp == 256) rc = launch_dma<half_t, 256, 256, 4, 2, 1, 128, 2>(p, grid, s);
      else if (wide == 1) rc = bp == 128 ? launch_dma<half_t, 128, 256, 2, 4, 1, 128, 2>(p, grid, s) : launch_dma<half_t, 64, 256, 1, 4, 1, 128, 2>(p, grid, s);
      else rc = bp == 128 ? launch_dma<half_t, 128, 256, 2, 4>(p, grid, s) : launch_dma<half_t, 64, 256, 1, 4>(p, grid, s);
    } else if (wide == 1) {
      if (bp == 256) rc = launch_dma<half_t, 256, 128, 4, 2, 1, 128, 2>(p, grid, s);
      else if (bp == 128 && bc == 128) rc = launch_dma<half_t, 128, 128, 2, 2, 1, 128, 2>(p, grid, s);
      else if (bp == 128) rc = launch_dma<half_t, 128, 64, 2, 2, 1, 128, 3>(p, grid, s);
      else if (bc == 128) rc = launch_dma<half_t, 64, 128, 2, 2, 1, 128, 3>(p, grid, s);
      else rc = launch_dma<half_t, 64, 64, 2, 2, 1, 128, 3>(p, grid, s);
    } else if (d->dtype == FT_F16) {
      if (bp == 256) rc = launch_dma<half_t, 256, 128, 4, 2>(p, grid, s);
      else if (bp == 128 && bc == 128)
        rc = ks == 2 ? launch_dma<half_t, 128, 128, 2, 2, 2>(p, grid, s) : launch_dma<half_t, 128, 128, 2, 2>(p, grid, s);
      else if (bp == 128) rc = launch_dma<half_t, 128, 64, 2, 2>(p, grid, s);
      else if (bc == 128)
        rc = ks == 4   ? launch_dma<half_t, 64, 128, 2, 2, 4>(p, grid, s)
             : ks == 2 ? launch_dma<half_t, 64, 128, 2, 2, 2>(p, grid, s)
                       : launch_dma<half_t, 64, 128, 2, 2>(p, grid, s);
      else
        rc = ks == 4   ? launch_dma<half_t, 64, 64, 2, 2, 4>(p, grid, s)
             : ks == 2 ? launch_dma<half_t, 64, 64, 2, 2, 2>(p, grid, s)
                       : launch_dma<half_t, 64, 64, 2, 2>(p, grid, s);
    } else {
      if (bp == 128 && bc == 128) rc = launch_dma<float, 128, 128, 2, 2>(p, grid, s);
      else if (bp == 128) rc = launch_dma<float, 128, 64, 2, 2>(p, grid, s);
      else if (bc == 128) rc = launch_dma<float, 64, 128, 2, 2>(p, grid, s);
      else rc = launch_dma<float, 64, 64, 2, 2>(p, grid, s);
    }
    if (rc != FT_OK) return rc;
    FT_LAUNCH_CHECK("conv_igemm_dma_kernel");
    if (sk > 1) {
      p.res = res_saved;
      const size_t total = (size_t)p.nph * p.M * (p.Cout_pad / 4);
      const unsigned rgrid = (unsigned)(total / 256 + 1 > 4096 ? 4096 : total / 256 + 1);
      if (d->dtype == FT_F16) hipLaunchKernelGGL(conv_splitk_reduce_kernel<half_t>, dim3(rgrid), dim3(256), 0, s, p);
      else hipLaunchKernelGGL(conv_splitk_reduce_kernel<float>, dim3(rgrid), dim3(256), 0, s, p);
      FT_LAUNCH_CHECK("conv_splitk_reduce_kernel");
    }
    return FT_OK;
  }
  if (g.dma) return FT_ERR_UNSUPPORTED;  // packed for the dma layout but the activation buffer is >= 2 GiB

  static const bool no_pflow = getenv("FT_CONV_NO_PFLOW") != nullptr;   // dev: A/B against the few-output kernel
  if (!no_pflow && !d->transposed && d->Cout <= 2 && d->dtype == FT_F16 && d->kh == 3 && d->kw == 3 && d->stride == 1 &&
      d->pad == 1 && !d->has_residual && d->x_wpitch == 0 && d->x_cstride % 8 == 0 && d->x_coff % 8 == 0 && x_bytes < (1ull << 31)) {
    p.x_bytes = (unsigned)x_bytes;
    {
      // the matrix-pipe form (12 x 16 output tiles, input streamed once): FT_CONV_PFLOW_MFMA=0 keeps the dot-product kernel
      static const bool no_mfma = getenv("FT_CONV_PFLOW_MFMA") && atoi(getenv("FT_CONV_PFLOW_MFMA")) == 0;
      const int nchunks = ceil_div(g.cin_groups, 4);                 // 32-channel chunks
      const int ty = ceil_div(d->Ho, 12), tx = ceil_div(d->Wo, 16);
      const size_t lds = (size_t)4 * 16384 + (size_t)nchunks * 2048;
      if (!no_mfma && (long long)d->N * ty * tx >= 256 && lds <= 160 * 1024 &&
          (long long)ty * 12 * tx * 16 * 100 <= (long long)d->Ho * d->Wo * 125) {     // ragged edges waste < 25 % of the tiles
        p.h_ty = ty;
        p.h_tx = tx;
        p.npt = d->N * ty * tx;
        auto k = conv_pflow_mfma_kernel;
        if (lds > 64 * 1024) FT_RAISE_LDS(k, 160 * 1024);
        hipLaunchKernelGGL(k, dim3(p.npt), dim3(256), lds, s, p, nchunks);
        FT_LAUNCH_CHECK("conv_pflow_mfma_kernel");
        return FT_OK;
      }
    }
    const long long t16 = (long long)ceil_div(d->Ho, 8) * ceil_div(d->Wo, 16), t8 = (long long)ceil_div(d->Ho, 16) * ceil_div(d->Wo, 8);
    const int tw = t16 <= t8 ? 16 : 8, th = 128 / tw;
    p.h_ty = ceil_div(d->Ho, th);
    p.h_tx = ceil_div(d->Wo, tw);
    p.npt = d->N * p.h_ty * p.h_tx;
    // one patch per workgroup walks ALL channels serially: only worth it with at least one patch per CU (measured on
    // FlowNet2S: predict_flow2/3 75 -> 49 us, 37 -> 28 us; the deep low-resolution predict_flow4..6 stay on the
    // few-output kernel below, which splits K over the lanes)
    if (p.npt >= 256) {
      const size_t lds = 2 * (size_t)6 * 4 * 1024 + 2 * 9 * 2 * 128 + 128 * 2 * 4;
      dim3 grid(p.npt);
      if (tw == 16) hipLaunchKernelGGL(conv_pflow_kernel<16>, grid, dim3(256), lds, s, p);
      else hipLaunchKernelGGL(conv_pflow_kernel<8>, grid, dim3(256), lds, s, p);
      FT_LAUNCH_CHECK("conv_pflow_kernel");
      return FT_OK;
    }
  }
  if (!d->transposed && d->Cout <= 4) {
    const int nco = d->Cout <= 2 ? 2 : 4;
    if (x_bytes < (1ull << 31) && g.cout_pad >= nco) {
      p.x_bytes = (unsigned)x_bytes;
      // pixels per workgroup: enough workgroups to fill the chip; a wave takes PIX = 8 / nco pixels per pass
      int ppb = ceil_div(p.M, 2048);
      ppb = ppb < 16 ? 16 : (ppb > 128 ? 128 : ppb);
      ppb = round_up(ppb, 16);
      dim3 grid(ceil_div(p.M, ppb));
#define FT_FEWOUT(T, N, P)                                          \
  do {                                                              \
    auto k = conv_fewout_kernel<T, N, P>;                           \
    hipLaunchKernelGGL(k, grid, dim3(256), 0, s, p, ppb);           \
  } while (0)
      if (d->dtype == FT_F16) { if (nco == 2) FT_FEWOUT(half_t, 2, 4); else FT_FEWOUT(half_t, 4, 2); }
      else { if (nco == 2) FT_FEWOUT(float, 2, 4); else FT_FEWOUT(float, 4, 2); }
#undef FT_FEWOUT
      FT_LAUNCH_CHECK("conv_fewout_kernel");
      return FT_OK;
    }
  }

  const int bc = g.cout_pad % 128 == 0 && d->Cout > 64 ? 128 : (d->Cout <= 32 ? 32 : 64);
  p.npt = ceil_div(p.M, kBP);
  p.nct = g.cout_pad / bc;
  if ((long long)p.npt * p.nct * p.nph > 0x7fffffffLL) return FT_ERR_UNSUPPORTED;
  dim3 grid(p.npt * p.nct * p.nph);
  if (d->dtype == FT_F16) {
    if (bc == 128) launch_generic<half_t, 128, 2, 2>(p, grid, s);
    else if (bc == 64) launch_generic<half_t, 64, 2, 2>(p, grid, s);
    else launch_generic<half_t, 32, 4, 1>(p, grid, s);
  } else {
    if (bc == 128) launch_generic<float, 128, 2, 2>(p, grid, s);
    else if (bc == 64) launch_generic<float, 64, 2, 2>(p, grid, s);
    else launch_generic<float, 32, 4, 1>(p, grid, s);
  }
  FT_LAUNCH_CHECK("conv_igemm_kernel");
  return FT_OK;
}
