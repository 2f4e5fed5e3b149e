// 256 x 256 implicit-GEMM conv tile on an 8-phase ping-pong schedule (fp16, channel-aligned layers with Cin % 64 == 0 and
// Cout % 256 == 0): the ConvTranspose2d(4,2,1) head of the pose net (lib/pose/models/pose_deconv.py:19-30), the stride-2
// 3x3 entry convs of the ResNet stages (lib/pose/models/blocks.py:95-97) and FlowNet's wide convs / deconvs
// (lib/flownet/networks/FlowNetS.py:24-45).  Same math, same packed weights, same epilogue as conv_igemm_dma_kernel
// (conv_common.h); what differs is the K-loop.
//
// Why a second K-loop.  With 64 x 64 wave tiles (the 8-wave 128 x 256 form of conv_igemm_dma_kernel) every MFMA needs 1 KiB
// of LDS fragment reads — the CU's whole 128 B/clk at full matrix rate — and one barrier per K-step puts every wave of the
// workgroup in the same phase at the same time (profiles/README.md, "What bounds the kernels").  Here
//   * the wave tile is 128 output channels x 64 pixels: 24 ds_read_b128 per 32 MFMAs (768 B per MFMA);
//   * the eight waves are two groups of four (one wave of each group per SIMD) that run ONE BARRIER APART: while group 0
//     is in its MFMA segment group 1 issues its LDS reads and the LDS-DMA of the next operands, and vice versa — the matrix
//     pipe of a SIMD always has one wave feeding it;
//   * a K-tile (64 channels of one tap: 128-byte rows) is four 16-KiB half-tiles (W0, W1: 128 weight rows each; P0, P1: 128
//     pixel rows each), double-buffered (128 KiB).  A wave's rows are split over BOTH halves of an operand (weight tiles
//     0-1 in W0, 2-3 in W1; pixel tile 0 in P0, 1 in P1), so each of the four phases of a K-tile finishes one half-tile and
//     the next phases can refill it while the K-tile is still being multiplied:
//         phase 1: read P0 (4 x b128) + W0 (8)   MFMA W0 x P0   stage W1 of K-tile t+1
//         phase 2: read P1 (4)                   MFMA W0 x P1   stage P0 of K-tile t+2   (P0 reads retired before the barrier)
//         phase 3: read W1 (8)                   MFMA W1 x P1   stage W0 of K-tile t+2
//         phase 4: -                             MFMA W1 x P0   stage P1 of K-tile t+2, s_waitcnt vmcnt(6): K-tile t+1 landed
//     Three half-tiles stay in flight across every barrier; a half-tile is read no earlier than one phase after the wait
//     that retired it and refilled no earlier than two phases after its last read (one phase for P0, whose reads are
//     retired by lgkmcnt before the reading phase's first barrier) — the rules of the hardware guide's 8-phase GEMM
//     template (cdna_hip_programming.md §5, "The 256² 8-phase template"), applied to implicit-GEMM operand addressing.
//   * LDS rows are 128 bytes, 16-byte chunk c of row r sits at chunk c ^ (r / 2 % 8): conflict-free ds_read_b128; the
//     permutation is applied to the DMA's per-lane SOURCE address (the LDS side of buffer_load ... lds is lane-linear).
// Padding taps, ragged pixel tiles and K-tiles past the end are out-of-range buffer offsets (zeros).
#include "conv_common.h"

namespace ft {

namespace {
constexpr int kT8 = 256;            // tile edge (pixels and output channels)
constexpr int kRowB = 128;          // bytes of K per tile row per K-tile (64 fp16 channels)
constexpr int kHalfB = 128 * kRowB; // one half-tile: 16 KiB
constexpr int kDbufB = 4 * kHalfB;  // W0 W1 P0 P1
constexpr int kRingB = 2 * kDbufB;  // 128 KiB
}  // namespace

__global__ __launch_bounds__(512, 2) void conv_igemm8_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  constexpr unsigned kOOB = 0x80000000u;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
  const int wp = wave & 3, wc = wave >> 2;                     // pixel quarter / channel half; wc is also the ping-pong group

  int ctile, phase, ptile, ksplit;
  {
    const int tiles = p.npt * p.nct * p.nph;
    const int total = tiles * p.sk;
    const int b = blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = b & 7, loc = b >> 3;
    int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    ksplit = logical / tiles;
    logical -= ksplit * tiles;
    ctile = logical % p.nct;
    const int t = logical / p.nct;
    phase = t % p.nph;
    ptile = t / p.nph;
  }
  const int py = phase >> 1, px = phase & 1;
  const int m0 = ptile * kT8;
  const int co0 = ctile * kT8;
  const int dbase_y = p.transposed ? py : -p.pad;
  const int dbase_x = p.transposed ? px : -p.pad_x;
  // K-tiles of this workgroup: [kt_lo, kt_hi) of p.nk (p.sk > 1: the ksplit-th slice)
  const int nk_sk = (p.nk + p.sk - 1) / p.sk;
  const int kt_lo = ksplit * nk_sk;
  const int kt_hi = kt_lo + nk_sk < p.nk ? kt_lo + nk_sk : p.nk;
  const int nkt = kt_hi > kt_lo ? kt_hi - kt_lo : 0;

  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(p.w) + (size_t)(phase * p.Cout_pad + co0) * p.Kpad * 2, 0, kT8 * p.Kpad * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);

  // ---- loader constants.  One DMA instruction of the workgroup fills 64 rows x 128 B; a half-tile is two (u = 0, 1).
  // Row-in-half rho = u * 64 + wave * 8 + lane / 8; LDS chunk position lane % 8 holds source chunk pos ^ (rho / 2 % 8).
  const int lrow = lane >> 3, pos = lane & 7;
  const int lc16 = (pos ^ (((wave & 1) << 2) | (lrow >> 1))) << 4;
  // weights: logical row of (half h, u) = u * 128 + h * 64 + wave * 8 + lrow; the (h, u) part rides in the scalar offset
  const int krow = p.Kpad * 2;
  const unsigned w_voff = (p.dbg & 256) ? kOOB : (unsigned)((wave * 8 + lrow) * krow + lc16);
  // pixels: logical row of (half h, u) = (u * 2 + wave / 4) * 64 + h * 32 + (wave & 3) * 8 + lrow
  int b_base[2][2];
  unsigned b_mask[2][2];
  const int cstride_b = p.x_cstride * 2;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = m0 + (u * 2 + (wave >> 2)) * 64 + h * 32 + (wave & 3) * 8 + lrow;
      unsigned mask = 0;
      int base = 0;
      if (m < p.M) {
        const int n = m / p.HqWq;
        const int rem = m - n * p.HqWq;
        const int qy = rem / p.Wq;
        const int qx = rem - qy * p.Wq;
        const int iy0 = qy * p.sy + dbase_y, ix0 = qx * p.sy + dbase_x;
        base = ((n * p.Hi + iy0) * p.Wi + ix0) * cstride_b + p.x_coff * 2 + lc16;
        for (int ky = 0; ky < p.kh; ++ky)
          for (int kx = 0; kx < p.kw; ++kx) {
            const int iy = iy0 + p.dmul * ky, ix = ix0 + p.dmul * kx;
            if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi) mask |= 1u << (ky * p.kw + kx);
          }
      }
      if (p.dbg & 64) mask = 0;
      b_base[h][u] = base;
      b_mask[h][u] = mask;
    }

  // ---- staging state: K-tile `s_kt` is the one whose P0 / W0 / P1 are staged next (its W1 follows one K-tile later)
  int s_kt = kt_lo, s_cc, s_ky, s_kx;
  {
    const int tap0 = kt_lo / p.kc;
    s_cc = kt_lo - tap0 * p.kc;
    s_ky = tap0 / p.kw;
    s_kx = tap0 - s_ky * p.kw;
  }
  unsigned cur_voff[2][2];
  auto refresh = [&]() {
    const bool live = s_kt < kt_hi;
    const int tap = s_ky * p.kw + s_kx;
    const int delta = ((p.dmul * s_ky) * p.Wi + p.dmul * s_kx) * cstride_b;
    const unsigned tapbit = live ? (1u << tap) : 0u;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int u = 0; u < 2; ++u) cur_voff[h][u] = (b_mask[h][u] & tapbit) ? (unsigned)(b_base[h][u] + delta) : kOOB;
  };
  refresh();
  auto advance = [&]() {
    ++s_kt;
    bool changed = s_kt == kt_hi;          // past the end: every pixel offset goes out of range
    if (++s_cc == p.kc) {
      s_cc = 0;
      if (++s_kx == p.kw) { s_kx = 0; ++s_ky; }
      changed = true;
    }
    if (changed) refresh();
  };
  // stage weight half h of K-tile kt into double buffer D
  auto stage_w = [&](auto Dc, auto hc, int kt) {
    constexpr int D = decltype(Dc)::value, h = decltype(hc)::value;
    const unsigned v = kt < kt_hi ? w_voff : kOOB;
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr)(smem + D * kDbufB + h * kHalfB + u * 8192 + wave * 1024), 16, v,
                                               kt * kRowB + (u * 128 + h * 64) * krow, 0, 0);
  };
  // stage pixel half h of the K-tile the staging state points at
  auto stage_p = [&](auto Dc, auto hc) {
    constexpr int D = decltype(Dc)::value, h = decltype(hc)::value;
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr)(smem + D * kDbufB + (2 + h) * kHalfB + u * 8192 + wave * 1024), 16,
                                               cur_voff[h][u], s_cc * kRowB, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // ---- fragment read offsets: row rho of a half-tile, chunk (2 kk + lhi) ^ (rho / 2 % 8); kk rides in an XOR of (kk << 5)
  const int l31 = lane & 31, lhi = lane >> 5;
  const int fkey = ((lhi ^ ((l31 >> 1) & 7)) << 4);
  const int p_off = (wp * 32 + l31) * kRowB + fkey;              // inside P0 / P1
  const int w_off = (wc * 64 + l31) * kRowB + fkey;              // weight tile 0 / 2 of the wave inside W0 / W1; tile 1 / 3 = + 32 rows

  float16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4_t wf[2][4], pf0[4], pf1[4];

  // ---- prologue: K-tile 0 whole, K-tile 1 without its W1 (phase 1 of K-tile 0 stages that) ---------------------------
  stage_p(I0{}, I0{});
  stage_w(I0{}, I0{}, s_kt);
  stage_p(I0{}, I1{});
  stage_w(I0{}, I1{}, s_kt);
  advance();
  stage_p(I1{}, I0{});
  stage_w(I1{}, I0{}, s_kt);
  stage_p(I1{}, I1{});
  advance();                                  // s_kt = kt_lo + 2
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  FT_LDS_BARRIER();
  if (wc == 1) FT_LDS_BARRIER();              // group 1 runs one barrier behind group 0 from here on

#define FT8_MFMA(i, j, W, P)                                                                                          \
  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, W), __builtin_bit_cast(half8_t, P), \
                                                     acc[i][j], 0, 0, 0)

  // one K-tile = four phases on double buffer D
  auto ktile = [&](auto Dc) {
    constexpr int D = decltype(Dc)::value;
    using DC = std::integral_constant<int, D>;
    using DN = std::integral_constant<int, D ^ 1>;
    const char* const base = smem + D * kDbufB;
    // ---- phase 1
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) pf0[kk] = *reinterpret_cast<const uint4_t*>(base + 2 * kHalfB + (p_off ^ (kk << 5)));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) wf[ii][kk] = *reinterpret_cast<const uint4_t*>(base + ii * 32 * kRowB + (w_off ^ (kk << 5)));
    __builtin_amdgcn_sched_barrier(0);
    stage_w(DN{}, I1{}, s_kt - 1);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");    // this wave's P0 reads are done: P0 may be refilled after the barrier
    FT_LDS_BARRIER();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      FT8_MFMA(0, 0, wf[0][kk], pf0[kk]);
      FT8_MFMA(1, 0, wf[1][kk], pf0[kk]);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    FT_LDS_BARRIER();
    // ---- phase 2
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) pf1[kk] = *reinterpret_cast<const uint4_t*>(base + 3 * kHalfB + (p_off ^ (kk << 5)));
    __builtin_amdgcn_sched_barrier(0);
    stage_p(DC{}, I0{});
    FT_LDS_BARRIER();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      FT8_MFMA(0, 1, wf[0][kk], pf1[kk]);
      FT8_MFMA(1, 1, wf[1][kk], pf1[kk]);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    FT_LDS_BARRIER();
    // ---- phase 3
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        wf[ii][kk] = *reinterpret_cast<const uint4_t*>(base + kHalfB + ii * 32 * kRowB + (w_off ^ (kk << 5)));
    __builtin_amdgcn_sched_barrier(0);
    stage_w(DC{}, I0{}, s_kt);
    FT_LDS_BARRIER();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      FT8_MFMA(2, 1, wf[0][kk], pf1[kk]);
      FT8_MFMA(3, 1, wf[1][kk], pf1[kk]);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    FT_LDS_BARRIER();
    // ---- phase 4
    stage_p(DC{}, I1{});
    advance();
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // everything of the next K-tile has landed (three half-tiles stay in flight)
    FT_LDS_BARRIER();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      FT8_MFMA(2, 0, wf[0][kk], pf0[kk]);
      FT8_MFMA(3, 0, wf[1][kk], pf0[kk]);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    FT_LDS_BARRIER();
  };
#undef FT8_MFMA

  int t = 0;
  for (; t + 2 <= nkt; t += 2) {
    ktile(I0{});
    ktile(I1{});
  }
  if (t < nkt) ktile(I0{});
  if (wc == 0) FT_LDS_BARRIER();              // group 0 catches up with group 1's extra barrier
  // the loads still in flight are the out-of-range tail stages: let them land (as zeros) before LDS changes hands
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (p.dbg & 4) {
    if (acc[0][0][0] == 12345.678f) p.y[0] = 1;
    return;
  }
  if (p.sk > 1) {
    // cross-workgroup split-K: raw fp32 partial tile -> workspace [ksplit][phase * M + pixel][Cout_pad]; the scale / shift /
    // residual / activation epilogue runs in conv_splitk_reduce_kernel once every slice has landed
    float* wsb = p.ws + ((size_t)ksplit * p.nph + phase) * (size_t)p.M * p.Cout_pad;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m0 + wp * 64 + j * 32 + l31;
      if (m >= p.M) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int cb = co0 + wc * 128 + i * 32 + 8 * rg + 4 * lhi;
          const float4_t v = {acc[i][j][rg * 4], acc[i][j][rg * 4 + 1], acc[i][j][rg * 4 + 2], acc[i][j][rg * 4 + 3]};
          *reinterpret_cast<float4_t*>(wsb + (size_t)m * p.Cout_pad + cb) = v;
        }
    }
    return;
  }
  conv_epilogue<half_t, kT8, kT8, 4, 2, false, 512>(p, acc, smem, kRingB, m0, co0, py, px);
#endif
}

int launch_igemm8(const ConvParams& p, unsigned grid, hipStream_t s) {
  constexpr size_t lds = (size_t)kRingB + (size_t)kT8 * 8;
  FT_RAISE_LDS(conv_igemm8_kernel, lds);
  hipLaunchKernelGGL(conv_igemm8_kernel, dim3(grid), dim3(512), lds, s, p);
  return FT_OK;
}

}  // namespace ft
