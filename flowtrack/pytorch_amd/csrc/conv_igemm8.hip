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
//   * ONE workgroup per CU (128 KiB ring, 8 waves x ~230 registers) means nothing else on the CU can cover a tile's prologue
//     (first operands from L2 / HBM: ~2 us) or epilogue (measured with the shared LDS-transposing epilogue: 40 of 131 us on the
//     pose head's last deconv, the chip's matrix pipes idle in lock-step three times per launch).  So the kernel is
//     PERSISTENT — one workgroup per CU walks its XCD's share of the tiles — and the tile boundary is software-pipelined:
//     after the last MFMA the NEXT tile's first seven half-tiles are put in flight, THEN the finished tile leaves straight
//     from the accumulator registers (no LDS, no barrier): folded BN / bias + activation, lane pairs exchange halves with
//     v_permlane32_swap so every lane stores 16 bytes.  The fused tail 1x1 conv (heat maps) also runs from registers: the
//     activated accumulators ARE the MFMA B operand of a [32 x 128] x [128 x 64 pixel] product per wave (the tail weights'
//     K order is permuted to match), the two channel halves of the workgroup meet through a 32-KiB LDS slab.
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

namespace {
// 32 KiB behind the ring: [0, 24 K) MODE 0: eight 2-KiB wave-private transposition buffers / MODE 1: the partial sums the two
// channel halves of the workgroup exchange (8 waves x 12 registers x 64 lanes x 4 B); [24 K, 28 K) folded-BN scale,
// [28 K, 32 K) shift of every output channel (<= 1024), read by the epilogues with ds_read — an ordinary global load there
// would make hipcc drain the LDS-DMA queue with a vmcnt(0) per load (measured: 64 serialized L2 round trips per tile)
constexpr int kSlabB = 32 * 1024;
constexpr int kScaleOff = 24 * 1024, kShiftOff = 28 * 1024;
constexpr int kTailBiasOff = 2048;     // bytes behind kShiftOff: the fused tail's 32 biases (MODE 1)
constexpr int kMaxCout8 = 1024;
typedef uint32_t uint2v_t __attribute__((ext_vector_type(2)));
}  // namespace
namespace {
// Two 16-byte global loads the compiler does not see as loads (it would wait vmcnt(0) for them beside LDS-DMA) and the counted
// wait that hands their results over: the registers pass THROUGH the wait statement, so no use can be scheduled above it.
__device__ __forceinline__ void ld2x16_asm(uint4_t& a, uint4_t& b, unsigned voff, const char* sbase) {
  asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"
               : "=&v"(a), "=&v"(b) : "v"(voff), "s"(sbase) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm_asm(uint4_t& a, uint4_t& b) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
}  // namespace
#ifndef FT8_STORE_AUX
#define FT8_STORE_AUX FT_YSTORE_BUF_AUX   // cache policy of the direct NHWC stores (dev A/B: 0 = plain, 16 = sc1 write-through, 2 = nt)
#endif

// MODE: 0 = NHWC fp16 stores, 1 = fused tail 1x1 conv, 2 = split-K partial tiles (fp32, workspace)
template <int MODE>
__global__ __launch_bounds__(512, 2) void conv_igemm8_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  constexpr unsigned kOOB = 0x80000000u;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
  const int wp = wave & 3, wc = wave >> 2;                     // pixel quarter / channel half; wc is also the ping-pong group
  const int l31 = lane & 31, lhi = lane >> 5;

  // ---- this workgroup's tiles: XCD x owns one contiguous range of the logical tile order (output-channel tile fastest, then
  // transposed-conv phase, then pixel tile, then K slice), its workgroups take every x_slots-th tile of it — so the 32 CUs of
  // an XCD work on 32 neighbouring tiles at any time and meet in one L2
  const int tiles = p.npt * p.nct * p.nph;
  const int total = tiles * p.sk;
  const int xcd = blockIdx.x & 7;
  int local = blockIdx.x >> 3;
  int x_start, x_count, x_slots;
  {
    const int q = total >> 3, r = total & 7;
    x_start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    x_count = q + (xcd < r ? 1 : 0);
    x_slots = ((int)gridDim.x - xcd + 7) >> 3;
  }
  if (local >= x_count) return;

  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  // ONE descriptor over the whole packed weight set; a tile's (phase, channel tile) block is a scalar byte offset — the
  // descriptors never change, so nothing about them is carried around the K-loop (a per-tile descriptor copied into the
  // lagged W1 state was kept in SGPRs that the tap-change code of the same loop overwrote: wrong upper channels now and then)
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.w), 0, p.w_bytes, 0x00020000);
  const int krow = p.Kpad * 2;
  const int cstride_b = p.x_cstride * 2;
  const int nk_sk = (p.nk + p.sk - 1) / p.sk;

  // ---- loader constants that do not depend on the tile.  One DMA instruction of the workgroup fills 64 rows x 128 B; a
  // half-tile is two (u = 0, 1).  Row-in-half rho = u * 64 + wave * 8 + lane / 8; LDS chunk position lane % 8 holds source
  // chunk pos ^ (rho / 2 % 8).
  const int lrow = lane >> 3, pos = lane & 7;
  const int lc16 = (pos ^ (((wave & 1) << 2) | (lrow >> 1))) << 4;
  // weights: logical row of (half h, u) = u * 128 + h * 64 + wave * 8 + lrow; the (h, u) part rides in the scalar offset
  const unsigned w_voff = (p.dbg & 256) ? kOOB : (unsigned)((wave * 8 + lrow) * krow + lc16);

  // ---- folded-BN scale / shift of every output channel -> LDS (1 / 0 where the layer has none), before any DMA is in flight
  {
    float* const ls = reinterpret_cast<float*>(smem + kRingB + kScaleOff);
    float* const lh = reinterpret_cast<float*>(smem + kRingB + kShiftOff);
    for (int c = tid; c < p.Cout_pad; c += 512) {
      ls[c] = p.scale ? p.scale[c] : 1.f;
      lh[c] = p.shift ? p.shift[c] : 0.f;
    }
    // MODE 1: the fused tail's bias as well (Cout <= 256 there: the upper half of the shift table is free).  Read with
    // `bt[co]` in the tail's last lines it was TWELVE ordinary loads per lane per tile, each behind a vmcnt(0) = a serialized
    // L2 round trip that also waits for the next tile's LDS-DMA and for the store before it.
    if constexpr (MODE == 1) {
      if (tid < 32) lh[kTailBiasOff / 4 + tid] = reinterpret_cast<const float*>(p.tail_w + (size_t)32 * kT8 * 2)[tid];
    }
    __syncthreads();
  }

  // ---- staging state: the DMA stream runs two K-tiles ahead of the MFMAs and straight across tile boundaries — before the
  // last two K-tiles of a tile are multiplied the state below is re-made for the workgroup's NEXT tile, so that tile's first
  // operands are in LDS before this tile's last MFMA (no prologue between tiles).
  int s_woff = 0;                                         // byte offset of the staged tile's (phase, channel tile) weight block
  int b_base[2][2];                                       // byte offset of tap (0, 0) of the four pixel rows this lane loads
  unsigned b_yx[2][2];                                    // their (iy0 + 0x4000) | (ix0 + 0x4000) << 16, 0xffffffff past the last pixel
  unsigned cur_voff[2][2];                                // offsets of the current tap (out of range where it falls off the map)
  int kt_hi = 0, s_kt = 0, s_cc = 0, s_ky = 0, s_kx = 0;
  // W1 of a K-tile is staged one K-tile after its P0 / W0 / P1: a lagged copy of what it needs
  int w1_woff = 0, w1_kt = 0;
  bool w1_live = false;
  auto refresh = [&]() __attribute__((always_inline)) {
    const bool live = s_kt < kt_hi;
    const int dy = p.dmul * s_ky, dx = p.dmul * s_kx;
    const int delta = (dy * p.Wi + dx) * cstride_b;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int iy = (int)(b_yx[h][u] & 0xffffu) - 0x4000 + dy, ix = (int)(b_yx[h][u] >> 16) - 0x4000 + dx;
        const bool ok = live && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        cur_voff[h][u] = ok ? (unsigned)(b_base[h][u] + delta) : kOOB;
      }
  };
  // tile `loc` of this XCD's range -> (m0, co0, phase, ksplit, number of K-tiles) and the loader state for its first K-tile
  auto setup = [&](int loc, int& m0, int& co0, int& phase, int& ksplit, int& nkt) __attribute__((always_inline)) {
    int logical = x_start + loc;
    ksplit = logical / tiles;
    logical -= ksplit * tiles;
    const int ctile = logical % p.nct;
    const int t = logical / p.nct;
    phase = t % p.nph;
    m0 = (t / p.nph) * kT8;
    co0 = ctile * kT8;
    const int py = phase >> 1, px = phase & 1;
    const int dbase_y = p.transposed ? py : -p.pad;
    const int dbase_x = p.transposed ? px : -p.pad_x;
    const int kt_lo = ksplit * nk_sk;
    kt_hi = kt_lo + nk_sk < p.nk ? kt_lo + nk_sk : p.nk;
    nkt = kt_hi > kt_lo ? kt_hi - kt_lo : 0;
    s_woff = (phase * p.Cout_pad + co0) * krow;
    // pixels: logical row of (half h, u) = (u * 2 + wave / 4) * 64 + h * 32 + (wave & 3) * 8 + lrow
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int m = m0 + (u * 2 + (wave >> 2)) * 64 + h * 32 + (wave & 3) * 8 + lrow;
        unsigned yx = 0xffffffffu;                        // (decodes to coordinates far outside any map)
        int base = 0;
        if (m < p.M && !(p.dbg & 64)) {
          const int n = m / p.HqWq;
          const int rem = m - n * p.HqWq;
          const int qy = rem / p.Wq;
          const int qx = rem - qy * p.Wq;
          const int iy0 = qy * p.sy + dbase_y, ix0 = qx * p.sy + dbase_x;
          base = ((n * p.Hi + iy0) * p.Wi + ix0) * cstride_b + p.x_coff * 2 + lc16;
          yx = (unsigned)(iy0 + 0x4000) | ((unsigned)(ix0 + 0x4000) << 16);
        }
        b_base[h][u] = base;
        b_yx[h][u] = yx;
      }
    // staging position: K-tile s_kt is the one whose P0 / W0 / P1 are staged next (its W1 follows one K-tile later)
    s_kt = kt_lo;
    const int tap0 = kt_lo / p.kc;
    s_cc = kt_lo - tap0 * p.kc;
    s_ky = tap0 / p.kw;
    s_kx = tap0 - s_ky * p.kw;
    refresh();
  };
  // no tile left: every further stage is out of range (zeros, no traffic)
  auto kill = [&]() __attribute__((always_inline)) {
    kt_hi = s_kt = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int u = 0; u < 2; ++u) cur_voff[h][u] = kOOB;
  };
  // called when P0 / W0 / P1 of K-tile s_kt have been issued: remember what its W1 needs, move to the tile's next K-tile
  auto advance = [&]() __attribute__((always_inline)) {
    w1_woff = s_woff;
    w1_kt = s_kt;
    w1_live = s_kt < kt_hi;
    ++s_kt;
    if (++s_cc == p.kc) {
      s_cc = 0;
      if (++s_kx == p.kw) { s_kx = 0; ++s_ky; }
      refresh();
    }
  };
  // stage weight half 0 of K-tile kt into the double buffer at byte offset dofs (0 or kDbufB)
  auto stage_w0 = [&](int dofs, int kt) __attribute__((always_inline)) {
    const unsigned v = kt < kt_hi ? w_voff : kOOB;
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr)(smem + dofs + u * 8192 + wave * 1024), 16, v,
                                               s_woff + kt * kRowB + (u * 128) * krow, 0, 0);
  };
  // W1 of the K-tile whose other three half-tiles went out one K-tile ago
  auto stage_w1 = [&](int dofs) __attribute__((always_inline)) {
    const unsigned v = w1_live ? w_voff : kOOB;
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr)(smem + dofs + kHalfB + u * 8192 + wave * 1024), 16, v,
                                               w1_woff + w1_kt * kRowB + (u * 128 + 64) * krow, 0, 0);
  };
  // stage pixel half h of the K-tile the staging state points at
  auto stage_p = [&](int dofs, auto hc) __attribute__((always_inline)) {
    constexpr int h = decltype(hc)::value;
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr)(smem + dofs + (2 + h) * kHalfB + u * 8192 + wave * 1024), 16,
                                               cur_voff[h][u], s_cc * kRowB, 0, 0);
  };
  // the first tile's first seven half-tiles: K-tile 0 whole, K-tile 1 without its W1 (phase 1 of K-tile 0 stages that)
  auto prologue = [&]() __attribute__((always_inline)) {
    stage_p(0, I0{});
    stage_w0(0, s_kt);
    stage_p(0, I1{});
    advance();
    stage_w1(0);
    stage_p(kDbufB, I0{});
    stage_w0(kDbufB, s_kt);
    stage_p(kDbufB, I1{});
    advance();
  };

  // ---- fragment read offsets: row rho of a half-tile, chunk (2 kk + lhi) ^ (rho / 2 % 8), one address register per k-slice;
  // they include the double buffer's offset and flip to the other buffer (XOR kDbufB) after every K-tile, so the K-loop body
  // exists once
  const int fkey = ((lhi ^ ((l31 >> 1) & 7)) << 4);
  int rd_p[4], rd_w[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    rd_p[kk] = ((wp * 32 + l31) * kRowB + fkey) ^ (kk << 5);              // inside P0 / P1
    rd_w[kk] = ((wc * 64 + l31) * kRowB + fkey) ^ (kk << 5);              // weight tile 0 / 2 of the wave inside W0 / W1; tile 1 / 3 = + 32 rows
  }
  int dofs = 0;                                                            // byte offset of the double buffer being multiplied

  float16_t acc[4][2];
  uint4_t wf[2][4], pf0[4], pf1[4];

#define FT8_MFMA(i, j, W, P)                                                                                          \
  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, W), __builtin_bit_cast(half8_t, P), \
                                                     acc[i][j], 0, 0, 0)

  // one K-tile = four phases
  auto ktile = [&]() __attribute__((always_inline)) {
    // ---- phase 1
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) pf0[kk] = *reinterpret_cast<const uint4_t*>(smem + rd_p[kk] + 2 * kHalfB);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) wf[ii][kk] = *reinterpret_cast<const uint4_t*>(smem + rd_w[kk] + ii * 32 * kRowB);
    __builtin_amdgcn_sched_barrier(0);
    stage_w1(dofs ^ kDbufB);
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
    for (int kk = 0; kk < 4; ++kk) pf1[kk] = *reinterpret_cast<const uint4_t*>(smem + rd_p[kk] + 3 * kHalfB);
    __builtin_amdgcn_sched_barrier(0);
    stage_p(dofs, I0{});
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
      for (int kk = 0; kk < 4; ++kk) wf[ii][kk] = *reinterpret_cast<const uint4_t*>(smem + rd_w[kk] + kHalfB + ii * 32 * kRowB);
    __builtin_amdgcn_sched_barrier(0);
    stage_w0(dofs, s_kt);
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
    stage_p(dofs, I1{});
    advance();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { rd_p[kk] ^= kDbufB; rd_w[kk] ^= kDbufB; }
    dofs ^= kDbufB;
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

  // ---- results straight from the accumulator registers -------------------------------------------------------------------
  // acc[i][j][4 rg + e] = channel co0 + wc * 128 + i * 32 + 8 rg + 4 lhi + e of pixel m0 + wp * 64 + j * 32 + l31.
  const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  // activation without branches: act(v) = max(v, 0) + s * min(v, 0), s = 0 (ReLU; exact, +0 for negatives), the slope
  // (LeakyReLU; the same single rounding as v * slope) or 1 (none; max + min of one number is the number)
  const float act_s = p.act == FT_ACT_RELU ? 0.f : (p.act == FT_ACT_LEAKY ? p.slope : 1.f);
  auto act1 = [&](float v) __attribute__((always_inline)) { return __builtin_fmaf(act_s, __builtin_fminf(v, 0.f), __builtin_fmaxf(v, 0.f)); };
  // folded BN / bias + activation of the four values of (i, j, rg), packed to fp16
  auto activated = [&](int i, int j, int rg, int cb) __attribute__((always_inline)) -> uint2v_t {
    const float4_t sc = *reinterpret_cast<const float4_t*>(smem + kRingB + kScaleOff + cb * 4);
    const float4_t sh = *reinterpret_cast<const float4_t*>(smem + kRingB + kShiftOff + cb * 4);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = act1(acc[i][j][rg * 4 + e] * sc[e] + sh[e]);
    const half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    return __builtin_bit_cast(uint2v_t, h);
  };
  // output pixel (element index into the NHWC / NCHW map) of pixel-tile row j of this lane, or -1 past the end
  auto out_pixel = [&](int m0, int py, int px, int j, int l31, int& n_out, int& pix_out) __attribute__((always_inline)) -> bool {
    const int m = m0 + wp * 64 + j * 32 + l31;
    if (m >= p.M) return false;
    const int n = m / p.HqWq;
    const int rem = m - n * p.HqWq;
    const int qy = rem / p.Wq;
    const int qx = rem - qy * p.Wq;
    n_out = n;
    pix_out = (qy * p.omul + py) * p.Wo + (qx * p.omul + px);
    return true;
  };
  auto epilogue_store = [&](int m0, int co0, int phase) __attribute__((always_inline)) {
    // NHWC fp16.  In the accumulator layout a lane owns one pixel and 4-channel runs: stored directly, every instruction
    // would touch 32 pixel rows with 16-32 bytes each (measured: slower than the MFMAs it follows).  So each wave turns its
    // 64 pixel x 128 channel tile around in a PRIVATE 2-KiB piece of LDS, 32 pixels x 32 channels at a time (64-byte rows,
    // chunk c of row r at c ^ (r / 2 % 4); a wave's LDS operations execute in issue order, no barrier and no wait is
    // involved), and stores 16 bytes per lane with 4 consecutive lanes covering the 64 contiguous bytes of one pixel.
    const int py = phase >> 1, px = phase & 1;
    char* const tb = smem + kRingB + wave * 2048;
    const int rrow = lane >> 2, rpos = lane & 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      unsigned vrow[2];
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int row = ps * 16 + rrow;
        const int m = m0 + wp * 64 + j * 32 + row;
        unsigned v = kOOB;
        if (m < p.M) {
          const int n = m / p.HqWq;
          const int rem = m - n * p.HqWq;
          const int qy = rem / p.Wq;
          const int qx = rem - qy * p.Wq;
          const int pix = (n * p.Ho + qy * p.omul + py) * p.Wo + (qx * p.omul + px);
          v = (unsigned)((pix * p.y_cstride + p.y_coff + co0 + wc * 128 + ((rpos ^ ((row >> 1) & 3)) << 3)) * 2);
        }
        vrow[ps] = v;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const uint2v_t h = activated(i, j, rg, co0 + wc * 128 + i * 32 + 8 * rg + 4 * lhi);
          *reinterpret_cast<uint2v_t*>(tb + l31 * 64 + ((rg ^ ((l31 >> 1) & 3)) << 4) + lhi * 8) = h;
        }
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const int row = ps * 16 + rrow;
          const uint4_t o = *reinterpret_cast<const uint4_t*>(tb + row * 64 + rpos * 16);
          const int cchunk = co0 + wc * 128 + i * 32 + ((rpos ^ ((row >> 1) & 3)) << 3);
          // The channel offset of the pass rides in the VECTOR offset, the scalar offset stays the literal 0.  With an SGPR
          // there, hipcc (ROCm 7.2) treats a 16-byte buffer store as free of the "VALU overwrites store data" hazard and
          // schedules the next address arithmetic into the data registers right behind it; gfx950 does have the hazard:
          // measured as ~3 % of the pixels of a pass with a register offset carrying the NEXT pass's address temporaries
          // instead of results, only in the wave group that runs later, gone with any change of the schedule.
          const unsigned v = (cchunk < p.Cout && !(p.dbg & 4)) ? vrow[ps] + i * 64 : kOOB;
          __builtin_amdgcn_raw_buffer_store_b128(o, rsrc_y, v, 0, FT8_STORE_AUX);
        }
      }
    }
  };
  auto epilogue_partial = [&](int m0, int co0, int phase, int ksplit) __attribute__((always_inline)) {
    // cross-workgroup split-K: raw fp32 partial tile -> workspace [ksplit][phase * M + pixel][Cout_pad]; the scale / shift /
    // activation epilogue runs in conv_splitk_reduce_kernel once every slice has landed
    float* wsb = p.ws + ((size_t)ksplit * p.nph + phase) * (size_t)p.M * p.Cout_pad;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m0 + wp * 64 + j * 32 + l31;
      if (m >= p.M || (p.dbg & 4)) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int cb = co0 + wc * 128 + i * 32 + 8 * rg + 4 * lhi;
          const float4_t v = {acc[i][j][rg * 4], acc[i][j][rg * 4 + 1], acc[i][j][rg * 4 + 2], acc[i][j][rg * 4 + 3]};
          *reinterpret_cast<float4_t*>(wsb + (size_t)m * p.Cout_pad + cb) = v;
        }
    }
  };
  auto epilogue_tail = [&](int m0, int phase) __attribute__((always_inline)) {
    // fused tail 1x1 conv: out2[co2][pixel] = sum_c Wt[co2][c] * act(...)[c][pixel] + bt[co2], co2 < tail_cout <= 24.  Per wave
    // a [32 x 128] x [128 x 64] product on the matrix cores: the activated accumulators of (i, rg pair) ARE a 16-deep B
    // operand (lane = pixel, 8 halves = channels 8 rg + 4 lhi + {0..3} of rg = 2 pr and 2 pr + 1), the tail weights are
    // gathered in that K order (two 8-byte pieces per lane); hi + lo weight halves as in the shared epilogue (fp32-grade
    // weights: the heatmap conv decides the arg-max).  The two channel halves of the workgroup (wc = 0 / 1) each hold a partial
    // sum over 128 channels: wave (wp, wc) hands the partial of pixel tile j = 1 - wc to its partner (wp, 1 - wc) through LDS
    // and finishes tile j = wc itself, so only one 32 x 32 partial is live per wave at a time.
    // The weight fragments come through INLINE-ASM loads with hand-counted vmcnt: the next tile's LDS-DMA is in flight here, and
    // beside it hipcc waits vmcnt(0) for every ordinary load (one exposed L2 round trip each: 15 us per tile measured).  Three
    // steps of weights are in flight; the first wait of a tile also retires that DMA, which the K-loop needs landed anyway.
    const int py = phase >> 1, px = phase & 1;
    // (the lane id passes through an empty asm as well: everything derived from it below is recomputed here instead of being
    // carried — spilled — across the K-loop; a reload beside the LDS-DMA costs a vmcnt(0) each)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int lane = lane_e, l31 = lane_e & 31, lhi = lane_e >> 5;
    // (the pointer passes through an empty asm so that nothing derived from it — lane addresses, the bias values — is
    // hoisted out of the tile loop: kept live across the K-loop those cost 54 spilled registers, reloaded here one by one)
    const char* tw = p.tail_w;
    asm volatile("" : "+s"(tw));
    // the tail weights in THIS kernel's operand order (4th section of the tail pack, include/flowtrack_hip.h): fp16
    // [channel half wc][step st = (i, pr)][hi, lo][lane][8], lane (l31 = tail output, lhi) holding the weights of channels
    // wc * 128 + st * 16 + 4 lhi + {0..3} and + 8 + {0..3}: one coalesced 16-byte load per lane per (step, hi / lo) — gathered
    // from the row-major tables as 8-byte pieces the same loads took 9 us per tile (32 cache lines per instruction)
    const char* const wperm = tw + (size_t)2 * 32 * kT8 * 2 + 128 + (size_t)wc * 8 * 2048;
    const unsigned lane16 = (unsigned)lane * 16u;
    uint4_t w[3][2];                                              // [slot][hi, lo]
    auto issue = [&](auto slotc, auto stepc) __attribute__((always_inline)) {
      constexpr int sl = decltype(slotc)::value, st = decltype(stepc)::value;
      ld2x16_asm(w[sl][0], w[sl][1], lane16, wperm + st * 2048);
    };
    auto partial = [&](auto jc, float16_t& a2) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
#pragma unroll
      for (int r = 0; r < 16; ++r) a2[r] = 0.f;
      using I2 = std::integral_constant<int, 2>;
      issue(I0{}, I0{});
      issue(I1{}, I1{});
      issue(I2{}, I2{});
      static_for<8>([&](auto stc) {
        constexpr int st = decltype(stc)::value;
        constexpr int i = st >> 1, pr = st & 1, sl = st % 3;
        // this step's four pieces have landed once at most the pieces of the later steps in flight are outstanding
        constexpr int later = (st + 2 < 8 ? 2 : 7 - st) * 2;
        wait_vm_asm<later>(w[sl][0], w[sl][1]);
        __builtin_amdgcn_sched_barrier(0);
        const int c0 = wc * 128 + i * 32 + 16 * pr + 4 * lhi;
        const uint2v_t ha = activated(i, j, 2 * pr, c0), hb = activated(i, j, 2 * pr + 1, c0 + 8);
        const uint4_t tb = {ha[0], ha[1], hb[0], hb[1]};
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, w[sl][0]), __builtin_bit_cast(half8_t, tb), a2, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, w[sl][1]), __builtin_bit_cast(half8_t, tb), a2, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (st + 3 < 8) issue(std::integral_constant<int, sl>{}, std::integral_constant<int, st + 3>{});   // (its slot is free again)
      });
    };
    float4_t* slab = reinterpret_cast<float4_t*>(smem + kRingB);
    float16_t a2;
    auto send = [&](auto jc) __attribute__((always_inline)) {
      partial(jc, a2);
#pragma unroll
      for (int q4 = 0; q4 < 3; ++q4) {                             // registers 0..11 = tail outputs 0..23
        const float4_t v = {a2[4 * q4], a2[4 * q4 + 1], a2[4 * q4 + 2], a2[4 * q4 + 3]};
        slab[((wp * 2 + (1 - wc)) * 3 + q4) * 64 + lane] = v;       // read by wave (wp, 1 - wc)
      }
    };
    if (wc == 0) { send(I1{}); partial(I0{}, a2); } else { send(I0{}); partial(I1{}, a2); }
    // (raw barrier: __syncthreads() would drain the next tile's LDS-DMA, which is in flight here, with a vmcnt(0))
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    FT_LDS_BARRIER();
    if (!(p.dbg & 4)) {
      const long long hw = (long long)p.Ho * p.Wo;
      int n, pix;
      if (out_pixel(m0, py, px, wc, l31, n, pix)) {
#pragma unroll
        for (int q4 = 0; q4 < 3; ++q4) {
          const float4_t o = slab[((wp * 2 + wc) * 3 + q4) * 64 + lane];
          const float4_t bq = *reinterpret_cast<const float4_t*>(smem + kRingB + kShiftOff + kTailBiasOff + (8 * q4 + 4 * lhi) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int co2 = e + 8 * q4 + 4 * lhi;
            if (co2 < p.tail_cout) {
              const float v = a2[4 * q4 + e] + o[e] + bq[e];
              if (p.out_layout == FT_LAYOUT_NHWC)
                reinterpret_cast<half_t*>(p.y)[((long long)n * hw + pix) * p.y_cstride + p.y_coff + co2] = (half_t)v;
              else
                reinterpret_cast<float*>(p.y)[((long long)n * p.tail_cout + co2) * hw + pix] = v;
            }
          }
        }
      }
    }
  };

  // ---- the K-tile loop, straight through the workgroup's tiles --------------------------------------------------------------
  int o_m0, o_co0, o_phase, o_ksplit, o_nkt;              // the tile being multiplied
  int n_m0 = 0, n_co0 = 0, n_phase = 0, n_ksplit = 0, n_nkt = 0;   // the tile being staged (once the stream has moved on)
  setup(local, o_m0, o_co0, o_phase, o_ksplit, o_nkt);
  prologue();
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");        // K-tile 0 of the first tile has landed
  FT_LDS_BARRIER();
  if (wc == 1) FT_LDS_BARRIER();                            // group 1 runs one barrier behind group 0 through the K-loop
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int t_rem = o_nkt;                                        // K-tiles of the current tile still to multiply (>= 2 at a tile's start)
  bool more = false;
  for (;;) {
    if (t_rem == 2) {
      // the last two K-tiles of a tile stage the first two of the NEXT one: re-make the staging state now
      local += x_slots;
      more = local < x_count;
      if (more) setup(local, n_m0, n_co0, n_phase, n_ksplit, n_nkt);
      else kill();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (phase 1 counts LDS reads with lgkmcnt: nothing else may be pending)
    }
    ktile();
    if (--t_rem == 0) {
      if (wc == 0) FT_LDS_BARRIER();                        // group 0 catches up: both groups meet at the tile boundary
      // (the next tile's K-tile 0 landed behind the phase-4 wait just passed; its K-tile 1 is in flight)
      if constexpr (MODE == 1) epilogue_tail(o_m0, o_phase);
      else if constexpr (MODE == 2) epilogue_partial(o_m0, o_co0, o_phase, o_ksplit);
      else epilogue_store(o_m0, o_co0, o_phase);
      if (!more) break;
      o_m0 = n_m0; o_co0 = n_co0; o_phase = n_phase; o_ksplit = n_ksplit; o_nkt = n_nkt;
      t_rem = o_nkt;
      __builtin_amdgcn_sched_barrier(0);                    // (the zeroing below must not be scheduled up into the epilogue: 128 live registers)
      if (wc == 1) FT_LDS_BARRIER();
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
#endif
}

int launch_igemm8(const ConvParams& p, unsigned tiles, hipStream_t s) {
  constexpr size_t lds = (size_t)kRingB + (size_t)kSlabB;
  static int cus[64] = {};
  int dev = 0;
  FT_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return FT_ERR_INVALID_ARG;
  if (!cus[dev]) {
    int n = 0;
    FT_HIP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    cus[dev] = n > 0 ? n : 256;
  }
  const unsigned grid = tiles < (unsigned)cus[dev] ? tiles : (unsigned)cus[dev];
  if (p.tail_w) {
    FT_RAISE_LDS(conv_igemm8_kernel<1>, lds);
    hipLaunchKernelGGL(conv_igemm8_kernel<1>, dim3(grid), dim3(512), lds, s, p);
  } else if (p.sk > 1) {
    FT_RAISE_LDS(conv_igemm8_kernel<2>, lds);
    hipLaunchKernelGGL(conv_igemm8_kernel<2>, dim3(grid), dim3(512), lds, s, p);
  } else {
    FT_RAISE_LDS(conv_igemm8_kernel<0>, lds);
    hipLaunchKernelGGL(conv_igemm8_kernel<0>, dim3(grid), dim3(512), lds, s, p);
  }
  return FT_OK;
}

}  // namespace ft
