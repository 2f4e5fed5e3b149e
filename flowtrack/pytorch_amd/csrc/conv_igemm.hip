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
//   One 256-thread workgroup (4 wave64) computes a BP x BC tile; operands are staged global ->
//   registers -> LDS (XOR-swizzled 16-byte chunks so the ds_read_b128 fragment reads are
//   bank-conflict free), double-buffered with one barrier per K-step.  fp16 uses
//   v_mfma_f32_32x32x16_f16 (fp32 accumulate); fp32 uses v_mfma_f32_32x32x2_f32 (exact fp32 FMA
//   chain) so the parity mode and the fast mode share every line of index arithmetic.
#include "ft_common.h"

namespace ft {

struct ConvParams {
  const char* x;
  const char* w;
  const float* scale;
  const float* shift;
  const char* res;
  char* y;
  int M;            // pixels per phase = N * Hq * Wq
  int HqWq, Wq;     // pixel-grid decode
  int Hi, Wi;
  int sy;           // input step per pixel-grid step (conv stride; 1 for transposed)
  int x_cstride, x_coff;
  int kh, kw;       // taps per phase (2x2 for transposed)
  int dmul;         // +1 conv, -1 transposed
  int pad;          // conv padding (unused for transposed)
  int transposed;
  int cin_groups;   // roundup8(Cin) / VEC
  int nk;           // K-steps
  int Kpad;         // elements per packed weight row
  int Cout, Cout_pad;
  int Ho, Wo, omul; // output tensor size; 1 (conv) or 2 (transposed) output step per grid step
  int y_cstride, y_coff, out_layout;
  int res_cstride, res_coff;
  int act;
  float slope;
};

template <typename T> struct Elem;
template <> struct Elem<half_t> { static constexpr int VEC = 8; };
template <> struct Elem<float> { static constexpr int VEC = 4; };

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  if (act == FT_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == FT_ACT_LEAKY) return v > 0.f ? v : v * slope;
  return v;
}

// One 32-byte-per-row K slice for every (i, j) MFMA tile of the wave.
template <int MT_C, int MT_P>
__device__ __forceinline__ void mma_slice(const uint4_t (&a)[MT_C], const uint4_t (&b)[MT_P],
                                          float16_t (&acc)[MT_C][MT_P], half_t*) {
#pragma unroll
  for (int i = 0; i < MT_C; ++i)
#pragma unroll
    for (int j = 0; j < MT_P; ++j)
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
          __builtin_bit_cast(half8_t, a[i]), __builtin_bit_cast(half8_t, b[j]), acc[i][j], 0, 0, 0);
}
template <int MT_C, int MT_P>
__device__ __forceinline__ void mma_slice(const uint4_t (&a)[MT_C], const uint4_t (&b)[MT_P],
                                          float16_t (&acc)[MT_C][MT_P], float*) {
  // whole-vector bit casts: __builtin_bit_cast on a single vector ELEMENT lvalue reads element 0
  float4_t af[MT_C], bf[MT_P];
#pragma unroll
  for (int i = 0; i < MT_C; ++i) af[i] = __builtin_bit_cast(float4_t, a[i]);
#pragma unroll
  for (int j = 0; j < MT_P; ++j) bf[j] = __builtin_bit_cast(float4_t, b[j]);
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int i = 0; i < MT_C; ++i)
#pragma unroll
      for (int j = 0; j < MT_P; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
}

__device__ __forceinline__ void store4(half_t* dst, const float (&v)[4]) {
  half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
  *reinterpret_cast<half4_t*>(dst) = h;
}
__device__ __forceinline__ void store4(float* dst, const float (&v)[4]) {
  float4_t f = {v[0], v[1], v[2], v[3]};
  *reinterpret_cast<float4_t*>(dst) = f;
}
__device__ __forceinline__ void load4(const half_t* src, float (&v)[4]) {
  half4_t h = *reinterpret_cast<const half4_t*>(src);
  v[0] = (float)h[0]; v[1] = (float)h[1]; v[2] = (float)h[2]; v[3] = (float)h[3];
}
__device__ __forceinline__ void load4(const float* src, float (&v)[4]) {
  float4_t f = *reinterpret_cast<const float4_t*>(src);
  v[0] = f[0]; v[1] = f[1]; v[2] = f[2]; v[3] = f[3];
}

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

  const int phase = blockIdx.z;
  const int py = phase >> 1, px = phase & 1;
  const int m0 = blockIdx.x * BP;
  const int co0 = blockIdx.y * BC;
  const int dbase_y = p.transposed ? py : -p.pad;
  const int dbase_x = p.transposed ? px : -p.pad;

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

  // ---- epilogue: scale/shift (+residual) + activation, NHWC runs of 4 or NCHW fp32 ---------
#pragma unroll
  for (int j = 0; j < MT_P; ++j) {
    const int m = m0 + wp * WT_P + j * 32 + l31;
    if (m >= p.M) continue;
    const int n = m / p.HqWq;
    const int rem = m - n * p.HqWq;
    const int qy = rem / p.Wq;
    const int qx = rem - qy * p.Wq;
    const int oy = qy * p.omul + py, ox = qx * p.omul + px;
    const size_t opix = ((size_t)n * p.Ho + oy) * p.Wo + ox;
#pragma unroll
    for (int i = 0; i < MT_C; ++i) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int cb = co0 + wc * WT_C + i * 32 + 8 * rg + 4 * lhi;
        if (cb >= p.Cout) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][rg * 4 + e];
        if (p.scale) {
          const float4_t s = *reinterpret_cast<const float4_t*>(p.scale + cb);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= s[e];
        }
        if (p.shift) {
          const float4_t s = *reinterpret_cast<const float4_t*>(p.shift + cb);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += s[e];
        }
        const bool full = cb + 3 < p.Cout;
        if (p.res) {
          const T* rp = reinterpret_cast<const T*>(p.res) + opix * p.res_cstride + p.res_coff + cb;
          if (full) {
            float r4[4];
            load4(rp, r4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += r4[e];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (cb + e < p.Cout) v[e] += (float)rp[e];
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act, p.slope);
        if (p.out_layout == FT_LAYOUT_NHWC) {
          T* yp = reinterpret_cast<T*>(p.y) + opix * p.y_cstride + p.y_coff + cb;
          if (full) {
            store4(yp, v);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (cb + e < p.Cout) yp[e] = (T)v[e];
          }
        } else {
          float* yp = reinterpret_cast<float*>(p.y);
          const size_t hw = (size_t)p.Ho * p.Wo;
          const size_t pix = (size_t)oy * p.Wo + ox;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (cb + e < p.Cout) yp[((size_t)n * p.Cout + cb + e) * hw + pix] = v[e];
        }
      }
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------
constexpr int kBKB = 64;  // bytes of K per tile row per step (all instantiations)
constexpr int kBP = 128;

static int pick_bc(int cout) {
  if (cout <= 32) return 32;
  if (cout % 128 == 0) return 128;
  return 64;
}

struct Geometry {
  int nphases, ntaps, cin8, cout_pad, kpad, bc, nk, cin_groups, vec;
};

static int validate(const ft_conv_desc* d) {
  if (!d) return FT_ERR_INVALID_ARG;
  if (d->dtype != FT_F16 && d->dtype != FT_F32) return FT_ERR_INVALID_ARG;
  if (d->N <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Cin <= 0 || d->Cout <= 0) return FT_ERR_INVALID_ARG;
  if (d->x_cstride % 8 || d->x_coff % 8 || d->x_coff < 0) return FT_ERR_INVALID_ARG;
  if (d->x_cstride < d->x_coff + round_up(d->Cin, 8)) return FT_ERR_INVALID_ARG;
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
    if (d->y_cstride < d->y_coff + d->Cout) return FT_ERR_INVALID_ARG;
  } else if (d->out_layout != FT_LAYOUT_NCHW_F32) {
    return FT_ERR_INVALID_ARG;
  }
  if (d->has_residual) {
    if (d->res_cstride % 4 || d->res_coff % 4 || d->res_coff < 0) return FT_ERR_INVALID_ARG;
    if (d->res_cstride < d->res_coff + d->Cout) return FT_ERR_INVALID_ARG;
  }
  if (d->act < FT_ACT_NONE || d->act > FT_ACT_LEAKY) return FT_ERR_INVALID_ARG;
  return FT_OK;
}

static int geometry(const ft_conv_desc* d, Geometry* g) {
  int st = validate(d);
  if (st != FT_OK) return st;
  g->vec = d->dtype == FT_F16 ? 8 : 4;
  g->nphases = d->transposed ? 4 : 1;
  g->ntaps = d->transposed ? 4 : d->kh * d->kw;
  g->cin8 = round_up(d->Cin, 8);
  g->bc = pick_bc(d->Cout);
  g->cout_pad = round_up(d->Cout, g->bc);
  g->cin_groups = g->cin8 / g->vec;
  const int ch = kBKB / 16;
  g->nk = ceil_div(g->ntaps * g->cin_groups, ch);
  g->kpad = g->nk * ch * g->vec;
  return FT_OK;
}

template <typename T, int BC, int WGP, int WGC>
static void launch(const ConvParams& p, dim3 grid, hipStream_t s) {
  constexpr size_t lds = 2 * (size_t)(BC + kBP) * kBKB;
  hipLaunchKernelGGL((conv_igemm_kernel<T, kBP, BC, WGP, WGC, kBKB>), grid, dim3(256), lds, s, p);
}

template <typename T>
static void dispatch(int bc, const ConvParams& p, dim3 grid, hipStream_t s) {
  if (bc == 128) launch<T, 128, 2, 2>(p, grid, s);
  else if (bc == 64) launch<T, 64, 2, 2>(p, grid, s);
  else launch<T, 32, 4, 1>(p, grid, s);
}

}  // namespace ft

using namespace ft;

extern "C" int ft_conv_pack_geometry(const ft_conv_desc* d, int* nphases, int* ntaps, int* cin8,
                                     int* cout_pad, int* kpad) {
  Geometry g;
  int st = geometry(d, &g);
  if (st != FT_OK) return st;
  if (nphases) *nphases = g.nphases;
  if (ntaps) *ntaps = g.ntaps;
  if (cin8) *cin8 = g.cin8;
  if (cout_pad) *cout_pad = g.cout_pad;
  if (kpad) *kpad = g.kpad;
  return FT_OK;
}

extern "C" int ft_conv_tap_source(const ft_conv_desc* d, int phase, int tap, int* ky, int* kx) {
  Geometry g;
  int st = geometry(d, &g);
  if (st != FT_OK) return st;
  if (phase < 0 || phase >= g.nphases || tap < 0 || tap >= g.ntaps || !ky || !kx) return FT_ERR_INVALID_ARG;
  if (d->transposed) {
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
  return 2.0 * d->N * (double)d->Ho * d->Wo * d->Cout * d->Cin * taps;
}

extern "C" int ft_conv2d_fwd(const ft_conv_desc* d, const void* x, const void* w_packed,
                             const float* scale, const float* shift, const void* residual, void* y,
                             ft_stream_t stream) {
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
  p.transposed = d->transposed;
  p.cin_groups = g.cin_groups;
  p.nk = g.nk;
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

  dim3 grid(ceil_div(p.M, kBP), g.cout_pad / g.bc, g.nphases);
  hipStream_t s = as_stream(stream);
  if (d->dtype == FT_F16) dispatch<half_t>(g.bc, p, grid, s);
  else dispatch<float>(g.bc, p, grid, s);
  FT_LAUNCH_CHECK("conv_igemm_kernel");
  return FT_OK;
}
