// Microbenchmark: how fast can a CU pull L2-resident operand bytes (a) into LDS with buffer_load ... lds,
// (b) into VGPRs with buffer_load_dwordx4, (c) both at once.  Decides whether the weight operand of the
// implicit-GEMM kernel should bypass LDS.   hipcc --offload-arch=gfx950 -O3 opdeliv.hip -o opdeliv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float float4v __attribute__((ext_vector_type(4)));

template <int MODE, int INFLIGHT, int ROWSTRIDE>
__global__ __launch_bounds__(256) void deliver(const char* __restrict__ src, size_t src_bytes, int iters, float* sink,
                                               int shared_src) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)src_bytes, 0x00020000);
  // every wave walks 1 KiB chunks; shared_src: all blocks walk the same region (weights), else block-private region
  const unsigned region = shared_src ? 0u : (unsigned)((blockIdx.x % 256u) * 65536u);  // 32 regions x 64 KiB per XCD: L2 hits, L1 misses
  unsigned off = region + wave * 1024u + lane * 16u;
  const unsigned stride = nw * 1024u;
  const unsigned wrap = shared_src ? (unsigned)src_bytes : 65536u;
  float4v acc = {0, 0, 0, 0};
  char* my_lds = lds + wave * (INFLIGHT * 1024);
  const unsigned gregion = (blockIdx.x % 64u) * 262144u;   // row-gather: 512 rows x 512 B per region, 8 regions per XCD
  for (int it = 0; it < iters; ++it) {
    if (ROWSTRIDE > 0) {
      // conv-like: a wave-load = 16 rows x 64 B (4 lanes per row), K-steps walk along the row
      const unsigned ksteps_per_row = ROWSTRIDE / 64;
      const unsigned kk = it % ksteps_per_row, blk = (it / ksteps_per_row) % (262144u / (ROWSTRIDE * nw * 16u * INFLIGHT));
#pragma unroll
      for (int j = 0; j < INFLIGHT; ++j) {
        const unsigned row = (blk * INFLIGHT + j) * nw * 16u + wave * 16u + (lane >> 2);
        const unsigned o = gregion + row * ROWSTRIDE + kk * 64u + (lane & 3) * 16u;
        if (MODE == 0)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(my_lds + j * 1024), 16, o, 0, 0, 0);
        else
          acc += __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0));
      }
      if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      continue;
    }
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int j = 0; j < INFLIGHT; ++j) {
        unsigned o = region + (off - region + j * stride) % wrap;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(my_lds + j * 1024), 16, o, 0, 0, 0);
      }
    }
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int j = 0; j < INFLIGHT; ++j) {
        unsigned o = region + (off - region + (j + INFLIGHT) * stride) % wrap;
        float4v v = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0));
        acc += v;
      }
    }
    off = region + (off - region + 2 * INFLIGHT * stride) % wrap;
    if (MODE != 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  }
  if (MODE != 1) acc.x += ((float*)my_lds)[lane];
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
#endif
}

template <int MODE, int INF, int RS = 0>
double run(const char* src, size_t bytes, int grid, int threads, int iters, float* sink, int shared_src, size_t lds_bytes) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((deliver<MODE, INF, RS>), dim3(grid), dim3(threads), lds_bytes, 0, src, bytes, 4, sink, shared_src);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((deliver<MODE, INF, RS>), dim3(grid), dim3(threads), lds_bytes, 0, src, bytes, iters, sink, shared_src);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double per_wave_iter = (MODE == 2 ? 2.0 : 1.0) * INF * 1024.0;
  const double total = (double)grid * (threads / 64) * iters * per_wave_iter;
  return total / (ms * 1e-3) / 1e12;
}

int main() {
  const size_t bytes = 64u << 20;  // 64 MiB source: block-private 64 KiB regions stay L2/MALL resident
  char* src; float* sink;
  CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 1, bytes)); CK(hipMalloc(&sink, 64));
  const int iters = 2000;
  printf("mode: 0 = buffer_load..lds only, 1 = VGPR loads only, 2 = both (bytes counted for both)\n");
  for (int shared_src = 0; shared_src <= 1; ++shared_src) {
    const size_t sb = shared_src ? (2u << 20) : bytes;
    for (int wg_per_cu = 1; wg_per_cu <= 4; ++wg_per_cu) {
      const int grid = 256 * wg_per_cu;
      const size_t ldsb = 160 * 1024 / wg_per_cu > 65536 ? 65536 : 160 * 1024 / wg_per_cu / 1024 * 1024;
      printf("shared_src=%d  %d WG/CU x 4 waves, 4 x 1KiB in flight per wave per kind:  lds %.2f TB/s   vgpr %.2f TB/s   both %.2f TB/s\n",
             shared_src, wg_per_cu, run<0, 4>(src, sb, grid, 256, iters, sink, shared_src, ldsb),
             run<1, 4>(src, sb, grid, 256, iters, sink, shared_src, ldsb), run<2, 4>(src, sb, grid, 256, iters, sink, shared_src, ldsb));
    }
    printf("shared_src=%d  3 WG/CU x 4 waves, 8 in flight:  lds %.2f   vgpr %.2f   both %.2f TB/s\n", shared_src,
           run<0, 8>(src, sb, 768, 256, iters, sink, shared_src, 49152), run<1, 8>(src, sb, 768, 256, iters, sink, shared_src, 49152),
           run<2, 8>(src, sb, 768, 256, iters, sink, shared_src, 49152));
  }
  printf("row-gather (16 rows x 64 B per wave-load), 3 WG/CU x 4 waves x 4 in flight:\n");
  printf("  rowstride 512:  lds %.2f  vgpr %.2f TB/s\n", run<0, 4, 512>(src, bytes, 768, 256, iters, sink, 0, 49152), run<1, 4, 512>(src, bytes, 768, 256, iters, sink, 0, 49152));
  printf("  rowstride 128:  lds %.2f  vgpr %.2f TB/s\n", run<0, 4, 128>(src, bytes, 768, 256, iters, sink, 0, 49152), run<1, 4, 128>(src, bytes, 768, 256, iters, sink, 0, 49152));
  printf("  rowstride 2048: lds %.2f  vgpr %.2f TB/s\n", run<0, 4, 2048>(src, bytes, 768, 256, iters, sink, 0, 49152), run<1, 4, 2048>(src, bytes, 768, 256, iters, sink, 0, 49152));
  printf("  rowstride 512, 8 in flight:  lds %.2f  vgpr %.2f TB/s\n", run<0, 8, 512>(src, bytes, 768, 256, iters, sink, 0, 49152), run<1, 8, 512>(src, bytes, 768, 256, iters, sink, 0, 49152));
  printf("  rowstride 512, 2 WG/CU:  lds %.2f  vgpr %.2f TB/s\n", run<0, 4, 512>(src, bytes, 512, 256, iters, sink, 0, 49152), run<1, 4, 512>(src, bytes, 512, 256, iters, sink, 0, 49152));
  return 0;
}
