// Microbenchmark: how fast can every CU pull the SAME stream (a layer's packed weights, 1-8 MiB) out of L2 straight into
// registers (buffer_load_b128, DEPTH loads in flight per wave), and does it matter that all workgroups walk the stream in
// lock-step from the same start?   hipcc --offload-arch=gfx950 -O3 l2_stream.hip -o l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

// MODE 0: every workgroup starts at byte 0; 1: workgroup b starts at (b * rot) mod S (rot in KiB);
template <int NW, int DEPTH>
__global__ __launch_bounds__(64 * NW, 1) void stream_kernel(const char* __restrict__ w, unsigned s_bytes, int pieces_per_wave, unsigned rot_bytes,
                                                            unsigned* sink) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(w), 0, s_bytes, 0x00020000);
  unsigned pos = (unsigned)(((unsigned long long)blockIdx.x * rot_bytes + (unsigned)wave * 1024u) % s_bytes);
  uint4_t ring[DEPTH];
  uint4_t acc = {0, 0, 0, 0};
  auto next = [&]() {
    const unsigned p = pos;
    pos += NW * 1024u;
    if (pos >= s_bytes) pos -= s_bytes;
    return p;
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) ring[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u, next(), 0);
  for (int i = 0; i < pieces_per_wave; i += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const uint4_t v = ring[d];
      acc ^= v;
      ring[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u, next(), 0);
    }
  }
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) acc ^= ring[d];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[threadIdx.x] = acc.x;
}

template <int NW, int DEPTH>
static void run(const char* w, unsigned s_bytes, size_t per_wg_bytes, unsigned rot, unsigned* sink, int grid) {
  const int ppw = (int)(per_wg_bytes / 1024 / NW);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((stream_kernel<NW, DEPTH>), dim3(grid), dim3(64 * NW), 0, 0, w, s_bytes, ppw, rot, sink);
  CK(hipEventRecord(e0));
  const int reps = 5;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((stream_kernel<NW, DEPTH>), dim3(grid), dim3(64 * NW), 0, 0, w, s_bytes, ppw, rot, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double bytes = (double)grid * ppw * NW * 1024.0;
  printf("S %5u KiB  waves %d depth %2d rot %6u KiB  grid %d: %7.1f us  %6.2f TB/s  %6.1f GB/s per CU\n", s_bytes >> 10, NW, DEPTH, rot >> 10, grid,
         ms * 1e3, bytes / ms * 1e-9, bytes / grid / ms * 1e-6);
}

int main() {
  char* w; unsigned* sink;
  CK(hipMalloc(&w, 64u << 20)); CK(hipMemset(w, 1, 64u << 20)); CK(hipMalloc(&sink, 4096));
  const size_t per_wg = 8u << 20;
  for (unsigned s : {1u << 20, 2u << 20, 3u << 20, 8u << 20, 32u << 20}) {
    for (unsigned rot : {0u, 8u << 10, 37u << 10, 256u << 10}) {
      run<4, 8>(w, s, per_wg, rot, sink, 256);
      run<4, 16>(w, s, per_wg, rot, sink, 256);
      run<4, 32>(w, s, per_wg, rot, sink, 256);
      run<8, 8>(w, s, per_wg, rot, sink, 256);
      run<8, 16>(w, s, per_wg, rot, sink, 256);
    }
  }
  // two workgroups per CU
  run<4, 16>(w, 2u << 20, per_wg, 0, sink, 512);
  run<4, 16>(w, 2u << 20, per_wg, 37u << 10, sink, 512);
  return 0;
}
