// Microbenchmark: the weight-stream pattern of bottleneck_stream_direct_kernel in isolation.  Every wave owns an 8-KiB block
// per step ([step][wave][8 fragments]), keeps a 3-slot register ring two steps ahead (8 loads per step, counted vmcnt) and
// "computes" for SLEEP x 64 cycles per step.  How many cycles does a step take as a function of the compute time?
//   hipcc --offload-arch=gfx950 -O3 l2_burst.hip -o l2_burst
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

template <int SLEEP, int AHEAD>
__global__ __launch_bounds__(256, 1) void burst_kernel(const char* __restrict__ w, unsigned s_bytes, int steps, unsigned* sink, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(w), 0, s_bytes, 0x00020000);
  constexpr int NS = AHEAD + 1;
  uint4_t ring[NS][8];
  uint4_t acc = {0, 0, 0, 0};
  unsigned pos = (unsigned)wave * 8192u;
  auto issue = [&](auto slotc) {
    constexpr int S = decltype(slotc)::value;
#pragma unroll
    for (int f = 0; f < 8; ++f) ring[S][f] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u, pos + f * 1024u, 0);
    pos += 4 * 8192u;
    if (pos >= s_bytes) pos -= s_bytes;
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if constexpr (AHEAD >= 1) issue(std::integral_constant<int, 0>{});
  if constexpr (AHEAD >= 2) issue(std::integral_constant<int, 1>{});
  if constexpr (AHEAD >= 3) issue(std::integral_constant<int, 2>{});
  auto body = [&](auto sc) {
    constexpr int S = decltype(sc)::value;
    issue(std::integral_constant<int, (S + AHEAD) % NS>{});
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * AHEAD) : "memory");
#pragma unroll
    for (int f = 0; f < 8; ++f) acc ^= ring[S][f];
    if constexpr (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
  };
  for (int s = 0; s < steps; s += NS) {
    if constexpr (NS >= 1) body(std::integral_constant<int, 0>{});
    if constexpr (NS >= 2) body(std::integral_constant<int, 1 % NS>{});
    if constexpr (NS >= 3) body(std::integral_constant<int, 2 % NS>{});
    if constexpr (NS >= 4) body(std::integral_constant<int, 3 % NS>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[threadIdx.x] = acc.x;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SLEEP, int AHEAD>
static void run(const char* w, unsigned s_bytes, unsigned* sink, unsigned long long* cyc) {
  const int steps = 144;
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((burst_kernel<SLEEP, AHEAD>), dim3(256), dim3(256), 0, 0, w, s_bytes, steps, sink, cyc);
  CK(hipDeviceSynchronize());
  unsigned long long h[256];
  CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  double m = 0;
  for (int i = 0; i < 256; ++i) m += (double)h[i];
  m /= 256;
  printf("sleep %2d (x64 cyc) ahead %d: %7.0f cycles per step (32 KiB per CU per step -> %5.1f B/clk/CU)\n", SLEEP, AHEAD, m / steps, 32768.0 / (m / steps));
}

int main() {
  char* w; unsigned* sink; unsigned long long* cyc;
  const unsigned s = 144u * 32768u;   // phase 2 of the 256-plane block: 4.7 MB? no: 144 steps x 32 KiB = 4.5 MiB; use 1.18 MB wrap like W2
  CK(hipMalloc(&w, s)); CK(hipMemset(w, 1, s)); CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&cyc, 256 * 8));
  for (unsigned sb : {1179648u, 2228224u}) {
    printf("stream %u bytes\n", sb);
    run<0, 2>(w, sb, sink, cyc); run<4, 2>(w, sb, sink, cyc); run<8, 2>(w, sb, sink, cyc); run<11, 2>(w, sb, sink, cyc); run<16, 2>(w, sb, sink, cyc);
    run<0, 3>(w, sb, sink, cyc); run<8, 3>(w, sb, sink, cyc); run<11, 3>(w, sb, sink, cyc);
    run<8, 1>(w, sb, sink, cyc); run<11, 1>(w, sb, sink, cyc);
  }
  return 0;
}
