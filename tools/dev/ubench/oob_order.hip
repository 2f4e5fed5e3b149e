// Does an all-lanes-out-of-range `buffer_load ... lds` retire BEFORE an older in-range one (so that a counted
// s_waitcnt vmcnt(1) no longer proves the older load has landed)?   hipcc --offload-arch=gfx950 -O3 oob_order.hip -o oob_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>   // 0: younger load out of range, 1: younger load in range (L2-hot address), 2: vmcnt(0)
__global__ __launch_bounds__(256) void probe(const unsigned* __restrict__ src, size_t n_words, const unsigned* __restrict__ hot,
                                             unsigned* __restrict__ bad) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) unsigned lds[4][2][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, (int)(n_words * 4 > 0x7fffffff ? 0x7fffffff : n_words * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(hot), 0, 4096, 0x00020000);
  for (int i = 0; i < 256; ++i) { lds[wave][0][lane * 4 + (i & 3)] = 0xdeadbeefu; }
  __syncthreads();
  // cold 1 KiB per wave: unique, strided far apart
  const size_t chunk = ((size_t)blockIdx.x * 4 + wave) * 4099 % (n_words / 256);
  const unsigned voff = (unsigned)(chunk * 1024 + lane * 16);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)&lds[wave][0][0], 16, voff, 0, 0, 0);
  if (MODE == 0)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)&lds[wave][1][0], 16, 0x80000000u, 0, 0, 0);
  else
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (__attribute__((address_space(3))) void*)&lds[wave][1][0], 16, lane * 16, 0, 0, 0);
  if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  // own wave's data: no barrier needed
  const unsigned got = lds[wave][0][lane * 4];
  const unsigned want = src[chunk * 256 + lane * 4];
  if (got != want) atomicAdd(bad, 1u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

int main() {
  const size_t n_words = (size_t)1 << 28;   // 1 GiB
  unsigned *src, *hot, *bad;
  CK(hipMalloc(&src, n_words * 4)); CK(hipMalloc(&hot, 4096)); CK(hipMalloc(&bad, 4));
  unsigned* h = (unsigned*)malloc(n_words * 4);
  for (size_t i = 0; i < n_words; ++i) h[i] = (unsigned)(i * 2654435761u) | 1u;
  CK(hipMemcpy(src, h, n_words * 4, hipMemcpyHostToDevice)); CK(hipMemset(hot, 0, 4096));
  for (int mode = 0; mode < 3; ++mode) {
    unsigned total = 0;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemset(bad, 0, 4));
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(16384), dim3(256), 0, 0, src, n_words, hot, bad);
      if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(16384), dim3(256), 0, 0, src, n_words, hot, bad);
      if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(16384), dim3(256), 0, 0, src, n_words, hot, bad);
      CK(hipDeviceSynchronize());
      unsigned b; CK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost)); total += b;
    }
    printf("mode %d (%s): lanes that read the older load's LDS before it landed: %u of %u\n", mode,
           mode == 0 ? "younger load all out of range, vmcnt(1)" : mode == 1 ? "younger load in range (L2-hot), vmcnt(1)" : "vmcnt(0)",
           total, 5u * 16384u * 256u);
  }
  return 0;
}
