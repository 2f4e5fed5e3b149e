// Is "issuing wave: s_waitcnt vmcnt(N) (older loads done) -> s_barrier -> OTHER wave: ds_read" enough to see LDS-DMA data?
// Mimics conv_stem_kernel's prologue: per wave one cold 1-KiB patch load, then two L2-hot weight loads, vmcnt(1), barrier,
// then every wave checks the patch piece of its neighbour.   hipcc --offload-arch=gfx950 -O3 dma_visibility.hip -o dma_visibility
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>   // 0: vmcnt(1) + s_barrier; 1: vmcnt(1) lgkmcnt(0) + s_barrier; 2: vmcnt(0) + s_barrier; 3: mode 0 plus a table ds_write before
__global__ __launch_bounds__(256) void probe(const unsigned* __restrict__ src, size_t n_words, const unsigned* __restrict__ hot,
                                             unsigned* __restrict__ bad) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];   // [0..3071]: 3 weight slots of 4 KiB, [3072..4095+]: patch 4 KiB, then a table
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, 0x7fffffff, 0x00020000);
  __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(hot), 0, 65536, 0x00020000);
  unsigned* patch = lds + 3072;
  long long* table = reinterpret_cast<long long*>(lds + 3072 + 1024);
  const size_t chunk = ((size_t)blockIdx.x * 4 + wave) * 4099 % (n_words / 256);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(patch + wave * 256), 16, (unsigned)(chunk * 1024 + lane * 16), 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (lds_ptr)(lds + wave * 256), 16, (unsigned)(wave * 1024 + lane * 16), 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (lds_ptr)(lds + 1024 + wave * 256), 16, (unsigned)(wave * 1024 + lane * 16), 64, 0, 0);
  if (MODE == 3 && threadIdx.x < 128) table[threadIdx.x] = (long long)blockIdx.x * 128 + threadIdx.x;
  if (MODE == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
  else if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  const int nb = (wave + 1) & 3;
  const size_t nchunk = ((size_t)blockIdx.x * 4 + nb) * 4099 % (n_words / 256);
  const unsigned got = patch[nb * 256 + lane * 4 + 1];
  const unsigned want = src[nchunk * 256 + lane * 4 + 1];
  if (got != want) atomicAdd(bad, 1u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

int main() {
  const size_t n_words = (size_t)1 << 28;
  unsigned *src, *hot, *bad;
  CK(hipMalloc(&src, n_words * 4)); CK(hipMalloc(&hot, 65536 + 4096)); CK(hipMalloc(&bad, 4));
  unsigned* h = (unsigned*)malloc(n_words * 4);
  for (size_t i = 0; i < n_words; ++i) h[i] = (unsigned)(i * 2654435761u) | 1u;
  CK(hipMemcpy(src, h, n_words * 4, hipMemcpyHostToDevice)); CK(hipMemset(hot, 0, 65536 + 4096));
  const size_t lds = (3072 + 1024) * 4 + 128 * 8;
  for (int mode = 0; mode < 4; ++mode) {
    unsigned total = 0;
    for (int rep = 0; rep < 8; ++rep) {
      CK(hipMemset(bad, 0, 4));
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(16384), dim3(256), lds, 0, src, n_words, hot, bad);
      if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(16384), dim3(256), lds, 0, src, n_words, hot, bad);
      if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(16384), dim3(256), lds, 0, src, n_words, hot, bad);
      if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(16384), dim3(256), lds, 0, src, n_words, hot, bad);
      CK(hipDeviceSynchronize());
      unsigned b; CK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost)); total += b;
    }
    const char* names[4] = {"vmcnt(1) + barrier", "vmcnt(1) lgkmcnt(0) + barrier", "vmcnt(0) + barrier", "table ds_write, vmcnt(1) + barrier"};
    printf("mode %d (%s): lanes that read a neighbour's patch piece before it was visible: %u of %u\n", mode, names[mode], total, 8u * 16384u * 256u);
  }
  return 0;
}
