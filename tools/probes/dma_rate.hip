// Per-CU LDS-DMA fetch rate (global_load_lds_dwordx4, 1 KiB per wave instruction) by source and by bytes in flight: what bounds
// a GEMM tile whose operands reach the CU once (fmt_gemm_rb_kernel).  One workgroup per CU (LDS 128 KiB), W waves, every wave
// keeps P pieces in flight behind a counted vmcnt and walks `span` bytes cyclically; no barriers, nothing is read back.
//   shared:  all workgroups of an XCD (ids congruent mod 8) walk the SAME 1 MiB window from different starting points (L2 hits)
//   private: every workgroup walks its own window (Infinity Cache / HBM by total footprint)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/dma_rate.hip -o build_ab/dma_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int P>
__global__ __launch_bounds__(1024) void dma_kernel(const char* __restrict__ base, size_t wg_stride, unsigned span, int iters, int bar) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const char* src = base + (size_t)blockIdx.x * wg_stride;
  // start offset: workgroups spread over the window
  unsigned off = (unsigned)(((size_t)blockIdx.x * 37 * 1024) % span);
  unsigned char* dst = ring + (w * P) * 1024;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      unsigned o = off + (unsigned)((it * nw * P + p * nw + w) * 1024);
      o %= span;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + o + lane * 16),
                                       (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
    }
    wait_vmcnt<P>();  // the previous round has landed; this one stays in flight
    if (bar) __builtin_amdgcn_s_barrier();
  }
  wait_vmcnt<0>();
}

template <int P>
void run(const char* tag, const char* buf, size_t wg_stride, unsigned span, int waves, int bar) {
  const int iters = 200;
  const int smem = waves * P * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dma_kernel<P>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  // pad the LDS so that ONE workgroup fits per CU
  const int lds = smem > 100 * 1024 ? smem : 100 * 1024;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(dma_kernel<P>, dim3(256), dim3(waves * 64), lds, nullptr, buf, wg_stride, span, iters, bar);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, nullptr));
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(dma_kernel<P>, dim3(256), dim3(waves * 64), lds, nullptr, buf, wg_stride, span, iters, bar);
  CK(hipEventRecord(e1, nullptr));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)iters * waves * P * 1024;  // per workgroup and launch
  const double us = ms * 1e3 / reps;
  printf("%-34s waves %2d  P %2d (%3d KiB in flight) bar %d: %7.1f us  %6.1f GB/s per CU  %5.2f TB/s chip\n", tag, waves, P, waves * P, bar, us,
         bytes / us * 1e-3, bytes * 256 / us * 1e-6);
}

int main() {
  const size_t total = (size_t)1 << 30;
  char* buf;
  CK(hipMalloc(&buf, total));
  CK(hipMemset(buf, 1, total));
  for (int bar = 0; bar < 2; ++bar) {
    for (int waves : {4, 8, 16}) {
      run<2>("shared 1 MiB (L2)", buf, 0, 1u << 20, waves, bar);
      run<4>("shared 1 MiB (L2)", buf, 0, 1u << 20, waves, bar);
      run<8>("shared 1 MiB (L2)", buf, 0, 1u << 20, waves, bar);
      if (waves <= 8) run<12>("shared 1 MiB (L2)", buf, 0, 1u << 20, waves, bar);
    }
    for (int waves : {8, 16}) {
      run<4>("private 256 KiB (256 x = 64 MiB, MALL)", buf, 256 << 10, 256u << 10, waves, bar);
      run<8>("private 256 KiB (256 x = 64 MiB, MALL)", buf, 256 << 10, 256u << 10, waves, bar);
      run<4>("private 4 MiB (1 GiB, HBM)", buf, 4 << 20, 4u << 20, waves, bar);
      run<8>("private 4 MiB (1 GiB, HBM)", buf, 4 << 20, 4u << 20, waves, bar);
    }
  }
  return 0;
}
