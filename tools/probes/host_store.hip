// How fast can N workgroups of a kernel move device memory into pinned host memory (PCIe posted writes)?
// hipcc --offload-arch=gfx950 -O3 -o host_store host_store.hip && ./host_store
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, unsigned long long n16) {
  const unsigned long long stride = (unsigned long long)gridDim.x * 256;
  unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride);
      else dst[i + u * stride] = v[u];
    }
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}
int main() {
  const size_t bytes = 100ull << 20;
  void *d, *h;
  hipMalloc(&d, bytes);
  hipMemset(d, 1, bytes);
  hipHostMalloc(&h, bytes, hipHostMallocDefault);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemcpyAsync D2H: %.2f ms = %.1f GB/s\n", ms, bytes / ms / 1e6);
  }
  for (int nwg : {1, 2, 4, 8, 16, 32, 64, 128, 256, 1024}) {
    float ms[4];
    for (int v = 0; v < 4; ++v) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (v == 0) hipLaunchKernelGGL((copy_kernel<4, false>), dim3(nwg), dim3(256), 0, 0, (const u32x4*)d, (u32x4*)h, bytes / 16);
        if (v == 1) hipLaunchKernelGGL((copy_kernel<4, true>), dim3(nwg), dim3(256), 0, 0, (const u32x4*)d, (u32x4*)h, bytes / 16);
        if (v == 2) hipLaunchKernelGGL((copy_kernel<16, false>), dim3(nwg), dim3(256), 0, 0, (const u32x4*)d, (u32x4*)h, bytes / 16);
        if (v == 3) hipLaunchKernelGGL((copy_kernel<1, false>), dim3(nwg), dim3(256), 0, 0, (const u32x4*)d, (u32x4*)h, bytes / 16);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[v], e0, e1);
      }
    }
    printf("%4d WGs: unroll4 %.1f GB/s | unroll4 nt-store %.1f | unroll16 %.1f | unroll1 %.1f\n", nwg, bytes / ms[0] / 1e6,
           bytes / ms[1] / 1e6, bytes / ms[2] / 1e6, bytes / ms[3] / 1e6);
  }
  return 0;
}
