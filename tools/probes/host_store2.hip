// Does a PCIe-bound host copy inside a launch slow the launch's device-memory traffic?  Grid = ncopy copy workgroups
// (spread over the XCDs, or all on one XCD: ids = 0 mod 8) + 4096 workgroups of device-to-device streaming.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stream(const u32x4* __restrict__ src, u32x4* __restrict__ dst, unsigned long long n16, unsigned wg,
                                       unsigned nwg) {
  const unsigned long long stride = (unsigned long long)nwg * 256;
  unsigned long long i = (unsigned long long)wg * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
    u32x4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}
// mode 0: copy ids are 0..ncopy-1; mode 1: copy ids are 0, 8, 16, .. (one XCD)
__global__ __launch_bounds__(256) void mixed(const u32x4* hsrc, u32x4* hdst, unsigned long long hn16, unsigned ncopy, int mode,
                                             const u32x4* dsrc, u32x4* ddst, unsigned long long dn16, unsigned ncomp) {
  unsigned x = blockIdx.x;
  if (mode == 0) {
    if (x < ncopy) { stream(hsrc, hdst, hn16, x, ncopy); return; }
    x -= ncopy;
  } else {
    if (x < 8 * ncopy) {
      if ((x & 7) == 0) { stream(hsrc, hdst, hn16, x >> 3, ncopy); return; }
      x -= (x >> 3) + 1;
    } else {
      x -= ncopy;
    }
  }
  stream(dsrc, ddst, dn16, x, ncomp);
}
int main() {
  const size_t hb = 11ull << 20, db = 2048ull << 20;
  void *d, *h, *a, *b;
  hipMalloc(&d, hb); hipMemset(d, 1, hb);
  hipHostMalloc(&h, hb, hipHostMallocDefault);
  hipMalloc(&a, db); hipMalloc(&b, db); hipMemset(a, 2, db);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const unsigned ncomp = 4096;
  for (unsigned ncopy : {0u, 4u, 8u, 16u}) for (int mode : {0, 1}) {
    if (ncopy == 0 && mode == 1) continue;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(mixed, dim3(ncopy + ncomp), dim3(256), 0, 0, (const u32x4*)d, (u32x4*)h, ncopy ? hb / 16 : 0, ncopy, mode,
                         (const u32x4*)a, (u32x4*)b, db / 16, ncomp);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    printf("copy wgs %2u (%s): %.3f ms for 2 GiB d2d (+ %zu MB to host)  -> d2d %.2f TB/s\n", ncopy, mode ? "one XCD" : "spread", ms, ncopy ? hb >> 20 : 0,
           2.0 * db / ms / 1e9);
  }
  // host copy alone for reference
  for (unsigned ncopy : {4u, 8u, 16u}) {
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(mixed, dim3(ncopy), dim3(256), 0, 0, (const u32x4*)d, (u32x4*)h, hb / 16, ncopy, 0, (const u32x4*)a, (u32x4*)b, 0ull, 1u);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    printf("host copy alone, %2u wgs: %.3f ms (%.1f GB/s)\n", ncopy, ms, hb / ms / 1e6);
  }
  return 0;
}
