// Do LDS-DMA issue and MFMA issue of the two waves of a SIMD overlap?  One 512-thread workgroup per CU: waves 0-3 (one per SIMD)
// run a register-operand MFMA loop, waves 4-7 (their SIMD partners) stream 1-KiB LDS-DMA pieces from an L2-resident window
// behind a counted vmcnt.  Timed: MFMA waves alone, DMA waves alone, both together; and the same with ds_read_b128 bursts in
// place of the DMA.  If "both" ~ max(alone) the partner's issue is free; if ~ sum, a GEMM step costs DMA issue + MFMA whatever
// the wave roles (what fmt_gemm_rbs_kernel / fmt_gemm_dma_kernel measure: DESIGN.md section 6).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mfma_dma_overlap.hip -o build_ab/mfma_dma_overlap
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

// mode bits: 1 = MFMA waves work, 2 = partner waves issue LDS-DMA, 4 = partner waves issue ds_read_b128 bursts instead
// per outer iteration: MFMA waves issue 24 MFMAs (16x16x32: 384 pipe clocks); partner waves issue `pieces` DMA pieces / 12 reads
__global__ __launch_bounds__(512) void overlap_kernel(const char* __restrict__ src, int iters, int mode, int pieces, float* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (w < 4) {
    if (!(mode & 1)) return;
    u4 a, b;
    a.x = 0x3c003c00u + (lane & 7u);
    a.y = a.x + 1u, a.z = a.x + 2u, a.w = a.x + 3u;
    b = a + u4{5u, 6u, 7u, 8u};
    f4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += acc[j][0];
    if (t == 12345.f) *sink = t;
  } else {
    const int lw = w - 4;
    if (mode & 2) {
      unsigned off = (unsigned)(((size_t)blockIdx.x * 37 + lw * 11) * 1024) & ((1u << 20) - 1u);
      for (int it = 0; it < iters; ++it) {
        for (int p = 0; p < pieces; ++p) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off + lane * 16),
                                           (__attribute__((address_space(3))) void*)(lds + (lw * 16 + (p & 15)) * 1024), 16, 0, 0);
          off = (off + 4096u) & ((1u << 20) - 1u);
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (mode & 4) {
      u4 acc = u4{0u, 0u, 0u, 0u};
      for (int it = 0; it < iters; ++it) {
        u4 v[12];
#pragma unroll
        for (int p = 0; p < 12; ++p) asm volatile("ds_read_b128 %0, %1" : "=v"(v[p]) : "v"((unsigned)(lane * 16 + ((lw * 12 + p) & 63) * 1024)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int p = 0; p < 12; ++p) acc ^= v[p];
      }
      if ((acc.x ^ acc.y) == 0x12345u) *sink = 1.f;
    }
  }
}

int main() {
  char* buf;
  float* sink;
  CK(hipMalloc(&buf, (size_t)2 << 20));
  CK(hipMemset(buf, 0, (size_t)2 << 20));
  CK(hipMalloc(&sink, 64));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(overlap_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 4000;
  auto run = [&](int mode, int pieces) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0, nullptr));
      hipLaunchKernelGGL(overlap_kernel, dim3(256), dim3(512), 128 * 1024, nullptr, buf, iters, mode, pieces, sink);
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0) best = ms < best ? ms : best;
    }
    return best * 1e3f;  // us
  };
  const float mf = run(1, 0);
  printf("MFMA waves alone (24 MFMAs per iteration and SIMD): %8.1f us = %.0f clocks per iteration at 2.4 GHz\n", mf, mf * 2400.f / iters);
  for (int pieces : {2, 4, 7, 8}) {
    const float d = run(2, pieces), both = run(3, pieces);
    printf("LDS-DMA partner, %d pieces per iteration and wave: alone %8.1f us (%.0f clocks per iteration, %.0f per piece), with the MFMA waves %8.1f us "
           "(max %.1f, sum %.1f)\n", pieces, d, d * 2400.f / iters, d * 2400.f / iters / pieces, both, mf > d ? mf : d, mf + d);
  }
  const float rd = run(4, 0), both = run(5, 0);
  printf("ds_read_b128 partner, 12 reads per iteration and wave: alone %8.1f us, with the MFMA waves %8.1f us (max %.1f, sum %.1f)\n", rd, both,
         mf > rd ? mf : rd, mf + rd);
  return 0;
}
