// Lab for fmt_gemm_big_kernel (persistent 256 x 256 tiles) against fmt_gemm_dma_kernel on the adaLN projection's shape
// (M = evaluations x rows, N = 51 200, K = 1 024, fp32 out), checked on sampled outputs against fp32 dot products.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I comfyui-float_optimized_amd/csrc -I tools/probes tools/probes/gemm_big_lab.hip -o build_ab/gemm_big_lab
// (see fmt_big_kernels.hpp for what was measured: a tie with fmt_gemm_dma_kernel in the pipeline)
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#ifndef BIG_NS
#define BIG_NS 4
#endif
#ifndef BIG_MI
#define BIG_MI 6
#endif
#include "fmt_big8_kernels.hpp"

void fh_set_error(const char*, ...) {}
int g_fh_profiling = 0;
#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)
static float h2f(u16 v) { return (float)__builtin_bit_cast(_Float16, v); }

// BIG_STRESS=<mode>: the kernel on a high-priority stream BESIDE a bandwidth-bound kernel on a low-priority stream (round 6: under
// FLOAT_AMD_OVERLAP=prio the FMT chain came out wrong from the second window on, and only with this kernel in it)
__global__ __launch_bounds__(256) void hammer_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n, int lds_bytes) {
  extern __shared__ unsigned char hl[];
  if (lds_bytes && threadIdx.x == 0) hl[0] = 1;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 50, rows = argc > 2 ? atoi(argv[2]) : 180;
  const int M = steps * rows, N = 51200, K = 1024, KB = K / 32;
  constexpr int MI = BIG_MI, RB = MI * 32;
  const int nrb = (M + RB - 1) / RB, Mp = nrb * RB;
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::vector<u16> hA((size_t)Mp * K), hW((size_t)N * K);
  for (auto& v : hA) v = FP16::host_from_float(U(rng));
  for (auto& v : hW) v = FP16::host_from_float(U(rng) * 0.05f);
  std::vector<float> hb(N);
  for (auto& v : hb) v = U(rng);
  u16 *dA, *dW;
  float *db, *dout;
  CK(hipMalloc(&dA, hA.size() * 2));
  CK(hipMalloc(&dW, hW.size() * 2));
  CK(hipMalloc(&db, N * 4));
  const size_t out_rows = std::max((size_t)Mp, (size_t)steps * 192 + 192);
  CK(hipMalloc(&dout, out_rows * N * 4));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dout, 0xff, out_rows * N * 4));
  BigArgs g{dA, dW, db, dout, M, N, K, N, nrb, N / 256};
#ifdef BIG4
  constexpr int NS = 4, smem = NS * 28 * 1024 + 4 * 4096, NTHR = 256;
  auto kern = fmt_gemm_big4_kernel<FP16, NS>;
#else
  constexpr int NS = BIG_NS, smem = NS * (2 * MI + 16) * 1024 + 8 * 4096, NTHR = 512;
  auto kern = fmt_gemm_big_kernel<FP16, MI, NS>;
#endif
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  int ncu = 0;
  CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
  const int grid = (ncu / 8) * 8;
  auto launch = [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), smem, nullptr, g); };
  launch();
  CK(hipDeviceSynchronize());
  // sampled check: 4000 random outputs + the corners of the last row block
  std::vector<float> out((size_t)M * 0 + 1);
  double num = 0, den = 0, mx = 0;
  std::mt19937 r2(7);
  int bad = 0;
  for (int t = 0; t < 4000; ++t) {
    int row = t < 3900 ? (int)(r2() % M) : (M - 1 - (int)(r2() % 16)), n = (int)(r2() % N);
    float got;
    CK(hipMemcpy(&got, dout + (size_t)row * N + n, 4, hipMemcpyDeviceToHost));
    double ref = hb[n];
    for (int k = 0; k < K; ++k) ref += (double)h2f(hA[fmt_pack_off(row, k, KB)]) * h2f(hW[fmt_pack_off(n, k, KB)]);
    const double d = got - ref;
    num += d * d, den += ref * ref, mx = std::max(mx, std::fabs(d));
    if (std::fabs(d) > 1e-2 || d != d) {
      if (bad < 12 && getenv("BIG_SHOW_BAD")) printf("  bad: row %d (%d in block %d) col %d (%d in block %d): got %g want %g\n", row, row % 192, row / 192, n, n % 256, n / 256, got, ref);
      ++bad;
    }
  }
  // rows >= M must be untouched (0xff pattern = NaN)
  float tail;
  CK(hipMemcpy(&tail, dout + (size_t)M * N + 5, 4, hipMemcpyDeviceToHost));
  printf("big kernel check: rel-L2 %.2e max %.2e bad %d; row M untouched: %s\n", std::sqrt(num / den), mx, bad, std::isnan(tail) || Mp == M ? "yes" : "NO");
  if (const char* e = getenv("BIG_STRESS")) {
    const int mode = atoi(e);  // 1: priorities, 2: two plain streams; BIG_STRESS_LDS = dynamic LDS bytes of the other kernel
    const int hlds = getenv("BIG_STRESS_LDS") ? atoi(getenv("BIG_STRESS_LDS")) : 0;
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t sh, sl;
    CK(hipStreamCreateWithPriority(&sh, hipStreamNonBlocking, mode == 1 ? hi : lo));
    CK(hipStreamCreateWithPriority(&sl, hipStreamNonBlocking, lo));
    uint4 *hs, *hd;
    const size_t hn = (size_t)1 << 26;  // 1 GiB each
    CK(hipMalloc(&hs, hn * 16));
    CK(hipMalloc(&hd, hn * 16));
    CK(hipMemset(hs, 1, hn * 16));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(hammer_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemset(dout, 0xff, out_rows * N * 4));
      CK(hipDeviceSynchronize());
      for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(hammer_kernel, dim3(2048), dim3(256), hlds, sl, hs, hd, hn, hlds);
      usleep(2000);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), smem, sh, g);
      CK(hipDeviceSynchronize());
      std::mt19937 r3(11 + rep);
      int nbad = 0;
      double worst = 0;
      for (int t = 0; t < 3000; ++t) {
        int row = (int)(r3() % M), n = (int)(r3() % N);
        float got;
        CK(hipMemcpy(&got, dout + (size_t)row * N + n, 4, hipMemcpyDeviceToHost));
        double ref = hb[n];
        for (int k = 0; k < K; ++k) ref += (double)h2f(hA[fmt_pack_off(row, k, KB)]) * h2f(hW[fmt_pack_off(n, k, KB)]);
        const double d = got - ref;
        if (!(std::fabs(d) <= 1e-2)) {
          if (nbad < 8) printf("  stress bad: row %d (%d in block %d) col %d (%d in block %d): got %g want %g\n", row, row % 192, row / 192, n, n % 256, n / 256, got, ref);
          ++nbad;
        }
        worst = std::max(worst, std::fabs(d));
      }
      printf("stress mode %d (other kernel's LDS %d B) rep %d: bad %d of 3000, worst %.3g\n", mode, hlds, rep, nbad, worst);
    }
    return 0;
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / 5;
    printf("fmt_gemm_big_kernel  M=%d: %8.1f us  %7.1f TFLOP/s (%.3f of 2.5 PFLOP/s)\n", M, us, 2.0 * M * N * K / us * 1e-6, 2.0 * M * N * K / us * 1e-6 / 2500);
  }
  if (const char* e = getenv("BIG_LOOP")) {  // keep the chip busy for a clock / power sample from outside (rocm-smi)
    const int n = atoi(e);
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < n; ++i) launch();
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("loop of %d launches: %8.1f us per launch\n", n, ms * 1e3 / n);
    return 0;
  }
  // the pipeline's regime: ONE launch per window between ~16 ms of a light launch chain (no power cap, the clock has to come up)
  auto spaced = [&](auto&& fn, const char* name) {
    double sum = 0, best = 1e30;
    for (int rep = 0; rep < 12; ++rep) {
      CK(hipDeviceSynchronize());
      usleep(15000);
      CK(hipEventRecord(e0, nullptr));
      fn();
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 2) sum += ms * 1e3, best = std::min(best, (double)ms * 1e3);
    }
    printf("%s, one launch every 15 ms: %8.1f us average, %8.1f best\n", name, sum / 10, best);
  };
  const bool do_spaced = getenv("SPACED") != nullptr;
  if (do_spaced) spaced(launch, "fmt_gemm_big_kernel");
  // the 192 x 320 one-tile-per-workgroup kernel on the padded layout (steps x 192 rows)
  {
    const int Mpad = 192;
    std::vector<u16> hA2((size_t)steps * Mpad * K + (size_t)12 * 16 * K, 0);
    for (int z = 0; z < steps; ++z)
      for (int r = 0; r < std::min(rows, Mpad); ++r)
        for (int k = 0; k < K; ++k) hA2[(size_t)z * Mpad * K + fmt_pack_off(r, k, KB)] = hA[fmt_pack_off(z * rows + r, k, KB)];
    u16* dA2;
    CK(hipMalloc(&dA2, hA2.size() * 2));
    CK(hipMemcpy(dA2, hA2.data(), hA2.size() * 2, hipMemcpyHostToDevice));
    GemmArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.A = dA2, ga.W = dW, ga.bias = db, ga.K = K, ga.N = N, ga.M = std::min(rows, Mpad), ga.out_f32 = dout, ga.ldo = N;
    ga.zcount = steps, ga.zgroup = 2, ga.a_zstride = (size_t)Mpad * K, ga.o_zstride = (size_t)Mpad * N, ga.mblk = 1;
    constexpr int smem2 = 4 * 32 * 1024;
    auto k2 = fmt_gemm_dma_kernel<FP16, 4, 4, 1>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, smem2));
    if (const char* e = getenv("DMA_LOOP")) {  // the library kernel under the same outside sample
      const int n = atoi(e);
      CK(hipEventRecord(e0, nullptr));
      for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k2, dim3((N / 320) * steps), dim3(512), smem2, nullptr, ga);
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("loop of %d launches of fmt_gemm_dma_kernel: %8.1f us per launch\n", n, ms * 1e3 / n);
      return 0;
    }
    if (do_spaced) spaced([&] { hipLaunchKernelGGL(k2, dim3((N / 320) * steps), dim3(512), smem2, nullptr, ga); }, "fmt_gemm_dma_kernel");
    if (rows <= 192) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k2, dim3((N / 320) * steps), dim3(512), smem2, nullptr, ga);
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / 5;
        printf("fmt_gemm_dma_kernel  M=%d: %8.1f us  %7.1f TFLOP/s\n", M, us, 2.0 * M * N * K / us * 1e-6);
      }
    }
  }
  return 0;
}
