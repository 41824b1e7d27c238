// GEMM lab for the stacked-clip step chain: the row-blocked LDS-DMA tile (fmt_gemm_rb_kernel) against the weight-streaming
// tiling (fmt_gemm_kernel<.., 3, 4, 4, ..>) on the chain's four shapes at 360 / 720 / 1440 / 2880 rows, checked against a
// naive fp32 GEMM over the same packed operands.  Weights rotate over 8 buffers per layer (as the 8 blocks of the model do:
// they come from the Infinity Cache, not from L2).  Random fp16 operands.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I comfyui-float_optimized_amd/csrc tools/probes/gemm_lab.hip -o gpurun_out/gemm_lab
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "fmt_rb_kernels.hpp"

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

// naive reference on the packed operands: out[row][n] = sum_k A(row, k) W(n, k)   (fp32 accumulate)
__global__ void ref_kernel(const u16* A, const u16* W, float* out, int M, int N, int K) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
  if (n >= N || row >= M) return;
  const int KB = K / 32;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc += FP16::to_float(A[fmt_pack_off(row, k, KB)]) * FP16::to_float(W[fmt_pack_off(n, k, KB)]);
  out[(size_t)row * N + n] = acc;
}

static float h2f(u16 v) { return (float)__builtin_bit_cast(_Float16, v); }

struct Shape {
  const char* name;
  int N, K, epi, ksplit;
};

template <class Kern>
float time_launches(Kern launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 16; ++i) launch(i);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, nullptr));
  for (int i = 0; i < reps; ++i) launch(i);
  CK(hipEventRecord(e1, nullptr));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return ms * 1e3f / reps;
}

template <int MI, int NJ, int KW, int NS, int EPI>
void launch_rb(GemmArgs g) {
  constexpr int smem = fmt_rb_smem(MI, NJ, KW, NS), ROWS = 32 * MI, BN = 32 * NJ;
  auto kern = fmt_gemm_rb_kernel<FP16, MI, NJ, KW, NS, EPI>;
  static bool primed = false;
  if (!primed) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    primed = true;
  }
  g.mblk = (g.M + ROWS - 1) / ROWS;
  if (EPI != EPI_PARTIAL) g.ksplit = 1;
  const int grid = (g.N / BN) * g.mblk * g.ksplit;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * KW), smem, nullptr, g);
}

template <int MI, int NJ, int KPS, int NS, int EPI>
void launch_rbs(GemmArgs g) {
  constexpr int smem = fmt_rb_smem(MI, NJ, KPS, NS), ROWS = 32 * MI, BN = 32 * NJ;
  auto kern = fmt_gemm_rbs_kernel<FP16, MI, NJ, KPS, NS, EPI>;
  static bool primed = false;
  if (!primed) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    primed = true;
  }
  g.mblk = (g.M + ROWS - 1) / ROWS;
  if (EPI != EPI_PARTIAL) g.ksplit = 1;
  const int grid = (g.N / BN) * g.mblk * g.ksplit;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, nullptr, g);
}

template <int EPI>
void launch_old(GemmArgs g) {
  constexpr int MTW = 3, NT = 4, NW = 4, smem = NW * MTW * 16 * NT * 16 * 4;
  auto kern = fmt_gemm_kernel<FP16, MTW, NT, NW, EPI>;
  static bool primed = false;
  if (!primed) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    primed = true;
  }
  g.mblk = ((g.M + 15) / 16 + MTW - 1) / MTW;
  if (EPI != EPI_PARTIAL) g.ksplit = 1;
  hipLaunchKernelGGL(kern, dim3((g.N / (NT * 16)) * g.mblk * g.ksplit), dim3(NW * 64), smem, nullptr, g);
}

template <int EPI>
void run_variant(const char* tag, int mi, int nj, GemmArgs g, const std::vector<u16*>& Ws, int reps, bool old) {
  auto launch = [&](int i) {
    GemmArgs gi = g;
    gi.W = Ws[i % Ws.size()];
    if (old) return launch_old<EPI>(gi);
    if (mi == 3 && nj == 4) launch_rb<3, 4, 2, 4, EPI>(gi);
    else if (mi == 3 && nj == 2) launch_rb<3, 2, 2, 4, EPI>(gi);
    else if (mi == 6 && nj == 4) launch_rb<6, 4, 2, 3, EPI>(gi);
    else if (mi == 3 && nj == 3) launch_rb<3, 3, 2, 4, EPI>(gi);
    else if (mi == 103 && nj == 4) launch_rbs<3, 4, 2, 4, EPI>(gi);  // specialised waves
    else if (mi == 103 && nj == 2) launch_rbs<3, 2, 2, 4, EPI>(gi);
    else if (mi == 106 && nj == 4) launch_rbs<6, 4, 2, 3, EPI>(gi);
  };
  const float us = time_launches(launch, reps);
  const double fl = 2.0 * g.M * g.N * g.K;
  printf("  %-28s %7.2f us  %7.1f TFLOP/s\n", tag, us, fl / us * 1e-6);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 200;
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  const int Mmax = 2880 + 192, Kmax = 4096, Nmax = 4096;
  std::vector<u16> hA((size_t)Mmax * Kmax), hW((size_t)Nmax * Kmax);
  for (auto& v : hA) v = FP16::host_from_float(U(rng));
  for (auto& v : hW) v = FP16::host_from_float(U(rng) * 0.05f);
  std::vector<float> hb(Nmax);
  for (auto& v : hb) v = U(rng);
  u16* dA;
  float *db, *dref, *dout;
  u16* dout16;
  CK(hipMalloc(&dA, hA.size() * 2));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
  std::vector<u16*> Ws(8);
  for (auto& p : Ws) {
    CK(hipMalloc(&p, hW.size() * 2));
    CK(hipMemcpy(p, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
  }
  CK(hipMalloc(&db, Nmax * 4));
  CK(hipMemcpy(db, hb.data(), Nmax * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dref, (size_t)Mmax * Nmax * 4));
  CK(hipMalloc(&dout, (size_t)8 * Mmax * Nmax * 4));
  CK(hipMalloc(&dout16, (size_t)Mmax * Nmax * 2));
  unsigned long long* dsat;
  CK(hipMalloc(&dsat, 8));
  CK(hipMemset(dsat, 0, 8));

  const Shape shapes[] = {{"qkv", 3072, 1024, EPI_T16, 1}, {"proj", 1024, 1024, EPI_PARTIAL, 4}, {"proj/2", 1024, 1024, EPI_PARTIAL, 2},
                          {"fc1", 4096, 1024, EPI_GELU_P16, 1}, {"fc2", 1024, 4096, EPI_PARTIAL, 4}, {"fc2/8", 1024, 4096, EPI_PARTIAL, 8}};
  const int only_m = argc > 2 ? atoi(argv[2]) : 0;
  for (int M : {360, 720, 1440, 2880}) {
    if (only_m && M != only_m) continue;
    for (const Shape& sh : shapes) {
      printf("M=%d %s (N=%d K=%d ksplit=%d)\n", M, sh.name, sh.N, sh.K, sh.ksplit);
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      g.A = dA;
      g.W = Ws[0];
      g.bias = db;
      g.K = sh.K;
      g.N = sh.N;
      g.M = M;
      g.sat = dsat;
      g.out_f32 = dout;
      g.ldo = sh.N;
      g.ksplit = sh.ksplit;
      g.slab_stride = (size_t)Mmax * sh.N;
      g.out16 = dout16;
      g.ldo16 = sh.epi == EPI_T16 ? sh.N : sh.N / 32;
      // ---- correctness of every variant against the naive GEMM
      hipLaunchKernelGGL(ref_kernel, dim3((sh.N + 255) / 256, M), dim3(256), 0, nullptr, dA, Ws[0], dref, M, sh.N, sh.K);
      CK(hipDeviceSynchronize());
      std::vector<float> ref((size_t)M * sh.N), got((size_t)M * sh.N);
      CK(hipMemcpy(ref.data(), dref, ref.size() * 4, hipMemcpyDeviceToHost));
      auto check = [&](const char* tag) {
        CK(hipDeviceSynchronize());
        double num = 0, den = 0, mx = 0;
        if (sh.epi == EPI_PARTIAL) {
          std::vector<float> slab((size_t)M * sh.N);
          std::fill(got.begin(), got.end(), 0.f);
          for (int k = 0; k < sh.ksplit; ++k) {
            CK(hipMemcpy(slab.data(), dout + (size_t)k * g.slab_stride, slab.size() * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < got.size(); ++i) got[i] += slab[i];
          }
          for (size_t i = 0; i < got.size(); ++i) {
            const double d = got[i] - ref[i];
            num += d * d, den += (double)ref[i] * ref[i], mx = std::max(mx, std::fabs(d));
          }
        } else {
          std::vector<u16> o16((size_t)(M + 16) * sh.N);
          CK(hipMemcpy(o16.data(), dout16, o16.size() * 2, hipMemcpyDeviceToHost));
          for (int r = 0; r < M; ++r)
            for (int n = 0; n < sh.N; ++n) {
              float x = ref[(size_t)r * sh.N + n] + hb[n], y;
              if (sh.epi == EPI_GELU_P16) {
                x = 0.5f * x * (1.f + tanhf(0.7978845608f * (x + 0.044715f * x * x * x)));
                y = h2f(o16[fmt_pack_off(r, n, sh.N / 32)]);
              } else {
                y = h2f(o16[(size_t)r * sh.N + n]);
              }
              const double d = y - x;
              num += d * d, den += (double)x * x, mx = std::max(mx, std::fabs(d));
            }
        }
        printf("  check %-22s rel-L2 %.2e  max %.2e %s\n", tag, std::sqrt(num / den), mx, std::sqrt(num / den) < 2e-3 ? "ok" : "FAIL");
      };
      auto variant = [&](const char* tag, int mi, int nj, bool old, int kskew = 0) {
        g.kskew = kskew;
        if (sh.N % (nj * 32)) return;
        const int kw = 2;
        if ((sh.K / 32) % (sh.ksplit * kw * (mi > 100 ? 2 : 1))) return;
        CK(hipMemset(dout, 0, (size_t)sh.ksplit * g.slab_stride * 4));
        CK(hipMemset(dout16, 0, (size_t)(M + 16) * sh.N * 2));
        std::vector<u16*> one(1, Ws[0]);
        if (sh.epi == EPI_T16) run_variant<EPI_T16>(tag, mi, nj, g, one, 1, old);
        else if (sh.epi == EPI_GELU_P16) run_variant<EPI_GELU_P16>(tag, mi, nj, g, one, 1, old);
        else run_variant<EPI_PARTIAL>(tag, mi, nj, g, one, 1, old);
        check(tag);
        if (sh.epi == EPI_T16) run_variant<EPI_T16>(tag, mi, nj, g, Ws, reps, old);
        else if (sh.epi == EPI_GELU_P16) run_variant<EPI_GELU_P16>(tag, mi, nj, g, Ws, reps, old);
        else run_variant<EPI_PARTIAL>(tag, mi, nj, g, Ws, reps, old);
      };
      variant("old 48x64 4w", 0, 4, true);
      variant("rb 96x128 8w", 3, 4, false);
      variant("rb 96x64 8w", 3, 2, false);
      variant("rb 96x96 8w", 3, 3, false);
      variant("rb 192x128 8w ns3", 6, 4, false);
      variant("rbs 96x128", 103, 4, false);
      variant("rbs 96x64", 103, 2, false);
      variant("rbs 192x128 ns3", 106, 4, false);
    }
  }
#ifdef RB_STAMPS
  {
    // per-workgroup timeline of the 96 x 128 tiling on fc1 at 720 rows (weights rotating as above)
    unsigned long long* dst;
    CK(hipMalloc(&dst, 4096 * 32));
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = dA, g.bias = db, g.K = 1024, g.N = 4096, g.M = 720, g.sat = dsat, g.out16 = dout16, g.ldo16 = 4096 / 32;
    g.vout = reinterpret_cast<float*>(dst);
    for (int kskew = 0; kskew < 2; ++kskew) {
      g.kskew = kskew;
      for (int i = 0; i < 20; ++i) {
        g.W = Ws[i % 8];
        launch_rbs<3, 4, 2, 4, EPI_GELU_P16>(g);
      }
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> st(256 * 4);
      CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull, t3 = 0;
      for (int b = 0; b < 256; ++b) t0 = std::min(t0, st[b * 4]), t3 = std::max(t3, st[b * 4 + 3]);
      printf("stamps kskew=%d: kernel span %.2f us\n", kskew, (t3 - t0) * 0.01);
      std::vector<double> a(256), b1(256), c(256), d(256);
      for (int b = 0; b < 256; ++b) {
        a[b] = (st[b * 4] - t0) * 0.01, b1[b] = (st[b * 4 + 1] - st[b * 4]) * 0.01, c[b] = (st[b * 4 + 2] - st[b * 4 + 1]) * 0.01,
        d[b] = (st[b * 4 + 3] - st[b * 4 + 2]) * 0.01;
      }
      auto pr = [&](const char* n, std::vector<double> v) {
        std::sort(v.begin(), v.end());
        printf("  %-22s min %.2f  med %.2f  max %.2f us\n", n, v[0], v[128], v[255]);
      };
      pr("start after first WG", a);
      pr("prologue (stage 0)", b1);
      pr("K loop (16 stages)", c);
      pr("epilogue", d);
    }
  }
#endif
  unsigned long long sat = 0;
  CK(hipMemcpy(&sat, dsat, 8, hipMemcpyDeviceToHost));
  printf("range counter %llu\n", sat);
  return 0;
}
