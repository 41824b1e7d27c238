// Probe: wall cost of one dependent tiny kernel in a captured chain, idle vs with a second stream
// keeping some CUs busy (DVFS hypothesis).  hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
__global__ void tiny(float* p, int n) { int i = blockIdx.x*blockDim.x+threadIdx.x; if (i<n) p[i] = p[i]*1.0001f + 1.f; }
__global__ void burner(float* p, long iters) {
  float a = p[threadIdx.x], b = 1.0001f;
  for (long i = 0; i < iters; ++i) { a = a*b + 0.5f; b = b*0.9999f + 1e-6f; }
  p[blockIdx.x*blockDim.x+threadIdx.x] = a + b;
}
__global__ void clk(unsigned long long* out) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float a = 1.f; for (int i = 0; i < 200000; ++i) a = a*1.0001f+0.1f;
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[0] = t1-t0; out[1] = r1-r0; out[2] = (unsigned long long)a;
}
int main() {
  float *d, *d2; unsigned long long* dc; CK(hipMalloc(&d, 1<<20)); CK(hipMalloc(&d2, 64<<20)); CK(hipMalloc(&dc, 64));
  CK(hipMemset(d,0,1<<20)); CK(hipMemset(d2,0,64<<20));
  hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
  const int N = 2000;
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(48), dim3(256), 0, s, d, 48*256);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  auto run = [&](const char* tag) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipStreamSynchronize(s));
      auto t0 = std::chrono::high_resolution_clock::now();
      CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
      auto t1 = std::chrono::high_resolution_clock::now();
      printf("%s graph: %.2f us/kernel\n", tag, std::chrono::duration<double, std::micro>(t1-t0).count()/N);
    }
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(48), dim3(256), 0, s, d, 48*256);
    CK(hipStreamSynchronize(s));
    auto t1 = std::chrono::high_resolution_clock::now();
    printf("%s eager: %.2f us/kernel\n", tag, std::chrono::duration<double, std::micro>(t1-t0).count()/N);
    hipLaunchKernelGGL(clk, dim3(1), dim3(64), 0, s, dc); CK(hipStreamSynchronize(s));
    unsigned long long h[3]; CK(hipMemcpy(h, dc, 24, hipMemcpyDeviceToHost));
    printf("%s shader clock ~ %.0f MHz\n", tag, (double)h[0]/(double)h[1]*100.0);
    return 0;
  };
  run("idle");
  // keep 64 CUs busy on another stream for ~1 s
  hipLaunchKernelGGL(burner, dim3(64), dim3(256), 0, s2, d2, 400000000L);
  run("busy64");
  CK(hipStreamSynchronize(s2));
  hipLaunchKernelGGL(burner, dim3(1024), dim3(256), 0, s2, d2, 100000000L);
  run("busy1024");
  CK(hipDeviceSynchronize());
  run("after");
  return 0;
}
