// Probe: a chain of small dependent kernels (the FMT step chain's regime: ~190 workgroups, a few KB in and out per workgroup,
// data produced on other XCDs) run three ways:
//   A  one stream, the launch boundary is the dependency (what the chain does today)
//   B  launched AHEAD on two alternating streams; kernel i+1 starts while kernel i runs, pre-loads its "weights" (independent
//      data), then waits in-kernel for kernel i's arrival counter (sc1 poll by one lane + barrier), reads kernel i's output with
//      sc1 loads, computes, stores sc1, drains, arrives.  At most two kernels are co-resident (stream order), so the wait cannot
//      deadlock as long as two grids fit the chip together.
//   C  the kernels of B on ONE stream (the in-kernel wait is then always satisfied): what the protocol itself costs.
// hipcc --offload-arch=gfx950 -O3 ahead_chain.hip -o ahead_chain && ./ahead_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
typedef __attribute__((address_space(1))) unsigned int gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;

// one stage: WG b reads rows [b*RPW, ..) of `in` (written by ANOTHER workgroup mapping of the previous stage: rotated by 37),
// adds the weights' checksum, writes `out`.  wbytes of weights per WG.
template <bool WAIT, bool SC1>
__global__ __launch_bounds__(256) void stage(const float* __restrict__ in, float* __restrict__ out, const float4* __restrict__ W,
                                             int wvec, unsigned* cnt_prev, unsigned* cnt_mine, unsigned expect, int nwg) {
  __shared__ float red[4];
  const int b = blockIdx.x, t = threadIdx.x;
  // weights: independent of the previous stage -> in flight before the wait
  float4 acc = make_float4(0, 0, 0, 0);
  const float4* w = W + (size_t)b * wvec;
  for (int i = t; i < wvec; i += 256) { float4 v = w[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  if (WAIT) {
    if (t == 0) {
      unsigned spins = 0;
      while (__hip_atomic_load((gu32*)cnt_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect && ++spins < (1u << 26)) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
  const int src = (b + 37) % nwg;  // produced by another workgroup (another XCD: 37 is odd)
  float x[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float* p = in + ((size_t)src * 1024 + k * 256 + t);
    if (SC1) x[k] = __builtin_bit_cast(float, __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    else x[k] = *p;
  }
  float s = x[0] + x[1] + x[2] + x[3];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  const float tot = (red[0] + red[1] + red[2] + red[3]) * (1.0f / 1024.f) + (acc.x + acc.y + acc.z + acc.w) * 1e-9f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float v = x[k] * 0.5f + tot * 0.5f + 1e-3f;
    float* p = out + ((size_t)b * 1024 + k * 256 + t);
    if (SC1) __hip_atomic_store((gu32*)p, __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
  }
  if (WAIT) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) __hip_atomic_fetch_add(cnt_mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main(int argc, char** argv) {
  const int NWG = argc > 1 ? atoi(argv[1]) : 192, N = 400, WKB = argc > 2 ? atoi(argv[2]) : 32;
  const int wvec = WKB * 1024 / 16;
  float *buf[2]; float4* W; unsigned* cnt;
  CK(hipMalloc(&buf[0], (size_t)NWG * 4096)); CK(hipMalloc(&buf[1], (size_t)NWG * 4096));
  CK(hipMalloc(&W, (size_t)NWG * WKB * 1024 * 8)); CK(hipMalloc(&cnt, (N + 1) * 4));
  CK(hipMemset(buf[0], 0, (size_t)NWG * 4096)); CK(hipMemset(W, 0, (size_t)NWG * WKB * 1024 * 8));
  hipStream_t s[2]; CK(hipStreamCreate(&s[0])); CK(hipStreamCreate(&s[1]));
  auto timeit = [&](const char* tag, auto fn) {
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemset(cnt, 0, (N + 1) * 4)); CK(hipMemset(cnt, 0xff, 4));  // stage 0 never waits
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::high_resolution_clock::now();
      fn();
      CK(hipDeviceSynchronize());
      auto t1 = std::chrono::high_resolution_clock::now();
      if (rep) printf("%-28s %.2f us/stage\n", tag, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    }
    return 0;
  };
  // weights alternate between two halves so consecutive stages stream different bytes (64 x NWG x WKB total = beyond L2)
  auto Wof = [&](int i) { return W + (size_t)(i % 8) * NWG * wvec; };
  timeit("A boundary, plain", [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL((stage<false, false>), dim3(NWG), dim3(256), 0, s[0], buf[i & 1], buf[(i + 1) & 1], Wof(i), wvec, cnt + i, cnt + i + 1, (unsigned)NWG, NWG); });
  timeit("A boundary, sc1", [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL((stage<false, true>), dim3(NWG), dim3(256), 0, s[0], buf[i & 1], buf[(i + 1) & 1], Wof(i), wvec, cnt + i, cnt + i + 1, (unsigned)NWG, NWG); });
  timeit("C in-kernel wait, 1 stream", [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL((stage<true, true>), dim3(NWG), dim3(256), 0, s[0], buf[i & 1], buf[(i + 1) & 1], Wof(i), wvec, cnt + i, cnt + i + 1, (unsigned)NWG, NWG); });
  timeit("B ahead, 2 streams", [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL((stage<true, true>), dim3(NWG), dim3(256), 0, s[i & 1], buf[i & 1], buf[(i + 1) & 1], Wof(i), wvec, cnt + i, cnt + i + 1, (unsigned)NWG, NWG); });
  // graph forms
  for (int two = 0; two < 2; ++two) {
    hipGraph_t g; hipGraphExec_t ge; hipEvent_t ef, ej; CK(hipEventCreate(&ef)); CK(hipEventCreate(&ej));
    CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
    if (two) { CK(hipEventRecord(ef, s[0])); CK(hipStreamWaitEvent(s[1], ef, 0)); }
    for (int i = 0; i < N; ++i) {
      hipStream_t st = s[two ? (i & 1) : 0];
      if (two) hipLaunchKernelGGL((stage<true, true>), dim3(NWG), dim3(256), 0, st, buf[i & 1], buf[(i + 1) & 1], Wof(i), wvec, cnt + i, cnt + i + 1, (unsigned)NWG, NWG);
      else hipLaunchKernelGGL((stage<false, true>), dim3(NWG), dim3(256), 0, st, buf[i & 1], buf[(i + 1) & 1], Wof(i), wvec, cnt + i, cnt + i + 1, (unsigned)NWG, NWG);
    }
    if (two) { CK(hipEventRecord(ej, s[1])); CK(hipStreamWaitEvent(s[0], ej, 0)); }
    CK(hipStreamEndCapture(s[0], &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    timeit(two ? "B ahead, graph 2 branches" : "A boundary, graph", [&] { hipGraphLaunch(ge, s[0]); });
  }
  float h; CK(hipMemcpy(&h, buf[0], 4, hipMemcpyDeviceToHost)); printf("check %g\n", h);
  return 0;
}
