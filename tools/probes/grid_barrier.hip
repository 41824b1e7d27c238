// Probe: a barrier across ALL 256 workgroups (one per CU) inside a kernel, hierarchical (8 group counters of 32 arrivals, the
// last arriver of a group bumps a top counter, the last of those writes 8 per-group release words), with the data of each
// stage exchanged ACROSS XCDs (a workgroup reads 4 KB written by workgroup b + 37 in the previous stage).
// Variants of the data path: SC1 = sc1 (agent-scope, write-through) stores + sc1 loads; FENCE = plain stores / loads bracketed
// by agent-scope release / acquire fences (buffer_wbl2 sc1 / buffer_inv sc1).  Compare with ~4.5 us per launch boundary.
// hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier && ./grid_barrier [stages] [KB of "weights" per stage]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
typedef __attribute__((address_space(1))) unsigned int gu32;
typedef float f4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 load_sc1(const float4* p) {
  f4v v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float4 load_sc0(const float4* p) {
  f4v v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store_sc1(float4* p, float4 v) {
  const f4v u = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(u) : "memory");
}

struct Bar {
  unsigned* grp;   // [8] stride 32 words (own 128-byte lines)
  unsigned* top;   // [1]
  unsigned* rel;   // [8] stride 32 words: release generation per group
};
// generation gen (1, 2, ...): every workgroup calls once per generation; thread 0 only
__device__ __forceinline__ bool grid_barrier(const Bar& b, unsigned gen, unsigned nwg, unsigned* info) {
  const unsigned g = blockIdx.x & 7u, per = nwg >> 3;
  const unsigned old = __hip_atomic_fetch_add((gu32*)(b.grp + g * 32), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (old + 1u == gen * per) {
    const unsigned t = __hip_atomic_fetch_add((gu32*)b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t + 1u == gen * 8u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) __hip_atomic_store((gu32*)(b.rel + i * 32), gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  unsigned spins = 0;
  while (__hip_atomic_load((gu32*)(b.rel + g * 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
    if (++spins > (1u << 22)) { atomicAdd(&info[2], 1u); return false; }
    __builtin_amdgcn_s_sleep(1);
  }
  return true;
}

template <int MODE /* 0 SC1, 1 FENCE, 2 barrier only, 3 sc1 stores + buffer_inv sc1 by one wave + plain loads, 4 same with the inv by every wave */>
__global__ __launch_bounds__(256) void chain(float4* buf, Bar bar, unsigned* info, const float4* W, int wvec, int S, unsigned gen0) {
  const int t = threadIdx.x, b = blockIdx.x, nwg = gridDim.x;
  float4 acc = make_float4(1.f, 2.f, 3.f, 4.f);
  float4 wsum = make_float4(0, 0, 0, 0);
  for (int s = 0; s < S; ++s) {
    // "weights" of the NEXT stage: independent of the hand-off, requested before the barrier wait (only their sum is used)
    const float4* w = W + ((size_t)((s & 7) * nwg + b)) * wvec;
    float4 wv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wv[i] = (t + i * 256 < wvec) ? w[t + i * 256] : make_float4(0, 0, 0, 0);
    const int src = (b + 37) % nwg;
    const float4* in = buf + ((size_t)(s & 1) * nwg + src) * 256 + t;
    float4* out = buf + ((size_t)((s + 1) & 1) * nwg + b) * 256 + t;
    if (MODE != 2) {
      float4 v = MODE == 0 ? load_sc1(in) : (MODE == 5 ? load_sc0(in) : *in);
      if (MODE == 5 || MODE == 6) {  // re-read operand: 16 more rows from other workgroups, each read by many workgroups of this XCD
        const float4* big = buf + (size_t)(s & 1) * nwg * 256;
        for (int i = 0; i < 16; ++i) {
          const float4* q = big + ((size_t)((b * 7 + i * 13) % nwg)) * 256 + t;
          const float4 u = MODE == 5 ? load_sc0(q) : load_sc1(q);
          v.z += u.x * 1e-3f;
        }
      }
      if (MODE >= 3) {  // the real stages re-read their operand from many workgroups: 64 KB more per workgroup, L2 hits after the first
        const float4* big = buf + (size_t)(s & 1) * nwg * 256;
        for (int i = 0; i < 16; ++i) { const float4 u = big[((size_t)((b * 7 + i * 13) % nwg)) * 256 + t]; v.z += u.x * 1e-6f; }
      }
      acc.x = acc.x * 0.5f + v.x * 0.5f + 1e-3f;
      acc.y = acc.y * 0.5f + v.y * 0.5f;
      acc.z += v.z * 1e-3f;
      acc.w = v.w;
      if (MODE == 0 || MODE >= 3) store_sc1(out, acc);  // 5: sc1 stores + sc0 (L1-bypassing, L2-cached) loads; 6: the same data path with sc1 loads
      else *out = acc;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { wsum.x += wv[i].x; wsum.y += wv[i].y; }
    if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) grid_barrier(bar, gen0 + (unsigned)s + 1u, (unsigned)nwg, info);
    __syncthreads();
    if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (MODE == 3 && t < 64) asm volatile("buffer_inv sc1" ::: "memory");
    if (MODE == 4) asm volatile("buffer_inv sc1" ::: "memory");
    if (MODE >= 3) __syncthreads();
  }
  if (t == 0 && b == 0) { info[4] = __float_as_uint(acc.x); info[5] = __float_as_uint(wsum.x + wsum.y); }
}

int main(int argc, char** argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 64, WKB = argc > 2 ? atoi(argv[2]) : 16, NWG = 256;
  const int wvec = WKB * 1024 / 16;
  float4 *buf, *W; unsigned *ctr, *info;
  CK(hipMalloc(&buf, (size_t)2 * NWG * 256 * 16)); CK(hipMalloc(&W, (size_t)8 * NWG * wvec * 16)); CK(hipMalloc(&ctr, 32 * 4 * 20)); CK(hipMalloc(&info, 64));
  CK(hipMemset(buf, 0, (size_t)2 * NWG * 256 * 16)); CK(hipMemset(W, 0, (size_t)8 * NWG * wvec * 16));
  Bar bar{ctr, ctr + 32 * 8, ctr + 32 * 9};
  const char* names[7] = {"SC1 loads/stores", "FENCE (wbl2 / inv)", "barrier only", "sc1 st + inv(1 wave)", "sc1 st + inv(all)", "sc1 st + sc0 ld x17", "sc1 st + sc1 ld x17"};
  for (int mode = 0; mode < 7; ++mode) {
    if (mode == 1 || mode == 3 || mode == 4) continue;
    for (int S2 : {S / 4, S}) {
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ctr, 0, 32 * 4 * 20)); CK(hipMemset(info, 0, 64)); CK(hipMemset(buf, 0, (size_t)2 * NWG * 256 * 16));
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::high_resolution_clock::now();
        if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(NWG), dim3(256), 0, 0, buf, bar, info, W, wvec, S2, 0u);
        else if (mode == 1) hipLaunchKernelGGL(chain<1>, dim3(NWG), dim3(256), 0, 0, buf, bar, info, W, wvec, S2, 0u);
        else if (mode == 2) hipLaunchKernelGGL(chain<2>, dim3(NWG), dim3(256), 0, 0, buf, bar, info, W, wvec, S2, 0u);
        else if (mode == 3) hipLaunchKernelGGL(chain<3>, dim3(NWG), dim3(256), 0, 0, buf, bar, info, W, wvec, S2, 0u);
        else if (mode == 4) hipLaunchKernelGGL(chain<4>, dim3(NWG), dim3(256), 0, 0, buf, bar, info, W, wvec, S2, 0u);
        else if (mode == 5) hipLaunchKernelGGL(chain<5>, dim3(NWG), dim3(256), 0, 0, buf, bar, info, W, wvec, S2, 0u);
        else hipLaunchKernelGGL(chain<6>, dim3(NWG), dim3(256), 0, 0, buf, bar, info, W, wvec, S2, 0u);
        CK(hipDeviceSynchronize());
        auto t1 = std::chrono::high_resolution_clock::now();
        unsigned h[16]; CK(hipMemcpy(h, info, 64, hipMemcpyDeviceToHost));
        const double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
        if (rep) printf("%-20s %3d stages %8.1f us = %.2f us/stage; timeouts %u; x %.9g\n", names[mode], S2, us, us / S2, h[2], __builtin_bit_cast(float, h[4]));
      }
    }
  }
  return 0;
}
