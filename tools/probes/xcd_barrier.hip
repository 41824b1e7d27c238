// Probe: what does a hand-off BETWEEN WORKGROUPS OF ONE XCD cost inside a kernel?  (The FMT step chain pays ~4.5 us per launch
// boundary; tools/probes/ahead_chain.hip measured 6.7 us for an in-kernel wait on data produced on OTHER XCDs.)
// 256 workgroups, one per CU; a workgroup reads its XCC_ID, takes a member slot of that XCD, and runs S stages:
//   read 4 KB written in the previous stage by ANOTHER member of the same XCD -> trivial arithmetic -> write 4 KB ->
//   barrier among the XCD's members.
// Variants: LOCAL = atomics and loads that stop at the XCD's own L2 (atomics without sc1, data loads with sc0 = miss the CU's
// vector L1); AGENT = agent-scope (sc1) atomics, loads and stores, i.e. coherent across XCDs (what a cross-XCD hand-off needs).
// hipcc --offload-arch=gfx950 -O3 xcd_barrier.hip -o xcd_barrier && ./xcd_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
typedef __attribute__((address_space(1))) unsigned int gu32;

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_sc0(const float4* p) {
  f4v v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float4 load_sc1(const float4* p) {
  f4v v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store_sc1(float4* p, float4 v) {
  const f4v u = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(u) : "memory");
}

// buf: [2][8 XCDs][64 member slots][256 float4]; cnt: [S + 1][8]; slots: [8] member tickets; info: diagnostics
template <int MODE /* 0 LOCAL, 1 AGENT */>
__global__ __launch_bounds__(256) void chain(float4* buf, unsigned* cnt, unsigned* slots, unsigned* info, int S, int per_xcd) {
  __shared__ unsigned sh[2];
  const int t = threadIdx.x;
  if (t == 0) {
    const unsigned x = xcc_id();
    sh[0] = x;
    sh[1] = atomicAdd(&slots[x], 1u);  // default (agent) scope: fine, once per kernel
    if ((blockIdx.x & 7) != x) atomicAdd(&info[0], 1u);  // round-robin placement assumption violated
  }
  __syncthreads();
  const unsigned x = sh[0], me = sh[1];
  if (me >= (unsigned)per_xcd) {  // more members than expected on this XCD: bail out (the others would wait for ever otherwise)
    if (t == 0) atomicAdd(&info[1], 1u);
    return;
  }
  float4 acc = make_float4(1.f, 2.f, 3.f, 4.f);
  for (int s = 0; s < S; ++s) {
    const unsigned src = (me + 7u) % (unsigned)per_xcd;
    const float4* in = buf + ((size_t)((s & 1) * 8 + x) * 64 + src) * 256 + t;
    float4* out = buf + ((size_t)(((s + 1) & 1) * 8 + x) * 64 + me) * 256 + t;
    float4 v = MODE == 0 ? load_sc0(in) : load_sc1(in);
    acc.x = acc.x * 0.5f + v.x * 0.5f + 1e-3f;
    acc.y = acc.y * 0.5f + v.y * 0.5f;
    acc.z += v.z * 1e-3f;
    acc.w = v.w;
    if (MODE == 0) *out = acc;
    else store_sc1(out, acc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
      unsigned* c = cnt + (size_t)s * 8 + x;
      if (MODE == 0) {
        // atomics always execute in the L2; without sc1 they stop at THIS XCD's L2 (sc0 = return the old value)
        unsigned one = 1u, zero = 0u, seen;
        asm volatile("global_atomic_add %0, %1, off" ::"v"(c), "v"(one) : "memory");
        unsigned spins = 0;
        for (;;) {
          asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(seen) : "v"(c), "v"(zero) : "memory");
          if (seen >= (unsigned)per_xcd) break;
          if (++spins > (1u << 22)) { atomicAdd(&info[2], 1u); break; }
          __builtin_amdgcn_s_sleep(1);
        }
      } else {
        __hip_atomic_fetch_add((gu32*)c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load((gu32*)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)per_xcd) {
          if (++spins > (1u << 22)) { atomicAdd(&info[2], 1u); break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
    }
    __syncthreads();
  }
  if (t == 0 && me == 0) info[4 + x] = __float_as_uint(acc.x);
}

int main(int argc, char** argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 64, NWG = 256, PER = NWG / 8;
  float4* buf; unsigned *cnt, *slots, *info;
  CK(hipMalloc(&buf, (size_t)2 * 8 * 64 * 256 * 16)); CK(hipMalloc(&cnt, (size_t)(S + 1) * 8 * 4)); CK(hipMalloc(&slots, 32)); CK(hipMalloc(&info, 64));
  CK(hipMemset(buf, 0, (size_t)2 * 8 * 64 * 256 * 16));
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemset(cnt, 0, (size_t)(S + 1) * 8 * 4)); CK(hipMemset(slots, 0, 32)); CK(hipMemset(info, 0, 64));
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::high_resolution_clock::now();
      if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(NWG), dim3(256), 0, 0, buf, cnt, slots, info, S, PER);
      else hipLaunchKernelGGL(chain<1>, dim3(NWG), dim3(256), 0, 0, buf, cnt, slots, info, S, PER);
      CK(hipDeviceSynchronize());
      auto t1 = std::chrono::high_resolution_clock::now();
      unsigned h[16]; CK(hipMemcpy(h, info, 64, hipMemcpyDeviceToHost));
      unsigned sl[8]; CK(hipMemcpy(sl, slots, 32, hipMemcpyDeviceToHost));
      const double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
      if (rep) printf("%s: %d stages in %.1f us = %.2f us/stage (launch + sync included); misplaced %u, overflow %u, timeouts %u; members %u %u %u %u %u %u %u %u; x0 %g\n",
                      mode ? "AGENT (sc1)" : "LOCAL (L2 of the XCD)", S, us, us / S, h[0], h[1], h[2], sl[0], sl[1], sl[2], sl[3], sl[4], sl[5], sl[6], sl[7], __builtin_bit_cast(float, h[4]));
    }
  }
  // same number of stages as separate launches (plain) for reference
  return 0;
}
