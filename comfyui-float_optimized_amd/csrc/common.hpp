// Shared device/host helpers for libfloat_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/float_hip.h"

// ---------------------------------------------------------------- error plumbing
void fh_set_error(const char* fmt, ...);
#define FH_CHECK_HIP(expr)                                                                   \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      fh_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return FLOAT_E_HIP;                                                                    \
    }                                                                                        \
  } while (0)
#define FH_REQUIRE(cond, ...)     \
  do {                            \
    if (!(cond)) {                \
      fh_set_error(__VA_ARGS__);  \
      return FLOAT_E_INVALID;     \
    }                             \
  } while (0)

// ---------------------------------------------------------------- 16-bit operand types
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // one 16-byte MFMA operand fragment
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

// Write-through (sc1) stores for what a kernel of the FMT step chain hands to the next launch.  A kernel ends with an
// agent-scope release that writes its dirty L2 lines back before the next dispatch may start; lines stored write-through are
// already on their way while the kernel still runs.  Relaxed agent-scope atomic stores are how HIP spells "store with sc1"
// (8 bytes at most per instruction).  Which store sites gain was measured by swapping libraries on one box, ms per 250
// evaluations (results bitwise the same): none 82.4-82.8; LayerNorm (1) 81.0-81.5; + attention (2) 81.0-81.3; + the 16-bit GEMM
// epilogues (8: qkv, fc1) 80.0-80.2 - the default, FMT_WT = 11; the fp32 epilogues do NOT gain: split-K slabs (4) 83.5-86.1,
// gate * residual (16) and x-embed (32) neutral; a 16-byte `global_store_dwordx4 ... sc1` by inline asm instead of two
// 8-byte stores 80.7-81.0.
#ifndef FMT_WT
#define FMT_WT 11  // bit mask of the store sites that write through: 1 LayerNorm, 2 attention, 4 / 16 / 32 fp32 GEMM epilogues (slabs / gate * residual / x-embed), 8 16-bit GEMM epilogues
#endif
// ON: per instantiation.  Stacked clips do not gain (720 rows: 176.0 ms per 250 evaluations with write-through everywhere,
// 170.9 without: four times the bytes in 8-byte pieces), so only the single-clip forms of the kernels pass ON = true: the
// LayerNorm / attention launches of at most 256 rows and the GEMM tilings with 8 or 16 K-splitting waves (stacked clips run
// 4).  A per-launch run-time switch instead cost the whole gain (both store paths in every epilogue: 83.6 vs 80.6 ms).
typedef __attribute__((address_space(1))) unsigned long long fh_gu64;
template <int SITE, bool ON>
__device__ __forceinline__ void fh_store8_wt(void* p, unsigned long long v) {
  if constexpr (ON && (FMT_WT & SITE) != 0) __hip_atomic_store((fh_gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *reinterpret_cast<unsigned long long*>(p) = v;
}
template <int SITE, bool ON>
__device__ __forceinline__ void fh_store16_wt(void* p, const u32x4& v) {
  if constexpr (ON && (FMT_WT & SITE) != 0) {
    fh_store8_wt<SITE, true>(p, ((unsigned long long)v[1] << 32) | v[0]);
    fh_store8_wt<SITE, true>(reinterpret_cast<char*>(p) + 8, ((unsigned long long)v[3] << 32) | v[2]);
  } else {
    *reinterpret_cast<u32x4*>(p) = v;
  }
}
template <int SITE, bool ON>
__device__ __forceinline__ void fh_store_f4_wt(float* p, const float4& v) {
  if constexpr (ON && (FMT_WT & SITE) != 0)
    fh_store16_wt<SITE, true>(p, u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)});
  else *reinterpret_cast<float4*>(p) = v;
}

// Every operand type T gives: elem (storage element), pack8 (8 consecutive elements = one lane's share of an MFMA fragment
// and the unit of the packed epilogue stores), load8 / load8_nt / store8 / store4 (+ _wt forms), get / set of one element as float, mfma.
#define FH_PACK16_HELPERS                                                                                                   \
  typedef u16 elem;                                                                                                         \
  typedef u32x4 pack8;                                                                                                      \
  static constexpr bool is32 = false;                                                                                       \
  static constexpr int EB = 2;  /* bytes per element; a pack8 is 8 * EB bytes */                                          \
  static __device__ __forceinline__ pack8 zero8() { return u32x4{0u, 0u, 0u, 0u}; }                                         \
  static __device__ __forceinline__ pack8 load8(const elem* p) { return *reinterpret_cast<const u32x4*>(p); }               \
  static __device__ __forceinline__ pack8 load8_nt(const elem* p) {                                                         \
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));                                                   \
  }                                                                                                                         \
  static __device__ __forceinline__ void store8(elem* p, const pack8& v) { *reinterpret_cast<u32x4*>(p) = v; }              \
  static __device__ __forceinline__ float get(const pack8& v, int j) { return to_float(reinterpret_cast<const u16*>(&v)[j]); } \
  static __device__ __forceinline__ void set(pack8& v, int j, float x) { reinterpret_cast<u16*>(&v)[j] = from_float(x); }   \
  static __device__ __forceinline__ void store4(elem* p, float a, float b, float c, float d) {                              \
    ushort4 o;                                                                                                              \
    o.x = from_float(a);                                                                                                    \
    o.y = from_float(b);                                                                                                    \
    o.z = from_float(c);                                                                                                    \
    o.w = from_float(d);                                                                                                    \
    *reinterpret_cast<ushort4*>(p) = o;                                                                                     \
  }                                                                                                                         \
  template <int SITE, bool ON>                                                                                              \
  static __device__ __forceinline__ void store8_wt(elem* p, const pack8& v) { fh_store16_wt<SITE, ON>(p, v); }              \
  template <int SITE, bool ON>                                                                                              \
  static __device__ __forceinline__ void store4_wt(elem* p, float a, float b, float c, float d) {                           \
    if constexpr (ON && (FMT_WT & SITE) != 0)                                                                               \
      fh_store8_wt<SITE, true>(p, (unsigned long long)from_float(a) | ((unsigned long long)from_float(b) << 16) |           \
                                      ((unsigned long long)from_float(c) << 32) | ((unsigned long long)from_float(d) << 48)); \
    else store4(p, a, b, c, d);                                                                                             \
  }

struct BF16 {
  typedef bf16x8_t vec8;
  static constexpr bool is_fp16 = false;
  static __device__ __forceinline__ u16 from_float(float x) {
    __bf16 b = (__bf16)x;  // RNE; hipcc emits v_cvt_pk_bf16_f32 on gfx950 (keeps NaN a NaN)
    return __builtin_bit_cast(u16, b);
  }
  static __device__ __forceinline__ float to_float(u16 v) {
    return __builtin_bit_cast(float, ((uint32_t)v) << 16);
  }
  static __device__ __forceinline__ f32x4 mfma(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vec8, a), __builtin_bit_cast(vec8, b), c, 0, 0, 0);
  }
  FH_PACK16_HELPERS
  static u16 host_from_float(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
  }
};

struct FP16 {
  typedef f16x8_t vec8;
  static constexpr bool is_fp16 = true;  // the range-tracked 16-bit stores of the decoder kernels (dec_store4) apply
  static __device__ __forceinline__ u16 from_float(float x) {
    // saturate instead of overflowing to inf: activations beyond 65504 would poison a frame
    x = fminf(fmaxf(x, -65504.f), 65504.f);
    _Float16 h = (_Float16)x;
    return __builtin_bit_cast(u16, h);
  }
  static __device__ __forceinline__ float to_float(u16 v) { return (float)__builtin_bit_cast(_Float16, v); }
  static __device__ __forceinline__ f32x4 mfma(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(vec8, a), __builtin_bit_cast(vec8, b), c, 0, 0, 0);
  }
  FH_PACK16_HELPERS
  static u16 host_from_float(float x) {
    // round-to-nearest-even fp32 -> fp16 on the host (weight packing)
    uint32_t u;
    memcpy(&u, &x, 4);
    uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t ax = u & 0x7fffffffu;
    if (ax > 0x7f800000u) return (u16)(sign | 0x7e00u);
    if (ax >= 0x477ff000u) return (u16)(sign | 0x7bffu);  // saturate to 65504
    if (ax < 0x33000001u) return (u16)sign;               // underflow to 0
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7fffffu) | 0x800000u;
    int shift;
    uint32_t base;
    if (e < -14) {  // subnormal half
      shift = 13 + (-14 - e);
      base = 0;
    } else {
      shift = 13;
      base = (uint32_t)(e + 15) << 10;
      m &= 0x7fffffu;
    }
    uint32_t r = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1u))) r++;
    return (u16)(sign | (base + r));
  }
};

// fp32 operands (FLOAT_DT_FP32): the verification mode of the FMT operator.  Same chain, same packed operand order with
// 4-byte elements; a lane's 8 consecutive k of a fragment feed 8 v_mfma_f32_16x16x4_f32 (lane group g supplies k = 8g + j in
// step j, the instruction sums its 4 groups: any assignment of k to groups works as long as A and B agree).  Exact fp32
// products and sums (an fmaf chain), 1/16 of the 16-bit MFMA rate - irrelevant for a mode that exists to hold the HIP logic
// to the reference at 1e-4.
struct F32x8 {
  f32x4 lo, hi;
};
struct FP32 {
  typedef float elem;
  static constexpr bool is_fp16 = false;
  typedef F32x8 pack8;
  static constexpr bool is32 = true;
  static constexpr int EB = 4;
  static __device__ __forceinline__ float from_float(float x) { return x; }
  static __device__ __forceinline__ float to_float(float v) { return v; }
  static float host_from_float(float x) { return x; }
  static __device__ __forceinline__ pack8 zero8() { return pack8{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}; }
  static __device__ __forceinline__ pack8 load8(const elem* p) {
    return pack8{*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4)};
  }
  static __device__ __forceinline__ pack8 load8_nt(const elem* p) { return load8(p); }
  static __device__ __forceinline__ void store8(elem* p, const pack8& v) {
    *reinterpret_cast<f32x4*>(p) = v.lo;
    *reinterpret_cast<f32x4*>(p + 4) = v.hi;
  }
  static __device__ __forceinline__ float get(const pack8& v, int j) { return j < 4 ? v.lo[j] : v.hi[j - 4]; }
  static __device__ __forceinline__ void set(pack8& v, int j, float x) {
    if (j < 4) v.lo[j] = x;
    else v.hi[j - 4] = x;
  }
  static __device__ __forceinline__ void store4(elem* p, float a, float b, float c, float d) {
    *reinterpret_cast<f32x4*>(p) = f32x4{a, b, c, d};
  }
  template <int SITE, bool ON>
  static __device__ __forceinline__ void store8_wt(elem* p, const pack8& v) { store8(p, v); }  // verification mode: plain stores
  template <int SITE, bool ON>
  static __device__ __forceinline__ void store4_wt(elem* p, float a, float b, float c, float d) { store4(p, a, b, c, d); }
  static __device__ __forceinline__ f32x4 mfma(const pack8& a, const pack8& b, f32x4 c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[j], b.lo[j], c, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[j], b.hi[j], c, 0, 0, 0);
    return c;
  }
};

// Coherent loads (COH = true): data another workgroup stored EARLIER IN THE SAME KERNEL (fmt_mega_kernel, whose stages meet at
// an in-kernel grid barrier instead of a launch boundary).  Such data is stored write-through (sc1) and must be read past the
// CU's vector L1 and this XCD's possibly stale L2 copy: agent-scope relaxed atomic loads are how HIP spells "load with sc1"
// (8 bytes at most per instruction), and unlike inline asm the compiler still schedules them and counts their waits.
template <bool COH>
__device__ __forceinline__ u32x4 fh_load16(const void* p) {
  if constexpr (COH) {
    const unsigned long long lo = __hip_atomic_load((const fh_gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load((const fh_gu64*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return u32x4{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
  } else {
    return *reinterpret_cast<const u32x4*>(p);
  }
}
template <bool COH>
__device__ __forceinline__ float4 fh_load_f4(const float* p) {
  if constexpr (COH) {
    const u32x4 v = fh_load16<true>(p);
    return float4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
  } else {
    return *reinterpret_cast<const float4*>(p);
  }
}
// store-site selector: inside the persistent kernel every activation store writes through, whatever FMT_WT says
// Read-once fp32 rows (the modulation rows of an evaluation: 37 MB per clip; split-K slabs on their only read): a
// non-temporal load, so that they do not displace what the launch chain re-reads (weights, the residual stream).
// MI355X, FMT sampling (two boxes, alternating libraries): 78.2-79.0 vs 79.0-79.4 ms per clip, 141.1 vs 145.0 per 4 clips,
// 391.7 vs 395.0 per 16 (modulation rows and gates: -2.3 % per 4 clips; the slabs another -0.4 %).
typedef float fh_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 fh_load_f4_stream(const float* p) {
  const fh_f4v t = __builtin_nontemporal_load(reinterpret_cast<const fh_f4v*>(p));
  return float4{t.x, t.y, t.z, t.w};
}
template <bool COH>
constexpr int fh_site(int site) { return COH ? 0xFF : site; }

// ---------------------------------------------------------------- range tracking of 16-bit stores (float_{fmt,enc,aud}_saturation)
// FP16::from_float clamps at +-65504 (a NaN becomes -65504): outside the decoder an out-of-range activation degrades the
// result instead of poisoning it, but it must not do so silently.  Every 16-bit activation store of the FMT, encoder and
// audio kernels goes through fh_cvt / fh_set / fh_store4 (or packs first and calls fh_track_pack): the thread keeps the
// running maximum of the stored half-precision magnitudes, two u16 lanes in one register (v_and_b32 + v_pk_max_u16 per two
// values, as dec_cvt2 does in the decoder), and fh_range_flush adds 1 to the handle's 64-bit device counter once per thread
// if a magnitude reached 0x7bff = 65504, the clamp value.  The counter counts threads with at least one clamped store -
// zero or not is what matters.  BF16 (fp32's exponent range) and FP32 instantiations compile to the plain conversions.
typedef unsigned short fh_us2 __attribute__((ext_vector_type(2)));
template <class T>
__device__ __forceinline__ void fh_track_u32(unsigned& m, unsigned packed2) {  // two packed halves
  if constexpr (T::is_fp16)
    m = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(fh_us2, m), __builtin_bit_cast(fh_us2, packed2 & 0x7fff7fffu)));
}
template <class T>
__device__ __forceinline__ void fh_track_pack(unsigned& m, const typename T::pack8& u) {
  if constexpr (T::is_fp16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) fh_track_u32<T>(m, u[i]);
  }
}
template <class T>
__device__ __forceinline__ typename T::elem fh_cvt(float x, unsigned& m) {
  const typename T::elem h = T::from_float(x);
  if constexpr (T::is_fp16) fh_track_u32<T>(m, (unsigned)h);
  return h;
}
template <class T>
__device__ __forceinline__ void fh_set(typename T::pack8& v, int j, float x, unsigned& m) {
  if constexpr (T::is_fp16) reinterpret_cast<u16*>(&v)[j] = fh_cvt<T>(x, m);
  else T::set(v, j, x);
}
template <class T>
__device__ __forceinline__ void fh_store4(typename T::elem* p, float a, float b, float c, float d, unsigned& m) {
  if constexpr (T::is_fp16) {
    uint2 o;
    o.x = (unsigned)T::from_float(a) | ((unsigned)T::from_float(b) << 16);
    o.y = (unsigned)T::from_float(c) | ((unsigned)T::from_float(d) << 16);
    fh_track_u32<T>(m, o.x);
    fh_track_u32<T>(m, o.y);
    *reinterpret_cast<uint2*>(p) = o;
  } else {
    T::store4(p, a, b, c, d);
  }
}
template <class T, int SITE, bool ON>
__device__ __forceinline__ void fh_store4_wt(typename T::elem* p, float a, float b, float c, float d, unsigned& m) {
  if constexpr (T::is_fp16) {
    const unsigned lo = (unsigned)T::from_float(a) | ((unsigned)T::from_float(b) << 16);
    const unsigned hi = (unsigned)T::from_float(c) | ((unsigned)T::from_float(d) << 16);
    fh_track_u32<T>(m, lo);
    fh_track_u32<T>(m, hi);
    if constexpr (ON && (FMT_WT & SITE) != 0) fh_store8_wt<SITE, true>(p, (unsigned long long)lo | ((unsigned long long)hi << 32));
    else *reinterpret_cast<uint2*>(p) = uint2{lo, hi};
  } else {
    T::template store4_wt<SITE, ON>(p, a, b, c, d);
  }
}
template <class T>
__device__ __forceinline__ void fh_range_flush(unsigned long long* ctr, unsigned m) {
  if constexpr (T::is_fp16) {
    if (ctr && ((m & 0xffffu) >= 0x7bffu || (m >> 16) >= 0x7bffu)) atomicAdd(ctr, 1ull);
  }
}

// ---------------------------------------------------------------- small device math
__device__ __forceinline__ float fh_silu(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float fh_sigmoid(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float fh_gelu_tanh(float x) {
  // 0.5 x (1 + tanh(u)) == x * sigmoid(2u): one v_exp + one v_rcp instead of libm tanhf (which cost the
  // fc1 epilogue 1.4 us per launch); exp overflow -> rcp(inf) = 0 -> 0, underflow -> x, both the right limits
  const float k0 = 0.7978845608028654f;  // sqrt(2/pi)
  const float u = k0 * (x + 0.044715f * x * x * x);
  return x * __frcp_rn(1.f + __expf(-2.f * u));
}
__device__ __forceinline__ float fh_gelu_erf(float x) {  // exact GELU: 0.5 x (1 + erf(x / sqrt(2)))
  return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
}
__device__ __forceinline__ float fh_lrelu_s2(float x) {  // leaky_relu(x, 0.2) * sqrt(2)
  return (x > 0.f ? x : 0.2f * x) * 1.4142135623730951f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------- host-side tensor table
struct TensorTable {
  std::map<std::string, const float_tensor_t*> m;
  TensorTable(const float_tensor_t* t, int n) {
    for (int i = 0; i < n; ++i) m[t[i].name] = &t[i];
  }
  const float_tensor_t* find(const std::string& k) const {
    auto it = m.find(k);
    return it == m.end() ? nullptr : it->second;
  }
  static int64_t numel(const float_tensor_t* t) {
    int64_t n = 1;
    for (int i = 0; i < t->ndim; ++i) n *= t->shape[i];
    return n;
  }
};

// Simple owning list of device allocations for a handle.
struct DevicePool {
  std::vector<void*> ptrs;
  size_t total = 0;
  template <typename T>
  int alloc(T** out, size_t count, bool zero = true) {
    void* p = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      fh_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
      return FLOAT_E_NOMEM;
    }
    if (zero) {
      e = hipMemset(p, 0, bytes);
      if (e != hipSuccess) {
        fh_set_error("hipMemset failed: %s", hipGetErrorString(e));
        return FLOAT_E_HIP;
      }
    }
    ptrs.push_back(p);
    total += bytes;
    *out = (T*)p;
    return FLOAT_OK;
  }
  void release() {
    for (void* p : ptrs) (void)hipFree(p);
    ptrs.clear();
  }
};

// Stream-ordered device-to-device copy of `bytes` bytes (a multiple of 4) as a KERNEL: memcpy nodes of a caller's stream
// capture did not replay reproducibly on ROCm 7.2 (fmt_copy_kernel), kernel nodes do.  misc.hip.
int fh_copy_d2d(void* dst, const void* src, size_t bytes, hipStream_t s);

// Direction (styledecoder.py:428-444) helpers shared by the encoder and the decoder operators (enc_api.hip):
// Q of the Householder QR of (W + 1e-8) with LAPACK sign conventions, and y = alpha * W x + b in fp32.
void fh_direction_q(const float* w, int dim, int dim_motion, std::vector<float>* Q);
int fh_linear_f32(const float* x, const float* W, const float* b, float alpha, float* y, int N, int K, hipStream_t s);

// Kernel-class profiling (float_profile_ms): events recorded on the caller's stream.
struct ProfileSlot {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
};
extern int g_fh_profiling;
bool fh_prof_pair(int which, hipEvent_t* start, hipEvent_t* stop);
void fh_prof_begin(int which, hipStream_t s);
void fh_prof_end(int which, hipStream_t s);
