// HIP kernels of the Synthesis decoder (reference styledecoder.py:195-425, 497-534), batched over
// frames.  Activations are NHWC in the operand type T (FP16 in production; FP32 = the verification mode, the same kernels
// with 4-byte elements on v_mfma_f32_16x16x4_f32); accumulation, style/demod, flow and rgb pyramids are fp32 in every mode.
// Every kernel is written against T::elem / T::pack8 (8 consecutive channels = one lane's share of an MFMA fragment, 8 * EB
// bytes); an LDS row holds 32 channels = 4 packs, the XOR swizzles permute packs, so the layouts carry over unchanged.
//
// ModulatedConv2d is evaluated as "scale the input channels by the style, convolve with the SHARED
// weight, scale the output channels by the demodulation factor" (identical algebra to modulating
// the weight per sample, styledecoder.py:241-246), so one weight tensor serves every frame of the
// batch and the conv becomes an implicit GEMM with M = frames*pixels.
#pragma once
#include "common.hpp"
#include <type_traits>

// ------------------------------------------------------------------------------------------
// Range check of the fp16 mode (float_dec_saturation).  fp16 ends at 65504; a checkpoint whose activations leave that range
// must not come out as plausible-looking wrong frames.  Every 16-bit activation store of the decoder goes through dec_store4 /
// dec_pack8: the values are converted WITHOUT a clamp (an overflow becomes inf and poisons what it touches - loud, not
// silent), and the thread keeps the per-half maximum of the packed results' magnitudes (v_and_b32 + v_pk_max_u16 per two
// values: what the clamp's v_med3_f32 per value used to cost, so the check is free); once per output tile dec_sat_flush adds
// 1 to the site's 64-bit device counter if a half reached the inf / NaN encodings (>= 0x7c00).  The counter counts
// (thread, tile) groups of 16..128 outputs with at least one such value - zero or not is what matters.  Cost ladder, ms per
// 250 frames on one box: exact count with a compare + branch per 4 values 30.2 vs 28.1 without; running fp32 max
// (v_max3_f32) + clamp 27.65 vs 26.80; this form: DESIGN.md.  FP32 instantiations compile to the plain store.
constexpr int kDecSatSites = 40;  // [0..15] StyledConv outputs (index = conv), [16..23] flow kernel of level li, 32 constant input, 33 skip features, 34 non-finite styles
typedef unsigned short dec_us2 __attribute__((ext_vector_type(2)));
typedef _Float16 dec_h2 __attribute__((ext_vector_type(2)));
template <class T>
__device__ __forceinline__ unsigned dec_cvt2(float a, float b, unsigned& m) {  // two floats -> packed fp16, range-tracked
  static_assert(T::is_fp16, "fp16 path");
  const unsigned u = __builtin_bit_cast(unsigned, dec_h2{(_Float16)a, (_Float16)b});
#ifndef DEC_NO_SAT  // A/B build only (make EXTRA=-DDEC_NO_SAT): what the tracking costs
  m = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(dec_us2, m), __builtin_bit_cast(dec_us2, u & 0x7fff7fffu)));
#endif
  return u;
}
template <class T>
__device__ __forceinline__ void dec_sat_flush(unsigned long long* ctr, unsigned& m) {
  if constexpr (T::is_fp16) {
    if ((m & 0xffffu) >= 0x7c00u || (m >> 16) >= 0x7c00u) atomicAdd(ctr, 1ull);
    m = 0u;
  }
}
template <class T>
__device__ __forceinline__ void dec_store4(typename T::elem* p, float a, float b, float c, float d, unsigned& m) {
  if constexpr (!T::is_fp16) {  // fp32 (verification mode), bf16 (the encoder borrows dec_conv16_kernel): plain stores
    T::store4(p, a, b, c, d);
  } else {
    uint2 o;
    o.x = dec_cvt2<T>(a, b, m);
    o.y = dec_cvt2<T>(c, d, m);
    *reinterpret_cast<uint2*>(p) = o;
  }
}
template <class T>
__device__ __forceinline__ typename T::pack8 dec_pack8(const float (&v)[8], unsigned& m) {
  if constexpr (!T::is_fp16) {
    typename T::pack8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) T::set(o, i, v[i]);
    return o;
  } else {
    return u32x4{dec_cvt2<T>(v[0], v[1], m), dec_cvt2<T>(v[2], v[3], m), dec_cvt2<T>(v[4], v[5], m), dec_cvt2<T>(v[6], v[7], m)};
  }
}

// Packs addressed as a uniform base plus a 32-bit BYTE offset: the global_load / global_store take the base in scalar registers
// and the offset as it is (an element offset costs a 64-bit shift-and-add per access: 38 of the flow kernel's 800 vector
// instructions per iteration).
template <class T>
__device__ __forceinline__ typename T::pack8 dec_load8_b(const void* base, unsigned byte_off) {
  return T::load8(reinterpret_cast<const typename T::elem*>(reinterpret_cast<const unsigned char*>(base) + byte_off));
}
template <class T>
__device__ __forceinline__ typename T::pack8 dec_load8_b_nt(const void* base, unsigned byte_off) {
  return T::load8_nt(reinterpret_cast<const typename T::elem*>(reinterpret_cast<const unsigned char*>(base) + byte_off));
}
template <class T>
__device__ __forceinline__ void dec_store8_b(void* base, unsigned byte_off, const typename T::pack8& v) {
  T::store8(reinterpret_cast<typename T::elem*>(reinterpret_cast<unsigned char*>(base) + byte_off), v);
}
// acc[i] += w * v[i] for the 8 elements of a pack.  fp16: v_fma_mix_f32 converts the half inside the FMA - one instruction per
// element where the compiler's v_cvt_f32_f16 + half a v_pk_fma_f32 is 1.5 (it prefers those; right for elements used by several
// FMAs, wrong for a bilinear tap, which is used once).  Same single rounding as the fused form it replaces.
template <class T>
__device__ __forceinline__ void dec_scale8(float (&acc)[8], const typename T::pack8& v, float w) {  // acc[i] = w * v[i]
  if constexpr (T::is_fp16) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(acc[2 * d]) : "v"(v[d]), "v"(w));
      asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(acc[2 * d + 1]) : "v"(v[d]), "v"(w));
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = w * T::get(v, i);
  }
}
template <class T>
__device__ __forceinline__ void dec_axpy8(float (&acc)[8], const typename T::pack8& v, float w) {
  if constexpr (T::is_fp16) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[2 * d]) : "v"(v[d]), "v"(w));
      asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[2 * d + 1]) : "v"(v[d]), "v"(w));
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += w * T::get(v, i);
  }
}

// ------------------------------------------------------------------------------------------
// out[f][j] = epi( sum_k in[f][k]^(1|2) * Wt[k][j] ), fp32, Wt stored k-major so lanes read
// consecutive j.  Used for the 22 EqualLinear style modulations (styledecoder.py:229,241) in one
// launch and for the demodulation factors rsqrt(scale^2 * sum_i s_i^2 * Wsq[o][i] + 1e-8)
// (styledecoder.py:244-245 with the weight factored out).
enum { SG_STYLE = 0, SG_DEMOD = 1 };
template <int MODE, int FB>
__global__ __launch_bounds__(256) void dec_small_gemm_kernel(const float* __restrict__ in, int ld_in, const float* __restrict__ in2,
                                                             const float* __restrict__ Wt, int K, int N,
                                                             const float* __restrict__ bias, float alpha,
                                                             float* __restrict__ out, int ld_out, int F) {
  extern __shared__ float sin_[];  // [FB][K]
  const int f0 = blockIdx.y * FB;
  for (int i = threadIdx.x; i < FB * K; i += 256) {
    const int fl = i / K, k = i % K;
    float v = 0.f;
    if (f0 + fl < F) {
      v = in[(size_t)(f0 + fl) * ld_in + k];
      if (MODE == SG_STYLE && in2) v += in2[k];  // latent = s_r + r_d[t]  (FLOAT.py:158)
      if (MODE == SG_DEMOD) v = v * v;
    }
    sin_[i] = v;
  }
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  float acc[FB];
#pragma unroll
  for (int f = 0; f < FB; ++f) acc[f] = 0.f;
  for (int k = 0; k < K; ++k) {
    const float w = Wt[(size_t)k * N + j];
#pragma unroll
    for (int f = 0; f < FB; ++f) acc[f] += sin_[f * K + k] * w;
  }
#pragma unroll
  for (int f = 0; f < FB; ++f) {
    if (f0 + f >= F) break;
    float v;
    if (MODE == SG_STYLE) v = acc[f] * alpha + bias[j];
    else v = rsqrtf(acc[f] * alpha + 1e-8f);
    out[(size_t)(f0 + f) * ld_out + j] = v;
  }
}

// Demodulation factors of every StyledConv in ONE launch (blockIdx.z = layer):
// d[f][o] = rsqrt( 1/(Cin*9) * sum_i s[f][i]^2 * Wsq[o][i] + 1e-8 )   (styledecoder.py:244-245)
//
// Range: the producers store x * s (activation times the consumer's style) in the operand type, and fp16 ends at 65504.  A
// demodulated conv does not depend on the scale of its style - with s = a s',
//   rsqrt(sum (W s)^2 + eps) * sum W s x  ==  rsqrt(sum (W s')^2 + eps / a^2) * sum W s' x
// exactly - so dec_style_norm_kernel divides every StyledConv's style by a = max |s| (per frame and layer, in place, before
// anybody reads it) and leaves eps / a^2 for the demodulation: what is stored is x * s' with |s'| <= 1, whatever the checkpoint's
// modulation weights are (the remedy StyleGAN2-ADA uses for its fp16 layers).  ToFlow's style is not normalised: that 1x1
// conv has no demodulation and is evaluated in fp32 (dec_flow_kernel).
struct DemodLayer {
  const float* WsqT;  // [Cin][Cout]
  int cin, cout, style_off, demod_off;
};
struct DemodArgs {
  DemodLayer L[16];
  float* styles;      // normalised in place by dec_style_norm_kernel
  float* eps;         // [F][16]: 1e-8 / a^2 per (frame, layer)
  float* demod;
  int ld_s, ld_d, F;
  int normalise;      // 0: a = 1 (A/B switch, FLOAT_DEC_STYLE_NORM=0)
  unsigned long long* sat;  // counters base (site 34: non-finite styles) or nullptr
};
static __global__ __launch_bounds__(256) void dec_style_norm_kernel(DemodArgs g) {  // grid (layers, frames)
  const DemodLayer L = g.L[blockIdx.x];
  const int f = blockIdx.y;
  float* s = g.styles + (size_t)f * g.ld_s + L.style_off;
  __shared__ float red[4];
  float m = 0.f;
  for (int i = threadIdx.x; i < L.cin; i += 256) m = fmaxf(m, fabsf(s[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;  // NaN / inf anywhere in the segment makes this non-finite (fmaxf skips NaN)
  for (int i = threadIdx.x; i < L.cin; i += 256) sum += s[i] * 0.f;
  if (sum != 0.f && g.sat) atomicAdd(g.sat + 34, 1ull);  // 0 * finite == 0; NaN != 0
  if (!g.normalise || !(m > 0.f) || !(m < 3e38f)) m = 1.f;  // all-zero / non-finite styles: leave them alone
  const float inv = 1.f / m;
  for (int i = threadIdx.x; i < L.cin; i += 256) s[i] *= inv;
  if (threadIdx.x == 0) g.eps[f * 16 + blockIdx.x] = 1e-8f * inv * inv;
}
template <int FB>
__global__ __launch_bounds__(256) void dec_demod_all_kernel(DemodArgs g) {
  extern __shared__ float s2[];  // [FB][cin]
  const DemodLayer L = g.L[blockIdx.z];
  if ((int)blockIdx.x * 256 >= L.cout) return;
  const int f0 = blockIdx.y * FB;
  for (int i = threadIdx.x; i < FB * L.cin; i += 256) {
    const int fl = i / L.cin, k = i - fl * L.cin;
    float v = 0.f;
    if (f0 + fl < g.F) v = g.styles[(size_t)(f0 + fl) * g.ld_s + L.style_off + k];
    s2[i] = v * v;
  }
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= L.cout) return;
  float acc[FB];
#pragma unroll
  for (int f = 0; f < FB; ++f) acc[f] = 0.f;
  for (int k = 0; k < L.cin; ++k) {
    const float w = L.WsqT[(size_t)k * L.cout + j];
#pragma unroll
    for (int f = 0; f < FB; ++f) acc[f] += s2[f * L.cin + k] * w;
  }
  const float alpha = 1.0f / (float)(L.cin * 9);
#pragma unroll
  for (int f = 0; f < FB; ++f)
    if (f0 + f < g.F)
      g.demod[(size_t)(f0 + f) * g.ld_d + L.demod_off + j] = 1.0f / sqrtf(acc[f] * alpha + g.eps[(f0 + f) * 16 + blockIdx.z]);
}

// ConstantInput repeated over the batch and pre-scaled by conv1's style (styledecoder.py:289-299, 513-514).
template <class T>
__global__ void dec_input_kernel(typename T::elem* __restrict__ out, const float* __restrict__ cin_hwc, const float* __restrict__ s,
                                 int ld_s, int F, int HW, int C, unsigned long long* sat) {
  const int idx = (blockIdx.x * blockDim.x + threadIdx.x) * 4;  // C % 4 == 0: 4 consecutive channels of one pixel
  if (idx >= F * HW * C) return;
  const int c = idx % C, f = idx / (HW * C);
  const float4 x = *reinterpret_cast<const float4*>(cin_hwc + idx % (HW * C));
  const float4 sv = *reinterpret_cast<const float4*>(s + (size_t)f * ld_s + c);
  unsigned sm = 0u;
  dec_store4<T>(out + idx, x.x * sv.x, x.y * sv.y, x.z * sv.z, x.w * sv.w, sm);
  dec_sat_flush<T>(sat, sm);
}

// ------------------------------------------------------------------------------------------
// Implicit-GEMM convolution over a tap list.  M = output pixels (256 per workgroup: nf frames x
// th x tw), N = BN output channels, K = taps x Cin in chunks of 32 input channels.  The input halo
// tile and the weight slab of the chunk are staged in LDS once and re-used by every tap.
// Device-to-host copy that rides along a compute launch (float_dec_frames_host): the first `nwg` workgroups of
// the grid (x < nwg, y == 0; dispatched first) move n16 16-byte units from device memory to pinned host memory while the
// rest of the grid computes, so the frames of batch i cross PCIe under the kernels of batch i+1 WITHOUT a second stream (a
// copy issued on another stream left the FMT chain that followed 12 ms slower per clip).  nwg is a multiple of 8, so the
// XCD classes (id mod 8) of the compute workgroups are unchanged.
struct CopyTail {
  const u32x4* src;
  u32x4* dst;
  unsigned long long n16;
  unsigned nwg;
  unsigned pace;  // 0: as fast as the stores are accepted
};

__device__ __forceinline__ void dec_copy_tail(const CopyTail& ct, unsigned wg) {
  const unsigned long long stride = (unsigned long long)ct.nwg * 256;
  unsigned long long i = (unsigned long long)wg * 256 + threadIdx.x;
  if (ct.pace) {
    // paced: one 1-KiB store per wave, then a pause of pace x 512 clocks, so that the copy stays below the PCIe rate and
    // its posted writes do not pile up in front of the compute workgroups' memory traffic
    for (; i < ct.n16; i += stride) {
      ct.dst[i] = __builtin_nontemporal_load(ct.src + i);
      for (unsigned k = 0; k < ct.pace; ++k) __builtin_amdgcn_s_sleep(8);
    }
    return;
  }
  for (; i + 3 * stride < ct.n16; i += 4 * stride) {  // 4 loads in flight per lane
    const u32x4 a = __builtin_nontemporal_load(ct.src + i), b = __builtin_nontemporal_load(ct.src + i + stride);
    const u32x4 c = __builtin_nontemporal_load(ct.src + i + 2 * stride), d = __builtin_nontemporal_load(ct.src + i + 3 * stride);
    ct.dst[i] = a;
    ct.dst[i + stride] = b;
    ct.dst[i + 2 * stride] = c;
    ct.dst[i + 3 * stride] = d;
  }
  for (; i < ct.n16; i += stride) ct.dst[i] = __builtin_nontemporal_load(ct.src + i);
}
#ifdef DEC_STAMPS
// diagnostic build only (make CXXFLAGS+=-DDEC_STAMPS): wall-clock (100 MHz) of the copy workgroups and of the compute
// workgroups of the LAST carrying launch: [copy first start, copy last end, compute first start, compute last end]
__device__ unsigned long long g_dec_stamps[4];
#define DEC_STAMP_MIN(i) if (threadIdx.x == 0) atomicMin(&g_dec_stamps[i], __builtin_amdgcn_s_memrealtime())
#define DEC_STAMP_MAX(i) if (threadIdx.x == 0) atomicMax(&g_dec_stamps[i], __builtin_amdgcn_s_memrealtime())
#else
#define DEC_STAMP_MIN(i)
#define DEC_STAMP_MAX(i)
#endif
// Where the lanes of a partial tile store instead of branching around the store (dec_zblur_kernel: the filter loop stays one
// basic block, so the compiler can run the LDS reads of the following rows ahead of the current row's arithmetic): 16 bytes per
// lane, never read.
static __device__ unsigned long long g_dec_sink[64 * 2];

// diagnostic build only (make EXTRA=-DDEC_PHASES): 10-ns wall clock spent by wave 0 of every workgroup between phase marks,
// summed per phase: [0..7] dec_zblur_kernel, [8..15] dec_conv16_kernel (float_dec_debug_phases)
#ifdef DEC_PHASES
__device__ unsigned long long g_dec_phase[64 * 16];  // 64 replicas (workgroup id mod 64) x 16 slots
#define DEC_PH_BEGIN                                              \
  unsigned long long ph_t = __builtin_amdgcn_s_memrealtime();    \
  unsigned ph_acc[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#define DEC_PH(i)                                                      \
  do {                                                                 \
    if (threadIdx.x == 0) {                                            \
      const unsigned long long ph_n = __builtin_amdgcn_s_memrealtime(); \
      ph_acc[(i) & 7] += (unsigned)(ph_n - ph_t);                      \
      ph_t = ph_n;                                                     \
    }                                                                  \
  } while (0)
#define DEC_PH_COUNT(i) do { if (threadIdx.x == 0) ph_acc[(i) & 7] += 1u; } while (0)
#define DEC_PH_END(base)                                                                                   \
  do {                                                                                                     \
    if (threadIdx.x == 0) {                                                                                \
      for (int ph_i = 0; ph_i < 8; ++ph_i)                                                                 \
        if (ph_acc[ph_i]) atomicAdd(&g_dec_phase[(blockIdx.x & 63) * 16 + (base) + ph_i], (unsigned long long)ph_acc[ph_i]); \
    }                                                                                                      \
  } while (0)
#else
#define DEC_PH_BEGIN
#define DEC_PH(i)
#define DEC_PH_COUNT(i)
#define DEC_PH_END(base)
#endif
// at the top of a kernel that may carry a copy: the copy workgroups do their share and leave; BID = the compute block id
#define DEC_COPY_PROLOGUE(g, BID)                                   \
  if (blockIdx.x < (g).ct.nwg) {                                    \
    if (blockIdx.y == 0) {                                          \
      DEC_STAMP_MIN(0);                                             \
      dec_copy_tail((g).ct, blockIdx.x);                            \
      __syncthreads();                                              \
      DEC_STAMP_MAX(1);                                             \
    }                                                               \
    return;                                                         \
  }                                                                 \
  DEC_STAMP_MIN(2);                                                 \
  const unsigned BID = blockIdx.x - (g).ct.nwg;

// C/8 lanes share one pixel (8 channels each); `warp` never leaves registers.
struct FlowArgs {
  const void* x;       // [F][R][R][C] conv2 output (unscaled), T::elem
  const void* feat;    // [R][R][C], T::elem
  const float* pflow;  // [F][R/2][R/2][4] (3 channels + pad: one 16-byte load per tap) or nullptr
  float upk_flow[8], upk_rgb[8];  // per-axis taps {ky[4], kx[4]} of ToFlow's / ToRGB's Upsample (K = ky (x) kx; (1,3,3,1)/4 each by default)
  const float* prgb;   // [F][R/2][R/2][4] or nullptr
  const float* wflow;  // [3][C], already * 1/sqrt(C)
  const float* sflow;  // [F][ld_s] style of the ToFlow conv (offset applied)
  const float* bflow;  // [3]
  const float* wrgb;   // [3][C], already * 1/sqrt(C)
  const float* oflow;  // dec_flowlast_kernel: [F][R][R][4] ToFlow's sums from conv2's epilogue (dec_conv16_kernel FLOWM)
  const float* grgb;   // [R][R][4]: ToRGB's 1x1 conv applied to the skip features themselves (dec_feat_rgb_kernel, once per clip)
  const float* b1;     // [3] FusedLeakyReLU bias
  const float* b2;     // [3] ToRGB bias
  const float* snext;  // [F][ld_s] or nullptr
  void* xnext;         // [F][R][R][C] T::elem, or nullptr (last level)
  float* flow_out;     // [F][R][R][4]
  float* rgb_out;      // [F][R][R][4]
  const float* lin;    // [R] identity grid: np.linspace(-1, 1, R) (float64) cast to float32
  float* final_out;    // last level: frames
  int final_mode;      // 0: none, 1: HWC clamp(-1,1)*0.5+0.5 (FLOAT.py:149-152), 2: raw CHW
  int write_pyr;       // store flow_out / rgb_out (0 on the last level: nobody reads them)
  int F, R, C, ld_s;
  int nbands, band_pix;  // the image is cut into nbands runs of band_pix consecutive pixels (multiple of gpb*PIX)
  CopyTail ct;           // copy that rides along (nwg == 0: none)
  unsigned long long* sat;  // saturation counter of this level's xnext stores
};

__device__ __forceinline__ void up2_tap3(const float* __restrict__ prev, int f, int Rp, int Y, int X, float out[3], const float (&k)[8]) {
  // Upsample of a 3-channel map (styledecoder.py:74-90): zero-insert x2, pad (2,1), correlate with the flipped 4 x 4 kernel K =
  // ky (x) kx (k = {ky[4], kx[4]}: make_kernel([1,3,3,1]) * 4 -> (1, 3, 3, 1) / 4 per axis unless the checkpoint's `upsample.kernel`
  // buffer says otherwise).  Output 2m takes p[m-1], p[m] with weights k[3], k[1]; output 2m+1 takes p[m], p[m+1] with k[2], k[0]
  // (default: .25, .75 / .75, .25 - the products below are then exact and equal to the literal-tap code this replaces).
  // The map is stored with 4 floats per pixel, so a tap is ONE 16-byte load for all three channels (the 4-byte-per-lane
  // version issued 32 scattered load instructions per 4 pixels and cost 5 of the decoder's 34 ms).
  const int my = Y >> 1, mx = X >> 1;
  const int y0 = (Y & 1) ? my : my - 1, x0 = (X & 1) ? mx : mx - 1;
  const float wy[2] = {(Y & 1) ? k[2] : k[3], (Y & 1) ? k[0] : k[1]};
  const float wx[2] = {(X & 1) ? k[6] : k[7], (X & 1) ? k[4] : k[5]};
  // every tap is loaded from a clamped (valid) address and an outside tap gets weight 0: no branch per tap (a divergent
  // branch around each load cost an exec save / restore and kept the loads from being issued back to back)
  float4 t[4];
  float tw[4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int yy = y0 + a, xx = x0 + b;
      const bool in = yy >= 0 && yy < Rp && xx >= 0 && xx < Rp;
      const int yc = min(max(yy, 0), Rp - 1), xc = min(max(xx, 0), Rp - 1);
      t[a * 2 + b] = *reinterpret_cast<const float4*>(prev + ((size_t)(f * Rp + yc) * Rp + xc) * 4);
      tw[a * 2 + b] = in ? 1.f : 0.f;
    }
  out[0] = out[1] = out[2] = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float wgt = wy[a] * wx[b] * tw[a * 2 + b];
      out[0] += wgt * t[a * 2 + b].x;
      out[1] += wgt * t[a * 2 + b].y;
      out[2] += wgt * t[a * 2 + b].z;
    }
}

// 1 - 2/(1 + e^{2x}); saturates cleanly at +-1.  EXACT: IEEE division (the fp32 verification mode); else v_rcp_f32 (1 ulp:
// 1e-7 on a value that positions a sample to 1/R of a pixel at best) - a division is ~10 instructions, the flow kernel had 12.
template <bool EXACT>
__device__ __forceinline__ float fh_tanh_fast(float x) {
  const float d = 1.f + __expf(2.f * x);
  return EXACT ? 1.f - 2.f / d : 1.f - 2.f * __builtin_amdgcn_rcpf(d);
}
template <bool EXACT>
__device__ __forceinline__ float fh_sigmoid_t(float x) {
  const float d = 1.f + __expf(-x);
  return EXACT ? 1.f / d : __builtin_amdgcn_rcpf(d);
}

// One run of PIX consecutive pixels of a row (first pixel p0 = Y * R + X0 of frame f) for one lane group: ToFlow (1x1
// modulated conv + up-sampled previous flow -> tanh / sigmoid), grid_sample of the skip features, blend, ToRGB, the pyramids,
// the next level's input, the final frame.  xu[k] = this lane's 8 channels (c0 ..) of conv2's output at pixel k.
// What dec_flow_pixels needs per frame, formed once per workgroup: the nine bias values (they were re-loaded from global memory
// for every run of pixels) and the frame's base pointers, so that every access inside the pixel loop is a uniform base plus a
// 32-bit lane offset (the 64-bit index arithmetic per access was 90 of the kernel's 1180 vector instructions per iteration).
template <class T>
struct FlowFrame {
  float bf[3], b1[3], b2[3];
  const float *pflow, *prgb;      // this frame's previous pyramids or nullptr
  float *flow_out, *rgb_out;      // this frame's pyramids
  float* final_hwc;               // this frame's output image (final_mode 1)
  float* final_chw;               // this frame's raw output (final_mode 2)
  typename T::elem* xnext;        // this frame's next-level input or nullptr
};
template <class T>
__device__ __forceinline__ FlowFrame<T> dec_flow_frame(const FlowArgs& g, int f) {
  FlowFrame<T> ff;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    ff.bf[j] = g.bflow[j];
    ff.b1[j] = g.b1[j];
    ff.b2[j] = g.b2[j];
  }
  const size_t npix = (size_t)g.R * g.R, npp = (size_t)(g.R >> 1) * (g.R >> 1);
  ff.pflow = g.pflow ? g.pflow + (size_t)f * npp * 4 : nullptr;
  ff.prgb = g.prgb ? g.prgb + (size_t)f * npp * 4 : nullptr;
  ff.flow_out = g.flow_out ? g.flow_out + (size_t)f * npix * 4 : nullptr;
  ff.rgb_out = g.rgb_out ? g.rgb_out + (size_t)f * npix * 4 : nullptr;
  ff.final_hwc = g.final_out ? g.final_out + (size_t)f * npix * 3 : nullptr;
  ff.final_chw = ff.final_hwc;
  ff.xnext = g.xnext ? reinterpret_cast<typename T::elem*>(g.xnext) + (size_t)f * npix * g.C : nullptr;
  return ff;
}

// ToRGB without a channel loop (round 6).  ToRGB's conv is an UNMODULATED 1x1 conv of the warped features, and the warp is
// linear in the features: conv(mask * sum_t bil_t feat[tap_t]) = sum_t (mask bil_t) conv(feat)[tap_t].  conv(feat) =: G does not
// depend on the frame - it is formed once per clip (dec_feat_rgb_kernel, [R][R][4] fp32) and the pixel's owner lane blends four
// 16-byte taps of it: 12 FMAs per pixel where every lane of the pixel spent 24 FMAs + a cross-lane reduction, in fp32 from
// fp32 sums (the old order rounded nothing either; the two differ by summation order only).  LAST (no next level: xnext ==
// nullptr) needs the warped features for nothing else, so the last level gathers no features at all.
template <class T, int PIX, bool LAST>
__device__ __forceinline__ void dec_flow_pixels(const FlowArgs& g, const FlowFrame<T>& ff, const float* __restrict__ sw, int p0, int sub,
                                                int lpp, const typename T::pack8 (&xu)[PIX], unsigned& sm) {
  typedef typename T::pack8 P8;
  constexpr unsigned EB = T::EB;
  const int C = g.C, c0 = sub * 8;
  const unsigned cb0 = (unsigned)c0 * EB;  // byte offset of this lane's 8 channels inside a pixel
  const bool owner = sub < PIX;
  const float bf0 = ff.bf[0], bf1 = ff.bf[1], bf2 = ff.bf[2];
  const float b10 = ff.b1[0], b11 = ff.b1[1], b12 = ff.b1[2], b20 = ff.b2[0], b21 = ff.b2[1], b22 = ff.b2[2];
  const int R = g.R, npix = R * R, Rp = R >> 1;
  const float fR = (float)R;
  {
    const int Y = p0 / R, X0 = p0 - Y * R;  // R % PIX == 0: the PIX pixels share a row
    float upf[3] = {0.f, 0.f, 0.f};  // up-sampled previous flow of THIS lane's pixel
    if (ff.pflow && owner) up2_tap3(ff.pflow, 0, Rp, Y, X0 + sub, upf, g.upk_flow);
    const float gy = g.lin[Y];
    float o[PIX][3];
#pragma unroll
    for (int k = 0; k < PIX; ++k) o[k][0] = o[k][1] = o[k][2] = 0.f;
#ifndef DEC_DIAG_NOFLOWDOT  // timing-only build (WRONG frames): no ToFlow channel loop, no reduction - the ceiling of moving ToFlow into
    // conv2's epilogue at the levels that keep this kernel
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float4 w0 = *reinterpret_cast<const float4*>(sw + j * C + c0);
      const float4 w1 = *reinterpret_cast<const float4*>(sw + j * C + c0 + 4);
      const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int k = 0; k < PIX; ++k) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) a += wv[i] * T::get(xu[k], i);
        o[k][j] = a;
      }
    }
#endif
#pragma unroll
    for (int k = 0; k < PIX; ++k) {
      // bias + up-sampled previous flow enter the sum once, in the lane that owns the pixel
      const bool mine = sub == k;
      o[k][0] += mine ? upf[0] + bf0 : 0.f;
      o[k][1] += mine ? upf[1] + bf1 : 0.f;
      o[k][2] += mine ? upf[2] + bf2 : 0.f;
    }
#ifndef DEC_DIAG_NOFLOWDOT
    for (int d = 1; d < lpp; d <<= 1) {
#pragma unroll
      for (int k = 0; k < PIX; ++k) {
        o[k][0] += __shfl_xor(o[k][0], d, 64);
        o[k][1] += __shfl_xor(o[k][1], d, 64);
        o[k][2] += __shfl_xor(o[k][2], d, 64);
      }
    }
#endif
    // Sample position, tap addresses and blend weights of a pixel are the same in all its lanes: lane `sub` forms them for
    // pixel sub & (PIX - 1) ONLY and the group reads them by lane index (ds_bpermute: LDS pipe) - 9 values per pixel instead of
    // ~100 vector instructions per pixel in every lane (r03 listing: a third of the kernel's 1 200 instructions per iteration,
    // and the kernel is VALU-bound).
    float mask[PIX], wg[PIX][4];
    unsigned fo[PIX][4];
    P8 fu[LAST ? 1 : PIX][4];
    float4 gt[4];  // owner lane: the four taps of G = ToRGB(features) of its pixel
    float gw[4];
    {
      const int ks = sub & (PIX - 1);
      float f0 = o[0][0], f1 = o[0][1], f2 = o[0][2];
#pragma unroll
      for (int k = 1; k < PIX; ++k)
        if (ks == k) {
          f0 = o[k][0];
          f1 = o[k][1];
          f2 = o[k][2];
        }
      const float sx = fh_tanh_fast<T::is32>(f0) + g.lin[X0 + ks], sy = fh_tanh_fast<T::is32>(f1) + gy;
      const float mk = fh_sigmoid_t<T::is32>(f2);
      // grid_sample, align_corners=False: pixel = ((coord + 1) * size - 1) / 2
      const float ix = ((sx + 1.f) * fR - 1.f) * 0.5f, iy = ((sy + 1.f) * fR - 1.f) * 0.5f;
      const float fx0 = floorf(ix), fy0 = floorf(iy);
      const int x0 = (int)fx0, y0 = (int)fy0;  // |ix| <= R + 1: tanh bounds the sample position
      const float axk = ix - fx0, ayk = iy - fy0;
      float own_w[4];
      unsigned own_o[4], own_p[4];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          // zero padding of grid_sample: the tap is read from the clamped position and weighted 0 when it lies outside
          const int yy = y0 + a, xx = x0 + b;
          const bool in = yy >= 0 && yy < R && xx >= 0 && xx < R;
          const int yc = min(max(yy, 0), R - 1), xc = min(max(xx, 0), R - 1);
          own_p[a * 2 + b] = (unsigned)(yc * R + xc);
          own_o[a * 2 + b] = own_p[a * 2 + b] * (unsigned)C * EB;  // bytes
          own_w[a * 2 + b] = (a ? ayk : 1.f - ayk) * (b ? axk : 1.f - axk) * mk * (in ? 1.f : 0.f);
        }
      // lanes sub >= PIX hold a copy of pixel sub & (PIX - 1): a valid address, the value is dropped
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        gt[t] = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(g.grgb) + own_p[t] * 16u);
        gw[t] = own_w[t];
      }
      if constexpr (!LAST) {
        const int gb = (int)(threadIdx.x & 63) - sub;  // first lane of this pixel group
#pragma unroll
        for (int k = 0; k < PIX; ++k) {
          mask[k] = __shfl(mk, gb + k, 64);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            wg[k][t] = __shfl(own_w[t], gb + k, 64);
            fo[k][t] = (unsigned)__shfl((int)own_o[t], gb + k, 64);
          }
        }
#pragma unroll
        for (int k = 0; k < PIX; ++k)
#pragma unroll
          for (int t = 0; t < 4; ++t) fu[k][t] = dec_load8_b<T>(g.feat, fo[k][t] + cb0);
      }
    }
    float upr[3] = {0.f, 0.f, 0.f};
    if (ff.prgb && owner) up2_tap3(ff.prgb, 0, Rp, Y, X0 + sub, upr, g.upk_rgb);
    if constexpr (!LAST) {
#pragma unroll
      for (int k = 0; k < PIX; ++k) {
        float fw[8];  // the warped, masked features: sum over the 4 taps of weight * feature
        dec_scale8<T>(fw, fu[k][0], wg[k][0]);
#pragma unroll
        for (int t = 1; t < 4; ++t) dec_axpy8<T>(fw, fu[k][t], wg[k][t]);
        const float4 n0 = *reinterpret_cast<const float4*>(sw + 6 * C + c0);
        const float4 n1 = *reinterpret_cast<const float4*>(sw + 6 * C + c0 + 4);
        const float sn[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
        const float om = 1.f - mask[k];
        float ov[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ov[i] = (fw[i] + T::get(xu[k], i) * om) * sn[i];
        dec_store8_b<T>(ff.xnext, (unsigned)((p0 + k) * C) * EB + cb0, dec_pack8<T>(ov, sm));
      }
    }
    if (owner) {
      // ToRGB's conv of the warped features = the four taps of G blended with the (masked) bilinear weights
      float r0 = gw[0] * gt[0].x, r1 = gw[0] * gt[0].y, r2 = gw[0] * gt[0].z;
#pragma unroll
      for (int t = 1; t < 4; ++t) {
        r0 += gw[t] * gt[t].x;
        r1 += gw[t] * gt[t].y;
        r2 += gw[t] * gt[t].z;
      }
      // this lane's pixel: select its flow sums out of the unrolled arrays
      float f0 = o[0][0], f1 = o[0][1], f2 = o[0][2];
#pragma unroll
      for (int k = 1; k < PIX; ++k)
        if (sub == k) {
          f0 = o[k][0];
          f1 = o[k][1];
          f2 = o[k][2];
        }
      const float v0 = fh_lrelu_s2(r0 + b10) + b20 + upr[0], v1 = fh_lrelu_s2(r1 + b11) + b21 + upr[1],
                  v2 = fh_lrelu_s2(r2 + b12) + b22 + upr[2];
      const unsigned po = (unsigned)(p0 + sub);
      if (g.write_pyr) {  // the pyramids are only read by the next level
        *reinterpret_cast<float4*>(ff.flow_out + po * 4u) = float4{f0, f1, f2, 0.f};
        *reinterpret_cast<float4*>(ff.rgb_out + po * 4u) = float4{v0, v1, v2, 0.f};
      }
      if (g.final_mode == 1) {
        // one 12-byte store per pixel (global_store_dwordx3): consecutive lanes own consecutive pixels, a wave writes 768
        // contiguous bytes with one instruction instead of three strided ones
        typedef float f3v __attribute__((ext_vector_type(3)));
        f3v o3 = {fminf(fmaxf(v0, -1.f), 1.f) * 0.5f + 0.5f, fminf(fmaxf(v1, -1.f), 1.f) * 0.5f + 0.5f,
                  fminf(fmaxf(v2, -1.f), 1.f) * 0.5f + 0.5f};
        __builtin_memcpy(ff.final_hwc + po * 3u, &o3, 12);
      } else if (g.final_mode == 2) {
        float* fo = ff.final_chw + po;
        fo[0] = v0;
        fo[npix] = v1;
        fo[2 * (size_t)npix] = v2;
      }
    }
  }
}

struct ConvArgs {
  const void* X;   // [F][Hi][Wi][Cin] T::elem, already multiplied by the layer's (normalised) style
  const void* Wt;  // [ntaps][Cout][Cin] T::elem
  void* Y;         // [F][OH][OW][Cout] T::elem
  const float* demod;  // [F][ldd] (offset applied) or nullptr
  const float* bias;   // [Cout] or nullptr
  const float* snext;  // [F][lds] (offset applied) or nullptr: style of the consumer layer
  int F, Hi, Wi, Cin, Cout;
  int Ho, Wo;          // output grid computed by this launch (per frame)
  int OH, OW, sy, sx, py, px;  // output pixel (oy*sy+py, ox*sx+px) of the OH x OW image
  int ldd, lds;
  int ntaps, dymin, dxmin, hh, hw;  // halo tile = (th + dy range) x (tw + dx range)
  int lth, ltw, lnf;                // log2 of tile height / width / frames per tile
  int tiles_x, tiles_y;
  int act;                          // 1: + bias, leaky_relu(0.2) * sqrt(2)
  int tpw;                          // dec_conv16_kernel: consecutive tiles per workgroup
  signed char dy[9], dx[9];
  CopyTail ct;                      // dec_conv16_kernel / dec_zblur_kernel: copy that rides along (nwg == 0: none)
  // dec_conv16_kernel / dec_zblur_kernel, ncb > 0: 1-D grid of ngroups x ncb compute workgroups, decoded by dec_group_cb
  // (ncb == 0: the output-channel block is blockIdx.y)
  unsigned ncb, ngroups;
  unsigned long long* sat;          // saturation counter of this layer's output stores
  // dec_zblur_kernel: 1-D taps of the Blur (weight of z[X - 1 + b] in output X; {.25, .75, .75, .25} for blur_kernel [1,3,3,1]);
  // fir_sym: fir is exactly {.25, .75, .75, .25} - the vertical pass with literal taps, mirrored rows added first (the code
  // every released checkpoint runs); else four FMAs with the taps from here
  float fir[4];
  int fir_sym;
  // dec_conv16_kernel<.., FLOWM = 1> (the LAST level's conv2): ToFlow's 1x1 conv in the epilogue, on the matrix pipe
  const void* wfrag;  // [F][NT / 2][fp32: 1 | 16-bit: 2 (hi, lo * 2048)][64 lanes] T::pack8: dec_flowfrag_kernel
  float* oflow;       // [F][OH][OW][4] fp32: sum_c wflow[j][c] s[f][c] V[c] (no bias) per pixel
};

// (tile group, output-channel block) of compute workgroup `bid`.  A layer with more than 32 output channels runs ncb workgroups
// per tile group that all stage the same input halo; as grid.y they were a whole batch apart (at 256 px: 268 MB of input per
// 32-frame batch between the first and the second read - every re-read came from HBM).  Here they are ncb consecutive slots of
// ONE XCD (ids congruent mod 8 share an L2 under round-robin placement: a speed assumption only), so the halo is fetched
// once and re-read from that L2 while it is hot.  The groups past the last multiple of 8 use the plain order.
__device__ __forceinline__ void dec_group_cb(unsigned bid, unsigned ngroups, unsigned ncb, unsigned& grp, unsigned& cb) {
  const unsigned g8 = ngroups & ~7u, cut = g8 * ncb;
  if (bid < cut) {
    const unsigned slot = bid >> 3;
    cb = slot % ncb;
    grp = (slot / ncb) * 8 + (bid & 7);
  } else {
    const unsigned r = bid - cut;
    cb = r % ncb;
    grp = g8 + r / ncb;
  }
}

template <class T, int NT>
__global__ __launch_bounds__(256, T::is32 ? 1 : 2) void dec_conv_kernel(ConvArgs g) {
  typedef typename T::elem E;
  typedef typename T::pack8 P8;
  constexpr int RB = 32 * T::EB, CB = 8 * T::EB;  // bytes of a 32-channel LDS row / of one pack of 8 channels
  const E* const X = reinterpret_cast<const E*>(g.X);
  const E* const Wt = reinterpret_cast<const E*>(g.Wt);
  constexpr int BN = NT * 16;
  constexpr int KC = 32;
  constexpr int MAXA = 9;  // halo pixels * 4 chunks / 256 threads, worst case 16 x 6 x 6
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int th = 1 << g.lth, tw = 1 << g.ltw, nf = 1 << g.lnf;
  int tile = blockIdx.x;
  const int tx = tile % g.tiles_x;
  tile /= g.tiles_x;
  const int ty = tile % g.tiles_y;
  const int fb = tile / g.tiles_y;
  const int n0 = blockIdx.y * BN;
  const int npix = nf * g.hh * g.hw;
  unsigned char* sA = smem;                 // [npix][RB]
  unsigned char* sB = smem + npix * RB;     // [ntaps][BN][RB]

  // global element offsets of this thread's halo chunks (constant over the channel loop)
  int aoff[MAXA];
  const int fpix = g.hh * g.hw;
  const float inv_fpix = 1.0f / (float)fpix, inv_hw = 1.0f / (float)g.hw;
  const int iy0 = ty * th + g.dymin, ix0 = tx * tw + g.dxmin;
#pragma unroll
  for (int i = 0; i < MAXA; ++i) {
    const int e = tid + i * 256;
    const int p = e >> 2, ch = e & 3;
    aoff[i] = -1;
    if (p < npix) {
      // small-integer division by multiplication with the fp32 reciprocal (exact for p < 2^12)
      const int fl = (int)(((float)p + 0.5f) * inv_fpix);
      const int rem = p - fl * fpix;
      const int hy = (int)(((float)rem + 0.5f) * inv_hw), hx = rem - hy * g.hw;
      const int f = fb * nf + fl, iy = iy0 + hy, ix = ix0 + hx;
      if (f < g.F && iy >= 0 && iy < g.Hi && ix >= 0 && ix < g.Wi) aoff[i] = ((f * g.Hi + iy) * g.Wi + ix) * g.Cin + ch * 8;
      else aoff[i] = -2;  // in-tile, zero padding
    }
  }
  // LDS pixel index of this lane's row in each of the wave's 4 m-tiles (tap (dymin,dxmin))
  int pbase[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = (w * 4 + mt) * 16 + r16;
    const int x = m & (tw - 1), y = (m >> g.ltw) & (th - 1), fl = m >> (g.ltw + g.lth);
    pbase[mt] = fl * g.hh * g.hw + y * g.hw + x;
  }

  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nb = g.ntaps * BN * 4;  // 16-byte chunks of the weight slab
  for (int c0 = 0; c0 < g.Cin; c0 += KC) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXA; ++i) {
      if (aoff[i] != -1) {
        P8 v = T::zero8();
        if (aoff[i] >= 0) v = T::load8(X + (size_t)aoff[i] + c0);
        *reinterpret_cast<P8*>(sA + (size_t)(tid + i * 256) * CB) = v;
      }
    }
    for (int e = tid; e < nb; e += 256) {
      const int row = e >> 2, ch = e & 3;   // row = tap * BN + n
      const int tap = row / BN, n = row - tap * BN;
      const P8 v = T::load8(Wt + ((size_t)tap * g.Cout + n0 + n) * g.Cin + c0 + ch * 8);
      *reinterpret_cast<P8*>(sB + (size_t)e * CB) = v;
    }
    __syncthreads();
    for (int t = 0; t < g.ntaps; ++t) {
      const int shift = (g.dy[t] - g.dymin) * g.hw + (g.dx[t] - g.dxmin);
      P8 a[4], b[NT];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const P8*>(sA + (size_t)(pbase[mt] + shift) * RB + q * CB);
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const P8*>(sB + (size_t)((t * BN + j * 16 + r16) * RB + q * CB));
#pragma unroll
      // operands swapped on purpose: D = W_tile (16 channels x K) * X_tile^T (K x 16 pixels), so a lane
      // ends up with 4 CONSECUTIVE CHANNELS of one pixel (row = q*4 + reg -> channel, col = r16 -> pixel)
      // and the epilogue stores 8 contiguous bytes per tile instead of four scattered 2-byte values
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[mt][j] = T::mfma(b[j], a[mt], acc[mt][j]);
    }
  }

  // epilogue: lane -> pixel m = tile row r16, channels n0 + j*16 + q*4 .. +3
  unsigned sm = 0u;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = (w * 4 + mt) * 16 + r16;
    const int x = m & (tw - 1), y = (m >> g.ltw) & (th - 1), fl = m >> (g.ltw + g.lth);
    const int f = fb * nf + fl, oy = ty * th + y, ox = tx * tw + x;
    if (f >= g.F || oy >= g.Ho || ox >= g.Wo) continue;
    E* yp = reinterpret_cast<E*>(g.Y) + ((size_t)(f * g.OH + oy * g.sy + g.py) * g.OW + ox * g.sx + g.px) * g.Cout;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = n0 + j * 16 + q * 4;
      float v[4] = {acc[mt][j][0], acc[mt][j][1], acc[mt][j][2], acc[mt][j][3]};
      if (g.demod) {
        const float4 d = *reinterpret_cast<const float4*>(g.demod + (size_t)f * g.ldd + co);
        v[0] *= d.x;
        v[1] *= d.y;
        v[2] *= d.z;
        v[3] *= d.w;
      }
      if (g.act) {
        const float4 bb = *reinterpret_cast<const float4*>(g.bias + co);
        v[0] = fh_lrelu_s2(v[0] + bb.x);
        v[1] = fh_lrelu_s2(v[1] + bb.y);
        v[2] = fh_lrelu_s2(v[2] + bb.z);
        v[3] = fh_lrelu_s2(v[3] + bb.w);
      }
      if (g.snext) {
        const float4 sn = *reinterpret_cast<const float4*>(g.snext + (size_t)f * g.lds + co);
        v[0] *= sn.x;
        v[1] *= sn.y;
        v[2] *= sn.z;
        v[3] *= sn.w;
      }
      dec_store4<T>(yp + co, v[0], v[1], v[2], v[3], sm);
    }
  }
  dec_sat_flush<T>(g.sat, sm);
}

// Specialisation of the conv for 16x16-pixel tiles of one frame and a dense TY x TX tap window (the 3x3 convs of every level
// from 16x16 upward; image sizes are multiples of 16 here).  What the counters and the listing asked for:
//   * r01 PMC: 42 % of LDS cycles bank conflicts -> LDS rows are 64 B (32 channels) and the 16-byte chunk index is XOR-ed with
//     (halo column >> 1) & 3, conflict-free for the ds_read_b128 fragment reads at any tile alignment (brute-forced over the
//     gfx950 lane groups, tools/probes/lds_swizzle.py);
//   * a wave's 4 m-tiles are image rows w*4 .. w*4+3 of the tile (x = r16), so tap (ty, tx) of m-tile mt reads halo row
//     w*4 + mt + ty at column shift tx: only (3 + TY) * TX distinct A fragments per chunk (3x3: 18 ds_read_b128, not 36);
//   * a workgroup walks `tpw` consecutive tiles and all K chunks as one item stream, and the global loads of item i+1 are
//     issued into registers before the MFMAs of item i (single LDS buffer);
//   * r03 listing: 700-800 vector instructions per item against 72 MFMAs (staging addresses with a division and 64-bit
//     multiplies per 16-byte load, swizzled LDS addresses re-derived per item, 30 instructions per epilogue fragment): the
//     wave spent two to three times the MFMA time issuing them.  Now every per-lane address is formed ONCE per workgroup - a
//     32-bit byte offset from a per-item scalar base for the loads, a register + immediate for the LDS accesses (the swizzle
//     depends on the halo column only, so fragment rows differ by a constant), one offset for the stores - tiles are decoded
//     by carrying (f, ty, tx) along instead of dividing, interior tiles load without any bounds logic, border tiles take one
//     4-bit mask per chunk (top / bottom / left / right halo membership, formed once) against the tile's border bits.
// DB = 1 (round 4, review item 2a): the halo tile, the weight slab and the epilogue operands are DOUBLE-BUFFERED in LDS: item
// i + 1 is written into the other buffer while item i is multiplied (the writes are spread between the three tap-column
// groups of MFMAs), and the K loop has ONE barrier per item instead of two.  115 KB of LDS at 64 output channels: one
// workgroup per CU instead of two.  Bitwise the same results.  MEASURED: see DESIGN.md section 7 (FLOAT_DEC_CONV_DB selects it).
// FLOWM = 1 (round 6; the last level's conv2, whose output V feeds ToFlow and nothing else): V is never stored.  The epilogue's
// values - rounded to the operand type exactly as the stored V was - are the B operand of one more MFMA per m-tile against
// ToFlow's per-frame folded weights (A fragments from dec_flowfrag_kernel: row r holds output r & 3, so every lane group gets
// the three sums of its pixel; 16-bit operands: weights as hi + lo * 2^-11, two MFMAs, exact products, i.e. fp32 weights as
// before), and the lane that owns pixel (row w*4 + q, column r16) stores 16 bytes.  At 512 px that is 4 MB per frame written
// and read back instead of 16.8 + 16.8, and the channel loop + cross-lane reduction of the flow kernel are gone.
template <class T, int NT, int TY, int TX, int DB = 0, int FLOWM = 0>
__global__ __launch_bounds__(256, (T::is32 || DB) ? 1 : 2) void dec_conv16_kernel(ConvArgs g) {
  DEC_COPY_PROLOGUE(g, bid)
  DEC_PH_BEGIN
  typedef typename T::elem E;
  typedef typename T::pack8 P8;
  typedef float v2f __attribute__((ext_vector_type(2)));
  constexpr int EB = T::EB, RB = 32 * EB, CB = 8 * EB;
  constexpr int BN = NT * 16, HH = 15 + TY, HW = 15 + TX, NPIX = HH * HW, NTAPS = TY * TX;
  constexpr int NA = (NPIX * 4 + 255) / 256, NBROWS = NTAPS * BN, NB = (NBROWS * 4 + 255) / 256;
  static_assert(64 % BN == 0, "weight rows of one staging round are whole taps apart");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sA = smem;              // [NPIX][RB], chunk ^ ((halo column >> 1) & 3)
  unsigned char* const sB = smem + NPIX * RB;  // [NTAPS][BN][RB], chunk ^ ((row >> 1) & 3)
  float* const sE = reinterpret_cast<float*>(sB + NBROWS * RB);  // [3][BN]: demod, bias, next style of the tile's (frame, channels)
  constexpr int BUF = NPIX * RB + NBROWS * RB + 3 * BN * (int)sizeof(float);  // DB: the second buffer set follows the first
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);  // the wave index is uniform: scalar arithmetic
  const int r16 = lane & 15, q = lane >> 4;
  const int Wi = g.Wi, Cin = g.Cin;

  // ---- per-lane constants of the staging pass.  Chunk e = tid + 256 i of the halo tile: pixel p = e >> 2, pack ch = e & 3;
  // lanes past the last pixel repeat the last pixel's chunk (same bytes to the same place: no mask anywhere).
  unsigned a_off[NA];  // byte offset of the chunk from the halo origin of the tile in X
  unsigned a_lds[NA];  // its LDS byte offset
  unsigned a_edge = 0u;  // 4 bits per chunk: halo row above / below, halo column left / right of the 16x16 tile
  const int ch = tid & 3;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int p = min((tid >> 2) + 64 * i, NPIX - 1);
    const int hy = p / HW, hx = p - hy * HW;
    a_off[i] = (unsigned)(((hy * Wi + hx) * Cin + ch * 8) * EB);
    a_lds[i] = (unsigned)(p * RB + (ch ^ ((hx >> 1) & 3)) * CB);
    const unsigned m = (hy < -g.dymin ? 1u : 0u) | (hy >= 16 - g.dymin ? 2u : 0u) | (hx < -g.dxmin ? 4u : 0u) | (hx >= 16 - g.dxmin ? 8u : 0u);
    a_edge |= m << (4 * i);
  }
  const unsigned a_safe = (unsigned)(((-g.dymin * Wi - g.dxmin) * Cin + ch * 8) * EB);  // pixel (0, 0) of the tile: always inside
  // weight rows: row = (tid >> 2) + 64 i = tap * BN + n; 64 / BN taps per round, the last round repeats the last row
  const int brow = tid >> 2;
  const unsigned b_off = (unsigned)((((brow / BN) * g.Cout + brow % BN) * Cin + ch * 8) * EB);
  const int brow_l = min(brow + 64 * (NB - 1), NBROWS - 1);
  const unsigned b_off_l = (unsigned)((((brow_l / BN) * g.Cout + brow_l % BN) * Cin + ch * 8) * EB);
  const unsigned b_lds = (unsigned)(brow * RB + (ch ^ ((brow >> 1) & 3)) * CB);
  const unsigned b_lds_l = (unsigned)(brow_l * RB + (ch ^ ((brow_l >> 1) & 3)) * CB);
  const size_t b_round = (size_t)(64 / BN) * g.Cout * Cin * EB;  // bytes between staging rounds

  // ---- per-lane constants of the MFMA pass: A fragment (hr, tx) = halo row w*4 + hr, columns r16 + tx; B fragment rows t*BN + j*16 + r16
  unsigned fa[TX];
#pragma unroll
  for (int tx = 0; tx < TX; ++tx) fa[tx] = (unsigned)(((w * 4) * HW + r16 + tx) * RB + (q ^ (((r16 + tx) >> 1) & 3)) * CB);
  const unsigned fb = (unsigned)(r16 * RB + (q ^ ((r16 >> 1) & 3)) * CB);
  // ---- epilogue: output pixel (ty*16 + w*4 + mt, tx*16 + r16), channels n0 + j*16 + q*4 .. +3
  const unsigned y_off = (unsigned)((((w * 4) * g.sy * g.OW + r16 * g.sx) * g.Cout + q * 4) * EB);
  const size_t y_row = (size_t)g.sy * g.OW * g.Cout * EB;  // bytes between the wave's m-tiles

  const int tiles_pf = g.tiles_x * g.tiles_y;
  const int total = tiles_pf * g.F;
  const int nchunks = Cin >> 5;
  unsigned grp = bid, cb = blockIdx.y;
  if (g.ncb) dec_group_cb(bid, g.ngroups, g.ncb, grp, cb);
  const int tile0 = grp * g.tpw;
  const int ntile = min(g.tpw, total - tile0);
  const int nitems = ntile * nchunks;
  const int n0 = cb * BN;
  const unsigned char* const Wn = reinterpret_cast<const unsigned char*>(g.Wt) + (size_t)n0 * Cin * EB;

  // the tile being staged (runs one item ahead of the tile being computed) and the tile being computed: (frame, ty, tx)
  int sf, sty, stx, schunk = 0;
  {
    sf = tile0 / tiles_pf;
    const int rem = tile0 - sf * tiles_pf;
    sty = rem / g.tiles_x;
    stx = rem - sty * g.tiles_x;
  }
  int cf = sf, cty = sty, ctx = stx;
  unsigned s_edge = 0u;  // border bits of the staged tile (uniform): top, bottom, left, right
  unsigned s_hit = 0u;   // per chunk: 4-bit field != 0 <=> the chunk lies outside the image (zero padding)

  P8 ra[NA], rb[NB];
  float4 re = float4{0.f, 0.f, 0.f, 0.f};
  bool first = true;
  constexpr int NPK = NT / 2, NHL = T::is32 ? 1 : 2;  // FLOWM: weight fragments per frame (packs of 32 channels x (hi, lo))
  static_assert(!FLOWM || (!DB && (NT == 2 || NT == 4)), "ToFlow epilogue: single-buffered kernel, 32 or 64 output channels");
  P8 wa[FLOWM ? NPK : 1][NHL];
  auto issue = [&]() {
    const unsigned edge = (sty == 0 ? 1u : 0u) | (sty == g.tiles_y - 1 ? 2u : 0u) | (stx == 0 ? 4u : 0u) | (stx == g.tiles_x - 1 ? 8u : 0u);
    s_edge = edge;
    const long long org = ((long long)(sf * g.Hi + sty * 16 + g.dymin) * Wi + stx * 16 + g.dxmin) * Cin + (schunk << 5);
    const unsigned char* const Xo = reinterpret_cast<const unsigned char*>(g.X) + org * EB;
#ifdef DEC_DIAG_NOLOAD  // timing-only build (WRONG frames): halo loads for the workgroup's first item only (the registers keep what they had)
    if (!first) {
    } else
#endif
    if (edge == 0u) {
#pragma unroll
      for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const P8*>(Xo + a_off[i]);
    } else {
      s_hit = a_edge & (edge * 0x11111111u);
#pragma unroll
      for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const P8*>(Xo + (((s_hit >> (4 * i)) & 15u) ? a_safe : a_off[i]));
    }
#ifdef DEC_DIAG_WSTAGE1  // timing-only build (WRONG frames): the weight slab is fetched and staged for the workgroup's first item only -
    // the ceiling of any weight-stationary form of this kernel
    if (first) {
#else
    if (nchunks > 1 || schunk == 0) {
#endif
      const unsigned char* const Wc = Wn + (size_t)(schunk << 5) * EB;
#pragma unroll
      for (int i = 0; i < NB - 1; ++i) rb[i] = *reinterpret_cast<const P8*>(Wc + i * b_round + b_off);
      rb[NB - 1] = *reinterpret_cast<const P8*>(Wc + b_off_l);
    }
    // the tile's per-(frame, channel) epilogue operands travel with its last chunk: thread t < 3 BN / 4 stages 4 floats of
    // (demod | bias | next style); read in the epilogue they cost ~1 us of exposed latency per tile (r03 in-kernel stamps)
    if (schunk == nchunks - 1 && tid < 3 * BN / 4) {
      const int which = tid / (BN / 4), co = n0 + (tid % (BN / 4)) * 4;
      re = float4{which == 1 ? 0.f : 1.f, which == 1 ? 0.f : 1.f, which == 1 ? 0.f : 1.f, which == 1 ? 0.f : 1.f};
      if (which == 0 && g.demod) re = *reinterpret_cast<const float4*>(g.demod + (size_t)sf * g.ldd + co);
      if (which == 1 && g.act) re = *reinterpret_cast<const float4*>(g.bias + co);
      if (which == 2 && g.snext) re = *reinterpret_cast<const float4*>(g.snext + (size_t)sf * g.lds + co);
    }
    // advance the staging cursor
    if (++schunk == nchunks) {
      schunk = 0;
      if (++stx == g.tiles_x) {
        stx = 0;
        if (++sty == g.tiles_y) {
          sty = 0;
          ++sf;
        }
      }
    }
  };

  f32x4 acc[4][NT];
  int chunk = 0;
  issue();
  DEC_PH(8);
  if constexpr (DB) {
    // write the staged registers (item `it`) into buffer set `bs`, in three parts: part 0 = halo tile, 1 = weights, 2 = epilogue operands
    const bool wdb = nchunks > 1;  // one chunk per tile: the weight slab is staged once and shared by both buffer sets
    auto commit = [&](int bs, int part, bool wfirst, bool lastchunk) {
      unsigned char* const dA = sA + bs * BUF;
      unsigned char* const dB = sB + (wdb ? bs * BUF : 0);
      if (part == 0) {
        if (s_edge == 0u) {
#pragma unroll
          for (int i = 0; i < NA; ++i) *reinterpret_cast<P8*>(dA + a_lds[i]) = ra[i];
        } else {
#pragma unroll
          for (int i = 0; i < NA; ++i) *reinterpret_cast<P8*>(dA + a_lds[i]) = ((s_hit >> (4 * i)) & 15u) ? T::zero8() : ra[i];
        }
      } else if (part == 1) {
        if (wdb || wfirst) {
#pragma unroll
          for (int i = 0; i < NB - 1; ++i) *reinterpret_cast<P8*>(dB + b_lds + i * 64 * RB) = rb[i];
          *reinterpret_cast<P8*>(dB + b_lds_l) = rb[NB - 1];
        }
      } else {
        if (lastchunk && tid < 3 * BN / 4) *reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(sE) + bs * BUF + tid * 16) = re;
      }
    };
    // chunk index of the item being STAGED (issue() advanced schunk already): the item whose registers are pending
    int pchunk = 0;  // chunk of the pending (loaded, not yet committed) item
    commit(0, 0, true, nchunks == 1);
    commit(0, 1, true, nchunks == 1);
    commit(0, 2, true, nchunks == 1);
    if (nitems > 1) {
      pchunk = nchunks > 1 ? 1 : 0;
      issue();
    }
    __syncthreads();
    for (int item = 0; item < nitems; ++item) {
      const int cur = item & 1;
      const bool more = item + 1 < nitems;
      const bool plast = pchunk == nchunks - 1;
      const unsigned char* const cA = sA + cur * BUF;
      const unsigned char* const cB = sB + (wdb ? cur * BUF : 0);
      if (chunk == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int tx = 0; tx < TX; ++tx) {
        P8 b[TY][NT];
#pragma unroll
        for (int ty = 0; ty < TY; ++ty)
#pragma unroll
          for (int j = 0; j < NT; ++j) b[ty][j] = *reinterpret_cast<const P8*>(cB + fb + ((ty * TX + tx) * BN + j * 16) * RB);
        P8 a[3 + TY];
#pragma unroll
        for (int hr = 0; hr < 3 + TY; ++hr) a[hr] = *reinterpret_cast<const P8*>(cA + fa[tx] + hr * HW * RB);
        // the next item's LDS image, one part behind each tap column's fragment reads: it is written while the MFMAs below run
        if (more && tx < 3) commit(cur ^ 1, tx, false, plast);
#pragma unroll
        for (int hr = 0; hr < 3 + TY; ++hr) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const int ty = hr - mt;
            if (ty >= 0 && ty < TY) {
#pragma unroll
              for (int j = 0; j < NT; ++j) acc[mt][j] = T::mfma(b[ty][j], a[hr], acc[mt][j]);
            }
          }
        }
      }
      if (more) {
        if (TX < 3) {
#pragma unroll
          for (int part = TX; part < 3; ++part) commit(cur ^ 1, part, false, plast);
        }
        if (item + 2 < nitems) {
          pchunk = (pchunk + 1 == nchunks) ? 0 : pchunk + 1;
          issue();  // the registers are free again: item + 2 is in flight during the next item
        }
      }
      if (++chunk == nchunks) {
        chunk = 0;
        const float* const cE = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(sE) + cur * BUF);
        // v = lrelu(acc * d + b) * (sqrt2 * s): leaky_relu(0.2) as max(v, 0.2 v) = med3(v, slope v, +inf) (one instruction; fmaxf
        // costs a canonicalising v_max first), slope = 1 when the layer has no activation; the sqrt(2) rides in the style
        v2f ed[NT][2], eb[NT][2], es[NT][2];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float4 d = *reinterpret_cast<const float4*>(cE + j * 16 + q * 4);
          const float4 bb = *reinterpret_cast<const float4*>(cE + BN + j * 16 + q * 4);
          float4 sn = *reinterpret_cast<const float4*>(cE + 2 * BN + j * 16 + q * 4);
          if (g.act) {
            sn.x *= 1.4142135623730951f;
            sn.y *= 1.4142135623730951f;
            sn.z *= 1.4142135623730951f;
            sn.w *= 1.4142135623730951f;
          }
          ed[j][0] = v2f{d.x, d.y};
          ed[j][1] = v2f{d.z, d.w};
          eb[j][0] = v2f{bb.x, bb.y};
          eb[j][1] = v2f{bb.z, bb.w};
          es[j][0] = v2f{sn.x, sn.y};
          es[j][1] = v2f{sn.z, sn.w};
        }
        const float slope = g.act ? 0.2f : 1.0f;
        unsigned char* yt = reinterpret_cast<unsigned char*>(g.Y) +
                            ((((size_t)cf * g.OH + (size_t)cty * 16 * g.sy + g.py) * g.OW + (size_t)ctx * 16 * g.sx + g.px) * g.Cout + n0) * EB;
        unsigned sm = 0u;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            v2f v0 = v2f{acc[mt][j][0], acc[mt][j][1]} * ed[j][0] + eb[j][0];
            v2f v1 = v2f{acc[mt][j][2], acc[mt][j][3]} * ed[j][1] + eb[j][1];
            const v2f l0 = slope * v0, l1 = slope * v1;
            v0 = v2f{__builtin_amdgcn_fmed3f(v0.x, l0.x, __builtin_inff()), __builtin_amdgcn_fmed3f(v0.y, l0.y, __builtin_inff())} * es[j][0];
            v1 = v2f{__builtin_amdgcn_fmed3f(v1.x, l1.x, __builtin_inff()), __builtin_amdgcn_fmed3f(v1.y, l1.y, __builtin_inff())} * es[j][1];
            dec_store4<T>(reinterpret_cast<E*>(yt + mt * y_row + y_off + j * 16 * EB), v0.x, v0.y, v1.x, v1.y, sm);
          }
        }
        dec_sat_flush<T>(g.sat, sm);
        // the next tile to compute
        if (++ctx == g.tiles_x) {
          ctx = 0;
          if (++cty == g.tiles_y) {
            cty = 0;
            ++cf;
          }
        }
      }
      __syncthreads();  // the one barrier of the item: buffer `cur` is free, buffer `cur ^ 1` is complete
    }
  } else {
    for (int item = 0; item < nitems; ++item) {
      __syncthreads();  // every wave is done reading the previous item's tiles
      if (s_edge == 0u) {
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<P8*>(sA + a_lds[i]) = ra[i];
      } else {
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<P8*>(sA + a_lds[i]) = ((s_hit >> (4 * i)) & 15u) ? T::zero8() : ra[i];
      }
#ifdef DEC_DIAG_WSTAGE1
      if (first) {
#else
      if (nchunks > 1 || first) {
#endif
#pragma unroll
        for (int i = 0; i < NB - 1; ++i) *reinterpret_cast<P8*>(sB + b_lds + i * 64 * RB) = rb[i];
        *reinterpret_cast<P8*>(sB + b_lds_l) = rb[NB - 1];
        first = false;
      }
      if (chunk == nchunks - 1 && tid < 3 * BN / 4) *reinterpret_cast<float4*>(sE + tid * 4) = re;
      __syncthreads();
      DEC_PH(9);
      if (item + 1 < nitems) issue();  // in flight while this item computes
      if constexpr (FLOWM) {
        if (chunk == nchunks - 1) {  // ToFlow's weight fragments of the tile's frame: in registers when the epilogue starts
          const P8* const wf = reinterpret_cast<const P8*>(g.wfrag) + (size_t)cf * (NPK * NHL * 64) + lane;
#pragma unroll
          for (int pk = 0; pk < NPK; ++pk)
#pragma unroll
            for (int hl = 0; hl < NHL; ++hl) wa[pk][hl] = wf[(pk * NHL + hl) * 64];
        }
      }
      if (chunk == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int tx = 0; tx < TX; ++tx) {
        P8 b[TY][NT];
#pragma unroll
        for (int ty = 0; ty < TY; ++ty)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
#ifdef DEC_DIAG_READS36  // timing-only build (WRONG frames): two thirds of the fragment reads, every MFMA - the bound on a 32-pixel-wide M tile
            if (ty == TY - 1 && TY == 3) {
              b[ty][j] = b[0][j];
              continue;
            }
#endif
            b[ty][j] = *reinterpret_cast<const P8*>(sB + fb + ((ty * TX + tx) * BN + j * 16) * RB);
          }
#ifdef DEC_DIAG_READS36
        P8 a_seen[3 + TY];
#endif
#pragma unroll
        for (int hr = 0; hr < 3 + TY; ++hr) {
#ifdef DEC_DIAG_READS36
          const P8 a = (hr >= 4 && TY == 3) ? a_seen[hr - 4] : *reinterpret_cast<const P8*>(sA + fa[tx] + hr * HW * RB);
          a_seen[hr] = a;
#else
          const P8 a = *reinterpret_cast<const P8*>(sA + fa[tx] + hr * HW * RB);
#endif
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const int ty = hr - mt;
            if (ty >= 0 && ty < TY) {
#pragma unroll
              for (int j = 0; j < NT; ++j) {
#ifdef DEC_DIAG_NOMFMA  // timing-only build (WRONG frames): one VALU operation that consumes both fragments instead of the MFMA
                if constexpr (!T::is32) acc[mt][j][0] += __builtin_bit_cast(float, b[ty][j][0] ^ a[mt & 3]);
#else
                acc[mt][j] = T::mfma(b[ty][j], a, acc[mt][j]);  // D[channel][pixel], see dec_conv_kernel
#endif
              }
            }
          }
        }
      }
      DEC_PH(10);
      if (++chunk == nchunks) {
        chunk = 0;
        DEC_PH_COUNT(15);
        // v = lrelu(acc * d + b) * (sqrt2 * s): leaky_relu(0.2) as max(v, 0.2 v) = med3(v, slope v, +inf) (one instruction; fmaxf
        // costs a canonicalising v_max first), slope = 1 when the layer has no activation; the sqrt(2) rides in the style
        v2f ed[NT][2], eb[NT][2], es[NT][2];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float4 d = *reinterpret_cast<const float4*>(sE + j * 16 + q * 4);
          const float4 bb = *reinterpret_cast<const float4*>(sE + BN + j * 16 + q * 4);
          float4 sn = *reinterpret_cast<const float4*>(sE + 2 * BN + j * 16 + q * 4);
          if (g.act) {
            sn.x *= 1.4142135623730951f;
            sn.y *= 1.4142135623730951f;
            sn.z *= 1.4142135623730951f;
            sn.w *= 1.4142135623730951f;
          }
          ed[j][0] = v2f{d.x, d.y};
          ed[j][1] = v2f{d.z, d.w};
          eb[j][0] = v2f{bb.x, bb.y};
          eb[j][1] = v2f{bb.z, bb.w};
          es[j][0] = v2f{sn.x, sn.y};
          es[j][1] = v2f{sn.z, sn.w};
        }
        const float slope = g.act ? 0.2f : 1.0f;
        unsigned char* yt = reinterpret_cast<unsigned char*>(g.Y) +
                            ((((size_t)cf * g.OH + (size_t)cty * 16 * g.sy + g.py) * g.OW + (size_t)ctx * 16 * g.sx + g.px) * g.Cout + n0) * EB;
        unsigned sm = 0u;
        f32x4 dh[FLOWM ? 4 : 1], dl[FLOWM ? 4 : 1];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          float vv[NT * 4];
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            v2f v0 = v2f{acc[mt][j][0], acc[mt][j][1]} * ed[j][0] + eb[j][0];
            v2f v1 = v2f{acc[mt][j][2], acc[mt][j][3]} * ed[j][1] + eb[j][1];
            const v2f l0 = slope * v0, l1 = slope * v1;
            v0 = v2f{__builtin_amdgcn_fmed3f(v0.x, l0.x, __builtin_inff()), __builtin_amdgcn_fmed3f(v0.y, l0.y, __builtin_inff())} * es[j][0];
            v1 = v2f{__builtin_amdgcn_fmed3f(v1.x, l1.x, __builtin_inff()), __builtin_amdgcn_fmed3f(v1.y, l1.y, __builtin_inff())} * es[j][1];
            if constexpr (FLOWM) {
              vv[j * 4] = v0.x;
              vv[j * 4 + 1] = v0.y;
              vv[j * 4 + 2] = v1.x;
              vv[j * 4 + 3] = v1.y;
            } else {
              dec_store4<T>(reinterpret_cast<E*>(yt + mt * y_row + y_off + j * 16 * EB), v0.x, v0.y, v1.x, v1.y, sm);
            }
          }
          if constexpr (FLOWM) {
            // K slot s of pack pk in lane group q <-> channel (2 pk + (s >> 2)) * 16 + q * 4 + (s & 3): the order dec_flowfrag_kernel packs
            dh[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dl[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int pk = 0; pk < NPK; ++pk) {
              const float v8[8] = {vv[pk * 8], vv[pk * 8 + 1], vv[pk * 8 + 2], vv[pk * 8 + 3], vv[pk * 8 + 4], vv[pk * 8 + 5], vv[pk * 8 + 6], vv[pk * 8 + 7]};
              const P8 bp = dec_pack8<T>(v8, sm);  // the rounding (and range check) the stored V had
              dh[mt] = T::mfma(wa[pk][0], bp, dh[mt]);
              if constexpr (NHL == 2) dl[mt] = T::mfma(wa[pk][1], bp, dl[mt]);
            }
          }
        }
        if constexpr (FLOWM) {
          // D[row 4 q' + i][pixel r16] = output i of m-tile mt's pixel, the same in every lane group q': lane (r16, q) keeps m-tile q
          f32x4 oh = dh[0], ol = dl[0];
#pragma unroll
          for (int mt = 1; mt < 4; ++mt)
            if (q == mt) {
              oh = dh[mt];
              ol = dl[mt];
            }
          float4 o4;
          if constexpr (NHL == 2) o4 = float4{oh[0] + ol[0] * (1.f / 2048.f), oh[1] + ol[1] * (1.f / 2048.f), oh[2] + ol[2] * (1.f / 2048.f), 0.f};
          else o4 = float4{oh[0], oh[1], oh[2], 0.f};
          *reinterpret_cast<float4*>(g.oflow + ((((size_t)cf * g.OH + cty * 16 + w * 4 + q) * g.OW + ctx * 16 + r16) << 2)) = o4;
        }
        dec_sat_flush<T>(g.sat, sm);
        DEC_PH(11);
        // the next tile to compute
        if (++ctx == g.tiles_x) {
          ctx = 0;
          if (++cty == g.tiles_y) {
            cty = 0;
            ++cf;
          }
        }
      }
    }
  }
  DEC_PH_END(8);
}

// All four parity classes of the stride-2 transposed conv in ONE launch (styledecoder.py:250-257): output
// row u = 2m + pu takes kernel rows (ky=2, y=m-1),(ky=0, y=m) for pu = 0 and (ky=1, y=m) for pu = 1 (same in
// x), so the classes read the SAME 17x17 input halo of a 16x16 block of (m, n) positions.  One staging of
// the halo and of the 9 tap matrices feeds 4 + 2 + 2 + 1 tap products; the separate-launch version staged
// the halo four times for a quarter of the MFMA work each (r01: 165 TFLOP/s vs 530-810 for the 3x3 convs).
// Weights: [9][Cout][Cin] in class order (0,0),(0,1),(1,0),(1,1), taps by ascending (dy, dx).
template <class T>
__global__ __launch_bounds__(256, T::is32 ? 1 : 2) void dec_zconv4_kernel(ConvArgs g) {
  typedef typename T::elem E;
  typedef typename T::pack8 P8;
  constexpr int RB = 32 * T::EB, CB = 8 * T::EB;
  const E* const X = reinterpret_cast<const E*>(g.X);
  const E* const Wt = reinterpret_cast<const E*>(g.Wt);
  constexpr int NT = 2, BN = 32, HW = 17, NPIX = HW * HW, NTAPS = 9;
  constexpr int NA = (NPIX * 4 + 255) / 256, NBC = NTAPS * BN * 4, NB = (NBC + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;              // [17*17][RB], pack-swizzled
  unsigned char* sB = smem + NPIX * RB;  // [9][BN][RB], pack-swizzled
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  // swizzled LDS offsets of the four distinct input shifts (dy, dx) in {-1,0}^2, two per register
  unsigned aaddr[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int mrow = (w * 4 + mt), x = r16;  // position (m = tile_y*16 + mrow, n = tile_x*16 + x)
#pragma unroll
    for (int sh = 0; sh < 4; ++sh) {
      const int P = (mrow + (sh >> 1)) * HW + x + (sh & 1);  // halo origin is (m-1, n-1)
      const unsigned off = (unsigned)(P * RB + (q ^ ((P >> 1) & 3)) * CB);
      if (sh & 1) aaddr[mt][sh >> 1] |= off << 16;
      else aaddr[mt][sh >> 1] = off;
    }
  }
  const int baddr = r16 * RB + (q ^ ((r16 >> 1) & 3)) * CB;
  const int tiles_pf = g.tiles_x * g.tiles_y;
  const int tile = blockIdx.x;
  const int f = tile / tiles_pf, rem = tile - f * tiles_pf;
  const int ty = rem / g.tiles_x, tx = rem - ty * g.tiles_x;
  const int n0 = blockIdx.y * BN;
  const int nchunks = g.Cin >> 5;
  const int iy0 = ty * 16 - 1, ix0 = tx * 16 - 1;

  f32x4 acc[4][4][NT];  // [class][m-tile][n-tile]
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[c][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  P8 ra[NA], rb[NB];
  auto issue = [&](int chunk) {
    const int c0 = chunk << 5;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + i * 256, p = e >> 2, ch = e & 3;
      ra[i] = T::zero8();
      if (p < NPIX) {
        const int hy = p / HW, hx = p - hy * HW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        if (iy >= 0 && iy < g.Hi && ix >= 0 && ix < g.Wi)
          ra[i] = T::load8(X + ((size_t)(f * g.Hi + iy) * g.Wi + ix) * g.Cin + c0 + ch * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int e = tid + i * 256;
      if (e < NBC) {
        const int row = e >> 2, ch = e & 3;
        const int tap = row / BN, n = row - tap * BN;
        rb[i] = T::load8(Wt + ((size_t)tap * g.Cout + n0 + n) * g.Cin + c0 + ch * 8);
      }
    }
  };
  issue(0);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + i * 256, p = e >> 2, ch = e & 3;
      if (p < NPIX) *reinterpret_cast<P8*>(sA + p * RB + (ch ^ ((p >> 1) & 3)) * CB) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int e = tid + i * 256;
      if (e < NBC) {
        const int row = e >> 2, ch = e & 3;
        *reinterpret_cast<P8*>(sB + row * RB + (ch ^ ((row >> 1) & 3)) * CB) = rb[i];
      }
    }
    __syncthreads();
    if (chunk + 1 < nchunks) issue(chunk + 1);
    // input shift sh = 2*(dy+1) + (dx+1): 0 (-1,-1)  1 (-1,0)  2 (0,-1)  3 (0,0); the A fragments of one shift
    // serve every (class, tap) that reads it: class 0 taps 0..3 = sh 0..3; class 1 (pv=1) taps 4,5 = sh 1,3;
    // class 2 (pu=1) taps 6,7 = sh 2,3; class 3 tap 8 = sh 3
    constexpr int kNum[4] = {1, 2, 2, 4};
    constexpr int kTap[4][4] = {{0, 0, 0, 0}, {1, 4, 0, 0}, {2, 6, 0, 0}, {3, 5, 7, 8}};
    constexpr int kCls[4][4] = {{0, 0, 0, 0}, {0, 1, 0, 0}, {0, 2, 0, 0}, {0, 1, 2, 3}};
#pragma unroll
    for (int sh = 0; sh < 4; ++sh) {
      P8 a[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        a[mt] = *reinterpret_cast<const P8*>(sA + ((sh & 1) ? (aaddr[mt][sh >> 1] >> 16) : (aaddr[mt][sh >> 1] & 0xffffu)));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u < kNum[sh]) {
          const int t = kTap[sh][u], c = kCls[sh][u];
          P8 b[NT];
#pragma unroll
          for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const P8*>(sB + baddr + (t * BN + j * 16) * RB);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[c][mt][j] = T::mfma(b[j], a[mt], acc[c][mt][j]);
        }
      }
    }
  }
  // epilogue: z[f][2m+pu][2n+pv][co] = acc * demod, for the positions that exist (u, v <= R = OH-1)
  float4 ed[NT];
  unsigned sm = 0u;
#pragma unroll
  for (int j = 0; j < NT; ++j) ed[j] = *reinterpret_cast<const float4*>(g.demod + (size_t)f * g.ldd + n0 + j * 16 + q * 4);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int mpos = ty * 16 + w * 4 + mt, npos = tx * 16 + r16;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int u = 2 * mpos + (c >> 1), v = 2 * npos + (c & 1);
      if (u >= g.OH || v >= g.OW) continue;
      E* yp = reinterpret_cast<E*>(g.Y) + ((size_t)(f * g.OH + u) * g.OW + v) * g.Cout;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = n0 + j * 16 + q * 4;
        const float4 d = ed[j];
        dec_store4<T>(yp + co, acc[c][mt][j][0] * d.x, acc[c][mt][j][1] * d.y, acc[c][mt][j][2] * d.z, acc[c][mt][j][3] * d.w, sm);
      }
    }
  }
  dec_sat_flush<T>(g.sat, sm);
}

// Transposed conv AND its FIR blur in one launch, for the levels whose z tensor is HBM traffic (r01: at 512x512 the separate
// kernels ran at 3.4 TB/s - z is written once and read once, 2 x 16.8 MB per frame, for nothing).  A workgroup computes the
// z values of a 16x16 block of (m, n) positions like dec_zconv4_kernel - 32x32 z pixels x 32 channels - parks them in LDS
// (in the operand type, the rounding the z tensor had), and filters the 28x28 output pixels whose 4x4 support lies inside:
// block origin (14 ty - 1, 14 tx - 1), output tile origin (28 ty, 28 tx), out[Y][X] = sum_ab k[a] k[b] z[Y - 1 + a][X - 1 + b].
// Positions outside the image read zero inputs, so their z (the blur's zero padding) comes out as exact zeros by itself.
// (16/14)^2 = 1.31x the MFMA work of the unfused kernel buys 59 -> 25 MB of traffic per frame at 512x512.
//
// The filter runs on the matrix pipe (round 3; the listing of the VALU filter: 2 600 vector instructions per tile against 144
// MFMAs, 2 x 17 x 4 z packs converted to fp32 by every thread - the kernel was issue-bound at 211 TFLOP/s):
//   * the conv accumulates D[position n][channel] (pixel operand first), so a lane holds 4 consecutive n of one channel for
//     both column parities = 8 CONSECUTIVE z columns: one pack, written to the z tile as [z row][channel][32 z columns];
//   * the horizontal pass of z row zr is ONE MFMA per (16 channels, 16 output columns): A = the z row (a lane's pack = 8
//     consecutive columns of its channel, K = the 32 columns of the tile), B = the banded filter matrix Bx[zc][X] =
//     k[zc - X - 1] (constants in 2 x 4 registers), fp32 accumulation of exact products - the sums the VALU filter formed;
//   * its result D[channel][X] leaves 4 consecutive channels of one output column in a lane, row after row, so the vertical
//     pass is 4 packed fp32 operations on a 3-row register history and the epilogue stores 8 bytes per lane like the convs.
// Wave (j, half) filters channels 16 j .. + 15 of output rows 14 half .. + 13: 17 fragment reads, 34 MFMAs.
template <class T>
__global__ __launch_bounds__(256, T::is32 ? 1 : 2) void dec_zblur_kernel(ConvArgs g) {
  DEC_COPY_PROLOGUE(g, bid)
  DEC_PH_BEGIN
  typedef typename T::elem E;
  typedef typename T::pack8 P8;
  typedef float v2f __attribute__((ext_vector_type(2)));
  constexpr int EB = T::EB, RB = 32 * EB, CB = 8 * EB;
  constexpr int NT = 2, BN = 32, HW = 17, NPIX = HW * HW, NTAPS = 9;
  constexpr int NA = (NPIX * 4 + 255) / 256, NBROWS = NTAPS * BN, NB = (NBROWS * 4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sA = smem;              // [17*17][RB], chunk ^ ((halo column >> 1) & 3)
  unsigned char* const sB = smem + NPIX * RB;  // [9][BN][RB], chunk ^ ((row >> 1) & 3)
  unsigned char* const sZ = smem;              // after the K loop: [32 z rows][32 channels][RB = 32 z columns], chunk ^ ((channel >> 1) & 3)
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: scalar arithmetic
  const int r16 = lane & 15, q = lane >> 4;
  const int Wi = g.Wi, Cin = g.Cin;
  const int tiles_pf = g.tiles_x * g.tiles_y;
  unsigned grp = bid, cb = blockIdx.y;
  if (g.ncb) dec_group_cb(bid, g.ngroups, g.ncb, grp, cb);
  const int f = (int)grp / tiles_pf, rem = (int)grp - f * tiles_pf;
  const int ty = rem / g.tiles_x, tx = rem - ty * g.tiles_x;
  const int n0 = cb * BN;
  const int nchunks = Cin >> 5;
  const int iy0 = ty * 14 - 2, ix0 = tx * 14 - 2;  // halo origin: position (m - 1, n - 1) of the block's first (m, n)

  // ---- staging constants (one tile per workgroup: the border logic runs once).  Chunk e = tid + 256 i: pixel p = e >> 2, pack
  // ch = e & 3; lanes past the last pixel repeat the last pixel's chunk.  A chunk outside the image loads a valid address and
  // is stored as zeros.
  const int ch = tid & 3;
  const bool interior = iy0 >= 0 && iy0 + HW <= g.Hi && ix0 >= 0 && ix0 + HW <= Wi;
  unsigned a_off[NA], a_lds[NA], a_bad = 0u;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int p = min((tid >> 2) + 64 * i, NPIX - 1);
    const int hy = p / HW, hx = p - hy * HW;
    const bool in = (unsigned)(iy0 + hy) < (unsigned)g.Hi && (unsigned)(ix0 + hx) < (unsigned)Wi;
    a_off[i] = (unsigned)((((in ? hy : 2) * Wi + (in ? hx : 2)) * Cin + ch * 8) * EB);  // (2, 2) = position (14 ty, 14 tx): inside
    a_lds[i] = (unsigned)(p * RB + (ch ^ ((hx >> 1) & 3)) * CB);
    a_bad |= (in ? 0u : 1u) << i;
  }
  const unsigned char* const Xo = reinterpret_cast<const unsigned char*>(g.X) + ((long long)(f * g.Hi + iy0) * Wi + ix0) * Cin * EB;
  const int brow = tid >> 2;
  const unsigned b_off = (unsigned)((((brow / BN) * g.Cout + brow % BN) * Cin + ch * 8) * EB);
  const int brow_l = min(brow + 64 * (NB - 1), NBROWS - 1);
  const unsigned b_off_l = (unsigned)((((brow_l / BN) * g.Cout + brow_l % BN) * Cin + ch * 8) * EB);
  const unsigned b_lds = (unsigned)(brow * RB + (ch ^ ((brow >> 1) & 3)) * CB);
  const unsigned b_lds_l = (unsigned)(brow_l * RB + (ch ^ ((brow_l >> 1) & 3)) * CB);
  const size_t b_round = (size_t)(64 / BN) * g.Cout * Cin * EB;
  const unsigned char* const Wn = reinterpret_cast<const unsigned char*>(g.Wt) + (size_t)n0 * Cin * EB;
  // A fragment of input shift (dy, dx) in {-1, 0}^2 and m-tile mt: halo row w*4 + mt + dy + 1, columns r16 + dx + 1
  unsigned fa[2];
#pragma unroll
  for (int dx = 0; dx < 2; ++dx) fa[dx] = (unsigned)(((w * 4) * HW + r16 + dx) * RB + (q ^ (((r16 + dx) >> 1) & 3)) * CB);
  const unsigned fb = (unsigned)(r16 * RB + (q ^ ((r16 >> 1) & 3)) * CB);

  f32x4 acc[4][4][NT];  // [class pu*2 + pv][m-tile][n-tile]: D[position n = 4q + reg][channel j*16 + r16]
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[c][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  P8 ra[NA], rb[NB];
  float dm[NT];
  float4 b4, s4;
  auto issue = [&](int chunk) {
    const unsigned char* const Xc = Xo + (size_t)(chunk << 5) * EB;
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const P8*>(Xc + a_off[i]);
    const unsigned char* const Wc = Wn + (size_t)(chunk << 5) * EB;
#pragma unroll
    for (int i = 0; i < NB - 1; ++i) rb[i] = *reinterpret_cast<const P8*>(Wc + i * b_round + b_off);
    rb[NB - 1] = *reinterpret_cast<const P8*>(Wc + b_off_l);
  };
  issue(0);
  DEC_PH(0);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    __syncthreads();
    if (interior) {
#pragma unroll
      for (int i = 0; i < NA; ++i) *reinterpret_cast<P8*>(sA + a_lds[i]) = ra[i];
    } else {
#pragma unroll
      for (int i = 0; i < NA; ++i) *reinterpret_cast<P8*>(sA + a_lds[i]) = ((a_bad >> i) & 1u) ? T::zero8() : ra[i];
    }
#pragma unroll
    for (int i = 0; i < NB - 1; ++i) *reinterpret_cast<P8*>(sB + b_lds + i * 64 * RB) = rb[i];
    *reinterpret_cast<P8*>(sB + b_lds_l) = rb[NB - 1];
    __syncthreads();
    DEC_PH(1);
    if (chunk + 1 < nchunks) {
      issue(chunk + 1);
    } else {
      // what the phases after the K loop need from memory, requested before the MFMAs that hide the latency
#pragma unroll
      for (int j = 0; j < NT; ++j) dm[j] = g.demod[(size_t)f * g.ldd + n0 + j * 16 + r16];
      const int co = n0 + (w & 1) * 16 + q * 4;
      b4 = *reinterpret_cast<const float4*>(g.bias + co);
      s4 = *reinterpret_cast<const float4*>(g.snext + (size_t)f * g.lds + co);
    }
    // input shift sh = 2*(dy+1) + (dx+1): 0 (-1,-1)  1 (-1,0)  2 (0,-1)  3 (0,0); the A fragments of one shift serve every
    // (class, tap) that reads it: class 0 taps 0..3 = sh 0..3; class 1 (pv=1) taps 4,5 = sh 1,3; class 2 (pu=1) taps 6,7 =
    // sh 2,3; class 3 tap 8 = sh 3
    constexpr int kNum[4] = {1, 2, 2, 4};
    constexpr int kTap[4][4] = {{0, 0, 0, 0}, {1, 4, 0, 0}, {2, 6, 0, 0}, {3, 5, 7, 8}};
    constexpr int kCls[4][4] = {{0, 0, 0, 0}, {0, 1, 0, 0}, {0, 2, 0, 0}, {0, 1, 2, 3}};
#pragma unroll
    for (int sh = 0; sh < 4; ++sh) {
      P8 a[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const P8*>(sA + fa[sh & 1] + (mt + (sh >> 1)) * HW * RB);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u < kNum[sh]) {
          const int t = kTap[sh][u], c = kCls[sh][u];
          P8 b[NT];
#pragma unroll
          for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const P8*>(sB + fb + (t * BN + j * 16) * RB);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[c][mt][j] = T::mfma(a[mt], b[j], acc[c][mt][j]);
        }
      }
    }
    DEC_PH(2);
  }
  // ---- z tile -> LDS.  Lane (r16, q) holds channel j*16 + r16 of z pixels (zr = 2 (w*4 + mt) + pu, zc = 2 (4q + reg) + pv):
  // pv = 0 / 1 interleave to z columns 8q .. 8q + 7, one pack.
  __syncthreads();  // every wave is done with the operand tiles
  unsigned sm = 0u;
  {
    const unsigned zw = (unsigned)(((8 * w) * 32 + r16) * RB + (q ^ ((r16 >> 1) & 3)) * CB);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int pu = 0; pu < 2; ++pu)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const f32x4 ev = acc[pu * 2][mt][j], od = acc[pu * 2 + 1][mt][j];
          const v2f d = v2f{dm[j], dm[j]};
          const v2f z0 = v2f{ev[0], od[0]} * d, z1 = v2f{ev[1], od[1]} * d, z2 = v2f{ev[2], od[2]} * d, z3 = v2f{ev[3], od[3]} * d;
          const float v[8] = {z0.x, z0.y, z1.x, z1.y, z2.x, z2.y, z3.x, z3.y};
          *reinterpret_cast<P8*>(sZ + zw + ((2 * mt + pu) * 32 + j * 16) * RB) = dec_pack8<T>(v, sm);
        }
  }
  __syncthreads();
  DEC_PH(3);
  // ---- FIR + bias + lrelu*sqrt2 + next style.  Output row Yl (tile-local) takes z rows Yl + 1 .. Yl + 4, output column X
  // takes z columns X + 1 .. X + 4, weights g.fir per axis ((1, 3, 3, 1) / 4 for the released checkpoints: the up-sampling gain
  // 4 included).  The horizontal taps are MFMA operands, i.e. rounded to the operand type (exact for (1, 3, 3, 1) / 4).
  {
    const int j2 = w & 1, half = w >> 1;
    const float f0 = g.fir[0], f1 = g.fir[1], f2 = g.fir[2], f3 = g.fir[3];
    // b4, s4: this wave's bias / next style (channels n0 + j2*16 + q*4 .. +3), loaded under the last chunk's MFMAs.  The asm
    // makes the compiler wait for them HERE: left to the first use inside the row loop it put an `s_waitcnt vmcnt(0)` in front
    // of every row's arithmetic, which on gfx9 also waits for the previous row's STORES - 28 store round trips in series, 8.9
    // of the 15 us a 512-px tile took (r03 in-kernel stamps).
    asm volatile("" : "+v"(b4.x), "+v"(b4.y), "+v"(b4.z), "+v"(b4.w), "+v"(s4.x), "+v"(s4.y), "+v"(s4.z), "+v"(s4.w));
    const v2f bs[2] = {v2f{b4.x, b4.y}, v2f{b4.z, b4.w}};
    const v2f sn[2] = {v2f{s4.x * 1.4142135623730951f, s4.y * 1.4142135623730951f},
                       v2f{s4.z * 1.4142135623730951f, s4.w * 1.4142135623730951f}};  // leaky_relu's sqrt(2) rides in the style
    const unsigned zrd = (unsigned)((((half * 14 + 1) * 32) + j2 * 16 + r16) * RB + (q ^ ((r16 >> 1) & 3)) * CB);
    // stores: lane (r16, q) writes 4 channels of output pixel (28 ty + yl, 28 tx + 16 xt + r16); lanes / rows outside the tile's
    // 28 x 28 block or outside the image store into g_dec_sink instead (no branch: the row loop is one basic block)
    unsigned char* const yt = reinterpret_cast<unsigned char*>(g.Y) + ((((size_t)f * g.OH + ty * 28) * g.OW + tx * 28) * g.Cout + n0) * EB;
    unsigned char* const sink = reinterpret_cast<unsigned char*>(g_dec_sink) + lane * 16;
    unsigned char* yp[2];
    unsigned ystep[2];  // bytes between output rows; 0 for a lane that stores into the sink
#pragma unroll
    for (int xt = 0; xt < 2; ++xt) {
#ifdef DEC_ZB_NOSTORE
      const bool ok = false;
#else
      const bool ok = 16 * xt + r16 < 28 && tx * 28 + 16 * xt + r16 < g.OW;
#endif
      yp[xt] = ok ? yt + (size_t)((16 * xt + r16) * g.Cout + j2 * 16 + q * 4) * EB : sink;
      ystep[xt] = ok ? (unsigned)(g.OW * g.Cout * EB) : 0u;
    }
    // FULL: all 28 output rows of the tile exist (every tile but the last row of tiles): no per-row test
    auto filter = [&](auto full_rows, auto symmetric) {
      constexpr bool FULL = decltype(full_rows)::value, SYM = decltype(symmetric)::value;
      P8 bx[2];  // Bx[zc = 8q + k][X = 16 xt + r16] = tap (zc - X - 1), zero outside the band and past the tile's 28 columns
#pragma unroll
      for (int xt = 0; xt < 2; ++xt) {
        bx[xt] = T::zero8();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int t = 8 * q + k - (16 * xt + r16) - 1;
          float kv;
          if constexpr (SYM) kv = (t == 0 || t == 3) ? 0.25f : ((t == 1 || t == 2) ? 0.75f : 0.f);
          else kv = t == 0 ? f0 : (t == 1 ? f1 : (t == 2 ? f2 : (t == 3 ? f3 : 0.f)));
          T::set(bx[xt], k, (16 * xt + r16 < 28) ? kv : 0.f);
        }
      }
      // all 17 z rows of the wave requested at once, then the 34 independent MFMAs back to back, then the vertical pass: written
      // row by row (read -> 2 MFMAs -> arithmetic -> store) the compiler kept that order and every row paid an LDS round trip
      // and an MFMA drain with 2 waves per SIMD to hide them (in-kernel stamps: 4.8 us of a 13-us tile at 512 px)
      P8 za[17];
#pragma unroll
      for (int r = 0; r < 17; ++r) za[r] = *reinterpret_cast<const P8*>(sZ + zrd + r * 32 * RB);
      f32x4 hd[17][2];  // D[channel 4q + reg][X = 16 xt + r16] of z row r
#pragma unroll
      for (int r = 0; r < 17; ++r)
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) hd[r][xt] = T::mfma(za[r], bx[xt], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int r = 3; r < 17; ++r) {
        const int yl = half * 14 + r - 3;
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) {
          v2f v[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const v2f h0 = v2f{hd[r - 3][xt][2 * i], hd[r - 3][xt][2 * i + 1]}, h1 = v2f{hd[r - 2][xt][2 * i], hd[r - 2][xt][2 * i + 1]};
            const v2f h2 = v2f{hd[r - 1][xt][2 * i], hd[r - 1][xt][2 * i + 1]}, h = v2f{hd[r][xt][2 * i], hd[r][xt][2 * i + 1]};
            v2f t;
            if constexpr (SYM) {  // (1, 3, 3, 1) / 4 as literals: what every released checkpoint runs
              t = 0.25f * (h0 + h) + bs[i];
              t = 0.75f * (h1 + h2) + t;
            } else {
              t = f0 * h0 + bs[i];
              t = f3 * h + t;
              t = f1 * h1 + t;
              t = f2 * h2 + t;
            }
            const v2f lo = 0.2f * t;
            v[i] = v2f{__builtin_amdgcn_fmed3f(t.x, lo.x, __builtin_inff()), __builtin_amdgcn_fmed3f(t.y, lo.y, __builtin_inff())} * sn[i];  // leaky_relu(0.2)
          }
          unsigned char* const dst = (FULL || ty * 28 + yl < g.OH) ? yp[xt] + (unsigned)yl * ystep[xt] : sink;
          dec_store4<T>(reinterpret_cast<E*>(dst), v[0].x, v[0].y, v[1].x, v[1].y, sm);
        }
      }
      };
    const bool full = ty * 28 + 28 <= g.OH;
    if (g.fir_sym) {
      if (full) filter(std::true_type{}, std::true_type{});
      else filter(std::false_type{}, std::true_type{});
    } else {
      if (full) filter(std::true_type{}, std::false_type{});
      else filter(std::false_type{}, std::false_type{});
    }
  }
  dec_sat_flush<T>(g.sat, sm);
  DEC_PH(4);
  DEC_PH_COUNT(7);
  DEC_PH_END(0);
}

// Second half of the up-sampling StyledConv: 4x4 FIR (pad 1,1; [1,3,3,1]^2/64 * 4 unless the checkpoint's loader says otherwise) over the
// transposed-conv output z (R+1 x R+1), then + bias, leaky-relu*sqrt2, and the style of the next
// conv (styledecoder.py:209-213,255-258 then 320-325).  A thread makes a 2 (rows) x 4 (pixels) x 8
// (channels) block: 5 rows x 7 columns of z are read once (4.4 loads per output instead of 16), each
// row is filtered horizontally once and feeds both output rows (18 VALU per output element, was 31).
template <class T>
__global__ __launch_bounds__(256) void dec_blur_kernel(const typename T::elem* __restrict__ z, typename T::elem* __restrict__ out,
                                                       int F, int R, int C, const float* __restrict__ bias,
                                                       const float* __restrict__ snext, int lds, unsigned long long* sat,
                                                       float fir0, float fir1, float fir2, float fir3) {
  typedef typename T::pack8 P8;
  const int c8 = C >> 3, xq = R >> 2, yh = R >> 1;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)F * yh * xq * c8) return;
  const int cg = (int)(idx % c8);
  size_t p = idx / c8;
  const int X0 = (int)(p % xq) * 4;
  p /= xq;
  const int Y0 = (int)(p % yh) * 2;
  const int f = (int)(p / yh);
  const int Z = R + 1;
  const float k1[4] = {fir0, fir1, fir2, fir3};  // weight of z[X - 1 + t] in output X ((1, 3, 3, 1) / 4 for blur_kernel [1,3,3,1])
  float acc[2][4][8];
#pragma unroll
  for (int y = 0; y < 2; ++y)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[y][j][i] = 0.f;
#pragma unroll
  for (int a = 0; a < 5; ++a) {  // z rows Y0-1 .. Y0+3
    const int zy = Y0 + a - 1;
    if (zy < 0 || zy > R) continue;
    P8 u[7];
#pragma unroll
    for (int b = 0; b < 7; ++b) {
      const int zx = X0 + b - 1;
      u[b] = (zx >= 0 && zx <= R) ? T::load8(z + ((size_t)(f * Z + zy) * Z + zx) * C + cg * 8) : T::zero8();
    }
    float h[4][8];  // horizontal FIR of this row for the 4 output columns
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) h[j][i] = 0.f;
#pragma unroll
    for (int b = 0; b < 7; ++b) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = T::get(u[b], i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int t = b - j;
        if (t >= 0 && t < 4) {
#pragma unroll
          for (int i = 0; i < 8; ++i) h[j][i] += k1[t] * v[i];
        }
      }
    }
    // row a contributes to output row y with vertical tap a - y
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int t = a - y;
      if (t >= 0 && t < 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[y][j][i] += k1[t] * h[j][i];
      }
    }
  }
  float bs[8], sn[8];
  unsigned sm = 0u;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    bs[i] = bias[cg * 8 + i];
    sn[i] = snext[(size_t)f * lds + cg * 8 + i];
  }
#pragma unroll
  for (int y = 0; y < 2; ++y)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float ov[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) ov[i] = fh_lrelu_s2(acc[y][j][i] + bs[i]) * sn[i];
      T::store8(out + ((size_t)(f * R + Y0 + y) * R + X0 + j) * C + cg * 8, dec_pack8<T>(ov, sm));
    }
  dec_sat_flush<T>(sat, sm);
}

// ------------------------------------------------------------------------------------------
// ToFlow + ToRGB of one level, fused (styledecoder.py:399-425, 378-386, wiring 521-529):
//   o      = 1x1 modconv(x)(no demod) + bias (+ Upsample(prev flow))           3 channels
//   grid   = tanh(o[0:2]) + identity(linspace(-1,1,R)),  mask = sigmoid(o[2])
//   warp   = grid_sample(feat, grid, bilinear, zeros, align_corners=False) * mask
//   xnext  = (warp + x * (1 - mask)) * style_of_next_conv                       -> next level
//   rgb    = lrelu(conv1x1(warp)/sqrt(C) + b1)*sqrt2 + b2 (+ Upsample(prev rgb))
// PIX consecutive pixels of a row per lane group and iteration: every x load of the iteration is
// issued before the first flow is reduced, and every feature gather before the first blend.  The
// per-frame folded weights (flow conv * style, rgb conv, next style: 7 x C floats) live in LDS, not in
// registers, which keeps the kernel at <=128 VGPRs (4 waves per SIMD) - the first version held them in
// 56 registers per lane and ran at ONE wave per SIMD (r01 PMC: waves parked 56 % of their life).
template <class T, int PIX, bool LAST>
__global__ __launch_bounds__(256) void dec_flow_kernel(FlowArgs g) {
  DEC_COPY_PROLOGUE(g, bid)
  __shared__ __attribute__((aligned(16))) float sw[7 * 512];  // [wf0 wf1 wf2 | (unused since round 6: ToRGB reads G) | sn][C]
  const int C = g.C;
  const int lpp = C >> 3;            // lanes per pixel (4..64)
  const int gpb = 256 / lpp;         // lane groups per block
  // Block -> (band of consecutive pixels, frame).  All frames of one band run back to back on ONE XCD (block ids
  // congruent mod 8 share an XCD), so the band's slice of the skip features is fetched into that XCD's L2 once
  // and serves every frame of the batch; a plain (x = pixels, y = frame) grid swept the whole 16.8 MB map per
  // frame and sent the bilinear gathers to the Infinity Cache.
  int f, band;
  {
    const int id = bid, nb = g.nbands;
    if ((nb & 7) == 0) {
      const int slot = id >> 3;
      f = slot % g.F;
      band = (slot / g.F) * 8 + (id & 7);
    } else {
      f = id / nb;
      band = id % nb;
    }
  }
  const int sub = threadIdx.x % lpp, grp = threadIdx.x / lpp;
  const int c0 = sub * 8;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float s = g.sflow[(size_t)f * g.ld_s + c];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      sw[j * C + c] = g.wflow[j * C + c] * s;
    }
    sw[6 * C + c] = g.snext ? g.snext[(size_t)f * g.ld_s + c] : 0.f;
  }
  __syncthreads();
  const int npix = g.R * g.R;
  const int pend = min(npix, (band + 1) * g.band_pix);
  unsigned sm = 0u;
  const FlowFrame<T> ff = dec_flow_frame<T>(g, f);
  const typename T::elem* const xf = reinterpret_cast<const typename T::elem*>(g.x) + (size_t)f * npix * C;
  for (int p0 = band * g.band_pix + grp * PIX; p0 < pend; p0 += gpb * PIX) {
    typename T::pack8 xu[PIX];
#pragma unroll
    for (int k = 0; k < PIX; ++k) xu[k] = dec_load8_b_nt<T>(xf, (unsigned)(((p0 + k) * C + c0) * T::EB));  // read once: leave L2 to the features
    dec_flow_pixels<T, PIX, LAST>(g, ff, sw, p0, sub, lpp, xu, sm);
  }
  dec_sat_flush<T>(g.sat, sm);
  DEC_STAMP_MAX(3);
}

// A fragments of ToFlow's folded weights for dec_conv16_kernel<.., FLOWM = 1>: w[j][c] = wflow[j][c] * s[f][c] (styledecoder.py:
// 399-425: a modulated 1x1 conv WITHOUT demodulation; wflow already / sqrt(C)).  Lane (row r = l & 15, group gq = l >> 4) of
// pack pk holds output j = r & 3 (j == 3: zeros) at K slot s <-> channel (2 pk + (s >> 2)) * 16 + gq * 4 + (s & 3).  16-bit operand
// types: two packs per (pk): hi = rnd(w), lo = rnd((w - hi) * 2048) - x * hi and x * lo are exact in the MFMA's fp32
// accumulator, so the sum is the fp32-weight sum of the VALU kernel it replaces (the lo scale keeps it out of fp16's subnormals).
template <class T>
__global__ __launch_bounds__(64) void dec_flowfrag_kernel(typename T::pack8* __restrict__ out, const float* __restrict__ wflow,
                                                          const float* __restrict__ sflow, int ld_s, int C) {
  constexpr int NHL = T::is32 ? 1 : 2;
  const int f = blockIdx.x, pk = blockIdx.y, l = threadIdx.x, r = l & 15, gq = l >> 4, j = r & 3;
  float w[8], lo[8];
#pragma unroll
  for (int sl = 0; sl < 8; ++sl) {
    const int c = (2 * pk + (sl >> 2)) * 16 + gq * 4 + (sl & 3);
    w[sl] = j < 3 ? wflow[j * C + c] * sflow[(size_t)f * ld_s + c] : 0.f;
  }
  typename T::pack8* const o = out + ((size_t)f * gridDim.y + pk) * NHL * 64 + l;
  typename T::pack8 hi;
#pragma unroll
  for (int sl = 0; sl < 8; ++sl) T::set(hi, sl, w[sl]);
  o[0] = hi;
  if constexpr (NHL == 2) {
    typename T::pack8 lp;
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) {
      lo[sl] = (w[sl] - T::get(hi, sl)) * 2048.f;
      T::set(lp, sl, lo[sl]);
    }
    o[64] = lp;
  }
}

// The last level after dec_conv16_kernel<.., FLOWM = 1>: one lane per pixel, no channels anywhere.  ToFlow's sums come from
// g.oflow, the warped-and-converted features are four taps of G (dec_feat_rgb_kernel), the rest is dec_flow_pixels' owner-lane
// arithmetic: + bias + Upsample(previous flow) -> tanh / sigmoid -> sample position -> ToRGB -> + Upsample(previous rgb) -> frame.
template <class T>
__global__ __launch_bounds__(256) void dec_flowlast_kernel(FlowArgs g) {
  DEC_COPY_PROLOGUE(g, bid)
  const int R = g.R, npix = R * R, Rp = R >> 1;
  // block -> (frame, run of 256 pixels): all frames of a run back to back on one XCD (ids congruent mod 8), like dec_flow_kernel
  const int runs = (npix + 255) >> 8;
  int f, run;
  if ((runs & 7) == 0) {
    const int slot = (int)bid >> 3;
    f = slot % g.F;
    run = (slot / g.F) * 8 + ((int)bid & 7);
  } else {
    f = (int)bid / runs;
    run = (int)bid % runs;
  }
  const int p = run * 256 + (int)threadIdx.x;
  if (p >= npix) return;
  const FlowFrame<T> ff = dec_flow_frame<T>(g, f);
  const int Y = p / R, X = p - Y * R;
  const float4 o = *reinterpret_cast<const float4*>(g.oflow + (((size_t)f * npix + p) << 2));
  float upf[3] = {0.f, 0.f, 0.f};
  if (ff.pflow) up2_tap3(ff.pflow, 0, Rp, Y, X, upf, g.upk_flow);
  const float f0 = o.x + (upf[0] + ff.bf[0]), f1 = o.y + (upf[1] + ff.bf[1]), f2 = o.z + (upf[2] + ff.bf[2]);
  const float fR = (float)R;
  const float sx = fh_tanh_fast<T::is32>(f0) + g.lin[X], sy = fh_tanh_fast<T::is32>(f1) + g.lin[Y];
  const float mk = fh_sigmoid_t<T::is32>(f2);
  const float ix = ((sx + 1.f) * fR - 1.f) * 0.5f, iy = ((sy + 1.f) * fR - 1.f) * 0.5f;  // grid_sample, align_corners=False
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const float axk = ix - fx0, ayk = iy - fy0;
  float4 gt[4];
  float gw[4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int yy = y0 + a, xx = x0 + b;
      const bool in = yy >= 0 && yy < R && xx >= 0 && xx < R;
      const int yc = min(max(yy, 0), R - 1), xc = min(max(xx, 0), R - 1);
      gt[a * 2 + b] = *reinterpret_cast<const float4*>(g.grgb + ((size_t)(yc * R + xc) << 2));
      gw[a * 2 + b] = (a ? ayk : 1.f - ayk) * (b ? axk : 1.f - axk) * mk * (in ? 1.f : 0.f);
    }
  float upr[3] = {0.f, 0.f, 0.f};
  if (ff.prgb) up2_tap3(ff.prgb, 0, Rp, Y, X, upr, g.upk_rgb);
  float r0 = gw[0] * gt[0].x, r1 = gw[0] * gt[0].y, r2 = gw[0] * gt[0].z;
#pragma unroll
  for (int t = 1; t < 4; ++t) {
    r0 += gw[t] * gt[t].x;
    r1 += gw[t] * gt[t].y;
    r2 += gw[t] * gt[t].z;
  }
  const float v0 = fh_lrelu_s2(r0 + ff.b1[0]) + ff.b2[0] + upr[0], v1 = fh_lrelu_s2(r1 + ff.b1[1]) + ff.b2[1] + upr[1],
              v2 = fh_lrelu_s2(r2 + ff.b1[2]) + ff.b2[2] + upr[2];
  const unsigned po = (unsigned)p;
  if (g.write_pyr) {
    *reinterpret_cast<float4*>(ff.flow_out + po * 4u) = float4{f0, f1, f2, 0.f};
    *reinterpret_cast<float4*>(ff.rgb_out + po * 4u) = float4{v0, v1, v2, 0.f};
  }
  if (g.final_mode == 1) {
    typedef float f3v __attribute__((ext_vector_type(3)));
    f3v o3 = {fminf(fmaxf(v0, -1.f), 1.f) * 0.5f + 0.5f, fminf(fmaxf(v1, -1.f), 1.f) * 0.5f + 0.5f,
              fminf(fmaxf(v2, -1.f), 1.f) * 0.5f + 0.5f};
    __builtin_memcpy(ff.final_hwc + po * 3u, &o3, 12);
  } else if (g.final_mode == 2) {
    float* fo = ff.final_chw + po;
    fo[0] = v0;
    fo[npix] = v1;
    fo[2 * (size_t)npix] = v2;
  }
  DEC_STAMP_MAX(3);
}

// G = ToRGB's 1x1 conv (styledecoder.py:368-386: unmodulated, weight / sqrt(C) folded into wrgb) applied to a level's skip
// features, once per clip: G[p][j] = sum_c wrgb[j][c] * feat[p][c], fp32, 4 floats per pixel (one 16-byte tap for dec_flow_kernel).
template <class T>
__global__ __launch_bounds__(256) void dec_feat_rgb_kernel(float* __restrict__ out, const typename T::elem* __restrict__ feat,
                                                           const float* __restrict__ wrgb, int C, int npix) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= npix) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int c = 0; c < C; c += 8) {
    const typename T::pack8 v = T::load8(feat + (size_t)p * C + c);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x = T::get(v, i);
      a0 += wrgb[c + i] * x;
      a1 += wrgb[C + c + i] * x;
      a2 += wrgb[2 * C + c + i] * x;
    }
  }
  *reinterpret_cast<float4*>(out + (size_t)p * 4) = float4{a0, a1, a2, 0.f};
}

// Encoder skip feature NCHW fp32 -> NHWC 16-bit (once per clip).
template <class T>
__global__ void dec_feat_pack_kernel(typename T::elem* __restrict__ out, const float* __restrict__ in, int C, int HW,
                                     unsigned long long* sat) {
  const int idx = (blockIdx.x * blockDim.x + threadIdx.x) * 4;  // 4 consecutive channels of one pixel
  if (idx >= C * HW) return;
  const int c = idx % C, p = idx / C;
  const float* ip = in + (size_t)c * HW + p;
  unsigned sm = 0u;
  dec_store4<T>(out + idx, ip[0], ip[HW], ip[2 * (size_t)HW], ip[3 * (size_t)HW], sm);
  dec_sat_flush<T>(sat, sm);
}

// ------------------------------------------------------------------------------------------
// Layout converters of the unit-op test hooks (float_dec_debug_*): fp32 NCHW <-> NHWC T::elem (optionally times a per-frame
// style row, what a producer's epilogue does), and the 3-channel pyramids <-> their 4-floats-per-pixel form.
template <class T>
__global__ void dec_dbg_pack_kernel(typename T::elem* __restrict__ out, const float* __restrict__ in, const float* __restrict__ s,
                                    int ld_s, int F, int C, int HW, unsigned long long* sat) {
  const size_t idx = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (idx >= (size_t)F * C * HW) return;
  const int c = (int)(idx % C);
  const size_t fp = idx / C;
  const int p = (int)(fp % HW), f = (int)(fp / HW);
  const float* ip = in + ((size_t)f * C + c) * HW + p;
  float v[4] = {ip[0], ip[HW], ip[2 * (size_t)HW], ip[3 * (size_t)HW]};
  if (s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] *= s[(size_t)f * ld_s + c + i];
  }
  unsigned sm = 0u;
  dec_store4<T>(out + idx, v[0], v[1], v[2], v[3], sm);
  dec_sat_flush<T>(sat, sm);
}
template <class T>
__global__ void dec_dbg_unpack_kernel(float* __restrict__ out, const typename T::elem* __restrict__ in, int F, int C, int HW) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)F * C * HW) return;
  const int c = (int)(idx % C);
  const size_t fp = idx / C;
  const int p = (int)(fp % HW), f = (int)(fp / HW);
  out[((size_t)f * C + c) * HW + p] = T::to_float(in[idx]);
}
// dir 0: (F,3,HW) -> [F][HW][4];  dir 1: back
static __global__ void dec_dbg_pyr_kernel(float* __restrict__ out, const float* __restrict__ in, int F, int HW, int dir) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)F * HW) return;
  const int p = (int)(idx % HW), f = (int)(idx / HW);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (dir == 0) out[idx * 4 + c] = in[((size_t)f * 3 + c) * HW + p];
    else out[((size_t)f * 3 + c) * HW + p] = in[idx * 4 + c];
  }
  if (dir == 0) out[idx * 4 + 3] = 0.f;
}
