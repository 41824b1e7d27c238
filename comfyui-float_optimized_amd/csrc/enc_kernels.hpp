// HIP kernels of the appearance encoder (reference encoder.py:146-247: ConvLayer / ResBlock /
// EncoderApp / Encoder.fc), run once per clip.  Activations are NHWC in the operand type T like the decoder's (16-bit in
// production, fp32 in the verification mode: the same kernels on T::elem / T::pack8), so the skip features can be handed to
// the decoder without a round trip; accumulation is fp32.
#pragma once
#include "common.hpp"

// ------------------------------------------------------------------------------------------
// convs.0 = ConvLayer(3, C, 1): EqualConv2d 1x1 (scale 1/sqrt(3), no bias) + FusedLeakyReLU
// (encoder.py:212, 146-181).  Reads the fp32 NCHW image, writes NHWC 16-bit (+ optional fp32 NCHW).
template <class T>
__global__ __launch_bounds__(256) void enc_first_kernel(const float* __restrict__ img, const float* __restrict__ w /*[C][3] scaled*/,
                                                        const float* __restrict__ bias, typename T::elem* __restrict__ out,
                                                        float* __restrict__ out_f32, int HW, int C, unsigned long long* sat) {
  const int c8 = C >> 3;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)HW * c8) return;
  const int cg = (int)(idx % c8);
  const size_t p = idx / c8;
  const float r = img[p], g = img[(size_t)HW + p], b = img[2 * (size_t)HW + p];
  typename T::pack8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = cg * 8 + i;
    const float v = fh_lrelu_s2(w[c * 3 + 0] * r + w[c * 3 + 1] * g + w[c * 3 + 2] * b + bias[c]);
    T::set(o, i, v);
    if (out_f32) out_f32[(size_t)c * HW + p] = v;
  }
  unsigned rm = 0u;
  fh_track_pack<T>(rm, o);
  fh_range_flush<T>(sat, rm);
  T::store8(out + p * C + cg * 8, o);
}

// ------------------------------------------------------------------------------------------
// Blur(kernel, pad=(p,p)) of a down-sampling ConvLayer (encoder.py:59-75, 160-166): zero-pad by p, correlate with the FLIPPED
// 4 x 4 FIR (upfirdn2d, encoder.py:28-29); output (R + 2p - 3)^2.  One thread = one pixel x 8 channels.
// EncFir: k[a][b] = weight of in[Y + a - pad][X + b - pad] = the layer's `kernel` buffer flipped in both axes
// (make_kernel([1,3,3,1]) = (1,3,3,1) (x) (1,3,3,1) / 64 for every released checkpoint; exact in fp32 either way).
struct EncFir {
  float k[16];
};
template <class T>
__global__ __launch_bounds__(256) void enc_blur_kernel(const typename T::elem* __restrict__ in, typename T::elem* __restrict__ out,
                                                       int R, int C, int pad, EncFir fir, unsigned long long* sat) {
  const int Ro = R + 2 * pad - 3, c8 = C >> 3;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)Ro * Ro * c8) return;
  const int cg = (int)(idx % c8);
  const size_t p = idx / c8;
  const int X = (int)(p % Ro), Y = (int)(p / Ro);
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int iy = Y + a - pad;
    if (iy < 0 || iy >= R) continue;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int ix = X + b - pad;
      if (ix < 0 || ix >= R) continue;
      const typename T::pack8 u = T::load8(in + ((size_t)iy * R + ix) * C + cg * 8);
      const float kk = fir.k[a * 4 + b];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += kk * T::get(u, i);
    }
  }
  typename T::pack8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) T::set(o, i, acc[i]);
  unsigned rm = 0u;
  fh_track_pack<T>(rm, o);
  fh_range_flush<T>(sat, rm);
  T::store8(out + p * C + cg * 8, o);
}

// ------------------------------------------------------------------------------------------
// EqualConv2d as an implicit GEMM on MFMA (encoder.py:88-106): k x k taps, stride 1|2, zero padding.
//   M = output pixels (64 per workgroup, 16 per wave), N = NT*16 output channels, K = taps x Cin in
//   32-channel steps.  Both operand fragments come straight from global memory: a lane's 16 bytes are 8
//   consecutive channels of one pixel (A) or of one (tap, output channel) weight row (B), so four lanes
//   cover a 64-byte run; neighbouring taps/pixels re-hit L1/L2.  The encoder runs once per clip
//   (39 GFLOP), so there is no LDS staging here - the per-frame decoder convs are the tuned ones.
//   Operands are swapped (D = W * X^T): a lane ends with 4 consecutive channels of one pixel.
// Epilogue (all optional): + bias, leaky_relu(0.2) * sqrt(2); (y + skip) / sqrt(2) of the ResBlock
// (encoder.py:196-197); NHWC 16-bit store and/or fp32 NCHW store (the reference's feature format).
struct EncConvArgs {
  const void* X;    // [Hi][Wi][Cin] T::elem
  const void* W;    // [k*k][Cout][Cin] T::elem, 1/sqrt(Cin k^2) folded in
  void* Y;          // [Ho][Wo][Cout] T::elem or nullptr
  float* Yf32;      // [Cout][Ho][Wo] or nullptr
  const float* bias;  // [Cout] or nullptr (with act)
  const void* skip;   // [Ho][Wo][Cout] T::elem or nullptr
  int Hi, Wi, Cin, Cout, Ho, Wo, k, stride, pad, act;
  unsigned long long* sat;  // range counter of the 16-bit stores (fh_range_flush)
};

template <class T, int NT>
__global__ __launch_bounds__(256) void enc_conv_kernel(EncConvArgs g) {
  typedef typename T::elem E;
  typedef typename T::pack8 P8;
  const E* const gX = reinterpret_cast<const E*>(g.X);
  const E* const gW = reinterpret_cast<const E*>(g.W);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int npix = g.Ho * g.Wo;
  const int m = blockIdx.x * 64 + w * 16 + r16;
  const int n0 = blockIdx.y * NT * 16;
  const bool mvalid = m < npix;
  const int oy = mvalid ? m / g.Wo : 0, ox = mvalid ? m - oy * g.Wo : 0;
  f32x4 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nchunk = g.Cin >> 5;
  for (int ty = 0; ty < g.k; ++ty) {
    const int iy = oy * g.stride - g.pad + ty;
    for (int tx = 0; tx < g.k; ++tx) {
      const int ix = ox * g.stride - g.pad + tx;
      const bool ok = mvalid && iy >= 0 && iy < g.Hi && ix >= 0 && ix < g.Wi;
      const E* xa = gX + ((size_t)(ok ? iy : 0) * g.Wi + (ok ? ix : 0)) * g.Cin + q * 8;
      const E* wb = gW + ((size_t)(ty * g.k + tx) * g.Cout + n0 + r16) * g.Cin + q * 8;
      for (int c = 0; c < nchunk; c += 2) {
        P8 a[2], b[2][NT];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool live = c + h < nchunk;
          a[h] = (ok && live) ? T::load8(xa + (c + h) * 32) : T::zero8();
#pragma unroll
          for (int j = 0; j < NT; ++j) b[h][j] = live ? T::load8(wb + (size_t)j * 16 * g.Cin + (c + h) * 32) : T::zero8();
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[j] = T::mfma(b[h][j], a[h], acc[j]);
      }
    }
  }
  if (!mvalid) return;
  const float inv_s2 = 0.70710678118654752f;
  unsigned rm = 0u;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int co = n0 + j * 16 + q * 4;
    float v[4] = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
    if (g.act) {
      const float4 bb = *reinterpret_cast<const float4*>(g.bias + co);
      v[0] = fh_lrelu_s2(v[0] + bb.x);
      v[1] = fh_lrelu_s2(v[1] + bb.y);
      v[2] = fh_lrelu_s2(v[2] + bb.z);
      v[3] = fh_lrelu_s2(v[3] + bb.w);
    }
    if (g.skip) {
      const E* sp = reinterpret_cast<const E*>(g.skip) + (size_t)m * g.Cout + co;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (v[r] + T::to_float(sp[r])) * inv_s2;
    }
    if (g.Y) fh_store4<T>(reinterpret_cast<E*>(g.Y) + (size_t)m * g.Cout + co, v[0], v[1], v[2], v[3], rm);
    if (g.Yf32) {
#pragma unroll
      for (int r = 0; r < 4; ++r) g.Yf32[(size_t)(co + r) * npix + m] = v[r];
    }
  }
  fh_range_flush<T>(g.sat, rm);
}

// ------------------------------------------------------------------------------------------
// y[n] = alpha * sum_k x[k] W[n][k] + b[n], fp32, one wave per output: the EqualLinear chain of
// Encoder.fc (encoder.py:242-247, no activation) and Direction's lambda @ Q^T (styledecoder.py:441-444).
__global__ __launch_bounds__(256) void enc_linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ b, float alpha, float* __restrict__ y, int N,
                                                         int K) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += x[k] * W[(size_t)n * K + k];
  s = wave_sum(s);
  if (lane == 0) y[n] = alpha * s + (b ? b[n] : 0.f);
}

// NHWC 16-bit -> fp32 NCHW (reference feature format), for callers that want the reference's tensors.
template <class T>
__global__ void enc_unpack_kernel(const typename T::elem* __restrict__ in, float* __restrict__ out, int HW, int C) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)HW * C) return;
  const int c = (int)(idx / HW);
  const size_t p = idx % HW;
  out[idx] = T::to_float(in[p * C + c]);
}
