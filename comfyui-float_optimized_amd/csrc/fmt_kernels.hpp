// HIP kernels of the Flow-Matching-Transformer evaluation (reference FMT.py:277-401).
//
// Shapes: M = Bc * n_tok rows (Bc = 1/3/4 CFG rows x 60 tokens), so every linear layer is a
// weight-streaming GEMM with a short M.  One workgroup owns ALL rows of a 32-column slab of the
// output and its 4 waves split K; the weights are therefore read from HBM exactly once (non-temporal),
// partial sums meet in LDS, and the epilogue (bias / GELU / gate*residual / CFG+Euler) is applied once
// per output element, in a fixed order -> bitwise reproducible, no atomics.
#pragma once
#include "common.hpp"

enum {
  EPI_F32 = 0,       // out_f32 = acc + bias
  EPI_T16 = 1,       // out16   = T(acc + bias)
  EPI_SILU_T16 = 2,  // out16   = T(silu(acc + bias))
  EPI_GELU_T16 = 3,  // out16   = T(gelu_tanh(acc + bias))
  EPI_GATE_RES = 4,  // out_f32 += gate * (acc + bias)                       (FMT.py:174-175)
  EPI_XEMBED = 5,    // out_f32[b*ntok + r] = acc + bias + pos[r], b < bc     (FMT.py:319-320)
  EPI_CFG = 6        // CFG combine (+ Euler update) on the final linear      (FMT.py:375-399)
};

struct GemmArgs {
  const u16* A;   // [rows][lda], K contiguous, rows padded to a multiple of 16 (pad rows are zero)
  const u16* W;   // [N][K]  (torch Linear layout), K padded to a multiple of 256
  const float* bias;
  int lda, K, M, N;
  float* out_f32;
  int ldo;
  u16* out16;
  int ldo16;
  const float* gate;
  int ldg;
  const float* pos;
  int bc, ntok, n_prev;
  // EPI_CFG
  float a_cfg, r_cfg, e_cfg, dt;
  float* vout;   // (ntok, N) combined velocity (float_fmt_eval) or nullptr
  float* xcur;   // (ntok - n_prev, N) Euler state or nullptr
  u16* xin16;    // next evaluation's x_embedder input rows (ntok, ldx) or nullptr
  int ldx;
};

template <class T, int MT, int NT, int EPI>
__global__ __launch_bounds__(256) void fmt_gemm_kernel(GemmArgs g) {
  constexpr int BN = NT * 16;
  constexpr int ROWS = MT * 16;
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4][ROWS][BN]
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * BN;
  const int m0 = blockIdx.y * ROWS;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int kw = g.K >> 2;
  const u16* Ap = g.A + (size_t)(m0 + r16) * g.lda + w * kw + q * 8;
  const u16* Wp = g.W + (size_t)(n0 + r16) * g.K + w * kw + q * 8;
  // Two k-steps of fragments in flight per wave (kw is a multiple of 64): the loads of step k+1 are
  // issued before the MFMAs of step k, so ~28 x 16 B per lane are outstanding against HBM/L2.
  u32x4 a0[MT], b0[NT], a1[MT], b1[NT];
  auto load = [&](u32x4 (&a)[MT], u32x4 (&b)[NT], int k) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
      b[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(Wp + (size_t)j * 16 * g.K + k));
#pragma unroll
    for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const u32x4*>(Ap + (size_t)i * 16 * g.lda + k);
  };
  auto mma = [&](const u32x4 (&a)[MT], const u32x4 (&b)[NT]) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = T::mfma(a[i], b[j], acc[i][j]);
  };
  load(a0, b0, 0);
  for (int k = 0; k < kw; k += 64) {
    load(a1, b1, k + 32);
    mma(a0, b0);
    if (k + 64 < kw) load(a0, b0, k + 64);
    mma(a1, b1);
  }

  // C/D map of mfma_f32_16x16x32: col = lane & 15, row = (lane >> 4) * 4 + reg
  float* my = red + w * (ROWS * BN);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) my[(i * 16 + q * 4 + r) * BN + j * 16 + r16] = acc[i][j][r];
  __syncthreads();

  constexpr int S = ROWS * BN;
  if constexpr (EPI == EPI_CFG) {
    // rows of the tile: b * ntok + i.  Combine the CFG rows of token i, then (optionally) Euler.
    for (int idx = threadIdx.x; idx < g.ntok * BN; idx += 256) {
      const int i = idx / BN, c = idx % BN, n = n0 + c;
      const float bias = g.bias[n];
      float v[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (b < g.bc) {
          const int o = (b * g.ntok + i) * BN + c;
          v[b] = red[o] + red[S + o] + red[2 * S + o] + red[3 * S + o] + bias;
        } else {
          v[b] = 0.f;
        }
      }
      float out;
      if (g.bc == 1) {
        out = v[0];
      } else if (g.bc == 3) {  // [uncond | all | audio-only]
        out = v[0] + g.a_cfg * (v[2] - v[0]) + g.e_cfg * (v[1] - v[2]);
      } else {  // [null-ref | uncond | all | audio-only]
        out = v[0] + g.r_cfg * (v[1] - v[0]) + g.a_cfg * (v[3] - v[1]) + g.e_cfg * (v[2] - v[3]);
      }
      if (g.vout) g.vout[(size_t)i * g.N + n] = out;
      if (g.xcur && i >= g.n_prev) {
        const size_t xo = (size_t)(i - g.n_prev) * g.N + n;
        const float xn = g.xcur[xo] + g.dt * out;  // x_{k+1} = x_k + dt * v  (fixed-grid Euler)
        g.xcur[xo] = xn;
        g.xin16[(size_t)i * g.ldx + n] = T::from_float(xn);
      }
    }
  } else {
    for (int idx = threadIdx.x; idx < S; idx += 256) {
      const int r = idx / BN, c = idx % BN;
      const int row = m0 + r, n = n0 + c;
      if (row >= g.M) continue;
      const float v = red[idx] + red[S + idx] + red[2 * S + idx] + red[3 * S + idx] + g.bias[n];
      if constexpr (EPI == EPI_F32) {
        g.out_f32[(size_t)row * g.ldo + n] = v;
      } else if constexpr (EPI == EPI_T16) {
        g.out16[(size_t)row * g.ldo16 + n] = T::from_float(v);
      } else if constexpr (EPI == EPI_SILU_T16) {
        g.out16[(size_t)row * g.ldo16 + n] = T::from_float(fh_silu(v));
      } else if constexpr (EPI == EPI_GELU_T16) {
        g.out16[(size_t)row * g.ldo16 + n] = T::from_float(fh_gelu_tanh(v));
      } else if constexpr (EPI == EPI_GATE_RES) {
        const size_t o = (size_t)row * g.ldo + n;
        g.out_f32[o] = g.out_f32[o] + g.gate[(size_t)row * g.ldg + n] * v;
      } else if constexpr (EPI == EPI_XEMBED) {
        const float val = v + g.pos[(size_t)row * g.N + n];
        for (int b = 0; b < g.bc; ++b) g.out_f32[(size_t)(b * g.ntok + row) * g.ldo + n] = val;
      }
    }
  }
}

// LayerNorm (no affine, biased variance, eps 1e-6) + framewise modulate, one wave per token row
// (FMT.py:157,168-169,174-175,197).  out = T( (x-mu)*rstd * (1 + scale[row]) + shift[row] ).
template <class T, int NV>
__global__ __launch_bounds__(256) void fmt_lnmod_kernel(const float* __restrict__ x, int M, const float* __restrict__ shift,
                                                        const float* __restrict__ scale, int ldm, u16* __restrict__ out,
                                                        int ldo) {
  constexpr int D = NV * 256;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (size_t)row * D;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = *reinterpret_cast<const float4*>(xr + i * 256 + lane * 4);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mu = wave_sum(s) * (1.f / D);
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
    s2 += (a * a + b * b) + (c * c + d * d);
  }
  const float rstd = rsqrtf(wave_sum(s2) * (1.f / D) + 1e-6f);
  const float* sh = shift + (size_t)row * ldm;
  const float* sc = scale + (size_t)row * ldm;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 256 + lane * 4;
    const float4 a = *reinterpret_cast<const float4*>(sh + c);
    const float4 b = *reinterpret_cast<const float4*>(sc + c);
    ushort4 o;
    o.x = T::from_float((v[i].x - mu) * rstd * (1.f + b.x) + a.x);
    o.y = T::from_float((v[i].y - mu) * rstd * (1.f + b.y) + a.y);
    o.z = T::from_float((v[i].z - mu) * rstd * (1.f + b.z) + a.z);
    o.w = T::from_float((v[i].w - mu) * rstd * (1.f + b.w) + a.w);
    *reinterpret_cast<ushort4*>(out + (size_t)row * ldo + c) = o;
  }
}

// Banded attention (FMT.py:71-88 with the mask of FMT.py:15-19): query i sees keys |i-j| <= window.
// One workgroup per (cfg row b, head h); 4 lanes per query, 32 of the 128 head dims each.  With at
// most 2*window+1 keys per query this is 0.3 % of the evaluation's flops - LDS/MFMA tiling would
// only add latency, so q/k/v are read straight from L2 in 64-byte pieces and the softmax is online.
template <class T>
__global__ __launch_bounds__(256) void fmt_attn_kernel(const u16* __restrict__ qkv, int ld, u16* __restrict__ out, int ldo,
                                                       int ntok, int heads, int D, int window) {
  constexpr int HD = 128, PD = 32;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int qi = threadIdx.x >> 2, part = threadIdx.x & 3;
  if (qi >= ntok) return;  // whole quads leave together
  const int d0 = h * HD + part * PD;
  const u16* qp = qkv + (size_t)(b * ntok + qi) * ld + d0;
  float qf[PD];
#pragma unroll
  for (int i = 0; i < PD / 8; ++i) {
    const uint4 u = *reinterpret_cast<const uint4*>(qp + i * 8);
    const u16* e = reinterpret_cast<const u16*>(&u);
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[i * 8 + j] = T::to_float(e[j]);
  }
  const float scale = rsqrtf((float)HD);
  float m = -INFINITY, l = 0.f;
  float o[PD];
#pragma unroll
  for (int i = 0; i < PD; ++i) o[i] = 0.f;
  const int j0 = max(0, qi - window), j1 = min(ntok - 1, qi + window);
  for (int kj = j0; kj <= j1; ++kj) {
    const u16* kp = qkv + (size_t)(b * ntok + kj) * ld + D + d0;
    const u16* vp = kp + D;
    float dot = 0.f;
    uint4 vu[PD / 8];
#pragma unroll
    for (int i = 0; i < PD / 8; ++i) {
      const uint4 u = *reinterpret_cast<const uint4*>(kp + i * 8);
      vu[i] = *reinterpret_cast<const uint4*>(vp + i * 8);
      const u16* e = reinterpret_cast<const u16*>(&u);
#pragma unroll
      for (int j = 0; j < 8; ++j) dot += qf[i * 8 + j] * T::to_float(e[j]);
    }
    dot += __shfl_xor(dot, 1, 64);
    dot += __shfl_xor(dot, 2, 64);
    const float s = dot * scale;
    const float mn = fmaxf(m, s);
    const float alpha = __expf(m - mn);
    const float p = __expf(s - mn);
    l = l * alpha + p;
    m = mn;
#pragma unroll
    for (int i = 0; i < PD / 8; ++i) {
      const u16* e = reinterpret_cast<const u16*>(&vu[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[i * 8 + j] = o[i * 8 + j] * alpha + p * T::to_float(e[j]);
    }
  }
  const float inv = 1.f / l;
  u16* op = out + (size_t)(b * ntok + qi) * ldo + d0;
#pragma unroll
  for (int i = 0; i < PD / 8; ++i) {
    uint4 u;
    u16* e = reinterpret_cast<u16*>(&u);
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = T::from_float(o[i * 8 + j] * inv);
    *reinterpret_cast<uint4*>(op + i * 8) = u;
  }
}

// Condition rows for c_embedder: [wr | wa | we | 0-pad] per (cfg row b, token i) with the CFG nulling
// pattern given as bit masks over b (FMT.py:322-333, 360-373, 382-392).
template <class T>
__global__ void fmt_build_cond_kernel(u16* __restrict__ out, int ld, int bc, int ntok, int n_prev, int dim_w, int dim_a,
                                      int dim_e, const float* __restrict__ wr, const float* __restrict__ wa,
                                      const float* __restrict__ prev_wa, const float* __restrict__ we, int we_len,
                                      const float* __restrict__ prev_we, unsigned wr_mask, unsigned wa_mask,
                                      unsigned we_mask) {
  const int row = blockIdx.x;
  const int b = row / ntok, i = row % ntok;
  const bool on_r = (wr_mask >> b) & 1, on_a = (wa_mask >> b) & 1, on_e = (we_mask >> b) & 1;
  for (int c = threadIdx.x; c < ld; c += blockDim.x) {
    float v = 0.f;
    if (c < dim_w) {
      v = on_r ? wr[c] : 0.f;
    } else if (c < dim_w + dim_a) {
      const int k = c - dim_w;
      // prev_wa is replicated UN-nulled into every CFG row; only the current window is nulled
      // (prev_wa_cat = [prev_wa]*3, FMT.py:366 vs audio_cat, FMT.py:360)
      if (i < n_prev) v = prev_wa[i * dim_a + k];
      else if (on_a) v = wa[(i - n_prev) * dim_a + k];
    } else if (c < dim_w + dim_a + dim_e) {
      const int k = c - dim_w - dim_a;
      if (on_e) {
        if (we_len == 1) v = we[k];
        else v = (i < n_prev) ? prev_we[i * dim_e + k] : we[(i - n_prev) * dim_e + k];
      }
    }
    out[(size_t)row * ld + c] = T::from_float(v);
  }
}

// Sinusoidal timestep features [cos(t f_k) | sin(t f_k)], k < 128 (FMT.py:118-123).
template <class T>
__global__ void fmt_tsin_kernel(u16* __restrict__ out, const float* __restrict__ ts, const float* __restrict__ freqs,
                                int n_steps) {
  const int s = blockIdx.x;
  if (s >= n_steps) return;
  const int k = threadIdx.x;  // 0..255
  const float arg = ts[s] * freqs[k & 127];
  out[(size_t)s * 256 + k] = T::from_float(k < 128 ? cosf(arg) : sinf(arg));
}

// sc16[row] = T(silu(t_emb + c_cond[row]))   (c = t + c_embedder(.), FMT.py:335; SiLU of FMT.py:164,187)
template <class T>
__global__ void fmt_silu_c_kernel(u16* __restrict__ out, const float* __restrict__ temb, const float* __restrict__ ccond,
                                  int M, int D) {
  const int idx = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (idx >= M * D) return;
  const int c = idx % D;
  const float4 a = *reinterpret_cast<const float4*>(ccond + idx);
  const float4 t = *reinterpret_cast<const float4*>(temb + c);
  ushort4 o;
  o.x = T::from_float(fh_silu(a.x + t.x));
  o.y = T::from_float(fh_silu(a.y + t.y));
  o.z = T::from_float(fh_silu(a.z + t.z));
  o.w = T::from_float(fh_silu(a.w + t.w));
  *reinterpret_cast<ushort4*>(out + idx) = o;
}

// Euler state and x_embedder input rows for a new window: xcur = x0; xin16 = [prev_x ; x0].
template <class T>
__global__ void fmt_init_x_kernel(float* __restrict__ xcur, u16* __restrict__ xin16, int ldx, const float* __restrict__ x0,
                                  const float* __restrict__ prev_x, int n_prev, int n_cur, int dim_w) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int ntok = n_prev + n_cur;
  if (idx >= ntok * dim_w) return;
  const int i = idx / dim_w, c = idx % dim_w;
  float v;
  if (i < n_prev) {
    v = prev_x[i * dim_w + c];
  } else {
    v = x0[(i - n_prev) * dim_w + c];
    xcur[(i - n_prev) * dim_w + c] = v;
  }
  xin16[(size_t)i * ldx + c] = T::from_float(v);
}

// Window slice with replicate padding along time (FLOAT.py:224-227): dst[i] = src[min(t0+i, T-1)].
__global__ void fmt_slice_pad_kernel(float* __restrict__ dst, const float* __restrict__ src, int t0, int T, int n, int dim) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * dim) return;
  const int i = idx / dim, c = idx % dim;
  const int t = min(t0 + i, T - 1);
  dst[idx] = src[(size_t)t * dim + c];
}
