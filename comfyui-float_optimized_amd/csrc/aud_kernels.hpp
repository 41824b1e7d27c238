// HIP kernels of the audio conditioning encoder: wav2vec2-base as the
// reference drives it (src/nodes/models/wav2vec2.py:33-98,184-197: feature extractor -> linear
// interpolation to T video frames -> feature projection -> transformer encoder with all hidden states) and the
// 12x768 -> 512 audio projection (FLOAT.py:338-342).  The arithmetic of the transformers package is restated
// from its published module definitions (Wav2Vec2FeatureEncoder / FeatureProjection / PositionalConvEmbedding /
// EncoderLayer, feat_extract_norm = "group", do_stable_layer_norm = false: the bundled wav2vec2_base config).
// The transformer's Linear layers run on the FMT's weight-streaming GEMM (fmt_gemm.hpp); this file holds the rest.
// Every kernel is written against the operand type T (T::elem / T::pack8): 16-bit activations in production, FP32 = the
// verification mode (the same launch chain with 4-byte elements on v_mfma_f32_16x16x4_f32).
#pragma once
#include "common.hpp"
#include "fmt_pack.hpp"

// ------------------------------------------------------------------------------------------
// Layer 0 of the feature extractor: Conv1d(1 -> C, k, stride, no bias) -> GroupNorm(C groups = per-channel
// statistics over time, affine) -> GELU.  The conv is 10 MACs per output, so it is recomputed by the
// statistics pass and by the apply pass instead of being stored in fp32.
// Pass 1: partial (sum, sum of squares) per (time chunk, channel).
template <int KW>
__global__ __launch_bounds__(256) void aud_conv0_stats_kernel(const float* __restrict__ x, int n_samples, const float* __restrict__ w,
                                                              int stride, int L, int C, int tchunk, float* __restrict__ part) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= C) return;
  float wk[KW];
#pragma unroll
  for (int k = 0; k < KW; ++k) wk[k] = w[c * KW + k];
  const int t0 = blockIdx.x * tchunk, t1 = min(t0 + tchunk, L);
  float s = 0.f, s2 = 0.f;
  for (int t = t0; t < t1; ++t) {
    const float* xp = x + (size_t)t * stride;
    float y = 0.f;
#pragma unroll
    for (int k = 0; k < KW; ++k) y += wk[k] * xp[k];
    s += y;
    s2 += y * y;
  }
  part[((size_t)blockIdx.x * C + c) * 2 + 0] = s;
  part[((size_t)blockIdx.x * C + c) * 2 + 1] = s2;
}

// Pass 2: per-channel scale = gamma * rstd, shift = beta - mean * scale (GroupNorm eps), fp64 final sums.
// One WAVE per channel (its lanes stride over the chunks, fp64 partial sums, shuffle reduction): as one THREAD per channel the
// 2 x nchunk dependent loads of a 10-s clip (500 chunks) took 122 us.
__global__ __launch_bounds__(256) void aud_gn_final_kernel(const float* __restrict__ part, int nchunk, int C, int L,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                           float* __restrict__ scale_shift) {
  const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  double s = 0.0, s2 = 0.0;
  for (int i = lane; i < nchunk; i += 64) {
    const float2 p = *reinterpret_cast<const float2*>(part + ((size_t)i * C + c) * 2);
    s += (double)p.x;
    s2 += (double)p.y;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if (lane != 0) return;
  const double mean = s / L;
  const double var = fmax(s2 / L - mean * mean, 0.0);  // biased variance, like torch group_norm
  const float sc = gamma[c] * (float)(1.0 / sqrt(var + (double)eps));
  scale_shift[c] = sc;
  scale_shift[C + c] = beta[c] - (float)mean * sc;
}

// Pass 3: y = gelu(conv * scale + shift) -> NLC 16-bit [L][C]; one thread = one time step x 8 channels.
template <class T, int KW>
__global__ __launch_bounds__(256) void aud_conv0_apply_kernel(const float* __restrict__ x, const float* __restrict__ w, int stride, int L,
                                                              int C, const float* __restrict__ scale_shift,
                                                              typename T::elem* __restrict__ out, unsigned long long* sat) {
  const int c8 = C >> 3;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)L * c8) return;
  const int cg = (int)(idx % c8);
  const size_t t = idx / c8;
  float xv[KW];
#pragma unroll
  for (int k = 0; k < KW; ++k) xv[k] = x[t * stride + k];
  typename T::pack8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = cg * 8 + i;
    float y = 0.f;
#pragma unroll
    for (int k = 0; k < KW; ++k) y += w[c * KW + k] * xv[k];
    T::set(o, i, fh_gelu_erf(y * scale_shift[c] + scale_shift[C + c]));
  }
  unsigned rm = 0u;
  fh_track_pack<T>(rm, o);
  fh_range_flush<T>(sat, rm);
  T::store8(out + t * C + cg * 8, o);
}

// ------------------------------------------------------------------------------------------
// Conv1d(Cin -> N, k, stride, no padding) + GELU on NLC 16-bit activations, as ONE GEMM: row t of the A operand
// is the contiguous window x[t*stride .. t*stride + k) x Cin of the input buffer, i.e. A is row-major with
// leading dimension stride*Cin and K = k*Cin (rows overlap; nothing is copied).  W is [N][K] with K ordered
// (tap, channel).  128 x 64 output tile per workgroup, K in steps of 64 through swizzled, double-buffered LDS
// (64-byte half-rows, 16-byte chunk index XOR (row>>1)&3: conflict-free ds_read_b128, as in the decoder);
// the global loads of step s+1 are issued before the MFMAs of step s.  Operands swapped (D = W A^T) so a lane
// holds 4 consecutive output channels of one row.
struct AudGemmArgs {
  const void* A;  // T::elem
  long long lda;  // elements between consecutive rows of A
  const void* W;  // [N][K] T::elem
  const float* bias;  // [N] or nullptr
  void* out;      // [M][ldc] T::elem
  int M, N, K, ldc, act;  // act 1: GELU(erf)
  unsigned long long* sat;  // range counter of the 16-bit stores (fh_range_flush)
};

template <class T>
__global__ __launch_bounds__(256) void aud_gemm_tile_kernel(AudGemmArgs g) {
  constexpr int BM = 128, BN = 64, BK = 64;
  unsigned rm = 0u;  // range tracker (fh_range_flush)
  typedef typename T::elem E;
  typedef typename T::pack8 P8;
  constexpr int EB = T::EB, CB = 8 * EB;  // bytes per element / per pack of 8
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2][(BM + BN) * BK * EB]
  const E* const gA = reinterpret_cast<const E*>(g.A);
  const E* const gW = reinterpret_cast<const E*>(g.W);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  // staging: A tile = 128 rows x 8 chunks (16 B) = 1024 chunks -> 4 per thread; B tile = 64 x 8 = 512 -> 2 per thread
  // LDS image per operand: [half h = chunk>>2][row][4 chunks], chunk' = (chunk&3) ^ ((row>>1)&3)
  auto lds_off = [](int rows, int row, int chunk) { return ((chunk >> 2) * rows + row) * (4 * CB) + ((chunk & 3) ^ ((row >> 1) & 3)) * CB; };
  P8 ra[4], rb[2];
  auto issue = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256, row = e >> 3, ch = e & 7;
      const int m = m0 + row;
      ra[i] = (m < g.M) ? T::load8(gA + (size_t)m * g.lda + k0 + ch * 8) : T::zero8();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * 256, row = e >> 3, ch = e & 7;
      rb[i] = T::load8(gW + (size_t)(n0 + row) * g.K + k0 + ch * 8);
    }
  };
  auto commit = [&](int buf) {
    unsigned char* sA = smem + buf * (BM + BN) * BK * EB;
    unsigned char* sB = sA + BM * BK * EB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256;
      *reinterpret_cast<P8*>(sA + lds_off(BM, e >> 3, e & 7)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * 256;
      *reinterpret_cast<P8*>(sB + lds_off(BN, e >> 3, e & 7)) = rb[i];
    }
  };
  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nsteps = g.K / BK;
  issue(0);
  commit(0);
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    if (s + 1 < nsteps) issue((s + 1) * BK);
    const unsigned char* sA = smem + buf * (BM + BN) * BK * EB;
    const unsigned char* sB = sA + BM * BK * EB;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      P8 a[2], b[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const P8*>(sA + lds_off(BM, w * 32 + i * 16 + r16, kb * 4 + q));
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const P8*>(sB + lds_off(BN, j * 16 + r16, kb * 4 + q));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = T::mfma(b[j], a[i], acc[i][j]);
    }
    if (s + 1 < nsteps) {
      commit(buf ^ 1);  // last read in step s-1; a barrier has passed since
      __syncthreads();
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + w * 32 + i * 16 + r16;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + j * 16 + q * 4;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (g.bias) {
        const float4 bb = *reinterpret_cast<const float4*>(g.bias + n);
        v[0] += bb.x;
        v[1] += bb.y;
        v[2] += bb.z;
        v[3] += bb.w;
      }
      if (g.act) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fh_gelu_erf(v[r]);
      }
      fh_store4<T>(reinterpret_cast<E*>(g.out) + (size_t)m * g.ldc + n, v[0], v[1], v[2], v[3], rm);
    }
  }
  fh_range_flush<T>(g.sat, rm);
}

// ------------------------------------------------------------------------------------------
// linear_interpolation(features, seq_len) (wav2vec2.py:184-197: F.interpolate(mode='linear', align_corners=True)
// along time) fused with the feature projection's LayerNorm (Wav2Vec2FeatureProjection.layer_norm, affine).
// One wave per output frame; writes the packed 16-bit A operand of the projection GEMM (K = C).
template <class T, int NV>
__global__ __launch_bounds__(256) void aud_interp_ln_kernel(const typename T::elem* __restrict__ f, int L, int Tn,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            typename T::elem* __restrict__ out, unsigned long long* sat) {
  constexpr int C = NV * 256;
  unsigned rm = 0u;  // range tracker (fh_range_flush)
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= Tn) return;
  const float scale = (Tn > 1) ? (float)(L - 1) / (float)(Tn - 1) : 0.f;
  const float src = scale * (float)t;
  const int i0 = min((int)src, L - 1), i1 = min(i0 + 1, L - 1);
  const float l1 = src - (float)i0, l0 = 1.f - l1;
  float v[NV][4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 256 + lane * 4;
    const typename T::elem* a = f + (size_t)i0 * C + c;
    const typename T::elem* b = f + (size_t)i1 * C + c;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[i][e] = l0 * T::to_float(a[e]) + l1 * T::to_float(b[e]);
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mu = wave_sum(s) * (1.f / C);
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) s2 += (v[i][e] - mu) * (v[i][e] - mu);
  const float rstd = rsqrtf(wave_sum(s2) * (1.f / C) + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 256 + lane * 4;
    const float4 gm = *reinterpret_cast<const float4*>(gamma + c);
    const float4 bt = *reinterpret_cast<const float4*>(beta + c);
    fh_store4<T>(out + fmt_pack_off(t, c, C / 32), (v[i][0] - mu) * rstd * gm.x + bt.x, (v[i][1] - mu) * rstd * gm.y + bt.y,
                 (v[i][2] - mu) * rstd * gm.z + bt.z, (v[i][3] - mu) * rstd * gm.w + bt.w, rm);
  }
  fh_range_flush<T>(sat, rm);
}

// ------------------------------------------------------------------------------------------
// y = LayerNorm(a + res) * gamma + beta (eps), one wave per row, D = NV*256:
//   the encoder's first LayerNorm (hidden + positional embedding), each layer's post-attention and final
//   LayerNorm (Wav2Vec2EncoderLayer.forward), and the audio projection's LayerNorm + SiLU (FLOAT.py:338-342).
// Outputs (each optional): fp32 row-major (the residual stream), packed 16-bit A operand of the next GEMM
// (KB = D/32), and a second packed copy at column offset `stack_col` of a wider operand (KB = stack_kb): the
// (T, layers*D) stack of hidden states that feeds the audio projection (FLOAT.py:345-352).
struct AudLnArgs {
  const float* a;
  const float* res;  // or nullptr
  const float *gamma, *beta;
  float eps;
  float* out_f32;
  void* out_p16;    // T::elem
  void* out_stack;  // T::elem
  int stack_col, stack_kb;
  int M, silu;
  int keep_sum;  // 1: out_f32 = a + res (the un-normalised residual stream of a pre-LayerNorm encoder), 0: the LayerNorm output
  unsigned long long* sat;  // range counter of the 16-bit stores
};

template <class T, int NV>
__global__ __launch_bounds__(256) void aud_ln_kernel(AudLnArgs g) {
  constexpr int D = NV * 256;
  unsigned rm = 0u;  // range tracker (fh_range_flush)
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= g.M) return;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 256 + lane * 4;
    v[i] = *reinterpret_cast<const float4*>(g.a + (size_t)row * D + c);
    if (g.res) {
      const float4 r = *reinterpret_cast<const float4*>(g.res + (size_t)row * D + c);
      v[i].x += r.x;
      v[i].y += r.y;
      v[i].z += r.z;
      v[i].w += r.w;
    }
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mu = wave_sum(s) * (1.f / D);
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float d0 = v[i].x - mu, d1 = v[i].y - mu, d2 = v[i].z - mu, d3 = v[i].w - mu;
    s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  const float rstd = rsqrtf(wave_sum(s2) * (1.f / D) + g.eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 256 + lane * 4;
    const float4 gm = *reinterpret_cast<const float4*>(g.gamma + c);
    const float4 bt = *reinterpret_cast<const float4*>(g.beta + c);
    float4 y;
    y.x = (v[i].x - mu) * rstd * gm.x + bt.x;
    y.y = (v[i].y - mu) * rstd * gm.y + bt.y;
    y.z = (v[i].z - mu) * rstd * gm.z + bt.z;
    y.w = (v[i].w - mu) * rstd * gm.w + bt.w;
    if (g.silu) {
      y.x = fh_silu(y.x);
      y.y = fh_silu(y.y);
      y.z = fh_silu(y.z);
      y.w = fh_silu(y.w);
    }
    if (g.out_f32) *reinterpret_cast<float4*>(g.out_f32 + (size_t)row * D + c) = g.keep_sum ? v[i] : y;
    typedef typename T::elem E;
    if (g.out_p16) fh_store4<T>(reinterpret_cast<E*>(g.out_p16) + fmt_pack_off(row, c, D / 32), y.x, y.y, y.z, y.w, rm);
    if (g.out_stack) fh_store4<T>(reinterpret_cast<E*>(g.out_stack) + fmt_pack_off(row, g.stack_col + c, g.stack_kb), y.x, y.y, y.z, y.w, rm);
  }
  fh_range_flush<T>(g.sat, rm);
}

// ------------------------------------------------------------------------------------------
// Wav2Vec2PositionalConvEmbedding: grouped Conv1d(D, D, k = 128, padding 64, groups 16) on the time axis, drop
// the last output (even kernel), GELU.  Weight-norm is folded on the host.  Two adjacent groups are merged into
// one block-diagonal 96 x 96 slab per tap so that K per tap is 3 MFMA k-blocks (48 channels per group is not a
// multiple of 32); the zero blocks cost 2x the flops of a 2.4 GFLOP layer.
//   workgroup = (16 output frames, group pair); its 4 waves split the taps, partial sums meet in LDS.
//   A fragments are converted from the fp32 residual stream on the fly; W is [pair][tap][96 out][96 in] 16-bit.
template <class T, int GP /* channels per merged group */>
__global__ __launch_bounds__(256) void aud_posconv_kernel(const float* __restrict__ x, int Tn, int D, const typename T::elem* __restrict__ W,
                                                          const float* __restrict__ bias, int ktaps, int pad, float* __restrict__ out,
                                                          unsigned long long* sat) {
  typedef typename T::pack8 P8;
  constexpr int NT = GP / 16, KBN = GP / 32;
  unsigned rm = 0u;  // range tracker (fh_range_flush)
  __shared__ float red[4][16][GP];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int t0 = blockIdx.x * 16, gp = blockIdx.y;
  f32x4 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int tpw = ktaps / 4;
  for (int kk = w * tpw; kk < (w + 1) * tpw; ++kk) {
    const int t = t0 + r16 + kk - pad;
    const bool ok = t >= 0 && t < Tn;
    const float* xr = x + (size_t)(ok ? t : 0) * D + gp * GP + q * 8;
    const typename T::elem* wr = W + (((size_t)gp * ktaps + kk) * GP + r16) * GP + q * 8;
#pragma unroll
    for (int kb = 0; kb < KBN; ++kb) {
      P8 a = T::zero8();
      if (ok) {
        const float4 f0 = *reinterpret_cast<const float4*>(xr + kb * 32);
        const float4 f1 = *reinterpret_cast<const float4*>(xr + kb * 32 + 4);
        const float fv[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) T::set(a, e, fv[e]);
        fh_track_pack<T>(rm, a);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const P8 b = T::load8(wr + (size_t)j * 16 * GP + kb * 32);
        acc[j] = T::mfma(a, b, acc[j]);  // D[row = frame q*4+reg][col = channel r16]
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][q * 4 + r][j * 16 + r16] = acc[j][r];
  fh_range_flush<T>(sat, rm);
  __syncthreads();
  for (int idx = threadIdx.x; idx < 16 * GP; idx += 256) {
    const int r = idx / GP, c = idx % GP;
    const int t = t0 + r;
    if (t >= Tn) continue;
    const float v = red[0][r][c] + red[1][r][c] + red[2][r][c] + red[3][r][c] + bias[gp * GP + c];
    out[(size_t)t * D + gp * GP + c] = fh_gelu_erf(v);
  }
}

// ------------------------------------------------------------------------------------------
// Wav2Vec2Attention (eager): softmax(q k^T * head_dim^-0.5) v over ALL frames, no mask.  One wave per (query,
// head): lanes own keys (scores) and then head dimensions (weighted sum of V).  Head dim 64.  qkv is row-major
// 16-bit [T][3*D] (q | k | v); the output is the packed A operand of out_proj (K = D).
// Keys go through in tiles of kAudKeyTile with an online softmax (running max m, running sum l, the accumulator rescaled by
// exp(m_old - m_new) when a tile raises the max), so the LDS score row is one tile long whatever the clip length: the
// reference has no length limit (FLOAT.py:190-198) and neither has this kernel (rounds 1-2 kept the whole row in LDS and
// stopped at 3900 / 10 000 frames).  A clip of at most one tile takes the same arithmetic path as before.
constexpr int kAudKeyTile = 2048;
template <class T>
__global__ __launch_bounds__(256) void aud_attn_kernel(const typename T::elem* __restrict__ qkv, int Tn, int D, int heads,
                                                       typename T::elem* __restrict__ out, unsigned long long* sat) {
  typedef typename T::elem E;
  constexpr int HD = 64;
  __shared__ float sm[4 * (HD + kAudKeyTile)];  // per wave: q[64] + p[tile]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int qi = blockIdx.x * 4 + w, h = blockIdx.y;
  float* sq = sm + (size_t)w * (HD + kAudKeyTile);
  float* sp = sq + HD;
  const bool live = qi < Tn;
  const int ld = 3 * D;
  if (live) sq[lane] = T::to_float(qkv[(size_t)qi * ld + h * HD + lane]) * 0.125f;  // head_dim^-0.5 = 1/8
  __syncthreads();
  if (!live) return;
  float m = -INFINITY, l = 0.f, acc = 0.f;
  const E* vbase = qkv + 2 * D + h * HD + lane;
  for (int j0 = 0; j0 < Tn; j0 += kAudKeyTile) {
    const int nt = min(kAudKeyTile, Tn - j0);
    float mx = -INFINITY;
    for (int j = lane; j < nt; j += 64) {
      const E* kp = qkv + (size_t)(j0 + j) * ld + D + h * HD;
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < HD; c += 8) {
        const typename T::pack8 u = T::load8(kp + c);
#pragma unroll
        for (int i = 0; i < 8; ++i) dot += sq[c + i] * T::get(u, i);
      }
      sp[j] = dot;
      mx = fmaxf(mx, dot);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const float mn = fmaxf(m, mx);
    const float alpha = __expf(m - mn);  // first tile: exp(-inf) = 0 on acc = l = 0
    float sum = 0.f;
    for (int j = lane; j < nt; j += 64) {
      const float p = __expf(sp[j] - mn);
      sp[j] = p;
      sum += p;
    }
    l = l * alpha + wave_sum(sum);
    m = mn;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    const E* vp = vbase + (size_t)j0 * ld;
    float a = 0.f;
    int j = 0;
    for (; j + 4 <= nt; j += 4) {
      const float v0 = T::to_float(vp[(size_t)(j + 0) * ld]), v1 = T::to_float(vp[(size_t)(j + 1) * ld]);
      const float v2 = T::to_float(vp[(size_t)(j + 2) * ld]), v3 = T::to_float(vp[(size_t)(j + 3) * ld]);
      a += sp[j] * v0 + sp[j + 1] * v1 + sp[j + 2] * v2 + sp[j + 3] * v3;
    }
    for (; j < nt; ++j) a += sp[j] * T::to_float(vp[(size_t)j * ld]);
    acc = acc * alpha + a;
    __builtin_amdgcn_wave_barrier();  // the next tile overwrites the score row
    __threadfence_block();
  }
  unsigned rm = 0u;
  out[fmt_pack_off(qi, h * HD + lane, D / 32)] = fh_cvt<T>(acc / l, rm);
  fh_range_flush<T>(sat, rm);
}

// ------------------------------------------------------------------------------------------
// The same attention on the matrix pipe (16-bit operand types; round 4).  aud_attn_kernel above gives every query its own
// wave, and each wave streams ALL keys and values of its head from L2: 6000 waves x 128 KB = 770 MB per layer at 500 frames,
// 87 us per launch - 1.04 of the audio encoder's 1.58 ms per 10-s clip, 2.1 of the speech-emotion model's 6.5 ms.  Here a
// workgroup owns 64 queries of one head (16 per wave) and walks the keys in tiles of 64:
//   S^T = K Q^T   MFMA(A = K tile rows [16 keys x 32 dims], B = Q^T): operands straight from global memory, a lane's 16 bytes
//                 are 8 consecutive dims of one key / query; D[key][query] leaves a lane with 4 consecutive keys of ONE query
//                 per 16-key sub-tile, so the softmax statistics of a query are in-lane sums + two cross-group exchanges;
//   O^T += V^T P^T  MFMA(A = V^T [16 dims x 32 keys], B = P^T): the probabilities go from the S^T accumulators to the B operand
//                 without leaving the lane (contraction index e of lane group g = key 16*(e/4) + 4g + e%4 of the 32-key block);
//                 V^T is staged in LDS transposed with the keys in exactly that order (row stride 144 B: conflict-free
//                 ds_read_b128), once per workgroup and tile.
// Online softmax over the tiles as before (running max, accumulators rescaled by exp(m_old - m_new)); keys past Tn get
// probability 0.  Output packed as the A operand of the out-projection.  K / V bytes per layer: 49 MB instead of 770.
template <class T>
__global__ __launch_bounds__(256) void aud_attn_mfma_kernel(const typename T::elem* __restrict__ qkv, int Tn, int D, int heads,
                                                            typename T::elem* __restrict__ out, unsigned long long* sat) {
  static_assert(!T::is32, "16-bit operand types only (the fp32 mode runs aud_attn_kernel)");
  typedef typename T::elem E;
  typedef typename T::pack8 P8;
  constexpr int HD = 64, KT = 64, VSTR = 72;  // V^T row stride in elements (144 B)
  __shared__ __attribute__((aligned(16))) E sV[HD * VSTR];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c16 = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, ld = 3 * D;
  const int q_raw = blockIdx.x * 64 + w * 16 + c16, qi = min(q_raw, Tn - 1);
  const E* const qp = qkv + (size_t)qi * ld + h * HD + g * 8;
  const P8 qB0 = T::load8(qp), qB1 = T::load8(qp + 32);  // B operand: query c16, dims kb*32 + g*8 .. +7
  f32x4 oT[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) oT[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;
  // staging of V: thread -> (key = tid / 4 [+ 0], 16 dims = two 8-packs) of the tile
  const int vkey = tid >> 2, vd0 = (tid & 3) * 16;
  const int vpos = (vkey & 32) + ((vkey >> 2) & 3) * 8 + ((vkey >> 4) & 1) * 4 + (vkey & 3);  // key' of this thread's key
  for (int j0 = 0; j0 < Tn; j0 += KT) {
    // K fragments of the tile (this wave's own copy, from L2): sub-tile st, dims kb*32 + g*8
    P8 kA[4][2];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const E* kp = qkv + (size_t)min(j0 + st * 16 + c16, Tn - 1) * ld + D + h * HD + g * 8;
      kA[st][0] = T::load8(kp);
      kA[st][1] = T::load8(kp + 32);
    }
    const E* vp = qkv + (size_t)min(j0 + vkey, Tn - 1) * ld + 2 * D + h * HD + vd0;
    const P8 v0 = T::load8(vp), v1 = T::load8(vp + 8);
    __syncthreads();  // the previous tile's V^T has been read by every wave
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sV[(vd0 + i) * VSTR + vpos] = reinterpret_cast<const E*>(&v0)[i];
      sV[(vd0 + 8 + i) * VSTR + vpos] = reinterpret_cast<const E*>(&v1)[i];
    }
    __syncthreads();
    f32x4 sT[4];
    float mx = -INFINITY;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
      a = T::mfma(kA[st][0], qB0, a);
      a = T::mfma(kA[st][1], qB1, a);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = j0 + st * 16 + g * 4 + r;
        a[r] = key < Tn ? a[r] * 0.125f : -INFINITY;  // head_dim^-0.5
        mx = fmaxf(mx, a[r]);
      }
      sT[st] = a;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);       // finite: key j0 exists for every query
    const float alpha = __expf(m - mn);  // first tile: exp(-inf) = 0 on zero accumulators
    m = mn;
    float sum = 0.f;
    P8 pB[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float pv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        pv[e] = __expf(sT[kb * 2 + (e >> 2)][e & 3] - mn);
        sum += pv[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) T::set(pB[kb], e, pv[e]);
    }
    l = l * alpha + sum;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 o = oT[dt] * alpha;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const P8 vA = *reinterpret_cast<const P8*>(sV + (dt * 16 + c16) * VSTR + kb * 32 + g * 8);
        o = T::mfma(vA, pB[kb], o);
      }
      oT[dt] = o;
    }
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (q_raw >= Tn) return;
  const float inv = 1.f / l;
  unsigned rm = 0u;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
    fh_store4<T>(out + fmt_pack_off(q_raw, h * HD + dt * 16 + g * 4, D / 32), oT[dt][0] * inv, oT[dt][1] * inv, oT[dt][2] * inv, oT[dt][3] * inv, rm);
  fh_range_flush<T>(sat, rm);
}

// ------------------------------------------------------------------------------------------
// feat_extract_norm = "layer" (wav2vec2-large / the speech-emotion model, Wav2Vec2LayerNormConvLayer): every conv is
// followed by LayerNorm over the CHANNELS of each time step (affine) and GELU.
// Layer 0: Conv1d(1 -> C, k = 10) + bias, LayerNorm, GELU; one wave per time step, 8 channels per lane (C = 512).
template <class T, int KW>
__global__ __launch_bounds__(256) void aud_conv0_ln_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                           int stride, int L, int C, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, typename T::elem* __restrict__ out,
                                                           unsigned long long* sat) {
  const int lane = threadIdx.x & 63;
  unsigned rm = 0u;  // range tracker (fh_range_flush)
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= L) return;
  float xv[KW];
#pragma unroll
  for (int k = 0; k < KW; ++k) xv[k] = x[(size_t)t * stride + k];
  const int per = C / 64;  // channels per lane (<= 8), contiguous
  float y[8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    y[i] = 0.f;
    if (i < per) {
      const int c = lane * per + i;
      float a = bias ? bias[c] : 0.f;
#pragma unroll
      for (int k = 0; k < KW; ++k) a += w[c * KW + k] * xv[k];
      y[i] = a;
      s += a;
    }
  }
  const float mu = wave_sum(s) / (float)C;
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i < per) s2 += (y[i] - mu) * (y[i] - mu);
  const float rstd = rsqrtf(wave_sum(s2) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i < per) {
      const int c = lane * per + i;
      out[(size_t)t * C + c] = fh_cvt<T>(fh_gelu_erf((y[i] - mu) * rstd * gamma[c] + beta[c]), rm);
    }
  fh_range_flush<T>(sat, rm);
}

// Layers 1..n-1: the conv GEMM stores conv + bias (16-bit, row-major); this pass normalises each row over its C channels
// (affine) and applies GELU in place.  One wave per row.
template <class T, int NV>
__global__ __launch_bounds__(256) void aud_rowln_gelu_kernel(typename T::elem* __restrict__ x, int M, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, unsigned long long* sat) {
  constexpr int C = NV * 256;
  unsigned rm = 0u;  // range tracker (fh_range_flush)
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float v[NV][4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const typename T::elem* a = x + (size_t)row * C + i * 256 + lane * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[i][e] = T::to_float(a[e]);
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mu = wave_sum(s) * (1.f / C);
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) s2 += (v[i][e] - mu) * (v[i][e] - mu);
  const float rstd = rsqrtf(wave_sum(s2) * (1.f / C) + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 256 + lane * 4;
    const float4 gm = *reinterpret_cast<const float4*>(gamma + c);
    const float4 bt = *reinterpret_cast<const float4*>(beta + c);
    fh_store4<T>(x + (size_t)row * C + c, fh_gelu_erf((v[i][0] - mu) * rstd * gm.x + bt.x), fh_gelu_erf((v[i][1] - mu) * rstd * gm.y + bt.y),
                 fh_gelu_erf((v[i][2] - mu) * rstd * gm.z + bt.z), fh_gelu_erf((v[i][3] - mu) * rstd * gm.w + bt.w), rm);
  }
  fh_range_flush<T>(sat, rm);
}

// ------------------------------------------------------------------------------------------
// Classification head of the speech-emotion model (wav2vec2_ser.py:23-38,58-75,94-96; FLOAT.py:396-401):
// mean over time -> dense -> tanh -> out_proj -> softmax.  fp32; one workgroup.
__global__ __launch_bounds__(256) void aud_meanpool_kernel(const float* __restrict__ h, int Tn, int D, float* __restrict__ out) {
  // a workgroup owns 64 channels: its 4 waves take every 4th frame (coalesced 256-byte rows), partial sums meet in LDS in a
  // fixed order (one thread per channel walking all Tn frames took 116 us at 500 frames)
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (c < D)
    for (int t = w; t < Tn; t += 4) s += h[(size_t)t * D + c];
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && c < D) out[c] = ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) / (float)Tn;
}

// y = act(W x + b), one wave per output; act 1 = tanh
__global__ __launch_bounds__(256) void aud_dense_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ b,
                                                        float* __restrict__ y, int N, int K, int act) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += x[k] * W[(size_t)n * K + k];
  s = wave_sum(s);
  if (lane == 0) {
    s += b[n];
    y[n] = act ? tanhf(s) : s;
  }
}

__global__ void aud_softmax_kernel(const float* __restrict__ x, float* __restrict__ y, int n) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float m = -INFINITY;
  for (int i = 0; i < n; ++i) m = fmaxf(m, x[i]);
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += expf(x[i] - m);
  for (int i = 0; i < n; ++i) y[i] = expf(x[i] - m) / s;
}
