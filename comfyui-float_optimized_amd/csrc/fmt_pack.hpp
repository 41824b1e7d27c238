// Operand layout and argument block of the FMT's weight-streaming GEMM (fmt_kernels.hpp), shared with the
// operators that feed it packed activations (aud_kernels.hpp).
#pragma once
#include "common.hpp"

enum {
  EPI_F32 = 0,       // out_f32 = acc + bias                                   (row-major fp32)
  EPI_T16 = 1,       // out16   = T(acc + bias)                                (row-major, feeds attention)
  EPI_SILU_P16 = 2,  // out16   = T(silu(acc + bias))                          (packed, feeds a GEMM)
  EPI_GELU_P16 = 3,  // out16   = T(gelu_tanh(acc + bias))                     (packed, feeds a GEMM)
  EPI_GATE_RES = 4,  // out_f32 += gate * (acc + bias)                         (FMT.py:174-175)
  EPI_XEMBED = 5,    // out_f32[b*ntok + r] = acc + bias + pos[r], b < bc       (FMT.py:319-320)
  EPI_CFG = 6,       // CFG combine (+ Euler update) on the final linear        (FMT.py:375-399)
  EPI_PARTIAL = 7,   // slab[ks][row][n] = acc  (split-K slice ks, no bias): summed, gated and added to the
                     // residual stream by the LayerNorm kernel that follows (fmt_lnmod_kernel<.., KS>)
  EPI_GELUERF_P16 = 8  // out16 = T(gelu_erf(acc + bias)) packed: wav2vec2's feed-forward (HF ACT2FN["gelu"])
};

// element offset of (row, k) in a packed operand with KB = K/32 k-blocks
__host__ __device__ __forceinline__ size_t fmt_pack_off(int row, int k, int KB) {
  return ((size_t)((row >> 4) * KB + (k >> 5)) * 64 + (row & 15) + 16 * ((k >> 3) & 3)) * 8 + (k & 7);
}

struct GemmArgs {
  const u16* A;   // packed [row tiles][KB][64][8]; pad rows/columns are zero
  const u16* W;   // packed [N/16][KB][64][8]
  const float* bias;
  int K, M, N;    // K padded to a multiple of 128
  int mblk;       // number of row blocks (grid = N/BN * mblk workgroups)
  float* out_f32;
  int ldo;
  u16* out16;
  int ldo16;      // row-major leading dim (EPI_T16) or KB of the consumer (packed epilogues)
  const float* gate;
  int ldg;
  const float* pos;
  int bc, ntok, n_prev;
  // EPI_CFG
  float a_cfg, r_cfg, e_cfg, dt;
  float* vout;   // (ntok, N) combined velocity (float_fmt_eval) or nullptr
  float* xcur;   // (ntok - n_prev, N) Euler state or nullptr
  u16* xin16;    // next evaluation's x_embedder input, packed with KB = ldx
  int ldx;
  // EPI_PARTIAL: K is cut into ksplit slices, one per workgroup; slice ks writes out_f32 + ks * slab_stride
  int ksplit;
  size_t slab_stride;
};

