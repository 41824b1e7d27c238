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

// Weights of a LATER GEMM of the chain, touched one dword per 128-byte line by the workgroups of a kernel that runs
// before it, so that the GEMM finds them in its XCD's L2 (clean L2 lines survive a kernel boundary; r01: LayerNorm touching
// qkv / fc1 took the chain from 91.3 to 88.4 ms).  Mirrors fmt_gemm_kernel's block decode: column block bx (nt 16-column
// tiles) of K slice ks runs on XCD (bx % P) + ks * P with P = 8 / ksplit; a tile's K slice is one contiguous run.
struct TouchSpec {
  const char* W;         // nullptr: nothing to touch
  unsigned run_shift;    // log2 of the 128-byte lines of one run (one 16-column tile's K slice)
  unsigned nt_shift;     // log2 of the 16-column tiles per column block
  unsigned p_shift;      // log2 of P
  unsigned tile_bytes;   // bytes of one 16-column tile over all of K
  unsigned total;        // lines per XCD = column blocks per XCD * tiles * run lines
};

// line l (of t.total) of XCD xcd's share; every size is a power of two, so this is shifts and masks
__device__ __forceinline__ unsigned fmt_touch_line(const TouchSpec& t, unsigned xcd, unsigned l) {
  if (l >= t.total) return 0u;
  const unsigned r = l >> t.run_shift, off = l & ((1u << t.run_shift) - 1u);
  const unsigned cb = r >> t.nt_shift, jn = r & ((1u << t.nt_shift) - 1u);
  const unsigned ks = xcd >> t.p_shift, bx = (cb << t.p_shift) + (xcd & ((1u << t.p_shift) - 1u));
  return *reinterpret_cast<const unsigned*>(t.W + (size_t)((bx << t.nt_shift) + jn) * t.tile_bytes +
                                            ((size_t)((ks << t.run_shift) + off) << 7));
}
// `n` lines per lane, lanes of the XCD's workgroups interleaved; the dwords must stay live until fmt_touch_retire
template <int N>
__device__ __forceinline__ void fmt_touch(const TouchSpec& t, unsigned xcd, unsigned first, unsigned stride, unsigned (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = fmt_touch_line(t, xcd, first + (unsigned)i * stride);
}
// keeps the touched dwords' registers reserved until the loads have landed, at no other cost
template <int N>
__device__ __forceinline__ void fmt_touch_retire(const unsigned (&v)[N]) {
  unsigned o = 0u;
#pragma unroll
  for (int i = 0; i < N; ++i) o |= v[i];
  asm volatile("" ::"v"(o));
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
  int nclip;      // independent clips stacked along the rows (row = (clip * bc + cfg row) * ntok + token); 0 / 1 = one clip
  // EPI_CFG
  float a_cfg, r_cfg, e_cfg, dt;
  float* vout;   // (ntok, N) combined velocity (float_fmt_eval) or nullptr
  float* xcur;   // (ntok - n_prev, N) Euler state or nullptr
  u16* xin16;    // next evaluation's x_embedder input, packed with KB = ldx
  int ldx;
  // EPI_PARTIAL: K is cut into ksplit slices, one per workgroup; slice ks writes out_f32 + ks * slab_stride
  int ksplit;
  size_t slab_stride;
  TouchSpec touch;  // weights of a later GEMM to pull into L2 (W == nullptr: none)
  // fmt_gemm_wide_kernel only: `zcount` independent row batches (the Euler steps of a window) against the same weights in one
  // launch; batch z reads A + z * a_zstride (elements) and writes out_f32 + z * o_zstride (elements).  0 / 1 = a single batch.
  int zcount, zgroup;  // zgroup: column blocks of an XCD that share their activation tiles through L2 (block decode)
  size_t a_zstride, o_zstride;
  // EPI_CFG: rows are token-blocked, row = (i / 16) * (bc * 16) + b * 16 + i % 16 for token i, CFG row b, so a workgroup
  // with bc row tiles holds every CFG row of its 16 tokens (else: row = b * ntok + i and the workgroup holds all rows)
  int tokblk;
  unsigned long long* sat;  // range counter of the launch's 16-bit stores (fh_range_flush) or nullptr
};

