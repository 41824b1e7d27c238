// HIP kernels of the Flow-Matching-Transformer evaluation (reference FMT.py:277-401).
//
// Shapes: M = Bc * n_tok rows (Bc = 1/3/4 CFG rows x 60 tokens), so every linear layer is a
// weight-streaming GEMM with a short M.  Both operands live in HBM in MFMA-FRAGMENT-MAJOR order:
//   packed[tile16][kb][lane(64)][8]   tile16 = row/16 (A) or n/16 (W), kb = k/32,
//                                     lane = (row%16) + 16*((k/8)%4), 8 consecutive k per lane
// so the 16-byte-per-lane operand of v_mfma_f32_16x16x32 for one (tile, kb) is ONE contiguous 1 KiB
// wave load (8 full 128-B lines) and consecutive k-steps are consecutive KiBs - no LDS staging, no
// strided rows hitting one L2 channel.  A workgroup owns MTW row tiles x NT column tiles; its 4 waves
// split K, partial sums meet in LDS, and the epilogue (bias / SiLU / GELU / gate*residual / CFG+Euler)
// is applied once per output element in a fixed order -> bitwise reproducible, no atomics.  Weights
// are read from HBM once (non-temporal); workgroups that share a weight column block differ only in
// blockIdx.y, i.e. by a multiple of gridDim.x (a multiple of 8), so they sit on one XCD's L2.
#pragma once
#include "common.hpp"
#include "fmt_pack.hpp"

// The kernel's body as a device function: `bid` of `nblk` workgroups (blockIdx.x of gridDim.x for the stand-alone launch, the
// stage's virtual block for fmt_mega_kernel).  COH: the activations (A operand, residual stream) were produced earlier in the
// same kernel -> coherent loads (fh_load16<true>), and every activation store writes through.
template <class T, int MTW, int NT, int NW, int EPI, bool COH>
__device__ __forceinline__ void fmt_gemm_body(const GemmArgs& g, const unsigned bid, const unsigned nblk) {
  constexpr int BN = NT * 16;
  constexpr int ROWS = MTW * 16;
  constexpr bool WT = NW >= 8 || COH;  // the single-clip tilings (stacked clips split K over 4 waves): see common.hpp, FMT_WT
  // A operand: plain (L2-cached) loads also inside the persistent kernel.  There the operand lives in a buffer that is written
  // exactly ONCE per launch (write-through) before its first read - no cache of any level can hold an older copy of it - and
  // it is re-read by every column block of the XCD: as agent-scope (sc1) loads those re-reads all crossed the fabric
  // (415 vs 320 us per evaluation).  The residual stream and the split-K slabs (read once, by one workgroup) stay coherent.
  auto ldA = [](const typename T::elem* p) -> typename T::pack8 { return T::load8(p); };
  constexpr int NTHR = NW * 64;
  // k-steps of operands in flight per wave.  With 4 k-blocks per wave (K = 1024 over 8 waves) PF = 4 puts the whole
  // K slice in flight at once: one memory round trip instead of two for the 48x64 tilings (7 fragments per k-step,
  // 112 operand registers + 48 accumulators still fit two waves per SIMD)
  constexpr int PF = T::is32 ? 1 : ((MTW + NT <= 7) ? 4 : ((MTW + NT <= 8) ? 3 : 2));
  typedef typename T::elem E;
  typedef typename T::pack8 P8;
  extern __shared__ __attribute__((aligned(16))) float red[];  // [NW][ROWS][BN]
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  // 1-D grid of (column blocks) x (row blocks).  Row blocks of one column block read the same
  // weights: give them consecutive slots on ONE XCD (ids congruent mod 8 share an XCD's L2).
  int bx, by, ks = 0;
  {
    const int nbn = g.N / BN, id = (int)bid;
    const int nbx = (EPI == EPI_PARTIAL) ? nbn * g.ksplit : nbn;  // (column block, K slice) pairs
    if ((nbx & 7) == 0) {
      const int slot = id >> 3;
      by = slot % g.mblk;
      bx = (slot / g.mblk) * 8 + (id & 7);
    } else {
      bx = id % nbx;
      by = id / nbx;
    }
    if constexpr (EPI == EPI_PARTIAL) {
      if ((nbx & 7) == 0 && (8 % g.ksplit) == 0) {
        // K slice <-> XCD affinity: the 8/ksplit XCDs that own slice ks fetch only that slice of the
        // activations (r01 PMC: with slices spread over all XCDs fc2 fetched its 1.5 MB operand 8 times,
        // 12.6 MB against 8.4 MB of weights)
        const int P = 8 / g.ksplit, x = bx & 7;
        ks = x / P;
        bx = (bx >> 3) * P + (x % P);
      } else {
        ks = bx / nbn;
        bx = bx % nbn;
      }
    }
  }
  const int nb0 = bx * NT;
  const int mt0 = by * MTW;
  const int n0 = nb0 * 16, m0 = mt0 * 16;

  f32x4 acc[MTW][NT];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int KB = g.K >> 5;
  const int KBs = (EPI == EPI_PARTIAL) ? KB / g.ksplit : KB;  // k-blocks of this workgroup's K slice
  const int KBw = KBs / NW;                                   // k-blocks per wave
  const E* Ap = reinterpret_cast<const E*>(g.A) + ((size_t)mt0 * KB + ks * KBs + w * KBw) * 512 + lane * 8;
  const E* Wp = reinterpret_cast<const E*>(g.W) + ((size_t)nb0 * KB + ks * KBs + w * KBw) * 512 + lane * 8;
  const size_t tstride = (size_t)KB * 512;
  // EPI_GATE_RES with one epilogue item per thread: fetch residual and gate now, not after the K loop
  constexpr bool kEarlyRes = (EPI == EPI_GATE_RES) && (MTW * 16 * (NT * 2) <= NW * 64);
  float4 res_pf[2], gate_pf[2];
  if constexpr (kEarlyRes) {
    const int r = threadIdx.x / (NT * 2), cg = threadIdx.x % (NT * 2);
    const int row = m0 + r, nb = n0 + cg * 8;
    if (r < ROWS && row < g.M) {
      const float* o = g.out_f32 + (size_t)row * g.ldo + nb;
      const float* gt = g.gate + (size_t)row * g.ldg + nb;
      res_pf[0] = fh_load_f4<COH>(o);
      res_pf[1] = fh_load_f4<COH>(o + 4);
      gate_pf[0] = fh_load_f4_stream(gt);
      gate_pf[1] = fh_load_f4_stream(gt + 4);
    }
  }

  P8 a[PF][MTW], b[PF][NT];
#pragma unroll
  for (int p = 0; p < PF; ++p) {
    if (p < KBw) {
#pragma unroll
      for (int j = 0; j < NT; ++j) b[p][j] = T::load8_nt(Wp + j * tstride + (size_t)p * 512);
#pragma unroll
      for (int i = 0; i < MTW; ++i) a[p][i] = ldA(Ap + i * tstride + (size_t)p * 512);
    }
  }
  // only the epilogue kinds whose launches are given a TouchSpec carry the code (2 registers, a branch)
  constexpr bool kTouch = (EPI == EPI_GELU_P16 || EPI == EPI_T16 || EPI == EPI_GATE_RES || EPI == EPI_PARTIAL);
  unsigned touched[2] = {0u, 0u};
  if constexpr (kTouch) {
    if (g.touch.W) fmt_touch(g.touch, bid & 7, (bid >> 3) * NTHR + threadIdx.x, (nblk >> 3) * NTHR, touched);
  }
  for (int kb0 = 0; kb0 < KBw; kb0 += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int kb = kb0 + p;
      if (kb < KBw) {
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = T::mfma(a[p][i], b[p][j], acc[i][j]);
        if (kb + PF < KBw) {
#pragma unroll
          for (int j = 0; j < NT; ++j)
            b[p][j] = T::load8_nt(Wp + j * tstride + (size_t)(kb + PF) * 512);
#pragma unroll
          for (int i = 0; i < MTW; ++i) a[p][i] = ldA(Ap + i * tstride + (size_t)(kb + PF) * 512);
        }
      }
    }
  }

  if constexpr (kTouch) {
    if (g.touch.W) fmt_touch_retire(touched);
  }
  // C/D map of mfma_f32_16x16x32: col = lane & 15, row = (lane >> 4) * 4 + reg
  float* my = red + w * (ROWS * BN);
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) my[(i * 16 + q * 4 + r) * BN + j * 16 + r16] = acc[i][j][r];
  __syncthreads();

  constexpr int S = ROWS * BN;
  constexpr int CG = BN / 8;  // 8-column groups per row
  unsigned rm = 0u;           // range tracker of this thread's 16-bit stores (fh_range_flush)
  auto slab_sum = [&](int o) {
    float a = red[o];
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) a += red[ww * S + o];
    return a;
  };
  if constexpr (EPI == EPI_CFG) {
    // rows of the tile: b * ntok + i (one clip, all rows in the workgroup), or b * 16 + i % 16 of (token block, clip) pair `by`
    // (g.tokblk: by = token block * nclip + clip).  Combine the CFG rows of token i, then (optionally) Euler.
    const int nclip = g.nclip > 1 ? g.nclip : 1;
    const int tb = g.tokblk ? by / nclip : 0, q = g.tokblk ? by - tb * nclip : 0;
    const int ni = g.tokblk ? 16 : g.ntok, i0 = tb * 16, rstride = g.tokblk ? 16 : g.ntok;
    float* const vout_q = g.vout ? g.vout + (size_t)q * g.ntok * g.N : nullptr;
    float* const xcur_q = g.xcur ? g.xcur + (size_t)q * (g.ntok - g.n_prev) * g.N : nullptr;
    for (int idx = threadIdx.x; idx < ni * BN; idx += NTHR) {
      const int il = idx / BN, i = i0 + il, c = idx % BN, n = n0 + c;
      if (i >= g.ntok) break;
      const float bias = g.bias[n];
      float v[4];
#pragma unroll
      for (int b2 = 0; b2 < 4; ++b2) {
        if (b2 < g.bc) {
          const int o = (b2 * rstride + il) * BN + c;
          v[b2] = slab_sum(o) + bias;
        } else {
          v[b2] = 0.f;
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
      if (vout_q) vout_q[(size_t)i * g.N + n] = out;
      if (xcur_q && i >= g.n_prev) {
        const size_t xo = (size_t)(i - g.n_prev) * g.N + n;
        const float xn = xcur_q[xo] + g.dt * out;  // x_{k+1} = x_k + dt * v  (fixed-grid Euler)
        xcur_q[xo] = xn;
        reinterpret_cast<E*>(g.xin16)[fmt_pack_off(q * g.ntok + i, n, g.ldx)] = fh_cvt<T>(xn, rm);
      }
    }
  } else {
    for (int idx = threadIdx.x; idx < ROWS * CG; idx += NTHR) {
      const int r = idx / CG, cg = idx % CG;
      const int row = m0 + r, nb = n0 + cg * 8;
      if (row >= g.M) continue;
      float v[8];
      {
        const float* p0 = red + r * BN + cg * 8;
#pragma unroll
        for (int e = 0; e < 8; e += 4) {
          float4 a4 = *reinterpret_cast<const float4*>(p0 + e);
#pragma unroll
          for (int ww = 1; ww < NW; ++ww) {
            const float4 t4 = *reinterpret_cast<const float4*>(p0 + ww * S + e);
            a4.x += t4.x;
            a4.y += t4.y;
            a4.z += t4.z;
            a4.w += t4.w;
          }
          float4 bb = float4{0.f, 0.f, 0.f, 0.f};
          if constexpr (EPI != EPI_PARTIAL) bb = *reinterpret_cast<const float4*>(g.bias + nb + e);
          v[e + 0] = a4.x + bb.x;
          v[e + 1] = a4.y + bb.y;
          v[e + 2] = a4.z + bb.z;
          v[e + 3] = a4.w + bb.w;
        }
      }
      if constexpr (EPI == EPI_F32 || EPI == EPI_PARTIAL) {
        float* o = g.out_f32 + (size_t)ks * g.slab_stride + (size_t)row * g.ldo + nb;
        fh_store_f4_wt<fh_site<COH>(4), WT>(o, float4{v[0], v[1], v[2], v[3]});
        fh_store_f4_wt<fh_site<COH>(4), WT>(o + 4, float4{v[4], v[5], v[6], v[7]});
      } else if constexpr (EPI == EPI_T16 || EPI == EPI_SILU_P16 || EPI == EPI_GELU_P16 || EPI == EPI_GELUERF_P16) {
        P8 u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float x = v[i];
          if constexpr (EPI == EPI_SILU_P16) x = fh_silu(x);
          if constexpr (EPI == EPI_GELU_P16) x = fh_gelu_tanh(x);
          if constexpr (EPI == EPI_GELUERF_P16) x = fh_gelu_erf(x);
          T::set(u, i, x);
        }
        fh_track_pack<T>(rm, u);
        E* const o16 = reinterpret_cast<E*>(g.out16);
        if constexpr (EPI == EPI_T16) T::template store8_wt<fh_site<COH>(8), WT>(o16 + (size_t)row * g.ldo16 + nb, u);
        else T::template store8_wt<fh_site<COH>(8), WT>(o16 + fmt_pack_off(row, nb, g.ldo16), u);
      } else if constexpr (EPI == EPI_GATE_RES) {
        float* o = g.out_f32 + (size_t)row * g.ldo + nb;
        const float* gt = g.gate + (size_t)row * g.ldg + nb;
#pragma unroll
        for (int e = 0; e < 8; e += 4) {
          float4 x, gg;
          if constexpr (kEarlyRes) {
            x = res_pf[e / 4];
            gg = gate_pf[e / 4];
          } else {
            x = fh_load_f4<COH>(o + e);
            gg = fh_load_f4_stream(gt + e);
          }
          x.x += gg.x * v[e + 0];
          x.y += gg.y * v[e + 1];
          x.z += gg.z * v[e + 2];
          x.w += gg.w * v[e + 3];
          fh_store_f4_wt<fh_site<COH>(16), WT>(o + e, x);
        }
      } else if constexpr (EPI == EPI_XEMBED) {
        // input row = clip * ntok + token (the CFG rows of a clip share x); output rows (clip * bc + b2) * ntok + token
        const int q = row / g.ntok, tok = row - q * g.ntok;
        const float* ps = g.pos + (size_t)tok * g.N + nb;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += ps[i];
        for (int b2 = 0; b2 < g.bc; ++b2) {
          float* o = g.out_f32 + (size_t)((q * g.bc + b2) * g.ntok + tok) * g.ldo + nb;
          fh_store_f4_wt<fh_site<COH>(32), WT>(o, float4{v[0], v[1], v[2], v[3]});
          fh_store_f4_wt<fh_site<COH>(32), WT>(o + 4, float4{v[4], v[5], v[6], v[7]});
        }
      }
    }
  }
  if constexpr (EPI == EPI_CFG || EPI == EPI_T16 || EPI == EPI_SILU_P16 || EPI == EPI_GELU_P16 || EPI == EPI_GELUERF_P16) fh_range_flush<T>(g.sat, rm);
}

template <class T, int MTW, int NT, int NW, int EPI>
__global__ __launch_bounds__(NW * 64) void fmt_gemm_kernel(GemmArgs g) {
  fmt_gemm_body<T, MTW, NT, NW, EPI, false>(g, blockIdx.x, gridDim.x);
}

// Wide-N variant for the fused adaLN projection (N = depth*6D + 2D = 51 200, K = D): here re-reading
// the activations per 32-column slab dominates the L2->CU traffic (r01 PMC: 1.8 TB/s of HBM+write
// traffic, waves 85 % stalled), so A goes through LDS ONCE per workgroup and serves 128 columns:
//   * the fragment-major HBM image of A is exactly the LDS image a wave wants (1 KiB per fragment,
//     lane-linear: conflict-free ds_read_b128); chunks of 4 k-blocks are double-buffered through
//     registers so the loads of chunk c+1 fly while chunk c is multiplied;
//   * the 4 waves own 32 columns each for the whole K (no K split, no LDS reduction); weights stream
//     straight to registers, non-temporal;
//   * operands are swapped (D = W_tile * A_tile^T) so a lane holds 4 consecutive columns of one row and
//     the fp32 result is stored as float4.
// NWV = 8: two waves per SIMD - waves 0..3 own the upper half of the row tiles, waves 4..7 the lower half, both halves the
// same four 32-column slices (each weight fragment is then loaded by two waves, from L2; the LDS reads per MFMA do not change).
// (Round 2 also ran this body with the step chain's epilogues for stacked clips - 96 / 192-row tiles, EPI_T16 / EPI_GELU_P16 /
// EPI_PARTIAL - and it was slower than the 48 x 64 tiling at 360-720 rows: 167 / 194 ms per batch of 2 / 4 clips against
// 118 / 181; removed in round 3, DESIGN.md "Negative results".)
template <class T, int MTW, int KCH /* k-blocks (of 32) per LDS chunk */, int NWV = 4>
__global__ __launch_bounds__(NWV * 64) void fmt_gemm_wide_kernel(GemmArgs g) {
  constexpr int EPI = EPI_F32;
  constexpr int NF = MTW * KCH;            // 1-KiB A fragments per chunk
  constexpr int NFW = (NF + NWV - 1) / NWV;  // fragments staged by one wave
  constexpr int WR = NWV / 4, MTR = MTW / WR;  // row halves, row tiles per wave
  static_assert(MTW % WR == 0, "row tiles must split evenly over the wave rows");
  extern __shared__ __attribute__((aligned(16))) unsigned char sA[];  // [2][MTW][KCH][1 KiB]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wc = w & 3, wr = w >> 2;
  const int r16 = lane & 15, q = lane >> 4;
  const int nbn = g.N >> 7;
  const int nbx = (EPI == EPI_PARTIAL) ? nbn * g.ksplit : nbn;  // (column block, K slice) pairs
  int bx, by, bz = 0, ks = 0;
  {
    const int id = blockIdx.x;
    const int nz = g.zcount > 1 ? g.zcount : 1;
    if (nz > 1 && (nbx & 7) == 0 && g.zgroup > 1) {
      // Batched over the Euler steps of a window (the hoisted adaLN projection): ids congruent mod 8 share an XCD's L2.
      // An XCD walks groups of `zgroup` column blocks; inside a group the step runs slowest and the column blocks fastest, so
      // the ~32 workgroups in flight on the XCD are (32 / zgroup) steps x zgroup column blocks: the group's weight slabs
      // (262 KB each) stay in L2 for the whole group and every activation tile (393 KB) is fetched from the Infinity Cache
      // once per group instead of once per column block.  The last group of an XCD may be narrower.
      const int slot = id >> 3, G = g.zgroup, ncx = nbx >> 3, rows = g.mblk * nz;
      const int full = ncx / G, per_grp = G * rows;
      int grp = slot / per_grp, rem = slot - grp * per_grp, gw = G;
      if (grp >= full) {
        grp = full;
        rem = slot - full * per_grp;
        gw = ncx - full * G;
      }
      const int zy = rem / gw;
      bz = zy / g.mblk;
      by = zy - bz * g.mblk;
      bx = (grp * G + (rem - zy * gw)) * 8 + (id & 7);
    } else if ((nbx & 7) == 0) {
      const int slot = id >> 3, zy = slot % (g.mblk * nz);
      bz = zy / g.mblk;
      by = zy - bz * g.mblk;
      bx = (slot / (g.mblk * nz)) * 8 + (id & 7);
    } else {
      bx = id % nbx;
      const int zy = id / nbx;
      bz = zy / g.mblk;
      by = zy - bz * g.mblk;
    }
    if constexpr (EPI == EPI_PARTIAL) {
      if ((nbx & 7) == 0 && (8 % g.ksplit) == 0) {
        const int P = 8 / g.ksplit, x = bx & 7;
        ks = x / P;
        bx = (bx >> 3) * P + (x % P);
      } else {
        ks = bx / nbn;
        bx = bx % nbn;
      }
    }
  }
  const int mt0 = by * MTW;
  const int nb0 = bx * 8 + wc * 2;  // this wave's two 16-column tiles
  const int KB = g.K >> 5;
  const int KBs = (EPI == EPI_PARTIAL) ? KB / g.ksplit : KB;  // k-blocks of this workgroup's K slice
  const int nchunk = KBs / KCH;
  const size_t tstride = (size_t)KB * 512;
  const u16* Wp = g.W + (size_t)nb0 * tstride + (size_t)ks * KBs * 512 + lane * 8;
  const u16* Ag = g.A + (size_t)bz * g.a_zstride + (size_t)mt0 * tstride + (size_t)ks * KBs * 512 + lane * 8;
  float* const outz = g.out_f32 + (size_t)bz * g.o_zstride + (size_t)ks * g.slab_stride;

  f32x4 acc[MTR][2];
#pragma unroll
  for (int i = 0; i < MTR; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Register-staged software pipeline (all ordinary loads, so hipcc's counted vmcnt waits keep the next
  // chunk in flight): while chunk c is multiplied out of LDS buffer c&1 and the b registers, the A
  // fragments and the weights of chunk c+1 are already on their way.
  u32x4 an[NFW], bn[KCH][2], bc[KCH][2];
  auto issue = [&](int c) {
    const int kc = c * KCH;
#pragma unroll
    for (int f = 0; f < NFW; ++f) {
      const int fi = f * NWV + w;
      if (fi < NF) {
        const int i = fi / KCH, kk = fi % KCH;
        an[f] = *reinterpret_cast<const u32x4*>(Ag + i * tstride + (size_t)(kc + kk) * 512);
      }
    }
#pragma unroll
    for (int kk = 0; kk < KCH; ++kk) {
      bn[kk][0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(Wp + (size_t)(kc + kk) * 512));
      bn[kk][1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(Wp + tstride + (size_t)(kc + kk) * 512));
    }
  };
  auto commit = [&](int buf) {  // registers -> LDS buffer `buf`, weights -> current set
#pragma unroll
    for (int f = 0; f < NFW; ++f) {
      const int fi = f * NWV + w;
      if (fi < NF) *reinterpret_cast<u32x4*>(sA + (buf * NF + fi) * 1024 + lane * 16) = an[f];
    }
#pragma unroll
    for (int kk = 0; kk < KCH; ++kk) {
      bc[kk][0] = bn[kk][0];
      bc[kk][1] = bn[kk][1];
    }
  };
  issue(0);
  commit(0);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunk) issue(c + 1);
#pragma unroll
    for (int kk = 0; kk < KCH; ++kk) {
#pragma unroll
      for (int i = 0; i < MTR; ++i) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(sA + (buf * NF + (wr * MTR + i) * KCH + kk) * 1024 + lane * 16);
        acc[i][0] = T::mfma(bc[kk][0], a, acc[i][0]);
        acc[i][1] = T::mfma(bc[kk][1], a, acc[i][1]);
      }
    }
    if (c + 1 < nchunk) {
      commit(buf ^ 1);  // the other buffer was last read in iteration c-1, and a barrier has passed since
      __syncthreads();
    }
  }
  // D[n = q*4 + reg][m = r16]: lane -> row m0 + i*16 + r16, columns n0 + j*16 + q*4 .. +3
#pragma unroll
  for (int i = 0; i < MTR; ++i) {
    const int row = (mt0 + wr * MTR + i) * 16 + r16;
    if (row >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = (nb0 + j) * 16 + q * 4;
      float4 o = float4{acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if constexpr (EPI != EPI_PARTIAL) {
        const float4 bb = *reinterpret_cast<const float4*>(g.bias + n);
        o.x += bb.x;
        o.y += bb.y;
        o.z += bb.z;
        o.w += bb.w;
      }
      if constexpr (EPI == EPI_F32 || EPI == EPI_PARTIAL) {
        *reinterpret_cast<float4*>(outz + (size_t)row * g.ldo + n) = o;
      } else if constexpr (EPI == EPI_T16) {
        T::store4(g.out16 + (size_t)row * g.ldo16 + n, o.x, o.y, o.z, o.w);
      } else {
        static_assert(EPI == EPI_GELU_P16, "epilogue not built for the wide GEMM");
        T::store4(g.out16 + fmt_pack_off(row, n, g.ldo16), fh_gelu_tanh(o.x), fh_gelu_tanh(o.y), fh_gelu_tanh(o.z), fh_gelu_tanh(o.w));
      }
    }
  }
}

// The hoisted adaLN projection on a 192 x 320 tile, 8 waves, both operands by LDS-DMA (fmt_gemm_dma_kernel).
// fmt_gemm_wide_kernel above tops out near 680 TFLOP/s: four waves with 2 column tiles each read one LDS fragment per two
// MFMAs (LDS-read-bound), stage through registers, and need 2-3 co-resident workgroups to hide their own barriers.  Here:
//   * the tile is 12 row tiles x 20 column tiles (51 200 = 160 x 320), waves laid out 2 x 4, each 6 x 5 tiles: 11 fragment
//     reads per 30 MFMAs; 12 + 20 = 32 fragments of 1 KiB per k-block, i.e. exactly 4 LDS-DMA instructions per wave and stage
//     (the fragment-major HBM images are lane-linear, so the LDS image needs no swizzle and every ds_read_b128 is conflict-free);
//   * a ring of 4 stages (128 KiB), three k-blocks in flight behind a COUNTED vmcnt and ONE raw s_barrier per k-block; the
//     fragments of k-block s+1 are read into a second register set while the MFMAs of k-block s run (LDS reads by inline asm:
//     hipcc would otherwise drain the DMA queue before every ds_read).
// Hazards: stage s+4 overwrites the buffer of stage s, whose fragments every wave has in registers (lgkmcnt(0)) before it
// arrives at the barrier of step s; a stage is read one barrier after the wait that retires this wave's share of it.
template <int OFF>
__device__ __forceinline__ u32x4 fh_ds_read128(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void fh_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <class T, int NWC /* wave columns: 4 -> 8 waves, 320-column tile (2 -> 4 waves, 160 columns) */, int NS /* ring stages */,
          int STG /* 1: the two wave rows run half a step apart (see DMA_STEP_STG) */>
__global__ __launch_bounds__(NWC * 128) void fmt_gemm_dma_kernel(GemmArgs g) {
  constexpr int RT = 12, CT = 5 * NWC, NF = RT + CT, NW = 2 * NWC, STAGE = NF * 1024;
  constexpr int IMAX = (NF + NW - 1) / NW, ILO = NF / NW, NHI = NF % NW;  // DMA instructions per wave and stage: IMAX for waves < NHI
  static_assert(NS >= 3 && NS <= 4, "ring of 3 or 4 stages");
  extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];  // [NS][NF fragments][1 KiB]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w / NWC, wc = w % NWC;
  const int r16 = lane & 15, q = lane >> 4;
  const int nbx = g.N / (CT * 16);
  int bx, by, bz;
  {
    // same walk as fmt_gemm_wide_kernel: an XCD (ids congruent mod 8) takes groups of `zgroup` column blocks, the step slowest
    const int id = blockIdx.x, nz = g.zcount > 1 ? g.zcount : 1;
    if ((nbx & 7) == 0 && g.zgroup > 1) {
      const int slot = id >> 3, G = g.zgroup, ncx = nbx >> 3, rows = g.mblk * nz;
      const int full = ncx / G, per_grp = G * rows;
      int grp = slot / per_grp, rem = slot - grp * per_grp, gw = G;
      if (grp >= full) {
        grp = full;
        rem = slot - full * per_grp;
        gw = ncx - full * G;
      }
      const int zy = rem / gw;
      bz = zy / g.mblk;
      by = zy - bz * g.mblk;
      bx = (grp * G + (rem - zy * gw)) * 8 + (id & 7);
    } else {
      bx = id % nbx;
      const int zy = id / nbx;
      bz = zy / g.mblk;
      by = zy - bz * g.mblk;
    }
  }
  const int KB = g.K >> 5;
  const size_t tstride = (size_t)KB * 512;
  // this wave's fragments of a stage: fi = w + NW * i; fi < 12: row tile fi of A, else column tile fi - 12 of W
  const u16* src[IMAX];
#pragma unroll
  for (int i = 0; i < IMAX; ++i) {
    const int fi = min(w + NW * i, NF - 1);
    src[i] = (fi < RT ? g.A + (size_t)bz * g.a_zstride + (size_t)(by * RT + fi) * tstride
                      : g.W + (size_t)(bx * CT + fi - RT) * tstride) + lane * 8;
  }
  auto issue = [&](int s) {
    unsigned char* dst = ring + (s % NS) * STAGE + w * 1024;
#pragma unroll
    for (int i = 0; i < IMAX; ++i)
      if (i < ILO || w < NHI)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)s * 512),
                                         (__attribute__((address_space(3))) void*)(dst + i * NW * 1024), 16, 0, 0);
  };
  // wait until at most K of this wave's stages are still in flight (waves < NHI issue one more instruction per stage)
#define DMA_WAIT(K)                                   \
  do {                                                \
    if (NHI == 0 || w < NHI) fh_wait_vmcnt<(K)*IMAX>(); \
    else fh_wait_vmcnt<(K)*ILO>();                    \
  } while (0)
  const unsigned abase = (unsigned)(lane * 16 + wr * 6 * 1024), bbase = (unsigned)(lane * 16 + (RT + wc * 5) * 1024);

  f32x4 acc[6][5];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 a0[6], b0[5], a1[6], b1[5];

#define DMA_READ(S, AR, BR)                                        \
  do {                                                             \
    const unsigned so = (unsigned)(((S) % NS) * STAGE);            \
    AR[0] = fh_ds_read128<0>(abase + so);                          \
    AR[1] = fh_ds_read128<1024>(abase + so);                       \
    AR[2] = fh_ds_read128<2048>(abase + so);                       \
    AR[3] = fh_ds_read128<3072>(abase + so);                       \
    AR[4] = fh_ds_read128<4096>(abase + so);                       \
    AR[5] = fh_ds_read128<5120>(abase + so);                       \
    BR[0] = fh_ds_read128<0>(bbase + so);                          \
    BR[1] = fh_ds_read128<1024>(bbase + so);                       \
    BR[2] = fh_ds_read128<2048>(bbase + so);                       \
    BR[3] = fh_ds_read128<3072>(bbase + so);                       \
    BR[4] = fh_ds_read128<4096>(bbase + so);                       \
  } while (0)
  // step s: stage s+1 has landed everywhere after the barrier; stage s+NS goes into the buffer of stage s, whose fragments every
  // wave holds in registers since before it arrived at this barrier; the reads of s+1 fly under the MFMAs of s
#define DMA_STEP(S, AC, BC, AN, BN)                                                            \
  do {                                                                                         \
    const int s_ = (S);                                                                        \
    if (s_ + 1 < KB) {                                                                         \
      const int rem = KB - 2 - s_; /* stages issued beyond s+1 */                              \
      if (NS == 4 && rem >= 2) DMA_WAIT(2);                                                    \
      else if (rem >= 1) DMA_WAIT(1);                                                          \
      else DMA_WAIT(0);                                                                        \
    }                                                                                          \
    __builtin_amdgcn_s_barrier();                                                              \
    if (s_ + NS < KB) issue(s_ + NS);                                                          \
    if (s_ + 1 < KB) DMA_READ(s_ + 1, AN, BN);                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    _Pragma("unroll") for (int i = 0; i < 6; ++i)                                              \
      _Pragma("unroll") for (int j = 0; j < 5; ++j) acc[i][j] = T::mfma(BC[j], AC[i], acc[i][j]); \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
    __builtin_amdgcn_sched_barrier(0);                                                         \
  } while (0)

  // Staggered form (STG = 1; 1021 vs 1045 us per launch of the hoisted projection).  In lock step all 8 waves read fragments at
  // the same time and multiply at the same time, and ablation builds showed the LDS time and the MFMA time of a step adding up
  // rather than overlapping.  Here the wave rows run half a step apart: a step is [barrier A | DMA issue + fragment reads of s+1, waited for |
  // barrier B | MFMAs of s], and row 1 executes one extra barrier first - while row 0 multiplies, row 1 (the other wave of
  // each SIMD) reads, and vice versa.  Row 0 reads stage s+1 one barrier before row 1 does, so a wave of row 0 waits for its
  // share of stage s+1 before its barrier A, a wave of row 1 for its share of stage s+2 before its barrier B (the same global
  // barrier).  Stage s+NS is issued after both rows have read stage s.  Row 0 runs one extra barrier at the end: every wave
  // executes 2 KB + 2 barriers.
#define DMA_WAIT_UPTO(K)                      \
  do {                                        \
    const int k_ = (K);                       \
    if (NS == 4 && k_ >= 2) DMA_WAIT(2);      \
    else if (k_ >= 1) DMA_WAIT(1);            \
    else DMA_WAIT(0);                         \
  } while (0)
#define DMA_STEP_STG(S, AC, BC, AN, BN)                                                        \
  do {                                                                                         \
    const int s_ = (S);                                                                        \
    if (!g1 && s_ + 1 < KB) DMA_WAIT_UPTO(KB - 2 - s_);                                        \
    __builtin_amdgcn_s_barrier();                                                              \
    if (s_ + NS < KB) issue(s_ + NS);                                                          \
    if (s_ + 1 < KB) DMA_READ(s_ + 1, AN, BN);                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    if (g1 && s_ + 2 < KB) {                                                                   \
      DMA_WAIT_UPTO(KB - 3 - s_);                                                              \
    }                                                                                          \
    __builtin_amdgcn_s_barrier();                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    _Pragma("unroll") for (int i = 0; i < 6; ++i)                                              \
      _Pragma("unroll") for (int j = 0; j < 5; ++j) acc[i][j] = T::mfma(BC[j], AC[i], acc[i][j]); \
    __builtin_amdgcn_sched_barrier(0);                                                         \
  } while (0)

  const bool g1 = wr == 1;
#ifdef DMA_CLK
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
#endif
#pragma unroll
  for (int s = 0; s < NS; ++s) issue(s);
  if constexpr (STG) DMA_WAIT(NS - 2);  // stages 0 and 1: row 0 reads stage 1 right after the first barrier of the loop
  else DMA_WAIT(NS - 1);
  __builtin_amdgcn_s_barrier();
  DMA_READ(0, a0, b0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (STG) {
    if (g1) __builtin_amdgcn_s_barrier();
    for (int s = 0; s < KB; s += 2) {
      DMA_STEP_STG(s, a0, b0, a1, b1);
      DMA_STEP_STG(s + 1, a1, b1, a0, b0);
    }
    if (!g1) __builtin_amdgcn_s_barrier();
  } else {
    for (int s = 0; s < KB; s += 2) {
      DMA_STEP(s, a0, b0, a1, b1);
      DMA_STEP(s + 1, a1, b1, a0, b0);
    }
  }
#ifdef DMA_CLK
  if ((blockIdx.x == 3000 || blockIdx.x == 6001) && threadIdx.x == 0) {
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    printf("tile loop: %llu core clocks, %llu wall ticks\n", c1 - c0, w1 - w0);
  }
#endif
#undef DMA_STEP_STG
#undef DMA_WAIT_UPTO
#undef DMA_STEP
#undef DMA_READ
#undef DMA_WAIT
  // D[n = q*4 + reg][m = r16]: lane -> one row, 4 consecutive columns per (i, j)
  float* const outz = g.out_f32 + (size_t)bz * g.o_zstride;
  // Through LDS (the ring is idle: nobody reads it after the barrier of the last step), 96 rows per pass, so that a store
  // instruction writes 1 KiB of one output row - 8 whole lines - instead of 16 half lines 205 KB apart (straight from the
  // accumulators the launch took 1260 us instead of 1150).  Row stride OW + 4
  // floats: the 16 rows of a lane group then fall on the 16 different 16-byte bank slots.
  constexpr int OW = CT * 16, OSTR = OW + 4, C4 = OW / 4, NTHR = NW * 64;
  static_assert(96 * OSTR * 4 <= NS * STAGE, "output staging does not fit the ring");
  float* const so = reinterpret_cast<float*>(ring);
  float4 bb[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) bb[j] = *reinterpret_cast<const float4*>(g.bias + (bx * CT + wc * 5 + j) * 16 + q * 4);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (wr == p) {
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j)
          *reinterpret_cast<float4*>(so + (i * 16 + r16) * OSTR + (wc * 5 + j) * 16 + q * 4) =
              float4{acc[i][j][0] + bb[j].x, acc[i][j][1] + bb[j].y, acc[i][j][2] + bb[j].z, acc[i][j][3] + bb[j].w};
    }
    __syncthreads();
    {
#pragma unroll
      for (int k = 0; k < (96 * C4) / NTHR; ++k) {
        const int idx = k * NTHR + threadIdx.x, rl = idx / C4, c4 = idx - rl * C4;
        const int row = by * (RT * 16) + p * 96 + rl;
        if (row < g.M)
          *reinterpret_cast<float4*>(outz + (size_t)row * g.ldo + bx * OW + c4 * 4) = *reinterpret_cast<const float4*>(so + rl * OSTR + c4 * 4);
      }
    }
    if (p == 0) __syncthreads();
  }
}

// LayerNorm (no affine, biased variance, eps 1e-6) + framewise modulate, one wave per token row
// (FMT.py:157,168-169,174-175,197).  out = T( (x-mu)*rstd * (1 + scale[row]) + shift[row] ), written in
// the packed A-operand order of the consuming GEMM (K = D).
// KS > 0: the residual update of the preceding split-K GEMM (EPI_PARTIAL) is folded in first,
//   x[row] += gate[row] * (bias + slab[0][row] + ... + slab[KS-1][row])      (fixed order: bitwise reproducible)
// and the new residual row is written back, so the gated add costs no launch and no extra round trip:
// every operand of the row is requested before the first one is used.
struct LnRed {
  const float* slab;   // [KS][rows][D] partial sums of the GEMM
  size_t slab_stride;
  const float* bias;   // [D]
  const float* gate;   // row stride ldm (a column block of the modulation matrix)
};

constexpr int kLnTouch = 6;  // lines per lane: 192 single-wave workgroups cover 8 XCDs x 1 MB (fc1) with 6

// Body: workgroup `bid` of `nblk`, `wpb` waves per workgroup AS THE ROW MAPPING SEES IT (1 = one row per workgroup, the
// XCD-grouped mapping; fmt_mega_kernel calls it with the first wave of its 8-wave workgroups).  COH as in fmt_gemm_body.
template <class T, int NV, int KS, bool TOUCH, bool WT_, bool COH>
__device__ __forceinline__ void fmt_lnmod_body(float* __restrict__ x, int M, const float* __restrict__ shift,
                                               const float* __restrict__ scale, int ldm, u16* __restrict__ out, const LnRed& red,
                                               const TouchSpec& pf, int ntok, int perm, unsigned long long* sat, const unsigned bid,
                                               const unsigned nblk, const int wpb) {
  constexpr int D = NV * 256;
  constexpr bool WT = WT_ || COH;
  const int lane = threadIdx.x & 63;
  int row = bid * wpb + (threadIdx.x >> 6);
  unsigned touched[kLnTouch] = {};
  const unsigned tfirst = (bid >> 3) * 64 + lane, tstride = (nblk >> 3) * 64;
  const bool touch = TOUCH && wpb == 1;
  if (wpb == 1) {
    // One row per workgroup: a 128-byte line of the packed output holds the 16-byte pieces of 8 consecutive rows, so give
    // those 8 rows to workgroups of ONE XCD (ids congruent mod 8 share an XCD's L2, where the pieces merge into full lines
    // before they are written back; ablation: the stores cost 1 of the kernel's 4.7 us when 8 XCDs each owned a piece).
    const int x = bid & 7, j = bid >> 3;
    row = (((j >> 3) * 8 + x) << 3) + (j & 7);
  }
  if (row >= M) {
    if (touch) {
      fmt_touch(pf, bid & 7, tfirst, tstride, touched);
      fmt_touch_retire(touched);
    }
    return;
  }
  float* xr = x + (size_t)row * D;
#ifdef FMT_DIAG_HOTMOD  // timing-only build (WRONG results): every row reads the modulation row 0 - what the fp32 slab's reads cost
  const size_t mrow = 0;
#else
  const size_t mrow = (size_t)row;
#endif
  const float* sh = shift + mrow * ldm;
  const float* sc = scale + mrow * ldm;
  float4 v[NV], a[NV], b[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 256 + lane * 4;
    v[i] = fh_load_f4<COH>(xr + c);
    a[i] = fh_load_f4_stream(sh + c);
    b[i] = fh_load_f4_stream(sc + c);
  }
  if (KS == 0 && touch) fmt_touch(pf, bid & 7, tfirst, tstride, touched);  // behind the row's own loads (in-order retirement)
  if constexpr (KS > 0) {
    float4 p[KS][NV], gt[NV], bi[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = i * 256 + lane * 4;
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        const float* ps = red.slab + k * red.slab_stride + (size_t)row * D + c;
        if constexpr (COH) p[k][i] = fh_load_f4<true>(ps);
        else p[k][i] = fh_load_f4_stream(ps);  // the slab's only read
      }
      gt[i] = fh_load_f4_stream(red.gate + mrow * ldm + c);
      bi[i] = *reinterpret_cast<const float4*>(red.bias + c);
    }
    if (touch) fmt_touch(pf, bid & 7, tfirst, tstride, touched);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float4 t = p[0][i];
#pragma unroll
      for (int k = 1; k < KS; ++k) {
        t.x += p[k][i].x;
        t.y += p[k][i].y;
        t.z += p[k][i].z;
        t.w += p[k][i].w;
      }
      v[i].x += gt[i].x * (t.x + bi[i].x);
      v[i].y += gt[i].y * (t.y + bi[i].y);
      v[i].z += gt[i].z * (t.z + bi[i].z);
      v[i].w += gt[i].w * (t.w + bi[i].w);
      fh_store_f4_wt<fh_site<COH>(1), WT>(xr + i * 256 + lane * 4, v[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mu = wave_sum(s) * (1.f / D);
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float d0 = v[i].x - mu, d1 = v[i].y - mu, d2 = v[i].z - mu, d3 = v[i].w - mu;
    s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  const float rstd = rsqrtf(wave_sum(s2) * (1.f / D) + 1e-6f);
  // perm = (clips * bc) * 16: token-blocked output rows for the CFG head GEMM (GemmArgs::tokblk); b counts the
  // (clip, CFG row) sequences, so a token block holds clip-major, CFG-row-minor row tiles
  unsigned rm = 0u;
  int orow = row;
  if (perm) {
    const int b = row / ntok, i = row - b * ntok;
    orow = (i >> 4) * perm + b * 16 + (i & 15);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 256 + lane * 4;
    fh_store4_wt<T, fh_site<COH>(1), WT>(reinterpret_cast<typename T::elem*>(out) + fmt_pack_off(orow, c, D / 32),
              (v[i].x - mu) * rstd * (1.f + b[i].x) + a[i].x, (v[i].y - mu) * rstd * (1.f + b[i].y) + a[i].y,
              (v[i].z - mu) * rstd * (1.f + b[i].z) + a[i].z, (v[i].w - mu) * rstd * (1.f + b[i].w) + a[i].w, rm);
  }
  fh_range_flush<T>(sat, rm);
  if (touch) fmt_touch_retire(touched);
}

template <class T, int NV, int KS, bool TOUCH, bool WT>
__global__ __launch_bounds__(256) void fmt_lnmod_kernel(float* __restrict__ x, int M, const float* __restrict__ shift,
                                                        const float* __restrict__ scale, int ldm, u16* __restrict__ out,
                                                        LnRed red, TouchSpec pf, int ntok, int perm, unsigned long long* sat) {
  fmt_lnmod_body<T, NV, KS, TOUCH, WT, false>(x, M, shift, scale, ldm, out, red, pf, ntok, perm, sat, blockIdx.x, gridDim.x,
                                              (int)(blockDim.x >> 6));
}

// Banded attention (FMT.py:71-88 with the mask of FMT.py:15-19): query i sees keys |i-j| <= window.
// LPQ lanes per query, 128/LPQ head dims each; a workgroup is a run of blockDim.x/LPQ queries of one (cfg row, head)
// (blockIdx.y = run), so the 24 (row, head) pairs spread over many CUs - with one 512-thread workgroup per pair only 24 CUs
// pulled q/k/v and the kernel took 5.2 us; single-wave workgroups of 8 queries take 4.1 us.  With at most 2*window+1 keys
// per query this is 0.3 % of the evaluation's flops - MFMA/LDS tiling would only add latency - so q/k/v come straight from
// L2.  For window <= 2 every load of the query's band is issued before the first use (one memory round trip); wider
// windows loop.  Output is written in the packed A-operand order of the proj GEMM (K = D).
// Body: head `h`, run `run` of `qpw` queries (nthr = qpw * LPQ threads take part), `lin` of `nwg` workgroups for the touch.
template <class T, int LPQ, bool TOUCH, bool WT_, bool COH>
__device__ __forceinline__ void fmt_attn_body(const u16* __restrict__ qkv, int ld, u16* __restrict__ out, int ntok, int M, int D,
                                              int window, const TouchSpec& pf, unsigned long long* sat, const int h, const int run,
                                              const int nthr, const unsigned lin, const unsigned nwg) {
  constexpr int HD = 128, PD = HD / LPQ, NU = PD / 8;
  constexpr bool WT = WT_ || COH;
  auto ld8 = [](const typename T::elem* p) -> typename T::pack8 {
    if constexpr (COH && !T::is32) return fh_load16<true>(p);
    else return T::load8(p);
  };
  // blockIdx.x = head, blockIdx.y = run of blockDim.x / LPQ consecutive ROWS of the (cfg rows x tokens) batch: with 8 rows per
  // workgroup the run is one 8-row group of the packed output, whose 128-byte lines then come whole from a single wave
  const int row_ = run * (nthr / LPQ) + threadIdx.x / LPQ, part = threadIdx.x % LPQ;
  unsigned touched[2] = {0u, 0u};
  // issued BEHIND the wave's own q / k / v loads (memory operations retire in order: in front of them, the wave would wait
  // for the touched HBM lines before it could use operands that come from the Infinity Cache)
  auto touch = [&]() {
    if constexpr (TOUCH) {
      fmt_touch(pf, lin & 7, (lin >> 3) * nthr + threadIdx.x, (nwg >> 3) * nthr, touched);
    }
  };
  if (row_ >= M) {  // whole LPQ-lane groups leave together
    touch();
    if constexpr (TOUCH) fmt_touch_retire(touched);
    return;
  }
  const int b = row_ / ntok, qi = row_ - b * ntok;
  const int d0 = h * HD + part * PD;
  const float scale = rsqrtf((float)HD);
  typedef typename T::elem E;
  typedef typename T::pack8 P8;
  const E* base = reinterpret_cast<const E*>(qkv) + (size_t)(b * ntok) * ld + d0;
  float qf[PD], o[PD];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const P8 uu = ld8(base + (size_t)qi * ld + u * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[u * 8 + j] = T::get(uu, j) * scale;
  }
#pragma unroll
  for (int i = 0; i < PD; ++i) o[i] = 0.f;
  float m = -INFINITY, l = 0.f;
  auto fold = [&](const P8* k, const P8* v, bool valid) {
    float dot = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
      for (int j = 0; j < 8; ++j) dot += qf[u * 8 + j] * T::get(k[u], j);
    }
#pragma unroll
    for (int d = 1; d < LPQ; d <<= 1) dot += __shfl_xor(dot, d, 64);
    if (!valid) return;
    const float mn = fmaxf(m, dot);
    const float alpha = __expf(m - mn), p = __expf(dot - mn);
    l = l * alpha + p;
    m = mn;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[u * 8 + j] = o[u * 8 + j] * alpha + p * T::get(v[u], j);
    }
  };
  if (window <= 2) {
    P8 kk[5][NU], vv[5][NU];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const int kj = min(max(qi + t - 2, 0), ntok - 1);
      const E* kp = base + (size_t)kj * ld + D;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        kk[t][u] = ld8(kp + u * 8);
        vv[t][u] = ld8(kp + D + u * 8);
      }
    }
    touch();
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const int kj = qi + t - 2;
      const bool valid = kj >= 0 && kj < ntok && (t - 2 >= -window) && (t - 2 <= window);
      fold(kk[t], vv[t], valid);
    }
  } else {
    touch();
    for (int kj = qi - window; kj <= qi + window; ++kj) {
      const int kc = min(max(kj, 0), ntok - 1);
      const E* kp = base + (size_t)kc * ld + D;
      P8 k[NU], v[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        k[u] = ld8(kp + u * 8);
        v[u] = ld8(kp + D + u * 8);
      }
      fold(k, v, kj >= 0 && kj < ntok);
    }
  }
  const float inv = 1.f / l;
  const int row = b * ntok + qi;
  unsigned rm = 0u;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    P8 uo;
#pragma unroll
    for (int j = 0; j < 8; ++j) T::set(uo, j, o[u * 8 + j] * inv);
    fh_track_pack<T>(rm, uo);
    T::template store8_wt<fh_site<COH>(2), WT>(reinterpret_cast<E*>(out) + fmt_pack_off(row, d0 + u * 8, D / 32), uo);
  }
  fh_range_flush<T>(sat, rm);
  if constexpr (TOUCH) fmt_touch_retire(touched);
}

template <class T, int LPQ, bool TOUCH, bool WT>
__global__ __launch_bounds__(512) void fmt_attn_kernel(const u16* __restrict__ qkv, int ld, u16* __restrict__ out, int ntok,
                                                       int M, int D, int window, TouchSpec pf, unsigned long long* sat) {
  fmt_attn_body<T, LPQ, TOUCH, WT, false>(qkv, ld, out, ntok, M, D, window, pf, sat, blockIdx.x, blockIdx.y, blockDim.x,
                                          blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
}

// Banded attention AND attn.proj in one launch (FMT.py:71-89): attention is per head, the projection sums over heads, so a
// workgroup = (row block of MTW*16 rows, 128-column block of proj, head h) computes the head's attention output for its rows
// (the same register scheme as fmt_attn_kernel: LPQ = 8 lanes per query, every load of the band issued before the first use),
// leaves it in LDS in A-fragment order and multiplies it by rows h*128.. of W_proj: split-K over the heads with NO exchange
// between workgroups.  Slab h of the EPI_PARTIAL buffer receives the partial sums; the LayerNorm launch that follows folds
// bias + gate * sum_h slab[h] into the residual stream in a fixed order (fmt_lnmod_kernel<.., KS = heads>).  The attention
// of a row block is recomputed by each of its N/128 column blocks (0.03 MFLOP); the proj weights are requested first, so
// their round trip overlaps the q/k/v round trip.  Heads <-> XCDs (ids congruent mod 8 share an L2): XCD h fetches only head
// h's columns of q|k|v and the k-slice h of W_proj.  Operands swapped (D = W_tile * O_tile^T): a lane holds 4 consecutive
// columns of one row, stored as one float4.  HPW = 2: two heads per workgroup (wave = (column tile, head), 64-column blocks),
// the pair summed through LDS: half the slab bytes, two attention passes.
// MEASURED (round 4, ms per 250 evaluations, same box): two launches 80.2-80.8; this kernel 81.4 (HPW = 1), 81.7-82.0 (HPW = 2),
// 80.8 with non-temporal slab stores - a tie, so the two-launch chain stays the default (FLOAT_FMT_ATTNPROJ=1|2 selects this
// one).  Diagnostic builds: without the slab stores 76.5, with the LayerNorm reading one hot slab 79.9, attention arithmetic
// free (0.2): the launch it removes (4.7 us) is paid back by heads x M x N x 4 B of partial sums leaving each XCD's L2 at the
// kernel boundary (5.9 MB, ~2.7 us) and coming back into the LayerNorm (+0.9 us).
// sum over the LPQ = 8 lanes of a query by DPP (quad_perm xor 1, xor 2, then row_half_mirror: lane i <-> 7 - i of an 8-lane
// group, whose quads hold their own sums by then): 3 VALU instructions instead of 3 ds_bpermute round trips
__device__ __forceinline__ float fh_sum8_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  return v;
}

template <class T, int MTW, int HPW /* heads per workgroup: K slices = heads / HPW */>
__global__ __launch_bounds__(512) void fmt_attnproj_kernel(const u16* __restrict__ qkv, int ld, GemmArgs g, int ntok, int D, int window) {
  constexpr int HD = 128, LPQ = 8, NU = HD / (LPQ * 8), KBH = HD / 32, NWV = 8, ROWS = MTW * 16;
  constexpr int NCT = NWV / HPW;                       // 16-column tiles per workgroup: wave = (column tile, head of the group)
  constexpr int NPAIR = ROWS * HPW, NPASS = (NPAIR * LPQ + NWV * 64 - 1) / (NWV * 64);
  typedef typename T::elem E;
  typedef typename T::pack8 P8;
  __shared__ __attribute__((aligned(16))) E sO[HPW * MTW * KBH * 512];  // [head][row tile][k-block][lane][8]
  __shared__ __attribute__((aligned(16))) float sR[(HPW > 1 ? (HPW - 1) : 1) * NCT * MTW * 256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r16 = lane & 15, q4 = lane >> 4;
  const int ct = w % NCT, hh = w / NCT;
  int ks, bx, by;
  {
    // (column block, K slice) pairs <-> XCDs as in fmt_gemm_kernel: slice ks on the 8 / ksplit XCDs that own it
    const int id = blockIdx.x, nbn = g.N / (NCT * 16), nbx = nbn * g.ksplit;
    if ((nbx & 7) == 0 && (8 % g.ksplit) == 0) {
      const int slot = id >> 3, P = 8 / g.ksplit, x = id & 7;
      by = slot % g.mblk;
      ks = x / P;
      bx = (slot / g.mblk) * P + (x % P);
    } else {
      const int bxl = id % nbx;
      by = id / nbx;
      ks = bxl / nbn;
      bx = bxl - ks * nbn;
    }
  }
  const int KB = g.K >> 5;
  const int h0 = ks * HPW;      // first head of this workgroup
  const int nt = bx * NCT + ct; // this wave's 16-column tile of proj
  const E* Wp = reinterpret_cast<const E*>(g.W) + ((size_t)nt * KB + (h0 + hh) * KBH) * 512 + lane * 8;
  P8 b[KBH];
#pragma unroll
  for (int kk = 0; kk < KBH; ++kk) b[kk] = T::load8_nt(Wp + (size_t)kk * 512);

  const float scale = rsqrtf((float)HD);
  unsigned rm = 0u;
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    const int pair = pass * (NWV * 64 / LPQ) + threadIdx.x / LPQ, part = threadIdx.x % LPQ;
    if (pair < NPAIR) {
      const int ah = pair / ROWS, qslot = pair - ah * ROWS;  // head of the group, row of the block
      const int row_ = by * ROWS + qslot;
      P8 uo[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) uo[u] = T::zero8();
      if (row_ < g.M) {  // whole LPQ-lane groups take the branch together
        const int bq = row_ / ntok, qi = row_ - bq * ntok;
        // lane `part` owns head dims u*64 + part*8 .. +7: one load instruction reads 128 contiguous bytes per query
        const E* base = reinterpret_cast<const E*>(qkv) + (size_t)(bq * ntok) * ld + (h0 + ah) * HD + part * 8;
        float qf[NU * 8], o[NU * 8];
        P8 qq[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) qq[u] = T::load8(base + (size_t)qi * ld + u * 64);
#pragma unroll
        for (int i = 0; i < NU * 8; ++i) o[i] = 0.f;
        float l = 0.f;
        auto dotk = [&](const P8* k) {
          float dot = 0.f;
#pragma unroll
          for (int u = 0; u < NU; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) dot += qf[u * 8 + j] * T::get(k[u], j);
          }
          return fh_sum8_dpp(dot);
        };
        if (window <= 2) {
          // the band's five keys: every load issued before the first use, the five dot products reduced side by side, then
          // a plain (not running) softmax over them
          P8 kk[5][NU], vv[5][NU];
#pragma unroll
          for (int t = 0; t < 5; ++t) {
            const int kj = min(max(qi + t - 2, 0), ntok - 1);
            const E* kp = base + (size_t)kj * ld + D;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
              kk[t][u] = T::load8(kp + u * 64);
              vv[t][u] = T::load8(kp + D + u * 64);
            }
          }
#pragma unroll
          for (int u = 0; u < NU; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[u * 8 + j] = T::get(qq[u], j) * scale;
          }
          float dot[5], m = -INFINITY;
#pragma unroll
          for (int t = 0; t < 5; ++t) {
            const int kj = qi + t - 2;
            const bool valid = kj >= 0 && kj < ntok && (t - 2 >= -window) && (t - 2 <= window);
            dot[t] = valid ? dotk(kk[t]) : -INFINITY;
            m = fmaxf(m, dot[t]);
          }
#pragma unroll
          for (int t = 0; t < 5; ++t) {
            const float p = __expf(dot[t] - m);  // exp(-inf) = 0 for the keys outside the band / the sequence
            l += p;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
#pragma unroll
              for (int j = 0; j < 8; ++j) o[u * 8 + j] += p * T::get(vv[t][u], j);
            }
          }
        } else {
#pragma unroll
          for (int u = 0; u < NU; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[u * 8 + j] = T::get(qq[u], j) * scale;
          }
          float m = -INFINITY;
          for (int kj = max(qi - window, 0); kj <= min(qi + window, ntok - 1); ++kj) {
            const E* kp = base + (size_t)kj * ld + D;
            P8 k[NU], v[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
              k[u] = T::load8(kp + u * 64);
              v[u] = T::load8(kp + D + u * 64);
            }
            const float dot = dotk(k);
            const float mn = fmaxf(m, dot);
            const float alpha = __expf(m - mn), p = __expf(dot - mn);
            l = l * alpha + p;
            m = mn;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
#pragma unroll
              for (int j = 0; j < 8; ++j) o[u * 8 + j] = o[u * 8 + j] * alpha + p * T::get(v[u], j);
            }
          }
        }
        const float inv = 1.f / l;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
#pragma unroll
          for (int j = 0; j < 8; ++j) T::set(uo[u], j, o[u * 8 + j] * inv);
          fh_track_pack<T>(rm, uo[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) T::store8(sO + ah * (MTW * KBH * 512) + fmt_pack_off(qslot, u * 64 + part * 8, KBH), uo[u]);
    }
  }
  fh_range_flush<T>(g.sat, rm);
  __syncthreads();

  f32x4 acc[MTW];
#pragma unroll
  for (int i = 0; i < MTW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const E* const sOh = sO + hh * (MTW * KBH * 512);
#pragma unroll
  for (int kk = 0; kk < KBH; ++kk) {
#pragma unroll
    for (int i = 0; i < MTW; ++i) acc[i] = T::mfma(b[kk], T::load8(sOh + (i * KBH + kk) * 512 + lane * 8), acc[i]);
  }
  if constexpr (HPW > 1) {
    // the group's heads meet in LDS, summed in head order by the wave of head 0
    if (hh > 0) {
#pragma unroll
      for (int i = 0; i < MTW; ++i) *reinterpret_cast<f32x4*>(sR + (((hh - 1) * NCT + ct) * MTW + i) * 256 + lane * 4) = acc[i];
    }
    __syncthreads();
    if (hh > 0) return;
#pragma unroll
    for (int o = 1; o < HPW; ++o) {
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(sR + (((o - 1) * NCT + ct) * MTW + i) * 256 + lane * 4);
        acc[i] += t;
      }
    }
  }
  // D[n = q4*4 + reg][m = r16]
  float* const outp = g.out_f32 + (size_t)ks * g.slab_stride + nt * 16 + q4 * 4;
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
    const int row = (by * MTW + i) * 16 + r16;
    // non-temporal: the slabs are read once, by the LayerNorm launch, from other XCDs (80.8 vs 81.4 ms per 250 evaluations with plain stores)
    if (row < g.M) __builtin_nontemporal_store(acc[i], reinterpret_cast<f32x4*>(outp + (size_t)row * g.ldo));
  }
}

// ------------------------------------------------------------------------------------------
// ONE evaluation of the step chain as ONE persistent kernel (fmt_mega_kernel): x-embed, 8 x [LN, qkv, attention, proj, LN,
// fc1, fc2], LN, head = 59 stages that were 59 dependent launches.  The stages are the SAME bodies (fmt_gemm_body /
// fmt_lnmod_body / fmt_attn_body: same tilings, same arithmetic, bit for bit the launch chain's results); what changes is how
// they meet: 256 workgroups (one per CU, all resident) pass a grid barrier between stages instead of a kernel boundary.
// MEASURED (round 4, MI355X, one box): 375 us per evaluation against 322 for the launch chain - it LOSES, so it is opt-in
// (FLOAT_FMT_MEGA=1) and the launch chain stays the default.  In-kernel stamps (-DMEGA_STAMPS, tools/probes/mega_stamps.py), us
// per stage: bodies LN 2.7 / qkv 3.9 / attention 2.6 / proj 2.8 / fc1 5.5 / fc2 4.6 (the launch chain's kernels minus ~2.5 us
// of boundary: 2.2 / 2.9 / 2.2 / 2.4 / 4.9 / 4.4), + 0.4-1.7 draining the write-through stores, + 1.6-2.5 in the barrier:
// the software hand-off costs what the hardware's launch boundary costs, the bodies pay for coherent loads and sc1 stores,
// and a 256 x 512-thread launch adds ~29 us per evaluation (workgroup dispatch skew ~6 us at the first stage).
// Probes (tools/probes/grid_barrier.hip): hierarchical in-kernel barrier 2.1 us, 2.8 us with a stage's data exchanged through
// sc1 stores / loads; agent-scope release / acquire FENCES (buffer_wbl2 / buffer_inv) 31 us, buffer_inv sc1 alone 4-9 us -
// so the kernel never fences: every activation store of a stage writes through (sc1), a
// workgroup drains its stores (s_waitcnt vmcnt(0)) before it arrives, and every activation load of the next stage is an
// agent-scope load (COH = true).  Weights, biases and modulations are read-only for the whole kernel: plain loads.
// Barrier: 8 group counters (workgroups congruent mod 8 = one XCD, 32 arrivals each on their own 128-byte line), the last
// arriver of a group bumps a top counter, the last of those writes the generation into 8 per-group release words that the
// waiters poll - three dependent memory round trips instead of 256 serialized atomics on one line.  Counters only ever count
// up: generation = launches so far (a device word read at kernel start, bumped by workgroup 0 at the end) x barriers per
// launch + stage, so a hipGraph can replay the same launch.  A waiter that does not see its release within ~0.3 s (a CU was
// not available, so not all workgroups are resident) sets the error word (and its host-mapped twin) and leaves; the host
// reports it at the next FMT call on the handle, whatever its dtype (mega_poll in fmt_api.hip: that call fails, the barrier
// words are cleared and the handle falls back to the launch chain) instead of hanging the GPU.
enum { MS_XEMBED = 0, MS_LN, MS_QKV, MS_ATTN, MS_PROJ, MS_FC1, MS_FC2, MS_HEAD };
struct MegaStage {
  int kind;
  unsigned nblk;                 // workgroups with work in this stage (the rest only pass the barrier)
  GemmArgs g;                    // GEMM stages; gate / head fields are patched per launch
  long long gate_off;            // MS_PROJ: gate = mod + gate_off
  long long shift_off, scale_off, red_gate_off;  // MS_LN: modulation rows / gate of the folded split-K GEMM, from `mod`
  int ks;                        // MS_LN: slabs folded first (0 / 4)
  const float* red_bias;
  u16* ln_out;
  int perm;
  TouchSpec pf;                  // MS_LN / MS_ATTN: weights of the next GEMM to pull into L2
  u16* att_out;                  // MS_ATTN: this block's attention output (A operand of its proj: a write-once buffer)
};
struct MegaCtx {                 // per handle and CFG shape
  float* xres;
  const u16* qkv16;
  const float* slab;
  size_t slab_stride;
  int M, D, ntok, ldm, window, heads;
  unsigned long long* sat;
};
struct MegaDyn {                 // per launch
  const float* mod;
  float dt, a, r, e;
  int euler;
  float* vout;
};
struct MegaSync {
  unsigned* grp;   // [8] x 32 words
  unsigned* top;   // [1]
  unsigned* rel;   // [8] x 32 words
  unsigned* seq;   // launches so far
  unsigned* err;   // != 0: a barrier timed out
  unsigned* err_host;  // the same flag in host-mapped memory: the host reads it at its next call without a device copy
  unsigned long long* stamps;  // diagnostic builds (-DMEGA_STAMPS): [stage][3] s_memrealtime of workgroup `stamp_wg`
  unsigned stamp_wg;
};

// (A flat form - every workgroup adds 1 to its group's counter and polls all 8 counters, two dependent round trips instead of
// four - measured WORSE: 2.4-4.0 us per barrier against 1.6-2.5, 256 pollers x 8 lines.)  Called by thread 0.
__device__ __forceinline__ bool fmt_grid_barrier(const MegaSync& sy, unsigned gen, unsigned nwg) {
  typedef __attribute__((address_space(1))) unsigned int gu32;
  const unsigned g = blockIdx.x & 7u, per = nwg >> 3;
  const unsigned old = __hip_atomic_fetch_add((gu32*)(sy.grp + g * 32), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (old + 1u == gen * per) {
    const unsigned t = __hip_atomic_fetch_add((gu32*)sy.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t + 1u == gen * 8u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) __hip_atomic_store((gu32*)(sy.rel + i * 32), gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  unsigned spins = 0;
  while ((int)(__hip_atomic_load((gu32*)(sy.rel + g * 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - gen) < 0) {
    if (++spins > (1u << 18) || ((spins & 63u) == 0u && __hip_atomic_load((gu32*)sy.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
      __hip_atomic_store((gu32*)sy.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (sy.err_host) __hip_atomic_store((gu32*)sy.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  return true;
}

template <class T>
__global__ __launch_bounds__(512) void fmt_mega_kernel(const MegaStage* __restrict__ tab, int nstage, MegaDyn d, MegaCtx c,
                                                       MegaSync sy) {
  typedef __attribute__((address_space(1))) unsigned int gu32;
  __shared__ unsigned s_flag[2];
  const unsigned bid = blockIdx.x, nwg = gridDim.x;
  if (threadIdx.x == 0) {
    s_flag[0] = __hip_atomic_load((gu32*)sy.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_flag[1] = 1u;
  }
  __syncthreads();
  const unsigned seq0 = s_flag[0];
  const unsigned gen0 = seq0 * (unsigned)(nstage - 1);
#ifdef MEGA_STAMPS
#define MEGA_STAMP(k) do { if (sy.stamps && bid == sy.stamp_wg && threadIdx.x == 0) sy.stamps[s * 3 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MEGA_STAMP(k) do { } while (0)
#endif
  for (int s = 0; s < nstage; ++s) {
    const MegaStage& st = tab[s];
    MEGA_STAMP(0);
    if (bid < st.nblk) {
      switch (st.kind) {
        case MS_XEMBED:
          fmt_gemm_body<T, 4, 1, 8, EPI_XEMBED, true>(st.g, bid, st.nblk);
          break;
        case MS_LN:
          if (threadIdx.x < 64) {
            const LnRed red{c.slab, c.slab_stride, st.red_bias, d.mod + st.red_gate_off};
            if (st.ks == 0)
              fmt_lnmod_body<T, 4, 0, true, true, true>(c.xres, c.M, d.mod + st.shift_off, d.mod + st.scale_off, c.ldm, st.ln_out, red,
                                                        st.pf, c.ntok, st.perm, c.sat, bid, st.nblk, 1);
            else
              fmt_lnmod_body<T, 4, 4, true, true, true>(c.xres, c.M, d.mod + st.shift_off, d.mod + st.scale_off, c.ldm, st.ln_out, red,
                                                        st.pf, c.ntok, st.perm, c.sat, bid, st.nblk, 1);
          }
          break;
        case MS_QKV:
          fmt_gemm_body<T, 3, 4, 8, EPI_T16, true>(st.g, bid, st.nblk);
          break;
        case MS_ATTN:
          if (threadIdx.x < 128)
            fmt_attn_body<T, 16, true, true, true>(c.qkv16, 3 * c.D, st.att_out, c.ntok, c.M, c.D, c.window, st.pf, c.sat,
                                                   (int)(bid % (unsigned)c.heads), (int)(bid / (unsigned)c.heads), 128, bid, st.nblk);
          break;
        case MS_PROJ: {
          GemmArgs g = st.g;
          g.gate = d.mod + st.gate_off;
          fmt_gemm_body<T, 3, 1, 8, EPI_GATE_RES, true>(g, bid, st.nblk);
          break;
        }
        case MS_FC1:
          fmt_gemm_body<T, 3, 4, 8, EPI_GELU_P16, true>(st.g, bid, st.nblk);
          break;
        case MS_FC2:
          fmt_gemm_body<T, 3, 4, 8, EPI_PARTIAL, true>(st.g, bid, st.nblk);
          break;
        default: {  // MS_HEAD
          GemmArgs g = st.g;
          g.a_cfg = d.a;
          g.r_cfg = d.r;
          g.e_cfg = d.e;
          g.dt = d.dt;
          if (!d.euler) {
            g.xcur = nullptr;
            g.xin16 = nullptr;
            g.vout = d.vout;
          }
          fmt_gemm_body<T, 3, 1, 8, EPI_CFG, true>(g, bid, st.nblk);
          break;
        }
      }
    }
    MEGA_STAMP(1);
    if (s + 1 < nstage) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this workgroup's write-through stores have left
      __syncthreads();
      MEGA_STAMP(2);
      if (threadIdx.x == 0) {
        // the next stage's descriptor (5 lines of the table) into the scalar cache while the barrier is awaited (-0.3 us per stage)
        int pre = 0;
#pragma unroll
        for (int i = 0; i < (int)(sizeof(MegaStage) + 63) / 64; ++i) pre ^= reinterpret_cast<const int*>(&tab[s + 1])[i * 16];
        asm volatile("" ::"s"(pre));
        if (!fmt_grid_barrier(sy, gen0 + (unsigned)s + 1u, nwg)) s_flag[1] = 0u;
      }
      __syncthreads();
      if (s_flag[1] == 0u) return;
    }
  }
  if (bid == 0 && threadIdx.x == 0) __hip_atomic_store((gu32*)sy.seq, seq0 + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Condition rows for c_embedder: [wr | wa | we | 0-pad] per (cfg row b, token i) with the CFG nulling
// pattern given as bit masks over b (FMT.py:322-333, 360-373, 382-392).
template <class T>
__global__ void fmt_build_cond_kernel(u16* __restrict__ out, int ld, int bc, int ntok, int n_prev, int dim_w, int dim_a,
                                      int dim_e, const float* __restrict__ wr, const float* __restrict__ wa,
                                      const float* __restrict__ prev_wa, const float* __restrict__ we, int we_len,
                                      const float* __restrict__ prev_we, unsigned wr_mask, unsigned wa_mask,
                                      unsigned we_mask, unsigned long long* sat) {
  // row = (clip * bc + b) * ntok + i; per-clip tensors are stacked: wr (clips, dim_w), wa (clips, n_cur, dim_a),
  // prev_wa (clips, n_prev, dim_a), we (clips, we_len, dim_e), prev_we (clips, n_prev, dim_e)
  const int row = blockIdx.x;
  const int s = row / ntok, i = row % ntok, q = s / bc, b = s - q * bc, n_cur = ntok - n_prev;
  const bool on_r = (wr_mask >> b) & 1, on_a = (wa_mask >> b) & 1, on_e = (we_mask >> b) & 1;
  wr += (size_t)q * dim_w;
  wa += (size_t)q * n_cur * dim_a;
  prev_wa += (size_t)q * n_prev * dim_a;
  we += (size_t)q * we_len * dim_e;
  if (prev_we) prev_we += (size_t)q * n_prev * dim_e;
  unsigned rm = 0u;
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
    reinterpret_cast<typename T::elem*>(out)[fmt_pack_off(row, c, ld / 32)] = fh_cvt<T>(v, rm);
  }
  fh_range_flush<T>(sat, rm);
}

// Sinusoidal timestep features [cos(t f_k) | sin(t f_k)], k < 128 (FMT.py:118-123), one row per evaluation of a window.
// The evaluation times are formed here: grid point i of torch.linspace(0, 1, nfe) in fp32 (symmetric around the midpoint:
// i * step below it, 1 - (nfe-1-i) * step above), and for a Runge-Kutta scheme with `stages` stages per step
// t = g_i + c_j (g_{i+1} - g_i) for evaluation s = i * stages + j.  nfe == 0: the single explicit time t0 (float_fmt_eval).
// Separately rounded multiplies and adds (no fma contraction): the times must equal the host's / torch's bit for bit.
__device__ __forceinline__ float fmt_linspace01(int i, int n) {
  if (n <= 1) return 0.f;
  const float step = __fdiv_rn(1.0f, (float)(n - 1));
  return (i < n / 2) ? __fmul_rn(step, (float)i) : __fsub_rn(1.0f, __fmul_rn(step, (float)(n - 1 - i)));
}
template <class T>
__global__ void fmt_tsin_kernel(u16* __restrict__ out, const float* __restrict__ freqs, int n_steps, float t0, int nfe,
                                int stages, float c0, float c1, float c2, float c3) {
  const int s = blockIdx.x;
  if (s >= n_steps) return;
  float t = t0;
  if (nfe > 0) {
    const int i = s / stages, j = s - i * stages;
    const float gi = fmt_linspace01(i, nfe);
    t = gi;
    if (j > 0) {
      const float cj = j == 1 ? c1 : (j == 2 ? c2 : c3);
      t = __fadd_rn(gi, __fmul_rn(__fsub_rn(fmt_linspace01(i + 1, nfe), gi), cj));
    }
  }
  (void)c0;
  const int k = threadIdx.x;  // 0..255
  const float arg = t * freqs[k & 127];
  reinterpret_cast<typename T::elem*>(out)[fmt_pack_off(s, k, 8)] = T::from_float(k < 128 ? cosf(arg) : sinf(arg));
}

// sc16[row] = T(silu(t_emb + c_cond[row]))   (c = t + c_embedder(.), FMT.py:335; SiLU of FMT.py:164,187)
// written packed: the A operand of the fused adaLN projection.  8 columns per thread.
template <class T>
__global__ void fmt_silu_c_kernel(u16* __restrict__ out, const float* __restrict__ temb, const float* __restrict__ ccond,
                                  int M, int D, size_t step_stride, int dense_rows, unsigned long long* sat) {
  // blockIdx.y = Euler step: all steps of a window are produced by one launch when they fit.  dense_rows = 0: one packed image
  // per step, step_stride elements apart; dense_rows = M: ONE packed image whose row z * M + r is row r of step z (fmt_gemm_big4_kernel)
  const int idx = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (idx >= M * D) return;
  const int c = idx % D, row = idx / D + (int)blockIdx.y * dense_rows;
  typename T::elem* const o = reinterpret_cast<typename T::elem*>(out) + (dense_rows ? (size_t)0 : (size_t)blockIdx.y * step_stride);
  temb += (size_t)blockIdx.y * D;
  typename T::pack8 u;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float4 a = *reinterpret_cast<const float4*>(ccond + idx + 4 * h);
    const float4 t = *reinterpret_cast<const float4*>(temb + c + 4 * h);
    T::set(u, 4 * h + 0, fh_silu(a.x + t.x));
    T::set(u, 4 * h + 1, fh_silu(a.y + t.y));
    T::set(u, 4 * h + 2, fh_silu(a.z + t.z));
    T::set(u, 4 * h + 3, fh_silu(a.w + t.w));
  }
  unsigned rm = 0u;
  fh_track_pack<T>(rm, u);
  fh_range_flush<T>(sat, rm);
  T::store8(o + fmt_pack_off(row, c, D / 32), u);
}

// Euler state and x_embedder input rows for a new window, per clip q: xcur[q] = x0[q]; xin16 rows q * ntok + i = [prev_x[q] ; x0[q]].
template <class T>
__global__ void fmt_init_x_kernel(float* __restrict__ xcur, u16* __restrict__ xin16, int ldx, const float* __restrict__ x0,
                                  const float* __restrict__ prev_x, int n_prev, int n_cur, int dim_w, int nclip,
                                  unsigned long long* sat) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int ntok = n_prev + n_cur;
  if (idx >= nclip * ntok * dim_w) return;
  const int q = idx / (ntok * dim_w), r = idx - q * ntok * dim_w;
  const int i = r / dim_w, c = r % dim_w;
  float v;
  if (i < n_prev) {
    v = prev_x[((size_t)q * n_prev + i) * dim_w + c];
  } else {
    v = x0[((size_t)q * n_cur + (i - n_prev)) * dim_w + c];
    xcur[((size_t)q * n_cur + (i - n_prev)) * dim_w + c] = v;
  }
  unsigned rm = 0u;
  reinterpret_cast<typename T::elem*>(xin16)[fmt_pack_off(q * ntok + i, c, ldx)] = fh_cvt<T>(v, rm);
  fh_range_flush<T>(sat, rm);
}

// Explicit Runge-Kutta glue for the non-Euler fixed-grid solvers: y = xcur + sum_m coef[m] * k_m over the
// current-window rows of every clip (coef already includes dt).  final = 0: y is the next stage's input (packed 16-bit
// rows of the x_embedder operand); final = 1: y becomes the new state and the next step's stage-0 input.
// kbuf: [stage][clip][ntok][W] with stage stride kstride.
template <class T>
__global__ void fmt_rk_combine_kernel(float* __restrict__ xcur, const float* __restrict__ kbuf, size_t kstride, int nk, float c0,
                                      float c1, float c2, float c3, int final, u16* __restrict__ xin16, int ldx, int n_prev,
                                      int n_cur, int W, int nclip, unsigned long long* sat) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nclip * n_cur * W) return;
  const int q = idx / (n_cur * W), r = idx - q * n_cur * W;
  const int i = r / W, c = r % W, ntok = n_prev + n_cur;
  const float cf[4] = {c0, c1, c2, c3};
  float y = xcur[idx];
  for (int m = 0; m < nk; ++m) y += cf[m] * kbuf[(size_t)m * kstride + ((size_t)q * ntok + n_prev + i) * W + c];
  if (final) xcur[idx] = y;
  unsigned rm = 0u;
  reinterpret_cast<typename T::elem*>(xin16)[fmt_pack_off(q * ntok + n_prev + i, c, ldx)] = fh_cvt<T>(y, rm);
  fh_range_flush<T>(sat, rm);
}

// Window slice with replicate padding along time (FLOAT.py:224-227), per clip: dst[q][i] = src[q][min(t0+i, T-1)].
__global__ void fmt_slice_pad_kernel(float* __restrict__ dst, const float* __restrict__ src, int t0, int T, int n, int dim,
                                     int nclip) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nclip * n * dim) return;
  const int q = idx / (n * dim), r = idx - q * n * dim;
  const int i = r / dim, c = r % dim;
  const int t = min(t0 + i, T - 1);
  dst[idx] = src[((size_t)q * T + t) * dim + c];
}

// AR hand-off (FLOAT.py:217-222; nodes_adv.py:682-686), per clip: dst[q] = the last n_prev rows of src[q] (n_cur rows).
__global__ void fmt_tail_kernel(float* __restrict__ dst, const float* __restrict__ src, int n_prev, int n_cur, int dim, int nclip) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nclip * n_prev * dim) return;
  const int q = idx / (n_prev * dim), r = idx - q * n_prev * dim;
  dst[idx] = src[((size_t)q * n_cur + (n_cur - n_prev)) * dim + r];
}

// Test hook (float_fmt_debug): fp32 rows -> the row-major 16-bit q|k|v operand of fmt_attn_kernel, and its packed output back.
template <class T>
__global__ void fmt_dbg_to16_kernel(u16* __restrict__ dst, const float* __restrict__ src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) reinterpret_cast<typename T::elem*>(dst)[i] = T::from_float(src[i]);
}
template <class T>
__global__ void fmt_dbg_unpack_kernel(float* __restrict__ dst, const u16* __restrict__ src, int rows, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * D) return;
  const int r = i / D, c = i - r * D;
  dst[i] = T::to_float(reinterpret_cast<const typename T::elem*>(src)[fmt_pack_off(r, c, D / 32)]);
}

// Device-side copies / fills of the sampling calls.  Kernels, not hipMemcpyAsync / hipMemsetAsync: when the caller captures the
// stream those become memcpy / memset NODES, and replays of such a graph were not reproducible on ROCm 7.2 (the r_d copy of
// window k raced with window k+1's writes to the Euler state: the first 16 frames of a window differed from replay to replay);
// kernel nodes of one captured stream keep their order.  rows x width floats, pitches in floats; src == nullptr fills zeros.
__global__ void fmt_copy_kernel(float* __restrict__ dst, size_t dpitch, const float* __restrict__ src, size_t spitch, int width, int rows) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * width) return;
  const int r = idx / width, c = idx - r * width;
  dst[(size_t)r * dpitch + c] = src ? src[(size_t)r * spitch + c] : 0.f;
}
