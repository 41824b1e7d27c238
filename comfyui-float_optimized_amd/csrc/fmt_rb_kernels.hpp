// Row-blocked GEMM of the FMT step chain for STACKED clips (float_fmt_sample_batch: M = clips x CFG rows x tokens >= 360 rows).
//
// fmt_gemm_kernel (fmt_kernels.hpp) is a weight-streaming tiling: 48 x 64 tiles whose operands come straight from L2 into
// registers, every workgroup re-reading its 48 activation rows per column block and its weight columns per row block.  At one
// clip (180 rows) that is the right trade (the chain is launch-latency-bound); at 720 rows it moves 960 x 224 KB = 215 MB
// through the L2 -> CU paths per GEMM and runs at 0.13-0.16 of the MFMA peak (4 x the rows cost 2.2 x the time).
// Here a workgroup owns a (2 MI x 16) x (2 NJ x 16) tile - 96 x 64, 96 x 128 or 192 x 128 by the row count - and both
// operands reach the CU ONCE per tile, by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction) into a ring of NS stages:
//   * the fragment-major HBM images of A and W (fmt_pack.hpp) are lane-linear, so the LDS image needs no swizzle and every
//     ds_read_b128 of a fragment is conflict-free; a stage holds KPS k-blocks of (RT + CT) fragments;
//   * NS - 1 stages in flight behind a COUNTED vmcnt, ONE raw s_barrier per stage (fmt_gemm_dma_kernel's skeleton);
//   * epilogues of the step chain on the tile left in LDS: EPI_T16 (qkv: row-major 16-bit), EPI_GELU_P16 (fc1: GELU, packed
//     for fc2, whole 1-KiB fragments per 64 threads), EPI_PARTIAL (fc2 / proj: fp32 split-K slab, folded by the next
//     LayerNorm launch), EPI_F32.  Operands swapped (D = W_tile A_tile^T): a lane holds 4 consecutive columns of one row.
// Block decode as fmt_gemm_kernel: the row blocks of a column block sit on ONE XCD (ids congruent mod 8), K slices <-> XCDs.
// Rows past M inside the last row block are read (the operand buffers are padded, Mpad) and never stored.
//
// What bounds it (tools/probes/dma_rate.hip, gemm_lab.hip; MI355X): a CU takes in 125-137 GB/s by LDS-DMA from its XCD's L2,
// but the Infinity Cache hands out ~7.5 TB/s chip-wide (28-31 GB/s per CU when every CU misses L2) and HBM 6 TB/s.  At
// 720 rows each of the 256 tiles needs 458 KB (fc1: 96 + 128 rows of K = 1024) - the least any 256-way split of this GEMM can
// need - so the tile runs at the fetch rate, not at the MFMA rate: in-kernel stamps (-DRB_STAMPS) put fc1 at 1.5 us until the
// first stage has landed + 16 stages x 0.38 us + 1.9 us epilogue and store drain = 10.0 us against 16.6 for the 48 x 64 tiling.
#pragma once
#include <type_traits>

#include "fmt_kernels.hpp"

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N) (immediate offsets of the inline-asm LDS reads)
template <int I0, int N, class F>
__device__ __forceinline__ void fh_static_for(F&& f) {
  if constexpr (I0 < N) {
    f(std::integral_constant<int, I0>{});
    fh_static_for<I0 + 1, N>(f);
  }
}

#ifndef RB_DMA_AUX
#define RB_DMA_AUX 0
#endif
// dynamic LDS of a tiling: the ring, or the fp32 output tile (row stride BN + 4) where that is larger
constexpr int fmt_rb_smem(int MI, int NJ, int KW, int NS) {
  const int ring = NS * KW * (2 * MI + 2 * NJ) * 1024, tile = 32 * MI * (32 * NJ + 4) * 4;
  return ring > tile ? ring : tile;
}

// SPECIALISED waves: 4 consumer waves (one per SIMD, a 2 x 2 grid of wave tiles of MI x NJ fragments, every k-block of a
// stage) + 4 loader waves that issue every LDS-DMA piece.  A symmetric form (round 5: all 8 waves issue their share of a
// stage right after its barrier, then multiply, two K-groups meeting in LDS) measured 0.475 us per stage against 0.33 us with
// the MFMAs removed - the issue of a stage's 28 pieces and its 24 MFMAs per SIMD added up: 12.5 us per fc1 launch against
// 10.0 for this form.  Here the matrix pipe of a SIMD belongs to its consumer wave while its loader wave issues, waits and
// arrives at the barrier.  Epilogue: the consumers leave the tile in LDS, all 8 waves convert and store it; the bias of a
// thread's columns (the same columns in every pass) is requested before the K loop.
// Measured and not kept: K walks skewed between the sharers of an operand tile (no change), rings of 3 / 5 / 6 stages (no
// change: the ring is not latency-bound), non-temporal DMA (+8 %), 4 waves that load and multiply (+40 %).
template <class T, int MI, int NJ, int KPS /* k-blocks per stage */, int NS, int EPI>
__global__ __launch_bounds__(512) void fmt_gemm_rbs_kernel(GemmArgs g) {
  constexpr int RT = 2 * MI, CT = 2 * NJ, NF = RT + CT, SF = KPS * NF, STAGE = SF * 1024, IPL = SF / 4;
  constexpr int ROWS = RT * 16, BN = CT * 16, NTHR = 512;
  static_assert(SF % 4 == 0, "the pieces of a stage must split evenly over the 4 loader waves");
  static_assert(NS >= 3 && NS <= 6, "ring of 3 to 6 stages");
  typedef typename T::elem E;
  typedef typename T::pack8 P8;
  extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];  // [NS][KPS][NF fragments][1 KiB]; then the output tile
#ifdef RB_STAMPS
#define RB_STAMP(k) do { if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(g.vout)[blockIdx.x * 4 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RB_STAMP(k) do { } while (0)
#endif
  RB_STAMP(0);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool loader = w >= 4;
  const int wr = (w >> 1) & 1, wc = w & 1, lw = w & 3;
  const int r16 = lane & 15, q = lane >> 4;
  int bx, by, ks = 0;
  {
    const int nbn = g.N / BN, id = (int)blockIdx.x;
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
        const int P = 8 / g.ksplit, x = bx & 7;
        ks = x / P;
        bx = (bx >> 3) * P + (x % P);
      } else {
        ks = bx / nbn;
        bx = bx % nbn;
      }
    }
  }
  const int m0 = by * ROWS, n0 = bx * BN;
  // this thread's columns in the epilogue passes (idx = threadIdx.x + pass * 512: the column group does not change with the pass)
  constexpr bool PACKED = (EPI == EPI_GELU_P16 || EPI == EPI_SILU_P16 || EPI == EPI_GELUERF_P16);
  constexpr int CG = BN / 8;
  static_assert(512 % CG == 0 && (512 / 64) % (BN / 32) == 0, "a thread keeps its column group across the epilogue passes");
  const int ecol = PACKED ? (((int)(threadIdx.x >> 6) % (BN / 32)) * 32 + (lane >> 4) * 8) : ((int)(threadIdx.x % CG) * 8);
  float4 bias0 = float4{0.f, 0.f, 0.f, 0.f}, bias1 = bias0;
  if constexpr (EPI != EPI_PARTIAL) {
    bias0 = *reinterpret_cast<const float4*>(g.bias + n0 + ecol);
    bias1 = *reinterpret_cast<const float4*>(g.bias + n0 + ecol + 4);
  }
  const int KB = g.K >> 5;
  const int KBs = (EPI == EPI_PARTIAL) ? KB / g.ksplit : KB;  // k-blocks of this workgroup's K slice
  const int S = KBs / KPS;                                    // stages
  const size_t tstride = (size_t)KB * 512;

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (loader) {
    // pieces of a stage: f = lw + 4 i -> k-block f / NF of the stage, fragment fi = f % NF (fi < RT: row tile of A)
    const u16* src[IPL];
#pragma unroll
    for (int i = 0; i < IPL; ++i) {
      const int f = lw + 4 * i, kk = f / NF, fi = f - kk * NF;
      src[i] = (fi < RT ? g.A + (size_t)(by * RT + fi) * tstride : g.W + (size_t)(bx * CT + fi - RT) * tstride) +
               (size_t)(ks * KBs + kk) * 512 + lane * 8;
    }
    auto issue = [&](int s) {
      unsigned char* dst = ring + (s % NS) * STAGE + lw * 1024;
#pragma unroll
      for (int i = 0; i < IPL; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)s * (KPS * 512)),
                                         (__attribute__((address_space(3))) void*)(dst + i * 4 * 1024), 16, 0, RB_DMA_AUX);
    };
#define RBS_WAIT_UPTO(K)                                     \
  do {                                                       \
    const int k_ = (K);                                      \
    if (NS >= 6 && k_ >= 4) fh_wait_vmcnt<4 * IPL>();        \
    else if (NS >= 5 && k_ >= 3) fh_wait_vmcnt<3 * IPL>();   \
    else if (NS >= 4 && k_ >= 2) fh_wait_vmcnt<2 * IPL>();   \
    else if (k_ >= 1) fh_wait_vmcnt<IPL>();                  \
    else fh_wait_vmcnt<0>();                                 \
  } while (0)
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (s < S) issue(s);
    if (S >= NS) fh_wait_vmcnt<(NS - 1) * IPL>();  // stage 0 (the two bias loads are older still)
    else RBS_WAIT_UPTO(S - 1);
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s + 1 < S; ++s) {
      // stage s + 1 has landed before barrier s: the consumers read it right after
      RBS_WAIT_UPTO(S - 2 - s);
      __builtin_amdgcn_s_barrier();
      if (s + NS < S) issue(s + NS);  // into the buffer of stage s, whose last k-block every consumer holds in registers
    }
#undef RBS_WAIT_UPTO
  } else {
    const unsigned abase = (unsigned)(lane * 16 + (wr * MI) * 1024);
    const unsigned bbase = (unsigned)(lane * 16 + (RT + wc * NJ) * 1024);
    // the consumer's unit is the k-block: the fragments of k-block kb + 1 are read into the second register set while the MFMAs
    // of k-block kb run; before the first k-block of stage s + 1 is read the wave passes barrier s (its S barriers pair with
    // the loaders': one before stage 0, one per stage boundary)
    u32x4 a0[MI], b0[NJ], a1[MI], b1[NJ];
    const int KBt = S * KPS;
    auto read_frags = [&](int kb, u32x4(&ar)[MI], u32x4(&br)[NJ]) {
      const int s = kb / KPS, kk = kb - s * KPS;
      const unsigned so = (unsigned)((s % NS) * STAGE + kk * (NF * 1024));
      fh_static_for<0, MI>([&](auto i) { ar[i.value] = fh_ds_read128<i.value * 1024>(abase + so); });
      fh_static_for<0, NJ>([&](auto j) { br[j.value] = fh_ds_read128<j.value * 1024>(bbase + so); });
    };
#define RBS_STEP(KB_, AC, BC, AN, BN_)                                                                 \
  do {                                                                                                 \
    const int nk_ = (KB_) + 1;                                                                         \
    if (nk_ < KBt) {                                                                                   \
      if (nk_ % KPS == 0) __builtin_amdgcn_s_barrier();                                                \
      read_frags(nk_, AN, BN_);                                                                        \
    }                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                     \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[i][j] = T::mfma(BC[j], AC[i], acc[i][j]);     \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
  } while (0)
    __builtin_amdgcn_s_barrier();
    read_frags(0, a0, b0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    RB_STAMP(1);
    for (int kb = 0; kb < KBt; kb += 2) {
      RBS_STEP(kb, a0, b0, a1, b1);
      if (kb + 1 < KBt) RBS_STEP(kb + 1, a1, b1, a0, b0);
    }
#undef RBS_STEP
  }

  // ---- the tile through LDS (the ring is idle: every DMA has landed, every fragment read has returned), row stride BN + 4
  constexpr int OSTR = BN + 4;
  static_assert(ROWS * OSTR * 4 <= fmt_rb_smem(MI, NJ, KPS, NS), "output staging does not fit the allocation");
  float* const so = reinterpret_cast<float*>(ring);
  __syncthreads();
  RB_STAMP(2);
  if (!loader) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        *reinterpret_cast<float4*>(so + ((wr * MI + i) * 16 + r16) * OSTR + (wc * NJ + j) * 16 + q * 4) =
            float4{acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
  }
  __syncthreads();
  unsigned rm = 0u;
  if constexpr (PACKED) {
    // packed for the consuming GEMM: 64 consecutive threads write one whole 1-KiB fragment (16 rows x 32 columns)
    constexpr int FR = RT * (BN / 32);
#pragma unroll
    for (int p = 0; p < (FR * 64 + NTHR - 1) / NTHR; ++p) {
      const int fr = (int)(threadIdx.x >> 6) + p * 8, t = fr / (BN / 32);
      const int r = t * 16 + (lane & 15), row = m0 + r;
      if (fr >= FR || row >= g.M) continue;
      const float4 x0 = *reinterpret_cast<const float4*>(so + r * OSTR + ecol), x1 = *reinterpret_cast<const float4*>(so + r * OSTR + ecol + 4);
      const float v[8] = {x0.x + bias0.x, x0.y + bias0.y, x0.z + bias0.z, x0.w + bias0.w, x1.x + bias1.x, x1.y + bias1.y, x1.z + bias1.z, x1.w + bias1.w};
      P8 u;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = v[e];
        if constexpr (EPI == EPI_SILU_P16) x = fh_silu(x);
        if constexpr (EPI == EPI_GELU_P16) x = fh_gelu_tanh(x);
        if constexpr (EPI == EPI_GELUERF_P16) x = fh_gelu_erf(x);
        T::set(u, e, x);
      }
      fh_track_pack<T>(rm, u);
      T::store8(reinterpret_cast<E*>(g.out16) + fmt_pack_off(row, n0 + ecol, g.ldo16), u);
    }
  } else {
#pragma unroll
    for (int p = 0; p < (ROWS * CG + NTHR - 1) / NTHR; ++p) {
      const int r = (int)(threadIdx.x / CG) + p * (NTHR / CG), row = m0 + r;
      if (r >= ROWS || row >= g.M) continue;
      const float4 x0 = *reinterpret_cast<const float4*>(so + r * OSTR + ecol), x1 = *reinterpret_cast<const float4*>(so + r * OSTR + ecol + 4);
      const float v[8] = {x0.x + bias0.x, x0.y + bias0.y, x0.z + bias0.z, x0.w + bias0.w, x1.x + bias1.x, x1.y + bias1.y, x1.z + bias1.z, x1.w + bias1.w};
      if constexpr (EPI == EPI_F32 || EPI == EPI_PARTIAL) {
        float* o = g.out_f32 + (size_t)ks * g.slab_stride + (size_t)row * g.ldo + n0 + ecol;
        *reinterpret_cast<float4*>(o) = float4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<float4*>(o + 4) = float4{v[4], v[5], v[6], v[7]};
      } else {
        static_assert(EPI == EPI_T16, "epilogue not built for the row-blocked GEMM");
        P8 u;
#pragma unroll
        for (int e = 0; e < 8; ++e) T::set(u, e, v[e]);
        fh_track_pack<T>(rm, u);
        T::store8(reinterpret_cast<E*>(g.out16) + (size_t)row * g.ldo16 + n0 + ecol, u);
      }
    }
  }
  if constexpr (EPI != EPI_F32 && EPI != EPI_PARTIAL) fh_range_flush<T>(g.sat, rm);
#ifdef RB_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  RB_STAMP(3);
#endif
#undef RB_STAMP
}
