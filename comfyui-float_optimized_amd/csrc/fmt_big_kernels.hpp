// The FMT's large GEMM - the adaLN projection of every evaluation of a window (FMT.py:163-166, 187-190: M = evaluations x rows
// = 9 000 per clip, N = 51 200, K = 1 024, fp32 modulations out) - as a PERSISTENT kernel on 192 x 256 tiles with ONE wave per
// SIMD (round 5; the 8-wave forms it came from are lab kernels: tools/probes/fmt_big_kernels.hpp):
//   * rows DENSE over the evaluations (47 row blocks of 192 instead of 50 evaluations padded from 180 to 192 rows: -6 %);
//   * one workgroup of 4 waves per CU walks its share of the tile list; the LDS-DMA ring (4 stages of 28 KiB) never stops at a
//     tile boundary: the first operands of tile t + 1 land while tile t is being stored;
//   * wave tiles of 96 x 128 (6 x 8 fragments: 192 accumulator registers, the whole register file of a SIMD for one wave): 14
//     fragment reads per 48 MFMAs instead of the 20 of two 96 x 64 tiles; 7 LDS-DMA pieces per wave and k-block exactly; every
//     piece and every fragment read is placed between MFMAs; control flow carried in registers (a CU has one scalar unit);
//   * the epilogue does not touch the ring: 16 rows x 64 columns at a time through a wave-PRIVATE 4-KB LDS slab (XOR-swizzled
//     granules, no workgroup barrier), whole 256-byte runs per NON-TEMPORAL store (the 1.8 GB of modulations of a launch stream past
//     the caches); stores and the bias loads are counted by hand in the vmcnt waits - the kernel must compile
//     WITHOUT scratch (a reload is a vector-memory operation);
//   * tile order: an XCD's 32 workgroups hold 8 row blocks x 4 column blocks at any time (82 % L2 hits).
// Bitwise the numbers of fmt_gemm_dma_kernel (the same MFMA sequence per output element, one bias add).  MI355X, 9 000 x 51 200
// x 1 024, one launch every 15 ms: 842 us (943 with `sc1` stores) against 1 100 for fmt_gemm_dma_kernel; sustained loops 885 (934)
// against 1 083, at the socket's 1 400-W cap (1.97 GHz); without its stores 746 us = 0.51 of 2.5 PFLOP/s.
#pragma once
#include "fmt_rb_kernels.hpp"

struct BigArgs {
  const u16* A;       // packed [row tiles][KB][64][8], rows dense, readable up to nrb row blocks
  const u16* W;       // packed [N/16][KB][64][8]
  const float* bias;  // [N]
  float* out;         // [M][ldo] fp32
  int M, N, K, ldo;
  int nrb, ncb;       // row blocks (192 or 256 rows: the kernel's MI) / column blocks of 256 (ncb a multiple of 8)
};

template <class T, int NS>
__global__ __launch_bounds__(256) void fmt_gemm_big4_kernel(BigArgs g) {
  constexpr int MI = 6, NJ = 8, RT = 12, CT = 16, NF = RT + CT, STAGE = NF * 1024, ROWS = RT * 16;
  constexpr int NP = NF / 4;          // 7 pieces per wave and stage
  constexpr int NST = MI * 8 + 2;     // an epilogue: MI x 2 halves x 4 row stores + the next tile's two bias loads
  static_assert(NS == 4, "ring of 4 stages");
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];  // [NS][28 fragments][1 KiB] ring | [4 waves][4 KiB] staging
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = w >> 1, wc = w & 1, r16 = lane & 15, q = lane >> 4;
  const int KB = g.K >> 5;
  const size_t tstride = (size_t)KB * 512;
  const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int nc8 = g.ncb >> 3;
  const int per_grp = 4 * g.nrb, full = nc8 >> 2;
  const int nent = nc8 * g.nrb;
  const int ntl = slot < nent ? (nent - slot + nslot - 1) / nslot : 0;
  if (ntl == 0) return;
  auto tile_of = [&](int j, int& rb, int& cb) {
    const int e = slot + j * nslot;
    int grp = e / per_grp, rem = e - grp * per_grp, ncol = 4;
    if (grp >= full) {
      grp = full;
      rem = e - full * per_grp;
      ncol = nc8 - full * 4;
    }
    rb = rem / ncol;
    cb = x + 8 * (grp * 4 + (rem - rb * ncol));
  };
  const u16* src[NP];
  auto set_tile_ptrs = [&](int j) {
    int rb, cb;
    tile_of(j, rb, cb);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int f = w + 4 * i;
      src[i] = f < RT ? g.A + (size_t)(rb * RT + f) * tstride : g.W + (size_t)(cb * CT + f - RT) * tstride;  // wave-uniform
    }
  };
  int ij = 0, ikb = 0;
  unsigned ioff = (unsigned)(w * 1024);
  set_tile_ptrs(0);
  auto issue_piece = [&](int i) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + lane * 8),
                                     (__attribute__((address_space(3))) void*)(lds + ioff + i * 4 * 1024), 16, 0, 0);
  };
  auto issue_advance = [&]() {
#pragma unroll
    for (int i = 0; i < NP; ++i) src[i] += 512;
    ioff += STAGE;
    if (ioff >= (unsigned)(NS * STAGE)) ioff -= (unsigned)(NS * STAGE);
    if (++ikb == KB) {
      ikb = 0;
      ij = min(ij + 1, ntl - 1);
      set_tile_ptrs(ij);
    }
  };
  struct Bias2 {
    f32x4 v[2];
  };
  auto load_bias = [&](int j, Bias2& b) {
    int rb, cb;
    tile_of(min(j, ntl - 1), rb, cb);
    const float* p = g.bias + cb * 256 + wc * 128 + (lane & 15) * 4;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b.v[0]) : "v"(p) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off offset:256" : "=v"(b.v[1]) : "v"(p) : "memory");
  };
  const unsigned abase = (unsigned)(lane * 16 + (wr * MI) * 1024), bbase = (unsigned)(lane * 16 + (RT + wc * NJ) * 1024);
  f32x4 acc[MI][NJ];
  u32x4 a0[MI], b0[NJ], a1[MI], b1[NJ];
  unsigned roff = 0u;
  constexpr int WCAP = 63;  // vmcnt is a 6-bit counter: a wait for fewer outstanding operations than necessary is always safe
#define BIG4_STEP(EXTRA, AC, BC, AN, BN_)                                                             \
  do {                                                                                                \
    if (EXTRA) fh_wait_vmcnt<((NS - 2) * NP + NST < WCAP ? (NS - 2) * NP + NST : WCAP)>();            \
    else fh_wait_vmcnt<(NS - 2) * NP>();                                                              \
    __builtin_amdgcn_s_barrier();                                                                     \
    fh_static_for<0, MI * NJ>([&](auto m) {                                                           \
      constexpr int r_ = m.value / 3;                                                                 \
      if constexpr (m.value % 3 == 0 && r_ < MI + NJ) {                                               \
        if constexpr (r_ < NJ) BN_[r_] = fh_ds_read128<r_ * 1024>(bbase + roff);                      \
        else AN[r_ - NJ] = fh_ds_read128<(r_ - NJ) * 1024>(abase + roff);                             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
      }                                                                                               \
      constexpr int d_ = (m.value - 1) / 6;                                                           \
      if constexpr (m.value >= 1 && (m.value - 1) % 6 == 0 && d_ < NP) {                              \
        issue_piece(d_);                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
      }                                                                                               \
      acc[m.value / NJ][m.value % NJ] = T::mfma(BC[m.value % NJ], AC[m.value / NJ], acc[m.value / NJ][m.value % NJ]); \
      __builtin_amdgcn_sched_barrier(0);                                                              \
    });                                                                                               \
    issue_advance();                                                                                  \
    roff += STAGE;                                                                                    \
    if (roff >= (unsigned)(NS * STAGE)) roff -= (unsigned)(NS * STAGE);                               \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  } while (0)

  Bias2 bias_cur;
  load_bias(0, bias_cur);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int s = 0; s < NS; ++s) {
#pragma unroll
    for (int i = 0; i < NP; ++i) issue_piece(i);
    issue_advance();
  }
  fh_wait_vmcnt<(NS - 1) * NP>();
  __builtin_amdgcn_s_barrier();
  fh_static_for<0, NJ>([&](auto j) { b0[j.value] = fh_ds_read128<j.value * 1024>(bbase + roff); });
  fh_static_for<0, MI>([&](auto i) { a0[i.value] = fh_ds_read128<i.value * 1024>(abase + roff); });
  roff += STAGE;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  const unsigned stg_lds = (unsigned)(NS * STAGE + w * 4096);
  bool extra = false;
  for (int j = 0; j < ntl; ++j) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    BIG4_STEP(extra, a0, b0, a1, b1);
    BIG4_STEP(extra, a1, b1, a0, b0);
    BIG4_STEP(extra, a0, b0, a1, b1);
    BIG4_STEP(false, a1, b1, a0, b0);
    for (int kb = 4; kb < KB; kb += 2) {
      BIG4_STEP(false, a0, b0, a1, b1);
      BIG4_STEP(false, a1, b1, a0, b0);
    }
    int rb, cb;
    tile_of(j, rb, cb);
    const int row0 = rb * ROWS + wr * (MI * 16), col0 = cb * 256 + wc * 128;
    const bool whole = row0 + MI * 16 <= g.M;
    typedef __attribute__((address_space(1))) char gchar;
    gchar* obase = (gchar*)(g.out + (size_t)row0 * g.ldo + col0);
    const unsigned ovoff = (unsigned)((q * g.ldo + (lane & 15) * 4) * 4);
    const size_t ostep = (size_t)16 * g.ldo;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      f32x4 v[2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const unsigned ad = stg_lds + (unsigned)(r16 * 256 + (((jj * 4 + q) ^ r16) << 4));
          asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(acc[i][h * 4 + jj]) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int r = p * 4 + q;
          const unsigned ad = stg_lds + (unsigned)(r * 256 + (((lane & 15) ^ r) << 4));
          asm volatile("ds_read_b128 %0, %1" : "=v"(v[h][p]) : "v"(ad) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int r = i * 16 + p * 4 + q;
        if (whole || row0 + r < g.M) {
          const f32x4 o0 = v[0][p] + bias_cur.v[0], o1 = v[1][p] + bias_cur.v[1];
#if defined(BIG_NO_STORE)  // timing-only (tools/probes/gemm_big_lab.hip)
          asm volatile("" ::"v"(o0), "v"(o1));
#else
          // non-temporal: the 1.8 GB of modulations of a launch are next read by the step chain's LayerNorms, evaluations later;
          // streamed past the caches they leave the operand tiles (105 MB of weights, the activation rows) where they are.
          // MI355X, one launch every 15 ms: 842 us against 954 with `sc1` (write-through, kept out of the L2s only), 970 plain
#ifdef BIG4_PLAIN_STORE
          asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(ovoff), "v"(o0), "s"(obase) : "memory");
          asm volatile("global_store_dwordx4 %0, %1, %2 offset:256" ::"v"(ovoff), "v"(o1), "s"(obase) : "memory");
#else
          asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(ovoff), "v"(o0), "s"(obase) : "memory");
          asm volatile("global_store_dwordx4 %0, %1, %2 offset:256 nt" ::"v"(ovoff), "v"(o1), "s"(obase) : "memory");
#endif
#endif
        }
        obase += ostep;
        asm volatile("" : "+s"(obase));
      }
    }
    load_bias(j + 1, bias_cur);  // 2 of the NST operations, behind the stores
#ifdef BIG4_SAFE_EPI  // diagnostic: no store may be outstanding when the next tile's counted waits begin
    if (true) {
#else
    if (!whole) {
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      extra = false;
    } else {
      extra = true;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef BIG4_STEP
}
