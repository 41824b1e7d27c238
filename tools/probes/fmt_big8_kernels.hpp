// LAB ONLY (tools/probes/gemm_big_lab.hip) - a measured negative result of round 5, not part of the library.
//
// The FMT's large GEMM - the adaLN projection of every evaluation of a window (FMT.py:163-166, 187-190: M = evaluations x rows
// = 9 000 per clip, N = 51 200, K = 1 024, fp32 modulations out) - as a PERSISTENT kernel on 192 x 256 tiles, written to take
// what fmt_gemm_dma_kernel (192 x 320 tiles, one workgroup per tile, 0.36 of the dense MFMA peak) loses OUTSIDE its K loop:
//   * rows DENSE over the evaluations (47 row blocks of 192 instead of 50 evaluations padded from 180 to 192 rows: -6 %);
//   * one workgroup per CU walks its share of the tile list and the LDS-DMA ring never stops at a tile boundary: the first
//     operands of tile t + 1 land while tile t is being stored (no ~1 us wait per tile);
//   * the epilogue does not touch the ring: a wave passes its 96 x 64 accumulators 16 rows at a time through a PRIVATE 4-KB LDS
//     slab (XOR-swizzled granules, conflict-free, no workgroup barrier) and stores whole 256-byte rows; the bias arrives by one
//     register load per tile; stores and that load are counted by hand in the vmcnt waits (the kernel must compile without
//     spills: scratch reloads are vector-memory operations);
//   * tile order: an XCD's 32 workgroups hold 8 row blocks x 4 column blocks at any time (FETCH_SIZE: 1.55 GB of the 8.4 GB of
//     tile fetches of a launch leave the L2s: 82 % hits).
// MEASURED (MI355X, uniform random operands; DESIGN.md sections 6 / 7, profiles/r05_clock_power.txt):
//   * first form: 1 094-1 110 us per launch against 1 232-1 273 for fmt_gemm_dma_kernel on the same box; IN THE PIPELINE a tie (FMT
//     sampling 81.6 vs 81.3 ms per clip, 161.5 vs 160.3 per 4 clips, 469 vs 460 per 16);
//   * its K loop spent a third of its time in the scalar unit (integer divisions and branch chains per step): LEAN control below;
//   * sustained loops sit at the socket's 1 400-W cap and 1.92-1.97 GHz whatever the issue order (exact piece counts, pieces and
//     fragment reads spread between the MFMAs - BIG_INTERLEAVE / BIG_DMA_EVERY -, SIMD partners in opposite phases - BIG_PHASE);
//   * 945 us sustained (8 waves, this kernel) / 934 (fmt_gemm_big4_kernel below: one wave per SIMD, 96 x 128 wave tiles; 746 =
//     0.51 of 2.5 PFLOP/s without its stores) against 1 083 for the library kernel.  256-row tiles of the 8-wave form (MI = 8) spill.
#pragma once
#include "fmt_big_kernels.hpp"  // -I comfyui-float_optimized_amd/csrc BEFORE -I tools/probes: BigArgs and fmt_gemm_big4_kernel (the library kernel)

#ifndef BIG_SPREAD
#define BIG_SPREAD 1  // 1: a wave issues its LDS-DMA pieces in front of MFMA chunk `wave index` of the step; 0: all waves behind the barrier
#endif
#ifdef BIG_NO_MFMA  // timing-only build: the fetch, the fragment reads and the barriers alone
#define BIG_MFMA(m, AC, BC) asm volatile("" ::"v"(AC[(m) / NJ]), "v"(BC[(m) % NJ]))
#else
#define BIG_MFMA(m, AC, BC) acc[(m) / NJ][(m) % NJ] = T::mfma(BC[(m) % NJ], AC[(m) / NJ], acc[(m) / NJ][(m) % NJ])
#endif
#ifndef BIG_INTERLEAVE
#define BIG_INTERLEAVE 0  // > 0: one fragment read of the next stage in front of every BIG_INTERLEAVE-th MFMA
#endif
#ifndef BIG_DMA_EVERY
#define BIG_DMA_EVERY 0  // > 0 (with BIG_INTERLEAVE > 0): LDS-DMA piece d of the step in front of MFMA 1 + d * BIG_DMA_EVERY instead of all behind the barrier
#endif
#ifndef BIG_PHASE
#define BIG_PHASE 0
#endif
#define BIG_ISSUE_ALWAYS()                                          \
  do {                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < NP; ++i_) issue_piece(i_); \
    issue_advance();                                                \
  } while (0)
#ifdef BIG_NO_DMA   // timing-only: no LDS-DMA in the loop (the counted waits find nothing to wait for)
#define BIG_ISSUE() do { } while (0)
#define BIG_ISSUE_PIECE(d) do { } while (0)
#define BIG_ISSUE_ADVANCE() do { } while (0)
#else
#define BIG_ISSUE() BIG_ISSUE_ALWAYS()
#define BIG_ISSUE_PIECE(d) issue_piece(d)
#define BIG_ISSUE_ADVANCE() issue_advance()
#endif
#ifdef BIG_NO_READS  // timing-only: no fragment reads in the loop
#define BIG_READS(AN, BN_) do { } while (0)
#else
#define BIG_READS(AN, BN_) read_next(AN, BN_)
#endif

template <class T, int MI /* row fragments per wave: 6 -> 192-row tiles, 8 -> 256-row tiles */, int NS>
__global__ __launch_bounds__(512) void fmt_gemm_big_kernel(BigArgs g) {
  constexpr int NJ = 4, RT = 2 * MI, CT = 16, NF = RT + CT, STAGE = NF * 1024, ROWS = RT * 16;
  constexpr int IMAX = (NF + 7) / 8, ILO = NF / 8, NHI = NF % 8;  // pieces per wave and stage: IMAX for waves < NHI
  constexpr int NST = MI * 4 + 1;  // vector-memory operations of an epilogue: MI x 4 row stores + the next tile's bias load
  static_assert(NS >= 3 && NS <= 4, "ring of 3 or 4 stages");
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];  // [NS][32 fragments][1 KiB] ring | [8 waves][4 KiB] staging
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = w >> 2, wc = w & 3, r16 = lane & 15, q = lane >> 4;
  const int KB = g.K >> 5;
  const size_t tstride = (size_t)KB * 512;
  // ---- this workgroup's tiles: entries slot, slot + nslot, ... of its XCD's list
  const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int nc8 = g.ncb >> 3;                       // column blocks of this XCD: x + 8 c
  const int per_grp = 4 * g.nrb, full = nc8 >> 2;   // entries of a full group of 4 column blocks
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
  const int Stot = ntl * KB;  // stages of this workgroup

  // ---- LEAN control (round 5, late): the first form of this loop recomputed (tile, k-block) from a global stage number in every
  // step - integer divisions, 64-bit multiplies, a chain of run-time branches for the wait counts: 1 213 scalar instructions and
  // 63 branches per step and wave, and with EVERYTHING else removed (no DMA, no reads, no MFMA, no stores) the kernel still took
  // 607 of its ~1 100 us: a CU has ONE scalar unit for its 8 waves.  Here every per-step quantity is carried along: source
  // pointers advance by 1 KiB, ring slots wrap by compare, the wait counts are immediates (the DMA never stops: behind the last
  // real stage the last tile is issued again, so the counts hold to the end; every wave issues IMAX pieces, a duplicate where
  // the stage has fewer), and the tile switch is one rarely taken branch.
  const u16* src[IMAX];
  auto set_tile_ptrs = [&](int j) {
    int rb, cb;
    tile_of(j, rb, cb);
#pragma unroll
    for (int i = 0; i < IMAX; ++i) {
      const int f = min(w + 8 * i, NF - 1);
      src[i] = f < RT ? g.A + (size_t)(rb * RT + f) * tstride : g.W + (size_t)(cb * CT + f - RT) * tstride;  // wave-uniform
    }
  };
  int ij = 0, ikb = 0;
  unsigned ioff = (unsigned)(w * 1024);  // LDS byte offset of this wave's first piece in the slot being filled
  set_tile_ptrs(0);
  // this wave's pieces of a stage: NP = IMAX for waves < NHI, ILO for the others
  auto issue_piece = [&](int i) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + lane * 8),
                                     (__attribute__((address_space(3))) void*)(lds + ioff + i * 8 * 1024), 16, 0, 0);
  };
  auto issue_advance = [&]() {
#pragma unroll
    for (int i = 0; i < IMAX; ++i) src[i] += 512;
    ioff += STAGE;
    if (ioff >= (unsigned)(NS * STAGE)) ioff -= (unsigned)(NS * STAGE);
    if (++ikb == KB) {
      ikb = 0;
      ij = min(ij + 1, ntl - 1);  // behind the last tile: the last tile again (never read)
      set_tile_ptrs(ij);
    }
  };
  // bias of this wave's 64 columns, 4 per lane (lane % 16), by a register load the compiler does not see (no wait of its own)
  // (loaded INTO the loop-carried registers behind the last use of the old value: a copy of a freshly loaded value may be emitted
  // as a register move in front of which the compiler, not knowing that the asm is an asynchronous load, puts no wait)
  auto load_bias = [&](int j, f32x4& v) {
    int rb, cb;
    tile_of(min(j, ntl - 1), rb, cb);
    const float* p = g.bias + cb * 256 + wc * 64 + (lane & 15) * 4;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  };

  const unsigned abase = (unsigned)(lane * 16 + (wr * MI) * 1024), bbase = (unsigned)(lane * 16 + (RT + wc * NJ) * 1024);
  f32x4 acc[MI][NJ];
  u32x4 a0[MI], b0[NJ], a1[MI], b1[NJ];
  unsigned roff = 0u;  // LDS byte offset of the slot to read next
  auto read_next = [&](u32x4(&ar)[MI], u32x4(&br)[NJ]) {
    fh_static_for<0, MI>([&](auto i) { ar[i.value] = fh_ds_read128<i.value * 1024>(abase + roff); });
    fh_static_for<0, NJ>([&](auto j) { br[j.value] = fh_ds_read128<j.value * 1024>(bbase + roff); });
    roff += STAGE;
    if (roff >= (unsigned)(NS * STAGE)) roff -= (unsigned)(NS * STAGE);
  };
  auto run = [&](auto np_c) {
  constexpr int NP = decltype(np_c)::value;
  // one step: stage s + 1 has landed (behind it: NS - 2 younger stages, and - EXTRA - the last epilogue's NST operations)
#define BIG_STEP(EXTRA, AC, BC, AN, BN_)                                                        \
  do {                                                                                          \
    if (EXTRA) fh_wait_vmcnt<(NS - 2) * NP + NST>();                                          \
    else fh_wait_vmcnt<(NS - 2) * NP>();                                                      \
    __builtin_amdgcn_s_barrier();                                                               \
    if (BIG_INTERLEAVE == 0 || BIG_DMA_EVERY == 0) BIG_ISSUE();                                 \
    if (BIG_INTERLEAVE == 0) {                                                                  \
      BIG_READS(AN, BN_);                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                        \
      _Pragma("unroll") for (int m = 0; m < MI * NJ; ++m) BIG_MFMA(m, AC, BC);                  \
    } else {                                                                                    \
      /* the MI + NJ fragment reads of stage s + 1 one by one between the MFMAs of stage s (every BIG_INTERLEAVE MFMAs):   */ \
      /* 8 waves that all read first fill the LDS queue, and the last of them reaches its MFMAs hundreds of clocks late    */ \
      fh_static_for<0, MI * NJ>([&](auto m) {                                                   \
        constexpr int il_ = BIG_INTERLEAVE > 0 ? BIG_INTERLEAVE : 1, r_ = m.value / il_;                                            \
        if constexpr (m.value % il_ == 0 && r_ < MI + NJ) {                          \
          if constexpr (r_ < NJ) BN_[r_] = fh_ds_read128<r_ * 1024>(bbase + roff);              \
          else AN[r_ - NJ] = fh_ds_read128<(r_ - NJ) * 1024>(abase + roff);                     \
          __builtin_amdgcn_sched_barrier(0);                                                    \
        }                                                                                       \
        if constexpr (BIG_DMA_EVERY > 0) {                                                      \
          /* LDS-DMA piece d in front of MFMA 1 + d * BIG_DMA_EVERY: a piece's issue blocks its wave while the CU's intake   */ \
          /* queue is full (~27 clocks per KiB), and pieces issued first put that in front of EVERY wave's MFMAs             */ \
          /* BIG_PHASE: the two waves of a SIMD (w and w + 4: NP = IMAX and ILO) issue in opposite halves     */ \
          constexpr int de_ = BIG_DMA_EVERY > 0 ? BIG_DMA_EVERY : 1;                            \
          constexpr int m0_ = (BIG_PHASE && NP != IMAX) ? MI * NJ / 2 + 1 : 1, d_ = (m.value - m0_) / de_; \
          if constexpr (m.value >= m0_ && (m.value - m0_) % de_ == 0 && d_ < NP) {              \
            BIG_ISSUE_PIECE(d_);                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                  \
          }                                                                                     \
        }                                                                                       \
        BIG_MFMA(m.value, AC, BC);                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                      \
      });                                                                                       \
      if (BIG_DMA_EVERY > 0) BIG_ISSUE_ADVANCE();                                               \
      roff += STAGE;                                                                            \
      if (roff >= (unsigned)(NS * STAGE)) roff -= (unsigned)(NS * STAGE);                       \
    }                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                          \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)

  f32x4 bias_cur;
  load_bias(0, bias_cur);  // older than every DMA
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int s = 0; s < NS; ++s) BIG_ISSUE_ALWAYS();
  fh_wait_vmcnt<(NS - 1) * NP>();
  __builtin_amdgcn_s_barrier();
  read_next(a0, b0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  const unsigned stg_lds = (unsigned)(NS * STAGE + w * 4096);  // this wave's private slab: [16 rows][16 granules of 16 B], granule ^ row
  bool extra = false;  // the previous tile's epilogue left NST operations in the counter
  for (int j = 0; j < ntl; ++j) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the first NS - 1 steps behind an epilogue count its operations as younger than the stage they wait for
    static_assert(NS == 4, "the peeled steps below are written for a ring of 4");
    BIG_STEP(extra, a0, b0, a1, b1);
    BIG_STEP(extra, a1, b1, a0, b0);
    BIG_STEP(extra, a0, b0, a1, b1);
    BIG_STEP(false, a1, b1, a0, b0);
    for (int kb = 4; kb < KB; kb += 2) {
      BIG_STEP(false, a0, b0, a1, b1);
      BIG_STEP(false, a1, b1, a0, b0);
    }
    // ---- epilogue of tile j from the accumulators; the ring keeps filling with tile j + 1 meanwhile
    int rb, cb;
    tile_of(j, rb, cb);
    const int row0 = rb * ROWS + wr * (MI * 16), col0 = cb * 256 + wc * 64;
    const bool whole = row0 + MI * 16 <= g.M;   // every store of this wave is issued: the hand count of NST holds
    // running output address = scalar base (4 rows further per store) + this lane's 32-bit offset.  The base is kept opaque:
    // the compiler would otherwise precompute one 64-bit address per store at kernel start and spill them - and scratch
    // reloads are vector-memory operations that would break the hand count of NST.  The kernel must compile WITHOUT spills.
    typedef __attribute__((address_space(1))) char gchar;  // global address space kept explicit: behind the opaque asm the
    typedef __attribute__((address_space(1))) f32x4 gf32x4;  // compiler would emit FLAT stores (out of order, two counters)
    gchar* obase = (gchar*)(g.out + (size_t)row0 * g.ldo + col0);  // wave-uniform, advanced per store
    const unsigned ovoff = (unsigned)((q * g.ldo + (lane & 15) * 4) * 4);                      // this lane's byte offset in a 4-row group
    const size_t ostep = (size_t)16 * g.ldo;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      // 16 rows x 64 columns through the slab: write in accumulator order (row r16, granule jj * 4 + q), read back by rows
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const unsigned ad = stg_lds + (unsigned)(r16 * 256 + (((jj * 4 + q) ^ r16) << 4));
        asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(acc[i][jj]) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      f32x4 v[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int r = p * 4 + q;  // lane -> row p * 4 + lane / 16, granule lane % 16
        const unsigned ad = stg_lds + (unsigned)(r * 256 + (((lane & 15) ^ r) << 4));
        asm volatile("ds_read_b128 %0, %1" : "=v"(v[p]) : "v"(ad) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int r = i * 16 + p * 4 + q;
        if (whole || row0 + r < g.M) {
          f32x4 o = v[p] + bias_cur;
#if defined(BIG_NO_STORE)
          asm volatile("" ::"v"(o));
#elif defined(BIG_STORE_PLAIN)
          *(gf32x4*)(obase + ovoff) = o;
#elif defined(BIG_STORE_NT)
          __builtin_nontemporal_store(o, (gf32x4*)(obase + ovoff));
#else
          // write-through and not kept in the XCD's L2 (sc1): the 1.8 GB of modulations of a launch would otherwise pass through
          // the 4-MB L2s as dirty lines and push the operand tiles out
          asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(ovoff), "v"(o), "s"(obase) : "memory");
#endif
        }
        obase += ostep;
        asm volatile("" : "+s"(obase));
      }
    }
    load_bias(j + 1, bias_cur);  // 1 of the NST operations, behind the stores; used by the next epilogue
    if (!whole) {  // fewer stores than NST may have been issued: start the count afresh (the prefetched stages land too)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      extra = false;
    } else {
      extra = true;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the re-issued stages behind the last tile: no LDS-DMA may outlive the workgroup
  };
  // two code paths: waves < NHI issue IMAX pieces per stage, the others ILO (a surplus piece would land in the NEXT stage's slot)
  if (NHI == 0 || w < NHI) run(std::integral_constant<int, IMAX>{});
  else run(std::integral_constant<int, ILO>{});
#undef BIG_STEP
}

