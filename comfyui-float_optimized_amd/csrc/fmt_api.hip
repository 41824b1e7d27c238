// C-ABI entry points of the FMT operator (include/float_hip.h) and the host-side launch chain.
#include <math.h>
#include <stdlib.h>

#include "fmt_gemm.hpp"
#include "fmt_kernels.hpp"
#include "fmt_rb_kernels.hpp"
#include "fmt_big_kernels.hpp"

namespace {

typedef FmtLin Lin;

struct Blk {
  Lin qkv, proj, fc1, fc2;
};

constexpr int kMaxTok = 80;      // tokens per window (n_prev + n_cur); the workspace is sized for a 4-way CFG batch of them (320 rows)
constexpr int kMaxSteps = 1024;  // FloatAdvancedParameters.nfe max is 1000 (nodes_adv.py:184-190)

}  // namespace

struct float_fmt {
  float_fmt_cfg_t cfg;
  DevicePool pool;
  int D, ntok, Mpad, Kc, Ntot, Kx;
  int Bmax = 1;  // clips per launch chain the workspace is sized for (float_fmt_cfg_t::max_batch)
  Lin x_embed, t0, t2, c_embed, adaln_all, final_lin;
  std::vector<Blk> blk;
  float* pos = nullptr;
  float* freqs = nullptr;
  // workspace
  u16 *cond16, *sc16, *h16, *hfin16, *qkv16, *att16, *hid16, *xin16, *tsin16, *th16;
  float *ccond, *xres, *xcur, *temb, *vout;
  unsigned long long* sat = nullptr;  // range counter of every 16-bit activation store of the handle's launches (float_fmt_saturation)
  float* slab = nullptr;  // [8][Mpad][D] split-K partial sums (EPI_PARTIAL; the fused attention + proj launch writes one slab per head)
  float *wa_c, *we_c, *prev_x, *prev_wa, *prev_we, *x0_c;
  int method = 0;          // FLOAT_ODE_*
  int attnproj = 0;        // heads per workgroup of the fused attention + proj launch (FLOAT_FMT_ATTNPROJ), 0 = two launches
  // the step chain of one evaluation as ONE persistent kernel (fmt_mega_kernel): stage table per CFG shape, barrier words
  int mega_on = 0;         // FLOAT_FMT_MEGA=1 selects it; default 0 = the 59-launch chain (faster: fmt_kernels.hpp, fmt_mega_kernel)
  int n_cu = 0;
  struct MegaPlan {
    MegaStage* dev = nullptr;
    int nstage = 0, bc = 0, tried = 0;
    MegaCtx ctx{};
  } mega[5];               // by CFG rows (1, 3, 4)
  unsigned* mega_sync = nullptr;  // [8 x 32 | 32 | 8 x 32 | seq | err] words
  unsigned* mega_err_host = nullptr;  // host-mapped twin of the err word (hipHostMalloc): read by mega_poll without a copy
  u16* mega_ws = nullptr;         // write-once A operands of the persistent kernel: per block h16 x 2, att16, hid16; + the head's
  float* kbuf = nullptr;   // [4][kMaxTok][dim_w] stage velocities of the Runge-Kutta solvers
  // Modulations of up to kScSteps evaluations of a window, [step][Mmod][Ntot] fp32: c = t_emb + c_embedder(wr, wa, we) does not
  // depend on x (FMT.py:333-335, 163-166), so every adaLN projection of those evaluations is ONE GEMM per window.
  float* modall = nullptr;
  int Mmod = 0;
  size_t mod_zs = 0;  // floats between two evaluations' modulations in modall, as the last run_mod_all laid them out
  // hipGraph cache for the per-window chain, keyed by (nfe, bc, we_len, method, scales); least recently used entry evicted
  struct GraphKey {
    int nfe, bc, we_len, method, nclip;
    int prio;  // priority of the stream the graph is launched on (run_mod_all picks its kernel by it)
    float a, r, e;
    bool operator==(const GraphKey& o) const { return memcmp(this, &o, sizeof(GraphKey)) == 0; }
  };
  struct GraphEntry {
    GraphKey key;
    hipGraphExec_t exec;
    uint64_t used;
  };
  std::vector<GraphEntry> graphs;
  uint64_t graph_clock = 0;
  hipStream_t cap_stream = nullptr;
  int cap_prio = 0;  // while capturing: the priority of the stream the graph will be launched on
  // state of an incremental sample (float_fmt_sample_begin / _next)
  struct {
    const float *wr, *wa, *we, *noise;
    float* r_d;
    int T, we_len, nfe, include_r, next, n_chunks, B;
    int first = 0, total = 0;                                  // first window of the job / windows of the whole clip
    const float *hist_x = nullptr, *hist_wa = nullptr, *hist_we = nullptr;  // history of window `first` (nullptr: zeros)
    float a, r, e;
    std::vector<float> ts;
    bool active = false;
  } job;
};

namespace {

int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Stream-ordered device copies as kernel launches (see fmt_copy_kernel: memcpy / memset nodes of a caller's capture did not
// replay reproducibly).
int dev_copy2d(float* dst, size_t dpitch, const float* src, size_t spitch, int width, int rows, hipStream_t s) {
  if (width <= 0 || rows <= 0) return FLOAT_OK;
  hipLaunchKernelGGL(fmt_copy_kernel, dim3((unsigned)(((size_t)rows * width + 255) / 256)), dim3(256), 0, s, dst, dpitch, src, spitch, width, rows);
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}
int dev_copy(float* dst, const float* src, size_t n, hipStream_t s) { return dev_copy2d(dst, n, src, n, (int)n, 1, s); }
int dev_zero(float* dst, size_t n, hipStream_t s) { return dev_copy2d(dst, n, nullptr, n, (int)n, 1, s); }


// Weight packing ON THE DEVICE (round 6): the fp32 rows of a Linear cross PCIe once as they are and a kernel writes the
// fragment-major image - 8 consecutive k of a row = one pack (fmt_pack_off), converted with the conversion every activation
// store uses (round to nearest even, fp16 saturating at 65504).  On the host the same loop ran at ~5 ns per weight on ONE
// thread: 0.86 s for the FMT, 1.74 s for the speech-emotion model, 3.4 s per InferenceAgent.to_target(); now the time of the
// copies (tools/probes/retarget_time.py; INTEGRATION.md "Residency").  FLOAT_PACK_HOST=1 keeps the host loop (the A/B switch;
// the two images are equal bit for bit for finite weights - tests/test_variants_gpu.py).
template <class T>
__global__ __launch_bounds__(256) void fmt_pack_w_kernel(typename T::elem* __restrict__ out, const float* __restrict__ w, int N_each,
                                                         int K, int KB, int n0) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int gpr = KB * 4;  // packs per row
  const int n = (int)(idx / gpr), k0 = (int)(idx % gpr) * 8;
  if (n >= N_each) return;
  typename T::pack8 p;
#pragma unroll
  for (int i = 0; i < 8; ++i) T::set(p, i, k0 + i < K ? w[(size_t)n * K + k0 + i] : 0.f);
  T::store8(out + fmt_pack_off(n0 + n, k0, KB), p);
}

template <class T>
int pack_linear_pool(DevicePool* pool, const TensorTable& tt, const std::vector<std::string>& names, int N_each, int K,
                     Lin* out) {
  // Concatenate the named Linear layers along N (used to fuse every adaLN projection into one GEMM).
  typedef typename T::elem E;
  constexpr size_t esz = sizeof(E) / sizeof(u16);  // u16 slots per element (Lin::W is typed u16* for every operand type)
  const int Kp = round_up(K, 128);
  const int N = N_each * (int)names.size();
  static const bool on_host = getenv("FLOAT_PACK_HOST") && atoi(getenv("FLOAT_PACK_HOST")) != 0;
  std::vector<E> hw;
  if (on_host) hw.assign((size_t)N * Kp, (E)0);
  std::vector<float> hb(N, 0.f);
  int rc;
  if ((rc = pool->alloc(&out->W, (size_t)N * Kp * esz, false))) return rc;
  if ((rc = pool->alloc(&out->b, hb.size(), false))) return rc;
  float* stage = nullptr;  // one Linear's fp32 rows on the device
  if (!on_host) FH_CHECK_HIP(hipMalloc(&stage, (size_t)N_each * K * sizeof(float)));
  struct Free {
    float* p;
    ~Free() {
      if (p) (void)hipFree(p);
    }
  } free_stage{stage};
  int n0 = 0;
  for (const std::string& nm : names) {
    const float_tensor_t* w = tt.find(nm + ".weight");
    const float_tensor_t* b = tt.find(nm + ".bias");
    if (!w || !b) {
      fh_set_error("missing checkpoint tensor '%s.weight/.bias'", nm.c_str());
      return FLOAT_E_MISSING;
    }
    if (w->ndim != 2 || w->shape[0] != N_each || w->shape[1] != K || TensorTable::numel(b) != N_each) {
      fh_set_error("tensor '%s.weight' has shape (%lld,%lld), expected (%d,%d)", nm.c_str(), (long long)w->shape[0],
                   (long long)(w->ndim > 1 ? w->shape[1] : 0), N_each, K);
      return FLOAT_E_INVALID;
    }
    if (on_host) {
      for (int n = 0; n < N_each; ++n) {
        const float* src = w->data + (size_t)n * K;
        for (int k = 0; k < K; ++k) hw[fmt_pack_off(n0 + n, k, Kp / 32)] = T::host_from_float(src[k]);
      }
    } else {
      // (null stream: the copy returns when the rows are on the device, the kernel runs before the next copy into `stage`)
      FH_CHECK_HIP(hipMemcpy(stage, w->data, (size_t)N_each * K * sizeof(float), hipMemcpyHostToDevice));
      const size_t packs = (size_t)N_each * (Kp / 8);
      hipLaunchKernelGGL((fmt_pack_w_kernel<T>), dim3((unsigned)((packs + 255) / 256)), dim3(256), 0, nullptr,
                         reinterpret_cast<E*>(out->W), stage, N_each, K, Kp / 32, n0);
      FH_CHECK_HIP(hipGetLastError());
    }
    for (int n = 0; n < N_each; ++n) hb[n0 + n] = b->data[n];
    n0 += N_each;
  }
  if (on_host) FH_CHECK_HIP(hipMemcpy(out->W, hw.data(), hw.size() * sizeof(E), hipMemcpyHostToDevice));
  FH_CHECK_HIP(hipMemcpy(out->b, hb.data(), hb.size() * sizeof(float), hipMemcpyHostToDevice));
  if (!on_host) FH_CHECK_HIP(hipDeviceSynchronize());  // the packed image is complete (and `stage` idle) when the call returns
  out->N = N;
  out->K = Kp;
  return FLOAT_OK;
}

template <class T>
int pack_linear(float_fmt* h, const TensorTable& tt, const std::vector<std::string>& names, int N_each, int K, Lin* out) {
  return pack_linear_pool<T>(&h->pool, tt, names, N_each, K, out);
}

constexpr int kWtRows = 256;  // LayerNorm / attention launches of at most this many rows store write-through (common.hpp, FMT_WT)
// Wide-N path (fused adaLN projection): LDS-staged A, 128 columns per workgroup.
int g_fmt_wide_variant = 7;  // FLOAT_FMT_WIDE_VARIANT: 6 / 7 = LDS-DMA 192 x 320 tile where the shape allows (else 2): lock step / wave rows half a step apart; register-staged 192 x 128 family: 0 = 96 rows x 4 k-blocks per chunk, 1 = 96 x 2, 2 = 192 x 2, 3 = 192 x 4, 4 / 5 = 8 waves
template <class T, int MTW, int KCH, int NWV = 4>
int launch_wide_t(GemmArgs g, bool prime, hipStream_t s) {
  constexpr int smem = 2 * MTW * KCH * 1024;
  auto kern = fmt_gemm_wide_kernel<T, MTW, KCH, NWV>;
  if (prime) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      (void)hipGetLastError();
    return FLOAT_OK;
  }
  const int mt_total = (g.M + 15) / 16;
  g.mblk = (mt_total + MTW - 1) / MTW;
  g.ksplit = 1;
  const dim3 grid((g.N / 128) * g.mblk * (g.zcount > 1 ? g.zcount : 1));
  hipEvent_t e0, e1;
  if (fh_prof_pair(2, &e0, &e1)) hipExtLaunchKernelGGL(kern, grid, dim3(NWV * 64), smem, s, e0, e1, 0, g);
  else hipLaunchKernelGGL(kern, grid, dim3(NWV * 64), smem, s, g);
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}
// 192 x 320 tiles with both operands by LDS-DMA (fmt_gemm_dma_kernel): 8 waves, ring of 4 stages, one workgroup per CU; N in
// blocks of 320 columns and an even number (>= 4) of k-blocks.  Measured per launch of the hoisted projection (50 x 180 rows,
// 0.944 TFLOP), rocprofv3, bitwise the same numbers in every form:
//   register-staged 192 x 128 tile (variant 2)                                   1376 us  (686 TFLOP/s)
//   LDS-DMA tile, waves in lock step, stores straight from the accumulators      1260
//   + output through LDS (whole lines per store)                                 1150
//   + 2 column blocks per XCD group instead of 4 (FLOAT_FMT_ZGROUP)              1045     (variant 6)
//   + wave rows half a step apart (variant 7, the default)                       1021     (924 TFLOP/s, 37 % of the MFMA peak)
//   the same with the DMA pieces issued between the MFMA rows                    1115
// Not faster: 4 waves / 160 columns / ring of 3 with two workgroups per CU (1458), fragment reads spread between the MFMA rows
// (1172), a staggered start of the first workgroup generation, non-temporal stores.  In-kernel clocks (s_memtime /
// s_memrealtime) put a 32-step tile at ~49 000 clocks at 2.1 GHz, of which the bare barrier skeleton is a third.
template <class T, int NWC, int NS, int STG>
int launch_dma_t(GemmArgs g, bool prime, hipStream_t s) {
  constexpr int smem = NS * (12 + 5 * NWC) * 1024, BN = 80 * NWC;
  auto kern = fmt_gemm_dma_kernel<T, NWC, NS, STG>;
  if (prime) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      (void)hipGetLastError();
    return FLOAT_OK;
  }
  g.mblk = ((g.M + 15) / 16 + 11) / 12;
  const dim3 grid((g.N / BN) * g.mblk * (g.zcount > 1 ? g.zcount : 1));
  hipEvent_t e0, e1;
  if (fh_prof_pair(2, &e0, &e1)) hipExtLaunchKernelGGL(kern, grid, dim3(NWC * 128), smem, s, e0, e1, 0, g);
  else hipLaunchKernelGGL(kern, grid, dim3(NWC * 128), smem, s, g);
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}
bool dma_shape_ok(const GemmArgs& g, int bn) { return g.N % bn == 0 && g.K % 64 == 0 && g.K >= 128; }

// The hoisted projection of a whole batch of evaluations on DENSE rows: the persistent one-wave-per-SIMD kernel
// (fmt_big_kernels.hpp), one workgroup per CU.  N in column blocks of 256, eight of them per XCD group; an even number (>= 4) of
// k-blocks (the K loop is unrolled by two behind four peeled steps).  Bitwise the numbers of fmt_gemm_dma_kernel.
// FLOAT_FMT_BIG=0 keeps the one-tile-per-workgroup kernels on rows padded per evaluation (the A/B switch).
int g_fmt_big = 1;
constexpr int kBigMinRows = 1536;  // below 8 row blocks the padded layout's kernels stay (a single evaluation: 180 rows)
constexpr int kBigSmem = 4 * 28 * 1024 + 4 * 4096;
bool big_shape_ok(int rows_total, int N, int K, int n_cu) {
  return g_fmt_big && rows_total >= kBigMinRows && N % 2048 == 0 && K % 64 == 0 && K >= 128 && n_cu >= 8;
}
template <class T>
int launch_big4(const u16* A, const FmtLin& L, float* out, int rows_total, int ldo, int n_cu, bool prime, hipStream_t s) {
  if constexpr (T::is32) {
    fh_set_error("the fp32 verification mode has no persistent projection kernel");
    return FLOAT_E_INVALID;
  } else {
    auto kern = fmt_gemm_big4_kernel<T, 4>;
    if (prime) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kBigSmem) != hipSuccess)
        (void)hipGetLastError();
      return FLOAT_OK;
    }
    BigArgs g{A, L.W, L.b, out, rows_total, L.N, L.K, ldo, (rows_total + 191) / 192, L.N / 256};
    const dim3 grid((unsigned)((n_cu / 8) * 8));
    hipEvent_t e0, e1;
    if (fh_prof_pair(2, &e0, &e1)) hipExtLaunchKernelGGL(kern, grid, dim3(256), kBigSmem, s, e0, e1, 0, g);
    else hipLaunchKernelGGL(kern, grid, dim3(256), kBigSmem, s, g);
    FH_CHECK_HIP(hipGetLastError());
    return FLOAT_OK;
  }
}

template <class T>
int launch_wide(const GemmArgs& g, bool prime, hipStream_t s) {
  const int mt = (g.M + 15) / 16;
  if (prime) {
    (void)launch_big4<T>(nullptr, FmtLin{}, nullptr, 0, 0, 0, true, s);
    (void)launch_dma_t<T, 4, 4, 0>(g, true, s);
    (void)launch_dma_t<T, 4, 4, 1>(g, true, s);
    (void)launch_wide_t<T, 4, 4>(g, true, s);
    (void)launch_wide_t<T, 5, 4>(g, true, s);
    (void)launch_wide_t<T, 6, 4>(g, true, s);
    (void)launch_wide_t<T, 6, 2>(g, true, s);
    (void)launch_wide_t<T, 12, 2>(g, true, s);
    (void)launch_wide_t<T, 12, 4>(g, true, s);
    (void)launch_wide_t<T, 12, 2, 8>(g, true, s);
    (void)launch_wide_t<T, 12, 4, 8>(g, true, s);
    return FLOAT_OK;
  }
  if (g_fmt_wide_variant == 6 && mt > 4 && dma_shape_ok(g, 320)) return launch_dma_t<T, 4, 4, 0>(g, false, s);
  if (g_fmt_wide_variant == 7 && mt > 4 && dma_shape_ok(g, 320)) return launch_dma_t<T, 4, 4, 1>(g, false, s);
  if (mt <= 4) return launch_wide_t<T, 4, 4>(g, false, s);
  // 192-row blocks also for the stacked clips of a batch (mt > 12): the last block reads up to 11 row tiles past the batch (the
  // operand buffers are padded for it, the rows are never stored); 80-row blocks ran the batched projection at 240 TFLOP/s
  // against 700 for 192-row ones
  const int variant = (g_fmt_wide_variant == 6 || g_fmt_wide_variant == 7) ? 2 : g_fmt_wide_variant;  // shapes the LDS-DMA tile does not take
  if (mt <= 12 || variant == 2 || variant >= 4) {
    switch (variant) {
      case 1: return launch_wide_t<T, 6, 2>(g, false, s);
      case 2: return launch_wide_t<T, 12, 2>(g, false, s);
      case 3: return launch_wide_t<T, 12, 4>(g, false, s);
      case 4: return launch_wide_t<T, 12, 2, 8>(g, false, s);  // two waves per SIMD, rows split over the wave pairs
      case 5: return launch_wide_t<T, 12, 4, 8>(g, false, s);
      default: return launch_wide_t<T, 6, 4>(g, false, s);
    }
  }
  return launch_wide_t<T, 5, 4>(g, false, s);
}
bool g_fmt_wide = true;  // FLOAT_FMT_WIDE=0 falls back to the generic tiling (A/B measurement)

// ---- GEMM instantiation table: (row tiles, column tiles, waves splitting K) per workgroup ----
template <class T, int MTW, int NT, int NW, int EPI>
int launch_gemm_t(GemmArgs g, bool prime, hipStream_t s) {
  constexpr int smem = NW * MTW * 16 * NT * 16 * (int)sizeof(float);
  auto kern = fmt_gemm_kernel<T, MTW, NT, NW, EPI>;
  if (smem > 160 * 1024) {  // gfx950: 160 KiB of LDS per workgroup
    if (!prime) fh_set_error("GEMM tiling %dx%d tiles with %d waves needs %d B of LDS", MTW, NT, NW, smem);
    return FLOAT_E_INVALID;
  }
  if (prime) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      (void)hipGetLastError();
    return FLOAT_OK;
  }
  const int mt_total = (g.M + 15) / 16;
  g.mblk = (mt_total + MTW - 1) / MTW;
  if (EPI != EPI_PARTIAL || g.ksplit < 1) g.ksplit = 1;
  dim3 grid((g.N / (NT * 16)) * g.mblk * g.ksplit);
  hipEvent_t e0, e1;
  if (fh_prof_pair(0, &e0, &e1)) hipExtLaunchKernelGGL(kern, grid, dim3(NW * 64), smem, s, e0, e1, 0, g);
  else hipLaunchKernelGGL(kern, grid, dim3(NW * 64), smem, s, g);
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

// Which tilings exist.  Split tilings (a quarter/third/half of the CFG rows per workgroup) carry every
// epilogue and 4/8/16 K-splitting waves; full-height tilings (4 waves, LDS bound) only the two
// epilogues that need them: EPI_CFG (all CFG rows of a token in one workgroup) and EPI_F32.
#define FMT_SPLIT_SHAPES(X, NW, EPI) \
  X(3, 1, NW, EPI) X(3, 2, NW, EPI) X(3, 4, NW, EPI) X(5, 1, NW, EPI) X(5, 2, NW, EPI) X(5, 4, NW, EPI) X(4, 1, NW, EPI) \
  X(4, 2, NW, EPI) X(6, 2, NW, EPI) X(2, 1, NW, EPI) X(1, 1, NW, EPI)
#define FMT_FOR_SPLIT(X, EPI) FMT_SPLIT_SHAPES(X, 4, EPI) FMT_SPLIT_SHAPES(X, 8, EPI) FMT_SPLIT_SHAPES(X, 16, EPI)
#define FMT_FOR_FULL(X, EPI) X(12, 2, 4, EPI) X(15, 2, 4, EPI) X(12, 1, 4, EPI) X(15, 1, 4, EPI) X(12, 1, 8, EPI) X(15, 1, 8, EPI)

template <class T, int EPI>
int launch_gemm(const GemmArgs& g, int mtw, int nt, int nw, bool prime, hipStream_t s) {
  if (!prime && (g.N % (nt * 16) || g.K % (32 * nw * ((EPI == EPI_PARTIAL && g.ksplit > 1) ? g.ksplit : 1)))) {
    fh_set_error("gemm shape N=%d K=%d not tileable by %d columns / %d waves", g.N, g.K, nt * 16, nw);
    return FLOAT_E_INVALID;
  }
#define FMT_CASE(MTW, NT, NW, E) \
  if (mtw == MTW && nt == NT && nw == NW) return launch_gemm_t<T, MTW, NT, NW, E>(g, prime, s);
  if constexpr (T::is32) {
    // the fp32 verification mode runs a handful of 16-column tilings (pick_tiling): speed is not its point
    FMT_CASE(1, 1, 4, EPI) FMT_CASE(2, 1, 4, EPI) FMT_CASE(3, 1, 4, EPI) FMT_CASE(4, 1, 4, EPI) FMT_CASE(5, 1, 4, EPI)
    if constexpr (EPI == EPI_CFG) {
      FMT_CASE(1, 1, 8, EPI) FMT_CASE(3, 1, 8, EPI) FMT_CASE(4, 1, 8, EPI)
    }
  } else {
    FMT_FOR_SPLIT(FMT_CASE, EPI)
    if constexpr (EPI == EPI_F32 || EPI == EPI_CFG) {
      FMT_FOR_FULL(FMT_CASE, EPI)
    }
  }
#undef FMT_CASE
  fh_set_error("no GEMM tiling (%d x %d tiles, %d waves) for epilogue %d", mtw, nt, nw, EPI);
  return FLOAT_E_INVALID;
}

// ---- stacked clips (>= kRbMinRows rows): the row-blocked LDS-DMA tile (fmt_rb_kernels.hpp) for qkv / proj / fc1 / fc2
template <class T, int MI, int NJ, int KPS, int NS, int EPI>
int launch_rbs_t(GemmArgs g, bool prime, hipStream_t s) {
  constexpr int smem = fmt_rb_smem(MI, NJ, KPS, NS), ROWS = 32 * MI, BN = 32 * NJ;
  auto kern = fmt_gemm_rbs_kernel<T, MI, NJ, KPS, NS, EPI>;
  if (prime) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      (void)hipGetLastError();
    return FLOAT_OK;
  }
  if (EPI != EPI_PARTIAL || g.ksplit < 1) g.ksplit = 1;
  FH_REQUIRE(g.N % BN == 0 && (g.K / 32) % (g.ksplit * KPS) == 0 && g.K / 32 / g.ksplit / KPS >= 1,
             "row-blocked GEMM: N=%d K=%d not tileable by %d columns / %d K slices of %d-k-block stages", g.N, g.K, BN, g.ksplit, KPS);
  g.mblk = (g.M + ROWS - 1) / ROWS;
  g.touch.W = nullptr;
  const dim3 grid((unsigned)((g.N / BN) * g.mblk * g.ksplit));
  hipEvent_t e0, e1;
  if (fh_prof_pair(3, &e0, &e1)) hipExtLaunchKernelGGL(kern, grid, dim3(512), smem, s, e0, e1, 0, g);
  else hipLaunchKernelGGL(kern, grid, dim3(512), smem, s, g);
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}
// tile shapes built: 0 = 96 x 64 (two workgroups per CU), 1 = 96 x 128, 2 = 192 x 128 (ring of 3)
template <class T, int EPI>
int launch_rbs(const GemmArgs& g, int shape, bool prime, hipStream_t s) {
  if constexpr (T::is32) {
    fh_set_error("the fp32 verification mode has no row-blocked tiling");
    return FLOAT_E_INVALID;
  } else {
  if (prime) {
    (void)launch_rbs_t<T, 3, 2, 2, 4, EPI>(g, true, s);
    (void)launch_rbs_t<T, 3, 4, 2, 4, EPI>(g, true, s);
    (void)launch_rbs_t<T, 6, 4, 2, 3, EPI>(g, true, s);
    return FLOAT_OK;
  }
  switch (shape) {
    case 0: return launch_rbs_t<T, 3, 2, 2, 4, EPI>(g, false, s);
    case 1: return launch_rbs_t<T, 3, 4, 2, 4, EPI>(g, false, s);
    default: return launch_rbs_t<T, 6, 4, 2, 3, EPI>(g, false, s);
  }
  }
}
// Which tile, per layer and row count (tools/probes/gemm_lab.hip on MI355X, us per launch incl. the launch boundary, weights
// rotating over 8 buffers; 48 x 64 tiling -> best row-blocked tile):
//   rows    qkv (3072 x 1024)     proj (1024 x 1024)      fc1 (4096 x 1024)      fc2 (1024 x 4096, 4 K slices)
//    360     9.7 ->  6.7 (96x64)   5.1 (kept)             10.9 ->  8.0 (96x64)   10.9 ->  8.3 (96x64)
//    720    14.4 ->  9.9 (96x128)  6.6 ->  6.4 (96x64 /2) 18.8 -> 13.2 (96x128)  19.0 -> 12.8 (96x128)
//   1440    24.4 -> 16.4 (96x64)  11.6 ->  9.8 (96x64 /2) 33.1 -> 22.5 (192x128) 34.1 -> 20.9 (192x128)
//   2880    53.7 -> 31.7 (192x128) 22.2 -> 15.7 (192x128 /2) 74.5 -> 43.3 (192x128) 71.8 -> 39.1 (192x128)
// FLOAT_FMT_RB=0 keeps the 48 x 64 tiling (the A/B switch); FLOAT_FMT_RB_QKV / _PROJ / _FC1 / _FC2 = "shape[,ksplit]" override.
constexpr int kRbMinRows = 300;
int g_fmt_rb = 1;
struct RbPlan {
  int shape = -1, ksplit = 1;  // shape < 0: the weight-streaming tiling
};
enum { RB_QKV = 0, RB_PROJ, RB_FC1, RB_FC2 };
RbPlan pick_rb(int layer, int M) {
  static const char* const envs[4] = {"FLOAT_FMT_RB_QKV", "FLOAT_FMT_RB_PROJ", "FLOAT_FMT_RB_FC1", "FLOAT_FMT_RB_FC2"};
  RbPlan p;
  if (!g_fmt_rb || M < kRbMinRows) return p;
  const int tier = M < 540 ? 0 : (M < 1100 ? 1 : (M < 2200 ? 2 : 3));
  static const int shapes[4][4] = {/* qkv */ {0, 1, 0, 2}, /* proj */ {-1, 0, 0, 2}, /* fc1 */ {0, 1, 2, 2}, /* fc2 */ {0, 1, 2, 2}};
  p.shape = shapes[layer][tier];
  // fc2: 4 K slices fill the CUs up to 1440 rows; from 2200 rows on 2 slices do (15 x 8 tiles x 2) and halve the fp32 slabs the
  // next LayerNorm folds (16 clips: 415.7 vs 438.6 ms per 250 evaluations; 8 clips the other way round: 251.6 vs 241.9)
  p.ksplit = layer == RB_PROJ ? 2 : (layer == RB_FC2 ? (tier == 3 ? 2 : 4) : 1);
  if (const char* e = getenv(envs[layer])) {
    int sh = p.shape, ks = p.ksplit;
    if (sscanf(e, "%d,%d", &sh, &ks) >= 1) {
      p.shape = sh;
      if (layer == RB_PROJ || layer == RB_FC2) p.ksplit = (ks == 1 || ks == 2 || ks == 4 || ks == 8) ? ks : p.ksplit;
    }
  }
  return p;
}

template <class T, int EPI>
void prime_epi() {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  if constexpr (T::is32) {
    for (int mtw = 1; mtw <= 5; ++mtw) (void)launch_gemm<T, EPI>(g, mtw, 1, 4, true, nullptr);
    if (EPI == EPI_CFG)
      for (int mtw : {1, 3, 4}) (void)launch_gemm<T, EPI>(g, mtw, 1, 8, true, nullptr);
    return;
  }
  static const int shapes[][2] = {{3, 1}, {3, 2}, {3, 4}, {5, 1}, {5, 2}, {5, 4}, {4, 1}, {4, 2}, {6, 2}, {2, 1}, {1, 1}};
  for (auto& c : shapes)
    for (int nw : {4, 8, 16}) (void)launch_gemm<T, EPI>(g, c[0], c[1], nw, true, nullptr);
  if (EPI == EPI_F32 || EPI == EPI_CFG) {
    for (int nt : {1, 2}) {
      (void)launch_gemm<T, EPI>(g, 12, nt, 4, true, nullptr);
      (void)launch_gemm<T, EPI>(g, 15, nt, 4, true, nullptr);
    }
    (void)launch_gemm<T, EPI>(g, 12, 1, 8, true, nullptr);
    (void)launch_gemm<T, EPI>(g, 15, 1, 8, true, nullptr);
  }
}
template <class T>
void prime_kernels() {
  prime_epi<T, EPI_F32>();
  prime_epi<T, EPI_T16>();
  prime_epi<T, EPI_SILU_P16>();
  prime_epi<T, EPI_GELU_P16>();
  prime_epi<T, EPI_GATE_RES>();
  prime_epi<T, EPI_XEMBED>();
  prime_epi<T, EPI_CFG>();
  prime_epi<T, EPI_PARTIAL>();
  prime_epi<T, EPI_GELUERF_P16>();
  if constexpr (!T::is32) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    (void)launch_wide<T>(g, true, nullptr);
    (void)launch_rbs<T, EPI_T16>(g, 0, true, nullptr);
    (void)launch_rbs<T, EPI_GELU_P16>(g, 0, true, nullptr);
    (void)launch_rbs<T, EPI_PARTIAL>(g, 0, true, nullptr);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fmt_mega_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 48 * 64 * 4) != hipSuccess)
      (void)hipGetLastError();
  }
}

// Tiling choice: split the rows over row blocks so that narrow layers still fill the 256 CUs, and
// split K over as many waves as keeps >= 4 k-steps per wave (one prefetch round per wave).
struct Tiling {
  int mtw, nt, nw;
};
int g_fmt_full_nw = 8;  // FLOAT_FMT_FULL_NW: waves of the full-height (CFG epilogue) tiling
int g_fmt_plan_override[6] = {0, 0, 0, 0, 0, 0};  // FLOAT_FMT_PLAN="mtw,nt,nw (narrow), mtw,nt,nw (wide)": tuning aid
int pick_nw(int K, int forced) {
  const int KB = K / 32;
  if (forced && KB % forced == 0) return forced;
  if (KB >= 128 && KB % 16 == 0) return 16;
  if (KB >= 32 && KB % 8 == 0) return 8;
  return 4;
}
Tiling pick_tiling(int M, int N, int K, bool need_full_rows) {
  const int mt = (M + 15) / 16;
  if (need_full_rows) {
    if (mt <= 4) return {4, 1, pick_nw(K, 0)};
    // 16 columns per workgroup: twice the workgroups of the 32-column tile; 8 K-splitting waves keep twice
    // the operand bytes in flight (each of the 32 workgroups streams the whole 393 KB activation operand)
    return {mt <= 12 ? 12 : 15, 1, (g_fmt_full_nw == 8 && (K / 32) % 8 == 0) ? 8 : 4};
  }
  int split = mt <= 4 ? 4 : (mt <= 12 ? 3 : 5);
  if (mt <= 2) split = mt;
  const int blocks = (mt + split - 1) / split;
  const bool wide = N >= 16384;  // the fused adaLN projection
  if (mt >= 12) {  // tuning overrides only apply to the CFG-batched shapes (buffers hold 240 rows)
    const int* o = g_fmt_plan_override + (wide ? 3 : 0);
    if (o[0]) {
      int nw = o[0] >= 12 ? 4 : pick_nw(K, o[2]);
      while (nw > 4 && nw * o[0] * 16 * o[1] * 16 * 4 > 160 * 1024) nw >>= 1;
      return {o[0], o[1], nw};
    }
  }
  if (wide) return {mt <= 4 ? 4 : 6, 2, std::min(8, pick_nw(K, 0))};
  // stacked clips (float_fmt_sample_batch, more than 15 row tiles): the one-clip tile (48 x 64) with 4 K-splitting waves, so that
  // two or three workgroups share a CU (49 KB of LDS each instead of 98).  Measured per 250 evaluations, B = 2 / 4 clips:
  // 120.7 / 184.5 ms against 136.4 / 200.6 with the 80-row tiles this function would pick below, 130.1 / 208.1 with 8 waves
  // (one clip: 85.0).  Operands come straight from L2 per workgroup, so the traffic grows with rows x column blocks: the
  // batched chain wants an LDS-staged large-tile kernel like fmt_gemm_wide_kernel with these epilogues.
  if (mt > 15 && N % 64 == 0) return {3, 4, 4};
  // column tiles per workgroup: the widest (<= 4) that still gives >= ~200 workgroups, so that each CU
  // runs ONE workgroup (two back-to-back workgroups per CU double the latency chain of the layer)
  int nt = 1;
  if (mt >= 5) {
    if ((N / 64) * blocks >= 192 && N % 64 == 0) nt = 4;
    else if ((N / 32) * blocks >= 192 && N % 32 == 0) nt = 2;
  } else if ((N / 16) * blocks > 512) {
    nt = 2;
  }
  return {split, nt, pick_nw(K, 0)};
}

GemmArgs base_args(const u16* A, const Lin& L, int M) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = A;
  g.W = L.W;
  g.bias = L.b;
  g.K = L.K;
  g.N = L.N;
  g.M = M;
  return g;
}

// fp32 verification mode: 16 columns per workgroup, 4 K-splitting waves, the row split of the 16-bit tilings capped at 5 tiles
Tiling pick_tiling32(int M) {
  const int mt = (M + 15) / 16;
  return {mt <= 5 ? mt : (mt <= 12 ? 3 : 5), 1, 4};
}

// Per-layer tiling overrides of the one-clip chain (tuning aid): FLOAT_FMT_PLAN_QKV / _PROJ / _FC1 / _FC2 = "mtw,nt,nw".
struct LayerPlan {
  int v[3] = {0, 0, 0};
  explicit LayerPlan(const char* env) {
    if (const char* e = getenv(env)) sscanf(e, "%d,%d,%d", &v[0], &v[1], &v[2]);
  }
  bool on() const { return v[0] > 0; }
};

template <class T, int EPI>
int run_gemm(const GemmArgs& g, hipStream_t s, bool need_full_rows = false, const LayerPlan* plan = nullptr) {
  if constexpr (T::is32) {
    FH_REQUIRE(!need_full_rows, "the fp32 mode has no all-rows CFG epilogue tiling (token-blocked head only)");
    const Tiling t = pick_tiling32(g.M);
    GemmArgs g2 = g;
    g2.touch.W = nullptr;
    return launch_gemm<T, EPI>(g2, t.mtw, t.nt, t.nw, false, s);
  }
  if (plan && plan->on() && (g.M + 15) / 16 <= 15) {
    GemmArgs g2 = g;
    g2.touch.W = nullptr;  // the touch descriptors follow the default block decode
    return launch_gemm<T, EPI>(g2, plan->v[0], plan->v[1], plan->v[2], false, s);
  }
  const Tiling t = pick_tiling(g.M, g.N, g.K, need_full_rows);
  return launch_gemm<T, EPI>(g, t.mtw, t.nt, t.nw, false, s);
}

// Split-K GEMM whose gated residual add happens in the next LayerNorm launch (EPI_PARTIAL + LnRed).
// The tiling is the one a GEMM with ksplit * N columns and K / ksplit would get: same workgroup
// count, a fraction of the activation bytes per workgroup.
int g_fmt_fc2_split = 4;   // FLOAT_FMT_FC2_SPLIT: K slices of mlp.fc2 (0 = in-GEMM gate*residual epilogue)
int g_fmt_proj_split = 0;  // FLOAT_FMT_PROJ_SPLIT: same for attn.proj
struct PendingRed {
  int ks = 0;
  LnRed red{};
};

template <class T>
int run_gemm_partial(float_fmt* h, GemmArgs g, int ksplit, hipStream_t s, const LayerPlan* plan = nullptr) {
  g.ksplit = ksplit;
  g.out_f32 = h->slab;
  g.ldo = g.N;
  g.slab_stride = (size_t)h->Mpad * g.N;
  if (plan && plan->on() && !T::is32 && (g.M + 15) / 16 <= 15) {
    g.touch.W = nullptr;
    return launch_gemm<T, EPI_PARTIAL>(g, plan->v[0], plan->v[1], plan->v[2], false, s);
  }
  Tiling t = T::is32 ? pick_tiling32(g.M) : pick_tiling(g.M, g.N * ksplit, g.K / ksplit, false);
  while (t.nt > 1 && g.N % (t.nt * 16)) t.nt >>= 1;
  if (T::is32) g.touch.W = nullptr;
  return launch_gemm<T, EPI_PARTIAL>(g, t.mtw, t.nt, t.nw, false, s);
}

// Touch descriptor for the weights of GEMM `L` as it will be launched for M rows (ksplit = 0: plain GEMM, else EPI_PARTIAL
// with that many K slices), to be executed by `lanes` threads per XCD with at most `per_lane` lines each; W = nullptr when the
// GEMM's block decode is not the XCD-affine one or the lanes cannot cover it.
// FLOAT_FMT_TOUCH bit mask - who pulls whose weights: 1 LayerNorm -> qkv and fc1 (64: only LN2 -> fc1, 128: only LN1 -> qkv),
// 2 attention -> proj, 4 fc1 -> fc2, 8 qkv -> proj, 16 proj -> fc1, 32 fc2 -> the next block's qkv / the head.
// Default 2 + 4 + 32 + 128 (r01, ms per 250 evaluations, same box: none 90.8, 2+4 87.6, 4+32 86.0-86.7, 2+4+32 83.2-83.5 after the
// head change, + LN1 -> qkv 82.5; touching fc1's weights - from LayerNorm, proj or qkv - never paid).
// Round 2, with the adaLN weights out of the step (105 MB less cycling through the Infinity Cache per evaluation): LN2 -> fc1
// now pays too: 230 = 166 + 64 gives 83.6 vs 84.9 ms (proj -> fc1 instead: 85.8; attention or qkv as extra pullers: 85.2-86.0).
int g_fmt_touch = 230;
TouchSpec make_touch(const Lin& L, int M, int ksplit, unsigned lanes, unsigned per_lane, int force_nt = 0) {
  TouchSpec t{};
  Tiling tl = ksplit ? pick_tiling(M, L.N * ksplit, L.K / ksplit, false) : pick_tiling(M, L.N, L.K, false);
  if (force_nt) tl.nt = force_nt;
  if (ksplit)
    while (tl.nt > 1 && L.N % (tl.nt * 16)) tl.nt >>= 1;
  const int ks = ksplit ? ksplit : 1;
  if (L.N % (tl.nt * 16) || L.K % (32 * ks) || 8 % ks) return t;
  const int nbn = L.N / (tl.nt * 16);
  if ((nbn * ks) % 8) return t;
  auto lg = [](unsigned v) {  // log2 of a power of two, else -1
    int n = 0;
    while ((1u << n) < v) ++n;
    return (1u << n) == v ? n : -1;
  };
  const unsigned P = 8 / ks, run_lines = (unsigned)(L.K / 32 / ks) * 8u;
  if (lg(run_lines) < 0 || lg((unsigned)tl.nt) < 0) return t;
  t.run_shift = (unsigned)lg(run_lines);
  t.nt_shift = (unsigned)lg((unsigned)tl.nt);
  t.p_shift = (unsigned)lg(P);
  t.tile_bytes = (unsigned)(L.K / 32) * 1024u;
  t.total = ((unsigned)nbn / P) * (unsigned)tl.nt * run_lines;
  if ((size_t)t.total > (size_t)lanes * per_lane) return t;
  t.W = reinterpret_cast<const char*>(L.W);
  return t;
}

// The same for a row-blocked GEMM (fmt_gemm_rbs_kernel, no K split): only the FIRST stages of every weight column tile - what
// each of its workgroups waits for before it can start (1.5 of fc1's 10 us at 720 rows: every CU asks for cold lines at once).
// Its block decode puts column block bx on XCD bx % 8, like the 48 x 64 tiling's.  FLOAT_FMT_RB_TOUCH = k-blocks to pull (0 = off).
// Measured, ms per 250 evaluations of 4 / 16 clips: none 152.2 / 408.4, 4 k-blocks 151.8, 8: 151.4, 16: 150.6 / 404.6 (kept),
// 32 (the whole K of every tile): 150.8 / 410.0.
int g_fmt_rb_touch = 16;
TouchSpec make_touch_rb(const Lin& L, int shape, unsigned lanes, unsigned per_lane) {
  TouchSpec t{};
  const int ct = shape == 0 ? 4 : 8;  // 16-column tiles per column block: 96 x 64 | 96 x 128, 192 x 128
  const int kb = g_fmt_rb_touch;
  if (kb <= 0 || (kb & (kb - 1)) || L.N % (ct * 16) || (L.N / (ct * 16)) % 8 || L.K / 32 < kb) return t;
  const unsigned run_lines = (unsigned)kb * 8u;  // a k-block of a column tile is 1 KiB = 8 lines, consecutive k-blocks are consecutive
  auto lg = [](unsigned v) {
    int n = 0;
    while ((1u << n) < v) ++n;
    return n;
  };
  t.run_shift = (unsigned)lg(run_lines);
  t.nt_shift = (unsigned)lg((unsigned)ct);
  t.p_shift = 3;
  t.tile_bytes = (unsigned)(L.K / 32) * 1024u;
  t.total = (unsigned)(L.N / (ct * 16) / 8) * (unsigned)ct * run_lines;
  if ((size_t)t.total > (size_t)lanes * per_lane) return t;
  t.W = reinterpret_cast<const char*>(L.W);
  return t;
}

// threads per XCD of the launch run_gemm makes for a plain (M, N, K) GEMM
unsigned gemm_lanes_per_xcd(int M, int N, int K) {
  const Tiling t = pick_tiling(M, N, K, false);
  const int mblk = ((M + 15) / 16 + t.mtw - 1) / t.mtw;
  return (unsigned)((N / (t.nt * 16)) * mblk / 8) * (unsigned)(t.nw * 64);
}

template <class T>
int launch_lnmod(float_fmt* h, int M, const float* shift, const float* scale, hipStream_t s, PendingRed* pend = nullptr,
                 const Lin* next = nullptr, u16* out = nullptr, int perm = 0, int touch_bit = 1, int rb_shape = -1) {
  const int nv = h->D / 256;
  // one row (wave) per workgroup: 180 single-wave workgroups spread over 180 CUs (4 rows per workgroup: +0.4 %)
  static const int rpw = getenv("FLOAT_FMT_LN_ROWS") ? std::max(1, std::min(4, atoi(getenv("FLOAT_FMT_LN_ROWS")))) : 1;
  // rpw == 1: the kernel maps ids to rows in groups of 8 rows per XCD -> 64 row slots per group of 64 ids
  dim3 grid(rpw == 1 ? ((M + 63) / 64) * 64 : (M + rpw - 1) / rpw), block(64 * rpw);
  const int ks = pend ? pend->ks : 0;
  const bool wt = M <= kWtRows;  // write-through outputs for one clip's rows only (common.hpp, FMT_WT)
  LnRed red{};
  if (ks) red = pend->red;
  TouchSpec pf{};
  if (next && (g_fmt_touch & (1 | touch_bit)) && rpw == 1 && !T::is32)
    pf = rb_shape >= 0 ? make_touch_rb(*next, rb_shape, (grid.x / 8) * 64, 6) : make_touch(*next, M, 0, (grid.x / 8) * 64, 6);
#define LN_LAUNCH(NV, KS)                                                                                                          \
  do {                                                                                                                             \
    if (pf.W && wt) hipLaunchKernelGGL((fmt_lnmod_kernel<T, NV, KS, true, true>), grid, block, 0, s, h->xres, M, shift, scale, h->Ntot, out ? out : h->h16, red, pf, h->ntok, perm, h->sat); \
    else if (pf.W) hipLaunchKernelGGL((fmt_lnmod_kernel<T, NV, KS, true, false>), grid, block, 0, s, h->xres, M, shift, scale, h->Ntot, out ? out : h->h16, red, pf, h->ntok, perm, h->sat); \
    else if (wt) hipLaunchKernelGGL((fmt_lnmod_kernel<T, NV, KS, false, true>), grid, block, 0, s, h->xres, M, shift, scale, h->Ntot, out ? out : h->h16, red, pf, h->ntok, perm, h->sat); \
    else hipLaunchKernelGGL((fmt_lnmod_kernel<T, NV, KS, false, false>), grid, block, 0, s, h->xres, M, shift, scale, h->Ntot, out ? out : h->h16, red, pf, h->ntok, perm, h->sat);   \
  } while (0)
#define LN_CASE(NV)                     \
  case NV:                              \
    if (ks == 0) LN_LAUNCH(NV, 0);      \
    else if (ks == 1) LN_LAUNCH(NV, 1); \
    else if (ks == 2) LN_LAUNCH(NV, 2); \
    else if (ks == 4) LN_LAUNCH(NV, 4); \
    else LN_LAUNCH(NV, 8);              \
    break;
  switch (nv) {
    LN_CASE(1) LN_CASE(2) LN_CASE(4) LN_CASE(8)
    default:
      fh_set_error("dim_h %d unsupported (must be 256*{1,2,4,8})", h->D);
      return FLOAT_E_INVALID;
  }
#undef LN_CASE
#undef LN_LAUNCH
  if (pend) pend->ks = 0;
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

// Banded attention over the M rows of qkv16 -> att16 (packed operand of attn.proj); `pull`: the GEMM whose weights the
// workgroups touch meanwhile (TouchSpec), or nullptr.
template <class T>
void launch_attn(float_fmt* h, int M, const Lin* pull, hipStream_t s) {
  const float_fmt_cfg_t& c = h->cfg;
  const int D = h->D, ntok = h->ntok;
  // queries per workgroup / lanes per query (FLOAT_FMT_ATTN="qpw,lpq"): one 8-row output group per workgroup by default
  static int qpw = 8, lpq = 16;  // r01: 16 lanes per query (8 dims each) 81.5-81.9 ms per 250 evaluations, 8 lanes 82.3-82.9
  static const bool parsed = [] {
    if (const char* v = getenv("FLOAT_FMT_ATTN")) sscanf(v, "%d,%d", &qpw, &lpq);
    if (lpq != 16) lpq = 8;
    qpw = std::max(1, std::min(512 / lpq, qpw));
    return true;
  }();
  (void)parsed;
  dim3 grid(c.heads, (M + qpw - 1) / qpw), block(qpw * lpq);
  TouchSpec pf{};
  if (pull && !T::is32) pf = make_touch(*pull, M, 0, (grid.x * grid.y / 8) * block.x, 2);
#define ATTN_LAUNCH(LPQ, TCH)                                                                                                  \
  do {                                                                                                                         \
    if (M <= kWtRows) hipLaunchKernelGGL((fmt_attn_kernel<T, LPQ, TCH, true>), grid, block, 0, s, h->qkv16, 3 * D, h->att16, ntok, M, D, c.attn_window, pf, h->sat); \
    else hipLaunchKernelGGL((fmt_attn_kernel<T, LPQ, TCH, false>), grid, block, 0, s, h->qkv16, 3 * D, h->att16, ntok, M, D, c.attn_window, pf, h->sat);             \
  } while (0)
  if (lpq == 16) {
    if (pf.W) ATTN_LAUNCH(16, true);
    else ATTN_LAUNCH(16, false);
  } else {
    if (pf.W) ATTN_LAUNCH(8, true);
    else ATTN_LAUNCH(8, false);
  }
#undef ATTN_LAUNCH
}

// Banded attention + attn.proj as one launch (fmt_attnproj_kernel): slab[head] = attention_head(qkv16) @ W_proj[:, head]^T for
// the M rows; the caller hands the fold (bias, gate, residual) to the next LayerNorm launch through PendingRed with ks = heads.
// FLOAT_FMT_ATTNPROJ=1|2 (heads per workgroup; read at float_fmt_create) selects it; the default is the two-launch form
// (fmt_attn_kernel, then the proj GEMM), which measured the same or faster - see the kernel's header.
int attnproj_hpw(const float_fmt* h) {
  const int hd = h->cfg.heads, hpw = h->attnproj;
  if (hpw <= 0 || h->D != hd * 128 || hd % hpw) return 0;
  const int ks = hd / hpw;
  return (ks == 1 || ks == 2 || ks == 4 || ks == 8) && (hpw == 1 || hpw == 2) ? hpw : 0;
}
template <class T>
int launch_attnproj(float_fmt* h, int M, const Lin& proj, hipStream_t s) {
  const float_fmt_cfg_t& c = h->cfg;
  const int hpw = attnproj_hpw(h);
  GemmArgs g = base_args(nullptr, proj, M);
  g.sat = h->sat;
  g.out_f32 = h->slab;
  g.ldo = g.N;
  g.slab_stride = (size_t)h->Mpad * g.N;
  g.ksplit = c.heads / hpw;
  g.mblk = ((M + 15) / 16 + 2) / 3;
  const dim3 grid((unsigned)(g.ksplit * (g.N / (128 / hpw)) * g.mblk));
  hipEvent_t e0, e1;
  const bool prof = fh_prof_pair(0, &e0, &e1);
#define AP_LAUNCH(HPW)                                                                                                                              \
  do {                                                                                                                                              \
    if (prof) hipExtLaunchKernelGGL((fmt_attnproj_kernel<T, 3, HPW>), grid, dim3(512), 0, s, e0, e1, 0, h->qkv16, 3 * h->D, g, h->ntok, h->D, c.attn_window); \
    else hipLaunchKernelGGL((fmt_attnproj_kernel<T, 3, HPW>), grid, dim3(512), 0, s, h->qkv16, 3 * h->D, g, h->ntok, h->D, c.attn_window);         \
  } while (0)
  if (hpw == 2) AP_LAUNCH(2);
  else AP_LAUNCH(1);
#undef AP_LAUNCH
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

// Modulation half of the evaluations [e0, e0 + n) of a window: depends only on t and the window's conditions, NOT on x
// (c = t_emb + c_embedder([wr, wa, we]), FMT.py:333-335; adaLN_modulation = Linear(SiLU(c)), FMT.py:163-166, 187-190), so it is
// taken out of the Euler step:
//   sc[z]     = silu(t_emb[e0 + z] + c_cond)                          one launch, blockIdx.y = z
//   modall[z] = sc[z] @ W_adaLN_all^T + b   (Mmod x Ntot fp32 per z)  ONE GEMM launch of n row batches against the 105 MB of
//                                                                     weights (every block's adaLN projection + the head's)
// The step chain then reads its 37 MB slab and no longer streams those weights, which leaves the 208 MB of block weights
// alone in the 256 MB Infinity Cache.  FLOAT_FMT_HOIST=0 launches the same kernel once per evaluation (n = 1) instead
// (bitwise the same numbers; the A/B switch of the measurement).
constexpr int kScSteps = 64;  // evaluations per modulation batch; longer grids run in batches of this many
int g_fmt_hoist = 1;
int g_fmt_zgroup = 0;  // FLOAT_FMT_ZGROUP: column blocks of an XCD that share activation tiles through L2; 0 = per kernel: 4 (fmt_gemm_wide_kernel), 2 (fmt_gemm_dma_kernel)

static int stream_priority(hipStream_t s) {
  int prio = 0;
  if (hipStreamGetPriority(s, &prio) != hipSuccess) {
    (void)hipGetLastError();
    prio = 0;
  }
  return prio;
}

template <class T>
int run_mod_all(float_fmt* h, int M, int e0, int n, hipStream_t s) {
  const int D = h->D;
  FH_REQUIRE(n >= 1 && n <= kScSteps, "modulation batch of %d evaluations (max %d)", n, kScSteps);
  // dense rows for the persistent kernel (row z * M + r of one packed image), else one padded image per evaluation
  // Not on a stream of non-default priority: with the chain on a HIGH-priority stream beside a decoder on a second stream
  // (FLOAT_AMD_OVERLAP=prio) every window after the first came out wrong, and only with this kernel in the chain (round 6:
  // tools/probes/overlap_check.py; the kernel alone passes the same stress in tools/probes/gemm_big_lab.hip, cause not found -
  // DESIGN.md section 7).  fmt_gemm_dma_kernel gives the same numbers bit for bit.
  const int prio = (s == h->cap_stream && s) ? h->cap_prio : stream_priority(s);
  const bool big = !T::is32 && g_fmt_wide && prio == 0 && big_shape_ok(n * M, h->adaln_all.N, h->adaln_all.K, h->n_cu);
  hipLaunchKernelGGL((fmt_silu_c_kernel<T>), dim3((M * D / 8 + 255) / 256, n), dim3(256), 0, s, h->sc16, h->temb + (size_t)e0 * D,
                     h->ccond, M, D, (size_t)h->Mpad * D, big ? M : 0, h->sat);
  h->mod_zs = big ? (size_t)M * h->Ntot : (size_t)h->Mmod * h->Ntot;
  if constexpr (!T::is32) {
    if (big) return launch_big4<T>(h->sc16, h->adaln_all, h->modall, n * M, h->Ntot, h->n_cu, false, s);
  }
  GemmArgs g = base_args(h->sc16, h->adaln_all, M);
  g.sat = h->sat;
  g.out_f32 = h->modall;
  g.ldo = h->Ntot;
  g.zcount = n;
  const bool dma = (g_fmt_wide_variant == 6 || g_fmt_wide_variant == 7) && (M + 15) / 16 > 4 && dma_shape_ok(g, 320);
  g.zgroup = g_fmt_zgroup > 0 ? g_fmt_zgroup : (dma ? 2 : 4);
  g.a_zstride = (size_t)h->Mpad * D;
  g.o_zstride = (size_t)h->Mmod * h->Ntot;
  if constexpr (!T::is32) {
    if (g_fmt_wide && g.N % 128 == 0 && g.K % 128 == 0) return launch_wide<T>(g, false, s);
  }
  for (int z = 0; z < n; ++z) {  // shapes the wide kernel does not tile: the generic GEMM, one batch at a time
    GemmArgs gz = g;
    gz.A = g.A + (size_t)z * g.a_zstride * (sizeof(typename T::elem) / sizeof(u16));
    gz.out_f32 = g.out_f32 + (size_t)z * g.o_zstride;
    gz.zcount = 0;
    int rc = run_gemm<T, EPI_F32>(gz, s);
    if (rc) return rc;
  }
  return FLOAT_OK;
}

// ---- the persistent evaluation kernel (fmt_mega_kernel): stage table of run_blocks' chain for one clip
constexpr int kMegaWgs = 256, kMegaSmem = 8 * 48 * 64 * 4;
MegaSync mega_sync_of(const float_fmt* h) {
  unsigned* m = h->mega_sync;
  static const int wg = getenv("FLOAT_FMT_MEGA_STAMP_WG") ? atoi(getenv("FLOAT_FMT_MEGA_STAMP_WG")) : 0;
  return MegaSync{m, m + 8 * 32, m + 9 * 32, m + 17 * 32, m + 18 * 32, h->mega_err_host, reinterpret_cast<unsigned long long*>(m + 20 * 32), (unsigned)wg};
}
// The persistent kernel's barrier watchdog, looked at by EVERY FMT call of the handle before it queues new work (the flag is
// host-mapped: no copy, no synchronisation when it is clear): a timeout in an earlier call means that call's results are
// invalid.  This call fails with the message; the device is drained, the cached window graphs (they hold the persistent
// kernel) are destroyed, the barrier words are cleared (their generation counters are out of step for good otherwise) and the
// handle runs the launch chain from here on.
int mega_poll(float_fmt* h) {
  if (!h->mega_err_host || *reinterpret_cast<volatile unsigned*>(h->mega_err_host) == 0u) return FLOAT_OK;
  FH_CHECK_HIP(hipDeviceSynchronize());  // graphs may be queued on other streams than the caller's
  for (auto& gr : h->graphs) (void)hipGraphExecDestroy(gr.exec);
  h->graphs.clear();
  if (h->mega_sync) FH_CHECK_HIP(hipMemset(h->mega_sync, 0, (size_t)(32 * 20) * sizeof(unsigned)));
  *reinterpret_cast<volatile unsigned*>(h->mega_err_host) = 0u;
  h->mega_on = 0;
  h->job.active = false;
  fh_set_error("fmt_mega_kernel: a grid barrier timed out in an earlier call (not all %d workgroups were resident) - the results of "
               "that call are invalid; the handle falls back to the launch chain (FLOAT_FMT_MEGA=0 selects it from the start)", 256);
  return FLOAT_E_HIP;
}
// The chain's shapes this kernel is built for: one clip, 3 CFG rows of 60 tokens (M = 180), dim_h 1024, the default launch
// options - i.e. exactly the tilings run_blocks would pick.  Anything else keeps the launch chain.
template <class T>
bool mega_shape_ok(const float_fmt* h, int nclip, int bc) {
  if (T::is32 || !h->mega_on || nclip != 1 || bc != 3 || h->D != 1024 || h->cfg.heads != 8 || h->n_cu < kMegaWgs) return false;
  if (attnproj_hpw(h) || g_fmt_fc2_split != 4 || g_fmt_proj_split != 0) return false;
  const int M = bc * h->ntok;
  auto is = [](Tiling t, int a, int b, int c) { return t.mtw == a && t.nt == b && t.nw == c; };
  const Blk& B = h->blk[0];
  return (M + 15) / 16 == 12 && (h->ntok + 15) / 16 == 4 && h->x_embed.K % 256 == 0 && h->final_lin.K % 256 == 0 &&
         is(pick_tiling(M, B.qkv.N, B.qkv.K, false), 3, 4, 8) && is(pick_tiling(M, B.proj.N, B.proj.K, false), 3, 1, 8) &&
         is(pick_tiling(M, B.fc1.N, B.fc1.K, false), 3, 4, 8) && is(pick_tiling(M, B.fc2.N * 4, B.fc2.K / 4, false), 3, 4, 8) &&
         B.fc2.K % 512 == 0;
}

template <class T>
int build_mega(float_fmt* h, int bc) {
  float_fmt::MegaPlan& P = h->mega[bc];
  P.tried = 1;
  {
    // every one of the 256 workgroups must be resident at once (a plain launch: nothing else checks it)
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(fmt_mega_kernel<T>), 512, kMegaSmem) != hipSuccess) {
      (void)hipGetLastError();
      per_cu = 0;
    }
    if (per_cu * h->n_cu < kMegaWgs) {
      h->mega_on = 0;  // the launch chain
      return FLOAT_OK;
    }
  }
  const float_fmt_cfg_t& c = h->cfg;
  const int D = h->D, ntok = h->ntok, M = bc * ntok;
  std::vector<MegaStage> st;
  // A operands of the GEMM stages: one buffer per producing stage, written once per launch (see fmt_gemm_body, ldA)
  const size_t esz = sizeof(typename T::elem) / sizeof(u16);
  const size_t n_h = (size_t)h->Mpad * D * esz, n_hid = (size_t)h->Mpad * c.mlp_hidden * esz;
  if (!h->mega_ws) {
    int rc0 = h->pool.alloc(&h->mega_ws, (size_t)c.depth * (3 * n_h + n_hid), true);
    if (rc0) return rc0;
  }
  auto ws_h1 = [&](int b) { return h->mega_ws + (size_t)b * (3 * n_h + n_hid); };
  auto ws_h2 = [&](int b) { return ws_h1(b) + n_h; };
  auto ws_att = [&](int b) { return ws_h1(b) + 2 * n_h; };
  auto ws_hid = [&](int b) { return ws_h1(b) + 3 * n_h; };
  auto gemm_stage = [&](int kind, GemmArgs g, int mtw, int nt) {
    MegaStage m;
    memset(&m, 0, sizeof(m));
    m.kind = kind;
    g.sat = h->sat;
    g.mblk = ((g.M + 15) / 16 + mtw - 1) / mtw;
    if (g.ksplit < 1) g.ksplit = 1;
    m.g = g;
    m.nblk = (unsigned)((g.N / (nt * 16)) * g.mblk * g.ksplit);
    return m;
  };
  auto ln_stage = [&](int b_mod, int which, int ks, const float* bias, int gate_col, const Lin* next, int touch_bit, u16* out, int perm) {
    MegaStage m;
    memset(&m, 0, sizeof(m));
    m.kind = MS_LN;
    m.nblk = (unsigned)(((M + 63) / 64) * 64);
    m.shift_off = (long long)b_mod * 6 * D + (long long)which * D;
    m.scale_off = m.shift_off + D;
    m.ks = ks;
    m.red_bias = bias;
    m.red_gate_off = (long long)b_mod * 6 * D + (long long)gate_col * D;
    m.ln_out = out ? out : h->h16;
    m.perm = perm;
    if (next && (g_fmt_touch & (1 | touch_bit))) m.pf = make_touch(*next, M, 0, (m.nblk / 8) * 64, 6);
    return m;
  };
  {  // x_embedder + pos_embed (run_blocks): 8 K-splitting waves here instead of 4 (every stage runs the 512-thread workgroup)
    GemmArgs g = base_args(h->xin16, h->x_embed, ntok);
    g.out_f32 = h->xres;
    g.ldo = D;
    g.pos = h->pos;
    g.bc = bc;
    g.ntok = ntok;
    st.push_back(gemm_stage(MS_XEMBED, g, 4, 1));
  }
  for (int b = 0; b < c.depth; ++b) {
    const Blk& B = h->blk[b];
    // LN1: folds the previous block's fc2 slabs (gate_mlp of block b - 1)
    if (b == 0) st.push_back(ln_stage(b, 0, 0, nullptr, 0, &B.qkv, 128, ws_h1(b), 0));
    else {
      MegaStage m = ln_stage(b, 0, 4, h->blk[b - 1].fc2.b, 0, &B.qkv, 128, ws_h1(b), 0);
      m.red_gate_off = (long long)(b - 1) * 6 * D + 5LL * D;
      st.push_back(m);
    }
    {
      GemmArgs g = base_args(ws_h1(b), B.qkv, M);
      g.out16 = h->qkv16;
      g.ldo16 = 3 * D;
      if (g_fmt_touch & 8) g.touch = make_touch(B.proj, M, 0, gemm_lanes_per_xcd(M, g.N, g.K), 2);
      st.push_back(gemm_stage(MS_QKV, g, 3, 4));
    }
    {
      MegaStage m;
      memset(&m, 0, sizeof(m));
      m.kind = MS_ATTN;
      m.nblk = (unsigned)(c.heads * ((M + 7) / 8));
      if (g_fmt_touch & 2) m.pf = make_touch(B.proj, M, 0, (m.nblk / 8) * 128, 2);
      m.att_out = ws_att(b);
      st.push_back(m);
    }
    {
      GemmArgs g = base_args(ws_att(b), B.proj, M);
      g.out_f32 = h->xres;
      g.ldo = D;
      g.ldg = h->Ntot;
      if (g_fmt_touch & 16) g.touch = make_touch(B.fc1, M, 0, gemm_lanes_per_xcd(M, g.N, g.K), 2);
      MegaStage m = gemm_stage(MS_PROJ, g, 3, 1);
      m.gate_off = (long long)b * 6 * D + 2LL * D;
      st.push_back(m);
    }
    st.push_back(ln_stage(b, 3, 0, nullptr, 0, &B.fc1, 64, ws_h2(b), 0));
    {
      GemmArgs g = base_args(ws_h2(b), B.fc1, M);
      g.out16 = ws_hid(b);
      g.ldo16 = B.fc2.K / 32;
      if (g_fmt_touch & 4) g.touch = make_touch(B.fc2, M, 4, gemm_lanes_per_xcd(M, g.N, g.K), 2);
      st.push_back(gemm_stage(MS_FC1, g, 3, 4));
    }
    {
      GemmArgs g = base_args(ws_hid(b), B.fc2, M);
      g.ksplit = 4;
      g.out_f32 = h->slab;
      g.ldo = g.N;
      g.slab_stride = (size_t)h->Mpad * g.N;
      if (g_fmt_touch & 32) {
        const unsigned lanes = gemm_lanes_per_xcd(M, g.N * 4, g.K / 4);
        if (b + 1 < c.depth) g.touch = make_touch(h->blk[b + 1].qkv, M, 0, lanes, 2);
        else g.touch = make_touch(h->final_lin, M, 0, lanes, 2, 1);
      }
      st.push_back(gemm_stage(MS_FC2, g, 3, 4));
    }
  }
  {
    const int nblk = (ntok + 15) / 16, seqs = bc;
    MegaStage m = ln_stage(c.depth, 0, 4, h->blk[c.depth - 1].fc2.b, 0, nullptr, 0, h->hfin16, seqs * 16);
    m.red_gate_off = (long long)(c.depth - 1) * 6 * D + 5LL * D;
    st.push_back(m);
    GemmArgs g = base_args(h->hfin16, h->final_lin, nblk * seqs * 16);
    g.tokblk = 1;
    g.nclip = 1;
    g.bc = bc;
    g.ntok = ntok;
    g.n_prev = c.n_prev;
    g.xcur = h->xcur;
    g.xin16 = h->xin16;
    g.ldx = h->Kx / 32;
    st.push_back(gemm_stage(MS_HEAD, g, bc, 1));
  }
  for (const MegaStage& m : st) FH_REQUIRE(m.nblk <= (unsigned)kMegaWgs, "persistent kernel: a stage needs %u workgroups", m.nblk);
  int rc;
  if (!h->mega_sync && (rc = h->pool.alloc(&h->mega_sync, 32 * 20 + 2 * 3 * 64, true))) return rc;  // + stamps of <= 64 stages
  if (!h->mega_err_host) {
    FH_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->mega_err_host), 64, hipHostMallocMapped));
    memset(h->mega_err_host, 0, 64);
  }
  if ((rc = h->pool.alloc(&P.dev, st.size(), false))) return rc;
  FH_CHECK_HIP(hipMemcpy(P.dev, st.data(), st.size() * sizeof(MegaStage), hipMemcpyHostToDevice));
  P.nstage = (int)st.size();
  P.bc = bc;
  P.ctx = MegaCtx{h->xres, h->qkv16, h->slab, (size_t)h->Mpad * D, M, D, ntok, h->Ntot, c.attn_window, c.heads, h->sat};
  return FLOAT_OK;
}

template <class T>
int run_mega(float_fmt* h, int bc, const float* modbuf, bool euler, float dt, float a, float r, float e, hipStream_t s, float* vout_to) {
  float_fmt::MegaPlan& P = h->mega[bc];
  MegaDyn d{modbuf, dt, a, r, e, euler ? 1 : 0, vout_to ? vout_to : h->vout};
  hipEvent_t e0, e1;
  if (fh_prof_pair(0, &e0, &e1))
    hipExtLaunchKernelGGL((fmt_mega_kernel<T>), dim3(kMegaWgs), dim3(512), kMegaSmem, s, e0, e1, 0, P.dev, P.nstage, d, P.ctx, mega_sync_of(h));
  else hipLaunchKernelGGL((fmt_mega_kernel<T>), dim3(kMegaWgs), dim3(512), kMegaSmem, s, P.dev, P.nstage, d, P.ctx, mega_sync_of(h));
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

// Block chain of an evaluation on the rows staged in the workspace, using the modulations in modbuf.
// euler: update xcur/xin16 with dt, else write vout.
template <class T>
int run_blocks(float_fmt* h, int nclip, int bc, const float* modbuf, bool euler, float dt, float a, float r, float e, hipStream_t s,
               float* vout_to = nullptr) {
  const float_fmt_cfg_t& c = h->cfg;
  const int D = h->D, ntok = h->ntok, M = nclip * bc * ntok;
  int rc;
  if constexpr (!T::is32) {
    if (bc <= 4 && mega_shape_ok<T>(h, nclip, bc)) {
      if (h->mega[bc].dev) return run_mega<T>(h, bc, modbuf, euler, dt, a, r, e, s, vout_to);  // table built at create
    }
  }
  // x_embedder + pos_embed; the CFG rows share x, so 60 rows are computed and broadcast
  {
    GemmArgs g = base_args(h->xin16, h->x_embed, nclip * ntok);
    g.sat = h->sat;
    g.out_f32 = h->xres;
    g.ldo = D;
    g.pos = h->pos;
    g.bc = bc;
    g.ntok = ntok;
    if ((rc = run_gemm<T, EPI_XEMBED>(g, s))) return rc;
  }
  PendingRed pend;  // residual update left to the next LayerNorm launch
  // stacked clips: the row-blocked LDS-DMA tile (fmt_rb_kernels.hpp); its launches carry no touch descriptors; the LayerNorm in
  // front of qkv / fc1 pulls the first stages of their weights into the XCDs' L2s (make_touch_rb)
  RbPlan rb_qkv, rb_proj, rb_fc1, rb_fc2;
  if constexpr (!T::is32) {
    rb_qkv = pick_rb(RB_QKV, M), rb_proj = pick_rb(RB_PROJ, M), rb_fc1 = pick_rb(RB_FC1, M), rb_fc2 = pick_rb(RB_FC2, M);
  }
  auto split_ok = [&](int ks, const Lin& L) { return (ks == 1 || ks == 2 || ks == 4) && L.K % (128 * ks) == 0; };
  for (int b = 0; b < c.depth; ++b) {
    const float* mod = modbuf + (size_t)b * 6 * D;  // shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
    const Blk& B = h->blk[b];
    if ((rc = launch_lnmod<T>(h, M, mod, mod + D, s, &pend, &B.qkv, nullptr, 0, 128, rb_qkv.shape))) return rc;
    {
      GemmArgs g = base_args(h->h16, B.qkv, M);
      g.sat = h->sat;
      g.out16 = h->qkv16;
      g.ldo16 = 3 * D;
      if ((g_fmt_touch & 8) && attnproj_hpw(h)) g.touch = make_touch(B.proj, M, c.heads / attnproj_hpw(h), gemm_lanes_per_xcd(M, g.N, g.K), 2, 8 / attnproj_hpw(h));  // k-slices <-> XCDs as the fused launch decodes them
      else if ((g_fmt_touch & 8) && !split_ok(g_fmt_proj_split, B.proj)) g.touch = make_touch(B.proj, M, 0, gemm_lanes_per_xcd(M, g.N, g.K), 2);
      static const LayerPlan plan("FLOAT_FMT_PLAN_QKV");
      if (rb_qkv.shape >= 0) rc = launch_rbs<T, EPI_T16>(g, rb_qkv.shape, false, s);
      else rc = run_gemm<T, EPI_T16>(g, s, false, &plan);
      if (rc) return rc;
    }
    if (rb_proj.shape >= 0) {
      launch_attn<T>(h, M, nullptr, s);
      GemmArgs gp = base_args(h->att16, B.proj, M);
      gp.ksplit = rb_proj.ksplit;
      gp.out_f32 = h->slab;
      gp.ldo = gp.N;
      gp.slab_stride = (size_t)h->Mpad * gp.N;
      if ((rc = launch_rbs<T, EPI_PARTIAL>(gp, rb_proj.shape, false, s))) return rc;
      pend.ks = rb_proj.ksplit;
      pend.red = LnRed{h->slab, (size_t)h->Mpad * D, B.proj.b, mod + 2 * D};
    } else if (attnproj_hpw(h)) {
      if ((rc = launch_attnproj<T>(h, M, B.proj, s))) return rc;
      pend.ks = c.heads / attnproj_hpw(h);
      pend.red = LnRed{h->slab, (size_t)h->Mpad * D, B.proj.b, mod + 2 * D};
    } else if (split_ok(g_fmt_proj_split, B.proj)) {
      launch_attn<T>(h, M, nullptr, s);
      GemmArgs gp = base_args(h->att16, B.proj, M);
      if ((rc = run_gemm_partial<T>(h, gp, g_fmt_proj_split, s))) return rc;
      pend.ks = g_fmt_proj_split;
      pend.red = LnRed{h->slab, (size_t)h->Mpad * D, B.proj.b, mod + 2 * D};
    } else {
      launch_attn<T>(h, M, (g_fmt_touch & 2) ? &B.proj : nullptr, s);
      GemmArgs g = base_args(h->att16, B.proj, M);
      g.sat = h->sat;
      g.out_f32 = h->xres;
      g.ldo = D;
      g.gate = mod + 2 * D;
      g.ldg = h->Ntot;
      if (g_fmt_touch & 16) g.touch = make_touch(B.fc1, M, 0, gemm_lanes_per_xcd(M, g.N, g.K), 2);
      static const LayerPlan plan("FLOAT_FMT_PLAN_PROJ");
      if ((rc = run_gemm<T, EPI_GATE_RES>(g, s, false, &plan))) return rc;
    }
    if ((rc = launch_lnmod<T>(h, M, mod + 3 * D, mod + 4 * D, s, &pend, &B.fc1, nullptr, 0, 64, rb_fc1.shape))) return rc;
    {
      GemmArgs g = base_args(h->h16, B.fc1, M);
      g.sat = h->sat;
      g.out16 = h->hid16;
      g.ldo16 = B.fc2.K / 32;  // packed for fc2
      if ((g_fmt_touch & 4) && rb_fc2.shape < 0)
        g.touch = make_touch(B.fc2, M, split_ok(g_fmt_fc2_split, B.fc2) ? g_fmt_fc2_split : 0, gemm_lanes_per_xcd(M, g.N, g.K), 2);
      static const LayerPlan plan("FLOAT_FMT_PLAN_FC1");
      if (rb_fc1.shape >= 0) rc = launch_rbs<T, EPI_GELU_P16>(g, rb_fc1.shape, false, s);
      else rc = run_gemm<T, EPI_GELU_P16>(g, s, false, &plan);
      if (rc) return rc;
    }
    if (rb_fc2.shape >= 0) {
      GemmArgs g = base_args(h->hid16, B.fc2, M);
      g.ksplit = rb_fc2.ksplit;
      g.out_f32 = h->slab;
      g.ldo = g.N;
      g.slab_stride = (size_t)h->Mpad * g.N;
      if ((rc = launch_rbs<T, EPI_PARTIAL>(g, rb_fc2.shape, false, s))) return rc;
      pend.ks = rb_fc2.ksplit;
      pend.red = LnRed{h->slab, (size_t)h->Mpad * D, B.fc2.b, mod + 5 * D};
    } else if (split_ok(g_fmt_fc2_split, B.fc2)) {
      GemmArgs g = base_args(h->hid16, B.fc2, M);
      g.sat = h->sat;
      if (g_fmt_touch & 32) {
        const unsigned lanes = gemm_lanes_per_xcd(M, g.N * g_fmt_fc2_split, g.K / g_fmt_fc2_split);
        if (b + 1 < c.depth) g.touch = make_touch(h->blk[b + 1].qkv, M, 0, lanes, 2);
        else g.touch = make_touch(h->final_lin, M, 0, lanes, 2, 1);  // the head GEMM runs 16-column workgroups
      }
      static const LayerPlan plan("FLOAT_FMT_PLAN_FC2");
      if ((rc = run_gemm_partial<T>(h, g, g_fmt_fc2_split, s, &plan))) return rc;
      pend.ks = g_fmt_fc2_split;
      pend.red = LnRed{h->slab, (size_t)h->Mpad * D, B.fc2.b, mod + 5 * D};
    } else {
      GemmArgs g = base_args(h->hid16, B.fc2, M);
      g.sat = h->sat;
      g.out_f32 = h->xres;
      g.ldo = D;
      g.gate = mod + 5 * D;
      g.ldg = h->Ntot;
      if ((rc = run_gemm<T, EPI_GATE_RES>(g, s))) return rc;
    }
  }
  {
    const float* mod = modbuf + (size_t)c.depth * 6 * D;  // shift, scale (FMT.py:196)
    // head: token-blocked rows (every CFG row of 16 tokens in one workgroup: 4 x 32 workgroups of bc row tiles) unless the
    // CFG batch is not one of the combine's shapes; then all rows per workgroup (32 workgroups)
    static const bool tokblk_on = !getenv("FLOAT_FMT_NO_TOKBLK");
    const bool tokblk = (tokblk_on || nclip > 1) && (bc == 1 || bc == 3 || bc == 4) && h->final_lin.K % 256 == 0;
    FH_REQUIRE(tokblk || nclip == 1, "batched sampling needs the token-blocked head GEMM");
    const int nblk = (ntok + 15) / 16, seqs = nclip * bc;
    if ((rc = launch_lnmod<T>(h, M, mod, mod + D, s, &pend, nullptr, tokblk ? h->hfin16 : nullptr, tokblk ? seqs * 16 : 0))) return rc;
    GemmArgs g = base_args(tokblk ? h->hfin16 : h->h16, h->final_lin, tokblk ? nblk * seqs * 16 : M);
    g.sat = h->sat;
    g.tokblk = tokblk ? 1 : 0;
    g.nclip = nclip;
    g.bc = bc;
    g.ntok = ntok;
    g.n_prev = c.n_prev;
    g.a_cfg = a;
    g.r_cfg = r;
    g.e_cfg = e;
    g.dt = dt;
    if (euler) {
      g.xcur = h->xcur;
      g.xin16 = h->xin16;
      g.ldx = h->Kx / 32;
    } else {
      g.vout = vout_to ? vout_to : h->vout;
    }
    // token-blocked: a workgroup = the bc row tiles of one (token block, clip) pair; row blocks = token blocks x clips
    if (tokblk) rc = launch_gemm<T, EPI_CFG>(g, bc, 1, 8, false, s);
    else rc = run_gemm<T, EPI_CFG>(g, s, true);
    if (rc) return rc;
  }
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

// torch.linspace(0, 1, n) in fp32: symmetric evaluation around the midpoint.
void linspace01(int n, std::vector<float>* ts) {
  ts->resize(n);
  if (n == 1) {
    (*ts)[0] = 0.f;
    return;
  }
  const float step = 1.0f / (float)(n - 1);
  const int half = n / 2;
  for (int i = 0; i < n; ++i) (*ts)[i] = (i < half) ? (0.f + step * (float)i) : (1.0f - step * (float)(n - 1 - i));
}

// t-embedding MLP (FMT.py:128-131) for every evaluation time of a window, rows = evaluations.  The times are formed on the
// device (fmt_tsin_kernel) from (nfe, stage offsets), so no host buffer is read by the stream: the call can be captured.
struct TimeSpec {
  float t;        // nfe == 0: the one explicit time of float_fmt_eval
  int nfe, stages;
  float c[4];     // stage offsets of the Runge-Kutta scheme (Euler: {0})
};
template <class T>
int prepare_time(float_fmt* h, const TimeSpec& ts, int n, hipStream_t s) {
  int rc;
  hipLaunchKernelGGL((fmt_tsin_kernel<T>), dim3(n), dim3(256), 0, s, h->tsin16, h->freqs, n, ts.t, ts.nfe, ts.stages, ts.c[0],
                     ts.c[1], ts.c[2], ts.c[3]);
  GemmArgs g = base_args(h->tsin16, h->t0, n);
  g.sat = h->sat;
  g.out16 = h->th16;
  g.ldo16 = h->t2.K / 32;
  if ((rc = launch_gemm<T, EPI_SILU_P16>(g, 4, 1, T::is32 ? 4 : pick_nw(g.K, 0), false, s))) return rc;
  GemmArgs g2 = base_args(h->th16, h->t2, n);
  g2.sat = h->sat;
  g2.out_f32 = h->temb;
  g2.ldo = h->D;
  if ((rc = launch_gemm<T, EPI_F32>(g2, 4, 1, T::is32 ? 4 : pick_nw(g2.K, 0), false, s))) return rc;
  return FLOAT_OK;
}

struct CfgMode {
  int bc;
  unsigned wr_mask, wa_mask, we_mask;
  int nclip = 1;  // clips stacked along the rows
};

CfgMode cfg_mode(float a, float r, float e, int include_r) {
  if (a == 1.0f && r == 1.0f && e == 1.0f) return {1, 1u, 1u, 1u, 1};  // FMT.py:346,400-401
  if (!include_r) return {3, 0b111u, 0b110u, 0b010u, 1};                // [null_wa,wa,wa] [null_we,we,null_we]
  return {4, 0b1110u, 0b1100u, 0b0100u, 1};                             // FMT.py:382-384
}

// Stage the conditions of one window (device pointers) and compute c_cond = c_embedder([wr,wa,we]).
template <class T>
int stage_window(float_fmt* h, const CfgMode& m, const float* x0, const float* wa, const float* wr, const float* we,
                 int we_len, const float* prev_x, const float* prev_wa, const float* prev_we, hipStream_t s) {
  const float_fmt_cfg_t& c = h->cfg;
  const int M = m.nclip * m.bc * h->ntok;
  hipLaunchKernelGGL((fmt_build_cond_kernel<T>), dim3(M), dim3(256), 0, s, h->cond16, h->Kc, m.bc, h->ntok, c.n_prev,
                     c.dim_w, c.dim_a, c.dim_e, wr, wa, prev_wa, we, we_len, prev_we, m.wr_mask, m.wa_mask, m.we_mask, h->sat);
  GemmArgs g = base_args(h->cond16, h->c_embed, M);
  g.sat = h->sat;
  g.out_f32 = h->ccond;
  g.ldo = h->D;
  int rc;
  if ((rc = run_gemm<T, EPI_F32>(g, s))) return rc;
  const int n = m.nclip * h->ntok * c.dim_w;
  hipLaunchKernelGGL((fmt_init_x_kernel<T>), dim3((n + 255) / 256), dim3(256), 0, s, h->xcur, h->xin16, h->Kx / 32, x0, prev_x,
                     c.n_prev, c.n_cur, c.dim_w, m.nclip, h->sat);
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

int check_common(float_fmt* h, const void* we, int we_len, const void* prev_we) {
  FH_REQUIRE(h != nullptr, "null FMT handle");
  FH_REQUIRE(we_len == 1 || we_len == h->cfg.n_cur,
             "Dynamic emotion latent `we` time dimension (%d) does not match audio latent `wa` time dimension (%d).",
             we_len, h->cfg.n_cur);
  FH_REQUIRE(!(we_len > 1 && prev_we == nullptr),
             "`we` is dynamic (T>1), but prev_we was not provided with prev_x/prev_wa.");
  (void)we;
  return FLOAT_OK;
}

// Fixed-grid explicit Runge-Kutta schemes of torchdiffeq's solver list (reference src/nodes/__init__.py:15-23):
// stage times t0 + c_j dt, stage inputs y0 + dt sum_m a[j][m] k_m, update dt sum_j b_j k_j.  torchdiffeq is not
// installed where this is built, so these are its published rules (midpoint; "rk4" = the 3/8-rule
// rk4_alt_step_func; heun2 / heun3 Butcher tableaux) - parity with the package itself is unpinned.
struct Tableau {
  int s;
  float c[4], a[4][3], b[4];
};
const Tableau& tableau(int method) {
  static const Tableau T[5] = {
      {1, {0.f}, {{0.f}}, {1.f}},                                                                        // euler
      {2, {0.f, 0.5f}, {{0.f}, {0.5f}}, {0.f, 1.f}},                                                     // midpoint
      {4, {0.f, 1.f / 3, 2.f / 3, 1.f}, {{0.f}, {1.f / 3}, {-1.f / 3, 1.f}, {1.f, -1.f, 1.f}}, {0.125f, 0.375f, 0.375f, 0.125f}},  // rk4 (3/8)
      {2, {0.f, 1.f}, {{0.f}, {1.f}}, {0.5f, 0.5f}},                                                     // heun2
      {3, {0.f, 1.f / 3, 2.f / 3}, {{0.f}, {1.f / 3}, {0.f, 2.f / 3}}, {0.25f, 0.f, 0.75f}},             // heun3
  };
  return T[method];
}

// evaluation times of a window: Euler -> the grid itself; RK -> t_i + c_j (t_{i+1} - t_i), step-major
TimeSpec time_spec(int method, int nfe) {
  const Tableau& tb = tableau(method);
  TimeSpec ts{};
  ts.nfe = nfe;
  ts.stages = tb.s;
  for (int j = 0; j < 4; ++j) ts.c[j] = tb.c[j];
  return ts;
}
int n_evals(int method, int nfe) { return (nfe - 1) * tableau(method).s; }

// The evaluations of one window, eager and single-stream (this is also what gets captured into the window's hipGraph and
// the profiling path).  Modulations come in batches of up to kScSteps evaluations (run_mod_all); FLOAT_FMT_HOIST=0 makes the
// batch one evaluation long.
template <class T>
int run_window_steps(float_fmt* h, const CfgMode& m, int nfe, const std::vector<float>& ts, float a, float r, float e,
                     hipStream_t s) {
  const Tableau& tb = tableau(h->method);
  const float_fmt_cfg_t& c = h->cfg;
  const size_t kstride = (size_t)h->Bmax * kMaxTok * c.dim_w;  // one stage's velocities: [clip][ntok][dim_w]
  const int n = m.nclip * c.n_cur * c.dim_w, rows = m.nclip * m.bc * h->ntok;
  const int nev = n_evals(h->method, nfe), batch = g_fmt_hoist ? kScSteps : 1;
  const bool euler = h->method == FLOAT_ODE_EULER;
  int rc;
  for (int ev = 0; ev < nev; ++ev) {
    const int i = ev / tb.s, j = ev - i * tb.s, z = ev % batch;
    const float dt = ts[i + 1] - ts[i];
    if (z == 0 && (rc = run_mod_all<T>(h, rows, ev, std::min(batch, nev - ev), s))) return rc;
    const float* mod = h->modall + (size_t)z * h->mod_zs;  // the layout run_mod_all chose for this batch
    if (euler) {
      if ((rc = run_blocks<T>(h, m.nclip, m.bc, mod, true, dt, a, r, e, s))) return rc;
      continue;
    }
    // fixed-grid explicit Runge-Kutta: stage j evaluates at y0 + dt sum_m a[j][m] k_m, the update is dt sum_j b_j k_j
    if (j > 0) {
      hipLaunchKernelGGL((fmt_rk_combine_kernel<T>), dim3((n + 255) / 256), dim3(256), 0, s, h->xcur, h->kbuf, kstride, j,
                         dt * tb.a[j][0], dt * tb.a[j][1], dt * tb.a[j][2], 0.f, 0, h->xin16, h->Kx / 32, c.n_prev, c.n_cur,
                         c.dim_w, m.nclip, h->sat);
    }
    if ((rc = run_blocks<T>(h, m.nclip, m.bc, mod, false, 0.f, a, r, e, s, h->kbuf + (size_t)j * kstride))) return rc;
    if (j == tb.s - 1) {
      hipLaunchKernelGGL((fmt_rk_combine_kernel<T>), dim3((n + 255) / 256), dim3(256), 0, s, h->xcur, h->kbuf, kstride, tb.s,
                         dt * tb.b[0], dt * tb.b[1], dt * tb.b[2], dt * tb.b[3], 1, h->xin16, h->Kx / 32, c.n_prev, c.n_cur,
                         c.dim_w, m.nclip, h->sat);
    }
  }
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

// Same chain, replayed from a cached hipGraph (all pointers are workspace-internal, so the graph is reusable across windows
// and clips for a given (nfe, method, cfg mode, scales)).  The cache holds kMaxGraphs executables; the least recently used
// one is destroyed when a new key arrives (a caller sweeping a CFG scale would otherwise keep a ~3000-node graph per value).
constexpr size_t kMaxGraphs = 8;
template <class T>
int run_window_steps_graph(float_fmt* h, const CfgMode& m, int we_len, int nfe, const std::vector<float>& ts, float a,
                           float r, float e, hipStream_t s) {
  float_fmt::GraphKey key;
  memset(&key, 0, sizeof(key));
  key.nfe = nfe;
  key.method = h->method;
  key.bc = m.bc;
  key.nclip = m.nclip;
  key.we_len = we_len;
  key.a = a;
  key.r = r;
  key.e = e;
  key.prio = stream_priority(s);
  float_fmt::GraphEntry* hit = nullptr;
  for (auto& g : h->graphs)
    if (g.key == key) hit = &g;
  if (!hit) {
    if (!h->cap_stream) FH_CHECK_HIP(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
    hipGraph_t graph = nullptr;
    h->cap_prio = key.prio;
    FH_CHECK_HIP(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
    int rc = run_window_steps<T>(h, m, nfe, ts, a, r, e, h->cap_stream);
    hipError_t ce = hipStreamEndCapture(h->cap_stream, &graph);
    if (rc) {
      if (graph) (void)hipGraphDestroy(graph);
      return rc;
    }
    FH_CHECK_HIP(ce);
    hipGraphExec_t exec = nullptr;
    FH_CHECK_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    if (h->graphs.size() >= kMaxGraphs) {
      size_t lru = 0;
      for (size_t i = 1; i < h->graphs.size(); ++i)
        if (h->graphs[i].used < h->graphs[lru].used) lru = i;
      // an executable that is still queued must not be destroyed under it - and it may be queued on ANOTHER stream than `s`
      // (the window sampler and the overlapped pipeline run the chain on side streams): eviction is rare (a ninth distinct
      // chain shape), so wait for the whole device
      FH_CHECK_HIP(hipDeviceSynchronize());
      (void)hipGraphExecDestroy(h->graphs[lru].exec);
      h->graphs.erase(h->graphs.begin() + lru);
    }
    h->graphs.push_back({key, exec, 0});
    hit = &h->graphs.back();
  }
  hit->used = ++h->graph_clock;
  FH_CHECK_HIP(hipGraphLaunch(hit->exec, s));
  return FLOAT_OK;
}

template <class T>
int window_impl(float_fmt* h, const float* x0, const float* wa, const float* wr, const float* we, int we_len,
                const float* prev_x, const float* prev_wa, const float* prev_we, int nfe, const std::vector<float>& ts,
                float a, float r, float e, int include_r, hipStream_t s, int nclip = 1) {
  CfgMode m = cfg_mode(a, r, e, include_r);
  m.nclip = nclip;
  // only the all-rows-per-workgroup head (FLOAT_FMT_NO_TOKBLK, a debugging aid) is limited to 15 row tiles
  FH_REQUIRE(!getenv("FLOAT_FMT_NO_TOKBLK") || m.bc * h->ntok <= 240,
             "%d-way CFG of %d tokens is %d rows; the all-rows CFG epilogue GEMM holds at most 240 (15 row tiles)", m.bc, h->ntok,
             m.bc * h->ntok);
  int rc = stage_window<T>(h, m, x0, wa, wr, we, we_len, prev_x, prev_wa, prev_we, s);
  if (rc) return rc;
  if (nfe <= 1) return FLOAT_OK;  // a one-point grid has no evaluation: the sample is x0 (FLOAT.py:188,247-248)
  // A caller that is itself capturing `s` gets the chain launched straight into its capture (a graph cannot be launched or
  // captured from inside another capture); so does the profiling pass, whose events need eager launches.
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cs) != hipSuccess) {
    (void)hipGetLastError();
    cs = hipStreamCaptureStatusNone;
  }
  if (h->cfg.use_graph && !g_fh_profiling && cs == hipStreamCaptureStatusNone)
    return run_window_steps_graph<T>(h, m, we_len, nfe, ts, a, r, e, s);
  return run_window_steps<T>(h, m, nfe, ts, a, r, e, s);
}

template <class T>
int eval_impl(float_fmt* h, float t, const float* x, const float* wa, const float* wr, const float* we, int we_len,
              const float* prev_x, const float* prev_wa, const float* prev_we, float a, float r, float e, int include_r,
              float* out, hipStream_t s) {
  TimeSpec tsp{};
  tsp.t = t;
  int rc = prepare_time<T>(h, tsp, 1, s);
  if (rc) return rc;
  const CfgMode m = cfg_mode(a, r, e, include_r);
  // only the all-rows-per-workgroup head (FLOAT_FMT_NO_TOKBLK, a debugging aid) is limited to 15 row tiles
  FH_REQUIRE(!getenv("FLOAT_FMT_NO_TOKBLK") || m.bc * h->ntok <= 240,
             "%d-way CFG of %d tokens is %d rows; the all-rows CFG epilogue GEMM holds at most 240 (15 row tiles)", m.bc, h->ntok,
             m.bc * h->ntok);
  if ((rc = stage_window<T>(h, m, x, wa, wr, we, we_len, prev_x, prev_wa, prev_we, s))) return rc;
  if ((rc = run_mod_all<T>(h, m.bc * h->ntok, 0, 1, s))) return rc;
  if ((rc = run_blocks<T>(h, 1, m.bc, h->modall, false, 0.f, a, r, e, s))) return rc;
  return dev_copy(out, h->vout, (size_t)h->ntok * h->cfg.dim_w, s);
}

// One window of the auto-regressive loop (FLOAT.py:214-251; nodes_adv.py:605-688).
template <class T>
int sample_window(float_fmt* h, int k, hipStream_t s) {
  const float_fmt_cfg_t& c = h->cfg;
  auto& J = h->job;
  const int L = c.n_cur, P = c.n_prev, Tn = J.T, B = J.B;
  const bool dynamic = J.we_len > 1;
  int rc;
  auto blocks = [](int n) { return dim3((n + 255) / 256); };
  if (k == J.first) {
    if ((rc = prepare_time<T>(h, time_spec(h->method, J.nfe), std::max(1, n_evals(h->method, J.nfe)), s))) return rc;
    // chunk 0 starts from zero history (FLOAT.py:217-219, nodes_adv.py:591-593); a job that starts at a later window
    // (float_fmt_sample_begin_range) from the history its caller hands over, or from zeros too
    if (J.hist_x) rc = dev_copy(h->prev_x, J.hist_x, (size_t)B * P * c.dim_w, s);
    else rc = dev_zero(h->prev_x, (size_t)B * P * c.dim_w, s);
    if (rc) return rc;
    if (J.hist_wa) rc = dev_copy(h->prev_wa, J.hist_wa, (size_t)B * P * c.dim_a, s);
    else rc = dev_zero(h->prev_wa, (size_t)B * P * c.dim_a, s);
    if (rc) return rc;
    if (J.hist_we) rc = dev_copy(h->prev_we, J.hist_we, (size_t)B * P * c.dim_e, s);
    else rc = dev_zero(h->prev_we, (size_t)B * P * c.dim_e, s);
    if (rc) return rc;
  } else if (P > 0) {
    // AR hand-off: last P frames of the previous final sample / (padded) wa window / we window, per clip
    hipLaunchKernelGGL(fmt_tail_kernel, blocks(B * P * c.dim_w), dim3(256), 0, s, h->prev_x, h->xcur, P, L, c.dim_w, B);
    hipLaunchKernelGGL(fmt_tail_kernel, blocks(B * P * c.dim_a), dim3(256), 0, s, h->prev_wa, h->wa_c, P, L, c.dim_a, B);
    if (dynamic) hipLaunchKernelGGL(fmt_tail_kernel, blocks(B * P * c.dim_e), dim3(256), 0, s, h->prev_we, h->we_c, P, L, c.dim_e, B);
  }
  hipLaunchKernelGGL(fmt_slice_pad_kernel, blocks(B * L * c.dim_a), dim3(256), 0, s, h->wa_c, J.wa, k * L, Tn, L, c.dim_a, B);
  if (dynamic) hipLaunchKernelGGL(fmt_slice_pad_kernel, blocks(B * L * c.dim_e), dim3(256), 0, s, h->we_c, J.we, k * L, Tn, L, c.dim_e, B);
  // x0 is copied into the workspace so the window chain only ever sees handle-owned pointers; noise is (windows, clips, L, dim_w)
  if ((rc = dev_copy(h->x0_c, J.noise + (size_t)k * B * L * c.dim_w, (size_t)B * L * c.dim_w, s))) return rc;
  rc = window_impl<T>(h, h->x0_c, h->wa_c, J.wr, dynamic ? h->we_c : J.we, dynamic ? L : 1, h->prev_x, h->prev_wa,
                      dynamic ? h->prev_we : nullptr, J.nfe, J.ts, J.a, J.r, J.e, J.include_r, s, B);
  if (rc) return rc;
  const int rows = (k == J.total - 1) ? (Tn - k * L) : L;  // trim to T (FLOAT.py:252)
  return dev_copy2d(J.r_d + (size_t)k * L * c.dim_w, (size_t)Tn * c.dim_w, h->xcur, (size_t)L * c.dim_w, rows * c.dim_w, B, s);
}

template <class T>
int create_impl(float_fmt* h, const TensorTable& tt) {
  const float_fmt_cfg_t& c = h->cfg;
  const int D = c.dim_h;
  int rc;
  prime_kernels<T>();
  if ((rc = pack_linear<T>(h, tt, {"x_embedder.proj"}, D, c.dim_w, &h->x_embed))) return rc;
  if ((rc = pack_linear<T>(h, tt, {"t_embedder.mlp.0"}, D, 256, &h->t0))) return rc;
  if ((rc = pack_linear<T>(h, tt, {"t_embedder.mlp.2"}, D, D, &h->t2))) return rc;
  if ((rc = pack_linear<T>(h, tt, {"c_embedder"}, D, c.dim_w + c.dim_a + c.dim_e, &h->c_embed))) return rc;
  h->blk.resize(c.depth);
  std::vector<std::string> ada;
  for (int b = 0; b < c.depth; ++b) {
    const std::string p = "blocks." + std::to_string(b) + ".";
    if ((rc = pack_linear<T>(h, tt, {p + "attn.qkv"}, 3 * D, D, &h->blk[b].qkv))) return rc;
    if ((rc = pack_linear<T>(h, tt, {p + "attn.proj"}, D, D, &h->blk[b].proj))) return rc;
    if ((rc = pack_linear<T>(h, tt, {p + "mlp.fc1"}, c.mlp_hidden, D, &h->blk[b].fc1))) return rc;
    if ((rc = pack_linear<T>(h, tt, {p + "mlp.fc2"}, D, c.mlp_hidden, &h->blk[b].fc2))) return rc;
    ada.push_back(p + "adaLN_modulation.1");
  }
  if ((rc = pack_linear<T>(h, tt, ada, 6 * D, D, &h->adaln_all))) return rc;
  // the head's adaLN (2D outputs) rides at the end of the same fused projection
  {
    Lin tail;
    if ((rc = pack_linear<T>(h, tt, {"decoder.adaLN_modulation.1"}, 2 * D, D, &tail))) return rc;
    Lin fused;
    fused.N = h->adaln_all.N + tail.N;
    fused.K = h->adaln_all.K;
    constexpr size_t esz = sizeof(typename T::elem) / sizeof(u16);
    if ((rc = h->pool.alloc(&fused.W, (size_t)fused.N * fused.K * esz, false))) return rc;
    if ((rc = h->pool.alloc(&fused.b, (size_t)fused.N, false))) return rc;
    FH_CHECK_HIP(hipMemcpy(fused.W, h->adaln_all.W, (size_t)h->adaln_all.N * fused.K * esz * sizeof(u16), hipMemcpyDeviceToDevice));
    FH_CHECK_HIP(hipMemcpy(fused.W + (size_t)h->adaln_all.N * fused.K * esz, tail.W, (size_t)tail.N * fused.K * esz * sizeof(u16),
                           hipMemcpyDeviceToDevice));
    FH_CHECK_HIP(hipMemcpy(fused.b, h->adaln_all.b, (size_t)h->adaln_all.N * sizeof(float), hipMemcpyDeviceToDevice));
    FH_CHECK_HIP(hipMemcpy(fused.b + h->adaln_all.N, tail.b, (size_t)tail.N * sizeof(float), hipMemcpyDeviceToDevice));
    h->adaln_all = fused;  // the two source copies stay in the pool until destroy (small: ~100 MB, freed with handle)
  }
  if ((rc = pack_linear<T>(h, tt, {"decoder.linear"}, c.dim_w, D, &h->final_lin))) return rc;
  return FLOAT_OK;
}

}  // namespace


// ---------------------------------------------------------------- GEMM service (fmt_gemm.hpp)
int fmt_pack_linear(DevicePool* pool, int dtype, const TensorTable& tt, const std::vector<std::string>& names, int N_each, int K,
                    FmtLin* out) {
  FH_REQUIRE(dtype == FLOAT_DT_BF16 || dtype == FLOAT_DT_FP16 || dtype == FLOAT_DT_FP32, "the GEMM service: unknown dtype %d", dtype);
  if (dtype == FLOAT_DT_FP32) return pack_linear_pool<FP32>(pool, tt, names, N_each, K, out);
  return dtype == FLOAT_DT_BF16 ? pack_linear_pool<BF16>(pool, tt, names, N_each, K, out)
                                : pack_linear_pool<FP16>(pool, tt, names, N_each, K, out);
}

int fmt_pack_linear_raw(DevicePool* pool, int dtype, const float* w, const float* b, int N, int K, FmtLin* out) {
  std::vector<float> zeros;
  if (!b) {
    zeros.assign(N, 0.f);
    b = zeros.data();
  }
  float_tensor_t t[2];
  memset(t, 0, sizeof(t));
  t[0].name = "raw.weight";
  t[0].data = w;
  t[0].ndim = 2;
  t[0].shape[0] = N;
  t[0].shape[1] = K;
  t[1].name = "raw.bias";
  t[1].data = b;
  t[1].ndim = 1;
  t[1].shape[0] = N;
  TensorTable tt(t, 2);
  return fmt_pack_linear(pool, dtype, tt, {"raw"}, N, K, out);
}

GemmArgs fmt_gemm_args(const u16* A, const FmtLin& L, int M) { return base_args(A, L, M); }

void fmt_gemm_prime(int dtype) {
  if (dtype == FLOAT_DT_BF16) prime_kernels<BF16>();
  else if (dtype == FLOAT_DT_FP32) prime_kernels<FP32>();
  else prime_kernels<FP16>();
}

template <class T>
static int gemm_run_t(int epi, const GemmArgs& g, hipStream_t s) {
  switch (epi) {
    case EPI_F32: return run_gemm<T, EPI_F32>(g, s);
    case EPI_T16: return run_gemm<T, EPI_T16>(g, s);
    case EPI_SILU_P16: return run_gemm<T, EPI_SILU_P16>(g, s);
    case EPI_GELU_P16: return run_gemm<T, EPI_GELU_P16>(g, s);
    case EPI_GELUERF_P16: return run_gemm<T, EPI_GELUERF_P16>(g, s);
    default: fh_set_error("fmt_gemm_run: epilogue %d is not exported", epi); return FLOAT_E_INVALID;
  }
}

int fmt_gemm_run(int dtype, int epi, GemmArgs g, hipStream_t s) {
  if (dtype == FLOAT_DT_FP32) return gemm_run_t<FP32>(epi, g, s);
  return dtype == FLOAT_DT_BF16 ? gemm_run_t<BF16>(epi, g, s) : gemm_run_t<FP16>(epi, g, s);
}

template <class T>
static int debug_impl(float_fmt* h, int what, const float* in, float* out, hipStream_t s) {
  const int D = h->D, ntok = h->ntok;
  if (what == 0) {
    return dev_copy(out, h->pos, (size_t)ntok * D, s);
  }
  if (what == 2) {  // diagnostic builds (-DMEGA_STAMPS): per stage [start, body end, stores drained] of the persistent kernel's last launch, in us
    FH_REQUIRE(h->mega_sync && h->mega[3].nstage > 0 && h->mega[3].nstage <= 64, "no persistent-kernel plan");
    FH_CHECK_HIP(hipStreamSynchronize(s));
    const int n = h->mega[3].nstage * 3;
    std::vector<unsigned long long> st(n);
    FH_CHECK_HIP(hipMemcpy(st.data(), h->mega_sync + 20 * 32, n * 8, hipMemcpyDeviceToHost));
    std::vector<float> us(n);
    for (int i = 0; i < n; ++i) us[i] = st[i] ? (float)((double)(st[i] - st[0]) * 0.01) : -1.f;  // 100 MHz
    FH_CHECK_HIP(hipMemcpy(out, us.data(), n * 4, hipMemcpyHostToDevice));
    return FLOAT_OK;
  }
  FH_REQUIRE(what == 1 && in != nullptr, "float_fmt_debug: unknown request %d (or null input)", what);
  const int n = ntok * 3 * D;
  hipLaunchKernelGGL((fmt_dbg_to16_kernel<T>), dim3((n + 255) / 256), dim3(256), 0, s, h->qkv16, in, n);
  launch_attn<T>(h, ntok, nullptr, s);
  hipLaunchKernelGGL((fmt_dbg_unpack_kernel<T>), dim3((ntok * D + 255) / 256), dim3(256), 0, s, out, h->att16, ntok, D);
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

extern "C" {

int float_fmt_create(const float_fmt_cfg_t* cfg, const float_tensor_t* tensors, int32_t n_tensors, float_fmt_t** out) {
  FH_REQUIRE(cfg && tensors && out, "null argument to float_fmt_create");
  FH_REQUIRE(cfg->dim_h % 256 == 0 && cfg->dim_h / 256 <= 8 && ((cfg->dim_h / 256) & (cfg->dim_h / 256 - 1)) == 0,
             "dim_h=%d unsupported", cfg->dim_h);
  FH_REQUIRE(cfg->dim_h / cfg->heads == 128, "head_dim must be 128 (dim_h=%d heads=%d)", cfg->dim_h, cfg->heads);
  FH_REQUIRE(cfg->dim_w % 128 == 0 && cfg->mlp_hidden % 128 == 0, "dim_w / mlp_hidden must be multiples of 128");
  FH_REQUIRE(cfg->n_prev + cfg->n_cur <= kMaxTok, "at most %d tokens per window (got %d)", kMaxTok, cfg->n_prev + cfg->n_cur);
  FH_REQUIRE(cfg->n_prev >= 0 && cfg->n_prev <= cfg->n_cur, "n_prev must be in [0, n_cur]");
  FH_REQUIRE(cfg->dtype == FLOAT_DT_BF16 || cfg->dtype == FLOAT_DT_FP16 || cfg->dtype == FLOAT_DT_FP32, "unknown dtype %d", cfg->dtype);
  FH_REQUIRE(cfg->max_batch >= 0 && cfg->max_batch <= 16, "max_batch must be in [0, 16] (got %d)", cfg->max_batch);
  float_fmt* h = new float_fmt();
  h->cfg = *cfg;
  h->D = cfg->dim_h;
  h->ntok = cfg->n_prev + cfg->n_cur;
  // rows of every activation buffer: the 4-way CFG batch plus the row tiles a row-blocked tiling reads past it (row blocks
  // are 3-6 tiles; their operand loads are not guarded, the rows are zero and their results are dropped)
  h->Bmax = cfg->max_batch > 0 ? cfg->max_batch : 1;
  h->Mpad = 16 * ((h->Bmax * 4 * h->ntok + 15) / 16 + 12);  // + the row tiles a 192-row block reads past the last row
  h->Kc = round_up(cfg->dim_w + cfg->dim_a + cfg->dim_e, 128);
  h->Kx = round_up(cfg->dim_w, 128);
  if (const char* wd = getenv("FLOAT_FMT_WIDE")) g_fmt_wide = atoi(wd) != 0;
  if (const char* v = getenv("FLOAT_FMT_FC2_SPLIT")) g_fmt_fc2_split = atoi(v);
  if (const char* v = getenv("FLOAT_FMT_TOUCH")) g_fmt_touch = atoi(v);
  if (const char* v = getenv("FLOAT_FMT_FULL_NW")) g_fmt_full_nw = atoi(v);
  if (const char* v = getenv("FLOAT_FMT_WIDE_VARIANT")) g_fmt_wide_variant = atoi(v);
  if (const char* v = getenv("FLOAT_FMT_PROJ_SPLIT")) g_fmt_proj_split = atoi(v);
  if (const char* v = getenv("FLOAT_FMT_RB")) g_fmt_rb = atoi(v) != 0;
  if (const char* v = getenv("FLOAT_FMT_RB_TOUCH")) g_fmt_rb_touch = atoi(v);
  h->attnproj = getenv("FLOAT_FMT_ATTNPROJ") ? atoi(getenv("FLOAT_FMT_ATTNPROJ")) : 0;
  h->mega_on = getenv("FLOAT_FMT_MEGA") ? atoi(getenv("FLOAT_FMT_MEGA")) : 0;
  {
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) h->n_cu = ncu;
    else (void)hipGetLastError();
  }
  if (const char* v = getenv("FLOAT_FMT_HOIST")) g_fmt_hoist = atoi(v) != 0;
  if (const char* v = getenv("FLOAT_FMT_ZGROUP")) g_fmt_zgroup = std::max(0, atoi(v));
  if (const char* v = getenv("FLOAT_FMT_BIG")) g_fmt_big = atoi(v) != 0;
  if (const char* pl = getenv("FLOAT_FMT_PLAN"))
    sscanf(pl, "%d,%d,%d,%d,%d,%d", &g_fmt_plan_override[0], &g_fmt_plan_override[1], &g_fmt_plan_override[2],
           &g_fmt_plan_override[3], &g_fmt_plan_override[4], &g_fmt_plan_override[5]);
  h->Ntot = cfg->depth * 6 * h->D + 2 * h->D;
  TensorTable tt(tensors, n_tensors);
  int rc = (cfg->dtype == FLOAT_DT_BF16) ? create_impl<BF16>(h, tt)
           : (cfg->dtype == FLOAT_DT_FP16) ? create_impl<FP16>(h, tt) : create_impl<FP32>(h, tt);
  const int D = h->D, Mp = h->Mpad;
  const size_t esz = cfg->dtype == FLOAT_DT_FP32 ? 2 : 1;  // u16 slots per operand element
  auto A = [&](auto** p, size_t n) {
    // operand buffers (typed u16*) hold 4-byte elements in the fp32 mode
    if (!rc) rc = h->pool.alloc(p, std::is_same<std::remove_reference_t<decltype(**p)>, u16>::value ? n * esz : n, true);
  };
  A(&h->pos, (size_t)kMaxTok * D);
  A(&h->freqs, 128);
  A(&h->cond16, (size_t)Mp * h->Kc);
  A(&h->sc16, (size_t)kScSteps * Mp * D + (size_t)12 * 16 * D);  // + the row tiles a 192-row block reads past the last batch
  A(&h->h16, (size_t)Mp * D);
  A(&h->hfin16, (size_t)h->Bmax * 64 * ((kMaxTok + 15) / 16) * D);  // token-blocked rows of the head GEMM, clips x 4 CFG rows x 16 tokens per block
  A(&h->qkv16, (size_t)Mp * 3 * D);
  A(&h->att16, (size_t)Mp * D);
  A(&h->hid16, (size_t)Mp * cfg->mlp_hidden);
  A(&h->xin16, (size_t)(h->Bmax * kMaxTok + 6 * 16) * h->Kx);  // + the row tiles a row-blocked tiling reads past the last token (zero)
  A(&h->tsin16, (size_t)kMaxSteps * 256);
  A(&h->th16, (size_t)kMaxSteps * D);
  A(&h->ccond, (size_t)Mp * D);
  // modulations of a batch of evaluations (run_mod_all): never zero-filled or read before written, up to 3.1 GB at the
  // default shape (64 x 240 x 51 200 fp32) of the 288 GB
  h->Mmod = 16 * ((h->Bmax * 4 * h->ntok + 15) / 16);
  if (!rc) rc = h->pool.alloc(&h->modall, (size_t)kScSteps * h->Mmod * h->Ntot, false);
  A(&h->kbuf, (size_t)4 * h->Bmax * kMaxTok * cfg->dim_w);
  A(&h->xres, (size_t)Mp * D);
  A(&h->slab, (size_t)8 * Mp * D);
  A(&h->sat, 1);
  A(&h->xcur, (size_t)h->Bmax * cfg->n_cur * cfg->dim_w);
  A(&h->temb, (size_t)kMaxSteps * D);
  A(&h->vout, (size_t)h->Bmax * kMaxTok * cfg->dim_w);
  A(&h->wa_c, (size_t)h->Bmax * cfg->n_cur * cfg->dim_a);
  A(&h->we_c, (size_t)h->Bmax * cfg->n_cur * cfg->dim_e);
  A(&h->x0_c, (size_t)h->Bmax * cfg->n_cur * cfg->dim_w);
  A(&h->prev_x, (size_t)h->Bmax * cfg->n_cur * cfg->dim_w);
  A(&h->prev_wa, (size_t)h->Bmax * cfg->n_cur * cfg->dim_a);
  A(&h->prev_we, (size_t)h->Bmax * cfg->n_cur * cfg->dim_e);
  if (!rc) {
    // pos_embed: from the checkpoint when present, else regenerated like the VA loader does
    // (nodes_vadv_loader.py:822-840): sin on even, cos on odd feature indices (FMT.py:29-37).
    std::vector<float> pos((size_t)h->ntok * D);
    const float_tensor_t* pe = tt.find("pos_embed");
    if (pe && TensorTable::numel(pe) == (int64_t)h->ntok * D) {
      memcpy(pos.data(), pe->data, pos.size() * sizeof(float));
    } else {
      for (int p = 0; p < h->ntok; ++p)
        for (int j = 0; j < D; ++j) {
          const float ang = (float)((double)p / pow(10000.0, 2.0 * (double)(j / 2) / (double)D));
          pos[(size_t)p * D + j] = (j & 1) ? cosf(ang) : sinf(ang);
        }
    }
    std::vector<float> fr(128);
    for (int k = 0; k < 128; ++k) fr[k] = expf(-logf(10000.0f) * (float)k / 128.0f);
    if (hipMemcpy(h->pos, pos.data(), pos.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->freqs, fr.data(), fr.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
      fh_set_error("hipMemcpy of FMT tables failed");
      rc = FLOAT_E_HIP;
    }
  }
  if (!rc && cfg->dtype != FLOAT_DT_FP32) {
    // stage table of the persistent evaluation kernel for the 3-way CFG shape, built now: the first evaluation may already
    // run under stream capture, where nothing can be allocated or copied
    const bool ok = cfg->dtype == FLOAT_DT_BF16 ? mega_shape_ok<BF16>(h, 1, 3) : mega_shape_ok<FP16>(h, 1, 3);
    if (ok) rc = cfg->dtype == FLOAT_DT_BF16 ? build_mega<BF16>(h, 3) : build_mega<FP16>(h, 3);
  }
  if (rc) {
    float_fmt_destroy(h);
    return rc;
  }
  *out = h;
  return FLOAT_OK;
}

void float_fmt_destroy(float_fmt_t* h) {
  if (!h) return;
  for (auto& g : h->graphs) (void)hipGraphExecDestroy(g.exec);
  if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
  if (h->mega_err_host) (void)hipHostFree(h->mega_err_host);
  h->pool.release();
  delete h;
}

int float_fmt_eval(float_fmt_t* h, float t, const float* x, const float* wa, const float* wr, const float* we,
                   int32_t we_len, const float* prev_x, const float* prev_wa, const float* prev_we, float a_cfg,
                   float r_cfg, float e_cfg, int32_t include_r_cfg, float* out, void* stream) {
  int rc = check_common(h, we, we_len, prev_we);
  if (rc) return rc;
  if ((rc = mega_poll(h))) return rc;
  FH_REQUIRE(x && wa && wr && we && prev_x && prev_wa && out, "null tensor argument to float_fmt_eval");
  hipStream_t s = (hipStream_t)stream;
  return h->cfg.dtype == FLOAT_DT_BF16
             ? eval_impl<BF16>(h, t, x, wa, wr, we, we_len, prev_x, prev_wa, prev_we, a_cfg, r_cfg, e_cfg, include_r_cfg, out, s)
         : h->cfg.dtype == FLOAT_DT_FP16
             ? eval_impl<FP16>(h, t, x, wa, wr, we, we_len, prev_x, prev_wa, prev_we, a_cfg, r_cfg, e_cfg, include_r_cfg, out, s)
             : eval_impl<FP32>(h, t, x, wa, wr, we, we_len, prev_x, prev_wa, prev_we, a_cfg, r_cfg, e_cfg, include_r_cfg, out, s);
}

int float_fmt_sample_chunk(float_fmt_t* h, const float* x0, const float* wa, const float* wr, const float* we,
                           int32_t we_len, const float* prev_x, const float* prev_wa, const float* prev_we, int32_t nfe,
                           float a_cfg, float r_cfg, float e_cfg, int32_t include_r_cfg, float* out, void* stream) {
  int rc = check_common(h, we, we_len, prev_we);
  if (rc) return rc;
  if ((rc = mega_poll(h))) return rc;
  FH_REQUIRE(x0 && wa && wr && we && prev_x && prev_wa && out, "null tensor argument to float_fmt_sample_chunk");
  FH_REQUIRE(nfe >= 1 && n_evals(h->method, nfe) < kMaxSteps, "nfe=%d: too many evaluations (max %d)", nfe, kMaxSteps);
  hipStream_t s = (hipStream_t)stream;
  std::vector<float> ts;
  linspace01(nfe, &ts);
  const float_fmt_cfg_t& c = h->cfg;
  // stage caller tensors into handle-owned buffers (the graph path needs stable addresses)
  if ((rc = dev_copy(h->x0_c, x0, (size_t)c.n_cur * c.dim_w, s))) return rc;
  if ((rc = dev_copy(h->wa_c, wa, (size_t)c.n_cur * c.dim_a, s))) return rc;
  if ((rc = dev_copy(h->prev_x, prev_x, (size_t)c.n_prev * c.dim_w, s))) return rc;
  if ((rc = dev_copy(h->prev_wa, prev_wa, (size_t)c.n_prev * c.dim_a, s))) return rc;
  if (we_len > 1) {
    if ((rc = dev_copy(h->we_c, we, (size_t)c.n_cur * c.dim_e, s))) return rc;
    if ((rc = dev_copy(h->prev_we, prev_we, (size_t)c.n_prev * c.dim_e, s))) return rc;
  }
  const float* we_p = we_len > 1 ? h->we_c : we;
  const float* pwe_p = we_len > 1 ? h->prev_we : nullptr;
  const TimeSpec tsp = time_spec(h->method, nfe);
  const int nev = std::max(1, n_evals(h->method, nfe));
#define FMT_CHUNK(TT)                                                                                                          \
  do {                                                                                                                           \
    if ((rc = prepare_time<TT>(h, tsp, nev, s))) return rc;                                                                      \
    rc = window_impl<TT>(h, h->x0_c, h->wa_c, wr, we_p, we_len, h->prev_x, h->prev_wa, pwe_p, nfe, ts, a_cfg, r_cfg, e_cfg,      \
                         include_r_cfg, s);                                                                                      \
  } while (0)
  if (c.dtype == FLOAT_DT_BF16) FMT_CHUNK(BF16);
  else if (c.dtype == FLOAT_DT_FP16) FMT_CHUNK(FP16);
  else FMT_CHUNK(FP32);
#undef FMT_CHUNK
  if (rc) return rc;
  return dev_copy(out, h->xcur, (size_t)c.n_cur * c.dim_w, s);
}

int float_fmt_debug(float_fmt_t* h, int32_t what, const float* in, float* out, void* stream) {
  FH_REQUIRE(h != nullptr && out != nullptr, "null argument to float_fmt_debug");
  return h->cfg.dtype == FLOAT_DT_BF16   ? debug_impl<BF16>(h, what, in, out, (hipStream_t)stream)
         : h->cfg.dtype == FLOAT_DT_FP16 ? debug_impl<FP16>(h, what, in, out, (hipStream_t)stream)
                                         : debug_impl<FP32>(h, what, in, out, (hipStream_t)stream);
}

int float_fmt_saturation(float_fmt_t* h, uint64_t* total, int32_t reset, void* stream) {
  FH_REQUIRE(h && total, "null argument to float_fmt_saturation");
  FH_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  unsigned long long v = 0;
  FH_CHECK_HIP(hipMemcpy(&v, h->sat, sizeof(v), hipMemcpyDeviceToHost));
  *total = v;
  if (reset) {  // on the caller's stream: ordered against the launches that add to the counter there (not the NULL stream's memset)
    FH_CHECK_HIP(hipMemsetAsync(h->sat, 0, sizeof(v), (hipStream_t)stream));
    FH_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  }
  if (int rc = mega_poll(h)) return rc;  // the persistent kernel's barrier watchdog (also polled by every other call)
  return FLOAT_OK;
}

int float_fmt_set_method(float_fmt_t* h, int32_t method) {
  FH_REQUIRE(h != nullptr, "null FMT handle");
  FH_REQUIRE(method >= FLOAT_ODE_EULER && method <= FLOAT_ODE_HEUN3, "unknown ODE method %d", method);
  h->method = method;
  return FLOAT_OK;
}

int float_fmt_sample_begin(float_fmt_t* h, const float* wr, const float* wa, int32_t T, const float* we, int32_t we_len,
                           const float* noise, int32_t nfe, float a_cfg, float r_cfg, float e_cfg, int32_t include_r_cfg,
                           float* r_d) {
  FH_REQUIRE(h != nullptr, "null FMT handle");
  FH_REQUIRE(wr && wa && we && noise && r_d, "null tensor argument to float_fmt_sample");
  FH_REQUIRE(T >= 1, "T must be >= 1 (got %d)", T);
  FH_REQUIRE(we_len == 1 || we_len == T,
             "Dynamic emotion latent `we` time dimension (%d) does not match audio latent `wa` time dimension (%d).",
             we_len, T);
  FH_REQUIRE(nfe >= 1 && n_evals(h->method, nfe) < kMaxSteps, "nfe=%d: too many evaluations (max %d)", nfe, kMaxSteps);
  auto& J = h->job;
  J.wr = wr;
  J.wa = wa;
  J.we = we;
  J.noise = noise;
  J.r_d = r_d;
  J.T = T;
  J.we_len = we_len;
  J.nfe = nfe;
  J.include_r = include_r_cfg;
  J.a = a_cfg;
  J.r = r_cfg;
  J.e = e_cfg;
  J.B = 1;
  J.next = 0;
  J.n_chunks = (T + h->cfg.n_cur - 1) / h->cfg.n_cur;
  J.first = 0;
  J.total = J.n_chunks;
  J.hist_x = J.hist_wa = J.hist_we = nullptr;
  linspace01(nfe, &J.ts);
  J.active = true;
  return FLOAT_OK;
}

int float_fmt_sample_begin_range(float_fmt_t* h, const float* wr, const float* wa, int32_t T, const float* we, int32_t we_len,
                                 const float* noise, int32_t nfe, float a_cfg, float r_cfg, float e_cfg, int32_t include_r_cfg,
                                 float* r_d, int32_t first_window, int32_t end_window, const float* hist_x, const float* hist_wa,
                                 const float* hist_we) {
  int rc = float_fmt_sample_begin(h, wr, wa, T, we, we_len, noise, nfe, a_cfg, r_cfg, e_cfg, include_r_cfg, r_d);
  if (rc) return rc;
  auto& J = h->job;
  J.active = false;
  FH_REQUIRE(first_window >= 0 && first_window < end_window && end_window <= J.total,
             "window range [%d, %d) outside the clip's %d windows", first_window, end_window, J.total);
  FH_REQUIRE(we_len == 1 || hist_we != nullptr || hist_x == nullptr,
             "a dynamic-emotion job that starts from a history needs hist_we too");
  J.first = J.next = first_window;
  J.n_chunks = end_window;
  J.hist_x = hist_x;
  J.hist_wa = hist_wa;
  J.hist_we = we_len > 1 ? hist_we : nullptr;
  J.active = true;
  return FLOAT_OK;
}

int float_fmt_sample_next(float_fmt_t* h, void* stream, int32_t* window_done, int32_t* windows_left) {
  FH_REQUIRE(h != nullptr && h->job.active, "float_fmt_sample_next without float_fmt_sample_begin");
  if (int prc = mega_poll(h)) return prc;
  auto& J = h->job;
  hipStream_t s = (hipStream_t)stream;
  const int k = J.next;
  int rc = h->cfg.dtype == FLOAT_DT_BF16   ? sample_window<BF16>(h, k, s)
           : h->cfg.dtype == FLOAT_DT_FP16 ? sample_window<FP16>(h, k, s)
                                           : sample_window<FP32>(h, k, s);
  if (rc) {
    J.active = false;
    return rc;
  }
  J.next = k + 1;
  if (window_done) *window_done = k;
  if (windows_left) *windows_left = J.n_chunks - J.next;
  if (J.next >= J.n_chunks) J.active = false;
  return FLOAT_OK;
}

int float_fmt_sample_batch(float_fmt_t* h, int32_t n_clips, const float* wr, const float* wa, int32_t T, const float* we,
                           int32_t we_len, const float* noise, int32_t nfe, float a_cfg, float r_cfg, float e_cfg,
                           int32_t include_r_cfg, float* r_d, void* stream) {
  FH_REQUIRE(h != nullptr, "null FMT handle");
  FH_REQUIRE(n_clips >= 1 && n_clips <= h->Bmax, "batch of %d clips, the handle was created with max_batch = %d", n_clips, h->Bmax);
  int rc = float_fmt_sample_begin(h, wr, wa, T, we, we_len, noise, nfe, a_cfg, r_cfg, e_cfg, include_r_cfg, r_d);
  if (rc) return rc;
  h->job.B = n_clips;
  int32_t left = 1;
  while (left > 0)
    if ((rc = float_fmt_sample_next(h, stream, nullptr, &left))) return rc;
  return FLOAT_OK;
}

int float_fmt_sample(float_fmt_t* h, const float* wr, const float* wa, int32_t T, const float* we, int32_t we_len,
                     const float* noise, int32_t nfe, float a_cfg, float r_cfg, float e_cfg, int32_t include_r_cfg,
                     float* r_d, void* stream) {
  int rc = float_fmt_sample_begin(h, wr, wa, T, we, we_len, noise, nfe, a_cfg, r_cfg, e_cfg, include_r_cfg, r_d);
  if (rc) return rc;
  int32_t left = 1;
  while (left > 0)
    if ((rc = float_fmt_sample_next(h, stream, nullptr, &left))) return rc;
  return FLOAT_OK;
}

}  // extern "C"
