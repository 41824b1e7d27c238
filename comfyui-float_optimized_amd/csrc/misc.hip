// Error string, ABI version and per-kernel-class event profiling for libfloat_hip.so.
#include <stdarg.h>

#include "common.hpp"

static thread_local char g_err[512] = "";
int g_fh_profiling = 0;

void fh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {
constexpr int kClasses = 4;
struct Slot {
  std::vector<hipEvent_t> pool;                          // recycled events
  std::vector<std::pair<hipEvent_t, hipEvent_t>> live;  // recorded pairs not yet read
  hipEvent_t open = nullptr;
};
Slot g_slots[kClasses];

hipEvent_t get_event(Slot& s) {
  if (!s.pool.empty()) {
    hipEvent_t e = s.pool.back();
    s.pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

// Kernel-exact timing: the pair is handed to hipExtLaunchKernelGGL, which stamps the events with the
// dispatch's own begin/end timestamps (what rocprofv3 --kernel-trace reports), not with the time the
// stream reached a separately recorded event (that adds ~3 us of event handling per launch).
bool fh_prof_pair(int which, hipEvent_t* start, hipEvent_t* stop) {
  if (!g_fh_profiling || which < 0 || which >= kClasses) return false;
  Slot& s = g_slots[which];
  *start = get_event(s);
  *stop = get_event(s);
  s.live.emplace_back(*start, *stop);
  return true;
}

void fh_prof_begin(int which, hipStream_t st) {
  if (!g_fh_profiling || which < 0 || which >= kClasses) return;
  Slot& s = g_slots[which];
  s.open = get_event(s);
  (void)hipEventRecord(s.open, st);
}

void fh_prof_end(int which, hipStream_t st) {
  if (!g_fh_profiling || which < 0 || which >= kClasses) return;
  Slot& s = g_slots[which];
  if (!s.open) return;
  hipEvent_t e = get_event(s);
  (void)hipEventRecord(e, st);
  s.live.emplace_back(s.open, e);
  s.open = nullptr;
}

__global__ void fh_copy_words_kernel(unsigned* __restrict__ dst, const unsigned* __restrict__ src, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

int fh_copy_d2d(void* dst, const void* src, size_t bytes, hipStream_t s) {
  FH_REQUIRE(bytes % 4 == 0, "fh_copy_d2d: %zu bytes is not a multiple of 4", bytes);
  if (!bytes) return FLOAT_OK;
  const size_t n = bytes / 4;
  hipLaunchKernelGGL(fh_copy_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (unsigned*)dst, (const unsigned*)src, n);
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

// ---- measured peaks of the device the caller is on (float_probe_peaks): what bench.py prints beside the spec-sheet peaks of
// its roofline objects ("print measured-peak values next to every roofline fraction").
typedef unsigned int probe_u4 __attribute__((ext_vector_type(4)));
typedef _Float16 probe_h8 __attribute__((ext_vector_type(8)));
typedef float probe_f4 __attribute__((ext_vector_type(4)));
typedef float probe_f16v __attribute__((ext_vector_type(16)));

// streaming read: 16 bytes per lane, 8 loads in flight per thread, grid-stride over a buffer far beyond the Infinity Cache
__global__ __launch_bounds__(256) void probe_read_kernel(const probe_u4* __restrict__ in, size_t n16, unsigned* __restrict__ sink) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  probe_u4 acc = probe_u4{0u, 0u, 0u, 0u};
  for (; i + 7 * stride < n16; i += 8 * stride) {
    probe_u4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(in + i + k * stride);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc ^= v[k];
  }
  for (; i < n16; i += stride) acc ^= __builtin_nontemporal_load(in + i);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) *sink = 1u;  // keeps the loads; never true for a zeroed buffer
}

__global__ __launch_bounds__(256) void probe_copy_kernel(const probe_u4* __restrict__ in, probe_u4* __restrict__ out, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    probe_u4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = __builtin_nontemporal_load(in + i + k * stride);
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(v[k], out + i + k * stride);
  }
  for (; i < n16; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}

// dense fp16 MFMA issue rate: every wave keeps 8 (16x16x32) or 4 (32x32x16) independent accumulators busy, operands in
// registers.  Inline asm: with the builtins hipcc moved the 16x16x32 accumulators between AGPRs and VGPRs inside the loop (32
// v_accvgpr moves per 8 MFMAs) and the loop measured half the rate of the 32x32x16 one.
template <int BIG>
__global__ __launch_bounds__(256) void probe_mfma_kernel(int iters, float* __restrict__ sink) {
  probe_u4 a, b;
  a.x = 0x1c001c00u + (threadIdx.x & 7u);  // small positive fp16 pairs
  a.y = a.x + 1u, a.z = a.x + 2u, a.w = a.x + 3u;
  b = a + probe_u4{5u, 6u, 7u, 8u};
  float total = 0.f;
  if (BIG) {
    probe_f16v acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last results have landed before they are read
#pragma unroll
    for (int j = 0; j < 4; ++j) total += acc[j][0] + acc[j][15];
  } else {
    probe_f4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = probe_f4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; ++j) total += acc[j][0] + acc[j][3];
  }
  if (total == 12345.678f) *sink = total;
}

extern "C" {

int float_probe_peaks(float* hbm_read_gbps, float* hbm_copy_gbps, float* mfma16_tflops, float* mfma32_tflops, int32_t* n_cu) {
  int dev = 0, ncu = 0;
  FH_CHECK_HIP(hipGetDevice(&dev));
  FH_CHECK_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  if (n_cu) *n_cu = ncu;
  const size_t bytes = (size_t)2 << 30;  // 2 GiB: 8 x the Infinity Cache
  const size_t n16 = bytes / 16;
  void *a = nullptr, *b = nullptr;
  float* sink = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  auto cleanup = [&] {
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (sink) (void)hipFree(sink);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
  };
#define PROBE_CHECK(x)            \
  do {                            \
    hipError_t e_ = (x);          \
    if (e_ != hipSuccess) {       \
      fh_set_error("float_probe_peaks: %s", hipGetErrorString(e_)); \
      cleanup();                  \
      return FLOAT_E_HIP;         \
    }                             \
  } while (0)
  PROBE_CHECK(hipMalloc(&a, bytes));
  PROBE_CHECK(hipMalloc(&b, bytes));
  PROBE_CHECK(hipMalloc(&sink, 64));
  PROBE_CHECK(hipMemset(a, 0, bytes));
  PROBE_CHECK(hipMemset(b, 0, bytes));
  PROBE_CHECK(hipEventCreate(&e0));
  PROBE_CHECK(hipEventCreate(&e1));
  auto best_ms = [&](auto launch) -> float {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {  // the first one warms up
      (void)hipEventRecord(e0, nullptr);
      launch();
      (void)hipEventRecord(e1, nullptr);
      (void)hipEventSynchronize(e1);
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms > 0.f) best = std::min(best, ms);
    }
    return best;
  };
  const unsigned grid = (unsigned)ncu * 8u;
  float ms = best_ms([&] { hipLaunchKernelGGL(probe_read_kernel, dim3(grid), dim3(256), 0, nullptr, (const probe_u4*)a, n16, (unsigned*)sink); });
  if (hbm_read_gbps) *hbm_read_gbps = (float)((double)bytes / 1e9 / ((double)ms * 1e-3));
  ms = best_ms([&] { hipLaunchKernelGGL(probe_copy_kernel, dim3(grid), dim3(256), 0, nullptr, (const probe_u4*)a, (probe_u4*)b, n16); });
  if (hbm_copy_gbps) *hbm_copy_gbps = (float)(2.0 * (double)bytes / 1e9 / ((double)ms * 1e-3));
  const int iters = 20000;
  const unsigned mgrid = (unsigned)ncu * 2u;  // 8 waves per CU, 2 per SIMD
  const double waves = (double)mgrid * 4.0;
  ms = best_ms([&] { hipLaunchKernelGGL((probe_mfma_kernel<0>), dim3(mgrid), dim3(256), 0, nullptr, iters, sink); });
  if (mfma16_tflops) *mfma16_tflops = (float)(waves * iters * 8.0 * (2.0 * 16 * 16 * 32) / 1e12 / ((double)ms * 1e-3));
  ms = best_ms([&] { hipLaunchKernelGGL((probe_mfma_kernel<1>), dim3(mgrid), dim3(256), 0, nullptr, iters, sink); });
  if (mfma32_tflops) *mfma32_tflops = (float)(waves * iters * 4.0 * (2.0 * 32 * 32 * 16) / 1e12 / ((double)ms * 1e-3));
  PROBE_CHECK(hipGetLastError());
#undef PROBE_CHECK
  cleanup();
  return FLOAT_OK;
}

int float_hip_abi_version(void) { return FLOAT_HIP_ABI_VERSION; }

const char* float_last_error(void) { return g_err; }

int float_set_profiling(int32_t on) {
  g_fh_profiling = on ? 1 : 0;
  if (on) {
    for (auto& s : g_slots) {
      for (auto& p : s.live) {
        s.pool.push_back(p.first);
        s.pool.push_back(p.second);
      }
      s.live.clear();
    }
  }
  return FLOAT_OK;
}

// A stream whose kernels may only run on CUs [cu_begin, cu_end) of the device (hipExtStreamCreateWithCUMask):
// lets the caller give the latency-bound FMT chain and the throughput-bound decoder disjoint CU sets so
// that the chain's tiny dependent kernels never queue behind decoder workgroups.
int float_stream_create_cu_range(int32_t cu_begin, int32_t cu_end, void** stream_out) {
  FH_REQUIRE(stream_out && cu_begin >= 0 && cu_end > cu_begin && cu_end <= 1024, "bad CU range [%d,%d)", cu_begin, cu_end);
  uint32_t mask[32];
  memset(mask, 0, sizeof(mask));
  for (int c = cu_begin; c < cu_end; ++c) mask[c >> 5] |= 1u << (c & 31);
  hipStream_t s = nullptr;
  FH_CHECK_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)((cu_end + 31) / 32), mask));
  *stream_out = (void*)s;
  return FLOAT_OK;
}

int float_stream_destroy(void* stream) {
  if (stream) FH_CHECK_HIP(hipStreamDestroy((hipStream_t)stream));
  return FLOAT_OK;
}

double float_profile_ms(int32_t which, int64_t* n_launches) {
  if (which < 0 || which >= kClasses) return -1.0;
  Slot& s = g_slots[which];
  double total = 0.0;
  int64_t n = 0;
  for (auto& p : s.live) {
    float ms = 0.f;
    (void)hipEventSynchronize(p.second);
    if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
      total += ms;
      ++n;
    }
    s.pool.push_back(p.first);
    s.pool.push_back(p.second);
  }
  s.live.clear();
  if (n_launches) *n_launches = n;
  return n ? total / (double)n : -1.0;
}

}  // extern "C"
