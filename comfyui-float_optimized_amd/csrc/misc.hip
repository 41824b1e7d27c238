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

extern "C" {

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
