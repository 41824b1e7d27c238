// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the DECODER's access shapes (MI355X_MICROARCH.md, HBM:
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Every kernel moves a
// KNOWN byte count exactly once over a buffer of 2 GiB (8 x the Infinity Cache, nothing is re-read), so counter / bytes is the
// factor of that shape:
//   read16_stream   16 B per lane, a wave reads 1 KiB contiguous                           (the guide's reference: 1/2)
//   read16_halo<C>  the 3x3 conv's halo staging (dec_conv16_kernel): 4 lanes read the 64 B of one 32-channel chunk of a pixel,
//                   pixels C x 2 B apart (C = 32 / 64 / 128 channels NHWC); the other chunks of the pixel in later passes
//   read16_halo_rows<C>  the same with the halo's geometry: 18-pixel runs of an image row, rows of a 512-px image apart
//   write16_stream  16 B per lane, 1 KiB contiguous per wave                               (the guide's reference: exact)
//   write8_px<C>    the conv epilogue's dec_store4: 8 B per lane, the 4 lanes of a pixel write 32 B contiguous, pixels C x 2 B
//                   apart, the pixel's other 32-B pieces in later passes (different column tiles of the same workgroup)
//   write8_stream   8 B per lane, 512 B contiguous per wave
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes); tools/make_calibration_json.py turns the two
// counter files into profiles/rNN_counter_calibration.json.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/counter_calib.hip -o build_ab/counter_calib
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void keep(u4 v, unsigned* sink) {
  if ((v.x ^ v.y ^ v.z ^ v.w) == 0x9e3779b9u) *sink = 1u;  // never true for a zeroed buffer; keeps the loads
}

__global__ __launch_bounds__(256) void read16_stream(const u4* __restrict__ in, size_t n16, unsigned* sink) {
  u4 acc = u4{0u, 0u, 0u, 0u};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) acc ^= in[i];
  keep(acc, sink);
}

// pixel p = gid >> 2, 16-byte piece ch = gid & 3 of the pixel's 64-byte chunk c; all pixels of chunk 0 first, then chunk 1, ...
template <int C>
__global__ __launch_bounds__(256) void read16_halo(const unsigned char* __restrict__ in, size_t npix, unsigned* sink) {
  u4 acc = u4{0u, 0u, 0u, 0u};
  for (int c = 0; c < C / 32; ++c)
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < npix * 4; g += (size_t)gridDim.x * 256)
      acc ^= *reinterpret_cast<const u4*>(in + (g >> 2) * (size_t)(C * 2) + c * 64 + (g & 3) * 16);
  keep(acc, sink);
}

// the halo's geometry: a workgroup stages 18 x 18 pixels of a 512-px-wide image per tile (tiles 16 apart: the 2-pixel overlap
// is read twice, as in the conv), 4 lanes per pixel, chunk after chunk
template <int C>
__global__ __launch_bounds__(256) void read16_halo_rows(const unsigned char* __restrict__ in, int frames, unsigned* sink) {
  constexpr int W = 512, TPR = W / 16;
  u4 acc = u4{0u, 0u, 0u, 0u};
  const int ntile = frames * TPR * TPR;
  for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
    const int f = t / (TPR * TPR), r = t - f * TPR * TPR, ty = r / TPR, tx = r - ty * TPR;
    for (int c = 0; c < C / 32; ++c)
      for (int e = threadIdx.x; e < 18 * 18 * 4; e += 256) {
        const int p = e >> 2, hy = p / 18, hx = p - hy * 18;
        const int y = min(max(ty * 16 + hy - 1, 0), W - 1), x = min(max(tx * 16 + hx - 1, 0), W - 1);
        acc ^= *reinterpret_cast<const u4*>(in + ((size_t)(f * W + y) * W + x) * (C * 2) + c * 64 + (e & 3) * 16);
      }
  }
  keep(acc, sink);
}

__global__ __launch_bounds__(256) void write16_stream(u4* __restrict__ out, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = u4{1u, 2u, 3u, (unsigned)i};
}

__global__ __launch_bounds__(256) void write8_stream(u2* __restrict__ out, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) out[i] = u2{1u, (unsigned)i};
}

// pixel p = gid >> 2, 8-byte piece q = gid & 3 of the pixel's 32-byte run j (16 output channels); all pixels of run 0 first
template <int C>
__global__ __launch_bounds__(256) void write8_px(unsigned char* __restrict__ out, size_t npix) {
  for (int j = 0; j < C / 16; ++j)
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < npix * 4; g += (size_t)gridDim.x * 256)
      *reinterpret_cast<u2*>(out + (g >> 2) * (size_t)(C * 2) + j * 32 + (g & 3) * 8) = u2{1u, (unsigned)g};
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  unsigned char *a, *b;
  unsigned* sink;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&b, bytes));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 0, bytes));
  CK(hipMemset(b, 0, bytes));
  CK(hipDeviceSynchronize());
  const dim3 grid(2048), blk(256);
  // bytes moved per kernel: printed for the summary script (kernel name, bytes read, bytes written)
  hipLaunchKernelGGL(read16_stream, grid, blk, 0, nullptr, (const u4*)a, bytes / 16, sink);
  printf("CALIB read16_stream %zu 0\n", bytes);
  hipLaunchKernelGGL(read16_halo<32>, grid, blk, 0, nullptr, a, bytes / 64, sink);
  printf("CALIB read16_halo<32> %zu 0\n", bytes);
  hipLaunchKernelGGL(read16_halo<64>, grid, blk, 0, nullptr, a, bytes / 128, sink);
  printf("CALIB read16_halo<64> %zu 0\n", bytes);
  hipLaunchKernelGGL(read16_halo<128>, grid, blk, 0, nullptr, a, bytes / 256, sink);
  printf("CALIB read16_halo<128> %zu 0\n", bytes);
  {
    // 18 x 18 halos over 512-px images: 324 / 256 of the image bytes are requested (the overlap)
    const int f32 = (int)(bytes / ((size_t)512 * 512 * 64)), f64 = (int)(bytes / ((size_t)512 * 512 * 128));
    hipLaunchKernelGGL(read16_halo_rows<32>, grid, blk, 0, nullptr, a, f32, sink);
    printf("CALIB read16_halo_rows<32> %zu 0 requested %zu\n", (size_t)f32 * 512 * 512 * 64, (size_t)f32 * 32 * 32 * 324 * 64);
    hipLaunchKernelGGL(read16_halo_rows<64>, grid, blk, 0, nullptr, a, f64, sink);
    printf("CALIB read16_halo_rows<64> %zu 0 requested %zu\n", (size_t)f64 * 512 * 512 * 128, (size_t)f64 * 32 * 32 * 324 * 128);
  }
  hipLaunchKernelGGL(write16_stream, grid, blk, 0, nullptr, (u4*)b, bytes / 16);
  printf("CALIB write16_stream 0 %zu\n", bytes);
  hipLaunchKernelGGL(write8_stream, grid, blk, 0, nullptr, (u2*)b, bytes / 8);
  printf("CALIB write8_stream 0 %zu\n", bytes);
  hipLaunchKernelGGL(write8_px<32>, grid, blk, 0, nullptr, b, bytes / 64);
  printf("CALIB write8_px<32> 0 %zu\n", bytes);
  hipLaunchKernelGGL(write8_px<64>, grid, blk, 0, nullptr, b, bytes / 128);
  printf("CALIB write8_px<64> 0 %zu\n", bytes);
  hipLaunchKernelGGL(write8_px<128>, grid, blk, 0, nullptr, b, bytes / 256);
  printf("CALIB write8_px<128> 0 %zu\n", bytes);
  CK(hipDeviceSynchronize());
  return 0;
}
