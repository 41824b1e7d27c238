// The FMT's weight-streaming GEMM (fmt_kernels.hpp) as a service for the other operators of the
// library: packed 16-bit weights + bias from named fp32 host tensors, and a launch with the tiling the
// FMT uses for the shape.  Implemented in fmt_api.hip (the only translation unit that instantiates the
// kernel templates).
#pragma once
#include "fmt_pack.hpp"

struct FmtLin {
  u16* W = nullptr;    // packed [N/16][K/32][64][8] 16-bit (fmt_pack_off), K padded to a multiple of 128
  float* b = nullptr;  // [N]
  int N = 0, K = 0;
};

// Concatenates the named Linear layers (each (N_each, K) with bias) along N.  dtype = FLOAT_DT_*.
int fmt_pack_linear(DevicePool* pool, int dtype, const TensorTable& tt, const std::vector<std::string>& names, int N_each, int K,
                    FmtLin* out);
// Same from raw host arrays (w row-major (N, K), b (N) or nullptr -> zeros).
int fmt_pack_linear_raw(DevicePool* pool, int dtype, const float* w, const float* b, int N, int K, FmtLin* out);

// g.A packed activations (row tiles of 16, K = L.K), M rows; epilogue fields of g filled by the caller.
// epi: EPI_F32 / EPI_T16 / EPI_GELUERF_P16 / EPI_SILU_P16 / EPI_GELU_P16.
int fmt_gemm_run(int dtype, int epi, GemmArgs g, hipStream_t s);
GemmArgs fmt_gemm_args(const u16* A, const FmtLin& L, int M);
void fmt_gemm_prime(int dtype);  // raise the dynamic-LDS limit of every tiling once per process
