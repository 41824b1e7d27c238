// C-ABI entry points of the appearance encoder (include/float_hip.h, section "encoder"):
// EncoderApp.forward + Encoder.fc + Direction (reference encoder.py:203-247, styledecoder.py:428-444,
// called from FLOAT.py:283-291 once per clip).
#include <math.h>

#include "dec_kernels.hpp"  // dec_conv16_kernel: the LDS-staged 3x3 conv of the decoder, used for the ResBlocks' conv1
#include "enc_kernels.hpp"

namespace {

struct EConv {
  void* W = nullptr;     // [k*k][Cout][Cin] T::elem, equalised-lr scale folded in
  float* bias = nullptr; // [Cout] FusedLeakyReLU bias or nullptr
  int cin = 0, cout = 0, k = 0;
};

struct ResBlk {
  EConv conv1, conv2, skip;
  int R = 0;  // input resolution
  EncFir fir2, firs;  // the Blur in front of conv2 / of the skip conv, flipped (enc_blur_taps)
};

struct FcLayer {
  float* W = nullptr;
  float* b = nullptr;
  int n = 0, k = 0;
};

}  // namespace

struct float_enc {
  float_enc_cfg_t cfg;
  DevicePool pool;
  int C0 = 0;
  float *w0 = nullptr, *b0 = nullptr;  // convs.0: [C0][3] scaled, [C0]
  std::vector<ResBlk> blocks;
  EConv last;                          // EqualConv2d(C, dim, 4, padding 0, no bias)
  std::vector<FcLayer> fc;
  float* Q = nullptr;                  // [dim][dim_motion] of QR(direction.weight + 1e-8), or nullptr
  // activations: res[i] (NHWC 16-bit) for i = 0..n_blocks (res[0] = convs.0 output), scratch for
  // conv1 output, the two blurred images and the skip branch
  std::vector<void*> res;  // T::elem
  std::vector<int> resR, resC;
  void *t1 = nullptr, *tb = nullptr, *tsk = nullptr;
  float *s_r = nullptr, *fcA = nullptr, *fcB = nullptr;
  unsigned long long* sat = nullptr;  // range counter of every 16-bit activation store (float_enc_saturation): conv1 of the ResBlocks runs the decoder's kernel (overflow -> inf, counted), the other layers clamp at 65504 and count
};

namespace {

template <class T>
int pack_conv(float_enc* h, const TensorTable& tt, const std::string& wname, const std::string& bname, EConv* out) {
  const float_tensor_t* w = tt.find(wname);
  if (!w) {
    fh_set_error("missing checkpoint tensor '%s'", wname.c_str());
    return FLOAT_E_MISSING;
  }
  FH_REQUIRE(w->ndim == 4 && w->shape[2] == w->shape[3], "tensor '%s' is not (Cout,Cin,k,k)", wname.c_str());
  const int co = (int)w->shape[0], ci = (int)w->shape[1], k = (int)w->shape[2];
  FH_REQUIRE(ci % 32 == 0 && co % 32 == 0, "'%s': channel counts must be multiples of 32 (got %d -> %d)", wname.c_str(), ci, co);
  const float scale = 1.0f / sqrtf((float)(ci * k * k));  // EqualConv2d.scale (encoder.py:93)
  typedef typename T::elem E;
  std::vector<E> hw((size_t)k * k * co * ci);
  for (int o = 0; o < co; ++o)
    for (int i = 0; i < ci; ++i)
      for (int t = 0; t < k * k; ++t)
        hw[((size_t)t * co + o) * ci + i] = T::host_from_float(w->data[((size_t)o * ci + i) * k * k + t] * scale);
  int rc;
  E* dW = nullptr;
  if ((rc = h->pool.alloc(&dW, hw.size(), false))) return rc;
  out->W = dW;
  FH_CHECK_HIP(hipMemcpy(dW, hw.data(), hw.size() * sizeof(E), hipMemcpyHostToDevice));
  out->cin = ci;
  out->cout = co;
  out->k = k;
  if (!bname.empty()) {
    const float_tensor_t* b = tt.find(bname);
    if (!b || TensorTable::numel(b) != co) {
      fh_set_error("missing or mis-shaped checkpoint tensor '%s'", bname.c_str());
      return FLOAT_E_MISSING;
    }
    if ((rc = h->pool.alloc(&out->bias, (size_t)co, false))) return rc;
    FH_CHECK_HIP(hipMemcpy(out->bias, b->data, (size_t)co * sizeof(float), hipMemcpyHostToDevice));
  }
  return FLOAT_OK;
}

// Q of the Householder QR of A (m x n, m >= n), LAPACK geqrf/orgqr conventions (what torch.linalg.qr
// runs on the CPU): beta_j = -sign(alpha_j) * ||x_j||, H_j = I - tau v v^T, Q = H_0 ... H_{n-1} [:, :n].
void householder_q(const std::vector<double>& A_in, int m, int n, std::vector<double>* Q) {
  std::vector<double> A(A_in);  // row-major m x n
  std::vector<double> tau(n, 0.0);
  for (int j = 0; j < n; ++j) {
    double xnorm = 0.0;
    for (int i = j + 1; i < m; ++i) xnorm += A[(size_t)i * n + j] * A[(size_t)i * n + j];
    xnorm = sqrt(xnorm);
    const double alpha = A[(size_t)j * n + j];
    if (xnorm == 0.0) {
      tau[j] = 0.0;
      continue;
    }
    const double nrm = sqrt(alpha * alpha + xnorm * xnorm);
    const double beta = alpha >= 0.0 ? -nrm : nrm;
    tau[j] = (beta - alpha) / beta;
    const double inv = 1.0 / (alpha - beta);
    for (int i = j + 1; i < m; ++i) A[(size_t)i * n + j] *= inv;  // v (v_j = 1 implicit)
    A[(size_t)j * n + j] = beta;
    for (int c = j + 1; c < n; ++c) {  // apply H_j to the trailing columns
      double s = A[(size_t)j * n + c];
      for (int i = j + 1; i < m; ++i) s += A[(size_t)i * n + j] * A[(size_t)i * n + c];
      s *= tau[j];
      A[(size_t)j * n + c] -= s;
      for (int i = j + 1; i < m; ++i) A[(size_t)i * n + c] -= s * A[(size_t)i * n + j];
    }
  }
  Q->assign((size_t)m * n, 0.0);
  for (int j = 0; j < n; ++j) (*Q)[(size_t)j * n + j] = 1.0;
  for (int j = n - 1; j >= 0; --j) {  // Q = H_0 (H_1 (... H_{n-1} I))
    for (int c = 0; c < n; ++c) {
      double s = (*Q)[(size_t)j * n + c];
      for (int i = j + 1; i < m; ++i) s += A[(size_t)i * n + j] * (*Q)[(size_t)i * n + c];
      s *= tau[j];
      (*Q)[(size_t)j * n + c] -= s;
      for (int i = j + 1; i < m; ++i) (*Q)[(size_t)i * n + c] -= s * A[(size_t)i * n + j];
    }
  }
}

// The Blur of a down-sampling ConvLayer is a registered 4 x 4 buffer (encoder.py:59-75: make_kernel([1,3,3,1]) = outer / 64) that
// the reference's strict load takes from the checkpoint; upfirdn2d correlates with it FLIPPED (encoder.py:28-29).  Any 4 x 4 values
// are applied as they are; no buffer in the state: the default.  Another size is refused (the layer's padding belongs to 4 taps).
int enc_blur_taps(const TensorTable& tt, const std::string& key, EncFir* f) {
  static const float k1[4] = {0.125f, 0.375f, 0.375f, 0.125f};
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b) f->k[a * 4 + b] = k1[a] * k1[b];
  const float_tensor_t* kb = tt.find(key);
  if (!kb) return FLOAT_OK;
  if (TensorTable::numel(kb) != 16 || kb->ndim != 2 || kb->shape[0] != 4) {
    fh_set_error("'%s' is not a 4 x 4 kernel; encoder blur kernels of other sizes are not implemented", key.c_str());
    return FLOAT_E_INVALID;
  }
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b) f->k[a * 4 + b] = kb->data[(3 - a) * 4 + (3 - b)];
  return FLOAT_OK;
}

template <class T>
int create_impl(float_enc* h, const TensorTable& tt) {
  const float_enc_cfg_t& c = h->cfg;
  int rc;
  const std::string p = "net_app.convs.";
  {  // convs.0: (C0, 3, 1, 1) + FusedLeakyReLU bias (1, C0, 1, 1)
    const float_tensor_t* w = tt.find(p + "0.0.weight");
    const float_tensor_t* b = tt.find(p + "0.1.bias");
    if (!w || !b) {
      fh_set_error("missing checkpoint tensor '%s0.0.weight' / '%s0.1.bias'", p.c_str(), p.c_str());
      return FLOAT_E_MISSING;
    }
    FH_REQUIRE(w->ndim == 4 && w->shape[1] == 3 && w->shape[2] == 1 && w->shape[0] % 32 == 0, "convs.0.0.weight must be (C,3,1,1)");
    h->C0 = (int)w->shape[0];
    std::vector<float> hw((size_t)h->C0 * 3);
    const float sc = 1.0f / sqrtf(3.0f);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = w->data[i] * sc;
    if ((rc = h->pool.alloc(&h->w0, hw.size(), false))) return rc;
    if ((rc = h->pool.alloc(&h->b0, (size_t)h->C0, false))) return rc;
    FH_CHECK_HIP(hipMemcpy(h->w0, hw.data(), hw.size() * sizeof(float), hipMemcpyHostToDevice));
    FH_CHECK_HIP(hipMemcpy(h->b0, b->data, (size_t)h->C0 * sizeof(float), hipMemcpyHostToDevice));
  }
  int R = c.size, C = h->C0, i = 1;
  size_t max_t1 = 0, max_tb = 0, max_sk = 0;
  h->resR.push_back(R);
  h->resC.push_back(C);
  while (R > 4) {
    const std::string q = p + std::to_string(i) + ".";
    ResBlk B;
    B.R = R;
    if ((rc = pack_conv<T>(h, tt, q + "conv1.0.weight", q + "conv1.1.bias", &B.conv1))) return rc;
    if ((rc = pack_conv<T>(h, tt, q + "conv2.1.weight", q + "conv2.2.bias", &B.conv2))) return rc;
    if ((rc = pack_conv<T>(h, tt, q + "skip.1.weight", "", &B.skip))) return rc;
    if ((rc = enc_blur_taps(tt, q + "conv2.0.kernel", &B.fir2)) || (rc = enc_blur_taps(tt, q + "skip.0.kernel", &B.firs))) return rc;
    FH_REQUIRE(B.conv1.cin == C && B.conv1.cout == C && B.conv1.k == 3 && B.conv2.cin == C && B.conv2.k == 3 &&
                   B.skip.cin == C && B.skip.k == 1 && B.skip.cout == B.conv2.cout,
               "ResBlock %d has unexpected shapes", i);
    max_t1 = std::max(max_t1, (size_t)R * R * C);
    max_tb = std::max(max_tb, (size_t)(R + 1) * (R + 1) * C);
    max_sk = std::max(max_sk, (size_t)(R / 2) * (R / 2) * B.conv2.cout);
    C = B.conv2.cout;
    R /= 2;
    h->blocks.push_back(B);
    h->resR.push_back(R);
    h->resC.push_back(C);
    ++i;
  }
  if ((rc = pack_conv<T>(h, tt, p + std::to_string(i) + ".weight", "", &h->last))) return rc;
  FH_REQUIRE(h->last.k == 4 && h->last.cin == C && h->last.cout == c.dim, "final EqualConv2d must be (%d,%d,4,4)", c.dim, C);
  for (int j = 0;; ++j) {
    const float_tensor_t* w = tt.find("fc." + std::to_string(j) + ".weight");
    const float_tensor_t* b = tt.find("fc." + std::to_string(j) + ".bias");
    if (!w) break;
    FH_REQUIRE(b && w->ndim == 2 && TensorTable::numel(b) == w->shape[0], "fc.%d has unexpected shapes", j);
    FcLayer L;
    L.n = (int)w->shape[0];
    L.k = (int)w->shape[1];
    if ((rc = h->pool.alloc(&L.W, (size_t)L.n * L.k, false))) return rc;
    if ((rc = h->pool.alloc(&L.b, (size_t)L.n, false))) return rc;
    FH_CHECK_HIP(hipMemcpy(L.W, w->data, (size_t)L.n * L.k * sizeof(float), hipMemcpyHostToDevice));
    FH_CHECK_HIP(hipMemcpy(L.b, b->data, (size_t)L.n * sizeof(float), hipMemcpyHostToDevice));
    h->fc.push_back(L);
  }
  FH_REQUIRE(!h->fc.empty() && h->fc.front().k == c.dim && h->fc.back().n == c.dim_motion,
             "Encoder.fc must map %d -> %d", c.dim, c.dim_motion);
  if (const float_tensor_t* dw = tt.find("direction.weight")) {  // Direction (styledecoder.py:431-436)
    FH_REQUIRE(dw->ndim == 2 && dw->shape[0] == c.dim && dw->shape[1] == c.dim_motion, "direction.weight must be (%d,%d)", c.dim,
               c.dim_motion);
    std::vector<float> Qf;
    fh_direction_q(dw->data, c.dim, c.dim_motion, &Qf);
    if ((rc = h->pool.alloc(&h->Q, Qf.size(), false))) return rc;
    FH_CHECK_HIP(hipMemcpy(h->Q, Qf.data(), Qf.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  typedef typename T::elem E;
  auto alloc_e = [&](void** p, size_t n) {
    E* b = nullptr;
    const int r = h->pool.alloc(&b, n, true);
    *p = b;
    return r;
  };
  for (size_t l = 0; l < h->resR.size(); ++l) {
    void* b = nullptr;
    if ((rc = alloc_e(&b, (size_t)h->resR[l] * h->resR[l] * h->resC[l]))) return rc;
    h->res.push_back(b);
  }
  if ((rc = alloc_e(&h->t1, max_t1))) return rc;
  if ((rc = alloc_e(&h->tb, max_tb))) return rc;
  if ((rc = alloc_e(&h->tsk, max_sk))) return rc;
  if ((rc = h->pool.alloc(&h->sat, 1, true))) return rc;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_conv16_kernel<T, 2, 3, 3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            32 * 1024 * T::EB);
  (void)hipGetLastError();
  if ((rc = h->pool.alloc(&h->s_r, (size_t)c.dim, true))) return rc;
  if ((rc = h->pool.alloc(&h->fcA, (size_t)std::max(c.dim, 64), true))) return rc;
  if ((rc = h->pool.alloc(&h->fcB, (size_t)std::max(c.dim, 64), true))) return rc;
  return FLOAT_OK;
}

template <class T>
int launch_conv(const EConv& L, const void* X, int Hi, int Wi, int stride, int pad, void* Y, float* Yf32, const void* skip,
                hipStream_t st, unsigned long long* sat) {
  EncConvArgs g;
  memset(&g, 0, sizeof(g));
  g.sat = sat;
  g.X = X;
  g.W = L.W;
  g.Y = Y;
  g.Yf32 = Yf32;
  g.bias = L.bias;
  g.skip = skip;
  g.Hi = Hi;
  g.Wi = Wi;
  g.Cin = L.cin;
  g.Cout = L.cout;
  g.k = L.k;
  g.stride = stride;
  g.pad = pad;
  g.Ho = (Hi + 2 * pad - L.k) / stride + 1;
  g.Wo = (Wi + 2 * pad - L.k) / stride + 1;
  g.act = L.bias ? 1 : 0;
  const int mb = (g.Ho * g.Wo + 63) / 64;
  if (L.cout % 64 == 0) hipLaunchKernelGGL((enc_conv_kernel<T, 4>), dim3(mb, L.cout / 64), dim3(256), 0, st, g);
  else hipLaunchKernelGGL((enc_conv_kernel<T, 2>), dim3(mb, L.cout / 32), dim3(256), 0, st, g);
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

// conv1 of a ResBlock - EqualConv2d(C, C, 3, padding 1) + FusedLeakyReLU on R x R (encoder.py:184-186), 34 of the encoder's
// 39 GFLOP - on the decoder's 16 x 16-tile kernel (halo tile and weight slab staged in swizzled LDS, 32 output channels per
// workgroup) from 16 px up: the direct-from-global kernel below ran the encoder at 39 TFLOP/s.  Same weight layout
// ([tap][Cout][Cin], taps row-major), no demodulation, no next-layer style.
template <class T>
int launch_conv3x3_tiles(float_enc* h, const EConv& L, const void* X, int R, void* Y, hipStream_t st) {
  static const int kDy[9] = {-1, -1, -1, 0, 0, 0, 1, 1, 1}, kDx[9] = {-1, 0, 1, -1, 0, 1, -1, 0, 1};
  constexpr size_t RB = 32 * T::EB;
  ConvArgs g;
  memset(&g, 0, sizeof(g));
  g.X = X;
  g.Wt = L.W;
  g.Y = Y;
  g.bias = L.bias;
  g.act = 1;
  g.sat = h->sat;
  g.F = 1;
  g.Hi = g.Wi = g.Ho = g.Wo = g.OH = g.OW = R;
  g.Cin = L.cin;
  g.Cout = L.cout;
  g.sy = g.sx = 1;
  g.ntaps = 9;
  for (int t = 0; t < 9; ++t) {
    g.dy[t] = (signed char)kDy[t];
    g.dx[t] = (signed char)kDx[t];
  }
  g.dymin = g.dxmin = -1;
  g.tiles_x = g.tiles_y = (R + 15) / 16;
  g.tpw = 1;
  const unsigned ngroups = (unsigned)(g.tiles_x * g.tiles_y), ncb = (unsigned)(L.cout / 32);
  dim3 grid(ngroups, ncb);
  if (ncb > 1) {  // the channel blocks of a tile side by side on one XCD (dec_group_cb)
    g.ngroups = ngroups;
    g.ncb = ncb;
    grid = dim3(ngroups * ncb, 1);
  }
  const size_t smem = (size_t)18 * 18 * RB + (size_t)9 * 32 * RB + 3 * 32 * sizeof(float);  // halo, weights, epilogue operands
  hipLaunchKernelGGL((dec_conv16_kernel<T, 2, 3, 3>), grid, dim3(256), smem, st, g);
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

template <class T>
int forward_impl(float_enc* h, const float* img, float* s_r, float* lam, float* r_s, float* const* feats, int n_feats,
                 hipStream_t st) {
  const float_enc_cfg_t& c = h->cfg;
  const int nb = (int)h->blocks.size();
  // feats[i] is the map of resolution 8 << i = res[nb - 1 - i]  (res[::-1][2:], encoder.py:231)
  auto feat_out = [&](int res_idx) -> float* {
    const int i = nb - 1 - res_idx;
    return (feats && i >= 0 && i < n_feats) ? feats[i] : nullptr;
  };
  int rc;
  typedef typename T::elem E;
  {
    const int HW = c.size * c.size;
    const size_t tot = (size_t)HW * (h->C0 / 8);
    hipLaunchKernelGGL((enc_first_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, img, h->w0, h->b0,
                       reinterpret_cast<E*>(h->res[0]), feat_out(0), HW, h->C0, h->sat);
  }
  for (int b = 0; b < nb; ++b) {
    const ResBlk& B = h->blocks[b];
    const int R = B.R, C = B.conv1.cin;
    const E* x = reinterpret_cast<const E*>(h->res[b]);
    // skip: Blur pad (1,1) -> 1x1 stride 2, no bias / activation (encoder.py:191)
    {
      const size_t tot = (size_t)(R - 1) * (R - 1) * (C / 8);
      hipLaunchKernelGGL((enc_blur_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, x, reinterpret_cast<E*>(h->tb), R, C, 1, B.firs, h->sat);
    }
    if ((rc = launch_conv<T>(B.skip, h->tb, R - 1, R - 1, 2, 0, h->tsk, nullptr, nullptr, st, h->sat))) return rc;
    // conv1 3x3 + act; conv2: Blur pad (2,2) -> 3x3 stride 2 + act; (out + skip) / sqrt(2)
    static const bool tiles_on = !getenv("FLOAT_ENC_NO_TILES");
    if (tiles_on && R >= 16 && B.conv1.cin % 32 == 0 && B.conv1.cout % 32 == 0) rc = launch_conv3x3_tiles<T>(h, B.conv1, x, R, h->t1, st);
    else rc = launch_conv<T>(B.conv1, x, R, R, 1, 1, h->t1, nullptr, nullptr, st, h->sat);
    if (rc) return rc;
    {
      const size_t tot = (size_t)(R + 1) * (R + 1) * (C / 8);
      hipLaunchKernelGGL((enc_blur_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const E*>(h->t1),
                         reinterpret_cast<E*>(h->tb), R, C, 2, B.fir2, h->sat);
    }
    if ((rc = launch_conv<T>(B.conv2, h->tb, R + 1, R + 1, 2, 0, h->res[b + 1], feat_out(b + 1), h->tsk, st, h->sat))) return rc;
  }
  // EqualConv2d(C, dim, 4): 4x4 -> 1x1 = s_r (encoder.py:219, 231)
  if ((rc = launch_conv<T>(h->last, h->res[nb], 4, 4, 1, 0, nullptr, h->s_r, nullptr, st, h->sat))) return rc;
  if (s_r && (rc = fh_copy_d2d(s_r, h->s_r, (size_t)c.dim * sizeof(float), st))) return rc;
  // Encoder.fc: EqualLinear chain without activation (encoder.py:242-247, 101-143)
  const float* cur = h->s_r;
  float* pp[2] = {h->fcA, h->fcB};
  for (size_t j = 0; j < h->fc.size(); ++j) {
    const FcLayer& L = h->fc[j];
    float* dst = pp[j & 1];
    hipLaunchKernelGGL(enc_linear_kernel, dim3((L.n + 3) / 4), dim3(256), 0, st, cur, L.W, L.b, 1.0f / sqrtf((float)L.k), dst, L.n, L.k);
    cur = dst;
  }
  if (lam && (rc = fh_copy_d2d(lam, cur, (size_t)c.dim_motion * sizeof(float), st))) return rc;
  if (r_s) {
    FH_REQUIRE(h->Q != nullptr, "r_s requested but the encoder was created without 'direction.weight'");
    hipLaunchKernelGGL(enc_linear_kernel, dim3((c.dim + 3) / 4), dim3(256), 0, st, cur, h->Q, (const float*)nullptr, 1.0f, r_s, c.dim,
                       c.dim_motion);
  }
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

}  // namespace

void fh_direction_q(const float* w, int dim, int dim_motion, std::vector<float>* Q) {
  std::vector<double> A((size_t)dim * dim_motion), Qd;
  for (size_t k = 0; k < A.size(); ++k) A[k] = (double)(w[k] + 1e-8f);  // weight + 1e-8 in fp32, as the reference adds it
  householder_q(A, dim, dim_motion, &Qd);
  Q->assign(Qd.begin(), Qd.end());
}

int fh_linear_f32(const float* x, const float* W, const float* b, float alpha, float* y, int N, int K, hipStream_t s) {
  hipLaunchKernelGGL(enc_linear_kernel, dim3((N + 3) / 4), dim3(256), 0, s, x, W, b, alpha, y, N, K);
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

extern "C" {

int float_enc_create(const float_enc_cfg_t* cfg, const float_tensor_t* tensors, int32_t n_tensors, float_enc_t** out) {
  FH_REQUIRE(cfg && tensors && out, "null argument to float_enc_create");
  FH_REQUIRE(cfg->size >= 64 && cfg->size <= 1024 && (cfg->size & (cfg->size - 1)) == 0,
             "encoder size must be a power of two in [64, 1024] (got %d)", cfg->size);
  FH_REQUIRE(cfg->dim > 0 && cfg->dim % 32 == 0 && cfg->dim_motion > 0 && cfg->dim_motion <= cfg->dim, "bad dim / dim_motion");
  FH_REQUIRE(cfg->dtype == FLOAT_DT_BF16 || cfg->dtype == FLOAT_DT_FP16 || cfg->dtype == FLOAT_DT_FP32, "unknown dtype %d", cfg->dtype);
  float_enc* h = new float_enc();
  h->cfg = *cfg;
  TensorTable tt(tensors, n_tensors);
  int rc = (cfg->dtype == FLOAT_DT_BF16) ? create_impl<BF16>(h, tt)
           : (cfg->dtype == FLOAT_DT_FP32) ? create_impl<FP32>(h, tt) : create_impl<FP16>(h, tt);
  if (rc) {
    float_enc_destroy(h);
    return rc;
  }
  *out = h;
  return FLOAT_OK;
}

void float_enc_destroy(float_enc_t* h) {
  if (!h) return;
  h->pool.release();
  delete h;
}

int float_enc_forward(float_enc_t* h, const float* img, float* s_r, float* lam, float* r_s, float* const* feats, int32_t n_feats,
                      void* stream) {
  FH_REQUIRE(h && img, "null argument to float_enc_forward");
  FH_REQUIRE(n_feats == 0 || feats != nullptr, "feats is null but n_feats = %d", n_feats);
  FH_REQUIRE(n_feats <= (int)h->blocks.size(), "at most %d feature maps (8..%d), got %d", (int)h->blocks.size(), h->cfg.size, n_feats);
  hipStream_t st = (hipStream_t)stream;
  if (h->cfg.dtype == FLOAT_DT_FP32) return forward_impl<FP32>(h, img, s_r, lam, r_s, feats, n_feats, st);
  return h->cfg.dtype == FLOAT_DT_BF16 ? forward_impl<BF16>(h, img, s_r, lam, r_s, feats, n_feats, st)
                                       : forward_impl<FP16>(h, img, s_r, lam, r_s, feats, n_feats, st);
}

int float_enc_saturation(float_enc_t* h, uint64_t* total, int32_t reset, void* stream) {
  FH_REQUIRE(h && total, "null argument to float_enc_saturation");
  FH_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  unsigned long long v = 0;
  FH_CHECK_HIP(hipMemcpy(&v, h->sat, sizeof(v), hipMemcpyDeviceToHost));
  *total = v;
  if (reset) {  // on the caller's stream: ordered against the launches that add to the counter there (not the NULL stream's memset)
    FH_CHECK_HIP(hipMemsetAsync(h->sat, 0, sizeof(v), (hipStream_t)stream));
    FH_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  }
  return FLOAT_OK;
}

int float_enc_feats16(float_enc_t* h, const void** feats16, int32_t* channels, int32_t max_feats, int32_t* n_out) {
  FH_REQUIRE(h && feats16 && n_out, "null argument to float_enc_feats16");
  const int nb = (int)h->blocks.size();
  const int n = std::min((int)max_feats, nb);
  for (int i = 0; i < n; ++i) {
    feats16[i] = h->res[nb - 1 - i];
    if (channels) channels[i] = h->resC[nb - 1 - i];
  }
  *n_out = n;
  return FLOAT_OK;
}

int float_enc_export_feats16(float_enc_t* h, void* const* dst, int32_t n_feats, void* stream) {
  FH_REQUIRE(h && dst, "null argument to float_enc_export_feats16");
  const int nb = (int)h->blocks.size();
  FH_REQUIRE(n_feats >= 1 && n_feats <= nb, "float_enc_export_feats16: %d maps asked, the encoder keeps %d", n_feats, nb);
  const size_t eb = h->cfg.dtype == FLOAT_DT_FP32 ? 4 : 2;
  for (int i = 0; i < n_feats; ++i) {
    FH_REQUIRE(dst[i] != nullptr, "float_enc_export_feats16: dst[%d] is null", i);
    const int l = nb - 1 - i;
    int rc = fh_copy_d2d(dst[i], h->res[l], (size_t)h->resR[l] * h->resR[l] * h->resC[l] * eb, (hipStream_t)stream);
    if (rc) return rc;
  }
  return FLOAT_OK;
}

}  // extern "C"
