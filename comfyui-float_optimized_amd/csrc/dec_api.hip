// C-ABI entry points of the decoder operator (include/float_hip.h): weight packing and the
// per-batch launch chain of Synthesis.forward (reference styledecoder.py:497-534).
#include <math.h>

#include "dec_kernels.hpp"

namespace {

struct Styled {  // one StyledConv (styledecoder.py:302-325)
  int cin = 0, cout = 0;
  bool up = false;
  u16* W = nullptr;       // plain: [9][Cout][Cin]; up: four parity classes, [4+2+2+1][Cout][Cin]
  float* WsqT = nullptr;  // [Cin][Cout] sum over taps of W^2 (fp32)
  float* abias = nullptr; // [Cout] FusedLeakyReLU bias
  int style_off = 0, demod_off = 0;
};

struct Level {  // ToFlow + ToRGB of one resolution
  int R = 0, C = 0;
  float *wflow = nullptr, *bflow = nullptr, *wrgb = nullptr, *b1 = nullptr, *b2 = nullptr;
  int style_off = 0;
  u16* feat = nullptr;  // [R][R][C]
};

int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

}  // namespace

struct float_dec {
  float_dec_cfg_t cfg;
  DevicePool pool;
  int n_levels = 0, Stot = 0, Dtot = 0;
  std::vector<Styled> convs;  // [0] = conv1, [1 + i] = convs.i
  std::vector<Level> levels;
  float* WmT = nullptr;   // [style_dim][Stot] every modulation EqualLinear, k-major
  float* bm = nullptr;    // [Stot]
  float* cin_hwc = nullptr;  // ConstantInput as [4][4][512]
  bool feats_set = false;
  // workspace (max_frames)
  float *styles = nullptr, *demod = nullptr;
  u16 *bufP = nullptr, *bufQ = nullptr, *bufZ = nullptr;
  float *flowA = nullptr, *flowB = nullptr, *rgbA = nullptr, *rgbB = nullptr;
};

namespace {

const float_tensor_t* need(const TensorTable& tt, const std::string& k, int64_t numel) {
  const float_tensor_t* t = tt.find(k);
  if (!t) {
    fh_set_error("missing checkpoint tensor '%s'", k.c_str());
    return nullptr;
  }
  if (numel >= 0 && TensorTable::numel(t) != numel) {
    fh_set_error("tensor '%s' has %lld elements, expected %lld", k.c_str(), (long long)TensorTable::numel(t), (long long)numel);
    return nullptr;
  }
  return t;
}

template <class T>
int upload16(float_dec* h, const std::vector<float>& src, u16** dst) {
  std::vector<u16> tmp(src.size());
  for (size_t i = 0; i < src.size(); ++i) tmp[i] = T::host_from_float(src[i]);
  int rc = h->pool.alloc(dst, tmp.size(), false);
  if (rc) return rc;
  FH_CHECK_HIP(hipMemcpy(*dst, tmp.data(), tmp.size() * sizeof(u16), hipMemcpyHostToDevice));
  return FLOAT_OK;
}

int upload32(float_dec* h, const std::vector<float>& src, float** dst) {
  int rc = h->pool.alloc(dst, src.size(), false);
  if (rc) return rc;
  FH_CHECK_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice));
  return FLOAT_OK;
}

// Parity classes of conv_transpose2d(stride 2, 3x3): output row u = 2m + pu receives kernel rows
// ky with 2y + ky = u: pu = 0 -> (ky=0, y=m), (ky=2, y=m-1); pu = 1 -> (ky=1, y=m).
struct ClassTaps {
  int n;
  int ky[4], kx[4], dy[4], dx[4];
};
ClassTaps class_taps(int pu, int pv) {
  ClassTaps c;
  c.n = 0;
  const int kys[2][2] = {{0, 2}, {1, -1}}, dys[2][2] = {{0, -1}, {0, 0}};
  for (int a = 0; a < 2; ++a) {
    if (kys[pu][a] < 0) continue;
    for (int b = 0; b < 2; ++b) {
      if (kys[pv][b] < 0) continue;
      c.ky[c.n] = kys[pu][a];
      c.dy[c.n] = dys[pu][a];
      c.kx[c.n] = kys[pv][b];
      c.dx[c.n] = dys[pv][b];
      ++c.n;
    }
  }
  return c;
}

template <class T>
int pack_styled(float_dec* h, const TensorTable& tt, const std::string& p, int cin, int cout, bool up, Styled* s,
                std::vector<float>* WmT_host, std::vector<float>* bm_host, int style_dim) {
  s->cin = cin;
  s->cout = cout;
  s->up = up;
  const float_tensor_t* w = need(tt, p + ".conv.weight", (int64_t)cout * cin * 9);
  const float_tensor_t* mw = need(tt, p + ".conv.modulation.weight", (int64_t)cin * style_dim);
  const float_tensor_t* mb = need(tt, p + ".conv.modulation.bias", cin);
  const float_tensor_t* ab = need(tt, p + ".activate.bias", cout);
  if (!w || !mw || !mb || !ab) return FLOAT_E_MISSING;
  const float scale = 1.0f / sqrtf((float)(cin * 9));  // styledecoder.py:223-224
  std::vector<float> packed((size_t)9 * cout * cin);
  auto W = [&](int o, int i, int ky, int kx) { return w->data[(((size_t)o * cin + i) * 3 + ky) * 3 + kx] * scale; };
  if (!up) {
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx)
        for (int o = 0; o < cout; ++o)
          for (int i = 0; i < cin; ++i) packed[(((size_t)(ky * 3 + kx)) * cout + o) * cin + i] = W(o, i, ky, kx);
  } else {
    size_t t0 = 0;
    for (int pu = 0; pu < 2; ++pu)
      for (int pv = 0; pv < 2; ++pv) {
        const ClassTaps c = class_taps(pu, pv);
        for (int t = 0; t < c.n; ++t, ++t0)
          for (int o = 0; o < cout; ++o)
            for (int i = 0; i < cin; ++i) packed[(t0 * cout + o) * cin + i] = W(o, i, c.ky[t], c.kx[t]);
      }
  }
  int rc;
  if ((rc = upload16<T>(h, packed, &s->W))) return rc;
  std::vector<float> wsq((size_t)cin * cout, 0.f);
  for (int o = 0; o < cout; ++o)
    for (int i = 0; i < cin; ++i) {
      double a = 0;
      for (int k = 0; k < 9; ++k) {
        const double v = w->data[((size_t)o * cin + i) * 9 + k];
        a += v * v;
      }
      wsq[(size_t)i * cout + o] = (float)a;
    }
  if ((rc = upload32(h, wsq, &s->WsqT))) return rc;
  if ((rc = upload32(h, std::vector<float>(ab->data, ab->data + cout), &s->abias))) return rc;
  s->style_off = (int)bm_host->size();
  for (int i = 0; i < cin; ++i) bm_host->push_back(mb->data[i]);
  WmT_host->insert(WmT_host->end(), mw->data, mw->data + (size_t)cin * style_dim);  // [cin][style_dim], transposed later
  return FLOAT_OK;
}

template <class T>
int create_impl(float_dec* h, const TensorTable& tt) {
  const int size = h->cfg.size, sdim = h->cfg.style_dim;
  static const int chan[] = {0, 0, 512, 512, 512, 512, 256, 128, 64, 32, 16};  // by log2(res), styledecoder.py:457-467
  const int log_size = ilog2(size);
  h->n_levels = log_size - 2;
  std::vector<float> wm_rows, bm_host;  // rows = modulation outputs, [Stot][sdim]
  int rc;
  h->convs.resize(1 + 2 * h->n_levels);
  if ((rc = pack_styled<T>(h, tt, "conv1", chan[2], chan[2], false, &h->convs[0], &wm_rows, &bm_host, sdim))) return rc;
  int cin = chan[2];
  for (int li = 0; li < h->n_levels; ++li) {
    const int cout = chan[li + 3];
    if ((rc = pack_styled<T>(h, tt, "convs." + std::to_string(2 * li), cin, cout, true, &h->convs[1 + 2 * li], &wm_rows,
                             &bm_host, sdim)))
      return rc;
    if ((rc = pack_styled<T>(h, tt, "convs." + std::to_string(2 * li + 1), cout, cout, false, &h->convs[2 + 2 * li],
                             &wm_rows, &bm_host, sdim)))
      return rc;
    cin = cout;
  }
  int doff = 0;
  for (auto& s : h->convs) {
    s.demod_off = doff;
    doff += s.cout;
  }
  h->Dtot = doff;
  h->levels.resize(h->n_levels);
  for (int li = 0; li < h->n_levels; ++li) {
    Level& L = h->levels[li];
    L.R = 8 << li;
    L.C = chan[li + 3];
    const std::string pf = "to_flows." + std::to_string(li), pr = "to_rgbs." + std::to_string(li);
    const float_tensor_t* fw = need(tt, pf + ".conv.weight", 3 * L.C);
    const float_tensor_t* fmw = need(tt, pf + ".conv.modulation.weight", (int64_t)L.C * sdim);
    const float_tensor_t* fmb = need(tt, pf + ".conv.modulation.bias", L.C);
    const float_tensor_t* fb = need(tt, pf + ".bias", 3);
    const float_tensor_t* rw = need(tt, pr + ".conv.0.weight", 3 * L.C);
    const float_tensor_t* rb1 = need(tt, pr + ".conv.1.bias", 3);
    const float_tensor_t* rb2 = need(tt, pr + ".bias", 3);
    if (!fw || !fmw || !fmb || !fb || !rw || !rb1 || !rb2) return FLOAT_E_MISSING;
    const float sc = 1.0f / sqrtf((float)L.C);  // 1x1: fan_in = C (styledecoder.py:134,223)
    std::vector<float> a(3 * L.C), b(3 * L.C);
    for (int i = 0; i < 3 * L.C; ++i) {
      a[i] = fw->data[i] * sc;
      b[i] = rw->data[i] * sc;
    }
    if ((rc = upload32(h, a, &L.wflow))) return rc;
    if ((rc = upload32(h, b, &L.wrgb))) return rc;
    if ((rc = upload32(h, std::vector<float>(fb->data, fb->data + 3), &L.bflow))) return rc;
    if ((rc = upload32(h, std::vector<float>(rb1->data, rb1->data + 3), &L.b1))) return rc;
    if ((rc = upload32(h, std::vector<float>(rb2->data, rb2->data + 3), &L.b2))) return rc;
    L.style_off = (int)bm_host.size();
    for (int i = 0; i < L.C; ++i) bm_host.push_back(fmb->data[i]);
    wm_rows.insert(wm_rows.end(), fmw->data, fmw->data + (size_t)L.C * sdim);
    if ((rc = h->pool.alloc(&L.feat, (size_t)L.R * L.R * L.C, true))) return rc;
  }
  h->Stot = (int)bm_host.size();
  std::vector<float> wmT((size_t)sdim * h->Stot);
  for (int j = 0; j < h->Stot; ++j)
    for (int k = 0; k < sdim; ++k) wmT[(size_t)k * h->Stot + j] = wm_rows[(size_t)j * sdim + k];
  if ((rc = upload32(h, wmT, &h->WmT))) return rc;
  if ((rc = upload32(h, bm_host, &h->bm))) return rc;
  // ConstantInput (1,512,4,4) -> HWC
  const float_tensor_t* ci = need(tt, "input.input", (int64_t)chan[2] * 16);
  if (!ci) return FLOAT_E_MISSING;
  std::vector<float> hwc((size_t)16 * chan[2]);
  for (int c = 0; c < chan[2]; ++c)
    for (int p = 0; p < 16; ++p) hwc[(size_t)p * chan[2] + c] = ci->data[(size_t)c * 16 + p];
  if ((rc = upload32(h, hwc, &h->cin_hwc))) return rc;
  // workspace
  const size_t F = (size_t)h->cfg.max_frames;
  size_t act = 0;
  for (int li = 0; li < h->n_levels; ++li) {
    const size_t R = 8u << li;
    act = std::max(act, (R + 1) * (R + 1) * (size_t)chan[li + 3]);
    act = std::max(act, (R / 2) * (R / 2) * (size_t)chan[li + 2]);
  }
  if ((rc = h->pool.alloc(&h->styles, F * h->Stot, true))) return rc;
  if ((rc = h->pool.alloc(&h->demod, F * h->Dtot, true))) return rc;
  if ((rc = h->pool.alloc(&h->bufP, F * act, true))) return rc;
  if ((rc = h->pool.alloc(&h->bufQ, F * act, true))) return rc;
  if ((rc = h->pool.alloc(&h->bufZ, F * act, true))) return rc;
  const size_t sk = F * (size_t)size * size * 3;
  if ((rc = h->pool.alloc(&h->flowA, sk, true))) return rc;
  if ((rc = h->pool.alloc(&h->flowB, sk, true))) return rc;
  if ((rc = h->pool.alloc(&h->rgbA, sk, true))) return rc;
  if ((rc = h->pool.alloc(&h->rgbB, sk, true))) return rc;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_conv_kernel<T, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_conv_kernel<T, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  return FLOAT_OK;
}

template <class T>
int launch_conv(const u16* X, int Hi, int Wi, const Styled& s, const u16* Wt, int ntaps, const int* dy, const int* dx,
                u16* Y, int Ho, int Wo, int OH, int OW, int sy, int sx, int py, int px, int F, const float* demod, int ldd,
                const float* bias, int act, const float* snext, int lds, hipStream_t st) {
  ConvArgs g;
  memset(&g, 0, sizeof(g));
  g.X = X;
  g.Wt = Wt;
  g.Y = Y;
  g.demod = demod;
  g.bias = bias;
  g.snext = snext;
  g.F = F;
  g.Hi = Hi;
  g.Wi = Wi;
  g.Cin = s.cin;
  g.Cout = s.cout;
  g.Ho = Ho;
  g.Wo = Wo;
  g.OH = OH;
  g.OW = OW;
  g.sy = sy;
  g.sx = sx;
  g.py = py;
  g.px = px;
  g.ldd = ldd;
  g.lds = lds;
  g.ntaps = ntaps;
  g.act = act;
  int dymin = 9, dymax = -9, dxmin = 9, dxmax = -9;
  for (int t = 0; t < ntaps; ++t) {
    g.dy[t] = (signed char)dy[t];
    g.dx[t] = (signed char)dx[t];
    dymin = std::min(dymin, dy[t]);
    dymax = std::max(dymax, dy[t]);
    dxmin = std::min(dxmin, dx[t]);
    dxmax = std::max(dxmax, dx[t]);
  }
  const int big = std::max(Ho, Wo);
  const int tdim = big > 8 ? 16 : (big > 4 ? 8 : 4);
  g.lth = g.ltw = ilog2(tdim);
  g.lnf = 8 - 2 * g.lth;  // th * tw * nf == 256
  g.dymin = dymin;
  g.dxmin = dxmin;
  g.hh = tdim + dymax - dymin;
  g.hw = tdim + dxmax - dxmin;
  g.tiles_x = (Wo + tdim - 1) / tdim;
  g.tiles_y = (Ho + tdim - 1) / tdim;
  const int nf = 1 << g.lnf;
  const int fblocks = (F + nf - 1) / nf;
  const int npix = nf * g.hh * g.hw;
  FH_REQUIRE(npix * 4 <= 9 * 256, "conv halo tile too large (%d pixels)", npix);
  const int bn = s.cout >= 64 ? 64 : 32;
  FH_REQUIRE(s.cout % bn == 0 && s.cin % 32 == 0, "conv channels (%d -> %d) not tileable", s.cin, s.cout);
  const size_t smem = (size_t)npix * 64 + (size_t)ntaps * bn * 64;
  dim3 grid(g.tiles_x * g.tiles_y * fblocks, s.cout / bn);
  fh_prof_begin(1, st);
  if (bn == 64) hipLaunchKernelGGL((dec_conv_kernel<T, 4>), grid, dim3(256), smem, st, g);
  else hipLaunchKernelGGL((dec_conv_kernel<T, 2>), grid, dim3(256), smem, st, g);
  fh_prof_end(1, st);
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

// Decode `n` frames (n <= max_frames) whose latents are s_r + r_d[i].
template <class T>
int decode_batch(float_dec* h, const float* s_r, const float* r_d, int n, float* out, int final_mode, hipStream_t st) {
  const int sdim = h->cfg.style_dim;
  int rc;
  // 1. every style modulation of every layer: styles[n][Stot]
  {
    constexpr int FB = 8;
    dim3 grid((h->Stot + 255) / 256, (n + FB - 1) / FB);
    hipLaunchKernelGGL((dec_small_gemm_kernel<SG_STYLE, FB>), grid, dim3(256), FB * sdim * sizeof(float), st, r_d, sdim, s_r,
                       h->WmT, sdim, h->Stot, h->bm, 1.0f / sqrtf((float)sdim), h->styles, h->Stot, n);
  }
  // 2. demodulation factors of the 15 StyledConvs: demod[n][Dtot]
  for (const Styled& s : h->convs) {
    constexpr int FB = 8;
    dim3 grid((s.cout + 255) / 256, (n + FB - 1) / FB);
    hipLaunchKernelGGL((dec_small_gemm_kernel<SG_DEMOD, FB>), grid, dim3(256), FB * s.cin * sizeof(float), st,
                       h->styles + s.style_off, h->Stot, (const float*)nullptr, s.WsqT, s.cin, s.cout, (const float*)nullptr,
                       1.0f / (float)(s.cin * 9), h->demod + s.demod_off, h->Dtot, n);
  }
  // 3. constant input * style(conv1), conv1 @ 4x4
  u16 *P = h->bufP, *Q = h->bufQ, *Z = h->bufZ;
  {
    const Styled& c1 = h->convs[0];
    const int tot = n * 16 * c1.cin;
    hipLaunchKernelGGL((dec_input_kernel<T>), dim3((tot + 255) / 256), dim3(256), 0, st, Q, h->cin_hwc, h->styles + c1.style_off,
                       h->Stot, n, 16, c1.cin);
    static const int dy9[9] = {-1, -1, -1, 0, 0, 0, 1, 1, 1}, dx9[9] = {-1, 0, 1, -1, 0, 1, -1, 0, 1};
    const Styled& nx = h->convs[1];
    if ((rc = launch_conv<T>(Q, 4, 4, c1, c1.W, 9, dy9, dx9, P, 4, 4, 4, 4, 1, 1, 0, 0, n, h->demod + c1.demod_off, h->Dtot,
                             c1.abias, 1, h->styles + nx.style_off, h->Stot, st)))
      return rc;
  }
  float *flow_prev = nullptr, *rgb_prev = nullptr, *flow_cur = h->flowA, *rgb_cur = h->rgbA;
  for (int li = 0; li < h->n_levels; ++li) {
    const Level& L = h->levels[li];
    const Styled& up = h->convs[1 + 2 * li];
    const Styled& c2 = h->convs[2 + 2 * li];
    const int R = L.R, Ri = R / 2;
    // 3a. transposed conv (stride 2) as four parity-class convolutions into z (R+1 x R+1), demodulated
    size_t t0 = 0;
    for (int pu = 0; pu < 2; ++pu)
      for (int pv = 0; pv < 2; ++pv) {
        const ClassTaps c = class_taps(pu, pv);
        if ((rc = launch_conv<T>(P, Ri, Ri, up, up.W + t0 * up.cout * up.cin, c.n, c.dy, c.dx, Z, Ri + 1 - pu, Ri + 1 - pv, R + 1,
                                 R + 1, 2, 2, pu, pv, n, h->demod + up.demod_off, h->Dtot, nullptr, 0, nullptr, 0, st)))
          return rc;
        t0 += c.n;
      }
    // 3b. FIR blur + bias + lrelu, scaled by conv2's style
    {
      const size_t tot = (size_t)n * R * R * (up.cout / 8);
      hipLaunchKernelGGL((dec_blur_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, Z, Q, n, R, up.cout, up.abias,
                         h->styles + c2.style_off, h->Stot);
    }
    // 3c. conv2 (plain 3x3), unscaled output feeds ToFlow
    {
      static const int dy9[9] = {-1, -1, -1, 0, 0, 0, 1, 1, 1}, dx9[9] = {-1, 0, 1, -1, 0, 1, -1, 0, 1};
      if ((rc = launch_conv<T>(Q, R, R, c2, c2.W, 9, dy9, dx9, P, R, R, R, R, 1, 1, 0, 0, n, h->demod + c2.demod_off, h->Dtot,
                               c2.abias, 1, nullptr, 0, st)))
        return rc;
    }
    // 3d. ToFlow + warp + blend + ToRGB
    {
      const bool last = (li == h->n_levels - 1);
      FlowArgs g;
      memset(&g, 0, sizeof(g));
      g.x = P;
      g.feat = L.feat;
      g.pflow = flow_prev;
      g.prgb = rgb_prev;
      g.wflow = L.wflow;
      g.sflow = h->styles + L.style_off;
      g.bflow = L.bflow;
      g.wrgb = L.wrgb;
      g.b1 = L.b1;
      g.b2 = L.b2;
      g.snext = last ? nullptr : h->styles + h->convs[1 + 2 * (li + 1)].style_off;
      g.xnext = last ? nullptr : Q;
      g.flow_out = flow_cur;
      g.rgb_out = rgb_cur;
      g.final_out = last ? out : nullptr;
      g.final_mode = last ? final_mode : 0;
      g.F = n;
      g.R = R;
      g.C = L.C;
      g.ld_s = h->Stot;
      const int ppb = 256 / (L.C / 8);
      int bx = (R * R + ppb - 1) / ppb;
      bx = std::min(bx, 2048);
      hipLaunchKernelGGL((dec_flow_kernel<T>), dim3(bx, n), dim3(256), 0, st, g);
    }
    std::swap(P, Q);  // next level's input is the blended tensor just written to Q
    flow_prev = flow_cur;
    rgb_prev = rgb_cur;
    flow_cur = (flow_cur == h->flowA) ? h->flowB : h->flowA;
    rgb_cur = (rgb_cur == h->rgbA) ? h->rgbB : h->rgbA;
  }
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

template <class T>
int frames_impl(float_dec* h, const float* s_r, const float* r_d, int n_frames, float* out, int final_mode, hipStream_t st) {
  const int F = h->cfg.max_frames, S = h->cfg.size, sdim = h->cfg.style_dim;
  for (int f0 = 0; f0 < n_frames; f0 += F) {
    const int n = std::min(F, n_frames - f0);
    int rc = decode_batch<T>(h, s_r, r_d + (size_t)f0 * sdim, n, out + (size_t)f0 * S * S * 3, final_mode, st);
    if (rc) return rc;
  }
  return FLOAT_OK;
}

template <class T>
int set_feats_impl(float_dec* h, const float* const* feats, hipStream_t st) {
  for (int li = 0; li < h->n_levels; ++li) {
    const Level& L = h->levels[li];
    const int tot = L.C * L.R * L.R;
    hipLaunchKernelGGL((dec_feat_pack_kernel<T>), dim3((tot + 255) / 256), dim3(256), 0, st, L.feat, feats[li], L.C, L.R * L.R);
  }
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

}  // namespace

extern "C" {

int float_dec_create(const float_dec_cfg_t* cfg, const float_tensor_t* tensors, int32_t n_tensors, float_dec_t** out) {
  FH_REQUIRE(cfg && tensors && out, "null argument to float_dec_create");
  FH_REQUIRE(cfg->size >= 64 && cfg->size <= 512 && (cfg->size & (cfg->size - 1)) == 0,
             "decoder size must be a power of two in [64, 512] (got %d)", cfg->size);
  FH_REQUIRE(cfg->style_dim > 0 && cfg->style_dim <= 2048, "style_dim %d unsupported", cfg->style_dim);
  FH_REQUIRE(cfg->max_frames >= 1 && cfg->max_frames <= 128, "max_frames must be in [1,128] (got %d)", cfg->max_frames);
  FH_REQUIRE(cfg->dtype == FLOAT_DT_BF16 || cfg->dtype == FLOAT_DT_FP16, "unknown dtype %d", cfg->dtype);
  float_dec* h = new float_dec();
  h->cfg = *cfg;
  TensorTable tt(tensors, n_tensors);
  int rc = (cfg->dtype == FLOAT_DT_BF16) ? create_impl<BF16>(h, tt) : create_impl<FP16>(h, tt);
  if (rc) {
    float_dec_destroy(h);
    return rc;
  }
  *out = h;
  return FLOAT_OK;
}

void float_dec_destroy(float_dec_t* h) {
  if (!h) return;
  h->pool.release();
  delete h;
}

int float_dec_set_feats(float_dec_t* h, const float* const* feats, int32_t n_feats, void* stream) {
  FH_REQUIRE(h && feats, "null argument to float_dec_set_feats");
  FH_REQUIRE(n_feats == h->n_levels, "expected %d feature maps (8..%d), got %d", h->n_levels, h->cfg.size, n_feats);
  for (int i = 0; i < n_feats; ++i) FH_REQUIRE(feats[i] != nullptr, "feats[%d] is null", i);
  hipStream_t st = (hipStream_t)stream;
  int rc = h->cfg.dtype == FLOAT_DT_BF16 ? set_feats_impl<BF16>(h, feats, st) : set_feats_impl<FP16>(h, feats, st);
  if (!rc) h->feats_set = true;
  return rc;
}

static int dec_run(float_dec_t* h, const float* s_r, const float* r_d, int32_t n_frames, float* out, int mode, void* stream) {
  FH_REQUIRE(h && s_r && r_d && out, "null argument to float_dec_frames");
  FH_REQUIRE(h->feats_set, "float_dec_set_feats must be called before decoding");
  FH_REQUIRE(n_frames >= 1, "n_frames must be >= 1 (got %d)", n_frames);
  hipStream_t st = (hipStream_t)stream;
  return h->cfg.dtype == FLOAT_DT_BF16 ? frames_impl<BF16>(h, s_r, r_d, n_frames, out, mode, st)
                                       : frames_impl<FP16>(h, s_r, r_d, n_frames, out, mode, st);
}

int float_dec_frames(float_dec_t* h, const float* s_r, const float* r_d, int32_t n_frames, float* out_hwc, void* stream) {
  return dec_run(h, s_r, r_d, n_frames, out_hwc, 1, stream);
}

int float_dec_frames_raw(float_dec_t* h, const float* s_r, const float* r_d, int32_t n_frames, float* out_chw, void* stream) {
  return dec_run(h, s_r, r_d, n_frames, out_chw, 2, stream);
}

}  // extern "C"
