// C-ABI entry points of the decoder operator (include/float_hip.h): weight packing and the
// per-batch launch chain of Synthesis.forward (reference styledecoder.py:497-534).
#include <math.h>
#include <stdlib.h>

#include "dec_kernels.hpp"

namespace {

struct Styled {  // one StyledConv (styledecoder.py:302-325)
  int cin = 0, cout = 0;
  bool up = false;
  void* W = nullptr;      // T::elem; plain: [9][Cout][Cin]; up: four parity classes, [4+2+2+1][Cout][Cin]
  float* WsqT = nullptr;  // [Cin][Cout] sum over taps of W^2 (fp32)
  float* abias = nullptr; // [Cout] FusedLeakyReLU bias
  int style_off = 0, demod_off = 0;
  // up: 1-D taps of the Blur behind the transposed conv (styledecoder.py:209-213: make_kernel(k) * 4, applied by upfirdn2d as a
  // true convolution): fir[b] = weight of z[X - 1 + b] in output X = 2 k[3 - b] / sum(k); {0.25, 0.75, 0.75, 0.25} for [1,3,3,1]
  float fir[4] = {0.25f, 0.75f, 0.75f, 0.25f};
};

struct Level {  // ToFlow + ToRGB of one resolution
  int R = 0, C = 0;
  float *wflow = nullptr, *bflow = nullptr, *wrgb = nullptr, *b1 = nullptr, *b2 = nullptr;
  float* lin = nullptr;  // [R] np.linspace(-1, 1, R) as float32
  int style_off = 0;
  void* feat = nullptr;  // [R][R][C] T::elem
  float* grgb = nullptr;  // [R][R][4]: ToRGB's conv of `feat` (dec_feat_rgb_kernel, refreshed whenever the features are set)
  float upk_flow[8], upk_rgb[8];  // per-axis taps of the two Upsamples (upsample_taps)
};

int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

}  // namespace

constexpr int kStyleCap = 256;

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
  float* dirQ = nullptr;  // [style_dim][motion_dim] of QR(direction.weight + 1e-8) (styledecoder.py:435-436), if present
  int motion_dim = 0;
  // workspace.  Frames go through the decoder in three nested batches:
  //   style batch (<= kStyleCap frames): every style modulation + demod factor in two launches;
  //   low batch   (<= lo_frames): levels up to 32x32, where one frame is only 16..1024 pixels and
  //                               the 512-channel weights dominate, so many frames share a launch;
  //   high batch  (<= max_frames): levels 64..size, where activations are 17 MB per frame.
  int lo_levels = 0, lo_frames = 0;
  float *styles = nullptr, *demod = nullptr, *eps = nullptr;  // eps: [kStyleCap][16] = 1e-8 / (style normaliser)^2
  void *loA = nullptr, *loB = nullptr, *loZ = nullptr, *loX = nullptr;  // T::elem: low phase ping/pong/z + hand-over tensor
  void *hiA = nullptr, *hiB = nullptr, *hiZ = nullptr;
  // last level with <= 64 channels (round 6): ToFlow's conv in conv2's epilogue (dec_conv16_kernel FLOWM) - its per-frame weight
  // fragments and the [max_frames][size][size][4] sums that replace the stored V; FLOAT_DEC_FLOW_EPI=0 keeps dec_flow_kernel there
  void* wfrag = nullptr;
  float* oflow = nullptr;
  unsigned long long* sat = nullptr;  // [kDecSatSites] saturation counters (dec_kernels.hpp), device
  bool style_norm = true;
  float *loFlow[2] = {nullptr, nullptr}, *loRgb[2] = {nullptr, nullptr};
  float *hiFlow[2] = {nullptr, nullptr}, *hiRgb[2] = {nullptr, nullptr};
  // float_dec_frames_host, ride-along mode: the frames of the previous high batch still to be copied to the host by copy
  // workgroups inside the next batch's launches (CopyTail, dec_kernels.hpp)
  struct {
    const float* src = nullptr;
    float* dst = nullptr;       // device-side address of the pinned destination (what the copy workgroups store through)
    float* dst_host = nullptr;  // the same position as the caller's host pointer (hipMemcpyAsync of what no launch took)
    size_t left16 = 0;   // 16-byte units not yet handed to a launch
    double wleft = 0.0;  // sum of the weights of the carrying launches still to come in this batch
  } ride;
  std::vector<hipEvent_t> copy_events;  // float_dec_frames_host: one per high batch in flight, recycled across calls
  hipEvent_t join_event = nullptr;
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
int upload_elem(DevicePool* pool, const std::vector<float>& src, void** dst) {
  typedef typename T::elem E;
  std::vector<E> tmp(src.size());
  for (size_t i = 0; i < src.size(); ++i) tmp[i] = T::host_from_float(src[i]);
  E* d = nullptr;
  int rc = pool->alloc(&d, tmp.size(), false);
  if (rc) return rc;
  *dst = d;
  FH_CHECK_HIP(hipMemcpy(d, tmp.data(), tmp.size() * sizeof(E), hipMemcpyHostToDevice));
  return FLOAT_OK;
}

template <class T>
int alloc_elem(DevicePool* pool, void** dst, size_t count) {
  typename T::elem* d = nullptr;
  int rc = pool->alloc(&d, count, true);
  *dst = d;
  return rc;
}

int upload32(DevicePool* pool, const std::vector<float>& src, float** dst) {
  int rc = pool->alloc(dst, src.size(), false);
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
  // taps listed with ascending input offset (dy = -1 first), the order dec_conv16_kernel enumerates
  const int kys[2][2] = {{2, 0}, {1, -1}}, dys[2][2] = {{-1, 0}, {0, 0}};
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

// 1-D taps of an up-sampling FIR (the Blur behind a transposed conv, styledecoder.py:209-213).  What the reference ends up
// with: Synthesis(blur_kernel=...) builds make_kernel(k) * 4 = outer(k, k) * 4 / sum(k)^2 as a registered BUFFER, and the strict
// load_state_dict (nodes_vadv_loader.py:632) then overwrites it with the checkpoint's `<conv>.blur.kernel` - so the checkpoint's
// buffer wins when it is there, the loader's widget (`blur_kernel`, optional tensor of 4 taps) only when it is not, [1,3,3,1]
// otherwise.  upfirdn2d convolves (correlates with the flipped kernel, styledecoder.py:28-29): fir[b] = weight of z[X - 1 + b]
// in output X.  A buffer must be a 4 x 4 outer product a (x) a (what make_kernel produces); anything else is refused.
int blur_taps(const TensorTable& tt, const std::string& buffer_key, float fir[4]) {
  const float_tensor_t* wk = tt.find("blur_kernel");
  if (wk && TensorTable::numel(wk) != 4) {
    fh_set_error("blur_kernel has %lld taps; the HIP decoder implements 4-tap kernels", (long long)TensorTable::numel(wk));
    return FLOAT_E_INVALID;
  }
  if (const float_tensor_t* kb = tt.find(buffer_key)) {
    if (TensorTable::numel(kb) != 16 || kb->ndim != 2 || kb->shape[0] != 4) {
      fh_set_error("'%s' is not a 4 x 4 kernel; the HIP decoder implements 4-tap blur kernels", buffer_key.c_str());
      return FLOAT_E_INVALID;
    }
    double r[4] = {0, 0, 0, 0}, S = 0, amax = 0;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        r[i] += kb->data[i * 4 + j];
        S += kb->data[i * 4 + j];
        amax = std::max(amax, (double)fabsf(kb->data[i * 4 + j]));
      }
    if (!(S > 1e-12)) {
      fh_set_error("'%s' does not have a positive sum", buffer_key.c_str());
      return FLOAT_E_INVALID;
    }
    double a[4];
    for (int i = 0; i < 4; ++i) a[i] = r[i] / sqrt(S);  // K = a (x) a  =>  row sums = a_i * sum(a), S = sum(a)^2
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j)
        if (fabs(kb->data[i * 4 + j] - a[i] * a[j]) > 1e-5 * amax) {
          fh_set_error("'%s' is not an outer product k (x) k (make_kernel's form); other blur kernels are not implemented", buffer_key.c_str());
          return FLOAT_E_INVALID;
        }
    for (int b = 0; b < 4; ++b) fir[b] = (float)a[3 - b];
    return FLOAT_OK;
  }
  if (wk) {
    double sum = 0;
    for (int b = 0; b < 4; ++b) sum += wk->data[b];
    if (!(fabs(sum) > 1e-12)) {
      fh_set_error("blur_kernel sums to zero");
      return FLOAT_E_INVALID;
    }
    for (int b = 0; b < 4; ++b) fir[b] = (float)(2.0 * wk->data[3 - b] / sum);
  }
  return FLOAT_OK;
}

// The Upsample of ToRGB / ToFlow (styledecoder.py:373,394) is built with its default [1,3,3,1] whatever the loader's widget says
// (:489-491): make_kernel(k) * 4 as a registered 4 x 4 BUFFER `upsample.kernel`, which the strict load (nodes_vadv_loader.py:632)
// overwrites with the checkpoint's.  dec_flow_kernel applies it per axis: the buffer must be a rank-1 4 x 4 matrix K = ky (x) kx
// (every make_kernel of a 1-D kernel is; a x b with different factors passes too); taps = {ky[4], kx[4]}.  No buffer in the
// state: (1, 3, 3, 1) / 4 per axis.  Another size (the Upsample's padding belongs to 4 taps) or a rank > 1 kernel is refused.
int upsample_taps(const TensorTable& tt, const std::string& key, float taps[8]) {
  static const float dflt[4] = {0.25f, 0.75f, 0.75f, 0.25f};
  for (int i = 0; i < 8; ++i) taps[i] = dflt[i & 3];
  const float_tensor_t* kb = tt.find(key);
  if (!kb) return FLOAT_OK;
  if (TensorTable::numel(kb) != 16 || kb->ndim != 2 || kb->shape[0] != 4) {
    fh_set_error("'%s' is not a 4 x 4 kernel; ToRGB / ToFlow up-sampling kernels of other sizes are not implemented", key.c_str());
    return FLOAT_E_INVALID;
  }
  double r[4] = {0, 0, 0, 0}, c[4] = {0, 0, 0, 0}, S = 0, amax = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      const double v = kb->data[i * 4 + j];
      r[i] += v, c[j] += v, S += v;
      amax = std::max(amax, fabs(v));
    }
  if (!(S > 1e-12)) {
    fh_set_error("'%s' sums to %.3g: the per-axis split K = ky (x) kx of the flow kernel needs a positive sum (INTEGRATION.md, "
                 "'FIR buffers'); such an up-sampling kernel is not implemented", key.c_str(), S);
    return FLOAT_E_INVALID;
  }
  // K_ij = u_i v_j  =>  row sums u_i sum(v), column sums v_j sum(u), S = sum(u) sum(v): K_ij = r_i c_j / S
  const double rs = sqrt(S);
  double resid = 0.0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) resid = std::max(resid, fabs(kb->data[i * 4 + j] - r[i] * c[j] / S));
  if (resid > 1e-5 * amax) {
    fh_set_error("'%s' is not a rank-1 kernel ky (x) kx (make_kernel's form): max |K - r c^T / sum| = %.3g against the limit 1e-5 * max|K| "
                 "= %.3g; other up-sampling kernels are not implemented (INTEGRATION.md, 'FIR buffers')", key.c_str(), resid, 1e-5 * amax);
    return FLOAT_E_INVALID;
  }
  for (int i = 0; i < 4; ++i) {
    taps[i] = (float)(r[i] / rs);
    taps[4 + i] = (float)(c[i] / rs);
  }
  return FLOAT_OK;
}

template <class T>
int pack_styled(DevicePool* pool, const TensorTable& tt, const std::string& p, int cin, int cout, bool up, Styled* s,
                std::vector<float>* WmT_host, std::vector<float>* bm_host, int style_dim) {
  s->cin = cin;
  s->cout = cout;
  s->up = up;
  const float_tensor_t* w = need(tt, p + ".conv.weight", (int64_t)cout * cin * 9);
  const float_tensor_t* mw = need(tt, p + ".conv.modulation.weight", (int64_t)cin * style_dim);
  const float_tensor_t* mb = need(tt, p + ".conv.modulation.bias", cin);
  const float_tensor_t* ab = need(tt, p + ".activate.bias", cout);
  if (!w || !mw || !mb || !ab) return FLOAT_E_MISSING;
  if (up) {
    int rc = blur_taps(tt, p + ".conv.blur.kernel", s->fir);
    if (rc) return rc;
  }
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
  if ((rc = upload_elem<T>(pool, packed, &s->W))) return rc;
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
  if ((rc = upload32(pool, wsq, &s->WsqT))) return rc;
  if ((rc = upload32(pool, std::vector<float>(ab->data, ab->data + cout), &s->abias))) return rc;
  s->style_off = (int)bm_host->size();
  for (int i = 0; i < cin; ++i) bm_host->push_back(mb->data[i]);
  WmT_host->insert(WmT_host->end(), mw->data, mw->data + (size_t)cin * style_dim);  // [cin][style_dim], transposed later
  return FLOAT_OK;
}

// dynamic LDS above the 64 KiB default: 64 KiB per workgroup with 16-bit operands (2 workgroups per CU), twice that in the
// fp32 verification mode (the z tile of dec_zblur_kernel: 32 x 32 x 128 B)
static int env_int(const char* name, int dflt, int lo, int hi);
template <class T>
int raise_lds_limits() {
  const int lim = std::max(32 * 1024 * T::EB, env_int("FLOAT_DEC_LDS_PAD", 0, 0, 160 * 1024));
#define CONV16_ATTR(NTv, TYv, TXv) \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_conv16_kernel<T, NTv, TYv, TXv>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  CONV16_ATTR(4, 3, 3) CONV16_ATTR(2, 3, 3)
#undef CONV16_ATTR
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_conv16_kernel<T, 4, 3, 3, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_conv16_kernel<T, 2, 3, 3, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_flowlast_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  if constexpr (!T::is32) {  // the double-buffered form (FLOAT_DEC_CONV_DB): two buffer sets, up to 115 KB
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_conv16_kernel<T, 4, 3, 3, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_conv16_kernel<T, 2, 3, 3, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  }
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_conv_kernel<T, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_conv_kernel<T, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_zconv4_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_zblur_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
#define FLOW_ATTR(PIXv, LASTv) \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_flow_kernel<T, PIXv, LASTv>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  FLOW_ATTR(4, false) FLOW_ATTR(2, false) FLOW_ATTR(1, false) FLOW_ATTR(4, true) FLOW_ATTR(2, true) FLOW_ATTR(1, true)
#undef FLOW_ATTR
  (void)hipGetLastError();
  return FLOAT_OK;
}

template <class T>
int create_impl(float_dec* h, const TensorTable& tt) {
  const int size = h->cfg.size, sdim = h->cfg.style_dim;
  const int log_size = ilog2(size);
  // channels by log2(resolution): the reference's table (styledecoder.py:457-467: 512 up to 32 px, then 256 / 128 / 64 / 32 / 16
  // times channel_multiplier) is read off the checkpoint itself - the output channels of every level's convs - so any
  // channel_multiplier loads (the kernels take channel counts in multiples of 32, the flow kernel up to 512)
  int chan[12] = {0, 0, 512, 512, 512, 512, 256, 128, 64, 32, 16, 0};
  {
    const float_tensor_t* c1 = tt.find("conv1.conv.weight");
    if (c1 && c1->ndim == 5) chan[2] = (int)c1->shape[1];
    for (int li = 0; li < log_size - 2; ++li) {
      const float_tensor_t* w = tt.find("convs." + std::to_string(2 * li) + ".conv.weight");
      if (w && w->ndim == 5) chan[li + 3] = (int)w->shape[1];
    }
    for (int l = 2; l <= log_size; ++l)
      FH_REQUIRE(chan[l] >= 32 && chan[l] <= 512 && chan[l] % 32 == 0 && (chan[l] & (chan[l] - 1)) == 0,
                 "decoder level %d px has %d channels: the HIP decoder takes powers of two in [32, 512]", 1 << l, chan[l]);
  }
  h->n_levels = log_size - 2;
  std::vector<float> wm_rows, bm_host;  // rows = modulation outputs, [Stot][sdim]
  int rc;
  h->convs.resize(1 + 2 * h->n_levels);
  if ((rc = pack_styled<T>(&h->pool, tt, "conv1", chan[2], chan[2], false, &h->convs[0], &wm_rows, &bm_host, sdim))) return rc;
  int cin = chan[2];
  for (int li = 0; li < h->n_levels; ++li) {
    const int cout = chan[li + 3];
    if ((rc = pack_styled<T>(&h->pool, tt, "convs." + std::to_string(2 * li), cin, cout, true, &h->convs[1 + 2 * li], &wm_rows,
                             &bm_host, sdim)))
      return rc;
    if ((rc = pack_styled<T>(&h->pool, tt, "convs." + std::to_string(2 * li + 1), cout, cout, false, &h->convs[2 + 2 * li],
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
    if ((rc = upsample_taps(tt, pf + ".upsample.kernel", L.upk_flow)) || (rc = upsample_taps(tt, pr + ".upsample.kernel", L.upk_rgb))) return rc;
    const float sc = 1.0f / sqrtf((float)L.C);  // 1x1: fan_in = C (styledecoder.py:134,223)
    std::vector<float> a(3 * L.C), b(3 * L.C);
    for (int i = 0; i < 3 * L.C; ++i) {
      a[i] = fw->data[i] * sc;
      b[i] = rw->data[i] * sc;
    }
    if ((rc = upload32(&h->pool, a, &L.wflow))) return rc;
    if ((rc = upload32(&h->pool, b, &L.wrgb))) return rc;
    if ((rc = upload32(&h->pool, std::vector<float>(fb->data, fb->data + 3), &L.bflow))) return rc;
    if ((rc = upload32(&h->pool, std::vector<float>(rb1->data, rb1->data + 3), &L.b1))) return rc;
    if ((rc = upload32(&h->pool, std::vector<float>(rb2->data, rb2->data + 3), &L.b2))) return rc;
    L.style_off = (int)bm_host.size();
    for (int i = 0; i < L.C; ++i) bm_host.push_back(fmb->data[i]);
    wm_rows.insert(wm_rows.end(), fmw->data, fmw->data + (size_t)L.C * sdim);
    if ((rc = alloc_elem<T>(&h->pool, &L.feat, (size_t)L.R * L.R * L.C))) return rc;
    if ((rc = h->pool.alloc(&L.grgb, (size_t)L.R * L.R * 4, true))) return rc;
    {
      // np.linspace(-1, 1, R): start + i*step in float64, last element forced to stop, cast to f32
      std::vector<float> lin(L.R);
      const double step = 2.0 / (double)(L.R - 1);
      for (int i = 0; i < L.R; ++i) lin[i] = (float)(-1.0 + (double)i * step);
      lin[L.R - 1] = 1.0f;
      if ((rc = upload32(&h->pool, lin, &L.lin))) return rc;
    }
  }
  h->Stot = (int)bm_host.size();
  std::vector<float> wmT((size_t)sdim * h->Stot);
  for (int j = 0; j < h->Stot; ++j)
    for (int k = 0; k < sdim; ++k) wmT[(size_t)k * h->Stot + j] = wm_rows[(size_t)j * sdim + k];
  if ((rc = upload32(&h->pool, wmT, &h->WmT))) return rc;
  if ((rc = upload32(&h->pool, bm_host, &h->bm))) return rc;
  // ConstantInput (1,512,4,4) -> HWC
  const float_tensor_t* ci = need(tt, "input.input", (int64_t)chan[2] * 16);
  if (!ci) return FLOAT_E_MISSING;
  std::vector<float> hwc((size_t)16 * chan[2]);
  for (int c = 0; c < chan[2]; ++c)
    for (int p = 0; p < 16; ++p) hwc[(size_t)p * chan[2] + c] = ci->data[(size_t)c * 16 + p];
  if ((rc = upload32(&h->pool, hwc, &h->cin_hwc))) return rc;
  // workspace
  h->lo_levels = 0;
  for (int li = 0; li < h->n_levels - 1; ++li)
    if ((8 << li) <= 32) h->lo_levels = li + 1;  // levels 8,16,32 (never the last level)
  h->lo_frames = std::min(128, std::max(h->cfg.max_frames, 8 * h->cfg.max_frames));
  const size_t FH = (size_t)h->cfg.max_frames, FL = (size_t)h->lo_frames;
  size_t act_lo = 16 * (size_t)chan[2], act_hi = 0, x_lo = 0;
  for (int li = 0; li < h->n_levels; ++li) {
    const size_t R = 8u << li;
    const size_t need = std::max((R + 1) * (R + 1) * (size_t)chan[li + 3], (R / 2) * (R / 2) * (size_t)chan[li + 2]);
    if (li < h->lo_levels) {
      act_lo = std::max(act_lo, need);
      x_lo = R * R * (size_t)chan[li + 3];
    } else {
      act_hi = std::max(act_hi, need);
    }
  }
  if ((rc = h->pool.alloc(&h->styles, (size_t)kStyleCap * h->Stot, true))) return rc;
  if ((rc = h->pool.alloc(&h->demod, (size_t)kStyleCap * h->Dtot, true))) return rc;
  if ((rc = h->pool.alloc(&h->eps, (size_t)kStyleCap * 16, true))) return rc;
  if ((rc = h->pool.alloc(&h->sat, (size_t)kDecSatSites, true))) return rc;
  if ((rc = alloc_elem<T>(&h->pool, &h->loA, FL * act_lo))) return rc;
  if ((rc = alloc_elem<T>(&h->pool, &h->loB, FL * act_lo))) return rc;
  if ((rc = alloc_elem<T>(&h->pool, &h->loZ, FL * act_lo))) return rc;
  if ((rc = alloc_elem<T>(&h->pool, &h->loX, FL * std::max(x_lo, (size_t)16 * chan[2])))) return rc;
  if ((rc = alloc_elem<T>(&h->pool, &h->hiA, FH * act_hi))) return rc;
  if ((rc = alloc_elem<T>(&h->pool, &h->hiB, FH * act_hi))) return rc;
  if ((rc = alloc_elem<T>(&h->pool, &h->hiZ, FH * act_hi))) return rc;
  {
    static const bool flow_epi = env_int("FLOAT_DEC_FLOW_EPI", 1, 0, 1) != 0;
    const int cl = h->levels.back().C;
    if (flow_epi && (cl == 32 || cl == 64) && size >= 16 && size % 16 == 0) {
      if ((rc = alloc_elem<T>(&h->pool, &h->wfrag, FH * (size_t)(cl / 32) * (T::is32 ? 1 : 2) * 64 * 8))) return rc;
      if ((rc = h->pool.alloc(&h->oflow, FH * (size_t)size * size * 4, true))) return rc;
    }
  }
  const size_t sk_lo = FL * 32 * 32 * 4, sk_hi = FH * (size_t)size * size * 4;  // flow / rgb pyramids: 4 floats per pixel
  for (int i = 0; i < 2; ++i) {
    if ((rc = h->pool.alloc(&h->loFlow[i], sk_lo, true))) return rc;
    if ((rc = h->pool.alloc(&h->loRgb[i], sk_lo, true))) return rc;
    if ((rc = h->pool.alloc(&h->hiFlow[i], sk_hi, true))) return rc;
    if ((rc = h->pool.alloc(&h->hiRgb[i], sk_hi, true))) return rc;
  }
  return raise_lds_limits<T>();
}

// Copy workgroups per carrying launch (a multiple of 8), the lowest resolution whose launches carry a share, and the pause
// between a wave's 1-KiB stores in units of 512 clocks.  Unpaced, the copy saturates PCIe (55 GB/s) and its posted writes
// queue in front of the compute workgroups' memory traffic: the 512-px flow launch took 665 us instead of 508 with a 217 us
// copy inside; at ~45 GB/s (16 workgroups, pace 4) it takes 548 (in-kernel stamps, -DDEC_STAMPS).  Round 3: the launches got
// shorter (flow kernel -35 %), pace 3 (~52 GB/s) leaves less of the last share exposed: 27.66 vs 28.03 ms per 250 frames.
static int env_int(const char* name, int dflt, int lo, int hi) {  // tuning knobs: anything outside [lo, hi] falls back to the default
  const char* v = getenv(name);
  if (!v || !*v) return dflt;
  char* end = nullptr;
  const long x = strtol(v, &end, 10);
  return (end && *end == 0 && x >= lo && x <= hi) ? (int)x : dflt;
}
// FLOAT_DEC_LDS_PAD=<bytes>: every launch of the level kernels (3x3 conv, up-conv + blur, flow) asks for at least that much dynamic
// LDS, i.e. the decoder's occupancy is capped (82 000: ONE workgroup per CU instead of two - room for a 98-KB workgroup of the FMT
// chain beside it when the two stages overlap on two streams, pipeline.generate_to_host_overlap).  0 = off.
static const int kLdsPad = env_int("FLOAT_DEC_LDS_PAD", 0, 0, 160 * 1024);
static inline size_t dec_smem(size_t need) { return std::max(need, (size_t)kLdsPad); }
static const unsigned kRideWgs = (unsigned)env_int("FLOAT_DEC_RIDE_WGS", 16, 0, 64) / 8 * 8;
static const int kRideMinRes = env_int("FLOAT_DEC_RIDE_MIN_RES", 64, 64, 512);
static const unsigned kRidePace = (unsigned)env_int("FLOAT_DEC_RIDE_PACE", 3, 0, 64);

// Weight of a carrying launch (kind 0 = up-conv + blur, 1 = conv2, 2 = flow / warp / ToRGB) at resolution R: the share of the
// pending copy it takes is proportional to it, so that every share ends inside its launch (per 32-frame batch, ~us; other
// resolutions: equal shares).
static double ride_weight(int R, int kind) {
  static const bool equal = getenv("FLOAT_DEC_RIDE_EQUAL") != nullptr;
  if (equal) return 1.0;
  // launch durations (tools/probes/trace_sequence.py on the round-3 kernels: {203,155,61},{178,175,118},{242,226,215},{318,256,336})
  // shifted toward the flow launches, which absorb a share without getting longer while the 512-px convs are stretched by theirs
  // (a trace of the carrying batch: +49 / +56 us there, +6 on the flow launch): 26.35 vs 26.80 ms per 250 frames decode + hand-over
  // round 6 (launches now {178,142,52},{158,155,99},{220,206,175},{277,235,146}: the last level's flow launch is a third of what it
  // was): decode + hand-over per 250 frames 25.1-25.3 ms with the row below against 26.4-26.9 with round 5's {..,{290,225,370}},
  // 25.4 / 25.7 / 25.3 / 25.6 for four neighbours, 26.8 with equal shares (tools/probes/dec_host2.py, FLOAT_DEC_RIDE_W)
  static double w[4][3] = {{170, 140, 50}, {170, 155, 100}, {220, 200, 180}, {300, 250, 120}};
  static const bool tuned = [] {  // tuning aid: FLOAT_DEC_RIDE_W="12 numbers", rows 64 / 128 / 256 / 512 px x (up-conv, conv2, flow)
    const char* e = getenv("FLOAT_DEC_RIDE_W");
    double t[12];
    if (e && sscanf(e, "%lf %lf %lf %lf %lf %lf %lf %lf %lf %lf %lf %lf", t, t + 1, t + 2, t + 3, t + 4, t + 5, t + 6, t + 7, t + 8, t + 9, t + 10, t + 11) == 12)
      for (int i = 0; i < 12; ++i)
        if (t[i] > 0.0 && t[i] < 1e6) w[i / 3][i % 3] = t[i];
    return true;
  }();
  (void)tuned;
  const int li = R == 64 ? 0 : R == 128 ? 1 : R == 256 ? 2 : R == 512 ? 3 : -1;
  return li < 0 ? 250.0 : w[li][kind];
}

// The share of the pending device-to-host copy that the next carrying launch takes.
static CopyTail take_ride(float_dec* h, int R, int kind) {
  CopyTail ct{};
  auto& r = h->ride;
  if (!r.left16 || r.wleft <= 0.0 || !kRideWgs || R < kRideMinRes) return ct;
  const double w = ride_weight(R, kind);
  size_t n = (size_t)((double)r.left16 * std::min(1.0, w / r.wleft)) + 1;
  n = std::min(n, r.left16);
  r.wleft -= w;
  if (r.wleft < 1e-9) n = r.left16;  // the batch's last carrier takes what is left
  ct.src = reinterpret_cast<const u32x4*>(r.src);
  ct.dst = reinterpret_cast<u32x4*>(r.dst);
  ct.n16 = n;
  ct.nwg = kRideWgs;
  ct.pace = kRidePace;
#ifdef DEC_STAMPS  // diagnostic build only: probes that give wrong frames
  static const int test = env_int("FLOAT_DEC_RIDE_TEST", 0, 0, 2);
  if (test == 1) ct.dst = const_cast<u32x4*>(ct.src);  // device -> device instead of device -> host
  if (test == 2) ct.n16 = 1;                            // copy workgroups with nothing to do
#endif
  r.src += n * 4;
  r.dst += n * 4;
  r.dst_host += n * 4;
  r.left16 -= n;
  return ct;
}

// FLOAT_DEC_CB_ORDER=0: output-channel blocks as grid.y (the round-1 order; A/B switch of dec_group_cb)
static const bool g_dec_cb_order = !(getenv("FLOAT_DEC_CB_ORDER") && atoi(getenv("FLOAT_DEC_CB_ORDER")) == 0);

// h != nullptr: the launch may carry a share of the pending device-to-host copy (16x16-tile kernel only)
template <class T>
int launch_conv(float_dec* h, const void* X, int Hi, int Wi, const Styled& s, const void* Wt, int ntaps, const int* dy, const int* dx,
                void* Y, int Ho, int Wo, int OH, int OW, int sy, int sx, int py, int px, int F, const float* demod, int ldd,
                const float* bias, int act, const float* snext, int lds, unsigned long long* sat, hipStream_t st,
                const void* wfrag = nullptr, float* oflow = nullptr) {
  constexpr size_t RB = 32 * T::EB;
  ConvArgs g;
  memset(&g, 0, sizeof(g));
  g.wfrag = wfrag;
  g.oflow = oflow;
  g.sat = sat;
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
  // Output channels per workgroup: 64 (NT = 4) in the 16x16-tile kernel where the layer has them - each A fragment feeds twice
  // the MFMAs (22.75 vs 23.10 ms per 250 frames since the kernel's address arithmetic went; before that the 32-channel tiles'
  // doubled workgroup count won, 33.9 vs 35.2) - and 32 in the generic low-resolution kernel; FLOAT_DEC_CONV_BN /
  // FLOAT_DEC_CONV_BN_LO = 32 | 64 are the A/B switches.
  static const int bn_hi = env_int("FLOAT_DEC_CONV_BN", 64, 32, 64);
  static const int bn_lo = env_int("FLOAT_DEC_CONV_BN_LO", 32, 32, 64);
  const bool tile16 = tdim == 16;
  const int bn = (s.cout >= 64 && (tile16 ? bn_hi : bn_lo) == 64) ? 64 : 32;
  FH_REQUIRE(s.cout % bn == 0 && s.cin % 32 == 0, "conv channels (%d -> %d) not tileable", s.cin, s.cout);
  const int ty_taps = dymax - dymin + 1, tx_taps = dxmax - dxmin + 1;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const bool prof = fh_prof_pair(1, &e0, &e1);
  if (tdim == 16 && ty_taps == 3 && tx_taps == 3 && ntaps == 9 && Ho % 16 == 0 && Wo % 16 == 0 && Ho == Hi && Wo == Wi) {
    // dense TY x TX window on 16x16 tiles: compile-time geometry, swizzled LDS, register prefetch
    const int total = g.tiles_x * g.tiles_y * F;
    static const int tpw_env = env_int("FLOAT_DEC_TPW", 0, 0, 4096);  // tuning aid
    g.tpw = tpw_env ? tpw_env : (total >= 16384 ? 4 : (total >= 4096 ? 2 : 1));
    const size_t smem = (size_t)(15 + ty_taps) * (15 + tx_taps) * RB + (size_t)ntaps * bn * RB + 3 * bn * sizeof(float);  // halo, weights, epilogue operands
    if (h) g.ct = take_ride(h, Ho, 1);
    dim3 grid((total + g.tpw - 1) / g.tpw + g.ct.nwg, s.cout / bn);
    if (g_dec_cb_order && s.cout / bn > 1) {  // channel blocks of a tile group side by side on one XCD (dec_group_cb)
      g.ngroups = (unsigned)((total + g.tpw - 1) / g.tpw);
      g.ncb = (unsigned)(s.cout / bn);
      grid = dim3(g.ngroups * g.ncb + g.ct.nwg, 1);
    }
    // FLOAT_DEC_CONV_DB bit mask: 1 = double-buffered LDS for the 64-channel tiles, 2 = for the 32-channel tiles (A/B switch)
    static const int db_env = env_int("FLOAT_DEC_CONV_DB", 0, 0, 3);
    const bool db = !oflow && !T::is32 && ((bn == 64 && (db_env & 1)) || (bn == 32 && (db_env & 2)));
    FH_REQUIRE(!oflow || s.cout == bn, "ToFlow epilogue needs the layer's %d output channels in one block of %d", s.cout, bn);
#define CONV16(NTv, TYv, TXv)                                                                                    \
  if (bn == NTv * 16 && ty_taps == TYv && tx_taps == TXv) {                                                       \
    if constexpr (!T::is32) {                                                                                     \
      if (db) {                                                                                                   \
        if (prof) hipExtLaunchKernelGGL((dec_conv16_kernel<T, NTv, TYv, TXv, 1>), grid, dim3(256), 2 * smem, st, e0, e1, 0, g); \
        else hipLaunchKernelGGL((dec_conv16_kernel<T, NTv, TYv, TXv, 1>), grid, dim3(256), 2 * smem, st, g);      \
      }                                                                                                           \
    }                                                                                                             \
    if (!db && oflow) { /* ToFlow in the epilogue: the whole channel range in one block (cout == bn) */           \
      if (prof) hipExtLaunchKernelGGL((dec_conv16_kernel<T, NTv, TYv, TXv, 0, 1>), grid, dim3(256), dec_smem(smem), st, e0, e1, 0, g); \
      else hipLaunchKernelGGL((dec_conv16_kernel<T, NTv, TYv, TXv, 0, 1>), grid, dim3(256), dec_smem(smem), st, g); \
    } else if (!db) {                                                                                             \
      if (prof) hipExtLaunchKernelGGL((dec_conv16_kernel<T, NTv, TYv, TXv>), grid, dim3(256), dec_smem(smem), st, e0, e1, 0, g); \
      else hipLaunchKernelGGL((dec_conv16_kernel<T, NTv, TYv, TXv>), grid, dim3(256), dec_smem(smem), st, g);     \
    }                                                                                                             \
  }
    CONV16(4, 3, 3) CONV16(2, 3, 3)
#undef CONV16
  } else {
    FH_REQUIRE(!oflow, "ToFlow epilogue: only on the 16x16-tile 3x3 kernel (%d x %d)", Ho, Wo);
    const size_t smem = (size_t)npix * RB + (size_t)ntaps * bn * RB;
    dim3 grid(g.tiles_x * g.tiles_y * fblocks, s.cout / bn);
    if (prof) {
      if (bn == 64) hipExtLaunchKernelGGL((dec_conv_kernel<T, 4>), grid, dim3(256), smem, st, e0, e1, 0, g);
      else hipExtLaunchKernelGGL((dec_conv_kernel<T, 2>), grid, dim3(256), smem, st, e0, e1, 0, g);
    } else {
      if (bn == 64) hipLaunchKernelGGL((dec_conv_kernel<T, 4>), grid, dim3(256), smem, st, g);
      else hipLaunchKernelGGL((dec_conv_kernel<T, 2>), grid, dim3(256), smem, st, g);
    }
  }
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

static const int kDy9[9] = {-1, -1, -1, 0, 0, 0, 1, 1, 1}, kDx9[9] = {-1, 0, 1, -1, 0, 1, -1, 0, 1};

// The up-sampling StyledConv (styledecoder.py:302-325 with upsample=True: conv_transpose2d stride 2 -> Blur -> + bias ->
// leaky_relu * sqrt2) for `n` frames: x_in (Ri x Ri, already scaled by the layer's style) -> *U_out (2Ri x 2Ri, scaled by
// `snext`, the consumer's style).  Three forms by size: transposed conv + blur in one launch from `zblur_min` px up (the
// result lands in Zb: x_in may alias U), all four parity classes in one launch + blur kernel from 16 px up, class by class
// through the generic kernel below.  h != nullptr: the fused launch may carry a share of the pending device-to-host copy.
template <class T>
int launch_upconv(float_dec* h, const Styled& up, int Ri, int n, const void* x_in, void* Zb, void* U, void** U_out,
                  const float* demod, int ldd, const float* snext, int lds, unsigned long long* sat, hipStream_t st) {
  typedef typename T::elem E;
  constexpr size_t RB = 32 * T::EB;
  const int R = 2 * Ri;
  int rc;
  static const bool fuse_z = !getenv("FLOAT_DEC_NO_ZFUSE");
  static const int zblur_min = env_int("FLOAT_DEC_ZBLUR_MIN", 64, 16, 4096);
  if (R >= zblur_min && up.cout % 32 == 0 && up.cin % 32 == 0) {
    ConvArgs z;
    memset(&z, 0, sizeof(z));
    z.X = x_in;
    z.Wt = up.W;
    z.Y = Zb;
    z.demod = demod;
    z.bias = up.abias;
    z.snext = snext;
    z.lds = lds;
    z.F = n;
    z.Hi = z.Wi = Ri;
    z.Cin = up.cin;
    z.Cout = up.cout;
    z.OH = z.OW = R;
    z.ldd = ldd;
    z.sat = sat;
    for (int b = 0; b < 4; ++b) z.fir[b] = up.fir[b];
    z.fir_sym = (up.fir[0] == 0.25f && up.fir[1] == 0.75f && up.fir[2] == 0.75f && up.fir[3] == 0.25f) ? 1 : 0;
    z.tiles_x = z.tiles_y = (R + 27) / 28;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool prof = fh_prof_pair(1, &e0, &e1);
    if (h) z.ct = take_ride(h, R, 0);
    dim3 grid(z.tiles_x * z.tiles_y * n + z.ct.nwg, up.cout / 32);
    if (g_dec_cb_order && up.cout / 32 > 1) {
      z.ngroups = (unsigned)(z.tiles_x * z.tiles_y * n);
      z.ncb = (unsigned)(up.cout / 32);
      grid = dim3(z.ngroups * z.ncb + z.ct.nwg, 1);
    }
    const size_t smem = 32 * 32 * RB;
    if (prof) hipExtLaunchKernelGGL((dec_zblur_kernel<T>), grid, dim3(256), dec_smem(smem), st, e0, e1, 0, z);
    else hipLaunchKernelGGL((dec_zblur_kernel<T>), grid, dim3(256), dec_smem(smem), st, z);
    *U_out = Zb;
    FH_CHECK_HIP(hipGetLastError());
    return FLOAT_OK;
  }
  if (Ri + 1 > 8 && up.cout % 32 == 0 && fuse_z) {
    ConvArgs z;
    memset(&z, 0, sizeof(z));
    z.X = x_in;
    z.Wt = up.W;
    z.Y = Zb;
    z.demod = demod;
    z.F = n;
    z.Hi = z.Wi = Ri;
    z.Cin = up.cin;
    z.Cout = up.cout;
    z.OH = z.OW = R + 1;
    z.ldd = ldd;
    z.sat = sat;
    z.tiles_x = z.tiles_y = (Ri + 1 + 15) / 16;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool prof = fh_prof_pair(1, &e0, &e1);
    dim3 grid(z.tiles_x * z.tiles_y * n, up.cout / 32);
    const size_t smem = 17 * 17 * RB + 9 * 32 * RB;
    if (prof) hipExtLaunchKernelGGL((dec_zconv4_kernel<T>), grid, dim3(256), smem, st, e0, e1, 0, z);
    else hipLaunchKernelGGL((dec_zconv4_kernel<T>), grid, dim3(256), smem, st, z);
  } else {
    size_t t0 = 0;
    for (int pu = 0; pu < 2; ++pu)
      for (int pv = 0; pv < 2; ++pv) {
        const ClassTaps c = class_taps(pu, pv);
        if ((rc = launch_conv<T>(nullptr, x_in, Ri, Ri, up, reinterpret_cast<const E*>(up.W) + t0 * up.cout * up.cin, c.n, c.dy, c.dx, Zb,
                                 Ri + 1 - pu, Ri + 1 - pv, R + 1, R + 1, 2, 2, pu, pv, n, demod, ldd, nullptr, 0, nullptr, 0, sat, st)))
          return rc;
        t0 += c.n;
      }
  }
  // FIR blur + bias + lrelu, scaled by the consumer's style
  const size_t tot = (size_t)n * (R / 2) * (R / 4) * (up.cout / 8);
  hipLaunchKernelGGL((dec_blur_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const E*>(Zb),
                     reinterpret_cast<E*>(U), n, R, up.cout, up.abias, snext, lds, sat, up.fir[0], up.fir[1], up.fir[2], up.fir[3]);
  *U_out = U;
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

// ToFlow + warp + blend + ToRGB of one level (dec_flow_kernel); g holds everything but the grid.
// Grid: ~2048 workgroups in total (8 per CU) so that every lane group runs many pixel iterations and the per-workgroup
// prologue (56 per-lane weight values) is amortised; one row of workgroups per frame.
template <class T>
int launch_flow(float_dec* h, FlowArgs g, hipStream_t st) {
  if (g.oflow) {  // one lane per pixel (dec_flowlast_kernel)
    const int runs = (g.R * g.R + 255) / 256;
    if (h) g.ct = take_ride(h, g.R, 2);
    const size_t lpad = (size_t)kLdsPad;
    hipLaunchKernelGGL((dec_flowlast_kernel<T>), dim3(runs * g.F + g.ct.nwg), dim3(256), lpad, st, g);
    FH_CHECK_HIP(hipGetLastError());
    return FLOAT_OK;
  }
  const int R = g.R, n = g.F;
  const int lpp = g.C / 8, gpb = 256 / lpp;
  static const int pix_env = env_int("FLOAT_DEC_FLOW_PIX", 0, 0, 4);  // tuning aid
  const int pix = (pix_env == 1 || pix_env == 2 || pix_env == 4) ? pix_env : (lpp <= 8 ? 4 : 2);
  const int step = gpb * pix;  // pixels one workgroup covers per iteration
  const int max_bx = (R * R + step - 1) / step;
  static const int wg_env = env_int("FLOAT_DEC_FLOW_WGS", 2048, 8, 1 << 20);
  int bx = std::max(1, std::min(max_bx, (wg_env + n - 1) / n));
  if (bx >= 8) bx &= ~7;  // bands in multiples of 8: band <-> XCD affinity (dec_flow_kernel)
  g.band_pix = ((R * R + bx - 1) / bx + step - 1) / step * step;
  g.nbands = bx = (R * R + g.band_pix - 1) / g.band_pix;
  if (h) g.ct = take_ride(h, R, 2);
  // dynamic LDS only as an occupancy cap (FLOAT_DEC_LDS_PAD); the kernel's own 14 KB are static
  const size_t pad = kLdsPad > 14 * 1024 ? (size_t)kLdsPad - 14 * 1024 : 0;
  const dim3 grid(bx * n + g.ct.nwg);
#define FLOW_LAUNCH(PIXv)                                                                                  \
  if (g.xnext) hipLaunchKernelGGL((dec_flow_kernel<T, PIXv, false>), grid, dim3(256), pad, st, g);         \
  else hipLaunchKernelGGL((dec_flow_kernel<T, PIXv, true>), grid, dim3(256), pad, st, g);
  if (pix == 4) { FLOW_LAUNCH(4) }
  else if (pix == 2) { FLOW_LAUNCH(2) }
  else { FLOW_LAUNCH(1) }
#undef FLOW_LAUNCH
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

// One resolution level for `n` frames: x_in (R/2, scaled by the up-conv's style) -> z -> U -> V ->
// flow/warp/blend/rgb.  U and the next level's input may alias (U is dead once conv2 has run).
template <class T>
int run_level(float_dec* h, int li, int n, const void* x_in, void* Zb, void* U, void* V, void* xnext, const float* styles,
              const float* demod, const float* flow_prev, const float* rgb_prev, float* flow_cur, float* rgb_cur,
              float* final_out, int final_mode, hipStream_t st) {
  const Level& L = h->levels[li];
  const Styled& up = h->convs[1 + 2 * li];
  const Styled& c2 = h->convs[2 + 2 * li];
  const int R = L.R;
  int rc;
  if ((rc = launch_upconv<T>(h, up, R / 2, n, x_in, Zb, U, &U, demod + up.demod_off, h->Dtot, styles + c2.style_off, h->Stot,
                             h->sat + 1 + 2 * li, st)))
    return rc;
  // conv2 (plain 3x3); its unscaled output feeds ToFlow.  (Running the flow phase in conv2's epilogue at C <= 64 - the conv2
  // tile through LDS instead of memory, -33.6 MB per frame at 512 px - was built in round 2, was bitwise equal and slower:
  // 30.5-30.9 vs 26.2 ms per 250 frames; removed in round 3, see DESIGN.md "Negative results".)
  const bool last = (li == h->n_levels - 1);
  // last level, <= 64 channels: ToFlow's conv rides in conv2's epilogue (V is not stored), dec_flowlast_kernel finishes the frame
  const bool epi = last && h->oflow && R % 16 == 0 && R >= 16 && (c2.cout == 32 || c2.cout == 64) && L.C == c2.cout;
  if (epi) {
    typedef typename T::pack8 P8;
    hipLaunchKernelGGL((dec_flowfrag_kernel<T>), dim3(n, c2.cout / 32), dim3(64), 0, st, reinterpret_cast<P8*>(h->wfrag), L.wflow,
                       styles + L.style_off, h->Stot, L.C);
  }
  if ((rc = launch_conv<T>(h, U, R, R, c2, c2.W, 9, kDy9, kDx9, V, R, R, R, R, 1, 1, 0, 0, n, demod + c2.demod_off, h->Dtot,
                           c2.abias, 1, nullptr, 0, h->sat + 2 + 2 * li, st, epi ? h->wfrag : nullptr, epi ? h->oflow : nullptr)))
    return rc;
  FlowArgs g;
  memset(&g, 0, sizeof(g));
  g.x = V;
  g.feat = L.feat;
  g.pflow = flow_prev;
  g.prgb = rgb_prev;
  memcpy(g.upk_flow, L.upk_flow, sizeof(g.upk_flow));
  memcpy(g.upk_rgb, L.upk_rgb, sizeof(g.upk_rgb));
  g.wflow = L.wflow;
  g.sflow = styles + L.style_off;
  g.bflow = L.bflow;
  g.wrgb = L.wrgb;
  g.grgb = L.grgb;
  g.b1 = L.b1;
  g.b2 = L.b2;
  g.lin = L.lin;
  g.snext = last ? nullptr : styles + h->convs[1 + 2 * (li + 1)].style_off;
  g.xnext = last ? nullptr : xnext;
  g.flow_out = flow_cur;
  g.rgb_out = rgb_cur;
  g.final_out = last ? final_out : nullptr;
  g.final_mode = last ? final_mode : 0;
  g.write_pyr = (!last || getenv("FLOAT_DEC_WRITE_PYR")) ? 1 : 0;
  g.F = n;
  g.R = R;
  g.C = L.C;
  g.ld_s = h->Stot;
  g.sat = h->sat + 16 + li;
  g.oflow = epi ? h->oflow : nullptr;
#ifdef DEC_STAMPS
  if (last) {
    const unsigned long long init[4] = {~0ull, 0ull, ~0ull, 0ull};
    FH_CHECK_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_dec_stamps), init, sizeof(init), 0, hipMemcpyHostToDevice, st));
  }
#endif
  return launch_flow<T>(h, g, st);
}

// Low phase for `n` frames (n <= lo_frames): constant input, conv1, levels 8..32.  Leaves the
// next level's (scaled) input in loX and the flow / rgb pyramids in loFlow[k] / loRgb[k]; returns k.
template <class T>
int run_low(float_dec* h, int n, const float* styles, const float* demod, int* skip_idx, hipStream_t st) {
  int rc;
  const Styled& c1 = h->convs[0];
  const int tot = n * 16 * c1.cin / 4;
  hipLaunchKernelGGL((dec_input_kernel<T>), dim3((tot + 255) / 256), dim3(256), 0, st, reinterpret_cast<typename T::elem*>(h->loB),
                     h->cin_hwc, styles + c1.style_off, h->Stot, n, 16, c1.cin, h->sat + 32);
  void* first_out = h->lo_levels > 0 ? h->loA : h->loX;
  if ((rc = launch_conv<T>(nullptr, h->loB, 4, 4, c1, c1.W, 9, kDy9, kDx9, first_out, 4, 4, 4, 4, 1, 1, 0, 0, n, demod + c1.demod_off,
                           h->Dtot, c1.abias, 1, styles + h->convs[1].style_off, h->Stot, h->sat + 0, st)))
    return rc;
  int cur = 0;
  const float *fp = nullptr, *rp = nullptr;
  for (int li = 0; li < h->lo_levels; ++li) {
    void* xnext = (li == h->lo_levels - 1) ? h->loX : h->loA;
    if ((rc = run_level<T>(h, li, n, h->loA, h->loZ, h->loA, h->loB, xnext, styles, demod, fp, rp, h->loFlow[cur], h->loRgb[cur],
                           nullptr, 0, st)))
      return rc;
    fp = h->loFlow[cur];
    rp = h->loRgb[cur];
    cur ^= 1;
  }
  *skip_idx = cur ^ 1;  // buffers holding the last written pyramids (unused when lo_levels == 0)
  return FLOAT_OK;
}

// High phase for `n` frames (n <= max_frames) that sit at frame offset `off` inside the low batch.
template <class T>
int run_high(float_dec* h, int n, int off, const float* styles, const float* demod, int skip_idx, float* out, int final_mode,
             hipStream_t st) {
  int rc;
  const int l0 = h->lo_levels;
  typedef typename T::elem E;
  const void* x_in;
  const float *fp = nullptr, *rp = nullptr;
  if (l0 > 0) {
    const Level& P = h->levels[l0 - 1];
    x_in = reinterpret_cast<const E*>(h->loX) + (size_t)off * P.R * P.R * P.C;
    fp = h->loFlow[skip_idx] + (size_t)off * P.R * P.R * 4;
    rp = h->loRgb[skip_idx] + (size_t)off * P.R * P.R * 4;
  } else {
    x_in = reinterpret_cast<const E*>(h->loX) + (size_t)off * 16 * h->convs[0].cout;
  }
  int cur = 0;
  for (int li = l0; li < h->n_levels; ++li) {
    if ((rc = run_level<T>(h, li, n, x_in, h->hiZ, h->hiA, h->hiB, h->hiA, styles, demod, fp, rp, h->hiFlow[cur], h->hiRgb[cur],
                           out, final_mode, st)))
      return rc;
    x_in = h->hiA;
    fp = h->hiFlow[cur];
    rp = h->hiRgb[cur];
    cur ^= 1;
  }
  return FLOAT_OK;
}

// host != nullptr: every finished high batch is copied to host + (its offset) on `cs` while `st` renders the next one
// (float_dec_frames_host); `st` is made to wait for the last copy before the call returns.
template <class T>
int frames_impl(float_dec* h, const float* s_r, const float* r_d, int n_frames, float* out, int final_mode, hipStream_t st,
                float* host = nullptr, hipStream_t cs = nullptr, float* host_dev = nullptr) {
  const int S = h->cfg.size, sdim = h->cfg.style_dim, FH = h->cfg.max_frames, FL = h->lo_frames;
  size_t n_copy = 0;
  // same-stream hand-over: copy workgroups ride along the next batch's launches unless FLOAT_DEC_COPY=memcpy
  static const bool ride_on = !(getenv("FLOAT_DEC_COPY") && !strcmp(getenv("FLOAT_DEC_COPY"), "memcpy"));
  // the copy workgroups store straight through `host`: only when it is device-accessible (pinned / registered) host memory
  // (host_dev = its device-side address); a pageable destination takes the hipMemcpyAsync path, batch by batch, in order
  const bool ride = host_dev && cs == st && ride_on && (((size_t)S * S * 3 * sizeof(float)) % 16 == 0) &&
                    ((uintptr_t)host_dev % 16 == 0) && ((uintptr_t)out % 16 == 0);
  h->ride.left16 = 0;
  h->ride.wleft = 0.0;
  // Ragged clips put their SHORT piece first at every level (style chunk, low group, high batch): the last batch of the call is
  // then a full one.  A short last batch carried the previous full batch's copy in launches too short for it (its launches took
  // as long as a full batch's) and the short first batch carries nothing.  FLOAT_DEC_SHORT_FIRST=0: remainder last.
  static const bool short_first = env_int("FLOAT_DEC_SHORT_FIRST", 1, 0, 1) != 0;
  auto piece = [](int done, int total, int cap) {
    const int r = total % cap;
    return (short_first && done == 0 && r) ? r : std::min(cap, total - done);
  };
  for (int s0 = 0, ns = 0; s0 < n_frames; s0 += ns) {
    ns = piece(s0, n_frames, kStyleCap);
    // every style modulation (22 EqualLinears) and every demod factor for `ns` frames: 2 launches
    {
      constexpr int FB = 8;
      dim3 grid((h->Stot + 255) / 256, (ns + FB - 1) / FB);
      hipLaunchKernelGGL((dec_small_gemm_kernel<SG_STYLE, FB>), grid, dim3(256), FB * sdim * sizeof(float), st,
                         r_d + (size_t)s0 * sdim, sdim, s_r, h->WmT, sdim, h->Stot, h->bm, 1.0f / sqrtf((float)sdim), h->styles,
                         h->Stot, ns);
      DemodArgs d;
      memset(&d, 0, sizeof(d));
      int maxc = 0, maxcin = 0;
      FH_REQUIRE(h->convs.size() <= 16, "too many styled convs");
      for (size_t i = 0; i < h->convs.size(); ++i) {
        d.L[i] = {h->convs[i].WsqT, h->convs[i].cin, h->convs[i].cout, h->convs[i].style_off, h->convs[i].demod_off};
        maxc = std::max(maxc, h->convs[i].cout);
        maxcin = std::max(maxcin, h->convs[i].cin);
      }
      d.styles = h->styles;
      d.eps = h->eps;
      d.demod = h->demod;
      d.ld_s = h->Stot;
      d.ld_d = h->Dtot;
      d.F = ns;
      d.normalise = h->style_norm ? 1 : 0;
      d.sat = h->sat;
      // every StyledConv's style divided by its max |s| per frame, eps / max^2 left for the demodulation (dec_kernels.hpp)
      hipLaunchKernelGGL(dec_style_norm_kernel, dim3((unsigned)h->convs.size(), ns), dim3(256), 0, st, d);
      dim3 g2((maxc + 255) / 256, (ns + FB - 1) / FB, (unsigned)h->convs.size());
      hipLaunchKernelGGL((dec_demod_all_kernel<FB>), g2, dim3(256), FB * maxcin * sizeof(float), st, d);
    }
    for (int a0 = 0, na = 0; a0 < ns; a0 += na) {
      na = piece(a0, ns, FL);
      const float* st_a = h->styles + (size_t)a0 * h->Stot;
      const float* dm_a = h->demod + (size_t)a0 * h->Dtot;
      int skip_idx = 0;
      int rc = run_low<T>(h, na, st_a, dm_a, &skip_idx, st);
      if (rc) return rc;
      for (int b0 = 0, nb = 0; b0 < na; b0 += nb) {
        nb = piece(b0, na, FH);
        // ride-along hand-over: the very last batch of the call has no successor to carry its copy.  Cutting it in two so that
        // only FLOAT_DEC_RIDE_TAIL frames' copy stays exposed was measured and is off: 30.1 ms per 250 frames without, 31.0-31.8
        // with a tail of 4..16 frames (the smaller launches lose more than the shorter copy gains)
        static const int tail = getenv("FLOAT_DEC_RIDE_TAIL") ? atoi(getenv("FLOAT_DEC_RIDE_TAIL")) : 0;
        const bool last_of_call = (s0 + a0 + b0 + nb == n_frames);
        if (host && ride && last_of_call && tail > 0 && nb > tail) nb -= tail;
        const size_t off = (size_t)(s0 + a0 + b0) * S * S * 3;
        if (host && ride) {
          // launches of this batch that carry a share: up-conv, conv2 and flow kernel of every level from kRideMinRes up
          double wsum = 0.0;
          for (int li = h->lo_levels; li < h->n_levels; ++li)
            if (h->levels[li].R >= kRideMinRes)
              for (int kind = 0; kind < 3; ++kind) wsum += ride_weight(h->levels[li].R, kind);
          h->ride.wleft = h->ride.left16 ? wsum : 0.0;
        }
        rc = run_high<T>(h, nb, b0, st_a + (size_t)b0 * h->Stot, dm_a + (size_t)b0 * h->Dtot, skip_idx, out + off, final_mode, st);
        if (rc) return rc;
        const size_t bytes = (size_t)nb * S * S * 3 * sizeof(float);
        if (host && ride) {
          if (h->ride.left16)  // what no launch took (a decoder without carrying levels): plain copy, in order
            FH_CHECK_HIP(hipMemcpyAsync(h->ride.dst_host, h->ride.src, h->ride.left16 * 16, hipMemcpyDeviceToHost, st));
          h->ride.src = out + off;  // this batch crosses PCIe under the next one's kernels
          h->ride.dst = host_dev + off;
          h->ride.dst_host = host + off;
          h->ride.left16 = bytes / 16;
        } else if (host && cs == st) {  // in-order copy behind the batch's last kernel
          FH_CHECK_HIP(hipMemcpyAsync(host + off, out + off, bytes, hipMemcpyDeviceToHost, st));
        } else if (host) {
          if (n_copy >= h->copy_events.size()) {
            hipEvent_t e;
            FH_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            h->copy_events.push_back(e);
          }
          hipEvent_t e = h->copy_events[n_copy++];
          FH_CHECK_HIP(hipEventRecord(e, st));
          FH_CHECK_HIP(hipStreamWaitEvent(cs, e, 0));
          FH_CHECK_HIP(hipMemcpyAsync(host + off, out + off, bytes, hipMemcpyDeviceToHost, cs));
        }
      }
    }
  }
  if (host && ride && h->ride.left16) {  // the last batch has no successor to ride along
    FH_CHECK_HIP(hipMemcpyAsync(h->ride.dst_host, h->ride.src, h->ride.left16 * 16, hipMemcpyDeviceToHost, st));
    h->ride.left16 = 0;
  }
  if (host && n_copy) {  // join: work queued on `st` after this call sees the frames in host memory
    if (!h->join_event) FH_CHECK_HIP(hipEventCreateWithFlags(&h->join_event, hipEventDisableTiming));
    FH_CHECK_HIP(hipEventRecord(h->join_event, cs));
    FH_CHECK_HIP(hipStreamWaitEvent(st, h->join_event, 0));
  }
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

// G = ToRGB(features) of every level, after the features changed
template <class T>
int feats_rgb_impl(float_dec* h, hipStream_t st) {
  for (int li = 0; li < h->n_levels; ++li) {
    const Level& L = h->levels[li];
    const int npix = L.R * L.R;
    hipLaunchKernelGGL((dec_feat_rgb_kernel<T>), dim3((npix + 255) / 256), dim3(256), 0, st, L.grgb,
                       reinterpret_cast<const typename T::elem*>(L.feat), L.wrgb, L.C, npix);
  }
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

template <class T>
int set_feats_impl(float_dec* h, const float* const* feats, hipStream_t st) {
  for (int li = 0; li < h->n_levels; ++li) {
    const Level& L = h->levels[li];
    const int tot = L.C * L.R * L.R / 4;
    hipLaunchKernelGGL((dec_feat_pack_kernel<T>), dim3((tot + 255) / 256), dim3(256), 0, st, reinterpret_cast<typename T::elem*>(L.feat),
                       feats[li], L.C, L.R * L.R, h->sat + 33);
  }
  FH_CHECK_HIP(hipGetLastError());
  return feats_rgb_impl<T>(h, st);
}

// Run `call` with T = the handle's operand type.
#define DEC_DISPATCH(dtype, call)                          \
  ((dtype) == FLOAT_DT_FP32 ? ([&] { typedef FP32 T; return call; })() : ([&] { typedef FP16 T; return call; })())

// One StyledConv / one ToFlow + ToRGB level on caller data (float_dec_debug_*): a private pool, the production launchers.
struct UnitCtx {
  DevicePool pool;
  ~UnitCtx() { pool.release(); }
};

template <class T>
int unit_styles(UnitCtx* u, const float_tensor_t* mw, const float_tensor_t* mb, int cin, int sdim, const float* style, int F,
                float** styles_out, hipStream_t st) {
  // s = EqualLinear(style): style @ (W / sqrt(sdim))^T + b (styledecoder.py:229,241)
  std::vector<float> wmT((size_t)sdim * cin);
  for (int j = 0; j < cin; ++j)
    for (int k = 0; k < sdim; ++k) wmT[(size_t)k * cin + j] = mw->data[(size_t)j * sdim + k];
  float *WmT = nullptr, *bm = nullptr, *styles = nullptr;
  int rc;
  if ((rc = upload32(&u->pool, wmT, &WmT))) return rc;
  if ((rc = upload32(&u->pool, std::vector<float>(mb->data, mb->data + cin), &bm))) return rc;
  if ((rc = u->pool.alloc(&styles, (size_t)F * cin, true))) return rc;
  constexpr int FB = 8;
  dim3 grid((cin + 255) / 256, (F + FB - 1) / FB);
  hipLaunchKernelGGL((dec_small_gemm_kernel<SG_STYLE, FB>), grid, dim3(256), FB * sdim * sizeof(float), st, style, sdim,
                     (const float*)nullptr, WmT, sdim, cin, bm, 1.0f / sqrtf((float)sdim), styles, cin, F);
  *styles_out = styles;
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

template <class T>
int unit_styled_conv(const float_dec_unit_t* cfg, const TensorTable& tt, const float* x, const float* style, float* out,
                     uint64_t* saturated, int style_norm, hipStream_t st) {
  typedef typename T::elem E;
  const int cin = cfg->cin, cout = cfg->cout, Ri = cfg->res, F = cfg->n_frames, sdim = cfg->style_dim;
  const int Ro = cfg->upsample ? 2 * Ri : Ri;
  UnitCtx u;
  int rc;
  if ((rc = raise_lds_limits<T>())) return rc;
  Styled s;
  std::vector<float> wm_rows, bm_host;
  if ((rc = pack_styled<T>(&u.pool, tt, "sc", cin, cout, cfg->upsample != 0, &s, &wm_rows, &bm_host, sdim))) return rc;
  float* styles = nullptr;
  if ((rc = unit_styles<T>(&u, tt.find("sc.conv.modulation.weight"), tt.find("sc.conv.modulation.bias"), cin, sdim, style, F, &styles, st)))
    return rc;
  float *demod = nullptr, *eps = nullptr, *ones = nullptr;
  unsigned long long* sat = nullptr;
  if ((rc = u.pool.alloc(&demod, (size_t)F * cout, true))) return rc;
  if ((rc = u.pool.alloc(&eps, (size_t)F * 16, true))) return rc;
  if ((rc = u.pool.alloc(&sat, 1, true))) return rc;
  if ((rc = upload32(&u.pool, std::vector<float>((size_t)F * cout, 1.0f), &ones))) return rc;
  DemodArgs d;
  memset(&d, 0, sizeof(d));
  d.L[0] = {s.WsqT, cin, cout, 0, 0};
  d.styles = styles;
  d.eps = eps;
  d.demod = demod;
  d.ld_s = cin;
  d.ld_d = cout;
  d.F = F;
  d.normalise = style_norm;
  d.sat = nullptr;
  constexpr int FB = 8;
  hipLaunchKernelGGL(dec_style_norm_kernel, dim3(1, F), dim3(256), 0, st, d);
  hipLaunchKernelGGL((dec_demod_all_kernel<FB>), dim3((cout + 255) / 256, (F + FB - 1) / FB, 1), dim3(256), FB * cin * sizeof(float), st, d);
  E *X = nullptr, *Z = nullptr, *U = nullptr;
  const size_t nin = (size_t)F * Ri * Ri * cin, nout = (size_t)F * (Ro + 1) * (Ro + 1) * cout;
  if ((rc = u.pool.alloc(&X, nin, true))) return rc;
  if ((rc = u.pool.alloc(&Z, nout, true))) return rc;
  if ((rc = u.pool.alloc(&U, nout, true))) return rc;
  hipLaunchKernelGGL((dec_dbg_pack_kernel<T>), dim3((unsigned)((nin / 4 + 255) / 256)), dim3(256), 0, st, X, x, styles, cin, F, cin,
                     Ri * Ri, sat);
  void* Y = U;
  if (cfg->upsample) {
    if ((rc = launch_upconv<T>(nullptr, s, Ri, F, X, Z, U, &Y, demod, cout, ones, cout, sat, st))) return rc;
  } else {
    if ((rc = launch_conv<T>(nullptr, X, Ri, Ri, s, s.W, 9, kDy9, kDx9, U, Ri, Ri, Ri, Ri, 1, 1, 0, 0, F, demod, cout, s.abias, 1, nullptr,
                             0, sat, st)))
      return rc;
  }
  const size_t no = (size_t)F * Ro * Ro * cout;
  hipLaunchKernelGGL((dec_dbg_unpack_kernel<T>), dim3((unsigned)((no + 255) / 256)), dim3(256), 0, st, out, reinterpret_cast<const E*>(Y), F,
                     cout, Ro * Ro);
  FH_CHECK_HIP(hipGetLastError());
  FH_CHECK_HIP(hipStreamSynchronize(st));
  unsigned long long n = 0;
  FH_CHECK_HIP(hipMemcpy(&n, sat, sizeof(n), hipMemcpyDeviceToHost));
  if (saturated) *saturated = n;
  return FLOAT_OK;
}

template <class T>
int unit_flow_level(const float_dec_unit_t* cfg, const TensorTable& tt, const float* x, const float* feat, const float* style,
                    const float* prev_flow, const float* prev_rgb, float* out_flow, float* out_blend, float* out_rgb, hipStream_t st) {
  typedef typename T::elem E;
  const int C = cfg->cin, R = cfg->res, F = cfg->n_frames, sdim = cfg->style_dim, Rp = R / 2;
  UnitCtx u;
  int rc;
  const float_tensor_t* fw = need(tt, "to_flow.conv.weight", 3 * C);
  const float_tensor_t* fmw = need(tt, "to_flow.conv.modulation.weight", (int64_t)C * sdim);
  const float_tensor_t* fmb = need(tt, "to_flow.conv.modulation.bias", C);
  const float_tensor_t* fb = need(tt, "to_flow.bias", 3);
  const float_tensor_t* rw = need(tt, "to_rgb.conv.0.weight", 3 * C);
  const float_tensor_t* rb1 = need(tt, "to_rgb.conv.1.bias", 3);
  const float_tensor_t* rb2 = need(tt, "to_rgb.bias", 3);
  if (!fw || !fmw || !fmb || !fb || !rw || !rb1 || !rb2) return FLOAT_E_MISSING;
  const float sc = 1.0f / sqrtf((float)C);
  std::vector<float> a(3 * C), b(3 * C), lin(R);
  for (int i = 0; i < 3 * C; ++i) {
    a[i] = fw->data[i] * sc;
    b[i] = rw->data[i] * sc;
  }
  const double step = 2.0 / (double)(R - 1);
  for (int i = 0; i < R; ++i) lin[i] = (float)(-1.0 + (double)i * step);
  lin[R - 1] = 1.0f;
  FlowArgs g;
  memset(&g, 0, sizeof(g));
  float *wflow, *wrgb, *bflow, *b1, *b2, *dlin, *styles = nullptr, *ones, *pf = nullptr, *pr = nullptr, *fo, *ro;
  if ((rc = upload32(&u.pool, a, &wflow)) || (rc = upload32(&u.pool, b, &wrgb))) return rc;
  if ((rc = upload32(&u.pool, std::vector<float>(fb->data, fb->data + 3), &bflow))) return rc;
  if ((rc = upload32(&u.pool, std::vector<float>(rb1->data, rb1->data + 3), &b1))) return rc;
  if ((rc = upload32(&u.pool, std::vector<float>(rb2->data, rb2->data + 3), &b2))) return rc;
  if ((rc = upload32(&u.pool, lin, &dlin))) return rc;
  if ((rc = upload32(&u.pool, std::vector<float>((size_t)F * C, 1.0f), &ones))) return rc;
  if ((rc = unit_styles<T>(&u, fmw, fmb, C, sdim, style, F, &styles, st))) return rc;
  E *X, *Ft, *XN;
  unsigned long long* sat;
  if ((rc = u.pool.alloc(&sat, 1, true))) return rc;
  if ((rc = u.pool.alloc(&X, (size_t)F * R * R * C, true)) || (rc = u.pool.alloc(&XN, (size_t)F * R * R * C, true))) return rc;
  if ((rc = u.pool.alloc(&Ft, (size_t)R * R * C, true))) return rc;
  if ((rc = u.pool.alloc(&fo, (size_t)F * R * R * 4, true)) || (rc = u.pool.alloc(&ro, (size_t)F * R * R * 4, true))) return rc;
  const size_t nx = (size_t)F * R * R * C;
  hipLaunchKernelGGL((dec_dbg_pack_kernel<T>), dim3((unsigned)((nx / 4 + 255) / 256)), dim3(256), 0, st, X, x, (const float*)nullptr, 0, F,
                     C, R * R, sat);
  hipLaunchKernelGGL((dec_dbg_pack_kernel<T>), dim3((unsigned)(((size_t)R * R * C / 4 + 255) / 256)), dim3(256), 0, st, Ft, feat,
                     (const float*)nullptr, 0, 1, C, R * R, sat);
  if (prev_flow) {
    if ((rc = u.pool.alloc(&pf, (size_t)F * Rp * Rp * 4, true))) return rc;
    hipLaunchKernelGGL(dec_dbg_pyr_kernel, dim3((F * Rp * Rp + 255) / 256), dim3(256), 0, st, pf, prev_flow, F, Rp * Rp, 0);
  }
  if (prev_rgb) {
    if ((rc = u.pool.alloc(&pr, (size_t)F * Rp * Rp * 4, true))) return rc;
    hipLaunchKernelGGL(dec_dbg_pyr_kernel, dim3((F * Rp * Rp + 255) / 256), dim3(256), 0, st, pr, prev_rgb, F, Rp * Rp, 0);
  }
  float* G;
  if ((rc = u.pool.alloc(&G, (size_t)R * R * 4, true))) return rc;
  hipLaunchKernelGGL((dec_feat_rgb_kernel<T>), dim3((R * R + 255) / 256), dim3(256), 0, st, G, Ft, wrgb, C, R * R);
  g.grgb = G;
  g.x = X;
  g.feat = Ft;
  g.pflow = pf;
  g.prgb = pr;
  if ((rc = upsample_taps(tt, "to_flow.upsample.kernel", g.upk_flow)) || (rc = upsample_taps(tt, "to_rgb.upsample.kernel", g.upk_rgb))) return rc;
  g.wflow = wflow;
  g.sflow = styles;
  g.bflow = bflow;
  g.wrgb = wrgb;
  g.b1 = b1;
  g.b2 = b2;
  g.lin = dlin;
  g.snext = ones;
  g.xnext = XN;
  g.flow_out = fo;
  g.rgb_out = ro;
  g.write_pyr = 1;
  g.F = F;
  g.R = R;
  g.C = C;
  g.ld_s = C;
  g.sat = sat;
  if ((rc = launch_flow<T>(nullptr, g, st))) return rc;
  if (out_flow) hipLaunchKernelGGL(dec_dbg_pyr_kernel, dim3((F * R * R + 255) / 256), dim3(256), 0, st, out_flow, fo, F, R * R, 1);
  if (out_rgb) hipLaunchKernelGGL(dec_dbg_pyr_kernel, dim3((F * R * R + 255) / 256), dim3(256), 0, st, out_rgb, ro, F, R * R, 1);
  if (out_blend)
    hipLaunchKernelGGL((dec_dbg_unpack_kernel<T>), dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, st, out_blend, XN, F, C, R * R);
  FH_CHECK_HIP(hipGetLastError());
  FH_CHECK_HIP(hipStreamSynchronize(st));
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
  // fp16 operands in production: with bf16 the 512-px frames sat at the 40 dB limit (40.7 dB, max |d| 0.15 on a [0,1] pixel: the
  // flow field positions the bilinear sampling, 8 mantissa bits are too few there); fp16 runs at the same MFMA rate.
  // FLOAT_DT_FP32 is the verification mode: the same launch chain with fp32 activations and weights (1/16 of the MFMA rate).
  FH_REQUIRE(cfg->dtype == FLOAT_DT_FP16 || cfg->dtype == FLOAT_DT_FP32,
             "the decoder supports FLOAT_DT_FP16 operands (and FLOAT_DT_FP32 for verification) only (got dtype %d)", cfg->dtype);
  float_dec* h = new float_dec();
  h->cfg = *cfg;
  h->style_norm = env_int("FLOAT_DEC_STYLE_NORM", 1, 0, 1) != 0;
  TensorTable tt(tensors, n_tensors);
  int rc = DEC_DISPATCH(cfg->dtype, create_impl<T>(h, tt));
  if (!rc) {
    if (const float_tensor_t* dw = tt.find("direction.weight")) {
      if (dw->ndim == 2 && dw->shape[0] == cfg->style_dim) {
        std::vector<float> Q;
        h->motion_dim = (int)dw->shape[1];
        fh_direction_q(dw->data, cfg->style_dim, h->motion_dim, &Q);
        rc = h->pool.alloc(&h->dirQ, Q.size(), false);
        if (!rc && hipMemcpy(h->dirQ, Q.data(), Q.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = FLOAT_E_HIP;
      }
    }
  }
  if (rc) {
    float_dec_destroy(h);
    return rc;
  }
  *out = h;
  return FLOAT_OK;
}

int float_dec_direction(float_dec_t* h, const float* lam, float* r_s, void* stream) {
  FH_REQUIRE(h && lam && r_s, "null argument to float_dec_direction");
  FH_REQUIRE(h->dirQ != nullptr, "the decoder checkpoint has no (style_dim, motion_dim) 'direction.weight'");
  return fh_linear_f32(lam, h->dirQ, nullptr, 1.0f, r_s, h->cfg.style_dim, h->motion_dim, (hipStream_t)stream);
}

void float_dec_destroy(float_dec_t* h) {
  if (!h) return;
  for (hipEvent_t e : h->copy_events) (void)hipEventDestroy(e);
  if (h->join_event) (void)hipEventDestroy(h->join_event);
  h->pool.release();
  delete h;
}

int float_dec_set_feats(float_dec_t* h, const float* const* feats, int32_t n_feats, void* stream) {
  FH_REQUIRE(h && feats, "null argument to float_dec_set_feats");
  FH_REQUIRE(n_feats == h->n_levels, "expected %d feature maps (8..%d), got %d", h->n_levels, h->cfg.size, n_feats);
  for (int i = 0; i < n_feats; ++i) FH_REQUIRE(feats[i] != nullptr, "feats[%d] is null", i);
  hipStream_t st = (hipStream_t)stream;
  int rc = DEC_DISPATCH(h->cfg.dtype, set_feats_impl<T>(h, feats, st));
  if (!rc) h->feats_set = true;
  return rc;
}

int float_dec_set_feats16(float_dec_t* h, const void* const* feats16, int32_t n_feats, int32_t dtype, void* stream) {
  FH_REQUIRE(h && feats16, "null argument to float_dec_set_feats16");
  FH_REQUIRE(n_feats == h->n_levels, "expected %d feature maps (8..%d), got %d", h->n_levels, h->cfg.size, n_feats);
  FH_REQUIRE(dtype == h->cfg.dtype, "feature dtype %d differs from the decoder's (%d)", dtype, h->cfg.dtype);
  for (int i = 0; i < n_feats; ++i) FH_REQUIRE(feats16[i] != nullptr, "feats16[%d] is null", i);
  hipStream_t st = (hipStream_t)stream;
  const size_t eb = dtype == FLOAT_DT_FP32 ? 4 : 2;
  for (int li = 0; li < h->n_levels; ++li) {
    const Level& L = h->levels[li];
    int rc = fh_copy_d2d(L.feat, feats16[li], (size_t)L.R * L.R * L.C * eb, st);
    if (rc) return rc;
  }
  int rc = DEC_DISPATCH(h->cfg.dtype, feats_rgb_impl<T>(h, st));
  if (rc) return rc;
  h->feats_set = true;
  return FLOAT_OK;
}

static int dec_run(float_dec_t* h, const float* s_r, const float* r_d, int32_t n_frames, float* out, int mode, void* stream) {
  FH_REQUIRE(h && s_r && r_d && out, "null argument to float_dec_frames");
  FH_REQUIRE(h->feats_set, "float_dec_set_feats must be called before decoding");
  FH_REQUIRE(n_frames >= 1, "n_frames must be >= 1 (got %d)", n_frames);
  hipStream_t st = (hipStream_t)stream;
  return DEC_DISPATCH(h->cfg.dtype, frames_impl<T>(h, s_r, r_d, n_frames, out, mode, st));
}

int float_dec_frames(float_dec_t* h, const float* s_r, const float* r_d, int32_t n_frames, float* out_hwc, void* stream) {
  return dec_run(h, s_r, r_d, n_frames, out_hwc, 1, stream);
}

int float_dec_frames_host(float_dec_t* h, const float* s_r, const float* r_d, int32_t n_frames, float* out_hwc, float* host_hwc,
                          void* stream, void* copy_stream) {
  FH_REQUIRE(h && s_r && r_d && out_hwc && host_hwc, "null argument to float_dec_frames_host");
  FH_REQUIRE(h->feats_set, "float_dec_set_feats must be called before decoding");
  FH_REQUIRE(n_frames >= 1, "n_frames must be >= 1 (got %d)", n_frames);
  hipStream_t st = (hipStream_t)stream, cs = copy_stream ? (hipStream_t)copy_stream : st;
  // Is host_hwc memory a kernel may store through?  Only pinned (hipHostMalloc) or registered (hipHostRegister) host memory
  // has a device-side address; for anything else - pageable memory - the frames go by hipMemcpyAsync behind each batch.
  float* host_dev = nullptr;
  {
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    if (hipPointerGetAttributes(&at, host_hwc) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer)
      host_dev = reinterpret_cast<float*>(at.devicePointer);
    else
      (void)hipGetLastError();  // "invalid value" for pageable memory: not an error of this call
  }
  return DEC_DISPATCH(h->cfg.dtype, frames_impl<T>(h, s_r, r_d, n_frames, out_hwc, 1, st, host_hwc, cs, host_dev));
}

int float_dec_saturation(float_dec_t* h, uint64_t* total, uint64_t* per_site, int32_t reset, void* stream) {
  FH_REQUIRE(h && total, "null argument to float_dec_saturation");
  hipStream_t st = (hipStream_t)stream;
  unsigned long long host[kDecSatSites];
  FH_CHECK_HIP(hipStreamSynchronize(st));
  FH_CHECK_HIP(hipMemcpy(host, h->sat, sizeof(host), hipMemcpyDeviceToHost));
  uint64_t sum = 0;
  for (int i = 0; i < kDecSatSites; ++i) {
    sum += host[i];
    if (per_site) per_site[i] = host[i];
  }
  *total = sum;
  if (reset) {  // on the caller's stream: ordered against the launches that add to the counters there
    FH_CHECK_HIP(hipMemsetAsync(h->sat, 0, sizeof(host), st));
    FH_CHECK_HIP(hipStreamSynchronize(st));
  }
  return FLOAT_OK;
}

int float_dec_debug_styled_conv(const float_dec_unit_t* u, const float_tensor_t* tensors, int32_t n_tensors, const float* x,
                                const float* style, float* out, uint64_t* saturated, void* stream) {
  FH_REQUIRE(u && tensors && x && style && out, "null argument to float_dec_debug_styled_conv");
  FH_REQUIRE(u->dtype == FLOAT_DT_FP16 || u->dtype == FLOAT_DT_FP32, "unit op: dtype %d unsupported", u->dtype);
  FH_REQUIRE(u->cin % 32 == 0 && u->cout % 32 == 0 && u->cin > 0 && u->cout > 0, "unit op: channels must be multiples of 32");
  FH_REQUIRE(u->res >= 4 && (u->res & (u->res - 1)) == 0 && u->res <= 512, "unit op: resolution must be a power of two in [4, 512]");
  FH_REQUIRE(u->n_frames >= 1 && u->style_dim >= 1 && u->style_dim <= 2048, "unit op: bad batch / style_dim");
  TensorTable tt(tensors, n_tensors);
  return DEC_DISPATCH(u->dtype, unit_styled_conv<T>(u, tt, x, style, out, saturated, (u->flags & 1) ? 0 : 1, (hipStream_t)stream));
}

int float_dec_debug_flow_level(const float_dec_unit_t* u, const float_tensor_t* tensors, int32_t n_tensors, const float* x,
                               const float* feat, const float* style, const float* prev_flow, const float* prev_rgb, float* out_flow,
                               float* out_blend, float* out_rgb, void* stream) {
  FH_REQUIRE(u && tensors && x && feat && style, "null argument to float_dec_debug_flow_level");
  FH_REQUIRE(u->dtype == FLOAT_DT_FP16 || u->dtype == FLOAT_DT_FP32, "unit op: dtype %d unsupported", u->dtype);
  FH_REQUIRE(u->cin >= 32 && u->cin <= 512 && (u->cin & (u->cin - 1)) == 0, "unit op: channels must be a power of two in [32, 512]");
  FH_REQUIRE(u->res >= 8 && (u->res & (u->res - 1)) == 0 && u->res <= 512, "unit op: resolution must be a power of two in [8, 512]");
  FH_REQUIRE(u->n_frames >= 1 && u->style_dim >= 1 && u->style_dim <= 2048, "unit op: bad batch / style_dim");
  TensorTable tt(tensors, n_tensors);
  return DEC_DISPATCH(u->dtype, unit_flow_level<T>(u, tt, x, feat, style, prev_flow, prev_rgb, out_flow, out_blend, out_rgb, (hipStream_t)stream));
}

#ifdef DEC_PHASES
int float_dec_debug_phases(unsigned long long* out16, int reset) {
  static unsigned long long all[64 * 16];
  FH_CHECK_HIP(hipDeviceSynchronize());
  FH_CHECK_HIP(hipMemcpyFromSymbol(all, HIP_SYMBOL(g_dec_phase), sizeof(all)));
  for (int i = 0; i < 16; ++i) {
    out16[i] = 0;
    for (int r = 0; r < 64; ++r) out16[i] += all[r * 16 + i];
  }
  if (reset) {
    memset(all, 0, sizeof(all));
    FH_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_dec_phase), all, sizeof(all)));
  }
  return FLOAT_OK;
}
#endif

#ifdef DEC_STAMPS
int float_dec_debug_stamps(unsigned long long* out4) {
  FH_CHECK_HIP(hipDeviceSynchronize());
  FH_CHECK_HIP(hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_dec_stamps), 4 * sizeof(unsigned long long)));
  return FLOAT_OK;
}
#endif

int float_dec_feat_shape(float_dec_t* h, int32_t i, int32_t* channels, int32_t* resolution) {
  FH_REQUIRE(h && channels && resolution, "null argument to float_dec_feat_shape");
  FH_REQUIRE(i >= 0 && i < h->n_levels, "feature index %d out of range (the decoder takes %d maps)", i, h->n_levels);
  *channels = h->levels[i].C;
  *resolution = h->levels[i].R;
  return FLOAT_OK;
}

int float_dec_frames_raw(float_dec_t* h, const float* s_r, const float* r_d, int32_t n_frames, float* out_chw, void* stream) {
  return dec_run(h, s_r, r_d, n_frames, out_chw, 2, stream);
}

}  // extern "C"
