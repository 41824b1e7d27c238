// C-ABI entry points of the audio conditioning encoder (include/float_hip.h, section "audio encoder"):
// AudioEncoder.inference (reference FLOAT.py:370-375) = Wav2VecModel.forward(seq_len) with all hidden states
// (src/nodes/models/wav2vec2.py:33-98) + audio_projection (FLOAT.py:338-342).
#include <math.h>

#include "aud_kernels.hpp"
#include "fmt_gemm.hpp"

namespace {

struct LnP {
  float *g = nullptr, *b = nullptr;
};

struct ConvL {
  void* W = nullptr;  // [N][k*Cin] T::elem, K ordered (tap, channel)
  float* bias = nullptr;  // conv_bias = true (speech-emotion model)
  LnP ln;                 // feat_extract_norm = "layer": LayerNorm over channels after the conv
  int cin = 0, cout = 0, k = 0, stride = 0;
};

struct TLayer {
  FmtLin qkv, out, ff1, ff2;
  LnP ln1, ln2;
};

int round_up(int x, int m) { return (x + m - 1) / m * m; }

}  // namespace

struct float_aud {
  float_aud_cfg_t cfg;
  DevicePool pool;  // weights
  DevicePool ws;    // workspace, regrown when a longer clip arrives
  int C = 0, D = 0;
  float* w0 = nullptr;  // [C][k0] fp32
  float* b0 = nullptr;  // conv bias of layer 0 (conv_bias) or nullptr
  LnP gn;               // GroupNorm (or, feat_norm_layer, LayerNorm) affine of layer 0
  float *cls_dense_w = nullptr, *cls_dense_b = nullptr, *cls_out_w = nullptr, *cls_out_b = nullptr;  // classification head
  float *pooled = nullptr, *cls_h = nullptr, *cls_logits = nullptr;
  std::vector<ConvL> convs;  // layers 1..n-1
  LnP fp_ln;
  FmtLin fp_proj;
  void* pos_w = nullptr;  // [pairs][taps][GP][GP] T::elem
  float* pos_b = nullptr;
  int pos_gp = 0, pos_pairs = 0;
  LnP enc_ln;
  std::vector<TLayer> layers;
  FmtLin aproj;
  LnP aproj_ln;
  // workspace
  size_t cap_samples = 0;
  int cap_T = 0, Mp = 0;
  void *fa = nullptr, *fb = nullptr;  // T::elem (every activation buffer below: bytes = elements * element size of cfg.dtype)
  float *part = nullptr, *scsh = nullptr;
  void *x16 = nullptr, *hp16 = nullptr, *qkv16 = nullptr, *att16 = nullptr, *hid16 = nullptr, *stack16 = nullptr;
  float *hproj = nullptr, *pos = nullptr, *h = nullptr, *h1 = nullptr, *y = nullptr, *yproj = nullptr;
  unsigned long long* sat = nullptr;  // range counter of the 16-bit activation stores (float_aud_saturation)
};

namespace {

const float_tensor_t* need(const TensorTable& tt, const std::string& name) {
  const float_tensor_t* t = tt.find(name);
  if (!t) fh_set_error("missing checkpoint tensor '%s'", name.c_str());
  return t;
}

int upload(DevicePool* pool, const float* src, size_t n, float** out) {
  int rc = pool->alloc(out, n, false);
  if (rc) return rc;
  FH_CHECK_HIP(hipMemcpy(*out, src, n * sizeof(float), hipMemcpyHostToDevice));
  return FLOAT_OK;
}

int load_ln(float_aud* h, const TensorTable& tt, const std::string& p, int n, LnP* out) {
  const float_tensor_t* w = need(tt, p + ".weight");
  const float_tensor_t* b = need(tt, p + ".bias");
  if (!w || !b) return FLOAT_E_MISSING;
  FH_REQUIRE(TensorTable::numel(w) == n && TensorTable::numel(b) == n, "'%s' must have %d elements", p.c_str(), n);
  int rc;
  if ((rc = upload(&h->pool, w->data, n, &out->g))) return rc;
  return upload(&h->pool, b->data, n, &out->b);
}

template <class T>
int create_impl(float_aud* h, const TensorTable& tt) {
  const float_aud_cfg_t& c = h->cfg;
  const std::string W2 = "wav2vec2.";
  int rc;
  h->C = c.conv_dim[0];
  h->D = c.hidden;
  // ---- feature extractor (Wav2Vec2FeatureEncoder, feat_extract_norm = "group", conv_bias = false)
  {
    const std::string p = W2 + "feature_extractor.conv_layers.0.";
    const float_tensor_t* w = need(tt, p + "conv.weight");
    if (!w) return FLOAT_E_MISSING;
    FH_REQUIRE(w->ndim == 3 && w->shape[0] == h->C && w->shape[1] == 1 && w->shape[2] == c.conv_kernel[0],
               "conv_layers.0.conv.weight must be (%d,1,%d)", h->C, c.conv_kernel[0]);
    const float_tensor_t* cb = tt.find(p + "conv.bias");
    FH_REQUIRE((cb != nullptr) == (c.conv_bias != 0), "conv_bias=%d but '%sconv.bias' is %s", c.conv_bias, p.c_str(), cb ? "present" : "absent");
    if (cb && (rc = upload(&h->pool, cb->data, (size_t)h->C, &h->b0))) return rc;
    if ((rc = upload(&h->pool, w->data, (size_t)h->C * c.conv_kernel[0], &h->w0))) return rc;
    if ((rc = load_ln(h, tt, p + "layer_norm", h->C, &h->gn))) return rc;
  }
  for (int i = 1; i < c.n_conv; ++i) {
    const std::string p = W2 + "feature_extractor.conv_layers." + std::to_string(i) + ".";
    const float_tensor_t* w = need(tt, p + "conv.weight");
    if (!w) return FLOAT_E_MISSING;
    ConvL L;
    L.cin = c.conv_dim[i - 1];
    L.cout = c.conv_dim[i];
    L.k = c.conv_kernel[i];
    L.stride = c.conv_stride[i];
    FH_REQUIRE(w->ndim == 3 && w->shape[0] == L.cout && w->shape[1] == L.cin && w->shape[2] == L.k, "conv_layers.%d.conv.weight must be (%d,%d,%d)",
               i, L.cout, L.cin, L.k);
    const float_tensor_t* cb = tt.find(p + "conv.bias");
    FH_REQUIRE((cb != nullptr) == (c.conv_bias != 0), "conv_bias=%d but conv layer %d %s a bias", c.conv_bias, i, cb ? "has" : "lacks");
    if (cb && (rc = upload(&h->pool, cb->data, (size_t)L.cout, &L.bias))) return rc;
    FH_REQUIRE((tt.find(p + "layer_norm.weight") != nullptr) == (c.feat_norm_layer != 0),
               "feat_norm_layer=%d does not match the checkpoint (conv layer %d)", c.feat_norm_layer, i);
    if (c.feat_norm_layer && (rc = load_ln(h, tt, p + "layer_norm", L.cout, &L.ln))) return rc;
    typedef typename T::elem E;
    std::vector<E> hw((size_t)L.cout * L.k * L.cin);
    for (int n = 0; n < L.cout; ++n)
      for (int ci = 0; ci < L.cin; ++ci)
        for (int t = 0; t < L.k; ++t) hw[((size_t)n * L.k + t) * L.cin + ci] = T::host_from_float(w->data[((size_t)n * L.cin + ci) * L.k + t]);
    E* dW = nullptr;
    if ((rc = h->pool.alloc(&dW, hw.size(), false))) return rc;
    L.W = dW;
    FH_CHECK_HIP(hipMemcpy(dW, hw.data(), hw.size() * sizeof(E), hipMemcpyHostToDevice));
    h->convs.push_back(L);
  }
  // ---- feature projection (LayerNorm + Linear)
  if ((rc = load_ln(h, tt, W2 + "feature_projection.layer_norm", h->C, &h->fp_ln))) return rc;
  if ((rc = fmt_pack_linear(&h->pool, c.dtype, tt, {W2 + "feature_projection.projection"}, h->D, h->C, &h->fp_proj))) return rc;
  // ---- positional conv embedding, weight-norm folded: w = g * v / ||v||, norm over (out, in) per tap (dim = 2)
  {
    const std::string p = W2 + "encoder.pos_conv_embed.conv.";
    const int cpg = h->D / c.pos_groups, K = c.pos_k;
    const float_tensor_t* v = tt.find(p + "parametrizations.weight.original1");
    const float_tensor_t* g = tt.find(p + "parametrizations.weight.original0");
    if (!v) {
      v = tt.find(p + "weight_v");
      g = tt.find(p + "weight_g");
    }
    const float_tensor_t* plain = tt.find(p + "weight");
    const float_tensor_t* src = v ? v : plain;
    if (!src || (v && !g)) {
      fh_set_error("missing checkpoint tensor '%sweight' (or its weight-norm pair original0/original1, weight_g/weight_v)", p.c_str());
      return FLOAT_E_MISSING;
    }
    FH_REQUIRE(src->ndim == 3 && src->shape[0] == h->D && src->shape[1] == cpg && src->shape[2] == K, "pos_conv_embed weight must be (%d,%d,%d)",
               h->D, cpg, K);
    std::vector<float> wf((size_t)h->D * cpg * K);
    if (v) {
      FH_REQUIRE(TensorTable::numel(g) == K, "pos_conv_embed weight_g must have %d elements", K);
      for (int t = 0; t < K; ++t) {
        double n2 = 0.0;
        for (size_t oc = 0; oc < (size_t)h->D * cpg; ++oc) n2 += (double)v->data[oc * K + t] * v->data[oc * K + t];
        const float sc = g->data[t] / (float)sqrt(n2);
        for (size_t oc = 0; oc < (size_t)h->D * cpg; ++oc) wf[oc * K + t] = v->data[oc * K + t] * sc;
      }
    } else {
      memcpy(wf.data(), plain->data, wf.size() * sizeof(float));
    }
    int merge = 1;
    while ((cpg * merge) % 32 != 0 && merge < c.pos_groups) merge *= 2;
    const int GP = cpg * merge;
    FH_REQUIRE(GP % 32 == 0 && GP <= 128 && c.pos_groups % merge == 0, "positional conv with %d channels per group is not supported", cpg);
    h->pos_gp = GP;
    h->pos_pairs = c.pos_groups / merge;
    typedef typename T::elem E;
    std::vector<E> hw((size_t)h->pos_pairs * K * GP * GP, T::host_from_float(0.f));
    for (int o = 0; o < h->D; ++o) {
      const int grp = o / cpg, pair = grp / merge, ol = o - pair * GP;
      for (int ci = 0; ci < cpg; ++ci) {
        const int cl = (grp % merge) * cpg + ci;  // input channel inside the merged slab
        for (int t = 0; t < K; ++t) hw[(((size_t)pair * K + t) * GP + ol) * GP + cl] = T::host_from_float(wf[((size_t)o * cpg + ci) * K + t]);
      }
    }
    E* dPW = nullptr;
    if ((rc = h->pool.alloc(&dPW, hw.size(), false))) return rc;
    h->pos_w = dPW;
    FH_CHECK_HIP(hipMemcpy(dPW, hw.data(), hw.size() * sizeof(E), hipMemcpyHostToDevice));
    const float_tensor_t* b = need(tt, p + "bias");
    if (!b) return FLOAT_E_MISSING;
    if ((rc = upload(&h->pool, b->data, h->D, &h->pos_b))) return rc;
  }
  if ((rc = load_ln(h, tt, W2 + "encoder.layer_norm", h->D, &h->enc_ln))) return rc;
  // ---- transformer layers (post-LN, do_stable_layer_norm = false)
  h->layers.resize(c.layers);
  for (int l = 0; l < c.layers; ++l) {
    const std::string p = W2 + "encoder.layers." + std::to_string(l) + ".";
    TLayer& L = h->layers[l];
    if ((rc = fmt_pack_linear(&h->pool, c.dtype, tt, {p + "attention.q_proj", p + "attention.k_proj", p + "attention.v_proj"}, h->D, h->D, &L.qkv)))
      return rc;
    if ((rc = fmt_pack_linear(&h->pool, c.dtype, tt, {p + "attention.out_proj"}, h->D, h->D, &L.out))) return rc;
    if ((rc = fmt_pack_linear(&h->pool, c.dtype, tt, {p + "feed_forward.intermediate_dense"}, c.intermediate, h->D, &L.ff1))) return rc;
    if ((rc = fmt_pack_linear(&h->pool, c.dtype, tt, {p + "feed_forward.output_dense"}, h->D, c.intermediate, &L.ff2))) return rc;
    if ((rc = load_ln(h, tt, p + "layer_norm", h->D, &L.ln1))) return rc;
    if ((rc = load_ln(h, tt, p + "final_layer_norm", h->D, &L.ln2))) return rc;
  }
  if (c.num_labels > 0) {
    // ---- classification head (wav2vec2_ser.py:23-38): dense -> tanh -> out_proj, fp32
    const float_tensor_t* dw = need(tt, "classifier.dense.weight");
    const float_tensor_t* db = need(tt, "classifier.dense.bias");
    const float_tensor_t* ow = need(tt, "classifier.out_proj.weight");
    const float_tensor_t* ob = need(tt, "classifier.out_proj.bias");
    if (!dw || !db || !ow || !ob) return FLOAT_E_MISSING;
    FH_REQUIRE(TensorTable::numel(dw) == (int64_t)h->D * h->D && TensorTable::numel(ow) == (int64_t)c.num_labels * h->D,
               "classifier head must be dense (%d,%d) and out_proj (%d,%d)", h->D, h->D, c.num_labels, h->D);
    if ((rc = upload(&h->pool, dw->data, (size_t)h->D * h->D, &h->cls_dense_w))) return rc;
    if ((rc = upload(&h->pool, db->data, (size_t)h->D, &h->cls_dense_b))) return rc;
    if ((rc = upload(&h->pool, ow->data, (size_t)c.num_labels * h->D, &h->cls_out_w))) return rc;
    if ((rc = upload(&h->pool, ob->data, (size_t)c.num_labels, &h->cls_out_b))) return rc;
    if ((rc = h->pool.alloc(&h->pooled, (size_t)h->D))) return rc;
    if ((rc = h->pool.alloc(&h->cls_h, (size_t)h->D))) return rc;
    if ((rc = h->pool.alloc(&h->cls_logits, (size_t)std::max(c.num_labels, 16)))) return rc;
  } else {
    // ---- audio projection: Linear(layers*D | D -> dim_w) + LayerNorm + SiLU (FLOAT.py:338-342)
    const int din = c.only_last ? h->D : c.layers * h->D;
    if ((rc = fmt_pack_linear(&h->pool, c.dtype, tt, {"audio_projection.0"}, c.dim_w, din, &h->aproj))) return rc;
    if ((rc = load_ln(h, tt, "audio_projection.1", c.dim_w, &h->aproj_ln))) return rc;
  }
  fmt_gemm_prime(c.dtype);
  return FLOAT_OK;
}

// Frames per call: the attention kernel tiles the keys (aud_kernels.hpp), so nothing in the operator depends on the clip
// length any more; what is left is 32-bit element indices in the row-major buffers (Tn * 3 * D and Tn * intermediate must
// stay below 2^31) and the O(Tn^2) cost of full attention: 200 000 frames per call = 2.2 h of audio at 25 fps.
constexpr int kAudMaxFrames = 200000;

// float_aud_reserve: the only place the operator allocates after create.  Run-time calls check the capacity and refuse.
int ensure_workspace(float_aud* h, int n_samples, int Tn, hipStream_t st) {
  if ((size_t)n_samples <= h->cap_samples && Tn <= h->cap_T) return FLOAT_OK;
  const float_aud_cfg_t& c = h->cfg;
  FH_CHECK_HIP(hipStreamSynchronize(st));  // earlier calls may still use the old buffers
  h->ws.release();
  h->ws = DevicePool();
  const size_t ns = std::max((size_t)n_samples, h->cap_samples);
  const int Tc = std::max(Tn, h->cap_T);
  const size_t L0 = (ns - c.conv_kernel[0]) / c.conv_stride[0] + 1;
  const size_t L1 = c.n_conv > 1 ? (L0 - c.conv_kernel[1]) / c.conv_stride[1] + 1 : 1;
  const int Mp = round_up(Tc, 80) + 80;
  const int D = h->D, C = h->C;
  int rc = 0;
  auto A = [&](auto** p, size_t n) {
    if (!rc) rc = h->ws.alloc(p, n, true);
  };
  const size_t esz = c.dtype == FLOAT_DT_FP32 ? 4 : 2;  // bytes per activation element
  auto AE = [&](void** p, size_t n) {  // n elements of the operand type
    unsigned char* b = nullptr;
    if (!rc) rc = h->ws.alloc(&b, n * esz, true);
    *p = b;
  };
  AE(&h->fa, L0 * C + 64);
  AE(&h->fb, L1 * C + 64);
  A(&h->part, ((L0 + 63) / 64) * C * 2);
  A(&h->scsh, (size_t)2 * C);
  AE(&h->x16, (size_t)Mp * C);
  AE(&h->hp16, (size_t)Mp * D);
  AE(&h->qkv16, (size_t)Mp * 3 * D);
  AE(&h->att16, (size_t)Mp * D);
  AE(&h->hid16, (size_t)Mp * c.intermediate);
  AE(&h->stack16, (size_t)Mp * std::max(h->aproj.K, 128));
  A(&h->hproj, (size_t)Mp * D);
  A(&h->pos, (size_t)Mp * D);
  A(&h->h, (size_t)Mp * D);
  A(&h->h1, (size_t)Mp * D);
  A(&h->y, (size_t)Mp * D);
  A(&h->yproj, (size_t)Mp * std::max(c.dim_w, 256));
  if (rc) return rc;
  h->cap_samples = ns;
  h->cap_T = Tc;
  h->Mp = Mp;
  return FLOAT_OK;
}

template <class T>
int launch_ln(int D, const AudLnArgs& g, hipStream_t st) {
  dim3 grid((g.M + 3) / 4);
  switch (D / 256) {
    case 1: hipLaunchKernelGGL((aud_ln_kernel<T, 1>), grid, dim3(256), 0, st, g); break;
    case 2: hipLaunchKernelGGL((aud_ln_kernel<T, 2>), grid, dim3(256), 0, st, g); break;
    case 3: hipLaunchKernelGGL((aud_ln_kernel<T, 3>), grid, dim3(256), 0, st, g); break;
    case 4: hipLaunchKernelGGL((aud_ln_kernel<T, 4>), grid, dim3(256), 0, st, g); break;
    default: fh_set_error("LayerNorm width %d unsupported", D); return FLOAT_E_INVALID;
  }
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

// Length of the feature sequence the extractor produces for n_samples (before any interpolation).
int feature_len(const float_aud_cfg_t& c, int n_samples) {
  long L = n_samples;
  for (int i = 0; i < c.n_conv; ++i) {
    if (L < c.conv_kernel[i]) return 0;
    L = (L - c.conv_kernel[i]) / c.conv_stride[i] + 1;
  }
  return (int)L;
}

// Tn: number of frames fed to the transformer: seq_len (interpolated, audio conditioning) or, Tn <= 0, the
// extractor's own length (speech-emotion model, no interpolation).  wa: (Tn, dim_w) or scores: (num_labels).
template <class T>
int inference_impl(float_aud* h, const float* a, int n_samples, int Tn, float* out, hipStream_t st) {
  typedef typename T::elem E;
  const float_aud_cfg_t& c = h->cfg;
  const int C = h->C, D = h->D;
  int rc;
  const int Lfeat = feature_len(c, n_samples);
  FH_REQUIRE(Lfeat >= 1, "audio too short for the feature extractor (%d samples)", n_samples);
  if (Tn <= 0) Tn = Lfeat;
  FH_REQUIRE(Tn <= kAudMaxFrames, "%d frames: one call takes at most %d transformer frames (32-bit element indices)", Tn, kAudMaxFrames);
  FH_REQUIRE((size_t)n_samples <= h->cap_samples && Tn <= h->cap_T,
             "clip of %d samples / %d frames exceeds the reserved workspace (%zu samples / %d frames): call float_aud_reserve "
             "first (run-time calls do not allocate)", n_samples, Tn, h->cap_samples, h->cap_T);
  // ---- feature extractor
  int L = (n_samples - c.conv_kernel[0]) / c.conv_stride[0] + 1;
  FH_REQUIRE(c.conv_kernel[0] == 10, "first conv kernel must be 10 (got %d)", c.conv_kernel[0]);
  if (c.feat_norm_layer) {
    hipLaunchKernelGGL((aud_conv0_ln_kernel<T, 10>), dim3((L + 3) / 4), dim3(256), 0, st, a, h->w0, h->b0, c.conv_stride[0], L, C, h->gn.g, h->gn.b,
                       1e-5f, reinterpret_cast<E*>(h->fa), h->sat);
  } else {
    const int tchunk = 64, nchunk = (L + tchunk - 1) / tchunk;
    hipLaunchKernelGGL((aud_conv0_stats_kernel<10>), dim3(nchunk, (C + 255) / 256), dim3(256), 0, st, a, n_samples, h->w0, c.conv_stride[0], L, C,
                       tchunk, h->part);
    hipLaunchKernelGGL(aud_gn_final_kernel, dim3((C + 3) / 4), dim3(256), 0, st, h->part, nchunk, C, L, h->gn.g, h->gn.b, 1e-5f, h->scsh);
    const size_t tot = (size_t)L * (C / 8);
    hipLaunchKernelGGL((aud_conv0_apply_kernel<T, 10>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, a, h->w0, c.conv_stride[0], L, C,
                       h->scsh, reinterpret_cast<E*>(h->fa), h->sat);
  }
  E *cur = reinterpret_cast<E*>(h->fa), *nxt = reinterpret_cast<E*>(h->fb);
  for (const ConvL& Lc : h->convs) {
    const int Lo = (L - Lc.k) / Lc.stride + 1;
    AudGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.sat = h->sat;
    g.A = cur;
    g.lda = (long long)Lc.stride * Lc.cin;
    g.W = Lc.W;
    g.bias = Lc.bias;
    g.out = nxt;
    g.M = Lo;
    g.N = Lc.cout;
    g.K = Lc.k * Lc.cin;
    g.ldc = Lc.cout;
    g.act = c.feat_norm_layer ? 0 : 1;
    {
      constexpr int smem = 2 * (128 + 64) * 64 * T::EB;  // 48 KiB with 16-bit operands, 96 KiB in the fp32 mode
      // the dynamic-LDS limit is a per-device function attribute: raised at float_aud_create on the handle's device (create_impl)
      hipLaunchKernelGGL((aud_gemm_tile_kernel<T>), dim3((Lo + 127) / 128, Lc.cout / 64), dim3(256), smem, st, g);
    }
    if (c.feat_norm_layer) {  // LayerNorm over channels + GELU, in place
      dim3 grid((Lo + 3) / 4);
      if (Lc.cout == 512) hipLaunchKernelGGL((aud_rowln_gelu_kernel<T, 2>), grid, dim3(256), 0, st, nxt, Lo, Lc.ln.g, Lc.ln.b, 1e-5f, h->sat);
      else hipLaunchKernelGGL((aud_rowln_gelu_kernel<T, 1>), grid, dim3(256), 0, st, nxt, Lo, Lc.ln.g, Lc.ln.b, 1e-5f, h->sat);
    }
    std::swap(cur, nxt);
    L = Lo;
  }
  FH_CHECK_HIP(hipGetLastError());
  // ---- (interpolate to Tn frames +) feature-projection LayerNorm -> packed x16; projection -> hproj (fp32)
  {
    dim3 grid((Tn + 3) / 4);
    switch (C / 256) {
      case 1: hipLaunchKernelGGL((aud_interp_ln_kernel<T, 1>), grid, dim3(256), 0, st, cur, L, Tn, h->fp_ln.g, h->fp_ln.b, c.ln_eps, reinterpret_cast<E*>(h->x16), h->sat); break;
      case 2: hipLaunchKernelGGL((aud_interp_ln_kernel<T, 2>), grid, dim3(256), 0, st, cur, L, Tn, h->fp_ln.g, h->fp_ln.b, c.ln_eps, reinterpret_cast<E*>(h->x16), h->sat); break;
      default: fh_set_error("conv_dim %d unsupported", C); return FLOAT_E_INVALID;
    }
    GemmArgs g = fmt_gemm_args(reinterpret_cast<const u16*>(h->x16), h->fp_proj, Tn);
    g.sat = h->sat;
    g.out_f32 = h->hproj;
    g.ldo = D;
    if ((rc = fmt_gemm_run(c.dtype, EPI_F32, g, st))) return rc;
  }
  // ---- encoder.  Post-LayerNorm (wav2vec2-base): hidden = LN(hidden + gelu(pos_conv(hidden))), then per layer
  // h = LN(h + attn(h)); h = LN(h + ffn(h)).  Stable (pre-)LayerNorm (the speech-emotion model): hidden += pos, per layer
  // h += attn(LN(h)); h += ffn(LN(h)); one LayerNorm after the last layer.  `h->h` is the fp32 residual stream,
  // `h->hp16` the packed operand of the next GEMM in both cases.
  auto ln = [&](const float* a_in, const float* res, const LnP& p, float* o32, void* o16, int keep_sum, void* stack, int stack_col) {
    AudLnArgs g;
    memset(&g, 0, sizeof(g));
    g.sat = h->sat;
    g.a = a_in;
    g.res = res;
    g.gamma = p.g;
    g.beta = p.b;
    g.eps = c.ln_eps;
    g.out_f32 = o32;
    g.out_p16 = o16;
    g.keep_sum = keep_sum;
    g.out_stack = stack;
    g.stack_col = stack_col;
    g.stack_kb = h->aproj.K / 32;
    g.M = Tn;
    return launch_ln<T>(D, g, st);
  };
  {
    dim3 grid((Tn + 15) / 16, h->pos_pairs);
#define POS_CASE(GP) \
  case GP: hipLaunchKernelGGL((aud_posconv_kernel<T, GP>), grid, dim3(256), 0, st, h->hproj, Tn, D, reinterpret_cast<const E*>(h->pos_w), h->pos_b, c.pos_k, c.pos_k / 2, h->pos, h->sat); break;
    switch (h->pos_gp) {
      POS_CASE(32) POS_CASE(64) POS_CASE(96) POS_CASE(128)
      default: fh_set_error("merged positional group width %d unsupported", h->pos_gp); return FLOAT_E_INVALID;
    }
#undef POS_CASE
    if (c.stable_ln) rc = ln(h->hproj, h->pos, h->layers[0].ln1, h->h, h->hp16, 1, nullptr, 0);
    else rc = ln(h->hproj, h->pos, h->enc_ln, h->h, h->hp16, 0, nullptr, 0);
    if (rc) return rc;
  }
  for (int l = 0; l < c.layers; ++l) {
    const TLayer& Ly = h->layers[l];
    {
      GemmArgs g = fmt_gemm_args(reinterpret_cast<const u16*>(h->hp16), Ly.qkv, Tn);
      g.sat = h->sat;
      g.out16 = reinterpret_cast<u16*>(h->qkv16);
      g.ldo16 = 3 * D;
      if ((rc = fmt_gemm_run(c.dtype, EPI_T16, g, st))) return rc;
    }
    {
      // 16-bit operands: the matrix-pipe kernel (64 queries per workgroup, K / V shared); fp32 mode and FLOAT_AUD_ATTN_MFMA=0: one wave per query
      static const bool mfma_on = !(getenv("FLOAT_AUD_ATTN_MFMA") && atoi(getenv("FLOAT_AUD_ATTN_MFMA")) == 0);
      if constexpr (!T::is32) {
        if (mfma_on)
          hipLaunchKernelGGL((aud_attn_mfma_kernel<T>), dim3((Tn + 63) / 64, c.heads), dim3(256), 0, st, reinterpret_cast<const E*>(h->qkv16), Tn, D,
                             c.heads, reinterpret_cast<E*>(h->att16), h->sat);
      }
      if (T::is32 || !mfma_on)
        hipLaunchKernelGGL((aud_attn_kernel<T>), dim3((Tn + 3) / 4, c.heads), dim3(256), 0, st, reinterpret_cast<const E*>(h->qkv16), Tn, D, c.heads,
                         reinterpret_cast<E*>(h->att16), h->sat);
    }
    {
      GemmArgs g = fmt_gemm_args(reinterpret_cast<const u16*>(h->att16), Ly.out, Tn);
      g.sat = h->sat;
      g.out_f32 = h->y;
      g.ldo = D;
      if ((rc = fmt_gemm_run(c.dtype, EPI_F32, g, st))) return rc;
    }
    // post-LN: h1 = layer_norm(h + attn);  stable: h1 = h + attn, operand = final_layer_norm(h1)
    if ((rc = ln(h->y, h->h, c.stable_ln ? Ly.ln2 : Ly.ln1, h->h1, h->hp16, c.stable_ln, nullptr, 0))) return rc;
    {
      GemmArgs g = fmt_gemm_args(reinterpret_cast<const u16*>(h->hp16), Ly.ff1, Tn);
      g.sat = h->sat;
      g.out16 = reinterpret_cast<u16*>(h->hid16);
      g.ldo16 = Ly.ff2.K / 32;
      if ((rc = fmt_gemm_run(c.dtype, EPI_GELUERF_P16, g, st))) return rc;
    }
    {
      GemmArgs g = fmt_gemm_args(reinterpret_cast<const u16*>(h->hid16), Ly.ff2, Tn);
      g.sat = h->sat;
      g.out_f32 = h->y;
      g.ldo = D;
      if ((rc = fmt_gemm_run(c.dtype, EPI_F32, g, st))) return rc;
    }
    if (c.stable_ln) {
      // h = h1 + ffn; operand = next layer's layer_norm(h), or, after the last layer, h = encoder.layer_norm(h)
      const bool last = l == c.layers - 1;
      if ((rc = ln(h->y, h->h1, last ? h->enc_ln : h->layers[l + 1].ln1, h->h, last ? nullptr : h->hp16, last ? 0 : 1, nullptr, 0))) return rc;
    } else {
      // h = final_layer_norm(h1 + ffn): also hidden_states[l + 1] of the stack (FLOAT.py:345-352)
      const bool stack = c.num_labels == 0 && (!c.only_last || l == c.layers - 1);
      if ((rc = ln(h->y, h->h1, Ly.ln2, h->h, h->hp16, 0, stack ? h->stack16 : nullptr, c.only_last ? 0 : l * D))) return rc;
    }
  }
  if (c.num_labels > 0) {
    // ---- mean over time -> dense -> tanh -> out_proj -> softmax (wav2vec2_ser.py:58-75,94-96; FLOAT.py:396-401)
    hipLaunchKernelGGL(aud_meanpool_kernel, dim3((D + 63) / 64), dim3(256), 0, st, h->h, Tn, D, h->pooled);
    hipLaunchKernelGGL(aud_dense_kernel, dim3((D + 3) / 4), dim3(256), 0, st, h->pooled, h->cls_dense_w, h->cls_dense_b, h->cls_h, D, D, 1);
    hipLaunchKernelGGL(aud_dense_kernel, dim3((c.num_labels + 3) / 4), dim3(256), 0, st, h->cls_h, h->cls_out_w, h->cls_out_b, h->cls_logits,
                       c.num_labels, D, 0);
    hipLaunchKernelGGL(aud_softmax_kernel, dim3(1), dim3(64), 0, st, h->cls_logits, out, c.num_labels);
  } else {
    // ---- audio projection: Linear -> LayerNorm -> SiLU
    GemmArgs g = fmt_gemm_args(reinterpret_cast<const u16*>(h->stack16), h->aproj, Tn);
    g.sat = h->sat;
    g.out_f32 = h->yproj;
    g.ldo = c.dim_w;
    if ((rc = fmt_gemm_run(c.dtype, EPI_F32, g, st))) return rc;
    AudLnArgs n;
    memset(&n, 0, sizeof(n));
    n.sat = h->sat;
    n.a = h->yproj;
    n.gamma = h->aproj_ln.g;
    n.beta = h->aproj_ln.b;
    n.eps = 1e-5f;  // nn.LayerNorm default (FLOAT.py:340)
    n.out_f32 = out;
    n.M = Tn;
    n.silu = 1;
    if ((rc = launch_ln<T>(c.dim_w, n, st))) return rc;
  }
  FH_CHECK_HIP(hipGetLastError());
  return FLOAT_OK;
}

}  // namespace

extern "C" {

int float_aud_create(const float_aud_cfg_t* cfg, const float_tensor_t* tensors, int32_t n_tensors, float_aud_t** out) {
  FH_REQUIRE(cfg && tensors && out, "null argument to float_aud_create");
  FH_REQUIRE(cfg->n_conv >= 2 && cfg->n_conv <= 8, "n_conv must be in [2,8] (got %d)", cfg->n_conv);
  for (int i = 0; i < cfg->n_conv; ++i) {
    FH_REQUIRE(cfg->conv_dim[i] == cfg->conv_dim[0] && cfg->conv_dim[i] % 256 == 0 && cfg->conv_dim[i] <= 512,
               "conv_dim must be one value, 256 or 512 (layer %d: %d)", i, cfg->conv_dim[i]);
    FH_REQUIRE(cfg->conv_kernel[i] >= 1 && cfg->conv_stride[i] >= 1, "bad conv kernel/stride at layer %d", i);
  }
  FH_REQUIRE(cfg->hidden % 256 == 0 && cfg->hidden <= 1024, "hidden size %d unsupported (multiple of 256, <= 1024)", cfg->hidden);
  FH_REQUIRE(cfg->heads > 0 && cfg->hidden / cfg->heads == 64 && cfg->hidden % cfg->heads == 0, "head dim must be 64 (hidden %d, heads %d)",
             cfg->hidden, cfg->heads);
  FH_REQUIRE(cfg->intermediate % 128 == 0 && cfg->layers >= 1, "bad intermediate size / layer count");
  FH_REQUIRE(cfg->pos_k % 4 == 0 && cfg->pos_groups >= 1 && cfg->hidden % cfg->pos_groups == 0, "bad positional conv shape");
  FH_REQUIRE(cfg->num_labels > 0 || (cfg->dim_w % 256 == 0 && cfg->dim_w <= 1024), "dim_w %d unsupported", cfg->dim_w);
  FH_REQUIRE(cfg->num_labels >= 0 && cfg->num_labels <= 64, "num_labels %d unsupported", cfg->num_labels);
  FH_REQUIRE(cfg->dtype == FLOAT_DT_BF16 || cfg->dtype == FLOAT_DT_FP16 || cfg->dtype == FLOAT_DT_FP32, "unknown dtype %d", cfg->dtype);
  float_aud* h = new float_aud();
  h->cfg = *cfg;
  TensorTable tt(tensors, n_tensors);
  int rc = (cfg->dtype == FLOAT_DT_BF16) ? create_impl<BF16>(h, tt)
           : (cfg->dtype == FLOAT_DT_FP32) ? create_impl<FP32>(h, tt) : create_impl<FP16>(h, tt);
  if (!rc) rc = h->pool.alloc(&h->sat, 1);
  if (!rc) {
    // dynamic-LDS limit of the conv GEMM on THIS device (a second GPU of the process gets its own: the attribute is per device)
    const int eb = cfg->dtype == FLOAT_DT_FP32 ? 4 : 2, smem = 2 * (128 + 64) * 64 * eb;
    const void* k = cfg->dtype == FLOAT_DT_BF16   ? reinterpret_cast<const void*>(aud_gemm_tile_kernel<BF16>)
                    : cfg->dtype == FLOAT_DT_FP32 ? reinterpret_cast<const void*>(aud_gemm_tile_kernel<FP32>)
                                                  : reinterpret_cast<const void*>(aud_gemm_tile_kernel<FP16>);
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) (void)hipGetLastError();
  }
  if (rc) {
    float_aud_destroy(h);
    return rc;
  }
  *out = h;
  return FLOAT_OK;
}

void float_aud_destroy(float_aud_t* h) {
  if (!h) return;
  h->pool.release();
  h->ws.release();
  delete h;
}

int float_aud_classify(float_aud_t* h, const float* a, int32_t n_samples, float* scores, void* stream) {
  FH_REQUIRE(h && a && scores, "null argument to float_aud_classify");
  FH_REQUIRE(h->cfg.num_labels > 0, "this handle was created with the audio projection head (num_labels = 0)");
  FH_REQUIRE(n_samples >= 400, "audio too short: %d samples (the feature extractor needs >= 400)", n_samples);
  hipStream_t st = (hipStream_t)stream;
  if (h->cfg.dtype == FLOAT_DT_FP32) return inference_impl<FP32>(h, a, n_samples, 0, scores, st);
  return h->cfg.dtype == FLOAT_DT_BF16 ? inference_impl<BF16>(h, a, n_samples, 0, scores, st)
                                       : inference_impl<FP16>(h, a, n_samples, 0, scores, st);
}

int float_aud_saturation(float_aud_t* h, uint64_t* total, int32_t reset, void* stream) {
  FH_REQUIRE(h && total, "null argument to float_aud_saturation");
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

int float_aud_reserve(float_aud_t* h, int32_t n_samples, int32_t seq_len, void* stream) {
  FH_REQUIRE(h != nullptr, "null audio handle");
  FH_REQUIRE(n_samples >= 400, "audio too short: %d samples (the feature extractor needs >= 400)", n_samples);
  const int Lfeat = feature_len(h->cfg, n_samples);
  const int Tn = seq_len > 0 ? seq_len : Lfeat;
  FH_REQUIRE(Tn >= 1 && Tn <= kAudMaxFrames, "%d frames: one call takes at most %d transformer frames (32-bit element indices)", Tn, kAudMaxFrames);
  return ensure_workspace(h, n_samples, Tn, (hipStream_t)stream);
}

int float_aud_inference(float_aud_t* h, const float* a, int32_t n_samples, int32_t seq_len, float* wa, void* stream) {
  FH_REQUIRE(h && a && wa, "null argument to float_aud_inference");
  FH_REQUIRE(h->cfg.num_labels == 0, "this handle was created with the classification head (num_labels = %d)", h->cfg.num_labels);
  FH_REQUIRE(seq_len >= 1, "seq_len must be >= 1 (got %d)", seq_len);
  FH_REQUIRE(n_samples >= 400, "audio too short: %d samples (the feature extractor needs >= 400)", n_samples);
  hipStream_t st = (hipStream_t)stream;
  if (h->cfg.dtype == FLOAT_DT_FP32) return inference_impl<FP32>(h, a, n_samples, seq_len, wa, st);
  return h->cfg.dtype == FLOAT_DT_BF16 ? inference_impl<BF16>(h, a, n_samples, seq_len, wa, st)
                                       : inference_impl<FP16>(h, a, n_samples, seq_len, wa, st);
}

}  // extern "C"
