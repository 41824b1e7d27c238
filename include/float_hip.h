/* float_hip.h - C ABI of libfloat_hip.so: the FLOAT hot path on MI355X (gfx950).
 *
 * Two operators replace the two loops of the reference (set-soft/ComfyUI-FLOAT_Optimized):
 *
 *   FMT  - FlowMatchingTransformer.forward_with_cfv and the fixed-grid Euler loop around it
 *          (reference src/nodes/models/float/FMT.py:277-401, FLOAT.py:209-253,
 *          nodes_adv.py:545-694).
 *   DEC  - Synthesis.forward + per-frame post-process
 *          (reference src/nodes/models/float/styledecoder.py:497-534, FLOAT.py:113-169,
 *          nodes_vadv.py:447-464).
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 on success or a FLOAT_E_* code,
 *     float_last_error() gives the message (thread-local).  No exception crosses the ABI.
 *   - all tensor arguments of the run-time calls are DEVICE pointers to contiguous fp32
 *     (the reference's tensors are fp32 everywhere); `stream` is a hipStream_t passed as void*.
 *     All work is enqueued on that stream; nothing synchronises the device.
 *   - batch size is 1 at this level, exactly like the reference decode loop (FLOAT.py:140);
 *     the host mirror loops over batch items as FloatProcess does (nodes.py:189-209).  The one
 *     batched entry is float_fmt_sample_batch (the reference's samplers take a batch).
 *   - a handle owns its packed weights and a fixed workspace allocated at create time; no
 *     allocation happens inside the run-time calls (float_aud_reserve is the explicit exception).
 *     Every run-time call enqueues kernels only (device-to-device moves included: memcpy / memset
 *     nodes of a caller's stream capture did not replay reproducibly on ROCm 7.2), except
 *     float_dec_frames_host, whose last batch goes to the host by hipMemcpyAsync.  Capture by the
 *     caller (hipStreamBeginCapture on `stream`) is tested for the float_fmt_* calls.
 *   - calls on one handle must be serialised by the caller; different handles are independent.
 */
#ifndef FLOAT_HIP_H
#define FLOAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLOAT_HIP_ABI_VERSION 6

enum {
  FLOAT_OK = 0,
  FLOAT_E_INVALID = 1,   /* bad argument / shape mismatch (ValueError in the host mirror) */
  FLOAT_E_MISSING = 2,   /* a required checkpoint tensor is absent (KeyError) */
  FLOAT_E_HIP = 3,       /* HIP runtime error (RuntimeError) */
  FLOAT_E_NOMEM = 4
};

/* MFMA operand type; accumulation, statistics, softmax, residual stream and ODE state are fp32 in every mode.
 * FLOAT_DT_FP32 (every operator): the verification mode - fp32 operands on v_mfma_f32_16x16x4_f32, the same launch chain and
 * the same kernels with 4-byte elements, held to the reference goldens at 1e-4 (tests/test_fmt_fp32_gpu.py,
 * tests/test_dec_fp32_gpu.py); 1/16 of the 16-bit MFMA rate, no tuned tilings. */
enum { FLOAT_DT_BF16 = 0, FLOAT_DT_FP16 = 1, FLOAT_DT_FP32 = 2 };

/* One checkpoint tensor, named with the reference's state-dict key (prefix stripped):
 * e.g. "blocks.0.attn.qkv.weight" (FMT) or "convs.3.conv.weight" (decoder).  `data` is a HOST
 * pointer to contiguous fp32; it is only read during *_create. */
typedef struct {
  const char* name;
  const float* data;
  int32_t ndim;
  int64_t shape[6];
} float_tensor_t;

/* ---------------------------------------------------------------- FMT ------------- */
/* Shape contract = reference BaseOptions (options/base_options.py:36-45). */
typedef struct {
  int32_t dim_w, dim_a, dim_e, dim_h;
  int32_t depth, heads;
  int32_t mlp_hidden;      /* int(dim_h * mlp_ratio) */
  int32_t n_prev, n_cur;   /* num_prev_frames, int(wav2vec_sec*fps); n_prev + n_cur <= 80 */
  int32_t attn_window;     /* |i-j| <= window is visible (FMT.py:15-19) */
  int32_t dtype;           /* FLOAT_DT_* */
  int32_t use_graph;       /* 0: eager launches; != 0: replay each window's chain from a cached hipGraph (at most 8
                              graphs per handle, least recently used evicted).  1 and 2 are the same (1 used to put the
                              adaLN GEMM on a parallel branch; that GEMM now runs once per window, not per step). */
  int32_t max_batch;       /* clips float_fmt_sample_batch may stack per launch chain (sizes the workspace: rows = max_batch x 4 CFG
                              rows x tokens; the per-window modulation slab is 64 x rows x (depth * 6 + 2) * dim_h fp32, 3.1 GB per
                              clip at the default shape); 0 = 1 */
} float_fmt_cfg_t;

typedef struct float_fmt float_fmt_t;

int float_fmt_create(const float_fmt_cfg_t* cfg, const float_tensor_t* tensors, int32_t n_tensors,
                     float_fmt_t** out);
void float_fmt_destroy(float_fmt_t* h);

/* Fixed-grid solver of the sampling calls, the reference's TORCHDIFFEQ_FIXED_STEP_SOLVERS list
 * (src/nodes/__init__.py:15-23; options/base_options.py:50).  Default FLOAT_ODE_EULER (fused update in the last
 * GEMM's epilogue); the Runge-Kutta schemes evaluate the field s times per step (2/4/2/3). */
enum { FLOAT_ODE_EULER = 0, FLOAT_ODE_MIDPOINT = 1, FLOAT_ODE_RK4 = 2, FLOAT_ODE_HEUN2 = 3, FLOAT_ODE_HEUN3 = 4 };
int float_fmt_set_method(float_fmt_t* h, int32_t method);

/* Range check of the 16-bit modes of the FMT, encoder and audio operators (no reference counterpart: the reference is fp32).
 * fp16 ends at 65504.  Outside the decoder a 16-bit activation store CLAMPS at +-65504 (a NaN becomes -65504) so that one
 * outlier degrades the result instead of poisoning it - and every such store is COUNTED: each thread keeps the maximum
 * magnitude of the halves it stores and adds 1 to the handle's device counter if one reached the clamp value.  (The ResBlock
 * conv1 of the encoder runs the decoder's conv kernel, which stores inf instead of clamping and counts it the same way.)
 *   total: threads with at least one clamped / non-finite 16-bit store since the handle was created / last reset - zero or not
 *   is what matters.  Synchronises `stream`.  A non-zero total means the result is NOT the reference's within the stated
 *   tolerance: run that checkpoint with dtype FLOAT_DT_FP32 (or bf16 for the FMT / audio operators, whose exponent range is
 *   fp32's).  Always 0 for bf16 and fp32 handles. */
int float_fmt_saturation(float_fmt_t* h, uint64_t* total, int32_t reset, void* stream);

/* forward_with_cfv (FMT.py:342-401), B = 1.
 *   x, wa: (n_cur, dim)   wr: (dim_w)   we: (we_len, dim_e) with we_len 1 (static) or n_cur
 *   prev_x, prev_wa: (n_prev, dim)   prev_we: (n_prev, dim_e) - required iff we_len > 1
 *   out: (n_prev + n_cur, dim_w)
 * All three scales == 1 -> a single conditional pass (FMT.py:400-401); otherwise the 3-row
 * batch [uncond | all | audio-only], or the 4-row batch when include_r_cfg != 0. */
int float_fmt_eval(float_fmt_t* h, float t, const float* x, const float* wa, const float* wr,
                   const float* we, int32_t we_len, const float* prev_x, const float* prev_wa,
                   const float* prev_we, float a_cfg, float r_cfg, float e_cfg, int32_t include_r_cfg,
                   float* out, void* stream);

/* One window of the ODE loop (FLOAT.py:229-248): Euler over linspace(0,1,nfe), i.e. nfe-1
 * evaluations starting from x0 (n_cur, dim_w); out = final sample (n_cur, dim_w). */
int float_fmt_sample_chunk(float_fmt_t* h, const float* x0, const float* wa, const float* wr,
                           const float* we, int32_t we_len, const float* prev_x, const float* prev_wa,
                           const float* prev_we, int32_t nfe, float a_cfg, float r_cfg, float e_cfg,
                           int32_t include_r_cfg, float* out, void* stream);

/* The whole auto-regressive loop (FLOAT.py:209-253; nodes_adv.py:578-694).
 *   wr: (dim_w)   wa: (T, dim_a)   we: (1|T, dim_e)   noise: (ceil(T/n_cur), n_cur, dim_w),
 *   the explicit form of the reference's sequential randn draws (FLOAT.py:215)
 *   r_d: (T, dim_w).  Last window replicate-padded, prev_* handed off on the device. */
int float_fmt_sample(float_fmt_t* h, const float* wr, const float* wa, int32_t T, const float* we,
                     int32_t we_len, const float* noise, int32_t nfe, float a_cfg, float r_cfg,
                     float e_cfg, int32_t include_r_cfg, float* r_d, void* stream);

/* B independent clips of equal length through ONE launch chain (the reference's samplers take a batch: nodes_vadv.py:618-735
 * -> nodes_adv.py:545-694, x0 = randn(B, 50, 512) at FLOAT.py:215): the clips' rows are stacked, so every weight is read once
 * per evaluation for all of them.  Each clip's result is what float_fmt_sample gives for it alone, bit for bit where the
 * GEMM tilings coincide and within rounding otherwise (the tiling depends on the row count).
 *   wr: (B, dim_w)   wa: (B, T, dim_a)   we: (B, we_len, dim_e)   noise: (windows, B, n_cur, dim_w)   r_d: (B, T, dim_w)
 * n_clips <= max_batch of the handle. */
int float_fmt_sample_batch(float_fmt_t* h, int32_t n_clips, const float* wr, const float* wa, int32_t T,
                           const float* we, int32_t we_len, const float* noise, int32_t nfe, float a_cfg,
                           float r_cfg, float e_cfg, int32_t include_r_cfg, float* r_d, void* stream);

/* The same loop one window at a time, so the caller can overlap the decode of window k (on another
 * stream) with the sampling of window k+1: _begin only records the job (pointers must stay valid
 * until the last _next), each _next enqueues ONE window on `stream` and reports its index and how
 * many remain.  r_d rows [k*n_cur, min(T,(k+1)*n_cur)) are complete once that stream work is done. */
int float_fmt_sample_begin(float_fmt_t* h, const float* wr, const float* wa, int32_t T, const float* we,
                           int32_t we_len, const float* noise, int32_t nfe, float a_cfg, float r_cfg,
                           float e_cfg, int32_t include_r_cfg, float* r_d);
int float_fmt_sample_next(float_fmt_t* h, void* stream, int32_t* window_done, int32_t* windows_left);

/* A job over the windows [first_window, end_window) of the clip only, starting from the history the caller hands over: the
 * multi-GPU window shard (a rank samples its range from zero history, receives its predecessor's boundary latents by an RCCL
 * all_gather and re-solves from them).  No reference counterpart (the reference has no distributed code); the window body is
 * FLOAT.py:214-251 unchanged.  Same arguments as float_fmt_sample_begin (full-clip wa / we / noise / r_d pointers), plus
 *   hist_x (n_prev, dim_w), hist_wa (n_prev, dim_a), hist_we (n_prev, dim_e; read only with we_len > 1): the last n_prev rows
 *   of the previous window's sample / padded wa window / we window, or all NULL = zeros, the history of window 0.
 * Only r_d rows of the job's windows are written.  Continue with float_fmt_sample_next (windows_left counts to end_window). */
int float_fmt_sample_begin_range(float_fmt_t* h, const float* wr, const float* wa, int32_t T, const float* we,
                                 int32_t we_len, const float* noise, int32_t nfe, float a_cfg, float r_cfg,
                                 float e_cfg, int32_t include_r_cfg, float* r_d, int32_t first_window,
                                 int32_t end_window, const float* hist_x, const float* hist_wa, const float* hist_we);

/* Stream capture: every float_fmt_* run-time call may be issued while `stream` is being captured by the caller
 * (hipStreamBeginCapture); the chain is then launched straight into that capture instead of replaying the handle's own
 * graph, and no host memory is read by the stream (evaluation times are formed on the device).
 *
 * Test hook for the two tables the model builds instead of loading (FMT.py:15-40, 234-236, 249-250):
 *   what = 0: out (n_prev + n_cur, dim_h) = the positional table the handle adds in x_embedder (the checkpoint's
 *             `pos_embed` when given, else regenerated: nodes_vadv_loader.py:822-840), `in` unused;
 *   what = 1: the attention kernel of the chain on caller data: in (n_prev + n_cur, 3 * dim_h) = [q | k | v] rows of one
 *             CFG row, out (n_prev + n_cur, dim_h) = softmax(q k^T / sqrt(128) + band mask) v per head; with q = k = 0 and
 *             one-hot v rows the output shows which keys each query may see (enc_dec_mask, FMT.py:15-19).
 * in / out are fp32 device pointers. */
int float_fmt_debug(float_fmt_t* h, int32_t what, const float* in, float* out, void* stream);

/* ---------------------------------------------------------------- decoder --------- */
typedef struct {
  int32_t size;        /* output resolution, 64..512 (styledecoder.py:448) */
  int32_t style_dim;   /* 512 */
  int32_t dtype;       /* FLOAT_DT_FP16 (activations / conv weights), or FLOAT_DT_FP32 = the verification mode (the same launch
                          chain with 4-byte activations and weights); bf16 is refused: too few mantissa bits for the warp */
  int32_t max_frames;  /* frames decoded per internal batch (sizes the workspace) */
} float_dec_cfg_t;

typedef struct float_dec float_dec_t;

/* tensors: the Synthesis state dict (prefix `motion_autoencoder.dec.` stripped), reference key names.
 * Blur FIR of the up-sampling StyledConvs (styledecoder.py:209-213): the checkpoint's `convs.N.conv.blur.kernel` buffer where
 * present (what the reference ends up with after its strict load_state_dict, nodes_vadv_loader.py:632), else the optional
 * 1-D tensor `blur_kernel` (the loader's widget: Synthesis(blur_kernel=...), styledecoder.py:448), else [1,3,3,1].  4-tap
 * kernels only, a buffer must be make_kernel's outer product k (x) k.  ToRGB / ToFlow `to_rgbs.N.upsample.kernel` /
 * `to_flows.N.upsample.kernel` buffers (styledecoder.py:373,394: make_kernel([1,3,3,1]) * 4 by construction, the checkpoint's
 * after the strict load): any rank-1 4 x 4 kernel ky (x) kx is applied per level and per module; another size or a kernel of
 * rank > 1 is refused (FLOAT_E_INVALID). */
int float_dec_create(const float_dec_cfg_t* cfg, const float_tensor_t* tensors, int32_t n_tensors,
                     float_dec_t** out);
void float_dec_destroy(float_dec_t* h);

/* Per clip: the encoder's skip features, reference order (encoder.py:220-231):
 * feats[i] = (C_i, R_i, R_i) fp32 NCHW, R_i = 8 << i.  Repacked to NHWC 16-bit in the handle. */
int float_dec_set_feats(float_dec_t* h, const float* const* feats, int32_t n_feats, void* stream);

/* decode_latent_into_processed_images (FLOAT.py:113-169) without the host copy:
 *   s_r: (style_dim)   r_d: (n_frames, style_dim)
 *   out: (n_frames, size, size, 3) fp32 in [0,1] = clamp(rgb,-1,1)*0.5+0.5, HWC. */
int float_dec_frames(float_dec_t* h, const float* s_r, const float* r_d, int32_t n_frames,
                     float* out_hwc, void* stream);

/* The same with the reference's hand-over to host memory (FLOAT.py:139,157-167: frames land in a pre-allocated CPU tensor):
 * frames are rendered into out_hwc (device, n_frames * size * size * 3) and every finished batch of max_frames frames is
 * copied to host_hwc (same layout).
 *   host_hwc: PINNED or REGISTERED host memory (hipHostMalloc / hipHostRegister / torch pin_memory), 16-byte aligned, gets
 *     the fast path: copy workgroups inside the next batch's launches store the frames straight through its device-side
 *     address (hipPointerGetAttributes decides; nothing is assumed).  Pageable memory is accepted too and takes one
 *     hipMemcpyAsync behind each batch on `stream` (staged by the runtime, slower, same bytes).
 *   copy_stream == NULL or == stream: the copies ride along / are queued on `stream` behind each batch (in order).
 *   another stream: the copy of batch i runs there while `stream` renders batch i+1, and `stream` is made to wait for the
 *     last copy before the call returns.  Measured on MI355X / ROCm 7.2 this does NOT pay inside the whole path: the
 *     decode + copy phase drops from 42 to 30 ms per 250 frames, but a device-to-host copy issued on a second stream leaves
 *     the FMT chain that follows 12 ms slower (84.6 -> 97 ms, even after a full device synchronisation), see DESIGN.md.
 * Either way, synchronising `stream` (or an event recorded on it) means the frames are in host memory. */
int float_dec_frames_host(float_dec_t* h, const float* s_r, const float* r_d, int32_t n_frames,
                          float* out_hwc, float* host_hwc, void* stream, void* copy_stream);

/* Shape (channels, resolution) of skip feature i as float_dec_set_feats reads it: feats[i] must hold
 * channels * resolution * resolution floats.  Lets the caller validate tensors wired from an arbitrary encoder. */
int float_dec_feat_shape(float_dec_t* h, int32_t i, int32_t* channels, int32_t* resolution);

/* Same, but returns the un-clamped NCHW rgb of Synthesis.forward (styledecoder.py:532-534);
 * parity tests use it to look under the clamp. */
int float_dec_frames_raw(float_dec_t* h, const float* s_r, const float* r_d, int32_t n_frames,
                         float* out_chw, void* stream);

/* Range check of the 16-bit mode.  fp16 ends at 65504; the operator keeps what it stores small by construction (every
 * StyledConv's style is divided by its max |s| per frame, the factor goes into the demodulation's epsilon: exact algebra,
 * dec_kernels.hpp) and COUNTS what still did not fit: such a value is stored as inf and poisons what it touches (loud, not a
 * plausible-looking wrong frame); every thread keeps the maximum magnitude of the 16-bit values it stores and adds 1 to its
 * layer's counter per output tile that held an inf / NaN - the count is of (thread, tile) groups of 16..128 values, zero or not
 * is what matters.
 *   total: the sum since the handle was created / last reset;  per_site: NULL or FLOAT_DEC_SAT_SITES counters
 *   ([0..15] output of StyledConv i (0 = conv1, 1 + 2l / 2 + 2l = up-conv / conv of level l), [16 + l] blend of level l,
 *   [32] constant input, [33] skip features, [34] style vectors with a NaN / inf in them, i.e. a non-finite latent).
 *   Synchronises `stream`.  A non-zero total means the frames are NOT the reference's: run that checkpoint with dtype
 *   FLOAT_DT_FP32.  Always 0 in the fp32 mode (except [34]). */
#define FLOAT_DEC_SAT_SITES 40
int float_dec_saturation(float_dec_t* h, uint64_t* total, uint64_t* per_site, int32_t reset, void* stream);

/* Test hooks: ONE op of the decoder on caller data through the production kernels and launchers (what the full chain cannot
 * show: which op is off).  All pointers device fp32; tensors are host fp32 with the reference module's own key names.
 *   float_dec_debug_styled_conv: StyledConv.forward(x, style) (styledecoder.py:302-325, noise weight 0):
 *     leaky_relu(ModulatedConv2d(x, style) + bias) * sqrt(2).  keys `sc.conv.weight` (1,cout,cin,3,3),
 *     `sc.conv.modulation.weight` (cin,style_dim), `sc.conv.modulation.bias` (cin), `sc.activate.bias` (cout); optional
 *     `sc.conv.blur.kernel` (4,4) / `blur_kernel` (4) as in float_dec_create.
 *     x (n_frames,cin,res,res) NCHW, style (n_frames,style_dim), out (n_frames,cout,R',R'), R' = res or 2 res (upsample).
 *     Which kernel runs follows the production rules: plain conv res >= 16 -> dec_conv16_kernel, below -> dec_conv_kernel;
 *     up-conv 2 res >= 64 -> dec_zblur_kernel, res >= 8 -> dec_zconv4_kernel + dec_blur_kernel, res = 4 -> per-class
 *     dec_conv_kernel + dec_blur_kernel.  *saturated = the range counter of the op's 16-bit output stores (float_dec_saturation's rule: a value beyond fp16's range is stored as inf and counted).  flags bit 0: no style normalisation.
 *   float_dec_debug_flow_level: ToFlow (styledecoder.py:399-425) + ToRGB (:368-386) of one level: keys `to_flow.conv.weight`,
 *     `to_flow.conv.modulation.weight|bias`, `to_flow.bias`, `to_rgb.conv.0.weight`, `to_rgb.conv.1.bias`, `to_rgb.bias`.
 *     x (F,C,R,R) the conv output, feat (C,R,R) the skip feature (the reference repeats it over the batch), style (F,style_dim),
 *     prev_flow / prev_rgb (F,3,R/2,R/2) or NULL;  out_flow (F,3,R,R) = ToFlow's `out` before tanh / sigmoid,
 *     out_blend (F,C,R,R) = feat_warp + x (1 - mask), out_rgb (F,3,R,R) = ToRGB(feat_warp, prev_rgb).  Each output optional. */
typedef struct {
  int32_t dtype;      /* FLOAT_DT_FP16 | FLOAT_DT_FP32 */
  int32_t cin, cout;  /* flow level: cin = C, cout unused */
  int32_t res;        /* input resolution */
  int32_t upsample;
  int32_t n_frames;
  int32_t style_dim;
  int32_t flags;
} float_dec_unit_t;
int float_dec_debug_styled_conv(const float_dec_unit_t* u, const float_tensor_t* tensors, int32_t n_tensors, const float* x,
                                const float* style, float* out, uint64_t* saturated, void* stream);
int float_dec_debug_flow_level(const float_dec_unit_t* u, const float_tensor_t* tensors, int32_t n_tensors, const float* x,
                               const float* feat, const float* style, const float* prev_flow, const float* prev_rgb,
                               float* out_flow, float* out_blend, float* out_rgb, void* stream);

/* Direction.forward (styledecoder.py:428-444; FLOAT.py:289-291, nodes_vadv.py:479-533): r_s = lam @ Q^T with
 * Q from the QR of (direction.weight + 1e-8), factorised once at create time.  lam: (motion_dim), r_s: (style_dim). */
int float_dec_direction(float_dec_t* h, const float* lam, float* r_s, void* stream);

/* Same hand-over without the fp32 round trip: feats16[i] = (R_i, R_i, C_i) NHWC device buffers of element type `dtype`
 * (must equal the decoder's: 16-bit, or fp32 in the verification mode), e.g. the ones float_enc_feats16 returns. */
int float_dec_set_feats16(float_dec_t* h, const void* const* feats16, int32_t n_feats, int32_t dtype,
                          void* stream);

/* ---------------------------------------------------------------- encoder --------- */
/* Appearance / motion encoder, once per clip: EncoderApp.forward
 * (reference src/nodes/models/float/encoder.py:203-231), Encoder.fc (encoder.py:242-247) and
 * Direction (styledecoder.py:428-444, QR hoisted to create time), as called by
 * FLOAT.encode_image_into_latent / inference (FLOAT.py:283-291).
 * Checkpoint keys: `net_app.convs.*`, `fc.*` (prefix `motion_autoencoder.enc.` stripped) and,
 * optionally, `direction.weight` (from `motion_autoencoder.dec.`) to get r_s as well.  The Blur buffers of the down-sampling
 * ConvLayers (`net_app.convs.N.conv2.0.kernel`, `net_app.convs.N.skip.0.kernel`, encoder.py:59-75) are applied as the
 * checkpoint holds them (any 4 x 4 values; make_kernel([1,3,3,1]) where the state has none); another size is refused. */
typedef struct {
  int32_t size;        /* input resolution, power of two in [64, 1024] */
  int32_t dim;         /* 512 */
  int32_t dim_motion;  /* 20 */
  int32_t dtype;       /* FLOAT_DT_* of activations / conv weights (FLOAT_DT_FP32 = verification mode, feeds an fp32 decoder);
                          accumulation, s_r, fc are fp32 */
} float_enc_cfg_t;

typedef struct float_enc float_enc_t;

int float_enc_create(const float_enc_cfg_t* cfg, const float_tensor_t* tensors, int32_t n_tensors,
                     float_enc_t** out);
void float_enc_destroy(float_enc_t* h);

/* img: (3, size, size) fp32 NCHW in [-1,1] (generate.py:34-39).  Outputs (device, fp32, each optional):
 *   s_r (dim); lam = Encoder.fc(s_r) (dim_motion); r_s = Direction(lam) (dim; needs direction.weight);
 *   feats[i] (C_i, R_i, R_i) NCHW with R_i = 8 << i, the reference's res[::-1][2:] order
 *   (n_feats <= log2(size) - 2; NULL entries are skipped). */
int float_enc_forward(float_enc_t* h, const float* img, float* s_r, float* lam, float* r_s,
                      float* const* feats, int32_t n_feats, void* stream);

/* The NHWC 16-bit feature maps left in the handle by the last float_enc_forward (valid until the
 * next one), reference order; *n_out = how many were written (<= max_feats). */
int float_enc_saturation(float_enc_t* h, uint64_t* total, int32_t reset, void* stream); /* see float_fmt_saturation */
int float_enc_feats16(float_enc_t* h, const void** feats16, int32_t* channels, int32_t max_feats,
                      int32_t* n_out);
/* Stream-ordered copies of those maps into caller buffers (dst[i]: R_i x R_i x C_i elements of the operator's type, reference
 * order): a batch of portraits (nodes.py:189-209) keeps one set per item and hands it to float_dec_set_feats16 when the item
 * is decoded, instead of encoding the item a second time. */
int float_enc_export_feats16(float_enc_t* h, void* const* dst, int32_t n_feats, void* stream);

/* ---------------------------------------------------------------- audio encoder --- */
/* Audio conditioning, once per clip: AudioEncoder.inference (reference
 * src/nodes/models/float/FLOAT.py:370-375) = Wav2VecModel.forward(input_values, seq_len,
 * output_hidden_states=True) (src/nodes/models/wav2vec2.py:33-98: feature extractor ->
 * linear_interpolation(align_corners=True) to seq_len frames (:184-197) -> feature projection -> encoder),
 * hidden_states[1:] stacked per frame (FLOAT.py:345-352) and audio_projection = Linear -> LayerNorm ->
 * SiLU (FLOAT.py:338-342).  The wav2vec2 arithmetic itself lives in the `transformers` package
 * (requirements.txt:8, not vendored); it is restated from that package's module definitions for the bundled
 * wav2vec2_base config (src/nodes/model_configs/wav2vec2_base/config.json: feat_extract_norm "group",
 * conv_bias false, do_stable_layer_norm false, eager attention without mask).
 * Checkpoint keys: `wav2vec2.*` and `audio_projection.{0,1}.*` (prefix `audio_encoder.` stripped);
 * the positional conv accepts `parametrizations.weight.original0/1`, `weight_g/weight_v` or a plain `weight`. */
typedef struct {
  int32_t n_conv;              /* feature-extractor layers (7) */
  int32_t conv_dim[8], conv_kernel[8], conv_stride[8];
  int32_t hidden, layers, heads, intermediate;   /* 768, 12, 12, 3072 */
  int32_t pos_k, pos_groups;   /* num_conv_pos_embeddings 128, groups 16 */
  int32_t dim_w;               /* 512 */
  int32_t only_last;           /* opt.only_last_features */
  int32_t dtype;               /* FLOAT_DT_* of activations / weights (FLOAT_DT_FP32 = verification mode); statistics and the
                                  residual stream are fp32 */
  float ln_eps;                /* layer_norm_eps 1e-5 */
  /* Architecture switches of the wav2vec2-large family used by the speech-emotion model
   * (src/nodes/model_configs/emotion_ser/config.json): all 0 for wav2vec2-base. */
  int32_t feat_norm_layer;     /* feat_extract_norm == "layer": LayerNorm over channels after every conv */
  int32_t stable_ln;           /* do_stable_layer_norm: pre-LayerNorm encoder layers + one final LayerNorm */
  int32_t conv_bias;           /* feature-extractor convs carry a bias */
  int32_t num_labels;          /* > 0: classification head (float_aud_classify) instead of the audio projection */
} float_aud_cfg_t;

typedef struct float_aud float_aud_t;

int float_aud_create(const float_aud_cfg_t* cfg, const float_tensor_t* tensors, int32_t n_tensors,
                     float_aud_t** out);
void float_aud_destroy(float_aud_t* h);

/* Sizes the workspace for clips of up to n_samples samples and seq_len transformer frames (seq_len <= 0: the feature
 * extractor's own length, as float_aud_classify uses it).  The ONLY call that allocates after create: it synchronises
 * `stream` first when buffers have to grow (earlier work may still read the old ones) and never shrinks.  Not capturable. */
int float_aud_reserve(float_aud_t* h, int32_t n_samples, int32_t seq_len, void* stream);
int float_aud_saturation(float_aud_t* h, uint64_t* total, int32_t reset, void* stream); /* see float_fmt_saturation */

/* a: (n_samples) fp32 device, the normalised 16 kHz waveform already replicate-padded by the caller to a
 * multiple of seq_len * sampling_rate / fps when needed (FLOAT.py:371-373);  wa: (seq_len, dim_w) fp32.
 * Allocation-free: a clip beyond the reserved capacity is refused (FLOAT_E_INVALID) - call float_aud_reserve first. */
int float_aud_inference(float_aud_t* h, const float* a, int32_t n_samples, int32_t seq_len, float* wa,
                        void* stream);

/* Speech-to-emotion (reference Audio2Emotion.predict_emotion, FLOAT.py:378-401, on
 * Wav2Vec2ForSpeechClassification, src/nodes/models/wav2vec2_ser.py:41-118): wav2vec2 encoder on the un-interpolated
 * feature sequence, mean over time, dense -> tanh -> out_proj, softmax.  Handle created with num_labels > 0 and the
 * keys `wav2vec2.*`, `classifier.dense.*`, `classifier.out_proj.*` (prefix `emotion_encoder.wav2vec2_for_emotion.`
 * stripped).  scores: (num_labels) fp32 device. */
int float_aud_classify(float_aud_t* h, const float* a, int32_t n_samples, float* scores, void* stream);

/* ---------------------------------------------------------------- misc ------------ */
int float_hip_abi_version(void);
const char* float_last_error(void);
/* A HIP stream restricted to CUs [cu_begin, cu_end) (hipExtStreamCreateWithCUMask).  Used to run the FMT
 * chain and the decoder concurrently on disjoint CU sets (pipeline.generate(overlap="cu")). */
int float_stream_create_cu_range(int32_t cu_begin, int32_t cu_end, void** stream_out);
int float_stream_destroy(void* stream);
/* Average device time (ms) of the kernels launched by the last timed call, by class, measured
 * with hipEvents on the caller's stream when profiling is on.  which: 0 = FMT GEMMs of the step chain (weight-streaming
 * tiling), 1 = decoder convs, 2 = the once-per-window adaLN modulation GEMM of the FMT, 3 = FMT GEMMs of a stacked-clip
 * step chain (row-blocked LDS-DMA tiling, float_fmt_sample_batch).  Returns <0 when profiling is off. */
int float_set_profiling(int32_t on);
double float_profile_ms(int32_t which, int64_t* n_launches);
/* Measured peaks of the current device, printed by bench.py beside the spec-sheet peaks of its roofline objects: streaming
 * read and copy (read + write) bandwidth over 2-GiB buffers in GB/s, dense fp16 MFMA rate with register operands
 * (v_mfma_f32_16x16x32_f16 / _32x32x16_f16) in TFLOP/s, compute-unit count.  Allocates and frees 4 GiB; ~0.1 s; synchronises
 * the NULL stream.  Every pointer optional. */
int float_probe_peaks(float* hbm_read_gbps, float* hbm_copy_gbps, float* mfma16_tflops, float* mfma32_tflops, int32_t* n_cu);

#ifdef __cplusplus
}
#endif
#endif /* FLOAT_HIP_H */
