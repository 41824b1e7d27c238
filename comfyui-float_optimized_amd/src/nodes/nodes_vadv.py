"""Very-advanced (VA) nodes that drive the hot-path operators one stage at a time (reference nodes_vadv.py):
Apply FLOAT Encoder, FLOAT Get Identity Reference, Sample Motion Sequence RD, Apply FLOAT Synthesis, plus the audio
pair (feature extract + projection), which this build runs as ONE fused operator (float_aud_*).  Same class
attributes, widget names/defaults and return tuples as the reference; tensors between nodes are CPU tensors like there.
The emotion nodes run the speech-emotion operator (float_aud_classify)."""
import contextlib
import math
import os

import torch

from ... import host_models
from ...fmt import draw_noise
from ...pipeline import report_range
from . import EMOTIONS, TORCHDIFFEQ_FIXED_STEP_SOLVERS
from . import main_logger as logger
from .nodes_vadv_loader import BASE_CATEGORY, build_audio_encoder
from .options.base_options import BaseOptions

SUFFIX = "(VA)"


@contextlib.contextmanager
def model_to_target(*ops):
    """The reference wraps every VA node's model call in `with model_to_target(logger, model)` (nodes_vadv.py:107,192,275,347,
    437,520,697,807): on the target device inside, back on the host after.  Here the operators stay resident by default (288 GB
    of HBM; a rebuild costs seconds of host-side packing); FLOAT_AMD_OFFLOAD=always releases every handle after the call and
    the next call rebuilds it from the host weights the loader object keeps (native.rebuildable) - bitwise the same results."""
    for op in ops:
        op.to_target()
    try:
        yield
    finally:
        if os.environ.get("FLOAT_AMD_OFFLOAD", "never").lower() == "always":
            for op in ops:
                torch.cuda.current_stream(op.device).synchronize()
                op.offload()


class ApplyFloatEncoder:
    UNIQUE_NAME = "ApplyFloatEncoder"
    DISPLAY_NAME = "Apply FLOAT Encoder"
    DESCRIPTION = "Encodes a batch of reference images: appearance pipe {h_source, feats} and the motion coefficients r_s_lambda."
    CATEGORY = BASE_CATEGORY

    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {"ref_image": ("IMAGE", {}), "float_encoder": ("FLOAT_ENCODER_MODEL", {})}}

    RETURN_TYPES = ("FLOAT_APPEARANCE_PIPE", "TORCH_TENSOR", "FLOAT_ENCODER_MODEL")
    RETURN_NAMES = ("appearance_pipe (Ws→r)", "r_s_lambda_latent", "float_encoder_out")
    FUNCTION = "apply_encoder"

    def apply_encoder(self, ref_image, float_encoder):
        size, nc = float_encoder.inferred_input_size, BaseOptions().input_nc
        if not isinstance(ref_image, torch.Tensor):
            raise TypeError("Input 'ref_image' must be a torch.Tensor.")
        if ref_image.ndim != 4:
            raise ValueError("Input 'ref_image' is %dD, must be 4D (B, H, W, C)." % ref_image.ndim)
        _, hh, ww, ch = ref_image.shape
        if hh != size or ww != size:
            raise ValueError("Image size %dx%d does not match Encoder's inferred input_size %d." % (hh, ww, size))
        if ch != nc:
            raise ValueError("Image channels %d does not match expected input_nc %d." % (ch, nc))
        s = ref_image.permute(0, 3, 1, 2).contiguous() * 2.0 - 1.0  # nodes_vadv.py:345-347
        s_r, lam, feats = [], [], None
        with model_to_target(float_encoder):
            for b in range(s.shape[0]):  # the operator's batch is 1 (include/float_hip.h)
                sb, lb, fb, _ = float_encoder.encode_image_into_latent(s[b])
                s_r.append(sb)
                lam.append(lb)
                feats = [[f] for f in fb] if feats is None else [acc + [f] for acc, f in zip(feats, fb)]
            pipe = {"h_source": torch.cat(s_r).cpu(), "feats": [torch.cat(f).cpu() for f in feats]}  # .cpu() synchronises
            lam = torch.cat(lam).cpu()
            if float_encoder.dtype == "fp16":
                report_range({"encoder": float_encoder.saturation(reset=True)}, "ApplyFloatEncoder")
        return (pipe, lam, float_encoder)


class FloatGetIdentityReferenceVA:
    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {"r_s_lambda_latent": ("TORCH_TENSOR", {}), "float_synthesis": ("FLOAT_SYNTHESIS_MODEL", {})}}

    RETURN_TYPES = ("FLOAT_SYNTHESIS_MODEL", "TORCH_TENSOR")
    RETURN_NAMES = ("float_synthesis_out", "r_s_latent (Wr→s)")
    FUNCTION = "get_identity_reference_batch"
    CATEGORY = BASE_CATEGORY
    DESCRIPTION = "r_s = Direction(r_s_lambda): the identity-specific motion reference latent."
    UNIQUE_NAME = "FloatGetIdentityReferenceVA"
    DISPLAY_NAME = "FLOAT Get Identity Reference"

    def get_identity_reference_batch(self, r_s_lambda_latent, float_synthesis):
        if not isinstance(r_s_lambda_latent, torch.Tensor):
            raise TypeError("Input 'r_s_lambda_latent' must be a torch.Tensor, got %s" % type(r_s_lambda_latent))
        if r_s_lambda_latent.ndim != 2:
            raise ValueError("Input 'r_s_lambda_latent' must be a 2D tensor (Batch, DimM), got %dD." % r_s_lambda_latent.ndim)
        if r_s_lambda_latent.shape[1] != float_synthesis.inferred_motion_dim:
            raise ValueError("Dimension 1 of 'r_s_lambda_latent' should be (%d), got %d."
                             % (float_synthesis.inferred_motion_dim, r_s_lambda_latent.shape[1]))
        with model_to_target(float_synthesis):
            r_s = float_synthesis.direction(r_s_lambda_latent).cpu()
        return (float_synthesis, r_s)


class FloatSampleMotionSequenceRD_VA:
    UNIQUE_NAME = "FloatSampleMotionSequenceRD_VA"
    DISPLAY_NAME = "Sample Motion Sequence RD"
    DESCRIPTION = "Runs the FMT ODE sampling loop (windows of wav2vec_sec*fps frames with num_prev_frames of context) -> r_d latents."
    CATEGORY = BASE_CATEGORY

    @classmethod
    def INPUT_TYPES(cls):
        o = BaseOptions()
        f01 = {"min": 0.0, "max": 10.0, "step": 0.1}
        tol = {"min": 1e-9, "max": 1e-1, "step": 1e-6, "precision": 9}
        prob = {"min": 0.0, "max": 1.0, "step": 0.01}
        return {"required": {
            "r_s_latent": ("TORCH_TENSOR", {}),
            "wa_latent": ("TORCH_TENSOR", {}),
            "audio_num_frames": ("INT", {"forceInput": True}),
            "we_latent": ("TORCH_TENSOR", {}),
            "float_fmt_model": ("FLOAT_FMT_MODEL", {}),
            "a_cfg_scale": ("FLOAT", dict(default=o.a_cfg_scale, **f01)),
            "r_cfg_scale": ("FLOAT", dict(default=o.r_cfg_scale, **f01)),
            "e_cfg_scale": ("FLOAT", dict(default=o.e_cfg_scale, **f01)),
            "include_r_cfg": ("BOOLEAN", {"default": False}),
            "nfe": ("INT", {"default": o.nfe, "min": 1, "max": 1000}),
            "torchdiffeq_ode_method": (TORCHDIFFEQ_FIXED_STEP_SOLVERS, {"default": o.torchdiffeq_ode_method}),
            "ode_atol": ("FLOAT", dict(default=o.ode_atol, **tol)),
            "ode_rtol": ("FLOAT", dict(default=o.ode_rtol, **tol)),
            "audio_dropout_prob": ("FLOAT", dict(default=o.audio_dropout_prob, **prob)),
            "ref_dropout_prob": ("FLOAT", dict(default=o.ref_dropout_prob, **prob)),
            "emotion_dropout_prob": ("FLOAT", dict(default=o.emotion_dropout_prob, **prob)),
            "fix_noise_seed": ("BOOLEAN", {"default": o.fix_noise_seed}),
            "seed": ("INT", {"default": o.seed, "min": 0, "max": 0xffffffffffffffff}),
        }}

    RETURN_TYPES = ("TORCH_TENSOR", "FLOAT_FMT_MODEL")
    RETURN_NAMES = ("r_d_latents (Wr→D)", "float_fmt_model_out")
    FUNCTION = "sample_rd_sequence_va"

    def sample_rd_sequence_va(self, r_s_latent, wa_latent, we_latent, audio_num_frames, float_fmt_model, a_cfg_scale, r_cfg_scale,
                              e_cfg_scale, include_r_cfg, nfe, torchdiffeq_ode_method, ode_atol, ode_rtol, audio_dropout_prob,
                              ref_dropout_prob, emotion_dropout_prob, fix_noise_seed, seed):
        if not all(isinstance(t, torch.Tensor) for t in (r_s_latent, wa_latent, we_latent)):
            raise TypeError("All latent inputs must be torch.Tensors.")
        B = wa_latent.shape[0]
        if not (r_s_latent.shape[0] == B and we_latent.shape[0] == B):
            raise ValueError("Batch size mismatch among r_s, wa, we latents.")
        if wa_latent.shape[1] != audio_num_frames:
            logger.warning("wa_latent time dim (%d) != audio_num_frames (%d).", wa_latent.shape[1], audio_num_frames)
        # dropout probabilities are inert at inference (FMT.py:271-275 with train=False); atol/rtol do not act on a
        # fixed-grid solver: accepted for graph compatibility
        fmt, cfg = float_fmt_model, float_fmt_model.cfg
        T = wa_latent.shape[1]
        n_chunks = int(math.ceil(T / cfg.num_frames_for_clip))
        dev = fmt.device
        # one generator for the whole batch, seeded once, drawn chunk by chunk on the target device
        # (nodes_vadv.py:678-695, nodes_adv.py:578-612); without a seed the global RNG is used
        if fix_noise_seed or seed != BaseOptions().seed:
            noise = draw_noise(n_chunks, B, cfg, seed, device=dev)
        else:
            noise = torch.stack([torch.randn(B, cfg.num_frames_for_clip, cfg.dim_w, device=dev) for _ in range(n_chunks)])
        with model_to_target(fmt):
            fmt.set_method(torchdiffeq_ode_method)  # a rebuilt handle starts from the default solver
            r_d = fmt.sample(r_s_latent, wa_latent, we_latent, noise, nfe, a_cfg_scale, r_cfg_scale, e_cfg_scale, include_r_cfg)
            r_d_cpu = r_d.cpu()  # synchronises
            if fmt.dtype == "fp16":
                report_range({"fmt": fmt.saturation(reset=True)}, "FloatSampleMotionSequenceRD_VA")
        return (r_d_cpu, fmt)


class ApplyFloatSynthesis:
    UNIQUE_NAME = "ApplyFloatSynthesis"
    DISPLAY_NAME = "Apply FLOAT Synthesis"
    DESCRIPTION = "Decodes r_d latents into frames with the appearance pipe of Apply FLOAT Encoder."
    CATEGORY = BASE_CATEGORY

    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {"appearance_pipe": ("FLOAT_APPEARANCE_PIPE", {}), "float_synthesis": ("FLOAT_SYNTHESIS_MODEL", {}),
                             "r_d_latents": ("TORCH_TENSOR", {})}}

    RETURN_TYPES = ("IMAGE", "FLOAT_SYNTHESIS_MODEL")
    RETURN_NAMES = ("images", "float_synthesis_out")
    FUNCTION = "apply_synthesis"

    def apply_synthesis(self, appearance_pipe, float_synthesis, r_d_latents):
        try:
            s_r, feats = appearance_pipe["h_source"], appearance_pipe["feats"]
        except KeyError as e:
            raise KeyError("Input 'appearance_pipe' is missing an expected key: %s." % e)
        if not all(isinstance(t, torch.Tensor) for t in (s_r, r_d_latents)):
            raise TypeError("s_r_latent and r_d_latents must be torch.Tensors.")
        if not (isinstance(feats, list) and all(isinstance(t, torch.Tensor) for t in feats)):
            raise TypeError("appearance_pipe['feats'] must be a list of Tensors.")
        B = s_r.shape[0]
        if not (r_d_latents.shape[0] == B and all(f.shape[0] == B for f in feats)):
            raise ValueError("Batch size mismatch in inputs for Synthesis.")
        size = float_synthesis.inferred_size
        if r_d_latents.shape[1] == 0:
            return (torch.empty((0, size, size, BaseOptions().input_nc), dtype=torch.float32), float_synthesis)
        # nodes_vadv.py:437-462: frame t of item b decodes s_r[b] + r_d[b, t].  The frames of every item land in ONE pinned host
        # tensor through float_dec_frames_host (the reference's pre-allocated CPU tensor, FLOAT.py:139), item b at rows b*T..
        T = r_d_latents.shape[1]
        host = torch.empty((B * T, size, size, BaseOptions().input_nc), dtype=torch.float32, pin_memory=True)
        staging = None
        with model_to_target(float_synthesis):
            for b in range(B):
                float_synthesis.set_feats([f[b:b + 1] for f in feats])
                staging = float_synthesis.decode_into_host(s_r[b:b + 1], r_d_latents[b], host[b * T:(b + 1) * T], staging)
            torch.cuda.current_stream(float_synthesis.device).synchronize()
            if float_synthesis.dtype == "fp16":  # an fp16 decoder that left its range must not hand over black regions silently
                report_range({"decoder": float_synthesis.saturation(reset=True)}, "ApplyFloatSynthesis")
        del staging
        return (host, float_synthesis)


class FloatAudioPreprocessAndFeatureExtract:
    UNIQUE_NAME = "FloatAudioPreprocessAndFeatureExtract"
    DISPLAY_NAME = "FLOAT Audio Feature Extract"
    DESCRIPTION = ("Validates and normalises mono audio at the wav2vec2 rate.  In this build the wav2vec2 features are not "
                   "materialised: the returned `wav2vec_features` is a deferred handle that FLOAT Apply Audio Projection "
                   "turns into wa with the fused operator (float_aud_*).")
    CATEGORY = BASE_CATEGORY

    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {
            "audio": ("AUDIO", {}),
            "wav2vec_pipe": ("WAV2VEC_PIPE", {}),
            "target_fps": ("FLOAT", {"default": 25.0, "min": 1.0, "step": 0.1}),
            "only_last_features": ("BOOLEAN", {"default": False}),
        }}

    RETURN_TYPES = ("TORCH_TENSOR", "INT", "TORCH_TENSOR", "WAV2VEC_PIPE", "AUDIO", "FLOAT")
    RETURN_NAMES = ("wav2vec_features", "audio_num_frames", "processed_audio_features", "wav2vec_pipe_out", "audio", "fps")
    FUNCTION = "extract_features_with_custom_model"

    def extract_features_with_custom_model(self, audio, wav2vec_pipe, target_fps, only_last_features):
        if not isinstance(wav2vec_pipe, dict) or "state" not in wav2vec_pipe:
            raise TypeError("wav2vec_pipe is not in the expected format (Load Wav2Vec Model output).")
        if not isinstance(audio, dict) or "waveform" not in audio or "sample_rate" not in audio:
            raise TypeError("Input 'audio' must be a ComfyUI AUDIO dictionary.")
        w, sr, want = audio["waveform"], audio["sample_rate"], wav2vec_pipe["expected_sr"]
        if sr != want:
            raise ValueError("Input audio SR (%d Hz) != expected SR (%d Hz). Resample upstream." % (sr, want))
        if w.ndim == 2:
            w = w.unsqueeze(1)
        elif w.ndim == 3:
            if w.shape[1] != 1:
                raise ValueError("Input audio must be mono.")
        else:
            raise ValueError("audio['waveform'] must be 2D or 3D.")
        a = torch.cat([host_models.preprocess_audio(w[b], sr, want) for b in range(w.shape[0])])  # zero-mean / unit-variance per item
        n_frames = math.ceil(a.shape[1] * target_fps / want)
        deferred = DeferredWav2VecFeatures(a, n_frames, wav2vec_pipe, target_fps, only_last_features)
        return (deferred, n_frames, a.cpu(), wav2vec_pipe, audio, target_fps)


class DeferredWav2VecFeatures:
    """Stand-in for the (B, T, layers*hidden) wav2vec2 feature tensor between the two audio nodes."""

    def __init__(self, audio, n_frames, pipe, fps, only_last):
        self.audio, self.n_frames, self.pipe, self.fps, self.only_last = audio, n_frames, pipe, fps, only_last
        c = pipe["config"]
        self.shape = (audio.shape[0], n_frames, c.hidden_size if only_last else c.hidden_size * c.num_hidden_layers)
        self.ndim = 3


class FloatApplyAudioProjection:
    UNIQUE_NAME = "FloatApplyAudioProjection"
    DISPLAY_NAME = "FLOAT Apply Audio Projection"
    DESCRIPTION = "wav2vec2 (all hidden states) + audio projection as one HIP operator -> wa_latent (B, T, dim_a)."
    CATEGORY = BASE_CATEGORY

    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {"wav2vec_features": ("TORCH_TENSOR", {}), "projection_layer": ("AUDIO_PROJECTION_LAYER", {})}}

    RETURN_TYPES = ("TORCH_TENSOR",)
    RETURN_NAMES = ("wa_latent",)
    FUNCTION = "apply_projection"

    def apply_projection(self, wav2vec_features, projection_layer):
        if not isinstance(wav2vec_features, DeferredWav2VecFeatures):
            raise TypeError("Input 'wav2vec_features' must come from FLOAT Audio Feature Extract of this build "
                            "(the features are produced and projected by one fused operator).")
        if wav2vec_features.shape[2] != projection_layer["inferred_input_feature_dim"]:
            raise TypeError("Input 'wav2vec_features' wrong size has %d, expected %d. `only_last_features` mismatch?"
                            % (wav2vec_features.shape[2], projection_layer["inferred_input_feature_dim"]))
        key = (id(wav2vec_features.pipe), wav2vec_features.only_last)
        cache = projection_layer.setdefault("_encoders", {})
        if key not in cache:
            cache[key] = build_audio_encoder(wav2vec_features.pipe, projection_layer, wav2vec_features.only_last)
        enc = cache[key]
        with model_to_target(enc):
            enc.fps = wav2vec_features.fps
            wa = enc.inference(wav2vec_features.audio, wav2vec_features.n_frames).cpu()
        return (wa,)


class FloatExtractEmotionWithCustomModel:
    UNIQUE_NAME = "FloatExtractEmotionWithCustomModel"
    DISPLAY_NAME = "FLOAT Extract Emotion from Features"
    DESCRIPTION = "Emotion conditioning we (B,1,num_labels): predicted from the normalised audio, or one-hot for a named emotion."
    CATEGORY = BASE_CATEGORY

    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {
            "processed_audio_features": ("TORCH_TENSOR", {}),
            "emotion_model_pipe": ("EMOTION_MODEL_PIPE", {}),
            "emotion": (EMOTIONS, {"default": "none"}),
        }}

    RETURN_TYPES = ("TORCH_TENSOR", "EMOTION_MODEL_PIPE")
    RETURN_NAMES = ("we_latent", "emotion_model_pipe_out")
    FUNCTION = "extract_emotion_from_features"

    def extract_emotion_from_features(self, processed_audio_features, emotion_model_pipe, emotion):
        if not isinstance(emotion_model_pipe, tuple) or len(emotion_model_pipe) != 3:
            raise TypeError("emotion_model_pipe is not in the expected format (model, feature_extractor_ref, config_dict).")
        model, _, info = emotion_model_pipe
        if not isinstance(processed_audio_features, torch.Tensor):
            raise TypeError("Input 'processed_audio_features' must be a torch.Tensor.")
        if processed_audio_features.ndim != 2:
            raise ValueError("Input 'processed_audio_features' must be a 2D tensor (Batch, NumSamplesAfterPrep), got %dD with shape %s."
                             % (processed_audio_features.ndim, tuple(processed_audio_features.shape)))
        B, n = processed_audio_features.shape[0], info.get("num_labels")
        if n is None:
            raise ValueError("Number of labels (num_labels) not found in emotion model config from pipe.")
        name = str(emotion).lower()
        idx = (info.get("label2id") or {}).get(name) if name != "none" else None
        if name != "none" and idx is None:  # nodes_vadv.py:265-271: unknown name -> predict from the audio
            logger.warning("Specified emotion '%s' not found in the emotion model's label2id map. Predicting from audio instead.", name)
        if idx is None:
            with model_to_target(model):
                we = model.predict_emotion(processed_audio_features).unsqueeze(1).cpu()
        else:
            we = torch.nn.functional.one_hot(torch.tensor(idx), num_classes=n).float()[None, None].repeat(B, 1, 1)
        return (we.cpu(), emotion_model_pipe)


class FloatExtractEmotionWithCustomModelDyn:
    UNIQUE_NAME = "FloatExtractEmotionWithCustomModelDyn"
    DISPLAY_NAME = "FLOAT Extract Emotion (Dynamic)"
    DESCRIPTION = "Per-chunk speech-emotion scores, nearest-neighbour up-sampled to one vector per video frame (dynamic we)."
    CATEGORY = BASE_CATEGORY

    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {
            "audio": ("AUDIO", {}),
            "emotion_model_pipe": ("EMOTION_MODEL_PIPE", {}),
            "target_fps": ("FLOAT", {"default": 25.0, "min": 1.0, "max": 120.0, "step": 0.1}),
            "chunk_duration_sec": ("FLOAT", {"default": 2.0, "min": 0.5, "max": 10.0, "step": 0.1}),
        }}

    RETURN_TYPES = ("TORCH_TENSOR", "EMOTION_MODEL_PIPE", "TORCH_TENSOR")
    RETURN_NAMES = ("we_latent_dynamic", "emotion_model_pipe_out", "emotion_sequence")
    FUNCTION = "extract_dynamic_emotion"

    def extract_dynamic_emotion(self, audio, emotion_model_pipe, target_fps, chunk_duration_sec):
        model, normalise, info = emotion_model_pipe
        want = info.get("sampling_rate", 16000)
        if not isinstance(audio, dict) or "waveform" not in audio or "sample_rate" not in audio:
            raise TypeError("Input 'audio' must be a ComfyUI AUDIO dictionary.")
        w, sr = audio["waveform"], audio["sample_rate"]
        if sr != want:
            raise ValueError("Input audio SR (%d) must match emotion model's expected SR (%d). Please resample upstream." % (sr, want))
        if w.ndim == 2:
            w = w.unsqueeze(1)
        if w.shape[1] != 1:
            raise ValueError("Input audio must be mono (1 channel).")
        B, total = w.shape[0], w.shape[-1]
        chunk = int(chunk_duration_sec * sr)
        if chunk == 0:
            raise ValueError("Chunk duration is too small for the sample rate.")
        n_chunks = math.ceil(total / chunk)
        seq = []
        with model_to_target(model):
            for b in range(B):  # nodes_vadv.py:812-826: every chunk is normalised on its own, then classified
                for i in range(n_chunks):
                    piece = normalise(w[b, :, i * chunk:(i + 1) * chunk], sr, want)
                    if piece.shape[1] < 400:  # shorter than the feature extractor's receptive field: pad by replication
                        piece = torch.nn.functional.pad(piece[:, None], (0, 400 - piece.shape[1]), mode="replicate")[:, 0]
                    seq.append(model.predict_emotion(piece))
            seq = torch.cat(seq, dim=0).view(B, n_chunks, info["num_labels"]).cpu()
        T = math.ceil(total / sr * target_fps)
        if n_chunks > 1:  # nearest-neighbour up-sampling (nodes_vadv.py:835-838)
            we = torch.nn.functional.interpolate(seq.transpose(1, 2), size=T, mode="nearest").transpose(1, 2)
        else:
            we = seq.repeat(1, T, 1)
        return (we, emotion_model_pipe, seq)
