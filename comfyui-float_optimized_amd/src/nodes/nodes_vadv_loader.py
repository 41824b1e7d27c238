"""Very-advanced (VA) loaders of the split checkpoint layout (reference nodes_vadv_loader.py): each part of the
model is its own .safetensors under `models/float/...` (utils/downloader.py:20-42), its architecture is INFERRED
from tensor shapes, and the loaded object is the MI355X HIP operator instead of an nn.Module.  Same class
attributes, widget names/defaults and return tuples as the reference; the bodies build EncoderHIP / SynthesisHIP /
FlowMatchingTransformerHIP / AudioEncoderHIP / Audio2EmotionHIP."""
import ast
import os
import re

import torch

from ...audio import Audio2EmotionHIP, AudioEncoderHIP
from ...config import AudioConfig, FmtConfig, emotion_audio_config
from ...decoder import SynthesisHIP
from ...encoder import EncoderHIP
from ...fmt import FlowMatchingTransformerHIP
from ...weights import ENC_CHANNELS
from . import main_logger as logger
from .nodes import _device_options
from .options.base_options import BaseOptions

SUFFIX = "(VA)"
BASE_CATEGORY = "FLOAT/Very Advanced"
FILE_CATEGORY = BASE_CATEGORY + "/Loaders"
MOTION_AE_DIR = "float/motion_autoencoder"   # nodes_vadv_loader.py:30
FMT_SUBDIR = "float/fmt"                     # :31
AUDIO_PROJ_DIR = "float/audio_projections"
WAV2VEC_DIR = "audio"
# key prefixes of the parts inside the unified FLOAT.safetensors (utils/downloader.py:35-42)
EXTRACTION_PREFIXES = {
    "encoder": "motion_autoencoder.enc",
    "decoder": "motion_autoencoder.dec",
    "projection": "audio_encoder.audio_projection",
    "fmt": "fmt",
    "wav2vec2_base": "audio_encoder.wav2vec2",
    "emotion_ser": "emotion_encoder.wav2vec2_for_emotion",
}


def models_dir():
    try:
        import folder_paths
        return folder_paths.models_dir
    except Exception:
        return os.environ.get("FLOAT_MODELS_DIR", os.path.join(os.path.expanduser("~"), "ComfyUI", "models"))


def unified_model_path():
    return os.path.join(models_dir(), "float", "FLOAT.safetensors")


def look_for_models(sub_dir, default_name, dirs=False):
    d = os.path.join(models_dir(), sub_dir)
    found = []
    if os.path.isdir(d):
        found = sorted(f for f in os.listdir(d) if (os.path.isdir(os.path.join(d, f)) if dirs else f.endswith(".safetensors")))
    return found or [default_name]


def extract_part(unified_path, part_key, dst_path):
    """Write one part of FLOAT.safetensors as its own file with the prefix stripped (utils/downloader.py:60-120)."""
    from safetensors.torch import load_file, save_file
    pre = EXTRACTION_PREFIXES[part_key] + "."
    sd = {k[len(pre):]: v.contiguous() for k, v in load_file(unified_path, device="cpu").items() if k.startswith(pre)}
    if not sd:
        raise KeyError("no '%s*' tensors in %s" % (pre, unified_path))
    os.makedirs(os.path.dirname(dst_path), exist_ok=True)
    save_file(sd, dst_path)
    return dst_path


def ensure_model_part_exists(part_key, sub_dir, file_name):
    """The part file if present; else extracted from the unified model; there is no download in this build."""
    path = os.path.join(models_dir(), sub_dir, file_name)
    if os.path.exists(path):
        return path
    if os.path.exists(unified_model_path()):
        logger.info("extracting '%s' from %s", part_key, unified_model_path())
        try:
            return extract_part(unified_model_path(), part_key, path)
        except KeyError as e:
            raise FileNotFoundError("%s weights file not found: %s (%s)" % (part_key, path, e)) from e
    raise FileNotFoundError("%s weights file not found: %s (and no unified FLOAT.safetensors to extract it from)" % (part_key, path))


def _load_sd(path):
    from safetensors.torch import load_file
    return load_file(path, device="cpu")


def safe_parse_list_str(list_str, expected_type=int):
    try:
        v = ast.literal_eval(list_str)
    except Exception as e:
        raise ValueError("not a Python list literal: %r" % (list_str,)) from e
    if not isinstance(v, (list, tuple)) or not all(isinstance(x, expected_type) for x in v):
        raise ValueError("expected a list of %s, got %r" % (expected_type.__name__, list_str))
    return list(v)


_INV_CHANNELS = {}
for _size, _ch in ENC_CHANNELS.items():  # largest size that uses a channel count (nodes_vadv_loader.py:350-357)
    _INV_CHANNELS[_ch] = max(_INV_CHANNELS.get(_ch, 0), _size)


class LoadFloatEncoderModel:
    UNIQUE_NAME = "LoadFloatEncoderModel"
    DISPLAY_NAME = "Load FLOAT Encoder"
    DESCRIPTION = "Loads motion_autoencoder/encoder.safetensors; input size, dim_w and dim_m are inferred from the weights."
    DEFAULT_ENCODER_FILENAME = "encoder.safetensors"
    CATEGORY = FILE_CATEGORY

    @classmethod
    def INPUT_TYPES(cls):
        device_options, default_device = _device_options()
        return {"required": {
            "encoder_file": (look_for_models(MOTION_AE_DIR, cls.DEFAULT_ENCODER_FILENAME), {}),
            "target_device": (device_options, {"default": default_device}),
            "cudnn_benchmark": ("BOOLEAN", {"default": False}),
        }}

    RETURN_TYPES = ("INT", "INT", "INT", "FLOAT_ENCODER_MODEL")
    RETURN_NAMES = ("inferred_input_size", "dim_w", "dim_m", "float_encoder")
    FUNCTION = "load_encoder_infer_arch"

    def load_encoder_infer_arch(self, encoder_file, target_device, cudnn_benchmark):
        sd = _load_sd(ensure_model_part_exists("encoder", MOTION_AE_DIR, encoder_file))
        for key in ("fc.4.weight", "fc.0.weight", "net_app.convs.0.0.weight"):  # nodes_vadv_loader.py:419-447
            if key not in sd:
                raise KeyError("Key '%s' not found for encoder architecture inference." % key)
        dim_m, dim_w = sd["fc.4.weight"].shape[0], sd["fc.0.weight"].shape[0]
        c0 = sd["net_app.convs.0.0.weight"].shape[0]
        if c0 not in _INV_CHANNELS:
            raise ValueError("Cannot infer input_size: Out channels (%d) not in the channel map %s" % (c0, _INV_CHANNELS))
        size = _INV_CHANNELS[c0]
        enc = EncoderHIP(sd, size, dim_w, dim_m, target_device, dtype=os.environ.get("FLOAT_AMD_DEC_DTYPE", "fp16"))
        enc.inferred_input_size, enc.dim_w, enc.dim_m = size, dim_w, dim_m
        enc.cudnn_benchmark_setting = cudnn_benchmark  # kept for graph compatibility; MIOpen is not on this path
        enc.target_device = torch.device(target_device)
        return (size, dim_w, dim_m, enc)


class LoadFloatSynthesisModel:
    UNIQUE_NAME = "LoadFloatSynthesisModel"
    DISPLAY_NAME = "Load FLOAT Synthesis"
    DESCRIPTION = "Loads motion_autoencoder/decoder.safetensors; size, style_dim and motion_dim are inferred from the weights."
    DEFAULT_SYNTHESIS_FILENAME = "decoder.safetensors"
    CATEGORY = FILE_CATEGORY

    @classmethod
    def INPUT_TYPES(cls):
        device_options, default_device = _device_options()
        return {"required": {
            "synthesis_file": (look_for_models(MOTION_AE_DIR, cls.DEFAULT_SYNTHESIS_FILENAME), {}),
            "target_device": (device_options, {"default": default_device}),
            "channel_multiplier": ("INT", {"default": 1, "min": 1, "max": 8}),
            "blur_kernel_str": ("STRING", {"default": "[1, 3, 3, 1]"}),
            "cudnn_benchmark": ("BOOLEAN", {"default": False}),
        }}

    RETURN_TYPES = ("FLOAT_SYNTHESIS_MODEL", "INT", "INT", "INT")
    RETURN_NAMES = ("float_synthesis", "inferred_size", "inferred_style_dim", "inferred_motion_dim")
    FUNCTION = "load_synthesis_infer_arch"

    def load_synthesis_infer_arch(self, synthesis_file, target_device, channel_multiplier, blur_kernel_str, cudnn_benchmark):
        path = ensure_model_part_exists("decoder", MOTION_AE_DIR, synthesis_file)
        try:
            blur_kernel = safe_parse_list_str(blur_kernel_str, int)
        except ValueError as e:
            raise ValueError("Invalid blur_kernel_str format: %s. Must be Python list syntax e.g. '[1,3,3,1]'" % e)
        if len(blur_kernel) != 4:  # the Blur's padding follows the tap count (styledecoder.py:209-213); the operator has the 4-tap form
            raise ValueError("the HIP decoder implements 4-tap blur kernels (the released checkpoints use [1,3,3,1]); got %s" % (blur_kernel,))
        sd = _load_sd(path)
        # channel_multiplier: the operator reads every level's channel count off the weights (like the reference's
        # load_state_dict would fail on a mismatch, nodes_vadv_loader.py:567-611, a widget that contradicts the file is refused)
        w512 = sd.get("convs.%d.conv.weight" % (2 * 3))  # 64-px level: 256 * channel_multiplier output channels
        if w512 is not None and w512.shape[1] != 256 * channel_multiplier:
            raise ValueError("channel_multiplier=%d does not match the checkpoint (64-px level has %d channels = multiplier %g)"
                             % (channel_multiplier, w512.shape[1], w512.shape[1] / 256.0))
        for key in ("conv1.conv.modulation.weight", "direction.weight"):  # nodes_vadv_loader.py:574-589
            if key not in sd:
                raise KeyError("Key '%s' for synthesis architecture inference not found." % key)
        style_dim = sd["conv1.conv.modulation.weight"].shape[1]
        motion_dim = sd["direction.weight"].shape[1]
        n_rgb = 0
        for k in sd:  # to_rgbs.0 .. to_rgbs.(log_size-3)  (nodes_vadv_loader.py:599-617)
            m = re.match(r"to_rgbs\.(\d+)\.conv\.0\.weight$", k)
            if m:
                n_rgb = max(n_rgb, int(m.group(1)) + 1)
        if n_rgb == 0:
            raise ValueError("Could not determine number of to_rgb layers to infer size.")
        size = 2 ** (n_rgb + 2)
        dec = SynthesisHIP(sd, size, style_dim, target_device, dtype=os.environ.get("FLOAT_AMD_DEC_DTYPE", "fp16"),
                           max_frames=int(os.environ.get("FLOAT_AMD_DEC_BATCH", "32")), blur_kernel=blur_kernel)
        dec.inferred_size, dec.inferred_style_dim, dec.inferred_motion_dim = size, style_dim, motion_dim
        dec.channel_multiplier_setting, dec.blur_kernel_setting = channel_multiplier, blur_kernel
        dec.cudnn_benchmark_setting = cudnn_benchmark
        dec.target_device = torch.device(target_device)
        return (dec, size, style_dim, motion_dim)


class LoadFMTModel:
    UNIQUE_NAME = "LoadFMTModel"
    DISPLAY_NAME = "Load FLOAT FMT Model"
    DESCRIPTION = ("Loads fmt/fmt.safetensors; dim_h, dim_w, dim_a, depth and mlp_ratio are inferred, the temporal structure "
                   "(fps, wav2vec_sec, num_prev_frames) and attention window come from the widgets.")
    DEFAULT_FMT_FILENAME = "fmt.safetensors"
    CATEGORY = FILE_CATEGORY

    @classmethod
    def INPUT_TYPES(cls):
        device_options, default_device = _device_options()
        o = BaseOptions()
        return {"required": {
            "fmt_file": (look_for_models(FMT_SUBDIR, cls.DEFAULT_FMT_FILENAME), {}),
            "target_device": (device_options, {"default": default_device}),
            "cudnn_benchmark": ("BOOLEAN", {"default": False}),
            "dim_e": ("INT", {"default": o.dim_e, "min": 1, "max": 100}),
            "num_heads": ("INT", {"default": o.num_heads, "min": 1, "max": 32}),
            "attention_window": ("INT", {"default": o.attention_window, "min": 1, "max": 20}),
            "num_prev_frames": ("INT", {"default": o.num_prev_frames, "min": 0, "max": 100}),
            "fps": ("FLOAT", {"default": o.fps, "min": 1.0, "max": 120.0, "step": 0.1}),
            "wav2vec_sec": ("FLOAT", {"default": o.wav2vec_sec, "min": 0.1, "max": 10.0, "step": 0.1}),
        }}

    RETURN_TYPES = ("FLOAT_FMT_MODEL", "FLOAT", "ADV_FLOAT_DICT", "INT")
    RETURN_NAMES = ("float_fmt_model", "fps", "fmt_options_out", "conditioning_chunk_size")
    FUNCTION = "load_fmt_model"

    def load_fmt_model(self, fmt_file, target_device, cudnn_benchmark, dim_e, num_heads, attention_window, num_prev_frames, fps,
                       wav2vec_sec):
        sd = _load_sd(ensure_model_part_exists("fmt", FMT_SUBDIR, fmt_file))
        # structural parameters from tensor shapes (nodes_vadv_loader.py:741-781)
        dim_h, dim_w = sd["x_embedder.proj.weight"].shape
        depth = 1 + max([int(m.group(1)) for m in (re.match(r".*blocks\.(\d+)\..*", k) for k in sd) if m], default=-1)
        if depth == 0:
            raise KeyError("Could not find FMT blocks to infer fmt_depth.")
        mlp_ratio = sd["blocks.0.mlp.fc1.weight"].shape[0] / dim_h
        dim_a = sd["c_embedder.weight"].shape[1] - dim_w - dim_e
        if dim_a <= 0:
            raise ValueError("Inferred dim_a (%d) is not positive. Check c_embedder weights, inferred_dim_w_for_x_embedder (%d), "
                             "or dim_e (%d)." % (dim_a, dim_w, dim_e))
        opt = BaseOptions()
        opt.rank = str(target_device)
        opt.dim_h, opt.fmt_depth, opt.mlp_ratio, opt.dim_w, opt.dim_a, opt.dim_e = dim_h, depth, mlp_ratio, dim_w, dim_a, dim_e
        opt.num_heads, opt.attention_window, opt.num_prev_frames, opt.fps, opt.wav2vec_sec = (num_heads, attention_window,
                                                                                             num_prev_frames, fps, wav2vec_sec)
        n_total = num_prev_frames + int(wav2vec_sec * fps)
        if "pos_embed" in sd:  # nodes_vadv_loader.py:806-820
            if sd["pos_embed"].shape[2] != dim_h:
                raise ValueError("Saved 'pos_embed' hidden dim (%d) conflicts with inferred/set opt.dim_h (%d)."
                                 % (sd["pos_embed"].shape[2], dim_h))
            if sd["pos_embed"].shape[1] != n_total:
                logger.warning("Saved 'pos_embed' is for %d total frames, the options give %d.", sd["pos_embed"].shape[1], n_total)
        # like the reference, pos_embed / alignment_mask of the file are NOT loaded: they are regenerated from the
        # widgets (nodes_vadv_loader.py:822-840), so attention_window / num_prev_frames / fps / wav2vec_sec steer the model
        sd = {k: v for k, v in sd.items() if k not in ("pos_embed", "alignment_mask")}
        cfg = FmtConfig.from_options(opt)
        # the VA sampler takes batches (nodes_vadv.py:618-735): up to FLOAT_AMD_FMT_MAX_BATCH clips share one launch chain.
        # Default 1 (one clip per chain, larger batches are cut into groups: 3.1 GB of workspace); every further clip of the
        # stack costs another 3.1 GB at the default shape whether or not a workflow ever batches, so stacking is opt-in
        fmt = FlowMatchingTransformerHIP(sd, cfg, target_device, dtype=os.environ.get("FLOAT_AMD_FMT_DTYPE", "fp16"),
                                         max_batch=max(1, min(16, int(os.environ.get("FLOAT_AMD_FMT_MAX_BATCH", "1")))))
        fmt.opt = opt
        fmt.final_construction_options = {k: v for k, v in vars(opt).items() if not k.startswith("_")}
        fmt.cudnn_benchmark_setting = cudnn_benchmark
        fmt.target_device = torch.device(target_device)
        out = dict(vars(opt))
        out["rank"] = str(out.get("rank"))
        return (fmt, fps, out, int(num_prev_frames + wav2vec_sec * fps))


class LoadWav2VecModel:
    UNIQUE_NAME = "LoadWav2VecModel"
    DISPLAY_NAME = "Load Wav2Vec Model (for Audio Encoding)"
    DESCRIPTION = "Loads a wav2vec2-base folder (config.json + model.safetensors) for the audio conditioning operator."
    DEFAULT_FOLDER = "wav2vec2-base-960h"
    CATEGORY = FILE_CATEGORY

    @classmethod
    def INPUT_TYPES(cls):
        device_options, default_device = _device_options()
        return {"required": {
            "model_folder": (look_for_models(WAV2VEC_DIR, cls.DEFAULT_FOLDER, dirs=True), {}),
            "target_device": (device_options, {"default": default_device}),
        }}

    RETURN_TYPES = ("INT", "WAV2VEC_PIPE")
    RETURN_NAMES = ("sampling_rate", "wav2vec_pipe")
    FUNCTION = "load_float_wav2vec_model"

    def load_float_wav2vec_model(self, model_folder, target_device):
        """Returns (sampling_rate, (state, config)): the operator itself is built by Load Audio Projection Layer's consumer,
        because float_aud_* fuses wav2vec2 and the projection (include/float_hip.h)."""
        folder = os.path.join(models_dir(), WAV2VEC_DIR, model_folder)
        wpath = os.path.join(folder, "model.safetensors")
        if not os.path.exists(wpath):
            wpath = ensure_model_part_exists("wav2vec2_base", os.path.join(WAV2VEC_DIR, model_folder), "model.safetensors")
        sd = {("wav2vec2." + (k[len("wav2vec2."):] if k.startswith("wav2vec2.") else k)): v for k, v in _load_sd(wpath).items()
              if not k.startswith(("lm_head.", "quantizer.", "project_"))}
        cfg_path = os.path.join(folder, "config.json")
        if os.path.exists(cfg_path):
            from transformers import Wav2Vec2Config
            cfg = AudioConfig.from_hf(Wav2Vec2Config.from_pretrained(folder))
        else:
            cfg = AudioConfig()
        pipe = {"state": sd, "config": cfg, "target_device": torch.device(target_device), "expected_sr": 16000}
        return (16000, pipe)


class LoadAudioProjectionLayer:
    UNIQUE_NAME = "LoadAudioProjectionLayer"
    DISPLAY_NAME = "Load Audio Projection Layer"
    DESCRIPTION = "Loads audio_projections/projection.safetensors; input feature width and dim_a are inferred from the weights."
    DEFAULT_FILENAME = "projection.safetensors"
    CATEGORY = FILE_CATEGORY

    @classmethod
    def INPUT_TYPES(cls):
        device_options, default_device = _device_options()
        return {"required": {
            "projection_file": (look_for_models(AUDIO_PROJ_DIR, cls.DEFAULT_FILENAME), {}),
            "target_device": (device_options, {"default": default_device}),
        }}

    RETURN_TYPES = ("AUDIO_PROJECTION_LAYER", "INT", "INT")
    RETURN_NAMES = ("projection_layer", "inferred_input_dim", "dim_a")
    FUNCTION = "load_projection_layer"

    def load_projection_layer(self, projection_file, target_device):
        sd = _load_sd(ensure_model_part_exists("projection", AUDIO_PROJ_DIR, projection_file))
        if "0.weight" not in sd or "1.weight" not in sd:
            raise KeyError("projection file must hold the Sequential keys 0.weight/0.bias (Linear) and 1.weight/1.bias (LayerNorm)")
        dim_a, din = sd["0.weight"].shape
        layer = {"state": {"audio_projection." + k: v for k, v in sd.items()}, "inferred_input_feature_dim": din, "dim_a": dim_a,
                 "target_device": torch.device(target_device)}
        return (layer, din, dim_a)


class LoadEmotionRecognitionModel:
    UNIQUE_NAME = "LoadEmotionRecognitionModel"
    DISPLAY_NAME = "Load Emotion Recognition Model"
    DESCRIPTION = "Loads a wav2vec2 speech-emotion folder (config.json + model.safetensors) as the HIP classification operator."
    DEFAULT_FOLDER = "wav2vec-english-speech-emotion-recognition"
    CATEGORY = FILE_CATEGORY

    @classmethod
    def INPUT_TYPES(cls):
        device_options, default_device = _device_options()
        return {"required": {
            "model_folder": (look_for_models(WAV2VEC_DIR, cls.DEFAULT_FOLDER, dirs=True), {}),
            "target_device": (device_options, {"default": default_device}),
        }}

    RETURN_TYPES = ("EMOTION_MODEL_PIPE", "INT")
    RETURN_NAMES = ("emotion_model_pipe", "dim_e")
    FUNCTION = "load_emotion_model"

    def load_emotion_model(self, model_folder, target_device):
        """(model, feature_extractor_ref, config_dict) like the reference (nodes_vadv_loader.py:287-340); the model is an
        Audio2EmotionHIP, the feature-extractor slot holds the normaliser this build applies (zero mean / unit variance)."""
        sub = os.path.join(WAV2VEC_DIR, model_folder)
        folder = os.path.join(models_dir(), sub)
        wpath = ensure_model_part_exists("emotion_ser", sub, "model.safetensors")
        sd = _load_sd(wpath)
        cfg_path = os.path.join(folder, "config.json")
        id2label = dict(Audio2EmotionHIP.id2label)
        if os.path.exists(cfg_path):
            from transformers import Wav2Vec2Config
            hf = Wav2Vec2Config.from_pretrained(folder)
            if not getattr(hf, "num_labels", None):
                raise ValueError("Missing `num_labels` in emotion recognition config")
            cfg = AudioConfig.from_hf(hf, num_labels=hf.num_labels)
            if getattr(hf, "id2label", None):
                id2label = {int(k): str(v) for k, v in hf.id2label.items()}
        else:
            cfg = emotion_audio_config()
        model = Audio2EmotionHIP(sd, cfg, target_device, dtype=os.environ.get("FLOAT_AMD_AUD_DTYPE", "fp16"))
        model.target_device = torch.device(target_device)
        info = {"num_labels": cfg.num_labels, "id2label": id2label, "label2id": {v: k for k, v in id2label.items()},
                "sampling_rate": 16000, "model_path": folder}
        from ... import host_models
        return ((model, host_models.preprocess_audio, info), cfg.num_labels)


def build_audio_encoder(wav2vec_pipe, projection_layer, only_last_features=None):
    """WAV2VEC_PIPE + AUDIO_PROJECTION_LAYER -> AudioEncoderHIP (the fused float_aud_* operator)."""
    cfg = wav2vec_pipe["config"]
    din = projection_layer["inferred_input_feature_dim"]
    only_last = (din == cfg.hidden_size) if only_last_features is None else only_last_features
    if din != (cfg.hidden_size if only_last else cfg.hidden_size * cfg.num_hidden_layers):
        raise TypeError("projection input width %d does not match the wav2vec2 features (%d per layer x %d layers); "
                        "`only_last_features` mismatch?" % (din, cfg.hidden_size, cfg.num_hidden_layers))
    cfg = AudioConfig(**{**cfg.__dict__, "dim_w": projection_layer["dim_a"], "only_last_features": only_last})
    sd = dict(wav2vec_pipe["state"])
    sd.update(projection_layer["state"])
    return AudioEncoderHIP(sd, cfg, wav2vec_pipe["target_device"], dtype=os.environ.get("FLOAT_AMD_AUD_DTYPE", "fp16"))
