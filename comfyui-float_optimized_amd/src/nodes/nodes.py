"""Simple nodes (reference nodes.py:23-222): Load FLOAT Models (Opt) and FLOAT Process (Opt), same
class attributes, widget names, return tuples and batch/seed semantics; the body runs the MI355X
hot path (HIP) with the host-side encoders on PyTorch-ROCm."""
import contextlib
import os

import torch

from . import EMOTIONS, FLOAT_UNIFIED_MODEL, SYNTHETIC_MODEL, main_logger
from .generate import InferenceAgent
from .options.base_options import BaseOptions

try:  # ComfyUI is optional: the nodes also run head-less (tests, bench)
    import folder_paths
    _MODELS_DIR = folder_paths.models_dir
except Exception:  # pragma: no cover
    _MODELS_DIR = os.environ.get("FLOAT_MODELS_DIR", os.path.join(os.path.expanduser("~"), "ComfyUI", "models"))


def _device_options():
    opts = ["cuda:%d" % i for i in range(torch.cuda.device_count())] or ["cuda:0"]
    return opts, opts[0]


class LoadFloatModels:
    @classmethod
    def INPUT_TYPES(s):
        device_options, default_device = _device_options()
        float_models_path = os.path.join(_MODELS_DIR, "float")
        files = []
        if os.path.isdir(float_models_path):
            files = sorted(f for f in os.listdir(float_models_path) if f.lower().endswith((".safetensors", ".pth")))
        if not files:
            files = [BaseOptions.ckpt_filename]
        files = files + [SYNTHETIC_MODEL]
        return {
            "required": {
                "model": (files, {"default": FLOAT_UNIFIED_MODEL}),
                "target_device": (device_options, {"default": default_device}),
                "cudnn_benchmark": ("BOOLEAN", {"default": False}, ),
            },
            "optional": {
                "advanced_float_options": ("ADV_FLOAT_DICT",)
            }
        }

    RETURN_TYPES = ("FLOAT_PIPE",)
    RETURN_NAMES = ("float_pipe",)
    FUNCTION = "loadmodel"
    CATEGORY = "FLOAT"
    DESCRIPTION = "Models are auto-downloaded to /ComfyUI/models/float"
    UNIQUE_NAME = "LoadFloatModelsOpt"
    DISPLAY_NAME = "Load FLOAT Models (Opt)"

    def loadmodel(self, model, target_device, cudnn_benchmark, advanced_float_options=None):
        opt = BaseOptions()
        if advanced_float_options is not None and isinstance(advanced_float_options, dict):
            for key, value in advanced_float_options.items():
                if hasattr(opt, key):
                    setattr(opt, key, value)
                else:
                    main_logger.warning("opt_instance has no attribute '%s' from advanced_float_options.", key)
        from . import TORCHDIFFEQ_FIXED_STEP_SOLVERS
        if opt.torchdiffeq_ode_method not in TORCHDIFFEQ_FIXED_STEP_SOLVERS:
            raise ValueError("unknown fixed-step ODE method %r" % (opt.torchdiffeq_ode_method,))
        opt.rank = torch.device(target_device)
        opt.cudnn_benchmark = cudnn_benchmark  # accepted for graph compatibility; MIOpen is not on this path
        opt.ckpt_path = os.path.join(_MODELS_DIR, "float", model)
        if model.lower().endswith(".pth"):
            raise ValueError("legacy float.pth + separate wav2vec folders is not supported; use FLOAT.safetensors")
        return (InferenceAgent(opt),)


class FloatProcess:
    @classmethod
    def INPUT_TYPES(s):
        return {
            "required": {
                "ref_image": ("IMAGE",),
                "ref_audio": ("AUDIO",),
                "float_pipe": ("FLOAT_PIPE",),
                "a_cfg_scale": ("FLOAT", {"default": 2.0, "min": 1.0, "step": 0.1}),
                "e_cfg_scale": ("FLOAT", {"default": 1.0, "min": 1.0, "step": 0.1}),
                "fps": ("FLOAT", {"default": 25, "step": 1}),
                "emotion": (EMOTIONS, {"default": "none"}),
                "face_align": ("BOOLEAN", {"default": True}, ),
                "seed": ("INT", {"default": 62064758300528, "min": 0, "max": 0xffffffffffffffff}),
            },
        }

    RETURN_TYPES = ("IMAGE", "AUDIO", "FLOAT")
    RETURN_NAMES = ("images", "ref_audio", "fps")
    FUNCTION = "floatprocess"
    CATEGORY = "FLOAT"
    DESCRIPTION = "Float Processing"
    UNIQUE_NAME = "FloatProcessOpt"
    DISPLAY_NAME = "FLOAT Process (Opt)"

    def floatprocess(self, ref_image, ref_audio, float_pipe, a_cfg_scale, e_cfg_scale, fps, emotion, face_align, seed):
        # reference nodes.py:173-175: `with model_to_target(main_logger, float_pipe.G)` - operators resident inside the call
        ctx = float_pipe.model_to_target() if hasattr(float_pipe, "model_to_target") else contextlib.nullcontext()
        with ctx:
            return self._floatprocess(ref_image, ref_audio, float_pipe, a_cfg_scale, e_cfg_scale, fps, emotion, face_align, seed)

    def _floatprocess(self, ref_image, ref_audio, float_pipe, a_cfg_scale, e_cfg_scale, fps, emotion, face_align, seed):
        float_pipe.opt.fps = fps
        image_batch_size = ref_image.shape[0]
        audio_waveform = ref_audio['waveform']
        audio_sample_rate = ref_audio['sample_rate']
        audio_batch_size = audio_waveform.shape[0]
        target_batch_size = max(image_batch_size, audio_batch_size)
        all_images, used_audio = [], []
        # item i: image min(i, Bi-1), audio min(i, Ba-1), seed + i  (reference nodes.py:189-209)
        if target_batch_size > 1 and hasattr(float_pipe, "infer_device_batch") and os.environ.get("FLOAT_AMD_BATCH_CLIPS", "1") != "0":
            # the items of one call have equal length (one audio tensor): their FMT chains run stacked
            # (InferenceAgent.infer_device_batch -> float_fmt_sample_batch), each with its own noise stream of seed + i
            items = []
            for i in range(target_batch_size):
                ii, ai = min(i, image_batch_size - 1), min(i, audio_batch_size - 1)
                wf = audio_waveform[ai:ai + 1]
                items.append(float_pipe.host_inputs(ref_image[ii:ii + 1], {'waveform': wf, 'sample_rate': audio_sample_rate}, not face_align))
                used_audio.append(wf.cpu())
            all_images = float_pipe.infer_device_batch(items, a_cfg_scale, float_pipe.opt.r_cfg_scale, e_cfg_scale,
                                                       None if emotion == "none" else emotion,
                                                       [seed + i for i in range(target_batch_size)])
            target_range = range(0)
        else:
            target_range = range(target_batch_size)
        for i in target_range:
            ii, ai = min(i, image_batch_size - 1), min(i, audio_batch_size - 1)
            img = ref_image[ii:ii + 1]
            wf = audio_waveform[ai:ai + 1]
            images_thwc = float_pipe.run_inference(None, img, {'waveform': wf, 'sample_rate': audio_sample_rate},
                                                   a_cfg_scale=a_cfg_scale, r_cfg_scale=float_pipe.opt.r_cfg_scale,
                                                   e_cfg_scale=e_cfg_scale, emo=None if emotion == "none" else emotion,
                                                   no_crop=not face_align, seed=seed + i)
            all_images.append(images_thwc.cpu())
            used_audio.append(wf.cpu())
        if target_batch_size == 1:
            out_audio = ref_audio
        else:
            cat = torch.cat([w.squeeze(0) for w in used_audio], dim=1).unsqueeze(0)
            out_audio = {'waveform': cat.to(audio_waveform.device), 'sample_rate': audio_sample_rate}
        # one item: hand the agent's (pinned) tensor on as it is - torch.cat would copy 786 MB for nothing
        images = all_images[0] if target_batch_size == 1 else torch.cat(all_images, dim=0)
        return (images, out_audio, fps,)
