"""FLOAT Advanced Options node (reference nodes_adv.py:130-235): twelve widgets -> ADV_FLOAT_DICT."""
from . import RGBA_CONVERSION_STRATEGIES, TORCHDIFFEQ_FIXED_STEP_SOLVERS

BASE_CATEGORY = "FLOAT/Advanced"


def _num(kind, default, **kw):
    d = {"default": default}
    d.update(kw)
    return (kind, d)


class FloatAdvancedParameters:
    @classmethod
    def INPUT_TYPES(cls):
        prob = dict(min=0.0, max=1.0, step=0.01, display="number")
        tol = dict(min=1e-9, max=1e-1, step=1e-6, display="number", precision=9)
        return {
            "required": {
                "r_cfg_scale": _num("FLOAT", 1.0, min=1.0, step=0.1),
                "attention_window": _num("INT", 2, min=1, max=10, step=1, display="number"),
                "audio_dropout_prob": _num("FLOAT", 0.1, **prob),
                "ref_dropout_prob": _num("FLOAT", 0.1, **prob),
                "emotion_dropout_prob": _num("FLOAT", 0.1, **prob),
                "ode_atol": _num("FLOAT", 1e-5, **tol),
                "ode_rtol": _num("FLOAT", 1e-5, **tol),
                "nfe": _num("INT", 10, min=1, max=1000, step=1, display="number"),
                "torchdiffeq_ode_method": (TORCHDIFFEQ_FIXED_STEP_SOLVERS, {"default": "euler"}),
                "face_margin": _num("FLOAT", 1.6, min=1.2, max=2.0, step=0.1, display="number"),
                "rgba_conversion": (RGBA_CONVERSION_STRATEGIES, {"default": "blend_with_color"}),
                "bkg_color_hex": ("STRING", {"default": "#000000"}),
            }
        }

    RETURN_TYPES = ("ADV_FLOAT_DICT",)
    RETURN_NAMES = ("advanced_options",)
    FUNCTION = "get_options"
    CATEGORY = BASE_CATEGORY
    DESCRIPTION = "FLOAT Advanced Options"
    UNIQUE_NAME = "FloatAdvancedParameters"
    DISPLAY_NAME = "FLOAT Advanced Options"

    def get_options(self, r_cfg_scale, attention_window, audio_dropout_prob, ref_dropout_prob, emotion_dropout_prob,
                    ode_atol, ode_rtol, nfe, torchdiffeq_ode_method, face_margin, rgba_conversion, bkg_color_hex):
        return (dict(r_cfg_scale=r_cfg_scale, attention_window=attention_window, audio_dropout_prob=audio_dropout_prob,
                     ref_dropout_prob=ref_dropout_prob, emotion_dropout_prob=emotion_dropout_prob, ode_atol=ode_atol,
                     ode_rtol=ode_rtol, nfe=nfe, torchdiffeq_ode_method=torchdiffeq_ode_method, face_margin=face_margin,
                     rgba_conversion=rgba_conversion, bkg_color_hex=bkg_color_hex),)
