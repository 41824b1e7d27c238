"""Node surface of ComfyUI-FLOAT_Optimized on the MI355X hot path (reference src/nodes/__init__.py).
Only the three north-star nodes are registered: Load FLOAT Models (Opt), FLOAT Process (Opt) and
FLOAT Advanced Options."""
import logging

__version__ = "0.1.0"
NODES_NAME = "FLOAT_Optimized"
EMOTIONS = ['none', 'angry', 'disgust', 'fear', 'happy', 'neutral', 'sad', 'surprise']
TORCHDIFFEQ_FIXED_STEP_SOLVERS = ["euler", "midpoint", "rk4", "heun2", "heun3"]
RGBA_CONVERSION_STRATEGIES = ["blend_with_color", "discard_alpha", "replace_with_color"]
FLOAT_UNIFIED_MODEL = "FLOAT.safetensors"
SYNTHETIC_MODEL = "synthetic (seeded random weights)"

main_logger = logging.getLogger(NODES_NAME)

from .nodes import LoadFloatModels, FloatProcess  # noqa: E402
from .nodes_adv import FloatAdvancedParameters  # noqa: E402

_NODES = (LoadFloatModels, FloatProcess, FloatAdvancedParameters)
NODE_CLASS_MAPPINGS = {c.UNIQUE_NAME: c for c in _NODES}
NODE_DISPLAY_NAME_MAPPINGS = {c.UNIQUE_NAME: c.DISPLAY_NAME for c in _NODES}
