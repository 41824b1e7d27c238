"""Node surface of ComfyUI-FLOAT_Optimized on the MI355X hot path (reference src/nodes/__init__.py): the three
north-star nodes - Load FLOAT Models (Opt), FLOAT Process (Opt), FLOAT Advanced Options - and the very-advanced (VA)
loaders / stage nodes that expose the operators of the path one by one (display names carry the module's "(VA)"
suffix like the reference's register_nodes does)."""
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

from . import nodes_vadv, nodes_vadv_loader  # noqa: E402

_NODES = (LoadFloatModels, FloatProcess, FloatAdvancedParameters)
_VA_NODES = (nodes_vadv_loader.LoadFloatEncoderModel, nodes_vadv_loader.LoadFloatSynthesisModel, nodes_vadv_loader.LoadFMTModel,
             nodes_vadv_loader.LoadWav2VecModel, nodes_vadv_loader.LoadAudioProjectionLayer, nodes_vadv.ApplyFloatEncoder,
             nodes_vadv.FloatGetIdentityReferenceVA, nodes_vadv.FloatSampleMotionSequenceRD_VA, nodes_vadv.ApplyFloatSynthesis,
             nodes_vadv.FloatAudioPreprocessAndFeatureExtract, nodes_vadv.FloatApplyAudioProjection,
             nodes_vadv_loader.LoadEmotionRecognitionModel, nodes_vadv.FloatExtractEmotionWithCustomModel,
             nodes_vadv.FloatExtractEmotionWithCustomModelDyn)
NODE_CLASS_MAPPINGS = {c.UNIQUE_NAME: c for c in _NODES + _VA_NODES}
NODE_DISPLAY_NAME_MAPPINGS = {c.UNIQUE_NAME: c.DISPLAY_NAME for c in _NODES}
NODE_DISPLAY_NAME_MAPPINGS.update({c.UNIQUE_NAME: c.DISPLAY_NAME + " " + nodes_vadv.SUFFIX for c in _VA_NODES})
