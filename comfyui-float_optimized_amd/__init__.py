"""ComfyUI-FLOAT_Optimized, MI355X-native hot path (FMT Euler sampling + Synthesis decoder).

The directory name carries a hyphen like any ComfyUI custom-node checkout, so it is loaded
by path (ComfyUI does the same): see tests/util.py::load_pkg.  Sub-modules are imported
lazily; `native` refuses to work without the HIP library (no CPU fallback in the product).
"""
import importlib

__version__ = "0.1.0"

_SUBMODULES = ("config", "weights", "native", "fmt", "decoder", "encoder", "audio", "pipeline", "distributed", "host_models")


def __getattr__(name):
    if name in _SUBMODULES:
        mod = importlib.import_module("." + name, __name__)
        globals()[name] = mod
        return mod
    if name in ("NODE_CLASS_MAPPINGS", "NODE_DISPLAY_NAME_MAPPINGS"):
        nodes = importlib.import_module(".src.nodes", __name__)
        return getattr(nodes, name)
    raise AttributeError(name)
