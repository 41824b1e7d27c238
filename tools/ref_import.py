"""Import the upstream reference (read-only, /root/reference) as a numerical checker.

BUILD-CONTAINER ONLY.  The reference never travels to the GPU box; this module is
used by tools/make_goldens.py (fixture generation) and by the optional
`-m "not gpu"` tests that cross-check oracle/ against the live reference when
/root/reference exists.  Nothing under the product package imports this file.

The reference depends on packages that are absent here (seconohe, comfy, timm,
torchdiffeq, cv2, librosa, face_alignment, folder_paths).  Each is replaced by a
minimal stand-in injected into sys.modules *before* import:

  * timm.layers.use_fused_attn -> True (torch>=2 behaviour, FMT.py:60,75)
  * timm.models.vision_transformer.Mlp -> fc1/act/fc2 (published timm layout,
    timm>=1.0.9; parameter names fix the checkpoint keys blocks.N.mlp.fc1/fc2)
  * torchdiffeq.odeint -> fixed-grid Euler: y_{i+1} = y_i + (t_{i+1}-t_i) f(t_i,y_i)
    (torchdiffeq's documented `method='euler'`; unpinned in requirements.txt:3)
"""
import importlib
import logging
import os
import sys
import types

REF_ROOT = os.environ.get("FLOAT_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "src", "nodes", "models", "float"))


_stubs = {}  # every stand-in module this tool created, by name (they leave sys.modules again: see _remove_stubs)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    _stubs[name] = m
    return m


def stub(name):
    """A stand-in module by name, after it has left sys.modules (tools/make_goldens.py patches `seconohe.torch.model_to_target`)."""
    return _stubs[name]


def _remove_stubs():
    """The stand-ins only have to exist while the reference modules are imported: they bind what they need at import time.
    Left in sys.modules they would answer imports of code this tool does not own - the product's optional `import
    face_alignment` and `import folder_paths` (ComfyUI's module: nodes_vadv_loader.models_dir) took the stand-ins for the real
    thing, which made two CPU tests depend on the order of the test files."""
    for name, m in _stubs.items():
        if sys.modules.get(name) is m and not name.startswith("floatref"):  # (the namespace packages of the reference tree stay)
            del sys.modules[name]


def _restore_stubs():
    for name, m in _stubs.items():
        sys.modules.setdefault(name, m)


def _install_stubs():
    import torch
    import torch.nn as nn
    import transformers  # noqa: F401  (must be imported before timm is stubbed)

    if "seconohe" not in sys.modules:
        _mod("seconohe")
        _mod("seconohe.logger", initialize_logger=lambda name: logging.getLogger(name))
        _mod("seconohe.torch",
             get_torch_device_options=lambda: (["cpu"], "cpu"),
             model_to_target=None,
             get_canonical_device=lambda d: torch.device(d),
             get_offload_device=lambda: torch.device("cpu"))
        _mod("seconohe.downloader", download_file=None)
        sys.modules["seconohe"].logger = sys.modules["seconohe.logger"]

    if "comfy" not in sys.modules:
        class ProgressBar:
            def __init__(self, total):
                self.total = total

            def update(self, n):
                pass

        comfy = _mod("comfy")
        comfy.utils = _mod("comfy.utils", ProgressBar=ProgressBar)
        comfy.model_management = _mod("comfy.model_management",
                                      unet_offload_device=lambda: torch.device("cpu"))
        _mod("folder_paths", models_dir="/tmp/float_models")

    if "timm" not in sys.modules:
        class Mlp(nn.Module):
            # Linear -> act -> Linear, timm vision_transformer.Mlp (drop=0, norm=None)
            def __init__(self, in_features, hidden_features=None, out_features=None,
                         act_layer=nn.GELU, drop=0.0, **kw):
                super().__init__()
                out_features = out_features or in_features
                hidden_features = hidden_features or in_features
                self.fc1 = nn.Linear(in_features, hidden_features)
                self.act = act_layer()
                self.fc2 = nn.Linear(hidden_features, out_features)

            def forward(self, x):
                return self.fc2(self.act(self.fc1(x)))

        _mod("timm")
        _mod("timm.layers", use_fused_attn=lambda: True)
        _mod("timm.models")
        _mod("timm.models.vision_transformer", Mlp=Mlp)

    if "torchdiffeq" not in sys.modules:
        def odeint(func, y0, t, method="euler", **kw):
            assert method == "euler", "only fixed-grid Euler is stood in for"
            ys = [y0]
            y = y0
            for i in range(len(t) - 1):
                y = y + (t[i + 1] - t[i]) * func(t[i], y)
                ys.append(y)
            return torch.stack(ys, 0)

        _mod("torchdiffeq", odeint=odeint)

    for name in _EMPTY_STUBS:
        if name not in sys.modules:
            _mod(name)


# Empty stand-ins that only have to exist while the reference modules are imported (`import cv2` at module scope).  They
# are taken out of sys.modules again at the end of load(): code this tool does not own (the product's optional
# `import face_alignment`) must see the package as absent, not as an importable module without attributes.
_EMPTY_STUBS = ("cv2", "librosa", "face_alignment")
_loaded = {}


def import_ref(name):
    """importlib.import_module("floatref." + name) for a reference module imported after load() (tools/make_goldens.py): the
    stand-ins exist for the duration of the import only."""
    load()
    _restore_stubs()
    try:
        return importlib.import_module("floatref." + name)
    finally:
        _remove_stubs()


def load():
    """Returns a namespace with the reference classes/functions on the hot path."""
    if _loaded:
        return _loaded["ns"]
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    _install_stubs()
    # Namespace packages so src/nodes/__init__.py runs but the top-level
    # __init__.py (needs seconohe.register_nodes + ComfyUI) does not.
    root = _mod("floatref")
    root.__path__ = [REF_ROOT]
    src = _mod("floatref.src")
    src.__path__ = [os.path.join(REF_ROOT, "src")]
    nodes = importlib.import_module("floatref.src.nodes")
    FMT = importlib.import_module("floatref.src.nodes.models.float.FMT")
    styledecoder = importlib.import_module("floatref.src.nodes.models.float.styledecoder")
    encoder = importlib.import_module("floatref.src.nodes.models.float.encoder")
    generator = importlib.import_module("floatref.src.nodes.models.float.generator")
    base_options = importlib.import_module("floatref.src.nodes.options.base_options")
    ns = types.SimpleNamespace(nodes=nodes, FMT=FMT, styledecoder=styledecoder, encoder=encoder,
                               generator=generator, base_options=base_options)
    try:
        ns.nodes_adv = importlib.import_module("floatref.src.nodes.nodes_adv")
    except Exception as e:  # pragma: no cover - optional
        ns.nodes_adv = None
        ns.nodes_adv_error = e
    try:
        ns.FLOAT = importlib.import_module("floatref.src.nodes.models.float.FLOAT")
    except Exception as e:  # pragma: no cover - optional
        ns.FLOAT = None
        ns.FLOAT_error = e
    _remove_stubs()
    _loaded["ns"] = ns
    return ns
