"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol
include/float_hip.h declares; no compute is called (no GPU here)."""
import os
import re

from tests.util import ROOT, load_pkg

pkg = load_pkg()


def test_library_exports_header_symbols():
    hdr = open(os.path.join(ROOT, "include", "float_hip.h")).read()
    declared = set(re.findall(r"\b(float_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"float_tensor_t", "float_fmt_cfg_t", "float_dec_cfg_t"}
    assert declared == set(pkg.native.EXPORTS), declared ^ set(pkg.native.EXPORTS)
    L = pkg.native.lib()
    for name in declared:
        assert hasattr(L, name)
    assert L.float_hip_abi_version() == 1


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under the package may reference it."""
    pdir = os.path.join(ROOT, "comfyui-float_optimized_amd")
    for base, _, files in os.walk(pdir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                src = open(os.path.join(base, f)).read()
                assert "float_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f
