"""The LDS swizzles of the decoder's conv kernels (csrc/dec_kernels.hpp: halo tiles by halo column, weight rows / z tile by row)
are conflict-free for ds_read_b128 on gfx950 at every tile alignment; the unswizzled layout is not (r01 PMC: 42 % of LDS cycles)."""
import importlib.util
import os

from .util import ROOT

spec = importlib.util.spec_from_file_location("lds_swizzle", os.path.join(ROOT, "tools", "probes", "lds_swizzle.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)


def test_kernel_swizzles_have_no_bank_conflicts():
    assert mod.conflicts(mod.pixel_swizzle) == 0
    assert mod.conflicts(mod.column_swizzle) == 0
    assert mod.conflicts(lambda P, x: 0) > 0
