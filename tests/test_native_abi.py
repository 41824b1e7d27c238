"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol
include/float_hip.h declares; no compute is called (no GPU here)."""
import os
import re

from tests.util import ROOT, load_pkg

pkg = load_pkg()


def test_library_exports_header_symbols():
    hdr = open(os.path.join(ROOT, "include", "float_hip.h")).read()
    declared = set(re.findall(r"\b(float_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"float_tensor_t", "float_fmt_cfg_t", "float_dec_cfg_t", "float_dec_unit_t"}
    assert declared == set(pkg.native.EXPORTS), declared ^ set(pkg.native.EXPORTS)
    L = pkg.native.lib()
    for name in declared:
        assert hasattr(L, name)
    assert L.float_hip_abi_version() == 6  # v6: float_enc_export_feats16; v5: float_probe_peaks, optional `blur_kernel` tensor; v4: float_{fmt,enc,aud}_saturation (v3: fp32 decoder, float_dec_saturation, float_dec_debug_* unit ops)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under the package may reference it."""
    pdir = os.path.join(ROOT, "comfyui-float_optimized_amd")
    for base, _, files in os.walk(pdir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                src = open(os.path.join(base, f)).read()
                assert "float_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_argument_validation_needs_no_gpu():
    """Error paths of the C ABI that return before any HIP call: status code + message, no exception across
    the boundary (the host mirror turns FLOAT_E_INVALID into ValueError)."""
    import ctypes as C
    N = pkg.native
    L = N.lib()
    h = C.c_void_p()
    assert L.float_fmt_create(None, None, 0, C.byref(h)) == 1
    assert b"null argument" in L.float_last_error()
    cfg = N.FmtCfg(512, 512, 7, 1000, 8, 8, 4000, 10, 50, 2, 0, 0)  # dim_h not a multiple of 256
    arr = (N.FloatTensor * 1)()
    assert L.float_fmt_create(C.byref(cfg), arr, 0, C.byref(h)) == 1 and b"dim_h" in L.float_last_error()
    dcfg = N.DecCfg(100, 512, 0, 8)  # size not a power of two
    assert L.float_dec_create(C.byref(dcfg), arr, 0, C.byref(h)) == 1 and b"power of two" in L.float_last_error()
    ecfg = N.EncCfg(100, 512, 20, 1)  # encoder size not a power of two
    assert L.float_enc_create(C.byref(ecfg), arr, 0, C.byref(h)) == 1 and b"power of two" in L.float_last_error()
    assert L.float_enc_forward(None, None, None, None, None, None, 0, None) == 1
    acfg = N.AudCfg()
    acfg.n_conv = 1  # a one-layer feature extractor is not a wav2vec2
    assert L.float_aud_create(C.byref(acfg), arr, 0, C.byref(h)) == 1 and b"n_conv" in L.float_last_error()
    acfg.n_conv = 7
    for i in range(7):
        acfg.conv_dim[i], acfg.conv_kernel[i], acfg.conv_stride[i] = 512, 3, 2
    acfg.hidden, acfg.heads = 768, 8  # head dim 96
    assert L.float_aud_create(C.byref(acfg), arr, 0, C.byref(h)) == 1 and b"head dim" in L.float_last_error()
    assert L.float_aud_inference(None, None, 0, 0, None, None) == 1 and L.float_aud_classify(None, None, 0, None, None) == 1
    assert L.float_dec_direction(None, None, None, None) == 1 and L.float_dec_set_feats16(None, None, 0, 0, None) == 1
    assert L.float_fmt_sample_next(None, None, None, None) == 1
    try:
        N.check(1)
    except ValueError as e:
        assert "float_fmt_sample_begin" in str(e)
    else:
        raise AssertionError("check(1) must raise ValueError")
