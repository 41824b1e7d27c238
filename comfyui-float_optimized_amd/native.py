"""ctypes binding of libfloat_hip.so (include/float_hip.h).  No fallback: if the library is
missing or fails to load, importing anything that computes raises NativeLibraryError."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libfloat_hip.so")

FLOAT_DT_BF16, FLOAT_DT_FP16, FLOAT_DT_FP32 = 0, 1, 2
ODE_METHODS = {"euler": 0, "midpoint": 1, "rk4": 2, "heun2": 3, "heun3": 4}
DTYPES = {"bf16": FLOAT_DT_BF16, "bfloat16": FLOAT_DT_BF16, "fp16": FLOAT_DT_FP16, "float16": FLOAT_DT_FP16,
          "fp32": FLOAT_DT_FP32, "float32": FLOAT_DT_FP32}  # fp32: the verification mode of the FMT and decoder operators
DTYPE_NAMES = {FLOAT_DT_BF16: "bf16", FLOAT_DT_FP16: "fp16", FLOAT_DT_FP32: "fp32"}
DEC_SAT_SITES = 40
ABI_VERSION = 6


def canon_dtype(dtype):
    """'float16' / 'bfloat16' / 'float32' -> 'fp16' / 'bf16' / 'fp32': the operator mirrors keep the canonical name, so
    that every `op.dtype == "fp16"` test (range checks, watchdogs) sees an fp16 handle however it was asked for."""
    try:
        return DTYPE_NAMES[DTYPES[str(dtype)]]
    except KeyError:
        raise ValueError("unknown dtype %r (one of %s)" % (dtype, ", ".join(sorted(DTYPES)))) from None


class NativeLibraryError(RuntimeError):
    pass


class FloatTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("ndim", C.c_int32),
                ("shape", C.c_int64 * 6)]


class FmtCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dim_w", "dim_a", "dim_e", "dim_h", "depth", "heads", "mlp_hidden",
                                         "n_prev", "n_cur", "attn_window", "dtype", "use_graph", "max_batch")]


class DecCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("size", "style_dim", "dtype", "max_frames")]


class DecUnit(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dtype", "cin", "cout", "res", "upsample", "n_frames", "style_dim", "flags")]


class EncCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("size", "dim", "dim_motion", "dtype")]


class AudCfg(C.Structure):
    _fields_ = [("n_conv", C.c_int32), ("conv_dim", C.c_int32 * 8), ("conv_kernel", C.c_int32 * 8), ("conv_stride", C.c_int32 * 8)] + \
               [(n, C.c_int32) for n in ("hidden", "layers", "heads", "intermediate", "pos_k", "pos_groups", "dim_w", "only_last",
                                         "dtype")] + [("ln_eps", C.c_float)] + \
               [(n, C.c_int32) for n in ("feat_norm_layer", "stable_ln", "conv_bias", "num_labels")]


_F = C.POINTER(C.c_float)
_SIGNATURES = {
    "float_hip_abi_version": (C.c_int, []),
    "float_last_error": (C.c_char_p, []),
    "float_set_profiling": (C.c_int, [C.c_int32]),
    "float_probe_peaks": (C.c_int, [C.POINTER(C.c_float)] * 4 + [C.POINTER(C.c_int32)]),
    "float_profile_ms": (C.c_double, [C.c_int32, C.POINTER(C.c_int64)]),
    "float_stream_create_cu_range": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "float_stream_destroy": (C.c_int, [C.c_void_p]),
    "float_fmt_create": (C.c_int, [C.POINTER(FmtCfg), C.POINTER(FloatTensor), C.c_int32, C.POINTER(C.c_void_p)]),
    "float_fmt_destroy": (None, [C.c_void_p]),
    "float_fmt_set_method": (C.c_int, [C.c_void_p, C.c_int32]),
    "float_fmt_debug": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "float_fmt_eval": (C.c_int, [C.c_void_p, C.c_float] + [C.c_void_p] * 4 + [C.c_int32] + [C.c_void_p] * 3 +
                       [C.c_float] * 3 + [C.c_int32, C.c_void_p, C.c_void_p]),
    "float_fmt_sample_chunk": (C.c_int, [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int32] + [C.c_void_p] * 3 +
                               [C.c_int32] + [C.c_float] * 3 + [C.c_int32, C.c_void_p, C.c_void_p]),
    "float_fmt_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                   C.c_int32] + [C.c_float] * 3 + [C.c_int32, C.c_void_p, C.c_void_p]),
    "float_fmt_sample_batch": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                         C.c_void_p, C.c_int32] + [C.c_float] * 3 + [C.c_int32, C.c_void_p, C.c_void_p]),
    "float_fmt_sample_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                         C.c_int32] + [C.c_float] * 3 + [C.c_int32, C.c_void_p]),
    "float_fmt_sample_next": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "float_fmt_sample_begin_range": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                               C.c_int32] + [C.c_float] * 3 + [C.c_int32, C.c_void_p, C.c_int32, C.c_int32] +
                                     [C.c_void_p] * 3),
    "float_dec_create": (C.c_int, [C.POINTER(DecCfg), C.POINTER(FloatTensor), C.c_int32, C.POINTER(C.c_void_p)]),
    "float_dec_destroy": (None, [C.c_void_p]),
    "float_dec_set_feats": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_void_p]),
    "float_dec_frames": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "float_dec_frames_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "float_dec_feat_shape": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "float_dec_frames_raw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "float_dec_direction": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "float_dec_saturation": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int32, C.c_void_p]),
    "float_dec_debug_styled_conv": (C.c_int, [C.POINTER(DecUnit), C.POINTER(FloatTensor), C.c_int32, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]),
    "float_dec_debug_flow_level": (C.c_int, [C.POINTER(DecUnit), C.POINTER(FloatTensor), C.c_int32] + [C.c_void_p] * 9),
    "float_dec_set_feats16": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_void_p]),
    "float_enc_create": (C.c_int, [C.POINTER(EncCfg), C.POINTER(FloatTensor), C.c_int32, C.POINTER(C.c_void_p)]),
    "float_enc_destroy": (None, [C.c_void_p]),
    "float_enc_forward": (C.c_int, [C.c_void_p] * 5 + [C.POINTER(C.c_void_p), C.c_int32, C.c_void_p]),
    "float_aud_create": (C.c_int, [C.POINTER(AudCfg), C.POINTER(FloatTensor), C.c_int32, C.POINTER(C.c_void_p)]),
    "float_aud_destroy": (None, [C.c_void_p]),
    "float_aud_reserve": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "float_aud_classify": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "float_aud_inference": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "float_enc_feats16": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32)]),
    "float_enc_export_feats16": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_void_p]),
    "float_fmt_saturation": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_int32, C.c_void_p]),
    "float_enc_saturation": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_int32, C.c_void_p]),
    "float_aud_saturation": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_int32, C.c_void_p]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


def lib():
    """Load (once) and return the shared library; raises NativeLibraryError if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            "libfloat_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C comfyui-float_optimized_amd/csrc` (there is no CPU fallback)")
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:
        raise NativeLibraryError("cannot load %s: %s" % (LIB_PATH, e)) from e
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise NativeLibraryError("libfloat_hip.so does not export %s" % name) from e
        fn.restype = res
        fn.argtypes = args
    if L.float_hip_abi_version() != ABI_VERSION:
        raise NativeLibraryError("libfloat_hip.so ABI version mismatch")
    _lib = L
    return L


_ERR_TYPES = {1: ValueError, 2: KeyError, 3: RuntimeError, 4: MemoryError}


def check(rc):
    if rc != 0:
        msg = lib().float_last_error().decode("utf-8", "replace")
        raise _ERR_TYPES.get(rc, RuntimeError)(msg)


def tensor_table(state):
    """dict[str, torch.Tensor] -> (ctypes array of float_tensor_t, keep-alive list)."""
    keep = []
    arr = (FloatTensor * len(state))()
    for i, (k, v) in enumerate(state.items()):
        t = v.detach().to("cpu", torch.float32).contiguous()
        nm = k.encode()
        keep.append((t, nm))
        arr[i].name = nm
        arr[i].data = C.cast(t.data_ptr(), _F)
        arr[i].ndim = t.dim()
        for d in range(t.dim()):
            arr[i].shape[d] = t.shape[d]
    return arr, keep


def dev_ptr(t, name="tensor"):
    """fp32 contiguous CUDA(HIP) tensor -> raw device pointer."""
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError("%s must live on the GPU" % name)
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise TypeError("%s must be contiguous float32" % name)
    return C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def cu_range_stream(cu_begin, cu_end, device=None):
    """torch stream object over a HIP stream whose kernels only run on CUs [cu_begin, cu_end)."""
    p = C.c_void_p()
    check(lib().float_stream_create_cu_range(int(cu_begin), int(cu_end), C.byref(p)))
    return torch.cuda.ExternalStream(p.value, device=device)


def saturation(fn_name, handle, device=None, reset=False):
    """float_{fmt,enc,aud}_saturation of a handle: threads with a clamped / non-finite 16-bit activation store since the
    handle was created (or last reset).  Synchronises the current stream."""
    tot = C.c_uint64(0)
    check(getattr(lib(), fn_name)(handle, C.byref(tot), 1 if reset else 0, stream_ptr(device)))
    return int(tot.value)


def set_profiling(on):
    check(lib().float_set_profiling(1 if on else 0))


def profile_ms(which):
    n = C.c_int64(0)
    ms = lib().float_profile_ms(which, C.byref(n))
    return ms, n.value


def rebuildable(cls):
    """Class decorator for the operator mirrors (one C handle in `_h`, released by `close()`): the object remembers its
    constructor arguments - the host state dict among them - so that a released handle can be re-created.
      op.offload()    frees every device allocation of the handle (packed weights, workspaces, graphs);
      op.to_target()  re-creates it where it was (packing is deterministic: the rebuilt operator is bitwise the first);
      op.resident     whether the handle exists.
    The counterpart of the reference's `with model_to_target(logger, model)` around every node call (nodes.py:173-175,
    nodes_vadv.py:107,192,275,347,437,520,697,807): weights on the target device inside, on the host after."""
    init = cls.__init__

    def __init__(self, *a, **k):
        if "_ctor" not in self.__dict__:  # the outermost class of an inheritance chain records; re-creation keeps the record
            self._ctor = (a, k)
        init(self, *a, **k)

    def offload(self):
        self.close()

    def to_target(self):
        if not self.__dict__.get("_h"):
            a, k = self._ctor
            type(self).__init__(self, *a, **k)
        return self

    cls.__init__ = __init__
    cls.offload = offload
    cls.to_target = to_target
    cls.resident = property(lambda self: bool(self.__dict__.get("_h")))
    return cls


def probe_peaks(device=None):
    """float_probe_peaks: {hbm_read_GBps, hbm_copy_GBps, mfma_f16_16x16x32_TFLOPs, mfma_f16_32x32x16_TFLOPs, compute_units} measured
    on `device` right now (4 GiB of scratch, ~0.1 s)."""
    v = [C.c_float(0) for _ in range(4)]
    n = C.c_int32(0)
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        torch.cuda.synchronize()
        check(lib().float_probe_peaks(C.byref(v[0]), C.byref(v[1]), C.byref(v[2]), C.byref(v[3]), C.byref(n)))
    return {"hbm_read_GBps": round(v[0].value, 1), "hbm_copy_GBps": round(v[1].value, 1), "mfma_f16_16x16x32_TFLOPs": round(v[2].value, 1),
            "mfma_f16_32x32x16_TFLOPs": round(v[3].value, 1), "compute_units": n.value}
