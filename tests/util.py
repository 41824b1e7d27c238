"""Shared test helpers: load the hyphen-named package by path, locate fixtures."""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "comfyui-float_optimized_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
PKG_NAME = "float_amd"


def load_pkg():
    if PKG_NAME in sys.modules:
        return sys.modules[PKG_NAME]
    spec = importlib.util.spec_from_file_location(
        PKG_NAME, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[PKG_NAME] = mod
    spec.loader.exec_module(mod)
    return mod


def golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: (torch.from_numpy(z[k]) if z[k].ndim else z[k].item()) for k in z.files}


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_abs(a, b):
    return float((a.double() - b.double()).abs().max())


def seeded_normal(seed, *shape, std=1.0):
    """RandomState(seed).standard_normal(shape) * std as fp32: the input generator of tools/make_goldens.py (`rnd`)."""
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32) * std)


def sample_inputs(cfg, seed, T, dynamic, noise_seed=15):
    """Inputs of tools/make_goldens.py::gen_fmt_sample regenerated from the seed (the full-length fixtures
    fmt_sample_config2 / config5 hold the reference's r_d only): wa, r_s, we (static softmax vector, or per-window
    vectors nearest-upsampled to T like nodes_vadv.py:829-840), and the reference's sequential noise draws."""
    import math
    L = cfg.num_frames_for_clip
    wa = seeded_normal(seed + 11, 1, T, cfg.dim_a)
    r_s = seeded_normal(seed + 12, 1, cfg.dim_w)
    n_chunks = int(math.ceil(T / L))
    if dynamic:
        we_w = torch.softmax(seeded_normal(seed + 13, 1, n_chunks, cfg.dim_e), -1)
        idx = torch.clamp((torch.arange(T).float() * n_chunks / T).long(), max=n_chunks - 1)
        we = we_w[:, idx]
    else:
        we = torch.softmax(seeded_normal(seed + 13, 1, 1, cfg.dim_e), -1)
    g = torch.Generator("cpu")
    g.manual_seed(int(noise_seed))
    noise = torch.stack([torch.randn(1, L, cfg.dim_w, generator=g) for _ in range(n_chunks)])
    return dict(wa=wa, r_s=r_s, we=we, noise=noise)
