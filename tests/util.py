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
