import sys
sys.path.insert(0, ".")
from tests.util import load_pkg
pkg = load_pkg()
import torch
for i in range(2):
    print(pkg.native.probe_peaks("cuda:0"))
