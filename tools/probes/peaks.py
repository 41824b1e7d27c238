"""Measured peaks of the box (float_probe_peaks): streaming read / copy bandwidth, register-operand fp16 MFMA loops.  Three runs,
one JSON line each; `python tools/probes/peaks.py > profiles/<round>_peaks.json` on the GPU box."""
import json
import sys

sys.path.insert(0, ".")
from tests.util import load_pkg  # noqa: E402

pkg = load_pkg()
import torch  # noqa: E402

p = torch.cuda.get_device_properties(0)
for i in range(3):
    print(json.dumps(dict(pkg.native.probe_peaks("cuda:0"), name=p.name, arch=getattr(p, "gcnArchName", None))))
