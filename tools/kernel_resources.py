#!/usr/bin/env python3
"""tools/kernel_resources.py -> tests/golden/kernel_resources.json: VGPRs, scratch bytes per lane, waves per SIMD and LDS bytes
of every kernel of afec_amd/csrc as hipcc compiles it (-Rpass-analysis=kernel-resource-usage), the record
tests/test_isa_hazards_cpu.py holds later builds against.  Run it after a deliberate change of a kernel file."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_isa_hazards_cpu as t  # noqa: E402

kernels = {}
with tempfile.TemporaryDirectory() as d:
    for source in sorted(os.listdir(t.CSRC)):
        if source.endswith(".hip"):
            kernels.update(t.kernel_resources(t.device_isa(os.path.join(t.CSRC, source), d)[1]))
out = {"_how": "tools/kernel_resources.py: hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage, per kernel",
       "kernels": dict(sorted(kernels.items()))}
with open(os.path.join(ROOT, "tests", "golden", "kernel_resources.json"), "w") as f:
    json.dump(out, f, indent=1)
    f.write("\n")
print(len(kernels), "kernels")
