"""One 20 s file per afx_extract_batch call, every descriptor: the workload to profile per kernel
(rocprofv3 --kernel-trace --stats -- python3 tools/onefile_profile.py)."""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import afec_amd as afx
rng = np.random.default_rng(0)
plan = afx.Plan()
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
x = rng.uniform(-1, 1, int(44100 * secs)).astype(np.float32)
for _ in range(20):
    plan.extract([x], afx.D_ALL_PER_FRAME | afx.D_EFFECTIVE_LENGTH)
