#!/usr/bin/env python3
"""frames32 classes on the C4 share's shape (12 500 one-second stereo files through LoadSample): ms per launch by mask."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import afec_amd as afx
import bench
files = bench.make_c4_files(12500, 1234)
plan = afx.Plan(max_analysis_ms=0)
star = afx.D_MFCC | afx.D_SPECTRAL_STATS & ~afx.D_SPECTRAL_FLUX
for name, mask in (("mfcc (class 0)", afx.D_MFCC), ("mfcc + statistics (class 1)", star), ("mfcc + magnitude (class 3, no bands kernel)", afx.D_MFCC | afx.D_MAGNITUDE),
                   ("all spectral", afx.D_ALL_LOW_LEVEL)):
    b, _ = plan.batch_from_raw([(f, 2) for f in files], mask)
    for _ in range(5):
        b.run()
    b.sync()
    ms = min(b.run_timed(10) / 10 for _ in range(3))
    print(f"{name:45s} {ms:7.3f} ms  info {b.info()}")
    b.close()
