#!/usr/bin/env python3
"""The literal BASELINE configs[1] shape -- ONE resident buffer of 10 000 frames, MFCC only -- on both STFT kernels:
time per launch (HIP events over 200 launches) and what the planner chose."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import afec_amd as afx

rng = np.random.Generator(np.random.MT19937(4321))
x = (rng.random(9999 * 1024 + 2048, dtype=np.float32) * 2 - 1).astype(np.float32)
for frames in (10000, 5000, 20000, 40000):
    xs = x[:(frames - 1) * 1024 + 2048] if frames <= 10000 else np.tile(x, 4)[:(frames - 1) * 1024 + 2048]
    for name, fk in (("wave64", afx.FRAME_KERNEL_WAVE64), ("halfwave", afx.FRAME_KERNEL_HALFWAVE), ("auto", afx.FRAME_KERNEL_AUTO)):
        plan = afx.Plan(max_analysis_ms=0, frame_kernel=fk)
        b = plan.batch([xs], afx.D_MFCC)
        for _ in range(20):
            b.run()
        b.sync()
        best = min(b.run_timed(200) / 200 for _ in range(3))
        info = b.info()
        print(f"{frames:6d} frames {name:9s}: {best * 1e3:7.2f} us per launch = {frames / best / 1e3:7.1f} M frames/s   "
              f"kernel {info['frame_kernel']} K={info['chunk_frames']} chunks={info['n_chunks']} blocks={info['grid_blocks']}")
        b.close()
        plan.close()
