import time, numpy as np, sys
sys.path.insert(0, '.')
import afec_amd as afx
rng = np.random.default_rng(0)
plan = afx.Plan()
for secs, nrep in ((2.0, 200), (20.0, 50)):
    x = rng.uniform(-1, 1, int(44100 * secs)).astype(np.float32)
    for mask, name in ((afx.D_C2, "mfcc"), (afx.D_ALL_LOW_LEVEL, "all")):
        plan.extract([x], mask)
        t0 = time.perf_counter()
        for _ in range(nrep):
            plan.extract([x], mask)
        dt = (time.perf_counter() - t0) / nrep
        print(f"one file of {secs:4.1f} s, {name:4s}: {dt*1e3:7.3f} ms per afx_extract_batch call  ({1/dt:7.0f} files/s)")
