import time, numpy as np, sys
sys.path.insert(0, '.')
import afec_amd as afx
rng = np.random.default_rng(0)
plan = afx.Plan()
for secs, nrep in ((2.0, 200), (20.0, 50)):
    x = rng.uniform(-1, 1, int(44100 * secs)).astype(np.float32)
    for mask, name in ((afx.D_C2, "mfcc"), (afx.D_ALL_LOW_LEVEL, "all"), (afx.D_ALL_PER_FRAME | afx.D_EFFECTIVE_LENGTH, "everything")):
        plan.extract([x], mask)
        t0 = time.perf_counter()
        for _ in range(nrep):
            plan.extract([x], mask)
        dt = (time.perf_counter() - t0) / nrep
        print(f"one file of {secs:4.1f} s, {name:10s}: {dt*1e3:7.3f} ms per afx_extract_batch call  ({1/dt:7.0f} files/s)")

# PCIe-inclusive rate of a 10k-frame buffer from pageable and from page-locked host memory
from afec_amd.capi import pinned_array
plan0 = afx.Plan(max_analysis_ms=0)
n = 2048 + 1024 * 9999
x = rng.uniform(-1, 1, n).astype(np.float32)
xp, owner = pinned_array((n,), np.float32)
xp[:] = x
for name, buf in (("pageable", x), ("pinned", xp)):
    plan0.extract([buf], afx.D_C2)
    t0 = time.perf_counter()
    for _ in range(10):
        plan0.extract([buf], afx.D_C2)
    dt = (time.perf_counter() - t0) / 10
    print(f"10k-frame buffer from {name:8s} host memory: {dt*1e3:6.2f} ms per call = {10000/dt/1e6:5.2f} M frames/s incl. H2D/D2H")

# the LoadSample front end, one decoded file per call (afx_batch_create_from_raw + run + fetch)
for secs, ch in ((2.0, 1), (20.0, 2)):
    raw = np.round(rng.uniform(-0.5, 0.5, int(44100 * secs) * ch) * 32767).astype(np.int16)
    for _ in range(3):
        b, info = plan.batch_from_raw([(raw, ch)], afx.D_ALL_PER_FRAME); b.run(); b.fetch(); b.close()
    t0 = time.perf_counter()
    for _ in range(50):
        b, info = plan.batch_from_raw([(raw, ch)], afx.D_ALL_PER_FRAME); b.run(); b.fetch(); b.close()
    dt = (time.perf_counter() - t0) / 50
    print(f"one {ch}-channel 16-bit file of {secs:4.1f} s through the front end, every descriptor: {dt*1e3:7.3f} ms per file")
