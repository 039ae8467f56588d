#!/usr/bin/env python3
"""Which STFT kernel for a crawler batch?  n one-second stereo files, every per-frame descriptor + statistics: ms per run
on the 64-lane and on the half-wave frame kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import afec_amd as afx
import bench
pool = bench.make_c4_files(64, 99)
mask = afx.D_ALL_PER_FRAME | afx.D_STATISTICS
for n in (64, 128, 256, 512, 1024, 2048):
    files = [(pool[i % 64], 2) for i in range(n)]
    row = []
    for name, fk in (("wave64", afx.FRAME_KERNEL_WAVE64), ("halfwave", afx.FRAME_KERNEL_HALFWAVE)):
        plan = afx.Plan(frame_kernel=fk)
        b, _ = plan.batch_from_raw(files, mask)
        for _ in range(5):
            b.run()
        b.sync()
        ms = min(b.run_timed(20) / 20 for _ in range(3))
        row.append((name, ms, b.info()["chunk_frames"], b.total_frames))
        b.close(); plan.close()
    print(f"{n:5d} files, {row[0][3]:6d} frames: " + ", ".join(f"{nm} {ms:6.3f} ms (K={k})" for nm, ms, k, _ in row))
