#!/usr/bin/env python3
"""Parity and timing summary of the rhythm tracker on the GPU box (writes gpurun_out/<round>/rhythm_report.md).

Parity: per golden signal, the fraction of bit-equal onset-function values and the largest deviation of every scalar
from the oracle.  Timing: HIP-event time of the rhythm kernels for (a) many short files (the C4 share: 10 k-sample
one-shots), (b) 4-second loops, (c) one 20-second file (the serial worst case: one workgroup)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import afec_amd as afx  # noqa: E402
from tests import _oracle  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out", os.environ.get("AFX_ROUND", "r03"))
os.makedirs(OUT, exist_ok=True)
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "rhythm.npz"))
lines = ["# rhythm tracker: GPU vs oracle, and kernel time", ""]

plan = afx.Plan()
o = _oracle.Oracle()
names = ["loop120", "loop95", "oneshot", "melody"]
xs = [GOLD[f"pcm_{n}"].astype(np.float64) / 32768.0 for n in names]
b = plan.batch(xs, afx.D_RHYTHM | afx.D_STATISTICS)
b.run()
r = b.fetch_rhythm(onset_functions=True)
lines += ["| signal | frames | complex fn bit-equal | power fn bit-equal | max rel err onset fn | onsets (c/p) | worst scalar rel err |",
          "|---|---|---|---|---|---|---|"]
for i, (n, x) in enumerate(zip(names, xs)):
    ref = o.run_rhythm(x, cap=True)
    sl = slice(r["offsets"][i], r["offsets"][i + 1])
    odf, oref = r["onset_functions"][sl], ref["odf"].T.astype(np.float32)
    rel = np.abs(odf.astype(np.float64) - oref) / (np.abs(oref).max(axis=0) + 1e-30)
    s, w = r["scalars"][i], ref["scalars"]
    srel = np.max(np.abs(s - w) / np.maximum(np.abs(w), 1e-12))
    lines.append(f"| {n} | {odf.shape[0]} | {np.mean(odf[:, 0] == oref[:, 0]):.4f} | {np.mean(odf[:, 1] == oref[:, 1]):.4f} | "
                 f"{rel.max():.2e} | {int(s[0])}/{int(s[6])} | {srel:.2e} |")
b.close()


def timed(bufs, label, mask=afx.D_RHYTHM, steps=5):
    bt = plan.batch(bufs, mask)
    bt.run(); bt.sync()
    ms = bt.run_timed(steps) / steps
    frames = int(bt.rhythm_frames()[-1])
    lines.append(f"| {label} | {len(bufs)} | {frames} | {ms:.3f} | {frames / ms / 1e3:.2f} |")
    bt.close()
    return ms


ONLY = os.environ.get("AFX_RT_ONLY", "")
rng = np.random.default_rng(3)
lines += ["", "| workload | files | 512/128 frames | ms per pass (HIP events) | M frames/s |", "|---|---|---|---|---|"]
short = [(rng.uniform(-1, 1, 10240) * np.exp(-np.arange(10240) / 3000.0)).astype(np.float32) for _ in range(256)]
if ONLY in ("", "short"):
    timed(short * 32, "8192 one-shots of 10240 samples")
loops = [GOLD["pcm_loop120"].astype(np.float32) / 32768.0] * 1024
if ONLY in ("", "loops"):
    timed(loops, "1024 loops of 4 s")
long_x = np.tile(GOLD["pcm_loop95"].astype(np.float32) / 32768.0, 5)[:20 * 44100]
if ONLY in ("", "long"):
    timed([long_x], "one 20 s file")
if ONLY in ("", "long256"):
    timed([long_x] * 256, "256 files of 20 s")
plan.close()
open(os.path.join(OUT, "rhythm_report.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
