"""SURVEY 8(d) "parity gate": max / mean relative error per descriptor of the HIP path against the
reference goldens and the oracle, written as a table (gpurun_out/parity_report.md; the copy under
profiles/ is the one kept).  Relative error = |got - ref| / max(|ref|, floor) with the descriptor's absolute
floor from tests/_tol.py divided by its rtol (so "1.0" in units of the tolerance is the bar)."""
import os

import numpy as np
import pytest

import afec_amd as afx
from tests import _tol
from tests._oracle import FIELDS, NEIGH_FIELDS, Oracle

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# exactly flat spectrum: decided by rounding noise in any implementation (see test_gpu_parity.py)
SKIP = {("impulse", "sub_complexity"), ("impulse", "sub_flux"), ("impulse", "spectral_flux"),
        ("impulse", "spectral_complexity")}


def c3_like_files(n_files, seed):
    rng = np.random.default_rng(seed)
    n = 88200
    t = np.arange(n) / 44100.0
    out = []
    for _ in range(n_files):
        x = np.zeros(n)
        for _ in range(int(rng.integers(1, 4))):
            x += rng.uniform(0.2, 0.6) * np.sin(2 * np.pi * rng.uniform(110.0, 4000.0) * t + rng.uniform(0, 6.28))
        x += rng.uniform(0.2, 0.8) * rng.uniform(-1, 1, n) * np.exp(-t / rng.uniform(0.05, 0.5))
        x[:2205] = 0.0
        out.append((x / np.max(np.abs(x))).astype(np.float32))
    return out


def test_write_parity_report():
    plan, oracle = afx.Plan(max_analysis_ms=0), Oracle()
    acc = {}   # field -> list of (relative errors array, error in units of the tolerance)

    def add(field, got, ref, rtol, atol):
        got, ref = np.asarray(got, dtype=np.float64).reshape(-1), np.asarray(ref, dtype=np.float64).reshape(-1)
        err = np.abs(got - ref)
        floor = atol / rtol if rtol > 0 else 0.0
        denom = np.maximum(np.abs(ref), floor) if floor > 0 else np.where(ref != 0, np.abs(ref), 1.0)
        if rtol > 0 or atol > 0:
            units = err / (rtol * np.abs(ref) + atol)
        else:                                   # exact descriptors: any difference is a failure
            units = np.where(err != 0, np.inf, 0.0)
        acc.setdefault(field, []).append((err / denom, units))

    def run(name, x, ref_spec, ref_neigh):
        res = plan.extract([x], afx.D_ALL_PER_FRAME)
        for field, (a, b) in FIELDS.items():
            if field == "mag" or (name, field) in SKIP:
                continue
            add(field, res[field], ref_spec[:, a:b], *_tol.GPU_TOL[field])
        for field, col in NEIGH_FIELDS.items():
            if (name, field) in SKIP:
                continue
            add(field, res[field], ref_neigh[:, col], *_tol.NEIGH_TOL[field])

    zf, zn = np.load(os.path.join(GOLD, "frames.npz")), np.load(os.path.join(GOLD, "neighbours.npz"))
    n_golden = 0
    for k in zf.files:
        if k.startswith("in_"):
            name = k[3:]
            run(name, zf[k], zf["ref_" + name], zn["ref_" + name])     # against the reference's own objects
            n_golden += zf["ref_" + name].shape[0]
    n_oracle = 0
    for i, x in enumerate(c3_like_files(24, 99)):
        x64 = x.astype(np.float64)
        run(f"c3_{i}", x, oracle.run(x64), oracle.run_neighbours(x64))  # against the pinned oracle
        n_oracle += oracle.num_frames(x.size, False)
    plan.close()

    lines = ["# Parity report: HIP path (f64) vs the reference goldens and the oracle", "",
             f"{n_golden} golden frames (reference's own objects) + {n_oracle} frames of 24 C3-like files (oracle).",
             "Relative error = |got - ref| / max(|ref|, floor); the last column is the worst error in units of the",
             "test tolerance (rtol |ref| + atol of tests/_tol.py; <= 1 passes; the bar is rtol = 1e-4).", "",
             "| descriptor | values | max rel err | mean rel err | worst / tolerance |", "|---|---|---|---|---|"]
    worst_overall = 0.0
    for field, parts in acc.items():
        rel = np.concatenate([p[0] for p in parts])
        units = np.concatenate([p[1] for p in parts])
        worst = float(np.max(units)) if units.size else 0.0
        worst_overall = max(worst_overall, worst)
        lines.append(f"| {field} | {rel.size} | {rel.max():.2e} | {rel.mean():.2e} | "
                     + ("exact" if rel.max() == 0 else f"{worst:.3g}") + " |")
    text = "\n".join(lines) + "\n"
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "parity_report.md"), "w") as f:
            f.write(text)
    except OSError:
        pass
    print(text)
    assert worst_overall <= 1.0, text
