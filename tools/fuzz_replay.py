#!/usr/bin/env python3
"""tools/fuzz_replay.py <seed> <round> [stats] -> gpurun_out/fuzz/replay_<seed>_<round>.npz

The inputs (buffers, descriptor mask) of one round of tests/fuzz_gpu.py, regenerated WITHOUT a GPU: no draw of the fuzz
depends on a result, so its generator can be fast-forwarded (tests.fuzz_gpu.draw_round).  For the analysis of a warning or
a mismatch the soak logged with its seed and round -- e.g. against the oracle and the reference's own objects
(oracle/_ref/ref_driver through tests/golden/make_golden.run_ref): profiles/r06/fuzz_seed93.log was analysed this way."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.fuzz_gpu import draw_round  # noqa: E402


def replay(seed, target, stats_mode=False):
    rng = np.random.default_rng(seed)
    for _ in range(target):
        draw_round(rng, stats_mode)
    return draw_round(rng, stats_mode)


if __name__ == "__main__":
    seed, target = int(sys.argv[1]), int(sys.argv[2])
    drawn = replay(seed, target, len(sys.argv) > 3 and sys.argv[3] == "stats")
    out = os.path.join(ROOT, "gpurun_out", "fuzz")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, f"replay_{seed}_{target}.npz")
    np.savez_compressed(path, mask=np.array(drawn["mask"]), **{f"buf{i}": b for i, b in enumerate(drawn["bufs"])})
    print(f"{path}: mask {drawn['mask']:#x}, buffers {[(b.size, str(b.dtype)) for b in drawn['bufs']]}")
