#!/usr/bin/env python3
"""Interleaved A/B timing of several builds of libafx_hip.so in one process-per-run loop.

usage: ab_bench.py [--rounds N] [--precision f64|f32] [--mask c2|stats|all] lib1.so lib2.so ...
Prints Mframes/s per library per round (bench.py is run as a subprocess with AFX_LIBRARY set)."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--precision", default="f64")
    ap.add_argument("--mask", default="c2", choices=["c2", "star", "stats", "all", "frame", "neighbours"])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("libs", nargs="+")
    args = ap.parse_args()
    res = {l: [] for l in args.libs}
    for _ in range(args.rounds):
        for l in args.libs:
            env = dict(os.environ, AFX_LIBRARY=os.path.abspath(l))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(args.steps), "--warmup", "3",
                                  "--no-cpu-baseline", "--precision", args.precision, "--mask", args.mask],
                                 env=env, capture_output=True, text=True)
            try:
                d = json.loads(out.stdout.strip().splitlines()[-1])
                res[l].append(d["value"] / 1e6)
            except Exception:
                res[l].append(float("nan"))
                sys.stderr.write(out.stderr[-500:])
    for l in args.libs:
        print(os.path.basename(l), " ".join(f"{v:7.1f}" for v in res[l]))


if __name__ == "__main__":
    main()
