#!/usr/bin/env python3
"""Profile one bench.py configuration on the GPU box with rocprofv3 and write a per-kernel summary.

usage: profile_config.py <tag> [bench.py args ...]      (run from the repo root; writes gpurun_out/<round>/)

Separate passes, as MI355X_MICROARCH.md prescribes (never --pmc together with a trace domain):
  1. --kernel-trace --stats                    -> average duration per kernel
  2. --pmc FETCH_SIZE                          -> bytes fetched over the fabric (x2: gfx950 counts 64 B per 128-B request)
  3. --pmc WRITE_SIZE
  4. --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
Output: profile_<tag>.json {frames_per_step, kernels: {name: {...per step...}}, totals per frame} and the raw
kernel_stats csv.  The program after `--` is python3 itself (no env / shell wrapper)."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("AFX_ROUND", "r06")
# steps of a few ms (C3: 3.7 ms) need more warm-up launches to get past the ~30 ms the clocks take to settle from an
# idle GPU (profiles/r03/clock_ramp.txt): AFX_PROF_WARMUP / AFX_PROF_STEPS
STEPS, WARMUP = int(os.environ.get("AFX_PROF_STEPS", "10")), int(os.environ.get("AFX_PROF_WARMUP", "2"))


def run_pass(tag, name, prof_args, bench_args):
    d = f"/tmp/prof_{tag}_{name}"
    shutil.rmtree(d, ignore_errors=True)
    # bench.py's clock probe stays off under the profiler: its pass is a REPEAT of the timed launches with a wave resident
    # beside them (the trace would count those launches, 4 % slower, as steps; the counter passes serialise dispatches and
    # the probe kernel would block them).  The clock of a profiled run is GRBM_GUI_ACTIVE / 8 XCDs / kernel time (sq pass);
    # the clock of the un-profiled line is the line's own (tools/lease_report.py puts them side by side)
    probe = ["--no-clock-probe"]
    cmd = ["rocprofv3"] + prof_args + ["--output-format", "csv", "-d", d, "-o", "p", "--", "python3",
                                       os.path.join(ROOT, "bench.py"), "--steps", str(STEPS), "--warmup", str(WARMUP),
                                       "--no-cpu-baseline", "--no-single", "--no-spot-check", "--no-side-stream",
                                       "--no-sharded-crawl"] + probe + bench_args
    # --no-side-stream: the rhythm kernels on the batch's own stream, so that a kernel's duration is its own
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return d, (json.loads(line[-1]) if line else None), r


def short(name):
    n = name.replace("void afx::(anonymous namespace)::", "").replace("afx::(anonymous namespace)::", "")
    return n.split("(")[0]


def main():
    tag, bench_args = sys.argv[1], sys.argv[2:]
    out_dir = os.path.join(ROOT, "gpurun_out", ROUND)
    os.makedirs(out_dir, exist_ok=True)
    launches = STEPS + WARMUP

    d, bench, r = run_pass(tag, "trace", ["--kernel-trace", "--stats"], bench_args)
    if bench is None:
        sys.stderr.write(r.stdout[-2000:] + r.stderr[-2000:])
        sys.exit(1)
    frames = bench["config"]["frames_per_gpu_per_step"]
    kernels = collections.OrderedDict()
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(out_dir, f"{tag}_kernel_stats.csv"))
        for row in csv.DictReader(open(stats[0])):
            if "afx::" not in row["Name"]:
                continue
            k = kernels.setdefault(short(row["Name"]), {})
            k["calls_per_step"] = int(row["Calls"]) / launches
            k["ms_per_step"] = float(row["TotalDurationNs"]) / launches * 1e-6
    # steady-state durations: the --stats rows mix the warm-up launches (cold caches, clock ramp) with the timed ones;
    # from the trace itself, per kernel, only the launches behind the first WARMUP steps count
    trace = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if trace:
        per = collections.defaultdict(list)
        for row in csv.DictReader(open(trace[0])):
            if "afx::" in row["Kernel_Name"]:
                per[short(row["Kernel_Name"])].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
        with open(os.path.join(out_dir, f"{tag}_kernel_steady.csv"), "w") as fo:
            fo.write("kernel,launches_counted,launches_dropped,mean_ns,min_ns,max_ns\n")
            for name, runs in per.items():
                runs.sort()
                per_step = max(1, round(len(runs) / launches))
                steady = [dur for _, dur in runs[per_step * WARMUP:]]
                if not steady:
                    continue
                k = kernels.setdefault(name, {})
                k["ms_per_step_steady"] = sum(steady) / (len(steady) / per_step) * 1e-6
                fo.write(f"{name},{len(steady)},{per_step * WARMUP},{sum(steady) / len(steady):.0f},{min(steady)},{max(steady)}\n")
    passes = [("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
              ("sq", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                      "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"])]
    if os.environ.get("AFX_PROF_TRACE_ONLY"):     # A/B timing of builds: the kernel trace alone
        passes = []
    for name, counters in passes:
        d, _, r = run_pass(tag, name, ["--pmc"] + counters, bench_args)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if "afx::" not in row["Kernel_Name"]:
                    continue
                k = kernels.setdefault(short(row["Kernel_Name"]), {})
                k[row["Counter_Name"]] = k.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"]) / launches
    tot = collections.defaultdict(float)
    for k in kernels.values():
        for c, v in k.items():
            tot[c] += v
    hbm_bytes = (2.0 * tot.get("FETCH_SIZE", 0.0) + tot.get("WRITE_SIZE", 0.0)) * 1024.0
    clock = None
    if tot.get("GRBM_GUI_ACTIVE") and tot.get("ms_per_step"):
        clock = tot["GRBM_GUI_ACTIVE"] / 8.0 / (tot["ms_per_step"] * 1e-3) / 1e9   # profiled runs: lower bound of the free-running clock
    sys.path.insert(0, ROOT)
    import afec_amd
    summary = {
        "build_info": afec_amd.build_info(),
        "bench_args": bench_args,
        "frames_per_step": frames,
        "host": os.uname().nodename,
        "traced_run": {"frames_per_s": bench["value"], "ms_per_step": bench["ms_per_step"],
                       "clock_ghz_in_run": bench["roofline"].get("clock_ghz_in_run")},   # the bench line of the kernel-trace pass itself
        "unprofiled_value_frames_per_s": bench["value"],
        "kernels": kernels,
        "per_frame": {
            "hbm_bytes": hbm_bytes / frames if hbm_bytes else None,
            "fetch_bytes_x2": 2.0 * tot.get("FETCH_SIZE", 0.0) * 1024.0 / frames,
            "write_bytes": tot.get("WRITE_SIZE", 0.0) * 1024.0 / frames,
            "valu_instructions": tot.get("SQ_INSTS_VALU", 0.0) / frames,
            "valu_cycles": 4.0 * tot.get("SQ_ACTIVE_INST_VALU", 0.0) / frames,
            "algorithmic_bytes": bench["roofline"]["algorithmic_bytes_per_frame"],
        },
        "kernel_ms_per_step": tot.get("ms_per_step"),
        "clock_ghz_grbm": clock,
    }
    json.dump(summary, open(os.path.join(out_dir, f"profile_{tag}.json"), "w"), indent=1)
    pf = summary["per_frame"]
    print(f"{tag}: {frames} frames/step, {summary['kernel_ms_per_step']:.3f} ms kernels/step, HBM {pf['hbm_bytes']:.0f} B/frame "
          f"(algorithmic {pf['algorithmic_bytes']}), VALU {pf['valu_instructions']:.0f} instr = {pf['valu_cycles']:.0f} cycles per frame")
    for n, k in kernels.items():
        print(f"   {n:60s} {k.get('ms_per_step', 0):8.3f} ms  valu {4.0 * k.get('SQ_ACTIVE_INST_VALU', 0) / frames:8.0f} cyc/frame "
              f"fetch x2 {2048.0 * k.get('FETCH_SIZE', 0) / frames:8.0f} B write {1024.0 * k.get('WRITE_SIZE', 0) / frames:7.0f} B")


if __name__ == "__main__":
    main()
