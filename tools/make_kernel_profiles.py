#!/usr/bin/env python3
"""profiles/kernel_profiles.json (what bench.py's roofline object quotes) from the per-configuration summaries that
tools/profile_config.py wrote on the GPU box (gpurun_out/<round>/profile_<tag>.json); the summaries and kernel-stats
files are copied to profiles/<round>/ alongside.

usage: make_kernel_profiles.py [round]      (run from the repo root after a GPU round)"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r06"
SRC = os.path.join(ROOT, "gpurun_out", ROUND)
DST = os.path.join(ROOT, "profiles", ROUND)
CLOCK_GHZ = 2.0    # fallback when a profile has no GRBM_GUI_ACTIVE: in-kernel (s_memtime / s_memrealtime, stamps build) at steady
                   # state under the f64 load: 1.92-2.07 over boxes (profiles/r03/clock_ramp.txt)
LIMITER = {"c2_f64": "f64 VALU work; sustained clock 1.9-2.1 of 2.4 GHz under this f64 + LDS + HBM load; LDS ~70 % busy beside it"}

os.makedirs(DST, exist_ok=True)
out = {"_comment": "Per bench configuration, from rocprofv3 passes on MI355X (tools/profile_config.py; raw summaries "
                   f"profiles/{ROUND}/profile_<tag>.json): HBM bytes per frame = (2*FETCH_SIZE + WRITE_SIZE)*1024/frames (gfx950 "
                   "FETCH_SIZE half-count correction, calibrated for 16-B streaming reads only), VALU pipe cycles per frame = "
                   "4*SQ_ACTIVE_INST_VALU/frames summed over the kernels of one step (rhythm kernels on the batch's own stream: "
                   "bench.py --no-side-stream), clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time of the same profiled run."}
builds = set()
for f in sorted(glob.glob(os.path.join(SRC, "profile_*.json"))):
    builds.add(json.load(open(f)).get("build_info"))
if len(builds) != 1:
    sys.exit(f"the profiles of {SRC} come from more than one build (or none): {builds}")
out["_build_info"] = builds.pop()   # afx_build_info() of the library the counters were measured on (bench.py checks it)
for f in sorted(glob.glob(os.path.join(SRC, "profile_*.json"))):
    tag = os.path.basename(f)[len("profile_"):-len(".json")]
    d = json.load(open(f))
    # steady-state time per step (warm-up launches dropped, tools/profile_config.py) where the trace gave it
    ms = {k: v.get("ms_per_step_steady", v.get("ms_per_step", 0.0)) for k, v in d["kernels"].items()}
    kernels = sorted(d["kernels"], key=lambda k: -ms[k])
    out[tag] = {
        "bytes_per_frame": d["per_frame"]["hbm_bytes"],
        "valu_cycles_per_frame": d["per_frame"]["valu_cycles"],
        "valu_instructions_per_frame": d["per_frame"]["valu_instructions"],
        # the clock the counters themselves saw: GRBM_GUI_ACTIVE / 8 XCDs / kernel time of the profiled run
        "clock_ghz": round(d.get("clock_ghz_grbm") or CLOCK_GHZ, 3),
        "host": d.get("host"),
        "frames_profiled": d["frames_per_step"],
        "limiter": LIMITER.get(tag, "f64 VALU issue"),
        "kernels": kernels,
        "kernel_ms_per_step": {k: round(ms[k], 4) for k in kernels},
        "dominant_kernel": kernels[0] if kernels else None,
        "source": f"profiles/{ROUND}/profile_{tag}.json",
    }
    shutil.copy(f, DST)
for pat in ("*_kernel_stats.csv", "*_kernel_steady.csv", "*_pmc_summary.csv", "*_pmc.csv", "rhythm_report.md", "parity_report.md", "ubench_*.txt",
            "bench_default.json", "pytest_gpu.log", "*_pmc_kernels.csv", "single_buffer.txt", "kernel_choice_by_batch_size.txt",
            "halfwave_classes_on_c4.txt", "e2e_cpu_accounting.txt", "shards8_busy_cpus.txt", "fuzz_*.log", "crawl_soak.log",
            "lease_report.json", "lease_report.txt", "lease_host.txt", "bench_*.json"):
    for f in glob.glob(os.path.join(SRC, pat)):
        shutil.copy(f, DST)
json.dump(out, open(os.path.join(ROOT, "profiles", "kernel_profiles.json"), "w"), indent=1)
print("wrote profiles/kernel_profiles.json:", ", ".join(k for k in out if k != "_comment"))
