#!/usr/bin/env python3
"""One lease, one record (tools/gpu.sh lease): the driver's bench line and the rocprofv3 profiles of the same build taken
on the same box, side by side -- for c2 (the headline), c3 and c4 (the chain rates of the line) the line's ms per step and
in-run clock next to the profiled run's steady kernel time and clock.  The round-5 review's finding was a profile whose
kernel time exceeded the timed run's step on another box; here both come from one host and the report says whether
profiled kernel time <= 1.02 x the line's ms per step.

usage: lease_report.py        (run from the repo root after `bench` and `profiles`; writes gpurun_out/<round>/lease_report.json)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("AFX_ROUND", "r06")
O = os.path.join(ROOT, "gpurun_out", ROUND)


def load(name):
    try:
        text = open(os.path.join(O, name)).read().strip()
        return json.loads(text.splitlines()[-1]) if name.startswith("bench") else json.loads(text)
    except (OSError, ValueError, IndexError):
        return None


def main():
    line = load("bench_default.json")
    host = open(os.path.join(O, "lease_host.txt")).read().split() if os.path.exists(os.path.join(O, "lease_host.txt")) else [os.uname().nodename]
    rows = []
    if line is None:
        sys.exit("no bench_default.json in " + O)
    cfg = line["config"]
    pairs = [("c2_f64", "headline (C2 x 512, MFCC)", {"ms_per_step": line["ms_per_step"], "kernel_ms_per_step": line["roofline"]["launch_ms"],
                                                       "frames_per_s": line["value"], "clock_ghz_in_run": line["roofline"].get("clock_ghz_in_run")}),
             ("c3", "config.c3_frames_per_s", cfg.get("c3_frames_per_s")),
             ("c3_all", "config.c3_spectral_set_frames_per_s", cfg.get("c3_spectral_set_frames_per_s")),
             ("c4", "config.c4_share_frames_per_s", cfg.get("c4_share_frames_per_s")),
             ("c4_crawler", "config.c4_share_at_crawler_shape", cfg.get("c4_share_at_crawler_shape"))]
    for tag, what, timed in pairs:
        prof = load(f"profile_{tag}.json")
        if not prof or not timed:
            rows.append({"tag": tag, "what": what, "missing": "profile" if not prof else "line object"})
            continue
        steady = sum(k.get("ms_per_step_steady", k.get("ms_per_step", 0.0)) for k in prof["kernels"].values())
        step = timed.get("kernel_ms_per_step") or timed["ms_per_step"]
        rows.append({"tag": tag, "what": what, "line_ms_per_step": timed["ms_per_step"], "line_kernel_ms_per_step": timed.get("kernel_ms_per_step"),
                     "line_frames_per_s": timed["frames_per_s"], "line_clock_ghz_in_run": timed.get("clock_ghz_in_run"),
                     "profiled_kernel_ms_per_step_steady": steady, "profiled_run_ms_per_step": (prof.get("traced_run") or {}).get("ms_per_step"),
                     "profiled_run_clock_ghz_in_run": (prof.get("traced_run") or {}).get("clock_ghz_in_run"),
                     "profiled_clock_ghz_grbm": prof.get("clock_ghz_grbm"), "profile_host": prof.get("host"),
                     "profiled_over_line": steady / step, "within_1.02": steady <= 1.02 * step,
                     # what the ratio compares: the profile sums the durations of kernels run one after the other on ONE stream
                     # (bench.py --no-side-stream); the line's step overlaps the time-domain kernels with the spectral chain on
                     # the batch's side stream (c3 / c4: the sum may exceed the step by the overlap), and in the crawler's shape
                     # five batches run at once (durations of concurrent kernels stretch: their sum is not a wall time)
                     "comparable": tag != "c4_crawler",
                     "same_build": prof.get("build_info") == (line["roofline"].get("profile_build") or prof.get("build_info"))})
    out = {"host": host[0], "when": host[1] if len(host) > 1 else None, "bench_value_frames_per_s": line["value"], "rows": rows}
    json.dump(out, open(os.path.join(O, "lease_report.json"), "w"), indent=1)
    print(f"lease on {out['host']} ({out['when']}): headline {line['value'] / 1e6:.1f} M frames/s")
    for r in rows:
        if "missing" in r:
            print(f"  {r['tag']:11s} {r['what']}: no {r['missing']}")
            continue
        note = ""
        if not r["within_1.02"]:
            note = ("  > 1.02: the line overlaps the time-domain kernels on the side stream, the profile runs them in a row" if r["comparable"]
                    else "  five batches at once: a sum of stretched durations, not a wall time")
        clock = r["line_clock_ghz_in_run"] or float("nan")
        grbm = r["profiled_clock_ghz_grbm"] or float("nan")
        print(f"  {r['tag']:11s} line {r['line_ms_per_step']:8.3f} ms/step, clock {clock:.3f} GHz | profiled kernels "
              f"{r['profiled_kernel_ms_per_step_steady']:8.3f} ms (x{r['profiled_over_line']:.3f}{note}), GRBM clock of the profiled run {grbm:.3f}")


if __name__ == "__main__":
    main()
