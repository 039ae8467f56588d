#!/bin/bash
# GPU timeline of the streaming host driver: union of kernel intervals and of copy intervals vs the wall time of the crawl
set -u
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ctl
AFEC_CRAWL_TIMING=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ctl -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload c4 --end-to-end --files ${1:-50000} --workers ${2:-3} --files-per-batch ${3:-256} 2>&1 | grep "afec crawl" | tail -2
python3 - <<'PY'
import csv, glob
def union(iv):
    iv.sort(); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
k = []
for f in glob.glob("/tmp/ctl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): k.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
c = []
for f in glob.glob("/tmp/ctl/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): c.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "")))
# the last crawl of the run (bench repeats it): take the final 40 % of the kernel time span
t0, t1 = min(a for a, _, _ in k), max(b for _, b, _ in k)
lo = t1 - 0.30 * (t1 - t0)
kk = [(a, b) for a, b, _ in k if a >= lo]; cc = [(a, b) for a, b, _ in c if a >= lo]
span = max(b for _, b in kk) - min(a for a, _ in kk)
print(f"window {span / 1e6:.1f} ms: kernels busy (union) {union(kk) / 1e6:.1f} ms = {union(kk) / span:.2f}, sum of kernel durations {sum(b - a for a, b in kk) / 1e6:.1f} ms, "
      f"copies busy (union) {union(cc) / 1e6:.1f} ms, {len(kk)} kernels, {len(cc)} copies")
import collections
d = collections.Counter(); n = collections.Counter()
import re
def short(name):
    while True:
        t = re.sub(r"<[^<>]*>", "", name)
        if t == name: break
        name = t
    return name.replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1].split(" ")[-1]
for a, b, name in k:
    if a >= lo: d[short(name)] += b - a; n[short(name)] += 1
for name, t in d.most_common(40): print(f"   {name:42s} {t / 1e6:8.2f} ms in {n[name]} launches, {t / n[name] / 1e3:7.1f} us each")
both = kk + cc
print(f"kernels or copies busy (union) {union(both) / 1e6:.1f} ms")
PY

if [ -n "${KEEP_TRACE:-}" ]; then mkdir -p $GRAFT_REPO_ROOT/gpurun_out/ctl; cp $(find /tmp/ctl -name "*kernel_trace.csv") $GRAFT_REPO_ROOT/gpurun_out/ctl/kernel_trace.csv; cp $(find /tmp/ctl -name "*memory_copy_trace.csv") $GRAFT_REPO_ROOT/gpurun_out/ctl/memory_copy_trace.csv; fi
