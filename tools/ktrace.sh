#!/bin/bash
# kernel-trace summary of one bench configuration on the GPU box: tools/ktrace.sh <tag> [bench args]
set -u
TAG=$1; shift
ROOT=$(pwd); O=$ROOT/gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/kt_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -o k -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-single "$@" > /tmp/kt_$TAG.log 2>&1
f=$(find /tmp/kt_$TAG -name "*kernel_stats.csv" | head -1)
cp $f $O/${TAG}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "afx::" in r["Name"]:
        print("%-70s calls %4s avg %9.1f us min %9.1f" % (r["Name"].replace("void afx::(anonymous namespace)::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
