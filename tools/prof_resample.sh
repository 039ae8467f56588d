#!/bin/bash
# per-kernel time of the sample-rate conversion (rocprofv3 --kernel-trace --stats around tools/resample_report.py)
set -u
O=$PWD/gpurun_out/${AFX_ROUND:-r03}; mkdir -p $O
python3 tools/resample_report.py "$@" | tee $O/resample_report.txt
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prs
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prs -o p -- python3 $GRAFT_REPO_ROOT/tools/resample_report.py "$@" > /dev/null 2>&1
f=$(find /tmp/prs -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && { cp $f $O/resample_kernel_stats.csv; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "afx" in r["Name"] and ("resample" in r["Name"] or "load_" in r["Name"]):
        print(f'{r["Name"][:80]:80s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e6:9.3f} ms min {float(r["MinNs"])/1e6:9.3f} max {float(r["MaxNs"])/1e6:9.3f}')
PY
}
