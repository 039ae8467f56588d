#!/bin/bash
# per-kernel time of the rhythm tracker for each workload of tools/rhythm_report.py (rocprofv3 --kernel-trace --stats)
set -u
O=$PWD/gpurun_out/${AFX_ROUND:-r03}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for w in short loops long long256; do
  rm -rf /tmp/prt_$w
  AFX_RT_ONLY=$w rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prt_$w -o p -- python3 $GRAFT_REPO_ROOT/tools/rhythm_report.py > /dev/null 2>&1
  f=$(find /tmp/prt_$w -name "*kernel_stats.csv" | head -1)
  echo "== $w"
  [ -n "$f" ] && { cp $f $O/rhythm_${w}_kernel_stats.csv; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "afx" in r["Name"]:
        print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e6:9.3f} ms total {float(r["TotalDurationNs"])/1e6:9.3f} ms')
PY
  }
done
