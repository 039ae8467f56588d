#!/bin/bash
# A/B of resample kernel variants built under afec_amd/lib/var/<name>/ : parity of the conversion tests + kernel times
cd /tmp; export TMPDIR=/tmp
for d in $GRAFT_REPO_ROOT/afec_amd/lib/var/*/; do
  v=$(basename $d)
  export AFX_LIBRARY=$d/libafx_hip.so
  ok=$(cd $GRAFT_REPO_ROOT && timeout 300 python -m pytest tests/test_gpu_resample.py -m gpu -q -x --timeout 200 2>&1 | tail -1)
  rm -rf /tmp/prs_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prs_$v -o p -- python3 $GRAFT_REPO_ROOT/tools/resample_report.py 12500 1.0 48000 96000 22050 > /dev/null 2>&1
  f=$(find /tmp/prs_$v -name "*kernel_trace.csv" | head -1)
  echo "== $v: $ok"
  python3 - "$f" <<'PY'
import csv, sys, collections
per = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "resample_filter" in r["Kernel_Name"]:
        per["filter"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("   filter kernel ms per launch (48000 x3, 96000 x3, 22050 x3):", " ".join(f"{v:.2f}" for v in per["filter"]))
PY
done
