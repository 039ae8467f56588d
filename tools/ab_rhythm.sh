#!/bin/bash
# A/B of rhythm kernel variants built under afec_amd/lib/var/<name>/ : the rhythm parity tests + the kernels' times on the
# C4 share (bench.py --workload c4 --mask everything under rocprofv3 --kernel-trace)
cd /tmp; export TMPDIR=/tmp
for d in $GRAFT_REPO_ROOT/afec_amd/lib/var/*/; do
  v=$(basename $d)
  export AFX_LIBRARY=$d/libafx_hip.so
  ok=$(cd $GRAFT_REPO_ROOT && timeout 600 python -m pytest tests/test_gpu_rhythm.py tests/test_real_audio.py -m gpu -q -x --timeout 300 2>&1 | tail -1)
  rm -rf /tmp/prt_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prt_$v -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-single --no-side-stream --workload c4 --mask everything > /dev/null 2>&1
  f=$(find /tmp/prt_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v: $ok"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "onset_function" in r["Name"] or "rhythm_post" in r["Name"]:
        print(f'   {r["Name"].replace("void afx::(anonymous namespace)::", "")[:50]:50s} avg {float(r["AverageNs"])/1e6:8.3f} ms min {float(r["MinNs"])/1e6:8.3f}')
PY
done
