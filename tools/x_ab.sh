#!/bin/bash
# A/B of two builds of the library on one box: headline bench (no CPU baseline, no secondaries), alternating
for i in 1 2 3; do
  for L in "$@"; do
    v=$(AFX_LIBRARY=$GRAFT_REPO_ROOT/afec_amd/lib/$L/libafx_hip.so python bench.py --no-cpu-baseline --no-single --no-spot-check --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), round(d['roofline']['launch_ms'],3))")
    echo "$L: $v"
  done
done
