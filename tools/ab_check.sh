#!/bin/bash
# GPU box: parity gate + interleaved timing of every variant library under afec_amd/lib/var/ (built by
# tools/ab_variants.sh build or by hand).  usage: tools/ab_check.sh [rounds] [-- bench args]
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
rounds=${1:-2}; shift || true; [ "${1:-}" = "--" ] && shift
for d in $ROOT/afec_amd/lib/var/*/; do
  name=$(basename $d)
  r=$(AFX_LIBRARY=$d/libafx_hip.so timeout 600 python -m pytest $ROOT/tests/test_gpu_halfwave.py $ROOT/tests/test_gpu_parity.py -x -q 2>&1 | tail -1)
  echo "parity $name: $r"
done
for r in $(seq $rounds); do
  for d in $ROOT/afec_amd/lib/var/*/; do
    name=$(basename $d)
    AFX_LIBRARY=$d/libafx_hip.so python $ROOT/bench.py --no-cpu-baseline --no-single --steps 20 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,1), 'Mframes/s', round(d['roofline']['launch_ms'],4), 'ms')"
  done
done
