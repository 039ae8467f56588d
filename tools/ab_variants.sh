#!/bin/bash
# Build variants of libafx_hip.so with -D switches and time them interleaved on the GPU box.
# usage (here): tools/ab_variants.sh build "name1:-DX=1 -DY=0" "name2:..."   -> afec_amd/lib/var/<name>/libafx_hip.so
#       (GPU):  tools/ab_variants.sh run [rounds] -- bench args
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = "build" ]; then
  shift
  rm -rf $ROOT/afec_amd/lib/var
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    make -C $ROOT/afec_amd/csrc OUT=../lib/var/$name EXTRA="$flags" ../lib/var/$name/libafx_hip.so 2>&1 | grep -E "error|Error" 
    rm -rf $ROOT/afec_amd/lib/var/$name/obj
  done
  ls $ROOT/afec_amd/lib/var
else
  shift
  rounds=${1:-3}; shift; [ "${1:-}" = "--" ] && shift
  for r in $(seq $rounds); do
    for d in $ROOT/afec_amd/lib/var/*/; do
      name=$(basename $d)
      AFX_LIBRARY=$d/libafx_hip.so python $ROOT/bench.py --no-cpu-baseline --no-single --steps 20 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,1), 'Mframes/s', round(d['roofline']['launch_ms'],4), 'ms')"
    done
  done
fi
