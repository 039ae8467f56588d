#!/bin/bash
# A/B of builds of the library on one box, per kernel: steady kernel durations (kernel trace only) of one bench
# configuration for each build under afec_amd/lib/<name>/.  usage: x_kernel_ab.sh "<bench args>" <name>...
set -u
ARGS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export AFX_ROUND=${AFX_ROUND:-r06x} AFX_PROF_TRACE_ONLY=1
for rep in 1 2; do
  for L in "$@"; do
    AFX_LIBRARY=$ROOT/afec_amd/lib/$L/libafx_hip.so python $ROOT/tools/profile_config.py ab_$L $ARGS > /dev/null 2>&1
    echo "== $L (pass $rep): $(python - <<PY
import json
d = json.load(open("$ROOT/gpurun_out/$AFX_ROUND/profile_ab_$L.json"))
ks = sorted(d["kernels"].items(), key=lambda kv: -kv[1].get("ms_per_step_steady", 0))
print("total %.3f ms | " % sum(k.get("ms_per_step_steady", 0) for _, k in ks) + ", ".join("%s %.3f" % (n.split("<")[0], k.get("ms_per_step_steady", 0)) for n, k in ks[:6]))
PY
)"
  done
done
