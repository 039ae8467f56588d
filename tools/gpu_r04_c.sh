#!/bin/bash
# round 4: quick check of a kernel change: the tests that cover it + the C4 profile (kernel trace pass only unless FULL=1)
set -u
export AFX_ROUND=r04
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest ${TESTS:-tests/test_gpu_neighbours.py tests/test_gpu_parity.py tests/test_gpu_end_to_end.py} -m gpu -q -x --timeout 200 --timeout-method thread 2>&1 | tail -8
if [ "${FULL:-0}" = "1" ]; then
  python tools/profile_config.py ${TAG:-c4} --workload c4 --mask ${MASK:-frame}
else
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-single --no-spot-check --no-side-stream --workload ${WORKLOAD:-c4} --mask ${MASK:-frame} > /tmp/kt.log 2>&1
  tail -c 300 /tmp/kt.log; echo
  python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/kt/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=0
for r in rows[:18]:
    n=r['Name'].replace('void afx::(anonymous namespace)::','').split('(')[0]
    ms=float(r['TotalDurationNs'])/13*1e-6
    tot+=ms
    print(f"{n:55s} calls {r['Calls']:>5s}  {ms:8.3f} ms/step")
print('sum', round(tot,3))
PY
fi
