#!/bin/bash
# round 4: whole-step rates of builds on C3 / C4 (overlapping kernel chains: the sum of kernels says little)
set -u
O=gpurun_out/r04x; mkdir -p $O
for rep in 1 2; do
for L in "$@"; do
for W in "c3 --mask frame --steps 40 --warmup 15" "c4 --mask frame --steps 20 --warmup 5" "c4 --mask everything --steps 20 --warmup 5"; do
echo "== $L bench $W: $(AFX_LIBRARY=$GRAFT_REPO_ROOT/afec_amd/lib/$L/libafx_hip.so python bench.py --workload $W --no-cpu-baseline --no-single --no-spot-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), 'M frames/s', round(d['ms_per_step'],3))")"
done; done; done 2>&1 | tee $O/ab_steps.txt
AFX_LIBRARY=$GRAFT_REPO_ROOT/afec_amd/lib/$2/libafx_hip.so timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest_full.log 2>&1; tail -3 $O/pytest_full.log
