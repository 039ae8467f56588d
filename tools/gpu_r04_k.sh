#!/bin/bash
# round 4: whole-step rates of builds (overlapping kernel chains: the sum of kernels says little)
set -u
O=gpurun_out/r04x; mkdir -p $O
for rep in 1 2; do
for L in "$@"; do
for W in "c3 --mask frame --steps 40 --warmup 15" "c4 --mask frame --steps 20 --warmup 5" "c4 --mask everything --steps 20 --warmup 5" "c2 --mask all --steps 10 --warmup 3" "c2 --mask star --steps 20 --warmup 5" "c2 --steps 20 --warmup 5"; do
echo "== $L bench $W: $(AFX_LIBRARY=$GRAFT_REPO_ROOT/afec_amd/lib/$L/libafx_hip.so python bench.py --workload $W --no-cpu-baseline --no-single --no-spot-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), 'M frames/s', round(d['ms_per_step'],3), 'K', d['config'].get('chunk_frames'))")"
done; done; done 2>&1 | tee $O/ab_steps.txt
