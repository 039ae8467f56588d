#!/bin/bash
# round 4: full GPU tests on the product build, then per-kernel A/B of builds on the C4 share (per-frame set, everything) and --mask all
set -u
O=gpurun_out/r04x; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; tail -3 $O/pytest_full.log; grep -E "^FAILED|^E  " $O/pytest_full.log | head -20
bash tools/x_kernel_ab.sh "--workload c4 --mask frame" "$@" 2>&1 | tee $O/ab_c4.txt
bash tools/x_kernel_ab.sh "--workload c4 --mask everything" "$@" 2>&1 | tee $O/ab_c4_everything.txt
bash tools/x_kernel_ab.sh "--mask all" "$@" 2>&1 | tee $O/ab_all.txt
