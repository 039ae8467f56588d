#!/bin/bash
# round 4: full GPU tests + fuzz on the product build, then per-kernel A/B of builds
set -u
O=gpurun_out/r04x; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; tail -3 $O/pytest_full.log; grep -E "^FAILED|^E  " $O/pytest_full.log | head -20
timeout 300 python tests/fuzz_gpu.py 90 51 > $O/fuzz_seed51.log 2>&1; tail -2 $O/fuzz_seed51.log
AFX_FUZZ_KERNEL=halfwave AFX_FUZZ_STATS=1 timeout 300 python tests/fuzz_gpu.py 60 52 > $O/fuzz_stats_seed52.log 2>&1; tail -2 $O/fuzz_stats_seed52.log
bash tools/x_kernel_ab.sh "--workload c4 --mask everything" "$@" 2>&1 | tee $O/ab_c4_everything.txt
bash tools/x_kernel_ab.sh "--workload c3 --mask frame" "$@" 2>&1 | tee $O/ab_c3.txt
