#!/bin/bash
# round 4: full GPU tests + fuzz on the product build, then per-kernel A/B of builds
set -u
O=gpurun_out/r04x; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; tail -3 $O/pytest_full.log; grep -E "^FAILED|^E  " $O/pytest_full.log | head
AFX_FUZZ_KERNEL=halfwave AFX_FUZZ_STATS=1 timeout 300 python tests/fuzz_gpu.py 45 93 > $O/fuzz_stats_seed93.log 2>&1; tail -2 $O/fuzz_stats_seed93.log
bash tools/x_kernel_ab.sh "--mask star" "$@" 2>&1 | tee $O/ab_star.txt
