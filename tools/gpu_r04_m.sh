#!/bin/bash
set -u
O=gpurun_out/r04x; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest_full.log 2>&1; tail -3 $O/pytest_full.log; grep -E "^FAILED|^E  " $O/pytest_full.log | head
bash tools/x_kernel_ab.sh "--workload c4 --mask frame" v12 v13 2>&1 | tee $O/ab_c4.txt
bash tools/x_kernel_ab.sh "--mask all" v12 v13 2>&1 | tee $O/ab_all.txt
bash tools/x_kernel_ab.sh "--mask star" v12 v13 2>&1 | tee $O/ab_star.txt
bash tools/x_ab.sh v13 x_lolds 2>&1 | tee $O/ab_headline_lolds.txt
