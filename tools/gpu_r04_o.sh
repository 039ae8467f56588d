#!/bin/bash
# round 4: per-kernel A/B of builds on the C4 share and C3 (kernel trace) + whole-step rates
set -u
O=gpurun_out/r04x; mkdir -p $O
bash tools/x_kernel_ab.sh "--workload c4 --mask frame" "$@" 2>&1 | tee $O/ab_c4.txt
bash tools/x_kernel_ab.sh "--mask all" "$@" 2>&1 | tee $O/ab_all.txt
