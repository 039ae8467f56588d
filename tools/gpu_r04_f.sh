#!/bin/bash
# round 4: bands_kernel variants -- parity tests on the product build, then per-kernel A/B
set -u
O=gpurun_out/r04x; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_bands.log 2>&1; tail -5 $O/pytest_bands.log
bash tools/x_kernel_ab.sh "--workload c4 --mask frame" "$@" 2>&1 | tee $O/ab_bands_c4.txt
