#!/bin/bash
# upper bound of what a cheaper layout of the band kernel's plain reductions can give: a build WITHOUT them (wrong results,
# timing only: afec_amd/lib/ablate_sums, built outside the tree) against the current one
set -u
export AFX_ROUND=r06
O=gpurun_out/r06; mkdir -p $O; rm -f $O/ab.txt
bash tools/gpu.sh "ab=--workload c4 --mask frame@current,ablate_sums" > /dev/null 2>&1
bash tools/gpu.sh "ab=--mask all@current,ablate_sums" > /dev/null 2>&1
cat $O/ab.txt
