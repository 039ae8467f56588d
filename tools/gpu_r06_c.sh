#!/bin/bash
# round 6, third GPU session: the magnitude class that leaves the upper spectrum out (frames32_kernel<6>) against the build
# before it, the GPU suite on it, and the band-sum layouts micro-benchmark
set -u
export AFX_ROUND=r06
O=gpurun_out/r06; mkdir -p $O
rm -f $O/ab.txt
bash tools/gpu.sh tests > $O/stage_tests.txt 2>&1; tail -4 $O/stage_tests.txt
bash tools/gpu.sh "ab=--mask all@after_cleanup,class6" > /dev/null 2>&1
bash tools/gpu.sh "ab=--workload c3 --mask all --steps 40 --warmup 15@after_cleanup,class6" > /dev/null 2>&1
bash tools/gpu.sh "ab=--workload c4 --mask all@after_cleanup,class6" > /dev/null 2>&1
cat $O/ab.txt
hipcc --offload-arch=gfx950 -O3 -o /tmp/bands_rows16 tools/ubench/bands_rows16.hip && for i in 1 2; do /tmp/bands_rows16; done | tee $O/ubench_bands_rows16.txt
python tools/profile_config.py all_f64 --mask all | head -6
python tools/profile_config.py c3_all --workload c3 --mask all | head -8
