#!/bin/bash
# round 6, fourth GPU session: where the magnitude class without the upper spectrum pays (batch size), the band-sum layouts
set -u
export AFX_ROUND=r06
O=gpurun_out/r06; mkdir -p $O
rm -f $O/ab.txt
hipcc --offload-arch=gfx950 -O3 -o /tmp/bands_rows16 tools/ubench/bands_rows16.hip && for i in 1 2 3; do /tmp/bands_rows16; done | tee $O/ubench_bands_rows16.txt
for FILES in 2000 4000 8000 25000; do
  echo "## c4 --files $FILES --mask all" >> $O/ab.txt
  bash tools/gpu.sh "ab=--workload c4 --files $FILES --mask all@after_cleanup,class6" > /dev/null 2>&1
done
echo "## c3 --mask all" >> $O/ab.txt
bash tools/gpu.sh "ab=--workload c3 --mask all@after_cleanup,class6" > /dev/null 2>&1
cat $O/ab.txt
