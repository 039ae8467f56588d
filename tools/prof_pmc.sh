#!/bin/bash
# rocprofv3 PMC passes + kernel trace of one bench.py configuration.
# usage: [PMC_PASSES='A B;C D'] [PMC_LIB=path] prof_pmc.sh <tag> <kernel-substring> [bench.py args...]; outputs gpurun_out/r03/<tag>_*.csv
set -u
TAG=$1; KSUB=$2; shift 2
ROOT=$(pwd)
O=$ROOT/gpurun_out/${AFX_ROUND:-r04}; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"
 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_VMEM"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
if [ -n "${PMC_PASSES:-}" ]; then IFS=';' read -ra PASSES <<< "$PMC_PASSES"; fi
[ -n "${PMC_LIB:-}" ] && export AFX_LIBRARY=$PMC_LIB
i=0
for P in "${PASSES[@]}"; do
  rm -rf /tmp/pmc_${TAG}_$i
  timeout 600 rocprofv3 --pmc $P --output-format csv -d /tmp/pmc_${TAG}_$i -o p -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-single --no-spot-check --no-side-stream "$@" > /tmp/pmc_${TAG}_$i.log 2>&1
  i=$((i+1))
done
python3 $ROOT/tools/summarize_pmc.py "$KSUB" $(find /tmp/pmc_${TAG}_* -name "*counter_collection.csv") > $O/${TAG}_pmc_summary.csv
rm -rf /tmp/kt_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -o k -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-single --no-spot-check --no-side-stream "$@" > /tmp/kt_$TAG.log 2>&1
cp $(find /tmp/kt_$TAG -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats.csv
cat $O/${TAG}_pmc_summary.csv
head -5 $O/${TAG}_kernel_stats.csv
