#!/bin/bash
# rocprofv3 PMC passes of the sample-rate conversion kernels (12 500 one-second files at 48 kHz); separate passes, no trace domain
set -u
ROOT=$(pwd); O=$ROOT/gpurun_out/${AFX_ROUND:-r03}; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for P in "${PASSES[@]}"; do
  rm -rf /tmp/pmc_rs_$i
  timeout 600 rocprofv3 --pmc $P --output-format csv -d /tmp/pmc_rs_$i -o p -- python3 $ROOT/tools/resample_report.py 12500 1.0 48000 > /tmp/pmc_rs_$i.log 2>&1
  i=$((i+1))
done
for k in resample_filter resample_plan resample_mix; do
  echo "== $k"
  python3 $ROOT/tools/summarize_pmc.py $k $(find /tmp/pmc_rs_* -name "*counter_collection.csv")
done | tee $O/resample_pmc_summary.csv
