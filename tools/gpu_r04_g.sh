#!/bin/bash
# round 4: bands_kernel variants: batch-independence + parity tests per build, then per-kernel A/B
set -u
O=gpurun_out/r04x; mkdir -p $O
for L in "$@"; do
  AFX_LIBRARY=$GRAFT_REPO_ROOT/afec_amd/lib/$L/libafx_hip.so timeout 900 python -m pytest tests -q -m gpu -k "invarian or depend or c4_share or split or contrast or parity or halfwave" > $O/pytest_$L.log 2>&1; echo "== $L: $(tail -1 $O/pytest_$L.log)"; grep -E "^FAILED|^E  " $O/pytest_$L.log | head -12
done
bash tools/x_kernel_ab.sh "--workload c4 --mask frame" "$@" 2>&1 | tee $O/ab_bands_c4.txt
bash tools/x_kernel_ab.sh "--mask all" "$@" 2>&1 | tee $O/ab_bands_all.txt
