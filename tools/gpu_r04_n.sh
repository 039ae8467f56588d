#!/bin/bash
# round 4: full GPU tests + fuzz on the product build
set -u
O=gpurun_out/r04x; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; tail -3 $O/pytest_full.log; grep -E "^FAILED|^E  " $O/pytest_full.log | head
timeout 300 python tests/fuzz_gpu.py 60 81 > $O/fuzz_seed81.log 2>&1; tail -2 $O/fuzz_seed81.log
AFX_FUZZ_KERNEL=halfwave AFX_FUZZ_STATS=1 timeout 300 python tests/fuzz_gpu.py 45 82 > $O/fuzz_stats_seed82.log 2>&1; tail -2 $O/fuzz_stats_seed82.log
