#!/bin/bash
# round 4: counters of the C4 share with the current build (per-kernel PMC), full GPU tests
set -u
export AFX_ROUND=r04x
O=gpurun_out/r04x; mkdir -p $O
python tools/profile_config.py c4 --workload c4 --mask frame | head -14
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; tail -3 $O/pytest_full.log
