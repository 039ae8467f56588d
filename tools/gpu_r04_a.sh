#!/bin/bash
# round 4, step A: GPU test suite of the f32-arena build + C3 / C4 profiles
set -u
export AFX_ROUND=r04
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 150 --timeout-method thread > $O/pytest_gpu_a.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_a.log; tail -15 $O/pytest_gpu_a.log
AFX_PROF_WARMUP=12 AFX_PROF_STEPS=20 python tools/profile_config.py c3 --workload c3 --mask frame
python tools/profile_config.py c4 --workload c4 --mask frame
python tools/profile_config.py c4_everything --workload c4 --mask everything
