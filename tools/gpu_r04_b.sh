#!/bin/bash
# round 4, step B: GPU suite after the host-side changes + the default bench line (parity spot check, asm volatile)
set -u
export AFX_ROUND=r04
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 200 --timeout-method thread > $O/pytest_gpu_b.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_b.log; tail -30 $O/pytest_gpu_b.log
timeout 600 python bench.py > $O/bench_default_b.json 2> $O/bench_default_b.err; tail -c 1500 $O/bench_default_b.json; echo; tail -3 $O/bench_default_b.err
