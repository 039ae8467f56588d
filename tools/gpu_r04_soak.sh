#!/bin/bash
# round 4: longer soaks of the final build -- random batches against the oracle (both frame kernels), the crawl with the
# database on (row digests compared)
set -u
O=gpurun_out/r04; mkdir -p $O
timeout 500 python tests/fuzz_gpu.py 360 101 > $O/fuzz_seed101_360s.log 2>&1; tail -1 $O/fuzz_seed101_360s.log
AFX_FUZZ_KERNEL=halfwave AFX_FUZZ_STATS=1 timeout 400 python tests/fuzz_gpu.py 240 102 > $O/fuzz_stats_seed102_240s.log 2>&1; tail -1 $O/fuzz_stats_seed102_240s.log
AFX_FUZZ_KERNEL=wave64 AFX_FUZZ_STATS=1 timeout 300 python tests/fuzz_gpu.py 120 103 > $O/fuzz_stats_wave64_seed103_120s.log 2>&1; tail -1 $O/fuzz_stats_wave64_seed103_120s.log
timeout 420 python tools/crawl_soak.py 300 > $O/crawl_soak_300s.log 2>&1; tail -2 $O/crawl_soak_300s.log
