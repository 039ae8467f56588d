#!/bin/bash
# round 4: the default bench line + the profiles of every configuration (kernel trace + PMC passes)
set -u
export AFX_ROUND=r04
O=gpurun_out/r04; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json; echo; tail -2 $O/bench_default.err
python tools/profile_config.py c2_f64 | head -3
python tools/profile_config.py star_f64 --mask star | head -4
python tools/profile_config.py all_f64 --mask all | head -6
python tools/profile_config.py frame_f64 --mask frame | head -12
AFX_PROF_WARMUP=12 AFX_PROF_STEPS=20 python tools/profile_config.py c3 --workload c3 --mask frame | head -12
python tools/profile_config.py c4 --workload c4 --mask frame | head -12
python tools/profile_config.py c4_everything --workload c4 --mask everything | head -16
