#!/bin/bash
# Final GPU round of a build: full GPU test suite, default bench line, per-configuration rocprofv3 summaries
# (tools/profile_config.py: kernel trace + separate PMC passes), rhythm kernel stats / PMC, parity report.
set -u
O=gpurun_out/${AFX_ROUND:-r03}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --timeout 150 --timeout-method thread > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
cp gpurun_out/parity_report.md $O/parity_report.md 2>/dev/null
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json; echo
python tools/profile_config.py c2_f64
python tools/profile_config.py star_f64 --mask star
python tools/profile_config.py all_f64 --mask all
python tools/profile_config.py frame_f64 --mask frame
AFX_PROF_WARMUP=12 AFX_PROF_STEPS=20 python tools/profile_config.py c3 --workload c3 --mask frame
python tools/profile_config.py c4 --workload c4 --mask frame
python tools/profile_config.py c4_everything --workload c4 --mask everything
bash tools/prof_pmc.sh c2hw frames32 > /dev/null 2>&1
bash tools/prof_rhythm.sh | grep -v stats_kernel
bash tools/prof_rhythm_pmc.sh short
python tools/rhythm_report.py > /dev/null 2>&1      # last: the profiled runs above overwrite the report with partial ones
bash tools/prof_resample.sh 12500 1.0 48000 96000 22050 | tail -12      # sample-rate conversion: call times + per-kernel times
AFEC_CRAWL_TIMING=1 timeout 300 python tools/e2e_sweep.py 12500 8:512 6:512 > $O/e2e_cpu_accounting.txt 2>&1; grep -v "round trip =" $O/e2e_cpu_accounting.txt | tail -6
