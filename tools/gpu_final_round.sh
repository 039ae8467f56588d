#!/bin/bash
# Final GPU round of a build: full GPU test suite, default bench line, per-configuration rocprofv3 summaries
# (tools/profile_config.py: kernel trace + separate PMC passes), per-kernel wait counters, the small measurements the
# design notes quote, soaks.  AFX_ROUND names the output directory under gpurun_out/; SOAK_S the seconds per soak.
set -u
export AFX_ROUND=${AFX_ROUND:-r04}
O=gpurun_out/$AFX_ROUND; mkdir -p $O
S=${SOAK_S:-300}
timeout 1500 python -m pytest tests -m gpu -q --timeout 200 --timeout-method thread > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
cp gpurun_out/parity_report.md $O/parity_report.md 2>/dev/null
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json; echo
python tools/profile_config.py c2_f64 | head -2
python tools/profile_config.py star_f64 --mask star | head -3
python tools/profile_config.py all_f64 --mask all | head -5
python tools/profile_config.py frame_f64 --mask frame | head -8
AFX_PROF_WARMUP=12 AFX_PROF_STEPS=20 python tools/profile_config.py c3 --workload c3 --mask frame | head -9
python tools/profile_config.py c4 --workload c4 --mask frame | head -9
python tools/profile_config.py c4_everything --workload c4 --mask everything | head -12
python tools/pmc_all_kernels.py c4 --workload c4 --mask frame > /dev/null 2>&1
python tools/pmc_all_kernels.py c2 > /dev/null 2>&1
python tools/single_buffer.py > $O/single_buffer.txt 2>&1; cat $O/single_buffer.txt
python tools/x_batchsize.py > $O/kernel_choice_by_batch_size.txt 2>&1; cat $O/kernel_choice_by_batch_size.txt
python tools/x_classes.py > $O/halfwave_classes_on_c4.txt 2>&1; cat $O/halfwave_classes_on_c4.txt
bash tools/prof_rhythm.sh 2>/dev/null | grep -v stats_kernel | tail -12
AFEC_CRAWL_TIMING=1 timeout 300 python tools/e2e_sweep.py 12500 8:512 6:512 > $O/e2e_cpu_accounting.txt 2>&1; grep -v "round trip =" $O/e2e_cpu_accounting.txt | tail -6
timeout 300 python tools/shards8_cpus.py > $O/shards8_busy_cpus.txt 2>&1; tail -4 $O/shards8_busy_cpus.txt
timeout $((S + 120)) python tests/fuzz_gpu.py $S 41 > $O/fuzz_seed41.log 2>&1; tail -2 $O/fuzz_seed41.log
AFX_FUZZ_KERNEL=halfwave AFX_FUZZ_STATS=1 timeout $((S / 2 + 120)) python tests/fuzz_gpu.py $((S / 2)) 42 > $O/fuzz_stats_seed42.log 2>&1; tail -2 $O/fuzz_stats_seed42.log
timeout $((S + 120)) python tools/crawl_soak.py $((S / 2)) > $O/crawl_soak.log 2>&1; tail -3 $O/crawl_soak.log
