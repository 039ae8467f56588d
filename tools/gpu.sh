#!/bin/bash
# One GPU-box script for the stages a round keeps repeating (replaces round 4's fifteen gpu_r04_[a-o].sh one-offs).
# usage (from the repo root, normally under gpurun):  [AFX_ROUND=r06] bash tools/gpu.sh <stage> [<stage> ...]
#   tests[=<pytest args>]        the GPU suite (default: tests -m gpu)              -> $O/pytest_gpu.log
#   record                       the GPU suite with AFX_TOL_RECORD: the worst relative error of every descriptor over every
#                                check_gpu() call (tests/_tol.py; ceilings not enforced)  -> $O/observed_errors.json
#   bench[=<bench.py args>]      one bench.py line (default: the driver's)           -> $O/bench_default.json
#   gpus2                        `bench.py --gpus 2` run plainly, both ranks on device 0 -> $O/bench_gpus2.json
#   profile=<tag>[,<args>]       tools/profile_config.py <tag> <args>  (kernel trace + PMC passes)
#   profiles                     every configuration of profiles/kernel_profiles.json
#   lease                        ONE lease, one record: the driver's bench line, then every profile, then tools/lease_report.py
#                                (host name, the line's rates and in-run clocks beside the profiled kernel times of c2 / c3 / c4)
#                                                                                     -> $O/lease_report.{json,txt}
#   ab=<bench args>@<lib>,<lib>  steady per-kernel durations of builds under afec_amd/lib/<lib>/ (kernel trace only)
#   steps=<lib>,<lib>            whole-step rates of builds on the six bench configurations, two passes
#   fuzz=<seconds>,<seed>[,stats|halfwave]  tests/fuzz_gpu.py (stats: the half-wave statistics classes; halfwave: random masks, half-wave kernels forced)
#   slow                         the tests behind AFX_SLOW_TESTS=1                   -> $O/slow_tests.log
#   soak=<seconds>               tools/crawl_soak.py
#   small                        single_buffer / x_batchsize / x_classes / e2e accounting / shards8
# Every stage prints a short tail; the full outputs stay under gpurun_out/$AFX_ROUND/.
set -u
export AFX_ROUND=${AFX_ROUND:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export GRAFT_REPO_ROOT=$ROOT      # tools/x_kernel_ab.sh reads it (outside gpurun it is not set)
O=$ROOT/gpurun_out/$AFX_ROUND; mkdir -p $O
lib_of() { echo "$ROOT/afec_amd/lib/$1/libafx_hip.so"; }

for STAGE in "$@"; do
  NAME=${STAGE%%=*}; ARG=""; [ "$STAGE" != "$NAME" ] && ARG=${STAGE#*=}
  echo "#### $STAGE"
  case $NAME in
    tests)
      timeout 1500 python -m pytest ${ARG:-tests -m gpu} -q --timeout 300 --timeout-method thread > $O/pytest_gpu.log 2>&1
      echo "pytest rc=$?" >> $O/pytest_gpu.log; grep -E "^FAILED|^E  " $O/pytest_gpu.log | head -20; tail -3 $O/pytest_gpu.log ;;
    record)
      rm -f $O/observed_errors.json
      AFX_TOL_RECORD=$O/observed_errors.json timeout 1500 python -m pytest tests -m gpu -q --timeout 300 --timeout-method thread > $O/pytest_record.log 2>&1
      tail -2 $O/pytest_record.log; cat $O/observed_errors.json ;;
    bench)
      timeout 900 python bench.py $ARG > $O/bench_default.json 2> $O/bench_default.err
      echo "rc=$?"; tail -c 600 $O/bench_default.json; echo; tail -3 $O/bench_default.err ;;
    gpus2)
      AFX_BENCH_DEVICE=0 timeout 600 python bench.py --gpus 2 --no-cpu-baseline --no-single > $O/bench_gpus2.json 2> $O/bench_gpus2.err
      echo "rc=$?"; head -c 700 $O/bench_gpus2.json; echo; tail -3 $O/bench_gpus2.err ;;
    profile)
      TAG=${ARG%%,*}; REST=""; [ "$ARG" != "$TAG" ] && REST=${ARG#*,}
      python tools/profile_config.py $TAG $REST | head -16 ;;
    profiles)
      python tools/profile_config.py c2_f64 | head -3
      python tools/profile_config.py star_f64 --mask star | head -4
      python tools/profile_config.py all_f64 --mask all | head -6
      python tools/profile_config.py frame_f64 --mask frame | head -12
      AFX_PROF_WARMUP=12 AFX_PROF_STEPS=20 python tools/profile_config.py c3 --workload c3 --mask frame | head -12
      python tools/profile_config.py c4 --workload c4 --mask frame | head -12
      python tools/profile_config.py c4_everything --workload c4 --mask everything | head -16
      AFX_PROF_WARMUP=12 AFX_PROF_STEPS=20 python tools/profile_config.py c3_all --workload c3 --mask all | head -8
      python tools/profile_config.py c4_crawler --workload c4 --mask frame --batch-files 512 --in-flight 5 --frame-kernel wave64 | head -12 ;;
    lease)
      hostname > $O/lease_host.txt; date -u +%FT%TZ >> $O/lease_host.txt
      timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
      echo "bench rc=$?"; tail -2 $O/bench_default.err
      bash $0 profiles
      python tools/lease_report.py | tee $O/lease_report.txt ;;
    ab)
      BARGS=${ARG%%@*}; LIBS=${ARG#*@}
      AFX_ROUND=${AFX_ROUND}x bash tools/x_kernel_ab.sh "$BARGS" ${LIBS//,/ } 2>&1 | tee -a $O/ab.txt ;;
    steps)
      for rep in 1 2; do for L in ${ARG//,/ }; do
        for W in "c3 --mask frame --steps 40 --warmup 15" "c4 --mask frame --steps 20 --warmup 5" "c4 --mask everything --steps 20 --warmup 5" \
                 "c2 --mask all --steps 10 --warmup 3" "c2 --mask star --steps 20 --warmup 5" "c2 --steps 20 --warmup 5"; do
          echo "== $L bench $W: $(AFX_LIBRARY=$(lib_of $L) python bench.py --workload $W --no-cpu-baseline --no-single --no-spot-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), 'M frames/s', round(d['ms_per_step'],3), 'K', d['config'].get('chunk_frames'))")"
        done; done; done 2>&1 | tee -a $O/ab_steps.txt ;;
    slow)   # the tests behind AFX_SLOW_TESTS=1 (minutes of oracle time each)
      AFX_SLOW_TESTS=1 timeout 900 python -m pytest tests/test_gpu_resample.py -m gpu -q -k "2_28" --timeout 800 --timeout-method thread > $O/slow_tests.log 2>&1
      tail -3 $O/slow_tests.log ;;
    fuzz)
      IFS=, read -r SEC SEED KIND <<< "$ARG"
      if [ "${KIND:-}" = "stats" ]; then
        AFX_FUZZ_KERNEL=halfwave AFX_FUZZ_STATS=1 timeout $((SEC + 120)) python tests/fuzz_gpu.py $SEC $SEED > $O/fuzz_stats_seed$SEED.log 2>&1; tail -2 $O/fuzz_stats_seed$SEED.log
      elif [ "${KIND:-}" = "halfwave" ]; then
        AFX_FUZZ_KERNEL=halfwave timeout $((SEC + 120)) python tests/fuzz_gpu.py $SEC $SEED > $O/fuzz_halfwave_seed$SEED.log 2>&1; tail -2 $O/fuzz_halfwave_seed$SEED.log
      else
        timeout $((SEC + 120)) python tests/fuzz_gpu.py $SEC $SEED > $O/fuzz_seed$SEED.log 2>&1; tail -2 $O/fuzz_seed$SEED.log
      fi ;;
    soak)
      timeout $((ARG + 120)) python tools/crawl_soak.py $ARG > $O/crawl_soak.log 2>&1; tail -3 $O/crawl_soak.log ;;
    small)
      python tools/single_buffer.py > $O/single_buffer.txt 2>&1; cat $O/single_buffer.txt
      python tools/x_batchsize.py > $O/kernel_choice_by_batch_size.txt 2>&1; cat $O/kernel_choice_by_batch_size.txt
      python tools/x_classes.py > $O/halfwave_classes_on_c4.txt 2>&1; cat $O/halfwave_classes_on_c4.txt
      AFEC_CRAWL_TIMING=1 timeout 300 python tools/e2e_sweep.py 12500 8:512 6:512 > $O/e2e_cpu_accounting.txt 2>&1; grep -v "round trip =" $O/e2e_cpu_accounting.txt | tail -6
      timeout 300 python tools/shards8_cpus.py > $O/shards8_busy_cpus.txt 2>&1; tail -4 $O/shards8_busy_cpus.txt ;;
    *) echo "unknown stage $STAGE"; exit 2 ;;
  esac
done
