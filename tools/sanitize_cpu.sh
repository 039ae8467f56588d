#!/bin/bash
# CPU sanitizer runs (GPU sanitizers are not available on the pool).  Writes profiles/<round>/sanitize_cpu.txt and
# profiles/<round>/tsan_crawler.txt.
#   1. ASan + UBSan: the WAV reader, the column encoder and the oracle (tests/sanitize/sanitize_main.cpp);
#   2. ASan + UBSan: the C-ABI's host code (afec_amd/csrc/afx_plan / workspace / batch_plan / batch_create / batch_run /
#      batch_fetch .cpp) on the mock device of tests/sanitize/hipstub + mock_kernels.cpp, fuzzed ragged batches
#      (tests/sanitize/fuzz_host_abi.cpp), and the crawler driver on the same stack;
#   3. TSan: the streaming sharded crawler + sqlite pool above that stack (tests/sanitize/tsan_crawler.cpp: G = 1, 2, 8
#      mock devices, injected batch failures, a lost device, an external abort), and the C-ABI fuzz with four threads on
#      shared plans.
set -u
cd "$(dirname "$0")/.."
R=${AFX_ROUND:-r06}
mkdir -p /tmp/afx_san profiles/$R
g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -ffp-contract=off \
    -o /tmp/afx_san/sanitize tests/sanitize/sanitize_main.cpp afec_amd/host/WaveFile.cpp afec_amd/host/DescriptorColumns.cpp \
    -x c oracle/afx_oracle.c oracle/afx_oracle_rhythm.c oracle/afx_oracle_resample.c -x none -lm
tests/sanitize/build.sh asan > /dev/null
tests/sanitize/build.sh tsan > /dev/null
{
  echo "# $(g++ --version | head -1); $(date -u +%F)"
  echo "# 1. g++ -fsanitize=address,undefined (-fno-sanitize-recover): tests/sanitize/sanitize_main.cpp + afec_amd/host/{WaveFile,DescriptorColumns}.cpp + oracle/*.c"
  ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 /tmp/afx_san/sanitize 2>&1
  echo "exit code $?"
  echo "# 2. the same flags: afec_amd/csrc/afx_{plan,workspace,batch_plan,batch_create,batch_run,batch_fetch}.cpp on the mock device"
  echo "#    (tests/sanitize/hipstub, mock_kernels.cpp), tests/sanitize/fuzz_host_abi.cpp <rounds> <seed> <threads>"
  for SEED in 1 2 3; do
    ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 /tmp/afx_san/fuzz_host_abi_asan 250 $SEED 1 2>&1 | tail -25
    echo "exit code ${PIPESTATUS[0]}"
  done
  echo "#    afec_amd/host/*.cpp (crawler, sqlite pool, WAV reader) above it: tests/sanitize/tsan_crawler.cpp built with address,undefined"
  ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 /tmp/afx_san/tsan_crawler_asan 600 2>&1 | tail -25
  echo "exit code ${PIPESTATUS[0]}"
} | tee profiles/$R/sanitize_cpu.txt
{
  echo "# $(g++ --version | head -1); $(date -u +%F)"
  echo "# g++ -fsanitize=thread: afec_amd/host/{Crawler,SampleAnalyser,DescriptorColumns,SqlitePool,WaveFile}.cpp + the C-ABI's host code"
  echo "# (afec_amd/csrc/afx_*.cpp) on the mock device; tests/sanitize/tsan_crawler.cpp <files>"
  TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1" /tmp/afx_san/tsan_crawler_tsan 600 > /tmp/afx_san/tsan_crawler.out 2>&1
  RC=$?
  echo "ThreadSanitizer warnings: $(grep -c 'WARNING: ThreadSanitizer' /tmp/afx_san/tsan_crawler.out)"
  grep -A12 'WARNING: ThreadSanitizer' /tmp/afx_san/tsan_crawler.out | head -60
  grep '^tsan_crawler' /tmp/afx_san/tsan_crawler.out
  echo "exit code $RC"
  echo "# the C-ABI fuzz with four threads on two shared plans: tests/sanitize/fuzz_host_abi.cpp 40 5 4"
  TSAN_OPTIONS="halt_on_error=0" /tmp/afx_san/fuzz_host_abi_tsan 40 5 4 > /tmp/afx_san/tsan_fuzz.out 2>&1
  RC=$?
  echo "ThreadSanitizer warnings: $(grep -c 'WARNING: ThreadSanitizer' /tmp/afx_san/tsan_fuzz.out)"
  grep -A12 'WARNING: ThreadSanitizer' /tmp/afx_san/tsan_fuzz.out | head -60
  grep '^fuzz_host_abi' /tmp/afx_san/tsan_fuzz.out
  echo "exit code $RC"
} | tee profiles/$R/tsan_crawler.txt
