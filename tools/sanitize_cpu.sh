#!/bin/bash
# CPU sanitizer run (ASan + UBSan) of the host-side logic that needs no GPU and of the oracle; GPU sanitizers are not
# available on the pool.  Writes profiles/<round>/sanitize_cpu.txt.
set -eu
cd "$(dirname "$0")/.."
R=${AFX_ROUND:-r03}
mkdir -p /tmp/afx_san profiles/$R
g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -ffp-contract=off \
    -o /tmp/afx_san/sanitize tests/sanitize/sanitize_main.cpp afec_amd/host/WaveFile.cpp afec_amd/host/DescriptorColumns.cpp \
    -x c oracle/afx_oracle.c oracle/afx_oracle_rhythm.c oracle/afx_oracle_resample.c -x none -lm
{
  echo "# g++ -fsanitize=address,undefined (-fno-sanitize-recover): tests/sanitize/sanitize_main.cpp + afec_amd/host/{WaveFile,DescriptorColumns}.cpp + oracle/*.c"
  ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 /tmp/afx_san/sanitize 2>&1
  echo "exit code $?"
} | tee profiles/$R/sanitize_cpu.txt
