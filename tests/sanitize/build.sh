#!/bin/bash
# tests/sanitize/build.sh <asan|tsan|plain> [target ...] -> /tmp/afx_san/<target>_<kind>
#   fuzz_host_abi  the C-ABI's host code (afec_amd/csrc/afx_*.cpp) on the mock device, fuzzed ragged batches
#   tsan_crawler   the streaming sharded crawler + sqlite pool (afec_amd/host) above the real C-ABI host code, mock device
#   host_lib       (on request only) the same sources as a shared library with libafx_host.so's entry points, for the
#                  Python side of the sharded crawl (afec_amd/hostlib.py, bench.py) on 8 mock devices
# TEST INFRASTRUCTURE: the HIP runtime is tests/sanitize/hipstub (host memory), the kernels are mock_kernels.cpp.
set -eu
cd "$(dirname "$0")/../.."
KIND=${1:-asan}; shift || true
TARGETS=${*:-fuzz_host_abi tsan_crawler}
case $KIND in
  asan) SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined" ;;
  tsan) SAN="-fsanitize=thread" ;;
  *) SAN="" ;;
esac
mkdir -p /tmp/afx_san
ABI="afec_amd/csrc/afx_plan.cpp afec_amd/csrc/afx_workspace.cpp afec_amd/csrc/afx_batch_plan.cpp afec_amd/csrc/afx_batch_create.cpp afec_amd/csrc/afx_batch_run.cpp afec_amd/csrc/afx_batch_fetch.cpp"
MOCK="tests/sanitize/mock_kernels.cpp tests/sanitize/hipstub/hip_stub.cpp"
HOST="afec_amd/host/Crawler.cpp afec_amd/host/SampleAnalyser.cpp afec_amd/host/DescriptorColumns.cpp afec_amd/host/SqlitePool.cpp afec_amd/host/WaveFile.cpp afec_amd/host/SyntheticInput.cpp"
FLAGS="-std=c++17 -O1 -g -fno-omit-frame-pointer $SAN -Itests/sanitize/hipstub -Iinclude -DAFX_SRC_HASH=\"mock\""
for T in $TARGETS; do
  case $T in
    fuzz_host_abi) g++ $FLAGS -o /tmp/afx_san/${T}_$KIND tests/sanitize/fuzz_host_abi.cpp $MOCK $ABI -lpthread ;;
    tsan_crawler)  g++ $FLAGS -o /tmp/afx_san/${T}_$KIND tests/sanitize/tsan_crawler.cpp $MOCK $ABI $HOST -lpthread -ldl ;;
    host_lib)      g++ $FLAGS -shared -fPIC -o /tmp/afx_san/libafx_host_mock_$KIND.so $MOCK $ABI $HOST -lpthread -ldl
                   echo /tmp/afx_san/libafx_host_mock_$KIND.so; continue ;;
    *) echo "unknown target $T"; exit 2 ;;
  esac
  echo /tmp/afx_san/${T}_$KIND
done
