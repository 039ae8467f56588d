#!/bin/bash
# quick GPU check of the C2 path: parity tests that touch it, A/B bench, stage stamps, optional PMC summary
set -u
O=gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_robustness.py tests/test_gpu_halfwave.py -x -q 2>&1 | tail -3
for hw in 1 0 1; do
  timeout 300 python bench.py --frame-kernel $(case $hw in 0) echo wave64;; 2) echo halfwave;; *) echo auto;; esac) --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_hw${hw}.json 2>$O/bench_hw${hw}.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_hw${hw}.json").read().strip().splitlines()[-1])
    print("halfwave=$hw", round(d["value"]/1e6,1), "Mframes/s", "launch_ms", round(d["roofline"]["launch_ms"],4), "single", round((d["config"]["single_10k_frame_buffer_frames_per_s"] or 0)/1e6,1))
except Exception as e:
    print("halfwave=$hw bench failed", e); print(open("$O/bench_hw${hw}.err").read()[-800:])
PY
done
if [ -f afec_amd/lib/stamps/libafx_hip.so ]; then
  AFX_LIBRARY=afec_amd/lib/stamps/libafx_hip.so python bench.py --no-cpu-baseline --no-single --steps 5 --warmup 1 2>&1 | grep stamps
fi
if [ "${1:-}" = "pmc" ]; then
  bash tools/prof_pmc.sh c2hw frames32 | grep -E "SQ_ACTIVE_INST_VALU|SQ_INSTS_VALU|SQ_WAVE_CYCLES|SQ_WAIT|FETCH|LDS_BANK|LDS_IDX|frames32"
fi
