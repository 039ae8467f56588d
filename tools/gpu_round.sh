#!/bin/bash
# One gpurun call: parity tests, A/B bench of the frame kernels, micro-benchmarks.  Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out/r02
O=gpurun_out/r02
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
for hw in 1 0 1 0; do
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
for u in instr_cost mfma_f64 valu_f64; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/ubench/$u.hip -o /tmp/$u 2>/dev/null && timeout 120 /tmp/$u > $O/ubench_$u.txt 2>&1
done
tail -3 $O/ubench_mfma_f64.txt
