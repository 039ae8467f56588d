#!/bin/bash
# Host-side check on the GPU box: CPU accounting of the crawl (AFEC_CRAWL_TIMING), the crawler tests, the end-to-end
# object of the default bench line.
O=gpurun_out/${AFX_ROUND:-r03}b; mkdir -p $O
AFEC_CRAWL_TIMING=1 timeout 300 python tools/e2e_sweep.py 12500 8:512 6:512 > $O/e2e_cpu.txt 2>&1
grep -v "round trip =" $O/e2e_cpu.txt | tail -14
timeout 400 python -m pytest tests/test_gpu_crawler.py tests/test_real_audio.py tests/test_host_cpp.py -m gpu -q -x 2>&1 | tail -3
timeout 500 python bench.py --no-cpu-baseline 2> $O/bench_e2e.err > $O/bench_e2e.json
python - <<PY
import json
b = json.loads(open("$O/bench_e2e.json").read().strip().splitlines()[-1])
print(b["value"], b["roofline"]["bound"], b["roofline"].get("profile_stale"))
print(json.dumps(b["config"]["end_to_end_host_driver"], indent=0))
PY
