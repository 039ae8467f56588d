#!/bin/bash
# round 6, first GPU session: the new bench legs (sharded crawl, clock probe, chain objects) and what the probe costs
set -u
export AFX_ROUND=r06
O=gpurun_out/r06; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 1500 python -m pytest tests/test_gpu_bench.py tests/test_gpu_parity.py -m gpu -q -x --timeout 900 --timeout-method thread > $O/pytest_bench.log 2>&1
echo "pytest rc=$?"; grep -E "^FAILED|^E  " $O/pytest_bench.log | head -20; tail -3 $O/pytest_bench.log
for i in 1 2 3; do
  for P in "" "--no-clock-probe"; do
    timeout 300 python bench.py --steps 20 --warmup 5 --no-single --no-cpu-baseline --no-sharded-crawl $P 2>>$O/probe_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('probe' if d['roofline']['clock_probe'] else 'plain', round(d['value']/1e6,2), 'M frames/s', round(d['ms_per_step'],4), 'ms', d['roofline']['clock_probe'])" | tee -a $O/probe_ab.txt
  done
done
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -3 $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/bench_default.json').read().strip().splitlines()[-1])
c=d['config']
print('value', d['value']/1e6, 'ms', d['ms_per_step'], 'clock', d['roofline']['clock_probe'])
print('valu', d['roofline']['valu'])
for k in ('c3_frames_per_s','c3_spectral_set_frames_per_s','c4_share_frames_per_s','c4_share_at_crawler_shape'):
    v=c[k]; print(k, v and {kk:v[kk] for kk in ('frames_per_s','ms_per_step','frame_kernel','clock_ghz_in_run','batches')}, v and v['parity_spot_check'].get('passed'))
print('e2e', {k:v for k,v in c['end_to_end_host_driver'].items() if k!='workload'})
print('sharded', {k:v for k,v in c['sharded_crawl'].items() if k!='workload'})
PY
AFX_BENCH_DEVICE=0 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_gpus2.json 2> $O/bench_gpus2.err; echo "gpus2 rc=$?"; tail -3 $O/bench_gpus2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/bench_gpus2.json').read().strip().splitlines()[-1])
print('gpus2 value', d['value']/1e6, d['ranks'])
print('sharded', {k:v for k,v in d['config']['sharded_crawl'].items() if k!='workload'})
PY
