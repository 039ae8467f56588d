#!/bin/bash
# round 6, second GPU session: the whole GPU suite on the cleaned sources, kernels before / after the removal of the
# experiment switches (must be the same code: A/B within 1 %), the default line
set -u
export AFX_ROUND=r06
O=gpurun_out/r06; mkdir -p $O
bash tools/gpu.sh tests > $O/stage_tests.txt 2>&1; tail -4 $O/stage_tests.txt
bash tools/gpu.sh "ab=--workload c4 --mask frame@before_cleanup,after_cleanup" > /dev/null 2>&1
bash tools/gpu.sh "ab=--mask all@before_cleanup,after_cleanup" > /dev/null 2>&1
bash tools/gpu.sh "ab=--mask c2@before_cleanup,after_cleanup" > /dev/null 2>&1
bash tools/gpu.sh "ab=--mask star@before_cleanup,after_cleanup" > /dev/null 2>&1
cat $O/ab.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -3 $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/bench_default.json').read().strip().splitlines()[-1])
c=d['config']
print('value', d['value']/1e6, 'ms', d['ms_per_step'], 'clock', d['roofline']['clock_probe'])
for k in ('c3_frames_per_s','c3_spectral_set_frames_per_s','c4_share_frames_per_s','c4_share_at_crawler_shape'):
    v=c[k]; print(k, v and {kk:v[kk] for kk in ('frames_per_s','ms_per_step','frame_kernel','clock_ghz_in_run')}, v and v['parity_spot_check'].get('passed'))
print('star', c['star_descriptor_set_frames_per_s'], 'all', c['all_spectral_descriptors_frames_per_s'], 'single', c['single_10k_frame_buffer_frames_per_s'])
print('sharded', {k:v for k,v in c['sharded_crawl'].items() if k in ('files_per_s','busy_host_cpus','workers_per_device')}, 'e2e', c['end_to_end_host_driver']['files_per_s'])
PY
