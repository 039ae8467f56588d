O=gpurun_out/r02; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity_report.py -x -q 2>&1 | tail -3
cp gpurun_out/parity_report.md $O/parity_report.md
for m in frame everything; do timeout 300 python bench.py --no-cpu-baseline --no-single --mask $m --steps 5 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m c2x64', round(d['value']/1e6,2),'Mframes/s', round(d['ms_per_step'],2),'ms')"; done
for m in frame everything; do timeout 300 python bench.py --no-cpu-baseline --no-single --workload c4 --mask $m --steps 5 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m c4', round(d['value']/1e6,2),'Mframes/s', round(d['ms_per_step'],2),'ms', d['config']['files_per_gpu_per_step'])"; done
timeout 300 python bench.py --no-cpu-baseline --workload c4 --end-to-end 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('e2e c4', d['value'], d['config'])"
