#!/bin/bash
# How long the GPU takes to reach steady clocks under the headline kernel: durations of 300 back-to-back launches of a
# 640 000-frame batch (rocprofv3 --kernel-trace) from an idle GPU, ten at a time; then the headline rate by batch size
# with the driver's --steps 20 --warmup 5, and the in-kernel clock of the stamps build at both ends.
O=$PWD/gpurun_out/${AFX_ROUND:-r03}; mkdir -p $O
ROOT=$PWD
{
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kts
rocprofv3 --kernel-trace --output-format csv -d /tmp/kts -o k -- python3 $ROOT/bench.py --buffers 64 --steps 300 --warmup 5 --no-cpu-baseline --no-single > /tmp/kts.log 2>&1
f=$(find /tmp/kts -name "*kernel_trace.csv" | head -1)
echo "# frames32_kernel<0>, 64 buffers (640 000 frames) per launch, 305 launches back to back from an idle GPU"
python3 - "$f" <<'PY'
import csv, sys
d=[(int(r["Start_Timestamp"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in csv.DictReader(open(sys.argv[1])) if "frames32_kernel" in r["Kernel_Name"]]
d.sort(); t0=d[0][0]
for i in range(0,len(d),10):
    seg=d[i:i+10]
    print(f"launches {i:3d}..{i+len(seg)-1:3d} from {(seg[0][0]-t0)/1e6:7.1f} ms: mean {sum(x[1] for x in seg)/len(seg)/1e6:.3f} ms = {640000/(sum(x[1] for x in seg)/len(seg)/1e9)/1e6:.0f} M frames/s")
PY
cd $ROOT
echo "# python bench.py --steps 20 --warmup 5 --buffers B (un-profiled)"
for b in 64 128 256 512 1024; do
  python bench.py --buffers $b --steps 20 --warmup 5 --no-cpu-baseline --no-single 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$b buffers:', round(d['value']/1e6,1), 'M frames/s,', round(d['ms_per_step'],3), 'ms per step, frac', round(d['roofline']['frac'],4))"
done
if [ -f afec_amd/lib/stamps/libafx_hip.so ]; then
  echo "# in-kernel core clock (s_memtime / s_memrealtime of the stamps build), last launch of --steps 20 --warmup 5"
  for b in 64 512; do echo -n "$b buffers: "; AFX_LIBRARY=$ROOT/afec_amd/lib/stamps/libafx_hip.so python bench.py --no-cpu-baseline --no-single --buffers $b --steps 20 --warmup 5 2>&1 | grep -E "core clock" | sed 's/.*span/span/'; done
fi
} | tee $O/clock_ramp.txt
