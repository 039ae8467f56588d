cd /tmp && export TMPDIR=/tmp
for P in 0.5 1.5 3 6 12; do
  rm -rf /tmp/kt3
  AFX_X_PROLOGUE=$P AFX_LIBRARY=$GRAFT_REPO_ROOT/afec_amd/lib/x_tune/libafx_hip.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-single --no-spot-check --no-side-stream --workload ${W:-c4} --mask frame > /tmp/kt3.log 2>&1
  echo "prologue $P: $(grep -o '"chunk_frames": [0-9]*' /tmp/kt3.log) $(grep -h 'frames32' $(find /tmp/kt3 -name '*kernel_stats.csv') | cut -d, -f1-4 | tr '\n' ' ')"
done
