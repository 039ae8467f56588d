#!/bin/bash
# PMC counters of the rhythm kernels on one workload of tools/rhythm_report.py (separate pass, no trace domains)
set -u
W=${1:-short}
O=$PWD/gpurun_out/${AFX_ROUND:-r03}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prp_$W
AFX_RT_ONLY=$W rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d /tmp/prp_$W -o p -- python3 $GRAFT_REPO_ROOT/tools/rhythm_report.py > /dev/null 2>&1
python3 - "$W" "$O" <<'PY'
import csv, glob, sys, collections
w, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(f"/tmp/prp_{w}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rhythm" not in k and "onset" not in k: continue
        k = k.split("(")[0].replace("void afx::(anonymous namespace)::", "").replace("afx::(anonymous namespace)::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
with open(f"{out}/rhythm_{w}_pmc.csv", "w") as fo:
    for k, c in acc.items():
        d = max(n[k], 1)
        line = f"{k} dispatches {n[k]}: " + ", ".join(f"{a}={v / d:.4g}" for a, v in sorted(c.items()))
        print(line); fo.write(line + "\n")
        if c.get("GRBM_GUI_ACTIVE"):
            # VALU pipe busy fraction: 4 cycles per wave instruction over (GUI cycles x 256 CUs x 4 SIMDs)
            print(f"   VALU busy = {4 * c['SQ_ACTIVE_INST_VALU'] / (c['GRBM_GUI_ACTIVE'] / 8 * 256 * 4):.3f} of the SIMD issue slots (GRBM cycles / 8 XCDs)")
PY
