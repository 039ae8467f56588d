#!/usr/bin/env python3
"""SQ counters of every kernel of one bench.py configuration: where do the waves' cycles go?

usage (on the GPU box, from the repo root): pmc_all_kernels.py <tag> [bench.py args ...]   -> gpurun_out/<round>/<tag>_pmc_kernels.csv

Three rocprofv3 --pmc passes (never together with a trace domain); per kernel and dispatch: VALU instructions, VALU busy
cycles, wave cycles, and the share of its waves' lifetime spent waiting for any instruction result / for LDS / for
vector memory, with the clock from GRBM_GUI_ACTIVE."""
import collections
import csv
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("AFX_ROUND", "r06")
PASSES = ["SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY",
          "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU",
          "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"]


def short(name):
    return name.replace("void afx::(anonymous namespace)::", "").replace("afx::(anonymous namespace)::", "").split("(")[0]


def main():
    tag, bench_args = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for i, counters in enumerate(PASSES):
        d = f"/tmp/pmck_{tag}_{i}"
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3", "--pmc"] + counters.split() + ["--output-format", "csv", "-d", d, "-o", "p", "--", "python3",
                                                          os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--no-cpu-baseline",
                                                          "--no-single", "--no-spot-check", "--no-side-stream", "--no-clock-probe", "--no-sharded-crawl"] + bench_args
        subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "afx::" not in r["Kernel_Name"]:
                    continue
                k = short(r["Kernel_Name"])
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if i == 0 and r["Counter_Name"] == "SQ_WAVES":
                    dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out_dir = os.path.join(ROOT, "gpurun_out", ROUND)
    os.makedirs(out_dir, exist_ok=True)
    rows = []
    for k, c in acc.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        wc = m.get("SQ_WAVE_CYCLES", 0) or 1
        ns = sum(dur[k]) / max(1, len(dur[k]))
        rows.append((ns, k, m, wc))
    rows.sort(reverse=True)
    with open(os.path.join(out_dir, f"{tag}_pmc_kernels.csv"), "w") as fo:
        hdr = "kernel,dispatch_us,valu_insts_M,valu_busy_of_wave_cycles,wait_any,wait_inst_any,wait_inst_lds,wait_inst_vmem,lds_active,lds_bank_conflict_of_lds_active,salu_insts_M,clock_ghz"
        fo.write(hdr + "\n")
        print(hdr)
        for ns, k, m, wc in rows:
            clock = m.get("GRBM_GUI_ACTIVE", 0) / 8.0 / ns if ns else 0
            line = (f"{k},{ns * 1e-3:.1f},{m.get('SQ_INSTS_VALU', 0) * 1e-6:.2f},{4 * m.get('SQ_ACTIVE_INST_VALU', 0) / wc:.3f},"
                    f"{4 * m.get('SQ_WAIT_ANY', 0) / wc:.3f},{4 * m.get('SQ_WAIT_INST_ANY', 0) / wc:.3f},{4 * m.get('SQ_WAIT_INST_LDS', 0) / wc:.3f},"
                    f"{4 * m.get('SQ_WAIT_INST_VMEM', 0) / wc:.3f},{4 * m.get('SQ_ACTIVE_INST_LDS', 0) / wc:.3f},"
                    f"{m.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, m.get('SQ_LDS_IDX_ACTIVE', 0)):.3f},{m.get('SQ_INSTS_SALU', 0) * 1e-6:.2f},{clock:.2f}")
            fo.write(line + "\n")
            print(line)


if __name__ == "__main__":
    main()
