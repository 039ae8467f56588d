#!/usr/bin/env python3
"""Per-stage instruction budget of a kernel's hot loop, read from the ISA the compiler wrote.

usage: tools/isa_budget.py <file.s> <kernel-substring> [first_line last_line]

The hot loop is taken as the innermost basic-block run that contains the most `sched_barrier` markers (or the
explicit line range); it is cut into stages at every `; sched_barrier` comment and at every `s_waitcnt` that
follows a run of DS / VMEM operations.  For each stage the tool prints the number of VALU instructions by kind
(f64 fma/mul/add, conversions, moves / selects, DPP, permlane, transcendental), the DS and VMEM instructions,
the scalar instructions and an issue-cycle estimate (gfx950: f64 FMA / ADD / MUL 4 cycles per wave-instruction,
v_permlane*_swap 8, v_rsq_f64 / v_rcp_f64 16, 32-bit VALU 4, DPP moves 4).
"""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith("v_"):
        if "permlane" in op:
            return "permlane"
        if op.startswith(("v_rsq", "v_rcp", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
            return "trans"
        if op.endswith("_dpp") or "dpp" in op:
            return "dpp"
        if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_fmac_f64", "v_max_f64", "v_min_f64", "v_ldexp_f64",
                          "v_frexp", "v_pk_")):
            return "f64"
        if op.startswith("v_cvt"):
            return "cvt"
        if op.startswith(("v_mov", "v_cndmask", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane")):
            return "mov/sel"
        if op.startswith("v_cmp"):
            return "cmp"
        return "valu32"
    if op.startswith("ds_"):
        return "ds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


CYCLES = {"f64": 4, "permlane": 8, "trans": 16, "dpp": 4, "cvt": 4, "mov/sel": 4, "cmp": 4, "valu32": 4}


def main():
    path, kern = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    # the kernel's body
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and kern in l and re.match(r"^_Z\w+:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    if len(sys.argv) >= 5:
        lo, hi = int(sys.argv[3]) - 1, int(sys.argv[4])
    else:
        # basic blocks: the one (label .. next backward branch to it) with the most sched_barriers
        labels = {}
        for i in range(start, end):
            m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
            if m:
                labels[m.group(1)] = i
        best = None
        for i in range(start, end):
            m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", lines[i])
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                lo_, hi_ = labels[m.group(1)], i + 1
                nb = sum("sched_barrier" in l for l in lines[lo_:hi_])
                inner = not any(re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", lines[j]) and
                                labels.get(re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", lines[j]).group(1), 1 << 30) < j
                                and labels[re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", lines[j]).group(1)] > lo_
                                for j in range(lo_ + 1, hi_ - 1))
                if best is None or (nb, inner) > (best[0], best[1]):
                    best = (nb, inner, lo_, hi_)
        lo, hi = best[2], best[3]
    print("kernel %s: loop lines %d..%d of %s" % (kern, lo + 1, hi, path))
    stages = OrderedDict()
    cur = 0
    stages[cur] = Counter()
    ops = Counter()
    for l in lines[lo:hi]:
        s = l.strip()
        if "sched_barrier" in s:
            cur += 1
            stages[cur] = Counter()
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        if op.startswith("s_nop") or op.startswith("s_waitcnt"):
            stages[cur]["wait" if op.startswith("s_waitcnt") else "nop"] += 1
            continue
        c = classify(op)
        stages[cur][c] += 1
        ops[op] += 1
    kinds = ["f64", "cvt", "mov/sel", "cmp", "valu32", "dpp", "permlane", "trans", "ds", "vmem", "salu", "wait"]
    print("%-6s" % "stage" + "".join("%9s" % k for k in kinds) + "%9s%9s" % ("VALU", "cycles"))
    tot = Counter()
    for st, c in stages.items():
        valu = sum(c[k] for k in CYCLES)
        cyc = sum(c[k] * CYCLES[k] for k in CYCLES)
        print("%-6d" % st + "".join("%9d" % c[k] for k in kinds) + "%9d%9d" % (valu, cyc))
        tot.update(c)
    valu = sum(tot[k] for k in CYCLES)
    cyc = sum(tot[k] * CYCLES[k] for k in CYCLES)
    print("%-6s" % "all" + "".join("%9d" % tot[k] for k in kinds) + "%9d%9d" % (valu, cyc))
    print("top opcodes:", ", ".join("%s %d" % kv for kv in ops.most_common(28)))


if __name__ == "__main__":
    main()
