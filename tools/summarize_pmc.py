#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per dispatch of a kernel.

usage: summarize_pmc.py <kernel-substring> <dir-or-csv> [...]
"""
import collections
import csv
import glob
import os
import sys


def main():
    pat = sys.argv[1]
    acc = collections.defaultdict(list)
    dur = []
    for arg in sys.argv[2:]:
        files = [arg] if arg.endswith(".csv") else glob.glob(os.path.join(arg, "*counter_collection.csv"))
        for f in files:
            for r in csv.DictReader(open(f)):
                if pat in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    if "Start_Timestamp" in r and r["Counter_Name"]:
                        dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in sorted(acc):
        v = acc[k]
        print(f"{k},{len(v)},{sum(v)/len(v):.1f}")
    if dur:
        print(f"dispatch_ns_mean,{len(dur)},{sum(dur)/len(dur):.1f}")


if __name__ == "__main__":
    main()
