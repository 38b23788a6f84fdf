#!/usr/bin/env python3
"""Per-kernel sums of the SQ counters of a rocprofv3 --pmc pass (tools/prof_round2_sq.sh): the 50 MB counter_collection.csv of a
bench step reduced to one line per kernel.   python tools/summarize_sq.py <counter_collection.csv> <out.csv>"""
import collections
import csv
import sys

NAMES = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("k_"):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
with open(sys.argv[2], "w") as f:
    f.write("kernel,dispatches (2 steps: warm-up + 1)," + ",".join(NAMES) + "\n")
    for k in sorted(acc, key=lambda k: -acc[k]["SQ_WAVE_CYCLES"]):
        f.write(k + "," + str(len(disp[k])) + "," + ",".join(str(int(acc[k][n])) for n in NAMES) + "\n")
