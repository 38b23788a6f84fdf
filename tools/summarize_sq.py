#!/usr/bin/env python3
"""Per kernel, from the --pmc passes tools/prof_sq.sh left under <dir>/p1 p2 p3: where the wave cycles go.
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave (MI355X_MICROARCH.md, rocprofv3 PMC slots):
WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY (issuing) ~ WAVE_CYCLES."""
import collections
import csv
import glob
import os
import sys

src = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
cnt = collections.defaultdict(int)
vgpr = {}
for p in sorted(glob.glob(os.path.join(src, "p*"))):
    for path in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"]
            if name.startswith("void "):
                name = name[5:]
            name = name.split("(")[0].split("<")[0]
            if not name.startswith("k_"):
                continue
            tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
            vgpr[name] = (r["VGPR_Count"], r["LDS_Block_Size"])
            if p.endswith("p1") and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
                cnt[name] += 1
rows = sorted(tot, key=lambda k: -dur[k])
print("kernel                launches      ms  vgpr    lds | waves  wave_cyc(G)  %wait_any %wait_inst %active | valu/wave salu/wave lds/wave vmem_rd/wave vmem_wr/wave branch/wave | issue slots used (active quad-cycles / (gui_active x 1024 SIMDs / 4))")
for k in rows[:24]:
    c = tot[k]
    w = max(c.get("SQ_WAVES", 0), 1)
    wc = max(c.get("SQ_WAVE_CYCLES", 0), 1)
    gui = c.get("GRBM_GUI_ACTIVE", 0)
    used = c.get("SQ_ACTIVE_INST_ANY", 0) / (gui * 1024 / 4) if gui else float("nan")
    print("%-22s %7d %7.2f %5s %6s | %9d %8.2f %9.1f %9.1f %7.1f | %8.0f %8.0f %7.0f %9.0f %9.0f %8.0f | %.3f   busy %.2f" % (
        k, cnt[k], dur[k], vgpr[k][0], vgpr[k][1], w, wc / 1e9, 100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc,
        100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc, c.get("SQ_INSTS_VALU", 0) / w, c.get("SQ_INSTS_SALU", 0) / w, c.get("SQ_INSTS_LDS", 0) / w,
        c.get("SQ_INSTS_VMEM_RD", 0) / w, c.get("SQ_INSTS_VMEM_WR", 0) / w, c.get("SQ_INSTS_BRANCH", 0) / w, used,
        c.get("SQ_BUSY_CYCLES", 0) / max(gui, 1)))
print()
for k in rows[:24]:
    print(k, {a: round(b) for a, b in sorted(tot[k].items())})
