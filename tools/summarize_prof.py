#!/usr/bin/env python3
"""Turns the rocprofv3 outputs a gpurun call left under gpurun_out/prof/ into the small summaries committed under
profiles/: per-kernel stats (as rocprofv3 --stats wrote them), per-kernel HBM counters (FETCH_SIZE / WRITE_SIZE, from
separate --pmc passes as MI355X_MICROARCH.md prescribes) and profiles/<tag>_pmc.json which bench.py reads to fill
roofline.traffic."""
import collections
import csv
import json
import os
import shutil
import sys

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
pre = sys.argv[4] if len(sys.argv) > 4 else "r1"                     # the -o name the rocprofv3 runs were given
workload = json.loads(sys.argv[5]) if len(sys.argv) > 5 else None       # what bench.py was run on (it only takes traffic for the same)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", pre + "_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
for f in ("bench_stats.json", "bench_fetch.json", "bench_write.json"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, tag + "_" + f))


def agg(path, cname):
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if name.startswith("void "):                                  # (a kernel template: "void k_arith_model<true>(...)" - one kernel under one name)
            name = name[5:]
        if "<" in name.split("(")[0]:
            name = name.split("<")[0] + "(" + name.split("(", 1)[1] if "(" in name else name.split("<")[0]
        if r["Counter_Name"] == cname and name.startswith("k_"):
            out[name].append(float(r["Counter_Value"]))
    return out


fetch = agg(os.path.join(src, "fetch", pre + "_counter_collection.csv"), "FETCH_SIZE")
write = agg(os.path.join(src, "write", pre + "_counter_collection.csv"), "WRITE_SIZE")
rows = []
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, [0]), write.get(k, [0])
    rows.append({"kernel": k, "dispatches": max(len(f), len(w)), "FETCH_SIZE_KB_per_dispatch_mean": sum(f) / len(f),
                 "FETCH_SIZE_KB_max": max(f), "WRITE_SIZE_KB_per_dispatch_mean": sum(w) / len(w), "WRITE_SIZE_KB_max": max(w)})
with open(os.path.join(dst, tag + "_pmc_hbm.csv"), "w", newline="") as fh:
    wr = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
    wr.writeheader()
    wr.writerows(rows)
# per launch HBM traffic of each kernel in bytes: FETCH_SIZE counts 64 B per 128 B request on gfx950 (x2, calibrated
# for wide coalesced reads only - MI355X_MICROARCH.md section HBM); WRITE_SIZE is taken as reported (uncalibrated)
STEPS = 3   # the PMC passes run --steps 2 --warmup 1 with --pin-codecs: every dispatch belongs to one of 3 identical steps
pmc = {}
for r in rows:
    per_dispatch = 2 * 1024 * r["FETCH_SIZE_KB_per_dispatch_mean"] + 1024 * r["WRITE_SIZE_KB_per_dispatch_mean"]
    pmc[r["kernel"].split("(")[0]] = {"fetch_kb_reported": r["FETCH_SIZE_KB_per_dispatch_mean"], "write_kb_reported": r["WRITE_SIZE_KB_per_dispatch_mean"],
                                      "dispatches_per_step": r["dispatches"] / STEPS, "traffic_bytes": int(per_dispatch),
                                      "traffic_bytes_per_step": int(per_dispatch * r["dispatches"] / STEPS)}
json.dump({"workload": workload, "source": "GZ_NO_PIPELINE=1 GZ_ZIP_NO_OVERLAP=1 rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of: python bench.py --steps 2 --warmup 1 --no-cpu "
                     "--pin-codecs. (Counter collection serialises kernels, which the persistent chain kernel cannot live with: GZ_NO_PIPELINE runs the "
                     "same kernels over the same data one after the other - per step the bytes are the same, only the number of launches they are "
                     "spread over differs, hence traffic_bytes_per_step.)",
           "correction": "traffic = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 (gfx950 FETCH_SIZE half-count; WRITE_SIZE uncalibrated)", "kernels": pmc},
          open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1)
print(open(os.path.join(dst, tag + "_pmc_hbm.csv")).read())
