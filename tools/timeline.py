"""One bench step from a rocprofv3 --kernel-trace CSV: when the persistent chain of the long (QUAL) streams starts and ends relative
to the step's first kernel, and every dispatch longer than a threshold in between - which handle's queue it ran on, its grid.
Usage: python tools/timeline.py <kernel_trace.csv> [threshold ms, default 0.5] [only the first W and the last W ms of the step]"""
import csv
import sys


def main(path, thr, win=None):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0].split(" ")[-1]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r["Queue_Id"], r["Grid_Size_X"], r["Grid_Size_Y"]))
    rows.sort()
    chains = [r for r in rows if r[2] == "k_arith_chain"]
    if not chains:
        print("no chain dispatch found")
        return
    longest = max(r[1] - r[0] for r in chains)
    c = [r for r in chains if r[1] - r[0] > 0.8 * longest][-1]           # the last step's long chain
    first = [r for r in rows if r[2] == "k_nl_count" and r[0] < c[0]]
    t0 = first[-1][0] if first else c[0]
    nxt = [r for r in rows if r[2] == "k_nl_count" and r[0] > c[0]]
    t1 = nxt[0][0] if nxt else c[1] + 10_000_000
    print("step (first kernel to the next step's first kernel): %.3f ms" % ((t1 - t0) / 1e6))
    print("long chain: starts %.3f, ends %.3f (%.3f ms); last kernel of the step ends %.3f" %
          ((c[0] - t0) / 1e6, (c[1] - t0) / 1e6, (c[1] - c[0]) / 1e6, (max(r[1] for r in rows if t0 <= r[0] < t1) - t0) / 1e6))
    for s, e, n, q, gx, gy in rows:
        if win is not None and win * 1e6 < s - t0 and s < c[1] - win * 1e6:
            continue
        if t0 <= s < t1 and (e - s) > thr * 1e6 and n != "k_low_gate":
            print("  %8.3f .. %8.3f (%7.3f) q%-3s %s grid %s x %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n, gx, gy))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5, float(sys.argv[3]) if len(sys.argv) > 3 else None)
