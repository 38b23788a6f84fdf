"""The lead-in and the tail of one bench step from a rocprofv3 --kernel-trace CSV: what runs before the persistent chain of the long
streams starts, and after it ends (times in ms relative to the chain's start / end). Usage: python tools/timeline.py <kernel_trace.csv>"""
import csv
import sys


def main(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0].split(" ")[-1]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "")))
    rows.sort()
    chains = [r for r in rows if r[2].startswith("k_arith_chain") and r[1] - r[0] > 30e6]
    if not chains:
        print("no long chain dispatch found")
        return
    c = chains[-1]
    print("chain: %.3f ms" % ((c[1] - c[0]) / 1e6))
    lead = [r for r in rows if c[0] - 12e6 <= r[0] < c[0] + 0.2e6]
    # the step starts at the first newline scan before the chain
    starts = [r for r in lead if r[2].startswith("k_nl_count")]
    t0 = starts[-1][0] if starts else lead[0][0]
    print("lead-in: step start -> chain start = %.3f ms" % ((c[0] - t0) / 1e6))
    for s, e, n, q in lead:
        if s >= t0:
            print("  %8.3f .. %8.3f  (%7.3f)  q%-3s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n))
    tail = [r for r in rows if r[1] > c[1] - 0.3e6 and r[0] < c[1] + 8e6]
    print("tail: relative to the chain's end")
    last = c[1]
    for s, e, n, q in tail:
        if s - last > 3e6:
            break
        print("  %8.3f .. %8.3f  (%7.3f)  q%-3s %s" % ((s - c[1]) / 1e6, (e - c[1]) / 1e6, (e - s) / 1e6, q, n))
        last = max(last, e)
    print("tail: chain end -> last kernel end = %.3f ms" % ((last - c[1]) / 1e6))


if __name__ == "__main__":
    main(sys.argv[1])
