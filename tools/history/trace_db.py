"""per-step timeline from a rocprofv3 --kernel-trace sqlite database (rocpd schema): the kernels of the LAST step of a bench.py
run, by start time, with gaps and overlaps - tools/trace_db.py <results.db> [n_steps]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % disp)]
scols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
namecol = "kernel_name" if "kernel_name" in scols else "display_name"
rows = list(cur.execute("select s.%s, d.start, d.end, d.queue_id from %s d join %s s on d.kernel_id = s.id order by d.start" % (namecol, disp, sym)))
print(len(rows), "dispatches")
# steps are delimited by k_nl_count (first kernel of a call)
starts = [i for i, r in enumerate(rows) if "k_nl_count" in r[0]]
if len(starts) < 2:
    starts = [0]
a = starts[-1]
step = rows[a:]
t0 = step[0][1]
agg = {}
for name, s, e, q in step:
    name = name.split("(")[0]
    if name.startswith("_Z"):                      # _Z<len><name>...
        k = 2
        while name[k].isdigit():
            k += 1
        name = name[k:k + int(name[2:k])]
    d = agg.setdefault(name, [0, 0.0, 1e18, 0])
    d[0] += 1; d[1] += (e - s) / 1e6; d[2] = min(d[2], (s - t0) / 1e6); d[3] = max(d[3], (e - t0) / 1e6)
print("last step: %.2f ms from first start to last end" % ((max(r[2] for r in step) - t0) / 1e6))
for name, (n, ms, first, last) in sorted(agg.items(), key=lambda kv: kv[1][2]):
    print("%-28s x%-4d %8.3f ms   first start %8.3f  last end %8.3f" % (name, n, ms, first, last))

if len(sys.argv) > 2:                               # every dispatch of the step that took longer than argv[2] ms
    lim = float(sys.argv[2])
    for name, s, e, q in step:
        if (e - s) / 1e6 >= lim:
            print("%10.3f  +%8.3f ms  q%-3s %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, name[:60]))
