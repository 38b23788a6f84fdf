"""TEST INFRASTRUCTURE: generates tests/golden/assign_golden.json from the REFERENCE'S OWN src/codec.c compiled in place by
`make -C oracle ref` (oracle/_ref/libassignref.so + oracle/ref_assign_shim.c) - row a8 of SURVEY 8(a):

  sort   tables of { codec, size, clock } in trial order -> the order qsort (.., codec_assign_sorter) leaves them in (src/codec.c:128-173,
         :334; the C library's qsort of this container - glibc 2.35's merge sort - with the reference's comparator, which is not a strict
         weak order: the result depends on both), in the three modes of the sorter
  run    codec_assign_best_codec (src/codec.c:234-389) on a context section: the sample, the twelve candidates (the eight htscodecs forms
         by the reference's own coders; BZ2 / BSC / LZMA with scripted payload sizes), every trial's time from a scripted clock, the
         decision tree in front of the trials (a complex codec stays, --best's lock-in, VBlock 10's second look, the file's codec is
         inherited, < 50 bytes is not tested) and the commit to the file behind them (not from a small VBlock, not VBlock 1's local
         when the data type says its beginning may not be representative, unless it is the last) -> the codec returned, the file
         context's codec and counter afterwards, how many trials ran, and the four best rows of the sorted table (what --show-codec
         prints)

Only runs where /root/reference exists; the vectors are committed, the reference is not.     python tests/golden/make_assign_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import cases          # noqa: E402
import pyoracle       # noqa: E402


def main():
    R = pyoracle.AssignRef()
    out = {"sort": [], "run": []}
    for tests in cases.assign_sort_tables():
        for mode in (0, 1, 2):
            out["sort"].append({"mode": mode, "rows": [list(t) for t in tests], "order": [t[0] for t in R.sort(tests, mode)]})
    for c in cases.assign_run_cases():
        data = cases.assign_run_data(c)
        res, rows = R.run(c["in"], c["dict_id"], c["txt_len"], c["vb_size"], data, c["ticks"])
        out["run"].append({"case": c, "sha1": hashlib.sha1(data).hexdigest(), "out": res, "top4": [list(r) for r in rows]})
    path = os.path.join(HERE, "assign_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    tested = sum(1 for r in out["run"] if r["out"][4])
    print("%s: %d sorted tables, %d runs of codec_assign_best_codec (%d with trials), %d bytes" % (path, len(out["sort"]), len(out["run"]), tested, os.path.getsize(path)))


if __name__ == "__main__":
    main()
