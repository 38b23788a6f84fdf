"""TEST INFRASTRUCTURE: generates tests/golden/order_golden.json from the REFERENCE'S OWN src/zip.c compiled in place by `make -C oracle ref`
(oracle/_ref/liborderref.so + oracle/ref_order_shim.c) - row a15 of SURVEY 8(a): the order in which zip_compress_all_contexts_local / _b250
(src/zip.c:247-342), called as zip_compress_one_vb calls them (:565-585) with one compute thread, hand a VBlock's context sections to the
section writer, on the random context tables of tests/cases.py::section_order_tables (did_i shuffled, every DEP level, locals the merge
makes, VBlock 1 and later ones).

Only runs where /root/reference exists; the vectors are committed, the reference is not.     python tests/golden/make_order_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import cases          # noqa: E402
import pyoracle       # noqa: E402


def main():
    R = pyoracle.OrderRef()
    n_tables = 120
    orders = [[[i, t] for i, t in R.order(ctxs, vb_i)] for ctxs, vb_i in cases.section_order_tables(n_tables)]
    path = os.path.join(HERE, "order_golden.json")
    with open(path, "w") as f:
        json.dump({"n_tables": n_tables, "orders": orders}, f, separators=(",", ":"))
    print("%s: %d tables, %d sections" % (path, len(orders), sum(len(o) for o in orders)))


if __name__ == "__main__":
    main()
