"""TEST INFRASTRUCTURE: generates tests/golden/merge_golden.json from the REFERENCE'S OWN src/context.c compiled in place by
`make -C oracle ref` (oracle/_ref/libmergeref.so + oracle/ref_merge_shim.c) - the LOOP of row a4 of SURVEY 8(a): ctx_merge_in_one_vctx
(src/context.c:938-1079) with ctx_commit_node (:269-316), ctx_insert_to_dict (:50-71), ctx_drop_all_the_same (:795-871) over the
reference's own hash.c and seg.c, on the multi-VBlock word streams of tests/cases.py::merge_loop_scenarios: per merge the word index of
every node, the singletons' text that went to local, whether the b250 was dropped; per step the file context's dictionary, counts,
failed singletons and whether its dictionary may be removed. (The VBlock side - which snips are nodes, their counts - comes from the
oracle's ctx_seg_column, itself pinned by ctx_golden.json.)

Only runs where /root/reference exists; the vectors are committed, the reference is not.     python tests/golden/make_merge_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import parity         # noqa: E402
import pyoracle       # noqa: E402


def main():
    O = pyoracle.Oracle()
    n = 600
    got = parity.merge_loop_run(lambda est: pyoracle.MergeRef(est), O.ctx_seg_column, n)
    path = os.path.join(HERE, "merge_golden.json")
    with open(path, "w") as f:
        json.dump({"n": n, "scenarios": got}, f, indent=0)
    print("%s: %d scenarios, %d merges" % (path, len(got), sum(len(s) for s in got.values())))


if __name__ == "__main__":
    main()
