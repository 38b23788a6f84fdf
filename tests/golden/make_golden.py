#!/usr/bin/env python3
"""Generates tests/golden/hts_golden.json -- run ONLY in the build container, where /root/reference exists.

Inputs are seeded synthetic streams (genozip_amd/synth.py); expected outputs are produced by the reference's own
vendored htscodecs sources, compiled in place into oracle/_ref/libhtsref.so (oracle/Makefile `make ref`), called
exactly like Genozip's codec_htscodecs.c does (order bytes 0x01/0x19/0x81/0x99, capacity = bound + 1 KB).
The fixture holds data only: (generator parameters -> length + sha1 of the codec output, plus the full hex for
short outputs). No reference source text is stored.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import pyoracle as po            # noqa: E402
from genozip_amd import synth    # noqa: E402

SIZES = [0, 1, 7, 8, 19, 20, 21, 49, 50, 1000, 99999, 500001]
NSYMS = [1, 2, 4, 5, 16, 17, 256]
ORDERS = [0x01, 0x19, 0x81, 0x99]
HEX_LIMIT = 192


def main():
    po.build(ref=True)
    ref = po.Ref()
    cases = []
    seed = 1000
    for n in SIZES:
        for kind in synth.STREAM_KINDS:
            for nsym in NSYMS:
                if kind == "u32be" and nsym != 256:
                    continue
                if n >= 99999 and nsym in (2, 5, 17):
                    continue  # keep the fixture and the CPU test time small
                seed += 1
                data = synth.stream(kind, seed, n, nsym).tobytes()
                assert len(data) == n
                for engine in ("rans", "arith"):
                    for order in ORDERS:
                        out = ref.hts_compress(engine, data, order)
                        c = {"kind": kind, "seed": seed, "n": n, "nsym": nsym, "engine": engine, "order": order,
                             "in_sha1": hashlib.sha1(data).hexdigest(), "out_len": len(out),
                             "out_sha1": hashlib.sha1(out).hexdigest()}
                        if len(out) <= HEX_LIMIT:
                            c["out_hex"] = out.hex()
                        cases.append(c)
    # one VB-sized stream (16 MiB) per codec: quality-like order-1 structure
    data = synth.markov_bytes(77, 16 << 20, 40, 33).tobytes()
    for engine in ("rans", "arith"):
        for order in ORDERS:
            out = ref.hts_compress(engine, data, order)
            cases.append({"kind": "markov40_q", "seed": 77, "n": len(data), "nsym": 40, "engine": engine, "order": order,
                          "in_sha1": hashlib.sha1(data).hexdigest(), "out_len": len(out),
                          "out_sha1": hashlib.sha1(out).hexdigest()})
    with open(os.path.join(HERE, "hts_golden.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "reference": "divonlan/genozip 15.0.86 src/htscodecs (compiled in place)",
                   "cases": cases}, f, indent=0, separators=(",", ":"))
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
