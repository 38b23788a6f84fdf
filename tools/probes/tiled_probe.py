#!/usr/bin/env python3
"""k_arith_model_tiled on the device, smallest cases first: order-1 streams of a small alphabet through codec 16 (ARTB) against the oracle.
usage: tiled_probe.py [n ...]   (GZ_NO_PIPELINE=1: one piece; default: position chunks behind the persistent chain)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle                                                # noqa: E402
from genozip_amd import synth                                  # noqa: E402
from genozip_amd.codec import Engine                           # noqa: E402

E = Engine(device=0)
O = pyoracle.Oracle()
for a in sys.argv[1:] or ["100", "5000", "9000", "70000", "300000"]:
    n = int(a.lstrip("q"))
    data = (synth.quality_diverse(1, max(1, n // 150)) if a.startswith("q") else synth.markov_bytes(3, n, 40, 33)).tobytes()   # q<n>: quality scores, ~n of them
    t = time.time()
    got = E.compress_many([(16, data)])[0]
    dt = time.time() - t
    want = O.codec_compress(16, data)
    print("n %d: %s in %.3f s (%d bytes)" % (n, "OK" if got == want else "DIFFERS", dt, len(got)), flush=True)
