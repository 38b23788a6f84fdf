#!/usr/bin/env python3
"""GPU: decode throughput of the codecs (a14) on long streams: 30 MB transposed-DP-like bytes and a 6.9 MB quality stream"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from genozip_amd import synth, workload as W
from genozip_amd.codec import Engine
from genozip_amd.lib import CODEC_NAMES
E = Engine(device=0)
h = synth.u32(77, 3000 * 10000)
dp = (18 + (h % np.uint32(13)) + ((h >> np.uint32(8)) % np.uint32(13))).astype(np.uint8).tobytes()
qual = W.quality_rows(W._NP, 82, 0, 46000, "div").reshape(-1).astype(np.uint8).tobytes()
for name, data in (("dp30M", dp), ("qual6.9M", qual)):
    for c in (6, 7, 16, 17):
        z = E.compress_many([(c, data)])[0]
        E.uncompress_many([(c, z, len(data))])
        t0 = time.perf_counter()
        back = E.uncompress_many([(c, z, len(data))])
        dt = time.perf_counter() - t0
        assert back[0] == data
        print("%-9s %-5s %8.1f ms  %7.1f MB/s  (ratio %.3f)" % (name, CODEC_NAMES[c], dt * 1e3, len(data) / dt / 1e6, len(z) / len(data)), flush=True)
# many streams at once: 64 quality streams
items = [(16, qual)] * 64
zs = E.compress_many(items)
t0 = time.perf_counter(); back = E.uncompress_many([(16, z, len(qual)) for z in zs]); dt = time.perf_counter() - t0
print("64 x qual6.9M ARTB %8.1f ms  %7.1f MB/s" % (dt * 1e3, 64 * len(qual) / dt / 1e6))
