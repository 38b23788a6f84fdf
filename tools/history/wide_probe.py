#!/usr/bin/env python3
"""GPU probe: the arithmetic coder (ARTB, order 1) on wide-alphabet streams of growing length - where does the time of
a 40 MB FORMAT/PL b250 go? Usage: python tools/wide_probe.py [copies]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from genozip_amd import synth
from genozip_amd.codec import Engine

E = Engine(device=0)
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def pl_like(seed, n):
    """bytes like the PIZ-format b250 of tools/config_bench.py's PL column: 80 % one-byte codes 0..99, the rest two-byte codes"""
    h = synth.u32(seed, n)
    one = (h % np.uint32(100)).astype(np.uint8)
    out = one.copy()
    two = (h % np.uint32(10)) >= np.uint32(8)
    out[two] = (0x80 | ((h >> np.uint32(8)) % np.uint32(0x20))).astype(np.uint8)[two]
    return out.tobytes()


for name, gen in (("PL-like", pl_like), ("uniform 200", lambda s, n: synth.uniform_bytes(s, n, 200).tobytes()),
                  ("skewed 200", lambda s, n: synth.skewed_bytes(s, n, 200, 0.5).tobytes())):
    for n in (2500000, 10000000, 40000000):
        d = gen(7, n)
        bufs = [E.mem.upload(d) for _ in range(copies)]
        tab, outs = E.make_stream_table([(16, b, len(d)) for b in bufs])
        E.compress_table(tab, copies); E.sync()
        E.profile(True, reset=True)
        E.compress_table(tab, copies); E.sync()
        E.profile(False)
        pr = E.profile_results()
        print("%-12s n %9d x%d  out %9d B  " % (name, n, copies, tab[0].out_len) + "  ".join("%s %.1f ms x%d" % (k.replace("k_", ""), v[0], v[1]) for k, v in sorted(pr.items(), key=lambda kv: -kv[1][0])[:4]), flush=True)
