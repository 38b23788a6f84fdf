#!/usr/bin/env python3
"""GPU probe: per-kernel time of the small striped arith streams of the FASTQ workload (QNAME x / y locals)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from genozip_amd import workload as W
from genozip_amd.codec import Engine

E = Engine(device=0)
lane, tile, x, y = W.name_fields(7, 0, 11000)
kinds = {"x u16 BE, ARTW": x.astype(">u2").tobytes(), "y u32 BE, ARTW": y.astype(">u4").tobytes(),
         "x, ARTB (unstriped)": x.astype(">u2").tobytes()}
copies = 176
for name, d in kinds.items():
    codec = 16 if "ARTB" in name else 17
    bufs = [E.mem.upload(d) for _ in range(copies)]
    tab, outs = E.make_stream_table([(codec, b, len(d)) for b in bufs])
    E.compress_table(tab, copies); E.sync()
    E.profile(True, reset=True)
    for _ in range(3):
        E.compress_table(tab, copies); E.sync()
    E.profile(False)
    pr = E.profile_results()
    print("%-24s in %6d out %6d B  " % (name, len(d), tab[0].out_len) + "  ".join("%s %.2f ms" % (k.replace("k_", ""), v[0] / 3) for k, v in sorted(pr.items(), key=lambda kv: -kv[1][0])[:7]))
