#!/usr/bin/env python3
"""GPU probe: per-kernel time of the arithmetic coder on streams of different character (how often neighbouring
symbols of a model overtake each other decides whether the batched model update applies)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from genozip_amd import synth, workload as W
from genozip_amd.codec import Engine

E = Engine(device=0)
n, copies = 1800000, 64
kinds = {
    "qual_div (random walk: ties)": W.quality_rows(W._TH(torch.device("cuda", 0)), 5, 0, n // 150, "div").cpu().numpy().tobytes(),
    "skewed iid 40 symbols": synth.skewed_bytes(1, n, 40, 0.85, 33).tobytes(),
    "skewed iid 8 symbols": synth.skewed_bytes(2, n, 8, 0.6, 48).tobytes(),
    "qual_bin (4 levels)": W.quality_rows(W._TH(torch.device("cuda", 0)), 5, 0, n // 150, "bin").cpu().numpy().tobytes(),
}
only = sys.argv[1] if len(sys.argv) > 1 else ""
for name, d in kinds.items():
    if only not in name: continue
    bufs = [E.mem.upload(d) for _ in range(copies)]
    tab, outs = E.make_stream_table([(16, b, len(d)) for b in bufs])
    E.compress_table(tab, copies); E.sync()
    E.profile(True, reset=True)
    for _ in range(3):
        E.compress_table(tab, copies); E.sync()
    E.profile(False)
    pr = E.profile_results()
    print("%-32s out %7d B  " % (name, tab[0].out_len) + "  ".join("%s %.1f ms" % (k.replace("k_", ""), v[0] / 3) for k, v in sorted(pr.items(), key=lambda kv: -kv[1][0])[:5]))
