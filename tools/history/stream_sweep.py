#!/usr/bin/env python3
"""GPU probe (SURVEY 8d kernel-level bench): N concurrent 16 MiB quality streams through one codec call batch, MB/s in"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genozip_amd import workload as W
from genozip_amd.codec import Engine

E = Engine(device=0)
dev = torch.device("cuda", 0)
n = 16 << 20
base = W.quality_rows(W._TH(dev), 5, 0, n // 150 + 1, "div").reshape(-1)[:n].contiguous()
for codec, name in ((6, "RANB"), (16, "ARTB"), (17, "ARTW")):
    for count in (1, 16, 64):
        bufs = [base.clone() for _ in range(count)]
        tab, outs = E.make_stream_table([(codec, b, n) for b in bufs])
        E.compress_table(tab, count); E.sync()
        t0 = time.perf_counter()
        E.compress_table(tab, count); E.sync()
        dt = time.perf_counter() - t0
        print("%s  %3d x 16 MiB: %7.1f ms  %8.1f MB/s  (out %d B per stream)" % (name, count, dt * 1e3, count * n / dt / 1e6, tab[0].out_len), flush=True)
        del bufs, tab, outs
