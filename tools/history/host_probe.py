#!/usr/bin/env python3
"""GPU probe: how much of a bench step is host time (planning / uploads / launches before the GPU gets its first kernel)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from genozip_amd.codec import Engine
E = Engine(device=0)
dev = torch.device("cuda", 0)
wl = bench.RankWorkload(E, 1000000, 4 << 20, "div", seed_base=1, device=dev)
wl.codecs = {"QUAL": 16, "Q1NAME": 8, "Q2NAME": 16, "Q3NAME": 17, "Q4NAME": 17}
wl.step_prepare(); E.sync(); wl.build_tables()
for _ in range(2): wl.step()
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); wl.step_prepare(); t1 = time.perf_counter()
    E.vb_compress_table(wl.vtab, len(wl.vbs)); t2 = time.perf_counter()
    E.sync(); t3 = time.perf_counter()
    print("prepare (enqueue) %.2f ms   vb_compress_batch (plan + enqueue) %.2f ms   sync (wait + read back) %.2f ms   total %.2f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t3-t0)*1e3))
