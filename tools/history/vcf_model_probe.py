#!/usr/bin/env python3
"""GPU: which (leaf, context) is the long pole of the model kernel on a VCF VBlock from text (needs a -DGZ_MODEL_DEBUG build:
genozip_amd/libgenozip_amd_dbg.so; GZ_DEBUG_PIPE=1 lists the leaves)"""
import argparse, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
import bench
from genozip_amd.codec import Engine
a = argparse.Namespace(vcf_samples=10000, vcf_lines=int(sys.argv[1]) if len(sys.argv) > 1 else 3000, vcf_vbs=1)
E = Engine(device=0, lib_path=os.environ.get("GZ_LIB_PATH") or (os.path.join(ROOT, "genozip_amd", "libgenozip_amd_dbg.so") if os.environ.get("GZ_DBG_LIB") else None))
wl = bench.VcfWorkload(E, a, torch.device("cuda", 0))
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    wl.step(None)
    torch.cuda.synchronize(); print("step %.1f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
z = wl.zbuf[:wl.offs[-1]].cpu().numpy().tobytes()
for st, codec, did, ulen, pay, _ in bench.walk_sections(z):
    print(st, codec, did, ulen, len(pay), file=sys.stderr)
