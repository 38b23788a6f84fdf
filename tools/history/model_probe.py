#!/usr/bin/env python3
"""GPU probe: what the model kernel costs per symbol on ONE long stream of a given character and codec, nothing else running
(GZ_NO_PIPELINE=1: sort, model, chain, low one after the other) - the wide-alphabet hot context of BAM's packed qualities against
the narrow one of the same qualities unpacked, and a FASTQ quality stream."""
import os
import sys
os.environ.setdefault("GZ_NO_PIPELINE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                      # noqa: E402
from genozip_amd import workload as W   # noqa: E402
from genozip_amd.codec import Engine    # noqa: E402

E = Engine(device=0, lib_path=os.environ.get("GZ_LIB_PATH"))
n_reads = 46000
qbin = W.quality_rows(W._NP, 4000, 0, n_reads, "bin").reshape(-1).astype(np.uint8).tobytes()
qdiv = W.quality_rows(W._NP, 4000, 0, n_reads, "div").reshape(-1).astype(np.uint8).tobytes()
cases = [("bin ARTb (packed 4/byte: BAM's choice)", 18, qbin), ("bin ARTB (unpacked)", 16, qbin), ("bin ARTW", 17, qbin), ("div ARTB (FASTQ)", 16, qdiv)]
for name, codec, d in cases:
    copies = 8
    bufs = [E.mem.upload(d) for _ in range(copies)]
    tab, outs = E.make_stream_table([(codec, b, len(d)) for b in bufs])
    E.compress_table(tab, copies); E.sync()
    E.profile(True, reset=True)
    for _ in range(3):
        E.compress_table(tab, copies); E.sync()
    E.profile(False)
    pr = E.profile_results()
    ms = {k: v[0] / 3 for k, v in pr.items()}
    print("%-42s %d B -> %d B   model %.2f ms (%.1f ns / input byte)  chain %.2f ms   others: %s" % (
        name, len(d), tab[0].out_len, ms.get("k_arith_model", 0), ms.get("k_arith_model", 0) * 1e6 / len(d), ms.get("k_arith_chain", 0),
        ", ".join("%s %.2f" % (k.replace("k_", ""), v) for k, v in sorted(ms.items(), key=lambda kv: -kv[1]) if k not in ("k_arith_model", "k_arith_chain"))[:120]))
