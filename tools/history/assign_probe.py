"""which trial compression of codec_assign_best_codec is slow? every (context sample, codec) of the bench's first VBlock on its own,
wall-clock per call (batch of one) - tools/assign_probe.py [vb_mb]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from genozip_amd.codec import Engine
from genozip_amd import workload as W
from genozip_amd.lib import CODEC_NAMES, SIMPLE_CODECS

vb_mb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = W.reads_per_vb(vb_mb << 20)
lane, tile, x, y = W.name_fields(1, 0, n)
xv = 10000 + x.astype(np.int64) % 20000
yv = 10000 + y.astype(np.int64) % 80000
samples = {
    "Q1NAME.local u8 lane": (lane + 1).astype(np.uint8).tobytes(),
    "Q3NAME.local x delta BE": np.diff(np.concatenate([[0], xv])).astype(">i4").tobytes() if False else None,
}
def zz(d, w):                      # interlace + big endian
    d = np.where(d < 0, -2 * d - 1, 2 * d)
    return d.astype(">u%d" % w).tobytes()
dx, dy = np.diff(np.concatenate([[0], xv])), np.diff(np.concatenate([[0], yv]))
samples["Q3NAME.local x delta"] = zz(dx, 2 if np.abs(dx).max() < 32768 else 4)
samples["Q4NAME.local y delta"] = zz(dy, 2 if np.abs(dy).max() < 32768 else 4)
samples["Q2NAME.b250 tile"] = (tile % 127).astype(np.uint8).tobytes()
samples["QUAL.local"] = W.quality_rows(W._NP, 0x100001, 0, 700).tobytes()
samples = {k: v[:99999] for k, v in samples.items() if v}
E = Engine(device=0)
E.compress_many([(16, b"x" * 1000)])
for name, data in samples.items():
    row = []
    for c in SIMPLE_CODECS:
        E.sync(); t0 = time.perf_counter(); out = E.compress_many([(c, data)]); dt = (time.perf_counter() - t0) * 1e3
        row.append("%s %.1f ms (%d B)" % (CODEC_NAMES[c], dt, len(out[0])))
    print("%-24s %6d B: %s" % (name, len(data), "  ".join(row)))
