#!/usr/bin/env python3
"""GPU probe: k_acgt_pack / k_acgt_unpack on the SEQ of the FASTQ-PE-1M config (2 x 150 MB), against the HBM roofline
(algorithmic bytes: read n + write n/4 + write n exceptions)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from genozip_amd.codec import Engine

E = Engine(device=0)
n = 300_000_000
dev = torch.device("cuda", 0)
seq = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (n,), device=dev)]
seq[torch.randint(0, n, (n // 1000,), device=dev)] = ord("N")
pl = E.L.gz_acgt_packed_len(n)
packed = torch.empty(pl + 16, dtype=torch.uint8, device=dev)
x = torch.empty(n + 16, dtype=torch.uint8, device=dev)
out = torch.empty(n + 16, dtype=torch.uint8, device=dev)
has_x = C.c_int(0)
for name in ("pack", "unpack"):
    E.profile(True, reset=True)
    for _ in range(5):
        if name == "pack":
            E.L.gz_acgt_pack(E.h, seq.data_ptr(), n, packed.data_ptr(), x.data_ptr(), C.byref(has_x))
        else:
            E.L.gz_acgt_unpack(E.h, packed.data_ptr(), x.data_ptr(), n, out.data_ptr())
    E.profile(False)
    ms = [v[0] / v[1] for k, v in E.profile_results().items() if "acgt" in k][0]
    print("k_acgt_%s: %.3f ms for %d bases = %.0f GB/s algorithmic (%.1f %% of 8 TB/s)" % (name, ms, n, 2.25 * n / ms / 1e6, 2.25 * n / ms / 1e6 / 80))
assert torch.equal(out[:n], seq)
print("round trip ok, has_x =", has_x.value)
