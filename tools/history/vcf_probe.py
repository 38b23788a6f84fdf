#!/usr/bin/env python3
"""GPU: N1 for VCF at BASELINE configs[3]'s VBlock size - 3 000 data lines x 10 000 samples (0.45 GB of text): newline index, tab index,
FORMAT subfields of every sample as columns (gz_text_lines, gz_byte_index, gz_vcf_sample_columns); kernel times from the library's
HIP events. Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np          # noqa: E402
import torch                # noqa: E402
import parity               # noqa: E402
from genozip_amd.codec import Engine   # noqa: E402

n_lines, n_samples = 3000, 10000
small = parity.vcf_text(30, n_samples, seed=5)
hdr_end = small.index(b"\nchr1") + 1
body = small[hdr_end:]
text = small[:hdr_end] + body * (n_lines // 30)
E = Engine(device=0)
lo, ll = E.text_lines(text)
data = np.array([i for i in range(len(lo)) if text[int(lo[i]):int(lo[i]) + 1] != b"#"])
E.profile(True, reset=True)
for _ in range(3):
    bad, io, il, mi = E.vcf_sample_columns(text, lo[data], ll[data], n_samples, 3)
E.profile(False)
prof = E.profile_results()
ms = {k: round(v[0] / 3, 3) for k, v in prof.items()}
tot = sum(ms.values())
print(json.dumps({"text_mb": round(len(text) / 1e6, 1), "lines": len(data), "samples": n_samples, "items": int(io.size), "n_bad": bad,
                  "kernels_ms": ms, "ms_total": round(tot, 3), "text_gb_s": round(len(text) / 1e9 / (tot / 1e3), 1)}))
