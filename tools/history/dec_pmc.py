#!/usr/bin/env python3
"""GPU: one quality stream through the arithmetic decoder (for rocprofv3 --pmc)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from genozip_amd import workload as W
from genozip_amd.codec import Engine
E = Engine(device=0)
qual = W.quality_rows(W._NP, 82, 0, 14000, "div").reshape(-1).astype(np.uint8).tobytes()
z = E.compress_many([(16, qual)])[0]
back = E.uncompress_many([(16, z, len(qual))])
assert back[0] == qual
print(len(qual))
