#!/usr/bin/env python3
"""GPU probe: one 10 MB order-0 and order-1 rANS stream (for rocprofv3 --pmc runs on k_rans_encode)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from genozip_amd import synth
from genozip_amd.codec import Engine
E = Engine(device=0)
h = synth.u32(77, 10_000_000)
d = (18 + (h % np.uint32(13)) + ((h >> np.uint32(8)) % np.uint32(13))).astype(np.uint8).tobytes()
for c in (6, 8):
    E.compress_many([(c, d)])
