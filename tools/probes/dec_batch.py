"""kernel-level view of one batched decode (rocprofv3 --kernel-trace --stats -- python tools/probes/dec_batch.py [streams] [codec])"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
from genozip_amd import synth
from genozip_amd.codec import Engine
E = Engine(device=0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
codec = int(sys.argv[2]) if len(sys.argv) > 2 else 16
data = synth.quality_diverse(1, 1400).tobytes()
comp = E.compress_many([(codec, data)])[0]
for rep in range(2):
    t = time.perf_counter()
    backs = E.uncompress_many([(codec, comp, len(data))] * nb)
    print("%d streams of %d symbols: %.1f ms" % (nb, len(data), (time.perf_counter() - t) * 1e3), flush=True)
assert all(b == data for b in backs)
