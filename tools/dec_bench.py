"""a14 (decoders): time per symbol of the arithmetic decoder, one stream and a batch (one gpurun call):  python tools/dec_bench.py [reads]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from genozip_amd import synth
from genozip_amd.codec import Engine

E = Engine(device=0, lib_path=os.environ.get("GZ_LIB"))      # (GZ_LIB: a probe build, e.g. one made with -DGZ_DEC_PROFILE)
CODECS = [int(c) for c in os.environ.get("GZ_DEC_CODECS", "16,17,18,19,6").split(",")]
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 45000
prof = {"div": synth.quality_diverse(1, reads).tobytes()}
if hasattr(synth, "quality_binned"):
    prof["bin"] = synth.quality_binned(1, reads).tobytes()
rng = np.random.default_rng(1)
prof["bytes200"] = rng.choice(200, size=len(prof["div"]), p=rng.dirichlet(np.ones(200) * 0.2)).astype(np.uint8).tobytes()
for name, data in prof.items():
    for codec in CODECS:
        comp = E.compress_many([(codec, data)])[0]
        E.uncompress_many([(codec, comp, len(data))])
        t = time.perf_counter()
        back = E.uncompress_many([(codec, comp, len(data))])[0]
        dt = time.perf_counter() - t
        assert back == data
        line = "%-9s codec %2d flags %02x: %8d symbols in %7.1f ms = %6.1f ns/symbol (one stream)" % (name, codec, comp[0], len(data), dt * 1e3, dt * 1e9 / len(data))
        for nb, part in ((32, 8), (1024, 32)):
            small = data[:len(data) // part]
            comp_s = E.compress_many([(codec, small)])[0]
            t = time.perf_counter()
            backs = E.uncompress_many([(codec, comp_s, len(small))] * nb)
            dtb = time.perf_counter() - t
            assert all(b == small for b in backs)
            line += "; %d streams of %d: %7.1f ms = %5.2f ns/symbol overall" % (nb, len(small), dtb * 1e3, dtb * 1e9 / (nb * len(small)))
        print(line, flush=True)
