"""What ONE context's model wave costs (k_arith_model), alone on the device: order-1 streams (codec ARTB, no trial) in which every second byte is
'A' - so ONE context ('A') sees half of the stream and every other context is followed by 'A' alone (no events, nothing to compute) - with the
successor of 'A' drawn from a distribution of choice:
    constant      one symbol: no order change ever - the cost of a quiet batch of 64 occurrences (fetch, closed formulas, stores)
    zipf40        40 symbols, frequency ~ 1 / rank: a settled quality-like context
    dominant+20   95 % one symbol, 5 % spread evenly over 20 rare ones (the hot context of the VCF configuration's zero-dominated planes):
                  the rare symbols' counts stay near each other, nearly every occurrence of one overtakes its neighbour
    uniform40     40 equally likely symbols: the worst case for order changes
    uniform200    200 equally likely symbols: the models that live in LDS tables (rounds)
Reports ns per occurrence of the hot context = the k_arith_model launch (gz_profile) / occurrences; with a -DGZ_MODEL_PHASES build of the library
(GZ_LIB=...) the library itself prints events per batch and us per batch beside it.     python tools/probes/model_probe.py [n_bytes]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from genozip_amd.codec import Engine   # noqa: E402


def stream(kind, n, seed=3):
    rng = np.random.default_rng(seed)
    h = n // 2
    if kind == "constant":
        s = np.full(h, 66, dtype=np.uint8)
    elif kind == "zipf40":
        p = 1.0 / np.arange(1, 41); p /= p.sum()
        s = (66 + rng.choice(40, size=h, p=p)).astype(np.uint8)
    elif kind == "dominant+20":
        p = np.array([0.95] + [0.05 / 20] * 20)
        s = (66 + rng.choice(21, size=h, p=p)).astype(np.uint8)
    elif kind == "uniform40":
        s = (66 + rng.integers(0, 40, size=h)).astype(np.uint8)
    else:
        s = (40 + rng.integers(0, 200, size=h)).astype(np.uint8)
    out = np.empty(2 * h, dtype=np.uint8)
    out[0::2] = 65; out[1::2] = s
    return out.tobytes()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16 << 20
    E = Engine(device=0, lib_path=os.environ.get("GZ_LIB")) if os.environ.get("GZ_LIB") else Engine(device=0)
    for kind in ("constant", "zipf40", "dominant+20", "uniform40", "uniform200"):
        data = stream(kind, n)
        E.compress_many([(16, data)])                      # warm
        E.profile(True, reset=True)
        out = E.compress_many([(16, data)])[0]
        E.profile(False)
        prof = E.profile_results()
        ms, launches = prof.get("k_arith_model", (0.0, 0))
        chain = prof.get("k_arith_chain", (0.0, 0))[0]
        print("%-12s %9d occurrences of the hot context: k_arith_model %8.2f ms in %d launches = %6.2f ns per occurrence = %5.2f us per batch of 64; chain %7.2f ms = %5.2f ns per symbol; %d -> %d bytes"
              % (kind, n // 2, ms, launches, ms * 1e6 / (n // 2), ms * 1e3 / (n // 2 / 64), chain, chain * 1e6 / n, n, len(out)), flush=True)
    E.close()


if __name__ == "__main__":
    main()
