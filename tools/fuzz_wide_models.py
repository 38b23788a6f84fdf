"""Run on the GPU box: randomly drawn streams (parity.wide_models_random) through the arithmetic coders against the oracle - more and
longer ones than the test suite takes: python tools/fuzz_wide_models.py [rounds of 250 cases] [longest stream] [smallest alphabet]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import parity                                                    # noqa: E402
import pyoracle                                                  # noqa: E402
from genozip_amd.codec import Engine                             # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
max_n = int(sys.argv[2]) if len(sys.argv) > 2 else 250000
min_sym = int(sys.argv[3]) if len(sys.argv) > 3 else 65
E, O, t, tot = Engine(device=0), pyoracle.Oracle(), time.time(), 0
for rnd in range(rounds):
    tot += parity.wide_models_random(E, O, 250, seed0=20000 + 1000 * rnd + 7 * min_sym, max_n=max_n, min_sym=min_sym)
    print("round", rnd, "ok:", tot, "streams identical to the oracle's,", round(time.time() - t, 1), "s", flush=True)
