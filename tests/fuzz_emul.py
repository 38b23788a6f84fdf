#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (not collected by pytest): open-ended random cross-checks of the CPU-emulated product build
(tests/emul) against the oracle.   python tests/fuzz_emul.py codecs|b250|wide|driver [seed] [seconds]
Round 1: 4 224 codec cases (8 codecs x random structure x sizes around every threshold, with round trips) and 57 120
b250 columns (both generation paths, dictionary sizes around every VARL boundary, ONE_UP runs) - no mismatch.
Round 2: `wide` - 608 streams of 5 000 - 70 000 bytes over alphabets of 66 - 256 byte values with holes, uniform / few successors /
mixed / skewed, also as the bytes of 16-bit integers, through the arithmetic coders: the eventful batches of the models' LDS way
(round 3's serial LDS batches) incl. halvings and position chunks - no mismatch. `driver` - 164 runs of the whole VBlock driver (tests/parity.py::
fastq_zip: 9 - 120 reads per mate, 1 - 2 calls, uniform / binned scores, the three DOMQ modes, small VBlocks first, speculation forced or
not) against the oracle's composition - no mismatch.
Round 4 (the wide models in LDS tables, their batches in rounds - d_model_batch_rounds; LDS-mask counts): `wide` 388 streams, `driver` 48 runs
- no mismatch; on the GPU tools/fuzz_wide_models.py: 92 500 randomly drawn streams (alphabets of 2 - 256 symbols, up to 3 MB) - no mismatch."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emul"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np           # noqa: E402
from hostmem import HostMem  # noqa: E402
from genozip_amd.codec import Engine  # noqa: E402
import pyoracle              # noqa: E402


def engine():
    if os.environ.get("GZ_FUZZ_GPU"):                         # the real library on cuda:0 instead of the CPU stand-in (GPU box; round 6: the driver fuzz at sizes that fill many blocks of the chain's loop)
        return Engine(device=0), pyoracle.Oracle()
    so = os.path.join(ROOT, "tests", "emul", "libgenozip_amd_emul.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.run(["sh", os.path.join(ROOT, "tests", "emul", "build_emul.sh")], check=True)
    return Engine(lib_path=so, mem=HostMem()), pyoracle.Oracle()

def fuzz_codecs(seed, seconds):
    E, O = engine()
    rng = np.random.RandomState(seed)
    def gen():
        n = int(rng.choice([0, 1, 7, 8, 19, 20, 21, 49, 50, 51, 100, 300, 1000, 2500, 4100]) + rng.randint(0, 3))
        kind = rng.randint(0, 7)
        if kind == 0: d = rng.randint(0, 256, n)
        elif kind == 1: d = rng.randint(0, int(rng.randint(1, 6)), n)
        elif kind == 2: d = np.repeat(rng.randint(0, 256, n // 8 + 1), rng.randint(1, 30, n // 8 + 1))[:n]
        elif kind == 3: d = np.minimum(rng.geometric(0.3, n), 255) + int(rng.randint(0, 100)) % 200
        elif kind == 4: d = np.full(n, int(rng.randint(0, 256)))
        elif kind == 5: d = (np.cumsum(rng.randint(0, 3, n)) % int(rng.randint(2, 200)))
        else: d = np.where(rng.rand(n) < 0.9, 70, rng.randint(33, 75, n))
        return np.asarray(d, dtype=np.uint8)[:n].tobytes()
    t0 = time.time(); cases = 0; bad = 0
    while time.time() - t0 < seconds:
        items = []
        for _ in range(12):
            d = gen()
            for c in (6, 7, 8, 9, 16, 17, 18, 19):
                items.append((c, d))
        got = E.compress_many(items)
        for (c, d), g in zip(items, got):
            cases += 1
            w = O.codec_compress(c, d)
            if g != w:
                bad += 1
                print("MISMATCH codec", c, "len", len(d), d[:40].hex()); pass
        back = E.uncompress_many([(c, g, len(d)) for (c, d), g in zip(items, got)])
        for (c, d), b in zip(items, back):
            if b != d:
                bad += 1; print("ROUNDTRIP MISMATCH codec", c, "len", len(d))
    print("cases", cases, "bad", bad, "in %.0f s" % (time.time() - t0))



def fuzz_b250(seed, seconds):
    E, O = engine()
    rng = np.random.RandomState(seed)
    t0 = time.time(); cases = bad = 0
    while time.time() - t0 < seconds:
        jobs = []
        for _ in range(10):
            ol = int(rng.choice([0, 1, 100, 126, 127, 128, 1000, 1023, 1024, 1025, 16508, 16509, 16510, 30000, 2113660, 2113700]))
            nn = int(rng.choice([0, 1, 5, 24, 1024, 3000]))
            if ol + nn == 0: nn = 1
            n = int(rng.choice([1, 2, 3, 63, 64, 65, 300, 5000, 40000]))
            hi = ol + nn
            mode = rng.randint(0, 4)
            if mode == 0: ni = rng.randint(0, hi, n)
            elif mode == 1: ni = np.minimum(hi - 1, np.cumsum(rng.randint(0, 2, n)) + int(rng.randint(0, max(1, hi))))     # runs of +1 (ONE_UP)
            elif mode == 2: ni = np.full(n, int(rng.randint(0, hi)))
            else: ni = np.where(rng.rand(n) < 0.5, rng.randint(0, min(hi, 127), n), rng.randint(0, hi, n))
            ni = ni.astype(np.int64)
            sp = rng.rand(n)
            ni[sp < 0.03] = -3; ni[sp > 0.98] = -4
            n2w = [int(x) for x in rng.randint(0, hi + 50, nn)]
            if mode == 1 and nn: n2w = [int(x) for x in np.minimum(hi + 40, np.cumsum(rng.randint(0, 2, nn)) + int(rng.randint(0, hi)))]
            jobs.append((O.b250_seg_array(ni.astype(np.int32), ol), ol, n2w))
        got = E.b250_generate_many(jobs)
        for (seg, ol, n2w), g in zip(jobs, got):
            cases += 1
            if g != O.b250_generate(seg, ol, n2w):
                bad += 1; print("MISMATCH", len(seg), ol, len(n2w))
    print("cases", cases, "bad", bad)


def fuzz_wide(seed, seconds):
    E, O = engine()
    rng = np.random.RandomState(seed)
    t0 = time.time(); cases = bad = 0
    while time.time() - t0 < seconds:
        items = []
        for _ in range(4):
            n = int(rng.choice([5000, 9000, 17000, 30000, 70000]))
            nsym = int(rng.choice([66, 98, 128, 129, 200, 256]))
            alpha = np.sort(rng.choice(256, nsym, replace=False))              # holes in the alphabet
            kind = rng.randint(0, 4)
            if kind == 0: idx = rng.randint(0, nsym, n)                           # uniform: every symbol an event
            elif kind == 1: idx = np.cumsum(rng.randint(0, 5, n)) % nsym         # order-1 structure, few successors
            elif kind == 2: idx = np.where(rng.rand(n) < 0.5, rng.randint(0, nsym, n), rng.randint(0, 8, n))
            else: idx = np.minimum(rng.geometric(0.02, n), nsym - 1)            # skewed, wide
            d = alpha[idx].astype(np.uint8)
            if rng.rand() < 0.3:                                                 # as the bytes of 16-bit integers
                d = np.stack([d, (idx % 7).astype(np.uint8)], axis=1).reshape(-1)[:n]
            items.append((int(rng.choice([16, 17, 18, 19])), d.tobytes()))
        got = E.compress_many(items)
        for (c, d), g in zip(items, got):
            cases += 1
            if g != O.codec_compress(c, d):
                bad += 1; print("MISMATCH codec", c, "len", len(d))
    print("cases", cases, "bad", bad)


def fuzz_driver(seed, seconds):
    import random
    import parity
    E, O = engine()
    rnd = random.Random(seed)
    t0 = time.time(); runs = bad = 0
    while time.time() - t0 < seconds:
        nr = rnd.choice([9, 17, 33, 50, 77, 120]) if not os.environ.get("GZ_FUZZ_GPU") else rnd.choice([9, 50, 120, 333, 777, 1500, 2600, 4100, 5200 + rnd.randrange(0, 900)])
        q = tuple(rnd.choice(["uniform", "bin"]) for _ in range(2))
        domq, sf = rnd.choice([0, 0, 0, 1, 13]), rnd.random() < 0.4
        if rnd.random() < 0.5: os.environ["GZ_ZIP_SPECULATION"] = "always"
        else: os.environ.pop("GZ_ZIP_SPECULATION", None)
        mono = tuple(rnd.choice([0, 0, 2, 3, 7, -1]) for _ in range(2))
        if rnd.random() < 0.3: os.environ["GZ_ZIP_PREDICTION"] = "prior"
        else: os.environ.pop("GZ_ZIP_PREDICTION", None)
        try:
            parity.fastq_zip(E, O, nr, n_calls=rnd.choice([1, 2]), qual=q, domq=domq, small_first=sf, mono=mono)
        except Exception as e:                      # noqa: BLE001
            bad += 1; print("FAIL", nr, q, domq, sf, mono, os.environ.get("GZ_ZIP_SPECULATION"), repr(e)[:300])
        runs += 1
    os.environ.pop("GZ_ZIP_SPECULATION", None); os.environ.pop("GZ_ZIP_PREDICTION", None)
    print("runs", runs, "bad", bad)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "codecs"
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 60
    {"codecs": fuzz_codecs, "b250": fuzz_b250, "wide": fuzz_wide, "driver": fuzz_driver}[which](seed, seconds)
