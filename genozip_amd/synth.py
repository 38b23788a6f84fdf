"""Deterministic synthetic inputs: byte streams for the codec tests and the FASTQ / VCF shaped
workloads of BASELINE.json. Everything derives from a counter-based splitmix64 so that numpy versions, platforms
and a C implementation agree bit for bit."""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n, start=0):
    """n pseudo random uint64: value i is a pure function of (seed, start+i)"""
    with np.errstate(over="ignore"):
        z = (np.arange(start, start + n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) \
            + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def u32(seed, n, start=0):
    return (splitmix64(seed, n, start) >> np.uint64(32)).astype(np.uint32)


def uniform_bytes(seed, n, nsym=256, base=0):
    return ((u32(seed, n).astype(np.uint64) * np.uint64(nsym)) >> np.uint64(32)).astype(np.uint8) + np.uint8(base)


def skewed_bytes(seed, n, nsym=4, ratio=0.5, base=0):
    """symbol k has probability ~ ratio**k (normalised)"""
    p = ratio ** np.arange(nsym, dtype=np.float64)
    cum = np.floor(np.cumsum(p / p.sum()) * 4294967296.0).astype(np.uint64)
    cum[-1] = np.uint64(1 << 32)
    return np.searchsorted(cum, u32(seed, n).astype(np.uint64), side="right").astype(np.uint8) + np.uint8(base)


def markov_bytes(seed, n, nsym=40, base=33):
    """random walk with steps in {-1,0,+1}, reflected into [0,nsym): strong order-1 structure"""
    if n == 0:
        return np.zeros(0, np.uint8)
    steps = (u32(seed, n) % np.uint32(3)).astype(np.int64) - 1
    pos = np.cumsum(steps)
    period = 2 * max(1, nsym - 1)
    pos = np.mod(pos, period) if nsym > 1 else np.zeros(n, np.int64)
    pos = np.where(pos >= nsym, period - pos, pos)
    return pos.astype(np.uint8) + np.uint8(base)


def u32be_increasing(seed, n_bytes, max_step=50):
    m = (n_bytes + 3) // 4
    v = np.cumsum(u32(seed, m) % np.uint32(max_step), dtype=np.uint64).astype(np.uint32)
    return np.frombuffer(v.astype(">u4").tobytes(), np.uint8)[:n_bytes].copy()


def run_bytes(seed, n, nsym=8, max_run=40, base=0):
    if n == 0:
        return np.zeros(0, np.uint8)
    m = n  # upper bound on number of runs
    lens = (u32(seed, m) % np.uint32(max_run)).astype(np.int64) + 1
    k = int(np.searchsorted(np.cumsum(lens), n)) + 1
    vals = ((u32(seed + 1, k).astype(np.uint64) * np.uint64(nsym)) >> np.uint64(32)).astype(np.uint8) + np.uint8(base)
    return np.repeat(vals, lens[:k])[:n]


def quality_binned(seed, n_reads, read_len=150):
    """NovaSeq-like 4 level qualities 'F' ':' ',' '#' with per-read Markov runs, >=85% 'F' (SURVEY 8d profile Q-bin)"""
    n = n_reads * read_len
    r = u32(seed, n)
    # two-state chain approximated by thresholding a smoothed field: long 'F' stretches with short dips
    dip = r < np.uint32(int(0.04 * 2**32))
    lvl = (u32(seed + 7, n) % np.uint32(100))
    q = np.full(n, ord("F"), np.uint8)
    q[dip & (lvl < 60)] = ord(":")
    q[dip & (lvl >= 60) & (lvl < 90)] = ord(",")
    q[dip & (lvl >= 90)] = ord("#")
    # extend each dip by one position to create runs
    ext = np.roll(dip, 1); ext[0] = False
    q[ext & ~dip] = ord(":")
    return q.reshape(n_reads, read_len)


def quality_diverse(seed, n_reads, read_len=150):
    """40-level Phred with position dependent decay and local correlation (SURVEY 8d profile Q-div)"""
    n = n_reads * read_len
    pos = np.tile(np.arange(read_len, dtype=np.float64), n_reads)
    mean = 38.0 - 12.0 * (pos / read_len) ** 2
    noise = (u32(seed, n).astype(np.float64) / 2**32 + u32(seed + 3, n).astype(np.float64) / 2**32
             + u32(seed + 5, n).astype(np.float64) / 2**32 - 1.5) * 6.0
    walk = np.cumsum((u32(seed + 9, n) % np.uint32(3)).astype(np.float64) - 1.0)
    walk = walk - np.repeat(walk.reshape(n_reads, read_len)[:, 0], read_len)
    q = np.clip(np.rint(mean + noise + 0.15 * walk), 2, 41).astype(np.uint8) + np.uint8(33)
    return q.reshape(n_reads, read_len)


def bases(seed, n_reads, read_len=150, n_rate=0.001):
    n = n_reads * read_len
    r = u32(seed, n)
    b = np.frombuffer(b"ACGT", np.uint8)[(r & np.uint32(3)).astype(np.intp)].copy()
    b[u32(seed + 11, n) < np.uint32(int(n_rate * 2**32))] = ord("N")
    return b.reshape(n_reads, read_len)


STREAM_KINDS = ("uniform", "skew", "markov", "u32be", "runs", "highsym")


def stream(kind, seed, n, nsym=256):
    """the corpus used by tests/golden and the parity tests"""
    if kind == "uniform":
        return uniform_bytes(seed, n, nsym)
    if kind == "skew":
        return skewed_bytes(seed, n, nsym, 0.5 if nsym <= 16 else 0.9)
    if kind == "markov":
        return markov_bytes(seed, n, nsym, 0)
    if kind == "u32be":
        return u32be_increasing(seed, n)
    if kind == "runs":
        return run_bytes(seed, n, min(nsym, 255) or 1)
    if kind == "highsym":
        return uniform_bytes(seed, n, nsym, 256 - nsym)
    raise ValueError(kind)
