"""Synthetic FASTQ-PE workload of BASELINE.json (configs[0..1]: 150 bp paired-end, 1 M read pairs) expressed as the
per-VBlock context streams that enter the hot path.

The metric is quoted on the context streams (the per-data-type rules that turn tokens into snips stay with the caller,
SURVEY.md 8f row N1), so the workload is generated directly in the form the segmenter hands to the context engine
(fastq_text() below gives the matching FASTQ text for the seg-side kernels' probe): per VBlock
    QUAL   local  LT_BLOB   n_reads x 150 quality bytes            (fastq_seg_QUAL  src/fastq_qual.c:24)
    Q1NAME b250            lane  : tiny dictionary                  (qname tokens   src/qname.c:715-866)
    Q2NAME b250            tile  : ~600 words, slowly varying, runs
    Q3NAME local LT_UINT16 x coordinate
    Q4NAME local LT_UINT32 y coordinate, increasing within a tile
SEQ (NONREF -> CODEC_ACGT -> LZMA, src/fastq_seq.c:139-154) is outside the path and not generated.

Everything is integer arithmetic on a counter-based hash, written once against a tiny array-namespace shim so that
numpy (host; CPU baseline, tests) and torch (device; the full batch is generated in HBM) give identical bytes.
"""
import numpy as np

READ_LEN = 150
RECORD_BYTES = 63 + 1 + READ_LEN + 1 + 1 + 1 + READ_LEN + 1     # "@name\nSEQ\n+\nQUAL\n" of the synthetic Illumina read
_C1, _C2, _C3 = 0x9E3779B97F4A7C15, 0xBF58476D1CE4E5B9, 0x94D049BB133111EB


def _s64(v):
    v &= 0xFFFFFFFFFFFFFFFF
    return v - (1 << 64) if v >= (1 << 63) else v


class _NP:
    int64 = np.int64

    @staticmethod
    def arange(a, b):
        return np.arange(a, b, dtype=np.int64)

    @staticmethod
    def mul(x, c):
        with np.errstate(over="ignore"):
            return (x.astype(np.uint64) * np.uint64(c & 0xFFFFFFFFFFFFFFFF)).astype(np.int64)

    @staticmethod
    def add(x, c):
        with np.errstate(over="ignore"):
            return (x.astype(np.uint64) + np.uint64(c & 0xFFFFFFFFFFFFFFFF)).astype(np.int64)

    @staticmethod
    def lsr(x, k):
        return (x.astype(np.uint64) >> np.uint64(k)).astype(np.int64)

    cumsum = staticmethod(lambda x: np.cumsum(x, dtype=np.int64))
    clip = staticmethod(lambda x, lo, hi: np.clip(x, lo, hi))
    where = staticmethod(np.where)
    repeat_rows = staticmethod(lambda v, k: np.repeat(v, k))
    tile = staticmethod(lambda v, k: np.tile(v, k))
    to_u8 = staticmethod(lambda x: x.astype(np.uint8))


class _TH:
    def __init__(self, device):
        import torch
        self.t, self.dev, self.int64 = torch, device, torch.int64

    def arange(self, a, b):
        return self.t.arange(a, b, dtype=self.t.int64, device=self.dev)

    def mul(self, x, c):
        return x * _s64(c)                      # int64 multiplication wraps

    def add(self, x, c):
        return x + _s64(c)

    def lsr(self, x, k):
        return (x >> k) & ((1 << (64 - k)) - 1)

    def cumsum(self, x):
        return self.t.cumsum(x, 0)

    def clip(self, x, lo, hi):
        return self.t.clamp(x, lo, hi)

    def where(self, c, a, b):
        if not self.t.is_tensor(a):
            a = self.t.full_like(b if self.t.is_tensor(b) else c, a, dtype=self.t.int64)
        if not self.t.is_tensor(b):
            b = self.t.full_like(a, b)
        return self.t.where(c, a, b)

    def repeat_rows(self, v, k):
        return self.t.repeat_interleave(v, k)

    def tile(self, v, k):
        return v.repeat(k)

    def to_u8(self, x):
        return x.to(self.t.uint8)


def _hash(xp, seed, idx):
    """splitmix64 of (seed, counter) -> non-negative 32-bit values in int64 (same stream as synth.splitmix64 >> 32)"""
    z = xp.add(xp.mul(xp.add(idx, 1), _C1), seed)
    z = xp.mul(z ^ xp.lsr(z, 30), _C2)
    z = xp.mul(z ^ xp.lsr(z, 27), _C3)
    z = z ^ xp.lsr(z, 31)
    return xp.lsr(z, 32)


def quality_rows(xp, seed, read0, n_reads, profile="div"):
    """n_reads x 150 Phred+33 bytes for reads [read0, read0+n_reads) of the file with this seed.
    profile "div": 40-level, position dependent decay + per-read random walk (SURVEY 8d Q-div);
    profile "bin": NovaSeq 4-level F : , # with >= 85 % F (SURVEY 8d Q-bin)."""
    L = READ_LEN
    idx = xp.arange(read0 * L, (read0 + n_reads) * L)
    pos = idx % L
    h1, h2 = _hash(xp, seed, idx), _hash(xp, seed + 0x51ED, idx)
    if profile == "bin":
        dip = (h1 % 100) < 5
        lvl = h2 % 100
        q = xp.where(dip, xp.where(lvl < 55, 58, xp.where(lvl < 88, 44, 35)), 70)
        prev_dip = (_hash(xp, seed, idx - 1) % 100) < 5
        q = xp.where(prev_dip & ~dip & (pos > 0), 58, q)
        return xp.to_u8(q)
    steps = (h1 % 3) - 1
    walk = xp.cumsum(steps)
    row_start = xp.repeat_rows(walk[::L] - steps[::L], L)
    walk = walk - row_start
    noise = (h2 % 5) + (xp.lsr(h2, 8) % 5) + (xp.lsr(h2, 16) % 5) - 6
    q = 38 - (12 * pos * pos) // (L * L) + (walk * 3) // 8 + noise
    return xp.to_u8(xp.clip(q, 2, 41) + 33)


def name_fields(seed, read0, n_reads, xp=None):
    """lane / tile node indices and x / y coordinates of reads [read0, read0+n_reads) (numpy on the host by default, or
    the torch shim on the device). Reads are emitted tile by tile (~4000 reads per tile), x random, y increasing inside
    a tile. Both mates of a pair share the names: `seed` is the pair's."""
    host = xp is None
    xp = xp or _NP
    idx = xp.arange(read0, read0 + n_reads)
    tile_no = idx // 4000
    lane = (tile_no // 156) % 4                     # dictionary of 4 lanes
    tile = tile_no % 624                            # dictionary of 624 tiles
    h = _hash(xp, seed + 0x7A11, idx)
    x = 1000 + h % 32000
    y = 1000 + (idx % 4000) * 9 + xp.lsr(h, 16) % 9
    if host:
        x, y = x.astype(np.uint16), y.astype(np.uint32)
    return lane, tile, x, y


def reads_per_vb(vb_bytes):
    return max(1, vb_bytes // RECORD_BYTES)


def vb_ranges(n_reads, vb_bytes):
    """(read0, n_reads) of every VBlock of one mate file (txtfile_read_vblock cuts at record boundaries, src/txtfile.c:1228)"""
    per = reads_per_vb(vb_bytes)
    return [(r0, min(per, n_reads - r0)) for r0 in range(0, n_reads, per)]


_HEAD, _TAIL = b"@A00123:45:HXXXXXXXX:", b":N:0:ACGTACGT+TGCATGCA\n"


def fastq_text(seed, read0, n_reads, mate=1, profile="div", xp=None, n_rate=0):
    """the FASTQ text of reads [read0, read0 + n_reads) of mate `mate` (1 / 2) of the pair with this seed - the lines the
    segmenter's front end splits: `@A00123:45:HXXXXXXXX:<lane>:<tile>:<x>:<y> <mate>:N:0:ACGTACGT+TGCATGCA`, SEQ, `+`, QUAL.
    Fixed-width numeric fields (no leading zeros) keep every record RECORD_BYTES long, so the text is assembled as a
    matrix - with numpy on the host (returns bytes) or, xp = _TH(device), with torch in HBM (returns a uint8 tensor):
    identical bytes. The names are the pair's; SEQ and QUAL are the mate's own. n_rate: one base in n_rate is 'N'."""
    host = xp is None
    xp = xp or _NP
    lane, tile, x, y = name_fields(seed, read0, n_reads, xp)
    mseed = seed + 0x100000 * mate
    if host:
        rec = np.zeros((n_reads, RECORD_BYTES), dtype=np.uint8)
        const = lambda b: np.frombuffer(b, dtype=np.uint8)                         # noqa: E731
        to_u8 = lambda v: v.astype(np.uint8)                                       # noqa: E731
        table = lambda b, i: np.frombuffer(b, dtype=np.uint8)[i.astype(np.int64)]  # noqa: E731
    else:
        t = xp.t
        rec = t.zeros((n_reads, RECORD_BYTES), dtype=t.uint8, device=xp.dev)
        const = lambda b: t.tensor(list(b), dtype=t.uint8, device=xp.dev)          # noqa: E731
        to_u8 = lambda v: v.to(t.uint8)                                            # noqa: E731
        table = lambda b, i: t.tensor(list(b), dtype=t.uint8, device=xp.dev)[i]    # noqa: E731
    at = 0

    def put(b):
        nonlocal at
        rec[:, at:at + len(b)] = const(b)
        at += len(b)

    def digits(v, width):
        nonlocal at
        for k in range(width):
            rec[:, at + k] = to_u8(48 + (v // 10 ** (width - 1 - k)) % 10)
        at += width

    put(_HEAD); digits(lane + 1, 1); put(b":"); digits(1101 + tile, 4); put(b":"); digits(10000 + (x if not host else x.astype(np.int64)) % 20000, 5); put(b":")
    digits(10000 + (y if not host else y.astype(np.int64)) % 80000, 5); put(b" "); digits(lane * 0 + mate, 1); put(_TAIL)
    idx = xp.arange(read0 * READ_LEN, (read0 + n_reads) * READ_LEN)
    h = _hash(xp, mseed + 0x5E9, idx)
    bases = table(b"ACGT", h % 4)
    if n_rate:
        bases = xp.where((xp.lsr(h, 8) % n_rate) == 0, 78, bases.astype(np.int64) if host else bases.to(xp.int64))
        bases = to_u8(bases)
    rec[:, at:at + READ_LEN] = bases.reshape(n_reads, READ_LEN)
    at += READ_LEN
    put(b"\n+\n")
    rec[:, at:at + READ_LEN] = quality_rows(xp, mseed, read0, n_reads, profile).reshape(n_reads, READ_LEN)
    at += READ_LEN
    put(b"\n")
    assert at == RECORD_BYTES
    return rec.tobytes() if host else rec.reshape(-1)


# ---- BASELINE configs[2]: aligned reads as SAM text ---------------------------------------------------------------------
SAM_MAX_RECORD = 44 + 4 + 5 + 10 + 3 + 9 + 2 + 10 + 4 + READ_LEN + 1 + READ_LEN + 1 + 20        # widest line: every variable field at its widest


def sam_text(seed, read0, n_reads, profile="bin", xp=None):
    """alignment lines [read0, read0 + n_reads) of a coordinate-sorted SAM file of 150 bp reads on one 100 Mb contig (no header lines):
        A00123:45:HXXXXXXXX:<lane>:<tile>:<x>:<y> FLAG chr1 POS MAPQ CIGAR = PNEXT TLEN SEQ QUAL NM:i:<n> AS:i:<n>
    CIGAR 90 % 150M, 8 % with one insertion / deletion, 2 % soft-clipped; FLAG of properly paired reads; MAPQ mostly 60; QUAL binned
    (profile "bin") or 40-level ("div"). Lines differ in width (CIGAR, FLAG, MAPQ, TLEN), so a record is laid out in a matrix as wide as
    the widest line with 0 bytes where a field is shorter, and the text is the matrix without its 0 bytes - numpy on the host or, xp =
    _TH (device), torch in HBM: identical bytes. Returns (bytes | uint8 tensor)."""
    host = xp is None
    xp = xp or _NP
    lane, tile, x, y = name_fields(seed, read0, n_reads, xp)
    idx = xp.arange(read0, read0 + n_reads)
    h1, h2, h3 = _hash(xp, seed + 0x5A11, idx), _hash(xp, seed + 0x5A12, idx), _hash(xp, seed + 0x5A13, idx)
    if host:
        rec = np.zeros((n_reads, SAM_MAX_RECORD), dtype=np.uint8)
        const = lambda b: np.frombuffer(b, dtype=np.uint8)                         # noqa: E731
        to_u8 = lambda v: v.astype(np.uint8)                                       # noqa: E731
        i64 = lambda v: v.astype(np.int64)                                         # noqa: E731
    else:
        t = xp.t
        rec = t.zeros((n_reads, SAM_MAX_RECORD), dtype=t.uint8, device=xp.dev)
        const = lambda b: t.tensor(list(b), dtype=t.uint8, device=xp.dev)          # noqa: E731
        to_u8 = lambda v: v.to(t.uint8)                                            # noqa: E731
        i64 = lambda v: v.to(t.int64)                                              # noqa: E731
    at = 0

    def put(b):
        nonlocal at
        rec[:, at:at + len(b)] = const(b)
        at += len(b)

    def digits(v, width, fixed=True):
        """decimal, right-aligned in `width` columns; not fixed: leading zeros become 0 bytes (dropped from the text)"""
        nonlocal at
        v = i64(v)
        for k in range(width):
            p = 10 ** (width - 1 - k)
            d = 48 + (v // p) % 10
            if not fixed and k < width - 1:
                d = xp.where(v >= p, d, 0)
            rec[:, at + k] = to_u8(d)
        at += width

    def choice(table, sel):
        """one of the byte strings of `table` per row (sel: int64 index), left-aligned in the widest one's columns"""
        nonlocal at
        w = max(len(b) for b in table)
        m = np.zeros((len(table), w), dtype=np.uint8)
        for i, b in enumerate(table):
            m[i, :len(b)] = np.frombuffer(b, dtype=np.uint8)
        rec[:, at:at + w] = (m if host else xp.t.tensor(m, device=xp.dev))[sel]
        at += w

    put(b"A00123:45:HXXXXXXXX:"); digits(lane + 1, 1); put(b":"); digits(1101 + tile, 4); put(b":"); digits(10000 + i64(x) % 20000, 5); put(b":")
    digits(10000 + i64(y) % 80000, 5); put(b"\t")
    choice([b"99", b"147", b"83", b"163"], h1 % 4); put(b"\tchr1\t")
    pos = 100000 + idx * 97 + h2 % 60                                                  # increasing: coordinate-sorted
    digits(pos, 9, fixed=False); put(b"\t")
    choice([b"60", b"60", b"60", b"0", b"23", b"60", b"40", b"60"], xp.lsr(h1, 8) % 8); put(b"\t")
    k = xp.lsr(h1, 16) % 100
    cig = xp.where(k < 90, 0, xp.where(k < 98, 1 + k % 4, 5))
    choice([b"150M", b"70M2D80M", b"40M1I109M", b"100M3D50M", b"75M2I73M", b"20S130M"], cig); put(b"\t=\t")
    digits(pos + 150 + xp.lsr(h2, 8) % 100, 9, fixed=False); put(b"\t")
    digits(300 + xp.lsr(h2, 8) % 100, 3); put(b"\t")
    bidx = xp.arange(read0 * READ_LEN, (read0 + n_reads) * READ_LEN)
    hb = _hash(xp, seed + 0x5E9, bidx)
    tab = np.frombuffer(b"ACGT", dtype=np.uint8)
    bases = (tab[(hb % 4).astype(np.int64)] if host else xp.t.tensor(list(b"ACGT"), dtype=xp.t.uint8, device=xp.dev)[hb % 4])
    rec[:, at:at + READ_LEN] = bases.reshape(n_reads, READ_LEN); at += READ_LEN
    put(b"\t")
    rec[:, at:at + READ_LEN] = quality_rows(xp, seed + 0x100000, read0, n_reads, profile).reshape(n_reads, READ_LEN); at += READ_LEN
    put(b"\tNM:i:"); digits(h3 % 4, 1); put(b"\tAS:i:"); digits(150 - xp.lsr(h3, 8) % 9, 3); put(b"\n")
    assert at <= SAM_MAX_RECORD
    flat = rec.reshape(-1)
    out = flat[flat != 0]
    return out.tobytes() if host else out


SAM_REF_NAMES = [b"chr1"]
BAM_MAX_RECORD = 4 + 32 + 39 + 12 + READ_LEN // 2 + READ_LEN + 8


def bam_records(seed, read0, n_reads, profile="bin", xp=None):
    """the same alignments as sam_text (seed, read0, n_reads, profile) as the RECORDS of an uncompressed BAM stream (SAMv1 4.2; reference
    names SAM_REF_NAMES): block_size, the fixed fields, read_name, 1-3 CIGAR operations, 4-bit bases, Phred scores, NM / AS as uint8
    (samtools' smallest type). Records differ in length only by their CIGAR, so a record is laid out in a matrix as wide as the longest
    with a mask over the CIGAR columns a record does not have; the stream is the matrix under its mask. numpy on the host or, xp = _TH
    (device), torch in HBM: identical bytes. gz_bam_to_sam of these records is sam_text's text (tests)."""
    host = xp is None
    xp = xp or _NP
    lane, tile, x, y = name_fields(seed, read0, n_reads, xp)
    idx = xp.arange(read0, read0 + n_reads)
    h1, h2, h3 = _hash(xp, seed + 0x5A11, idx), _hash(xp, seed + 0x5A12, idx), _hash(xp, seed + 0x5A13, idx)
    if host:
        rec = np.zeros((n_reads, BAM_MAX_RECORD), dtype=np.uint8)
        keep = np.ones((n_reads, BAM_MAX_RECORD), dtype=bool)
        const = lambda b: np.frombuffer(b, dtype=np.uint8)                         # noqa: E731
        to_u8 = lambda v: v.astype(np.uint8)                                       # noqa: E731
        i64 = lambda v: v.astype(np.int64)                                         # noqa: E731
        table = lambda v: np.array(v, dtype=np.int64)                              # noqa: E731
    else:
        t = xp.t
        rec = t.zeros((n_reads, BAM_MAX_RECORD), dtype=t.uint8, device=xp.dev)
        keep = t.ones((n_reads, BAM_MAX_RECORD), dtype=t.bool, device=xp.dev)
        const = lambda b: t.tensor(list(b), dtype=t.uint8, device=xp.dev)          # noqa: E731
        to_u8 = lambda v: v.to(t.uint8)                                            # noqa: E731
        i64 = lambda v: v.to(t.int64)                                              # noqa: E731
        table = lambda v: t.tensor(v, dtype=t.int64, device=xp.dev)                # noqa: E731
    at = 0

    def put(b):
        nonlocal at
        rec[:, at:at + len(b)] = const(b)
        at += len(b)

    def le(v, nbytes):
        nonlocal at
        v = i64(v)
        for k in range(nbytes):
            rec[:, at + k] = to_u8(xp.lsr(v, 8 * k) % 256)
        at += nbytes

    def digits(v, width):
        nonlocal at
        v = i64(v)
        for k in range(width):
            rec[:, at + k] = to_u8(48 + (v // 10 ** (width - 1 - k)) % 10)
        at += width

    k100 = xp.lsr(h1, 16) % 100
    cig = xp.where(k100 < 90, 0, xp.where(k100 < 98, 1 + k100 % 4, 5))
    M, I, D, S = 0, 1, 2, 4
    ops = [[150 << 4 | M, 0, 0], [70 << 4 | M, 2 << 4 | D, 80 << 4 | M], [40 << 4 | M, 1 << 4 | I, 109 << 4 | M], [100 << 4 | M, 3 << 4 | D, 50 << 4 | M],
           [75 << 4 | M, 2 << 4 | I, 73 << 4 | M], [20 << 4 | S, 130 << 4 | M, 0]]
    n_ops = table([1, 3, 3, 3, 3, 2])[cig]
    ref_len = table([150, 152, 149, 153, 148, 130])[cig]
    pos = 100000 + idx * 97 + h2 % 60                                                  # (1-based, as in sam_text)
    beg, end = pos - 1, pos - 1 + ref_len - 1                                          # reg2bin (SAMv1 5.3) on [beg, end]
    bin_ = xp.where(xp.lsr(beg, 14) == xp.lsr(end, 14), 4681 + xp.lsr(beg, 14), xp.where(xp.lsr(beg, 17) == xp.lsr(end, 17), 585 + xp.lsr(beg, 17),
           xp.where(xp.lsr(beg, 20) == xp.lsr(end, 20), 73 + xp.lsr(beg, 20), xp.where(xp.lsr(beg, 23) == xp.lsr(end, 23), 9 + xp.lsr(beg, 23),
           xp.where(xp.lsr(beg, 26) == xp.lsr(end, 26), 1 + xp.lsr(beg, 26), 0)))))
    le(32 + 39 + 4 * n_ops + READ_LEN // 2 + READ_LEN + 8, 4)                          # block_size
    le(idx * 0, 4); le(pos - 1, 4)                                                     # refID, pos
    le(idx * 0 + 39, 1)                                                                # l_read_name
    le(table([60, 60, 60, 0, 23, 60, 40, 60])[xp.lsr(h1, 8) % 8], 1)                   # mapq
    le(bin_, 2); le(n_ops, 2)
    le(table([99, 147, 83, 163])[h1 % 4], 2)                                           # flag
    le(idx * 0 + READ_LEN, 4); le(idx * 0, 4)                                          # l_seq, next_refID
    le(pos + 150 + xp.lsr(h2, 8) % 100 - 1, 4)                                         # next_pos
    le(300 + xp.lsr(h2, 8) % 100, 4)                                                   # tlen
    put(b"A00123:45:HXXXXXXXX:"); digits(lane + 1, 1); put(b":"); digits(1101 + tile, 4); put(b":"); digits(10000 + i64(x) % 20000, 5); put(b":")
    digits(10000 + i64(y) % 80000, 5); put(b"\0")
    opt = table(ops)[cig]                                                              # [n, 3]
    for j in range(3):
        if j:
            keep[:, at:at + 4] = (n_ops > j).reshape(-1, 1)
        le(opt[:, j], 4)
    bidx = xp.arange(read0 * READ_LEN, (read0 + n_reads) * READ_LEN)
    code = table([1, 2, 4, 8])[_hash(xp, seed + 0x5E9, bidx) % 4].reshape(n_reads, READ_LEN)
    rec[:, at:at + READ_LEN // 2] = to_u8(code[:, 0::2] * 16 + code[:, 1::2]); at += READ_LEN // 2
    rec[:, at:at + READ_LEN] = (quality_rows(xp, seed + 0x100000, read0, n_reads, profile) - 33).reshape(n_reads, READ_LEN); at += READ_LEN
    put(b"NMC"); le(h3 % 4, 1); put(b"ASC"); le(150 - xp.lsr(h3, 8) % 9, 1)
    assert at == BAM_MAX_RECORD and READ_LEN % 2 == 0
    out = rec.reshape(-1)[keep.reshape(-1)]
    return out.tobytes() if host else out


# ---- BASELINE configs[3]: a multi-sample VCF as text ---------------------------------------------------------------------
def vcf_text(seed, line0, n_lines, n_samples, xp=None):
    """data lines [line0, line0 + n_lines) of a VCF with n_samples samples, FORMAT GT:DP:PL (no header lines):
        chr1 POS ID REF ALT QUAL PASS DP=<n>;AF=0.<nn> GT:DP:PL  <GT>:<DP>:<PL0>,<PL1>,<PL2> x n_samples
    GT 0/0, 0/1, 1/1 with a per-site allele frequency (most sites rare), DP ~ Poisson (30) (18-42, peaked), PL consistent with GT (0 at the called genotype). Cells
    differ in width (the PL numbers), so a line is laid out in a matrix as wide as the widest with 0 bytes where a number is shorter and the
    text is the matrix without its 0 bytes - numpy on the host or torch in HBM (xp = _TH (device)): identical bytes."""
    host = xp is None
    xp = xp or _NP
    li = xp.arange(line0, line0 + n_lines)
    hl, hl2 = _hash(xp, seed + 0x7C1, li), _hash(xp, seed + 0x7C2, li)
    FIX = 96                                                                           # columns of the fixed fields at their widest (asserted below)
    CELL = 1 + 3 + 1 + 2 + 1 + 3 + 1 + 3 + 1 + 3                                       # tab GT : DP : PL0 , PL1 , PL2
    W_ = FIX + n_samples * CELL
    if host:
        rec = np.zeros((n_lines, W_), dtype=np.uint8)
        const = lambda b: np.frombuffer(b, dtype=np.uint8)                         # noqa: E731
        to_u8 = lambda v: v.astype(np.uint8)                                       # noqa: E731
        tab = lambda m: m                                                          # noqa: E731
    else:
        t = xp.t
        rec = t.zeros((n_lines, W_), dtype=t.uint8, device=xp.dev)
        const = lambda b: t.tensor(list(b), dtype=t.uint8, device=xp.dev)          # noqa: E731
        to_u8 = lambda v: v.to(t.uint8)                                            # noqa: E731
        tab = lambda m: t.tensor(m, device=xp.dev)                                 # noqa: E731
    at = 0

    def put(b):
        nonlocal at
        rec[:, at:at + len(b)] = const(b)
        at += len(b)

    def digits(dst, col, v, width, fixed):
        for k in range(width):
            p = 10 ** (width - 1 - k)
            d = 48 + (v // p) % 10
            if not fixed and k < width - 1:
                d = xp.where(v >= p, d, 0)
            dst[..., col + k] = to_u8(d)

    put(b"chr1\t")
    digits(rec, at, 1000000 + li * 211 + hl % 200, 9, False); at += 9
    put(b"\t")
    has_id = (hl2 % 3) == 0                                                           # a third of the sites carry an rs id, the others '.'
    rs = 1000000 + xp.lsr(hl2, 4) % 9000000
    rec[:, at] = to_u8(xp.where(has_id, 114, 46)); rec[:, at + 1] = to_u8(xp.where(has_id, 115, 0))
    for k in range(7):
        rec[:, at + 2 + k] = to_u8(xp.where(has_id, 48 + (rs // 10 ** (6 - k)) % 10, 0))
    at += 9
    put(b"\t")
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    rec[:, at] = (acgt[(hl % 4).astype(np.int64)] if host else tab(acgt)[hl % 4]); at += 1
    put(b"\t")
    rec[:, at] = (acgt[((hl + 1 + xp.lsr(hl, 8) % 3) % 4).astype(np.int64)] if host else tab(acgt)[(hl + 1 + xp.lsr(hl, 8) % 3) % 4]); at += 1
    put(b"\t"); digits(rec, at, 30 + xp.lsr(hl, 12) % 60, 2, True); at += 2
    put(b"\tPASS\tDP="); digits(rec, at, 20 * n_samples + hl2 % (10 * n_samples), 7, False); at += 7
    put(b";AF=0."); digits(rec, at, xp.lsr(hl2, 8) % 100, 2, True); at += 2
    put(b"\tGT:DP:PL")
    assert at <= FIX
    # the samples: an n_lines x n_samples x CELL block
    cidx = (li * n_samples).reshape(-1, 1) + xp.arange(0, n_samples).reshape(1, -1)
    hc = _hash(xp, seed + 0x7C3, cidx.reshape(-1)).reshape(n_lines, n_samples)
    af = (xp.lsr(hl2, 16) % 100).reshape(-1, 1)                                      # per-site alt allele frequency in percent-ish: most sites rare
    af = xp.where(af < 70, af % 5, af % 60)
    u = hc % 100
    g = xp.where(u < af, 1, 0) + xp.where(xp.lsr(hc, 8) % 100 < af, 1, 0)              # 0, 1, 2 alt alleles
    dp = 18 + xp.lsr(hc, 16) % 13 + xp.lsr(hc, 24) % 13                               # ~ Poisson (30): peaked around 30, 18 .. 42
    cell = (np.zeros((n_lines, n_samples, CELL), dtype=np.uint8) if host else xp.t.zeros((n_lines, n_samples, CELL), dtype=xp.t.uint8, device=xp.dev))
    cell[..., 0] = 9                                                                   # the tab in front of the sample
    cell[..., 1] = to_u8(48 + xp.where(g == 2, 1, 0)); cell[..., 2] = 47; cell[..., 3] = to_u8(48 + xp.where(g >= 1, 1, 0)); cell[..., 4] = 58
    digits(cell, 5, dp, 2, True); cell[..., 7] = 58
    pl0 = xp.where(g == 0, 0, xp.where(g == 1, 3 * dp, 9 * dp)); pl1 = xp.where(g == 1, 0, 3 * dp); pl2 = xp.where(g == 2, 0, xp.where(g == 1, 3 * dp, 9 * dp))
    cap = lambda v: xp.where(v > 255, 255, v)                                          # noqa: E731
    digits(cell, 8, cap(pl0), 3, False); cell[..., 11] = 44
    digits(cell, 12, cap(pl1), 3, False); cell[..., 15] = 44
    digits(cell, 16, cap(pl2), 3, False)
    rec[:, FIX:] = cell.reshape(n_lines, n_samples * CELL)
    # the end of line: one more column behind the last cell
    nlcol = (np.full((n_lines, 1), 10, dtype=np.uint8) if host else xp.t.full((n_lines, 1), 10, dtype=xp.t.uint8, device=xp.dev))
    rec = (np.concatenate([rec, nlcol], axis=1) if host else xp.t.cat([rec, nlcol], dim=1))
    flat = rec.reshape(-1)
    out = flat[flat != 0]
    return out.tobytes() if host else out
