"""CPU: the oracle (oracle/gz_oracle.c) against the committed golden vectors, the survey's KATs and - where it was
built - the reference's own htscodecs (oracle/_ref)."""
import numpy as np
import pytest

import cases
from genozip_amd import synth

# SURVEY.md 8c bootstrap vectors (hex of the full codec output incl. the order byte)
K1 = b"ACGT" * 16
K2 = bytes((ord(":") if i % 10 == 9 else ord(",") if i % 7 == 3 else ord("F")) for i in range(60))
KATS = [
    (K1, "rans", 0x01, "0140a0004143475400000004000200011000010002100000000310000010000200800000008000000080000000800000"),
    (K1, "rans", 0x81, "a040044143475410e4e4e4e4e4e4e4e4e4e4e4e4e4e4e4e4"),
    (K1, "rans", 0x99, "89400404040404b0014100b0014300b0014700b0015400"),
    (K1, "arith", 0x01, "01405500c62b2fb6f969c8bc864b77fa2bcc50"),
    (K1, "arith", 0x81, "8140044143475410e500fffffd03d855d19a00"),
    (K1, "arith", 0x19, "09400409090404114200ffffaf108fdd114400ffffb6c366eab0014700b0015400"),
    (K2, "rans", 0x01, "013ca0002c3a460000000100000300010107000001000007000009073030d3c817505eff7a309b871230ec3045"),
    (K2, "rans", 0x81, "a03c032c3a460f2aaa86aa62aaa826aa4aaaa2a6a86a"),
    (K2, "arith", 0x81, "a03c032c3a460f2aaa86aa62aaa826aa4aaaa2a6a86a"),
    (K2, "arith", 0x01, "013c4700fffe6b806a74c88cb5228329f1d3ca04a6e14900"),
    (K2, "rans", 0x99, "893c04070a070ab0022c4602bf5fb0032c3a46049a682a29b0022c4602fb7db0032c3a4604a829a61a"),
]


@pytest.mark.parametrize("data,engine,order,hexout", KATS)
def test_survey_kats(oracle, data, engine, order, hexout):
    assert oracle.hts_compress(engine, data, order).hex() == hexout
    assert oracle.hts_uncompress(engine, bytes.fromhex(hexout), len(data)) == data


def test_golden_vectors(oracle):
    """every committed vector (made by tests/golden/make_golden.py from the reference's htscodecs)"""
    n = 0
    shift_margin = 1.0
    for c in cases.golden_cases():
        data = cases.golden_input(c)
        out = oracle.hts_compress(c["engine"], data, c["order"])
        cases.check_golden(c, out)
        if c["engine"] == "rans" and len(data) >= 8:
            r = oracle.last_shift_ratio()
            if r == r:
                shift_margin = min(shift_margin, abs(r - 1.01))
        if len(data) <= 100000:
            assert oracle.hts_uncompress(c["engine"], out, len(data)) == data
        n += 1
    assert n > 3000
    # compute_shift (rANS_static4x16pr.c:681) compares e10/e12 with 1.01 in floating point: the corpus stays far
    # enough from the boundary that a last-ulp difference between libm / FMA choices cannot flip a decision
    assert shift_margin > 1e-6


def test_against_reference_build(oracle, ref):
    """oracle == the reference's own code on fresh inputs (only where oracle/_ref could be built)"""
    for name, data in cases.edge_streams(70000):
        for (engine, order) in cases.CODEC_OF:
            a = oracle.hts_compress(engine, data, order)
            b = ref.hts_compress(engine, data, order)
            assert a == b, (name, engine, hex(order))
            if data:
                assert ref.hts_uncompress(engine, a, len(data)) == data
                assert oracle.hts_uncompress(engine, b, len(data)) == data


def test_codec_surface(oracle):
    data = synth.markov_bytes(9, 5000, 40, 33).tobytes()
    for codec in (1, 6, 7, 8, 9, 16, 17, 18, 19):
        est = oracle.est_size(codec, len(data))
        comp = oracle.codec_compress(codec, data)
        assert oracle.codec_uncompress(codec, comp, len(data)) == data
        if codec != 1:
            # "too small" -> caller retries; the coder's own test is against the htscodecs bound = est_size - 1 KB (codec_htscodecs.c:26-33)
            assert oracle.codec_compress(codec, data, cap=est - 1024 - 1, soft_fail=True) is None
            assert oracle.codec_compress(codec, data, cap=est - 1024) == comp
            assert oracle.codec_compress(codec, data, cap=2 * est) == comp                   # capacity independent
    # est_size values: 1 KB + htscodecs bound (codec_htscodecs.c:26-33)
    assert oracle.est_size(6, 0) == 1024 + 198948 and oracle.est_size(16, 0) == 1024 + 198931
    many = oracle.codec_compress_many([6, 16, 9], [data, data, data], 3)
    assert many == [oracle.codec_compress(c, data) for c in (6, 16, 9)]


def test_b250_kats(oracle):
    """SURVEY.md A.2 table (hand-derived from src/b250.c:29-43,82-110)"""
    kat = {0: "00", 126: "7e", -2: "7f", 127: "8000", 16508: "bffd", -3: "bffe", -4: "bfff", 16509: "c00000",
           2113660: "dfffff", 2113661: "e020407d"}
    for wi, hx in kat.items():
        assert oracle.b250_piz([wi]).hex() == hx
    # seg format: little endian, tag last; new nodes (>= ol_nodes_len) always 4 bytes
    assert oracle.b250_seg([5], 100).hex() == "05"
    assert oracle.b250_seg([127], 1000).hex() == "0080"
    assert oracle.b250_seg([5], 3).hex() == "050000e0"
    # generate: node->word, ONE_UP only when the dictionary has > 1024 words and both neighbours are >= 0
    seg = oracle.b250_seg([1, 2, 3, 10, 2000, 2001], 2000)
    out = oracle.b250_generate(seg, 2000, [500, 501])
    assert oracle.b250_decode(out) == [1, 2, 3, 10, 500, 501]
    assert out.hex() == "017f7f0a8175" + "7f"
    small = oracle.b250_generate(oracle.b250_seg([1, 2, 3], 100), 100, [])
    assert small.hex() == "010203"          # dictionary <= 1024 words: no ONE_UP
    for seed in range(5):
        ni, n2w = cases.b250_case(100 + seed, 3000, 1500, 700)
        seg = oracle.b250_seg(ni, 1500)
        out = oracle.b250_generate(seg, 1500, n2w)
        want = [(n2w[x - 1500] if x >= 1500 else x) for x in ni]
        assert oracle.b250_decode(out) == want
        assert len(out) <= len(seg)


def test_local_and_sections(oracle):
    import pyoracle as po
    import struct
    import zlib
    # interlace KATs (src/context.h:99-100): 2,-5 -> 4,9 ; int8 extremes
    lt, b = oracle.local_generate(1, struct.pack("<4b", 2, -5, -128, 127))
    assert list(b) == [4, 9, 255, 254]
    lt, b = oracle.local_generate(5, struct.pack("<2i", -1, 1))
    assert b.hex() == "00000001" + "00000002"
    lt, b = oracle.local_generate(6, struct.pack("<I", 0x11223344))
    assert b.hex() == "11223344"
    # transpose after BGEN (src/zip.c:185-219): 2 rows x 3 cols of u16
    lt, b = oracle.local_generate(4, struct.pack("<6H", 1, 2, 3, 4, 5, 6), 3)
    assert lt == 15 and b.hex() == "000100040002000500030006"
    # adler32 == zlib's
    d = synth.uniform_bytes(3, 100000).tobytes()
    assert oracle.adler32(d) == zlib.adler32(d)
    # section header layout
    desc = po.GzoCtxSectionDesc(vblock_i=3, section_type=12, codec=6, sub_codec=0, flags=0x04, ltype=11, param=7,
                                b250_size_or_nothing_char=0xff)
    desc.dict_id[:] = list(b"QUAL\0\0\0\0")
    payload = synth.markov_bytes(1, 4000, 40, 33).tobytes()
    sec = oracle.section_compress(desc, payload)
    comp = oracle.codec_compress(6, payload)
    assert sec[:4].hex() == "27052012" and sec[40:] == comp
    assert struct.unpack(">IIIII", sec[4:24]) == (zlib.adler32(comp), 0, len(comp), len(payload), 3)
    assert list(sec[24:32]) == [12, 6, 0, 0x04, 11, 7, 0xff, 0] and sec[32:40] == b"QUAL\0\0\0\0"
    tiny = oracle.section_compress(desc, b"x" * 49)          # < 50 bytes: stored raw, codec byte rewritten to NONE
    assert tiny[25] == 1 and tiny[40:] == b"x" * 49


def test_acgt_kats(oracle):
    """hand-derived from codec_acgt.c:45-55,66-70 and reference.c:45-58 (parity unpinned beyond these: the reference ships
    no fixture and cannot be run): base i sits in bits 2(i%4) of byte i/4, whole little-endian 64-bit words"""
    p, x, has_x = oracle.acgt_pack(b"ACGT")
    assert (p, x, has_x) == (bytes([0b11100100]) + bytes(7), bytes(4), False)
    p, x, has_x = oracle.acgt_pack(b"ACGTacgtNNRYACGTA")
    assert p.hex() == "e4e440e400000000" and x == b"\0\0\0\0\1\1\1\1NNRY\0\0\0\0\0" and has_x
    assert oracle.acgt_unpack(p, x, 17) == b"ACGTacgtNNRYACGTA" and oracle.acgt_unpack(p, None, 17) == b"ACGTACGTAAACACGTA"
    assert oracle.acgt_pack(b"") == (b"", b"", False) and len(oracle.acgt_pack(b"A" * 33)[0]) == 16
    iupac = {"U": 3, "R": 0, "Y": 1, "S": 1, "W": 0, "K": 2, "M": 0, "B": 1, "D": 0, "H": 0, "V": 0, "N": 0, "-": 0, "*": 0}
    for ch, code in iupac.items():
        for c in (ch, ch.lower()):
            assert oracle.acgt_pack(c.encode())[0][0] == code and oracle.acgt_pack(c.encode())[1] == c.encode()


def test_seg_column_kats(oracle):
    """hand-derived known answers for the seg-side restatements (rows a1-a3, N1): the reference holds no fixtures"""
    import numpy as np
    text = b"chr1chr2chr1chrXchr2"
    r = oracle.ctx_seg_column(text, [0, 4, 8, 12, 16, 0xffffffff, 0], [4, 4, 4, 4, 4, 0, 0], ol_snips=[b"chr2"])
    # chr2 is node 0 (cloned); chr1 and chrX are new in order of first occurrence; then MISSING (-4) and EMPTY (-3)
    assert r["node_index"].tolist() == [1, 0, 1, 2, 0, -4, -3]
    assert r["dict"] == b"chr1\0chrX\0" and r["node_char_index"].tolist() == [0, 5] and r["node_snip_len"].tolist() == [4, 4]
    assert r["counts"].tolist() == [2, 2, 1]
    # new nodes: 4 bytes 111|node little endian; the cloned node 0: one byte; MISSING bf ff / EMPTY bf fe stored tag-last
    assert r["b250"].hex() == "010000e0" "00" "010000e0" "020000e0" "00" "ffbf" "febf"
    assert r["b250_count"] == 7 and not r["all_the_same"]
    # all the same: ONE entry however often it was appended (b250.c:117-141)
    r = oracle.ctx_seg_column(text, [0, 8], [4, 4])
    assert r["b250"].hex() == "000000e0" and r["b250_count"] == 2 and r["all_the_same"] and r["counts"].tolist() == [2]
    # dyn-int: first of UINT8 INT8 UINT16 INT16 UINT32 INT32 INT64 that holds all values (GZ_LT numbering)
    assert oracle.dyn_int_column([0, 200, -1]) == (3, bytes([0, 0, 200, 0, 255, 255]))              # INT16
    assert oracle.dyn_int_column([1, 2, 254], [0, 0, 0], nothing_char=46) == (2, bytes([1, 2, 254]))  # UINT8: 254 is the top with a nothing_char
    assert oracle.dyn_int_column([1, 2, 255], [0, 0, 0], nothing_char=46)[0] == 4                    # ... 255 needs UINT16
    assert oracle.dyn_int_column([5, 255], [1, 0], 46) == (4, bytes([255, 255, 255, 0]))             # nothing first; stored as the type's maximum
    assert oracle.dyn_int_column([-129])[0] == 3 and oracle.dyn_int_column([1 << 31])[0] == 6 and oracle.dyn_int_column([-(1 << 31) - 1])[0] == 7
    assert oracle.local_blob_column(text, [0, 4], [4, 4], True) == b"chr1\0chr2\0"
    # lines (a '\r' before the newline is not part of the line; a last line without newline counts), reads, tokens
    t = b"@r1:2 x\r\nACGT\n+\nFFFF\n@r2:3 y\nAC\n+r2\nF#"
    lo, ll = oracle.text_lines(t)
    assert lo.tolist() == [0, 9, 14, 16, 21, 29, 32, 36] and ll.tolist() == [7, 4, 1, 4, 7, 2, 3, 2]
    rc, cols = oracle.fastq_records(t, lo, ll)
    assert rc == 0 and [c[0].tolist() for c in cols] == [[1, 22], [9, 29], [15, 33], [16, 36]]
    assert [c[1].tolist() for c in cols] == [[6, 6], [4, 2], [0, 2], [4, 2]]
    nb, io, il = oracle.tokenize_column(t, cols[0][0], cols[0][1], b": ")
    assert nb == 0 and io.tolist() == [[1, 22], [4, 25], [6, 27]] and il.tolist() == [[2, 2], [1, 1], [1, 1]]
    assert oracle.fastq_records(t.replace(b"\n+r2", b"\n-r2"), lo, ll)[0] == -2


def test_ctx_golden_vectors(oracle):
    """the oracle's b250_seg_append / b250_zip_generate / dyn_int_append / dyn_int_transpose == the vectors generated from
    the reference's own src/b250.c and src/dyn_int.c (tests/golden/ctx_golden.json): rows a2, a3, a5, a7 PINNED"""
    import parity
    parity.ctx_golden(None, oracle)


def test_against_ctx_reference_build(oracle):
    """where /root/reference exists: the oracle against the reference's own b250.c / dyn_int.c compiled in place, on fresh
    random cases beyond the committed vectors"""
    import os
    import numpy as np
    import pytest
    import cases
    import pyoracle
    from genozip_amd import synth
    if not pyoracle.CtxRef.available():
        if os.path.isfile("/root/reference/src/b250.c"):
            pyoracle.build(ref=True)
        else:
            pytest.skip("oracle/_ref/libctxref.so not built (reference sources absent)")
    R = pyoracle.CtxRef()
    for seed in range(1000, 1120):
        ne = [1, 3, 50, 2000, 20000][seed % 5]
        ol, nn = [(0, 9), (100, 0), (1500, 700), (17000, 300), (2200000, 10), (1000, 30)][seed % 6]
        ni, n2w = cases.b250_ctx_case(seed, ne, ol, nn, seed % 11 == 0)
        seg, cnt, ats = R.b250_seg(ni, ol)
        assert seg == oracle.b250_seg(ni[:1] if ats else ni, ol) and cnt == len(ni), seed
        assert R.b250_generate(seg, cnt, ats, ol, n2w) == oracle.b250_generate(seg, ol, n2w), seed
    r = synth.u32(77, 30000).astype(np.int64)
    for k in range(60):
        n = [1, 2, 50, 400][k % 4]
        v = (r[k * 400:k * 400 + n] % [200, 300, 70000, 1 << 33, 1 << 40][k % 5]) - [0, 0, 100, 40000, 1 << 35][(k // 5) % 5]
        isn = (r[k * 400 + 1:k * 400 + 1 + n] % 5 == 0).astype(np.uint8) if k % 3 == 0 else None
        nc = 46 if k % 3 == 0 else 0
        assert R.dyn_int_column(v, isn, nc) == oracle.dyn_int_column(v, isn, nc), k


def test_assign_golden_vectors(oracle):
    """row a8 PINNED: the oracle's sorter and its composition of the candidate table == what the reference's own src/codec.c did
    (tests/golden/assign_golden.json: codec_assign_sorter under qsort on 780 tables; codec_assign_best_codec's winner and four best rows
    with the reference's coders underneath and a scripted clock)"""
    import parity
    assert parity.assign_golden_sort(oracle.assign_sort) == 780

    def best_table(data, rows, ns, mode):
        c0, sizes = oracle.assign_best(data)
        cand = (1, 6, 7, 8, 9, 16, 17, 18, 19)
        sample = min(len(data), 99999)
        tests = [(cd, sizes[i], (sample * ns[cd] / 1000.0) if i else 0.0) for i, cd in enumerate(cand)] + [(c, sz + 28, ck) for c, sz, ck in rows]
        return oracle.assign_sort(tests, mode)
    assert parity.assign_golden_run(best_table) > 100


def test_merge_golden_vectors(oracle):
    """the LOOP of row a4 PINNED: the oracle's ctx_merge == what the reference's own ctx_merge_in_one_vctx / ctx_commit_node /
    ctx_drop_all_the_same (src/context.c compiled in place) did on the multi-VBlock scenarios of tests/golden/merge_golden.json"""
    import parity
    import pyoracle
    assert parity.merge_loop_golden(lambda est: pyoracle.OracleZctx(oracle, est), oracle.ctx_seg_column) >= 50


def test_order_golden_vectors():
    """row a15 PINNED: the tests' own statement of the section order (parity._section_order_ref, what the composition of every driver test
    uses) == the order the reference's own zip_compress_all_contexts_local / _b250 produced (tests/golden/order_golden.json)"""
    import parity
    assert parity.section_order_golden(parity._section_order_ref) == 360
