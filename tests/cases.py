"""shared parity cases: the same checks run against the CPU-emulated build (small sizes) and the GPU (full sizes)"""
import hashlib
import json
import os

import numpy as np

from genozip_amd import synth
from genozip_amd.lib import SIMPLE_CODECS, CODEC_NONE

HERE = os.path.dirname(os.path.abspath(__file__))
CODEC_OF = {("rans", 0x01): 6, ("rans", 0x19): 7, ("rans", 0x81): 8, ("rans", 0x99): 9,
            ("arith", 0x01): 16, ("arith", 0x19): 17, ("arith", 0x81): 18, ("arith", 0x99): 19}


def golden_cases():
    with open(os.path.join(HERE, "golden", "hts_golden.json")) as f:
        return json.load(f)["cases"]


def golden_input(c):
    if c["kind"] == "markov40_q":
        data = synth.markov_bytes(c["seed"], c["n"], 40, 33).tobytes()
    else:
        data = synth.stream(c["kind"], c["seed"], c["n"], c["nsym"]).tobytes()
    assert hashlib.sha1(data).hexdigest() == c["in_sha1"], "synthetic generator drifted from the committed fixture"
    return data


def check_golden(c, out):
    assert len(out) == c["out_len"], (c["kind"], c["n"], c["nsym"], c["engine"], hex(c["order"]), len(out), c["out_len"])
    if "out_hex" in c:
        assert out.hex() == c["out_hex"], (c["kind"], c["n"], c["nsym"], c["engine"], hex(c["order"]))
    assert hashlib.sha1(out).hexdigest() == c["out_sha1"], (c["kind"], c["n"], c["nsym"], c["engine"], hex(c["order"]))


def edge_streams(max_n):
    """(name, bytes): the edge cases the codecs branch on -- empty, < 8 (order-1 dropped), <= 20 (stripe dropped),
    < 50 (stored raw by the section writer), 1 / 2 / 4 / 16 / 17 / 256 symbol alphabets (PACK widths and its
    refusal, the 256-wraps-to-0 quirk), tables over 1000 bytes (nested order-0 coding), incompressible (CAT)"""
    out = []
    seed = 5000
    for n in (0, 1, 2, 7, 8, 9, 19, 20, 21, 22, 23, 24, 49, 50, 51, 255, 256, 257, 1000, 1031, 4097, 6000, 20011, 65537, 300007):
        if n > max_n:
            continue
        for kind, nsym in (("uniform", 1), ("uniform", 2), ("uniform", 4), ("skew", 5), ("uniform", 16), ("uniform", 17),
                           ("markov", 40), ("uniform", 256), ("u32be", 256), ("runs", 8), ("highsym", 3), ("skew", 256)):
            seed += 1
            out.append(("%s%d_n%d" % (kind, nsym, n), synth.stream(kind, seed, n, nsym).tobytes()))
    # all 256 byte values present exactly once / in order: PACK's count byte wraps to 0
    if max_n >= 256:
        out.append(("perm256", bytes(range(256))))
        out.append(("perm256x3", bytes(range(256)) * 3))
    return out


def b250_case(seed, n_entries, ol_nodes, n_new, specials=True):
    """random seg-format b250: node indices (old: VARL 1-4 bytes, new: always 4 bytes), some EMPTY/MISSING, plus a
    node->word map for the new nodes. Consecutive runs are planted so that ONE_UP triggers."""
    r = synth.u32(seed, n_entries * 3)
    total = ol_nodes + n_new
    ni = (r[:n_entries] % np.uint32(max(1, total))).astype(np.int64)
    # plant increasing runs
    runs = (r[n_entries:2 * n_entries] % np.uint32(7)) == 0
    for i in range(1, n_entries):
        if runs[i]:
            ni[i] = min(total - 1, ni[i - 1] + 1)
    if specials:
        sp = r[2 * n_entries:] % np.uint32(50)
        ni[sp == 0] = -3
        ni[sp == 1] = -4
    n2w = (synth.u32(seed + 1, max(1, n_new)) % np.uint32(max(1, total + 50))).astype(np.int64)
    # make some new nodes map to "previous word + 1" so that converted neighbours trigger ONE_UP
    return [int(x) for x in ni], [int(x) for x in n2w[:n_new]]


# ---- cases of tests/golden/ctx_golden.json (generated from the reference's own b250.c / dyn_int.c) ------------------------
def b250_ctx_case(seed, n_entries, ol_nodes, n_new, all_the_same):
    """node indices of a column (old: < ol_nodes, new: ol_nodes .. ol_nodes + n_new - 1, some EMPTY / MISSING, planted runs) and
    the word index every new node gets in the merge"""
    ni, n2w = b250_case(seed, n_entries, ol_nodes, n_new, True)
    if all_the_same:
        ni = [ni[0]] * n_entries
    n2w = [int(x) % (ol_nodes + n_new + 40) for x in n2w]
    return ni, n2w


def dyn_int_cases():
    r = synth.u32(5, 4000).astype(np.int64)
    return [(r[:1000] % 200, None, 0), (r[:1000] % 256, None, 0), (r[:1000] % 256, None, 46), (r[:1000] % 300 - 20, None, 0), (r[:1000] % 100 - 50, None, 0),
            (r[:1000] % 70000, None, 0), (r[:1000] % 70000 - 5, None, 0), ((r[:777] << 3), None, 0), ((r[:500] << 3) - (1 << 34), None, 0),
            (r[:100] * 123456789 - (1 << 50), None, 0), (r[:1000] % 255, (r[:1000] % 7 == 0).astype(np.uint8), 46),
            (np.concatenate([[5], r[1:300] % 100 - 1]), np.concatenate([[1], np.zeros(299)]).astype(np.uint8), 46),
            (np.concatenate([[5], r[1:300] % 100]), np.concatenate([[1], np.zeros(299)]).astype(np.uint8), 46),
            (np.zeros(50, dtype=np.int64), np.ones(50, dtype=np.uint8), 46), (np.array([65535]), None, 46), (np.array([127, -128]), None, 0),
            (np.array([0, 4294967295]), None, 0), (np.array([0, 4294967295]), None, 46), (np.array([-1, 2147483647]), None, 0)]


TRANSPOSE_CASES = [(2, 1, 37, 50), (4, 2, 64, 255), (6, 4, 31, 300), (2, 1, 300, 1000), (2, 1, 1, 7), (4, 2, 9, 1)]


def hash_cases():
    r = synth.u32(91, 64)
    snips = [b"", b"A", b"1101", b"A00123:45:HXXXXXXXX", b"\x01", b"x" * 300] + [synth.uniform_bytes(int(k), 1 + int(k) % 40, 256).tobytes() for k in r[:40]]
    return [(hl, sn) for hl in (65521, 92681, 8388593) for sn in snips]


def domq_cases():
    """(name, QUAL lines of a VBlock)"""
    def lines_bin(n, seed):
        return [synth.quality_binned(seed + i, 1, 150 - (i % 3 == 0) * (i % 11))[0].tobytes() for i in range(n)]

    def lines_div(n, seed):
        return [synth.quality_diverse(seed + i, 1, 150)[0].tobytes() for i in range(n)]
    two = [l.replace(b"F", b"\x01").replace(b":", b"F").replace(b"\x01", b":") if i % 3 == 0 else l for i, l in enumerate(lines_bin(200, 11))]
    return [("bin", lines_bin(300, 1)), ("div", lines_div(50, 2)), ("mix", [l if i % 5 else lines_div(1, 900 + i)[0] for i, l in enumerate(lines_bin(300, 3))]),
            ("allF", [b"F" * 150] * 7), ("one", [b"F" * 150]), ("gaps", [b"" if i % 4 == 1 else l for i, l in enumerate(lines_bin(61, 5))]),
            ("startnz", [b"#" + l[1:] for l in lines_bin(40, 6)]), ("long", [b"F" * 600, b"F" * 600 + b":", b"F" * 100]), ("twodoms", two),
            ("tail", lines_bin(9, 21) + [b"F" * 150] * 3), ("ties", [b"FF::,,##" * 10, b"F" * 70 + b":,#;" * 2, b"F" * 75 + b";#,:"])]


def acgt_cases():
    """(name, NONREF.local bytes)"""
    iupac = (synth.uniform_bytes(7, 2000, 256) % 30 + 65).astype("uint8").tobytes() + b"acgtnryswkmbdhvu" * 9
    out = [("clean%d" % n, synth.bases(60 + n, 1, n, 0.0)[0].tobytes()) for n in (1, 3, 4, 5, 31, 32, 33, 63, 64, 65, 199, 200, 201, 4096, 100003)]
    out += [("dirty%d" % n, synth.bases(80 + n, 1, n, 0.01)[0].tobytes()) for n in (50, 257, 100001)]
    out += [("iupac", iupac), ("lower", synth.bases(3, 1, 1000, 0.0)[0].tobytes().lower()), ("allN", b"N" * 77), ("bytes", bytes(range(256)) * 3)]
    return out


def int_snip_cases():
    """snips of a numeric column: what str_get_int takes for an integer and what it leaves as text"""
    r = synth.u32(55, 600).astype(np.int64)
    fixed = [b"0", b"1", b"-1", b"-", b"-0", b"00", b"030", b"-030", b"+5", b"5 ", b" 5", b"12a", b"9223372036854775807", b"9223372036854775808", b"-9223372036854775808",
             b"-9223372036854775807", b"99999999999999999999", b"18446744073709551616", b"1e5", b"0x10", b".", b"1.0", b"2147483648", b"-2147483649", b"4294967295", b"4294967296", b"65535", b"65536", b"255", b"256", b"-128", b"-129"]
    return fixed + [b"%d" % (r[i] * (1 if i % 3 else -1) * (10 ** (i % 9))) for i in range(300)] + [b"1%02d" % (r[i] % 100) for i in range(50)]


def seg_node_cases():
    """(cloned words, snips of the column): none cloned / thousands cloned, all new / all known / mixed, long repeats, similar strings"""
    r = synth.u32(901, 40000).astype(np.int64)
    ol_a = [b"w%d" % i for i in range(17000)]
    return [([], [b"x%d" % (r[i] % 50) for i in range(3000)]),
            (ol_a, [b"w%d" % (r[i] % 20000) for i in range(8000)]),
            (ol_a[:300], [b"w%d" % (r[i] % 300) for i in range(2000)]),
            ([b"1101", b"1102"], [b"%d" % (1101 + i * 9 // 5000) for i in range(5000)]),
            ([b"a", b"aa", b"aaa"], [b"a" * (1 + r[i] % 9) for i in range(1000)] + [b"A00123:45:HXXXXXXXX"] * 20 + [b"\x01", b"\x05$", b"a\tb"]),
            ([], [b"id%07d" % i for i in range(6000)])]


def merge_hash_cases():
    """(name, estimated_entries, [(can_have_singletons, [(snip, count)] = the new nodes of one VBlock context)]): VBlocks that all cloned
    an empty dictionary (a batch) merging one after the other - words met again, new words, singletons, singletons met again (failed),
    the same snip a singleton twice, contexts that cannot have singletons, long chains in a small hash"""
    r = synth.u32(777, 4000).astype(np.int64)
    out = []
    ids = lambda lo, hi, rep: [(b"read%06d" % i, 1 + (r[i] % rep == 0)) for i in range(lo, hi)]   # noqa: E731
    out.append(("ids", 0, [(True, ids(0, 300, 7)), (True, ids(200, 500, 5)), (True, ids(0, 500, 11)), (False, ids(450, 600, 3)), (True, ids(0, 100, 1))]))
    out.append(("tiles", 0, [(False, [(b"%d" % (1101 + i), 40) for i in range(30)]), (False, [(b"%d" % (1110 + i), 3) for i in range(40)]), (True, [(b"%d" % (1100 + i), 1) for i in range(60)])]))
    out.append(("small_hash", 5, [(True, [(b"w%d" % (r[i] % 900), 1 + (r[i] % 3 == 0)) for i in range(v * 150, v * 150 + 400)]) for v in range(6)]))
    out.append(("twice", 100, [(True, [(b"a", 1), (b"b", 2)]), (True, [(b"a", 1)]), (True, [(b"a", 1), (b"c", 1)]), (True, [(b"c", 1), (b"c2", 1)]), (True, [(b"\x01", 1), (b"", 1)])]))
    out.append(("big", 60000, [(True, [(b"k%d_%d" % (r[(i * 7 + v) % 4000] % 5000, i % 3), 1 + (i % 4 == 0)) for i in range(1500)]) for v in range(4)]))
    return out


def section_cases():
    """(GzoCtxSectionDesc fields, payload bytes): context sections as zfile_compress_local_data / _b250_data hand them to comp_compress"""
    out = []
    k = 0
    for codec in (1, 6, 7, 8, 9, 16, 17, 18, 19):
        for n in (0, 1, 49, 50, 51, 700, 5000):
            k += 1
            is_local = k % 3 != 0
            data = synth.stream(("markov", "skew", "uniform", "runs")[k % 4], 8000 + k, n, (40, 5, 200, 8)[k % 4]).tobytes()
            out.append((dict(vblock_i=1 + k % 5, section_type=12 if is_local else 11, codec=codec, sub_codec=0, flags=(0, 0x20, 0x04, 0x01)[k % 4],
                             ltype=(11, 2, 6, 0)[k % 4] if is_local else 0, param=(0, 0, 7)[k % 3], b250_size_or_nothing_char=0xff if is_local and k % 4 in (1, 2) else 0 if is_local else 4,
                             dict_id=(b"Q%dNAME" % (k % 7) + b"\0" * 8)[:8]), data))
    return out


LOCAL_ORDER_CASES = [(1, 1), (2, 1), (3, 2), (4, 2), (5, 4), (6, 4), (7, 8), (8, 8), (9, 4), (10, 8)]


def ctx_golden():
    with open(os.path.join(HERE, "golden", "ctx_golden.json")) as f:
        return json.load(f)


def check_enc(got, want, what):
    if "hex" in want:
        assert got.hex() == want["hex"], what
    else:
        assert len(got) == want["len"] and hashlib.sha1(got).hexdigest() == want["sha1"], what


# ---- row a8: codec_assign_best_codec, pinned to the reference's own src/codec.c (tests/golden/make_assign_golden.py) ----------------------
ASSIGN_CAND = [1, 6, 7, 8, 9, 16, 17, 18, 19, 3, 5, 4]      # the trial order of src/codec.c:286-290: NONE, RAN x 4, ART x 4, BZ2, BSC, LZMA


def assign_sort_tables(rounds=260, seed=5):
    """tables of (codec, size, clock) around every threshold of codec_assign_sorter, 2 .. 12 rows"""
    rnd = np.random.default_rng(seed)
    out = []
    for r in range(rounds):
        n = int(rnd.integers(2, 13))
        base = float(rnd.choice([60, 90, 5000, 40000]))
        tests = []
        for c in ASSIGN_CAND[:n]:
            size = float(int(base * rnd.choice([1.0, 0.995, 0.99, 0.985, 0.975, 0.965, 0.95, 0.7, 1.3, 1.31])))
            clock = float(int(rnd.choice([0, 100, 900, 4999, 5000, 5001, 8000, 20000, 100000]) * rnd.choice([1.0, 0.19, 0.34, 0.66, 0.8, 0.86])))
            tests.append((c, size, clock))
        out.append(tests)
    return out


ASSIGN_NS = [[0] * 32,                                                                      # every device trial "under 5 ms"
             [0, 0, 0, 0, 0, 0, 4, 6, 5, 7] + [0] * 6 + [30, 45, 36, 50] + [0] * 12,       # a fast host: all under 5 ms at 50 KB
             [0, 0, 0, 0, 0, 0, 20, 30, 24, 36] + [0] * 6 + [80, 120, 90, 140] + [0] * 12]  # a slow one: the arithmetic coders over 5 ms


def assign_run_cases(n_random=420, seed=77):
    """inputs of assignref_run (oracle/ref_assign_shim.c): in[14], dict_id, txt_len, vb_size, the data's recipe, ns table -> ticks"""
    rnd = np.random.default_rng(seed)
    kinds = [("markov", 40), ("uniform", 4), ("u32be", 256), ("skew", 5), ("runs", 6), ("uniform", 200)]
    dids = {0: [0x11, 0x55, 0x41, 0x4c, 0, 0, 0, 0], 2: [0x51, 0x55, 0x41, 0x4c, 0, 0, 0, 0], 1: [0xd1, 0x55, 0x41, 0x4c, 0, 0, 0, 0]}   # field / DTYPE_2 / DTYPE_1 (dict_id.h:15-17)
    out = []
    for r in range(n_random):
        structured = r < 200
        mode = 0 if structured else int(rnd.choice([0, 0, 1, 2]))
        kind, nsym = kinds[int(rnd.integers(len(kinds)))]
        n = int(rnd.choice([30, 49, 50, 20000, 50000, 50000, 120000]))
        ns_i = int(rnd.integers(3)) if n in (20000, 50000) else 0
        sample = min(n, 99999)
        host_pay = [int(sample * f) for f in rnd.choice([0.2, 0.35, 0.5, 0.8, 1.1], 3)]
        host_clk = [int(x) for x in rnd.choice([300, 2000, 4900, 6000, 30000], 3)]
        ticks = [0] + [sample * ASSIGN_NS[ns_i][c] // 1000 for c in ASSIGN_CAND[1:9]] + host_clk
        dt = int(rnd.choice([0, 1, 2]))
        v_codec = int(rnd.choice([0, 0, 0, 0, 6, 13]))                       # UNKNOWN / a simple codec from the segmenter / a complex one (DOMQ)
        z_codec = int(rnd.choice([0, 0, 2]))                                 # nothing / a codec no trial can produce (so that a commit shows)
        inp = [mode, int(rnd.integers(2)), int(rnd.choice([1, 1, 2, 10, 10, 11])), v_codec, z_codec, int(rnd.choice([0, 3, 4, 5])) if mode == 1 else 0,
               int(rnd.random() < 0.15), int(rnd.choice([0, 0b100, 0b001, 0b010, 0b111])), int(rnd.random() < 0.2), int(rnd.random() < 0.15), int(rnd.random() < 0.1)] + host_pay
        out.append({"in": inp, "dict_id": dids[dt], "txt_len": int(rnd.choice([1000, 3 << 20, 5 << 20, 10 << 20])), "vb_size": int(rnd.choice([16 << 20, 6 << 20, 256 << 20])),
                    "data": [kind, 9000 + r, n, nsym], "ns": ns_i, "ticks": [int(t) for t in ticks]})
    return out


def assign_run_data(c):
    kind, seed, n, nsym = c["data"]
    return synth.stream(kind, seed, n, nsym).tobytes()


def assign_golden():
    with open(os.path.join(HERE, "golden", "assign_golden.json")) as f:
        return json.load(f)


# ---- row a4's loop: ctx_merge_in_one_vctx, pinned to the reference's own src/context.c (tests/golden/make_merge_golden.py) ------------------
def merge_loop_scenarios(n=600):
    """name -> (estimated_entries, [VBlock: dict(clone=k (the dictionary as it was after k VBlocks had merged), snips=[...], + keyword
    arguments of Zctx.merge)]). Multi-VBlock word streams with singletons that stay / fail, clones taken before earlier VBlocks merged (a
    batch), all-the-same contexts that can and cannot drop their b250 (flags, local data, pair rules, a different word), codecs inherited."""
    r = synth.u32(4242, 8 * n + 64).astype(np.int64)

    def ids(lo, hi):
        return [b"id%07d" % (i if r[i] % 9 else i - 1) for i in range(lo, hi)]

    def V(clone, snips, **kw):
        d = dict(clone=clone, snips=snips)
        d.update(kw)
        return d
    S = True
    return {
        "ids":    (0, [V(0, ids(0, n), can_have_singletons=S), V(1, ids(n // 2, n + n // 2), can_have_singletons=S), V(2, ids(0, 2 * n), can_have_singletons=S), V(3, ids(0, n // 3), can_have_singletons=S)]),
        "ids_batch": (5000, [V(0, ids(0, n), can_have_singletons=S), V(0, ids(n // 2, n + n // 2), can_have_singletons=S), V(0, ids(0, 2 * n), can_have_singletons=S),
                             V(3, ids(n, 3 * n), can_have_singletons=S), V(3, ids(0, n // 3), can_have_singletons=S)]),
        "tiles":  (0, [V(0, [b"%d" % (1101 + (i * 7 // n)) for i in range(n)]), V(1, [b"%d" % (1104 + (i * 9 // n)) for i in range(n)]),
                       V(1, [b"%d" % (1101 + (r[i] % 40)) for i in range(n)]), V(3, [b"%d" % (1090 + (r[i] % 80)) for i in range(n)])]),
        "big":    (0, [V(0, [b"w%d" % (r[i] % 3000) for i in range(n)]), V(1, [b"w%d" % (r[n + i] % 5000) for i in range(n)]), V(2, [b"w%d" % (i % 4000) for i in range(n)])]),
        "same":   (0, [V(0, [b"+"] * n), V(1, [b"+"] * n), V(2, [b"+"] * (n - 1) + [b"x"]), V(3, [b"+"] * n)]),
        "same_other_word": (0, [V(0, [b"a", b"b"] * 3), V(1, [b"b"] * 9), V(2, [b"a"] * 9), V(3, [b"b"] * 4)]),         # all-the-same on word 1: not droppable; then on word 0
        "same_flags": (0, [V(0, [b"+"] * 5, flags=0x01), V(1, [b"+"] * 5, flags=0x01), V(2, [b"+"] * 5, flags=0x02), V(3, [b"+"] * 5)]),   # store bits differ from VBlock 1's
        "same_local": (0, [V(0, [b"+"] * 5, local_len=40), V(1, [b"\x01"] * 5, local_len=40), V(2, [b"\x01"] * 5)]),    # local data: droppable only for a plain SNIP_LOOKUP
        "lookup_first": (0, [V(0, [b"\x01"] * 7, local_len=12), V(1, [b"\x01"] * 3, local_len=5), V(2, [b"\x01", b"z"], local_len=5)]),   # rm_dict_all_the_same, then overridden
        "same_pair": (0, [V(0, [b"="] * 6), V(1, [b"="] * 6, pair2_identical=True, b250_r1_len=3), V(2, [b"="] * 6, pair2_identical=True, local_r1_len=9),
                          V(3, [b"="] * 6, pair2_identical=True, local_r1_len=9, local_len=4), V(4, [b"="] * 6, no_drop_b250=True)]),
        "self_delta": (0, [V(0, [b"\x09" + b"1"] * 5), V(1, [b"\x09" + b"1"] * 5)]),                                    # SNIP_SELF_DELTA: never dropped
        "ston1":  (0, [V(0, [b"only-once"] + [b"rep"] * 5, can_have_singletons=S), V(1, [b"only-once", b"rep", b"new2", b"new2"], can_have_singletons=S),
                       V(2, [b"only-once", b"new3"], can_have_singletons=S), V(2, [b"new3", b"", None, b"rep"], can_have_singletons=S), V(4, [b"new3", b"new4"], can_have_singletons=S)]),
        "ston_mixed": (0, [V(0, [b"u%d" % i for i in range(40)] + [b"c"] * 9, can_have_singletons=S), V(1, [b"u%d" % i for i in range(20, 60)] + [b"c"], can_have_singletons=False),
                           V(2, [b"u%d" % i for i in range(70)], can_have_singletons=S), V(3, [b"v%d" % (i // 2) for i in range(50)], can_have_singletons=S)]),
    }


def merge_golden():
    with open(os.path.join(HERE, "golden", "merge_golden.json")) as f:
        return json.load(f)


# ---- row a15: the order of a VBlock's sections, pinned to the reference's own src/zip.c (tests/golden/make_order_golden.py) -----------------
def section_order_tables(n_tables=120):
    """[(ctxs = [(did_i, local_dep, has_local, ston_only, has_b250)] in shuffled table order, vblock_i)]"""
    out = []
    for seed in range(n_tables):
        r = synth.u32(7000 + seed, 200)
        n = 1 + int(r[0] % 30)
        dids = np.argsort(r[1:1 + n], kind="stable")
        ctxs = [(int(dids[i]) * 3 + 1, int(r[40 + i] % 3), bool(r[80 + i] % 4), bool(r[120 + i] % 5 == 0), bool(r[160 + i] % 3)) for i in range(n)]
        for vb_i in (1, 2, 7):
            out.append((ctxs, vb_i))
    return out


def order_golden():
    with open(os.path.join(HERE, "golden", "order_golden.json")) as f:
        return json.load(f)
