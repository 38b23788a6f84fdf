"""TEST INFRASTRUCTURE: makes tests/golden/hdr_golden.json from oracle/_ref/libhdrref.so - the reference's OWN on-disk structs
(src/sections.h, src/container.h) filled through their members by oracle/ref_hdr_shim.c. Run in the build container only
(needs /root/reference): `make -C oracle ref && python tests/golden/make_hdr_golden.py`.

Two kinds of vectors:
  layout : per struct its size and, per field, (offset, width in bytes, byte order) - found by handing the shim one distinctive value per
           field and looking where its bytes land. tests/gz_reader.py parses what the product writes with THIS map, so a field the
           product puts elsewhere (or in the other byte order) fails the reader-based tests.
  kat    : whole structs for fixed inputs (hex), compared byte for byte with what the product writes for the same inputs
           (tests/test_oracle.py::test_header_layouts, tests/test_emul.py).
"""
import ctypes as C
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
H = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libhdrref.so"))
u8, u32, u64, i64 = C.c_uint8, C.c_uint32, C.c_uint64, C.c_int64


def call(fn, *args):
    out = C.create_string_buffer(2048)
    n = getattr(H, fn)(*args, out)
    return out.raw[:n]


def locate(zero, probe, value, width):
    """where do the bytes of `value` (width bytes) sit in probe vs zero -> (offset, order)"""
    diff = [i for i in range(len(zero)) if zero[i] != probe[i]]
    assert diff, "field not found"
    lo, hi = diff[0], diff[-1]
    be, le = value.to_bytes(width, "big"), value.to_bytes(width, "little")
    for off in range(max(0, hi - width + 1), lo + 1):
        if probe[off:off + width] == be:
            return off, "be"
        if probe[off:off + width] == le:
            return off, "le"
    raise AssertionError("value bytes not found around %d..%d" % (lo, hi))


DID = b"ABCDEFGH"
V32, V64, V16 = 0x01020304, 0x1112131415161718, 0x2122


def ctx_args(**kw):
    a = dict(z_digest=0, clen=0, ulen=0, vblock_i=0, st=12, codec=0, sub_codec=0, flags=0, ltype=0, param=0, b250=0, dict_id=bytes(8))
    a.update(kw)
    return (u32(a["z_digest"]), u32(a["clen"]), u32(a["ulen"]), u32(a["vblock_i"]), a["st"], a["codec"], a["sub_codec"], a["flags"], a["ltype"], a["param"], a["b250"], a["dict_id"])


def layout_ctx():
    z = call("hdrref_ctx", *ctx_args())
    F = {}
    F["magic"] = [0, 4, "be"]
    for name, width, val in (("z_digest", 4, V32), ("clen", 4, V32), ("ulen", 4, V32), ("vblock_i", 4, V32)):
        F[{"clen": "data_compressed_len", "ulen": "data_uncompressed_len"}.get(name, name)] = list(locate(z, call("hdrref_ctx", *ctx_args(**{name: val})), val, width)[:1]) + [width, "be"]
    for name, key in (("section_type", "st"), ("codec", "codec"), ("sub_codec", "sub_codec"), ("flags", "flags"), ("ltype", "ltype"), ("param", "param")):
        base = ctx_args()
        p = call("hdrref_ctx", *ctx_args(**{key: 0x5a}))
        zz = call("hdrref_ctx", *ctx_args(**{key: 0}))
        F[name] = [locate(zz, p, 0x5a, 1)[0], 1, "be"]
    F["b250_size_or_nothing_char"] = [locate(call("hdrref_ctx", *ctx_args(st=11)), call("hdrref_ctx", *ctx_args(st=11, b250=4)), 4, 1)[0], 1, "be"]
    assert F["b250_size_or_nothing_char"][0] == locate(z, call("hdrref_ctx", *ctx_args(b250=0x7e)), 0x7e, 1)[0]
    p = call("hdrref_ctx", *ctx_args(dict_id=DID))
    F["dict_id"] = [p.index(DID), 8, "bytes"]
    return dict(size=len(z), fields=F)


def layout_vb():
    names = ("vblock_i", "recon_size", "z_data_bytes", "longest_line_len", "longest_seq_len")
    z = call("hdrref_vb", u32(0), u32(0), u32(0), u32(0), u32(0), 0)
    F = {"magic": [0, 4, "be"]}
    for i, nm in enumerate(names):
        args = [u32(0)] * 5
        args[i] = u32(V32)
        F[nm] = [locate(z, call("hdrref_vb", *args, 0), V32, 4)[0], 4, "be"]
    F["flags"] = [locate(z, call("hdrref_vb", u32(0), u32(0), u32(0), u32(0), u32(0), 0x5a), 0x5a, 1)[0], 1, "be"]
    F["section_type"] = [z.index(bytes([9]), 20), 1, "be"]
    return dict(size=len(z), fields=F)


def layout_dict():
    z = call("hdrref_dict", u32(0), u32(0), u32(0), 0, u32(0), 0, bytes(8))
    F = {"num_snips": [locate(z, call("hdrref_dict", u32(0), u32(0), u32(0), 0, u32(V32), 0, bytes(8)), V32, 4)[0], 4, "be"],
         "dict_id": [call("hdrref_dict", u32(0), u32(0), u32(0), 0, u32(0), 0, DID).index(DID), 8, "bytes"]}
    p = call("hdrref_dict", u32(0), u32(0), u32(0), 0, u32(0), 15, bytes(8))
    off = [i for i in range(len(z)) if z[i] != p[i]]
    assert len(off) == 1
    shift = (p[off[0]] // 15).bit_length() - 1
    assert p[off[0]] == 15 << shift
    F["all_the_same_wi"] = [off[0], 1, "bits", shift, 4]
    return dict(size=len(z), fields=F)


def layout_counts():
    z = call("hdrref_counts", u32(0), u32(0), u32(0), 0, i64(0), bytes(8))
    F = {"nodes_param": [locate(z, call("hdrref_counts", u32(0), u32(0), u32(0), 0, i64(V64), bytes(8)), V64, 8)[0], 8, "be"],
         "dict_id": [call("hdrref_counts", u32(0), u32(0), u32(0), 0, i64(0), DID).index(DID), 8, "bytes"]}
    return dict(size=len(z), fields=F)


def txt_args(**kw):
    a = dict(vblock_i=0, codec=0, pair=0, txt_data_size=0, txt_num_lines=0, max_lines_per_vb=0, src_codec=0, txt_filename=b"", txt_header_size=0, flav_prop=bytes(8))
    a.update(kw)
    return (u32(a["vblock_i"]), a["codec"], a["pair"], u64(a["txt_data_size"]), u64(a["txt_num_lines"]), u32(a["max_lines_per_vb"]), a["src_codec"], a["txt_filename"],
            u64(a["txt_header_size"]), a["flav_prop"])


def layout_txt():
    z = call("hdrref_txt", *txt_args())
    F = {}
    for nm, w, v in (("txt_data_size", 8, V64), ("txt_num_lines", 8, V64), ("max_lines_per_vb", 4, V32), ("txt_header_size", 8, V64), ("vblock_i", 4, V32)):
        F[nm] = [locate(z, call("hdrref_txt", *txt_args(**{nm: v})), v, w)[0], w, "be"]
    F["src_codec"] = [locate(z, call("hdrref_txt", *txt_args(src_codec=0x5a)), 0x5a, 1)[0], 1, "be"]
    p = call("hdrref_txt", *txt_args(pair=2))
    off = [i for i in range(len(z)) if z[i] != p[i]]
    assert len(off) == 1 and p[off[0]] == 2
    F["pair"] = [off[0], 1, "bits", 0, 2]
    F["txt_filename"] = [call("hdrref_txt", *txt_args(txt_filename=b"NAME.fq")).index(b"NAME.fq"), 256, "bytes"]
    fp = bytes([0xa1, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8])
    F["flav_prop"] = [call("hdrref_txt", *txt_args(flav_prop=fp)).index(fp), 8, "bytes"]
    F["z_digest"] = [4, 4, "be"]
    return dict(size=len(z), fields=F)


def gh_args(**kw):
    a = dict(z_digest=0, clen=0, ulen=0, flags=0, version=0, minor=0, data_type=0, recon_size=0, num_lines_field=0, num_sections=0, num_txt_files=0, created=b"",
             std_seq_len=0, std_seq_lR2=0, segconf_vb_size=0, lic_type=0)
    a.update(kw)
    return (u32(a["z_digest"]), u32(a["clen"]), u32(a["ulen"]), a["flags"], a["version"], a["minor"], C.c_uint16(a["data_type"]), u64(a["recon_size"]), u64(a["num_lines_field"]),
            u32(a["num_sections"]), a["num_txt_files"], a["created"], u32(a["std_seq_len"]), u32(a["std_seq_lR2"]), u32(a["segconf_vb_size"]), a["lic_type"])


def layout_genozip():
    z = call("hdrref_genozip", *gh_args())
    F = {}
    for nm, w, v in (("recon_size", 8, V64), ("num_sections", 4, V32), ("std_seq_len", 4, V32), ("std_seq_lR2", 4, V32), ("segconf_vb_size", 4, V32), ("data_type", 2, V16)):
        F[nm] = [locate(z, call("hdrref_genozip", *gh_args(**{nm: v})), v, w)[0], w, "be"]
    for nm in ("flags", "version", "num_txt_files", "lic_type"):
        F[nm] = [locate(z, call("hdrref_genozip", *gh_args(**{nm: 0x5a})), 0x5a, 1)[0], 1, "be"]
    F["created"] = [call("hdrref_genozip", *gh_args(created=b"CREATED")).index(b"CREATED"), 72, "bytes"]
    # the 64-bit little-endian word { minor : 14, is_modified : 1, private_file : 1, num_lines_bound : 48 }
    p = call("hdrref_genozip", *gh_args(minor=0x3fff))
    off = [i for i in range(len(z)) if z[i] != p[i]]
    word_at = off[0]
    assert int.from_bytes(p[word_at:word_at + 8], "little") == 0x3fff
    q = call("hdrref_genozip", *gh_args(num_lines_field=0xa1a2a3a4a5a6))
    assert int.from_bytes(q[word_at:word_at + 8], "little") == 0xa1a2a3a4a5a6 << 16
    F["minor_and_num_lines_word"] = [word_at, 8, "le"]
    return dict(size=len(z), fields=F)


def layout_secent():
    def se(**kw):
        a = dict(offset_delta=0, vblock_i_delta=0, comp=0, st=12, dict_id=bytes(8), is_dict_id=-1, dict_sec_i=0, num_lines=0, flags=0)
        a.update(kw)
        return call("hdrref_secent", u32(a["offset_delta"]), u32(a["vblock_i_delta"]), a["comp"], a["st"], a["dict_id"], a["is_dict_id"], u32(a["dict_sec_i"]), u32(a["num_lines"]), a["flags"])
    z = se()
    F = {"offset_delta": [locate(z, se(offset_delta=V32), V32, 4)[0], 4, "be"], "vblock_i_delta": [locate(z, se(vblock_i_delta=V32), V32, 4)[0], 4, "be"],
         "comp_i_plus_1": [locate(z, se(comp=0x5a), 0x5a, 1)[0], 1, "be"], "st": [locate(se(st=0), se(st=0x5a), 0x5a, 1)[0], 1, "be"],
         "flags": [locate(z, se(flags=0x5a), 0x5a, 1)[0], 1, "be"], "dict_id": [se(dict_id=DID, is_dict_id=1).index(DID), 8, "bytes"],
         "dict_sec_i": [locate(z, se(is_dict_id=0, dict_sec_i=V32), V32, 4)[0], 4, "be"], "num_lines": [locate(se(st=9), se(st=9, num_lines=V32), V32, 4)[0], 4, "be"]}
    F["is_dict_id"] = [F["dict_id"][0], 1, "be"]
    return dict(size=len(z), fields=F)


def layout_footer():
    z = call("hdrref_footer", u64(0))
    return dict(size=len(z), fields={"genozip_header_offset": [locate(z, call("hdrref_footer", u64(V64)), V64, 8)[0], 8, "be"], "magic": [z.index(bytes.fromhex("27052012")), 4, "be"]})


def main():
    sizes = (u32 * 32)()
    n = H.hdrref_sizes(sizes)
    names = ("SectionHeader", "SectionHeaderCtx", "SectionHeaderVbHeader", "SectionHeaderDictionary", "SectionHeaderCounts", "SectionHeaderTxtHeader",
             "SectionHeaderGenozipHeader", "SectionFooterGenozipHeader", "SectionEntFileFormat", "ContainerItem", "Container_0", "SEC_TXT_HEADER", "SEC_VB_HEADER", "SEC_DICT",
             "SEC_B250", "SEC_LOCAL", "SEC_COUNTS", "SEC_GENOZIP_HEADER", "DT_FASTQ", "NUM_QTYPES", "CODEC_ACGT", "CODEC_LZMA", "CODEC_XCGT", "LT_BLOB", "LT_CODEC", "LT_SUPP")
    G = dict(constants=dict(zip(names, list(sizes[:n]))),
             layout=dict(ctx=layout_ctx(), vb=layout_vb(), dict=layout_dict(), counts=layout_counts(), txt=layout_txt(), genozip=layout_genozip(), secent=layout_secent(), footer=layout_footer()))
    # ---- whole structs for fixed inputs
    K = []
    K.append(dict(kind="ctx", args=dict(z_digest=0xdeadbeef, clen=1234, ulen=56789, vblock_i=7, st=12, codec=16, sub_codec=0, flags=0x05, ltype=3, param=9, b250=0xff, dict_id="d1304e414d450000"),
                  hex=call("hdrref_ctx", *ctx_args(z_digest=0xdeadbeef, clen=1234, ulen=56789, vblock_i=7, st=12, codec=16, flags=5, ltype=3, param=9, b250=0xff, dict_id=bytes.fromhex("d1304e414d450000"))).hex()))
    K.append(dict(kind="ctx", args=dict(z_digest=1, clen=1, ulen=1, vblock_i=300, st=11, codec=1, sub_codec=0, flags=0x24, ltype=0, param=0, b250=4, dict_id="1451424954 4d4150".replace(" ", "")),
                  hex=call("hdrref_ctx", *ctx_args(z_digest=1, clen=1, ulen=1, vblock_i=300, st=11, codec=1, flags=0x24, b250=4, dict_id=bytes.fromhex("14514249544d4150"))).hex()))
    K.append(dict(kind="vb", args=dict(vblock_i=5, recon_size=14720000, z_data_bytes=3300000, longest_line_len=372, longest_seq_len=150),
                  hex=call("hdrref_vb", u32(5), u32(14720000), u32(3300000), u32(372), u32(150), 0).hex()))
    K.append(dict(kind="txt", args=dict(pair=1, txt_data_size=367000000, txt_num_lines=1000000, max_lines_per_vb=40000, txt_filename="reads_R1.fq"),
                  hex=call("hdrref_txt", *txt_args(vblock_i=1, codec=1, pair=1, txt_data_size=367000000, txt_num_lines=1000000, max_lines_per_vb=40000, src_codec=1, txt_filename=b"reads_R1.fq")).hex()))
    K.append(dict(kind="footer", args=dict(offset=0x123456789a), hex=call("hdrref_footer", u64(0x123456789a)).hex()))
    # section list entries (sections_list_memory_to_file_format, src/sections.c:481-534): a VB header, a first appearance of a dict_id, a later one
    K.append(dict(kind="secent", args=dict(offset_delta=400, vblock_i_delta=2, comp_i_plus_1=1, st=9, num_lines_delta=80000, flags=0),
                  hex=call("hdrref_secent", u32(400), u32(2), 1, 9, bytes(8), -1, u32(0), u32(80000), 0).hex()))
    K.append(dict(kind="secent", args=dict(offset_delta=84, vblock_i_delta=0, comp_i_plus_1=0, st=12, dict_id="d1314e414d450000", flags=4),
                  hex=call("hdrref_secent", u32(84), u32(0), 0, 12, bytes.fromhex("d1314e414d450000"), 1, u32(0), u32(0), 4).hex()))
    K.append(dict(kind="secent", args=dict(offset_delta=99, vblock_i_delta=0, comp_i_plus_1=0, st=11, dict_sec_i=17, flags=0x20),
                  hex=call("hdrref_secent", u32(99), u32(0), 0, 11, bytes(8), 0, u32(17), u32(0), 0x20).hex()))
    # containers: the QNAME flavor container and the FASTQ TOPLEVEL of genozip_amd/fastq.py, as the reference's struct lays them out
    import sys
    sys.path.insert(0, ROOT)
    from genozip_amd import fastq as fq
    def con(items, repeats, flags):
        blob = b"".join(d + (s + b"\0\0")[:2] for d, s in items)
        return call("hdrref_container", u32(len(items)), u32(repeats), flags, 0, 0, blob)
    q1 = [(fq.dict_id("Q0NAME", 1), bytes([8, 3])), (fq.dict_id("Q1NAME", 1), b":"), (fq.dict_id("Q2NAME", 1), b":"), (fq.dict_id("Q3NAME", 1), b":"), (fq.dict_id("Q4NAME", 1), b"")]
    K.append(dict(kind="container", args=dict(name="illumina-7", repeats=1, flags=0), hex=con(q1, 1, 0).hex()))
    top = [(fq.dict_id(t), b"") for t in ("QNAME", "QNAME2", "E1L", "SQBITMAP", "E2L", "E2L", "QUAL", "E2L")]
    K.append(dict(kind="container", args=dict(name="fastq-toplevel", repeats=40000, flags=0x5c), hex=con(top, 40000, 0x5c).hex()))
    G["kat"] = K
    with open(os.path.join(HERE, "hdr_golden.json"), "w") as f:
        json.dump(G, f, indent=1, sort_keys=True)
    print("layouts:", {k: v["size"] for k, v in G["layout"].items()}, "kats:", len(K))


if __name__ == "__main__":
    main()
