"""TEST INFRASTRUCTURE: an independent reader of what genozip_amd writes around the VBlocks - footer ->
SEC_GENOZIP_HEADER -> section list (file format, src/sections.h:496-515: 19-byte entries, deltas) -> SEC_TXT_HEADER / SEC_DICT /
SEC_COUNTS. Every field is located through tests/golden/hdr_golden.json - offsets, widths and byte order taken from the REFERENCE'S
OWN struct definitions (oracle/ref_hdr_shim.c compiled against src/sections.h; tests/golden/make_hdr_golden.py) - not through numbers
written down here: a field the product puts elsewhere, or in the other byte order, does not read back."""
import json
import os
import struct

MAGIC = 0x27052012
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hdr_golden.json")) as _f:
    GOLD = json.load(_f)
K = GOLD["constants"]
SEC_GENOZIP_HEADER, SEC_TXT_HEADER, SEC_VB_HEADER, SEC_DICT, SEC_B250, SEC_LOCAL, SEC_COUNTS = (K["SEC_GENOZIP_HEADER"], K["SEC_TXT_HEADER"], K["SEC_VB_HEADER"], K["SEC_DICT"],
                                                                                                K["SEC_B250"], K["SEC_LOCAL"], K["SEC_COUNTS"])


def field(struct_name, name, blob, at=0):
    """the value of a field of one of the reference's structs, located by the golden layout"""
    f = GOLD["layout"][struct_name]["fields"][name]
    raw = blob[at + f[0]:at + f[0] + f[1]]
    assert len(raw) == f[1], (struct_name, name, "truncated")
    if f[2] == "bytes":
        return bytes(raw)
    if f[2] == "bits":
        return (raw[0] >> f[3]) & ((1 << f[4]) - 1)
    return int.from_bytes(raw, "big" if f[2] == "be" else "little")


def size(struct_name):
    return GOLD["layout"][struct_name]["size"]


def _unzig(u):
    return -((u + 1) >> 1) if u & 1 else u >> 1


def read_file(blob, decode):
    """blob: the whole file; decode(codec, payload, ulen) -> bytes. -> dict(header fields, sections=[dict], dicts={dict_id: [words]}, counts={})"""
    foot = blob[-size("footer"):]
    assert field("footer", "magic", foot) == MAGIC, "footer"
    off = field("footer", "genozip_header_offset", foot)
    GH = size("genozip")
    h = blob[off:off + GH]
    assert field("ctx", "magic", h) == MAGIC and field("ctx", "section_type", h) == SEC_GENOZIP_HEADER
    clen, ulen = field("ctx", "data_compressed_len", h), field("ctx", "data_uncompressed_len", h)
    payload = decode(field("ctx", "codec", h), blob[off + GH:off + GH + clen], ulen)
    num_sections = field("genozip", "num_sections", h)
    SE = size("secent")
    assert len(payload) == SE * num_sections
    word = field("genozip", "minor_and_num_lines_word", h)         # { minor : 14, is_modified : 1, private : 1, num_lines_bound : 48 } little endian
    out = dict(version=(field("genozip", "version", h), word & 0x3fff), data_type=field("genozip", "data_type", h), recon_size=field("genozip", "recon_size", h),
               num_lines=int.from_bytes((word >> 16).to_bytes(8, "little"), "big"),   # zfile.c:965: BGEN64 of the 48-bit field
               vb_size=field("genozip", "segconf_vb_size", h), created=field("genozip", "created", h).split(b"\0")[0], flags=field("genozip", "flags", h),
               num_txt_files=field("genozip", "num_txt_files", h), std_seq_len=field("genozip", "std_seq_len", h), std_seq_lR2=field("genozip", "std_seq_lR2", h), sections=[])
    prev_off = prev_vb = prev_lines = 0
    prev_comp = None
    for i in range(num_sections):
        f = payload[SE * i:SE * i + SE]
        prev_off += field("secent", "offset_delta", f)
        prev_vb += _unzig(field("secent", "vblock_i_delta", f))
        c1 = field("secent", "comp_i_plus_1", f)
        comp = prev_comp if (i and c1 == 0) else (255 if c1 == 255 else c1 - 1)
        prev_comp = comp
        st = field("secent", "st", f)
        s = dict(offset=prev_off, vblock_i=prev_vb, comp_i=comp, st=st, flags=field("secent", "flags", f))
        if st in (SEC_DICT, SEC_B250, SEC_LOCAL, SEC_COUNTS):
            s["dict_id"] = field("secent", "dict_id", f) if field("secent", "is_dict_id", f) else out["sections"][field("secent", "dict_sec_i", f)]["dict_id"]
        elif st == SEC_VB_HEADER:
            prev_lines += _unzig(field("secent", "num_lines", f))
            s["num_lines"] = prev_lines
        out["sections"].append(s)
    for a, b in zip(out["sections"], out["sections"][1:]):
        a["size"] = b["offset"] - a["offset"]
    out["sections"][-1]["size"] = len(blob) - size("footer") - out["sections"][-1]["offset"]
    out["dicts"], out["counts"], out["txt_headers"], out["dict_flags"] = {}, {}, [], {}
    for s in out["sections"]:
        hd = blob[s["offset"]:s["offset"] + 400]
        assert field("ctx", "magic", hd) == MAGIC and field("ctx", "section_type", hd) == s["st"], (s, hd[:28])
        assert field("ctx", "vblock_i", hd) == s["vblock_i"] and field("ctx", "flags", hd) == s["flags"], ("section list vs header", s)
        clen, ulen, codec = field("ctx", "data_compressed_len", hd), field("ctx", "data_uncompressed_len", hd), field("ctx", "codec", hd)
        if s["st"] == SEC_TXT_HEADER:
            assert s["size"] == size("txt") + clen
            out["txt_headers"].append(dict(pair=field("txt", "pair", hd), txt_data_size=field("txt", "txt_data_size", hd), txt_num_lines=field("txt", "txt_num_lines", hd),
                                           max_lines_per_vb=field("txt", "max_lines_per_vb", hd), txt_filename=field("txt", "txt_filename", hd).split(b"\0")[0],
                                           comp_i=s["comp_i"], flav_prop=field("txt", "flav_prop", hd), src_codec=field("txt", "src_codec", hd)))
        elif s["st"] == SEC_DICT:
            D = size("dict")
            data = decode(codec, blob[s["offset"] + D:s["offset"] + D + clen], ulen)
            words = data[:-1].split(b"\0")
            assert len(words) == field("dict", "num_snips", hd) and field("dict", "dict_id", hd) == s["dict_id"]
            out["dicts"].setdefault(s["dict_id"], []).extend(words)
            out["dict_flags"][s["dict_id"]] = dict(all_the_same_wi=field("dict", "all_the_same_wi", hd))
        elif s["st"] == SEC_COUNTS:
            Cn = size("counts")
            data = decode(codec, blob[s["offset"] + Cn:s["offset"] + Cn + clen], ulen)
            assert field("counts", "dict_id", hd) == s["dict_id"]
            out["counts"][s["dict_id"]] = list(struct.unpack(">%dQ" % (ulen // 8), data))
    return out
