"""TEST INFRASTRUCTURE: an independent reader of what genozip_amd writes after the VBlocks - footer ->
SEC_GENOZIP_HEADER -> section list (file format, src/sections.h:496-515: 19-byte entries, deltas) -> SEC_DICT / SEC_COUNTS.
Written from the format description (sections.h:146-307), not from the product's writer."""
import struct

MAGIC = 0x27052012
SEC_GENOZIP_HEADER, SEC_TXT_HEADER, SEC_VB_HEADER, SEC_DICT, SEC_B250, SEC_LOCAL, SEC_COUNTS = 6, 8, 9, 10, 11, 12, 17


def _unzig(u):
    return -((u + 1) >> 1) if u & 1 else u >> 1


def read_file(blob, decode):
    """blob: the whole file; decode(codec, payload, ulen) -> bytes. -> dict(header fields, sections=[dict], dicts={dict_id: [words]}, counts={})"""
    off, magic = struct.unpack(">QI", blob[-12:])
    assert magic == MAGIC, "footer"
    h = blob[off:off + 720]
    assert struct.unpack(">I", h[:4])[0] == MAGIC and h[24] == SEC_GENOZIP_HEADER
    clen, ulen = struct.unpack(">II", h[12:20])
    payload = decode(h[25], blob[off + 720:off + 720 + clen], ulen)
    num_sections = struct.unpack(">I", h[48:52])[0]
    assert len(payload) == 19 * num_sections
    bits = int.from_bytes(h[40:48], "little")
    out = dict(version=(h[28], bits & 0x3fff), data_type=struct.unpack(">H", h[30:32])[0], recon_size=struct.unpack(">Q", h[32:40])[0],
               num_lines=int.from_bytes((bits >> 16).to_bytes(8, "little"), "big"),   # zfile.c:965: BGEN64 of the 48-bit field
               vb_size=struct.unpack(">I", h[715:719])[0], created=h[88:160].split(b"\0")[0], sections=[])
    prev_off = prev_vb = prev_lines = 0
    prev_comp = None
    for i in range(num_sections):
        f = payload[19 * i:19 * i + 19]
        prev_off += struct.unpack(">I", f[0:4])[0]
        prev_vb += _unzig(struct.unpack(">I", f[4:8])[0])
        comp = prev_comp if (i and f[8] == 0) else (255 if f[8] == 255 else f[8] - 1)
        prev_comp = comp
        s = dict(offset=prev_off, vblock_i=prev_vb, comp_i=comp, st=f[9], flags=f[18])
        if f[9] in (SEC_DICT, SEC_B250, SEC_LOCAL, SEC_COUNTS):
            s["dict_id"] = bytes(f[10:18]) if f[10] else out["sections"][struct.unpack(">I", f[14:18])[0]]["dict_id"]
        elif f[9] == SEC_VB_HEADER:
            prev_lines += _unzig(struct.unpack(">I", f[10:14])[0])
            s["num_lines"] = prev_lines
        out["sections"].append(s)
    for a, b in zip(out["sections"], out["sections"][1:]):
        a["size"] = b["offset"] - a["offset"]
    out["sections"][-1]["size"] = len(blob) - 12 - out["sections"][-1]["offset"]
    out["dicts"], out["counts"], out["txt_headers"] = {}, {}, []
    for s in out["sections"]:
        hd = blob[s["offset"]:s["offset"] + 44]
        assert struct.unpack(">I", hd[:4])[0] == MAGIC and hd[24] == s["st"], (s, hd[:28])
        if s["st"] == SEC_TXT_HEADER:                                       # src/sections.h:308-327
            th = blob[s["offset"]:s["offset"] + 400]
            out["txt_headers"].append(dict(pair=th[27] & 3, txt_data_size=struct.unpack(">Q", th[28:36])[0], txt_num_lines=struct.unpack(">Q", th[36:44])[0],
                                           max_lines_per_vb=struct.unpack(">I", th[44:48])[0], txt_filename=th[84:340].split(b"\0")[0], comp_i=s["comp_i"]))
        if s["st"] == SEC_DICT:
            clen, ulen = struct.unpack(">II", hd[12:20])
            data = decode(hd[26] if hd[25] == 13 else hd[25], blob[s["offset"] + 40:s["offset"] + 40 + clen], ulen)   # (CODEC_DOMQ: its sub-codec)
            words = data[:-1].split(b"\0")
            assert len(words) == struct.unpack(">I", hd[28:32])[0] and hd[32:40] == s["dict_id"]
            out["dicts"].setdefault(s["dict_id"], []).extend(words)
        elif s["st"] == SEC_COUNTS:
            clen, ulen = struct.unpack(">II", hd[12:20])
            data = decode(hd[25], blob[s["offset"] + 44:s["offset"] + 44 + clen], ulen)
            out["counts"][s["dict_id"]] = list(struct.unpack(">%dQ" % (ulen // 8), data))
    return out
