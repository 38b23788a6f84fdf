"""TEST INFRASTRUCTURE: generates tests/golden/ctx_golden.json from the REFERENCE'S OWN sources compiled in place by
`make -C oracle ref` (oracle/_ref/libctxref.so + oracle/ref_ctx_shim.c: b250.c, dyn_int.c, buffer.c, codec_domq.c, base64.c,
codec_acgt.c, hash.c; oracle/_ref/libcompref.so + oracle/ref_comp_shim.c: compressor.c, codec_none.c, libdeflate adler32,
htscodecs) - rows a2 / a5, a3 / a7, a6, a1's and a4's hash, a10, N2, N3 of SURVEY 8(a) / 8(f). Only runs where /root/reference exists; the
vectors (generator parameters + outputs as hex or sha1) are committed, the reference is not.

    python tests/golden/make_ctx_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import cases          # noqa: E402
import pyoracle       # noqa: E402
from genozip_amd import synth   # noqa: E402


def enc(b):
    return {"hex": b.hex()} if len(b) <= 96 else {"sha1": hashlib.sha1(b).hexdigest(), "len": len(b)}


def main():
    R = pyoracle.CtxRef()
    out = {"b250": [], "dyn_int": [], "transpose": [], "local_order": []}
    out["hash_do"] = [{"hash_len": hl, "snip_hex": sn.hex(), "hash": R.hash_do(hl, sn)} for hl, sn in cases.hash_cases()]
    import parity
    out["domq"] = []
    for name, ls in cases.domq_cases():
        t, o, l = parity._snip_column(ls)
        o = np.where(l == 0, 0, o).astype(np.uint32)
        r = R.domq(t, o, l)
        out["domq"].append({"name": name, "qual": enc(r["qual"]), "runs": enc(r["runs"]), "mplx": enc(r["mplx"]), "divr": enc(r["divr"]),
                            "denorm_snip": r["denorm_snip"].decode(), "param": r["param"], "fit": r["fit"]})
    if pyoracle.CompRef.available():        # a10: the reference's own comp_compress
        CR = pyoracle.CompRef()
        out["sections"] = []
        for f, data in cases.section_cases():
            d = pyoracle.GzoCtxSectionDesc(**{k: v for k, v in f.items() if k != "dict_id"})
            d.dict_id[:] = list(f["dict_id"])
            out["sections"].append({"codec": f["codec"], "n": len(data), "z": enc(CR.section(d, data))})
    out["seg_nodes"] = [{"node_index_sha1": hashlib.sha1(np.array(R.seg_nodes(ol, sn), dtype=np.int32).tobytes()).hexdigest(), "n": len(sn)}
                        for ol, sn in cases.seg_node_cases()]       # a1: hash_get_entry_for_seg
    f, v = R.str_get_int(cases.int_snip_cases())                 # a3: str_get_int
    out["str_get_int"] = {"is_int": "".join(str(x) for x in f), "values_sha1": hashlib.sha1(np.array([y for x, y in zip(f, v) if x], dtype=np.int64).tobytes()).hexdigest()}
    out["merge_hash"] = []                  # a4's hash and singleton tables: the reference's own hash.c
    for name, est, vbs in cases.merge_hash_cases():
        m = R.merge_hash(est, vbs)
        out["merge_hash"].append({"name": name, "word": m["word"], "ston": m["ston"], "dict": enc(m["dict"]), "n_failed": m["n_failed"], "hash_len": m["hash_len"]})
    out["acgt"] = []
    for name, seq in cases.acgt_cases():
        pk, x, hx, sub = R.acgt(seq)
        out["acgt"].append({"name": name, "packed": enc(pk), "x": enc(x if hx else b""), "has_x": hx, "sub_codec": sub})
    for lt, w in cases.LOCAL_ORDER_CASES:
        raw = synth.uniform_bytes(40 + lt, 500 * w, 256).tobytes()
        fo = R.local_to_file_order(lt, raw, w)
        lt2, back = R.local_to_native(lt, fo, w)
        assert back == raw and lt2 == lt
        out["local_order"].append({"ltype": lt, "w": w, "file": enc(fo)})
    for seed in range(60):
        ne = [1, 2, 7, 300, 5000, 40000][seed % 6]
        ol, nn = [(0, 5), (4, 0), (1500, 700), (17000, 300), (2200000, 10), (900, 200)][seed // 10]
        ats = seed % 7 == 0
        ni, n2w = cases.b250_ctx_case(seed, ne, ol, nn, ats)
        seg, cnt, flag = R.b250_seg(ni, ol)
        gen = R.b250_generate(seg, cnt, flag, ol, n2w)
        out["b250"].append({"seed": seed, "n": ne, "ol": ol, "n_new": nn, "ats": ats, "seg": enc(seg), "count": cnt, "all_the_same": flag, "piz": enc(gen)})
    for i, (vals, isn, nc) in enumerate(cases.dyn_int_cases()):
        lt, raw = R.dyn_int_column(vals, isn, nc)
        out["dyn_int"].append({"case": i, "ltype": lt, "raw": enc(raw)})
    for (lt, w, rows, cols) in cases.TRANSPOSE_CASES:
        raw = synth.uniform_bytes(9 + rows, rows * cols * w, 256).tobytes()
        lt2, tr = R.dyn_int_transpose(lt, raw, rows * cols, cols)
        out["transpose"].append({"ltype": lt, "w": w, "rows": rows, "cols": cols, "ltype_out": lt2, "out": enc(tr)})
    with open(os.path.join(HERE, "ctx_golden.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
