"""pyoracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes bindings for (a) oracle/liboracle.so, this repo's CPU restatement of the hot path, and (b), when present,
oracle/_ref/libhtsref.so, the reference's own vendored htscodecs sources compiled in place (see oracle/Makefile).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing under
genozip_amd/ does.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "liboracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libhtsref.so")
CTXREF_SO = os.path.join(_HERE, "_ref", "libctxref.so")
ORDERREF_SO = os.path.join(_HERE, "_ref", "liborderref.so")
MERGEREF_SO = os.path.join(_HERE, "_ref", "libmergeref.so")
ASSIGNREF_SO = os.path.join(_HERE, "_ref", "libassignref.so")
COMPREF_SO = os.path.join(_HERE, "_ref", "libcompref.so")

CODEC_NONE, CODEC_RANB, CODEC_RANW, CODEC_RANb, CODEC_RANw = 1, 6, 7, 8, 9
CODEC_ARTB, CODEC_ARTW, CODEC_ARTb, CODEC_ARTw = 16, 17, 18, 19
SIMPLE_CODECS = (CODEC_RANB, CODEC_RANW, CODEC_RANb, CODEC_RANw, CODEC_ARTB, CODEC_ARTW, CODEC_ARTb, CODEC_ARTw)
CODEC_ORDER = {6: 0x01, 7: 0x19, 8: 0x81, 9: 0x99, 16: 0x01, 17: 0x19, 18: 0x81, 19: 0x99}
CODEC_NAME = {1: "NONE", 6: "RANB", 7: "RANW", 8: "RANb", 9: "RANw", 16: "ARTB", 17: "ARTW", 18: "ARTb", 19: "ARTw"}


def build(ref=True):
    """make liboracle.so, and _ref/libhtsref.so when the reference sources are available (never at run time on
    the GPU box, which only uses the prebuilt files)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    if ref and os.path.isdir("/root/reference/src/htscodecs"):
        subprocess.run(["make", "-s", "-C", _HERE, "ref"], check=True)


_u8p = ctypes.POINTER(ctypes.c_uint8)


def _buf(b):
    return (ctypes.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) if len(b) else b"\0")


class GzoCtxSectionDesc(ctypes.Structure):
    _fields_ = [("vblock_i", ctypes.c_uint32), ("section_type", ctypes.c_uint8), ("codec", ctypes.c_uint8),
                ("sub_codec", ctypes.c_uint8), ("flags", ctypes.c_uint8), ("ltype", ctypes.c_uint8),
                ("param", ctypes.c_uint8), ("b250_size_or_nothing_char", ctypes.c_uint8),
                ("dict_id", ctypes.c_uint8 * 8)]


class Oracle:
    """this repo's CPU restatement"""

    def __init__(self, path=ORACLE_SO):
        if not os.path.exists(path):
            build(ref=False)
        L = self.L = ctypes.CDLL(path)
        for n in ("gzo_rans_compress", "gzo_arith_compress", "gzo_rans_uncompress", "gzo_arith_uncompress",
                  "gzo_b250_generate", "gzo_b250_piz_decode", "gzo_section_compress", "gzo_section_frame"):
            getattr(L, n).restype = ctypes.c_long
        L.gzo_last_shift_ratio.restype = ctypes.c_double
        for n in ("gzo_rans_bound", "gzo_arith_bound", "gzo_codec_est_size", "gzo_adler32", "gzo_b250_seg_put",
                  "gzo_b250_piz_put", "gzo_lt_width"):
            getattr(L, n).restype = ctypes.c_uint32
        L.gzo_codec_est_size.argtypes = [ctypes.c_int, ctypes.c_uint64]
        L.gzo_adler32.argtypes = [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t]

    # ---- htscodecs level
    def hts_compress(self, kind, data, order):
        data = bytes(data)
        bound = (self.L.gzo_rans_bound if kind == "rans" else self.L.gzo_arith_bound)(len(data), order) + 1024
        out = ctypes.create_string_buffer(bound)
        f = self.L.gzo_rans_compress if kind == "rans" else self.L.gzo_arith_compress
        n = f(data, len(data), out, bound, order)
        if n < 0:
            raise RuntimeError("oracle %s compress failed" % kind)
        return out.raw[:n]

    def hts_uncompress(self, kind, comp, out_len):
        comp = bytes(comp)
        out = ctypes.create_string_buffer(max(1, out_len))
        f = self.L.gzo_rans_uncompress if kind == "rans" else self.L.gzo_arith_uncompress
        n = f(comp, len(comp), out, out_len)
        if n != out_len:
            raise RuntimeError("oracle %s uncompress failed (%d)" % (kind, n))
        return out.raw[:out_len]

    def last_shift_ratio(self):
        return self.L.gzo_last_shift_ratio()

    # ---- codec level
    def est_size(self, codec, n):
        return self.L.gzo_codec_est_size(codec, n)

    def codec_compress(self, codec, data, cap=None, soft_fail=False):
        data = bytes(data)
        cap = self.est_size(codec, len(data)) if cap is None else cap
        out = ctypes.create_string_buffer(max(1, cap))
        ol = ctypes.c_uint32(cap)
        rc = self.L.gzo_codec_compress(codec, data, len(data), out, ctypes.byref(ol), int(soft_fail))
        if rc == 0:
            return None
        if rc != 1:
            raise RuntimeError("oracle codec_compress failed")
        return out.raw[:ol.value]

    def codec_uncompress(self, codec, comp, out_len):
        comp = bytes(comp)
        out = ctypes.create_string_buffer(max(1, out_len))
        if self.L.gzo_codec_uncompress(codec, comp, len(comp), out, ctypes.c_uint64(out_len)) != 1:
            raise RuntimeError("oracle codec_uncompress failed")
        return out.raw[:out_len]

    def codec_compress_many(self, codecs, datas, n_threads, replicas=1):
        n = len(datas)
        ins = [bytes(d) for d in datas]
        caps = [self.est_size(c, len(d)) for c, d in zip(codecs, ins)]
        outs = [ctypes.create_string_buffer(max(1, c)) for c in caps]
        a_codecs = (ctypes.c_int * n)(*codecs)
        a_ins = (ctypes.c_char_p * n)(*ins)
        a_lens = (ctypes.c_uint32 * n)(*[len(d) for d in ins])
        a_outs = (ctypes.c_void_p * n)(*[ctypes.addressof(o) for o in outs])
        a_olens = (ctypes.c_uint32 * n)(*caps)
        if self.L.gzo_codec_compress_many_rep(n, a_codecs, a_ins, a_lens, a_outs, a_olens, n_threads, int(replicas)) != 0:
            raise RuntimeError("oracle compress_many failed")
        return [o.raw[:l] for o, l in zip(outs, a_olens)]

    def assign_best(self, data):
        data = bytes(data)
        sizes = (ctypes.c_uint32 * 9)()
        c = self.L.gzo_codec_assign_best(data, len(data), sizes)
        return c, list(sizes)

    # ---- b250
    def b250_seg(self, node_indices, ol_nodes_len):
        out = bytearray()
        tmp = ctypes.create_string_buffer(4)
        for ni in node_indices:
            n = self.L.gzo_b250_seg_put(tmp, int(ni), ol_nodes_len)
            assert n
            out += tmp.raw[:n]
        return bytes(out)

    def b250_seg_array(self, node_indices, ol_nodes_len):
        """numpy int32 array of node indices -> seg-format bytes (C loop; for the multi-million-entry cases)"""
        import numpy as np
        a = np.ascontiguousarray(node_indices, dtype=np.int32)
        out = ctypes.create_string_buffer(4 * len(a) + 4)
        self.L.gzo_b250_seg_put_many.restype = ctypes.c_uint64
        n = self.L.gzo_b250_seg_put_many(out, a.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(len(a)), ol_nodes_len)
        assert n or not len(a)
        return out.raw[:n]

    def b250_piz(self, wis):
        out = bytearray()
        tmp = ctypes.create_string_buffer(4)
        for wi in wis:
            n = self.L.gzo_b250_piz_put(tmp, int(wi))
            out += tmp.raw[:n]
        return bytes(out)

    def b250_generate(self, seg, ol_nodes_len, node2word):
        seg = bytes(seg)
        n2w = (ctypes.c_int32 * max(1, len(node2word)))(*node2word)
        out = ctypes.create_string_buffer(max(1, len(seg)))
        n = self.L.gzo_b250_generate(seg, len(seg), ol_nodes_len, n2w, len(node2word), out)
        if n < 0:
            raise RuntimeError("oracle b250_generate failed")
        return out.raw[:n]

    def b250_decode(self, piz):
        piz = bytes(piz)
        wi = (ctypes.c_int32 * max(1, len(piz)))()
        n = self.L.gzo_b250_piz_decode(piz, len(piz), wi, len(piz))
        if n < 0:
            raise RuntimeError("oracle b250 decode failed")
        return list(wi[:n])

    # ---- local
    def local_generate(self, ltype, raw_native_le, transpose_cols=0):
        w = self.L.gzo_lt_width(ltype)
        buf = ctypes.create_string_buffer(bytes(raw_native_le), max(1, len(raw_native_le)))
        scratch = ctypes.create_string_buffer(max(1, len(raw_native_le)))
        lt = self.L.gzo_local_generate(ltype, buf, ctypes.c_uint64(len(raw_native_le) // w), transpose_cols, scratch)
        return lt, buf.raw[:len(raw_native_le)]

    # ---- CODEC_ACGT pre-transform
    def acgt_pack(self, seq):
        seq = bytes(seq)
        self.L.gzo_acgt_packed_len.restype = ctypes.c_uint64
        pl = self.L.gzo_acgt_packed_len(ctypes.c_uint64(len(seq)))
        packed = ctypes.create_string_buffer(max(1, pl))
        x = ctypes.create_string_buffer(max(1, len(seq)))
        has_x = self.L.gzo_acgt_pack(seq, ctypes.c_uint64(len(seq)), packed, x)
        return packed.raw[:pl], x.raw[:len(seq)], bool(has_x)

    def acgt_unpack(self, packed, x, n):
        out = ctypes.create_string_buffer(max(1, n))
        self.L.gzo_acgt_unpack(bytes(packed), None if x is None else bytes(x), ctypes.c_uint64(n), out)
        return out.raw[:n]

    # ---- seg-side appends, a column at a time (rows a1-a3)
    def ctx_seg_column(self, text, off, length, ol_snips=(), pre=b""):
        """-> dict(node_index, dict, node_char_index, node_snip_len, counts, b250, b250_count, all_the_same)
        pre: a snip whose node is created before anything is segged (R2 VBlocks' mate_lookup node, fastq.c:664-665)"""
        import numpy as np
        text = bytes(text)
        off = np.ascontiguousarray(off, dtype=np.uint32); length = np.ascontiguousarray(length, dtype=np.uint32)
        n = len(off)
        ol_dict = b"".join(bytes(w) + b"\0" for w in ol_snips)
        ol_len = np.array([len(w) for w in ol_snips], dtype=np.uint32)
        ol_ci = np.zeros(len(ol_snips), dtype=np.uint64)
        if len(ol_snips) > 1:
            ol_ci[1:] = np.cumsum(ol_len[:-1].astype(np.uint64) + 1)
        n_ol = len(ol_snips)

        class Col(ctypes.Structure):
            _fields_ = [("node_index", ctypes.c_void_p), ("dict", ctypes.c_void_p), ("dict_len", ctypes.c_uint64),
                        ("node_char_index", ctypes.c_void_p), ("node_snip_len", ctypes.c_void_p), ("n_new", ctypes.c_uint32),
                        ("counts", ctypes.c_void_p), ("b250", ctypes.c_void_p), ("b250_len", ctypes.c_uint64),
                        ("b250_count", ctypes.c_uint64), ("all_the_same", ctypes.c_int)]
        ni = np.zeros(max(1, n), dtype=np.int32)
        dic = np.zeros(int(length.astype(np.uint64).sum()) + n + 1 + len(pre) + 1, dtype=np.uint8)
        nci = np.zeros(n + 2, dtype=np.uint64); nsl = np.zeros(n + 2, dtype=np.uint32)
        counts = np.zeros(n_ol + n + 2, dtype=np.uint32)
        b250 = np.zeros(4 * n + 4, dtype=np.uint8)
        c = Col(ni.ctypes.data, dic.ctypes.data, 0, nci.ctypes.data, nsl.ctypes.data, 0, counts.ctypes.data, b250.ctypes.data, 0, 0, 0)
        rc = self.L.gzo_ctx_seg_column_pre(text, off.ctypes.data_as(ctypes.c_void_p), length.ctypes.data_as(ctypes.c_void_p),
                                           ctypes.c_uint64(n), ol_dict, ol_ci.ctypes.data_as(ctypes.c_void_p),
                                           ol_len.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(n_ol), bytes(pre), ctypes.c_uint32(len(pre)), ctypes.byref(c))
        assert rc == 0
        return dict(node_index=ni[:n].copy(), dict=dic[:c.dict_len].tobytes(), node_char_index=nci[:c.n_new].copy(),
                    node_snip_len=nsl[:c.n_new].copy(), counts=counts[:n_ol + c.n_new].copy(),
                    b250=b250[:c.b250_len].tobytes(), b250_count=int(c.b250_count), all_the_same=bool(c.all_the_same))

    def dyn_int_column(self, values, is_nothing=None, nothing_char=0):
        """-> (ltype, native little-endian bytes at the final width)"""
        import numpy as np
        v = np.ascontiguousarray(values, dtype=np.int64)
        m = None if is_nothing is None else np.ascontiguousarray(is_nothing, dtype=np.uint8)
        out = np.zeros(8 * len(v) + 8, dtype=np.uint8)
        lt = self.L.gzo_dyn_int_column(v.ctypes.data_as(ctypes.c_void_p), None if m is None else m.ctypes.data_as(ctypes.c_void_p),
                                       ctypes.c_uint64(len(v)), int(nothing_char), out.ctypes.data_as(ctypes.c_void_p))
        w = {1: 1, 2: 1, 3: 2, 4: 2, 5: 4, 6: 4, 7: 8}[lt]
        return lt, out[:w * len(v)].tobytes()

    def local_blob_column(self, text, off, length, add_nul=False, pre=b"", pad_to=0, pad_byte=0, want_off=False):
        import numpy as np
        text = bytes(text)
        off = np.ascontiguousarray(off, dtype=np.uint32); length = np.ascontiguousarray(length, dtype=np.uint32)
        if pre or pad_to or want_off:
            out = np.zeros(int(length.astype(np.uint64).sum()) + len(off) * (1 + len(pre) + pad_to) + 1, dtype=np.uint8)
            io = np.zeros(len(off) + 1, dtype=np.uint32)
            self.L.gzo_local_blob_column_ex.restype = ctypes.c_uint64
            n = self.L.gzo_local_blob_column_ex(text, off.ctypes.data_as(ctypes.c_void_p), length.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(len(off)),
                                                int(bool(add_nul)), bytes(pre), len(pre), int(pad_to), int(pad_byte), out.ctypes.data_as(ctypes.c_void_p),
                                                io.ctypes.data_as(ctypes.c_void_p))
            return (out[:n].tobytes(), io[:len(off)]) if want_off else out[:n].tobytes()
        out = np.zeros(int(length.astype(np.uint64).sum()) + len(off) + 1, dtype=np.uint8)
        self.L.gzo_local_blob_column.restype = ctypes.c_uint64
        n = self.L.gzo_local_blob_column(text, off.ctypes.data_as(ctypes.c_void_p), length.ctypes.data_as(ctypes.c_void_p),
                                         ctypes.c_uint64(len(off)), int(bool(add_nul)), out.ctypes.data_as(ctypes.c_void_p))
        return out[:n].tobytes()

    # ---- N1 for BAM
    def bam_records(self, bam):
        """-> record offsets, or raises ValueError(index of the bad record)"""
        import numpy as np
        bam = bytes(bam)
        out = np.zeros(len(bam) // 36 + 1, dtype=np.uint32)
        self.L.gzo_bam_records.restype = ctypes.c_int64
        n = self.L.gzo_bam_records(bam, ctypes.c_uint64(len(bam)), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(len(out)))
        if n < 0:
            raise ValueError(-1 - n)
        return out[:n].copy()

    def bam_to_sam(self, bam, rec_off, ref_names):
        """-> (text, line_off [n + 1]), or raises ValueError(index of the bad record)"""
        import numpy as np
        bam = bytes(bam)
        rec_off = np.ascontiguousarray(rec_off, dtype=np.uint32)
        names = b"".join(ref_names)
        noff = np.concatenate([[0], np.cumsum([len(x) for x in ref_names])]).astype(np.uint32)
        cap = 6 * len(bam) + 64 * len(rec_off) + 1024
        text, lo = np.zeros(cap, dtype=np.uint8), np.zeros(len(rec_off) + 1, dtype=np.uint32)
        self.L.gzo_bam_to_sam.restype = ctypes.c_int64
        n = self.L.gzo_bam_to_sam(bam, rec_off.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(len(rec_off)), names, noff.ctypes.data_as(ctypes.c_void_p), len(ref_names),
                                  text.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(cap), lo.ctypes.data_as(ctypes.c_void_p))
        if n < 0:
            raise ValueError(-1 - n)
        return text[:n].tobytes(), lo

    # ---- N1 (first part): lines, FASTQ records, tokens
    def text_lines(self, text):
        import numpy as np
        text = bytes(text)
        cap = text.count(b"\n") + 1
        off = np.zeros(cap, dtype=np.uint32); ln = np.zeros(cap, dtype=np.uint32)
        self.L.gzo_text_lines.restype = ctypes.c_uint64
        n = self.L.gzo_text_lines(text, ctypes.c_uint64(len(text)), off.ctypes.data_as(ctypes.c_void_p), ln.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(cap))
        return off[:n].copy(), ln[:n].copy()

    def fastq_records(self, text, line_off, line_len):
        """-> (rc, [(off, len)] for line 1, SEQ, line 3, QUAL)"""
        import numpy as np
        text = bytes(text)
        lo = np.ascontiguousarray(line_off, dtype=np.uint32); ll = np.ascontiguousarray(line_len, dtype=np.uint32)
        nr = len(lo) // 4
        cols = [np.zeros(max(1, nr), dtype=np.uint32) for _ in range(8)]
        self.L.gzo_fastq_records.restype = ctypes.c_long
        rc = self.L.gzo_fastq_records(text, lo.ctypes.data_as(ctypes.c_void_p), ll.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(len(lo)),
                                      *[c.ctypes.data_as(ctypes.c_void_p) for c in cols])
        return rc, [(cols[2 * i][:nr], cols[2 * i + 1][:nr]) for i in range(4)]

    def tokenize_column(self, text, off, length, seps):
        """-> (n_bad, item_off[n_items, n], item_len[n_items, n])"""
        import numpy as np
        text = bytes(text); seps = bytes(seps)
        off = np.ascontiguousarray(off, dtype=np.uint32); length = np.ascontiguousarray(length, dtype=np.uint32)
        n, ni = len(off), len(seps) + 1
        io = np.zeros((ni, max(1, n)), dtype=np.uint32); il = np.zeros((ni, max(1, n)), dtype=np.uint32)
        if n == 0:
            return 0, io[:, :0], il[:, :0]
        self.L.gzo_tokenize_column.restype = ctypes.c_uint64
        nb = self.L.gzo_tokenize_column(text, off.ctypes.data_as(ctypes.c_void_p), length.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(n),
                                        seps, len(seps), io.ctypes.data_as(ctypes.c_void_p), il.ctypes.data_as(ctypes.c_void_p))
        return int(nb), io, il

    def seg_integer_or_not(self, text, off, length, nothing_char=0, lookup_off=0):
        """-> (snip_off, snip_len, values, is_nothing)"""
        import numpy as np
        text = bytes(text)
        off = np.ascontiguousarray(off, dtype=np.uint32); length = np.ascontiguousarray(length, dtype=np.uint32)
        n = len(off)
        so = np.zeros(max(1, n), dtype=np.uint32); sl = np.zeros(max(1, n), dtype=np.uint32)
        v = np.zeros(max(1, n), dtype=np.int64); m = np.zeros(max(1, n), dtype=np.uint8)
        self.L.gzo_seg_integer_or_not.restype = ctypes.c_uint64
        nv = self.L.gzo_seg_integer_or_not(text, off.ctypes.data_as(ctypes.c_void_p), length.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(n),
                                           int(nothing_char), ctypes.c_uint32(lookup_off), so.ctypes.data_as(ctypes.c_void_p),
                                           sl.ctypes.data_as(ctypes.c_void_p), v.ctypes.data_as(ctypes.c_void_p), m.ctypes.data_as(ctypes.c_void_p))
        return so[:n].copy(), sl[:n].copy(), v[:nv].copy(), m[:nv].copy()

    def transpose_partial(self, data, rows, cols, w, missing, to_file=True):
        import numpy as np
        data = bytes(data); missing = np.ascontiguousarray(missing, dtype=np.uint8)
        n = len(data) // w
        out = ctypes.create_string_buffer(max(1, len(data)))
        self.L.gzo_transpose_partial.restype = ctypes.c_long
        rc = self.L.gzo_transpose_partial(data, ctypes.c_uint64(n), rows, cols, w, missing.ctypes.data_as(ctypes.c_void_p), out, int(to_file))
        if rc != n:
            raise RuntimeError("oracle transpose_partial: the mask does not leave %d elements" % n)
        return out.raw[:len(data)]

    # ---- sections
    def adler32(self, data, start=1):
        data = bytes(data)
        return self.L.gzo_adler32(start, data, len(data))

    def assign_sort(self, tests, mode=0):
        """gzo_assign_sort: [(codec, size, clock)] in trial order -> (winner, sorted)"""
        n = len(tests)
        tab = (GzoCodecTest * max(1, n))(*[GzoCodecTest(int(c), float(sz), float(ck)) for c, sz, ck in tests])
        w = self.L.gzo_assign_sort(tab, n, mode)
        if w < 0:
            raise RuntimeError("oracle assign_sort failed")
        return w, [(t.codec, t.size, t.clock) for t in tab[:n]]

    def assign_best_with(self, data, rows=(), clock_ns_per_byte=None, mode=0):
        """codec_assign_best_codec over the nine restated candidates + rows [(codec, PAYLOAD size, clock_us)] of the host's (a8 in full)"""
        c0, sizes = self.assign_best(data)
        if not c0:
            return 0
        cand = (1, 6, 7, 8, 9, 16, 17, 18, 19)
        sample = min(len(data), 99999)
        tests = [(cd, sizes[i], (sample * clock_ns_per_byte[cd] / 1000.0) if (i and clock_ns_per_byte is not None) else 0.0) for i, cd in enumerate(cand)]
        tests += [(c, sz + 28, ck) for c, sz, ck in rows]
        return self.assign_sort(tests, mode)[0]

    def section_frame(self, desc, payload, raw_len):
        payload = bytes(payload)
        z = ctypes.create_string_buffer(40 + len(payload) + 8)
        n = self.L.gzo_section_frame(ctypes.byref(desc), payload, len(payload), raw_len, z, ctypes.c_uint64(len(z)))
        if n < 0:
            raise RuntimeError("oracle section_frame failed")
        return z.raw[:n]

    def section_compress(self, desc, data):
        data = bytes(data)
        cap = 40 + self.est_size((desc.sub_codec or 1) if desc.codec in (13, 11) else desc.codec, len(data)) + 64 + len(data)
        z = ctypes.create_string_buffer(cap)
        n = self.L.gzo_section_compress(ctypes.byref(desc), data, len(data), z, ctypes.c_uint64(cap))
        if n < 0:
            raise RuntimeError("oracle section_compress failed")
        return z.raw[:n]


class GzoCodecTest(ctypes.Structure):
    _fields_ = [("codec", ctypes.c_int32), ("size", ctypes.c_float), ("clock", ctypes.c_float)]


class GzoDomq(ctypes.Structure):
    _fields_ = [("qual", ctypes.c_void_p), ("runs", ctypes.c_void_p), ("mplx", ctypes.c_void_p), ("divr", ctypes.c_void_p),
                ("qual_len", ctypes.c_uint64), ("runs_len", ctypes.c_uint64), ("mplx_len", ctypes.c_uint64), ("divr_len", ctypes.c_uint64),
                ("denorm", ctypes.c_uint8 * (95 * 95)), ("num_doms", ctypes.c_uint32), ("num_norm_qs", ctypes.c_uint32),
                ("has_diverse", ctypes.c_int), ("all_diverse", ctypes.c_int)]


def oracle_domq(O, text, off, length):
    """N3 through the oracle (gzo_domq_encode / gzo_domq_is_fit) -> same dict as Engine.domq_columns gives per column"""
    import numpy as np
    text = bytes(text) + b"\0"
    off = np.ascontiguousarray(off, dtype=np.uint32); length = np.ascontiguousarray(length, dtype=np.uint32)
    o = GzoDomq()
    O.L.gzo_domq_encode.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
    O.L.gzo_domq_is_fit.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
    O.L.gzo_domq_free.argtypes = [ctypes.c_void_p]
    if O.L.gzo_domq_encode(text, off.ctypes.data, length.ctypes.data, len(off), ctypes.byref(o)) != 0:
        raise RuntimeError("oracle domq: a score outside ' '..'~'")
    r = dict(qual=ctypes.string_at(o.qual, o.qual_len), runs=ctypes.string_at(o.runs, o.runs_len), mplx=ctypes.string_at(o.mplx, o.mplx_len),
             divr=ctypes.string_at(o.divr, o.divr_len), denorm=bytes(o.denorm[:o.num_doms * o.num_norm_qs]), num_doms=o.num_doms, num_norm_qs=o.num_norm_qs,
             has_diverse=bool(o.has_diverse), all_diverse=bool(o.all_diverse),
             fit=bool(O.L.gzo_domq_is_fit(text, off.ctypes.data, length.ctypes.data, len(off))))
    O.L.gzo_domq_free(ctypes.byref(o))
    return r


class GzoMerge(ctypes.Structure):
    _fields_ = [("vblock_i", ctypes.c_uint32), ("n_ol", ctypes.c_uint32), ("n_new", ctypes.c_uint32),
                ("dict", ctypes.c_void_p), ("node_char_index", ctypes.c_void_p), ("node_snip_len", ctypes.c_void_p), ("counts", ctypes.c_void_p),
                ("can_have_singletons", ctypes.c_uint8), ("flags", ctypes.c_uint8), ("no_drop_b250", ctypes.c_uint8), ("pair2_identical", ctypes.c_uint8),
                ("b250_len", ctypes.c_uint64), ("local_len", ctypes.c_uint64), ("b250_r1_len", ctypes.c_uint64), ("local_r1_len", ctypes.c_uint64),
                ("ats_node_index", ctypes.c_int32), ("dropped_b250", ctypes.c_uint8),
                ("node2word", ctypes.c_void_p), ("ston_local", ctypes.c_void_p), ("ston_len", ctypes.c_uint64), ("n_stons", ctypes.c_uint32)]


class OracleZctx:
    """row a4 the reference's way (gz_oracle.c): one file-level context; merge(col, ...) = ctx_merge_in_one_vctx of one
    VBlock context given as the dict ctx_seg_column returns. Same call shape as genozip_amd.codec.Zctx."""

    def __init__(self, oracle, estimated_entries=0):
        self.L = oracle.L
        self.L.gzo_zctx_create.restype = ctypes.c_void_p
        self.L.gzo_zctx_destroy.argtypes = [ctypes.c_void_p]
        self.L.gzo_ctx_merge.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.L.gzo_zctx_view.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 6
        self.z = self.L.gzo_zctx_create(ctypes.c_uint32(estimated_entries))

    def __del__(self):
        if getattr(self, "z", None):
            self.L.gzo_zctx_destroy(self.z)
            self.z = None

    def merge(self, vblock_i, n_ol, col, can_have_singletons=False, flags=0, local_len=0, no_drop_b250=False,
              pair2_identical=False, b250_r1_len=0, local_r1_len=0):
        """-> dict(node2word, ston_local, n_stons, dropped_b250)"""
        import numpy as np
        n_new = len(col["node_snip_len"])
        d = np.frombuffer(bytes(col["dict"]) + b"\0", dtype=np.uint8).copy()
        nci = np.ascontiguousarray(col["node_char_index"], dtype=np.uint64); nsl = np.ascontiguousarray(col["node_snip_len"], dtype=np.uint32)
        cnt = np.ascontiguousarray(col["counts"], dtype=np.uint32)
        n2w = np.zeros(max(1, n_new), dtype=np.int32); ston = np.zeros(len(d) + 8, dtype=np.uint8)
        ats = bool(col["all_the_same"])
        j = GzoMerge(vblock_i, n_ol, n_new, d.ctypes.data, nci.ctypes.data, nsl.ctypes.data, cnt.ctypes.data,
                     int(can_have_singletons and not ats), flags | (0x20 if ats else 0), int(no_drop_b250), int(pair2_identical),
                     len(col["b250"]), local_len, b250_r1_len, local_r1_len,
                     int(col["node_index"][0]) if ats and len(col["node_index"]) else -1, 0, n2w.ctypes.data, ston.ctypes.data, 0, 0)
        rc = self.L.gzo_ctx_merge(self.z, ctypes.byref(j))
        if rc != 0:
            raise RuntimeError("oracle ctx_merge failed")
        return dict(node2word=n2w[:n_new].copy(), ston_local=ston[:j.ston_len].tobytes(), n_stons=int(j.n_stons), dropped_b250=bool(j.dropped_b250))

    def view(self):
        """-> dict(dict bytes, n_words, counts, n_failed_singletons, rm_dict)"""
        import numpy as np
        dp, cp = ctypes.c_void_p(), ctypes.c_void_p()
        dl, nf = ctypes.c_uint64(), ctypes.c_uint64()
        nw, rm = ctypes.c_uint32(), ctypes.c_int()
        self.L.gzo_zctx_view(self.z, ctypes.byref(dp), ctypes.byref(dl), ctypes.byref(nw), ctypes.byref(cp), ctypes.byref(nf), ctypes.byref(rm))
        d = ctypes.string_at(dp.value, dl.value) if dl.value else b""
        c = np.frombuffer(ctypes.string_at(cp.value, 8 * nw.value), dtype=np.uint64).copy() if nw.value else np.zeros(0, np.uint64)
        return dict(dict=d, n_words=nw.value, counts=c, n_failed_singletons=nf.value, rm_dict=bool(rm.value))

    def words(self):
        d = self.view()["dict"]
        return d[:-1].split(b"\0") if d else []


class GzoPathPlan(ctypes.Structure):
    _fields_ = [("n_items", ctypes.c_uint32), ("item_kind", ctypes.c_uint8 * 16), ("seps", ctypes.c_uint8 * 32), ("sep_counts", ctypes.c_uint8 * 32),
                ("n_seps", ctypes.c_uint32), ("lcodec", ctypes.c_uint8 * 16), ("bcodec", ctypes.c_uint8 * 16), ("qual_codec", ctypes.c_uint8),
                ("aux_codec", ctypes.c_uint8 * 3), ("x_codec", ctypes.c_uint8), ("qual_bcodec", ctypes.c_uint8), ("domq", ctypes.c_uint8)]


def fastq_path_many(oracle, text, vbs, plan, codecs, domq, n_threads, replicas=1, ref=None):
    """bench.py's cpu_baseline.whole_path (oracle/gz_oracle_path.c): the whole path of every VBlock (text -> lines -> reads -> items -> seg
    columns -> merge -> b250 / local generation -> codecs -> framed sections) on a pthread pool, one VBlock per task, every VBlock
    `replicas` times. text: bytes / numpy uint8 (host); vbs: [(offset, length)]; plan: the dict of genozip_amd.fastq.illumina_plan;
    codecs: {("local" | "b250", tag): codec id} as the GPU run's file ended up with (missing: the VBlock runs the trial itself);
    ref: a pyoracle.Ref - the codec calls then go through the reference's own htscodecs. -> (seconds, z bytes per VBlock, stream bytes per VBlock)"""
    import numpy as np
    L = oracle.L
    P = GzoPathPlan()
    items = sorted([c for c in plan["ctxs"] if c["kind"] in (2, 3, 4)], key=lambda c: c["item"])       # GZ_FQ_ITEM_TEXT / _INT / _DELTA
    P.n_items = len(items)
    for i, c in enumerate(items):
        assert c["item"] == i
        P.item_kind[i] = {2: 0, 3: 1, 4: 2}[c["kind"]]
        P.lcodec[i] = codecs.get(("local", c["tag"]), 0); P.bcodec[i] = codecs.get(("b250", c["tag"]), 0)
    seps = bytes(plan["seps"])
    P.n_seps = len(seps)
    for i, b in enumerate(seps):
        P.seps[i] = b; P.sep_counts[i] = plan["sep_counts"][i]
    P.qual_codec = codecs.get(("local", "QUAL"), 0)
    for k, t in enumerate(("DOMQRUNS", "QUALMPLX", "DIVRQUAL")):
        P.aux_codec[k] = codecs.get(("local", t), 0)
    P.x_codec = codecs.get(("local", "NONREF_X"), 0) or 1
    P.qual_bcodec = codecs.get(("b250", "QUAL"), 0)
    P.domq = int(bool(domq))
    if ref is not None:
        L.gzo_path_use_codecs(ctypes.cast(ref.L.htsref_rans_compress, ctypes.c_void_p), ctypes.cast(ref.L.htsref_arith_compress, ctypes.c_void_p))
    else:
        L.gzo_path_use_codecs(None, None)
    t = np.ascontiguousarray(np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else text)
    n = len(vbs)
    off = np.array([v[0] for v in vbs], dtype=np.uint64); ln = np.array([v[1] for v in vbs], dtype=np.uint64)
    zl = (ctypes.c_long * n)(); st = (ctypes.c_uint64 * n)()
    L.gzo_fastq_path_many.restype = ctypes.c_double
    dt = L.gzo_fastq_path_many(t.ctypes.data_as(ctypes.c_void_p), off.ctypes.data_as(ctypes.c_void_p), ln.ctypes.data_as(ctypes.c_void_p), n, int(replicas),
                               ctypes.byref(P), zl, st, int(n_threads))
    if dt < 0:
        raise RuntimeError("oracle whole-path leg failed")
    return dt, list(zl), list(st)


class GzoTextPlan(ctypes.Structure):
    _fields_ = [("n_items", ctypes.c_uint32), ("item_kind", ctypes.c_uint8 * 64), ("seps", ctypes.c_uint8 * 64), ("sep_counts", ctypes.c_uint8 * 64), ("n_seps", ctypes.c_uint32),
                ("lcodec", ctypes.c_uint8 * 64), ("bcodec", ctypes.c_uint8 * 64), ("qual_codec", ctypes.c_uint8), ("aux_codec", ctypes.c_uint8 * 3), ("x_codec", ctypes.c_uint8),
                ("domq", ctypes.c_uint8), ("seq_pad", ctypes.c_uint8), ("reserved", ctypes.c_uint8), ("n_samples", ctypes.c_uint32), ("n_sub", ctypes.c_uint32),
                ("sub_kind", ctypes.c_uint8 * 8), ("sub_lcodec", ctypes.c_uint8 * 8), ("sub_bcodec", ctypes.c_uint8 * 8)]


def text_path_many(oracle, text, vbs, plan, codecs, domq, n_threads, replicas=1, ref=None):
    """fastq_path_many for the one-line-record plans (genozip_amd/sam.py, vcf.py: oracle/gz_oracle_path.c::gzo_text_vb_path): the whole path of every VBlock
    of SAM / VCF text on a pthread pool. Same arguments and result."""
    import numpy as np
    L = oracle.L
    P = GzoTextPlan()
    seps = bytes(plan["seps"])
    P.n_seps = len(seps); P.n_items = len(seps) + 1
    for i, b in enumerate(seps):
        P.seps[i] = b; P.sep_counts[i] = plan["sep_counts"][i]
    for i in range(P.n_items):
        P.item_kind[i] = 5
    ns = plan.get("n_samples", 0)
    for c in plan["ctxs"]:
        if c.get("per_sample"):
            j = c["item"]
            P.sub_kind[j] = 1 if c["kind"] == 3 else 0
            P.sub_lcodec[j] = codecs.get(("local", c["tag"]), 0); P.sub_bcodec[j] = codecs.get(("b250", c["tag"]), 0)
        elif c["kind"] in (2, 3, 4):                               # GZ_FQ_ITEM_TEXT / _INT / _DELTA
            i = c["item"]
            P.item_kind[i] = {2: 0, 3: 1, 4: 2}[c["kind"]]
            P.lcodec[i] = codecs.get(("local", c["tag"]), 0); P.bcodec[i] = codecs.get(("b250", c["tag"]), 0)
    if ns:
        P.n_samples, P.n_sub = ns, plan["n_subfields"]
        P.item_kind[P.n_items - 1] = 6
    else:
        P.item_kind[plan["seq_item"]] = 3; P.item_kind[plan["qual_item"]] = 4
        P.seq_pad = plan.get("seq_pad", 0)
    P.qual_codec = codecs.get(("local", "QUAL"), 0)
    for k, t in enumerate(("DOMQRUNS", "QUALMPLX", "DIVRQUAL")):
        P.aux_codec[k] = codecs.get(("local", t), 0)
    P.x_codec = codecs.get(("local", "NONREF_X"), 0) or 1
    P.domq = int(bool(domq))
    if ref is not None:
        L.gzo_path_use_codecs(ctypes.cast(ref.L.htsref_rans_compress, ctypes.c_void_p), ctypes.cast(ref.L.htsref_arith_compress, ctypes.c_void_p))
    else:
        L.gzo_path_use_codecs(None, None)
    t = np.ascontiguousarray(np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else text)
    n = len(vbs)
    off = np.array([v[0] for v in vbs], dtype=np.uint64); ln = np.array([v[1] for v in vbs], dtype=np.uint64)
    zl = (ctypes.c_long * n)(); st = (ctypes.c_uint64 * n)()
    L.gzo_text_path_many.restype = ctypes.c_double
    dt = L.gzo_text_path_many(t.ctypes.data_as(ctypes.c_void_p), off.ctypes.data_as(ctypes.c_void_p), ln.ctypes.data_as(ctypes.c_void_p), n, int(replicas),
                              ctypes.byref(P), zl, st, int(n_threads))
    if dt < 0:
        raise RuntimeError("oracle whole-path leg (text plans) failed")
    return dt, list(zl), list(st)


class Ref:
    """the reference's vendored htscodecs, compiled in place (only where oracle/_ref was built)"""

    def __init__(self, path=REF_SO):
        self.L = ctypes.CDLL(path)
        for n in ("htsref_rans_compress", "htsref_arith_compress", "htsref_rans_uncompress", "htsref_arith_uncompress"):
            getattr(self.L, n).restype = ctypes.c_long
        self.L.htsref_rans_bound.restype = ctypes.c_uint32
        self.L.htsref_arith_bound.restype = ctypes.c_uint32

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def hts_compress(self, kind, data, order, extra_cap=1024):
        data = bytes(data)
        bound = (self.L.htsref_rans_bound if kind == "rans" else self.L.htsref_arith_bound)(len(data), order) + extra_cap
        out = ctypes.create_string_buffer(bound)
        f = self.L.htsref_rans_compress if kind == "rans" else self.L.htsref_arith_compress
        n = f(data, len(data), out, bound, order)
        if n < 0:
            raise RuntimeError("ref %s compress failed" % kind)
        return out.raw[:n]

    def hts_uncompress(self, kind, comp, out_len):
        comp = bytes(comp)
        out = ctypes.create_string_buffer(max(1, out_len))
        f = self.L.htsref_rans_uncompress if kind == "rans" else self.L.htsref_arith_uncompress
        n = f(comp, len(comp), out, out_len)
        if n != out_len:
            raise RuntimeError("ref %s uncompress failed (%d)" % (kind, n))
        return out.raw[:out_len]

    def codec_compress(self, codec, data):
        kind = "rans" if codec < 16 else "arith"
        return self.hts_compress(kind, data, CODEC_ORDER[codec])

    def codec_compress_many(self, codecs, datas, n_threads, replicas=1):
        """C pthread pool over independent codec calls (bench.py's multithreaded CPU baseline), every call `replicas` times; returns
        (payloads, seconds)"""
        import time
        n = len(datas)
        ins = [bytes(d) for d in datas]
        caps = [(self.L.htsref_rans_bound if c < 16 else self.L.htsref_arith_bound)(len(d), CODEC_ORDER[c]) + 1024 for c, d in zip(codecs, ins)]
        outs = [ctypes.create_string_buffer(c) for c in caps]
        a_ar = (ctypes.c_int * n)(*[int(c >= 16) for c in codecs])
        a_or = (ctypes.c_int * n)(*[CODEC_ORDER[c] for c in codecs])
        a_in = (ctypes.c_char_p * n)(*ins)
        a_il = (ctypes.c_uint * n)(*[len(d) for d in ins])
        a_out = (ctypes.c_void_p * n)(*[ctypes.addressof(o) for o in outs])
        a_cap = (ctypes.c_uint * n)(*caps)
        a_ol = (ctypes.c_long * n)()
        t0 = time.perf_counter()
        rc = self.L.htsref_compress_many_rep(n, a_ar, a_or, a_in, a_il, a_out, a_cap, a_ol, n_threads, int(replicas))
        dt = time.perf_counter() - t0
        if rc != 0:
            raise RuntimeError("ref compress_many failed")
        return [o.raw[:l] for o, l in zip(outs, a_ol)], dt


class CtxRef:
    """the reference's OWN src/b250.c and src/dyn_int.c, compiled in place (oracle/Makefile target `ref`, oracle/ref_ctx_shim.c):
    rows a2 / a5 / a3 / a7 of SURVEY 8(a) as the reference computes them. Exists only where /root/reference does."""

    def __init__(self, path=CTXREF_SO):
        L = self.L = ctypes.CDLL(path)
        vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
        L.ctxref_b250_seg.restype = ctypes.c_long
        L.ctxref_b250_seg.argtypes = [vp, u32, u32, vp, ctypes.POINTER(u64), ctypes.POINTER(ctypes.c_int)]
        L.ctxref_b250_generate.restype = ctypes.c_long
        L.ctxref_b250_generate.argtypes = [ctypes.c_char_p, u32, u64, ctypes.c_int, u32, vp, u32, vp]
        L.ctxref_dyn_int_column.argtypes = [vp, vp, u64, ctypes.c_int, vp, ctypes.POINTER(u64)]
        L.ctxref_dyn_int_transpose.argtypes = [ctypes.c_int, ctypes.c_char_p, u64, u32, vp]
        L.ctxref_hash_do.restype = u32
        L.ctxref_hash_do.argtypes = [u32, ctypes.c_char_p, u32]
        L.ctxref_local_to_file_order.argtypes = [ctypes.c_int, vp, u64]
        L.ctxref_local_to_native.argtypes = [ctypes.c_int, vp, u64, u32]

    @staticmethod
    def available():
        return os.path.exists(CTXREF_SO)

    def b250_seg(self, node_indices, ol_nodes_len):
        """-> (seg-format bytes, count, all_the_same) after b250_seg_append of every node index"""
        import numpy as np
        ni = np.ascontiguousarray(node_indices, dtype=np.int32)
        out = np.zeros(4 * len(ni) + 16, dtype=np.uint8)
        cnt, ats = ctypes.c_uint64(), ctypes.c_int()
        n = self.L.ctxref_b250_seg(ni.ctypes.data, len(ni), ol_nodes_len, out.ctypes.data, ctypes.byref(cnt), ctypes.byref(ats))
        return out[:n].tobytes(), cnt.value, bool(ats.value)

    def b250_generate(self, seg, count, all_the_same, ol_nodes_len, node2word):
        import numpy as np
        n2w = np.ascontiguousarray(list(node2word) or [0], dtype=np.int32)
        out = np.zeros(len(seg) + 16, dtype=np.uint8)
        n = self.L.ctxref_b250_generate(bytes(seg), len(seg), count, int(all_the_same), ol_nodes_len, n2w.ctypes.data, len(node2word), out.ctypes.data)
        return out[:n].tobytes()

    def dyn_int_column(self, values, is_nothing=None, nothing_char=0):
        import numpy as np
        v = np.ascontiguousarray(values, dtype=np.int64)
        m = None if is_nothing is None else np.ascontiguousarray(is_nothing, dtype=np.uint8)
        out = np.zeros(8 * len(v) + 16, dtype=np.uint8)
        ln = ctypes.c_uint64()
        lt = self.L.ctxref_dyn_int_column(v.ctypes.data, m.ctypes.data if m is not None else None, len(v), nothing_char, out.ctypes.data, ctypes.byref(ln))
        return lt, out[:ln.value].tobytes()

    def dyn_int_transpose(self, ltype, data_file_order, n_elems, cols):
        import numpy as np
        out = np.zeros(len(data_file_order) + 16, dtype=np.uint8)
        lt = self.L.ctxref_dyn_int_transpose(ltype, bytes(data_file_order), n_elems, cols, out.ctypes.data)
        return lt, out[:len(data_file_order)].tobytes()

    def local_to_file_order(self, ltype, raw_native_le, width):
        """zip_generate_local's byte-order step with the reference's own converters (src/buffer.c:336-350)"""
        import numpy as np
        a = np.frombuffer(bytes(raw_native_le), dtype=np.uint8).copy()
        self.L.ctxref_local_to_file_order(ltype, a.ctypes.data, len(a) // width)
        return a.tobytes()

    def local_to_native(self, ltype, file_bytes, width, cols=0):
        """lt_desc[ltype].file_to_native (src/local_type.h:75-108) -> (resulting ltype, native bytes)"""
        import numpy as np
        a = np.frombuffer(bytes(file_bytes), dtype=np.uint8).copy()
        lt = self.L.ctxref_local_to_native(ltype, a.ctypes.data, len(a) // width, cols)
        return lt, a.tobytes()

    def domq(self, text, off, length):
        """the reference's own codec_domq_comp_init + codec_domq_compress on the QUAL lines of a VBlock (sub-codec = store)
        -> dict(qual, runs, mplx, divr, denorm_snip (base64 as segged into DOMQRUNS), param, sub_codec, fit)"""
        import numpy as np
        text = bytes(text) + b"\0"
        off = np.ascontiguousarray(off, dtype=np.uint32); length = np.ascontiguousarray(length, dtype=np.uint32)
        B = int(length.astype(np.uint64).sum())
        bufs = [np.zeros(2 * B + 2048, dtype=np.uint8) for _ in range(4)]
        lens = [ctypes.c_uint64() for _ in range(4)]
        snip = np.zeros(16384, dtype=np.uint8)
        sl, param, sub, fit = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_int()
        self.L.ctxref_domq.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32] + [ctypes.c_void_p] * 13
        args = []
        for b, l in zip(bufs, lens):
            args += [b.ctypes.data, ctypes.addressof(l)]
        self.L.ctxref_domq(text, len(text), off.ctypes.data, length.ctypes.data, len(off), *args, snip.ctypes.data, ctypes.addressof(sl),
                           ctypes.addressof(param), ctypes.addressof(sub), ctypes.addressof(fit))
        q, r, m, d = [b[:l.value].tobytes() for b, l in zip(bufs, lens)]
        return dict(qual=q, runs=r, mplx=m, divr=d, denorm_snip=snip[:sl.value].tobytes(), param=param.value, sub_codec=sub.value, fit=bool(fit.value))

    def hash_do(self, hash_len, snip):
        return self.L.ctxref_hash_do(hash_len, bytes(snip), len(snip))

    def merge_hash(self, estimated_entries, vbs):
        """the reference's own hash.c under the merge loop: vbs = [(can_have_singletons, [(snip, count), ...new nodes])]
        -> dict(word=[per VBlock list], ston=[per VBlock list], dict, n_failed, hash_len)"""
        import numpy as np
        flat = [(s, c) for _, nodes in vbs for s, c in nodes]
        n = len(flat)
        snips = b"".join(bytes(s) + b"\0" for s, _ in flat) + b"\0"
        sl = np.array([len(s) for s, _ in flat], dtype=np.uint32); cnt = np.array([c for _, c in flat], dtype=np.uint32)
        nn = np.array([len(nodes) for _, nodes in vbs], dtype=np.uint32); cs = np.array([int(c) for c, _ in vbs], dtype=np.uint8)
        word = np.zeros(max(1, n), dtype=np.int32); ston = np.zeros(max(1, n), dtype=np.uint8)
        dic = np.zeros(len(snips) + 64, dtype=np.uint8)
        dl, nf, hl = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint32()
        self.L.ctxref_merge_hash.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p] + [ctypes.c_void_p] * 8
        self.L.ctxref_merge_hash(estimated_entries, len(vbs), nn.ctypes.data, cs.ctypes.data, snips, sl.ctypes.data, cnt.ctypes.data, word.ctypes.data, ston.ctypes.data,
                                 dic.ctypes.data, ctypes.addressof(dl), ctypes.addressof(nf), ctypes.addressof(hl))
        out_w, out_s, at = [], [], 0
        for _, nodes in vbs:
            out_w.append([int(x) for x in word[at:at + len(nodes)]]); out_s.append([int(x) for x in ston[at:at + len(nodes)]]); at += len(nodes)
        return dict(word=out_w, ston=out_s, dict=dic[:dl.value].tobytes(), n_failed=nf.value, hash_len=hl.value)

    def seg_nodes(self, ol_words, snips):
        """the reference's own hash_get_entry_for_seg under ctx_create_node_do's add-a-node lines -> node index of every snip"""
        import numpy as np
        ol = b"".join(bytes(w) + b"\0" for w in ol_words) + b"\0"; sn = b"".join(bytes(w) + b"\0" for w in snips) + b"\0"
        oll = np.array([len(w) for w in ol_words], dtype=np.uint32); snl = np.array([len(w) for w in snips], dtype=np.uint32)
        out = np.zeros(max(1, len(snips)), dtype=np.int32)
        self.L.ctxref_seg_nodes.argtypes = [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p]
        assert self.L.ctxref_seg_nodes(len(ol_words), ol, oll.ctypes.data, len(snips), sn, snl.ctypes.data, out.ctypes.data) == 0
        return [int(x) for x in out[:len(snips)]]

    def str_get_int(self, snips):
        """the reference's own str_get_int -> ([is_int], [value])"""
        import numpy as np
        sn = b"".join(bytes(w) + b"\0" for w in snips) + b"\0"
        snl = np.array([len(w) for w in snips], dtype=np.uint32)
        f = np.zeros(max(1, len(snips)), dtype=np.uint8); v = np.zeros(max(1, len(snips)), dtype=np.int64)
        self.L.ctxref_str_get_int.argtypes = [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        self.L.ctxref_str_get_int(len(snips), sn, snl.ctypes.data, f.ctypes.data, v.ctypes.data)
        return [int(x) for x in f[:len(snips)]], [int(x) for x in v[:len(snips)]]

    def acgt(self, seq):
        """the reference's own codec_acgt_compress on a contiguous NONREF.local (sub-codec = store)
        -> (packed as handed to the sub-codec, NONREF_X.local, has_x, sub_codec)"""
        import numpy as np
        seq = bytes(seq); n = len(seq)
        packed = np.zeros(n // 4 + 64, dtype=np.uint8); x = np.zeros(n + 16, dtype=np.uint8)
        pl, sub, hx = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_int()
        self.L.ctxref_acgt.argtypes = [ctypes.c_char_p, ctypes.c_uint32] + [ctypes.c_void_p] * 5
        self.L.ctxref_acgt(seq, n, packed.ctypes.data, ctypes.addressof(pl), x.ctypes.data, ctypes.addressof(hx), ctypes.addressof(sub))
        return packed[:pl.value].tobytes(), x[:n].tobytes(), bool(hx.value), sub.value


class CompRef:
    """the reference's OWN src/compressor.c (comp_compress, row a10) with its codec_none.c, the vendored libdeflate adler32 and the
    vendored htscodecs, compiled in place (oracle/Makefile target `ref`, oracle/ref_comp_shim.c). Exists only where /root/reference does."""

    def __init__(self, path=COMPREF_SO):
        self.L = ctypes.CDLL(path)
        self.L.compref_section.restype = ctypes.c_long
        self.L.compref_section.argtypes = [ctypes.c_int] * 7 + [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64]

    @staticmethod
    def available():
        return os.path.exists(COMPREF_SO)

    def section(self, desc, data):
        """desc: GzoCtxSectionDesc -> header + payload as comp_compress appends them to z_data"""
        data = bytes(data)
        cap = 2 * len(data) + 300000
        out = ctypes.create_string_buffer(cap)
        n = self.L.compref_section(desc.section_type, desc.codec, desc.sub_codec, desc.flags, desc.ltype, desc.param, desc.b250_size_or_nothing_char,
                                   bytes(desc.dict_id), desc.vblock_i, data, len(data), out, cap)
        if n < 0:
            raise RuntimeError("compref_section failed")
        return out.raw[:n]


class AssignRef:
    """the reference's OWN src/codec.c - codec_assign_sorter under the C library's qsort, and codec_assign_best_codec with its own
    compressor.c / zfile.c / codec_htscodecs.c / htscodecs underneath - compiled in place (oracle/Makefile target `ref`,
    oracle/ref_assign_shim.c: the clock and the sizes of BZ2 / BSC / LZMA are scripted). Row a8. Exists only where /root/reference does."""
    NAMES = {"NONE": 1, "BZ2": 3, "LZMA": 4, "BSC": 5, "RANB": 6, "RANW": 7, "RANb": 8, "RANw": 9, "ARTB": 16, "ARTW": 17, "ARTb": 18, "ARTw": 19}

    def __init__(self, path=ASSIGNREF_SO):
        self.L = ctypes.CDLL(path)
        self.L.assignref_sort.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        self.L.assignref_run.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p]

    @staticmethod
    def available():
        return os.path.exists(ASSIGNREF_SO)

    def sort(self, tests, mode=0):
        """[(codec, size, clock)] in trial order -> the same rows as qsort (.., codec_assign_sorter) leaves them (src/codec.c:334)"""
        import numpy as np
        c = np.array([t[0] for t in tests], dtype=np.int32); s = np.array([t[1] for t in tests], dtype=np.float32); k = np.array([t[2] for t in tests], dtype=np.float32)
        self.L.assignref_sort(c.ctypes.data, s.ctypes.data, k.ctypes.data, len(tests), mode)
        return [(int(a), float(b), float(d)) for a, b, d in zip(c, s, k)]

    def run(self, inp, dict_id, txt_len, vb_size, data, ticks):
        """codec_assign_best_codec (src/codec.c:234): inp = the 14 integers of assignref_run. -> (out[5], [(name, size, clock)] the four
        best rows --show-codec printed, or [] when no trial ran)"""
        import numpy as np
        import re
        i = np.array(inp, dtype=np.int32); t = np.array(ticks, dtype=np.int32); o = np.zeros(8, dtype=np.int32)
        shown = ctypes.create_string_buffer(512)
        self.L.assignref_run(i.ctypes.data, bytes(dict_id).ljust(8, b"\0"), txt_len, vb_size, bytes(data), len(data), t.ctypes.data, o.ctypes.data, shown)
        rows = [(self.NAMES[m.group(1)], int(m.group(2)), int(m.group(3))) for m in re.finditer(r"\[(\w+)\s+(\d+) B\s+(\d+) ", shown.value.decode("utf8", "replace"))]
        return [int(x) for x in o[:5]], rows


class OrderRef:
    """the reference's OWN src/zip.c - zip_compress_all_contexts_local / _b250 in the sequence of zip_compress_one_vb, one compute thread -
    compiled in place (oracle/Makefile target `ref`, oracle/ref_order_shim.c): the order in which a VBlock's context sections reach z_data.
    Row a15. Exists only where /root/reference does."""

    def __init__(self, path=ORDERREF_SO):
        self.L = ctypes.CDLL(path)
        self.L.orderref_run.argtypes = [ctypes.c_uint32] + [ctypes.c_void_p] * 5 + [ctypes.c_uint32, ctypes.c_void_p]

    @staticmethod
    def available():
        return os.path.exists(ORDERREF_SO)

    def order(self, ctxs, vblock_i):
        """ctxs = [(did_i, local_dep, has_local, ston_only, has_b250)] -> [(index into ctxs, 'L' | 'B')]"""
        import numpy as np
        n = len(ctxs)
        did = np.array([c[0] for c in ctxs], dtype=np.uint16); dep = np.array([c[1] for c in ctxs], dtype=np.uint8)
        hl = np.array([c[2] for c in ctxs], dtype=np.uint8); so = np.array([c[3] for c in ctxs], dtype=np.uint8); hb = np.array([c[4] for c in ctxs], dtype=np.uint8)
        out = np.zeros(2 * n + 2, dtype=np.uint32)
        k = self.L.orderref_run(n, did.ctypes.data, dep.ctypes.data, hl.ctypes.data, so.ctypes.data, hb.ctypes.data, vblock_i, out.ctypes.data)
        at = {int(d): i for i, d in enumerate(did)}
        return [(at[int(x) // 2], "B" if x & 1 else "L") for x in out[:k]]


class MergeRef:
    """the reference's OWN src/context.c - ctx_merge_in_one_vctx with ctx_commit_node, ctx_insert_to_dict and ctx_drop_all_the_same - over its
    own hash.c / seg.c / b250.c, compiled in place (oracle/Makefile target `ref`, oracle/ref_merge_shim.c). The loop of row a4. ONE file context
    at a time (the library keeps it). Same call shape as OracleZctx / genozip_amd.codec.Zctx. Exists only where /root/reference does."""

    def __init__(self, estimated_entries=0, path=MERGEREF_SO):
        self.L = ctypes.CDLL(path)
        self.L.mergeref_merge.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_uint64] + [ctypes.c_void_p] * 6
        self.L.mergeref_view.restype = ctypes.c_uint64
        self.L.mergeref_view.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        self.L.mergeref_open(ctypes.c_uint32(estimated_entries))

    @staticmethod
    def available():
        return os.path.exists(MERGEREF_SO)

    def merge(self, vblock_i, n_ol, col, can_have_singletons=False, flags=0, local_len=0, no_drop_b250=False,
              pair2_identical=False, b250_r1_len=0, local_r1_len=0, lcodec=0, bcodec=0):
        import numpy as np
        n_new = len(col["node_snip_len"])
        d = np.frombuffer(bytes(col["dict"]) + b"\0", dtype=np.uint8).copy()
        nci = np.ascontiguousarray(col["node_char_index"], dtype=np.uint64); nsl = np.ascontiguousarray(col["node_snip_len"], dtype=np.uint32)
        cnt = np.ascontiguousarray(list(col["counts"]) + [0], dtype=np.uint32)
        ats = bool(col["all_the_same"])
        inp = np.array([vblock_i, n_ol, n_new, int(can_have_singletons and not ats), flags | (0x20 if ats else 0), int(no_drop_b250), int(pair2_identical), lcodec, bcodec,
                        int(col["node_index"][0]) if ats and len(col["node_index"]) else -1], dtype=np.int32)
        lens = np.array([len(col["b250"]), local_len, b250_r1_len, local_r1_len], dtype=np.uint64)
        n2w = np.zeros(max(1, n_new), dtype=np.int32); ston = np.zeros(len(d) + 8, dtype=np.uint8); out = np.zeros(8, dtype=np.int32)
        self.L.mergeref_merge(inp.ctypes.data, lens.ctypes.data, d.ctypes.data, len(col["dict"]), nci.ctypes.data, nsl.ctypes.data, cnt.ctypes.data,
                              n2w.ctypes.data, ston.ctypes.data, out.ctypes.data)
        assert out[0] == 1
        text = ston[:out[4]].tobytes()
        return dict(node2word=n2w[:n_new].copy(), ston_local=text, n_stons=text.count(b"\0"), dropped_b250=bool(out[1]), lcodec=int(out[2]), bcodec=int(out[3]))

    def view(self):
        import numpy as np
        o = np.zeros(12, dtype=np.int64)
        n = self.L.mergeref_view(None, 0, None, 0, o.ctypes.data)
        d = np.zeros(max(1, n), dtype=np.uint8); c = np.zeros(max(1, int(o[0])), dtype=np.uint64)
        self.L.mergeref_view(d.ctypes.data, n, c.ctypes.data, int(o[0]), o.ctypes.data)
        return dict(dict=d[:n].tobytes(), n_words=int(o[0]), counts=c[:int(o[0])].copy(), n_failed_singletons=int(o[1]), rm_dict=bool(o[2] and not o[3]),
                    all_the_same_wi=int(o[5]) if o[4] else -1, hash_len=int(o[6]), flags=int(o[7]), lcodec=int(o[8]), bcodec=int(o[9]))

    def words(self):
        d = self.view()["dict"]
        return d[:-1].split(b"\0") if d else []

    def commit_codec(self, is_local, codec):
        self.L.mergeref_commit_codec(int(is_local), int(codec))


def vcf_sample_items(text, line_off, line_len, n_samples, n_sub):
    """N1 for VCF, restated: the split vcf_seg_samples does (src/vcf_samples.c:1601 with seg_get_next_item, src/seg.c:153-198) - a data
    line is 9 tab-separated fields + n_samples samples, a sample its FORMAT subfields ':' separated, trailing ones may be left out.
    -> (n_bad, item_off [n_sub][lines * samples], item_len, missing) like Engine.vcf_sample_columns"""
    import numpy as np
    text = bytes(text)
    n = len(line_off)
    k = n * n_samples
    io = np.zeros((n_sub, max(1, k)), dtype=np.uint32); il = np.zeros((n_sub, max(1, k)), dtype=np.uint32); mi = np.ones((n_sub, max(1, k)), dtype=np.uint8)
    bad = 0
    for l in range(n):
        a, b = int(line_off[l]), int(line_off[l]) + int(line_len[l])
        fields, at = [], a
        for part in text[a:b].split(b"\t"):
            fields.append((at, len(part))); at += len(part) + 1
        if len(fields) != 9 + n_samples:
            bad += 1
        for s in range(n_samples):
            if 9 + s >= len(fields):
                continue
            o, ln = fields[9 + s]
            if s == n_samples - 1 and len(fields) > 9 + n_samples:          # (too many fields: the last sample runs to the end of the line)
                ln = b - o
            at = o
            for j, sub in enumerate(text[o:o + ln].split(b":")):
                if j < n_sub:
                    io[j, l * n_samples + s], il[j, l * n_samples + s], mi[j, l * n_samples + s] = at, len(sub), 0
                elif j == n_sub:
                    bad += 1
                at += len(sub) + 1
    io[mi == 1] = 0
    return bad, io[:, :k], il[:, :k], mi[:, :k]
