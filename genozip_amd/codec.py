"""Host-side mirror of the reference's codec / context plugin surface for the hot path, on top of
the C-ABI in include/genozip_amd.h.

Reference interface                                         here
----------------------------------------------------------  -------------------------------------------
codec_args[c].est_size      (src/codec.h:40)                Engine.est_size(codec, n)
codec_args[c].compress      (src/codec.h:17-27)             Engine.compress(codec, data, capacity, soft_fail)
codec_args[c].uncompress    (src/codec.h:29-38)             Engine.uncompress(codec, compressed, uncompressed_len)
codec_assign_best_codec     (src/codec.c:234)               Engine.assign_best(data)
b250_zip_generate           (src/b250.c:202)                Engine.b250_generate(seg, ol_nodes_len, node2word)
zip_generate_local          (src/zip.c:167)                 Engine.local_generate(ltype, raw, transpose_cols)
zip_compress_all_contexts_* + comp_compress                 Engine.vb_compress(vblocks)
   (src/zip.c:247,291; src/compressor.c:18)

Error behaviour follows the reference: compress() returns None when the capacity is below est_size and soft_fail is
set (the caller grows and retries, src/compressor.c:89-110) and raises otherwise.
"""
import ctypes as C

from . import lib as _lib
from .lib import (GzStream, GzB250Job, GzSection, GzVBlock, GZ_OK, GZ_TOO_SMALL)  # noqa: F401


class GenozipAMDError(RuntimeError):
    pass


class Section:
    """one b250/local section of a VBlock (the caller-decided fields of SectionHeaderCtx, src/sections.h:419-435)"""

    def __init__(self, data, section_type, codec, dict_id, ltype=0, flags=0, param=0, byte30=0, sub_codec=0,
                 data_len=None, data_len_dev=None):
        self.data, self.section_type, self.codec = data, section_type, codec
        self.dict_id = (bytes(dict_id) + b"\0" * 8)[:8]
        self.ltype, self.flags, self.param, self.byte30, self.sub_codec = ltype, flags, param, byte30, sub_codec
        self.data_len, self.data_len_dev = data_len, data_len_dev


class VBlock:
    def __init__(self, vblock_i, sections, recon_size=0, longest_line_len=0, longest_seq_len=0, digest=b"\0" * 16, vb_flags=0):
        self.vblock_i, self.sections = vblock_i, sections
        self.recon_size, self.longest_line_len, self.longest_seq_len = recon_size, longest_line_len, longest_seq_len
        self.digest, self.vb_flags = (bytes(digest) + b"\0" * 16)[:16], vb_flags
        self.z = None       # device buffer after vb_compress
        self.z_len = 0


class Zctx:
    """a file-level context (zctx): the host side of the dictionary merge, row a4 (ctx_merge_in_one_vctx,
    src/context.c:938-1079). merge() takes one VBlock context as ctx_seg_columns returns it."""

    def __init__(self, L, estimated_entries=0):
        self.L = L
        self.z = L.gz_zctx_create(estimated_entries)

    def close(self):
        if getattr(self, "z", None):
            self.L.gz_zctx_destroy(self.z)
            self.z = None

    __del__ = close

    def merge(self, vblock_i, n_ol, col, can_have_singletons=False, flags=0, local_len=0, no_drop_b250=False,
              pair2_identical=False, b250_r1_len=0, local_r1_len=0):
        """-> dict(node2word, ston_local, n_stons, dropped_b250)"""
        import numpy as np
        from .lib import GzMergeJob
        n_new = len(col["node_snip_len"])
        d = np.frombuffer(bytes(col["dict"]) + b"\0", dtype=np.uint8).copy()
        nci = np.ascontiguousarray(col["node_char_index"], dtype=np.uint64); nsl = np.ascontiguousarray(col["node_snip_len"], dtype=np.uint32)
        cnt = np.ascontiguousarray(col["counts"], dtype=np.uint32)
        n2w = np.zeros(max(1, n_new), dtype=np.int32); ston = np.zeros(len(d) + 8, dtype=np.uint8)
        ats = bool(col["all_the_same"])
        j = GzMergeJob()
        j.vblock_i, j.n_ol, j.n_new = vblock_i, n_ol, n_new
        j.dict, j.node_char_index, j.node_snip_len, j.counts = d.ctypes.data, nci.ctypes.data, nsl.ctypes.data, cnt.ctypes.data
        j.can_have_singletons = int(can_have_singletons and not ats)
        j.flags = flags | (0x20 if ats else 0)
        j.no_drop_b250, j.pair2_identical = int(no_drop_b250), int(pair2_identical)
        j.b250_len, j.local_len, j.b250_r1_len, j.local_r1_len = len(col["b250"]), local_len, b250_r1_len, local_r1_len
        j.ats_node_index = int(col["node_index"][0]) if ats and len(col["node_index"]) else -1
        j.node2word, j.ston_local, j.ston_cap = n2w.ctypes.data, ston.ctypes.data, len(ston)
        rc = self.L.gz_ctx_merge(self.z, C.byref(j))
        if rc != GZ_OK:
            raise GenozipAMDError("gz_ctx_merge failed (%d)" % rc)
        return dict(node2word=n2w[:n_new].copy(), ston_local=ston[:j.ston_len].tobytes(), n_stons=int(j.n_stons), dropped_b250=bool(j.dropped_b250))

    def view(self):
        import numpy as np
        from .lib import GzZctxView
        v = GzZctxView()
        self.L.gz_zctx_view(self.z, C.byref(v))
        d = C.string_at(v.dict, v.dict_len) if v.dict_len else b""
        c = np.frombuffer(C.string_at(v.counts, 8 * v.n_words), dtype=np.uint64).copy() if v.n_words else np.zeros(0, np.uint64)
        return dict(dict=d, n_words=v.n_words, counts=c, n_failed_singletons=v.n_failed_singletons,
                    rm_dict=bool(v.rm_dict_all_the_same), hash_len=v.hash_len)

    def words(self):
        d = self.view()["dict"]
        return d[:-1].split(b"\0") if d else []


class ZipFile:
    """z_file for the hot path: gz_zip_open / gz_fastq_zip_vblocks / gz_zip_close"""

    def __init__(self, E, plan):
        from .fastq import c_plan
        self.E, self.plan = E, plan
        self._cplan, self._keep = c_plan(plan)
        self.f = E.L.gz_zip_open(E.h, C.byref(self._cplan))
        if not self.f:
            raise GenozipAMDError("gz_zip_open failed: bad plan")
        import weakref
        E._zip_files.append(weakref.ref(self))          # (the handle must outlive the files opened on it)

    def close(self):
        if getattr(self, "f", None):
            if getattr(self.E, "h", None):
                self.E.L.gz_zip_close(self.f)
            self.f = None

    __del__ = close

    def set_host_codecs(self, trial=None, compress=None, clock_ns_per_byte=None, mode=0):
        """gz_zip_set_host_codecs: the host's candidates of codec_assign_best_codec (BZ2 3 / LZMA 4 / BSC 5).
        trial (dict_id bytes, is_local, sample bytes) -> [(codec, payload size, clock_us)]; compress (codec, data bytes) -> payload bytes"""
        from .lib import GzHostCodecs, GZ_HOST_TRIAL_FN, GZ_HOST_COMPRESS_FN
        if trial is None:
            self.E._check(self.E.L.gz_zip_set_host_codecs(self.f, None), "gz_zip_set_host_codecs")
            self._hostc = None
            return

        def c_trial(user, dict_id, is_local, sample, n, rows, max_rows):
            try:
                got = trial(bytes(dict_id[:8]), int(is_local), bytes(sample[:n]))[:max_rows]
                for i, (c, sz, ck) in enumerate(got):
                    rows[i].codec, rows[i].size, rows[i].clock_us = int(c), float(sz), float(ck)
                return len(got)
            except Exception:                                   # (an exception must not cross the C frames)
                import traceback
                traceback.print_exc()
                return -1

        def c_compress(user, codec, data, n, out, out_len):
            try:
                pay = compress(int(codec), bytes(data[:n]))
                if len(pay) > out_len[0]:
                    return 1
                C.memmove(out, pay, len(pay))
                out_len[0] = len(pay)
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 2

        hc = GzHostCodecs()
        hc.trial, hc.compress, hc.user, hc.mode = GZ_HOST_TRIAL_FN(c_trial), GZ_HOST_COMPRESS_FN(c_compress), None, mode
        clk = (C.c_float * 32)(*clock_ns_per_byte) if clock_ns_per_byte is not None else None
        hc.clock_ns_per_byte = C.cast(clk, C.POINTER(C.c_float)) if clk is not None else None
        self._hostc = (hc, clk)                                  # (the callbacks must outlive the file's use of them)
        self.E._check(self.E.L.gz_zip_set_host_codecs(self.f, C.byref(hc)), "gz_zip_set_host_codecs")

    def vb_table(self, vbs):
        """vbs: list of (text_off, text_len, vblock_i, r1 index or -1[, flags: GZ_VB_LAST_OF_FILE = 1])"""
        from .lib import GzFastqVB
        tab = (GzFastqVB * len(vbs))()
        for i, t in enumerate(vbs):
            tab[i].text_off, tab[i].text_len, tab[i].vblock_i, tab[i].r1 = t[:4]
            tab[i].flags = t[4] if len(t) > 4 else 0
        return tab

    def zip_table(self, text_buf, text_len, tab, n):
        rc = self.E.L.gz_fastq_zip_vblocks(self.f, self.E.mem.ptr(text_buf), text_len, tab, n)
        if rc != GZ_OK:
            raise GenozipAMDError("gz_fastq_zip_vblocks failed (%d): %s" % (rc, self.E.L.gz_last_error(self.E.h).decode()))

    def begin(self, text_buf, text_len, tab, n):
        """gz_fastq_zip_vblocks without its last wait (up to two calls in flight; text_buf and tab must stay alive until end())"""
        rc = self.E.L.gz_fastq_zip_begin(self.f, self.E.mem.ptr(text_buf), text_len, tab, n)
        if rc != GZ_OK:
            raise GenozipAMDError("gz_fastq_zip_begin failed (%d): %s" % (rc, self.E.L.gz_last_error(self.E.h).decode()))

    def end(self):
        """waits for the oldest call begun; its table holds the results"""
        rc = self.E.L.gz_fastq_zip_end(self.f)
        if rc != GZ_OK:
            raise GenozipAMDError("gz_fastq_zip_end failed (%d): %s" % (rc, self.E.L.gz_last_error(self.E.h).decode()))

    def speculation(self):
        """-> (hits, misses) of the handle: QUAL streams coded ahead with the previous file's codec, confirmed / refuted by the file's own trial"""
        h, m = C.c_uint32(0), C.c_uint32(0)
        self.E.L.gz_zip_speculation(self.f, C.byref(h), C.byref(m))
        return h.value, m.value

    def prediction(self):
        """-> (hits, misses) of the handle: sections coded ahead of their context's trial with a predicted codec, kept / coded again"""
        h, m = C.c_uint32(0), C.c_uint32(0)
        self.E.L.gz_zip_prediction(self.f, C.byref(h), C.byref(m))
        return h.value, m.value

    def reset(self):
        self.E._check(self.E.L.gz_zip_reset(self.f), "gz_zip_reset")

    def collect(self, tab, n, dst_buf, cap):
        """the last call's z_data, VBlock after VBlock, into dst_buf (device) -> offsets (n + 1)"""
        offs = (C.c_uint64 * (n + 1))()
        rc = self.E.L.gz_fastq_zip_collect(self.f, tab, n, self.E.mem.ptr(dst_buf), cap, offs)
        if rc != GZ_OK:
            raise GenozipAMDError("gz_fastq_zip_collect failed (%d)" % rc)
        return list(offs)

    # the three phases (a file dealt out over several processes; see genozip_amd/shard.py)
    def seg(self, text_buf, text_len, tab, n):
        """-> this process' merge blob (bytes)"""
        bp, bl = C.c_void_p(), C.c_uint64()
        rc = self.E.L.gz_fastq_zip_seg(self.f, self.E.mem.ptr(text_buf) if n else None, text_len, tab, n, C.byref(bp), C.byref(bl))
        if rc != GZ_OK:
            raise GenozipAMDError("gz_fastq_zip_seg failed (%d): %s" % (rc, self.E.L.gz_last_error(self.E.h).decode()))
        return C.string_at(bp.value, bl.value) if bl.value else b""

    @staticmethod
    def _blob_table(blobs):
        blobs = [bytes(b) for b in blobs if len(b)]
        n = len(blobs)
        ptrs = (C.c_void_p * max(1, n))(*[C.cast(C.c_char_p(b), C.c_void_p) for b in blobs])
        lens = (C.c_uint64 * max(1, n))(*[len(b) for b in blobs])
        return blobs, ptrs, lens, n

    def merge(self, blobs):
        """blobs: the merge blobs of ALL processes -> this process' codec votes (bytes)"""
        keep, ptrs, lens, n = self._blob_table(blobs)
        vp, vl = C.c_void_p(), C.c_uint64()
        rc = self.E.L.gz_fastq_zip_merge(self.f, ptrs, lens, n, C.byref(vp), C.byref(vl))
        if rc != GZ_OK:
            raise GenozipAMDError("gz_fastq_zip_merge failed (%d): %s" % (rc, self.E.L.gz_last_error(self.E.h).decode()))
        return C.string_at(vp.value, vl.value) if vl.value else b""

    def finish(self, votes):
        keep, ptrs, lens, n = self._blob_table(votes)
        rc = self.E.L.gz_fastq_zip_finish(self.f, ptrs, lens, n)
        if rc != GZ_OK:
            raise GenozipAMDError("gz_fastq_zip_finish failed (%d): %s" % (rc, self.E.L.gz_last_error(self.E.h).decode()))

    def results(self, tab):
        return [dict(z=self._download(t.z_data, t.z_len), seq_packed=self._download(t.seq_packed, t.seq_packed_len), n_bases=t.n_bases,
                     seq_has_x=bool(t.seq_has_x), n_reads=t.n_reads, n_sections=t.n_sections, vblock_i=t.vblock_i, text_len=t.text_len,
                     seq_section_index=t.seq_section_index) for t in tab]

    def zip_vblocks(self, text, vbs):
        """text: bytes; vbs as for vb_table -> list of dict(z=bytes, seq_packed=bytes, n_bases, seq_has_x, n_reads)"""
        import numpy as np
        buf = self.E.mem.upload(bytes(text) + b"\0" * 32)
        tab = self.vb_table(vbs)
        self.zip_table(buf, len(text), tab, len(vbs))
        out = []
        for t in tab:
            z = self._download(t.z_data, t.z_len)
            out.append(dict(z=z, seq_packed=self._download(t.seq_packed, t.seq_packed_len), n_bases=t.n_bases, seq_has_x=bool(t.seq_has_x),
                            n_reads=t.n_reads, n_sections=t.n_sections, vblock_i=t.vblock_i, text_len=t.text_len, seq_section_index=t.seq_section_index))
        # what goes to the writer: the same bytes, VBlock after VBlock, in one buffer (gz_fastq_zip_collect)
        total = sum(len(o["z"]) for o in out)
        dst = self.E.mem.upload(b"\xee" * (total + 64))
        offs = self.collect(tab, len(vbs), dst, total)
        assert offs[-1] == total and self.E.mem.download(dst, total + 64) == b"".join(o["z"] for o in out) + b"\xee" * 64, "gz_fastq_zip_collect"
        return out

    def _download(self, ptr, n):
        """bytes at a raw device pointer of the library's workspace"""
        if not n:
            return b""
        out = C.create_string_buffer(n)
        self.E._check(self.E.L.gz_download(self.E.h, out, ptr, n), "gz_download")
        return out.raw

    def insert_section(self, z, index, dict_id, codec, sub_codec, flags, ltype, param, payload, uncompressed_len):
        """gz_vb_insert_section: a section made on the host (NONREF: CODEC_ACGT's sub-codec is the host's) into a VBlock's z_data -> bytes"""
        L = self.E.L
        out, ol = C.create_string_buffer(len(z) + 40 + len(payload)), C.c_uint64(0)
        self.E._check(L.gz_vb_insert_section(z, len(z), index, dict_id, codec, sub_codec, flags, ltype, param, payload, len(payload), uncompressed_len,
                                             out, len(out), C.byref(ol)), "gz_vb_insert_section")
        return out.raw[:ol.value]

    def with_nonref(self, r, sub_compress, vb_size=16 << 20):
        """r: a VBlock result of zip_vblocks; sub_compress (packed bytes, vb_size) -> the LZMA stream of CODEC_ACGT's sub-codec (host work,
        SURVEY F8: the caller's encoder). -> z with the NONREF local section where it belongs (codec_acgt_compress, src/codec_acgt.c:64-177:
        codec ACGT, sub_codec LZMA - NONE under 50 packed bytes -, ltype LT_BLOB, flags.acgt_no_x when the VBlock has no NONREF_X,
        data_uncompressed_len = the number of bases: the reader sizes the packed words from it, :222-226)"""
        if not r["n_bases"]:
            return r["z"]
        from .fastq import dict_id
        packed = r["seq_packed"]
        sub = 4 if len(packed) >= 50 else 1
        payload = sub_compress(packed, vb_size) if sub == 4 else packed
        return self.insert_section(r["z"], r["seq_section_index"], dict_id("NONREF"), 10, sub, 0 if r["seq_has_x"] else 0x40, 11, 0, payload, r["n_bases"])

    def write_file(self, components, counts_ctxs=(), created=b"genozip_amd", vb_size=16 << 20, std_seq_len=0, std_seq_len_r2=0, vb_order=None, data_type=3):
        """components: [dict(name=bytes, pair=0 | 1 | 2, vbs=[VBlock results in the order they are written, each with z, n_reads, text_len]
        [, header=bytes: the component's header text (VCF / SAM), stored in its SEC_TXT_HEADER])]; data_type: DT_VCF 1, DT_SAM 2, DT_FASTQ 3
        -> the whole file: per component SEC_TXT_HEADER + its VBlocks (zfile_output_processed_vb), then zip_write_global_area (N4).
        vb_order = [(component, index into its vbs)]: the streamed form - both SEC_TXT_HEADERs first, then the VBlocks in the order the calls
        produced them (R1 and R2 VBlocks of a call next to each other)"""
        L, E = self.E.L, self.E
        zf = L.gz_zfile_create(data_type, vb_size)
        try:
            E._check(L.gz_zfile_set_fastq(zf, len(components), int(any(c.get("pair") for c in components)), std_seq_len, std_seq_len_r2), "gz_zfile_set_fastq")
            body = b""
            flav = bytes(8)                                    # QnameFlavorProp x NUM_QTYPES: no seq_len item, not mated, no cnn (Illumina-7, Illum-2bc)
            for ci, comp in enumerate(components):
                text = comp.get("header", b"")
                hdr, hl = C.create_string_buffer(400 + len(text)), C.c_uint64(0)
                E._check(L.gz_zfile_add_txt_header_text(zf, ci, comp.get("pair", 0), comp["name"], len(text) + sum(r["text_len"] for r in comp["vbs"]), sum(r["n_reads"] for r in comp["vbs"]),
                                                        max([r["n_reads"] for r in comp["vbs"]] + [0]), flav, 4, len(body), text, len(text), hdr, len(hdr), C.byref(hl)), "gz_zfile_add_txt_header_text")
                body += hdr.raw[:hl.value]
                for r in comp["vbs"] if vb_order is None else ():
                    E._check(L.gz_zfile_add_vblock(zf, r["z"], len(r["z"]), len(body), ci, r["n_reads"]), "gz_zfile_add_vblock")
                    body += r["z"]
            for ci, k in vb_order or ():
                r = components[ci]["vbs"][k]
                E._check(L.gz_zfile_add_vblock(zf, r["z"], len(r["z"]), len(body), ci, r["n_reads"]), "gz_zfile_add_vblock")
                body += r["z"]
            n = len(self.plan["ctxs"])
            zc = (C.c_void_p * n)(*[L.gz_zip_zctx(self.f, i) for i in range(n)])
            ids = b"".join(c["dict_id"] for c in self.plan["ctxs"])
            cs = bytes(int(i in counts_ctxs) for i in range(n))
            recon = sum(r["text_len"] for comp in components for r in comp["vbs"]) + sum(len(comp.get("header", b"")) for comp in components)
            lines = sum(r["n_reads"] for comp in components for r in comp["vbs"])
            cap = 1 << 20
            while True:
                out, ol = C.create_string_buffer(cap), C.c_uint64(0)
                rc = L.gz_zfile_write_global_area(zf, E.h, zc, ids, cs, n, len(body), recon, lines, created, out, cap, C.byref(ol))
                if rc == GZ_TOO_SMALL:                         # (the entries of a failed attempt are taken back: the same GzZFile again)
                    cap = ol.value + 64
                    continue
                E._check(rc, "gz_zfile_write_global_area")
                return body + out.raw[:ol.value]
        finally:
            L.gz_zfile_destroy(zf)

    def zctx_words(self, ctx_i):
        from .lib import GzZctxView
        v = GzZctxView()
        self.E.L.gz_zctx_view(self.E.L.gz_zip_zctx(self.f, ctx_i), C.byref(v))
        d = C.string_at(v.dict, v.dict_len) if v.dict_len else b""
        return d[:-1].split(b"\0") if d else []

    def zctx_view(self, ctx_i):
        """-> dict(n_words, all_the_same_wi (-1: not set), rm_dict, lcodec, bcodec) of the file-level context"""
        from .lib import GzZctxView
        v = GzZctxView()
        self.E.L.gz_zctx_view(self.E.L.gz_zip_zctx(self.f, ctx_i), C.byref(v))
        return dict(n_words=v.n_words, all_the_same_wi=v.all_the_same_wi, rm_dict=bool(v.rm_dict_all_the_same), lcodec=v.lcodec, bcodec=v.bcodec)


class Engine:
    def __init__(self, device=0, lib_path=None, mem=None, hip_stream=None):
        self.L = _lib.load(lib_path)
        if mem is None:
            from .mem import TorchMem
            mem = TorchMem(device)
        self.mem = mem
        self._zip_files = []
        err = C.c_int(0)
        self.h = self.L.gz_create(device, hip_stream, C.byref(err))
        if not self.h:
            raise GenozipAMDError("gz_create failed (%d): no usable GPU / HIP runtime - genozip_amd has no CPU fallback" % err.value)

    def close(self):
        if getattr(self, "h", None):
            for r in getattr(self, "_zip_files", []):
                zf = r()
                if zf is not None:
                    zf.close()
            self.L.gz_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise GenozipAMDError("%s failed (%d): %s" % (what, rc, self.L.gz_last_error(self.h).decode()))
        return rc

    def version(self):
        return self.L.gz_version().decode()

    def debug_record_inv(self, tot0, n):
        """(tests) the reciprocals the model kernel puts into the records of totals tot0 .. tot0 + n - 1, as numpy float64"""
        import numpy as np
        out = self.mem.alloc(8 * n)
        self._check(self.L.gz_debug_record_inv(self.h, tot0, n, self.mem.ptr(out)), "gz_debug_record_inv")
        return np.frombuffer(self.mem.download(out, 8 * n), dtype=np.float64).copy()

    def sync(self):
        return self._check(self.L.gz_sync(self.h), "gz_sync")

    def profile(self, enable=True, reset=False):
        """enable: False / True / 2 (only k_arith_chain and k_arith_model: gz_profile)"""
        self.L.gz_profile(self.h, int(enable), int(reset))

    def compress_lines(self, codec, lines, capacity=None, soft_fail=False):
        """COMPRESS() with a get_line callback (src/codec.h:23, codec_htscodecs.c:51-64): lines = list of bytes"""
        from .lib import GzGetLineCB
        keep = [C.create_string_buffer(bytes(l), max(1, len(l))) for l in lines]
        total = sum(len(l) for l in lines)

        def cb(user, i, line_p, len_p):
            line_p[0] = C.cast(keep[i], C.c_void_p).value
            len_p[0] = len(lines[i])
        cap = self.est_size(codec, total) if capacity is None else capacity
        out = C.create_string_buffer(max(1, cap))
        ol = C.c_uint32(cap)
        rc = self.L.gz_codec_compress_lines_host(self.h, codec, GzGetLineCB(cb), None, len(lines), total, out, C.byref(ol), int(soft_fail))
        if rc == 0 and soft_fail:                  # GZ_TOO_SMALL
            return None
        self._check(rc, "gz_codec_compress_lines_host")
        return out.raw[:ol.value]

    def profile_results(self):
        """{kernel name: (total ms, launches)} accumulated by gz_sync() since the last reset"""
        out, i = {}, 0
        name = C.create_string_buffer(64)
        ms, n = C.c_double(0), C.c_int(0)
        mx = C.c_double(0)
        self.profile_max = {}
        while self.L.gz_profile_get(self.h, i, name, 64, C.byref(ms), C.byref(n)):
            out[name.value.decode()] = (ms.value, n.value)
            if self.L.gz_profile_get_max(self.h, i, C.byref(mx)):
                self.profile_max[name.value.decode()] = mx.value       # the longest single launch
            i += 1
        return out

    def hip_stream(self):
        return self.L.gz_stream(self.h)

    # ---- codec_args[] ---------------------------------------------------------------------------------------
    def est_size(self, codec, n):
        return self.L.gz_codec_est_size(codec, n)

    def compress(self, codec, data, capacity=None, soft_fail=False):
        data = bytes(data)
        cap = self.est_size(codec, len(data)) if capacity is None else capacity
        out = C.create_string_buffer(max(1, cap))
        ol = C.c_uint32(cap)
        rc = self.L.gz_codec_compress_host(self.h, codec, data, len(data), out, C.byref(ol), int(soft_fail))
        if rc == GZ_TOO_SMALL and soft_fail:
            return None
        if rc != GZ_OK:
            raise GenozipAMDError("codec_compress(%s) failed (%d): %s" % (_lib.CODEC_NAMES.get(codec, codec), rc, self.L.gz_last_error(self.h).decode()))
        return out.raw[:ol.value]

    def uncompress(self, codec, compressed, uncompressed_len):
        compressed = bytes(compressed)
        out = C.create_string_buffer(max(1, uncompressed_len))
        rc = self.L.gz_codec_uncompress_host(self.h, codec, compressed, len(compressed), out, uncompressed_len)
        if rc != GZ_OK:
            raise GenozipAMDError("codec_uncompress(%s) failed (%d): %s" % (_lib.CODEC_NAMES.get(codec, codec), rc, self.L.gz_last_error(self.h).decode()))
        return out.raw[:uncompressed_len]

    # ---- stream tables on the device ----------------------------------------------------------------------------
    def make_stream_table(self, items):
        """items: (codec, device_buffer_in, in_len) -> (ctypes table, list of output buffers)"""
        n = len(items)
        tab = (GzStream * max(1, n))()
        outs = []
        for i, (codec, buf, in_len) in enumerate(items):
            cap = self.est_size(codec, in_len)
            ob = self.mem.alloc(cap + 16)
            outs.append(ob)
            tab[i].in_ = self.mem.ptr(buf)
            tab[i].in_len = in_len
            tab[i].in_len_dev = None
            tab[i].out = self.mem.ptr(ob)
            tab[i].out_cap = cap
            tab[i].codec = codec
        return tab, outs

    def compress_table(self, tab, n):
        self._check(self.L.gz_codec_compress_batch(self.h, tab, n), "gz_codec_compress_batch")

    def compress_many(self, items):
        """items: list of (codec, bytes). One batched launch; returns the payloads."""
        bufs = [self.mem.upload(d) for _, d in items]
        tab, outs = self.make_stream_table([(c, b, len(d)) for (c, d), b in zip(items, bufs)])
        self.compress_table(tab, len(items))
        self.sync()
        res = []
        for i in range(len(items)):
            if tab[i].status != GZ_OK:
                raise GenozipAMDError("stream %d: status %d" % (i, tab[i].status))
            res.append(self.mem.download(outs[i], tab[i].out_len))
        return res

    def uncompress_many(self, items):
        """items: list of (codec, compressed bytes, uncompressed_len)"""
        n = len(items)
        tab = (GzStream * max(1, n))()
        ins, outs = [], []
        for i, (codec, comp, ulen) in enumerate(items):
            ib = self.mem.upload(comp)
            ob = self.mem.alloc(ulen + 16)
            ins.append(ib)
            outs.append(ob)
            tab[i].in_ = self.mem.ptr(ib)
            tab[i].in_len = len(comp)
            tab[i].out = self.mem.ptr(ob)
            tab[i].out_cap = ulen
            tab[i].codec = codec
        self._check(self.L.gz_codec_uncompress_batch(self.h, tab, n), "gz_codec_uncompress_batch")
        self.sync()
        res = []
        for i in range(n):
            if tab[i].status != GZ_OK:
                raise GenozipAMDError("stream %d: decode status %d" % (i, tab[i].status))
            res.append(self.mem.download(outs[i], items[i][2]))
        return res

    def assign_best(self, data):
        buf = self.mem.upload(data)
        sizes = (C.c_uint32 * 9)()
        c = self._check(self.L.gz_codec_assign_best(self.h, self.mem.ptr(buf), len(data), sizes), "gz_codec_assign_best")
        return c, list(sizes)

    def assign_sort(self, tests, mode=0):
        """gz_codec_assign_sort: tests = [(codec, size, clock_us)] in trial order -> (winner, sorted list). Host only."""
        from .lib import GzCodecTest
        n = len(tests)
        tab = (GzCodecTest * max(1, n))(*[GzCodecTest(int(c), float(sz), float(ck)) for c, sz, ck in tests])
        w = self._check(self.L.gz_codec_assign_sort(tab, n, mode), "gz_codec_assign_sort")
        return w, [(t.codec, t.size, t.clock_us) for t in tab[:n]]

    def assign_best_ex(self, data, extra=(), clock_ns_per_byte=None, mode=0):
        """gz_codec_assign_best_ex: the nine device candidates + the caller's rows [(codec, framed size, clock_us)] -> (codec, sorted table)"""
        from .lib import GzCodecTest
        buf = self.mem.upload(data)
        ex = (GzCodecTest * max(1, len(extra)))(*[GzCodecTest(int(c), float(sz), float(ck)) for c, sz, ck in extra])
        out = (GzCodecTest * (9 + len(extra)))()
        clk = (C.c_float * 32)(*clock_ns_per_byte) if clock_ns_per_byte is not None else None
        c = self._check(self.L.gz_codec_assign_best_ex(self.h, self.mem.ptr(buf), len(data), ex, len(extra), clk, mode, out), "gz_codec_assign_best_ex")
        return c, [(t.codec, t.size, t.clock_us) for t in out] if c else []

    # ---- context engine -------------------------------------------------------------------------------------
    def b250_generate_many(self, jobs, r1=None):
        """jobs: list of (seg bytes, ol_nodes_len, node2word list) -> list of PIZ-format bytes; r1: optional list of the
        generated b250 of the same contexts in R1's VBlock (bytes or None) - an identical one gives None (dropped)"""
        import numpy as np
        n = len(jobs)
        tab = (GzB250Job * max(1, n))()
        keep, outs, lens = [], [], []
        for i, (seg, ol, n2w) in enumerate(jobs):
            sb = self.mem.upload(seg)
            nb = self.mem.upload(np.asarray(list(n2w) or [0], dtype=np.int32))
            ob = self.mem.alloc(len(seg) + 16)
            lb = self.mem.alloc(8)
            keep += [sb, nb]
            outs.append(ob)
            lens.append(lb)
            tab[i].seg = self.mem.ptr(sb)
            tab[i].seg_len = len(seg)
            tab[i].ol_nodes_len = ol
            tab[i].node2word = self.mem.ptr(nb)
            tab[i].n_new_nodes = len(n2w)
            tab[i].out = self.mem.ptr(ob)
            tab[i].out_len_dev = self.mem.ptr(lb)
            tab[i].status_dev = self.mem.ptr(lb) + 4
            if r1 is not None and r1[i] is not None:
                rb = self.mem.upload(r1[i]); rl = self.mem.upload(np.array([len(r1[i])], dtype=np.uint32))
                keep += [rb, rl]
                tab[i].r1 = self.mem.ptr(rb); tab[i].r1_len_dev = self.mem.ptr(rl)
        self._check(self.L.gz_b250_generate_batch(self.h, tab, n), "gz_b250_generate_batch")
        self.sync()
        res = []
        for i in range(n):
            ln, st = np.frombuffer(self.mem.download(lens[i], 8), dtype=np.int32)
            if st == 3:
                res.append(None)                         # identical to R1: no section
                continue
            if st != 1:
                raise GenozipAMDError("b250 %d: malformed seg-format stream or node index out of range (status %d)" % (i, st))
            res.append(self.mem.download(outs[i], int(ln)))
        return res

    def b250_generate(self, seg, ol_nodes_len, node2word):
        return self.b250_generate_many([(seg, ol_nodes_len, node2word)])[0]

    def local_generate(self, ltype, raw_native_le, transpose_cols=0):
        buf = self.mem.upload(raw_native_le)
        scratch = self.mem.alloc(len(raw_native_le) + 16)
        w = {3: 2, 4: 2, 15: 2, 5: 4, 6: 4, 9: 4, 16: 4, 7: 8, 8: 8, 10: 8, 12: 8}.get(ltype, 1)
        lt = self._check(self.L.gz_local_generate(self.h, ltype, self.mem.ptr(buf), len(raw_native_le) // w, transpose_cols, self.mem.ptr(scratch)), "gz_local_generate")
        self.sync()
        return lt, self.mem.download(buf, len(raw_native_le))

    def local_to_native(self, ltype, file_bytes, transpose_cols=0):
        buf = self.mem.upload(file_bytes)
        scratch = self.mem.alloc(len(file_bytes) + 16)
        w = {3: 2, 4: 2, 15: 2, 5: 4, 6: 4, 9: 4, 16: 4, 7: 8, 8: 8, 10: 8, 12: 8}.get(ltype, 1)
        lt = self._check(self.L.gz_local_to_native(self.h, ltype, self.mem.ptr(buf), len(file_bytes) // w, transpose_cols, self.mem.ptr(scratch)), "gz_local_to_native")
        self.sync()
        return lt, self.mem.download(buf, len(file_bytes))

    def adler32(self, data):
        buf = self.mem.upload(data)
        a = C.c_uint32(0)
        self._check(self.L.gz_adler32(self.h, self.mem.ptr(buf), len(data), C.byref(a)), "gz_adler32")
        return a.value

    def zctx(self, estimated_entries=0):
        return Zctx(self.L, estimated_entries)

    # ---- seg-side appends, a column at a time (rows a1-a3) ------------------------------------------------
    def ctx_seg_columns(self, columns, keep_on_device=False, dict_cap=None):
        """columns: list of (text bytes | device buffer, off u32[n], len u32[n], ol_snips) -> list of dicts with the
        keys node_index, dict, node_char_index, node_snip_len, counts, b250, b250_count, all_the_same
        (ctx_create_node_do + b250_seg_append over the whole column)"""
        import numpy as np
        from .lib import GzColumnJob, GzColumnResult
        nj = len(columns)
        tab = (GzColumnJob * max(1, nj))()
        keep, outs = [], []
        texts = {}
        for i, (text, off, length, ol_snips) in enumerate(columns):
            off = np.ascontiguousarray(off, dtype=np.uint32); length = np.ascontiguousarray(length, dtype=np.uint32)
            n, n_ol = len(off), len(ol_snips)
            if isinstance(text, (bytes, bytearray)):
                if id(text) not in texts:
                    texts[id(text)] = self.mem.upload(text)
                tbuf = texts[id(text)]
            else:
                tbuf = text
            ol_dict = b"".join(bytes(w) + b"\0" for w in ol_snips)
            ol_len = np.array([len(w) for w in ol_snips], dtype=np.uint32)
            ol_ci = np.zeros(n_ol, dtype=np.uint64)
            if n_ol > 1:
                ol_ci[1:] = np.cumsum(ol_len[:-1].astype(np.uint64) + 1)
            dict_cap_i = int(length.astype(np.uint64).sum()) + n if dict_cap is None else int(dict_cap)
            b = dict(off=self.mem.upload(off), len=self.mem.upload(length), ol_dict=self.mem.upload(ol_dict),
                     ol_ci=self.mem.upload(ol_ci), ol_len=self.mem.upload(ol_len), ni=self.mem.alloc(4 * n + 16),
                     dict=self.mem.alloc(dict_cap_i + 16), nci=self.mem.alloc(8 * n + 16), nsl=self.mem.alloc(4 * n + 16),
                     counts=self.mem.alloc(4 * (n + n_ol) + 16), b250=self.mem.alloc(4 * n + 16), res=self.mem.alloc(C.sizeof(GzColumnResult)))
            keep.append((tbuf, b))
            j = tab[i]
            j.text = self.mem.ptr(tbuf); j.off = self.mem.ptr(b["off"]); j.len = self.mem.ptr(b["len"]); j.n = n
            j.ol_dict = self.mem.ptr(b["ol_dict"]); j.ol_char_index = self.mem.ptr(b["ol_ci"]); j.ol_snip_len = self.mem.ptr(b["ol_len"]); j.n_ol = n_ol
            j.node_index = self.mem.ptr(b["ni"]); j.dict = self.mem.ptr(b["dict"]); j.dict_cap = dict_cap_i
            j.node_char_index = self.mem.ptr(b["nci"]); j.node_snip_len = self.mem.ptr(b["nsl"]); j.counts = self.mem.ptr(b["counts"])
            j.b250 = self.mem.ptr(b["b250"]); j.result_dev = self.mem.ptr(b["res"])
            outs.append((n, n_ol, b))
        self._check(self.L.gz_ctx_seg_columns(self.h, tab, nj), "gz_ctx_seg_columns")
        self.sync()
        if keep_on_device:
            return outs
        res = []
        for n, n_ol, b in outs:
            r = GzColumnResult.from_buffer_copy(self.mem.download(b["res"], C.sizeof(GzColumnResult)))
            if r.status != 1:
                raise GenozipAMDError("gz_ctx_seg_columns: dict capacity too small")
            res.append(dict(node_index=np.frombuffer(self.mem.download(b["ni"], 4 * n), dtype=np.int32),
                            dict=self.mem.download(b["dict"], r.dict_len),
                            node_char_index=np.frombuffer(self.mem.download(b["nci"], 8 * r.n_new), dtype=np.uint64),
                            node_snip_len=np.frombuffer(self.mem.download(b["nsl"], 4 * r.n_new), dtype=np.uint32),
                            counts=np.frombuffer(self.mem.download(b["counts"], 4 * (n_ol + r.n_new)), dtype=np.uint32),
                            b250=self.mem.download(b["b250"], r.b250_len), b250_count=int(r.b250_count),
                            all_the_same=bool(r.all_the_same)))
        return res

    def ctx_seg_column(self, text, off, length, ol_snips=(), dict_cap=None):
        return self.ctx_seg_columns([(bytes(text), off, length, list(ol_snips))], dict_cap=dict_cap)[0]

    def dyn_int_columns(self, columns):
        """columns: list of (int64 values, is_nothing | None, nothing_char) -> list of (ltype, native LE bytes)"""
        import numpy as np
        from .lib import GzDynIntJob, GzDynIntResult
        nj = len(columns)
        tab = (GzDynIntJob * max(1, nj))()
        keep = []
        for i, (values, is_nothing, nothing_char) in enumerate(columns):
            v = np.ascontiguousarray(values, dtype=np.int64)
            vb = self.mem.upload(v)
            mb = None if is_nothing is None else self.mem.upload(np.ascontiguousarray(is_nothing, dtype=np.uint8))
            ob = self.mem.alloc(8 * len(v) + 16); rb = self.mem.alloc(C.sizeof(GzDynIntResult))
            keep.append((vb, mb, ob, rb))
            j = tab[i]
            j.values = self.mem.ptr(vb); j.is_nothing = self.mem.ptr(mb) if mb is not None else None; j.n = len(v)
            j.nothing_char = int(nothing_char); j.out = self.mem.ptr(ob); j.result_dev = self.mem.ptr(rb)
        self._check(self.L.gz_dyn_int_columns(self.h, tab, nj), "gz_dyn_int_columns")
        self.sync()
        res = []
        for vb, mb, ob, rb in keep:
            r = GzDynIntResult.from_buffer_copy(self.mem.download(rb, C.sizeof(GzDynIntResult)))
            res.append((r.ltype, self.mem.download(ob, r.len)))
        return res

    def dyn_int_column(self, values, is_nothing=None, nothing_char=0):
        return self.dyn_int_columns([(values, is_nothing, nothing_char)])[0]

    def local_blob_columns(self, columns, want_off=False):
        """columns: list of (text bytes | device buffer, off, len, add_nul[, pre bytes, pad_to, pad_byte]) -> list of bytes
        (want_off: list of (bytes, item offsets))"""
        import numpy as np
        from .lib import GzBlobJob
        nj = len(columns)
        tab = (GzBlobJob * max(1, nj))()
        keep, texts, offs = [], {}, []
        for i, col in enumerate(columns):
            text, off, length, add_nul = col[:4]
            pre, pad_to, pad_byte = (tuple(col[4:]) + (b"", 0, 0))[:3] if len(col) > 4 else (b"", 0, 0)
            off = np.ascontiguousarray(off, dtype=np.uint32); length = np.ascontiguousarray(length, dtype=np.uint32)
            if isinstance(text, (bytes, bytearray)):
                if id(text) not in texts:
                    texts[id(text)] = self.mem.upload(text)
                tbuf = texts[id(text)]
            else:
                tbuf = text
            cap = int(length.astype(np.uint64).sum()) + len(off) * (1 + len(pre) + pad_to)
            ofb, lb, ob, rb = self.mem.upload(off), self.mem.upload(length), self.mem.alloc(cap + 16), self.mem.alloc(8)
            iob = self.mem.alloc(4 * len(off) + 16) if want_off else None
            keep.append((tbuf, ofb, lb, ob, rb))
            offs.append((iob, len(off)))
            j = tab[i]
            j.text = self.mem.ptr(tbuf); j.off = self.mem.ptr(ofb); j.len = self.mem.ptr(lb); j.n = len(off)
            j.add_nul = int(bool(add_nul)); j.out = self.mem.ptr(ob); j.out_len_dev = self.mem.ptr(rb)
            j.pre = (C.c_uint8 * 4)(*(bytes(pre) + b"\0" * 4)[:4]); j.pre_len = len(pre); j.pad_to = int(pad_to); j.pad_byte = int(pad_byte)
            j.item_off = self.mem.ptr(iob) if want_off else None
        self._check(self.L.gz_local_blob_columns(self.h, tab, nj), "gz_local_blob_columns")
        self.sync()
        blobs = [self.mem.download(ob, int(np.frombuffer(self.mem.download(rb, 8), dtype=np.uint64)[0])) for _, _, _, ob, rb in keep]
        if not want_off:
            return blobs
        return [(b, np.frombuffer(self.mem.download(iob, 4 * n), dtype=np.uint32).copy() if n else np.zeros(0, dtype=np.uint32)) for b, (iob, n) in zip(blobs, offs)]

    # ---- N3: CODEC_DOMQ's pre-transform -----------------------------------------------------------------------
    def domq_columns(self, columns):
        """columns: list of (text bytes, off, len) - the QUAL lines of a VBlock each -> list of dict(qual, runs, mplx, divr, denorm,
        num_doms, num_norm_qs, has_diverse, all_diverse, fit)"""
        import numpy as np
        from .lib import GzDomqJob, GzDomqResult, GzDomqFitJob
        nj = len(columns)
        tab, ftab = (GzDomqJob * max(1, nj))(), (GzDomqFitJob * max(1, nj))()
        keep = []
        for i, (text, off, length) in enumerate(columns):
            off = np.ascontiguousarray(off, dtype=np.uint32); length = np.ascontiguousarray(length, dtype=np.uint32)
            B, n = int(length.astype(np.uint64).sum()), len(off)
            b = dict(text=self.mem.upload(bytes(text) + b"\0"), off=self.mem.upload(off), len=self.mem.upload(length), qual=self.mem.alloc(2 * B + 16),
                     runs=self.mem.alloc(B + B // 254 + 16), mplx=self.mem.alloc(n + 16), divr=self.mem.alloc(B + 16), res=self.mem.alloc(C.sizeof(GzDomqResult)), fit=self.mem.alloc(16))
            keep.append(b)
            j = tab[i]
            j.text, j.off, j.len, j.n = self.mem.ptr(b["text"]), self.mem.ptr(b["off"]), self.mem.ptr(b["len"]), n
            j.qual, j.runs, j.mplx, j.divr, j.result_dev = self.mem.ptr(b["qual"]), self.mem.ptr(b["runs"]), self.mem.ptr(b["mplx"]), self.mem.ptr(b["divr"]), self.mem.ptr(b["res"])
            fj = ftab[i]
            fj.text, fj.off, fj.len, fj.n, fj.fit_dev = j.text, j.off, j.len, n, self.mem.ptr(b["fit"])
        self._check(self.L.gz_domq_fit(self.h, ftab, nj), "gz_domq_fit")
        self._check(self.L.gz_domq_columns(self.h, tab, nj), "gz_domq_columns")
        self.sync()
        out = []
        for b in keep:
            r = GzDomqResult.from_buffer_copy(self.mem.download(b["res"], C.sizeof(GzDomqResult)))
            if r.status != 1:
                raise GenozipAMDError("gz_domq_columns: a quality score outside ' '..'~'")
            out.append(dict(qual=self.mem.download(b["qual"], r.qual_len), runs=self.mem.download(b["runs"], r.runs_len), mplx=self.mem.download(b["mplx"], r.mplx_len),
                            divr=self.mem.download(b["divr"], r.divr_len), denorm=bytes(r.denorm[:r.num_doms * r.num_norm_qs]), num_doms=r.num_doms, num_norm_qs=r.num_norm_qs,
                            has_diverse=bool(r.has_diverse), all_diverse=bool(r.all_diverse), fit=bool(np.frombuffer(self.mem.download(b["fit"], 4), dtype=np.uint32)[0])))
        return out

    # ---- N1 (first part): lines, FASTQ records, tokens ----------------------------------------------------
    def text_lines(self, text, cap=None, on_device=False):
        """seg_get_next_line over the whole buffer -> (line_off, line_len) numpy arrays (or device buffers + count)"""
        import numpy as np
        tbuf = self.mem.upload(text) if isinstance(text, (bytes, bytearray)) else text
        n = len(text) if isinstance(text, (bytes, bytearray)) else int(tbuf.numel() if hasattr(tbuf, "numel") else tbuf.size)
        if cap is None:
            cap = (bytes(text).count(b"\n") + 1) if isinstance(text, (bytes, bytearray)) else n // 2 + 1
        ob, lb, rb = self.mem.alloc(4 * cap + 16), self.mem.alloc(4 * cap + 16), self.mem.alloc(16)
        self._check(self.L.gz_text_lines(self.h, self.mem.ptr(tbuf), n, self.mem.ptr(ob), self.mem.ptr(lb), cap, self.mem.ptr(rb)), "gz_text_lines")
        self.sync()
        res = np.frombuffer(self.mem.download(rb, 16), dtype=np.uint64)
        n_lines, status = int(res[0]), int(np.frombuffer(self.mem.download(rb, 16), dtype=np.int32)[2])
        if status != 1:
            raise GenozipAMDError("gz_text_lines: %d lines do not fit cap=%d" % (n_lines, cap))
        if on_device:
            return tbuf, ob, lb, rb, n_lines
        return (np.frombuffer(self.mem.download(ob, 4 * n_lines), dtype=np.uint32), np.frombuffer(self.mem.download(lb, 4 * n_lines), dtype=np.uint32))

    # ---- N1 for BAM: alignment records -> alignment lines ----------------------------------------------------
    def _bam_result(self, rb):
        from .lib import GzBamResult
        return GzBamResult.from_buffer_copy(self.mem.download(rb, C.sizeof(GzBamResult)))

    def bam_records(self, bam, n_ref, cap=None, on_device=False):
        """bam: the alignment records (bytes | device buffer) -> record offsets (numpy, or (device buffer, count) with on_device);
        raises on a chain that does not fit the stream. self.last_bam = the GzBamResult"""
        import numpy as np
        host = isinstance(bam, (bytes, bytearray))
        bbuf = self.mem.upload(bytes(bam) + b"\0" * 16) if host else bam
        n = len(bam) if host else int(bbuf.numel() if hasattr(bbuf, "numel") else bbuf.size)
        if cap is None:
            cap = n // 36 + 1
        ob, rb = self.mem.alloc(4 * cap + 16), self.mem.alloc(64)
        self._check(self.L.gz_bam_records(self.h, self.mem.ptr(bbuf), n, n_ref, self.mem.ptr(ob), cap, self.mem.ptr(rb)), "gz_bam_records")
        self.sync()
        r = self.last_bam = self._bam_result(rb)
        if r.status != 1:
            raise GenozipAMDError("gz_bam_records: status %d, record %d, %d records" % (r.status, r.first_bad, r.n_records))
        if on_device:
            return bbuf, ob, int(r.n_records)
        return np.frombuffer(self.mem.download(ob, 4 * int(r.n_records)), dtype=np.uint32).copy()

    def bam_to_sam(self, bam, rec_off, ref_names, text_cap=None, on_device=False, n_rec=None):
        """records -> the text of their alignment lines. ref_names: list of bytes (the header's reference names, in id order).
        -> (text bytes, line_off numpy [n + 1]); on_device: (device text buffer, text_len, device line_off buffer)"""
        import numpy as np
        host = isinstance(bam, (bytes, bytearray))
        bbuf = self.mem.upload(bytes(bam) + b"\0" * 16) if host else bam
        n = len(bam) if host else int(bbuf.numel() if hasattr(bbuf, "numel") else bbuf.size)
        if isinstance(rec_off, np.ndarray):
            n_rec = len(rec_off)
            robuf = self.mem.upload(np.ascontiguousarray(rec_off, dtype=np.uint32).tobytes() + b"\0" * 16)
        else:
            robuf = rec_off
        names = b"".join(ref_names)
        noff = np.concatenate([[0], np.cumsum([len(x) for x in ref_names])]).astype(np.uint32)
        nbuf, nobuf = self.mem.upload(names + b"\0" * 16), self.mem.upload(noff.tobytes() + b"\0" * 16)
        if text_cap is None:
            text_cap = 6 * n + 64 * n_rec + 1024                                      # (an int8 array element: 1 byte -> up to 5 characters)
        tb, lb, rb = self.mem.alloc(text_cap + 64), self.mem.alloc(4 * (n_rec + 1) + 16), self.mem.alloc(64)
        self._check(self.L.gz_bam_to_sam(self.h, self.mem.ptr(bbuf), n, self.mem.ptr(robuf), n_rec, self.mem.ptr(nbuf), self.mem.ptr(nobuf), len(ref_names),
                                         self.mem.ptr(tb), text_cap, self.mem.ptr(lb), self.mem.ptr(rb)), "gz_bam_to_sam")
        self.sync()
        r = self.last_bam = self._bam_result(rb)
        if r.status != 1:
            raise GenozipAMDError("gz_bam_to_sam: status %d, record %d, text %d bytes (cap %d)" % (r.status, r.first_bad, r.text_len, text_cap))
        if on_device:
            return tb, int(r.text_len), lb
        return self.mem.download(tb, int(r.text_len)), np.frombuffer(self.mem.download(lb, 4 * (n_rec + 1)), dtype=np.uint32).copy()

    def vcf_sample_columns(self, text, line_off, line_len, n_samples, n_sub):
        """the FORMAT subfields of every sample of the given data lines -> (n_bad, item_off [n_sub][lines * samples], item_len, missing)"""
        import numpy as np
        text = bytes(text)
        tbuf = self.mem.upload(text + b"\0" * 16)
        cap = text.count(b"\t") + 1
        ab, rb = self.mem.alloc(4 * (cap + 2) + 16), self.mem.alloc(16)
        self._check(self.L.gz_byte_index(self.h, self.mem.ptr(tbuf), len(text), 9, self.mem.ptr(ab), cap, self.mem.ptr(rb)), "gz_byte_index")
        lo = np.ascontiguousarray(line_off, dtype=np.uint32); ll = np.ascontiguousarray(line_len, dtype=np.uint32)
        n = len(lo)
        lob, llb = self.mem.upload(lo), self.mem.upload(ll)
        tot = max(1, n * n_samples * n_sub)
        io, il, mi, nb = self.mem.alloc(4 * tot + 16), self.mem.alloc(4 * tot + 16), self.mem.alloc(tot + 16), self.mem.alloc(16)
        self._check(self.L.gz_vcf_sample_columns(self.h, self.mem.ptr(tbuf), self.mem.ptr(lob), self.mem.ptr(llb), n, self.mem.ptr(ab), self.mem.ptr(rb), n_samples, n_sub,
                                                 self.mem.ptr(io), self.mem.ptr(il), self.mem.ptr(mi), self.mem.ptr(nb)), "gz_vcf_sample_columns")
        self.sync()
        k = n * n_samples
        return (int(np.frombuffer(self.mem.download(nb, 4), dtype=np.uint32)[0]),
                np.frombuffer(self.mem.download(io, 4 * k * n_sub), dtype=np.uint32).reshape(n_sub, k),
                np.frombuffer(self.mem.download(il, 4 * k * n_sub), dtype=np.uint32).reshape(n_sub, k),
                np.frombuffer(self.mem.download(mi, k * n_sub), dtype=np.uint8).reshape(n_sub, k))

    def byte_index(self, text, byte):
        """-> positions behind every occurrence of `byte` (numpy)"""
        import numpy as np
        text = bytes(text)
        tbuf = self.mem.upload(text + b"\0" * 16)
        cap = text.count(bytes([byte])) + 1
        ab, rb = self.mem.alloc(4 * (cap + 2) + 16), self.mem.alloc(16)
        self._check(self.L.gz_byte_index(self.h, self.mem.ptr(tbuf), len(text), byte, self.mem.ptr(ab), cap, self.mem.ptr(rb)), "gz_byte_index")
        self.sync()
        n = int(np.frombuffer(self.mem.download(rb, 8), dtype=np.uint64)[0])
        return np.frombuffer(self.mem.download(ab, 4 * (n + 1)), dtype=np.uint32)[1:].copy()

    def fastq_records(self, text):
        """-> (first_bad or None, [(off, len)] for line 1, SEQ, line 3, QUAL) from the text of whole reads"""
        import numpy as np
        tbuf, ob, lb, rb, n_lines = self.text_lines(text, on_device=True)
        nr = n_lines // 4
        cols = [self.mem.alloc(4 * nr + 16) for _ in range(8)]
        fb = self.mem.alloc(16)
        self._check(self.L.gz_fastq_records(self.h, self.mem.ptr(tbuf), self.mem.ptr(ob), self.mem.ptr(lb), self.mem.ptr(rb), nr,
                                            *[self.mem.ptr(c) for c in cols], self.mem.ptr(fb)), "gz_fastq_records")
        self.sync()
        raw = self.mem.download(fb, 16)
        n_reads, first_bad = int(np.frombuffer(raw, dtype=np.uint64)[0]), int(np.frombuffer(raw, dtype=np.uint32)[2])
        assert n_reads == nr
        arr = [np.frombuffer(self.mem.download(c, 4 * nr), dtype=np.uint32) for c in cols]
        return (None if first_bad == 0xffffffff else first_bad), [(arr[2 * i], arr[2 * i + 1]) for i in range(4)]

    def tokenize_column(self, text, off, length, seps):
        """-> (n_bad, item_off[n_items, n], item_len[n_items, n])"""
        import numpy as np
        seps = bytes(seps)
        off = np.ascontiguousarray(off, dtype=np.uint32); length = np.ascontiguousarray(length, dtype=np.uint32)
        n, ni = len(off), len(seps) + 1
        tbuf = self.mem.upload(text) if isinstance(text, (bytes, bytearray)) else text
        ofb, lb = self.mem.upload(off), self.mem.upload(length)
        io, il, nb = self.mem.alloc(4 * ni * n + 16), self.mem.alloc(4 * ni * n + 16), self.mem.alloc(16)
        self._check(self.L.gz_tokenize_column(self.h, self.mem.ptr(tbuf), self.mem.ptr(ofb), self.mem.ptr(lb), n, seps, len(seps),
                                              self.mem.ptr(io), self.mem.ptr(il), self.mem.ptr(nb)), "gz_tokenize_column")
        self.sync()
        return (int(np.frombuffer(self.mem.download(nb, 4), dtype=np.uint32)[0]),
                np.frombuffer(self.mem.download(io, 4 * ni * n), dtype=np.uint32).reshape(ni, n),
                np.frombuffer(self.mem.download(il, 4 * ni * n), dtype=np.uint32).reshape(ni, n))

    def seg_integer_or_not(self, text, off, length, nothing_char=0, lookup_off=0):
        """-> (snip_off, snip_len, values, is_nothing): seg_integer_or_not over a column"""
        import numpy as np
        off = np.ascontiguousarray(off, dtype=np.uint32); length = np.ascontiguousarray(length, dtype=np.uint32)
        n = len(off)
        tbuf = self.mem.upload(text) if isinstance(text, (bytes, bytearray)) else text
        ofb, lb = self.mem.upload(off), self.mem.upload(length)
        so, sl, vb, mb, nb = self.mem.alloc(4 * n + 16), self.mem.alloc(4 * n + 16), self.mem.alloc(8 * n + 16), self.mem.alloc(n + 16), self.mem.alloc(16)
        self._check(self.L.gz_seg_integer_or_not(self.h, self.mem.ptr(tbuf), self.mem.ptr(ofb), self.mem.ptr(lb), n, int(nothing_char), int(lookup_off),
                                                 self.mem.ptr(so), self.mem.ptr(sl), self.mem.ptr(vb), self.mem.ptr(mb), self.mem.ptr(nb)), "gz_seg_integer_or_not")
        self.sync()
        nv = int(np.frombuffer(self.mem.download(nb, 8), dtype=np.uint64)[0])
        return (np.frombuffer(self.mem.download(so, 4 * n), dtype=np.uint32), np.frombuffer(self.mem.download(sl, 4 * n), dtype=np.uint32),
                np.frombuffer(self.mem.download(vb, 8 * nv), dtype=np.int64), np.frombuffer(self.mem.download(mb, nv), dtype=np.uint8))

    def local_generate_partial(self, ltype, raw_native_le, rows, cols, missing, to_file=True):
        """dyn_int_transpose's partial case: present elements row-major -> file order, column-major (and back)"""
        import numpy as np
        buf = self.mem.upload(raw_native_le)
        scratch = self.mem.alloc(len(raw_native_le) + 16)
        mb = self.mem.upload(np.ascontiguousarray(missing, dtype=np.uint8))
        w = {2: 1, 4: 2, 6: 4, 28: 1, 29: 2, 30: 4}[ltype]
        f = self.L.gz_local_generate_partial if to_file else self.L.gz_local_partial_to_native
        lt = self._check(f(self.h, ltype, self.mem.ptr(buf), len(raw_native_le) // w, rows, cols, self.mem.ptr(mb), self.mem.ptr(scratch)), "gz_local_generate_partial")
        return lt, self.mem.download(buf, len(raw_native_le))

    # ---- CODEC_ACGT pre-transform (codec_acgt.c) ----------------------------------------------------------
    def acgt_pack(self, seq, in_place=False):
        """SEQ bytes -> (2-bit packed bytes, exception stream, has_x)"""
        n = len(seq)
        sbuf = self.mem.upload(seq) if n else self.mem.alloc(16)
        pl = self.L.gz_acgt_packed_len(n)
        pbuf = self.mem.alloc(pl + 16)
        xbuf = sbuf if in_place else self.mem.alloc(n + 16)
        has_x = C.c_int(0)
        self._check(self.L.gz_acgt_pack(self.h, self.mem.ptr(sbuf), n, self.mem.ptr(pbuf), self.mem.ptr(xbuf), C.byref(has_x)), "gz_acgt_pack")
        return self.mem.download(pbuf, pl), self.mem.download(xbuf, n), bool(has_x.value)

    def acgt_unpack(self, packed, x, n):
        pbuf = self.mem.upload(packed) if len(packed) else self.mem.alloc(16)
        xbuf = self.mem.upload(x) if x is not None and n else None
        out = self.mem.alloc(n + 16)
        self._check(self.L.gz_acgt_unpack(self.h, self.mem.ptr(pbuf), self.mem.ptr(xbuf) if xbuf is not None else None, n, self.mem.ptr(out)), "gz_acgt_unpack")
        return self.mem.download(out, n)

    # ---- the VBlock compute driver (zip_compress_one_vb for a batch of FASTQ VBlocks) ------------------------
    def zip_open(self, plan):
        """plan: dict as genozip_amd.fastq.illumina_plan() returns -> ZipFile"""
        return ZipFile(self, plan)

    # ---- VBlock section writer ----------------------------------------------------------------------------
    def vb_table(self, vblocks):
        """builds the C tables for gz_vb_compress_batch; section.data may be bytes (uploaded here) or a device buffer"""
        n = len(vblocks)
        vtab = (GzVBlock * max(1, n))()
        keep = []
        for i, vb in enumerate(vblocks):
            ns = len(vb.sections)
            stab = (GzSection * max(1, ns))()
            for k, s in enumerate(vb.sections):
                if isinstance(s.data, (bytes, bytearray)):
                    buf = self.mem.upload(s.data)
                    dlen = len(s.data)
                else:
                    buf, dlen = s.data, s.data_len
                keep.append(buf)
                stab[k].data = self.mem.ptr(buf)
                stab[k].data_len = dlen
                stab[k].data_len_dev = self.mem.ptr(s.data_len_dev) if s.data_len_dev is not None else None
                stab[k].section_type = s.section_type
                stab[k].codec = s.codec
                stab[k].sub_codec = s.sub_codec
                stab[k].flags = s.flags
                stab[k].ltype = s.ltype
                stab[k].param = s.param
                stab[k].b250_size_or_nothing_char = s.byte30
                stab[k].dict_id = (C.c_uint8 * 8)(*s.dict_id)
            cap = self.L.gz_vb_z_bound(stab, ns)
            vb.z = self.mem.alloc(cap + 16)
            vtab[i].vblock_i = vb.vblock_i
            vtab[i].recon_size = vb.recon_size
            vtab[i].longest_line_len = vb.longest_line_len
            vtab[i].longest_seq_len = vb.longest_seq_len
            vtab[i].digest = (C.c_uint8 * 16)(*vb.digest)
            vtab[i].vb_flags = vb.vb_flags
            vtab[i].sections = stab
            vtab[i].n_sections = ns
            vtab[i].z_data = self.mem.ptr(vb.z)
            vtab[i].z_cap = cap
            keep.append(stab)
        return vtab, keep

    def vb_compress_table(self, vtab, n):
        self._check(self.L.gz_vb_compress_batch(self.h, vtab, n), "gz_vb_compress_batch")

    def vb_compress(self, vblocks):
        """zip_compress_one_vb's section phase for a batch of VBlocks; returns list of z_data bytes"""
        vtab, keep = self.vb_table(vblocks)
        self.vb_compress_table(vtab, len(vblocks))
        self.sync()
        res = []
        for i, vb in enumerate(vblocks):
            if vtab[i].status != GZ_OK:
                raise GenozipAMDError("vblock %d: status %d" % (vb.vblock_i, vtab[i].status))
            vb.z_len = vtab[i].z_len
            res.append(self.mem.download(vb.z, vb.z_len))
        return res

    def vb_uncompress(self, z_bytes, total_uncompressed, max_sections=4096):
        """walks one VBlock's z_data; returns the list of decoded section payloads"""
        zb = self.mem.upload(z_bytes)
        ob = self.mem.alloc(total_uncompressed + 16)
        offs = (C.c_uint64 * (max_sections + 1))()
        ns = C.c_uint32(0)
        self._check(self.L.gz_vb_uncompress(self.h, self.mem.ptr(zb), len(z_bytes), self.mem.ptr(ob), total_uncompressed, offs, max_sections, C.byref(ns)), "gz_vb_uncompress")
        raw = self.mem.download(ob, total_uncompressed)
        M = (1 << 63) - 1                            # (GZ_SECTION_NOT_DECODED: a host coder's section - None)
        return [None if offs[i] >> 63 else raw[offs[i] & M:offs[i + 1] & M] for i in range(ns.value)]

    def vb_uncompress_many(self, items, max_sections=4096, download=True):
        """items: [(z bytes or a device buffer of self.mem with its length, total_uncompressed)] -> per VBlock the list of decoded section
        payloads, None for a section the device leaves to the host's coders (download=False: (device buffer, offsets, [decoded? per section])
        per VBlock - a host coder's stretch is zeros on the device). One gz_vb_uncompress_many call: every section of every VBlock in one batch."""
        n = len(items)
        zbs, obs = [], []
        zp, zl, op, oc = (C.c_void_p * max(1, n))(), (C.c_uint64 * max(1, n))(), (C.c_void_p * max(1, n))(), (C.c_uint64 * max(1, n))()
        for i, (z, total) in enumerate(items):
            if isinstance(z, (bytes, bytearray, memoryview)):
                zb, ln = self.mem.upload(bytes(z)), len(z)
            else:
                zb, ln = z
            ob = self.mem.alloc(total + 16)
            zbs.append(zb); obs.append(ob)
            zp[i], zl[i], op[i], oc[i] = self.mem.ptr(zb), ln, self.mem.ptr(ob), total
        offs = (C.c_uint64 * (max(1, n) * (max_sections + 1)))()
        ns = (C.c_uint32 * max(1, n))()
        self._check(self.L.gz_vb_uncompress_many(self.h, n, zp, zl, op, oc, offs, max_sections, ns), "gz_vb_uncompress_many")
        res = []
        for i, (_z, total) in enumerate(items):
            o = [offs[i * (max_sections + 1) + k] for k in range(ns[i] + 1)]
            M = (1 << 63) - 1                        # (GZ_SECTION_NOT_DECODED: a host coder's section - zeros on the device, None here)
            if not download:
                res.append((obs[i], [x & M for x in o], [not (x >> 63) for x in o[:-1]]))
                continue
            raw = self.mem.download(obs[i], total)
            res.append([None if o[k] >> 63 else raw[o[k] & M:o[k + 1] & M] for k in range(ns[i])])
        return res
