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


class Engine:
    def __init__(self, device=0, lib_path=None, mem=None, hip_stream=None):
        self.L = _lib.load(lib_path)
        if mem is None:
            from .mem import TorchMem
            mem = TorchMem(device)
        self.mem = mem
        err = C.c_int(0)
        self.h = self.L.gz_create(device, hip_stream, C.byref(err))
        if not self.h:
            raise GenozipAMDError("gz_create failed (%d): no usable GPU / HIP runtime - genozip_amd has no CPU fallback" % err.value)

    def close(self):
        if getattr(self, "h", None):
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

    def sync(self):
        return self._check(self.L.gz_sync(self.h), "gz_sync")

    def profile(self, enable=True, reset=False):
        self.L.gz_profile(self.h, int(enable), int(reset))

    def profile_results(self):
        """{kernel name: (total ms, launches)} accumulated by gz_sync() since the last reset"""
        out, i = {}, 0
        name = C.create_string_buffer(64)
        ms, n = C.c_double(0), C.c_int(0)
        while self.L.gz_profile_get(self.h, i, name, 64, C.byref(ms), C.byref(n)):
            out[name.value.decode()] = (ms.value, n.value)
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

    # ---- context engine -------------------------------------------------------------------------------------
    def b250_generate_many(self, jobs):
        """jobs: list of (seg bytes, ol_nodes_len, node2word list) -> list of PIZ-format bytes"""
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
        self._check(self.L.gz_b250_generate_batch(self.h, tab, n), "gz_b250_generate_batch")
        self.sync()
        res = []
        for i in range(n):
            ln = int(np.frombuffer(self.mem.download(lens[i], 4), dtype=np.uint32)[0])
            res.append(self.mem.download(outs[i], ln))
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
        return [raw[offs[i]:offs[i + 1]] for i in range(ns.value)]
