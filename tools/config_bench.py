#!/usr/bin/env python3
"""GPU: BASELINE.json's BAM and VCF configurations (configs[2], configs[3]) through the path on ONE GPU - the streams of
SURVEY 8(0) at the sizes the reference would give them, every VBlock of this GPU's share in one batch - with the
reference's own codec code timed on the host cores beside it and every section payload compared.

    python tools/config_bench.py bam [--reads 1000000]      22 VBlocks of 46 000 reads: CIGAR / FLAG / MAPQ b250,
                                                             binned QUAL local, POS u32 local
    python tools/config_bench.py vcf [--vbs 4]               4 VBlocks (one GPU's share of 33 at 8 GPUs) of 3 000 lines x
                                                             10 000 samples: FORMAT/DP u8 matrix (transposed), FORMAT/PL b250

Phases are timed separately (HIP events of the library): "generate" = b250_zip_generate + zip_generate_local
(+ transpose) of every context, "compress" = codec_compress + section writer of every VBlock. Prints one JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

import bench                                                             # noqa: E402  (cpu_baseline, varl_seg)
from genozip_amd import synth, workload as W                            # noqa: E402
from genozip_amd.codec import Engine, Section, VBlock                   # noqa: E402
from genozip_amd.lib import SEC_B250, SEC_LOCAL, LT_BLOB, LT_UINT8, LT_UINT32, CODEC_NAMES  # noqa: E402

def seg_b250(nodes, ol):
    """seg-time b250 (src/b250.c:112-163, 82-107): little-endian VARL, tag in the last byte; nodes new to the VB are 4 bytes"""
    v = np.asarray(nodes, dtype=np.int64)
    new = v >= ol
    ln = np.where(new, 4, np.where(v < 0, 2, np.where(v <= 126, 1, np.where(v <= 16508, 2, np.where(v <= 2113660, 3, 4)))))
    code = np.where(new, (7 << 29) | v, np.where(v == -3, 0xBFFE, np.where(v == -4, 0xBFFF, np.where(v <= 126, v,
                    np.where(v <= 16508, (2 << 14) | (v - 127), np.where(v <= 2113660, (6 << 21) | (v - 16509), (7 << 29) | v))))))
    off = np.cumsum(ln) - ln
    out = np.zeros(int(ln.sum()), dtype=np.uint8)
    for b in range(4):
        m = ln > b
        out[off[m] + b] = (code[m] >> (8 * b)) & 255
    return out.tobytes()


def bam_vblocks(n_reads, per_vb=46000):
    """per VBlock: [(name, kind, payload, extra)]"""
    out = []
    for v, r0 in enumerate(range(0, n_reads, per_vb)):
        n = min(per_vb, n_reads - r0)
        h = synth.u32(1000 + v, n)
        cigar = np.where(h % np.uint32(100) < np.uint32(90), 0, 1 + h % np.uint32(37)).astype(np.int32)       # 90 % "150M"
        flag = (synth.u32(2000 + v, n) % np.uint32(6)).astype(np.int32)
        mapq = np.where(synth.u32(3000 + v, n) % np.uint32(10) < np.uint32(8), 0, synth.u32(3500 + v, n) % np.uint32(40)).astype(np.int32)
        qual = W.quality_rows(W._NP, 4000 + v, 0, n, "bin").reshape(-1).astype(np.uint8).tobytes()
        pos = (10000 + r0 * 20 + np.cumsum(synth.u32(5000 + v, n) % np.uint32(40))).astype("<u4").tobytes()
        out.append([("CIGAR", "b250", seg_b250(cigar, 30), (30, [30 + k for k in range(8)])),
                    ("FLAG", "b250", seg_b250(flag, 6), (6, [])), ("MAPQ", "b250", seg_b250(mapq, 40), (40, [])),
                    ("QUAL", "blob", qual, None), ("POS", "u32", pos, None)])
    return out, n_reads * 330          # ~ bytes of SAM text per read (for orientation only)


def vcf_vblocks(n_vb, rows=3000, cols=10000):
    out = []
    for v in range(n_vb):
        h = synth.u32(77 + v, rows * cols)
        dp = (18 + (h % np.uint32(13)) + ((h >> np.uint32(8)) % np.uint32(13))).astype(np.uint8)
        dp[(h >> np.uint32(20)) % np.uint32(97) == 0] = 0
        n = rows * cols
        h = synth.u32(178 + v, n)
        ol, new = 3000, 1200
        ni = np.where(h % np.uint32(10) < np.uint32(8), h % np.uint32(100), h % np.uint32(ol + new)).astype(np.int32)
        ni[(h >> np.uint32(16)) % np.uint32(1000) == 0] = -3
        n2w = [int(x) for x in (synth.u32(279 + v, new) % np.uint32(ol + new + 40))]
        out.append([("DP", "u8tr", dp.tobytes(), cols), ("PL", "b250", seg_b250(ni, ol), (ol, n2w))])
    return out, n_vb * rows * cols * 16   # ~ 16 bytes of VCF text per sample cell


class Workload:
    """device-resident inputs + the C tables of one step (built once); mirrors bench.RankWorkload"""

    def __init__(self, E, vbs, device):
        from genozip_amd.lib import GzB250Job
        self.E, self.device = E, device
        mem = E.mem
        self.b250, self.locals = [], []
        self.meta = []
        for v, streams in enumerate(vbs):
            for name, kind, payload, extra in streams:
                if kind == "b250":
                    ol, n2w = extra
                    self.b250.append(dict(v=v, name=name, seg=mem.upload(payload), seg_len=len(payload), ol=ol,
                                          n2w=mem.upload(np.asarray(n2w or [0], dtype=np.int32)), n_new=len(n2w), out=mem.alloc(len(payload) + 16)))
                else:
                    raw = mem.upload(payload)
                    self.locals.append(dict(v=v, name=name, kind=kind, raw=raw, work=torch.empty_like(raw), scratch=torch.empty_like(raw),
                                            n=len(payload), cols=extra or 0))
        self.b250_len = torch.zeros(max(1, len(self.b250)), dtype=torch.int32, device=device)
        self.b250_tab = (GzB250Job * max(1, len(self.b250)))()
        for i, j in enumerate(self.b250):
            t = self.b250_tab[i]
            t.seg, t.seg_len, t.ol_nodes_len = mem.ptr(j["seg"]), j["seg_len"], j["ol"]
            t.node2word, t.n_new_nodes = mem.ptr(j["n2w"]), j["n_new"]
            t.out, t.out_len_dev = mem.ptr(j["out"]), self.b250_len.data_ptr() + 4 * i
        self.n_vb = len(vbs)
        self.stream_bytes = sum(j["seg_len"] for j in self.b250) + sum(j["n"] for j in self.locals)

    def generate(self):
        E = self.E
        if self.b250:
            E._check(E.L.gz_b250_generate_batch(E.h, self.b250_tab, len(self.b250)), "b250_generate_batch")
        for j in self.locals:
            j["work"].copy_(j["raw"], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        for j in self.locals:
            if j["kind"] == "u32":
                E._check(E.L.gz_local_generate(E.h, LT_UINT32, j["work"].data_ptr(), j["n"] // 4, 0, None), "local_generate")
            elif j["kind"] == "u8tr":
                j["lt"] = E._check(E.L.gz_local_generate(E.h, LT_UINT8, j["work"].data_ptr(), j["n"], j["cols"], j["scratch"].data_ptr()), "local_generate")

    def assign(self):
        """codec_assign_best_codec on the first VBlock's streams (committed for the later ones, codec.c:352-363)"""
        E = self.E
        self.generate()
        E.sync()
        lens = self.b250_len.cpu().numpy()
        self.codecs = {}
        for i, j in enumerate(self.b250):
            if j["v"] == 0:
                self.codecs[j["name"]] = E._check(E.L.gz_codec_assign_best(E.h, E.mem.ptr(j["out"]), int(lens[i]), None), "assign") or 6
        for j in self.locals:
            if j["v"] == 0:
                self.codecs[j["name"]] = E._check(E.L.gz_codec_assign_best(E.h, j["work"].data_ptr(), j["n"], None), "assign") or 6
        return self.codecs

    def build(self):
        E = self.E
        per_vb = [[] for _ in range(self.n_vb)]
        for j in self.locals:
            lt = {"blob": LT_BLOB, "u32": LT_UINT32, "u8tr": j.get("lt", LT_UINT8)}[j["kind"]]
            per_vb[j["v"]].append(Section(j["work"], SEC_LOCAL, self.codecs[j["name"]], j["name"].encode(), ltype=lt, byte30=0xff,
                                          param=0, data_len=j["n"]))
        for i, j in enumerate(self.b250):
            per_vb[j["v"]].append(Section(j["out"], SEC_B250, self.codecs[j["name"]], j["name"].encode(), byte30=4,
                                          data_len=j["seg_len"], data_len_dev=self.b250_len[i:i + 1]))
        self.vbs = [VBlock(v + 1, secs) for v, secs in enumerate(per_vb)]
        self.vtab, self._keep = E.vb_table(self.vbs)

    def compress(self):
        self.E.vb_compress_table(self.vtab, len(self.vbs))


def cpu_baseline(z_list, n_threads):
    """the reference's own rANS / arith code (oracle/_ref, or this repo's restatement where that was not built) over the same section
    payloads on a pthread pool: decode what the GPU wrote, encode it again (timed), compare byte for byte"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    kind = "reference" if pyoracle.Ref.available() else "port"
    R = pyoracle.Ref() if kind == "reference" else pyoracle.Oracle()
    codecs, datas, pays = [], [], []
    for z in z_list:
        for st, codec, did, ulen, pay, _domq in bench.walk_sections(z):
            if codec == 1:
                continue
            data = R.codec_uncompress(codec, pay, ulen) if kind == "port" else R.hts_uncompress("rans" if codec < 16 else "arith", pay, ulen)
            codecs.append(codec); datas.append(data); pays.append(bytes(pay))
    if kind == "reference":
        R.codec_compress_many(codecs[:8], datas[:8], 8)
        outs, dt = R.codec_compress_many(codecs, datas, n_threads)
    else:
        t0 = time.time(); outs = R.codec_compress_many(codecs, datas, n_threads); dt = time.time() - t0
    nbytes = sum(len(d) for d in datas)
    return {"value": round(nbytes / dt / 1e6, 1), "unit": "MB/s of context streams", "cores": n_threads, "kind": kind,
            "sample": "codec calls only: all %d coded sections (%.0f MB), %d threads on %d logical CPUs" % (len(datas), nbytes / 1e6, n_threads, os.cpu_count())}, \
        all(o == p for o, p in zip(outs, pays))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=("bam", "vcf"))
    ap.add_argument("--reads", type=int, default=1000000)
    ap.add_argument("--vbs", type=int, default=4)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    E = Engine(device=0)
    vbs, text_bytes = bam_vblocks(a.reads) if a.config == "bam" else vcf_vblocks(a.vbs)
    wl = Workload(E, vbs, device)
    codecs = wl.assign()
    wl.build()
    for _ in range(1):
        wl.generate(); wl.compress(); E.sync()
    t_gen = t_cmp = 0.0
    E.profile(True, reset=True)
    for _ in range(a.steps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        wl.generate(); E.sync()
        t1 = time.perf_counter()
        wl.compress(); E.sync()
        t2 = time.perf_counter()
        t_gen += t1 - t0; t_cmp += t2 - t1
    E.profile(False)
    prof = E.profile_results()
    ms_gen, ms_cmp = t_gen / a.steps * 1e3, t_cmp / a.steps * 1e3
    z_list = [E.mem.download(vb.z, int(wl.vtab[i].z_len)) for i, vb in enumerate(wl.vbs)]
    out = {"config": "BAM-1M (configs[2])" if a.config == "bam" else "VCF 10k samples, %d VBlocks of 3000 lines (configs[3], one GPU's share)" % a.vbs,
           "n_vblocks": wl.n_vb, "stream_mb": round(wl.stream_bytes / 1e6, 1), "text_mb_approx": round(text_bytes / 1e6),
           "codecs": {k: CODEC_NAMES[v] for k, v in codecs.items()},
           "ms_generate": round(ms_gen, 2), "ms_compress": round(ms_cmp, 2), "ms_per_step": round(ms_gen + ms_cmp, 2),
           "value": round(wl.stream_bytes / 1e6 / ((ms_gen + ms_cmp) / 1e3), 1), "unit": "MB/s of context streams",
           "compressed_mb": round(sum(len(z) for z in z_list) / 1e6, 2),
           "kernels_ms_per_step": {k: round(v[0] / a.steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:12]},
           "launches_per_step": {k: v[1] / a.steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:4]},
           "vb1_sections": [(s.dict_id.rstrip(b"\0").decode(), CODEC_NAMES[s.codec], s.data_len) for s in wl.vbs[0].sections]}
    if not a.no_cpu:
        cb, exact = cpu_baseline(z_list, min(os.cpu_count() or 1, 256))
        out["cpu_baseline"] = cb
        out["bit_exact"] = exact
    print(json.dumps(out))


if __name__ == "__main__":
    main()
