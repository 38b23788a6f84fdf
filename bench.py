#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: input MB/s of the Genozip context entropy-coding hot path
(b250_zip_generate + zip_generate_local + codec_compress + section writer) over the synthetic FASTQ-PE-1M workload.

    python bench.py [--gpus N --steps K --warmup W]      (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over the whole batch of VBlocks of this rank's FASTQ pair (inputs resident in
HBM when the timed region starts; outputs stay in HBM, N>1 additionally gathers the compressed VBlocks to rank 0 over
RCCL). Prints ONE JSON line (rank 0). See DESIGN.md section "Measurement".
"""
import argparse
import concurrent.futures
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402  (first: one HIP runtime per process, see genozip_amd/lib.py)

from genozip_amd import workload as W                                   # noqa: E402
from genozip_amd.codec import Engine, Section, VBlock                   # noqa: E402
from genozip_amd.lib import (SEC_B250, SEC_LOCAL, LT_BLOB, LT_UINT16, LT_UINT32, CODEC_NAMES, CODEC_RANB)  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def varl_seg(nodes, ol_nodes_len):
    """seg-time b250 (src/b250.c:112-163): little-endian VARL, tag in the last byte; nodes new to the VB are 4 bytes"""
    nodes = np.asarray(nodes, dtype=np.int64)
    out = bytearray()
    for v in nodes.tolist():
        if v >= ol_nodes_len:
            out += int((7 << 29) | v).to_bytes(4, "little")
        elif v <= 126:
            out.append(v)
        elif v <= 16508:
            out += int((2 << 14) | (v - 127)).to_bytes(2, "little")
        else:
            out += int((6 << 21) | (v - 16509)).to_bytes(3, "little")
    return bytes(out)


class RankWorkload:
    """the VBlocks of one FASTQ pair, resident in HBM, plus the C tables of one step"""

    def __init__(self, E, n_pairs, vb_bytes, profile, seed_base, device):
        self.E = E
        th = W._TH(device)
        ranges = W.vb_ranges(n_pairs, vb_bytes)
        self.n_vb = 2 * len(ranges)
        self.qual = []          # per mate: one tensor with all qualities, VBlocks are slices
        self.text_bytes = 2 * n_pairs * W.RECORD_BYTES
        self.vb_meta = []       # (mate, read0, n_reads)
        TILE_OL = 600           # tiles >= 600 are "new to the VB": exercised node -> word conversion
        b250_jobs, self.sections_src = [], []
        xs, ys = [], []
        for mate in (0, 1):
            seed = seed_base + mate
            q = torch.empty(n_pairs * W.READ_LEN, dtype=torch.uint8, device=device)
            CH = 200000
            for r0 in range(0, n_pairs, CH):
                n = min(CH, n_pairs - r0)
                q[r0 * W.READ_LEN:(r0 + n) * W.READ_LEN] = W.quality_rows(th, seed, r0, n, profile)
            self.qual.append(q)
            lane, tile, x, y = W.name_fields(seed, 0, n_pairs)
            for (r0, n) in ranges:
                self.vb_meta.append((mate, r0, n))
                b250_jobs.append((varl_seg(lane[r0:r0 + n], 4), 4, []))
                b250_jobs.append((varl_seg(tile[r0:r0 + n], TILE_OL), TILE_OL, list(range(TILE_OL, 624))))
                xs.append(x[r0:r0 + n])
                ys.append(y[r0:r0 + n])
        mem = E.mem
        # b250: seg-format inputs, node2word maps, outputs and device-resident lengths
        self.b250_in = [mem.upload(j[0]) for j in b250_jobs]
        self.b250_n2w = [mem.upload(np.asarray(j[2] or [0], dtype=np.int32)) for j in b250_jobs]
        self.b250_out = [mem.alloc(len(j[0]) + 16) for j in b250_jobs]
        self.b250_len = torch.zeros(len(b250_jobs), dtype=torch.int32, device=device)
        from genozip_amd.lib import GzB250Job
        self.b250_tab = (GzB250Job * len(b250_jobs))()
        for i, j in enumerate(b250_jobs):
            t = self.b250_tab[i]
            t.seg, t.seg_len, t.ol_nodes_len = mem.ptr(self.b250_in[i]), len(j[0]), j[1]
            t.node2word, t.n_new_nodes = mem.ptr(self.b250_n2w[i]), len(j[2])
            t.out, t.out_len_dev = mem.ptr(self.b250_out[i]), self.b250_len.data_ptr() + 4 * i
        self.b250_jobs = b250_jobs
        # x / y locals: raw native copies + working buffers (zip_generate_local works in place)
        self.x_raw = mem.upload(np.concatenate(xs).astype("<u2"))
        self.y_raw = mem.upload(np.concatenate(ys).astype("<u4"))
        self.x_work, self.y_work = torch.empty_like(self.x_raw), torch.empty_like(self.y_raw)
        self.x_off = np.concatenate([[0], np.cumsum([2 * len(a) for a in xs])])
        self.y_off = np.concatenate([[0], np.cumsum([4 * len(a) for a in ys])])
        self.n_x, self.n_y = sum(len(a) for a in xs), sum(len(a) for a in ys)
        self.codecs = None
        self.vtab = None

    def assign_codecs(self):
        """codec_assign_best_codec on the first VBlock's streams, committed for all later VBlocks (src/codec.c:352-363)"""
        E = self.E
        mate, r0, n = self.vb_meta[0]
        q = self.qual[mate][r0 * W.READ_LEN:(r0 + n) * W.READ_LEN]
        self.step_prepare()
        E.sync()
        lens = self.b250_len.cpu().numpy()
        res = {}
        for name, ptr, ln in (("QUAL", q.data_ptr(), q.numel()), ("Q1NAME", E.mem.ptr(self.b250_out[0]), int(lens[0])),
                              ("Q2NAME", E.mem.ptr(self.b250_out[1]), int(lens[1])),
                              ("Q3NAME", self.x_work.data_ptr(), int(self.x_off[1])), ("Q4NAME", self.y_work.data_ptr(), int(self.y_off[1]))):
            c = E._check(E.L.gz_codec_assign_best(E.h, ptr, ln, None), "assign_best")
            res[name] = c or CODEC_RANB      # UNKNOWN (< 50 bytes) -> RANB fallback of the section writer
        self.codecs = res
        return res

    def build_tables(self):
        E = self.E
        vbs = []
        for v, (mate, r0, n) in enumerate(self.vb_meta):
            q = self.qual[mate][r0 * W.READ_LEN:(r0 + n) * W.READ_LEN]
            paired = 0x04
            secs = [Section(q, SEC_LOCAL, self.codecs["QUAL"], b"QUAL", ltype=LT_BLOB, flags=paired, data_len=q.numel()),
                    Section(self.x_work[int(self.x_off[v]):int(self.x_off[v + 1])], SEC_LOCAL, self.codecs["Q3NAME"], b"Q3NAME", ltype=LT_UINT16,
                            byte30=0xff, data_len=int(self.x_off[v + 1] - self.x_off[v])),
                    Section(self.y_work[int(self.y_off[v]):int(self.y_off[v + 1])], SEC_LOCAL, self.codecs["Q4NAME"], b"Q4NAME", ltype=LT_UINT32,
                            byte30=0xff, data_len=int(self.y_off[v + 1] - self.y_off[v])),
                    Section(self.b250_out[2 * v], SEC_B250, self.codecs["Q1NAME"], b"Q1NAME", byte30=4, flags=paired,
                            data_len=len(self.b250_jobs[2 * v][0]), data_len_dev=self.b250_len[2 * v:2 * v + 1]),
                    Section(self.b250_out[2 * v + 1], SEC_B250, self.codecs["Q2NAME"], b"Q2NAME", byte30=4, flags=paired,
                            data_len=len(self.b250_jobs[2 * v + 1][0]), data_len_dev=self.b250_len[2 * v + 1:2 * v + 2])]
            vbs.append(VBlock(v + 1, secs, recon_size=n * W.RECORD_BYTES, longest_line_len=W.READ_LEN + 1, longest_seq_len=W.READ_LEN))
        self.vbs = vbs
        self.vtab, self._keep = E.vb_table(vbs)
        # bytes entering the path: every local as handed over by seg + every b250 in its seg-time form
        self.stream_bytes = sum(s.data_len for vb in vbs for s in vb.sections if s.section_type == SEC_LOCAL) \
            + sum(len(j[0]) for j in self.b250_jobs)

    def step_prepare(self):
        """b250_zip_generate for every b250 context + zip_generate_local for every integer local, whole batch at once"""
        E = self.E
        E._check(E.L.gz_b250_generate_batch(E.h, self.b250_tab, len(self.b250_jobs)), "b250_generate_batch")
        self.x_work.copy_(self.x_raw, non_blocking=True)
        self.y_work.copy_(self.y_raw, non_blocking=True)
        torch.cuda.current_stream().synchronize()       # the copies run on torch's stream, the library on its own
        E._check(E.L.gz_local_generate(E.h, LT_UINT16, self.x_work.data_ptr(), self.n_x, 0, None), "local_generate")
        E._check(E.L.gz_local_generate(E.h, LT_UINT32, self.y_work.data_ptr(), self.n_y, 0, None), "local_generate")

    def step(self):
        self.step_prepare()
        self.E.vb_compress_table(self.vtab, len(self.vbs))
        self.E.sync()


def cpu_baseline(rank_wl, z_list, n_threads):
    """the reference's own rANS/arith code (oracle/_ref: src/htscodecs compiled in place) - or, where that was not
    built, this repo's C restatement - over the SAME section payloads on the host cores; also the bit-exactness check"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    kind = "reference" if pyoracle.Ref.available() else "port"
    impl = pyoracle.Ref() if kind == "reference" else pyoracle.Oracle()
    O = pyoracle.Oracle()
    E = rank_wl.E
    tasks = []
    for vb in rank_wl.vbs:
        for s in vb.sections:
            n = s.data_len
            if s.data_len_dev is not None:
                n = int(s.data_len_dev.cpu().numpy()[0])
            tasks.append((s.codec, E.mem.download(s.data, n)))
    if kind == "reference":
        impl.codec_compress_many([t[0] for t in tasks[:8]], [t[1] for t in tasks[:8]], 8)            # warm up
        outs, dt = impl.codec_compress_many([t[0] for t in tasks], [t[1] for t in tasks], n_threads)  # C pthread pool
    else:
        t0 = time.time()
        outs = O.codec_compress_many([t[0] if len(t[1]) >= 50 else 1 for t in tasks], [t[1] for t in tasks], n_threads)
        dt = time.time() - t0
    nbytes = sum(len(d) for _, d in tasks)
    # bit-exactness: every section payload produced on the GPU == the CPU reference's
    exact, k = True, 0
    for z in z_list:
        at = 84
        while at < len(z):
            clen = int.from_bytes(z[at + 12:at + 16], "big")
            exact &= z[at + 40:at + 40 + clen] == outs[k]
            k += 1
            at += 40 + clen
    exact &= k == len(outs)
    return {"value": round(nbytes / dt / 1e6, 1), "unit": "MB/s", "cores": n_threads, "kind": kind,
            "sample": "all %d sections of rank 0's %d VBlocks (%.0f MB), codec calls only, %d threads on %d logical CPUs" %
                      (len(tasks), len(rank_wl.vbs), nbytes / 1e6, n_threads, os.cpu_count())}, bool(exact)


def seg_front_probe(E, device, target_mb=1536):
    """SURVEY 8f N1 + rows a1-a3 on FASTQ text resident in HBM - NOT part of `value` (the metric is quoted on the context
    streams); reported beside it because these kernels, unlike the coders, are bandwidth-bound: lines -> reads -> qname
    tokens -> per-token columns (node indices, dict, b250) + SEQ / QUAL gathered into their locals. HIP events of the
    library; `nl_scan` is the newline count over the whole text against the HBM roof."""
    n_reads = 20000
    text, rec = W.fastq_text(1, 0, n_reads)
    reps = max(1, (target_mb << 20) // len(text))
    block = torch.from_numpy(np.frombuffer(text, dtype=np.uint8).copy()).to(device)
    big = block.repeat(reps)
    E.sync()
    E.profile(True, reset=True)
    tb, ob, lb, rb, n_lines = E.text_lines(big, cap=4 * n_reads * reps + 8, on_device=True)      # the whole text
    bad, cols = E.fastq_records(text)                                                             # one VBlock's worth: reads,
    (qo, ql), (so, sl), _, (uo, ul) = cols
    nb, io, il = E.tokenize_column(block, qo, ql, b":::::: ")                                     # tokens,
    n_vb = 64
    E.ctx_seg_columns([(block, io[i], il[i], []) for i in range(io.shape[0])] * n_vb, keep_on_device=True)   # columns of 64 VBlocks in one call,
    E.local_blob_columns([(block, so, sl, False), (block, uo, ul, False)] * n_vb)                 # SEQ / QUAL -> locals
    E.profile(False)
    pr = E.profile_results()
    ms = {k: round(v[0], 3) for k, v in sorted(pr.items(), key=lambda kv: -kv[1][0])}
    nl = pr.get("k_nl_count", (0, 0))[0]
    gbs = big.numel() / (nl / 1e3) / 1e9 if nl else None
    assert n_lines == 4 * n_reads * reps and bad is None and nb == 0
    return {"text_mb": round(big.numel() / 1e6, 1), "columns": io.shape[0] * n_vb, "snips_per_column": n_reads,
            "kernels_ms": ms, "nl_scan": {"bound": "hbm", "achieved": round(gbs, 1) if gbs else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": round(gbs / HBM_PEAK_GBS, 4) if gbs else None}}


def per_launch(per_step, launches_per_step):
    return None if per_step is None else int(per_step / launches_per_step)


def pmc_traffic(kernel, a):
    """HBM bytes per STEP of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/round1_pmc.json, made by tools/summarize_prof.py); null for non-default workloads"""
    p = os.path.join(ROOT, "profiles", "round1_pmc.json")
    if not os.path.exists(p) or a.pairs != 1000000 or a.vb_mb != 4 or a.qual != "div":
        return None
    k = json.load(open(p))["kernels"].get(kernel)
    return k.get("traffic_bytes_per_step", k["traffic_bytes"]) if k else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=1000000, help="read pairs per rank (BASELINE configs[1]: 1 M)")
    ap.add_argument("--vb-mb", type=int, default=4, help="VBlock size in MiB (reference: --vblock; its small-file rule floors at 4)")
    ap.add_argument("--qual", default="div", choices=("div", "bin"))
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-seg-front", action="store_true", help="skip the (untimed) probe of the seg-side kernels")
    ap.add_argument("--pin-codecs", action="store_true", help="skip codec_assign_best and use the codecs it picks for this workload (profiling runs: every launch of a kernel is then a timed-region launch)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL over xGMI
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    E = Engine(device=local_rank)
    wl = RankWorkload(E, a.pairs, a.vb_mb << 20, a.qual, seed_base=1 + 2 * rank, device=device)
    PINNED = {"div": {"QUAL": 16, "Q1NAME": 8, "Q2NAME": 16, "Q3NAME": 17, "Q4NAME": 17},
              "bin": {"QUAL": 18, "Q1NAME": 8, "Q2NAME": 16, "Q3NAME": 17, "Q4NAME": 17}}
    if a.pin_codecs:
        wl.codecs = codecs = dict(PINNED[a.qual])
        wl.step_prepare()
        E.sync()
    else:
        codecs = wl.assign_codecs()
    wl.build_tables()

    pending = [None]

    def gather_wait():
        if pending[0] is not None:
            pending[0].wait()
            pending[0] = None

    def gather_to_rank0():
        # the only exchange step of the path: compressed VBlocks go to the writer rank. The payload is
        # packed into a staging buffer and the gather only STARTED: the transfer over xGMI runs beside the next step's
        # kernels; every gather is complete before the timed region ends (gather_wait below).
        if world == 1:
            return
        from genozip_amd.shard import gather_blobs
        gather_wait()
        blobs = [vb.z[:int(wl.vtab[i].z_len)] for i, vb in enumerate(wl.vbs)]
        if os.environ.get("GZ_SYNC_GATHER"):
            gather_blobs(dist, blobs, rank, world, device)
        else:
            pending[0] = gather_blobs(dist, blobs, rank, world, device, async_op=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        wl.step()
        gather_to_rank0()
    gather_wait()
    E.profile(True, reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        wl.step()
        gather_to_rank0()
    gather_wait()
    barrier()
    dt = time.perf_counter() - t0
    E.profile(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = dt / a.steps * 1e3
    stream_mb = wl.stream_bytes / 1e6
    value = world * stream_mb / (ms_per_step / 1e3)
    z_list = [E.mem.download(vb.z, int(wl.vtab[i].z_len)) for i, vb in enumerate(wl.vbs)]
    z_bytes = sum(len(z) for z in z_list)

    # ---- roofline of the dominant kernel, from HIP events on the library's stream
    prof = E.profile_results()
    dom = max(prof, key=lambda k: prof[k][0])
    dom_ms, dom_launches = prof[dom]
    per_step_launches = dom_launches / a.steps
    avg_launch_ms = dom_ms / dom_launches
    alg_bytes_per_step = wl.stream_bytes + z_bytes - 84 * wl.n_vb       # N_in + N_out of every stream
    alg_per_launch = alg_bytes_per_step / per_step_launches
    achieved = alg_per_launch / (avg_launch_ms / 1e3) / 1e9
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": per_launch(pmc_traffic(dom, a), per_step_launches),
                "avg_launch_ms": round(avg_launch_ms, 4), "launches_per_step": per_step_launches,
                "alg_bytes_per_launch": int(alg_per_launch),
                "kernels_ms_per_step": {k: round(v[0] / a.steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}}

    out = {"metric": "input MB/s compressed (bit-exact .genozip) at 1/2/4/8 GPUs vs CPU ref", "value": round(value, 1), "unit": "MB/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": "FASTQ-PE-1M per GPU (2 x %d reads x 150 bp): context streams of %d VBlocks of %d MiB "
                                  "(QUAL local + 2 QNAME-token b250 + 2 QNAME-token int locals) through b250_generate / "
                                  "local_generate / codec_compress / section writer; MB counted = bytes entering the path "
                                  "(%.1f MB per GPU; the FASTQ text they come from is %.0f MB); SEQ (ACGT pack + host LZMA) and the "
                                  "seg-side kernels (measured apart: seg_front) are outside the timed region" % (a.pairs, wl.n_vb, a.vb_mb, stream_mb, wl.text_bytes / 1e6),
                      "qual_profile": a.qual, "vb_mib": a.vb_mb,
                      "codecs": {k: CODEC_NAMES[v] for k, v in codecs.items()}, "compressed_mb": round(z_bytes / 1e6, 2),
                      "parallelism": "vblocks sharded over %d GPU(s), no data-path collective; RCCL gather of z_data" % world},
           "roofline": roofline}
    if world == 1 and not a.no_seg_front:
        out["seg_front"] = seg_front_probe(E, device)
    if not a.no_cpu and world == 1:                # (the CPU pool is timed on rank 0 of the 1-GPU run only)
        cb, exact = cpu_baseline(wl, z_list, min(os.cpu_count() or 1, 256))
        out["cpu_baseline"] = cb
        out["bit_exact"] = exact
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
