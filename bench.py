#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: input MB/s of Genozip's context entropy-coding hot path, from FASTQ text
resident in HBM to finished VBlock z_data, over the synthetic paired-end FASTQ workload (BASELINE configs[1]).

    python bench.py [--gpus N --steps K --warmup W]      N > 1: this script re-launches itself under torch.distributed.run,
                                                         one rank per GPU (or is launched that way by the driver)

A "step" is ONE PASS OVER THE WHOLE FILE PAIR through the library's VBlock compute driver (gz_fastq_zip_vblocks /
its three phases): text -> lines -> reads -> line-1 items -> seg-side columns (a1-a3) -> dictionary merge on the host in C
(a4) -> b250 generation (a5) -> local byte order (a6) -> R2 == R1 drops -> codec assignment on the first VBlock (a8) ->
sections in the reference's order (a15) -> rANS / arithmetic coders and section framing (a9-a13) -> VB headers (a16), a
new file (fresh dictionaries and codecs) every step. Inputs are resident in HBM when the timed region starts; outputs
stay in HBM (N > 1: + the RCCL gather of the compressed VBlocks to the writer rank).

What is NOT in the path: LZMA of the 2-bit packed SEQ - the pack itself (CODEC_ACGT's front) is in the step, the
packed bytes are handed back for the host's LZMA. `value` therefore counts the text WITHOUT the SEQ lines (the bytes this
path compresses to finished sections); text_mb_s (all of the text, SEQ leaving 2-bit packed) and stream_mb_s (bytes entering
the entropy coders) are printed beside it.

--scaling weak (default): every rank compresses its own file pair. --scaling strong: ONE file pair, its VBlock pairs dealt out
over the ranks (genozip_amd/shard.py): the dictionary merge and the codec choices are exchanged, z_data is what one process
would have written. --stream-reads R: BASELINE configs[4] at reduced scale - R read pairs per rank streamed through the same
file in calls of --batch-pairs VBlock pairs (dictionaries carried from call to call).

Prints ONE JSON line (rank 0). See DESIGN.md section "Measurement".
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")    # (genozip_amd/lib.py: must be set before the process's first HIP call)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHAIN_FLOOR_NS = 5.03           # three dependent vector instructions per symbol at the one-instruction-per-4-clocks issue rate of a wave: 12.07 clocks at 2.4 GHz - what no
                                # arrangement of this formulation can beat (tools/probes/chain_regs_probe.py: `p_pad0`, profiles/r06_chain_regs_probe_aligned.txt)
CHAIN_ALONE_NS = 5.41           # k_arith_chain's loop as it is, alone on the device: 13.00 clocks per symbol (profiles/r06_ubench_chain_single.txt; round 5: 6.3, round 4: 6.6)
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
METRIC = "input MB/s compressed (bit-exact .genozip) at 1/2/4/8 GPUs vs CPU ref"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=1000000, help="read pairs of the file (BASELINE configs[1]: 1 M)")
    ap.add_argument("--vb-mb", type=float, default=0, help="VBlock size in MiB (the reference's --vblock). Default 0: the reference's own rule for the file "
                    "(src/segconf.c:152-206, see vb_bytes below): 14.72 MB for a 1 M-read mate file; 16 MiB for --stream-reads")
    ap.add_argument("--qual", default=None, choices=("div", "bin"), help="quality profile (SURVEY 8d): 40-level (default for fastq) or Illumina-binned (default for bam)")
    ap.add_argument("--scaling", default="strong", choices=("weak", "strong"),
                    help="N > 1. strong (default: BASELINE's target is strong scaling of configs[1]): ONE file pair, its VBlock pairs dealt out over the GPUs; weak: a file pair per GPU")
    ap.add_argument("--stream-reads", type=int, default=0, help="stream this many read pairs per rank through one file (configs[4] at reduced scale)")
    ap.add_argument("--batch-pairs", type=int, default=112, help="VBlock pairs per call in --stream-reads mode (112 x 2 x 16 MiB = 3.76 GB: a call takes < 4 GB of text; "
                    "the more VBlocks a call holds, the better the long chains are hidden: 32 -> 5.6 GB/s, 64 -> 10.2, 112 -> 15.5)")
    ap.add_argument("--two-in-flight", action="store_true", help="--stream-reads: two calls in flight (gz_fastq_zip_begin / _end) instead of one at a time. Measured: no gain at 112 pairs "
                    "per call (15.5 -> 14.0 GB/s) - with 224 long streams per call the model kernels already fill the device; it helps small calls")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--pin-codecs", action="store_true", help="hard-code the codecs codec_assign_best_codec picks for this workload (profiling runs: no trial compressions in the step)")
    ap.add_argument("--config", default="fastq", choices=("fastq", "bam", "vcf"), help="fastq: BASELINE configs[1] (the headline, from text). bam: configs[2] from SAM text (genozip_amd/sam.py). vcf: configs[3], one GPU's share, at the "
                    "context-stream level (no VCF segmenter: the streams of SURVEY 8(0) enter generated). Same record layout, cpu_baseline beside it")
    ap.add_argument("--bam-binary", action="store_true", help="--config bam: start from the alignment RECORDS of an uncompressed BAM stream in HBM (N1 for BAM: gz_bam_records + gz_bam_to_sam "
                    "in every step, in front of the SAM plan) instead of from SAM text")
    ap.add_argument("--stream-level", action="store_true", help="--config bam / vcf: enter at the context-stream level (generated streams, tools/config_bench.py) instead of from text")
    ap.add_argument("--vcf-samples", type=int, default=10000)
    ap.add_argument("--vcf-lines", type=int, default=3000, help="data lines per VBlock")
    ap.add_argument("--vcf-vbs", type=int, default=4, help="VBlocks of this GPU (the file's 33 over 8 GPUs)")
    ap.add_argument("--warm-steps", type=int, default=3, help="extra steps with the handle's codec speculation ON, reported beside the (cold) headline; 0: none")
    return ap.parse_args()


def relaunch_under_torchrun(a):
    """`python bench.py --gpus N` on its own: start N ranks (one per GPU) with torch.distributed.run"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env).returncode)


def vb_bytes(a):
    """segconf_set_vb_size (src/segconf.c:152-206) for a plain-text FASTQ mate file on a host with more than 36 threads:
    min (what the number of used contexts gives: 1 MB each, at least 16 MB, doubled for > 36 threads,
         max (4 MB, est_seggable_size * 1.2 / min (est_max_threads = 30 for uncompressed text, threads)))"""
    if a.vb_mb:
        return int(a.vb_mb * (1 << 20))
    if a.stream_reads:
        return 16 << 20
    from genozip_amd import workload as W
    file_bytes = a.pairs * W.RECORD_BYTES                       # txt_file->est_seggable_size of one mate file
    by_contexts = 2 * max(20 << 20, 16 << 20)                   # ~20 contexts with data (the plan's), x 2: global_max_threads > 36
    return min(by_contexts, max(4 << 20, int(file_bytes * 1.2 / 30)))


def walk_sections(z):
    """[(section_type, codec, dict_id, uncompressed_len, payload, through DOMQ)] of one VBlock's z_data (SectionHeaderCtx,
    sections.h:146-167,419-435); codec = the coder of the payload: a CODEC_DOMQ section names it in sub_codec"""
    out, at = [], 84
    while at < len(z):
        clen = int.from_bytes(z[at + 12:at + 16], "big")
        domq = z[at + 25] == 13
        named = domq or z[at + 25] == 11                         # CODEC_DOMQ / CODEC_XCGT name the section, the stream's coder is in sub_codec
        out.append((z[at + 24], z[at + 26] if named else z[at + 25], bytes(z[at + 32:at + 40]), int.from_bytes(z[at + 16:at + 20], "big"), z[at + 40:at + 40 + clen], domq))
        at += 40 + clen
    return out


class Workload:
    """this rank's share of the file pair: the text of its VBlocks in HBM and the VBlock table of one call"""

    def __init__(self, E, a, rank, world, device):
        import torch
        from genozip_amd import workload as W
        from genozip_amd import fastq as fq
        from genozip_amd.shard import pairs_of_rank
        self.E, self.a, self.W = E, a, W
        strong = a.scaling == "strong" and (world > 1 or bool(os.environ.get("GZ_BENCH_FORCE_DIST")))
        seed = 1 if strong else 1 + 2 * rank                     # strong: the one file; weak: a file pair of its own per rank
        n_reads = a.stream_reads or a.pairs
        ranges = W.vb_ranges(n_reads, vb_bytes(a))
        if a.stream_reads:
            ranges = ranges[:a.batch_pairs]                       # one call's worth of text, streamed over and over with new vblock_i
        self.n_pairs_file = len(W.vb_ranges(n_reads, vb_bytes(a)))
        if a.stream_reads:                                        # R1 = 1..N, R2 = N+1..2N (writer.c:318-322), N = whole calls; call c holds R1 c*B+1.. and R2 N+c*B+1..
            self.n_pairs_file = -(-self.n_pairs_file // len(ranges)) * len(ranges)
        mine = pairs_of_rank(len(ranges), rank, world) if strong else list(range(len(ranges)))
        self.mine, self.ranges = mine, ranges
        th = W._TH(device)
        n_own = sum(ranges[k][1] for k in mine)
        self.n_reads_own = 2 * n_own
        self.text_len = 2 * n_own * W.RECORD_BYTES
        assert self.text_len < (1 << 32) - 64, "one call takes < 4 GB of text: use --stream-reads / fewer pairs per rank"
        self.text = torch.empty(self.text_len + 64, dtype=torch.uint8, device=device)
        at, self.vb = 0, []
        for mate in (1, 2):
            for i, k in enumerate(mine):
                r0, n = ranges[k]
                CH = 100000
                for c0 in range(0, n, CH):
                    m = min(CH, n - c0)
                    self.text[at + c0 * W.RECORD_BYTES: at + (c0 + m) * W.RECORD_BYTES] = W.fastq_text(seed, r0 + c0, m, mate=mate, profile=a.qual, xp=th)
                vblock_i = (k + 1) if mate == 1 else (self.n_pairs_file + k + 1)
                self.vb.append((at, n * W.RECORD_BYTES, vblock_i, -1 if mate == 1 else i))
                at += n * W.RECORD_BYTES
        if device.type == "cuda":
            torch.cuda.synchronize()
        PIN = {"div": {"QUAL": 16, "Q1NAME": (8, 0), "Q2NAME": (0, 16), "Q3NAME": (17, 0), "Q4NAME": (17, 0)},
               "bin": {"QUAL": 18, "Q1NAME": (8, 0), "Q2NAME": (0, 16), "Q3NAME": (17, 0), "Q4NAME": (17, 0)}}
        self.plan = fq.illumina_plan(paired=True, vb_size=vb_bytes(a))       # (a file's last, short VBlock does not set codecs: codec.c:352)
        if a.pin_codecs:
            for c in self.plan["ctxs"]:
                p = PIN[a.qual].get(c["tag"])
                if isinstance(p, int):
                    c["lcodec"] = p
                elif p:
                    c["lcodec"], c["bcodec"] = p
        self.F = E.zip_open(self.plan)
        self.tab = self.F.vb_table(self.vb)
        self.tab2 = None
        self.zbuf = None
        self.offs = None
        self.calls_per_step = 1 if not a.stream_reads else max(1, -(-self.n_pairs_file // len(ranges)))

    def qual_offsets(self, tv):
        import numpy as np
        RB, L = self.W.RECORD_BYTES, self.W.READ_LEN
        return np.arange(len(tv) // RB, dtype=np.int64) * RB + RB - L - 1

    def step(self, dist):
        from genozip_amd.shard import zip_vblocks_sharded
        F, n = self.F, len(self.vb)
        F.reset()
        if self.calls_per_step > 1 and not (self.a.scaling == "strong" and dist is not None) and self.a.two_in_flight:
            # the stream: two calls in flight (gz_fastq_zip_begin / _end) - the next call's seg / merge and coders run beside the
            # previous call's long chains; the same text stands for the next stretch of the file, with later vblock_i
            if self.tab2 is None:
                self.tab2 = F.vb_table(self.vb)
            tabs, in_flight = (self.tab, self.tab2), 0
            for call in range(self.calls_per_step):
                t = tabs[call % 2]
                if in_flight == 2:
                    F.end(); in_flight -= 1
                for e, v in zip(t, self.vb):
                    e.vblock_i = v[2] + len(self.ranges) * call
                F.begin(self.text, self.text_len, t, n); in_flight += 1
            while in_flight:
                F.end(); in_flight -= 1
            last = tabs[(self.calls_per_step - 1) % 2]
        else:
            for call in range(self.calls_per_step):
                if call:
                    for t in self.tab:                             # the next stretch of the stream: same text, later VBlocks
                        t.vblock_i += len(self.ranges)
                zip_vblocks_sharded(F, dist if self.a.scaling == "strong" else None, self.text, self.text_len, self.tab, n)
            last = self.tab
            if self.calls_per_step > 1:
                for t, v in zip(self.tab, self.vb):
                    t.vblock_i = v[2]
        self.last_tab = last
        total = sum(t.z_len for t in last)
        if self.zbuf is None or self.zbuf.numel() < total + 64:
            import torch
            self.zbuf = torch.empty(int(total * 1.05) + 4096, dtype=torch.uint8, device=self.text.device)
        self.offs = F.collect(last, n, self.zbuf, self.zbuf.numel())
        return self.offs[-1]


# nominal cost of the reference's coders on one core, ns per input byte, from BASELINE.md section 2 (the reference's own htscodecs objects, 16 MiB
# streams): [quality-like row, big-endian u32 row] - the fastest and the slowest data of that table - by codec id
REF_CLOCK_NS_PER_BYTE = {6: (1e3 / 232, 1e3 / 168), 7: (1e3 / 124, 1e3 / 99), 8: (1e3 / 330, 1e3 / 129), 9: (1e3 / 104, 1e3 / 52),
                         16: (1e3 / 89, 1e3 / 14.5), 17: (1e3 / 32, 1e3 / 10), 18: (1e3 / 87, 1e3 / 14), 19: (1e3 / 33, 1e3 / 11)}


def clocked_selection(E, wl, z_all, R, kind):
    """codec_assign_best_codec's trials carry clock() (src/codec.c:322-334) and above 5 ms per trial the sorter trades size for time
    (:149-165); the device trials carry no clock and count as "under 5 ms". Outside the timed region: the file's deciding samples (the first
    VBlock's streams of >= 50 bytes, first 99 999 bytes each) through gz_codec_assign_best_ex again with nominal per-codec costs from
    BASELINE.md section 2 - the table's fastest data (quality-like) and its slowest (big-endian u32) - and which contexts would get another codec"""
    from genozip_amd.lib import CODEC_NAMES
    out = {"tables": "ns per byte by codec, BASELINE.md section 2: quality-like row / big-endian u32 row", "samples": 0, "changed_fast_row": {}, "changed_slow_row": {}}
    try:
        for st, codec, did, ulen, pay, domq in walk_sections(z_all[0]):
            if ulen < 50:
                continue
            data = bytes(pay) if codec == 1 else (R.codec_uncompress(codec, pay, ulen) if kind == "port" else R.hts_uncompress("rans" if codec < 16 else "arith", pay, ulen))
            tag = next((c["tag"] for c in wl.plan["ctxs"] if c["dict_id"] == did), did.hex())
            name = ("b250:" if st == 11 else "local:") + tag
            base = E.assign_best_ex(data)[0]
            out["samples"] += 1
            for row, key in ((0, "changed_fast_row"), (1, "changed_slow_row")):
                clk = [0.0] * 32
                for c, v in REF_CLOCK_NS_PER_BYTE.items():
                    clk[c] = v[row]
                w = E.assign_best_ex(data, clock_ns_per_byte=clk)[0]
                if w != base:
                    out[key][name] = "%s -> %s" % (CODEC_NAMES.get(base, base), CODEC_NAMES.get(w, w))
    except Exception as e:                                        # noqa: BLE001 (a side figure)
        out["error"] = repr(e)
    return out


def cpu_leg(wl, z_all, n_threads, E=None):
    """The host's cores on the same file, two legs (both on a C pthread pool, tasks >= 4 x threads so that no thread idles at the end):
    (1) WHOLE PATH (cpu_baseline.value): oracle/gz_oracle_path.c - this repo's C restatement of the path (lines, reads, items, seg columns with
        their hash tables, merge, b250 / local generation, CODEC_DOMQ's transform, codec calls through the reference's own htscodecs where
        oracle/_ref is built, framing), one VBlock per task as the reference's dispatcher runs them (src/dispatcher.c:544-618): the same work as
        the GPU step. FASTQ plans only.
    (2) CODEC CALLS ONLY (cpu_baseline.codec_only): the reference's own rANS / arith code over the SAME section payloads: decodes every section
        the GPU wrote with the reference's decoder, re-encodes with the reference's encoder (timed) and compares with the GPU's payload byte for
        byte (`bit_exact`), checks the decoded QUAL against the text's quality lines."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    kind = "reference" if pyoracle.Ref.available() else "port"
    R = pyoracle.Ref() if kind == "reference" else pyoracle.Oracle()
    O = pyoracle.Oracle()
    tasks, payloads, task_vb, qual_ok = [], [], [], True
    text = wl.text[:wl.text_len].cpu().numpy()
    L = wl.W.READ_LEN
    qual_id = next((c["dict_id"] for c in wl.plan["ctxs"] if c["tag"] == "QUAL" and c["kind"] == 6), None)     # (GZ_FQ_QUAL; a VCF plan has none)
    file_codecs, is_domq = {}, False
    wl.ref_decoded = [[] for _ in z_all]                        # every section as the reference's decoder gives it back (decode_leg compares with it)
    for v, ((off, ln, vi, r1), z) in enumerate(zip(wl.vb, z_all)):
        for st, codec, did, ulen, pay, domq in walk_sections(z):
            data = bytes(pay) if codec == 1 else (R.codec_uncompress(codec, pay, ulen) if kind == "port" else R.hts_uncompress("rans" if codec < 16 else "arith", pay, ulen))
            wl.ref_decoded[v].append(data)
            tag = next((c["tag"] for c in wl.plan["ctxs"] if c["dict_id"] == did), None)
            if tag and codec != 1:
                file_codecs[("b250" if st == 11 else "local", tag)] = codec
            if did == qual_id and st == 12:
                is_domq |= bool(domq)
                tv = text[off:off + ln]
                qo = wl.qual_offsets(tv)                                                   # where every record's QUAL (L scores) starts in the VBlock's text
                lines = tv[qo[:, None] + np.arange(L)]
                # (FASTQ: a line of one repeated score is a special snip in QUAL's b250 and no part of the local / of CODEC_DOMQ's streams, fastq_qual.c:33-36,74)
                keep = ~(lines == lines[:, :1]).all(axis=1) if not wl.plan.get("record_lines") else np.ones(len(qo), dtype=bool)
                if domq:                       # the stream CODEC_DOMQ leaves of these quality lines, by the CPU restatement of codec_domq.c
                    qual_ok &= data == pyoracle.oracle_domq(O, tv.tobytes(), qo.astype(np.uint32), np.where(keep, L, 0).astype(np.uint32))["qual"]
                else:
                    qual_ok &= data == lines[keep].tobytes()
            if codec != 1:
                tasks.append((codec, data)); payloads.append(bytes(pay)); task_vb.append(v)
    if not tasks:
        return None, False
    codecs, datas = [t[0] for t in tasks], [t[1] for t in tasks]
    nbytes = sum(len(d) for d in datas)

    def pool(nt, idx=None, replicas=1):
        cs = codecs if idx is None else [codecs[i] for i in idx]
        ds = datas if idx is None else [datas[i] for i in idx]
        if kind == "reference":
            outs, dt = R.codec_compress_many(cs, ds, nt, replicas)
        else:
            t0 = time.time(); outs = O.codec_compress_many(cs, ds, nt, replicas); dt = time.time() - t0
        return outs, dt, sum(len(d) for d in ds) * replicas
    pool(8, list(range(min(8, len(tasks)))))                                    # warm up
    outs, dt_all, _ = pool(n_threads)                                           # the file as it is: 150 tasks on 256 threads
    n_diff = sum(o != p for o, p in zip(outs, payloads))
    exact = n_diff == 0 and qual_ok
    if not exact:
        bad = next((i for i, (o, p) in enumerate(zip(outs, payloads)) if o != p), None)
        sys.stderr.write("bit_exact: %d of %d payloads differ from the reference coder's, decoded QUAL %s%s\n" % (n_diff, len(payloads), "matches" if qual_ok else "DIFFERS from the text",
                         "" if bad is None else "; first: task %d of VBlock %d, codec %d, %d bytes in%s" % (bad, task_vb[bad] + 1, codecs[bad], len(datas[bad]),
                         (": data %s library %s reference %s" % (datas[bad].hex(), payloads[bad].hex(), outs[bad].hex())) if len(datas[bad]) <= 256 else "")))
    # tasks >= 4 x threads: the same sections again and again (a file with more VBlocks of the same kind) until no thread idles at the end -
    # bounded to ~10 s of pool time by the rate just measured
    reps = max(1, -(-4 * n_threads // len(tasks)))
    reps = max(1, min(reps, int((4.0 if os.environ.get("GZ_BENCH_SIDE_LEG") else 10.0) / max(dt_all, 1e-3)) or 1))
    _, dt_rep, nb_rep = pool(n_threads, None, reps)
    # one thread (BASELINE configs[0]) on a bounded sample: the sections of the first VBlock pair
    first = [i for i, v in enumerate(task_vb) if v in (0, len(wl.vb) // 2)]
    _, dt_1, nb_1 = pool(1, first)
    scale = wl.value_bytes / nbytes                                             # same unit as `value`: text without SEQ per second
    codec_only = {"value": round(nb_rep / dt_rep / 1e6 * scale, 1), "unit": "MB/s", "cores": n_threads, "kind": kind,
                  "sample": "codec calls only (no seg / merge / generate on the CPU side): the %d coded sections of rank 0's %d VBlocks (%.0f MB of streams) x %d = %d tasks on %d threads "
                            "(%s); in the unit of `value` (text without SEQ lines: x %.3f)" % (len(tasks), len(wl.vb), nbytes / 1e6, reps, reps * len(tasks), n_threads, usable_cpus()[1], scale),
                  "stream_mb_s": round(nb_rep / dt_rep / 1e6, 1),
                  "this_file_alone": {"value": round(nbytes / dt_all / 1e6 * scale, 1), "stream_mb_s": round(nbytes / dt_all / 1e6, 1), "tasks": len(tasks),
                                      "note": "the file's own sections once: fewer tasks than threads, the longest streams set the time (what round 3 reported as cpu_baseline)"},
                  "one_thread": {"stream_mb_s": round(nb_1 / dt_1 / 1e6, 1), "sample": "sections of the first VBlock pair (%.1f MB)" % (nb_1 / 1e6)}}
    out = dict(codec_only)
    out["codec_only"] = codec_only
    # (1) the whole path: the FASTQ plans (gzo_fastq_vb_path) and the one-line-record plans of SAM / VCF text (gzo_text_vb_path)
    if getattr(wl, "plan", None) is not None:
        try:
            vbs = [(off, ln) for (off, ln, vi, r1) in wl.vb]
            ref = R if kind == "reference" else None
            path_many = pyoracle.text_path_many if wl.plan.get("record_lines") else pyoracle.fastq_path_many
            big = sum(ln for _o, ln in vbs) > (1 << 30)                                          # (a few VBlocks of hundreds of MB: every pass costs seconds - one pass is the sample)
            if not big:
                path_many(O, text, vbs[:2], wl.plan, file_codecs, is_domq, 2, 1, ref)          # warm up
            dt1, zl, stb = path_many(O, text, vbs, wl.plan, file_codecs, is_domq, n_threads, 1, ref)
            reps_w = max(1, -(-4 * n_threads // len(vbs)))
            reps_w = 1 if big else max(1, min(reps_w, int((5.0 if os.environ.get("GZ_BENCH_SIDE_LEG") else 15.0) / max(dt1, 1e-3)) or 1))
            dtw = dt1
            if reps_w > 1:
                dtw, zl, stb = path_many(O, text, vbs, wl.plan, file_codecs, is_domq, n_threads, reps_w, ref)
            dt_one = None if big else path_many(O, text, vbs[:1], wl.plan, file_codecs, is_domq, 1, 1, ref)[0]   # (big: not measured - and not derived from the threaded pass, which would assume perfect scaling)
            per_vb_value = wl.value_bytes / max(1, getattr(wl, "calls_per_step", 1))
            whole = {"value": round(per_vb_value * reps_w / dtw / 1e6, 1), "unit": "MB/s", "cores": n_threads, "kind": "port",
                     "codecs": "the reference's htscodecs (oracle/_ref)" if ref is not None else "this repo's C restatement",
                     "sample": "the WHOLE path of the file's %d VBlocks x %d = %d tasks, one VBlock per task on %d threads (%s): text -> lines -> reads -> items -> seg columns "
                               "(hash tables, dictionaries, b250) -> merge -> generate -> %scodec calls (the file's codecs, no trials) -> framed sections "
                               "(oracle/gz_oracle_path.c; every VBlock with file-level contexts of its own: no merge mutex, which favours the CPU); %.1f s"
                               % (len(vbs), reps_w, reps_w * len(vbs), n_threads, usable_cpus()[1], "CODEC_DOMQ's transform -> " if is_domq else "", dtw),
                     "z_bytes": int(sum(zl)), "stream_bytes": int(sum(stb)),
                     "this_file_alone": {"value": round(per_vb_value / dt1 / 1e6, 1), "tasks": len(vbs)},
                     "one_thread": {"value": round(per_vb_value / len(vbs) / dt_one / 1e6, 1) if dt_one else None,
                                    "sample": "the first VBlock" if dt_one else "not measured: a VBlock of this workload is hundreds of MB (the threaded pass above includes its first-touch costs: no warm-up pass either)"}}
            out = dict(whole)
            out["whole_path"] = whole
            out["codec_only"] = codec_only
        except Exception as e:                                    # noqa: BLE001 (the codec leg still stands)
            out["whole_path"] = {"error": repr(e)}
    out["host"] = usable_cpus()[1]
    if E is not None:
        out["codec_selection_with_clock"] = clocked_selection(E, wl, z_all, R, kind)
    return out, bool(exact)


def usable_cpus():
    """the CPUs this process may really use: logical CPUs, its affinity mask and the container's CFS quota (cgroup v2 cpu.max / v1 cfs_quota_us) -
    more threads than that only buy throttling. Returns (threads to use, a sentence for the record). The MI355X boxes of this pool: 256 logical
    CPUs (2 x EPYC 9575F), cpu.max = 1600000 100000 -> 16."""
    n = os.cpu_count() or 1
    note = "%d logical CPUs" % n
    try:
        a = len(os.sched_getaffinity(0))
        if a < n:
            n, note = a, note + ", affinity mask of %d" % a
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None and quota < n:
        n, note = max(1, int(quota + 0.5)), note + ", cgroup CPU quota of %.1f CPUs (cpu.max): threads beyond it are throttled, not run" % quota
    return min(n, 256), note


def gpu_over_cpu(out, cb):
    """the honest ratios, at top level: this GPU against this host's cores on the same file, every CPU thread kept busy"""
    if not cb:
        return None
    g = {"note": "value / cpu_baseline figures of the same unit; tasks >= 4 x threads on the CPU side (a file with enough VBlocks to keep every core busy)"}
    if isinstance(cb.get("whole_path"), dict) and cb["whole_path"].get("value"):
        g["whole_path"] = round(out["value"] / cb["whole_path"]["value"], 2)
    co = cb.get("codec_only") or cb
    if co.get("value"):
        g["codec_calls_only"] = round(out["value"] / co["value"], 2)
        if co.get("this_file_alone", {}).get("value"):
            g["codec_calls_only_this_file_alone"] = round(out["value"] / co["this_file_alone"]["value"], 2)
    return g


def pmc_traffic(kernel, a):
    """HBM bytes per STEP of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/r06_pmc.json, made by tools/summarize_prof.py: counter collection serialises kernels, so it cannot happen inside a timed
    run); null when there is no such file for this workload"""
    p = next((q for q in (os.path.join(ROOT, "profiles", n) for n in ("r06_pmc.json", "r05_pmc.json", "round5_pmc.json")) if os.path.exists(q)), None)
    if p is None:
        return None
    d = json.load(open(p))
    if d.get("workload") != {"pairs": a.pairs, "vb_bytes": vb_bytes(a), "qual": a.qual}:
        return None
    k = d["kernels"].get(kernel)
    return k.get("traffic_bytes_per_step") if k else None


class SamWorkload:
    """BASELINE configs[2] FROM TEXT: 1 M aligned 150 bp reads as SAM alignment lines in HBM (genozip_amd/workload.py::sam_text), VBlocks by
    the reference's rule, through the VBlock compute driver with the one-line-record plan of genozip_amd/sam.py"""

    def __init__(self, E, a, device):
        import torch
        from genozip_amd import workload as W, sam as sm
        self.E, self.a, self.W = E, a, W
        n = a.pairs
        th = W._TH(device)
        CH = 100000
        parts = [W.sam_text(3, r0, min(CH, n - r0), profile="bin" if a.qual == "bin" else "div", xp=th) for r0 in range(0, n, CH)]
        self.text_len = int(sum(p.numel() for p in parts))
        self.text = torch.empty(self.text_len + 64, dtype=torch.uint8, device=device)
        at = 0
        for p in parts:
            self.text[at:at + p.numel()] = p; at += p.numel()
        del parts
        import numpy as np
        self.bam = None
        if a.bam_binary:
            # the same alignments as the records of an uncompressed BAM stream (workload.py::bam_records): every step finds the records
            # (gz_bam_records) and writes their alignment lines (gz_bam_to_sam) in front of the SAM plan; the text generated above is
            # only what that conversion is compared with, once, here
            parts = [W.bam_records(3, r0, min(CH, n - r0), profile="bin" if a.qual == "bin" else "div", xp=th) for r0 in range(0, n, CH)]
            self.bam_len = int(sum(p.numel() for p in parts))
            self.bam = torch.empty(self.bam_len + 64, dtype=torch.uint8, device=device)
            at = 0
            for p in parts:
                self.bam[at:at + p.numel()] = p; at += p.numel()
            del parts
            self.ref_names = W.SAM_REF_NAMES
            want = self.text
            _, rob, nrec = E.bam_records(self.bam[:self.bam_len], len(self.ref_names), cap=n + 16, on_device=True)
            tb, tl, lb = E.bam_to_sam(self.bam[:self.bam_len], rob, self.ref_names, text_cap=self.text_len + 4096, on_device=True, n_rec=nrec)
            assert nrec == n and tl == self.text_len and bool(torch.equal(tb[:tl], want[:tl])), "BAM records -> alignment lines differ from the SAM text generator"
            self.text = tb
            rec_off = np.frombuffer(E.mem.download(rob, 4 * nrec), dtype=np.uint32).astype(np.int64)
            line_off = np.frombuffer(E.mem.download(lb, 4 * (nrec + 1)), dtype=np.uint32).astype(np.int64)
            # VBlocks of BAM bytes: segconf_set_vb_size with est_max_threads = 60 for a BGZF / BAM source (src/segconf.c:186-203), cut at records
            vbb = int(a.vb_mb * (1 << 20)) if a.vb_mb else min(2 * (20 << 20), max(4 << 20, int(self.bam_len * 1.2 / 60)))
            rcuts, at = [0], 0
            while at < nrec:
                k = int(np.searchsorted(rec_off, rec_off[at] + vbb, side="right")) - 1      # the last record that starts within vb_size of the VBlock's first
                k = max(at + 1, min(nrec, k))
                rcuts.append(k); at = k
            self.rec_cuts = rcuts
            cuts = [int(line_off[k]) for k in rcuts]
            self.vb_bytes = vbb
        else:
            # VBlocks: segconf_set_vb_size's figure for the file (as vb_bytes for FASTQ), cut at line ends
            vbb = int(a.vb_mb * (1 << 20)) if a.vb_mb else min(2 * (20 << 20), max(4 << 20, int(self.text_len * 1.2 / 30)))
            self.vb_bytes = vbb
            nl = torch.nonzero(self.text[:self.text_len] == 10).reshape(-1)
            ends = (nl + 1).cpu().numpy()
            cuts, at = [0], 0
            while at < self.text_len:
                k = int(np.searchsorted(ends, at + vbb, side="right")) - 1
                nxt = int(ends[k]) if k >= 0 and ends[k] > at else int(ends[np.searchsorted(ends, at, side="right")])
                cuts.append(min(nxt, self.text_len)); at = cuts[-1]
        self.vb = [(cuts[i], cuts[i + 1] - cuts[i], i + 1, -1) for i in range(len(cuts) - 1)]
        self.n_reads_own = n
        self.plan = sm.sam_plan(vb_size=vbb, aux_tags=[("NM", "i"), ("AS", "i")])       # (a context per optional field behind the AUX container, sam_seg_aux_all)
        self.F = E.zip_open(self.plan)
        self.tab = self.F.vb_table(self.vb)
        self.zbuf, self.offs, self.calls_per_step = None, None, 1

    def qual_offsets(self, tv):
        import numpy as np
        tabs = np.flatnonzero(tv == 9).reshape(-1, 12)                      # 12 tabs a line: QUAL sits behind the 10th
        return tabs[:, 9].astype(np.int64) + 1

    def step(self, dist):
        import torch
        F, n = self.F, len(self.vb)
        F.reset()
        if getattr(self, "bam", None) is not None:           # N1 for BAM: records -> alignment lines, in the step
            _, rob, nrec = self.E.bam_records(self.bam[:self.bam_len], len(self.ref_names), cap=self.n_reads_own + 16, on_device=True)
            self.text, tl, _lb = self.E.bam_to_sam(self.bam[:self.bam_len], rob, self.ref_names, text_cap=self.text_len + 4096, on_device=True, n_rec=nrec)
            assert tl == self.text_len
        F.zip_table(self.text, self.text_len, self.tab, n)
        total = sum(t.z_len for t in self.tab)
        if self.zbuf is None or self.zbuf.numel() < total + 64:
            self.zbuf = torch.empty(int(total * 1.05) + 4096, dtype=torch.uint8, device=self.text.device)
        self.offs = F.collect(self.tab, n, self.zbuf, self.zbuf.numel())
        return self.offs[-1]


class VcfWorkload:
    """BASELINE configs[3] FROM TEXT, one GPU's share at 8 GPUs: VBlocks of 3 000 data lines x 10 000 samples (the 512 MB the reference's rule
    clamps a VBlock of this file to, src/segconf.c:102,160-175) as VCF text in HBM (genozip_amd/workload.py::vcf_text), through the VBlock
    compute driver with the per-sample plan of genozip_amd/vcf.py"""

    def __init__(self, E, a, device):
        import torch
        from genozip_amd import workload as W, vcf as vc
        self.E, self.a, self.W = E, a, W
        self.n_samples, self.lines_per_vb, n_vb = a.vcf_samples, a.vcf_lines, a.vcf_vbs
        th = W._TH(device)
        parts, self.vb, at = [], [], 0
        for v in range(n_vb):
            ln = 0
            for l0 in range(0, self.lines_per_vb, 250):
                p = W.vcf_text(4, v * self.lines_per_vb + l0, min(250, self.lines_per_vb - l0), self.n_samples, xp=th)
                parts.append(p); ln += p.numel()
            self.vb.append((at, ln, v + 1, -1)); at += ln
        self.text_len = at
        assert self.text_len < (1 << 32) - 64, "one call takes < 4 GB of text"
        self.text = torch.empty(self.text_len + 64, dtype=torch.uint8, device=device)
        at = 0
        for p in parts:
            self.text[at:at + p.numel()] = p; at += p.numel()
        del parts
        self.plan = vc.vcf_plan(self.n_samples, vb_size=512 << 20)
        self.F = E.zip_open(self.plan)
        self.tab = self.F.vb_table(self.vb)
        self.zbuf, self.offs, self.calls_per_step, self.value_bytes = None, None, 1, self.text_len

    step = SamWorkload.step


def sam_leg(a):
    return text_leg(a, SamWorkload)


def vcf_leg(a):
    return text_leg(a, VcfWorkload)


def text_leg(a, WL):
    """BASELINE configs[2] on one GPU, from SAM text resident in HBM (N1 for SAM, genozip_amd/sam.py) to finished VBlocks"""
    import torch
    from genozip_amd.codec import Engine
    from genozip_amd.lib import CODEC_NAMES
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    E = Engine(device=0)
    wl = WL(E, a, device)
    is_sam = WL is SamWorkload
    os.environ["GZ_ZIP_PRIOR_ONLY"] = "1"
    for _ in range(max(1, a.warmup)):
        wl.step(None)
    dt, prof, prof_max, prof_table = profiled_steps(E, lambda: wl.step(None), a.steps, torch.cuda.synchronize)
    ms = dt / a.steps * 1e3
    # warm: a handle that has seen a file of the kind codes every stream ahead of its context's trial with the codec that file got (gz_zip_prediction)
    warm = None
    if a.warm_steps:
        os.environ.pop("GZ_ZIP_PRIOR_ONLY", None)
        wl.step(None)
        h0, m0 = wl.F.prediction()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(a.warm_steps):
            wl.step(None)
        torch.cuda.synchronize()
        wms = (time.perf_counter() - t1) / a.warm_steps * 1e3
        h1, m1 = wl.F.prediction()
        warm = {"ms_per_step": round(wms, 3), "steps": a.warm_steps, "prediction_hits": h1 - h0, "prediction_misses": m1 - m0,
                "note": "the handle remembers the codecs of the previous file of the kind: every stream is coded ahead of its context's trial compressions, which still run and decide"}
    z_total = wl.offs[-1]
    zhost = wl.zbuf[:z_total].cpu().numpy().tobytes()
    z_all = [zhost[wl.offs[i]:wl.offs[i + 1]] for i in range(len(wl.vb))]
    secs = [s_ for z in z_all for s_ in walk_sections(z)]
    stream_bytes = sum(s_[3] for s_ in secs)
    n_bases = sum(int(t.n_bases) for t in wl.tab)
    wl.value_bytes = wl.text_len - n_bases                      # the text without its SEQ fields (2-bit packed in the step, LZMA outside the path: as for FASTQ); VCF: all of it
    dom = max(prof, key=lambda k: prof[k][0])
    dom_ms, dom_n = prof[dom]
    alg = stream_bytes + z_total
    ach = alg / (dom_n / a.steps) / (dom_ms / dom_n / 1e3) / 1e9
    long_secs = [s_ for s_ in secs if s_[3] >= (1 << 20) and s_[1] in (16, 17, 18, 19)]
    codecs = {}
    for st, codec, did, ulen, pay, _d in walk_sections(z_all[0]):
        tag = next((c["tag"] for c in wl.plan["ctxs"] if c["dict_id"] == did), did.hex())
        codecs[("b250:" if st == 11 else "local:") + tag] = CODEC_NAMES.get(codec, str(codec))
    out = {"metric": METRIC, "value": round(wl.value_bytes / 1e6 / (ms / 1e3), 1), "unit": "MB/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": (("BAM-1M (BASELINE configs[2]) FROM BAM RECORDS: %d aligned 150 bp reads as the records of an uncompressed BAM stream in HBM (%.1f MB; coordinate-sorted, CIGAR 90 %% 150M, "
                                    "QUAL profile %s), %d VBlocks of %.2f MB of BAM (segconf_set_vb_size for a BGZF source); per step: the record chain (gz_bam_records) and the records' alignment "
                                    "lines (gz_bam_to_sam: N1 for BAM, bam_seg_txt_line's conversions) -> the one-line-record plan of genozip_amd/sam.py -> a1-a16; a new file every step. MB counted in "
                                    "`value` = the alignment lines WITHOUT their SEQ fields (as for SAM text), so that the figure compares with the SAM leg" % (a.pairs, wl.bam_len / 1e6, a.qual, len(wl.vb), wl.vb_bytes / 1e6))
                                   if getattr(wl, "bam", None) is not None else
                                   ("BAM-1M (BASELINE configs[2]) FROM TEXT: %d aligned 150 bp reads as SAM alignment lines (coordinate-sorted, CIGAR 90 %% 150M, QUAL profile %s), %d VBlocks of %.2f MB; "
                                   "the whole path per step from text in HBM through the one-line-record plan of genozip_amd/sam.py (N1 for SAM: fields by tab, QNAME by flavor -> a1-a16); "
                                   "a new file every step. MB counted in `value` = text WITHOUT the SEQ fields (2-bit packed in the step, LZMA outside the path)" % (a.pairs, a.qual, len(wl.vb), wl.vb_bytes / 1e6)))
                                  if is_sam else
                                  ("VCF %d samples (BASELINE configs[3]) FROM TEXT, one GPU's share at 8 GPUs: %d VBlocks of %d data lines x %d samples (FORMAT GT:DP:PL), %.0f MB of text; the whole path per "
                                   "step from text in HBM through the per-sample plan of genozip_amd/vcf.py (N1 for VCF: fixed fields by tab, FORMAT subfields of every sample -> GT / PL b250 columns "
                                   "of lines x samples entries, DP a dyn-int matrix written transposed, a1-a16); a new file every step. MB counted in `value` = the text"
                                   % (wl.n_samples, len(wl.vb), wl.lines_per_vb, wl.n_samples, wl.text_len / 1e6)),
                      "text_mb_per_step": round(wl.text_len / 1e6, 1), "stream_mb_per_step": round(stream_bytes / 1e6, 1), "compressed_mb_per_step": round(z_total / 1e6, 2), "codecs": codecs,
                      "codec_prediction": "hits %d, misses %d (sections coded ahead of their context's trial with a predicted codec - cold: the built-in prior -, kept / coded again)" % wl.F.prediction()},
           "text_mb_s": round(wl.text_len / 1e6 / (ms / 1e3), 1), "stream_mb_s": round(stream_bytes / 1e6 / (ms / 1e3), 1),
           "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": None,
                        "avg_launch_ms": round(dom_ms / dom_n, 4), "launches_per_step": dom_n / a.steps, "longest_launch_ms": round(prof_max.get(dom, 0), 3),
                        "long_streams": len(long_secs), "symbols_of_longest_stream": max([s_[3] for s_ in long_secs] + [0]),
                        "kernel_ms_per_step_summed_over_concurrent_launches": kernel_table(prof, prof_table, a.steps, 12)}}
    if warm:
        warm["value"] = round(wl.value_bytes / 1e6 / (warm["ms_per_step"] / 1e3), 1)
        out["warm"] = warm
    if not a.no_cpu:
        cb, exact = cpu_leg(wl, z_all, usable_cpus()[0], E)
        out["cpu_baseline"] = cb
        out["bit_exact"] = exact
        out["gpu_over_cpu"] = gpu_over_cpu(out, cb)
        out["decode"] = decode_leg(E, wl, z_all)
        if out["decode"]["equals_reference_decoder"] is False:
            out["bit_exact"] = False
        out["bit_exact_payloads"] = out["bit_exact"]
        out["file_exact"] = file_exact_leg(wl, z_all)
        if out["file_exact"].get("checked"):
            out["bit_exact"] = bool(out["bit_exact"] and out["file_exact"]["identical"] == out["file_exact"]["vblocks"])
    print(json.dumps(out))


def file_exact_leg(wl, z_all):
    """bit_exact as a statement about the FILE: every VBlock's z_data of the last timed step against the oracle's composition of the whole path -
    tests/parity.py::fastq_zip_expected, what every driver test compares with: the reference's per-context functions restated (oracle/gz_oracle.c),
    chained VBlock by VBlock with the real merged dictionaries, codecs found by the reference's trial rule - run once over the same text, outside
    the timed region, on one host thread. (The payload-level check of cpu_leg - every section's payload is what the reference's own coder makes of
    the bytes its own decoder gives back - says nothing about WHICH bytes a section holds: a wrong word index would pass it.) A streamed workload's
    step holds several calls on one file: the VBlocks compared are the last call's, and the composition cannot know the dictionaries the earlier
    calls left - such lines say so and keep the payload-level check alone."""
    t0 = time.perf_counter()
    try:
        for p_ in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
            if p_ not in sys.path:
                sys.path.insert(0, p_)
        import parity
        import pyoracle
        if getattr(wl, "calls_per_step", 1) > 1:
            return {"checked": False, "why": "several calls per step on one file: the last call's VBlocks depend on dictionaries of the calls before"}
        O = pyoracle.Oracle()
        text = wl.text[:wl.text_len].cpu().numpy().tobytes()
        vbs = [tuple(int(x) for x in v[:4]) for v in wl.vb]
        if wl.plan.get("n_samples"):
            return {"checked": False, "why": "VCF plan: the composition for per-sample columns lives in tests/parity.py::vcf_zip (small sizes)"}
        want, _ = parity.fastq_zip_expected(O, wl.plan, text, vbs)
        bad = [v for v, (w, z) in enumerate(zip(want, z_all)) if bytes(w["z"]) != bytes(z)]
        return {"checked": True, "vblocks": len(z_all), "identical": len(z_all) - len(bad), "first_difference": None if not bad else {"vblock": bad[0] + 1, "at": parity._first_diff(bytes(z_all[bad[0]]), bytes(want[bad[0]]["z"]))},
                "oracle_z_bytes": int(sum(len(w["z"]) for w in want)), "seconds": round(time.perf_counter() - t0, 1)}
    except Exception as e:                                        # noqa: BLE001
        return {"checked": False, "error": repr(e)[:300], "seconds": round(time.perf_counter() - t0, 1)}


SIDE_LEGS = (("bam_text", "BASELINE configs[2], from SAM text", ["--config", "bam"]),
             ("bam_records", "BASELINE configs[2], from the records of an uncompressed BAM stream", ["--config", "bam", "--bam-binary"]),
             ("vcf_share", "BASELINE configs[3], one GPU's share at 8 GPUs", ["--config", "vcf"]),
             ("streamed", "BASELINE configs[4] at reduced scale: 8 M read pairs streamed through one file in calls of 112 VBlock pairs", ["--stream-reads", "8000000"]))


def side_legs():
    """every other BASELINE configuration beside the headline, so that the driver's ONE default run carries all five: each is this same script
    run on its configuration (a process of its own, 3 timed steps, its CPU leg and its decode leg once), reduced here to its key figures.
    A leg that fails leaves its error text; the headline never depends on them. GZ_BENCH_NO_SIDE_LEGS=1 leaves them out."""
    out = {}
    for name, what, args in SIDE_LEGS:
        t0 = time.perf_counter()
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__)] + args + ["--steps", "3", "--warmup", "1", "--warm-steps", "0"],
                               env=dict(os.environ, GZ_BENCH_SIDE_LEG="1", GZ_BENCH_NO_SIDE_LEGS="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            d = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1])
            r = d.get("roofline", {}); cb = d.get("cpu_baseline", {}); dec = d.get("decode") or {}
            out[name] = {"what": what, "ms_per_step": d["ms_per_step"], "value": d["value"], "unit": d["unit"], "steps": d["steps"], "bit_exact": d.get("bit_exact"),
                         "dominant_kernel": r.get("kernel"), "roofline_frac": r.get("frac"), "chain_ns_per_symbol": (r.get("critical_path") or {}).get("ns_per_symbol"),
                         "gpu_over_cpu": {k: v for k, v in (d.get("gpu_over_cpu") or {}).items() if k != "note"},
                         "cpu_baseline": {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind")},
                         "decode": {"ms": dec.get("ms"), "uncompressed_mb_s": dec.get("uncompressed_mb_s"), "equals_reference_decoder": dec.get("equals_reference_decoder"),
                                    "cpu_decode_mb_s": dec.get("cpu_decode_mb_s"), "device_full_mb_s": (dec.get("throughput") or {}).get("uncompressed_mb_s")},
                         "text_mb_per_step": d["config"].get("text_mb_per_step"), "wall_s": round(time.perf_counter() - t0, 1)}
        except Exception as e:                                    # noqa: BLE001
            out[name] = {"what": what, "error": repr(e)[:300], "wall_s": round(time.perf_counter() - t0, 1)}
    return out


def decode_leg(E, wl, z_all, reps=3):
    """row a14 at full size: every section of every VBlock the step wrote, decoded again on the GPU in ONE call (gz_vb_uncompress_many: section
    walk + adler32 check by two kernels, all payloads as one batch of streams) and compared with what the reference's own decoder made of the
    same payloads (cpu_leg). Timed with the compressed VBlocks resident in HBM; a side figure, never `value`."""
    if os.environ.get("GZ_BENCH_SIDE_LEG"):
        reps = 1
    totals = [sum(ulen for _st, _c, _d, ulen, _p, _q in walk_sections(z)) for z in z_all]
    items = [((E.mem.upload(bytes(z)), len(z)), t) for z, t in zip(z_all, totals)]
    E.sync()
    best, res = None, None
    for _ in range(reps):
        t0 = time.perf_counter()
        res = E.vb_uncompress_many(items, download=False)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    same = None
    if getattr(wl, "ref_decoded", None):
        same = True
        for (ob, offs, _dec), total, want in zip(res, totals, wl.ref_decoded):
            raw = E.mem.download(ob, total)
            same &= len(want) == len(offs) - 1 and all(raw[offs[k]:offs[k + 1]] == want[k] for k in range(len(want)))
    nsec = sum(len(o) - 1 for _ob, o, _dec in res)
    out = {"ms": round(best * 1e3, 2), "vblocks": len(z_all), "sections": nsec, "uncompressed_mb": round(sum(totals) / 1e6, 1), "compressed_mb": round(sum(len(z) for z in z_all) / 1e6, 1),
           "uncompressed_mb_s": round(sum(totals) / 1e6 / best, 1), "equals_reference_decoder": same,
           "note": "gz_vb_uncompress_many over the step's own output, compressed VBlocks in HBM, best of %d; one wave per stream (a serial adaptive coder), the longest stream sets the time" % reps}
    out.update(cpu_decode_leg(z_all))
    if out.get("cpu_decode_mb_s"):
        out["gpu_over_cpu_decode"] = round(out["uncompressed_mb_s"] / out["cpu_decode_mb_s"], 3)
    # The figure above is a LATENCY: one wave per stream, the longest stream sets the time, and a file's few hundred streams leave most of the
    # device idle. What the device decodes per second when it is full - a read-ahead holding many files' VBlocks, as piz would have them - is
    # measured here: the same output several times over in ONE call (separate outputs), as many copies as bring the call to ~2000 long streams
    # or 16 GB of output, whichever comes first.
    try:
        long_streams = sum(1 for z in z_all for _st, codec, _did, ulen, _p, _q in walk_sections(z) if codec != 1 and ulen >= (1 << 20))
        copies = max(2, min(16, -(-2000 // max(1, long_streams)), int((16 << 30) // max(1, sum(totals)))))
        if os.environ.get("GZ_BENCH_SIDE_LEG") and sum(totals) * copies > (6 << 30):
            copies = max(2, int((6 << 30) // max(1, sum(totals))))
        many = items * copies
        E.vb_uncompress_many(many[:len(items)], download=False)
        t0 = time.perf_counter()
        E.vb_uncompress_many(many, download=False)
        dt = time.perf_counter() - t0
        out["throughput"] = {"copies_in_one_call": copies, "long_streams_in_flight": long_streams * copies, "ms": round(dt * 1e3, 1),
                             "uncompressed_mb_s": round(sum(totals) * copies / 1e6 / dt, 1),
                             "gpu_over_cpu_decode": round(sum(totals) * copies / 1e6 / dt / out["cpu_decode_mb_s"], 3) if out.get("cpu_decode_mb_s") else None,
                             "note": "the device full: the step's output %d times over in one gz_vb_uncompress_many call" % copies}
    except Exception as e:                                        # noqa: BLE001
        out["throughput"] = {"error": repr(e)[:200]}
    return out


def cpu_decode_leg(z_all):
    """the same sections through the reference's own decoders (htscodecs' rans_uncompress_to_4x16 / arith_uncompress_to, src/codec_htscodecs.c:100-123,
    compiled in place: oracle/_ref) on the host's usable cores, a section per task - the file's sections as often as it takes to give every thread
    four tasks, a bounded sample of them when the file is large. What codec_uncompress costs on the CPU side, beside the device's figure."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle
        from concurrent.futures import ThreadPoolExecutor
        if not pyoracle.Ref.available():
            return {"cpu_decode_mb_s": None, "cpu_decode_note": "oracle/_ref not built"}
        R = pyoracle.Ref()
        n_threads = usable_cpus()[0]
        secs = [(codec, bytes(pay), ulen) for z in z_all for _st, codec, _did, ulen, pay, _dq in walk_sections(z) if codec != 1 and ulen]
        secs.sort(key=lambda t: -t[2])
        budget, pick, total = 600 << 20, [], 0                     # at most ~600 MB of output per pass: a few seconds on 16 threads
        for t in secs:
            if total + t[2] > budget and pick:
                continue
            pick.append(t); total += t[2]
        reps = max(1, -(-4 * n_threads // max(1, len(pick))))
        tasks = pick * reps

        def one(t):
            R.hts_uncompress("rans" if t[0] < 16 else "arith", t[1], t[2])
            return t[2]
        with ThreadPoolExecutor(max_workers=n_threads) as ex:
            list(ex.map(one, tasks[:n_threads]))                    # (threads up, pages touched)
            t0 = time.perf_counter()
            done = sum(ex.map(one, tasks))
            dt = time.perf_counter() - t0
        return {"cpu_decode_mb_s": round(done / 1e6 / dt, 1), "cpu_decode_cores": n_threads,
                "cpu_decode_note": "the reference's own decoders (oracle/_ref: htscodecs compiled in place) on %d threads: %d of the output's %d coded sections x %d = %d tasks, %.0f MB decoded in %.2f s"
                                   % (n_threads, len(pick), len(secs), reps, len(tasks), done / 1e6, dt)}
    except Exception as e:                                        # noqa: BLE001
        return {"cpu_decode_mb_s": None, "cpu_decode_note": repr(e)[:200]}


def kernel_table(prof, table, n_steps, top):
    """ms per step by kernel: the timed steps' figures where a kernel was timed there, the untimed step's otherwise"""
    t = dict(table)
    t.update({k: v[0] / n_steps for k, v in prof.items()})
    return {k: round(v, 3) for k, v in sorted(t.items(), key=lambda kv: -kv[1])[:top]}


def profiled_steps(E, step, n_steps, sync):
    """The timed region. Every kernel launch between two HIP events costs the host as much as the launch itself (~700 launches a step): the
    table of all kernels comes from ONE untimed step in front (gz_profile mode 1), the timed steps carry events only on the launches of the two
    kernels a step can be as long as (mode 2: k_arith_chain, k_arith_model - the dominant kernel of every configuration is one of them; its
    launches are timed live, over the timed region, on the streams they run on). -> (seconds, {kernel: (ms, launches)} of the timed steps,
    {kernel: longest launch}, {kernel: ms per step} of the untimed step)"""
    mode = int(os.environ.get("GZ_BENCH_PROF_MODE", "2"))          # (1: every launch in the timed region as well, as rounds 1 - 5 did)
    E.profile(True, reset=True)
    step()
    sync()
    E.profile(False)
    table = {k: v[0] for k, v in E.profile_results().items()}
    E.profile(mode, reset=True)
    sync()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    E.profile(False)
    prof = E.profile_results()
    return dt, prof, dict(E.profile_max), table



def config_leg(a):
    """--stream-level: BASELINE configs[2] (BAM-1M) / configs[3] (VCF 10 k samples, one GPU's share) entered at the CONTEXT-STREAM level - the
    context streams of SURVEY 8(0) generated at the sizes the reference would give them; a step = b250 generation + local byte order /
    transposes + codecs + section framing of every VBlock (tools/config_bench.py). The default legs of these configurations start from text /
    BAM records through the one-line-record plans (sam_leg / vcf_leg); this entry stays for the streams alone."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    import config_bench as cb
    from genozip_amd.codec import Engine
    from genozip_amd.lib import CODEC_NAMES
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    E = Engine(device=0)
    vbs, text_bytes = cb.bam_vblocks(1000000) if a.config == "bam" else cb.vcf_vblocks(4)
    wl = cb.Workload(E, vbs, device)
    codecs = wl.assign()
    wl.build()
    for _ in range(max(1, a.warmup)):
        wl.generate(); wl.compress(); E.sync()
    E.profile(True, reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        wl.generate(); wl.compress(); E.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    E.profile(False)
    prof = E.profile_results()
    prof_max = dict(E.profile_max)
    ms = dt / a.steps * 1e3
    z_list = [E.mem.download(vb.z, int(wl.vtab[i].z_len)) for i, vb in enumerate(wl.vbs)]
    z_total = sum(len(z) for z in z_list)
    dom = max(prof, key=lambda k: prof[k][0])
    dom_ms, dom_n = prof[dom]
    alg = wl.stream_bytes + z_total
    per_launch = alg / (dom_n / a.steps)
    ach = per_launch / (dom_ms / dom_n / 1e3) / 1e9
    out = {"metric": METRIC, "value": round(wl.stream_bytes / 1e6 / (ms / 1e3), 1), "unit": "MB/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": ("BAM-1M (BASELINE configs[2]): 22 VBlocks of 46 000 aligned reads - CIGAR / FLAG / MAPQ b250, binned QUAL local, POS u32 local" if a.config == "bam" else
                                   "VCF 10 k samples (BASELINE configs[3]), one GPU's share at 8 GPUs: 4 VBlocks of 3 000 lines x 10 000 samples - FORMAT/DP u8 matrix (transposed), FORMAT/PL b250")
                                  + "; entered at the CONTEXT-STREAM level (--stream-level; the default legs start from text): MB counted = bytes of context streams",
                      "n_vblocks": wl.n_vb, "stream_mb_per_step": round(wl.stream_bytes / 1e6, 1), "text_mb_approx": round(text_bytes / 1e6), "compressed_mb_per_step": round(z_total / 1e6, 2),
                      "codecs": {k: CODEC_NAMES[v] for k, v in codecs.items()}},
           "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": None,
                        "avg_launch_ms": round(dom_ms / dom_n, 4), "launches_per_step": dom_n / a.steps, "longest_launch_ms": round(prof_max.get(dom, 0), 3),
                        "kernel_ms_per_step_summed_over_concurrent_launches": {k: round(v[0] / a.steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:12]}}}
    if not a.no_cpu:
        c, exact = cb.cpu_baseline(z_list, usable_cpus()[0])
        c["unit"] = "MB/s"
        out["cpu_baseline"] = c
        out["bit_exact"] = exact
    print(json.dumps(out))


def main():
    a = parse_args()
    if a.qual is None:
        a.qual = "bin" if a.config == "bam" else "div"
    if a.config == "bam" and not a.stream_level:
        return sam_leg(a)
    if a.config == "vcf" and not a.stream_level:
        return vcf_leg(a)
    if a.config != "fastq":
        return config_leg(a)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(a)
    import numpy as np   # noqa: F401
    import torch
    from genozip_amd.codec import Engine
    from genozip_amd.lib import CODEC_NAMES

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # the other BASELINE configurations, each a process of its own, BEFORE this process touches the device: a process that merely holds a HIP
    # context (streams, hardware queues) beside them costs them 2 - 10 % (measured: the streamed leg 141 -> 155 ms behind the headline's process)
    legs = None
    if (world == 1 and not a.no_cpu and not a.stream_reads and a.qual == "div" and a.pairs == 1000000 and not a.vb_mb and "WORLD_SIZE" not in os.environ
            and not os.environ.get("GZ_BENCH_NO_SIDE_LEGS") and not os.environ.get("GZ_BENCH_EMUL")):
        legs = side_legs()
    # GZ_BENCH_EMUL=1 (tests/test_shard.py, no GPU): the same script on CPU ranks - the product's sources on the CPU stand-in of the HIP runtime
    # (tests/emul), torch tensors in host memory, the gloo backend. It exercises the N > 1 plumbing of this file, not a measurement.
    emul = bool(os.environ.get("GZ_BENCH_EMUL"))
    dist = None
    # GZ_BENCH_FORCE_DIST=1 (tests/test_gpu.py, launched under torch.distributed.run with ONE rank): the process group is made and every
    # exchange of the N-GPU path is taken although there is nobody to exchange with (genozip_amd/shard.py::FORCE_AT_WORLD_1) - RCCL sees the
    # code on the one GPU a test box has
    force_dist = bool(os.environ.get("GZ_BENCH_FORCE_DIST")) and "WORLD_SIZE" in os.environ
    if world > 1 or force_dist:
        import torch.distributed as dist
        if emul:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL over xGMI
        if force_dist and world == 1:
            from genozip_amd import shard as _sh
            _sh.FORCE_AT_WORLD_1 = True
        try:                                                      # (RCCL announces its version through C stdio: out with it now, so that the JSON line stays the LAST line of stdout)
            import ctypes
            dist.barrier()
            ctypes.CDLL(None).fflush(None)
        except Exception:                                         # noqa: BLE001
            pass
    if emul:
        sys.path.insert(0, os.path.join(ROOT, "tests", "emul"))
        from hostmem import TorchCpuMem
        device = torch.device("cpu")
        torch.cuda.synchronize = lambda *x, **k: None
        E = Engine(lib_path=os.path.join(ROOT, "tests", "emul", "libgenozip_amd_emul.so"), mem=TorchCpuMem())
    else:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        E = Engine(device=local_rank)
    from genozip_amd import shard as shard_mod
    wl = Workload(E, a, rank, world, device)
    RB, L = wl.W.RECORD_BYTES, wl.W.READ_LEN
    per_call_text = wl.text_len
    wl.text_bytes = per_call_text * wl.calls_per_step
    wl.value_bytes = wl.text_bytes - wl.n_reads_own * wl.calls_per_step * (L + 1)          # the text without its SEQ lines

    pending = [None]

    def gather_wait():
        if pending[0] is not None:
            pending[0].wait()
            pending[0] = None

    def gather_to_rank0(total):
        # the final exchange of the path: compressed VBlocks go to the writer rank. The gather is only STARTED:
        # the transfer over xGMI runs beside the next step's kernels; every gather is complete before the timed region ends
        if dist is None:
            return
        from genozip_amd.shard import gather_blobs
        gather_wait()
        blobs = [wl.zbuf[:total]]
        if os.environ.get("GZ_SYNC_GATHER"):
            gather_blobs(dist, blobs, rank, world, device)
        else:
            pending[0] = gather_blobs(dist, blobs, rank, world, device, async_op=True)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The headline is the COLD file: every step compresses a new file that knows nothing of the previous one - what the handle learned
    # from its previous file (the coder its QUAL streams got, gz_zip_speculation) is not used in the timed region: the long streams
    # start with the library's built-in prior (an order-1 adaptive coder), as the first file on a fresh handle does, and the file's own
    # trial compressions confirm or refute it. The warm figure (a service compressing file after file of the same kind) is measured
    # afterwards and reported beside it.
    prior_user = os.environ.get("GZ_ZIP_PRIOR_ONLY")
    os.environ["GZ_ZIP_PRIOR_ONLY"] = "1"
    for _ in range(a.warmup):
        gather_to_rank0(wl.step(dist))
    gather_wait()
    def one_step():
        gather_to_rank0(wl.step(dist))

    def drain():
        gather_wait()
        barrier()

    def timed_sync():                                          # (the statistics of the exchange count the timed steps only)
        drain()
        if not timed_sync.armed:
            shard_mod.reset_stats(); timed_sync.armed = True
    timed_sync.armed = False
    dt, prof, prof_max, prof_table = profiled_steps(E, one_step, a.steps, timed_sync)
    coll = dict(shard_mod.STATS)
    warm_ms = None
    del os.environ["GZ_ZIP_PRIOR_ONLY"]
    if prior_user is not None and not a.warm_steps:
        os.environ["GZ_ZIP_PRIOR_ONLY"] = prior_user
    if a.warm_steps and not a.stream_reads:
        gather_to_rank0(wl.step(dist))                       # (the step that teaches the handle the coder)
        gather_wait(); barrier()
        t1 = time.perf_counter()
        for _ in range(a.warm_steps):
            gather_to_rank0(wl.step(dist))
        gather_wait(); barrier()
        warm_ms = (time.perf_counter() - t1) / a.warm_steps * 1e3

    # SURVEY 8d names two quality profiles for this file: the headline is the 40-level one (Q-div: plain arithmetic-coded streams, the long
    # pole of the step); the binned one (Q-bin: NovaSeq's 4 levels, QUAL through DOMQ) is measured beside it on the 1-GPU run
    other = None
    if world == 1 and not a.stream_reads and a.qual == "div" and not a.no_cpu:
        import copy
        try:                                                      # (a side figure: it must never cost the run its headline)
            b = copy.copy(a); b.qual = "bin"
            wl2 = Workload(E, b, rank, world, device)
            prior_was = os.environ.get("GZ_ZIP_PRIOR_ONLY")
            os.environ["GZ_ZIP_PRIOR_ONLY"] = "1"
            wl2.step(None)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(3):
                wl2.step(None)
            torch.cuda.synchronize()
            ms2 = (time.perf_counter() - t1) / 3 * 1e3
            val2 = (wl2.text_len - wl2.n_reads_own * (L + 1)) / 1e6 / (ms2 / 1e3)
            other = {"qual_profile": "bin", "ms_per_step": round(ms2, 3), "value": round(val2, 1), "unit": "MB/s", "steps": 3,
                     "note": "the same file with SURVEY 8d's binned quality profile (cold, as the headline); its bit-exactness is what tests/test_gpu.py checks"}
            del wl2
        except Exception as e:                                    # noqa: BLE001
            other = {"qual_profile": "bin", "error": repr(e)}
        finally:
            if prior_was is None:
                os.environ.pop("GZ_ZIP_PRIOR_ONLY", None)
            else:
                os.environ["GZ_ZIP_PRIOR_ONLY"] = prior_was

    # N > 1, the default (strong) run: configs[1] is ONE small file whose step is the latency of one VBlock's range-coder chain - dealing its
    # VBlocks out cannot make that shorter (expect ~1.0 - 1.1 x at any N). What DOES scale with the number of GPUs is the
    # streamed configuration (configs[4]: every GPU streams its share of one 600 M-read file in calls of 112 VBlock pairs, no exchange but the
    # gather of z_data): measured beside the headline, every rank a stream of its own, so that one SCALE run shows both.
    streamed = None
    if world > 1 and not a.stream_reads and a.scaling == "strong" and not os.environ.get("GZ_BENCH_NO_STREAM_SHARE"):
        import copy
        ok, sv, st_ms, n_call_pairs = 1.0, 0.0, 0.0, a.batch_pairs
        try:
            b = copy.copy(a); b.scaling = "weak"; b.stream_reads = int(os.environ.get("GZ_BENCH_STREAM_SHARE_READS", "8000000")); b.vb_mb = a.vb_mb if emul else 0
            wl3 = Workload(E, b, rank, world, device)
            n_call_pairs = len(wl3.ranges)
            wl3.text_bytes = wl3.text_len * wl3.calls_per_step
            wl3.value_bytes = wl3.text_bytes - wl3.n_reads_own * wl3.calls_per_step * (L + 1)
            gather_wait()
            wl_keep, wl = wl, wl3                                   # (gather_to_rank0 reads wl.zbuf)
            gather_to_rank0(wl3.step(None)); gather_wait(); barrier()
            t1 = time.perf_counter()
            for _ in range(2):
                gather_to_rank0(wl3.step(None))
            gather_wait(); barrier()
            st_ms = (time.perf_counter() - t1) / 2 * 1e3
            sv = wl3.value_bytes
            wl = wl_keep
            del wl3
        except Exception as e:                                    # noqa: BLE001 (a side figure: it must never cost the run its headline)
            ok = 0.0
            sys.stderr.write("rank %d: streamed share failed: %r\n" % (rank, e))
        t = torch.tensor([ok, sv, st_ms], dtype=torch.float64, device=device)
        tm = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM); dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        if float(t[0]) == world and float(tm[2]) > 0:
            streamed = {"value": round(float(t[1]) / 1e6 / (float(tm[2]) / 1e3), 1), "unit": "MB/s", "ms_per_step": round(float(tm[2]), 3), "steps": 2, "scaling": "weak",
                        "workload": "configs[4] at reduced scale: every GPU streams %d read pairs of its own through one file object in calls of %d VBlock pairs (16 MiB VBlocks), "
                                    "z_data gathered to rank 0; whole-job text-without-SEQ MB/s, max over ranks of the time" % (b.stream_reads, n_call_pairs)}
        else:
            streamed = {"error": "a rank failed (see stderr)"}

    # per-rank byte counts -> whole-job sums
    z_total = wl.offs[-1]
    zhost = wl.zbuf[:z_total].cpu().numpy().tobytes()
    z_all = [zhost[wl.offs[i]:wl.offs[i + 1]] for i in range(len(wl.vb))]
    stream_bytes = sum(s[3] for z in z_all for s in walk_sections(z)) * wl.calls_per_step
    sums = torch.tensor([dt, wl.text_bytes, wl.value_bytes, stream_bytes, z_total * wl.calls_per_step, warm_ms or 0.0], dtype=torch.float64, device=device)
    if dist is not None:
        mx = torch.stack([sums[0], sums[5]])
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        sums[0] = mx[0]; sums[5] = mx[1]
    dt, text_b, value_b, stream_b, z_b, warm_ms_all = [float(x) for x in sums.cpu()]
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = dt / a.steps * 1e3
    value = value_b / 1e6 / (ms_per_step / 1e3)

    # ---- roofline of the dominant kernel, from HIP events on the library's streams (gz_profile)
    dom = max(prof, key=lambda k: prof[k][0])
    dom_ms, dom_launches = prof[dom]
    per_step_launches = dom_launches / a.steps
    avg_launch_ms = dom_ms / dom_launches
    alg_bytes_per_step = (stream_bytes + z_total * wl.calls_per_step)             # rank 0: N_in + N_out of every stream
    alg_per_launch = alg_bytes_per_step / per_step_launches
    achieved = alg_per_launch / (max(avg_launch_ms, 1e-6) / 1e3) / 1e9
    tr = pmc_traffic(dom, a)
    # The critical path. The dominant kernel is launched several times per step on different streams (the persistent launch over the long
    # QUAL streams + the short-leaf launches of trials and section writer): its launches overlap, their sum is NOT time on the step's
    # critical path - the LONGEST launch is. That launch codes the long streams (sections of >= 1 MB); what bounds it is the issue rate
    # of one wave per stream, not HBM: three dependent vector instructions per symbol (12.07 clocks = 5.03 ns at 2.4 GHz: the floor) + a
    # lane hop every 16 symbols, the checkpoints and the operand loads in the hops' wait states = 13.00 clocks = 5.41 ns with the loop alone
    # on the device (tools/ubench_chain_f64.hip; round 5: 15.1, round 4: 15.8, rounds 1-3: seven scalar integer instructions, 12.7 ns).
    # ns_per_symbol below is the whole launch / the symbols of its longest stream: it includes the wait for the first chunk's models.
    secs_all = [s for z in z_all for s in walk_sections(z)]
    long_secs = [s for s in secs_all if s[3] >= (1 << 20) and s[1] in (16, 17, 18, 19)]
    longest_ms = prof_max.get(dom, avg_launch_ms)
    crit = None
    if long_secs:
        sym = max(s[3] for s in long_secs)
        bytes_long = sum(s[3] + len(s[4]) for s in long_secs) * wl.calls_per_step
        crit = {"kernel": dom, "longest_launch_ms": round(longest_ms, 3), "streams_in_it": len(long_secs), "symbols_of_longest_stream": sym,
                "ns_per_symbol": round(longest_ms * 1e6 / sym, 2), "issue_rate_floor_ns_per_symbol": CHAIN_FLOOR_NS, "loop_alone_ns_per_symbol": CHAIN_ALONE_NS,
                "issue_rate_frac": round(CHAIN_FLOOR_NS / (longest_ms * 1e6 / sym), 3),
                "alg_bytes_in_it": bytes_long, "hbm_achieved_gbs": round(bytes_long / (longest_ms / 1e3) / 1e9, 3),
                "hbm_frac": round(bytes_long / (longest_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 6),
                "share_of_step": round(longest_ms / ms_per_step, 3)}
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None if tr is None else int(tr / per_step_launches),
                "avg_launch_ms": round(avg_launch_ms, 4), "launches_per_step": per_step_launches,
                "alg_bytes_per_launch": int(alg_per_launch),
                "critical_path": crit,
                "whole_step": {"alg_bytes": int(text_b / world + z_total * wl.calls_per_step), "achieved_gbs": round((text_b / world + z_total * wl.calls_per_step) / (ms_per_step / 1e3) / 1e9, 2),
                               "frac": round((text_b / world + z_total * wl.calls_per_step) / (ms_per_step / 1e3) / 1e9 / HBM_PEAK_GBS, 6), "note": "T + Z of one step / step time (SURVEY 8d, per pipeline unit)"},
                "kernel_ms_per_step_summed_over_concurrent_launches": kernel_table(prof, prof_table, a.steps, 24),
                "note": "kernel_ms_per_step_summed_over_concurrent_launches adds up launches that run side by side on different streams: a sum of device time, not a critical path; "
                        "k_arith_chain / k_arith_model: HIP events over the timed steps, the other kernels: one untimed step in front of them (events on every launch cost the host "
                        "as much as the launches do: GZ_BENCH_PROF_MODE=1 times everything inside the timed region, as rounds 1 - 5 did)"}

    codecs = {}
    for z in z_all[:1] + z_all[len(z_all) // 2:len(z_all) // 2 + 1]:
        for st, codec, did, ulen, pay, _domq in walk_sections(z):
            tag = next((c["tag"] for c in wl.plan["ctxs"] if c["dict_id"] == did), did.hex())
            codecs.setdefault(("b250:" if st == 11 else "local:") + tag, CODEC_NAMES.get(codec, str(codec)))
    mode = ("stream of %d read pairs per GPU in calls of %d VBlock pairs" % (a.stream_reads, len(wl.ranges))) if a.stream_reads else \
           ("FASTQ-PE-1M: ONE file pair (2 x %d reads x 150 bp), its %d VBlock pairs dealt out over the GPUs" % (a.pairs, wl.n_pairs_file)) if (a.scaling == "strong" and world > 1) else \
           ("FASTQ-PE-1M per GPU (2 x %d reads x 150 bp), %d VBlock pairs" % (a.pairs, wl.n_pairs_file))
    out = {"metric": METRIC, "value": round(value, 1), "unit": "MB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
           "dtype": "u8", "data": "synthetic",
           "config": {"workload": mode + ", VBlocks of %.2f MB (%s); the WHOLE path per step from FASTQ text in HBM: lines / reads / line-1 items -> seg columns (a1-a3) -> "
                                   "host dictionary merge in C (a4) -> b250 / local generation (a5-a7) -> codec assignment (a8) -> sections in DEP / did_i order (a15) -> "
                                   "rANS / arith + framing (a9-a13, a16); a new file every step. MB counted in `value` = text WITHOUT the SEQ lines "
                                   "(SEQ is 2-bit packed in the step and handed to the host's LZMA, which is outside the path, SURVEY F8)" % (vb_bytes(a) / 1e6, "--vb-mb" if a.vb_mb else "the reference's own rule for this file, src/segconf.c:152-206"),
                      "qual_profile": a.qual, "vb_bytes": vb_bytes(a), "codecs": codecs,
                      "qual_codec_speculation": "hits %d, misses %d since the handle was made (warm-up included): the long QUAL streams of a new file start with the codec "
                                                "the handle's previous file got; the file's own trial compressions still run in the step and decide (gz_zip_speculation)" % wl.F.speculation(),
                      "codec_prediction": "hits %d, misses %d since the handle was made: sections coded ahead of their context's trial compressions with a predicted codec (cold: the built-in prior by "
                                          "kind of stream), kept / coded again once the trials had decided (gz_zip_prediction)" % wl.F.prediction(),
                      "text_mb_per_step": round(text_b / 1e6, 1), "stream_mb_per_step": round(stream_b / 1e6, 1), "compressed_mb_per_step": round(z_b / 1e6, 2),
                      "parallelism": "vblocks sharded over %d GPU(s), no data-path collective; host exchange of new dictionary words (strong scaling only); RCCL gather of z_data" % world},
           "text_mb_s": round(text_b / 1e6 / (ms_per_step / 1e3), 1), "stream_mb_s": round(stream_b / 1e6 / (ms_per_step / 1e3), 1),
           "headline": "cold: a new file every step, nothing learned from the previous file is used (GZ_ZIP_PRIOR_ONLY: the long QUAL streams start with the built-in prior coder, the file's own trial decides)",
           "warm": None if not warm_ms_all else {"ms_per_step": round(warm_ms_all, 3), "value": round(value_b / 1e6 / (warm_ms_all / 1e3), 1), "steps": a.warm_steps,
                                                 "note": "the handle remembers the previous file's QUAL coder and starts the long streams with it (gz_zip_speculation)"},
           "roofline": roofline}
    if dist is not None:
        steps = max(1, a.steps)
        out["rccl"] = {"world": world, "backend": dist.get_backend(), "rank": 0,
                       "bytes_gathered_per_step": int(coll.get("gather_bytes", 0) / steps), "gather_host_ms_per_step": round(coll.get("gather_ms", 0.0) / steps, 3),
                       "gathers_per_step": coll.get("gathers", 0) / steps,
                       "exchange_bytes_per_step": int(coll.get("exchange_bytes", 0) / steps), "exchange_ms_per_step": round(coll.get("exchange_ms", 0.0) / steps, 3),
                       "exchanges_per_step": coll.get("exchanges", 0) / steps,
                       "phases_host_ms_per_step": {"seg": round(coll.get("seg_phase_ms", 0.0) / steps, 3),
                                                   "merge_replayed_for_all_vblocks": round(coll.get("merge_phase_ms", 0.0) / steps, 3),
                                                   "finish": round(coll.get("finish_phase_ms", 0.0) / steps, 3)},
                       "note": "rank 0's figures. gather: z_data of every rank to the writer rank, point-to-point at exact sizes, started asynchronously (host time = packing + posting); "
                               "exchange (strong scaling only): merge blobs and codec votes as byte tensors, all_gather; the merge phase replays ALL VBlocks' merges on every rank"}
        if a.scaling == "strong" and not a.stream_reads:
            out["expectation"] = ("strong scaling of configs[1] is ~1.0 - 1.1 x at any N: the step is the latency of ONE VBlock's range-coder chain (6.0 M symbols x 6.9 ns), "
                                  "which dealing VBlocks out does not shorten (DESIGN.md section 5); `streamed_share` is the configuration that scales")
        if streamed:
            out["streamed_share"] = streamed
    if other:
        out["other_profile"] = other
    if not a.no_cpu and world == 1:                # (the CPU pool is timed on rank 0 of the 1-GPU run only)
        cb, exact = cpu_leg(wl, z_all, usable_cpus()[0], E)
        out["cpu_baseline"] = cb
        out["bit_exact"] = exact
        out["gpu_over_cpu"] = gpu_over_cpu(out, cb)
        out["decode"] = decode_leg(E, wl, z_all)
        if out["decode"]["equals_reference_decoder"] is False:
            out["bit_exact"] = False
        out["bit_exact_payloads"] = out["bit_exact"]
        out["file_exact"] = file_exact_leg(wl, z_all)
        if out["file_exact"].get("checked"):
            out["bit_exact"] = bool(out["bit_exact"] and out["file_exact"]["identical"] == out["file_exact"]["vblocks"])
        if legs is not None:
            out["configs"] = legs
    if dist is not None:
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:                                         # noqa: BLE001
            pass
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
        os._exit(0)                                               # (nothing behind the JSON line: libraries' exit-time chatter stays in their buffers)


if __name__ == "__main__":
    main()
