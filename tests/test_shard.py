"""CPU, world_size 2, gloo: the N>1 path of bench.py -- VBlock sharding and the gather of compressed VBlocks to the
writer rank (the only exchange step of the path)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from genozip_amd import synth
from genozip_amd.shard import gather_blobs, vblocks_of_rank


def _blob(v):
    return synth.uniform_bytes(1000 + v, 100 + 37 * v).tobytes()


def _worker(rank, world, port, n_vb, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = vblocks_of_rank(n_vb, rank, world, pair_size=2)
    blobs = [torch.frombuffer(bytearray(_blob(v)), dtype=torch.uint8) for v in mine]
    got = gather_blobs(dist, blobs, rank, world, torch.device("cpu"))
    # the overlapped form bench.py uses: start, overwrite the blobs (the next batch's output), wait
    pending = gather_blobs(dist, blobs, rank, world, torch.device("cpu"), async_op=True)
    for b in blobs:
        b.zero_()
    got2 = pending.wait()
    # bench.py's loop: step k's gather is waited for only when step k+1 wants to start its own
    pend, last = None, None
    for step in range(3):
        for i, v in enumerate(mine):
            blobs[i].copy_(torch.frombuffer(bytearray(_blob(v)), dtype=torch.uint8))
            blobs[i][0] = step                                   # "this step's output"
        if pend is not None:
            pend.wait()
        pend = gather_blobs(dist, blobs, rank, world, torch.device("cpu"), async_op=True)
    last = pend.wait()
    if rank == 0:
        ok = True
        for r in range(world):
            want = [_blob(v) for v in vblocks_of_rank(n_vb, r, world, pair_size=2)]
            ok &= [bytes(t.numpy().tobytes()) for t in got[r]] == want
            ok &= [bytes(t.numpy().tobytes()) for t in got2[r]] == want
            ok &= [bytes(t.numpy().tobytes()) for t in last[r]] == [bytes([2]) + w[1:] for w in want]
        q.put(ok)
    else:
        assert got is None and got2 is None
    dist.barrier()
    dist.destroy_process_group()


def test_sharding_covers_every_vblock_once():
    for world in (1, 2, 4, 8):
        for n in (0, 1, 7, 176):
            seen = sorted(v for r in range(world) for v in vblocks_of_rank(n, r, world, pair_size=2))
            assert seen == list(range(n))
        # R1/R2 VBlocks of a pair (2k, 2k+1) stay together
        for r in range(world):
            mine = set(vblocks_of_rank(176, r, world, pair_size=2))
            assert all((v ^ 1) in mine for v in mine)


def test_gather_to_writer_rank_gloo_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 11, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


# ---- real VBlocks under a process group: the file of test_emul_fastq_zip dealt out over 2 ranks -----------------------------
def _pair_file(n_reads, n_pairs, qual="uniform", tiny=False):
    """-> (text, [(text_off, text_len, vblock_i, r1 index)], vb_size) of a paired FASTQ in the reference's VBlock order: R1's VBlocks
    1..n_pairs, then R2's n_pairs+1..2 n_pairs, R2 VBlock k pairing R1 VBlock k. tiny: VBlocks of growing size (1 : 2 : 3 ...) and a
    vb_size by which the first of each mate is too small to set the file's codecs (codec.c:352)"""
    import numpy as np
    import parity
    r1 = parity.fastq_text(n_reads, seed=500, mate=1, qual=qual)
    r2 = parity.fastq_text(n_reads, seed=500, mate=2, qual_seed=900, qual=qual)
    vbs, text = [], r1 + r2
    for m, t in enumerate((r1, r2)):
        nl = np.flatnonzero(np.frombuffer(t, dtype=np.uint8) == 10)
        tri = n_pairs * (n_pairs + 1) // 2
        at = [n_reads * (k * (k + 1) // 2) // tri if tiny else n_reads * k // n_pairs for k in range(1, n_pairs)]
        cuts = [0] + [int(nl[4 * a - 1]) + 1 for a in at] + [len(t)]
        for k in range(n_pairs):
            vbs.append((m * len(r1) + cuts[k], cuts[k + 1] - cuts[k], m * n_pairs + k + 1, k if m else -1))
    return text, vbs, (len(r1) // 2 if tiny else 0)


def _zip_worker(rank, world, port, n_reads, n_pairs, q, qual="uniform", tiny=False):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "oracle"), here, os.path.join(here, "emul")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hostmem import HostMem
    from genozip_amd.codec import Engine
    from genozip_amd import fastq as fq
    from genozip_amd.shard import pairs_of_rank, zip_vblocks_sharded
    E = Engine(lib_path=os.path.join(here, "emul", "libgenozip_amd_emul.so"), mem=HostMem())
    text, vbs, vb_size = _pair_file(n_reads, n_pairs, qual, tiny)
    F = E.zip_open(fq.illumina_plan(paired=True, vb_size=vb_size))
    mine = pairs_of_rank(n_pairs, rank, world)
    # this rank's VBlocks in ascending vblock_i: its R1 VBlocks, then its R2 VBlocks naming them
    own = [vbs[k] for k in mine] + [vbs[n_pairs + k] for k in mine]
    own = [(o, l, vi, (mine.index(r1) if r1 >= 0 else -1)) for (o, l, vi, r1) in own]
    buf = E.mem.upload(text + b"\0" * 32)
    tab = F.vb_table(own)
    zip_vblocks_sharded(F, dist, buf, len(text), tab, len(own))
    res = {r["vblock_i"]: r["z"] for r in F.results(tab)}
    words = F.zctx_words(3)
    allres = [None] * world
    dist.all_gather_object(allres, (res, words))
    if rank == 0:
        q.put(allres)
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("qual,tiny", [("uniform", False), pytest.param("bin", False, marks=pytest.mark.thorough), ("uniform", True)], ids=["uniform", "bin", "tiny"])
def test_file_dealt_out_over_2_ranks_equals_one_rank(emul_engine, qual, tiny):
    """the N>1 form of the whole path (strong scaling: ONE file, its VBlock pairs dealt out): every rank segs and compresses its
    own VBlocks through the emulated build, the dictionary merge and the codec choices are exchanged - and every VBlock's z_data
    is byte-identical to what a single process makes of the same file. qual = "bin": the file's first VBlock (rank 0's) makes QUAL go
    through CODEC_DOMQ - rank 1 learns that in the merge, from rank 0's blob. tiny: VBlock 1 (rank 0's) is too small to set codecs
    for the file: it keeps its own, VBlock 2 (rank 1's) sets them for VBlock 3 (rank 0's) and the rest"""
    from genozip_amd import fastq as fq
    n_reads, n_pairs = 72, 3
    text, vbs, vb_size = _pair_file(n_reads, n_pairs, qual, tiny)
    F = emul_engine.zip_open(fq.illumina_plan(paired=True, vb_size=vb_size))
    buf = emul_engine.mem.upload(text + b"\0" * 32)
    tab = F.vb_table(vbs)
    F.zip_table(buf, len(text), tab, len(vbs))
    one = {r["vblock_i"]: r["z"] for r in F.results(tab)}
    words_one = F.zctx_words(3)
    F.close()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_zip_worker, args=(r, 2, port, n_reads, n_pairs, q, qual, tiny)) for r in range(2)]
    for p in procs:
        p.start()
    allres = q.get(timeout=300)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = {}
    for res, words in allres:
        assert words == words_one                       # identical dictionaries on every rank
        got.update(res)
    assert sorted(got) == sorted(one) == list(range(1, 2 * n_pairs + 1))
    for vi in one:
        assert got[vi] == one[vi], "VBlock %d differs between the 2-rank and the 1-rank run" % vi


def test_strong_scaling_at_one_rank_equals_weak(emul_engine):
    """N = 1: the strong-scaling form (three phases + the exchanges of merge blobs and codec votes, here within a process group of ONE
    rank) writes the bytes of the plain one-process call (the weak form's step): the curve bench.py --scaling strong starts from is
    the same work as its N = 1 weak point"""
    from genozip_amd import fastq as fq
    n_reads, n_pairs = 60, 2
    text, vbs, vb_size = _pair_file(n_reads, n_pairs)
    F = emul_engine.zip_open(fq.illumina_plan(paired=True, vb_size=vb_size))
    buf = emul_engine.mem.upload(text + b"\0" * 32)
    tab = F.vb_table(vbs)
    F.zip_table(buf, len(text), tab, len(vbs))
    one = {r["vblock_i"]: r["z"] for r in F.results(tab)}
    F.close()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_zip_worker, args=(0, 1, port, n_reads, n_pairs, q))
    p.start()
    allres = q.get(timeout=300)
    p.join(120)
    assert p.exitcode == 0 and len(allres) == 1
    assert allres[0][0] == one


def test_bench_gpus_2_under_gloo(emul_engine):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one rank per "GPU"), on CPU ranks: GZ_BENCH_EMUL=1 runs the same
    script over the product's sources on the CPU stand-in (tests/emul) with the gloo backend - the strong-scaling path (merge blobs and votes
    as byte tensors, the point-to-point gather of z_data), the `rccl` record and the streamed share beside the headline all execute; the
    numbers mean nothing here, the plumbing does"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, GZ_BENCH_EMUL="1", GZ_BENCH_STREAM_SHARE_READS="300")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "bench.py"), "--gpus", "2", "--pairs", "600", "--vb-mb", "0.05", "--steps", "1", "--warmup", "0", "--no-cpu", "--warm-steps", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert p.stdout.decode().strip().splitlines()[-1].startswith("{")          # (the JSON line is the LAST line of stdout: the driver reads it there)
    d = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["rccl"]["world"] == 2 and d["rccl"]["backend"] == "gloo"
    assert d["rccl"]["bytes_gathered_per_step"] > 0 and d["rccl"]["exchanges_per_step"] == 2 and d["rccl"]["exchange_bytes_per_step"] > 0
    assert "error" not in d["streamed_share"] and d["streamed_share"]["scaling"] == "weak" and "expectation" in d
