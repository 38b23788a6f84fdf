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
