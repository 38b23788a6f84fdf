"""Multi-GPU plumbing of the path: VBlocks are independent, so they are dealt out to one process per
GPU with no collective on the data path; the only exchange is the hand-over of the finished, variable-length z_data
blobs to the writer rank -- torch.distributed (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU
tests): an all_gather of the byte counts, then every rank sends its payload to the writer at its exact size. When ONE file is dealt out
(strong scaling) the merge blobs and codec votes are exchanged as byte tensors (all_gather_bytes)."""
import torch


def vblocks_of_rank(n_vblocks, rank, world, pair_size=1):
    """VBlock i -> rank (i // pair_size) % world. pair_size=2 keeps the R1/R2 VBlocks of a FASTQ pair on one GPU
    (R2 consults R1's sections, src/fastq.c:956-976)"""
    return [i for i in range(n_vblocks) if (i // pair_size) % world == rank]


class PendingGather:
    """a gather in flight: wait() completes it and (on dst) returns the per-rank lists of blob tensors"""

    def __init__(self, works, keep, finish):
        self.works, self.keep, self.finish = works, keep, finish

    def wait(self):
        for w in self.works:
            if w is not None:
                w.wait()
        self.works = []
        out = self.finish() if self.finish else None
        self.keep, self.finish = None, None          # the staging buffers may go now
        return out


# GZ_BENCH_FORCE_DIST / tests: take the exchange paths even in a group of ONE rank - the all_gathers of the merge blobs and votes run as
# collectives of the backend (nccl == RCCL on HBM tensors), and the gather to the writer rank sends the payload to ITSELF point-to-point
# (a loop-back ncclSend / ncclRecv pair, checked against what was sent): the one GPU a test box has then sees every call of the N-GPU path
FORCE_AT_WORLD_1 = False


# what the collectives of this module moved and how long the host waited for them, since reset_stats (): bench.py's "rccl" record
STATS = {}


def reset_stats():
    STATS.clear()
    STATS.update(gathers=0, gather_bytes=0, gather_ms=0.0, exchanges=0, exchange_bytes=0, exchange_ms=0.0, merge_phase_ms=0.0, seg_phase_ms=0.0, finish_phase_ms=0.0)


reset_stats()


def _now():
    import time
    return time.perf_counter()


def _coll_device(dist, device):
    """tensors of a collective live where the backend works: HBM for nccl (= RCCL), host memory for gloo (the CPU tests)"""
    return torch.device("cpu") if dist.get_backend() == "gloo" else device


def gather_blobs(dist, blobs, rank, world, device, dst=0, async_op=False):
    """blobs: list of 1-D uint8 tensors on `device` (this rank's compressed VBlocks, in order).
    Returns on dst: list (per rank) of lists of byte strings' tensors; elsewhere None.
    One all_gather of (count, bytes) per rank, then every rank SENDS its lengths and its payload to dst at their exact sizes
    (point-to-point: ncclSend / ncclRecv under the nccl backend - xGMI links are point-to-point anyway; nothing is padded to the largest
    rank's size). async_op: the payload is first packed into a staging buffer of its own, the transfers are only started, and a
    PendingGather is returned: they then run beside the next batch's kernels - wait() before starting the next gather. The packing copy
    runs on torch's current stream, the library writes z_data on its own stream and the two are not ordered with each other: this
    function therefore waits for the copy before it returns, so that the blobs really may be overwritten (by any stream) as soon as it
    has returned."""
    t0 = _now()
    cdev = _coll_device(dist, device)
    lens_host = [int(b.numel()) for b in blobs]
    n_local = torch.tensor([len(blobs), sum(lens_host)], dtype=torch.int64, device=cdev)
    counts = [torch.zeros(2, dtype=torch.int64, device=cdev) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts_host = torch.stack(counts).cpu().numpy()                 # ONE read-back, not one per element
    lens = torch.tensor(lens_host if lens_host else [0], dtype=torch.int64, device=cdev)
    pay = torch.empty(max(1, sum(lens_host)), dtype=torch.uint8, device=cdev)
    if len(blobs):
        torch.cat([b.to(cdev) for b in blobs], out=pay[:sum(lens_host)])
        if pay.is_cuda:
            torch.cuda.current_stream(pay.device).synchronize()      # the staging copy has read the blobs
    len_bufs = pay_bufs = None
    ops = []
    if rank == dst:
        len_bufs = [torch.empty(max(1, int(counts_host[r, 0])), dtype=torch.int64, device=cdev) if r != rank else lens for r in range(world)]
        pay_bufs = [torch.empty(max(1, int(counts_host[r, 1])), dtype=torch.uint8, device=cdev) if r != rank else pay for r in range(world)]
        for r in range(world):
            if r != rank and counts_host[r, 0]:
                ops.append(dist.P2POp(dist.irecv, len_bufs[r], r))
                if counts_host[r, 1]:
                    ops.append(dist.P2POp(dist.irecv, pay_bufs[r], r))
    elif len(blobs):
        ops.append(dist.P2POp(dist.isend, lens, dst))
        if sum(lens_host):
            ops.append(dist.P2POp(dist.isend, pay[:sum(lens_host)], dst))
    loop = None
    if FORCE_AT_WORLD_1 and world == 1 and sum(lens_host):          # the loop-back pair: the payload to this same rank through the backend's send / recv
        loop = torch.empty_like(pay)
        ops += [dist.P2POp(dist.isend, pay[:sum(lens_host)], rank), dist.P2POp(dist.irecv, loop[:sum(lens_host)], rank)]
    works = dist.batch_isend_irecv(ops) if ops else []
    moved = int(counts_host[:, 1].sum() - counts_host[dst, 1]) if rank == dst else sum(lens_host)
    if loop is not None:
        moved += sum(lens_host)

    def finish():
        STATS["gathers"] += 1; STATS["gather_bytes"] += moved
        if loop is not None:
            assert torch.equal(loop[:sum(lens_host)], pay[:sum(lens_host)]), "loop-back send / recv returned other bytes"
        if rank != dst:
            return None
        out = []
        for r in range(world):
            n = int(counts_host[r, 0])
            la = len_bufs[r].cpu().numpy()[:n]                      # (the writer needs the lengths on the host anyway)
            ends = la.cumsum()
            out.append([pay_bufs[r][int(e - ln):int(e)] for e, ln in zip(ends, la)])
        return out

    if async_op:
        STATS["gather_ms"] += (_now() - t0) * 1e3
        return PendingGather(works, (lens, pay, len_bufs, pay_bufs, loop), finish)
    for w in works:
        w.wait()
    if pay.is_cuda:
        torch.cuda.current_stream(pay.device).synchronize()
    STATS["gather_ms"] += (_now() - t0) * 1e3
    return finish()


def all_gather_bytes(dist, data, device):
    """every rank's byte string to every rank, as tensors: an all_gather of the lengths, then an all_gather of the payloads padded to the
    longest (the merge blobs and codec votes of a call: a few KB to tens of KB per rank - padding costs nothing here, and every rank needs
    every blob). -> list of bytes, by rank"""
    t0 = _now()
    import numpy as np
    world = dist.get_world_size()
    cdev = _coll_device(dist, device)
    n = torch.tensor([len(data)], dtype=torch.int64, device=cdev)
    ns = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(world)]
    dist.all_gather(ns, n)
    lens = [int(x) for x in torch.cat(ns).cpu().numpy()]
    cap = max(1, max(lens))
    mine = torch.zeros(cap, dtype=torch.uint8, device=cdev)
    if len(data):
        mine[:len(data)] = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(cdev)
    bufs = [torch.empty(cap, dtype=torch.uint8, device=cdev) for _ in range(world)]
    dist.all_gather(bufs, mine)
    out = [bufs[r][:lens[r]].cpu().numpy().tobytes() for r in range(world)]
    STATS["exchanges"] += 1; STATS["exchange_bytes"] += sum(lens) - len(data); STATS["exchange_ms"] += (_now() - t0) * 1e3
    return out


def pairs_of_rank(n_pairs, rank, world):
    """VBlock pair k (R1 VBlock k and R2 VBlock k of a paired FASTQ; a single VBlock for an unpaired file) -> rank k % world:
    the dealing of ONE file over the GPUs (strong scaling). R2 consults R1's sections, so the two stay together
    (src/fastq.c:956-976, src/zip.c:613-620)."""
    return [k for k in range(n_pairs) if k % world == rank]


def zip_vblocks_sharded(zf, dist, text_buf, text_len, tab, n, device=None):
    """gz_fastq_zip_vblocks for a file whose VBlocks are dealt out over the ranks of `dist` (None: one process): the ordered
    dictionary merge is the one exchange step on the way - every rank contributes the new words of its VBlocks
    (a few KB) as a byte tensor (all_gather_bytes), all ranks replay the merge of the whole call in vblock_i order and so hold identical
    dictionaries; codec choices for contexts the file has none for yet are exchanged the same way (lowest vblock_i wins, as in a serial
    run). The payload bytes never leave their GPU before the final gather (gather_blobs). The three phases' host times go to STATS."""
    t0 = _now()
    blob = zf.seg(text_buf, text_len, tab, n)
    t1 = _now(); STATS["seg_phase_ms"] += (t1 - t0) * 1e3
    if dist is None or (dist.get_world_size() == 1 and not FORCE_AT_WORLD_1):
        votes = zf.merge([blob])
        t2 = _now(); STATS["merge_phase_ms"] += (t2 - t1) * 1e3
        zf.finish([votes])
        STATS["finish_phase_ms"] += (_now() - t2) * 1e3
        return
    if device is None:
        device = text_buf.device if hasattr(text_buf, "device") else torch.device("cpu")
    blobs = all_gather_bytes(dist, blob, device)
    t2 = _now()
    votes = zf.merge(blobs)                                    # (the merge of ALL VBlocks of the call is replayed here: it does not shrink with N)
    t3 = _now(); STATS["merge_phase_ms"] += (t3 - t2) * 1e3
    all_votes = all_gather_bytes(dist, votes, device)
    t4 = _now()
    zf.finish(all_votes)
    STATS["finish_phase_ms"] += (_now() - t4) * 1e3
