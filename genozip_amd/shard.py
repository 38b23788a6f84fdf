"""Multi-GPU plumbing of the path: VBlocks are independent, so they are dealt out to one process per
GPU with no collective on the data path; the only exchange is the hand-over of the finished, variable-length z_data
blobs to the writer rank -- torch.distributed (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU
tests): an all_gather of the byte counts, then one gather of the padded payloads."""
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


def gather_blobs(dist, blobs, rank, world, device, dst=0, async_op=False):
    """blobs: list of 1-D uint8 tensors on `device` (this rank's compressed VBlocks, in order).
    Returns on dst: list (per rank) of lists of byte strings' tensors; elsewhere None.
    async_op: the payload is first packed into a staging buffer of its own, the gather is only started, and a
    PendingGather is returned: the transfer over xGMI then runs beside the next batch's kernels - wait() before starting
    the next gather. The packing copy runs on torch's current stream, the library writes z_data on its own stream and the
    two are not ordered with each other: this function therefore waits for the copy before it returns, so that the blobs
    really may be overwritten (by any stream) as soon as it has returned."""
    lens_host = [int(b.numel()) for b in blobs]
    lens = torch.tensor(lens_host, dtype=torch.int64, device=device)
    n_local = torch.tensor([len(blobs), sum(lens_host)], dtype=torch.int64, device=device)
    counts = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts_host = torch.stack(counts).cpu().numpy()                 # ONE read-back, not one per element
    max_n, max_bytes = int(counts_host[:, 0].max()), int(counts_host[:, 1].max())
    lens_pad = torch.zeros(max(1, max_n), dtype=torch.int64, device=device)
    lens_pad[:len(blobs)] = lens
    pay = torch.zeros(max(1, max_bytes), dtype=torch.uint8, device=device)
    if len(blobs):
        torch.cat(blobs, out=pay[:sum(lens_host)])
        if pay.is_cuda:
            torch.cuda.current_stream(pay.device).synchronize()      # the staging copy has read the blobs
    len_bufs = [torch.empty_like(lens_pad) for _ in range(world)] if rank == dst else None
    pay_bufs = [torch.empty_like(pay) for _ in range(world)] if rank == dst else None
    w1 = dist.gather(lens_pad, len_bufs, dst=dst, async_op=async_op)
    w2 = dist.gather(pay, pay_bufs, dst=dst, async_op=async_op)

    def finish():
        if rank != dst:
            return None
        lens_all = torch.stack(len_bufs).cpu().numpy()              # (the writer needs the lengths on the host anyway)
        out = []
        for r in range(world):
            n = int(counts_host[r, 0])
            ends = lens_all[r, :n].cumsum()
            out.append([pay_bufs[r][int(e - ln):int(e)] for e, ln in zip(ends, lens_all[r, :n])])
        return out

    if async_op:
        return PendingGather([w1, w2], (lens_pad, pay, len_bufs, pay_bufs), finish)
    return finish()


def pairs_of_rank(n_pairs, rank, world):
    """VBlock pair k (R1 VBlock k and R2 VBlock k of a paired FASTQ; a single VBlock for an unpaired file) -> rank k % world:
    the dealing of ONE file over the GPUs (strong scaling). R2 consults R1's sections, so the two stay together
    (src/fastq.c:956-976, src/zip.c:613-620)."""
    return [k for k in range(n_pairs) if k % world == rank]


def zip_vblocks_sharded(zf, dist, text_buf, text_len, tab, n):
    """gz_fastq_zip_vblocks for a file whose VBlocks are dealt out over the ranks of `dist` (None: one process): the ordered
    dictionary merge is the one exchange step on the way - every rank contributes the new words of its VBlocks
    (a few KB), all ranks replay the merge of the whole call in vblock_i order and so hold identical dictionaries; codec
    choices for contexts the file has none for yet are exchanged the same way (lowest vblock_i wins, as in a serial run).
    The payload bytes never leave their GPU before the final gather (gather_blobs)."""
    blob = zf.seg(text_buf, text_len, tab, n)
    if dist is None or dist.get_world_size() == 1:
        zf.finish([zf.merge([blob])])
        return
    world = dist.get_world_size()
    blobs = [None] * world
    dist.all_gather_object(blobs, blob)
    votes = zf.merge(blobs)
    all_votes = [None] * world
    dist.all_gather_object(all_votes, votes)
    zf.finish(all_votes)
