"""(run by tests/test_gpu.py::test_rccl_sees_the_sharding_code in a process of its own) genozip_amd/shard.py's exchanges on HBM tensors over
the nccl backend (= RCCL) in a group of ONE rank: the all_gathers of byte strings, the gather to the writer rank with its loop-back
send / recv pair, synchronous and asynchronous. Prints OK."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genozip_amd import shard   # noqa: E402


def main():
    port = sys.argv[1]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % port, rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    shard.FORCE_AT_WORLD_1 = True
    shard.reset_stats()
    blob = bytes(range(256)) * 37
    assert shard.all_gather_bytes(dist, blob, dev) == [blob]
    assert shard.all_gather_bytes(dist, b"", dev) == [b""]
    g = torch.Generator(device="cpu"); g.manual_seed(5)
    blobs = [torch.randint(0, 256, (n,), dtype=torch.uint8, generator=g).to(dev) for n in (1, 70000, 3, 1 << 20)]
    got = shard.gather_blobs(dist, blobs, 0, 1, dev)
    assert len(got) == 1 and all(torch.equal(a, b) for a, b in zip(got[0], blobs))
    pend = shard.gather_blobs(dist, blobs, 0, 1, dev, async_op=True)
    for b in blobs:
        b.zero_()                                  # (the blobs may be overwritten as soon as the call has returned)
    got = pend.wait()
    assert [int(x.numel()) for x in got[0]] == [1, 70000, 3, 1 << 20] and int(got[0][3].sum()) > 0
    assert shard.STATS["gathers"] == 2 and shard.STATS["gather_bytes"] == 2 * (1 + 70000 + 3 + (1 << 20)) and shard.STATS["exchanges"] == 2
    dist.barrier()
    dist.destroy_process_group()
    print("OK")


if __name__ == "__main__":
    main()
