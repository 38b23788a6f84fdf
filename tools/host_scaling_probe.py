#!/usr/bin/env python3
"""What the GPU box's host really gives a process: logical CPUs, affinity, cgroup quota - and how the CPU legs of bench.py scale with the
number of threads there (the reference's htscodecs on quality-like streams; the whole-path composition on FASTQ VBlocks).
    python tools/host_scaling_probe.py > gpurun_out/host_scaling.txt"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np          # noqa: E402
import pyoracle             # noqa: E402
from genozip_amd import fastq as fq, workload as W, synth   # noqa: E402

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/memory.max"):
    try:
        print(p, open(p).read().strip())
    except OSError:
        pass
try:
    print([l for l in open("/proc/cpuinfo") if l.startswith("model name")][0].strip())
    print([l for l in open("/proc/meminfo")][0].strip())
except OSError:
    pass
R = pyoracle.Ref() if pyoracle.Ref.available() else pyoracle.Oracle()
O = pyoracle.Oracle()
q = synth.quality_diverse(3, 12000).tobytes()[:1500000]
for nt in (1, 8, 16, 32, 64, 128, 256):
    n = 4 * nt
    outs, dt = R.codec_compress_many([16] * n, [q] * n, nt) if isinstance(R, pyoracle.Ref) else (None, None)
    if dt is None:
        t0 = time.time(); O.codec_compress_many([16] * n, [q] * n, nt); dt = time.time() - t0
    print("codec ARTB %3d threads x 4 tasks of 1.5 MB: %.3f s  %.1f MB/s  (%.1f per thread)" % (nt, dt, n * len(q) / dt / 1e6, n * len(q) / dt / 1e6 / nt))
nr = 40000
t = np.frombuffer(W.fastq_text(1, 0, nr, mate=1, profile="div"), dtype=np.uint8)
plan = fq.illumina_plan(paired=True)
codecs = {("local", "QUAL"): 16, ("local", "Q1NAME"): 8, ("b250", "Q2NAME"): 16, ("local", "Q3NAME"): 17, ("local", "Q4NAME"): 17}
ref = R if isinstance(R, pyoracle.Ref) else None
for nt in (1, 8, 16, 32, 64, 128, 256):
    for reps in (1, 4):
        dt, _, _ = pyoracle.fastq_path_many(O, t, [(0, len(t))] * nt, plan, codecs, 0, nt, reps, ref)
        print("whole path %3d threads x %d VBlocks of 14.7 MB: %.3f s  %.1f MB/s text-without-SEQ (%.1f per thread)" % (nt, reps, dt, nt * reps * (len(t) - nr * 151) / dt / 1e6,
                                                                                                                     reps * (len(t) - nr * 151) / dt / 1e6))
