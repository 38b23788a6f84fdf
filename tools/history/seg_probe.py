"""GPU: throughput of the seg-side column kernels (rows a1-a3) on FASTQ-PE-1M-shaped input: 176 VBlocks x 8 QNAME
token columns of ~23 000 reads each through gz_ctx_seg_columns in ONE call, and the SEQ / QUAL gather of every read
into its context's local through gz_local_blob_columns. Usage: python tools/seg_probe.py [n_vb] [reads_per_vb]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from genozip_amd.codec import Engine  # noqa: E402
from genozip_amd import synth  # noqa: E402


def main():
    n_vb = int(sys.argv[1]) if len(sys.argv) > 1 else 176
    reads = int(sys.argv[2]) if len(sys.argv) > 2 else 11400
    E = Engine(device=0)
    # one VBlock's text: reads x (qname line, SEQ, +, QUAL); the same text is used for every VBlock (different buffers)
    r = synth.u32(5, reads * 4).astype(np.int64)
    lines, cols, seq_off, seq_len = [], [[] for _ in range(8)], [], []
    at = 0
    parts = []
    for i in range(reads):
        toks = [b"@A00123", b"45", b"HXXXXXXXX", b"%d" % (1 + i * 4 // reads), b"%d" % (1101 + i * 70 // reads),
                b"%d" % (1000 + r[i] % 30000), b"%d" % (1000 + (i * 37) % 36000), b"1:N:0:ACGTACGT+TGCATGCA"]
        seps = [b":"] * 6 + [b" ", b"\n"]
        for c, (t, s) in enumerate(zip(toks, seps)):
            cols[c].append((at, len(t)))
            parts.append(t + s); at += len(t) + 1
        seq = synth.uniform_bytes(i, 150, 4).tobytes().translate(bytes.maketrans(b"\x00\x01\x02\x03", b"ACGT"))
        seq_off.append(at); parts.append(seq + b"\n+\n"); at += 153
        cols_q = at
        parts.append(bytes([33 + (x % 40) for x in synth.uniform_bytes(i + reads, 150, 40)]) + b"\n"); at += 151
        seq_len.append(cols_q)
    text = b"".join(parts)
    tbufs = [E.mem.upload(text) for _ in range(n_vb)]
    colarr = [(np.array([o for o, _ in c], dtype=np.uint32), np.array([l for _, l in c], dtype=np.uint32)) for c in cols]
    jobs = [(tb, o, l, []) for tb in tbufs for (o, l) in colarr]
    so = np.array(seq_off, dtype=np.uint32); qo = np.array(seq_len, dtype=np.uint32); l150 = np.full(reads, 150, dtype=np.uint32)
    for rep in range(3):
        E.mem.sync(); t0 = time.time()
        outs = E.ctx_seg_columns(jobs, keep_on_device=True)
        E.mem.sync(); t1 = time.time()
        blobs = E.local_blob_columns([(tb, so, l150, False) for tb in tbufs] + [(tb, qo, l150, False) for tb in tbufs])
        E.mem.sync(); t2 = time.time()
        print("rep %d: %d columns x %d snips: %.1f ms (incl. host tables + uploads); %d blobs of %d B: %.1f ms; text %.1f MB"
              % (rep, len(jobs), reads, (t1 - t0) * 1e3, 2 * n_vb, 150 * reads, (t2 - t1) * 1e3, n_vb * len(text) / 1e6))
    big = E.mem.upload(text * n_vb)
    E.profile(True, True)
    tb_, ob_, lb_, rb_, n_lines = E.text_lines(big, cap=4 * reads * n_vb + 8, on_device=True)
    assert n_lines == 4 * reads * n_vb
    E.ctx_seg_columns(jobs, keep_on_device=True)
    E.local_blob_columns([(tb, so, l150, False) for tb in tbufs] + [(tb, qo, l150, False) for tb in tbufs])
    for k, (ms, n) in sorted(E.profile_results().items(), key=lambda kv: -kv[1][0]):
        print("  %-18s %8.3f ms  x%d" % (k, ms, n))
    # one big column
    n = 30000000
    words = [b"%d,%d,%d" % (i % 97, (i * 7) % 255, (i * 13) % 255) for i in range(2000)]
    wt = b"".join(words)
    lens = np.array([len(w) for w in words], dtype=np.uint32)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint32)
    rr = synth.u32(99, n).astype(np.int64)
    pick = np.minimum(rr % 2000, (rr >> 11) % 2000)
    o, l = starts[pick], lens[pick]
    E.profile(True, True)
    t0 = time.time(); E.ctx_seg_columns([(wt, o, l, words[::2])], keep_on_device=True); t1 = time.time()
    print("FORMAT/PL-like column of %d snips: %.1f ms wall (incl. uploads)" % (n, (t1 - t0) * 1e3))
    for k, (ms, nn) in sorted(E.profile_results().items(), key=lambda kv: -kv[1][0]):
        print("  %-18s %8.3f ms  x%d" % (k, ms, nn))


if __name__ == "__main__":
    main()
