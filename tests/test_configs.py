"""GPU (-m gpu): the streams BASELINE.json's BAM and VCF configs feed into the path, one VBlock's worth at
the sizes the reference would give them, against the oracle byte for byte (the oracle's C loops are fast enough for
these) plus the size-independent properties (transpose == numpy's, decode(encode(x)) == x)."""
import numpy as np
import pytest

from genozip_amd import synth, workload as W
from genozip_amd.lib import LT_UINT8, LT_UINT8_TR, LT_UINT32

pytestmark = pytest.mark.gpu


def _same_as_oracle(E, oracle, items):
    got = E.compress_many(items)
    for (c, d), g in zip(items, got):
        assert g == oracle.codec_compress(c, d), (c, len(d))
    # ... and back on the device, every one of them (the arithmetic decoder finds a symbol with the whole wave: 30 MB streams included)
    back = E.uncompress_many([(c, g, len(d)) for (c, d), g in zip(items, got)])
    assert all(b == d for b, (_, d) in zip(back, items))
    return got


def test_vcf_vblock_streams(gpu_engine, oracle):
    """config 4, one 512 MB VBlock ~ 3 000 lines x 10 000 samples: FORMAT/DP as a transposed u8 matrix (a7, 30 MB)
    and FORMAT/PL as a b250 of 3 x 10^7 entries (a5), both then through the simple codecs"""
    E = gpu_engine
    rows, cols = 3000, 10000
    h = synth.u32(77, rows * cols)
    dp = (18 + (h % np.uint32(13)) + ((h >> np.uint32(8)) % np.uint32(13))).astype(np.uint8)       # ~ 30, like Poisson (30)
    dp[(h >> np.uint32(20)) % np.uint32(97) == 0] = 0                                               # missing samples
    raw = dp.tobytes()
    lt, tr = E.local_generate(LT_UINT8, raw, cols)
    assert lt == LT_UINT8_TR and tr == np.ascontiguousarray(dp.reshape(rows, cols).T).tobytes()
    assert (lt, tr) == oracle.local_generate(LT_UINT8, raw, cols)
    assert E.local_to_native(lt, tr, cols) == (LT_UINT8, raw)
    _same_as_oracle(E, oracle, [(6, tr), (8, tr), (16, tr), (17, tr)])                              # RANB, RANb, ARTB, ARTW (9 candidate leaves, one run-length) on 30 MB each

    n = 3 * 10 ** 7
    h = synth.u32(78, n)
    ol, new = 3000, 1200                                    # words already in the dictionary / added by this VBlock
    ni = np.where(h % np.uint32(10) < np.uint32(8), h % np.uint32(100), h % np.uint32(ol + new)).astype(np.int32)
    ni[(h >> np.uint32(16)) % np.uint32(1000) == 0] = -3    # WI_EMPTY (b250.c)
    n2w = [int(x) for x in (synth.u32(79, new) % np.uint32(ol + new + 40))]
    seg = oracle.b250_seg_array(ni, ol)
    piz = E.b250_generate(seg, ol, n2w)
    assert piz == oracle.b250_generate(seg, ol, n2w)
    _same_as_oracle(E, oracle, [(6, piz), (16, piz)])


def test_bam_vblock_streams(gpu_engine, oracle):
    """config 3, one 16 MiB VBlock ~ 46 000 reads: CIGAR b250 (a handful of distinct CIGARs), binned QUAL local (6.9 MB),
    POS / TLEN as u32 locals (BGEN, a6), FLAG / MAPQ b250s"""
    E = gpu_engine
    reads = 46000
    h = synth.u32(81, reads)
    cigar = np.where(h % np.uint32(100) < np.uint32(90), 0, 1 + h % np.uint32(37)).astype(np.int32)    # 90 % "150M"
    seg = oracle.b250_seg_array(cigar, 30)                 # 30 known CIGARs, 8 new ones
    n2w = [30 + k for k in range(8)]
    piz = E.b250_generate(seg, 30, n2w)
    assert piz == oracle.b250_generate(seg, 30, n2w)
    qual = W.quality_rows(W._NP, 82, 0, reads, "bin").reshape(-1).astype(np.uint8).tobytes()
    pos = (10000 + np.cumsum(synth.u32(83, reads) % np.uint32(40))).astype("<u4").tobytes()
    lt, pos_be = E.local_generate(LT_UINT32, pos)
    assert (lt, pos_be) == oracle.local_generate(LT_UINT32, pos)
    flag = oracle.b250_generate(oracle.b250_seg_array((synth.u32(84, reads) % np.uint32(6)).astype(np.int32), 6), 6, [])
    _same_as_oracle(E, oracle, [(16, piz), (6, piz), (16, qual), (18, qual), (6, qual), (9, pos_be), (17, pos_be), (16, flag)])
