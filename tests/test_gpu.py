"""GPU (-m gpu): the parity tests proper. Everything goes through the C-ABI of libgenozip_amd.so on cuda:0 and is
compared byte for byte with the oracle / the committed golden vectors of the reference."""
import hashlib

import numpy as np
import pytest

import cases
import parity
from genozip_amd import synth
from genozip_amd.lib import SIMPLE_CODECS

pytestmark = pytest.mark.gpu


def test_native_library_is_the_one_running(gpu_engine):
    import os
    assert "gfx950" in gpu_engine.version()
    maps = open("/proc/self/maps").read()
    assert "libgenozip_amd.so" in maps and "libgenozip_amd_emul" not in maps.replace("libgenozip_amd.so", "")


def test_chain_block_boundaries(gpu_engine, oracle):
    assert parity.chain_block_boundaries(gpu_engine, oracle) > 300


def test_codec_edge_cases(gpu_engine, oracle):
    parity.codec_edge_cases(gpu_engine, oracle, max_n=300007)


def test_host_call_surface(gpu_engine, oracle):
    parity.host_call_surface(gpu_engine, oracle)
    parity.compress_lines(gpu_engine, oracle)


def test_golden_vectors_all(gpu_engine):
    """every committed reference vector incl. the 16 MiB VBlock-sized streams"""
    assert parity.golden(gpu_engine, max_n=1 << 30) > 3000


def test_assign_best(gpu_engine, oracle):
    parity.assign_best(gpu_engine, oracle, n=150000)


def test_b250(gpu_engine, oracle):
    parity.b250(gpu_engine, oracle, 180000)


def test_local_and_transpose(gpu_engine, oracle):
    parity.local(gpu_engine, oracle, 1000, 250)
    parity.local(gpu_engine, oracle, 33, 7)


def test_vblocks(gpu_engine, oracle):
    parity.vblocks(gpu_engine, oracle, 6, 300000)


def test_full_size_round_trip_properties(gpu_engine):
    """BASELINE-sized streams (a whole 16 MiB VBlock's QUAL) where the oracle would be slow: size-independent
    properties instead - decode(encode(x)) == x on the device for every codec, and the batched call is deterministic"""
    E = gpu_engine
    q = synth.quality_diverse(11, 46000).tobytes()          # ~6.9 MB, the QUAL of one 16 MiB FASTQ VBlock
    items = [(c, q) for c in SIMPLE_CODECS]
    a = E.compress_many(items)
    b = E.compress_many(items)
    assert [hashlib.sha1(x).hexdigest() for x in a] == [hashlib.sha1(x).hexdigest() for x in b]
    back = E.uncompress_many([(c, x, len(q)) for c, x in zip(SIMPLE_CODECS, a)])
    assert all(y == q for y in back)
    assert all(len(x) < len(q) // 2 for x in a)


def test_many_vblocks_concurrently(gpu_engine, oracle):
    """a batch shaped like the bench: many VBlocks x sections in one launch == one-at-a-time results"""
    E = gpu_engine
    items = []
    for v in range(24):
        items.append((7 + 10 * (v % 2), synth.quality_diverse(100 + v, 3000).tobytes()))
        items.append((6, synth.markov_bytes(200 + v, 40000, 12, 48).tobytes()))
        items.append((9, synth.u32be_increasing(300 + v, 50000).tobytes()))
    got = E.compress_many(items)
    for (c, d), g in zip(items, got):
        assert g == oracle.codec_compress(c, d)


def test_acgt_pack(gpu_engine, oracle):
    """N2: a VBlock's worth of SEQ (46 000 reads x 150 bp)"""
    parity.acgt(gpu_engine, oracle, 6900000)


def test_seg_columns(gpu_engine, oracle):
    """rows a1-a3 at a VBlock's size: 46 000 entries per column (FASTQ), every column shape of parity._column_cases"""
    parity.seg_columns(gpu_engine, oracle, 46000)


def test_transpose_partial(gpu_engine, oracle):
    """a7's partial case at odd shapes and at 1 000 x 3 000 cells"""
    parity.transpose_partial(gpu_engine, oracle, 37, 23)
    parity.transpose_partial(gpu_engine, oracle, 1000, 3000)


def test_b250_malformed(gpu_engine, oracle):
    parity.b250_malformed(gpu_engine, oracle, 300000)


def test_fastq_zip_driver(gpu_engine, oracle):
    """FASTQ text -> z_data through gz_fastq_zip_vblocks (a1-a16 + N1 in one call, paired VBlocks, two calls sharing the
    file's dictionaries) == the oracle's step-by-step composition"""
    parity.fastq_zip(gpu_engine, oracle, 6000)
    parity.fastq_zip(gpu_engine, oracle, 3000, qual=("bin", "uniform"))          # QUAL through CODEC_DOMQ (decided by the first VBlock)
    parity.fastq_zip(gpu_engine, oracle, 2000, qual=("uniform", "bin"))          # ... or not, also for later VBlocks that would fit
    parity.fastq_zip(gpu_engine, oracle, 1000, n_calls=1, qual=("uniform",), domq=13)
    parity.fastq_zip(gpu_engine, oracle, 1000, n_calls=1, qual=("bin",), domq=1)
    parity.fastq_zip(gpu_engine, oracle, 1500, small_first=True)                         # VBlocks too small to set the file's codecs (codec.c:352)
    parity.fastq_zip(gpu_engine, oracle, 1500, qual=("bin", "bin"), small_first=True)


@pytest.mark.gpu
def test_fastq_zip_monochar(gpu_engine, oracle):
    """QUAL lines of one repeated score (fastq_qual.c:33-36: FASTQ_SPECIAL_monochar_QUAL snips in QUAL's b250, the lines left out of the
    local and of CODEC_DOMQ's streams) in both mates of a pair, whose R2 VBlocks start SQBITMAP with the mate_lookup node (fastq.c:664-665);
    calls in which EVERY line is one (QUAL.local stays empty)"""
    parity.fastq_zip(gpu_engine, oracle, 3000, mono=(7, 5))
    parity.fastq_zip(gpu_engine, oracle, 3000, qual=("bin", "bin"), mono=(7, 5))
    parity.fastq_zip(gpu_engine, oracle, 1200, qual=("bin", "uniform"), mono=(3, 11), small_first=True)
    parity.fastq_zip(gpu_engine, oracle, 600, mono=(0, -1))
    parity.fastq_zip(gpu_engine, oracle, 600, mono=(-1, 4))
    # through CODEC_DOMQ with VBlocks in which every line is one (no line reaches CODEC_DOMQ: no streams, DOMQRUNS' b250 = WORD_INDEX_EMPTY)
    parity.fastq_zip(gpu_engine, oracle, 900, qual=("bin", "uniform"), small_first=True, mono=(0, -1))
    parity.fastq_zip(gpu_engine, oracle, 600, qual=("uniform", "bin"), domq=13, mono=(-1, 2))


@pytest.mark.gpu
def test_chain_checkpoint_guard(oracle, monkeypatch):
    """the range coder chain runs in double precision under a rounding mode switched by inline asm, in a generated loop that only the GPU
    executes (gz_intrin.h): what guards it is k_chain_expand's replay of every 64-symbol slice against the chain's next checkpoint. A forced
    mismatch (GZ_DEBUG_CHAIN_FAULT: slice k - 1 of every arithmetic leaf counts as missed) must fail the stream - no bytes, an error - and a
    handle made without it codes the same data to the oracle's bytes"""
    from genozip_amd.codec import Engine, GenozipAMDError
    from genozip_amd import synth
    data = synth.quality_diverse(5, 3000).tobytes()
    monkeypatch.setenv("GZ_DEBUG_CHAIN_FAULT", "3")
    E = Engine(device=0)
    try:
        for codec in (16, 17, 18, 19):
            with pytest.raises(GenozipAMDError):
                E.compress_many([(codec, data)])
        assert E.compress_many([(6, data)])[0] == oracle.codec_compress(6, data)      # (rANS has no chain: untouched)
    finally:
        E.close()
    monkeypatch.delenv("GZ_DEBUG_CHAIN_FAULT")
    E2 = Engine(device=0)
    try:
        assert E2.compress_many([(16, data)])[0] == oracle.codec_compress(16, data)
    finally:
        E2.close()


@pytest.mark.gpu
def test_e2e_files_sha256(gpu_engine):
    """whole .genozip files (FASTQ pair and single file with monochar QUAL lines, plain and through CODEC_DOMQ; SAM with a context per tag;
    multi-sample VCF) written by the HIP library are, byte for byte, the files the REFERENCE'S OWN genounzip has decoded back into the
    original text in the build container (tests/golden/e2e_sha256.json, made by tests/golden/make_e2e_golden.py with the same recipe on
    the CPU stand-in): reference-decoder evidence for HIP-made files in every GPU run"""
    import e2e_files
    assert e2e_files.check_against_golden(gpu_engine) == len(e2e_files.CASES)


@pytest.mark.gpu
def test_fastq_zip_deferred_columns(gpu_engine, oracle, monkeypatch):
    """the columns' kernels queued behind the launch of the long streams (the driver's way for few, large VBlocks), forced for files of test size"""
    monkeypatch.setenv("GZ_ZIP_DEFER", "always")
    monkeypatch.setenv("GZ_ZIP_EARLY_MIN", "0")
    parity.fastq_zip(gpu_engine, oracle, 3000, qual=("uniform", "bin"), mono=(0, 5))
    monkeypatch.delenv("GZ_ZIP_EARLY_MIN")
    parity.fastq_zip(gpu_engine, oracle, 1500, small_first=True)
    assert parity.sam_zip(gpu_engine, oracle, 3000, n_calls=1) == 2


@pytest.mark.gpu
def test_fastq_zip_prediction(gpu_engine, oracle):
    """streams coded ahead of their contexts' trial compressions with a predicted codec (the built-in prior; what the handle remembers): the
    oracle's bytes whether the prediction was right or wrong"""
    hits, misses = parity.fastq_zip_prediction(gpu_engine, oracle, 3000)
    assert hits > 0


@pytest.mark.gpu
def test_fastq_zip_early_path(gpu_engine, oracle, monkeypatch):
    """the QUAL streams coded ahead of the merge on the second handle (the driver's way for long streams, >= GZ_ZIP_EARLY_MIN scores),
    forced for streams of test size: the same bytes as when QUAL is coded with the rest"""
    monkeypatch.setenv("GZ_ZIP_EARLY_MIN", "0")
    parity.fastq_zip(gpu_engine, oracle, 6000)
    parity.fastq_zip(gpu_engine, oracle, 3000, qual=("bin", "uniform"))
    parity.fastq_zip(gpu_engine, oracle, 1500, small_first=True)
    assert parity.sam_zip(gpu_engine, oracle, 3000, n_calls=1) == 2


@pytest.mark.gpu
def test_fastq_zip_speculation(gpu_engine, oracle):
    a, c, d = parity.fastq_zip_speculation(gpu_engine, oracle, 2500)
    assert a["qual_lcodec"] and d["qual_mode"] == 13


@pytest.mark.gpu
def test_assign_sort_and_host_codecs(gpu_engine, oracle):
    """a8 in full: the reference's sorter over all twelve candidates, the host's BZ2 / LZMA rows inside the driver's trials, sections
    coded by the host's coder framed with the rest"""
    parity.assign_sort(gpu_engine, oracle)
    used = parity.fastq_zip_host_codecs(gpu_engine, oracle, 3000)
    assert 3 in used or 4 in used
    parity.fastq_zip(gpu_engine, oracle, 2500, host=parity.host_codecs_for_tests(clock_bz2=100.0, clock_lzma=20000.0))
    data = bytes(range(256)) * 40 + b"ACGT" * 20000
    c, table = gpu_engine.assign_best_ex(data, extra=[(3, 600.0, 200.0), (4, 500.0, 40000.0)])
    oc = oracle.assign_best_with(data, [(3, 600.0 - 28, 200.0), (4, 500.0 - 28, 40000.0)])
    assert c == oc and table[0][0] == c and len(table) == 11


def test_fastq_zip_two_in_flight(gpu_engine, oracle):
    parity.fastq_zip_two_in_flight(gpu_engine, oracle, 4000)


def test_fastq_zip_errors(gpu_engine, oracle):
    parity.fastq_zip_errors(gpu_engine, oracle)


def test_c_host_program():
    """tests/c/zip_fastq.c: plain C11 against include/genozip_amd.h and the real library, no Python in the loop"""
    import os
    import subprocess
    import tempfile
    import test_abi
    with tempfile.TemporaryDirectory() as d:
        exe = test_abi._build_c_program(os.path.join(test_abi.ROOT, "genozip_amd"), "libgenozip_amd.so", os.path.join(d, "zip_fastq"))
        r = subprocess.run([exe, "46000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_ctx_golden_vectors(gpu_engine, oracle):
    """b250 seg append / generation, dyn-int columns and the matrix transpose against vectors made from the reference's own
    src/b250.c and src/dyn_int.c (tests/golden/ctx_golden.json)"""
    parity.ctx_golden(gpu_engine, oracle)


def test_merge_chain(gpu_engine, oracle):
    """seg columns -> host dictionary merge (a4) -> b250 generation over VBlocks that share dictionaries"""
    parity.merge_chain(gpu_engine, oracle, 46000)


def test_decode_malformed(gpu_engine, oracle):
    parity.decode_malformed(gpu_engine, oracle)


def test_decode_foreign_arith(gpu_engine, ref):
    parity.decode_foreign_arith(gpu_engine, ref, n=400000)


def test_b250_pair_identical(gpu_engine, oracle):
    parity.b250_pair_identical(gpu_engine, oracle, 300000)


def test_vcf_and_sam_front(gpu_engine, oracle):
    parity.vcf_front(gpu_engine, oracle, 300, 400)
    parity.sam_front(gpu_engine, oracle, 20000)


def test_bam_front(gpu_engine, oracle):
    """N1 for BAM (bam_seg_txt_line, src/bam_seg.c:425-520): the records of an uncompressed BAM stream found chunk-parallel and turned
    into alignment lines == the oracle's serial walk and conversion == the SAM text they were encoded from (30 000 records, ~12 MB:
    ~190 chunks of the record chain), incl. records longer than a chunk, every optional-field type, malformed streams"""
    assert parity.bam_front(gpu_engine, oracle, 30000) == 30000


def test_fastq_front(gpu_engine, oracle):
    """N1 (first part) chained into a1-a3 on a VBlock's worth of FASTQ text (23 000 reads, ~7 MB)"""
    parity.fastq_front(gpu_engine, oracle, 23000)


def test_seg_random(gpu_engine, oracle):
    parity.seg_random(gpu_engine, oracle, 60, seed=77)


def test_seg_column_vcf_sized(gpu_engine, oracle):
    """a FORMAT/PL-like column: 3 million snips, ~2 000 distinct, half of them already in the cloned dictionary"""
    import numpy as np
    from genozip_amd import synth
    n = 3000000
    words = [b"%d,%d,%d" % (i % 97, (i * 7) % 255, (i * 13) % 255) for i in range(2000)]
    text = b"".join(words)
    lens = np.array([len(w) for w in words], dtype=np.uint32)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint32)
    r = synth.u32(99, n).astype(np.int64)
    pick = np.minimum(r % 2000, (r >> 11) % 2000)                    # skewed towards the low indices
    got = gpu_engine.ctx_seg_column(text, starts[pick], lens[pick], words[::2])
    want = oracle.ctx_seg_column(text, starts[pick], lens[pick], words[::2])
    for key in ("node_index", "node_char_index", "node_snip_len", "counts"):
        assert np.array_equal(got[key], want[key]), key
    for key in ("dict", "b250", "b250_count", "all_the_same"):
        assert got[key] == want[key], key


def test_two_handles_in_flight(gpu_engine, oracle):
    """two GzHandles (two host threads in the reference's threading model) with batches in flight at the same time: the
    persistent chain kernels of both must get along (process-wide budget in gz_host.cpp)"""
    from genozip_amd.codec import Engine
    E1, E2 = gpu_engine, Engine(device=0)
    d1 = [synth.quality_diverse(900 + k, 2000).tobytes() for k in range(6)]      # 300 KB each: several position chunks
    d2 = [synth.markov_bytes(950 + k, 250000, 30, 40).tobytes() for k in range(6)]
    b1 = [E1.mem.upload(d) for d in d1]; b2 = [E2.mem.upload(d) for d in d2]
    t1, o1 = E1.make_stream_table([(16, b, len(d)) for b, d in zip(b1, d1)])
    t2, o2 = E2.make_stream_table([(17, b, len(d)) for b, d in zip(b2, d2)])
    for _ in range(3):
        E1.compress_table(t1, len(d1)); E2.compress_table(t2, len(d2))
        E1.sync(); E2.sync()
    for k, d in enumerate(d1):
        assert E1.mem.download(o1[k], t1[k].out_len) == oracle.codec_compress(16, d)
    for k, d in enumerate(d2):
        assert E2.mem.download(o2[k], t2[k].out_len) == oracle.codec_compress(17, d)


def test_unpipelined_mode(oracle):
    """GZ_NO_PIPELINE=1 (what the PMC profiling passes use): the same kernels one after the other, same bytes"""
    import os
    from genozip_amd.codec import Engine
    os.environ["GZ_NO_PIPELINE"] = "1"
    try:
        E = Engine(device=0)
    finally:
        del os.environ["GZ_NO_PIPELINE"]
    items = [(16, synth.quality_diverse(77, 2500).tobytes()), (17, synth.u32be_increasing(78, 120000).tobytes()),
             (18, synth.markov_bytes(79, 300000, 4, 33).tobytes()), (19, synth.uniform_bytes(80, 200000, 200).tobytes())]
    got = E.compress_many(items)
    for (c, d), g in zip(items, got):
        assert g == oracle.codec_compress(c, d), (c, len(d))


def test_chunk_and_tile_boundaries(gpu_engine, oracle):
    """sizes around the 64 K chunking threshold, whole / broken 4096-position sort tiles, 16 chunks + 1 byte; data that
    exercises the run-length events (long runs, runs of exactly 3k), wide alphabets and tiny ones - all arith codecs"""
    E = gpu_engine
    items = []
    seed = 31000
    for n in (65535, 65536, 65537, 69632, 69633, 131071, 262145, 1048577):
        for kind, nsym in (("runs", 8), ("uniform", 256), ("markov", 40), ("skew", 3)):
            seed += 1
            d = synth.stream(kind, seed, n, nsym).tobytes()
            for c in ((16, 17) if n > 300000 else (16, 17, 18, 19)):
                items.append((c, d))
    three = (bytes([7]) * 4 + bytes([9]) * 7 + bytes([11]) * 10 + bytes([13])) * 9000      # runs of r = 3, 6, 9, 0
    items += [(17, three), (19, three), (17, bytes(200000)), (19, bytes([5]) * 70001)]
    got = E.compress_many(items)
    for (c, d), g in zip(items, got):
        assert g == oracle.codec_compress(c, d), (c, len(d))


def test_sam_zip_driver(gpu_engine, oracle):
    """N1 for SAM (BASELINE configs[2] from text): alignment lines through the driver's one-line-record plan == the oracle's composition,
    4 VBlocks over 2 calls, with and without optional fields, binned (CODEC_DOMQ) and 40-level qualities"""
    assert parity.sam_zip(gpu_engine, oracle, 3000) == 4
    assert parity.sam_zip(gpu_engine, oracle, 3000, tags=True) == 4                                                        # (a context per optional field behind the AUX container)
    assert parity.sam_zip(gpu_engine, oracle, 1500, n_calls=1, qual="uniform", aux=False, via_bam=True) == 2        # (from BAM records: N1 for BAM in front)


def test_vcf_zip_driver(gpu_engine, oracle):
    """N1 for VCF (BASELINE configs[3] from text): data lines through the driver's per-sample plan == the oracle's composition, 4 VBlocks
    over 2 calls: fixed fields, GT / PL b250 columns of lines x samples entries, DP transposed (LT_UINT8_TR)"""
    assert parity.vcf_zip(gpu_engine, oracle, 60, 300) == 4


def test_header_layouts(gpu_engine):
    """a9 / a16 / N4: SectionHeaderCtx, SectionHeaderVbHeader, SectionHeaderTxtHeader and the plan's containers as the product writes them
    == the reference's own structs filled through their members (tests/golden/hdr_golden.json from oracle/ref_hdr_shim.c)"""
    assert parity.header_kats(gpu_engine) >= 6


def test_domq(gpu_engine, oracle):
    """N3: CODEC_DOMQ's pre-transform at VBlock size (46 000 lines) == the oracle, whose restatement is pinned to the reference's
    own codec_domq.c by tests/golden/ctx_golden.json (checked for the product in test_ctx_golden_vectors)"""
    parity.domq(gpu_engine, oracle, 46000)


def _bench_workload(gpu_engine, **kw):
    import argparse
    import torch
    import bench
    a = argparse.Namespace(pairs=1000000, vb_mb=16, qual="div", scaling="weak", stream_reads=0, batch_pairs=64, pin_codecs=False, two_in_flight=False)
    for k, v in kw.items():
        setattr(a, k, v)
    return bench, bench.Workload(gpu_engine, a, 0, 1, torch.device("cuda", 0))


def _check_vblocks(gpu_engine, oracle, bench, wl, z_all, decode_vbs):
    """size-independent properties of finished VBlocks: the VB header names its own length, every section's adler32 holds (checked
    by gz_vb_uncompress on the device), sections come in ascending (DEP level, did_i) order with the b250s last, and the QUAL
    section decodes to the text's quality lines"""
    import numpy as np
    RB, L = wl.W.RECORD_BYTES, wl.W.READ_LEN
    qual_id = next(c["dict_id"] for c in wl.plan["ctxs"] if c["tag"] == "QUAL")
    text = wl.text[:wl.text_len]
    for v, ((off, ln, vi, r1), z) in enumerate(zip(wl.vb, z_all)):
        assert int.from_bytes(z[40:44], "big") == len(z) and int.from_bytes(z[20:24], "big") == vi
        secs = bench.walk_sections(z)
        kinds = [s[0] for s in secs]
        assert kinds == sorted(kinds, reverse=True), "locals (12) before b250s (11)"
        if v in decode_vbs:
            total = sum(s[3] for s in secs)
            dec = gpu_engine.vb_uncompress(z, total)
            assert len(dec) == len(secs)
            q = next(d for d, s in zip(dec, secs) if s[2] == qual_id and s[0] == 12)
            want = text[off:off + ln].cpu().numpy().reshape(-1, RB)[:, RB - L - 1:RB - 1]
            assert q == want.tobytes(), "QUAL of VBlock %d" % vi


def test_full_size_fastq_file(gpu_engine, oracle):
    """BASELINE configs[1] at full size (2 x 1 M reads, 16 MiB VBlocks) through the driver, as bench.py runs it: properties that do not
    need the oracle at this size - framing, order, adler32 and QUAL round trip of the first / a middle / the last VBlock, the same
    bytes when the file is compressed again, every read accounted for - plus the oracle's own coder on one QUAL section"""
    bench, wl = _bench_workload(gpu_engine)
    total = wl.step(None)
    z1 = wl.zbuf[:total].cpu().numpy().tobytes()
    offs = list(wl.offs)
    z_all = [z1[offs[i]:offs[i + 1]] for i in range(len(wl.vb))]
    assert sum(t.n_reads for t in wl.tab) == 2 * 1000000
    _check_vblocks(gpu_engine, oracle, bench, wl, z_all, {0, len(z_all) // 2, len(z_all) - 1})
    # every one of the 2 M read names, rebuilt from the six QNAME and three QNAME2 item contexts' sections + the file's dictionaries
    # (b250 word indices -> snips; lookups and self-deltas through the locals), == line 1 of the text
    got = [dict(z=z, n_reads=int(t.n_reads)) for z, t in zip(z_all, wl.tab)]
    text_host = wl.text[:wl.text_len].cpu().numpy()
    dec = lambda codec, pay, ulen: bytes(pay) if codec == 1 else oracle.codec_uncompress(codec, pay, ulen)   # noqa: E731
    assert parity.check_qnames(wl.F, wl.plan, text_host, wl.vb, got, dec) == 2 * 1000000
    total2 = wl.step(None)                                   # a new file with the same text: the same bytes
    assert total2 == total and wl.zbuf[:total].cpu().numpy().tobytes() == z1
    # one QUAL payload against the CPU restatement of the coder (6.9 MB: a few seconds)
    st, codec, did, ulen, pay, _ = next(s for s in bench.walk_sections(z_all[3]) if s[3] > 1000000)
    off, ln = wl.vb[3][0], wl.vb[3][1]
    RB, L = wl.W.RECORD_BYTES, wl.W.READ_LEN
    raw = wl.text[off:off + ln].cpu().numpy().reshape(-1, RB)[:, RB - L - 1:RB - 1].tobytes()
    assert oracle.codec_compress(codec, raw) == bytes(pay)


def test_streamed_file(gpu_engine, oracle):
    """BASELINE configs[4] in small: ONE file streamed through the driver in several calls (dictionaries and codecs carried from call to
    call, vblock_i continuing): every call's VBlocks hold the properties above; the codecs chosen in the first call are the ones
    every later call uses; the dictionaries only grow"""
    bench, wl = _bench_workload(gpu_engine, pairs=0, stream_reads=400000, batch_pairs=3, vb_mb=4)
    from genozip_amd.shard import zip_vblocks_sharded
    F, n = wl.F, len(wl.vb)
    F.reset()
    codecs_first, words = None, 0
    for call in range(3):
        if call:
            for t in wl.tab:
                t.vblock_i += len(wl.ranges)                # (R1 c*B+1.., R2 N+c*B+1..: the numbering bench.py streams with)
        zip_vblocks_sharded(F, None, wl.text, wl.text_len, wl.tab, n)
        z_all = [r["z"] for r in F.results(wl.tab)]
        vb_now = [(o, l, int(t.vblock_i), r1) for (o, l, _, r1), t in zip(wl.vb, wl.tab)]
        wl_view = type("V", (), dict(W=wl.W, plan=wl.plan, text=wl.text, text_len=wl.text_len, vb=vb_now))
        _check_vblocks(gpu_engine, oracle, bench, wl_view, z_all, {0, n - 1})
        codecs = {(s[0], s[2]): s[1] for z in z_all for s in bench.walk_sections(z) if s[3] >= 50}
        if codecs_first is None:
            codecs_first = codecs
        else:
            assert all(codecs_first.get(k, c) == c for k, c in codecs.items()), "a committed codec changed between calls"
        w = len(F.zctx_words(3))
        assert w >= words
        words = w
    for t, v in zip(wl.tab, wl.vb):
        t.vblock_i = v[2]


def test_streamed_wgs_share(gpu_engine, oracle):
    """BASELINE configs[4] at ONE GPU's full share: 600 M reads over 8 GPUs = 75 M read pairs = 55 GB of FASTQ text per GPU, streamed
    through ONE file object in calls of 112 VBlock pairs (3.76 GB of NEW text per call - the generator continues the read numbering,
    nothing is re-used), vblock_i as the reader wants them (R1 1..N, R2 N+1..2N). Per call: every VBlock's header names its own length
    and number, sections in (DEP level, did_i) order (16 of the 224 VBlocks read back, the others counted), every section's adler32
    and - in the first and the last call - the QUAL round trip of the first and last VBlock;
    over the calls: the codecs of the first call hold, the dictionaries only grow, every read is accounted for.
    GZ_TEST_WGS_CALLS: fewer calls (default: the whole share)"""
    import os
    import torch
    bench, wl = _bench_workload(gpu_engine, pairs=0, stream_reads=75000000, batch_pairs=112, vb_mb=0)
    from genozip_amd.shard import zip_vblocks_sharded
    W, F, n = wl.W, wl.F, len(wl.vb)
    n_calls = int(os.environ.get("GZ_TEST_WGS_CALLS", wl.calls_per_step))
    assert wl.calls_per_step >= 15 and n == 224
    reads_per_call = sum(r[1] for r in wl.ranges)
    th = W._TH(wl.text.device)
    F.reset()
    codecs_first, words, reads, text_bytes, z_bytes = None, 0, 0, 0, 0
    for call in range(n_calls):
        if call:                                            # the next stretch of the two files: new reads, later VBlocks
            at = 0
            for mate in (1, 2):
                for r0, m_reads in wl.ranges:
                    for c0 in range(0, m_reads, 1000000):           # (whole VBlocks at a time)
                        m = min(1000000, m_reads - c0)
                        wl.text[at + c0 * W.RECORD_BYTES: at + (c0 + m) * W.RECORD_BYTES] = W.fastq_text(1, call * reads_per_call + r0 + c0, m, mate=mate, profile="div", xp=th)
                    at += m_reads * W.RECORD_BYTES
            torch.cuda.synchronize()
            for t in wl.tab:
                t.vblock_i += len(wl.ranges)
        zip_vblocks_sharded(F, None, wl.text, wl.text_len, wl.tab, n)
        # 16 of the call's 224 VBlocks are read back and taken apart (all of them: 1.4 GB per call through the host - the test's time),
        # the others are counted
        pick = sorted({0, 1, n // 2 - 1, n // 2, n - 2, n - 1} | {(7 * call + 13 * k) % n for k in range(10)})
        vb_now = [(o, l, int(t.vblock_i), r1) for (o, l, _, r1), t in zip(wl.vb, wl.tab)]
        assert [v[2] for v in vb_now[:2]] == [call * 112 + 1, call * 112 + 2] and vb_now[112][2] == wl.n_pairs_file + call * 112 + 1
        assert all(t.status == 1 and t.z_len > 84 for t in wl.tab)          # (GZ_OK)
        z_all = [F._download(wl.tab[v].z_data, wl.tab[v].z_len) for v in pick]
        res = [dict(n_reads=int(t.n_reads)) for t in wl.tab]
        wl_view = type("V", (), dict(W=W, plan=wl.plan, text=wl.text, text_len=wl.text_len, vb=[vb_now[v] for v in pick]))
        # (the device decodes a 6.8 M-symbol QUAL section in ~5 s - one wave, section 3 of DESIGN.md: the round trip in the first and the last call)
        _check_vblocks(gpu_engine, oracle, bench, wl_view, z_all, {0, len(pick) - 1} if call in (0, n_calls - 1) else set())
        codecs = {(s[0], s[2]): s[1] for z in z_all for s in bench.walk_sections(z) if s[3] >= 50}
        if codecs_first is None:
            codecs_first = codecs
        else:
            assert all(codecs_first.get(k, c) == c for k, c in codecs.items()), "a committed codec changed between calls"
        w = len(F.zctx_words(3))
        assert w >= words
        words = w
        reads += sum(r["n_reads"] for r in res); text_bytes += wl.text_len; z_bytes += sum(int(t.z_len) for t in wl.tab)
    assert reads == 2 * reads_per_call * n_calls and text_bytes == reads * W.RECORD_BYTES
    if n_calls == wl.calls_per_step:
        assert reads >= 2 * 75000000 and text_bytes > 54e9
    for t, v in zip(wl.tab, wl.vb):
        t.vblock_i = v[2]


def test_rans_tables(gpu_engine, oracle):
    parity.rans_tables(gpu_engine, oracle)


def test_vcf_retest(gpu_engine, oracle):
    parity.vcf_retest(gpu_engine, oracle, 40, 100)


def test_wide_models(gpu_engine, oracle):
    assert parity.wide_models(gpu_engine, oracle, big=True) > 300
    assert parity.wide_models_random(gpu_engine, oracle, 400) == 400


def test_record_reciprocals(gpu_engine):
    """the reciprocal the model kernel computes for a symbol's record (d_record_inv: reciprocal seed + one Newton step, 16 zero low bits),
    for EVERY total a model can have, lies in the interval for which tests/test_magic.py shows range / tot to be exact - with the 16 low
    bits a record's cum can set on top"""
    from fractions import Fraction
    import struct
    n = 65536 + 32
    inv = gpu_engine.debug_record_inv(1, n - 1)
    bits = inv.view(np.uint64)
    assert not (bits & np.uint64(0xffff)).any()
    hi = (bits | np.uint64(0xffff)).view(np.float64)
    worst = 0.0
    for d in range(1, n):
        exact = Fraction(1, d << 45)
        lo_f, hi_f = Fraction(float(inv[d - 1])), Fraction(float(hi[d - 1]))
        assert exact <= lo_f and hi_f <= exact * (1 + Fraction(1, 1 << 33)), d
        worst = max(worst, float(hi_f / exact - 1))
    assert worst < 2.0 ** -33


def test_assign_golden_vectors(gpu_engine):
    """row a8 PINNED on the device: gz_codec_assign_best_ex - the nine trial compressions on the GPU, the host's rows, the sorter - picks the
    codec and builds the table the reference's own codec_assign_best_codec did on the same samples with the same clock
    (tests/golden/assign_golden.json, oracle/ref_assign_shim.c)"""
    def best_table(data, rows, ns, mode):
        return gpu_engine.assign_best_ex(data, [(c, sz + 28, ck) for c, sz, ck in rows], [float(x) for x in ns], mode)
    assert parity.assign_golden_run(best_table) > 100


def test_chain_timeout_falls_back(oracle, monkeypatch):
    """the persistent range-coder chain follows the model kernels chunk by chunk; when it never hears from them (GZ_DEBUG_STARVE_CHAIN: the
    progress announcements are left out, as under a tool that serialises kernels) it gives up after a bounded wait - and gz_sync must then run
    the batch again unpipelined and hand out the reference's bytes (COMPRESS: false only for "too small", src/compressor.c:89-110), with a
    warning and a count; the VBlock compute driver, whose long QUAL streams are coded ahead on a second handle, must write its sections again"""
    from genozip_amd.codec import Engine
    monkeypatch.setenv("GZ_DEBUG_STARVE_CHAIN", "1")
    E = Engine(device=0)
    monkeypatch.delenv("GZ_DEBUG_STARVE_CHAIN")
    qual = synth.quality_diverse(7, 2000).tobytes()                       # 300 KB: several position chunks, the pipelined path
    for codec in (16, 17):
        got = E.compress_many([(codec, qual)])[0]
        assert got == oracle.codec_compress(codec, qual)
    assert E.L.gz_chain_fallbacks(E.h) == 2 and b"warning" in E.L.gz_last_warning(E.h) and b"warning" not in E.L.gz_last_error(E.h)
    assert E.uncompress_many([(16, E.compress_many([(16, qual)])[0], len(qual))])[0] == qual
    parity.fastq_zip(E, oracle, 2500, n_calls=1)                          # whole path, QUAL coded ahead on the background handle
    E.close()


def test_tiled_models_exact(oracle, monkeypatch):
    """GZ_MODEL_TILED=1 (an experiment of round 5 that is exact and moves a third less through HBM, but slower - DESIGN section 3 - and therefore
    off): order-1 leaves of at most 64 distinct bytes go through k_arith_model_tiled - one workgroup per leaf, the sort inside LDS tiles, the
    records out coalesced - instead of k_ctx_* + k_arith_model. Same bytes: single streams in one piece and in position chunks behind the
    persistent chain, every arithmetic codec, and a whole FASTQ call"""
    from genozip_amd.codec import Engine
    monkeypatch.setenv("GZ_MODEL_TILED", "1")
    E = Engine(device=0)
    monkeypatch.delenv("GZ_MODEL_TILED")
    items = [(16, synth.quality_diverse(3, 30000).tobytes()), (16, synth.markov_bytes(3, 70000, 40, 33).tobytes()), (16, synth.markov_bytes(5, 100, 10, 60).tobytes()),
             (17, synth.quality_diverse(4, 4000).tobytes()), (18, synth.quality_binned(5, 6000).tobytes()), (19, synth.markov_bytes(6, 300000, 64, 0).tobytes()),
             (16, bytes(70000)), (16, synth.uniform_bytes(8, 100000, 200).tobytes())]        # (the last one: a wide alphabet - the other kernels' leaf)
    got = E.compress_many(items)
    for (c, d), g in zip(items, got):
        assert g == oracle.codec_compress(c, d), (c, len(d))
    parity.fastq_zip(E, oracle, 2500, n_calls=2)
    E.close()


def test_rccl_sees_the_sharding_code():
    """the N-GPU path on the ONE GPU of the test box, over RCCL: (1) genozip_amd/shard.py's exchanges on HBM tensors in an nccl group of one
    rank (all_gather of byte strings, the gather to the writer rank with a loop-back ncclSend / ncclRecv pair); (2) `bench.py --gpus 1
    --scaling strong` launched the way the driver launches N ranks (torch.distributed.run), with GZ_BENCH_FORCE_DIST=1: the file's merge
    blobs and codec votes go through all_gather, the z_data through the gather, and the line carries an `rccl` record with backend nccl"""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def port():
        s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
        return p
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "rccl_world1.py"), str(port())], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0 and "OK" in p.stdout.decode().split(), p.stderr.decode()[-3000:]      # (RCCL prints its version banner behind it)
    env = dict(os.environ, GZ_BENCH_FORCE_DIST="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port()),
                        os.path.join(root, "bench.py"), "--gpus", "1", "--scaling", "strong", "--pairs", "60000", "--steps", "2", "--warmup", "1", "--no-cpu", "--warm-steps", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    d = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["rccl"]["world"] == 1 and d["rccl"]["backend"] == "nccl", d.get("rccl")
    assert d["rccl"]["exchanges_per_step"] == 2 and d["rccl"]["exchange_bytes_per_step"] == 0 and d["rccl"]["bytes_gathered_per_step"] > 0, d["rccl"]


def test_sam_and_vcf_drivers_at_size(gpu_engine, oracle):
    """the SAM and VCF plans through the driver at sizes where the long-stream machinery works (position chunks, the persistent chain, streams
    coded ahead on the background handle, multi-workgroup b250 generation): 2 x 60 000 alignment lines a call (9 MB of QUAL a VBlock), with
    optional fields; 2 x 400 data lines x 2 000 samples (0.8 M entries a per-sample column) - each VBlock's z_data == the oracle's
    composition. (The full-size configurations are compared with the same composition in every bench line: `file_exact`.)"""
    assert parity.sam_zip(gpu_engine, oracle, 120000, n_calls=1, tags=True) == 2
    assert parity.sam_zip(gpu_engine, oracle, 120000, n_calls=1, qual="uniform", aux=False) == 2
    assert parity.vcf_zip(gpu_engine, oracle, 400, 2000, n_calls=1) == 2


def test_profile_modes(gpu_engine, oracle):
    """gz_profile: 1 = every kernel launch between two HIP events, 2 = only the two kernels a step can be as long as (k_arith_chain,
    k_arith_model) - bench.py times its steps in mode 2 (events on ~700 launches a step cost the host as much as the launches) and takes the
    per-kernel table from one untimed step in mode 1. The coded bytes do not depend on the mode."""
    E = gpu_engine
    data = synth.quality_diverse(1, 3000).tobytes()
    want = oracle.codec_compress(16, data)
    E.profile(True, reset=True)
    assert E.compress_many([(16, data), (6, data)])[0] == want
    E.profile(False)
    full = E.profile_results()
    assert "k_arith_chain" in full and any(k.startswith("k_arith_model") for k in full) and len(full) > 6, sorted(full)
    E.profile(2, reset=True)
    assert E.compress_many([(16, data), (6, data)])[0] == want
    E.profile(False)
    heavy = E.profile_results()
    assert heavy and all(k == "k_arith_chain" or k.startswith("k_arith_model") for k in heavy), sorted(heavy)
    assert heavy["k_arith_chain"][1] == full["k_arith_chain"][1]                      # the same launches, the same count
    E.profile(False, reset=True)
