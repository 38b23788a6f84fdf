"""CPU: the UNMODIFIED product sources (host orchestrator + kernels), compiled against the CPU stand-in of the HIP
runtime in tests/emul, checked against the oracle on small inputs. Catches logic errors without a GPU; the same
checks run at full size on the GPU in tests/test_gpu.py."""
import os

import pytest

import parity


def test_emul_codec_edge_cases(emul_engine, oracle):
    parity.codec_edge_cases(emul_engine, oracle, max_n=1031, thin_from=24)


def test_emul_host_call_surface(emul_engine, oracle):
    parity.host_call_surface(emul_engine, oracle)
    parity.compress_lines(emul_engine, oracle)


def test_emul_golden_small(emul_engine):
    assert parity.golden(emul_engine, max_n=260, stride=3) > 800


def test_emul_assign_best(emul_engine, oracle):
    parity.assign_best(emul_engine, oracle, n=6000)


def test_emul_b250(emul_engine, oracle):
    parity.b250(emul_engine, oracle, 4000)


def test_emul_local(emul_engine, oracle):
    parity.local(emul_engine, oracle, 37, 50)


def test_emul_vblocks(emul_engine, oracle):
    parity.vblocks(emul_engine, oracle, 3, 6000)


def test_emul_arith_long_streams(emul_engine, oracle):
    """long enough for several 16384-byte normalisation tiles and thousands of 64-symbol replay slices"""
    from genozip_amd import synth
    items = [(16, synth.markov_bytes(3, 70000, 40, 33).tobytes()), (16, synth.uniform_bytes(4, 50000, 7).tobytes()),
             (18, synth.skewed_bytes(5, 90000, 4, 0.3).tobytes()), (17, synth.u32be_increasing(6, 80000).tobytes()),
             (16, bytes(60000)), (16, synth.skewed_bytes(7, 120000, 2, 0.02).tobytes()),
             # position chunks (> 64 K) through the wide-alphabet models and the all-zero special case
             (16, synth.uniform_bytes(8, 100000, 200).tobytes()), (16, synth.markov_bytes(9, 70000, 100, 20).tobytes()), (16, bytes(70000)),
             # more than 131 072 output bytes: two ranges of k_low_norm, what leaves the second one goes through k_low_carry
             (16, synth.uniform_bytes(12, 300000, 40).tobytes())]
    # wide alphabet (200 symbols), but every context byte is followed by only 100 / 40 of them: over position chunks such a context
    # runs with its own alphabet (k_ctx_succ: two register planes / one instead of four)
    import numpy as np
    for seed, width in ((10, 100), (11, 40)):
        r = np.frombuffer(synth.uniform_bytes(seed, 110000, width).tobytes(), dtype=np.uint8).astype(np.int64)
        items.append((16, (np.cumsum(r) % 200).astype(np.uint8).tobytes()))
    got = emul_engine.compress_many(items)
    for (c, d), g in zip(items, got):
        assert g == oracle.codec_compress(c, d), (c, len(d))


def test_emul_acgt(emul_engine, oracle):
    parity.acgt(emul_engine, oracle, 5000)


def test_emul_seg_columns(emul_engine, oracle):
    parity.seg_columns(emul_engine, oracle, 3000)


def test_emul_vcf_and_sam_front(emul_engine, oracle):
    parity.vcf_front(emul_engine, oracle, 40, 25)
    parity.sam_front(emul_engine, oracle, 200)


def test_emul_bam_front(emul_engine, oracle):
    """N1 for BAM: the record chain and the records' alignment lines (the kernels on the CPU stand-in) == the serial restatement"""
    assert parity.bam_front(emul_engine, oracle, 600) == 600


def test_emul_fastq_front(emul_engine, oracle):
    parity.fastq_front(emul_engine, oracle, 700)


def test_emul_b250_long(emul_engine, oracle):
    """b250s beyond GZ_B250_BIG take the multi-workgroup path (super chunks of 256 x 64 bytes)"""
    parity.b250(emul_engine, oracle, 90000)


def test_emul_b250_malformed(emul_engine, oracle):
    parity.b250_malformed(emul_engine, oracle, 60000)


def test_emul_arith_many_long_leaves(emul_engine, oracle):
    """nine leaves that all span position chunks (three chain workgroups, the last one partly filled)"""
    from genozip_amd import synth
    items = [(16, synth.markov_bytes(40 + i, 66000 + 997 * i, 30 + i, 33).tobytes()) for i in range(9)]
    got = emul_engine.compress_many(items)
    for (c, d), g in zip(items, got):
        assert g == oracle.codec_compress(c, d), (c, len(d))


def test_emul_transpose_partial(emul_engine, oracle):
    parity.transpose_partial(emul_engine, oracle, 37, 23)
    parity.transpose_partial(emul_engine, oracle, 1, 300)
    parity.transpose_partial(emul_engine, oracle, 300, 1)


def test_emul_b250_pair_identical(emul_engine, oracle):
    parity.b250_pair_identical(emul_engine, oracle, 60000)


def test_emul_seg_random(emul_engine, oracle):
    parity.seg_random(emul_engine, oracle, 25)


def test_emul_decode_malformed(emul_engine, oracle):
    parity.decode_malformed(emul_engine, oracle)


def test_emul_decode_foreign_arith(emul_engine, ref):
    parity.decode_foreign_arith(emul_engine, ref, sizes=(4, 75, 130, 256), n=12000)


def test_emul_merge_chain(emul_engine, oracle):
    parity.merge_chain(emul_engine, oracle, 900)


def test_emul_fastq_zip(emul_engine, oracle):
    parity.fastq_zip(emul_engine, oracle, 100)
    parity.fastq_zip(emul_engine, oracle, 54, small_first=True)                         # VBlocks too small to set the file's codecs


def test_emul_fastq_zip_monochar(emul_engine, oracle):
    """QUAL lines of one repeated score (fastq_qual.c:33-36: FASTQ_SPECIAL_monochar_QUAL snips in QUAL's b250, the lines left out of the
    local and of CODEC_DOMQ's streams) in both mates of a pair, whose R2 VBlocks start SQBITMAP with the mate_lookup node (fastq.c:664-665);
    a call in which EVERY line is one (QUAL.local stays empty: its b250 is all-the-same with a special snip)"""
    parity.fastq_zip(emul_engine, oracle, 60, mono=(7, 5))
    parity.fastq_zip(emul_engine, oracle, 60, qual=("bin", "bin"), mono=(7, 5))
    parity.fastq_zip(emul_engine, oracle, 40, mono=(0, -1))
    # a file that goes through CODEC_DOMQ with VBlocks in which EVERY line is one: CODEC_DOMQ gets no line there (no streams, DOMQRUNS' b250 the one
    # entry WORD_INDEX_EMPTY) - found by tests/fuzz_emul.py driver in round 5
    parity.fastq_zip(emul_engine, oracle, 33, qual=("bin", "uniform"), small_first=True, mono=(0, -1))
    parity.fastq_zip(emul_engine, oracle, 20, qual=("uniform", "bin"), domq=13, mono=(-1, 2))


def test_emul_fastq_zip_early_path(emul_engine, oracle, monkeypatch):
    """the QUAL streams coded ahead of the merge (what the driver does for LONG streams, >= GZ_ZIP_EARLY_MIN scores: their trial is waited for in
    the seg phase, the streams start on the second handle) - forced here for streams of test size: the same bytes as the ordinary way"""
    monkeypatch.setenv("GZ_ZIP_EARLY_MIN", "0")
    parity.fastq_zip(emul_engine, oracle, 100)
    parity.fastq_zip(emul_engine, oracle, 72, qual=("bin", "uniform"))
    parity.fastq_zip(emul_engine, oracle, 54, small_first=True)
    assert parity.sam_zip(emul_engine, oracle, 300, n_calls=1) == 2


def test_emul_fastq_zip_deferred_columns(emul_engine, oracle, monkeypatch):
    """the columns' kernels queued behind the launch of the long streams (what the driver does for few, large VBlocks - the persistent chain's
    workgroups then find free compute units at once), forced for files of test size, with the long streams coded ahead: the same bytes"""
    monkeypatch.setenv("GZ_ZIP_DEFER", "always")
    monkeypatch.setenv("GZ_ZIP_EARLY_MIN", "0")
    parity.fastq_zip(emul_engine, oracle, 72, qual=("uniform", "bin"), mono=(0, 5))
    monkeypatch.delenv("GZ_ZIP_EARLY_MIN")
    parity.fastq_zip(emul_engine, oracle, 54, small_first=True)
    assert parity.sam_zip(emul_engine, oracle, 200, n_calls=1) == 2


def test_emul_fastq_zip_speculation(emul_engine, oracle):
    parity.fastq_zip_speculation(emul_engine, oracle, 42)


def test_emul_fastq_zip_prediction(emul_engine, oracle):
    parity.fastq_zip_prediction(emul_engine, oracle, 60)


def test_emul_chain_block_boundaries(emul_engine, oracle):
    parity.chain_block_boundaries(emul_engine, oracle, big=False)


def test_emul_wide_models(emul_engine, oracle):
    assert parity.wide_models(emul_engine, oracle, big=False) > 60
    assert parity.wide_models_random(emul_engine, oracle, 24, max_n=6000) == 24


def test_emul_assign_sort(emul_engine, oracle):
    parity.assign_sort(emul_engine, oracle, rounds=1500)


def test_emul_fastq_zip_host_codecs(emul_engine, oracle):
    parity.fastq_zip_host_codecs(emul_engine, oracle, 240)
    # a paired file: R2 sections identical to R1's are dropped whatever their codec; BZ2 cheap, LZMA dear
    parity.fastq_zip(emul_engine, oracle, 60, host=parity.host_codecs_for_tests(clock_bz2=100.0, clock_lzma=20000.0))


def test_emul_fastq_zip_two_in_flight_small(emul_engine, oracle):
    """two calls in flight (gz_fastq_zip_begin / _end), the smallest form that still has a call begun while the one before it is unfinished -
    in every default run; the longer sequence is the thorough variant below"""
    parity.fastq_zip_two_in_flight(emul_engine, oracle, 12, n_calls=3)


@pytest.mark.thorough
def test_emul_fastq_zip_two_in_flight(emul_engine, oracle):
    parity.fastq_zip_two_in_flight(emul_engine, oracle, 30, n_calls=4)


def test_emul_fastq_zip_errors(emul_engine, oracle):
    parity.fastq_zip_errors(emul_engine, oracle)


def test_emul_fastq_zip_domq(emul_engine, oracle):
    """QUAL through CODEC_DOMQ inside the driver: the file's first VBlock decides (binned scores: a fit), later calls follow even
    with scores that would not fit"""
    parity.fastq_zip(emul_engine, oracle, 72, qual=("bin", "uniform"))


def test_emul_fastq_zip_domq_modes_small(emul_engine, oracle):
    """--force-domq on scores that do not fit and --no-domqual on scores that do, at the smallest size, in every default run"""
    parity.fastq_zip(emul_engine, oracle, 16, n_calls=1, qual=("uniform",), domq=13)
    parity.fastq_zip(emul_engine, oracle, 16, n_calls=1, qual=("bin",), domq=1)


@pytest.mark.thorough
def test_emul_fastq_zip_domq_modes(emul_engine, oracle):
    """forced (--force-domq) on scores that do not fit; refused (--no-domqual) on scores that do; small VBlocks first"""
    parity.fastq_zip(emul_engine, oracle, 36, n_calls=1, qual=("uniform",), domq=13)
    parity.fastq_zip(emul_engine, oracle, 36, n_calls=1, qual=("bin",), domq=1)
    parity.fastq_zip(emul_engine, oracle, 54, qual=("bin", "bin"), small_first=True)


def test_emul_ctx_golden(emul_engine, oracle):
    parity.ctx_golden(emul_engine, oracle)


def test_emul_domq(emul_engine, oracle):
    parity.domq(emul_engine, oracle, 700)


def test_emul_domq_many_lines(emul_engine, oracle):
    """more than 1024 lines per VBlock: every wave of k_domq_scan walks several 64-line tiles, the running position / last
    non-dominant score / offsets pass from tile to tile and from wave to wave"""
    parity.domq(emul_engine, oracle, 2600)


def test_section_order_contexts_out_of_order(emul_engine):
    """a15 is computed in the library (gz_section_order, src/zip.c:247-342,565-585), whatever order the caller's context table is
    in: random tables of contexts (did_i shuffled, every DEP level, merge-made locals, vb_i = 1 and later) == SURVEY A.7 restated"""
    import ctypes as C
    import numpy as np
    from genozip_amd.lib import GzSecOrderIn
    from genozip_amd import synth
    L = emul_engine.L
    for seed in range(40):
        r = synth.u32(7000 + seed, 200)
        n = 1 + int(r[0] % 30)
        dids = np.argsort(r[1:1 + n], kind="stable")                    # a permutation: the table is NOT in did_i order
        ctxs = [(int(dids[i]) * 3 + 1, int(r[40 + i] % 3), bool(r[80 + i] % 4), bool(r[120 + i] % 5 == 0), bool(r[160 + i] % 3)) for i in range(n)]
        arr = (GzSecOrderIn * n)()
        for i, (did, dep, hl, so, hb) in enumerate(ctxs):
            arr[i].did_i, arr[i].local_dep, arr[i].has_local, arr[i].ston_only_local, arr[i].has_b250 = did, dep, int(hl), int(so), int(hb)
        for vb_i in (1, 2, 7):
            out = (C.c_uint32 * (2 * n))()
            k = L.gz_section_order(arr, n, vb_i, out)
            got = [(out[j] // 2, "B" if out[j] & 1 else "L") for j in range(k)]
            assert got == parity._section_order_ref(ctxs, vb_i), (seed, vb_i)


def test_emul_header_layouts(emul_engine):
    """SectionHeaderCtx / VbHeader / TxtHeader and the plan's containers, byte for byte against the reference's own struct definitions"""
    assert parity.header_kats(emul_engine) >= 6


def test_emul_sam_zip(emul_engine, oracle):
    """N1 for SAM: configs[2] from text - 4 VBlocks over 2 calls through the one-line-record plan == the oracle's composition"""
    assert parity.sam_zip(emul_engine, oracle, 200) == 4
    assert parity.sam_zip(emul_engine, oracle, 120, n_calls=1, qual="uniform", aux=False, via_bam=True) == 2      # (from BAM records)


def test_emul_sam_zip_tags_small(emul_engine, oracle):
    assert parity.sam_zip(emul_engine, oracle, 90, tags=True) == 4                                                       # (a context per optional field, in every default run)


@pytest.mark.thorough
def test_emul_sam_zip_tags(emul_engine, oracle):
    assert parity.sam_zip(emul_engine, oracle, 400, tags=True) == 4                                                      # (a context per optional field)


def test_emul_vcf_zip(emul_engine, oracle):
    """N1 for VCF: configs[3] from text - 4 VBlocks over 2 calls through the per-sample plan == the oracle's composition"""
    assert parity.vcf_zip(emul_engine, oracle, 12, 40) == 4


def test_emul_rans_tables(emul_engine, oracle):
    parity.rans_tables(emul_engine, oracle, scale=0.25)


def test_emul_vcf_retest(emul_engine, oracle):
    parity.vcf_retest(emul_engine, oracle, 12, 40)


def test_emul_e2e_files_sha256(emul_engine):
    """the recipe of tests/e2e_files.py still makes the files whose sha256 was recorded when the reference's genounzip decoded them
    (tests/golden/e2e_sha256.json): a change that moves a byte of a whole file shows here, and needs the golden re-made (= re-verified)"""
    import os
    import e2e_files
    import pytest
    if not os.path.exists(os.path.join(e2e_files.ROOT, "oracle", "_ref", "liblzmaref.so")) and not os.path.isdir("/root/reference/src"):
        pytest.skip("oracle/_ref/liblzmaref.so is not built and the reference's sources are not here")
    assert e2e_files.check_against_golden(emul_engine, ["fastq_pair_monochar", "fastq_single_domq_monochar", "vcf"]) == 3


def test_emul_tiled_models(emul_engine, oracle, monkeypatch):
    """GZ_MODEL_TILED=1 (opt-in): k_arith_model_tiled - a workgroup per leaf, contexts sorted inside LDS tiles, records out in
    stream order - gives the oracle's bytes: one piece, position chunks (the models' state through mstate), byte 0 as a symbol, a tile's end
    inside a stream, 64 symbols, and a wide alphabet beside them (the other kernels' leaf)"""
    from hostmem import HostMem
    from genozip_amd.codec import Engine
    from genozip_amd import synth
    monkeypatch.setenv("GZ_MODEL_TILED", "1")
    E = Engine(lib_path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "emul", "libgenozip_amd_emul.so"), mem=HostMem())   # (emul_engine has built it)
    monkeypatch.delenv("GZ_MODEL_TILED")
    items = [(16, synth.markov_bytes(3, 160000, 40, 33).tobytes()), (16, synth.markov_bytes(5, 100, 10, 60).tobytes()), (16, synth.markov_bytes(6, 9000, 64, 0).tobytes()),
             (17, synth.quality_diverse(4, 60).tobytes()), (18, synth.quality_binned(5, 40).tobytes()), (16, bytes(5000)), (16, synth.uniform_bytes(8, 20000, 200).tobytes())]
    got = E.compress_many(items)
    for (c, d), g in zip(items, got):
        assert g == oracle.codec_compress(c, d), (c, len(d))
    E.close()
