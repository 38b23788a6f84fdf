"""End to end against the REFERENCE'S OWN DECODER (build container only): FASTQ text -> the product's VBlock driver (the unmodified
sources on the CPU stand-in of tests/emul) + global area -> a .genozip file -> the reference's shipped `genounzip` (15.0.86, untarred from
/root/reference/installers into a temporary directory: never into the repo, never to the GPU box) -> the text again, byte for byte.

This is what pins the rows no reference source builds for outside its tree: a1's special snips, the a4 merge loop and
word indices, a8's codec ids in the headers, a9 / a16 header fields, a15 section order as far as the reader depends on it, N1 (the
FASTQ segmenter's snips and containers) and N4 (SEC_TXT_HEADER, SEC_DICT, section list, SEC_GENOZIP_HEADER, footer): the reference
itself reads all of them back into the original text. NONREF's payload (CODEC_ACGT's sub-codec, LZMA: host work outside the path,
SURVEY F8) is made by the reference's vendored LZMA SDK compiled in place (oracle/_ref/liblzmaref.so).

The decoder ends every run in this container with a segmentation fault AFTER the output is complete: its exit path walks the System V
shared memory segments of the machine (ref_cache_iterator, src/ref_cache.c:337-347) and the sandbox has a foreign one (`ipcs -m`: key
0xa3e3ccca, 56 bytes). The binary prints its own call stack when that happens - str_trim <- ref_cache_iterator <- main - and `_run`
accepts exactly that: exit code 0, or a crash whose call stack names ref_cache_iterator under main and nothing of the reading path;
anything else fails the test. The output files are compared in either case.
"""
import ctypes as C
import os
import subprocess
import tarfile

import pytest

import parity

TAR = "/root/reference/installers/genozip-linux-x86_64.tar"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.exists(TAR), reason="the reference's installers are not here (GPU box)")


@pytest.fixture(scope="module")
def genounzip(tmp_path_factory):
    d = tmp_path_factory.mktemp("ref_bin")
    with tarfile.open(TAR) as t:
        t.extractall(d)
    exe = os.path.join(d, "genozip-linux-x86_64", "genounzip")
    assert os.path.exists(exe)
    return exe


@pytest.fixture(scope="module")
def lzma_sub():
    so = os.path.join(ROOT, "oracle", "_ref", "liblzmaref.so")
    if not os.path.exists(so):
        import pyoracle
        pyoracle.build(ref=True)
    L = C.CDLL(so)
    L.lzmaref_compress.restype = C.c_long
    L.lzmaref_compress.argtypes = [C.c_char_p, C.c_uint32, C.c_uint64, C.c_char_p, C.c_uint32]

    def compress(data, vb_size):
        out = C.create_string_buffer(len(data) + len(data) // 2 + 10000)
        n = L.lzmaref_compress(data, len(data), vb_size, out, len(out))
        assert n > 0
        return out.raw[:n]
    return compress


def exit_is_clean_or_the_known_exit_path_crash(returncode, log):
    """0, or the segmentation fault in the exit path's walk over the machine's shared memory segments (the call stack the binary prints)"""
    if returncode == 0:
        return True
    stack = log[log.find("Call stack"):] if "Call stack" in log else ""
    return "ref_cache_iterator" in stack and "main+" in stack and not any(w in stack for w in ("piz_", "zfile_", "reconstruct", "sections_", "dict_io", "ctx_"))


def _run(exe, args, cwd):
    p = subprocess.run([exe] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    log = p.stdout.decode(errors="replace")
    assert exit_is_clean_or_the_known_exit_path_crash(p.returncode, log), (p.returncode, log[-2500:])
    return log


def _zip(E, plan, calls, lzma_sub):
    """calls: list of (text, vbs) -> VBlock results (with NONREF spliced in) in vblock_i order, and the open file"""
    F = E.zip_open(plan)
    out = []
    for text, vbs in calls:
        got = F.zip_vblocks(text, vbs)
        for g in got:
            g["z"] = F.with_nonref(g, lzma_sub)
        out += got
    return F, out


def _cut(text, n_parts):
    """whole reads: -> [(offset, length)]"""
    import numpy as np
    nl = np.flatnonzero(np.frombuffer(text, dtype=np.uint8) == 10)
    n_reads = len(nl) // 4
    cuts = [0] + [int(nl[4 * (n_reads * k // n_parts) - 1]) + 1 for k in range(1, n_parts)] + [len(text)]
    return [(a, b - a) for a, b in zip(cuts, cuts[1:])]


@pytest.mark.parametrize("qual,dirty,mono", [pytest.param("uniform", False, 0, marks=pytest.mark.thorough), ("uniform", True, 0), ("bin", False, 0),
                                             ("uniform", False, 7), ("bin", True, 5), pytest.param("uniform", False, -1, marks=pytest.mark.thorough)])
def test_single_file_round_trip(emul_engine, genounzip, lzma_sub, tmp_path, qual, dirty, mono):
    """one FASTQ file, 3 VBlocks in 2 calls (the second clones the first's dictionaries): reads of different lengths, N bases
    (NONREF_X), binned scores (the file goes through CODEC_DOMQ); mono: QUAL lines of one repeated score - FASTQ_SPECIAL_monochar_QUAL snips
    in QUAL's b250, the lines left out of QUAL.local and of CODEC_DOMQ's streams (fastq_qual.c:33-36,74)"""
    from genozip_amd import fastq as fq
    text = parity.fastq_text(450, seed=41, mate=1, qual=qual, dirty_seq=dirty, mono=mono)
    parts = _cut(text, 3)
    plan = fq.illumina_plan(paired=False)
    F, vbs = _zip(emul_engine, plan, [(text, [(parts[0][0], parts[0][1], 1, -1)]), (text, [(parts[1][0], parts[1][1], 2, -1), (parts[2][0], parts[2][1], 3, -1)])], lzma_sub)
    blob = F.write_file([dict(name=b"reads.fq", pair=0, vbs=vbs)], std_seq_len=150)
    F.close()
    (tmp_path / "reads.fq.genozip").write_bytes(blob)
    log = _run(genounzip, ["-f", "-o", "out.fq", "reads.fq.genozip"], tmp_path)
    out = (tmp_path / "out.fq").read_bytes() if (tmp_path / "out.fq").exists() else b""
    assert out == text, log[:3000]


@pytest.mark.parametrize("qual,mono", [("uniform", 0), ("bin", 0), ("uniform", 6), pytest.param("bin", 4, marks=pytest.mark.thorough)])
def test_paired_files_round_trip(emul_engine, genounzip, lzma_sub, tmp_path, qual, mono):
    """--pair: R1 and R2 as two components of one file (R1's VBlocks 1..n, R2's n+1..2n: the reader pairs vblock_i with vblock_i - n,
    src/writer.c:318-322); R2 sections identical to R1's are left out, R1's carry flags.paired (zfile.c:292-294,323-325); every R2 VBlock's
    SQBITMAP starts with the mate_lookup node (fastq.c:664-665: one more dictionary word); mono: monochar QUAL lines in both mates"""
    from genozip_amd import fastq as fq
    r1 = parity.fastq_text(360, seed=51, mate=1, qual=qual, mono=mono)
    r2 = parity.fastq_text(360, seed=51, mate=2, qual_seed=333, qual=qual, dirty_seq=True, mono=mono + (mono > 0))
    p1, p2 = _cut(r1, 2), _cut(r2, 2)
    text = r1 + r2
    vbs = [(p1[0][0], p1[0][1], 1, -1), (p1[1][0], p1[1][1], 2, -1), (len(r1) + p2[0][0], p2[0][1], 3, 0), (len(r1) + p2[1][0], p2[1][1], 4, 1)]
    plan = fq.illumina_plan(paired=True)
    F, res = _zip(emul_engine, plan, [(text, vbs)], lzma_sub)
    blob = F.write_file([dict(name=b"reads_R1.fq", pair=1, vbs=res[:2]), dict(name=b"reads_R2.fq", pair=2, vbs=res[2:])], std_seq_len=150, std_seq_len_r2=150)
    F.close()
    (tmp_path / "pair.genozip").write_bytes(blob)
    log = _run(genounzip, ["-f", "pair.genozip"], tmp_path)
    got1 = (tmp_path / "reads_R1.fq").read_bytes() if (tmp_path / "reads_R1.fq").exists() else b""
    got2 = (tmp_path / "reads_R2.fq").read_bytes() if (tmp_path / "reads_R2.fq").exists() else b""
    assert got1 == r1 and got2 == r2, (sorted(os.listdir(tmp_path)), log[:3000])


@pytest.mark.parametrize("interleaved", [False, pytest.param(True, marks=pytest.mark.thorough)])
def test_streamed_pairs_round_trip(emul_engine, genounzip, lzma_sub, tmp_path, interleaved):
    """the streamed form (BASELINE configs[4]: one file object, several calls, dictionaries and codecs carried from call to call). A v15 FASTQ
    pair has exactly two components (sections.c:832-835) and R2's VBlock is R1's + the number of R1 VBlocks (writer.c:318-322), so the caller
    numbers R1 1..N and R2 N+1..2N with N known up front, and every call holds some VBlocks of both; the reference's decoder reconstructs both
    texts - whether the file holds R1 whole in front of R2, or the VBlocks in the order the calls made them"""
    from genozip_amd import fastq as fq
    plan = fq.illumina_plan(paired=True)
    K, per = 2, 2
    N = K * per
    r1s = [parity.fastq_text(200, seed=61 + k, mate=1) for k in range(K)]
    r2s = [parity.fastq_text(200, seed=61 + k, mate=2, qual_seed=444 + k) for k in range(K)]
    calls = []
    for k in range(K):
        p1, p2 = _cut(r1s[k], per), _cut(r2s[k], per)
        vbs = [(p1[j][0], p1[j][1], k * per + j + 1, -1) for j in range(per)] + [(len(r1s[k]) + p2[j][0], p2[j][1], N + k * per + j + 1, j) for j in range(per)]
        calls.append((r1s[k] + r2s[k], vbs))
    F, res = _zip(emul_engine, plan, calls, lzma_sub)
    R1 = [r for k in range(K) for r in res[2 * per * k:2 * per * k + per]]
    R2 = [r for k in range(K) for r in res[2 * per * k + per:2 * per * (k + 1)]]
    order = [(m, k * per + j) for k in range(K) for m in (0, 1) for j in range(per)] if interleaved else None
    blob = F.write_file([dict(name=b"reads_R1.fq", pair=1, vbs=R1), dict(name=b"reads_R2.fq", pair=2, vbs=R2)], std_seq_len=150, std_seq_len_r2=150, vb_order=order)
    F.close()
    (tmp_path / "stream.genozip").write_bytes(blob)
    log = _run(genounzip, ["-f", "stream.genozip"], tmp_path)
    for name, want in ((b"reads_R1.fq", b"".join(r1s)), (b"reads_R2.fq", b"".join(r2s))):
        p = tmp_path / name.decode()
        assert p.exists() and p.read_bytes() == want, (name, sorted(os.listdir(tmp_path)), log[:2000])


VCF_HEADER = (b"##fileformat=VCFv4.2\n##contig=<ID=chr1,length=248956422>\n##INFO=<ID=DP,Number=1,Type=Integer,Description=\"d\">\n"
              b"##INFO=<ID=AF,Number=A,Type=Float,Description=\"a\">\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"g\">\n"
              b"##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"d\">\n##FORMAT=<ID=PL,Number=G,Type=Integer,Description=\"p\">\n"
              b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t")


def test_vcf_round_trip(emul_engine, genounzip, tmp_path):
    """a multi-sample VCF (BASELINE configs[3]'s shape: GT:DP:PL per sample) through the VCF plan of genozip_amd/vcf.py - 4 VBlocks over 2 calls,
    VBlock 1 not setting the FORMAT locals' codec - + its header text in SEC_TXT_HEADER: the reference's decoder gives the file back byte for
    byte. What that pins beyond the FASTQ tests: the per-sample columns (lines x samples entries per FORMAT context), the nested SAMPLES
    container (repeats = samples, drop_final_repsep), a7's transposed matrix (FORMAT/DP: LT_UINT8_TR with param 0 = "the file's samples",
    which the reader un-transposes, dyn_int.c / piz side), the integer / delta rules of the fixed fields, a header component"""
    import numpy as np
    from genozip_amd import vcf as vc
    NS = 24
    header = VCF_HEADER + b"\t".join(b"S%d" % i for i in range(NS)) + b"\n"
    plan = vc.vcf_plan(NS)
    F = emul_engine.zip_open(plan)
    res, texts, vb_i = [], [], 0
    for call, nl_ in enumerate((40, 30)):
        text = parity.vcf_full_text(nl_, NS, seed=5 + call)
        nl = np.flatnonzero(np.frombuffer(text, dtype=np.uint8) == 10)
        cut = int(nl[(2 * nl_) // 3 - 1]) + 1
        res += F.zip_vblocks(text, [(0, cut, vb_i + 1, -1), (cut, len(text) - cut, vb_i + 2, -1)])
        vb_i += 2
        texts.append(text)
    blob = F.write_file([dict(name=b"cohort.vcf", pair=0, vbs=res, header=header)], data_type=1)
    F.close()
    (tmp_path / "cohort.vcf.genozip").write_bytes(blob)
    log = _run(genounzip, ["-f", "-o", "out.vcf", "cohort.vcf.genozip"], tmp_path)
    out = (tmp_path / "out.vcf").read_bytes() if (tmp_path / "out.vcf").exists() else b""
    assert out == header + b"".join(texts), log[:3000]


SAM_HEADER = b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:248956422\n@PG\tID:bwa\tPN:bwa\tVN:0.7.17\n"


@pytest.mark.parametrize("qual,aux,dirty,header,tags", [pytest.param("uniform", False, False, b"", False, marks=pytest.mark.thorough), ("bin", True, True, SAM_HEADER, False), ("uniform", True, True, SAM_HEADER, True)])
def test_sam_round_trip(emul_engine, genounzip, lzma_sub, tmp_path, qual, aux, dirty, header, tags):
    """aligned reads as SAM text (BASELINE configs[2]'s shape) through the SAM plan of genozip_amd/sam.py - 4 VBlocks over 2 calls - and the
    reference's decoder gives the text back byte for byte. What that pins beyond the FASTQ / VCF tests: the one-line-record plan's items
    (the eleven mandatory fields by tab, QNAME by its flavor, the optional fields), CIGAR's snips { SNIP_SPECIAL, SAM_SPECIAL_CIGAR } + text
    (src/sam_cigar.c:717-720) which the reader analyses for the lengths of SEQ and QUAL, SQBITMAP's verbatim special (src/sam_seq.c:806-821),
    NONREF with every read padded to whole bytes of the 2-bit packing (:224-229) incl. NONREF_X for bases that are not ACGT, QUAL as LT_BLOB
    and through CODEC_DOMQ with seq_len taken from the CIGAR, FLAG / POS as value-storing contexts, a header component for SAM; tags: the
    optional fields behind the AUX container of sam_seg_aux_all (src/sam_seg.c:1363-1433: "NM:i:" / "AS:i:" as item prefixes) with a context
    per tag instead of one textual item"""
    import random
    import numpy as np
    from genozip_amd import sam as sm
    plan = sm.sam_plan(has_aux=aux, aux_tags=[("NM", "i"), ("AS", "i")] if tags else None)
    F = emul_engine.zip_open(plan)
    res, texts, vb_i = [], [], 0
    for call, nr in enumerate((240, 150)):
        text = parity.sam_aligned_text(nr, seed=21 + call, qual=qual, aux=aux)
        if dirty:                                              # bases that are not ACGT -> NONREF_X (CODEC_XCGT)
            rnd, lines = random.Random(7 + call), []
            for ln in text.split(b"\n")[:-1]:
                fl = ln.split(b"\t")
                sq = bytearray(fl[9])
                for _ in range(rnd.randrange(4)):
                    sq[rnd.randrange(len(sq))] = ord("N")
                fl[9] = bytes(sq)
                lines.append(b"\t".join(fl))
            text = b"\n".join(lines) + b"\n"
        nl = np.flatnonzero(np.frombuffer(text, dtype=np.uint8) == 10)
        cut = int(nl[(2 * nr) // 3 - 1]) + 1
        got = F.zip_vblocks(text, [(0, cut, vb_i + 1, -1), (cut, len(text) - cut, vb_i + 2, -1)])
        for g in got:
            g["z"] = F.with_nonref(g, lzma_sub)
        res += got
        vb_i += 2
        texts.append(text)
    blob = F.write_file([dict(name=b"reads.sam", pair=0, vbs=res, header=header)], data_type=2)
    F.close()
    (tmp_path / "reads.sam.genozip").write_bytes(blob)
    log = _run(genounzip, ["-f", "-o", "out.sam", "reads.sam.genozip"], tmp_path)
    out = (tmp_path / "out.sam").read_bytes() if (tmp_path / "out.sam").exists() else b""
    assert out == header + b"".join(texts), log[:3000]


def test_host_codecs_round_trip(emul_engine, genounzip, lzma_sub, tmp_path):
    """a8 with the host's candidates (gz_zip_set_host_codecs): BZ2 (bzip2 -9) and LZMA (the reference's own LZMA SDK, compiled in place)
    join every trial of the driver; the quality lines repeat, so a coder with a memory wins QUAL (and whatever else it wins): those
    sections are coded on the host, framed by the library - and the reference's genounzip reads the file back, byte for byte"""
    import bz2
    from genozip_amd import fastq as fq
    text0 = parity.fastq_text(450, seed=61, mate=1)
    lines = text0.split(b"\n")
    base = [(lines[3 + 4 * k] * 2)[:200] for k in range(3)]
    for r in range(450):
        lines[4 * r + 3] = base[r % 3][:len(lines[4 * r + 1])]
    text = b"\n".join(lines)
    parts = _cut(text, 3)

    def compress(codec, data):
        return bz2.compress(data, 9) if codec == 3 else lzma_sub(data, 16 << 20)

    def trial(dict_id, is_local, sample):
        return [(3, len(compress(3, sample)), 300.0), (4, len(compress(4, sample)), 9000.0)]
    plan = fq.illumina_plan(paired=False)
    F = emul_engine.zip_open(plan)
    F.set_host_codecs(trial, compress)
    vbs, used = [], set()
    for call in ([(parts[0][0], parts[0][1], 1, -1)], [(parts[1][0], parts[1][1], 2, -1), (parts[2][0], parts[2][1], 3, -1)]):
        got = F.zip_vblocks(text, call)
        for g in got:
            z, p = g["z"], 84
            while p < len(z):
                used.add(z[p + 25]); p += 40 + int.from_bytes(z[p + 12:p + 16], "big")
            g["z"] = F.with_nonref(g, lzma_sub)
        vbs += got
    assert used & {3, 4}, used
    blob = F.write_file([dict(name=b"reads.fq", pair=0, vbs=vbs)], std_seq_len=150)
    F.close()
    (tmp_path / "reads.fq.genozip").write_bytes(blob)
    log = _run(genounzip, ["-f", "-o", "out.fq", "reads.fq.genozip"], tmp_path)
    out = (tmp_path / "out.fq").read_bytes() if (tmp_path / "out.fq").exists() else b""
    assert out == text, (used, log[:3000])
