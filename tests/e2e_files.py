"""TEST INFRASTRUCTURE: fixed inputs -> whole .genozip files through the product's VBlock driver + global area writer, the same recipe for
the CPU stand-in (tests/emul), where the reference's own `genounzip` reads every file back (tests/golden/make_e2e_golden.py, build
container only), and for the HIP library on the MI355X (tests/test_gpu.py::test_e2e_files_sha256): the files must be the SAME BYTES, so the
sha256 of a file the reference's decoder has accepted (tests/golden/e2e_sha256.json) vouches for the file the GPU wrote.

What a file pins beyond the per-function goldens: a1's special snips, the a4 merge loop / word indices / all-the-same drops, a8's codec
choices as this library makes them, a15's section order, a9 / a16 header values, N1's snips and containers, N4's global area.
NONREF's payload (CODEC_ACGT's sub-codec LZMA: host work outside the path) comes from the reference's vendored LZMA SDK
compiled in place (oracle/_ref/liblzmaref.so, which travels to the GPU box)."""
import ctypes as C
import hashlib
import os

import numpy as np

import parity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "e2e_sha256.json")


def lzma_sub():
    so = os.path.join(ROOT, "oracle", "_ref", "liblzmaref.so")
    if not os.path.exists(so):
        import pyoracle
        pyoracle.build(ref=True)
    if not os.path.exists(so):
        raise RuntimeError("oracle/_ref/liblzmaref.so is not built: NONREF's LZMA payload cannot be made (make -C oracle ref, build container)")
    L = C.CDLL(so)
    L.lzmaref_compress.restype = C.c_long
    L.lzmaref_compress.argtypes = [C.c_char_p, C.c_uint32, C.c_uint64, C.c_char_p, C.c_uint32]

    def compress(data, vb_size):
        out = C.create_string_buffer(len(data) + len(data) // 2 + 10000)
        n = L.lzmaref_compress(data, len(data), vb_size, out, len(out))
        assert n > 0
        return out.raw[:n]
    return compress


def _cut(text, n_parts, lines_per_record=4):
    nl = np.flatnonzero(np.frombuffer(text, dtype=np.uint8) == 10)
    n = len(nl) // lines_per_record
    cuts = [0] + [int(nl[lines_per_record * (n * k // n_parts) - 1]) + 1 for k in range(1, n_parts)] + [len(text)]
    return [(a, b - a) for a, b in zip(cuts, cuts[1:])]


def _zip(E, plan, calls, lzma):
    F = E.zip_open(plan)
    out = []
    for text, vbs in calls:
        got = F.zip_vblocks(text, vbs)
        for g in got:
            g["z"] = F.with_nonref(g, lzma)
        out += got
    return F, out


def fastq_pair(E, lzma, qual="uniform", mono=6, n=300):
    """an R1 / R2 pair, two VBlocks per mate in ONE call: pair-identical drops, flags.paired, R2's mate_lookup node in SQBITMAP
    (fastq.c:664-665), monochar QUAL lines (fastq_qual.c:33-36), N bases in R2 (NONREF_X / CODEC_XCGT)"""
    from genozip_amd import fastq as fq
    r1 = parity.fastq_text(n, seed=51, mate=1, qual=qual, mono=mono)
    r2 = parity.fastq_text(n, seed=51, mate=2, qual_seed=333, qual=qual, dirty_seq=True, mono=mono + (mono > 0))
    p1, p2 = _cut(r1, 2), _cut(r2, 2)
    text = r1 + r2
    vbs = [(p1[0][0], p1[0][1], 1, -1), (p1[1][0], p1[1][1], 2, -1), (len(r1) + p2[0][0], p2[0][1], 3, 0), (len(r1) + p2[1][0], p2[1][1], 4, 1)]
    F, res = _zip(E, fq.illumina_plan(paired=True), [(text, vbs)], lzma)
    blob = F.write_file([dict(name=b"reads_R1.fq", pair=1, vbs=res[:2]), dict(name=b"reads_R2.fq", pair=2, vbs=res[2:])], std_seq_len=150, std_seq_len_r2=150)
    F.close()
    return blob, {"reads_R1.fq": r1, "reads_R2.fq": r2}, []


def fastq_single(E, lzma, qual="bin", mono=5, n=360):
    """one file, 3 VBlocks over 2 calls (the second clones the first's dictionaries), binned scores: the file goes through CODEC_DOMQ,
    monochar lines stay out of its streams"""
    from genozip_amd import fastq as fq
    text = parity.fastq_text(n, seed=41, mate=1, qual=qual, dirty_seq=True, mono=mono)
    parts = _cut(text, 3)
    F, vbs = _zip(E, fq.illumina_plan(paired=False), [(text, [(parts[0][0], parts[0][1], 1, -1)]), (text, [(parts[1][0], parts[1][1], 2, -1), (parts[2][0], parts[2][1], 3, -1)])], lzma)
    blob = F.write_file([dict(name=b"reads.fq", pair=0, vbs=vbs)], std_seq_len=150)
    F.close()
    return blob, {"out.fq": text}, ["-o", "out.fq"]


def fastq_single_domq_gap(E, lzma):
    """binned scores (the file goes through CODEC_DOMQ) with a VBlock in the middle whose EVERY line is one repeated score: CODEC_DOMQ gets no line
    at all there - no QUAL / DOMQRUNS / QUALMPLX / DIVRQUAL locals, and DOMQRUNS' b250 is the one entry WORD_INDEX_EMPTY (the base64 of an empty
    denormalisation table is a snip of length 0, codec_domq.c:240-244, context.c:331-335)"""
    from genozip_amd import fastq as fq
    a = parity.fastq_text(150, seed=71, mate=1, qual="bin", mono=9)
    b = parity.fastq_text(60, seed=72, mate=1, qual="bin", mono=-1)
    c = parity.fastq_text(120, seed=73, mate=1, qual="bin")
    text = a + b + c
    F, vbs = _zip(E, fq.illumina_plan(paired=False), [(text, [(0, len(a), 1, -1)]), (text, [(len(a), len(b), 2, -1), (len(a) + len(b), len(c), 3, -1)])], lzma)
    blob = F.write_file([dict(name=b"reads.fq", pair=0, vbs=vbs)], std_seq_len=150)
    F.close()
    return blob, {"out.fq": text}, ["-o", "out.fq"]


SAM_HEADER = b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:248956422\n@PG\tID:bwa\tPN:bwa\tVN:0.7.17\n"


def sam_tags(E, lzma):
    """aligned reads as SAM text, 4 VBlocks over 2 calls, a context per optional tag behind the AUX container, a header component"""
    from genozip_amd import sam as sm
    plan = sm.sam_plan(has_aux=True, aux_tags=[("NM", "i"), ("AS", "i")])
    F = E.zip_open(plan)
    res, texts, vb_i = [], [], 0
    for call, nr in enumerate((200, 120)):
        text = parity.sam_aligned_text(nr, seed=21 + call, qual="bin", aux=True)
        nl = np.flatnonzero(np.frombuffer(text, dtype=np.uint8) == 10)
        cut = int(nl[(2 * nr) // 3 - 1]) + 1
        got = F.zip_vblocks(text, [(0, cut, vb_i + 1, -1), (cut, len(text) - cut, vb_i + 2, -1)])
        for g in got:
            g["z"] = F.with_nonref(g, lzma)
        res += got
        vb_i += 2
        texts.append(text)
    blob = F.write_file([dict(name=b"reads.sam", pair=0, vbs=res, header=SAM_HEADER)], data_type=2)
    F.close()
    return blob, {"out.sam": SAM_HEADER + b"".join(texts)}, ["-o", "out.sam"]


VCF_HEADER = (b"##fileformat=VCFv4.2\n##contig=<ID=chr1,length=248956422>\n##INFO=<ID=DP,Number=1,Type=Integer,Description=\"d\">\n"
              b"##INFO=<ID=AF,Number=A,Type=Float,Description=\"a\">\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"g\">\n"
              b"##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"d\">\n##FORMAT=<ID=PL,Number=G,Type=Integer,Description=\"p\">\n"
              b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t")


def vcf(E, lzma, NS=24):
    """a multi-sample VCF (GT:DP:PL), 4 VBlocks over 2 calls, FORMAT/DP a transposed matrix, its header text in SEC_TXT_HEADER"""
    from genozip_amd import vcf as vc
    header = VCF_HEADER + b"\t".join(b"S%d" % i for i in range(NS)) + b"\n"
    F = E.zip_open(vc.vcf_plan(NS))
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
    return blob, {"out.vcf": header + b"".join(texts)}, ["-o", "out.vcf"]


CASES = {
    "fastq_pair_monochar": lambda E, lz: fastq_pair(E, lz, "uniform", 6),
    "fastq_pair_domq_monochar": lambda E, lz: fastq_pair(E, lz, "bin", 4, n=240),
    "fastq_single_domq_monochar": lambda E, lz: fastq_single(E, lz, "bin", 5),
    "fastq_single_all_monochar": lambda E, lz: fastq_single(E, lz, "uniform", -1, n=150),
    "fastq_single_domq_vb_all_monochar": fastq_single_domq_gap,
    "sam_tags": sam_tags,
    "vcf": vcf,
}


def sha(blob):
    return hashlib.sha256(blob).hexdigest()


def check_against_golden(E, names=None):
    """-> number of files made and found to be the bytes the reference's decoder has accepted"""
    import json
    want = json.load(open(GOLDEN))["files"]
    lz = lzma_sub()
    n = 0
    for name in names or CASES:
        blob, _texts, _args = CASES[name](E, lz)
        assert name in want, "no golden for %s: python tests/golden/make_e2e_golden.py (build container)" % name
        assert len(blob) == want[name]["size"] and sha(blob) == want[name]["sha256"], \
            "%s: %d bytes, sha256 %s - the file genounzip accepted has %d bytes, sha256 %s" % (name, len(blob), sha(blob), want[name]["size"], want[name]["sha256"])
        n += 1
    return n
