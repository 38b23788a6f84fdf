"""Run ON THE GPU BOX (gpurun): .genozip files made by the HIP library - a FASTQ pair, a SAM file from BAM records, a multi-sample VCF -
into gpurun_out/e2e/. tools/e2e_check_local.py (build container) then offers them to the reference's own genounzip and compares the text
with what the same generators give there (numpy on both sides: identical bytes). The NONREF payload (LZMA, host work outside the path)
comes from the reference's vendored LZMA SDK compiled in place (oracle/_ref/liblzmaref.so, which travels with the snapshot).
Sizes: argv[1] reads per mate / alignments (default 100000)."""
import ctypes as C
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np                                               # noqa: E402


def texts(n):
    """the inputs, from the vectorised generators (host side)"""
    from genozip_amd import workload as W
    import parity
    r1 = W.fastq_text(11, 0, n, mate=1, profile="div")
    r2 = W.fastq_text(11, 0, n, mate=2, profile="div", n_rate=3)
    sam = W.sam_text(3, 0, n, profile="bin")
    bam = W.bam_records(3, 0, n, profile="bin")
    vcf_hdr = (b"##fileformat=VCFv4.2\n##contig=<ID=chr1,length=248956422>\n##INFO=<ID=DP,Number=1,Type=Integer,Description=\"d\">\n"
               b"##INFO=<ID=AF,Number=A,Type=Float,Description=\"a\">\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"g\">\n"
               b"##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"d\">\n##FORMAT=<ID=PL,Number=G,Type=Integer,Description=\"p\">\n"
               b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t") + b"\t".join(b"S%d" % i for i in range(500)) + b"\n"
    vcf = [parity.vcf_full_text(120, 500, seed=5 + k) for k in range(2)]
    return r1, r2, sam, bam, vcf_hdr, vcf


def cut_lines(text, n_parts, lines_per_record):
    nl = np.flatnonzero(np.frombuffer(text, dtype=np.uint8) == 10)
    n_rec = len(nl) // lines_per_record
    cuts = [0] + [int(nl[lines_per_record * (n_rec * k // n_parts) - 1]) + 1 for k in range(1, n_parts)] + [len(text)]
    return [(a, b - a) for a, b in zip(cuts, cuts[1:])]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    out = os.path.join(ROOT, "gpurun_out", "e2e")
    os.makedirs(out, exist_ok=True)
    from genozip_amd.codec import Engine
    from genozip_amd import fastq as fq, sam as sm, vcf as vc
    if os.environ.get("E2E_EMUL"):                               # (a dry run of this script in the build container, on the CPU stand-in)
        sys.path.insert(0, os.path.join(ROOT, "tests", "emul"))
        from hostmem import HostMem
        E = Engine(lib_path=os.path.join(ROOT, "tests", "emul", "libgenozip_amd_emul.so"), mem=HostMem())
    else:
        E = Engine(device=0)
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "liblzmaref.so"))
    L.lzmaref_compress.restype = C.c_long
    L.lzmaref_compress.argtypes = [C.c_char_p, C.c_uint32, C.c_uint64, C.c_char_p, C.c_uint32]

    def lz(data, vb_size):
        buf = C.create_string_buffer(len(data) + len(data) // 2 + 10000)
        k = L.lzmaref_compress(data, len(data), vb_size, buf, len(buf))
        assert k > 0
        return buf.raw[:k]

    r1, r2, sam, bam, vcf_hdr, vcf = texts(n)
    rec = {"n": n, "version": E.version(), "engine": "the CPU stand-in of tests/emul (dry run)" if os.environ.get("E2E_EMUL") else "HIP on cuda:0"}
    # FASTQ pair: R1 VBlocks 1..4, R2 5..8, one call
    p1, p2 = cut_lines(r1, 4, 4), cut_lines(r2, 4, 4)
    text = r1 + r2
    vbs = [(o, l, i + 1, -1) for i, (o, l) in enumerate(p1)] + [(len(r1) + o, l, 5 + i, i) for i, (o, l) in enumerate(p2)]
    F = E.zip_open(fq.illumina_plan(paired=True))
    res = F.zip_vblocks(text, vbs)
    for g in res:
        g["z"] = F.with_nonref(g, lz)
    blob = F.write_file([dict(name=b"reads_R1.fq", pair=1, vbs=res[:4]), dict(name=b"reads_R2.fq", pair=2, vbs=res[4:])], std_seq_len=150, std_seq_len_r2=150)
    F.close()
    open(os.path.join(out, "pair.genozip"), "wb").write(blob)
    rec["fastq"] = {"file": "pair.genozip", "bytes": len(blob), "R1_sha256": hashlib.sha256(r1).hexdigest(), "R2_sha256": hashlib.sha256(r2).hexdigest(), "text_bytes": len(text)}
    # SAM from BAM records: gz_bam_records + gz_bam_to_sam on the device, then the SAM plan; 3 VBlocks over 2 calls
    hdr = b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:248956422\n"
    made, lo = E.bam_to_sam(bam, E.bam_records(bam, 1), [b"chr1"])
    assert made == sam
    parts = cut_lines(made, 3, 1)
    F = E.zip_open(sm.sam_plan(has_aux=True))
    res = F.zip_vblocks(made, [(parts[0][0], parts[0][1], 1, -1)]) + F.zip_vblocks(made, [(parts[1][0], parts[1][1], 2, -1), (parts[2][0], parts[2][1], 3, -1)])
    for g in res:
        g["z"] = F.with_nonref(g, lz)
    blob = F.write_file([dict(name=b"reads.sam", pair=0, vbs=res, header=hdr)], data_type=2)
    F.close()
    open(os.path.join(out, "reads.sam.genozip"), "wb").write(blob)
    rec["sam"] = {"file": "reads.sam.genozip", "bytes": len(blob), "sha256": hashlib.sha256(hdr + sam).hexdigest(), "text_bytes": len(hdr) + len(sam), "bam_bytes": len(bam)}
    # VCF: 2 calls x 2 VBlocks
    F = E.zip_open(vc.vcf_plan(500))
    res, vb_i = [], 0
    for t in vcf:
        p = cut_lines(t, 2, 1)
        res += F.zip_vblocks(t, [(p[0][0], p[0][1], vb_i + 1, -1), (p[1][0], p[1][1], vb_i + 2, -1)])
        vb_i += 2
    blob = F.write_file([dict(name=b"cohort.vcf", pair=0, vbs=res, header=vcf_hdr)], data_type=1)
    F.close()
    open(os.path.join(out, "cohort.vcf.genozip"), "wb").write(blob)
    rec["vcf"] = {"file": "cohort.vcf.genozip", "bytes": len(blob), "sha256": hashlib.sha256(vcf_hdr + b"".join(vcf)).hexdigest(), "text_bytes": len(vcf_hdr) + sum(len(t) for t in vcf)}
    json.dump(rec, open(os.path.join(out, "made.json"), "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
