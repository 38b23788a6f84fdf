"""Host-side helpers for BAM input (SURVEY 8(0) configs[2]; N1 for BAM). The VBlock compute driver takes the alignment RECORDS of an
uncompressed BAM stream (gz_bam_records / gz_bam_to_sam in the library turn them into alignment lines in HBM, which the one-line-record
plan of genozip_amd/sam.py segs); what sits in front of the first record - magic, header text, reference names - is the txt-header
component and is read here on the host (bam_header..., src/sam_header.c; the BAM format: SAMv1 section 4.2). The BGZF layer around a
.bam file is I/O (out of scope): these functions see the inflated stream.

sam_to_bam() is the inverse of the library's conversion for the TESTS and the bench's workload: SAM text -> BAM records (samtools'
encoding rules: integers in the smallest type that holds them, bin by reg2bin, 4-bit bases, Phred scores)."""
import struct

BASES = "=ACMGRSVTWYHKDBN"
CIGAR_OPS = "MIDNSHP=X"


def parse_header(stream):
    """stream: bytes of the inflated BAM file -> (header text, [reference names], [reference lengths], offset of the first alignment)"""
    if stream[:4] != b"BAM\1":
        raise ValueError("not a BAM stream")
    l_text, = struct.unpack_from("<i", stream, 4)
    text = stream[8:8 + l_text]
    at = 8 + l_text
    n_ref, = struct.unpack_from("<i", stream, at)
    at += 4
    names, lens = [], []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", stream, at)
        names.append(stream[at + 4:at + 4 + l_name - 1])
        lens.append(struct.unpack_from("<i", stream, at + 4 + l_name)[0])
        at += 8 + l_name
    return text.rstrip(b"\0"), names, lens, at


def make_header(text, names, lens):
    out = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(names))
    for n, l in zip(names, lens):
        out += struct.pack("<i", len(n) + 1) + n + b"\0" + struct.pack("<i", l)
    return out


def reg2bin(beg, end):
    """SAMv1 5.3"""
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def _aux(field):
    tag, typ, val = field.split(b":", 2)
    if typ == b"A":
        return tag + b"A" + val[:1]
    if typ == b"i":
        v = int(val)
        for code, fmt, lo, hi in ((b"C", "<B", 0, 255), (b"c", "<b", -128, 127), (b"S", "<H", 0, 65535), (b"s", "<h", -32768, 32767), (b"I", "<I", 0, 2 ** 32 - 1), (b"i", "<i", -2 ** 31, 2 ** 31 - 1)):
            if lo <= v <= hi:
                return tag + code + struct.pack(fmt, v)
        raise ValueError("integer out of range")
    if typ in (b"Z", b"H"):
        return tag + typ + val + b"\0"
    if typ == b"B":
        st, *vals = val.split(b",")
        fmt = {b"c": "<b", b"C": "<B", b"s": "<h", b"S": "<H", b"i": "<i", b"I": "<I"}[st]
        return tag + b"B" + st + struct.pack("<I", len(vals)) + b"".join(struct.pack(fmt, int(v)) for v in vals)
    raise ValueError("optional field type %r is not handled" % typ)


def sam_to_bam(text, ref_names):
    """alignment lines (no header lines) -> the BAM records, concatenated"""
    rid = {n: i for i, n in enumerate(ref_names)}
    code = {ord(c): i for i, c in enumerate(BASES)}
    out = []
    for line in text.split(b"\n"):
        if not line:
            continue
        f = line.split(b"\t")
        qname, flag, rname, pos, mapq, cigar, rnext, pnext, tlen, seq, qual = f[:11]
        ref_id = -1 if rname == b"*" else rid[rname]
        next_ref = -1 if rnext == b"*" else ref_id if rnext == b"=" else rid[rnext]
        ops, num, ref_len = [], 0, 0
        if cigar != b"*":
            for ch in cigar:
                if 48 <= ch <= 57:
                    num = num * 10 + ch - 48
                else:
                    op = CIGAR_OPS.index(chr(ch))
                    ops.append(num << 4 | op)
                    if op in (0, 2, 3, 7, 8):
                        ref_len += num
                    num = 0
        l_seq = 0 if seq == b"*" else len(seq)
        sq = bytearray((l_seq + 1) // 2)
        for i in range(l_seq):
            sq[i >> 1] |= code.get(seq[i], 15) << (0 if i & 1 else 4)
        ql = b"\xff" * l_seq if qual == b"*" else bytes(q - 33 for q in qual)
        p0 = int(pos) - 1
        body = struct.pack("<iiBBHHHIiii", ref_id, p0, len(qname) + 1, int(mapq), reg2bin(p0, p0 + max(1, ref_len)) if p0 >= 0 else 4680, len(ops), int(flag), l_seq,
                           next_ref, int(pnext) - 1, int(tlen))
        body += qname + b"\0" + b"".join(struct.pack("<I", o) for o in ops) + bytes(sq) + ql + b"".join(_aux(a) for a in f[11:])
        out.append(struct.pack("<I", len(body)) + body)
    return b"".join(out)
