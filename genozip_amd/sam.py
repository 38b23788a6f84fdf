"""Host-side set-up of the VBlock compute driver for SAM text (BASELINE configs[2]: aligned reads; SURVEY 8f N1 for SAM / BAM), as DATA:
a GzFastqPlan with record_lines = 1 - a record is one line, its items the eleven mandatory tab-separated fields (sam_seg_txt_line,
src/sam_seg.c; the reference's bam_seg_txt_line, src/bam_seg.c:425, walks the same fields in their binary form) with QNAME split further
by its flavor (Illumina-7, src/qname_flavors.h:40-49), and the optional fields as one last item.

    A00123:45:HXXXXXXXX:1:1101:10000:10000 <tab> FLAG RNAME POS MAPQ CIGAR RNEXT PNEXT TLEN SEQ QUAL [<tab> optional fields]

How each field reaches its context follows the reference's segmenter where the driver has the means (SURVEY 8(0), row configs[2]), and
the snips are the reference's own encodings - **the reference's genounzip reconstructs the SAM text from a file made with this plan**
(tests/test_e2e_genounzip.py::test_sam_round_trip):
  QNAME items   as in FASTQ (qname_seg_qf, src/qname.c:715-806): textual / integer in local / self-delta
  FLAG          snips -> dictionary + b250, the context storing the value (sam_seg_FLAG; the reader's last_flags, src/sam_private.h)
  MAPQ RNAME RNEXT   snips -> dictionary + b250 (seg_by_did)
  CIGAR         { SNIP_SPECIAL, SAM_SPECIAL_CIGAR } + the CIGAR text as the snip (sam_seg_CIGAR, src/sam_cigar.c:708-760: the reader's
                sam_cigar_special_CIGAR analyses it - seq_len, reference consumed - for SEQ and QUAL, :940-1033)
  POS           delta against the previous line in a dyn-int local (sam_seg_POS -> seg_pos_field: a delta snip per line in the b250 in
                the reference; here the same deltas as integers in local, the form seg_self_delta gives ordered QNAME items), value stored
  PNEXT TLEN    seg_integer_or_not: dyn-int local ('*' / non-numbers as snips)
  SEQ           no reference genome: sam_seg_SEQ's verbatim branch (src/sam_seq.c:744-757) - the bases into NONREF.local, 'A's after
                every read up to a multiple of 4 (sam_seg_SEQ_pad_nonref, :224-229) -> CODEC_ACGT's 2-bit pack + NONREF_X; SQBITMAP's
                snip { SNIP_SPECIAL, SAM_SPECIAL_SEQ, '0','0','0','0','0','1' }: the last flag is force_verbatim (:806-821,1064,1101-1113)
  QUAL          QUAL.local (LT_BLOB, or LT_CODEC through CODEC_DOMQ when the file's first VBlock is a fit, codec_assign_best_qual_codec,
                src/codec.c:391-450; the reference also considers CODEC_NORMQ for SAM, which is not built), seq_len scores per line
  optional      aux_tags given (a file whose records all carry the same tags in the same order - what segconf would find; BASELINE
                configs[2]'s NM:i AS:i): the AUX container of sam_seg_aux_all (src/sam_seg.c:1363-1433: an item per tag, "TAG:TYPE:" as
                the item's prefix, a tab between items) and a context per tag, dict_id "NM:i" (sam_seg_aux_field) - integers through
                seg_integer_or_not (dyn-int local), the other types as snips; the tag names themselves are GZ_FQ_ITEM_EXPECT items: a record
                with other tags fails the call (the plan does not describe the file). The reference goes further per tag (NM / MD from the
                sequence, AS against the mate's ...): not built. Without aux_tags: everything behind QUAL as one textual item
The TOPLEVEL container has the reference's items in the reference's order (sam_seg_finalize, src/sam_seg.c:572-593) less BUDDY (mates are
not looked up) and with the optional fields as one item. What is NOT the reference's: which fields the reference would predict from
mates, MD / NM from the sequence, the reference-based SEQ - its alignment analysis is out of scope. Parity for this plan: the
CPU restatement's composition in tests/parity.py + the reference's own reader."""
from .fastq import (dict_id, container, container_snip, DTYPE_FIELD, DTYPE_1, STORE_INT, SNIP_SELF_DELTA, SNIP_SPECIAL, CI0_COLONn,
                    CON_FILTER_ITEMS, CON_IS_TOPLEVEL, CON_CALLBACK)
from .fastq import DTYPE_2, CON_PX_SEP, CON_FILTER_REPEATS
from .lib import (GZ_FQ_CONST, GZ_FQ_ITEM_TEXT, GZ_FQ_ITEM_INT, GZ_FQ_ITEM_DELTA, GZ_FQ_SEQ, GZ_FQ_QUAL, GZ_FQ_QUAL_AUX, GZ_FQ_TOPLEVEL, GZ_FQ_ITEM_EXPECT)

CI0_NATIVE_NEXT = 0x88                                                                  # src/container.h:50: reconstructing SAM, the separator is separator[1]

SAM_SPECIAL_CIGAR, SAM_SPECIAL_QUAL, SAM_SPECIAL_SEQ = 32 + 0, 32 + 16, 32 + 18          # src/sam.h:858,874,876 (+32: seg.h:33)


def aux_container(tags):
    """sam_seg_aux_all's container for a record with these optional fields (src/sam_seg.c:1371-1432): item 0 has no context (it carries
    the translator of the container itself for BAM output, not built), then an item per tag { dict_id "TG:t", separator
    { CI0_NATIVE_NEXT, tab } }, the last one without its tab; prefixes: container-wide (empty), item 0's (empty), "TG:t:" per tag
    -> the snip of the AUX context"""
    sep = bytes([CON_PX_SEP])
    items = [(bytes(8), b"")] + [(dict_id("%s:%s" % (t, ty), DTYPE_2), bytes([CI0_NATIVE_NEXT, 9]) if k + 1 < len(tags) else bytes([0x80, 0])) for k, (t, ty) in enumerate(tags)]
    con = container(items, repeats=1, flags=CON_FILTER_REPEATS)
    return container_snip(con, sep + sep + sep + b"".join(("%s:%s:" % (t, ty)).encode() + sep for t, ty in tags))


def sam_plan(has_aux=True, qual_codec=0, estimated_entries=0, domq=0, vb_size=0, aux_tags=None):
    """aux_tags: [("NM", "i"), ("AS", "i")] - every record carries exactly these optional fields, in this order (type as SAM spells it)"""
    P = []
    if aux_tags:
        has_aux = False

    def ctx(tag, did_i, kind, dtype=DTYPE_FIELD, item=0, flags=0, snip=b"", local_dep=0, con_len=0, lcodec=0):
        P.append(dict(tag=tag, dict_id=dict_id(tag, dtype), did_i=did_i, kind=kind, item=item, flags=flags, snip=snip, pair_identical=False, no_stons=False,
                      lcodec=lcodec, bcodec=0, pair_assisted_b250=False, local_dep=local_dep, nothing_char=0, con_len=con_len, segs_per_line=0))

    q1 = container([(dict_id("Q0NAME", DTYPE_1), bytes([CI0_COLONn, 3])), (dict_id("Q1NAME", DTYPE_1), b":"), (dict_id("Q2NAME", DTYPE_1), b":"),
                    (dict_id("Q3NAME", DTYPE_1), b":"), (dict_id("Q4NAME", DTYPE_1), b"")], repeats=1)
    # did_i in the order of the GENDICT lines of src/sam.h (only the relative order matters)
    ctx("QNAME", 1, GZ_FQ_CONST, snip=container_snip(q1))
    ctx("Q0NAME", 2, GZ_FQ_ITEM_TEXT, DTYPE_1, item=0)
    ctx("Q1NAME", 3, GZ_FQ_ITEM_INT, DTYPE_1, item=1)
    ctx("Q2NAME", 4, GZ_FQ_ITEM_TEXT, DTYPE_1, item=2)
    ctx("Q3NAME", 5, GZ_FQ_ITEM_DELTA, DTYPE_1, item=3, flags=STORE_INT, snip=bytes([SNIP_SELF_DELTA]) + b"$")
    ctx("Q4NAME", 6, GZ_FQ_ITEM_DELTA, DTYPE_1, item=4, flags=STORE_INT, snip=bytes([SNIP_SELF_DELTA]) + b"$")
    ctx("AUX", 53, GZ_FQ_ITEM_TEXT, item=15) if has_aux else None
    ctx("SQBITMAP", 54, GZ_FQ_CONST, snip=bytes([SNIP_SPECIAL, SAM_SPECIAL_SEQ]) + b"000001")       # (src/sam_seq.c:806-821)
    ctx("NONREF_X", 56, GZ_FQ_SEQ, local_dep=1)
    ctx("QUAL", 80, GZ_FQ_QUAL, lcodec=qual_codec)
    for k, tag in enumerate(("DOMQRUNS", "QUALMPLX", "DIVRQUAL")):
        ctx(tag, 81 + k, GZ_FQ_QUAL_AUX, item=k, local_dep=2)
    cigar_lead = bytes([SNIP_SPECIAL, SAM_SPECIAL_CIGAR])
    fields = [("FLAG", 100, GZ_FQ_ITEM_TEXT, 5, STORE_INT, b""), ("RNAME", 0, GZ_FQ_ITEM_TEXT, 6, 0, b""), ("POS", 101, GZ_FQ_ITEM_DELTA, 7, STORE_INT, b""),
              ("MAPQ", 102, GZ_FQ_ITEM_TEXT, 8, 0, b""), ("CIGAR", 103, GZ_FQ_ITEM_TEXT, 9, 0, cigar_lead), ("RNEXT", 104, GZ_FQ_ITEM_TEXT, 10, 0, b""),
              ("PNEXT", 105, GZ_FQ_ITEM_INT, 11, 0, b""), ("TLEN", 106, GZ_FQ_ITEM_INT, 12, 0, b"")]
    for tag, did, kind, item, flags, lead in fields:
        ctx(tag, did, kind, item=item, flags=flags, snip=(bytes([SNIP_SELF_DELTA]) + b"$") if kind == GZ_FQ_ITEM_DELTA else lead)
    n_items = 15
    if aux_tags:
        # items behind QUAL: "NM:i" (up to the 2nd ':'), its value (up to the tab), "AS:i", its value ... the last value is the rest
        ctx("AUX", 53, GZ_FQ_CONST, snip=aux_container(aux_tags))
        for k, (t, ty) in enumerate(aux_tags):
            ctx("%s:%s?" % (t, ty), 400 + 2 * k, GZ_FQ_ITEM_EXPECT, item=15 + 2 * k, snip=("%s:%s" % (t, ty)).encode())
            P[-1]["dict_id"] = bytes(8)
            ctx("%s:%s" % (t, ty), 140 + k, GZ_FQ_ITEM_INT if ty == "i" else GZ_FQ_ITEM_TEXT, DTYPE_2, item=16 + 2 * k)
        n_items = 15 + 2 * len(aux_tags)
    order = ["QNAME", "FLAG", "RNAME", "POS", "MAPQ", "CIGAR", "RNEXT", "PNEXT", "TLEN", "SQBITMAP", "QUAL"] + (["AUX"] if has_aux or aux_tags else [])
    top = container([(dict_id(t), b"\t" if i + 1 < len(order) else b"") for i, t in enumerate(order)] + [(dict_id("EOL"), b"")],
                    flags=CON_FILTER_ITEMS | CON_IS_TOPLEVEL | CON_CALLBACK)             # (src/sam_seg.c:572-577)
    ctx("TOPLEVEL", 120, GZ_FQ_TOPLEVEL, snip=top, con_len=len(top))
    ctx("EOL", 121, GZ_FQ_CONST, snip=b"\n")
    P.sort(key=lambda c: c["did_i"])
    seps = b"::::" + b"\t" * (11 if has_aux or aux_tags else 10)
    counts = [3, 1, 1, 1] + [1] * (11 if has_aux or aux_tags else 10)
    if aux_tags:
        for k in range(len(aux_tags)):
            seps += b":" + (b"\t" if k + 1 < len(aux_tags) else b"")
            counts += [2] + ([1] if k + 1 < len(aux_tags) else [])
    assert len(seps) == n_items - 1 or not aux_tags
    return dict(ctxs=P, seps=seps, sep_counts=counts, paired=False, estimated_entries=estimated_entries, qual_codec=domq, vb_size=vb_size, line3_empty=0, vb_1_not_representative=0b100,
                record_lines=1, seq_item=13, qual_item=14, seq_pad=4)
