"""Host-side mirror of the reference's FASTQ segmenter set-up for the VBlock compute driver (gz_fastq_zip_vblocks):
what segconf + fastq_seg_initialize + qname_seg_initialize decide once per file, as DATA (a GzFastqPlan).

illumina_plan(): the Illumina-7 QNAME flavor with an Illum-2bc QNAME2 (src/qname_flavors.h:40-49,194-200,1095,1205), e.g.
    @A00123:45:HXXXXXXXX:1:1101:10000:10000 1:N:0:ACGTACGT+TGCATGCA
line 1 is one container of 8 items:  instrument:run:flowcell (3rd ':') | lane ':' | tile ':' | x ':' | y ' ' | read:filter:control (3rd ':')
| barcode1 '+' | barcode2. Per src/qname.c:715-806 and the flavor table: Q0NAME textual; Q1NAME (lane) integer in local;
Q2NAME (tile) textual; Q3NAME / Q4NAME (x, y) the ordered items -> seg_self_delta (FASTQ is always sorted_by_qname,
qname.c:415); QNAME2's items textual.
The constant snips are the reference's own encodings, built here from the format (not opaque): the containers of QNAME / QNAME2
(qname_zip_initialize src/qname.c:223-232 -> container_prepare_snip src/container.c:35-64), the TOPLEVEL container of a file whose
line 3 is a bare '+' (fastq_seg_finalize src/fastq.c:845-943), SEQ's special snip (src/fastq_seq.c:139-146), end of line "\n"
(SEG_EOL src/seg.h:344). With them the reference's genounzip reconstructs the text (tests/test_e2e_genounzip.py).
"""
import base64
import ctypes as C
import struct

from .lib import (GzFastqCtx, GzFastqPlan, GZ_FQ_CONST, GZ_FQ_ITEM_TEXT, GZ_FQ_ITEM_INT, GZ_FQ_ITEM_DELTA, GZ_FQ_SEQ, GZ_FQ_QUAL, GZ_FQ_QUAL_AUX,
                  GZ_FQ_TOPLEVEL, GZ_FQ_SEQ_SNIP)

DTYPE_FIELD, DTYPE_1, DTYPE_2 = 0, 1, 2
STORE_INT = 1
SNIP_LOOKUP, SNIP_CONTAINER, SNIP_SELF_DELTA, SNIP_SPECIAL = 1, 4, 5, 8               # src/context.h:33-41
FASTQ_SPECIAL_unaligned_SEQ, FASTQ_SPECIAL_mate_lookup, FASTQ_SPECIAL_monochar_QUAL = 32 + 0, 32 + 2, 32 + 9   # src/dict_id_gen.h:2483 (+32: seg.h:33)
CON_PX_SEP = 4                                                                          # src/container.h:15
CI0_COLONn = 8                                                                          # src/container.h:41
# Container flag bits (src/container.h:81-90, the byte behind nitems_lo)
CON_FILTER_REPEATS, CON_FILTER_ITEMS, CON_IS_TOPLEVEL, CON_CALLBACK = 1 << 2, 1 << 3, 1 << 4, 1 << 6
CON_DROP_FINAL_REPSEP = 1 << 1


def dict_id(tag, dtype=DTYPE_FIELD):
    """dict_id_make (src/dict_id.c:14-40): 8 bytes (a longer tag: its first and last 4), the type in the two top bits of the first:
    FIELD 00, TYPE_1 11, TYPE_2 as is (01)"""
    t = tag.encode()
    b = bytearray((t + b"\0" * 8)[:8] if len(t) <= 8 else t[:4] + t[-4:])
    b[0] = (b[0] & 0x3f) if dtype == DTYPE_FIELD else (b[0] | 0xc0) if dtype == DTYPE_1 else b[0]
    return bytes(b)


def container(items, repeats=0, flags=0, repsep=b"\0\0"):
    """the binary Container (src/container.h:74-92, packed): a little-endian word { nitems_hi : 3, unused : 4, no_translation : 1,
    repeats : 24 }, nitems_lo, the flag byte, repsep[2], then 12 bytes per item { dict_id[8], did_i_small (PIZ only), separator[2],
    translator }. items: [(dict_id bytes, separator bytes of length <= 2)]"""
    n = len(items)
    out = struct.pack("<I", (n >> 8) | (repeats << 8)) + bytes([n & 0xff, flags]) + (repsep + b"\0\0")[:2]
    for did, sep in items:
        out += did + b"\0" + (bytes(sep) + b"\0\0")[:2] + b"\0"
    return out


def container_snip(con, prefixes=b""):
    """container_prepare_snip (src/container.c:35-64)"""
    return bytes([SNIP_CONTAINER]) + base64.b64encode(con) + prefixes


def fastq_toplevel(has_qname2=True, dc=b"@"):
    """-> (binary container with repeats = 0, prefixes) of fastq_seg_finalize (src/fastq.c:845-943) for a file that is not --deep,
    has no EXTRA / AUX / SAUX fields, whose description sits on line 1 and whose line 3 is a bare '+' (L3_EMPTY: the LINE3 item goes and
    '+' becomes the prefix of the following E2L, :925) and without a seq_len item. What is left of the 16 items: QNAME, [QNAME2,] E1L,
    SQBITMAP, E2L, E2L, QUAL, E2L; prefixes: container-wide (empty), '@' for QNAME, ' ' for QNAME2, '+' for the second E2L"""
    sep = bytes([CON_PX_SEP])
    items = [("QNAME", dc)] + ([("QNAME2", b" ")] if has_qname2 else []) + [("E1L", b""), ("SQBITMAP", b""), ("E2L", b""), ("E2L", b"+"), ("QUAL", b""), ("E2L", b"")]
    con = container([(dict_id(t), b"") for t, _ in items], flags=CON_FILTER_REPEATS | CON_FILTER_ITEMS | CON_IS_TOPLEVEL | CON_CALLBACK)
    prefixes = sep + sep + b"".join(px + sep for _, px in items) + sep          # start, container-wide, one per item, end
    return con, prefixes


# did_i follow the order of the #pragma GENDICT lines of src/sam.h:19-140 (FASTQ shares SAM's Dids, src/fastq.h:13-60);
# only their relative order matters here (sections appear in ascending did_i)
def illumina_plan(paired=True, qual_codec=0, estimated_entries=0, domq=0, vb_size=0):
    """-> list of dict(tag, dict_id, did_i, kind, item, flags, snip, ...) for GzFastqPlan.
    qual_codec: hard-coded coder of the QUAL stream (0: codec_assign_best_codec); domq: 0 the reference's own rule (the file's first
    VBlock decides, codec.c:391-450), 1 (CODEC_NONE) --no-domqual, 13 (CODEC_DOMQ) --force-domq; vb_size: segconf.vb_size (VBlocks
    of at most MIN (4 MB, vb_size / 2) of text do not set codecs for the file, codec.c:352; 0: every VBlock may)"""
    P = []

    def ctx(tag, did_i, kind, dtype=DTYPE_FIELD, item=0, flags=0, snip=b"", pair_identical=False, no_stons=False, lcodec=0, bcodec=0,
            pair_assisted_b250=False, local_dep=0, nothing_char=0, con_len=0, segs_per_line=0, r2_node=b""):
        P.append(dict(tag=tag, dict_id=dict_id(tag, dtype), did_i=did_i, kind=kind, item=item, flags=flags, snip=snip,
                      pair_identical=pair_identical, no_stons=no_stons or (paired and tag[0] in "Qq" and tag != "QUAL"), lcodec=lcodec, bcodec=bcodec,
                      pair_assisted_b250=pair_assisted_b250, local_dep=local_dep, nothing_char=nothing_char, con_len=con_len, segs_per_line=segs_per_line,
                      r2_node=r2_node))

    pi = True                          # fastq_zip_use_pair_identical: QNAME subfields, QNAME2, LINE3, E1L, E2L, TOPLEVEL (fastq.c:238-243)
    q1 = container([(dict_id("Q0NAME", DTYPE_1), bytes([CI0_COLONn, 3])), (dict_id("Q1NAME", DTYPE_1), b":"), (dict_id("Q2NAME", DTYPE_1), b":"),
                    (dict_id("Q3NAME", DTYPE_1), b":"), (dict_id("Q4NAME", DTYPE_1), b"")], repeats=1)          # con_illumina_7 without its unused mate item (qname.c:44-60)
    q2 = container([(dict_id("q0NAME", DTYPE_1), bytes([CI0_COLONn, 3])), (dict_id("q1NAME", DTYPE_1), b"+"), (dict_id("q2NAME", DTYPE_1), b"")], repeats=1)   # con_qname2_2bc
    ctx("QNAME", 1, GZ_FQ_CONST, snip=container_snip(q1), pair_identical=pi, no_stons=True)
    ctx("Q0NAME", 2, GZ_FQ_ITEM_TEXT, DTYPE_1, item=0, pair_identical=pi)
    ctx("Q1NAME", 3, GZ_FQ_ITEM_INT, DTYPE_1, item=1, pair_identical=pi)
    ctx("Q2NAME", 4, GZ_FQ_ITEM_TEXT, DTYPE_1, item=2, pair_identical=pi)
    ctx("Q3NAME", 5, GZ_FQ_ITEM_DELTA, DTYPE_1, item=3, flags=STORE_INT, snip=bytes([SNIP_SELF_DELTA]) + b"$", pair_identical=pi)
    ctx("Q4NAME", 6, GZ_FQ_ITEM_DELTA, DTYPE_1, item=4, flags=STORE_INT, snip=bytes([SNIP_SELF_DELTA]) + b"$", pair_identical=pi)
    ctx("QNAME2", 18, GZ_FQ_CONST, snip=container_snip(q2), pair_identical=pi, no_stons=True)
    ctx("q0NAME", 19, GZ_FQ_ITEM_TEXT, DTYPE_1, item=5, pair_identical=pi)
    ctx("q1NAME", 20, GZ_FQ_ITEM_TEXT, DTYPE_1, item=6, pair_identical=pi)
    ctx("q2NAME", 21, GZ_FQ_ITEM_TEXT, DTYPE_1, item=7, pair_identical=pi)
    # every R2 VBlock creates the mate_lookup node in SQBITMAP before it segs a read (fastq_seg_initialize, fastq.c:664-665)
    ctx("SQBITMAP", 54, GZ_FQ_SEQ_SNIP, snip=bytes([SNIP_SPECIAL, FASTQ_SPECIAL_unaligned_SEQ]) + b" ", pair_assisted_b250=True,
        r2_node=bytes([SNIP_SPECIAL, FASTQ_SPECIAL_mate_lookup]) if paired else b"")
    ctx("NONREF_X", 56, GZ_FQ_SEQ, local_dep=1)                        # (NONREF itself: did_i 55, sam.h:77-79)
    # a line of one repeated score segs { SNIP_SPECIAL, FASTQ_SPECIAL_monochar_QUAL, score } and stays out of QUAL.local, all others
    # { SNIP_LOOKUP } (fastq_seg_QUAL, fastq_qual.c:24-47)
    ctx("QUAL", 80, GZ_FQ_QUAL, lcodec=qual_codec, snip=bytes([SNIP_SPECIAL, FASTQ_SPECIAL_monochar_QUAL]))
    for k, tag in enumerate(("DOMQRUNS", "QUALMPLX", "DIVRQUAL")):     # "these 3 must be right after SAM_QUAL" (src/sam.h:108-110)
        ctx(tag, 81 + k, GZ_FQ_QUAL_AUX, item=k, local_dep=2)
    top, top_px = fastq_toplevel(has_qname2=True)
    ctx("TOPLEVEL", 90, GZ_FQ_TOPLEVEL, snip=top + top_px, con_len=len(top), pair_identical=pi, no_stons=True)
    ctx("E1L", 96, GZ_FQ_CONST, snip=b"\n", pair_identical=pi)
    ctx("E2L", 97, GZ_FQ_CONST, snip=b"\n", pair_identical=pi, segs_per_line=3)      # the ends of lines 2, 3 and 4 (fastq.c:1300-1304)
    return dict(ctxs=P, seps=b":::: :+", sep_counts=[3, 1, 1, 1, 1, 3, 1], paired=paired, estimated_entries=estimated_entries, qual_codec=domq, vb_size=vb_size,
                line3_empty=1)


def c_plan(plan):
    """dict plan -> (GzFastqPlan, keep-alive list)"""
    n = len(plan["ctxs"])
    arr = (GzFastqCtx * n)()
    keep = [arr]
    for i, c in enumerate(plan["ctxs"]):
        a = arr[i]
        a.dict_id = (C.c_uint8 * 8)(*c["dict_id"])
        a.did_i, a.kind, a.item, a.local_dep, a.flags = c["did_i"], c["kind"], c["item"], c["local_dep"], c["flags"]
        a.no_stons, a.lcodec, a.bcodec = int(c["no_stons"]), c["lcodec"], c["bcodec"]
        a.pair_identical, a.pair_assisted_b250, a.nothing_char = int(c["pair_identical"]), int(c["pair_assisted_b250"]), c["nothing_char"]
        a.snip, a.snip_len = c["snip"], len(c["snip"])
        a.con_len, a.segs_per_line = c.get("con_len", 0), c.get("segs_per_line", 0)
        a.per_sample, a.transposed = int(c.get("per_sample", 0)), int(c.get("transposed", 0))
        a.r2_node, a.r2_node_len = c.get("r2_node") or None, len(c.get("r2_node") or b"")
        keep.append(c["snip"])
        keep.append(c.get("r2_node"))
    p = GzFastqPlan()
    p.ctxs, p.n_ctxs = arr, n
    p.seps = plan["seps"]
    p.sep_counts = (C.c_uint8 * 32)(*(list(plan["sep_counts"]) + [0] * (32 - len(plan["sep_counts"]))))
    p.n_seps, p.paired, p.estimated_entries = len(plan["seps"]), int(plan["paired"]), plan["estimated_entries"]
    p.qual_codec = plan.get("qual_codec", 0)
    p.vb_size = plan.get("vb_size", 0)
    p.line3_empty = plan.get("line3_empty", 0)
    p.vb_1_not_representative = plan.get("vb_1_not_representative", 0)
    p.seq_pad = plan.get("seq_pad", 0)
    p.record_lines, p.seq_item, p.qual_item = plan.get("record_lines", 0), plan.get("seq_item", 0), plan.get("qual_item", 0)
    p.n_samples, p.n_subfields = plan.get("n_samples", 0), plan.get("n_subfields", 0)
    return p, keep
