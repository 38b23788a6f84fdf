"""Host-side mirror of the reference's FASTQ segmenter set-up for the VBlock compute driver (gz_fastq_zip_vblocks):
what segconf + fastq_seg_initialize + qname_seg_initialize decide once per file, as DATA (a GzFastqPlan).

illumina_plan(): the Illumina-7 QNAME flavor with an Illum-2bc QNAME2 (src/qname_flavors.h:40-49,1095,1205), e.g.
    @A00123:45:HXXXXXXXX:1:1101:10000:10000 1:N:0:ACGTACGT+TGCATGCA
line 1 is one container of 8 items:  instrument:run:flowcell (3rd ':') | lane ':' | tile ':' | x ':' | y ' ' | read:filter:control (3rd ':')
| barcode1 '+' | barcode2. Per src/qname.c:715-806 and the flavor table: Q0NAME textual; Q1NAME (lane) integer in local;
Q2NAME (tile) textual; Q3NAME / Q4NAME (x, y) the ordered items -> seg_self_delta (FASTQ is always sorted_by_qname,
qname.c:415); QNAME2's items textual. Constant snips (containers, SEQ's special snip, line 3, EOLs, TOPLEVEL) are what
the reference builds at segconf time: they are opaque bytes here (their exact content is container / special-snip encoding
of the reference's segmenter, SURVEY 2 OUT OF SCOPE) - every line segs the same one, so the context is all-the-same.
"""
import ctypes as C

from .lib import (GzFastqCtx, GzFastqPlan, GZ_FQ_CONST, GZ_FQ_ITEM_TEXT, GZ_FQ_ITEM_INT, GZ_FQ_ITEM_DELTA, GZ_FQ_SEQ, GZ_FQ_QUAL, GZ_FQ_QUAL_AUX)

DTYPE_FIELD, DTYPE_1, DTYPE_2 = 0, 1, 2
STORE_INT = 1
SNIP_SELF_DELTA, SNIP_CONTAINER, SNIP_SPECIAL = 5, 4, 8


def dict_id(tag, dtype=DTYPE_FIELD):
    """dict_id_make (src/dict_id.h:17-19): 8 bytes, the type in the two top bits of the first"""
    b = bytearray((tag.encode() + b"\0" * 8)[:8])
    b[0] = (b[0] & 0x3f) if dtype == DTYPE_FIELD else (b[0] | 0x80) if dtype == DTYPE_1 else b[0]
    return bytes(b)


# did_i follow the order of the #pragma GENDICT lines of src/sam.h:19-86 (FASTQ shares SAM's Dids, src/fastq.h:13-60);
# only their relative order matters here (sections appear in ascending did_i)
def illumina_plan(paired=True, qual_codec=0, estimated_entries=0, domq=0, vb_size=0):
    """-> list of dict(tag, dict_id, did_i, kind, item, flags, snip, ...) for GzFastqPlan.
    qual_codec: hard-coded coder of the QUAL stream (0: codec_assign_best_codec); domq: 0 the reference's own rule (the file's first
    VBlock decides, codec.c:391-450), 1 (CODEC_NONE) --no-domqual, 13 (CODEC_DOMQ) --force-domq; vb_size: segconf.vb_size (VBlocks
    of at most MIN (4 MB, vb_size / 2) of text do not set codecs for the file, codec.c:352; 0: every VBlock may)"""
    P = []

    def ctx(tag, did_i, kind, dtype=DTYPE_FIELD, item=0, flags=0, snip=b"", pair_identical=False, no_stons=False, lcodec=0, bcodec=0,
            pair_assisted_b250=False, local_dep=0, nothing_char=0):
        P.append(dict(tag=tag, dict_id=dict_id(tag, dtype), did_i=did_i, kind=kind, item=item, flags=flags, snip=snip,
                      pair_identical=pair_identical, no_stons=no_stons or (paired and tag[0] in "Qq" and tag != "QUAL"), lcodec=lcodec, bcodec=bcodec,
                      pair_assisted_b250=pair_assisted_b250, local_dep=local_dep, nothing_char=nothing_char))

    pi = True                          # fastq_zip_use_pair_identical: QNAME subfields, QNAME2, LINE3, E1L, E2L, TOPLEVEL (fastq.c:238-243)
    ctx("QNAME", 1, GZ_FQ_CONST, snip=bytes([SNIP_CONTAINER]) + b"<illumina-7 container>", pair_identical=pi)
    ctx("Q0NAME", 2, GZ_FQ_ITEM_TEXT, DTYPE_1, item=0, pair_identical=pi)
    ctx("Q1NAME", 3, GZ_FQ_ITEM_INT, DTYPE_1, item=1, pair_identical=pi)
    ctx("Q2NAME", 4, GZ_FQ_ITEM_TEXT, DTYPE_1, item=2, pair_identical=pi)
    ctx("Q3NAME", 5, GZ_FQ_ITEM_DELTA, DTYPE_1, item=3, flags=STORE_INT, snip=bytes([SNIP_SELF_DELTA]) + b"$", pair_identical=pi)
    ctx("Q4NAME", 6, GZ_FQ_ITEM_DELTA, DTYPE_1, item=4, flags=STORE_INT, snip=bytes([SNIP_SELF_DELTA]) + b"$", pair_identical=pi)
    ctx("QNAME2", 18, GZ_FQ_CONST, snip=bytes([SNIP_CONTAINER]) + b"<illum-2bc container>", pair_identical=pi)
    ctx("q0NAME", 19, GZ_FQ_ITEM_TEXT, DTYPE_1, item=5, pair_identical=pi)
    ctx("q1NAME", 20, GZ_FQ_ITEM_TEXT, DTYPE_1, item=6, pair_identical=pi)
    ctx("q2NAME", 21, GZ_FQ_ITEM_TEXT, DTYPE_1, item=7, pair_identical=pi)
    ctx("SQBITMAP", 40, GZ_FQ_CONST, snip=bytes([SNIP_SPECIAL]) + b"<unaligned SEQ>", pair_assisted_b250=True)
    ctx("NONREF_X", 42, GZ_FQ_SEQ, local_dep=1)
    ctx("QUAL", 80, GZ_FQ_QUAL, lcodec=qual_codec)
    for k, tag in enumerate(("DOMQRUNS", "QUALMPLX", "DIVRQUAL")):     # "these 3 must be right after SAM_QUAL" (src/sam.h:108-110)
        ctx(tag, 81 + k, GZ_FQ_QUAL_AUX, item=k, local_dep=2)
    ctx("TOPLEVEL", 90, GZ_FQ_CONST, snip=bytes([SNIP_CONTAINER]) + b"<fastq toplevel>", pair_identical=pi)
    ctx("E1L", 96, GZ_FQ_CONST, snip=b"\n", pair_identical=pi)
    ctx("E2L", 97, GZ_FQ_CONST, snip=b"\n", pair_identical=pi)
    ctx("LINE3", 98, GZ_FQ_CONST, snip=b"", pair_identical=pi)        # replaced below: an empty line 3 is the snip ""
    P[-1]["snip"] = bytes([SNIP_SPECIAL]) + b"<line3 = empty>"
    return dict(ctxs=P, seps=b":::: :+", sep_counts=[3, 1, 1, 1, 1, 3, 1], paired=paired, estimated_entries=estimated_entries, qual_codec=domq, vb_size=vb_size)


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
        keep.append(c["snip"])
    p = GzFastqPlan()
    p.ctxs, p.n_ctxs = arr, n
    p.seps = plan["seps"]
    p.sep_counts = (C.c_uint8 * 16)(*(list(plan["sep_counts"]) + [0] * (16 - len(plan["sep_counts"]))))
    p.n_seps, p.paired, p.estimated_entries = len(plan["seps"]), int(plan["paired"]), plan["estimated_entries"]
    p.qual_codec = plan.get("qual_codec", 0)
    p.vb_size = plan.get("vb_size", 0)
    return p, keep
