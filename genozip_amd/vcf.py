"""Host-side set-up of the VBlock compute driver for multi-sample VCF text (BASELINE configs[3]; SURVEY 8f N1 for VCF), as DATA: a
GzFastqPlan with record_lines = 1 and n_samples > 0 - a record is one data line, its items the nine fixed tab-separated fields
(vcf_seg_txt_line, src/vcf_seg.c) and, for every sample, the ':'-separated FORMAT subfields (vcf_seg_samples, src/vcf_samples.c:1601).

    CHROM POS ID REF ALT QUAL FILTER INFO FORMAT  S1 ... Sn        with FORMAT = GT:DP:PL

How each field reaches its context (SURVEY 8(0), row configs[3], as far as the driver has the means):
  CHROM ID REF ALT FILTER FORMAT   snips -> dictionary + b250
  POS           delta against the previous line in a dyn-int local (the reference: a delta snip per line; same deltas)
  QUAL          seg_integer_or_not: dyn-int local
  INFO          one snip per line (the reference segs every INFO tag into a context of its own: not built); lines differ, so
                zip_handle_unique_words_ctxs (src/zip.c:136-166) hands the dictionary to local, as for any unique-ID field
  FORMAT/GT     one snip per sample per line -> dictionary + b250 of lines x samples entries (the reference turns GT into a haplotype matrix
                for CODEC_PBWT, src/vcf_format_GT.c / src/codec_pbwt.c: out of scope)
  FORMAT/DP     integers -> dyn-int local of lines x samples, TRANSPOSED to samples x lines (ctx->dyn_transposed, src/vcf_samples.c:121-130,
                917-919; dyn_int_transpose src/dyn_int.c:45-132): LT_UINT8_TR, param 0 = the file's number of samples
  FORMAT/PL     one snip per sample per line -> b250 of lines x samples entries (src/vcf_samples.c:1134-1156; the reference multiplexes the
                snips over two dictionaries by the sample's dosage: not built - one dictionary here)
The TOPLEVEL / samples containers are built in the reference's container FORMAT with this repo's own (simpler) choice of items - every
subfield a plain snip or integer. That is a valid encoding: the reference's own genounzip gives the VCF back byte for byte
(tests/test_e2e_genounzip.py::test_vcf_round_trip); byte-level parity of the sections: the CPU restatement's composition,
tests/parity.py::vcf_zip."""
from .fastq import (dict_id, container, DTYPE_FIELD, DTYPE_2, STORE_INT, SNIP_SELF_DELTA, CON_FILTER_REPEATS, CON_FILTER_ITEMS, CON_IS_TOPLEVEL, CON_CALLBACK, CON_DROP_FINAL_REPSEP)
from .lib import (GZ_FQ_CONST, GZ_FQ_ITEM_TEXT, GZ_FQ_ITEM_INT, GZ_FQ_ITEM_DELTA, GZ_FQ_TOPLEVEL)


def vcf_plan(n_samples, estimated_entries=0, vb_size=0):
    P = []

    def ctx(tag, did_i, kind, dtype=DTYPE_FIELD, item=0, flags=0, snip=b"", con_len=0, per_sample=0, transposed=0):
        P.append(dict(tag=tag, dict_id=dict_id(tag, dtype), did_i=did_i, kind=kind, item=item, flags=flags, snip=snip, pair_identical=False, no_stons=bool(per_sample),
                      lcodec=0, bcodec=0, pair_assisted_b250=False, local_dep=0, nothing_char=0, con_len=con_len, segs_per_line=0, per_sample=per_sample, transposed=transposed))

    fixed = [("CHROM", GZ_FQ_ITEM_TEXT), ("POS", GZ_FQ_ITEM_DELTA), ("ID", GZ_FQ_ITEM_TEXT), ("REF", GZ_FQ_ITEM_TEXT), ("ALT", GZ_FQ_ITEM_TEXT), ("QUAL", GZ_FQ_ITEM_INT),
             ("FILTER", GZ_FQ_ITEM_TEXT), ("INFO", GZ_FQ_ITEM_TEXT), ("FORMAT", GZ_FQ_ITEM_TEXT)]
    for i, (tag, kind) in enumerate(fixed):
        ctx(tag, i, kind, item=i, flags=STORE_INT if kind == GZ_FQ_ITEM_DELTA else 0, snip=(bytes([SNIP_SELF_DELTA]) + b"$") if kind == GZ_FQ_ITEM_DELTA else b"")
    ctx("GT", 20, GZ_FQ_ITEM_TEXT, DTYPE_2, item=0, per_sample=1)
    ctx("DP", 21, GZ_FQ_ITEM_INT, DTYPE_2, item=1, per_sample=1, transposed=1)
    ctx("PL", 22, GZ_FQ_ITEM_TEXT, DTYPE_2, item=2, per_sample=1)
    smp = container([(dict_id("GT", DTYPE_2), b":"), (dict_id("DP", DTYPE_2), b":"), (dict_id("PL", DTYPE_2), b"")], repeats=n_samples, repsep=b"\t\0", flags=CON_DROP_FINAL_REPSEP)   # (vcf_samples.c:1267)
    ctx("SAMPLES", 30, GZ_FQ_CONST, snip=b"\x04" + __import__("base64").b64encode(smp))
    top = container([(dict_id(t), b"\t") for t, _ in fixed] + [(dict_id("SAMPLES"), b""), (dict_id("EOL"), b"")],
                    flags=CON_FILTER_REPEATS | CON_FILTER_ITEMS | CON_IS_TOPLEVEL | CON_CALLBACK)
    ctx("TOPLEVEL", 40, GZ_FQ_TOPLEVEL, snip=top, con_len=len(top))
    ctx("EOL", 41, GZ_FQ_CONST, snip=b"\n")
    return dict(ctxs=P, seps=b"\t" * 9, sep_counts=[1] * 9, paired=False, estimated_entries=estimated_entries, qual_codec=0, vb_size=vb_size, line3_empty=0, vb_1_not_representative=0b110,
                record_lines=1, seq_item=0, qual_item=0, n_samples=n_samples, n_subfields=3)
