/* gz_oracle_path.c -- TEST / MEASUREMENT INFRASTRUCTURE, not shipped, never linked into the product.
 *
 * The WHOLE PATH of one FASTQ VBlock on ONE CPU core: zip_compress_one_vb (src/zip.c:510-601) for the Illumina plan of
 * genozip_amd/fastq.py, composed from this directory's one-at-a-time restatements (gz_oracle.c) in the order the reference runs them:
 *   text -> lines (seg_get_next_line) -> reads (fastq_seg_get_lines) -> line-1 items (qname_seg_qf's separators)
 *        -> per item: seg_by_ctx / seg_integer_or_not / seg_self_delta (hash lookups, dictionaries, seg-format b250, dyn-int locals)
 *        -> SEQ -> NONREF.local -> CODEC_ACGT's 2-bit pack (+ NONREF_X); QUAL -> QUAL.local, or CODEC_DOMQ's four streams
 *        -> ctx_merge_in_one_vctx of every context into a file-level context -> b250_zip_generate, zip_generate_local
 *        -> comp_compress of every section (codec call, adler32, header) -> VB header
 * and a pthread pool that runs one VBlock per task - what the reference's dispatcher does with its compute threads
 * (src/dispatcher.c:544-618). This is bench.py's `cpu_baseline.whole_path`: the SAME WORK as the GPU step, timed on the host's cores.
 *
 * What it is not: a byte-for-byte twin of the product's z_data. Every VBlock merges into file-level contexts OF ITS OWN (as if it were
 * its file's first: no cloned dictionary, no mutex between VBlocks - which favours the CPU: the reference serialises the merge per
 * context, src/context.c:944-946), constant-snip contexts (containers, EOLs: nothing per line) and the pair rules are left out, and the
 * codecs are given by the caller (the ones the GPU run's file ended up with) instead of being found by trial - a VBlock whose plan
 * says 0 for a stream runs the nine-candidate trial of codec_assign_best_codec on it (what the file's first VBlock pays).
 * LZMA of the packed SEQ is outside the path on both sides.
 * The codec calls go through gzo_codec_compress, or - when the caller has handed in the reference's own htscodecs entry points
 * (oracle/_ref/libhtsref.so) - through those. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <time.h>
#include <malloc.h>
#include "gz_oracle.h"

typedef long (*GzoHtsFn) (const unsigned char *in, unsigned in_size, unsigned char *out, unsigned out_cap, int order);
static GzoHtsFn path_rans = NULL, path_arith = NULL;
void gzo_path_use_codecs (void *rans_fn, void *arith_fn) { path_rans = (GzoHtsFn)rans_fn; path_arith = (GzoHtsFn)arith_fn; }

typedef struct {
    uint32_t n_items;              /* line-1 items (n_seps + 1) */
    uint8_t  item_kind[16];        /* 0: seg_by_ctx (text), 1: seg_integer_or_not, 2: seg_self_delta */
    uint8_t  seps[32], sep_counts[32]; uint32_t n_seps;   /* item i ends at the sep_counts[i]-th seps[i] (CI0_COLONn) */
    uint8_t  lcodec[16], bcodec[16];   /* per item; 0: codec_assign_best_codec on this VBlock's stream */
    uint8_t  qual_codec, aux_codec[3], x_codec, qual_bcodec;
    uint8_t  domq;                 /* QUAL through CODEC_DOMQ */
} GzoPathPlan;

static int order_of (int codec) { static const int o[4] = { 0x01, 0x19, 0x81, 0x99 }; return o[(codec - (codec < 16 ? 6 : 16)) & 3]; }

/* comp_compress of one section (compressor.c:18-175): < 50 bytes stored, else the codec; adler32; 40-byte header */
static long path_section (uint8_t sec_type, int codec, uint8_t hdr_codec, uint8_t ltype, uint32_t vblock_i, const uint8_t *data, uint32_t n, uint8_t *z, uint64_t z_cap, uint64_t *streams)
{
    GzoCtxSectionDesc d; memset (&d, 0, sizeof (d));
    d.vblock_i = vblock_i; d.section_type = sec_type; d.ltype = ltype; d.b250_size_or_nothing_char = sec_type == 11 ? 4 : 0;
    if (!codec) codec = n >= 50 ? gzo_codec_assign_best (data, n, NULL) : GZO_CODEC_RANB;       /* codec.c:309-312, zfile.c:300,337 */
    if (!codec) codec = GZO_CODEC_RANB;
    *streams += n;
    d.codec = hdr_codec ? hdr_codec : (uint8_t)codec; d.sub_codec = hdr_codec ? (uint8_t)codec : 0;
    if (!hdr_codec && n < 50) return gzo_section_compress (&d, data, n, z, z_cap);            /* stored (compressor.c:56-58) */
    /* (a complex codec's stream is coded by its sub-codec whatever its length, codec_domq.c:508-520) */
    const uint32_t cap = gzo_codec_est_size (codec, n);
    uint32_t l = cap;
    uint8_t *pay = malloc ((size_t)cap + 64);
    if (!pay) return -1;
    long r = -1;
    if (codec == GZO_CODEC_NONE) { memcpy (pay, data, n); r = gzo_section_frame (&d, pay, n, n, z, z_cap); }
    else if (path_rans) { const long pl = (codec < 16 ? path_rans : path_arith) (data, n, pay, cap, order_of (codec)); if (pl >= 0) r = gzo_section_frame (&d, pay, (uint32_t)pl, n, z, z_cap); }
    else if (gzo_codec_compress (codec, data, n, pay, &l, 0) == 1) r = gzo_section_frame (&d, pay, l, n, z, z_cap);
    free (pay);
    return r;
}

/* one context: merge into a file-level context of its own, generate the b250, write it */
static long path_b250 (const GzoColumn *col, uint32_t n, uint64_t local_len, int bcodec, uint32_t vblock_i, uint8_t *z, uint64_t z_cap, uint64_t *streams)
{
    GzoZctx *Z = gzo_zctx_create (0);
    if (!Z) return -1;
    GzoMerge m; memset (&m, 0, sizeof (m));
    int32_t *n2w = malloc (((size_t)col->n_new + 1) * 4); uint8_t *ston = malloc ((size_t)col->dict_len + 16);
    long r = -1;
    if (n2w && ston) {
        m.vblock_i = vblock_i; m.n_new = col->n_new; m.dict = col->dict; m.node_char_index = col->node_char_index; m.node_snip_len = col->node_snip_len; m.counts = col->counts;
        m.can_have_singletons = !local_len && !col->all_the_same; m.flags = col->all_the_same ? 0x20 : 0; m.b250_len = col->b250_len; m.local_len = local_len;
        m.ats_node_index = col->all_the_same && n ? col->node_index[0] : -1; m.node2word = n2w; m.ston_local = ston;
        if (gzo_ctx_merge (Z, &m) == 0) {
            r = 0;
            if (!m.dropped_b250 && col->b250_len) {
                uint8_t *out = malloc ((size_t)col->b250_len + 16);
                const long l = out ? gzo_b250_generate (col->b250, (uint32_t)col->b250_len, 0, n2w, col->n_new, out) : -1;
                r = l < 0 ? -1 : path_section (11, bcodec, 0, 0, vblock_i, out, (uint32_t)l, z, z_cap, streams);
                free (out);
            }
            if (r >= 0 && m.ston_len) { const long r2 = path_section (12, 0, 0, 0, vblock_i, ston, (uint32_t)m.ston_len, z + r, z_cap - (uint64_t)r, streams); r = r2 < 0 ? -1 : r + r2; }
        }
    }
    free (n2w); free (ston); gzo_zctx_destroy (Z);
    return r;
}

static int col_alloc (GzoColumn *c, uint64_t n, uint64_t bytes)
{
    memset (c, 0, sizeof (*c));
    c->node_index = malloc ((n + 1) * 4); c->dict = malloc (bytes + n + 16); c->node_char_index = malloc ((n + 1) * 8); c->node_snip_len = malloc ((n + 1) * 4);
    c->counts = malloc ((n + 1) * 4); c->b250 = malloc (4 * n + 16);
    return c->node_index && c->dict && c->node_char_index && c->node_snip_len && c->counts && c->b250;
}
static void col_free (GzoColumn *c) { free (c->node_index); free (c->dict); free (c->node_char_index); free (c->node_snip_len); free (c->counts); free (c->b250); }

/* -> bytes of z_data, or -1. *streams: bytes that entered the codecs */
long gzo_fastq_vb_path (const uint8_t *text, uint64_t text_len, uint32_t vblock_i, const GzoPathPlan *P, uint8_t *z, uint64_t z_cap, uint64_t *streams)
{
    long rc = -1; uint64_t zl = GZO_VB_HEADER_LEN; *streams = 0;
    if (z_cap < GZO_VB_HEADER_LEN || text_len > 0xfffffff0ull) return -1;
    uint64_t cap = text_len / 16 + 1024;
    uint32_t *lo = malloc (cap * 4), *ll = malloc (cap * 4);
    if (!lo || !ll) { free (lo); free (ll); return -1; }
    uint64_t nl = gzo_text_lines (text, text_len, lo, ll, cap);
    if (nl > cap) { free (lo); free (ll); cap = nl + 8; lo = malloc (cap * 4); ll = malloc (cap * 4); if (!lo || !ll) { free (lo); free (ll); return -1; } nl = gzo_text_lines (text, text_len, lo, ll, cap); }
    const uint64_t n = nl / 4;
    uint32_t *rec = malloc ((8 * n + 8) * 4);
    uint32_t n_flat = 0; uint8_t flat[64];
    for (uint32_t i = 0; i < P->n_seps; i++) for (uint32_t k = 0; k < P->sep_counts[i] && n_flat < 63; k++) flat[n_flat++] = P->seps[i];
    uint32_t *fo = malloc (((uint64_t)(n_flat + 1) * n + 8) * 4), *fl = malloc (((uint64_t)(n_flat + 1) * n + 8) * 4);
    uint32_t *io = malloc ((n + 8) * 4), *il = malloc ((n + 8) * 4), *so = malloc ((n + 8) * 4), *sl = malloc ((n + 8) * 4);
    int64_t *vals = malloc ((n + 8) * 8); uint8_t *isn = malloc (n + 8), *ltext = NULL, *blob = NULL, *packed = NULL, *dyn = malloc ((n + 8) * 8);
    GzoColumn col; memset (&col, 0, sizeof (col));
    if (!rec || !fo || !fl || !io || !il || !so || !sl || !vals || !isn || !dyn) goto done;
    uint32_t *l1o = rec, *l1l = rec + n, *sqo = rec + 2 * n, *sql = rec + 3 * n, *l3o = rec + 4 * n, *l3l = rec + 5 * n, *qo = rec + 6 * n, *ql = rec + 7 * n;
    if (gzo_fastq_records (text, lo, ll, nl, l1o, l1l, sqo, sql, l3o, l3l, qo, ql) != 0) goto done;
    if (gzo_tokenize_column (text, l1o, l1l, n, flat, n_flat, fo, fl) != 0) goto done;
    /* SNIP_LOOKUP lives behind the text for seg_integer_or_not's column: a copy of the text with one more byte would double the traffic;
       the snip column only needs SOME address that holds the byte - a one-byte text of its own */
    ltext = malloc (16); if (!ltext) goto done; ltext[0] = 1;
    uint32_t at_flat = 0;
    for (uint32_t it = 0; it < P->n_items; it++) {
        const uint32_t k = it < P->n_seps ? P->sep_counts[it] : 1;
        for (uint64_t r = 0; r < n; r++) { io[r] = fo[(uint64_t)at_flat * n + r]; il[r] = fo[(uint64_t)(at_flat + k - 1) * n + r] + fl[(uint64_t)(at_flat + k - 1) * n + r] - io[r]; }
        at_flat += k;
        uint64_t bytes = 0; for (uint64_t r = 0; r < n; r++) bytes += il[r];
        uint64_t local_len = 0;
        if (P->item_kind[it] == 2) {                                   /* seg_self_delta (qname.c:750-759): the delta against the previous line */
            int64_t prev = 0;
            for (uint64_t r = 0; r < n; r++) { int64_t v = 0; for (uint32_t c = 0; c < il[r]; c++) v = v * 10 + (text[io[r] + c] - '0'); vals[r] = v - prev; prev = v; }
            const int lt = gzo_dyn_int_column (vals, NULL, n, 0, dyn);
            const uint32_t w = gzo_lt_width (lt);
            gzo_local_generate (lt, dyn, n, 0, NULL);
            const long r1 = n ? path_section (12, P->lcodec[it], 0, (uint8_t)lt, vblock_i, dyn, (uint32_t)(n * w), z + zl, z_cap - zl, streams) : 0;
            if (r1 < 0) goto done;
            zl += (uint64_t)r1;
            continue;                                                  /* (its b250: one constant snip) */
        }
        const uint8_t *ctext = text; const uint32_t *co = io, *cl = il;
        if (P->item_kind[it] == 1) {                                   /* seg_integer_or_not (qname.c:773-775) */
            /* the lookup byte: gzo_seg_integer_or_not points integer snips at lookup_off IN text; here the byte behind the text's last line end is
               not ours to write, so integers are pointed at offset 0 of a text that holds SNIP_LOOKUP there, and the others are not expected */
            const uint64_t nv = gzo_seg_integer_or_not (text, io, il, n, 0, 0xfffffff0u, so, sl, vals, isn);
            int all_int = nv == n;
            if (!all_int) goto done;                                   /* (the synthetic files' lane fields are integers) */
            for (uint64_t r = 0; r < n; r++) { so[r] = 0; sl[r] = 1; }
            ctext = ltext; co = so; cl = sl; bytes = n;
            const int lt = gzo_dyn_int_column (vals, isn, nv, 0, dyn);
            const uint32_t w = gzo_lt_width (lt);
            gzo_local_generate (lt, dyn, nv, 0, NULL);
            local_len = nv * w;
            const long r1 = nv ? path_section (12, P->lcodec[it], 0, (uint8_t)lt, vblock_i, dyn, (uint32_t)local_len, z + zl, z_cap - zl, streams) : 0;
            if (r1 < 0) goto done;
            zl += (uint64_t)r1;
        }
        if (!col_alloc (&col, n, bytes)) goto done;
        if (gzo_ctx_seg_column (ctext, co, cl, n, NULL, NULL, NULL, 0, &col) != 0) goto done;
        const long r2 = path_b250 (&col, (uint32_t)n, local_len, P->bcodec[it], vblock_i, z + zl, z_cap - zl, streams);
        col_free (&col); memset (&col, 0, sizeof (col));
        if (r2 < 0) goto done;
        zl += (uint64_t)r2;
    }
    /* SEQ: NONREF.local -> 2 bits per base (+ NONREF_X when a base is not ACGT); the packed bytes leave the path for the host's LZMA */
    {
        uint64_t nb = 0; for (uint64_t r = 0; r < n; r++) nb += sql[r];
        blob = malloc (nb + n + 64); packed = malloc (gzo_acgt_packed_len (nb) + 64);
        if (!blob || !packed) goto done;
        const uint64_t L = gzo_local_blob_column (text, sqo, sql, n, 0, blob);
        if (gzo_acgt_pack (blob, L, packed, blob)) {
            const long r3 = path_section (12, P->x_codec, 11 /* CODEC_XCGT */, 27, vblock_i, blob, (uint32_t)L, z + zl, z_cap - zl, streams);
            if (r3 < 0) goto done;
            zl += (uint64_t)r3;
        }
        free (blob); blob = NULL;
    }
    /* QUAL (fastq_seg_QUAL, fastq_qual.c:24-47): a line of one repeated score segs a special snip and stays out of the local / of CODEC_DOMQ's
       streams, every other line segs SNIP_LOOKUP; the context's b250 like any other column */
    {
        uint8_t *slots = malloc (4 * n + 16);
        if (!slots) goto done;
        for (uint64_t r = 0; r < n; r++) {
            const uint8_t *q = text + qo[r]; const uint8_t c = q[0];
            uint32_t i = 1; while (i < ql[r] && q[i] == c) i++;
            if (i >= ql[r]) { slots[4 * r] = 8; slots[4 * r + 1] = 41; slots[4 * r + 2] = c; so[r] = (uint32_t)(4 * r); sl[r] = 3; ql[r] = 0; }
            else { slots[4 * r] = 1; so[r] = (uint32_t)(4 * r); sl[r] = 1; }
        }
        uint64_t nq = 0; for (uint64_t r = 0; r < n; r++) nq += ql[r];
        long r6 = -1;
        if (col_alloc (&col, n, 3 * n) && gzo_ctx_seg_column (slots, so, sl, n, NULL, NULL, NULL, 0, &col) == 0)
            r6 = path_b250 (&col, (uint32_t)n, nq, P->qual_bcodec, vblock_i, z + zl, z_cap - zl, streams);
        col_free (&col); memset (&col, 0, sizeof (col)); free (slots);
        if (r6 < 0) goto done;
        zl += (uint64_t)r6;
    }
    if (P->domq) {
        GzoDomq dq; memset (&dq, 0, sizeof (dq));
        if (gzo_domq_encode (text, qo, ql, n, &dq) != 0) goto done;
        const uint8_t *s[4] = { dq.qual, dq.runs, dq.mplx, dq.divr }; const uint64_t sn[4] = { dq.qual_len, dq.runs_len, dq.mplx_len, dq.divr_len };
        int ok = 1;
        for (int k = 0; k < 4 && ok; k++) if (sn[k]) {
            const long r4 = k ? path_section (12, P->aux_codec[k - 1], 0, 27, vblock_i, s[k], (uint32_t)sn[k], z + zl, z_cap - zl, streams)
                              : path_section (12, P->qual_codec, 13 /* CODEC_DOMQ */, 13, vblock_i, s[k], (uint32_t)sn[k], z + zl, z_cap - zl, streams);
            if (r4 < 0) ok = 0; else zl += (uint64_t)r4;
        }
        gzo_domq_free (&dq);
        if (!ok) goto done;
    }
    else {
        uint64_t nq = 0; for (uint64_t r = 0; r < n; r++) nq += ql[r];
        blob = malloc (nq + 64);
        if (!blob) goto done;
        const uint64_t L = gzo_local_blob_column (text, qo, ql, n, 0, blob);
        const long r5 = L ? path_section (12, P->qual_codec, 0, GZO_LT_BLOB, vblock_i, blob, (uint32_t)L, z + zl, z_cap - zl, streams) : 0;
        if (r5 < 0) goto done;
        zl += (uint64_t)r5;
    }
    gzo_vb_header_write (z, vblock_i, (uint32_t)text_len, 0, 0, NULL, 0);
    gzo_vb_header_patch (z, (uint32_t)zl);
    rc = (long)zl;
done:
    col_free (&col);
    free (lo); free (ll); free (rec); free (fo); free (fl); free (io); free (il); free (so); free (sl); free (vals); free (isn); free (dyn); free (ltext); free (blob); free (packed);
    return rc;
}

/* ---- the same for the one-line-record plans of genozip_amd/sam.py (BASELINE configs[2]: sam_seg_txt_line / bam_seg_txt_line's chain, src/sam_seg.c,
 * src/bam_seg.c:425-520) and genozip_amd/vcf.py (configs[3]: vcf_seg_txt_line + vcf_seg_samples, src/vcf_samples.c:1601): a record is one line, its
 * items come from the plan's separators; an item is a snip column, an integer column (seg_integer_or_not: dyn-int local + SNIP_LOOKUP), a delta against
 * the previous line, SEQ (bases to NONREF.local, padded per read, 2-bit packed) or QUAL (QUAL.local or CODEC_DOMQ's streams); a VCF line's last item
 * holds n_samples samples of n_sub ':'-separated subfields, each a snip column of lines x samples entries or an integer matrix written transposed
 * (dyn_int_transpose, src/dyn_int.c:45-132). Same rules as gzo_fastq_vb_path: contexts of the VBlock's own, codecs given by the caller. ---- */
typedef struct {
    uint32_t n_items;              /* items of a line: n_seps + 1 */
    uint8_t  item_kind[64];        /* 0 snips, 1 seg_integer_or_not, 2 delta against the previous line, 3 SEQ, 4 QUAL, 5 not a context (a tag name the plan expects), 6 the samples */
    uint8_t  seps[64], sep_counts[64]; uint32_t n_seps;
    uint8_t  lcodec[64], bcodec[64];
    uint8_t  qual_codec, aux_codec[3], x_codec, domq, seq_pad, reserved;
    uint32_t n_samples, n_sub;
    uint8_t  sub_kind[8], sub_lcodec[8], sub_bcodec[8];      /* per FORMAT subfield: 0 snips, 1 integers in a matrix written transposed */
} GzoTextPlan;

static long path_int_column (const uint8_t *text, const uint32_t *io, const uint32_t *il, uint64_t n, uint32_t transpose_cols, int lcodec, int bcodec, uint32_t vblock_i,
                             uint8_t *z, uint64_t z_cap, uint64_t *streams)
{
    /* seg_integer_or_not: the numbers into a dyn-int local (written transposed for a samples matrix), SNIP_LOOKUP into the b250; other snips stay snips */
    uint32_t *so = malloc ((n + 8) * 4), *sl = malloc ((n + 8) * 4); int64_t *vals = malloc ((n + 8) * 8); uint8_t *isn = malloc (n + 8), *dyn = malloc ((n + 8) * 8), *scr = malloc ((n + 8) * 8);
    uint8_t *ctext = NULL;
    GzoColumn col; memset (&col, 0, sizeof (col));
    long rc = -1, zl = 0;
    if (!so || !sl || !vals || !isn || !dyn || !scr) goto out;
    uint64_t tlen = 0; for (uint64_t r = 0; r < n; r++) if (io[r] + il[r] > tlen) tlen = (uint64_t)io[r] + il[r];
    /* (the lookup byte must live in the text the column points into: a private copy with one more byte, as the caller of seg_integer_or_not has it) */
    ctext = malloc (tlen + 16); if (!ctext) goto out;
    memcpy (ctext, text, tlen); ctext[tlen] = 1;
    const uint64_t nv = gzo_seg_integer_or_not (ctext, io, il, n, 0, (uint32_t)tlen, so, sl, vals, isn);
    if (nv) {
        const int lt = gzo_dyn_int_column (vals, isn, nv, 0, dyn);
        const uint32_t w = gzo_lt_width (lt);
        gzo_local_generate (lt, dyn, nv, transpose_cols && nv == n ? transpose_cols : 0, scr);
        const long r1 = path_section (12, lcodec, 0, (uint8_t)lt, vblock_i, dyn, (uint32_t)(nv * w), z, z_cap, streams);
        if (r1 < 0) goto out;
        zl = r1;
    }
    uint64_t bytes = 0; for (uint64_t r = 0; r < n; r++) bytes += sl[r];
    if (!col_alloc (&col, n, bytes) || gzo_ctx_seg_column (ctext, so, sl, n, NULL, NULL, NULL, 0, &col) != 0) goto out;
    const long r2 = path_b250 (&col, (uint32_t)n, nv ? nv : 0, bcodec, vblock_i, z + zl, z_cap - (uint64_t)zl, streams);
    if (r2 < 0) goto out;
    rc = zl + r2;
out:
    col_free (&col); free (so); free (sl); free (vals); free (isn); free (dyn); free (scr); free (ctext);
    return rc;
}

long gzo_text_vb_path (const uint8_t *text, uint64_t text_len, uint32_t vblock_i, const GzoTextPlan *P, uint8_t *z, uint64_t z_cap, uint64_t *streams)
{
    long rc = -1; uint64_t zl = GZO_VB_HEADER_LEN; *streams = 0;
    if (z_cap < GZO_VB_HEADER_LEN || text_len > 0xfffffff0ull) return -1;
    uint64_t cap = text_len / 32 + 1024;
    uint32_t *lo = malloc (cap * 4), *ll = malloc (cap * 4);
    if (!lo || !ll) { free (lo); free (ll); return -1; }
    uint64_t n = gzo_text_lines (text, text_len, lo, ll, cap);
    if (n > cap) { free (lo); free (ll); cap = n + 8; lo = malloc (cap * 4); ll = malloc (cap * 4); if (!lo || !ll) { free (lo); free (ll); return -1; } n = gzo_text_lines (text, text_len, lo, ll, cap); }
    uint32_t n_flat = 0; uint8_t flat[256];
    for (uint32_t i = 0; i < P->n_seps; i++) for (uint32_t k = 0; k < P->sep_counts[i] && n_flat < 255; k++) flat[n_flat++] = P->seps[i];
    uint32_t *fo = malloc (((uint64_t)(n_flat + 1) * n + 8) * 4), *fl = malloc (((uint64_t)(n_flat + 1) * n + 8) * 4);
    uint32_t *io = malloc ((n + 8) * 4), *il = malloc ((n + 8) * 4);
    int64_t *vals = malloc ((n + 8) * 8); uint8_t *dyn = malloc ((n + 8) * 8), *blob = NULL, *packed = NULL;
    uint32_t *so = NULL, *sl = NULL;
    GzoColumn col; memset (&col, 0, sizeof (col));
    if (!fo || !fl || !io || !il || !vals || !dyn) goto done;
    if (gzo_tokenize_column (text, lo, ll, n, flat, n_flat, fo, fl) != 0) goto done;
    uint32_t at_flat = 0;
    for (uint32_t it = 0; it < P->n_items; it++) {
        const uint32_t k = it < P->n_seps ? P->sep_counts[it] : 1;
        for (uint64_t r = 0; r < n; r++) { io[r] = fo[(uint64_t)at_flat * n + r]; il[r] = fo[(uint64_t)(at_flat + k - 1) * n + r] + fl[(uint64_t)(at_flat + k - 1) * n + r] - io[r]; }
        at_flat += k;
        const int kind = P->item_kind[it];
        long r1 = 0;
        if (kind == 5) continue;
        if (kind == 0) {
            uint64_t bytes = 0; for (uint64_t r = 0; r < n; r++) bytes += il[r];
            if (!col_alloc (&col, n, bytes) || gzo_ctx_seg_column (text, io, il, n, NULL, NULL, NULL, 0, &col) != 0) goto done;
            r1 = path_b250 (&col, (uint32_t)n, 0, P->bcodec[it], vblock_i, z + zl, z_cap - zl, streams);
            col_free (&col); memset (&col, 0, sizeof (col));
        }
        else if (kind == 1) r1 = path_int_column (text, io, il, n, 0, P->lcodec[it], P->bcodec[it], vblock_i, z + zl, z_cap - zl, streams);
        else if (kind == 2) {
            int64_t prev = 0;
            for (uint64_t r = 0; r < n; r++) { int64_t v = 0; for (uint32_t c = 0; c < il[r]; c++) v = v * 10 + (text[io[r] + c] - '0'); vals[r] = v - prev; prev = v; }
            const int lt = gzo_dyn_int_column (vals, NULL, n, 0, dyn);
            gzo_local_generate (lt, dyn, n, 0, NULL);
            r1 = n ? path_section (12, P->lcodec[it], 0, (uint8_t)lt, vblock_i, dyn, (uint32_t)(n * gzo_lt_width (lt)), z + zl, z_cap - zl, streams) : 0;
        }
        else if (kind == 3) {                                          /* SEQ: the bases, every read padded with 'A' to a multiple of seq_pad (sam_seg_SEQ_pad_nonref, sam_seq.c:224-229) */
            uint64_t nb = 0; for (uint64_t r = 0; r < n; r++) nb += il[r] + P->seq_pad;
            blob = malloc (nb + 64); packed = malloc (gzo_acgt_packed_len (nb) + 64);
            if (!blob || !packed) goto done;
            const uint64_t L = gzo_local_blob_column_ex (text, io, il, n, 0, NULL, 0, P->seq_pad, 'A', blob, NULL);
            if (gzo_acgt_pack (blob, L, packed, blob)) r1 = path_section (12, P->x_codec, 11 /* CODEC_XCGT */, 27, vblock_i, blob, (uint32_t)L, z + zl, z_cap - zl, streams);
            free (blob); blob = NULL; free (packed); packed = NULL;
        }
        else if (kind == 4) {                                          /* QUAL */
            if (P->domq) {
                GzoDomq dq; memset (&dq, 0, sizeof (dq));
                if (gzo_domq_encode (text, io, il, n, &dq) != 0) goto done;
                const uint8_t *s[4] = { dq.qual, dq.runs, dq.mplx, dq.divr }; const uint64_t sn[4] = { dq.qual_len, dq.runs_len, dq.mplx_len, dq.divr_len };
                for (int q = 0; q < 4 && r1 >= 0; q++) if (sn[q]) {
                    const long r4 = q ? path_section (12, P->aux_codec[q - 1], 0, 27, vblock_i, s[q], (uint32_t)sn[q], z + zl + r1, z_cap - zl - (uint64_t)r1, streams)
                                      : path_section (12, P->qual_codec, 13 /* CODEC_DOMQ */, 13, vblock_i, s[q], (uint32_t)sn[q], z + zl + r1, z_cap - zl - (uint64_t)r1, streams);
                    r1 = r4 < 0 ? -1 : r1 + r4;
                }
                gzo_domq_free (&dq);
            }
            else {
                uint64_t nq = 0; for (uint64_t r = 0; r < n; r++) nq += il[r];
                blob = malloc (nq + 64);
                if (!blob) goto done;
                const uint64_t L = gzo_local_blob_column (text, io, il, n, 0, blob);
                r1 = L ? path_section (12, P->qual_codec, 0, GZO_LT_BLOB, vblock_i, blob, (uint32_t)L, z + zl, z_cap - zl, streams) : 0;
                free (blob); blob = NULL;
            }
        }
        else if (kind == 6 && P->n_samples && P->n_sub) {              /* vcf_seg_samples: samples by tab, subfields by ':' (trailing ones may be left out: empty) */
            const uint64_t ns = P->n_samples, k2 = n * ns;
            so = malloc (((uint64_t)P->n_sub * k2 + 8) * 4); sl = malloc (((uint64_t)P->n_sub * k2 + 8) * 4);
            if (!so || !sl) goto done;
            memset (sl, 0, ((uint64_t)P->n_sub * k2 + 8) * 4);
            for (uint64_t r = 0; r < n; r++) {
                uint32_t a = io[r]; const uint32_t e = io[r] + il[r];
                for (uint64_t s_ = 0; s_ < ns && a <= e; s_++) {
                    uint32_t b = a; while (b < e && text[b] != '\t') b++;
                    uint32_t c0 = a;
                    for (uint32_t j = 0; j < P->n_sub && c0 <= b; j++) {
                        uint32_t c1 = c0; while (c1 < b && text[c1] != ':') c1++;
                        so[(uint64_t)j * k2 + r * ns + s_] = c0; sl[(uint64_t)j * k2 + r * ns + s_] = c1 - c0;
                        c0 = c1 + 1;
                    }
                    a = b + 1;
                }
            }
            for (uint32_t j = 0; j < P->n_sub && r1 >= 0; j++) {
                long rj;
                if (P->sub_kind[j] == 1) rj = path_int_column (text, so + (uint64_t)j * k2, sl + (uint64_t)j * k2, k2, (uint32_t)ns, P->sub_lcodec[j], P->sub_bcodec[j], vblock_i, z + zl + r1, z_cap - zl - (uint64_t)r1, streams);
                else {
                    uint64_t bytes = 0; for (uint64_t q = 0; q < k2; q++) bytes += sl[(uint64_t)j * k2 + q];
                    rj = -1;
                    if (col_alloc (&col, k2, bytes) && gzo_ctx_seg_column (text, so + (uint64_t)j * k2, sl + (uint64_t)j * k2, k2, NULL, NULL, NULL, 0, &col) == 0)
                        rj = path_b250 (&col, (uint32_t)k2, 1 /* no_stons: a per-sample context keeps its words */, P->sub_bcodec[j], vblock_i, z + zl + r1, z_cap - zl - (uint64_t)r1, streams);
                    col_free (&col); memset (&col, 0, sizeof (col));
                }
                r1 = rj < 0 ? -1 : r1 + rj;
            }
            free (so); free (sl); so = sl = NULL;
        }
        if (r1 < 0) goto done;
        zl += (uint64_t)r1;
    }
    gzo_vb_header_write (z, vblock_i, (uint32_t)text_len, 0, 0, NULL, 0);
    gzo_vb_header_patch (z, (uint32_t)zl);
    rc = (long)zl;
done:
    col_free (&col);
    free (lo); free (ll); free (fo); free (fl); free (io); free (il); free (vals); free (dyn); free (blob); free (packed); free (so); free (sl);
    return rc;
}

/* ---- one VBlock per task on a pthread pool (src/dispatcher.c:544-618) ---- */
typedef struct {
    const uint8_t *text; const uint64_t *off, *len; int n; int replicas; const GzoPathPlan *plan; uint64_t z_cap;
    long *z_len; uint64_t *streams; int next, failed; pthread_mutex_t mu; const GzoTextPlan *tplan;
} PathJob;

static void *path_worker (void *arg)
{
    PathJob *j = arg;
    uint8_t *z = malloc (j->z_cap);
    if (!z) { j->failed = 1; return NULL; }
    for (;;) {
        pthread_mutex_lock (&j->mu);
        const int t = j->next++;
        pthread_mutex_unlock (&j->mu);
        if (t >= j->n * j->replicas) break;
        const int i = t % j->n;
        uint64_t st = 0;
        const long l = j->tplan ? gzo_text_vb_path (j->text + j->off[i], j->len[i], (uint32_t)t + 1, j->tplan, z, j->z_cap, &st)
                                : gzo_fastq_vb_path (j->text + j->off[i], j->len[i], (uint32_t)t + 1, j->plan, z, j->z_cap, &st);
        if (l < 0) j->failed = 1;
        if (t < j->n) { j->z_len[i] = l; j->streams[i] = st; }
    }
    free (z);
    return NULL;
}

/* every VBlock `replicas` times (tasks >= 4 x threads keep every thread busy to the end); returns seconds, < 0 on failure */
static double path_many (const uint8_t *text, const uint64_t *off, const uint64_t *len, int n, int replicas, const GzoPathPlan *plan, const GzoTextPlan *tplan,
                         long *z_len, uint64_t *streams, int n_threads);
double gzo_fastq_path_many (const uint8_t *text, const uint64_t *off, const uint64_t *len, int n, int replicas, const GzoPathPlan *plan,
                            long *z_len, uint64_t *streams, int n_threads)
{
    return path_many (text, off, len, n, replicas, plan, NULL, z_len, streams, n_threads);
}
/* the one-line-record plans (SAM / VCF) */
double gzo_text_path_many (const uint8_t *text, const uint64_t *off, const uint64_t *len, int n, int replicas, const GzoTextPlan *plan,
                           long *z_len, uint64_t *streams, int n_threads)
{
    return path_many (text, off, len, n, replicas, NULL, plan, z_len, streams, n_threads);
}
static double path_many (const uint8_t *text, const uint64_t *off, const uint64_t *len, int n, int replicas, const GzoPathPlan *plan, const GzoTextPlan *tplan,
                         long *z_len, uint64_t *streams, int n_threads)
{
    uint64_t longest = 0;
    for (int i = 0; i < n; i++) if (len[i] > longest) longest = len[i];
    /* (a compute thread of the reference recycles its VBlock's buffers, src/vblock.c:253: keep freed blocks in the threads' heaps instead of
        handing 100 MB per VBlock back to the kernel and faulting it in again - with 256 threads that is what would be measured otherwise) */
    mallopt (M_MMAP_THRESHOLD, 1 << 30); mallopt (M_TRIM_THRESHOLD, 1 << 30); mallopt (M_TOP_PAD, 64 << 20);
    PathJob j = { text, off, len, n, replicas < 1 ? 1 : replicas, plan, longest + longest / 8 + (1 << 20), z_len, streams, 0, 0, PTHREAD_MUTEX_INITIALIZER, tplan };
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    pthread_t th[1024];
    struct timespec t0, t1;
    clock_gettime (CLOCK_MONOTONIC, &t0);
    for (int t = 1; t < n_threads; t++) pthread_create (&th[t], NULL, path_worker, &j);
    path_worker (&j);
    for (int t = 1; t < n_threads; t++) pthread_join (th[t], NULL);
    clock_gettime (CLOCK_MONOTONIC, &t1);
    return j.failed ? -1.0 : (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
