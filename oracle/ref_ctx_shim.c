/* oracle/ref_ctx_shim.c -- TEST INFRASTRUCTURE ONLY (see oracle/Makefile, target "ref").
 *
 * Compiled together with the reference's own src/b250.c and src/dyn_int.c WHERE THEY LIE under /root/reference (never
 * copied) into oracle/_ref/libctxref.so, against the reference's own headers (-iquote): what b250_seg_append,
 * b250_zip_generate, dyn_int_append and dyn_int_transpose (rows a3, a7) do to a Context is then the
 * reference's own code running, and tests/golden/ctx_golden.json is generated from it (tests/golden/make_ctx_golden.py).
 *
 * This file supplies (a) the ~25 symbols those two objects import from the rest of the reference - allocation of a Buffer,
 * the global option structs (all zero = defaults), diagnostics - as plain stand-ins written here, and (b) small entry
 * points with plain-C signatures that build a VBlock / Context by hand (zeroed, the few fields the functions read set
 * explicitly), call the reference's functions and hand the buffers back. Nothing here is product code.
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <string.h>
#include "genozip.h"
#include "vblock.h"
#include "context.h"
#include "buffer.h"
#include "flags.h"
#include "segconf.h"
#include "file.h"
#include "b250.h"
#include "dyn_int.h"
#include "local_type.h"
#include "data_types.h"
#include "seg.h"
#include "strings.h"
#include "profiler.h"
#include "bits.h"
#include "hash.h"
#include "codec.h"
#include "sam.h"
#include "reference.h"

/* ---- (a) what b250.o / dyn_int.o import -------------------------------------------------------------------------------- */
Flags flag;
SegConf segconf;
FileP txt_file, z_file;
FILE *info_stream;
CommandType primary_command = ZIP;
FileMode READ = "rb", WRITE = "wb", WRITEREAD = "wb+";
const LocalTypeDesc lt_desc[NUM_LOCAL_TYPES] = LOCALTYPE_DESC;
DataTypeProperties dt_props[NUM_DATATYPES], dt_props_def;
static uint32_t shim_num_samples;
__attribute__((constructor)) static void shim_defaults (void) { flag.show_time_comp_i = COMP_NONE; flag.command = ZIP; flag.is_lten = true; }   /* --show-time off (flags.c default); we are genozip, not genounzip */

void buf_alloc_do (VBlockP vb, BufferP buf, uint64_t requested_size, float grow_at_least_factor, rom name, FUNCLINE)
{
    if (buf->size >= requested_size && buf->data) return;
    uint64_t sz = requested_size * (grow_at_least_factor > 1 ? grow_at_least_factor : 1) + 64;
    char *m = realloc (buf->memory, sz + 16);
    if (!m) abort ();
    buf->memory = m; buf->data = m + 8; buf->size = sz; buf->vb = vb; buf->name = name; buf->type = BUF_REGULAR;
}
void buf_free_do (BufferP buf, FUNCLINE) { buf->len = 0; buf->param = 0; }
/* src/buffer.c (compiled in place for its byte order / interlace / transpose functions) imports a few more */
VBlockP evb;
uint64_t buf_mem_size (ConstBufferP buf) { return buf->size; }
void buf_overlay_do (VBlockP vb, BufferP top_buf, BufferP bottom_buf, uint64_t start_in_bottom, bool copy_len, FUNCLINE, rom name)
{   /* the top buffer shares the bottom one's memory (enough of src/buf_struct.c's overlay for codec_acgt_compress) */
    top_buf->data = bottom_buf->data + start_in_bottom; top_buf->size = bottom_buf->size - start_in_bottom; top_buf->vb = vb;
    top_buf->type = BUF_REGULAR; top_buf->memory = NULL; if (copy_len) top_buf->len = bottom_buf->len;
}
void buf_set_shared (BufferP buf) {}
void buf_destroy_do (BufferP buf, FUNCLINE) { buf->data = NULL; buf->len = 0; buf->size = 0; buf->memory = NULL; }
void codec_show_time (VBlockP vb, rom name, rom subname, Codec codec) {}
rom buf_type_name (ConstBufferP buf) { return "buf"; }
void bits_clear_region_do (BitsP bits, uint64_t start, uint64_t len, FUNCLINE) { abort (); }
void bits_set_region (BitsP bits, uint64_t start, uint64_t len) { abort (); }
bool file_put_data (rom filename, const void *data, uint64_t len, mode_t mode) { return false; }
void file_gzip (char *filename) {}
void warn (rom fmt, ...) {}
/* (buf_copy_do is the reference's own: src/buffer.c; str_time, str_str_s_, char_to_printable, str_int_commas, str_is_zero: src/strings.c) */
const BufDescType buf_desc (ConstBufferP buf) { BufDescType d = {}; return d; }
void error_assert_failed (rom func, uint32_t line, rom fmt, ...) { va_list a; va_start (a, fmt); fprintf (stderr, "reference ASSERT in %s:%u: ", func, line); vfprintf (stderr, fmt, a); fprintf (stderr, "\n"); va_end (a); abort (); }
void error_assertinp_failed (rom fmt, ...) { va_list a; va_start (a, fmt); vfprintf (stderr, fmt, a); va_end (a); abort (); }
Codec codec_assign_best_codec (VBlockP vb, ContextP ctx, BufferP data, SectionType st) { return CODEC_UNKNOWN; }
void ctx_decrement_count (VBlockP vb, ContextP ctx, WordIndex node_index) {}
bool fastq_zip_use_pair_identical (DictId dict_id) { return false; }
bool is_fastq_pair_2 (VBlockP vb) { return false; }
StrText vb_name (VBlockP vb) { StrText s = { "shim" }; return s; }
StrText line_name (VBlockP vb) { StrText s = { "shim" }; return s; }
rom lt_name (LocalType lt) { return "lt"; }
rom store_type_name (StoreType st) { return "store"; }
StrText1K seg_error (VBlockP vb) { StrText1K s = {}; return s; }
void show_time_one (VBlockP vb, rom res, uint64_t delta) {}
uint32_t vcf_header_get_num_samples (void) { return shim_num_samples; }

/* ---- (b) entry points --------------------------------------------------------------------------------------------------- */
static VBlockP new_vb (void)
{
    VBlockP vb = calloc (1, sizeof (VBlock));
    vb->data_type = DT_FASTQ; vb->vblock_i = 2;
    return vb;
}
static void free_ctx (ContextP c) { free (c->b250.memory); free (c->local.memory); free (c->nodes.memory); free (c); }

/* b250_seg_append (src/b250.c:112) for every node index in turn. Returns the byte length; *count / *all_the_same as the context holds them */
long ctxref_b250_seg (const int32_t *node_index, uint32_t n, uint32_t ol_len, uint8_t *out, uint64_t *count, int *all_the_same)
{
    VBlockP vb = new_vb ();
    ContextP ctx = calloc (1, sizeof (Context));
    ctx->dict_id.num = 0x1234; ctx->ol_nodes.len32 = ol_len; ctx->flags.all_the_same = false;
    for (uint32_t i = 0; i < n; i++) b250_seg_append (vb, ctx, node_index[i]);
    const long len = (long)ctx->b250.len;
    if (len) memcpy (out, ctx->b250.data, len);
    *count = ctx->b250.count; *all_the_same = ctx->flags.all_the_same;
    free_ctx (ctx); free (vb);
    return len;
}

/* b250_zip_generate (src/b250.c:202) on a seg-format buffer. nodes = the VBlock's nodes after the merge (word indices). Returns the length */
long ctxref_b250_generate (const uint8_t *seg, uint32_t seg_len, uint64_t count, int all_the_same, uint32_t ol_len,
                           const int32_t *node2word, uint32_t n_new, uint8_t *out)
{
    VBlockP vb = new_vb ();
    ContextP ctx = calloc (1, sizeof (Context));
    ctx->dict_id.num = 0x1234; ctx->ol_nodes.len32 = ol_len; ctx->nodes_converted = true; ctx->flags.all_the_same = all_the_same;
    buf_alloc_do (vb, &ctx->b250, seg_len + 8, 1, "b250", __FUNCTION__, __LINE__);
    memcpy (ctx->b250.data, seg, seg_len); ctx->b250.len = seg_len; ctx->b250.count = count;
    buf_alloc_do (vb, &ctx->nodes, (uint64_t)(n_new + 1) * sizeof (CtxNode), 1, "nodes", __FUNCTION__, __LINE__);
    memcpy (ctx->nodes.data, node2word, (size_t)n_new * 4); ctx->nodes.len = n_new;
    (void)b250_zip_generate (vb, ctx);
    const long len = (long)ctx->b250.len;
    if (len) memcpy (out, ctx->b250.data, len);
    ctx->b250.data = ctx->b250.memory + 8;            /* (the function shifts the start of the buffer) */
    free_ctx (ctx); free (vb);
    return len;
}

/* dyn_int_append (src/dyn_int.c:304) / dyn_int_append_nothing_char (:324) for a column, then the final type (dyn_int_get_ltype :27).
 * Returns the ltype; *len = bytes in out (native little endian) */
int ctxref_dyn_int_column (const int64_t *values, const uint8_t *is_nothing, uint64_t n, int nothing_char, uint8_t *out, uint64_t *len)
{
    VBlockP vb = new_vb ();
    ContextP ctx = calloc (1, sizeof (Context));
    ctx->dict_id.num = 0x1234; ctx->ltype = LT_DYN_INT; ctx->nothing_char = (char)nothing_char;
    for (uint64_t i = 0; i < n; i++)
        if (is_nothing && is_nothing[i]) dyn_int_append_nothing_char (vb, ctx, 0);
        else dyn_int_append (vb, ctx, values[i], 0);
    const LocalType lt = n ? dyn_int_get_ltype (ctx) : LT_UINT8;
    *len = ctx->local.len * lt_width (ctx) * 0 + ctx->local.len * lt_desc[lt].width;
    if (*len) memcpy (out, ctx->local.data, *len);
    free_ctx (ctx); free (vb);
    return (int)lt;
}

/* dyn_int_transpose (src/dyn_int.c:45), full matrix: data = rows x cols elements of `ltype` (LT_UINT8/16/32) already in file byte
 * order (zip_generate_local orders before it transposes, src/zip.c:185-219). Returns the resulting ltype */
int ctxref_dyn_int_transpose (int ltype, const uint8_t *data, uint64_t n_elems, uint32_t cols, uint8_t *out)
{
    VBlockP vb = new_vb ();
    vb->data_type = DT_VCF;
    ContextP ctxs = calloc (MAX_DICTS, sizeof (Context));      /* (the function looks at CTX(VCF_COPY_SAMPLE)) */
    memcpy (&vb->ca.contexts, &ctxs, 0);                        /* no-op: contexts live inside the VBlock */
    ContextP ctx = &vb->ca.contexts[200];
    ctx->dict_id.num = 0x1234; ctx->ltype = (LocalType)ltype; ctx->dyn_transposed = true;
    const unsigned w = lt_desc[ltype].width;
    buf_alloc_do (vb, &ctx->local, n_elems * w + 8, 1, "local", __FUNCTION__, __LINE__);
    memcpy (ctx->local.data, data, n_elems * w); ctx->local.len = n_elems;
    if (cols <= 255) ctx->local.n_cols = cols; else { ctx->local.n_cols = 0; shim_num_samples = cols; }
    dyn_int_transpose (vb, ctx);
    memcpy (out, ctx->local.data, n_elems * w);
    const int lt = ctx->ltype;
    free (ctx->local.memory); free (vb->scratch.memory); free (ctxs); free (vb);
    return lt;
}

/* zip_generate_local's byte-order step (src/zip.c:178-216) with the reference's own converters (src/buffer.c:336-350): in place,
 * n elements of `ltype`, native little endian -> file order */
void ctxref_local_to_file_order (int ltype, uint8_t *data, uint64_t n)
{
    Buffer b = {}; b.data = (char *)data; b.len = n; b.size = n * 8;
    switch (ltype) {
        case LT_UINT32 : case LT_FLOAT32 : BGEN_u32_buf (&b, NULL); break;
        case LT_UINT16 : BGEN_u16_buf (&b, NULL); break;
        case LT_UINT64 : case LT_FLOAT64 : BGEN_u64_buf (&b, NULL); break;
        case LT_INT8   : interlace_d8_buf (&b, NULL); break;
        case LT_INT16  : BGEN_interlace_d16_buf (&b, NULL); break;
        case LT_INT32  : BGEN_interlace_d32_buf (&b, NULL); break;
        case LT_INT64  : BGEN_interlace_d64_buf (&b, NULL); break;
        default : break;
    }
}

/* the PIZ side of a local type (lt_desc[ltype].file_to_native, src/local_type.h:75-108 as piz_adjust_one_local applies it,
 * src/piz.c:219-245): in place; cols for the transposed types. Returns the resulting ltype */
int ctxref_local_to_native (int ltype, uint8_t *data, uint64_t n, uint32_t cols)
{
    VBlockP vb = new_vb ();
    Buffer b = {}; 
    buf_alloc_do (vb, &b, n * 8 + 8, 1, "local", __FUNCTION__, __LINE__);
    memcpy (b.data, data, n * lt_desc[ltype].width); b.len = n; b.vb = vb;
    if (cols <= 255) b.n_cols = cols; else { b.n_cols = 0; shim_num_samples = cols; }
    LocalType lt = (LocalType)ltype;
    if (lt_desc[ltype].file_to_native) lt_desc[ltype].file_to_native (&b, &lt);
    memcpy (data, b.data, n * lt_desc[ltype].width);
    free (b.memory); free (vb->scratch.memory); free (vb);
    return (int)lt;
}

/* hash_do (src/hash.h:30-52, a static inline of the reference's header): the bucket of a snip */
uint32_t ctxref_hash_do (uint32_t hash_len, const char *snip, uint32_t snip_len) { return hash_do (hash_len, snip, snip_len); }

/* ---- N3: the reference's own src/codec_domq.c (+ src/base64.c), compiled in place ---------------------------------------------
 * What codec_domq.o imports beyond the above: the codec table (a sub-codec that stores = CODEC_NONE is all that is needed: the
 * four streams are read back uncompressed), the snip the denormalisation table is segged as, and SAM / PIZ-side functions
 * that are never reached from here. */
static uint8_t shim_snip[16384]; static uint32_t shim_snip_len;
WordIndex seg_by_ctx_ex (VBlockP vb, STRp(snip), ContextP ctx, uint32_t add_bytes, bool *is_new) { memcpy (shim_snip, snip, snip_len); shim_snip_len = snip_len; return 0; }
WordIndex ctx_peek_next_snip (VBlockP vb, ContextP ctx, pSTRp (snip)) { abort (); }
void ctx_set_ltype (VBlockP vb, int ltype, ...) {}
void error_asspiz (rom func, uint32_t line, rom fmt, ...) { abort (); }
rom codec_name (Codec codec) { return "codec"; }
void sam_xcons_split_qual_line (VBlockP vb_, BufferP ql_buf) { abort (); }
void sam_reconstruct_missing_quality (VBlockP vb, ReconType reconstruct) { abort (); }
void sam_xcons_reconstruct_QUAL (VBlockP vb, ContextP ctx, uint32_t qual_len, bool reconstruct) { abort (); }
static COMPRESS (shim_store) { memcpy (compressed, uncompressed, *uncompressed_len); *compressed_len = *uncompressed_len; return true; }
static uint32_t shim_est (Codec codec, uint64_t len) { return (uint32_t)len; }
CodecArgs codec_args[NUM_CODECS] = { [CODEC_NONE] = { .is_simple = true, .name = "NONE", .compress = shim_store, .est_size = shim_est } };
extern COMPRESS (codec_domq_compress);

static char *dq_text; static const uint32_t *dq_off, *dq_len;
static COMPRESSOR_CALLBACK (dq_get_line)
{
    *line_data = dq_text + dq_off[vb_line_i]; *line_data_len = dq_len[vb_line_i] < maximum_size ? dq_len[vb_line_i] : maximum_size;
    if (is_rev) *is_rev = 0;
}

/* codec_domq_comp_init + codec_domq_compress over the QUAL lines of one VBlock. Outputs: the four locals, the denormalisation table
 * snip (base64 as segged), QUAL's param; *fit = what codec_domq_comp_init answers without `force`. Returns 0. */
int ctxref_domq (const uint8_t *text, uint64_t text_len, const uint32_t *off, const uint32_t *len, uint32_t n_lines,
                 uint8_t *qual, uint64_t *qual_len, uint8_t *runs, uint64_t *runs_len, uint8_t *mplx, uint64_t *mplx_len, uint8_t *divr, uint64_t *divr_len,
                 uint8_t *snip, uint32_t *snip_len, uint32_t *param, uint32_t *sub_codec, int *fit)
{
    VBlockP vb = new_vb ();
    vb->lines.len = n_lines;
    dq_text = malloc (text_len + 8); memcpy (dq_text, text, text_len); dq_off = off; dq_len = len;
    ContextP q = &vb->ca.contexts[SAM_QUAL];
    for (int k = 0; k < 4; k++) q[k].dict_id.num = 0x1234 + k;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_lines; i++) total += len[i];
    q->local.len = total;                                   /* (QUAL's local holds no data before the codec runs: the callback supplies the lines) */
    *fit = codec_domq_comp_init (vb, SAM_QUAL, dq_get_line, false);
    memset (q, 0, 4 * sizeof (Context)); for (int k = 0; k < 4; k++) q[k].dict_id.num = 0x1234 + k;
    q->local.len = total;
    memcpy (dq_text, text, text_len);
    shim_snip_len = 0;
    codec_domq_comp_init (vb, SAM_QUAL, dq_get_line, true);
    SectionHeaderCtx hd = {};
    uint32_t ulen = (uint32_t)total, clen = (uint32_t)(2 * total + 1024);
    char *comp = malloc (clen);
    codec_domq_compress (vb, q, (SectionHeaderP)&hd, NULL, &ulen, dq_get_line, comp, &clen, true, "QUAL");
    memcpy (qual, comp, clen); *qual_len = clen;
    memcpy (runs, q[1].local.data, q[1].local.len); *runs_len = q[1].local.len;
    memcpy (mplx, q[2].local.data, q[2].local.len); *mplx_len = q[2].local.len;
    memcpy (divr, q[3].local.data, q[3].local.len); *divr_len = q[3].local.len;
    memcpy (snip, shim_snip, shim_snip_len); *snip_len = shim_snip_len;
    *param = q->local.prm8[0]; *sub_codec = hd.sub_codec;
    free (comp); free (dq_text);
    for (int k = 0; k < 4; k++) free (q[k].local.memory);      /* (qual_line / normalize_buf alias other Buffers of the Context: left alone) */
    free (vb);
    return 0;
}

/* ---- N2: the reference's own src/codec_acgt.c, compiled in place ---------------------------------------------------------------
 * It imports the base -> 2-bit table of src/reference.c (:45-58; reference.c itself drags in the genome machinery): stated here from
 * its description - A C G T (either case) = 0 1 2 3, U = T, an IUPAC code = the alphabetically lowest base it stands for, anything
 * else 0. So the TABLE is this file's; the packing, the exception stream and the sub-codec rule are the reference's own code. */
#define LO(c) ((c) + 32)
const uint8_t _acgt_encode[256] = {   /* (everything not named: 0, as A R W M D H V N are) */
    ['C']=1, ['Y']=1, ['S']=1, ['B']=1, [LO('C')]=1, [LO('Y')]=1, [LO('S')]=1, [LO('B')]=1,
    ['G']=2, ['K']=2, [LO('G')]=2, [LO('K')]=2,
    ['T']=3, ['U']=3, [LO('T')]=3, [LO('U')]=3 };
extern COMPRESS (codec_acgt_compress);
static COMPRESS (shim_store_named) { memcpy (compressed, uncompressed, *uncompressed_len); *compressed_len = *uncompressed_len; return true; }

/* codec_acgt_compress over a contiguous NONREF.local: packed = what is handed to the sub-codec, x = NONREF_X.local (seq_len bytes) when
 * *has_x, *sub_codec = header->sub_codec (LZMA unless the packed data is under 50 bytes) */
int ctxref_acgt (const uint8_t *seq, uint32_t n, uint8_t *packed, uint32_t *packed_len, uint8_t *x, int *has_x, uint32_t *sub_codec)
{
    VBlockP vb = new_vb ();
    static Context zc[MAX_DICTS]; static File zf; z_file = &zf;        /* (ZCTX(..)->lcodec of NONREF_X is read) */
    codec_args[CODEC_LZMA].compress = shim_store_named; codec_args[CODEC_LZMA].est_size = shim_est;   /* the sub-codec stores */
    ContextP c = &vb->ca.contexts[SAM_NONREF];
    c[0].dict_id.num = 0x1234; c[1].dict_id.num = 0x1235; c[0].did_i = SAM_NONREF; c[1].did_i = SAM_NONREF_X;
    buf_alloc_do (vb, &c->local, n + 16, 1, "local", __FUNCTION__, __LINE__);
    memcpy (c->local.data, seq, n); c->local.len = n;
    SectionHeaderCtx hd = {};
    uint32_t ulen = n, clen = n + 1024;
    char *comp = malloc (clen);
    codec_acgt_compress (vb, c, (SectionHeaderP)&hd, c->local.data, &ulen, NULL, comp, &clen, false, "NONREF");
    memcpy (packed, comp, clen); *packed_len = clen;
    *has_x = !hd.flags.ctx.acgt_no_x;
    if (*has_x) memcpy (x, c->local.data, n);
    *sub_codec = hd.sub_codec;
    free (comp); free (c->local.memory); free (vb->scratch.memory); free (vb);
    return 0;
}

/* ---- a4 (its hash): the reference's own src/hash.c, compiled in place ----------------------------------------------------------
 * hash_alloc_global + hash_global_get_entry with the singleton tables (src/hash.c:229-240,280-482) decide, for every new node of
 * a VBlock context, whether it is a word the file already has, a new word, a singleton (diverted to local) or a failed singleton.
 * The loop around them is ctx_merge_in_one_vctx / ctx_commit_node (src/context.c:1000-1034,269-316; context.c does not build
 * outside the reference's tree), stated here in its essentials: count == 1 in a context that can have singletons allows a
 * singleton; a singleton's node gets the word index of the SNIP_LOOKUP snip; a new word is appended to the dictionary + NUL. */
#include "hash.h"
rom report_support_if_unexpected (void) { return ""; }
#include "dict_io.h"
StrText16K str_snip_ex (DataType dt, STRp(snip), bool add_quote) { static StrText16K s; return s; }

static WordIndex shim_commit (ContextP zctx, STRp(snip), bool allow_singleton, int *was_singleton)
{
    CtxNode *upd;
    WordIndex wi = hash_global_get_entry (zctx, STRa(snip), allow_singleton, false, &upd);
    if (wi != WORD_INDEX_NONE && !upd) return wi;                       /* an existing word */
    if (upd) {                                                          /* a new word: ctx_insert_to_dict (context.c:50-71) */
        buf_alloc_do (NULL, &zctx->dict, zctx->dict.len + snip_len + 1, 2, "dict", __FUNCTION__, __LINE__);
        upd->char_index = zctx->dict.len;
        memcpy (zctx->dict.data + zctx->dict.len, snip, snip_len); zctx->dict.data[zctx->dict.len + snip_len] = 0;
        zctx->dict.len += snip_len + 1;
        return wi;
    }
    *was_singleton = 1;                                                 /* a singleton: its node points at the SNIP_LOOKUP word */
    char lookup[1] = { SNIP_LOOKUP };
    int dummy = 0;
    return shim_commit (zctx, lookup, 1, false, &dummy);
}

/* VBlock contexts merging one after the other into one file context. Per VBlock: its new nodes (snips NUL-separated, with their
 * lengths and counts) and whether the context can have singletons. Out: per node the word index and whether it went to local;
 * the dictionary, the number of failed singletons, the prime the hash was given */
int ctxref_merge_hash (uint32_t estimated_entries, uint32_t n_vb, const uint32_t *n_nodes, const uint8_t *can_ston, const char *snips,
                       const uint32_t *snip_len, const uint32_t *count, int32_t *word_out, uint8_t *ston_out,
                       uint8_t *dict_out, uint64_t *dict_len_out, uint64_t *n_failed_out, uint32_t *hash_len_out)
{
    Context *zctx = calloc (1, sizeof (Context));
    strcpy (zctx->tag_name, "CTX");
    hash_alloc_global (zctx, estimated_entries);
    uint64_t at = 0, k = 0;
    for (uint32_t v = 0; v < n_vb; v++)
        for (uint32_t i = 0; i < n_nodes[v]; i++, k++) {
            int ston = 0;
            word_out[k] = shim_commit (zctx, snips + at, snip_len[k], can_ston[v] && count[k] == 1, &ston);
            ston_out[k] = (uint8_t)ston;
            at += snip_len[k] + 1;
        }
    memcpy (dict_out, zctx->dict.data, zctx->dict.len); *dict_len_out = zctx->dict.len;
    *n_failed_out = zctx->num_failed_singletons; *hash_len_out = zctx->global_hash.len32;
    free (zctx->dict.memory); free (zctx->nodes.memory); free (zctx->global_hash.memory); free (zctx->ston_hash.memory); free (zctx->ston_ents.memory); free (zctx);
    return 0;
}

/* ---- a1 (its hash): snip -> node index through hash_get_entry_for_seg (src/hash.c:529-576: the dictionary cloned from the file,
 * then the VBlock's own hash), under the few lines of ctx_create_node_do (src/context.c:320-404) that add a node when there is none:
 * the node index of a snip = its index among the cloned words, or the number of cloned words + its rank among the VBlock's new ones */
int ctxref_seg_nodes (uint32_t n_ol, const char *ol_snips, const uint32_t *ol_len, uint32_t n, const char *snips, const uint32_t *snip_len, int32_t *node_index_out)
{
    Context *zctx = calloc (1, sizeof (Context)), *vctx = calloc (1, sizeof (Context));
    VBlockP vb = new_vb ();
    strcpy (zctx->tag_name, "CTX"); strcpy (vctx->tag_name, "CTX");
    hash_alloc_global (zctx, n_ol + 1000);
    uint64_t at = 0;
    for (uint32_t i = 0; i < n_ol; i++) { int s = 0; if (shim_commit (zctx, ol_snips + at, ol_len[i], false, &s) != (WordIndex)i) return -1; at += ol_len[i] + 1; }
    vctx->global_hash = zctx->global_hash; vctx->ol_nodes = zctx->nodes; vctx->ol_dict = zctx->dict;        /* ctx_clone: overlays */
    vctx->num_new_entries_prev_merged_vb = 1000;                                                             /* (sizes the VBlock's hash: hash.c:76-78) */
    at = 0;
    for (uint32_t i = 0; i < n; i++) {
        rom in_dict;
        WordIndex wi = hash_get_entry_for_seg (vb, vctx, snips + at, snip_len[i], &in_dict);
        if (wi == WORD_INDEX_NONE) {                                                                         /* a new node (context.c:372-398) */
            buf_alloc_do (vb, &vctx->nodes, (vctx->nodes.len + 1) * sizeof (CtxNode), 2, "nodes", __FUNCTION__, __LINE__);
            buf_alloc_do (vb, &vctx->dict, vctx->dict.len + snip_len[i] + 1, 2, "dict", __FUNCTION__, __LINE__);
            CtxNode *nd = &((CtxNode *)vctx->nodes.data)[vctx->nodes.len++];
            *nd = (CtxNode){ .char_index = vctx->dict.len, .snip_len = snip_len[i], .next = NO_NEXT };
            memcpy (vctx->dict.data + vctx->dict.len, snips + at, snip_len[i]); vctx->dict.data[vctx->dict.len + snip_len[i]] = 0;
            vctx->dict.len += snip_len[i] + 1;
            wi = n_ol + vctx->nodes.len32 - 1;
        }
        node_index_out[i] = wi;
        at += snip_len[i] + 1;
    }
    free (vctx->nodes.memory); free (vctx->dict.memory); free (vctx->local_hash.memory);
    free (zctx->dict.memory); free (zctx->nodes.memory); free (zctx->global_hash.memory); free (zctx); free (vctx); free (vb);
    return 0;
}

/* ---- a3 (what is an integer): the reference's own str_get_int (src/strings.c:315-341), which seg_integer_or_not (src/seg.c:531-560)
 * asks for every snip of a numeric column */
void error_exit (bool show_stack, bool in_assert) { abort (); }
void progress_newline (void) {}
int ctxref_str_get_int (uint32_t n, const char *snips, const uint32_t *snip_len, uint8_t *is_int_out, int64_t *value_out)
{
    uint64_t at = 0;
    for (uint32_t i = 0; i < n; i++) { value_out[i] = 0; is_int_out[i] = str_get_int (snips + at, snip_len[i], &value_out[i]); at += snip_len[i] + 1; }
    return 0;
}
