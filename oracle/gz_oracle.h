/* gz_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the Genozip entropy-coding hot path that genozip_amd implements in HIP.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * library (genozip_amd/csrc -> libgenozip_amd.so) never links or calls it.
 *
 * Every function cites the reference file:line whose behaviour it restates (paths relative to /root/reference).
 * Parity status: the rANS-4x16 / arith codecs are PINNED byte-for-byte against the reference's own vendored
 * htscodecs sources compiled in place (oracle/_ref, see oracle/Makefile and tests/golden/). The Genozip-authored
 * pieces (b250 VARL, local transforms, section framing) are pinned only by the hand-derived KATs of SURVEY.md
 * Appendix A -- beyond those: PARITY UNPINNED (the reference ships no fixtures and cannot be built here).
 */
#ifndef GZ_ORACLE_H
#define GZ_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* htscodecs order-byte flags (src/htscodecs/arith_dynamic.h:43-50) */
#define GZO_X_PACK   0x80
#define GZO_X_RLE    0x40
#define GZO_X_CAT    0x20
#define GZO_X_NOSZ   0x10
#define GZO_X_STRIPE 0x08
#define GZO_X_EXT    0x04
#define GZO_X_ORDER  0x03

/* Genozip codec ids = file-format values (src/genozip.h:325-360) */
enum { GZO_CODEC_UNKNOWN = 0, GZO_CODEC_NONE = 1, GZO_CODEC_RANB = 6, GZO_CODEC_RANW = 7, GZO_CODEC_RANb = 8,
       GZO_CODEC_RANw = 9, GZO_CODEC_ARTB = 16, GZO_CODEC_ARTW = 17, GZO_CODEC_ARTb = 18, GZO_CODEC_ARTw = 19 };

/* ---- htscodecs level (rows a12-a14) ---- */
uint32_t gzo_rans_bound  (uint32_t size, int order);              /* rANS_static4x16pr.c:357 */
uint32_t gzo_arith_bound (uint32_t size, int order);              /* arith_dynamic.c:74      */

/* return compressed length, or -1 when out_cap < bound (the reference's NULL return) */
long gzo_rans_compress   (const uint8_t *in, uint32_t in_size, uint8_t *out, uint32_t out_cap, int order); /* rANS_static4x16pr.c:1151 */
long gzo_arith_compress  (const uint8_t *in, uint32_t in_size, uint8_t *out, uint32_t out_cap, int order); /* arith_dynamic.c:615      */
/* return number of bytes decoded (== out_len on success) or -1 on malformed input */
long gzo_rans_uncompress (const uint8_t *in, uint32_t in_size, uint8_t *out, uint32_t out_len);           /* rANS_static4x16pr.c:1358 */
long gzo_arith_uncompress(const uint8_t *in, uint32_t in_size, uint8_t *out, uint32_t out_len);           /* arith_dynamic.c:860      */

/* diagnostic: the e10/e12 ratio compute_shift() saw in the most recent order-1 rANS call of this thread
 * (rANS_static4x16pr.c:681) -- tests use it to prove inputs are far from the 1.01 decision boundary */
double gzo_last_shift_ratio (void);

/* ---- codec plugin surface (rows a10/a11; src/codec.h:17-40, src/codec_htscodecs.c:17-33,77-123) ---- */
uint32_t gzo_codec_est_size (int codec, uint64_t uncompressed_len);
/* returns 1 on success; 0 if *out_len (capacity) < est_size and soft_fail; -1 on any other error */
int gzo_codec_compress   (int codec, const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t *out_len, int soft_fail);
int gzo_codec_uncompress (int codec, const uint8_t *in, uint32_t in_len, uint8_t *out, uint64_t out_len);
/* many independent streams on a pthread pool (the CPU baseline of bench.py): returns 0 if all succeeded */
int gzo_codec_compress_many (int n, const int *codecs, const uint8_t *const *ins, const uint32_t *in_lens,
                             uint8_t *const *outs, uint32_t *out_lens /* in: capacity, out: length */, int n_threads);
int gzo_codec_compress_many_rep (int n, const int *codecs, const uint8_t *const *ins, const uint32_t *in_lens,
                                 uint8_t *const *outs, uint32_t *out_lens, int n_threads, int replicas);

/* deterministic stand-in for codec_assign_best_codec (src/codec.c:234-363): smallest framed size on
 * the first min(len,99999) bytes, ties -> lower codec id; returns GZO_CODEC_UNKNOWN when len < 50 */
int gzo_codec_assign_best (const uint8_t *in, uint32_t in_len, uint32_t *sizes_out /* [9] or NULL */);

/* ---- b250 (rows a2, a5; src/b250.c:60-110,112-163,202-267) ---- */
/* seg-time encoding of one entry (little endian, type tag in LAST byte); new nodes (>= ol_nodes_len) always 4 B.
 * returns bytes written (1..4) */
uint32_t gzo_b250_seg_put (uint8_t *dst, int32_t node_index, uint32_t ol_nodes_len);
uint64_t gzo_b250_seg_put_many (uint8_t *dst, const int32_t *node_index, uint64_t n, uint32_t ol_nodes_len);
/* PIZ-format big-endian encoding of one word index (tag in first byte); returns bytes written */
uint32_t gzo_b250_piz_put (uint8_t *dst, int32_t wi);
/* b250_zip_generate: seg-format buffer -> PIZ VARL format. node2word[i] is the word_index of VB-local node
 * (ol_nodes_len + i). out must hold seg_len bytes. Returns output length (<= seg_len), -1 on malformed input. */
long gzo_b250_generate (const uint8_t *seg, uint32_t seg_len, uint32_t ol_nodes_len,
                        const int32_t *node2word, uint32_t n_new_nodes, uint8_t *out);
/* decode a PIZ VARL stream into word indices (ONE_UP resolved); returns count or -1 */
long gzo_b250_piz_decode (const uint8_t *b, uint32_t len, int32_t *wi_out, uint32_t wi_cap);

/* ---- local generation (rows a6, a7; src/zip.c:167-219, src/buffer.c:336-350, src/context.h:99-101,
 *      src/dyn_int.c:45-132) ---- */
enum { GZO_LT_INT8 = 1, GZO_LT_UINT8 = 2, GZO_LT_INT16 = 3, GZO_LT_UINT16 = 4, GZO_LT_INT32 = 5, GZO_LT_UINT32 = 6,
       GZO_LT_INT64 = 7, GZO_LT_UINT64 = 8, GZO_LT_FLOAT32 = 9, GZO_LT_FLOAT64 = 10, GZO_LT_BLOB = 11,
       GZO_LT_BITMAP = 12, GZO_LT_UINT8_TR = 14, GZO_LT_UINT16_TR = 15, GZO_LT_UINT32_TR = 16 };
uint32_t gzo_lt_width (int ltype);
/* in-place: native little-endian elements -> file byte order (BGEN, interlace). n = element count */
int  gzo_local_to_file_order (int ltype, void *data, uint64_t n);
/* rows x cols -> cols x rows of `width`-byte elements (element bytes untouched) */
void gzo_transpose (const void *src, void *dst, uint32_t rows, uint32_t cols, uint32_t width);
/* full zip_generate_local for integer ltypes: returns final ltype (e.g. LT_UINT8_TR when transposed) */
int  gzo_local_generate (int ltype, void *data, uint64_t n, uint32_t transpose_cols /*0 = no transpose*/, void *scratch);
uint8_t gzo_bitmap_param (uint64_t nbits); /* zip.c:181 */

/* ---- section framing (rows a9, a10, a16; src/sections.h:146-167,342-369,419-435, src/compressor.c:55-58,
 *      114-161, src/zfile.c:288-395,1108-1144) ---- */
uint32_t gzo_adler32 (uint32_t adler, const uint8_t *buf, size_t len); /* == libdeflate_adler32, compressor.c:161 */

typedef struct {
    uint32_t vblock_i;       /* 1-based */
    uint8_t  section_type;   /* 11 = SEC_B250, 12 = SEC_LOCAL (src/genozip.h:378-379) */
    uint8_t  codec;          /* requested codec; < 50 B payload is rewritten to CODEC_NONE (compressor.c:56-58) */
    uint8_t  sub_codec;
    uint8_t  flags;          /* struct FlagsCtx packed LSB first (sections.h:99-118) */
    uint8_t  ltype;
    uint8_t  param;
    uint8_t  b250_size_or_nothing_char;
    uint8_t  dict_id[8];
} GzoCtxSectionDesc;

#define GZO_SECTION_HEADER_LEN     28
#define GZO_CTX_SECTION_HEADER_LEN 40
#define GZO_VB_HEADER_LEN          84
#define GZO_MAGIC                  0x27052012u

/* comp_compress for a SEC_B250 / SEC_LOCAL section: header(40) || payload appended at z; returns total bytes
 * appended or -1. z_cap = remaining capacity. */
long gzo_section_compress (const GzoCtxSectionDesc *d, const uint8_t *data, uint32_t data_len, uint8_t *z, uint64_t z_cap);
/* 84-byte VB header (no payload): zfile_compress_vb_header; z_data_bytes patched later by gzo_vb_header_patch */
void gzo_vb_header_write (uint8_t *z, uint32_t vblock_i, uint32_t recon_size, uint32_t longest_line_len,
                          uint32_t longest_seq_len, const uint8_t digest[16], uint8_t flags);
void gzo_vb_header_patch (uint8_t *z, uint32_t z_data_bytes); /* zfile.c:1139-1144 */

/* ---- CODEC_ACGT pre-transform (codec_acgt.c:45-55,64-129, reference.c:45-58): SEQ -> 2 bits per base
 * (little-endian 64-bit words, base i in bits 2i..2i+1, excess bits of the top word clear) + the exception stream
 * NONREF_X (0 for ACGT, 1 for acgt, the character itself otherwise). The LZMA sub-codec stays outside the path. */
uint64_t gzo_acgt_packed_len (uint64_t n);                     /* bytes: whole 64-bit words */
int  gzo_acgt_pack (const uint8_t *seq, uint64_t n, uint8_t *packed, uint8_t *x);   /* x may be seq; returns has_x */
void gzo_acgt_unpack (const uint8_t *packed, const uint8_t *x /* or NULL */, uint64_t n, uint8_t *seq);   /* codec_acgt.c:177-199,232-246 */

/* ---- seg-side appends (rows a1-a3): a whole column of one context at once ---------------------------------------
 * Test infrastructure like the rest of this file. What the segmenter leaves behind in a context after evaluating a
 * column of snips one by one (ctx_create_node_do context.c:320-404, hash_get_entry_for_seg hash.c:530-576,
 * ctx_insert_to_dict context.c:50-71, b250_seg_append b250.c:112-163): for every snip its node index - the index in
 * ol_nodes when the snip is in the dictionary cloned from the file-level context, else ol_nodes_len + the rank of
 * its first occurrence in this VBlock; the VBlock's own dictionary (new snips in order of first occurrence, each
 * followed by a NUL) and nodes (char_index, snip_len); counts[] (one per node, occurrences in this VBlock; empty and
 * missing snips are not counted, context.c:331-335); and the seg-format b250 with its all-the-same collapse (while
 * every entry is the same node only ONE entry is stored and `count` goes up).
 * A snip with len 0 is WORD_INDEX_EMPTY (-3), or WORD_INDEX_MISSING (-4) when off == GZO_SNIP_MISSING (the reference's
 * snip == NULL). The hash table itself (hash.h:30-52) never reaches the file; only its effect does. */
#define GZO_SNIP_MISSING 0xffffffffu
typedef struct {
    int32_t  *node_index;        /* [n] */
    uint8_t  *dict;              /* [sum of (len+1) over the snips] */
    uint64_t  dict_len;
    uint64_t *node_char_index;   /* [n] */
    uint32_t *node_snip_len;     /* [n] */
    uint32_t  n_new;
    uint32_t *counts;            /* [n_ol + n] */
    uint8_t  *b250;              /* [4 n] */
    uint64_t  b250_len;
    uint64_t  b250_count;
    int       all_the_same;
} GzoColumn;
int gzo_ctx_seg_column (const uint8_t *text, const uint32_t *off, const uint32_t *len, uint64_t n,
                        const uint8_t *ol_dict, const uint64_t *ol_char_index, const uint32_t *ol_snip_len, uint32_t n_ol,
                        GzoColumn *out);
/* ... with a node created before anything is segged (R2 VBlocks' mate_lookup node in SQBITMAP, fastq.c:664-665) */
int gzo_ctx_seg_column_pre (const uint8_t *text, const uint32_t *off, const uint32_t *len, uint64_t n,
                            const uint8_t *ol_dict, const uint64_t *ol_char_index, const uint32_t *ol_snip_len, uint32_t n_ol,
                            const uint8_t *pre_snip, uint32_t pre_len, GzoColumn *out);

/* dyn_int_append over a whole column (dyn_int.c:17,232-320, dyn_int_get_ltype :27-43): the final local type is the
 * first of UINT8, INT8, UINT16, INT16, UINT32, INT32, INT64 that holds every value (one less at the top when the
 * context has a nothing_char, whose entries are stored as the type's maximum, :322-345); out = the values in that
 * width, native little endian (zip_generate_local then orders them). Returns the ltype (GZ_LT_* numbering). */
int gzo_dyn_int_column (const int64_t *values, const uint8_t *is_nothing /* or NULL */, uint64_t n, int nothing_char,
                        uint8_t *out /* 8 n */);

/* seg_add_to_local_fixed_do over a column (seg.c:1268-1287): the snips one after the other, each followed by a NUL
 * if add_nul. Returns the length. */
uint64_t gzo_local_blob_column (const uint8_t *text, const uint32_t *off, const uint32_t *len, uint64_t n, int add_nul, uint8_t *out);
uint64_t gzo_local_blob_column_ex (const uint8_t *text, const uint32_t *off, const uint32_t *len, uint64_t n, int add_nul,
                                   const uint8_t *pre, uint32_t pre_len, uint32_t pad_to, uint8_t pad_byte, uint8_t *out, uint32_t *item_off);

/* ---- N1 (first part): the line buffer -> per-field (offset, length) columns -----------------------------------------
 * seg_get_next_line (seg.c:200-236) over a whole buffer: start and length of every line, the length without the
 * newline and without a '\r' before it; a last line without a newline counts. Returns the number of lines
 * (all of them, even beyond cap; only the first cap are written). */
int64_t gzo_bam_records (const uint8_t *bam, uint64_t n, uint32_t *rec_off, uint64_t cap);
int64_t gzo_bam_to_sam (const uint8_t *bam, const uint32_t *rec_off, uint64_t n_rec, const uint8_t *ref_names, const uint32_t *ref_name_off, int32_t n_ref,
                        uint8_t *text, uint64_t cap, uint32_t *line_off);
uint64_t gzo_text_lines (const uint8_t *text, uint64_t n, uint32_t *off, uint32_t *len, uint64_t cap);
/* fastq_seg_get_lines (fastq.c:1002-1135): every 4 lines are a read: '@' + line 1, SEQ, '+' + line 3, QUAL. The
 * columns hold line 1 without its '@', SEQ, line 3 without its '+', QUAL. Returns 0, or -1 - k where read k is the
 * first that is malformed (line 1 not starting with '@', line 3 not with '+', QUAL and SEQ of different lengths;
 * :1008-1010,1076,1121). Lines beyond the last whole read are ignored. */
long gzo_fastq_records (const uint8_t *text, const uint32_t *line_off, const uint32_t *line_len, uint64_t n_lines,
                        uint32_t *l1_off, uint32_t *l1_len, uint32_t *seq_off, uint32_t *seq_len,
                        uint32_t *l3_off, uint32_t *l3_len, uint32_t *qual_off, uint32_t *qual_len);
/* The items of a container whose separators are known (the QNAME flavors, qname_flavors.h:21-49 with qname.c:715-866;
 * seg_get_next_item seg.c:153-198): item i of snip k ends at the first seps[i] after item i-1, the last item is the
 * rest. item_off / item_len are item-major: [i * n + k]. A snip that lacks a separator is "bad" (the reference then
 * segs it whole): item 0 = the whole snip, the others empty. Returns the number of bad snips. */
uint64_t gzo_tokenize_column (const uint8_t *text, const uint32_t *off, const uint32_t *len, uint64_t n,
                              const uint8_t *seps, uint32_t n_seps, uint32_t *item_off, uint32_t *item_len);

/* seg_integer_or_not over a column (seg.c:531-560 with str_get_int strings.c:315-341): a snip that is a decimal integer
 * the text can be rebuilt from ("" "-" "030" "-0" are not) goes to the context's dyn-int local and leaves the
 * one-character snip SNIP_LOOKUP in the b250; the context's nothing_char alone does the same with a "nothing" entry;
 * anything else stays a snip. snip_off/snip_len = the column for gzo_ctx_seg_column (lookup_off = where a SNIP_LOOKUP
 * byte lives in text), values/is_nothing = the compacted column for gzo_dyn_int_column. Returns the number of values.
 * Numbers beyond int64 are "not an integer" (the reference's overflow test there relies on signed wrap-around). */
uint64_t gzo_seg_integer_or_not (const uint8_t *text, const uint32_t *off, const uint32_t *len, uint64_t n,
                                 int nothing_char, uint32_t lookup_off, uint32_t *snip_off, uint32_t *snip_len,
                                 int64_t *values, uint8_t *is_nothing);

/* dyn_int_transpose, partial case (dyn_int.c:64-72,89-96,104-129; VCF samples copied by VCF_COPY_SAMPLE are absent from
 * local): `in` holds, in row-major order, only the elements of the rows x cols matrix whose missing[r*cols+c] is 0;
 * `out` receives the same elements in column-major order (to_file) - or the other way round (PIZ). Elements are w bytes
 * (already in file byte order: BGEN comes first, zip.c:185-219). Returns the number of elements, -1 if it is not
 * n_present. */
long gzo_transpose_partial (const uint8_t *in, uint64_t n_present, uint32_t rows, uint32_t cols, uint32_t w,
                            const uint8_t *missing, uint8_t *out, int to_file);


/* ---- row a4: the ordered dictionary merge (context.c:269-316,938-1079; hash.c:227-482), the reference's structures ----
 * PARITY UNPINNED. One GzoZctx per file-level context; gzo_ctx_merge = ctx_merge_in_one_vctx for one VBlock context. */
typedef struct GzoZctx GzoZctx;
typedef struct {
    uint32_t vblock_i, n_ol, n_new;
    const uint8_t *dict; const uint64_t *node_char_index; const uint32_t *node_snip_len; const uint32_t *counts;
    uint8_t can_have_singletons, flags, no_drop_b250, pair2_identical;
    uint64_t b250_len, local_len, b250_r1_len, local_r1_len;
    int32_t ats_node_index;
    uint8_t dropped_b250;                          /* out */
    int32_t *node2word;                            /* out [n_new] */
    uint8_t *ston_local; uint64_t ston_len; uint32_t n_stons;   /* out: must hold the VBlock's dict length */
} GzoMerge;
uint32_t gzo_hash_next_size_up (uint64_t size);
uint32_t gzo_hash_do (uint32_t hash_len, const uint8_t *snip, uint32_t snip_len);   /* hash.h:30-52 */
GzoZctx *gzo_zctx_create (uint32_t estimated_entries);
void gzo_zctx_destroy (GzoZctx *z);
int gzo_ctx_merge (GzoZctx *z, GzoMerge *j);
void gzo_zctx_view (const GzoZctx *z, const uint8_t **dict, uint64_t *dict_len, uint32_t *n_words, const uint64_t **counts,
                    uint64_t *n_failed, int *rm_dict);


/* ---- N3: CODEC_DOMQ's pre-transform (src/codec_domq.c:69-134,139-249,347-503) - QUAL lines -> four streams ------------------
 * QUAL (non-dominant normalised scores, a `no_doms` marker where no run precedes one, a final marker after a trailing run),
 * DOMQRUNS (run lengths of the dominant score: 0-254 = a run of that many, 255 = 254 and the run continues; runs span lines),
 * QUALMPLX (per line: index of its dom in the denormalisation table, | 0x80 for a diverse line), DIVRQUAL (the normalised
 * scores of diverse lines: < 85 % dom). denorm = num_doms x num_norm_qs table (-> base64 snip in DOMQRUNS' dictionary);
 * QUAL's section param = num_norm_qs | 0x80. Lines of length 0 take no part. */
typedef struct {
    uint8_t *qual, *runs, *mplx, *divr; uint64_t qual_len, runs_len, mplx_len, divr_len;
    uint8_t denorm[95 * 95]; uint32_t num_doms, num_norm_qs; int has_diverse, all_diverse;
} GzoDomq;
int  gzo_domq_is_fit (const uint8_t *text, const uint32_t *off, const uint32_t *len, uint64_t n_lines);
int  gzo_domq_encode (const uint8_t *text, const uint32_t *off, const uint32_t *len, uint64_t n_lines, GzoDomq *out);   /* -1: a byte outside ' '..'~' */
void gzo_domq_free (GzoDomq *o);

/* ---- a8 in full: the sorter of the twelve candidates' trials (src/codec.c:122-173,338) and a section around a payload made by a
 * codec that is not restated here (the host's BZ2 / LZMA / BSC) */
typedef struct { int32_t codec; float size; float clock; } GzoCodecTest;
int  gzo_assign_sort (GzoCodecTest *tests, int n, int mode);       /* sorted in place, returns tests[0].codec; n <= 64 */
long gzo_section_frame (const GzoCtxSectionDesc *d, const uint8_t *payload, uint32_t payload_len, uint32_t raw_len, uint8_t *z, uint64_t z_cap);

#ifdef __cplusplus
}
#endif
#endif
