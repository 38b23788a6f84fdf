/* oracle/ref_assign_shim.c -- TEST INFRASTRUCTURE ONLY (see oracle/Makefile, target "ref").
 *
 * Pins SURVEY 8a row a8. The reference's own src/codec.c is compiled as part of THIS translation unit (the #include below names the
 * file where it lies under /root/reference - nothing is copied), so that its static codec_assign_sorter and the whole of
 * codec_assign_best_codec (src/codec.c:128-173, :234-389) run as the reference wrote them, together with the reference's own
 * src/compressor.c, src/zfile.c, src/codec_none.c, src/codec_htscodecs.c, the vendored htscodecs and libdeflate's adler32 (compiled in
 * place by the Makefile) into oracle/_ref/libassignref.so. tests/golden/assign_golden.json is generated from it
 * (tests/golden/make_assign_golden.py).
 *
 * What this file supplies: (a) the clock - codec_assign_best_codec times every trial with clock() (:322-334); here clock() is a script
 * (the k-th trial takes the k-th entry of a table of ticks), which is what makes the reference's choice reproducible; (b) the three host
 * coders BZ2 / BSC / LZMA as candidates with scripted payload sizes (their compressors are sequential LZ / BWT coders outside the path,
 * SURVEY 2.1 - the candidate LOOP and the SORTER are what is pinned, with any sizes); (c) a VBlock / Context / zctx built by hand, mutexes
 * that do nothing (one thread), allocation of a Buffer, the option structs; (d) entry points with plain-C signatures. Everything else
 * the linked objects import and never reach is named by the generated stubs file (oracle/gen_ref_stubs.py). Nothing here is product code.
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <string.h>
#include <time.h>

/* (a) the scripted clock: call 2 k is the start of trial k, call 2 k + 1 its end */
static long shim_ticks[64], shim_now; static int shim_clock_calls, shim_n_ticks;
clock_t clock (void)
{
    if (shim_clock_calls & 1) { const int k = shim_clock_calls >> 1; shim_now += k < shim_n_ticks ? shim_ticks[k] : 0; }
    shim_clock_calls++;
    return (clock_t)shim_now;
}

#include "codec.c"                  /* the reference's own file, in place (-iquote $(REF)/src) */

#include "context.h"
#include "buffer.h"
#include "flags.h"
#include "segconf.h"
#include "file.h"
#include "sections.h"
#include "compressor.h"
#include "local_type.h"
#include "data_types.h"
#include "crypt.h"
#include "mutex.h"

/* ---- (c) what codec.c / compressor.c / zfile.c need from outside ------------------------------------------------------------- */
Flags flag;
SegConf segconf;
VBlockP evb;
FileP z_file, txt_file;
FILE *info_stream;
const LocalTypeDesc lt_desc[NUM_LOCAL_TYPES] = LOCALTYPE_DESC;
DataTypeProperties dt_props[NUM_DATATYPES], dt_props_def;
__attribute__((constructor)) static void shim_defaults (void) { flag.show_time_comp_i = COMP_NONE; flag.command = ZIP; info_stream = stderr; }

void buf_alloc_do (VBlockP vb, BufferP buf, uint64_t requested_size, float grow_at_least_factor, rom name, FUNCLINE)
{
    if (buf->size >= requested_size && buf->data) return;
    uint64_t sz = requested_size * (grow_at_least_factor > 1 ? grow_at_least_factor : 1) + 64;
    char *m = realloc (buf->memory, sz + 16);
    if (!m) abort ();
    buf->memory = m; buf->data = m + 8; buf->size = sz; buf->vb = vb; buf->name = name; buf->type = BUF_REGULAR;
}
void buf_free_do (BufferP buf, FUNCLINE) { free (buf->memory); memset (buf, 0, sizeof (*buf)); }    /* (codec_alloc_do looks for a buffer that is not allocated) */
void buf_destroy_do (BufferP buf, FUNCLINE) { free (buf->memory); memset (buf, 0, sizeof (*buf)); }
void *buf_low_level_malloc (size_t size, bool zero, FUNCLINE) { void *p = zero ? calloc (1, size ? size : 1) : malloc (size ? size : 1); if (!p) abort (); return p; }
void buf_low_level_free (void *p, FUNCLINE) { free (p); }
void *buf_low_level_realloc (void *p, size_t size, rom name, FUNCLINE) { return realloc (p, size); }
const BufDescType buf_desc (ConstBufferP buf) { BufDescType d = {}; return d; }
bool mutex_lock_do (MutexP mutex, bool blocking, FUNCLINE) { return true; }
void mutex_unlock_do (MutexP mutex, FUNCLINE) {}
void warn (rom fmt, ...) {}
rom report_support (void) { return ""; }
StrText vb_name (VBlockP vb) { StrText s = { "shim" }; return s; }
rom st_name (SectionType st) { return "SEC_section"; }
void show_time_one (VBlockP vb, rom res, uint64_t delta) {}
void error_assert_failed (rom func, uint32_t line, rom fmt, ...) { va_list a; va_start (a, fmt); fprintf (stderr, "reference ASSERT in %s:%u: ", func, line); vfprintf (stderr, fmt, a); fprintf (stderr, "\n"); va_end (a); abort (); }
bool crypt_get_encrypted_len (uint32_t *data_encrypted_len, uint32_t *padding_len) { if (padding_len) *padding_len = 0; return false; }
uint32_t crypt_max_padding_len (void) { return 0; }
void sections_add_to_list (VBlockP vb, SectionHeaderUnionP header) {}
uint32_t st_header_size (SectionType sec_type) { return sec_type == SEC_B250 || sec_type == SEC_LOCAL ? sizeof (SectionHeaderCtx) : sizeof (SectionHeader); }
LocalGetLineCB *zip_get_local_data_callback (DataType dt, ContextP ctx) { return NULL; }     /* contiguous data (the callback form only gathers lines, codec_htscodecs.c:51-64) */
static ContextP shim_zctx;
ContextP ctx_get_zctx_from_vctx (ConstContextP vctx, bool create_if_missing, bool follow_alias) { return shim_zctx; }

/* (b) BZ2 / BSC / LZMA as candidates: a payload of the scripted size */
static uint32_t shim_host_payload[3];
static bool shim_host_compress (int which, uint32_t *compressed_len)
{
    if (shim_host_payload[which] > *compressed_len) abort ();
    *compressed_len = shim_host_payload[which];
    return true;
}
COMPRESS (codec_bz2_compress)  { memset (compressed, 0, shim_host_payload[0]); return shim_host_compress (0, compressed_len); }
COMPRESS (codec_bsc_compress)  { memset (compressed, 0, shim_host_payload[1]); return shim_host_compress (1, compressed_len); }
COMPRESS (codec_lzma_compress) { memset (compressed, 0, shim_host_payload[2]); return shim_host_compress (2, compressed_len); }
uint32_t codec_bsc_est_size (Codec codec, uint64_t uncompressed_len) { return uncompressed_len + 1024; }

/* ---- (d) entry points --------------------------------------------------------------------------------------------------------- */

/* qsort (tests, n, sizeof (CodecTest), codec_assign_sorter) - src/codec.c:334 - on rows { codec, size, clock }. mode: 0 normal, 1 --best, 2 --fast */
void assignref_sort (int32_t *codec, float *size, float *clk, int n, int mode)
{
    CodecTest t[64];
    flag.best = mode == 1; flag.fast = mode == 2;
    for (int i = 0; i < n; i++) { t[i].codec = (Codec)codec[i]; t[i].size = size[i]; t[i].clock = clk[i]; }
    qsort (t, n, sizeof (CodecTest), codec_assign_sorter);
    for (int i = 0; i < n; i++) { codec[i] = t[i].codec; size[i] = t[i].size; clk[i] = t[i].clock; }
    flag.best = flag.fast = false;
}

/* codec_assign_best_codec (src/codec.c:234) for one context section.
 * in[]:  0 mode (0 normal, 1 --best, 2 --fast)  1 is_local (else b250)  2 vblock_i  3 the codec the segmenter left in the context
 *        4 zctx codec  5 zctx count  6 hard_coded  7 dt_props.vb_1_not_representative bits  8 is_last_vb_in_txt_file  9 flag.no_lzma
 *        10 vb is evb  11 .. 13 payload sizes of BZ2 / BSC / LZMA
 * txt_len / vb_size: vb->txt_data.len, segconf.vb_size (:352). ticks[12]: what clock() measures for trial k.
 * out[]: 0 the returned codec  1 the context's codec afterwards  2 zctx codec  3 zctx count  4 number of clock() calls (2 x trials run)
 * shown_text: the line --show-codec prints: the first FOUR rows of the sorted table as [name size clock] (:366-377) */
void assignref_run (const int32_t *in, const uint8_t *dict_id, uint64_t txt_len, uint64_t vb_size, const uint8_t *data, uint64_t len,
                    const int32_t *ticks, int32_t *out, char *shown_text /* 512 bytes or NULL: what --show-codec printed */)
{
    VBlockP vb = calloc (1, sizeof (VBlock));
    ContextP ctx = calloc (1, sizeof (Context)), zctx = calloc (1, sizeof (Context));
    const bool is_local = in[1];
    flag.best = in[0] == 1; flag.fast = in[0] == 2; flag.no_lzma = in[9];
    vb->data_type = DT_SAM; vb->vblock_i = in[2]; vb->txt_data.len = txt_len; vb->is_last_vb_in_txt_file = in[8];
    dt_props[DT_SAM].vb_1_not_representative = (uint8_t)in[7];
    segconf.vb_size = vb_size;
    evb = in[10] ? vb : NULL;
    memcpy (&ctx->dict_id, dict_id, 8); memcpy (&zctx->dict_id, dict_id, 8);
    ctx->ltype = LT_UINT8; ctx->no_callback = true;
    if (is_local) { ctx->lcodec = in[3]; zctx->lcodec = in[4]; zctx->lcodec_count = in[5]; ctx->lcodec_hard_coded = in[6]; }
    else          { ctx->bcodec = in[3]; zctx->bcodec = in[4]; zctx->bcodec_count = in[5]; }
    BufferP b = is_local ? &ctx->local : &ctx->b250;
    buf_alloc_do (vb, b, len + 8, 1, "data", __FUNCTION__, __LINE__);
    memcpy (b->data, data, len); b->len = len;
    for (int k = 0; k < 3; k++) shim_host_payload[k] = in[11 + k];
    char *shown = NULL; size_t shown_len = 0;
    FILE *was = info_stream;
    info_stream = open_memstream (&shown, &shown_len);      /* --show-codec prints the four best rows of the sorted table (:366-377) */
    flag.show_codec = true;
    shim_zctx = zctx; shim_now = 0; shim_clock_calls = 0; shim_n_ticks = 12;
    for (int k = 0; k < 12; k++) shim_ticks[k] = ticks[k];
    out[0] = codec_assign_best_codec (vb, ctx, NULL, is_local ? SEC_LOCAL : SEC_B250);
    out[1] = is_local ? ctx->lcodec : ctx->bcodec;
    out[2] = is_local ? zctx->lcodec : zctx->bcodec;
    out[3] = is_local ? zctx->lcodec_count : zctx->bcodec_count;
    out[4] = shim_clock_calls;
    fclose (info_stream); info_stream = was; flag.show_codec = false;
    if (shown_text) { snprintf (shown_text, 512, "%s", shown ? shown : ""); }
    free (shown);
    flag.best = flag.fast = false; flag.no_lzma = false; evb = NULL;
    free (b->memory); free (vb->z_data_test.memory);
    for (unsigned i = 0; i < NUM_CODEC_BUFS; i++) free (vb->codec_bufs[i].memory);
    free (ctx); free (zctx); free (vb);
}
