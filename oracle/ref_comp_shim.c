/* oracle/ref_comp_shim.c -- TEST INFRASTRUCTURE ONLY (see oracle/Makefile, target "ref").
 *
 * Compiled together with the reference's own src/compressor.c (comp_compress: SURVEY 8a row a10), src/codec_none.c, the vendored
 * libdeflate's adler32.c (the z_digest of every section, compressor.c:161) and the vendored htscodecs, all WHERE THEY LIE under
 * /root/reference (never copied), into oracle/_ref/libcompref.so against the reference's own headers. What comp_compress does to
 * a section - "under 50 bytes a simple codec becomes NONE" (:56-58), the call of the codec, data_compressed_len /
 * data_uncompressed_len in big endian, the adler32 of the payload, header + payload appended to z_data - is then the reference's
 * own code running, and tests/golden/ctx_golden.json holds sections it made (tests/golden/make_ctx_golden.py).
 *
 * This file supplies what compressor.o imports from parts of the reference that do not build outside its tree: allocation of a
 * Buffer, the option struct (all zero = defaults, no encryption), the section list (not kept), the header size of a context
 * section (sections.h: sizeof (SectionHeaderCtx)), and the codec table with the eight htscodecs entries as src/codec_htscodecs.c:26-123
 * fills them (that file itself needs the build's generated profiler fields): est_size = 1 KB + the coder's bound, compress = the
 * coder with the order of the codec. The SectionHeaderCtx handed to comp_compress is assembled here from plain arguments (what
 * zfile_compress_local_data / _b250_data put into it, src/zfile.c:288-364, is row a9 and stays unpinned). Nothing here is product code.
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
#include "file.h"
#include "codec.h"
#include "crypt.h"
#include "zfile.h"
#include "sections.h"
#include "compressor.h"
#include "htscodecs/rANS_static4x16.h"
#include "htscodecs/arith_dynamic.h"

Flags flag;
VBlockP evb;
FileP z_file;
__attribute__((constructor)) static void shim_defaults (void) { flag.show_time_comp_i = COMP_NONE; flag.command = ZIP; }

void buf_alloc_do (VBlockP vb, BufferP buf, uint64_t requested_size, float grow_at_least_factor, rom name, FUNCLINE)
{
    if (buf->size >= requested_size && buf->data) return;
    uint64_t sz = requested_size * (grow_at_least_factor > 1 ? grow_at_least_factor : 1) + 64;
    char *m = realloc (buf->memory, sz + 16);
    if (!m) abort ();
    buf->memory = m; buf->data = m + 8; buf->size = sz; buf->vb = vb; buf->name = name; buf->type = BUF_REGULAR;
}
void buf_free_do (BufferP buf, FUNCLINE) { buf->len = 0; buf->param = 0; }
void buf_copy_do (VBlockP vb, BufferP dst, ConstBufferP src, uint64_t bytes_per_entry, uint64_t src_start_entry, uint64_t max_entries, FUNCLINE, rom name) { abort (); }
void warn (rom fmt, ...) {}
rom report_support (void) { return ""; }
StrText vb_name (VBlockP vb) { StrText s = {}; return s; }
rom codec_name (Codec codec) { return "codec"; }
rom st_name (SectionType st) { return "section"; }
void codec_verify_free_all (VBlockP vb, rom op, Codec codec) {}
void codec_free_all (VBlockP vb) {}
Codec codec_assign_best_codec (VBlockP vb, ContextP ctx, BufferP data, SectionType st) { return CODEC_UNKNOWN; }
/* no encryption (crypt.c: all of these answer "not encrypted") */
bool crypt_get_encrypted_len (uint32_t *data_encrypted_len, uint32_t *padding_len) { if (padding_len) *padding_len = 0; return false; }
uint32_t crypt_max_padding_len (void) { return 0; }
void crypt_pad (uint8_t *data, uint32_t data_len, uint32_t padding_len) {}
void crypt_do (VBlockP vb, uint8_t *data, uint32_t data_len, VBIType vb_i, SectionType sec_type, bool is_header) {}
void sections_add_to_list (VBlockP vb, SectionHeaderUnionP header) {}
void sections_show_header (SectionHeaderUnionP header, VBlockP vb, CompIType comp_i, uint64_t offset, char rw) {}
void zfile_output_processed_vb (VBlockP vb) {}
uint32_t st_header_size (SectionType sec_type) { return sizeof (SectionHeaderCtx); }   /* SEC_B250 / SEC_LOCAL (sections.c abouts table) */

/* src/codec_htscodecs.c:17-33,51-123 for contiguous data: the order of each codec, 1 KB + the coder's own bound, the coder */
static const int hts_order[NUM_CODECS] = { [CODEC_RANB] = 0x01, [CODEC_RANW] = 0x19, [CODEC_RANb] = 0x81, [CODEC_RANw] = 0x99,
                                           [CODEC_ARTB] = 0x01, [CODEC_ARTW] = 0x19, [CODEC_ARTb] = 0x81, [CODEC_ARTw] = 0x99 };
static uint32_t shim_rans_est  (Codec codec, uint64_t len) { return 1024 + rans_compress_bound_4x16 (len, hts_order[codec]); }
static uint32_t shim_arith_est (Codec codec, uint64_t len) { return 1024 + arith_compress_bound (len, hts_order[codec]); }
static COMPRESS (shim_rans)  { return !!rans_compress_to_4x16 (vb, (uint8_t *)uncompressed, *uncompressed_len, (uint8_t *)compressed, compressed_len, hts_order[header->codec]); }
static COMPRESS (shim_arith) { return !!arith_compress_to (vb, (uint8_t *)uncompressed, *uncompressed_len, (uint8_t *)compressed, compressed_len, hts_order[header->codec]); }
extern COMPRESS (codec_none_compress);
extern uint32_t codec_none_est_size (Codec codec, uint64_t uncompressed_len);
#define HTS(c, f, e) [CODEC_##c] = { .is_simple = true, .name = #c, .compress = f, .est_size = e }
CodecArgs codec_args[NUM_CODECS] = { [CODEC_NONE] = { .is_simple = true, .name = "NONE", .compress = codec_none_compress, .est_size = codec_none_est_size },
    HTS (RANB, shim_rans, shim_rans_est), HTS (RANW, shim_rans, shim_rans_est), HTS (RANb, shim_rans, shim_rans_est), HTS (RANw, shim_rans, shim_rans_est),
    HTS (ARTB, shim_arith, shim_arith_est), HTS (ARTW, shim_arith, shim_arith_est), HTS (ARTb, shim_arith, shim_arith_est), HTS (ARTw, shim_arith, shim_arith_est) };

/* comp_compress of one context section; byte30 = nothing_char / b250_size. Returns the bytes appended to z_data (header + payload) */
long compref_section (int section_type, int codec, int sub_codec, int flags, int ltype, int param, int byte30, const uint8_t *dict_id,
                      uint32_t vblock_i, const uint8_t *data, uint32_t len, uint8_t *out, uint64_t out_cap)
{
    VBlockP vb = calloc (1, sizeof (VBlock));
    vb->vblock_i = vblock_i;
    SectionHeaderCtx h = { .magic = BGEN32 (GENOZIP_MAGIC), .section_type = section_type, .data_uncompressed_len = BGEN32 (len), .codec = codec,
                           .sub_codec = sub_codec, .vblock_i = BGEN32 (vblock_i), .ltype = ltype, .param = param };
    memcpy (&h.flags, &flags, 1);
    memcpy (&h.dict_id, dict_id, 8);
    ((uint8_t *)&h)[30] = (uint8_t)byte30;
    Buffer z = {};
    comp_compress (vb, NULL, &z, (SectionHeaderP)&h, (rom)data, NULL, "section");
    long n = (long)z.len;
    if ((uint64_t)n > out_cap) n = -1; else memcpy (out, z.data, n);
    free (z.memory); free (vb);
    return n;
}
