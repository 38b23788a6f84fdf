/* oracle/ref_hdr_shim.c -- TEST INFRASTRUCTURE ONLY (see oracle/Makefile, target "ref").
 *
 * Compiled against the reference's OWN headers (src/sections.h, src/container.h, where they lie under /root/reference) into
 * oracle/_ref/libhdrref.so. It holds no algorithm: every function below fills one of the reference's on-disk structs THROUGH ITS
 * STRUCT MEMBERS with the values handed in and copies out the raw bytes, so that tests/golden/hdr_golden.json pins every field
 * offset, width and endianness of the layouts rows a9 / a16 / N4 write: SectionHeader, SectionHeaderCtx, SectionHeaderVbHeader,
 * SectionHeaderDictionary, SectionHeaderCounts, SectionHeaderTxtHeader, SectionHeaderGenozipHeader, SectionFooterGenozipHeader,
 * SectionEntFileFormat, Container / ContainerItem. Big-endian conversion is the caller's business in the reference (BGEN32 at
 * every assignment, e.g. src/zfile.c:1110-1120): done here with the reference's own BGEN macros (endianness.h). Nothing here is
 * product code.
 */
#include <string.h>
#include <stddef.h>
#include "genozip.h"
#include "sections.h"
#include "container.h"
#include "endianness.h"

#define OUT(h) do { memcpy (out, &(h), sizeof (h)); return (int)sizeof (h); } while (0)

int hdrref_ctx (uint32_t z_digest, uint32_t clen, uint32_t ulen, uint32_t vblock_i, int st, int codec, int sub_codec, int flags, int ltype, int param,
                int b250_size_or_nothing, const uint8_t *dict_id, uint8_t *out)
{
    SectionHeaderCtx h; memset (&h, 0, sizeof (h));
    h.magic = BGEN32 (GENOZIP_MAGIC); h.z_digest = BGEN32 (z_digest); h.data_compressed_len = BGEN32 (clen); h.data_uncompressed_len = BGEN32 (ulen);
    h.vblock_i = BGEN32 (vblock_i); h.section_type = st; h.codec = codec; h.sub_codec = sub_codec; h.flags.flags = (uint8_t)flags;
    h.ltype = ltype; h.param = (uint8_t)param;
    if (st == SEC_B250) h.b250_size = b250_size_or_nothing; else h.nothing_char = (char)b250_size_or_nothing;
    memcpy (h.dict_id.id, dict_id, 8);
    OUT (h);
}

int hdrref_vb (uint32_t vblock_i, uint32_t recon_size, uint32_t z_data_bytes, uint32_t longest_line_len, uint32_t longest_seq_len, int flags, uint8_t *out)
{
    SectionHeaderVbHeader h; memset (&h, 0, sizeof (h));
    h.magic = BGEN32 (GENOZIP_MAGIC); h.section_type = SEC_VB_HEADER; h.vblock_i = BGEN32 (vblock_i); h.codec = CODEC_NONE; h.flags.flags = (uint8_t)flags;
    h.z_digest = BGEN32 (1);                         /* comp_compress: adler32 of the (empty) payload (compressor.c:161) */
    h.recon_size = BGEN32 (recon_size); h.z_data_bytes = BGEN32 (z_data_bytes); h.longest_line_len = BGEN32 (longest_line_len); h.longest_seq_len = BGEN32 (longest_seq_len);
    OUT (h);
}

int hdrref_dict (uint32_t z_digest, uint32_t clen, uint32_t ulen, int codec, uint32_t num_snips, int all_the_same_wi, const uint8_t *dict_id, uint8_t *out)
{
    SectionHeaderDictionary h; memset (&h, 0, sizeof (h));
    h.magic = BGEN32 (GENOZIP_MAGIC); h.z_digest = BGEN32 (z_digest); h.data_compressed_len = BGEN32 (clen); h.data_uncompressed_len = BGEN32 (ulen);
    h.section_type = SEC_DICT; h.codec = codec; h.num_snips = BGEN32 (num_snips); h.flags.dictionary.all_the_same_wi = all_the_same_wi;
    memcpy (h.dict_id.id, dict_id, 8);
    OUT (h);
}

int hdrref_counts (uint32_t z_digest, uint32_t clen, uint32_t ulen, int codec, int64_t nodes_param, const uint8_t *dict_id, uint8_t *out)
{
    SectionHeaderCounts h; memset (&h, 0, sizeof (h));
    h.magic = BGEN32 (GENOZIP_MAGIC); h.z_digest = BGEN32 (z_digest); h.data_compressed_len = BGEN32 (clen); h.data_uncompressed_len = BGEN32 (ulen);
    h.section_type = SEC_COUNTS; h.codec = codec; h.nodes_param = (int64_t)BGEN64 ((uint64_t)nodes_param);
    memcpy (h.dict_id.id, dict_id, 8);
    OUT (h);
}

int hdrref_txt (uint32_t vblock_i, int codec, int pair, uint64_t txt_data_size, uint64_t txt_num_lines, uint32_t max_lines_per_vb, int src_codec,
                const char *txt_filename, uint64_t txt_header_size, const uint8_t *flav_prop /* NUM_QTYPES x 2 */, uint8_t *out)
{
    SectionHeaderTxtHeader h; memset (&h, 0, sizeof (h));
    h.magic = BGEN32 (GENOZIP_MAGIC); h.z_digest = BGEN32 (1) /* adler32 of no payload */; h.section_type = SEC_TXT_HEADER; h.codec = codec; h.vblock_i = BGEN32 (vblock_i);
    h.flags.txt_header.pair = pair;
    h.txt_data_size = BGEN64 (txt_data_size); h.txt_num_lines = BGEN64 (txt_num_lines); h.max_lines_per_vb = BGEN32 (max_lines_per_vb); h.src_codec = src_codec;
    strncpy (h.txt_filename, txt_filename, TXT_FILENAME_LEN - 1);
    h.txt_header_size = BGEN64 (txt_header_size);
    memcpy (h.flav_prop, flav_prop, sizeof (h.flav_prop));
    OUT (h);
}

int hdrref_flav_prop (int has_seq_len, int is_consensus, int is_mated, int cnn, int is_tokenized, uint8_t *out)
{
    QnameFlavorProp p; memset (&p, 0, sizeof (p));
    p.has_seq_len = has_seq_len; p.is_consensus = is_consensus; p.is_mated = is_mated; p.cnn = cnn; p.is_tokenized = is_tokenized;
    OUT (p);
}

int hdrref_genozip (uint32_t z_digest, uint32_t clen, uint32_t ulen, int flags, int version, int minor, uint16_t data_type, uint64_t recon_size, uint64_t num_lines_bound_field,
                    uint32_t num_sections, int num_txt_files, const char *created, uint32_t std_seq_len, uint32_t std_seq_lR2, uint32_t segconf_vb_size, int lic_type, uint8_t *out)
{
    SectionHeaderGenozipHeader h; memset (&h, 0, sizeof (h));
    h.magic = BGEN32 (GENOZIP_MAGIC); h.z_digest = BGEN32 (z_digest); h.data_compressed_len = BGEN32 (clen); h.data_uncompressed_len = BGEN32 (ulen);
    h.section_type = SEC_GENOZIP_HEADER; h.codec = CODEC_NONE; h.flags.flags = (uint8_t)flags;
    h.genozip_version = version; h.genozip_minor_ver = minor; h.data_type = BGEN16 (data_type); h.recon_size = BGEN64 (recon_size);
    h.num_lines_bound = num_lines_bound_field;       /* the 48-bit field as given: see tests - the reader applies BGEN64 to it (zfile.c:965) */
    h.num_sections = BGEN32 (num_sections); h.num_txt_files = num_txt_files;
    strncpy (h.created, created, FILE_METADATA_LEN - 1);
    h.fastq.segconf_std_seq_len = BGEN32 (std_seq_len); h.fastq.segconf_std_seq_lR2 = BGEN32 (std_seq_lR2);
    h.segconf_vb_size = BGEN32 (segconf_vb_size); h.lic_type = lic_type;
    OUT (h);
}

int hdrref_footer (uint64_t offset, uint8_t *out)
{
    SectionFooterGenozipHeader f; memset (&f, 0, sizeof (f));
    f.genozip_header_offset = BGEN64 (offset); f.magic = BGEN32 (GENOZIP_MAGIC);
    OUT (f);
}

/* one entry of the section list in file format: the caller gives the (already delta / zig-zag coded) values of the fields */
int hdrref_secent (uint32_t offset_delta, uint32_t vblock_i_delta, int comp_i_plus_1, int st, const uint8_t *dict_id, int is_dict_id,
                   uint32_t dict_sec_i, uint32_t num_lines_delta, int flags, uint8_t *out)
{
    SectionEntFileFormat e; memset (&e, 0, sizeof (e));
    e.offset_delta = BGEN32 (offset_delta); e.vblock_i_delta = BGEN32 (vblock_i_delta); e.comp_i_plus_1 = comp_i_plus_1; e.st = st; e.flags.flags = (uint8_t)flags;
    if (st == SEC_VB_HEADER) e.num_lines = BGEN32 (num_lines_delta);
    else if (is_dict_id == 1) memcpy (e.dict_id.id, dict_id, 8);
    else if (is_dict_id == 0) { e.is_dict_id = 0; e.dict_sec_i = BGEN32 (dict_sec_i); }
    OUT (e);
}

/* a Container of n items: items = n x (dict_id[8], sep0, sep1); -> the binary struct as container_prepare_snip base64s it */
int hdrref_container (uint32_t n_items, uint32_t repeats, int con_flags /* bit0 drop_final_item_sep_of_final_repeat ... bit7 drop_final_item_sep */,
                      int repsep0, int repsep1, const uint8_t *items, uint8_t *out)
{
    Container_MAX_FIELDS c; memset (&c, 0, sizeof (c));
    c.repeats = repeats; c.nitems_lo = n_items & 0xff; c.nitems_hi = n_items >> 8;
    c.drop_final_item_sep_of_final_repeat = con_flags & 1; c.drop_final_repsep = (con_flags >> 1) & 1; c.filter_repeats = (con_flags >> 2) & 1;
    c.filter_items = (con_flags >> 3) & 1; c.is_toplevel = (con_flags >> 4) & 1; c.keep_empty_item_sep = (con_flags >> 5) & 1;
    c.callback = (con_flags >> 6) & 1; c.drop_final_item_sep = (con_flags >> 7) & 1;
    c.repsep[0] = (char)repsep0; c.repsep[1] = (char)repsep1;
    for (uint32_t i = 0; i < n_items; i++) {
        memcpy (c.items[i].dict_id.id, items + 10 * i, 8);
        c.items[i].separator[0] = items[10 * i + 8]; c.items[i].separator[1] = items[10 * i + 9];
    }
    const size_t sz = con_sizeof_ (n_items);
    memcpy (out, &c, sz);
    return (int)sz;
}

int hdrref_sizes (uint32_t *out)
{
    out[0] = sizeof (SectionHeader); out[1] = sizeof (SectionHeaderCtx); out[2] = sizeof (SectionHeaderVbHeader); out[3] = sizeof (SectionHeaderDictionary);
    out[4] = sizeof (SectionHeaderCounts); out[5] = sizeof (SectionHeaderTxtHeader); out[6] = sizeof (SectionHeaderGenozipHeader);
    out[7] = sizeof (SectionFooterGenozipHeader); out[8] = sizeof (SectionEntFileFormat); out[9] = sizeof (ContainerItem); out[10] = sizeof (Container_0);
    out[11] = SEC_TXT_HEADER; out[12] = SEC_VB_HEADER; out[13] = SEC_DICT; out[14] = SEC_B250; out[15] = SEC_LOCAL; out[16] = SEC_COUNTS; out[17] = SEC_GENOZIP_HEADER;
    out[18] = DT_FASTQ; out[19] = NUM_QTYPES; out[20] = CODEC_ACGT; out[21] = CODEC_LZMA; out[22] = CODEC_XCGT; out[23] = LT_BLOB; out[24] = LT_CODEC; out[25] = LT_SUPP;
    return 26;
}
