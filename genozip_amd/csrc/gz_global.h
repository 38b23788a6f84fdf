// gz_global.h -- SURVEY 8(f) N4: what follows the VBlocks in a .genozip file, HOST code (included by gz_host.cpp).
//
// Reference: zip_write_global_area src/zip.c:416-507 -> dict_io_compress_dictionaries src/dict_io.c:45-193 (SEC_DICT, one
// section per fragment of <= 1 MB whole words; dictionaries < 50 B stored raw, < 1 KB ARTB, else codec_assign_best_codec),
// ctx_compress_counts src/context.c:1612-1651 (SEC_COUNTS: big-endian u64 per word, protection bit cleared),
// sections_add_to_list src/sections.c:105-135 + sections_list_memory_to_file_format :481-534 (the section list: 19-byte
// entries, offsets and vblock_i as deltas, a dict_id only at its first appearance), SEC_GENOZIP_HEADER + footer
// src/sections.h:169-307 (layout: offsets probed from the reference's headers).
//
// The reference's own writer of SEC_GENOZIP_HEADER (zfile_compress_genozip_header, declared in src/zfile.h:19) is not among the shipped
// sources - it lives in the closed licence module. The header written here follows the struct (src/sections.h:169-300; every offset
// below is pinned to the reference's own headers by oracle/ref_hdr_shim.c -> tests/golden/hdr_golden.json) and what the reader does
// with it (zfile_read_genozip_header, src/zfile.c:889-1060); the licence fields (license_hash, lic_type) stay zero. The reference's
// shipped genounzip (15.0.86) accepts it and reconstructs the FASTQ text from files written here: tests/test_e2e_genounzip.py.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../include/genozip_amd.h"

struct GzSecEnt { uint64_t offset; uint32_t size; uint32_t vblock_i; uint32_t num_lines; uint8_t st, comp_i, flags; uint8_t dict_id[8]; };   // SectionEnt, sections.h:584-604
struct GzZFile {
    uint16_t data_type; uint32_t vb_size; uint8_t num_txt_files = 1;
    uint8_t paired = 0; uint32_t std_seq_len = 0, std_seq_len_r2 = 0;
    std::vector<GzSecEnt> list;
};

enum { GZ_SEC_GENOZIP_HEADER = 6, GZ_SEC_TXT_HEADER = 8, GZ_SEC_DICT = 10, GZ_SEC_COUNTS = 17, GZ_COMP_NONE = 255 };

extern "C" GzZFile *gz_zfile_create (uint16_t data_type, uint32_t vb_size_bytes)
{
    GzZFile *z = new GzZFile ();
    z->data_type = data_type; z->vb_size = vb_size_bytes;
    return z;
}
extern "C" void gz_zfile_destroy (GzZFile *z) { delete z; }

// sections_add_to_list for every section of a finished VBlock that was appended to the file at file_offset
// (zfile_output_processed_vb_ext src/zfile.c:1160-1178: the VBlock's list joins the file's in the order the VBlocks are written)
extern "C" int gz_zfile_add_vblock (GzZFile *zf, const uint8_t *z, uint64_t z_len, uint64_t file_offset, uint8_t comp_i, uint32_t num_lines)
{
    if (!zf || !z || z_len < 84 || gz_rd_be32 (z) != 0x27052012u || z[24] != GZ_SEC_VB_HEADER || gz_rd_be32 (z + 40) != z_len) return GZ_ERR_CORRUPT;
    const uint32_t vblock_i = gz_rd_be32 (z + 20);
    GzSecEnt e; memset (&e, 0, sizeof (e));
    e.offset = file_offset; e.size = 84; e.vblock_i = vblock_i; e.num_lines = num_lines; e.st = GZ_SEC_VB_HEADER; e.comp_i = comp_i; e.flags = z[27];
    zf->list.push_back (e);
    for (uint64_t at = 84; at < z_len;) {
        if (at + 40 > z_len || gz_rd_be32 (z + at) != 0x27052012u) return GZ_ERR_CORRUPT;
        const uint32_t clen = gz_rd_be32 (z + at + 12);
        if (at + 40 + clen > z_len) return GZ_ERR_CORRUPT;
        memset (&e, 0, sizeof (e));
        e.offset = file_offset + at; e.size = 40 + clen; e.vblock_i = vblock_i; e.st = z[at + 24]; e.comp_i = comp_i; e.flags = z[at + 27];
        memcpy (e.dict_id, z + at + 32, 8);
        zf->list.push_back (e);
        at += 40 + clen;
    }
    return GZ_OK;
}

static inline void gz_be64 (uint8_t *p, uint64_t v) { gz_be32 (p, (uint32_t)(v >> 32)); gz_be32 (p + 4, (uint32_t)v); }
static inline uint32_t gz_host_adler32 (const uint8_t *p, size_t n)
{
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < n; i++) { a = (a + p[i]) % 65521u; b = (b + a) % 65521u; }
    return (b << 16) | a;
}

extern "C" int gz_zfile_set_fastq (GzZFile *zf, uint8_t num_txt_files, uint8_t paired, uint32_t std_seq_len, uint32_t std_seq_len_r2)
{
    if (!zf || !num_txt_files) return GZ_ERR_ARG;
    zf->num_txt_files = num_txt_files; zf->paired = paired; zf->std_seq_len = std_seq_len; zf->std_seq_len_r2 = std_seq_len_r2;
    return GZ_OK;
}

// SEC_TXT_HEADER of a component without header text (FASTQ): txtheader_compress (src/txtheader.c:65-111: one fragment, vblock_i = 1, written
// even when empty :40-41; the codec of < 50 bytes is NONE) with the fields zfile_update_txt_header_section_header fills in when the component
// is through (src/zfile.c:1068-1105). 400 bytes (src/sections.h:308-327)
// with the component's header text (VCF: the ## lines and #CHROM; SAM: the @ lines) as the section's payload, stored (CODEC_NONE): the
// reference compresses it with whatever codec_assign_best_codec finds (txtheader_compress, src/txtheader.c:46-111); the reader takes any
extern "C" int gz_zfile_add_txt_header_text (GzZFile *zf, uint8_t comp_i, uint8_t pair, const char *txt_filename, uint64_t txt_data_size, uint64_t txt_num_lines,
                                             uint32_t max_lines_per_vb, const uint8_t *flav_prop, uint32_t n_flav_prop, uint64_t file_offset,
                                             const uint8_t *header_text, uint32_t header_len, uint8_t *out, uint64_t out_cap, uint64_t *out_len)
{
    if (!zf || !out_len || pair > 2 || n_flav_prop > 4 || (n_flav_prop && !flav_prop) || (header_len && !header_text)) return GZ_ERR_ARG;
    *out_len = (uint64_t)GZ_TXT_HEADER_LEN + header_len;
    if (!out || out_cap < *out_len) return GZ_TOO_SMALL;
    memset (out, 0, GZ_TXT_HEADER_LEN);
    gz_be32 (out, 0x27052012u); gz_be32 (out + 4, gz_host_adler32 (header_text, header_len)); gz_be32 (out + 12, header_len); gz_be32 (out + 16, header_len); gz_be32 (out + 20, 1);
    out[24] = GZ_SEC_TXT_HEADER; out[25] = GZ_CODEC_NONE; out[27] = pair;                          // FlagsTxtHeader.pair (fastq_zip_set_txt_header_flags)
    gz_be64 (out + 28, txt_data_size); gz_be64 (out + 36, txt_num_lines); gz_be32 (out + 44, max_lines_per_vb);
    out[48] = GZ_CODEC_NONE;                                                                       // src_codec: plain text
    if (txt_filename) strncpy ((char *)out + 84, txt_filename, 255);
    gz_be64 (out + 340, header_len);                                                               // txt_header_size
    for (uint32_t q = 0; q < n_flav_prop; q++) { out[348 + 2 * q] = flav_prop[2 * q]; out[349 + 2 * q] = flav_prop[2 * q + 1]; }
    if (header_len) memcpy (out + GZ_TXT_HEADER_LEN, header_text, header_len);
    GzSecEnt e; memset (&e, 0, sizeof (e));
    e.offset = file_offset; e.size = (uint32_t)*out_len; e.vblock_i = 1; e.st = GZ_SEC_TXT_HEADER; e.comp_i = comp_i; e.flags = pair;
    zf->list.push_back (e);
    return GZ_OK;
}
extern "C" int gz_zfile_add_txt_header (GzZFile *zf, uint8_t comp_i, uint8_t pair, const char *txt_filename, uint64_t txt_data_size, uint64_t txt_num_lines,
                                        uint32_t max_lines_per_vb, const uint8_t *flav_prop, uint32_t n_flav_prop, uint64_t file_offset, uint8_t *out)
{
    uint64_t n = 0;
    return gz_zfile_add_txt_header_text (zf, comp_i, pair, txt_filename, txt_data_size, txt_num_lines, max_lines_per_vb, flav_prop, n_flav_prop, file_offset, NULL, 0, out, out ? GZ_TXT_HEADER_LEN : 0, &n);
}

// a section made on the host (NONREF: the sub-codec of CODEC_ACGT is the host's) into a finished VBlock
extern "C" int gz_vb_insert_section (const uint8_t *z, uint64_t z_len, uint32_t index, const uint8_t dict_id[8], uint8_t codec, uint8_t sub_codec,
                                     uint8_t flags, uint8_t ltype, uint8_t param, const uint8_t *payload, uint32_t payload_len, uint32_t uncompressed_len,
                                     uint8_t *out, uint64_t out_cap, uint64_t *out_len)
{
    if (!z || !dict_id || (payload_len && !payload) || !out_len || z_len < 84 || gz_rd_be32 (z) != 0x27052012u || z[24] != GZ_SEC_VB_HEADER || gz_rd_be32 (z + 40) != z_len) return GZ_ERR_ARG;
    uint64_t at = 84;
    for (uint32_t k = 0; k < index; k++) {
        if (at + 40 > z_len || gz_rd_be32 (z + at) != 0x27052012u) return GZ_ERR_ARG;            // (fewer sections than `index`)
        at += 40 + (uint64_t)gz_rd_be32 (z + at + 12);
    }
    if (at > z_len) return GZ_ERR_CORRUPT;
    const uint64_t total = z_len + 40 + payload_len;
    *out_len = total;
    if (total > 0xffffffffull) return GZ_ERR_ARG;
    if (!out || out_cap < total) return GZ_TOO_SMALL;
    memcpy (out, z, at);
    uint8_t *h = out + at;
    memset (h, 0, 40);
    gz_be32 (h, 0x27052012u); gz_be32 (h + 4, gz_host_adler32 (payload, payload_len)); gz_be32 (h + 12, payload_len); gz_be32 (h + 16, uncompressed_len);
    memcpy (h + 20, z + 20, 4);                                                                    // vblock_i
    h[24] = GZ_SEC_LOCAL; h[25] = codec; h[26] = sub_codec; h[27] = flags; h[28] = ltype; h[29] = param;
    memcpy (h + 32, dict_id, 8);
    if (payload_len) memcpy (h + 40, payload, payload_len);
    memcpy (h + 40 + payload_len, z + at, z_len - at);
    gz_be32 (out + 40, (uint32_t)total);                                                           // z_data_bytes (zfile.c:1144)
    return GZ_OK;
}
static inline uint32_t gz_zigzag32 (int32_t n) { return n < 0 ? ((uint32_t)(-(int64_t)n) << 1) - 1 : (uint32_t)n << 1; }   // INTERLACE, context.h:99-100

// comp_compress for a global section whose header is hdr_len bytes (28 common + specific): payload through the codec (host form)
static int global_section (GzHandle *h, std::vector<uint8_t> &out, uint8_t *hdr, uint32_t hdr_len, int codec, const uint8_t *data, uint32_t len)
{
    if (len < 50) codec = GZ_CODEC_NONE;                                   // compressor.c:56-58
    std::vector<uint8_t> pay (gz_codec_est_size (codec, len) + 16);
    uint32_t plen = (uint32_t)pay.size ();
    if (codec == GZ_CODEC_NONE) { if (len) memcpy (pay.data (), data, len); plen = len; }
    else { const int rc = gz_codec_compress_host (h, codec, data, len, pay.data (), &plen, 0); if (rc != GZ_OK) return rc; }
    gz_be32 (hdr + 0, 0x27052012u);
    gz_be32 (hdr + 4, gz_host_adler32 (pay.data (), plen));                // z_digest = adler32 (1, payload) (compressor.c:161)
    gz_be32 (hdr + 8, 0); gz_be32 (hdr + 12, plen); gz_be32 (hdr + 16, len);
    hdr[25] = (uint8_t)codec;
    out.insert (out.end (), hdr, hdr + hdr_len);
    out.insert (out.end (), pay.begin (), pay.begin () + plen);
    return GZ_OK;
}

// zip_write_global_area for the contexts given: dictionaries, counts (of the contexts flagged), the genozip header with the
// section list as its payload, the footer. file_offset = where the global area starts in the file. out: host buffer.
extern "C" int gz_zfile_write_global_area (GzZFile *zf, GzHandle *h, GzZctx *const *zctx, const uint8_t *dict_ids /* n x 8 */, const uint8_t *counts_section /* n, or NULL */,
                                           uint32_t n_ctx, uint64_t file_offset, uint64_t recon_size, uint64_t num_lines, const char *created,
                                           uint8_t *out_host, uint64_t out_cap, uint64_t *out_len)
{
    if (!zf || !h || (n_ctx && (!zctx || !dict_ids)) || !out_host || !out_len) return GZ_ERR_ARG;
    std::vector<uint8_t> out;
    int rc;
    // (the entries of the global area join zf->list as they are written: taken back on every return that is not GZ_OK, so that the call
    //  can be repeated - e.g. with a larger buffer after GZ_TOO_SMALL - on the same GzZFile)
    struct Undo { std::vector<GzSecEnt> &l; size_t n; bool keep; ~Undo () { if (!keep) l.resize (n); } } undo { zf->list, zf->list.size (), false };
    // ---- SEC_DICT (dict_io.c:45-193): contexts in order, fragments of whole words below 1 MB (or twice the longest word)
    uint32_t frag_i = 0;
    for (uint32_t c = 0; c < n_ctx; c++) {
        GzZctxView v;
        if ((rc = gz_zctx_view (zctx[c], &v)) != GZ_OK) return rc;
        if (!v.n_words || v.rm_dict_all_the_same) continue;               // dict_io.c:84-91
        const int codec = v.dict_len < 50 ? GZ_CODEC_NONE : v.dict_len < 1024 ? GZ_CODEC_ARTB : gz_codec_assign_best_host (h, v.dict, (uint32_t)(v.dict_len < 99999 ? v.dict_len : 99999));
        uint32_t frag_size = 1u << 20;
        for (uint32_t w = 0; w < v.n_words; w++) if (v.snip_len[w] * 2 > frag_size) { frag_size = 2; while (frag_size < 2 * v.snip_len[w]) frag_size <<= 1; }   // dict_io.c:101-106
        for (uint32_t w = 0; w < v.n_words;) {
            const uint32_t w0 = w; uint32_t len = 0;
            while (w < v.n_words && len + v.snip_len[w] + 1 < frag_size) { len += v.snip_len[w] + 1; w++; }   // dict_io.c:116-122
            if (w == w0) return GZ_ERR;
            uint8_t hd[40]; memset (hd, 0, sizeof (hd));
            hd[24] = GZ_SEC_DICT; gz_be32 (hd + 28, w - w0); memcpy (hd + 32, dict_ids + 8 * (size_t)c, 8);
            // vblock_i: the fragment's place in the fan-out over all dictionaries (dict_io.c:146); flags: zctx->dict_flags - the word
            // every VBlock that dropped its all-the-same b250 reconstructs (FlagsDict.all_the_same_wi, bits 2-5; context.c:846-849)
            gz_be32 (hd + 20, ++frag_i);
            hd[27] = v.all_the_same_wi >= 0 ? (uint8_t)((v.all_the_same_wi & 15) << 2) : 0;
            GzSecEnt e; memset (&e, 0, sizeof (e));
            e.offset = file_offset + out.size (); e.st = GZ_SEC_DICT; e.comp_i = GZ_COMP_NONE; memcpy (e.dict_id, dict_ids + 8 * (size_t)c, 8);
            e.vblock_i = frag_i; e.flags = hd[27];
            if ((rc = global_section (h, out, hd, 40, codec ? codec : GZ_CODEC_ARTB, v.dict + v.char_index[w0], len)) != GZ_OK) return rc;
            e.size = (uint32_t)(file_offset + out.size () - e.offset);
            zf->list.push_back (e);
        }
    }
    // ---- SEC_COUNTS (context.c:1612-1651)
    for (uint32_t c = 0; c < n_ctx; c++) {
        if (!counts_section || !counts_section[c]) continue;
        GzZctxView v;
        if ((rc = gz_zctx_view (zctx[c], &v)) != GZ_OK) return rc;
        if (!v.n_words) continue;
        std::vector<uint8_t> be ((size_t)v.n_words * 8);
        for (uint32_t w = 0; w < v.n_words; w++) gz_be64 (be.data () + 8 * (size_t)w, v.counts[w] & ~0x8000000000000000ull);
        const int codec = gz_codec_assign_best_host (h, be.data (), (uint32_t)(be.size () < 99999 ? be.size () : 99999));
        uint8_t hd[44]; memset (hd, 0, sizeof (hd));
        hd[24] = GZ_SEC_COUNTS; memcpy (hd + 36, dict_ids + 8 * (size_t)c, 8);
        GzSecEnt e; memset (&e, 0, sizeof (e));
        e.offset = file_offset + out.size (); e.st = GZ_SEC_COUNTS; e.comp_i = GZ_COMP_NONE; memcpy (e.dict_id, dict_ids + 8 * (size_t)c, 8);
        if ((rc = global_section (h, out, hd, 44, codec ? codec : GZ_CODEC_RANB, be.data (), (uint32_t)be.size ())) != GZ_OK) return rc;
        e.size = (uint32_t)(file_offset + out.size () - e.offset);
        zf->list.push_back (e);
    }
    // ---- the section list in file format (sections.c:481-534), the genozip header itself being its last entry
    GzSecEnt g; memset (&g, 0, sizeof (g));
    g.offset = file_offset + out.size (); g.st = GZ_SEC_GENOZIP_HEADER; g.comp_i = GZ_COMP_NONE;
    zf->list.push_back (g);
    std::vector<uint8_t> fl (zf->list.size () * 19, 0);
    {
        uint64_t prev_off = 0; uint32_t prev_vb = 0, prev_lines = 0; int prev_comp = -1;
        std::vector<std::pair<uint64_t, uint32_t>> first;                  // dict_id -> first section index
        for (size_t i = 0; i < zf->list.size (); i++) {
            const GzSecEnt &s = zf->list[i];
            uint8_t *f = fl.data () + 19 * i;
            if (s.offset < prev_off || s.offset - prev_off > 0xffffffffull) return GZ_ERR;
            gz_be32 (f + 0, (uint32_t)(s.offset - prev_off));
            gz_be32 (f + 4, gz_zigzag32 ((int32_t)s.vblock_i - (int32_t)prev_vb));
            f[8] = (i && (int)s.comp_i == prev_comp) ? 0 : s.comp_i == GZ_COMP_NONE ? GZ_COMP_NONE : (uint8_t)(1 + s.comp_i);
            f[9] = s.st; f[18] = s.flags;
            const bool dicted = s.st == GZ_SEC_DICT || s.st == GZ_SEC_B250 || s.st == GZ_SEC_LOCAL || s.st == GZ_SEC_COUNTS;
            if (dicted) {
                uint64_t id; memcpy (&id, s.dict_id, 8);
                size_t k = 0;
                while (k < first.size () && first[k].first != id) k++;
                if (k == first.size ()) { memcpy (f + 10, s.dict_id, 8); first.push_back ({ id, (uint32_t)i }); }
                else { f[10] = 0; gz_be32 (f + 14, first[k].second); }   // is_dict_id = 0: copy from that section
            }
            else if (s.st == GZ_SEC_VB_HEADER) { gz_be32 (f + 10, gz_zigzag32 ((int32_t)s.num_lines - (int32_t)prev_lines)); prev_lines = s.num_lines; }
            prev_off = s.offset; prev_vb = s.vblock_i; prev_comp = s.comp_i;
        }
    }
    // ---- SEC_GENOZIP_HEADER (720-byte header, sections.h:169-300) + payload + footer (sections.h:303-307)
    uint8_t gh[720]; memset (gh, 0, sizeof (gh));
    gh[24] = GZ_SEC_GENOZIP_HEADER;
    gh[27] = zf->paired ? 1 : 0;                                          // FlagsGenozipHeader: dt_specific = the file holds an R1 / R2 pair; no digest, no aligner
    gh[28] = 15;                                                          // genozip_version (format parity: 15.0.86)
    gh[30] = (uint8_t)(zf->data_type >> 8); gh[31] = (uint8_t)zf->data_type;
    gz_be64 (gh + 32, recon_size);
    // minor version : 14, is_modified : 1, private_file : 1, num_lines_bound : 48 - one little-endian 64-bit word (sections.h:174-177). The
    // reader takes BGEN64 of the 48-bit field (zfile.c:965, sections_show.c:389), so what the field holds is the low 48 bits of the byte-
    // swapped count: counts below 65 536 read back as 0 (only progress display uses the number)
    { const uint64_t sw = __builtin_bswap64 (num_lines) & 0xffffffffffffull;
      const uint64_t bits = (86ull & 0x3fff) | (sw << 16); memcpy (gh + 40, &bits, 8); }
    gz_be32 (gh + 48, (uint32_t)zf->list.size ());
    gh[55] = zf->num_txt_files;
    if (created) strncpy ((char *)gh + 88, created, 71);
    gz_be32 (gh + 460, zf->std_seq_len); gz_be32 (gh + 464, zf->std_seq_len_r2);   // fastq.segconf_std_seq_len / _lR2 (sections.h:248-257)
    gz_be32 (gh + 715, zf->vb_size);
    if ((rc = global_section (h, out, gh, 720, GZ_CODEC_NONE, fl.data (), (uint32_t)fl.size ())) != GZ_OK) return rc;
    zf->list.back ().size = (uint32_t)(file_offset + out.size () - g.offset);
    uint8_t foot[12];
    gz_be64 (foot, g.offset); gz_be32 (foot + 8, 0x27052012u);
    out.insert (out.end (), foot, foot + 12);
    *out_len = out.size ();
    if (out.size () > out_cap) return GZ_TOO_SMALL;
    memcpy (out_host, out.data (), out.size ());
    undo.keep = true;
    return GZ_OK;
}
