/* genozip_amd.h -- C-ABI of libgenozip_amd.so: the MI355X (gfx950) implementation of Genozip's context
 * entropy-coding hot path. Plain C, pointers and sizes only; no torch / C++ types.
 *
 * Every entry point names the interface of the reference (divonlan/genozip 15.0.86, paths relative to
 * /root/reference) that it stands in for. INTEGRATION.md shows the reference-side stub that would bind them.
 *
 * Memory spaces: names ending in _host take host pointers (the library stages through HBM itself);
 * everything else takes DEVICE pointers (hipMalloc'd, e.g. torch tensors' data_ptr()) and is asynchronous on the
 * handle's HIP stream until gz_sync() / a *_host call.
 *
 * Determinism contract: for the simple codecs the payload is a pure function of
 * (codec id, input bytes) and is byte-identical to the reference's, provided the output capacity is at least the
 * coder's own bound (rans_compress_bound_4x16 / arith_compress_bound = gz_codec_est_size() - 1 KB; below that the
 * reference's coders report "too small", src/htscodecs/rANS_static4x16pr.c:1158, arith_dynamic.c:622 - so does this
 * library). Genozip itself always hands over gz_codec_est_size() bytes (src/compressor.c:63-67).
 */
#ifndef GENOZIP_AMD_H
#define GENOZIP_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Codec ids == file-format values, src/genozip.h:325-360 (also the index into the reference's codec_args[],
 * src/codec.h:65-117) */
typedef enum {
    GZ_CODEC_UNKNOWN = 0, GZ_CODEC_NONE = 1,
    GZ_CODEC_RANB = 6,  GZ_CODEC_RANW = 7,  GZ_CODEC_RANb = 8,  GZ_CODEC_RANw = 9,   /* rANS 4x16  */
    GZ_CODEC_ARTB = 16, GZ_CODEC_ARTW = 17, GZ_CODEC_ARTb = 18, GZ_CODEC_ARTw = 19   /* arith_dynamic */
} GzCodec;

/* SectionType values used on this path, src/genozip.h:378-379 */
enum { GZ_SEC_VB_HEADER = 9, GZ_SEC_B250 = 11, GZ_SEC_LOCAL = 12 };

/* LocalType, src/local_type.h:14-57 (file-format values) */
typedef enum {
    GZ_LT_SINGLETON = 0, GZ_LT_INT8 = 1, GZ_LT_UINT8 = 2, GZ_LT_INT16 = 3, GZ_LT_UINT16 = 4, GZ_LT_INT32 = 5,
    GZ_LT_UINT32 = 6, GZ_LT_INT64 = 7, GZ_LT_UINT64 = 8, GZ_LT_FLOAT32 = 9, GZ_LT_FLOAT64 = 10, GZ_LT_BLOB = 11,
    GZ_LT_BITMAP = 12, GZ_LT_CODEC = 13, GZ_LT_UINT8_TR = 14, GZ_LT_UINT16_TR = 15, GZ_LT_UINT32_TR = 16,
    GZ_LT_STRING = 26, GZ_LT_SUPP = 27
} GzLocalType;

/* return codes */
enum { GZ_OK = 1, GZ_TOO_SMALL = 0, GZ_ERR = -1, GZ_ERR_NO_DEVICE = -2, GZ_ERR_ARG = -3, GZ_ERR_HIP = -4, GZ_ERR_CORRUPT = -5 };

typedef struct GzHandle GzHandle;   /* stands in for VBlockP: owns the HIP stream and the scratch arena
                                       (the reference's vb->codec_bufs[], src/codec.c:30-63) */

/* ---- lifetime ------------------------------------------------------------------------------------------ */
/* device = HIP device ordinal; hip_stream = a hipStream_t to run on, or NULL for a private stream.
 * Fails (returns NULL, *err set) when no gfx950-capable device / runtime is present: there is NO CPU fallback. */
GzHandle *gz_create (int device, void *hip_stream, int *err);
/* the same with all streams (but the range coder chain's) at the lowest priority: for long-running batches that run beside another
 * handle's short ones (the VBlock compute driver codes the long QUAL streams on such a handle) */
GzHandle *gz_create_background (int device, int *err);
void      gz_destroy (GzHandle *h);
int       gz_sync (GzHandle *h);
const char *gz_last_error (GzHandle *h);
const char *gz_version (void);
/* Per-kernel timing: when enabled, every kernel launch of the compression pipeline is bracketed by two HIP events on
 * the handle's stream; gz_sync() folds them into per-kernel totals (this is the reference's --show-time /
 * START_TIMER..COPY_TIMER instrumentation, src/profiler.h:96-166, re-expressed for a GPU stream).
 * gz_profile_get(idx) enumerates the accumulated entries until it returns 0. enable = 2: only the launches of the two kernels a step can
 * be as long as (k_arith_chain, k_arith_model) - the events of ALL launches cost the host as much as the launches do (~700 a step). */
void gz_profile (GzHandle *h, int enable, int reset);
int  gz_profile_get (GzHandle *h, int idx, char *name, int name_cap, double *total_ms, int *launches);
/* the longest single launch of entry idx of the walk begun with gz_profile_get (h, 0, ...) */
int  gz_profile_get_max (GzHandle *h, int idx, double *max_ms);
/* the HIP stream work is queued on (a hipStream_t) - for timing with HIP events on the right stream */
void     *gz_stream (GzHandle *h);
/* How often gz_sync ran a batch a second time, unpipelined, because the arithmetic coder's persistent chain kernel never heard from the
 * model kernels it follows (kernels serialised by a profiling tool, too few hardware queues, another tenant). The results are those of the
 * second run - the reference's COMPRESS contract: false only for "too small" (src/compressor.c:89-110). The return code is that of the second
 * run and gz_last_error is left alone; gz_last_warning says what happened (a string that stays until the next fallback).
 * CONTRACT for the asynchronous batch calls (gz_codec_compress_batch, gz_vb_compress_batch): the second run reads the caller's tables and INPUT
 * buffers again, so everything a queued batch reads must stay as it was until the gz_sync that hands its results out has returned - not
 * overwritten, not freed, not overlapping any of the batch's outputs (the host-pointer forms and the VBlock compute driver keep to this by
 * themselves: staging buffers / the per-file workspace). A gz_emit_after in front of a VBlock batch is remembered with the batch: the second
 * run's section writer waits for the other handle again. */
uint32_t gz_chain_fallbacks (GzHandle *h);
const char *gz_last_warning (GzHandle *h);
/* (tests) The reciprocal the arithmetic coder's model kernel puts into a symbol's record for the model totals tot0 .. tot0 + n - 1 (a double
 * with 16 zero low bits, two words each into out_dev): RC_Encode's range / tot (c_range_coder.h:100) is exact with it as long as it lies in
 * [2^-45 / tot, 2^-45 / tot * (1 + 2^-33)] - tests/test_magic.py shows that for the interval, tests/test_gpu.py::test_record_reciprocals that
 * every total's value made on the device is inside. Synchronous. */
int gz_debug_record_inv (GzHandle *h, uint32_t tot0, uint32_t n, uint32_t *out_dev);
/* The section-writing kernels of h's NEXT gz_vb_compress_batch (layout, emit, adler32) wait until everything queued on `other`
 * so far has completed - for sections precompressed on another handle while this one's own coders run (no host wait). */
int gz_emit_after (GzHandle *h, GzHandle *other);
/* everything queued on h from now on waits (on the device) for what is queued on `other` so far */
int gz_wait_for (GzHandle *h, GzHandle *other);
/* plain copies between host memory and HBM for callers that have no HIP runtime of their own in reach (a C host program;
 * results that live in the library's workspace, e.g. GzFastqVB.z_data). Synchronous; they wait for the handle's stream first. */
int gz_download (GzHandle *h, void *host_dst, const void *dev_src, uint64_t n);
int gz_upload (GzHandle *h, void *dev_dst, const void *host_src, uint64_t n);
void *gz_dev_alloc (GzHandle *h, uint64_t n);      /* hipMalloc / hipFree on the handle's device */
void gz_dev_free (GzHandle *h, void *p);

/* ---- codec plugin surface: codec_args[codec].est_size / .compress / .uncompress ---------------------------- */

/* src/codec.h:40, src/codec_htscodecs.c:26-33, src/codec_none.c:44 */
uint32_t gz_codec_est_size (int codec, uint64_t uncompressed_len);

/* COMPRESS() src/codec.h:17-27 as implemented by codec_RANB_compress ... codec_ARTw_compress
 * (src/codec_htscodecs.c:87-94) and codec_none_compress (src/codec_none.c:13).
 * *compressed_len: in = capacity, out = payload length. Returns GZ_OK; GZ_TOO_SMALL iff capacity < est_size - 1 KB and
 * soft_fail (the caller grows and retries, src/compressor.c:89-110); GZ_ERR* otherwise. Host pointers. */
int gz_codec_compress_host (GzHandle *h, int codec, const uint8_t *uncompressed, uint32_t uncompressed_len,
                            uint8_t *compressed, uint32_t *compressed_len, int soft_fail);

/* The same for data the caller only has line by line - COMPRESS()'s second input option, `LocalGetLineCB get_line_cb`
 * (src/codec.h:23, as codec_hts_compress uses it: src/codec_htscodecs.c:51-64): get_line (user, line_i, &line, &line_len) is called
 * for line_i = 0 .. n_lines - 1 (for_line) and what it returns is concatenated - straight into pinned staging memory, from where
 * it goes to the device in one copy - and compressed as one stream. uncompressed_len: the total the caller expects (what
 * *uncompressed_len holds in the reference); a different total from the callbacks is GZ_ERR_CORRUPT (the reference's ASSERT :61).
 * (With the text resident on the device the gather is gz_local_blob_columns and the data never leaves HBM.) */
typedef void (*GzGetLineCB) (void *user, uint32_t line_i, const uint8_t **line, uint32_t *line_len);
int gz_codec_compress_lines_host (GzHandle *h, int codec, GzGetLineCB get_line, void *user, uint32_t n_lines, uint32_t uncompressed_len,
                                  uint8_t *compressed, uint32_t *compressed_len, int soft_fail);

/* UNCOMPRESS() src/codec.h:29-38: codec_rans_uncompress / codec_arith_uncompress (src/codec_htscodecs.c:100,116),
 * codec_none_uncompress (src/codec_none.c:39). Host pointers. */
int gz_codec_uncompress_host (GzHandle *h, int codec, const uint8_t *compressed, uint32_t compressed_len,
                              uint8_t *uncompressed, uint64_t uncompressed_len);

/* One entry of a stream table: an independent codec call. All pointers are DEVICE pointers.
 * in_len_dev (optional): device address of a uint32 holding the actual length, for inputs produced on the GPU
 * (then in_len is the upper bound used for planning). */
typedef struct {
    const uint8_t *in;            /* uncompressed (compress) / compressed (uncompress) bytes         */
    uint32_t       in_len;
    const uint32_t *in_len_dev;   /* NULL, or device-resident actual length (<= in_len)              */
    uint8_t       *out;           /* destination                                                      */
    uint32_t       out_cap;       /* compress: capacity, normally gz_codec_est_size ; uncompress: exact length */
    int32_t        codec;
    /* results, valid after gz_sync(): */
    uint32_t       out_len;
    int32_t        status;        /* GZ_OK / GZ_TOO_SMALL / GZ_ERR_CORRUPT                            */
    uint32_t      *out_len_dev;   /* optional (compress): the payload length is also left on the device, for a
                                     section writer that runs before the host has seen it (GzSection.precompressed) */
} GzStream;

/* Batched, asynchronous forms over a stream table (host array of descriptors of device buffers). Many streams
 * are coded concurrently: stripes, trial methods and streams all become independent GPU work items.
 * Results (out_len, status) are written back into the table by gz_sync(). */
int gz_codec_compress_batch   (GzHandle *h, GzStream *streams, int n_streams);
int gz_codec_uncompress_batch (GzHandle *h, GzStream *streams, int n_streams);

/* codec_assign_best_codec (src/codec.c:234-363) with the deterministic rule of SURVEY.md A.8 (smallest framed size
 * of the first min(len,99999) bytes; ties -> lower codec id; < 50 bytes -> GZ_CODEC_UNKNOWN). in = device pointer.
 * sizes_out (host, 9 entries: NONE,RANB,RANW,RANb,RANw,ARTB,ARTW,ARTb,ARTw) may be NULL. Synchronous. */
int gz_codec_assign_best (GzHandle *h, const uint8_t *in, uint32_t in_len, uint32_t *sizes_out);

/* codec_assign_best_codec in full. The reference tries TWELVE candidates on the sample - NONE, the eight htscodecs forms, BZ2, BSC,
 * LZMA (src/codec.c:286-295) - measures size (framed: + the 28-byte header, :328-331; NONE: the bare sample, :324) and the CPU time
 * of every trial (clock(), :322-334), and sorts them with codec_assign_sorter (:128-173): the smaller size when both took <= 5 ms
 * (or --best), otherwise a five-level trade-off of size against time, exact ties to the lower codec id. The first of the sorted
 * table is the context's codec. BZ2 / BSC / LZMA are sequential LZ / BWT coders that stay on the host: their trials
 * come from the caller (GzCodecTest rows it fills from its own codec_bz2_compress / codec_bsc_compress / codec_lzma_compress,
 * sizes framed), the nine others run here. */
typedef struct { int32_t codec; float size; float clock_us; } GzCodecTest;         /* CodecTest, src/codec.c:122-126 */
/* GZ_ASSIGN_BEST / _FAST select the SORTER's branch for flag.best / flag.fast (src/codec.c:133-143) and nothing else: --best's "keep the
 * previously selected codec when it comes second with the same size" (:342-343) and its BEST_LOCK_IN_THREASHOLD re-test / lock-in counters
 * (:20,267-274), the dropping of BSC / LZMA for evb buffers and flag.no_lzma (:292-295) are the caller's (which rows it hands in, which codec it
 * keeps): a file made with mode 1 is this sorter's choice per trial, not everything `genozip --best` does around it. */
enum { GZ_ASSIGN_NORMAL = 0, GZ_ASSIGN_BEST = 1, GZ_ASSIGN_FAST = 2 };
enum { GZ_CODEC_BZ2 = 3, GZ_CODEC_LZMA = 4, GZ_CODEC_BSC = 5 };                      /* src/genozip.h:325-360                  */
/* The sorter: tests[0..n) in the order the reference tries them (:286-289) -> sorted in place, returns tests[0].codec. The
 * reference's comparator is not a strict order, so what qsort makes of it depends on the C library: this is glibc's top-down
 * merge sort, what the reference's Linux builds run. */
int gz_codec_assign_sort (GzCodecTest *tests, int n, int mode);
/* What ONE call of codec_assign_best_codec does with a context section whose codec the segmenter left open, normal mode (src/codec.c:259-283,
 * 309-312, 352-363): bit 0 the trials run (else the section takes the file's codec z_codec, or none), bit 1 their result is committed to the
 * file's context (not from a VBlock of at most MIN (4 MB, vb_size / 2) of text; a local codec not from VBlock 1 of a context whose beginning
 * may not be representative - dt_props.vb_1_not_representative by kind of dict_id, :199-209 - unless it is the file's last), bit 2 it is
 * VBlock 10's second look (RETEST_VB_I). The VBlock compute driver decides with this function; host only. vb_size 0: no VBlock is small. */
int gz_codec_assign_rule (uint32_t vblock_i, uint64_t text_len, uint64_t vb_size, int last_of_file, int is_local,
                          int not_representative, int hard_coded, int z_codec, uint32_t data_len);
/* The nine device candidates on the first min (in_len, 99 999) bytes of `in` (device) + the caller's `extra` rows, sorted.
 * clock_ns_per_byte (32 entries, by codec id): NULL - every device trial counts as "fast enough" (<= 5 ms: what the reference sees for samples of 100 KB on
 * any current CPU except with ARTW / ARTw on wide alphabets), so that among the device candidates the smaller size wins and the
 * result does not depend on a clock; or a table by codec id of nominal costs from which clock_us = sample x cost / 1000 is made.
 * tests_out: 9 + n_extra rows, sorted (may be NULL). Returns the codec, GZ_CODEC_UNKNOWN under 50 bytes (:311-312). Synchronous. */
int gz_codec_assign_best_ex (GzHandle *h, const uint8_t *in, uint32_t in_len, const GzCodecTest *extra, int n_extra,
                             const float *clock_ns_per_byte, int mode, GzCodecTest *tests_out);

/* ---- context engine pieces -------------------------------------------------------------------------------- */

/* b250_zip_generate (src/b250.c:202-267): seg-format b250 (little-endian VARL, tag in last byte; nodes new to the
 * VB are 4-byte node indices) -> PIZ-format VARL (big endian, tag first, node->word converted, ONE_UP applied when
 * the dictionary has > 1024 words). node2word[i] = word index of VB-local node (ol_nodes_len + i) after the merge
 * (ctx->nodes after ctx_merge_in_vb_ctx, src/context.c:1032). out must hold seg_len bytes; *out_len_dev receives
 * the length. All pointers device. */
int gz_b250_generate (GzHandle *h, const uint8_t *seg, uint32_t seg_len, uint32_t ol_nodes_len,
                      const int32_t *node2word, uint32_t n_new_nodes, uint8_t *out, uint32_t *out_len_dev);

/* the same for many contexts (all VBlocks x contexts of a batch) in one launch; all pointers device */
typedef struct {
    const uint8_t  *seg; uint32_t seg_len;
    const uint32_t *seg_len_dev;     /* optional device-resident actual length (<= seg_len) */
    uint32_t        ol_nodes_len;
    const int32_t  *node2word; uint32_t n_new_nodes;
    uint8_t        *out;             /* seg_len bytes */
    uint32_t       *out_len_dev;     /* receives the generated length */
    int32_t        *status_dev;      /* optional: receives 1 ok / -5 malformed input / 3 dropped (see r1)           */
    const uint8_t  *r1; const uint32_t *r1_len_dev;   /* optional (paired FASTQ, R2's VBlock): the generated b250 of the same
                                      * context in R1's VBlock; an identical R2 b250 is dropped (src/b250.c:270-277):
                                      * *out_len_dev = 0, status 3, no section is written                              */
} GzB250Job;
int gz_b250_generate_batch (GzHandle *h, const GzB250Job *jobs, int n_jobs);

/* zip_generate_local (src/zip.c:167-219): native little-endian elements -> file order (big endian; signed types
 * zig-zag "interlaced" first, src/context.h:99-101, src/buffer.c:336-350), then optional matrix transpose
 * rows x cols -> cols x rows (dyn_int_transpose, src/dyn_int.c:45-132, full-matrix case). In place on `data`
 * (device), `scratch` (device, same size) needed when transpose_cols != 0. Returns the resulting ltype
 * (e.g. GZ_LT_UINT8_TR) or a negative error. n = element count. */
int gz_local_generate (GzHandle *h, int ltype, void *data, uint64_t n, uint32_t transpose_cols, void *scratch);

/* PIZ inverse (lt_desc[].file_to_native, src/local_type.h:75-108; BGEN_transpose_u##n##_buf src/buffer.c:364-391) */
int gz_local_to_native (GzHandle *h, int ltype, void *data, uint64_t n, uint32_t transpose_cols, void *scratch);

/* ---- section writer --------------------------------------------------------------------------------------- */

/* One b250 / local section of one VBlock: the fields of SectionHeaderCtx (src/sections.h:146-167,419-435) that the
 * caller decides, as zfile_compress_b250_data / zfile_compress_local_data fill them (src/zfile.c:288-364). */
typedef struct {
    const uint8_t  *data;         /* device: section payload before compression (ctx->b250.data / ctx->local.data) */
    uint32_t        data_len;     /* bytes (upper bound if data_len_dev != NULL)                               */
    const uint32_t *data_len_dev; /* device-resident actual length, or NULL. 0 there = dropped on the device   */
                                  /* (an R2 b250 identical to R1's, src/b250.c:270-277): no section is written  */
    uint8_t  section_type;        /* GZ_SEC_B250 / GZ_SEC_LOCAL                                                */
    uint8_t  codec;               /* ctx->bcodec / ctx->lcodec ; UNKNOWN -> RANB (zfile.c:300,337)             */
    uint8_t  sub_codec;
    uint8_t  flags;               /* struct FlagsCtx, LSB first: store:2 paired:1 store_delta:1 spl_custom:1
                                     all_the_same:1 ctx_specific_flag:1 store_per_line:1 (sections.h:99-118) */
    uint8_t  ltype;
    uint8_t  param;
    uint8_t  b250_size_or_nothing_char; /* byte 30: B250_VARL(4) for b250, nothing_char for integer locals   */
    uint8_t  dict_id[8];
    uint8_t  precompressed;       /* data / data_len(_dev) already ARE the codec's payload (gz_codec_compress_batch ran ahead,
                                     possibly on another handle: gz_emit_after): only framed here; `codec` = the effective
                                     codec, raw_len = data_uncompressed_len of the header                           */
    uint32_t raw_len;
    uint8_t  hdr_codec;           /* nonzero: a complex codec whose primary stream this is (GZ_CODEC_DOMQ): it goes to the header's
                                     codec byte and the coder of the stream (`codec`, as given: the 50-byte rule is for simple codecs)
                                     to sub_codec, as codec_domq_compress sets them (src/codec_domq.c:487-500)                */
} GzSection;

typedef struct {
    uint32_t vblock_i;            /* 1-based */
    uint32_t recon_size, longest_line_len, longest_seq_len;
    uint8_t  digest[16];
    uint8_t  vb_flags;
    const GzSection *sections;    /* in the order they must appear in z_data                   */
    uint32_t n_sections;
    uint8_t *z_data;              /* device: receives SEC_VB_HEADER (84 B) + every section (40 B header + payload) */
    uint64_t z_cap;               /* >= gz_vb_z_bound()                                                        */
    /* results after gz_sync(): */
    uint64_t z_len;               /* == z_data_bytes patched into the VB header (zfile.c:1139-1144)            */
    int32_t  status;
    uint32_t mark_section;        /* in: an index into `sections` (or n_sections)                                       */
    uint32_t mark_index;          /* out: how many sections were WRITTEN in front of it (a section generated on the device can be
                                     dropped there, src/b250.c:270-277): where a section made by the host afterwards goes in    */
} GzVBlock;

uint64_t gz_vb_z_bound (const GzSection *sections, uint32_t n_sections);

/* zip_compress_one_vb's codec phase for a batch of VBlocks (src/zip.c:560-585): zfile_compress_vb_header, then for
 * every section comp_compress (src/compressor.c:18: < 50 B -> CODEC_NONE, payload, z_digest = adler32, big-endian
 * lengths) appended to z_data. All VBlocks and all their sections run concurrently on the GPU. Asynchronous. */
int gz_vb_compress_batch (GzHandle *h, GzVBlock *vbs, int n_vbs);

/* section-level decode for the round trip: walks z_data (device), checks magic / adler32 (src/zfile.c:212-218),
 * decodes every section payload into out (device, concatenated in order), writes per-section offsets (n+1 entries,
 * host). Synchronous. */
int gz_vb_uncompress (GzHandle *h, const uint8_t *z_data, uint64_t z_len, uint8_t *out, uint64_t out_cap,
                      uint64_t *section_offsets_host, uint32_t max_sections, uint32_t *n_sections_out);

/* the same for n_vbs VBlocks at once (what piz does per VBlock in zfile_read_section_do / comp_uncompress, src/zfile.c:212-218,
 * src/compressor.c:196-246, here for a whole batch): the section headers of every VBlock are walked and every payload's adler32 is
 * checked by two kernels, then ALL payloads of ALL VBlocks are decoded as one batch (gz_codec_uncompress_batch) - a wave per stream,
 * hundreds of streams at a time. z_data[v] / out[v]: device; section_offsets_host: n_vbs rows of max_sections + 1 entries (or NULL);
 * n_sections_out: n_vbs entries. A section whose codec the device has no decoder for - one of the host's coders (BZ2 / LZMA / BSC:
 * gz_zip_set_host_codecs), CODEC_ACGT (this library's own NONREF section: gz_vb_insert_section, codec = ACGT, sub_codec = LZMA - the reference's
 * codec_acgt_uncompress runs the sub-codec and unpacks), any other codec of src/genozip.h:326-360 - is checked (magic, adler32) but not
 * decoded: its stretch of out[v] is zeroed and its entry of section_offsets_host carries GZ_SECTION_NOT_DECODED on top of the offset - the
 * caller's own codec_args[codec].uncompress fills it. A codec byte the format does not have (0, >= GZ_NUM_CODECS): GZ_ERR_CORRUPT. Synchronous. */
#define GZ_SECTION_NOT_DECODED (1ull << 63)
#define GZ_NUM_CODECS 42            /* NUM_CODECS, src/genozip.h:360 */
int gz_vb_uncompress_many (GzHandle *h, int n_vbs, const uint8_t *const *z_data, const uint64_t *z_len, uint8_t *const *out,
                           const uint64_t *out_cap, uint64_t *section_offsets_host, uint32_t max_sections, uint32_t *n_sections_out);

/* adler32 of a device buffer (libdeflate_adler32 as used by src/compressor.c:161). Synchronous. */
int gz_adler32 (GzHandle *h, const uint8_t *data, uint64_t len, uint32_t *adler_out);

/* ---- CODEC_ACGT pre-transform ------------------------------------------------------------------
 * The part of codec_acgt_compress (src/codec_acgt.c:64-137) that is not the LZMA sub-codec: SEQ bases -> 2 bits each
 * (codec_acgt_pack :45-55 with the table of src/reference.c:45-58; little-endian 64-bit words, excess bits clear) plus
 * the NONREF_X exception stream (:66-70,108-110: 0 for ACGT, 1 for acgt, the character otherwise). x may be seq itself
 * (the reference overlays NONREF_X.local on NONREF.local). *has_x_host = 0 means "no exceptions": the caller sets
 * flags.acgt_no_x and drops NONREF_X (:133-137). The packed bytes then go to the sub-codec (LZMA, host) and x through
 * the ordinary codec path (gz_codec_compress_*). Synchronous. */
uint64_t gz_acgt_packed_len (uint64_t n_bases);
int gz_acgt_pack (GzHandle *h, const uint8_t *seq, uint64_t n_bases, uint8_t *packed, uint8_t *x, int *has_x_host);
/* codec_acgt_uncompress + codec_xcgt_uncompress (src/codec_acgt.c:177-199,218-246); x == NULL: flags.acgt_no_x */
int gz_acgt_unpack (GzHandle *h, const uint8_t *packed, const uint8_t *x, uint64_t n_bases, uint8_t *seq);

/* ---- seg-side appends, a column at a time -------------------------------------------------
 * What the segmenter leaves in a context after evaluating the snips of one field of every line: node indices, the
 * VBlock's new dictionary words and nodes, counts and the seg-format b250 (ctx_create_node_do src/context.c:320-404
 * with hash_get_entry_for_seg src/hash.c:530-576 and ctx_insert_to_dict src/context.c:50-71; b250_seg_append
 * src/b250.c:112-163) - computed for the whole column at once: a snip's node index is its index in ol_nodes (the
 * dictionary cloned from the file-level context, ctx_clone) or ol_nodes.len + the rank of its first occurrence in the
 * VBlock. All pointers device; asynchronous on the handle's stream (results valid after gz_sync).
 *   snip k = text[off[k] .. off[k]+len[k]); len 0 = WORD_INDEX_EMPTY, or WORD_INDEX_MISSING if off[k] == GZ_SNIP_MISSING
 *   (the reference's snip == NULL, context.c:331-335). Limits: n < 2^30, the VBlock's dictionary < 4 GB, at most 65 535
 *   columns per call.
 * The b250 produced here is the input of gz_b250_generate once the host has merged the new words (a4) and knows
 * node2word[]. */
#define GZ_SNIP_MISSING 0xffffffffu
typedef struct {
    uint64_t dict_len;            /* bytes of `dict` (every new snip + its NUL)                                   */
    uint64_t b250_len;            /* bytes of `b250`: ONE entry while the column is all-the-same (b250.c:117-141)  */
    uint64_t b250_count;          /* entries appended (ctx->b250.count)                                            */
    uint32_t n_new;               /* nodes new to the VBlock (ctx->nodes.len)                                      */
    uint32_t all_the_same;        /* ctx->flags.all_the_same                                                       */
    int32_t  status;              /* 1 ok; 0: dict_cap too small (dict not written, everything else is valid)      */
    uint32_t reserved;
} GzColumnResult;
typedef struct {
    const uint8_t  *text; const uint32_t *off, *len; uint32_t n;
    const uint8_t  *ol_dict; const uint64_t *ol_char_index; const uint32_t *ol_snip_len; uint32_t n_ol;   /* CtxNode.char_index / .snip_len of ol_nodes */
    int32_t  *node_index;         /* [n]                                                                           */
    uint8_t  *dict; uint64_t dict_cap;
    uint64_t *node_char_index; uint32_t *node_snip_len;      /* [n]: the CtxNode fields of the new nodes           */
    uint32_t *counts;             /* [n_ol + n]: occurrences in this VBlock per node (vctx->counts)                */
    uint8_t  *b250;               /* [4 n]                                                                         */
    GzColumnResult *result_dev;
} GzColumnJob;
int gz_ctx_seg_columns (GzHandle *h, const GzColumnJob *jobs, int n_jobs);

/* dyn_int_append over a column (src/dyn_int.c:232-320; final type dyn_int_get_ltype :27-43 with lt_order :17): the
 * values at the narrowest of UINT8, INT8, UINT16, INT16, UINT32, INT32, INT64 that holds them all (one less at the top
 * when the context has a nothing_char; entries flagged in is_nothing are stored as the type's maximum, :322-345),
 * native little endian - gz_local_generate then puts them in file order. out: 8 n bytes, 8-byte aligned. */
typedef struct { uint64_t len; int32_t ltype; uint32_t width; uint32_t order; uint32_t reserved; } GzDynIntResult;
typedef struct {
    const int64_t *values; const uint8_t *is_nothing /* or NULL */; uint64_t n; uint32_t nothing_char;
    uint8_t *out; GzDynIntResult *result_dev;
    const uint64_t *n_dev;        /* optional: device-resident actual count (<= n), e.g. gz_int_columns' n_values_dev */
} GzDynIntJob;
int gz_dyn_int_columns (GzHandle *h, const GzDynIntJob *jobs, int n_jobs);

/* seg_add_to_local_fixed_do over a column (src/seg.c:1268-1287): the field of every line gathered into the context's
 * local, each followed by a NUL if add_nul - e.g. SEQ of every read -> NONREF.local, QUAL -> QUAL.local.
 * pre_len (<= 4) bytes `pre` in front of every item: a field whose snips carry a constant lead-in, e.g. sam_seg_CIGAR's
 *   { SNIP_SPECIAL, SAM_SPECIAL_CIGAR } (src/sam_cigar.c:717-720); with item_off (device, n entries, or NULL) = where each item
 *   starts in `out`, the gathered items are the column gz_ctx_seg_columns then segs (length = len + pre_len).
 * pad_to (0 or a power of two) / pad_byte: every item is followed by pad_byte's up to the next multiple of pad_to - SAM's
 *   verbatim SEQ in NONREF.local: sam_seg_SEQ_pad_nonref adds 'A's so that every read starts a new byte of the 2-bit packing
 *   (src/sam_seq.c:224-229). `out` needs n * (pre_len + add_nul + pad_to) bytes beyond the items. */
typedef struct {
    const uint8_t *text; const uint32_t *off, *len; uint32_t n; uint32_t add_nul;
    uint8_t *out; uint64_t *out_len_dev;
    uint8_t pre[4]; uint32_t pre_len; uint32_t pad_to; uint32_t pad_byte;
    uint32_t *item_off, *item_len;    /* (item_len: len + pre_len of every item - the lengths of that column) */
} GzBlobJob;
int gz_local_blob_columns (GzHandle *h, const GzBlobJob *jobs, int n_jobs);

/* ---- N1 (first part): the line buffer -> lines -> FASTQ records -> tokens -------------------------
 * What the segmenter's line loop does before any context is touched, for the whole buffer at once. All pointers
 * device, asynchronous, text < 4 GB per call.
 * gz_text_lines: seg_get_next_line (src/seg.c:200-236) - offset and length of every line (length without the newline
 *   and without a '\r' before it; a last line without a newline counts). off/len hold `cap` entries; n_lines is the
 *   true count, status 0 if it exceeds cap (then only the first cap lines are written).
 * gz_fastq_records: fastq_seg_get_lines (src/fastq.c:1002-1135) - every 4 lines are a read; the 8 output arrays of
 *   max_reads entries are (offset, length) of line 1 without its '@', SEQ, line 3 without its '+', QUAL. first_bad =
 *   the first read that is not '@'.. / SEQ / '+'.. / QUAL with len(QUAL) == len(SEQ) (the reference aborts there,
 *   :1008-1010,1076,1121), 0xffffffff if none.
 * gz_tokenize_column: the items of a container whose separators are known - the QNAME flavors (src/qname_flavors.h:21-49,
 *   src/qname.c:715-866; seg_get_next_item src/seg.c:153-198): item i of a snip ends at the first seps[i] after item
 *   i-1, the last item is the rest; item-major output [i * n + k], ready to be columns of gz_ctx_seg_columns /
 *   gz_dyn_int_columns. A snip lacking a separator counts in *n_bad_dev and stays whole in item 0 (the reference
 *   segs such a qname as one snip). n_seps <= 31. */
typedef struct { uint64_t n_lines; int32_t status; uint32_t reserved; /* gz_text_lines: 1 if a line ended in \r\n (the \r is not part of its length) */ } GzLinesResult;
int gz_text_lines (GzHandle *h, const uint8_t *text, uint64_t n_bytes, uint32_t *line_off, uint32_t *line_len, uint32_t cap,
                   GzLinesResult *result_dev);
typedef struct { uint64_t n_reads; uint32_t first_bad; uint32_t reserved; } GzFastqResult;
int gz_fastq_records (GzHandle *h, const uint8_t *text, const uint32_t *line_off, const uint32_t *line_len,
                      const GzLinesResult *lines_dev, uint32_t max_reads,
                      uint32_t *l1_off, uint32_t *l1_len, uint32_t *seq_off, uint32_t *seq_len,
                      uint32_t *l3_off, uint32_t *l3_len, uint32_t *qual_off, uint32_t *qual_len, GzFastqResult *result_dev);
int gz_tokenize_column (GzHandle *h, const uint8_t *text, const uint32_t *off, const uint32_t *len, uint32_t n,
                        const char *seps, uint32_t n_seps, uint32_t *item_off, uint32_t *item_len, uint32_t *n_bad_dev);

/* seg_integer_or_not over a column (src/seg.c:531-560; str_get_int src/strings.c:315-341): a snip that is a decimal
 * integer the text can be rebuilt from goes to the context's dyn-int local and leaves the one-character snip
 * SNIP_LOOKUP in the b250; so does the context's nothing_char on its own (as a "nothing" entry); everything else stays a
 * snip. Outputs: snip_off/snip_len [n] = the column for gz_ctx_seg_columns (lookup_off = offset in `text` of a byte
 * holding SNIP_LOOKUP); values/is_nothing [n] = the compacted column for gz_dyn_int_columns, *n_values_dev entries.
 * Numbers beyond int64 stay snips. All pointers device, asynchronous. */
int gz_seg_integer_or_not (GzHandle *h, const uint8_t *text, const uint32_t *off, const uint32_t *len, uint32_t n,
                           uint32_t nothing_char, uint32_t lookup_off, uint32_t *snip_off, uint32_t *snip_len,
                           int64_t *values, uint8_t *is_nothing, uint64_t *n_values_dev);

/* dyn_int_transpose, partial case (src/dyn_int.c:64-72,89-96,104-129): a VCF FORMAT field some of whose samples were
 * copied by VCF_COPY_SAMPLE has only the remaining elements in local - those whose missing[r * cols + c] is 0, in
 * row-major order. gz_local_generate_partial puts them in file byte order and then in COLUMN-major order; returns
 * GZ_LT_UINT{8,16,32}_PTR (28..30). gz_local_partial_to_native is the PIZ inverse (takes the _PTR type, returns the
 * base type). ltype: GZ_LT_UINT8/16/32; data: n_present elements, in place; scratch: as large as data; missing:
 * rows * cols bytes. All device. A mask that does not leave n_present elements -> GZ_ERR at the next gz_sync-ing call
 * is avoided by checking here: these two calls synchronise and return GZ_ERR_CORRUPT. */
enum { GZ_LT_UINT8_PTR = 28, GZ_LT_UINT16_PTR = 29, GZ_LT_UINT32_PTR = 30 };
int gz_local_generate_partial (GzHandle *h, int ltype, void *data, uint64_t n_present, uint32_t rows, uint32_t cols,
                               const uint8_t *missing, void *scratch);
int gz_local_partial_to_native (GzHandle *h, int ltype, void *data, uint64_t n_present, uint32_t rows, uint32_t cols,
                                const uint8_t *missing, void *scratch);

/* ---- a4: the ordered dictionary merge (HOST) -------------------------------------------------------
 * ctx_merge_in_one_vctx (src/context.c:938-1079) with ctx_commit_node (:269-316), hash_global_get_entry (src/hash.c:444-482),
 * the singleton tables (src/hash.c:280-366) and ctx_drop_all_the_same (src/context.c:795-871): the words a VBlock added to
 * its copy of a context go into the file-level context (zctx) in VBlock order and get their word indices; a word seen
 * exactly once so far (count 1 in this VBlock, not yet in the dictionary, not a known singleton) is diverted to the
 * VBlock's `local` and its node then points at the SNIP_LOOKUP word. Host pointers throughout, no GPU work: this is the
 * serial step between gz_ctx_seg_columns (which produces dict / node_char_index / node_snip_len / counts / n_new of every
 * VBlock context) and gz_b250_generate (which consumes node2word). One GzZctx per context of the file; a context must see
 * its VBlocks in order (vb_i = 1 first, context.c:944), different contexts may be merged on different host threads. */
typedef struct GzZctx GzZctx;
/* hash_alloc_global (src/hash.c:227-239): global_hash.len = gz_hash_next_size_up (3 x estimated_entries), where
 * estimated_entries is hash_get_estimated_entries' figure (src/hash.c:109-225; 0 -> 1000). The size never reaches the file
 * but decides which singletons share a bucket. */
GzZctx  *gz_zctx_create (uint32_t estimated_entries);
void     gz_zctx_destroy (GzZctx *z);
uint32_t gz_hash_next_size_up (uint64_t size);                              /* src/hash.c:26-48 */
typedef struct {
    /* in: the VBlock's context */
    uint32_t vblock_i;               /* 1-based */
    uint32_t n_ol;                   /* vctx->ol_nodes.len: words of the zctx at the time this VBlock cloned it         */
    uint32_t n_new;                  /* vctx->nodes.len                                                               */
    const uint8_t  *dict;            /* vctx->dict, node_char_index / node_snip_len [n_new], counts [n_ol + n_new]    */
    const uint64_t *node_char_index;
    const uint32_t *node_snip_len;
    const uint32_t *counts;
    uint8_t  can_have_singletons;    /* ctx_can_have_singletons (src/context.h:263-265): ltype == LT_SINGLETON, !no_stons,
                                        store != STORE_INDEX, !all_the_same                                           */
    uint8_t  flags;                  /* vctx->flags (struct FlagsCtx as in GzSection.flags)                           */
    uint8_t  no_drop_b250;           /* vctx->no_drop_b250                                                            */
    uint8_t  pair2_identical;        /* is_fastq_pair_2 && fastq_zip_use_pair_identical (dict_id) (context.c:810)     */
    uint64_t b250_len, local_len;    /* vctx->b250.len / local.len before the merge                                   */
    uint64_t b250_r1_len, local_r1_len;
    int32_t  ats_node_index;         /* the one b250 entry of an all-the-same context (b250_seg_get_last)             */
    /* in / out */
    uint8_t  lcodec, bcodec;         /* 0 -> inherited from the zctx (context.c:980-981)                              */
    /* out */
    uint8_t  dropped_b250;           /* ctx_drop_all_the_same freed the b250: no B250 section for this VBlock         */
    uint8_t  reserved;
    int32_t *node2word;              /* [n_new] word index of every VBlock node (vctx->nodes after the merge, :1032)  */
    uint8_t *ston_local; uint64_t ston_cap;   /* singletons' text, each followed by NUL, appended to vctx->local (:297) */
    uint64_t ston_len; uint32_t n_stons;      /* ston_cap >= the VBlock's dict length always suffices                   */
} GzMergeJob;
/* Returns GZ_OK; GZ_TOO_SMALL when ston_local cannot hold the singletons; GZ_ERR_ARG (n_ol beyond the zctx's words, ...) */
int gz_ctx_merge (GzZctx *z, GzMergeJob *job);
typedef struct {
    const uint8_t *dict; uint64_t dict_len;            /* zctx->dict: every word + NUL, in word-index order (-> SEC_DICT)  */
    const uint64_t *char_index; const uint32_t *snip_len; const uint64_t *counts;   /* [n_words]                          */
    uint32_t n_words, hash_len;
    uint64_t n_singletons, n_failed_singletons;
    uint8_t  flags; uint8_t rm_dict_all_the_same; uint8_t lcodec, bcodec; int32_t all_the_same_wi;
} GzZctxView;
/* what ctx_clone (src/zip.c:528) overlays into the next VBlock - pointers stay valid until the next gz_ctx_merge */
int gz_zctx_view (const GzZctx *z, GzZctxView *view);
/* codec_assign_best_codec's commit to the file-level context (src/codec.c:352-363) */
int gz_zctx_commit_codec (GzZctx *z, int is_local, int codec);

/* ---- N1 for BAM: the alignment records of an uncompressed BAM stream -> alignment lines (SURVEY 8(0) configs[2]) ----------------
 * bam_seg_txt_line (src/bam_seg.c:425-520) reads a record's fields in their binary form, converts SEQ / CIGAR / QUAL to their textual
 * forms (bam_seq_to_sam src/bam_seq.c:58-103, sam_cigar_binary_to_textual src/sam_cigar.c:155-206, bam_rewrite_qual src/bam_seg.c:
 * 276-284) and segs them with the functions SAM uses; the contexts of a BAM file are SAM's. Here: records -> the text of their
 * alignment lines in HBM, which the one-line-record plan of the VBlock driver (genozip_amd/sam.py) then segs.
 * `bam` (device) points at the FIRST ALIGNMENT (the caller has read magic, header text and reference names - the txt-header component,
 * host work) and holds whole records; < 4 GB per call. The BGZF layer in front of it is I/O and not part of this library.
 * gz_bam_records: the offsets of the records (every record names the next by its block_size, bam_unconsumed_scan_forwards
 *   src/bam_seg.c:49-67). status GZ_ST_CORRUPT: a block_size that does not fit (first_bad = index of that record; bam_seg.c:444-447) or
 *   a last record that does not end with the stream; GZ_ST_TOO_SMALL: more than cap records (n_records is the true count).
 *   n_rewalked: 64 KB chunks whose first record was not where the heuristic of bam_unconsumed_scan_backwards (:76-130) found one.
 * gz_bam_to_sam: QNAME FLAG RNAME POS MAPQ CIGAR RNEXT PNEXT TLEN SEQ QUAL [TAG:TYPE:VALUE ..] '\n' per record (RNAME / RNEXT through
 *   the header's reference names: ref_names is their concatenation, ref_name_off their n_ref + 1 offsets; '*' for -1, '=' for
 *   RNEXT == RNAME), integers of every width as TYPE i, Z / H / A, B arrays of integers. line_off: n_rec + 1 entries or NULL.
 *   status GZ_ST_CORRUPT (first_bad): fields that do not fit their record, a reference id outside the header, a malformed optional
 *   field, or a float (f, B:f: kept binary by the reference behind bam_piz_special_FLOAT, src/sam.h:863 - not built);
 *   GZ_ST_TOO_SMALL: the text needs text_len > text_cap bytes (nothing is written). */
typedef struct { uint64_t n_records, text_len; int32_t status; uint32_t first_bad, n_rewalked, reserved; } GzBamResult;
int gz_bam_records (GzHandle *h, const uint8_t *bam, uint64_t n_bytes, int32_t n_ref, uint32_t *rec_off, uint32_t cap, GzBamResult *result_dev);
int gz_bam_to_sam (GzHandle *h, const uint8_t *bam, uint64_t n_bytes, const uint32_t *rec_off, uint32_t n_rec,
                   const uint8_t *ref_names, const uint32_t *ref_name_off, int32_t n_ref,
                   uint8_t *text, uint64_t text_cap, uint32_t *line_off, GzBamResult *result_dev);

/* ---- the VBlock compute driver: zip_compress_one_vb for a batch of FASTQ VBlocks (SURVEY 8(a) a15 + 8(f) N1) ----------------
 * What zip_compress_one_vb (src/zip.c:510-601) does between "text of a VBlock in memory" and "z_data ready to be written",
 * for a whole batch of VBlocks at once and with the per-field rules of fastq_seg_txt_line (src/fastq.c:1249-1309) and
 * qname_seg_qf (src/qname.c:715-806) given as DATA - the plan below is what segconf + *_seg_initialize decide once per file:
 *   text -> lines -> reads -> line-1 items (gz_text_lines / gz_fastq_records / tokenizer)
 *        -> seg-side appends of every context, a column at a time (a1-a3)
 *        -> dictionary merge in VBlock order on the host (a4)                          [the one serial step]
 *        -> b250 generation (a5), local byte order (a6), R2 == R1 drops (src/b250.c:270-277, src/zip.c:224-234)
 *        -> codec per context (a8: assign on the first VBlock that has the data, commit to the file-level context)
 *        -> sections in the order of zip_compress_all_contexts_local / _b250 (a15: DEP_L0..L2 x ascending did_i, locals
 *           that only exist because of the merge (singletons) after the others unless vb_i == 1, then the b250s,
 *           src/zip.c:247-342,565-585) -> comp_compress (a9-a13) -> VB header with z_data_bytes (a16).
 * Kinds of context (how a line's field reaches the context):                                                              */
enum {
    GZ_FQ_CONST      = 1,  /* every line segs the same snip (a container, SEQ's special snip, line 3, an EOL ...): the b250
                              is all-the-same; nothing is computed per line                                                */
    GZ_FQ_ITEM_TEXT  = 2,  /* seg_by_ctx of a line-1 item (qname.c:790): dictionary + b250. With a `snip` (<= 4 bytes): every snip is
                              `snip` + the item - a field the reference segs behind a constant lead-in, e.g. sam_seg_CIGAR's
                              { SNIP_SPECIAL, SAM_SPECIAL_CIGAR } + the CIGAR text (src/sam_cigar.c:717-720)                       */
    GZ_FQ_ITEM_INT   = 3,  /* seg_integer_or_not of a line-1 item (qname.c:773-775): dyn-int local + SNIP_LOOKUP b250     */
    GZ_FQ_ITEM_DELTA = 4,  /* seg_self_delta of a line-1 item (qname.c:750-759, sorted_by_qname): the delta against the
                              previous line in a dyn-int local, the constant snip `snip` (SNIP_SELF_DELTA '$') in the b250  */
    GZ_FQ_SEQ        = 5,  /* SEQ of every read -> NONREF.local, then CODEC_ACGT's 2-bit pack (fastq_seq.c:139-154,
                              codec_acgt.c:64-137): the packed bytes are handed back for the LZMA sub-codec (host, out of
                              scope); dict_id names the NONREF_X exception stream, written when there are any    */
    GZ_FQ_QUAL       = 6,  /* QUAL of every read -> QUAL.local (fastq_qual.c:24-60 through the get_line callback). When the plan
                              also holds its three GZ_FQ_QUAL_AUX contexts, codec_assign_best_qual_codec (src/codec.c:391-450) is
                              followed as far as FASTQ can go: the file's first VBlock decides between CODEC_DOMQ (N3: its QUAL
                              lines pass codec_domq_qual_data_is_a_fit_for_domq) and a plain LT_BLOB local, for the whole file.
                              With a `snip` (FASTQ: { SNIP_SPECIAL, FASTQ_SPECIAL_monochar_QUAL }, 2 bytes) the context also gets
                              fastq_seg_QUAL's b250 (src/fastq_qual.c:24-47): a line that is one score repeated (str_is_monochar) segs
                              `snip` + that score and is left out of QUAL.local (dl->dont_compress_QUAL: the callback hands the codecs
                              0 bytes for it, :74), every other line segs { SNIP_LOOKUP } (seg_simple_lookup) - a file without such
                              lines ends up with neither b250 nor dictionary (ctx_drop_all_the_same, src/context.c:795-871). Without a
                              `snip` the context has its local only (the SAM plan)                                               */
    GZ_FQ_QUAL_AUX   = 7,  /* DOMQRUNS / QUALMPLX / DIVRQUAL (item 0 / 1 / 2) of the plan's QUAL context: LT_SUPP locals at DEP_L2
                              (codec_domq.c:308-313); DOMQRUNS' dictionary takes the denormalisation table (:240-244)              */
    GZ_FQ_TOPLEVEL   = 8,  /* the TOPLEVEL container, segged ONCE per VBlock with repeats = the VBlock's reads (fastq_seg_finalize,
                              src/fastq.c:845-943 -> container_seg): `snip` holds the binary Container (its first con_len bytes: 8 +
                              12 per item, src/container.h:74-92, the 24 repeat bits zero) followed by its prefixes; the driver sets
                              the repeats of every VBlock and segs SNIP_CONTAINER + base64 (Container) + prefixes
                              (container_prepare_snip, src/container.c:35-64): one b250 entry                                      */
    GZ_FQ_SEQ_SNIP   = 9,  /* SQBITMAP: the snip fastq_seg_SEQ segs for every read, which carries the read's length
                              (src/fastq_seq.c:45-154: `snip` + decimal seq_len, `snip` being SNIP_SPECIAL, FASTQ_SPECIAL_unaligned_SEQ,
                              ' '); a read of one repeated base takes that base in the place of ' ' and is left out of NONREF
                              (:120-126), an empty read is `snip` with '*' and no length (:113-117). Generated and evaluated on the
                              device like a textual item. Must come before the plan's GZ_FQ_SEQ context                          */
    GZ_FQ_ITEM_EXPECT = 10 /* no context: the item must be exactly `snip` in every record - text the plan's containers carry as a PREFIX
                              (the "NM:i" of an optional field whose value is the next item: the AUX container's prefixes, sam_seg_aux_all
                              src/sam_seg.c:1363-1433). A record where it is not makes the call fail with GZ_ERR_CORRUPT: the plan does not
                              describe this file (flavor / tag-layout discovery is segconf's job, host). dict_id / did_i are ignored     */
};
typedef struct {
    uint8_t  dict_id[8];
    uint16_t did_i;               /* position in the VBlock's context array: sections appear in ascending did_i           */
    uint8_t  kind;                /* GZ_FQ_*                                                                              */
    uint8_t  item;                /* GZ_FQ_ITEM_*: index of the line-1 item                                               */
    uint8_t  local_dep;           /* LocalDepType DEP_L0..DEP_L2 (src/context_struct.h:26)                                */
    uint8_t  flags;               /* struct FlagsCtx without paired / all_the_same (set here)                             */
    uint8_t  no_stons;            /* ctx->no_stons (qname items of paired files: src/qname.c:389-393)                      */
    uint8_t  lcodec, bcodec;      /* hard-coded codec, or 0: codec_assign_best_codec                                      */
    uint8_t  pair_identical;      /* fastq_zip_use_pair_identical (src/fastq.c:238-243)                                    */
    uint8_t  pair_assisted_b250;  /* fastq_zip_use_pair_assisted (.., SEC_B250) (src/fastq.c:224-234)                      */
    uint8_t  nothing_char;        /* GZ_FQ_ITEM_INT                                                                       */
    const uint8_t *snip; uint32_t snip_len;   /* GZ_FQ_CONST / GZ_FQ_ITEM_DELTA / GZ_FQ_TOPLEVEL / GZ_FQ_SEQ_SNIP: the snip (host pointer) */
    uint32_t con_len;             /* GZ_FQ_TOPLEVEL: bytes of the binary Container at the start of `snip`                  */
    uint8_t  per_sample;          /* VCF (plan.n_samples): the item is FORMAT subfield `item` of EVERY sample of every line (vcf_seg_samples,
                                     src/vcf_samples.c:1601): the column has lines x samples entries, line by line                         */
    uint8_t  transposed;          /* per_sample GZ_FQ_ITEM_INT: ctx->dyn_transposed (src/vcf_samples.c:121-130) - the lines x samples matrix of
                                     an unsigned dyn-int local is written samples x lines, LT_UINTn_TR, param 0 = "as many columns as the
                                     file has samples" (dyn_int_transpose, src/dyn_int.c:45-132)                              */
    uint8_t  segs_per_line;       /* GZ_FQ_CONST: how many times a read segs the snip (E2L: 3, src/fastq.c:1300-1304); 0 = once.
                                     Only the word's count depends on it                                                    */
    const uint8_t *r2_node; uint32_t r2_node_len;   /* GZ_FQ_SEQ_SNIP / GZ_FQ_ITEM_TEXT: a node every R2 VBlock (GzFastqVB.r1 >= 0) creates in
                                     this context BEFORE it segs anything - fastq_seg_initialize's ctx_create_node (VB, FASTQ_SQBITMAP,
                                     { SNIP_SPECIAL, FASTQ_SPECIAL_mate_lookup }) (src/fastq.c:664-665; ctx_create_node_is_new,
                                     src/context.c:402-409: the node's count stays 0). Unless the file's dictionary has the word already it
                                     is the VBlock's first new node and the merge makes it a dictionary word. Host pointer, <= 16 bytes; NULL: none */
} GzFastqCtx;
typedef struct {
    const GzFastqCtx *ctxs; uint32_t n_ctxs;
    char     seps[32]; uint8_t sep_counts[32]; uint32_t n_seps;   /* line 1 (without '@') as one container: item i ends at the
                                   sep_counts[i]-th seps[i] (CI0_COLONn, src/qname_flavors.h:40-49); n_seps + 1 items        */
    uint8_t  paired;              /* --pair: R2 VBlocks name their R1 VBlock                                              */
    uint32_t estimated_entries;   /* hash_get_estimated_entries' figure for the dictionaries (0: default)                  */
    uint8_t  qual_codec;          /* 0: as the reference decides (above); GZ_CODEC_NONE: --no-domqual; GZ_CODEC_DOMQ: --force-domq    */
    uint64_t vb_size;             /* segconf.vb_size, the size the caller cuts VBlocks to (src/segconf.c:152-206). A VBlock whose text is
                                     not longer than MIN (4 MB, vb_size / 2) tests codecs for itself but does not set them for the
                                     file (src/codec.c:352: "don't let tiny VBs set the codec for everyone"). 0: every VBlock may    */
    uint8_t  record_lines;        /* 0 / 4: FASTQ, a record is 4 lines, the items are line 1's. 1: a record is ONE line (SAM: sam_seg_txt_line,
                                     src/sam_seg.c) - the items are the line's (tab-separated fields; QNAME further by its flavor) and SEQ /
                                     QUAL are the items seq_item / qual_item                                                          */
    uint8_t  seq_item, qual_item;
    uint32_t n_samples;           /* VCF: every record carries this many samples behind its 9 fixed fields (items 0-8 of the line; the plan's
                                     separators are the 9 tabs), each n_subfields ':'-separated FORMAT subfields (0: no samples)           */
    uint8_t  n_subfields;
    uint8_t  line3_empty;         /* segconf.line3 == L3_EMPTY: line 3 is "+" alone, takes no context (the '+' is a prefix of the TOPLEVEL
                                     container) and anything else there is an error (fastq_seg_LINE3, src/fastq_desc.c:33-37)          */
    uint8_t  seq_pad;             /* 0 (FASTQ) or 4: SAM's verbatim SEQ - every read's bases in NONREF.local are followed by 'A's up to a multiple of
                                     4, so that every read starts a new byte of the 2-bit packing (sam_seg_SEQ_pad_nonref, src/sam_seq.c:224-229,746-751) */
    uint8_t  vb_1_not_representative;   /* DTP (vb_1_not_representative) of the data type (src/data_types.h:57,148-152: VCF 0b110, SAM / BAM 0b100,
                                     FASTQ 0): bit 0 field contexts, bit 1 DTYPE_1, bit 2 DTYPE_2 (dict_id[0] >> 6: 0, 2 or 3, 1; src/dict_id.h:
                                     15-17). The beginning of such a file may not speak for the rest (src/codec.c:199-209), so for these contexts
                                     VBlock 1 does not set the file's LOCAL codec unless it is its file's last (:358-362), and VBlock 10 tests
                                     local and b250 again and sets what it finds (RETEST_VB_I, :22,274-277) - contexts with a codec given
                                     in the plan excepted                                                                              */
} GzFastqPlan;
typedef struct {
    uint64_t text_off, text_len;  /* in: the VBlock's slice of the text: whole reads                                      */
    uint32_t vblock_i;            /* in: 1-based; VBlocks of one call must be in ascending order                          */
    int32_t  r1;                  /* in: index within this call of the R1 VBlock an R2 VBlock pairs with (< own index), -1 */
    uint32_t n_reads;             /* out                                                                                  */
    int32_t  status;              /* out: GZ_OK, GZ_ERR_CORRUPT (not FASTQ / a line 1 that does not fit the container)     */
    uint8_t *z_data; uint64_t z_len;                     /* out, device: SEC_VB_HEADER + sections; valid until the next call */
    uint8_t *seq_packed; uint64_t seq_packed_len;        /* out, device: NONREF 2 bits per base                             */
    uint64_t n_bases; uint32_t seq_has_x; uint32_t n_sections;
    uint32_t seq_section_index;   /* out: where among the VBlock's sections (0 = right behind the VB header) the NONREF local section
                                     belongs (a15) once the host's sub-codec has made it: gz_vb_insert_section                      */
    uint32_t flags;               /* in: GZ_VB_LAST_OF_FILE                                                                */
} GzFastqVB;
#define GZ_VB_LAST_OF_FILE 1u     /* vb->is_last_vb_in_txt_file (src/codec.c:361; only looked at with vb_1_not_representative)         */
typedef struct GzZipFile GzZipFile;   /* z_file for this path: the file-level contexts and committed codecs */
GzZipFile *gz_zip_open (GzHandle *h, const GzFastqPlan *plan);
/* The host's candidates in the driver's codec assignment (a8 in full, see gz_codec_assign_best_ex). With these set, every trial the
 * driver makes (the first VBlock with >= 50 bytes of a context's stream, VBlock 10's second look ...) hands the SAME sample to
 * `trial` (host memory), which appends rows for the candidates it ran - codec id, PAYLOAD size (the 28-byte header is added here),
 * clock_us - and returns how many; a context whose winner is one of them has its sections compressed by `compress` (host in, host
 * out, *out_len in: capacity, out: payload bytes; return 0 on success) and framed like any other. Both run on the calling thread.
 * mode / clock_ns_per_byte as in gz_codec_assign_best_ex. NULL: back to the device candidates alone. While host candidates are set
 * the QUAL streams are not coded ahead of the merge (their trial needs the host's rows). */
typedef struct {
    int  (*trial) (void *user, const uint8_t dict_id[8], int is_local, const uint8_t *sample, uint32_t sample_len, GzCodecTest *rows, int max_rows);
    int  (*compress) (void *user, int codec, const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t *out_len);
    void  *user;
    const float *clock_ns_per_byte;
    int    mode;
} GzHostCodecs;
int gz_zip_set_host_codecs (GzZipFile *f, const GzHostCodecs *hc);
void       gz_zip_close (GzZipFile *f);
/* text: device, text_len bytes, with at least 16 writable bytes of slack behind them (a SNIP_LOOKUP byte is parked there);
 * < 4 GB per call. Synchronous (three waits inside: line index, seg results for the merge, z lengths). */
int gz_fastq_zip_vblocks (GzZipFile *f, uint8_t *text, uint64_t text_len, GzFastqVB *vbs, int n_vbs);
/* The same in three phases, for a file whose VBlocks are dealt out to several processes (one per GPU): the ordered
 * dictionary merge is the path's one exchange step. Every process calls gz_fastq_zip_seg on ITS VBlocks and gets a "merge
 * blob" (host memory owned by f: per VBlock and context the new words, counts and the handful of facts
 * ctx_merge_in_one_vctx reads); the processes exchange blobs (tiny: new words only) and every one of them hands ALL blobs to
 * gz_fastq_zip_merge, which replays the merge of the whole call in vblock_i order - so all processes end up with identical
 * dictionaries and each with the word indices of its own VBlocks - generates b250s / locals and returns this process'
 * codec "votes" (context, local / b250, vblock_i, codec) for contexts the file has no codec for yet; after exchanging the
 * votes, gz_fastq_zip_finish commits for every such context the vote with the lowest vblock_i (what a serial run commits,
 * src/codec.c:352-363) and writes the sections. vblock_i are global: unique over all processes and calls and ascending within a
 * call (a later call may fill numbers an earlier one left out: a streamed pair of files numbers R1 1..N and R2 N+1..2N,
 * src/writer.c:318-322, and each call holds some VBlocks of both; the call that holds the lowest number not merged yet is the
 * one that decides what the file's first VBlock decides); an R2 VBlock and its R1 VBlock belong to the same process. gz_fastq_zip_vblocks == the three phases with the own blob only. */
int gz_fastq_zip_seg (GzZipFile *f, uint8_t *text, uint64_t text_len, GzFastqVB *vbs, int n_vbs, const void **blob_out, uint64_t *blob_len_out);
int gz_fastq_zip_merge (GzZipFile *f, const void *const *blobs, const uint64_t *blob_lens, int n_blobs, const void **votes_out, uint64_t *votes_len_out);
int gz_fastq_zip_finish (GzZipFile *f, const void *const *votes, const uint64_t *votes_lens, int n_votes);
/* The same with up to TWO calls in flight, for a stream of calls on one file: _begin = gz_fastq_zip_vblocks without its last wait
 * (text, vbs and what vbs points at must stay as they are until the call has been ended); _end waits for the OLDEST call begun and
 * fills in its vbs[] (z_data valid until that set of buffers is used again: two calls later). The next call's seg and merge phases
 * and its coders run while the previous call's long streams are still being coded - the reference's VBlocks of different ages in
 * flight on its compute threads. Merges happen in the order of the begins: the bytes are those of one call at a time. */
int gz_fastq_zip_begin (GzZipFile *f, uint8_t *text, uint64_t text_len, GzFastqVB *vbs, int n_vbs);
int gz_fastq_zip_end (GzZipFile *f);
/* Speculation (no counterpart in the reference; results are the same with and without): a handle remembers which coder the QUAL
 * stream of its previous file was given by codec_assign_best_codec. The next file's long QUAL streams are handed to that coder as soon
 * as they are gathered - before the new file's own trial compressions (which still decide) are through. A trial that chooses otherwise
 * discards that work. Done by itself only where it pays: calls of at most 64 VBlocks whose plain (not CODEC_DOMQ) QUAL streams reach 5 M
 * scores (the long pole of such a call). hits / misses of the handle the file was opened on; environment: GZ_ZIP_NO_SPECULATION=1 never,
 * GZ_ZIP_SPECULATION=always whatever the sizes. */
void gz_zip_speculation (const GzZipFile *f, uint32_t *hits, uint32_t *misses);
/* Predicted coding (no counterpart in the reference either; same bytes with and without): in a call whose contexts still wait for their
 * trial compressions (a file's first call), every such stream is handed to the coders at once with a PREDICTED codec, on the second handle,
 * beside the trials - the codec the handle's previous file gave the same (dict_id, local | b250): a handle that has seen a file of the kind
 * (GZ_ZIP_PRIOR_ONLY=1 forgets it; GZ_ZIP_PREDICTION=prior: a cold handle uses a built-in prior by kind of stream - b250: ARTB; integers: RANb
 * for one byte, ARTW for more; QUAL: ARTB, through CODEC_DOMQ ARTb - which pays only where it is right: gz_zip.h has the measurements). The
 * trials decide as always; a section they confirm is only framed afterwards, the others are coded again with the rest. hits / misses:
 * sections kept / coded again, of the handle the file was opened on. GZ_ZIP_NO_PREDICTION=1: never. Not done while the long QUAL streams of
 * the call are being coded ahead (the second handle is theirs) or with the host's candidates in the race. */
void gz_zip_prediction (const GzZipFile *f, uint32_t *hits, uint32_t *misses);
/* a new file with the same plan (fresh dictionaries and codecs; the device workspace is kept) */
int gz_zip_reset (GzZipFile *f);
/* the z_data of the last call's VBlocks one after the other into dst (device) - what is handed to the writer
 * (zfile_output_processed_vb_ext, src/zfile.c:1160) or to the gather to the writer rank; offsets_host: n_vbs + 1 entries */
int gz_fastq_zip_collect (GzZipFile *f, const GzFastqVB *vbs, int n_vbs, uint8_t *dst, uint64_t cap, uint64_t *offsets_host);
/* the file-level context of plan context i (for the global area writer / inspection) */
GzZctx *gz_zip_zctx (GzZipFile *f, uint32_t ctx_i);
/* section order of one VBlock (a15): given n contexts' (did_i, local_dep, has_local, local_is_ston_only, has_b250) returns the
 * order of sections as indices 2 * i (local of context i) / 2 * i + 1 (its b250); returns the count (src/zip.c:247-342,565-585) */
typedef struct { uint16_t did_i; uint8_t local_dep, has_local, ston_only_local, has_b250; } GzSecOrderIn;
uint32_t gz_section_order (const GzSecOrderIn *ctxs, uint32_t n, uint32_t vblock_i, uint32_t *order_out);

/* batched / asynchronous forms used by the driver (all pointers device unless noted) */
/* (gz_tokenize_column_n reads the snips 16 bytes at a time: 15 readable bytes are needed behind the last snip of the text) */
int gz_tokenize_column_n (GzHandle *h, const uint8_t *text, const uint32_t *off, const uint32_t *len, uint32_t n,
                          const char *seps, const uint8_t *sep_counts, uint32_t n_seps, uint32_t *item_off, uint32_t *item_len, uint32_t *n_bad_dev);
typedef struct {
    const uint8_t *text; const uint32_t *off, *len; uint32_t n; uint32_t nothing_char; uint32_t lookup_off;
    uint32_t mode;                /* 0: seg_integer_or_not; 1: seg_self_delta (values = deltas against the previous snip)   */
    uint32_t *snip_off, *snip_len; int64_t *values; uint8_t *is_nothing; uint64_t *n_values_dev; int32_t *status_dev;
} GzIntColJob;
int gz_int_columns (GzHandle *h, const GzIntColJob *jobs, int n_jobs);
/* N1 for tab-separated data types (VCF, SAM). gz_byte_index: after[0] = 0, after[k + 1] = the position behind the k-th occurrence
 * of `byte` in the text (cap + 2 entries), result_dev->n_lines = the number of occurrences (status 0 if > cap): the newline index of
 * gz_text_lines for any byte. A SAM line's fields are then gz_tokenize_column_n with eleven tabs (the twelfth item: the optional
 * fields). gz_vcf_sample_columns: vcf_seg_samples' split (src/vcf_samples.c:1601) - for the data lines given (not the header), the
 * FORMAT subfields of every sample by position: item [j][line * n_samples + s] = subfield j of sample s (':' separated; a sample that
 * leaves trailing subfields out has them `missing`: length 0 and, if `missing` is given, a 1 there - the mask
 * gz_local_generate_partial takes). Which context a position feeds is the line's FORMAT field (item 8 of the line, for the caller to
 * group lines by). *n_bad_dev counts lines that do not have 9 + n_samples fields and samples with more than n_subfields. */
int gz_byte_index (GzHandle *h, const uint8_t *text, uint64_t n_bytes, uint8_t byte, uint32_t *after, uint32_t cap, GzLinesResult *result_dev);
int gz_vcf_sample_columns (GzHandle *h, const uint8_t *text, const uint32_t *line_off, const uint32_t *line_len, uint32_t n_lines,
                           const uint32_t *tab_after, const GzLinesResult *tabs_dev, uint32_t n_samples, uint32_t n_subfields,
                           uint32_t *item_off, uint32_t *item_len, uint8_t *missing, uint32_t *n_bad_dev);
typedef struct { void *data; uint64_t n; const GzDynIntResult *dyn_dev; int32_t ltype; uint32_t *len_dev; } GzLocalJob;
int gz_local_generate_batch (GzHandle *h, const GzLocalJob *jobs, int n_jobs);
typedef struct { const uint8_t *seq; const uint64_t *n_dev; uint64_t n_max; uint8_t *packed; uint8_t *x; uint32_t *has_x_dev; uint64_t *packed_len_dev; } GzAcgtJob;
int gz_acgt_pack_batch (GzHandle *h, const GzAcgtJob *jobs, int n_jobs);

/* ---- N4: the global area of the file (HOST) -----------------------------------------------------------
 * zip_write_global_area (src/zip.c:416-507) for this path's contexts: SEC_DICT (dict_io_compress_dictionaries,
 * src/dict_io.c:45-193), SEC_COUNTS (ctx_compress_counts, src/context.c:1612-1651), the section list in file format
 * (sections_list_memory_to_file_format, src/sections.c:481-534) as the payload of SEC_GENOZIP_HEADER (src/sections.h:169-300),
 * and the footer (:303-307). The reference's own writer of SEC_GENOZIP_HEADER is not in its shipped sources (closed licence module): the
 * header follows the struct and what the reader does with it, the licence fields are zero. The reference's shipped genounzip reads
 * files written through these calls back into the original text (tests/test_e2e_genounzip.py). */
typedef struct GzZFile GzZFile;
GzZFile *gz_zfile_create (uint16_t data_type /* DT_FASTQ = 3 */, uint32_t vb_size_bytes);
void     gz_zfile_destroy (GzZFile *zf);
/* sections_add_to_list (src/sections.c:105-135) for a finished VBlock written to the file at file_offset; z: HOST copy of z_data */
int gz_zfile_add_vblock (GzZFile *zf, const uint8_t *z, uint64_t z_len, uint64_t file_offset, uint8_t comp_i, uint32_t num_lines);
int gz_zfile_write_global_area (GzZFile *zf, GzHandle *h, GzZctx *const *zctx, const uint8_t *dict_ids /* n_ctx x 8 */,
                                const uint8_t *counts_section /* n_ctx flags or NULL */, uint32_t n_ctx, uint64_t file_offset,
                                uint64_t recon_size, uint64_t num_lines, const char *created,
                                uint8_t *out_host, uint64_t out_cap, uint64_t *out_len);
int gz_codec_assign_best_host (GzHandle *h, const uint8_t *in_host, uint32_t in_len);
/* SEC_TXT_HEADER of one component (txtheader_compress + zfile_update_txt_header_section_header, src/txtheader.c:46-111,
 * src/zfile.c:1068-1105; layout src/sections.h:308-327): FASTQ has no header text, the section is its 400-byte header alone and is
 * written in front of the component's VBlocks. pair: 0 not paired, 1 R1, 2 R2 (FlagsTxtHeader.pair); flav_prop: NUM_QTYPES x 2 bytes
 * (QnameFlavorProp, src/sections.h:296-305) or NULL. out_host: 400 bytes. The section joins zf's list at file_offset. */
#define GZ_TXT_HEADER_LEN 400
int gz_zfile_add_txt_header (GzZFile *zf, uint8_t comp_i, uint8_t pair, const char *txt_filename, uint64_t txt_data_size, uint64_t txt_num_lines,
                             uint32_t max_lines_per_vb, const uint8_t *flav_prop, uint32_t n_flav_prop, uint64_t file_offset, uint8_t *out_host);
/* the same for a component WITH header text (VCF: ## lines and #CHROM, SAM: @ lines): the text is the section's payload, stored (CODEC_NONE);
 * txt_data_size counts it. out_host: GZ_TXT_HEADER_LEN + header_len bytes (GZ_TOO_SMALL: *out_len says how many) */
int gz_zfile_add_txt_header_text (GzZFile *zf, uint8_t comp_i, uint8_t pair, const char *txt_filename, uint64_t txt_data_size, uint64_t txt_num_lines,
                                  uint32_t max_lines_per_vb, const uint8_t *flav_prop, uint32_t n_flav_prop, uint64_t file_offset,
                                  const uint8_t *header_text, uint32_t header_len, uint8_t *out_host, uint64_t out_cap, uint64_t *out_len);
/* What the reader needs of the file as a whole that is not in the sections (SectionHeaderGenozipHeader, src/sections.h:169-300):
 * set before gz_zfile_write_global_area. paired: the file holds an R1 / R2 pair (z_flags.dt_specific, v14_is_paired);
 * std_seq_len / std_seq_len_r2: segconf.std_seq_len, std_seq_lR2 (longest SEQ of the sampled reads, src/fastq.c:711) */
int gz_zfile_set_fastq (GzZFile *zf, uint8_t num_txt_files, uint8_t paired, uint32_t std_seq_len, uint32_t std_seq_len_r2);
/* A section made on the host into a finished VBlock's z_data (host copies): the 40-byte SectionHeaderCtx is built here (lengths,
 * adler32 of the payload, vblock_i of the VBlock), the section goes in front of the VBlock's section number `index` (GzFastqVB.
 * seq_section_index for NONREF, whose payload is the host's CODEC_ACGT sub-codec output: codec = GZ_CODEC_ACGT, sub_codec = LZMA 4 /
 * NONE 1, src/codec_acgt.c:157-169) and z_data_bytes of the VB header is patched (zfile_update_compressed_vb_header, src/zfile.c:1139).
 * out_host may not overlap z. Returns GZ_TOO_SMALL with *out_len = the size needed. */
int gz_vb_insert_section (const uint8_t *z, uint64_t z_len, uint32_t index, const uint8_t dict_id[8], uint8_t codec, uint8_t sub_codec,
                          uint8_t flags, uint8_t ltype, uint8_t param, const uint8_t *payload, uint32_t payload_len, uint32_t uncompressed_len,
                          uint8_t *out_host, uint64_t out_cap, uint64_t *out_len);

/* ---- N3: CODEC_DOMQ's pre-transform (src/codec_domq.c:69-134,139-249,347-503) -------------------------------
 * The QUAL lines of a VBlock -> QUAL (non-dominant normalised scores + `no_doms` markers), DOMQRUNS (run lengths of the
 * dominant score, continuing across lines), QUALMPLX (a byte per line: row in the denormalisation table, | 0x80 = diverse),
 * DIVRQUAL (normalised scores of lines whose dominant score covers < 85 %), and the denormalisation table num_doms x
 * num_norm_qs (the caller base64-codes it into DOMQRUNS' dictionary; QUAL's section param = num_norm_qs | 0x80, its ltype
 * LT_CODEC, codec DOMQ with the sub-codec that coded the stream). Lines of length 0 take no part. All pointers device,
 * asynchronous. Capacities with B = sum of the lengths: qual 2 B + 16, runs B + B / 254 + 16, mplx n + 16, divr B + 16. */
enum { GZ_CODEC_DOMQ = 13, GZ_LT_CODEC_ = 13 };
/* codecs that only ever appear in section headers of this path: CODEC_ACGT (NONREF: the host's sub-codec makes the payload), CODEC_XCGT
 * (NONREF_X: the name of the section, its stream coded by the sub-codec), LZMA (src/genozip.h:325-360) */
enum { GZ_CODEC_ACGT = 10, GZ_CODEC_XCGT = 11 };                  /* (GZ_CODEC_LZMA = 4: with the host codecs above) */
typedef struct {
    uint64_t qual_len, runs_len, mplx_len, divr_len;
    uint32_t num_doms, num_norm_qs, has_diverse, all_diverse;   /* all_diverse: QUAL is the single byte 'X', sub-codec NONE (:490-494) */
    int32_t  status;              /* 1 ok; -5: a score outside ' '..'~'                                                   */
    uint32_t reserved;
    uint8_t  denorm[95 * 95 + 7];
} GzDomqResult;
typedef struct {
    const uint8_t *text; const uint32_t *off, *len; uint32_t n;
    uint8_t *qual, *runs, *mplx, *divr;
    GzDomqResult *result_dev;
    const uint32_t *only_if_dev;  /* optional: the job is skipped (nothing written, *result_dev included) when this device word is 0 -
                                     e.g. the fit flag gz_domq_fit left for the VBlock that decides for the file                  */
} GzDomqJob;
int gz_domq_columns (GzHandle *h, const GzDomqJob *jobs, int n_jobs);
/* codec_domq_qual_data_is_a_fit_for_domq (:69-134) per VBlock: *fit_dev = 1 when more than half of the first (up to) 10 lines
 * have a score that fills more than half of their first 2500 / lines bytes */
typedef struct { const uint8_t *text; const uint32_t *off, *len; uint32_t n; uint32_t *fit_dev; } GzDomqFitJob;
int gz_domq_fit (GzHandle *h, const GzDomqFitJob *jobs, int n_jobs);

#ifdef __cplusplus
}
#endif
#endif /* GENOZIP_AMD_H */
