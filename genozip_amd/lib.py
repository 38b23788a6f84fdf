"""ctypes binding of include/genozip_amd.h (libgenozip_amd.so).

There is deliberately no fallback: if the HIP library is missing, or no GPU is visible, loading / creating a handle
raises. (tests/emul builds the same sources against a CPU stand-in of the HIP runtime for logic tests; that library
is only ever loaded by tests, through the explicit `path` argument.)
"""
import ctypes as C
import os

# The library runs its kernels on ~14 HIP streams per pair of handles, some of which hold one-thread gate kernels that wait for the range
# coder's long chains. The HIP runtime maps all streams onto 4 hardware queues by default, and whatever shares a queue with a gate waits
# with it. 8 queues: measured 4 / 6 / 8 / 12 / 16 / 24 - with one call in flight 8 and 12 are the same (default step 46.2 vs 46.5 ms, streamed
# 167 vs 167 ms per two calls), with two calls in flight 12 is much slower (214 vs 177 ms: beyond 8 the runtime time-slices the queues,
# whole queues stand still for ~11 ms at a time) - DESIGN.md section 0. The variable is read when the HIP runtime starts: it has to be in the
# environment before the first HIP call of the process (the library's own constructor sets it too, for C hosts); a value the host set wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libgenozip_amd.so")

# codec ids == file format values (reference: src/genozip.h:325-360)
CODEC_UNKNOWN, CODEC_NONE = 0, 1
CODEC_RANB, CODEC_RANW, CODEC_RANb, CODEC_RANw = 6, 7, 8, 9
CODEC_ARTB, CODEC_ARTW, CODEC_ARTb, CODEC_ARTw = 16, 17, 18, 19
SIMPLE_CODECS = (CODEC_RANB, CODEC_RANW, CODEC_RANb, CODEC_RANw, CODEC_ARTB, CODEC_ARTW, CODEC_ARTb, CODEC_ARTw)
CODEC_NAMES = {0: "UNKNOWN", 1: "NONE", 6: "RANB", 7: "RANW", 8: "RANb", 9: "RANw", 16: "ARTB", 17: "ARTW", 18: "ARTb", 19: "ARTw"}

SEC_VB_HEADER, SEC_B250, SEC_LOCAL = 9, 11, 12

LT_INT8, LT_UINT8, LT_INT16, LT_UINT16, LT_INT32, LT_UINT32, LT_INT64, LT_UINT64 = 1, 2, 3, 4, 5, 6, 7, 8
LT_FLOAT32, LT_FLOAT64, LT_BLOB, LT_BITMAP = 9, 10, 11, 12
LT_UINT8_TR, LT_UINT16_TR, LT_UINT32_TR = 14, 15, 16

GZ_OK, GZ_TOO_SMALL, GZ_ERR, GZ_ERR_NO_DEVICE, GZ_ERR_ARG, GZ_ERR_HIP, GZ_ERR_CORRUPT = 1, 0, -1, -2, -3, -4, -5


class GzStream(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("in_len", C.c_uint32), ("in_len_dev", C.c_void_p), ("out", C.c_void_p),
                ("out_cap", C.c_uint32), ("codec", C.c_int32), ("out_len", C.c_uint32), ("status", C.c_int32), ("out_len_dev", C.c_void_p)]


class GzB250Job(C.Structure):
    _fields_ = [("seg", C.c_void_p), ("seg_len", C.c_uint32), ("seg_len_dev", C.c_void_p), ("ol_nodes_len", C.c_uint32),
                ("node2word", C.c_void_p), ("n_new_nodes", C.c_uint32), ("out", C.c_void_p), ("out_len_dev", C.c_void_p),
                ("status_dev", C.c_void_p), ("r1", C.c_void_p), ("r1_len_dev", C.c_void_p)]


class GzColumnResult(C.Structure):
    _fields_ = [("dict_len", C.c_uint64), ("b250_len", C.c_uint64), ("b250_count", C.c_uint64), ("n_new", C.c_uint32),
                ("all_the_same", C.c_uint32), ("status", C.c_int32), ("reserved", C.c_uint32)]


class GzColumnJob(C.Structure):
    _fields_ = [("text", C.c_void_p), ("off", C.c_void_p), ("len", C.c_void_p), ("n", C.c_uint32),
                ("ol_dict", C.c_void_p), ("ol_char_index", C.c_void_p), ("ol_snip_len", C.c_void_p), ("n_ol", C.c_uint32),
                ("node_index", C.c_void_p), ("dict", C.c_void_p), ("dict_cap", C.c_uint64),
                ("node_char_index", C.c_void_p), ("node_snip_len", C.c_void_p), ("counts", C.c_void_p),
                ("b250", C.c_void_p), ("result_dev", C.c_void_p)]


class GzDynIntResult(C.Structure):
    _fields_ = [("len", C.c_uint64), ("ltype", C.c_int32), ("width", C.c_uint32), ("order", C.c_uint32), ("reserved", C.c_uint32)]


class GzDynIntJob(C.Structure):
    _fields_ = [("values", C.c_void_p), ("is_nothing", C.c_void_p), ("n", C.c_uint64), ("nothing_char", C.c_uint32),
                ("out", C.c_void_p), ("result_dev", C.c_void_p), ("n_dev", C.c_void_p)]


class GzBlobJob(C.Structure):
    _fields_ = [("text", C.c_void_p), ("off", C.c_void_p), ("len", C.c_void_p), ("n", C.c_uint32), ("add_nul", C.c_uint32),
                ("out", C.c_void_p), ("out_len_dev", C.c_void_p),
                ("pre", C.c_uint8 * 4), ("pre_len", C.c_uint32), ("pad_to", C.c_uint32), ("pad_byte", C.c_uint32), ("item_off", C.c_void_p), ("item_len", C.c_void_p)]


class GzBamResult(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("text_len", C.c_uint64), ("status", C.c_int32), ("first_bad", C.c_uint32), ("n_rewalked", C.c_uint32), ("reserved", C.c_uint32)]


class GzSection(C.Structure):
    _fields_ = [("data", C.c_void_p), ("data_len", C.c_uint32), ("data_len_dev", C.c_void_p),
                ("section_type", C.c_uint8), ("codec", C.c_uint8), ("sub_codec", C.c_uint8), ("flags", C.c_uint8),
                ("ltype", C.c_uint8), ("param", C.c_uint8), ("b250_size_or_nothing_char", C.c_uint8),
                ("dict_id", C.c_uint8 * 8), ("precompressed", C.c_uint8), ("raw_len", C.c_uint32), ("hdr_codec", C.c_uint8)]


class GzVBlock(C.Structure):
    _fields_ = [("vblock_i", C.c_uint32), ("recon_size", C.c_uint32), ("longest_line_len", C.c_uint32),
                ("longest_seq_len", C.c_uint32), ("digest", C.c_uint8 * 16), ("vb_flags", C.c_uint8),
                ("sections", C.POINTER(GzSection)), ("n_sections", C.c_uint32), ("z_data", C.c_void_p),
                ("z_cap", C.c_uint64), ("z_len", C.c_uint64), ("status", C.c_int32), ("mark_section", C.c_uint32), ("mark_index", C.c_uint32)]


class GzMergeJob(C.Structure):
    _fields_ = [("vblock_i", C.c_uint32), ("n_ol", C.c_uint32), ("n_new", C.c_uint32),
                ("dict", C.c_void_p), ("node_char_index", C.c_void_p), ("node_snip_len", C.c_void_p), ("counts", C.c_void_p),
                ("can_have_singletons", C.c_uint8), ("flags", C.c_uint8), ("no_drop_b250", C.c_uint8), ("pair2_identical", C.c_uint8),
                ("b250_len", C.c_uint64), ("local_len", C.c_uint64), ("b250_r1_len", C.c_uint64), ("local_r1_len", C.c_uint64),
                ("ats_node_index", C.c_int32), ("lcodec", C.c_uint8), ("bcodec", C.c_uint8), ("dropped_b250", C.c_uint8), ("reserved", C.c_uint8),
                ("node2word", C.c_void_p), ("ston_local", C.c_void_p), ("ston_cap", C.c_uint64), ("ston_len", C.c_uint64), ("n_stons", C.c_uint32)]


class GzZctxView(C.Structure):
    _fields_ = [("dict", C.c_void_p), ("dict_len", C.c_uint64), ("char_index", C.c_void_p), ("snip_len", C.c_void_p), ("counts", C.c_void_p),
                ("n_words", C.c_uint32), ("hash_len", C.c_uint32), ("n_singletons", C.c_uint64), ("n_failed_singletons", C.c_uint64),
                ("flags", C.c_uint8), ("rm_dict_all_the_same", C.c_uint8), ("lcodec", C.c_uint8), ("bcodec", C.c_uint8), ("all_the_same_wi", C.c_int32)]


class GzDomqResult(C.Structure):
    _fields_ = [("qual_len", C.c_uint64), ("runs_len", C.c_uint64), ("mplx_len", C.c_uint64), ("divr_len", C.c_uint64),
                ("num_doms", C.c_uint32), ("num_norm_qs", C.c_uint32), ("has_diverse", C.c_uint32), ("all_diverse", C.c_uint32),
                ("status", C.c_int32), ("reserved", C.c_uint32), ("denorm", C.c_uint8 * (95 * 95 + 7))]


class GzDomqJob(C.Structure):
    _fields_ = [("text", C.c_void_p), ("off", C.c_void_p), ("len", C.c_void_p), ("n", C.c_uint32), ("qual", C.c_void_p), ("runs", C.c_void_p),
                ("mplx", C.c_void_p), ("divr", C.c_void_p), ("result_dev", C.c_void_p), ("only_if_dev", C.c_void_p)]


class GzDomqFitJob(C.Structure):
    _fields_ = [("text", C.c_void_p), ("off", C.c_void_p), ("len", C.c_void_p), ("n", C.c_uint32), ("fit_dev", C.c_void_p)]


class GzFastqCtx(C.Structure):
    _fields_ = [("dict_id", C.c_uint8 * 8), ("did_i", C.c_uint16), ("kind", C.c_uint8), ("item", C.c_uint8), ("local_dep", C.c_uint8),
                ("flags", C.c_uint8), ("no_stons", C.c_uint8), ("lcodec", C.c_uint8), ("bcodec", C.c_uint8), ("pair_identical", C.c_uint8),
                ("pair_assisted_b250", C.c_uint8), ("nothing_char", C.c_uint8), ("snip", C.c_char_p), ("snip_len", C.c_uint32),
                ("con_len", C.c_uint32), ("per_sample", C.c_uint8), ("transposed", C.c_uint8), ("segs_per_line", C.c_uint8),
                ("r2_node", C.c_char_p), ("r2_node_len", C.c_uint32)]


class GzFastqPlan(C.Structure):
    _fields_ = [("ctxs", C.POINTER(GzFastqCtx)), ("n_ctxs", C.c_uint32), ("seps", C.c_char * 32), ("sep_counts", C.c_uint8 * 32),
                ("n_seps", C.c_uint32), ("paired", C.c_uint8), ("estimated_entries", C.c_uint32), ("qual_codec", C.c_uint8), ("vb_size", C.c_uint64),
                ("record_lines", C.c_uint8), ("seq_item", C.c_uint8), ("qual_item", C.c_uint8), ("n_samples", C.c_uint32), ("n_subfields", C.c_uint8), ("line3_empty", C.c_uint8),
                ("seq_pad", C.c_uint8), ("vb_1_not_representative", C.c_uint8)]


class GzFastqVB(C.Structure):
    _fields_ = [("text_off", C.c_uint64), ("text_len", C.c_uint64), ("vblock_i", C.c_uint32), ("r1", C.c_int32), ("n_reads", C.c_uint32),
                ("status", C.c_int32), ("z_data", C.c_void_p), ("z_len", C.c_uint64), ("seq_packed", C.c_void_p), ("seq_packed_len", C.c_uint64),
                ("n_bases", C.c_uint64), ("seq_has_x", C.c_uint32), ("n_sections", C.c_uint32), ("seq_section_index", C.c_uint32), ("flags", C.c_uint32)]


class GzSecOrderIn(C.Structure):
    _fields_ = [("did_i", C.c_uint16), ("local_dep", C.c_uint8), ("has_local", C.c_uint8), ("ston_only_local", C.c_uint8), ("has_b250", C.c_uint8)]


GZ_FQ_CONST, GZ_FQ_ITEM_TEXT, GZ_FQ_ITEM_INT, GZ_FQ_ITEM_DELTA, GZ_FQ_SEQ, GZ_FQ_QUAL, GZ_FQ_QUAL_AUX, GZ_FQ_TOPLEVEL, GZ_FQ_SEQ_SNIP, GZ_FQ_ITEM_EXPECT = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10

# every symbol include/genozip_amd.h declares (checked by tests/test_abi.py)
GzGetLineCB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32))

class GzCodecTest(C.Structure):
    """CodecTest (src/codec.c:122-126): one candidate's trial"""
    _fields_ = [("codec", C.c_int32), ("size", C.c_float), ("clock_us", C.c_float)]


GZ_HOST_TRIAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_uint8), C.c_uint32, C.POINTER(GzCodecTest), C.c_int)
GZ_HOST_COMPRESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_uint8), C.c_uint32, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32))


class GzHostCodecs(C.Structure):
    _fields_ = [("trial", GZ_HOST_TRIAL_FN), ("compress", GZ_HOST_COMPRESS_FN), ("user", C.c_void_p), ("clock_ns_per_byte", C.POINTER(C.c_float)), ("mode", C.c_int)]


ABI_SYMBOLS = (
    "gz_create", "gz_create_background", "gz_destroy", "gz_sync", "gz_last_error", "gz_last_warning", "gz_version", "gz_stream", "gz_profile", "gz_profile_get", "gz_profile_get_max",
    "gz_download", "gz_upload", "gz_dev_alloc", "gz_dev_free", "gz_emit_after", "gz_wait_for",
    "gz_codec_est_size", "gz_codec_compress_host", "gz_codec_uncompress_host",
    "gz_codec_compress_batch", "gz_codec_uncompress_batch", "gz_codec_assign_best",
    "gz_b250_generate", "gz_b250_generate_batch", "gz_local_generate", "gz_local_to_native",
    "gz_vb_z_bound", "gz_vb_compress_batch", "gz_vb_uncompress", "gz_vb_uncompress_many", "gz_adler32",
    "gz_acgt_packed_len", "gz_acgt_pack", "gz_acgt_unpack",
    "gz_ctx_seg_columns", "gz_dyn_int_columns", "gz_local_blob_columns",
    "gz_text_lines", "gz_fastq_records", "gz_tokenize_column", "gz_seg_integer_or_not",
    "gz_local_generate_partial", "gz_local_partial_to_native",
    "gz_zctx_create", "gz_zctx_destroy", "gz_hash_next_size_up", "gz_ctx_merge", "gz_zctx_view", "gz_zctx_commit_codec",
    "gz_zip_open", "gz_zip_close", "gz_fastq_zip_vblocks", "gz_fastq_zip_seg", "gz_fastq_zip_merge", "gz_fastq_zip_finish", "gz_zip_zctx", "gz_section_order",
    "gz_zip_reset", "gz_fastq_zip_collect",
    "gz_domq_columns", "gz_domq_fit", "gz_codec_compress_lines_host", "gz_byte_index", "gz_vcf_sample_columns", "gz_fastq_zip_begin", "gz_fastq_zip_end", "gz_zip_speculation", "gz_zip_prediction",
    "gz_zfile_create", "gz_zfile_destroy", "gz_zfile_add_vblock", "gz_zfile_write_global_area", "gz_codec_assign_best_host",
    "gz_zfile_add_txt_header", "gz_zfile_add_txt_header_text", "gz_zfile_set_fastq", "gz_vb_insert_section",
    "gz_tokenize_column_n", "gz_int_columns", "gz_local_generate_batch", "gz_acgt_pack_batch",
    "gz_bam_records", "gz_bam_to_sam",
    "gz_codec_assign_sort", "gz_codec_assign_best_ex", "gz_zip_set_host_codecs", "gz_debug_record_inv", "gz_codec_assign_rule", "gz_chain_fallbacks",
)


def load(path=None):
    path = path or DEFAULT_LIB
    if not os.path.exists(path):
        raise RuntimeError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(genozip_amd has no CPU fallback)" % path)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (SONAME libamdhip64.so.7). If it is
    # already loaded, the dynamic linker resolves our DT_NEEDED libamdhip64.so.7 to that copy; loaded the other way
    # round the process would hold two runtimes and the second one to initialise finds no device. So when torch
    # is going to be used for the HBM buffers, it must come first.
    if not os.path.basename(path).startswith("libgenozip_amd_emul"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(path)
    L.gz_create.restype = C.c_void_p
    L.gz_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_int)]
    L.gz_destroy.argtypes = [C.c_void_p]
    L.gz_sync.argtypes = [C.c_void_p]
    L.gz_last_error.restype = C.c_char_p
    L.gz_last_error.argtypes = [C.c_void_p]
    L.gz_last_warning.restype = C.c_char_p
    L.gz_last_warning.argtypes = [C.c_void_p]
    L.gz_version.restype = C.c_char_p
    L.gz_stream.restype = C.c_void_p
    L.gz_chain_fallbacks.argtypes = [C.c_void_p]
    L.gz_chain_fallbacks.restype = C.c_uint32
    L.gz_debug_record_inv.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.gz_codec_assign_rule.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32]
    L.gz_stream.argtypes = [C.c_void_p]
    L.gz_profile.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.gz_profile.restype = None
    L.gz_profile_get.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.gz_profile_get_max.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    L.gz_codec_est_size.restype = C.c_uint32
    L.gz_codec_est_size.argtypes = [C.c_int, C.c_uint64]
    L.gz_codec_compress_host.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_uint32, C.c_char_p, C.POINTER(C.c_uint32), C.c_int]
    L.gz_codec_uncompress_host.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint64]
    L.gz_codec_compress_batch.argtypes = [C.c_void_p, C.POINTER(GzStream), C.c_int]
    L.gz_codec_uncompress_batch.argtypes = [C.c_void_p, C.POINTER(GzStream), C.c_int]
    L.gz_codec_assign_best.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    L.gz_b250_generate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.gz_b250_generate_batch.argtypes = [C.c_void_p, C.POINTER(GzB250Job), C.c_int]
    L.gz_local_generate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
    L.gz_local_to_native.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
    L.gz_vb_z_bound.restype = C.c_uint64
    L.gz_vb_z_bound.argtypes = [C.POINTER(GzSection), C.c_uint32]
    L.gz_vb_compress_batch.argtypes = [C.c_void_p, C.POINTER(GzVBlock), C.c_int]
    L.gz_vb_uncompress.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint32)]
    L.gz_vb_uncompress_many.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                        C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint32)]
    L.gz_adler32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32)]
    L.gz_acgt_packed_len.restype = C.c_uint64
    L.gz_acgt_packed_len.argtypes = [C.c_uint64]
    L.gz_acgt_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    L.gz_acgt_unpack.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.gz_ctx_seg_columns.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.gz_dyn_int_columns.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.gz_local_blob_columns.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.gz_text_lines.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.gz_bam_records.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p]
    L.gz_bam_to_sam.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.gz_fastq_records.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32] + [C.c_void_p] * 9
    L.gz_seg_integer_or_not.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32] + [C.c_void_p] * 5
    for f in (L.gz_local_generate_partial, L.gz_local_partial_to_native):
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    L.gz_tokenize_column.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32,
                                     C.c_void_p, C.c_void_p, C.c_void_p]
    L.gz_emit_after.argtypes = [C.c_void_p, C.c_void_p]
    L.gz_wait_for.argtypes = [C.c_void_p, C.c_void_p]
    L.gz_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.gz_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.gz_dev_alloc.restype = C.c_void_p
    L.gz_dev_alloc.argtypes = [C.c_void_p, C.c_uint64]
    L.gz_dev_free.argtypes = [C.c_void_p, C.c_void_p]
    L.gz_zctx_create.restype = C.c_void_p
    L.gz_zctx_create.argtypes = [C.c_uint32]
    L.gz_zctx_destroy.argtypes = [C.c_void_p]
    L.gz_hash_next_size_up.restype = C.c_uint32
    L.gz_hash_next_size_up.argtypes = [C.c_uint64]
    L.gz_ctx_merge.argtypes = [C.c_void_p, C.POINTER(GzMergeJob)]
    L.gz_zctx_view.argtypes = [C.c_void_p, C.POINTER(GzZctxView)]
    L.gz_zctx_commit_codec.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.gz_zip_open.restype = C.c_void_p
    L.gz_zip_open.argtypes = [C.c_void_p, C.POINTER(GzFastqPlan)]
    L.gz_zip_close.argtypes = [C.c_void_p]
    L.gz_fastq_zip_vblocks.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(GzFastqVB), C.c_int]
    L.gz_fastq_zip_seg.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(GzFastqVB), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.gz_fastq_zip_merge.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.gz_fastq_zip_finish.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_int]
    L.gz_domq_columns.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.gz_fastq_zip_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
    L.gz_fastq_zip_end.argtypes = [C.c_void_p]
    L.gz_zip_speculation.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.gz_zip_speculation.restype = None
    L.gz_zip_prediction.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.gz_zip_prediction.restype = None
    L.gz_byte_index.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint8, C.c_void_p, C.c_uint32, C.c_void_p]
    L.gz_vcf_sample_columns.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gz_codec_compress_lines_host.restype = C.c_int
    L.gz_codec_compress_lines_host.argtypes = [C.c_void_p, C.c_int, GzGetLineCB, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32), C.c_int]
    L.gz_domq_fit.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.gz_zfile_create.restype = C.c_void_p
    L.gz_zfile_create.argtypes = [C.c_uint16, C.c_uint32]
    L.gz_zfile_destroy.argtypes = [C.c_void_p]
    L.gz_zfile_add_vblock.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint8, C.c_uint32]
    L.gz_zfile_write_global_area.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64,
                                             C.c_char_p, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.gz_codec_assign_best_host.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32]
    L.gz_codec_assign_sort.argtypes = [C.POINTER(GzCodecTest), C.c_int, C.c_int]
    L.gz_codec_assign_best_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(GzCodecTest), C.c_int, C.POINTER(C.c_float), C.c_int, C.POINTER(GzCodecTest)]
    L.gz_zip_set_host_codecs.argtypes = [C.c_void_p, C.POINTER(GzHostCodecs)]
    L.gz_zfile_add_txt_header.argtypes = [C.c_void_p, C.c_uint8, C.c_uint8, C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint64, C.c_char_p]
    L.gz_zfile_add_txt_header_text.argtypes = [C.c_void_p, C.c_uint8, C.c_uint8, C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint64,
                                               C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.gz_zfile_set_fastq.argtypes = [C.c_void_p, C.c_uint8, C.c_uint8, C.c_uint32, C.c_uint32]
    L.gz_vb_insert_section.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_char_p, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint8, C.c_char_p, C.c_uint32,
                                       C.c_uint32, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.gz_zip_reset.argtypes = [C.c_void_p]
    L.gz_fastq_zip_collect.argtypes = [C.c_void_p, C.POINTER(GzFastqVB), C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.gz_zip_zctx.restype = C.c_void_p
    L.gz_zip_zctx.argtypes = [C.c_void_p, C.c_uint32]
    L.gz_section_order.restype = C.c_uint32
    L.gz_section_order.argtypes = [C.POINTER(GzSecOrderIn), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    L.gz_tokenize_column_n.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_uint32,
                                       C.c_void_p, C.c_void_p, C.c_void_p]
    L.gz_int_columns.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.gz_local_generate_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.gz_acgt_pack_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    return L
