/* oracle/ref_merge_shim.c -- TEST INFRASTRUCTURE ONLY (see oracle/Makefile, target "ref").
 *
 * Pins the LOOP of SURVEY 8a row a4. The reference's own src/context.c is compiled as part of THIS translation unit (the #include below
 * names the file where it lies under /root/reference - nothing is copied), so that its static ctx_merge_in_one_vctx (:938-1079),
 * ctx_commit_node (:269-316), ctx_insert_to_dict (:50-71) and ctx_drop_all_the_same (:795-871) run as the reference wrote them, on top of
 * the reference's own src/hash.c (hash_global_get_entry, the singleton tables), src/seg.c (seg_add_to_local_fixed_do: where a singleton's
 * text goes), src/b250.c, src/strings.c compiled in place by the Makefile into oracle/_ref/libmergeref.so. tests/golden/merge_golden.json
 * is generated from it (tests/golden/make_merge_golden.py).
 *
 * What this file supplies: a File with one file-level context and a VBlock with one context, filled by hand from plain arguments (the
 * fields the merge reads, by name); mutexes that do nothing (one thread); allocation of a Buffer; the option structs. Everything else the
 * linked objects import and never reach is named by the generated stubs file (oracle/gen_ref_stubs.py). Nothing here is product code.
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <string.h>

#include "context.c"                /* the reference's own file, in place (-iquote $(REF)/src) */

#include "b250.h"
#include "seg.h"
#include "local_type.h"

Flags flag;
SegConf segconf;
VBlockP evb;
FileP z_file, txt_file;
FILE *info_stream;
CommandType primary_command = ZIP;
const LocalTypeDesc lt_desc[NUM_LOCAL_TYPES] = LOCALTYPE_DESC;
DataTypeProperties dt_props[NUM_DATATYPES], dt_props_def;
DataTypeFields dt_fields[NUM_DATATYPES] = { [DT_FASTQ] = { .num_fields = 4 /* SHIM_DID + 1 */ } };
static int shim_pair2;
__attribute__((constructor)) static void shim_defaults (void) { flag.show_time_comp_i = COMP_NONE; flag.show_stats_comp_i = COMP_NONE; flag.command = ZIP; info_stream = stderr; }

void buf_alloc_do (VBlockP vb, BufferP buf, uint64_t requested_size, float grow_at_least_factor, rom name, FUNCLINE)
{
    if (buf->size >= requested_size && buf->data) return;
    uint64_t sz = requested_size * (grow_at_least_factor > 1 ? grow_at_least_factor : 1) + 64;
    char *m = realloc (buf->memory, sz + 16);
    if (!m) abort ();
    if (sz > buf->size) memset (m + 8 + buf->size, 0, sz - buf->size);       /* (buf_alloc_zero callers rely on new memory being zero: buf_alloc_do's own init_type is theirs to apply - all our buffers start zeroed) */
    buf->memory = m; buf->data = m + 8; buf->size = sz; buf->vb = vb; buf->name = name; buf->type = BUF_REGULAR;
}
void buf_free_do (BufferP buf, FUNCLINE) { free (buf->memory); memset (buf, 0, sizeof (*buf)); }
void buf_destroy_do (BufferP buf, FUNCLINE) { free (buf->memory); memset (buf, 0, sizeof (*buf)); }
void buf_set_shared (BufferP buf) {}                    /* (a lock for overlays other threads hold: one thread here) */
void buf_set_promiscuous_do (VBlockP vb, BufferP buf, rom name, FUNCLINE) {}
const BufDescType buf_desc (ConstBufferP buf) { BufDescType d = {}; return d; }
bool mutex_lock_do (MutexP mutex, bool blocking, FUNCLINE) { return true; }
void mutex_unlock_do (MutexP mutex, FUNCLINE) {}
void warn (rom fmt, ...) {}
rom report_support (void) { return ""; }
StrText vb_name (VBlockP vb) { StrText s = { "shim" }; return s; }
StrText line_name (VBlockP vb) { StrText s = { "shim" }; return s; }
void show_time_one (VBlockP vb, rom res, uint64_t delta) {}
void error_assert_failed (rom func, uint32_t line, rom fmt, ...) { va_list a; va_start (a, fmt); fprintf (stderr, "reference ASSERT in %s:%u: ", func, line); vfprintf (stderr, fmt, a); fprintf (stderr, "\n"); va_end (a); abort (); }
bool is_fastq_pair_2 (VBlockP vb) { return shim_pair2; }
bool fastq_zip_use_pair_identical (DictId dict_id) { return true; }
Codec codec_assign_best_codec (VBlockP vb, ContextP ctx, BufferP data, SectionType st) { return CODEC_UNKNOWN; }

#define SHIM_DID 3

/* a new file-level context: zeroed, its global hash of hash_alloc_global (estimated_entries) entries (src/hash.c:227-239; in the reference
 * the first VBlock to merge sizes it from its own statistics, hash_get_estimated_entries - the size never reaches the file) */
void mergeref_open (uint32_t estimated_entries)
{
    if (!z_file) { z_file = calloc (1, sizeof (File)); evb = calloc (1, sizeof (VBlock)); }
    z_file->data_type = DT_FASTQ;
    ContextP zctx = ZCTX (SHIM_DID);
    free (zctx->dict.memory); free (zctx->nodes.memory); free (zctx->counts.memory); free (zctx->global_hash.memory); free (zctx->ston_hash.memory); free (zctx->ston_ents.memory);
    memset (zctx, 0, sizeof (Context));
    zctx->did_i = zctx->dict_did_i = SHIM_DID; zctx->st_did_i = DID_NONE;
    strcpy (zctx->tag_name, "CTX"); memcpy (&zctx->dict_id, "CTX\0\0\0\0", 8);
    z_file->ca.num_contexts = SHIM_DID + 1;
    hash_alloc_global (zctx, estimated_entries);
}

/* ctx_merge_in_one_vctx for one VBlock's context.
 * in[]: 0 vblock_i  1 n_ol (vctx->ol_nodes.len: the words the file context had when the VBlock cloned it)  2 n_new  3 can have singletons
 *       (ltype == LT_SINGLETON, no_stons off - ctx_can_have_singletons reads the rest from flags)  4 flags byte (struct FlagsCtx)
 *       5 no_drop_b250  6 R2 of a pair with a pair-identical context  7 lcodec  8 bcodec  9 the one node index of an all-the-same b250 (or -1)
 * lens[]: b250.len, local.len, b250R1.len, localR1.len.   dict / char_index / snip_len [n_new] / counts [n_ol + n_new]: the VBlock's nodes.
 * out[]: 0 merged (1)  1 the b250 was dropped  2 lcodec  3 bcodec  4 bytes appended to local  5 singletons moved */
void mergeref_merge (const int32_t *in, const uint64_t *lens, const uint8_t *dict, uint64_t dict_len, const uint64_t *char_index, const uint32_t *snip_len,
                     const uint32_t *counts, int32_t *node2word, uint8_t *ston_local, int32_t *out)
{
    VBlockP vb = calloc (1, sizeof (VBlock));
    vb->data_type = DT_FASTQ; vb->vblock_i = in[0]; vb->ca.num_contexts = SHIM_DID + 1;
    ContextP vctx = &vb->ca.contexts[SHIM_DID], zctx = ZCTX (SHIM_DID);
    const uint32_t n_ol = in[1], n_new = in[2];
    vctx->did_i = SHIM_DID; vctx->st_did_i = DID_NONE; strcpy (vctx->tag_name, "CTX"); vctx->dict_id = zctx->dict_id;
    vctx->ol_nodes.len = n_ol;
    vctx->ltype = in[3] ? LT_SINGLETON : LT_BLOB;
    memcpy (&vctx->flags, &in[4], 1);
    vctx->no_drop_b250 = in[5]; shim_pair2 = in[6]; vctx->lcodec = in[7]; vctx->bcodec = in[8];
    if (n_new) {
        buf_alloc_do (vb, &vctx->dict, dict_len + 8, 1, "dict", __FUNCTION__, __LINE__);
        memcpy (vctx->dict.data, dict, dict_len); vctx->dict.len = dict_len;
        buf_alloc_do (vb, &vctx->nodes, (uint64_t)(n_new + 1) * sizeof (CtxNode), 1, "nodes", __FUNCTION__, __LINE__);
        for (uint32_t i = 0; i < n_new; i++) ((CtxNode *)vctx->nodes.data)[i] = (CtxNode){ .char_index = char_index[i], .snip_len = snip_len[i], .next = NO_NEXT };
        vctx->nodes.len = n_new;
    }
    buf_alloc_do (vb, &vctx->counts, (uint64_t)(n_ol + n_new + 1) * 4, 1, "counts", __FUNCTION__, __LINE__);
    memcpy (vctx->counts.data, counts, (size_t)(n_ol + n_new) * 4); vctx->counts.len = n_ol + n_new;
    if (vctx->flags.all_the_same) {                           /* the one entry of an all-the-same b250, written by the reference's own b250_seg_append */
        vctx->flags.all_the_same = false;
        b250_seg_append (vb, vctx, in[9]);
        if (!vctx->flags.all_the_same) abort ();
    }
    else if (lens[0]) { buf_alloc_do (vb, &vctx->b250, lens[0] + 8, 1, "b250", __FUNCTION__, __LINE__); vctx->b250.len = lens[0]; }
    const uint64_t local_before = lens[1];
    if (lens[1]) { buf_alloc_do (vb, &vctx->local, lens[1] + 8, 1, "local", __FUNCTION__, __LINE__); vctx->local.len = lens[1]; }
    vctx->b250R1.len = lens[2]; vctx->localR1.len = lens[3];
    if (vb->vblock_i == 1) zctx->vb_1_pending_merges = 1;     /* (the merges VBlock 1 still owes: counted down by the function, :1070-1074) */
    zctx->counts.len = zctx->nodes.len;                        /* (as the reference keeps it: one count per word) */
    const uint64_t stons_before = zctx->ston_ents.len;

    ContextP zout = NULL;
    out[0] = ctx_merge_in_one_vctx (vb, vctx, &zout);

    for (uint32_t i = 0; i < n_new; i++) node2word[i] = ((int32_t *)vctx->nodes.data)[i];
    out[1] = vctx->flags.all_the_same && !vctx->b250.len && !buf_is_alloc (&vctx->b250);
    out[2] = vctx->lcodec; out[3] = vctx->bcodec;
    out[4] = (int32_t)(vctx->local.len - local_before);
    out[5] = (int32_t)(zctx->ston_ents.len - stons_before);
    if (out[4]) memcpy (ston_local, vctx->local.data + local_before, out[4]);
    free (vctx->dict.memory); free (vctx->nodes.memory); free (vctx->counts.memory); free (vctx->b250.memory); free (vctx->local.memory); free (vctx->local_hash.memory);
    free (vb);
}

/* the file context: the dictionary, word count, counts, and what ctx_drop_all_the_same left behind.
 * out[]: 0 words  1 failed singletons  2 rm_dict_all_the_same  3 override_rm_dict_ats  4 all_the_same_wi_is_set  5 dict_flags.all_the_same_wi
 *        6 global hash length  7 flags byte  8 lcodec  9 bcodec */
uint64_t mergeref_view (uint8_t *dict_out, uint64_t dict_cap, uint64_t *counts_out, uint64_t counts_cap, int64_t *out)
{
    ContextP zctx = ZCTX (SHIM_DID);
    if (zctx->dict.len <= dict_cap && zctx->dict.len) memcpy (dict_out, zctx->dict.data, zctx->dict.len);
    for (uint64_t i = 0; i < zctx->nodes.len && i < counts_cap; i++) counts_out[i] = ((uint64_t *)zctx->counts.data)[i] & 0x7fffffffffffffffull;
    out[0] = zctx->nodes.len; out[1] = zctx->num_failed_singletons; out[2] = zctx->rm_dict_all_the_same; out[3] = zctx->override_rm_dict_ats;
    out[4] = zctx->all_the_same_wi_is_set; out[5] = zctx->dict_flags.all_the_same_wi; out[6] = zctx->global_hash.len32;
    uint8_t fl; memcpy (&fl, &zctx->flags, 1); out[7] = fl; out[8] = zctx->lcodec; out[9] = zctx->bcodec;
    return zctx->dict.len;
}

/* codec_assign_best_codec's commit (src/codec.c:352-363) as the next merges see it (:980-981) */
void mergeref_commit_codec (int is_local, int codec) { if (is_local) ZCTX (SHIM_DID)->lcodec = codec; else ZCTX (SHIM_DID)->bcodec = codec; }
