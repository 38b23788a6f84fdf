/* oracle/ref_order_shim.c -- TEST INFRASTRUCTURE ONLY (see oracle/Makefile, target "ref").
 *
 * Pins SURVEY 8a row a15. The reference's own src/zip.c is compiled as part of THIS translation unit (the #include below names the file
 * where it lies under /root/reference - nothing is copied), so that its static zip_compress_all_contexts_local / _b250 (:247-342) and
 * zip_generate_local (:167-244) run as the reference wrote them, in the sequence zip_compress_one_vb calls them (:565-585: the locals
 * before the merge unless it is VBlock 1, then - after the merge has put singletons into the locals of the contexts that had none -
 * the locals again, then the b250s), into oracle/_ref/liborderref.so. tests/golden/order_golden.json is generated from it
 * (tests/golden/make_order_golden.py).
 *
 * What is pinned is the ORDER in which a VBlock's context sections reach z_data with one compute thread (global_max_threads == 1: the
 * reference "forces predictability with single thread", :259,306 - with more it picks contexts at random). So the two functions that
 * append a section, zfile_compress_local_data and zfile_compress_b250_data (src/zfile.c:288-364), are stand-ins here that note which
 * context they were called for; b250_zip_generate says "kept"; the merge is played by setting the fields it sets (local.len of the
 * singleton-only contexts, dict_merged). Everything else the object imports and never reaches is named by the generated stubs file
 * (oracle/gen_ref_stubs.py). Nothing here is product code.
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <string.h>

#include "zip.c"                    /* the reference's own file, in place (-iquote $(REF)/src) */

Flags flag;
SegConf segconf;
VBlockP evb;
FileP z_file, txt_file;
FILE *info_stream;
uint32_t global_max_threads = 1;
const LocalTypeDesc lt_desc[NUM_LOCAL_TYPES] = LOCALTYPE_DESC;
DataTypeProperties dt_props[NUM_DATATYPES], dt_props_def;
__attribute__((constructor)) static void shim_defaults (void) { flag.show_time_comp_i = COMP_NONE; flag.command = ZIP; flag.is_lten = true; info_stream = stderr; }

static uint32_t *shim_out, shim_n;
uint32_t zfile_compress_local_data (VBlockP vb, ContextP ctx, uint32_t sample_size) { shim_out[shim_n++] = 2 * (uint32_t)(ctx - vb->ca.contexts); return 0; }
uint32_t zfile_compress_b250_data (VBlockP vb, ContextP ctx) { shim_out[shim_n++] = 2 * (uint32_t)(ctx - vb->ca.contexts) + 1; return 0; }
bool b250_zip_generate (VBlockP vb, ContextP ctx) { return true; }
LocalType dyn_int_get_ltype (ContextP ctx) { return ctx->ltype; }
Codec codec_assign_best_codec (VBlockP vb, ContextP ctx, BufferP data, SectionType st) { return CODEC_RANB; }
bool is_fastq_pair_2 (VBlockP vb) { return false; }
void threads_log_by_vb (ConstVBlockP vb, rom task_name, rom event, int time_usec) {}
void show_time_one (VBlockP vb, rom res, uint64_t delta) {}
void error_assert_failed (rom func, uint32_t line, rom fmt, ...) { va_list a; va_start (a, fmt); fprintf (stderr, "reference ASSERT in %s:%u: ", func, line); vfprintf (stderr, fmt, a); fprintf (stderr, "\n"); va_end (a); abort (); }

/* n contexts of a VBlock: did_i (their index among the VBlock's contexts), local_dep, has_local (data in local when the VBlock has been
 * segmented), ston_only (no local after segmentation, but the merge moves singletons into it), has_b250. Returns the number of sections;
 * out[k] = 2 * did_i (the local of the context) or 2 * did_i + 1 (its b250) in the order they were handed to the section writer */
uint32_t orderref_run (uint32_t n, const uint16_t *did_i, const uint8_t *local_dep, const uint8_t *has_local, const uint8_t *ston_only, const uint8_t *has_b250,
                       uint32_t vblock_i, uint32_t *out)
{
    VBlockP vb = calloc (1, sizeof (VBlock));
    vb->data_type = DT_FASTQ; vb->vblock_i = vblock_i;
    uint32_t max_did = 0;
    for (uint32_t i = 0; i < n; i++) if (did_i[i] > max_did) max_did = did_i[i];
    vb->ca.num_contexts = max_did + 1;
    for (uint32_t i = 0; i < n; i++) {
        ContextP c = &vb->ca.contexts[did_i[i]];
        c->did_i = did_i[i]; c->dict_id.num = 0x4142434400ull + did_i[i]; c->ltype = LT_BLOB; c->local_dep = local_dep[i];
        if (has_local[i] && !ston_only[i]) c->local.len = 7;
        if (has_b250[i]) c->b250.len = 5;
    }
    if (!txt_file) { txt_file = calloc (1, sizeof (File)); txt_file->data_type = DT_FASTQ; }     /* (DTPT (zip_comp_cb): no data-type callback) */
    shim_out = out; shim_n = 0;
    /* zip_compress_one_vb, src/zip.c:565-585 */
    if (vb->vblock_i != 1) zip_compress_all_contexts_local (vb);
    for (uint32_t i = 0; i < n; i++) {                      /* ctx_merge_in_vb_ctx: singletons into the locals that had nothing (context.c:297), dict_merged (:1102) */
        ContextP c = &vb->ca.contexts[did_i[i]];
        if (has_local[i] && ston_only[i]) c->local.len = 3;
        c->dict_merged = true;
    }
    zip_compress_all_contexts_local (vb);
    zip_compress_all_contexts_b250 (vb);
    free (vb);
    return shim_n;
}
