// gz_zip.h -- the VBlock compute driver (host; included at the end of gz_host.cpp): zip_compress_one_vb (src/zip.c:510-601)
// for a batch of FASTQ VBlocks whose text is resident in HBM. See include/genozip_amd.h ("the VBlock compute driver") for
// the flow; the reference functions behind every step are cited where the step is.
//
// Batch semantics (the reference's own output depends on thread timing): every VBlock of one call clones the
// file-level dictionaries as they were when the call started (ctx_clone, src/zip.c:528 - what the reference does when its
// compute threads all start before any has merged), then the VBlocks merge strictly in vblock_i order. A call with ONE
// VBlock reproduces the single-thread order of SURVEY A.6.
#pragma once
#include <algorithm>
#include <map>
#include <chrono>

// ---------------------------------------------------------------------------------------------------------
// a15: order of the sections of one VBlock (zip_compress_all_contexts_local src/zip.c:291-342 called at :566 for
// vb_i != 1 and at :583 for whatever is left; zip_compress_all_contexts_b250 :247-289; single compute thread: ascending
// did_i within a dependency level)
// ---------------------------------------------------------------------------------------------------------
extern "C" uint32_t gz_section_order (const GzSecOrderIn *c, uint32_t n, uint32_t vblock_i, uint32_t *order)
{
    if (!c || !order) return 0;
    std::vector<uint32_t> idx (n);
    for (uint32_t i = 0; i < n; i++) idx[i] = i;
    std::stable_sort (idx.begin (), idx.end (), [&] (uint32_t a, uint32_t b) { return c[a].did_i < c[b].did_i; });
    uint32_t k = 0;
    // first pass (before the merge, vb_i != 1 only): locals that exist before the merge; second pass: the rest - locals
    // that only came into existence during the merge (singletons), or, for vb_i = 1, all of them
    for (int pass = vblock_i == 1 ? 1 : 0; pass < 2; pass++)
        for (int dep = 0; dep < 3; dep++)
            for (uint32_t i : idx) {
                if (!c[i].has_local || c[i].local_dep != dep) continue;
                const bool pre_merge = !c[i].ston_only_local;
                if (vblock_i == 1 || (pass == 0) == pre_merge) order[k++] = 2 * i;
            }
    for (uint32_t i : idx) if (c[i].has_b250) order[k++] = 2 * i + 1;
    return k;
}

struct ZipCol {                    // one (VBlock, context) of this process
    uint32_t n = 0;                // snips (reads of the VBlock)
    int col_job = -1, dyn_job = -1, blob_job = -1, icol_job = -1;
    uint8_t *b250_seg = NULL, *b250_out = NULL;
    uint8_t *local = NULL; uint64_t local_cap = 0; uint32_t *sec_len_dev = NULL;
    // host side after the read-back / the merge
    uint32_t n_ol = 0, n_new = 0; bool all_the_same = false; uint64_t seg_b250_len = 0, b250_count = 0;
    uint64_t local_len = 0; int ltype = 0;
    bool has_b250 = false, has_local = false, ston_only_local = false;
    std::vector<uint8_t> host_b250;            // constant-snip contexts: the generated b250, written by the host
    const uint8_t *sec_b250 = NULL; uint32_t sec_b250_len = 0;
    std::vector<uint8_t> ston_local;
    size_t n2w_at = 0;
    uint8_t lcodec = 0, bcodec = 0;
    int early = -1;                            // index of this local's stream in the batch coded ahead on the second handle
    int early_b = -1;                          // the same for its b250 (predicted coding)
    bool host_len = false;                     // a dyn-int local whose final byte length the host knows (transposed here, not by the batch of local jobs)
    bool pre_node = false;                     // an R2 VBlock whose context's r2_node is new to the file: the VBlock's first new node (fastq.c:664-665)
    bool host_local = false;                   // the local was put together on the host (ston_local holds it): uploaded with the small payloads
};

// One (VBlock, context) as the merge sees it - what a process has to tell the others when the VBlocks of a file are dealt out
// to several processes (the ordered dictionary merge is the one exchange step of the path): fixed part + the
// VBlock's new words. Serialised into the "merge blob" of gz_fastq_zip_seg.
struct ZipMergeRec {
    uint32_t state;                // 0: nothing to merge; 1: a device column; 2: constant snip; 3: one snip of the VBlock's own (DOMQRUNS)
    uint32_t n, n_ol, n_new;
    uint64_t dict_len, seg_b250_len, b250_count, local_len;
    uint64_t domq_local_len;       // QUAL and its three: the length of the local if the file's QUAL goes through CODEC_DOMQ (local_len: if not)
    int32_t  ats_node; uint32_t all_the_same;
    // followed (state 1) by dict [dict_len], node_char_index [n_new], node_snip_len [n_new], counts [n_ol + n_new], each padded to 8 bytes;
    // (state 3) by the snip [dict_len], padded to 8 bytes
};
// qual: bit 0 the VBlock's QUAL was tested, bit 1 it is a fit for DOMQ (codec_domq.c:69-134), bit 2 a score outside ' '..'~'
struct ZipBlobVB { uint32_t vblock_i, r1_vblock_i, n_ctx, qual; };   // then n_ctx x (ZipMergeRec + payload)
// is_local: 0 b250, 1 local; | 2: the VBlock does not set the file's codec (too small, codec.c:352; VBlock 1 of a context whose beginning
// may not be representative, :358-362) - its choice holds for itself only; | 4: VBlock 10's second look (RETEST_VB_I, :274-277), which
// replaces the file's codec from that VBlock on
struct ZipVote { uint32_t ctx, is_local, vblock_i, codec; };
static inline bool zip_vb_commits (const GzFastqPlan &plan, const GzFastqVB &vb)
{
    return !plan.vb_size || vb.text_len > std::min<uint64_t> ((uint64_t)4 << 20, plan.vb_size / 2);
}

// src/codec.c:199-209 for a context of this plan
static inline bool zip_not_representative (const GzFastqPlan &plan, const GzFastqCtx &X)
{
    const uint32_t t = X.dict_id[0] >> 6;                                 // 0 field, 1 DTYPE_2, 2 / 3 DTYPE_1 (src/dict_id.h:15-17)
    return (plan.vb_1_not_representative >> (t == 0 ? 0 : t == 1 ? 2 : 1)) & 1;
}
#define ZIP_RETEST_VB_I 10u            // src/codec.c:22

// What ONE call of codec_assign_best_codec does with a context section whose codec the segmenter left open (normal mode: neither --best
// nor --fast; src/codec.c:259-283, 309-312, 352-363) - the decisions in front of the trials and behind them, as one rule:
//   bit 0  the trials run (else: the section takes the file's codec z_codec - or none: < 50 bytes and the file has none)
//   bit 1  their result is committed to the file's context - not from a VBlock of at most MIN (4 MB, vb_size / 2) of text (:352), and a
//          LOCAL codec not from VBlock 1 of a context whose beginning may not be representative unless it is the file's last (:358-362)
//   bit 2  it is VBlock 10's second look (RETEST_VB_I, :274-277): the trials run although the file has a codec
// pinned to the reference's own function: tests/golden/assign_golden.json (oracle/ref_assign_shim.c)
extern "C" int gz_codec_assign_rule (uint32_t vblock_i, uint64_t text_len, uint64_t vb_size, int last_of_file, int is_local,
                                     int not_representative, int hard_coded, int z_codec, uint32_t data_len)
{
    const bool retest = vblock_i == ZIP_RETEST_VB_I && !(is_local && hard_coded) && not_representative;
    if (!retest && z_codec) return 0;                                    // inherited (:280-281)
    if (data_len < 50) return 0;                                         // MIN_LEN_FOR_COMPRESSION (:311-312)
    const bool big = !vb_size || text_len > std::min<uint64_t> ((uint64_t)4 << 20, vb_size / 2);
    const bool commits = big && (!is_local || vblock_i > (not_representative ? 1u : 0u) || last_of_file);
    return 1 | (commits ? 2 : 0) | (retest && z_codec ? 4 : 0);
}

struct ZipVBState { std::vector<uint8_t> has_b250, has_local; std::vector<std::vector<uint8_t>> host_b250; };
struct ZipDomq { uint8_t *out[4] = { NULL, NULL, NULL, NULL }; GzDomqResult res; uint32_t fit = 0; std::vector<uint8_t> snip; };   // one VBlock's QUAL through k_domq

struct ZipCall {                   // what lives between the phases of one call
    int phase = 0;
    uint8_t *text = NULL; uint64_t text_len = 0; GzFastqVB *vbs = NULL; uint32_t NV = 0;
    std::vector<uint32_t> r0;
    std::vector<ZipCol> col;
    std::vector<GzColumnJob> col_jobs;
    std::vector<GzColumnResult> colres; std::vector<GzDynIntResult> dynres; std::vector<uint64_t> blobres, acgtres; std::vector<uint32_t> vbstat;
    std::vector<int> acgt_of_vb;
    GzDynIntResult *d_dynres = NULL; uint32_t *d_seclen = NULL; int32_t *d_b250st = NULL;
    std::vector<uint8_t> blob;      // this process' merge blob
    std::vector<ZipVote> votes;
    std::vector<GzVBlock> V; std::vector<std::vector<GzSection>> secs; std::vector<int32_t> b250st; int32_t *b250st_pinned = NULL;   // phase 3, between launch and wait
    std::vector<ZipDomq> domq;                 // per VBlock, when the file's QUAL may go / goes through CODEC_DOMQ
    int qual_mode_applied = -1;
    std::vector<std::vector<uint8_t>> own_snip; // per (VBlock, context): the snip a GZ_FQ_TOPLEVEL context segs in this VBlock
    std::vector<GzStream> early;               // the streams coded ahead (results arrive when the second handle is synchronised)
    bool predicted = false;                    // ... with predicted codecs, beside their contexts' trials (the merge phase)
    uint32_t *d_early_len = NULL;
    // speculation: the long streams were handed to the coders with the codec the handle's previous file ended up with, before this
    // file's own trial (a8) was through; the trial (queued on the main handle once the seg phase has its results) confirms or refutes
    std::vector<GzStream> spec_trial; int spec_codec = 0; size_t spec_t = 0; bool spec = false, spec_pending = false;
    std::vector<GzStream> trial;               // the trial compressions queued on the second handle in the seg phase: the coder keeps pointers into
                                               // it (GzStream.status / out_len) until that handle is synchronised, so it lives as long as the call
    std::vector<int32_t> n2w_host;
    std::map<uint32_t, ZipVBState> vbstate;   // by vblock_i: every VBlock of the call, own or not
};

// ---------------------------------------------------------------------------------------------------------
struct GzZipFile {
    GzHandle *h;
    GzHandle *h2 = NULL;                   // the long streams (QUAL) are coded here, ahead of and beside everything else
    GzFastqPlan plan;
    std::vector<GzFastqCtx> ctxs;
    std::vector<std::vector<uint8_t>> snips, r2_nodes;
    std::vector<GzZctx *> zctx;
    std::vector<ArenaBlock> ws;            // device workspace of one call (bump allocated, reused by the next call)
    std::vector<uint8_t> stage;            // host staging
    // the vblock_i merged so far, as disjoint ascending ranges: calls number their VBlocks freely as long as none comes twice (a
    // streamed pair of files is R1 = 1..N, R2 = N+1..2N, writer.c:318-322, while each call holds some VBlocks of both)
    std::vector<std::pair<uint32_t, uint32_t>> merged;
    uint32_t next_vblock_i () const { return (merged.empty () || merged[0].first > 1) ? 1 : merged[0].second + 1; }   // the lowest not merged yet
    bool was_merged (uint32_t i) const { for (auto &r : merged) if (i >= r.first && i <= r.second) return true; return false; }
    void add_merged (uint32_t i) {
        size_t k = 0; while (k < merged.size () && merged[k].second + 1 < i) k++;
        if (k < merged.size () && merged[k].first <= i + 1) { if (i < merged[k].first) merged[k].first = i; if (i > merged[k].second) merged[k].second = i; }
        else merged.insert (merged.begin () + k, std::make_pair (i, i));
        if (k + 1 < merged.size () && merged[k].second + 1 >= merged[k + 1].first) { merged[k].second = merged[k + 1].second; merged.erase (merged.begin () + k + 1); }
    }
    // zctx->qual_codec of the QUAL context (codec.c:403-407,445): -1 not decided yet (the file's first VBlock will), 0 a plain
    // LT_BLOB local, GZ_CODEC_DOMQ
    int qual_ctx = -1, aux[3] = { -1, -1, -1 }, qual_mode = 0;
    int seq_snip_ctx = -1;                 // the plan's GZ_FQ_SEQ_SNIP context (SQBITMAP)
    hipEvent_t ev_early = NULL;
    uint8_t *pinned = NULL; size_t pinned_size = 0;      // host memory the device can write while the host does something else
    ZipCall call;
    // ---- two calls in flight (gz_fastq_zip_begin / _end): everything above that belongs to ONE call - the handles its work is
    // queued on, its workspace, its staging memory, its state - exists twice; `other` holds the set that is not the current one
    // (the older call while two are in flight). The file-level state (dictionaries, codecs, vblock_i) is one.
    struct Lane { GzHandle *h = NULL, *h2 = NULL; std::vector<ArenaBlock> ws; std::vector<uint8_t> stage; hipEvent_t ev_early = NULL;
                  uint8_t *pinned = NULL; size_t pinned_size = 0; ZipCall call; bool busy = false; } other;
    bool busy = false, other_made = false;
    GzHandle *h_user = NULL;                             // the handle the file was opened on (owns the profile of all of them)
    // a8 in full: the host's candidates (BZ2 / BSC / LZMA) and the reference's sorter (gz_zip_set_host_codecs)
    GzHostCodecs hostc = { NULL, NULL, NULL, NULL, 0 };
    std::vector<float> hostc_clock;
};

extern "C" int gz_zip_set_host_codecs (GzZipFile *f, const GzHostCodecs *hc)
{
    if (!f || (hc && (hc->mode < 0 || hc->mode > 2))) return GZ_ERR_ARG;
    if (!hc) { f->hostc = GzHostCodecs { NULL, NULL, NULL, NULL, 0 }; f->hostc_clock.clear (); return GZ_OK; }
    f->hostc = *hc;
    if (hc->clock_ns_per_byte) { f->hostc_clock.assign (hc->clock_ns_per_byte, hc->clock_ns_per_byte + 32); f->hostc.clock_ns_per_byte = f->hostc_clock.data (); }
    return GZ_OK;
}

static void zip_swap_lanes (GzZipFile *f)
{
    std::swap (f->h, f->other.h); std::swap (f->h2, f->other.h2); f->ws.swap (f->other.ws); f->stage.swap (f->other.stage);
    std::swap (f->ev_early, f->other.ev_early); std::swap (f->pinned, f->other.pinned); std::swap (f->pinned_size, f->other.pinned_size);
    std::swap (f->call, f->other.call); std::swap (f->busy, f->other.busy);
}

static uint8_t *zip_pinned (GzZipFile *f, size_t bytes)
{
    if (f->pinned_size >= bytes) return f->pinned;
    if (f->pinned) (void)hipHostFree (f->pinned);
    f->pinned = NULL; f->pinned_size = 0;
    if (hipHostMalloc ((void **)&f->pinned, bytes + bytes / 2, hipHostMallocDefault) != hipSuccess) { f->h->err = "hipHostMalloc failed"; return NULL; }
    f->pinned_size = bytes + bytes / 2;
    return f->pinned;
}

static void zip_init_qual_mode (GzZipFile *f)
{
    const bool possible = f->qual_ctx >= 0 && f->aux[0] >= 0 && f->aux[1] >= 0 && f->aux[2] >= 0;
    f->qual_mode = !possible || f->plan.qual_codec == GZ_CODEC_NONE ? 0 : f->plan.qual_codec == GZ_CODEC_DOMQ ? GZ_CODEC_DOMQ : -1;
}

// base64_encode (src/base64.c: the standard alphabet, '=' padded) of the denormalisation table -> the snip segged into DOMQRUNS
static void zip_base64 (const uint8_t *in, size_t n, std::vector<uint8_t> &out)
{
    static const char A[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    out.clear ();
    for (size_t i = 0; i < n; i += 3) {
        const uint32_t b0 = in[i], b1 = i + 1 < n ? in[i + 1] : 0, b2 = i + 2 < n ? in[i + 2] : 0, w = b0 << 16 | b1 << 8 | b2;
        out.push_back (A[w >> 18]); out.push_back (A[(w >> 12) & 63]);
        out.push_back (i + 1 < n ? A[(w >> 6) & 63] : '='); out.push_back (i + 2 < n ? A[w & 63] : '=');
    }
}

static void *ws_alloc (GzZipFile *f, size_t bytes)
{
    bytes = (bytes + 255) & ~(size_t)255;
    if (!bytes) bytes = 256;
    for (auto &b : f->ws) if (b.size - b.used >= bytes) { void *p = b.base + b.used; b.used += bytes; return p; }
    size_t sz = (size_t)512 << 20;
    while (sz < bytes) sz *= 2;
    ArenaBlock nb; nb.size = sz; nb.used = bytes;
    if (hipMalloc ((void **)&nb.base, sz) != hipSuccess) {
        nb.size = bytes;
        if (hipMalloc ((void **)&nb.base, bytes) != hipSuccess) { f->h->err = "hipMalloc failed (zip workspace)"; return NULL; }
    }
    f->ws.push_back (nb);
    return nb.base;
}

static GzZipFile *zip_open_failed (GzZipFile *f) { for (auto z : f->zctx) gz_zctx_destroy (z); delete f; return NULL; }

extern "C" GzZipFile *gz_zip_open (GzHandle *h, const GzFastqPlan *plan)
{
    if (!h || !plan || !plan->ctxs || !plan->n_ctxs || plan->n_seps > GZ_TOK_MAX_SEPS) return NULL;
    if (plan->seq_pad & (plan->seq_pad - 1) || plan->seq_pad > 64) return NULL;
    GzZipFile *f = new GzZipFile ();
    f->h = h; f->h_user = h; f->plan = *plan;
    f->ctxs.assign (plan->ctxs, plan->ctxs + plan->n_ctxs);
    f->snips.resize (plan->n_ctxs); f->r2_nodes.resize (plan->n_ctxs);
    for (uint32_t i = 0; i < plan->n_ctxs; i++) {
        GzFastqCtx &c = f->ctxs[i];
        if (c.snip && c.snip_len) f->snips[i].assign (c.snip, c.snip + c.snip_len);
        c.snip = f->snips[i].data ();
        if (c.r2_node && c.r2_node_len) {
            if (c.r2_node_len > 16 || (c.kind != GZ_FQ_SEQ_SNIP && c.kind != GZ_FQ_ITEM_TEXT) || memchr (c.r2_node, 0, c.r2_node_len)) return zip_open_failed (f);
            f->r2_nodes[i].assign (c.r2_node, c.r2_node + c.r2_node_len);
        }
        else c.r2_node_len = 0;
        c.r2_node = f->r2_nodes[i].data ();
        if ((c.kind == GZ_FQ_ITEM_TEXT || c.kind == GZ_FQ_ITEM_INT || c.kind == GZ_FQ_ITEM_DELTA) && c.item > plan->n_seps) return zip_open_failed (f);
        if ((c.kind == GZ_FQ_CONST || c.kind == GZ_FQ_ITEM_DELTA) && !c.snip_len) return zip_open_failed (f);
        if (c.kind == GZ_FQ_ITEM_TEXT && c.snip_len > 4) return zip_open_failed (f);
        if (c.kind == GZ_FQ_ITEM_EXPECT && (c.snip_len > 16 || c.item > plan->n_seps)) return zip_open_failed (f);                 // (a lead-in of every snip: at most 4 bytes)
        if (c.kind == GZ_FQ_TOPLEVEL && (c.con_len < 8 || c.con_len > c.snip_len || (c.con_len - 8) % 12)) return zip_open_failed (f);   // Container_0 + n ContainerItem
        if (c.kind == GZ_FQ_SEQ_SNIP) { if (!c.snip_len || c.snip_len > 4 || f->seq_snip_ctx >= 0) return zip_open_failed (f); f->seq_snip_ctx = (int)i; }
        if (c.kind == GZ_FQ_QUAL) { if (f->qual_ctx >= 0 || c.snip_len > 3) return zip_open_failed (f); f->qual_ctx = (int)i; }     // (one QUAL per plan; its snip: the lead-in of a monochar line's)
        if (c.kind == GZ_FQ_QUAL_AUX) { if (c.item > 2 || f->aux[c.item] >= 0) return zip_open_failed (f); f->aux[c.item] = (int)i; }
        f->zctx.push_back (gz_zctx_create (plan->estimated_entries));
        if (c.lcodec) gz_zctx_commit_codec (f->zctx.back (), 1, c.lcodec);
        if (c.bcodec) gz_zctx_commit_codec (f->zctx.back (), 0, c.bcodec);
    }
    f->plan.ctxs = f->ctxs.data ();
    zip_init_qual_mode (f);
    { const char *e = getenv ("GZ_ZIP_NO_OVERLAP"); int err = 0; if (!(e && *e && *e != '0')) f->h2 = gz_create_background (h->device, &err); }
    if (f->h2) { f->h2->profiling = h->profiling; h->helpers.push_back (f->h2); }
    return f;
}

extern "C" void gz_zip_close (GzZipFile *f)
{
    if (!f) return;
    (void)hipSetDevice (f->h_user->device);
    for (int lane = 0; lane < 2; lane++) {
        if (f->h) {
            (void)gz_sync (f->h);
            if (f->h2) (void)gz_sync (f->h2);
            auto &hl = f->h_user->helpers;
            for (GzHandle *o : { f->h2, f->h == f->h_user ? (GzHandle *)NULL : f->h })
                if (o) { hl.erase (std::remove (hl.begin (), hl.end (), o), hl.end ()); gz_destroy (o); }
            if (f->pinned) (void)hipHostFree (f->pinned);
            if (f->ev_early) (void)hipEventDestroy (f->ev_early);
            for (auto &b : f->ws) (void)hipFree (b.base);
            f->h = f->h2 = NULL; f->pinned = NULL; f->ev_early = NULL; f->ws.clear ();
        }
        zip_swap_lanes (f);
    }
    for (auto z : f->zctx) gz_zctx_destroy (z);
    delete f;
}

// a new file with the same plan: fresh dictionaries and codecs, the device workspace is kept
extern "C" int gz_zip_reset (GzZipFile *f)
{
    if (!f) return GZ_ERR_ARG;
    int rc = GZ_OK;
    for (int lane = 0; lane < 2; lane++) {                 // (a call that failed half way, or was never ended, may have left work anywhere)
        if (f->h) { const int r1 = gz_sync (f->h); if (rc >= 0 && r1 < 0) rc = r1; }
        if (f->h2) { const int r2 = gz_sync (f->h2); if (rc >= 0 && r2 < 0) rc = r2; }
        f->call = ZipCall (); f->busy = false;
        zip_swap_lanes (f);
    }
    if (f->h != f->h_user) zip_swap_lanes (f);             // (start again on the caller's handle)
    if (rc < 0) return rc;
    for (size_t i = 0; i < f->zctx.size (); i++) {
        gz_zctx_destroy (f->zctx[i]);
        f->zctx[i] = gz_zctx_create (f->plan.estimated_entries);
        if (f->ctxs[i].lcodec) gz_zctx_commit_codec (f->zctx[i], 1, f->ctxs[i].lcodec);
        if (f->ctxs[i].bcodec) gz_zctx_commit_codec (f->zctx[i], 0, f->ctxs[i].bcodec);
    }
    f->merged.clear ();
    f->call = ZipCall ();
    zip_init_qual_mode (f);
    return GZ_OK;
}

// the z_data of the VBlocks of the last call, one after the other, into a buffer of the caller (device): what goes to the
// writer (zfile_output_processed_vb_ext, src/zfile.c:1160) - or into the gather to the writer rank. offsets_host: n + 1 entries
extern "C" int gz_fastq_zip_collect (GzZipFile *f, const GzFastqVB *vbs, int n_vbs, uint8_t *dst, uint64_t cap, uint64_t *offsets_host)
{
    if (!f || n_vbs < 0 || (n_vbs && (!vbs || !dst)) || !offsets_host) return GZ_ERR_ARG;
    GzHandle *h = f->h;
    HIPCHK (h, hipSetDevice (h->device));
    uint64_t at = 0, longest = 0;
    std::vector<GzdPiece> pieces (n_vbs);
    for (int v = 0; v < n_vbs; v++) {
        offsets_host[v] = at;
        if (at + vbs[v].z_len > cap) return GZ_TOO_SMALL;
        pieces[v].src = vbs[v].z_data; pieces[v].dst = dst + at; pieces[v].len = vbs[v].z_len;
        longest = std::max<uint64_t> (longest, vbs[v].z_len);
        at += vbs[v].z_len;
    }
    offsets_host[n_vbs] = at;
    if (n_vbs && longest) {                                    // one launch for all of them
        void *d;
        int rc = upload (h, pieces.data (), pieces.size () * sizeof (GzdPiece), &d);
        if (rc != GZ_OK) return rc;
        const uint32_t slices = (uint32_t)std::min<uint64_t> (64, std::max<uint64_t> (1, longest / 65536));
        KLAUNCH (h, k_pieces_copy, dim3 ((uint32_t)n_vbs, slices), dim3 (256), 0, (const GzdPiece *)d);
    }
    HIPCHK (h, hipStreamSynchronize (h->stream));
    return GZ_OK;
}

extern "C" GzZctx *gz_zip_zctx (GzZipFile *f, uint32_t i) { return f && i < f->zctx.size () ? f->zctx[i] : NULL; }

// ---- the batched forms --------------------------------------------------------------------------------------------------
extern "C" int gz_tokenize_column_n (GzHandle *h, const uint8_t *text, const uint32_t *off, const uint32_t *len, uint32_t n,
                                     const char *seps, const uint8_t *counts, uint32_t n_seps, uint32_t *item_off, uint32_t *item_len, uint32_t *n_bad_dev)
{
    if (!h || !n_bad_dev || n_seps > GZ_TOK_MAX_SEPS || (n_seps && !seps) || (n && (!text || !off || !len || !item_off || !item_len))) return GZ_ERR_ARG;
    HIPCHK (h, hipSetDevice (h->device));
    GzdTokensN T;
    memset (&T, 0, sizeof (T));
    T.text = text; T.off = off; T.len = len; T.n = n; T.n_seps = n_seps; T.item_off = item_off; T.item_len = item_len; T.n_bad = n_bad_dev;
    for (uint32_t i = 0; i < n_seps; i++) { T.seps[i] = (uint8_t)seps[i]; T.counts[i] = counts && counts[i] ? counts[i] : 1; }
    HIPCHK (h, hipMemsetAsync (n_bad_dev, 0, 4, h->stream));
    if (n) KLAUNCH (h, k_tokenize_n, dim3 ((n + 255) / 256), dim3 (256), 64, T);
    HIPCHK (h, hipGetLastError ());
    return GZ_OK;
}

extern "C" int gz_int_columns (GzHandle *h, const GzIntColJob *jobs, int n_jobs)
{
    if (!h || (n_jobs && !jobs) || n_jobs < 0 || n_jobs > 65535) return GZ_ERR_ARG;
    if (!n_jobs) return GZ_OK;
    HIPCHK (h, hipSetDevice (h->device));
    std::vector<GzdIntCol> J (n_jobs);
    uint32_t max_tiles = 1;
    for (int i = 0; i < n_jobs; i++) {
        const GzIntColJob &u = jobs[i];
        if (!u.n_values_dev || !u.status_dev || (u.n && (!u.text || !u.off || !u.len || !u.values))) return GZ_ERR_ARG;
        if (u.mode == 0 && u.n && (!u.snip_off || !u.snip_len || !u.is_nothing)) return GZ_ERR_ARG;
        GzdIntCol &d = J[i];
        d.text = u.text; d.off = u.off; d.len = u.len; d.n = u.n; d.nothing_char = u.nothing_char; d.lookup_off = u.lookup_off; d.mode = u.mode;
        d.snip_off = u.snip_off; d.snip_len = u.snip_len; d.values = u.values; d.is_nothing = u.is_nothing; d.n_values = u.n_values_dev; d.status = u.status_dev;
        const uint32_t tiles = (u.n + 255) / 256;
        if (!(d.tile = (uint64_t *)arena_alloc (h, ((size_t)tiles + 1) * 8))) return GZ_ERR_HIP;
        if (tiles > max_tiles) max_tiles = tiles;
    }
    void *dj;
    int rc;
    if ((rc = upload (h, J.data (), J.size () * sizeof (GzdIntCol), &dj)) != GZ_OK) return rc;
    KLAUNCH (h, k_icol_count, dim3 (max_tiles, n_jobs), dim3 (256), 2048, (GzdIntCol *)dj);
    KLAUNCH (h, k_icol_scan, dim3 (n_jobs), dim3 (256), 2048, (GzdIntCol *)dj);
    KLAUNCH (h, k_icol_write, dim3 (max_tiles, n_jobs), dim3 (256), 2048, (GzdIntCol *)dj);
    HIPCHK (h, hipGetLastError ());
    return GZ_OK;
}

extern "C" int gz_local_generate_batch (GzHandle *h, const GzLocalJob *jobs, int n_jobs)
{
    if (!h || (n_jobs && !jobs) || n_jobs < 0 || n_jobs > 65535) return GZ_ERR_ARG;
    if (!n_jobs) return GZ_OK;
    HIPCHK (h, hipSetDevice (h->device));
    std::vector<GzdLocalJob> J (n_jobs);
    uint64_t max_tiles = 1;
    for (int i = 0; i < n_jobs; i++) {
        J[i].data = (uint8_t *)jobs[i].data; J[i].n = jobs[i].n; J[i].dyn = jobs[i].dyn_dev; J[i].ltype = jobs[i].ltype; J[i].len_dev = jobs[i].len_dev;
        if (jobs[i].n && !jobs[i].data) return GZ_ERR_ARG;
        max_tiles = std::max<uint64_t> (max_tiles, (jobs[i].n + 1023) / 1024);
    }
    if (max_tiles > 0x7fffffffull) return GZ_ERR_ARG;
    void *dj;
    int rc;
    if ((rc = upload (h, J.data (), J.size () * sizeof (GzdLocalJob), &dj)) != GZ_OK) return rc;
    KLAUNCH (h, k_local_order_jobs, dim3 ((uint32_t)max_tiles, n_jobs), dim3 (256), 0, (const GzdLocalJob *)dj);
    HIPCHK (h, hipGetLastError ());
    return GZ_OK;
}

extern "C" int gz_acgt_pack_batch (GzHandle *h, const GzAcgtJob *jobs, int n_jobs)
{
    if (!h || (n_jobs && !jobs) || n_jobs < 0 || n_jobs > 65535) return GZ_ERR_ARG;
    if (!n_jobs) return GZ_OK;
    HIPCHK (h, hipSetDevice (h->device));
    std::vector<GzdAcgtJob> J (n_jobs);
    uint64_t max_tiles = 1;
    for (int i = 0; i < n_jobs; i++) {
        const GzAcgtJob &u = jobs[i];
        if (!u.has_x_dev || (u.n_max && (!u.seq || !u.packed || !u.x))) return GZ_ERR_ARG;
        J[i].seq = u.seq; J[i].n_dev = u.n_dev; J[i].n_max = u.n_max; J[i].packed = u.packed; J[i].x = u.x; J[i].has_x = u.has_x_dev; J[i].packed_len = u.packed_len_dev;
        HIPCHK (h, hipMemsetAsync (u.has_x_dev, 0, 4, h->stream));
        max_tiles = std::max<uint64_t> (max_tiles, (gz_acgt_packed_len (u.n_max) / 4 + 255) / 256);
    }
    void *dj;
    int rc;
    if ((rc = upload (h, J.data (), J.size () * sizeof (GzdAcgtJob), &dj)) != GZ_OK) return rc;
    KLAUNCH (h, k_acgt_pack_jobs, dim3 ((uint32_t)max_tiles, n_jobs), dim3 (256), 512, (const GzdAcgtJob *)dj);
    HIPCHK (h, hipGetLastError ());
    return GZ_OK;
}

// ---------------------------------------------------------------------------------------------------------
// the driver
// ---------------------------------------------------------------------------------------------------------
// PIZ-format b250 entry of one word index (src/b250.c:82-110), for the all-the-same b250s the host writes itself
static uint32_t zip_piz_put (uint8_t *d, int64_t wi)
{
    if (wi >= 0 && wi <= 126) { d[0] = (uint8_t)wi; return 1; }
    if (wi <= 16508) { const uint32_t v = (uint32_t)wi - 127; d[0] = 0x80 | (uint8_t)(v >> 8); d[1] = (uint8_t)v; return 2; }
    if (wi <= 2113660) { const uint32_t v = (uint32_t)wi - 16509; d[0] = 0xC0 | (uint8_t)(v >> 16); d[1] = (uint8_t)(v >> 8); d[2] = (uint8_t)v; return 3; }
    d[0] = 0xE0 | (uint8_t)(wi >> 24); d[1] = (uint8_t)(wi >> 16); d[2] = (uint8_t)(wi >> 8); d[3] = (uint8_t)wi;
    return 4;
}

static inline bool zip_is_host_codec (int c) { return c == GZ_CODEC_BZ2 || c == GZ_CODEC_LZMA || c == GZ_CODEC_BSC; }

// The winner among the nine device candidates (payload sizes of their trials on the sample at `in`) and, if the file has them, the
// host's: the reference's sorter (gz_assign_pick). Without host candidates and clocks this is "the smallest framed size, the first
// of equals"  - the device trials all count as fast enough (codec.c:140-143).
static int zip_pick_codec (GzHandle *h, GzZipFile *f, const uint8_t dict_id[8], int is_local, const uint8_t *in, uint32_t sample, const uint32_t payload[8], int *codec)
{
    GzCodecTest extra[8]; int n_extra = 0;
    if (f->hostc.trial) {
        std::vector<uint8_t> host (sample);
        HIPCHK (h, hipMemcpy (host.data (), in, sample, hipMemcpyDeviceToHost));
        n_extra = f->hostc.trial (f->hostc.user, dict_id, is_local, host.data (), sample, extra, 8);
        if (n_extra < 0 || n_extra > 8) { h->err = "host codec trial"; return GZ_ERR; }
        for (int i = 0; i < n_extra; i++) {
            if (!zip_is_host_codec (extra[i].codec)) { h->err = "host codec trial: not a host codec"; return GZ_ERR_ARG; }
            extra[i].size += 28;                                                         // framed (codec.c:328-331)
        }
    }
    *codec = gz_assign_pick (sample, payload, extra, n_extra, f->hostc.clock_ns_per_byte, f->hostc.mode, NULL);
    return GZ_OK;
}

// codec_assign_best_codec (codec.c:234-363) for several streams in ONE batch of trial compressions; who[i] = whose stream i is
static int zip_assign_best_many (GzHandle *h, GzZipFile *f, const std::vector<const uint8_t *> &ptr, const std::vector<uint32_t> &len, const std::vector<ZipVote> &who, std::vector<int> &best)
{
    static const int cand[8] = { GZ_CODEC_RANB, GZ_CODEC_RANW, GZ_CODEC_RANb, GZ_CODEC_RANw, GZ_CODEC_ARTB, GZ_CODEC_ARTW, GZ_CODEC_ARTb, GZ_CODEC_ARTw };
    const size_t n = ptr.size ();
    best.assign (n, GZ_CODEC_UNKNOWN);
    std::vector<GzStream> S;
    std::vector<size_t> owner;
    for (size_t i = 0; i < n; i++) {
        const uint32_t sample = len[i] < 99999 ? len[i] : 99999;                         // codec.c:309
        if (sample < 50) continue;                                                       // codec.c:311-312
        for (int c = 0; c < 8; c++) {
            GzStream s; memset (&s, 0, sizeof (s));
            s.in = ptr[i]; s.in_len = sample; s.codec = cand[c]; s.out_cap = gz_codec_est_size (cand[c], sample);
            if (!(s.out = (uint8_t *)ws_alloc (f, (size_t)s.out_cap + 16))) return GZ_ERR_HIP;
            S.push_back (s); owner.push_back (i);
        }
    }
    if (S.empty ()) return GZ_OK;
    int rc;
    if ((rc = gz_codec_compress_batch (h, S.data (), (int)S.size ())) != GZ_OK) return rc;
    if ((rc = gz_sync (h)) < 0) return rc;
    for (size_t k = 0; k < S.size (); k += 8) {
        const size_t i = owner[k];
        uint32_t payload[8];
        for (int c = 0; c < 8; c++) { if (S[k + c].status != GZ_OK) return GZ_ERR; payload[c] = S[k + c].out_len; }
        if ((rc = zip_pick_codec (h, f, f->ctxs[who[i].ctx].dict_id, (int)(who[i].is_local & 1), ptr[i], len[i] < 99999 ? len[i] : 99999, payload, &best[i])) != GZ_OK) return rc;
    }
    return GZ_OK;
}

// GZ_ZIP_TIMING=1: wall-clock milliseconds between the marks of a call, to stderr (where the host side of a step goes)
struct ZipTimer {
    bool on; std::chrono::steady_clock::time_point t0, last; std::string line;
    ZipTimer () { const char *e = getenv ("GZ_ZIP_TIMING"); on = e && *e && *e != '0'; t0 = last = std::chrono::steady_clock::now (); }
    void mark (const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now ();
        char b[96]; snprintf (b, sizeof (b), " %s %.2f", what, std::chrono::duration<double, std::milli> (now - last).count ());
        line += b; last = now;
    }
    void done (const char *phase) {
        if (!on) return;
        fprintf (stderr, "[gz_zip %s %.2f ms]%s\n", phase, std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now () - t0).count (), line.c_str ());
    }
};

#define ZCHK(call) do { int rc_ = (call); if (rc_ != GZ_OK) return rc_ < 0 ? rc_ : GZ_ERR; } while (0)
#define WS(var, type, count) type *var = (type *)ws_alloc (f, (size_t)(count) * sizeof (type)); if (!var) return GZ_ERR_HIP

static void zip_apply_qual_mode (GzZipFile *f, int mode);
static inline void blob_put (std::vector<uint8_t> &b, const void *p, size_t n)
{
    const size_t at = b.size ();
    b.resize (at + ((n + 7) & ~(size_t)7), 0);
    if (n) memcpy (b.data () + at, p, n);
}

// ---------------------------------------------------------------------------------------------------------
// phase 1: text -> the columns of every (VBlock, context) of THIS process; returns the merge blob
// ---------------------------------------------------------------------------------------------------------
extern "C" int gz_fastq_zip_seg (GzZipFile *f, uint8_t *text, uint64_t text_len, GzFastqVB *vbs, int n_vbs, const void **blob_out, uint64_t *blob_len_out)
{
    if (!f || !blob_out || !blob_len_out) return GZ_ERR_ARG;
    // (a previous call that was given up half way - in any phase, also inside its seg phase with f->call.phase still 0 - may have left
    //  trial or long streams running on the second handle: in the workspace that is about to be reused, with results going to f->call)
    if (f->h2) (void)gz_sync (f->h2);
    if (n_vbs == 0) {                                      // a process without VBlocks in this call still takes part in the merge
        int rc0 = gz_sync (f->h);
        if (rc0 < 0) return rc0;
        for (auto &b : f->ws) b.used = 0;
        f->call = ZipCall ();
        f->call.phase = 1;
        *blob_out = NULL; *blob_len_out = 0;
        return GZ_OK;
    }
    if (!text || !vbs || n_vbs < 0 || n_vbs > 16384 || text_len >= 0xfffffff0ull) return GZ_ERR_ARG;
    GzHandle *h = f->h;
    const uint32_t NC = (uint32_t)f->ctxs.size (), NV = (uint32_t)n_vbs;
    if ((uint64_t)NC * NV > 60000) { h->err = "too many (VBlock, context) pairs for one call"; return GZ_ERR_ARG; }
    for (uint32_t v = 0; v < NV; v++) {
        if (vbs[v].text_off + vbs[v].text_len > text_len || (v && vbs[v].vblock_i <= vbs[v - 1].vblock_i) || !vbs[v].vblock_i ||
            vbs[v].r1 >= (int32_t)v || (v && vbs[v].text_off < vbs[v - 1].text_off + vbs[v - 1].text_len)) { h->err = "VBlock table: offsets / order / r1"; return GZ_ERR_ARG; }
        vbs[v].status = GZ_ERR; vbs[v].z_data = NULL; vbs[v].z_len = 0; vbs[v].n_reads = 0; vbs[v].seq_packed = NULL; vbs[v].seq_packed_len = 0;
        vbs[v].n_bases = 0; vbs[v].seq_has_x = 0; vbs[v].n_sections = 0;
    }
    HIPCHK (h, hipSetDevice (h->device));
    int rc;
    ZipTimer T;
    if ((rc = gz_sync (h)) < 0) return rc;
    for (auto &b : f->ws) b.used = 0;
    f->call = ZipCall ();
    ZipCall &K = f->call;
    K.text = text; K.text_len = text_len; K.vbs = vbs; K.NV = NV;
    K.votes.clear ();

    // ---- lines of the whole text, first line of every VBlock (seg_get_next_line, src/seg.c:200-236) -------------------------
    const uint8_t lookup_byte[16] = { 1 };                                  // SNIP_LOOKUP parked behind the text
    HIPCHK (h, hipMemcpyAsync (text + text_len, lookup_byte, 16, hipMemcpyHostToDevice, h->stream));
    const uint32_t lookup_off = (uint32_t)text_len;
    uint32_t line_cap = (uint32_t)(text_len / 16 + 1024);
    uint32_t *line_off = NULL, *line_len = NULL;
    struct ABlock { GzLinesResult lines; uint32_t bad_bound, n_bad_items; GzFastqResult fq; uint32_t n_line3, n_bad_samples, n_missing, n_unexpected; GzLinesResult tabs; } ;
    WS (d_a, ABlock, 1);
    WS (d_vb_off, uint64_t, 2 * NV + 2);
    WS (d_first_line, uint32_t, 2 * NV + 2);
    std::vector<uint64_t> vb_off (2 * NV + 2);                              // starts, then ends
    for (uint32_t v = 0; v < NV; v++) { vb_off[v] = vbs[v].text_off; vb_off[NV + v] = vbs[v].text_off + vbs[v].text_len; }
    HIPCHK (h, hipMemcpyAsync (d_vb_off, vb_off.data (), vb_off.size () * 8, hipMemcpyHostToDevice, h->stream));
    ABlock a;
    std::vector<uint32_t> first_line (2 * NV + 2);
    for (int attempt = 0;; attempt++) {
        line_off = (uint32_t *)ws_alloc (f, ((size_t)line_cap + 8) * 4); line_len = (uint32_t *)ws_alloc (f, ((size_t)line_cap + 8) * 4);
        if (!line_off || !line_len) return GZ_ERR_HIP;
        HIPCHK (h, hipMemsetAsync (d_a, 0, sizeof (ABlock), h->stream));
        ZCHK (gz_text_lines (h, text, text_len, line_off, line_len, line_cap, &d_a->lines));
        hipLaunchKernelGGL (k_vb_bounds, dim3 ((2 * NV + 63) / 64), dim3 (64), 0, h->stream, (const uint32_t *)line_off, (const GzLinesResult *)&d_a->lines,
                            (const uint64_t *)d_vb_off, 2 * NV, text_len, d_first_line, &d_a->bad_bound);
        HIPCHK (h, hipMemcpyAsync (&a, d_a, sizeof (a), hipMemcpyDeviceToHost, h->stream));
        HIPCHK (h, hipMemcpyAsync (first_line.data (), d_first_line, 2 * (size_t)NV * 4, hipMemcpyDeviceToHost, h->stream));
        if ((rc = gz_sync (h)) < 0) return rc;
        if (a.lines.status == GZ_ST_OK) break;
        if (attempt || a.lines.n_lines > 0xfffffff0ull) { h->err = "line index does not fit"; return GZ_ERR; }
        line_cap = (uint32_t)a.lines.n_lines + 8;                          // (short lines: once more with the exact count)
    }
    T.mark ("lines+sync");
    if (a.bad_bound) { h->err = "a VBlock does not start / end at the start of a line"; return GZ_ERR_CORRUPT; }
    // (the reference segs every line's end into E1L / E2L: "\n" or "\r\n", fastq.c:1191-1203; the plans this driver takes make those
    //  contexts one constant snip, so a text with \r\n lines must not get through silently)
    if (a.lines.reserved) { h->err = "lines end in \\r\\n: not expressible with a constant end-of-line snip"; return GZ_ERR_CORRUPT; }
    const uint64_t n_lines = a.lines.n_lines;
    // a record is 4 lines (FASTQ) or - plan.record_lines == 1 - one line whose SEQ and QUAL are two of its items (SAM)
    const uint32_t RL = f->plan.record_lines == 1 ? 1 : 4;
    if (n_lines % RL) { h->err = "the text does not hold whole reads (4 lines each)"; return GZ_ERR_CORRUPT; }
    const uint32_t R = (uint32_t)(n_lines / RL);                           // records of the whole text (VBlocks hold whole records)
    K.r0.resize (NV + 1);
    std::vector<uint32_t> &r0 = K.r0;
    std::vector<uint32_t> r_end (NV);
    for (uint32_t v = 0; v < NV; v++) {
        if (first_line[v] % RL || first_line[NV + v] % RL) { h->err = "a VBlock does not hold whole reads (4 lines each)"; return GZ_ERR_CORRUPT; }
        r0[v] = first_line[v] / RL; r_end[v] = first_line[NV + v] / RL;
        vbs[v].n_reads = r_end[v] - r0[v];
    }

    // ---- reads, items, the columns of every (VBlock, context) ---------------------------------------------------------------
    const uint32_t NI = f->plan.n_seps + 1;
    WS (rec, uint32_t, (size_t)8 * (R + 8));
    uint32_t *l1_off = rec, *l1_len = rec + (R + 8), *seq_off = rec + 2 * (size_t)(R + 8), *seq_len = rec + 3 * (size_t)(R + 8),
             *l3_off = rec + 4 * (size_t)(R + 8), *l3_len = rec + 5 * (size_t)(R + 8), *qual_off = rec + 6 * (size_t)(R + 8), *qual_len = rec + 7 * (size_t)(R + 8);
    WS (item_off, uint32_t, (size_t)NI * R + 8);
    WS (item_len, uint32_t, (size_t)NI * R + 8);
    bool tokenized = false;
    if (RL == 4) ZCHK (gz_fastq_records (h, text, line_off, line_len, &d_a->lines, R, l1_off, l1_len, seq_off, seq_len, l3_off, l3_len, qual_off, qual_len, &d_a->fq));
    else {
        // one-line records (sam_seg_txt_line's split of a line into its tab-separated fields, src/sam_seg.c; QNAME further by its
        // flavor): the whole line is the container; SEQ and QUAL are the items the plan names. The items are needed before anything
        // else can be queued, so the tokenizer runs first here.
        if (f->plan.seq_item > f->plan.n_seps || f->plan.qual_item > f->plan.n_seps) { h->err = "plan: seq_item / qual_item"; return GZ_ERR_ARG; }
        l1_off = line_off; l1_len = line_len;
        HIPCHK (h, hipMemsetAsync (&d_a->fq, 0xff, sizeof (GzFastqResult), h->stream));          // (first_bad = none)
        HIPCHK (h, hipMemsetAsync (l3_len, 0, ((size_t)R + 8) * 4, h->stream));
        ZCHK (gz_tokenize_column_n (h, text, l1_off, l1_len, R, f->plan.seps, f->plan.sep_counts, f->plan.n_seps, item_off, item_len, &d_a->n_bad_items));
        tokenized = true;
        seq_off = item_off + (size_t)f->plan.seq_item * R; seq_len = item_len + (size_t)f->plan.seq_item * R;
        qual_off = item_off + (size_t)f->plan.qual_item * R; qual_len = item_len + (size_t)f->plan.qual_item * R;
    }
    // (FASTQ: line 1 of every read, gathered - see where the tokenizer is queued. Items are (offset, length) into `itext`; the SNIP_LOOKUP
    //  byte integer items point at is parked behind it, at ilookup)
    uint8_t *names = NULL; uint32_t *names_off = NULL, *names_len = NULL;
    const uint8_t *itext = text; uint32_t ilookup = lookup_off;
    if (RL == 4 && R && !getenv ("GZ_ZIP_NO_NAMES")) {
        const uint32_t names_cap = (uint32_t)text_len;          // (line 1 of every read: less than the text)
        if (!(names = (uint8_t *)ws_alloc (f, (size_t)names_cap + 64)) || !(names_off = (uint32_t *)ws_alloc (f, ((size_t)R + 8) * 4)) ||
            !(names_len = (uint32_t *)ws_alloc (f, ((size_t)R + 8) * 4))) return GZ_ERR_HIP;
        HIPCHK (h, hipMemcpyAsync (names + names_cap, lookup_byte, 16, hipMemcpyHostToDevice, h->stream));
        itext = names; ilookup = names_cap;
    }
    // VCF: the FORMAT subfields of every sample of every line as columns of lines x samples entries (vcf_seg_samples' split)
    const uint32_t NS = f->plan.n_samples, NSUB = f->plan.n_subfields;
    uint32_t *s_off = NULL, *s_len = NULL;
    const uint64_t cells = (uint64_t)R * NS;
    if (NS) {
        if (RL != 1 || !NSUB || cells * NSUB > 0xfffffff0ull) { h->err = "plan: samples need one-line records; at most 2^32 sample subfields per call"; return GZ_ERR_ARG; }
        uint32_t tab_cap = (uint32_t)std::min<uint64_t> (text_len, cells + 16ull * R + 1024);
        uint32_t *tab_after = NULL;
        for (int attempt = 0;; attempt++) {
            if (!(tab_after = (uint32_t *)ws_alloc (f, ((size_t)tab_cap + 8) * 4))) return GZ_ERR_HIP;
            ZCHK (gz_byte_index (h, text, text_len, '\t', tab_after, tab_cap, &d_a->tabs));
            GzLinesResult tr;
            HIPCHK (h, hipMemcpyAsync (&tr, &d_a->tabs, sizeof (tr), hipMemcpyDeviceToHost, h->stream));
            if ((rc = gz_sync (h)) < 0) return rc;
            if (tr.status == GZ_ST_OK) break;
            if (attempt || tr.n_lines > 0xfffffff0ull) { h->err = "tab index does not fit"; return GZ_ERR; }
            tab_cap = (uint32_t)tr.n_lines + 8;
        }
        uint8_t *missing = (uint8_t *)ws_alloc (f, cells * NSUB + 64);
        if (!(s_off = (uint32_t *)ws_alloc (f, (cells * NSUB + 8) * 4)) || !(s_len = (uint32_t *)ws_alloc (f, (cells * NSUB + 8) * 4)) || !missing) return GZ_ERR_HIP;
        HIPCHK (h, hipMemsetAsync (missing, 0, cells * NSUB, h->stream));
        ZCHK (gz_vcf_sample_columns (h, text, line_off, line_len, R, tab_after, &d_a->tabs, NS, NSUB, s_off, s_len, missing, &d_a->n_bad_samples));
        if (cells) KLAUNCH (h, k_any_set, dim3 ((uint32_t)((cells * NSUB + 4095) / 4096)), dim3 (256), 0, (const uint8_t *)missing, cells * NSUB, &d_a->n_missing);
    }
    // SQBITMAP's snip of every read, and what NONREF takes of it (fastq_seg_SEQ): a read of one repeated base is not stored
    uint8_t *sq_slots = NULL; uint32_t *sq_off = NULL, *sq_len = NULL, *nonref_len = NULL;
    if (f->seq_snip_ctx >= 0 || f->plan.line3_empty) {
        const GzFastqCtx *X = f->seq_snip_ctx >= 0 ? &f->ctxs[f->seq_snip_ctx] : NULL;
        if (!(sq_slots = (uint8_t *)ws_alloc (f, (size_t)R * 16 + 64)) || !(sq_off = (uint32_t *)ws_alloc (f, ((size_t)R + 8) * 4)) ||
            !(sq_len = (uint32_t *)ws_alloc (f, ((size_t)R + 8) * 4)) || !(nonref_len = (uint32_t *)ws_alloc (f, ((size_t)R + 8) * 4))) return GZ_ERR_HIP;
        GzdSeqSnip S; memset (&S, 0, sizeof (S));
        S.text = text; S.seq_off = seq_off; S.seq_len = seq_len; S.l3_len = l3_len; S.n = R;
        S.prefix_len = X ? X->snip_len : 1; if (X) memcpy (S.prefix, X->snip, X->snip_len); else S.prefix[0] = ' ';
        S.slots = sq_slots; S.snip_off = sq_off; S.snip_len = sq_len; S.nonref_len = nonref_len; S.n_line3 = &d_a->n_line3;
        if (R) KLAUNCH (h, k_seq_snips, dim3 ((R + 255) / 256), dim3 (256), 0, S);
        if (f->seq_snip_ctx < 0) nonref_len = NULL;                        // (only the line-3 check was wanted: SEQ as the plan without SQBITMAP has it)
    }
    // QUAL's snip of every read (fastq_seg_QUAL, src/fastq_qual.c:24-47), and what QUAL.local and CODEC_DOMQ take of the line: a line of one
    // repeated score segs the special snip and is left out (the callback's 0 bytes, :74) - from here on qual_len is that length
    uint8_t *q_slots = NULL; uint32_t *q_off = NULL, *q_len = NULL;
    const bool qual_col = f->qual_ctx >= 0 && f->ctxs[f->qual_ctx].snip_len != 0;
    if (qual_col) {
        const GzFastqCtx &X = f->ctxs[f->qual_ctx];
        uint32_t *qual_eff = NULL;
        if (!(q_slots = (uint8_t *)ws_alloc (f, (size_t)R * 4 + 64)) || !(q_off = (uint32_t *)ws_alloc (f, ((size_t)R + 8) * 4)) ||
            !(q_len = (uint32_t *)ws_alloc (f, ((size_t)R + 8) * 4)) || !(qual_eff = (uint32_t *)ws_alloc (f, ((size_t)R + 8) * 4))) return GZ_ERR_HIP;
        GzdQualSnip S; memset (&S, 0, sizeof (S));
        S.text = text; S.qual_off = qual_off; S.qual_len = qual_len; S.n = R;
        S.prefix_len = X.snip_len; memcpy (S.prefix, X.snip, X.snip_len);
        S.slots = q_slots; S.snip_off = q_off; S.snip_len = q_len; S.eff_len = qual_eff;
        if (R) KLAUNCH (h, k_qual_snips, dim3 ((R + 255) / 256), dim3 (256), 0, S);
        qual_len = qual_eff;
    }
    WS (d_vbstat, uint32_t, 2 * (size_t)NV + 2);

    // the dictionaries as every VBlock of this call clones them (ctx_clone)
    struct OlDev { const uint8_t *dict = NULL; const uint64_t *ci = NULL; const uint32_t *sl = NULL; uint32_t n = 0; };
    std::vector<OlDev> ol (NC), ol_r2 (NC);
    std::vector<uint8_t> has_pre (NC, 0);                  // the context's r2_node is not a word of the file yet: an R2 VBlock's first new node
    std::vector<std::vector<uint8_t>> pre_stage (NC);      // (host side of the uploads below: lives until the stream has taken them)
    for (uint32_t c = 0; c < NC; c++) {
        const uint8_t k = f->ctxs[c].kind;
        if (k != GZ_FQ_ITEM_TEXT && k != GZ_FQ_ITEM_INT && k != GZ_FQ_SEQ_SNIP && !(k == GZ_FQ_QUAL && qual_col)) continue;
        GzZctxView zv; gz_zctx_view (f->zctx[c], &zv);
        ol[c].n = zv.n_words;
        const GzFastqCtx &X = f->ctxs[c];
        has_pre[c] = X.r2_node_len && zctx_find (f->zctx[c], gz_snip_mix (X.r2_node, X.r2_node_len), X.r2_node, X.r2_node_len) == GZ_NO_WORD;
        if (zv.n_words) {
            uint8_t *d = (uint8_t *)ws_alloc (f, zv.dict_len + 16); uint64_t *ci = (uint64_t *)ws_alloc (f, (size_t)zv.n_words * 8); uint32_t *sl = (uint32_t *)ws_alloc (f, (size_t)zv.n_words * 4);
            if (!d || !ci || !sl) return GZ_ERR_HIP;
            HIPCHK (h, hipMemcpyAsync (d, zv.dict, zv.dict_len, hipMemcpyHostToDevice, h->stream));
            HIPCHK (h, hipMemcpyAsync (ci, zv.char_index, (size_t)zv.n_words * 8, hipMemcpyHostToDevice, h->stream));
            HIPCHK (h, hipMemcpyAsync (sl, zv.snip_len, (size_t)zv.n_words * 4, hipMemcpyHostToDevice, h->stream));
            ol[c].dict = d; ol[c].ci = ci; ol[c].sl = sl;
        }
        if (!has_pre[c]) continue;
        // ctx_create_node in front of everything an R2 VBlock segs (fastq.c:664-665): the node is the VBlock's first new one, ol_nodes.len, and
        // every snip new to the VBlock comes behind it. The column kernels number new nodes from the cloned words on, so R2 VBlocks are
        // given the cloned dictionary with that snip as one more word at its end: the same node indices, a count of 0 for it
        const uint32_t nw = zv.n_words + 1;
        std::vector<uint8_t> &st = pre_stage[c];
        st.resize ((size_t)zv.dict_len + X.r2_node_len + 1 + 8 + (size_t)nw * 12 + 16);
        uint8_t *sd = st.data (); uint64_t *sci = (uint64_t *)(sd + ((zv.dict_len + X.r2_node_len + 1 + 7) & ~(uint64_t)7)); uint32_t *ssl = (uint32_t *)(sci + nw);
        if (zv.dict_len) memcpy (sd, zv.dict, zv.dict_len);
        memcpy (sd + zv.dict_len, X.r2_node, X.r2_node_len); sd[zv.dict_len + X.r2_node_len] = 0;
        if (zv.n_words) { memcpy (sci, zv.char_index, (size_t)zv.n_words * 8); memcpy (ssl, zv.snip_len, (size_t)zv.n_words * 4); }
        sci[zv.n_words] = zv.dict_len; ssl[zv.n_words] = X.r2_node_len;
        uint8_t *dd = (uint8_t *)ws_alloc (f, st.size ());
        if (!dd) return GZ_ERR_HIP;
        HIPCHK (h, hipMemcpyAsync (dd, sd, st.size (), hipMemcpyHostToDevice, h->stream));
        ol_r2[c].dict = dd; ol_r2[c].ci = (const uint64_t *)(dd + ((uint8_t *)sci - sd)); ol_r2[c].sl = (const uint32_t *)(dd + ((uint8_t *)ssl - sd)); ol_r2[c].n = nw;
    }

    K.col.assign ((size_t)NV * NC, ZipCol ());
    auto COL = [&] (uint32_t v, uint32_t c) -> ZipCol & { return K.col[(size_t)v * NC + c]; };
    std::vector<GzIntColJob> icol_jobs; std::vector<GzDynIntJob> dyn_jobs; std::vector<GzBlobJob> blob_jobs, pre_jobs; std::vector<GzAcgtJob> acgt_jobs;
    std::vector<GzDomqJob> domq_jobs; std::vector<GzDomqFitJob> fit_jobs;
    const int qmode0 = f->qual_mode;                       // as the call finds it
    K.domq.clear (); K.qual_mode_applied = -1;
    GzDomqResult *d_domqres = NULL; uint32_t *d_fit = NULL;
    if (qmode0) {
        K.domq.resize (NV);
        if (!(d_domqres = (GzDomqResult *)ws_alloc (f, (size_t)NV * sizeof (GzDomqResult) + 16)) || !(d_fit = (uint32_t *)ws_alloc (f, (size_t)NV * 4 + 16))) return GZ_ERR_HIP;
        HIPCHK (h, hipMemsetAsync (d_fit, 0, (size_t)NV * 4 + 16, h->stream));
        HIPCHK (h, hipMemsetAsync (d_domqres, 0, (size_t)NV * sizeof (GzDomqResult), h->stream));
    }
    std::vector<GzColumnJob> &col_jobs = K.col_jobs;
    K.acgt_of_vb.assign (NV, -1);
    const size_t max_jobs = (size_t)NV * NC + 1;
    WS (d_colres, GzColumnResult, max_jobs);
    WS (d_dynres, GzDynIntResult, max_jobs);
    WS (d_icolres, uint64_t, 2 * max_jobs);          // n_values, status
    WS (d_blobres, uint64_t, max_jobs);
    WS (d_acgtres, uint64_t, 2 * max_jobs);          // has_x (u32) | pad, packed_len
    WS (d_seclen, uint32_t, 2 * max_jobs);           // device-resident payload length of every (VBlock, context) local / b250 section
    WS (d_b250st, int32_t, max_jobs);
    K.d_dynres = d_dynres; K.d_seclen = d_seclen; K.d_b250st = d_b250st;
    HIPCHK (h, hipMemsetAsync (d_icolres, 0, 2 * max_jobs * 8, h->stream));
    HIPCHK (h, hipMemsetAsync (d_seclen, 0, 2 * max_jobs * 4, h->stream));
    HIPCHK (h, hipMemsetAsync (d_b250st, 0, max_jobs * 4, h->stream));

    for (uint32_t v = 0; v < NV; v++) {
        const uint32_t n = vbs[v].n_reads, rr = r0[v];
        for (uint32_t c = 0; c < NC; c++) {
            const GzFastqCtx &X = f->ctxs[c];
            ZipCol &Z = COL (v, c);
            const bool ps = X.per_sample && NS;                            // a FORMAT subfield of every sample: lines x samples entries
            if (X.per_sample && (!NS || X.item >= NSUB)) { h->err = "plan: a per-sample context without samples / beyond n_subfields"; return GZ_ERR_ARG; }
            const uint32_t nn = ps ? n * NS : n;
            if (X.kind == GZ_FQ_ITEM_EXPECT) { Z.n = 0; continue; }             // (no context: checked below, once for the whole call)
            Z.n = nn;
            Z.sec_len_dev = d_seclen + 2 * ((size_t)v * NC + c);
            const uint32_t *io = ps ? s_off + (size_t)X.item * cells + (size_t)rr * NS : item_off + (size_t)X.item * R + rr,
                           *il = ps ? s_len + (size_t)X.item * cells + (size_t)rr * NS : item_len + (size_t)X.item * R + rr;
            const uint32_t *coff = io, *clen = il;
            if (X.kind == GZ_FQ_ITEM_INT || X.kind == GZ_FQ_ITEM_DELTA) {
                GzIntColJob j; memset (&j, 0, sizeof (j));
                j.text = ps ? text : itext; j.off = io; j.len = il; j.n = nn; j.nothing_char = X.nothing_char; j.lookup_off = ps ? lookup_off : ilookup; j.mode = X.kind == GZ_FQ_ITEM_DELTA;
                int64_t *vals = (int64_t *)ws_alloc (f, ((size_t)nn + 1) * 8); uint8_t *isn = (uint8_t *)ws_alloc (f, (size_t)nn + 16);
                if (!vals || !isn) return GZ_ERR_HIP;
                j.values = vals; j.is_nothing = isn;
                if (X.kind == GZ_FQ_ITEM_INT) {
                    uint32_t *so = (uint32_t *)ws_alloc (f, ((size_t)nn + 1) * 4), *sl = (uint32_t *)ws_alloc (f, ((size_t)nn + 1) * 4);
                    if (!so || !sl) return GZ_ERR_HIP;
                    j.snip_off = so; j.snip_len = sl; coff = so; clen = sl;
                }
                Z.icol_job = (int)icol_jobs.size ();
                j.n_values_dev = d_icolres + 2 * (size_t)Z.icol_job; j.status_dev = (int32_t *)(d_icolres + 2 * (size_t)Z.icol_job + 1);
                icol_jobs.push_back (j);
                GzDynIntJob dj; memset (&dj, 0, sizeof (dj));
                dj.values = vals; dj.is_nothing = isn; dj.n = nn; dj.nothing_char = X.nothing_char; dj.n_dev = j.n_values_dev;
                if (!(Z.local = (uint8_t *)ws_alloc (f, ((size_t)nn + 1) * 8))) return GZ_ERR_HIP;
                Z.local_cap = (uint64_t)nn * 8;
                dj.out = Z.local; Z.dyn_job = (int)dyn_jobs.size (); dj.result_dev = d_dynres + Z.dyn_job;
                dyn_jobs.push_back (dj);
            }
            if (X.kind == GZ_FQ_ITEM_TEXT || X.kind == GZ_FQ_ITEM_INT || X.kind == GZ_FQ_SEQ_SNIP || (X.kind == GZ_FQ_QUAL && qual_col)) {
                GzColumnJob j; memset (&j, 0, sizeof (j));
                j.text = ps ? text : itext; j.off = coff; j.len = clen; j.n = nn;
                if (X.kind == GZ_FQ_SEQ_SNIP) { j.text = sq_slots; j.off = sq_off + rr; j.len = sq_len + rr; }   // (generated text: 16-byte slots)
                if (X.kind == GZ_FQ_QUAL) { j.text = q_slots; j.off = q_off + rr; j.len = q_len + rr; }           // (generated text: 4-byte slots)
                uint64_t lead_bytes = 0;
                if (X.kind == GZ_FQ_ITEM_TEXT && X.snip_len) {
                    // every snip is `snip` + the item (sam_seg_CIGAR, src/sam_cigar.c:717-720): the column is gathered into a text of its own
                    GzBlobJob pj; memset (&pj, 0, sizeof (pj));
                    pj.text = ps ? text : itext; pj.off = coff; pj.len = clen; pj.n = nn; pj.pre_len = X.snip_len; memcpy (pj.pre, X.snip, X.snip_len);
                    lead_bytes = (uint64_t)nn * X.snip_len;
                    pj.out = (uint8_t *)ws_alloc (f, vbs[v].text_len + lead_bytes + 64);
                    pj.item_off = (uint32_t *)ws_alloc (f, ((size_t)nn + 1) * 4); pj.item_len = (uint32_t *)ws_alloc (f, ((size_t)nn + 1) * 4);
                    pj.out_len_dev = (uint64_t *)ws_alloc (f, 8);
                    if (!pj.out || !pj.item_off || !pj.item_len || !pj.out_len_dev) return GZ_ERR_HIP;
                    pre_jobs.push_back (pj);
                    j.text = pj.out; j.off = pj.item_off; j.len = pj.item_len;
                }
                Z.pre_node = vbs[v].r1 >= 0 && has_pre[c];
                const OlDev &O = Z.pre_node ? ol_r2[c] : ol[c];
                j.ol_dict = O.dict; j.ol_char_index = O.ci; j.ol_snip_len = O.sl; j.n_ol = O.n;
                // the dictionary of a column cannot exceed its snips + a NUL each; an item is at most the line
                const uint64_t dict_cap = X.kind == GZ_FQ_ITEM_INT ? (uint64_t)nn * 24 + 64 : X.kind == GZ_FQ_SEQ_SNIP ? (uint64_t)nn * 17 + 64 : X.kind == GZ_FQ_QUAL ? (uint64_t)nn * 5 + 64 :
                                          vbs[v].text_len + nn + lead_bytes + 64;
                j.node_index = (int32_t *)ws_alloc (f, ((size_t)nn + 1) * 4); j.dict = (uint8_t *)ws_alloc (f, dict_cap); j.dict_cap = dict_cap;
                j.node_char_index = (uint64_t *)ws_alloc (f, ((size_t)nn + 1) * 8); j.node_snip_len = (uint32_t *)ws_alloc (f, ((size_t)nn + 1) * 4);
                j.counts = (uint32_t *)ws_alloc (f, ((size_t)nn + O.n + 1) * 4); j.b250 = (uint8_t *)ws_alloc (f, (size_t)nn * 4 + 16);
                if (!j.node_index || !j.dict || !j.node_char_index || !j.node_snip_len || !j.counts || !j.b250) return GZ_ERR_HIP;
                Z.col_job = (int)col_jobs.size (); j.result_dev = d_colres + Z.col_job;
                Z.b250_seg = j.b250; Z.n_ol = ol[c].n;
                col_jobs.push_back (j);
            }
            if (X.kind == GZ_FQ_QUAL && qmode0 && n) {
                // N3: the four streams of CODEC_DOMQ (capacities: include/genozip_amd.h at gz_domq_columns)
                const uint64_t B = vbs[v].text_len;
                const size_t cap[4] = { (size_t)(2 * B + 64), (size_t)(B + B / 254 + 64), (size_t)n + 64, (size_t)B + 64 };
                ZipDomq &D = K.domq[v];
                for (int k = 0; k < 4; k++) if (!(D.out[k] = (uint8_t *)ws_alloc (f, cap[k]))) return GZ_ERR_HIP;
                GzDomqJob dj; memset (&dj, 0, sizeof (dj));
                dj.text = text; dj.off = qual_off + rr; dj.len = qual_len + rr; dj.n = n;
                dj.qual = D.out[0]; dj.runs = D.out[1]; dj.mplx = D.out[2]; dj.divr = D.out[3]; dj.result_dev = d_domqres + v;
                // (this process holds the file's first VBlock: nobody needs the streams if that one is not a fit)
                if (qmode0 < 0 && vbs[0].vblock_i == f->next_vblock_i () && vbs[0].n_reads) dj.only_if_dev = d_fit;
                domq_jobs.push_back (dj);
                if (qmode0 < 0) { GzDomqFitJob fj; memset (&fj, 0, sizeof (fj)); fj.text = text; fj.off = dj.off; fj.len = dj.len; fj.n = n; fj.fit_dev = d_fit + v; fit_jobs.push_back (fj); }
            }
            if (X.kind == GZ_FQ_SEQ || (X.kind == GZ_FQ_QUAL && qmode0 <= 0)) {
                GzBlobJob j; memset (&j, 0, sizeof (j));
                j.text = text; j.off = (X.kind == GZ_FQ_SEQ ? seq_off : qual_off) + rr; j.len = (X.kind == GZ_FQ_SEQ ? (nonref_len ? nonref_len : seq_len) : qual_len) + rr; j.n = n;
                Z.local_cap = vbs[v].text_len + 64 + (X.kind == GZ_FQ_SEQ ? (uint64_t)f->plan.seq_pad * n : 0);
                if (!(Z.local = (uint8_t *)ws_alloc (f, Z.local_cap + 64))) return GZ_ERR_HIP;
                j.out = Z.local; Z.blob_job = (int)blob_jobs.size (); j.out_len_dev = d_blobres + Z.blob_job;
                if (X.kind == GZ_FQ_SEQ && f->plan.seq_pad) { j.pad_to = f->plan.seq_pad; j.pad_byte = 'A'; }   // (sam_seg_SEQ_pad_nonref, src/sam_seq.c:224-229)
                blob_jobs.push_back (j);
                if (X.kind == GZ_FQ_SEQ) {
                    GzAcgtJob aj; memset (&aj, 0, sizeof (aj));
                    aj.seq = Z.local; aj.n_dev = j.out_len_dev; aj.n_max = Z.local_cap;
                    aj.packed = (uint8_t *)ws_alloc (f, gz_acgt_packed_len (Z.local_cap) + 64);
                    aj.x = Z.local;                                         // NONREF_X overlays NONREF (codec_acgt.c:66-70)
                    if (!aj.packed) return GZ_ERR_HIP;
                    K.acgt_of_vb[v] = (int)acgt_jobs.size ();
                    aj.has_x_dev = (uint32_t *)(d_acgtres + 2 * (size_t)K.acgt_of_vb[v]); aj.packed_len_dev = d_acgtres + 2 * (size_t)K.acgt_of_vb[v] + 1;
                    vbs[v].seq_packed = aj.packed;
                    acgt_jobs.push_back (aj);
                }
            }
        }
    }
    T.mark ("plan");
    // SEQ / QUAL are gathered first: the QUAL streams are the long pole of the whole call. When the file has no codec for them yet,
    // the trial compressions (a8) of the first VBlock's QUAL sit in front of that pole: their input - the first 99 999 bytes of
    // QUAL.local (codec.c:309) - is gathered on its own, ahead of everything, and they start on the second handle right away
    static const int trial_codecs[8] = { GZ_CODEC_RANB, GZ_CODEC_RANW, GZ_CODEC_RANb, GZ_CODEC_RANw, GZ_CODEC_ARTB, GZ_CODEC_ARTW, GZ_CODEC_ARTb, GZ_CODEC_ARTw };
    std::vector<GzStream> &trial = K.trial;               // 8 per candidate form of QUAL (plain / through DOMQ) that needs a codec
    trial.clear (); trial.reserve (16);                   // (the coder keeps pointers into it until the second handle is synchronised: never reallocated)
    std::vector<uint32_t> trial_ctx; std::vector<int> trial_domq;
    const bool own_first = vbs[0].vblock_i == f->next_vblock_i ();   // (vblock_i are consecutive over the processes: this one opens the call)
    bool want_trial = false;
    // (with the host's candidates in the race QUAL is tested in the merge phase like every other context: its trial needs the host's rows)
    if (f->h2 && !f->hostc.trial && own_first && f->qual_ctx >= 0 && vbs[0].n_reads && zip_vb_commits (f->plan, vbs[0])) { GzZctxView zv; gz_zctx_view (f->zctx[f->qual_ctx], &zv); want_trial = !zv.lcodec; }
    // Speculation. The handle remembers which coder the QUAL stream of its previous file ended up with. If it does, the long streams
    // are handed to the coders with THAT codec as soon as they are gathered, and this file's own trial - which decides, as always -
    // runs afterwards on the main handle, while the host merges: 4-5 ms of trial compressions no longer sit in front of the long
    // pole. A trial that chooses otherwise throws the work away (the streams are then coded with the rest: one call of one file pays).
    // It pays where the long streams are the long pole of the call - few VBlocks, of the reference's usual size (measured, ms per step
    // without / with: one file pair of 1 M reads in 14.7 MB VBlocks 98.7 / 96.0; in 4 MB VBlocks 42.0 / 47.3; 112 pairs per call
    // 286 / 297: there the trial only gets in the way of the rest) - so: at most 64 VBlocks, the longest with >= 5 M scores.
    // (GZ_ZIP_PRIOR_ONLY=1: what the handle has learned from its previous files is not used - every file starts from the built-in prior,
    //  as the first file on a fresh handle does: the "cold" figure of bench.py)
    static const int prior_guess[2] = { GZ_CODEC_ARTB, 0 };
    const char *prior_env = getenv ("GZ_ZIP_PRIOR_ONLY");
    const int *guess0 = (prior_env && *prior_env && *prior_env != '0') ? prior_guess : f->h_user->zip_qual_guess;
    const int guess[2] = { zip_is_host_codec (guess0[0]) ? 0 : guess0[0], zip_is_host_codec (guess0[1]) ? 0 : guess0[1] };
    uint64_t longest_text = 0;
    for (uint32_t v = 0; v < NV; v++) longest_text = std::max<uint64_t> (longest_text, vbs[v].text_len);
    const char *spec_env = getenv ("GZ_ZIP_SPECULATION");                  // "always": whatever the sizes (tests)
    const bool spec_always = spec_env && !strcmp (spec_env, "always");
    // QUAL through CODEC_DOMQ (binned scores, 104.3 / 110.3): not by itself either.
    bool may_spec = want_trial && !getenv ("GZ_ZIP_NO_SPECULATION") && (qmode0 == 0 ? guess[0] : qmode0 > 0 ? (spec_always && guess[1]) : (guess[0] || (spec_always && guess[1]))) &&
                    (spec_always || (NV <= 64 && longest_text >= 10000000));
    auto add_trials = [&] (const uint8_t *in, const uint32_t *len_dev, int as_domq) -> int {
        std::vector<GzStream> &T8 = may_spec ? K.spec_trial : trial;
        const size_t first = T8.size ();
        for (int k = 0; k < 8; k++) {
            GzStream st; memset (&st, 0, sizeof (st));
            st.in = in; st.in_len = 99999; st.in_len_dev = len_dev;                                          // (low half of a 64-bit length)
            st.codec = trial_codecs[k]; st.out_cap = gz_codec_est_size (st.codec, 99999);
            if (!(st.out = (uint8_t *)ws_alloc (f, (size_t)st.out_cap + 64))) return GZ_ERR_HIP;
            T8.push_back (st);
        }
        trial_ctx.push_back ((uint32_t)f->qual_ctx); trial_domq.push_back (as_domq);
        if (may_spec) return GZ_OK;                        // (launched once the seg phase has its results)
        int r = gz_wait_for (f->h2, h);
        if (r == GZ_OK && (r = gz_codec_compress_batch (f->h2, trial.data () + first, 8)) != GZ_OK) h->err = f->h2->err;
        return r;
    };
    K.spec_trial.reserve (16);
    if (want_trial && qmode0 <= 0) {                      // the plain form: enough of VBlock 0's reads to hold 99 999 bytes
        GzBlobJob tj; memset (&tj, 0, sizeof (tj));
        tj.text = text; tj.off = qual_off + r0[0]; tj.len = qual_len + r0[0]; tj.n = std::min<uint32_t> (vbs[0].n_reads, 100000);
        WS (d_trial_len, uint64_t, 2);
        if (!(tj.out = (uint8_t *)ws_alloc (f, vbs[0].text_len + 128))) return GZ_ERR_HIP;
        tj.out_len_dev = d_trial_len;
        ZCHK (gz_local_blob_columns (h, &tj, 1));
        ZCHK (add_trials (tj.out, (const uint32_t *)d_trial_len, 0));
    }
    for (size_t at = 0; at < blob_jobs.size (); at += 32768) ZCHK (gz_local_blob_columns (h, blob_jobs.data () + at, (int)std::min<size_t> (32768, blob_jobs.size () - at)));
    ZCHK (gz_domq_fit (h, fit_jobs.data (), (int)fit_jobs.size ()));
    ZCHK (gz_domq_columns (h, domq_jobs.data (), (int)domq_jobs.size ()));
    // the few numbers the launch of the long streams needs come back on their own, ahead of the columns' results (pinned memory)
    const size_t eb_blob = blob_jobs.size () * 8, eb_fit = qmode0 ? (size_t)NV * 4 + 8 : 0, eb_res = qmode0 ? (size_t)NV * sizeof (GzDomqResult) : 0;
    // (... and, further on in the same stretch, everything else the seg kernels report)
    auto up64 = [] (size_t x) { return (x + 63) & ~(size_t)63; };
    const size_t n_col_jobs = col_jobs.size ();
    const size_t rb0 = up64 (eb_blob + eb_fit + eb_res + 64);
    const size_t rb_col = rb0, rb_pack = rb_col + up64 (n_col_jobs * sizeof (GzColumnResult)), rb_tot = rb_pack + up64 (n_col_jobs * sizeof (GzdPackJob)), rb_dyn = rb_tot + 64,
                 rb_icol = rb_dyn + up64 (dyn_jobs.size () * sizeof (GzDynIntResult)), rb_acgt = rb_icol + up64 (icol_jobs.size () * 16), rb_stat = rb_acgt + up64 (acgt_jobs.size () * 16),
                 rb_a = rb_stat + up64 (2 * (size_t)NV * 4), rb_end = rb_a + up64 (sizeof (ABlock));
    uint8_t *eb = zip_pinned (f, rb_end + 64);
    if (!eb) return GZ_ERR_HIP;
    if (eb_blob) HIPCHK (h, hipMemcpyAsync (eb, d_blobres, eb_blob, hipMemcpyDeviceToHost, h->stream));
    if (qmode0 && NV) {
        HIPCHK (h, hipMemcpyAsync (eb + eb_blob, d_fit, (size_t)NV * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK (h, hipMemcpyAsync (eb + eb_blob + eb_fit, d_domqres, eb_res, hipMemcpyDeviceToHost, h->stream));
    }
    if (!f->ev_early) HIPCHK (h, hipEventCreateWithFlags (&f->ev_early, hipEventDisableTiming));
    HIPCHK (h, hipEventRecord (f->ev_early, h->stream));
    // (not decided yet whether QUAL goes through DOMQ: both forms are tried, the read-back says which one counts)
    if (want_trial && qmode0) ZCHK (add_trials (K.domq[0].out[0], (const uint32_t *)&d_domqres[0].qual_len, 1));
    // The columns' kernels (line 1 -> items -> contexts) fill the device for ~2.5 ms of a 1 M-pair file. Queued HERE - behind the gathers, in front of
    // the launch of the long streams - they are what the persistent chain's workgroups (a whole compute unit each) wait for before they can start at
    // all: measured, the chain kernel started executing ~3.3 ms into the default step although it was launched at ~1.6. With few VBlocks, whose step is
    // the latency of the long chains, they are therefore queued AFTER the long streams' launch (defer_columns): the main path that needs them has 20 ms
    // of slack there. With many VBlocks (the streamed form) the device's time is what counts and they go first, as they always did.
    const size_t NCJ = col_jobs.size ();
    std::vector<GzdPackJob> pack (NCJ);
    uint8_t *d_staging = NULL;
    std::vector<uint64_t> icolres;
    uint64_t pack_total[2] = { 0, 1 };
    uint8_t *rb = eb;                                      // (one page-locked stretch for both read-backs)
    K.blobres.resize (blob_jobs.size () + 1);
    auto queue_columns = [&] () -> int {
        // (the items of line 1 and the VBlock statistics are only needed from here on: queued behind the QUAL gather, which the long pole waits for)
        if (!tokenized) {
            // FASTQ: line 1 of every read is gathered into a text of its own first (`names`, allocated above: one coalesced pass over a third of the
            // text's cache lines), and the tokenizer and every item kernel behind it - each a pass of its own with a thread per 2 - 8 byte snip -
            // read THAT: 63 bytes per read side by side instead of a 128-byte line of the original text per snip and pass, 368 bytes apart
            // (profiles/round4_pmc.json: 9.4 GB of the step's 50 were k_tokenize_n / k_icol_* / k_col_insert fetching such lines)
            if (names) {
                GzBlobJob nj; memset (&nj, 0, sizeof (nj));
                nj.text = text; nj.off = l1_off; nj.len = l1_len; nj.n = R; nj.out = names; nj.item_off = names_off; nj.item_len = names_len;
                WS (d_names_len, uint64_t, 2);
                nj.out_len_dev = d_names_len;
                ZCHK (gz_local_blob_columns (h, &nj, 1));
                ZCHK (gz_tokenize_column_n (h, names, names_off, l1_len, R, f->plan.seps, f->plan.sep_counts, f->plan.n_seps, item_off, item_len, &d_a->n_bad_items));
            }
            else ZCHK (gz_tokenize_column_n (h, text, l1_off, l1_len, R, f->plan.seps, f->plan.sep_counts, f->plan.n_seps, item_off, item_len, &d_a->n_bad_items));
        }
        hipLaunchKernelGGL (k_vb_stats, dim3 (NV), dim3 (256), 2048, h->stream, (const uint32_t *)line_off, (const uint32_t *)seq_len, (const uint32_t *)d_first_line,
                            (const uint64_t *)(d_vb_off + NV), NV, RL, d_vbstat);
        ZCHK (gz_int_columns (h, icol_jobs.data (), (int)icol_jobs.size ()));
        // (column tables hold at most 65 535 rows per call)
        for (uint32_t c = 0; c < NC && R; c++) {                                     // text the plan's containers carry as prefixes must be there, in every record
            const GzFastqCtx &X = f->ctxs[c];
            if (X.kind != GZ_FQ_ITEM_EXPECT) continue;
            GzdExpect E; memset (&E, 0, sizeof (E));
            E.text = itext; E.off = item_off + (size_t)X.item * R; E.len = item_len + (size_t)X.item * R; E.n = R; E.want_len = X.snip_len; memcpy (E.want, X.snip, X.snip_len);
            E.n_bad = &d_a->n_unexpected;
            KLAUNCH (h, k_item_expect, dim3 ((R + 255) / 256), dim3 (256), 0, E);
        }
        for (size_t at = 0; at < pre_jobs.size (); at += 32768) ZCHK (gz_local_blob_columns (h, pre_jobs.data () + at, (int)std::min<size_t> (32768, pre_jobs.size () - at)));
        for (size_t at = 0; at < col_jobs.size (); at += 32768) ZCHK (gz_ctx_seg_columns (h, col_jobs.data () + at, (int)std::min<size_t> (32768, col_jobs.size () - at)));
        for (size_t at = 0; at < dyn_jobs.size (); at += 32768) ZCHK (gz_dyn_int_columns (h, dyn_jobs.data () + at, (int)std::min<size_t> (32768, dyn_jobs.size () - at)));
        ZCHK (gz_acgt_pack_batch (h, acgt_jobs.data (), (int)acgt_jobs.size ()));

        // what the merge needs from every column, packed into one stretch
        uint64_t pack_cap = 64;
        for (size_t k = 0; k < NCJ; k++) {
            const GzColumnJob &j = col_jobs[k];
            pack[k].dict = j.dict; pack[k].nci = j.node_char_index; pack[k].nsl = j.node_snip_len; pack[k].counts = j.counts; pack[k].n_ol = j.n_ol; pack[k].res = j.result_dev;
            pack_cap += j.dict_cap + 16ull * j.n + 4ull * j.n_ol + 64;
        }
        // (worst case = every snip a new word; the usual case is a few hundred bytes per column)
        const uint64_t pack_cap_used = std::min<uint64_t> (pack_cap, (uint64_t)256 << 20);
        WS (d_pack, GzdPackJob, NCJ + 1);
        WS (d_pack_total, uint64_t, 2);
        if (!(d_staging = (uint8_t *)ws_alloc (f, pack_cap_used))) return GZ_ERR_HIP;
        if (NCJ) {
            HIPCHK (h, hipMemcpyAsync (d_pack, pack.data (), NCJ * sizeof (GzdPackJob), hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL (k_pack_sizes, dim3 (1), dim3 (256), 257 * 8, h->stream, d_pack, (uint32_t)NCJ, pack_cap_used, d_pack_total);
            hipLaunchKernelGGL (k_pack_copy, dim3 ((uint32_t)NCJ), dim3 (256), 0, h->stream, (const GzdPackJob *)d_pack, d_staging, (const uint64_t *)d_pack_total);
        }
        // ---- read back (second wait): queued behind the seg kernels now, into page-locked memory - a copy into pageable memory would
        // hold the host until the stream gets there, and the host has the long streams to launch in the meantime
        K.colres.resize (NCJ + 1); K.dynres.resize (dyn_jobs.size () + 1);
        icolres.assign (2 * icol_jobs.size () + 2, 0);
        K.acgtres.resize (2 * acgt_jobs.size () + 2); K.vbstat.resize (2 * (size_t)NV + 2);
        if (NCJ) {
            HIPCHK (h, hipMemcpyAsync (rb + rb_col, d_colres, NCJ * sizeof (GzColumnResult), hipMemcpyDeviceToHost, h->stream));
            HIPCHK (h, hipMemcpyAsync (rb + rb_pack, d_pack, NCJ * sizeof (GzdPackJob), hipMemcpyDeviceToHost, h->stream));
            HIPCHK (h, hipMemcpyAsync (rb + rb_tot, d_pack_total, 16, hipMemcpyDeviceToHost, h->stream));
        }
        if (!dyn_jobs.empty ())  HIPCHK (h, hipMemcpyAsync (rb + rb_dyn, d_dynres, dyn_jobs.size () * sizeof (GzDynIntResult), hipMemcpyDeviceToHost, h->stream));
        if (!icol_jobs.empty ()) HIPCHK (h, hipMemcpyAsync (rb + rb_icol, d_icolres, icol_jobs.size () * 16, hipMemcpyDeviceToHost, h->stream));
        if (!acgt_jobs.empty ()) HIPCHK (h, hipMemcpyAsync (rb + rb_acgt, d_acgtres, acgt_jobs.size () * 16, hipMemcpyDeviceToHost, h->stream));
        HIPCHK (h, hipMemcpyAsync (rb + rb_stat, d_vbstat, 2 * (size_t)NV * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK (h, hipMemcpyAsync (rb + rb_a, d_a, sizeof (a), hipMemcpyDeviceToHost, h->stream));
        return GZ_OK;
    };
    const char *defer_env = getenv ("GZ_ZIP_DEFER");                      // "always": whatever the sizes (tests); "never"
    const bool defer_columns = f->h2 && !(defer_env && !strcmp (defer_env, "never")) && ((defer_env && !strcmp (defer_env, "always")) || (NV <= 64 && longest_text >= 10000000));
    if (!defer_columns) ZCHK (queue_columns ());
    T.mark ("queue");
    // ---- as soon as the gathered QUAL (and what CODEC_DOMQ makes of it) is there - the columns are still being evaluated:
    HIPCHK (h, hipEventSynchronize (f->ev_early));
    T.mark ("early-wait");
    if (eb_blob) memcpy (K.blobres.data (), eb, eb_blob);
    const uint32_t *fits = (const uint32_t *)(eb + eb_blob); const GzDomqResult *domqres = (const GzDomqResult *)(eb + eb_blob + eb_fit);
    for (uint32_t v = 0; qmode0 && v < NV; v++) {
        ZipDomq &D = K.domq[v];
        D.fit = fits[v];
        if (!vbs[v].n_reads) { memset (&D.res, 0, sizeof (D.res)); D.res.status = 1; continue; }
        memcpy (&D.res, &domqres[v], sizeof (D.res));
        if (qmode0 < 0 && vbs[0].vblock_i == f->next_vblock_i () && vbs[0].n_reads && !fits[0]) { memset (&D.res, 0, sizeof (D.res)); D.res.status = 1; }   // (skipped)
        if (D.res.status == 1) zip_base64 (D.res.denorm, (size_t)D.res.num_doms * D.res.num_norm_qs, D.snip);      // codec_domq.c:232-244
    }
    // this process holds the call's first VBlock and the file has not decided yet: that VBlock decides (codec.c:403-445), so the
    // long streams can be handed to the coders below; every process arrives at the same in the merge (from the blobs)
    int qmode = qmode0;
    if (qmode0 < 0 && own_first && NV) qmode = K.domq[0].fit ? GZ_CODEC_DOMQ : 0;
    if (f->qual_ctx >= 0) for (uint32_t v = 0; v < NV; v++) {
        ZipCol &Z = COL (v, (uint32_t)f->qual_ctx);
        if (Z.blob_job >= 0) { Z.local_len = K.blobres[Z.blob_job]; Z.ltype = GZ_LT_BLOB; Z.has_local = Z.local_len != 0; }
    }
    // ---- the long streams go first, on a handle of their own: QUAL locals (length known on the host, no dependence on the merge)
    // are handed to the coders as soon as they are gathered, so that their strictly serial chains run beside the trial compressions and the short
    // sections instead of after them. Their codec must be known for that: committed in the file, or decided here by trial on
    // the call's first VBlock - which only the process that owns that VBlock may do (a serial run commits VBlock 1's choice).
    K.early.clear ();
    if (qmode >= 0) {
        if (qmode == GZ_CODEC_DOMQ) for (uint32_t v = 0; v < NV; v++) if (K.domq[v].res.status != 1) {
            vbs[v].status = GZ_ERR_CORRUPT; h->err = "QUAL: a score outside ' '..'~' (codec_domq.c:150-153)"; return GZ_ERR_CORRUPT; }
        zip_apply_qual_mode (f, qmode);
    }
    // Coding the QUAL streams ahead of everything else costs the seg phase a wait for their trial compressions (~10 ms: one serial model
    // + chain over a 100 KB sample) and the launch itself. That buys the strictly serial chains of LONG streams a head start of the whole
    // merge phase - and nothing when the streams are short (QUAL through CODEC_DOMQ in 16 MB VBlocks: ~0.5 M symbols, 8 ms of chain): then
    // QUAL is a context like any other - tested in the merge phase with the rest, coded in the finish phase (BAM from text: 70.7 -> see
    // DESIGN section 4). Same bytes either way: the same sample, the same rule.
    uint64_t longest = 0;
    if (f->qual_ctx >= 0) for (uint32_t v = 0; v < NV; v++) longest = std::max<uint64_t> (longest, COL (v, (uint32_t)f->qual_ctx).local_len);
    const char *early_env = getenv ("GZ_ZIP_EARLY_MIN");
    const uint64_t early_min = early_env ? strtoull (early_env, NULL, 10) : 1500000ull;
    bool early_worth = longest >= early_min || spec_always;
    // (nothing to code ahead on the device: the host's candidates are in the race and the file has no codec yet, or the file's codec is one of the host's)
    { GzZctxView zq; if (f->qual_ctx >= 0) { gz_zctx_view (f->zctx[f->qual_ctx], &zq); if ((f->hostc.trial && !zq.lcodec) || zip_is_host_codec (zq.lcodec)) early_worth = false; } }
    if (!early_worth && f->h2 && NV && qmode >= 0 && f->qual_ctx >= 0) {
        GzZctxView zv; gz_zctx_view (f->zctx[f->qual_ctx], &zv);
        if (zv.lcodec && !zip_is_host_codec (zv.lcodec)) early_worth = true;                 // (the file knows its codec: nothing to wait for, the streams may as well start now)
    }
    if (!early_worth) { may_spec = false; K.spec_trial.clear (); }
    if (may_spec) {                                        // now that the lengths are known: is it worth it?
        if (qmode < 0 || !guess[qmode == GZ_CODEC_DOMQ] || ((longest < 5000000 || qmode == GZ_CODEC_DOMQ) && !spec_always)) {
            // no: the trial after all, on the second handle, and the long streams wait for it (as without speculation, a little later)
            may_spec = false;
            trial.swap (K.spec_trial);
            int r = gz_wait_for (f->h2, h);
            for (size_t t = 0; r == GZ_OK && t < trial_domq.size (); t++)
                if ((r = gz_codec_compress_batch (f->h2, trial.data () + 8 * t, 8)) != GZ_OK) h->err = f->h2->err;
            if (r != GZ_OK) { (void)gz_sync (f->h2); return r; }
        }
    }
    if (f->h2 && NV && early_worth) {
        if (!trial.empty () && (rc = gz_sync (f->h2)) < 0) { h->err = f->h2->err; return rc; }
        for (uint32_t c = 0; c < NC; c++) {
            if (f->ctxs[c].kind != GZ_FQ_QUAL || qmode < 0) continue;
            GzZctxView zv; gz_zctx_view (f->zctx[c], &zv);
            int codec = zv.lcodec;
            if (!codec && may_spec) {                                  // speculation: the previous file's coder for this form of QUAL
                const int g = guess[qmode == GZ_CODEC_DOMQ];
                if (g && COL (0, c).local_len >= 50) { codec = g; K.spec = true; K.spec_codec = g; }
            }
            for (size_t t = 0; !codec && !may_spec && t < trial_ctx.size (); t++) {
                if (trial_ctx[t] != c || trial_domq[t] != (qmode == GZ_CODEC_DOMQ) || COL (0, c).local_len < 50) continue;            // codec.c:311-312: too small a sample decides nothing
                const uint32_t sample = (uint32_t)std::min<uint64_t> (COL (0, c).local_len, 99999);
                uint32_t best_size = sample; codec = GZ_CODEC_NONE;                     // NONE: the bare length (codec.c:324)
                for (int k = 0; k < 8; k++) {
                    const GzStream &st = trial[8 * t + k];
                    if (st.status != GZ_OK) { h->err = "trial compression failed"; return GZ_ERR; }
                    if (st.out_len + 28 < best_size) { best_size = st.out_len + 28; codec = st.codec; }   // framed (codec.c:328-331), ties -> first
                }
                K.votes.push_back ({ c, 1, vbs[0].vblock_i, (uint32_t)codec });
            }
            if (!codec) continue;                                     // (not decided here: coded with the rest, below)
            for (uint32_t v = 0; v < NV; v++) {
                ZipCol &Z = COL (v, c);
                if (!Z.has_local || Z.local_len < 50 || Z.local_len > 0xffffffffull) continue;
                GzStream st; memset (&st, 0, sizeof (st));
                st.in = Z.local; st.in_len = (uint32_t)Z.local_len; st.codec = codec; st.out_cap = gz_codec_est_size (codec, Z.local_len);
                if (!(st.out = (uint8_t *)ws_alloc (f, (size_t)st.out_cap + 64))) return GZ_ERR_HIP;
                Z.early = (int)K.early.size (); Z.lcodec = (uint8_t)codec;
                K.early.push_back (st);
            }
        }
        T.mark ("early-plan");
        if (!K.early.empty ()) {
            if (!(K.d_early_len = (uint32_t *)ws_alloc (f, K.early.size () * 4))) return GZ_ERR_HIP;
            for (size_t k = 0; k < K.early.size (); k++) K.early[k].out_len_dev = K.d_early_len + k;
            // (their inputs were gathered on this handle's stream: complete, the event above was waited for)
            GzHandle *h2 = f->h2;
            if ((rc = gz_codec_compress_batch (h2, K.early.data (), (int)K.early.size ())) != GZ_OK) { h->err = h2->err; return rc; }
        }
    }
    T.mark ("early");
    if (defer_columns) { ZCHK (queue_columns ()); T.mark ("queue-columns"); }
    rc = gz_sync (h);
    // (the results of the seg kernels were copied to page-locked memory behind them, see above: into their places)
    if (NCJ) { memcpy (K.colres.data (), rb + rb_col, NCJ * sizeof (GzColumnResult)); memcpy (pack.data (), rb + rb_pack, NCJ * sizeof (GzdPackJob)); memcpy (pack_total, rb + rb_tot, 16); }
    if (!dyn_jobs.empty ())  memcpy (K.dynres.data (), rb + rb_dyn, dyn_jobs.size () * sizeof (GzDynIntResult));
    if (!icol_jobs.empty ()) memcpy (icolres.data (), rb + rb_icol, icol_jobs.size () * 16);
    if (!acgt_jobs.empty ()) memcpy (K.acgtres.data (), rb + rb_acgt, acgt_jobs.size () * 16);
    memcpy (K.vbstat.data (), rb + rb_stat, 2 * (size_t)NV * 4);
    memcpy (&a, rb + rb_a, sizeof (a));
    // (from here on the second handle may be at work on this call's buffers: it is waited for before an error is returned)
#define ZIP_FAIL(code) do { if (f->h2) (void)gz_sync (f->h2); return (code); } while (0)
    if (rc < 0) ZIP_FAIL (rc);
    T.mark ("seg-sync");
    if (a.fq.first_bad != 0xffffffffu) {
        for (uint32_t v = 0; v < NV; v++) if (a.fq.first_bad >= r0[v] && a.fq.first_bad < r_end[v]) vbs[v].status = GZ_ERR_CORRUPT;
        h->err = "not FASTQ: a read is not '@'.. / SEQ / '+'.. / QUAL of SEQ's length (fastq.c:1008-1010,1076,1121)"; ZIP_FAIL (GZ_ERR_CORRUPT);
    }
    if (f->plan.line3_empty && a.n_line3) { h->err = "line 3 of a read is more than '+' (the plan says L3_EMPTY; fastq_desc.c:35-37)"; ZIP_FAIL (GZ_ERR_CORRUPT); }
    if (NS && (a.n_bad_samples || a.n_missing)) { h->err = "VCF: a line without 9 + n_samples fields, a sample with more subfields than the plan's, or one that leaves subfields out (not supported by this driver)"; ZIP_FAIL (GZ_ERR_CORRUPT); }
    if (a.n_unexpected) { h->err = "an item the plan expects to be constant (GZ_FQ_ITEM_EXPECT: a tag name its container carries as a prefix) is something else in some record"; ZIP_FAIL (GZ_ERR_CORRUPT); }
    if (a.n_bad_items) { h->err = "a line 1 does not fit the container of the plan (the reference would re-discover the flavor, qname.c:823-826)"; ZIP_FAIL (GZ_ERR_CORRUPT); }
    for (size_t k = 0; k < icol_jobs.size (); k++)
        if ((int32_t)icolres[2 * k + 1] == GZ_ST_CORRUPT) { h->err = "an ordered item is not an integer (qname.c:750-756)"; ZIP_FAIL (GZ_ERR_CORRUPT); }
    for (size_t k = 0; k < NCJ; k++) if (K.colres[k].status != 1) { h->err = "column dictionary capacity"; ZIP_FAIL (GZ_ERR); }
    if (!pack_total[1]) { h->err = "merge staging buffer too small"; ZIP_FAIL (GZ_ERR); }
    // speculation: this file's own trial, on the main handle - idle now until the merge (host) has the dictionaries - and only of
    // the form of QUAL the file uses; its sizes are read in the merge phase, with the other trial compressions'
    if (!K.spec_trial.empty () && qmode >= 0 && COL (0, (uint32_t)f->qual_ctx).local_len >= 50) {     // (codec.c:311-312: too small a sample decides nothing)
        for (size_t t = 0; t < trial_domq.size () && !K.spec_pending; t++) if (trial_domq[t] == (qmode == GZ_CODEC_DOMQ)) {
            if ((rc = gz_codec_compress_batch (h, K.spec_trial.data () + 8 * t, 8)) != GZ_OK) ZIP_FAIL (rc);
            K.spec_pending = true; K.spec_t = t;
        }
    }
    f->stage.resize (pack_total[0] + 16);
    if (pack_total[0] && hipMemcpy (f->stage.data (), d_staging, pack_total[0], hipMemcpyDeviceToHost) != hipSuccess) { h->err = "hipMemcpy (merge staging)"; ZIP_FAIL (GZ_ERR_HIP); }

    // ---- the merge blob: per VBlock, per context, what ctx_merge_in_one_vctx reads of the VBlock's context ------------------
    K.blob.clear ();
    K.own_snip.assign ((size_t)NV * NC, std::vector<uint8_t> ());
    for (uint32_t v = 0; v < NV; v++) {
        ZipBlobVB hv = { vbs[v].vblock_i, vbs[v].r1 >= 0 ? vbs[vbs[v].r1].vblock_i : 0, NC, 0 };
        if (qmode0) hv.qual = (qmode0 < 0 ? 1u : 0u) | (K.domq[v].fit ? 2u : 0u) | (K.domq[v].res.status != 1 ? 4u : 0u);
        blob_put (K.blob, &hv, sizeof (hv));
        for (uint32_t c = 0; c < NC; c++) {
            const GzFastqCtx &X = f->ctxs[c];
            ZipCol &Z = COL (v, c);
            ZipMergeRec r; memset (&r, 0, sizeof (r));
            if (Z.dyn_job >= 0) { Z.local_len = K.dynres[Z.dyn_job].len; Z.ltype = K.dynres[Z.dyn_job].ltype; Z.has_local = Z.local_len != 0; }
            if (Z.blob_job >= 0 && X.kind == GZ_FQ_SEQ) { Z.local_len = K.blobres[Z.blob_job]; Z.ltype = GZ_LT_BLOB; Z.has_local = Z.local_len != 0; }   // (QUAL: above)
            if (X.kind == GZ_FQ_SEQ) {                                     // NONREF itself leaves the path 2-bit packed; what stays is NONREF_X
                const int aj = K.acgt_of_vb[v];
                vbs[v].n_bases = Z.local_len; vbs[v].seq_packed_len = K.acgtres[2 * aj + 1]; vbs[v].seq_has_x = (uint32_t)K.acgtres[2 * aj] != 0;
                Z.has_local = vbs[v].seq_has_x != 0; Z.ltype = GZ_LT_SUPP;   // NONREF_X.ltype (codec_acgt.c:36-41)
            }
            // (QUAL: ctx->local.len as the segmenter leaves it - the scores of the lines that are not one repeated score; under CODEC_DOMQ alone
            //  nothing is gathered: QUALMPLX's byte per such line stands in, only its being zero is looked at)
            r.n = Z.n; r.local_len = X.kind == GZ_FQ_QUAL ? (Z.blob_job >= 0 ? K.blobres[Z.blob_job] : qmode0 && Z.n ? K.domq[v].res.mplx_len : 0) : Z.local_len; r.ats_node = -1;
            if (qmode0 && Z.n && (X.kind == GZ_FQ_QUAL || X.kind == GZ_FQ_QUAL_AUX)) {
                const ZipDomq &D = K.domq[v];
                r.domq_local_len = !D.res.mplx_len ? 0 : X.kind == GZ_FQ_QUAL ? D.res.qual_len : X.item == 0 ? D.res.runs_len : X.item == 1 ? D.res.mplx_len : D.res.divr_len;
                // seg_by_ctx (denorm_snip) (codec_domq.c:244): one b250 entry. A VBlock whose every line is one repeated score hands CODEC_DOMQ no
                // line at all: no dom, an EMPTY table, and seg_by_ctx of a snip of length 0 is WORD_INDEX_EMPTY (context.c:331-335) - state 3 with
                // dict_len 0
                if (X.kind == GZ_FQ_QUAL_AUX && X.item == 0 && D.res.status == 1) {
                    r.state = 3; r.n = 1; r.dict_len = D.snip.size ();
                    blob_put (K.blob, &r, sizeof (r)); blob_put (K.blob, D.snip.data (), D.snip.size ());
                    continue;
                }
            }
            if (X.kind == GZ_FQ_SEQ || (X.kind == GZ_FQ_QUAL && Z.col_job < 0) || X.kind == GZ_FQ_QUAL_AUX || !Z.n) { blob_put (K.blob, &r, sizeof (r)); continue; }
            if (X.kind == GZ_FQ_TOPLEVEL) {
                // container_seg of the VBlock's TOPLEVEL (fastq.c:845-943): repeats = the VBlock's reads (Container.repeats: bits 8-31 of the
                // first little-endian word, container.h:74-80; written little endian since 15.0.84, container.c:48-55), then
                // SNIP_CONTAINER + base64 of the struct + the prefixes (container.c:35-64). One b250 entry.
                std::vector<uint8_t> con (X.snip, X.snip + X.con_len), b64;
                if (Z.n > 0xfffff0u) { h->err = "more reads in a VBlock than a container can repeat (CONTAINER_MAX_REPEATS)"; ZIP_FAIL (GZ_ERR_ARG); }
                con[1] = (uint8_t)Z.n; con[2] = (uint8_t)(Z.n >> 8); con[3] = (uint8_t)(Z.n >> 16);
                zip_base64 (con.data (), con.size (), b64);
                std::vector<uint8_t> &own = K.own_snip[(size_t)v * NC + c];
                own.assign (1, 4 /* SNIP_CONTAINER */); own.insert (own.end (), b64.begin (), b64.end ()); own.insert (own.end (), X.snip + X.con_len, X.snip + X.snip_len);
                r.state = 3; r.n = 1; r.dict_len = own.size ();
                blob_put (K.blob, &r, sizeof (r)); blob_put (K.blob, own.data (), own.size ());
                continue;
            }
            if (Z.col_job < 0) { r.state = 2; r.n = Z.n * (X.segs_per_line ? X.segs_per_line : 1); blob_put (K.blob, &r, sizeof (r)); continue; }
            const GzColumnResult &cr = K.colres[Z.col_job];
            const GzdPackJob &p = pack[Z.col_job];
            const uint32_t *counts = (const uint32_t *)(f->stage.data () + p.at[3]);
            // an R2 VBlock's pre-created node (fastq.c:664-665): node ol_nodes.len of the VBlock, in front of the column's own new nodes
            const uint32_t pre = Z.pre_node ? 1 : 0, pre_len = pre ? X.r2_node_len : 0;
            r.state = 1; r.n_ol = Z.n_ol; r.n_new = cr.n_new + pre; r.dict_len = cr.dict_len + (pre ? pre_len + 1 : 0); r.seg_b250_len = cr.b250_len; r.b250_count = cr.b250_count;
            r.all_the_same = cr.all_the_same != 0;
            if (r.all_the_same) {
                // the one node of the column: an ol word (the index with a count) or the first new node of the column
                for (uint32_t k = 0; k < Z.n_ol && r.ats_node < 0; k++) if (counts[k]) r.ats_node = (int32_t)k;
                if (r.ats_node < 0 && cr.n_new) r.ats_node = (int32_t)(Z.n_ol + pre);    // (-1: every snip was empty / missing: not droppable)
            }
            blob_put (K.blob, &r, sizeof (r));
            if (!pre) {
                blob_put (K.blob, f->stage.data () + p.at[0], cr.dict_len);
                blob_put (K.blob, f->stage.data () + p.at[1], 8 * (size_t)cr.n_new);
                blob_put (K.blob, f->stage.data () + p.at[2], 4 * (size_t)cr.n_new);
            }
            else {
                std::vector<uint8_t> d (r.dict_len); std::vector<uint64_t> ci (r.n_new); std::vector<uint32_t> sl (r.n_new);
                memcpy (d.data (), X.r2_node, pre_len); d[pre_len] = 0; if (cr.dict_len) memcpy (d.data () + pre_len + 1, f->stage.data () + p.at[0], cr.dict_len);
                ci[0] = 0; sl[0] = pre_len;
                const uint64_t *ci0 = (const uint64_t *)(f->stage.data () + p.at[1]); const uint32_t *sl0 = (const uint32_t *)(f->stage.data () + p.at[2]);
                for (uint32_t k = 0; k < cr.n_new; k++) { ci[k + 1] = ci0[k] + pre_len + 1; sl[k + 1] = sl0[k]; }
                blob_put (K.blob, d.data (), d.size ()); blob_put (K.blob, ci.data (), 8 * ci.size ()); blob_put (K.blob, sl.data (), 4 * sl.size ());
            }
            blob_put (K.blob, counts, 4 * ((size_t)Z.n_ol + r.n_new));                // (the pre-created node's count, 0, sits between the cloned words' and the new nodes')
        }
    }
    *blob_out = K.blob.data (); *blob_len_out = K.blob.size ();
    K.phase = 1;
    T.mark ("staging+blob+early"); T.done ("seg");
    return GZ_OK;
}

// QUAL and its three contexts of this process' VBlocks, once the file's QUAL codec is known (codec_domq_comp_init, codec_domq.c:299-323)
static void zip_apply_qual_mode (GzZipFile *f, int mode)
{
    ZipCall &K = f->call;
    if (K.qual_mode_applied == mode || f->qual_ctx < 0) return;
    K.qual_mode_applied = mode;
    const uint32_t NC = (uint32_t)f->ctxs.size ();
    for (uint32_t v = 0; v < K.NV; v++) {
        if (mode != GZ_CODEC_DOMQ || !K.vbs[v].n_reads) continue;          // (a plain local: as gathered; the three stay empty)
        const ZipDomq &D = K.domq[v];
        // (a VBlock whose every line is one repeated score: ctx->local.len is 0, QUAL is never compressed and codec_domq_compress - whose
        //  "no scores at all" output is the single byte 'X', codec_domq.c:490-494 - never runs: QUALMPLX has a byte per line that took part)
        const bool any = D.res.mplx_len != 0;
        const uint64_t len[4] = { any ? D.res.qual_len : 0, any ? D.res.runs_len : 0, D.res.mplx_len, any ? D.res.divr_len : 0 };
        for (int k = 0; k < 4; k++) {
            ZipCol &Z = K.col[(size_t)v * NC + (k ? f->aux[k - 1] : f->qual_ctx)];
            Z.local = D.out[k]; Z.local_len = len[k]; Z.local_cap = len[k]; Z.has_local = len[k] != 0;
            Z.ltype = k ? GZ_LT_SUPP : GZ_LT_CODEC;
        }
    }
}

// Predicted coding (no counterpart in the reference; results are the same with and without). A file's first call finds no codec for most
// contexts: the trial compressions of codec_assign_best_codec (a8) - eight candidates on a 100 KB sample each, every one of them a strictly
// serial model + chain - sit between generation and the coding of the sections, 6 - 11 ms in which the device does little else. So every
// stream that is waiting for a codec is handed to the coders right away with a PREDICTED codec, on the second handle, beside the trials;
// the trials decide as always, and a section whose prediction they confirm is only framed afterwards (like the QUAL streams coded ahead),
// the others are coded again with the rest. The prediction: what the handle's previous file ended up with for the same (dict_id, local |
// b250); a cold handle (or GZ_ZIP_PRIOR_ONLY=1) predicts nothing unless GZ_ZIP_PREDICTION=prior asks for the built-in prior by kind of stream.
static inline uint64_t zip_dict_key (const uint8_t id[8]) { uint64_t k; memcpy (&k, id, 8); return k; }
static int zip_predict_codec (const GzZipFile *f, const GzFastqCtx &X, const ZipCol &Z, bool is_local, uint32_t dyn_width, bool prior_only)
{
    if (!prior_only) {
        const auto it = f->h_user->zip_codec_memory.find (std::make_pair (zip_dict_key (X.dict_id), (int)is_local));
        if (it != f->h_user->zip_codec_memory.end ()) return zip_is_host_codec (it->second) ? 0 : it->second;
    }
    // (the built-in prior is off unless asked for, GZ_ZIP_PREDICTION=prior. Measured on the MI355X, ms per step without / with it: binned FASTQ
    //  - every prediction right - 26.9 / 24.8; BAM from text - 19 % of the sections wrong - 27.7 / 32.8: the coders working ahead slow the trials
    //  down, 9.2 -> 12.8 ms, and a batch that codes the wrong ones again takes as long as one that codes them all, the time of its slowest stream.
    //  Prediction pays when it is right: a handle that has seen a file of the kind)
    const char *pm = getenv ("GZ_ZIP_PREDICTION");
    const bool use_prior = pm && !strcmp (pm, "prior");
    if (!use_prior) return 0;
    if (!is_local) return GZ_CODEC_ARTB;                                   // word indices: an order-1 adaptive coder
    if (X.kind == GZ_FQ_QUAL) return Z.ltype == GZ_LT_CODEC ? GZ_CODEC_ARTb : GZ_CODEC_ARTB;       // what CODEC_DOMQ leaves: few symbols, packed
    if (X.kind == GZ_FQ_QUAL_AUX) return X.item == 0 ? GZ_CODEC_ARTW : X.item == 1 ? GZ_CODEC_RANB : GZ_CODEC_ARTb;
    if (Z.dyn_job >= 0) return dyn_width <= 1 ? GZ_CODEC_RANb : GZ_CODEC_ARTW;                    // integers: by byte planes when wider than one
    return GZ_CODEC_ARTB;
}

// ---------------------------------------------------------------------------------------------------------
// phase 2: the merge over ALL VBlocks of the call in vblock_i order (this process' and, when the file is dealt out over
// several processes, everybody else's), then b250 generation, locals into file order, R2 == R1 drops, and the trial
// compressions of contexts whose codec the file does not know yet -> votes
// ---------------------------------------------------------------------------------------------------------
extern "C" int gz_fastq_zip_merge (GzZipFile *f, const void *const *blobs, const uint64_t *blob_lens, int n_blobs, const void **votes_out, uint64_t *votes_len_out)
{
    if (!f || f->call.phase != 1 || n_blobs < 0 || (n_blobs && (!blobs || !blob_lens)) || !votes_out || !votes_len_out) return GZ_ERR_ARG;
    GzHandle *h = f->h;
    ZipCall &K = f->call;
    GzFastqVB *vbs = K.vbs;
    const uint32_t NC = (uint32_t)f->ctxs.size (), NV = K.NV;
    auto COL = [&] (uint32_t v, uint32_t c) -> ZipCol & { return K.col[(size_t)v * NC + c]; };
    HIPCHK (h, hipSetDevice (h->device));
    int rc;
    const uint8_t ATS = 0x20;
    ZipTimer T;

    // every VBlock of every blob, in vblock_i order
    struct Ent { uint32_t vblock_i; const uint8_t *p; };
    std::vector<Ent> ents;
    for (int b = 0; b < n_blobs; b++) {
        const uint8_t *p = (const uint8_t *)blobs[b], *end = p + blob_lens[b];
        while (p < end) {
            if ((size_t)(end - p) < sizeof (ZipBlobVB)) { h->err = "merge blob: truncated"; return GZ_ERR_CORRUPT; }
            const ZipBlobVB *hv = (const ZipBlobVB *)p;
            if (hv->n_ctx != NC) { h->err = "merge blob: made with another plan"; return GZ_ERR_CORRUPT; }
            ents.push_back ({ hv->vblock_i, p });
            p += sizeof (ZipBlobVB);
            for (uint32_t c = 0; c < NC; c++) {
                if ((size_t)(end - p) < sizeof (ZipMergeRec)) { h->err = "merge blob: truncated"; return GZ_ERR_CORRUPT; }
                const ZipMergeRec *r = (const ZipMergeRec *)p;
                p += sizeof (ZipMergeRec);
                // (lengths come from a peer's blob: sized in 64 bits and compared with what is left BEFORE the pointer moves)
                if (r->state > 3 || r->dict_len > (uint64_t)(end - p)) { h->err = "merge blob: corrupt record"; return GZ_ERR_CORRUPT; }
                uint64_t need = 0;
                if (r->state == 1) need = ((r->dict_len + 7) & ~7ull) + 8ull * r->n_new + ((4ull * r->n_new + 7) & ~7ull) + ((4ull * ((uint64_t)r->n_ol + r->n_new) + 7) & ~7ull);
                if (r->state == 3) need = (r->dict_len + 7) & ~7ull;
                if (need > (uint64_t)(end - p)) { h->err = "merge blob: truncated"; return GZ_ERR_CORRUPT; }
                if (r->state == 1 && r->n_ol > f->zctx[c]->snip_len.size ()) { h->err = "merge blob: more cloned words than the dictionary has"; return GZ_ERR_CORRUPT; }
                p += need;
            }
        }
    }
    std::stable_sort (ents.begin (), ents.end (), [] (const Ent &a, const Ent &b) { return a.vblock_i < b.vblock_i; });
    for (size_t i = 0; i < ents.size (); i++)
        if ((i && ents[i].vblock_i == ents[i - 1].vblock_i) || f->was_merged (ents[i].vblock_i)) { h->err = "merge: a vblock_i twice (in this call, or in an earlier one)"; return GZ_ERR_ARG; }
    // codec_assign_best_qual_codec (codec.c:391-450): the first VBlock of the file to get there decides for the file - in a serial
    // run VBlock 1. FASTQ has no SEQ-dependent QUAL codec, so it is DOMQ if that VBlock's lines are a fit, else a plain local
    if (f->qual_mode < 0 && !ents.empty ()) {
        const uint32_t q = ((const ZipBlobVB *)ents[0].p)->qual;
        if (!(q & 1)) { h->err = "merge blob: the first VBlock's QUAL was not tested"; return GZ_ERR_CORRUPT; }
        f->qual_mode = (q & 2) ? GZ_CODEC_DOMQ : 0;
    }
    const int qmode = f->qual_mode;
    if (qmode == GZ_CODEC_DOMQ)
        for (const Ent &e : ents) if (((const ZipBlobVB *)e.p)->qual & 4) { h->err = "QUAL: a score outside ' '..'~' (codec_domq.c:150-153)"; return GZ_ERR_CORRUPT; }
    zip_apply_qual_mode (f, qmode);
    std::map<uint32_t, uint32_t> own;                       // vblock_i -> index in vbs
    for (uint32_t v = 0; v < NV; v++) own[vbs[v].vblock_i] = v;

    K.n2w_host.clear ();
    for (const Ent &e : ents) {
        const ZipBlobVB *hv = (const ZipBlobVB *)e.p;
        const uint8_t *p = e.p + sizeof (ZipBlobVB);
        const auto it = own.find (hv->vblock_i);
        const bool mine = it != own.end ();
        const uint32_t v = mine ? it->second : 0;
        const bool is_r2 = hv->r1_vblock_i != 0;
        const ZipVBState *R1 = NULL;
        if (is_r2) {
            const auto r = K.vbstate.find (hv->r1_vblock_i);
            if (r == K.vbstate.end ()) { h->err = "merge: an R2 VBlock whose R1 VBlock is not part of the call"; return GZ_ERR_ARG; }
            R1 = &r->second;
        }
        ZipVBState &VS = K.vbstate[hv->vblock_i];
        VS.has_b250.assign (NC, 0); VS.has_local.assign (NC, 0); VS.host_b250.assign (NC, std::vector<uint8_t> ());
        for (uint32_t c = 0; c < NC; c++) {
            const GzFastqCtx &X = f->ctxs[c];
            GzZctx *z = f->zctx[c];
            const ZipMergeRec *r = (const ZipMergeRec *)p;
            p += sizeof (ZipMergeRec);
            ZipCol scratch;
            ZipCol &Z = mine ? COL (v, c) : scratch;
            const bool qual_kind = X.kind == GZ_FQ_QUAL || X.kind == GZ_FQ_QUAL_AUX;
            const uint64_t local_len = qual_kind ? (qmode == GZ_CODEC_DOMQ ? r->domq_local_len : X.kind == GZ_FQ_QUAL ? r->local_len : 0) : r->local_len;
            const uint8_t *payload3 = NULL;
            if (r->state == 3) { payload3 = p; p += (r->dict_len + 7) & ~7ull; }
            bool has_local = local_len != 0;
            if (X.kind == GZ_FQ_SEQ) has_local = mine ? Z.has_local : false;    // (NONREF_X takes no part in any pair rule)
            VS.has_local[c] = has_local;
            if (r->state == 0 || (r->state == 3 && X.kind == GZ_FQ_QUAL_AUX && qmode != GZ_CODEC_DOMQ)) continue;
            GzMergeJob m; memset (&m, 0, sizeof (m));
            m.vblock_i = hv->vblock_i;
            m.local_len = X.kind == GZ_FQ_QUAL ? r->local_len : local_len;          // (QUAL with a b250: ctx->local.len as the segmenter left it)
            m.pair2_identical = is_r2 && X.pair_identical;
            if (is_r2) { m.b250_r1_len = R1->has_b250[c]; m.local_r1_len = R1->has_local[c]; }
            if (r->state == 3 && !r->dict_len) {
                // a snip of length 0: WORD_INDEX_EMPTY - no node, a b250 of the one entry BF FE that is all-the-same and cannot be dropped
                // (ctx_drop_all_the_same: "word_index is negative", context.c:826)
                const uint32_t n_words = (uint32_t)z->snip_len.size ();
                std::vector<uint32_t> cnt ((size_t)n_words + 1, 0);
                int32_t w1 = -1; uint8_t no_ston[8]; const uint64_t one_ci = 0; const uint32_t one_sl = 0;
                m.n_ol = n_words; m.n_new = 0; m.dict = no_ston; m.node_char_index = &one_ci; m.node_snip_len = &one_sl; m.counts = cnt.data ();
                m.b250_len = 2; m.flags = X.flags | ATS; m.ats_node_index = -1; m.no_drop_b250 = 1;
                m.node2word = &w1; m.ston_local = no_ston; m.ston_cap = 0;
                if ((rc = gz_ctx_merge (z, &m)) != GZ_OK) { h->err = "gz_ctx_merge (empty snip)"; return rc < 0 ? rc : GZ_ERR; }
                VS.has_b250[c] = 1; VS.host_b250[c].assign ({ 0xBF, 0xFE });                    // WORD_INDEX_EMPTY (b250.c:29-43)
                if (mine) { Z.n_ol = n_words; Z.all_the_same = true; Z.b250_count = 1; Z.seg_b250_len = 2; Z.n_new = 0; Z.lcodec = m.lcodec; Z.bcodec = m.bcodec; Z.has_b250 = true; Z.host_b250 = VS.host_b250[c]; }
                continue;
            }
            if (r->state == 2 || r->state == 3) {
                const uint8_t *snip = r->state == 3 ? payload3 : X.snip; const uint32_t snip_len = r->state == 3 ? (uint32_t)r->dict_len : X.snip_len;
                // GZ_FQ_CONST / GZ_FQ_ITEM_DELTA: every line segs `snip` - one node, count = lines (b250_seg_append's
                // all-the-same collapse, b250.c:117-141); evaluated here, no device work. The snip is looked up in the
                // dictionary as it is NOW: a word added by an earlier VBlock of this call then counts as cloned, which
                // changes no byte (the node of a new word and the index of a cloned one convert to the same word index)
                const uint32_t found = zctx_find (z, gz_snip_mix (snip, snip_len), snip, snip_len);
                const uint32_t n_words = (uint32_t)z->snip_len.size ();
                const uint64_t seg_len = found == GZ_NO_WORD ? 4 : found <= 126 ? 1 : found <= 16508 ? 2 : found <= 2113660 ? 3 : 4;
                std::vector<uint32_t> cnt ((size_t)n_words + 1, 0);
                const int32_t node = found == GZ_NO_WORD ? (int32_t)n_words : (int32_t)found;
                cnt[(size_t)node] = r->n;
                const uint64_t one_ci = 0; const uint32_t one_sl = snip_len;
                int32_t w1 = -1; uint8_t no_ston[8];
                m.n_ol = n_words; m.n_new = found == GZ_NO_WORD; m.dict = snip; m.node_char_index = &one_ci; m.node_snip_len = &one_sl; m.counts = cnt.data ();
                m.b250_len = seg_len; m.flags = X.flags | ATS; m.ats_node_index = node;
                m.node2word = &w1; m.ston_local = no_ston; m.ston_cap = 0;
                if ((rc = gz_ctx_merge (z, &m)) != GZ_OK) { h->err = "gz_ctx_merge (constant snip)"; return rc < 0 ? rc : GZ_ERR; }
                VS.has_b250[c] = !m.dropped_b250;
                if (VS.has_b250[c]) { VS.host_b250[c].resize (4); VS.host_b250[c].resize (zip_piz_put (VS.host_b250[c].data (), found == GZ_NO_WORD ? w1 : (int64_t)found)); }
                // (b250.c:270-277) an R2 b250 identical to its R1 counterpart is dropped
                if (VS.has_b250[c] && is_r2 && X.pair_identical && R1->has_b250[c] && R1->host_b250[c] == VS.host_b250[c]) { /* kept in VS for later R2's; no section */ if (mine) Z.has_b250 = false; }
                else if (mine) Z.has_b250 = VS.has_b250[c];
                if (mine) { Z.n_ol = n_words; Z.all_the_same = true; Z.b250_count = r->n; Z.seg_b250_len = seg_len; Z.n_new = m.n_new; Z.lcodec = m.lcodec; Z.bcodec = m.bcodec;
                            if (Z.has_b250) Z.host_b250 = VS.host_b250[c]; }
                continue;
            }
            // a device column
            const uint8_t *dict = p; p += (r->dict_len + 7) & ~7ull;
            const uint64_t *nci = (const uint64_t *)p; p += 8ull * r->n_new;
            const uint32_t *nsl = (const uint32_t *)p; p += (4ull * r->n_new + 7) & ~7ull;
            const uint32_t *counts = (const uint32_t *)p; p += (4ull * ((uint64_t)r->n_ol + r->n_new) + 7) & ~7ull;
            if (mine) { Z.n_new = r->n_new; Z.all_the_same = r->all_the_same != 0; Z.seg_b250_len = r->seg_b250_len; Z.b250_count = r->b250_count; }
            // zip_handle_unique_words_ctxs (src/zip.c:136-166): a context without local whose every entry is a word new to the
            // VBlock (a unique ID) hands its whole dictionary to local; nodes and b250 are gone, nothing is merged
            if (!r->local_len && !X.no_stons && (X.flags & 3) != 3 && !r->all_the_same && r->n_new && r->n_new == r->b250_count && r->n_new >= r->n / 5 && r->b250_count != 1) {
                VS.has_local[c] = 1;
                if (mine) {
                    Z.has_b250 = false; Z.has_local = true; Z.ltype = GZ_LT_SINGLETON;
                    Z.local = K.col_jobs[Z.col_job].dict; Z.local_len = r->dict_len; Z.local_cap = Z.local_len;
                    if (Z.pre_node) { Z.ston_local.assign (dict, dict + r->dict_len); Z.host_local = true; }   // (the dictionary with the pre-created node's snip in front: from the blob)
                    GzZctxView zv; gz_zctx_view (z, &zv);
                    Z.lcodec = zv.lcodec; Z.bcodec = zv.bcodec;
                }
                continue;
            }
            // flags, singleton rule (zip_handle_unique_words_ctxs makes a context without local an LT_SINGLETON one;
            // ctx_can_have_singletons src/context.h:263-265)
            m.n_ol = r->n_ol; m.n_new = r->n_new; m.dict = dict; m.node_char_index = nci; m.node_snip_len = nsl; m.counts = counts;
            m.b250_len = r->seg_b250_len;
            m.flags = X.flags | (r->all_the_same ? ATS : 0);
            m.ats_node_index = r->ats_node;
            m.can_have_singletons = !r->local_len && !X.no_stons && (X.flags & 3) != 3 && !r->all_the_same;
            if (r->all_the_same && r->ats_node < 0) m.no_drop_b250 = 1;
            std::vector<int32_t> n2w_foreign; std::vector<uint8_t> ston_foreign;
            if (mine) {
                Z.n2w_at = K.n2w_host.size ();
                K.n2w_host.resize (K.n2w_host.size () + r->n_new + 1);
                m.node2word = K.n2w_host.data () + Z.n2w_at;
                Z.ston_local.resize ((size_t)r->dict_len + 8);
                m.ston_local = Z.ston_local.data (); m.ston_cap = Z.ston_local.size ();
            }
            else {
                n2w_foreign.resize ((size_t)r->n_new + 1); ston_foreign.resize ((size_t)r->dict_len + 8);
                m.node2word = n2w_foreign.data (); m.ston_local = ston_foreign.data (); m.ston_cap = ston_foreign.size ();
            }
            if ((rc = gz_ctx_merge (z, &m)) != GZ_OK) { h->err = "gz_ctx_merge"; return rc < 0 ? rc : GZ_ERR; }
            VS.has_b250[c] = !m.dropped_b250 && r->seg_b250_len != 0;
            if (m.ston_len) VS.has_local[c] = 1;
            if (mine) {
                Z.ston_local.resize (m.ston_len);
                Z.lcodec = m.lcodec; Z.bcodec = m.bcodec;
                Z.has_b250 = VS.has_b250[c];
                if (m.ston_len) { Z.has_local = true; Z.ston_only_local = true; Z.ltype = GZ_LT_SINGLETON; Z.local_len = m.ston_len; }
            }
        }
    }
    // locals of contexts without a b250 merge still inherit the committed codec (context.c:980-981)
    for (uint32_t c = 0; c < NC; c++) {
        GzZctxView zv; gz_zctx_view (f->zctx[c], &zv);
        for (uint32_t v = 0; v < NV; v++) { ZipCol &Z = COL (v, c); if (!Z.lcodec) Z.lcodec = zv.lcodec; if (!Z.bcodec) Z.bcodec = zv.bcodec; }
    }
    for (size_t i = 0; i < ents.size (); i++) f->add_merged (ents[i].vblock_i);

    T.mark ("merge");
    // ---- b250 generation, locals into file order, R2 == R1 drops --------------------------------------------------------------
    int32_t *d_n2w = (int32_t *)ws_alloc (f, (K.n2w_host.size () + 1) * 4);
    if (!d_n2w) return GZ_ERR_HIP;
    if (!K.n2w_host.empty ()) HIPCHK (h, hipMemcpyAsync (d_n2w, K.n2w_host.data (), K.n2w_host.size () * 4, hipMemcpyHostToDevice, h->stream));
    // small host-made payloads (constant b250s, singletons) go up in one copy
    std::vector<uint8_t> small; std::vector<std::pair<ZipCol *, std::pair<int, size_t>>> small_ref;
    for (auto &Z : K.col) {
        if (Z.has_b250 && !Z.host_b250.empty ()) { small_ref.push_back ({ &Z, { 0, small.size () } }); small.insert (small.end (), Z.host_b250.begin (), Z.host_b250.end ()); small.resize ((small.size () + 15) & ~(size_t)15); }
        if ((Z.ston_only_local || Z.host_local) && !Z.ston_local.empty ()) { small_ref.push_back ({ &Z, { 1, small.size () } }); small.insert (small.end (), Z.ston_local.begin (), Z.ston_local.end ()); small.resize ((small.size () + 15) & ~(size_t)15); }
    }
    uint8_t *d_small = (uint8_t *)ws_alloc (f, small.size () + 16);
    if (!d_small) return GZ_ERR_HIP;
    if (!small.empty ()) {
        f->stage.assign (small.begin (), small.end ());                     // (lives until the copy has run)
        HIPCHK (h, hipMemcpyAsync (d_small, f->stage.data (), small.size (), hipMemcpyHostToDevice, h->stream));
    }
    for (auto &sr : small_ref) {
        if (sr.second.first == 0) { sr.first->sec_b250 = d_small + sr.second.second; sr.first->sec_b250_len = (uint32_t)sr.first->host_b250.size (); }
        else sr.first->local = d_small + sr.second.second;
    }

    std::vector<GzB250Job> bjobs; std::vector<GzLocalJob> ljobs; std::vector<GzdSameJob> same;
    for (uint32_t v = 0; v < NV; v++)
        for (uint32_t c = 0; c < NC; c++) {
            ZipCol &Z = COL (v, c);
            const GzFastqCtx &X = f->ctxs[c];
            const bool is_r2 = vbs[v].r1 >= 0;
            if (Z.has_b250 && Z.col_job >= 0) {
                GzB250Job j; memset (&j, 0, sizeof (j));
                j.seg = Z.b250_seg; j.seg_len = (uint32_t)Z.seg_b250_len; j.ol_nodes_len = Z.n_ol;
                j.node2word = d_n2w + Z.n2w_at; j.n_new_nodes = Z.n_new;
                if (!(Z.b250_out = (uint8_t *)ws_alloc (f, Z.seg_b250_len + 16))) return GZ_ERR_HIP;
                j.out = Z.b250_out; j.out_len_dev = Z.sec_len_dev + 1; j.status_dev = K.d_b250st + ((size_t)v * NC + c);
                if (is_r2 && X.pair_identical) {
                    const ZipCol &R1 = COL ((uint32_t)vbs[v].r1, c);
                    if (R1.has_b250 && R1.b250_out) { j.r1 = R1.b250_out; j.r1_len_dev = R1.sec_len_dev + 1; }
                }
                Z.sec_b250 = Z.b250_out; Z.sec_b250_len = (uint32_t)Z.seg_b250_len;
                bjobs.push_back (j);
            }
            if (Z.has_local && Z.dyn_job >= 0 && X.transposed && X.per_sample && f->plan.n_samples) {
                // zip_generate_local for a dyn_transposed context (zip.c:185-219: byte order first, then dyn_int_transpose): the unsigned
                // lines x samples matrix goes out samples x lines as LT_UINTn_TR; a column that turned out signed (or with entries that are
                // not integers) stays as it is, like in the reference (dyn_int.c:52-56)
                const int lt = Z.ltype;
                const uint32_t w = lt == GZ_LT_UINT8 ? 1 : lt == GZ_LT_UINT16 ? 2 : lt == GZ_LT_UINT32 ? 4 : 0;
                if (w && Z.local_len == (uint64_t)Z.n * w) {
                    void *scratch = ws_alloc (f, Z.local_len + 64);
                    if (!scratch) return GZ_ERR_HIP;
                    const int r = gz_local_generate (h, lt, Z.local, Z.n, f->plan.n_samples, scratch);
                    if (r < 0) return r;
                    Z.ltype = r; Z.host_len = true;
                    continue;
                }
            }
            if (Z.has_local && Z.dyn_job >= 0) {
                GzLocalJob j; memset (&j, 0, sizeof (j));
                j.data = Z.local; j.n = Z.n; j.dyn_dev = K.d_dynres + Z.dyn_job; j.len_dev = Z.sec_len_dev;
                ljobs.push_back (j);
                if (is_r2 && X.pair_identical) {
                    const ZipCol &R1 = COL ((uint32_t)vbs[v].r1, c);
                    if (R1.has_local && R1.dyn_job >= 0 && R1.ltype == Z.ltype) {
                        GzdSameJob s; s.a = Z.local; s.a_len = Z.sec_len_dev; s.b = R1.local; s.b_len = R1.sec_len_dev; s.drop_len = Z.sec_len_dev; s.flag = NULL;
                        same.push_back (s);
                    }
                }
            }
        }
    if (!bjobs.empty ()) for (size_t at = 0; at < bjobs.size (); at += 32768) ZCHK (gz_b250_generate_batch (h, bjobs.data () + at, (int)std::min<size_t> (32768, bjobs.size () - at)));
    ZCHK (gz_local_generate_batch (h, ljobs.data (), (int)ljobs.size ()));
    if (!same.empty ()) {
        void *ds;
        if ((rc = upload (h, same.data (), same.size () * sizeof (GzdSameJob), &ds)) != GZ_OK) return rc;
        KLAUNCH (h, k_bufs_identical, dim3 ((uint32_t)same.size ()), dim3 (256), 0, (const GzdSameJob *)ds);
    }

    T.mark ("generate-queue");
    // ---- a8: contexts whose codec the file does not know yet: trial compressions on the first VBlock (of this process) that
    // has >= 50 bytes of the stream (codec.c:309-312); the lowest vblock_i of all processes' votes is committed in phase 3
    {
        std::vector<const uint8_t *> ptr; std::vector<uint32_t> len; std::vector<ZipVote> who;
        bool need = false;
        for (uint32_t c = 0; c < NC; c++) { if (f->ctxs[c].kind == GZ_FQ_ITEM_EXPECT) continue; GzZctxView zv; gz_zctx_view (f->zctx[c], &zv); if (!zv.lcodec || !zv.bcodec) need = true; }
        if (f->plan.vb_1_not_representative) for (uint32_t v = 0; v < NV; v++) if (vbs[v].vblock_i == ZIP_RETEST_VB_I) need = true;
        if (need || K.spec_pending) {
            std::vector<uint32_t> seclen (2 * (size_t)NV * NC);
            HIPCHK (h, hipMemcpyAsync (seclen.data (), K.d_seclen, seclen.size () * 4, hipMemcpyDeviceToHost, h->stream));
            if ((rc = gz_sync (h)) < 0) return rc;
            T.mark ("generate-sync");
            if (K.spec_pending) {                                         // the trial behind a speculation (seg phase): what does the file itself say?
                K.spec_pending = false;
                const uint32_t c = (uint32_t)f->qual_ctx;
                const uint32_t sample = (uint32_t)std::min<uint64_t> (COL (0, c).local_len, 99999);
                uint32_t best_size = sample; int codec = GZ_CODEC_NONE;                         // NONE: the bare length (codec.c:324)
                for (int k = 0; k < 8; k++) {
                    const GzStream &st = K.spec_trial[8 * K.spec_t + k];
                    if (st.status != GZ_OK) { h->err = "trial compression failed"; if (f->h2) (void)gz_sync (f->h2); return GZ_ERR; }
                    if (st.out_len + 28 < best_size) { best_size = st.out_len + 28; codec = st.codec; }   // framed (codec.c:328-331), ties -> first
                }
                K.votes.push_back ({ c, 1, vbs[0].vblock_i, (uint32_t)codec });
                if (K.spec && codec == K.spec_codec) f->h_user->zip_spec_hits++;
                else if (K.spec) {                                        // refuted: let the streams run out, then code them with the rest
                    f->h_user->zip_spec_misses++;
                    (void)gz_sync (f->h2);
                    for (uint32_t v = 0; v < NV; v++) { ZipCol &Z = COL (v, c); if (Z.early >= 0) { Z.early = -1; Z.lcodec = 0; } }
                    K.early.clear (); K.spec = false;
                }
            }
            for (uint32_t c = 0; c < NC; c++) {
                GzZctxView zv; gz_zctx_view (f->zctx[c], &zv);
                for (uint32_t is_local = 0; is_local < 2; is_local++) {
                    if (is_local ? zv.lcodec : zv.bcodec) continue;
                    bool voted = false;
                    for (const ZipVote &vt : K.votes) if (vt.ctx == c && vt.is_local == is_local) voted = true;
                    if (voted) continue;                                  // (decided above for the streams that were coded ahead)
                    for (uint32_t v = 0; v < NV; v++) {
                        ZipCol &Z = COL (v, c);
                        uint32_t L = 0; const uint8_t *p = NULL;
                        if (is_local && Z.has_local) { p = Z.local; L = Z.dyn_job >= 0 && !Z.host_len ? seclen[2 * ((size_t)v * NC + c)] : (uint32_t)Z.local_len; }
                        if (!is_local && Z.has_b250) { p = Z.sec_b250; L = Z.col_job >= 0 ? seclen[2 * ((size_t)v * NC + c) + 1] : Z.sec_b250_len; }
                        // a VBlock too small to speak for the file keeps its choice to itself, and the next one tests again (codec.c:352); so
                        // does VBlock 1 for the local of a context whose beginning may not be representative (:358-362)
                        const int rule = gz_codec_assign_rule (vbs[v].vblock_i, vbs[v].text_len, f->plan.vb_size, (vbs[v].flags & GZ_VB_LAST_OF_FILE) != 0, (int)is_local,
                                                               zip_not_representative (f->plan, f->ctxs[c]), 0, 0, L);
                        if (!(rule & 1)) continue;
                        const bool commits = (rule & 2) != 0;
                        ptr.push_back (p); len.push_back (L); who.push_back ({ c, is_local | (commits ? 0u : 2u), vbs[v].vblock_i, 0 });
                        if (commits) break;
                    }
                }
            }
            // RETEST_VB_I (codec.c:274-277): VBlock 10 looks again at the contexts whose first VBlocks may not have been representative
            if (f->plan.vb_1_not_representative)
                for (uint32_t v = 0; v < NV; v++) if (vbs[v].vblock_i == ZIP_RETEST_VB_I)
                    for (uint32_t c = 0; c < NC; c++) if (zip_not_representative (f->plan, f->ctxs[c]))
                        for (uint32_t is_local = 0; is_local < 2; is_local++) {
                            if (is_local ? f->ctxs[c].lcodec : f->ctxs[c].bcodec) continue;              // (hard-coded)
                            bool tested = false;                                                         // (nothing in the file yet: tested above)
                            for (const ZipVote &w : who) if (w.ctx == c && (w.is_local & 1) == is_local && w.vblock_i == ZIP_RETEST_VB_I) tested = true;
                            for (const ZipVote &w : K.votes) if (w.ctx == c && (w.is_local & 1) == is_local && w.vblock_i == ZIP_RETEST_VB_I) tested = true;
                            if (tested) continue;
                            ZipCol &Z = COL (v, c);
                            uint32_t L = 0; const uint8_t *p = NULL;
                            if (is_local && Z.has_local) { p = Z.local; L = Z.dyn_job >= 0 && !Z.host_len ? seclen[2 * ((size_t)v * NC + c)] : (uint32_t)Z.local_len; }
                            if (!is_local && Z.has_b250) { p = Z.sec_b250; L = Z.col_job >= 0 ? seclen[2 * ((size_t)v * NC + c) + 1] : Z.sec_b250_len; }
                            const int rule = gz_codec_assign_rule (vbs[v].vblock_i, vbs[v].text_len, f->plan.vb_size, (vbs[v].flags & GZ_VB_LAST_OF_FILE) != 0, (int)is_local,
                                                                   1, 0, GZ_CODEC_RANB /* whatever the file has */, L);
                            if (!(rule & 1)) continue;                                                   // (too short to test: :309-312, the file's codec stays)
                            ptr.push_back (p); len.push_back (L); who.push_back ({ c, is_local | 4u | ((rule & 2) ? 0u : 2u), vbs[v].vblock_i, 0 });
                        }
            // predicted coding: every stream that waits for one of these trials goes to the coders now, on the second handle (idle: no long
            // streams were coded ahead in this call), with a predicted codec - see zip_predict_codec
            if (f->h2 && K.early.empty () && !f->hostc.trial && !who.empty () && !getenv ("GZ_ZIP_NO_PREDICTION")) {
                const char *pe = getenv ("GZ_ZIP_PRIOR_ONLY");
                const bool prior_only = pe && *pe && *pe != '0';
                std::vector<std::pair<ZipCol *, int>> owner;           // (column, is_local) of every stream
                for (uint32_t c = 0; c < NC; c++) {
                    const GzFastqCtx &X = f->ctxs[c];
                    if (X.kind == GZ_FQ_SEQ || X.kind == GZ_FQ_ITEM_EXPECT) continue;
                    for (uint32_t is_local = 0; is_local < 2; is_local++) {
                        bool waiting = false;
                        for (const ZipVote &w : who) if (w.ctx == c && (w.is_local & 1) == is_local) waiting = true;
                        if (!waiting) continue;
                        for (uint32_t v = 0; v < NV; v++) {
                            ZipCol &Z = COL (v, c);
                            if (is_local ? (Z.lcodec != 0 || Z.early >= 0) : Z.bcodec != 0) continue;
                            uint32_t L = 0; const uint8_t *p = NULL;
                            if (is_local && Z.has_local) { p = Z.local; L = Z.dyn_job >= 0 && !Z.host_len ? seclen[2 * ((size_t)v * NC + c)] : (uint32_t)Z.local_len; }
                            if (!is_local && Z.has_b250) { p = Z.sec_b250; L = Z.col_job >= 0 ? seclen[2 * ((size_t)v * NC + c) + 1] : Z.sec_b250_len; }
                            if (L < 50 || !p) continue;                   // (stored, or dropped on the device)
                            const int codec = zip_predict_codec (f, X, Z, is_local != 0, Z.dyn_job >= 0 ? K.dynres[Z.dyn_job].width : 0, prior_only);
                            if (!codec) continue;
                            GzStream st; memset (&st, 0, sizeof (st));
                            st.in = p; st.in_len = L; st.codec = codec; st.out_cap = gz_codec_est_size (codec, L);
                            if (!(st.out = (uint8_t *)ws_alloc (f, (size_t)st.out_cap + 64))) return GZ_ERR_HIP;
                            K.early.push_back (st); owner.push_back ({ &Z, (int)is_local });
                        }
                    }
                }
                if (!K.early.empty ()) {
                    if (!(K.d_early_len = (uint32_t *)ws_alloc (f, K.early.size () * 4))) return GZ_ERR_HIP;
                    for (size_t k = 0; k < K.early.size (); k++) { K.early[k].out_len_dev = K.d_early_len + k; (owner[k].second ? owner[k].first->early : owner[k].first->early_b) = (int)k; }
                    // (their inputs are complete: this handle was synchronised above, behind the generation kernels; the second handle may still hold
                    //  the seg phase's QUAL trial that nobody waited for: one batch at a time per handle)
                    if ((rc = gz_sync (f->h2)) < 0) { h->err = f->h2->err; return rc; }
                    if ((rc = gz_codec_compress_batch (f->h2, K.early.data (), (int)K.early.size ())) != GZ_OK) { h->err = f->h2->err; (void)gz_sync (f->h2); return rc; }
                    K.predicted = true;
                }
            }
            std::vector<int> best;
            if ((rc = zip_assign_best_many (h, f, ptr, len, who, best)) != GZ_OK) { if (K.predicted) (void)gz_sync (f->h2); return rc; }
            for (size_t k = 0; k < who.size (); k++) if (best[k]) { who[k].codec = (uint32_t)best[k]; K.votes.push_back (who[k]); }
        }
    }
    *votes_out = K.votes.data (); *votes_len_out = K.votes.size () * sizeof (ZipVote);
    K.phase = 2;
    T.mark ("assign"); T.done ("merge");
    return GZ_OK;
}

// ---------------------------------------------------------------------------------------------------------
// phase 3: commit the codecs (lowest vblock_i wins, as in a serial run: codec.c:352-363), then the sections of this process'
// VBlocks in the reference's order (a15), compressed and framed (a9-a13, a16)
// ---------------------------------------------------------------------------------------------------------
static int zip_finish_launch (GzZipFile *f, const void *const *votes, const uint64_t *votes_lens, int n_votes);
static int zip_finish_wait (GzZipFile *f);
extern "C" int gz_fastq_zip_finish (GzZipFile *f, const void *const *votes, const uint64_t *votes_lens, int n_votes)
{
    const int rc = zip_finish_launch (f, votes, votes_lens, n_votes);
    if (rc != GZ_OK) return rc;
    return zip_finish_wait (f);
}

static int zip_finish_launch (GzZipFile *f, const void *const *votes, const uint64_t *votes_lens, int n_votes)
{
    if (!f || f->call.phase != 2 || n_votes < 0 || (n_votes && (!votes || !votes_lens))) return GZ_ERR_ARG;
    GzHandle *h = f->h;
    ZipCall &K = f->call;
    GzFastqVB *vbs = K.vbs;
    const uint32_t NC = (uint32_t)f->ctxs.size (), NV = K.NV;
    auto COL = [&] (uint32_t v, uint32_t c) -> ZipCol & { return K.col[(size_t)v * NC + c]; };
    HIPCHK (h, hipSetDevice (h->device));
    const uint8_t ATS = 0x20, PAIRED = 0x04;
    int rc;
    ZipTimer T;
    {
        std::map<std::pair<uint32_t, uint32_t>, ZipVote> win;
        for (int b = 0; b < n_votes; b++) {
            if (votes_lens[b] % sizeof (ZipVote)) { h->err = "votes: size"; return GZ_ERR_ARG; }
            const ZipVote *vt = (const ZipVote *)votes[b];
            for (size_t k = 0; k < votes_lens[b] / sizeof (ZipVote); k++) {
                if (vt[k].ctx >= NC || vt[k].is_local > 7) { h->err = "votes: context"; return GZ_ERR_ARG; }
                if (vt[k].is_local & 6) continue;                         // (a VBlock's own choice, VBlock 10's second look: below)
                auto key = std::make_pair (vt[k].ctx, vt[k].is_local);
                auto it = win.find (key);
                if (it == win.end () || vt[k].vblock_i < it->second.vblock_i) win[key] = vt[k];
            }
        }
        for (auto &w : win) {
            GzZctxView zv; gz_zctx_view (f->zctx[w.first.first], &zv);
            if (w.first.second ? zv.lcodec : zv.bcodec) continue;
            gz_zctx_commit_codec (f->zctx[w.first.first], (int)w.first.second, (int)w.second.codec);
            f->h_user->zip_codec_memory[std::make_pair (zip_dict_key (f->ctxs[w.first.first].dict_id), (int)w.first.second)] = (int)w.second.codec;   // (the next file's prediction)
            // (next file: speculation - only a coder the device can run ahead: a host codec's win, BZ2 / LZMA / BSC, is no guess for a file without them)
            if ((int)w.first.first == f->qual_ctx && w.first.second && !zip_is_host_codec ((int)w.second.codec)) f->h_user->zip_qual_guess[f->qual_mode == GZ_CODEC_DOMQ] = (int)w.second.codec;
            // (a VBlock in front of the one that assigned finds nothing in the file, as in a serial run: it had < 50 bytes, or is small)
            for (uint32_t v = 0; v < NV; v++) if (vbs[v].vblock_i >= w.second.vblock_i) {
                ZipCol &Z = COL (v, w.first.first); if (w.first.second) { if (!Z.lcodec) Z.lcodec = (uint8_t)w.second.codec; } else if (!Z.bcodec) Z.bcodec = (uint8_t)w.second.codec; }
        }
        for (int b = 0; b < n_votes; b++) {
            const ZipVote *vt = (const ZipVote *)votes[b];
            for (size_t k = 0; k < votes_lens[b] / sizeof (ZipVote); k++) if ((vt[k].is_local & 6) == 2)
                for (uint32_t v = 0; v < NV; v++) if (vbs[v].vblock_i == vt[k].vblock_i) {
                    ZipCol &Z = COL (v, vt[k].ctx); if (vt[k].is_local & 1) { if (!Z.lcodec) Z.lcodec = (uint8_t)vt[k].codec; } else if (!Z.bcodec) Z.bcodec = (uint8_t)vt[k].codec; }
        }
        // VBlock 10's second look replaces whatever the file had, for itself and - unless it is a small VBlock - for everything behind it
        for (int b = 0; b < n_votes; b++) {
            const ZipVote *vt = (const ZipVote *)votes[b];
            for (size_t k = 0; k < votes_lens[b] / sizeof (ZipVote); k++) if (vt[k].is_local & 4) {
                const bool sets = !(vt[k].is_local & 2), loc = vt[k].is_local & 1;
                if (sets) gz_zctx_commit_codec (f->zctx[vt[k].ctx], (int)loc, (int)vt[k].codec);
                for (uint32_t v = 0; v < NV; v++) if (sets ? vbs[v].vblock_i >= vt[k].vblock_i : vbs[v].vblock_i == vt[k].vblock_i) {
                    ZipCol &Z = COL (v, vt[k].ctx); (loc ? Z.lcodec : Z.bcodec) = (uint8_t)vt[k].codec; }
            }
        }
    }
    K.V.assign (NV, GzVBlock ()); K.secs.assign (NV, std::vector<GzSection> ());
    std::vector<GzVBlock> &V = K.V;
    std::vector<std::vector<GzSection>> &secs = K.secs;
    for (uint32_t v = 0; v < NV; v++) {
        const bool is_r2 = vbs[v].r1 >= 0, is_r1 = f->plan.paired && !is_r2;
        std::vector<GzSecOrderIn> in (NC + 1);
        for (uint32_t c = 0; c < NC; c++) {
            const ZipCol &Z = COL (v, c);
            in[c].did_i = f->ctxs[c].did_i; in[c].local_dep = f->ctxs[c].local_dep;
            if ((int)c == f->qual_ctx && f->qual_mode == GZ_CODEC_DOMQ) in[c].local_dep = 1;           // DEP_L1 (codec_domq.c:310)
            in[c].has_local = Z.has_local; in[c].ston_only_local = Z.ston_only_local; in[c].has_b250 = Z.has_b250;
            // NONREF itself - the context in front of NONREF_X (sam.h:77-79), DEP_L0 - leaves the path 2-bit packed for the host's
            // sub-codec; where its local section belongs among the VBlock's sections is reported (GzFastqVB.seq_section_index)
            if (f->ctxs[c].kind == GZ_FQ_SEQ) { memset (&in[NC], 0, sizeof (in[NC])); in[NC].did_i = f->ctxs[c].did_i - 1; in[NC].has_local = vbs[v].n_bases != 0; }
        }
        std::vector<uint32_t> order (2 * NC + 2);
        const uint32_t ns = gz_section_order (in.data (), NC + 1, vbs[v].vblock_i, order.data ());
        uint32_t mark = 0;
        for (uint32_t k = 0; k < ns; k++) {
            const uint32_t c = order[k] / 2; const bool is_b250 = order[k] & 1;
            if (c == NC) { mark = (uint32_t)secs[v].size (); continue; }
            const ZipCol &Z = COL (v, c);
            const GzFastqCtx &X = f->ctxs[c];
            GzSection s; memset (&s, 0, sizeof (s));
            memcpy (s.dict_id, X.dict_id, 8);
            s.flags = X.flags | (Z.all_the_same ? ATS : 0);
            if (is_b250) {
                s.section_type = GZ_SEC_B250; s.data = Z.sec_b250; s.data_len = Z.sec_b250_len;
                if (Z.col_job >= 0) s.data_len_dev = Z.sec_len_dev + 1;
                s.codec = Z.bcodec; s.b250_size_or_nothing_char = 4;                                   // B250_VARL
                if ((is_r1 && X.pair_identical) || (is_r2 && X.pair_assisted_b250)) s.flags |= PAIRED;    // zfile.c:292-294
                if (Z.early_b >= 0) {                                                                  // coded ahead with a predicted codec: confirmed?
                    const GzStream &es = K.early[Z.early_b];
                    if (es.codec == (s.codec ? s.codec : GZ_CODEC_RANB)) { s.precompressed = 1; s.raw_len = es.in_len; s.data = es.out; s.data_len = es.out_cap; s.data_len_dev = es.out_len_dev; f->h_user->zip_pred_hits++; }
                    else f->h_user->zip_pred_misses++;
                }
            }
            else {
                s.section_type = GZ_SEC_LOCAL; s.data = Z.local; s.data_len = (uint32_t)Z.local_len;
                if (Z.dyn_job >= 0 && !Z.host_len) { s.data_len = (uint32_t)Z.local_cap; s.data_len_dev = Z.sec_len_dev; }
                s.codec = Z.lcodec; s.ltype = (uint8_t)Z.ltype;
                if (Z.ltype == GZ_LT_CODEC) {                                                          // QUAL through CODEC_DOMQ (codec_domq.c:221,308-309,487-500)
                    // the stream's coder: the file's, as VBlock v would find it in a serial run (codec.c:280-281) - however short the
                    // stream; none known yet -> NONE; all lines diverse -> the single byte 'X', NONE (codec_domq.c:490-500)
                    s.hdr_codec = GZ_CODEC_DOMQ; s.param = (uint8_t)(K.domq[v].res.num_norm_qs | 0x80);
                    if (K.domq[v].res.all_diverse || !s.codec) s.codec = GZ_CODEC_NONE;
                }
                if (X.kind == GZ_FQ_SEQ) {
                    // NONREF_X: lcodec CODEC_XCGT names the section, the stream is coded by what codec_assign_best_codec gave (the file's; NONE
                    // while a sample under 50 bytes assigns none) and that goes to sub_codec (codec_acgt.c:142-153, USE_SUBCODEC: codec.h:109)
                    s.hdr_codec = GZ_CODEC_XCGT;
                    if (!s.codec) s.codec = GZ_CODEC_NONE;
                }
                const bool int_lt = (Z.ltype >= GZ_LT_INT8 && Z.ltype <= GZ_LT_UINT64) || (Z.ltype >= GZ_LT_UINT8_TR && Z.ltype <= GZ_LT_UINT32_TR);   // lt_max (ltype) != 0
                if (int_lt) s.b250_size_or_nothing_char = X.nothing_char ? X.nothing_char : 0xff;    // zfile.c:344-345
                if (is_r1 && X.pair_identical) s.flags |= PAIRED;                                     // zfile.c:323-325
                if (Z.early >= 0) {                                                                    // coded ahead on the second handle: only framed here
                    const GzStream &es = K.early[Z.early];
                    // (predicted coding: only if the trials ended up with the codec the stream was coded with; the QUAL streams coded ahead in the
                    //  seg phase were given the codec that is theirs by then)
                    if (!K.predicted || es.codec == (s.codec ? s.codec : GZ_CODEC_RANB)) {
                        s.precompressed = 1; s.raw_len = es.in_len; s.data = es.out; s.data_len = es.out_cap; s.data_len_dev = es.out_len_dev;
                        if (K.predicted) f->h_user->zip_pred_hits++;
                    }
                    else f->h_user->zip_pred_misses++;
                }
            }
            if (zip_is_host_codec (s.codec) && !s.precompressed) {
                // a codec of the host's (a8: its candidate won the context): the stream goes to the host, its coder's payload comes back
                // and is framed with the rest. Under 50 bytes a simple codec's section is stored (compressor.c:56-58)
                if (!f->hostc.compress) { h->err = "a context has a host codec (BZ2 / LZMA / BSC) and the file no host coder: gz_zip_set_host_codecs"; return GZ_ERR_ARG; }
                uint32_t L = s.data_len;
                if (s.data_len_dev) { HIPCHK (h, hipMemcpyAsync (&L, s.data_len_dev, 4, hipMemcpyDeviceToHost, h->stream)); HIPCHK (h, hipStreamSynchronize (h->stream)); }
                if (s.data_len_dev && !L) s.codec = GZ_CODEC_NONE;           // generated on the device and dropped there (an R2 b250 identical to R1's): the writer leaves it out
                else { s.data_len = L; s.data_len_dev = NULL; }
                if (s.codec == GZ_CODEC_NONE) ;
                else if (L < 50 && !s.hdr_codec) s.codec = GZ_CODEC_NONE;
                else {
                    std::vector<uint8_t> raw (L), pay ((size_t)L + L / 2 + 65536);
                    HIPCHK (h, hipMemcpyAsync (raw.data (), s.data, L, hipMemcpyDeviceToHost, h->stream)); HIPCHK (h, hipStreamSynchronize (h->stream));
                    uint32_t pl = (uint32_t)pay.size ();
                    if (f->hostc.compress (f->hostc.user, s.codec, raw.data (), L, pay.data (), &pl) != 0 || pl > pay.size ()) { h->err = "host codec: compress failed"; return GZ_ERR; }
                    uint8_t *d = (uint8_t *)ws_alloc (f, (size_t)pl + 64);
                    if (!d) return GZ_ERR_HIP;
                    HIPCHK (h, hipMemcpyAsync (d, pay.data (), pl, hipMemcpyHostToDevice, h->stream)); HIPCHK (h, hipStreamSynchronize (h->stream));
                    s.precompressed = 1; s.raw_len = L; s.data = d; s.data_len = pl;
                }
            }
            secs[v].push_back (s);
        }
        GzVBlock &B = V[v]; memset (&B, 0, sizeof (B));
        B.vblock_i = vbs[v].vblock_i; B.recon_size = (uint32_t)vbs[v].text_len; B.longest_line_len = K.vbstat[2 * v]; B.longest_seq_len = K.vbstat[2 * v + 1];
        B.sections = secs[v].data (); B.n_sections = (uint32_t)secs[v].size (); B.mark_section = mark;
        B.z_cap = gz_vb_z_bound (B.sections, B.n_sections);
        if (!(B.z_data = (uint8_t *)ws_alloc (f, B.z_cap + 64))) return GZ_ERR_HIP;
    }
    T.mark ("sections");
    if (!K.early.empty ()) ZCHK (gz_emit_after (h, f->h2));
    ZCHK (gz_vb_compress_batch (h, V.data (), (int)NV));
    T.mark ("compress-queue");
    K.b250st.assign ((size_t)NV * NC, 1);
    // (into page-locked memory - the seg phase's read-backs in it have been consumed: a copy into pageable memory would hold the host until
    //  the stream gets there, i.e. until this call's coders are through, and with it the next call of gz_fastq_zip_begin)
    if (!(K.b250st_pinned = (int32_t *)zip_pinned (f, K.b250st.size () * 4 + 64))) return GZ_ERR_HIP;
    HIPCHK (h, hipMemcpyAsync (K.b250st_pinned, K.d_b250st, K.b250st.size () * 4, hipMemcpyDeviceToHost, h->stream));
    K.phase = 3;
    T.done ("finish (queued)");
    (void)rc;
    return GZ_OK;
}

// the rest of phase 3: wait for the coders, results into the caller's VBlock table
static int zip_finish_wait (GzZipFile *f)
{
    if (!f || f->call.phase != 3) return GZ_ERR_ARG;
    GzHandle *h = f->h;
    ZipCall &K = f->call;
    GzFastqVB *vbs = K.vbs;
    const uint32_t NC = (uint32_t)f->ctxs.size (), NV = K.NV;
    auto COL = [&] (uint32_t v, uint32_t c) -> ZipCol & { return K.col[(size_t)v * NC + c]; };
    std::vector<GzVBlock> &V = K.V;
    std::vector<int32_t> &b250st = K.b250st;
    ZipTimer T;
    int rc = gz_sync (h);
    if (K.b250st_pinned && !b250st.empty ()) memcpy (b250st.data (), K.b250st_pinned, b250st.size () * 4);
    K.phase = 0;
    if (!K.early.empty ()) {
        const uint32_t fb = f->h2->chain_fallbacks;
        const int rc2 = gz_sync (f->h2);
        if (rc2 < 0) { h->err = f->h2->err; return rc2; }
        for (const GzStream &es : K.early) if (es.status != GZ_OK) { h->err = "a stream coded ahead failed"; return GZ_ERR; }
        if (f->h2->chain_fallbacks != fb) {
            // the streams coded ahead came out right only at the second attempt (gz_sync's fallback), AFTER the section writer of this handle
            // had framed what the first one left: the sections are written again from the payloads as they are now
            if ((rc = gz_vb_compress_batch (h, V.data (), (int)NV)) == GZ_OK) rc = gz_sync (h);
            h->err = f->h2->err;                                         // (the warning)
        }
    }
    T.mark ("compress-sync"); T.done ("finish");
    if (rc < 0) return rc;
    for (uint32_t v = 0; v < NV; v++) {
        for (uint32_t c = 0; c < NC; c++) { const ZipCol &Z = COL (v, c); if (Z.has_b250 && Z.col_job >= 0 && b250st[(size_t)v * NC + c] == -5) { h->err = "b250 generation: malformed stream"; return GZ_ERR; } }
        vbs[v].status = V[v].status; vbs[v].z_data = V[v].z_data; vbs[v].z_len = V[v].z_len; vbs[v].n_sections = V[v].n_sections;
        vbs[v].seq_section_index = V[v].mark_index;
        if (V[v].status != GZ_OK) rc = GZ_ERR;
    }
    return rc == GZ_ERR ? GZ_ERR : GZ_OK;
}

// all three phases in one process
extern "C" int gz_fastq_zip_vblocks (GzZipFile *f, uint8_t *text, uint64_t text_len, GzFastqVB *vbs, int n_vbs)
{
    const void *blob, *votes; uint64_t blob_len, votes_len;
    int rc;
    if ((rc = gz_fastq_zip_seg (f, text, text_len, vbs, n_vbs, &blob, &blob_len)) != GZ_OK) return rc;
    if ((rc = gz_fastq_zip_merge (f, &blob, &blob_len, 1, &votes, &votes_len)) != GZ_OK) return rc;
    return gz_fastq_zip_finish (f, &votes, &votes_len, 1);
}

// ---- two calls in flight ---------------------------------------------------------------------------------------------------------
// A stream of calls on one file (BASELINE configs[4]): the next call's seg and merge phases - and its coders - run while the
// previous call's long chains are still at work. What the reference gets from its pool of compute threads (VBlocks of different
// ages in flight, merging in order) is here two sets of handles / workspace taking turns. The order of the merges, and with it
// every dictionary and every byte written, is the same as with one call at a time.
static int zip_make_other_lane (GzZipFile *f)
{
    if (f->other_made) return GZ_OK;
    int err = 0;
    GzHandle *hb = gz_create (f->h_user->device, NULL, &err);
    if (!hb) { f->h_user->err = "gz_create failed (second lane)"; return err < 0 ? err : GZ_ERR_HIP; }
    hb->profiling = f->h_user->profiling; f->h_user->helpers.push_back (hb);
    f->other.h = hb;
    if (f->h2) {                                           // (no second handle with GZ_ZIP_NO_OVERLAP)
        f->other.h2 = gz_create_background (f->h_user->device, &err);
        if (f->other.h2) { f->other.h2->profiling = f->h_user->profiling; f->h_user->helpers.push_back (f->other.h2); }
    }
    f->other_made = true;
    return GZ_OK;
}

extern "C" int gz_fastq_zip_begin (GzZipFile *f, uint8_t *text, uint64_t text_len, GzFastqVB *vbs, int n_vbs)
{
    if (!f) return GZ_ERR_ARG;
    if (f->busy) { f->h->err = "two calls are in flight already: gz_fastq_zip_end first"; return GZ_ERR_ARG; }
    const void *blob, *votes; uint64_t blob_len, votes_len;
    int rc;
    if ((rc = gz_fastq_zip_seg (f, text, text_len, vbs, n_vbs, &blob, &blob_len)) != GZ_OK
        || (rc = gz_fastq_zip_merge (f, &blob, &blob_len, 1, &votes, &votes_len)) != GZ_OK
        || (rc = zip_finish_launch (f, &votes, &votes_len, 1)) != GZ_OK) {
        if (f->h != f->h_user) f->h_user->err = f->h->err;       // (gz_last_error is asked of the handle the file was opened on)
        return rc;
    }
    f->busy = true;
    if (!f->other.busy) {                                  // the other set is free: the next call is built there
        if ((rc = zip_make_other_lane (f)) != GZ_OK) return GZ_OK;     // (no second set: the next begin will ask for an end first)
        zip_swap_lanes (f);
    }
    return GZ_OK;
}

extern "C" int gz_fastq_zip_end (GzZipFile *f)
{
    if (!f) return GZ_ERR_ARG;
    if (f->other.busy) zip_swap_lanes (f);                 // the older call
    if (!f->busy) { f->h->err = "no call in flight"; return GZ_ERR_ARG; }
    const int rc = zip_finish_wait (f);
    f->busy = false;
    if (rc != GZ_OK && f->h != f->h_user) f->h_user->err = f->h->err;
    return rc;
}

extern "C" void gz_zip_prediction (const GzZipFile *f, uint32_t *hits, uint32_t *misses)
{
    if (hits)   *hits   = f ? f->h_user->zip_pred_hits : 0;
    if (misses) *misses = f ? f->h_user->zip_pred_misses : 0;
}

extern "C" void gz_zip_speculation (const GzZipFile *f, uint32_t *hits, uint32_t *misses)
{
    if (hits)   *hits   = f ? f->h_user->zip_spec_hits : 0;
    if (misses) *misses = f ? f->h_user->zip_spec_misses : 0;
}

// ---- N1 for VCF ----------------------------------------------------------------------------------------------------------------
extern "C" int gz_vcf_sample_columns (GzHandle *h, const uint8_t *text, const uint32_t *line_off, const uint32_t *line_len, uint32_t n_lines,
                                      const uint32_t *tab_after, const GzLinesResult *tabs_dev, uint32_t n_samples, uint32_t n_subfields,
                                      uint32_t *item_off, uint32_t *item_len, uint8_t *missing, uint32_t *n_bad_dev)
{
    if (!h || !n_bad_dev || !tabs_dev || !tab_after || (n_lines && (!text || !line_off || !line_len)) || !n_samples || !n_subfields
        || (uint64_t)n_lines * n_samples > 0xfffffff0ull || (n_lines && (!item_off || !item_len))) return GZ_ERR_ARG;
    HIPCHK (h, hipSetDevice (h->device));
    GzdVcf V; memset (&V, 0, sizeof (V));
    V.text = text; V.line_off = line_off; V.line_len = line_len; V.n_lines = n_lines; V.tab_after = tab_after; V.tabs = tabs_dev;
    V.n_samples = n_samples; V.n_sub = n_subfields; V.item_off = item_off; V.item_len = item_len; V.missing = missing; V.n_bad = n_bad_dev;
    if (!(V.first_tab = (uint32_t *)arena_alloc (h, ((size_t)n_lines + 1) * 4))) return GZ_ERR_HIP;
    HIPCHK (h, hipMemsetAsync (n_bad_dev, 0, 4, h->stream));
    if (n_lines) {
        KLAUNCH (h, k_vcf_line_tabs, dim3 ((n_lines + 255) / 256), dim3 (256), 0, V);
        const uint64_t total = (uint64_t)n_lines * n_samples;
        KLAUNCH (h, k_vcf_samples, dim3 ((uint32_t)((total + 255) / 256)), dim3 (256), 0, V);
    }
    HIPCHK (h, hipGetLastError ());
    return GZ_OK;
}
