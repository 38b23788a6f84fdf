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

// ---------------------------------------------------------------------------------------------------------
struct GzZipFile {
    GzHandle *h;
    GzFastqPlan plan;
    std::vector<GzFastqCtx> ctxs;
    std::vector<std::vector<uint8_t>> snips;
    std::vector<GzZctx *> zctx;
    std::vector<ArenaBlock> ws;            // device workspace of one call (bump allocated, reused by the next call)
    std::vector<uint8_t> stage;            // host staging
    uint32_t last_vblock_i = 0;
};

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

extern "C" GzZipFile *gz_zip_open (GzHandle *h, const GzFastqPlan *plan)
{
    if (!h || !plan || !plan->ctxs || !plan->n_ctxs || plan->n_seps > GZ_TOK_MAX_SEPS) return NULL;
    GzZipFile *f = new GzZipFile ();
    f->h = h; f->plan = *plan;
    f->ctxs.assign (plan->ctxs, plan->ctxs + plan->n_ctxs);
    f->snips.resize (plan->n_ctxs);
    for (uint32_t i = 0; i < plan->n_ctxs; i++) {
        GzFastqCtx &c = f->ctxs[i];
        if (c.snip && c.snip_len) f->snips[i].assign (c.snip, c.snip + c.snip_len);
        c.snip = f->snips[i].data ();
        if ((c.kind == GZ_FQ_ITEM_TEXT || c.kind == GZ_FQ_ITEM_INT || c.kind == GZ_FQ_ITEM_DELTA) && c.item > plan->n_seps) { delete f; return NULL; }
        if ((c.kind == GZ_FQ_CONST || c.kind == GZ_FQ_ITEM_DELTA) && !c.snip_len) { delete f; return NULL; }
        f->zctx.push_back (gz_zctx_create (plan->estimated_entries));
        if (c.lcodec) gz_zctx_commit_codec (f->zctx.back (), 1, c.lcodec);
        if (c.bcodec) gz_zctx_commit_codec (f->zctx.back (), 0, c.bcodec);
    }
    f->plan.ctxs = f->ctxs.data ();
    return f;
}

extern "C" void gz_zip_close (GzZipFile *f)
{
    if (!f) return;
    (void)hipSetDevice (f->h->device);
    (void)gz_sync (f->h);
    for (auto z : f->zctx) gz_zctx_destroy (z);
    for (auto &b : f->ws) (void)hipFree (b.base);
    delete f;
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
    if (n) KLAUNCH (h, k_tokenize_n, dim3 ((n + 255) / 256), dim3 (256), 0, T);
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

// codec_assign_best_codec (codec.c:234-363, rule of SURVEY A.8) for several streams in ONE batch of trial compressions
static int zip_assign_best_many (GzHandle *h, GzZipFile *f, const std::vector<const uint8_t *> &ptr, const std::vector<uint32_t> &len, std::vector<int> &best)
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
        uint32_t best_size = len[i] < 99999 ? len[i] : 99999;                            // NONE: the bare length (codec.c:324)
        int b = GZ_CODEC_NONE;
        for (int c = 0; c < 8; c++) {
            if (S[k + c].status != GZ_OK) return GZ_ERR;
            const uint32_t size = S[k + c].out_len + 28;                                 // framed (codec.c:328-331)
            if (size < best_size) { best_size = size; b = cand[c]; }
        }
        best[i] = b;
    }
    return GZ_OK;
}

struct ZipCol {                    // one (VBlock, context) on the device
    uint32_t n = 0;                // snips (reads of the VBlock)
    // column job outputs (ITEM_TEXT / ITEM_INT)
    int col_job = -1;              // index into the column job table
    uint8_t *b250_seg = NULL, *b250_out = NULL; uint32_t *b250_len_dev = NULL; int32_t *b250_status_dev = NULL;
    int32_t *node2word_dev = NULL;
    // local
    uint8_t *local = NULL; uint64_t local_cap = 0; int dyn_job = -1; int blob_job = -1; uint32_t *sec_len_dev = NULL;
    int icol_job = -1;
    // host side after the read-back
    uint32_t n_ol = 0, n_new = 0; bool all_the_same = false; uint64_t seg_b250_len = 0, b250_count = 0;
    uint64_t local_len = 0; int ltype = 0;
    bool has_b250 = false, has_local = false, ston_only_local = false, dropped_by_r1_host = false;
    std::vector<uint8_t> host_b250;            // CONST contexts: the generated b250 written by the host
    const uint8_t *sec_b250 = NULL; uint32_t sec_b250_len = 0;
    std::vector<uint8_t> ston_local;
    uint8_t lcodec = 0, bcodec = 0;
    int32_t ats_node = -1;
};

#define ZCHK(call) do { int rc_ = (call); if (rc_ != GZ_OK) return rc_ < 0 ? rc_ : GZ_ERR; } while (0)
#define WS(var, type, count) type *var = (type *)ws_alloc (f, (size_t)(count) * sizeof (type)); if (!var) return GZ_ERR_HIP

extern "C" int gz_fastq_zip_vblocks (GzZipFile *f, uint8_t *text, uint64_t text_len, GzFastqVB *vbs, int n_vbs)
{
    if (!f || !text || !vbs || n_vbs <= 0 || n_vbs > 16384 || text_len >= 0xfffffff0ull) return GZ_ERR_ARG;
    GzHandle *h = f->h;
    const uint32_t NC = (uint32_t)f->ctxs.size (), NV = (uint32_t)n_vbs;
    if ((uint64_t)NC * NV > 60000) { h->err = "too many (VBlock, context) pairs for one call"; return GZ_ERR_ARG; }
    for (uint32_t v = 0; v < NV; v++) {
        if (vbs[v].text_off + vbs[v].text_len > text_len || (v && vbs[v].vblock_i <= vbs[v - 1].vblock_i) || vbs[v].vblock_i <= f->last_vblock_i ||
            vbs[v].r1 >= (int32_t)v || (v && vbs[v].text_off < vbs[v - 1].text_off + vbs[v - 1].text_len)) { h->err = "VBlock table: offsets / order / r1"; return GZ_ERR_ARG; }
        vbs[v].status = GZ_ERR; vbs[v].z_data = NULL; vbs[v].z_len = 0; vbs[v].n_reads = 0; vbs[v].seq_packed = NULL; vbs[v].seq_packed_len = 0;
        vbs[v].n_bases = 0; vbs[v].seq_has_x = 0; vbs[v].n_sections = 0;
    }
    HIPCHK (h, hipSetDevice (h->device));
    int rc;
    if ((rc = gz_sync (h)) < 0) return rc;
    for (auto &b : f->ws) b.used = 0;

    // ---- phase A: lines of the whole text, first line of every VBlock (seg_get_next_line, src/seg.c:200-236) -------------
    const uint8_t lookup_byte[16] = { 1 };                                  // SNIP_LOOKUP parked behind the text
    HIPCHK (h, hipMemcpyAsync (text + text_len, lookup_byte, 16, hipMemcpyHostToDevice, h->stream));
    const uint32_t lookup_off = (uint32_t)text_len;
    uint32_t line_cap = (uint32_t)(text_len / 16 + 1024);
    uint32_t *line_off = NULL, *line_len = NULL;
    struct ABlock { GzLinesResult lines; uint32_t bad_bound, n_bad_items; GzFastqResult fq; } ;
    WS (d_a, ABlock, 1);
    WS (d_vb_off, uint64_t, 2 * NV + 2);
    WS (d_first_line, uint32_t, NV + 2);
    std::vector<uint64_t> vb_off (2 * NV + 2);
    for (uint32_t v = 0; v < NV; v++) { vb_off[v] = vbs[v].text_off; vb_off[NV + 1 + v] = vbs[v].text_off + vbs[v].text_len; }
    HIPCHK (h, hipMemcpyAsync (d_vb_off, vb_off.data (), vb_off.size () * 8, hipMemcpyHostToDevice, h->stream));
    ABlock a;
    std::vector<uint32_t> first_line (NV + 2);
    for (int attempt = 0;; attempt++) {
        line_off = (uint32_t *)ws_alloc (f, ((size_t)line_cap + 8) * 4); line_len = (uint32_t *)ws_alloc (f, ((size_t)line_cap + 8) * 4);
        if (!line_off || !line_len) return GZ_ERR_HIP;
        HIPCHK (h, hipMemsetAsync (d_a, 0, sizeof (ABlock), h->stream));
        ZCHK (gz_text_lines (h, text, text_len, line_off, line_len, line_cap, &d_a->lines));
        hipLaunchKernelGGL (k_vb_bounds, dim3 ((NV + 64) / 64), dim3 (64), 0, h->stream, (const uint32_t *)line_off, (const GzLinesResult *)&d_a->lines,
                            (const uint64_t *)d_vb_off, NV, d_first_line, &d_a->bad_bound);
        HIPCHK (h, hipMemcpyAsync (&a, d_a, sizeof (a), hipMemcpyDeviceToHost, h->stream));
        HIPCHK (h, hipMemcpyAsync (first_line.data (), d_first_line, (NV + 1) * 4, hipMemcpyDeviceToHost, h->stream));
        if ((rc = gz_sync (h)) < 0) return rc;
        if (a.lines.status == GZ_ST_OK) break;
        if (attempt || a.lines.n_lines > 0xfffffff0ull) { h->err = "line index does not fit"; return GZ_ERR; }
        line_cap = (uint32_t)a.lines.n_lines + 8;                          // (short lines: once more with the exact count)
    }
    if (a.bad_bound) { h->err = "a VBlock does not start at the start of a line"; return GZ_ERR_CORRUPT; }
    const uint64_t n_lines = a.lines.n_lines;
    const uint32_t R = (uint32_t)(n_lines / 4);                            // reads of the whole text
    std::vector<uint32_t> r0 (NV + 1);
    for (uint32_t v = 0; v <= NV; v++) {
        if (first_line[v] % 4) { h->err = "a VBlock does not hold whole reads (4 lines each)"; return GZ_ERR_CORRUPT; }
        r0[v] = first_line[v] / 4;
    }
    for (uint32_t v = 0; v < NV; v++) {
        // lines between the VBlocks (text the table leaves out) must not exist: VBlock v ends where v+1 starts or the text ends
        vbs[v].n_reads = r0[v + 1] - r0[v];
    }

    // ---- phase B: reads, items, the columns of every (VBlock, context) -----------------------------------------------------
    const uint32_t NI = f->plan.n_seps + 1;
    WS (rec, uint32_t, (size_t)8 * (R + 8));
    uint32_t *l1_off = rec, *l1_len = rec + (R + 8), *seq_off = rec + 2 * (size_t)(R + 8), *seq_len = rec + 3 * (size_t)(R + 8),
             *l3_off = rec + 4 * (size_t)(R + 8), *l3_len = rec + 5 * (size_t)(R + 8), *qual_off = rec + 6 * (size_t)(R + 8), *qual_len = rec + 7 * (size_t)(R + 8);
    ZCHK (gz_fastq_records (h, text, line_off, line_len, &d_a->lines, R, l1_off, l1_len, seq_off, seq_len, l3_off, l3_len, qual_off, qual_len, &d_a->fq));
    WS (item_off, uint32_t, (size_t)NI * R + 8);
    WS (item_len, uint32_t, (size_t)NI * R + 8);
    ZCHK (gz_tokenize_column_n (h, text, l1_off, l1_len, R, f->plan.seps, f->plan.sep_counts, f->plan.n_seps, item_off, item_len, &d_a->n_bad_items));
    WS (d_vbstat, uint32_t, 2 * (size_t)NV + 2);
    hipLaunchKernelGGL (k_vb_stats, dim3 (NV), dim3 (256), 2048, h->stream, (const uint32_t *)line_off, (const uint32_t *)seq_len, (const uint32_t *)d_first_line,
                        (const uint64_t *)(d_vb_off + NV + 1), d_vbstat);

    // the dictionaries as every VBlock of this call clones them (ctx_clone)
    struct OlDev { const uint8_t *dict = NULL; const uint64_t *ci = NULL; const uint32_t *sl = NULL; uint32_t n = 0; };
    std::vector<OlDev> ol (NC);
    for (uint32_t c = 0; c < NC; c++) {
        const uint8_t k = f->ctxs[c].kind;
        if (k != GZ_FQ_ITEM_TEXT && k != GZ_FQ_ITEM_INT) continue;
        GzZctxView zv; gz_zctx_view (f->zctx[c], &zv);
        ol[c].n = zv.n_words;
        if (!zv.n_words) continue;
        uint8_t *d = (uint8_t *)ws_alloc (f, zv.dict_len + 16); uint64_t *ci = (uint64_t *)ws_alloc (f, (size_t)zv.n_words * 8); uint32_t *sl = (uint32_t *)ws_alloc (f, (size_t)zv.n_words * 4);
        if (!d || !ci || !sl) return GZ_ERR_HIP;
        HIPCHK (h, hipMemcpyAsync (d, zv.dict, zv.dict_len, hipMemcpyHostToDevice, h->stream));
        HIPCHK (h, hipMemcpyAsync (ci, zv.char_index, (size_t)zv.n_words * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK (h, hipMemcpyAsync (sl, zv.snip_len, (size_t)zv.n_words * 4, hipMemcpyHostToDevice, h->stream));
        ol[c].dict = d; ol[c].ci = ci; ol[c].sl = sl;
    }

    std::vector<ZipCol> col ((size_t)NV * NC);
    auto COL = [&] (uint32_t v, uint32_t c) -> ZipCol & { return col[(size_t)v * NC + c]; };
    std::vector<GzIntColJob> icol_jobs; std::vector<GzColumnJob> col_jobs; std::vector<GzDynIntJob> dyn_jobs; std::vector<GzBlobJob> blob_jobs; std::vector<GzAcgtJob> acgt_jobs;
    std::vector<int> acgt_of_vb (NV, -1);
    // result block: everything the host reads back in one copy
    const size_t max_jobs = (size_t)NV * NC + 1;
    WS (d_colres, GzColumnResult, max_jobs);
    WS (d_dynres, GzDynIntResult, max_jobs);
    WS (d_icolres, uint64_t, 2 * max_jobs);          // n_values, status
    WS (d_blobres, uint64_t, max_jobs);
    WS (d_acgtres, uint64_t, 2 * max_jobs);          // has_x (u32) | pad, packed_len
    WS (d_seclen, uint32_t, 2 * max_jobs);           // device-resident payload length of every (VBlock, context) local / b250 section
    WS (d_b250st, int32_t, max_jobs);
    HIPCHK (h, hipMemsetAsync (d_icolres, 0, 2 * max_jobs * 8, h->stream));
    HIPCHK (h, hipMemsetAsync (d_seclen, 0, 2 * max_jobs * 4, h->stream));

    for (uint32_t v = 0; v < NV; v++) {
        const uint32_t n = vbs[v].n_reads, rr = r0[v];
        for (uint32_t c = 0; c < NC; c++) {
            const GzFastqCtx &X = f->ctxs[c];
            ZipCol &Z = COL (v, c);
            Z.n = n;
            Z.sec_len_dev = d_seclen + 2 * ((size_t)v * NC + c);
            const uint32_t *io = item_off + (size_t)X.item * R + rr, *il = item_len + (size_t)X.item * R + rr;
            const uint32_t *coff = io, *clen = il;
            if (X.kind == GZ_FQ_ITEM_INT || X.kind == GZ_FQ_ITEM_DELTA) {
                GzIntColJob j; memset (&j, 0, sizeof (j));
                j.text = text; j.off = io; j.len = il; j.n = n; j.nothing_char = X.nothing_char; j.lookup_off = lookup_off; j.mode = X.kind == GZ_FQ_ITEM_DELTA;
                int64_t *vals = (int64_t *)ws_alloc (f, ((size_t)n + 1) * 8); uint8_t *isn = (uint8_t *)ws_alloc (f, (size_t)n + 16);
                if (!vals || !isn) return GZ_ERR_HIP;
                j.values = vals; j.is_nothing = isn;
                if (X.kind == GZ_FQ_ITEM_INT) {
                    uint32_t *so = (uint32_t *)ws_alloc (f, ((size_t)n + 1) * 4), *sl = (uint32_t *)ws_alloc (f, ((size_t)n + 1) * 4);
                    if (!so || !sl) return GZ_ERR_HIP;
                    j.snip_off = so; j.snip_len = sl; coff = so; clen = sl;
                }
                Z.icol_job = (int)icol_jobs.size ();
                j.n_values_dev = d_icolres + 2 * (size_t)Z.icol_job; j.status_dev = (int32_t *)(d_icolres + 2 * (size_t)Z.icol_job + 1);
                icol_jobs.push_back (j);
                GzDynIntJob dj; memset (&dj, 0, sizeof (dj));
                dj.values = vals; dj.is_nothing = isn; dj.n = n; dj.nothing_char = X.nothing_char; dj.n_dev = j.n_values_dev;
                if (!(Z.local = (uint8_t *)ws_alloc (f, ((size_t)n + 1) * 8))) return GZ_ERR_HIP;
                Z.local_cap = (uint64_t)n * 8;
                dj.out = Z.local; Z.dyn_job = (int)dyn_jobs.size (); dj.result_dev = d_dynres + Z.dyn_job;
                dyn_jobs.push_back (dj);
            }
            if (X.kind == GZ_FQ_ITEM_TEXT || X.kind == GZ_FQ_ITEM_INT) {
                GzColumnJob j; memset (&j, 0, sizeof (j));
                j.text = text; j.off = coff; j.len = clen; j.n = n;
                j.ol_dict = ol[c].dict; j.ol_char_index = ol[c].ci; j.ol_snip_len = ol[c].sl; j.n_ol = ol[c].n;
                // the dictionary of a column cannot exceed its snips + a NUL each; an item is at most the line
                const uint64_t dict_cap = X.kind == GZ_FQ_ITEM_INT ? (uint64_t)n * 24 + 64 : vbs[v].text_len + n + 64;
                j.node_index = (int32_t *)ws_alloc (f, ((size_t)n + 1) * 4); j.dict = (uint8_t *)ws_alloc (f, dict_cap); j.dict_cap = dict_cap;
                j.node_char_index = (uint64_t *)ws_alloc (f, ((size_t)n + 1) * 8); j.node_snip_len = (uint32_t *)ws_alloc (f, ((size_t)n + 1) * 4);
                j.counts = (uint32_t *)ws_alloc (f, ((size_t)n + ol[c].n + 1) * 4); j.b250 = (uint8_t *)ws_alloc (f, (size_t)n * 4 + 16);
                if (!j.node_index || !j.dict || !j.node_char_index || !j.node_snip_len || !j.counts || !j.b250) return GZ_ERR_HIP;
                Z.col_job = (int)col_jobs.size (); j.result_dev = d_colres + Z.col_job;
                Z.b250_seg = j.b250; Z.n_ol = ol[c].n;
                col_jobs.push_back (j);
            }
            if (X.kind == GZ_FQ_SEQ || X.kind == GZ_FQ_QUAL) {
                GzBlobJob j; memset (&j, 0, sizeof (j));
                j.text = text; j.off = (X.kind == GZ_FQ_SEQ ? seq_off : qual_off) + rr; j.len = (X.kind == GZ_FQ_SEQ ? seq_len : qual_len) + rr; j.n = n;
                Z.local_cap = vbs[v].text_len + 64;
                if (!(Z.local = (uint8_t *)ws_alloc (f, Z.local_cap + 64))) return GZ_ERR_HIP;
                j.out = Z.local; Z.blob_job = (int)blob_jobs.size (); j.out_len_dev = d_blobres + Z.blob_job;
                blob_jobs.push_back (j);
                if (X.kind == GZ_FQ_SEQ) {
                    GzAcgtJob aj; memset (&aj, 0, sizeof (aj));
                    aj.seq = Z.local; aj.n_dev = j.out_len_dev; aj.n_max = Z.local_cap;
                    aj.packed = (uint8_t *)ws_alloc (f, gz_acgt_packed_len (Z.local_cap) + 64);
                    aj.x = Z.local;                                         // NONREF_X overlays NONREF (codec_acgt.c:66-70)
                    if (!aj.packed) return GZ_ERR_HIP;
                    acgt_of_vb[v] = (int)acgt_jobs.size ();
                    aj.has_x_dev = (uint32_t *)(d_acgtres + 2 * (size_t)acgt_of_vb[v]); aj.packed_len_dev = d_acgtres + 2 * (size_t)acgt_of_vb[v] + 1;
                    vbs[v].seq_packed = aj.packed;
                    acgt_jobs.push_back (aj);
                }
            }
        }
    }
    ZCHK (gz_int_columns (h, icol_jobs.data (), (int)icol_jobs.size ()));
    // (column tables hold at most 65 535 rows per call)
    for (size_t at = 0; at < col_jobs.size (); at += 32768) ZCHK (gz_ctx_seg_columns (h, col_jobs.data () + at, (int)std::min<size_t> (32768, col_jobs.size () - at)));
    for (size_t at = 0; at < dyn_jobs.size (); at += 32768) ZCHK (gz_dyn_int_columns (h, dyn_jobs.data () + at, (int)std::min<size_t> (32768, dyn_jobs.size () - at)));
    for (size_t at = 0; at < blob_jobs.size (); at += 32768) ZCHK (gz_local_blob_columns (h, blob_jobs.data () + at, (int)std::min<size_t> (32768, blob_jobs.size () - at)));
    ZCHK (gz_acgt_pack_batch (h, acgt_jobs.data (), (int)acgt_jobs.size ()));

    // what the merge needs from every column, packed into one stretch
    const size_t NCJ = col_jobs.size ();
    std::vector<GzdPackJob> pack (NCJ);
    uint64_t pack_cap = 64;
    for (size_t k = 0; k < NCJ; k++) {
        const GzColumnJob &j = col_jobs[k];
        pack[k].dict = j.dict; pack[k].nci = j.node_char_index; pack[k].nsl = j.node_snip_len; pack[k].counts = j.counts; pack[k].n_ol = j.n_ol; pack[k].res = j.result_dev;
        pack_cap += j.dict_cap + 16ull * j.n + 4ull * j.n_ol + 64;
    }
    // (worst case = every snip a new word; the usual case is a few hundred bytes per column)
    const uint64_t pack_cap_used = std::min<uint64_t> (pack_cap, (uint64_t)64 << 20);
    WS (d_pack, GzdPackJob, NCJ + 1);
    WS (d_pack_total, uint64_t, 2);
    uint8_t *d_staging = (uint8_t *)ws_alloc (f, pack_cap_used);
    if (!d_staging) return GZ_ERR_HIP;
    if (NCJ) {
        HIPCHK (h, hipMemcpyAsync (d_pack, pack.data (), NCJ * sizeof (GzdPackJob), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL (k_pack_sizes, dim3 (1), dim3 (1), 0, h->stream, d_pack, (uint32_t)NCJ, pack_cap_used, d_pack_total);
        hipLaunchKernelGGL (k_pack_copy, dim3 ((uint32_t)NCJ), dim3 (256), 0, h->stream, (const GzdPackJob *)d_pack, d_staging, (const uint64_t *)d_pack_total);
    }
    // ---- read back (second wait)
    std::vector<GzColumnResult> colres (NCJ + 1); std::vector<GzDynIntResult> dynres (dyn_jobs.size () + 1);
    std::vector<uint64_t> icolres (2 * icol_jobs.size () + 2), blobres (blob_jobs.size () + 1), acgtres (2 * acgt_jobs.size () + 2);
    std::vector<uint32_t> vbstat (2 * (size_t)NV + 2);
    uint64_t pack_total[2] = { 0, 1 };
    if (NCJ) {
        HIPCHK (h, hipMemcpyAsync (colres.data (), d_colres, NCJ * sizeof (GzColumnResult), hipMemcpyDeviceToHost, h->stream));
        HIPCHK (h, hipMemcpyAsync (pack.data (), d_pack, NCJ * sizeof (GzdPackJob), hipMemcpyDeviceToHost, h->stream));
        HIPCHK (h, hipMemcpyAsync (pack_total, d_pack_total, 16, hipMemcpyDeviceToHost, h->stream));
    }
    if (!dyn_jobs.empty ())  HIPCHK (h, hipMemcpyAsync (dynres.data (), d_dynres, dyn_jobs.size () * sizeof (GzDynIntResult), hipMemcpyDeviceToHost, h->stream));
    if (!icol_jobs.empty ()) HIPCHK (h, hipMemcpyAsync (icolres.data (), d_icolres, icol_jobs.size () * 16, hipMemcpyDeviceToHost, h->stream));
    if (!blob_jobs.empty ()) HIPCHK (h, hipMemcpyAsync (blobres.data (), d_blobres, blob_jobs.size () * 8, hipMemcpyDeviceToHost, h->stream));
    if (!acgt_jobs.empty ()) HIPCHK (h, hipMemcpyAsync (acgtres.data (), d_acgtres, acgt_jobs.size () * 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK (h, hipMemcpyAsync (vbstat.data (), d_vbstat, 2 * (size_t)NV * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK (h, hipMemcpyAsync (&a, d_a, sizeof (a), hipMemcpyDeviceToHost, h->stream));
    if ((rc = gz_sync (h)) < 0) return rc;
    if (a.fq.first_bad != 0xffffffffu) {
        for (uint32_t v = 0; v < NV; v++) if (a.fq.first_bad >= r0[v] && a.fq.first_bad < r0[v + 1]) vbs[v].status = GZ_ERR_CORRUPT;
        h->err = "not FASTQ: a read is not '@'.. / SEQ / '+'.. / QUAL of SEQ's length (fastq.c:1008-1010,1076,1121)"; return GZ_ERR_CORRUPT;
    }
    if (a.n_bad_items) { h->err = "a line 1 does not fit the container of the plan (the reference would re-discover the flavor, qname.c:823-826)"; return GZ_ERR_CORRUPT; }
    for (size_t k = 0; k < icol_jobs.size (); k++)
        if ((int32_t)icolres[2 * k + 1] == GZ_ST_CORRUPT) { h->err = "an ordered item is not an integer (qname.c:750-756)"; return GZ_ERR_CORRUPT; }
    for (size_t k = 0; k < NCJ; k++) if (colres[k].status != 1) { h->err = "column dictionary capacity"; return GZ_ERR; }
    if (!pack_total[1]) { h->err = "merge staging buffer too small"; return GZ_ERR; }
    f->stage.resize (pack_total[0] + 16);
    if (pack_total[0]) HIPCHK (h, hipMemcpy (f->stage.data (), d_staging, pack_total[0], hipMemcpyDeviceToHost));

    // ---- a4: the merge, context by context, VBlocks in order (ctx_merge_in_vb_ctx, src/zip.c:578) ---------------------------
    const uint8_t ATS = 0x20, PAIRED = 0x04;
    std::vector<int32_t> n2w_host;                      // all node2word arrays, uploaded in one copy
    std::vector<size_t> n2w_at ((size_t)NV * NC, 0);
    for (uint32_t c = 0; c < NC; c++) {
        const GzFastqCtx &X = f->ctxs[c];
        GzZctx *z = f->zctx[c];
        for (uint32_t v = 0; v < NV; v++) {
            ZipCol &Z = COL (v, c);
            const bool is_r2 = vbs[v].r1 >= 0;
            GzMergeJob m; memset (&m, 0, sizeof (m));
            m.vblock_i = vbs[v].vblock_i;
            if (Z.dyn_job >= 0) { Z.local_len = dynres[Z.dyn_job].len; Z.ltype = dynres[Z.dyn_job].ltype; Z.has_local = Z.local_len != 0; }
            if (Z.blob_job >= 0) { Z.local_len = blobres[Z.blob_job]; Z.ltype = GZ_LT_BLOB; Z.has_local = Z.local_len != 0; }
            if (X.kind == GZ_FQ_SEQ) {                                     // NONREF itself leaves the path 2-bit packed; what stays is NONREF_X
                const int aj = acgt_of_vb[v];
                vbs[v].n_bases = Z.local_len; vbs[v].seq_packed_len = acgtres[2 * aj + 1]; vbs[v].seq_has_x = (uint32_t)acgtres[2 * aj] != 0;
                Z.has_local = vbs[v].seq_has_x != 0; Z.ltype = GZ_LT_UINT8;  // NONREF_X.ltype (codec_acgt.c:34-35)
                continue;
            }
            if (X.kind == GZ_FQ_QUAL) continue;                            // no b250, nothing to merge
            if (!Z.n) continue;                                            // an empty VBlock segs nothing
            if (Z.col_job >= 0) {
                const GzColumnResult &r = colres[Z.col_job];
                const GzdPackJob &p = pack[Z.col_job];
                Z.n_new = r.n_new; Z.all_the_same = r.all_the_same != 0; Z.seg_b250_len = r.b250_len; Z.b250_count = r.b250_count;
                m.n_ol = Z.n_ol; m.n_new = r.n_new;
                m.dict = f->stage.data () + p.at[0]; m.node_char_index = (const uint64_t *)(f->stage.data () + p.at[1]);
                m.node_snip_len = (const uint32_t *)(f->stage.data () + p.at[2]); m.counts = (const uint32_t *)(f->stage.data () + p.at[3]);
                m.b250_len = r.b250_len;
                if (Z.all_the_same) {
                    // the one node of the column: an ol word (index with a count) or the VBlock's first new node
                    int32_t node = -1;
                    for (uint32_t k = 0; k < Z.n_ol && node < 0; k++) if (m.counts[k]) node = (int32_t)k;
                    if (node < 0 && r.n_new) node = (int32_t)Z.n_ol;
                    Z.ats_node = node;                                     // (-1: every snip was empty / missing: not droppable)
                }
            }
            else {
                // GZ_FQ_CONST / GZ_FQ_ITEM_DELTA: every line segs `snip` - one node, count = lines (b250_seg_append's
                // all-the-same collapse, b250.c:117-141); evaluated here, no device work. The snip is looked up in the
                // dictionary as it is NOW: a word added by an earlier VBlock of this call then counts as cloned, which
                // changes no byte (the node of a new word and the index of a cloned one convert to the same word index)
                const uint32_t found = zctx_find (z, gz_snip_mix (X.snip, X.snip_len), X.snip, X.snip_len);
                const uint32_t n_words = (uint32_t)z->snip_len.size ();
                Z.n_ol = n_words; Z.all_the_same = true; Z.b250_count = Z.n;
                Z.seg_b250_len = found == GZ_NO_WORD ? 4 : found <= 126 ? 1 : found <= 16508 ? 2 : found <= 2113660 ? 3 : 4;
                Z.n_new = found == GZ_NO_WORD; Z.ats_node = found == GZ_NO_WORD ? (int32_t)n_words : (int32_t)found;
                std::vector<uint32_t> cnt ((size_t)n_words + 1, 0);
                cnt[(size_t)Z.ats_node] = Z.n;
                const uint64_t one_ci = 0; const uint32_t one_sl = X.snip_len;
                int32_t w1 = -1; uint8_t no_ston[8];
                m.n_ol = n_words; m.n_new = Z.n_new; m.dict = X.snip; m.node_char_index = &one_ci; m.node_snip_len = &one_sl; m.counts = cnt.data ();
                m.b250_len = Z.seg_b250_len; m.local_len = Z.local_len;
                m.flags = X.flags | ATS; m.pair2_identical = is_r2 && X.pair_identical; m.ats_node_index = Z.ats_node;
                if (is_r2) { const ZipCol &R1 = COL ((uint32_t)vbs[v].r1, c); m.b250_r1_len = R1.has_b250 ? 1 : 0; m.local_r1_len = R1.has_local ? 1 : 0; }
                m.node2word = &w1; m.ston_local = no_ston; m.ston_cap = 0;
                if ((rc = gz_ctx_merge (z, &m)) != GZ_OK) { h->err = "gz_ctx_merge (constant snip)"; return rc < 0 ? rc : GZ_ERR; }
                Z.lcodec = m.lcodec; Z.bcodec = m.bcodec;
                Z.has_b250 = !m.dropped_b250;
                if (Z.has_b250) { Z.host_b250.resize (4); Z.host_b250.resize (zip_piz_put (Z.host_b250.data (), found == GZ_NO_WORD ? w1 : (int64_t)found)); }
                continue;
            }
            // zip_handle_unique_words_ctxs (src/zip.c:136-166): a context without local whose every entry is a word new to the
            // VBlock (a unique ID) hands its whole dictionary to local; nodes and b250 are gone, nothing is merged
            if (!Z.local_len && !X.no_stons && (X.flags & 3) != 3 && !Z.all_the_same && Z.n_new && Z.n_new == Z.b250_count && Z.n_new >= Z.n / 5 && Z.b250_count != 1) {
                Z.has_b250 = false; Z.has_local = true; Z.ltype = GZ_LT_SINGLETON;
                Z.local = col_jobs[Z.col_job].dict; Z.local_len = colres[Z.col_job].dict_len; Z.local_cap = Z.local_len;
                GzZctxView zv; gz_zctx_view (z, &zv);
                Z.lcodec = zv.lcodec; Z.bcodec = zv.bcodec;
                continue;
            }
            // device columns: flags, singleton rule (zip_handle_unique_words_ctxs src/zip.c:136-166 makes a context without
            // local an LT_SINGLETON one; ctx_can_have_singletons src/context.h:263-265)
            m.flags = X.flags | (Z.all_the_same ? ATS : 0);
            m.pair2_identical = is_r2 && X.pair_identical;
            m.local_len = Z.local_len;
            m.ats_node_index = Z.ats_node;
            if (is_r2) { const ZipCol &R1 = COL ((uint32_t)vbs[v].r1, c); m.b250_r1_len = R1.has_b250 ? 1 : 0; m.local_r1_len = R1.has_local ? 1 : 0; }
            m.can_have_singletons = !Z.local_len && !X.no_stons && (X.flags & 3) != 3 && !Z.all_the_same;
            if (Z.all_the_same && Z.ats_node < 0) m.no_drop_b250 = 1;
            n2w_at[(size_t)v * NC + c] = n2w_host.size ();
            n2w_host.resize (n2w_host.size () + Z.n_new + 1);
            m.node2word = n2w_host.data () + n2w_at[(size_t)v * NC + c];
            Z.ston_local.resize ((size_t)(colres[Z.col_job].dict_len) + 8);
            m.ston_local = Z.ston_local.data (); m.ston_cap = Z.ston_local.size ();
            if ((rc = gz_ctx_merge (z, &m)) != GZ_OK) { h->err = "gz_ctx_merge"; return rc < 0 ? rc : GZ_ERR; }
            Z.ston_local.resize (m.ston_len);
            Z.lcodec = m.lcodec; Z.bcodec = m.bcodec;
            Z.has_b250 = !m.dropped_b250 && Z.seg_b250_len != 0;
            if (m.ston_len) { Z.has_local = true; Z.ston_only_local = true; Z.ltype = GZ_LT_SINGLETON; Z.local_len = m.ston_len; }
        }
    }
    // locals of contexts without a b250 merge still inherit the committed codec (context.c:980-981)
    for (uint32_t c = 0; c < NC; c++) {
        GzZctxView zv; gz_zctx_view (f->zctx[c], &zv);
        for (uint32_t v = 0; v < NV; v++) { ZipCol &Z = COL (v, c); if (!Z.lcodec) Z.lcodec = zv.lcodec; if (!Z.bcodec) Z.bcodec = zv.bcodec; }
    }

    // ---- phase C: b250 generation, locals into file order, R2 == R1 drops --------------------------------------------------
    int32_t *d_n2w = (int32_t *)ws_alloc (f, (n2w_host.size () + 1) * 4);
    if (!d_n2w) return GZ_ERR_HIP;
    if (!n2w_host.empty ()) HIPCHK (h, hipMemcpyAsync (d_n2w, n2w_host.data (), n2w_host.size () * 4, hipMemcpyHostToDevice, h->stream));
    // small host-made payloads (constant b250s, singletons) go up in one copy
    std::vector<uint8_t> small; std::vector<std::pair<ZipCol *, std::pair<int, size_t>>> small_ref;
    for (auto &Z : col) {
        if (Z.has_b250 && !Z.host_b250.empty ()) { small_ref.push_back ({ &Z, { 0, small.size () } }); small.insert (small.end (), Z.host_b250.begin (), Z.host_b250.end ()); small.resize ((small.size () + 15) & ~(size_t)15); }
        if (Z.ston_only_local && !Z.ston_local.empty ()) { small_ref.push_back ({ &Z, { 1, small.size () } }); small.insert (small.end (), Z.ston_local.begin (), Z.ston_local.end ()); small.resize ((small.size () + 15) & ~(size_t)15); }
    }
    uint8_t *d_small = (uint8_t *)ws_alloc (f, small.size () + 16);
    if (!d_small) return GZ_ERR_HIP;
    if (!small.empty ()) HIPCHK (h, hipMemcpyAsync (d_small, small.data (), small.size (), hipMemcpyHostToDevice, h->stream));
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
                j.node2word = d_n2w + n2w_at[(size_t)v * NC + c]; j.n_new_nodes = Z.n_new;
                if (!(Z.b250_out = (uint8_t *)ws_alloc (f, Z.seg_b250_len + 16))) return GZ_ERR_HIP;
                j.out = Z.b250_out; j.out_len_dev = Z.sec_len_dev + 1; j.status_dev = d_b250st + ((size_t)v * NC + c);
                if (is_r2 && X.pair_identical) {
                    const ZipCol &R1 = COL ((uint32_t)vbs[v].r1, c);
                    if (R1.has_b250 && R1.b250_out) { j.r1 = R1.b250_out; j.r1_len_dev = R1.sec_len_dev + 1; }
                }
                Z.sec_b250 = Z.b250_out; Z.sec_b250_len = (uint32_t)Z.seg_b250_len;
                bjobs.push_back (j);
            }
            if (Z.has_b250 && Z.col_job < 0 && is_r2 && X.pair_identical) {            // host-made b250s: compared here (b250.c:270-277)
                const ZipCol &R1 = COL ((uint32_t)vbs[v].r1, c);
                if (R1.has_b250 && R1.host_b250 == Z.host_b250) Z.has_b250 = false;
            }
            if (Z.has_local && Z.dyn_job >= 0) {
                GzLocalJob j; memset (&j, 0, sizeof (j));
                j.data = Z.local; j.n = Z.n; j.dyn_dev = d_dynres + Z.dyn_job; j.len_dev = Z.sec_len_dev;
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

    // ---- a8: contexts whose codec the file does not know yet: trial compressions on the first VBlock that has >= 50 bytes of
    // the stream, committed to the file-level context (codec.c:309-312,352-363); until then the section writer's RANB fallback
    {
        std::vector<const uint8_t *> ptr; std::vector<uint32_t> len; std::vector<std::pair<uint32_t, int>> who;   // (context, is_local)
        bool need_lens = false;
        for (uint32_t c = 0; c < NC; c++) {
            GzZctxView zv; gz_zctx_view (f->zctx[c], &zv);
            if (!zv.lcodec || !zv.bcodec) need_lens = true;
        }
        if (need_lens) {
            std::vector<uint32_t> seclen (2 * (size_t)NV * NC);
            HIPCHK (h, hipMemcpyAsync (seclen.data (), d_seclen, seclen.size () * 4, hipMemcpyDeviceToHost, h->stream));
            if ((rc = gz_sync (h)) < 0) return rc;
            for (uint32_t c = 0; c < NC; c++) {
                GzZctxView zv; gz_zctx_view (f->zctx[c], &zv);
                for (int is_local = 0; is_local < 2; is_local++) {
                    if (is_local ? zv.lcodec : zv.bcodec) continue;
                    for (uint32_t v = 0; v < NV; v++) {
                        ZipCol &Z = COL (v, c);
                        uint32_t L = 0; const uint8_t *p = NULL;
                        if (is_local && Z.has_local) { p = Z.local; L = Z.dyn_job >= 0 ? seclen[2 * ((size_t)v * NC + c)] : (uint32_t)Z.local_len; }
                        if (!is_local && Z.has_b250) { p = Z.sec_b250; L = Z.col_job >= 0 ? seclen[2 * ((size_t)v * NC + c) + 1] : Z.sec_b250_len; }
                        if (L >= 50) { ptr.push_back (p); len.push_back (L); who.push_back ({ c, is_local }); break; }
                    }
                }
            }
            std::vector<int> best;
            if ((rc = zip_assign_best_many (h, f, ptr, len, best)) != GZ_OK) return rc;
            for (size_t k = 0; k < who.size (); k++) {
                if (!best[k]) continue;
                gz_zctx_commit_codec (f->zctx[who[k].first], who[k].second, best[k]);
                for (uint32_t v = 0; v < NV; v++) { ZipCol &Z = COL (v, who[k].first); if (who[k].second) { if (!Z.lcodec) Z.lcodec = (uint8_t)best[k]; } else if (!Z.bcodec) Z.bcodec = (uint8_t)best[k]; }
            }
        }
    }

    // ---- a15 + a9-a13 + a16: sections in the reference's order, compressed, framed -------------------------------------------
    std::vector<GzVBlock> V (NV);
    std::vector<std::vector<GzSection>> secs (NV);
    for (uint32_t v = 0; v < NV; v++) {
        const bool is_r2 = vbs[v].r1 >= 0, is_r1 = f->plan.paired && !is_r2;
        std::vector<GzSecOrderIn> in (NC);
        for (uint32_t c = 0; c < NC; c++) {
            const ZipCol &Z = COL (v, c);
            in[c].did_i = f->ctxs[c].did_i; in[c].local_dep = f->ctxs[c].local_dep; in[c].has_local = Z.has_local; in[c].ston_only_local = Z.ston_only_local; in[c].has_b250 = Z.has_b250;
        }
        std::vector<uint32_t> order (2 * NC);
        const uint32_t ns = gz_section_order (in.data (), NC, vbs[v].vblock_i, order.data ());
        for (uint32_t k = 0; k < ns; k++) {
            const uint32_t c = order[k] / 2; const bool is_b250 = order[k] & 1;
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
            }
            else {
                s.section_type = GZ_SEC_LOCAL; s.data = Z.local; s.data_len = (uint32_t)std::min<uint64_t> (Z.local_len, Z.local_cap ? Z.local_cap : Z.local_len);
                if (Z.dyn_job >= 0) { s.data_len = (uint32_t)Z.local_cap; s.data_len_dev = Z.sec_len_dev; }
                s.codec = Z.lcodec; s.ltype = (uint8_t)Z.ltype;
                const bool int_lt = Z.ltype >= GZ_LT_INT8 && Z.ltype <= GZ_LT_UINT64;
                if (int_lt) s.b250_size_or_nothing_char = X.nothing_char ? X.nothing_char : 0xff;    // zfile.c:344-345
                if (is_r1 && X.pair_identical) s.flags |= PAIRED;                                     // zfile.c:323-325
            }
            secs[v].push_back (s);
        }
        GzVBlock &B = V[v]; memset (&B, 0, sizeof (B));
        B.vblock_i = vbs[v].vblock_i; B.recon_size = (uint32_t)vbs[v].text_len; B.longest_line_len = vbstat[2 * v]; B.longest_seq_len = vbstat[2 * v + 1];
        B.sections = secs[v].data (); B.n_sections = (uint32_t)secs[v].size ();
        B.z_cap = gz_vb_z_bound (B.sections, B.n_sections);
        if (!(B.z_data = (uint8_t *)ws_alloc (f, B.z_cap + 64))) return GZ_ERR_HIP;
    }
    ZCHK (gz_vb_compress_batch (h, V.data (), (int)NV));
    std::vector<int32_t> b250st ((size_t)NV * NC, 1);
    HIPCHK (h, hipMemcpyAsync (b250st.data (), d_b250st, b250st.size () * 4, hipMemcpyDeviceToHost, h->stream));
    rc = gz_sync (h);
    if (rc < 0) return rc;
    for (uint32_t v = 0; v < NV; v++) {
        for (uint32_t c = 0; c < NC; c++) { const ZipCol &Z = COL (v, c); if (Z.has_b250 && Z.col_job >= 0 && b250st[(size_t)v * NC + c] == -5) { h->err = "b250 generation: malformed stream"; return GZ_ERR; } }
        vbs[v].status = V[v].status; vbs[v].z_data = V[v].z_data; vbs[v].z_len = V[v].z_len; vbs[v].n_sections = V[v].n_sections;
        if (V[v].status != GZ_OK) rc = GZ_ERR;
    }
    f->last_vblock_i = vbs[NV - 1].vblock_i;
    return rc == GZ_ERR ? GZ_ERR : GZ_OK;
}
