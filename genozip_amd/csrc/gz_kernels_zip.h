// gz_kernels_zip.h -- gfx950 kernels of the VBlock compute driver (gz_zip.h): the batched forms the driver needs so that
// one launch covers every (VBlock, context) of a batch.
//
//   k_vb_bounds        first line of every VBlock from its text offset (txtfile_read_vblock cuts at record boundaries,
//                      src/txtfile.c:1228: a VBlock starts where a line starts)
//   k_tokenize_n       items of a container whose separators may be "the n-th occurrence" (CI0_COLONn, qname_flavors.h:40-49)
//   k_icol_*           seg_integer_or_not (src/seg.c:531-560) / seg_self_delta (src/seg.c:688-718) over a table of columns
//   k_local_order_jobs zip_generate_local's byte order step (src/zip.c:185-216) over a table of locals whose type was decided
//                      on the device (dyn_int_get_ltype)
//   k_bufs_identical   "pair identical": an R2 local that equals its R1 counterpart is dropped (src/zip.c:224-234)
//   k_acgt_pack_jobs   codec_acgt_pack (src/codec_acgt.c:45-55) over a table of NONREF locals
#pragma once
#include "gz_device.h"
#include "gz_devutil.h"
#include "gz_kernels_seg.h"
#include "gz_kernels_ctx.h"

// ---- VBlock boundaries ------------------------------------------------------------------------------------------------
// one thread per boundary (the start and the end of every VBlock: the VBlocks of a call need not be adjacent): the index of
// the first line that starts at or after the offset (binary search over the line starts; the end of the text -> n_lines).
// A boundary inside a line is reported in *bad.
__global__ void k_vb_bounds (const uint32_t *line_off, const GzLinesResult *lines, const uint64_t *offs, uint32_t n, uint64_t text_len, uint32_t *first_line, uint32_t *bad)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t n_lines = lines->n_lines, want = offs[i];
    uint64_t lo = 0, hi = n_lines;
    while (lo < hi) { const uint64_t mid = (lo + hi) / 2; if (line_off[mid] < want) lo = mid + 1; else hi = mid; }
    first_line[i] = (uint32_t)lo;
    if (lo < n_lines ? line_off[lo] != want : want != text_len) atomicMax (bad, 1u);
}

// ---- tokenizer with n-th occurrence separators --------------------------------------------------------------------------
struct GzdTokensN {
    const uint8_t *text; const uint32_t *off, *len; uint32_t n;
    uint8_t seps[GZ_TOK_MAX_SEPS + 1], counts[GZ_TOK_MAX_SEPS + 1]; uint32_t n_seps;
    uint32_t *item_off, *item_len; uint32_t *n_bad;
};

// grid (tiles of 256 snips), 64 bytes of LDS: item i ends at the counts[i]-th seps[i] after item i-1; the last item is the rest.
// A thread per snip, 16 bytes a load (a line 1 is ~60 bytes: 4 requests instead of 60 - with half a million snips in flight their
// lines do not stay in L2, so every request goes out to the fabric); the text has 16 bytes of slack behind it.
__global__ void __launch_bounds__(256) k_tokenize_n (GzdTokensN T)
{
    uint8_t *sh = gz_lds;                                                        // [0..32) separators, [32..64) counts
    if (threadIdx.x <= GZ_TOK_MAX_SEPS) { sh[threadIdx.x] = T.seps[threadIdx.x]; sh[32 + threadIdx.x] = T.counts[threadIdx.x]; }
    __syncthreads ();
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= T.n) return;
    const uint32_t len = T.len[k], off = T.off[k], n_seps = T.n_seps;
    const uint8_t *s = T.text + off;
    uint32_t at = 0, i = 0, left = n_seps ? sh[32] : 0, sep = n_seps ? sh[0] : 0x100;
    for (uint32_t base = 0; base < len && i < n_seps; base += 16) {
        const gz_u32x4_unaligned v = *(const gz_u32x4_unaligned *)(s + base);
        #pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t c = (v[j >> 2] >> ((j & 3) * 8)) & 0xff, e = base + j;
            if (e < len && c == sep && !--left) {
                T.item_off[(uint64_t)i * T.n + k] = off + at; T.item_len[(uint64_t)i * T.n + k] = e - at;
                at = e + 1; i++;
                if (i < n_seps) { sep = sh[i]; left = sh[32 + i]; } else sep = 0x100;
            }
        }
    }
    if (i < n_seps) {
        atomicAdd (T.n_bad, 1u);
        for (uint32_t j = 0; j <= n_seps; j++) { T.item_off[(uint64_t)j * T.n + k] = off; T.item_len[(uint64_t)j * T.n + k] = j ? 0 : len; }
    }
    else { T.item_off[(uint64_t)n_seps * T.n + k] = off + at; T.item_len[(uint64_t)n_seps * T.n + k] = len - at; }
}

// ---- integer columns -------------------------------------------------------------------------------------------------
// mode 0: seg_integer_or_not - integers (and the nothing_char) go, compacted, to `values`, their snip becomes SNIP_LOOKUP
// mode 1: seg_self_delta - every snip must be an integer; values[k] = v[k] - v[k-1] with v[-1] = 0 (ctx->last_value starts
//         at 0 in every VBlock); a snip that is not an integer sets status to GZ_ST_CORRUPT (the reference would seg such a
//         line through another branch, src/qname.c:750-756: the caller then has to fall back to mode 0 for the column)
struct GzdIntCol {
    const uint8_t *text; const uint32_t *off, *len; uint32_t n; uint32_t nothing_char; uint32_t lookup_off; uint32_t mode;
    uint32_t *snip_off, *snip_len; int64_t *values; uint8_t *is_nothing; uint64_t *n_values; int32_t *status;
    uint64_t *tile;
};

__device__ static inline int d_icol_kind (const GzdIntCol &C, uint32_t k, int64_t *v)
{
    GzdIntSplit S;
    S.text = C.text; S.off = C.off; S.len = C.len; S.n = C.n; S.nothing_char = C.nothing_char; S.lookup_off = C.lookup_off;
    return d_int_or_not (S, k, v);
}

// grid (tiles, columns)
__global__ void __launch_bounds__(256) k_icol_count (GzdIntCol *cols)
{
    const GzdIntCol C = cols[blockIdx.y];
    if (blockIdx.x * 256 >= C.n) return;
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    int64_t v;
    const int kind = k < C.n ? d_icol_kind (C, k, &v) : 0;
    if (C.mode == 1 && k < C.n && kind != 1) *C.status = GZ_ST_CORRUPT;
    uint64_t total;
    (void)d_wg_scan_u64 (kind ? 1 : 0, threadIdx.x, &total);
    if (!threadIdx.x) C.tile[blockIdx.x] = total;
}

// grid (columns)
__global__ void __launch_bounds__(256) k_icol_scan (GzdIntCol *cols)
{
    const GzdIntCol C = cols[blockIdx.x];
    const uint64_t total = d_wg_scan_array (C.tile, (C.n + 255) / 256, threadIdx.x);
    if (!threadIdx.x) *C.n_values = C.mode == 1 ? C.n : total;
}

// grid (tiles, columns)
__global__ void __launch_bounds__(256) k_icol_write (GzdIntCol *cols)
{
    const GzdIntCol C = cols[blockIdx.y];
    if (blockIdx.x * 256 >= C.n) return;
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    int64_t v = 0;
    const int kind = k < C.n ? d_icol_kind (C, k, &v) : 0;
    if (C.mode == 1) {
        if (k >= C.n) return;
        int64_t prev = 0;
        if (k) (void)d_icol_kind (C, k - 1, &prev);
        C.values[k] = kind == 1 ? v - prev : 0;
        if (C.is_nothing) C.is_nothing[k] = 0;
        return;
    }
    uint64_t total;
    const uint64_t at = C.tile[blockIdx.x] + d_wg_scan_u64 (kind ? 1 : 0, threadIdx.x, &total);
    if (k >= C.n) return;
    if (kind) { C.values[at] = v; C.is_nothing[at] = kind == 2; C.snip_off[k] = C.lookup_off; C.snip_len[k] = 1; }
    else      { C.snip_off[k] = C.off[k]; C.snip_len[k] = C.len[k]; }
}

// ---- byte order of many locals ---------------------------------------------------------------------------------------
struct GzdLocalJob {
    uint8_t *data; uint64_t n;                 // elements (upper bound when dyn != NULL)
    const GzDynIntResult *dyn;                 // type / width / length decided on the device, or NULL
    int32_t ltype;                             // used when dyn == NULL
    uint32_t *len_dev;                         // optional: receives the byte length
};

// grid (tiles of 1024 elements, jobs)
__global__ void __launch_bounds__(256) k_local_order_jobs (const GzdLocalJob *jobs)
{
    const GzdLocalJob J = jobs[blockIdx.y];
    const int lt = J.dyn ? J.dyn->ltype : J.ltype;
    const uint32_t w = J.dyn ? J.dyn->width : (lt == GZ_LT_INT16 || lt == GZ_LT_UINT16) ? 2 : (lt == GZ_LT_INT32 || lt == GZ_LT_UINT32 || lt == GZ_LT_FLOAT32) ? 4
                                             : (lt == GZ_LT_INT64 || lt == GZ_LT_UINT64 || lt == GZ_LT_FLOAT64) ? 8 : 1;
    const uint64_t n = J.dyn ? J.dyn->len / (w ? w : 1) : J.n;
    if (!blockIdx.x && !threadIdx.x && J.len_dev) *J.len_dev = (uint32_t)(n * w);
    const bool is_signed = lt == GZ_LT_INT8 || lt == GZ_LT_INT16 || lt == GZ_LT_INT32 || lt == GZ_LT_INT64;
    if (w == 1 && !is_signed) return;
    const uint64_t mask = w == 8 ? ~0ull : ((1ull << (8 * w)) - 1), sign = 1ull << (8 * w - 1);
    for (int q = 0; q < 4; q++) {
        const uint64_t i = (uint64_t)blockIdx.x * 1024 + (uint64_t)q * 256 + threadIdx.x;
        if (i >= n) return;
        uint8_t *p = J.data + i * w;
        uint64_t v = 0;
        for (uint32_t k = 0; k < w; k++) v |= (uint64_t)p[k] << (8 * k);
        if (is_signed) v = (v & sign) ? ((((~v + 1) & mask) << 1) - 1) & mask : (v << 1) & mask;   // context.h:99-100
        for (uint32_t k = 0; k < w; k++) p[k] = (uint8_t)(v >> (8 * (w - 1 - k)));
    }
}

// ---- pair identical ---------------------------------------------------------------------------------------------------
struct GzdSameJob {
    const uint8_t *a; const uint32_t *a_len;   // R2's section payload and its device-resident length
    const uint8_t *b; const uint32_t *b_len;   // R1's
    uint32_t *drop_len;                        // set to 0 when identical (the section writer then leaves the section out)
    uint32_t *flag;                            // scratch, 1 = differs
};

// grid (jobs): one workgroup per pair
__global__ void __launch_bounds__(256) k_bufs_identical (const GzdSameJob *jobs)
{
    const GzdSameJob J = jobs[blockIdx.x];
    const uint32_t n = *J.a_len;
    __shared__ uint32_t differs;
    if (!threadIdx.x) differs = n != *J.b_len;
    __syncthreads ();
    if (differs) return;
    uint32_t d = 0;
    for (uint32_t i = threadIdx.x; i < n && !d; i += 256) d |= J.a[i] != J.b[i];
    if (d) differs = 1;
    __syncthreads ();
    if (!threadIdx.x && !differs) *J.drop_len = 0;
}

// ---- CODEC_ACGT pack over many NONREF locals -----------------------------------------------------------------------------
struct GzdAcgtJob { const uint8_t *seq; const uint64_t *n_dev; uint64_t n_max; uint8_t *packed; uint8_t *x; uint32_t *has_x; uint64_t *packed_len; };

// grid (tiles of 256 x 16 bases, jobs)
__global__ void __launch_bounds__(256) k_acgt_pack_jobs (const GzdAcgtJob *jobs)
{
    const GzdAcgtJob J = jobs[blockIdx.y];
    const uint64_t n = J.n_dev ? *J.n_dev : J.n_max;
    const uint64_t packed_bytes = ((2 * n + 63) / 64) * 8, groups = (n + 15) / 16, words = packed_bytes / 4;
    if (!blockIdx.x && !threadIdx.x && J.packed_len) *J.packed_len = packed_bytes;
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint16_t *tab = (uint16_t *)gz_lds;
    tab[threadIdx.x] = (uint16_t)(d_acgt_code (threadIdx.x) | (d_acgt_exception (threadIdx.x) << 8));
    __syncthreads ();
    uint32_t any = 0;
    if (g < words) {
        uint32_t w = 0;
        if (g < groups) {
            const uint64_t base = g * 16;
            const uint32_t m = base + 16 <= n ? 16u : (uint32_t)(n - base);
            gz_u32x4_unaligned v = { 0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u };
            if (m == 16) v = *(const gz_u32x4_unaligned *)(J.seq + base);
            else for (uint32_t k = 0; k < m; k++) v[k >> 2] = (v[k >> 2] & ~(0xffu << (8 * (k & 3)))) | ((uint32_t)J.seq[base + k] << (8 * (k & 3)));
            gz_u32x4_unaligned ev = { 0, 0, 0, 0 };
            #pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t t = tab[(v[k >> 2] >> (8 * (k & 3))) & 0xff];
                w |= (t & 3) << (2 * k);
                ev[k >> 2] |= (t >> 8) << (8 * (k & 3));
            }
            any = ev[0] | ev[1] | ev[2] | ev[3];
            if (m == 16) *(gz_u32x4_unaligned *)(J.x + base) = ev;
            else for (uint32_t k = 0; k < m; k++) J.x[base + k] = (uint8_t)(ev[k >> 2] >> (8 * (k & 3)));
        }
        ((uint32_t *)J.packed)[g] = w;
    }
    if (__ballot (any != 0) && (threadIdx.x & 63) == 0 && *(volatile uint32_t *)J.has_x == 0) atomicMax (J.has_x, 1u);
}

// ---- what the host needs for the merge, packed into one staging buffer -----------------------------------------------
// Per column: dict [dict_len], node_char_index [n_new], node_snip_len [n_new], counts [n_ol + n_new] - variable sizes that
// only the device knows. k_pack_sizes lays them out one after the other (8-byte aligned), k_pack_copy copies; the host then
// reads the layout and ONE stretch of memory instead of four small copies per column.
struct GzdPackJob {
    const uint8_t *dict; const uint64_t *nci; const uint32_t *nsl; const uint32_t *counts; uint32_t n_ol;
    const GzColumnResult *res;
    uint64_t at[4];            // out: offsets of the four parts in the staging buffer
};

// grid (1), 256 threads: thread t sizes jobs t, t + 256, ... of every stretch of 256 jobs (each size sits behind two dependent loads:
// one thread walking 1 568 columns took 1.4 ms of the streamed call's main path), thread 0 adds the stretch up, everybody places its own
__global__ void __launch_bounds__(256) k_pack_sizes (GzdPackJob *jobs, uint32_t n_jobs, uint64_t cap, uint64_t *total_out)
{
    uint64_t *s_len = (uint64_t *)gz_lds, &s_base = s_len[256];            // (257 * 8 bytes of dynamic LDS)
    const uint32_t tid = threadIdx.x;
    if (blockIdx.x) return;
    if (!tid) s_base = 0;
    for (uint32_t j0 = 0; j0 < n_jobs; j0 += 256) {
        const uint32_t j = j0 + tid;
        uint64_t sz[4] = { 0, 0, 0, 0 }, len = 0;
        if (j < n_jobs) {
            const GzdPackJob &J = jobs[j];
            const uint64_t n_new = J.res->n_new;
            sz[0] = J.res->status == 1 ? J.res->dict_len : 0; sz[1] = 8 * n_new; sz[2] = 4 * n_new; sz[3] = 4 * ((uint64_t)J.n_ol + n_new);
            for (int k = 0; k < 4; k++) len += (sz[k] + 7) & ~7ull;
        }
        s_len[tid] = len;
        __syncthreads ();
        if (!tid) { uint64_t at = s_base; for (int t = 0; t < 256; t++) { const uint64_t l = s_len[t]; s_len[t] = at; at += l; } s_base = at; }
        __syncthreads ();
        if (j < n_jobs) { uint64_t at = s_len[tid]; for (int k = 0; k < 4; k++) { jobs[j].at[k] = at; at += (sz[k] + 7) & ~7ull; } }
        __syncthreads ();
    }
    if (!tid) { total_out[0] = s_base; total_out[1] = s_base <= cap; }
}

// grid (jobs)
__global__ void __launch_bounds__(256) k_pack_copy (const GzdPackJob *jobs, uint8_t *staging, const uint64_t *total)
{
    if (!total[1]) return;
    const GzdPackJob J = jobs[blockIdx.x];
    const uint64_t n_new = J.res->n_new;
    const uint8_t *src[4] = { J.dict, (const uint8_t *)J.nci, (const uint8_t *)J.nsl, (const uint8_t *)J.counts };
    const uint64_t sz[4] = { J.res->status == 1 ? J.res->dict_len : 0, 8 * n_new, 4 * n_new, 4 * ((uint64_t)J.n_ol + n_new) };
    for (int k = 0; k < 4; k++)
        for (uint64_t i = threadIdx.x; i < sz[k]; i += 256) staging[J.at[k] + i] = src[k][i];
}

// ---- VB header statistics: longest record (vb->longest_line_len, seg.c: a FASTQ "line" is the 4-line read) and longest SEQ ----
// grid (VBlocks); first_line = [starts (n_vb) | ends (n_vb)]
__global__ void __launch_bounds__(256) k_vb_stats (const uint32_t *line_off, const uint32_t *seq_len, const uint32_t *first_line, const uint64_t *vb_end, uint32_t n_vb,
                                                    uint32_t RL /* lines per record: 4 FASTQ, 1 SAM */, uint32_t *out /* [n_vb][2] */)
{
    const uint32_t v = blockIdx.x;
    const uint32_t r0 = first_line[v] / RL, r1 = first_line[n_vb + v] / RL;
    uint32_t longest = 0, longest_seq = 0;
    for (uint32_t r = r0 + threadIdx.x; r < r1; r += 256) {
        const uint64_t end = r + 1 < r1 ? line_off[RL * (r + 1)] : vb_end[v];
        const uint32_t len = (uint32_t)(end - line_off[RL * r]);
        longest = len > longest ? len : longest;
        longest_seq = seq_len[r] > longest_seq ? seq_len[r] : longest_seq;
    }
    uint32_t *sh = (uint32_t *)gz_lds;
    sh[threadIdx.x] = longest; sh[256 + threadIdx.x] = longest_seq;
    __syncthreads ();
    for (int d = 128; d; d >>= 1) {
        if ((int)threadIdx.x < d) {
            if (sh[threadIdx.x + d] > sh[threadIdx.x]) sh[threadIdx.x] = sh[threadIdx.x + d];
            if (sh[256 + threadIdx.x + d] > sh[256 + threadIdx.x]) sh[256 + threadIdx.x] = sh[256 + threadIdx.x + d];
        }
        __syncthreads ();
    }
    if (!threadIdx.x) { out[2 * v] = sh[0]; out[2 * v + 1] = sh[256]; }
}

// ---- N1 for VCF: the FORMAT subfields of every sample of every data line as columns (vcf_seg_samples, src/vcf_samples.c:1601) -----
// A data line is CHROM POS ID REF ALT QUAL FILTER INFO FORMAT sample ... sample, tab separated; a sample is its subfields in FORMAT's
// order, ':' separated, trailing ones may be left out. With the positions of all tabs of the text known (gz_byte_index), sample s of
// a line starts behind the line's (9 + s)-th tab: k_vcf_line_tabs finds every line's first tab (binary search) and checks the tab
// count; k_vcf_samples, a thread per (line, sample), cuts the sample at its colons. Items are subfield-major:
// [j][line * n_samples + s] - columns for gz_ctx_seg_columns / gz_int_columns, matrices of lines x samples for the transposes.
struct GzdVcf {
    const uint8_t *text; const uint32_t *line_off, *line_len; uint32_t n_lines;
    const uint32_t *tab_after; const GzLinesResult *tabs;    // tab_after[k + 1] = position behind the k-th tab of the text
    uint32_t n_samples, n_sub;
    uint32_t *item_off, *item_len; uint8_t *missing;         // [n_sub][n_lines * n_samples]; missing may be NULL
    uint32_t *first_tab;                                     // scratch [n_lines]
    uint32_t *n_bad;                                         // lines whose tab count is not 8 + n_samples; samples with more than n_sub subfields
};

__global__ void __launch_bounds__(256) k_vcf_line_tabs (GzdVcf V)
{
    const uint32_t l = blockIdx.x * 256 + threadIdx.x;
    if (l >= V.n_lines) return;
    const uint64_t n_tabs = V.tabs->n_lines;
    const uint32_t off = V.line_off[l], end = off + V.line_len[l];
    uint64_t lo = 0, hi = n_tabs;                            // first tab at or behind the start of the line
    while (lo < hi) { const uint64_t mid = (lo + hi) / 2; if (V.tab_after[mid + 1] - 1 < off) lo = mid + 1; else hi = mid; }
    V.first_tab[l] = (uint32_t)lo;
    const uint64_t want = 8ull + V.n_samples, last = lo + want - 1;           // the line's last tab / the first one that must not be its
    const bool ok = V.n_samples && last < n_tabs && V.tab_after[last + 1] - 1 < end && (last + 1 >= n_tabs || V.tab_after[last + 2] - 1 >= end);
    if (!ok) atomicAdd (V.n_bad, 1u);
}

__global__ void __launch_bounds__(256) k_vcf_samples (GzdVcf V)
{
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x, total = (uint64_t)V.n_lines * V.n_samples;
    if (k >= total) return;
    const uint32_t l = (uint32_t)(k / V.n_samples), s = (uint32_t)(k % V.n_samples);
    const uint64_t t = (uint64_t)V.first_tab[l] + 8 + s, n_tabs = V.tabs->n_lines;
    const uint32_t line_end = V.line_off[l] + V.line_len[l];
    uint32_t a = t < n_tabs ? V.tab_after[t + 1] : line_end;
    uint32_t e = (s + 1 < V.n_samples && t + 1 < n_tabs) ? V.tab_after[t + 2] - 1 : line_end;
    if (a > line_end) a = line_end;
    if (e > line_end) e = line_end;
    if (e < a) e = a;
    uint32_t j = 0, at = a;
    for (uint32_t i = a; i <= e; i++) {
        if (i < e && V.text[i] != ':') continue;
        if (j < V.n_sub) { V.item_off[(uint64_t)j * total + k] = at; V.item_len[(uint64_t)j * total + k] = i - at; if (V.missing) V.missing[(uint64_t)j * total + k] = 0; }
        else if (j == V.n_sub) atomicAdd (V.n_bad, 1u);
        j++; at = i + 1;
    }
    for (; j < V.n_sub; j++) { V.item_off[(uint64_t)j * total + k] = 0; V.item_len[(uint64_t)j * total + k] = 0; if (V.missing) V.missing[(uint64_t)j * total + k] = 1; }
}

// ---- the z_data of many VBlocks, one after the other (gz_fastq_zip_collect) ---------------------------------------------------------
// One launch instead of a hipMemcpyAsync per VBlock (448 of them per call in the streamed form). The destinations are packed -
// arbitrary alignment against their sources: every thread assembles 16 destination-aligned bytes from five aligned words of the
// source (head and tail of a piece byte by byte).
struct GzdPiece { const uint8_t *src; uint8_t *dst; uint64_t len; };

// grid (pieces, slices), 256 threads
__global__ void __launch_bounds__(256) k_pieces_copy (const GzdPiece *pieces)
{
    const GzdPiece P = pieces[blockIdx.x];
    if (!P.len) return;
    const uint64_t head = P.len < 16 ? P.len : ((16 - ((uintptr_t)P.dst & 15)) & 15);      // bytes up to the first aligned destination
    const uint64_t n16 = (P.len - head) / 16;
    const uint64_t per = (n16 + gridDim.y - 1) / gridDim.y;
    const uint64_t c0 = (uint64_t)blockIdx.y * per, c1 = c0 + per < n16 ? c0 + per : n16;
    for (uint64_t c = c0 + threadIdx.x; c < c1; c += 256) {
        const uint8_t *sp = P.src + head + c * 16;
        const uint32_t sh = (uint32_t)((uintptr_t)sp & 3) * 8;
        const uint32_t *w = (const uint32_t *)((uintptr_t)sp & ~(uintptr_t)3);
        uint4 o;
        if (!sh) { o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3]; }
        else {
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];     // (w[4] holds the chunk's last bytes: inside the source)
            o.x = (uint32_t)((((uint64_t)w1 << 32) | w0) >> sh); o.y = (uint32_t)((((uint64_t)w2 << 32) | w1) >> sh);
            o.z = (uint32_t)((((uint64_t)w3 << 32) | w2) >> sh); o.w = (uint32_t)((((uint64_t)w4 << 32) | w3) >> sh);
        }
        *(uint4 *)(P.dst + head + c * 16) = o;
    }
    if (blockIdx.y == 0) {
        const uint64_t tail0 = head + n16 * 16;
        for (uint64_t i = threadIdx.x; i < head; i += 256) P.dst[i] = P.src[i];
        for (uint64_t i = tail0 + threadIdx.x; i < P.len; i += 256) P.dst[i] = P.src[i];
    }
}

// ---- SQBITMAP: the snip of every read (fastq_seg_SEQ, src/fastq_seq.c:45-154, the branches a file without reference / aligner takes) ------
// A read's bases go to NONREF.local verbatim and SQBITMAP gets { SNIP_SPECIAL, FASTQ_SPECIAL_unaligned_SEQ, ' ', decimal seq_len } (:139-146);
// a read that is one base repeated is not stored: the base takes the place of the ' ' (:120-126, dl->monochar = str_is_monochar, fastq.c:1267);
// an empty read is '*' without a length (:113-117). One thread per read: 16-byte slot of snip text (prefix <= 4 bytes + <= 10 digits), its
// (offset, length) for the column kernels, and the length the NONREF gather is to take (0 for a repeated base). The 150 bases of a read
// are compared with the first 16 bytes at a time.
struct GzdSeqSnip {
    const uint8_t *text; const uint32_t *seq_off, *seq_len, *l3_len; uint32_t n;
    uint8_t prefix[4]; uint32_t prefix_len;      // { SNIP_SPECIAL, code, ' ' }: the last byte is the one a repeated base / '*' replaces
    uint8_t *slots; uint32_t *snip_off, *snip_len, *nonref_len; uint32_t *n_line3;   // n_line3: reads whose line 3 is more than "+" (the plan says L3_EMPTY)
};

__global__ void __launch_bounds__(256) k_seq_snips (GzdSeqSnip S)
{
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= S.n) return;
    const uint32_t len = S.seq_len[r];
    const uint8_t *p = S.text + S.seq_off[r];
    bool mono = len != 0;
    if (len) {
        const uint8_t c = p[0];
        const uint64_t pat = 0x0101010101010101ull * c;
        uint32_t i = 0;
        for (; mono && i < len && ((uintptr_t)(p + i) & 7); i++) mono = p[i] == c;
        for (; mono && i + 8 <= len; i += 8) mono = *(const uint64_t *)(p + i) == pat;
        for (; mono && i < len; i++) mono = p[i] == c;
    }
    uint8_t *s = S.slots + (size_t)r * 16;
    uint32_t k = 0;
    for (; k + 1 < S.prefix_len; k++) s[k] = S.prefix[k];
    s[k++] = !len ? '*' : mono ? p[0] : S.prefix[S.prefix_len - 1];
    if (len) {
        char d[10]; int nd = 0;
        for (uint32_t v = len; v; v /= 10) d[nd++] = (char)('0' + v % 10);
        while (nd) s[k++] = (uint8_t)d[--nd];
    }
    for (uint32_t z = k; z < 16; z++) s[z] = 0;
    S.snip_off[r] = r * 16; S.snip_len[r] = k;
    S.nonref_len[r] = mono ? 0 : len;
    if (S.l3_len[r]) atomicAdd (S.n_line3, 1u);
}

// ---- QUAL: the snip of every read (fastq_seg_QUAL, src/fastq_qual.c:24-47) --------------------------------------------------------------
// A line that is one score repeated (str_is_monochar, src/strings.h:176-184: an empty line and a single score count) segs
// { SNIP_SPECIAL, FASTQ_SPECIAL_monochar_QUAL, score } and takes no part in QUAL.local (dl->dont_compress_QUAL: fastq_zip_qual hands the
// codecs 0 bytes for it, :74); every other line segs { SNIP_LOOKUP } (seg_simple_lookup, src/seg.c:134-137). One thread per read: a 4-byte
// slot of snip text, its (offset, length) for the column kernels, and the length the QUAL gather / CODEC_DOMQ are to take (0 for a repeated
// score). Like k_seq_snips the scores are compared with the first 8 at a time; a line that is not one score is left after its first bytes.
struct GzdQualSnip {
    const uint8_t *text; const uint32_t *qual_off, *qual_len; uint32_t n;
    uint8_t prefix[4]; uint32_t prefix_len;      // { SNIP_SPECIAL, code }: the score follows
    uint8_t *slots; uint32_t *snip_off, *snip_len, *eff_len;
};

__global__ void __launch_bounds__(256) k_qual_snips (GzdQualSnip S)
{
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= S.n) return;
    const uint32_t len = S.qual_len[r];
    const uint8_t *p = S.text + S.qual_off[r];
    const uint8_t c = p[0];                              // (an empty line: the byte behind it - what the reference's qual[0] reads)
    bool mono = true;
    {
        const uint64_t pat = 0x0101010101010101ull * c;
        uint32_t i = 0;
        for (; mono && i < len && ((uintptr_t)(p + i) & 7); i++) mono = p[i] == c;
        for (; mono && i + 8 <= len; i += 8) mono = *(const uint64_t *)(p + i) == pat;
        for (; mono && i < len; i++) mono = p[i] == c;
    }
    uint8_t *s = S.slots + (size_t)r * 4;
    uint32_t k = 0;
    if (mono) { for (; k < S.prefix_len; k++) s[k] = S.prefix[k]; s[k++] = c; }
    else s[k++] = 1;                                     // SNIP_LOOKUP (src/context.h:33)
    for (uint32_t z = k; z < 4; z++) s[z] = 0;
    S.snip_off[r] = r * 4; S.snip_len[r] = k;
    S.eff_len[r] = mono ? 0 : len;
}

// GZ_FQ_ITEM_EXPECT: the item of every record must be exactly `want` (<= 16 bytes) - grid (tiles of 256 records)
struct GzdExpect { const uint8_t *text; const uint32_t *off, *len; uint32_t n; uint8_t want[16]; uint32_t want_len; uint32_t *n_bad; };
__global__ void __launch_bounds__(256) k_item_expect (GzdExpect X)
{
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= X.n) return;
    bool same = X.len[r] == X.want_len;
    const uint8_t *p = X.text + X.off[r];
    for (uint32_t i = 0; same && i < X.want_len; i++) same = p[i] == X.want[i];
    if (!same) atomicAdd (X.n_bad, 1u);
}

// any byte set? (the `missing` mask of gz_vcf_sample_columns: a sample that leaves trailing subfields out) - grid (tiles of 256 x 16 bytes)
__global__ void __launch_bounds__(256) k_any_set (const uint8_t *p, uint64_t n, uint32_t *count)
{
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    uint32_t c = 0;
    for (uint64_t i = i0; i < i0 + 16 && i < n; i++) c += p[i] != 0;
    if (c) atomicAdd (count, c);
}
