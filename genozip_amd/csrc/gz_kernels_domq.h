// gz_kernels_domq.h -- SURVEY 8(f) N3: CODEC_DOMQ's pre-transform on the GPU (src/codec_domq.c; restated from the stream
// format described at the top of that file and in SURVEY 2.1, nothing copied).
//
// The QUAL lines of one VBlock become four streams (each then goes through the ordinary codecs):
//   QUAL      the non-dominant scores, normalised (rank of the score among the lines sharing the line's dominant score);
//             a `no_doms` marker (= num_norm_qs) before a score that no run of the dominant score precedes, and after a
//             trailing run                                                           (codec_domq.c:438-480)
//   DOMQRUNS  the lengths of the runs of the dominant score: 0-254 = a run of that length, 255 = 254 and the run goes on
//             (:347-356); runs continue across lines (the run counter is only reset by a non-dominant score, :418-466)
//   QUALMPLX  one byte per line: row of its dominant score in the denormalisation table, | 0x80 for a diverse line (:425-432)
//   DIVRQUAL  the normalised scores of "diverse" lines (dominant score < 85 % of the line, :157-158)
// and the denormalisation table (num_doms x num_norm_qs), which the host base64-codes into DOMQRUNS' dictionary (:231-236).
//
// One 256-thread workgroup per VBlock walks its lines in rounds of 256; everything that the reference does line after line
// with carried state (the run counter, the output cursors) is a workgroup scan per round plus a carry: the run before a
// line's first non-dominant score is (its position) - (position of the last non-dominant score before it) - 1, a max-scan.
#pragma once
#include "gz_device.h"
#include "gz_devutil.h"
#include "gz_kernels_seg.h"

#define GZ_DQ_FIRST 32
#define GZ_DQ_N     95
#define GZ_DOMQ_LDS (4096 + GZ_DQ_N * GZ_DQ_N * 4 + 4 * GZ_DQ_N * 4 + 1024)

struct GzdDomq {
    const uint8_t *text; const uint32_t *off, *len; uint32_t n;
    uint8_t *qual, *runs, *mplx, *divr;            // outputs: capacities 2 * bytes + 16, bytes + bytes / 254 + 16, n + 16, bytes + 16
    uint8_t *line_dom;                             // scratch [n]
    uint8_t *normalize;                            // scratch [95][95], indexed by the (uncompacted) dominant score
    GzDomqResult *res;
};

// inclusive -> exclusive max scan over the workgroup (identity -1); *total = max over all threads
__device__ static inline int64_t d_wg_scan_max (int64_t v, int tid, int64_t *total)
{
    int64_t *sh = (int64_t *)gz_lds;
    __syncthreads ();
    sh[tid] = v;
    __syncthreads ();
    for (int d = 1; d < 256; d <<= 1) {
        const int64_t o = tid >= d ? sh[tid - d] : -1;
        __syncthreads ();
        if (o > sh[tid]) sh[tid] = o;
        __syncthreads ();
    }
    *total = sh[255];
    const int64_t excl = tid ? sh[tid - 1] : -1;
    __syncthreads ();
    return excl;
}

__device__ static inline uint32_t d_dq_run_bytes (uint64_t r) { return (uint32_t)((r + 253) / 254); }
__device__ static inline void d_dq_put_run (uint8_t *dst, uint64_t r)                 // codec_domq.c:347-356
{
    while (r) { const uint32_t sub = r < 254 ? (uint32_t)r : 254; *dst++ = r <= 254 ? (uint8_t)sub : 255; r -= sub; }
}

// grid (VBlocks), 256 threads, GZ_DOMQ_LDS bytes
__global__ void __launch_bounds__(256) k_domq (GzdDomq *jobs)
{
    const GzdDomq &J = jobs[blockIdx.x];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    uint32_t *hist  = (uint32_t *)(gz_lds + 4096);            // [95][95] per dominant score
    uint32_t *whist = hist + GZ_DQ_N * GZ_DQ_N;               // [4][95] the line a wave is looking at
    uint32_t *misc  = whist + 4 * GZ_DQ_N;                    // [0..95) lines_with_dom  [96] bad  [97] has_diverse  [98] num_norm  [99] num_doms
    for (int i = tid; i < GZ_DQ_N * GZ_DQ_N + 4 * GZ_DQ_N + 128; i += 256) hist[i] = 0;
    __syncthreads ();

    // ---- 1. every line's dominant score and the histograms per dominant score (codec_domq.c:139-176): a wave per line
    for (uint32_t i = wave; i < J.n; i += 4) {
        const uint32_t len = J.len[i];
        if (!len) continue;                                                     // (wave-uniform)
        const uint8_t *s = J.text + J.off[i];
        uint32_t *h = whist + wave * GZ_DQ_N;
        for (uint32_t k = lane; k < len; k += 64) {
            const uint32_t c = s[k];
            if (c < GZ_DQ_FIRST || c > 126) misc[96] = 1; else atomicAdd (&h[c - GZ_DQ_FIRST], 1u);
        }
        gz_wave_sync ();
        // the largest count, the higher score among equals (:152-157): lanes hold scores lane and lane + 64
        const uint32_t c0 = h[lane], c1 = lane + 64 < GZ_DQ_N ? h[lane + 64] : 0;
        uint32_t best = c1 >= c0 && lane + 64 < GZ_DQ_N ? c1 : c0, bq = c1 >= c0 && lane + 64 < GZ_DQ_N ? (uint32_t)lane + 64 : (uint32_t)lane;
        for (int m = 32; m; m >>= 1) {
            const uint32_t ob = (uint32_t)__shfl ((int)best, lane ^ m), oq = (uint32_t)__shfl ((int)bq, lane ^ m);
            if (ob > best || (ob == best && oq > bq)) { best = ob; bq = oq; }
        }
        const bool diverse = 100u * best / len < 85u;                           // DOMQ_THRESHOLD
        atomicAdd (&hist[bq * GZ_DQ_N + lane], c0);
        if (lane + 64 < GZ_DQ_N) atomicAdd (&hist[bq * GZ_DQ_N + lane + 64], c1);
        if (!lane) { J.line_dom[i] = (uint8_t)(bq | (diverse ? 0x80 : 0)); atomicAdd (&misc[bq], 1u); if (diverse) misc[97] = 1; }
        gz_wave_sync ();
        h[lane] = 0; if (lane + 64 < GZ_DQ_N) h[lane + 64] = 0;
        gz_wave_sync ();
    }
    __syncthreads ();

    // ---- 2. tables (:178-249): compact the dominant scores in ascending order; within one, rank the scores by count,
    //         descending, equal counts in ascending score order (the reference's qsort is glibc's stable merge sort at this size)
    uint8_t *dom_to_cdom = (uint8_t *)(misc + 100);           // [95] (+ room)
    if (!tid) { uint32_t nd = 0; for (int q = 0; q < GZ_DQ_N; q++) if (misc[q]) dom_to_cdom[q] = (uint8_t)nd++; misc[99] = nd; misc[98] = 0; }
    __syncthreads ();
    for (int d = 0; d < GZ_DQ_N; d++) {
        if (!misc[d]) continue;                                                 // (uniform)
        if (tid < GZ_DQ_N) {
            const uint32_t mine = hist[d * GZ_DQ_N + tid];
            uint32_t rank = 0, nz = 0;
            for (int q = 0; q < GZ_DQ_N; q++) {
                const uint32_t o = hist[d * GZ_DQ_N + q];
                nz += o != 0;
                rank += o > mine || (o == mine && q < tid && o);
            }
            J.normalize[d * GZ_DQ_N + tid] = mine ? (uint8_t)rank : 0;
            if (!tid) atomicMax (&misc[98], nz);
        }
    }
    __syncthreads ();
    const uint32_t num_norm = misc[98], num_doms = misc[99], no_doms = num_norm;
    if (tid < GZ_DQ_N && misc[tid]) {                         // denormalisation rows, compacted to num_norm columns
        const uint32_t cd = dom_to_cdom[tid];
        for (uint32_t r = 0; r < num_norm; r++) J.res->denorm[cd * num_norm + r] = 0;
        for (int q = 0; q < GZ_DQ_N; q++) if (hist[tid * GZ_DQ_N + q]) J.res->denorm[cd * num_norm + J.normalize[tid * GZ_DQ_N + q]] = (uint8_t)(q + GZ_DQ_FIRST);
    }
    __threadfence_block ();
    __syncthreads ();

    // ---- 3. the four streams, 256 lines a round; carried from round to round: positions, cursors, the last non-dominant score
    uint64_t pos = 0, at_qual = 0, at_runs = 0, at_divr = 0, at_mplx = 0;
    int64_t last_nz = -1;
    uint32_t last_len = 0;
    for (uint32_t base = 0; base < J.n; base += 256) {
        const uint32_t i = base + tid;
        const uint32_t len = i < J.n ? J.len[i] : 0;
        const uint8_t ld = len ? J.line_dom[i] : 0;
        const bool diverse = ld & 0x80;
        const uint8_t *nrm = J.normalize + (ld & 0x7f) * GZ_DQ_N;
        const uint8_t *s = J.text + (len ? J.off[i] : 0);
        // pass 1: the line on its own
        uint32_t lead = 0, trail = 0, nnz = 0, inner_marks = 0, inner_run_bytes = 0;
        if (len && !diverse) {
            uint32_t run = 0;
            for (uint32_t k = 0; k < len; k++) {
                const uint32_t v = nrm[s[k] - GZ_DQ_FIRST];
                if (!v) { run++; continue; }
                if (!nnz) lead = run;
                else if (run) inner_run_bytes += d_dq_run_bytes (run);
                else inner_marks++;
                nnz++; run = 0;
            }
            trail = run;
        }
        uint64_t tot;
        const uint64_t L = len && !diverse ? len : 0;
        const uint64_t my_pos = pos + d_wg_scan_u64 (L, tid, &tot);
        int64_t mx;
        int64_t before = d_wg_scan_max (nnz ? (int64_t)(my_pos + L - 1 - trail) : -1, tid, &mx);
        if (last_nz > before) before = last_nz;
        const uint64_t run_before = nnz ? (my_pos + lead) - (uint64_t)(before + 1) : 0;
        const uint32_t q_bytes = nnz + inner_marks + (nnz && !run_before ? 1 : 0);
        const uint32_t r_bytes = inner_run_bytes + (nnz ? d_dq_run_bytes (run_before) : 0);
        uint64_t t2, t3;
        const uint64_t e2 = d_wg_scan_u64 (((uint64_t)q_bytes << 32) | r_bytes, tid, &t2);
        const uint64_t e3 = d_wg_scan_u64 (((uint64_t)(diverse ? len : 0) << 32) | (len ? 1u : 0u), tid, &t3);
        // pass 2: write
        if (len) J.mplx[at_mplx + (uint32_t)e3] = (uint8_t)(dom_to_cdom[ld & 0x7f] | (diverse ? 0x80 : 0));
        if (len && diverse) {
            uint8_t *d = J.divr + at_divr + (e3 >> 32);
            for (uint32_t k = 0; k < len; k++) d[k] = nrm[s[k] - GZ_DQ_FIRST];
        }
        else if (nnz) {
            uint8_t *q = J.qual + at_qual + (e2 >> 32), *r = J.runs + at_runs + (uint32_t)e2;
            uint64_t run = run_before;
            bool first = true;
            for (uint32_t k = 0; k < len; k++) {
                const uint32_t v = nrm[s[k] - GZ_DQ_FIRST];
                if (!v) { if (!first) run++; continue; }
                if (first) { run = run_before; first = false; }
                if (run) { d_dq_put_run (r, run); r += d_dq_run_bytes (run); }
                else *q++ = (uint8_t)no_doms;
                *q++ = (uint8_t)v;
                run = 0;
            }
        }
        pos += tot; if (mx > last_nz) last_nz = mx;
        at_qual += t2 >> 32; at_runs += (uint32_t)t2; at_divr += t3 >> 32; at_mplx += (uint32_t)t3;
        if (base + 256 >= J.n) { uint32_t *sh = (uint32_t *)gz_lds; __syncthreads (); if (J.n - 1 - base == (uint32_t)tid) sh[600] = len; __syncthreads (); last_len = sh[600]; }
    }
    // ---- the run the VBlock ends with (:468-480), "all diverse" (:490-494), results
    if (!tid) {
        const uint64_t runlen = pos - (uint64_t)(last_nz + 1);
        if (runlen && (at_runs || runlen < last_len)) { d_dq_put_run (J.runs + at_runs, runlen); at_runs += d_dq_run_bytes (runlen); J.qual[at_qual++] = (uint8_t)no_doms; }
        uint32_t all_diverse = 0;
        if (!at_qual) { J.qual[at_qual++] = 'X'; all_diverse = 1; }
        J.res->qual_len = at_qual; J.res->runs_len = at_runs; J.res->mplx_len = at_mplx; J.res->divr_len = at_divr;
        J.res->num_doms = num_doms; J.res->num_norm_qs = num_norm; J.res->has_diverse = misc[97]; J.res->all_diverse = all_diverse;
        J.res->status = misc[96] ? GZ_ST_CORRUPT : GZ_ST_OK;
    }
}

// codec_domq_qual_data_is_a_fit_for_domq (:69-134): the first (up to) 10 lines, 2500 / lines bytes of each: more than half of
// them must have a score that fills more than half of the sample. One thread per VBlock (the sample is 2.5 KB).
struct GzdDomqFit { const uint8_t *text; const uint32_t *off, *len; uint32_t n; uint32_t *fit; };
__global__ void k_domq_fit (const GzdDomqFit *jobs, uint32_t n_jobs)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_jobs) return;
    const GzdDomqFit &J = jobs[j];
    uint32_t sampled = J.n < 10 ? J.n : 10, tested = 0, with_dom = 0;
    const uint32_t per_line = sampled ? 2500 / sampled : 2500;
    for (uint32_t i = 0; i < sampled; i++) {
        const uint32_t l = J.len[i] < per_line ? J.len[i] : per_line;
        if (!l) { if (sampled < J.n) { sampled++; continue; } else break; }
        uint16_t h[GZ_DQ_N];
        for (int q = 0; q < GZ_DQ_N; q++) h[q] = 0;
        const uint8_t *s = J.text + J.off[i];
        bool dom = false;
        for (uint32_t k = 0; k < l; k++) { const uint32_t c = s[k] - GZ_DQ_FIRST; if (c < GZ_DQ_N && ++h[c] * 2u > l) dom = true; }   // (a count only grows: once past half, it stays past half)
        with_dom += dom; tested++;
    }
    *J.fit = tested && 100.0 * (double)with_dom / (double)tested > 50.0;
}
