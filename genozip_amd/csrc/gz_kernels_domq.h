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
// Everything the reference does line after line with carried state (the run counter that only a non-dominant score resets,
// the output cursors) is restated as per-line facts + scans over the lines, so that all lines of all VBlocks of a call are
// worked on at once:
//   k_domq_lines    a wave per line: histogram, dominant score, diverse?  -> per-VBlock histograms (LDS per workgroup, then atomics)
//   k_domq_tables   a workgroup per VBlock: compacted dominant scores, rank tables, the denormalisation table
//   k_domq_measure  a wave per line: normalised; number of non-dominant scores, leading / trailing run, bytes of the runs and
//                   markers inside the line
//   k_domq_scan     a workgroup per VBlock: positions, the run before every line's first non-dominant score (its position minus
//                   the position of the last one before it: a max-scan), output offsets of every line; the VBlock's last run
//   k_domq_write    a wave per line: the four streams
#pragma once
#include "gz_device.h"
#include "gz_devutil.h"
#include "gz_kernels_seg.h"

#define GZ_DQ_FIRST 32
#define GZ_DQ_N     95
#define GZ_DQ_HIST  (GZ_DQ_N * GZ_DQ_N)
#define GZ_DQ_MISC  128                              // u32 behind the histograms: [0..95) lines_with_dom [96] bad [97] has_diverse [98] num_norm [99] num_doms [100..124) dom_to_cdom bytes
#define GZ_DQ_LINES_PER_WG 256
#define GZ_DOMQ_LDS (4096 + GZ_DQ_HIST * 4 + 4 * GZ_DQ_N * 4 + GZ_DQ_MISC * 4)

struct GzdDomq {
    const uint8_t *text; const uint32_t *off, *len; uint32_t n;
    uint8_t *qual, *runs, *mplx, *divr;            // outputs: capacities 2 * bytes + 16, bytes + bytes / 254 + 16, n + 16, bytes + 16
    uint8_t *line_dom;                             // scratch [n]
    uint8_t *normalize;                            // scratch [95][95], indexed by the (uncompacted) dominant score
    uint32_t *hist;                                // scratch [95 * 95 + GZ_DQ_MISC], zero on entry
    uint32_t *rec;                                 // scratch [6][n]: L (length if not diverse), trail, lead, nnz, inner qual bytes, inner run bytes
    uint32_t *lo;                                  // scratch [5][n]: offsets of the line in qual / runs / divr / mplx, the run before its first non-dominant score
    GzDomqResult *res;
    const uint32_t *only_if;                       // optional: nothing is done (and res is left alone) when this device word is 0
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

__device__ static inline void d_dq_put_run_g (uint8_t *dst, uint64_t r)               // (the same through a GLOBAL pointer)
{
    while (r) { const uint32_t sub = r < 254 ? (uint32_t)r : 254; gz_stg_u8 (dst++, r <= 254 ? sub : 255u); r -= sub; }
}

// sum / exclusive prefix sum of a 64-bit value over the wave
__device__ static inline uint64_t d_wave_sum_u64 (uint64_t v, int lane)
{
    for (int m = 32; m; m >>= 1) {
        const uint32_t lo = (uint32_t)__shfl ((int)(uint32_t)v, lane ^ m), hi = (uint32_t)__shfl ((int)(uint32_t)(v >> 32), lane ^ m);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}
__device__ static inline uint64_t d_wave_excl_u64 (uint64_t v, int lane, uint64_t *total)
{
    uint64_t inc = v;
    for (int d = 1; d < 64; d <<= 1) {
        const int src = lane >= d ? lane - d : lane;
        const uint32_t lo = (uint32_t)__shfl ((int)(uint32_t)inc, src), hi = (uint32_t)__shfl ((int)(uint32_t)(inc >> 32), src);
        if (lane >= d) inc += ((uint64_t)hi << 32) | lo;
    }
    *total = ((uint64_t)(uint32_t)__shfl ((int)(uint32_t)(inc >> 32), 63) << 32) | (uint32_t)__shfl ((int)(uint32_t)inc, 63);
    return inc - v;
}

// The three per-line kernels below work a wave through 64 lines, one after the other, and everything a line needs is a chain of trips
// to memory: its length and offset, then its bytes, then (measure / write) a table look-up per byte. Left like that a wave spends
// 4-5 us per line waiting (1.2 + 0.95 + 0.9 ms for the 150 MB of QUAL of a 1 M-read file, 5-10 % of what HBM allows, in front of the
// launch of the long QUAL streams). So: the NEXT line's first 256 bytes are requested before this line is worked on and the length /
// offset of the line after that (two register sets taking turns: a loaded value cannot even be moved without waiting for it), and the
// normalisation tables are read from LDS.
// (Loads always happen, from a clamped index, and through GLOBAL pointers: a load under a condition comes with an exec-mask branch and
//  a wait for everything before it, and a load through a generic pointer is a flat one that every LDS wait waits for as well.)
struct GzdDqMeta { uint32_t len, off, ld, x[6]; };     // (ld, x: what measure / write need from the per-line tables - KIND 1 / 2)
template <int KIND> __device__ static __forceinline__ GzdDqMeta d_dq_meta (const GzdDomq &J, uint32_t i, uint32_t end)
{
    const uint32_t ic = i < end ? i : end - 1;                                  // (end >= 1)
    GzdDqMeta m; m.len = gz_ldg_u32 (J.len + ic); m.off = gz_ldg_u32 (J.off + ic);
    if (KIND >= 1) m.ld = gz_ldg_u8 (J.line_dom + ic);                          // (an empty line's is never written: only looked at if the line has bytes)
    if (KIND >= 2) {
        const size_t n = J.n; const uint32_t *o = J.lo + ic;
        m.x[0] = gz_ldg_u32 (o); m.x[1] = gz_ldg_u32 (o + n); m.x[2] = gz_ldg_u32 (o + 2 * n); m.x[3] = gz_ldg_u32 (o + 3 * n); m.x[4] = gz_ldg_u32 (o + 4 * n);
        m.x[5] = gz_ldg_u32 (J.rec + 3 * n + ic);
    }
    return m;
}
__device__ static __forceinline__ void d_dq_fetch (const uint8_t *text, const GzdDqMeta &m, int lane, uint32_t (&b)[4])
{
    #pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t k = (uint32_t)(q * 64 + lane);
        b[q] = gz_ldg_u8 (text + (m.len ? m.off + (k < m.len ? k : m.len - 1) : 0u));   // (beyond the line: its last byte again, nobody looks)
    }
}
// column of a score in the normalisation tables (a score outside '!' .. '~' - k_domq_lines has flagged the VBlock - stays inside the table)
__device__ static __forceinline__ uint32_t d_dq_index (uint32_t c) { return c - GZ_DQ_FIRST < GZ_DQ_N ? c - GZ_DQ_FIRST : 0u; }

// CHUNK (c0, byte of position c0 + lane - garbage beyond the line) over the line's 64-byte chunks: the first four from the registers in
// straight-line code - a load anywhere in there and its wait would also wait for the NEXT line's bytes, which are requested after this
// line's and return in order -, what a line has beyond 256 bytes from memory, in a loop of its own
#define GZ_DQ_FOR_CHUNKS(s, len, lane, b, CHUNK) do { \
        if ((len) > 0)   CHUNK (0u,   (b)[0]); \
        if ((len) > 64)  CHUNK (64u,  (b)[1]); \
        if ((len) > 128) CHUNK (128u, (b)[2]); \
        if ((len) > 192) CHUNK (192u, (b)[3]); \
        for (uint32_t c0_ = 256; c0_ < (len); c0_ += 64) { const uint32_t k_ = c0_ + (uint32_t)(lane); CHUNK (c0_, gz_ldg_u8 ((s) + (k_ < (len) ? k_ : (len) - 1))); } \
    } while (0)
// runs LINE (i, meta, bytes) over lines base + wave, + 4, ... < end with the look-ahead described above
#define GZ_DQ_FOR_LINES(KIND, J, base, wave, end, lane, LINE) do { \
        uint32_t i_ = (base) + (wave); \
        GzdDqMeta m0_ = d_dq_meta<KIND> (J, i_, end), m1_ = d_dq_meta<KIND> (J, i_ + 4, end), m2_, m3_; \
        uint32_t bA_[4], bB_[4]; \
        d_dq_fetch (J.text, m0_, lane, bA_); \
        while (i_ < (end)) { \
            d_dq_fetch (J.text, m1_, lane, bB_); m2_ = d_dq_meta<KIND> (J, i_ + 8, end); \
            LINE (i_, m0_, bA_); \
            i_ += 4; if (i_ >= (end)) break; \
            d_dq_fetch (J.text, m2_, lane, bA_); m3_ = d_dq_meta<KIND> (J, i_ + 8, end); \
            LINE (i_, m1_, bB_); \
            i_ += 4; \
            m0_ = m2_; m1_ = m3_; \
        } \
    } while (0)

// ---- 1. every line's dominant score and the histograms per dominant score (codec_domq.c:139-176)
// grid (lines / 256, VBlocks), 256 threads, GZ_DOMQ_LDS bytes: a wave per line, 64 lines per wave
__global__ void __launch_bounds__(256) k_domq_lines (const GzdDomq *jobs)
{
    const GzdDomq J = jobs[blockIdx.y];
    if (J.only_if && !*J.only_if) return;
    const uint32_t base = blockIdx.x * GZ_DQ_LINES_PER_WG;
    if (base >= J.n) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    uint32_t *hist  = (uint32_t *)(gz_lds + 4096);            // [95][95] per dominant score, this workgroup's lines
    uint32_t *whist = hist + GZ_DQ_HIST;                      // [4][95] the line a wave is looking at
    uint32_t *misc  = whist + 4 * GZ_DQ_N;
    for (int i = tid; i < GZ_DQ_HIST + 4 * GZ_DQ_N + GZ_DQ_MISC; i += 256) hist[i] = 0;
    __syncthreads ();
    const uint32_t end = base + GZ_DQ_LINES_PER_WG < J.n ? base + GZ_DQ_LINES_PER_WG : J.n;
    uint32_t *h = whist + wave * GZ_DQ_N;
    auto line = [&] (uint32_t i, const GzdDqMeta &m, const uint32_t (&b)[4]) {
        const uint32_t len = m.len;
        if (!len) return;                                                       // (wave-uniform)
        const uint8_t *s = J.text + m.off;
        auto chunk = [&] (uint32_t c0, uint32_t c) {
            if (c0 + lane < len) { if (c < GZ_DQ_FIRST || c > 126) misc[96] = 1; else atomicAdd (&h[c - GZ_DQ_FIRST], 1u); }
        };
        GZ_DQ_FOR_CHUNKS (s, len, lane, b, chunk);
        gz_wave_sync ();
        // the largest count, the higher score among equals (:152-157): lanes hold scores lane and lane + 64
        const uint32_t c0 = h[lane], c1 = lane + 64 < GZ_DQ_N ? h[lane + 64] : 0;
        uint32_t best = c1 >= c0 && lane + 64 < GZ_DQ_N ? c1 : c0, bq = c1 >= c0 && lane + 64 < GZ_DQ_N ? (uint32_t)lane + 64 : (uint32_t)lane;
        for (int mm = 32; mm; mm >>= 1) {
            const uint32_t ob = (uint32_t)__shfl ((int)best, lane ^ mm), oq = (uint32_t)__shfl ((int)bq, lane ^ mm);
            if (ob > best || (ob == best && oq > bq)) { best = ob; bq = oq; }
        }
        const bool diverse = 100u * best / len < 85u;                           // DOMQ_THRESHOLD
        if (c0) atomicAdd (&hist[bq * GZ_DQ_N + lane], c0);
        if (c1) atomicAdd (&hist[bq * GZ_DQ_N + lane + 64], c1);
        if (!lane) { gz_stg_u8 (J.line_dom + i, bq | (diverse ? 0x80u : 0u)); atomicAdd (&misc[bq], 1u); if (diverse) misc[97] = 1; }
        gz_wave_sync ();
        h[lane] = 0; if (lane + 64 < GZ_DQ_N) h[lane + 64] = 0;
        gz_wave_sync ();
    };
    GZ_DQ_FOR_LINES (0, J, base, wave, end, lane, line);
    __syncthreads ();
    for (int i = tid; i < GZ_DQ_HIST; i += 256) { const uint32_t v = hist[i]; if (v) atomicAdd (&J.hist[i], v); }
    if (tid < 98) { const uint32_t v = misc[tid]; if (v) { if (tid < GZ_DQ_N) atomicAdd (&J.hist[GZ_DQ_HIST + tid], v); else J.hist[GZ_DQ_HIST + tid] = 1; } }
}

// ---- 2. tables (:178-249): compact the dominant scores in ascending order; within one, rank the scores by count,
//         descending, equal counts in ascending score order (the reference's qsort is glibc's stable merge sort at this size)
// grid (VBlocks), 128 threads, 64 bytes of LDS
__global__ void __launch_bounds__(128) k_domq_tables (const GzdDomq *jobs)
{
    const GzdDomq J = jobs[blockIdx.x];
    if (J.only_if && !*J.only_if) return;
    const int tid = threadIdx.x;
    const uint32_t *hist = J.hist; uint32_t *misc = J.hist + GZ_DQ_HIST;
    uint8_t *dom_to_cdom = (uint8_t *)(misc + 100);
    uint32_t &s_norm = *(uint32_t *)gz_lds;
    if (!tid) { uint32_t nd = 0; for (int q = 0; q < GZ_DQ_N; q++) if (misc[q]) dom_to_cdom[q] = (uint8_t)nd++; misc[99] = nd; s_norm = 0; }
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
            if (!tid) atomicMax (&s_norm, nz);
        }
    }
    __threadfence_block ();
    __syncthreads ();
    const uint32_t num_norm = s_norm;
    if (!tid) misc[98] = num_norm;
    if (tid < GZ_DQ_N && misc[tid]) {                         // denormalisation rows, compacted to num_norm columns
        const uint32_t cd = dom_to_cdom[tid];
        for (uint32_t r = 0; r < num_norm; r++) J.res->denorm[cd * num_norm + r] = 0;
        for (int q = 0; q < GZ_DQ_N; q++) if (hist[tid * GZ_DQ_N + q]) J.res->denorm[cd * num_norm + J.normalize[tid * GZ_DQ_N + q]] = (uint8_t)(q + GZ_DQ_FIRST);
    }
}

// ---- 3a. a line on its own. grid (lines / 256, VBlocks), 256 threads, GZ_DQ_NORM_LDS bytes: a wave per line, 64 scores a step
#define GZ_DQ_NORM_LDS (GZ_DQ_HIST + 64)
__device__ static __forceinline__ const uint8_t *d_dq_norm_to_lds (const GzdDomq &J, int tid)
{
    uint8_t *t = gz_lds;
    for (int i = tid; i < GZ_DQ_HIST; i += 256) t[i] = J.normalize[i];
    __syncthreads ();
    return t;
}
__global__ void __launch_bounds__(256) k_domq_measure (const GzdDomq *jobs)
{
    const GzdDomq J = jobs[blockIdx.y];
    if (J.only_if && !*J.only_if) return;
    const uint32_t base = blockIdx.x * GZ_DQ_LINES_PER_WG;
    if (base >= J.n) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint8_t *norm = d_dq_norm_to_lds (J, tid);
    const uint32_t end = base + GZ_DQ_LINES_PER_WG < J.n ? base + GZ_DQ_LINES_PER_WG : J.n;
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0;
    auto line = [&] (uint32_t i, const GzdDqMeta &m, const uint32_t (&b)[4]) {
        const uint32_t len = m.len;
        const uint32_t ld = m.ld;
        uint32_t L = 0, trail = 0, lead = 0, nnz = 0, inner_q = 0, inner_r = 0;
        if (len && !(ld & 0x80)) {
            const uint8_t *nrm = norm + (ld & 0x7f) * GZ_DQ_N;
            const uint8_t *s = J.text + m.off;
            int64_t last = -1;                                                  // (wave-uniform) position of the last non-dominant score so far
            uint64_t acc = 0;                                                   // per lane: markers << 32 | run bytes
            auto chunk = [&] (uint32_t c0, uint32_t c) {
                const uint32_t k = c0 + lane;
                const uint32_t v = k < len ? nrm[d_dq_index (c)] : 0;
                const uint64_t mask = __ballot (v != 0);
                if (v) {
                    const uint64_t mb = mask & below;
                    const int64_t prev = mb ? (int64_t)c0 + 63 - __builtin_clzll (mb) : last;
                    if (prev >= 0) { const uint32_t run = (uint32_t)((int64_t)k - prev - 1); acc += run ? d_dq_run_bytes (run) : (1ull << 32); }
                }
                if (mask) {
                    if (last < 0) lead = c0 + (uint32_t)__builtin_ctzll (mask);
                    last = (int64_t)c0 + 63 - __builtin_clzll (mask);
                    nnz += (uint32_t)__popcll (mask);
                }
            };
            GZ_DQ_FOR_CHUNKS (s, len, lane, b, chunk);
            acc = d_wave_sum_u64 (acc, lane);
            L = len; trail = nnz ? len - 1 - (uint32_t)last : len;
            inner_q = nnz + (uint32_t)(acc >> 32); inner_r = (uint32_t)acc;
        }
        if (!lane) {
            uint32_t *r = J.rec + i; const size_t n = J.n;
            gz_stg_u32 (r, L); gz_stg_u32 (r + n, trail); gz_stg_u32 (r + 2 * n, lead); gz_stg_u32 (r + 3 * n, nnz); gz_stg_u32 (r + 4 * n, inner_q); gz_stg_u32 (r + 5 * n, inner_r);
        }
    };
    GZ_DQ_FOR_LINES (1, J, base, wave, end, lane, line);
}

// ---- 3b. the lines in order. grid (VBlocks), 1024 threads: each of the 16 waves owns a contiguous sixteenth of the VBlock's lines and
// walks it 64 lines at a time, lane = line (every table read and written in whole lines of memory; the first version had thread t walk
// lines [t T, (t + 1) T) on its own - 256 threads each touching a different line of memory per load, 0.95 ms for a 38 000-line VBlock).
// What runs through the lines - the position in the concatenation of the non-diverse lines, the position of the last non-dominant
// score so far, the output offsets - is a scan: inside the 64 lines by shuffles, from tile to tile in wave-uniform registers, from
// wave to wave through LDS (three passes: totals per wave; bytes per wave; the offsets).
#define GZ_DQ_SCAN_NT 1024
__device__ static __forceinline__ int64_t d_wave_shfl_i64 (int64_t v, int src)
{
    const uint32_t lo = (uint32_t)__shfl ((int)(uint32_t)(uint64_t)v, src), hi = (uint32_t)__shfl ((int)(uint32_t)((uint64_t)v >> 32), src);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
// exclusive max-scan over the wave (identity -1); *total = the maximum over all lanes
__device__ static __forceinline__ int64_t d_wave_exclmax_i64 (int64_t v, int lane, int64_t *total)
{
    int64_t inc = v;
    for (int d = 1; d < 64; d <<= 1) { const int64_t o = d_wave_shfl_i64 (inc, lane >= d ? lane - d : lane); if (lane >= d && o > inc) inc = o; }
    *total = d_wave_shfl_i64 (inc, 63);
    const int64_t left = d_wave_shfl_i64 (inc, lane ? lane - 1 : 0);
    return lane ? left : -1;
}
__global__ void __launch_bounds__(GZ_DQ_SCAN_NT) k_domq_scan (const GzdDomq *jobs)
{
    const GzdDomq J = jobs[blockIdx.x];
    if (J.only_if && !*J.only_if) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = GZ_DQ_SCAN_NT / 64;
    const size_t n = J.n;
    const uint32_t *rL = J.rec, *rT = J.rec + n, *rLead = J.rec + 2 * n, *rN = J.rec + 3 * n, *rQ = J.rec + 4 * n, *rR = J.rec + 5 * n;
    const uint32_t *misc = J.hist + GZ_DQ_HIST;
    uint64_t *shA = (uint64_t *)gz_lds, *shB = shA + NW;      // per wave: two sums
    int64_t *shM = (int64_t *)(shB + NW);                     // per wave: a maximum
    const uint32_t tiles = (J.n + 63) / 64, tpw = (tiles + NW - 1) / NW;
    const uint32_t w0 = (uint64_t)wave * tpw * 64 < J.n ? wave * tpw * 64 : J.n, w1 = (uint64_t)w0 + tpw * 64 < J.n ? w0 + tpw * 64 : J.n;
    // ---- pass 1: the length and the last non-dominant score of every wave's stretch
    uint64_t sumL = 0; int64_t lastnz = -1;                    // (relative to the start of the stretch)
    for (uint32_t t0 = w0; t0 < w1; t0 += 64) {
        const bool in = t0 + lane < w1; const uint32_t i = in ? t0 + lane : w1 - 1;
        uint32_t L = gz_ldg_u32 (rL + i), N = gz_ldg_u32 (rN + i); const uint32_t T = gz_ldg_u32 (rT + i);
        if (!in) { L = 0; N = 0; }
        uint64_t tot; const uint64_t ex = d_wave_excl_u64 (L, lane, &tot);
        int64_t mx; (void)d_wave_exclmax_i64 (N ? (int64_t)(sumL + ex + L - 1 - T) : -1, lane, &mx);
        if (mx > lastnz) lastnz = mx;
        sumL += tot;
    }
    if (!lane) { shA[wave] = sumL; shM[wave] = lastnz; }
    __syncthreads ();
    uint64_t pos0 = 0, tot_pos = 0; int64_t before0 = -1, last_nz = -1;
    for (int w = 0; w < NW; w++) {
        if (w == wave) { pos0 = tot_pos; before0 = last_nz; }
        if (shM[w] >= 0) last_nz = (int64_t)tot_pos + shM[w];
        tot_pos += shA[w];
    }
    __syncthreads ();
    // ---- passes 2 and 3: the bytes of every line in the four streams; their sums per wave, then every line's offsets
    uint64_t at_qr0 = 0, at_dm0 = 0, tot_qr = 0, tot_dm = 0;
    for (int pass = 0; pass < 2; pass++) {
        uint64_t pos = pos0, at_qr = at_qr0, at_dm = at_dm0; int64_t before = before0;
        for (uint32_t t0 = w0; t0 < w1; t0 += 64) {
            const bool in = t0 + lane < w1; const uint32_t i = in ? t0 + lane : w1 - 1;
            uint32_t L = gz_ldg_u32 (rL + i), N = gz_ldg_u32 (rN + i), len = gz_ldg_u32 (J.len + i);
            const uint32_t T = gz_ldg_u32 (rT + i), lead = gz_ldg_u32 (rLead + i), Q = gz_ldg_u32 (rQ + i), R = gz_ldg_u32 (rR + i), dom = gz_ldg_u8 (J.line_dom + i);
            if (!in) { L = 0; N = 0; len = 0; }
            const bool diverse = len && (dom & 0x80);
            uint64_t totL; const uint64_t exL = d_wave_excl_u64 (L, lane, &totL);
            const uint64_t my_pos = pos + exL;
            int64_t mx; const int64_t exm = d_wave_exclmax_i64 (N ? (int64_t)(my_pos + L - 1 - T) : -1, lane, &mx);
            const int64_t my_before = exm > before ? exm : before;
            const uint64_t run_before = N ? my_pos + lead - (uint64_t)(my_before + 1) : 0;
            const uint64_t q_bytes = in ? Q + (N && !run_before ? 1 : 0) : 0, r_bytes = in ? R + (N ? d_dq_run_bytes (run_before) : 0) : 0;
            const uint64_t qr = (q_bytes << 32) | r_bytes, dm = ((uint64_t)(diverse ? len : 0) << 32) | (len ? 1u : 0u);
            uint64_t tqr, tdm; const uint64_t eqr = d_wave_excl_u64 (qr, lane, &tqr), edm = d_wave_excl_u64 (dm, lane, &tdm);
            if (pass && in) {
                uint32_t *o = J.lo + i; const uint64_t a = at_qr + eqr, d = at_dm + edm;
                gz_stg_u32 (o, (uint32_t)(a >> 32)); gz_stg_u32 (o + n, (uint32_t)a); gz_stg_u32 (o + 2 * n, (uint32_t)(d >> 32)); gz_stg_u32 (o + 3 * n, (uint32_t)d);
                gz_stg_u32 (o + 4 * n, (uint32_t)run_before);
            }
            at_qr += tqr; at_dm += tdm; pos += totL;
            if (mx > before) before = mx;
        }
        if (!pass) {
            if (!lane) { shA[wave] = at_qr; shB[wave] = at_dm; }
            __syncthreads ();
            for (int w = 0; w < NW; w++) {
                if (w == wave) { at_qr0 = tot_qr; at_dm0 = tot_dm; }
                tot_qr += shA[w]; tot_dm += shB[w];
            }
        }
    }
    // ---- the run the VBlock ends with (:468-480), "all diverse" (:490-494), results
    if (!tid) {
        uint64_t at_qual = tot_qr >> 32, at_runs = (uint32_t)tot_qr;
        const uint32_t no_doms = misc[98], last_len = J.n ? J.len[J.n - 1] : 0;
        const uint64_t runlen = tot_pos - (uint64_t)(last_nz + 1);
        if (runlen && (at_runs || runlen < last_len)) { d_dq_put_run (J.runs + at_runs, runlen); at_runs += d_dq_run_bytes (runlen); J.qual[at_qual++] = (uint8_t)no_doms; }
        uint32_t all_diverse = 0;
        if (!at_qual) { J.qual[at_qual++] = 'X'; all_diverse = 1; }
        J.res->qual_len = at_qual; J.res->runs_len = at_runs; J.res->mplx_len = (uint32_t)tot_dm; J.res->divr_len = tot_dm >> 32;
        J.res->num_doms = misc[99]; J.res->num_norm_qs = no_doms; J.res->has_diverse = misc[97]; J.res->all_diverse = all_diverse;
        J.res->status = misc[96] ? GZ_ST_CORRUPT : GZ_ST_OK;
    }
}

// ---- 3c. the streams. grid (lines / 256, VBlocks), 256 threads, GZ_DQ_NORM_LDS bytes: a wave per line
__global__ void __launch_bounds__(256) k_domq_write (const GzdDomq *jobs)
{
    const GzdDomq J = jobs[blockIdx.y];
    if (J.only_if && !*J.only_if) return;
    const uint32_t base = blockIdx.x * GZ_DQ_LINES_PER_WG;
    if (base >= J.n) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint8_t *norm = d_dq_norm_to_lds (J, tid);
    const uint32_t end = base + GZ_DQ_LINES_PER_WG < J.n ? base + GZ_DQ_LINES_PER_WG : J.n;
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0;
    const size_t n = J.n;
    const uint32_t *misc = J.hist + GZ_DQ_HIST;
    const uint8_t *dom_to_cdom = (const uint8_t *)(misc + 100);
    const uint32_t no_doms = misc[98];
    auto line = [&] (uint32_t i, const GzdDqMeta &m, const uint32_t (&b)[4]) {
        const uint32_t len = m.len;
        if (!len) return;
        // (everything the line needs from the per-line tables at once: five offsets, the count of non-dominant scores)
        const uint32_t o_q = m.x[0], o_r = m.x[1], o_d = m.x[2], o_m = m.x[3], o_run = m.x[4], nnz = m.x[5], ld = m.ld;
        const bool diverse = ld & 0x80;
        const uint8_t *nrm = norm + (ld & 0x7f) * GZ_DQ_N;
        const uint8_t *s = J.text + m.off;
        if (!lane) gz_stg_u8 (J.mplx + o_m, dom_to_cdom[ld & 0x7f] | (diverse ? 0x80u : 0u));
        if (diverse) {
            uint8_t *d = J.divr + o_d;
            auto chunk = [&] (uint32_t c0, uint32_t c) { const uint32_t k = c0 + lane; if (k < len) gz_stg_u8 (d + k, nrm[d_dq_index (c)]); };
            GZ_DQ_FOR_CHUNKS (s, len, lane, b, chunk);
            return;
        }
        if (!nnz) return;                                                       // only the dominant score: the run goes on
        uint8_t *q = J.qual + o_q, *r = J.runs + o_r;
        const uint64_t run_before = o_run;
        int64_t last = -1;
        auto chunk = [&] (uint32_t c0, uint32_t c) {
            const uint32_t k = c0 + lane;
            const uint32_t v = k < len ? nrm[d_dq_index (c)] : 0;
            const uint64_t mask = __ballot (v != 0);
            if (!mask) return;                                                  // (wave-uniform)
            uint64_t run = 0;
            if (v) {
                const uint64_t mb = mask & below;
                const int64_t prev = mb ? (int64_t)c0 + 63 - __builtin_clzll (mb) : last;
                run = prev >= 0 ? (uint64_t)((int64_t)k - prev - 1) : run_before;
            }
            const uint64_t mine = v ? ((uint64_t)(run ? 1u : 2u) << 32) | d_dq_run_bytes (run) : 0;
            uint64_t tot;
            const uint64_t ex = d_wave_excl_u64 (mine, lane, &tot);
            if (v) {
                uint8_t *qq = q + (ex >> 32);
                if (run) d_dq_put_run_g (r + (uint32_t)ex, run); else gz_stg_u8 (qq++, no_doms);
                gz_stg_u8 (qq, v);
            }
            q += tot >> 32; r += (uint32_t)tot;
            last = (int64_t)c0 + 63 - __builtin_clzll (mask);
        };
        GZ_DQ_FOR_CHUNKS (s, len, lane, b, chunk);
    };
    GZ_DQ_FOR_LINES (2, J, base, wave, end, lane, line);
}

// codec_domq_qual_data_is_a_fit_for_domq (:69-134): the first (up to) 10 lines, 2500 / lines bytes of each: more than half of
// them must have a score that fills more than half of the sample. A wave per VBlock (it sits in front of the long streams' launch):
// 64 bytes of a line a step into an LDS histogram, then "is any count past half" over the 95 scores.
// grid (VBlocks), 64 threads, 512 bytes of LDS
struct GzdDomqFit { const uint8_t *text; const uint32_t *off, *len; uint32_t n; uint32_t *fit; };
__global__ void __launch_bounds__(64) k_domq_fit (const GzdDomqFit *jobs, uint32_t n_jobs)
{
    if (blockIdx.x >= n_jobs) return;
    const GzdDomqFit J = jobs[blockIdx.x];
    const int lane = threadIdx.x;
    uint32_t *h = (uint32_t *)gz_lds;                                           // [95] (+ room)
    uint32_t sampled = J.n < 10 ? J.n : 10, tested = 0, with_dom = 0;
    const uint32_t per_line = sampled ? 2500 / sampled : 2500;
    for (uint32_t i = 0; i < sampled; i++) {
        const uint32_t l = J.len[i] < per_line ? J.len[i] : per_line;
        if (!l) { if (sampled < J.n) { sampled++; continue; } else break; }     // (an empty line is not sampled: the next one is, :84-86)
        h[lane] = 0; h[lane + 64] = 0;
        gz_wave_sync ();
        const uint8_t *s = J.text + J.off[i];
        for (uint32_t k = lane; k < l; k += 64) { const uint32_t c = (uint32_t)s[k] - GZ_DQ_FIRST; if (c < GZ_DQ_N) atomicAdd (&h[c], 1u); }
        gz_wave_sync ();
        const bool mine = h[lane] * 2u > l || (lane + 64 < GZ_DQ_N && h[lane + 64] * 2u > l);
        with_dom += __ballot (mine) != 0; tested++;
        gz_wave_sync ();
    }
    if (!lane) *J.fit = tested && 100.0 * (double)with_dom / (double)tested > 50.0;
}
