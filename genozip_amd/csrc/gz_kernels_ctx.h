// gz_kernels_ctx.h -- context-engine kernels: b250 generation (src/b250.c:202-267), local byte-order / interlace
// (src/buffer.c:336-350, src/context.h:99-101), matrix transpose (src/dyn_int.c:45-132) and adler32.
#pragma once
#include "gz_device.h"
#include "gz_devutil.h"

// ======================================================================================================
// b250_zip_generate
//
// The seg-time b250 stream is a sequence of 1..4 byte little-endian VARL words whose length tag sits in the LAST
// byte, so it can only be parsed backwards (the reference walks it backwards, serially). Parallel version: cut the
// stream into 64-byte chunks counted from the end. Whatever the words look like, the backward chain of "last bytes"
// enters a chunk at one of its top 4 positions; every thread walks its chunk for all 4 possible entries, recording
// where it would leave and how many words it would see; one thread then threads the real chain through the chunk
// table; finally every thread re-walks its chunk knowing the real entry and scatters converted word indices.
// A second pass applies ONE_UP (it compares *converted* neighbours, b250.c:236,251) and re-encodes big-endian with
// the tag first; output positions come from a workgroup prefix sum of the encoded lengths.
// ======================================================================================================
#define GZ_B250_CHUNK 64

struct GzdB250Job {
    const uint8_t *seg; uint32_t seg_len; const uint32_t *seg_len_dev;
    uint32_t ol_nodes_len; const int32_t *node2word; uint32_t n_new_nodes;
    uint8_t *out; uint32_t *out_len_dev;
    int32_t *wi;             // scratch: seg_len + 1 words
    uint32_t *chunk_tab;     // scratch: 5 words per chunk (count for entry 0..3, packed exits) + 2 per chunk (entry, base)
    int32_t *status_dev;     // optional
    const uint8_t *r1; const uint32_t *r1_len_dev;   // optional: R1's b250 of the same context
};

__device__ static inline int d_varl_len (uint8_t tag) { return !(tag >> 7) ? 1 : (tag >> 6) == 2 ? 2 : (tag >> 5) == 6 ? 3 : 4; }

// value of the seg-format word whose last byte is at index `last` (b250.c:60-79)
__device__ static inline int32_t d_seg_value (const uint8_t *seg, int64_t last, int n)
{
    uint32_t v = 0;
    for (int k = 0; k < n; k++) v = (v << 8) | seg[last - k];
    switch (n) {
        case 1:  return (int32_t)v;
        case 2:  return v == 0xBFFE ? -3 : v == 0xBFFF ? -4 : (int32_t)(v & 0x3fff) + 127;
        case 3:  return (int32_t)(v & 0x1fffff) + 16509;
        default: return (int32_t)(v & 0x1fffffff);
    }
}

// PIZ-format code of a word index (b250.c:82-107): returns length, *code holds the bytes big-endian in its low bytes
__device__ static inline int d_varl_code (int32_t wi, uint32_t *code)
{
    if (wi == -2) { *code = 127;    return 1; }
    if (wi == -3) { *code = 0xBFFE; return 2; }
    if (wi == -4) { *code = 0xBFFF; return 2; }
    if (wi <= 126)     { *code = (uint32_t)wi; return 1; }
    if (wi <= 16508)   { *code = (2u << 14) | (uint32_t)(wi - 127); return 2; }
    if (wi <= 2113660) { *code = (6u << 21) | (uint32_t)(wi - 16509); return 3; }
    *code = (7u << 29) | (uint32_t)wi;
    return 4;
}

__global__ void __launch_bounds__(256) k_b250_generate (GzdB250Job *jobs)
{
    GzdB250Job &J = jobs[blockIdx.x];
    const int tid = threadIdx.x;
    uint32_t *sh = (uint32_t *)gz_lds;              // [0..255] per-thread sums, [256..] misc
    uint32_t seg_len = J.seg_len;
    if (J.seg_len_dev) { uint32_t v = *J.seg_len_dev; if (v < seg_len) seg_len = v; }
    if (!seg_len) { if (!tid) { *J.out_len_dev = 0; if (J.status_dev) *J.status_dev = GZ_ST_OK; } return; }
    const uint8_t *seg = J.seg;
    const uint32_t nchunks = (seg_len + GZ_B250_CHUNK - 1) / GZ_B250_CHUNK;
    uint32_t *ctab = J.chunk_tab;                   // [c*7 + e] count, [c*7+4] exits (4 x 8 bit, 0xff = malformed), [c*7+5] entry, [c*7+6] base

    // ---- A: speculative walks
    for (uint32_t c = tid; c < nchunks; c += 256) {
        int64_t hi = (int64_t)seg_len - (int64_t)c * GZ_B250_CHUNK, lo = hi - GZ_B250_CHUNK;
        if (lo < 0) lo = 0;
        uint32_t exits = 0;
        for (int e = 0; e < 4; e++) {
            int64_t p = hi - 1 - e;
            uint32_t cnt = 0;
            while (p >= lo) { cnt++; p -= d_varl_len (seg[p]); }
            uint32_t ex = (uint32_t)(lo - 1 - p);    // 0..3 ; for the first chunk of the stream only 0 is well-formed
            if (lo == 0 && ex != 0) ex = 0xff;
            ctab[c * 7 + e] = cnt;
            exits |= (ex & 0xff) << (8 * e);
        }
        ctab[c * 7 + 4] = exits;
    }
    __threadfence_block ();
    __syncthreads ();

    // ---- B: thread the real chain through the chunks
    if (!tid) {
        uint32_t state = 0, running = 0, ok = 1;
        for (uint32_t c = 0; c < nchunks; c++) {
            ctab[c * 7 + 5] = state;
            ctab[c * 7 + 6] = running;
            running += ctab[c * 7 + state];
            state = (ctab[c * 7 + 4] >> (8 * state)) & 0xff;
            if (state == 0xff) { ok = 0; break; }
        }
        sh[256] = running;       // number of words
        sh[257] = ok;
    }
    __syncthreads ();
    const uint32_t cnt = sh[256];
    if (!sh[257]) { if (!tid) { *J.out_len_dev = 0; if (J.status_dev) *J.status_dev = GZ_ST_CORRUPT; } return; }

    // ---- C: re-walk with the real entry, convert node -> word (context.h:109), store in reverse order
    int32_t *wi = J.wi;
    uint32_t bad = 0;
    for (uint32_t c = tid; c < nchunks; c += 256) {
        int64_t hi = (int64_t)seg_len - (int64_t)c * GZ_B250_CHUNK, lo = hi - GZ_B250_CHUNK;
        if (lo < 0) lo = 0;
        int64_t p = hi - 1 - ctab[c * 7 + 5];
        uint32_t k = ctab[c * 7 + 6];
        while (p >= lo) {
            int n = d_varl_len (seg[p]);
            if (p - n + 1 < 0) { bad = 1; break; }
            int32_t v = d_seg_value (seg, p, n);
            if (v >= 0 && (uint32_t)v >= J.ol_nodes_len) {
                uint32_t local = (uint32_t)v - J.ol_nodes_len;
                if (local >= J.n_new_nodes) { bad = 1; v = 0; }
                else v = J.node2word[local];
            }
            wi[k++] = v;
            p -= n;
        }
    }
    __threadfence_block ();
    __syncthreads ();

    // ---- D: ONE_UP + re-encode. Thread t owns words [t*per, (t+1)*per) in forward order.
    const bool one_up_ok = (uint64_t)J.n_new_nodes + J.ol_nodes_len > 1024;
    const uint32_t per = (cnt + 255) / 256;
    const uint32_t i0 = tid * per, i1 = i0 + per < cnt ? i0 + per : cnt;
    uint32_t mylen = 0;
    for (uint32_t i = i0; i < i1; i++) {
        int32_t cur = wi[cnt - 1 - i];
        if (one_up_ok && i && cur >= 0) { int32_t prev = wi[cnt - i]; if (prev >= 0 && cur == prev + 1) cur = -2; }
        uint32_t code;
        mylen += d_varl_code (cur, &code);
    }
    sh[tid] = mylen;
    sh[258 + tid] = bad;
    __syncthreads ();
    if (!tid) {
        uint32_t run = 0, anybad = 0;
        for (int t = 0; t < 256; t++) { uint32_t l = sh[t]; sh[t] = run; run += l; anybad |= sh[258 + t]; }
        sh[256] = run; sh[257] = anybad;
    }
    __syncthreads ();
    uint32_t o = sh[tid];
    for (uint32_t i = i0; i < i1; i++) {
        int32_t cur = wi[cnt - 1 - i];
        if (one_up_ok && i && cur >= 0) { int32_t prev = wi[cnt - i]; if (prev >= 0 && cur == prev + 1) cur = -2; }
        uint32_t code; int n = d_varl_code (cur, &code);
        for (int k = 0; k < n; k++) J.out[o + k] = (uint8_t)(code >> (8 * (n - 1 - k)));
        o += n;
    }
    if (!tid) {
        *J.out_len_dev = sh[257] ? 0 : sh[256];
        if (J.status_dev) *J.status_dev = sh[257] ? GZ_ST_CORRUPT : GZ_ST_OK;
    }
}

// ------------------------------------------------------------------------------------------------------
// The same for LONG b250s (a FORMAT/PL column of a VCF VBlock is 3 x 10^7 entries, 30-40 MB: 134 ms in one workgroup):
// the phases become kernels over all chunks. What made it one workgroup was the backward chain through the chunk
// table - but a chunk is a map {entry 0..3} -> (exit, words), and maps compose: every workgroup composes its 256
// chunks into one map of a "super chunk" (k_b250_walk), one thread chains the few thousand super chunks
// (k_b250_chain), every workgroup then chains its own 256 chunks from its known entry and converts (k_b250_convert);
// lengths, a scan over tiles of 256 words and the emission follow (k_b250_len / k_b250_scan / k_b250_emit).
// grid (super chunks, jobs) unless noted; jobs = the long ones only.
#define GZ_B250_SUPER 256                  // chunks per workgroup
#define GZ_B250_BIG   (128u * 1024u)       // seg bytes from which a b250 takes this path

struct GzdB250Big {
    GzdB250Job j;
    uint32_t *stab;          // per super chunk: [0..3] words for entry e, [4] exits (4 x 8 bit), [5] entry, [6] base
    uint64_t *tile;          // per 256 words: encoded bytes, then their exclusive scan
    uint32_t *info;          // [0] words, [1] ok, [2] bad node index seen
};

__device__ static inline uint32_t d_b250_seg_len (const GzdB250Job &J)
{
    uint32_t seg_len = J.seg_len;
    if (J.seg_len_dev) { const uint32_t v = *J.seg_len_dev; if (v < seg_len) seg_len = v; }
    return seg_len;
}

__global__ void __launch_bounds__(256) k_b250_walk (GzdB250Big *jobs)
{
    const GzdB250Big &B = jobs[blockIdx.y];
    const uint32_t seg_len = d_b250_seg_len (B.j);
    const uint32_t nchunks = (seg_len + GZ_B250_CHUNK - 1) / GZ_B250_CHUNK;
    const uint32_t c0 = blockIdx.x * GZ_B250_SUPER;
    if (c0 >= nchunks) return;
    const uint8_t *seg = B.j.seg;
    uint32_t *ctab = B.j.chunk_tab;
    uint32_t *sh = (uint32_t *)gz_lds;               // [c * 5 + e] words, [c * 5 + 4] exits of the workgroup's chunks
    const uint32_t c = c0 + threadIdx.x;
    if (c < nchunks) {
        int64_t hi = (int64_t)seg_len - (int64_t)c * GZ_B250_CHUNK, lo = hi - GZ_B250_CHUNK;
        if (lo < 0) lo = 0;
        uint32_t exits = 0;
        for (int e = 0; e < 4; e++) {
            int64_t p = hi - 1 - e;
            uint32_t cnt = 0;
            while (p >= lo) { cnt++; p -= d_varl_len (seg[p]); }
            uint32_t ex = (uint32_t)(lo - 1 - p);
            if (lo == 0 && ex != 0) ex = 0xff;
            ctab[c * 7 + e] = cnt; sh[threadIdx.x * 5 + e] = cnt;
            exits |= (ex & 0xff) << (8 * e);
        }
        ctab[c * 7 + 4] = exits; sh[threadIdx.x * 5 + 4] = exits;
    }
    __syncthreads ();
    if (threadIdx.x < 4) {                           // the super chunk's map for entry state e
        const uint32_t last = nchunks - c0 < GZ_B250_SUPER ? nchunks - c0 : GZ_B250_SUPER;
        uint32_t state = threadIdx.x, words = 0;
        for (uint32_t k = 0; k < last && state != 0xff; k++) { words += sh[k * 5 + state]; state = (sh[k * 5 + 4] >> (8 * state)) & 0xff; }
        B.stab[blockIdx.x * 7 + threadIdx.x] = words;
        sh[GZ_B250_SUPER * 5 + threadIdx.x] = state;
    }
    __syncthreads ();
    if (!threadIdx.x) B.stab[blockIdx.x * 7 + 4] = sh[GZ_B250_SUPER * 5] | (sh[GZ_B250_SUPER * 5 + 1] << 8) | (sh[GZ_B250_SUPER * 5 + 2] << 16) | (sh[GZ_B250_SUPER * 5 + 3] << 24);
}

// grid (jobs), one thread
__global__ void k_b250_chain (GzdB250Big *jobs)
{
    const GzdB250Big &B = jobs[blockIdx.x];
    const uint32_t seg_len = d_b250_seg_len (B.j);
    const uint32_t nchunks = (seg_len + GZ_B250_CHUNK - 1) / GZ_B250_CHUNK, nsuper = (nchunks + GZ_B250_SUPER - 1) / GZ_B250_SUPER;
    uint32_t state = 0, running = 0, ok = 1;
    for (uint32_t s = 0; s < nsuper; s++) {
        B.stab[s * 7 + 5] = state; B.stab[s * 7 + 6] = running;
        running += B.stab[s * 7 + state];
        state = (B.stab[s * 7 + 4] >> (8 * state)) & 0xff;
        if (state == 0xff) { ok = 0; break; }
    }
    B.info[0] = ok ? running : 0; B.info[1] = ok; B.info[2] = 0;
}

__global__ void __launch_bounds__(256) k_b250_convert (GzdB250Big *jobs)
{
    const GzdB250Big &B = jobs[blockIdx.y];
    const GzdB250Job &J = B.j;
    const uint32_t seg_len = d_b250_seg_len (J);
    const uint32_t nchunks = (seg_len + GZ_B250_CHUNK - 1) / GZ_B250_CHUNK;
    const uint32_t c0 = blockIdx.x * GZ_B250_SUPER;
    if (c0 >= nchunks || !B.info[1]) return;
    uint32_t *ctab = J.chunk_tab;
    uint32_t *sh = (uint32_t *)gz_lds;               // [k * 2] entry, [k * 2 + 1] base of the workgroup's chunks
    if (!threadIdx.x) {
        const uint32_t last = nchunks - c0 < GZ_B250_SUPER ? nchunks - c0 : GZ_B250_SUPER;
        uint32_t state = B.stab[blockIdx.x * 7 + 5], running = B.stab[blockIdx.x * 7 + 6];
        for (uint32_t k = 0; k < last; k++) {
            sh[k * 2] = state; sh[k * 2 + 1] = running;
            running += ctab[(c0 + k) * 7 + state];
            state = (ctab[(c0 + k) * 7 + 4] >> (8 * state)) & 0xff;
        }
    }
    __syncthreads ();
    const uint32_t c = c0 + threadIdx.x;
    if (c >= nchunks) return;
    const uint8_t *seg = J.seg;
    int32_t *wi = J.wi;
    int64_t hi = (int64_t)seg_len - (int64_t)c * GZ_B250_CHUNK, lo = hi - GZ_B250_CHUNK;
    if (lo < 0) lo = 0;
    int64_t p = hi - 1 - sh[threadIdx.x * 2];
    uint32_t k = sh[threadIdx.x * 2 + 1], bad = 0;
    while (p >= lo) {
        const int n = d_varl_len (seg[p]);
        if (p - n + 1 < 0) { bad = 1; break; }
        int32_t v = d_seg_value (seg, p, n);
        if (v >= 0 && (uint32_t)v >= J.ol_nodes_len) {       // node -> word (context.h:109)
            const uint32_t local = (uint32_t)v - J.ol_nodes_len;
            if (local >= J.n_new_nodes) { bad = 1; v = 0; }
            else v = J.node2word[local];
        }
        wi[k++] = v;
        p -= n;
    }
    if (bad) atomicMax (&B.info[2], 1u);
}

// forward word i of the stream, ONE_UP applied (b250.c:236,251): its code and length
__device__ static inline int d_b250_word (const GzdB250Job &J, const int32_t *wi, uint32_t cnt, uint32_t i, bool one_up_ok, uint32_t *code)
{
    int32_t cur = wi[cnt - 1 - i];
    if (one_up_ok && i && cur >= 0) { const int32_t prev = wi[cnt - i]; if (prev >= 0 && cur == prev + 1) cur = -2; }
    return d_varl_code (cur, code);
}

// grid (tiles of 256 words over seg_len - every word is at least a byte, jobs)
__global__ void __launch_bounds__(256) k_b250_len (GzdB250Big *jobs)
{
    const GzdB250Big &B = jobs[blockIdx.y];
    const uint32_t cnt = B.info[0], i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= cnt) return;
    const bool one_up_ok = (uint64_t)B.j.n_new_nodes + B.j.ol_nodes_len > 1024;
    uint32_t code, len = 0;
    if (i < cnt) len = (uint32_t)d_b250_word (B.j, B.j.wi, cnt, i, one_up_ok, &code);
    uint32_t *sh = (uint32_t *)gz_lds;
    sh[threadIdx.x] = len;
    __syncthreads ();
    for (int d = 128; d; d >>= 1) { if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d]; __syncthreads (); }
    if (!threadIdx.x) B.tile[blockIdx.x] = sh[0];
}

// grid (jobs)
__global__ void __launch_bounds__(256) k_b250_scan (GzdB250Big *jobs)
{
    const GzdB250Big &B = jobs[blockIdx.x];
    const uint32_t cnt = B.info[0], n_tiles = (cnt + 255) / 256, tid = threadIdx.x;
    uint64_t *sh = (uint64_t *)gz_lds;
    uint64_t carry = 0;
    for (uint32_t base = 0; base < n_tiles; base += 256) {
        const uint32_t t = base + tid;
        const uint64_t v = t < n_tiles ? B.tile[t] : 0;
        __syncthreads ();
        sh[tid] = v;
        __syncthreads ();
        for (int d = 1; d < 256; d <<= 1) {
            const uint64_t add = (int)tid >= d ? sh[tid - d] : 0;
            __syncthreads ();
            sh[tid] += add;
            __syncthreads ();
        }
        if (t < n_tiles) B.tile[t] = carry + sh[tid] - v;
        carry += sh[255];
    }
    if (!tid) {
        const bool ok = B.info[1] && !B.info[2];
        *B.j.out_len_dev = ok ? (uint32_t)carry : 0;
        if (B.j.status_dev) *B.j.status_dev = ok ? GZ_ST_OK : GZ_ST_CORRUPT;
    }
}

// grid like k_b250_len
__global__ void __launch_bounds__(256) k_b250_emit (GzdB250Big *jobs)
{
    const GzdB250Big &B = jobs[blockIdx.y];
    const uint32_t cnt = B.info[0], i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= cnt || !B.info[1] || B.info[2]) return;
    const bool one_up_ok = (uint64_t)B.j.n_new_nodes + B.j.ol_nodes_len > 1024;
    uint32_t code = 0, len = 0;
    if (i < cnt) len = (uint32_t)d_b250_word (B.j, B.j.wi, cnt, i, one_up_ok, &code);
    uint32_t *sh = (uint32_t *)gz_lds;
    sh[threadIdx.x] = len;
    __syncthreads ();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t add = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
        __syncthreads ();
        sh[threadIdx.x] += add;
        __syncthreads ();
    }
    uint8_t *o = B.j.out + B.tile[blockIdx.x] + (sh[threadIdx.x] - len);
    for (uint32_t k = 0; k < len; k++) o[k] = (uint8_t)(code >> (8 * (len - 1 - k)));
}

// paired FASTQ: an R2 b250 that came out identical to its R1 counterpart is dropped (b250.c:270-277). grid (jobs)
__global__ void __launch_bounds__(256) k_b250_pair_identical (GzdB250Job *jobs)
{
    GzdB250Job &J = jobs[blockIdx.x];
    if (!J.r1 || !J.r1_len_dev) return;
    const uint32_t n = *J.out_len_dev;
    uint32_t *sh = (uint32_t *)gz_lds;
    if (!threadIdx.x) sh[0] = (n && n == *J.r1_len_dev) ? 1u : 0u;
    __syncthreads ();
    if (!sh[0]) return;
    __syncthreads ();
    uint32_t diff = 0;
    for (uint32_t i = threadIdx.x; i < n && !diff; i += 256) diff = J.out[i] != J.r1[i];
    if (diff) sh[0] = 0;                                        // (benign race: every writer writes 0)
    __syncthreads ();
    if (!threadIdx.x && sh[0]) { *J.out_len_dev = 0; if (J.status_dev) *J.status_dev = GZ_ST_DROPPED; }
}

// ======================================================================================================
// element byte order: little-endian native <-> big-endian file order, zig-zag "interlace" for signed types
// ======================================================================================================
__global__ void k_local_order (uint8_t *data, uint64_t n, uint32_t w, int is_signed, int to_file)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t mask = w == 8 ? ~0ull : ((1ull << (8 * w)) - 1), sign = 1ull << (8 * w - 1);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint8_t *p = data + i * w;
        uint64_t v = 0;
        if (to_file) {
            for (uint32_t k = 0; k < w; k++) v |= (uint64_t)p[k] << (8 * k);
            if (is_signed) v = (v & sign) ? ((((~v + 1) & mask) << 1) - 1) & mask : (v << 1) & mask;   // context.h:99-100
            for (uint32_t k = 0; k < w; k++) p[k] = (uint8_t)(v >> (8 * (w - 1 - k)));
        }
        else {
            for (uint32_t k = 0; k < w; k++) v = (v << 8) | p[k];
            if (is_signed) v = (v & 1) ? (~(v >> 1)) & mask : (v >> 1);                              // context.h:101: -(u>>1)-1
            for (uint32_t k = 0; k < w; k++) p[k] = (uint8_t)(v >> (8 * k));
        }
    }
}

// ======================================================================================================
// dst[c*rows + r] = src[r*cols + c], elements of w bytes; 32x32 tiles through LDS, block = (32, 8)
// ======================================================================================================
__global__ void k_transpose (const uint8_t *src, uint8_t *dst, uint32_t rows, uint32_t cols, uint32_t w)
{
    uint32_t *tile = (uint32_t *)gz_lds;             // [32][33]
    const uint32_t c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (uint32_t j = threadIdx.y; j < 32; j += 8) {
        uint32_t r = r0 + j, c = c0 + threadIdx.x;
        if (r < rows && c < cols) {
            const uint8_t *p = src + ((uint64_t)r * cols + c) * w;
            uint32_t v = 0;
            for (uint32_t k = 0; k < w; k++) v |= (uint32_t)p[k] << (8 * k);
            tile[j * 33 + threadIdx.x] = v;
        }
    }
    __syncthreads ();
    for (uint32_t j = threadIdx.y; j < 32; j += 8) {
        uint32_t c = c0 + j, r = r0 + threadIdx.x;
        if (r < rows && c < cols) {
            uint32_t v = tile[threadIdx.x * 33 + j];
            uint8_t *p = dst + ((uint64_t)c * rows + r) * w;
            for (uint32_t k = 0; k < w; k++) p[k] = (uint8_t)(v >> (8 * k));
        }
    }
}

__global__ void __launch_bounds__(256) k_adler32 (const uint8_t *data, uint32_t len, uint32_t *out)
{
    uint32_t a = gz_adler32_wg (data, len, threadIdx.x);
    if (!threadIdx.x) *out = a;
}

// ======================================================================================================
// CODEC_ACGT pre-transform: SEQ -> 2 bits per base + exception stream (codec_acgt.c:45-55,64-129;
// the table of reference.c:45-58). HBM-bound: reads n, writes n / 4 + n. One thread per 16 bases = one 16-byte load,
// one packed dword, one 16-byte store of exceptions (x may be the seq buffer itself, like the reference's overlay).
// ======================================================================================================
__device__ static inline uint32_t d_acgt_code (uint32_t c)          // IUPAC codes map to the lowest of their bases, the rest to 0
{
    c |= 0x20;
    return (c == 'c' || c == 'y' || c == 's' || c == 'b') ? 1u : (c == 'g' || c == 'k') ? 2u : (c == 't' || c == 'u') ? 3u : 0u;
}
__device__ static inline uint32_t d_acgt_exception (uint32_t c)     // 0: ACGT, 1: acgt, else the character
{
    return (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? 0u : (c == 'a' || c == 'c' || c == 'g' || c == 't') ? 1u : c;
}

// grid-stride; packed_bytes = whole 64-bit words (the excess is cleared)
__global__ void __launch_bounds__(256) k_acgt_pack (const uint8_t *seq, uint64_t n, uint8_t *packed, uint64_t packed_bytes, uint8_t *x, uint32_t *has_x)
{
    const uint64_t groups = (n + 15) / 16, words = packed_bytes / 4;
    uint32_t any = 0;
    uint16_t *tab = (uint16_t *)gz_lds;                            // byte -> code | exception << 8 (one LDS read per base)
    tab[threadIdx.x] = (uint16_t)(d_acgt_code (threadIdx.x) | (d_acgt_exception (threadIdx.x) << 8));
    __syncthreads ();
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < words; g += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t w = 0;
        if (g < groups) {
            const uint64_t base = g * 16;
            const uint32_t m = base + 16 <= n ? 16u : (uint32_t)(n - base);          // bases of this group that exist
            gz_u32x4_unaligned v = { 0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u };   // 'A': code 0, exception 0
            if (m == 16) v = *(const gz_u32x4_unaligned *)(seq + base);
            else for (uint32_t k = 0; k < m; k++) v[k >> 2] = (v[k >> 2] & ~(0xffu << (8 * (k & 3)))) | ((uint32_t)seq[base + k] << (8 * (k & 3)));
            gz_u32x4_unaligned ev = { 0, 0, 0, 0 };
            #pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t t = tab[(v[k >> 2] >> (8 * (k & 3))) & 0xff];
                w |= (t & 3) << (2 * k);
                ev[k >> 2] |= (t >> 8) << (8 * (k & 3));
            }
            any |= ev[0] | ev[1] | ev[2] | ev[3];
            if (m == 16) *(gz_u32x4_unaligned *)(x + base) = ev;
            else for (uint32_t k = 0; k < m; k++) x[base + k] = (uint8_t)(ev[k >> 2] >> (8 * (k & 3)));
        }
        ((uint32_t *)packed)[g] = w;                               // (the arena and torch allocations are 4-byte aligned)
    }
    // (one flag for the whole stream: only the first waves to see an exception touch it)
    if (__ballot (any != 0) && (threadIdx.x & 63) == 0 && *(volatile uint32_t *)has_x == 0) atomicMax (has_x, 1u);
}

// x == NULL: no exceptions (flags.acgt_no_x). One thread per 16 bases.
__global__ void __launch_bounds__(256) k_acgt_unpack (const uint8_t *packed, const uint8_t *x, uint64_t n, uint8_t *seq)
{
    const uint64_t groups = (n + 15) / 16;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t base = g * 16;
        const uint32_t m = base + 16 <= n ? 16u : (uint32_t)(n - base);
        const uint32_t w = ((const uint32_t *)packed)[g];
        gz_u32x4_unaligned ev = { 0, 0, 0, 0 };
        if (x) {
            if (m == 16) ev = *(const gz_u32x4_unaligned *)(x + base);
            else for (uint32_t k = 0; k < m; k++) ev[k >> 2] |= (uint32_t)x[base + k] << (8 * (k & 3));
        }
        gz_u32x4_unaligned out = { 0, 0, 0, 0 };
        #pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t b = (0x54474341u >> (8 * ((w >> (2 * k)) & 3))) & 0xff;    // "ACGT"
            const uint32_t e = (ev[k >> 2] >> (8 * (k & 3))) & 0xff;
            out[k >> 2] |= (e == 0 ? b : e == 1 ? b + 32 : e) << (8 * (k & 3));
        }
        if (m == 16) *(gz_u32x4_unaligned *)(seq + base) = out;
        else for (uint32_t k = 0; k < m; k++) seq[base + k] = (uint8_t)(out[k >> 2] >> (8 * (k & 3)));
    }
}
