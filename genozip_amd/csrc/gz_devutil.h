// gz_devutil.h -- small device helpers shared by the encode and decode kernels
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gz_device.h"

// status values kept in GzdStream.status while a batch is in flight
#define GZ_ST_PENDING   2
#define GZ_ST_OK        1
#define GZ_ST_DROPPED   3             // a b250 identical to its R1 counterpart: no section (b250.c:270-277)
#define GZ_ST_TOO_SMALL 0
#define GZ_ST_CORRUPT   (-5)
#define GZ_ST_FAILED    (-1)          // internal failure (a kernel gave up): the stream / VBlock is reported as GZ_ERR

// All LDS scratch lives in the dynamic region (keeps its base 16-byte aligned; cdna_hip_programming.md G17)
extern __shared__ __attribute__((aligned(16))) uint8_t gz_lds[];

__host__ __device__ static inline int gz_codec_order (int codec)   // codec_htscodecs.c:17-20
{
    switch (codec) {
        case 6: case 16: return 0x01;
        case 7: case 17: return 0x19;
        case 8: case 18: return 0x81;
        case 9: case 19: return 0x99;
        default: return -1;
    }
}

// plane k of an n-byte stream holds bytes k, k+4, ... : len[k] = n/4 + (n%4 > k)   (rANS_static4x16pr.c:1178-1181)
__host__ __device__ static inline void gz_plane_geometry (uint32_t n, uint32_t *len, uint32_t *off)
{
    uint32_t o = 0;
    for (uint32_t k = 0; k < 4; k++) { len[k] = n / 4 + ((n % 4) > k); off[k] = o; o += len[k]; }
}

__host__ __device__ static inline uint32_t gz_vi_len (uint32_t v)
{
    return v < (1u << 7) ? 1 : v < (1u << 14) ? 2 : v < (1u << 21) ? 3 : v < (1u << 28) ? 4 : 5;
}

// 7-bit groups, most significant first (varint.h:206-240)
__host__ __device__ static inline uint32_t gz_vi_put (uint8_t *dst, uint32_t v)
{
    uint32_t n = gz_vi_len (v);
    for (uint32_t k = 0; k < n; k++) {
        uint32_t sh = 7 * (n - 1 - k);
        dst[k] = (uint8_t)(((v >> sh) & 0x7f) | (k + 1 < n ? 0x80 : 0));
    }
    return n;
}

// returns bytes consumed, 0 on truncation
__host__ __device__ static inline uint32_t gz_vi_get (const uint8_t *src, uint32_t avail, uint32_t *v)
{
    uint32_t acc = 0, n = 0;
    while (n < avail && n < 6) {
        uint8_t c = src[n++];
        acc = (acc << 7) | (c & 0x7f);
        if (!(c & 0x80)) { *v = acc; return n; }
    }
    *v = acc;
    return 0;
}

__host__ __device__ static inline uint32_t gz_pow2_ceil (uint32_t v)   // 0 -> 0
{
    v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
    return v + 1;
}

__host__ __device__ static inline void gz_be32 (uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v;
}

__host__ __device__ static inline uint32_t gz_rd_be32 (const uint8_t *p)
{
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

// adler32 (start value 1) of `len` bytes by a 256-thread workgroup. Thread t sums a contiguous slice:
// A = sum d_j, B = sum (m-j) d_j ; slices combine as a' = a + A, b' = b + m*a + B (mod 65521).
// Uses the first 2 KB + 8 bytes of gz_lds. Result valid in every thread.
typedef uint32_t gz_u32x4_unaligned __attribute__((vector_size (16), aligned (1)));

// adler32 = (b << 16 | a) with a = 1 + sum d_i, b = len + sum (len - i) d_i, both mod 65521: a weighted sum, so any
// thread can take any bytes. Thread t takes bytes [4096 k + 16 t, + 16) for k = 0, 1, ... (coalesced 16-byte loads).
// the same sums over bytes [lo, hi) of a buffer of len bytes only (lo a multiple of 16), without the constant terms: *a_out = sum d_i,
// *w_out = sum (len - i) d_i, both mod 65521, valid in every thread - several workgroups take a section's bytes between them (k_emit)
__device__ static inline void gz_adler32_part (const uint8_t *data, uint32_t len, uint32_t lo, uint32_t hi, int tid, uint32_t *a_out, uint32_t *w_out)
{
    uint32_t *sA = (uint32_t *)gz_lds, *sB = sA + 256, *res = sA + 512;
    uint64_t A = 0, W = 0;
    const uint32_t body = lo + ((hi - lo) & ~15u);
    uint32_t rounds = 0;
    const uint32_t step = 256 * 16;
    for (uint32_t i = lo + (uint32_t)tid * 16; i < body; i += step) {
        const gz_u32x4_unaligned v = *(const gz_u32x4_unaligned *)(data + i);
        uint32_t S = 0, T = 0;
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t w = v[k], b0 = w & 0xff, b1 = (w >> 8) & 0xff, b2 = (w >> 16) & 0xff, b3 = w >> 24;
            S += b0 + b1 + b2 + b3;
            T += (4 * k) * b0 + (4 * k + 1) * b1 + (4 * k + 2) * b2 + (4 * k + 3) * b3;
        }
        A += S; W += (uint64_t)(len - i) * S - T;
        if (++rounds == 65536) { W %= 65521u; rounds = 0; }
    }
    for (uint32_t i = body + tid; i < hi; i += 256) { const uint32_t d = data[i]; A += d; W += (uint64_t)(len - i) * d; }
    __syncthreads ();
    sA[tid] = (uint32_t)(A % 65521u);
    sB[tid] = (uint32_t)(W % 65521u);
    __syncthreads ();
    if (!tid) {
        uint64_t a = 0, b = 0;
        for (int t = 0; t < 256; t++) { a += sA[t]; b += sB[t]; }
        res[0] = (uint32_t)(a % 65521u); res[1] = (uint32_t)(b % 65521u);
    }
    __syncthreads ();
    *a_out = res[0]; *w_out = res[1];
}

__device__ static inline uint32_t gz_adler32_wg (const uint8_t *data, uint32_t len, int tid)
{
    uint32_t *sA = (uint32_t *)gz_lds, *sB = sA + 256, *res = sA + 512;
    uint64_t A = 0, W = 0;
    const uint32_t body = len & ~15u;
    uint32_t rounds = 0;
    auto take = [&] (const gz_u32x4_unaligned &v, uint32_t i) {
        uint32_t S = 0, T = 0;                                 // sum d_j, sum j d_j over the 16 bytes
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t w = v[k], b0 = w & 0xff, b1 = (w >> 8) & 0xff, b2 = (w >> 16) & 0xff, b3 = w >> 24;
            S += b0 + b1 + b2 + b3;
            T += (4 * k) * b0 + (4 * k + 1) * b1 + (4 * k + 2) * b2 + (4 * k + 3) * b3;
        }
        A += S;
        W += (uint64_t)(len - i) * S - T;                      // < 2^44 each
        if (++rounds == 65536) { W %= 65521u; rounds = 0; }
    };
    const uint32_t step = 256 * 16;
    uint32_t i = (uint32_t)tid * 16;
    for (; i + 3 * step < body; i += 4 * step) {               // (four loads in flight)
        const gz_u32x4_unaligned v0 = *(const gz_u32x4_unaligned *)(data + i), v1 = *(const gz_u32x4_unaligned *)(data + i + step),
                                 v2 = *(const gz_u32x4_unaligned *)(data + i + 2 * step), v3 = *(const gz_u32x4_unaligned *)(data + i + 3 * step);
        take (v0, i); take (v1, i + step); take (v2, i + 2 * step); take (v3, i + 3 * step);
    }
    for (; i < body; i += step) { const gz_u32x4_unaligned v = *(const gz_u32x4_unaligned *)(data + i); take (v, i); }
    for (uint32_t i = body + tid; i < len; i += 256) { const uint32_t d = data[i]; A += d; W += (uint64_t)(len - i) * d; }
    __syncthreads ();
    sA[tid] = (uint32_t)(A % 65521u);
    sB[tid] = (uint32_t)(W % 65521u);
    __syncthreads ();
    if (!tid) {
        uint64_t a = 1, b = len % 65521u;
        for (int t = 0; t < 256; t++) { a += sA[t]; b += sB[t]; }
        res[0] = (uint32_t)(((b % 65521u) << 16) | (a % 65521u));
    }
    __syncthreads ();
    return res[0];
}
