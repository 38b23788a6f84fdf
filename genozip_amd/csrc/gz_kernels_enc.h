// gz_kernels_enc.h -- gfx950 kernels of the compression direction.
//
// Pipeline over a table of streams / leaves (see gz_device.h), one launch per phase, every phase covering ALL
// leaves of ALL streams of ALL VBlocks in the batch:
//   k_resolve      stream lengths, effective codec, striping decision          (codec rules: compressor.c:56-58,
//                                                                                rANS_static4x16pr.c:1162)
//   k_stripe       byte i -> plane i%4                                          (rANS_static4x16pr.c:1174-1190)
//   k_leaf_prep    PACK (pack.c:58-154), unit prefix, alphabet of the coded bytes
//   k_hist         order-0 / order-1 histograms (utils.h:81,137)
//   k_rans_table   normalise, compute_shift, serialise + (optionally) order-0 code the table, build the encoder
//                  records (rANS_static4x16pr.c:113-203,254-322,376-433,626-687,729-796)
//   k_rans_encode  the 4 interleaved rANS states of a leaf on 4 lanes of one wave (rANS_word.h:280-320)
//   (arithmetic coder: gz_kernels_arith.h) (arith_dynamic.c:92-197,387-561, c_range_coder.h,
//                  c_simple_model.h)
//   k_select       CAT fallback, best method per plane, stream size
//   k_vb_layout    section offsets inside each VBlock's z_data (zip.c:560-585 order, given by the caller)
//   k_emit         final bytes: stripe meta + units; SectionHeaderCtx + adler32 in VBlock mode
#pragma once
#include "gz_device.h"
#include "gz_devutil.h"
#include <gz_intrin.h>

// ======================================================================================================
// k_resolve : one thread per stream
// ======================================================================================================
__global__ void k_resolve (GzdStream *streams, uint32_t n_streams, int section_mode)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_streams) return;
    GzdStream &S = streams[i];
    if (S.status != GZ_ST_PENDING) return;

    uint32_t n = S.in_len;
    if (S.in_len_dev) { uint32_t v = *S.in_len_dev; if (v < n) n = v; }
    S.n = n;

    int codec = S.codec_req;
    if (S.pre) { S.codec = S.hdr_codec ? S.hdr[26] : S.hdr[25]; S.engine = GZ_ENG_NONE; S.order = 0; S.striped = 0; return; }   // already coded: only framed
    if (section_mode && n < 50 && !S.hdr_codec) codec = 1 /* CODEC_NONE: compressor.c:56-58 (simple codecs) */;
    S.codec = (uint8_t)codec;

    int order = gz_codec_order (codec);
    S.engine = (codec >= 6 && codec <= 9) ? GZ_ENG_RANS : (codec >= 16 && codec <= 19) ? GZ_ENG_ARITH : GZ_ENG_NONE;
    if (order < 0) order = 0;
    if (n <= 20) order &= ~GZ_X_STRIPE;       // rANS_static4x16pr.c:1162, arith_dynamic.c:626
    S.order   = (uint8_t)order;
    S.striped = (order & GZ_X_STRIPE) ? 1 : 0;
}

// ======================================================================================================
// k_stripe : grid (n_streams, chunks)
// ======================================================================================================
__global__ void k_stripe (GzdStream *streams)
{
    const GzdStream S = streams[blockIdx.x];
    if (S.status != GZ_ST_PENDING || !S.striped) return;
    uint32_t n = S.n, off[4], len[4];
    gz_plane_geometry (n, len, off);
    // each thread moves 4 consecutive bytes of one plane: plane k, positions x..x+3 <- in[4x+k], in[4(x+1)+k], ...
    uint32_t quads = (n + 15) / 16;   // groups of 16 input bytes
    for (uint32_t g = blockIdx.y * blockDim.x + threadIdx.x; g < quads; g += gridDim.y * blockDim.x) {
        uint32_t base = g * 16;
        uint8_t b[16];
        #pragma unroll
        for (int j = 0; j < 16; j++) b[j] = (base + j < n) ? S.in[base + j] : 0;
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t x = g * 4;
            #pragma unroll
            for (int j = 0; j < 4; j++)
                if (x + j < len[k]) S.planes[off[k] + x + j] = b[4 * j + k];
        }
    }
}

// ======================================================================================================
// presence flags of the byte values in p[0..n): 16 bytes per thread and load (a long leaf has one workgroup to itself)
__device__ static inline void d_mark_present (uint32_t *flags, const uint8_t *p, uint32_t n, int tid)
{
    uint32_t head = (uint32_t)((16 - ((uintptr_t)p & 15)) & 15);
    if (head > n) head = n;
    for (uint32_t i = tid; i < head; i += 256) flags[p[i]] = 1;
    const uint4 *q = (const uint4 *)(p + head);
    const uint32_t nv = (n - head) / 16;
    for (uint32_t i = tid; i < nv; i += 256) {
        const uint4 v = q[i];
        const uint32_t w[4] = { v.x, v.y, v.z, v.w };
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            flags[w[k] & 0xff] = 1; flags[(w[k] >> 8) & 0xff] = 1; flags[(w[k] >> 16) & 0xff] = 1; flags[w[k] >> 24] = 1;
        }
    }
    for (uint32_t i = head + nv * 16 + tid; i < n; i += 256) flags[p[i]] = 1;
}

// k_leaf_prep : one 256-thread workgroup per leaf
// ======================================================================================================
// Which byte values occur in a leaf's source bytes: many workgroups per leaf. (k_leaf_prep, one workgroup per leaf, used to find
// out by itself - 0.9 ms for a 6 MB quality stream, in front of that stream's whole pipeline.)
// grid (leaves, GZ_PRES_SLICES), 256 threads, 1 KB of LDS; L.pres arrives zeroed (the leaf table is uploaded that way)
#define GZ_PRES_SLICES 32
__global__ void __launch_bounds__(256) k_presence (GzdStream *streams, GzdLeaf *leaves)
{
    GzdLeaf &L = leaves[blockIdx.x];
    const GzdStream &S = streams[L.stream];
    const int tid = threadIdx.x;
    if (!(S.status == GZ_ST_PENDING && S.engine == L.engine && ((L.plane == 0xff) ? !S.striped : S.striped))) return;   // (k_leaf_prep's "active")
    const uint8_t *src; uint32_t n;
    if (L.plane == 0xff) { src = S.in; n = S.n; }
    else { uint32_t len[4], off[4]; gz_plane_geometry (S.n, len, off); src = S.planes + off[L.plane]; n = len[L.plane]; }
    // slices of whole 16-byte units
    const uint32_t per = (((n + GZ_PRES_SLICES - 1) / GZ_PRES_SLICES) + 15) & ~15u;
    const uint32_t lo = blockIdx.y * per;
    if (lo >= n) return;
    const uint32_t cnt = n - lo < per ? n - lo : per;
    uint32_t *flags = (uint32_t *)gz_lds;
    flags[tid] = 0;
    __syncthreads ();
    d_mark_present (flags, src + lo, cnt, tid);
    __syncthreads ();
    const uint64_t m = __ballot (flags[tid] != 0);             // 4 waves x 64 byte values
    if (!(tid & 63) && m) { if ((uint32_t)m) atomicOr (&L.pres[(tid >> 5)], (uint32_t)m); if (m >> 32) atomicOr (&L.pres[(tid >> 5) + 1], (uint32_t)(m >> 32)); }
}

__global__ void __launch_bounds__(256) k_leaf_prep (GzdStream *streams, GzdLeaf *leaves)
{
    GzdLeaf &L = leaves[blockIdx.x];
    const GzdStream &S = streams[L.stream];
    const int tid = threadIdx.x;
    uint32_t *flags = (uint32_t *)gz_lds;          // [256] presence -> rank
    uint32_t *misc  = flags + 256;                 // [0] nsym  [1] max present  ; [16..] symbol list
    const bool arith = L.engine == GZ_ENG_ARITH;

    bool active = S.status == GZ_ST_PENDING && S.engine == L.engine &&
                  ((L.plane == 0xff) ? !S.striped : S.striped);
    if (!active) { if (!tid) { L.active = 0; L.unit_len = 0; } return; }
    if (L.succ) for (int i = tid; i < 1024; i += 256) L.succ[i] = 0;        // (k_ctx_succ fills it in)

    const uint8_t *src; uint32_t n;
    if (L.plane == 0xff) { src = S.in; n = S.n; }
    else { uint32_t len[4], off[4]; gz_plane_geometry (S.n, len, off); src = S.planes + off[L.plane]; n = len[L.plane]; }

    uint8_t  flag = L.method;
    bool     o1 = flag & 1, rle = arith && (flag & GZ_X_RLE), nosz = flag & GZ_X_NOSZ;
    bool     packed_on = false;
    const uint8_t *coded = src; uint32_t coded_n = n;
    uint32_t nsym_pack = 0;

    if (flag & GZ_X_PACK) {
        if (!n) flag &= ~GZ_X_PACK;
        else {
            flags[tid] = (L.pres[tid >> 5] >> (tid & 31)) & 1;      // (k_presence)
            __syncthreads ();
            if (!tid) {
                uint32_t ns = 0;
                for (int s = 0; s < 256; s++) if (flags[s]) { misc[16 + ns] = s; flags[s] = ns++; }
                misc[0] = ns;
            }
            __syncthreads ();
            nsym_pack = misc[0];
            if (nsym_pack > 16 && nsym_pack < 256) flag &= ~GZ_X_PACK;            // rANS_static4x16pr.c:1260-1264
            else {
                packed_on = true;
                if (nsym_pack <= 16) {
                    uint32_t per = nsym_pack > 4 ? 2 : nsym_pack > 2 ? 4 : nsym_pack > 1 ? 8 : 0;
                    coded = L.packed;
                    coded_n = per ? (n + per - 1) / per : 0;
                    uint32_t width = per ? 8 / per : 0;
                    for (uint32_t j = tid; j < coded_n; j += 256) {
                        uint32_t v = 0, base = j * per;
                        for (uint32_t t = 0; t < per && base + t < n; t++) v |= flags[src[base + t]] << (t * width);
                        L.packed[j] = (uint8_t)v;
                    }
                }
                // 256 distinct symbols: the count byte wraps to 0, data passes through unchanged and stays "packed"
            }
            __syncthreads ();
        }
    }

    if (o1 && coded_n < 8) { flag &= arith ? ~3 : ~1; o1 = false; }               // :1333-1336 / arith :803-806
    if (rle && !n) flag &= ~GZ_X_RLE;                                             // arith :798-800 (coder still RLE)

    if (!tid) {
        uint32_t p = 0;
        L.prefix[p++] = flag;
        if (!nosz) p += gz_vi_put (L.prefix + p, n);
        if (packed_on) {
            L.prefix[p++] = (uint8_t)nsym_pack;
            if (nsym_pack <= 16) for (uint32_t k = 0; k < nsym_pack; k++) L.prefix[p++] = (uint8_t)misc[16 + k];
            p += gz_vi_put (L.prefix + p, coded_n);
        }
        L.prefix_len = (uint8_t)p;
        L.active = 1; L.flag = flag; L.o1 = o1; L.rle = rle; L.packed_on = packed_on; L.cat = 0;
        L.src = src; L.n = n; L.coded = coded; L.coded_n = coded_n;
        L.arith_n = coded_n;                                  // (the run-length variant: k_rle_events puts its event count here)
        L.nctx = rle ? 768 : 256;
        L.tab_len = 0; L.pay_len = 0; L.overflow = 0; L.unit_len = 0; L.shift_bits = 0;
    }
    __syncthreads ();   // packed bytes written by all threads are read below

    // alphabet of the coded bytes: rank map for the order-1 histogram, max symbol for the arith models
    if (coded == src) flags[tid] = (L.pres[tid >> 5] >> (tid & 31)) & 1;   // (k_presence)
    else {
        flags[tid] = 0;
        __syncthreads ();
        d_mark_present (flags, coded, coded_n, tid);
    }
    __syncthreads ();
    if (!tid) {
        uint32_t ns = 0, mx = 0;
        for (int s = 0; s < 256; s++) {
            if (flags[s]) { L.symlist[ns] = (uint8_t)s; L.symrank[s] = (uint16_t)ns; ns++; mx = s; }
            else L.symrank[s] = 0xffff;
        }
        L.nsym = ns;
        L.max_sym = mx + 1;
    }

    // clear the histogram of rANS leaves (256 order-0 counters, or 256 rows x 256 + 256 totals)
    if (L.engine == GZ_ENG_RANS && L.F) {
        uint32_t cnt = o1 ? 256 * 256 + 256 : 256;
        for (uint32_t i = tid; i < cnt; i += 256) L.F[i] = 0;
    }
}

// ======================================================================================================
// k_hist : grid (n_leaves, GZ_HIST_CHUNKS), 256 threads, dynamic LDS = GZ_HIST_LDS bytes
// ======================================================================================================
#define GZ_HIST_CHUNKS 8
#define GZ_HIST_RANKS  120                       // order-1 counters kept in LDS when the alphabet is this small
#define GZ_HIST_LDS    ((GZ_HIST_RANKS + 1) * GZ_HIST_RANKS * 4 + 64)

__global__ void __launch_bounds__(256) k_hist (GzdLeaf *leaves)
{
    GzdLeaf &L = leaves[blockIdx.x];
    if (!L.active || L.engine != GZ_ENG_RANS) return;
    const int tid = threadIdx.x;
    const uint32_t n = L.coded_n;
    if (!n) return;
    const uint8_t *in = L.coded;
    uint32_t chunk = (n + gridDim.y - 1) / gridDim.y;
    uint32_t lo = blockIdx.y * chunk, hi = lo + chunk < n ? lo + chunk : n;
    if (lo >= hi) return;
    uint32_t *h = (uint32_t *)gz_lds;

    if (!L.o1) {
        h[tid] = 0;
        __syncthreads ();
        for (uint32_t i = lo + tid; i < hi; i += 256) atomicAdd (&h[in[i]], 1u);
        __syncthreads ();
        if (h[tid]) atomicAdd (&L.F[tid], h[tid]);
        return;
    }

    const uint32_t ns = L.nsym;
    const uint32_t q = n >> 2;
    if (ns <= GZ_HIST_RANKS) {
        // counters indexed by (rank of context, rank of symbol); context 0 may be absent from the data -> rank slot ns
        const uint32_t w = ns + 1;     // one extra row for the "context 0" of stream / quarter heads if byte 0 is absent
        for (uint32_t i = tid; i < w * ns; i += 256) h[i] = 0;
        __syncthreads ();
        const uint32_t r0 = L.symrank[0] != 0xffff ? L.symrank[0] : ns;
        for (uint32_t i = lo + tid; i < hi; i += 256) {
            uint32_t rc = i ? L.symrank[in[i - 1]] : r0;
            atomicAdd (&h[rc * ns + L.symrank[in[i]]], 1u);
        }
        if (!blockIdx.y && !tid)
            for (int k = 1; k < 4; k++) atomicAdd (&h[r0 * ns + L.symrank[in[k * q]]], 1u);  // quarter heads, :730-733
        __syncthreads ();
        for (uint32_t i = tid; i < w * ns; i += 256) {
            uint32_t v = h[i];
            if (!v) continue;
            uint32_t rc = i / ns, rs = i % ns;
            uint32_t c = rc == ns ? 0 : L.symlist[rc];
            atomicAdd (&L.F[c * 256 + L.symlist[rs]], v);
        }
    }
    else {
        for (uint32_t i = lo + tid; i < hi; i += 256) atomicAdd (&L.F[(i ? in[i - 1] : 0) * 256 + in[i]], 1u);
        if (!blockIdx.y && !tid)
            for (int k = 1; k < 4; k++) atomicAdd (&L.F[in[k * q]], 1u);
    }
}

// ======================================================================================================
// rANS table construction
// ======================================================================================================

// rANS_static4x16pr.c:113-160 -- see oracle/gz_oracle.c freq_scale for the prose. F may live in LDS or global.
__device__ static void d_freq_scale (uint32_t *F, uint32_t sum, uint32_t target)
{
    if (!sum) return;
    for (int pass = 0; ; pass++) {
        uint64_t mult = (((uint64_t)target) << 31) / sum + (uint32_t)((1u << 30) / sum);
        uint32_t big = 0, new_sum = 0;
        int big_at = 0;
        for (int s = 0; s < 256; s++) {
            uint32_t f = F[s];
            if (!f) continue;
            if (f > big) { big = f; big_at = s; }
            uint32_t g = (uint32_t)((f * mult) >> 31);
            if (!g) g = 1;
            F[s] = g;
            new_sum += g;
        }
        int32_t adjust = (int32_t)target - (int32_t)new_sum;
        if (adjust > 0) F[big_at] += (uint32_t)adjust;
        else if (adjust < 0) {
            uint32_t need = (uint32_t)(-adjust), fb = F[big_at];
            if (fb > need && (pass == 1 || fb / 2 >= need)) F[big_at] = fb - need;
            else if (pass == 0) { sum = new_sum; continue; }
            else {
                adjust += (int32_t)fb - 1;
                F[big_at] = 1;
                for (int s = 0; adjust && s < 256; s++) {
                    uint32_t f = F[s];
                    if (f < 2) continue;
                    int32_t step = (f > (uint32_t)(-adjust)) ? adjust : 1 - (int32_t)f;
                    F[s] = (uint32_t)((int32_t)f + step);
                    adjust -= step;
                }
            }
        }
        break;
    }
}

__device__ static inline void d_freq_shift_up (uint32_t *F, uint32_t sum, uint32_t target)
{
    if (!sum || sum == target) return;
    int sh = 0;
    while (sum < target) { sum <<= 1; sh++; }
    for (int s = 0; s < 256; s++) F[s] <<= sh;
}

// present[] as 0/1 words; writes the run-length coded symbol list (rANS_static4x16pr.c:179-203)
__device__ static uint32_t d_alphabet_put (uint8_t *dst, const uint32_t *present)
{
    uint32_t p = 0;
    int s = 0;
    while (s < 256) {
        if (!present[s]) { s++; continue; }
        int e = s;
        while (e + 1 < 256 && present[e + 1]) e++;
        dst[p++] = (uint8_t)s;
        if (e > s) { dst[p++] = (uint8_t)(s + 1); dst[p++] = (uint8_t)(e - (s + 1)); }
        s = e + 1;
    }
    dst[p++] = 0;
    return p;
}

// floor (n / d) for n < 2^52, d > 0 through ONE double-precision division instead of the 64-bit integer division's ~150 instructions.
// Both operands are exact doubles and IEEE division rounds the true quotient q correctly: |fl (q) - q| <= q x 2^-53 = n / (d x 2^53)
// < 1 / d. A q that is not an integer is at least 1 / d below the next integer, so rounding cannot reach it; rounding is monotonic and
// integers < 2^53 are representable, so it cannot fall below floor (q) either: the truncated double IS the integer quotient
__device__ static inline uint64_t d_div_u52 (uint64_t n, uint32_t d) { return (uint64_t)((double)n / (double)d); }

__device__ static inline GzRansSym d_rans_sym (uint32_t start, uint32_t freq, uint32_t bits)   // rANS_word.h:189-265
{
    GzRansSym r;
    r.x_max = ((0x8000u >> bits) << 16) * freq;
    uint32_t cmpl = ((1u << bits) - freq) & 0xffff, rsh;
    if (freq < 2) { r.rcp = ~0u; rsh = 0; r.bias = start + (1u << bits) - 1; }
    else {
        const uint32_t lg = 32u - (uint32_t)__clz ((int)(freq - 1));        // the smallest lg with freq <= 1 << lg
        r.rcp  = (uint32_t)d_div_u52 ((1ull << (lg + 31)) + freq - 1, freq);
        rsh    = lg - 1;
        r.bias = start;
    }
    r.cmpl_rsh = cmpl | (rsh << 16);
    return r;
}

// x -> x' and the renormalisation test for one symbol (rANS_word.h:280-320). q < 2^21 and cmpl <= 4096, so the
// product fits the 24-bit multiplier.
__device__ static inline uint32_t d_rans_advance (uint32_t x, const GzRansSym &r)
{
    uint32_t q = __umulhi (x, r.rcp) >> (r.cmpl_rsh >> 16);
    return x + r.bias + q * (r.cmpl_rsh & 0xffff);
}

__device__ static inline uint32_t d_uniform_u32 (uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane ((int)v); }

// The 4 interleaved states on lanes 0..3 of one wave. Lane k codes its own list of (record) steps, longest list
// first: round r (counting down) is coded by every lane whose list is longer than r. Within a round state 3 emits
// first, i.e. lands at the highest address of the backwards-growing stream. Called by ALL 64 lanes of the wave.
//   idx(k, r, hi, lo) loads the bytes that select the record of lane k's r-th step: record = syms[hi * 256 + lo]
// Only the state update x -> x' is serial. Which record a step needs is known from the input alone, so all 64 lanes work
// ahead for the four state lanes (lane 4j+k: state k, j-th round of a batch of 16), in three stages one batch apart: the
// input bytes -> the record they select -> hand-over through LDS (lds: 2 KB of this wave's own, two buffers).
// Returns payload length (bytes, incl. the 16 state bytes) or 0xffffffff on overflow; payload ends at buf+cap.
#define GZ_RANS_ENC_LDS 2048
template <typename IdxFn>
__device__ static uint32_t d_rans_encode_wave (uint32_t len_k, uint32_t rounds, uint8_t *buf, uint32_t cap, uint8_t *lds, const GzRansSym *syms, IdxFn idx)
{
    const int lane = threadIdx.x & 63;
    const int k_of = lane & 3, j_of = lane >> 2;
    const uint32_t len_mine = (uint32_t)__shfl ((int)len_k, k_of);      // list length of the state this lane fetches for
    uint4 *slot = (uint4 *)lds;                                        // [2][64]
    uint32_t x = 0x8000u;
    uint32_t used = 0;                 // bytes emitted so far (wave-uniform)
    bool overflow = false;
    const uint4 idle = make_uint4 (0xffffffffu, 0, 0, 0);              // x_max that never triggers
    const uint32_t NONE = 0xffffffffu;
    // (stage 1 only LOADS the bytes - hi = context byte or 0, lo = symbol, both untouched until stage 2 turns them into
    //  a table index: any arithmetic on them here would make the compiler wait for the loads on the spot)
    uint32_t nxt_hi = 0, nxt_lo = NONE;
    auto fetch_idx = [&] (uint32_t r_hi, uint32_t &hi, uint32_t &lo) {  // of rounds r_hi-1 ... r_hi-16
        hi = 0; lo = NONE;
        if (r_hi <= (uint32_t)j_of) return;
        const uint32_t r = r_hi - 1 - (uint32_t)j_of;
        if (r < len_mine) idx (k_of, r, hi, lo);
    };
    auto fetch_rec = [&] (uint32_t hi, uint32_t lo) -> uint4 { return lo == NONE ? idle : gz_ldg_u32x4 (syms + hi * 256 + lo); };
    fetch_idx (rounds, nxt_hi, nxt_lo);
    uint4 nxt = fetch_rec (nxt_hi, nxt_lo);
    fetch_idx (rounds > 16 ? rounds - 16 : 0, nxt_hi, nxt_lo);
    int b = 0;
    for (uint32_t r_hi = rounds; r_hi > 0 && !overflow; b ^= 1) {
        const uint32_t nb = r_hi < 16 ? r_hi : 16;
        slot[b * 64 + lane] = nxt;
        gz_wave_sync ();
        nxt = fetch_rec (nxt_hi, nxt_lo);                              // (its bytes were requested one batch ago)
        fetch_idx (r_hi - nb > 16 ? r_hi - nb - 16 : 0, nxt_hi, nxt_lo);   // in flight while this batch is coded
        uint4 cur = slot[b * 64 + (lane & 3)];                         // (the next round's record is read while this round works)
        if (used + 8 * nb + 16 + 8 <= cap) {
            // the usual case - no capacity test inside the batch, and nothing that makes the scalar unit wait for the
            // vector unit (a ballot feeding scalar arithmetic costs ~12 ns each time): counts by v_mbcnt / v_bcnt on the
            // ballot mask, "emit or not" as a select between the real address and a dump slot at the unused front of buf
            // (no predicate is ever combined on the scalar unit: lanes 4.. have list length 0 and a state that never
            //  reaches any x_max - the smallest is 8 << 16 -, records past the end of a list are `idle`)
            uint32_t used_v = used;
            const uint32_t len_eff = lane < 4 ? len_k : 0u, hi_bits = lane < 4 ? 0xfu >> lane : 0u;
            // (a full batch of 16 rounds is unrolled - round 6: the rolled loop was 38 instructions and a taken branch per round, a third of
            //  them loop control and the copy of the prefetched record; with j a constant the record's LDS address is an immediate)
            auto one_round = [&] (uint32_t j) __attribute__((always_inline)) {
                const uint32_t r = r_hi - 1 - j;
                GzRansSym s;
                s.x_max = cur.x; s.rcp = cur.y; s.bias = cur.z; s.cmpl_rsh = cur.w;
                cur = slot[b * 64 + ((j + 1) & 15) * 4 + (lane & 3)];
                const bool emit = x >= s.x_max;
                const uint32_t mlo = (uint32_t)__ballot (emit);          // (only lanes 0..3 can be set)
                const uint32_t below = gz_mbcnt ((uint64_t)mlo);
                used_v += 2 * (below + (uint32_t)__popcll ((unsigned long long)((mlo >> (lane & 31)) & hi_bits)));   // every lane 0..3: the whole count
                const uint32_t off = emit ? cap - used_v + 2 * below : 2u * (uint32_t)(lane & 3);
                gz_stg_u16 (buf + off, x);                               // (every lane: lanes 4.. never emit, they hit the dump slots too - no exec window, no branch)
                x = emit ? x >> 16 : x;
                const uint32_t xn = d_rans_advance (x, s);
                x = r < len_eff ? xn : x;
            };
            if (nb == 16) {
                #pragma unroll
                for (uint32_t j = 0; j < 16; j++) one_round (j);
            }
            else for (uint32_t j = 0; j < nb; j++) one_round (j);
            used = d_uniform_u32 (used_v);
            r_hi -= nb;
            continue;
        }
        for (uint32_t j = 0; j < nb; j++) {
            const uint32_t r = r_hi - 1 - j;
            const bool mine = lane < 4 && r < len_k;
            GzRansSym s;
            s.x_max = cur.x; s.rcp = cur.y; s.bias = cur.z; s.cmpl_rsh = cur.w;
            cur = slot[b * 64 + ((j + 1) & 15) * 4 + (lane & 3)];
            bool emit = mine && x >= s.x_max;
            uint64_t m = __ballot (emit) & 0xfull;
            uint32_t cnt = __popcll (m);
            if (used + 2 * cnt + 16 > cap) { overflow = true; break; }
            used += 2 * cnt;
            if (emit) {
                uint32_t below = __popcll (m & ((1ull << lane) - 1));       // emitting states with a smaller index
                gz_stg_u16 (buf + cap - used + 2 * below, x);
                x >>= 16;
            }
            if (mine) x = d_rans_advance (x, s);
        }
        r_hi -= nb;
    }
    if (overflow) return 0xffffffffu;
    used += 16;
    if (lane < 4) {
        uint8_t *p = buf + cap - used + 4 * lane;
        p[0] = (uint8_t)x; p[1] = (uint8_t)(x >> 8); p[2] = (uint8_t)(x >> 16); p[3] = (uint8_t)(x >> 24);
    }
    return used;
}

// order-0 frequency table + encoder records for `n` bytes whose histogram is in F (256 counters, clobbered).
// Serial (one thread). Returns table length. rANS_static4x16pr.c:405-432
__device__ static uint32_t d_o0_table (uint32_t *F, uint32_t n, uint8_t *tab, GzRansSym *syms)
{
    uint32_t stored = gz_pow2_ceil (n);
    if (stored > 4096) stored = 4096;
    d_freq_scale (F, n, stored);
    uint32_t p = d_alphabet_put (tab, F);
    for (int s = 0; s < 256; s++) if (F[s]) p += gz_vi_put (tab + p, F[s]);
    d_freq_scale (F, stored, 4096);
    uint32_t cum = 0;
    for (int s = 0; s < 256; s++) if (F[s]) { syms[s] = d_rans_sym (cum, F[s], 12); cum += F[s]; }
    return p;
}

// 2 x 257 doubles: log(1024 + k), log(4096 + k) computed by the HOST libm at start-up (bit-identical to what the
// reference's compute_shift gets from log(), rANS_static4x16pr.c:647-648)
struct GzLogTable { double l10[257], l12[257]; };

__device__ static inline double d_log_from_bits (int v)   // rANS_static4x16pr.c:617-620
{
    double a = (double)v;
    long long bits = __double_as_longlong (a);
    return (double)(bits - 4606921278410026770LL) * 1.539095918623324e-16;
}

// -DGZ_TABLE_DEBUG: where a workgroup's time goes (thread 0's 100 MHz clock at the phase boundaries; sums and maxima over the launch's
// order-1 workgroups, printed by gz_wait): 0 row totals, 1 compute_shift without 2, 2 its serial entropy sum, 3 rows (normalise,
// serialise, records), 4 table assembly, 5 nested histogram + table, 6 nested coding, 7 the rest
#ifdef GZ_TABLE_DEBUG
__device__ unsigned long long g_tab_sum[9], g_tab_max[9];
#define TAB_T(k) do { if (!tid) { const unsigned long long now_ = wall_clock64 (); ph_[k] += now_ - t_; t_ = now_; } } while (0)
#else
#define TAB_T(k) do { } while (0)
#endif

// one 256-thread workgroup per rANS leaf
__global__ void __launch_bounds__(256) k_rans_table (GzdLeaf *leaves, const GzLogTable *logs)
{
    GzdLeaf &L = leaves[blockIdx.x];
    if (!L.active || L.engine != GZ_ENG_RANS) return;
    const int tid = threadIdx.x;
    const uint32_t n = L.coded_n;
    if (!n) { if (!tid) L.tab_len = 0; return; }

    uint32_t *lds = (uint32_t *)gz_lds;
#ifdef GZ_TABLE_DEBUG
    unsigned long long ph_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, t_ = wall_clock64 ();
#endif

    if (!L.o1) {
        lds[tid] = L.F[tid];
        __syncthreads ();
        if (!tid) L.tab_len = d_o0_table (lds, n, L.tab, L.syms);
        return;
    }

    // ---------- order 1 ----------
    // Everything that is elementwise or a reduction runs wave-parallel (a wave per context row, lanes over the symbols);
    // what the reference's arithmetic makes order-dependent - the entropy sums of compute_shift, whose roundings depend
    // on the order of accumulation, and the zero-run serialisation of a row - is done by one lane, from LDS.
    uint32_t *present = lds;            // [256] 0/1, with [0] forced
    uint32_t *T       = lds + 256;      // [256] row totals
    uint32_t *target  = lds + 512;      // [256] stored total per row (S[] of compute_shift)
    uint32_t *rowlen  = lds + 768;      // [256] serialised bytes per row, later exclusive offsets
    uint32_t *shared  = lds + 1024;     // [0] bits  [1] tab cursor  [8..16) per-wave partial counts
    uint32_t *cellF   = lds + 1088;     // [256] the row's counts of the present symbols (compute_shift)
    double   *cellD   = (double *)(lds + 1088 + 256);            // [256][2] d10, d12
    const int wave = tid >> 6, lane = tid & 63;
    uint32_t *wrow    = lds + 1088 + 256 + 1024 + wave * 448;    // this wave's row: 256 counts + 768 serialised bytes
    uint8_t  *wbytes  = (uint8_t *)(wrow + 256);
    uint32_t *F = L.F;
    const uint32_t ns = L.nsym;

    present[tid] = (L.symrank[tid] != 0xffff) || tid == 0;
    T[tid] = 0; target[tid] = 0; rowlen[tid] = 0;
    __syncthreads ();
    // row totals: a wave per row, coalesced (only rows of present symbols can be non-empty)
    for (int c = wave; c < 256; c += 4) {
        if (!present[c]) continue;
        uint32_t t = F[c * 256 + lane] + F[c * 256 + 64 + lane] + F[c * 256 + 128 + lane] + F[c * 256 + 192 + lane];
        for (int m = 32; m; m >>= 1) t += (uint32_t)__shfl ((int)t, lane ^ m);
        if (!lane) T[c] = t;
    }
    __syncthreads ();
    TAB_T (0);

    // ---- compute_shift (rANS_static4x16pr.c:626-687). The entropy sums must be accumulated in the reference's
    //      order with the reference's roundings (fused e -= f*d, see oracle/gz_oracle.c o1_choose_bits): per row, thread k
    //      prepares the terms of present symbol k, then thread 0 adds them up in symbol order.
    {
        double e10 = 0, e12 = 0;
        uint32_t widest = 0;
        for (int c = 0; c < 256; c++) {
            if (!present[c]) continue;                                   // (uniform: present[] is shared)
            const uint32_t tc = T[c];
            uint32_t cap = gz_pow2_ceil (tc);
            const uint32_t f = (uint32_t)tid < ns ? F[c * 256 + L.symlist[tid]] : 0;
            const uint32_t ratio = f ? cap / f : 0;
            const uint64_t m10 = __ballot (f && ratio > 1024), m12 = __ballot (f && ratio > 4096), mc = __ballot (f != 0);
            if (!lane) { shared[8 + wave] = (uint32_t)__popcll (m10) | ((uint32_t)__popcll (m12) << 10) | ((uint32_t)__popcll (mc) << 20); }
            __syncthreads ();
            uint32_t b10 = 0, b12 = 0, cnt = 0, slot = 0;
            for (int w = 0; w < 4; w++) { const uint32_t v = shared[8 + w]; b10 += v & 1023; b12 += (v >> 10) & 1023; if (w == wave) slot = cnt; cnt += v >> 20; }
            if (f) {
                // (the row's non-empty cells close together, in symbol order: the sum below is one thread's, and the rows of a wide
                //  order-1 table are mostly empty)
                slot += (uint32_t)__popcll (mc & ((1ull << lane) - 1));
                const double l10 = logs->l10[b10], l12 = logs->l12[b12];
                int x10 = (int)(1024.0 * (double)f / (double)tc), x12 = (int)(4096.0 * (double)f / (double)tc);
                cellF[slot] = f;
                cellD[2 * slot]     = d_log_from_bits (x10 > 1 ? x10 : 1) - l10;
                cellD[2 * slot + 1] = d_log_from_bits (x12 > 1 ? x12 : 1) - l12;
            }
            __syncthreads ();
            TAB_T (1);
            if (!tid) {
                #pragma unroll 4
                for (uint32_t k = 0; k < cnt; k++) {
                    const double fk = (double)cellF[k];
                    e10 = fma (-fk, cellD[2 * k], e10) + 4.0;
                    e12 = fma (-fk, cellD[2 * k + 1], e12) + 6.0;
                }
                if (cnt < 64 && cap > 128) cap /= 2;
                if (cap > 1024)            cap /= 2;
                if (cap > 4096)            cap = 4096;
                target[c] = cap;
                if (cap > widest) widest = cap;
            }
            TAB_T (2);
            __syncthreads ();
        }
        if (!tid) shared[0] = (e10 / e12 < 1.01 || widest <= 1024) ? 10 : 12;
    }
    __syncthreads ();
    TAB_T (1);
    const uint32_t bits = shared[0];

    // ---- per row (a wave each; lane l owns symbols 4l..4l+3): normalise to the stored total (normalise_freq :113-160),
    //      serialise (:292-322), scale to 1 << bits, build the encoder records
    for (int c = wave; c < 256; c += 4) {
        if (!present[c]) continue;
        uint32_t fr[4];
        {
            const uint4 v = *(const uint4 *)(F + c * 256 + 4 * lane);
            fr[0] = v.x; fr[1] = v.y; fr[2] = v.z; fr[3] = v.w;
        }
        uint32_t tot = target[c];
        if (bits == 10 && tot > 1024) tot = 1024;
        uint32_t sum = T[c];
        for (int pass = 0; sum; pass++) {
            const uint64_t mult = d_div_u52 (((uint64_t)tot) << 31, sum) + (uint32_t)((1u << 30) / sum);
            uint32_t big = 0, big_at = 0, new_sum = 0;
            #pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t f = fr[j];
                if (!f) continue;
                if (f > big) { big = f; big_at = 4 * lane + j; }
                uint32_t g = (uint32_t)((f * mult) >> 31);
                if (!g) g = 1;
                fr[j] = g; new_sum += g;
            }
            for (int m = 32; m; m >>= 1) {
                new_sum += (uint32_t)__shfl ((int)new_sum, lane ^ m);
                const uint32_t ob = (uint32_t)__shfl ((int)big, lane ^ m), oa = (uint32_t)__shfl ((int)big_at, lane ^ m);
                if (ob > big || (ob == big && oa < big_at)) { big = ob; big_at = oa; }     // the FIRST of the largest (strict > in the reference)
            }
            int32_t adjust = (int32_t)tot - (int32_t)new_sum;
            const bool mine = (uint32_t)lane == big_at / 4;
            uint32_t fb = (uint32_t)__shfl ((int)fr[big_at & 3], (int)(big_at / 4));
            if (adjust > 0) { if (mine) fr[big_at & 3] += (uint32_t)adjust; break; }
            if (adjust == 0) break;
            const uint32_t need = (uint32_t)(-adjust);
            if (fb > need && (pass == 1 || fb / 2 >= need)) { if (mine) fr[big_at & 3] = fb - need; break; }
            if (pass == 0) { sum = new_sum; continue; }
            // the rare last resort: take it off the other symbols one by one, in symbol order - one lane, through LDS
            wrow[4 * lane] = fr[0]; wrow[4 * lane + 1] = fr[1]; wrow[4 * lane + 2] = fr[2]; wrow[4 * lane + 3] = fr[3];
            gz_wave_sync ();
            if (!lane) {
                adjust += (int32_t)fb - 1;
                wrow[big_at] = 1;
                for (int s2 = 0; adjust && s2 < 256; s2++) {
                    const uint32_t f = wrow[s2];
                    if (f < 2) continue;
                    const int32_t step = (f > (uint32_t)(-adjust)) ? adjust : 1 - (int32_t)f;
                    wrow[s2] = (uint32_t)((int32_t)f + step);
                    adjust -= step;
                }
            }
            gz_wave_sync ();
            fr[0] = wrow[4 * lane]; fr[1] = wrow[4 * lane + 1]; fr[2] = wrow[4 * lane + 2]; fr[3] = wrow[4 * lane + 3];
            break;
        }
        // serialise (:292-322): the counts of the present symbols in symbol order, a run of zero counts as "00, run-1" in front of the
        // next non-zero one (or at the end of the row). Every lane places its own four symbols: Z = the number of zero counts (of present
        // symbols) in front of a symbol is a prefix sum over the lanes, the Z of the last non-zero symbol before a lane a prefix maximum
        // (Z never decreases), and the byte offsets a third prefix sum
        uint32_t len;
        {
            uint32_t pz[4], zc = 0;
            #pragma unroll
            for (int j = 0; j < 4; j++) { pz[j] = (!fr[j] && present[4 * lane + j]) ? 1u : 0u; zc += pz[j]; }
            uint32_t zin = zc;
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl ((int)zin, lane - d < 0 ? 0 : lane - d); if (lane >= d) zin += o; }
            const uint32_t z_all = (uint32_t)__shfl ((int)zin, 63);
            uint32_t z = zin - zc, last = 0;              // z: zero counts in front of my first symbol; last: Z of my last non-zero symbol + 1 (0: none)
            uint32_t zs[4];
            #pragma unroll
            for (int j = 0; j < 4; j++) { zs[j] = z; if (fr[j]) last = z + 1; z += pz[j]; }
            uint32_t lin = last;
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl ((int)lin, lane - d < 0 ? 0 : lane - d); if (lane >= d && o > lin) lin = o; }
            const uint32_t last_all = (uint32_t)__shfl ((int)lin, 63);
            uint32_t prev = (uint32_t)__shfl ((int)lin, lane ? lane - 1 : 0);
            if (!lane) prev = 0;
            prev = prev ? prev - 1 : 0;                   // Z of the last non-zero symbol in front of this lane (none: no zero counted yet)
            uint32_t nb[4], run[4], mine = 0;
            #pragma unroll
            for (int j = 0; j < 4; j++) {
                run[j] = 0; nb[j] = 0;
                if (fr[j]) { run[j] = zs[j] - prev; prev = zs[j]; nb[j] = (run[j] ? 2u : 0u) + gz_vi_len (fr[j]); }
                mine += nb[j];
            }
            uint32_t oin = mine;
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl ((int)oin, lane - d < 0 ? 0 : lane - d); if (lane >= d) oin += o; }
            len = (uint32_t)__shfl ((int)oin, 63);
            uint32_t at = oin - mine;
            #pragma unroll
            for (int j = 0; j < 4; j++) if (fr[j]) {
                if (run[j]) { wbytes[at++] = 0; wbytes[at++] = (uint8_t)(run[j] - 1); }
                at += gz_vi_put (wbytes + at, fr[j]);
            }
            const uint32_t tail = z_all - (last_all ? last_all - 1 : 0);     // zero counts behind the row's last non-zero symbol
            if (tail) { if (!lane) { wbytes[len] = 0; wbytes[len + 1] = (uint8_t)(tail - 1); } len += 2; }
            if (!lane) rowlen[c] = len;
        }
        gz_wave_sync ();
        for (uint32_t i = lane; i < len; i += 64) L.rowbuf[c * GZ_ROW_SLOT + i] = wbytes[i];
        // scale up to 1 << bits (normalise_freq_shift :165-177), cumulative starts, encoder records
        int sh = 0;
        { uint32_t t2 = tot; while (t2 && t2 < (1u << bits)) { t2 <<= 1; sh++; } }
        uint32_t lsum = 0;
        #pragma unroll
        for (int j = 0; j < 4; j++) { fr[j] <<= sh; lsum += fr[j]; }
        uint32_t incl = lsum;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl ((int)incl, lane - d < 0 ? 0 : lane - d); if (lane >= d) incl += o; }
        uint32_t cum = incl - lsum;
        GzRansSym *rs = L.syms + c * 256 + 4 * lane;
        #pragma unroll
        for (int j = 0; j < 4; j++) { rs[j] = d_rans_sym (cum, fr[j], bits); cum += fr[j]; }
    }
    __syncthreads ();
    TAB_T (3);
    if (!tid) {
        uint32_t p = 0;
        L.tab[p++] = (uint8_t)(bits << 4);
        p += d_alphabet_put (L.tab + p, present);
        for (int c = 0; c < 256; c++) { uint32_t l = rowlen[c]; rowlen[c] = p; p += l; }
        shared[1] = p;
    }
    __syncthreads ();
    {
        uint32_t end = tid == 255 ? shared[1] : rowlen[tid + 1];
        const uint8_t *srcb = L.rowbuf + tid * GZ_ROW_SLOT;
        for (uint32_t o = rowlen[tid], k = 0; o < end; o++, k++) L.tab[o] = srcb[k];
    }
    __syncthreads ();
    TAB_T (4);
    uint32_t tab_len = shared[1];

    // ---- a table of more than 1000 bytes is itself order-0 coded if that saves at least 6 bytes (:779-792)
    if (tab_len > 1000) {
        const uint32_t raw = tab_len - 1;
        const uint8_t *tin = L.tab + 1;
        uint32_t *h = lds;                            // reuse: [256] histogram
        GzRansSym *tsyms = (GzRansSym *)(lds + 2048); // [256] records (4 KB) in LDS while thread 0 builds them
        __syncthreads ();
        h[tid] = 0;
        __syncthreads ();
        for (uint32_t i = tid; i < raw; i += 256) atomicAdd (&h[tin[i]], 1u);
        __syncthreads ();
        uint8_t *ttab = L.rowbuf;                     // nested table (< 1 KB), nested payload behind it
        if (!tid) shared[2] = d_o0_table (h, raw, ttab, tsyms);
        __syncthreads ();
        const uint32_t ttab_len = shared[2];
        TAB_T (5);
        uint8_t *tpay = L.rowbuf + 1024;
        const uint32_t tpay_cap = 256 * GZ_ROW_SLOT - 1024;
        // (coded by the stream coder itself, d_rans_encode_wave with its look-ahead: its records come from global memory, so the 256
        //  records go behind the row slots of rowbuf; a coder of its own with the records in LDS took 0.2 us per round - 2.6 of the 9 ms
        //  of the slowest workgroup of a file's first call, measured inside the kernel)
        GzRansSym *gsyms = (GzRansSym *)(L.rowbuf + 256 * GZ_ROW_SLOT);
        gsyms[tid] = tsyms[tid];
        __threadfence_block ();
        __syncthreads ();
        if (tid < 64) {
            uint32_t len_k = (raw >> 2) + ((raw & 3) > (uint32_t)(tid & 3));
            uint32_t rounds = (raw + 3) >> 2;
            uint32_t plen = d_rans_encode_wave (tid < 4 ? len_k : 0, rounds, tpay, tpay_cap, (uint8_t *)(lds + 3072), gsyms,
                                                [&] (int k, uint32_t r, uint32_t &hi, uint32_t &lo) { hi = 0; lo = gz_ldg_u8 (tin + 4 * r + k); });
            if (!tid) shared[3] = plen;
        }
        __syncthreads ();
        TAB_T (6);
        const uint32_t plen = shared[3];
        if (plen != 0xffffffffu && ttab_len + plen + 6 < tab_len) {
            const uint32_t packed = ttab_len + plen;
            uint32_t p = 1;
            if (!tid) {
                L.tab[0] |= 1;
                p += gz_vi_put (L.tab + p, raw);
                p += gz_vi_put (L.tab + p, packed);
                shared[4] = p;
            }
            __syncthreads ();     // everybody has finished reading the raw table (the encode above is complete)
            p = shared[4];
            for (uint32_t i = tid; i < ttab_len; i += 256) L.tab[p + i] = ttab[i];
            for (uint32_t i = tid; i < plen; i += 256)     L.tab[p + ttab_len + i] = tpay[tpay_cap - plen + i];
            tab_len = p + packed;
        }
    }
    if (!tid) { L.tab_len = tab_len; L.shift_bits = (uint8_t)bits; }
    TAB_T (7);
#ifdef GZ_TABLE_DEBUG
    if (!tid) {
        unsigned long long tot = 0;
        for (int k = 0; k < 8; k++) { tot += ph_[k]; atomicAdd (&g_tab_sum[k], ph_[k]); atomicMax (&g_tab_max[k], ph_[k]); }
        atomicAdd (&g_tab_sum[8], 1ull); atomicMax (&g_tab_max[8], tot);
    }
#endif
}

// ======================================================================================================
// k_rans_encode : one wave per leaf
// ======================================================================================================
__global__ void __launch_bounds__(64) k_rans_encode (GzdLeaf *leaves)
{
    GzdLeaf &L = leaves[blockIdx.x];
    if (!L.active || L.engine != GZ_ENG_RANS) return;
    const int lane = threadIdx.x;
    const uint32_t n = L.coded_n;
    if (!n) { if (!lane) L.pay_len = 0; return; }
    const uint8_t *in = L.coded;
    const GzRansSym *syms = L.syms;
    uint32_t plen;

    if (!L.o1) {
        uint32_t len_k = (n >> 2) + ((n & 3) > (uint32_t)(lane & 3));
        plen = d_rans_encode_wave (lane < 4 ? len_k : 0, (n + 3) >> 2, L.pay, L.pay_cap, gz_lds, syms,
                                   [&] (int k, uint32_t r, uint32_t &hi, uint32_t &lo) { hi = 0; lo = gz_ldg_u8 (in + 4 * r + k); });
    }
    else {
        // quarter k = [k*q, (k+1)*q), the last one extends to n and codes its surplus first, alone (:817-823);
        // the head of each quarter is coded in context 0 (:843-846)
        const uint32_t q = n >> 2;
        uint32_t len_k = lane == 3 ? n - 3 * q : q;
        plen = d_rans_encode_wave (lane < 4 ? len_k : 0, n - 3 * q, L.pay, L.pay_cap, gz_lds, syms,
                                   [&] (int k, uint32_t r, uint32_t &hi, uint32_t &lo) {
                                       uint32_t at = k * q + r;
                                       hi = r ? gz_ldg_u8 (in + at - 1) : 0u; lo = gz_ldg_u8 (in + at);
                                   });
    }
    if (!lane) { if (plen == 0xffffffffu) { L.overflow = 1; L.pay_len = 0; } else L.pay_len = plen; }
}

// (the arithmetic coder's encoder is gz_kernels_arith.h; what is left here is shared with the decoder, which keeps its
//  models in LDS)
#define GZ_ARITH_RUN_MODELS 258
#define GZ_ARITH_RUN_STRIDE 5      // tot + 4 slots (MAX_RUN == 4, arith_dynamic.c:384)

__device__ static inline uint32_t gz_arith_model_words (uint32_t max_sym, bool o1, bool rle)
{
    uint32_t w = (o1 ? max_sym : 1) * (max_sym + 1);
    if (rle) w += GZ_ARITH_RUN_MODELS * GZ_ARITH_RUN_STRIDE;
    return w;
}

// ======================================================================================================
// k_select : one thread per stream -- CAT fallback per leaf, best method per plane, stream size
// ======================================================================================================
__global__ void k_select (GzdStream *streams, GzdLeaf *leaves, uint32_t n_streams)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_streams) return;
    GzdStream &S = streams[i];
    if (S.status != GZ_ST_PENDING) return;

    if (S.pre) {                                                         // coded ahead, possibly on another handle: its length exists only now
        uint32_t n = S.in_len;
        if (S.in_len_dev) { const uint32_t v = *(volatile const uint32_t *)S.in_len_dev; if (v < n) n = v; }
        S.n = n; S.out_len = n;
        return;
    }
    if (S.engine == GZ_ENG_NONE) { S.out_len = S.n; return; }            // codec_none_compress

    uint32_t best_len[4] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu };
    uint32_t whole_len = 0;
    for (uint32_t k = 0; k < S.n_leaves; k++) {
        GzdLeaf &L = leaves[S.first_leaf + k];
        if (!L.active) continue;
        if (L.overflow == 2) { S.status = GZ_ST_FAILED; S.out_len = 0; return; }   // (the chain never heard from the models)
        uint32_t body = L.tab_len + L.pay_len;
        bool cat = L.overflow || body >= L.coded_n;                       // :1343-1348 / arith :845-850
        if (cat) {
            L.cat = 1;
            uint8_t f = L.prefix[0];
            f = (uint8_t)((f & ~3) | GZ_X_CAT);                           // PACK and NOSZ bits stay
            L.prefix[0] = f;
            body = L.coded_n;
        }
        L.unit_len = L.prefix_len + body;
        if (L.plane == 0xff) whole_len = L.unit_len;
        else if (L.unit_len < best_len[L.plane]) { best_len[L.plane] = L.unit_len; S.best_leaf[L.plane] = (uint8_t)k; }  // first smallest wins
    }
    if (!S.striped) { S.out_len = whole_len; S.best_leaf[0] = 0xff; return; }

    uint32_t total = 1 + gz_vi_len (S.n) + 1;
    for (int k = 0; k < 4; k++) { S.plane_unit_len[k] = best_len[k]; total += gz_vi_len (best_len[k]) + best_len[k]; }
    S.out_len = total;
}

// ======================================================================================================
// k_vb_layout : one thread per VBlock -- offsets of the sections inside z_data
// ======================================================================================================
__global__ void k_vb_layout (GzdVB *vbs, GzdStream *streams, uint32_t n_vbs)
{
    uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_vbs) return;
    GzdVB &V = vbs[v];
    uint64_t off = 84;                                   // SectionHeaderVbHeader first (zip.c:560)
    bool failed = false;
    uint32_t written_before_mark = 0;
    for (uint32_t k = 0; k < V.n_streams; k++) {
        GzdStream &S = streams[V.first_stream + k];
        if (S.status == GZ_ST_FAILED) failed = true;
        S.z_off = off;
        if (!S.n && S.in_len_dev) continue;              // generated on the device and dropped there (b250.c:270-277): no section
        off += 40 + (uint64_t)S.out_len;
        if (k < V.mark_stream) written_before_mark++;
    }
    V.mark_index = written_before_mark;
    V.z_len = off;
    V.status = failed ? GZ_ST_FAILED : off <= V.z_cap ? GZ_ST_OK : GZ_ST_TOO_SMALL;

    uint8_t *z = V.z_data;                               // zfile_compress_vb_header + zfile_update_compressed_vb_header
    if (V.status != GZ_ST_OK) return;
    for (int k = 0; k < 84; k++) z[k] = 0;
    gz_be32 (z + 0, 0x27052012u);
    gz_be32 (z + 4, 1);                                  // adler32 of an empty payload
    gz_be32 (z + 20, V.vblock_i);
    z[24] = 9; z[25] = 1; z[27] = V.vb_flags;
    gz_be32 (z + 36, V.recon_size);
    gz_be32 (z + 40, (uint32_t)off);                     // z_data_bytes
    gz_be32 (z + 44, V.longest_line_len);
    for (int k = 0; k < 16; k++) z[48 + k] = V.digest[k];
    gz_be32 (z + 80, V.longest_seq_len);
}

// ======================================================================================================
// k_emit : one 256-thread workgroup per stream -- writes the payload (and, in VBlock mode, the section header)
// ======================================================================================================
// 16 bytes per thread and access (neither side is aligned in general: the hardware takes unaligned vector accesses)
// (four loads ahead of their stores: source and destination may overlap for all the compiler knows, so it would not read ahead itself,
//  and one workgroup copying a 3 MB section 4 KB per trip to memory was 0.7 ms at the very end of the step)
__device__ static inline void d_copy (uint8_t *dst, const uint8_t *src, uint32_t n, int tid)
{
    const uint32_t body = n & ~15u, step = 256 * 16;
    uint32_t i = (uint32_t)tid * 16;
    for (; i + 3 * step < body; i += 4 * step) {
        const gz_u32x4_unaligned v0 = *(const gz_u32x4_unaligned *)(src + i), v1 = *(const gz_u32x4_unaligned *)(src + i + step),
                                 v2 = *(const gz_u32x4_unaligned *)(src + i + 2 * step), v3 = *(const gz_u32x4_unaligned *)(src + i + 3 * step);
        *(gz_u32x4_unaligned *)(dst + i) = v0; *(gz_u32x4_unaligned *)(dst + i + step) = v1;
        *(gz_u32x4_unaligned *)(dst + i + 2 * step) = v2; *(gz_u32x4_unaligned *)(dst + i + 3 * step) = v3;
    }
    for (; i < body; i += step) *(gz_u32x4_unaligned *)(dst + i) = *(const gz_u32x4_unaligned *)(src + i);
    for (uint32_t j = body + tid; j < n; j += 256) dst[j] = src[j];
}

// bytes [lo, hi) of the payload only (a long section's bytes go out by several workgroups): piece = n bytes of src that belong at dst + at
__device__ static inline void d_copy_part (uint8_t *dst, uint32_t at, const uint8_t *src, uint32_t n, uint32_t lo, uint32_t hi, int tid)
{
    const uint32_t a = at > lo ? at : lo, b = at + n < hi ? at + n : hi;
    if (a < b) d_copy (dst + a, src + (a - at), b - a, tid);
}

__device__ static void d_emit_unit (uint8_t *dst, uint32_t at, const GzdLeaf &L, uint32_t lo, uint32_t hi, int tid)
{
    d_copy_part (dst, at, L.prefix, L.prefix_len, lo, hi, tid);
    at += L.prefix_len;
    if (L.cat) { d_copy_part (dst, at, L.coded, L.coded_n, lo, hi, tid); return; }
    d_copy_part (dst, at, L.tab, L.tab_len, lo, hi, tid);
    const uint8_t *pay = L.engine == GZ_ENG_RANS ? L.pay + L.pay_cap - L.pay_len : L.pay;
    d_copy_part (dst, at + L.tab_len, pay, L.pay_len, lo, hi, tid);
}

// grid (streams, GZ_EMIT_SLICES). A section of GZ_EMIT_LONG bytes or more (the 3 MB QUAL sections at the very end of a step: one workgroup
// took 0.5 ms over each) is written by all GZ_EMIT_SLICES workgroups of its column, a stretch of the payload each: the adler32 sums of the
// stretches add up (gz_adler32_part), the last one through writes the header. Everything else is slice 0's alone.
#ifndef GZ_EMIT_SLICES
#define GZ_EMIT_SLICES 8
#endif
#define GZ_EMIT_LONG   (256u * 1024u)
__global__ void __launch_bounds__(256) k_emit (GzdStream *streams, GzdLeaf *leaves, GzdVB *vbs)
{
    GzdStream &S = streams[blockIdx.x];
    const int tid = threadIdx.x;
    if (S.status != GZ_ST_PENDING) return;
    const uint32_t slices = (S.vb >= 0 && S.out_len >= GZ_EMIT_LONG) ? GZ_EMIT_SLICES : 1, slice = blockIdx.y;
    if (slice >= slices) return;

    uint8_t *dst = S.out;
    if (S.vb >= 0) {
        GzdVB &V = vbs[S.vb];
        if (V.status != GZ_ST_OK) { if (!tid && !slice) S.status = GZ_ST_TOO_SMALL; return; }
        if (!S.n && S.in_len_dev) { if (!tid) S.status = GZ_ST_OK; return; }     // dropped: k_vb_layout left no room for it
        dst = V.z_data + S.z_off + 40;
    }
    else if (S.out_len > S.out_cap) { if (!tid) S.status = GZ_ST_TOO_SMALL; return; }

    // this workgroup's stretch of the payload (whole 16-byte groups; the last slice takes the rest)
    const uint32_t per = slices > 1 ? ((S.out_len / slices + 15) & ~15u) : 0xffffffffu;
    const uint32_t lo = slices > 1 ? slice * per : 0, hi = slices > 1 ? (slice + 1 == slices ? S.out_len : (lo + per < S.out_len ? lo + per : S.out_len)) : 0xffffffffu;

    if (S.engine == GZ_ENG_NONE) d_copy_part (dst, 0, S.in, S.n, lo, hi, tid);
    else if (!S.striped) d_emit_unit (dst, 0, leaves[S.first_leaf + S.whole_leaf], lo, hi, tid);
    else {
        // [order & ~NOSZ][varint n][4][varint unit length x4][unit x4]   (rANS_static4x16pr.c:1194-1225)
        uint8_t meta[32]; uint32_t p = 0;
        meta[p++] = (uint8_t)(S.order & ~GZ_X_NOSZ);
        p += gz_vi_put (meta + p, S.n);
        meta[p++] = 4;
        for (int k = 0; k < 4; k++) p += gz_vi_put (meta + p, S.plane_unit_len[k]);
        if (tid < (int)p && (uint32_t)tid >= lo && (uint32_t)tid < hi) dst[tid] = meta[tid];
        uint32_t o = p;
        for (int k = 0; k < 4; k++) {
            d_emit_unit (dst, o, leaves[S.first_leaf + S.best_leaf[k]], lo, hi, tid);
            o += S.plane_unit_len[k];
        }
    }

    // NB: status doubles as the entry test of this kernel: every thread (of every slice) must be past it before it changes
    if (S.vb < 0) { __syncthreads (); if (!tid) { if (S.out_len_dev) *S.out_len_dev = S.out_len; S.status = GZ_ST_OK; } return; }

    // ---- section header (comp_compress, compressor.c:114-161): adler32 of the payload by the whole workgroup
    __threadfence_block ();
    __syncthreads ();
    uint32_t adler;
    if (slices == 1) adler = gz_adler32_wg (dst, S.out_len, tid);
    else {
        uint32_t a, w;
        gz_adler32_part (dst, S.out_len, lo, hi < S.out_len ? hi : S.out_len, tid, &a, &w);
        uint32_t *last = (uint32_t *)gz_lds + 520;                  // (behind what gz_adler32_part uses of the LDS)
        if (!tid) {
            atomicAdd (&S.emit_a, a); atomicAdd (&S.emit_w, w);
            __threadfence ();
            *last = atomicAdd (&S.emit_done, 1u) + 1 == slices;
        }
        __syncthreads ();
        if (!*last) return;
        __threadfence ();
        const uint32_t ta = *(volatile uint32_t *)&S.emit_a, tw = *(volatile uint32_t *)&S.emit_w;
        adler = (((S.out_len % 65521u + tw) % 65521u) << 16) | ((1u + ta) % 65521u);
    }
    if (!tid) {
        uint8_t *h = vbs[S.vb].z_data + S.z_off;
        for (int k = 0; k < 40; k++) h[k] = S.hdr[k];
        gz_be32 (h + 4,  adler);
        gz_be32 (h + 12, S.out_len);
        gz_be32 (h + 16, S.pre ? S.raw_len : S.n);
        if (S.hdr_codec) { h[25] = S.hdr_codec; h[26] = S.codec; }          // codec_domq.c:487-500: header->sub_codec
        else h[25] = S.codec;
        S.status = GZ_ST_OK;
    }
}
