// gz_kernels_dec.h -- gfx950 kernels of the decompression direction (codec_rans_uncompress / codec_arith_uncompress,
// src/codec_htscodecs.c:100,116 -> rans_uncompress_to_4x16 rANS_static4x16pr.c:1358, arith_uncompress_to
// arith_dynamic.c:860). Needed for the round-trip proof of the encoder.
//
//   k_dec_parse    one thread per stream: flags, sizes, stripe / pack meta -> up to 4 decode leaves
//   k_dec_table    rANS: frequency tables -> reverse lookup tables (incl. the nested order-0 coded order-1 table)
//   k_rans_decode  4 states on 4 lanes of one wave, shared forward read pointer
//   k_arith_decode lane 0, models in LDS
//   k_dec_finish   CAT copies, unpack (pack.c:214-351), unstripe (utils.h:41-73), status
#pragma once
#include "gz_device.h"
#include "gz_devutil.h"

__device__ static bool d_parse_unit (GzdDecStream &S, GzdDecLeaf &L, const uint8_t *u, uint32_t unit_len,
                                     uint32_t expect_n, uint8_t *final_dst, uint8_t *packed_tmp, bool rans)
{
    if (!unit_len) return false;
    uint32_t q = 1, v;
    const uint8_t flag = u[0];
    if (flag & GZ_X_STRIPE) return false;                            // no nested striping
    if (!(flag & GZ_X_NOSZ)) {
        uint32_t used = gz_vi_get (u + q, unit_len - q, &v);
        if (!used || v != expect_n) return false;
        q += used;
    }
    L.engine = rans ? GZ_ENG_RANS : GZ_ENG_ARITH;
    L.o1  = rans ? (flag & 1) : ((flag & 3) == 1);
    L.rle = (flag & GZ_X_RLE) ? 1 : 0;
    L.cat = (flag & GZ_X_CAT) ? 1 : 0;
    if (rans && L.rle) return false;                                 // rle.c streams are never written by Genozip
    if (!rans && (flag & 4)) return false;                           // X_EXT
    L.n = expect_n; L.final_dst = final_dst;
    L.packed_on = 0; L.per = 1;
    uint32_t coded_n = expect_n;
    uint8_t *dst = final_dst;
    if (flag & GZ_X_PACK) {
        if (q >= unit_len) return false;
        uint32_t ns = u[q] ? u[q] : 256;
        L.packed_on = 1;
        if (ns > 16) { L.per = 1; q += 1; }
        else {
            L.per = ns <= 1 ? 0 : ns <= 2 ? 8 : ns <= 4 ? 4 : 2;
            if (q + 1 + ns > unit_len) return false;
            for (uint32_t k = 0; k < ns; k++) L.map[k] = u[q + 1 + k];
            q += 1 + ns;
        }
        uint32_t used = gz_vi_get (u + q, unit_len - q, &coded_n);
        if (!used || coded_n > expect_n) return false;
        q += used;
        dst = packed_tmp;
    }
    L.dst = dst; L.coded_n = coded_n;
    L.body = u + q; L.body_len = unit_len - q;
    if (!L.body_len) L.coded_n = 0;                                  // "in_size == 0": nothing was coded (:1583-1601)
    L.status = GZ_ST_PENDING;
    L.active = 1;
    return true;
}

__global__ void k_dec_parse (GzdDecStream *streams, GzdDecLeaf *leaves, uint32_t n_streams)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_streams) return;
    GzdDecStream &S = streams[i];
    if (S.status != GZ_ST_PENDING) return;
    GzdDecLeaf *L = leaves + S.first_leaf;
    for (int k = 0; k < 4; k++) L[k].active = 0;
    S.striped = 0; S.n_leaves = 0;
    if (S.codec == 1) { if (S.in_len != S.out_len) S.status = GZ_ST_CORRUPT; return; }
    const bool rans = S.codec >= 6 && S.codec <= 9;
    S.engine = rans ? GZ_ENG_RANS : GZ_ENG_ARITH;
    if (!S.in_len) { S.status = GZ_ST_CORRUPT; return; }
    const uint8_t *in = S.in;
    bool ok = true;
    if (in[0] & GZ_X_STRIPE) {
        uint32_t p = 1, ulen, used, clen[4], len[4], off[4];
        used = gz_vi_get (in + p, S.in_len - p, &ulen);
        ok = used && ulen == S.out_len;
        p += used;
        ok = ok && p < S.in_len && in[p] == 4;
        p++;
        for (int k = 0; ok && k < 4; k++) { used = gz_vi_get (in + p, S.in_len - p, &clen[k]); ok = used && clen[k] >= 1; p += used; }
        if (ok) {
            gz_plane_geometry (ulen, len, off);
            for (int k = 0; ok && k < 4; k++) {
                // (p <= in_len holds here; compared without the addition: a crafted 5-byte varint must not wrap)
                ok = clen[k] <= S.in_len - p &&
                     d_parse_unit (S, L[k], in + p, clen[k], len[k], S.tmp_planes + off[k], S.tmp_packed + off[k], rans);
                p += clen[k];
            }
            S.striped = 1; S.n_leaves = 4;
        }
    }
    else {
        ok = d_parse_unit (S, L[0], in, S.in_len, S.out_len, S.out, S.tmp_packed, rans);
        S.n_leaves = 1;
    }
    if (!ok) { S.status = GZ_ST_CORRUPT; for (int k = 0; k < 4; k++) L[k].active = 0; }
}

// ---- frequency table parsing -------------------------------------------------------------------------------

// symbol list (rANS_static4x16pr.c:205-252) -> present[256]; returns bytes consumed or 0
__device__ static uint32_t d_alphabet_get (const uint8_t *src, uint32_t avail, uint32_t *present)
{
    uint32_t p = 0;
    if (!avail) return 0;
    int run = 0, s = src[p++];
    for (;;) {
        present[s] = 1;
        if (run) { run--; s++; if (s > 255) return 0; }
        else {
            if (p >= avail) return 0;
            if (s + 1 == src[p]) { if (p + 1 >= avail) return 0; s = src[p++]; run = src[p++]; }
            else s = src[p++];
        }
        if (!s) break;
    }
    return p;
}

// order-0 table at `src` -> 4096-entry lookup (sym | (freq-1) << 8 | offset << 20). One thread. Returns bytes
// consumed or 0. F is a 256-word scratch.
__device__ static uint32_t d_o0_lut (const uint8_t *src, uint32_t avail, uint32_t *F, uint32_t *lut)
{
    for (int s = 0; s < 256; s++) F[s] = 0;
    uint32_t p = d_alphabet_get (src, avail, F), sum = 0;
    if (!p) return 0;
    for (int s = 0; s < 256; s++)
        if (F[s]) {
            uint32_t used = gz_vi_get (src + p, avail - p, &F[s]);
            if (!used) return 0;
            p += used; sum += F[s];
        }
    if (!sum || sum > 4096) return 0;
    int sh = 0;
    while ((sum << sh) < 4096) sh++;
    if ((sum << sh) != 4096) return 0;
    uint32_t c = 0;
    for (int s = 0; s < 256; s++) {
        uint32_t f = F[s] << sh;
        for (uint32_t k = 0; k < f; k++) lut[c + k] = (uint32_t)s | ((f - 1) << 8) | (k << 20);
        c += f;
    }
    return c == 4096 ? p : 0;
}

// The 4 decoder states on lanes 0..3; called by all 64 lanes. sym(k, r, x) consumes state x of lane k at its r-th
// step and returns the new (un-renormalised) state after storing the decoded byte. Renormalisation reads 16-bit
// words from one shared forward pointer in lane order (rANS_word.h:380-409).
template <typename StepFn>
__device__ static bool d_rans_decode_wave (const uint8_t *pay, uint32_t pay_len, uint32_t len_k, uint32_t rounds, StepFn step)
{
    const int lane = threadIdx.x & 63;
    if (pay_len < 16) return false;
    uint32_t x = 0;
    if (lane < 4) {
        const uint8_t *p = pay + 4 * lane;
        x = p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24);
    }
    bool bad = lane < 4 && x < 0x8000u;
    if (__ballot (bad)) return false;
    uint32_t pos = 16;
    for (uint32_t r = 0; r < rounds; r++) {
        bool mine = lane < 4 && r < len_k;
        if (mine) x = step (lane, r, x);
        bool need = mine && x < 0x8000u;
        uint64_t m = __ballot (need) & 0xfull;
        if (need) {
            uint32_t at = pos + 2 * __popcll (m & ((1ull << lane) - 1));
            if (at + 1 < pay_len) x = (x << 16) | pay[at] | (pay[at + 1] << 8);
        }
        pos += 2 * __popcll (m);
    }
    return true;
}

// one 256-thread workgroup per decode leaf; 20 KB dynamic LDS
// Workgroup -> leaf, for the kernels with one workgroup per leaf (four leaves per stream, of which an unstriped stream uses the first): the
// workgroups of a grid go round the 8 XCDs, so with leaf = blockIdx the working leaves 0, 4, 8 ... of a batch of unstriped streams would
// all land on XCDs 0 and 4 - a quarter of the chip (measured: 1024 quality streams decoded 114 at a time instead of 512). Leaf k of stream
// s is workgroup k * n_streams + s instead.
#define GZ_DEC_LEAF_OF_BLOCK ((blockIdx.x % (gridDim.x / 4)) * 4 + blockIdx.x / (gridDim.x / 4))

__global__ void __launch_bounds__(256) k_dec_table (GzdDecLeaf *leaves)
{
    GzdDecLeaf &L = leaves[GZ_DEC_LEAF_OF_BLOCK];
    if (!L.active || L.engine != GZ_ENG_RANS || L.cat || !L.coded_n) return;
    const int tid = threadIdx.x;
    uint32_t *lds = (uint32_t *)gz_lds;
    uint32_t *F = lds;               // [256]
    uint32_t *sh = lds + 256;        // [0] status/consumed ...
    uint32_t *lut0 = lds + 512;      // [4096] nested / order-0 lookup

    if (!L.o1) {
        if (!tid) {
            uint32_t used = d_o0_lut (L.body, L.body_len, F, (uint32_t *)L.lut);
            if (!used) L.status = GZ_ST_CORRUPT;
            else { L.pay = L.body + used; L.pay_len = L.body_len - used; }
        }
        return;
    }

    // ---- order 1 (rANS_static4x16pr.c:953-1016)
    const uint8_t *tab; uint32_t tab_avail;
    if (!tid) {
        sh[0] = 1;
        uint32_t bits = L.body_len ? L.body[0] >> 4 : 0;
        if (bits != 10 && bits != 12) sh[0] = 0;
        sh[1] = bits;
    }
    __syncthreads ();
    if (!sh[0]) { if (!tid) L.status = GZ_ST_CORRUPT; return; }
    const uint32_t bits = sh[1];
    const bool nested = L.body[0] & 1;
    if (nested) {
        if (!tid) {
            uint32_t p = 1, raw = 0, clen = 0, used;
            bool ok = (used = gz_vi_get (L.body + p, L.body_len - p, &raw)) != 0;
            p += used;
            ok = ok && (used = gz_vi_get (L.body + p, L.body_len - p, &clen)) != 0;
            p += used;
            ok = ok && raw <= GZ_TAB_CAP && p + clen + 16 <= L.body_len && clen >= 16;
            uint32_t tused = ok ? d_o0_lut (L.body + p, clen, F, lut0) : 0;
            ok = ok && tused;
            sh[0] = ok; sh[2] = raw; sh[3] = p + tused; sh[4] = clen - tused; sh[5] = p + clen;
        }
        __syncthreads ();
        if (!sh[0]) { if (!tid) L.status = GZ_ST_CORRUPT; return; }
        const uint32_t raw = sh[2];
        const uint8_t *npay = L.body + sh[3];
        uint8_t *dstt = L.tabtmp;
        if (tid < 64) {
            uint32_t len_k = (raw >> 2) + ((raw & 3) > (uint32_t)(tid & 3));
            bool ok = d_rans_decode_wave (npay, sh[4], tid < 4 ? len_k : 0, (raw + 3) >> 2,
                                          [&] (int k, uint32_t r, uint32_t x) {
                                              uint32_t e = lut0[x & 4095];
                                              dstt[4 * r + k] = (uint8_t)e;
                                              return (((e >> 8) & 0xfff) + 1) * (x >> 12) + (e >> 20);
                                          });
            if (!tid && !ok) sh[0] = 0;
        }
        __syncthreads ();
        if (!sh[0]) { if (!tid) L.status = GZ_ST_CORRUPT; return; }
        tab = L.tabtmp; tab_avail = raw;
    }
    else { tab = L.body + 1; tab_avail = L.body_len - 1; }

    // thread 0 walks the variable-length rows; the frequencies land in fc[c][s], then every thread expands its row
    uint32_t *present = F;
    present[tid] = 0;
    __syncthreads ();
    if (!tid) {
        bool ok = true;
        uint32_t p = d_alphabet_get (tab, tab_avail, present);
        ok = p != 0;
        for (int c = 0; ok && c < 256; c++) {
            if (!present[c]) continue;
            uint32_t *row = L.fc + c * 256, zeros = 0;
            for (int s = 0; ok && s < 256; s++) {
                row[s] = 0;
                if (!present[s]) continue;
                if (zeros) { zeros--; continue; }
                uint32_t f, used = gz_vi_get (tab + p, tab_avail - p, &f);
                if (!used) { ok = false; break; }
                p += used;
                if (!f) { if (p >= tab_avail) { ok = false; break; } zeros = tab[p++]; }
                row[s] = f;
            }
        }
        sh[0] = ok;
        sh[6] = p;
    }
    __syncthreads ();
    if (!sh[0]) { if (!tid) L.status = GZ_ST_CORRUPT; return; }
    {
        const int c = tid;
        bool ok = true;
        if (present[c]) {
            uint32_t *row = L.fc + c * 256, sum = 0;
            for (int s = 0; s < 256; s++) sum += row[s];
            if (sum > (1u << bits)) ok = false;
            else if (sum) {
                int shf = 0;
                while ((sum << shf) < (1u << bits)) shf++;
                ok = (sum << shf) == (1u << bits);
                uint32_t cum = 0;
                uint8_t *lrow = L.lut + ((size_t)c << bits);
                for (int s = 0; ok && s < 256; s++) {
                    uint32_t f = row[s] << shf;
                    if (!f) continue;
                    for (uint32_t k = 0; k < f; k++) lrow[cum + k] = (uint8_t)s;
                    row[s] = f | (cum << 16);
                    cum += f;
                }
            }
        }
        if (!ok) L.status = GZ_ST_CORRUPT;
    }
    if (!tid) {
        uint32_t consumed = nested ? sh[5] : 1 + sh[6];
        L.pay = L.body + consumed; L.pay_len = L.body_len - consumed; L.shift_bits = (uint8_t)bits;
    }
}

__global__ void __launch_bounds__(64) k_rans_decode (GzdDecLeaf *leaves)
{
    GzdDecLeaf &L = leaves[GZ_DEC_LEAF_OF_BLOCK];
    if (!L.active || L.engine != GZ_ENG_RANS || L.cat || !L.coded_n || L.status != GZ_ST_PENDING) return;
    const int lane = threadIdx.x;
    const uint32_t n = L.coded_n;
    uint8_t *out = L.dst;
    bool ok;
    if (!L.o1) {
        const uint32_t *lut = (const uint32_t *)L.lut;
        uint32_t len_k = (n >> 2) + ((n & 3) > (uint32_t)(lane & 3));
        ok = d_rans_decode_wave (L.pay, L.pay_len, lane < 4 ? len_k : 0, (n + 3) >> 2,
                                 [&] (int k, uint32_t r, uint32_t x) {
                                     uint32_t e = lut[x & 4095];
                                     out[4 * r + k] = (uint8_t)e;
                                     return (((e >> 8) & 0xfff) + 1) * (x >> 12) + (e >> 20);
                                 });
    }
    else {
        const uint32_t bits = L.shift_bits, mask = (1u << bits) - 1, q = n >> 2;
        const uint8_t *lut = L.lut; const uint32_t *fc = L.fc;
        uint32_t last = 0;
        uint32_t len_k = lane == 3 ? n - 3 * q : q;
        ok = d_rans_decode_wave (L.pay, L.pay_len, lane < 4 ? len_k : 0, n - 3 * q,
                                 [&] (int k, uint32_t r, uint32_t x) {
                                     uint32_t m = x & mask;
                                     uint32_t s = lut[((size_t)last << bits) + m];
                                     uint32_t e = fc[last * 256 + s];
                                     out[k * q + r] = (uint8_t)s;
                                     last = s;
                                     return (e & 0xffff) * (x >> bits) + m - (e >> 16);
                                 });
    }
    if (!lane) L.status = ok ? GZ_ST_OK : GZ_ST_CORRUPT;
}

// ---- adaptive arithmetic decoder (c_range_coder.h:55-68,111-127, c_simple_model.h:148-179) --------------------
// The decoder cannot be taken apart the way the encoder is (models / chain / low): which symbol comes next depends on the coder's
// state - one wave decodes a stream. Two routines share the range decoder below: d_model_decode, round 3's general one (a model row
// [tot, -, (freq | sym << 16, cum) x max_sym] in the LDS or in global memory, read and written back around every symbol, two divisions,
// ballot + v_readlane) - it still serves the run-length models of arith_dynamic.c:476-487, which Genozip never writes - and the literal
// models' d_lit_fast / d_lit_slow further down (round 5), built on measured instruction costs. Measured: DESIGN.md section 3 (Decode).
#define GZ_DEC_BAD 0xffffffffu
// The coded bytes: a window of 256 of them in a register (lane l: the big-endian word at wbase + 4 l, zeros beyond the stream), from which
// W - the next wvalid (a multiple of 8, >= 16 whenever a symbol starts) bits of the stream, at the top of 64 - is topped up a word at a time;
// the coder's normalisation (c_range_coder.h:121-126: while range < 2^24 take a byte) is then ONE shift of { code, W } by 0, 8 or 16 bits.
// (Beyond the end of the stream the reference stops shifting and flags an error; zeros come in here - only a malformed stream gets there.)
struct GzRcDec { uint32_t code, range; uint64_t W; uint32_t wvalid; const uint8_t *in; uint32_t len, fill, wbase, win; };

__device__ static __forceinline__ void d_rc_window (GzRcDec &rc, int lane)
{
    const uint32_t off = rc.wbase + 4 * (uint32_t)lane;
    uint32_t w = 0;
    if (off + 4 <= rc.len) w = __builtin_bswap32 (gz_ldg_u32 ((const uint32_t *)(rc.in + off)));     // (any alignment: a global load)
    else for (uint32_t k = 0; k < 4; k++) w = (w << 8) | (off + k < rc.len ? gz_ldg_u8 (rc.in + off + k) : 0u);
    rc.win = w;
}

__device__ static __forceinline__ void d_rc_refill (GzRcDec &rc, int lane)
{
    if ((int32_t)rc.wvalid < 0) rc.wvalid = 0;                   // (a malformed stream took more than it was entitled to)
    while (rc.wvalid <= 32) {
        if (rc.fill - rc.wbase >= 256) { rc.wbase = rc.fill; d_rc_window (rc, lane); }
        const uint32_t w = d_readlane (rc.win, (int)((rc.fill - rc.wbase) >> 2));
        rc.W |= ((uint64_t)w << 32) >> rc.wvalid;
        rc.wvalid += 32; rc.fill += 4;
    }
}

__device__ static __forceinline__ void d_rc_start (GzRcDec &rc, const uint8_t *in, uint32_t len, int lane)   // RC_StartDecode, c_range_coder.h:55-68
{
    rc.code = 0; rc.range = 0xffffffffu; rc.W = 0; rc.wvalid = 0; rc.in = in; rc.len = len; rc.fill = 0; rc.wbase = 0;
    d_rc_window (rc, lane);
    d_rc_refill (rc, lane);
    rc.code = (uint32_t)(rc.W >> 24); rc.W <<= 40; rc.wvalid -= 40;      // five bytes, the first of which falls out of the 32 bits
    d_rc_refill (rc, lane);
}

__device__ static __forceinline__ uint32_t d_dec_byte (GzRcDec &rc, int lane)
{
    const uint32_t b = (uint32_t)(rc.W >> 56);
    rc.W <<= 8; rc.wvalid -= 8;
    if (rc.wvalid < 16) d_rc_refill (rc, lane);
    return b;
}

// a / b for a < 2^32, 0 < b < 2^32: the hardware's reciprocal seed (v_rcp_f64) + one Newton step is good to ~2^-46, so the truncated
// product is off by at most one - corrected with one 64-bit product (instead of the ~40 instructions of the generic 32-bit division)
__device__ static __forceinline__ uint32_t d_udiv (uint32_t a, uint32_t b)
{
    const double d = (double)b;
    double r = gz_rcp_f64 (d);                                   // (v_rcp_f64 is a seed of about single precision: the division sequence refines it)
    r = __builtin_fma (__builtin_fma (-d, r, 1.0), r, r);        // one Newton step: ~2^-46
    uint32_t q = (uint32_t)((double)a * r);
    const uint64_t p = (uint64_t)q * b;
    q = p > a ? q - 1 : (a - p >= b ? q + 1 : q);
    return q;
}

// one symbol of model m (row layout above) by the whole wave; returns the symbol, or 0 as the reference's search does when the
// target lies beyond the model (a malformed stream: the caller's length / adler checks catch it).
// Per symbol ONE round trip to the LDS: every lane reads its entry (and the total) once, the neighbour a swap needs comes from the
// lane beside it (v_readlane), and every lane then writes its own entry back - lane 0 the total. A lane only ever writes entries it
// owns, so nothing has to be waited for between two symbols but the hardware's in-order LDS.
// MP: a pointer into the LDS (address space 3: ds_read / ds_write) or a generic one (models in global memory). Through a generic pointer
// an LDS access is a FLAT instruction, and every wait for it also waits for the global stores in flight - measured: 760 ns per symbol.
typedef __attribute__((address_space(3))) uint32_t *GzLdsU32P;
template <typename MP>
__device__ static inline uint32_t d_model_decode (MP m, uint32_t max_sym, GzRcDec &rc, int lane)
{
    const uint32_t tot = m[0];
    uint32_t x = 0, c = 0;
    if ((uint32_t)lane < max_sym) { x = m[2 + 2 * lane]; c = m[3 + 2 * lane]; }
    uint32_t target = 0;
    if (tot && rc.range >= tot) { rc.range = d_udiv (rc.range, tot); target = d_udiv (rc.code, rc.range); }
    if (target > 65519) return 0;
    uint32_t plane = 0;
    uint64_t hit = __ballot ((uint32_t)lane < max_sym && c <= target && target - c < (x & 0xffff));
    while (!hit) {                                              // (alphabets beyond 64 symbols: the next 64 entries)
        if (++plane * 64 >= max_sym) return 0;
        const uint32_t e = plane * 64 + lane;
        x = 0; c = 0;
        if (e < max_sym) { x = m[2 + 2 * e]; c = m[3 + 2 * e]; }
        hit = __ballot (e < max_sym && c <= target && target - c < (x & 0xffff));
    }
    const int l = __ffsll ((unsigned long long)hit) - 1;
    const uint32_t at = plane * 64 + l, ex = d_readlane (x, l), ec = d_readlane (c, l);
    rc.code  -= ec * rc.range;
    rc.range *= ex & 0xffff;
    for (int k = 0; k < 3 && rc.range < (1u << 24); k++) {
        rc.code = (rc.code << 8) + d_dec_byte (rc, lane);
        rc.range <<= 8;
    }
    const uint32_t sym = ex >> 16, e_new = ex + 16, t2 = tot + 16;
    if (t2 <= 65519) {
        // the neighbour to the left: the lane beside the symbol's (across a plane boundary: from the LDS)
        uint32_t left = 0xffffu, cl = 0;
        if (l > 0) { left = d_readlane (x, l - 1); cl = d_readlane (c, l - 1); }
        else if (at) { left = m[2 + 2 * (at - 1)]; cl = m[3 + 2 * (at - 1)]; }
        const bool swap = at && (e_new & 0xffff) > (left & 0xffff);      // one bubble step to the left (c_simple_model.h:139-145)
        const uint32_t e = plane * 64 + lane;
        if (e < max_sym) {
            if (e == at) { m[2 + 2 * e] = swap ? left : e_new; m[3 + 2 * e] = swap ? cl + (e_new & 0xffff) : ec; }
            else if (e + 1 == at) { if (swap) m[2 + 2 * e] = e_new; }                                    // (keeps its cumulative)
            else if (e > at) m[3 + 2 * e] = c + 16;
        }
        if (swap && l == 0 && !lane) m[2 + 2 * (at - 1)] = e_new;                                          // (the neighbour lives in the plane before)
        for (uint32_t j = plane + 1; j * 64 < max_sym; j++) {        // the planes behind: cum + 16
            const uint32_t e2 = j * 64 + lane;
            if (e2 < max_sym) m[3 + 2 * e2] += 16;
        }
        if (!lane) m[0] = t2;
        gz_wave_sync ();
        return sym;
    }
    // halve every frequency, rebuild total and cumulatives (every ~2000 symbols of a context: lane 0 walks the list)
    if (!lane) {
        m[2 + 2 * at] = e_new;
        uint32_t run = 0;
        for (uint32_t i = 0; i < max_sym; i++) {
            uint32_t v = m[2 + 2 * i], f = v & 0xffff;
            f -= f >> 1;
            m[2 + 2 * i] = (v & 0xffff0000u) | f; m[3 + 2 * i] = run;
            run += f;
        }
        m[0] = run;
    }
    gz_wave_sync ();
    if (at) {
        const uint32_t mine = m[2 + 2 * at], left = m[2 + 2 * (at - 1)], cl = m[3 + 2 * (at - 1)];
        gz_wave_sync ();
        if ((mine & 0xffff) > (left & 0xffff)) {
            if (!lane) { m[2 + 2 * (at - 1)] = mine; m[2 + 2 * at] = left; m[3 + 2 * at] = cl + (mine & 0xffff); }
            gz_wave_sync ();
        }
    }
    return sym;
}

#define GZ_DEC_ROW(ms) (2 * (ms) + 2)
#define GZ_DEC_RUN_ROW 10
#define GZ_DEC_LIT_ROW(ms) ((ms) <= 64 ? 128u : (ms) <= 128 ? 256u : 512u)   // words of a literal row: 64 entries per register plane, zeros beyond the alphabet - no lane needs a bounds check

// ---- the literal models ---------------------------------------------------------------------------------------------------------------
// ONE wave decodes a stream, and one wave is given an instruction every 4 - 5 clocks whatever the instruction (tools/probes/lat_probe.hip:
// dependent or not, VALU or SALU), a taken branch costs ~26, one not taken ~13, a value written by the VALU into an SGPR can be read by
// the SALU ~20 clocks later, the LDS answers after ~48, a global load after 200+. So a symbol costs what its instructions, its branches and
// its VALU -> SALU hand-overs add up to - d_model_decode above: ~1300 clocks (556 ns) - and the decoder of the literals is built to keep all
// three small:
//   * the current context's row lives in REGISTERS (NP entries per lane: list positions lane, lane + 64 ...): freq | sym << 16 and the
//     cumulative frequency of every entry; it is updated there, goes back to the LDS after every symbol and the next context's row is asked
//     for BEFORE that (a lane only reads and writes its own entries of a row, the LDS serves a wave in order; if the context stays, the
//     registers are kept and what was fetched is dropped): no branch, and the LDS latency lies under the update's instructions;
//   * ONE division, without a correction step: range / total = trunc ((range + 0.5) * rcp) with rcp = v_rcp_f64 and one Newton step
//     (relative error 2^-48.8 measured over every total a model can have, needed: below 2^-33); the total is the last entry's cumulative +
//     frequency (a v_readlane at a fixed lane);
//   * the second division is gone: "cum <= code / r < cum + freq" is "code - cum * r < freq * r" in wrapping 32-bit arithmetic
//     ((cum + freq) * r <= total * r <= range < 2^32: the products do not wrap, a code below cum * r wraps to something larger than
//     any freq * r), and the two products ARE the coder's new code and range (c_range_coder.h:118-119);
//   * the entry that is hit hands over its values through an exec window (v_cmpx, four v_readfirstlane, exec back on: gz_hit_window) -
//     no ballot / s_ff1 / v_readlane chain; its left neighbour's entry (for the bubble step) comes along, fetched beforehand with a
//     wave_shr:1 DPP move;
//   * the model's update is selects on per-lane predicates that need no lane number: "behind the hit" = the subtraction's borrow, "the
//     hit's left neighbour" = code - (cum + freq) * r equals the new code;
//   * the rare cases - a hit beyond list position 63, the halving of the frequencies, a window that runs low - leave the fast loop BEFORE
//     anything is changed and take d_lit_slow, the general routine.
// Measured: see DESIGN.md (decode).
#ifdef GZ_DEC_PROFILE
// (probe builds only, tools/dec_bench.py with GZ_LIB: s_memtime stamps along one symbol of the fast loop, each waited for - ~40 clocks a stamp)
#define GZ_NSTAMP 8
struct GzDecProf { uint64_t t[GZ_NSTAMP], sum[GZ_NSTAMP], n, fast_calls, slow, refills; };
#define GZ_STAMP(k) asm volatile ("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s" (st.P.t[k]) : : "memory")
#define GZ_PROF(x) x
#else
#define GZ_STAMP(k)
#define GZ_PROF(x)
#endif

template <int NP>
struct GzLitState {
    uint32_t x[NP], c[NP];            // the row of context `ctx`: entries lane + 64 j (freq | sym << 16, cumulative); 0 beyond the alphabet
    uint32_t ctx;
#ifdef GZ_DEC_PROFILE
    GzDecProf P;
#endif
};

template <int NP> __device__ static __forceinline__ uint32_t d_pick (const uint32_t (&v)[NP], uint32_t p)   // v[p], p uniform
{
    uint32_t r = 0;                                              // (masks, not selects: a chain of selects between array elements ends up as an indexed load from a copy of the array in scratch memory)
    for (int j = 0; j < NP; j++) r |= v[j] & (0u - (uint32_t)(p == (uint32_t)j));
    return r;
}

// floor (a / t) for a < 2^32, 0 < t < 2^16, as trunc (a * rcp) with a reciprocal that is too LARGE by a relative 2^-40 (+- the 2^-48.8 that
// v_rcp_f64 and one Newton step leave, measured over every t: tools/probes/lat_probe.hip): a / t is an integer k or at least 1 / t away
// from the next one, and the excess is below 2^32 / t * 2^-39.9 < 1 / t - the product never reaches k + 1 and never falls below k. The bias
// rides in the Newton step's constant (1 + 2^-40 instead of 1): no instruction of its own, and no correction step.
__device__ static __forceinline__ uint32_t d_div_exact (uint32_t a, uint32_t t)
{
    const double d = (double)t;
    double r = gz_rcp_f64 (d);
    r = __builtin_fma (__builtin_fma (-d, r, 1.0 + 0x1p-40), r, r);
    return (uint32_t)((double)a * r);
}

template <int NP, typename MP>
__device__ static __forceinline__ void d_lit_init (MP rows, uint32_t ms, bool o1, GzLitState<NP> &st, int lane)
{
    if (o1) for (uint32_t i = lane; i < ms * GZ_DEC_LIT_ROW (ms); i += 64) {
        const uint32_t k = i % GZ_DEC_LIT_ROW (ms), e = k / 2;
        rows[i] = e >= ms ? 0u : (k & 1) ? e /* cum: every frequency is 1 */ : (1u | (e << 16));
    }
    for (int j = 0; j < NP; j++) {
        const uint32_t e = (uint32_t)lane + 64 * j;
        st.x[j] = e < ms ? (1u | (e << 16)) : 0u; st.c[j] = e < ms ? e : 0u;
    }
    st.ctx = 0;
#ifdef GZ_DEC_PROFILE
    for (int k = 0; k < GZ_NSTAMP; k++) { st.P.t[k] = 0; st.P.sum[k] = 0; }
    st.P.n = st.P.fast_calls = st.P.slow = st.P.refills = 0;
#endif
}

template <int NP, typename MP>
__device__ static __forceinline__ void d_lit_switch (MP rows, uint32_t ms, GzLitState<NP> &st, uint32_t ctx, int lane)
{
    MP nw = rows + ctx * GZ_DEC_LIT_ROW (ms), old = rows + st.ctx * GZ_DEC_LIT_ROW (ms);
    uint32_t nx[NP], nc[NP];
    for (int j = 0; j < NP; j++) { const uint32_t e = (uint32_t)lane + 64 * j; nx[j] = nw[2 * e]; nc[j] = nw[2 * e + 1]; }
    for (int j = 0; j < NP; j++) { const uint32_t e = (uint32_t)lane + 64 * j; old[2 * e] = st.x[j]; old[2 * e + 1] = st.c[j]; }
    for (int j = 0; j < NP; j++) { st.x[j] = nx[j]; st.c[j] = nc[j]; }
    st.ctx = ctx;
}

// one literal in context ctx (uniform; < ms), the general way -> the symbol; 0 and no update when the code lies beyond the model (a
// malformed stream)
template <int NP, typename MP>
__device__ static inline uint32_t d_lit_slow (MP rows, uint32_t ms, GzLitState<NP> &st, uint32_t ctx, GzRcDec &rc, int lane)
{
    uint32_t e[NP];
    for (int j = 0; j < NP; j++) e[j] = (uint32_t)lane + 64 * j;
    if (ctx != st.ctx) d_lit_switch<NP, MP> (rows, ms, st, ctx, lane);
    if (rc.wvalid < 16) d_rc_refill (rc, lane);
    const uint32_t lp = (ms - 1) >> 6, ll = (ms - 1) & 63;
    const uint32_t tot = d_readlane (d_pick<NP> (st.c, lp) + (d_pick<NP> (st.x, lp) & 0xffff), (int)ll);
    const uint32_t r = rc.range >= tot ? d_div_exact (rc.range, tot) : 0;
    uint32_t d[NP], q[NP];
    for (int j = 0; j < NP; j++) { d[j] = rc.code - st.c[j] * r; q[j] = (st.x[j] & 0xffff) * r; }
    uint32_t plane = 0;
    uint64_t hit = __ballot (d[0] < q[0]);
    uint32_t hx = st.x[0], hc = st.c[0], hd = d[0], hq = q[0];
    for (int j = 1; j < NP; j++) if (!hit) {
        hit = __ballot (d[j] < q[j]);
        plane = j; hx = st.x[j]; hc = st.c[j]; hd = d[j]; hq = q[j];
    }
    if (!hit) return 0;
    const int l = __ffsll ((unsigned long long)hit) - 1;
    const uint32_t ex = d_readlane (hx, l);
    rc.code = d_readlane (hd, l);
    rc.range = d_readlane (hq, l);
    for (int k = 0; k < 3 && rc.range < (1u << 24); k++) {
        rc.code = (rc.code << 8) + d_dec_byte (rc, lane);
        rc.range <<= 8;
    }
    // ---- the model's update (c_simple_model.h:127-146), in the registers ----
    const uint32_t at = plane * 64 + l, ec = d_readlane (hc, l), e_new = ex + 16, f_new = e_new & 0xffff;
    if (tot + 16 <= 65519) {
        uint32_t left = 0xffffu;
        if (l > 0) left = d_readlane (hx, l - 1);
        else if (plane) left = d_readlane (d_pick<NP> (st.x, plane - 1), 63);
        const bool swap = at && f_new > (left & 0xffff);             // one bubble step to the left
        for (int j = 0; j < NP; j++) {
            st.c[j] += e[j] > at ? 16u : 0u;
            if (e[j] == at) { st.x[j] = swap ? left : e_new; st.c[j] = swap ? ec - (left & 0xffff) + f_new : ec; }
            if (swap && e[j] + 1 == at) st.x[j] = e_new;             // (keeps its cumulative)
        }
    }
    else {                                                       // halve every frequency, rebuild the cumulatives
        uint32_t run = 0;
        for (int j = 0; j < NP; j++) {
            if (e[j] == at) st.x[j] = e_new;
            uint32_t f = st.x[j] & 0xffff;
            f -= f >> 1;
            st.x[j] = (st.x[j] & 0xffff0000u) | f;
            const uint32_t inc = d_wave_incl_scan (f, lane);
            st.c[j] = run + inc - f;
            run += d_readlane (inc, 63);
        }
        if (at) {
            const uint32_t lp2 = (at - 1) >> 6, ll2 = (at - 1) & 63;
            const uint32_t mine = d_readlane (d_pick<NP> (st.x, plane), l), left = d_readlane (d_pick<NP> (st.x, lp2), (int)ll2);
            const uint32_t lc = d_readlane (d_pick<NP> (st.c, lp2), (int)ll2);
            if ((mine & 0xffff) > (left & 0xffff))
                for (int j = 0; j < NP; j++) {
                    if (e[j] + 1 == at) st.x[j] = mine;
                    if (e[j] == at) { st.x[j] = left; st.c[j] = lc + (mine & 0xffff); }
                }
        }
    }
    return ex >> 16;
}

// literals i .. of a stream without run lengths, the fast way, for as long as nothing rare comes up -> the index of the first literal NOT
// decoded (n: all done). `last`: the symbol before literal i. The output is buffered in obuf (symbol k of the current 64 in lane k % 64)
// and stored 64 at a time.
template <int NP, bool O1, typename MP>
__device__ static __forceinline__ uint32_t d_lit_fast (MP rows, uint32_t ms, GzLitState<NP> &st, GzRcDec &rc, uint32_t i, uint32_t n, uint32_t &last,
                                                      uint32_t &obuf, uint8_t *out, int lane)
{
    const uint32_t ll = (ms - 1) & 63, e0 = (uint32_t)lane, stride = GZ_DEC_LIT_ROW (ms), first = i;
    const bool top = ((ms - 1) >> 6) == 3;                       // (NP == 4: the last entry lies in plane 2 or 3)
    if (O1 && last != st.ctx) d_lit_switch<NP, MP> (rows, ms, st, last, lane);
    uint32_t code = rc.code, range = rc.range, wvalid = rc.wvalid;
    uint64_t W = rc.W;
    uint32_t cnt = n - i < 64 - (i & 63) ? n - i : 64 - (i & 63);          // to the end of the block of 64 / of the stream; the caller saw to wvalid >= 16
    GZ_PROF (st.P.fast_calls++);
    for (;;) {
        GZ_STAMP (0);
        const uint32_t f0 = st.x[0] & 0xffff;
        uint32_t tsum;
        if constexpr (NP == 1) tsum = st.c[0] + f0;
        else if constexpr (NP == 2) tsum = st.c[1] + (st.x[1] & 0xffff);
        else tsum = top ? st.c[3] + (st.x[3] & 0xffff) : st.c[2] + (st.x[2] & 0xffff);
        const uint32_t tot = d_readlane (tsum, (int)ll);
        const uint32_t r = d_div_exact (range, tot);
        const uint32_t p = st.c[0] * r, q = f0 * r, d = code - p, old_code = code;
        const uint32_t xl = gz_wave_shr1 (st.x[0], 0);           // lane l: the entry to the left of lane l's (lane 0: none, reads 0)
        uint32_t ex, nd, nq, exl;
        GZ_STAMP (1);
        gz_hit_window (d, q, st.x[0], xl, ex, nd, nq, exl);
        GZ_STAMP (2);
        if ((tot > 65519 - 16 ? 0xffffffffu : nd) >= nq) break;  // no hit among the first 64 entries (lane 0's values come back) / halving due
        // ---- the coder: code -= cum * r, range = freq * r, normalise (c_range_coder.h:118-126) ----
        const uint32_t sh = (uint32_t)__builtin_clz (nq) & 0x18;
        code = (uint32_t)(((((uint64_t)nd << 32) | (W >> 32)) << sh) >> 32);
        range = nq << sh;
        W <<= sh; wvalid -= sh;
        GZ_STAMP (3);
        // ---- the next context's row is asked for ----
        const uint32_t sym = ex >> 16;
        uint32_t nx[NP], nc[NP];
        if (O1) {
            MP nw = rows + sym * stride;
            for (int j = 0; j < NP; j++) { const uint32_t e = e0 + 64 * j; nx[j] = nw[2 * e]; nc[j] = nw[2 * e + 1]; }
        }
        GZ_STAMP (4);
        // ---- the model's update (c_simple_model.h:127-146): freq += 16, one bubble step to the left if that beats the neighbour ----
        // (f_new > f_left written as f_new - 1 > f_left - 1: no left neighbour reads 0, 0 - 1 is 0xffff - no swap, and the entries beyond
        // the alphabet, which pass for "left neighbours" when the hit is entry 0, are rewritten with that 0)
        const uint32_t e_new = ex + 16;
        const bool swap = ((ex + 15) & 0xffff) > ((exl - 1) & 0xffff);
        const uint32_t x_hit = swap ? exl : e_new, x_left = swap ? e_new : exl;
        uint32_t delta = swap ? (e_new & 0xffff) - (exl & 0xffff) : 0u;
        gz_opaque (delta);                                       // (or the compiler ANDs `swap` into the lanes' hit mask on the SALU: a wait for the VALU)
        const bool hit = d < q, behind = p > old_code, left = d - q == nd;
        uint32_t x0 = hit ? x_hit : st.x[0];
        x0 = left ? x_left : x0;
        uint32_t inc = hit ? delta : 0u;
        inc = behind ? 16u : inc;
        st.x[0] = x0; st.c[0] += inc;
        for (int j = 1; j < NP; j++) st.c[j] += 16;
        GZ_STAMP (5);
        // ---- the old row goes back, the new one takes its place unless the context stays ----
        if (O1) {
            MP old = rows + st.ctx * stride;
            for (int j = 0; j < NP; j++) { const uint32_t e = e0 + 64 * j; old[2 * e] = st.x[j]; old[2 * e + 1] = st.c[j]; }
            const bool stay = sym == st.ctx;
            for (int j = 0; j < NP; j++) { st.x[j] = stay ? st.x[j] : nx[j]; st.c[j] = stay ? st.c[j] : nc[j]; }
            st.ctx = sym;
        }
        obuf = (uint32_t)lane == (i & 63) ? sym : obuf;
        last = sym;
        i++;
#ifdef GZ_DEC_PROFILE
        GZ_STAMP (6);
        for (int k = 1; k <= 6; k++) st.P.sum[k] += st.P.t[k] - st.P.t[k - 1];
        if (st.P.t[7]) st.P.sum[0] += st.P.t[0] - st.P.t[7];
        st.P.t[7] = st.P.t[6]; st.P.n++;
#endif
        const uint32_t go = --cnt ? wvalid : 0u;
        if (go < 16) break;
    }
    GZ_PROF (st.P.t[7] = 0);
    if (i > first && !(i & 63)) gz_stg_u8 (out + (i - 64) + lane, obuf);
    rc.code = code; rc.range = range; rc.wvalid = wvalid; rc.W = W;
    return i;
}

template <int NP, typename MP>
__device__ static __forceinline__ void d_arith_decode_leaf (GzdDecLeaf &L, MP models, uint32_t ms, uint32_t n, int lane)
{
    const bool o1 = gz_first_lane (L.o1), rle = gz_first_lane (L.rle);
    GzLitState<NP> st;
    d_lit_init<NP, MP> (models, ms, o1, st, lane);
    MP runm = models + (o1 ? ms * GZ_DEC_LIT_ROW (ms) : 0);
    if (rle) for (uint32_t i = lane; i < 258 * GZ_DEC_RUN_ROW; i += 64) {
        const uint32_t k = i % GZ_DEC_RUN_ROW;
        runm[i] = k == 0 ? 4u : k == 1 ? 0u : (k & 1) ? (k - 2) / 2 : (1u | (((k - 2) / 2) << 16));
    }
    __syncthreads ();

    GzRcDec rc;
    d_rc_start (rc, L.body + 1, gz_first_lane (L.body_len - 1), lane);
    uint8_t *out = L.dst;
    uint32_t last = 0, obuf = 0;                                 // obuf: symbol i of the current 64 in lane i % 64, stored 64 at a time
    for (uint32_t i = 0; i < n; i++) {
        if (!rle) {                                              // the fast loop takes what it can, the general routine the literal it stopped at
            for (;;) {
                if (rc.wvalid < 16) { d_rc_refill (rc, lane); GZ_PROF (st.P.refills++); }
                i = o1 ? d_lit_fast<NP, true, MP> (models, ms, st, rc, i, n, last, obuf, out, lane)
                       : d_lit_fast<NP, false, MP> (models, ms, st, rc, i, n, last, obuf, out, lane);
                if (i >= n || rc.wvalid >= 16) break;
            }
            if (i >= n) break;
        }
        uint32_t s = d_lit_slow<NP, MP> (models, ms, st, o1 ? last : 0, rc, lane);
        GZ_PROF (st.P.slow++);
        if (s >= ms) s = 0;
        obuf = (uint32_t)lane == (i & 63) ? s : obuf;
        if ((i & 63) == 63) gz_stg_u8 (out + (i & ~63u) + lane, obuf);
        last = s;
        if (!rle) continue;
        uint32_t run = 0, d, ctx = s;                            // arith_dynamic.c:476-487
        do {
            d = d_model_decode<MP> (runm + ctx * GZ_DEC_RUN_ROW, 4, rc, lane);
            ctx = (ctx == s) ? 256 : ctx + (ctx < 257);
            run += d;
        } while (d == 3 && run < n);
        // (a run: flush what is buffered, then the wave writes the run 64 bytes at a time)
        const uint32_t r = run < n - 1 - i ? run : n - 1 - i;
        if (r) {
            if ((i & 63) != 63 && (uint32_t)lane <= (i & 63)) gz_stg_u8 (out + (i & ~63u) + lane, obuf);
            for (uint32_t k = lane; k < r; k += 64) gz_stg_u8 (out + i + 1 + k, s);
            const uint32_t first = i + 1;
            i += r;
            // (the buffer mirrors the 64-block the run ends in: its positions first .. i hold the run's symbol)
            const uint32_t pos = (i & ~63u) + lane;
            obuf = pos >= first && pos <= i ? s : obuf;
            if ((i & 63) == 63) gz_stg_u8 (out + (i & ~63u) + lane, obuf);
        }
    }
    if ((n & 63) && (uint32_t)lane < (n & 63)) gz_stg_u8 (out + (n & ~63u) + lane, obuf);
    if (!lane) L.status = GZ_ST_OK;
#ifdef GZ_DEC_PROFILE
    if (!lane && n > 100000 && st.P.n) printf ("[dec profile NP %d ms %u o1 %d] %u symbols: %llu fast (in %llu calls of the fast loop), %llu slow, %llu refills; clocks per fast symbol: loop %.1f | r, products %.1f | window %.1f | rare check %.1f | normalise, next row asked for %.1f | update %.1f | rows, out %.1f\n",
        NP, ms, (int)o1, n, (unsigned long long)st.P.n, (unsigned long long)st.P.fast_calls, (unsigned long long)st.P.slow, (unsigned long long)st.P.refills,
        (double)st.P.sum[0] / st.P.n, (double)st.P.sum[1] / st.P.n, (double)st.P.sum[2] / st.P.n, (double)st.P.sum[3] / st.P.n, (double)st.P.sum[4] / st.P.n, (double)st.P.sum[5] / st.P.n, (double)st.P.sum[6] / st.P.n);
#endif
}

template <typename MP>
__device__ static __forceinline__ void d_arith_decode_planes (GzdDecLeaf &L, MP models, uint32_t ms, uint32_t n, int lane)
{
    if (ms <= 64) d_arith_decode_leaf<1, MP> (L, models, ms, n, lane);
    else if (ms <= 128) d_arith_decode_leaf<2, MP> (L, models, ms, n, lane);
    else d_arith_decode_leaf<4, MP> (L, models, ms, n, lane);
}

__global__ void __launch_bounds__(64) k_arith_decode (GzdDecLeaf *leaves, uint32_t lds_words_lo, uint32_t lds_words_hi, int use_global)
{
    GzdDecLeaf &L = leaves[GZ_DEC_LEAF_OF_BLOCK];
    if (!L.active || L.engine != GZ_ENG_ARITH || L.cat || !L.coded_n) return;
    if (!L.body_len) return;
    // (what comes through a generic pointer is a divergent value to the compiler - and with it every branch and every loop that depends on
    // it, executed under exec masks with its variables in VGPRs: say that these are the same in every lane)
    const uint32_t n = gz_first_lane (L.coded_n);
    const uint32_t ms = gz_first_lane (L.body[0] ? L.body[0] : 256);
    const uint32_t words = (L.o1 ? ms * GZ_DEC_LIT_ROW (ms) : 1) + (L.rle ? 258 * GZ_DEC_RUN_ROW : 0);      // (an order-0 model lives in registers: class 1)
    if (words <= lds_words_lo || words > lds_words_hi) return;
    if (use_global) d_arith_decode_planes<uint32_t *> (L, L.models, ms, n, (int)threadIdx.x);
    else d_arith_decode_planes<GzLdsU32P> (L, (GzLdsU32P)gz_lds, ms, n, (int)threadIdx.x);
}

// one 256-thread workgroup per stream
__global__ void __launch_bounds__(256) k_dec_finish (GzdDecStream *streams, GzdDecLeaf *leaves)
{
    GzdDecStream &S = streams[blockIdx.x];
    const int tid = threadIdx.x;
    if (S.status != GZ_ST_PENDING) return;
    if (S.codec == 1) {
        for (uint32_t i = tid; i < S.out_len; i += 256) S.out[i] = S.in[i];
        __syncthreads ();       // status is also this kernel's entry test: change it only once everybody is past it
        if (!tid) S.status = GZ_ST_OK;
        return;
    }
    bool ok = true;
    for (uint32_t k = 0; k < S.n_leaves; k++) {
        GzdDecLeaf &L = leaves[S.first_leaf + k];
        if (!L.active) { ok = false; continue; }
        if (L.coded_n) {
            if (L.cat) {
                if (L.body_len < L.coded_n) ok = false;
                else for (uint32_t i = tid; i < L.coded_n; i += 256) L.dst[i] = L.body[i];
            }
            else if (L.status != GZ_ST_OK) ok = false;
        }
        __syncthreads ();
        if (L.packed_on && ok) {                               // pack.c:214-351
            const uint32_t per = L.per, n = L.n;
            if (per == 1) { if (L.coded_n != n) ok = false; else for (uint32_t i = tid; i < n; i += 256) L.final_dst[i] = L.dst[i]; }
            else if (per == 0) { for (uint32_t i = tid; i < n; i += 256) L.final_dst[i] = L.map[0]; }
            else {
                if ((n + per - 1) / per > L.coded_n) ok = false;
                else {
                    const uint32_t width = 8 / per, mask = (1u << width) - 1;
                    for (uint32_t i = tid; i < n; i += 256) L.final_dst[i] = L.map[(L.dst[i / per] >> ((i % per) * width)) & mask];
                }
            }
        }
        else if (!L.packed_on && ok && L.coded_n != L.n) ok = false;
        __syncthreads ();
    }
    if (ok && S.striped) {
        uint32_t len[4], off[4];
        gz_plane_geometry (S.out_len, len, off);
        for (uint32_t i = tid; i < S.out_len; i += 256) S.out[i] = S.tmp_planes[off[i & 3] + (i >> 2)];
    }
    __syncthreads ();
    if (!tid) S.status = ok ? GZ_ST_OK : GZ_ST_CORRUPT;
}

// ---- the sections of many VBlocks: found and checked on the device (zfile.c:212-218: magic, lengths, z_digest) -------------------------
struct GzdVbSec { uint64_t at; uint32_t clen, ulen, adler, codec, ok; uint32_t pad; };      // at: offset of the payload in the VBlock's z_data
struct GzdVbWalk { const uint8_t *z; uint64_t z_len, out_cap; uint32_t n_sections, status; }; // status: GZ_ST_OK / GZ_ST_CORRUPT

// one thread per VBlock: its section headers one after the other (40 bytes each, behind the 84 of the VBlock header)
__global__ void k_vb_walk (GzdVbWalk *vbs, GzdVbSec *secs, uint32_t n_vbs, uint32_t max_sections)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_vbs) return;
    GzdVbWalk &V = vbs[v];
    GzdVbSec *S = secs + (size_t)v * max_sections;
    V.n_sections = 0; V.status = GZ_ST_CORRUPT;
    const uint8_t *z = V.z;
    if (V.z_len < 84 || gz_rd_be32 (z) != 0x27052012u || z[24] != GZ_SEC_VB_HEADER || gz_rd_be32 (z + 40) != V.z_len) return;
    uint64_t at = 84, o = 0;
    uint32_t n = 0;
    while (at < V.z_len) {
        if (at + 40 > V.z_len || gz_rd_be32 (z + at) != 0x27052012u) return;
        const uint32_t clen = gz_rd_be32 (z + at + 12), ulen = gz_rd_be32 (z + at + 16);
        if (at + 40 + clen > V.z_len || o + ulen > V.out_cap || n >= max_sections) return;
        uint32_t codec = z[at + 25];
        if (codec == GZ_CODEC_DOMQ || codec == GZ_CODEC_XCGT) codec = z[at + 26];                 // USE_SUBCODEC (compressor.c:60-61)
        S[n].at = at + 40; S[n].clen = clen; S[n].ulen = ulen; S[n].adler = gz_rd_be32 (z + at + 4); S[n].codec = codec; S[n].ok = 0;
        n++; o += ulen; at += 40 + clen;
    }
    V.n_sections = n; V.status = GZ_ST_OK;
}

// grid (max_sections, n_vbs), 256 threads: a section's adler32 against its header's
__global__ void __launch_bounds__(256) k_vb_digests (const GzdVbWalk *vbs, GzdVbSec *secs, uint32_t max_sections)
{
    const GzdVbWalk &V = vbs[blockIdx.y];
    if (V.status != GZ_ST_OK || blockIdx.x >= V.n_sections) return;
    GzdVbSec &S = secs[(size_t)blockIdx.y * max_sections + blockIdx.x];
    const uint32_t a = gz_adler32_wg (V.z + S.at, S.clen, threadIdx.x);
    if (!threadIdx.x) S.ok = a == S.adler;
}
