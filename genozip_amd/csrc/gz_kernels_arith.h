// gz_kernels_arith.h -- the adaptive arithmetic coder (arith_dynamic.c:92-197, c_simple_model.h, c_range_coder.h),
// re-thought for a GPU.
//
// The reference walks a stream once, per symbol: search the context's frequency-sorted list for the symbol
// (accumulating the cumulative frequency), divide the range by the model total, update low/range, renormalise,
// bump the frequency, maybe halve all, maybe swap with the left neighbour. One lane doing that costs ~2000 cycles per
// symbol. But the triple (cum, freq, tot) fed to the range coder depends only on the *model history*, never on the
// coder state, and in order-1 mode the 256 models never interact. So:
//
//   k_arith_model  one WAVE per (leaf, context). The model lives in registers, entry e in lane e%64: finding a symbol
//                  is one ballot, its cumulative frequency a register kept up to date incrementally (every entry after
//                  the bumped one gains 16), the swap two writelanes. The wave scans the input for the positions that
//                  belong to its context and writes the 8-byte triple of every such position. All contexts of all
//                  leaves run concurrently.
//   k_arith_chain  one wave per leaf replays the triples through the range coder. Only range -> range/tot*freq ->
//                  renormalise is truly serial; it runs on wave-uniform values (the scalar unit), with the division
//                  replaced by a multiply by a per-divisor magic number looked up from a table built once on the host.
//                  64 triples are fetched per iteration with one coalesced load and handed to the chain by readlane.
//
// Leaves using the run-length variant (stripe plane 0 candidate of ARTW/ARTw) keep the serial kernel in
// gz_kernels_enc.h for now.
#pragma once
#include "gz_device.h"
#include "gz_devutil.h"

#define GZ_MODEL_LIMIT 65519u          // MAX_FREQ (c_simple_model.h:63)
#define GZ_MODEL_STEP  16u

struct GzDivMagic { uint32_t magic, shift; };   // q = ((((n - t) >> 1) + t) >> shift, t = mulhi(magic, n); divisor 1: shift = 0xff

__device__ static inline uint32_t d_readlane (uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane ((int)v, lane); }
// (clang 22 / ROCm 7.2 has no __builtin_amdgcn_writelane; compare + select costs one more VALU op than v_writelane_b32)
__device__ static inline uint32_t d_writelane (uint32_t val, int lane, uint32_t old) { return (int)(threadIdx.x & 63) == lane ? val : old; }

// J = registers per lane holding the model (entries e = j*64 + lane), J*64 >= max_sym
template <int J>
__device__ static void d_arith_model_wave (const uint8_t *in, uint32_t n, uint32_t ms, bool o1, uint32_t ctx, uint4 *recs, const GzDivMagic *magic_tab)
{
    const int lane = threadIdx.x & 63;
    uint32_t sym[J], freq[J], cum[J];
    #pragma unroll
    for (int j = 0; j < J; j++) {
        uint32_t e = j * 64 + lane;
        sym[j] = e; freq[j] = e < ms ? 1 : 0; cum[j] = e < ms ? e : ms;
    }
    uint32_t tot = ms;

    // The wave scans the whole stream for "its" positions, 4 x 64 positions per iteration; the bytes of the next
    // iteration are requested before the current ones are used, so the scan never waits for memory.
    uint32_t nx_s[4], nx_p[4];
    #pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t pos = k * 64 + lane;
        nx_s[k] = pos < n ? in[pos] : 0;
        nx_p[k] = (o1 && pos && pos < n) ? in[pos - 1] : 0;
    }
    for (uint32_t gbase = 0; gbase < n; gbase += 256) {
        uint32_t cs[4], cp[4];
        #pragma unroll
        for (int k = 0; k < 4; k++) { cs[k] = nx_s[k]; cp[k] = nx_p[k]; }
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t pos = gbase + 256 + k * 64 + lane;
            nx_s[k] = pos < n ? in[pos] : 0;
            nx_p[k] = (o1 && pos < n) ? in[pos - 1] : 0;
        }
      #pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t base = gbase + k * 64;
        if (base >= n) break;
        const uint32_t pos = base + lane;
        const uint32_t s_v = cs[k];
        const bool mine = pos < n && (!o1 || cp[k] == ctx);
        uint64_t todo = __ballot (mine);
        uint32_t out_lo = 0, out_hi = 0;
        while (todo) {
            const int b = __ffsll ((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const uint32_t s = d_readlane (s_v, b);
            // ---- find the symbol: global position p = jj*64 + pl
            int jj = 0, pl = 0;
            #pragma unroll
            for (int j = 0; j < J; j++) {
                uint64_t hit = __ballot (sym[j] == s && (uint32_t)(j * 64 + lane) < ms);
                if (hit) { jj = j; pl = __ffsll ((unsigned long long)hit) - 1; }
            }
            uint32_t f = 0, cu = 0;
            #pragma unroll
            for (int j = 0; j < J; j++) if (j == jj) { f = d_readlane (freq[j], pl); cu = d_readlane (cum[j], pl); }
            // ---- hand (cum, freq, tot) to the lane that owns this position
            out_lo = d_writelane (cu | (f << 16), b, out_lo);      // cum and freq both fit 16 bits
            out_hi = d_writelane (tot, b, out_hi);
            // ---- bump (c_simple_model.h:133-134): the entry gains 16, so does the cumulative of everything after it
            const uint32_t p = jj * 64 + pl;
            uint32_t f16 = f + GZ_MODEL_STEP;
            tot += GZ_MODEL_STEP;
            #pragma unroll
            for (int j = 0; j < J; j++) {
                if (j == jj) freq[j] = d_writelane (f16, pl, freq[j]);
                cum[j] += ((uint32_t)(j * 64 + lane) > p) ? GZ_MODEL_STEP : 0u;
            }
            // ---- halve everything when the total passes the limit (c_simple_model.h:106-115,136-137)
            if (tot > GZ_MODEL_LIMIT) {
                #pragma unroll
                for (int j = 0; j < J; j++) freq[j] -= freq[j] >> 1;
                uint32_t run = 0;
                #pragma unroll
                for (int j = 0; j < J; j++)
                    for (int l = 0; l < 64; l++) {
                        if ((uint32_t)(j * 64 + l) >= ms) break;
                        cum[j] = d_writelane (run, l, cum[j]);
                        run += d_readlane (freq[j], l);
                    }
                tot = run;
                #pragma unroll
                for (int j = 0; j < J; j++) {
                    if (j == jj) f16 = d_readlane (freq[j], pl);
                    cum[j] = ((uint32_t)(j * 64 + lane) < ms) ? cum[j] : run;
                }
            }
            // ---- keep approximately sorted: one bubble step to the left (c_simple_model.h:139-145)
            if (p > 0) {
                const int jq = (int)((p - 1) >> 6), lq = (int)((p - 1) & 63);
                uint32_t fl = 0, sl = 0, cl = 0;
                #pragma unroll
                for (int j = 0; j < J; j++) if (j == jq) { fl = d_readlane (freq[j], lq); sl = d_readlane (sym[j], lq); cl = d_readlane (cum[j], lq); }
                if (f16 > fl) {
                    #pragma unroll
                    for (int j = 0; j < J; j++) {
                        if (j == jq) { sym[j] = d_writelane (s, lq, sym[j]); freq[j] = d_writelane (f16, lq, freq[j]); }
                        if (j == jj) { sym[j] = d_writelane (sl, pl, sym[j]); freq[j] = d_writelane (fl, pl, freq[j]); cum[j] = d_writelane (cl + f16, pl, cum[j]); }
                    }
                }
            }
        }
        if (mine) { GzDivMagic mg = magic_tab[out_hi]; recs[pos] = make_uint4 (out_lo & 0xffff, out_lo >> 16, mg.magic, mg.shift); }
      }
    }
}

// Compact variant for leaves with at most 64 distinct byte values (every quality / token stream): only the symbols
// that occur are kept, one per lane, in list order. The max_sym - nsym entries of symbols that never occur all have
// frequency 1 for ever (halving leaves 1 alone) and never start a swap, so they are interchangeable: each present
// symbol just remembers how many of them sit directly in front of it (`gap`). Coding a symbol whose gap is > 0 swaps it
// with such an entry (its frequency, >= 17, always beats 1): gap--, and the next present symbol's gap++. With gap 0
// the left neighbour is the previous lane and the ordinary "swap if now larger" applies. cum includes the gaps.
// The common case (no swap, no halving) is ~25 instructions with a single vector->scalar decision.
__device__ static void d_arith_model_wave_compact (const uint8_t *in, uint32_t n, uint32_t ms, bool o1, uint32_t ctx,
                                                   uint4 *recs, const GzDivMagic *magic_tab, const uint8_t *symlist, uint32_t nsym)
{
    const int lane = threadIdx.x & 63;
    const bool live = (uint32_t)lane < nsym;
    uint32_t sym = live ? symlist[lane] : 0xffffffffu;
    uint32_t prev_sym = (live && lane) ? symlist[lane - 1] : 0xffffffffu;
    uint32_t gap = live ? (lane ? sym - prev_sym - 1 : sym) : 0;
    uint32_t freq = live ? 1 : 0;
    uint32_t cum = live ? sym : ms;                       // lane entries + absent entries before it == its byte value
    uint32_t tot = ms;
    const uint32_t n_absent = ms - nsym;

    uint32_t nx_s[4], nx_p[4];
    #pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t pos = k * 64 + lane;
        nx_s[k] = pos < n ? in[pos] : 0;
        nx_p[k] = (o1 && pos && pos < n) ? in[pos - 1] : 0;
    }
    for (uint32_t gbase = 0; gbase < n; gbase += 256) {
        uint32_t cs[4], cp[4];
        #pragma unroll
        for (int k = 0; k < 4; k++) { cs[k] = nx_s[k]; cp[k] = nx_p[k]; }
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t pos = gbase + 256 + k * 64 + lane;
            nx_s[k] = pos < n ? in[pos] : 0;
            nx_p[k] = (o1 && pos < n) ? in[pos - 1] : 0;
        }
      #pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t base = gbase + k * 64;
        if (base >= n) break;
        const uint32_t pos = base + lane;
        const uint32_t s_v = cs[k];
        const bool mine = pos < n && (!o1 || cp[k] == ctx);
        uint64_t todo = __ballot (mine);
        uint32_t out_cum = 0, out_freq = 0, out_tot = 0;
        while (todo) {
            const int b = __ffsll ((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const uint32_t s = d_readlane (s_v, b);
            const bool me = sym == s;                                    // exactly one lane
            const int r = __ffsll ((unsigned long long)__ballot (me)) - 1;
            const uint32_t f = d_readlane (freq, r), cu = d_readlane (cum, r);
            const bool owner = lane == b;
            out_cum = owner ? cu : out_cum; out_freq = owner ? f : out_freq; out_tot = owner ? tot : out_tot;
            // bump
            freq += me ? GZ_MODEL_STEP : 0u;
            cum  += lane > r ? GZ_MODEL_STEP : 0u;
            tot  += GZ_MODEL_STEP;
            if (tot > GZ_MODEL_LIMIT) {                                  // rare: halve, rebuild tot and cum
                freq -= freq >> 1;
                uint32_t run = 0;
                for (uint32_t l = 0; l < nsym; l++) {
                    run += d_readlane (gap, (int)l);
                    cum = d_writelane (run, (int)l, cum);
                    run += d_readlane (freq, (int)l);
                }
                uint32_t fsum = 0;
                for (uint32_t l = 0; l < nsym; l++) fsum += d_readlane (freq, (int)l);
                tot = fsum + n_absent;                                   // every absent entry still weighs 1
            }
            // one bubble step to the left (c_simple_model.h:139-145)
            const uint32_t f_now = d_readlane (freq, r);
            const bool over_absent = me && gap > 0;
            const bool over_left = (lane == r - 1) && freq < f_now;
            const uint64_t chg = __ballot (over_absent || over_left);
            if (chg) {
                const uint32_t g = d_readlane (gap, r);
                if (g > 0) {
                    gap += me ? 0xffffffffu : (lane == r + 1 ? 1u : 0u);  // the absent entry hops over: mine - 1, next + 1
                    cum += me ? 0xffffffffu : 0u;
                }
                else {
                    const int q = r - 1;
                    const uint32_t fl = d_readlane (freq, q), sl = d_readlane (sym, q), cl = d_readlane (cum, q), gl = d_readlane (gap, q);
                    const bool at_q = lane == q;
                    sym  = at_q ? s : (me ? sl : sym);
                    freq = at_q ? f_now : (me ? fl : freq);
                    gap  = at_q ? gl : (me ? 0u : gap);
                    cum  = me ? cl + f_now : cum;                        // lane q keeps its cumulative
                }
            }
        }
        if (mine) { GzDivMagic mg = magic_tab[out_tot]; recs[pos] = make_uint4 (out_cum, out_freq, mg.magic, mg.shift); }
      }
    }
}

// grid (n_leaves, 256): block y serves context y of leaf x
__global__ void __launch_bounds__(64) k_arith_model (GzdLeaf *leaves, const GzDivMagic *magic_tab)
{
    GzdLeaf &L = leaves[blockIdx.x];
    if (!L.active || L.engine != GZ_ENG_ARITH || L.rle || !L.coded_n) return;
    const uint32_t ctx = blockIdx.y, ms = L.max_sym;
    const bool o1 = L.o1;
    if (o1 ? (ctx >= ms || (ctx && L.symrank[ctx] == 0xffff)) : ctx != 0) return;   // a byte that never occurs is never a context
    uint4 *tr = (uint4 *)L.triples;
    if (L.nsym <= 64)   d_arith_model_wave_compact (L.coded, L.coded_n, ms, o1, ctx, tr, magic_tab, L.symlist, L.nsym);
    else if (ms <= 64)  d_arith_model_wave<1> (L.coded, L.coded_n, ms, o1, ctx, tr, magic_tab);
    else if (ms <= 128) d_arith_model_wave<2> (L.coded, L.coded_n, ms, o1, ctx, tr, magic_tab);
    else                d_arith_model_wave<4> (L.coded, L.coded_n, ms, o1, ctx, tr, magic_tab);
}

// ---- range coder chain ------------------------------------------------------------------------------------------
// Measured on MI355X (tools/ubench_chain.hip): ONE wave issues an instruction every ~3.2 ns (scalar) / 2.5 ns
// (vector) whether or not it depends on the previous one, a vector->scalar hand-over (v_readlane feeding s_*) costs
// ~12 ns, and nothing but the instruction count of the wave matters. So the chain is kept entirely on the scalar
// unit and written for few instructions per symbol:
//  * records arrive through scalar (SMEM) loads, 8 at a time, the next 8 in flight while 8 are coded. SMEM returns
//    out of order, so the only wait is "all": the lines are therefore pulled into L2 well ahead by a cheap vector
//    load whose result is never used, which turns the scalar loads into L2 hits;
//  * r = range / tot is a multiply-high and three shifts/adds with the per-record magic number;
//  * carry is the high half of a 64-bit low, so (carry:low) >> 24 is exactly "top byte | carry << 8";
//  * the byte-output logic of the reference (RC_ShiftLow, c_range_coder.h:70-88: hold back a byte while later carries
//    can still reach it, count pending 0xFF bytes) is NOT run here. That logic is a lazy big-number addition: the
//    stream is [0, T1, T2, ...] (Tj = top byte of low at the j-th shift) plus, for every shift that saw the carry
//    flag set, +1 at the byte before it. The chain only records the 16-bit event (Tj | carry_j << 8) per shift, four
//    events per 64-bit store, and k_arith_carry resolves all carries of all leaves in parallel afterwards.
struct GzRcU { uint64_t lowc, acc; uint32_t range, nev; uint64_t *ev; int lane; };

__device__ static inline void d_rcu_shift (GzRcU &rc)
{
    rc.acc = (rc.acc >> 16) | ((rc.lowc >> 24) << 48);               // event = (carry:low) >> 24, 9 significant bits
    rc.nev++;
    if (!(rc.nev & 3)) { if (!rc.lane) rc.ev[(rc.nev >> 2) - 1] = rc.acc; }
    rc.lowc = (uint64_t)((uint32_t)rc.lowc << 8);
}

// One symbol: r = range / tot by multiplication (tot >= 2), low += cum * r, range = r * freq, renormalise.
__device__ static inline void d_rcu_step (GzRcU &rc, uint32_t cum, uint32_t freq, uint32_t mg, uint32_t sh)
{
    const uint32_t t = __umulhi (mg, rc.range);
    const uint32_t r = (((rc.range - t) >> 1) + t) >> sh;            // c_range_coder.h:100
    rc.lowc += (uint64_t)(cum * r);
    rc.range = r * freq;
    while (rc.range < (1u << 24)) { rc.range <<= 8; d_rcu_shift (rc); }
}

typedef uint32_t gz_u32x4 __attribute__((vector_size (16)));
typedef const volatile __attribute__((address_space(4))) gz_u32x4 *GzConstRecP;   // volatile: keeps the prefetch a prefetch

#define GZ_CHAIN_BLOCK 8
#define GZ_CHAIN_TOUCH_AHEAD (16 * 1024)   // bytes: how far ahead of the scalar loads the vector unit pulls lines into L2

// one wave per leaf
__global__ void __launch_bounds__(64) k_arith_chain (GzdLeaf *leaves)
{
    GzdLeaf &L = leaves[blockIdx.x];
    if (!L.active || L.engine != GZ_ENG_ARITH || L.rle) return;
    const int lane = threadIdx.x;
    const uint32_t n = L.coded_n;
    GzConstRecP rec = (GzConstRecP)(uintptr_t)L.triples;       // padded: reads up to 64 KB past n stay inside the area
    const uint32_t *touch = (const uint32_t *)L.triples;
    uint32_t sink = 0;

    GzRcU rc;
    rc.lowc = 0; rc.acc = 0; rc.range = 0xffffffffu; rc.nev = 0; rc.ev = (uint64_t *)L.events; rc.lane = lane;

    if (n && L.max_sym == 1) {
        // a stream of zero bytes: the model total starts at 1, which the multiply-shift division cannot express
        for (uint32_t i = 0; i < n; i++) {
            const gz_u32x4 c = rec[i];
            const uint32_t t = __umulhi (c[2], rc.range);
            const uint32_t r = c[3] == 0xff ? rc.range : (((rc.range - t) >> 1) + t) >> c[3];
            rc.lowc += (uint64_t)(c[0] * r);
            rc.range = r * c[1];
            while (rc.range < (1u << 24)) { rc.range <<= 8; d_rcu_shift (rc); }
        }
    }
    else {
        const uint32_t nb = n & ~(uint32_t)(GZ_CHAIN_BLOCK - 1);
        if (nb) {
            for (uint32_t b = 0; b < GZ_CHAIN_TOUCH_AHEAD; b += 4096) sink += touch[(b >> 2) + lane * 16];   // 64 lanes x 64 B = 4 KB
            gz_u32x4 c0 = rec[0], c1 = rec[1], c2 = rec[2], c3 = rec[3], c4 = rec[4], c5 = rec[5], c6 = rec[6], c7 = rec[7];
            for (uint32_t i = 0; i < nb; i += GZ_CHAIN_BLOCK) {
                const uint32_t nx = i + GZ_CHAIN_BLOCK < nb ? i + GZ_CHAIN_BLOCK : i;   // the last block re-reads itself
                const gz_u32x4 p0 = rec[nx], p1 = rec[nx + 1], p2 = rec[nx + 2], p3 = rec[nx + 3],
                               p4 = rec[nx + 4], p5 = rec[nx + 5], p6 = rec[nx + 6], p7 = rec[nx + 7];
                if (!(i & 255)) sink += touch[((i * 16 + GZ_CHAIN_TOUCH_AHEAD) >> 2) + lane * 16];  // every 256 records = 4 KB
                d_rcu_step (rc, c0[0], c0[1], c0[2], c0[3]);
                d_rcu_step (rc, c1[0], c1[1], c1[2], c1[3]);
                d_rcu_step (rc, c2[0], c2[1], c2[2], c2[3]);
                d_rcu_step (rc, c3[0], c3[1], c3[2], c3[3]);
                d_rcu_step (rc, c4[0], c4[1], c4[2], c4[3]);
                d_rcu_step (rc, c5[0], c5[1], c5[2], c5[3]);
                d_rcu_step (rc, c6[0], c6[1], c6[2], c6[3]);
                d_rcu_step (rc, c7[0], c7[1], c7[2], c7[3]);
                c0 = p0; c1 = p1; c2 = p2; c3 = p3; c4 = p4; c5 = p5; c6 = p6; c7 = p7;
            }
        }
        for (uint32_t i = nb; i < n; i++) { const gz_u32x4 c = rec[i]; d_rcu_step (rc, c[0], c[1], c[2], c[3]); }
    }

    for (int k = 0; k < 5; k++) d_rcu_shift (rc);                        // RC_FinishEncode: 5 more shifts
    if (rc.nev & 3) { if (!lane) rc.ev[rc.nev >> 2] = rc.acc >> (16 * (4 - (rc.nev & 3))); }
    if (!lane) { L.n_events = rc.nev; L.touch_sink = sink; }
}

// Carry resolution, one 256-thread workgroup per leaf. With m shifts the output is m bytes: byte 0 is the coder's
// initial cache (0), byte i is T_i; the carry flag seen at shift j adds 1 at byte j-1 and ripples left through 0xFF
// bytes. Each thread owns a slice: it first finds whether a carry entering its slice from the right would leave it
// on the left (only if every byte is 0xFF after its own carries) and what it emits on its own; thread 0 chains the
// 256 slices; then every thread writes its bytes.
__global__ void __launch_bounds__(256) k_arith_carry (GzdLeaf *leaves)
{
    GzdLeaf &L = leaves[blockIdx.x];
    if (!L.active || L.engine != GZ_ENG_ARITH || L.rle) return;
    const int tid = threadIdx.x;
    const uint32_t m = L.n_events;
    const uint16_t *ev = (const uint16_t *)L.events;
    uint8_t *out = L.pay + 1;
    uint32_t *sh = (uint32_t *)gz_lds;          // [0..255] carry-out with carry-in 0, [256..511] with carry-in 1, [512..767] carry-in
    if (!tid) L.pay[0] = (uint8_t)(L.coded_n ? L.max_sym : 1);            // max_sym + 1 (256 wraps to 0), arith_dynamic.c:103-108

    const uint32_t per = (m + 255) / 256;
    const uint32_t lo = tid * per, hi = lo + per < m ? lo + per : m;
    // value of byte i before ripple: (i ? T_i : 0) + carry flag of shift i+1
    uint32_t c0 = 0, c1 = 1;
    for (uint32_t i = hi; i-- > lo; ) {
        const uint32_t raw = i ? (ev[i - 1] & 0xffu) : 0u;
        const uint32_t cin = ((uint32_t)ev[i] >> 8) & 1u;                 // carry flag recorded at shift i+1 == event index i
        c0 = (raw + cin + c0) >> 8;
        c1 = (raw + cin + c1) >> 8;
    }
    sh[tid] = c0; sh[256 + tid] = c1;
    __syncthreads ();
    if (!tid) {
        uint32_t carry = 0;
        for (int t = 255; t >= 0; t--) { sh[512 + t] = carry; carry = carry ? sh[256 + t] : sh[t]; }
    }
    __syncthreads ();
    uint32_t carry = sh[512 + tid];
    for (uint32_t i = hi; i-- > lo; ) {
        const uint32_t raw = i ? (ev[i - 1] & 0xffu) : 0u;
        const uint32_t cin = ((uint32_t)ev[i] >> 8) & 1u;
        const uint32_t v = raw + cin + carry;
        out[i] = (uint8_t)v;
        carry = v >> 8;
    }
    if (!tid) {
        if (m + 1 > L.pay_cap) { L.overflow = 1; L.pay_len = 0; }        // cannot happen: pay_cap >= 2n + 64
        else L.pay_len = m + 1;
        L.tab_len = 0;
    }
}
