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
#include <gz_intrin.h>

#define GZ_MODEL_LIMIT 65519u          // MAX_FREQ (c_simple_model.h:63)
#define GZ_MODEL_STEP  16u

struct GzDivMagic { uint32_t magic, shift; };   // q = ((((n - t) >> 1) + t) >> shift, t = mulhi(magic, n); divisor 1: shift = 0xff

// Values loaded from the leaf table arrive through vector loads, so the compiler must assume they differ per lane and
// turns every loop / branch on them into exec-mask code. They are wave-uniform: say so.
__device__ static inline uint32_t d_uniform (uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane ((int)v); }
template <typename T> __device__ static inline T *d_uniform_ptr (T *p)
{
    uint64_t a = (uint64_t)(uintptr_t)p;
    uint32_t lo = d_uniform ((uint32_t)a), hi = d_uniform ((uint32_t)(a >> 32));
    return (T *)(uintptr_t)(((uint64_t)hi << 32) | lo);
}

__device__ static inline uint32_t d_readlane (uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane ((int)v, lane); }
// (clang 22 / ROCm 7.2 has no __builtin_amdgcn_writelane; compare + select costs one more VALU op than v_writelane_b32)
__device__ static inline uint32_t d_writelane (uint32_t val, int lane, uint32_t old) { return (int)(threadIdx.x & 63) == lane ? val : old; }

// J = registers per lane holding the model (entries e = j*64 + lane), J*64 >= max_sym
template <int J>
__device__ static __forceinline__ void d_arith_model_wave (const uint8_t *in, uint32_t n, uint32_t ms, bool o1, uint32_t ctx, uint4 *recs, const GzDivMagic *magic_tab)
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
//
// Two ways through the occurrences of a 64-position chunk:
//  * one at a time (d_model_serial_step) - ~45 instructions and two vector->scalar decisions per occurrence;
//  * as a batch: as long as no occurrence causes a structural change (a swap, a halving), the triples of ALL the
//    chunk's occurrences follow from the current model plus prefix counts inside the chunk:
//        freq_j = F[p_j] + 16 * #{i < j : p_i == p_j}      cum_j = C[p_j] + 16 * #{i < j : p_i < p_j}
//        tot_j  = tot + 16 * j                              (p = list position of the occurrence's symbol)
//    and whether occurrence j would swap needs only the left neighbour's frequency at that time,
//    FL[p_j] + 16 * #{i < j : p_i == p_j - 1}. The lanes that own the positions fetch F, C, gap, FL with cross-lane
//    reads (the symbol -> list position map `where` is kept per static rank), a short loop over the occurrences
//    accumulates the counts with pure vector instructions, the longest valid prefix is accepted in one go, and only
//    the first occurrence that changes the structure (if any) takes the one-at-a-time path.
struct GzModelLane { uint32_t sym, srank, freq, cum, gap, where; };

__device__ static __forceinline__ void d_model_serial_step (GzModelLane &M, uint32_t &tot, int lane, int r, uint32_t nsym, uint32_t n_absent)
{
    const bool me = lane == r;
    // bump
    M.freq += me ? GZ_MODEL_STEP : 0u;
    M.cum  += lane > r ? GZ_MODEL_STEP : 0u;
    tot    += GZ_MODEL_STEP;
    if (tot > GZ_MODEL_LIMIT) {                                  // rare: halve, rebuild tot and cum
        M.freq -= M.freq >> 1;
        uint32_t run = 0, fsum = 0;
        for (uint32_t l = 0; l < nsym; l++) {
            run += d_readlane (M.gap, (int)l);
            M.cum = d_writelane (run, (int)l, M.cum);
            const uint32_t fq = d_readlane (M.freq, (int)l);
            run += fq; fsum += fq;
        }
        tot = fsum + n_absent;                                   // every absent entry still weighs 1
    }
    // one bubble step to the left (c_simple_model.h:139-145)
    const uint32_t f_now = d_readlane (M.freq, r);
    const bool over_absent = me && M.gap > 0;
    const bool over_left = (lane == r - 1) && M.freq < f_now;
    if (__ballot (over_absent || over_left)) {
        const uint32_t g = d_readlane (M.gap, r);
        if (g > 0) {
            M.gap += me ? 0xffffffffu : (lane == r + 1 ? 1u : 0u);  // the absent entry hops over: mine - 1, next + 1
            M.cum += me ? 0xffffffffu : 0u;
        }
        else {
            const int q = r - 1;
            const uint32_t fl = d_readlane (M.freq, q), sl = d_readlane (M.sym, q), cl = d_readlane (M.cum, q), gl = d_readlane (M.gap, q);
            const uint32_t rs = d_readlane (M.srank, r), rl = d_readlane (M.srank, q), s = d_readlane (M.sym, r);
            const bool at_q = lane == q;
            M.sym   = at_q ? s : (me ? sl : M.sym);
            M.srank = at_q ? rs : (me ? rl : M.srank);
            M.freq  = at_q ? f_now : (me ? fl : M.freq);
            M.gap   = at_q ? gl : (me ? 0u : M.gap);
            M.cum   = me ? cl + f_now : M.cum;                   // lane q keeps its cumulative
            M.where = lane == (int)rs ? (uint32_t)q : (lane == (int)rl ? (uint32_t)r : M.where);
        }
    }
}

// (force-inlined: as a called function its arguments would arrive in vector registers and every loop on them would
//  become exec-mask code)
__device__ static __forceinline__ void d_arith_model_wave_compact (const uint8_t *in, uint32_t n, uint32_t ms, bool o1, uint32_t ctx,
                                                   uint4 *recs, const GzDivMagic *magic_tab, const uint8_t *symlist,
                                                   const uint16_t *symrank, uint32_t nsym)
{
    const int lane = threadIdx.x & 63;
    const bool live = (uint32_t)lane < nsym;
    GzModelLane M;
    M.sym = live ? symlist[lane] : 0xffffffffu;
    const uint32_t prev_sym = (live && lane) ? symlist[lane - 1] : 0xffffffffu;
    M.gap = live ? (lane ? M.sym - prev_sym - 1 : M.sym) : 0;
    M.freq = live ? 1 : 0;
    M.cum = live ? M.sym : ms;                            // lane entries + absent entries before it == its byte value
    M.srank = lane; M.where = lane;
    uint32_t tot = ms;
    const uint32_t n_absent = ms - nsym;
    uint32_t b_try = 0, b_acc = 0, cool = 0; bool batch_on = true;

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
        const bool mine = pos < n && (!o1 || cp[k] == ctx);
        uint64_t todo = __ballot (mine);
        if (!todo) continue;
        const uint32_t rk = mine ? symrank[cs[k]] : 0;             // static rank of my position's symbol
        uint32_t out_cum = 0, out_freq = 0, out_tot = 0;
        while (todo) {
            if (batch_on && __popcll (todo) >= 3) {
                // ---- batch attempt over the pending occurrences
                b_try++;
                const bool occ = (todo >> lane) & 1;
                const uint32_t p  = (uint32_t)__shfl ((int)M.where, (int)rk);
                const uint32_t F  = (uint32_t)__shfl ((int)M.freq, (int)p), Cm = (uint32_t)__shfl ((int)M.cum, (int)p);
                const uint32_t G  = (uint32_t)__shfl ((int)M.gap, (int)p);
                const uint32_t FL = (uint32_t)__shfl ((int)M.freq, (int)(p ? p - 1 : 0));
                // prefix counts, one round per DISTINCT list position among the pending occurrences (a run of equal
                // symbols - the common case in binned qualities - costs one round, not one per occurrence)
                uint32_t eq = 0, lt = 0, eql = 0;
                const uint64_t below = (1ull << lane) - 1;
                for (uint64_t rem = todo; rem; ) {
                    const uint32_t q = d_readlane (p, __ffsll ((unsigned long long)rem) - 1);
                    const uint64_t mb = __ballot (occ && p == q);           // occurrences of the symbol at position q
                    rem &= ~mb;
                    const uint32_t before = (uint32_t)__popcll (mb & below);   // ... that precede me
                    eq  += (q == p) ? before : 0u;
                    lt  += (q < p) ? before : 0u;
                    eql += (q + 1 == p) ? before : 0u;
                }
                const uint32_t idx = __popcll (todo & ((1ull << lane) - 1));
                const uint32_t f = F + GZ_MODEL_STEP * eq, cu = Cm + GZ_MODEL_STEP * lt, fl = FL + GZ_MODEL_STEP * eql;
                const uint32_t tj = tot + GZ_MODEL_STEP * idx;
                const bool bad = occ && (G != 0 || (p > 0 && f + GZ_MODEL_STEP > fl) || tj + GZ_MODEL_STEP > GZ_MODEL_LIMIT);
                const uint64_t badm = __ballot (bad);
                const uint64_t acc = badm ? (todo & ((1ull << (__ffsll ((unsigned long long)badm) - 1)) - 1)) : todo;
                if (acc) {
                    if ((acc >> lane) & 1) { out_cum = cu; out_freq = f; out_tot = tj; }
                    uint32_t ceq = 0, clt = 0;
                    for (uint64_t rem = acc; rem; ) {
                        const uint32_t q = d_readlane (p, __ffsll ((unsigned long long)rem) - 1);
                        const uint64_t mb = __ballot (occ && p == q) & acc;
                        rem &= ~mb;
                        const uint32_t cnt = (uint32_t)__popcll (mb);
                        ceq += (q == (uint32_t)lane) ? cnt : 0u;
                        clt += (q < (uint32_t)lane) ? cnt : 0u;
                    }
                    M.freq += GZ_MODEL_STEP * ceq;
                    M.cum  += GZ_MODEL_STEP * clt;
                    tot    += GZ_MODEL_STEP * (uint32_t)__popcll (acc);
                    b_acc  += (uint32_t)__popcll (acc);
                    todo &= ~acc;
                    if (!todo) break;
                }
                // data whose neighbouring symbols keep overtaking each other (ties) defeats batching: stop trying
                if (b_try >= 16) { if (b_acc < 2 * b_try) { batch_on = false; cool = 0; } b_try = b_acc = 0; }
            }
            // ---- one occurrence the ordinary way: the first pending one
            if (!batch_on && ++cool >= 1024) batch_on = true;            // the model settles (warm-up swaps end): try again later
            const int b = __ffsll ((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const int r = (int)d_readlane (M.where, (int)d_readlane (rk, b));
            const uint32_t f = d_readlane (M.freq, r), cu = d_readlane (M.cum, r);
            const bool owner = lane == b;
            out_cum = owner ? cu : out_cum; out_freq = owner ? f : out_freq; out_tot = owner ? tot : out_tot;
            d_model_serial_step (M, tot, lane, r, nsym, n_absent);
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
    uint4 *tr = d_uniform_ptr ((uint4 *)L.triples);
    const uint8_t *coded = d_uniform_ptr (L.coded);
    const uint32_t n_u = d_uniform (L.coded_n), ms_u = d_uniform (ms), nsym_u = d_uniform (L.nsym);
    const bool o1_u = d_uniform (o1 ? 1u : 0u) != 0;
    if (nsym_u <= 64) { d_arith_model_wave_compact (coded, n_u, ms_u, o1_u, ctx, tr, magic_tab, L.symlist, L.symrank, nsym_u); return; }
    if (ms <= 64)       d_arith_model_wave<1> (L.coded, L.coded_n, ms, o1, ctx, tr, magic_tab);
    else if (ms <= 128) d_arith_model_wave<2> (L.coded, L.coded_n, ms, o1, ctx, tr, magic_tab);
    else                d_arith_model_wave<4> (L.coded, L.coded_n, ms, o1, ctx, tr, magic_tab);
}

// ---- range coder chain ------------------------------------------------------------------------------------------
// Measured on MI355X (tools/ubench_chain.hip): ONE wave issues an instruction every ~3.2 ns (scalar) / 2.5 ns
// (vector) whether or not it depends on the previous one, a vector->scalar hand-over (v_readlane feeding s_*) costs
// ~12 ns, and nothing but the instruction count of the wave matters. What is truly serial in the range coder
// (c_range_coder.h:97-109) is only
//        r = range / tot ;  range = (r * freq) << 8k      (k = bytes needed to bring range back above 2^24)
// `low` is not: low += cum * r followed by shifts is a big-number addition, and addition is associative. So
//   k_arith_chain  one wave per leaf, entirely on the scalar unit, ~12 instructions per symbol: SMEM record loads
//                  (8 at a time, the next 8 in flight; the lines are pulled into L2 well ahead by a vector "touch"
//                  load because SMEM returns out of order and can only be waited for as a whole), division by the
//                  per-record magic number, renormalisation by count-leading-zeros instead of a loop, and r stored
//                  four at a time. It never looks at cum or low.
//   k_arith_low    one workgroup per leaf, all threads: every thread replays low += cum * r for its own slice of the
//                  symbols from low = 0, emitting the byte that leaves the 32-bit window at every shift (plus the carry
//                  out of the window as a 9th bit) at its absolute output position (a prefix sum of the k's), and adds
//                  what is left in its window where the following slices' bytes go. Then the digits are normalised:
//                  carries ripple left inside each slice and, very rarely, across slices.
// This reproduces RC_ShiftLow's cache / pending-0xFF bookkeeping (c_range_coder.h:70-88) exactly: that logic is just
// a lazy form of the same addition ("[0, T1, T2, ...] plus 1 at the byte before every shift that saw a carry").
typedef uint32_t gz_u32x4 __attribute__((vector_size (16)));
typedef uint32_t gz_u32x16 __attribute__((vector_size (64)));
typedef const volatile __attribute__((address_space(4))) gz_u32x4 *GzConstRecP;   // volatile: keeps the prefetch a prefetch
typedef const volatile __attribute__((address_space(4))) gz_u32x16 *GzConstRec4P; // 4 records per load

#define GZ_CHAIN_BLOCK 8
#define GZ_CHAIN_TOUCH_AHEAD (16 * 1024)   // bytes: how far ahead of the scalar loads the vector unit pulls lines into L2

__device__ static inline uint32_t d_chain_step (uint32_t &range, uint32_t freq, uint32_t mg, uint32_t sh)
{
    const uint32_t t = __umulhi (mg, range);
    const uint32_t r = (((range - t) >> 1) + t) >> sh;               // range / tot, tot >= 2
    const uint32_t x = r * freq;                                     // >= 256: r >= 2^24 / 65535
    range = x << (__clz (x) & 0x18);                                 // 0, 1 or 2 bytes
    return r;
}

// one wave per leaf
__global__ void __launch_bounds__(64) k_arith_chain (GzdLeaf *leaves)
{
    GzdLeaf &L = leaves[blockIdx.x];
    if (!L.active || L.engine != GZ_ENG_ARITH || L.rle) return;
    const int lane = threadIdx.x;
    const uint32_t n = L.coded_n;
    GzConstRecP rec = (GzConstRecP)(uintptr_t)L.triples;       // padded: reads up to 64 KB past n stay inside the area
    const uint32_t *touch = (const uint32_t *)L.triples;
    uint32_t *rout = (uint32_t *)L.rvals;
    uint32_t sink = 0, range = 0xffffffffu;

    if (n && L.max_sym == 1) {
        // a stream of zero bytes: the model total starts at 1, which the multiply-shift division cannot express
        for (uint32_t i = 0; i < n; i++) {
            const gz_u32x4 c = rec[i];
            const uint32_t t = __umulhi (c[2], range);
            const uint32_t r = c[3] == 0xff ? range : (((range - t) >> 1) + t) >> c[3];
            const uint32_t x = r * c[1];
            range = x << (__clz (x) & 0x18);
            if (!lane) rout[i] = r;
        }
    }
    else {
        const uint32_t nb = n & ~(uint32_t)(GZ_CHAIN_BLOCK - 1);
        if (nb) {
            GzConstRec4P rec4 = (GzConstRec4P)(uintptr_t)L.triples;
            for (uint32_t b = 0; b < GZ_CHAIN_TOUCH_AHEAD; b += 4096) sink += touch[(b >> 2) + lane * 16];   // 64 lanes x 64 B = 4 KB
            gz_u32x16 ca = rec4[0], cb = rec4[1];
            for (uint32_t i = 0; i < nb; i += GZ_CHAIN_BLOCK) {
                const uint32_t nx = (i + GZ_CHAIN_BLOCK < nb ? i + GZ_CHAIN_BLOCK : i) >> 2;   // the last block re-reads itself
                const gz_u32x16 pa = rec4[nx], pb = rec4[nx + 1];
                if (!(i & 255)) sink += touch[((i * 16 + GZ_CHAIN_TOUCH_AHEAD) >> 2) + lane * 16];  // every 256 records = 4 KB
                const uint32_t r0 = d_chain_step (range, ca[1],  ca[2],  ca[3]);
                const uint32_t r1 = d_chain_step (range, ca[5],  ca[6],  ca[7]);
                const uint32_t r2 = d_chain_step (range, ca[9],  ca[10], ca[11]);
                const uint32_t r3 = d_chain_step (range, ca[13], ca[14], ca[15]);
                gz_scalar_store4 (rout + i, r0, r1, r2, r3);             // the chain never touches the vector unit
                const uint32_t r4 = d_chain_step (range, cb[1],  cb[2],  cb[3]);
                const uint32_t r5 = d_chain_step (range, cb[5],  cb[6],  cb[7]);
                const uint32_t r6 = d_chain_step (range, cb[9],  cb[10], cb[11]);
                const uint32_t r7 = d_chain_step (range, cb[13], cb[14], cb[15]);
                gz_scalar_store4 (rout + i + 4, r4, r5, r6, r7);
                ca = pa; cb = pb;
            }
        }
        for (uint32_t i = nb; i < n; i++) {
            const gz_u32x4 c = rec[i];
            const uint32_t r = d_chain_step (range, c[1], c[2], c[3]);
            if (!lane) rout[i] = r;
        }
    }
    gz_scalar_store_flush ();
    if (!lane) L.touch_sink = sink;
}

// ---- low: a big-number sum, one thread per symbol -----------------------------------------------------------------
// Symbol i adds a_i = cum_i * r_i into the 32-bit window of `low` after P_i bytes have left it (P = prefix sum of the
// per-symbol shift counts k_i = clz(r_i * freq_i) / 8). In the output stream (byte 0 = the coder's initial cache byte)
// that is: the four bytes of a_i are added at bytes P_i+1 .. P_i+4. Nothing else happens to low, so the whole thing is
//   k_low_count    one wave per 64-symbol slice (lane = symbol, coalesced): shifts in the slice = two ballots
//   k_low_scan     per leaf: exclusive prefix over the slices; m = total + 5 closing shifts
//   k_low_scatter  one wave per slice: P_i inside the slice again from two ballots; the lanes add their four bytes into
//                  a small LDS accumulator; the digits the slice owns are stored, the (up to 4) that spill into the
//                  following slices' digits are kept aside ...
//   k_low_resid    ... and added there
//   k_low_norm     per leaf: digits (which may exceed 255 by far: consecutive symbols without a shift pile up on the same
//                  bytes) -> bytes, carries rippling left (16 digits per thread as one 128-bit add)
#define GZ_LOW_SLICE 64
#define GZ_LOW_WG    256
#define GZ_LOW_SLICES_PER_WG 64          // each of the 4 waves of a workgroup walks 16 slices
struct GzdLowBlock { uint32_t leaf, first_slice; };

__device__ static inline uint32_t d_low_nslices (uint32_t n) { return n ? (n + GZ_LOW_SLICE - 1) / GZ_LOW_SLICE : 1; }

__global__ void __launch_bounds__(GZ_LOW_WG) k_low_count (GzdLeaf *leaves, const GzdLowBlock *blocks)
{
    const GzdLowBlock B = blocks[blockIdx.x];
    GzdLeaf &L = leaves[B.leaf];
    if (!L.active || L.engine != GZ_ENG_ARITH || L.rle) return;
    const uint32_t n = L.coded_n, ns = d_low_nslices (n);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint4 *rec = (const uint4 *)L.triples;
    const uint32_t *rv = (const uint32_t *)L.rvals;
    for (uint32_t q = 0; q < GZ_LOW_SLICES_PER_WG / 4; q++) {
        const uint32_t slice = B.first_slice + wave * (GZ_LOW_SLICES_PER_WG / 4) + q;
        if (slice >= ns) break;
        const uint32_t i = slice * GZ_LOW_SLICE + lane;
        const uint32_t k = i < n ? (uint32_t)__clz (rv[i] * rec[i].y) >> 3 : 0u;
        const uint32_t cnt = (uint32_t)__popcll (__ballot (k >= 1)) + (uint32_t)__popcll (__ballot (k == 2));
        if (!lane) ((uint32_t *)L.kpos)[slice] = cnt;
    }
}

// one 1024-thread workgroup per leaf
__global__ void __launch_bounds__(1024) k_low_scan (GzdLeaf *leaves)
{
    GzdLeaf &L = leaves[blockIdx.x];
    if (!L.active || L.engine != GZ_ENG_ARITH || L.rle) return;
    const int tid = threadIdx.x;
    uint32_t *sh = (uint32_t *)gz_lds;
    uint32_t *kpos = (uint32_t *)L.kpos;
    const uint32_t ns = d_low_nslices (L.coded_n);
    const uint32_t per = (ns + 1023) / 1024;
    const uint32_t a = tid * per < ns ? tid * per : ns, b = a + per < ns ? a + per : ns;
    uint32_t sum = 0;
    for (uint32_t i = a; i < b; i++) sum += kpos[i];
    sh[tid] = sum;
    __syncthreads ();
    if (!tid) {
        uint32_t run = 0;
        for (int t = 0; t < 1024; t++) { uint32_t c = sh[t]; sh[t] = run; run += c; }
        sh[1024] = run;
    }
    __syncthreads ();
    uint32_t run = sh[tid];
    for (uint32_t i = a; i < b; i++) { uint32_t c = kpos[i]; kpos[i] = run; run += c; }
    if (!tid) {
        const uint32_t m = sh[1024] + 5;                      // + RC_FinishEncode's 5 shifts
        kpos[ns] = sh[1024];
        L.n_events = m;
        ((uint32_t *)L.events)[0] = 0;                        // digit 0: the coder's initial cache byte
    }
}

// 4 waves, each with a 140-word LDS accumulator
__global__ void __launch_bounds__(GZ_LOW_WG) k_low_scatter (GzdLeaf *leaves, const GzdLowBlock *blocks)
{
    const GzdLowBlock B = blocks[blockIdx.x];
    GzdLeaf &L = leaves[B.leaf];
    if (!L.active || L.engine != GZ_ENG_ARITH || L.rle) return;
    const uint32_t n = L.coded_n, ns = d_low_nslices (n);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint4 *rec = (const uint4 *)L.triples;
    const uint32_t *rv = (const uint32_t *)L.rvals;
    const uint32_t *kpos = (const uint32_t *)L.kpos;
    uint32_t *dig = (uint32_t *)L.events;
    uint32_t *acc = (uint32_t *)gz_lds + wave * 144;          // up to 128 own digits + 5 closing / 4 spilling
    const uint64_t below = (1ull << lane) - 1;
    for (uint32_t q = 0; q < GZ_LOW_SLICES_PER_WG / 4; q++) {
        const uint32_t slice = B.first_slice + wave * (GZ_LOW_SLICES_PER_WG / 4) + q;
        const bool on = slice < ns;                            // (all waves keep hitting the barriers)
        const uint32_t i = slice * GZ_LOW_SLICE + lane;
        uint32_t k = 0, a = 0;
        if (on && i < n) { const uint4 c = rec[i]; const uint32_t r = rv[i]; k = (uint32_t)__clz (r * c.y) >> 3; a = c.x * r; }
        const uint64_t m1 = __ballot (k >= 1), m2 = __ballot (k == 2);
        const uint32_t P = (uint32_t)__popcll (m1 & below) + (uint32_t)__popcll (m2 & below);   // shifts before me in the slice
        const uint32_t K = (uint32_t)__popcll (m1) + (uint32_t)__popcll (m2);
        const bool last = on && slice == ns - 1;
        const uint32_t own = last ? K + 5 : K;                 // digits this slice owns: one per shift (+ the closing 5)
        for (uint32_t j = lane; j < 144; j += 64) acc[j] = 0;
        __syncthreads ();
        if (on && a) {
            atomicAdd (&acc[P],     a >> 24);
            atomicAdd (&acc[P + 1], (a >> 16) & 0xff);
            atomicAdd (&acc[P + 2], (a >> 8) & 0xff);
            atomicAdd (&acc[P + 3], a & 0xff);
        }
        __syncthreads ();
        if (on) {
            const uint32_t base = kpos[slice] + 1;             // output byte of this slice's first shift
            for (uint32_t j = lane; j < own; j += 64) dig[base + j] = acc[j];
            if (lane < 4) ((uint32_t *)L.resid)[slice * 4 + lane] = last ? 0u : acc[own + lane];
        }
        __syncthreads ();
    }
}

__global__ void __launch_bounds__(GZ_LOW_WG) k_low_resid (GzdLeaf *leaves, const GzdLowBlock *blocks)
{
    const GzdLowBlock B = blocks[blockIdx.x];
    GzdLeaf &L = leaves[B.leaf];
    if (!L.active || L.engine != GZ_ENG_ARITH || L.rle) return;
    const uint32_t ns = d_low_nslices (L.coded_n), m = L.n_events;
    if (threadIdx.x >= GZ_LOW_SLICES_PER_WG) return;
    const uint32_t slice = B.first_slice + threadIdx.x;
    if (slice + 1 >= ns) return;                               // the last slice owns everything it touches
    const uint4 r = ((const uint4 *)L.resid)[slice];
    if (!(r.x | r.y | r.z | r.w)) return;
    uint32_t *dig = (uint32_t *)L.events;
    const uint32_t at = ((const uint32_t *)L.kpos)[slice + 1] + 1;   // first digit of the next slice
    if (r.x && at < m)     atomicAdd (&dig[at], r.x);
    if (r.y && at + 1 < m) atomicAdd (&dig[at + 1], r.y);
    if (r.z && at + 2 < m) atomicAdd (&dig[at + 2], r.z);
    if (r.w && at + 3 < m) atomicAdd (&dig[at + 3], r.w);
}

// one 1024-thread workgroup per leaf. Tiles of 1024 x 16 digits are normalised from the end of the stream towards
// its start; a thread turns its 16 digits into 16 bytes (a 128-bit big-endian number in 4 words) plus a carry-out,
// carries then move one thread to the left per round as a 128-bit add until none is left (normally one round).
#define GZ_NORM_NT 1024
#define GZ_NORM_PER 16
__global__ void __launch_bounds__(GZ_NORM_NT) k_low_norm (GzdLeaf *leaves)
{
    GzdLeaf &L = leaves[blockIdx.x];
    if (!L.active || L.engine != GZ_ENG_ARITH || L.rle) return;
    const int tid = threadIdx.x;
    const uint32_t m = L.n_events;
    const uint32_t *dig = (const uint32_t *)L.events;
    uint8_t *out = L.pay + 1;
    uint32_t *sh = (uint32_t *)gz_lds;              // [0..NT] carries, [NT+8] "any carry left", [NT+9] carry into the next tile
    const uint32_t tile = GZ_NORM_NT * GZ_NORM_PER;
    const uint32_t ntiles = (m + tile - 1) / tile;
    if (!tid) sh[GZ_NORM_NT + 9] = 0;
    __syncthreads ();
    for (uint32_t t = ntiles; t-- > 0; ) {
        const uint32_t b0 = t * tile + tid * GZ_NORM_PER;
        uint32_t w[4] = { 0, 0, 0, 0 };             // w[0] most significant: bytes b0..b0+3
        uint32_t carry = 0;
        #pragma unroll
        for (int j = GZ_NORM_PER - 1; j >= 0; j--) {
            const uint32_t idx = b0 + j;
            const uint32_t v = (idx < m ? dig[idx] : 0u) + carry;
            w[j >> 2] |= (v & 0xff) << (8 * (3 - (j & 3)));
            carry = v >> 8;
        }
        // carries move one thread to the left per round. sh[t] = carry out of thread t (read by thread t-1);
        // sh[NT] = carry coming in from the tile to the right; thread 0's carries leave the tile (tile_out).
        uint32_t tile_out = carry;                   // meaningful in thread 0 only
        sh[tid] = carry;
        if (!tid) sh[GZ_NORM_NT] = sh[GZ_NORM_NT + 9];
        __syncthreads ();
        for (int round = 0; round < GZ_NORM_NT + 1; round++) {
            const uint32_t cin = sh[tid + 1];
            __syncthreads ();
            uint64_t a3 = (uint64_t)w[3] + cin;        w[3] = (uint32_t)a3;
            uint64_t a2 = (uint64_t)w[2] + (a3 >> 32); w[2] = (uint32_t)a2;
            uint64_t a1 = (uint64_t)w[1] + (a2 >> 32); w[1] = (uint32_t)a1;
            uint64_t a0 = (uint64_t)w[0] + (a1 >> 32); w[0] = (uint32_t)a0;
            const uint32_t cout = (uint32_t)(a0 >> 32);
            sh[tid] = cout;
            if (!tid) { tile_out += cout; sh[GZ_NORM_NT] = 0; sh[GZ_NORM_NT + 8] = 0; }
            __syncthreads ();
            if (tid && cout) sh[GZ_NORM_NT + 8] = 1;
            __syncthreads ();
            if (!sh[GZ_NORM_NT + 8]) break;
        }
        if (!tid) sh[GZ_NORM_NT + 9] = tile_out;
        #pragma unroll
        for (int j = 0; j < GZ_NORM_PER; j++) {
            const uint32_t idx = b0 + j;
            if (idx < m) out[idx] = (uint8_t)(w[j >> 2] >> (8 * (3 - (j & 3))));
        }
        __syncthreads ();
    }
    if (!tid) {
        L.pay[0] = (uint8_t)(L.coded_n ? L.max_sym : 1);                 // max_sym + 1 (256 wraps to 0), arith_dynamic.c:103-108
        if (m + 1 > L.pay_cap) { L.overflow = 1; L.pay_len = 0; }        // cannot happen: pay_cap >= 2n + 64
        else L.pay_len = m + 1;
        L.tab_len = 0;
    }
}
