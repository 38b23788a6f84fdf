// gz_kernels_arith.h -- the adaptive arithmetic coder (arith_dynamic.c:92-197,387-561, c_simple_model.h,
// c_range_coder.h), re-thought for a GPU. (DESIGN.md section 3 has the measurements behind every choice.)
//
// The reference walks a stream once, per symbol: search the context's frequency-sorted list for the symbol
// (accumulating the cumulative frequency), divide the range by the model total, update low/range, renormalise,
// bump the frequency, maybe halve all, maybe swap with the left neighbour. One lane doing that costs ~2000 cycles per
// symbol. The loop is taken apart by what depends on what:
//
//   k_rle_events   (run-length variant only) bytes -> coding events (model id, symbol): a segmented scan
//   k_ctx_*        a stable counting sort of the positions by context, so that a context's wave reads only its own
//   k_arith_model  the triple (cum, freq, tot) fed to the coder depends only on MODEL history, and the models never
//                  interact: one wave per (leaf, context), 64 occurrences at a time through closed formulas (prefix
//                  counts inside the batch: LDS masks + a DPP scan); order changes (swaps, hops) are events - up to 64
//                  symbols: the model in registers, events patched in one by one; wider alphabets: the model in LDS
//                  tables, the batch in ROUNDS (everything that no earlier event can reach commits at once). Writes a
//                  12-byte record per position: the reciprocal of tot as a double with cum in its low 16 bits, freq as
//                  the high word of a double.
//   k_arith_chain  what is truly serial: range -> range / tot * freq -> renormalise. One wave per leaf, three vector
//                  instructions per symbol in double precision, the state hopping a lane per symbol (gz_chain_asm.h);
//                  persistent, following the models position chunk by chunk.
//   k_low_*        low += cum * r is a big-number addition and addition is associative: one thread per symbol adds its
//                  bytes at the output position given by a prefix sum of the shift counts; then carries.
#pragma once
#include "gz_device.h"
#include "gz_devutil.h"
#include <gz_intrin.h>

#define GZ_MODEL_LIMIT 65519u          // MAX_FREQ (c_simple_model.h:63)
#define GZ_MODEL_STEP  16u

// range / tot without dividing and without the integer unit: in double precision, rounding toward zero,
//      fma (range * 2^-7, inv, 1.0) = 1 + floor (range / tot) * 2^-52        with inv >= 2^(7 - 52) / tot, too large by < 2^-32 of itself
// (the product inside the fma is exact; a quotient that is not an integer stays 1 / tot below the next one, so range / tot * error <
// 1 / tot - range * error < 1 - keeps the floor, and an inv that is never too small never falls below an exact quotient:
// tests/test_magic.py, in integer arithmetic for every total). The low half of the result IS the quotient as an integer.
//
// record of one symbol, written by the model for the chain and the low kernels: 12 bytes (rounds 1-4: 16 - { inv as a double, freq, cum })
//      lo = the low word of inv, the symbol's cum in its low 16 bits (cum < tot <= 65519) - chain and expand use the SAME double, cum and all
//      hi = the high word of inv
//      f  = the high word of the double freq * 2^45 (its low word is 0: freq < 2^16) - the chain's operand F as it is
// inv is made by the lane that writes the record, not looked up (rounds 1-4 fetched it from a 512 KB table: a 64-address gather per
// batch of every context's wave, a fifth of k_arith_model's time in the streamed form): the hardware's reciprocal seed and one Newton
// step with the constant 1 + 2^-34 give 1 / tot * (1 + 2^-34 +- 2^-40) whatever the seed's last bits are (v_rcp_f64 is good to ~2^-27;
// a seed of 2^-20 would do), clearing the 16 low bits for cum takes off < 2^-36, and 2^-34 + 2^-35 (cum) < 2^-32.
// (gz_debug_record_inv + tests/test_gpu.py::test_record_reciprocals: every total on the device against the admissible interval.)
struct __attribute__((packed, aligned (4))) GzRec { uint32_t lo, hi, f; };
#define GZ_REC_F_EXP (1023u + 45u)
__device__ static __forceinline__ void d_record_inv (uint32_t tot, uint32_t &lo, uint32_t &hi)
{
    const double d = (double)tot;
    double r = gz_rcp_f64 (d);
    r = __builtin_fma (__builtin_fma (-d, r, 1.0 + 0x1p-34), r, r);
    lo = (uint32_t)__double2loint (r) & 0xffff0000u; hi = (uint32_t)__double2hiint (r) - (45u << 20);      // * 2^-45
}
__device__ static __forceinline__ GzRec d_model_record (uint32_t cum, uint32_t freq, uint32_t tot)
{
    uint32_t lo, hi; d_record_inv (tot, lo, hi);
    GzRec r; r.lo = lo | cum; r.hi = hi;
    r.f = (uint32_t)__double2hiint ((double)freq) + (45u << 20);      // (the conversion is exact in any rounding mode)
    return r;
}
__device__ static inline uint32_t d_record_cum (uint32_t lo) { return lo & 0xffffu; }
__device__ static inline uint32_t d_record_freq (uint32_t f) { return ((f & 0xfffffu) | 0x100000u) >> (20u - ((f >> 20) - GZ_REC_F_EXP)); }

// (through a GLOBAL pointer, like the loads of gz_intrin.h: a store through a generic pointer is a FLAT one, which is also counted as an LDS operation
//  - the wait for the next batch's LDS reads then waits for the scattered store of the last one to be accepted as well)
__device__ static __forceinline__ void d_record_store (GzRec *at, GzRec v) { gz_stg_rec (at, v.lo, v.hi, v.f); }

// (tests) the reciprocal a record of total tot0 + thread would carry
__global__ void k_debug_record_inv (uint32_t tot0, uint32_t n, uint32_t *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { uint32_t lo, hi; d_record_inv (tot0 + i, lo, hi); out[2 * i] = lo; out[2 * i + 1] = hi; }
}

// Values loaded from the leaf table arrive through vector loads, so the compiler must assume they differ per lane and
// turns every loop / branch on them into exec-mask code. They are wave-uniform: say so.
__device__ static inline uint32_t d_uniform (uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane ((int)v); }
template <typename T> __device__ static inline T *d_uniform_ptr (T *p)
{
    uint64_t a = (uint64_t)(uintptr_t)p;
    uint32_t lo = d_uniform ((uint32_t)a), hi = d_uniform ((uint32_t)(a >> 32));
    return (T *)(uintptr_t)(((uint64_t)hi << 32) | lo);
}

__device__ static inline uint64_t d_uniform64 (uint64_t v) { return ((uint64_t)d_uniform ((uint32_t)(v >> 32)) << 32) | d_uniform ((uint32_t)v); }
__device__ static inline uint32_t d_readlane (uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane ((int)v, lane); }
// (clang 22 / ROCm 7.2 has no __builtin_amdgcn_writelane; compare + select costs one more VALU op than v_writelane_b32)
__device__ static inline uint32_t d_writelane (uint32_t val, int lane, uint32_t old) { return (int)(threadIdx.x & 63) == lane ? val : old; }

// ---- the adaptive models ----------------------------------------------------------------------------------------------
// Only the symbols that occur in the leaf are kept (nsym <= 256 of them), in list order, entry e in register plane
// e / 64 of lane e % 64 (J = 1, 2 or 4 planes; every quality / token stream has J = 1; with J > 1 the registers only carry
// the model between position chunks - its wave keeps it in LDS tables of the same content, see ROUNDS below). The max_sym - nsym entries of
// symbols that never occur all have frequency 1 for ever (halving leaves 1 alone) and never start a swap, so they are
// interchangeable: each present symbol just remembers how many of them sit directly in front of it (`gap`). Coding a
// symbol whose gap is > 0 swaps it with such an entry (its frequency, >= 17, always beats 1): gap--, and the next present
// symbol's gap++. With gap 0 the left neighbour is the previous entry and the ordinary "swap if now larger" applies.
// cum includes the gaps. `where` maps a symbol's static rank (its index in the leaf's sorted alphabet) to its entry.
//
// A wave works the occurrences of its context off 64 at a time. Two ways through a batch:
//  * one at a time (d_model_serial_step) - ~45 instructions and two vector->scalar decisions per occurrence;
//  * all at once: as long as no occurrence causes a structural change (a swap, a halving), the triples of ALL the
//    batch's occurrences follow from the current model plus prefix counts inside the batch:
//        freq_j = F[p_j] + 16 * #{i < j : p_i == p_j}      cum_j = C[p_j] + 16 * #{i < j : p_i < p_j}
//        tot_j  = tot + 16 * j                              (p = list position of the occurrence's symbol)
//    and whether occurrence j would swap needs only the left neighbour's frequency at that time,
//    FL[p_j] + 16 * #{i < j : p_i == p_j - 1}. The lanes fetch F, C, gap, FL with cross-lane reads (J = 1) or from the
//    LDS tables (J > 1); the counts come through LDS masks (d_batch_counts_lds); events (below) are patched in place
//    (J = 1) or end a round (J > 1).
template <int J> struct GzModel { uint32_t sym[J], srank[J], freq[J], cum[J], gap[J], where[J]; };

// R[idx / 64] of lane idx % 64, idx per lane
template <int J> __device__ static __forceinline__ uint32_t d_gather (const uint32_t (&R)[J], uint32_t idx)
{
    uint32_t v = (uint32_t)__shfl ((int)R[0], (int)(idx & 63));
    #pragma unroll
    for (int j = 1; j < J; j++) { const uint32_t t = (uint32_t)__shfl ((int)R[j], (int)(idx & 63)); v = (idx >> 6) == (uint32_t)j ? t : v; }
    return v;
}
// the same for a wave-uniform idx
// (tried: running only the copy of the code for the plane concerned, picked by a uniform branch - measurably slower than
//  doing every plane with a select)
template <int J> __device__ static __forceinline__ uint32_t d_peek (const uint32_t (&R)[J], uint32_t idx)
{
    uint32_t v = d_readlane (R[0], (int)(idx & 63));
    #pragma unroll
    for (int j = 1; j < J; j++) if ((idx >> 6) == (uint32_t)j) v = d_readlane (R[j], (int)(idx & 63));
    return v;
}

template <int J>
__device__ static __forceinline__ void d_model_serial_step (GzModel<J> &M, uint32_t &tot, int lane, uint32_t r, uint32_t nsym, uint32_t n_absent)
{
    // bump
    #pragma unroll
    for (int j = 0; j < J; j++) {
        const uint32_t e = (uint32_t)(j * 64 + lane);
        M.freq[j] += e == r ? GZ_MODEL_STEP : 0u;
        M.cum[j]  += e > r ? GZ_MODEL_STEP : 0u;
    }
    tot += GZ_MODEL_STEP;
    if (tot > GZ_MODEL_LIMIT) {                                  // rare: halve, rebuild tot and cum
        uint32_t run = 0, fsum = 0;
        #pragma unroll
        for (int j = 0; j < J; j++) {
            M.freq[j] -= M.freq[j] >> 1;
            for (uint32_t l = 0; l < 64 && (uint32_t)(j * 64) + l < nsym; l++) {
                run += d_readlane (M.gap[j], (int)l);
                M.cum[j] = d_writelane (run, (int)l, M.cum[j]);
                const uint32_t fq = d_readlane (M.freq[j], (int)l);
                run += fq; fsum += fq;
            }
        }
        tot = fsum + n_absent;                                   // every absent entry still weighs 1
    }
    // one bubble step to the left (c_simple_model.h:139-145)
    const uint32_t f_now = d_peek<J> (M.freq, r), g = d_peek<J> (M.gap, r);
    const uint32_t q = r ? r - 1 : 0;
    const uint32_t fl = d_peek<J> (M.freq, q);
    if (g > 0) {                                                 // the absent entry hops over: mine - 1, next + 1
        #pragma unroll
        for (int j = 0; j < J; j++) {
            const uint32_t e = (uint32_t)(j * 64 + lane);
            M.gap[j] += e == r ? 0xffffffffu : (e == r + 1 ? 1u : 0u);
            M.cum[j] += e == r ? 0xffffffffu : 0u;
        }
    }
    else if (r > 0 && fl < f_now) {
        const uint32_t sl = d_peek<J> (M.sym, q), cl = d_peek<J> (M.cum, q);
        const uint32_t rs = d_peek<J> (M.srank, r), rl = d_peek<J> (M.srank, q), s = d_peek<J> (M.sym, r);
        #pragma unroll
        for (int j = 0; j < J; j++) {
            const uint32_t e = (uint32_t)(j * 64 + lane);
            const bool at_q = e == q, me = e == r;
            M.sym[j]   = at_q ? s : (me ? sl : M.sym[j]);
            M.srank[j] = at_q ? rs : (me ? rl : M.srank[j]);
            M.freq[j]  = at_q ? f_now : (me ? fl : M.freq[j]);
            M.gap[j]   = me ? 0u : M.gap[j];                     // (the promoted symbol takes over its neighbour's gap: entry q keeps it)
            M.cum[j]   = me ? cl + f_now : M.cum[j];             // (entry q keeps its cumulative)
            M.where[j] = e == rs ? q : (e == rl ? r : M.where[j]);
        }
    }
}

// What a batch's occurrences add, from the list positions p of the pending ones (lanes of T): as an occurrence I need the number of
// EARLIER pending occurrences at my position (eq), at a lower one (lt) and at my left neighbour's (eql); as list entries lane + 64 j I
// need the number of ALL pending occurrences at my position (ceq[j]) and below it (clt[j]). Through the LDS: every pending occurrence
// ORs its lane bit into the word of its position (one ds_or_b64); a list entry then reads the set of ITS occurrences, an OR-scan over
// the entries (DPP, gz_wave_or_scan) makes the set of occurrences BELOW every entry, and an occurrence reads the sets of its position,
// of its left neighbour's and of everything below: the counts are popcounts of those under the mask of the earlier lanes. ~45 vector
// instructions and three trips to the LDS whatever the alphabet. (Round 1 took one ballot round per DISTINCT position of the batch - 20
// in a quality stream; rounds 2-3 one ballot per BIT of the position, most significant first, narrowing per-lane sets of "lanes that
// agree with me so far" - 35 vector instructions per bit, because those sets are 64-bit values: 210 of the ~560 instructions a batch
// of a quality context cost.)
// LDS: s_mask [2 + 64 J] (word 0 stands for "the position left of position 0": nobody; the last one for the position behind the last), s_low [64 J]
#define GZ_MLDS_OFF 512                   // (the model's tables: d_model_batch_rounds, below)
#define GZ_MLDS_BYTES 3072
#define GZ_CNT_OFF   (GZ_MLDS_OFF + GZ_MLDS_BYTES)
#define GZ_CNT_BYTES (258 * 8 + 256 * 8)
#define GZ_MODEL_LDS (GZ_CNT_OFF + GZ_CNT_BYTES)
// (s_mask [2 + 64 J], s_low [64 J]: where - a one-wave workgroup's fixed place by default; the tiled kernel's waves have one each)
template <int J>
__device__ static __forceinline__ void d_batch_counts_lds (uint32_t p, uint64_t T, int lane, uint64_t below,
                                                           uint32_t &eq, uint32_t &lt, uint32_t &eql, uint32_t (&ceq)[J], uint32_t (&clt)[J],
                                                           unsigned long long *s_mask = (unsigned long long *)(gz_lds + GZ_CNT_OFF),
                                                           unsigned long long *s_low = (unsigned long long *)(gz_lds + GZ_CNT_OFF) + 258)
{
    #pragma unroll
    for (int j = 0; j < J; j++) s_mask[1 + j * 64 + lane] = 0;
    if (!lane) s_mask[0] = 0;
    gz_wave_sync ();
    if ((T >> lane) & 1) atomicOr (&s_mask[1 + p], 1ull << lane);
    gz_wave_sync ();
    uint32_t run_lo = 0, run_hi = 0;                            // the occurrences at the planes before this one
    #pragma unroll
    for (int j = 0; j < J; j++) {
        const unsigned long long m = s_mask[1 + j * 64 + lane];
        const uint32_t m_lo = (uint32_t)m, m_hi = (uint32_t)(m >> 32);
        const uint32_t i_lo = gz_wave_or_scan (m_lo), i_hi = gz_wave_or_scan (m_hi);
        const uint32_t x_lo = (i_lo ^ m_lo) | run_lo, x_hi = (i_hi ^ m_hi) | run_hi;     // (the sets are disjoint: without mine)
        ceq[j] = (uint32_t)__popc (m_lo) + (uint32_t)__popc (m_hi);
        clt[j] = (uint32_t)__popc (x_lo) + (uint32_t)__popc (x_hi);
        s_low[j * 64 + lane] = ((unsigned long long)x_hi << 32) | x_lo;
        if (j + 1 < J) { run_lo |= d_readlane (i_lo, 63); run_hi |= d_readlane (i_hi, 63); }
    }
    gz_wave_sync ();
    const unsigned long long me = s_mask[1 + p], lf = s_mask[p], lo = s_low[p];
    eq = (uint32_t)__popcll (me & below); eql = (uint32_t)__popcll (lf & below); lt = (uint32_t)__popcll (lo & below);
}

// lanes 0 .. cnt-1 hold the next cnt occurrences of this context in stream order (rk = static rank of the symbol);
// on return they hold the (cum, freq, tot) the coder must see for them.
//
// Per attempt: every pending occurrence fetches its symbol's list position p, frequency, cumulative, gap and left
// neighbour's frequency from the model as it stands, and adds 16 x the number of earlier pending occurrences at the same /
// a lower / the left neighbour's position (one round per DISTINCT position). That is exact up to the first occurrence
// whose coding moves something. Those EVENTS are then taken in stream order, each one patching only the lanes it
// concerns instead of starting over:
//   hop   the symbol (at position r) has absent entries in front of it: one of them moves behind it. Later occurrences of
//         the symbol: cum - 1, gap - 1; later occurrences of the symbol at r + 1: gap + 1.
//   swap  the symbol a at r overtakes its left neighbour b at r - 1. Later occurrences of a: cum - freq_b(then), new left
//         neighbour r - 2; of b: cum + freq_a(then), left neighbour a; of the symbol at r + 1: left neighbour b.
//         (freq_x(then) = model frequency + 16 x earlier occurrences of x in the batch: a ballot and a mbcnt.)
// The model registers follow the order changes immediately; the batch's counts (ceq, clt per list position) are added
// at the end. Only a halving (once per ~2000 occurrences of a context) ends an attempt early: the prefix is committed,
// that one occurrence goes through d_model_serial_step, and the rest starts a new attempt.
template <int J>
__device__ static __forceinline__ void d_model_batch (GzModel<J> &M, uint32_t &tot, int lane, uint32_t cnt, uint32_t rk, uint32_t nsym,
                                                      uint32_t n_absent, uint32_t &out_cum, uint32_t &out_freq, uint32_t &out_tot, uint32_t &n_events,
                                                      unsigned long long *s_mask = (unsigned long long *)(gz_lds + GZ_CNT_OFF),
                                                      unsigned long long *s_low = (unsigned long long *)(gz_lds + GZ_CNT_OFF) + 258)
{
    uint64_t todo = cnt >= 64 ? ~0ull : (1ull << cnt) - 1;
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0;
    // (a model that has seen fewer than two occurrences per symbol is all ties: nearly every occurrence is an event, and
    //  one at a time is the cheaper way through them)
    const uint32_t settled = nsym + n_absent + 2 * GZ_MODEL_STEP * nsym;
    while (todo) {
        if (__popcll (todo) >= 3 && tot >= settled) {
            const bool occ = (todo >> lane) & 1;
            uint32_t p  = d_gather<J> (M.where, rk);
            const uint32_t F  = d_gather<J> (M.freq, p), Cm = d_gather<J> (M.cum, p);
            uint32_t G  = d_gather<J> (M.gap, p);
            const uint32_t FL = d_gather<J> (M.freq, p ? p - 1 : 0);
            // as an occurrence I count the earlier occurrences at my position / below it / at my left neighbour; as
            // list entries lane + 64 j I count what the whole batch adds to my frequency and cumulative
            uint32_t eq = 0, lt = 0, eql = 0, ceq[J], clt[J];
            d_batch_counts_lds<J> (p, todo, lane, below, eq, lt, eql, ceq, clt, s_mask, s_low);
            const uint32_t f = F + GZ_MODEL_STEP * eq, tj = tot + GZ_MODEL_STEP * gz_mbcnt (todo);
            uint32_t cu = Cm + GZ_MODEL_STEP * lt, fl = FL + GZ_MODEL_STEP * eql;
            // the first occurrence that would push the total over the limit ends the attempt
            const uint64_t halve_m = __ballot (occ && tj + GZ_MODEL_STEP > GZ_MODEL_LIMIT);
            const uint64_t acc = halve_m ? todo & ((1ull << (__ffsll ((unsigned long long)halve_m) - 1)) - 1) : todo;
            // ---- events, in stream order
            for (uint64_t pend = acc; ; ) {
                const uint64_t badm = __ballot (G != 0 || (p > 0 && f + GZ_MODEL_STEP > fl)) & pend;
                if (!badm) break;
                const int jv = __ffsll ((unsigned long long)badm) - 1;
                pend &= ~((2ull << jv) - 1);
                n_events++;
                const bool later = lane > jv;
                const uint32_t r = d_readlane (p, jv);
                if (d_readlane (G, jv)) {                                   // hop
                    #pragma unroll
                    for (int j = 0; j < J; j++) {
                        const uint32_t e = (uint32_t)(j * 64 + lane);
                        M.gap[j] += e == r ? 0xffffffffu : (e == r + 1 ? 1u : 0u);
                        M.cum[j] += e == r ? 0xffffffffu : 0u;
                    }
                    cu -= (later && p == r) ? 1u : 0u;
                    G  += later ? (p == r ? 0xffffffffu : (p == r + 1 ? 1u : 0u)) : 0u;
                }
                else {                                                      // swap list positions q = r - 1 and r
                    const uint32_t q = r - 1, q2 = q ? q - 1 : 0;
                    const bool is_a = p == r, is_b = p == q, is_c = p == r + 1;
                    const uint32_t Fa = d_peek<J> (M.freq, r), Fb = d_peek<J> (M.freq, q), F2 = d_peek<J> (M.freq, q2);
                    const uint32_t fa = Fa + GZ_MODEL_STEP * gz_mbcnt (__ballot (is_a) & todo);
                    const uint32_t fb = Fb + GZ_MODEL_STEP * gz_mbcnt (__ballot (is_b) & todo);
                    const uint32_t f2 = F2 + GZ_MODEL_STEP * gz_mbcnt (__ballot (q && p == q2) & todo);
                    const uint32_t gl = d_peek<J> (M.gap, q);
                    if (later) {
                        cu = is_a ? cu - fb : (is_b ? cu + fa : cu);
                        fl = is_a ? f2 : (is_b ? fa : (is_c ? fb : fl));
                        G  = is_a ? gl : (is_b ? 0u : G);
                    }
                    p = is_a ? q : (is_b ? r : p);                          // (all lanes: equal symbols keep equal positions)
                    // the model registers ...
                    const uint32_t sb = d_peek<J> (M.sym, q), cb = d_peek<J> (M.cum, q);
                    const uint32_t ra = d_peek<J> (M.srank, r), rb = d_peek<J> (M.srank, q), sa = d_peek<J> (M.sym, r);
                    // ... and the batch's counts per list position
                    const uint32_t ea = d_peek<J> (ceq, r), eb = d_peek<J> (ceq, q), lq = d_peek<J> (clt, q);
                    #pragma unroll
                    for (int j = 0; j < J; j++) {
                        const uint32_t e = (uint32_t)(j * 64 + lane);
                        const bool at_r = e == r, at_q = e == q;
                        M.sym[j]   = at_q ? sa : (at_r ? sb : M.sym[j]);
                        M.srank[j] = at_q ? ra : (at_r ? rb : M.srank[j]);
                        M.freq[j]  = at_q ? Fa : (at_r ? Fb : M.freq[j]);
                        M.gap[j]   = at_r ? 0u : M.gap[j];                  // (a takes over b's gap: entry q keeps it)
                        M.cum[j]   = at_r ? cb + Fa : M.cum[j];             // (entry q keeps its cumulative)
                        M.where[j] = e == ra ? q : (e == rb ? r : M.where[j]);
                        ceq[j] = at_q ? ea : (at_r ? eb : ceq[j]);
                        clt[j] = at_r ? lq + ea : clt[j];
                    }
                }
            }
            if (halve_m) {                                                  // commit the prefix only: recount it
                uint32_t d0, d1, d2;
                d_batch_counts_lds<J> (p, acc, lane, below, d0, d1, d2, ceq, clt, s_mask, s_low);
            }
            if ((acc >> lane) & 1) { out_cum = cu; out_freq = f; out_tot = tj; }
            #pragma unroll
            for (int j = 0; j < J; j++) { M.freq[j] += GZ_MODEL_STEP * ceq[j]; M.cum[j] += GZ_MODEL_STEP * clt[j]; }
            tot  += GZ_MODEL_STEP * (uint32_t)__popcll (acc);
            todo &= ~acc;
            if (!todo) break;
        }
        // ---- one occurrence the ordinary way: the first pending one
        const int b = __ffsll ((unsigned long long)todo) - 1;
        todo &= todo - 1;
        const uint32_t r = d_peek<J> (M.where, d_readlane (rk, b));
        const uint32_t f = d_peek<J> (M.freq, r), cu = d_peek<J> (M.cum, r);
        const bool owner = lane == b;
        out_cum = owner ? cu : out_cum; out_freq = owner ? f : out_freq; out_tot = owner ? tot : out_tot;
        d_model_serial_step<J> (M, tot, lane, r, nsym, n_absent);
    }
}

__device__ static __forceinline__ uint32_t d_wave_incl_scan (uint32_t v, int lane)
{
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl ((int)v, lane >= d ? lane - d : lane); v += lane >= d ? o : 0u; }
    return v;
}

// ---- contexts whose symbols keep overtaking each other: ROUNDS ------------------------------------------------------------------------
// Near-uniform frequencies (the low and high bytes of coordinates, hashes, packed qualities ...) make every other occurrence an order
// change, and the register batch above takes those one at a time, patching the batch's later lanes after each (~0.45 us per event on a
// 4-plane model: 850 ns per symbol on such a context; the first remedy - the model in the LDS and the occurrences one by one the way the
// reference does it, every look-up a broadcast read - still cost 0.32 us per occurrence). But what one occurrence's order change can touch
// is small: a swap of list positions r - 1 and r matters to later occurrences of the symbols at r - 1, r and r + 1 (the last one's left
// neighbour changes), a hop at r to those at r and r + 1 - and to nobody else: their frequencies, cumulatives (a swap leaves the sum in
// front of everything behind it alone) and neighbours are what the batch formulas of d_model_batch say. So the model of a WIDE alphabet
// (more than 64 symbols: two or four register planes otherwise) lives in the LDS for as long as its wave works on it (tables by list
// position: frequency, cumulative, gap, symbol rank; by symbol rank: position), and a ROUND takes every pending occurrence of a batch
// through the formulas at once - positions and entries gathered from the tables, prefix counts through the masks of d_batch_counts_lds -
// finds the occurrences that change the order ("events"), and commits the occurrences in front of the first one that has an EARLIER event
// at its own position or a neighbouring one: for all of those the formulas are exact, and their events do not touch each other, so each
// event's lane writes its own swap (or hop) into the tables. The rest is the next round's, gathered afresh - no patching of later lanes.
// Simulated on the model (tools/rounds_sim.py): a near-uniform 256-symbol context has 39 events per batch of 64 and needs 4.9 rounds, a
// 90-symbol geometric one 16 events and 5.7 rounds, 16 hot + 240 rare symbols 30 events and 9.6 rounds.
// Measured against what it replaced (register batches, and - picked by the clock, batch by batch - the serial LDS batches): binned FASTQ
// 44.7 -> 27.4 ms per step, BAM from text 53.4 -> 30.6, from records 41.8 -> 31.6, one VCF VBlock 1981 -> 1024 ms. One-plane models (a
// quality stream's contexts) are as fast either way (-DGZ_ROUNDS_MINJ=1: default step 46.0 -> 46.3 ms, streamed 175.0 -> 177.5) and stay
// in registers.
// The occurrence that pushes the total over the limit (once per ~4 000 occurrences) ends a round in front of it and is taken on its own.
// LDS: bytes GZ_MLDS_OFF .. of the workgroup's (one wave's) dynamic LDS, the masks behind them (GZ_CNT_OFF).
#define GZ_RT_FREQ (GZ_MLDS_OFF)           // uint32_t [256] by list position
#define GZ_RT_CUM  (GZ_MLDS_OFF + 1024)    // uint32_t [256] by list position
#define GZ_RT_GAP  (GZ_MLDS_OFF + 2048)    // uint16_t [256] by list position
#define GZ_RT_RANK (GZ_MLDS_OFF + 2560)    // uint8_t  [256] list position -> symbol rank
#define GZ_RT_POS  (GZ_MLDS_OFF + 2816)    // uint8_t  [256] symbol rank -> list position
// the model's registers (entry lane + 64 j of plane j) -> the tables, once per wave and context
template <int J>
__device__ static __forceinline__ void d_rounds_tables_in (const GzModel<J> &M, int lane, uint32_t nsym)
{
    uint32_t *t_freq = (uint32_t *)(gz_lds + GZ_RT_FREQ), *t_cum = (uint32_t *)(gz_lds + GZ_RT_CUM);
    uint16_t *t_gap = (uint16_t *)(gz_lds + GZ_RT_GAP);
    uint8_t *t_rank = gz_lds + GZ_RT_RANK, *t_pos = gz_lds + GZ_RT_POS;
    gz_wave_sync ();                                               // (the previous context of this workgroup is done with them)
    #pragma unroll
    for (int j = 0; j < J; j++) {
        const uint32_t e = (uint32_t)(j * 64 + lane);
        if (e < nsym) { t_freq[e] = M.freq[j]; t_cum[e] = M.cum[j]; t_gap[e] = (uint16_t)M.gap[j]; t_rank[e] = (uint8_t)M.srank[j]; t_pos[e] = (uint8_t)M.where[j]; }
    }
    gz_wave_sync ();
}
// ... and back, for the state that travels to the next position chunk
template <int J>
__device__ static __forceinline__ void d_rounds_tables_out (GzModel<J> &M, int lane, uint32_t nsym, const uint8_t *symlist)
{
    const uint32_t *t_freq = (const uint32_t *)(gz_lds + GZ_RT_FREQ), *t_cum = (const uint32_t *)(gz_lds + GZ_RT_CUM);
    const uint16_t *t_gap = (const uint16_t *)(gz_lds + GZ_RT_GAP);
    const uint8_t *t_rank = gz_lds + GZ_RT_RANK, *t_pos = gz_lds + GZ_RT_POS;
    gz_wave_sync ();
    #pragma unroll
    for (int j = 0; j < J; j++) {
        const uint32_t e = (uint32_t)(j * 64 + lane);
        if (e < nsym) { M.freq[j] = t_freq[e]; M.cum[j] = t_cum[e]; M.gap[j] = t_gap[e]; M.srank[j] = t_rank[e]; M.where[j] = t_pos[e]; M.sym[j] = symlist[M.srank[j]]; }
    }
}

// lanes 0 .. cnt-1 hold the next cnt occurrences of this context in stream order (rk = rank of the symbol in the context's alphabet);
// on return they hold the (cum, freq, tot) the coder must see for them, and the tables have moved on
template <int J>
__device__ static __forceinline__ void d_model_batch_rounds (uint32_t &tot, int lane, uint32_t cnt, uint32_t rk, uint32_t nsym, uint32_t n_absent,
                                                             uint32_t &out_cum, uint32_t &out_freq, uint32_t &out_tot, uint32_t &n_changes)
{
    uint32_t *t_freq = (uint32_t *)(gz_lds + GZ_RT_FREQ), *t_cum = (uint32_t *)(gz_lds + GZ_RT_CUM);
    uint16_t *t_gap = (uint16_t *)(gz_lds + GZ_RT_GAP);
    uint8_t *t_rank = gz_lds + GZ_RT_RANK, *t_pos = gz_lds + GZ_RT_POS;
    unsigned long long *s_mask = (unsigned long long *)(gz_lds + GZ_CNT_OFF), *s_low = s_mask + 258;
    uint64_t todo = cnt >= 64 ? ~0ull : (1ull << cnt) - 1;
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0, self = 1ull << lane;
    uint32_t t = d_uniform (tot), changes = 0;
    // (the masks are all zero between rounds: whoever sets a bit clears its word again)
    #pragma unroll
    for (int j = 0; j < J; j++) s_mask[1 + j * 64 + lane] = 0;
    if (lane < 2) s_mask[lane ? 1 + 64 * J : 0] = 0;
    while (todo) {
        if (t + GZ_MODEL_STEP > GZ_MODEL_LIMIT) {
            // ---- the occurrence that halves the model, on its own (c_simple_model.h:124-146): bump, halve, rebuild, one bubble step
            const int b = __ffsll ((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const uint32_t s = d_readlane (rk, b);
            const uint32_t p = d_uniform (t_pos[s]), q = p ? p - 1 : 0;
            const uint32_t f = d_uniform (t_freq[p]), g = d_uniform (t_gap[p]), c = d_uniform (t_cum[p]), rb = d_uniform (t_rank[q]);
            if (lane == b) { out_cum = c; out_freq = f; out_tot = t; }
            gz_wave_sync ();                                       // (everybody has read)
            uint32_t run = 0, fsum = 0;
            #pragma unroll
            for (int j = 0; j < J; j++) {
                const uint32_t e = (uint32_t)(j * 64 + lane);
                uint32_t x = 0, gp = 0;
                if (e < nsym) { x = t_freq[e] + (e == p ? GZ_MODEL_STEP : 0u); x -= x >> 1; t_freq[e] = x; gp = t_gap[e]; }
                const uint32_t incl = d_wave_incl_scan (x + gp, lane);
                if (e < nsym) t_cum[e] = run + incl - x;           // everything in front of me + my own gap
                run += (uint32_t)__shfl ((int)incl, 63);
                fsum += (uint32_t)__shfl ((int)d_wave_incl_scan (x, lane), 63);
            }
            t = d_uniform (fsum + n_absent);
            gz_wave_sync ();
            const uint32_t fn = d_uniform (t_freq[p]), fl = d_uniform (t_freq[q]);
            gz_wave_sync ();
            if (g > 0) {
                if (!lane) { t_gap[p] = (uint16_t)(g - 1); if (p + 1 < nsym) t_gap[p + 1] = (uint16_t)(t_gap[p + 1] + 1); t_cum[p] -= 1; }
                changes++;
            }
            else if (p > 0 && fl < fn) {
                if (!lane) { t_freq[q] = fn; t_freq[p] = fl; t_rank[q] = (uint8_t)s; t_rank[p] = (uint8_t)rb; t_pos[s] = (uint8_t)q; t_pos[rb] = (uint8_t)p; t_cum[p] = t_cum[p] - fl + fn; }
                changes++;
            }
            gz_wave_sync ();
            continue;
        }
        const bool pend = (todo >> lane) & 1;
        const uint32_t p = t_pos[rk], q = p ? p - 1 : 0;
        const uint32_t f = t_freq[p], fl = t_freq[q], g = t_gap[p], c = t_cum[p], rb = t_rank[q];
        const uint32_t g_nx = t_gap[p + 1 < nsym ? p + 1 : p];
        gz_wave_sync ();
        if (pend) atomicOr (&s_mask[1 + p], (unsigned long long)self);
        gz_wave_sync ();
        uint32_t x_lo[J], x_hi[J], run_lo = 0, run_hi = 0;
        #pragma unroll
        for (int j = 0; j < J; j++) {
            const unsigned long long m = s_mask[1 + j * 64 + lane];
            const uint32_t m_lo = (uint32_t)m, m_hi = (uint32_t)(m >> 32);
            const uint32_t i_lo = gz_wave_or_scan (m_lo), i_hi = gz_wave_or_scan (m_hi);
            x_lo[j] = (i_lo ^ m_lo) | run_lo; x_hi[j] = (i_hi ^ m_hi) | run_hi;           // the pending occurrences at positions below entry lane + 64 j
            s_low[j * 64 + lane] = ((unsigned long long)x_hi[j] << 32) | x_lo[j];
            if (j + 1 < J) { run_lo |= d_readlane (i_lo, 63); run_hi |= d_readlane (i_hi, 63); }
        }
        gz_wave_sync ();
        const unsigned long long me = s_mask[1 + p], lf = s_mask[p], rt = s_mask[2 + p], lo = s_low[p];
        gz_wave_sync ();
        if (pend) s_mask[1 + p] = 0;
        const uint32_t fj = f + GZ_MODEL_STEP * (uint32_t)__popcll (me & below), flj = fl + GZ_MODEL_STEP * (uint32_t)__popcll (lf & below);
        const uint32_t cj = c + GZ_MODEL_STEP * (uint32_t)__popcll (lo & below), tj = t + GZ_MODEL_STEP * gz_mbcnt (todo);
        const bool bad = pend && (g != 0 || (p > 0 && fj + GZ_MODEL_STEP > flj));
        const uint64_t badm = __ballot (bad);
        const bool dirty = pend && ((me | lf | rt) & badm & below) != 0;
        const uint64_t stop = __ballot (dirty || (pend && tj + GZ_MODEL_STEP > GZ_MODEL_LIMIT));
        const uint64_t C = stop ? todo & ((1ull << (__ffsll ((unsigned long long)stop) - 1)) - 1) : todo;   // (never empty: the first pending occurrence has nobody in front of it)
        const bool com = (C >> lane) & 1;
        if (com) { out_cum = cj; out_freq = fj; out_tot = tj; }
        // the tables: what the committed occurrences add (the last one at a position writes its frequency; every entry gains 16 per occurrence below it) ...
        if (com && (me & C & ~below & ~self) == 0) t_freq[p] = fj + GZ_MODEL_STEP;
        #pragma unroll
        for (int j = 0; j < J; j++) {
            const uint32_t e = (uint32_t)(j * 64 + lane);
            // (an LDS add, not a read and a write: nothing to wait for; entries past the alphabet's end are never looked at)
            atomicAdd (&t_cum[e], GZ_MODEL_STEP * ((uint32_t)__popc (x_lo[j] & (uint32_t)C) + (uint32_t)__popc (x_hi[j] & (uint32_t)(C >> 32))));
        }
        gz_wave_sync ();
        // ... then their order changes, every one by its own lane: none of them touches what another one does
        // (nothing is read back: entry p's cumulative after the additions above is c + 16 x the committed occurrences below p)
        if (com && bad) {
            const uint32_t c_now = c + GZ_MODEL_STEP * (uint32_t)__popcll (lo & C);
            if (g) { t_gap[p] = (uint16_t)(g - 1); if (p + 1 < nsym) t_gap[p + 1] = (uint16_t)(g_nx + 1); t_cum[p] = c_now - 1; }
            else {
                t_freq[q] = fj + GZ_MODEL_STEP; t_freq[p] = flj;
                t_rank[q] = (uint8_t)rk; t_rank[p] = (uint8_t)rb; t_pos[rk] = (uint8_t)q; t_pos[rb] = (uint8_t)p;
                t_cum[p] = c_now - flj + fj + GZ_MODEL_STEP;
            }
        }
        changes += (uint32_t)__popcll (badm & C);
        gz_wave_sync ();
        t += GZ_MODEL_STEP * (uint32_t)__popcll (C);
        todo &= ~C;
#ifdef GZ_MODEL_PHASES
        changes += 1u << 16;                                       // (rounds, for the phase report)
#endif
    }
    tot = t; n_changes = changes;
}

// ---- grouping the positions of an order-1 leaf by context ---------------------------------------------------------
// Without it every context's wave has to scan the whole stream for its positions (with ~40 contexts in a quality
// stream that scan was most of the instructions the model kernel executed). A stable counting sort by the context byte:
//   k_ctx_count    per tile of 4096 positions: how many positions each context has in it (LDS counters)
//   k_ctx_scan     per leaf: tile/context counts -> start index of every (tile, context) in the sorted order
//   k_ctx_scatter  per tile, one wave walking it 64 positions at a time: index = running count of my context (an LDS
//                  gather) + my rank among the lanes of this group with the same context (a ballot per bit of the
//                  context number: the lanes that agree with me on all of them)
// The chunk size of the model / chain pipeline is a multiple of the tile size, and every position chunk is sorted on its
// own (into entries [p0, p1) of the lists) right before its models run: the sort of chunk k+1 hides behind the chain of chunk k.
#define GZ_CTX_TILE 4096u

// Workgroups go to the 8 XCDs of the device round robin by their linear id, and every XCD has an L2 of its own. The workgroups of ONE
// leaf store into the same cache lines - the contexts' waves of the model kernel the 16-byte records of neighbouring positions, the tiles
// of the sort the ends of the same runs - and partial lines written through different L2s cost both time and traffic (tools/
// ubench_scatter.hip: records at stride 4 .. 40 - 0.9 TB/s and 2 x the bytes when the neighbours come from other XCDs, 2.9 - 3.4 TB/s
// and 1.0 - 1.2 x when they share one). So the x extent of a (leaf, y) grid is padded to a multiple of 8: the linear id y * extent +
// leaf is then congruent to the leaf's index modulo 8 for every y - all workgroups of a leaf sit on one XCD. (The order stays
// row by row: y = 0 of every leaf first - the busiest contexts are the low ones, and a long pole that starts late is a late step.
// Tried: 8 leaves x all y as consecutive ids - binned FASTQ 44.1 -> 52.3 ms, BAM 52.7 -> 55.6.)
#define GZ_XCD_GRID(li, by, n_list) const uint32_t li = blockIdx.x, by = blockIdx.y; if (li >= (n_list)) return
#define GZ_XCD_DIM(n_list, Y) dim3 ((((uint32_t)(n_list) + 7u) / 8u) * 8u, (uint32_t)(Y))

// (order 1, at most 64 distinct bytes: k_arith_model_tiled sorts inside its tiles and runs every context of the leaf itself)
__device__ static inline bool d_leaf_tiled (const GzdLeaf &L)
{
    return L.tile_models && L.active && L.engine == GZ_ENG_ARITH && L.o1 && !L.rle && L.nsym <= 64 && L.arith_n;
}
__device__ static inline bool d_ctx_sorted (const GzdLeaf &L) { return L.active && L.engine == GZ_ENG_ARITH && (L.o1 || L.rle) && L.arith_n && !d_leaf_tiled (L); }
__device__ static inline uint32_t d_ctx_ntiles (uint32_t n) { return (n + GZ_CTX_TILE - 1) / GZ_CTX_TILE; }
// context (model id) of coding event `pos`: the byte before it, or what k_rle_events wrote down
__device__ static inline uint32_t d_ctx_of (const GzdLeaf &L, const uint8_t *in, const uint16_t *ev_ctx, uint32_t pos)
{
    return ev_ctx ? ev_ctx[pos] : (pos ? in[pos - 1] : 0u);
}
#define GZ_CTX_MAX 768                      // LDS counters: 256 contexts, or the 514 models of the run-length variant

// grid: GZ_XCD_GRID over (listed leaves, tiles per chunk), 64 threads; positions [p0, p0 + chunk)
__global__ void __launch_bounds__(64) k_ctx_count (GzdLeaf *leaves, const uint32_t *list, uint32_t n_list, uint32_t p0, uint32_t chunk)
{
    GZ_XCD_GRID (li, by, n_list);
    GzdLeaf &L = leaves[list[li]];
    if (!d_ctx_sorted (L)) return;
    const uint32_t n = L.arith_n, tile = p0 / GZ_CTX_TILE + by, t0 = tile * GZ_CTX_TILE, nctx = L.nctx;
    if (t0 >= n || t0 - p0 >= chunk) return;
    const int lane = threadIdx.x;
    uint32_t *cnt = (uint32_t *)gz_lds;
    for (uint32_t e = lane; e < nctx; e += 64) cnt[e] = 0;
    __syncthreads ();
    const uint8_t *in = L.coded; const uint16_t *ev_ctx = L.ev_ctx;
    for (uint32_t g = 0; g < GZ_CTX_TILE; g += 64) {
        const uint32_t pos = t0 + g + lane;
        if (pos < n) atomicAdd (&cnt[d_ctx_of (L, in, ev_ctx, pos)], 1u);
    }
    __syncthreads ();
    for (uint32_t e = lane; e < nctx; e += 64) L.ctxoff[(size_t)tile * nctx + e] = cnt[e];
}

// one 256-thread workgroup per leaf: thread c owns contexts c, c + 256, c + 512. The occurrences of positions
// [p0, p0 + chunk) take entries [p0, p1) of the sorted lists, grouped by context: every position chunk is sorted on
// its own, just before its models run (ctxend = where each context's run of this chunk ends).
// row: which row of ctxend this chunk's run ends go to (the sort runs ahead of the models: a row per chunk in flight)
__global__ void __launch_bounds__(256) k_ctx_scan (GzdLeaf *leaves, const uint32_t *list, uint32_t p0, uint32_t chunk, uint32_t row)
{
    GzdLeaf &L = leaves[list[blockIdx.x]];
    if (!d_ctx_sorted (L) || L.arith_n <= p0) return;
    const uint32_t n = L.arith_n, p1 = (n - p0 > chunk) ? p0 + chunk : n, nctx = L.nctx;
    const uint32_t t0 = p0 / GZ_CTX_TILE, t1 = d_ctx_ntiles (p1);
    uint32_t *off = L.ctxoff, *sh = (uint32_t *)gz_lds;
    for (uint32_t c = threadIdx.x; c < nctx; c += 256) {
        uint32_t run = 0;
        for (uint32_t t = t0; t < t1; t++) { const uint32_t v = off[(size_t)t * nctx + c]; off[(size_t)t * nctx + c] = run; run += v; }
        sh[c] = run;
    }
    __syncthreads ();
    if (!threadIdx.x) { uint32_t b = p0; for (uint32_t e = 0; e < nctx; e++) { const uint32_t v = sh[e]; sh[e] = b; b += v; sh[GZ_CTX_MAX + e] = v; } }
    __syncthreads ();
    for (uint32_t c = threadIdx.x; c < nctx; c += 256) {
        const uint32_t base = sh[c];
        for (uint32_t t = t0; t < t1; t++) off[(size_t)t * nctx + c] += base;
        L.ctxend[(size_t)row * nctx + c] = base + sh[GZ_CTX_MAX + c];   // (a row per chunk: the sort runs ahead of the models)
    }
}

// grid: GZ_XCD_GRID over (listed leaves, tiles per chunk), 64 threads
__global__ void __launch_bounds__(64) k_ctx_scatter (GzdLeaf *leaves, const uint32_t *list, uint32_t n_list, uint32_t p0, uint32_t chunk)
{
    GZ_XCD_GRID (li, by, n_list);
    GzdLeaf &L = leaves[list[li]];
    if (!d_ctx_sorted (L)) return;
    const uint32_t n = L.arith_n, tile = p0 / GZ_CTX_TILE + by, t0 = tile * GZ_CTX_TILE, nctx = L.nctx;
    if (t0 >= n || t0 - p0 >= chunk) return;
    const int lane = threadIdx.x;
    uint32_t *cnt = (uint32_t *)gz_lds;
    uint8_t *rank_of = gz_lds + GZ_CTX_MAX * 4;
    for (uint32_t e = lane; e < nctx; e += 64) cnt[e] = L.ctxoff[(size_t)tile * nctx + e];
    for (int e = lane; e < 256; e += 64) rank_of[e] = (uint8_t)L.symrank[e];
    __syncthreads ();
    const uint8_t *in = L.coded, *ev_sym = L.ev_sym; const uint16_t *ev_ctx = L.ev_ctx;
    uint32_t *spos = L.spos; uint8_t *srk = L.srk;
    const uint32_t cbits = nctx > 256 ? 10 : 8;
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0;
    // (the bytes of the next group are requested before this group is worked on)
    uint32_t nx_c = 0xffffffffu, nx_s = 0;
    if (t0 + lane < n) { nx_c = d_ctx_of (L, in, ev_ctx, t0 + lane); nx_s = ev_sym ? ev_sym[t0 + lane] : in[t0 + lane]; }
    for (uint32_t g = 0; g < GZ_CTX_TILE && t0 + g < n; g += 64) {
        const uint32_t pos = t0 + g + lane;
        const bool valid = pos < n;
        const uint32_t c = nx_c, s = nx_s;
        nx_c = 0xffffffffu;
        if (g + 64 < GZ_CTX_TILE && pos + 64 < n) { nx_c = d_ctx_of (L, in, ev_ctx, pos + 64); nx_s = ev_sym ? ev_sym[pos + 64] : in[pos + 64]; }
        const uint32_t base = valid ? cnt[c] : 0u;
        // my rank among the lanes of this group with my context: the lanes that agree with me on every bit of the context number
        // (a ballot per BIT - 8, or 10 for the run-length variant's 514 models - where the first version took one per DISTINCT
        //  context of the group, ~30 in a quality stream)
        uint64_t same = __ballot (valid);
        for (uint32_t b = 0; b < cbits; b++) {
            const bool one = (c >> b) & 1;
            const uint64_t B = __ballot (one);
            same &= one ? B : ~B;
        }
        const uint32_t within = (uint32_t)__popcll (same & below);
        if (valid) {
            // (one scattered 4-byte store per position instead of a 4- and a 1-byte one: a leaf of fewer than 2^24 positions - srk == NULL -
            //  keeps the symbol's rank in the low byte of its entry)
            const uint32_t rk = ev_sym ? (s & 0xff) : rank_of[s];
            if (srk) { gz_stg_u32 (spos + base + within, pos); gz_stg_u8 (srk + base + within, rk); }
            else gz_stg_u32 (spos + base + within, (pos << 8) | rk);
            atomicAdd (&cnt[c], 1u);
        }
        __syncthreads ();                                  // (one wave: the next group's gather sees this group's counts)
    }
}

// ---- the run-length variant's coding events (arith_dynamic.c:387-448,496-561) ---------------------------------------
// A run of r + 1 equal bytes b is coded as: the literal b in the literal model of the previous literal (order 1) or model
// 0, then r as "base-4-ish" digits, 3 meaning "more follows": floor (r / 3) threes and a final digit r % 3 - the first
// digit in run model b, the second in run model 256, the rest in run model 257. Seen from the positions of the run
// (offset j = 0 .. r): position 0 emits the literal, every position with j > 0 and j % 3 == 0 a three, the last position
// the final digit. So every position emits 0 to 3 events and knows them from j and "am I the last": a segmented scan.
// One 1024-thread workgroup per leaf walks it 1024 positions at a time (offset of the running run and number of events
// so far carried along) and writes for every event its model id (run models: 256 +) and symbol.
__global__ void __launch_bounds__(1024) k_rle_events (GzdLeaf *leaves, const uint32_t *list)
{
    GzdLeaf &L = leaves[list[blockIdx.x]];
    if (!L.active || L.engine != GZ_ENG_ARITH || !L.rle) return;
    const uint32_t n = L.coded_n, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint8_t *in = L.coded;
    const bool o1 = L.o1;
    uint16_t *ev_ctx = L.ev_ctx; uint8_t *ev_sym = L.ev_sym;
    uint32_t *sh = (uint32_t *)gz_lds;                         // [0..15] wave aggregates (run start + 1)  [16..31] wave event counts
    uint32_t start_carry = 0, ev_carry = 0;                    // start (+1) of the run reaching into this tile; events before it
    uint32_t prev_lit_carry = 0;                               // the literal before that run
    for (uint32_t t0 = 0; t0 < n; t0 += 1024) {
        const uint32_t i = t0 + tid;
        const bool valid = i < n;
        const uint32_t b = valid ? in[i] : 0, bp = (valid && i) ? in[i - 1] : 0x100u, bn = (valid && i + 1 < n) ? in[i + 1] : 0x100u;
        const bool is_start = valid && (i == 0 || bp != b), is_last = valid && bn != b;
        // inclusive max-scan of (start position + 1) over the tile; 0 = none yet (the carried run continues)
        uint32_t st = is_start ? i + 1 : 0;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl ((int)st, (int)(lane >= (uint32_t)d ? lane - d : lane)); st = (lane >= (uint32_t)d && o > st) ? o : st; }
        if (lane == 63) sh[wave] = st;
        __syncthreads ();
        uint32_t before = start_carry;
        for (uint32_t w = 0; w < wave; w++) before = sh[w] > before ? sh[w] : before;
        st = st > before ? st : before;                        // start (+1) of my run
        const uint32_t j = valid ? i - (st - 1) : 0;           // my offset inside it
        // the literal before my run (context of my run's literal): the byte before the run start
        const uint32_t s0 = st - 1;
        const uint32_t lit_ctx = (o1 && valid && s0) ? in[s0 - 1] : 0u;
        const uint32_t e_lit = (valid && j == 0) ? 1u : 0u, e_three = (valid && j > 0 && j % 3 == 0) ? 1u : 0u, e_fin = is_last ? 1u : 0u;
        const uint32_t cnt = e_lit + e_three + e_fin;
        // exclusive sum-scan of the event counts
        uint32_t inc = cnt;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl ((int)inc, (int)(lane >= (uint32_t)d ? lane - d : lane)); inc += lane >= (uint32_t)d ? o : 0u; }
        if (lane == 63) sh[16 + wave] = inc;
        __syncthreads ();
        uint32_t at = ev_carry + inc - cnt;
        for (uint32_t w = 0; w < wave; w++) at += sh[16 + w];
        if (valid) {
            const uint32_t nth = j / 3;                        // digits of my run before this position's
            if (e_lit)   { ev_ctx[at] = (uint16_t)lit_ctx; ev_sym[at] = (uint8_t)L.symrank[b]; at++; }
            if (e_three) { const uint32_t k = nth - 1; ev_ctx[at] = (uint16_t)(256 + (k == 0 ? b : k == 1 ? 256u : 257u)); ev_sym[at] = 3; at++; }
            if (e_fin)   { const uint32_t k = nth;     ev_ctx[at] = (uint16_t)(256 + (k == 0 ? b : k == 1 ? 256u : 257u)); ev_sym[at] = (uint8_t)(j % 3); }
        }
        __syncthreads ();
        // carries for the next tile
        if (tid == 1023) { sh[32] = st; uint32_t tot = at + (e_fin ? 1u : 0u); sh[33] = tot; }
        __syncthreads ();
        start_carry = sh[32]; ev_carry = sh[33];
        (void)prev_lit_carry;
        __syncthreads ();
    }
    if (!tid) L.arith_n = ev_carry;
}

// A leaf with a wide alphabet (a binary plane: up to 256 byte values) usually still has contexts that are each followed
// by few distinct bytes (a plane of 11 000 bytes has ~43 occurrences per context). When the leaf is coded in one piece,
// such a context's model runs with one register plane and an alphabet of its own: the symbols that follow THIS context,
// found by a first pass over its occurrences (presence flags of the leaf ranks in LDS -> four 64-bit masks; local rank
// of a symbol = set bits below its leaf rank).
struct GzLocalAlpha { uint64_t m[4]; };

__device__ static inline uint32_t d_local_rank (const GzLocalAlpha &A, uint32_t s)
{
    const uint32_t j = s >> 6;
    const uint32_t c0 = (uint32_t)__popcll (A.m[0]), c1 = c0 + (uint32_t)__popcll (A.m[1]), c2 = c1 + (uint32_t)__popcll (A.m[2]);
    const uint32_t base = j == 0 ? 0u : (j == 1 ? c0 : (j == 2 ? c1 : c2));
    const uint64_t mj = j == 0 ? A.m[0] : (j == 1 ? A.m[1] : (j == 2 ? A.m[2] : A.m[3]));
    return base + (uint32_t)__popcll (mj & ((1ull << (s & 63)) - 1));
}

// Returns the number of distinct symbols among the context's occurrences [j0, j1) and, if there are at most 64, leaves
// their byte values (ascending) in lds_list[0..]. lds_flags: 256 bytes of LDS; one wave.
__device__ static inline uint32_t d_local_alphabet (GzLocalAlpha &A, const uint8_t *in, bool o1, const uint32_t *spos, const uint8_t *srk, const uint16_t *symrank,
                                                    const uint8_t *symlist, uint32_t j0, uint32_t j1, uint8_t *lds_flags, uint8_t *lds_list)
{
    const int lane = threadIdx.x & 63;
    ((uint32_t *)lds_flags)[lane] = 0;
    __syncthreads ();
    for (uint32_t j = j0 + lane; j < j1; j += 64) lds_flags[o1 ? (srk ? srk[j] : (spos[j] & 0xff)) : symrank[in[j]]] = 1;
    __syncthreads ();
    uint32_t nd = 0;
    #pragma unroll
    for (int k = 0; k < 4; k++) { A.m[k] = __ballot (lds_flags[k * 64 + lane] != 0); nd += (uint32_t)__popcll (A.m[k]); }
    if (nd <= 64) {
        #pragma unroll
        for (int k = 0; k < 4; k++) if ((A.m[k] >> lane) & 1) lds_list[d_local_rank (A, (uint32_t)(k * 64 + lane))] = symlist[k * 64 + lane];
    }
    __syncthreads ();
    return nd;
}

// The same for a leaf that spans position chunks, where a context's wave only ever sees one chunk's occurrences: which symbols follow
// which context byte anywhere in the leaf is found by one pass over the whole leaf before its first chunk (k_ctx_succ; a 256 x 256 bit
// matrix, 8 KB per leaf), so that a context keeps ONE alphabet - and with it one register layout in mstate - through all chunks.
// grid (listed leaves, ceil (longest leaf / 131072)), 256 threads, 8 KB of LDS (few workgroups: a leaf with a small alphabet leaves at once, and this launch sits in front of the first chunk of every long stream)
#define GZ_SUCC_SPAN 131072u
__global__ void __launch_bounds__(256) k_ctx_succ (GzdLeaf *leaves, const uint32_t *list)
{
    GzdLeaf &L = leaves[list[blockIdx.x]];
    if (!L.active || L.engine != GZ_ENG_ARITH || !L.succ || !L.o1 || L.rle || L.nsym <= 64) return;
    const uint32_t n = L.arith_n, p0 = blockIdx.y * GZ_SUCC_SPAN;
    if (p0 >= n) return;
    const uint32_t p1 = n - p0 > GZ_SUCC_SPAN ? p0 + GZ_SUCC_SPAN : n;
    uint32_t *m = (uint32_t *)gz_lds;                          // [256][8]
    const int tid = threadIdx.x;
    for (int i = tid; i < 2048; i += 256) m[i] = 0;
    __syncthreads ();
    const uint8_t *in = L.coded; const uint16_t *symrank = L.symrank;
    for (uint32_t pos = p0 + tid; pos < p1; pos += 256) {
        const uint32_t c = pos ? in[pos - 1] : 0u, s = symrank[in[pos]];
        atomicOr (&m[c * 8 + (s >> 5)], 1u << (s & 31));
    }
    __syncthreads ();
    uint32_t *g = (uint32_t *)L.succ;                          // (word s >> 6 of a row, bit s & 63: the same bytes as 32-bit words)
    for (int i = tid; i < 2048; i += 256) if (m[i]) atomicOr (&g[i], m[i]);
}

// (force-inlined: as a called function its arguments would arrive in vector registers and every loop on them would
//  become exec-mask code)
// The occurrences of this wave's context inside the position chunk are entries [j0, j1) of the leaf's sorted lists
// (order 1; srk = the symbol's rank in the leaf's alphabet), or simply positions [j0, j1) of the stream (order 0).
// J = 1: the model in registers (d_model_batch); J = 2, 4 (more than 64 symbols): the model in the LDS, the batches in rounds
// (d_model_batch_rounds) - the registers only carry it in and out (the state that travels between position chunks has one layout).
// -DGZ_MODEL_PHASES: where the waves of the hot contexts (>= GZ_MODEL_HOT occurrences in the launch, default 1 M) spend their time - lane 0's 100 MHz clock around
// the head of a batch (waits for the prefetched occurrences), the batch itself and its tail (record store, reciprocal fetch); sums over
// such waves, printed by gz_wait: 0 head, 1 register batches, 2 batches in rounds, 3 tail, 4 batches, 5 batches in rounds (count), 6 events, 7 waves
#ifdef GZ_MODEL_PHASES
#ifndef GZ_MODEL_HOT
#define GZ_MODEL_HOT 1000000u
#endif
__device__ unsigned long long g_mph[9];
#define MPH_T(k) do { const unsigned long long now_ = wall_clock64 (); mph_[k] += now_ - mt_; mt_ = now_; } while (0)
#else
#define MPH_T(k) do { } while (0)
#endif
#ifndef GZ_ROUNDS_MINJ
#define GZ_ROUNDS_MINJ 2                   // (-DGZ_ROUNDS_MINJ=1: the one-plane models in rounds as well)
#endif
template <int J, bool PK, bool O1, bool LA>
__device__ static __forceinline__ void d_arith_model_wave_ (const uint8_t *in, uint32_t ms, GzRec *recs,
                                                   const uint8_t *symlist, const uint16_t *symrank, uint32_t nsym,
                                                   const uint32_t *spos, const uint8_t *srk, uint32_t j0, uint32_t j1, bool first, bool save, uint32_t *st,
                                                   const GzLocalAlpha *la = nullptr)
{
    constexpr bool kRounds = J >= GZ_ROUNDS_MINJ;
    constexpr bool o1 = O1;                               // (sorted by context or not: compiled apart, like the list format - the fetch in front of a batch is the same loads every time)
    const int lane = threadIdx.x & 63;
    GzModel<J> M;
    uint32_t tot = ms;
    if (first) {
        #pragma unroll
        for (int j = 0; j < J; j++) {
            const uint32_t e = (uint32_t)(j * 64 + lane);
            const bool live = e < nsym;
            M.sym[j] = live ? symlist[e] : 0xffffffffu;
            const uint32_t prev_sym = (live && e) ? symlist[e - 1] : 0xffffffffu;
            M.gap[j] = live ? (e ? M.sym[j] - prev_sym - 1 : M.sym[j]) : 0;
            M.freq[j] = live ? 1 : 0;
            M.cum[j] = live ? M.sym[j] : ms;                  // present entries + absent entries before it == its byte value
            M.srank[j] = e; M.where[j] = e;
        }
    }
    else {                                                // resume where the previous position chunk stopped
        #pragma unroll
        for (int j = 0; j < J; j++) {
            uint32_t *s6 = st + (6 * j) * 64 + lane;
            M.sym[j] = s6[0]; M.srank[j] = s6[64]; M.freq[j] = s6[128]; M.cum[j] = s6[192]; M.gap[j] = s6[256]; M.where[j] = s6[320];
        }
        tot = d_uniform (st[6 * J * 64]);
    }
    const uint32_t n_absent = ms - nsym;
    if constexpr (kRounds) d_rounds_tables_in<J> (M, lane, nsym);
#ifdef GZ_MODEL_PHASES
    unsigned long long mph_[7] = { 0, 0, 0, 0, 0, 0, 0 }, mt_ = wall_clock64 (), mph_r_ = 0;
#endif

    // (a batch's records leave as soon as the batch is done: 12 bytes an occurrence, nothing to look up - until round 4 they carried the
    //  reciprocal of the total from a table, fetched while the NEXT batch was worked on)
    // The occurrences of the following batches are fetched while this one is being worked on. The raw loads (sorted position +
    // rank byte, or the input byte of an order-0 leaf) run FOUR TO EIGHT batches ahead: a batch without events is ~300 ns of work, a trip
    // to memory 1-2 us, so one batch ahead (as it was) left a context whose order is stable waiting for its next occurrences most of the
    // time (1.2 us per batch measured on a 25 M-occurrence genotype context). Two groups of four batches: A is being used up (picked by
    // a select - a register that a load still owes cannot even be MOVED without waiting for it, so no shifting ring), B is in flight
    // and becomes A once every four batches, when its loads are four batches old. What depends on a loaded value (the rank table of an
    // order-0 leaf, the context's own alphabet) is done one batch ahead, from A
    uint32_t nx_pos = 0, nx_rk = 0;
    uint32_t A_pos[4] = { 0, 0, 0, 0 }, A_raw[4] = { 0, 0, 0, 0 }, B_pos[4] = { 0, 0, 0, 0 }, B_raw[4] = { 0, 0, 0, 0 };
    uint32_t bi = 0, grp = j0;                            // bi: which batch of group A is being worked on; grp: position of A's first batch
    // (always loads - past the end the context's last occurrence again, which nobody looks at - and through GLOBAL pointers: a load
    //  under a condition leaves the compiler merging old and new value through a copy that has to wait for the load on the spot, and a
    //  load through a generic pointer is a FLAT one, for which it waits with vmcnt (0): measured inside the kernel, 0.5 of the 1.5 us of
    //  a quiet batch were that wait, once every four batches for the whole trip to memory)
    //  (Round 5: the same holds for a CHOICE between loads at run time - with "if (srk) two loads else one" inside the sorted case the hot
    //  context's wave of the VCF configuration, an UNSORTED one, took 10 % longer: 784 -> 864 ms of k_arith_model per step, found by
    //  bisecting; selected addresses or always-the-same-loads variants: 815 - 868. So which of the two list formats a batch has is a
    //  template parameter of the kernel, and the host keeps a batch to one format.)
    auto fetch_raw = [&] (uint32_t at, uint32_t &pos, uint32_t &raw) {
        const uint32_t a = at < j1 ? at : (j1 ? j1 - 1 : 0u);
        if (o1) {
            if constexpr (PK) { const uint32_t w = gz_ldg_u32 (spos + a); pos = w >> 8; raw = w & 0xff; }
            else { pos = gz_ldg_u32 (spos + a); raw = gz_ldg_u8 (srk + a); }
        }
        else    { pos = a; raw = gz_ldg_u8 (in + a); }
    };
    auto to_rank = [&] (uint32_t raw) -> uint32_t {
        uint32_t rk = o1 ? raw : gz_ldg_u16 (symrank + (raw & 0xff));
        if constexpr (LA) rk = d_local_rank (*la, rk);
        return rk;
    };
    #pragma unroll
    for (int k = 0; k < 4; k++) fetch_raw (j0 + 64 * k + lane, A_pos[k], A_raw[k]);
    #pragma unroll
    for (int k = 0; k < 4; k++) fetch_raw (j0 + 64 * (k + 4) + lane, B_pos[k], B_raw[k]);
    nx_pos = A_pos[0];
    if (j0 + lane < j1) nx_rk = to_rank (A_raw[0]);
#define GZ_WAVE_BATCH_HEAD \
        const uint32_t cnt = j1 - j < 64 ? j1 - j : 64; \
        const bool occ = (uint32_t)lane < cnt; \
        const uint32_t b_pos = nx_pos, b_rk = nx_rk; \
        if (++bi == 4) { \
            bi = 0; grp += 256; \
            _Pragma ("unroll") for (int k = 0; k < 4; k++) { A_pos[k] = B_pos[k]; A_raw[k] = B_raw[k]; } \
            _Pragma ("unroll") for (int k = 0; k < 4; k++) fetch_raw (grp + 64 * (k + 4) + lane, B_pos[k], B_raw[k]); \
        } \
        { \
            const uint32_t rp = bi == 0 ? A_pos[0] : bi == 1 ? A_pos[1] : bi == 2 ? A_pos[2] : A_pos[3]; \
            const uint32_t rr = bi == 0 ? A_raw[0] : bi == 1 ? A_raw[1] : bi == 2 ? A_raw[2] : A_raw[3]; \
            nx_pos = rp; nx_rk = (j + 64 + lane < j1) ? to_rank (rr) : 0u; \
        } \
        uint32_t out_cum = 0, out_freq = 0, out_tot = 0, n_ev = 0;
// (an EXPERIMENT of round 4 - the records stored in sorted (context-major) order, 64 consecutive ones per batch, to see what the
//  scattered stores cost: the streamed form's model launches 1.39 -> 1.18 ms. That is the most a coalescing scheme could win BEFORE
//  paying for the pass that puts the records into stream order - round 3's k_rec_unsort lost more than that - so they stay scattered.)
#define GZ_WAVE_BATCH_TAIL \
        if (occ) d_record_store (recs + b_pos, d_model_record (out_cum, out_freq, out_tot));
    for (uint32_t j = j0; j < j1; j += 64) {
        GZ_WAVE_BATCH_HEAD
        MPH_T (0);
        if constexpr (kRounds) { d_model_batch_rounds<J> (tot, lane, cnt, occ ? b_rk : 0u, nsym, n_absent, out_cum, out_freq, out_tot, n_ev); MPH_T (2); }
        else                   { d_model_batch<J> (M, tot, lane, cnt, occ ? b_rk : 0u, nsym, n_absent, out_cum, out_freq, out_tot, n_ev); MPH_T (1); }
        GZ_WAVE_BATCH_TAIL
        MPH_T (3);
#ifdef GZ_MODEL_PHASES
        mph_[4]++; mph_[5] += kRounds ? 1 : 0; mph_[6] += n_ev & 0xffff; mph_r_ += n_ev >> 16;
#endif
    }
#undef GZ_WAVE_BATCH_HEAD
#undef GZ_WAVE_BATCH_TAIL
#ifdef GZ_MODEL_PHASES
    if (!lane && j1 - j0 >= GZ_MODEL_HOT) { for (int k = 0; k < 7; k++) atomicAdd (&g_mph[k], mph_[k]); atomicAdd (&g_mph[7], 1ull); atomicAdd (&g_mph[8], mph_r_); }
#endif
    if (save) {
        if constexpr (kRounds) d_rounds_tables_out<J> (M, lane, nsym, symlist);
        #pragma unroll
        for (int j = 0; j < J; j++) {
            uint32_t *s6 = st + (6 * j) * 64 + lane;
            s6[0] = M.sym[j]; s6[64] = M.srank[j]; s6[128] = M.freq[j]; s6[192] = M.cum[j]; s6[256] = M.gap[j]; s6[320] = M.where[j];
        }
        if (!lane) st[6 * J * 64] = tot;
    }
}

// Position chunks: the model of positions [k*C, (k+1)*C) of every leaf is one launch; the chain follows chunk by
// chunk, so that only the first model chunk is not hidden behind the (longer, strictly serial) chain. The models'
// registers travel between launches through mstate.
#define GZ_MSTATE_WORDS 32                 // per lane and context: 6 per register plane (+ the total)
#define GZ_CHUNK_MIN    (64u * 1024u)      // positions; a multiple of GZ_CTX_TILE. Leaves up to this size are never split.

// The grids of these kernels run over a list of the plain arithmetic-coder leaves only (a VBlock's other leaves would
// otherwise cost a million workgroups per launch that exit at once - and the dispatcher, not the work, set the pace).
#define GZ_MODEL_GRID_Y 65                 // context 0 + one per present symbol of a leaf with up to 64 symbols

// grid: GZ_XCD_GRID over (listed leaves, GZ_MODEL_GRID_Y [+ GZ_MODEL_GRID_RUN for lists with run-length leaves])
#define GZ_MODEL_GRID_RUN 66               // run models: one per present symbol, 256 and 257
// (tried: a build of its own for the alphabets of up to 64 symbols - 42 instead of 87 vector registers, 8 instead of 5
//  waves per SIMD - launched beside one for the wide alphabets: no faster (4 M read pairs: 84.0 -> 86.8 ms per step).
//  With every SIMD holding several of these waves it is the issue slots, not the waves in flight, that run out.
//  Also tried: starting the busiest contexts of a chunk first (an order computed in k_ctx_scan) instead of in symbol
//  order - the launch is not waiting for its longest wave either: 84.0 -> 84.2 ms, 1 M pairs 33.95 -> 33.87.)
#ifdef GZ_MODEL_DEBUG
__device__ unsigned long long g_model_slowest;     // (10 ns ticks << 40) | (list index << 28) | (context << 18) | occurrences / 64
#define GZ_MODEL_T0 const unsigned long long t_dbg0 = wall_clock64 ()
#define GZ_MODEL_T1(ctx, occ) do { if (!(threadIdx.x & 63)) atomicMax (&g_model_slowest, ((wall_clock64 () - t_dbg0) << 40) | ((unsigned long long)((li & 0x7ff) | (chunk == 0xffffffffu ? 0x800 : 0)) << 28) | \
                                   ((unsigned long long)((ctx) & 0x3ff) << 18) | (unsigned long long)((occ) >> 6 > 0x3ffff ? 0x3ffff : (occ) >> 6)); } while (0)
#else
#define GZ_MODEL_T0 do {} while (0)
#define GZ_MODEL_T1(ctx, occ) do {} while (0)
#endif
template <int J, bool PK>                   // PK: the sorted lists of this batch carry position << 8 | rank (no srk array): k_ctx_scatter
__device__ static __forceinline__ void d_arith_model_wave (const uint8_t *in, uint32_t ms, bool o1, GzRec *recs,
                                                   const uint8_t *symlist, const uint16_t *symrank, uint32_t nsym,
                                                   const uint32_t *spos, const uint8_t *srk, uint32_t j0, uint32_t j1, bool first, bool save, uint32_t *st,
                                                   const GzLocalAlpha *la = nullptr)
{
    // (sorted by context or not, with a context's own alphabet or not: compiled apart - what the wave does in front of every batch is then
    //  the same instructions every time; with `o1` and `la` tested at run time the VCF step took 902 instead of 876 ms, the default step's
    //  model launches 27.8 instead of 24.8 ms)
    if (la) {
        if (o1) d_arith_model_wave_<J, PK, true, true>  (in, ms, recs, symlist, symrank, nsym, spos, srk, j0, j1, first, save, st, la);
        else    d_arith_model_wave_<J, PK, false, true> (in, ms, recs, symlist, symrank, nsym, spos, srk, j0, j1, first, save, st, la);
    }
    else {
        if (o1) d_arith_model_wave_<J, PK, true, false>  (in, ms, recs, symlist, symrank, nsym, spos, srk, j0, j1, first, save, st, la);
        else    d_arith_model_wave_<J, PK, false, false> (in, ms, recs, symlist, symrank, nsym, spos, srk, j0, j1, first, save, st, la);
    }
}

template <bool PK>
__global__ void __launch_bounds__(64) k_arith_model (GzdLeaf *leaves, const uint32_t *list, uint32_t n_list, uint32_t p0, uint32_t chunk, uint32_t row)
{
    GZ_XCD_GRID (li, by, n_list);
    GzdLeaf &L = leaves[list[li]];
    if (!L.active || L.engine != GZ_ENG_ARITH || L.arith_n <= p0 || d_leaf_tiled (L)) return;
    // (the blocks of the column that have no model to run - most of them, for a leaf of order 0 or with a small alphabet - leave before
    //  anything is made wave-uniform: the same tests as below)
    if (by >= GZ_MODEL_GRID_Y) { if (!L.rle) return; }
    else if (L.nsym <= 64 && by && (!L.o1 || by > L.nsym)) return;
    const uint32_t ms = L.max_sym;
    const bool o1 = L.o1, rle = L.rle;
    GzRec *tr = d_uniform_ptr ((GzRec *)L.triples);
    const uint8_t *coded = d_uniform_ptr (L.coded);
    const uint32_t n_u = d_uniform (L.arith_n), ms_u = d_uniform (ms), nsym_u = d_uniform (L.nsym), nctx = d_uniform (L.nctx);
    const bool o1_u = d_uniform (o1 ? 1u : 0u) != 0, rle_u = d_uniform (rle ? 1u : 0u) != 0;
    const bool sorted = o1_u || rle_u;                         // (the run-length variant always goes through the sorted lists)
    const uint32_t p1 = (n_u - p0 > chunk) ? p0 + chunk : n_u;
    const uint32_t *off = d_uniform_ptr (L.ctxoff), *spos = d_uniform_ptr (L.spos);
    const uint8_t *srk = d_uniform_ptr (L.srk);
    const uint32_t *cend = d_uniform_ptr (L.ctxend + (size_t)row * L.nctx);
    const uint32_t t0 = p0 / GZ_CTX_TILE;                      // (chunks are whole tiles)
    uint32_t *mstate = d_uniform_ptr (L.mstate);
    const uint64_t *succ = d_uniform_ptr (L.succ);

    // ---- the run models of the run-length variant: blocks GZ_MODEL_GRID_Y ... ; 4 symbols, all present from the start
    if (by >= GZ_MODEL_GRID_Y) {
        if (!rle_u) return;
        uint8_t *digits = gz_lds;                              // the alphabet { 0, 1, 2, 3 }
        if (threadIdx.x < 4) digits[threadIdx.x] = (uint8_t)threadIdx.x;
        __syncthreads ();
        for (uint32_t k = by - GZ_MODEL_GRID_Y; k < 258; k += GZ_MODEL_GRID_RUN) {
            if (k < 256 && L.symrank[k] == 0xffff) continue;   // a byte that never occurs has no runs
            const uint32_t ctx = 256 + k;
            const uint32_t j0 = d_uniform (off[(size_t)t0 * nctx + ctx]), j1 = d_uniform (cend[ctx]);
            if (j0 == j1 && p0) continue;
            d_arith_model_wave<1, PK> (coded, 4u, true, tr, digits, L.symrank, 4u, spos, srk, j0, j1, p0 == 0, p1 < n_u,
                                   mstate + (size_t)ctx * (GZ_MSTATE_WORDS * 64));
        }
        return;
    }
    // ---- the literal models
    if (nsym_u <= 64) {
        // block 0: context 0 (the context of position 0, whether byte 0 occurs or not); block y: the y-th present symbol
        uint32_t ctx = 0;
        if (by) {
            if (!o1_u || by > nsym_u) return;
            ctx = d_uniform (L.symlist[by - 1]);
            if (!ctx) return;                                  // (byte 0 is block 0's)
        }
        uint32_t *st = mstate + (size_t)ctx * (GZ_MSTATE_WORDS * 64);   // (only touched when the leaf spans chunks)
        uint32_t j0 = p0, j1 = p1;
        if (sorted) { j0 = d_uniform (off[(size_t)t0 * nctx + ctx]); j1 = d_uniform (cend[ctx]); }   // my run of the sorted lists
        GZ_MODEL_T0;
        d_arith_model_wave<1, PK> (coded, ms_u, sorted, tr, L.symlist, L.symrank, nsym_u, spos, srk, j0, j1, p0 == 0, p1 < n_u, st);
        GZ_MODEL_T1 (ctx, j1 - j0);
        return;
    }
    // wide alphabets: the contexts are dealt out over the blocks of the column
    for (uint32_t ctx = by; ctx < (o1_u ? ms_u : 1u); ctx += GZ_MODEL_GRID_Y) {
        if (ctx && L.symrank[ctx] == 0xffff) continue;         // a byte that never occurs is never a context
        uint32_t *st = mstate + (size_t)ctx * (GZ_MSTATE_WORDS * 64);
        uint32_t j0 = p0, j1 = p1;
        if (sorted) { j0 = d_uniform (off[(size_t)t0 * nctx + ctx]); j1 = d_uniform (cend[ctx]); }
        if (j0 == j1 && p0) continue;                           // (nothing of mine in this chunk: the saved state stands)
        GZ_MODEL_T0;
        if (p0 == 0 && p1 == n_u && j1 > j0) {                  // a leaf in one piece: try the context's own alphabet
            GzLocalAlpha la;
            uint8_t *lds_flags = gz_lds, *lds_list = gz_lds + 256;
            const uint32_t nd = d_local_alphabet (la, coded, sorted, spos, srk, L.symrank, L.symlist, j0, j1, lds_flags, lds_list);
            if (nd <= 64) {
                d_arith_model_wave<1, PK> (coded, ms_u, sorted, tr, lds_list, L.symrank, nd, spos, srk, j0, j1, true, false, st, &la);
                GZ_MODEL_T1 (ctx, j1 - j0);
                continue;
            }
            // (65 - 128 successors on two register planes instead of the leaf's four: measured slower here - binned-quality FASTQ 104.5 -> 109 ms
            //  per step - although it pays for leaves in position chunks, below)
        }
        else if (succ && o1_u && !rle_u) {                      // a leaf in position chunks: the context's alphabet over the whole leaf (k_ctx_succ)
            GzLocalAlpha la;
            uint32_t nd = 0;
            #pragma unroll
            for (int k = 0; k < 4; k++) { la.m[k] = d_uniform64 (succ[ctx * 4 + k]); nd += (uint32_t)__popcll (la.m[k]); }
            if (nd && nd <= 128) {
                uint8_t *lds_list = gz_lds + 256;
                __syncthreads ();                              // (the previous context of this block is done with the list)
                #pragma unroll
                for (int k = 0; k < 4; k++) if ((la.m[k] >> (threadIdx.x & 63)) & 1) lds_list[d_local_rank (la, (uint32_t)(k * 64 + (threadIdx.x & 63)))] = L.symlist[k * 64 + (threadIdx.x & 63)];
                __syncthreads ();
                if (nd <= 64) d_arith_model_wave<1, PK> (coded, ms_u, sorted, tr, lds_list, L.symrank, nd, spos, srk, j0, j1, p0 == 0, p1 < n_u, st, &la);
                else          d_arith_model_wave<2, PK> (coded, ms_u, sorted, tr, lds_list, L.symrank, nd, spos, srk, j0, j1, p0 == 0, p1 < n_u, st, &la);
                GZ_MODEL_T1 (ctx, j1 - j0);
                continue;
            }
        }
        if (nsym_u <= 128) d_arith_model_wave<2, PK> (coded, ms_u, sorted, tr, L.symlist, L.symrank, nsym_u, spos, srk, j0, j1, p0 == 0, p1 < n_u, st);
        else               d_arith_model_wave<4, PK> (coded, ms_u, sorted, tr, L.symlist, L.symrank, nsym_u, spos, srk, j0, j1, p0 == 0, p1 < n_u, st);
        GZ_MODEL_T1 (ctx, j1 - j0);
    }
}

// ---- the models of a small alphabet, a TILE of positions at a time (GZ_MODEL_TILED=1; OFF by default: exact, a third less traffic, slower) ----
// One wave per (leaf, context) stores the records of ITS occurrences: 12 bytes here, 12 bytes there, between the records of the other
// contexts' waves, which pass the same lines hundreds of microseconds earlier or later - every such store leaves the L2 as a 32-byte
// sector of its own (profiles/r05_pmc.json: 10.7 GB of WRITE_SIZE for 308 M records of 12 bytes), and the sort that groups the positions
// by context in front of it moves another 3.3 GB. This kernel is the other way to cut the loop: for the leaves that carry nearly all
// symbols of a FASTQ / BAM file - order 1, at most 64 distinct bytes: quality scores - ONE workgroup takes all contexts of a leaf through
// the position chunk tile by tile, and nothing but the tile's input bytes and its finished records crosses the LDS boundary:
//   1. the tile's bytes -> LDS; a stable counting sort by context inside the LDS (a wave per stretch of the tile, 64 positions at a
//      time: rank among the lanes with the same context through 64-bit LDS masks, like d_batch_counts_lds; counts per wave and context,
//      a scan, the scatter): per context the tile positions of its occurrences, in stream order;
//   2. the contexts, busiest first, are dealt out to the waves back and forth: a wave takes a model's registers from the LDS (2 words
//      per lane and context), runs the context's occurrences through d_model_batch 64 at a time as k_arith_model does, leaves (cum, freq,
//      tot) of each in the tile's slot of its position, and puts the registers back;
//   3. the tile's records leave in stream order, coalesced: the lane that stores a record makes it (d_model_record: the reciprocal).
// The models' state stays in the LDS from tile to tile and travels through mstate from one position chunk's launch to the next.
// MEASURED (round 5, profiles/r05b_tiled_*.txt; every output byte equal, tests/test_gpu.py::test_tiled_models_exact):
//   * HBM traffic of the default step 39.2 -> 27.4 GB (k_arith_model 12.0 + k_ctx_scatter 3.3 -> 4.0: 12 bytes written and 1 read per symbol);
//   * but ONE workgroup per leaf is 4 - 16 waves where the contexts' own waves are 34 spread over the device: with 4 waves and tiles of
//     4096 a leaf's models advance at 12.7 ns per position (clocks per tile, alone on the device: load 9 600, sort 7 100, scan + scatter
//     8 900, models 69 000 = 19 batches a wave at 3 600 - a context's few occurrences per tile make part-empty batches and every batch
//     waits for its list entries -, waiting for the slowest wave 11 600, records out 3 200), and the range coder's chain wants a position
//     every 6.3: default step 42.9 -> 76.6 ms, streamed 139 -> 226 ms. With 16 waves and tiles of 8192 (130 KB of LDS and every vector
//     register of a compute unit: one workgroup per compute unit) a workgroup only ever starts on a compute unit that is completely
//     empty, and beside the other streams' small workgroups that is rare: 358 ms per default step, 203 ms streamed.
//   So the records stay scattered (k_arith_model), and this kernel stays in the library as the measured alternative.
// (A first version dealt the contexts out through an LDS counter bumped by lane 0 of the free wave; the compiler built that loop around
//  exec masks and the kernel never came back from the device - with GZ_TM_DEBUG's early exits compiled in it did: found by bisecting
//  builds on the device. Inside the loop over contexts every statement is now executed by every lane.)
// k_ctx_count / _scan / _scatter and k_arith_model leave such leaves alone when the handle asks for this kernel (d_leaf_tiled).
#ifdef GZ_TM_NODBG
#define GZ_TM_STOP(k) do { } while (0)
#define GZ_TM_STOP_BREAK(k) do { } while (0)
#else
#define GZ_TM_STOP(k) do { if (dbg == (k)) return; } while (0)             // (GZ_TM_DEBUG: the kernel leaves after stage k - wrong results, for bisecting on the device)
#define GZ_TM_STOP_BREAK(k) if (dbg == (k)) break
#endif
#ifdef GZ_TM_PROFILE                    // (probe builds: where a workgroup's clocks go, printed by workgroup 0 of a launch with >= 8 tiles)
#define GZ_TM_T(k) do { const unsigned long long now_ = clock64 (); tmp_[k] += now_ - tmt_; tmt_ = now_; } while (0)
#else
#define GZ_TM_T(k) do { } while (0)
#endif
#ifndef GZ_TM_TILE
#define GZ_TM_TILE   4096u
#endif
#ifndef GZ_TM_WAVES
#define GZ_TM_WAVES  4
#endif
#define GZ_TM_SLOTS  65                     // context 0 + one per present symbol
#define GZ_TM_PER    (GZ_TM_TILE / (64 * GZ_TM_WAVES))        // rounds of 64 positions per wave and tile
#define GZ_TM_REC_CF 0                                          // uint32_t [TILE]: cum | freq << 16 of the position's symbol
#define GZ_TM_REC_T  (GZ_TM_REC_CF + GZ_TM_TILE * 4)            // uint16_t [TILE]: the total it was coded with
#define GZ_TM_IN     (GZ_TM_REC_T + GZ_TM_TILE * 2)             // uint8_t  [16 + TILE]: byte 15 + i = the stream's byte t0 + i - 1 (the context of position t0 + i)
#define GZ_TM_LIST   (GZ_TM_IN + GZ_TM_TILE + 32)               // uint16_t [TILE]: tile positions grouped by context
#define GZ_TM_STATE  (GZ_TM_LIST + GZ_TM_TILE * 2)              // uint32_t [SLOTS][2][64]: freq | cum << 16, sym | srank << 8 | where << 16 | gap << 24
#define GZ_TM_TOT    (GZ_TM_STATE + GZ_TM_SLOTS * 512)          // uint32_t [SLOTS + 3] the models' totals
#define GZ_TM_CNTW   (GZ_TM_TOT + 272)                          // uint32_t [WAVES][SLOTS + 3]: occurrences per wave and context, then their exclusive sum over the waves
#define GZ_TM_TOTAL  (GZ_TM_CNTW + GZ_TM_WAVES * 272)           // uint32_t [SLOTS + 3] per context in the tile
#define GZ_TM_START  (GZ_TM_TOTAL + 272)                        // uint32_t [SLOTS + 3] where its list starts
#define GZ_TM_ORDER  (GZ_TM_START + 272)                        // uint8_t  [80] contexts, busiest first
#define GZ_TM_MISC   (GZ_TM_ORDER + 80)                         // uint32_t [4]: the queue's counter, contexts with occurrences
#define GZ_TM_SLOTOF (GZ_TM_MISC + 16)                          // uint8_t  [256] byte -> context slot
#define GZ_TM_RANKOF (GZ_TM_SLOTOF + 256)                       // uint8_t  [256] byte -> rank in the leaf's alphabet
#define GZ_TM_SCR    (GZ_TM_RANKOF + 256)                       // per wave: uint64_t [66] masks, [64] d_batch_counts_lds' s_low
#define GZ_TM_SCR_BYTES 1088
#define GZ_TM_LDS    (GZ_TM_SCR + GZ_TM_WAVES * GZ_TM_SCR_BYTES)
// grid (listed leaves), 1024 threads, GZ_TM_LDS bytes of LDS; positions [p0, p0 + chunk) of every listed leaf that d_leaf_tiled
__global__ void __launch_bounds__(64 * GZ_TM_WAVES) k_arith_model_tiled (GzdLeaf *leaves, const uint32_t *list, uint32_t p0, uint32_t chunk, uint32_t dbg)
{
    GzdLeaf &L = leaves[list[blockIdx.x]];
    if (!d_leaf_tiled (L) || L.arith_n <= p0) return;
    const uint32_t tid = threadIdx.x, wave = d_uniform (tid >> 6);
    const int lane = (int)(tid & 63);
    const uint32_t n = d_uniform (L.arith_n), ms = d_uniform (L.max_sym), nsym = d_uniform (L.nsym), n_absent = ms - nsym;
    const uint32_t p1 = (n - p0 > chunk) ? p0 + chunk : n;
    const uint8_t *in = d_uniform_ptr (L.coded);
    GzRec *recs = d_uniform_ptr ((GzRec *)L.triples);
    uint32_t *mstate = d_uniform_ptr (L.mstate);
    uint32_t *rec_cf = (uint32_t *)(gz_lds + GZ_TM_REC_CF);
    uint16_t *rec_t = (uint16_t *)(gz_lds + GZ_TM_REC_T), *lst = (uint16_t *)(gz_lds + GZ_TM_LIST);
    uint8_t *tin = gz_lds + GZ_TM_IN + 15, *order = gz_lds + GZ_TM_ORDER, *slot_of = gz_lds + GZ_TM_SLOTOF, *rank_of = gz_lds + GZ_TM_RANKOF;
    uint32_t *state = (uint32_t *)(gz_lds + GZ_TM_STATE), *s_tot = (uint32_t *)(gz_lds + GZ_TM_TOT), *cntw = (uint32_t *)(gz_lds + GZ_TM_CNTW);
    uint32_t *total = (uint32_t *)(gz_lds + GZ_TM_TOTAL), *start = (uint32_t *)(gz_lds + GZ_TM_START), *misc = (uint32_t *)(gz_lds + GZ_TM_MISC);
    unsigned long long *scr = (unsigned long long *)(gz_lds + GZ_TM_SCR + wave * GZ_TM_SCR_BYTES);
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0;

    // ---- once per launch: the byte -> context slot / rank tables, the models' state (fresh, or as the previous chunk's launch left it)
    for (uint32_t b = tid; b < 256; b += 64 * GZ_TM_WAVES) {
        const uint32_t rk = L.symrank[b];
        rank_of[b] = (uint8_t)rk;
        slot_of[b] = (uint8_t)((b && rk != 0xffffu) ? rk + 1 : 0u);              // (byte 0 is slot 0's whether it occurs or not; a byte that never occurs is never a context)
    }
    for (uint32_t c = wave; c < GZ_TM_SLOTS; c += GZ_TM_WAVES) {
        uint32_t x, y, t;
        if (p0 == 0) {                                                  // d_arith_model_wave_'s `first`: every symbol once, the absent entries as gaps
            const uint32_t e = (uint32_t)lane;
            const bool live = e < nsym;
            const uint32_t sym = live ? L.symlist[e] : 0xffu, prev = (live && e) ? L.symlist[e - 1] : 0u;
            const uint32_t gap = live ? (e ? sym - prev - 1 : sym) : 0u;
            x = live ? (1u | (sym << 16)) : (ms << 16);
            y = sym | (e << 8) | (e << 16) | (gap << 24);
            t = ms;
        }
        else {
            const uint32_t *g = mstate + (size_t)c * (GZ_MSTATE_WORDS * 64);
            x = g[lane]; y = g[64 + lane]; t = d_uniform (g[128]);
        }
        state[c * 128 + lane] = x; state[c * 128 + 64 + lane] = y;
        s_tot[c] = t;
    }
    __syncthreads ();
    GZ_TM_STOP (1);

#ifdef GZ_TM_PROFILE
    unsigned long long tmp_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tmt_ = clock64 (), tm_batches = 0, tm_ctx = 0;
#endif
    for (uint32_t t0 = p0; t0 < p1; t0 += GZ_TM_TILE) {
        const uint32_t nt = p1 - t0 < GZ_TM_TILE ? p1 - t0 : GZ_TM_TILE;
        GZ_TM_T (7);
        // ---- 1. the tile's bytes; per wave: no occurrences yet, no mask set
        #pragma unroll
        for (uint32_t r = 0; r < GZ_TM_TILE / (64 * GZ_TM_WAVES); r++) {
            const uint32_t i = r * (64 * GZ_TM_WAVES) + tid;
            if (i < nt) tin[1 + i] = (uint8_t)gz_ldg_u8 (in + t0 + i);
        }
        if (!tid) tin[0] = t0 ? (uint8_t)gz_ldg_u8 (in + t0 - 1) : 0;
        for (uint32_t c = (uint32_t)lane; c < GZ_TM_SLOTS + 3; c += 64) cntw[wave * (GZ_TM_SLOTS + 3) + c] = 0;
        scr[lane] = 0; if (lane < 2) scr[64 + lane] = 0;
        __syncthreads ();
        GZ_TM_T (0);
        // ---- the sort: this wave's 512 positions, 64 at a time - where each one goes inside its (wave, context) run
        uint32_t co[GZ_TM_PER];                                         // context slot | place inside the wave's run of it << 8
        #pragma unroll
        for (uint32_t r = 0; r < GZ_TM_PER; r++) {
            const uint32_t i = wave * (64 * GZ_TM_PER) + r * 64 + (uint32_t)lane;
            const bool valid = i < nt;
            const uint32_t c = valid ? slot_of[tin[i]] : GZ_TM_SLOTS;            // (the byte before position t0 + i; beyond the tile: a slot of nobody's)
            if (valid) atomicOr (&scr[c], 1ull << lane);
            gz_wave_sync ();
            const unsigned long long mm = scr[c];
            const uint32_t old = cntw[wave * (GZ_TM_SLOTS + 3) + c];
            gz_wave_sync ();
            if (valid && (mm >> lane) == 1ull) { cntw[wave * (GZ_TM_SLOTS + 3) + c] = old + (uint32_t)__popcll (mm); scr[c] = 0; }   // (the context's last lane of this round)
            gz_wave_sync ();
            co[r] = c | ((old + (uint32_t)__popcll (mm & below)) << 8);
        }
        __syncthreads ();
        GZ_TM_T (1);
        GZ_TM_STOP (2);
        if (tid < GZ_TM_SLOTS) {                                        // per context: the waves' counts -> where each wave's run starts inside the context's list
            uint32_t run = 0;
            for (uint32_t w = 0; w < GZ_TM_WAVES; w++) { const uint32_t v = cntw[w * (GZ_TM_SLOTS + 3) + tid]; cntw[w * (GZ_TM_SLOTS + 3) + tid] = run; run += v; }
            total[tid] = run;
        }
        __syncthreads ();
        if (!wave) {                                                    // where the contexts' lists start; the contexts by occurrences, most first
            const uint32_t v = total[lane], v64 = total[64];
            const uint32_t inc = d_wave_incl_scan (v, lane);
            start[lane] = inc - v;
            if (lane == 63) start[64] = inc;
            uint32_t rank = 0, rank64 = 0;
            for (uint32_t k = 0; k < GZ_TM_SLOTS; k++) {
                const uint32_t tk = total[k];
                rank   += (tk > v   || (tk == v   && k < (uint32_t)lane)) ? 1u : 0u;
                rank64 += (tk > v64 || (tk == v64 && k < 64u)) ? 1u : 0u;
            }
            order[rank] = (uint8_t)lane;
            if (!lane) order[rank64] = 64;
            const uint64_t live = __ballot (v != 0);
            if (!lane) misc[1] = (uint32_t)__popcll (live) + (v64 ? 1u : 0u);
        }
        __syncthreads ();
        GZ_TM_STOP (3);
        #pragma unroll
        for (uint32_t r = 0; r < GZ_TM_PER; r++) {
            const uint32_t i = wave * (64 * GZ_TM_PER) + r * 64 + (uint32_t)lane;
            const uint32_t c = co[r] & 0xffu;
            if (i < nt) lst[start[c] + cntw[wave * (GZ_TM_SLOTS + 3) + c] + (co[r] >> 8)] = (uint16_t)i;
        }
        __syncthreads ();
        GZ_TM_STOP (4);
        GZ_TM_T (2);
        // ---- 2. the models: the contexts, busiest first, are dealt out to the waves back and forth (wave 0 .. 15, 15 .. 0, 0 .. 15 ...): the
        // busiest sixteen start at once and the small ones fill up behind them. (Everything in this loop is wave-uniform and every lane takes
        // part in every statement - no "lane 0 only" inside it: a counter in the LDS that the free wave's lane 0 bumped made the compiler build
        // the loop around exec masks, and that build never came back from the device.)
        const uint32_t n_live = d_uniform (misc[1]);
        for (uint32_t turn = 0; turn * GZ_TM_WAVES < n_live; turn++) {
            const uint32_t k = turn * GZ_TM_WAVES + ((turn & 1) ? GZ_TM_WAVES - 1 - wave : wave);
            if (k >= n_live) break;
            const uint32_t c = d_uniform (order[k]), nc = d_uniform (total[c]), s0 = d_uniform (start[c]);
            GzModel<1> M;
            { const uint32_t x = state[c * 128 + lane], y = state[c * 128 + 64 + lane];
              M.freq[0] = x & 0xffffu; M.cum[0] = x >> 16; M.sym[0] = y & 0xffu; M.srank[0] = (y >> 8) & 0xffu; M.where[0] = (y >> 16) & 0xffu; M.gap[0] = y >> 24; }
            uint32_t tot = d_uniform (s_tot[c]);
            for (uint32_t j = 0; j < nc; j += 64) {
                const uint32_t cnt = nc - j < 64 ? nc - j : 64;
                const bool occ = (uint32_t)lane < cnt;
                const uint32_t pos = lst[s0 + j + (occ ? (uint32_t)lane : 0u)];
                const uint32_t rk = occ ? rank_of[tin[1 + pos]] : 0u;
                uint32_t out_cum = 0, out_freq = 0, out_tot = 0, n_ev = 0;
                d_model_batch<1> (M, tot, lane, cnt, rk, nsym, n_absent, out_cum, out_freq, out_tot, n_ev, scr, scr + 66);
                if (occ) { rec_cf[pos] = out_cum | (out_freq << 16); rec_t[pos] = (uint16_t)out_tot; }
            }
            state[c * 128 + lane] = (M.freq[0] & 0xffffu) | (M.cum[0] << 16);
            state[c * 128 + 64 + lane] = (M.sym[0] & 0xffu) | ((M.srank[0] & 0xffu) << 8) | ((M.where[0] & 0xffu) << 16) | (M.gap[0] << 24);
            s_tot[c] = d_uniform (tot);                                 // (every lane the same word)
#ifdef GZ_TM_PROFILE
            tm_batches += (nc + 63) / 64; tm_ctx++;
#endif
            GZ_TM_STOP_BREAK (5);
        }
        GZ_TM_T (3);
        __syncthreads ();
        GZ_TM_T (4);
        GZ_TM_STOP (6);
        // ---- 3. the tile's records, in stream order
        #pragma unroll
        for (uint32_t r = 0; r < GZ_TM_TILE / (64 * GZ_TM_WAVES); r++) {
            const uint32_t i = r * (64 * GZ_TM_WAVES) + tid;
            if (i < nt) { const uint32_t cf = rec_cf[i]; d_record_store (recs + t0 + i, d_model_record (cf & 0xffffu, cf >> 16, rec_t[i])); }
        }
        __syncthreads ();
        GZ_TM_T (5);
    }
#ifdef GZ_TM_PROFILE
    if (!blockIdx.x && !lane && p1 - p0 >= 8 * GZ_TM_TILE)
        printf ("[tm] wave %u: %u tiles; clocks per tile: load %llu  sort rounds %llu  scan + scatter %llu  models %llu (%llu contexts, %llu batches per tile)  wait for the others %llu  records out %llu  loop %llu\n", wave,
                (p1 - p0) / GZ_TM_TILE, tmp_[0] / ((p1 - p0) / GZ_TM_TILE), tmp_[1] / ((p1 - p0) / GZ_TM_TILE), tmp_[2] / ((p1 - p0) / GZ_TM_TILE), tmp_[3] / ((p1 - p0) / GZ_TM_TILE),
                tm_ctx / ((p1 - p0) / GZ_TM_TILE), tm_batches / ((p1 - p0) / GZ_TM_TILE), tmp_[4] / ((p1 - p0) / GZ_TM_TILE), tmp_[5] / ((p1 - p0) / GZ_TM_TILE), tmp_[7] / ((p1 - p0) / GZ_TM_TILE));
#endif
    if (p1 < n)                                                         // the next position chunk's launch goes on from here
        for (uint32_t c = wave; c < GZ_TM_SLOTS; c += GZ_TM_WAVES) {
            uint32_t *g = mstate + (size_t)c * (GZ_MSTATE_WORDS * 64);
            g[lane] = state[c * 128 + lane]; g[64 + lane] = state[c * 128 + 64 + lane];
            if (!lane) g[128] = s_tot[c];
        }
}

// ---- range coder chain ------------------------------------------------------------------------------------------
// Measured on MI355X (tools/ubench_issue.hip, tools/ubench_chain_f64.hip): ONE wave issues an instruction every 4.25 clocks
// (5.25 for the 8-byte encodings) whether or not it depends on the previous one, a vector->scalar hand-over costs ~12 ns, scalar
// loads return out of order (only "wait for all" exists), a vector load in the loop costs 9+ clocks - nothing but the instruction
// count of the serial wave and the waits in it matter. What is truly serial in the range coder (c_range_coder.h:97-109) is only
//        r = range / tot ;  range = (r * freq) << 8k      (k = bytes needed to bring range back above 2^24)
// `low` is not: low += cum * r followed by shifts is a big-number addition, and addition is associative. So
//   k_arith_chain  one wave per leaf. Rounds 1-2 ran the recurrence on the scalar unit, seven integer instructions per symbol
//                  (+ inc, mulhi by a magic number, >> shift, * freq, clz, & 0x18, <<) fed by scalar loads: 30.5 clocks per
//                  symbol + the waits for those loads (13.6 ns alone on the device, 14.9 beside the other kernels of a step).
//                  Now THREE vector instructions in double precision (the state R is range * 2^-7 as a double):
//                      T = fma (R, 2^-45 / tot, 1.0)     rounding toward zero: 1 + floor (range / tot) * 2^-52 - the low word IS r
//                      R = fma (T, F, -F)                F = freq * 2^45: r * freq * 2^-7, exact
//                      R.hi = R.hi & 0x7fffff | 0x41000000     the exponent's low three bits stay, the others become those of
//                                                        [2^24, 2^32): exactly "shift left by whole bytes until >= 2^24"
//                  and NO operand fetch in the loop: lane j holds the operands of symbols base + 16 j .. + 15 (12-byte records
//                  loaded coalesced a block of 1024 symbols ahead, in the wait states of the lane hops), all lanes
//                  execute every step, the state hops to the next lane through a DPP read - inside a row of 16 lanes the second
//                  fma itself (v_fmac_f64_dpp row_newbcast) - after the 16 symbols a lane holds, because a DPP read of a fresh
//                  result costs two wait states (gz_chain_asm.h, written by tools/gen_chain_asm.py: one copy of a block's code, every
//                  8-byte instruction on an 8-byte boundary): 13.0 clocks = 5.4 ns per symbol, the same with 1 or 64 chains on the
//                  device (round 5: 15.1, round 4: 15.8). It never looks at cum or low, and stores only the state
//                  before every 64th symbol: k_chain_expand recomputes r = range / tot of every symbol from those for the low
//                  kernels - with the plain formulation of the same arithmetic (an integer multiply), and checks that it arrives
//                  at the chain's next checkpoint: the two check each other on every 64 symbols of every stream.
//   k_low_*        all threads: every thread replays low += cum * r for its own slice of 64 symbols from low = 0,
//                  emitting the byte that leaves the 32-bit window at every shift (plus the carry out of the window as
//                  a 9th bit) at its absolute output position (k_chain_expand / k_low_scan: a prefix sum of the k's;
//                  k_low_scatter), k_low_resid adds what is left in each window where the following slices' bytes
//                  go, and k_low_norm normalises the digits: carries ripple left inside a tile and, very rarely, across.
//                  The first three follow the chain chunk by chunk (k_low_gate), the last two run once at the end.
// This reproduces RC_ShiftLow's cache / pending-0xFF bookkeeping (c_range_coder.h:70-88) exactly: that logic is just
// a lazy form of the same addition ("[0, T1, T2, ...] plus 1 at the byte before every shift that saw a carry").
typedef uint32_t gz_u32x4 __attribute__((vector_size (16)));

#define GZ_CHAIN_R0_LO 0xffe00000u        // the coder's first range, 2^32 - 1, as the double (2^32 - 1) * 2^-7
#define GZ_CHAIN_R0_HI 0x417fffffu

// One symbol - the plain formulation (k_chain_expand; in the chain: the rest of a leaf that does not fill a block of the loop). The wave must have called gz_f64_round_toward_zero. Returns r = range / tot.
__device__ static inline uint32_t d_chain_step (uint32_t &rlo, uint32_t &rhi, uint32_t inv_lo, uint32_t inv_hi, uint32_t freq, uint32_t *shift_bytes = NULL)
{
    const double t = gz_fma_rtz (__hiloint2double ((int)rhi, (int)rlo), __hiloint2double ((int)inv_hi, (int)inv_lo), 1.0);
    const uint32_t r = (uint32_t)__double2loint (t);             // 1 + r * 2^-52: the integer sits in the low word
    const uint32_t rf = r * freq;                                // <= range < 2^32 (>= 256: r >= 2^24 / 65535)
    const double x = (double)rf * 0.0078125;                     // * 2^-7: exact
    rlo = (uint32_t)__double2loint (x);
    rhi = ((uint32_t)__double2hiint (x) & 0x007fffffu) | 0x41000000u;    // 0, 1 or 2 bytes up
    if (shift_bytes) *shift_bytes = (uint32_t)__clz (rf) >> 3;  // how many
    return r;
}

// positions [p0, p0 + chunk) (p0 and chunk are multiples of 256)
// Four leaves per workgroup, one per wave (= one per SIMD). For the long leaves the kernel is PERSISTENT: it is launched
// at the start of the step, asks for (nearly) the whole LDS of a compute unit although it uses none - so that no
// workgroup of another kernel that needs LDS fits beside it and the chain waves, the critical path of the whole step,
// share instruction fetch and issue with nobody - and then follows the model kernels chunk by chunk: `progress` is the
// number of position chunks whose records are complete (written by k_arith_progress, which the host queues behind every
// model launch). Records are only ever read after their chunk was announced, and never before by this kernel (the
// block requested ahead stays inside the announced chunks), so no stale copy of them can sit in a cache on the way.
#ifndef GZ_CHAIN_WAVES                   // (-DGZ_CHAIN_WAVES=1 / 2: a leaf or two per workgroup - measured, round 5: 42.7 - 42.9 ms per default step with 1, 2 or 4)
#define GZ_CHAIN_WAVES 4
#endif
#define GZ_CHAIN_LDS   (156 * 1024)

__global__ void k_arith_progress (uint32_t *progress, uint32_t chunks_done) { *progress = chunks_done; }

// (bounded: if the models never report - a failed launch - the chain gives up after a few seconds instead of hanging
//  the device; the leaf is then flagged and its stream fails)
__device__ static inline bool d_wait_progress (const uint32_t *progress, uint32_t want)
{
    for (uint32_t spins = 0; __atomic_load_n (progress, __ATOMIC_RELAXED) < want; spins++) {
        if (spins > 8000000u) return false;
        __builtin_amdgcn_s_sleep (16);
    }
    __atomic_thread_fence (__ATOMIC_ACQUIRE);
    gz_scalar_cache_inv ();
    return true;
}

// positions [i0, i1) one symbol at a time in the plain formulation (the end of a leaf that does not fill a block of the loop; i0 a
// multiple of 64): 64 records per trip to memory, fetched by the lanes and handed out by readlane; the state before every 64th
// symbol goes out as in the loop
__device__ static inline void d_chain_slow (uint32_t &rlo, uint32_t &rhi, int lane, uint32_t i0, uint32_t i1, const uint8_t *triples, uint32_t *ck)
{
    for (uint32_t g = i0; g < i1; g += 64) {
        gz_scalar_store2 (ck + 2 * (g >> 6), rlo, rhi);
        const GzRec mine = ((const GzRec *)triples)[g + lane];  // (beyond i1: inside the padded area, not looked at)
        const uint32_t fq = g + lane < i1 ? d_record_freq (mine.f) : 1u;
        const int m = i1 - g < 64 ? (int)(i1 - g) : 64;
        for (int j = 0; j < m; j++) (void)d_chain_step (rlo, rhi, d_readlane (mine.lo, j), d_readlane (mine.hi, j), d_readlane (fq, j));
    }
}

// positions [p0, p1) of one leaf (p0 a multiple of 64; p1 - p0 one of GZ_CHAIN_BLOCK unless p1 is the leaf's end)
// What leaves the chain is the state BEFORE every 64th symbol (8 bytes at ck + 2 * (i / 64)) and the state after the last symbol
// of the call (the next call's first checkpoint, or the leaf's closing one): one scalar store per 64 symbols.
__device__ static __forceinline__ void d_chain_chunk (uint32_t &rlo, uint32_t &rhi, int lane, uint32_t p0, uint32_t p1, const uint8_t *triples, uint32_t *ck)
{
    const uint32_t whole = p0 + (p1 - p0) / GZ_CHAIN_BLOCK * GZ_CHAIN_BLOCK;
    uint32_t i = p0;
    if (whole > p0) { gz_chain_blocks (rlo, rhi, triples + (size_t)p0 * GZ_CHAIN_REC, (whole - p0) / GZ_CHAIN_BLOCK, ck + 2 * (p0 >> 6)); i = whole; }
    if (i < p1) d_chain_slow (rlo, rhi, lane, i, p1, triples, ck);     // the end of the leaf
    gz_scalar_store2 (ck + 2 * ((p1 + 63) >> 6), rlo, rhi);
}

// progress == NULL: everything is there already, one piece (bounds is ignored). bounds [0 .. n_chunks]: where the position chunks start (the
// first ones are shorter than the rest: the first chunk's sort + models are the lead-in of the long streams - gz_host.cpp, arith_pipe_setup)
__device__ static __forceinline__ void d_arith_chain (GzdLeaf *leaves, const uint32_t *list, uint32_t n_list, const uint32_t *progress, const uint32_t *bounds,
                                                      uint32_t *fail, uint32_t *done, uint32_t n_chunks)
{
    const uint32_t li = blockIdx.x * GZ_CHAIN_WAVES + (threadIdx.x >> 6);
    if (li >= n_list) return;
    __builtin_amdgcn_s_setprio (3);
    gz_f64_round_toward_zero ();
    const int lane = threadIdx.x & 63;
    if (progress && !d_wait_progress (progress, 1)) { if (!lane) *fail = 1; return; }   // (the leaf table itself is only final once the models have started)
    GzdLeaf &L = leaves[list[li]];
    if (!L.active || L.engine != GZ_ENG_ARITH || !L.arith_n) {
        if (done && !lane) for (uint32_t k = 0; k < n_chunks; k++) atomicAdd (&done[k], 1u);   // (the low kernels count leaves per chunk)
        return;
    }
    const uint32_t n = d_uniform (L.arith_n);
    const uint8_t *triples = d_uniform_ptr (L.triples);        // (wave-uniform: keep them in scalar registers)
    uint32_t *ck = d_uniform_ptr ((uint32_t *)L.ckpt);          // (checkpoints: the state before every 64th symbol, 8 bytes each)
    uint32_t rlo = GZ_CHAIN_R0_LO, rhi = GZ_CHAIN_R0_HI;
    if (!progress) d_chain_chunk (rlo, rhi, lane, 0, n, triples, ck);
    else
        for (uint32_t k = 0; k < n_chunks; k++) {
            const uint32_t p0 = d_uniform (bounds[k]), pe = d_uniform (bounds[k + 1]);
            if (p0 >= n) break;
            if (k && !d_wait_progress (progress, k + 1)) { if (!lane) { L.overflow = 2; *fail = 1; } break; }
            d_chain_chunk (rlo, rhi, lane, p0, pe < n ? pe : n, triples, ck);
            if (done) {                                        // this leaf's checkpoints of chunk k are final: tell the low kernels
                gz_scalar_store_flush ();
                __threadfence ();
                if (!lane) {
                    atomicAdd (&done[k], 1u);
                    if (n <= pe) for (uint32_t k2 = k + 1; k2 < n_chunks; k2++) atomicAdd (&done[k2], 1u);   // (a short leaf has no later chunks)
                }
            }
        }
    gz_scalar_store_flush ();
}

__global__ void __launch_bounds__(64 * GZ_CHAIN_WAVES) k_arith_chain (GzdLeaf *leaves, const uint32_t *list, uint32_t n_list, const uint32_t *progress,
                                                                      const uint32_t *bounds, uint32_t *fail, uint32_t *done, uint32_t n_chunks)
{
    d_arith_chain (leaves, list, n_list, progress, bounds, fail, done, n_chunks);
}

// One thread: holds its stream until all `want` leaves of the persistent chain have finished a position chunk (the low
// kernels of that chunk are queued behind it). Bounded like d_wait_progress.
__global__ void k_low_gate (const uint32_t *done, uint32_t want, uint32_t *fail)
{
    for (uint32_t spins = 0; __atomic_load_n (done, __ATOMIC_RELAXED) < want; spins++) {
        if (spins > 8000000u) { *fail = 1; return; }
        __builtin_amdgcn_s_sleep (32);
    }
    __atomic_thread_fence (__ATOMIC_ACQUIRE);
}

// ---- low: a big-number sum, one thread per symbol -----------------------------------------------------------------
// Symbol i adds a_i = cum_i * r_i into the 32-bit window of `low` after P_i bytes have left it (P = prefix sum of the
// per-symbol shift counts k_i = clz(r_i * freq_i) / 8). In the output stream (byte 0 = the coder's initial cache byte)
// that is: the four bytes of a_i are added at bytes P_i+1 .. P_i+4. Nothing else happens to low, so the whole thing is
//   k_chain_expand (above) a = cum * r and the shift count k of every symbol, the shifts per slice
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
#define GZ_LOW_RUN (GZ_LOW_SLICES_PER_WG / 4)
struct GzdLowBlock { uint32_t leaf, first_slice; };

__device__ static inline uint32_t d_low_nslices (uint32_t n) { return n ? (n + GZ_LOW_SLICE - 1) / GZ_LOW_SLICE : 1; }

// a workgroup's 64 slices: from the table (whole leaves), or - following the chain chunk by chunk - grid (listed leaves,
// chunk / 4096) over the position chunk that starts at p0
__device__ static inline GzdLowBlock d_low_block (const GzdLowBlock *blocks, const uint32_t *list, uint32_t p0)
{
    if (blocks) return blocks[blockIdx.x];
    GzdLowBlock b;
    b.leaf = list[blockIdx.x]; b.first_slice = p0 / GZ_LOW_SLICE + blockIdx.y * GZ_LOW_SLICES_PER_WG;
    return b;
}

// What the low kernels need of every symbol, from the chain's checkpoints: a lane per 64-symbol slice replays the recurrence over its
// slice in the plain formulation (d_chain_step: any total, an ordinary multiply) and must arrive at the chain's NEXT checkpoint -
// the chain got there through its fused multiply-adds and 64 hops from lane to lane; a slice that does not fails the stream (it never
// has; the check costs one comparison per 64 symbols). Per symbol: a = cum * r, what it adds to low (the wave's 64 x 64 values go
// through LDS so that they are written row by row), and k = the bytes low and range move up after it - two bits, the slice's 64 of
// them are 16 bytes written by its lane, and their sum is the slice's entry in the prefix sum of output positions (k_low_scan).
// (Until round 4 this kernel wrote r, and k_low_count / k_low_scatter each read every symbol's 16-byte record again for freq and cum:
// 60 bytes of traffic per symbol between the three; round 4: 25; with the 12-byte record: 21.)
// Same grid as the low kernels that follow it (one workgroup = 64 slices), 64 threads, GZ_EXPAND_LDS bytes of LDS.
#define GZ_EXPAND_TILE_BYTES (64 * 33 * 4)          // 8 448: the tile of a = cum * r values, [64 slices][32 symbols + 1] - a slice's 64 symbols go out in two halves
                                                    // (a tile of all 64 made it 23.8 KB of LDS a workgroup of ONE wave: six waves a compute unit; now ten)
#define GZ_EXPAND_ROW 28                            // dwords of a slice's row in the staging area: 8 records x 3 + 4 (rows stay 16-byte aligned)
#define GZ_EXPAND_LDS (GZ_EXPAND_TILE_BYTES + 64 * GZ_EXPAND_ROW * 4)
// fault: 0, or (GZ_DEBUG_CHAIN_FAULT, tests only) 1 + the index of a slice that is treated as if it had missed the chain's checkpoint
__global__ void __launch_bounds__(64) k_chain_expand (GzdLeaf *leaves, const GzdLowBlock *blocks, const uint32_t *list, uint32_t p0, uint32_t fault)
{
    const GzdLowBlock B = d_low_block (blocks, list, p0);
    GzdLeaf &L = leaves[B.leaf];
    if (!L.active || L.engine != GZ_ENG_ARITH) return;
    const uint32_t n = L.arith_n, ns = d_low_nslices (n);
    if (B.first_slice >= ns) return;
    gz_f64_round_toward_zero ();
    const int lane = threadIdx.x;
    const uint8_t *rec = L.triples;
    uint32_t *av = (uint32_t *)L.rvals;
    uint32_t *tile = (uint32_t *)gz_lds;                        // [64 slices][33]
    const uint32_t slice = B.first_slice + lane, i0 = slice * GZ_LOW_SLICE;
    const bool mine = slice < ns;
    const uint32_t *ck = (const uint32_t *)L.ckpt + 2 * (size_t)(mine ? slice : 0);
    uint32_t rlo = ck[0], rhi = ck[1], kb[4] = { 0, 0, 0, 0 }, ksum = 0;
    const uint32_t m = !mine || i0 >= n ? 0u : (i0 + 64 <= n ? 64u : n - i0);      // (an empty leaf has one empty slice)
    // The records of 64 slices x 8 symbols at a time (96 bytes a slice), loaded COALESCED - six lanes take a slice's six 16-byte pieces,
    // 384 pieces in six load instructions - and handed to the slices' lanes through the LDS. (As `rec[i0 + j]` per lane this was one flat
    // load per symbol waited for on the spot; eight global loads per trip, still one line per lane, made 6.4 -> 4.7 ms of it per default
    // step: every line then came from the L2 eight times.) The next eight are in flight while these are worked on.
    uint32_t *stage = (uint32_t *)(gz_lds + GZ_EXPAND_TILE_BYTES);     // [64 slices][GZ_EXPAND_ROW]
    uint4 nx[6];
    auto load8 = [&] (uint32_t jbase) {
        #pragma unroll
        for (uint32_t it = 0; it < 6; it++) {
            const uint32_t g = it * 64 + (uint32_t)lane, s = g / 6, c = g % 6, sl = B.first_slice + s;
            const bool in = sl < ns && sl * GZ_LOW_SLICE + jbase < n;          // (a leaf's records are padded beyond n: whole pieces)
            nx[it] = gz_ldg_u32x4 (rec + (in ? ((size_t)sl * GZ_LOW_SLICE + jbase) * GZ_CHAIN_REC + c * 16 : 0));
        }
    };
    load8 (0);
    #pragma unroll
    for (uint32_t q = 0; q < 4; q++) {
        uint32_t kw = 0;
        #pragma unroll
        for (uint32_t h = 0; h < 2; h++) {
            const uint32_t j0 = q * 16 + h * 8;
            #pragma unroll
            for (uint32_t it = 0; it < 6; it++) { const uint32_t g = it * 64 + (uint32_t)lane; *(uint4 *)(stage + (g / 6) * GZ_EXPAND_ROW + (g % 6) * 4) = nx[it]; }
            gz_wave_sync ();
            if (j0 + 8 < 64) load8 (j0 + 8);
            uint4 c4[6];
            #pragma unroll
            for (uint32_t u = 0; u < 6; u++) c4[u] = *(const uint4 *)(stage + lane * GZ_EXPAND_ROW + u * 4);
            gz_wave_sync ();
            const uint32_t w[24] = { c4[0].x, c4[0].y, c4[0].z, c4[0].w, c4[1].x, c4[1].y, c4[1].z, c4[1].w, c4[2].x, c4[2].y, c4[2].z, c4[2].w,
                                     c4[3].x, c4[3].y, c4[3].z, c4[3].w, c4[4].x, c4[4].y, c4[4].z, c4[4].w, c4[5].x, c4[5].y, c4[5].z, c4[5].w };
            #pragma unroll
            for (uint32_t u = 0; u < 8; u++) if (j0 + u < m) {
                uint32_t k;
                const uint32_t r = d_chain_step (rlo, rhi, w[3 * u], w[3 * u + 1], d_record_freq (w[3 * u + 2]), &k);
                tile[lane * 33 + ((j0 + u) & 31)] = d_record_cum (w[3 * u]) * r;
                kw |= k << (2 * (h * 8 + u)); ksum += k;
            }
        }
        kb[q] = kw;
        if (q & 1) {                                            // 32 symbols of every slice are in the tile: out with them, two slices (2 x 128 bytes) a store
            gz_wave_sync ();
            const uint32_t half = q >> 1, hs = (uint32_t)lane >> 5, hl = (uint32_t)lane & 31;
            for (uint32_t s = 0; s < 64; s += 2) {
                const uint32_t sl = B.first_slice + s + hs, i = sl * GZ_LOW_SLICE + half * 32 + hl;
                if (sl < ns && i < n) gz_stg_u32 (av + i, tile[(s + hs) * 33 + hl]);
            }
            gz_wave_sync ();
        }
    }
    if (mine) {
        if (m && (rlo != ck[2] || rhi != ck[3] || slice + 1 == fault)) L.overflow = 2;
        ((uint32_t *)L.kpos)[slice] = ksum;
        ((uint4 *)L.kbits)[slice] = make_uint4 (kb[0], kb[1], kb[2], kb[3]);
    }
}

// one 1024-thread workgroup per leaf; the slices of positions [p0, p0 + chunk)
__global__ void __launch_bounds__(1024) k_low_scan (GzdLeaf *leaves, const uint32_t *list, uint32_t p0, uint32_t chunk)
{
    GzdLeaf &L = leaves[list[blockIdx.x]];
    if (!L.active || L.engine != GZ_ENG_ARITH) return;
    const uint32_t n = L.arith_n;
    if (p0 && n <= p0) return;
    const int tid = threadIdx.x;
    uint32_t *sh = (uint32_t *)gz_lds;
    uint32_t *kpos = (uint32_t *)L.kpos;
    const uint32_t ns = d_low_nslices (n);
    const uint32_t s0 = p0 / GZ_LOW_SLICE, s1 = (n - p0 > chunk) ? (p0 + chunk) / GZ_LOW_SLICE : ns;
    const uint32_t base = p0 ? L.low_base : 0u;
    const uint32_t per = (s1 - s0 + 1023) / 1024;
    const uint32_t a = s0 + tid * per < s1 ? s0 + tid * per : s1, b = a + per < s1 ? a + per : s1;
    uint32_t sum = 0;
    for (uint32_t i = a; i < b; i++) sum += kpos[i];
    sh[tid] = sum;
    __syncthreads ();
    if (!tid) {
        uint32_t run = base;
        for (int t = 0; t < 1024; t++) { uint32_t c = sh[t]; sh[t] = run; run += c; }
        sh[1024] = run;
    }
    __syncthreads ();
    uint32_t run = sh[tid];
    for (uint32_t i = a; i < b; i++) { uint32_t c = kpos[i]; kpos[i] = run; run += c; }
    if (!tid) {
        L.low_base = sh[1024];
        if (!p0) ((uint32_t *)L.events)[0] = 0;               // digit 0: the coder's initial cache byte
        if (s1 == ns) {
            kpos[ns] = sh[1024];
            L.n_events = sh[1024] + 5;                        // + RC_FinishEncode's 5 shifts
        }
    }
}

// 4 waves, each with a 140-word LDS accumulator
__global__ void __launch_bounds__(GZ_LOW_WG) k_low_scatter (GzdLeaf *leaves, const GzdLowBlock *blocks, const uint32_t *list, uint32_t p0)
{
    const GzdLowBlock B = d_low_block (blocks, list, p0);
    GzdLeaf &L = leaves[B.leaf];
    if (!L.active || L.engine != GZ_ENG_ARITH) return;
    const uint32_t n = L.arith_n, ns = d_low_nslices (n);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t *av = (const uint32_t *)L.rvals, *kbits = (const uint32_t *)L.kbits;
    const uint32_t *kpos = (const uint32_t *)L.kpos;
    uint32_t *dig = (uint32_t *)L.events;
    uint32_t *acc = (uint32_t *)gz_lds + wave * 144;          // up to 128 own digits + 5 closing / 4 spilling
    uint32_t *resid = (uint32_t *)L.resid;
    // What a slice adds beyond the digits it owns (up to 4) belongs to the following slices' digits. A wave walks GZ_LOW_RUN consecutive
    // slices, so it carries them into its next slice's accumulator itself (lanes 0..3 hold them); only what the last slice of a run leaves
    // goes through memory (resid, added by k_low_resid once every digit is stored: 1 slice in 16 instead of every one - that kernel was
    // 0.6 ms at the tail of the step).
    uint32_t carry_in = 0;
    // (the run's 16 x 64 values and shift counts are requested up front through global loads - one trip to memory per wave instead of
    //  one per slice - and the accumulators are a wave's own: wave-level syncs, no workgroup barrier)
    uint32_t a_all[GZ_LOW_RUN], k_all[GZ_LOW_RUN];
    const uint32_t slice0 = B.first_slice + wave * GZ_LOW_RUN;
    const uint32_t kp_all = gz_ldg_u32 (kpos + (lane < (int)GZ_LOW_RUN && slice0 + lane < ns ? slice0 + lane : 0));   // lane q: where slice q's digits start
    #pragma unroll
    for (uint32_t q = 0; q < GZ_LOW_RUN; q++) {
        const uint32_t slice = B.first_slice + wave * GZ_LOW_RUN + q, i = slice * GZ_LOW_SLICE + lane;
        const bool in = slice < ns && i < n;
        a_all[q] = gz_ldg_u32 (av + (in ? i : 0));
        k_all[q] = gz_ldg_u32 (kbits + (in ? slice * 4 + (lane >> 4) : 0));
        if (!in) { a_all[q] = 0; k_all[q] = 0; }
    }
    #pragma unroll
    for (uint32_t q = 0; q < GZ_LOW_RUN; q++) {
        const uint32_t slice = B.first_slice + wave * GZ_LOW_RUN + q;
        const bool on = slice < ns;
        const uint32_t a = a_all[q], k = (k_all[q] >> (2 * (lane & 15))) & 3u;
        const uint64_t m1 = __ballot (k >= 1), m2 = __ballot (k == 2);
        const uint32_t P = gz_mbcnt (m1) + gz_mbcnt (m2);     // shifts before me in the slice (the masks are scalars: v_mbcnt, no 64-bit and)
        const uint32_t K = (uint32_t)__popcll (m1) + (uint32_t)__popcll (m2);
        const bool last = on && slice == ns - 1;
        const uint32_t own = last ? K + 5 : K;                 // digits this slice owns: one per shift (+ the closing 5)
        // (this kernel runs at the device's vector issue rate - profiles/r05b_sq_*.txt -: only the words this slice can touch are cleared - its own
        //  digits and the four behind them; a quality stream's slice has ~25 shifts, one store instead of three)
        acc[lane] = lane < 4 ? carry_in : 0u;
        if (own + 4 > 64)  acc[64 + lane] = 0u;
        if (own + 4 > 128 && lane < 16) acc[128 + lane] = 0u;
        gz_wave_sync ();
        if (on && a) {
            atomicAdd (&acc[P],     a >> 24);
            atomicAdd (&acc[P + 1], (a >> 16) & 0xff);
            atomicAdd (&acc[P + 2], (a >> 8) & 0xff);
            atomicAdd (&acc[P + 3], a & 0xff);
        }
        gz_wave_sync ();
        carry_in = 0;
        if (on) {
            const uint32_t base = d_readlane (kp_all, (int)q) + 1;               // output byte of this slice's first shift (q is a constant of the unrolled loop: v_readlane, no trip to the LDS)
            for (uint32_t j = lane; j < own; j += 64) gz_stg_u32 (dig + base + j, acc[j]);      // (global, not flat: see d_record_store)
            if (lane < 4) {
                const uint32_t left = last ? 0u : acc[own + lane];
                if (q == GZ_LOW_RUN - 1) gz_stg_u32 (resid + (slice / GZ_LOW_RUN) * 4 + lane, left); else carry_in = left;
            }
        }
        gz_wave_sync ();
    }
}

// what the last slice of every run of GZ_LOW_RUN leaves for the slices after it: one thread per run, 16 table entries (of 4 runs) per wave
__global__ void __launch_bounds__(GZ_LOW_WG) k_low_resid (GzdLeaf *leaves, const GzdLowBlock *blocks, uint32_t n_blocks)
{
    const uint32_t bi = (blockIdx.x * (GZ_LOW_WG / 64) + (threadIdx.x >> 6)) * 16 + ((threadIdx.x & 63) >> 2);
    if (bi >= n_blocks) return;
    const GzdLowBlock B = blocks[bi];
    GzdLeaf &L = leaves[B.leaf];
    if (!L.active || L.engine != GZ_ENG_ARITH) return;
    const uint32_t ns = d_low_nslices (L.arith_n), m = L.n_events;
    const uint32_t slice = B.first_slice + (threadIdx.x & 3) * GZ_LOW_RUN + GZ_LOW_RUN - 1;
    if (slice + 1 >= ns) return;                               // the last slice owns everything it touches
    const uint4 r = ((const uint4 *)L.resid)[slice / GZ_LOW_RUN];
    if (!(r.x | r.y | r.z | r.w)) return;
    uint32_t *dig = (uint32_t *)L.events;
    const uint32_t at = ((const uint32_t *)L.kpos)[slice + 1] + 1;   // first digit of the next slice
    if (r.x && at < m)     atomicAdd (&dig[at], r.x);
    if (r.y && at + 1 < m) atomicAdd (&dig[at + 1], r.y);
    if (r.z && at + 2 < m) atomicAdd (&dig[at + 2], r.z);
    if (r.w && at + 3 < m) atomicAdd (&dig[at + 3], r.w);
}

// grid (leaves, ranges): a 1024-thread workgroup per GZ_NORM_RANGE tiles of a leaf's digits (one workgroup per leaf took 0.44 ms at the
// tail of the step for the 2.7 MB of a quality stream). Tiles of 1024 x 16 digits are normalised from the end of the range towards
// its start; a thread turns its 16 digits into 16 bytes (a 128-bit big-endian number in 4 words) plus a carry-out,
// carries then move one thread to the left per round as a 128-bit add until none is left (normally one round). What leaves a range
// on its left is added to the bytes before it by k_low_carry (addition is associative: the range before was normalised as if nothing came).
#define GZ_NORM_NT 1024
#define GZ_NORM_PER 16
#define GZ_NORM_RANGE 8
__global__ void __launch_bounds__(GZ_NORM_NT) k_low_norm (GzdLeaf *leaves, const uint32_t *list)
{
    GzdLeaf &L = leaves[list[blockIdx.x]];
    if (!L.active || L.engine != GZ_ENG_ARITH) return;
    const int tid = threadIdx.x;
    const uint32_t m = L.n_events;
    const uint32_t *dig = (const uint32_t *)L.events;
    uint8_t *out = L.pay + 1;
    uint32_t *range_out = (uint32_t *)L.resid;      // (k_low_resid is through with it) [range]: what leaves the range on its left
    uint32_t *sh = (uint32_t *)gz_lds;              // [0..NT] carries, [NT+8] "any carry left", [NT+9] carry into the next tile
    const uint32_t tile = GZ_NORM_NT * GZ_NORM_PER;
    const uint32_t ntiles = (m + tile - 1) / tile;
    const uint32_t t_lo = blockIdx.y * GZ_NORM_RANGE, t_hi = t_lo + GZ_NORM_RANGE < ntiles ? t_lo + GZ_NORM_RANGE : ntiles;
    if (t_lo >= ntiles) return;
    if (!tid) sh[GZ_NORM_NT + 9] = 0;
    __syncthreads ();
    for (uint32_t t = t_hi; t-- > t_lo; ) {
        const uint32_t b0 = t * tile + tid * GZ_NORM_PER;
        uint32_t w[4] = { 0, 0, 0, 0 };             // w[0] most significant: bytes b0..b0+3
        uint32_t carry = 0;
        uint32_t d16[GZ_NORM_PER];                   // (four 16-byte loads when the thread's digits all exist)
        if (b0 + GZ_NORM_PER <= m) {
            #pragma unroll
            for (int q = 0; q < GZ_NORM_PER / 4; q++) {
                const uint4 v4 = ((const uint4 *)(dig + b0))[q];
                d16[4 * q] = v4.x; d16[4 * q + 1] = v4.y; d16[4 * q + 2] = v4.z; d16[4 * q + 3] = v4.w;
            }
        }
        else {
            #pragma unroll
            for (int j = 0; j < GZ_NORM_PER; j++) d16[j] = b0 + j < m ? dig[b0 + j] : 0u;
        }
        #pragma unroll
        for (int j = GZ_NORM_PER - 1; j >= 0; j--) {
            const uint32_t v = d16[j] + carry;
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
        if (b0 + GZ_NORM_PER <= m) {                 // one 16-byte store (w[] is big-endian: byte b0 is the top of w[0])
            gz_u32x4_unaligned o = { __builtin_bswap32 (w[0]), __builtin_bswap32 (w[1]), __builtin_bswap32 (w[2]), __builtin_bswap32 (w[3]) };
            *(gz_u32x4_unaligned *)(out + b0) = o;
        }
        else {
            #pragma unroll
            for (int j = 0; j < GZ_NORM_PER; j++) {
                const uint32_t idx = b0 + j;
                if (idx < m) out[idx] = (uint8_t)(w[j >> 2] >> (8 * (3 - (j & 3))));
            }
        }
        __syncthreads ();
    }
    if (!tid) range_out[blockIdx.y] = sh[GZ_NORM_NT + 9];
}

// grid (leaves), 64 threads: what left every range of k_low_norm on its left goes into the bytes before it (it ripples on while a byte
// overflows: rarely beyond one); then the stream's first byte and length
__global__ void __launch_bounds__(64) k_low_carry (GzdLeaf *leaves, const uint32_t *list)
{
    GzdLeaf &L = leaves[list[blockIdx.x]];
    if (!L.active || L.engine != GZ_ENG_ARITH || threadIdx.x) return;
    const uint32_t m = L.n_events;
    const uint32_t tile = GZ_NORM_NT * GZ_NORM_PER, ntiles = (m + tile - 1) / tile, nranges = (ntiles + GZ_NORM_RANGE - 1) / GZ_NORM_RANGE;
    const uint32_t *range_out = (const uint32_t *)L.resid;
    uint8_t *out = L.pay + 1;
    for (uint32_t y = nranges; y-- > 1; ) {
        uint32_t c = range_out[y];
        for (uint32_t idx = y * GZ_NORM_RANGE * tile; c && idx-- > 0; ) { const uint32_t v = out[idx] + c; out[idx] = (uint8_t)v; c = v >> 8; }
    }
    L.pay[0] = (uint8_t)(L.coded_n ? L.max_sym : 1);                     // max_sym + 1 (256 wraps to 0), arith_dynamic.c:103-108
    if (m + 1 > L.pay_cap) { L.overflow = 1; L.pay_len = 0; }            // cannot happen: pay_cap >= 2n + 64
    else L.pay_len = m + 1;
    L.tab_len = 0;
}
