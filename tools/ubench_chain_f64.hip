// micro-benchmark + exactness check: the range coder's serial recurrence
//      r = range / tot ;  range = (r * freq) << 8k
// in double precision instead of seven scalar integer instructions. What is measured here (profiles/round4_ubench_chain_f64.txt):
//   * k_chain_hop    the product's loop (genozip_amd/csrc/gz_chain_asm.h, written by tools/gen_chain_asm.py): three dependent vector
//                    instructions per symbol, operands held by the lanes, the state hopping from lane to lane - its checkpoints and final
//                    range against the integer recurrence computed on the host (two data sets: one with 2 % of the totals below 256),
//                    clocks per symbol with 1 / 50 / 64 chains on the device
//   * k_chain_f64    a compiler-scheduled four-instruction form with scalar-register operands (how it started: exact, but its
//                    scalar loads cannot be waited for one at a time)
//   * k_chain_int    rounds 1-3's integer form, scalar operands, without the L2 touch of the product's old loop
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I tools tools/ubench_chain_f64.hip -o tools/ubench_f64.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((vector_size (16)));
typedef uint32_t u32x16 __attribute__((vector_size (64)));
typedef const __attribute__((address_space(4))) u32x16 *Rec4P;
typedef const __attribute__((address_space(4))) u32x4 *RecP;

__device__ static inline double d_as_double (uint32_t lo, uint32_t hi) { return __hiloint2double ((int)hi, (int)lo); }

#define GZ_F64_CONSTS uint32_t g_mask, g_expo; asm volatile ("v_mov_b32 %0, 0x7fffff\n\tv_mov_b32 %1, 0x41000000" : "=v"(g_mask), "=v"(g_expo))
#define d_step_f64(R, a, b, c, d, c52) d_step_f64_ (R, a, b, c, d, c52, g_mask, g_expo)
__device__ static __forceinline__ void d_step_f64_ (double &R, uint32_t inv_lo, uint32_t inv_hi, uint32_t f_lo, uint32_t f_hi, double c52, uint32_t g_mask, uint32_t g_expo)
{
    const double t = __builtin_fma (R, d_as_double (inv_lo, inv_hi), c52);
    const double rd = t - c52;
    const double P = rd * d_as_double (f_lo, f_hi);
    uint32_t hi;
    asm ("v_and_or_b32 %0, %1, %2, %3" : "=v"(hi) : "v"(__double2hiint (P)), "v"(g_mask), "v"(g_expo));   // (constants in registers: VOP3 takes no literals)
    R = d_as_double ((uint32_t)__double2loint (P), hi);
}

// integer form (the shipped chain's): record { freq, magic, shift, inc }
__device__ static __forceinline__ void d_step_int (uint32_t &range, uint32_t freq, uint32_t mg, uint32_t shw, uint32_t inc)
{
    const uint32_t r = __umulhi (mg, range + inc) >> (shw & 31);
    const uint32_t x = r * freq;
    range = x << (__builtin_clz (x) & 0x18);
}

// one wave per block; records of block b start at rec + b * n
__global__ void __launch_bounds__(64) k_chain_f64 (const uint8_t *recs, uint32_t n, double *out, uint32_t *ck)
{
    // round toward zero for double precision (MODE.FP_ROUND bits 3:2)
    asm volatile ("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");       // (inline asm: the compiler resets a mode it knows about)
    GZ_F64_CONSTS;
    Rec4P rec4 = (Rec4P)(uintptr_t)recs;            // (every block walks the same records)
    double R = 4294967295.0 / 128.0;
    const double c52 = 4503599627370496.0;
    u32x16 a0 = rec4[0], a1 = rec4[1], b0, b1;
#define HALF(C0, C1, N0, N1, K) do { \
        __builtin_amdgcn_s_waitcnt (0xC07F); \
        N0 = rec4[(i >> 2) + 2 * (K) + 2]; N1 = rec4[(i >> 2) + 2 * (K) + 3]; \
        __builtin_amdgcn_sched_barrier (0); \
        d_step_f64 (R, C0[0], C0[1], C0[2], C0[3], c52);   d_step_f64 (R, C0[4], C0[5], C0[6], C0[7], c52); \
        d_step_f64 (R, C0[8], C0[9], C0[10], C0[11], c52); d_step_f64 (R, C0[12], C0[13], C0[14], C0[15], c52); \
        d_step_f64 (R, C1[0], C1[1], C1[2], C1[3], c52);   d_step_f64 (R, C1[4], C1[5], C1[6], C1[7], c52); \
        d_step_f64 (R, C1[8], C1[9], C1[10], C1[11], c52); d_step_f64 (R, C1[12], C1[13], C1[14], C1[15], c52); \
    } while (0)
    for (uint32_t i = 0; i + 64 <= n; i += 64) {
        if (ck) { const uint32_t lo = __builtin_amdgcn_readfirstlane (__double2loint (R)), hi = __builtin_amdgcn_readfirstlane (__double2hiint (R));
                  asm volatile ("s_store_dwordx2 %0, %1, 0x0" : : "s"((uint64_t)lo | (uint64_t)hi << 32), "s"(ck + (size_t)blockIdx.x * (n / 32) + (i >> 5)) : "memory"); }
        HALF (a0, a1, b0, b1, 0); HALF (b0, b1, a0, a1, 1); HALF (a0, a1, b0, b1, 2); HALF (b0, b1, a0, a1, 3);
        HALF (a0, a1, b0, b1, 4); HALF (b0, b1, a0, a1, 5); HALF (a0, a1, b0, b1, 6); HALF (b0, b1, a0, a1, 7);
    }
#undef HALF
    if (threadIdx.x == 0) out[blockIdx.x] = R;
}

__global__ void __launch_bounds__(64) k_chain_int (const uint8_t *recs, uint32_t n, uint32_t *out, uint32_t *ck)
{
    Rec4P rec4 = (Rec4P)(uintptr_t)recs;            // (every block walks the same records)
    uint32_t range = 0xffffffffu;
    u32x16 a0 = rec4[0], a1 = rec4[1], b0, b1;
#define HALF(C0, C1, N0, N1, K) do { \
        __builtin_amdgcn_s_waitcnt (0xC07F); \
        N0 = rec4[(i >> 2) + 2 * (K) + 2]; N1 = rec4[(i >> 2) + 2 * (K) + 3]; \
        __builtin_amdgcn_sched_barrier (0); \
        d_step_int (range, C0[0], C0[1], C0[2], C0[3]);   d_step_int (range, C0[4], C0[5], C0[6], C0[7]); \
        d_step_int (range, C0[8], C0[9], C0[10], C0[11]); d_step_int (range, C0[12], C0[13], C0[14], C0[15]); \
        d_step_int (range, C1[0], C1[1], C1[2], C1[3]);   d_step_int (range, C1[4], C1[5], C1[6], C1[7]); \
        d_step_int (range, C1[8], C1[9], C1[10], C1[11]); d_step_int (range, C1[12], C1[13], C1[14], C1[15]); \
    } while (0)
    for (uint32_t i = 0; i + 64 <= n; i += 64) {
        if (ck) asm volatile ("s_store_dword %0, %1, 0x0" : : "s"(range), "s"(ck + (size_t)blockIdx.x * (n / 32) + (i >> 6)) : "memory");
        HALF (a0, a1, b0, b1, 0); HALF (b0, b1, a0, a1, 1); HALF (a0, a1, b0, b1, 2); HALF (b0, b1, a0, a1, 3);
        HALF (a0, a1, b0, b1, 4); HALF (b0, b1, a0, a1, 5); HALF (a0, a1, b0, b1, 6); HALF (b0, b1, a0, a1, 7);
    }
#undef HALF
    if (threadIdx.x == 0) out[blockIdx.x] = range;
}

#ifdef GZ_CHAIN_HDR
#include GZ_CHAIN_HDR            // (tools/probes/chain_variants.sh: the loop with parts left out)
#else
#include "../genozip_amd/csrc/gz_chain_asm.h"
#endif
// one symbol, any total (the fast loop leaves blocks with a total below 256 to this)
__device__ static inline void d_step_slow (uint32_t &rlo, uint32_t &rhi, const uint32_t *rec)
{
    const uint32_t il = __builtin_amdgcn_readfirstlane (rec[0]), ih = __builtin_amdgcn_readfirstlane (rec[1]), fq = __builtin_amdgcn_readfirstlane (rec[2]);
    const double t = __builtin_fma (__hiloint2double ((int)rhi, (int)rlo), __hiloint2double ((int)ih, (int)il), 4503599627370496.0);
    const uint32_t P = (uint32_t)__double2loint (t) * fq;
    const double Pd = (double)P * 0.0078125;
    rlo = (uint32_t)__double2loint (Pd); rhi = ((uint32_t)__double2hiint (Pd) & 0x007fffffu) | 0x41000000u;
}
template <int NOPS> __global__ void __launch_bounds__(64) k_chain_hop (const uint8_t *recs, uint32_t n, double *out, uint32_t *ck, size_t stride, uint64_t *cyc)
{
    asm volatile ("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
    const double R0 = 4294967295.0 / 128.0;
    uint32_t rlo = (uint32_t)__double2loint (R0), rhi = (uint32_t)__double2hiint (R0);
    const uint8_t *base = recs + (size_t)blockIdx.x * stride;
    uint32_t *c = ck + (size_t)blockIdx.x * (n / 32);
    uint32_t nblk = __builtin_amdgcn_readfirstlane (n / GZ_CHAIN_BLOCK), done = 0, slow = 0;
    const uint64_t t0 = __builtin_readcyclecounter (), w0 = wall_clock64 ();
    {
        const uint64_t b = (uint64_t)(uintptr_t)base, cc = (uint64_t)(uintptr_t)c;
        const uint32_t b_lo = __builtin_amdgcn_readfirstlane ((uint32_t)b), b_hi = __builtin_amdgcn_readfirstlane ((uint32_t)(b >> 32));
        const uint32_t c_lo = __builtin_amdgcn_readfirstlane ((uint32_t)cc), c_hi = __builtin_amdgcn_readfirstlane ((uint32_t)(cc >> 32));
        asm volatile (GZ_CHAIN_F64_ASM : [rlo] "+v"(rlo), [rhi] "+v"(rhi) : [blo] "s"(b_lo), [bhi] "s"(b_hi), [nblk] "s"(nblk), [clo] "s"(c_lo), [chi] "s"(c_hi) : GZ_CHAIN_F64_CLOBBERS);
        rlo = __builtin_amdgcn_readfirstlane (rlo); rhi = __builtin_amdgcn_readfirstlane (rhi);
        (void)done; (void)slow;
    }
    asm volatile ("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" : : : "memory");
    const uint64_t t1 = __builtin_readcyclecounter (), w1 = wall_clock64 ();
    if (threadIdx.x == 0) { out[blockIdx.x] = __hiloint2double ((int)rhi, (int)rlo); if (cyc) { cyc[3 * blockIdx.x] = t1 - t0; cyc[3 * blockIdx.x + 1] = w1 - w0; cyc[3 * blockIdx.x + 2] = slow; } }
}

// pure issue-rate probes: the four instructions with constant operands, dependent
__global__ void __launch_bounds__(64) k_probe_f64 (double *out, double inv, double F, int iters)
{
    asm volatile ("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
    GZ_F64_CONSTS;
    double R = 4294967295.0 / 128.0;
    const double c52 = 4503599627370496.0;
    const uint32_t il = __builtin_amdgcn_readfirstlane (__double2loint (inv)), ih = __builtin_amdgcn_readfirstlane (__double2hiint (inv));
    const uint32_t fl = __builtin_amdgcn_readfirstlane (__double2loint (F)), fh = __builtin_amdgcn_readfirstlane (__double2hiint (F));
    for (int i = 0; i < iters; i++) {
        d_step_f64 (R, il, ih, fl, fh, c52); d_step_f64 (R, il, ih, fl, fh, c52); d_step_f64 (R, il, ih, fl, fh, c52); d_step_f64 (R, il, ih, fl, fh, c52);
        d_step_f64 (R, il, ih, fl, fh, c52); d_step_f64 (R, il, ih, fl, fh, c52); d_step_f64 (R, il, ih, fl, fh, c52); d_step_f64 (R, il, ih, fl, fh, c52);
    }
    if (threadIdx.x == 0) out[blockIdx.x] = R;
}
// dependent v_fma_f64 only / dependent v_add_u32 only: the issue interval of each
__global__ void __launch_bounds__(64) k_probe_fma (double *out, double a, double b, int iters)
{
    double x = 1.0 + threadIdx.x * 1e-9;
    for (int i = 0; i < iters; i++) {
        x = __builtin_fma (x, a, b); x = __builtin_fma (x, a, b); x = __builtin_fma (x, a, b); x = __builtin_fma (x, a, b);
        x = __builtin_fma (x, a, b); x = __builtin_fma (x, a, b); x = __builtin_fma (x, a, b); x = __builtin_fma (x, a, b);
    }
    out[blockIdx.x * 64 + threadIdx.x] = x;
}
__global__ void __launch_bounds__(64) k_probe_iadd (uint32_t *out, uint32_t a, int iters)
{
    uint32_t x = threadIdx.x;
    for (int i = 0; i < iters; i++) {
        x = (x ^ a) + 3u; x = (x ^ a) + 3u; x = (x ^ a) + 3u; x = (x ^ a) + 3u;     // v_xad_u32: one instruction each
        x = (x ^ a) + 3u; x = (x ^ a) + 3u; x = (x ^ a) + 3u; x = (x ^ a) + 3u;
    }
    out[blockIdx.x * 64 + threadIdx.x] = x;
}

template <typename F> static float time_it (F f)
{
    hipEvent_t a, b; hipEventCreate (&a); hipEventCreate (&b);
    f (); hipDeviceSynchronize ();
    hipEventRecord (a); f (); hipEventRecord (b); hipEventSynchronize (b);
    float ms; hipEventElapsedTime (&ms, a, b); return ms;
}

static uint64_t rng_s = 88172645463325252ull;
static uint32_t rnd () { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (uint32_t)(rng_s >> 16); }

int main (int argc, char **argv)
{
    const uint32_t n = 21504u * 48;         // symbols per chain: a multiple of every block size the generator can make (64 x 8 .. 16)
    const int max_blocks = 256;
    // records: a model-like sequence of (tot, freq): tot walks 32760..65519 in steps of 16 most of the time, sometimes small
    std::vector<uint32_t> ri ((size_t)n * 4), rf ((size_t)n * 4);
    std::vector<uint32_t> ranges (n + 1);
    uint32_t range = 0xffffffffu, tot = 7;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t kind = rnd () % 100;
        if (kind < 2) tot = 1 + rnd () % 300; else if (kind < 4) tot = 1 + rnd () % 65519; else { tot += 16; if (tot > 65519) tot = 32760 + rnd () % 16; }
        uint32_t freq = (rnd () % 4 == 0) ? 1 + rnd () % tot : (rnd () % 2 ? tot - rnd () % (tot < 64 ? tot : 64) : 1 + rnd () % (tot < 600 ? tot : 600));
        if (freq > tot) freq = tot;
        if (!freq) freq = 1;
        if (i == 0) { tot = 40; freq = 1; }
        ranges[i] = range;
        const uint32_t r = range / tot, x = r * freq;
        range = x << (__builtin_clz (x) & 0x18);
        // integer record: magic by the round-up / round-down scheme (computed simply here with 64-bit checks on this numerator only
        // is not enough for the product; for the benchmark any exact-enough magic will do: use 64-bit exact division to make the record
        // of THIS benchmark right for THIS range: mg = floor (2^32 * 2^sh / tot) + 1 variant is not needed - timing only)
        uint32_t sh = 31 - __builtin_clz (tot | 1); uint64_t mg = (((uint64_t)1 << (32 + sh)) / tot) + 1; if (mg > 0xffffffffull) mg = 0xffffffffull;
        ri[(size_t)i * 4 + 0] = freq; ri[(size_t)i * 4 + 1] = (uint32_t)mg; ri[(size_t)i * 4 + 2] = sh; ri[(size_t)i * 4 + 3] = 0;
        // f64 record: inv = RU (128 / tot), F = freq / 128 with a 16-bit payload in its low dword
        double inv = 128.0 / tot; if (inv * tot < 128.0 || fma (inv, (double)tot, -128.0) < 0) inv = nextafter (inv, 1e300);
        double F = freq / 128.0; uint64_t fb; memcpy (&fb, &F, 8); fb |= rnd () & 0xffff; uint64_t ib; memcpy (&ib, &inv, 8);
        rf[(size_t)i * 4 + 0] = (uint32_t)ib; rf[(size_t)i * 4 + 1] = (uint32_t)(ib >> 32); rf[(size_t)i * 4 + 2] = (uint32_t)fb; rf[(size_t)i * 4 + 3] = (uint32_t)(fb >> 32);
    }
    ranges[n] = range;
    uint8_t *d_ri, *d_rf; double *d_out; uint32_t *d_ck;
    CHK (hipMalloc (&d_ri, (size_t)n * 16 * 2 + 65536)); CHK (hipMalloc (&d_rf, (size_t)n * 16 * 2 + 65536)); CHK (hipMalloc (&d_out, 1 << 20)); CHK (hipMalloc (&d_ck, (size_t)max_blocks * (n / 32) * 4 + 4096));
    // every block reads the same records (blockIdx * n * 16 would need 256 copies: give blocks > 0 the same data by using n_eff = 0 stride)
    CHK (hipMemcpy (d_ri, ri.data (), (size_t)n * 16, hipMemcpyHostToDevice)); CHK (hipMemcpy (d_rf, rf.data (), (size_t)n * 16, hipMemcpyHostToDevice));

    // records of the hop kernel { inv.lo | cum, inv.hi, the high word of freq * 2^45 }, inv = 2^-45 / tot rounded up to a multiple of 2^16
    // units of the last place: data set A = the random sequence above (2 % of the totals below 256), data set B = totals of a busy
    // context (32760 .. 65519) after a short start
    auto fhi = [] (uint32_t freq) { double F = ldexp ((double)freq, 45); uint64_t b; memcpy (&b, &F, 8); return (uint32_t)(b >> 32); };
    std::vector<uint32_t> ra ((size_t)n * 3), rb ((size_t)n * 3), ranges_b (n + 1);
    std::vector<uint64_t> tab (65536 + 32);
    for (uint32_t dv = 1; dv < tab.size (); dv++) { double inv = 128.0 / dv; if (fma (inv, (double)dv, -128.0) < 0) inv = nextafter (inv, 1e300); inv = ldexp (inv, -52); memcpy (&tab[dv], &inv, 8); tab[dv] = (tab[dv] + 0xffffull) & ~0xffffull; }
    {
        uint32_t range = 0xffffffffu, tot = 40;
        for (uint32_t i = 0; i < n; i++) {
            tot += 16; if (tot > 65519) tot = 32760 + rnd () % 16;
            uint32_t freq = (rnd () % 4 == 0) ? 1 + rnd () % tot : (rnd () % 2 ? tot - rnd () % 64 : 1 + rnd () % 600);
            if (freq > tot) freq = tot;
            if (!freq) freq = 1;
            ranges_b[i] = range;
            const uint32_t r = range / tot, x = r * freq;
            range = x << (__builtin_clz (x) & 0x18);
            const uint64_t iv = tab[tot] | ((i & 7) == 0 ? 0xffffu : (rnd () & 0xffff));
            rb[(size_t)i * 3 + 0] = (uint32_t)iv; rb[(size_t)i * 3 + 1] = (uint32_t)(iv >> 32); rb[(size_t)i * 3 + 2] = fhi (freq);
        }
        ranges_b[n] = range;
        for (uint32_t i = 0; i < n; i++) {          // data set A: the same (tot, freq) as rf / ri
            const uint32_t freq = ri[(size_t)i * 4 + 0];
            double inv; uint64_t ib = (uint64_t)rf[(size_t)i * 4 + 0] | (uint64_t)rf[(size_t)i * 4 + 1] << 32; memcpy (&inv, &ib, 8);
            const uint32_t tot_a = (uint32_t)floor (128.0 / inv + 0.5);
            const uint64_t iv = tab[tot_a] | ((i & 7) == 0 ? 0xffffu : (rnd () & 0xffff));
            ra[(size_t)i * 3 + 0] = (uint32_t)iv; ra[(size_t)i * 3 + 1] = (uint32_t)(iv >> 32); ra[(size_t)i * 3 + 2] = fhi (freq);
        }
    }
    const int copies = 64;
    uint8_t *d_ra, *d_rbc; uint64_t *d_cyc; CHK (hipMalloc (&d_ra, (size_t)n * 12 + 65536)); CHK (hipMalloc (&d_rbc, (size_t)copies * n * 12 + 131072)); CHK (hipMalloc (&d_cyc, 4096 * 24));
    d_ra += 4096; d_rbc += 4096;
    CHK (hipMemcpy (d_ra, ra.data (), (size_t)n * 12, hipMemcpyHostToDevice));
    for (int k = 0; k < copies; k++) CHK (hipMemcpy (d_rbc + (size_t)k * n * 12, rb.data (), (size_t)n * 12, hipMemcpyHostToDevice));
    for (int k = 0; k < 20; k++) hipLaunchKernelGGL (k_probe_fma, dim3 (1024), dim3 (64), 0, 0, d_out, 0.999, 0.001, 100000);   // warm the clocks up
    CHK (hipDeviceSynchronize ());

    auto check = [&] (const char *what, const std::vector<uint32_t> &want_r) {
        std::vector<uint64_t> ck (n / 64); double Rend; uint64_t cy[3];
        std::vector<uint32_t> raw (n / 32); hipMemcpy (raw.data (), d_ck, (size_t)(n / 32) * 4, hipMemcpyDeviceToHost);
        for (uint32_t s = 0; s < n / 64; s++) ck[s] = (uint64_t)raw[2 * s] | (uint64_t)raw[2 * s + 1] << 32;
        hipMemcpy (&Rend, d_out, 8, hipMemcpyDeviceToHost); hipMemcpy (cy, d_cyc, 24, hipMemcpyDeviceToHost);
        uint32_t bad = 0;
        for (uint32_t s = 0; s < n / 64; s++) { double R; memcpy (&R, &ck[s], 8); const double want = want_r[s * 64]; const double got = floor (R * 128.0);
            if (got != want) { if (bad < 3) printf ("MISMATCH at symbol %u: got %.3f (R*128 = %.6f) want %.0f\n", s * 64, got, R * 128.0, want); bad++; } }
        printf ("exactness [%s]: %u of %u checkpoints differ; end range %.0f want %u; %llu blocks the slow way\n", what, bad, n / 64, floor (Rend * 128.0), want_r[n], (unsigned long long)cy[2]);
    };
#define HOPTEST(NOPS, what) do { \
        CHK (hipMemset (d_ck, 0, (size_t)(n / 32) * 4)); hipLaunchKernelGGL (k_chain_hop<NOPS>, dim3 (1), dim3 (64), 0, 0, d_ra, n, d_out, d_ck, (size_t)0, d_cyc); CHK (hipDeviceSynchronize ()); check (what ", data A", ranges); \
        CHK (hipMemset (d_ck, 0, (size_t)(n / 32) * 4)); hipLaunchKernelGGL (k_chain_hop<NOPS>, dim3 (1), dim3 (64), 0, 0, d_rbc, n, d_out, d_ck, (size_t)0, d_cyc); CHK (hipDeviceSynchronize ()); check (what ", data B", ranges_b); \
        for (int blocks : { 1, 50, 64 }) { uint64_t cy[3]; \
            float a = time_it ([&] { hipLaunchKernelGGL (k_chain_hop<NOPS>, dim3 (blocks), dim3 (64), 0, 0, d_rbc, n, d_out, d_ck, (size_t)n * 12, d_cyc); }); \
            CHK (hipMemcpy (cy, d_cyc, 24, hipMemcpyDeviceToHost)); \
            printf ("  %s: %2d chains at once: %.2f ns/symbol, %.2f shader cycles/symbol, %.3f GHz\n", what, blocks, a * 1e6 / n, (double)cy[0] / n, (double)cy[0] / cy[1] * 0.1); } } while (0)
    HOPTEST (2, "hop");
    if (getenv ("UB_OLD_FORMS")) for (int blocks : { 1, 50 }) {
        float b = time_it ([&] { hipLaunchKernelGGL (k_chain_int, dim3 (blocks), dim3 (64), 0, 0, d_ri, n, (uint32_t *)d_out, d_ck); });
        printf ("%2d chains at once: integer chain, scalar operands without the L2 touch %.2f ns/symbol\n", blocks, b * 1e6 / n);
    }
    return 0;
}
