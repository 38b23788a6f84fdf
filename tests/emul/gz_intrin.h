// tests/emul/gz_intrin.h -- TEST INFRASTRUCTURE: portable stand-ins for genozip_amd/csrc/gz_intrin.h
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
static inline void gz_scalar_store4 (uint32_t *dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    if (emu.cur % 64 == 0) { dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = d; }
}
static inline void gz_scalar_store1 (uint32_t *dst, uint32_t a) { if (emu.cur % 64 == 0) dst[0] = a; }
static inline void gz_scalar_store_flush (void) {}
static inline uint32_t gz_mbcnt (uint64_t m) { return (uint32_t)__builtin_popcountll (m & ((1ull << (emu.cur % 64)) - 1)); }
template <int OFF> static inline void gz_scalar_store4_at (uint32_t *dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { gz_scalar_store4 (dst + OFF / 4, a, b, c, d); }
static inline void gz_wait_scalar_loads (void) {}
static inline void gz_sched_fence (void) {}
static inline void gz_scalar_cache_inv (void) {}
static inline void gz_touch (const void *p, uint32_t &pit) { pit += *(const volatile uint8_t *)p; }
static inline void gz_touch_done (uint32_t &) {}
static inline uint32_t gz_wave_or_scan (uint32_t v)
{
    const int lane = (int)(emu.cur % 64);
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl ((int)v, lane >= d ? lane - d : lane); if (lane >= d) v |= o; }
    return v;
}
static inline void gz_wave_sync (void) { (void)__ballot (1); }
static inline void gz_wait_vector_mem (void) {}
static inline uint32_t gz_ldg_u8 (const uint8_t *p) { return *p; }
static inline uint32_t gz_ldg_u16 (const uint16_t *p) { return *p; }
static inline uint32_t gz_ldg_u32 (const uint32_t *p) { return *p; }
static inline uint2 gz_ldg_u32x2 (const void *p) { return *(const uint2 *)p; }
static inline uint4 gz_ldg_u32x4 (const void *p) { return *(const uint4 *)p; }
static inline void gz_stg_u8 (uint8_t *p, uint32_t v) { *p = (uint8_t)v; }
static inline void gz_stg_u32 (uint32_t *p, uint32_t v) { *p = v; }
static inline void gz_stg_u16 (uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static inline void gz_stg_rec (void *p, uint32_t a, uint32_t b, uint32_t c) { uint32_t *q = (uint32_t *)p; q[0] = a; q[1] = b; q[2] = c; }
static inline uint32_t gz_wave_shr1 (uint32_t x, uint32_t fill)
{
    const int lane = (int)(emu.cur % 64);
    const uint32_t o = (uint32_t)emu_shfl ((int)x, lane ? lane - 1 : 0);
    return lane ? o : fill;
}
static inline void gz_hit_window (uint32_t d, uint32_t q, uint32_t x, uint32_t xl, uint32_t &ex, uint32_t &ed, uint32_t &eq, uint32_t &exl)
{
    const unsigned long long m = __ballot (d < q);
    const int l = m ? __builtin_ctzll (m) : 0;
    ex = (uint32_t)emu_shfl ((int)x, l); ed = (uint32_t)emu_shfl ((int)d, l); eq = (uint32_t)emu_shfl ((int)q, l); exl = (uint32_t)emu_shfl ((int)xl, l);
}
static inline void gz_opaque (uint32_t &) {}
static inline uint32_t gz_first_lane (uint32_t v) { return (uint32_t)emu_shfl ((int)v, 0); }
static inline double gz_rcp_f64 (double x) { return (double)(1.0f / (float)x) * (1.0 - 3e-8); }   // (a SEED of single precision, like v_rcp_f64: the caller's refinement and correction must hold)
// ---- the range coder chain in double precision (product: rounding mode register + an inline-asm loop over 64-symbol blocks) ----
#include <fenv.h>
#include <math.h>
#include <string.h>
static inline void gz_f64_round_toward_zero (void) {}
static inline double gz_fma_rtz (double a, double b, double c)
{
    const int was = fegetround (); fesetround (FE_TOWARDZERO);
    volatile double va = a, vb = b, vc = c; const double r = fma (va, vb, vc);
    fesetround (was); return r;
}
static inline void gz_scalar_store2 (uint32_t *dst, uint32_t a, uint32_t b) { if (emu.cur % 64 == 0) { dst[0] = a; dst[1] = b; } }
#include "gz_chain_asm.h"                                       // (GZ_CHAIN_BLOCK; the loop itself is not for this compiler)
// the same contract as the product's loop, one symbol at a time, in the loop's own arithmetic: records { inv.lo | cum, inv.hi, F.hi },
// T = fma (R, inv, 1.0) truncated (inv as it stands in the record, cum in its low bits), R' = fma (T, F, -F) with F = freq * 2^45, then the exponent bits
static inline void gz_chain_blocks (uint32_t &rlo, uint32_t &rhi, const uint8_t *recs, uint32_t nblk, uint32_t *ck)
{
    for (uint32_t b = 0; b < nblk; b++) {
        const uint32_t *rec = (const uint32_t *)(recs + (size_t)b * GZ_CHAIN_BLOCK * GZ_CHAIN_REC);
        for (int j = 0; j < GZ_CHAIN_BLOCK; j++) {
            if (!(j & 63)) gz_scalar_store2 (ck + 2 * (b * (GZ_CHAIN_BLOCK / 64) + j / 64), rlo, rhi);
            double inv, R, F; memcpy (&inv, rec + 3 * j, 8);
            uint64_t rb = (uint64_t)rlo | (uint64_t)rhi << 32; memcpy (&R, &rb, 8);
            const uint64_t fb = (uint64_t)rec[3 * j + 2] << 32; memcpy (&F, &fb, 8);
            const double t = gz_fma_rtz (R, inv, 1.0);
            uint64_t tb; memcpy (&tb, &t, 8); tb = (tb & 0xffffffffull) | 0x3ff0000000000000ull;     // (what the lane hop passes on: r under the constant high word)
            double t2; memcpy (&t2, &tb, 8);
            const double Pd = gz_fma_rtz (t2, F, -F);
            uint64_t pb; memcpy (&pb, &Pd, 8);
            rlo = (uint32_t)pb; rhi = ((uint32_t)(pb >> 32) & 0x007fffffu) | 0x41000000u;
        }
    }
}
