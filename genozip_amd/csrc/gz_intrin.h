// gz_intrin.h -- the few places where plain HIP C++ cannot express the instruction we want.
// (tests/emul/gz_intrin.h provides the same functions in portable C++ for the CPU-emulated test build.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t gz_sgpr4 __attribute__((ext_vector_type(4)));

// four wave-uniform dwords to a wave-uniform, 16-byte aligned address with ONE scalar instruction (s_store_dwordx4)
// instead of 4 v_mov + an exec-masked vector store. The values must be scalar (SALU results).
__device__ static inline void gz_scalar_store4 (uint32_t *dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    gz_sgpr4 v = { a, b, c, d };
    asm volatile ("s_store_dwordx4 %0, %1, 0x0" : : "s"(v), "s"(dst) : "memory");
}

// one wave-uniform dword to a wave-uniform address (s_store_dword)
__device__ static inline void gz_scalar_store1 (uint32_t *dst, uint32_t a)
{
    asm volatile ("s_store_dword %0, %1, 0x0" : : "s"(a), "s"(dst) : "memory");
}

// the same with an immediate byte offset (one base pointer serves several stores)
template <int OFF> __device__ static inline void gz_scalar_store4_at (uint32_t *dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    gz_sgpr4 v = { a, b, c, d };
    asm volatile ("s_store_dwordx4 %0, %1, %2" : : "s"(v), "s"(dst), "n"(OFF) : "memory");
}

// wait for every outstanding scalar load (they return out of order, so "all" is the only wait there is) without
// waiting for vector memory operations. A builtin, not inline asm: the compiler's own wait-count bookkeeping sees it.
__device__ static inline void gz_wait_scalar_loads (void) { __builtin_amdgcn_s_waitcnt (0xC07F); }   // vmcnt 63, expcnt 7, lgkmcnt 0

// wait for every outstanding vector memory operation, said through the builtin so that the compiler's bookkeeping knows: loaded values
// that are stored again behind branches otherwise get a vmcnt(0) in front of every store - which also waits for the stores before it
__device__ static inline void gz_wait_vector_mem (void) { __builtin_amdgcn_s_waitcnt (0x0F70); }   // vmcnt 0, expcnt 7, lgkmcnt 15

// keep the instruction scheduler from moving anything across this point (it likes to sink loads towards their use)
__device__ static inline void gz_sched_fence (void) { __builtin_amdgcn_sched_barrier (0); }

// scalar stores sit in the scalar data cache: write it back before anybody else (a later kernel) reads the data
__device__ static inline void gz_scalar_store_flush (void)
{
    asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" : : : "memory");
}

// number of set bits of a wave mask below my lane (v_mbcnt_lo + v_mbcnt_hi)
__device__ static inline uint32_t gz_mbcnt (uint64_t m)
{
    return __builtin_amdgcn_mbcnt_hi ((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo ((uint32_t)m, 0u));
}

// inclusive OR over the lanes up to mine, one 32-bit word per lane: shifts inside the rows of 16 lanes (zeros come in at a row's
// start), then the last lane of row 0 / 2 into row 1 / 3 and lane 31 into rows 2 and 3 (DPP row_shr, row_bcast:15, row_bcast:31)
__device__ static inline uint32_t gz_wave_or_scan (uint32_t v)
{
    v |= (uint32_t)__builtin_amdgcn_update_dpp (0, (int)v, 0x111, 0xf, 0xf, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp (0, (int)v, 0x112, 0xf, 0xf, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp (0, (int)v, 0x114, 0xf, 0xf, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp (0, (int)v, 0x118, 0xf, 0xf, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp (0, (int)v, 0x142, 0xa, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp (0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

// drop the scalar data cache (after an acquire, before scalar loads of data another kernel has just written)
__device__ static inline void gz_scalar_cache_inv (void) { asm volatile ("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" : : : "memory"); }

// Pull the 64-byte line at p towards this wave's L2 without ever waiting for it: a plain global load whose result lands in
// `pit`, a register the caller keeps alive (and untouched) until gz_touch_done. Written as inline asm because the
// compiler would otherwise either wait for the value where it is consumed or move the load to where it is.
__device__ static inline void gz_touch (const void *p, uint32_t &pit)
{
    asm volatile ("global_load_dword %0, %1, off" : "+v"(pit) : "v"(p) : "memory");
}
__device__ static inline void gz_touch_done (uint32_t &pit)
{
    asm volatile ("s_waitcnt vmcnt(0)" : "+v"(pit) : : "memory");
}

// all lanes of the wave have performed their LDS accesses so far before any lane performs a later one (one wave's LDS
// operations execute in order; this only has to keep the compiler from reordering them)
__device__ static inline void gz_wave_sync (void) { __builtin_amdgcn_fence (__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier (); }

// v_rcp_f64: the reciprocal to about one ulp (not the IEEE division sequence)
__device__ static inline double gz_rcp_f64 (double x) { return __builtin_amdgcn_rcp (x); }
// ---- the arithmetic decoder's hand-over of the entry that is hit (gz_kernels_dec.h) ----
// lane l's x as seen by lane l + 1 (lane 0 keeps `fill`): v_mov_b32_dpp wave_shr:1
__device__ static inline uint32_t gz_wave_shr1 (uint32_t x, uint32_t fill) { return (uint32_t)__builtin_amdgcn_update_dpp ((int)fill, (int)x, 0x138, 0xf, 0xf, false); }
// the values x, d, q, xl of the FIRST lane with d < q, as wave-uniform values (no lane with d < q: lane 0's): exec is narrowed to those lanes
// by the compare itself, v_readfirstlane reads the first of them, exec comes back on. Only where every lane is active.
__device__ static inline void gz_hit_window (uint32_t d, uint32_t q, uint32_t x, uint32_t xl, uint32_t &ex, uint32_t &ed, uint32_t &eq, uint32_t &exl)
{
    asm volatile ("v_cmpx_lt_u32_e32 vcc, %4, %5\n\t"
                  "s_nop 4\n\t"              // (v_readfirstlane straight after a VALU write of exec still reads under the OLD mask: measured, tools/probes/lat_probe.hip - 4 wait states do)
                  "v_readfirstlane_b32 %0, %6\n\tv_readfirstlane_b32 %1, %4\n\tv_readfirstlane_b32 %2, %5\n\tv_readfirstlane_b32 %3, %7\n\t"
                  "s_mov_b64 exec, -1"
                  : "=&s" (ex), "=&s" (ed), "=&s" (eq), "=&s" (exl) : "v" (d), "v" (q), "v" (x), "v" (xl) : "vcc");
}
__device__ static inline void gz_opaque (uint32_t &v) { asm ("" : "+s" (v)); }      // a wave-uniform value the optimiser shall not look through
__device__ static inline uint32_t gz_first_lane (uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane ((int)v); }   // a wave-uniform value into an SGPR

// loads through a GLOBAL pointer: a load through a generic one is a flat load, which also counts as an LDS operation -
// every wait for an LDS read would then wait for its trip to memory as well
__device__ static inline uint32_t gz_ldg_u8 (const uint8_t *p) { return *(const __attribute__((address_space(1))) uint8_t *)(uintptr_t)p; }
__device__ static inline uint32_t gz_ldg_u16 (const uint16_t *p) { return *(const __attribute__((address_space(1))) uint16_t *)(uintptr_t)p; }
__device__ static inline uint32_t gz_ldg_u32 (const uint32_t *p) { return *(const __attribute__((address_space(1))) uint32_t *)(uintptr_t)p; }
__device__ static inline uint2 gz_ldg_u32x2 (const void *p)
{
    typedef uint32_t gz_v2 __attribute__((ext_vector_type(2)));
    const gz_v2 v = *(const __attribute__((address_space(1))) gz_v2 *)(uintptr_t)p;
    return make_uint2 (v.x, v.y);
}
__device__ static inline uint4 gz_ldg_u32x4 (const void *p)
{
    typedef uint32_t gz_v4 __attribute__((ext_vector_type(4)));
    const gz_v4 v = *(const __attribute__((address_space(1))) gz_v4 *)(uintptr_t)p;
    return make_uint4 (v.x, v.y, v.z, v.w);
}
__device__ static inline void gz_stg_u8 (uint8_t *p, uint32_t v) { *(__attribute__((address_space(1))) uint8_t *)(uintptr_t)p = (uint8_t)v; }
__device__ static inline void gz_stg_u32 (uint32_t *p, uint32_t v) { *(__attribute__((address_space(1))) uint32_t *)(uintptr_t)p = v; }
__device__ static inline void gz_stg_rec (void *p, uint32_t a, uint32_t b, uint32_t c)     // 12 bytes, 4-byte aligned, through a GLOBAL pointer (global_store_dwordx3)
{
    typedef uint32_t gz_v3 __attribute__((ext_vector_type(3)));
    typedef gz_v3 __attribute__((aligned(4))) gz_v3_a4;
    const gz_v3 v = { a, b, c };
    *(__attribute__((address_space(1))) gz_v3_a4 *)(uintptr_t)p = v;
}
__device__ static inline void gz_stg_u16 (uint8_t *p, uint32_t v)     // 2 bytes, any alignment, through a GLOBAL pointer
{
    typedef uint16_t __attribute__((aligned(1))) gz_u16_unaligned;
    *(__attribute__((address_space(1))) gz_u16_unaligned *)(uintptr_t)p = (uint16_t)v;
}

// ---- double precision with truncation: the range coder chain (gz_kernels_arith.h) ----------------------------------------
// INVARIANT (k_arith_chain, k_chain_expand and whatever they call): the rounding mode below is switched behind the compiler's back, so
// these kernels must hold NO other double- or half-precision arithmetic - none that the compiler could fold at compile time (it folds with
// round-to-nearest), move across the s_setreg, or that needs the default mode; gz_fma_rtz's operands must never be compile-time constants
// together. The mode ends with the wave (both kernels are leaf kernels: nothing runs after them in the same wave). The generated loop
// (gz_chain_asm.h) names its registers itself (GZ_CHAIN_VREGS of the generated header, s36-s47: the clobber list keeps the compiler off them). What guards all of this at
// run time: k_chain_expand replays every 64-symbol slice in the other formulation and fails the stream (GZ_ST_FAILED) when it does not
// arrive at the chain's next checkpoint - tests/test_gpu.py::test_chain_checkpoint_guard forces such a mismatch and asserts the failure.
// MODE.FP_ROUND[3:2] = 3: double precision rounds toward zero in this wave from here on. (Inline asm: a mode change the
// compiler knows about is undone by it in front of the next floating point instruction.)
__device__ static inline void gz_f64_round_toward_zero (void) { asm volatile ("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3" : : : "memory"); }
// a * b + c in a wave that has called gz_f64_round_toward_zero
__device__ static inline double gz_fma_rtz (double a, double b, double c) { return __builtin_fma (a, b, c); }

#ifdef GZ_CHAIN_ASM_HDR
#include GZ_CHAIN_ASM_HDR               // (experiments: another build of the generated loop, tools/build_variant.sh)
#else
#include "gz_chain_asm.h"
#endif
// Whole blocks of GZ_CHAIN_BLOCK symbols of one leaf's chain (tools/gen_chain_asm.py explains the loop). (rlo, rhi) = the state,
// a double (range * 2^-7), wave-uniform in and out; recs = the 12-byte records of the first block; ck = where the first block's first
// checkpoint goes (8 bytes per 64 symbols, scalar stores).
__device__ static inline void gz_chain_blocks (uint32_t &rlo, uint32_t &rhi, const uint8_t *recs, uint32_t nblk, uint32_t *ck)
{
    const uint64_t b = (uint64_t)(uintptr_t)recs, c = (uint64_t)(uintptr_t)ck;
    const uint32_t b_lo = (uint32_t)__builtin_amdgcn_readfirstlane ((int)(uint32_t)b), b_hi = (uint32_t)__builtin_amdgcn_readfirstlane ((int)(uint32_t)(b >> 32));
    const uint32_t c_lo = (uint32_t)__builtin_amdgcn_readfirstlane ((int)(uint32_t)c), c_hi = (uint32_t)__builtin_amdgcn_readfirstlane ((int)(uint32_t)(c >> 32));
    const uint32_t nb = (uint32_t)__builtin_amdgcn_readfirstlane ((int)nblk);
    asm volatile (GZ_CHAIN_F64_ASM : [rlo] "+v"(rlo), [rhi] "+v"(rhi)
                                   : [blo] "s"(b_lo), [bhi] "s"(b_hi), [nblk] "s"(nb), [clo] "s"(c_lo), [chi] "s"(c_hi) : GZ_CHAIN_F64_CLOBBERS);
    rlo = (uint32_t)__builtin_amdgcn_readfirstlane ((int)rlo); rhi = (uint32_t)__builtin_amdgcn_readfirstlane ((int)rhi);   // (the state ends in lane 0)
}
// one 8-byte checkpoint through the scalar unit
__device__ static inline void gz_scalar_store2 (uint32_t *dst, uint32_t a, uint32_t b)
{
    const uint32_t sa = (uint32_t)__builtin_amdgcn_readfirstlane ((int)a), sb = (uint32_t)__builtin_amdgcn_readfirstlane ((int)b);   // (wave-uniform values that live in vector registers)
    asm volatile ("s_store_dwordx2 %0, %1, 0x0" : : "s"((uint64_t)sa | (uint64_t)sb << 32), "s"(dst) : "memory");
}
