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
static inline void gz_wave_sync (void) { (void)__ballot (1); }
static inline void gz_wait_vector_mem (void) {}
static inline uint32_t gz_ldg_u8 (const uint8_t *p) { return *p; }
static inline uint32_t gz_ldg_u16 (const uint16_t *p) { return *p; }
static inline uint32_t gz_ldg_u32 (const uint32_t *p) { return *p; }
static inline uint4 gz_ldg_u32x4 (const void *p) { return *(const uint4 *)p; }
static inline void gz_stg_u8 (uint8_t *p, uint32_t v) { *p = (uint8_t)v; }
static inline void gz_stg_u32 (uint32_t *p, uint32_t v) { *p = v; }
static inline void gz_stg_u16 (uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static inline double gz_rcp_f64 (double x) { return (double)(1.0f / (float)x) * (1.0 - 3e-8); }   // (a SEED of single precision, like v_rcp_f64: the caller's refinement and correction must hold)
