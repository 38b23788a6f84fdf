// tests/emul/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A minimal CPU stand-in for the slice of the HIP runtime + device intrinsics that genozip_amd/csrc uses, so that
// the UNMODIFIED product sources (gz_host.cpp + gz_kernels_*.h) can be compiled with g++ into
// tests/emul/libgenozip_amd_emul.so and exercised by the CPU test-suite (pytest -m "not gpu") in a container that
// has no GPU. It exists to catch logic errors before GPU time is spent; it is not shipped, the genozip_amd package
// never loads it, and it says nothing about performance.
//
// Execution model: blocks run one after another; the threads of a block are cooperative fibers. __syncthreads()
// parks a fiber until every live thread of the block has arrived, __ballot() until every live lane of its 64-wide
// wave has. A thread that returns from the kernel drops out of both (like a terminated wave on the hardware);
// threads stuck at different barriers are reported as a deadlock.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>
#include <functional>
#include <memory>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __shared__
#define __forceinline__ inline
#define __launch_bounds__(...)

struct dim3 { unsigned x, y, z; dim3 (unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x (x_), y (y_), z (z_) {} };
struct emu_uint3 { unsigned x, y, z; };

typedef int hipError_t;
typedef void *hipStream_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1 };

static inline const char *hipGetErrorString (hipError_t) { return "emulated"; }
static inline hipError_t hipGetDeviceCount (int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice (int) { return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags (hipStream_t *s, int) { *s = (void *)1; return hipSuccess; }
enum { hipDeviceAttributeMultiprocessorCount = 1 };
static inline hipError_t hipDeviceGetAttribute (int *v, int, int) { *v = 256; return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange (int *lo, int *hi) { *lo = 0; *hi = 0; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority (hipStream_t *s, int, int) { *s = (void *)1; return hipSuccess; }
static inline hipError_t hipStreamDestroy (hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize (hipStream_t) { return hipSuccess; }
static inline hipError_t hipGetLastError (void) { return hipSuccess; }
typedef void *hipEvent_t;
static inline hipError_t hipEventCreate (hipEvent_t *e) { *e = nullptr; return hipSuccess; }
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags (hipEvent_t *e, int) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent (hipStream_t, hipEvent_t, int) { return hipSuccess; }
static inline hipError_t hipEventDestroy (hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord (hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize (hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime (float *ms, hipEvent_t, hipEvent_t) { *ms = 0; return hipSuccess; }
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute (const void *, int, int) { return hipSuccess; }
static inline hipError_t hipMalloc (void **p, size_t n)
{
    size_t sz = (n + 255) & ~(size_t)255;
    *p = aligned_alloc (256, sz ? sz : 256);
    if (*p) memset (*p, 0xA5, sz ? sz : 256);   // poison: device memory is not zeroed
    return *p ? hipSuccess : hipErrorUnknown;
}
static inline hipError_t hipFree (void *p) { free (p); return hipSuccess; }
static inline hipError_t hipMemcpy (void *d, const void *s, size_t n, hipMemcpyKind) { memmove (d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync (void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove (d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync (void *d, int v, size_t n, hipStream_t) { memset (d, v, n); return hipSuccess; }
static inline hipError_t hipMemset (void *d, int v, size_t n) { memset (d, v, n); return hipSuccess; }
enum { hipHostMallocDefault = 0 };
static inline hipError_t hipHostMalloc (void **p, size_t n, unsigned) { *p = malloc (n); return *p ? hipSuccess : (hipError_t)2; }
static inline hipError_t hipHostFree (void *p) { free (p); return hipSuccess; }

// ---- execution context: the threads of a block are cooperative fibers on the calling OS thread ----
// (hand-rolled x86-64 context switch: a wave ballot costs 64 switches of a few ns, where OS threads + barriers cost
// a millisecond). A fiber runs until it reaches __syncthreads() / __ballot() or returns from the kernel.
static emu_uint3 threadIdx, blockIdx;
static dim3 blockDim, gridDim;

alignas(16) uint8_t gz_lds[163840];     // the dynamic LDS region of the block that is currently running

extern "C" void emu_switch (void **save_sp, void *load_sp);
__asm__ (
    ".text\n.globl emu_switch\n.type emu_switch,@function\nemu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size emu_switch,.-emu_switch\n");

enum { EMU_RUN = 0, EMU_WAIT_BLOCK = 1, EMU_WAIT_WAVE = 2, EMU_DONE = 3 };
struct EmuFiber { void *sp; int state; unsigned shfl_parity; };
struct EmuWave { unsigned alive, arrived; unsigned long long result; uint8_t slot[64]; };
struct EmuSched {
    static const unsigned MAXT = 1024;
    static const size_t STACK = 128 << 10;
    uint8_t *stacks = nullptr;
    EmuFiber f[MAXT];
    EmuWave wave[MAXT / 64];
    void *main_sp = nullptr;
    unsigned n = 0, alive = 0, blk_arrived = 0, cur = 0;
    dim3 block;
    const std::function<void ()> *body = nullptr;
};
static EmuSched emu;

static inline void emu_set_tid (unsigned t)
{
    threadIdx.x = t % emu.block.x; threadIdx.y = (t / emu.block.x) % emu.block.y; threadIdx.z = t / (emu.block.x * emu.block.y);
}
static inline void emu_yield (void) { unsigned me = emu.cur; emu_switch (&emu.f[me].sp, emu.main_sp); }
static inline void emu_release_block (void)
{
    for (unsigned i = 0; i < emu.n; i++) if (emu.f[i].state == EMU_WAIT_BLOCK) emu.f[i].state = EMU_RUN;
    emu.blk_arrived = 0;
}
static inline void emu_release_wave (unsigned w)
{
    EmuWave &W = emu.wave[w];
    unsigned long long m = 0;
    for (unsigned l = 0; l < 64; l++) {
        unsigned t = w * 64 + l;
        if (t < emu.n && emu.f[t].state == EMU_WAIT_WAVE) { if (W.slot[l]) m |= 1ull << l; emu.f[t].state = EMU_RUN; }
    }
    W.result = m; W.arrived = 0;
}
static void emu_fiber_entry (void)
{
    (*emu.body) ();
    unsigned me = emu.cur, w = me / 64;
    emu.f[me].state = EMU_DONE;
    emu.alive--;                                   // a finished thread no longer takes part in barriers
    emu.wave[w].alive--;
    if (emu.alive && emu.blk_arrived == emu.alive) emu_release_block ();
    if (emu.wave[w].alive && emu.wave[w].arrived == emu.wave[w].alive) emu_release_wave (w);
    emu_yield ();
    abort ();
}

static inline void emu_launch (dim3 grid, dim3 block, size_t shmem, const std::function<void ()> &body)
{
    const unsigned nthreads = block.x * block.y * block.z;
    if (shmem > sizeof (gz_lds) || nthreads > EmuSched::MAXT) abort ();
    if (!emu.stacks) emu.stacks = (uint8_t *)aligned_alloc (4096, EmuSched::MAXT * EmuSched::STACK);
    for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
    for (unsigned bx = 0; bx < grid.x; bx++) {
        emu.n = emu.alive = nthreads; emu.blk_arrived = 0; emu.block = block; emu.body = &body;
        blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz; blockDim = block; gridDim = grid;
        for (unsigned w = 0; w < (nthreads + 63) / 64; w++) {
            emu.wave[w].alive = nthreads - w * 64 < 64 ? nthreads - w * 64 : 64;
            emu.wave[w].arrived = 0; memset (emu.wave[w].slot, 0, 64);
        }
        for (unsigned t = 0; t < nthreads; t++) {
            void **top = (void **)(emu.stacks + (size_t)(t + 1) * EmuSched::STACK);   // 16-byte aligned
            top[-1] = nullptr;                          // fake return address of the entry function
            top[-2] = (void *)emu_fiber_entry;          // `ret` of the first switch lands here
            for (int k = 3; k <= 8; k++) top[-k] = nullptr;   // rbp rbx r12 r13 r14 r15
            emu.f[t].sp = (void *)(top - 8);
            emu.f[t].state = EMU_RUN; emu.f[t].shfl_parity = 0;
        }
        memset (gz_lds, 0x5A, shmem + 64 < sizeof (gz_lds) ? shmem + 64 : sizeof (gz_lds));   // LDS is not zeroed
        for (;;) {
            bool progressed = false;
            for (unsigned t = 0; t < nthreads; t++) {
                if (emu.f[t].state != EMU_RUN) continue;
                progressed = true;
                emu.cur = t; emu_set_tid (t);
                emu_switch (&emu.main_sp, emu.f[t].sp);
            }
            if (!emu.alive) break;
            if (!progressed) { fprintf (stderr, "hip emulator: deadlock - threads of a block wait at different barriers\n"); abort (); }
        }
    }
}

// EMU_PROFILE=1: seconds per kernel name on stderr when the process ends (where does an emulated test spend its time?)
#include <map>
#include <string>
#include <chrono>
struct EmuProfile {
    std::map<std::string, std::pair<double, unsigned long>> t; std::map<std::string, unsigned long long> thr; bool on;
    EmuProfile () { const char *e = getenv ("EMU_PROFILE"); on = e && *e && *e != '0'; }
    ~EmuProfile () { if (on) for (auto &k : t) fprintf (stderr, "[emu] %-28s %9.3f s  %8lu launches %12llu threads\n", k.first.c_str (), k.second.first, k.second.second, thr[k.first]); }
};
static EmuProfile emu_profile;
static inline void emu_launch_named (const char *name, dim3 grid, dim3 block, size_t shmem, const std::function<void ()> &body)
{
    if (!emu_profile.on) { emu_launch (grid, block, shmem, body); return; }
    const auto t0 = std::chrono::steady_clock::now ();
    emu_launch (grid, block, shmem, body);
    auto &e = emu_profile.t[name];
    e.first += std::chrono::duration<double> (std::chrono::steady_clock::now () - t0).count (); e.second++;
    emu_profile.thr[name] += (unsigned long long)grid.x * grid.y * grid.z * block.x * block.y * block.z;
}
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    emu_launch_named (#kern, (grid), (block), (shmem), [=] () { kern (__VA_ARGS__); })

// ---- device intrinsics ----
static inline void __syncthreads (void)
{
    unsigned me = emu.cur;
    emu.f[me].state = EMU_WAIT_BLOCK;
    if (++emu.blk_arrived == emu.alive) emu_release_block ();
    emu_yield ();
    emu_set_tid (me);
}
static inline void __threadfence_block (void) {}
// (the same value in every lane of a wave at the same point of the program, as on the device: the clock only ever picks between ways that
//  give the same result - k_arith_model's eventful batches)
static inline long long wall_clock64 (void) { return 0; }
// doubles by their two words
static inline double __hiloint2double (int hi, int lo) { uint64_t b = (uint64_t)(uint32_t)hi << 32 | (uint32_t)lo; double d; memcpy (&d, &b, 8); return d; }
static inline int __double2hiint (double d) { uint64_t b; memcpy (&b, &d, 8); return (int)(uint32_t)(b >> 32); }
static inline int __double2loint (double d) { uint64_t b; memcpy (&b, &d, 8); return (int)(uint32_t)b; }
static inline void __threadfence (void) {}

static inline unsigned long long __ballot (int pred)
{
    unsigned me = emu.cur, w = me / 64;
    EmuWave &W = emu.wave[w];
    W.slot[me % 64] = pred ? 1 : 0;
    emu.f[me].state = EMU_WAIT_WAVE;
    if (++W.arrived == W.alive) emu_release_wave (w);
    emu_yield ();
    emu_set_tid (me);
    return W.result;
}

// wave-wide exchange: every live lane deposits a value, then reads the source lane's
// (two sets of slots taking turns: a lane that is through one exchange deposits for the next in the other set, and cannot get to the
//  one after that - the first set again - before every lane has arrived at the next, i.e. has read this one: ONE barrier per exchange)
static inline int emu_shfl (int v, int src)
{
    unsigned me = emu.cur, w = me / 64;
    static int slots[2][EmuSched::MAXT];
    const unsigned par = emu.f[me].shfl_parity; emu.f[me].shfl_parity ^= 1;
    slots[par][me] = v;
    (void)__ballot (0);                     // all lanes have deposited
    return slots[par][w * 64 + (src & 63)];
}
#define __builtin_amdgcn_readlane(v, lane) emu_shfl ((v), (lane))
#define __shfl(v, lane) emu_shfl ((v), (lane))
#define __builtin_amdgcn_readfirstlane(v) emu_shfl ((v), 0)
static inline int __ffsll (unsigned long long v) { return __builtin_ffsll ((long long)v); }
struct uint2 { unsigned x, y; };
static inline uint2 make_uint2 (unsigned x, unsigned y) { uint2 r; r.x = x; r.y = y; return r; }
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4 (unsigned x, unsigned y, unsigned z, unsigned w) { uint4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

static inline int __clz (unsigned v) { return v ? __builtin_clz (v) : 32; }
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) emu_yield ()          /* lets the other threads of the block run */
// this runtime runs every launch to completion at enqueue time: a kernel that waits for a later launch must be queued after it
#define GZ_SEQUENTIAL_STREAMS 1
static inline int __popcll (unsigned long long v) { return __builtin_popcountll (v); }
static inline int __popc (unsigned v) { return __builtin_popcount (v); }
static inline unsigned __umulhi (unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline long long __double_as_longlong (double d) { long long v; memcpy (&v, &d, 8); return v; }
static inline double __longlong_as_double (long long v) { double d; memcpy (&d, &v, 8); return d; }
static inline unsigned atomicAdd (unsigned *p, unsigned v) { return __atomic_fetch_add (p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicOr (unsigned *p, unsigned v) { return __atomic_fetch_or (p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicOr (unsigned long long *p, unsigned long long v) { return __atomic_fetch_or (p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicCAS (unsigned *p, unsigned expect, unsigned v)
{
    __atomic_compare_exchange_n (p, &expect, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
    return expect;
}
static inline unsigned atomicMin (unsigned *p, unsigned v)
{
    unsigned old = __atomic_load_n (p, __ATOMIC_RELAXED);
    while (old > v && !__atomic_compare_exchange_n (p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline unsigned atomicMax (unsigned *p, unsigned v)
{
    unsigned old = __atomic_load_n (p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n (p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
