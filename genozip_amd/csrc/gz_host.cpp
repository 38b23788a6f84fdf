// gz_host.cpp -- host orchestrator + C-ABI of libgenozip_amd.so (compiled by hipcc together with the kernels).
//
// The reference runs one VBlock per pthread and calls codec_args[codec].compress once per section
// (src/zip.c:510-601, src/compressor.c:18). Here the host only PLANS: it turns a table of streams (sections) into
// a table of leaves (independent entropy-coding jobs), carves their scratch out of one HBM arena, uploads both
// tables and queues ~10 kernels on one HIP stream; every kernel covers all leaves of the whole batch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <atomic>
#include <vector>
#include <algorithm>
#include <map>
#include <chrono>
#include <string>

#include "../../include/genozip_amd.h"
#include "gz_device.h"
#include "gz_devutil.h"
#include "gz_kernels_enc.h"
#include "gz_kernels_arith.h"
#include "gz_kernels_dec.h"
#include "gz_kernels_ctx.h"
#include "gz_kernels_seg.h"
#include "gz_merge.h"
#include "gz_kernels_zip.h"
#include "gz_kernels_domq.h"
#include "gz_kernels_bam.h"

#define GZ_VERSION "genozip_amd 0.1 (gfx950; format parity: genozip 15.0.86)"

// ---------------------------------------------------------------------------------------------------------
// handle, arena
// ---------------------------------------------------------------------------------------------------------
struct Pending {          // results to hand back at gz_sync()
    int kind;             // 0 compress batch, 1 uncompress batch, 2 vb batch
    void *user; int n;
    void *dev_streams;    // GzdStream* / GzdDecStream*
    void *dev_vbs;        // GzdVB*
    size_t n_dev_streams;
    struct GzHandle *emit_after = NULL;   // (kind 2) the handle whose queued work this batch's section writer waited for: a replay waits for it again
};

struct ArenaBlock { uint8_t *base; size_t size, used; };

// More hardware queues than the HIP runtime's default of 4 (see genozip_amd/lib.py): only effective when this library is
// loaded before the process's first HIP call; never overrides what the host has set.
__attribute__((constructor)) static void gz_runtime_env (void) { setenv ("GPU_MAX_HW_QUEUES", "8", 0); }

// workgroups of persistent chain kernels in flight in this process / those of them that hold a whole compute unit
static std::atomic<int> g_chain_wgs (0), g_chain_cus (0);

#define GZ_MAX_CHUNKS 129    // position chunks of the arithmetic coder's pipeline: at most 16 - or GZ_ARITH_CHUNKS - (arith_pipe_setup) + a partial one
struct GzHandle {
    int device;
    hipStream_t stream;
    hipStream_t stream2;      // side stream: work that is independent of the arithmetic coder's long chain runs beside it
    hipEvent_t ev_fork, ev_join;
    hipStream_t stream3;      // the range coder chain of position chunk k runs here, beside the models of chunk k+1
    hipStream_t stream4;      // the models: like stream2 kept off the compute units reserved for the chain
    hipEvent_t ev_model_fork;
    hipStream_t stream5;      // model + chain of the leaves that fit one chunk
    hipEvent_t ev_small;
    hipStream_t stream6;      // the `low` kernels of the long leaves, following the chain chunk by chunk
    hipEvent_t ev_low;
    hipStream_t stream7;      // the context sort of position chunk k+1, beside the models of chunk k
    hipEvent_t ev_sort[GZ_MAX_CHUNKS];
    hipEvent_t ev_chain_go, ev_chain;   // the persistent chain may start / has finished
    int n_cu = 0;             // compute units of the device
    int chain_wgs_held = 0, chain_cus_held = 0;   // this handle's share of g_chain_wgs / g_chain_cus, returned at gz_sync
    uint32_t *d_fail = NULL;  // set by a kernel that gave up (the persistent chain when the models never report)
    GzHandle *emit_after = NULL;   // the next VBlock batch's section writer waits for this handle's queued work (gz_emit_after)
    GzHandle *emit_after_used = NULL;   // what the batch being queued right now has consumed of it
    hipEvent_t ev_other = NULL;
    std::string warn;         // gz_last_warning: a call that succeeded has something to say (the chain fallback)
    uint32_t tm_dbg = 0;
    bool tile_models = false; // GZ_MODEL_TILED=1: the leaves of small alphabets through k_arith_model_tiled (one workgroup per leaf, records leave coalesced) - exact, less traffic, SLOWER: off
    bool no_pipeline = false; // GZ_NO_PIPELINE=1: no persistent kernel (needed under tools that serialise kernels, e.g. rocprofv3 --pmc)
    bool in_fallback = false; // gz_sync is running a batch again, unpipelined, after its persistent chain never heard from the models
    uint32_t chain_fallbacks = 0;           // how often that has happened on this handle (gz_chain_fallbacks)
    bool debug_starve_chain = false;        // GZ_DEBUG_STARVE_CHAIN=1 (tests): the models' progress is never announced - the chain must time out, the batch must still come out right
    bool own_stream;
    std::vector<ArenaBlock> blocks;
    std::vector<Pending> pending;
    std::vector<void *> host_tmp;      // host staging to free at sync
    GzLogTable *d_logs;
    std::string err;
    size_t arena_block_size;
    // optional per-kernel timing with HIP events on this handle's stream (bench.py's roofline object)
    int profiling = 0;        // gz_profile: 0 off, 1 every kernel launch between two events, 2 only the two kernels a step can be as long as (k_arith_chain, k_arith_model)
    struct ProfRec { const char *name; hipEvent_t a, b; };
    std::vector<ProfRec> prof_open;
    std::vector<hipEvent_t> event_pool;            // (creating and destroying two events per launch costs more than the launch)
    struct ProfAcc { std::string name; double ms; int launches; double max_ms; };
    std::vector<ProfAcc> prof;
    bool background = false;               // gz_create_background
    std::vector<GzHandle *> helpers;       // handles that work for this one (the VBlock driver's second handle): profiled with it
    uint32_t zip_spec_hits = 0, zip_spec_misses = 0;
    // predicted coding (gz_zip.h): sections coded ahead of their context's trial with a predicted codec - kept / coded again; the codec every
    // (dict_id, local | b250) of the handle's previous files ended up with (the prediction of a warm handle)
    uint32_t zip_pred_hits = 0, zip_pred_misses = 0;
    std::map<std::pair<uint64_t, int>, int> zip_codec_memory;
    // the VBlock driver: the coder the long QUAL streams are started with before the file's own trial compressions are through
    // (plain / through CODEC_DOMQ) - gz_zip.h, "speculation". Starts as a built-in prior - an order-1 adaptive coder is what
    // codec_assign_best_codec's size rule gives quality strings - and follows what the handle's files actually got
    int zip_qual_guess[2] = { GZ_CODEC_ARTB, 0 };
    uint32_t debug_chain_fault = 0;        // GZ_DEBUG_CHAIN_FAULT=k (tests): k_chain_expand treats slice k - 1 of every arithmetic leaf as a checkpoint mismatch
    std::vector<ProfAcc> prof_view;        // gz_profile_get: this handle's totals + its helpers'
};

static inline hipEvent_t gz_event_get (GzHandle *h)
{
    if (!h->event_pool.empty ()) { hipEvent_t e = h->event_pool.back (); h->event_pool.pop_back (); return e; }
    hipEvent_t e = NULL;
    (void)hipEventCreate (&e);
    return e;
}

// KLAUNCH: hipLaunchKernelGGL bracketed by two events when profiling is on
// (mode 2: two events per launch are 4 - 5 us of host time, ~700 launches a step - as much as the launches themselves; the heavy kernels are a few dozen)
#define GZ_PROF_HEAVY(name) (!__builtin_strcmp (name, "k_arith_chain") || !__builtin_strncmp (name, "k_arith_model", 13))
#define KLAUNCH_ON(h, strm, kern, grid, block, shmem, ...) do { \
    GzHandle::ProfRec pr_; pr_.name = #kern; \
    const bool prof_ = (h)->profiling == 1 || ((h)->profiling == 2 && GZ_PROF_HEAVY (#kern)); \
    if (prof_) { pr_.a = gz_event_get (h); pr_.b = gz_event_get (h); (void)hipEventRecord (pr_.a, (strm)); } \
    hipLaunchKernelGGL (kern, grid, block, shmem, (strm), __VA_ARGS__); \
    if (prof_) { (void)hipEventRecord (pr_.b, (strm)); (h)->prof_open.push_back (pr_); } } while (0)
#define KLAUNCH(h, kern, grid, block, shmem, ...) KLAUNCH_ON (h, (h)->stream, kern, grid, block, shmem, __VA_ARGS__)

#define HIPCHK(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    (h)->err = std::string (#call) + ": " + hipGetErrorString (e_); return GZ_ERR_HIP; } } while (0)

static const size_t ARENA_ALIGN = 256;

static void *arena_alloc (GzHandle *h, size_t bytes)
{
    bytes = (bytes + ARENA_ALIGN - 1) & ~(ARENA_ALIGN - 1);
    if (!bytes) bytes = ARENA_ALIGN;
    for (auto &b : h->blocks)
        if (b.size - b.used >= bytes) { void *p = b.base + b.used; b.used += bytes; return p; }
    size_t sz = h->arena_block_size;
    while (sz < bytes) sz *= 2;
    ArenaBlock nb; nb.size = sz; nb.used = bytes;
    if (hipMalloc ((void **)&nb.base, sz) != hipSuccess) {
        // fall back to an exact-size block before giving up
        nb.size = bytes;
        if (hipMalloc ((void **)&nb.base, bytes) != hipSuccess) { h->err = "hipMalloc failed (arena)"; return NULL; }
    }
    h->blocks.push_back (nb);
    return nb.base;
}

static void arena_reset (GzHandle *h)
{
    // keep the blocks (the steady state re-uses them), just rewind
    for (auto &b : h->blocks) b.used = 0;
}

extern "C" const char *gz_version (void) { return GZ_VERSION; }

static GzHandle *gz_create_do (int device, void *hip_stream, int *err, bool background);
extern "C" GzHandle *gz_create (int device, void *hip_stream, int *err) { return gz_create_do (device, hip_stream, err, false); }
// a handle for long-running work that must not hold up another handle's short kernels: all its streams but the chain's at the
// lowest priority (the VBlock compute driver codes the long QUAL streams on such a handle)
extern "C" GzHandle *gz_create_background (int device, int *err) { return gz_create_do (device, NULL, err, true); }

static GzHandle *gz_create_do (int device, void *hip_stream, int *err, bool background)
{
    int ndev = 0;
    if (hipGetDeviceCount (&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        if (err) *err = GZ_ERR_NO_DEVICE;   // no CPU fallback: the caller must have a GPU
        return NULL;
    }
    if (hipSetDevice (device) != hipSuccess) { if (err) *err = GZ_ERR_HIP; return NULL; }
    GzHandle *h = new GzHandle ();
    h->device = device; h->d_logs = NULL; h->background = background;
    h->arena_block_size = (size_t)256 << 20;
    int prio_lo0 = 0, prio_hi0 = 0;
    if (hipDeviceGetStreamPriorityRange (&prio_lo0, &prio_hi0) != hipSuccess) prio_lo0 = prio_hi0 = 0;
    if (hip_stream) { h->stream = (hipStream_t)hip_stream; h->own_stream = false; }
    else {
        if (hipStreamCreateWithPriority (&h->stream, hipStreamNonBlocking, background ? prio_lo0 : (prio_lo0 + prio_hi0) / 2) != hipSuccess) { delete h; if (err) *err = GZ_ERR_HIP; return NULL; }
        h->own_stream = true;
    }
    // The chain (one wave per leaf, strictly serial) is the critical path: its few workgroups must not queue behind
    // the tens of thousands of the model kernel launched at the same moment, so its stream gets the highest priority and
    // the streams of the kernels that run beside it the lowest.
    if (hipDeviceGetAttribute (&h->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || h->n_cu <= 0) h->n_cu = 64;
    int prio_lo = 0, prio_hi = 0;
    if (hipDeviceGetStreamPriorityRange (&prio_lo, &prio_hi) != hipSuccess) prio_lo = prio_hi = 0;
    if (!background) prio_lo = (prio_lo + prio_hi) / 2;      // an ordinary handle's side streams: the middle; a background handle's: the lowest
    if (hipStreamCreateWithPriority (&h->stream2, hipStreamNonBlocking, prio_lo) != hipSuccess ||
        hipStreamCreateWithPriority (&h->stream3, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipStreamCreateWithPriority (&h->stream4, hipStreamNonBlocking, prio_lo) != hipSuccess ||
        hipStreamCreateWithPriority (&h->stream5, hipStreamNonBlocking, prio_lo) != hipSuccess ||
        hipStreamCreateWithPriority (&h->stream6, hipStreamNonBlocking, prio_lo) != hipSuccess ||
        hipStreamCreateWithPriority (&h->stream7, hipStreamNonBlocking, prio_lo) != hipSuccess ||
        hipEventCreateWithFlags (&h->ev_low, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags (&h->ev_small, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags (&h->ev_model_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags (&h->ev_chain, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags (&h->ev_chain_go, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags (&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags (&h->ev_join, hipEventDisableTiming) != hipSuccess) { delete h; if (err) *err = GZ_ERR_HIP; return NULL; }
    for (int k = 0; k < GZ_MAX_CHUNKS; k++)
        if (hipEventCreateWithFlags (&h->ev_sort[k], hipEventDisableTiming) != hipSuccess) { delete h; if (err) *err = GZ_ERR_HIP; return NULL; }
    // log(1024+k), log(4096+k) from the host libm: what the reference's compute_shift() sees (rANS_static4x16pr.c:647)
    GzLogTable lt;
    for (int k = 0; k <= 256; k++) { lt.l10[k] = log (1024.0 + k); lt.l12[k] = log (4096.0 + k); }
    if (hipMalloc ((void **)&h->d_logs, sizeof (lt)) != hipSuccess ||
        hipMemcpy (h->d_logs, &lt, sizeof (lt), hipMemcpyHostToDevice) != hipSuccess) {
        if (h->own_stream) (void)hipStreamDestroy (h->stream);
        delete h; if (err) *err = GZ_ERR_HIP; return NULL;
    }
    if (hipMalloc ((void **)&h->d_fail, 64) != hipSuccess || hipMemset (h->d_fail, 0, 64) != hipSuccess) { if (err) *err = GZ_ERR_HIP; gz_destroy (h); return NULL; }
    { const char *e = getenv ("GZ_NO_PIPELINE"); h->no_pipeline = e && *e && *e != '0'; }
    { const char *e = getenv ("GZ_MODEL_TILED"); h->tile_models = e && *e && *e != '0'; }
    { const char *e = getenv ("GZ_TM_DEBUG"); h->tm_dbg = e ? (uint32_t)atoi (e) : 0u; }
    { const char *e = getenv ("GZ_DEBUG_STARVE_CHAIN"); h->debug_starve_chain = e && *e && *e != '0'; }
    { const char *e = getenv ("GZ_DEBUG_CHAIN_FAULT"); h->debug_chain_fault = e ? (uint32_t)strtoul (e, NULL, 10) : 0; }   // (tests: a forced checkpoint mismatch must fail the stream)
    // the largest LDS class of the arithmetic coder needs more than the default 64 KB of dynamic LDS
    if (hipFuncSetAttribute ((const void *)k_arith_chain, hipFuncAttributeMaxDynamicSharedMemorySize, 163840) != hipSuccess ||
        hipFuncSetAttribute ((const void *)k_arith_model_tiled, hipFuncAttributeMaxDynamicSharedMemorySize, 163840) != hipSuccess ||
        hipFuncSetAttribute ((const void *)k_arith_decode, hipFuncAttributeMaxDynamicSharedMemorySize, 163840) != hipSuccess) {
        if (err) *err = GZ_ERR_HIP;
        gz_destroy (h);
        return NULL;
    }
    if (err) *err = GZ_OK;
    return h;
}

static void prof_collect (GzHandle *h);

extern "C" void gz_destroy (GzHandle *h)
{
    if (h) { g_chain_wgs.fetch_sub (h->chain_wgs_held); g_chain_cus.fetch_sub (h->chain_cus_held); h->chain_wgs_held = h->chain_cus_held = 0; }
    if (!h) return;
    (void)hipSetDevice (h->device);
    (void)hipStreamSynchronize (h->stream);
    prof_collect (h);
    for (auto e : h->event_pool) (void)hipEventDestroy (e);
    for (auto &b : h->blocks) (void)hipFree (b.base);
    for (auto p : h->host_tmp) free (p);
    (void)hipFree (h->d_logs);
    (void)hipFree (h->d_fail);
    if (h->own_stream) (void)hipStreamDestroy (h->stream);
    (void)hipStreamDestroy (h->stream2);
    (void)hipStreamDestroy (h->stream3);
    (void)hipStreamDestroy (h->stream4);
    (void)hipStreamDestroy (h->stream5);
    (void)hipStreamDestroy (h->stream6);
    (void)hipStreamDestroy (h->stream7);
    for (int k = 0; k < GZ_MAX_CHUNKS; k++) (void)hipEventDestroy (h->ev_sort[k]);
    (void)hipEventDestroy (h->ev_low);
    (void)hipEventDestroy (h->ev_small);
    (void)hipEventDestroy (h->ev_model_fork);
    (void)hipEventDestroy (h->ev_chain);
    (void)hipEventDestroy (h->ev_chain_go);
    (void)hipEventDestroy (h->ev_fork); (void)hipEventDestroy (h->ev_join);
    if (h->ev_other) (void)hipEventDestroy (h->ev_other);
    delete h;
}

// turn the recorded event pairs into per-kernel totals (the kernels have completed: called after a sync). Not done inside
// every gz_sync: ~100 elapsed-time queries per step are measurement work, not part of the step
static void prof_collect (GzHandle *h)
{
    for (auto &pr : h->prof_open) {
        float ms = 0;
        if (hipEventElapsedTime (&ms, pr.a, pr.b) == hipSuccess) {
            bool found = false;
            std::string nm (pr.name);                                  // ("k_arith_model<true>": one kernel under one name)
            const size_t lt = nm.find ('<'); if (lt != std::string::npos) nm.resize (lt);
            for (auto &acc : h->prof) if (acc.name == nm) { acc.ms += ms; acc.launches++; if (ms > acc.max_ms) acc.max_ms = ms; found = true; break; }
            if (!found) { GzHandle::ProfAcc acc; acc.name = nm; acc.ms = ms; acc.launches = 1; acc.max_ms = ms; h->prof.push_back (acc); }
        }
        h->event_pool.push_back (pr.a); h->event_pool.push_back (pr.b);
    }
    h->prof_open.clear ();
}

extern "C" void gz_profile (GzHandle *h, int enable, int reset)
{
    if (!h) return;
    if (h->pending.empty ()) prof_collect (h);          // (everything recorded so far has been synchronised)
    h->profiling = enable == 2 ? 2 : enable != 0;
    if (reset) h->prof.clear ();
    for (GzHandle *o : h->helpers) gz_profile (o, enable, reset);
}

extern "C" int gz_profile_get (GzHandle *h, int idx, char *name, int name_cap, double *total_ms, int *launches)
{
    if (!h || idx < 0) return 0;
    if (idx == 0) {                                         // (a walk starts at 0: gather this handle's and its helpers' totals)
        h->prof_view.clear ();
        std::vector<GzHandle *> all (1, h);
        all.insert (all.end (), h->helpers.begin (), h->helpers.end ());
        for (GzHandle *o : all) {
            if (o->pending.empty ()) prof_collect (o);
            for (auto &p : o->prof) {
                bool found = false;
                for (auto &acc : h->prof_view) if (acc.name == p.name) { acc.ms += p.ms; acc.launches += p.launches; if (p.max_ms > acc.max_ms) acc.max_ms = p.max_ms; found = true; break; }
                if (!found) h->prof_view.push_back (p);
            }
        }
    }
    if (idx >= (int)h->prof_view.size ()) return 0;
    if (name && name_cap > 0) { strncpy (name, h->prof_view[idx].name.c_str (), name_cap - 1); name[name_cap - 1] = 0; }
    if (total_ms) *total_ms = h->prof_view[idx].ms;
    if (launches) *launches = h->prof_view[idx].launches;
    return 1;
}

// the longest single launch of entry idx of the last walk (gz_profile_get from 0): the critical path of a kernel that is launched
// several times side by side on different streams is its longest launch, not the sum
extern "C" int gz_profile_get_max (GzHandle *h, int idx, double *max_ms)
{
    if (!h || idx < 0 || idx >= (int)h->prof_view.size () || !max_ms) return 0;
    *max_ms = h->prof_view[idx].max_ms;
    return 1;
}

// everything queued on h from now on waits until what is queued on `other` so far has completed (no host wait)
extern "C" int gz_wait_for (GzHandle *h, GzHandle *other)
{
    if (!h || !other || h == other || h->device != other->device) return GZ_ERR_ARG;
    HIPCHK (h, hipSetDevice (h->device));
    if (!h->ev_other) HIPCHK (h, hipEventCreateWithFlags (&h->ev_other, hipEventDisableTiming));
    HIPCHK (h, hipEventRecord (h->ev_other, other->stream));
    HIPCHK (h, hipStreamWaitEvent (h->stream, h->ev_other, 0));
    return GZ_OK;
}

extern "C" int gz_emit_after (GzHandle *h, GzHandle *other)
{
    if (!h || !other || h == other || h->device != other->device) return GZ_ERR_ARG;
    h->emit_after = other;
    return GZ_OK;
}

extern "C" int gz_download (GzHandle *h, void *dst, const void *src, uint64_t n)
{
    if (!h || (n && (!dst || !src))) return GZ_ERR_ARG;
    HIPCHK (h, hipSetDevice (h->device));
    HIPCHK (h, hipStreamSynchronize (h->stream));
    if (n) HIPCHK (h, hipMemcpy (dst, src, n, hipMemcpyDeviceToHost));
    return GZ_OK;
}

extern "C" int gz_upload (GzHandle *h, void *dst, const void *src, uint64_t n)
{
    if (!h || (n && (!dst || !src))) return GZ_ERR_ARG;
    HIPCHK (h, hipSetDevice (h->device));
    HIPCHK (h, hipStreamSynchronize (h->stream));
    if (n) HIPCHK (h, hipMemcpy (dst, src, n, hipMemcpyHostToDevice));
    return GZ_OK;
}

extern "C" void *gz_dev_alloc (GzHandle *h, uint64_t n)
{
    void *p = NULL;
    if (!h || hipSetDevice (h->device) != hipSuccess || hipMalloc (&p, n ? n : 1) != hipSuccess) return NULL;
    return p;
}

extern "C" void gz_dev_free (GzHandle *h, void *p) { if (h && p && hipSetDevice (h->device) == hipSuccess) (void)hipFree (p); }

extern "C" const char *gz_last_error (GzHandle *h) { return h ? h->err.c_str () : "no handle"; }
extern "C" const char *gz_last_warning (GzHandle *h) { return h ? h->warn.c_str () : ""; }
extern "C" void *gz_stream (GzHandle *h) { return h ? (void *)h->stream : NULL; }

// ---------------------------------------------------------------------------------------------------------
// bounds (host copies of the reference's arithmetic: rANS_static4x16pr.c:357-369, arith_dynamic.c:74-80,
// codec_htscodecs.c:26-33)
// ---------------------------------------------------------------------------------------------------------
static uint32_t rans_bound (uint32_t size, int order)
{
    int sz = (int)((order == 0 ? 1.05 * size + 257 * 3 + 4 : 1.05 * size + 257 * 257 * 3 + 4 + 257 * 3 + 4)
                   + ((order & GZ_X_PACK) ? 1 : 0) + ((order & GZ_X_RLE) ? 1 + 257 * 3 + 4 : 0) + 20
                   + ((order & GZ_X_STRIPE) ? 1 + 5 * 4 : 0));
    return (uint32_t)(sz + (sz & 1) + 2);
}

static uint32_t arith_bound (uint32_t size, int order)
{
    return (uint32_t)((order == 0 ? 1.05 * size + 257 * 3 + 4 : 1.05 * size + 257 * 257 * 3 + 4 + 257 * 3 + 4)
                      + ((order & GZ_X_PACK) ? 1 : 0) + ((order & GZ_X_RLE) ? 1 + 257 * 3 + 4 : 0) + 5);
}

static bool codec_is_rans  (int c) { return c >= 6 && c <= 9; }
static bool codec_is_arith (int c) { return c >= 16 && c <= 19; }
static bool codec_ok       (int c) { return c == GZ_CODEC_NONE || codec_is_rans (c) || codec_is_arith (c); }

// what the coders themselves compare the capacity with (rANS_static4x16pr.c:1158, arith_dynamic.c:622): Genozip's est_size
// is this + 1 KB (codec_htscodecs.c:26-33), so capacities in [bound, est_size) still succeed in the reference
static uint32_t codec_min_cap (int codec, uint32_t len)
{
    if (codec_is_rans (codec))  return rans_bound  (len, gz_codec_order (codec));
    if (codec_is_arith (codec)) return arith_bound (len, gz_codec_order (codec));
    return len;
}

extern "C" uint32_t gz_codec_est_size (int codec, uint64_t len)
{
    if (codec == GZ_CODEC_NONE) return (uint32_t)len;
    if (codec_is_rans (codec))  return 1024 + rans_bound  ((uint32_t)len, gz_codec_order (codec));
    if (codec_is_arith (codec)) return 1024 + arith_bound ((uint32_t)len, gz_codec_order (codec));
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// planning: streams -> leaves
// ---------------------------------------------------------------------------------------------------------
struct Plan {
    std::vector<GzdStream> streams;
    std::vector<GzdLeaf>   leaves;
    std::vector<GzdLowBlock> low_blocks;   // (leaf, first slice) of every 256-slice workgroup of the k_low_* kernels
    bool any_striped = false, any_rans = false, any_arith = false;
    uint32_t max_in = 0;
    uint32_t max_arith_n = 0;              // largest plain (model/chain) arith leaf
    bool any_arith_o1 = false;
    bool unpacked = false;                 // a stream of 2^24 bytes or more in the batch: the sorted lists keep ranks in an array of their own (positions need all 32 bits); else position << 8 | rank
    std::vector<uint32_t> plain_list, plain_nb, o1_list, rle_list;   // plain (model/chain) arith leaves and their size bounds; those of them that are order-1
};

static bool add_leaf (GzHandle *h, Plan &P, uint32_t stream, int engine, int plane, int method, uint32_t n_bound)
{
    GzdLeaf L;
    memset (&L, 0, sizeof (L));
    L.stream = stream; L.plane = (uint8_t)plane; L.method = (uint8_t)method; L.engine = (uint8_t)engine;
    L.tile_models = h->tile_models ? 1 : 0;
    const bool o1 = method & 1, pack = method & GZ_X_PACK;
    if (pack) { if (!(L.packed = (uint8_t *)arena_alloc (h, (size_t)n_bound + 16))) return false; }
    // rANS stops (-> CAT) once the payload exceeds the input; the arithmetic coder's scalar chain carries no capacity
    // checks, so its area holds the worst case of an adaptive model: 2 bytes per symbol (freq 1 of a total < 2^16)
    L.pay_cap = engine == GZ_ENG_ARITH ? ((method & GZ_X_RLE) ? 4 : 2) * n_bound + 64 : n_bound + 64;
    if (!(L.symlist = (uint8_t *)arena_alloc (h, 256)) || !(L.symrank = (uint16_t *)arena_alloc (h, 512))) return false;
    if (!(L.pay = (uint8_t *)arena_alloc (h, L.pay_cap))) return false;
    if (engine == GZ_ENG_RANS) {
        if (!(L.F    = (uint32_t *)arena_alloc (h, (o1 ? 256 * 256 + 256 : 256) * sizeof (uint32_t)))) return false;
        if (!(L.syms = (GzRansSym *)arena_alloc (h, (o1 ? 256 * 256 : 256) * sizeof (GzRansSym)))) return false;
        if (!(L.tab  = (uint8_t *)arena_alloc (h, o1 ? GZ_TAB_CAP : 1024))) return false;
        if (o1 && !(L.rowbuf = (uint8_t *)arena_alloc (h, 256 * GZ_ROW_SLOT + 256 * sizeof (GzRansSym)))) return false;
    }
    else {
        const bool rle = method & GZ_X_RLE;
        // what the coder sees: the bytes, or (run-length variant) up to 2 coding events per byte
        const uint32_t nb = rle ? 2 * n_bound : n_bound;
        const uint32_t nctx = rle ? 768 : 256;
        if (!(L.triples = (uint8_t *)arena_alloc (h, ((size_t)nb + 64) * GZ_CHAIN_REC + 16384))) return false;
        if (!(L.events  = (uint8_t *)arena_alloc (h, ((size_t)L.pay_cap + 128) * 4))) return false;
        if (!(L.rvals   = (uint8_t *)arena_alloc (h, ((size_t)nb + 64) * 4))) return false;
        const uint32_t ns = nb ? (nb + GZ_LOW_SLICE - 1) / GZ_LOW_SLICE : 1;
        if (!(L.kpos    = (uint8_t *)arena_alloc (h, ((size_t)ns + 2) * 4))) return false;
        if (!(L.kbits   = (uint8_t *)arena_alloc (h, ((size_t)ns + 2) * 16))) return false;
        if (!(L.ckpt    = (uint8_t *)arena_alloc (h, ((size_t)ns + 2) * 8))) return false;
        if (!(L.resid   = (uint8_t *)arena_alloc (h, ((size_t)ns + 1) * 16))) return false;
        for (uint32_t s0 = 0; s0 < ns; s0 += GZ_LOW_SLICES_PER_WG) { GzdLowBlock b; b.leaf = (uint32_t)P.leaves.size (); b.first_slice = s0; P.low_blocks.push_back (b); }
        if (o1 || rle) {
            const size_t nt = ((size_t)nb + GZ_CTX_TILE - 1) / GZ_CTX_TILE;
            if (!(L.spos   = (uint32_t *)arena_alloc (h, ((size_t)nb + 64) * 4))) return false;
            // (a leaf of fewer than 2^24 positions keeps a position's symbol rank in the low byte of its spos entry: no srk, one scattered store)
            L.srk = NULL;                                          // (one format for the whole batch: the model kernel is compiled for either, Plan::unpacked)
            if (P.unpacked && !(L.srk = (uint8_t *)arena_alloc (h, (size_t)nb + 64))) return false;
            if (!(L.ctxoff = (uint32_t *)arena_alloc (h, (nt + 1) * nctx * 4))) return false;
            // one row per position chunk (no chunk is smaller than GZ_CHUNK_MIN - but for the first one, which goes in up to four pieces: a short leaf has few)
            const size_t rows = std::min<size_t> (GZ_MAX_CHUNKS, (size_t)nb / GZ_CHUNK_MIN + 8);
            if (!(L.ctxend = (uint32_t *)arena_alloc (h, rows * nctx * 4))) return false;
            P.any_arith_o1 = true;
            P.o1_list.push_back ((uint32_t)P.leaves.size ());
        }
        if (rle) {
            if (!(L.ev_ctx = (uint16_t *)arena_alloc (h, ((size_t)nb + 64) * 2))) return false;
            if (!(L.ev_sym = (uint8_t *)arena_alloc (h, (size_t)nb + 64))) return false;
            P.rle_list.push_back ((uint32_t)P.leaves.size ());
        }
        if (nb > GZ_CHUNK_MIN && !(L.mstate = (uint32_t *)arena_alloc (h, (size_t)(rle ? 514 : 256) * GZ_MSTATE_WORDS * 64 * 4))) return false;
        static const bool no_succ = getenv ("GZ_NO_SUCC") != NULL;       // (experiments: the leaf's alphabet for every context, as before)
        if (nb > GZ_CHUNK_MIN && o1 && !rle && !no_succ && !(L.succ = (uint64_t *)arena_alloc (h, 8192))) return false;
        if (nb > P.max_arith_n) P.max_arith_n = nb;
        P.plain_list.push_back ((uint32_t)P.leaves.size ());
        P.plain_nb.push_back (nb);
    }
    P.leaves.push_back (L);
    return true;
}

// leaves of one stream. Candidate methods per plane: rANS tries {O1, RLE, PACK, O0} masked by the caller's order
// (rANS_static4x16pr.c:1203-1206); arith tries a fixed list per plane (arith_dynamic.c:684-687).
static bool plan_stream_leaves (GzHandle *h, Plan &P, uint32_t si)
{
    GzdStream &S = P.streams[si];
    const int codec = S.codec_req, order = gz_codec_order (codec);
    S.first_leaf = (uint32_t)P.leaves.size ();
    S.n_leaves = 0; S.whole_leaf = 0;
    if (codec == GZ_CODEC_NONE) return true;
    const int engine = codec_is_rans (codec) ? GZ_ENG_RANS : GZ_ENG_ARITH;
    const uint32_t n = S.in_len;
    (engine == GZ_ENG_RANS ? P.any_rans : P.any_arith) = true;

    if (order & GZ_X_STRIPE) {
        if (n > 20) {
            P.any_striped = true;
            if (!(S.planes = (uint8_t *)arena_alloc (h, (size_t)n + 16))) return false;
            uint32_t plen = n / 4 + 1;
            for (int k = 0; k < 4; k++) {
                if (engine == GZ_ENG_RANS) {
                    static const int m[4] = { 1, GZ_X_RLE, GZ_X_PACK, 0 };
                    for (int j = 0; j < 4; j++)
                        if ((order & m[j]) == m[j]) { if (!add_leaf (h, P, si, engine, k, m[j] | GZ_X_NOSZ, plen)) return false; S.n_leaves++; }
                }
                else {
                    static const int m[4][4] = { { 3, 1, GZ_X_RLE, 0 }, { 2, 1, 0, 0 }, { 2, 1, GZ_X_PACK, 0 }, { 2, 1, GZ_X_PACK, 0 } };
                    for (int j = 1; j <= m[k][0]; j++) {
                        if ((order & 3) == 0 && (m[k][j] & 1)) continue;
                        if (!add_leaf (h, P, si, engine, k, m[k][j] | GZ_X_NOSZ, plen)) return false;
                        S.n_leaves++;
                    }
                }
            }
        }
        // the actual length may be <= 20 (known only on the device when in_len_dev is given): unstriped fallback
        if (n <= 20 || S.in_len_dev) {
            S.whole_leaf = (uint8_t)S.n_leaves;
            if (!add_leaf (h, P, si, engine, 0xff, order & ~GZ_X_STRIPE, n < 20 ? n : 20)) return false;
            S.n_leaves++;
        }
    }
    else {
        S.whole_leaf = 0;
        if (!add_leaf (h, P, si, engine, 0xff, order, n)) return false;
        S.n_leaves = 1;
    }
    return true;
}

static int upload (GzHandle *h, const void *host, size_t bytes, void **dev)
{
    *dev = arena_alloc (h, bytes);
    if (!*dev) return GZ_ERR_HIP;
    if (bytes) {
        // the source vectors die when the planning function returns: stage through a heap copy that lives until sync
        // (page-locked staging was tried: no faster end to end)
        void *stage = malloc (bytes);
        if (!stage) return GZ_ERR;
        memcpy (stage, host, bytes);
        h->host_tmp.push_back (stage);
        HIPCHK (h, hipMemcpyAsync (*dev, stage, bytes, hipMemcpyHostToDevice, h->stream));
    }
    return GZ_OK;
}

// kernels that run beside the pipelined chain ask for this much LDS so that they never fit on a chain's compute unit
#define GZ_KEEP_OFF_LDS 4608                  // (the chain leaves 4096 bytes of a compute unit free)
// the decoder's models by LDS need (gz_kernels_dec.h: a literal row is 128 words per 64 entries): order 0 - 16 KB; order 1 up to 64 symbols
// - 42.5 KB, three to a compute unit; up to 80 symbols (quality scores) - 80 KB, two to a compute unit; up to 128 - 140 KB; beyond: global memory
#define GZ_ARITH_CLASSES 4
static const uint32_t ARITH_CLASS_WORDS[GZ_ARITH_CLASSES + 1] = { 0, 4096, 10880, 20480, 35840 };

// The arithmetic coder's pipeline of one batch (see gz_kernels_arith.h): which leaves, in how many position chunks
struct ArithPipe {
    uint32_t np = 0, no1 = 0, nlb = 0, nlb_small = 0, nbig = 0, nsmall = 0, chunk = 0, n_chunks = 1;
    const uint32_t *d_plain = NULL, *d_o1 = NULL, *d_big = NULL, *d_small = NULL, *d_rle = NULL;
    const GzdLowBlock *d_lb = NULL, *d_lb_small = NULL;
    uint32_t *d_progress = NULL;      // [0] model chunks announced  [16 + k] leaves whose chain is through chunk k
    uint32_t *d_bounds = NULL;        // [n_chunks + 1] where the position chunks start (device copy of bounds)
    std::vector<uint32_t> bounds;     // the first chunk of `chunk` positions goes in pieces: 1/8, 1/8, 1/4, 1/2 of it (whole sort tiles)
    bool pipelined = false, reserve_cu = false;
};

static int arith_pipe_setup (GzHandle *h, Plan &P, ArithPipe &A)
{
    A.np = (uint32_t)P.plain_list.size (); A.no1 = (uint32_t)P.o1_list.size (); A.nlb = (uint32_t)P.low_blocks.size ();
    if (!A.np) return GZ_OK;
    // position chunks: at most 32 per leaf (4 MB VBlocks - 8: 38.0 ms, 12: 37.7, 16: 37.6), none smaller than GZ_CHUNK_MIN, whole sort tiles
    uint32_t want_chunks = 32;                                  // (16 -> 32: the first chunk's models are the lead-in of the long streams; default step 96.2 -> 95.1 ms, streamed 280.8 -> 277.1)
    if (const char *e = getenv ("GZ_ARITH_CHUNKS")) { const int v = atoi (e); if (v >= 1) want_chunks = (uint32_t)std::min (v, GZ_MAX_CHUNKS - 1 - 6); }   // (experiments; the first- / last-chunk splits below add up to 3 bounds each: n_chunks stays within GZ_MAX_CHUNKS - ev_sort[], the ctxend rows)
    // (whole sort tiles AND whole blocks of the chain's loop: what a chunk leaves over goes one symbol at a time, d_chain_slow - with blocks
    //  of 768 symbols - round 5 - and chunks of whole tiles only, 256 symbols of every chunk did; round 6's blocks of 1024 divide the tiles)
    uint32_t unit = GZ_CTX_TILE;
    while (unit % GZ_CHAIN_BLOCK) unit += GZ_CTX_TILE;
    A.chunk = (P.max_arith_n + want_chunks - 1) / want_chunks;
    if (A.chunk < GZ_CHUNK_MIN) A.chunk = GZ_CHUNK_MIN;
    A.chunk = (A.chunk + unit - 1) / unit * unit;
    A.n_chunks = P.max_arith_n ? (P.max_arith_n + A.chunk - 1) / A.chunk : 1;
    // The chain of a long leaf can only start once the sort and the models of its first position chunk are through: the lead-in of the
    // whole step (0.9 ms of the default FASTQ step's 4.4 before the chain has its first records). GZ_ARITH_FIRST_SPLIT=1 lets the first
    // chunk go in pieces of 1/8, 1/8, 1/4 and 1/2 of a chunk. MEASURED WITHOUT EFFECT on the MI355X (round 5; ms per step without / with:
    // default FASTQ 46.1-46.4 / 46.2, the chain's launch 42.2-42.5 / 42.4; binned FASTQ 26.9 / 28.8; streamed 169.5 / 171.6): a chunk costs
    // ~0.2 ms of event hand-overs between the sort's and the models' streams whatever its size, so three more chunks cost what the earlier
    // start buys, and the chain catches up with the small pieces' models at once. Off by default; the chunk bounds stay explicit.
    A.bounds.clear ();
    {
        // (1: eighths; 2: two halves; 4: 1/4, 1/4, 1/2 - a piece must not be coded faster than the next one's sort + models take, ~0.3 ms)
        static const int split = getenv ("GZ_ARITH_FIRST_SPLIT") ? atoi (getenv ("GZ_ARITH_FIRST_SPLIT")) : 0;
        const uint32_t tiles = A.chunk / GZ_CTX_TILE;
        A.bounds.push_back (0);
        if (split && A.n_chunks > 1 && tiles >= 8) {
            const uint32_t cut[3] = { split == 1 ? (tiles + 7) / 8 : split == 4 ? (tiles + 3) / 4 : 0, split == 1 ? (tiles + 3) / 4 : 0, (tiles + 1) / 2 };
            for (int c = 0; c < 3; c++) if (cut[c] * GZ_CTX_TILE > A.bounds.back () && cut[c] < tiles) A.bounds.push_back (cut[c] * GZ_CTX_TILE);
        }
        for (uint32_t k = 1; k < A.n_chunks; k++) A.bounds.push_back (k * A.chunk);
        // The END of the longest leaves, the other way round: what follows the chain's last position chunk - that chunk's k_chain_expand /
        // k_low_scan / k_low_scatter, then resid / norm / carry and the section writer - is the tail of the step, and the first three are
        // proportional to the last chunk. GZ_ARITH_LAST_SPLIT=1 lets the last chunk go in pieces of 1/2, 1/4, 1/8, 1/8. MEASURED WITHOUT EFFECT
        // (round 5; ms per step without / with: default FASTQ 44.15-44.17 / 44.21-44.25, binned 26.3 / 26.0, streamed 157.4 / 161.8, VCF 903 /
        // 903): those kernels are not what the tail consists of. Off by default.
        {
            static const bool last_split = getenv ("GZ_ARITH_LAST_SPLIT") && getenv ("GZ_ARITH_LAST_SPLIT")[0] == '1';
            const uint32_t s0 = (A.n_chunks - 1) * A.chunk, span_tiles = P.max_arith_n > s0 ? (P.max_arith_n - s0) / GZ_CTX_TILE : 0;
            if (last_split && A.n_chunks > 1 && span_tiles >= 8) {
                const uint32_t cut[3] = { span_tiles / 2, span_tiles / 2 + span_tiles / 4, span_tiles / 2 + span_tiles / 4 + span_tiles / 8 };
                for (int c = 0; c < 3; c++) if (s0 + cut[c] * GZ_CTX_TILE > A.bounds.back ()) A.bounds.push_back (s0 + cut[c] * GZ_CTX_TILE);
            }
        }
        A.bounds.push_back (A.n_chunks * A.chunk);
        A.n_chunks = (uint32_t)A.bounds.size () - 1;
    }
    // leaves that fit one chunk go through model and chain in one piece on a stream of their own; only the long ones
    // take the pipeline (their first model chunk is the lead-in of the whole step: keep it free of other work)
    std::vector<uint32_t> big, small;
    for (size_t i = 0; i < P.plain_list.size (); i++) (P.plain_nb[i] > A.chunk ? big : small).push_back (P.plain_list[i]);
    A.nbig = (uint32_t)big.size (); A.nsmall = (uint32_t)small.size ();
    // The persistent chain waits (on the device) for the models, so it must never keep them from running: all its
    // workgroups are resident at once, and they may take at most half the wave slots of the device (4 waves each,
    // 32 slots per compute unit). More long leaves than that: no pipeline, everything in one piece (correct, slower).
    const uint32_t chain_wgs = (A.nbig + GZ_CHAIN_WAVES - 1) / GZ_CHAIN_WAVES;
    // (the budget is per process: several handles - one per host thread, INTEGRATION.md - share the device)
    A.pipelined = false; A.reserve_cu = false;
    // (on a background handle a batch whose longest leaf is under two chunks - the 99 999-byte sample of the QUAL trial, which sits in
    //  front of the long pole - goes in one piece: 6.5 -> 4.5 ms, it gains nothing from the pipeline and pays for its gates. The same
    //  rule on the main handle made the step 9 ms SLOWER: its trial batches then hold up the sections that run beside the long pole)
    if (A.nbig && !h->no_pipeline && !h->in_fallback && (!h->background || P.max_arith_n > 2 * GZ_CHUNK_MIN)) {
        const int wgs = (int)chain_wgs;
        if (g_chain_wgs.fetch_add (wgs) + wgs <= h->n_cu * 4) {
            A.pipelined = true; h->chain_wgs_held += wgs;
            if (g_chain_cus.fetch_add (wgs) + wgs <= h->n_cu / 4) { A.reserve_cu = true; h->chain_cus_held += wgs; }   // a whole compute unit each only while that leaves 3/4 to the rest
            else g_chain_cus.fetch_sub (wgs);
        }
        else g_chain_wgs.fetch_sub (wgs);
    }
    if (!A.pipelined) { A.nbig = 0; small = P.plain_list; big.clear (); A.nsmall = (uint32_t)small.size (); }
    if (getenv ("GZ_DEBUG_PIPE")) {
        uint64_t sum_big = 0; uint32_t mx = 0, mn = 0xffffffffu;
        for (size_t i = 0; i < P.plain_list.size (); i++) if (P.plain_nb[i] > A.chunk) { sum_big += P.plain_nb[i]; mx = std::max (mx, P.plain_nb[i]); mn = std::min (mn, P.plain_nb[i]); }
        std::map<uint32_t, std::pair<uint32_t, uint32_t>> kinds;      // (codec, method, plane) -> count, max nb
        for (size_t i = 0; i < P.plain_list.size (); i++) {
            const GzdLeaf &L = P.leaves[P.plain_list[i]];
            auto &k = kinds[((uint32_t)P.streams[L.stream].codec_req << 16) | ((uint32_t)L.method << 8) | L.plane];
            k.first++; k.second = std::max (k.second, P.plain_nb[i]);
        }
        for (auto &k : kinds) fprintf (stderr, "[pipe]   codec %u method 0x%02x plane %u: %u leaves, nb <= %u\n", k.first >> 16, (k.first >> 8) & 0xff, k.first & 0xff, k.second.first, k.second.second);
#ifdef GZ_MODEL_DEBUG
        for (int pass = 0; pass < 2; pass++) {
            const std::vector<uint32_t> &lst = pass ? small : big;
            for (size_t i = 0; i < lst.size (); i++) { const GzdLeaf &L = P.leaves[lst[i]];
                fprintf (stderr, "[list] %s %zu: codec %u method 0x%02x plane %u stream %u in_len %u\n", pass ? "small" : "big", i, P.streams[L.stream].codec_req, L.method, L.plane, L.stream, P.streams[L.stream].in_len); }
        }
#endif
        fprintf (stderr, "[pipe] bg %d np %u nbig %u nsmall %u max_arith_n %u chunk %u n_chunks %u pipelined %d reserve %d big: min %u max %u sum %llu\n", (int)h->background, A.np, A.nbig, A.nsmall,
                 P.max_arith_n, A.chunk, A.n_chunks, (int)A.pipelined, (int)A.reserve_cu, mn, mx, (unsigned long long)sum_big);
    }
    void *d;
    int rc = upload (h, P.plain_list.data (), P.plain_list.size () * 4, &d); A.d_plain = (const uint32_t *)d;
    if (rc == GZ_OK)             { rc = upload (h, P.low_blocks.data (), P.low_blocks.size () * sizeof (GzdLowBlock), &d); A.d_lb = (const GzdLowBlock *)d; }
    if (rc == GZ_OK && A.nbig)   { rc = upload (h, big.data (), big.size () * 4, &d); A.d_big = (const uint32_t *)d; }
    if (rc == GZ_OK && !P.rle_list.empty ()) { rc = upload (h, P.rle_list.data (), P.rle_list.size () * 4, &d); A.d_rle = (const uint32_t *)d; }
    if (rc == GZ_OK && A.nsmall) { rc = upload (h, small.data (), small.size () * 4, &d); A.d_small = (const uint32_t *)d; }
    if (rc != GZ_OK) return rc;
    if (A.pipelined) {
        if (A.n_chunks > GZ_MAX_CHUNKS) return GZ_ERR;            // (cannot happen: at most 16 / GZ_ARITH_CHUNKS chunks)
        if (!(A.d_progress = (uint32_t *)arena_alloc (h, 1024))) return GZ_ERR_HIP;
        HIPCHK (h, hipMemsetAsync (A.d_progress, 0, 1024, h->stream));
        { void *db; if ((rc = upload (h, A.bounds.data (), A.bounds.size () * 4, &db)) != GZ_OK) return rc; A.d_bounds = (uint32_t *)db; }
        if (A.nsmall) {                                           // the slices of the short leaves only
            std::vector<uint8_t> is_small (P.leaves.size (), 0);
            for (uint32_t l : small) is_small[l] = 1;
            std::vector<GzdLowBlock> lbs;
            for (const GzdLowBlock &b : P.low_blocks) if (is_small[b.leaf]) lbs.push_back (b);
            A.nlb_small = (uint32_t)lbs.size ();
            if (A.nlb_small) { rc = upload (h, lbs.data (), lbs.size () * sizeof (GzdLowBlock), &d); A.d_lb_small = (const GzdLowBlock *)d; }
            if (rc != GZ_OK) return rc;
        }
    }
    return GZ_OK;
}

// the persistent chain of the long leaves: launched before everything else so that it finds free compute units
static int arith_launch_chain (GzHandle *h, const ArithPipe &A, GzdLeaf *d_leaves)
{
    HIPCHK (h, hipEventRecord (h->ev_chain_go, h->stream));                     // (behind the uploads and the memset)
    HIPCHK (h, hipStreamWaitEvent (h->stream3, h->ev_chain_go, 0));
    KLAUNCH_ON (h, h->stream3, k_arith_chain, dim3 ((A.nbig + GZ_CHAIN_WAVES - 1) / GZ_CHAIN_WAVES), dim3 (64 * GZ_CHAIN_WAVES), A.reserve_cu ? GZ_CHAIN_LDS : 64,
                d_leaves, A.d_big, A.nbig, (const uint32_t *)A.d_progress, (const uint32_t *)A.d_bounds, h->d_fail, A.d_progress + 16, A.n_chunks);
    HIPCHK (h, hipEventRecord (h->ev_chain, h->stream3));
    return GZ_OK;
}

static int launch_encode (GzHandle *h, Plan &P, GzdStream *d_streams, GzdLeaf *d_leaves, GzdVB *d_vbs, uint32_t n_vbs, int section_mode)
{
    const uint32_t ns = (uint32_t)P.streams.size (), nl = (uint32_t)P.leaves.size ();
    if (!ns) return GZ_OK;
    ArithPipe A;
    int rc = arith_pipe_setup (h, P, A);
    if (rc != GZ_OK) return rc;
#ifndef GZ_SEQUENTIAL_STREAMS
    if (A.pipelined && (rc = arith_launch_chain (h, A, d_leaves)) != GZ_OK) return rc;
#endif
    KLAUNCH (h, k_resolve, dim3 ((ns + 255) / 256), dim3 (256), 0, d_streams, ns, section_mode);
    if (P.any_striped) {
        uint32_t chunks = (P.max_in / 16 + 255) / 256;
        if (chunks < 1) chunks = 1;
        if (chunks > 64) chunks = 64;
        KLAUNCH (h, k_stripe, dim3 (ns, chunks), dim3 (256), 0, d_streams);
    }
    if (nl) {
        KLAUNCH (h, k_presence, dim3 (nl, GZ_PRES_SLICES), dim3 (256), 1024, d_streams, d_leaves);
        KLAUNCH (h, k_leaf_prep, dim3 (nl), dim3 (256), 4096, d_streams, d_leaves);
        // fork: the rANS leaves and the run-length arith leaves do not depend on the model/chain kernels of the plain
        // arith leaves, so they run beside them on a second stream
        const bool fork = P.any_arith;
        hipStream_t side = fork ? h->stream2 : h->stream;
        if (fork) { HIPCHK (h, hipEventRecord (h->ev_fork, h->stream)); HIPCHK (h, hipStreamWaitEvent (side, h->ev_fork, 0)); }
        if (P.any_rans) {
            KLAUNCH_ON (h, side, k_hist, dim3 (nl, GZ_HIST_CHUNKS), dim3 (256), GZ_HIST_LDS, d_leaves);
            KLAUNCH_ON (h, side, k_rans_table, dim3 (nl), dim3 (256), 20480, d_leaves, (const GzLogTable *)h->d_logs);
            KLAUNCH_ON (h, side, k_rans_encode, dim3 (nl), dim3 (64), GZ_RANS_ENC_LDS, d_leaves);
        }
        if (A.np) {
            const uint32_t grid_y = GZ_MODEL_GRID_Y + (P.rle_list.empty () ? 0 : GZ_MODEL_GRID_RUN);
            if (!P.rle_list.empty ())                              // the run-length variant's coding events (before anything looks at arith_n)
                KLAUNCH (h, k_rle_events, dim3 ((uint32_t)P.rle_list.size ()), dim3 (1024), 256, d_leaves, A.d_rle);
            // sort (group the positions of the order-1 leaves by context), models, chain
            auto sort_chunk = [&] (hipStream_t st, const uint32_t *list, uint32_t n_list, uint32_t p0, uint32_t chunk, uint32_t span, uint32_t row) -> int {
                if (!A.no1 || !span) return GZ_OK;
                const uint32_t tiles = (span + GZ_CTX_TILE - 1) / GZ_CTX_TILE;
                KLAUNCH_ON (h, st, k_ctx_count, GZ_XCD_DIM (n_list, tiles), dim3 (64), GZ_CTX_MAX * 4, d_leaves, list, n_list, p0, chunk);
                KLAUNCH_ON (h, st, k_ctx_scan, dim3 (n_list), dim3 (256), GZ_CTX_MAX * 8, d_leaves, list, p0, chunk, row);
                KLAUNCH_ON (h, st, k_ctx_scatter, GZ_XCD_DIM (n_list, tiles), dim3 (64), GZ_CTX_MAX * 4 + 256, d_leaves, list, n_list, p0, chunk);
                return GZ_OK;
            };
            if (!A.pipelined) {
                if (h->tile_models) KLAUNCH (h, k_arith_model_tiled, dim3 (A.np), dim3 (64 * GZ_TM_WAVES), GZ_TM_LDS, d_leaves, A.d_plain, 0u, 0xffffffffu, h->tm_dbg);
                if ((rc = sort_chunk (h->stream, A.d_plain, A.np, 0u, 0xffffffffu, P.max_arith_n, 0u)) != GZ_OK) return rc;
                if (P.unpacked) KLAUNCH (h, k_arith_model<false>, GZ_XCD_DIM (A.np, grid_y), dim3 (64), GZ_MODEL_LDS, d_leaves, A.d_plain, A.np, 0u, 0xffffffffu, 0u);
                else            KLAUNCH (h, k_arith_model<true>,  GZ_XCD_DIM (A.np, grid_y), dim3 (64), GZ_MODEL_LDS, d_leaves, A.d_plain, A.np, 0u, 0xffffffffu, 0u);
                KLAUNCH (h, k_arith_chain, dim3 ((A.np + GZ_CHAIN_WAVES - 1) / GZ_CHAIN_WAVES), dim3 (64 * GZ_CHAIN_WAVES), 64,
                         d_leaves, A.d_plain, A.np, (const uint32_t *)NULL, (const uint32_t *)NULL, h->d_fail, (uint32_t *)NULL, 0u);
            }
            else {
                HIPCHK (h, hipEventRecord (h->ev_model_fork, h->stream));        // (after k_leaf_prep)
                HIPCHK (h, hipStreamWaitEvent (h->stream4, h->ev_model_fork, 0));
                HIPCHK (h, hipStreamWaitEvent (h->stream7, h->ev_model_fork, 0));
                // the sort of a chunk needs nothing from the models: with many leaves it runs ahead on its own stream.
                // (Tried: the leaves' models as two groups on two streams, each with its own progress counter, so that one group's
                //  launch fills the tail of the other's - no gain (33.8 vs 33.9 ms; 4 M pairs 84.9 vs 80.6): the model kernels
                //  are short of issue slots, not of waves.)
                if (A.n_chunks > GZ_MAX_CHUNKS) return GZ_ERR;
                // (wide alphabets: the symbols that follow each context byte anywhere in the leaf, before the first chunk's models)
                static const uint32_t sort_ahead_min = getenv ("GZ_SORT_AHEAD_MIN") ? (uint32_t)atoi (getenv ("GZ_SORT_AHEAD_MIN")) : 256u;
                hipStream_t sort_stream = A.nbig > sort_ahead_min ? h->stream7 : h->stream4;   // (measured: 702 leaves 84.0 -> 80.6 ms; 176 leaves 33.9 -> 34.2: the sort then only takes compute units from the models)
                if (A.no1) KLAUNCH_ON (h, sort_stream, k_ctx_succ, dim3 (A.nbig, (P.max_arith_n + GZ_SUCC_SPAN - 1) / GZ_SUCC_SPAN), dim3 (256), 8192, d_leaves, A.d_big);
                for (uint32_t k = 0; k < A.n_chunks; k++) {
                    const uint32_t p0 = A.bounds[k], len = A.bounds[k + 1] - p0;
                    if (p0 >= P.max_arith_n) break;
                    const uint32_t span = P.max_arith_n - p0 < len ? P.max_arith_n - p0 : len;
                    // (the leaves of small alphabets: sort, models and records of the chunk in one kernel, in front of the others' sort)
                    if (h->tile_models) KLAUNCH_ON (h, h->stream4, k_arith_model_tiled, dim3 (A.nbig), dim3 (64 * GZ_TM_WAVES), GZ_TM_LDS, d_leaves, A.d_big, p0, len, h->tm_dbg);
                    if ((rc = sort_chunk (sort_stream, A.d_big, A.nbig, p0, len, span, k)) != GZ_OK) return rc;
                    HIPCHK (h, hipEventRecord (h->ev_sort[k], sort_stream));
                    HIPCHK (h, hipStreamWaitEvent (h->stream4, h->ev_sort[k], 0));
                    if (P.unpacked) KLAUNCH_ON (h, h->stream4, k_arith_model<false>, GZ_XCD_DIM (A.nbig, grid_y), dim3 (64), GZ_MODEL_LDS, d_leaves, A.d_big, A.nbig, p0, len, k);
                    else            KLAUNCH_ON (h, h->stream4, k_arith_model<true>,  GZ_XCD_DIM (A.nbig, grid_y), dim3 (64), GZ_MODEL_LDS, d_leaves, A.d_big, A.nbig, p0, len, k);
                    if (!h->debug_starve_chain) hipLaunchKernelGGL (k_arith_progress, dim3 (1), dim3 (1), 0, h->stream4, A.d_progress, k + 1);
                }
#ifdef GZ_SEQUENTIAL_STREAMS
                if ((rc = arith_launch_chain (h, A, d_leaves)) != GZ_OK) return rc;
#endif
                // (the short leaves are queued BEFORE the gates below: streams share hardware queues, a gate spins until the chain is through
                //  its chunk, and whatever is queued behind a gate in the same hardware queue waits with it - one VCF VBlock's 7.5 M-symbol
                //  stripe planes, "short" next to a 30 M-entry b250, started 2.8 s late, after the long chain had finished)
                if (A.nsmall) {
                    HIPCHK (h, hipStreamWaitEvent (h->stream5, h->ev_model_fork, 0));
                    if (h->tile_models) KLAUNCH_ON (h, h->stream5, k_arith_model_tiled, dim3 (A.nsmall), dim3 (64 * GZ_TM_WAVES), GZ_TM_LDS, d_leaves, A.d_small, 0u, 0xffffffffu, h->tm_dbg);
                    if ((rc = sort_chunk (h->stream5, A.d_small, A.nsmall, 0u, 0xffffffffu, A.chunk, 0u)) != GZ_OK) return rc;
                    if (P.unpacked) KLAUNCH_ON (h, h->stream5, k_arith_model<false>, GZ_XCD_DIM (A.nsmall, grid_y), dim3 (64), GZ_MODEL_LDS, d_leaves, A.d_small, A.nsmall, 0u, 0xffffffffu, 0u);
                    else            KLAUNCH_ON (h, h->stream5, k_arith_model<true>,  GZ_XCD_DIM (A.nsmall, grid_y), dim3 (64), GZ_MODEL_LDS, d_leaves, A.d_small, A.nsmall, 0u, 0xffffffffu, 0u);
                    KLAUNCH_ON (h, h->stream5, k_arith_chain, dim3 ((A.nsmall + GZ_CHAIN_WAVES - 1) / GZ_CHAIN_WAVES), dim3 (64 * GZ_CHAIN_WAVES), GZ_KEEP_OFF_LDS,
                                d_leaves, A.d_small, A.nsmall, (const uint32_t *)NULL, (const uint32_t *)NULL, h->d_fail, (uint32_t *)NULL, 0u);
                    if (A.nlb_small) {
                        KLAUNCH_ON (h, h->stream5, k_chain_expand, dim3 (A.nlb_small), dim3 (64), GZ_EXPAND_LDS, d_leaves, A.d_lb_small, (const uint32_t *)NULL, 0u, h->debug_chain_fault);
                        KLAUNCH_ON (h, h->stream5, k_low_scan, dim3 (A.nsmall), dim3 (1024), 8192, d_leaves, A.d_small, 0u, 0xffffffffu);
                        KLAUNCH_ON (h, h->stream5, k_low_scatter, dim3 (A.nlb_small), dim3 (GZ_LOW_WG), GZ_KEEP_OFF_LDS, d_leaves, A.d_lb_small, (const uint32_t *)NULL, 0u);
                    }
                    HIPCHK (h, hipEventRecord (h->ev_small, h->stream5));
                    HIPCHK (h, hipStreamWaitEvent (h->stream, h->ev_small, 0));
                }
                // the `low` kernels of the long leaves follow the chain: a one-thread gate holds their stream until every
                // leaf is through chunk k; count / scan / scatter of that chunk then run beside the chain's next chunk
                HIPCHK (h, hipStreamWaitEvent (h->stream6, h->ev_model_fork, 0));
                for (uint32_t k = 0; k < A.n_chunks; k++) {
                    const uint32_t p0 = A.bounds[k], len = A.bounds[k + 1] - p0;
                    if (p0 >= P.max_arith_n) break;
                    const uint32_t span = P.max_arith_n - p0 < len ? P.max_arith_n - p0 : len;
                    const uint32_t wgs = (span + GZ_LOW_SLICE * GZ_LOW_SLICES_PER_WG - 1) / (GZ_LOW_SLICE * GZ_LOW_SLICES_PER_WG);
                    hipLaunchKernelGGL (k_low_gate, dim3 (1), dim3 (1), 0, h->stream6, (const uint32_t *)(A.d_progress + 16 + k), A.nbig, h->d_fail);
                    KLAUNCH_ON (h, h->stream6, k_chain_expand, dim3 (A.nbig, wgs), dim3 (64), GZ_EXPAND_LDS, d_leaves, (const GzdLowBlock *)NULL, A.d_big, p0, h->debug_chain_fault);
                    KLAUNCH_ON (h, h->stream6, k_low_scan, dim3 (A.nbig), dim3 (1024), 8192, d_leaves, A.d_big, p0, len);
                    KLAUNCH_ON (h, h->stream6, k_low_scatter, dim3 (A.nbig, wgs), dim3 (GZ_LOW_WG), GZ_KEEP_OFF_LDS, d_leaves, (const GzdLowBlock *)NULL, A.d_big, p0);
                }
                HIPCHK (h, hipEventRecord (h->ev_low, h->stream6));
                HIPCHK (h, hipStreamWaitEvent (h->stream, h->ev_chain, 0));
                HIPCHK (h, hipStreamWaitEvent (h->stream, h->ev_low, 0));
            }
            if (!A.pipelined) {
                KLAUNCH (h, k_chain_expand, dim3 (A.nlb), dim3 (64), GZ_EXPAND_LDS, d_leaves, A.d_lb, (const uint32_t *)NULL, 0u, h->debug_chain_fault);
                KLAUNCH (h, k_low_scan, dim3 (A.np), dim3 (1024), 8192, d_leaves, A.d_plain, 0u, 0xffffffffu);
                KLAUNCH (h, k_low_scatter, dim3 (A.nlb), dim3 (GZ_LOW_WG), 4 * 144 * 4, d_leaves, A.d_lb, (const uint32_t *)NULL, 0u);
            }
            // spills into the following slices' digits (they can reach any distance: only once every digit is stored), then bytes
            KLAUNCH (h, k_low_resid, dim3 ((A.nlb + 63) / 64), dim3 (GZ_LOW_WG), 0, d_leaves, A.d_lb, A.nlb);
            {   // (the digits of a leaf are at most 2 per coded byte + the closing ones: pay_cap)
                const uint32_t tile = GZ_NORM_NT * GZ_NORM_PER, ranges = (uint32_t)(((uint64_t)2 * P.max_arith_n + 64 + (uint64_t)tile * GZ_NORM_RANGE - 1) / ((uint64_t)tile * GZ_NORM_RANGE));
                KLAUNCH (h, k_low_norm, dim3 (A.np, ranges ? ranges : 1), dim3 (GZ_NORM_NT), 8192, d_leaves, A.d_plain);
                KLAUNCH (h, k_low_carry, dim3 (A.np), dim3 (64), 0, d_leaves, A.d_plain);
            }
        }
        if (fork) { HIPCHK (h, hipEventRecord (h->ev_join, side)); HIPCHK (h, hipStreamWaitEvent (h->stream, h->ev_join, 0)); }
    }
    if (n_vbs && h->emit_after) {                    // precompressed sections coded on another handle: only the writer waits for them
        if (!h->ev_other) HIPCHK (h, hipEventCreateWithFlags (&h->ev_other, hipEventDisableTiming));
        HIPCHK (h, hipEventRecord (h->ev_other, h->emit_after->stream));
        HIPCHK (h, hipStreamWaitEvent (h->stream, h->ev_other, 0));
        h->emit_after_used = h->emit_after;          // (one-shot for the caller; the batch's Pending record keeps it for a replay)
        h->emit_after = NULL;
    }
    KLAUNCH (h, k_select, dim3 ((ns + 255) / 256), dim3 (256), 0, d_streams, d_leaves, ns);
    if (n_vbs) KLAUNCH (h, k_vb_layout, dim3 ((n_vbs + 63) / 64), dim3 (64), 0, d_vbs, d_streams, n_vbs);
    KLAUNCH (h, k_emit, dim3 (ns, n_vbs ? GZ_EMIT_SLICES : 1), dim3 (256), 4096, d_streams, d_leaves, d_vbs);
    HIPCHK (h, hipGetLastError ());
    return GZ_OK;
}

extern "C" int gz_codec_compress_batch (GzHandle *h, GzStream *streams, int n_streams)
{
    if (!h || (n_streams && !streams) || n_streams < 0) return GZ_ERR_ARG;
    HIPCHK (h, hipSetDevice (h->device));
    Plan P;
    P.streams.resize (n_streams);
    for (int i = 0; i < n_streams; i++) if (streams[i].in_len >= (1u << 24)) P.unpacked = true;
    for (int i = 0; i < n_streams; i++) {
        GzStream &u = streams[i];
        GzdStream &S = P.streams[i];
        memset (&S, 0, sizeof (S));
        u.out_len = 0;
        if (!codec_ok (u.codec)) { u.status = GZ_ERR_ARG; S.status = GZ_ERR_ARG; continue; }
        S.in = u.in; S.in_len = u.in_len; S.in_len_dev = u.in_len_dev; S.out = u.out; S.out_cap = u.out_cap;
        S.codec_req = u.codec; S.vb = -1; S.out_len_dev = u.out_len_dev;
        // the reference's "output buffer too small" test (rANS_static4x16pr.c:1158, arith_dynamic.c:622)
        S.status = u.out_cap < codec_min_cap (u.codec, u.in_len) ? GZ_ST_TOO_SMALL : GZ_ST_PENDING;
        u.status = S.status;
        if (S.status != GZ_ST_PENDING) continue;
        if (u.in_len > P.max_in) P.max_in = u.in_len;
        if (!plan_stream_leaves (h, P, (uint32_t)i)) return GZ_ERR_HIP;
    }
    void *d_streams, *d_leaves;
    int rc;
    static const bool timing = getenv ("GZ_ZIP_TIMING") != NULL;
    const auto tm0 = std::chrono::steady_clock::now ();
    if ((rc = upload (h, P.streams.data (), P.streams.size () * sizeof (GzdStream), &d_streams)) != GZ_OK) return rc;
    if ((rc = upload (h, P.leaves.data (), P.leaves.size () * sizeof (GzdLeaf), &d_leaves)) != GZ_OK) return rc;
    const auto tm1 = std::chrono::steady_clock::now ();
    if ((rc = launch_encode (h, P, (GzdStream *)d_streams, (GzdLeaf *)d_leaves, NULL, 0, 0)) != GZ_OK) return rc;
    if (timing) fprintf (stderr, "[compress_batch bg %d n %d: upload %.2f launch %.2f ms]\n", (int)h->background, n_streams,
                         std::chrono::duration<double, std::milli> (tm1 - tm0).count (), std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now () - tm1).count ());
    Pending pd; pd.kind = 0; pd.user = streams; pd.n = n_streams; pd.dev_streams = d_streams; pd.dev_vbs = NULL; pd.n_dev_streams = P.streams.size ();
    h->pending.push_back (pd);
    return GZ_OK;
}

extern "C" uint64_t gz_vb_z_bound (const GzSection *sections, uint32_t n_sections)
{
    uint64_t b = 84;
    for (uint32_t i = 0; i < n_sections; i++) {
        int codec = sections[i].codec ? sections[i].codec : GZ_CODEC_RANB;
        uint64_t e = sections[i].precompressed ? sections[i].data_len : gz_codec_est_size (codec, sections[i].data_len);
        if (e < sections[i].data_len) e = sections[i].data_len;
        b += 40 + e;
    }
    return b;
}

extern "C" int gz_vb_compress_batch (GzHandle *h, GzVBlock *vbs, int n_vbs)
{
    if (!h || (n_vbs && !vbs) || n_vbs < 0) return GZ_ERR_ARG;
    HIPCHK (h, hipSetDevice (h->device));
    static const bool timing = getenv ("GZ_ZIP_TIMING") != NULL;
    const auto tm0 = std::chrono::steady_clock::now ();
    Plan P;
    std::vector<GzdVB> V (n_vbs);
    for (int v = 0; v < n_vbs; v++) for (uint32_t k = 0; k < vbs[v].n_sections; k++) if (vbs[v].sections[k].data_len >= (1u << 24)) P.unpacked = true;
    for (int v = 0; v < n_vbs; v++) {
        GzVBlock &u = vbs[v];
        GzdVB &D = V[v];
        memset (&D, 0, sizeof (D));
        D.z_data = u.z_data; D.z_cap = u.z_cap; D.first_stream = (uint32_t)P.streams.size (); D.n_streams = u.n_sections;
        D.vblock_i = u.vblock_i; D.recon_size = u.recon_size; D.longest_line_len = u.longest_line_len; D.longest_seq_len = u.longest_seq_len;
        memcpy (D.digest, u.digest, 16); D.vb_flags = u.vb_flags; D.status = GZ_ST_PENDING; D.mark_stream = u.mark_section;
        u.status = GZ_ST_PENDING; u.z_len = 0;
        for (uint32_t k = 0; k < u.n_sections; k++) {
            const GzSection &sec = u.sections[k];
            GzdStream S;
            memset (&S, 0, sizeof (S));
            int codec = sec.codec ? sec.codec : GZ_CODEC_RANB;            // zfile.c:300,337
            // (BZ2 / LZMA / BSC: only as payloads the host's coder has made - the section is framed here)
            if (!codec_ok (codec) && !(sec.precompressed && (codec == GZ_CODEC_BZ2 || codec == GZ_CODEC_LZMA || codec == GZ_CODEC_BSC))) { h->err = "unsupported codec in section"; return GZ_ERR_ARG; }
            S.in = sec.data; S.in_len = sec.data_len; S.in_len_dev = sec.data_len_dev;
            S.codec_req = codec; S.vb = v; S.sec_in_vb = k; S.status = GZ_ST_PENDING;
            S.out_cap = 0xffffffffu;
            uint8_t *hd = S.hdr;                                        // SectionHeaderCtx, sections.h:146-167,419-435
            gz_be32 (hd + 0, 0x27052012u);
            gz_be32 (hd + 20, u.vblock_i);
            hd[24] = sec.section_type; hd[25] = (uint8_t)codec; hd[26] = sec.sub_codec; hd[27] = sec.flags;
            if (sec.hdr_codec) { S.hdr_codec = sec.hdr_codec; hd[25] = sec.hdr_codec; hd[26] = (uint8_t)codec; }
            hd[28] = sec.ltype; hd[29] = sec.param; hd[30] = sec.b250_size_or_nothing_char; hd[31] = 0;
            memcpy (hd + 32, sec.dict_id, 8);
            if (sec.precompressed) { S.pre = 1; S.raw_len = sec.raw_len; S.codec_req = GZ_CODEC_NONE; P.streams.push_back (S); P.streams.back ().first_leaf = (uint32_t)P.leaves.size (); continue; }
            if (sec.data_len > P.max_in) P.max_in = sec.data_len;
            P.streams.push_back (S);
            // a section shorter than 50 bytes is stored raw; when the length is only known on the device we must
            // still plan the leaves of the requested codec
            if (sec.data_len < 50 && !sec.data_len_dev && !sec.hdr_codec) { P.streams.back ().codec_req = GZ_CODEC_NONE; P.streams.back ().hdr[25] = GZ_CODEC_NONE; }
            if (!plan_stream_leaves (h, P, (uint32_t)P.streams.size () - 1)) return GZ_ERR_HIP;
        }
    }
    void *d_streams, *d_leaves, *d_vbs;
    int rc;
    const auto tm1 = std::chrono::steady_clock::now ();
    if ((rc = upload (h, P.streams.data (), P.streams.size () * sizeof (GzdStream), &d_streams)) != GZ_OK) return rc;
    if ((rc = upload (h, P.leaves.data (), P.leaves.size () * sizeof (GzdLeaf), &d_leaves)) != GZ_OK) return rc;
    if ((rc = upload (h, V.data (), V.size () * sizeof (GzdVB), &d_vbs)) != GZ_OK) return rc;
    const auto tm2 = std::chrono::steady_clock::now ();
    if ((rc = launch_encode (h, P, (GzdStream *)d_streams, (GzdLeaf *)d_leaves, (GzdVB *)d_vbs, (uint32_t)n_vbs, 1)) != GZ_OK) return rc;
    if (timing) fprintf (stderr, "[vb_compress_batch bg %d vbs %d streams %zu leaves %zu (%zu B each): plan %.2f upload %.2f launch %.2f ms]\n", (int)h->background, n_vbs, P.streams.size (), P.leaves.size (),
                         sizeof (GzdLeaf), std::chrono::duration<double, std::milli> (tm1 - tm0).count (), std::chrono::duration<double, std::milli> (tm2 - tm1).count (),
                         std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now () - tm2).count ());
    Pending pd; pd.kind = 2; pd.user = vbs; pd.n = n_vbs; pd.dev_streams = d_streams; pd.dev_vbs = d_vbs; pd.n_dev_streams = P.streams.size ();
    pd.emit_after = h->emit_after_used; h->emit_after_used = NULL;
    h->pending.push_back (pd);
    return GZ_OK;
}

// ---------------------------------------------------------------------------------------------------------
// decode batch
// ---------------------------------------------------------------------------------------------------------
extern "C" int gz_codec_uncompress_batch (GzHandle *h, GzStream *streams, int n_streams)
{
    if (!h || (n_streams && !streams) || n_streams < 0) return GZ_ERR_ARG;
    HIPCHK (h, hipSetDevice (h->device));
    std::vector<GzdDecStream> DS (n_streams);
    std::vector<GzdDecLeaf> DL ((size_t)n_streams * 4);
    memset (DL.data (), 0, DL.size () * sizeof (GzdDecLeaf));
    bool any = false;
    for (int i = 0; i < n_streams; i++) {
        GzStream &u = streams[i];
        GzdDecStream &S = DS[i];
        memset (&S, 0, sizeof (S));
        u.out_len = 0;
        if (!codec_ok (u.codec)) { u.status = S.status = GZ_ERR_ARG; continue; }
        S.in = u.in; S.in_len = u.in_len; S.out = u.out; S.out_len = u.out_cap; S.codec = u.codec;
        S.status = GZ_ST_PENDING; S.first_leaf = (uint32_t)i * 4;
        u.status = GZ_ST_PENDING;
        any = true;
        if (u.codec == GZ_CODEC_NONE) continue;
        const bool rans = codec_is_rans (u.codec);
        size_t n = u.out_cap;
        if (!(S.tmp_planes = (uint8_t *)arena_alloc (h, n + 16))) return GZ_ERR_HIP;
        if (!(S.tmp_packed = (uint8_t *)arena_alloc (h, n + 64))) return GZ_ERR_HIP;
        for (int k = 0; k < 4; k++) {
            GzdDecLeaf &L = DL[(size_t)i * 4 + k];
            L.stream = (uint32_t)i;
            if (rans) {
                if (!(L.lut = (uint8_t *)arena_alloc (h, (size_t)256 << 12))) return GZ_ERR_HIP;
                if (!(L.fc  = (uint32_t *)arena_alloc (h, 256 * 256 * 4))) return GZ_ERR_HIP;
                if (!(L.tabtmp = (uint8_t *)arena_alloc (h, GZ_TAB_CAP))) return GZ_ERR_HIP;
            }
            else if (!(L.models = (uint32_t *)arena_alloc (h, ((size_t)256 * GZ_DEC_LIT_ROW (256) + 258 * GZ_DEC_RUN_ROW) * 4))) return GZ_ERR_HIP;   // (models beyond the LDS classes live here)
        }
    }
    void *d_streams, *d_leaves;
    int rc;
    if ((rc = upload (h, DS.data (), DS.size () * sizeof (GzdDecStream), &d_streams)) != GZ_OK) return rc;
    if ((rc = upload (h, DL.data (), DL.size () * sizeof (GzdDecLeaf), &d_leaves)) != GZ_OK) return rc;
    if (any) {
        const uint32_t ns = (uint32_t)n_streams;
        hipLaunchKernelGGL (k_dec_parse, dim3 ((ns + 63) / 64), dim3 (64), 0, h->stream, (GzdDecStream *)d_streams, (GzdDecLeaf *)d_leaves, ns);
        hipLaunchKernelGGL (k_dec_table, dim3 (ns * 4), dim3 (256), 20480, h->stream, (GzdDecLeaf *)d_leaves);
        hipLaunchKernelGGL (k_rans_decode, dim3 (ns * 4), dim3 (64), 0, h->stream, (GzdDecLeaf *)d_leaves);
        for (int c = 0; c < GZ_ARITH_CLASSES; c++)
            hipLaunchKernelGGL (k_arith_decode, dim3 (ns * 4), dim3 (64), ARITH_CLASS_WORDS[c + 1] * 4, h->stream,
                                (GzdDecLeaf *)d_leaves, ARITH_CLASS_WORDS[c], ARITH_CLASS_WORDS[c + 1], 0);
        hipLaunchKernelGGL (k_arith_decode, dim3 (ns * 4), dim3 (64), 0, h->stream, (GzdDecLeaf *)d_leaves, ARITH_CLASS_WORDS[GZ_ARITH_CLASSES], 0xffffffffu, 1);
        hipLaunchKernelGGL (k_dec_finish, dim3 (ns), dim3 (256), 0, h->stream, (GzdDecStream *)d_streams, (GzdDecLeaf *)d_leaves);
        HIPCHK (h, hipGetLastError ());
    }
    Pending pd; pd.kind = 1; pd.user = streams; pd.n = n_streams; pd.dev_streams = d_streams; pd.dev_vbs = NULL; pd.n_dev_streams = DS.size ();
    h->pending.push_back (pd);
    return GZ_OK;
}

// ---------------------------------------------------------------------------------------------------------
// sync: wait, fetch results, recycle the arena
// ---------------------------------------------------------------------------------------------------------
static int gz_sync_do (GzHandle *h);

extern "C" int gz_debug_record_inv (GzHandle *h, uint32_t tot0, uint32_t n, uint32_t *out_dev)
{
    if (!h || !out_dev) return GZ_ERR_ARG;
    if (!n) return GZ_OK;
    KLAUNCH (h, k_debug_record_inv, dim3 ((n + 255) / 256), dim3 (256), 0, tot0, n, out_dev);
    HIPCHK (h, hipStreamSynchronize (h->stream));
    return GZ_OK;
}

extern "C" int gz_sync (GzHandle *h)
{
    if (!h) return GZ_ERR_ARG;
    const int rc = gz_sync_do (h);
    if (rc == GZ_ERR_HIP) {
        // a failed wait / read-back: nothing of this batch may be handed out later (the tables of the callers may be gone
        // by the next sync) - drop the bookkeeping; the statuses in the callers' tables stay "pending"
        h->pending.clear ();
        for (auto p : h->host_tmp) free (p);
        h->host_tmp.clear ();
        arena_reset (h);
    }
    return rc;
}

static int gz_sync_do (GzHandle *h)
{
    HIPCHK (h, hipSetDevice (h->device));
    const hipError_t sync_err = hipStreamSynchronize (h->stream);
    g_chain_wgs.fetch_sub (h->chain_wgs_held); g_chain_cus.fetch_sub (h->chain_cus_held);
    h->chain_wgs_held = h->chain_cus_held = 0;
    HIPCHK (h, sync_err);
#ifdef GZ_MODEL_DEBUG
    if (!h->pending.empty ()) {
        unsigned long long v = 0, z = 0;
        (void)hipMemcpyFromSymbol (&v, HIP_SYMBOL (g_model_slowest), 8);
        (void)hipMemcpyToSymbol (HIP_SYMBOL (g_model_slowest), &z, 8);
        if (v) fprintf (stderr, "[model] bg %d slowest wave %.3f ms: %s list index %llu context %llu occurrences ~%llu\n", (int)h->background, (double)(v >> 40) / 1e5,
                        ((v >> 39) & 1) ? "small" : "big", (v >> 28) & 0x7ff, (v >> 18) & 0x3ff, (v & 0x3ffff) * 64);
    }
#endif
#ifdef GZ_MODEL_PHASES
    if (!h->pending.empty ()) {
        unsigned long long v[9], z[9] = { 0 };
        (void)hipMemcpyFromSymbol (v, HIP_SYMBOL (g_mph), sizeof (v));
        (void)hipMemcpyToSymbol (HIP_SYMBOL (g_mph), z, sizeof (z));
        if (v[7]) fprintf (stderr, "[phases] bg %d: %llu hot waves, %llu batches (%llu in rounds, %.2f rounds each), %.2f events per batch; us per batch: head %.3f register batches %.3f batches in rounds %.3f tail %.3f\n",
                           (int)h->background, v[7], v[4], v[5], v[5] ? (double)v[8] / (double)v[5] : 0.0, (double)v[6] / (double)v[4], (double)v[0] / 100.0 / (double)v[4],
                           v[4] > v[5] ? (double)v[1] / 100.0 / (double)(v[4] - v[5]) : 0.0, v[5] ? (double)v[2] / 100.0 / (double)v[5] : 0.0, (double)v[3] / 100.0 / (double)v[4]);
    }
#endif
#ifdef GZ_TABLE_DEBUG
    if (!h->pending.empty ()) {
        unsigned long long sm[9], mx[9], z[9] = { 0 };
        (void)hipMemcpyFromSymbol (sm, HIP_SYMBOL (g_tab_sum), sizeof (sm));
        (void)hipMemcpyFromSymbol (mx, HIP_SYMBOL (g_tab_max), sizeof (mx));
        (void)hipMemcpyToSymbol (HIP_SYMBOL (g_tab_sum), z, sizeof (z));
        (void)hipMemcpyToSymbol (HIP_SYMBOL (g_tab_max), z, sizeof (z));
        if (sm[8]) {
            fprintf (stderr, "[table] bg %d: %llu order-1 workgroups, the slowest %.3f ms; per phase mean / max in us:", (int)h->background, sm[8], (double)mx[8] / 1e5);
            for (int k = 0; k < 8; k++) fprintf (stderr, " %d: %.1f / %.1f", k, (double)sm[k] / (double)sm[8] / 100.0, (double)mx[k] / 100.0);
            fprintf (stderr, "\n");
        }
    }
#endif
    int rc = GZ_OK;
    bool device_failed = false;
    if (!h->pending.empty ()) {
        uint32_t f = 0;
        HIPCHK (h, hipMemcpy (&f, h->d_fail, 4, hipMemcpyDeviceToHost));
        if (f) { device_failed = true; HIPCHK (h, hipMemset (h->d_fail, 0, 4)); }
    }
    for (auto &pd : h->pending) {
        if (pd.kind == 0) {
            std::vector<GzdStream> S (pd.n_dev_streams);
            HIPCHK (h, hipMemcpy (S.data (), pd.dev_streams, S.size () * sizeof (GzdStream), hipMemcpyDeviceToHost));
            GzStream *u = (GzStream *)pd.user;
            for (int i = 0; i < pd.n; i++) {
                if (u[i].status != GZ_ST_PENDING) continue;
                u[i].status = S[i].status == GZ_ST_OK ? GZ_OK : S[i].status == GZ_ST_TOO_SMALL ? GZ_TOO_SMALL : GZ_ERR;
                u[i].out_len = S[i].status == GZ_ST_OK ? S[i].out_len : 0;
            }
        }
        else if (pd.kind == 1) {
            std::vector<GzdDecStream> S (pd.n_dev_streams);
            HIPCHK (h, hipMemcpy (S.data (), pd.dev_streams, S.size () * sizeof (GzdDecStream), hipMemcpyDeviceToHost));
            GzStream *u = (GzStream *)pd.user;
            for (int i = 0; i < pd.n; i++) {
                if (u[i].status != GZ_ST_PENDING) continue;
                u[i].status = S[i].status == GZ_ST_OK ? GZ_OK : GZ_ERR_CORRUPT;
                u[i].out_len = S[i].status == GZ_ST_OK ? S[i].out_len : 0;
            }
        }
        else {
            std::vector<GzdVB> V (pd.n);
            HIPCHK (h, hipMemcpy (V.data (), pd.dev_vbs, V.size () * sizeof (GzdVB), hipMemcpyDeviceToHost));
            GzVBlock *u = (GzVBlock *)pd.user;
            for (int i = 0; i < pd.n; i++) {
                u[i].status = V[i].status == GZ_ST_OK ? GZ_OK : V[i].status == GZ_ST_TOO_SMALL ? GZ_TOO_SMALL : GZ_ERR;
                u[i].z_len  = V[i].z_len; u[i].mark_index = V[i].mark_index;
                if (u[i].status == GZ_ERR) rc = GZ_ERR; else if (u[i].status != GZ_OK && rc != GZ_ERR) rc = GZ_TOO_SMALL;
            }
        }
    }
    if (!h->profiling || h->prof_open.size () > 4096) prof_collect (h);
    std::vector<Pending> again;
    if (device_failed && !h->in_fallback) again = h->pending;
    h->pending.clear ();
    for (auto p : h->host_tmp) free (p);
    h->host_tmp.clear ();
    arena_reset (h);
    if (device_failed && !h->in_fallback) {
        // The persistent chain kernel gave up waiting for the model kernels (kernels serialised by a tool, another tenant holding the hardware
        // queues, a host that initialised HIP before this library could ask for more queues): every stream / VBlock of this sync is suspect.
        // The reference's contract is "false only for too small, otherwise it works" (src/compressor.c:89-110): the batches run AGAIN, in the
        // unpipelined order of the same kernels - inputs and outputs are the caller's and still there, the tables too (they are pending) -
        // and the caller gets the result of that run, a warning in gz_last_error and a count in gz_chain_fallbacks.
        h->in_fallback = true;
        int rc2 = GZ_OK;
        for (auto &pd : again) {
            if (pd.kind == 0) rc2 = gz_codec_compress_batch (h, (GzStream *)pd.user, pd.n);
            else if (pd.kind == 2) { h->emit_after = pd.emit_after; rc2 = gz_vb_compress_batch (h, (GzVBlock *)pd.user, pd.n); }   // (the writer waits for the other handle again)
            if (rc2 < 0) break;                                           // (kind 1, decoding, has no chain: its results above stand)
        }
        if (rc2 >= 0) rc2 = gz_sync_do (h); else (void)gz_sync_do (h);
        h->in_fallback = false;
        h->chain_fallbacks++;
        if (rc2 >= 0) h->warn = "warning: the arithmetic coder's persistent chain kernel never heard from the model kernels (kernels serialised by a tool? "
                                "too few hardware queues?); the batch was run again unpipelined (GZ_NO_PIPELINE=1 selects that order from the start)";
        return rc2;
    }
    if (device_failed) {
        h->err = "the arithmetic coder failed in the unpipelined order as well";
        return GZ_ERR;
    }
    return rc;
}

extern "C" uint32_t gz_chain_fallbacks (GzHandle *h) { return h ? h->chain_fallbacks : 0; }

// ---------------------------------------------------------------------------------------------------------
// host-pointer single-call forms == the reference's COMPRESS()/UNCOMPRESS() signatures
// ---------------------------------------------------------------------------------------------------------
extern "C" int gz_codec_compress_host (GzHandle *h, int codec, const uint8_t *in, uint32_t in_len,
                                       uint8_t *out, uint32_t *out_len, int soft_fail)
{
    if (!h || !out_len || (in_len && !in) || !out) return GZ_ERR_ARG;
    if (!codec_ok (codec)) return GZ_ERR_ARG;
    const uint32_t est = gz_codec_est_size (codec, in_len);
    if (*out_len < codec_min_cap (codec, in_len)) return soft_fail ? GZ_TOO_SMALL : GZ_ERR;
    int rc;
    if ((rc = gz_sync (h)) < 0) return rc;           // own the arena
    // the staging buffers are NOT arena memory: gz_sync below recycles the arena before the payload is copied out
    uint8_t *d_buf = NULL;
    HIPCHK (h, hipMalloc ((void **)&d_buf, (size_t)in_len + 256 + (size_t)est + 16));
    uint8_t *d_in = d_buf, *d_out = d_buf + (((size_t)in_len + 255) & ~(size_t)255);
    GzStream s; memset (&s, 0, sizeof (s));
    s.in = d_in; s.in_len = in_len; s.out = d_out; s.out_cap = est; s.codec = codec;
    hipError_t e = in_len ? hipMemcpyAsync (d_in, in, in_len, hipMemcpyHostToDevice, h->stream) : hipSuccess;
    if (e != hipSuccess) { (void)hipFree (d_buf); h->err = std::string ("hipMemcpyAsync: ") + hipGetErrorString (e); return GZ_ERR_HIP; }
    rc = gz_codec_compress_batch (h, &s, 1);
    const int rc2 = gz_sync (h);                      // (also when the batch call failed: nothing stays pending)
    if (rc == GZ_OK && rc2 < 0) rc = rc2;
    if (rc == GZ_OK && s.status != GZ_OK) rc = s.status;
    if (rc == GZ_OK && s.out_len && (e = hipMemcpy (out, d_out, s.out_len, hipMemcpyDeviceToHost)) != hipSuccess) {
        h->err = std::string ("hipMemcpy: ") + hipGetErrorString (e); rc = GZ_ERR_HIP;
    }
    (void)hipFree (d_buf);
    if (rc == GZ_OK) *out_len = s.out_len;
    return rc;
}

extern "C" int gz_codec_compress_lines_host (GzHandle *h, int codec, GzGetLineCB get_line, void *user, uint32_t n_lines, uint32_t in_len,
                                             uint8_t *out, uint32_t *out_len, int soft_fail)
{
    if (!h || !out_len || !out || (n_lines && !get_line)) return GZ_ERR_ARG;
    if (!codec_ok (codec)) return GZ_ERR_ARG;
    if (*out_len < codec_min_cap (codec, in_len)) return soft_fail ? GZ_TOO_SMALL : GZ_ERR;
    HIPCHK (h, hipSetDevice (h->device));
    uint8_t *stage = NULL;
    HIPCHK (h, hipHostMalloc ((void **)&stage, (size_t)in_len + 64, hipHostMallocDefault));
    uint64_t at = 0;
    for (uint32_t i = 0; i < n_lines; i++) {
        const uint8_t *line = NULL; uint32_t len = 0;
        get_line (user, i, &line, &len);
        if (at + len > in_len || (len && !line)) { (void)hipHostFree (stage); h->err = "get_line: more bytes than uncompressed_len"; return GZ_ERR_CORRUPT; }
        if (len) memcpy (stage + at, line, len);
        at += len;
    }
    if (at != in_len) { (void)hipHostFree (stage); h->err = "get_line: total length != uncompressed_len (codec_htscodecs.c:61)"; return GZ_ERR_CORRUPT; }
    const int rc = gz_codec_compress_host (h, codec, stage, in_len, out, out_len, soft_fail);
    (void)hipHostFree (stage);
    return rc;
}

extern "C" int gz_codec_uncompress_host (GzHandle *h, int codec, const uint8_t *in, uint32_t in_len,
                                         uint8_t *out, uint64_t out_len)
{
    if (!h || (in_len && !in) || (out_len && !out) || out_len > 0xffffffffull) return GZ_ERR_ARG;
    if (!codec_ok (codec)) return GZ_ERR_ARG;
    int rc;
    if ((rc = gz_sync (h)) < 0) return rc;
    uint8_t *d_in  = (uint8_t *)arena_alloc (h, (size_t)in_len + 16);
    uint8_t *d_out = (uint8_t *)arena_alloc (h, (size_t)out_len + 16);
    if (!d_in || !d_out) return GZ_ERR_HIP;
    if (in_len) HIPCHK (h, hipMemcpyAsync (d_in, in, in_len, hipMemcpyHostToDevice, h->stream));
    GzStream s; memset (&s, 0, sizeof (s));
    s.in = d_in; s.in_len = in_len; s.out = d_out; s.out_cap = (uint32_t)out_len; s.codec = codec;
    if ((rc = gz_codec_uncompress_batch (h, &s, 1)) != GZ_OK) return rc;
    if ((rc = gz_sync (h)) < 0) return rc;
    if (s.status != GZ_OK) return s.status;
    if (out_len) HIPCHK (h, hipMemcpy (out, d_out, out_len, hipMemcpyDeviceToHost));
    return GZ_OK;
}

// ---------------------------------------------------------------------------------------------------------
// codec_assign_best_codec, deterministic rule
// ---------------------------------------------------------------------------------------------------------
extern "C" int gz_codec_assign_best (GzHandle *h, const uint8_t *in, uint32_t in_len, uint32_t *sizes_out)
{
    static const int cand[9] = { GZ_CODEC_NONE, GZ_CODEC_RANB, GZ_CODEC_RANW, GZ_CODEC_RANb, GZ_CODEC_RANw,
                                 GZ_CODEC_ARTB, GZ_CODEC_ARTW, GZ_CODEC_ARTb, GZ_CODEC_ARTw };
    if (!h || (in_len && !in)) return GZ_ERR_ARG;
    uint32_t sample = in_len < 99999 ? in_len : 99999;                  // codec.c:309
    if (sample < 50) return GZ_CODEC_UNKNOWN;                          // codec.c:311-312
    int rc;
    if ((rc = gz_sync (h)) < 0) return rc;
    GzStream s[8]; memset (s, 0, sizeof (s));
    for (int i = 0; i < 8; i++) {
        s[i].in = in; s[i].in_len = sample; s[i].codec = cand[i + 1];
        s[i].out_cap = gz_codec_est_size (cand[i + 1], sample);
        if (!(s[i].out = (uint8_t *)arena_alloc (h, s[i].out_cap + 16))) return GZ_ERR_HIP;
    }
    if ((rc = gz_codec_compress_batch (h, s, 8)) != GZ_OK) return rc;
    if ((rc = gz_sync (h)) < 0) return rc;
    int best = GZ_CODEC_NONE;
    uint32_t best_size = sample;                                       // NONE: bare length (codec.c:324)
    if (sizes_out) sizes_out[0] = sample;
    for (int i = 0; i < 8; i++) {
        if (s[i].status != GZ_OK) return GZ_ERR;
        uint32_t size = s[i].out_len + 28;                             // framed: + SectionHeader (codec.c:328-331)
        if (sizes_out) sizes_out[i + 1] = size;
        if (size < best_size) { best_size = size; best = cand[i + 1]; }
    }
    return best;
}

// codec_assign_sorter (src/codec.c:128-173), restated: < 0 when a goes first
static int gz_assign_cmp (const GzCodecTest &a, const GzCodecTest &b, int mode)
{
    if (mode == GZ_ASSIGN_FAST) {                                      // much faster at a modest cost in size
        if (a.clock_us < b.clock_us * 0.80f && a.size < b.size * 1.3f) return -1;
        if (b.clock_us < a.clock_us * 0.80f && b.size < a.size * 1.3f) return 1;
    }
    if (mode == GZ_ASSIGN_BEST || (a.clock_us <= 5000 && b.clock_us <= 5000)) {     // both fast enough: size, then time
        if (a.size != b.size) return a.size < b.size ? -1 : 1;
        return a.clock_us < b.clock_us ? -1 : a.clock_us > b.clock_us ? 1 : 0;
    }
    if (a.size < 100 && b.size < 100 && a.clock_us != b.clock_us) return a.clock_us < b.clock_us ? -1 : 1;   // both tiny: the faster
    static const float level[5][2] = { { 0.96f, 0.20f }, { 0.97f, 0.33f }, { 0.98f, 0.50f }, { 0.985f, 0.67f }, { 0.99f, 0.85f } };
    for (int l = 0; l < 5; l++) {
        if (a.size < b.size * level[l][0]) return -1;                  // significantly smaller
        if (b.size < a.size * level[l][0]) return 1;
        if (a.clock_us < b.clock_us * level[l][1]) return -1;          // similar size: significantly faster
        if (b.clock_us < a.clock_us * level[l][1]) return 1;
    }
    if (a.size == b.size) return a.codec < b.codec ? -1 : a.codec > b.codec ? 1 : 0;     // the lower codec id (the form without PACK)
    return a.size < b.size ? -1 : 1;
}

// qsort (tests, n, ..., codec_assign_sorter) (src/codec.c:338) as the C library of the reference's Linux builds runs it: glibc's
// qsort is a top-down merge sort for arrays this small (halves of n / 2 and n - n / 2 elements, the left element first unless the
// comparator says it is larger). The comparator is not a strict order - the outcome depends on the algorithm, so it is stated.
static void gz_assign_msort (GzCodecTest *b, int n, GzCodecTest *tmp, int mode)
{
    if (n <= 1) return;
    const int n1 = n / 2, n2 = n - n1;
    gz_assign_msort (b, n1, tmp, mode); gz_assign_msort (b + n1, n2, tmp, mode);
    int i = 0, j = n1, k = 0;
    while (i < n1 && j < n) tmp[k++] = gz_assign_cmp (b[i], b[j], mode) <= 0 ? b[i++] : b[j++];
    while (i < n1) tmp[k++] = b[i++];
    for (int x = 0; x < k; x++) b[x] = tmp[x];                          // (what is left of the right half is in place already)
}

extern "C" int gz_codec_assign_sort (GzCodecTest *tests, int n, int mode)
{
    if (!tests || n <= 0 || mode < 0 || mode > 2) return GZ_ERR_ARG;
    std::vector<GzCodecTest> tmp (n);
    gz_assign_msort (tests, n, tmp.data (), mode);
    return tests[0].codec;
}

// the rows of the nine device candidates from their trial results (payload lengths), in the reference's order, + the caller's; sorted
static int gz_assign_pick (uint32_t sample, const uint32_t payload[8], const GzCodecTest *extra, int n_extra, const float *ns_per_byte, int mode, GzCodecTest *tests_out)
{
    static const int cand[9] = { GZ_CODEC_NONE, GZ_CODEC_RANB, GZ_CODEC_RANW, GZ_CODEC_RANb, GZ_CODEC_RANw,
                                 GZ_CODEC_ARTB, GZ_CODEC_ARTW, GZ_CODEC_ARTb, GZ_CODEC_ARTw };
    std::vector<GzCodecTest> t (9 + (n_extra > 0 ? n_extra : 0));
    for (int i = 0; i < 9; i++) {
        t[i].codec = cand[i];
        t[i].size = i ? (float)(payload[i - 1] + 28) : (float)sample;   // framed (codec.c:328-331); NONE: the bare length (:324)
        t[i].clock_us = i && ns_per_byte ? (float)sample * ns_per_byte[cand[i]] / 1000.0f : 0.0f;
    }
    for (int i = 0; i < n_extra; i++) t[9 + i] = extra[i];
    const int codec = gz_codec_assign_sort (t.data (), (int)t.size (), mode);
    if (tests_out) memcpy (tests_out, t.data (), t.size () * sizeof (GzCodecTest));
    return codec;
}

extern "C" int gz_codec_assign_best_ex (GzHandle *h, const uint8_t *in, uint32_t in_len, const GzCodecTest *extra, int n_extra,
                                        const float *clock_ns_per_byte, int mode, GzCodecTest *tests_out)
{
    if (n_extra < 0 || (n_extra && !extra) || mode < 0 || mode > 2) return GZ_ERR_ARG;
    uint32_t sizes[9];
    const int rc = gz_codec_assign_best (h, in, in_len, sizes);
    if (rc < 0 || rc == GZ_CODEC_UNKNOWN) return rc;
    uint32_t payload[8];
    for (int i = 0; i < 8; i++) payload[i] = sizes[i + 1] - 28;
    return gz_assign_pick (sizes[0], payload, extra, n_extra, clock_ns_per_byte, mode, tests_out);
}

// ---------------------------------------------------------------------------------------------------------
// N3: CODEC_DOMQ's pre-transform (gz_kernels_domq.h)
// ---------------------------------------------------------------------------------------------------------
extern "C" int gz_domq_columns (GzHandle *h, const GzDomqJob *jobs, int n_jobs)
{
    if (!h || (n_jobs && !jobs) || n_jobs < 0) return GZ_ERR_ARG;
    if (!n_jobs) return GZ_OK;
    HIPCHK (h, hipSetDevice (h->device));
    std::vector<GzdDomq> J (n_jobs);
    uint32_t max_n = 1;
    size_t hist_words = 0;
    for (int i = 0; i < n_jobs; i++) hist_words += GZ_DQ_HIST + GZ_DQ_MISC;
    uint32_t *hist = (uint32_t *)arena_alloc (h, hist_words * 4);
    if (!hist) return GZ_ERR_HIP;
    HIPCHK (h, hipMemsetAsync (hist, 0, hist_words * 4, h->stream));
    for (int i = 0; i < n_jobs; i++) {
        const GzDomqJob &u = jobs[i];
        if (!u.result_dev || !u.qual || !u.mplx || (u.n && (!u.text || !u.off || !u.len || !u.runs || !u.divr))) return GZ_ERR_ARG;
        GzdDomq &d = J[i];
        d.text = u.text; d.off = u.off; d.len = u.len; d.n = u.n; d.qual = u.qual; d.runs = u.runs; d.mplx = u.mplx; d.divr = u.divr; d.res = u.result_dev; d.only_if = u.only_if_dev;
        d.hist = hist + (size_t)i * (GZ_DQ_HIST + GZ_DQ_MISC);
        if (!(d.line_dom = (uint8_t *)arena_alloc (h, (size_t)u.n + 16)) || !(d.normalize = (uint8_t *)arena_alloc (h, GZ_DQ_HIST))
            || !(d.rec = (uint32_t *)arena_alloc (h, ((size_t)6 * u.n + 4) * 4)) || !(d.lo = (uint32_t *)arena_alloc (h, ((size_t)5 * u.n + 4) * 4))) return GZ_ERR_HIP;
        if (u.n > max_n) max_n = u.n;
    }
    void *dj;
    int rc;
    if ((rc = upload (h, J.data (), J.size () * sizeof (GzdDomq), &dj)) != GZ_OK) return rc;
    const dim3 by_line ((max_n + GZ_DQ_LINES_PER_WG - 1) / GZ_DQ_LINES_PER_WG, (uint32_t)n_jobs);
    KLAUNCH (h, k_domq_lines, by_line, dim3 (256), GZ_DOMQ_LDS, (const GzdDomq *)dj);
    KLAUNCH (h, k_domq_tables, dim3 ((uint32_t)n_jobs), dim3 (128), 64, (const GzdDomq *)dj);
    KLAUNCH (h, k_domq_measure, by_line, dim3 (256), GZ_DQ_NORM_LDS, (const GzdDomq *)dj);
    KLAUNCH (h, k_domq_scan, dim3 ((uint32_t)n_jobs), dim3 (GZ_DQ_SCAN_NT), 4096, (const GzdDomq *)dj);
    KLAUNCH (h, k_domq_write, by_line, dim3 (256), GZ_DQ_NORM_LDS, (const GzdDomq *)dj);
    HIPCHK (h, hipGetLastError ());
    return GZ_OK;
}

extern "C" int gz_domq_fit (GzHandle *h, const GzDomqFitJob *jobs, int n_jobs)
{
    if (!h || (n_jobs && !jobs) || n_jobs < 0) return GZ_ERR_ARG;
    if (!n_jobs) return GZ_OK;
    HIPCHK (h, hipSetDevice (h->device));
    std::vector<GzdDomqFit> J (n_jobs);
    for (int i = 0; i < n_jobs; i++) {
        if (!jobs[i].fit_dev || (jobs[i].n && (!jobs[i].text || !jobs[i].off || !jobs[i].len))) return GZ_ERR_ARG;
        J[i].text = jobs[i].text; J[i].off = jobs[i].off; J[i].len = jobs[i].len; J[i].n = jobs[i].n; J[i].fit = jobs[i].fit_dev;
    }
    void *dj;
    int rc;
    if ((rc = upload (h, J.data (), J.size () * sizeof (GzdDomqFit), &dj)) != GZ_OK) return rc;
    KLAUNCH (h, k_domq_fit, dim3 ((uint32_t)n_jobs), dim3 (64), 512, (const GzdDomqFit *)dj, (uint32_t)n_jobs);
    HIPCHK (h, hipGetLastError ());
    return GZ_OK;
}

// the same for host memory (global-area sections: dictionaries, counts)
extern "C" int gz_codec_assign_best_host (GzHandle *h, const uint8_t *in, uint32_t in_len)
{
    if (!h || (in_len && !in)) return GZ_ERR_ARG;
    if (in_len < 50) return GZ_CODEC_UNKNOWN;
    uint8_t *d = NULL;
    HIPCHK (h, hipSetDevice (h->device));
    HIPCHK (h, hipMalloc ((void **)&d, (size_t)in_len + 16));
    int rc = hipMemcpy (d, in, in_len, hipMemcpyHostToDevice) == hipSuccess ? gz_codec_assign_best (h, d, in_len, NULL) : GZ_ERR_HIP;
    (void)hipFree (d);
    return rc;
}

// ---------------------------------------------------------------------------------------------------------
// context engine pieces
// ---------------------------------------------------------------------------------------------------------
extern "C" int gz_b250_generate_batch (GzHandle *h, const GzB250Job *jobs, int n_jobs)
{
    if (!h || (n_jobs && !jobs) || n_jobs < 0) return GZ_ERR_ARG;
    if (!n_jobs) return GZ_OK;
    HIPCHK (h, hipSetDevice (h->device));
    std::vector<GzdB250Job> J;                     // one workgroup each
    std::vector<GzdB250Big> B;                     // the long ones: kernels over all chunks (gz_kernels_ctx.h)
    uint32_t max_super = 0, max_tiles = 0;
    bool any_r1 = false;
    for (int i = 0; i < n_jobs; i++) {
        const GzB250Job &u = jobs[i];
        if (!u.out_len_dev || (u.seg_len && (!u.seg || !u.out))) return GZ_ERR_ARG;
        GzdB250Job d;
        memset (&d, 0, sizeof (d));
        d.seg = u.seg; d.seg_len = u.seg_len; d.seg_len_dev = u.seg_len_dev; d.ol_nodes_len = u.ol_nodes_len;
        d.node2word = u.node2word; d.n_new_nodes = u.n_new_nodes; d.out = u.out; d.out_len_dev = u.out_len_dev;
        d.status_dev = u.status_dev; d.r1 = u.r1; d.r1_len_dev = u.r1_len_dev;
        if (u.r1) any_r1 = true;
        // scratch: one int32 per possible word (every word is at least one byte) + the chunk table
        size_t nchunks = ((size_t)u.seg_len + GZ_B250_CHUNK - 1) / GZ_B250_CHUNK;
        if (!(d.wi = (int32_t *)arena_alloc (h, ((size_t)u.seg_len + 1) * 4))) return GZ_ERR_HIP;
        if (!(d.chunk_tab = (uint32_t *)arena_alloc (h, (nchunks + 1) * 7 * 4))) return GZ_ERR_HIP;
        if (u.seg_len < GZ_B250_BIG) { J.push_back (d); continue; }
        GzdB250Big b;
        b.j = d;
        const uint32_t nsuper = (uint32_t)((nchunks + GZ_B250_SUPER - 1) / GZ_B250_SUPER), tiles = (u.seg_len + 255) / 256;
        if (!(b.stab = (uint32_t *)arena_alloc (h, ((size_t)nsuper + 1) * 7 * 4))) return GZ_ERR_HIP;
        if (!(b.tile = (uint64_t *)arena_alloc (h, ((size_t)tiles + 1) * 8))) return GZ_ERR_HIP;
        if (!(b.info = (uint32_t *)arena_alloc (h, 16))) return GZ_ERR_HIP;
        if (nsuper > max_super) max_super = nsuper;
        if (tiles > max_tiles) max_tiles = tiles;
        B.push_back (b);
    }
    void *d_jobs;
    int rc;
    if (!J.empty ()) {
        if ((rc = upload (h, J.data (), J.size () * sizeof (GzdB250Job), &d_jobs)) != GZ_OK) return rc;
        KLAUNCH (h, k_b250_generate, dim3 ((uint32_t)J.size ()), dim3 (256), 4096, (GzdB250Job *)d_jobs);
        if (any_r1) KLAUNCH (h, k_b250_pair_identical, dim3 ((uint32_t)J.size ()), dim3 (256), 64, (GzdB250Job *)d_jobs);
    }
    if (!B.empty ()) {
        if ((rc = upload (h, B.data (), B.size () * sizeof (GzdB250Big), &d_jobs)) != GZ_OK) return rc;
        GzdB250Big *db = (GzdB250Big *)d_jobs;
        const uint32_t nb = (uint32_t)B.size ();
        KLAUNCH (h, k_b250_walk, dim3 (max_super, nb), dim3 (256), (GZ_B250_SUPER * 5 + 4) * 4, db);
        KLAUNCH (h, k_b250_chain, dim3 (nb), dim3 (1), 0, db);
        KLAUNCH (h, k_b250_convert, dim3 (max_super, nb), dim3 (256), GZ_B250_SUPER * 8, db);
        KLAUNCH (h, k_b250_len, dim3 (max_tiles, nb), dim3 (256), 1024, db);
        KLAUNCH (h, k_b250_scan, dim3 (nb), dim3 (256), 2048, db);
        KLAUNCH (h, k_b250_emit, dim3 (max_tiles, nb), dim3 (256), 1024, db);
        if (any_r1) {                                  // (GzdB250Big starts with its GzdB250Job: a table of the jobs alone)
            std::vector<GzdB250Job> bj (B.size ());
            for (size_t i = 0; i < B.size (); i++) bj[i] = B[i].j;
            if ((rc = upload (h, bj.data (), bj.size () * sizeof (GzdB250Job), &d_jobs)) != GZ_OK) return rc;
            KLAUNCH (h, k_b250_pair_identical, dim3 (nb), dim3 (256), 64, (GzdB250Job *)d_jobs);
        }
    }
    HIPCHK (h, hipGetLastError ());
    return GZ_OK;
}

extern "C" int gz_b250_generate (GzHandle *h, const uint8_t *seg, uint32_t seg_len, uint32_t ol_nodes_len,
                                 const int32_t *node2word, uint32_t n_new_nodes, uint8_t *out, uint32_t *out_len_dev)
{
    GzB250Job j; memset (&j, 0, sizeof (j));
    j.seg = seg; j.seg_len = seg_len; j.ol_nodes_len = ol_nodes_len; j.node2word = node2word; j.n_new_nodes = n_new_nodes;
    j.out = out; j.out_len_dev = out_len_dev;
    return gz_b250_generate_batch (h, &j, 1);
}

static uint32_t lt_width (int lt)   // local_type.h:75-108
{
    switch (lt) {
        case GZ_LT_INT16: case GZ_LT_UINT16: case GZ_LT_UINT16_TR: return 2;
        case GZ_LT_INT32: case GZ_LT_UINT32: case GZ_LT_FLOAT32: case GZ_LT_UINT32_TR: return 4;
        case GZ_LT_INT64: case GZ_LT_UINT64: case GZ_LT_FLOAT64: case GZ_LT_BITMAP: return 8;
        default: return 1;
    }
}

static int local_xform (GzHandle *h, int ltype, void *data, uint64_t n, uint32_t cols, void *scratch, int to_file)
{
    if (!h || (n && !data)) return GZ_ERR_ARG;
    HIPCHK (h, hipSetDevice (h->device));
    const uint32_t w = lt_width (ltype);
    const bool is_signed = ltype == GZ_LT_INT8 || ltype == GZ_LT_INT16 || ltype == GZ_LT_INT32 || ltype == GZ_LT_INT64;
    int base = ltype == GZ_LT_UINT8_TR ? GZ_LT_UINT8 : ltype == GZ_LT_UINT16_TR ? GZ_LT_UINT16 : ltype == GZ_LT_UINT32_TR ? GZ_LT_UINT32 : ltype;
    const bool swap = (w > 1 || is_signed) && ltype != GZ_LT_BITMAP;
    const uint32_t blocks = (uint32_t)((n + 1023) / 1024 > 4096 ? 4096 : (n + 1023) / 1024);
    int result = ltype;

    if (to_file) {
        if (swap && n) hipLaunchKernelGGL (k_local_order, dim3 (blocks ? blocks : 1), dim3 (256), 0, h->stream, (uint8_t *)data, n, w, (int)is_signed, 1);
        if (cols && n % cols == 0 && (base == GZ_LT_UINT8 || base == GZ_LT_UINT16 || base == GZ_LT_UINT32) && n) {
            if (!scratch) return GZ_ERR_ARG;
            uint32_t rows = (uint32_t)(n / cols);
            hipLaunchKernelGGL (k_transpose, dim3 ((cols + 31) / 32, (rows + 31) / 32), dim3 (32, 8), 32 * 33 * 4 + 128, h->stream,
                                (const uint8_t *)data, (uint8_t *)scratch, rows, cols, w);
            HIPCHK (h, hipMemcpyAsync (data, scratch, n * w, hipMemcpyDeviceToDevice, h->stream));
            result = base == GZ_LT_UINT8 ? GZ_LT_UINT8_TR : base == GZ_LT_UINT16 ? GZ_LT_UINT16_TR : GZ_LT_UINT32_TR;
        }
    }
    else {
        if (cols && (ltype == GZ_LT_UINT8_TR || ltype == GZ_LT_UINT16_TR || ltype == GZ_LT_UINT32_TR) && n) {
            if (!scratch || n % cols) return GZ_ERR_ARG;
            uint32_t rows = (uint32_t)(n / cols);           // file holds cols x rows; give back rows x cols
            hipLaunchKernelGGL (k_transpose, dim3 ((rows + 31) / 32, (cols + 31) / 32), dim3 (32, 8), 32 * 33 * 4 + 128, h->stream,
                                (const uint8_t *)data, (uint8_t *)scratch, cols, rows, w);
            HIPCHK (h, hipMemcpyAsync (data, scratch, n * w, hipMemcpyDeviceToDevice, h->stream));
            result = base;
        }
        if (swap && n) hipLaunchKernelGGL (k_local_order, dim3 (blocks ? blocks : 1), dim3 (256), 0, h->stream, (uint8_t *)data, n, w, (int)is_signed, 0);
    }
    HIPCHK (h, hipGetLastError ());
    return result;
}

extern "C" int gz_local_generate (GzHandle *h, int ltype, void *data, uint64_t n, uint32_t cols, void *scratch)
{ return local_xform (h, ltype, data, n, cols, scratch, 1); }

extern "C" int gz_local_to_native (GzHandle *h, int ltype, void *data, uint64_t n, uint32_t cols, void *scratch)
{ return local_xform (h, ltype, data, n, cols, scratch, 0); }

static int local_partial (GzHandle *h, int ltype, void *data, uint64_t n_present, uint32_t rows, uint32_t cols,
                          const uint8_t *missing, void *scratch, int to_file)
{
    const int base = to_file ? ltype : ltype == GZ_LT_UINT8_PTR ? GZ_LT_UINT8 : ltype == GZ_LT_UINT16_PTR ? GZ_LT_UINT16 : ltype == GZ_LT_UINT32_PTR ? GZ_LT_UINT32 : -1;
    if (!h || (base != GZ_LT_UINT8 && base != GZ_LT_UINT16 && base != GZ_LT_UINT32) || !rows || !cols || !missing) return GZ_ERR_ARG;
    if (n_present && (!data || !scratch)) return GZ_ERR_ARG;
    const uint64_t cells = (uint64_t)rows * cols;
    if (cells >= (1ull << 32) || n_present > cells) return GZ_ERR_ARG;
    int rc;
    if ((rc = gz_sync (h)) < 0) return rc;
    HIPCHK (h, hipSetDevice (h->device));
    const uint32_t w = lt_width (base);
    if (to_file && w > 1 && n_present) {
        const uint32_t blocks = (uint32_t)((n_present + 1023) / 1024 > 4096 ? 4096 : (n_present + 1023) / 1024);
        hipLaunchKernelGGL (k_local_order, dim3 (blocks), dim3 (256), 0, h->stream, (uint8_t *)data, n_present, w, 0, 1);
    }
    GzdPartial P;
    memset (&P, 0, sizeof (P));
    const uint32_t tiles = (uint32_t)((cells + 255) / 256);
    uint8_t *miss_t = (uint8_t *)arena_alloc (h, cells);
    P.rank_a = (uint32_t *)arena_alloc (h, cells * 4);
    P.tile_a = (uint64_t *)arena_alloc (h, ((size_t)tiles + 1) * 8);
    P.tile_b = (uint64_t *)arena_alloc (h, ((size_t)tiles + 1) * 8);
    P.status = (int32_t *)arena_alloc (h, 4);
    if (!miss_t || !P.rank_a || !P.tile_a || !P.tile_b || !P.status) return GZ_ERR_HIP;
    P.missing = missing; P.miss_t = miss_t; P.cells = cells; P.rows = rows; P.cols = cols; P.w = w; P.n_present = n_present;
    P.in = (const uint8_t *)data; P.out = (uint8_t *)scratch; P.to_file = to_file ? 1 : 0;
    const int32_t ok = GZ_ST_OK;
    HIPCHK (h, hipMemcpy (P.status, &ok, 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL (k_transpose, dim3 ((cols + 31) / 32, (rows + 31) / 32), dim3 (32, 8), 32 * 33 * 4 + 128, h->stream, missing, miss_t, rows, cols, 1u);
    KLAUNCH (h, k_ptr_count, dim3 (tiles, 2), dim3 (256), 2048, P);
    KLAUNCH (h, k_ptr_scan, dim3 (2), dim3 (256), 2048, P);
    KLAUNCH (h, k_ptr_rank, dim3 (tiles), dim3 (256), 2048, P);
    KLAUNCH (h, k_ptr_gather, dim3 (tiles), dim3 (256), 2048, P);
    if (n_present) HIPCHK (h, hipMemcpyAsync (data, scratch, n_present * w, hipMemcpyDeviceToDevice, h->stream));
    if (!to_file && w > 1 && n_present) {
        const uint32_t blocks = (uint32_t)((n_present + 1023) / 1024 > 4096 ? 4096 : (n_present + 1023) / 1024);
        hipLaunchKernelGGL (k_local_order, dim3 (blocks), dim3 (256), 0, h->stream, (uint8_t *)data, n_present, w, 0, 0);
    }
    HIPCHK (h, hipGetLastError ());
    HIPCHK (h, hipStreamSynchronize (h->stream));
    int32_t st = 0;
    HIPCHK (h, hipMemcpy (&st, P.status, 4, hipMemcpyDeviceToHost));
    if ((rc = gz_sync (h)) < 0) return rc;
    if (st != GZ_ST_OK) { h->err = "partial transpose: the mask does not leave n_present elements"; return GZ_ERR_CORRUPT; }
    if (!to_file) return base;
    return base == GZ_LT_UINT8 ? GZ_LT_UINT8_PTR : base == GZ_LT_UINT16 ? GZ_LT_UINT16_PTR : GZ_LT_UINT32_PTR;
}

extern "C" int gz_local_generate_partial (GzHandle *h, int ltype, void *data, uint64_t n_present, uint32_t rows, uint32_t cols, const uint8_t *missing, void *scratch)
{ return local_partial (h, ltype, data, n_present, rows, cols, missing, scratch, 1); }

extern "C" int gz_local_partial_to_native (GzHandle *h, int ltype, void *data, uint64_t n_present, uint32_t rows, uint32_t cols, const uint8_t *missing, void *scratch)
{ return local_partial (h, ltype, data, n_present, rows, cols, missing, scratch, 0); }

extern "C" int gz_adler32 (GzHandle *h, const uint8_t *data, uint64_t len, uint32_t *adler_out)
{
    if (!h || !adler_out || (len && !data) || len > 0xffffffffull) return GZ_ERR_ARG;
    int rc;
    if ((rc = gz_sync (h)) < 0) return rc;
    uint32_t *d = (uint32_t *)arena_alloc (h, 4);
    if (!d) return GZ_ERR_HIP;
    hipLaunchKernelGGL (k_adler32, dim3 (1), dim3 (256), 4096, h->stream, data, (uint32_t)len, d);
    HIPCHK (h, hipGetLastError ());
    HIPCHK (h, hipStreamSynchronize (h->stream));
    HIPCHK (h, hipMemcpy (adler_out, d, 4, hipMemcpyDeviceToHost));
    return GZ_OK;
}

// ---------------------------------------------------------------------------------------------------------
// CODEC_ACGT pre-transform
// ---------------------------------------------------------------------------------------------------------
extern "C" uint64_t gz_acgt_packed_len (uint64_t n_bases) { return ((2 * n_bases + 63) / 64) * 8; }

extern "C" int gz_acgt_pack (GzHandle *h, const uint8_t *seq, uint64_t n_bases, uint8_t *packed, uint8_t *x, int *has_x_host)
{
    if (!h || !has_x_host || (n_bases && (!seq || !packed || !x))) return GZ_ERR_ARG;
    int rc;
    if ((rc = gz_sync (h)) < 0) return rc;
    uint32_t *d = (uint32_t *)arena_alloc (h, 4);
    if (!d) return GZ_ERR_HIP;
    HIPCHK (h, hipMemsetAsync (d, 0, 4, h->stream));
    const uint64_t pb = gz_acgt_packed_len (n_bases);
    if (pb) {
        uint64_t blocks = (pb / 4 + 255) / 256;
        if (blocks > 16384) blocks = 16384;              // (grid-stride: a few iterations amortise the table set-up)
        KLAUNCH (h, k_acgt_pack, dim3 ((uint32_t)blocks), dim3 (256), 512, seq, n_bases, packed, pb, x, d);
    }
    HIPCHK (h, hipGetLastError ());
    HIPCHK (h, hipStreamSynchronize (h->stream));
    uint32_t f = 0;
    HIPCHK (h, hipMemcpy (&f, d, 4, hipMemcpyDeviceToHost));
    *has_x_host = f != 0;
    return gz_sync (h) < 0 ? GZ_ERR : GZ_OK;
}

extern "C" int gz_acgt_unpack (GzHandle *h, const uint8_t *packed, const uint8_t *x, uint64_t n_bases, uint8_t *seq)
{
    if (!h || (n_bases && (!seq || !packed))) return GZ_ERR_ARG;
    if (n_bases) {
        uint64_t blocks = ((n_bases + 15) / 16 + 255) / 256;
        if (blocks > 65536) blocks = 65536;
        KLAUNCH (h, k_acgt_unpack, dim3 ((uint32_t)blocks), dim3 (256), 0, packed, x, n_bases, seq);
    }
    HIPCHK (h, hipGetLastError ());
    HIPCHK (h, hipStreamSynchronize (h->stream));
    return gz_sync (h) < 0 ? GZ_ERR : GZ_OK;
}

// ---------------------------------------------------------------------------------------------------------
// seg-side appends, a column at a time (rows a1-a3; gz_kernels_seg.h)
// ---------------------------------------------------------------------------------------------------------
extern "C" int gz_ctx_seg_columns (GzHandle *h, const GzColumnJob *jobs, int n_jobs)
{
    if (!h || (n_jobs && !jobs) || n_jobs < 0 || n_jobs > 65535) return GZ_ERR_ARG;   // (one grid row per column)
    if (!n_jobs) return GZ_OK;
    HIPCHK (h, hipSetDevice (h->device));
    std::vector<GzdColumn> J (n_jobs);
    uint32_t max_n = 0, max_ol = 0, max_all = 0;
    // one allocation for all the tables: a single memset clears them
    std::vector<size_t> table_at (n_jobs);
    size_t table_words = 0;
    for (int i = 0; i < n_jobs; i++) {
        const GzColumnJob &u = jobs[i];
        if (!u.result_dev || u.n >= (1u << 30) || u.n_ol >= (1u << 30)) return GZ_ERR_ARG;
        if (u.n && (!u.off || !u.len || !u.node_index || !u.node_char_index || !u.node_snip_len || !u.counts || !u.b250)) return GZ_ERR_ARG;
        if (u.n_ol && (!u.ol_dict || !u.ol_char_index || !u.ol_snip_len || !u.counts)) return GZ_ERR_ARG;
        GzdColumn &d = J[i];
        memset (&d, 0, sizeof (d));
        d.text = u.text; d.off = u.off; d.len = u.len; d.n = u.n;
        d.ol_dict = u.ol_dict; d.ol_char_index = u.ol_char_index; d.ol_snip_len = u.ol_snip_len; d.n_ol = u.n_ol;
        d.node_index = u.node_index; d.dict = u.dict; d.dict_cap = u.dict ? u.dict_cap : 0;
        d.node_char_index = u.node_char_index; d.node_snip_len = u.node_snip_len; d.counts = u.counts; d.b250 = u.b250;
        d.result = u.result_dev;
        uint32_t bits = 4;                                         // at most half full
        while ((1ull << bits) < 2ull * ((uint64_t)u.n + u.n_ol)) bits++;
        d.table_bits = bits;
        table_at[i] = table_words; table_words += (size_t)1 << bits;
        const size_t tiles = ((size_t)u.n + GZ_COL_TILE - 1) / GZ_COL_TILE + 1;
        if (!(d.rep = (uint32_t *)arena_alloc (h, ((size_t)u.n + 1) * 4))) return GZ_ERR_HIP;
        if (!(d.rank = (uint32_t *)arena_alloc (h, ((size_t)u.n + 1) * 4))) return GZ_ERR_HIP;
        if (!(d.tile_a = (uint64_t *)arena_alloc (h, tiles * 8))) return GZ_ERR_HIP;
        if (!(d.tile_b = (uint64_t *)arena_alloc (h, tiles * 8))) return GZ_ERR_HIP;
        if (!(d.not_same = (uint32_t *)arena_alloc (h, 4))) return GZ_ERR_HIP;
        if (u.n > max_n) max_n = u.n;
        if (u.n_ol > max_ol) max_ol = u.n_ol;
        if (u.n + u.n_ol > max_all) max_all = u.n + u.n_ol;
    }
    uint32_t *tables = (uint32_t *)arena_alloc (h, table_words * 4);
    if (!tables) return GZ_ERR_HIP;
    for (int i = 0; i < n_jobs; i++) J[i].table = tables + table_at[i];
    HIPCHK (h, hipMemsetAsync (tables, 0xff, table_words * 4, h->stream));
    void *dj;
    int rc;
    if ((rc = upload (h, J.data (), J.size () * sizeof (GzdColumn), &dj)) != GZ_OK) return rc;
    GzdColumn *d_cols = (GzdColumn *)dj;
    const uint32_t t_n = (max_n + GZ_COL_TILE - 1) / GZ_COL_TILE, t_ol = (max_ol + 255) / 256, t_all = (max_all + 255) / 256;
    const dim3 g_n (t_n ? t_n : 1, n_jobs), g_jobs (n_jobs);
    KLAUNCH (h, k_col_clear, dim3 (t_all ? t_all : 1, n_jobs), dim3 (256), 0, d_cols);
    if (t_ol) KLAUNCH (h, k_col_insert_ol, dim3 (t_ol, n_jobs), dim3 (256), 0, d_cols);
    if (t_n) {
        KLAUNCH (h, k_col_insert, g_n, dim3 (256), 0, d_cols);
        KLAUNCH (h, k_col_first, g_n, dim3 (256), 2048, d_cols);
        KLAUNCH (h, k_col_scan_a, g_jobs, dim3 (256), 2048, d_cols);
        KLAUNCH (h, k_col_assign, g_n, dim3 (256), 2048, d_cols);
        KLAUNCH (h, k_col_node, g_n, dim3 (256), 2048, d_cols);
        KLAUNCH (h, k_col_counts, dim3 ((t_n + GZ_COUNT_TILES - 1) / GZ_COUNT_TILES, n_jobs), dim3 (256), GZ_COUNT_SLOTS * 8, d_cols);
        KLAUNCH (h, k_col_scan_b, g_jobs, dim3 (256), 2048, d_cols);
        KLAUNCH (h, k_col_b250, g_n, dim3 (256), 2048, d_cols);
    }
    HIPCHK (h, hipGetLastError ());
    return GZ_OK;
}

extern "C" int gz_dyn_int_columns (GzHandle *h, const GzDynIntJob *jobs, int n_jobs)
{
    if (!h || (n_jobs && !jobs) || n_jobs < 0 || n_jobs > 65535) return GZ_ERR_ARG;   // (one grid row per column)
    if (!n_jobs) return GZ_OK;
    HIPCHK (h, hipSetDevice (h->device));
    std::vector<GzdDynInt> J (n_jobs);
    uint64_t max_tiles = 1;
    for (int i = 0; i < n_jobs; i++) {
        const GzDynIntJob &u = jobs[i];
        if (!u.result_dev || (u.n && (!u.values || !u.out)) || ((uintptr_t)u.out & 7) || u.n >= (1ull << 40)) return GZ_ERR_ARG;
        GzdDynInt &d = J[i];
        d.values = u.values; d.is_nothing = u.is_nothing; d.n = u.n; d.nothing_char = u.nothing_char; d.out = u.out; d.result = u.result_dev;
        d.n_dev = u.n_dev;
        const uint64_t tiles = (u.n + GZ_DYN_TILE - 1) / GZ_DYN_TILE;
        if (!(d.tile_min = (int64_t *)arena_alloc (h, (tiles + 1) * 8))) return GZ_ERR_HIP;
        if (!(d.tile_max = (int64_t *)arena_alloc (h, (tiles + 1) * 8))) return GZ_ERR_HIP;
        if (tiles > max_tiles) max_tiles = tiles;
    }
    void *dj;
    int rc;
    if ((rc = upload (h, J.data (), J.size () * sizeof (GzdDynInt), &dj)) != GZ_OK) return rc;
    KLAUNCH (h, k_dyn_minmax, dim3 ((uint32_t)max_tiles, n_jobs), dim3 (256), 4096, (GzdDynInt *)dj);
    KLAUNCH (h, k_dyn_decide, dim3 (n_jobs), dim3 (256), 4096, (GzdDynInt *)dj);
    KLAUNCH (h, k_dyn_write, dim3 ((uint32_t)max_tiles, n_jobs), dim3 (256), 0, (GzdDynInt *)dj);
    HIPCHK (h, hipGetLastError ());
    return GZ_OK;
}

extern "C" int gz_local_blob_columns (GzHandle *h, const GzBlobJob *jobs, int n_jobs)
{
    if (!h || (n_jobs && !jobs) || n_jobs < 0 || n_jobs > 65535) return GZ_ERR_ARG;   // (one grid row per column)
    if (!n_jobs) return GZ_OK;
    HIPCHK (h, hipSetDevice (h->device));
    std::vector<GzdBlob> J (n_jobs);
    uint32_t max_tiles = 1;
    for (int i = 0; i < n_jobs; i++) {
        const GzBlobJob &u = jobs[i];
        if (!u.out_len_dev || (u.n && (!u.text || !u.off || !u.len || !u.out))) return GZ_ERR_ARG;
        GzdBlob &d = J[i];
        if (u.pre_len > 4 || (u.pad_to & (u.pad_to - 1)) || u.pad_to > 64) return GZ_ERR_ARG;
        d.text = u.text; d.off = u.off; d.len = u.len; d.n = u.n; d.add_nul = u.add_nul ? 1 : 0; d.out = u.out; d.out_len = u.out_len_dev;
        d.pre = (uint32_t)u.pre[0] | (uint32_t)u.pre[1] << 8 | (uint32_t)u.pre[2] << 16 | (uint32_t)u.pre[3] << 24; d.pre_len = u.pre_len;
        d.pad_mask = u.pad_to ? u.pad_to - 1 : 0; d.pad_byte = u.pad_byte; d.item_off = u.item_off; d.item_len = u.item_len;
        const uint32_t tiles = (u.n + GZ_COL_TILE - 1) / GZ_COL_TILE;
        if (!(d.tile = (uint64_t *)arena_alloc (h, ((size_t)tiles + 1) * 8))) return GZ_ERR_HIP;
        if (tiles > max_tiles) max_tiles = tiles;
    }
    void *dj;
    int rc;
    if ((rc = upload (h, J.data (), J.size () * sizeof (GzdBlob), &dj)) != GZ_OK) return rc;
    KLAUNCH (h, k_blob_sum, dim3 (max_tiles, n_jobs), dim3 (256), 2048, (GzdBlob *)dj);
    KLAUNCH (h, k_blob_scan, dim3 (n_jobs), dim3 (256), 2048, (GzdBlob *)dj);
    KLAUNCH (h, k_blob_copy, dim3 (max_tiles, n_jobs), dim3 (256), 2048, (GzdBlob *)dj);
    HIPCHK (h, hipGetLastError ());
    return GZ_OK;
}

extern "C" int gz_bam_records (GzHandle *h, const uint8_t *bam, uint64_t n_bytes, int32_t n_ref, uint32_t *rec_off, uint32_t cap, GzBamResult *result_dev)
{
    if (!h || !result_dev || (n_bytes && !bam) || (cap && !rec_off) || n_bytes >= 0xfffffff0ull || n_ref < 0) return GZ_ERR_ARG;
    HIPCHK (h, hipSetDevice (h->device));
    const uint32_t n_chunks = (uint32_t)((n_bytes + GZ_BAM_CHUNK - 1) / GZ_BAM_CHUNK);
    GzdBamChain B;
    B.bam = bam; B.n = n_bytes; B.rec_off = rec_off; B.cap = cap; B.result = result_dev; B.n_ref = n_ref;
    uint32_t *scr = (uint32_t *)arena_alloc (h, ((size_t)3 * (n_chunks + 1) + 4) * 4);
    B.tile = (uint64_t *)arena_alloc (h, ((size_t)n_chunks + 2) * 8);
    if (!scr || !B.tile) return GZ_ERR_HIP;
    B.entry = scr; B.exit_ = scr + (n_chunks + 1); B.count = scr + 2 * (size_t)(n_chunks + 1);
    uint32_t *first_wrong = scr + 3 * (size_t)(n_chunks + 1);
    HIPCHK (h, hipMemsetAsync (first_wrong, 0xff, 4, h->stream));
    if (n_chunks) {
        KLAUNCH (h, k_bam_entry, dim3 (n_chunks), dim3 (64), 0, B);
        KLAUNCH (h, k_bam_walk_count, dim3 ((n_chunks + 63) / 64), dim3 (64), 0, B, n_chunks);
        KLAUNCH (h, k_bam_check, dim3 ((n_chunks + 63) / 64), dim3 (64), 0, B, n_chunks, first_wrong);
    }
    KLAUNCH (h, k_bam_fix, dim3 (1), dim3 (256), 8192, B, n_chunks, (const uint32_t *)first_wrong);
    if (n_chunks) KLAUNCH (h, k_bam_walk_write, dim3 ((n_chunks + 63) / 64), dim3 (64), 0, B, n_chunks);
    HIPCHK (h, hipGetLastError ());
    return GZ_OK;
}

extern "C" int gz_bam_to_sam (GzHandle *h, const uint8_t *bam, uint64_t n_bytes, const uint32_t *rec_off, uint32_t n_rec,
                              const uint8_t *ref_names, const uint32_t *ref_name_off, int32_t n_ref,
                              uint8_t *text, uint64_t text_cap, uint32_t *line_off, GzBamResult *result_dev)
{
    if (!h || !result_dev || (n_rec && (!bam || !rec_off)) || (text_cap && !text) || n_ref < 0 || (n_ref && (!ref_names || !ref_name_off)) || n_bytes >= 0xfffffff0ull) return GZ_ERR_ARG;
    HIPCHK (h, hipSetDevice (h->device));
    GzdBamText T;
    T.bam = bam; T.n = n_bytes; T.rec_off = rec_off; T.n_rec = n_rec; T.ref_names = ref_names; T.ref_name_off = ref_name_off; T.n_ref = n_ref;
    T.text = text; T.text_cap = text_cap; T.line_off = line_off; T.result = result_dev;
    const uint32_t tiles = (n_rec + 255) / 256;
    T.len = (uint32_t *)arena_alloc (h, ((size_t)n_rec + 1) * 4);
    T.tile = (uint64_t *)arena_alloc (h, ((size_t)tiles + 1) * 8);
    if (!T.len || !T.tile) return GZ_ERR_HIP;
    HIPCHK (h, hipMemsetAsync (result_dev, 0xff, sizeof (GzBamResult), h->stream));     // (first_bad = none)
    if (tiles) KLAUNCH (h, k_bam_len, dim3 (tiles), dim3 (256), 2048, T);
    KLAUNCH (h, k_bam_scan, dim3 (1), dim3 (256), 2048, T);
    if (tiles) KLAUNCH (h, k_bam_write, dim3 (tiles), dim3 (256), 2048, T);
    HIPCHK (h, hipGetLastError ());
    return GZ_OK;
}

extern "C" int gz_text_lines (GzHandle *h, const uint8_t *text, uint64_t n_bytes, uint32_t *line_off, uint32_t *line_len, uint32_t cap,
                              GzLinesResult *result_dev)
{
    if (!h || !result_dev || (n_bytes && !text) || (cap && (!line_off || !line_len)) || n_bytes >= 0xffffffffull) return GZ_ERR_ARG;
    HIPCHK (h, hipSetDevice (h->device));
    GzdLines L;
    L.text = text; L.n = n_bytes; L.off = line_off; L.len = line_len; L.cap = cap; L.result = result_dev; L.byte4 = 0x0a0a0a0au; L.raw = 0;
    const uint32_t tiles = (uint32_t)((n_bytes + GZ_NL_TILE - 1) / GZ_NL_TILE);
    if (!(L.start = (uint32_t *)arena_alloc (h, ((size_t)cap + 2) * 4))) return GZ_ERR_HIP;
    if (!(L.tile = (uint64_t *)arena_alloc (h, ((size_t)tiles + 1) * 8))) return GZ_ERR_HIP;
    if (tiles) KLAUNCH (h, k_nl_count, dim3 (tiles), dim3 (256), 2048, L);
    KLAUNCH (h, k_nl_scan, dim3 (1), dim3 (256), 2048, L);
    if (tiles) KLAUNCH (h, k_nl_write, dim3 (tiles), dim3 (256), 2048, L);
    if (cap) KLAUNCH (h, k_lines_finish, dim3 ((cap + 255) / 256), dim3 (256), 0, L);
    HIPCHK (h, hipGetLastError ());
    return GZ_OK;
}

// positions of one separator byte (N1 for tab-separated data types): the newline machinery with another byte and no line rules
extern "C" int gz_byte_index (GzHandle *h, const uint8_t *text, uint64_t n_bytes, uint8_t byte, uint32_t *after, uint32_t cap, GzLinesResult *result_dev)
{
    if (!h || !result_dev || !after || (n_bytes && !text) || n_bytes >= 0xffffffffull) return GZ_ERR_ARG;
    HIPCHK (h, hipSetDevice (h->device));
    GzdLines L; memset (&L, 0, sizeof (L));
    L.text = text; L.n = n_bytes; L.cap = cap; L.result = result_dev; L.byte4 = 0x01010101u * byte; L.raw = 1; L.start = after;
    const uint32_t tiles = (uint32_t)((n_bytes + GZ_NL_TILE - 1) / GZ_NL_TILE);
    if (!(L.tile = (uint64_t *)arena_alloc (h, ((size_t)tiles + 1) * 8))) return GZ_ERR_HIP;
    if (tiles) KLAUNCH (h, k_nl_count, dim3 (tiles), dim3 (256), 2048, L);
    KLAUNCH (h, k_nl_scan, dim3 (1), dim3 (256), 2048, L);
    if (tiles) KLAUNCH (h, k_nl_write, dim3 (tiles), dim3 (256), 2048, L);
    HIPCHK (h, hipGetLastError ());
    return GZ_OK;
}

extern "C" int gz_fastq_records (GzHandle *h, const uint8_t *text, const uint32_t *line_off, const uint32_t *line_len,
                                 const GzLinesResult *lines_dev, uint32_t max_reads,
                                 uint32_t *l1_off, uint32_t *l1_len, uint32_t *seq_off, uint32_t *seq_len,
                                 uint32_t *l3_off, uint32_t *l3_len, uint32_t *qual_off, uint32_t *qual_len, GzFastqResult *result_dev)
{
    if (!h || !result_dev || !lines_dev) return GZ_ERR_ARG;
    if (max_reads && (!text || !line_off || !line_len || !l1_off || !l1_len || !seq_off || !seq_len || !l3_off || !l3_len || !qual_off || !qual_len)) return GZ_ERR_ARG;
    HIPCHK (h, hipSetDevice (h->device));
    GzdFastq F;
    F.text = text; F.line_off = line_off; F.line_len = line_len; F.lines = lines_dev; F.max_reads = max_reads; F.result = result_dev;
    uint32_t *cols[8] = { l1_off, l1_len, seq_off, seq_len, l3_off, l3_len, qual_off, qual_len };
    for (int i = 0; i < 8; i++) F.col[i] = cols[i];
    HIPCHK (h, hipMemsetAsync (result_dev, 0xff, sizeof (GzFastqResult), h->stream));   // first_bad = none (the kernel sets the rest)
    KLAUNCH (h, k_fastq_records, dim3 (max_reads ? (max_reads + 255) / 256 : 1), dim3 (256), 0, F);
    HIPCHK (h, hipGetLastError ());
    return GZ_OK;
}

extern "C" int gz_tokenize_column (GzHandle *h, const uint8_t *text, const uint32_t *off, const uint32_t *len, uint32_t n,
                                   const char *seps, uint32_t n_seps, uint32_t *item_off, uint32_t *item_len, uint32_t *n_bad_dev)
{
    if (!h || !n_bad_dev || n_seps > GZ_TOK_MAX_SEPS || (n_seps && !seps) || (n && (!text || !off || !len || !item_off || !item_len))) return GZ_ERR_ARG;
    HIPCHK (h, hipSetDevice (h->device));
    GzdTokens T;
    memset (&T, 0, sizeof (T));
    T.text = text; T.off = off; T.len = len; T.n = n; T.n_seps = n_seps; T.item_off = item_off; T.item_len = item_len; T.n_bad = n_bad_dev;
    for (uint32_t i = 0; i < n_seps; i++) T.seps[i] = (uint8_t)seps[i];
    HIPCHK (h, hipMemsetAsync (n_bad_dev, 0, 4, h->stream));
    if (n) KLAUNCH (h, k_tokenize, dim3 ((n + 255) / 256), dim3 (256), 0, T);
    HIPCHK (h, hipGetLastError ());
    return GZ_OK;
}

extern "C" int gz_seg_integer_or_not (GzHandle *h, const uint8_t *text, const uint32_t *off, const uint32_t *len, uint32_t n,
                                      uint32_t nothing_char, uint32_t lookup_off, uint32_t *snip_off, uint32_t *snip_len,
                                      int64_t *values, uint8_t *is_nothing, uint64_t *n_values_dev)
{
    if (!h || !n_values_dev || (n && (!text || !off || !len || !snip_off || !snip_len || !values || !is_nothing))) return GZ_ERR_ARG;
    HIPCHK (h, hipSetDevice (h->device));
    GzdIntSplit S;
    S.text = text; S.off = off; S.len = len; S.n = n; S.nothing_char = nothing_char; S.lookup_off = lookup_off;
    S.snip_off = snip_off; S.snip_len = snip_len; S.values = values; S.is_nothing = is_nothing; S.n_values = n_values_dev;
    const uint32_t tiles = (n + 255) / 256;
    if (!(S.tile = (uint64_t *)arena_alloc (h, ((size_t)tiles + 1) * 8))) return GZ_ERR_HIP;
    if (tiles) KLAUNCH (h, k_int_count, dim3 (tiles), dim3 (256), 2048, S);
    KLAUNCH (h, k_int_scan, dim3 (1), dim3 (256), 2048, S);
    if (tiles) KLAUNCH (h, k_int_write, dim3 (tiles), dim3 (256), 2048, S);
    HIPCHK (h, hipGetLastError ());
    return GZ_OK;
}

// ---------------------------------------------------------------------------------------------------------
// VBlock decode: the sections of many VBlocks found and checked by two kernels, their payloads decoded in ONE batch
// ---------------------------------------------------------------------------------------------------------
extern "C" int gz_vb_uncompress_many (GzHandle *h, int n_vbs, const uint8_t *const *z_data, const uint64_t *z_len, uint8_t *const *out,
                                      const uint64_t *out_cap, uint64_t *section_offsets_host, uint32_t max_sections, uint32_t *n_sections_out)
{
    if (!h || n_vbs < 0 || !max_sections || (n_vbs && (!z_data || !z_len || !out || !out_cap || !n_sections_out))) return GZ_ERR_ARG;
    int rc;
    if ((rc = gz_sync (h)) < 0) return rc;
    if (!n_vbs) return GZ_OK;
    std::vector<GzdVbWalk> W ((size_t)n_vbs);
    for (int v = 0; v < n_vbs; v++) {
        if (!z_data[v] || z_len[v] < 84) return GZ_ERR_ARG;
        W[v].z = z_data[v]; W[v].z_len = z_len[v]; W[v].out_cap = out_cap[v]; W[v].n_sections = 0; W[v].status = GZ_ST_CORRUPT;
    }
    void *d_w;
    if ((rc = upload (h, W.data (), W.size () * sizeof (GzdVbWalk), &d_w)) != GZ_OK) return rc;
    const size_t n_secs = (size_t)n_vbs * max_sections;
    GzdVbSec *d_s = (GzdVbSec *)arena_alloc (h, n_secs * sizeof (GzdVbSec));
    if (!d_s) return GZ_ERR_HIP;
    KLAUNCH (h, k_vb_walk, dim3 (((uint32_t)n_vbs + 63) / 64), dim3 (64), 0, (GzdVbWalk *)d_w, d_s, (uint32_t)n_vbs, max_sections);
    KLAUNCH (h, k_vb_digests, dim3 (max_sections, (uint32_t)n_vbs), dim3 (256), 4096, (const GzdVbWalk *)d_w, d_s, max_sections);
    HIPCHK (h, hipGetLastError ());
    HIPCHK (h, hipStreamSynchronize (h->stream));
    std::vector<GzdVbSec> S (n_secs);
    HIPCHK (h, hipMemcpy (W.data (), d_w, W.size () * sizeof (GzdVbWalk), hipMemcpyDeviceToHost));
    HIPCHK (h, hipMemcpy (S.data (), d_s, n_secs * sizeof (GzdVbSec), hipMemcpyDeviceToHost));
    std::vector<GzStream> work;
    for (int v = 0; v < n_vbs; v++) {
        if (W[v].status != GZ_ST_OK) { h->err = "bad VB header / section magic / section overflow"; return GZ_ERR_CORRUPT; }
        n_sections_out[v] = W[v].n_sections;
        uint64_t o = 0;
        uint64_t *offs = section_offsets_host ? section_offsets_host + (size_t)v * (max_sections + 1) : NULL;
        for (uint32_t i = 0; i < W[v].n_sections; i++) {
            const GzdVbSec &sec = S[(size_t)v * max_sections + i];
            if (!sec.ok) { h->err = "section adler32 mismatch"; return GZ_ERR_CORRUPT; }
            if (offs) offs[i] = o;
            const int sc = (int)sec.codec;
            if (sec.ulen && codec_ok (sc)) {
                GzStream s; memset (&s, 0, sizeof (s));
                s.in = z_data[v] + sec.at; s.in_len = sec.clen; s.out = out[v] + o; s.out_cap = sec.ulen; s.codec = sc;
                work.push_back (s);
            }
            else if (sec.ulen && sc > 0 && sc < GZ_NUM_CODECS) {
                // a codec of the file format the device has no decoder for - the host's sequential coders (BZ2 / LZMA / BSC: SURVEY 2.1), a complex
                // codec with an uncompress of its own (CODEC_ACGT: this library's own NONREF section, codec_acgt_uncompress runs the LZMA sub-codec
                // and then unpacks - src/codec.h:108, compressor.c:210-236), anything else of src/genozip.h:326-360 - is left to the caller's
                // codec_args[codec].uncompress: its stretch of `out` is zeroed and its offset carries GZ_SECTION_NOT_DECODED
                if (o + sec.ulen > out_cap[v]) { h->err = "output too small"; return GZ_ERR_CORRUPT; }
                HIPCHK (h, hipMemsetAsync (out[v] + o, 0, sec.ulen, h->stream));
                if (offs) offs[i] |= GZ_SECTION_NOT_DECODED;
            }
            else if (sec.ulen) { h->err = "section with a codec byte the file format does not have"; return GZ_ERR_CORRUPT; }
            o += sec.ulen;
        }
        if (offs) offs[W[v].n_sections] = o;
    }
    if (!work.empty ()) {
        if ((rc = gz_codec_uncompress_batch (h, work.data (), (int)work.size ())) != GZ_OK) return rc;
        if ((rc = gz_sync (h)) < 0) return rc;
        for (auto &s : work) if (s.status != GZ_OK) { h->err = "section payload corrupt"; return GZ_ERR_CORRUPT; }
    }
    return GZ_OK;
}

extern "C" int gz_vb_uncompress (GzHandle *h, const uint8_t *z_data, uint64_t z_len, uint8_t *out, uint64_t out_cap,
                                 uint64_t *section_offsets_host, uint32_t max_sections, uint32_t *n_sections_out)
{
    if (!h || !z_data || !n_sections_out || z_len < 84) return GZ_ERR_ARG;
    return gz_vb_uncompress_many (h, 1, &z_data, &z_len, &out, &out_cap, section_offsets_host, max_sections, n_sections_out);
}

#include "gz_zip.h"
#include "gz_global.h"
