// What scattered record stores / position stores cost on MI355X, and what rocprofv3's WRITE_SIZE / FETCH_SIZE report for them
// (the guide: "WRITE_SIZE is uncalibrated: calibrate on a known byte count in your own access pattern").
// The access patterns are the model kernel's: wave (c, b) - context c of S, batch b - stores the records of positions (64 b + lane) S + c,
// i.e. S waves fill S neighbouring records each; grid order decides whether those S waves sit on one XCD (one L2) or on S of them.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_scatter.hip -o tools/ubench_scatter.bin     run: tools/ubench_scatter.bin [MiB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf ("%s: %s\n", #x, hipGetErrorString (e)); return 1; } } while (0)

// one wave per workgroup: block id -> (context, batch); same_xcd: the S contexts of a batch take block ids 8 apart (one XCD), else adjacent ids
template <typename T> __global__ void __launch_bounds__(64) k_scatter_store (T *out, uint32_t n_rec, uint32_t S, int same_xcd, uint32_t per_wave)
{
    const uint32_t id = blockIdx.x, n_batches = n_rec / (64 * S * per_wave);
    uint32_t c, b;
    if (same_xcd) { const uint32_t x = id % 8, r = id / 8; c = r % S; b = (r / S) * 8 + x; }      // ids x, x + 8, x + 16 .. share an XCD
    else          { c = id % S; b = id / S; }
    if (b >= n_batches) return;
    for (uint32_t k = 0; k < per_wave; k++) {
        const uint32_t pos = ((b * per_wave + k) * 64 + threadIdx.x) * S + c;
        T v; uint32_t *w = (uint32_t *)&v;
        for (uint32_t i = 0; i < sizeof (T) / 4; i++) w[i] = pos + i;
        out[pos] = v;
    }
}

template <typename T> __global__ void __launch_bounds__(64) k_scatter_load (const T *in, uint32_t n_rec, uint32_t S, int same_xcd, uint32_t per_wave, uint32_t *sink)
{
    const uint32_t id = blockIdx.x, n_batches = n_rec / (64 * S * per_wave);
    uint32_t c, b;
    if (same_xcd) { const uint32_t x = id % 8, r = id / 8; c = r % S; b = (r / S) * 8 + x; }
    else          { c = id % S; b = id / S; }
    if (b >= n_batches) return;
    uint32_t acc = 0;
    for (uint32_t k = 0; k < per_wave; k++) {
        const uint32_t pos = ((b * per_wave + k) * 64 + threadIdx.x) * S + c;
        const T v = in[pos]; const uint32_t *w = (const uint32_t *)&v;
        for (uint32_t i = 0; i < sizeof (T) / 4; i++) acc += w[i];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

struct R16 { uint32_t w[4]; };
struct R8  { uint32_t w[2]; };
struct R4  { uint32_t w[1]; };

template <typename T> static int run (const char *name, void *buf, size_t bytes, uint32_t S, int same_xcd, bool load, uint32_t *sink)
{
    const uint32_t per_wave = 16;
    const uint32_t n_rec = (uint32_t)(bytes / sizeof (T)) / (64 * S * per_wave * 8) * (64 * S * per_wave * 8);
    const uint32_t grid = n_rec / (64 * per_wave);
    hipEvent_t e0, e1; CHK (hipEventCreate (&e0)); CHK (hipEventCreate (&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CHK (hipEventRecord (e0, 0));
        if (load) hipLaunchKernelGGL (k_scatter_load<T>, dim3 (grid), dim3 (64), 0, 0, (const T *)buf, n_rec, S, same_xcd, per_wave, sink);
        else      hipLaunchKernelGGL (k_scatter_store<T>, dim3 (grid), dim3 (64), 0, 0, (T *)buf, n_rec, S, same_xcd, per_wave);
        CHK (hipEventRecord (e1, 0)); CHK (hipEventSynchronize (e1));
        float ms; CHK (hipEventElapsedTime (&ms, e0, e1)); if (rep && ms < best) best = ms;
    }
    printf ("%-6s %-5s %2u B records, stride %3u, %s: %8.1f MB in %7.3f ms = %7.1f GB/s\n", load ? "load" : "store", name, (unsigned)sizeof (T), S,
            same_xcd ? "neighbours on one XCD " : "neighbours on S XCDs  ", (double)n_rec * sizeof (T) / 1e6, best, (double)n_rec * sizeof (T) / 1e6 / best);
    return 0;
}

int main (int argc, char **argv)
{
    const size_t bytes = (size_t)(argc > 1 ? atoi (argv[1]) : 1024) << 20;
    void *buf; uint32_t *sink;
    CHK (hipMalloc (&buf, bytes)); CHK (hipMalloc ((void **)&sink, 64)); CHK (hipMemset (buf, 1, bytes));
    const uint32_t strides[] = { 1, 4, 16, 40 };
    for (int load = 0; load < 2; load++)
        for (uint32_t S : strides)
            for (int same = 0; same < (S > 1 ? 2 : 1); same++) {
                if (run<R16> ("rec16", buf, bytes, S, same, load, sink)) return 1;
                if (run<R8>  ("rec8",  buf, bytes, S, same, load, sink)) return 1;
                if (run<R4>  ("pos4",  buf, bytes, S, same, load, sink)) return 1;
            }
    return 0;
}
