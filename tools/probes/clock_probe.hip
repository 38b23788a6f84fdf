// what clock does a nearly idle MI355X give ONE wave? s_memtime (shader clock) against s_memrealtime (100 MHz) around a chain of dependent
// integer adds, alone and beside a grid that keeps every compute unit busy:   hipcc --offload-arch=gfx950 -O2 clock_probe.hip -o clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k_chain (uint64_t *out, uint32_t iters)
{
    uint32_t v = threadIdx.x;
    const uint64_t c0 = clock64 (), w0 = wall_clock64 ();
    for (uint32_t i = 0; i < iters; i++) { asm volatile ("v_add_u32 %0, %0, %0\n\tv_xor_b32 %0, %0, 1\n\tv_add_u32 %0, %0, 3\n\tv_xor_b32 %0, %0, 5" : "+v" (v)); }
    const uint64_t c1 = clock64 (), w1 = wall_clock64 ();
    if (!threadIdx.x && !blockIdx.x) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = v; }
}
__global__ void k_busy (uint32_t *sink, uint32_t iters)
{
    float a = threadIdx.x, b = 1.0001f;
    for (uint32_t i = 0; i < iters; i++) { a = a * b + 0.5f; b = b * 0.9999f + a * 1e-9f; }
    if (a == 12345.f) sink[0] = 1;
}
int main ()
{
    uint64_t *d, h[3]; uint32_t *sink;
    hipMalloc (&d, 24); hipMalloc (&sink, 4);
    hipStream_t s1, s2; hipStreamCreate (&s1); hipStreamCreate (&s2);
    for (int busy = 0; busy < 2; busy++)
        for (int rep = 0; rep < 3; rep++) {
            if (busy) hipLaunchKernelGGL (k_busy, dim3 (256 * 8), dim3 (256), 0, s2, sink, 40000000u);
            hipLaunchKernelGGL (k_chain, dim3 (1), dim3 (64), 0, s1, d, 20000000u);
            hipStreamSynchronize (s1);
            hipMemcpy (h, d, 24, hipMemcpyDeviceToHost);
            printf ("%s: 80 M dependent VALU ops: %llu shader clocks, %llu ref ticks (100 MHz) = %.1f ms -> shader clock %.0f MHz, %.2f clocks / op, %.2f ns / op\n",
                    busy ? "beside a busy grid" : "alone", (unsigned long long)h[0], (unsigned long long)h[1], h[1] / 1e5, h[0] / (h[1] / 100.0), h[0] / 8e7, h[1] * 10.0 / 8e7);
            hipDeviceSynchronize ();
        }
    return 0;
}
