// micro-benchmark: cost of dependent scalar / vector instruction chains in a single wave on gfx950
// (informs the design of the serial coder chains; see DESIGN.md "what a single wave can do")
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// scalar dependent chain: x = mulhi(x, m) + c, 4 instrs per iteration (mulhi, add, xor, or) all dependent
__global__ void k_salu_dep (uint32_t *out, uint32_t seed, int iters)
{
    uint32_t x = __builtin_amdgcn_readfirstlane (seed), m = 0x9E3779B9u;
    for (int i = 0; i < iters; i++) {
        x = __umulhi (x, m) + 12345u; x ^= 0x55aa55aau; x |= 1u; x += 7u;
        x = __umulhi (x, m) + 12345u; x ^= 0x55aa55aau; x |= 1u; x += 7u;
        x = __umulhi (x, m) + 12345u; x ^= 0x55aa55aau; x |= 1u; x += 7u;
        x = __umulhi (x, m) + 12345u; x ^= 0x55aa55aau; x |= 1u; x += 7u;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
// two independent scalar chains interleaved
__global__ void k_salu_2 (uint32_t *out, uint32_t seed, int iters)
{
    uint32_t x = __builtin_amdgcn_readfirstlane (seed), y = x ^ 0x1234567u, m = 0x9E3779B9u;
    for (int i = 0; i < iters; i++) {
        x = __umulhi (x, m) + 12345u; y = __umulhi (y, m) + 54321u; x ^= 0x55aa55aau; y ^= 0x33cc33ccu; x |= 1u; y |= 1u; x += 7u; y += 9u;
        x = __umulhi (x, m) + 12345u; y = __umulhi (y, m) + 54321u; x ^= 0x55aa55aau; y ^= 0x33cc33ccu; x |= 1u; y |= 1u; x += 7u; y += 9u;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = x + y;
}
// vector dependent chain (per lane different values so it stays on the VALU)
__global__ void k_valu_dep (uint32_t *out, uint32_t seed, int iters)
{
    uint32_t x = seed + threadIdx.x, m = 0x9E3779B9u;
    for (int i = 0; i < iters; i++) {
        x = __umulhi (x, m) + 12345u; x ^= 0x55aa55aau; x |= 1u; x += 7u;
        x = __umulhi (x, m) + 12345u; x ^= 0x55aa55aau; x |= 1u; x += 7u;
        x = __umulhi (x, m) + 12345u; x ^= 0x55aa55aau; x |= 1u; x += 7u;
        x = __umulhi (x, m) + 12345u; x ^= 0x55aa55aau; x |= 1u; x += 7u;
    }
    out[blockIdx.x * 64 + threadIdx.x] = x;
}
// scalar chain with a data dependent branch every 5 instructions
__global__ void k_salu_br (uint32_t *out, uint32_t seed, int iters)
{
    uint32_t x = __builtin_amdgcn_readfirstlane (seed), m = 0x9E3779B9u, acc = 0;
    for (int i = 0; i < iters; i++) {
        x = __umulhi (x, m) + 12345u; x ^= 0x55aa55aau; x |= 1u; x += 7u;
        if (x & 0x100) { acc += x >> 3; acc ^= x; }
        x = __umulhi (x, m) + 12345u; x ^= 0x55aa55aau; x |= 1u; x += 7u;
        if (x & 0x200) { acc += x >> 5; acc ^= x; }
    }
    if (threadIdx.x == 0) out[blockIdx.x] = x + acc;
}
// readlane fed scalar chain: v holds per-lane values, chain reads lane k
__global__ void k_readlane (uint32_t *out, const uint32_t *in, int iters)
{
    uint32_t v = in[threadIdx.x], x = 1;
    for (int i = 0; i < iters; i++)
        for (int k = 0; k < 64; k++) { uint32_t a = (uint32_t)__builtin_amdgcn_readlane ((int)v, k); x = __umulhi (x | 0x80000000u, a) + a; }
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}

template <typename F> static float time_it (F f)
{
    hipEvent_t a, b; hipEventCreate (&a); hipEventCreate (&b);
    f (); hipDeviceSynchronize ();
    hipEventRecord (a); f (); hipEventRecord (b); hipEventSynchronize (b);
    float ms; hipEventElapsedTime (&ms, a, b); return ms;
}

int main ()
{
    uint32_t *out, *in; CHK (hipMalloc (&out, 1 << 20)); CHK (hipMalloc (&in, 256));
    CHK (hipMemset (in, 0x5b, 256));
    const int iters = 200000;
    for (int blocks : { 1, 176, 1024, 4096 }) {
        float a = time_it ([&] { hipLaunchKernelGGL (k_salu_dep, dim3 (blocks), dim3 (64), 0, 0, out, 12345u, iters); });
        float b = time_it ([&] { hipLaunchKernelGGL (k_salu_2, dim3 (blocks), dim3 (64), 0, 0, out, 12345u, iters); });
        float c = time_it ([&] { hipLaunchKernelGGL (k_valu_dep, dim3 (blocks), dim3 (64), 0, 0, out, 12345u, iters); });
        float d = time_it ([&] { hipLaunchKernelGGL (k_salu_br, dim3 (blocks), dim3 (64), 0, 0, out, 12345u, iters); });
        float e = time_it ([&] { hipLaunchKernelGGL (k_readlane, dim3 (blocks), dim3 (64), 0, 0, out, in, iters / 64); });
        printf ("blocks %4d: salu_dep %.2f ns/instr | salu_2chains %.2f ns/instr | valu_dep %.2f ns/instr | salu+branch %.2f ns/iter(~14 instr) | readlane-chain %.2f ns/elem (3 instr)\n",
                blocks, a * 1e6 / (iters * 16.0), b * 1e6 / (iters * 16.0), c * 1e6 / (iters * 16.0), d * 1e6 / (iters * 1.0), e * 1e6 / ((iters / 64) * 64.0));
    }
    return 0;
}
