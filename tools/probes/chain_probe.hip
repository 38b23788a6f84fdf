// what paces the range-coder chain's three dependent instructions per symbol (gz_chain_asm.h: v_fma_f64, v_fma_f64, v_and_or_b32 =
// 15.8 clocks)? The same dependency pattern in loop bodies of different lengths and encodings, one wave on an idle chip.
//   hipcc --offload-arch=gfx950 -O2 chain_probe.hip -o chain_probe && ./chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define R2(x) x x
#define R4(x) R2(x) R2(x)
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)
#define R256(x) R64(x) R64(x) R64(x) R64(x)
// the whole loop is one asm statement over fixed registers (as in gz_chain_asm.h): R v[60:61], T v[62:63], inv v[64:65], 2^52 v[56:57], F v[68:69], G v[70:71]
#define PRO "v_mov_b32 v60, 0\n v_mov_b32 v61, 0x41700000\n v_mov_b32 v64, 0\n v_mov_b32 v65, 0x3f900000\n v_mov_b32 v56, 0\n v_mov_b32 v57, 0x43300000\n" \
            "v_mov_b32 v68, 0\n v_mov_b32 v69, 0x3ff40000\n v_mov_b32 v70, 0\n v_mov_b32 v71, 0xc3340000\n v_mov_b32 v58, 0x7fffff\n v_mov_b32 v59, 0x41000000\n" \
            "s_mov_b32 s40, %1\n s_memtime s[42:43]\n s_waitcnt lgkmcnt(0)\n1:\n"
#define EPI "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 1b\n s_memtime s[44:45]\n s_waitcnt lgkmcnt(0)\n s_sub_u32 %0, s44, s42\n"
#define KERNEL(name, body, nsym)                                                                                     \
__global__ void name (uint64_t *out, uint32_t iters)                                                                 \
{                                                                                                                    \
    uint32_t clocks;                                                                                                 \
    asm volatile (PRO body EPI : "=s" (clocks) : "s" (iters) : "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v68", "v69", "v70", "v71", \
                  "s40", "s42", "s43", "s44", "s45", "scc", "memory");                                                \
    if (!threadIdx.x) { out[0] = clocks; out[2] = nsym; }                                                            \
}
#define S3 "v_fma_f64 v[62:63], v[60:61], v[64:65], v[56:57]\n v_fma_f64 v[60:61], v[62:63], v[68:69], v[70:71]\n v_and_or_b32 v61, v61, v58, v59\n"
#define S4 "v_fma_f64 v[62:63], v[60:61], v[64:65], v[56:57]\n v_fma_f64 v[60:61], v[62:63], v[68:69], v[70:71]\n v_and_b32 v61, v61, v58\n v_or_b32 v61, v61, v59\n"
#define S2 "v_fma_f64 v[62:63], v[60:61], v[64:65], v[56:57]\n v_fma_f64 v[60:61], v[62:63], v[68:69], v[70:71]\n"
#define SN "v_fma_f64 v[62:63], v[60:61], v[64:65], v[56:57]\n v_fma_f64 v[60:61], v[62:63], v[68:69], v[70:71]\n v_and_or_b32 v61, v61, v58, v59\n s_nop 0\n"
#define SB "v_fma_f64 v[62:63], v[60:61], v[64:65], v[56:57]\n v_fma_f64 v[60:61], v[62:63], v[68:69], v[70:71]\n v_bfi_b32 v61, v58, v61, v59\n"
#define SI "v_fma_f64 v[62:63], v[60:61], v[64:65], v[56:57]\n v_mov_b32 v72, v73\n v_fma_f64 v[60:61], v[62:63], v[68:69], v[70:71]\n v_mov_b32 v74, v75\n v_and_or_b32 v61, v61, v58, v59\n v_mov_b32 v76, v77\n"
KERNEL (k_s3_8, R8 (S3), 8)
KERNEL (k_s3_64, R64 (S3), 64)
KERNEL (k_s3_512, R256 (S3) R256 (S3), 512)
KERNEL (k_s4_512, R256 (S4) R256 (S4), 512)
KERNEL (k_s2_512, R256 (S2) R256 (S2), 512)
KERNEL (k_sn_512, R256 (SN) R256 (SN), 512)
KERNEL (k_sb_512, R256 (SB) R256 (SB), 512)
KERNEL (k_si_512, R256 (SI) R256 (SI), 512)
struct { const char *name; void (*k) (uint64_t *, uint32_t); uint32_t iters; } P[] = {
    { "fma, fma, and_or x 8 per loop iteration", k_s3_8, 400000 },
    { "fma, fma, and_or x 64", k_s3_64, 50000 },
    { "fma, fma, and_or x 512 (the chain's block)", k_s3_512, 6000 },
    { "fma, fma, and, or (two 4-byte instructions) x 512", k_s4_512, 6000 },
    { "fma, fma x 512 (no exponent step)", k_s2_512, 6000 },
    { "fma, fma, and_or, s_nop x 512", k_sn_512, 6000 },
    { "fma, fma, bfi x 512", k_sb_512, 6000 },
    { "fma, fma, and_or each followed by an independent v_mov x 512", k_si_512, 6000 },
};
int main ()
{
    uint64_t *d, h[3];
    (void)hipMalloc (&d, 24);
    for (unsigned i = 0; i < sizeof (P) / sizeof (P[0]); i++) {
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL (P[i].k, dim3 (1), dim3 (64), 0, 0, d, P[i].iters); (void)hipDeviceSynchronize (); }
        (void)hipMemcpy (h, d, 24, hipMemcpyDeviceToHost);
        printf ("%-62s %7.2f clocks per symbol\n", P[i].name, (double)(uint32_t)h[0] / ((double)P[i].iters * (double)h[2]));
    }
    return 0;
}
