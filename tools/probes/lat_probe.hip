// dependent-chain cost of the instruction sequences a one-wave serial decoder is made of (gfx950, one wave on an idle chip), in shader clocks per
// link of the chain:   hipcc --offload-arch=gfx950 -O2 lat_probe.hip -o lat_probe && ./lat_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#define REP8(x) x x x x x x x x
#define ITERS 200000u
// body: one asm string with 8 links of the chain; v = "+v" carried VGPR, s = "+s" carried SGPR
#define PROBE(name, body, links)                                                                               \
__global__ void name (uint64_t *out, uint32_t *lds_init)                                                       \
{                                                                                                              \
    __shared__ uint32_t lds[1024];                                                                             \
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (i * 4) & 0xfff;                                     \
    __syncthreads ();                                                                                          \
    uint32_t v = threadIdx.x & 3, s = 1, t = 0; double d = 1.5;                                                 \
    uint64_t m = 0;                                                                                            \
    const uint64_t c0 = clock64 ();                                                                            \
    for (uint32_t i = 0; i < ITERS; i++) { asm volatile (body : "+v" (v), "+s" (s), "+v" (t), "+v" (d), "+s" (m) : : "vcc", "scc", "memory"); } \
    const uint64_t c1 = clock64 ();                                                                            \
    if (!threadIdx.x) { out[0] = c1 - c0; out[1] = v + s + t + (uint32_t)d + (uint32_t)m; out[2] = links; }     \
}
PROBE (p_empty, "s_nop 0\n", 1)
PROBE (p_valu, REP8 ("v_add_u32 %0, %0, 1\n"), 8)
PROBE (p_salu, REP8 ("s_add_u32 %1, %1, 1\n"), 8)
PROBE (p_smul, REP8 ("s_mul_i32 %1, %1, 3\n"), 8)
PROBE (p_vmullo, REP8 ("v_mul_lo_u32 %0, %0, 3\n"), 8)
PROBE (p_vmul24, REP8 ("v_mul_u32_u24 %0, %0, 3\n"), 8)
PROBE (p_readlane_mov, REP8 ("v_readlane_b32 %1, %0, 3\n v_mov_b32 %0, %1\n"), 8)
PROBE (p_readfirst_mov, REP8 ("v_readfirstlane_b32 %1, %0\n v_mov_b32 %0, %1\n"), 8)
PROBE (p_salu_readlane_idx, REP8 ("s_and_b32 %1, %1, 3\n s_nop 3\n v_readlane_b32 %1, %0, %1\n"), 8)
PROBE (p_cmp_ff1_readlane, REP8 ("v_cmp_eq_u32 vcc, 3, %0\n s_ff1_i32_b64 %1, vcc\n s_nop 3\n v_readlane_b32 %1, %0, %1\n v_mov_b32 %0, %1\n"), 8)
PROBE (p_cmp_cndmask, REP8 ("v_cmp_lt_u32 vcc, 1, %0\n v_cndmask_b32 %0, %0, %2, vcc\n"), 8)
PROBE (p_scmp_cselect_cndmask, REP8 ("s_cmp_lt_u32 %1, 5\n s_cselect_b64 vcc, -1, 0\n v_cndmask_b32 %0, %0, %2, vcc\n v_readfirstlane_b32 %1, %0\n"), 8)
PROBE (p_ds_read_chain, REP8 ("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n"), 8)
PROBE (p_ds_write_read, REP8 ("ds_write_b32 %0, %0\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n"), 8)
PROBE (p_cvt_mul_cvt, REP8 ("v_cvt_f64_u32 %3, %0\n v_mul_f64 %3, %3, 1.0\n v_cvt_u32_f64 %0, %3\n"), 8)
PROBE (p_fma64, REP8 ("v_fma_f64 %3, %3, 1.0, %3\n"), 8)
PROBE (p_rcp64, REP8 ("v_rcp_f64 %3, %3\n"), 8)
PROBE (p_branch_taken, REP8 ("s_cmp_lg_u32 %1, 0\n s_cbranch_scc1 1f\n s_add_u32 %1, %1, 7\n1:\n s_add_u32 %1, %1, 1\n"), 8)
PROBE (p_branch_not_taken, REP8 ("s_cmp_eq_u32 %1, 0\n s_cbranch_scc1 1f\n s_add_u32 %1, %1, 1\n1:\n"), 8)
PROBE (p_vbranch_vccz, REP8 ("v_cmp_eq_u32 vcc, 77, %0\n s_cbranch_vccnz 1f\n v_add_u32 %0, %0, 1\n1:\n"), 8)
PROBE (p_ballot_branch, REP8 ("v_cmp_ne_u32 vcc, 77, %0\n s_cmp_eq_u64 vcc, 0\n s_cbranch_scc1 1f\n v_add_u32 %0, %0, 1\n1:\n"), 8)
PROBE (p_saveexec, REP8 ("v_cmp_lt_u32 vcc, 1, %0\n s_and_saveexec_b64 %4, vcc\n v_add_u32 %0, %0, 1\n s_or_b64 exec, exec, %4\n"), 8)
PROBE (p_sgpr_to_valu, REP8 ("s_add_u32 %1, %1, 1\n v_add_u32 %0, %1, %0\n v_readfirstlane_b32 %1, %0\n"), 8)
PROBE (p_memtime, REP8 ("s_memtime %4\n s_waitcnt lgkmcnt(0)\n"), 8)
PROBE (p_valu_indep, REP8 ("v_add_u32 %0, %0, 1\n v_add_u32 %2, %2, 1\n"), 8)
PROBE (p_dpp_shr, REP8 ("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"), 8)
PROBE (p_bpermute, REP8 ("ds_bpermute_b32 %0, %2, %0\n s_waitcnt lgkmcnt(0)\n"), 8)

PROBE (p_salu_to_valu, REP8 ("s_add_u32 %1, %1, 1\n v_add_u32 %2, %1, %2\n"), 8)
PROBE (p_valu_sgpr_to_salu, REP8 ("v_readfirstlane_b32 %1, %0\n s_add_u32 %1, %1, 1\n v_mov_b32 %0, %1\n"), 8)
PROBE (p_valu_sgpr_to_salu_far, REP8 ("v_readfirstlane_b32 %1, %0\n v_add_u32 %2, %2, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %2, %2, 1\n s_add_u32 %1, %1, 1\n v_mov_b32 %0, %1\n"), 8)
PROBE (p_lshl64, REP8 ("v_lshlrev_b64 %3, 1, %3\n"), 8)
PROBE (p_cmpx_window, REP8 ("v_cmpx_eq_u32 vcc, 2, %2\n v_readfirstlane_b32 %1, %0\n s_mov_b64 exec, -1\n v_add_u32 %0, %1, %0\n"), 8)
PROBE (p_cmpx_window4, REP8 ("v_cmpx_eq_u32 vcc, 2, %2\n v_readfirstlane_b32 %1, %0\n v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %0\n v_readfirstlane_b32 s22, %0\n s_mov_b64 exec, -1\n v_add_u32 %0, %1, %0\n"), 8)
PROBE (p_wave_shr, REP8 ("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n"), 8)
PROBE (p_readlane_vsel, REP8 ("v_readfirstlane_b32 %1, %2\n s_nop 3\n v_readlane_b32 %1, %0, %1\n v_mov_b32 %0, %1\n"), 8)
PROBE (p_vcmp_sgpr_cndmask, REP8 ("v_cmp_eq_u32 %4, %0, %2\n v_cndmask_b32 %0, %0, %2, %4\n"), 8)
PROBE (p_ffbh, REP8 ("v_ffbh_u32 %0, %0\n"), 8)
PROBE (p_branch_loop_like, REP8 ("s_add_u32 %1, %1, 1\n s_cmp_lg_u32 %1, 0\n s_cbranch_scc0 1f\n1:\n"), 8)
PROBE (p_early_cmp_late_branch, REP8 ("v_cmp_eq_u32 vcc, 77, %0\n v_add_u32 %2, %2, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %2, %2, 1\n s_cbranch_vccnz 1f\n v_add_u32 %0, %0, 1\n1:\n"), 8)
PROBE (p_six_valu, REP8 ("v_add_u32 %2, %2, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %0, %0, 1\n"), 8)
PROBE (p_ds_read_b64_indep, REP8 ("ds_read_b64 %3, %2\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n s_waitcnt lgkmcnt(0)\n"), 8)
PROBE (p_twelve_valu, REP8 ("v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n"), 8)
// accuracy of v_rcp_f64 over the totals a model can have, raw and after one / two Newton steps: the largest relative error seen, in units of 2^-53
__global__ void k_rcp_accuracy (double *out)
{
    double worst0 = 0, worst1 = 0, worst2 = 0;
    for (uint32_t t = 1 + threadIdx.x; t < 65536; t += 64) {
        const double d = (double)t;
        double r0 = __builtin_amdgcn_rcp (d);
        double r1 = __builtin_fma (__builtin_fma (-d, r0, 1.0), r0, r0);
        double r2 = __builtin_fma (__builtin_fma (-d, r1, 1.0), r1, r1);
        // d * r - 1 evaluated exactly in fma: the relative error of r
        const double e0 = fabs (__builtin_fma (d, r0, -1.0)), e1 = fabs (__builtin_fma (d, r1, -1.0)), e2 = fabs (__builtin_fma (d, r2, -1.0));
        worst0 = e0 > worst0 ? e0 : worst0; worst1 = e1 > worst1 ? e1 : worst1; worst2 = e2 > worst2 ? e2 : worst2;
    }
    for (int o = 32; o; o >>= 1) {
        const double a = __shfl_xor (worst0, o), b = __shfl_xor (worst1, o), c = __shfl_xor (worst2, o);
        worst0 = a > worst0 ? a : worst0; worst1 = b > worst1 ? b : worst1; worst2 = c > worst2 ? c : worst2;
    }
    if (!threadIdx.x) { out[0] = worst0; out[1] = worst1; out[2] = worst2; }
}

// does v_readfirstlane see the exec mask a v_cmpx has just written? (how many wait states the hand-over window of gz_intrin.h needs)
#define WINDOW(nops) asm volatile ("v_cmpx_eq_u32_e32 vcc, %1, %2\n\t" nops "v_readfirstlane_b32 %0, %3\n\ts_mov_b64 exec, -1" : "=&s" (got) : "v" (lane), "v" (want), "v" (val) : "vcc")
__global__ void k_window_hazard (uint32_t *out)
{
    const uint32_t lane = threadIdx.x, val = 1000 + lane;
    uint32_t bad0 = 0, bad1 = 0, bad2 = 0, bad4 = 0;
    for (uint32_t want = 0; want < 64; want++) {
        uint32_t got;
        WINDOW (""); bad0 += got != 1000 + want;
        WINDOW ("s_nop 0\n\t"); bad1 += got != 1000 + want;
        WINDOW ("s_nop 1\n\t"); bad2 += got != 1000 + want;
        WINDOW ("s_nop 3\n\t"); bad4 += got != 1000 + want;
    }
    if (!lane) { out[0] = bad0; out[1] = bad1; out[2] = bad2; out[3] = bad4; }
}
struct { const char *name; void (*k) (uint64_t *, uint32_t *); const char *what; } P[] = {
    { "empty loop", p_empty, "s_nop; loop overhead (s_add, s_cmp, s_cbranch taken)" },
    { "v_add_u32 chain", p_valu, "dependent VALU" },
    { "2 independent v_add chains", p_valu_indep, "per pair" },
    { "s_add_u32 chain", p_salu, "dependent SALU" },
    { "s_mul_i32 chain", p_smul, "" },
    { "v_mul_lo_u32 chain", p_vmullo, "" },
    { "v_mul_u32_u24 chain", p_vmul24, "" },
    { "v_readlane (const lane) -> v_mov", p_readlane_mov, "VGPR -> SGPR -> VGPR" },
    { "v_readfirstlane -> v_mov", p_readfirst_mov, "" },
    { "s_and -> s_nop 3 -> v_readlane (sgpr lane)", p_salu_readlane_idx, "" },
    { "v_cmp -> s_ff1 -> v_readlane -> v_mov", p_cmp_ff1_readlane, "the search's tail" },
    { "v_cmp -> v_cndmask", p_cmp_cndmask, "" },
    { "s_cmp -> s_cselect_b64 -> v_cndmask -> v_readfirstlane", p_scmp_cselect_cndmask, "" },
    { "ds_read_b32 dependent chain", p_ds_read_chain, "LDS latency" },
    { "ds_write_b32, ds_read_b32 same address", p_ds_write_read, "" },
    { "v_cvt_f64_u32 -> v_mul_f64 -> v_cvt_u32_f64", p_cvt_mul_cvt, "" },
    { "v_fma_f64 chain", p_fma64, "" },
    { "v_rcp_f64 chain", p_rcp64, "" },
    { "s_cmp + s_cbranch taken (skips 1) + s_add", p_branch_taken, "" },
    { "s_cmp + s_cbranch not taken + s_add", p_branch_not_taken, "" },
    { "v_cmp + s_cbranch_vccnz not taken + v_add", p_vbranch_vccz, "" },
    { "v_cmp + s_cmp_eq_u64 vcc + s_cbranch + v_add", p_ballot_branch, "" },
    { "v_cmp + s_and_saveexec + v_add + s_or exec", p_saveexec, "" },
    { "s_add -> v_add (sgpr operand) -> v_readfirstlane", p_sgpr_to_valu, "" },
    { "s_memtime + s_waitcnt", p_memtime, "a stamp" },
    { "v_add_u32_dpp row_shr chain", p_dpp_shr, "" },
    { "ds_bpermute chain", p_bpermute, "" },
    { "s_add -> v_add (sgpr operand), chain through SALU only", p_salu_to_valu, "SALU-written SGPR read by VALU, nothing waits for the VALU" },
    { "v_readfirstlane -> s_add -> v_mov", p_valu_sgpr_to_salu, "VALU-written SGPR read by SALU at once" },
    { "v_readfirstlane -> 4 v_add (other chain) -> s_add -> v_mov", p_valu_sgpr_to_salu_far, "the same, four instructions later" },
    { "v_lshlrev_b64 chain", p_lshl64, "" },
    { "v_cmpx -> v_readfirstlane -> s_mov exec -> v_add", p_cmpx_window, "broadcast of the hit lane through an exec window" },
    { "v_cmpx -> 4 v_readfirstlane -> s_mov exec -> v_add", p_cmpx_window4, "" },
    { "v_mov_b32_dpp wave_shr:1 chain", p_wave_shr, "" },
    { "v_readfirstlane -> s_nop 3 -> v_readlane (that lane) -> v_mov", p_readlane_vsel, "VALU-written lane select" },
    { "v_cmp_e64 (sgpr pair) -> v_cndmask", p_vcmp_sgpr_cndmask, "" },
    { "v_ffbh_u32 chain", p_ffbh, "" },
    { "s_add + s_cmp + s_cbranch_scc0 not taken", p_branch_loop_like, "" },
    { "v_cmp, 5 v_add, s_cbranch_vccnz not taken, v_add", p_early_cmp_late_branch, "a compare issued early, the branch late" },
    { "6 v_add (two chains)", p_six_valu, "reference for the row above" },
    { "ds_read_b64, 12 v_add, s_waitcnt", p_ds_read_b64_indep, "LDS latency under independent work" },
    { "12 v_add", p_twelve_valu, "reference for the row above" },
};
int main ()
{
    uint64_t *d, h[3];
    hipMalloc (&d, 24);
    double empty = 0;
    for (unsigned i = 0; i < sizeof (P) / sizeof (P[0]); i++) {
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL (P[i].k, dim3 (1), dim3 (64), 0, 0, d, (uint32_t *)0); hipDeviceSynchronize (); }
        hipMemcpy (h, d, 24, hipMemcpyDeviceToHost);
        const double per_iter = (double)h[0] / ITERS;
        if (!i) empty = per_iter;
        printf ("%-58s %7.1f clocks per loop iteration, %6.1f per link (loop overhead %.1f taken off)  %s\n", P[i].name, per_iter, i ? (per_iter - empty) / (double)h[2] : per_iter, empty, P[i].what);
    }
    uint32_t *dw, hw[4];
    hipMalloc (&dw, 16);
    hipLaunchKernelGGL (k_window_hazard, dim3 (1), dim3 (64), 0, 0, dw); hipDeviceSynchronize ();
    hipMemcpy (hw, dw, 16, hipMemcpyDeviceToHost);
    printf ("v_cmpx -> v_readfirstlane, wrong lane read in 64 trials: %u with no wait state, %u with 1, %u with 2, %u with 4\n", hw[0], hw[1], hw[2], hw[3]);
    double *dd, hd[3];
    hipMalloc (&dd, 24);
    hipLaunchKernelGGL (k_rcp_accuracy, dim3 (1), dim3 (64), 0, 0, dd); hipDeviceSynchronize ();
    hipMemcpy (hd, dd, 24, hipMemcpyDeviceToHost);
    printf ("v_rcp_f64 over 1 .. 65535: largest relative error %.3g (2^%.1f); after one Newton step %.3g (2^%.1f); after two %.3g (2^%.1f)\n",
            hd[0], log2 (hd[0] + 1e-300), hd[1], log2 (hd[1] + 1e-300), hd[2], log2 (hd[2] + 1e-300));
    return 0;
}
