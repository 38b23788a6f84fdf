// issue-interval probes for one wave on gfx950: dependent chains of single instructions and of the range coder step
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_issue.hip -o tools/ubench_issue.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
// registers: v[10:11] state A, v[12:13] state B, v[20:21] inv, v[22:23] F, v[24:25] 2^52, v26 mask, v27 expo; s20.. scalar state
#define SETUP \
    "v_mov_b32 v10, 0xffe00000\n v_mov_b32 v11, 0x417fffff\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0x40100000\n" \
    "v_mov_b32 v20, 0x9abcdef0\n v_mov_b32 v21, 0x3f6a36e2\n v_mov_b32 v22, 0x00001234\n v_mov_b32 v23, 0x406d4c00\n" \
    "v_mov_b32 v24, 0\n v_mov_b32 v25, 0x43300000\n v_mov_b32 v26, 0x7fffff\n v_mov_b32 v27, 0x41000000\n" \
    "s_mov_b32 s20, 0xfedcba98\n s_mov_b32 s21, 0x10624dd3\n s_mov_b32 s22, 12\n s_mov_b32 s23, 1\n s_mov_b32 s24, 30000\n s_nop 4\n"
#define CLOB "v10","v11","v12","v13","v14","v15","v20","v21","v22","v23","v24","v25","v26","v27","s20","s21","s22","s23","s24","s25","s26","s27","scc","vcc","memory"

#define KERNEL(name, pre, body) \
__global__ void __launch_bounds__(64) name (uint64_t *cyc, int iters, uint32_t *sink) { \
    uint64_t t0, t1; \
    asm volatile ("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3\n" SETUP pre : : : CLOB); \
    asm volatile ("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory"); \
    for (int i = 0; i < iters; i++) asm volatile (REP64(body) : : : CLOB); \
    asm volatile ("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) : : "memory"); \
    uint32_t x; asm volatile ("s_mov_b64 exec, -1\n v_mov_b32 %0, v10" : "=v"(x) : : "memory"); \
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; sink[blockIdx.x] = x; } }

KERNEL (k_fma64,   "", "v_fma_f64 v[12:13], v[10:11], v[20:21], v[24:25]\n v_fma_f64 v[10:11], v[12:13], v[20:21], v[24:25]\n")          // 2 per body
KERNEL (k_add64,   "", "v_add_f64 v[12:13], v[10:11], v[24:25]\n v_add_f64 v[10:11], v[12:13], -v[24:25]\n")
KERNEL (k_mul64,   "", "v_mul_f64 v[12:13], v[10:11], v[22:23]\n v_mul_f64 v[10:11], v[12:13], v[20:21]\n")
KERNEL (k_andor,   "", "v_and_or_b32 v12, v10, v26, v27\n v_and_or_b32 v10, v12, v26, v27\n")
KERNEL (k_addu32,  "", "v_add_u32 v12, v10, v26\n v_add_u32 v10, v12, v27\n")
KERNEL (k_addu32_inplace, "", "v_add_u32 v10, v10, v26\n v_add_u32 v10, v10, v27\n")
KERNEL (k_salu,    "", "s_add_u32 s25, s20, s23\n s_add_u32 s20, s25, s22\n")
KERNEL (k_step,    "", "v_fma_f64 v[12:13], v[10:11], v[20:21], v[24:25]\n v_add_f64 v[12:13], v[12:13], -v[24:25]\n v_mul_f64 v[10:11], v[12:13], v[22:23]\n v_and_or_b32 v11, v11, v26, v27\n")   // 1 step per body
KERNEL (k_step_exec1, "s_mov_b64 exec, 1\n", "v_fma_f64 v[12:13], v[10:11], v[20:21], v[24:25]\n v_add_f64 v[12:13], v[12:13], -v[24:25]\n v_mul_f64 v[10:11], v[12:13], v[22:23]\n v_and_or_b32 v11, v11, v26, v27\n")
KERNEL (k_step_cvt, "s_mov_b64 exec, 1\n", "v_fma_f64 v[12:13], v[10:11], v[20:21], v[24:25]\n v_cvt_f64_u32 v[14:15], v12\n v_mul_f64 v[10:11], v[14:15], v[22:23]\n v_and_or_b32 v11, v11, v26, v27\n")
KERNEL (k_step_u24, "s_mov_b64 exec, 1\n v_mov_b32 v15, 0x42c00000\n", "v_fma_f64 v[12:13], v[10:11], v[20:21], v[24:25]\n v_mul_u32_u24 v14, v12, v22\n v_add_f64 v[10:11], v[14:15], -v[24:25]\n v_and_or_b32 v11, v11, v26, v27\n")
// the shipped integer step on the scalar unit: + inc, mulhi, >> shift, * freq, clz, & 0x18, <<
KERNEL (k_step_int, "", "s_add_u32 s25, s20, s23\n s_mul_hi_u32 s25, s25, s21\n s_lshr_b32 s25, s25, s22\n s_mul_i32 s25, s25, s24\n s_flbit_i32_b32 s26, s25\n s_and_b32 s26, s26, 0x18\n s_lshl_b32 s20, s25, s26\n")
// the f64 step with an operand fetch instruction in it (latency out of the picture: always the same address)
#define STEP4 "v_fma_f64 v[12:13], v[10:11], v[20:21], v[24:25]\n v_add_f64 v[12:13], v[12:13], -v[24:25]\n v_mul_f64 v[10:11], v[12:13], v[22:23]\n v_and_or_b32 v11, v11, v26, v27\n"
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)
#define KERNEL_LD(name, pre, LOADSTEP, tailwait) \
__global__ void __launch_bounds__(64) name (uint64_t *cyc, int iters, uint32_t *sink, const uint8_t *p) { \
    uint64_t t0, t1; const uint64_t pp = (uint64_t)(uintptr_t)p; __shared__ uint32_t lds[1024]; lds[threadIdx.x] = 0; __syncthreads (); \
    const uint32_t lo = __builtin_amdgcn_readfirstlane ((uint32_t)pp), hi = __builtin_amdgcn_readfirstlane ((uint32_t)(pp >> 32)); \
    const uint64_t ps = (uint64_t)lo | (uint64_t)hi << 32; \
    typedef uint32_t u4 __attribute__((vector_size (16))); u4 rs = { lo, hi & 0xffff, 0x100000u, 0x00020000u }; \
    asm volatile ("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3\n" SETUP "v_mov_b32 v28, 0\n s_mov_b32 s27, 64\n" pre : : : CLOB, "v28"); \
    asm volatile ("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory"); \
    for (int i = 0; i < iters; i++) asm volatile (REP32(LOADSTEP) tailwait REP32(LOADSTEP) tailwait : : "s"(ps), "s"(rs) : CLOB, "v28", "v40", "v41", "v42", "v43", "s60","s61","s62","s63","s64","s65","s66","s67","s68","s69","s70","s71","s72","s73","s74","s75"); \
    asm volatile ("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) : : "memory"); \
    uint32_t x; asm volatile ("s_mov_b64 exec, -1\n v_mov_b32 %0, v10" : "=v"(x) : : "memory"); \
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; sink[blockIdx.x] = x + lds[5]; } }
KERNEL_LD (k_ld_global64, "", "global_load_dwordx4 v[40:43], v28, %0 offset:64\n" STEP4, "s_waitcnt vmcnt(0)\n")
KERNEL_LD (k_ld_global1, "s_mov_b64 exec, 1\n", "global_load_dwordx4 v[40:43], v28, %0 offset:64\n" STEP4, "s_waitcnt vmcnt(0)\n")
KERNEL_LD (k_ld_buffer64, "", "buffer_load_dwordx4 v[40:43], off, %1, s27\n" STEP4, "s_waitcnt vmcnt(0)\n")
KERNEL_LD (k_ld_buffer1, "s_mov_b64 exec, 1\n", "buffer_load_dwordx4 v[40:43], off, %1, s27\n" STEP4, "s_waitcnt vmcnt(0)\n")
KERNEL_LD (k_ld_lds64, "", "ds_read_b128 v[40:43], v28 offset:64\n" STEP4, "s_waitcnt lgkmcnt(0)\n")
KERNEL_LD (k_ld_lds1, "s_mov_b64 exec, 1\n", "ds_read_b128 v[40:43], v28 offset:64\n" STEP4, "s_waitcnt lgkmcnt(0)\n")
// scalar operands: one s_load_dwordx16 per 4 steps, the steps read their operands from scalar registers
#define STEP4S(a, b) "v_fma_f64 v[12:13], v[10:11], s[" #a ":" #b "], v[24:25]\n v_add_f64 v[12:13], v[12:13], -v[24:25]\n v_mul_f64 v[10:11], v[12:13], v[22:23]\n v_and_or_b32 v11, v11, v26, v27\n"
KERNEL_LD (k_ld_smem, "s_mov_b64 exec, 1\n s_mov_b32 s20, 0x9abcdef0\n s_mov_b32 s21, 0x3f6a36e2\n", "s_load_dwordx16 s[60:75], %0, 0x40\n" STEP4S(20,21) STEP4S(20,21) STEP4S(20,21) STEP4S(20,21), "s_waitcnt lgkmcnt(0)\n")
// the state hops one lane per step: r (the low half of T) is read from the lane before
KERNEL_LD (k_hop_u24, "v_mov_b32 v15, 0x42c00000\n", "v_fma_f64 v[12:13], v[10:11], v[20:21], v[24:25]\n s_nop 1\n v_mul_u32_u24_dpp v14, v12, v22 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f64 v[10:11], v[14:15], -v[24:25]\n v_and_or_b32 v11, v11, v26, v27\n", "s_nop 0\n")
KERNEL_LD (k_hop_u24_rowshr, "v_mov_b32 v15, 0x42c00000\n", "v_fma_f64 v[12:13], v[10:11], v[20:21], v[24:25]\n s_nop 1\n v_mul_u32_u24_dpp v14, v12, v22 row_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f64 v[10:11], v[14:15], -v[24:25]\n v_and_or_b32 v11, v11, v26, v27\n", "s_nop 0\n")
#define HOP "v_fma_f64 v[12:13], v[10:11], v[20:21], v[24:25]\n s_nop 1\n v_mul_u32_u24_dpp v14, v12, v22 row_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f64 v[10:11], v[14:15], -v[24:25]\n v_and_or_b32 v11, v11, v26, v27\n"
KERNEL_LD (k_hop_16, "v_mov_b32 v15, 0x42c00000\n s_mov_b64 exec, 0xffff\n", HOP, "s_nop 0\n")
KERNEL_LD (k_hop_32, "v_mov_b32 v15, 0x42c00000\n s_mov_b64 exec, 0xffffffff\n", HOP, "s_nop 0\n")
KERNEL_LD (k_hop_2, "v_mov_b32 v15, 0x42c00000\n s_mov_b64 exec, 3\n", HOP, "s_nop 0\n")
KERNEL_LD (k_step_16, "s_mov_b64 exec, 0xffff\n", STEP4, "s_nop 0\n")
KERNEL_LD (k_step_32, "s_mov_b64 exec, 0xffffffff\n", STEP4, "s_nop 0\n")
KERNEL_LD (k_step_1, "s_mov_b64 exec, 1\n", STEP4, "s_nop 0\n")
KERNEL_LD (k_step_64, "", STEP4, "s_nop 0\n")
#define STEP3 "v_fma_f64 v[12:13], v[10:11], v[20:21], v[24:25]\n v_fma_f64 v[10:11], v[12:13], v[22:23], v[30:31]\n v_and_or_b32 v11, v11, v26, v27\n"
#define STEP3M "v_mov_b32 v13, v25\n v_fmac_f64 v[12:13], v[10:11], v[20:21]\n v_fmac_f64 v[10:11], v[12:13], v[22:23]\n v_and_or_b32 v11, v11, v26, v27\n"
#define STEP2M "v_fmac_f64 v[12:13], v[10:11], v[20:21]\n v_fmac_f64 v[10:11], v[12:13], v[22:23]\n v_and_or_b32 v11, v11, v26, v27\n"
KERNEL_LD (k_step3_64, "v_mov_b32 v30, 0\n v_mov_b32 v31, 0xc3300000\n", STEP3, "s_nop 0\n")
KERNEL_LD (k_step3_1, "v_mov_b32 v30, 0\n v_mov_b32 v31, 0xc3300000\n s_mov_b64 exec, 1\n", STEP3, "s_nop 0\n")
KERNEL_LD (k_step3_2, "v_mov_b32 v30, 0\n v_mov_b32 v31, 0xc3300000\n s_mov_b64 exec, 3\n", STEP3, "s_nop 0\n")
KERNEL_LD (k_step2m_64, "", STEP2M, "s_nop 0\n")
KERNEL_LD (k_step2m_1, "s_mov_b64 exec, 1\n", STEP2M, "s_nop 0\n")
KERNEL_LD (k_hop_u24_nonop, "v_mov_b32 v15, 0x42c00000\n", "v_fma_f64 v[12:13], v[10:11], v[20:21], v[24:25]\n v_mul_u32_u24_dpp v14, v12, v22 row_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f64 v[10:11], v[14:15], -v[24:25]\n v_and_or_b32 v11, v11, v26, v27\n", "s_nop 0\n")

int main ()
{
    uint64_t *cyc; uint32_t *sink; uint8_t *buf; CHK (hipMalloc (&cyc, 8192)); CHK (hipMalloc (&sink, 8192)); CHK (hipMalloc (&buf, 1 << 20)); CHK (hipMemset (buf, 0, 1 << 20));
    const int iters = 20000;
#define RUN(k, per, what) do { for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL (k, dim3 (1), dim3 (64), 0, 0, cyc, iters, sink); CHK (hipDeviceSynchronize ()); } \
        uint64_t c; CHK (hipMemcpy (&c, cyc, 8, hipMemcpyDeviceToHost)); printf ("%-46s %6.2f cycles per %s\n", #k, (double)c / ((double)iters * 64 * per), what); } while (0)
    RUN (k_fma64, 2, "instruction"); RUN (k_add64, 2, "instruction"); RUN (k_mul64, 2, "instruction"); RUN (k_andor, 2, "instruction");
    RUN (k_addu32, 2, "instruction"); RUN (k_addu32_inplace, 2, "instruction"); RUN (k_salu, 2, "instruction");
    RUN (k_step, 1, "step (fma, add, mul, and_or)"); RUN (k_step_exec1, 1, "step, one lane active"); RUN (k_step_cvt, 1, "step (fma, cvt_f64_u32, mul, and_or), one lane");
    RUN (k_step_u24, 1, "step (fma, mul_u32_u24, add, and_or), one lane"); RUN (k_step_int, 1, "step (7 scalar integer instructions)");
#define RUNLD(k, div, what) do { for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL (k, dim3 (1), dim3 (64), 0, 0, cyc, iters, sink, buf); CHK (hipDeviceSynchronize ()); } \
        uint64_t c; CHK (hipMemcpy (&c, cyc, 8, hipMemcpyDeviceToHost)); printf ("%-46s %6.2f cycles per %s\n", #k, (double)c / ((double)iters * 64 * div), what); } while (0)
    RUNLD (k_ld_global64, 1, "step + same-address global_load_dwordx4, 64 lanes"); RUNLD (k_ld_global1, 1, "step + global_load_dwordx4, one lane");
    RUNLD (k_ld_buffer64, 1, "step + buffer_load_dwordx4 (scalar address), 64 lanes"); RUNLD (k_ld_buffer1, 1, "step + buffer_load_dwordx4, one lane");
    RUNLD (k_ld_lds64, 1, "step + ds_read_b128, 64 lanes"); RUNLD (k_ld_lds1, 1, "step + ds_read_b128, one lane");
    RUNLD (k_ld_smem, 4, "step, scalar operands, s_load_dwordx16 per 4 steps");
    RUNLD (k_hop_u24, 1, "step (fma, nop, mul_u24 dpp wave_ror, add, and_or)"); RUNLD (k_hop_u24_rowshr, 1, "the same with row_ror"); RUNLD (k_hop_u24_nonop, 1, "row_ror without the nop (wrong results, timing only)");
    RUNLD (k_hop_16, 1, "hop step, 16 lanes active"); RUNLD (k_hop_32, 1, "hop step, 32 lanes active"); RUNLD (k_hop_2, 1, "hop step, 2 lanes active");
    RUNLD (k_step_1, 1, "plain step, 1 lane"); RUNLD (k_step_16, 1, "plain step, 16 lanes"); RUNLD (k_step_32, 1, "plain step, 32 lanes"); RUNLD (k_step_64, 1, "plain step, 64 lanes");
    RUNLD (k_step3_64, 1, "3-instruction step (fma, fma, and_or), 64 lanes"); RUNLD (k_step3_1, 1, "3-instruction step, 1 lane"); RUNLD (k_step3_2, 1, "3-instruction step, 2 lanes");
    RUNLD (k_step2m_64, 1, "3-instruction step with v_fmac_f64 (timing only), 64 lanes"); RUNLD (k_step2m_1, 1, "the same, 1 lane");
    RUNLD (k_step_1, 1, "plain step, 1 lane (again)"); RUNLD (k_hop_16, 1, "hop step, 16 lanes active (again)");
    return 0;
}
