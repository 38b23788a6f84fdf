#!/usr/bin/env python3
"""Writes genozip_amd/csrc/gz_chain_asm.h: the inner loop of the range-coder chain (k_arith_chain) as ONE inline-asm statement.

What is serial in the range coder (c_range_coder.h:97-109) is  r = range / tot ; range = (r * freq) << 8k.  Here, per symbol:

    v_fma_f64      T, R, inv, 2^52            round toward zero: T = 2^52 + floor (R * inv); inv = RU (2^7 / tot), R = range * 2^-7.
                                              The low half of T IS r as an integer.
    s_nop 1                                   (a DPP read of a register a vector instruction has just written: 2 wait states -
                                              measured: without them the result is wrong)
    v_mul_u32_u24  P, T.lo (lane - 1), freq   r * freq (r < 2^24 when tot >= 256), r read from the lane BEFORE (DPP wave_ror:1)
    v_add_f64      R, {P, 0x42c00000}, -2^45  the pair is the double 2^45 + P * 2^-7: R = P * 2^-7, exact
    v_and_or_b32   R.hi, R.hi, mask, exp      the exponent's low 3 bits stay, the others become "2^24 <= range < 2^32": that IS
                                              "shift left by whole bytes until >= 2^24" (P >= 2^8 always)

No operand goes through a scalar register and no load sits in the loop: lane j of the wave holds the record of symbol base + j (ONE
coalesced vector load per 64 symbols, requested a block ahead), every lane executes every step, and the state HOPS one lane per
symbol - the DPP read above (lane j + 1 multiplies lane j's r by freq_j, which it holds as "the frequency before mine"). Only the
diagonal carries meaning; what the other lanes compute is never looked at. 26 clocks per symbol (tools/ubench_chain_f64.hip; the
seven-instruction integer form on the scalar unit: 30.5 + the waits for its scalar loads), whatever else the device is doing.

A block in which some total is below 256 (r may then need more than 24 bits) is left to the caller, as is the rest of a leaf that
does not fill a block. Everything between the labels is written here, loop control included: the compiler schedules nothing in it.
"""
import sys

# registers
R, T = "v[60:61]", "v[62:63]"
RLO, RHI, TLO = "v60", "v61", "v62"
P, PLO = "v[52:53]", "v52"      # v53 = 0x42c00000
C45 = "v[54:55]"                # 2^45
C52 = "v[56:57]"                # 2^52
MASK, EXPO, OFF16 = "v58", "v59", "v50"
SETS = [dict(inv="v[64:65]", rec="v[64:67]", fq="v66", fp="v68"), dict(inv="v[70:71]", rec="v[70:73]", fq="v72", fp="v74")]
BASE, CK, TMP = "s[40:41]", "s[44:45]", "s[46:47]"
CLOB_V = [50, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65, 66, 67, 68, 70, 71, 72, 73, 74]
CLOB_S = [40, 41, 42, 44, 45, 46, 47]
DPP = "wave_ror:1 row_mask:0xf bank_mask:0xf"


def block(a, cur, nxt, tag):
    """64 symbols with the operand set `cur` (its load was issued a block ago)"""
    a("s_waitcnt vmcnt(0)")
    a("s_cmp_lt_u32 s42, 2")                                 # the block after this one, if the call has one
    a(f"s_cbranch_scc1 2{tag}f")
    a(f"global_load_dwordx4 {nxt['rec']}, {OFF16}, {BASE} offset:1024")
    a(f"2{tag}:")
    a(f"v_mov_b32_dpp {cur['fp']}, {cur['fq']} {DPP}")       # the frequency before mine
    a(f"v_readfirstlane_b32 s46, {RLO}")                     # the range before the block goes out
    a(f"v_readfirstlane_b32 s47, {RHI}")
    a(f"v_cmp_lt_f64 vcc, 0.5, {cur['inv']}")               # a total below 256 somewhere in the block
    a("s_nop 1")
    a(f"s_store_dwordx2 {TMP}, {CK}, 0x0")
    a("s_add_u32 s44, s44, 8")
    a("s_addc_u32 s45, s45, 0")
    a("s_cbranch_vccnz 9f")
    for j in range(64):
        a(f"v_fma_f64 {T}, {R}, {cur['inv']}, {C52}")
        if j < 63:
            a("s_nop 1")
            a(f"v_mul_u32_u24_dpp {PLO}, {TLO}, {cur['fp']} {DPP}")
        else:                                                # into lane 0, which holds the next block's first record
            a(f"v_mul_u32_u24 {PLO}, {TLO}, {cur['fq']}")
            a("s_nop 1")
            a(f"v_mov_b32_dpp {PLO}, {PLO} {DPP}")
        a(f"v_add_f64 {R}, {P}, -{C45}")
        a(f"v_and_or_b32 {RHI}, {RHI}, {MASK}, {EXPO}")
    a("s_add_u32 s40, s40, 1024")
    a("s_addc_u32 s41, s41, 0")
    a("s_sub_u32 s42, s42, 1")


def body():
    L = []
    a = L.append
    # operands: [rlo] [rhi] (v, in/out): the state - in: the same in every lane; out: valid in lane 0
    #           [blo] [bhi] (s): the records of the first block; [nblk] (s): blocks, >= 1; [clo] [chi] (s): where the first checkpoint goes
    #           [left] (s, out): 0, or the number of blocks not done: the first of them holds a total below 256 (its checkpoint is written)
    a("v_mov_b32 v60, %[rlo]")
    a("v_mov_b32 v61, %[rhi]")
    a("s_mov_b32 s40, %[blo]")
    a("s_mov_b32 s41, %[bhi]")
    a("s_mov_b32 s42, %[nblk]")
    a("s_mov_b32 s44, %[clo]")
    a("s_mov_b32 s45, %[chi]")
    a("v_mbcnt_lo_u32_b32 v50, -1, 0")
    a("v_mbcnt_hi_u32_b32 v50, -1, v50")
    a("v_lshlrev_b32 v50, 4, v50")                           # lane * 16: my record
    a("v_mov_b32 v53, 0x42c00000")
    a("v_mov_b32 v54, 0")
    a("v_mov_b32 v55, 0x42c00000")
    a("v_mov_b32 v56, 0")
    a("v_mov_b32 v57, 0x43300000")
    a("v_mov_b32 v58, 0x7fffff")
    a("v_mov_b32 v59, 0x41000000")
    a("s_nop 4")
    a(f"global_load_dwordx4 {SETS[0]['rec']}, {OFF16}, {BASE}")
    a("1:")
    block(a, SETS[0], SETS[1], "0")
    a("s_cmp_eq_u32 s42, 0")
    a("s_cbranch_scc1 9f")
    block(a, SETS[1], SETS[0], "1")
    a("s_cmp_lg_u32 s42, 0")
    a("s_cbranch_scc1 1b")
    a("9:")
    a("s_waitcnt vmcnt(0) lgkmcnt(0)")
    a("s_mov_b32 %[left], s42")
    a("s_nop 1")
    a("v_mov_b32 %[rlo], v60")
    a("v_mov_b32 %[rhi], v61")
    return L


def main():
    out = sys.argv[1]
    clob = ", ".join([f'"v{i}"' for i in CLOB_V] + [f'"s{i}"' for i in CLOB_S])
    with open(out, "w") as f:
        f.write("// gz_chain_asm.h -- generated by tools/gen_chain_asm.py (the comments are there) - do not edit\n#pragma once\n")
        f.write("#define GZ_CHAIN_F64_ASM \\\n")
        for ln in body():
            f.write(f'    "{ln}\\n\\t" \\\n')
        f.write("\n")
        f.write(f"#define GZ_CHAIN_F64_CLOBBERS {clob}, \"memory\", \"scc\", \"vcc\"\n")


if __name__ == "__main__":
    main()
