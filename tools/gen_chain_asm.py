#!/usr/bin/env python3
"""Writes genozip_amd/csrc/gz_chain_asm.h: the inner loop of the range-coder chain (k_arith_chain) as ONE inline-asm statement.

What is serial in the range coder (c_range_coder.h:97-109) is  r = range / tot ; range = (r * freq) << 8k.  Here, per symbol:

    v_fma_f64      T, R, inv, 2^52            round toward zero: T = 2^52 + floor (R * inv); inv = RU (2^7 / tot), R = range * 2^-7.
                                              The low half of T IS r as an integer.
    v_mul_u32_u24  P, T.lo, freq              r * freq (r < 2^24 when tot >= 256)
    v_add_f64      R, {P, 0x42c00000}, -2^45  the pair is the double 2^45 + P * 2^-7: R = P * 2^-7, exact
    v_and_or_b32   R.hi, R.hi, mask, exp      the exponent's low 3 bits stay, the others become "2^24 <= range < 2^32": that IS
                                              "shift left by whole bytes until >= 2^24" (P >= 2^8 always)

No operand goes through a scalar register and no load sits in the loop: lane L of the wave holds the records of the PER symbols
base + PER * L .. + PER - 1 (PER coalesced 16-byte loads per 64 * PER symbols, requested a block ahead), every lane executes every
step, and the state walks through a lane's PER symbols and then HOPS to the next lane: the multiply of a lane's first symbol reads
r from the lane before through DPP (wave_ror:1) and multiplies it by that lane's last frequency, which it holds as "the frequency
before mine". A DPP read of a register a vector instruction has just written needs two wait states (s_nop 1: measured - without
them the result is wrong, and they cost 6 clocks), which is why a lane takes PER symbols in a row and not one. Only the diagonal
carries meaning; what the other lanes compute is never looked at.
Clocks per symbol, alone on the device (tools/ubench_chain_f64.hip): PER = 1: 25.8, PER = 4: see profiles/; the seven-instruction
integer form on the scalar unit (rounds 1-3): 30.5 + the waits for its scalar loads.

A block in which some total is below 256 (r may then need more than 24 bits) is left to the caller, as is the rest of a leaf that
does not fill a block. Everything between the labels is written here, loop control included: the compiler schedules nothing in it.
"""
import sys

PER = 8                         # symbols a lane takes in a row
BLOCK = 64 * PER

# registers
R, T = "v[60:61]", "v[62:63]"
RLO, RHI, TLO = "v60", "v61", "v62"
P, PLO = "v[52:53]", "v52"      # v53 = 0x42c00000
C45 = "v[54:55]"                # 2^45
C52 = "v[56:57]"                # 2^52
MASK, EXPO, OFF = "v58", "v59", "v50"
VCMP = "v51"


def regset(base):
    """PER records of 4 registers each + the frequency before mine"""
    return dict(rec=[f"v[{base + 4 * k}:{base + 4 * k + 3}]" for k in range(PER)], inv=[f"v[{base + 4 * k}:{base + 4 * k + 1}]" for k in range(PER)],
                fq=[f"v{base + 4 * k + 2}" for k in range(PER)], fp=f"v{base + 4 * PER}", first=base, last=base + 4 * PER)


SETS = [regset(64), regset(64 + 4 * PER + 2)]
CLOB_V = [50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63] + [r for s in SETS for r in range(s["first"], s["last"] + 1)]
CLOB_S = [36, 37, 40, 41, 42, 44, 45, 46, 47, 48, 49]
BASE, NEXT, CK, TMP = "s[40:41]", "s[36:37]", "s[44:45]", "s[46:47]"     # NEXT = BASE + a block (beyond the 13-bit offset of a load)
DPP = "wave_ror:1 row_mask:0xf bank_mask:0xf"


def loads(a, s, base):
    for k in range(PER):
        a(f"global_load_dwordx4 {s['rec'][k]}, {OFF}, {base} offset:{16 * k}")


def block(a, cur, nxt, tag):
    """BLOCK symbols with the operand set `cur` (its loads were issued a block ago)"""
    a("s_waitcnt vmcnt(0)")
    a("s_cmp_lt_u32 s42, 2")                                 # the block after this one, if the call has one
    a(f"s_cbranch_scc1 2{tag}f")
    loads(a, nxt, NEXT)
    a(f"2{tag}:")
    a(f"v_mov_b32_dpp {cur['fp']}, {cur['fq'][PER - 1]} {DPP}")   # the frequency before mine: the last of the lane before
    # a total below 256 somewhere in the block? (inv > 0.5)
    a(f"v_cmp_lt_f64 vcc, 0.5, {cur['inv'][0]}")
    for k in range(1, PER):
        a(f"v_cmp_lt_f64 s[48:49], 0.5, {cur['inv'][k]}")
        a("s_or_b64 vcc, vcc, s[48:49]")
    a("s_nop 1")
    a("s_cbranch_vccnz 9f")
    for j in range(BLOCK):
        lane, k = divmod(j, PER)
        if j % 64 == 0:                                      # the state before every 64th symbol goes out: it sits in lane j / PER
            a("s_nop 0")
            a(f"v_readlane_b32 s46, {RLO}, {lane}")
            a(f"v_readlane_b32 s47, {RHI}, {lane}")
            a("s_nop 2")
            a(f"s_store_dwordx2 {TMP}, {CK}, 0x{8 * (j // 64):x}")
        a(f"v_fma_f64 {T}, {R}, {cur['inv'][k]}, {C52}")
        if k < PER - 1:                                      # the lane's next symbol: in place
            a(f"v_mul_u32_u24 {PLO}, {TLO}, {cur['fq'][k]}")
        elif j < BLOCK - 1:                                  # the next lane's first symbol
            a("s_nop 1")
            a(f"v_mul_u32_u24_dpp {PLO}, {TLO}, {cur['fp']} {DPP}")
        else:                                                # lane 0, which holds the next block's first records
            a(f"v_mul_u32_u24 {PLO}, {TLO}, {cur['fq'][k]}")
            a("s_nop 1")
            a(f"v_mov_b32_dpp {PLO}, {PLO} {DPP}")
        a(f"v_add_f64 {R}, {P}, -{C45}")
        a(f"v_and_or_b32 {RHI}, {RHI}, {MASK}, {EXPO}")
    a(f"s_add_u32 s44, s44, {8 * (BLOCK // 64)}")
    a("s_addc_u32 s45, s45, 0")
    a(f"s_add_u32 s40, s40, {16 * BLOCK}")
    a("s_addc_u32 s41, s41, 0")
    a(f"s_add_u32 s36, s36, {16 * BLOCK}")
    a("s_addc_u32 s37, s37, 0")
    a("s_sub_u32 s42, s42, 1")


def body():
    L = []
    a = L.append
    # operands: [rlo] [rhi] (v, in/out): the state - in: the same in every lane; out: valid in lane 0
    #           [blo] [bhi] (s): the records of the first block; [nblk] (s): blocks, >= 1; [clo] [chi] (s): where the first checkpoint goes
    #           [left] (s, out): 0, or the number of blocks not done: the first of them holds a total below 256 (NO checkpoint of it is written)
    a("v_mov_b32 v60, %[rlo]")
    a("v_mov_b32 v61, %[rhi]")
    a("s_mov_b32 s40, %[blo]")
    a("s_mov_b32 s41, %[bhi]")
    a(f"s_add_u32 s36, s40, {16 * BLOCK}")
    a("s_addc_u32 s37, s41, 0")
    a("s_mov_b32 s42, %[nblk]")
    a("s_mov_b32 s44, %[clo]")
    a("s_mov_b32 s45, %[chi]")
    a("v_mbcnt_lo_u32_b32 v50, -1, 0")
    a("v_mbcnt_hi_u32_b32 v50, -1, v50")
    a(f"v_mul_u32_u24 v50, {16 * PER}, v50")                 # lane * 16 * PER: my records
    a("v_mov_b32 v53, 0x42c00000")
    a("v_mov_b32 v54, 0")
    a("v_mov_b32 v55, 0x42c00000")
    a("v_mov_b32 v56, 0")
    a("v_mov_b32 v57, 0x43300000")
    a("v_mov_b32 v58, 0x7fffff")
    a("v_mov_b32 v59, 0x41000000")
    a("s_nop 4")
    loads(a, SETS[0], BASE)
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
        f.write(f"#define GZ_CHAIN_BLOCK {BLOCK}        // symbols per block of the loop: 64 lanes x {PER} in a row\n")
        f.write("#define GZ_CHAIN_F64_ASM \\\n")
        for ln in body():
            f.write(f'    "{ln}\\n\\t" \\\n')
        f.write("\n")
        f.write(f"#define GZ_CHAIN_F64_CLOBBERS {clob}, \"memory\", \"scc\", \"vcc\"\n")


if __name__ == "__main__":
    main()
