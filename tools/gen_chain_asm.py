#!/usr/bin/env python3
"""Writes genozip_amd/csrc/gz_chain_asm.h: the inner loop of the range-coder chain (k_arith_chain) as ONE inline-asm statement.

What is serial in the range coder (c_range_coder.h:97-109) is  r = range / tot ; range = (r * freq) << 8k.  Here, per symbol, THREE
dependent vector instructions in double precision (the state R is range * 2^-7 as a double, rounding is toward zero):

    v_fma_f64      T, R, inv, 2^52            T = 2^52 + floor (R * inv) = 2^52 + r;  inv = 2^7 / tot rounded up
    v_fma_f64      R, T, F, G                 F = freq * 2^-7, G = -2^52 * F:  (2^52 + r) * F - 2^52 * F = r * freq * 2^-7, exact
    v_and_or_b32   R.hi, R.hi, mask, exp      the exponent's low 3 bits stay, the others become "2^24 <= range < 2^32": that IS
                                              "shift left by whole bytes until >= 2^24" (r * freq >= 2^8 always)

No operand goes through a scalar register and no load sits in the loop: lane L of the wave holds the records of the PER symbols
base + PER * L .. + PER - 1 (PER coalesced 16-byte loads per block of 64 * PER symbols, requested a block ahead; F and G are made
from the record's freq by all lanes at once: three instructions per record, 0.05 per symbol), every lane executes every step, and
the state walks through a lane's PER symbols and then HOPS to the next lane: the low word of T - r, the only word of T that is not a
constant - is read from the lane before through DPP (wave_ror:1), and multiplied by that lane's last F / G, which the lane holds as
"the F and G before mine". A DPP read of a register a vector instruction has just written needs two wait states (s_nop 1: measured -
without them the result is wrong), which is why a lane takes PER symbols in a row and not one. Only the diagonal carries meaning;
what the other lanes compute is never looked at.
Clocks per symbol, alone on the device (tools/ubench_chain_f64.hip; profiles/round4_ubench_chain_f64.txt): see there; the
seven-instruction integer form on the scalar unit (rounds 1-3): 30.5 + the waits for its scalar loads.

Everything between the labels is written here, loop control included: the compiler schedules nothing in it. The rest of a leaf that
does not fill a block is the caller's.
"""
import os
import sys

# (experiments only, tools/probes/chain_variants.sh: what each part of the loop costs - the product's header is made with none of these set)
X_CKPT = int(os.environ.get("GZ_GEN_CKPT", "64"))            # a checkpoint every so many symbols (0: none)
X_HOP = os.environ.get("GZ_GEN_HOP", "1") == "1"             # 0: no wait states and no DPP move at a lane's last symbol (WRONG results, timing only)
X_PREP = os.environ.get("GZ_GEN_PREP", "1") == "1"           # 0: F and G are not made from the records (WRONG results)
X_NOROT = os.environ.get("GZ_GEN_NOROT", "0")                # 1: every symbol reads the FIRST symbol's operand registers; inv / fg: only those do (WRONG results)
X_SPREAD = int(os.environ.get("GZ_GEN_SPREAD", "0"))          # n > 0: the next block's loads are issued one at a time, n symbols into every 64, instead of eight in a row at the block's start
X_X3 = os.environ.get("GZ_GEN_X3", "0") == "1"               # 1: 12 of a record's 16 bytes are loaded (the chain never looks at cum)
X_COAL = os.environ.get("GZ_GEN_COALESCED", "0") == "1"      # 1: load k of a block reads 1 KB in a row (lane l: record 64 k + l) - what a block-transposed record layout would give (WRONG results with today's layout)
X_TOUCH = os.environ.get("GZ_GEN_TOUCH", "0") == "1"          # (micro-benchmark only - the step faulted with it) 1: two one-dword loads per block touch every line of the block AFTER the next one (its records, and the page, are then near when the real loads ask)
X_ONE = os.environ.get("GZ_GEN_ONE", "0") == "1"             # 1: T = 1 + r * 2^-52 (the constant is the inline 1.0, inv is scaled by 2^-52, F by 2^52): one operand less from the register file
X_CKPT_FORM = os.environ.get("GZ_GEN_CKPT_FORM", "sstore")   # sstore: v_readlane x 2 + s_store_dwordx2 (the product's); none-wait: the same without the s_nops

PER = int(os.environ.get("GZ_GEN_PER", "12"))    # symbols a lane takes in a row (8 until round 5: 12 - 14 are ~0.6 ms faster on the default step, 16 slower: DESIGN.md section 3)
RPS = int(os.environ.get("GZ_GEN_RPS", "6" if PER > 8 else "8"))   # registers per symbol: beyond 8 symbols a lane F takes the place of freq and cum (6 x 16 x 2 sets + 8 = 200 registers)
BLOCK = 64 * PER

# registers
LOWREGS = PER * RPS * 2 + 8 + 64 > 256 or os.environ.get("GZ_GEN_LOW", "0") == "1"          # 16 symbols a lane: the record registers are v56 .. v255, everything else moves below them
_b = 40 if LOWREGS else 50
OFF = f"v{_b}"
OFF2 = f"v{_b + 1}"
OFFT, TCH0, TCH1 = f"v{_b - 3}", f"v{_b - 2}", f"v{_b - 1}"      # (GZ_GEN_TOUCH: lane * 128, two registers nobody reads)
T2, T2LO, T2HI = f"v[{_b + 2}:{_b + 3}]", f"v{_b + 2}", f"v{_b + 3}"      # T2 = { r read from the lane before, 0x43300000 }
_c = _b + 4 if LOWREGS else 56
C52, C52LO, C52HI = f"v[{_c}:{_c + 1}]", f"v{_c}", f"v{_c + 1}"          # 2^52
MASK, EXPO = f"v{_c + 2}", f"v{_c + 3}"
R, RLO, RHI = f"v[{_c + 4}:{_c + 5}]", f"v{_c + 4}", f"v{_c + 5}"
T, TLO = f"v[{_c + 6}:{_c + 7}]", f"v{_c + 6}"
FIXED_V = [_b, _b + 2, _b + 3] + ([_b + 1] if X_COAL else []) + ([_b - 3, _b - 2, _b - 1] if X_TOUCH else []) + list(range(_c, _c + 8))
FIRST = _c + 8


def regset(base):
    """per symbol 8 registers: inv.lo inv.hi freq cum | F.lo (0) F.hi | G.lo (0) G.hi; then the F and G before mine (2 pairs).
    RPS == 6: inv.lo inv.hi freq cum | G.lo (0) G.hi, and F is made IN PLACE of freq and cum (v_cvt_f64_u32 of freq: low word 0)"""
    sym = [base + RPS * k for k in range(PER)]
    tail = base + RPS * PER
    fo, go = (4, 6) if RPS == 8 else (2, 4)
    return dict(rec=[f"v[{b}:{b + 3}]" for b in sym], inv=[f"v[{b}:{b + 1}]" for b in sym], fq=[f"v{b + 2}" for b in sym],
                F=[f"v[{b + fo}:{b + fo + 1}]" for b in sym], Flo=[f"v{b + fo}" for b in sym], Fhi=[f"v{b + fo + 1}" for b in sym],
                G=[f"v[{b + go}:{b + go + 1}]" for b in sym], Glo=[f"v{b + go}" for b in sym], Ghi=[f"v{b + go + 1}" for b in sym],
                Fp=f"v[{tail}:{tail + 1}]", Fplo=f"v{tail}", Fphi=f"v{tail + 1}", Gp=f"v[{tail + 2}:{tail + 3}]", Gplo=f"v{tail + 2}", Gphi=f"v{tail + 3}",
                first=base, last=tail + 3, invhi=[f"v{b + 1}" for b in sym])


SETS = [regset(FIRST), regset(FIRST + RPS * PER + 4)]
assert SETS[1]['last'] <= 255, 'out of vector registers'
CLOB_V = FIXED_V + [r for s in SETS for r in range(s["first"], s["last"] + 1)]
CLOB_S = [36, 37, 40, 41, 42, 44, 45, 46, 47] + ([38, 39] if X_SPREAD or X_TOUCH else [])
BASE, NEXT, CK, TMP = "s[40:41]", "s[36:37]", "s[44:45]", "s[46:47]"     # NEXT = BASE + a block (beyond the 13-bit offset of a load)
DPP = "wave_ror:1 row_mask:0xf bank_mask:0xf"


def loads(a, s, base):
    for k in range(PER):
        if X_X3:
            b = s['first'] + RPS * k
            a(f"global_load_dwordx3 v[{b}:{b + 2}], {OFF}, {base} offset:{16 * k}")
        elif X_COAL:                                         # (a load's offset field takes up to 4095: the second half of the block through a second address register)
            a(f"global_load_dwordx4 {s['rec'][k]}, {OFF if k < 4 else OFF2}, {base} offset:{1024 * (k % 4)}")
        else:
            a(f"global_load_dwordx4 {s['rec'][k]}, {OFF}, {base} offset:{16 * k}")


def block(a, cur, nxt, tag):
    """BLOCK symbols with the operand set `cur` (its loads were issued a block ago)"""
    a("s_waitcnt vmcnt(2)" if X_TOUCH else "s_waitcnt vmcnt(0)")     # (the two touches were asked for after this block's records: they may still be on their way)
    a("s_cmp_lt_u32 s42, 2")                                 # the block after this one, if the call has one
    if X_SPREAD:                                             # (no next block: the loads read this block's records again - nobody looks at them)
        a("s_cselect_b32 s38, s40, s36")
        a("s_cselect_b32 s39, s41, s37")
    else:
        a(f"s_cbranch_scc1 2{tag}f")
        loads(a, nxt, NEXT)
        a(f"2{tag}:")
    if X_TOUCH:                                              # (fewer than three blocks left: this block's records again)
        a("s_cmp_lt_u32 s42, 3")
        a(f"s_add_u32 s38, s36, {16 * BLOCK}")
        a("s_addc_u32 s39, s37, 0")
        a("s_cselect_b32 s38, s40, s38")
        a("s_cselect_b32 s39, s41, s39")
        a(f"global_load_dword {TCH0}, {OFFT}, s[38:39] offset:0")
        a(f"global_load_dword {TCH1}, {OFFT}, s[38:39] offset:64")
    for k in range(PER if X_PREP else 0):                    # F = freq * 2^-7, G = -2^52 * F (the low words are and stay 0)
        a(f"v_cvt_f64_u32 {cur['F'][k]}, {cur['fq'][k]}")
    if X_ONE:
        for k in range(PER if X_PREP else 0):
            a(f"v_add_u32 {cur['Fhi'][k]}, 0x02d00000, {cur['Fhi'][k]}")   # exponent - 7 + 52
        for k in range(PER if X_PREP else 0):
            a(f"v_xor_b32 {cur['Ghi'][k]}, 0x80000000, {cur['Fhi'][k]}")   # G = -F
        for k in range(PER if X_PREP else 0):
            a(f"v_add_u32 {cur['invhi'][k]}, 0xfcc00000, {cur['invhi'][k]}")   # inv * 2^-52
    else:
        for k in range(PER if X_PREP else 0):
            a(f"v_add_u32 {cur['Fhi'][k]}, 0xff900000, {cur['Fhi'][k]}")       # exponent - 7
        for k in range(PER if X_PREP else 0):
            a(f"v_add_u32 {cur['Ghi'][k]}, 0x83400000, {cur['Fhi'][k]}")       # exponent + 52, sign
    a("s_nop 1")
    a(f"v_mov_b32_dpp {cur['Fphi']}, {cur['Fhi'][PER - 1]} {DPP}")          # the F and G before mine: the last of the lane before
    a(f"v_mov_b32_dpp {cur['Gphi']}, {cur['Ghi'][PER - 1]} {DPP}")
    for j in range(BLOCK):
        lane, k = divmod(j, PER)
        if j % 64 == 0 and X_CKPT and j % X_CKPT == 0:       # the state before every 64th symbol goes out: it sits in lane j / PER
            a("s_nop 0")
            a(f"v_readlane_b32 s46, {RLO}, {lane}")
            a(f"v_readlane_b32 s47, {RHI}, {lane}")
            a("s_nop 2")
            a(f"s_store_dwordx2 {TMP}, {CK}, 0x{8 * (j // 64):x}")
        if X_SPREAD and j % 64 == X_SPREAD:
            a(f"global_load_dwordx4 {nxt['rec'][j // 64]}, {OFF}, s[38:39] offset:{16 * (j // 64)}")
        ki = 0 if X_NOROT in ("1", "inv") else k
        kf = 0 if X_NOROT in ("1", "fg") else k
        a(f"v_fma_f64 {T}, {R}, {cur['inv'][ki]}, {'1.0' if X_ONE else C52}")
        if k < PER - 1:                                      # the lane's next symbol: in place
            a(f"v_fma_f64 {R}, {T}, {cur['F'][kf]}, {cur['G'][kf]}")
        elif not X_HOP:
            a(f"v_fma_f64 {R}, {T}, {cur['Fp']}, {cur['Gp']}")
        else:                                                # the next lane's first symbol (lane 0: the next block's)
            a("s_nop 1")
            a(f"v_mov_b32_dpp {T2LO}, {TLO} {DPP}")
            a(f"v_fma_f64 {R}, {T2}, {cur['Fp']}, {cur['Gp']}")
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
    a(f"v_mov_b32 {RLO}, %[rlo]")
    a(f"v_mov_b32 {RHI}, %[rhi]")
    a("s_mov_b32 s40, %[blo]")
    a("s_mov_b32 s41, %[bhi]")
    a(f"s_add_u32 s36, s40, {16 * BLOCK}")
    a("s_addc_u32 s37, s41, 0")
    a("s_mov_b32 s42, %[nblk]")
    a("s_mov_b32 s44, %[clo]")
    a("s_mov_b32 s45, %[chi]")
    a(f"v_mbcnt_lo_u32_b32 {OFF}, -1, 0")
    a(f"v_mbcnt_hi_u32_b32 {OFF}, -1, {OFF}")
    a(f"v_mul_u32_u24 {OFF}, {16 if X_COAL else 16 * PER}, {OFF}")
    if X_COAL:
        a(f"v_add_u32 {OFF2}, 0x1000, {OFF}")
    if X_TOUCH:
        a(f"v_mbcnt_lo_u32_b32 {OFFT}, -1, 0")
        a(f"v_mbcnt_hi_u32_b32 {OFFT}, -1, {OFFT}")
        a(f"v_mul_u32_u24 {OFFT}, {BLOCK * 16 // 64}, {OFFT}")                 # lane * 16 * PER: my records
    a(f"v_mov_b32 {T2HI}, {'0x3ff00000' if X_ONE else '0x43300000'}")       # the high word of 2^52 + r (of 1 + r * 2^-52)
    a(f"v_mov_b32 {C52LO}, 0")
    a(f"v_mov_b32 {C52HI}, 0x43300000")
    a(f"v_mov_b32 {MASK}, 0x7fffff")
    a(f"v_mov_b32 {EXPO}, 0x41000000")
    for s in SETS:                                           # the low words of every F and G: 0, never written again (v_cvt_f64_u32 rewrites F's with 0)
        for k in range(PER):
            a(f"v_mov_b32 {s['Glo'][k]}, 0")
        a(f"v_mov_b32 {s['Fplo']}, 0")
        a(f"v_mov_b32 {s['Gplo']}, 0")
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
    a("s_nop 1")
    a(f"v_mov_b32 %[rlo], {RLO}")
    a(f"v_mov_b32 %[rhi], {RHI}")
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
