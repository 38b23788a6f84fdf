#!/usr/bin/env python3
"""Writes genozip_amd/csrc/gz_chain_asm.h: the inner loop of the range-coder chain (k_arith_chain) as ONE inline-asm statement.

What is serial in the range coder (c_range_coder.h:97-109) is  r = range / tot ; range = (r * freq) << 8k.  Here, per symbol, THREE
dependent vector instructions in double precision (the state R is range * 2^-7 as a double, rounding is toward zero):

    v_fma_f64      T, R, inv, 1.0             T = 1 + floor (R * 2^7 / tot) * 2^-52 = 1 + r * 2^-52;  inv = 2^-45 / tot rounded up
    v_fma_f64      R, T, F, -F                F = freq * 2^45:  (1 + r * 2^-52) * F - F = r * freq * 2^-7, exact
    v_and_or_b32   R.hi, R.hi, mask, exp      the exponent's low 3 bits stay, the others become "2^24 <= range < 2^32": that IS
                                              "shift left by whole bytes until >= 2^24" (r * freq >= 2^8 always)

The record of a symbol is TWELVE bytes (16 in rounds 1-4: { inv as a double, freq, cum }):  { inv.lo | cum,  inv.hi,  the high word of F }
- F has no low word (freq < 2^16), and the reciprocal needs no more than 36 of its 52 mantissa bits (range * error < 1 is all the
quotient asks for: tests/test_magic.py), so the table's entries are rounded up to a multiple of 2^16 units and the record carries the
symbol's cum in those 16 bits: the chain uses the double as it is. No operand goes through a scalar register and no load is waited
for in the loop: lane L of the wave holds the operands of the PER symbols base + PER * L .. + PER - 1; per record ONE 12-byte load a
block of 64 * PER symbols ahead, straight into the registers the loop reads ([inv.lo inv.hi | F.hi x]), and one v_lshlrev_b64 that
turns [F.hi x] into the pair [0 F.hi] when the block starts (tuples are 64-bit aligned: a 12-byte load cannot end on a pair's high
word). (An 8-byte record { tot | cum << 16, F.hi } with the reciprocal looked up by the lanes was built first - round 5,
profiles/r05_*: the same 6.6 ns per symbol alone on the device, 0.4 ns slower inside a step, where the bursts of 64-address
look-ups of 50 chains meet the other kernels' traffic.)
Every lane executes every step, and the state walks through a lane's PER symbols and then HOPS to the next lane: the low word of T -
r, the only word of T that is not a constant - is read from the lane before through DPP (wave_ror:1), and multiplied by that lane's
last F, which the lane holds as "the F before mine". A DPP read of a register a vector instruction has just written needs two wait
states (s_nop 1: measured - without them the result is wrong), which is why a lane takes PER symbols in a row and not one. Only the
diagonal carries meaning; what the other lanes compute is never looked at.
Clocks per symbol, alone on the device (tools/ubench_chain_f64.hip; profiles/r05_ubench_chain_rec12.txt): see there.

Everything between the labels is written here, loop control included: the compiler schedules nothing in it. The rest of a leaf that
does not fill a block is the caller's.
"""
import os
import sys

# (experiments only, tools/probes/chain_variants.sh: what each part of the loop costs - the product's header is made with none of these set)
X_CKPT = int(os.environ.get("GZ_GEN_CKPT", "64"))            # a checkpoint every so many symbols (0: none - WRONG results, timing only)
X_HOP = os.environ.get("GZ_GEN_HOP", "1") == "1"             # 0: no wait states and no DPP move at a lane's last symbol (WRONG results, timing only)
X_PREP = os.environ.get("GZ_GEN_PREP", "1") == "1"           # 0: the operands are not made from the records (WRONG results)
X_CKPT_FORM = os.environ.get("GZ_GEN_CKPT_FORM", "sstore")   # sstore: s_nop, v_readlane x 2, s_nop, s_store_dwordx2 per checkpoint (5 instructions); vsave: the lane that holds the
                                                             # state keeps it - two v_cndmask under a one-lane mask -, the block's checkpoints leave in ONE vector store (2 + 1 / 12)

PER = int(os.environ.get("GZ_GEN_PER", "12"))    # symbols a lane takes in a row
BLOCK = 64 * PER
REC = 12                                         # bytes per record

# registers: everything the loop touches is named here (the clobber list keeps the compiler off it)
_b = 40
OFF = f"v{_b}"                                                             # lane * PER * 8: my records inside a block
T2, T2LO, T2HI = f"v[{_b + 2}:{_b + 3}]", f"v{_b + 2}", f"v{_b + 3}"      # T2 = { r read from the lane before, 0x3ff00000 }
MASK, EXPO = f"v{_b + 4}", f"v{_b + 5}"
R, RLO, RHI = f"v[{_b + 6}:{_b + 7}]", f"v{_b + 6}", f"v{_b + 7}"
T, TLO = f"v[{_b + 8}:{_b + 9}]", f"v{_b + 8}"
CKOFF = f"v{_b + 1}"                                                       # (vsave) where my saved state goes: my checkpoint's slot, or the dump
FIXED_V = list(range(_b, _b + 10))
FIRST = _b + 10
CK_LANES = [(64 * c) // PER for c in range(BLOCK // 64)]                   # checkpoint c = the state before symbol 64 c of a block: it sits in this lane when that lane starts its symbol 64 c % PER
NSAVE = 1 if X_CKPT_FORM == "vsave" else 0                                 # (vsave) a lane holds at most one of a block's checkpoints: one pair
assert X_CKPT_FORM != "vsave" or len(set(CK_LANES)) == len(CK_LANES), "two checkpoints of a block in one lane"
CKM0 = 48                                                                  # (vsave) s[48 + 2 c : 49 + 2 c] = the mask of checkpoint c's lane; the pair behind them: all of them


def regset(base):
    """per symbol 4 registers: inv.lo inv.hi | F.lo (the load puts F.hi here; 0 after the shift) F.hi; then the F before mine (a pair)"""
    sym = [base + 4 * k for k in range(PER)]
    tail = base + 4 * PER
    return dict(rec=[f"v[{b}:{b + 2}]" for b in sym], inv=[f"v[{b}:{b + 1}]" for b in sym],
                F=[f"v[{b + 2}:{b + 3}]" for b in sym], Flo=[f"v{b + 2}" for b in sym], Fhi=[f"v{b + 3}" for b in sym],
                Fp=f"v[{tail}:{tail + 1}]", Fplo=f"v{tail}", Fphi=f"v{tail + 1}", first=base, last=tail + 1)


SETS = [regset(FIRST), regset(FIRST + 4 * PER + 2)]
SAVE0 = SETS[1]['last'] + 1
SAVE = [(f"v[{SAVE0 + 2 * k}:{SAVE0 + 2 * k + 1}]", f"v{SAVE0 + 2 * k}", f"v{SAVE0 + 2 * k + 1}") for k in range(NSAVE)]
LAST_V = SAVE0 + 2 * NSAVE - 1
assert LAST_V <= 255, 'out of vector registers'
CLOB_V = FIXED_V + list(range(FIRST, LAST_V + 1))
CLOB_S = [36, 37, 40, 41, 42, 44, 45, 46, 47] + (list(range(CKM0, CKM0 + 2 * len(CK_LANES) + 2)) if X_CKPT_FORM == "vsave" else [])
NEXT, CK, TMP = "s[40:41]", "s[44:45]", "s[46:47]"      # NEXT = the records of the next block
DPP = "wave_ror:1 row_mask:0xf bank_mask:0xf"


def loads(a, s, base):
    for k in range(PER):
        a(f"global_load_dwordx3 {s['rec'][k]}, {OFF}, {base} offset:{REC * k}")


def make_operands(a, s):
    """the records of set s have arrived: [F.hi x] -> [0 F.hi]"""
    if not X_PREP:
        return
    for k in range(PER):
        a(f"v_lshlrev_b64 {s['F'][k]}, 32, {s['F'][k]}")


def block(a, cur, nxt, tag):
    """BLOCK symbols with the operand set `cur` (its loads were issued a block ago)"""
    a("s_waitcnt vmcnt(0)")                                  # my records (asked for a block ago)
    a("s_cmp_lt_u32 s42, 2")                                 # the block after this one, if the call has one
    a(f"s_cbranch_scc1 2{tag}f")
    loads(a, nxt, NEXT)
    a(f"2{tag}:")
    make_operands(a, cur)
    a("s_nop 1")
    a(f"v_mov_b32_dpp {cur['Fphi']}, {cur['Fhi'][PER - 1]} {DPP}")          # the F before mine: the last of the lane before
    for j in range(BLOCK):
        lane, k = divmod(j, PER)
        if j % 64 == 0 and X_CKPT and j % X_CKPT == 0:       # the state before every 64th symbol goes out: it sits in lane j / PER
            if X_CKPT_FORM == "vsave":                       # every lane executes it, the mask keeps it to the lane that holds the state
                c = j // 64
                a(f"v_cndmask_b32 {SAVE[0][1]}, {SAVE[0][1]}, {RLO}, s[{CKM0 + 2 * c}:{CKM0 + 2 * c + 1}]")
                a(f"v_cndmask_b32 {SAVE[0][2]}, {SAVE[0][2]}, {RHI}, s[{CKM0 + 2 * c}:{CKM0 + 2 * c + 1}]")
            else:
                a("s_nop 0")
                a(f"v_readlane_b32 s46, {RLO}, {lane}")
                a(f"v_readlane_b32 s47, {RHI}, {lane}")
                a("s_nop 2")
                a(f"s_store_dwordx2 {TMP}, {CK}, 0x{8 * (j // 64):x}")
        a(f"v_fma_f64 {T}, {R}, {cur['inv'][k]}, 1.0")
        if k < PER - 1:                                      # the lane's next symbol: in place
            a(f"v_fma_f64 {R}, {T}, {cur['F'][k]}, -{cur['F'][k]}")
        elif not X_HOP:
            a(f"v_fma_f64 {R}, {T}, {cur['Fp']}, -{cur['Fp']}")
        else:                                                # the next lane's first symbol (lane 0: the next block's)
            a("s_nop 1")
            a(f"v_mov_b32_dpp {T2LO}, {TLO} {DPP}")
            a(f"v_fma_f64 {R}, {T2}, {cur['Fp']}, -{cur['Fp']}")
        a(f"v_and_or_b32 {RHI}, {RHI}, {MASK}, {EXPO}")
    if X_CKPT_FORM == "vsave" and X_CKPT:                    # the block's checkpoints: one store by the lanes that hold one
        n = len(CK_LANES)
        a(f"s_mov_b64 exec, s[{CKM0 + 2 * n}:{CKM0 + 2 * n + 1}]")
        a(f"global_store_dwordx2 {CKOFF}, {SAVE[0][0]}, {CK}")
        a("s_mov_b64 exec, -1")
    a(f"s_add_u32 s44, s44, {8 * (BLOCK // 64)}")
    a("s_addc_u32 s45, s45, 0")
    a(f"s_add_u32 s40, s40, {REC * BLOCK}")
    a("s_addc_u32 s41, s41, 0")
    a("s_sub_u32 s42, s42, 1")


def body():
    L = []
    a = L.append
    # operands: [rlo] [rhi] (v, in/out): the state - in: the same in every lane; out: valid in lane 0
    #           [blo] [bhi] (s): the records of the first block; [nblk] (s): blocks, >= 1; [clo] [chi] (s): where the first checkpoint goes
    a(f"v_mov_b32 {RLO}, %[rlo]")
    a(f"v_mov_b32 {RHI}, %[rhi]")
    a("s_mov_b32 s36, %[blo]")
    a("s_mov_b32 s37, %[bhi]")
    a(f"s_add_u32 s40, s36, {REC * BLOCK}")
    a("s_addc_u32 s41, s37, 0")
    a("s_mov_b32 s42, %[nblk]")
    a("s_mov_b32 s44, %[clo]")
    a("s_mov_b32 s45, %[chi]")
    a(f"v_mbcnt_lo_u32_b32 {OFF}, -1, 0")
    a(f"v_mbcnt_hi_u32_b32 {OFF}, -1, {OFF}")
    if X_CKPT_FORM == "vsave":                               # lane L's checkpoint, if it holds one, is number ceil (PER L / 64): its slot
        a(f"v_mad_u32_u24 {CKOFF}, {PER}, {OFF}, 63")
        a(f"v_lshrrev_b32 {CKOFF}, 6, {CKOFF}")
        a(f"v_lshlrev_b32 {CKOFF}, 3, {CKOFF}")
        allm = 0
        for c, ln in enumerate(CK_LANES):
            m = 1 << ln; allm |= m
            a(f"s_mov_b32 s{CKM0 + 2 * c}, 0x{m & 0xffffffff:x}")
            a(f"s_mov_b32 s{CKM0 + 2 * c + 1}, 0x{m >> 32:x}")
        a(f"s_mov_b32 s{CKM0 + 2 * len(CK_LANES)}, 0x{allm & 0xffffffff:x}")
        a(f"s_mov_b32 s{CKM0 + 2 * len(CK_LANES) + 1}, 0x{allm >> 32:x}")
    a(f"v_mul_u32_u24 {OFF}, {REC * PER}, {OFF}")
    a(f"v_mov_b32 {T2HI}, 0x3ff00000")                       # the high word of 1 + r * 2^-52
    a(f"v_mov_b32 {MASK}, 0x7fffff")
    a(f"v_mov_b32 {EXPO}, 0x41000000")
    for s in SETS:
        a(f"v_mov_b32 {s['Fplo']}, 0")                       # the low word of the F before mine: 0, never written again
    a("s_nop 4")
    loads(a, SETS[0], "s[36:37]")                            # the first block's records
    a("3:")
    block(a, SETS[0], SETS[1], "0")
    a("s_cmp_eq_u32 s42, 0")
    a("s_cbranch_scc1 9f")
    block(a, SETS[1], SETS[0], "1")
    a("s_cmp_lg_u32 s42, 0")
    a("s_cbranch_scc1 3b")
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
        f.write(f"#define GZ_CHAIN_REC {REC}           // bytes per record: {{ inv.lo | cum, inv.hi, the high word of freq * 2^45 as a double }}\n")
        f.write(f"#define GZ_CHAIN_VREGS \"v{min(CLOB_V)}-v{max(CLOB_V)}\"   // the vector registers the loop names itself\n")
        f.write("#define GZ_CHAIN_F64_ASM \\\n")
        for ln in body():
            f.write(f'    "{ln}\\n\\t" \\\n')
        f.write("\n")
        f.write(f"#define GZ_CHAIN_F64_CLOBBERS {clob}, \"memory\", \"scc\", \"vcc\"\n")


if __name__ == "__main__":
    main()
