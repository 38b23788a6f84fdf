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
turns [F.hi x] into the pair [0 F.hi] (tuples are 64-bit aligned: a 12-byte load cannot end on a pair's high word).
Every lane executes every step, and the state walks through a lane's PER symbols and then HOPS to the next lane: the low word of T -
r, the only word of T that is not a constant - is read from the lane before through DPP (wave_ror:1), and multiplied by that lane's
last F, which the lane holds as "the F before mine". A DPP read of a register a vector instruction has just written needs two wait
states (measured - without them the result is wrong), which is why a lane takes PER symbols in a row and not one. Only the diagonal
carries meaning; what the other lanes compute is never looked at.

Round 6 - what the loop paid for beyond its three instructions, measured (tools/probes/chain_regs_probe.py, profiles/r06_chain_regs_probe*.txt):
  * A RUN OF 8-BYTE INSTRUCTIONS WHOSE ADDRESSES ARE NOT MULTIPLES OF 8 ISSUES ONE PER 5 CLOCKS INSTEAD OF ONE PER 4 (the three instructions
    of a symbol are all 8-byte encodings: 15.07 instead of 12.07 clocks a symbol, whatever registers they name; tools/probes/issue_probe.py:
    every instruction costs a wave 4.07 clocks - 4 or 8 bytes, vector or scalar, dependent or not - and the fetch keeps up with 8 bytes a slot
    only as aligned words; a 4-byte instruction in between pays the deficit back). The hop's `s_nop 1` - the loop's only 4-byte
    instruction - flipped the parity at every lane, so every other lane's 36 instructions ran misaligned (13.8 clocks a symbol "bare").
    Now: `.p2align 3` in front of the loop, and 4-byte instructions only ever in pairs; the emitter below keeps count and
    tests/test_abi.py checks every 8-byte instruction's address in the assembled loop.
  * an instruction that has nothing to do with the chain costs 4 clocks wherever it stands (a load 13) - except in the two wait states
    in front of a hop's DPP read, which have to be there anyway: the next block's loads (two a hop), the shifts that make its operands,
    the scalar stores of the checkpoints all go THERE instead of s_nop; a hop costs 12 clocks in all (1.0 a symbol at 12 a lane).
  * the next block's operand shifts happen late in the current block (its loads were issued in the first hops and are waited for in
    hop 40: long there); nothing is left at a block's head.
  * A HOP INSIDE A ROW OF 16 LANES IS THE SECOND FMA ITSELF: gfx950 has one double-precision instruction with a DPP operand,
    `v_fmac_f64_dpp D, S0, S1 row_newbcast:n` (D = S0 * S1 + D, S0 read from lane n of the reader's row). Lane L + 1 holds "the F before
    mine" (Fp) and its negative (NFp); `v_mov_b64 R, NFp` - one of the two wait states the DPP read of T needs anyway - then
    `v_fmac_f64_dpp R, T, Fp row_newbcast:(L % 16)` is R = T(lane L) * Fp - Fp in lane L + 1: the move of r and the second fma in one
    instruction, 8 clocks a hop instead of 12. The four hops of a block that cross into the next row (and the wave's wrap-around) keep
    the old form (v_mov_b32_dpp wave_ror:1, then the fma). 13.35 -> 13.07 clocks a symbol alone, default step 39.7 -> 38.7 ms
    (profiles/r06_ubench_chain_fused.txt, r06_ab_chain_fused.txt; 8 / 10 / 16 symbols a lane: 13.45 / - / 12.85 alone, no better in the
    step - the 52 KB of the 16-symbol loop do not stay in the instruction cache beside the other kernels: 43.5 ms).
  * ONE COPY OF THE BLOCK'S CODE, 16 SYMBOLS A LANE. The loop used to be two copies of the block (the register sets taking turns): 39 KB at 12
    symbols a lane, and the instruction cache is shared with a compute unit that runs the model kernels (+ 4.6 % in the step against alone).
    Now the one block always reads set 0; the next block's operands are loaded and made in set 1 as before and moved over at the block's end
    (2 PER + 2 `v_mov_b64`, 4 bytes each, an even number: 0.13 clocks a symbol at 16 a lane). 26 KB at 16 symbols a lane (1024-symbol blocks,
    a checkpoint always at a lane's first symbol): 13.00 clocks a symbol alone, the chain's launch 35.3 -> 34.35 ms in the step (12 / 24 a lane in
    one copy: 13.23 / 12.82 alone, 34.7 / 35.1 in the step). profiles/r06_ubench_chain_single.txt, r06_ab_chain_single.txt.

Everything between the labels is written here, loop control included: the compiler schedules nothing in it. The rest of a leaf that
does not fill a block is the caller's.
"""
import os
import re
import sys

# (experiments only: the product's header is made with none of these set)
X_CKPT = int(os.environ.get("GZ_GEN_CKPT", "64"))            # a checkpoint every so many symbols (0: none - WRONG results, timing only)
X_FILL = os.environ.get("GZ_GEN_FILL", "1") == "1"           # 0: the hops' wait states are s_nop, loads and shifts at the head of a block (aligned all the same)
X_FUSE = os.environ.get("GZ_GEN_FUSE", "1") == "1"           # 1: a hop inside a row of 16 lanes is the second fma itself (v_fmac_f64_dpp row_newbcast), 0: every hop moves r first
X_SINGLE = os.environ.get("GZ_GEN_SINGLE", "1") == "1"       # 1: ONE copy of a block's code - the next block's operands are made in the second register set as before and
                                                             #    moved into the first at the block's end (half the code for PER + 2 v_mov_b64 per block)
X_ALIGN = os.environ.get("GZ_GEN_ALIGN", "1") == "1"         # 0: no .p2align, a single s_nop 1 in the hops (round 5's parity flips: for A / B)

PER = int(os.environ.get("GZ_GEN_PER", "16"))    # symbols a lane takes in a row
BLOCK = 64 * PER
REC = 12                                         # bytes per record

# registers: everything the loop touches is named here (the clobber list keeps the compiler off it)
_b = 40
OFF = f"v{_b}"                                                             # lane * PER * 12: my records inside a block
T2, T2LO, T2HI = f"v[{_b + 2}:{_b + 3}]", f"v{_b + 2}", f"v{_b + 3}"      # T2 = { r read from the lane before, 0x3ff00000 }
MASK, EXPO = f"v{_b + 4}", f"v{_b + 5}"
R, RLO, RHI = f"v[{_b + 6}:{_b + 7}]", f"v{_b + 6}", f"v{_b + 7}"
T, TLO = f"v[{_b + 8}:{_b + 9}]", f"v{_b + 8}"
FIXED_V = list(range(_b, _b + 10))
FIRST = _b + 10


def regset(base):
    """per symbol 4 registers: inv.lo inv.hi | F.lo (the load puts F.hi here; 0 after the shift) F.hi; then the F before mine (a pair)"""
    sym = [base + 4 * k for k in range(PER)]
    tail = base + 4 * PER
    return dict(rec=[f"v[{b}:{b + 2}]" for b in sym], inv=[f"v[{b}:{b + 1}]" for b in sym],
                F=[f"v[{b + 2}:{b + 3}]" for b in sym], Flo=[f"v{b + 2}" for b in sym], Fhi=[f"v{b + 3}" for b in sym],
                Fp=f"v[{tail}:{tail + 1}]", Fplo=f"v{tail}", Fphi=f"v{tail + 1}",
                NFp=f"v[{tail + 2}:{tail + 3}]", NFplo=f"v{tail + 2}", NFphi=f"v{tail + 3}", first=base, last=tail + 3)


SETS = [regset(FIRST), regset(FIRST + 4 * PER + 4)]
LAST_V = SETS[1]['last']
assert LAST_V <= 255, 'out of vector registers'
CLOB_V = FIXED_V + list(range(FIRST, LAST_V + 1))
CLOB_S = [36, 37, 40, 41, 42, 43, 44, 45, 46, 47]
NEXT, CK, TMP = "s[40:41]", "s[44:45]", "s[46:47]"      # NEXT = the records of the next block (of this one, when there is none: loaded, not used)
DPP = "wave_ror:1 row_mask:0xf bank_mask:0xf"
NOP2 = ["s_nop 0", "s_nop 0"]                           # two wait states in eight bytes

_INLINE_F = {"0.5", "-0.5", "1.0", "-1.0", "2.0", "-2.0", "4.0", "-4.0"}
_FOUR = ("s_nop", "s_waitcnt", "s_cbranch", "s_branch", "s_cmp", "s_mov_b32", "s_add_u32", "s_addc_u32", "s_sub_u32", "s_cselect_b32", "v_mov_b32", "v_mul_u32_u24", "v_xor_b32", "v_mov_b64")
_EIGHT = ("v_fma_f64", "v_and_or_b32", "v_lshlrev_b64", "global_load", "global_store", "s_store", "v_readlane_b32", "v_mbcnt")


def size(ins):
    """bytes of the encoding (gfx9: VOP3 / DPP / memory = 8; SOP / VOP1 / VOP2 = 4, + 4 with a literal constant) - tests/test_abi.py holds this against the assembler"""
    m = ins.split()[0]
    if m.endswith("_dpp") or m.endswith("_e64") or m.startswith(_EIGHT):
        return 8
    assert m.startswith(_FOUR), ins
    n = 4
    for op in re.split(r"[,\s]+", ins)[1:]:
        if re.fullmatch(r"-?(0x[0-9a-fA-F]+|\d+)", op):
            v = int(op, 0)
            if not -16 <= v <= 64:
                n = 8
        elif op in _INLINE_F:
            pass
    return n


class Emitter:
    """the instruction list + how many bytes it is behind an 8-byte boundary; an 8-byte instruction is never let out at an odd word"""

    def __init__(self):
        self.L = []
        self.odd = False
        self.pads = 0

    def __call__(self, ins):
        if ins.endswith(":") or ins.startswith("."):
            self.L.append(ins)
            return
        n = size(ins)
        if n == 8 and self.odd and X_ALIGN:
            self.L.append("s_nop 0")
            self.odd = False
            self.pads += 1
        self.L.append(ins)
        if n == 4:
            self.odd = not self.odd

    def label(self, name):
        """a branch target: on an 8-byte boundary whichever way it is reached"""
        if X_ALIGN:
            if self.odd:
                self("s_nop 0")
            self.L.append(".p2align 3")
        self.L.append(name + ":")


def loads(a, s, base):
    for k in range(PER):
        a(f"global_load_dwordx3 {s['rec'][k]}, {OFF}, {base} offset:{REC * k}")


def shifts(s):
    """the records of set s have arrived: [F.hi x] -> [0 F.hi]"""
    return [f"v_lshlrev_b64 {s['F'][k]}, 32, {s['F'][k]}" for k in range(PER)]


def fp_mov(s):
    return f"v_mov_b32_dpp {s['Fphi']}, {s['Fhi'][PER - 1]} {DPP}"          # the F before mine: the last of the lane before


def nfp_make(s):
    return f"v_xor_b32 {s['NFphi']}, 0x80000000, {s['Fphi']}"               # ... and its negative, the addend a fused hop starts from


def hop_fillers(cur, nxt):
    """what stands in the wait states of each of a block's 64 hops: the next block's loads, its operand shifts, the checkpoints' stores"""
    H = [[] for _ in range(64)]
    if X_FILL:
        ld = [f"global_load_dwordx3 {nxt['rec'][k]}, {OFF}, {NEXT} offset:{REC * k}" for k in range(PER)]
        for i, ins in enumerate(ld):
            H[i // 2].append(ins)
        h = 40
        assert PER // 2 < h
        H[h] += ["s_waitcnt vmcnt(0)", "s_nop 0"]
        for i, ins in enumerate(shifts(nxt)):
            H[h + 1 + i // 2].append(ins)
        hf = h + 1 + (PER + 1) // 2 + 1
        assert hf < 64
        H[hf].append(fp_mov(nxt))
        if X_FUSE:
            H[hf + 1].append(nfp_make(nxt))
    if X_CKPT:
        for c in range(BLOCK // 64):
            if (64 * c) % X_CKPT == 0:
                H[(64 * c) // PER].append(f"s_store_dwordx2 {TMP}, {CK}, 0x{8 * c:x}")        # what the two v_readlane in front of symbol 64 c have read
    for h in range(64):
        if X_FUSE and (h + 1) % 16:                          # a hop inside a row: R = -Fp (one of the two wait states), then the fused second fma
            H[h].insert(0, f"v_mov_b64 {R}, {cur['NFp']}")
        if not X_ALIGN:
            H[h] = ["s_nop 1"] if not H[h] else H[h] + ([] if len(H[h]) >= 2 else ["s_nop 0"])
            continue
        while len(H[h]) < 2:
            H[h].append("s_nop 0")
        if sum(size(i) == 4 for i in H[h]) % 2:              # 4-byte instructions in pairs: the move in its 8-byte encoding, or one more s_nop
            if H[h][0].startswith("v_mov_b64 "):
                H[h][0] = H[h][0].replace("v_mov_b64 ", "v_mov_b64_e64 ")
            else:
                H[h].append("s_nop 0")
    return H


def block(a, cur, nxt, tag):
    """BLOCK symbols with the operand set `cur` (loaded and made during the block before); the set `nxt` is loaded and made on the way"""
    if not X_FILL:
        a("s_waitcnt vmcnt(0)")                              # my records (asked for a block ago)
        a("s_nop 0")
        loads(a, nxt, NEXT)
        for ins in shifts(cur):
            a(ins)
        a("s_nop 0")
        a("s_nop 0")
        a(fp_mov(cur))
        if X_FUSE:
            a(nfp_make(cur))
    H = hop_fillers(cur, nxt)
    for j in range(BLOCK):
        lane, k = divmod(j, PER)
        if j % 64 == 0 and X_CKPT and j % X_CKPT == 0:       # the state before every 64th symbol goes out: it sits in lane j / PER
            a(f"v_readlane_b32 s46, {RLO}, {lane}")          # (R.hi is one instruction old: the read of R.lo is the wait state a v_readlane of a fresh result needs)
            a(f"v_readlane_b32 s47, {RHI}, {lane}")
        a(f"v_fma_f64 {T}, {R}, {cur['inv'][k]}, 1.0")
        if k < PER - 1:                                      # the lane's next symbol: in place
            a(f"v_fma_f64 {R}, {T}, {cur['F'][k]}, -{cur['F'][k]}")
        else:                                                # the next lane's first symbol (lane 0: the next block's)
            for ins in H[lane]:
                a(ins)
            if X_FUSE and (lane + 1) % 16:                   # lane + 1 reads T from lane `lane` of its row and multiplies by its own "F before mine": R = T * Fp + (-Fp)
                a(f"v_fmac_f64_dpp {R}, {T}, {cur['Fp']} row_newbcast:{lane % 16} row_mask:0xf bank_mask:0xf")
            else:
                a(f"v_mov_b32_dpp {T2LO}, {TLO} {DPP}")
                a(f"v_fma_f64 {R}, {T2}, {cur['Fp']}, -{cur['Fp']}")
        a(f"v_and_or_b32 {RHI}, {RHI}, {MASK}, {EXPO}")
    a(f"s_add_u32 s44, s44, {8 * (BLOCK // 64)}")
    a("s_addc_u32 s45, s45, 0")
    a("s_sub_u32 s42, s42, 1")                               # blocks left, the one that starts now included
    a("s_cmp_lt_u32 s42, 2")                                 # is there one behind it? Then its records are the next to load
    a(f"s_cselect_b32 s43, 0, {REC * BLOCK}")
    a("s_add_u32 s40, s40, s43")
    a("s_addc_u32 s41, s41, 0")


def body():
    a = Emitter()
    # operands: [rlo] [rhi] (v, in/out): the state - in: the same in every lane; out: valid in lane 0
    #           [blo] [bhi] (s): the records of the first block; [nblk] (s): blocks, >= 1; [clo] [chi] (s): where the first checkpoint goes
    if X_ALIGN:
        a(".p2align 3")
    a(f"v_mov_b32 {RLO}, %[rlo]")
    a(f"v_mov_b32 {RHI}, %[rhi]")
    a("s_mov_b32 s36, %[blo]")
    a("s_mov_b32 s37, %[bhi]")
    a("s_mov_b32 s42, %[nblk]")
    a("s_mov_b32 s44, %[clo]")
    a("s_mov_b32 s45, %[chi]")
    a("s_cmp_lt_u32 s42, 2")                                 # the records "of the next block": the second block's, or the first's again (a call of one block)
    a(f"s_cselect_b32 s43, 0, {REC * BLOCK}")
    a("s_add_u32 s40, s36, s43")
    a("s_addc_u32 s41, s37, 0")
    a(f"v_mbcnt_lo_u32_b32 {OFF}, -1, 0")
    a(f"v_mbcnt_hi_u32_b32 {OFF}, -1, {OFF}")
    a(f"v_mul_u32_u24 {OFF}, {REC * PER}, {OFF}")
    a(f"v_mov_b32 {T2HI}, 0x3ff00000")                       # the high word of 1 + r * 2^-52
    a(f"v_mov_b32 {MASK}, 0x7fffff")
    a(f"v_mov_b32 {EXPO}, 0x41000000")
    for s in SETS:
        a(f"v_mov_b32 {s['Fplo']}, 0")                       # the low word of the F before mine: 0, never written again
        a(f"v_mov_b32 {s['NFplo']}, 0")                      # (and of its negative)
    a("s_nop 4")
    loads(a, SETS[0], "s[36:37]")                            # the first block's records
    if X_FILL:
        a("s_waitcnt vmcnt(0)")
        for ins in shifts(SETS[0]):
            a(ins)
        a("s_nop 1")
        a(fp_mov(SETS[0]))
        if X_FUSE:
            a(nfp_make(SETS[0]))
    a.label("3")
    block(a, SETS[0], SETS[1], "0")
    a("s_cmp_eq_u32 s42, 0")
    a("s_cbranch_scc1 9f")
    if X_SINGLE:                                             # the next block's operands into the registers the (one) block reads
        for k in range(PER):
            a(f"v_mov_b64 {SETS[0]['inv'][k]}, {SETS[1]['inv'][k]}")
            a(f"v_mov_b64 {SETS[0]['F'][k]}, {SETS[1]['F'][k]}")
        a(f"v_mov_b64 {SETS[0]['Fp']}, {SETS[1]['Fp']}")
        a(f"v_mov_b64 {SETS[0]['NFp']}, {SETS[1]['NFp']}")
        a("s_branch 3b")
        a("s_nop 0")
    else:
        a.label("4")
        block(a, SETS[1], SETS[0], "1")
        a("s_cmp_lg_u32 s42, 0")
        a("s_cbranch_scc1 3b")
    a.label("9")
    a("s_waitcnt vmcnt(0) lgkmcnt(0)")
    a("s_nop 1")
    a(f"v_mov_b32 %[rlo], {RLO}")
    a(f"v_mov_b32 %[rhi], {RHI}")
    return a


def main():
    out = sys.argv[1]
    clob = ", ".join([f'"v{i}"' for i in CLOB_V] + [f'"s{i}"' for i in CLOB_S])
    a = body()
    with open(out, "w") as f:
        f.write("// gz_chain_asm.h -- generated by tools/gen_chain_asm.py (the comments are there) - do not edit\n#pragma once\n")
        f.write(f"#define GZ_CHAIN_BLOCK {BLOCK}        // symbols per block of the loop: 64 lanes x {PER} in a row\n")
        f.write(f"#define GZ_CHAIN_REC {REC}           // bytes per record: {{ inv.lo | cum, inv.hi, the high word of freq * 2^45 as a double }}\n")
        f.write(f"#define GZ_CHAIN_VREGS \"v{min(CLOB_V)}-v{max(CLOB_V)}\"   // the vector registers the loop names itself\n")
        f.write(f"#define GZ_CHAIN_PADS {a.pads}           // s_nop the emitter had to put in front of an 8-byte instruction at an odd word (none inside a lane's symbols)\n")
        f.write("#define GZ_CHAIN_F64_ASM \\\n")
        for ln in a.L:
            f.write(f'    "{ln}\\n\\t" \\\n')
        f.write("\n")
        f.write(f"#define GZ_CHAIN_F64_CLOBBERS {clob}, \"memory\", \"scc\", \"vcc\"\n")


if __name__ == "__main__":
    main()
