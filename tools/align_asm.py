#!/usr/bin/env python3
"""Round 6 - MEASURED AND NOT USED BY THE PRODUCT BUILD: no configuration gets faster with it, the arithmetic decoder gets
5 % slower. The premise: on gfx950 a run of 8-byte instructions at addresses that are not multiples of 8 issues one per 5 clocks instead of one
per 4 (tools/probes/chain_regs_probe.py). What tools/probes/issue_probe.py found afterwards: only a RUN does - one 4-byte instruction in between
pays the fetch deficit back, and compiled code is full of them - so what this pass removes cost nothing, and what it adds (four bytes per
promotion) raises the byte rate of streams that were within the fetch rate. Kept as the record of that experiment.

In the compiler's code about half of the 8-byte encodings (VOP3, DPP / SDWA, every memory instruction, anything with a 32-bit literal) sit at an
odd word, 11 - 24 % of ALL instructions of this library's kernels. This pass moves them, without adding an instruction: a 4-byte VALU
instruction (VOP1 / VOP2 / VOPC, the compiler's `_e32` forms) has an 8-byte VOP3 encoding of the same operation (`_e64`) that issues in the
same 4 clocks when it is aligned itself - promoting one flips the parity of everything behind it. Addresses are a property of the layout,
not of the path taken, so a function is one linear sequence: a two-state dynamic programme (parity after every instruction) chooses the
promotions that leave the fewest 8-byte instructions at odd words.

    align_asm.py <in.s> <out.s>        in.s: `hipcc --cuda-device-only -S` output; out.s: the same instructions, some `_e32` as `_e64`

Sizes come from the assembler itself (the input is assembled and disassembled once); a promoted form the assembler refuses (two scalar
operands on the constant bus, an opcode without a VOP3 form) stays as it was. Functions that hold `.p2align` of their own in their body
(the range coder's chain loop, tools/gen_chain_asm.py: aligned by construction) are left alone. Prints what it did per kernel.
"""
import os
import re
import subprocess
import sys
import tempfile

BIN = "/opt/rocm/lib/llvm/bin/"
MC = [BIN + "llvm-mc", "-triple=amdgcn-amd-amdhsa", "-mcpu=gfx950"]
PROMOTE_COST = 0.01          # (an 8-byte encoding where 4 would do: code size only)


def assemble(src_path, obj_path):
    r = subprocess.run(MC + ["-filetype=obj", src_path, "-o", obj_path], capture_output=True, text=True)
    return r.returncode, r.stderr


def disassemble(obj_path):
    """-> { function: [(mnemonic, size)] } in address order"""
    out = subprocess.run([BIN + "llvm-objdump", "-d", obj_path], check=True, capture_output=True, text=True).stdout
    funcs, cur = {}, None
    for ln in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", ln)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        m = re.match(r"\s+(\S+).*//\s*([0-9A-F]+):((?: [0-9A-F]{8})+)", ln)
        if m and cur is not None:
            cur.append((m.group(1), 4 * len(m.group(3).split()), int(m.group(2), 16)))
    return funcs


def is_instruction(ln):
    s = ln.strip()
    return bool(s) and not s.startswith((";", ".", "//")) and not re.match(r"^[A-Za-z_.$][\w.$@]*:", s)


def functions_of(lines):
    """-> [(name, first line index, last line index)] of the code between `name:` and the end of its section / `.Lfunc_endN:`"""
    out, name, start = [], None, None
    for i, ln in enumerate(lines):
        m = re.match(r"^([A-Za-z_][\w.$]*):\s*(;.*)?$", ln)
        if m and not ln.startswith(".L") and name is None:
            name, start = m.group(1), i + 1
        elif name is not None and (ln.startswith(".Lfunc_end") or ln.strip().startswith(".section")):   # (a kernel's descriptor - another section - stands in front of .Lfunc_end)
            out.append((name, start, i))
            name = None
    return out


# an s_nop 0 put in front of an instruction costs one issue slot (4 clocks = 4 misplaced 8-byte instructions): it pays in front of runs of
# five and more that no promotion can reach - 57 places in the whole library, 340 of 14 958 misplaced instructions. OFF by default
# (GZ_ALIGN_NOPS=1): not worth a second kind of change to the compiler's code; never next to a pc-relative address computation.
NOP_COST = 4.0 if os.environ.get("GZ_ALIGN_NOPS") == "1" else 1e9


def plan(sizes, flexible, no_nop=()):
    """sizes[i] in (4, 8); flexible[i]: a 4-byte instruction that may become 8. -> (indices to promote, indices that get an s_nop 0 in front,
    8-byte instructions at odd words before / after)"""
    INF = 1e18
    n = len(sizes)
    cost = [[INF, INF] for _ in range(n + 1)]
    back = [[None, None] for _ in range(n + 1)]
    cost[0][0] = 0.0                                           # a function starts on a 256-byte boundary
    for i in range(n):
        for p in (0, 1):
            c = cost[i][p]
            if c >= INF:
                continue
            opts = [(sizes[i], 0.0, 0)]
            if flexible[i]:
                opts.append((8, PROMOTE_COST, 0))
            if p and sizes[i] == 8 and NOP_COST < 1e8 and i not in no_nop:
                opts.append((8, NOP_COST, 1))                  # (nop first: the instruction then sits at an even word)
            for sz, extra, nop in opts:
                pe = p ^ nop
                cc = c + extra + (1.0 if sz == 8 and pe else 0.0)
                q = pe ^ (1 if sz == 4 else 0)
                if cc < cost[i + 1][q]:
                    cost[i + 1][q] = cc
                    back[i + 1][q] = (p, sz, nop)
    p = 0 if cost[n][0] <= cost[n][1] else 1
    promote, nops = set(), set()
    for i in range(n, 0, -1):
        pp, sz, nop = back[i][p]
        if nop:
            nops.add(i - 1)
        elif sz != sizes[i - 1]:
            promote.add(i - 1)
        p = pp
    before, par = 0, 0
    for s in sizes:
        before += s == 8 and par
        par ^= s == 4
    after, par = 0, 0
    for i, s in enumerate(sizes):
        s = 8 if i in promote else s
        par ^= i in nops
        after += s == 8 and par
        par ^= s == 4
    return promote, nops, before, after


def main(src, dst):
    lines = open(src).read().split("\n")
    with tempfile.TemporaryDirectory() as td:
        obj = os.path.join(td, "a.o")
        rc, err = assemble(src, obj)
        if rc:
            sys.exit("the input does not assemble:\n" + err[:2000])
        dis = disassemble(obj)
        work = []                                               # (name, [line index of every instruction], sizes)
        for name, a, b in functions_of(lines):
            idx = [i for i in range(a, b) if is_instruction(lines[i])]
            body_align = any(lines[i].strip().startswith((".p2align", ".balign", ".align")) for i in range(a, b))
            d = dis.get(name)
            if d is None or body_align:
                if body_align:
                    print("%-60s left alone (alignment directives of its own)" % name[:60])
                continue
            # (behind a function's last instruction the assembler's output goes on with the padding to the next one: s_code_end / s_nop)
            if len(d) < len(idx) or any(lines[i].split()[0] != d[k][0] for k, i in enumerate(idx)):
                k = next((k for k, i in enumerate(idx) if k >= len(d) or lines[i].split()[0] != d[k][0]), 0)
                print("%-60s SKIPPED: source and disassembly part at instruction %d (%s)" % (name[:60], k, lines[idx[k]].strip()[:40]))
                continue
            work.append((name, idx, [s for _m, s, _a in d[:len(idx)]]))
        # which 4-byte `_e32` lines have a VOP3 form the assembler takes with the same operands: try them all at once, drop the refused
        cand = {}
        for name, idx, sizes in work:
            for k, i in enumerate(idx):
                if sizes[k] == 4 and re.match(r"\s*v_\w+_e32\b", lines[i]):
                    cand[i] = re.sub(r"(v_\w+)_e32\b", r"\1_e64", lines[i], count=1)
        refused = set()
        for _round in range(6):
            trial = list(lines)
            for i, new in cand.items():
                if i not in refused:
                    trial[i] = new
            tp = os.path.join(td, "t.s")
            open(tp, "w").write("\n".join(trial))
            rc, err = assemble(tp, os.path.join(td, "t.o"))
            bad = {int(m.group(1)) - 1 for m in re.finditer(r":(\d+):\d+: error:", err)}
            if not rc and not bad:
                break
            if not bad:
                sys.exit("the assembler failed without naming a line:\n" + err[:2000])
            refused |= bad
        else:
            sys.exit("promotions keep failing")
        tot_b = tot_a = tot_p = tot_n = tot_nop = 0
        n_nops = {}
        out = list(lines)
        for name, idx, sizes in work:
            flex = [idx[k] in cand and idx[k] not in refused for k in range(len(idx))]
            pcrel = {k + d for k, i in enumerate(idx) if "s_getpc" in lines[i] or "@rel32" in lines[i] for d in (-1, 0, 1, 2, 3)}
            promote, nops, before, after = plan(sizes, flex, pcrel)
            for k in promote:
                out[idx[k]] = cand[idx[k]]
            for k in nops:
                out[idx[k]] = "\ts_nop 0\n" + out[idx[k]]
            tot_b += before; tot_a += after; tot_p += len(promote); tot_n += len(idx); tot_nop += len(nops); n_nops[name] = len(nops)
            if len(idx) > 400:
                print("%-60s %6d instructions, 8-byte at an odd word: %5d -> %4d (%4.1f %% -> %4.1f %% of all), %5d promoted, %3d s_nop" %
                      (name[:60], len(idx), before, after, 100.0 * before / len(idx), 100.0 * after / len(idx), len(promote), len(nops)))
        open(dst, "w").write("\n".join(out))
        rc, err = assemble(dst, os.path.join(td, "o.o"))
        if rc:
            sys.exit("the output does not assemble:\n" + err[:2000])
        # the proof: same instruction count per function, and the misaligned count the plan promised
        dis2 = disassemble(os.path.join(td, "o.o"))
        mis = 0
        for name, idx, sizes in work:
            d2 = dis2[name][:len(idx) + n_nops[name]]
            src = [w for i in idx for w in (["s_nop"] if out[i].startswith("\ts_nop 0\n") else []) + [out[i].split("\n")[-1].split()[0]]]
            assert len(d2) == len(src) and all(w == d2[k][0] for k, w in enumerate(src)), name
            mis += sum(1 for _m, s, a in d2 if s == 8 and a % 8)
        assert mis == tot_a, (mis, tot_a)
        print("all: %d instructions in %d functions, 8-byte at an odd word %d -> %d, %d promotions (%d candidates refused by the assembler), %d s_nop put in" %
              (tot_n, len(work), tot_b, tot_a, tot_p, len(refused), tot_nop))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
