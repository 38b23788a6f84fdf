#!/usr/bin/env python3
"""Round 6: what does the range-coder chain's loop pay for beyond its three dependent instructions (12.07 clocks a symbol with fixed
operand registers, 13.8 with the product's rotating ones, 15.1 with hops, loads and checkpoints)? Writes one .hip file with a kernel per
variant of the loop body - timing only, the values are whatever they are - and a main() that prints clocks per symbol for each.

    python tools/probes/chain_regs_probe.py /tmp/chain_regs_probe.hip && hipcc --offload-arch=gfx950 -O2 /tmp/chain_regs_probe.hip -o tools/probes/chain_regs_probe.bin
    (GPU box)  tools/probes/chain_regs_probe.bin
"""
import sys

R, T = 46, 48            # the product's state registers (pairs)
MASK, EXPO = 44, 45
T2 = 42                  # the hop's pair { r from the lane before, 0x3ff00000 }
KERNELS = []             # (name, description, [asm lines of one loop iteration], symbols per iteration, uses_memory)


def p(n):
    return f"v[{n}:{n + 1}]"


def sym(inv, F, r=R, t=T, neg=True, mask=f"v{MASK}", expo=f"v{EXPO}", g=None, c="1.0", norm="and_or"):
    L = [f"v_fma_f64 {p(t)}, {p(r)}, {p(inv)}, {c}",
         f"v_fma_f64 {p(r)}, {p(t)}, {p(F)}, " + (f"-{p(F)}" if g is None else p(g))]
    if norm == "and_or":
        L.append(f"v_and_or_b32 v{r + 1}, v{r + 1}, {mask}, {expo}")
    elif norm == "bfi":
        L.append(f"v_bfi_b32 v{r + 1}, {mask}, v{r + 1}, {expo}")
    return L


def add(name, desc, lines, nsym, mem=False, pad=None):
    KERNELS.append((name, desc, lines, nsym, mem, pad))


def rep(lines, nsym, target=480):
    """repeat a body until an iteration has about `target` symbols (loop control then costs nothing)"""
    k = max(1, target // nsym)
    return lines * k, nsym * k


# ---- A: where the operands sit ----
def layout(name, desc, n, inv_of, F_of, g_of=None, **kw):
    L = []
    for k in range(n):
        kk = dict(kw)
        if g_of:
            kk["g"] = g_of(k)
        L += sym(inv_of(k), F_of(k), **kk)
    L, ns = rep(L, n)
    add(name, desc, L, ns)


layout("a_fixed", "inv v[64:65], F v[66:67] for every symbol", 12, lambda k: 64, lambda k: 66)
layout("a_product", "the product's layout: 12 sets, inv v[50+4k], F v[52+4k]", 12, lambda k: 50 + 4 * k, lambda k: 52 + 4 * k)
layout("a_inv_rot", "only inv rotates (v[50+4k]), F fixed v[120:121]", 12, lambda k: 50 + 4 * k, lambda k: 120)
layout("a_F_rot", "only F rotates (v[52+4k]), inv fixed v[120:121]", 12, lambda k: 120, lambda k: 52 + 4 * k)
layout("a_quad", "inv v[52+4k], F v[54+4k]: a symbol's operands in ONE aligned group of four", 12, lambda k: 52 + 4 * k, lambda k: 54 + 4 * k)
layout("a_soa", "inv v[64+2k], F v[128+2k] (two arrays)", 12, lambda k: 64 + 2 * k, lambda k: 128 + 2 * k)
layout("a_p2", "product layout, 2 sets", 2, lambda k: 50 + 4 * k, lambda k: 52 + 4 * k)
layout("a_p3", "product layout, 3 sets", 3, lambda k: 50 + 4 * k, lambda k: 52 + 4 * k)
layout("a_p4", "product layout, 4 sets", 4, lambda k: 50 + 4 * k, lambda k: 52 + 4 * k)
layout("a_p6", "product layout, 6 sets", 6, lambda k: 50 + 4 * k, lambda k: 52 + 4 * k)
layout("a_p24", "product layout, 24 sets", 24, lambda k: 50 + 4 * k, lambda k: 52 + 4 * k)
layout("a_p48", "product layout, 48 sets", 48, lambda k: 50 + 4 * k, lambda k: 52 + 4 * k)
layout("a_rt_quad", "product layout, R v[44:45] T v[46:47] (one group; mask / expo v40 v41)", 12, lambda k: 50 + 4 * k, lambda k: 52 + 4 * k, r=44, t=46, mask="v40", expo="v41")
layout("a_rt_swap", "product layout, R v[48:49] T v[46:47]", 12, lambda k: 50 + 4 * k, lambda k: 52 + 4 * k, r=48, t=46)
layout("a_rt_far", "product layout, R v[200:201] T v[204:205]", 12, lambda k: 50 + 4 * k, lambda k: 52 + 4 * k, r=200, t=204)
layout("a_stride8", "inv v[50+8k], F v[52+8k]", 12, lambda k: 50 + 8 * k, lambda k: 52 + 8 * k)
layout("a_stride6", "inv v[50+6k], F v[52+6k] (the bank pair alternates)", 12, lambda k: 50 + 6 * k, lambda k: 52 + 6 * k)
layout("a_F_then_inv", "F v[50+4k], inv v[52+4k]", 12, lambda k: 52 + 4 * k, lambda k: 50 + 4 * k)
layout("a_same_reg", "inv and F the SAME rotating pair v[50+4k] (two reads of one register)", 12, lambda k: 50 + 4 * k, lambda k: 50 + 4 * k)
layout("a_mask_sgpr", "product layout, the mask in a scalar register", 12, lambda k: 50 + 4 * k, lambda k: 52 + 4 * k, mask="s50")
layout("a_bfi", "product layout, v_bfi_b32 for the exponent step", 12, lambda k: 50 + 4 * k, lambda k: 52 + 4 * k, norm="bfi")
layout("a_magic_reg", "product layout, 1.0 from a register pair v[40:41]", 12, lambda k: 50 + 4 * k, lambda k: 52 + 4 * k, c="v[40:41]")
layout("a_g_fixed", "fixed operands, the addend of the second fma a register pair of its own (rounds 4's form)", 12, lambda k: 64, lambda k: 66, g=68)
layout("a_g_rot", "inv v[50+6k], F +2, G +4: three rotating pairs", 12, lambda k: 50 + 6 * k, lambda k: 52 + 6 * k, g_of=lambda k: 54 + 6 * k)
layout("a_no_norm", "product layout without the exponent step (two instructions)", 12, lambda k: 50 + 4 * k, lambda k: 52 + 4 * k, norm="none")
layout("a_fixed_no_norm", "fixed operands without the exponent step", 12, lambda k: 64, lambda k: 66, norm="none")

# ---- C: what an independent instruction costs, by where it stands ----
FILL = {"vmov": "v_mov_b32 v30, v31", "snop": "s_nop 0", "smov": "s_mov_b32 s52, s53", "shl64": "v_lshlrev_b64 v[32:33], 1, v[34:35]",
        "vmov_dpp": "v_mov_b32_dpp v30, v31 wave_ror:1 row_mask:0xf bank_mask:0xf"}
for fname, f in FILL.items():
    for pos in (0, 1, 2):          # after the first fma, after the second, after the exponent step
        L = []
        for k in range(12):
            s = sym(50 + 4 * k, 52 + 4 * k)
            s.insert(pos + 1, f)
            L += s
        L, ns = rep(L, 12)
        add(f"c_{fname}_{pos}", f"product layout, `{f}` after instruction {pos + 1} of every symbol", L, ns)
for pos in (0, 1, 2):
    L = []
    for k in range(12):
        s = sym(64, 66)
        s.insert(pos + 1, FILL["vmov"])
        L += s
    L, ns = rep(L, 12)
    add(f"c_fixed_vmov_{pos}", f"fixed operands, an independent v_mov after instruction {pos + 1} of every symbol", L, ns)

# ---- H: the hop ----
DPP = "wave_ror:1 row_mask:0xf bank_mask:0xf"
FP = 100                  # "the F before mine"


def lane_step(per, hop, fill=None):
    L = []
    for k in range(per):
        inv, F = 50 + 4 * (k % 12), 52 + 4 * (k % 12)
        if k < per - 1 or hop == "none":
            L += sym(inv, F)
            continue
        L.append(f"v_fma_f64 {p(T)}, {p(R)}, {p(inv)}, 1.0")
        if hop == "product":
            L += ["s_nop 1", f"v_mov_b32_dpp v{T2}, v{T} {DPP}"]
        elif hop == "fill":
            L += list(fill) + [f"v_mov_b32_dpp v{T2}, v{T} {DPP}"]
        elif hop == "nowait":       # (wrong: a stale read - what do the two wait states cost?)
            L += [f"v_mov_b32_dpp v{T2}, v{T} {DPP}"]
        elif hop == "onewait":
            L += ["s_nop 0", f"v_mov_b32_dpp v{T2}, v{T} {DPP}"]
        L.append(f"v_fma_f64 {p(R)}, {p(T2)}, {p(FP)}, -{p(FP)}")
        L.append(f"v_and_or_b32 v{R + 1}, v{R + 1}, v{MASK}, v{EXPO}")
    return L


for per in (12, 24):
    for hop, fill, d in (("none", None, "no hop"), ("product", None, "the product's hop (s_nop 1, v_mov_b32_dpp)"), ("nowait", None, "hop without wait states (stale read)"),
                         ("onewait", None, "hop with one wait state"),
                         ("fill", [FILL["vmov"], FILL["vmov"]], "hop, the wait states filled with two independent v_mov"),
                         ("fill", [FILL["shl64"], FILL["shl64"]], "hop, the wait states filled with two independent v_lshlrev_b64"),
                         ("fill", [FILL["shl64"]], "hop, ONE independent v_lshlrev_b64 in front of the DPP move"),
                         ("fill", [FILL["vmov"], FILL["vmov"], FILL["vmov"], FILL["vmov"]], "hop, four independent v_mov in front of the DPP move")):
        L, ns = rep(lane_step(per, hop, fill), per)
        tag = hop if hop != "fill" else "fill" + str(len(fill)) + ("s" if "lshl" in fill[0] else "m")
        add(f"h_{per}_{tag}", f"{per} symbols a lane, {d}", L, ns)

# ---- M: memory instructions among the symbols (v36 = lane * 144, s[54:55] = the buffer) ----
LD = {"gx3": "global_load_dwordx3 v[{d}:{d2}], v36, s[54:55] offset:{o}", "gx4": "global_load_dwordx4 v[{d}:{d3}], v36, s[54:55] offset:{o}",
      "gx2": "global_load_dwordx2 v[{d}:{d1}], v36, s[54:55] offset:{o}", "gx1": "global_load_dword v{d}, v36, s[54:55] offset:{o}",
      "dsb128": "ds_read_b128 v[{d}:{d3}], v37 offset:{o}", "dsb96": "ds_read_b96 v[{d}:{d2}], v37 offset:{o}", "dsb64": "ds_read_b64 v[{d}:{d1}], v37 offset:{o}"}


def ld(kind, i):
    d = 140 + 4 * (i % 12)
    return LD[kind].format(d=d, d1=d + 1, d2=d + 2, d3=d + 3, o=(16 * i) % 2048)


for kind in LD:
    # (i) 12 loads back to back at the head of 768 symbols, (ii) one load every 64 symbols, (iii) one load in the hop's wait states
    body = []
    for i in range(12):
        body.append(ld(kind, i))
    for j in range(64):
        body += lane_step(12, "product")
    wait = "s_waitcnt vmcnt(0) lgkmcnt(0)"
    add(f"m_{kind}_head", f"768 symbols with hops, 12 x {kind} back to back in front", [wait] + body, 768, True)
    body = []
    for j in range(64):
        if j % 5 == 0 and j // 5 < 12:
            body.append(ld(kind, j // 5))
        body += lane_step(12, "product")
    add(f"m_{kind}_spread", f"768 symbols with hops, one {kind} in front of every 5th lane step (12 in all)", [wait] + body, 768, True)
    body = []
    for j in range(64):
        if j % 5 == 0 and j // 5 < 12:
            body += lane_step(12, "fill", [ld(kind, j // 5), "s_nop 0"])
        else:
            body += lane_step(12, "product")
    add(f"m_{kind}_inhop", f"768 symbols with hops, 12 x {kind} each in a hop's wait states (load + s_nop 0 instead of s_nop 1)", [wait] + body, 768, True)
body = []
for j in range(64):
    body += lane_step(12, "product")
add("m_none", "768 symbols with hops, no loads", body, 768, True)

# ---- K: checkpoints ----
for form, d in (("product", "s_nop 0, v_readlane x 2, s_nop 2, s_store_dwordx2 in front of every 64th symbol"), ("inhop", "the two v_readlane in a hop's wait states, the store after the hop"),
                ("none", "no checkpoints")):
    body = []
    for j in range(64):
        if j % 5 == 0 and j // 5 < 12:
            if form == "product":
                body += ["s_nop 0", f"v_readlane_b32 s56, v{R}, {j}", f"v_readlane_b32 s57, v{R + 1}, {j}", "s_nop 2", f"s_store_dwordx2 s[56:57], s[58:59], 0x{8 * (j // 5):x}"]
                body += lane_step(12, "product")
            elif form == "inhop":
                body += lane_step(12, "fill", [f"v_readlane_b32 s56, v{R}, {j}", f"v_readlane_b32 s57, v{R + 1}, {j}"]) + [f"s_store_dwordx2 s[56:57], s[58:59], 0x{8 * (j // 5):x}"]
            else:
                body += lane_step(12, "product")
        else:
            body += lane_step(12, "product")
    add(f"k_{form}", f"768 symbols with hops, 12 checkpoints: {d}", body, 768, True)

# ---- P: round 2 of this probe - the address of an 8-byte instruction (everything above runs wherever the prologue left the loop) ----
for pad in range(0, 17):
    L = []
    for k in range(12):
        L += sym(50 + 4 * k, 52 + 4 * k)
    L, ns = rep(L, 12)
    add(f"p_pad{pad}", f"product layout, the loop label {4 * pad} bytes behind a 64-byte boundary", L, ns, pad=pad)
E64 = "v_mov_b32_e64 v30, v31"
HOPS = {"nop1": ["s_nop 1"], "nop0x2": ["s_nop 0", "s_nop 0"], "nop1_nop0": ["s_nop 1", "s_nop 0"], "vmov_x2": [FILL["vmov"], FILL["vmov"]], "e64_x2": [E64, E64], "e64_x1": [E64],
        "e64_nop0x2": [E64, "s_nop 0", "s_nop 0"], "shl64_x2": [FILL["shl64"], FILL["shl64"]], "nop3_nop0": ["s_nop 3", "s_nop 0"], "none": []}
for per in (12, 16, 24):
    for hn, hf in HOPS.items():
        L, ns = rep(lane_step(per, "fill", hf), per)
        add(f"q_{per}_{hn}", f"aligned loop, {per} symbols a lane, in front of the hop's DPP move: {', '.join(hf) or 'nothing (stale)'}", L, ns, pad=0)
    L, ns = rep(lane_step(per, "none"), per)
    add(f"q_{per}_nohop", f"aligned loop, {per} symbols a lane, no hop", L, ns, pad=0)
for fname, f in list(FILL.items()) + [("e64", E64), ("vmov2", FILL["vmov"] + "\\n\\t" + FILL["vmov"]), ("readlane", "v_readlane_b32 s56, v46, 5"), ("gx3", ld("gx3", 0)), ("dsb96", ld("dsb96", 0))]:
    L = []
    for k in range(12):
        s_ = sym(50 + 4 * k, 52 + 4 * k)
        s_.insert(3, f)
        L += s_
    L, ns = rep(L, 12)
    add(f"r_{fname}", f"aligned loop, `{f.replace(chr(92), '/')}` after every symbol", L, ns, pad=0)


def emit(out):
    f = open(out, "w")
    f.write("// generated by tools/probes/chain_regs_probe.py - timing only\n#include <hip/hip_runtime.h>\n#include <stdio.h>\n#include <stdint.h>\n#include <string.h>\n")
    clob = ", ".join(f'"v{i}"' for i in range(28, 256)) + ', "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s62", "s63", "s64", "s65", "scc", "vcc", "memory"'
    pro = ["s_mov_b32 s54, %[blo]", "s_mov_b32 s55, %[bhi]", "s_mov_b32 s58, %[clo]", "s_mov_b32 s59, %[chi]", "s_mov_b32 s50, 0x7fffff", "s_mov_b32 s52, 0", "s_mov_b32 s53, 0",
           "v_mbcnt_lo_u32_b32 v36, -1, 0", "v_mbcnt_hi_u32_b32 v36, -1, v36", "v_mul_u32_u24 v36, 144, v36", "v_mov_b32 v37, v36",
           f"v_mov_b32 v{MASK}, 0x7fffff", f"v_mov_b32 v{EXPO}, 0x41000000", "v_mov_b32 v40, 0", "v_mov_b32 v41, 0x3ff00000", f"v_mov_b32 v{T2}, 0", f"v_mov_b32 v{T2 + 1}, 0x3ff00000",
           "v_mov_b32 v30, 0", "v_mov_b32 v31, 0", "v_mov_b32 v32, 0", "v_mov_b32 v33, 0", "v_mov_b32 v34, 1", "v_mov_b32 v35, 0"]
    for r in (44, 46, 48, 200, 204):                       # the state wherever a variant keeps it: 2^25 * 2^-7 .. as a double
        pro += [f"v_mov_b32 v{r}, 0", f"v_mov_b32 v{r + 1}, 0x41700000"]
    pro += [f"v_mov_b32 v{MASK}, 0x7fffff", f"v_mov_b32 v{EXPO}, 0x41000000"]
    for r in range(50, 256, 2):                            # every operand pair: a plausible reciprocal (2^-45 / 40000) - F = 1000 * 2^45 would need the right slot; timing only: 1.25
        if r in (200, 204):
            continue
        pro += [f"v_mov_b32 v{r}, 0", f"v_mov_b32 v{r + 1}, 0x3ff40000"]
    pro += ["s_mov_b32 s60, %[it]", "s_nop 4", "s_memtime s[62:63]", "s_waitcnt lgkmcnt(0)", "1:"]
    epi = ["s_sub_u32 s60, s60, 1", "s_cmp_lg_u32 s60, 0", "s_cbranch_scc1 1b", "s_waitcnt vmcnt(0) lgkmcnt(0)", "s_memtime s[64:65]", "s_waitcnt lgkmcnt(0)", "s_sub_u32 %[clk], s64, s62"]
    for name, desc, lines, ns, mem, pad in KERNELS:
        f.write(f"__global__ void __launch_bounds__(64) k_{name} (uint64_t *out, uint32_t iters, const uint8_t *buf, uint8_t *ck)\n{{\n    uint32_t clocks;\n    __shared__ uint8_t lds[16384];\n")
        f.write("    if (iters == 0xffffffffu) lds[threadIdx.x] = 1;\n")
        f.write("    asm volatile (\n")
        al = [] if pad is None else [".p2align 6"] + ["s_nop 0"] * pad
        for ln in pro[:-1] + al + pro[-1:] + lines + epi:
            f.write(f'        "{ln}\\n\\t"\n')
        f.write('        : [clk] "=s" (clocks) : [it] "s" (iters), [blo] "s" ((uint32_t)(uintptr_t)buf), [bhi] "s" ((uint32_t)((uintptr_t)buf >> 32)), [clo] "s" ((uint32_t)(uintptr_t)ck), [chi] "s" ((uint32_t)((uintptr_t)ck >> 32))\n')
        f.write(f"        : {clob});\n")
        f.write("    if (!threadIdx.x) out[0] = clocks;\n    if (iters == 0xffffffffu) out[1] = lds[5];\n}\n")
    f.write("struct P { const char *name, *desc; void (*k) (uint64_t *, uint32_t, const uint8_t *, uint8_t *); uint32_t nsym; } Ps[] = {\n")
    for name, desc, lines, ns, mem, pad in KERNELS:
        f.write(f'    {{ "{name}", "{desc}", k_{name}, {ns} }},\n')
    f.write("};\n")
    f.write(r"""
int main (int argc, char **argv)
{
    uint64_t *d, h[2]; uint8_t *buf, *ck;
    if (hipMalloc (&d, 16) != hipSuccess || hipMalloc (&buf, 1 << 20) != hipSuccess || hipMalloc (&ck, 1 << 20) != hipSuccess) { printf ("no device\n"); return 1; }
    (void)hipMemset (buf, 0, 1 << 20);
    for (unsigned i = 0; i < sizeof (Ps) / sizeof (Ps[0]); i++) {
        if (argc > 1 && !strstr (Ps[i].name, argv[1])) continue;
        const uint32_t iters = 3000000u / Ps[i].nsym;
        double best = 1e30;
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL (Ps[i].k, dim3 (1), dim3 (64), 0, 0, d, iters, buf, ck);
            if (hipDeviceSynchronize () != hipSuccess) { printf ("%s: failed\n", Ps[i].name); return 1; }
            (void)hipMemcpy (h, d, 16, hipMemcpyDeviceToHost);
            const double c = (double)(uint32_t)h[0] / ((double)iters * Ps[i].nsym);
            if (c < best) best = c;
        }
        printf ("%-18s %7.3f   %s\n", Ps[i].name, best, Ps[i].desc);
        fflush (stdout);
    }
    return 0;
}
""")
    f.close()


if __name__ == "__main__":
    emit(sys.argv[1])
    print(len(KERNELS), "kernels")
