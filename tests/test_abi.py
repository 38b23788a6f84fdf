"""CPU: libgenozip_amd.so (the real hipcc build for gfx950) loads and exports every symbol include/genozip_amd.h
declares; without a GPU the product path refuses to work instead of falling back to anything."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def real_lib():
    import __graft_entry__ as g
    so = os.path.join(ROOT, "genozip_amd", "libgenozip_amd.so")
    if os.path.isdir("/opt/rocm/bin") and os.path.exists("/opt/rocm/bin/hipcc"):
        g.build()                      # (rebuilds only when a source is newer than the library)
    from genozip_amd import lib
    return lib.load(so)


def test_header_symbols_exported(real_lib):
    from genozip_amd import lib
    hdr = open(os.path.join(ROOT, "include", "genozip_amd.h")).read()
    declared = set(re.findall(r"\b(gz_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(lib.ABI_SYMBOLS), declared ^ set(lib.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(real_lib, name), name
    assert b"gfx950" in real_lib.gz_version()


def test_est_size_matches_reference_arithmetic(real_lib, oracle):
    for codec in (1, 6, 7, 8, 9, 16, 17, 18, 19):
        for n in (0, 1, 49, 50, 1000, 99999, 16 << 20):
            assert real_lib.gz_codec_est_size(codec, n) == oracle.est_size(codec, n)


def test_assign_sorter_of_the_real_library(real_lib, oracle):
    """gz_codec_assign_sort is host code of the library (the reference's codec_assign_sorter + qsort, src/codec.c:128-173,338): the
    real build's against the oracle's restatement, no GPU involved"""
    import random
    from genozip_amd.lib import GzCodecTest
    rnd = random.Random(9)
    cand = [1, 6, 7, 8, 9, 16, 17, 18, 19, 3, 5, 4]
    for _ in range(3000):
        base = rnd.choice([70, 4000, 60000])
        tests = [(c, float(int(base * rnd.choice([1, 0.99, 0.985, 0.97, 0.95, 0.5, 1.29, 1.31]))), float(rnd.choice([0, 300, 4999, 5001, 9000, 60000]) * rnd.choice([1, 0.2, 0.5, 0.8])))
                 for c in cand[:rnd.randrange(2, 13)]]
        for mode in (0, 1, 2):
            tab = (GzCodecTest * len(tests))(*[GzCodecTest(c, s, k) for c, s, k in tests])
            w = real_lib.gz_codec_assign_sort(tab, len(tests), mode)
            assert (w, [(t.codec, t.size, t.clock_us) for t in tab]) == oracle.assign_sort(tests, mode), (mode, tests)


def test_chain_loop_header_is_what_its_generator_writes(tmp_path):
    """genozip_amd/csrc/gz_chain_asm.h (the range coder chain's inner loop, one inline-asm statement) is generated: the committed file
    must be what tools/gen_chain_asm.py writes today"""
    import subprocess
    import sys
    out = tmp_path / "gz_chain_asm.h"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_chain_asm.py"), str(out)], check=True)
    assert out.read_text() == open(os.path.join(ROOT, "genozip_amd", "csrc", "gz_chain_asm.h")).read()


def test_chain_loop_instructions_are_aligned(tmp_path):
    """Round 6 (tools/probes/chain_regs_probe.py): on gfx950 an 8-byte instruction whose address is not a multiple of 8 costs a clock more,
    and the chain's loop is made of 8-byte instructions. The generator keeps 4-byte instructions in pairs behind a .p2align 3: assemble
    the committed loop (one word behind an 8-byte boundary, as an inline-asm statement may start) and look at every instruction's address;
    its own idea of the encodings' sizes is held against the assembler's on the way."""
    import re
    import shutil
    import subprocess
    import sys
    mc, dump = "/opt/rocm/lib/llvm/bin/llvm-mc", "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(mc) and os.path.exists(dump)):
        mc, dump = shutil.which("llvm-mc"), shutil.which("llvm-objdump")
    if not (mc and dump):
        pytest.skip("no llvm-mc / llvm-objdump")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_chain_asm
    src = open(os.path.join(ROOT, "genozip_amd", "csrc", "gz_chain_asm.h")).read()
    lines = re.findall(r'^\s+"(.*)\\n\\t" \\$', src, re.M)
    assert len(lines) > 2000
    for k, v in {"%[rlo]": "v0", "%[rhi]": "v1", "%[blo]": "s0", "%[bhi]": "s1", "%[nblk]": "s2", "%[clo]": "s3", "%[chi]": "s4"}.items():
        lines = [ln.replace(k, v) for ln in lines]
    s, o = tmp_path / "chain.s", tmp_path / "chain.o"
    s.write_text(".text\n.p2align 8\nchain:\n s_nop 0\n" + "\n".join(lines) + "\n")
    subprocess.run([mc, "-triple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-filetype=obj", str(s), "-o", str(o)], check=True)
    dis = subprocess.run([dump, "-d", str(o)], check=True, capture_output=True, text=True).stdout
    ins = [ln for ln in lines if not ln.endswith(":") and not ln.startswith(".")]
    got = []
    for ln in dis.splitlines():
        m = re.match(r"\s+(\S+).*//\s*([0-9A-F]+):((?: [0-9A-F]{8})+)\s*(<.*>)?\s*$", ln)
        if m:
            got.append((m.group(1), int(m.group(2), 16), 4 * len(m.group(3).split())))
    body = [g for g in got if g[1] >= 4]                      # (behind the word put in front)
    pads = len(body) - len(ins)                               # what .p2align put in
    assert 0 <= pads <= 8
    assert [g for g in body if g[2] == 8 and g[1] % 8] == []
    i = 0
    for mnem, addr, n in body:                                # the generator's size() against the assembler, instruction by instruction
        if i < len(ins) and ins[i].split()[0].replace("_e64", "").replace("_e32", "").startswith(mnem.replace("_e64", "").replace("_e32", "").replace("_dpp", "")):
            assert gen_chain_asm.size(ins[i]) == n, (ins[i], n)
            i += 1
        else:
            assert mnem == "s_nop", (mnem, ins[i] if i < len(ins) else None)
    assert i == len(ins)


def test_no_cpu_fallback(real_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    err = ctypes.c_int(0)
    h = real_lib.gz_create(0, None, ctypes.byref(err))
    assert not h and err.value < 0
    from genozip_amd.codec import Engine
    with pytest.raises(RuntimeError):
        Engine(device=0)


def test_oracle_is_not_linked_into_the_product():
    so = os.path.join(ROOT, "genozip_amd", "libgenozip_amd.so")
    blob = open(so, "rb").read()
    assert b"gzo_" not in blob and b"liboracle" not in blob and b"htsref" not in blob
    for f in os.listdir(os.path.join(ROOT, "genozip_amd")):
        if f.endswith(".py"):
            src = open(os.path.join(ROOT, "genozip_amd", f)).read()
            assert "oracle" not in src.replace("no oracle", ""), f


def _build_c_program(lib_dir, lib_name, out):
    import subprocess
    src = os.path.join(ROOT, "tests", "c", "zip_fastq.c")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", out,
                    "-L", lib_dir, "-l:" + lib_name, "-Wl,-rpath," + lib_dir], check=True)
    return out


def test_c_host_program_against_the_header(emul_engine, tmp_path):
    """a plain C11 program (tests/c/zip_fastq.c) compiled against include/genozip_amd.h drives the whole path - text ->
    seg -> merge -> generate -> compress -> decode - without Python; here linked with the CPU-emulated build of the product
    sources, on the GPU box with the real library (tests/test_gpu.py)"""
    import subprocess
    exe = _build_c_program(os.path.join(ROOT, "tests", "emul"), "libgenozip_amd_emul.so", str(tmp_path / "zip_fastq_emul"))
    r = subprocess.run([exe, "600"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_assign_golden_vectors_of_the_real_library(real_lib):
    """row a8 PINNED, host side of the real build (no GPU involved): gz_codec_assign_sort against the order the reference's own
    codec_assign_sorter + qsort produced, gz_codec_assign_rule - what the VBlock compute driver decides with - against what the
    reference's own codec_assign_best_codec did (tests/golden/assign_golden.json)"""
    import parity
    from genozip_amd.lib import GzCodecTest

    def sorter(tests, mode):
        tab = (GzCodecTest * len(tests))(*[GzCodecTest(int(c), float(s), float(k)) for c, s, k in tests])
        w = real_lib.gz_codec_assign_sort(tab, len(tests), mode)
        return w, [(t.codec, t.size, t.clock_us) for t in tab]
    assert parity.assign_golden_sort(sorter) == 780
    assert parity.assign_golden_rule(real_lib.gz_codec_assign_rule) > 100


def test_merge_golden_vectors_of_the_real_library(real_lib, oracle):
    """the LOOP of row a4 PINNED, host code of the real build (no GPU involved): gz_ctx_merge against what the reference's own
    ctx_merge_in_one_vctx did (tests/golden/merge_golden.json, oracle/ref_merge_shim.c)"""
    import parity
    from genozip_amd.codec import Zctx
    assert parity.merge_loop_golden(lambda est: Zctx(real_lib, est), oracle.ctx_seg_column) >= 50


def test_order_golden_vectors_of_the_real_library(real_lib):
    """row a15 PINNED, host code of the real build: gz_section_order - what the VBlock compute driver orders its sections with - against
    the order the reference's own zip_compress_all_contexts_local / _b250 produced (tests/golden/order_golden.json, oracle/ref_order_shim.c)"""
    import parity
    assert parity.section_order_golden(parity.lib_section_order(real_lib)) == 360


def test_c_multi_gpu_route_compiles_against_the_headers():
    """tests/c/zip_fastq_rccl.c - the N-GPU route for a host written in C: gz_fastq_zip_seg / _merge / _finish / _collect with the merge blobs,
    the votes and the finished z_data carried by RCCL's own C API (ncclAllGather, ncclSend / ncclRecv) - type-checks against
    include/genozip_amd.h and <rccl/rccl.h>: the exchange format is the C-ABI's, the transport the host's"""
    import subprocess
    if not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("no RCCL headers here")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                    "-fsyntax-only", os.path.join(ROOT, "tests", "c", "zip_fastq_rccl.c")], check=True)
