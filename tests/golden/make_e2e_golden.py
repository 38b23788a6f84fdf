#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (build container only): makes tests/golden/e2e_sha256.json.
Every file of tests/e2e_files.py is written by the product's unmodified sources on the CPU stand-in of the HIP runtime (tests/emul),
decoded by the REFERENCE'S OWN `genounzip` (15.0.86, untarred from /root/reference/installers into a temporary directory) and compared with
the original text byte for byte; only then its sha256 and size are recorded. tests/test_gpu.py::test_e2e_files_sha256 writes the same
files with the HIP library on the MI355X and asserts the same sha256: a file the GPU wrote is then a file the reference has read.

    python tests/golden/make_e2e_golden.py
"""
import json
import os
import subprocess
import sys
import tarfile
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emul")):
    sys.path.insert(0, p)
TAR = "/root/reference/installers/genozip-linux-x86_64.tar"


def main():
    import e2e_files
    from hostmem import HostMem
    from genozip_amd.codec import Engine
    so = os.path.join(ROOT, "tests", "emul", "libgenozip_amd_emul.so")
    subprocess.run(["sh", os.path.join(ROOT, "tests", "emul", "build_emul.sh")], check=True)
    E = Engine(lib_path=so, mem=HostMem())
    lz = e2e_files.lzma_sub()
    files = {}
    with tempfile.TemporaryDirectory() as d:
        with tarfile.open(TAR) as t:
            t.extractall(d)
        exe = os.path.join(d, "genozip-linux-x86_64", "genounzip")
        for name, make in e2e_files.CASES.items():
            blob, texts, args = make(E, lz)
            w = os.path.join(d, name)
            os.mkdir(w)
            open(os.path.join(w, name + ".genozip"), "wb").write(blob)
            # (the decoder ends with a segmentation fault AFTER its output is complete in this sandbox: in its exit path's walk over the machine's
            #  shared memory segments, as the call stack it prints says - tests/test_e2e_genounzip.py accepts that crash and no other)
            p = subprocess.run([exe, "-f"] + args + [name + ".genozip"], cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
            from test_e2e_genounzip import exit_is_clean_or_the_known_exit_path_crash
            assert exit_is_clean_or_the_known_exit_path_crash(p.returncode, p.stdout.decode(errors="replace")), (name, p.returncode, p.stdout.decode(errors="replace")[-2000:])
            for fn, want in texts.items():
                got = open(os.path.join(w, fn), "rb").read() if os.path.exists(os.path.join(w, fn)) else None
                assert got == want, (name, fn, p.stdout.decode(errors="replace")[:2000])
            files[name] = {"sha256": e2e_files.sha(blob), "size": len(blob), "decoded": sorted(texts)}
            print(name, len(blob), files[name]["sha256"][:16], "decoded by genounzip:", ", ".join(sorted(texts)))
    json.dump({"made_by": "tests/golden/make_e2e_golden.py", "decoder": "genounzip 15.0.86 (the reference's shipped binary)", "files": files},
              open(e2e_files.GOLDEN, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
