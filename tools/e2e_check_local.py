"""Build container: the .genozip files the HIP library wrote on the GPU box (tools/e2e_gpu_files.py -> gpurun_out/e2e/) through the
REFERENCE'S OWN genounzip (untarred from /root/reference/installers into a temporary directory), compared with the texts the same
generators give here. Prints one line per file; the summary goes to profiles/."""
import hashlib
import json
import os
import subprocess
import sys
import tarfile
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")]


def main():
    import e2e_gpu_files as gen
    src = os.path.join(ROOT, "gpurun_out", "e2e")
    made = json.load(open(os.path.join(src, "made.json")))
    r1, r2, sam, bam, vcf_hdr, vcf = gen.texts(made["n"])
    hdr = b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:248956422\n"
    want = {"reads_R1.fq": r1, "reads_R2.fq": r2, "out.sam": hdr + sam, "out.vcf": vcf_hdr + b"".join(vcf)}
    ok = True
    with tempfile.TemporaryDirectory() as d:
        with tarfile.open("/root/reference/installers/genozip-linux-x86_64.tar") as t:
            t.extractall(d)
        exe = os.path.join(d, "genozip-linux-x86_64", "genounzip")
        for f, args, outs in (("pair.genozip", ["-f", "pair.genozip"], ["reads_R1.fq", "reads_R2.fq"]), ("reads.sam.genozip", ["-f", "-o", "out.sam", "reads.sam.genozip"], ["out.sam"]),
                              ("cohort.vcf.genozip", ["-f", "-o", "out.vcf", "cohort.vcf.genozip"], ["out.vcf"])):
            blob = open(os.path.join(src, f), "rb").read()
            open(os.path.join(d, f), "wb").write(blob)
            p = subprocess.run([exe] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
            for o in outs:
                got = open(os.path.join(d, o), "rb").read() if os.path.exists(os.path.join(d, o)) else b""
                same = got == want[o]
                ok &= same
                print("%-20s %9d bytes of .genozip (made by %s, library %s) -> genounzip -> %-12s %10d bytes, sha256 %s : %s" %
                      (f, len(blob), made.get("engine", "?"), made["version"], o, len(got), hashlib.sha256(got).hexdigest()[:16], "IDENTICAL to the input text" if same else "DIFFERS " + p.stdout.decode(errors="replace")[-300:]))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
