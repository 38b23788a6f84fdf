import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emul")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")
    config.addinivalue_line("markers", "thorough: a further variant of a CPU-emulated check that the default run leaves out to stay within "
                                       "minutes (GZ_TEST_THOROUGH=1 runs them; the GPU suite runs the same checks at full size anyway)")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("GZ_TEST_THOROUGH"):
        return
    skip = pytest.mark.skip(reason="thorough variant: GZ_TEST_THOROUGH=1")
    for it in items:
        if "thorough" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    pyoracle.build(ref=False)
    return pyoracle.Oracle()


@pytest.fixture(scope="session")
def ref():
    """the reference's own htscodecs, compiled in place -- only where oracle/_ref has been built"""
    import pyoracle
    if not pyoracle.Ref.available():
        if os.path.isdir("/root/reference/src/htscodecs"):
            pyoracle.build(ref=True)
        else:
            pytest.skip("oracle/_ref not built (reference sources absent)")
    return pyoracle.Ref()


@pytest.fixture(scope="session")
def emul_engine():
    """the product sources compiled against the CPU stand-in of the HIP runtime (tests/emul) -- logic tests only"""
    so = os.path.join(ROOT, "tests", "emul", "libgenozip_amd_emul.so")
    srcs = [os.path.join(ROOT, "genozip_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "genozip_amd", "csrc"))]
    srcs.append(os.path.join(ROOT, "tests", "emul", "hip", "hip_runtime.h"))
    def stale():
        return not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale():
        import fcntl                                   # (pytest-xdist: one worker builds, the others wait for it)
        with open(so + ".lock", "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            if stale():
                subprocess.run(["sh", os.path.join(ROOT, "tests", "emul", "build_emul.sh")], check=True)
    from hostmem import HostMem
    from genozip_amd.codec import Engine
    return Engine(lib_path=so, mem=HostMem())


@pytest.fixture(scope="session")
def gpu_engine():
    """the real thing: libgenozip_amd.so on cuda:0. Fails (does not skip) if the HIP library is missing."""
    from genozip_amd.codec import Engine
    return Engine(device=0)
