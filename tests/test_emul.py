"""CPU: the UNMODIFIED product sources (host orchestrator + kernels), compiled against the CPU stand-in of the HIP
runtime in tests/emul, checked against the oracle on small inputs. Catches logic errors without a GPU; the same
checks run at full size on the GPU in tests/test_gpu.py."""
import parity


def test_emul_codec_edge_cases(emul_engine, oracle):
    parity.codec_edge_cases(emul_engine, oracle, max_n=4097)


def test_emul_host_call_surface(emul_engine, oracle):
    parity.host_call_surface(emul_engine, oracle)


def test_emul_golden_small(emul_engine):
    assert parity.golden(emul_engine, max_n=1000) > 1500


def test_emul_assign_best(emul_engine, oracle):
    parity.assign_best(emul_engine, oracle, n=6000)


def test_emul_b250(emul_engine, oracle):
    parity.b250(emul_engine, oracle, 4000)


def test_emul_local(emul_engine, oracle):
    parity.local(emul_engine, oracle, 37, 50)


def test_emul_vblocks(emul_engine, oracle):
    parity.vblocks(emul_engine, oracle, 3, 6000)
