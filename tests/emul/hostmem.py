"""TEST INFRASTRUCTURE: 'device' buffers for the CPU-emulated build (tests/emul) are plain host arrays."""
import numpy as np


class HostMem:
    def alloc(self, nbytes):
        return np.full(max(1, int(nbytes)), 0xA5, dtype=np.uint8)

    def upload(self, data):
        arr = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data.view(np.uint8).reshape(-1)
        return arr.copy() if arr.size else self.alloc(1)

    @staticmethod
    def ptr(buf):
        return buf.ctypes.data

    def download(self, buf, nbytes=None):
        return (buf if nbytes is None else buf[:nbytes]).tobytes()

    def sync(self):
        pass


class TorchCpuMem:
    """the same for callers that keep their buffers as torch tensors (bench.py's workload classes under GZ_BENCH_EMUL=1: the N > 1 path
    of bench.py on CPU ranks with the gloo backend - tests/test_shard.py)"""

    def __init__(self):
        import torch
        self.torch = torch
        self.device = torch.device("cpu")

    def alloc(self, nbytes):
        return self.torch.full((max(1, int(nbytes)),), 0xA5, dtype=self.torch.uint8)

    def upload(self, data):
        arr = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data.view(np.uint8).reshape(-1)
        return self.torch.from_numpy(arr.copy()) if arr.size else self.alloc(1)

    @staticmethod
    def ptr(buf):
        return buf.data_ptr()

    def download(self, buf, nbytes=None):
        return (buf if nbytes is None else buf[:nbytes]).numpy().tobytes()

    def sync(self):
        pass
