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
