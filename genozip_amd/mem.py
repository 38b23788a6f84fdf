"""Device memory plumbing: PyTorch owns HBM buffers and streams, the C-ABI gets raw pointers."""
import numpy as np


class TorchMem:
    """HBM buffers as torch uint8 tensors on cuda:<device> (ROCm)."""

    def __init__(self, device=0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("genozip_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.torch = torch
        self.device = torch.device("cuda", device)

    def alloc(self, nbytes):
        return self.torch.empty(max(1, int(nbytes)), dtype=self.torch.uint8, device=self.device)

    def upload(self, data):
        arr = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data.view(np.uint8).reshape(-1)
        if arr.size == 0:
            return self.alloc(1)
        arr = np.ascontiguousarray(arr)
        if not arr.flags.writeable:                      # (bytes objects give read-only views; torch wants to own a writable array)
            arr = arr.copy()
        return self.torch.from_numpy(arr).to(self.device)

    @staticmethod
    def ptr(buf):
        return buf.data_ptr()

    def download(self, buf, nbytes=None):
        t = buf if nbytes is None else buf[:nbytes]
        return t.cpu().numpy().tobytes()

    def sync(self):
        self.torch.cuda.synchronize(self.device)
