"""
Device plumbing: torch (ROCm) supplies HBM allocations, the current HIP stream and
torch.distributed (RCCL); every computation goes through libdedalus_hip.so.
No CPU fallback: constructing a Device without a visible gfx950 GPU raises.
"""

import ctypes as C
import os

import numpy as np

from . import libhip


class Device:
    _instance = None

    def __init__(self, index=None):
        import torch
        if not torch.cuda.is_available():
            raise libhip.DdhError("no HIP device visible: dedalus_amd has no CPU fallback")
        if index is None:
            index = int(os.environ.get("LOCAL_RANK", "0"))
        self.torch = torch
        self.index = index
        torch.cuda.set_device(index)
        self.tdev = torch.device("cuda", index)
        libhip.load()
        libhip.call("ddh_init", index)

    @classmethod
    def get(cls):
        if cls._instance is None:
            cls._instance = Device()
        return cls._instance

    # -- memory ---------------------------------------------------------------------------------
    def empty(self, shape, dtype=np.float64):
        t = self.torch
        td = {np.dtype(np.float64): t.float64, np.dtype(np.complex128): t.complex128,
              np.dtype(np.int32): t.int32}[np.dtype(dtype)]
        return t.empty(tuple(int(s) for s in np.atleast_1d(shape)), dtype=td, device=self.tdev)

    def zeros(self, shape, dtype=np.float64):
        a = self.empty(shape, dtype)
        a.zero_()
        return a

    def from_host(self, a):
        a = np.ascontiguousarray(a)
        return self.torch.from_numpy(a).to(self.tdev)

    def to_host(self, t):
        return t.detach().cpu().numpy()

    @property
    def stream(self):
        return C.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def sync(self):
        self.torch.cuda.synchronize()


def ptr(t):
    return C.c_void_p(t.data_ptr())
