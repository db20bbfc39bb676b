"""Device-resident entry points on torch CUDA(HIP) tensors: torch is plumbing only (HBM
buffers, streams, torch.distributed); all arithmetic is in libozk_hip.so."""
import torch

from . import lib as _lib


def _ptr(t):
    return int(t.data_ptr())


def _stream():
    return int(torch.cuda.current_stream().cuda_stream)


class VarMsmWorkspace:
    """Pre-allocated workspace + output for repeated device-resident MSMs of size n."""

    def __init__(self, n, type_=1, device="cuda"):
        L = _lib.load()
        self.n, self.type = n, type_
        self.bytes = int(L.ozk_var_msm_workspace_bytes(n, type_))
        if self.bytes == 0:
            raise _lib.OzkError("workspace size query failed")
        self.ws = torch.empty(self.bytes, dtype=torch.uint8, device=device)
        self.out = torch.zeros(192 if type_ == 1 else 384, dtype=torch.uint8, device=device)

    def run(self, d_bases, d_scalars):
        """d_bases: uint8 [n*96|192], d_scalars: uint8 [n*32] — wire format, in HBM.
        Asynchronous on the current stream; returns the output tensor (192|384 B)."""
        L = _lib.load()
        _lib.check(L.ozk_var_msm_dev(_ptr(d_bases), _ptr(d_scalars), self.n, self.type, _ptr(self.out),
                                     _ptr(self.ws), self.bytes, _stream()))
        return self.out


def gen_g1_bases(n, seed, device="cuda"):
    L = _lib.load()
    out = torch.empty(n * 96, dtype=torch.uint8, device=device)
    _lib.check(L.ozk_gen_bases_dev(seed, n, 1, _ptr(out), _stream()))
    return out


def points_sum(d_points, k, type_=1):
    L = _lib.load()
    out = torch.zeros(192 if type_ == 1 else 384, dtype=torch.uint8, device=d_points.device)
    _lib.check(L.ozk_points_sum_dev(_ptr(d_points), k, type_, _ptr(out), _stream()))
    return out


SPLITMIX_MASK = (1 << 64) - 1


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & SPLITMIX_MASK
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & SPLITMIX_MASK
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & SPLITMIX_MASK
    return x ^ (x >> 31)


def gen_base_logs(n, seed):
    """the k_i of gen_g1_bases, for CPU-side checking"""
    out = []
    for i in range(n):
        k = splitmix64((seed + i) & SPLITMIX_MASK)
        out.append(k if k else 1)
    return out
