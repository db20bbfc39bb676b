"""Device-resident FFT / fixed-base timing probe (HIP events), with correctness spot checks."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import lib as ozk  # noqa: E402
from oracle import bn254 as o  # noqa: E402


def ptr(t):
    return int(t.data_ptr())


def main():
    L = ozk.load()
    st = int(torch.cuda.current_stream().cuda_stream)
    for logn in [int(a) for a in sys.argv[1:]] or [20, 22]:
        n = 1 << logn
        rng = np.random.default_rng(3)
        a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        a[:, 31] &= 0x1F
        d_in = torch.from_numpy(a.reshape(-1)).cuda()
        d_out = torch.empty(n * 64, dtype=torch.uint8, device="cuda")
        wsb = int(L.ozk_fft_workspace_bytes(n))
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        w = o.fr_root_of_unity(n)
        om = ctypes.create_string_buffer(o.to_le32(w), 32)
        run = lambda: ozk.check(L.ozk_fft_dev(ptr(d_in), n, ctypes.cast(om, ctypes.c_void_p), ptr(d_out), ptr(ws), wsb, st))
        run()
        torch.cuda.synchronize()
        out = d_out.cpu().numpy().reshape(n, 64)
        s = sum(int.from_bytes(r.tobytes(), "little") for r in a) % o.R
        ok = int.from_bytes(out[0].tobytes(), "little") == s
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("FFT n=2^%d ok=%s %.3f ms  %.1f Melem/s  %.1f GB/s algorithmic (64 B/elem)" % (
            logn, ok, ms, n / ms / 1e3, n * 64 / ms / 1e6), flush=True)
    # fixed base G1: n = 2^20, window 17 (the reference's choice at 2^20)
    n = 1 << 20
    window = 17
    outerc = (254 + window - 1) // window
    rng = np.random.default_rng(4)
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x1F
    d_sc = torch.from_numpy(sc.reshape(-1)).cuda()
    base = torch.from_numpy(np.frombuffer(o.g1_to_wire(o.G1.one), dtype=np.uint8).copy()).cuda()
    d_out = torch.empty(n * 192, dtype=torch.uint8, device="cuda")
    wsb = int(L.ozk_fixed_batch_msm_workspace_bytes(outerc, window, n, 1))
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    run = lambda: ozk.check(L.ozk_fixed_batch_msm_dev(outerc, window, n, ptr(base), ptr(d_sc), 1, ptr(d_out), ptr(ws), wsb, st))
    run()
    torch.cuda.synchronize()
    k = int.from_bytes(sc[5].tobytes(), "little")
    got = bytes(d_out[5 * 192:6 * 192].cpu().numpy())
    ok = got == o.g1_out_be(o.G1.to_affine(o.G1.mul(o.G1.one, k)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("FixedBase G1 n=2^20 window=17 ok=%s %.3f ms %.1f Mscalar-mul/s" % (ok, ms, n / ms / 1e3), flush=True)


if __name__ == "__main__":
    main()
