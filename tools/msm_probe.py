"""First-light probe: device-resident G1 MSM at several sizes, checked against the
discrete-log identity, timed with HIP events.  Usage: python tools/msm_probe.py [logn ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev  # noqa: E402
from oracle import bn254 as o  # noqa: E402  (checker only)


def rand_scalars(n, seed):
    rng = np.random.default_rng(seed)
    b = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    b[:, 31] &= 0x1F  # < 2^253 < r
    return b


def main():
    logs = [int(a) for a in sys.argv[1:]] or [10, 16, 20]
    for logn in logs:
        n = 1 << logn
        t0 = time.time()
        bases = dev.gen_g1_bases(n, seed=2)
        sc_np = rand_scalars(n, 1)
        scalars = torch.from_numpy(sc_np.reshape(-1)).cuda()
        ws = dev.VarMsmWorkspace(n)
        torch.cuda.synchronize()
        t1 = time.time()
        out = ws.run(bases, scalars)
        torch.cuda.synchronize()
        t2 = time.time()
        res = bytes(out.cpu().numpy())
        ks = dev.gen_base_logs(n, 2)
        acc = 0
        for i in range(n):
            acc += int.from_bytes(sc_np[i].tobytes(), "little") * ks[i]
        want = o.g1_out_le(o.G1.to_affine(o.G1.mul(o.G1.one, acc % o.R)))
        ok = res == want
        reps = 5
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(reps):
            ws.run(bases, scalars)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / reps
        print("n=2^%d ok=%s first=%.1f ms steady=%.3f ms  %.1f Mscalar-mul/s (setup %.1fs)" % (
            logn, ok, (t2 - t1) * 1e3, ms, n / ms / 1e3, t1 - t0), flush=True)


if __name__ == "__main__":
    main()
