#!/usr/bin/env python3
"""ozk_var_msm_host wall time against OZK_HOST_SLICES (G1; G2 with repeated bases), and the device-resident MSM
at the slice sizes.  usage: host_slices_probe.py LOGN [K ...]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev  # noqa: E402
from octopuszk_amd import lib as ozk  # noqa: E402
from oracle import bn254 as o  # noqa: E402

L = ozk.load()
vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ks = [int(x) for x in sys.argv[2:]] or [1, 2, 4]
    n = 1 << logn
    rng = np.random.default_rng(1)
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x1F
    g1 = np.frombuffer(bytes(dev.gen_g1_bases(n, seed=2).cpu().numpy()), dtype=np.uint8)
    G = o.G2
    pts = [np.frombuffer(o.g2_to_wire(G.to_affine(G.mul(G.one, int(k)))), dtype=np.uint8) for k in rng.integers(1, 1 << 62, size=64)]
    g2 = np.ascontiguousarray(np.stack(pts)[rng.integers(0, 64, size=n)]).reshape(-1)
    for ln in range(16, logn + 1):
        m = 1 << ln
        d_b, d_s = torch.from_numpy(g1[:m * 96].copy()).cuda(), torch.from_numpy(sc[:m].reshape(-1).copy()).cuda()
        ws = dev.VarMsmWorkspace(m, 1)
        ws.run(d_b, d_s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ws.run(d_b, d_s)
        torch.cuda.synchronize()
        print("device-resident G1 2^%d: %.2f ms" % (ln, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
        del ws
    for type_, bases, ob in ((1, g1, 192), (2, g2, 384)):
        ref = None
        for k in ks:
            os.environ["OZK_HOST_SLICES"] = str(k)
            ozk.check(L.ozk_tuning_reload())
            out = np.zeros(ob, dtype=np.uint8)
            ts = []
            for _ in range(6):
                b, s = np.array(bases, copy=True), np.array(sc, copy=True)
                t0 = time.perf_counter()
                ozk.check(L.ozk_var_msm_host(vp(b), vp(s), n, type_, 0, vp(out)))
                ts.append((time.perf_counter() - t0) * 1e3)
            ref = ref or bytes(out)
            assert bytes(out) == ref
            print("G%d 2^%d slices=%-2d  min %.2f median %.2f ms" % (type_, logn, k, min(ts[1:]), sorted(ts[1:])[2]), flush=True)


main()
