#!/usr/bin/env python3
"""Wall time of ozk_var_double_msm_host (G1 + G2 over the same scalars) against OZK_HOST_SLICES.  usage: double_host_probe.py LOGN [K ...]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev, lib as ozk
from oracle import bn254 as o
L = ozk.load()
vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ks = [int(x) for x in sys.argv[2:]] or [1, 4]
n = 1 << logn
rng = np.random.default_rng(1)
sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
g1 = np.frombuffer(bytes(dev.gen_g1_bases(n, seed=2).cpu().numpy()), dtype=np.uint8)
pts = [np.frombuffer(o.g2_to_wire(o.G2.to_affine(o.G2.mul(o.G2.one, int(k)))), dtype=np.uint8) for k in rng.integers(1, 1 << 62, size=64)]
g2 = np.ascontiguousarray(np.stack(pts)[rng.integers(0, 64, size=n)]).reshape(-1)
ref = None
for k in ks:
    os.environ["OZK_HOST_SLICES"] = str(k)
    ozk.check(L.ozk_tuning_reload())
    out = np.zeros(576, dtype=np.uint8)
    ts = []
    for _ in range(5):
        a, b, s = np.array(g1, copy=True), np.array(g2, copy=True), np.array(sc, copy=True)
        t0 = time.perf_counter()
        ozk.check(L.ozk_var_double_msm_host(vp(a), vp(b), vp(s), n, 0, vp(out)))
        ts.append((time.perf_counter() - t0) * 1e3)
    ref = ref or bytes(out)
    assert bytes(out) == ref
    print("double MSM 2^%d slices=%-2d  min %.2f median %.2f ms  (%d MiB over PCIe)" % (logn, k, min(ts[1:]), sorted(ts[1:])[2], n * 320 >> 20), flush=True)
