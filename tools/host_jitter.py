#!/usr/bin/env python3
"""Call-time distribution (N calls each, fresh pageable inputs) of three host entry points.  usage: host_jitter.py [N=30]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev, lib as ozk
from oracle import bn254 as o
L = ozk.load()
vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << 20
rng = np.random.default_rng(1)
sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
g1 = np.frombuffer(bytes(dev.gen_g1_bases(n, seed=2).cpu().numpy()), dtype=np.uint8)
m = n // 4
pts = [np.frombuffer(o.g2_to_wire(o.G2.to_affine(o.G2.mul(o.G2.one, int(k)))), dtype=np.uint8) for k in rng.integers(1, 1 << 62, size=64)]
g2 = np.ascontiguousarray(np.stack(pts)[rng.integers(0, 64, size=m)]).reshape(-1)
bw = np.frombuffer(o.g1_to_wire(o.G1.one), dtype=np.uint8)
out = np.zeros(576, dtype=np.uint8)
fout = np.zeros(n * 192, dtype=np.uint8)
def stat(name, f, mk):
    ts = []
    for _ in range(N + 1):
        a = mk()
        t0 = time.perf_counter(); f(*a); ts.append((time.perf_counter() - t0) * 1e3)
    ts = sorted(ts[1:])
    print("%-34s min %.2f median %.2f p90 %.2f max %.2f ms  (>2x median: %d of %d)" % (name, ts[0], ts[len(ts) // 2], ts[int(len(ts) * 0.9)], ts[-1], sum(t > 2 * ts[len(ts) // 2] for t in ts), len(ts)), flush=True)
cp = lambda a: np.array(a, copy=True)
stat("ozk_var_msm_host G1 2^20", lambda b, s: ozk.check(L.ozk_var_msm_host(vp(b), vp(s), n, 1, 0, vp(out))), lambda: (cp(g1), cp(sc)))
stat("ozk_var_double_msm_host 2^18", lambda b1, b2, s: ozk.check(L.ozk_var_double_msm_host(vp(b1), vp(b2), vp(s), m, 0, vp(out))), lambda: (cp(g1[:m * 96]), cp(g2), cp(sc[:m])))
stat("ozk_fixed_batch_msm_host G1 2^20", lambda s: ozk.check(L.ozk_fixed_batch_msm_host(15, 17, 15, 1 << 17, n, 254, vp(bw), vp(s), 1, 0, vp(fout))), lambda: (cp(sc),))
