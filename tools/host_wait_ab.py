#!/usr/bin/env python3
"""A/B of the host waits inside one process: blocks of 20 calls alternate between polling waits (default) and the
runtime's blocking waits (OZK_HOST_BLOCKING_WAITS=1, read by the library at every wait), six rounds, for three entry
points.  Prints min / median / p90 / max per mode."""
import ctypes, gc, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev, lib as ozk
from oracle import bn254 as o
gc.disable()
L = ozk.load()
vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
n = 1 << 20
rng = np.random.default_rng(1)
sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
g1 = np.frombuffer(bytes(dev.gen_g1_bases(n, seed=2).cpu().numpy()), dtype=np.uint8)
bw = np.frombuffer(o.g1_to_wire(o.G1.one), dtype=np.uint8)
out_small = np.zeros(576, dtype=np.uint8); out_fixed = np.zeros(n * 192, dtype=np.uint8)
m = n // 4
g2 = np.frombuffer(bytes(dev.gen_g1_bases(2 * m, seed=5).cpu().numpy()), dtype=np.uint8)   # (192 m bytes: any field elements do for timing)
cp = lambda a: np.array(a, copy=True)
def var():
    b, s = cp(g1), cp(sc); t0 = time.perf_counter()
    ozk.check(L.ozk_var_msm_host(vp(b), vp(s), n, 1, 0, vp(out_small))); return (time.perf_counter() - t0) * 1e3
def fixed():
    s = cp(sc); t0 = time.perf_counter()
    ozk.check(L.ozk_fixed_batch_msm_host(15, 17, 15, 1 << 17, n, 254, vp(bw), vp(s), 1, 0, vp(out_fixed))); return (time.perf_counter() - t0) * 1e3
def dbl():
    b1, s = cp(g1[:m * 96]), cp(sc[:m]); t0 = time.perf_counter()
    ozk.check(L.ozk_var_msm_host(vp(b1), vp(s), m, 1, 0, vp(out_small))); return (time.perf_counter() - t0) * 1e3
def dbl2():
    b1, b2, s = cp(g1[:m * 96]), cp(g2), cp(sc[:m]); t0 = time.perf_counter()
    ozk.check(L.ozk_var_double_msm_host(vp(b1), vp(b2), vp(s), m, 0, vp(out_small))); return (time.perf_counter() - t0) * 1e3
calls = {"ozk_var_double_msm_host 2^18": dbl2, "ozk_var_msm_host G1 2^20": var, "ozk_fixed_batch_msm_host G1 2^20": fixed, "ozk_var_msm_host G1 2^18": dbl}
for f in calls.values(): f()
res = {(k, md): [] for k in calls for md in ("poll", "block")}
for rnd in range(6):
    for md in ("poll", "block"):
        os.environ["OZK_HOST_BLOCKING_WAITS"] = "1" if md == "block" else "0"
        for k, f in calls.items():
            for _ in range(20): res[(k, md)].append(f())
for (k, md), ts in res.items():
    ts = sorted(ts)
    print("%-34s %-5s min %.2f median %.2f p90 %.2f max %.2f ms (%d calls)" % (k, md, ts[0], ts[len(ts) // 2], ts[int(len(ts) * 0.9)], ts[-1], len(ts)))
