#!/usr/bin/env python3
"""Wall-time distribution of ozk_fixed_batch_msm_host (G1, window 17, 2^20 scalars).  usage: fb_host_probe.py [reps] [compact]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import lib as ozk
from oracle import bn254 as o
L = ozk.load()
vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
compact = len(sys.argv) > 2
n = 1 << 20
sc = np.random.default_rng(1).integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
bw = np.frombuffer(o.g1_to_wire(o.G1.one), dtype=np.uint8)
out = np.zeros(n * (96 if compact else 192), dtype=np.uint8)
ts = []
for _ in range(reps + 1):
    s = np.array(sc, copy=True)
    t0 = time.perf_counter()
    if compact:
        ozk.check(L.ozk_fixed_batch_msm_compact_host(15, 17, n, vp(bw), vp(s), 1, 0, vp(out)))
    else:
        ozk.check(L.ozk_fixed_batch_msm_host(15, 17, 15, 1 << 17, n, 254, vp(bw), vp(s), 1, 0, vp(out)))
    ts.append((time.perf_counter() - t0) * 1e3)
print("slices=%s hwq=%s: min %.2f median %.2f max %.2f | %s" % (os.environ.get("OZK_HOST_SLICES", "-"), os.environ.get("GPU_MAX_HW_QUEUES", "-"),
      min(ts[1:]), sorted(ts[1:])[len(ts) // 2], max(ts[1:]), " ".join("%.1f" % t for t in ts[1:])), flush=True)
