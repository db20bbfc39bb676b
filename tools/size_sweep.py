#!/usr/bin/env python3
"""Device-resident G1 MSM time and plan per size.  usage: size_sweep.py [lo=10] [hi=21]"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev, lib as ozk
L = ozk.load()
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 10
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 21
n = 1 << hi
bases = dev.gen_g1_bases(n, seed=2)
sc = np.random.default_rng(1).integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
d_sc = torch.from_numpy(sc.reshape(-1)).cuda()
for ln in range(lo, hi + 1):
    for m in ((1 << ln), (1 << ln) + (1 << ln) // 2):
        if m > n: continue
        c, w = ctypes.c_int32(), ctypes.c_int32()
        ozk.check(L.ozk_var_msm_plan(m, ctypes.byref(c), ctypes.byref(w)))
        ws = dev.VarMsmWorkspace(m, 1)
        ws.run(bases[:m * 96], d_sc[:m * 32]); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): ws.run(bases[:m * 96], d_sc[:m * 32])
        torch.cuda.synchronize()
        print("n=%8d (2^%.2f) c=%2d W=%2d  %.3f ms" % (m, np.log2(m), c.value, w.value, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
        del ws
