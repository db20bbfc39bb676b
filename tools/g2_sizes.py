#!/usr/bin/env python3
"""Device-resident G2 VarMSM time by size (bases = multiples of the generator made by the fixed-base entry)."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev, lib as ozk
from oracle import bn254 as o
L = ozk.load()
ptr = lambda t: int(t.data_ptr())
st = int(torch.cuda.current_stream().cuda_stream)
nmax = 1 << 21
rng = np.random.default_rng(3)
ks = rng.integers(0, 256, size=(nmax, 32), dtype=np.uint8); ks[:, 8:] = 0
base = torch.from_numpy(np.frombuffer(o.g2_to_wire(o.G2.one), dtype=np.uint8).copy()).cuda()
bases = torch.empty(nmax * 192, dtype=torch.uint8, device="cuda")
wsb = int(L.ozk_fixed_batch_msm_workspace_bytes(16, 16, nmax, 2))
wsf = torch.empty(wsb, dtype=torch.uint8, device="cuda")
d_ks = torch.from_numpy(ks.reshape(-1)).pin_memory().cuda()
ozk.check(L.ozk_fixed_batch_msm_compact_dev(16, 16, nmax, ptr(base), ptr(d_ks), 2, ptr(bases), ptr(wsf), wsb, st))
torch.cuda.synchronize(); del wsf
sc = rng.integers(0, 256, size=(nmax, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
d_sc = torch.from_numpy(sc.reshape(-1)).pin_memory().cuda()
for logn in (16, 17, 18, 19, 20, 21):
    n = 1 << logn
    ws = dev.VarMsmWorkspace(n, 2)
    b, s = bases[:n * 192], d_sc[:n * 32]
    ws.run(b, s); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8): ws.run(b, s)
    torch.cuda.synchronize()
    print("G2 n=2^%d: %.3f ms" % (logn, (time.perf_counter() - t0) / 8 * 1e3), flush=True)
