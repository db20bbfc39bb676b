"""Time one device-resident G1 MSM of 2^logn pairs (median of 7), for knob sweeps: msm_time.py <logn>"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev
logn = int(sys.argv[1]); n = 1 << logn
bases = dev.gen_g1_bases(n, seed=2)
sc = np.random.default_rng(1).integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
d_sc = torch.from_numpy(sc.reshape(-1)).cuda()
ws = dev.VarMsmWorkspace(n, 1)
ts = []
for _ in range(9):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ws.run(bases, d_sc); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("2^%d L1=%s: %.3f ms" % (logn, os.environ.get("OZK_MSM_L1", "auto"), sorted(ts[2:])[3] * 1e3), flush=True)
