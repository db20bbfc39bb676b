import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev
logn = int(sys.argv[1])
n = 1 << logn
bases = dev.gen_g1_bases(n, seed=2)
sc = np.random.default_rng(1).integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
d_sc = torch.from_numpy(sc.reshape(-1)).cuda()
ws = dev.VarMsmWorkspace(n, 1)
for _ in range(3):
    ws.run(bases, d_sc); torch.cuda.synchronize()
