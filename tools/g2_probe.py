import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev, lib as ozk
from oracle import bn254 as o
L = ozk.load(); st = int(torch.cuda.current_stream().cuda_stream)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n = 1 << logn
rng = np.random.default_rng(5)
sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
ksc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); ksc[:, 8:] = 0
d_sc = torch.from_numpy(sc.reshape(-1)).cuda(); d_k = torch.from_numpy(ksc.reshape(-1)).cuda()
base2 = torch.from_numpy(np.frombuffer(o.g2_to_wire(o.G2.one), dtype=np.uint8).copy()).cuda()
out_be = torch.empty(n * 384, dtype=torch.uint8, device="cuda")
wsb = int(L.ozk_fixed_batch_msm_workspace_bytes(4, 16, n, 2)); wsf = torch.empty(wsb, dtype=torch.uint8, device="cuda")
ozk.check(L.ozk_fixed_batch_msm_dev(4, 16, n, int(base2.data_ptr()), int(d_k.data_ptr()), 2, int(out_be.data_ptr()), int(wsf.data_ptr()), wsb, st))
torch.cuda.synchronize()
be = out_be.cpu().numpy().reshape(n, 6, 64)
wire = np.ascontiguousarray(be[:, :, ::-1][:, :, :32]).reshape(-1)
d_b2 = torch.from_numpy(wire.copy()).cuda()
ws2 = dev.VarMsmWorkspace(n, 2)
for _ in range(3):
    ws2.run(d_b2, d_sc)
torch.cuda.synchronize()
print("done")
