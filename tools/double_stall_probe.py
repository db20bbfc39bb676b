#!/usr/bin/env python3
"""Which earlier activity of the process makes ozk_var_double_msm_host (2^18) alternate between 6 and 20-30 ms calls
(tools/host_path.py shows it, a process that only makes *_host calls does not).
usage: double_stall_probe.py [ws] [handle] [torchwork] [reuse] [sleep]"""
import ctypes, gc, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev, lib as ozk
gc.disable()
L = ozk.load()
vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
flags = set(sys.argv[1:])
n = 1 << 20; m = n // 4
rng = np.random.default_rng(1)
sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
g1 = np.frombuffer(bytes(dev.gen_g1_bases(n, seed=2).cpu().numpy()), dtype=np.uint8)
g2 = np.frombuffer(bytes(dev.gen_g1_bases(2 * m, seed=5).cpu().numpy()), dtype=np.uint8)
out = np.zeros(576, dtype=np.uint8)
cp = lambda a: np.array(a, copy=True)
if "ws" in flags:
    d_b, d_s = torch.from_numpy(g1.copy()).cuda(), torch.from_numpy(sc.reshape(-1)).cuda()
    ws = dev.VarMsmWorkspace(n, 1)
    for _ in range(6): ws.run(d_b, d_s)
    torch.cuda.synchronize()
for _ in range(3):
    ozk.check(L.ozk_var_msm_host(vp(cp(g1)), vp(cp(sc)), n, 1, 0, vp(out)))
if "handle" in flags:
    h = ctypes.c_void_p()
    ozk.check(L.ozk_bases_create_host(vp(g1), n, 1, 0, ctypes.byref(h)))
    for _ in range(3): ozk.check(L.ozk_var_msm_bases_host(h, vp(cp(sc)), n, vp(out)))
    ozk.check(L.ozk_bases_destroy(h))
if "torchwork" in flags:
    a = torch.empty(m * 192, dtype=torch.uint8, device="cuda"); b = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
    a.zero_(); b.zero_(); torch.cuda.synchronize(); _ = a.cpu().numpy(); del a, b
if "realg2" in flags:   # the G2 bases as tools/host_path.py makes them: fixed-base batch on the torch stream, workspace freed after
    from oracle import bn254 as o
    ks = rng.integers(0, 256, size=(m, 32), dtype=np.uint8); ks[:, 8:] = 0
    st = int(torch.cuda.current_stream().cuda_stream)
    base2 = torch.from_numpy(np.frombuffer(o.g2_to_wire(o.G2.one), dtype=np.uint8).copy()).cuda()
    g2d = torch.empty(m * 192, dtype=torch.uint8, device="cuda")
    wsb = int(L.ozk_fixed_batch_msm_workspace_bytes(16, 16, m, 2))
    wsf = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    ozk.check(L.ozk_fixed_batch_msm_compact_dev(16, 16, m, int(base2.data_ptr()), int(torch.from_numpy(ks.reshape(-1)).cuda().data_ptr()),
                                                2, int(g2d.data_ptr()), int(wsf.data_ptr()), wsb, st))
    torch.cuda.synchronize()
    g2 = g2d.cpu().numpy()
    del wsf, g2d
fixed = (cp(g1[:m * 96]), cp(g2), cp(sc[:m]))
ts = []
for i in range(9):
    args = fixed if "reuse" in flags else (cp(g1[:m * 96]), cp(g2), cp(sc[:m]))
    if "sleep" in flags: time.sleep(0.03)
    t0 = time.perf_counter()
    ozk.check(L.ozk_var_double_msm_host(vp(args[0]), vp(args[1]), vp(args[2]), m, 0, vp(out)))
    ts.append((time.perf_counter() - t0) * 1e3)
print("%-28s %s" % (" ".join(sorted(flags)) or "(plain)", " ".join("%.1f" % t for t in ts)), flush=True)
