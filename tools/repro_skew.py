import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev, lib as ozk
from oracle import bn254 as o
L = ozk.load()
# pollute the caching allocator with garbage
g = torch.randint(0, 255, (3 << 30,), dtype=torch.uint8, device="cuda"); del g
n = 1 << 20
rng = np.random.default_rng(10)
lows = rng.integers(0, 1 << 63, size=n, dtype=np.uint64); neg = rng.integers(0, 2, size=n).astype(bool)
vals = [(o.R - int(v)) if ng else int(v) for v, ng in zip(lows, neg)]
sc = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint8).copy()
base = o.G1.to_affine(o.G1.mul(o.G1.one, 987654321))
bases = np.frombuffer(o.g1_to_wire(base) * n, dtype=np.uint8).copy()
d_b = torch.from_numpy(bases).cuda(); d_s = torch.from_numpy(sc).cuda()
a, b, c = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
ozk.check(L.ozk_var_msm_stage_bytes(n, 1, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
tb = int(L.ozk_var_msm_tail_bytes(n, 1))
mk = lambda k: torch.randint(0, 255, (k,), dtype=torch.uint8, device="cuda")
sorted_, sws, aws, tail, out = mk(a.value), mk(b.value), mk(c.value), mk(tb), mk(192)
st = int(torch.cuda.current_stream().cuda_stream)
p = lambda t: int(t.data_ptr())
print("sort", flush=True)
ozk.check(L.ozk_var_msm_sort_dev(p(d_b), p(d_s), n, 1, p(sorted_), a.value, p(sws), b.value, st)); torch.cuda.synchronize()
print("accum", flush=True)
ozk.check(L.ozk_var_msm_accum_dev(n, 1, p(sorted_), a.value, p(aws), c.value, p(tail), tb, st)); torch.cuda.synchronize()
print("tail", flush=True)
ozk.check(L.ozk_var_msm_tail_dev(n, 1, p(tail), tb, p(out), st)); torch.cuda.synchronize()
want = o.g1_out_le(o.G1.to_affine(o.G1.mul(base, sum(vals) % o.R)))
print("ok", bytes(out.cpu().numpy()) == want)
