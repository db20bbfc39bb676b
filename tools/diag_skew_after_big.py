"""Diagnostic: run a large MSM (2^24, plain-window plan), release it, then the profiler-shaped 2^20 MSM
stage by stage with a synchronisation and a progress line after every stage."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev, lib as ozk
from oracle import bn254 as o

L = ozk.load()
def ptr(t): return ctypes.c_void_p(t.data_ptr())
def log(*a):
    print("[%.1f]" % time.time(), *a, flush=True)

big = int(sys.argv[1]) if len(sys.argv) > 1 else 24
if big:
    n = 1 << big
    bases = dev.gen_g1_bases(n, seed=11)
    sc = np.random.default_rng(12).integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
    d_sc = torch.from_numpy(sc.reshape(-1)).cuda()
    ws = dev.VarMsmWorkspace(n, 1)
    out = ws.run(bases, d_sc); torch.cuda.synchronize()
    log("big MSM done")
    h = n // 2
    ws2 = dev.VarMsmWorkspace(h, 1)
    for k in range(2):
        ws2.run(bases[k * h * 96:(k + 1) * h * 96], d_sc[k * h * 32:(k + 1) * h * 32]); torch.cuda.synchronize()
    log("halves done")
    del ws, ws2, bases, d_sc, out, sc
n = 1 << 20
rng = np.random.default_rng(10)
lows = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
neg = rng.integers(0, 2, size=n).astype(bool)
vals = [(o.R - int(v)) if ng else int(v) for v, ng in zip(lows, neg)]
sc = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint8).copy()
base = o.G1.to_affine(o.G1.mul(o.G1.one, 987654321))
bases = np.frombuffer(o.g1_to_wire(base) * n, dtype=np.uint8).copy()
d_bases, d_scalars = torch.from_numpy(bases).cuda(), torch.from_numpy(sc).cuda()
log("inputs ready")
sb, swb, ab = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
ozk.check(L.ozk_var_msm_stage_bytes(n, 1, ctypes.byref(sb), ctypes.byref(swb), ctypes.byref(ab)))
tb = int(L.ozk_var_msm_tail_bytes(n, 1))
fill = int(os.environ.get("DIAG_FILL", "-1"))
def buf(nbytes):
    t = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    if fill >= 0: t.fill_(fill)
    return t
d_sorted, d_sortws, d_acc, d_tail = buf(sb.value), buf(swb.value), buf(ab.value), buf(tb)
d_out = torch.zeros(192, dtype=torch.uint8, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
reps = int(os.environ.get("DIAG_REPS", "3"))
refill = int(os.environ.get("DIAG_REFILL", "0"))
quiet = reps > 5
bad = []
for rep in range(reps):
    if refill and fill >= 0:
        for t in (d_sorted, d_sortws, d_acc, d_tail):
            t.fill_(fill)
        torch.cuda.synchronize()
    ozk.check(L.ozk_var_msm_sort_dev(ptr(d_bases), ptr(d_scalars), n, 1, ptr(d_sorted), sb.value, ptr(d_sortws), swb.value, st))
    torch.cuda.synchronize()
    if not quiet: log("rep", rep, "sort ok")
    ozk.check(L.ozk_var_msm_accum_dev(n, 1, ptr(d_sorted), sb.value, ptr(d_acc), ab.value, ptr(d_tail), tb, st))
    torch.cuda.synchronize()
    if not quiet: log("rep", rep, "accum ok")
    ozk.check(L.ozk_var_msm_tail_dev(n, 1, ptr(d_tail), tb, ptr(d_out), st))
    torch.cuda.synchronize()
    ok = bytes(d_out.cpu().numpy()) == o.g1_out_le(o.G1.to_affine(o.G1.mul(base, sum(vals) % o.R)))
    if not ok: bad.append(rep)
    if not quiet: log("rep", rep, "result ok =", ok)
log("reps", reps, "fill", fill, "refill", refill, "bad reps:", bad)
