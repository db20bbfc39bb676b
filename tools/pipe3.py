#!/usr/bin/env python3
"""Three-stage schedule (sort of MSM k+1 | accumulate of k | tail of k-1 on three streams) against the two-stage
pipeline, 2^20 G1.  usage: pipe3.py [reps]"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev, lib as ozk
L = ozk.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = 1 << 20
bases = dev.gen_g1_bases(n, seed=2)
sc = np.random.default_rng(10).integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
d_sc = torch.from_numpy(sc.reshape(-1)).cuda()
p = lambda t: ctypes.c_void_p(t.data_ptr())
sb, swb, awb = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
ozk.check(L.ozk_var_msm_stage_bytes(n, 1, ctypes.byref(sb), ctypes.byref(swb), ctypes.byref(awb)))
tb = int(L.ozk_var_msm_tail_bytes(n, 1))
buf = lambda b: torch.empty(b, dtype=torch.uint8, device="cuda")
sorted_ = [buf(sb.value) for _ in range(2)]
sort_ws, accum_ws = buf(swb.value), buf(awb.value)
tails = [buf(tb) for _ in range(2)]
outs = [torch.zeros(192, dtype=torch.uint8, device="cuda") for _ in range(2)]
S, A, T = torch.cuda.Stream(), torch.cuda.current_stream(), torch.cuda.Stream()
ev = lambda: [torch.cuda.Event() for _ in range(2)]
sort_done, accum_done, tail_done = ev(), ev(), ev()
cnt = [0]
def submit():
    k = cnt[0]; s = k % 2
    if k >= 2:
        S.wait_event(accum_done[s])          # sorted set s free again
    ozk.check(L.ozk_var_msm_sort_dev(p(bases), p(d_sc), n, 1, p(sorted_[s]), sb.value, p(sort_ws), swb.value, ctypes.c_void_p(S.cuda_stream)))
    sort_done[s].record(S)
    A.wait_event(sort_done[s])
    if k >= 2:
        A.wait_event(tail_done[s])           # tail buffer s free again
    ozk.check(L.ozk_var_msm_accum_dev(n, 1, p(sorted_[s]), sb.value, p(accum_ws), awb.value, p(tails[s]), tb, ctypes.c_void_p(A.cuda_stream)))
    accum_done[s].record(A)
    T.wait_event(accum_done[s])
    ozk.check(L.ozk_var_msm_tail_dev(n, 1, p(tails[s]), tb, p(outs[s]), ctypes.c_void_p(T.cuda_stream)))
    tail_done[s].record(T)
    cnt[0] += 1
for _ in range(6): submit()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): submit()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("three-stage: %.1f Mscalar-mul/s (%.3f ms per MSM)" % (reps * n / dt / 1e6, dt / reps * 1e3), flush=True)
r3 = bytes(outs[(cnt[0] - 1) % 2].cpu().numpy())
pipe = dev.VarMsmPipeline(n, 1, depth=2)
t = None
for _ in range(6): t = pipe.submit(bases, d_sc)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): t = pipe.submit(bases, d_sc)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("two-stage:   %.1f Mscalar-mul/s (%.3f ms per MSM)" % (reps * n / dt / 1e6, dt / reps * 1e3), flush=True)
assert bytes(pipe.result(t).cpu().numpy()) == r3
