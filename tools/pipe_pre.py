#!/usr/bin/env python3
"""Two-stage pipeline with the base conversion of MSM k+1 on a third stream during the accumulation of MSM k
(ozk_var_msm_prepare_dev into a double-buffered record array, then ozk_var_msm_head_prepared_dev), against the plain
two-stage pipeline.  2^20 G1.  usage: pipe_pre.py [reps]"""
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
ws_bytes = int(L.ozk_var_msm_head_workspace_bytes(n, 1)); tb = int(L.ozk_var_msm_tail_bytes(n, 1)); pb = int(L.ozk_var_msm_prepared_bytes(n, 1))
buf = lambda b: torch.empty(b, dtype=torch.uint8, device="cuda")
ws = buf(ws_bytes); tails = [buf(tb) for _ in range(2)]; aff = [buf(pb) for _ in range(2)]
outs = [torch.zeros(192, dtype=torch.uint8, device="cuda") for _ in range(2)]
main, pre, side = torch.cuda.current_stream(), torch.cuda.Stream(), torch.cuda.Stream()
ev = lambda: [torch.cuda.Event() for _ in range(2)]
pre_done, head_done, tail_done = ev(), ev(), ev()
order = []
for _ in range(2):
    e = ctypes.c_void_p(); ozk.check(L.ozk_order_event_create(ctypes.byref(e))); order.append(e)
cnt = [0]
def submit():
    k = cnt[0]; s = k % 2
    if k >= 2: pre.wait_event(head_done[s])       # record array s free again (its accumulation is done)
    ozk.check(L.ozk_var_msm_prepare_dev(p(bases), n, 1, p(aff[s]), pb, ctypes.c_void_p(pre.cuda_stream)))
    pre_done[s].record(pre)
    main.wait_event(pre_done[s])
    if k >= 2: main.wait_event(tail_done[s])
    prev = order[(k - 1) % 2] if k else None
    ozk.check(L.ozk_var_msm_head_prepared_dev(p(aff[s]), p(d_sc), n, 1, p(ws), ws_bytes, p(tails[s]), tb, ctypes.c_void_p(main.cuda_stream), prev))
    head_done[s].record(main)
    side.wait_event(head_done[s])
    ozk.check(L.ozk_var_msm_tail_ordered_dev(n, 1, p(tails[s]), tb, p(outs[s]), ctypes.c_void_p(side.cuda_stream), order[s]))
    tail_done[s].record(side)
    cnt[0] += 1
for _ in range(6): submit()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): submit()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("conversion ahead: %.1f Mscalar-mul/s (%.3f ms per MSM)" % (reps * n / dt / 1e6, dt / reps * 1e3), flush=True)
r3 = bytes(outs[(cnt[0] - 1) % 2].cpu().numpy())
pipe = dev.VarMsmPipeline(n, 1, depth=2)
t = None
for _ in range(6): t = pipe.submit(bases, d_sc)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): t = pipe.submit(bases, d_sc)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("two-stage:        %.1f Mscalar-mul/s (%.3f ms per MSM)" % (reps * n / dt / 1e6, dt / reps * 1e3), flush=True)
assert bytes(pipe.result(t).cpu().numpy()) == r3
