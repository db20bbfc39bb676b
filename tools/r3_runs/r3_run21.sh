#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -60
import ctypes, os, sys, time
import numpy as np
os.environ["OZK_HOST_TRACE"] = "1"
sys.path.insert(0, os.getcwd())
from octopuszk_amd import device as dev, lib as ozk
from oracle import bn254 as o
import gc; gc.disable()
L = ozk.load()
vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
n = 1 << 18
rng = np.random.default_rng(1)
sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
g1 = np.frombuffer(bytes(dev.gen_g1_bases(n, seed=2).cpu().numpy()), dtype=np.uint8)
pts = [np.frombuffer(o.g2_to_wire(o.G2.to_affine(o.G2.mul(o.G2.one, int(k)))), dtype=np.uint8) for k in rng.integers(1, 1 << 62, size=64)]
g2 = np.ascontiguousarray(np.stack(pts)[rng.integers(0, 64, size=n)]).reshape(-1)
out = np.zeros(576, dtype=np.uint8)
for i in range(10):
    a, b, s = np.array(g1, copy=True), np.array(g2, copy=True), np.array(sc, copy=True)
    t0 = time.perf_counter()
    ozk.check(L.ozk_var_double_msm_host(vp(a), vp(b), vp(s), n, 0, vp(out)))
    ms = (time.perf_counter() - t0) * 1e3
    st = (ctypes.c_double * 10)(); L.ozk_host_call_stats(st)
    print("call %d: %.2f ms  stage_wait %.2f (%d waits) memcpy_in %.2f sync %.2f" % (i, ms, st[2], int(st[7]), st[3], st[6]), flush=True)
PY
