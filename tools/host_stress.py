#!/usr/bin/env python3
"""Concurrent host entry points at sizes that take the sliced / ranged paths: T threads each issue a mix of
ozk_var_msm_host (2^19 pairs: two slices), ozk_fixed_batch_msm_compact_host (2^17 scalars: ranges) and
ozk_fft_compact_host (2^18), results compared with the same calls made alone.  usage: host_stress.py [threads=4] [rounds=6]"""
import ctypes, os, sys, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev, lib as ozk
from oracle import bn254 as o
L = ozk.load()
vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
R = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = 1 << 19
rng = np.random.default_rng(7)
g1 = np.frombuffer(bytes(dev.gen_g1_bases(n, seed=2).cpu().numpy()), dtype=np.uint8)
sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
bw = np.frombuffer(o.g1_to_wire(o.G1.one), dtype=np.uint8)
nf = 1 << 17
nt = 1 << 18
om = np.frombuffer(o.to_le32(o.fr_root_of_unity(nt)), dtype=np.uint8)

def var(i):
    out = np.zeros(192, dtype=np.uint8)
    m = n - 1000 * i
    ozk.check(L.ozk_var_msm_host(vp(g1), vp(sc), m, 1, 0, vp(out)))
    return out.tobytes()
def fixed(i):
    out = np.zeros(nf * 96, dtype=np.uint8)
    s = np.ascontiguousarray(sc[i * 100:i * 100 + nf])
    ozk.check(L.ozk_fixed_batch_msm_compact_host(15, 17, nf, vp(bw), vp(s), 1, 0, vp(out)))
    return out.tobytes()
def fft(i):
    out = np.zeros(nt * 32, dtype=np.uint8)
    a = np.ascontiguousarray(sc[i * 10:i * 10 + nt])
    ozk.check(L.ozk_fft_compact_host(vp(a), nt, vp(om), 0, vp(out)))
    return out.tobytes()
jobs = [(f, i) for i in range(3) for f in (var, fixed, fft)]
want = {(f.__name__, i): f(i) for f, i in jobs}
bad = []
def worker(tid):
    for r in range(R):
        for k, (f, i) in enumerate(jobs):
            f2, i2 = jobs[(k + tid + r) % len(jobs)]
            if f2(i2) != want[(f2.__name__, i2)]:
                bad.append((tid, r, f2.__name__, i2))
ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
[t.start() for t in ths]; [t.join() for t in ths]
print("threads %d rounds %d calls %d mismatches %d %s" % (T, R, T * R * len(jobs), len(bad), bad[:5]))
sys.exit(1 if bad else 0)
