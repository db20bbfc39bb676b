#!/usr/bin/env python3
"""One process, N repetitions of the scenario of tests/test_sharded_gpu.py::test_in_process_sharded_entry_equals_single_call
[1-5001-3] (which failed once in a full run of the round and never again): the in-process sharded entry (three host
threads, three concurrent MSMs on one device) and the single call, each compared with the oracle's bytes; also the same
slices issued by three Python threads through ozk_var_msm_host.  Prints every mismatch with the side it is on."""
import ctypes, os, sys, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import lib
from oracle import bn254 as o
L = lib.load()
_a = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(_a[0]) if _a else 300
n, shards = 5001, 3
rng = np.random.default_rng(n + shards)
G = o.G1
pts = [G.to_affine(G.mul(G.one, int(k))) for k in rng.integers(1, 1 << 62, size=16)]
pts[3] = G.zero
bw = np.frombuffer(b"".join(o.g1_to_wire(pts[i % 16]) for i in range(n)), dtype=np.uint8)
sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
sc[:, 31] &= 0x1F
vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
scal = [int.from_bytes(sc[i].tobytes(), "little") for i in range(n)]
want = o.g1_out_le(G.to_affine(o.pippenger_msm(G, scal, [pts[i % 16] for i in range(n)])))
base_n, rem = n // shards, n % shards
cuts = []
for i in range(shards):
    lo = i * base_n + min(i, rem); cnt = base_n + (1 if i < rem else 0); cuts.append((lo, cnt))
want_part = [o.g1_out_le(G.to_affine(o.pippenger_msm(G, scal[lo:lo + cnt], [pts[i % 16] for i in range(lo, lo + cnt)]))) for lo, cnt in cuts]
bad = {"sharded": 0, "single": 0, "threads": 0}
# other work between the iterations, so that the recycled contexts' arenas hold varied stale data (other sizes, G2)
g2pts = [o.G2.to_affine(o.G2.mul(o.G2.one, int(k))) for k in rng.integers(1, 1 << 62, size=4)]
junk = []
for m_, t_ in ((4, 1), (301, 2), (4099, 1), (70000, 1), (900, 2), (33000, 1)):
    b_ = np.frombuffer(b"".join((o.g1_to_wire(pts[(i * 7) % 16]) if t_ == 1 else o.g2_to_wire(g2pts[i % 4])) for i in range(m_)), dtype=np.uint8)
    s_ = rng.integers(0, 256, size=(m_, 32), dtype=np.uint8); s_[:, 31] &= 0x1F
    junk.append((m_, t_, b_, s_))
def scramble(it):
    outs = [np.zeros(384, dtype=np.uint8) for _ in range(3)]
    def w(j):
        m_, t_, b_, s_ = junk[(it * 3 + j) % len(junk)]
        lib.check(L.ozk_var_msm_host(vp(b_), vp(s_), m_, t_, j, vp(outs[j])))
    th = [threading.Thread(target=w, args=(j,)) for j in range(1 + it % 3)]
    [t.start() for t in th]; [t.join() for t in th]
for it in range(N):
    if "--scramble" in sys.argv: scramble(it)
    got, one = np.zeros(192, dtype=np.uint8), np.zeros(192, dtype=np.uint8)
    lib.check(L.ozk_var_msm_sharded_host(vp(bw), vp(sc), n, 1, shards, vp(got)))
    lib.check(L.ozk_var_msm_host(vp(bw), vp(sc), n, 1, 0, vp(one)))
    if bytes(got) != want:
        bad["sharded"] += 1; print("iteration %d: SHARDED differs" % it, flush=True)
    if bytes(one) != want:
        bad["single"] += 1; print("iteration %d: SINGLE differs" % it, flush=True)
    parts = [np.zeros(192, dtype=np.uint8) for _ in cuts]
    def work(i):
        lo, cnt = cuts[i]
        b, s = bw[lo * 96:(lo + cnt) * 96], sc[lo:lo + cnt]
        lib.check(L.ozk_var_msm_host(vp(np.ascontiguousarray(b)), vp(np.ascontiguousarray(s)), cnt, 1, i, vp(parts[i])))
    th = [threading.Thread(target=work, args=(i,)) for i in range(shards)]
    [t.start() for t in th]; [t.join() for t in th]
    for i in range(shards):
        if bytes(parts[i]) != want_part[i]:
            bad["threads"] += 1; print("iteration %d: concurrent slice %d differs" % (it, i), flush=True)
    if it % 4 == 3 and "--keep-contexts" not in sys.argv:   # contexts re-created now and then: their first use is part of the scenario
        lib.check(L.ozk_host_cache_release())
print("iterations %d, mismatches %s" % (N, bad))
