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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
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
for it in range(N):
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
    if it % 4 == 3:   # contexts re-created now and then: their first use is part of the scenario
        lib.check(L.ozk_host_cache_release())
print("iterations %d, mismatches %s" % (N, bad))
