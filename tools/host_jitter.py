#!/usr/bin/env python3
"""Call-time distribution of the JNI-shaped host entry points, with the library's own account of where each call's
wall time went (ozk_host_call_stats: context acquire, arena growth, waits for a pinned staging buffer, host memcpys
into / out of the ring, enqueueing, stream synchronisation).

    host_jitter.py [N=200] [--interleaved] [--fresh-out]

plain:        N calls of each entry point in turn, fresh pageable INPUT buffers per call.
--interleaved the entry points alternate, with torch GPU work and large numpy allocations in between (what a JVM with
              other task threads and a garbage collector looks like to the library: round 2 saw single calls of
              10-38 ms only in this form).
--fresh-out   the OUTPUT buffer of every call is a new, never-touched allocation too (np.empty: the kernel maps and
              zeroes its pages on first write, inside the call's download memcpy).
Prints min / median / p90 / p99 / max per entry point, and the breakdown of the five slowest calls."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev, lib as ozk  # noqa: E402
from oracle import bn254 as o  # noqa: E402

L = ozk.load()
vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if args else 200
if "--no-gc" in sys.argv:   # is a slow call the Python collector's pause on the caller's side of the boundary?
    import gc
    gc.disable()
INTERLEAVED = "--interleaved" in sys.argv
FRESH_OUT = "--fresh-out" in sys.argv
FIELDS = ("acquire", "reserve", "stage_wait", "memcpy_in", "memcpy_out", "enqueue", "sync")

n = 1 << 20
rng = np.random.default_rng(1)
sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
sc[:, 31] &= 0x1F
g1 = np.frombuffer(bytes(dev.gen_g1_bases(n, seed=2).cpu().numpy()), dtype=np.uint8)
bw = np.frombuffer(o.g1_to_wire(o.G1.one), dtype=np.uint8)
out_small = np.zeros(576, dtype=np.uint8)
out_fixed = np.zeros(n * 192, dtype=np.uint8)
cp = lambda a: np.array(a, copy=True)


def call_var():
    b, s = cp(g1), cp(sc)
    t0 = time.perf_counter()
    ozk.check(L.ozk_var_msm_host(vp(b), vp(s), n, 1, 0, vp(out_small)))
    return (time.perf_counter() - t0) * 1e3


def call_fixed():
    s = cp(sc)
    dst = np.empty(n * 192, dtype=np.uint8) if FRESH_OUT else out_fixed
    t0 = time.perf_counter()
    ozk.check(L.ozk_fixed_batch_msm_host(15, 17, 15, 1 << 17, n, 254, vp(bw), vp(s), 1, 0, vp(dst)))
    return (time.perf_counter() - t0) * 1e3


def call_fft():
    m = 1 << 21
    s = cp(sc.reshape(-1)[:m * 32])
    dst = np.empty(m * 64, dtype=np.uint8) if FRESH_OUT else out_fixed[:m * 64]
    om = np.frombuffer(int(pow(19103219067921713944291392827692070036145651957329286315305642004821462161904,
                               21888242871839275222246405745257275088548364400416034343698204186575808495617 // m,
                               21888242871839275222246405745257275088548364400416034343698204186575808495617)).to_bytes(32, "little"),
                       dtype=np.uint8).copy()
    t0 = time.perf_counter()
    ozk.check(L.ozk_fft_host(vp(s), m, vp(om), 0, vp(dst)))
    return (time.perf_counter() - t0) * 1e3


def stats():
    st = (ctypes.c_double * 10)()
    ozk.check(L.ozk_host_call_stats(st))
    return list(st)


def noise(k):
    """what else a prover process does between native calls"""
    a = torch.empty(1 << 24, dtype=torch.float32, device="cuda").normal_()
    b = (a * a).sum()
    junk = np.ones((48 + 16 * (k % 5)) << 20, dtype=np.uint8)       # a large allocation, touched, then dropped
    _ = float(b)
    del junk, a


calls = {"ozk_var_msm_host G1 2^20": call_var, "ozk_fixed_batch_msm_host G1 2^20": call_fixed,
         "ozk_fft_host 2^21": call_fft}
rows = {k: [] for k in calls}
for f in calls.values():   # cold calls (context creation, arena growth) are not part of the distribution
    f()
if INTERLEAVED:
    names = list(calls)
    for i in range(N * len(names)):
        nm = names[i % len(names)]
        ms = calls[nm]()
        rows[nm].append((ms, stats()))
        noise(i)
else:
    for nm, f in calls.items():
        for _ in range(N):
            ms = f()
            rows[nm].append((ms, stats()))
print("mode: %s%s, %d calls per entry point" % ("interleaved" if INTERLEAVED else "plain", ", fresh output buffers" if FRESH_OUT else "", N))
for nm, r in rows.items():
    ts = sorted(x[0] for x in r)
    q = lambda p: ts[min(len(ts) - 1, int(len(ts) * p))]
    print("%-34s min %.2f median %.2f p90 %.2f p99 %.2f max %.2f ms  (> 2x median: %d of %d)"
          % (nm, ts[0], q(0.5), q(0.9), q(0.99), ts[-1], sum(t > 2 * q(0.5) for t in ts), len(ts)), flush=True)
    med = sorted(r, key=lambda x: x[0])[len(r) // 2]
    print("    median call : " + "  ".join("%s %.2f" % (f, v) for f, v in zip(FIELDS, med[1])) + "  (stage waits %d, memcpys %d)" % (med[1][7], med[1][8]))
    for ms, st in sorted(r, key=lambda x: -x[0])[:5]:
        print("    %8.2f ms (inside the library %.2f) : " % (ms, st[9]) + "  ".join("%s %.2f" % (f, v) for f, v in zip(FIELDS, st)))
