#!/usr/bin/env python3
"""Level-1 kernel time of a LONE 2^20 MSM for different digit distributions (device clock stamped by the kernel):
  uniform     every lane's chunk crosses ~one bucket boundary (run-end store + run start, divergent);
  all_equal   every scalar the same: one bucket per window and half scalar, NO run boundary inside a chunk, bases
              read in index order — the pure accumulation loop;
  two_values  two buckets per window.
What the run-boundary machinery and the random gather cost = uniform - all_equal."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev  # noqa: E402
from octopuszk_amd import lib as ozk  # noqa: E402

L = ozk.load()
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << logn
bases = dev.gen_g1_bases(n, seed=2)
rng = np.random.default_rng(1)


def scal(kind):
    if kind == "uniform":
        b = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        b[:, 31] &= 0x1F
    elif kind == "all_equal":
        one = rng.integers(0, 256, size=(1, 32), dtype=np.uint8)
        one[:, 31] &= 0x1F
        b = np.repeat(one, n, axis=0)
    else:
        two = rng.integers(0, 256, size=(2, 32), dtype=np.uint8)
        two[:, 31] &= 0x1F
        b = two[rng.integers(0, 2, size=n)]
    return torch.from_numpy(np.ascontiguousarray(b).reshape(-1)).pin_memory().cuda()


ws = dev.VarMsmWorkspace(n, 1)
for kind in ("uniform", "all_equal", "two_values", "uniform"):
    s = scal(kind)
    ws.run(bases, s)
    torch.cuda.synchronize()
    ozk.check(L.ozk_prof_enable(2))
    for _ in range(5):
        ws.run(bases, s)
        torch.cuda.synchronize()
    st = (ctypes.c_double * 4)()
    k = ctypes.c_int()
    ozk.check(L.ozk_prof_dominant_kernel_stats(st, ctypes.byref(k)))
    ozk.check(L.ozk_prof_enable(0))
    print("%-10s level 1 alone: mean %.3f median %.3f min %.3f max %.3f ms over %d launches" % (kind, st[0], st[1], st[2], st[3], k.value), flush=True)
