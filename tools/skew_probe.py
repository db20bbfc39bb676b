"""Reference-profiler-shaped inputs (VariableBaseMSMProfiling.java:19-31): ONE base repeated n times,
scalars = 64-bit values or r - 64-bit values (Fp.random).  Checks the result and times the MSM."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev  # noqa: E402
from oracle import bn254 as o  # noqa: E402


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = 1 << logn
    rng = np.random.default_rng(10)
    lows = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
    neg = rng.integers(0, 2, size=n).astype(bool)
    vals = [(o.R - int(v)) if ng else int(v) for v, ng in zip(lows, neg)]
    sc = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint8).copy()
    base = o.G1.to_affine(o.G1.mul(o.G1.one, 987654321))
    bases = np.frombuffer(o.g1_to_wire(base) * n, dtype=np.uint8).copy()
    d_b = torch.from_numpy(bases).cuda()
    d_s = torch.from_numpy(sc).cuda()
    ws = dev.VarMsmWorkspace(n, 1)
    out = ws.run(d_b, d_s)
    torch.cuda.synchronize()
    want = o.g1_out_le(o.G1.to_affine(o.G1.mul(base, sum(vals) % o.R)))
    ok = bytes(out.cpu().numpy()) == want
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ws.run(d_b, d_s)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("skewed (one base, Fp.random scalars) n=2^%d ok=%s %.3f ms %.1f Mscalar-mul/s" % (logn, ok, ms, n / ms / 1e3))


if __name__ == "__main__":
    main()
