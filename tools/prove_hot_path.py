#!/usr/bin/env python3
"""The GPU part of one Groth16 prove at BASELINE.json cfg-5 scale (2^20 constraints), composed from the
library's device entry points — a measurement harness, not a prover (SURVEY.md §8f N1 needs the R1CS and
the key generator on top):

    witness map      R1CStoQAP.R1CStoQAPWitness (R1CStoQAP.java:163-230)       domain 2^21
    A, B1, deltaABC  three G1 variable-base MSMs over the key queries           2^20 pairs each
    H                one G1 MSM over the H query                                2^21 pairs
    B2               one G2 MSM                                                 2^20 pairs
(SerialProver.java:36-118).  Keys are synthetic (k_i G), prepared once (ozk_var_msm_prepare_dev), the G1
MSMs are issued through VarMsmPipeline (two in flight), the G2 MSM runs on its own stream.
"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev  # noqa: E402
from octopuszk_amd import lib as ozk  # noqa: E402
from oracle import bn254 as o  # noqa: E402

L = ozk.load()


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def scalars(n, seed):
    s = np.random.default_rng(seed).integers(0, 256, size=(n, 32), dtype=np.uint8)
    s[:, 31] &= 0x1F
    return torch.from_numpy(s.reshape(-1)).cuda()


def g2_bases(n):
    # k G2 for 64-bit k through the fixed-base path, converted to the var-MSM wire format
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ks = np.random.default_rng(5).integers(0, 256, size=(n, 32), dtype=np.uint8)
    ks[:, 8:] = 0
    base = torch.from_numpy(np.frombuffer(o.g2_to_wire(o.G2.one), dtype=np.uint8).copy()).cuda()
    out = torch.empty(n * 384, dtype=torch.uint8, device="cuda")
    wsb = int(L.ozk_fixed_batch_msm_workspace_bytes(4, 16, n, 2))
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    ozk.check(L.ozk_fixed_batch_msm_dev(4, 16, n, ptr(base), ptr(torch.from_numpy(ks.reshape(-1)).cuda()), 2, ptr(out),
                                        ptr(ws), wsb, st))
    torch.cuda.synchronize()
    be = out.cpu().numpy().reshape(n, 6, 64)
    return torch.from_numpy(np.ascontiguousarray(be[:, :, ::-1][:, :, :32]).reshape(-1).copy()).cuda()


def main():
    logc = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n, m = 1 << logc, 1 << (logc + 1)
    # ---- one-time: keys on the device, prepared
    p1, p2 = dev.VarMsmPipeline(n, 1, depth=2), dev.VarMsmPipeline(m, 1, depth=2)
    keys = {k: p1.prepare(dev.gen_g1_bases(n, seed=s)) for k, s in (("A", 31), ("B1", 32), ("deltaABC", 33))}
    keys["H"] = p2.prepare(dev.gen_g1_bases(m, seed=34))
    b2 = g2_bases(n)
    g2ws = dev.VarMsmWorkspace(n, 2)
    s_g2 = torch.cuda.Stream()
    # ---- per proof: witness-dependent inputs
    w = scalars(n, 41)
    ev = [scalars(m, 42 + k) for k in range(3)]
    d_h = torch.empty((m + 1) * 32, dtype=torch.uint8, device="cuda")
    qws_b = int(L.ozk_qap_witness_workspace_bytes(m))
    qws = torch.empty(qws_b, dtype=torch.uint8, device="cuda")
    om = ctypes.create_string_buffer(o.to_le32(o.fr_root_of_unity(m)), 32)
    gg = ctypes.create_string_buffer(o.to_le32(o.FR_MULT_GEN), 32)

    def prove():
        main_s = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(main_s)
        s_g2.wait_event(ready)
        with torch.cuda.stream(s_g2):            # the G2 MSM has nothing to wait for
            g2ws.run(b2, w)
        st = ctypes.c_void_p(main_s.cuda_stream)
        ozk.check(L.ozk_qap_witness_dev(ptr(ev[0]), ptr(ev[1]), ptr(ev[2]), m, ctypes.cast(om, ctypes.c_void_p),
                                        ctypes.cast(gg, ctypes.c_void_p), ptr(d_h), ptr(qws), qws_b, st))
        t = [p1.submit(keys[k], w, prepared=True) for k in ("A", "B1")]
        outs = [p1.result(t[0])]
        t.append(p1.submit(keys["deltaABC"], w, prepared=True))
        outs.append(p1.result(t[1]))
        th = p2.submit(keys["H"], d_h[:m * 32], prepared=True)   # coefficients of H are the scalars
        outs.append(p1.result(t[2]))
        outs.append(p2.result(th))
        main_s.wait_stream(s_g2)
        return outs

    for _ in range(2):
        prove()
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        prove()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print("Groth16 prove, GPU hot path at 2^%d constraints (witness map 2^%d, 3 x G1 MSM 2^%d, G1 MSM 2^%d, G2 MSM 2^%d): "
          "%.2f ms per proof, inputs resident in HBM" % (logc, logc + 1, logc, logc + 1, logc, ms), flush=True)


if __name__ == "__main__":
    main()
