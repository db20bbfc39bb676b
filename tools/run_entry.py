#!/usr/bin/env python3
"""Run ONE entry point a few times with inputs resident in HBM (for `rocprofv3 --kernel-trace --stats -- python3
tools/run_entry.py <what> [reps]`).  what: fft22 | fixed_g1 | fixed_g2 | fixed_g1_rebuild | fixed_g2_rebuild (window table rebuilt per call, the form of rounds 1-3) | var_g2 | qap21 | var_g1 | prove"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev  # noqa: E402
from octopuszk_amd import lib as ozk  # noqa: E402

L = ozk.load()
FR = 21888242871839275222246405745257275088548364400416034343698204186575808495617
FR_ROOT = 19103219067921713944291392827692070036145651957329286315305642004821462161904
G1_ONE = (1).to_bytes(32, "little") + (2).to_bytes(32, "little") + (1).to_bytes(32, "little")
G2_ONE = b"".join(v.to_bytes(32, "little") for v in (
    10857046999023057135944570762232829481370756359578518086990519993285655852781,
    11559732032986387107991004021392285783925812861821192530917403151452391805634,
    8495653923123431417604973247489272438418190587263600148770280649306958101930,
    4082367875863433681332203403145435568316851327593401208105741076214120093531, 1, 0))


def ptr(t):
    return int(t.data_ptr())


def scalars(n, seed, bits64=False):
    b = np.random.default_rng(seed).integers(0, 256, size=(n, 32), dtype=np.uint8)
    if bits64:
        b[:, 8:] = 0
    else:
        b[:, 31] &= 0x1F
    return torch.from_numpy(b.reshape(-1)).cuda()


def le32(v):
    return ctypes.create_string_buffer(int(v).to_bytes(32, "little"), 32)


def main():
    what = sys.argv[1]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    st = int(torch.cuda.current_stream().cuda_stream)
    if what == "fft22":
        n = 1 << 22
        d_in, d_out = scalars(n, 3), torch.empty(n * 64, dtype=torch.uint8, device="cuda")
        wsb = int(L.ozk_fft_workspace_bytes(n))
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        om = le32(pow(FR_ROOT, FR // n, FR))
        fn = lambda: ozk.check(L.ozk_fft_dev(ptr(d_in), n, ctypes.cast(om, ctypes.c_void_p), ptr(d_out), ptr(ws), wsb, st))
    elif what in ("fixed_g1", "fixed_g2", "fixed_g1_rebuild", "fixed_g2_rebuild"):
        n, bn = 1 << 20, 1 if "g1" in what else 2
        sc = scalars(n, 4)
        base_host = np.frombuffer(G1_ONE if bn == 1 else G2_ONE, dtype=np.uint8).copy()
        base = torch.from_numpy(base_host).cuda()
        d_out = torch.empty(n * (192 if bn == 1 else 384), dtype=torch.uint8, device="cuda")
        wsb = int(L.ozk_fixed_batch_msm_workspace_bytes(15, 17, n, bn))
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        if what.endswith("_rebuild"):   # base in device memory: the table is built inside every call
            fn = lambda: ozk.check(L.ozk_fixed_batch_msm_dev(15, 17, n, ptr(base), ptr(sc), bn, ptr(d_out), ptr(ws), wsb, st))
        else:                           # base as host bytes: the table comes from the per-device cache after the first call
            fn = lambda: ozk.check(L.ozk_fixed_batch_msm_base_dev(15, 17, n, base_host.ctypes.data_as(ctypes.c_void_p), ptr(sc), bn,
                                                                  ptr(d_out), 0, ptr(ws), wsb, st))
    elif what == "var_g2":
        n = 1 << 20
        base = torch.from_numpy(np.frombuffer(G2_ONE, dtype=np.uint8).copy()).cuda()
        bases = torch.empty(n * 192, dtype=torch.uint8, device="cuda")
        wsb = int(L.ozk_fixed_batch_msm_workspace_bytes(16, 16, n, 2))
        wsf = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        ozk.check(L.ozk_fixed_batch_msm_compact_dev(16, 16, n, ptr(base), ptr(scalars(n, 6, True)), 2, ptr(bases), ptr(wsf), wsb, st))
        torch.cuda.synchronize()
        del wsf
        sc = scalars(n, 5)
        ws = dev.VarMsmWorkspace(n, 2)
        fn = lambda: ws.run(bases, sc)
    elif what == "var_g1":
        n = 1 << 20
        bases, sc = dev.gen_g1_bases(n, seed=2), scalars(n, 1)
        ws = dev.VarMsmWorkspace(n, 1)
        fn = lambda: ws.run(bases, sc)
    elif what == "qap21":
        m = 1 << 21
        ev = [scalars(m, 20 + k) for k in range(3)]
        d_h = torch.empty((m + 1) * 32, dtype=torch.uint8, device="cuda")
        wsb = int(L.ozk_qap_witness_workspace_bytes(m))
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        om, g = le32(pow(FR_ROOT, FR // m, FR)), le32(5)
        fn = lambda: ozk.check(L.ozk_qap_witness_dev(ptr(ev[0]), ptr(ev[1]), ptr(ev[2]), m, ctypes.cast(om, ctypes.c_void_p),
                                                     ctypes.cast(g, ctypes.c_void_p), ptr(d_h), ptr(ws), wsb, st))
    elif what == "prove":
        from octopuszk_amd import zksnark as z
        logn = int(os.environ.get("OZK_PROVE_LOGN", "20"))
        r1cs, primary, auxiliary = z.serial_construct(1 << logn, 1023)
        crs = z.serial_setup_generate(r1cs)
        prover = z.SerialProver(crs.proving_key)
        fn = lambda: prover.prove(primary, auxiliary)
    else:
        raise SystemExit("unknown entry point " + what)
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%s: %.3f ms per call over %d calls" % (what, e0.elapsed_time(e1) / reps, reps), flush=True)


if __name__ == "__main__":
    main()
