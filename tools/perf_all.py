"""Device-resident timing of every hot-path entry point (HIP events, inputs resident in HBM),
with a correctness spot check each.  Output is committed under profiles/."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev  # noqa: E402
from octopuszk_amd import lib as ozk  # noqa: E402
from oracle import bn254 as o  # noqa: E402


def ptr(t):
    return int(t.data_ptr())


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def scalars(n, seed):
    rng = np.random.default_rng(seed)
    b = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    b[:, 31] &= 0x1F
    return b


def main():
    L = ozk.load()
    st = int(torch.cuda.current_stream().cuda_stream)
    print("device:", torch.cuda.get_device_name(0), flush=True)
    # ---- VarMSM G1
    for logn in (16, 20, 22):
        n = 1 << logn
        bases = dev.gen_g1_bases(n, seed=2)
        sc = scalars(n, 1)
        d_sc = torch.from_numpy(sc.reshape(-1)).cuda()
        ws = dev.VarMsmWorkspace(n, 1)
        ms = timeit(lambda: ws.run(bases, d_sc), 5)
        ks = dev.gen_base_logs(n, 2) if logn <= 20 else None
        ok = "-"
        if ks is not None:
            acc = sum(int.from_bytes(sc[i].tobytes(), "little") * ks[i] for i in range(n)) % o.R
            ok = bytes(ws.out.cpu().numpy()) == o.g1_out_le(o.G1.to_affine(o.G1.mul(o.G1.one, acc)))
        print("VarMSM G1  n=2^%-2d %8.3f ms  %8.1f Mscalar-mul/s  ok=%s" % (logn, ms, n / ms / 1e3, ok), flush=True)
    # ---- G2 bases: fixed-base G2 generator pass gives k_i * G2 in BE wire-out; convert to wire-in on host (small n)
    for logn in (16, 18):
        n = 1 << logn
        sc = scalars(n, 5)
        d_sc = torch.from_numpy(sc.reshape(-1)).cuda()
        ksc = scalars(n, 6)
        ksc[:, 8:] = 0  # 64-bit multipliers
        d_k = torch.from_numpy(ksc.reshape(-1)).cuda()
        window = 16
        outerc = 4
        base2 = torch.from_numpy(np.frombuffer(o.g2_to_wire(o.G2.one), dtype=np.uint8).copy()).cuda()
        out_be = torch.empty(n * 384, dtype=torch.uint8, device="cuda")
        wsb = int(L.ozk_fixed_batch_msm_workspace_bytes(outerc, window, n, 2))
        wsf = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        ozk.check(L.ozk_fixed_batch_msm_dev(outerc, window, n, ptr(base2), ptr(d_k), 2, ptr(out_be), ptr(wsf), wsb, st))
        torch.cuda.synchronize()
        be = out_be.cpu().numpy().reshape(n, 6, 64)
        wire = np.ascontiguousarray(be[:, :, ::-1][:, :, :32]).reshape(-1)  # 64-B BE -> 32-B LE
        d_b2 = torch.from_numpy(wire.copy()).cuda()
        ws2 = dev.VarMsmWorkspace(n, 2)
        ms = timeit(lambda: ws2.run(d_b2, d_sc), 3)
        acc = 0
        for i in range(n):
            acc += int.from_bytes(sc[i].tobytes(), "little") * int.from_bytes(ksc[i].tobytes(), "little")
        ok = bytes(ws2.out.cpu().numpy()) == o.g2_out_le(o.G2.to_affine(o.G2.mul(o.G2.one, acc % o.R)))
        print("VarMSM G2  n=2^%-2d %8.3f ms  %8.1f Mscalar-mul/s  ok=%s" % (logn, ms, n / ms / 1e3, ok), flush=True)
        del wsf, out_be
    # ---- FixedBase G1 / G2, field mul
    n = 1 << 20
    sc = scalars(n, 4)
    d_sc = torch.from_numpy(sc.reshape(-1)).cuda()
    for bn, name, window in ((1, "G1", 17), (2, "G2", 17)):
        outerc = (254 + window - 1) // window
        C = o.G1 if bn == 1 else o.G2
        wire = o.g1_to_wire(C.one) if bn == 1 else o.g2_to_wire(C.one)
        base = torch.from_numpy(np.frombuffer(wire, dtype=np.uint8).copy()).cuda()
        per = 192 if bn == 1 else 384
        d_out = torch.empty(n * per, dtype=torch.uint8, device="cuda")
        wsb = int(L.ozk_fixed_batch_msm_workspace_bytes(outerc, window, n, bn))
        wsf = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        fn = lambda: ozk.check(L.ozk_fixed_batch_msm_dev(outerc, window, n, ptr(base), ptr(d_sc), bn, ptr(d_out), ptr(wsf), wsb, st))
        ms = timeit(fn, 3)
        k = int.from_bytes(sc[7].tobytes(), "little")
        got = bytes(d_out[7 * per:8 * per].cpu().numpy())
        want = (o.g1_out_be if bn == 1 else o.g2_out_be)(C.to_affine(C.mul(C.one, k)))
        print("FixedBase %s n=2^20 w=%d %8.3f ms  %8.1f Mscalar-mul/s  ok=%s" % (name, window, ms, n / ms / 1e3, got == want), flush=True)
        del wsf, d_out
    fin = torch.from_numpy(np.concatenate([sc.reshape(-1), sc[3]])).cuda()
    fout = torch.empty(n * 64, dtype=torch.uint8, device="cuda")
    ms = timeit(lambda: ozk.check(L.ozk_field_batch_mul_dev(ptr(fin), n, ptr(fout), st)), 10)
    x, m = int.from_bytes(sc[9].tobytes(), "little"), int.from_bytes(sc[3].tobytes(), "little")
    ok = bytes(fout[9 * 64:10 * 64].cpu().numpy()) == int(x * m % o.R).to_bytes(64, "big")
    print("FieldMul   n=2^20      %8.3f ms  %8.1f Melem/s  %6.1f GB/s (96 B/elem)  ok=%s" % (ms, n / ms / 1e3, n * 96 / ms / 1e6, ok), flush=True)
    # ---- FFT
    for logn in (16, 20, 22):
        n = 1 << logn
        a = scalars(n, 3)
        d_in = torch.from_numpy(a.reshape(-1)).cuda()
        d_out = torch.empty(n * 64, dtype=torch.uint8, device="cuda")
        wsb = int(L.ozk_fft_workspace_bytes(n))
        wsf = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        om = ctypes.create_string_buffer(o.to_le32(o.fr_root_of_unity(n)), 32)
        fn = lambda: ozk.check(L.ozk_fft_dev(ptr(d_in), n, ctypes.cast(om, ctypes.c_void_p), ptr(d_out), ptr(wsf), wsb, st))
        ms = timeit(fn, 10)
        out0 = int.from_bytes(bytes(d_out[:64].cpu().numpy()), "little")
        s = sum(int.from_bytes(r.tobytes(), "little") for r in a) % o.R
        print("FFT Fr     n=2^%-2d      %8.3f ms  %8.1f Melem/s  %6.1f GB/s (64 B/elem)  ok=%s" % (logn, ms, n / ms / 1e3, n * 64 / ms / 1e6, out0 == s), flush=True)
    # ---- host-buffer (JNI-shaped) variable-base MSM: per-call upload vs prepared bases
    import time
    n = 1 << 20
    d_b = dev.gen_g1_bases(n, seed=2)
    hb = bytes(d_b.cpu().numpy())
    hs = bytes(scalars(n, 1).reshape(-1))
    out = ctypes.create_string_buffer(192)
    vp = lambda b: ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p)
    def host_call():
        ozk.check(L.ozk_var_msm_host(vp(hb), vp(hs), n, 1, 0, ctypes.cast(out, ctypes.c_void_p)))
    host_call()
    t0 = time.perf_counter()
    for _ in range(5):
        host_call()
    per_call = (time.perf_counter() - t0) / 5 * 1e3
    ref = out.raw
    h = ctypes.c_void_p()
    t0 = time.perf_counter()
    ozk.check(L.ozk_bases_create_host(vp(hb), n, 1, 0, ctypes.byref(h)))
    create_ms = (time.perf_counter() - t0) * 1e3
    def prep_call():
        ozk.check(L.ozk_var_msm_bases_host(h, vp(hs), n, ctypes.cast(out, ctypes.c_void_p)))
    prep_call()
    t0 = time.perf_counter()
    for _ in range(5):
        prep_call()
    prep_ms = (time.perf_counter() - t0) / 5 * 1e3
    print("VarMSM G1 host buffers n=2^20: per-call upload %8.3f ms | prepared bases %8.3f ms (one-time prepare %.1f ms)  ok=%s"
          % (per_call, prep_ms, create_ms, out.raw == ref), flush=True)
    L.ozk_bases_destroy(h)
    # ---- the double MSM native (G1 and G2 over the same scalars), host buffers, 2^18
    n2 = 1 << 18
    b1h = bytes(dev.gen_g1_bases(n2, seed=5).cpu().numpy())
    G2w = np.frombuffer(o.g2_to_wire(o.G2.to_affine(o.G2.mul(o.G2.one, 424242))), dtype=np.uint8)
    b2h = bytes(np.tile(G2w, n2))
    s2h = bytes(scalars(n2, 6).reshape(-1))
    out2 = ctypes.create_string_buffer(576)
    def dbl_call():
        ozk.check(L.ozk_var_double_msm_host(vp(b1h), vp(b2h), vp(s2h), n2, 0, ctypes.cast(out2, ctypes.c_void_p)))
    dbl_call()
    t0 = time.perf_counter()
    for _ in range(3):
        dbl_call()
    dbl_ms = (time.perf_counter() - t0) / 3 * 1e3
    o1, o2 = ctypes.create_string_buffer(192), ctypes.create_string_buffer(384)
    t0 = time.perf_counter()
    for _ in range(3):
        ozk.check(L.ozk_var_msm_host(vp(b1h), vp(s2h), n2, 1, 0, ctypes.cast(o1, ctypes.c_void_p)))
        ozk.check(L.ozk_var_msm_host(vp(b2h), vp(s2h), n2, 2, 0, ctypes.cast(o2, ctypes.c_void_p)))
    seq_ms = (time.perf_counter() - t0) / 3 * 1e3
    print("VarMSM double (G1 || G2) host buffers n=2^18: %8.3f ms | two separate calls %8.3f ms  ok=%s"
          % (dbl_ms, seq_ms, out2.raw == o1.raw + o2.raw), flush=True)
    # ---- QAP witness map (7 transforms + pointwise stages), device-resident
    for logm in (16, 21):
        m = 1 << logm
        ev = [torch.from_numpy(scalars(m, 40 + k).reshape(-1)).cuda() for k in range(3)]
        d_h = torch.empty((m + 1) * 32, dtype=torch.uint8, device="cuda")
        wsb = int(L.ozk_qap_witness_workspace_bytes(m))
        wsf = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        om = ctypes.create_string_buffer(o.to_le32(o.fr_root_of_unity(m)), 32)
        gg = ctypes.create_string_buffer(o.to_le32(o.FR_MULT_GEN), 32)
        fn = lambda: ozk.check(L.ozk_qap_witness_dev(ptr(ev[0]), ptr(ev[1]), ptr(ev[2]), m, ctypes.cast(om, ctypes.c_void_p),
                                                     ctypes.cast(gg, ctypes.c_void_p), ptr(d_h), ptr(wsf), wsb, st))
        ms = timeit(fn, 10)
        # 3 x 32 B in + 32 B out per domain point
        print("QAP witness m=2^%-2d     %8.3f ms  %8.1f Mpoint/s  %6.1f GB/s (128 B/point)" % (logm, ms, m / ms / 1e3, m * 128 / ms / 1e6), flush=True)


if __name__ == "__main__":
    main()
