"""GPU parity tests of FixedBaseMSM / field batch multiplication through the C ABI."""
import random

import pytest

from oracle import bn254 as o
import golden_util as gu

pytestmark = pytest.mark.gpu


def test_fixed_base_golden():
    from octopuszk_amd import fixed_base_msm as fb
    for case in gu.load("fixed_base.json")["cases"]:
        if case["curve"] in ("G1", "G2"):
            sc = bytes.fromhex(case["scalars"])
            n = len(sc) // 32
            w = case["window"]
            num_windows = case["outerc"]
            got = fb.batch_msm_native_helper(case["outerc"], w, num_windows, 1 << w, n, case["scalar_size"],
                                             bytes.fromhex(case["base"]), sc, 1 if case["curve"] == "G1" else 2, 0)
            assert got == bytes.fromhex(case["expected_out"]), (case["curve"], w)
        else:
            got = fb.field_batch_msm_native_helper(bytes.fromhex(case["field_mul_in"]), case["n"], 0)
            assert got == bytes.fromhex(case["expected_out"])


@pytest.mark.parametrize("n,window", [(1, 1), (5, 3), (300, 8), (1000, 11)])
def test_fixed_base_g1_vs_oracle(n, window):
    from octopuszk_amd import fixed_base_msm as fb
    rng = random.Random(n + window)
    base = o.G1.mul(o.G1.one, rng.randrange(1, o.R))          # Jacobian base, Z != 1
    scalars = [rng.randrange(o.R) for _ in range(n)]
    scalars[0] = 0
    got = fb.batch_msm(254, window, base, scalars, is_g1=True)
    for s, P in zip(scalars, got):
        assert P == o.G1.to_affine(o.fixed_base_mul(o.G1, base, 254, window, s))


def test_fixed_base_glv_edge_scalars():
    # the gather-add splits every scalar with the GLV endomorphism (msm_fixed.hip k_fb_main_glv)
    from octopuszk_amd import fixed_base_msm as fb
    lam = 4407920970296243842393367215006156084916469457145843978461
    scalars = [0, 1, 2, o.R - 1, o.R - 2, lam, lam - 1, lam + 1, o.R - lam, (1 << 127) - 1, 1 << 127, (1 << 127) + 1,
               1 << 128, 1 << 253, o.R // 2, 9931322734385697763, 147946756881789319010696353538189108491]
    for C, is_g1 in ((o.G1, True), (o.G2, False)):
        base = C.mul(C.one, 0x1234567)
        got = fb.batch_msm(254, 17, base, scalars, is_g1=is_g1)
        for s_, P in zip(scalars, got):
            assert P == C.to_affine(C.mul(base, s_)), hex(s_)


def test_fixed_base_infinity_base_and_small_orders():
    # every table entry is the point at infinity: the affine table (msm_fixed.hip k_fb_table_affine) must carry
    # the (0, 0) marker through the shared inversion; and scalars whose halves hit one window only
    from octopuszk_amd import fixed_base_msm as fb
    scalars = [0, 1, 5, o.R - 1, 1 << 130, (1 << 17) - 1, 1 << 17]
    for C, is_g1 in ((o.G1, True), (o.G2, False)):
        got = fb.batch_msm(254, 17, C.zero, scalars, is_g1=is_g1)
        for P in got:
            assert P == C.to_affine(C.zero)
        base = C.mul(C.one, 3)
        got = fb.batch_msm(254, 6, base, scalars, is_g1=is_g1)    # 22 windows of 6 bits per half
        for s_, P in zip(scalars, got):
            assert P == C.to_affine(C.mul(base, s_)), hex(s_)


def test_fixed_base_truncates_to_outerc_windows():
    # only the first outerc windows of the scalar are used (FixedBaseMSM.java:146-164)
    from octopuszk_amd import fixed_base_msm as fb
    base = o.G1.to_affine(o.G1.mul(o.G1.one, 5))
    scalars = [(1 << 200) + 77, 12345]
    got = fb.batch_msm(64, 8, base, scalars, is_g1=True)     # scalarSize 64 -> 8 windows of 8 bits
    assert got[0] == o.G1.to_affine(o.G1.mul(base, 77))
    assert got[1] == o.G1.to_affine(o.G1.mul(base, 12345))


def test_fixed_base_g2_and_double():
    from octopuszk_amd import fixed_base_msm as fb
    from octopuszk_amd.variable_base_msm import marshal_scalars
    rng = random.Random(77)
    b1 = o.G1.to_affine(o.G1.mul(o.G1.one, 31337))
    b2 = o.G2.mul(o.G2.one, 424242)
    scalars = [0, 1, o.R - 1] + [rng.randrange(o.R) for _ in range(20)]
    got2 = fb.batch_msm(254, 5, b2, scalars, is_g1=False)
    for s, P in zip(scalars, got2):
        assert P == o.G2.to_affine(o.fixed_base_mul(o.G2, b2, 254, 5, s))
    w1, w2 = 6, 4
    oc1, oc2 = (254 + w1 - 1) // w1, (254 + w2 - 1) // w2
    raw = fb.double_batch_msm_native_helper(oc1, w1, oc2, w2, oc1, 1 << w1, oc2, 1 << w2, len(scalars),
                                            o.g1_to_wire(b1), o.g2_to_wire(b2), marshal_scalars(scalars), 0)
    for i, s in enumerate(scalars):
        rec = raw[576 * i: 576 * (i + 1)]
        assert rec[:192] == o.g1_out_be(o.G1.to_affine(o.G1.mul(b1, s)))
        assert rec[192:] == o.g2_out_be(o.G2.to_affine(o.G2.mul(b2, s)))


def test_field_batch_mul():
    from octopuszk_amd import fixed_base_msm as fb
    rng = random.Random(3)
    xs = [0, 1, o.R - 1] + [rng.randrange(o.R) for _ in range(500)]
    m = rng.randrange(o.R)
    got = fb.field_batch_msm_native_helper(b"".join(o.to_le32(x) for x in xs + [m]), len(xs), 0)
    assert got == b"".join(int(x * m % o.R).to_bytes(64, "big") for x in xs)


@pytest.mark.parametrize("bn,compact", [(1, 0), (1, 1), (2, 0), (2, 1)])
def test_fixed_base_host_ranges_equal_one_range(bn, compact, monkeypatch):
    """The host entry points compute the per-scalar part in ranges and download range by range behind the
    computation (OZK_FB_HOST_RANGES): the bytes do not depend on the number of ranges, and sampled elements
    equal the oracle's."""
    import ctypes
    import numpy as np
    from octopuszk_amd import lib
    L = lib.load()
    n, w, outerc = 40000 + 7 * bn, 11, 24
    rng = np.random.default_rng(bn * 2 + compact)
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x1F
    G = o.G1 if bn == 1 else o.G2
    base = G.mul(G.one, 0x1234567 + compact)
    bw = np.frombuffer(o.g1_to_wire(base) if bn == 1 else o.g2_to_wire(base), dtype=np.uint8)
    per = (192 if bn == 1 else 384) // (2 if compact else 1)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    outs = {}
    try:
        for k in (1, 4, 9):
            monkeypatch.setenv("OZK_FB_HOST_RANGES", str(k))
            lib.check(L.ozk_tuning_reload())
            out = np.zeros(n * per, dtype=np.uint8)
            if compact:
                lib.check(L.ozk_fixed_batch_msm_compact_host(outerc, w, n, vp(bw), vp(sc), bn, 0, vp(out)))
            else:
                lib.check(L.ozk_fixed_batch_msm_host(outerc, w, outerc, 1 << w, n, 254, vp(bw), vp(sc), bn, 0, vp(out)))
            outs[k] = out.tobytes()
    finally:
        monkeypatch.delenv("OZK_FB_HOST_RANGES", raising=False)
        lib.check(L.ozk_tuning_reload())
    assert outs[1] == outs[4] == outs[9]
    cw = 32 if compact else 64
    for i in (0, n // 2 + 1, n - 1):
        s = int.from_bytes(sc[i].tobytes(), "little") % o.R
        want = G.to_affine(G.mul(base, s))
        rec = outs[4][i * per:(i + 1) * per]
        if bn == 1:
            got = [int.from_bytes(rec[j * cw:(j + 1) * cw], "little" if compact else "big") for j in range(2)]
            assert (got[0], got[1]) == (want[0], want[1]), i
        else:
            got = [int.from_bytes(rec[j * cw:(j + 1) * cw], "little" if compact else "big") for j in range(4)]
            assert ((got[0], got[1]), (got[2], got[3])) == (want[0], want[1]), i


def test_fixed_base_table_cache_same_bytes_as_per_call_tables(monkeypatch):
    """The window table of a base is cached per device (msm_fixed.hip, FbTab): every call — first (miss), repeated
    (hit), after six other bases have pushed it out of the four slots (rebuild), G1 and G2, both output layouts, host
    and device-resident entry — returns the bytes of the per-call-table path (OZK_FB_TABLE_CACHE=0), and sampled
    elements equal the oracle's."""
    import ctypes
    import numpy as np
    import torch
    from octopuszk_amd import lib
    L = lib.load()
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    n, w, outerc = 3000, 13, 20
    rng = np.random.default_rng(77)
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x1F
    d_sc = torch.from_numpy(sc.reshape(-1)).cuda()
    st = int(torch.cuda.current_stream().cuda_stream)

    def bases(bn):
        G = o.G1 if bn == 1 else o.G2
        return [G.mul(G.one, 0xABCDEF + 17 * k) for k in range(7)]

    def host_call(bn, bw, compact):
        out = np.zeros(n * (192 if bn == 1 else 384) // (2 if compact else 1), dtype=np.uint8)
        if compact:
            lib.check(L.ozk_fixed_batch_msm_compact_host(outerc, w, n, vp(bw), vp(sc), bn, 0, vp(out)))
        else:
            lib.check(L.ozk_fixed_batch_msm_host(outerc, w, outerc, 1 << w, n, 254, vp(bw), vp(sc), bn, 0, vp(out)))
        return out.tobytes()

    def dev_call(bn, bw, compact):
        wsb = int(L.ozk_fixed_batch_msm_workspace_bytes(outerc, w, n, bn))
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        out = torch.zeros(n * (192 if bn == 1 else 384) // (2 if compact else 1), dtype=torch.uint8, device="cuda")
        lib.check(L.ozk_fixed_batch_msm_base_dev(outerc, w, n, vp(bw), int(d_sc.data_ptr()), bn, int(out.data_ptr()), compact,
                                                 int(ws.data_ptr()), wsb, st))
        torch.cuda.synchronize()
        return bytes(out.cpu().numpy())

    for bn in (1, 2):
        G = o.G1 if bn == 1 else o.G2
        pts = bases(bn)
        wires = [np.frombuffer(o.g1_to_wire(P) if bn == 1 else o.g2_to_wire(P), dtype=np.uint8).copy() for P in pts]
        # reference bytes: per-call tables
        monkeypatch.setenv("OZK_FB_TABLE_CACHE", "0")
        lib.check(L.ozk_tuning_reload())
        ref = {(k, c): host_call(bn, wires[k], c) for k in range(7) for c in (0, 1)}
        monkeypatch.delenv("OZK_FB_TABLE_CACHE")
        lib.check(L.ozk_tuning_reload())
        # miss, hit, the other layout (hit), the device-resident entry (hit)
        assert host_call(bn, wires[0], 0) == ref[(0, 0)]
        assert host_call(bn, wires[0], 0) == ref[(0, 0)]
        assert host_call(bn, wires[0], 1) == ref[(0, 1)]
        assert dev_call(bn, wires[0], 0) == ref[(0, 0)]
        assert dev_call(bn, wires[0], 1) == ref[(0, 1)]
        # six more bases: base 0 leaves the four slots; then it is rebuilt
        for k in range(1, 7):
            assert dev_call(bn, wires[k], 1) == ref[(k, 1)], k
            assert host_call(bn, wires[k], 0) == ref[(k, 0)], k
        assert host_call(bn, wires[0], 1) == ref[(0, 1)]
        # against the oracle (compact layout: X | Y | Z little-endian)
        cw = 32
        per = 3 * cw * (1 if bn == 1 else 2)
        raw = ref[(3, 1)]
        for i in (0, 1, n // 2, n - 1):
            s_i = int.from_bytes(sc[i].tobytes(), "little")
            want = G.to_affine(G.mul(pts[3], s_i))
            rec = raw[i * per:(i + 1) * per]
            if bn == 1:
                got = tuple(int.from_bytes(rec[j * cw:(j + 1) * cw], "little") for j in range(3))
                assert (got[0], got[1]) == (want[0], want[1]) and got[2] == 1, i
            else:
                v = [int.from_bytes(rec[j * cw:(j + 1) * cw], "little") for j in range(6)]
                assert ((v[0], v[1]), (v[2], v[3])) == (tuple(want[0]), tuple(want[1])), i
    lib.check(L.ozk_host_cache_release())


def test_fixed_base_table_cache_under_concurrent_callers():
    """Four host threads issue fixed-base batches over SIX bases through the cached-table entry points at once (more
    bases than the four slots per device: tables are evicted while other threads hold theirs pinned); every result
    must be the bytes the same call returns alone with per-call tables."""
    import ctypes
    import threading
    import numpy as np
    from octopuszk_amd import lib
    L = lib.load()
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    n, w, outerc = 2500, 12, 22
    rng = np.random.default_rng(5)
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x1F
    cases = []
    for bn in (1, 2):
        G = o.G1 if bn == 1 else o.G2
        for k in range(3):
            P = G.mul(G.one, 0x5151 + 29 * k + bn)
            cases.append((bn, np.frombuffer(o.g1_to_wire(P) if bn == 1 else o.g2_to_wire(P), dtype=np.uint8).copy()))

    def call(bn, bw):
        out = np.zeros(n * (96 if bn == 1 else 192), dtype=np.uint8)
        lib.check(L.ozk_fixed_batch_msm_compact_host(outerc, w, n, vp(bw), vp(sc), bn, 0, vp(out)))
        return out.tobytes()

    os_env = __import__("os").environ
    os_env["OZK_FB_TABLE_CACHE"] = "0"
    lib.check(L.ozk_tuning_reload())
    try:
        ref = [call(bn, bw) for bn, bw in cases]
    finally:
        del os_env["OZK_FB_TABLE_CACHE"]
        lib.check(L.ozk_tuning_reload())
    bad = []

    def worker(t):
        for r in range(6):
            i = (t * 5 + r * 7) % len(cases)
            if call(*cases[i]) != ref[i]:
                bad.append((t, r, i))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not bad, bad
    lib.check(L.ozk_host_cache_release())
