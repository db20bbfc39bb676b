"""GPU parity tests of VariableBaseMSM through the C ABI against the oracle (bit-exact on
the affine-normalised point)."""
import random

import pytest

from oracle import bn254 as o
from oracle.javarand import JavaRandom

pytestmark = pytest.mark.gpu


def _rand_points(C, n, rng, affine=True):
    pts = []
    for _ in range(n):
        P = C.mul(C.one, rng.randrange(1, 1 << 64))
        pts.append(C.to_affine(P) if affine else P)
    return pts


def _run_g1(scalars, bases):
    from octopuszk_amd import variable_base_msm as vb
    raw = vb.variable_base_serial_msm_native_helper(vb.marshal_g1(bases), vb.marshal_scalars(scalars),
                                                    len(scalars), 1, 0)
    return raw


@pytest.mark.parametrize("n", [1, 2, 3, 17, 64, 257, 1023])
def test_g1_small_vs_oracle(n):
    rng = random.Random(100 + n)
    bases = _rand_points(o.G1, n, rng)
    scalars = [rng.randrange(o.R) for _ in range(n)]
    want = o.G1.to_affine(o.pippenger_msm(o.G1, scalars, bases))
    assert _run_g1(scalars, bases) == o.g1_out_le(want)


def test_g1_toy_kat():
    # SerialVariableBaseMSMTest.java:31-77 restated on BN254: (3,11,2,8).(5G,2G,7G,3G) = 75G
    G = o.G1
    bases = [G.to_affine(G.mul(G.one, k)) for k in (5, 2, 7, 3)]
    want = G.to_affine(G.mul(G.one, 75))
    assert _run_g1([3, 11, 2, 8], bases) == o.g1_out_le(want)


def test_g1_edge_scalars_and_points():
    G = o.G1
    rng = random.Random(5)
    bases = _rand_points(G, 40, rng)
    bases[3] = G.zero                      # infinity among the bases
    bases[4] = bases[5]                    # repeated base
    bases[6] = G.negate(bases[7])          # P and -P
    scalars = [rng.randrange(o.R) for _ in range(40)]
    scalars[0] = 0
    scalars[1] = 1
    scalars[2] = o.R - 1
    scalars[6] = scalars[7] = 12345        # P*s + (-P)*s = 0 inside one bucket
    scalars[4] = scalars[5]
    want = G.to_affine(o.naive_msm(G, scalars, bases))
    assert _run_g1(scalars, bases) == o.g1_out_le(want)
    # all-zero scalars -> infinity (0, 1, 0)
    assert _run_g1([0] * 40, bases) == o.g1_out_le(G.zero)


def test_g1_profiler_shaped_inputs():
    # VariableBaseMSMProfiling.java:19-31: ONE base repeated, Fp.random scalars (64-bit or r-64-bit)
    G = o.G1
    n = 300
    base = G.to_affine(G.mul(G.one, 987654321))
    jr = JavaRandom(10)
    scalars = [jr.next_long() % o.R for _ in range(n)]
    want = G.to_affine(G.mul(base, sum(scalars) % o.R))
    assert _run_g1(scalars, [base] * n) == o.g1_out_le(want)


def test_g1_jacobian_inputs():
    # reference-produced keys are un-normalised Jacobian points (Z != 1)
    G = o.G1
    rng = random.Random(9)
    bases = _rand_points(G, 50, rng, affine=False)
    scalars = [rng.randrange(o.R) for _ in range(50)]
    want = G.to_affine(o.naive_msm(G, scalars, bases))
    assert _run_g1(scalars, bases) == o.g1_out_le(want)


# ---------------------------------------------------------------- G2 and the double MSM
def _run_g2(scalars, bases):
    from octopuszk_amd import variable_base_msm as vb
    return vb.variable_base_serial_msm_native_helper(vb.marshal_g2(bases), vb.marshal_scalars(scalars),
                                                     len(scalars), 2, 0)


@pytest.mark.parametrize("n", [1, 3, 64, 300])
def test_g2_small_vs_oracle(n):
    rng = random.Random(200 + n)
    bases = _rand_points(o.G2, n, rng)
    scalars = [rng.randrange(o.R) for _ in range(n)]
    want = o.G2.to_affine(o.pippenger_msm(o.G2, scalars, bases))
    assert _run_g2(scalars, bases) == o.g2_out_le(want)


def test_g2_edge_cases():
    G = o.G2
    rng = random.Random(6)
    bases = _rand_points(G, 24, rng, affine=False)   # Jacobian (Z != 1) inputs
    bases[3] = G.zero                                 # fork quirk: (0,0,0) encodes infinity
    bases[4] = bases[5]
    bases[6] = G.negate(bases[7])
    scalars = [rng.randrange(o.R) for _ in range(24)]
    scalars[0], scalars[1], scalars[2] = 0, 1, o.R - 1
    scalars[6] = scalars[7] = 999
    want = G.to_affine(o.naive_msm(G, scalars, bases))
    assert _run_g2(scalars, bases) == o.g2_out_le(want)
    assert _run_g2([0] * 24, bases) == o.g2_out_le(G.zero_affine)


def test_double_msm():
    # variableBaseDoubleMSMNativeHelper: 576 B = G1 (192) || G2 (384)  (VariableBaseMSM.cu:1781-1784)
    from octopuszk_amd import variable_base_msm as vb
    rng = random.Random(8)
    n = 40
    b1 = _rand_points(o.G1, n, rng)
    b2 = _rand_points(o.G2, n, rng)
    sc = [rng.randrange(o.R) for _ in range(n)]
    raw = vb.variable_base_double_msm_native_helper(vb.marshal_g1(b1), vb.marshal_g2(b2), vb.marshal_scalars(sc), n, 0)
    assert raw[:192] == o.g1_out_le(o.G1.to_affine(o.naive_msm(o.G1, sc, b1)))
    assert raw[192:] == o.g2_out_le(o.G2.to_affine(o.naive_msm(o.G2, sc, b2)))


def test_serial_msm_mirror_chunks_and_sums():
    # VariableBaseMSM.serialMSM (VariableBaseMSM.java:199-338): chunk results summed with the group add
    from octopuszk_amd import variable_base_msm as vb
    rng = random.Random(13)
    n = 50
    bases = _rand_points(o.G1, n, rng)
    sc = [rng.randrange(o.R) for _ in range(n)]
    old = vb.G1_ITERATION_BATCH
    vb.G1_ITERATION_BATCH = 16  # force 4 chunks
    try:
        got = vb.serial_msm(sc, bases, o.G1.add, o.G1.zero, is_g1=True)
    finally:
        vb.G1_ITERATION_BATCH = old
    assert o.G1.equals(got, o.naive_msm(o.G1, sc, bases))


# ---------------------------------------------------------------- full-size, size-independent checks
def _device_msm(n, seed_b, seed_s):
    import numpy as np
    import torch
    from octopuszk_amd import device as dev
    bases = dev.gen_g1_bases(n, seed=seed_b)
    rng = np.random.default_rng(seed_s)
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x1F
    d_sc = torch.from_numpy(sc.reshape(-1)).cuda()
    ws = dev.VarMsmWorkspace(n, 1)
    out = ws.run(bases, d_sc)
    torch.cuda.synchronize()
    return bases, d_sc, sc, bytes(out.cpu().numpy())


@pytest.mark.parametrize("logn", [16, 20, 21])
def test_g1_full_size_discrete_log_identity(logn):
    """BASELINE configs: 2^16, 2^20 (1 GPU), 2^21 (= 2^24 / 8 GPUs).  Bases are k_i * G with known k_i,
    so sum s_i P_i must equal (sum s_i k_i mod r) * G — computed here with exact integers."""
    from octopuszk_amd import device as dev
    n = 1 << logn
    _, _, sc, got = _device_msm(n, 2, 1)
    ks = dev.gen_base_logs(n, 2)
    acc = 0
    for i in range(n):
        acc += int.from_bytes(sc[i].tobytes(), "little") * ks[i]
    assert got == o.g1_out_le(o.G1.to_affine(o.G1.mul(o.G1.one, acc % o.R)))


def test_g1_max_chunk_linearity():
    """2^23 pairs = the reference's maximum JNI chunk (VariableBaseMSM.java:211): the MSM over the
    whole range equals the sum of the MSMs over its two halves (HIP point sum)."""
    import torch
    from octopuszk_amd import device as dev
    n = 1 << 23
    bases, d_sc, _, whole = _device_msm(n, 7, 8)
    h = n // 2
    ws = dev.VarMsmWorkspace(h, 1)
    parts = []
    for k in range(2):
        out = ws.run(bases[k * h * 96:(k + 1) * h * 96], d_sc[k * h * 32:(k + 1) * h * 32])
        torch.cuda.synchronize()
        parts.append(out.clone())
    s = dev.points_sum(torch.cat(parts), 2, 1)
    torch.cuda.synchronize()
    assert bytes(s.cpu().numpy()) == whole
    assert whole != o.g1_out_le(o.G1.zero)


def test_g1_2pow24_plain_windows_equal_two_glv_halves():
    """BASELINE.json cfg-4 (N = 2^24).  One 2^24-pair call takes the plain 256-bit unsigned-window
    plan (the GLV plan needs 2n <= 2^24 sortable points); its result must equal the sum of the two
    2^23-pair halves, which take the GLV + signed-digit plan: two different code paths, same bytes."""
    import torch
    from octopuszk_amd import device as dev, lib
    L = lib.load()
    n = 1 << 24
    assert L.ozk_var_msm_glv(n) == 0 and L.ozk_var_msm_glv(n // 2) == 1
    bases, d_sc, _, whole = _device_msm(n, 11, 12)
    h = n // 2
    ws = dev.VarMsmWorkspace(h, 1)
    parts = []
    for k in range(2):
        out = ws.run(bases[k * h * 96:(k + 1) * h * 96], d_sc[k * h * 32:(k + 1) * h * 32])
        torch.cuda.synchronize()
        parts.append(out.clone())
    s = dev.points_sum(torch.cat(parts), 2, 1)
    torch.cuda.synchronize()
    assert bytes(s.cpu().numpy()) == whole
    assert whole != o.g1_out_le(o.G1.zero)


def test_pipelined_equals_serial():
    """Two MSMs in flight (head of k+1 overlapping tail of k) give the same bytes as serial calls."""
    import numpy as np
    import torch
    from octopuszk_amd import device as dev
    n = 1 << 14
    bases = dev.gen_g1_bases(n, seed=3)
    rng = np.random.default_rng(9)
    inputs = []
    for _ in range(5):
        sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        sc[:, 31] &= 0x1F
        inputs.append(torch.from_numpy(sc.reshape(-1)).cuda())
    ws = dev.VarMsmWorkspace(n, 1)
    serial = []
    for d_sc in inputs:
        out = ws.run(bases, d_sc)
        torch.cuda.synchronize()
        serial.append(bytes(out.cpu().numpy()))
    pipe = dev.VarMsmPipeline(n, 1, depth=2)
    got, prev = [], None
    for d_sc in inputs:
        t = pipe.submit(bases, d_sc)
        if prev is not None:
            got.append(bytes(pipe.result(prev).cpu().numpy()))
        prev = t
    got.append(bytes(pipe.result(prev).cpu().numpy()))
    assert got == serial and len(set(serial)) == 5


def test_g1_profiler_shaped_full_size():
    """VariableBaseMSMProfiling.java:19-31 at 2^20: ONE base repeated, Fp.random scalars (64-bit or
    r - 64-bit): half of every high window lands in one bucket (giant coarse bins, long runs)."""
    import numpy as np
    import torch
    from octopuszk_amd import device as dev
    n = 1 << 20
    rng = np.random.default_rng(10)
    lows = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
    neg = rng.integers(0, 2, size=n).astype(bool)
    vals = [(o.R - int(v)) if ng else int(v) for v, ng in zip(lows, neg)]
    sc = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint8).copy()
    base = o.G1.to_affine(o.G1.mul(o.G1.one, 987654321))
    bases = np.frombuffer(o.g1_to_wire(base) * n, dtype=np.uint8).copy()
    d_bases, d_scalars = torch.from_numpy(bases).cuda(), torch.from_numpy(sc).cuda()
    ws = dev.VarMsmWorkspace(n, 1)
    out = ws.run(d_bases, d_scalars)
    torch.cuda.synchronize()
    assert bytes(out.cpu().numpy()) == o.g1_out_le(o.G1.to_affine(o.G1.mul(base, sum(vals) % o.R)))


@pytest.mark.parametrize("kind", ["all_equal", "two_values", "all_one", "all_r_minus_1", "all_zero"])
def test_g1_degenerate_scalar_distributions_full_size(kind):
    """Every point in ONE bucket per window (all scalars equal): the extreme of the skew the big-bin sort
    and the multi-level reduction exist for; 2^20 pairs, bases k_i G with known k_i."""
    import numpy as np
    import torch
    from octopuszk_amd import device as dev
    n = (1 << 20) + 4321          # not a power of two either
    bases = dev.gen_g1_bases(n, seed=21)
    ks = dev.gen_base_logs(n, 21)
    c1 = 0x1234567890abcdef1234567890abcdef1234567890abcdef1234567890abcd % o.R
    c2 = (o.R - 0xfedcba9876543210fedcba98765) % o.R
    if kind == "all_equal":
        vals = [c1] * n
    elif kind == "two_values":
        vals = [c1 if (i * 7919) % 3 else c2 for i in range(n)]
    elif kind == "all_one":
        vals = [1] * n
    elif kind == "all_r_minus_1":
        vals = [o.R - 1] * n
    else:
        vals = [0] * n
    uniq = {v: np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint8) for v in set(vals)}
    sc = np.stack([uniq[v] for v in vals]).reshape(-1).copy()
    ws = dev.VarMsmWorkspace(n, 1)
    out = ws.run(bases, torch.from_numpy(sc).cuda())
    torch.cuda.synchronize()
    acc = sum(v * k for v, k in zip(vals, ks)) % o.R
    assert bytes(out.cpu().numpy()) == o.g1_out_le(o.G1.to_affine(o.G1.mul(o.G1.one, acc)))


def test_g1_profiler_shaped_repeatable():
    """Regression for the big-bin sort (k_sortbig_*): the same skewed 2^20 MSM run 25 times over poisoned
    workspaces must give the same, correct bytes every time.  A missing wait before a workgroup barrier
    (curve.cuh block_sync) used to drop one wave's LDS counts in ~1 run out of 5."""
    import numpy as np
    import torch
    from octopuszk_amd import device as dev
    n = 1 << 20
    rng = np.random.default_rng(10)
    lows = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
    neg = rng.integers(0, 2, size=n).astype(bool)
    vals = [(o.R - int(v)) if ng else int(v) for v, ng in zip(lows, neg)]
    sc = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint8).copy()
    base = o.G1.to_affine(o.G1.mul(o.G1.one, 987654321))
    bases = np.frombuffer(o.g1_to_wire(base) * n, dtype=np.uint8).copy()
    d_bases, d_scalars = torch.from_numpy(bases).cuda(), torch.from_numpy(sc).cuda()
    want = o.g1_out_le(o.G1.to_affine(o.G1.mul(base, sum(vals) % o.R)))
    ws = dev.VarMsmWorkspace(n, 1)
    bad = []
    for rep in range(25):
        ws.ws.fill_(0xFF if rep % 2 else 0)
        out = ws.run(d_bases, d_scalars)
        torch.cuda.synchronize()
        if bytes(out.cpu().numpy()) != want:
            bad.append(rep)
    assert bad == []


# ---- GLV endomorphism path (csrc/glv.cuh): scalars that stress the decomposition ----
_LAM = 4407920970296243842393367215006156084916469457145843978461
_GLV_EDGE = [0, 1, 2, o.R - 1, o.R - 2, _LAM, _LAM - 1, _LAM + 1, o.R - _LAM, o.R - _LAM - 1,
             (1 << 127) - 1, 1 << 127, (1 << 127) + 1, 1 << 128, 1 << 253, o.R // 2, o.R // 3,
             9931322734385697763, 147946756881789319010696353538189108491,
             # not reduced: the library reduces any 256-bit value mod r
             o.R, o.R + 1, (1 << 256) - 1, 5 * o.R + 7]


def _le32_any(s):
    return int(s).to_bytes(32, "little")


def test_g1_glv_edge_scalars():
    from octopuszk_amd import variable_base_msm as vb
    G = o.G1
    rng = random.Random(31)
    scalars = _GLV_EDGE + [rng.randrange(1 << 256) for _ in range(41)]
    bases = _rand_points(G, len(scalars), rng)
    bases[5] = G.zero
    raw = vb.variable_base_serial_msm_native_helper(vb.marshal_g1(bases), b"".join(map(_le32_any, scalars)),
                                                    len(scalars), 1, 0)
    want = G.to_affine(o.naive_msm(G, [s % o.R for s in scalars], bases))
    assert raw == o.g1_out_le(want)
    # each edge scalar on its own (a one-pair MSM is the scalar multiplication itself)
    for s in _GLV_EDGE:
        raw = vb.variable_base_serial_msm_native_helper(vb.marshal_g1(bases[:1]), _le32_any(s), 1, 1, 0)
        assert raw == o.g1_out_le(G.to_affine(G.mul(bases[0], s % o.R))), hex(s)


def test_g2_glv_edge_scalars():
    from octopuszk_amd import variable_base_msm as vb
    G = o.G2
    rng = random.Random(32)
    scalars = _GLV_EDGE + [rng.randrange(1 << 256) for _ in range(9)]
    bases = _rand_points(G, len(scalars), rng)
    raw = vb.variable_base_serial_msm_native_helper(vb.marshal_g2(bases), b"".join(map(_le32_any, scalars)),
                                                    len(scalars), 2, 0)
    want = G.to_affine(o.naive_msm(G, [s % o.R for s in scalars], bases))
    assert raw == o.g2_out_le(want)


@pytest.mark.parametrize("type_", [1, 2])
def test_glv_off_equals_glv_on(type_, monkeypatch):
    # OZK_MSM_GLV=0 runs the plain 256-bit windows; both must give the same bytes
    G = o.G1 if type_ == 1 else o.G2
    rng = random.Random(33 + type_)
    n = 700 if type_ == 1 else 150
    bases = _rand_points(G, 16, rng) * (n // 16 + 1)
    bases = bases[:n]
    scalars = [rng.randrange(o.R) for _ in range(n)]
    run = _run_g1 if type_ == 1 else _run_g2
    from octopuszk_amd import lib
    L = lib.load()
    on = run(scalars, bases)
    glv_on = L.ozk_var_msm_glv(n)
    monkeypatch.setenv("OZK_MSM_GLV", "0")
    assert L.ozk_var_msm_glv(n) == glv_on == 1    # tuning variables are cached per process ...
    L.ozk_tuning_reload()                          # ... until a reload (include/ozk.h)
    try:
        assert L.ozk_var_msm_glv(n) == 0
        off = run(scalars, bases)
    finally:
        monkeypatch.delenv("OZK_MSM_GLV")
        L.ozk_tuning_reload()
    assert L.ozk_var_msm_glv(n) == 1
    assert on == off


# ---- prepared bases (include/ozk.h): same bytes as the per-call path ----
@pytest.mark.parametrize("type_", [1, 2])
def test_prepared_bases_equal_per_call_path(type_):
    from octopuszk_amd import variable_base_msm as vb, lib
    G = o.G1 if type_ == 1 else o.G2
    rng = random.Random(70 + type_)
    n = 500 if type_ == 1 else 120
    bases = _rand_points(G, n // 2, rng) + _rand_points(G, n - n // 2, rng, affine=False)   # Z = 1 and Z != 1
    bases[3] = G.zero
    wire = vb.marshal_g1(bases) if type_ == 1 else vb.marshal_g2(bases)
    pb = vb.PreparedBases(wire, n, type_, 0)
    for k in range(3):   # one handle, several MSMs
        scalars = [rng.randrange(o.R) for _ in range(n)]
        scalars[0], scalars[1] = 0, o.R - 1
        sw = vb.marshal_scalars(scalars)
        got = pb.msm(sw)
        assert got == vb.variable_base_serial_msm_native_helper(wire, sw, n, type_, 0)
        if k == 0:
            want = G.to_affine(o.naive_msm(G, scalars, bases))
            assert got == (o.g1_out_le(want) if type_ == 1 else o.g2_out_le(want))
    with pytest.raises(lib.OzkError):
        pb.msm(vb.marshal_scalars([1] * (n - 1)))
    pb.close()


def test_prepared_pipeline_full_size():
    import torch
    from octopuszk_amd import device as dev
    n = 1 << 20
    bases = dev.gen_g1_bases(n, seed=77)
    rng = random.Random(78)
    import numpy as np
    sc = np.random.default_rng(79).integers(0, 256, size=(3, n, 32), dtype=np.uint8)
    sc[:, :, 31] &= 0x1F
    pipe = dev.VarMsmPipeline(n, 1, depth=2)
    prep = pipe.prepare(bases)
    outs = []
    for k in range(3):
        d_sc = torch.from_numpy(sc[k].reshape(-1)).cuda()
        a = bytes(pipe.result(pipe.submit(bases, d_sc)).cpu().numpy())
        b = bytes(pipe.result(pipe.submit(prep, d_sc, prepared=True)).cpu().numpy())
        assert a == b
        outs.append(a)
    assert len(set(outs)) == 3


def test_g2_mid_size_discrete_log_identity():
    """G2 at 2^15 (the LDS-resident level-1 accumulator, signed digits, GLV on the twist): bases k_i G2
    from the fixed-base path (itself oracle-checked in test_fixed_base_gpu.py), so that
    sum s_i P_i = (sum s_i k_i) G2 can be checked with exact integers."""
    import ctypes
    import numpy as np
    import torch
    from octopuszk_amd import device as dev, lib
    L = lib.load()
    n = 1 << 15
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(88)
    ks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    ks[:, 8:] = 0
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x1F
    base = torch.from_numpy(np.frombuffer(o.g2_to_wire(o.G2.one), dtype=np.uint8).copy()).cuda()
    out_be = torch.empty(n * 384, dtype=torch.uint8, device="cuda")
    wsb = int(L.ozk_fixed_batch_msm_workspace_bytes(4, 16, n, 2))
    wsf = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    lib.check(L.ozk_fixed_batch_msm_dev(4, 16, n, p(base), p(torch.from_numpy(ks.reshape(-1)).cuda()), 2, p(out_be), p(wsf),
                                        wsb, st))
    torch.cuda.synchronize()
    be = out_be.cpu().numpy().reshape(n, 6, 64)
    assert not be[:, :, :32].any()
    wire = np.ascontiguousarray(be[:, :, ::-1][:, :, :32]).reshape(-1).copy()
    ws = dev.VarMsmWorkspace(n, 2)
    out = ws.run(torch.from_numpy(wire).cuda(), torch.from_numpy(sc.reshape(-1)).cuda())
    torch.cuda.synchronize()
    acc = sum(int.from_bytes(sc[i].tobytes(), "little") * int.from_bytes(ks[i].tobytes(), "little") for i in range(n)) % o.R
    assert bytes(out.cpu().numpy()) == o.g2_out_le(o.G2.to_affine(o.G2.mul(o.G2.one, acc)))


def test_host_entry_points_are_thread_safe():
    """The reference's natives are called from concurrent Spark task threads: four host
    threads issue G1 / G2 / prepared MSMs at the same time; every result must equal the sequential one."""
    import threading
    from octopuszk_amd import variable_base_msm as vb
    rng = random.Random(91)
    jobs = []
    for k in range(8):
        type_ = 1 if k % 3 else 2
        G = o.G1 if type_ == 1 else o.G2
        n = 3000 + 17 * k if type_ == 1 else 300 + k
        pts = _rand_points(G, 24, rng)
        bases = [pts[i % 24] for i in range(n)]
        wire = vb.marshal_g1(bases) if type_ == 1 else vb.marshal_g2(bases)
        sw = vb.marshal_scalars([rng.randrange(o.R) for _ in range(n)])
        jobs.append((wire, sw, n, type_))
    want = [vb.variable_base_serial_msm_native_helper(w, s, n, t, 0) for (w, s, n, t) in jobs]
    got = [[None] * len(jobs) for _ in range(4)]
    errs = []

    def worker(tid):
        try:
            for rep in range(2):
                for j in range(tid, len(jobs) + tid):
                    w, s, n, t = jobs[j % len(jobs)]
                    if (j + rep) % 2:
                        got[tid][j % len(jobs)] = vb.variable_base_serial_msm_native_helper(w, s, n, t, tid)
                    else:
                        pb = vb.PreparedBases(w, n, t, tid)
                        got[tid][j % len(jobs)] = pb.msm(s)
                        pb.close()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert errs == []
    for tid in range(4):
        assert got[tid] == want
