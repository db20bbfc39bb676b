"""GPU parity of the end-to-end Groth16 path (BASELINE.json configs[4]): synthetic R1CS
(R1CSConstruction.serialConstruct) -> proving key from the fixed-base kernels (SerialSetup.generate) -> witness
map + 4 G1 MSMs + 2 double MSMs resident in HBM (SerialProver.prove)."""
import os

import pytest

from oracle import bn254 as o
from oracle import groth16 as g

pytestmark = pytest.mark.gpu


def _g1_wire_aff(P):
    return o.g1_to_wire(o.G1.to_affine(P))


def _g2_wire_aff(P):
    return o.g2_to_wire(o.G2.to_affine(P))


def _dev_bytes(t):
    return bytes(t.cpu().numpy())


def test_groth16_2p10_bit_exact_vs_oracle():
    """(i) every element of the proving key and the proof (A, B, C) equal the oracle's restatement of
    SerialSetup.generate / SerialProver.prove bit for bit (affine-normalised)."""
    from octopuszk_amd import zksnark as z
    nc, ni = 1 << 10, 15
    r1cs_o, primary, auxiliary = g.serial_construct(nc, ni)
    crs_o = g.serial_setup(r1cs_o)
    (A, B, C), info = g.serial_prove(crs_o, primary, auxiliary)

    r1cs, p2, a2 = z.serial_construct(nc, ni)
    assert (p2, a2) == (primary, auxiliary)
    crs = z.serial_setup_generate(r1cs)
    pk = crs.proving_key
    assert (crs.window_g1, crs.window_g2) == (crs_o.window_g1, crs_o.window_g2)
    assert crs.gen_g1 == _g1_wire_aff(crs_o.gen_g1) and crs.gen_g2 == _g2_wire_aff(crs_o.gen_g2)
    assert _dev_bytes(pk.alpha_g1) == _g1_wire_aff(crs_o.alpha_g1)
    assert _dev_bytes(pk.beta_g1) == _g1_wire_aff(crs_o.beta_g1)
    assert _dev_bytes(pk.delta_g1) == _g1_wire_aff(crs_o.delta_g1)
    assert _dev_bytes(pk.beta_g2) == _g2_wire_aff(crs_o.beta_g2)
    assert _dev_bytes(pk.delta_g2) == _g2_wire_aff(crs_o.delta_g2)
    assert _dev_bytes(crs.gamma_g2) == _g2_wire_aff(crs_o.gamma_g2)
    assert _dev_bytes(pk.delta_abc_g1) == b"".join(map(_g1_wire_aff, crs_o.delta_abc_g1))
    assert _dev_bytes(pk.query_a) == b"".join(map(_g1_wire_aff, crs_o.query_a))
    assert _dev_bytes(pk.query_b_g1) == b"".join(_g1_wire_aff(q[0]) for q in crs_o.query_b)
    assert _dev_bytes(pk.query_b_g2) == b"".join(_g2_wire_aff(q[1]) for q in crs_o.query_b)
    assert _dev_bytes(pk.query_h) == b"".join(map(_g1_wire_aff, crs_o.query_h))
    assert _dev_bytes(crs.gamma_abc_g1) == b"".join(map(_g1_wire_aff, crs_o.gamma_abc_g1))

    prover = z.SerialProver(pk)
    try:
        # constraint evaluation on the device == LinearCombination.evaluate on the host == the oracle
        import torch
        ev_host, _ = z.constraint_evaluations(r1cs, primary + auxiliary)
        ev_dev = prover.r1cs_dev.evaluate(torch.from_numpy(z.assignment_bytes(primary + auxiliary)).cuda())
        torch.cuda.synchronize()
        for k in range(3):
            assert bytes(ev_dev[k].cpu().numpy()) == z._le32(ev_host[k]), "ABC"[k]
        for _ in range(2):   # second call: buffers reused
            proof = prover.prove(primary, auxiliary)
            assert prover.coefficients_h() == info["H"]
            assert proof.g_a == o.g1_out_le(o.G1.to_affine(A))
            assert proof.g_b == o.g2_out_le(o.G2.to_affine(B))
            assert proof.g_c == o.g1_out_le(o.G1.to_affine(C))
    finally:
        prover.close()


@pytest.mark.parametrize("nc,ni", [(50, 50), (257, 1), (1000, 24)])
def test_groth16_small_shapes_vs_exponent(nc, ni):
    """ragged sizes (numInputs = numConstraints, a single input, non-power-of-two): proof == known scalars
    times the generators, and the verification equation holds in the exponent."""
    _check_by_known_scalars(nc, ni)


def test_groth16_2p20_known_scalars():
    """(ii) BASELINE.json configs[4] size: 2^20 constraints, 1023 inputs (ZKSNARKProfiling /
    serialzkSNARKProfiler.sh).  Every key element is a known scalar times a generator, so the expected proof
    is (a genG1, b genG2, c genG1) with a, b, c from exact integer arithmetic."""
    logn = int(os.environ.get("OZK_TEST_GROTH16_LOGN", "20"))
    _check_by_known_scalars(1 << logn, 1023)


def _check_by_known_scalars(nc, ni):
    from octopuszk_amd import zksnark as z
    R = o.R
    r1cs, primary, auxiliary = z.serial_construct(nc, ni)
    crs = z.serial_setup_generate(r1cs)
    prover = z.SerialProver(crs.proving_key)
    try:
        proof = prover.prove(primary, auxiliary)
        H = prover.coefficients_h()
    finally:
        prover.close()
    full = primary + auxiliary
    q, sec, sc = crs.qap, crs.secrets, crs.scalars
    r = s = z.fr_random()
    m = q.degree
    assert H[m - 1] == 0 and H[m] == 0                      # SerialProver.java:47-49
    a = (sec["alpha"] + sum(x * y for x, y in zip(full, q.At)) + r * sec["delta"]) % R
    b = (sec["beta"] + sum(x * y for x, y in zip(full, q.Bt)) + s * sec["delta"]) % R
    c = (sum(x * y for x, y in zip(full[ni:], sc["delta_abc"])) + sum(x * y for x, y in zip(H, sc["ht"]))
         + a * s + b * r - r * s * sec["delta"]) % R
    # Groth16 verification equation in the exponent (zkSNARK/Verifier.java:24-59): ties the host-side QAP
    # instance, the witness map on the GPU (H) and the key scalars together
    acc = sum(x * y for x, y in zip(primary, sc["gamma_abc"])) % R
    assert (a * b - sec["alpha"] * sec["beta"] - acc * sec["gamma"] - c * sec["delta"]) % R == 0
    gen = sec["generator"]
    assert proof.g_a == o.g1_out_le(o.G1.to_affine(o.G1.mul(o.G1.one, a * gen % R)))
    assert proof.g_b == o.g2_out_le(o.G2.to_affine(o.G2.mul(o.G2.one, b * gen % R)))
    assert proof.g_c == o.g1_out_le(o.G1.to_affine(o.G1.mul(o.G1.one, c * gen % R)))


def test_r1cs_evaluate_with_coefficients_and_long_rows():
    """ozk_r1cs_evaluate_dev on a matrix the synthetic circuits do not produce: random coefficients, empty rows,
    index-0 terms (which count as `one`, LinearCombination.java:45), rows of 1 ... 700 terms."""
    import numpy as np
    import torch
    from octopuszk_amd import zksnark as z
    rng = np.random.default_rng(7)
    nv, rows = 900, 300
    full = [1] + [int.from_bytes(rng.bytes(32), "little") % o.R for _ in range(nv - 1)]
    lens = rng.integers(0, 6, size=rows)
    lens[5], lens[17], lens[250] = 700, 65, 64
    ptr = np.concatenate(([0], np.cumsum(lens)))
    idx = rng.integers(0, nv, size=int(ptr[-1]))
    val = np.array([int.from_bytes(rng.bytes(32), "little") % o.R for _ in range(int(ptr[-1]))], dtype=object)
    val[3] = 0
    lc = z.LinearCombinations(ptr, idx, val)
    want = lc.evaluate(np.array(full, dtype=object))
    oracle_rows = [g.lc_evaluate([(int(idx[t]), int(val[t])) for t in range(ptr[i], ptr[i + 1])], full) for i in range(rows)]
    assert [int(x) for x in want] == oracle_rows
    r1 = z.R1CSRelation(lc, lc, lc, 1, nv - 1)
    r1.num_constraints = rows
    dev = z.R1CSDevice(r1)
    ev = dev.evaluate(torch.from_numpy(z.assignment_bytes(full)).cuda())
    torch.cuda.synchronize()
    got = bytes(ev[1].cpu().numpy())          # B: no extra rows
    assert got[:rows * 32] == z._le32(oracle_rows) and not any(got[rows * 32:])
    got_a = bytes(ev[0].cpu().numpy())        # A: row `rows` is the extra input_0 * 0 = 0 row, i.e. z_0 = 1
    assert got_a[:rows * 32] == z._le32(oracle_rows) and got_a[rows * 32:(rows + 1) * 32] == z._le32([1])


@pytest.mark.parametrize("shape", ["synthetic", "random_coefficients"])
def test_qap_instance_on_device_equals_host_and_oracle(shape):
    """R1CStoQAP.R1CStoQAPRelation on the device (Lagrange coefficients with shared inversions, sparse products over
    the transposed matrices, powers of t) against the host version in exact integers and the oracle's restatement:
    At, Bt, Ct, Ht, Zt."""
    import numpy as np
    from octopuszk_amd import zksnark as z
    rng = np.random.default_rng(11)
    if shape == "synthetic":
        r1cs, _, _ = z.serial_construct(700, 13)
    else:
        nv, rows, ni = 500, 300, 7
        mats = []
        for _ in range(3):
            lens = rng.integers(0, 5, size=rows)
            lens[4], lens[9] = 450, 66                       # long rows; variable columns get long too (below)
            ptr = np.concatenate(([0], np.cumsum(lens)))
            idx = rng.integers(0, nv, size=int(ptr[-1]))
            idx[ptr[4]:ptr[4] + 200] = 3                     # 200 terms of one variable: a long transposed row
            val = np.array([int.from_bytes(rng.bytes(32), "little") % o.R for _ in range(int(ptr[-1]))], dtype=object)
            mats.append(z.LinearCombinations(ptr, idx, val))
        r1cs = z.R1CSRelation(mats[0], mats[1], mats[2], ni, nv - ni)
    t = z.fr_random()
    host = z.r1cs_to_qap_relation(r1cs, t)
    dev = z.r1cs_to_qap_relation_dev(r1cs, t)
    assert isinstance(dev, z.QAPRelationDevice)
    assert dev.degree == host.degree and dev.Zt == host.Zt
    assert dev.At == host.At and dev.Bt == host.Bt and dev.Ct == host.Ct and dev.Ht == host.Ht
    lag = z._ints_from_dev(dev.d_lagrange)
    assert lag == z.lagrange_coefficients(t, host.degree) == g.lagrange_coefficients(t, host.degree)
    # a point of the domain takes the reference's indicator branch (host)
    w = z.root_of_unity(host.degree)
    ind = z.r1cs_to_qap_relation_dev(r1cs, pow(w, 5, o.R))
    assert not isinstance(ind, z.QAPRelationDevice) and ind.Zt == 0


def test_lagrange_entry_refuses_a_t_inside_the_domain():
    """ozk_qap_lagrange_dev with t = omega^k: one denominator t - omega^i is zero and the shared inversion of its lane
    would silently zero eight coefficients.  The entry point computes t^m on the host and returns
    OZK_E_INVALID; the reference takes its indicator branch there (FFTAuxiliary.java:262-276), as zksnark.py does on
    its side.  A t outside the domain still works through the same call."""
    import ctypes
    import torch
    from octopuszk_amd import lib
    from oracle import bn254 as o
    L = lib.load()
    m = 1 << 10
    omega = o.fr_root_of_unity(m)
    wsb = int(L.ozk_qap_lagrange_workspace_bytes(m))
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    out = torch.empty(m * 32, dtype=torch.uint8, device="cuda")
    zt = torch.empty(32, dtype=torch.uint8, device="cuda")
    st = int(torch.cuda.current_stream().cuda_stream)
    le = lambda v: ctypes.create_string_buffer(int(v).to_bytes(32, "little"), 32)
    for k in (0, 1, 7, m - 1):
        rc = L.ozk_qap_lagrange_dev(ctypes.cast(le(pow(omega, k, o.R)), ctypes.c_void_p), ctypes.cast(le(omega), ctypes.c_void_p), m,
                                    int(out.data_ptr()), int(zt.data_ptr()), int(ws.data_ptr()), wsb, st)
        assert rc != 0 and b"domain" in L.ozk_last_error(), k
    t = 123456789
    lib.check(L.ozk_qap_lagrange_dev(ctypes.cast(le(t), ctypes.c_void_p), ctypes.cast(le(omega), ctypes.c_void_p), m,
                                     int(out.data_ptr()), int(zt.data_ptr()), int(ws.data_ptr()), wsb, st))
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(m, 32)
    want = g.lagrange_coefficients(t, m)   # FFTAuxiliary.serialRadix2LagrangeCoefficients restated (oracle/groth16.py)
    zt_v = int.from_bytes(bytes(zt.cpu().numpy()), "little")
    assert zt_v == (pow(t, m, o.R) - 1) % o.R
    assert [int.from_bytes(bytes(r), "little") for r in got] == [w % o.R for w in want]
