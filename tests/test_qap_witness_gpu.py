"""GPU parity tests of the QAP witness map (ozk_qap_witness_host; R1CStoQAP.java:163-230) against
the oracle restatement, bit-exact, and a full-size known answer."""
import random

import numpy as np
import pytest

from oracle import bn254 as o
from oracle import coracle

pytestmark = pytest.mark.gpu


def _oracle_with_c_fft(a, b, c, g=o.FR_MULT_GEN):
    """qap_witness_coefficients_h with the transforms done by the C oracle (fast enough for 2^16)."""
    m = len(a)
    w = o.fr_root_of_unity(m)
    wi = pow(w, -1, o.R)
    minv = pow(m, -1, o.R)

    def fft(v, om):
        raw = coracle.fft_fr(b"".join(o.to_le32(x) for x in v), m, o.to_le32(om))
        return [int.from_bytes(raw[64 * i:64 * i + 32], "little") for i in range(m)]

    def coset(v, gg):
        out, cs = list(v), 1
        for i in range(m):
            out[i] = out[i] * cs % o.R
            cs = cs * gg % o.R
        return out

    ev = []
    for v in (a, b, c):
        coef = [x * minv % o.R for x in fft(v, wi)]
        ev.append(fft(coset(coef, g), w))
    zinv = pow(o.compute_z(g, m), -1, o.R)
    h = [(x * y - z) * zinv % o.R for x, y, z in zip(*ev)]
    h = [x * minv % o.R for x in fft(h, wi)]
    return coset(h, pow(g, -1, o.R)) + [0]


@pytest.mark.parametrize("m", [2, 4, 8, 64, 512])
def test_small_vs_python_oracle(m):
    from octopuszk_amd import r1cs_to_qap as q
    rng = random.Random(500 + m)
    a = [rng.randrange(o.R) for _ in range(m)]
    b = [rng.randrange(o.R) for _ in range(m)]
    a[0], b[1 % m] = 0, o.R - 1
    c_sat = [x * y % o.R for x, y in zip(a, b)]
    c_any = [rng.randrange(o.R) for _ in range(m)]
    assert q.coefficients_h(a, b, c_sat) == o.qap_witness_coefficients_h(a, b, c_sat)
    assert q.coefficients_h(a, b, c_any) == o.qap_witness_coefficients_h(a, b, c_any)


@pytest.mark.parametrize("logm", [10, 11, 13, 16])
def test_vs_c_oracle_transforms(logm):
    from octopuszk_amd import r1cs_to_qap as q
    m = 1 << logm
    rng = random.Random(600 + logm)
    a = [rng.randrange(o.R) for _ in range(m)]
    b = [rng.randrange(o.R) for _ in range(m)]
    c = [x * y % o.R for x, y in zip(a, b)]
    c[7] = (c[7] + 1) % o.R   # one violated constraint: still the same function
    if logm == 10:
        assert _oracle_with_c_fft(a, b, c) == o.qap_witness_coefficients_h(a, b, c)
    assert q.coefficients_h(a, b, c) == _oracle_with_c_fft(a, b, c)


def _sparse(m, terms):
    v = np.zeros((m, 32), dtype=np.uint8)
    for k, coef in terms:
        v[k] = np.frombuffer(o.to_le32(coef % o.R), dtype=np.uint8)
    return v


@pytest.mark.parametrize("logm", [12, 21])
def test_full_size_known_quotient(logm):
    """a = x^(m-1) + 3, b = x^(m-2) + 7x + 1, c = a b mod (x^m - 1)  =>  a b - c = (x^(m-3) + 7)(x^m - 1),
    i.e. H = x^(m-3) + 7 exactly.  2^21 is the H-query domain of BASELINE.json cfg-5.  The evaluations
    come from the library's own forward FFT (validated on its own in test_fft_gpu.py)."""
    import ctypes
    from octopuszk_amd import lib, r1cs_to_qap as q
    L = lib.load()
    m = 1 << logm
    w = o.to_le32(o.fr_root_of_unity(m))
    polys = [_sparse(m, [(m - 1, 1), (0, 3)]),
             _sparse(m, [(m - 2, 1), (1, 7), (0, 1)]),
             _sparse(m, [(m - 1, 1), (m - 2, 3), (m - 3, 1), (1, 21), (0, 10)])]
    evals = []
    for pv in polys:
        out = np.empty((m, 64), dtype=np.uint8)
        lib.check(L.ozk_fft_host(pv.ctypes.data_as(ctypes.c_void_p), m, ctypes.cast(ctypes.c_char_p(w), ctypes.c_void_p),
                                 0, out.ctypes.data_as(ctypes.c_void_p)))
        assert not out[:, 32:].any()
        evals.append(np.ascontiguousarray(out[:, :32]).tobytes())
    raw = q.qap_witness_native_helper(evals[0], evals[1], evals[2], m, w, o.to_le32(o.FR_MULT_GEN), 0)
    want = np.zeros((m + 1, 32), dtype=np.uint8)
    want[0, 0] = 7
    want[m - 3, 0] = 1
    got = np.frombuffer(raw, dtype=np.uint8).reshape(m + 1, 32)
    assert np.array_equal(got, want)


def test_rejects_bad_sizes():
    from octopuszk_amd import lib, r1cs_to_qap as q
    z = bytes(32 * 3)
    for m in (0, 1, 3, 6):
        with pytest.raises(lib.OzkError):
            q.qap_witness_native_helper(z, z, z, m, o.to_le32(1), o.to_le32(5), 0)
