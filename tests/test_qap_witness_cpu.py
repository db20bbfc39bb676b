"""Oracle restatement of the QAP witness map (R1CStoQAP.java:163-230) pinned by what the reference's
own check verifies (QAPRelation.isSatisfied, relations/qap/QAPRelation.java:95-130):
A(t) B(t) - C(t) = H(t) Z(t) at a random point, for evaluations that satisfy a_i b_i = c_i on the
domain — plus an independent polynomial-division model on a tiny domain."""
import random

from oracle import bn254 as o


def _poly_eval(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % o.R
    return acc


def _satisfied_evals(m, rng):
    a = [rng.randrange(o.R) for _ in range(m)]
    b = [rng.randrange(o.R) for _ in range(m)]
    a[3 % m] = 0
    b[5 % m] = 1
    return a, b, [x * y % o.R for x, y in zip(a, b)]


def test_h_times_z_equals_ab_minus_c_at_random_point():
    rng = random.Random(41)
    for m in (2, 4, 8, 64, 256):
        a, b, c = _satisfied_evals(m, rng)
        H = o.qap_witness_coefficients_h(a, b, c)
        assert len(H) == m + 1 and H[m] == 0 and H[m - 1] == 0   # deg H <= m - 2
        # coefficients of A, B, C: the inverse transforms of the evaluations
        ca, cb, cc = list(a), list(b), list(c)
        for v in (ca, cb, cc):
            o.radix2_inverse_fft(v)
        t = rng.randrange(o.R)
        lhs = (_poly_eval(ca, t) * _poly_eval(cb, t) - _poly_eval(cc, t)) % o.R
        assert lhs == _poly_eval(H, t) * o.compute_z(t, m) % o.R


def test_against_schoolbook_division_on_a_tiny_domain():
    rng = random.Random(42)
    m = 8
    a, b, c = _satisfied_evals(m, rng)
    ca, cb, cc = list(a), list(b), list(c)
    for v in (ca, cb, cc):
        o.radix2_inverse_fft(v)
    prod = [0] * (2 * m - 1)
    for i, x in enumerate(ca):
        for j, y in enumerate(cb):
            prod[i + j] = (prod[i + j] + x * y) % o.R
    for i, z in enumerate(cc):
        prod[i] = (prod[i] - z) % o.R
    # divide by x^m - 1: q_k = p_{k+m} + q_{k+m}
    q = [0] * (m - 1)
    for k in range(m - 2, -1, -1):
        q[k] = (prod[k + m] + (q[k + m] if k + m < m - 1 else 0)) % o.R
    rem = [(prod[k] + (q[k] if k < m - 1 else 0)) % o.R for k in range(m)]
    assert rem == [0] * m
    assert o.qap_witness_coefficients_h(a, b, c) == q + [0, 0]


def test_unsatisfied_inputs_are_still_a_function():
    # for arbitrary A, B, C the map is the degree-< m interpolation of (AB - C)/Z on the coset
    rng = random.Random(43)
    m = 16
    a = [rng.randrange(o.R) for _ in range(m)]
    b = [rng.randrange(o.R) for _ in range(m)]
    c = [rng.randrange(o.R) for _ in range(m)]
    H = o.qap_witness_coefficients_h(a, b, c)
    g = o.FR_MULT_GEN
    ca, cb, cc = list(a), list(b), list(c)
    for v in (ca, cb, cc):
        o.radix2_inverse_fft(v)
    w = o.fr_root_of_unity(m)
    for i in range(m):
        x = g * pow(w, i, o.R) % o.R
        want = (_poly_eval(ca, x) * _poly_eval(cb, x) - _poly_eval(cc, x)) * pow(o.compute_z(x, m), -1, o.R) % o.R
        assert _poly_eval(H, x) == want
