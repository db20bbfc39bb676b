"""CPU tests of the Groth16 caller layer: the oracle's restatement of serial setup + prove
is pinned by the Groth16 verification equation taken to the exponent (the reference's own acceptance test is
Verifier.verify == true, SerialzkSNARKTest.java:76-92, which needs a pairing; every key element here is a known
multiple of the generators, so the same equation can be checked in Fr), and the product's host-side mirror
(octopuszk_amd/zksnark.py: R1CS construction, constraint evaluation, QAP instance) equals the oracle's."""
import pytest

from oracle import bn254 as o
from oracle import groth16 as g


@pytest.mark.parametrize("nc,ni", [(8, 3), (32, 7), (45, 1)])
def test_oracle_prove_verifies_in_the_exponent(nc, ni):
    r1cs, primary, auxiliary = g.serial_construct(nc, ni)
    assert g.is_satisfied(r1cs, primary, auxiliary)          # R1CSConstruction.java:106
    crs = g.serial_setup(r1cs)
    (A, B, C), info = g.serial_prove(crs, primary, auxiliary)
    # SerialProver.java:44-55 (debug asserts): H has degree d - 2
    H, m = info["H"], info["m"]
    assert H[m - 2] != 0 and H[m - 1] == 0 and H[m] == 0
    a, b, c = g.proof_scalars(crs, info["full"], H, info["r"], info["s"])
    assert o.G1.equals(A, o.G1.mul(crs.gen_g1, a))
    assert o.G2.equals(B, o.G2.mul(crs.gen_g2, b))
    assert o.G1.equals(C, o.G1.mul(crs.gen_g1, c))
    assert g.verify_in_the_exponent(crs, primary, (a, b, c))
    # a wrong witness must not verify
    bad = list(info["full"])
    bad[ni + 1] = (bad[ni + 1] + 1) % o.R
    a2, b2, c2 = g.proof_scalars(crs, bad, H, info["r"], info["s"])
    assert not g.verify_in_the_exponent(crs, primary, (a2, b2, c2))


def test_oracle_qap_relation_is_satisfied_by_the_witness():
    # QAPRelation.isSatisfied (relations/qap/QAPRelation.java): A(t) B(t) - C(t) = H(t) Z(t)
    r1cs, primary, auxiliary = g.serial_construct(20, 4)
    t = g.fr_random()
    q = g.r1cs_to_qap_relation(r1cs, t)
    full, H, m, _ = g.r1cs_to_qap_witness(r1cs, primary, auxiliary)
    R = o.R
    at = sum(z * x for z, x in zip(full, q.At)) % R
    bt = sum(z * x for z, x in zip(full, q.Bt)) % R
    ct = sum(z * x for z, x in zip(full, q.Ct)) % R
    ht = sum(h * x for h, x in zip(H, q.Ht)) % R
    assert (at * bt - ct - ht * q.Zt) % R == 0


def test_oracle_window_sizes_follow_the_reference_tables():
    # SerialSetup.java:91-112 at the profiler's size: 2^20 constraints, 1023 inputs -> both windows 17
    nv = (1 << 20) + 3
    assert o.fixed_base_window_size(3 * nv, o.G1_FIXED_BASE_WINDOW_TABLE) == 17
    assert o.fixed_base_window_size(nv, o.G2_FIXED_BASE_WINDOW_TABLE) == 17


@pytest.mark.parametrize("nc,ni", [(8, 3), (33, 7), (600, 15)])
def test_product_host_mirror_equals_oracle(nc, ni):
    from octopuszk_amd import zksnark as z
    r1, p1, a1 = z.serial_construct(nc, ni)
    r2, p2, a2 = g.serial_construct(nc, ni)
    assert p1 == p2 and a1 == a2
    assert z.is_satisfied(r1, p1, a1)
    ev, m = z.constraint_evaluations(r1, p1 + a1)
    _, _, m2, (A, B, C) = g.r1cs_to_qap_witness(r2, p2, a2)
    assert m == m2
    assert [int(x) for x in ev[0]] == A and [int(x) for x in ev[1]] == B and [int(x) for x in ev[2]] == C
    t = z.fr_random()
    assert t == g.fr_random() == (-4972683369271453960) % o.R   # new Random(10).nextLong()
    q1, q2 = z.r1cs_to_qap_relation(r1, t), g.r1cs_to_qap_relation(r2, t)
    assert (q1.At, q1.Bt, q1.Ct, q1.Ht, q1.Zt, q1.degree) == (q2.At, q2.Bt, q2.Ct, q2.Ht, q2.Zt, q2.degree)


def test_product_lagrange_on_a_domain_point():
    # FFTAuxiliary.java:270-281: t in the domain -> indicator vector
    from octopuszk_amd import zksnark as z
    m = 16
    w = z.root_of_unity(m)
    assert z.lagrange_coefficients(pow(w, 5, o.R), m) == g.lagrange_coefficients(pow(w, 5, o.R), m)
    assert z.lagrange_coefficients(pow(w, 5, o.R), m)[5] == 1
