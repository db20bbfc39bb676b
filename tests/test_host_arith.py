"""CPU unit tests of the DEVICE arithmetic headers (fp29.cuh / ec.cuh) compiled for the
host, against the pure-Python oracle.  Covers the field ops, the lazy-reduction edge
values, and the group law incl. P+P, P+(-P), infinity (SURVEY.md §7 hard parts)."""
import ctypes
import os
import random
import subprocess

import pytest

from oracle import bn254 as o

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "hostcheck.cpp")
LIB = os.path.join(HERE, "native", "_hostcheck.so")


@pytest.fixture(scope="module")
def hc():
    deps = [SRC] + [os.path.join(HERE, "..", "octopuszk_amd", "csrc", f)
                    for f in ("fp29.cuh", "ec.cuh", "fq2.cuh", "glv.cuh", "msm_var.cuh", "curve.cuh", "consts_gen.h")]
    deps = [d for d in deps if os.path.exists(d)]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-shared", "-fPIC", "-o", LIB, SRC])
    return ctypes.CDLL(LIB)


def _w(v):
    return (ctypes.c_uint32 * 8)(*[(v >> (32 * i)) & 0xffffffff for i in range(8)])


def _r(buf, n=8):
    return sum(int(buf[i]) << (32 * i) for i in range(n))


def fe_op(hc, field, op, a, b=0):
    out = (ctypes.c_uint32 * 8)()
    getattr(hc, "hc_%s_op" % field)(op, _w(a), _w(b), out)
    return _r(out)


EDGE = lambda p: [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (1 << 253), (1 << 29) - 1, 1 << 29,
                  (1 << 232) - 1, 1 << 232, p >> 1, 3, (1 << 253) + 12345]


@pytest.mark.parametrize("field,p", [("fq", o.Q), ("fr", o.R)])
def test_field_ops(hc, field, p):
    rng = random.Random(7)
    vals = EDGE(p) + [rng.randrange(p) for _ in range(200)]
    for i, a in enumerate(vals):
        b = vals[(i * 7 + 3) % len(vals)]
        assert fe_op(hc, field, 0, a, b) == a * b % p
        assert fe_op(hc, field, 1, a) == a * a % p
        assert fe_op(hc, field, 2, a, b) == (a + b) % p
        assert fe_op(hc, field, 3, a, b) == (a - b) % p
        assert fe_op(hc, field, 5, a) == (-a) % p
        assert fe_op(hc, field, 6, a) == 8 * a % p
        assert fe_op(hc, field, 7, a, b) == (a - 3 * b) % p
        assert fe_op(hc, field, 8, a) == a
    for a in vals[:40]:
        want = pow(a, -1, p) if a % p else 0
        assert fe_op(hc, field, 9, a) == want      # Fermat


@pytest.mark.parametrize("field,p", [("fq", o.Q), ("fr", o.R)])
def test_reduce_q_one_subtraction_lands_below_17_16_p(hc, field, p):
    """fp29.cuh reduce_q: quotient estimated from the top limb, one table-row subtraction.  Inputs around every
    multiple of p and of (p >> 232) + 1 (where the estimate steps), the largest 256-bit values, random ones."""
    rng = random.Random(11)
    D = ((p >> 232) + 1) << 232
    vals = [0, 1, (1 << 256) - 1, (1 << 256) - 2, 1 << 255, (1 << 232) - 1, 1 << 232]
    for k in range(1, 6):
        for d in (-2, -1, 0, 1, 2):
            vals += [v for v in (k * p + d, k * D + d, k * D + d - (1 << 232)) if 0 <= v < (1 << 256)]
    vals += [rng.randrange(1 << 256) for _ in range(300)]
    fn = getattr(hc, "hc_%s_reduce_q" % field)
    for a in vals:
        for k in (1, 2, 4, 6, 7):
            out = (ctypes.c_uint32 * 8)()
            fn(k, _w(a), out)
            got = _r(out)
            assert got % p == k * a % p, (hex(a), k)
            assert got < p + (p >> 4), (hex(a), k, got / p)
            assert got < p + (48 << 232)          # the bound the derivation gives: (ptop + q + 1) 2^232


@pytest.mark.parametrize("field,p", [("fq", o.Q), ("fr", o.R)])
def test_safegcd_inversion(hc, field, p):
    # the divstep inversion against modular exponentiation, edge values, zero, non-canonical input
    rng = random.Random(8)
    vals = EDGE(p) + [p, p + 1, (1 << 256) - 1, (1 << 30) - 1, 1 << 30, (1 << 30) + 1, 1 << 60, (1 << 255) + 7]
    vals += [rng.randrange(p) for _ in range(2000)] + [rng.randrange(1 << b) for b in range(1, 255, 2)]
    for a in vals:
        want = pow(a % p, -1, p) if a % p else 0
        assert fe_op(hc, field, 4, a) == want, hex(a)
    for a in vals[:60]:
        b = vals[(7 * a + 3) % len(vals)] if False else vals[(vals.index(a) * 7 + 3) % len(vals)]
        big = (8 * a - b) % p
        assert fe_op(hc, field, 10, a, b) == (pow(big, -1, p) if big else 0)


def test_noncanonical_input_reduced(hc):
    # any 256-bit wire value is accepted and reduced mod p
    for a in [o.Q, o.Q + 5, (1 << 256) - 1, 2 * o.Q + 1]:
        assert fe_op(hc, "fq", 0, a, 1) == a % o.Q


def test_eq_mod_p(hc):
    assert hc.hc_fq_is_zero_after_sub(_w(5), _w(5)) == 1
    assert hc.hc_fq_is_zero_after_sub(_w(5), _w(6)) == 0
    assert hc.hc_fq_is_zero_after_sub(_w(0), _w(o.Q)) == 1


@pytest.mark.parametrize("field,p", [("fq", o.Q), ("fr", o.R)])
def test_loose_elements_equal_the_carried_computation(hc, field, p):
    """fp29.cuh FeL (round 3): sums and differences whose carry pass is left out, as factors of single, dual and
    quad products — same values as the normalised arithmetic, on the edge values that maximise the limbs."""
    rng = random.Random(17)
    vals = EDGE(p) + [p - 3, (1 << 254) % p, ((1 << 261) - 1) % p] + [rng.randrange(p) for _ in range(150)]
    for i, a in enumerate(vals):
        b = vals[(i * 11 + 5) % len(vals)]
        assert fe_op(hc, field, 20, a, b) == (a - b) * b % p
        assert fe_op(hc, field, 21, a, b) == (-a) * b % p
        assert fe_op(hc, field, 22, a, b) == (a * a - a * b - 2 * b * b) % p
        assert fe_op(hc, field, 23, a, b) == (a * a - b * b) % p
        assert fe_op(hc, field, 24, a, b) == 0
        assert fe_op(hc, field, 25, a, b) == ((a - b) * a - b * b) % p
        assert fe_op(hc, field, 26, a, b) == (2 * (a - b) + b) % p
        assert fe_op(hc, field, 27, a, b) == 0


def _pw(P):
    return (ctypes.c_uint32 * 24)(*[(c >> (32 * i)) & 0xffffffff for c in P for i in range(8)])


def g1_op(hc, op, P, Qp, k=0):
    out = (ctypes.c_uint32 * 24)()
    hc.hc_g1_op(op, _pw(P), _pw(Qp), k, out)
    return tuple(_r(out[8 * i:8 * i + 8]) for i in range(3))


def test_g1_group_law(hc):
    G = o.G1
    rng = random.Random(11)
    pts = [G.mul(G.one, rng.randrange(1, o.R)) for _ in range(12)]
    aff = [G.to_affine(P) for P in pts]
    for i in range(len(pts)):
        P, Qa = pts[i], aff[(i + 1) % len(pts)]
        assert G.equals(g1_op(hc, 0, P, pts[(i + 1) % len(pts)]), G.add(P, Qa))
        assert G.equals(g1_op(hc, 1, P, P), G.twice(P))
        assert G.equals(g1_op(hc, 2, P, Qa), G.add(P, Qa))
    P, Pa = pts[0], aff[0]
    # doubling and cancellation inside add / madd
    assert G.equals(g1_op(hc, 0, P, Pa), G.twice(P))
    assert G.equals(g1_op(hc, 2, P, Pa), G.twice(P))
    assert G.is_zero(g1_op(hc, 0, P, G.negate(Pa)))
    assert G.is_zero(g1_op(hc, 2, P, G.negate(Pa)))
    # infinity on either side
    assert G.equals(g1_op(hc, 0, G.zero, Pa), P)
    assert G.equals(g1_op(hc, 0, P, G.zero), P)
    assert G.equals(g1_op(hc, 2, G.zero, Pa), P)
    assert G.is_zero(g1_op(hc, 1, G.zero, G.zero))
    # long chains exercise the loop-carried lazy bounds
    assert G.equals(g1_op(hc, 3, P, aff[1], 200), G.add(P, G.mul(aff[1], 200)))
    assert G.equals(g1_op(hc, 4, P, pts[2], 100), G.add(P, G.mul(pts[2], 100)))
    assert G.equals(g1_op(hc, 5, P, P, 64), G.mul(P, 1 << 64))
    # repeated base: P + P + P ... hits the doubling branch on the first step
    assert G.equals(g1_op(hc, 3, Pa, Pa, 5), G.mul(P, 6))
    # XYZZ accumulator (the hot loop's coordinates): chains, doubling on the first step, cancellation
    assert G.equals(g1_op(hc, 7, Pa, aff[1], 200), G.add(P, G.mul(aff[1], 200)))
    assert G.equals(g1_op(hc, 7, Pa, Pa, 5), G.mul(P, 6))
    assert G.is_zero(g1_op(hc, 7, Pa, G.negate(Pa), 1))
    assert G.equals(g1_op(hc, 7, Pa, G.negate(Pa), 2), G.negate(P))
    assert G.equals(g1_op(hc, 7, G.zero, aff[2], 3), G.mul(aff[2], 3))
    assert G.equals(g1_op(hc, 7, Pa, G.zero, 4), P)
    # general XYZZ + XYZZ (level-1 in-wave merge): generic, doubling, cancellation, infinity
    Qa = aff[1]
    assert G.equals(g1_op(hc, 8, Pa, Qa, 7), G.mul(G.add(P, Qa), 8))
    assert G.equals(g1_op(hc, 9, Pa, Qa, 7), G.twice(G.add(P, G.mul(Qa, 7))))
    assert G.is_zero(g1_op(hc, 10, Pa, Qa, 7))
    assert G.equals(g1_op(hc, 11, Pa, Qa, 7), G.add(P, G.mul(Qa, 7)))
    assert G.equals(g1_op(hc, 8, Pa, Qa, 0), G.add(P, Qa))          # ZZ == 1 on both sides
    assert G.equals(g1_op(hc, 8, Pa, Pa, 3), G.mul(P, 8))           # A == B: doubling inside the add
    # the level-1 form (round 3): lazily carried mixed additions with the digit's sign
    for Pa_, Qa_ in ((Pa, aff[1]), (aff[4], aff[5])):
        Pj = Pa_
        assert G.equals(g1_op(hc, 12, Pa_, Qa_, 150), G.add(Pj, G.mul(Qa_, 150)))
        assert G.equals(g1_op(hc, 13, Pa_, Qa_, 150), G.add(Pj, G.negate(G.mul(Qa_, 150))))
        assert G.equals(g1_op(hc, 14, Pa_, Qa_, 7), G.add(Pj, Qa_))        # + - + - + - +
        assert G.equals(g1_op(hc, 14, Pa_, Qa_, 8), Pj)
    assert G.equals(g1_op(hc, 12, Pa, Pa, 5), G.mul(P, 6))                 # doubling on the first step
    assert G.equals(g1_op(hc, 13, Pa, G.negate(Pa), 5), G.mul(P, 6))       # ... through the negated y
    assert G.is_zero(g1_op(hc, 13, Pa, Pa, 1))                             # P - P
    assert G.equals(g1_op(hc, 13, Pa, Pa, 2), G.negate(P))                 # infinity, then a run start from -Q
    assert G.equals(g1_op(hc, 13, G.zero, aff[2], 3), G.negate(G.mul(aff[2], 3)))
    assert G.equals(g1_op(hc, 12, Pa, G.zero, 4), P)                       # infinity base
    # CurvesTest.java:27-82 identities on the HIP group law
    a = pts[3]
    assert G.equals(g1_op(hc, 0, G.mul(a, 76749407), G.mul(a, 44410867)), G.mul(a, 121160274))


def _w2(v):
    return (ctypes.c_uint32 * 16)(*[(c >> (32 * i)) & 0xffffffff for c in v for i in range(8)])


def fq2_op(hc, op, a, b=(0, 0)):
    out = (ctypes.c_uint32 * 16)()
    hc.hc_fq2_op(op, _w2(a), _w2(b), out)
    return (_r(out[0:8]), _r(out[8:16]))


def test_fq2_ops(hc):
    F = o.Fq2Ops
    rng = random.Random(21)
    vals = [(0, 0), (1, 0), (0, 1), (o.Q - 1, o.Q - 1), (o.Q - 1, 0), (5, o.Q - 2)]
    vals += [(rng.randrange(o.Q), rng.randrange(o.Q)) for _ in range(60)]
    for i, a in enumerate(vals):
        b = vals[(i * 5 + 1) % len(vals)]
        assert fq2_op(hc, 0, a, b) == F.mul(a, b)
        assert fq2_op(hc, 1, a) == F.sqr(a)
        assert fq2_op(hc, 2, a, b) == F.add(a, b)
        assert fq2_op(hc, 3, a, b) == F.sub(a, b)
        big = F.sub(F.add(F.add(a, a), F.add(a, a)), b)
        assert fq2_op(hc, 5, a, b) == F.mul(big, big)
        assert fq2_op(hc, 6, a, b) == F.sub(F.mul(a, big), F.mul(F.add(a, b), b))
        # lazily carried forms (fq2.cuh mul_lz / sqr_lz / mulsub_lz / sub_sub2)
        assert fq2_op(hc, 7, a, b) == F.mul(a, b)
        assert fq2_op(hc, 8, a) == F.sqr(a)
        assert fq2_op(hc, 9, a, b) == F.sub(F.mul(F.add(a, b), F.sub(a, b)), F.mul(F.sub(F.add(a, a), b), b))
        assert fq2_op(hc, 10, a, b) == F.sub(F.sub(F.sqr(a), F.mul(a, b)), F.add(F.sqr(b), F.sqr(b)))
        if a != (0, 0):
            assert fq2_op(hc, 4, a) == F.inv(a)


def _pw2(P):
    return (ctypes.c_uint32 * 48)(*[(c >> (32 * i)) & 0xffffffff for co in P for c in co for i in range(8)])


def g2_op(hc, op, P, Qp, k=0):
    out = (ctypes.c_uint32 * 48)()
    hc.hc_g2_op(op, _pw2(P), _pw2(Qp), k, out)
    v = [_r(out[8 * i:8 * i + 8]) for i in range(6)]
    return ((v[0], v[1]), (v[2], v[3]), (v[4], v[5]))


def test_g2_group_law(hc):
    G = o.G2
    rng = random.Random(12)
    pts = [G.mul(G.one, rng.randrange(1, o.R)) for _ in range(6)]
    aff = [G.to_affine(P) for P in pts]
    for i in range(len(pts)):
        P, Qa = pts[i], aff[(i + 1) % len(pts)]
        assert G.equals(g2_op(hc, 0, P, pts[(i + 1) % len(pts)]), G.add(P, Qa))
        assert G.equals(g2_op(hc, 1, P, P), G.twice(P))
        assert G.equals(g2_op(hc, 2, P, Qa), G.add(P, Qa))
    P, Pa = pts[0], aff[0]
    assert G.equals(g2_op(hc, 0, P, Pa), G.twice(P))
    assert G.equals(g2_op(hc, 2, P, Pa), G.twice(P))
    assert G.is_zero(g2_op(hc, 0, P, G.negate(Pa)))
    assert G.is_zero(g2_op(hc, 2, P, G.negate(Pa)))
    assert G.equals(g2_op(hc, 0, G.zero, Pa), P)
    assert G.equals(g2_op(hc, 2, G.zero, Pa), P)
    assert G.equals(g2_op(hc, 3, P, aff[1], 100), G.add(P, G.mul(aff[1], 100)))
    assert G.equals(g2_op(hc, 4, P, pts[2], 50), G.add(P, G.mul(pts[2], 50)))
    assert G.equals(g2_op(hc, 5, P, P, 40), G.mul(P, 1 << 40))
    assert G.equals(g2_op(hc, 7, Pa, aff[1], 60), G.add(P, G.mul(aff[1], 60)))
    assert G.equals(g2_op(hc, 7, Pa, Pa, 3), G.mul(P, 4))
    assert G.is_zero(g2_op(hc, 7, Pa, G.negate(Pa), 1))
    Qa = aff[1]
    assert G.equals(g2_op(hc, 8, Pa, Qa, 5), G.mul(G.add(P, Qa), 6))
    assert G.equals(g2_op(hc, 9, Pa, Qa, 5), G.twice(G.add(P, G.mul(Qa, 5))))
    assert G.is_zero(g2_op(hc, 10, Pa, Qa, 5))
    assert G.equals(g2_op(hc, 11, Pa, Qa, 5), G.add(P, G.mul(Qa, 5)))
    # the level-1 schedule over the lazily carried Fq2 products (round 3)
    assert G.equals(g2_op(hc, 12, Pa, aff[1], 80), G.add(P, G.mul(aff[1], 80)))
    assert G.equals(g2_op(hc, 12, Pa, Pa, 3), G.mul(P, 4))
    assert G.is_zero(g2_op(hc, 12, Pa, G.negate(Pa), 1))
    assert G.equals(g2_op(hc, 12, G.zero, aff[2], 4), G.mul(aff[2], 4))
    assert G.equals(g2_op(hc, 8, Pa, Pa, 2), G.mul(P, 6))
