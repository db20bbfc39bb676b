"""Pin the oracles: every KAT / property the reference's own tests hold for the hot path
(SURVEY.md §8c), the public 2*G1 KAT, the Java LCG, and C oracle == Python oracle."""
import random

import pytest

from oracle import bn254 as o
from oracle import coracle
from oracle.javarand import JavaRandom, fp_random

G1, G2 = o.G1, o.G2


def test_java_random_seed_10():
    # Configuration.java:52 default seed; SURVEY §8c: new Random(10).nextLong()
    assert JavaRandom(10).next_long() == -4972683369271453960
    assert fp_random(10, o.R) == (-4972683369271453960) % o.R


def test_public_kat_2g():
    x, y, z = G1.to_affine(G1.twice(G1.one))
    assert x == 1368015179489954701390400359078579693043519447331113978918064868415326638035
    assert y == 9918110051302171585080402603319702774565515993150576347155970296011118125764
    assert z == 1


def test_generators_on_curve_and_order():
    assert G1.on_curve(G1.one) and G2.on_curve(G2.one)
    assert G1.is_zero(G1.mul(G1.one, o.R)) and G2.is_zero(G2.mul(G2.one, o.R))


@pytest.mark.parametrize("C", [G1, G2], ids=["G1", "G2"])
def test_curvestest_group_identities(C):
    # src/test/java/algebra/curves/CurvesTest.java:27-82 (GroupTest)
    zero, one = C.zero, C.one
    two = C.add(one, one)
    three, four, five = C.add(two, one), C.add(two, two), C.add(C.add(two, two), one)
    assert C.equals(C.add(two, five), C.add(three, four))
    a = C.mul(one, 0x1234567890abcdef1234567)
    b = C.mul(one, 987654321987654321)
    assert C.is_zero(C.add(a, C.negate(a)))
    assert C.equals(C.twice(a), C.add(a, a))
    assert C.equals(C.add(a, b), C.add(b, a))
    assert C.equals(C.add(C.add(a, b), one), C.add(a, C.add(b, one)))
    assert C.equals(C.add(a, zero), a) and C.equals(C.add(zero, a), a)
    assert C.equals(C.add(C.mul(a, 76749407), C.mul(a, 44410867)), C.mul(a, 121160274))
    assert C.is_zero(C.twice(zero)) and C.is_zero(C.mul(a, o.R))
    assert C.equals(C.to_affine(a), a)


@pytest.mark.parametrize("C", [G1, G2], ids=["G1", "G2"])
def test_msm_toy_kat(C):
    # SerialVariableBaseMSMTest.java:31-77: (3,11,2,8).(5,2,7,3) = 75 — here on BN254 as 75*G
    bases = [C.mul(C.one, k) for k in (5, 2, 7, 3)]
    sc = [3, 11, 2, 8]
    want = C.mul(C.one, 75)
    for f in (o.naive_msm, o.pippenger_msm, o.sorted_msm, o.filtered_msm):
        assert C.equals(f(C, sc, bases), want)
    # DistributedVariableBaseMSMTest.java:92-124: four x (3 * 5) = 60
    assert C.equals(o.pippenger_msm(C, [3] * 4, [C.mul(C.one, 5)] * 4), C.mul(C.one, 60))


def test_fixed_base_window_kat():
    # SerialFixedBaseMSMTest.java:55-72 shape: base 7G, windowSize 2; table rows are i * 2^(2w) * B
    B = G1.mul(G1.one, 7)
    table = o.fixed_base_window_table(G1, B, 8, 2)
    assert len(table) == 4 and len(table[0]) == 4
    assert G1.equals(table[1][3], G1.mul(B, 3 * 4))
    for s in (0, 1, 200, 255):
        assert G1.equals(o.fixed_base_serial_msm(G1, 8, 2, table, s), G1.mul(B, s))
        assert G1.equals(o.fixed_base_mul(G1, B, 8, 2, s), G1.mul(B, s))
    assert o.fixed_base_window_size(1 << 20, o.G1_FIXED_BASE_WINDOW_TABLE) == 17
    assert o.fixed_base_window_size(1 << 23, o.G1_FIXED_BASE_WINDOW_TABLE) == 20


def test_fft_kat_m4():
    # SerialFFTTest.java:168-190: m = 4, input (2,5,3,8) == polynomial evaluation at omega^i
    a = [2, 5, 3, 8]
    w = o.fr_root_of_unity(4)
    b = list(a)
    o.serial_radix2_fft(b, w)
    assert b == o.naive_dft(a, w)


def test_fft_inverse_and_coset_roundtrip():
    # SerialFFTTest.java:192-264 (commented there) / DistributedFFTTest.java:70-172 properties
    rng = random.Random(3)
    a = [rng.randrange(o.R) for _ in range(64)]
    b = list(a)
    o.radix2_fft(b)
    assert b == o.naive_dft(a, o.fr_root_of_unity(64))
    o.radix2_inverse_fft(b)
    assert b == a
    o.radix2_coset_fft(b, o.FR_MULT_GEN)
    o.radix2_coset_inverse_fft(b, o.FR_MULT_GEN)
    assert b == a
    w22 = o.fr_root_of_unity(1 << 22)
    assert pow(w22, 1 << 22, o.R) == 1 and pow(w22, 1 << 21, o.R) != 1


def test_bitreverse_and_log2():
    assert [o.bitreverse(k, 3) for k in range(8)] == [0, 4, 2, 6, 1, 5, 3, 7]
    assert all(o.java_log2(1 << k) == k for k in range(1, 31))
    assert o.pippenger_window(1 << 16) == 11 and o.pippenger_window(1 << 20) == 14


def test_wire_codec():
    v = o.Q - 5
    assert o.to_le32(v)[::-1] == v.to_bytes(32, "big")
    assert o.to_fft_bytes(0) == b"\x00" * 4
    assert len(o.to_fft_bytes(o.R - 1)) == 32
    assert len(o.to_fft_bytes((1 << 248) - 1)) == 32  # BigInteger.toByteArray adds a sign byte
    P = G1.to_affine(G1.mul(G1.one, 99))
    assert o.g1_from_out_le(o.g1_out_le(P)) == P
    assert o.g1_from_out_be(o.g1_out_be(P)) == P


# ---------------------------------------------------------------- C oracle == Python oracle
def test_c_field_ops_match_python():
    rng = random.Random(1)
    for fi, p in ((0, o.Q), (1, o.R)):
        for _ in range(200):
            a, b = rng.randrange(p), rng.randrange(p)
            assert coracle.field_op(fi, 0, a, b) == a * b % p
            assert coracle.field_op(fi, 1, a, b) == (a + b) % p
            assert coracle.field_op(fi, 2, a, b) == (a - b) % p
        for a in (1, 2, p - 1, rng.randrange(p)):
            assert coracle.field_op(fi, 3, a, 0) == pow(a, -1, p)


@pytest.mark.parametrize("n", [1, 2, 3, 33, 300])
def test_c_pippenger_matches_python(n):
    rng = random.Random(n)
    bases = [G1.mul(G1.one, rng.randrange(1, 1 << 30)) for _ in range(n)]  # Jacobian, Z != 1
    bases[0] = G1.to_affine(bases[0])
    if n > 2:
        bases[2] = G1.zero
    sc = [rng.randrange(o.R) for _ in range(n)]
    sc[0] = 0 if n > 1 else sc[0]
    want = o.g1_out_le(G1.to_affine(o.pippenger_msm(G1, sc, bases)))
    bw = b"".join(o.g1_to_wire(P) for P in bases)
    sw = b"".join(o.to_le32(s) for s in sc)
    assert coracle.pippenger_g1(bw, sw, n) == want
    if n <= 33:
        assert coracle.naive_g1(bw, sw, n) == want


def test_c_fft_and_field_mul_match_python():
    rng = random.Random(2)
    for n in (1, 2, 4, 64, 1024):
        a = [rng.randrange(o.R) for _ in range(n)]
        w = o.fr_root_of_unity(n) if n > 1 else 1
        b = list(a)
        o.serial_radix2_fft(b, w)
        got = coracle.fft_fr(b"".join(o.to_le32(x) for x in a), n, o.to_le32(w))
        assert got == b"".join(int(x).to_bytes(64, "little") for x in b)
    xs = [rng.randrange(o.R) for _ in range(20)]
    m = rng.randrange(o.R)
    got = coracle.field_batch_mul(b"".join(o.to_le32(x) for x in xs + [m]), 20)
    assert got == b"".join(int(x * m % o.R).to_bytes(64, "big") for x in xs)


def test_c_fixed_base_matches_python():
    rng = random.Random(4)
    B = G1.mul(G1.one, 31337)
    sc = [0, 1, o.R - 1] + [rng.randrange(o.R) for _ in range(5)]
    for window in (2, 5, 17):
        outerc = (254 + window - 1) // window
        got = coracle.fixed_base_g1(o.g1_to_wire(B), b"".join(o.to_le32(s) for s in sc), len(sc), outerc, window)
        for i, s in enumerate(sc):
            want = G1.to_affine(o.fixed_base_mul(G1, B, 254, window, s))
            assert got[192 * i:192 * (i + 1)] == o.g1_out_be(want)


# --------------------------------------------------------------------------
# The reference's own KATs, VERBATIM (its toy group and its test field), through the same generic
# oracle functions that are used on BN254 — the literal pin of VERDICT r1 "weak" 2.
# --------------------------------------------------------------------------
class _ToyGroup:
    """algebra/groups/AdditiveIntegerGroup.java over LargeAdditiveIntegerGroupParameters
    (integers mod 143987564266532958): add, twice, zero; mul is AbstractGroup's double-and-add."""
    MOD = 143987564266532958
    zero = 0
    one = 1

    def add(self, a, b):
        return (a + b) % self.MOD

    def twice(self, a):
        return (a + a) % self.MOD

    def is_zero(self, a):
        return a % self.MOD == 0

    def equals(self, a, b):
        return a % self.MOD == b % self.MOD

    mul = o.Curve.mul   # AbstractGroup.java:29-51, generic over add / twice / zero


def test_reference_toy_group_kat_literal():
    T = _ToyGroup()
    sc, bases = [3, 11, 2, 8], [5, 2, 7, 3]
    # SerialVariableBaseMSMTest.java:31-77 (Naive / Sorted / BosCoster expect 75),
    # DistributedVariableBaseMSMTest.java:92-109 (distributedMSM expects 75)
    for f in (o.naive_msm, o.pippenger_msm, o.sorted_msm, o.bos_coster_msm, o.filtered_msm):
        assert f(T, sc, bases) == 75, f.__name__
    # DistributedVariableBaseMSMTest.java:111-124: four x (3 * 5) = 60
    for f in (o.naive_msm, o.pippenger_msm, o.sorted_msm, o.bos_coster_msm, o.filtered_msm):
        assert f(T, [3] * 4, [5] * 4) == 60, f.__name__
    # a scalar as wide as BN254's: the windows of pippengerMSM cover 254 bits whatever the group
    big = [o.R - 1, 1 << 200, 12345678901234567890, 7]
    bs = [17, 9, 1 << 50, 143987564266532957]
    want = sum(s * b for s, b in zip(big, bs)) % T.MOD
    assert o.pippenger_msm(T, big, bs) == o.naive_msm(T, big, bs) == want
    assert o.bos_coster_msm(T, big, bs) == want   # (R - 1) div 2^200 < 2^20 < 2^200 div 12345678901234567890: both branches


def test_bos_coster_on_bn254_equals_naive_and_pippenger():
    # VariableBaseMSM.bosCosterMSM (VariableBaseMSM.java:86-119) on the curve itself: positive scalars of mixed sizes,
    # so that the subtract branch, the 2^20-ratio branch and the one-pair tail (:113-116) all run
    import random
    rnd = random.Random(7)
    for G in (o.G1, o.G2):
        pts = [G.to_affine(G.mul(G.one, rnd.randrange(1, 1 << 40))) for _ in range(9)]
        sc = [rnd.randrange(1, o.R) for _ in range(5)] + [1, 3, rnd.randrange(1, 1 << 64), rnd.randrange(1, 1 << 200)]
        want = G.to_affine(o.naive_msm(G, sc, pts))
        assert G.to_affine(o.bos_coster_msm(G, sc, pts)) == want
        assert G.to_affine(o.pippenger_msm(G, sc, pts)) == want
    assert o.bos_coster_msm(o.G1, [], []) == o.G1.zero
    assert o.G1.to_affine(o.bos_coster_msm(o.G1, [5], [o.G1.one])) == o.G1.to_affine(o.G1.mul(o.G1.one, 5))


def test_reference_fft_kat_over_large_fp_literal():
    # SerialFFTTest.java:168-190 (PrimeFieldSerialFFTTest) over LargeFpParameters: the 181-bit modulus and
    # root 6 of algebra/fields/fieldparameters/LargeFpParameters.java:30-45; omega = root^(p div 4) (Fp.java:98-102)
    p = 1532495540865888858358347027150309183618765510462668801
    omega = pow(6, p // 4, p)
    a = [2, 5, 3, 8]
    b = list(a)
    o.serial_radix2_fft(b, omega, p)
    assert b == o.naive_dft(a, omega, p)
    assert b == [18, (2 + 5 * omega + 3 * omega ** 2 + 8 * omega ** 3) % p, (2 - 5 + 3 - 8) % p,
                 (2 + 5 * omega ** 3 + 3 * omega ** 6 + 8 * omega ** 9) % p]
