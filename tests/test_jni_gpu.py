"""The six JNI natives driven through a mock JNIEnv on the GPU: same bytes as the oracle."""
import random

import pytest

from oracle import bn254 as o
import jni_util as ju

pytestmark = pytest.mark.gpu


def _pts(C, n, rng):
    return [C.to_affine(C.mul(C.one, rng.randrange(1, 1 << 64))) for _ in range(n)]


def test_jni_var_msm_g1_g2_double():
    rng = random.Random(42)
    n = 33
    b1, b2 = _pts(o.G1, n, rng), _pts(o.G2, n, rng)
    sc = [rng.randrange(o.R) for _ in range(n)]
    w1 = b"".join(o.g1_to_wire(P) for P in b1)
    w2 = b"".join(o.g2_to_wire(P) for P in b2)
    ws = b"".join(o.to_le32(s) for s in sc)
    e1 = o.g1_out_le(o.G1.to_affine(o.naive_msm(o.G1, sc, b1)))
    e2 = o.g2_out_le(o.G2.to_affine(o.naive_msm(o.G2, sc, b2)))
    assert ju.var_msm(w1, ws, n, 1) == e1
    assert ju.var_msm(w2, ws, n, 2) == e2
    assert ju.var_double_msm(w1, w2, ws, n) == e1 + e2
    assert ju.var_msm(w1, ws, n, 1, task=7) == e1   # taskID % num_gpus


def test_jni_fixed_base_and_field():
    rng = random.Random(43)
    sc = [0, 1, o.R - 1] + [rng.randrange(o.R) for _ in range(10)]
    ws = b"".join(o.to_le32(s) for s in sc)
    B1 = o.G1.mul(o.G1.one, 777)
    B2 = o.G2.to_affine(o.G2.mul(o.G2.one, 888))
    w = 7
    oc = (254 + w - 1) // w
    got = ju.fixed_batch(oc, w, len(sc), 254, o.g1_to_wire(B1), ws, 1)
    assert got == b"".join(o.g1_out_be(o.G1.to_affine(o.G1.mul(B1, s))) for s in sc)
    got = ju.fixed_batch(oc, w, len(sc), 254, o.g2_to_wire(B2), ws, 2)
    assert got == b"".join(o.g2_out_be(o.G2.to_affine(o.G2.mul(B2, s))) for s in sc)
    got = ju.fixed_double_batch(oc, w, oc, w, len(sc), o.g1_to_wire(B1), o.g2_to_wire(B2), ws)
    assert got == b"".join(o.g1_out_be(o.G1.to_affine(o.G1.mul(B1, s))) + o.g2_out_be(o.G2.to_affine(o.G2.mul(B2, s)))
                           for s in sc)
    m = rng.randrange(o.R)
    got = ju.field_mul(ws + o.to_le32(m), len(sc))
    assert got == b"".join(int(s * m % o.R).to_bytes(64, "big") for s in sc)


def test_jni_fft_list_walk():
    rng = random.Random(44)
    n = 256
    a = [rng.randrange(o.R) for _ in range(n)]
    a[3], a[4] = 0, 1                       # short encodings (4 bytes) in the List<byte[]>
    w = o.fr_root_of_unity(n)
    got, refs = ju.fft([o.to_fft_bytes(x) for x in a], o.to_fft_bytes(w))
    b = list(a)
    o.serial_radix2_fft(b, w)
    assert got == b"".join(int(x).to_bytes(64, "little") for x in b)
    assert refs == n                         # every element's local reference was deleted


def test_jni_optional_qap_witness_native():
    # not a native of the reference: the optional binding of INTEGRATION.md §5
    rng = random.Random(45)
    m = 128
    a = [rng.randrange(o.R) for _ in range(m)]
    b = [rng.randrange(o.R) for _ in range(m)]
    c = [x * y % o.R for x, y in zip(a, b)]
    enc = lambda v: b"".join(o.to_le32(x) for x in v)
    got = ju.qap_witness(enc(a), enc(b), enc(c), m, o.to_fft_bytes(o.fr_root_of_unity(m)), o.to_fft_bytes(o.FR_MULT_GEN))
    want = o.qap_witness_coefficients_h(a, b, c)
    assert got == enc(want)
    with pytest.raises(ju.JavaException):
        ju.qap_witness(enc(a)[:-32], enc(b), enc(c), m, o.to_le32(o.fr_root_of_unity(m)), o.to_le32(5))
    with pytest.raises(ju.JavaException):
        ju.qap_witness(enc(a), enc(b), enc(c), 96, o.to_le32(1), o.to_le32(5))


def test_jni_optional_prepared_bases_natives():
    # prepareBases / ...PreparedNativeHelper / releaseBases: same bytes as the plain native
    rng = random.Random(46)
    n = 200
    for type_, C, wire in ((1, o.G1, o.g1_to_wire), (2, o.G2, o.g2_to_wire)):
        pts = _pts(C, n if type_ == 1 else 60, rng)
        k = len(pts)
        bw = b"".join(wire(P) for P in pts)
        sw = b"".join(o.to_le32(rng.randrange(o.R)) for _ in range(k))
        assert ju.prepared_msm(bw, sw, k, type_) == ju.var_msm(bw, sw, k, type_)
    with pytest.raises(ju.JavaException):
        ju.prepared_msm(bw[:-8], sw, k, 2)          # short bases array -> exception from prepare
    with pytest.raises(ju.JavaException):
        ju.prepared_msm(bw, sw[:-32], k, 2)         # short scalars -> exception from the MSM call


def test_jni_optional_compact_natives():
    # batchMSMCompactNativeHelper / serialRadix2FFTFlatNativeHelper: the
    # values of the reference-format natives in 32-byte little-endian elements; the fixed-base output is the
    # variable-base natives' wire-in format, so it feeds variableBaseSerialMSMNativeHelper as it is
    rng = random.Random(47)
    sc = [0, 1, o.R - 1] + [rng.randrange(o.R) for _ in range(20)]
    ws = b"".join(o.to_le32(s) for s in sc)
    w = 6
    oc = (254 + w - 1) // w
    for bn, C, wire in ((1, o.G1, o.g1_to_wire), (2, o.G2, o.g2_to_wire)):
        B = C.mul(C.one, 4242 + bn)
        got = ju.fixed_batch_compact(oc, w, len(sc), wire(B), ws, bn)
        assert got == b"".join(wire(C.to_affine(C.mul(B, s))) for s in sc)
        # ... and straight into the variable-base native: sum_i t_i (s_i B) = (sum t_i s_i) B
        ts = [rng.randrange(o.R) for _ in sc]
        tw = b"".join(o.to_le32(t) for t in ts)
        acc = sum(t * s for t, s in zip(ts, sc)) % o.R
        want = C.to_affine(C.mul(B, acc))
        assert ju.var_msm(got, tw, len(sc), bn) == (o.g1_out_le(want) if bn == 1 else o.g2_out_le(want))
    with pytest.raises(ju.JavaException):
        ju.fixed_batch_compact(oc, w, len(sc), o.g1_to_wire(o.G1.one), ws[:-1], 1)
    n = 512
    a = [rng.randrange(o.R) for _ in range(n)]
    om = o.fr_root_of_unity(n)
    b = list(a)
    o.serial_radix2_fft(b, om)
    flat = b"".join(o.to_le32(x) for x in a)
    assert ju.fft_flat(flat, n, o.to_fft_bytes(om)) == b"".join(o.to_le32(x) for x in b)
    with pytest.raises(ju.JavaException):
        ju.fft_flat(flat, 384, o.to_le32(om))
    with pytest.raises(ju.JavaException):
        ju.fft_flat(flat[:-32], n, o.to_le32(om))
