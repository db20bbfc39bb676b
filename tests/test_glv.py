"""GLV endomorphism (octopuszk_amd/csrc/glv.cuh): the decomposition compiled for the host
against an exact big-integer model, the 127-bit bound the halved window count relies on, and
the endomorphism identity phi(P) = lambda P against the oracle's group law on G1 and G2."""
import ctypes
import random

from oracle import bn254 as o
from test_host_arith import hc, _w, _r  # noqa: F401  (fixture + word helpers)

LAM = 4407920970296243842393367215006156084916469457145843978461
A1 = 9931322734385697763
B1 = -147946756881789319000765030803803410728
A2 = 147946756881789319010696353538189108491
B2 = 9931322734385697763


def model(k):
    k %= o.R
    g1 = (B2 << 256) // o.R
    g2 = ((-B1) << 256) // o.R
    c1 = (k * g1) >> 256
    c2 = (k * g2) >> 256
    return k - c1 * A1 - c2 * A2, -c1 * B1 - c2 * B2


def decompose(hc, k):
    out = (ctypes.c_uint32 * 10)()
    hc.hc_glv(_w(k), out)
    k1, k2 = _r(out[0:4], 4), _r(out[4:8], 4)
    return (-k1 if out[8] else k1), (-k2 if out[9] else k2)


def test_constants():
    assert (LAM * LAM + LAM + 1) % o.R == 0
    assert (A1 + B1 * LAM) % o.R == 0 and (A2 + B2 * LAM) % o.R == 0 and A1 * B2 - A2 * B1 == o.R


def test_decompose_matches_model_and_bound(hc):
    rng = random.Random(5)
    ks = [0, 1, 2, o.R - 1, o.R - 2, o.R, o.R + 1, (1 << 256) - 1, LAM, LAM - 1, LAM + 1, o.R - LAM,
          1 << 127, (1 << 127) - 1, 1 << 128, (1 << 253), 5 * o.R + 3, A1, A2, o.R // 2, o.R // 3]
    ks += [rng.randrange(1 << 256) for _ in range(3000)] + [rng.randrange(o.R) for _ in range(3000)]
    ks += [rng.randrange(1 << b) for b in range(1, 256, 3) for _ in range(4)]
    worst = 0
    for k in ks:
        k1, k2 = decompose(hc, k)
        assert (k1, k2) == model(k), hex(k)
        assert (k1 + k2 * LAM - k) % o.R == 0
        worst = max(worst, abs(k1), abs(k2))
    assert worst < 1 << 127


def test_endomorphism_identity(hc):
    out = (ctypes.c_uint32 * 16)()
    hc.hc_glv_beta(out)
    b1, b2 = _r(out[0:8]), _r(out[8:16])
    assert (b1 * b1 + b1 + 1) % o.Q == 0 and b2 == b1 * b1 % o.Q
    P = o.G1.to_affine(o.G1.mul(o.G1.one, 0xdecafbad12345))
    assert o.G1.equals((b1 * P[0] % o.Q, P[1], 1), o.G1.mul(P, LAM))
    Q2 = o.G2.to_affine(o.G2.mul(o.G2.one, 0xfeedbeef6789))
    phi = ((b2 * Q2[0][0] % o.Q, b2 * Q2[0][1] % o.Q), Q2[1], (1, 0))
    assert o.G2.equals(phi, o.G2.mul(Q2, LAM))


def test_signed_digit_recoding(hc):
    """sum_w sign_w (b_w + 1) 2^(c w) == +-|k| for every window size, with the half scalar's sign
    folded in; at c = 16 the code must never need magnitude 2^15 with a minus sign (u16 overflow)."""
    rng = random.Random(9)
    bound = 1 << 127
    for c in (2, 4, 5, 8, 13, 15, 16):
        W = (128 + c - 1) // c
        ks = [0, 1, (1 << 126) + 12345, int(2 ** 126.96), 0x8000, 0x8000 << 16, 0x7fff8000, 0x80008000,
              (0x8000 << 96) | (0x8000 << 32) | 0x8000, (1 << 112) * 32000 + 0xffff, sum(0x8000 << (16 * j) for j in range(7))]
        ks += [rng.randrange(bound // 2) for _ in range(300)]
        for k in ks:
            if c == 4 and k >> 124 >= 7:
                continue   # (the proven bound 2^126.97 leaves the top nibble < 8)
            for neg in (0, 1):
                codes = (ctypes.c_uint16 * W)()
                words = (ctypes.c_uint32 * 4)(*[(k >> (32 * i)) & 0xffffffff for i in range(4)])
                hc.hc_signed_digits(words, c, W, neg, codes)
                total = 0
                for w in range(W):
                    code = codes[w]
                    if code == 0:
                        continue
                    t = code - 1
                    b, s = t >> 1, t & 1
                    assert b < (1 << (c - 1))
                    total += (-(b + 1) if s else (b + 1)) << (c * w)
                assert total == (-k if neg else k), (c, hex(k), neg)
