"""Pure-Python (exact `int`) restatement of the reference's BN254a arithmetic.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference file:line it follows; paths are relative to
/root/reference/src/main/java/ unless they end in .cu (repo root of the
reference).  Python `int` == java.math.BigInteger (exact), so results are
bit-identical to the reference's serial CPU path.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

# --------------------------------------------------------------------------
# Parameters (algebra/curves/barreto_naehrig/bn254a/bn254a_parameters/*)
# --------------------------------------------------------------------------
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583  # BN254aFqParameters.java:33
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # BN254aFrParameters.java:33
FR_ROOT = 19103219067921713944291392827692070036145651957329286315305642004821462161904  # BN254aFrParameters.java:34
FR_S = 28  # BN254aFrParameters.java:39
FR_MULT_GEN = 5  # BN254aFrParameters.java:35
FQ2_NONRESIDUE = Q - 1  # BN254aFq2Parameters.java:38

G1_ONE = (1, 2, 1)  # BN254aG1Parameters.java:23-24
G1_ZERO = (0, 1, 0)  # BN254aG1Parameters.java:52-55
G2_ONE = (  # BN254aG2Parameters.java:25-32
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
    (1, 0),
)
G2_ZERO = ((0, 0), (0, 0), (0, 0))  # fork quirk, BN254aG2Parameters.java:60-68
G2_ZERO_AFFINE = ((0, 0), (1, 0), (0, 0))  # what toAffineCoordinates gives for O

G1_FIXED_BASE_WINDOW_TABLE = [  # BN254aG1Parameters.java:25-50
    1, 5, 11, 32, 55, 162, 360, 815, 2373, 6978, 7122, 0, 57818, 0, 169679,
    439759, 936073, 0, 4666555, 7580404, 0, 34552892]
G2_FIXED_BASE_WINDOW_TABLE = [  # BN254aG2Parameters.java:33-58
    1, 5, 10, 25, 59, 154, 334, 743, 2034, 4988, 8888, 26271, 39768, 106276,
    141703, 462423, 926872, 0, 4873049, 5706708, 0, 31673815]


# --------------------------------------------------------------------------
# Field "classes": tiny op tables so G1 (over Fq) and G2 (over Fq2) share the
# group-law restatement exactly as BNG1.java / BNG2.java share formulas.
# --------------------------------------------------------------------------
class FqOps:
    """algebra/fields/Fp.java:38-92 with modulus Q."""
    p = Q
    zero = 0
    one = 1

    @staticmethod
    def add(a, b):  # Fp.java:38-40
        return (a + b) % Q

    @staticmethod
    def sub(a, b):  # Fp.java:42-44
        return (a - b) % Q

    @staticmethod
    def mul(a, b):  # Fp.java:46-49
        return (a * b) % Q

    @staticmethod
    def sqr(a):  # Fp.java:86-88
        return (a * a) % Q

    @staticmethod
    def neg(a):  # Fp.java:82-84
        return (-a) % Q

    @staticmethod
    def inv(a):  # Fp.java:90-92 (BigInteger.modInverse)
        return pow(a, -1, Q)

    @staticmethod
    def is_zero(a):
        return a == 0

    @staticmethod
    def eq(a, b):
        return a == b


class Fq2Ops:
    """algebra/fields/Fp2.java:44-107, non-residue q-1 (u^2 = -1)."""
    zero = (0, 0)
    one = (1, 0)

    @staticmethod
    def add(a, b):  # Fp2.java:44-46
        return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)

    @staticmethod
    def sub(a, b):  # Fp2.java:48-50
        return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)

    @staticmethod
    def mul(a, b):  # Fp2.java:59-72 (Karatsuba)
        c0c0 = (a[0] * b[0]) % Q
        c1c1 = (a[1] * b[1]) % Q
        return ((c0c0 + FQ2_NONRESIDUE * c1c1) % Q,
                ((a[0] + a[1]) * (b[0] + b[1]) - c0c0 - c1c1) % Q)

    @staticmethod
    def sqr(a):  # Fp2.java:94-103 (complex squaring)
        c0c1 = (a[0] * a[1]) % Q
        factor = ((a[0] + a[1]) * (a[0] + FQ2_NONRESIDUE * a[1])) % Q
        return ((factor - c0c1 - FQ2_NONRESIDUE * c0c1) % Q, (c0c1 + c0c1) % Q)

    @staticmethod
    def neg(a):  # Fp2.java:90-92
        return ((-a[0]) % Q, (-a[1]) % Q)

    @staticmethod
    def inv(a):  # Fp2.java:105-115 (Algorithm 8)
        t0 = (a[0] * a[0]) % Q
        t1 = (a[1] * a[1]) % Q
        t2 = (t0 - FQ2_NONRESIDUE * t1) % Q
        t3 = pow(t2, -1, Q)
        return ((a[0] * t3) % Q, (-(a[1] * t3)) % Q)

    @staticmethod
    def is_zero(a):  # Fp2.java:78-80
        return a[0] == 0 and a[1] == 0

    @staticmethod
    def eq(a, b):
        return a[0] == b[0] and a[1] == b[1]


# --------------------------------------------------------------------------
# Group law (algebra/curves/barreto_naehrig/BNG1.java; BNG2.java is the same
# formulas over Fq2)
# --------------------------------------------------------------------------
class Curve:
    def __init__(self, F, one, zero, zero_affine, name):
        self.F = F
        self.one = one
        self.zero = zero
        self.zero_affine = zero_affine
        self.name = name

    def is_zero(self, P):  # BNG1.java:103-105
        return self.F.is_zero(P[2])

    def twice(self, P):  # BNG1.java:133-161 (dbl-2009-l, a = 0)
        F = self.F
        if self.is_zero(P):
            return P
        X1, Y1, Z1 = P
        A = F.sqr(X1)
        B = F.sqr(Y1)
        C = F.sqr(B)
        D = F.sub(F.sub(F.sqr(F.add(X1, B)), A), C)
        D = F.add(D, D)
        E = F.add(F.add(A, A), A)
        Fv = F.sqr(E)
        X3 = F.sub(Fv, F.add(D, D))
        eightC = F.add(C, C)
        eightC = F.add(eightC, eightC)
        eightC = F.add(eightC, eightC)
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), eightC)
        Y1Z1 = F.mul(Y1, Z1)
        Z3 = F.add(Y1Z1, Y1Z1)
        return (X3, Y3, Z3)

    def add(self, P, Qp):  # BNG1.java:38-97 (add-2007-bl)
        F = self.F
        if self.is_zero(P):
            return Qp
        if self.is_zero(Qp):
            return P
        X1, Y1, Z1 = P
        X2, Y2, Z2 = Qp
        Z1Z1 = F.sqr(Z1)
        Z2Z2 = F.sqr(Z2)
        U1 = F.mul(X1, Z2Z2)
        U2 = F.mul(X2, Z1Z1)
        Z1c = F.mul(Z1, Z1Z1)
        Z2c = F.mul(Z2, Z2Z2)
        S1 = F.mul(Y1, Z2c)
        S2 = F.mul(Y2, Z1c)
        if F.eq(U1, U2) and F.eq(S1, S2):
            return self.twice(P)
        H = F.sub(U2, U1)
        S2mS1 = F.sub(S2, S1)
        I = F.sqr(F.add(H, H))
        J = F.mul(H, I)
        r = F.add(S2mS1, S2mS1)
        V = F.mul(U1, I)
        X3 = F.sub(F.sub(F.sqr(r), J), F.add(V, V))
        S1J = F.mul(S1, J)
        Y3 = F.sub(F.mul(r, F.sub(V, X3)), F.add(S1J, S1J))
        Z3 = F.mul(F.sub(F.sub(F.sqr(F.add(Z1, Z2)), Z1Z1), Z2Z2), H)
        return (X3, Y3, Z3)

    def negate(self, P):  # BNG1.java:129-131
        return (P[0], self.F.neg(P[1]), P[2])

    def to_affine(self, P):  # BNG1.java:163-172
        F = self.F
        if self.is_zero(P):
            return (F.zero, F.one, F.zero)
        zi = F.inv(P[2])
        z2 = F.sqr(zi)
        z3 = F.mul(z2, zi)
        return (F.mul(P[0], z2), F.mul(P[1], z3), F.one)

    def equals(self, P, Qp):  # BNG1.java:191-224
        F = self.F
        if self.is_zero(P):
            return self.is_zero(Qp)
        if self.is_zero(Qp):
            return False
        Z1s = F.sqr(P[2])
        Z2s = F.sqr(Qp[2])
        if not F.eq(F.mul(P[0], Z2s), F.mul(Qp[0], Z1s)):
            return False
        Z1c = F.mul(P[2], Z1s)
        Z2c = F.mul(Qp[2], Z2s)
        return F.eq(F.mul(P[1], Z2c), F.mul(Qp[1], Z1c))

    def mul(self, P, scalar: int):  # algebra/groups/AbstractGroup.java:29-51
        if scalar == 1:
            return P
        result = self.zero
        found = False
        for i in range(scalar.bit_length() - 1, -1, -1):
            if found:
                result = self.twice(result)
            if (scalar >> i) & 1:
                found = True
                result = self.add(result, P)
        return result

    def on_curve(self, P) -> bool:
        """Not in the reference; used only to validate generated fixtures."""
        F = self.F
        if self.is_zero(P):
            return True
        x, y, _ = self.to_affine(P)
        lhs = F.sqr(y)
        rhs = F.add(F.mul(F.sqr(x), x), self.b)
        return F.eq(lhs, rhs)


G1 = Curve(FqOps, G1_ONE, G1_ZERO, G1_ZERO, "G1")
G1.b = 3
G2 = Curve(Fq2Ops, G2_ONE, G2_ZERO, G2_ZERO_AFFINE, "G2")
# twist coefficient 3/(9+u) (BN254aG2Parameters / SURVEY §0)
G2.b = Fq2Ops.mul((3, 0), Fq2Ops.inv((9, 1)))


# --------------------------------------------------------------------------
# common/MathUtils.java
# --------------------------------------------------------------------------
def java_log2(x: int) -> int:
    """MathUtils.java:8-10  (int)(Math.log(x)/Math.log(2)) in IEEE doubles."""
    return int(math.log(x) / math.log(2))


def bitreverse(n: int, bits: int) -> int:
    """MathUtils.java:43-53 (32-bit int semantics; inputs here are < 2^30)."""
    count = bits - 1
    reverse = n
    n >>= 1
    while n > 0:
        reverse = (reverse << 1) | (n & 1)
        n >>= 1
        count -= 1
    return (reverse << count) & ((1 << bits) - 1)


# --------------------------------------------------------------------------
# Variable-base MSM (algebra/msm/VariableBaseMSM.java, NaiveMSM.java)
# --------------------------------------------------------------------------
def naive_msm(C: Curve, scalars: Sequence[int], bases: Sequence):
    """NaiveMSM.java:33-46."""
    result = C.zero
    for s, b in zip(scalars, bases):
        result = C.add(result, C.mul(b, s))
    return result


def pippenger_window(length: int) -> int:
    """VariableBaseMSM.java:137-139 / algebra_msm_VariableBaseMSM.cu:1267-1270."""
    log2_length = max(1, java_log2(length))
    return log2_length - (log2_length // 3)


def pippenger_msm(C: Curve, scalars: Sequence[int], bases: Sequence, num_bits: int = 254):
    """VariableBaseMSM.java:134-188 — the serial CPU semantics of the hot path."""
    length = len(scalars)
    c = pippenger_window(length)
    num_buckets = 1 << c
    num_groups = (num_bits + c - 1) // c
    zero = C.zero
    result = zero
    mask = num_buckets - 1
    for k in range(num_groups - 1, -1, -1):
        buckets = [zero] * num_buckets
        for i in range(length):
            idx = (scalars[i] >> (k * c)) & mask  # testBit loop :157-160
            if idx == 0:  # :163
                continue
            buckets[idx] = C.add(buckets[idx], bases[i])  # :168
        running = zero
        for i in range(num_buckets - 1, 0, -1):  # :171-177
            running = C.add(running, buckets[i])
            result = C.add(result, running)
        if k > 0:  # :180-184
            for _ in range(c):
                result = C.twice(result)
    return result


def sorted_msm(C: Curve, scalars: Sequence[int], bases: Sequence):
    """VariableBaseMSM.java:41-56."""
    pairs = sorted(zip(scalars, bases), key=lambda t: t[0])
    result = C.zero
    base = C.zero
    for i in range(len(pairs) - 1, -1, -1):
        scalar = pairs[i][0] - pairs[i - 1][0] if i != 0 else pairs[i][0]
        base = C.add(base, pairs[i][1])
        result = C.add(result, C.mul(base, scalar))
    return result


BOS_COSTER_MSM_THRESHOLD = 1048576  # VariableBaseMSM.java:29


def bos_coster_msm(C: Curve, scalars: Sequence[int], bases: Sequence):
    """VariableBaseMSM.java:86-119 (bosCosterMSM): a max-priority queue on the scalar; the two largest pairs
    (e1, b1), (e2, b2) become (e1 - e2, b1) and (e2, b1 + b2) — e1 b1 + e2 b2 = (e1 - e2) b1 + e2 (b1 + b2) — unless
    e1 div e2 >= 2^20, in which case e1 b1 is multiplied out (AbstractGroup.mul) and e2 goes back.  Scalars must be
    positive: e2 = 0 divides by zero here as BigInteger.divide throws in the Java.  (java.util.PriorityQueue breaks
    ties arbitrarily; the group element does not depend on it.)"""
    import heapq
    heap = []  # (-scalar, sequence number, scalar, base): heapq is a min-heap, the counter keeps bases uncompared
    seq = 0
    for sc, b in zip(scalars, bases):
        heapq.heappush(heap, (-sc, seq, sc, b))
        seq += 1

    def poll():
        return heapq.heappop(heap)[2:] if heap else None

    result = C.zero
    while True:
        e1 = poll()
        if e1 is None:
            break
        e2 = poll()
        if e2 is None:
            break
        if e1[0] // e2[0] >= BOS_COSTER_MSM_THRESHOLD:  # :99-101
            result = C.add(result, C.mul(e1[1], e1[0]))
            heapq.heappush(heap, (-e2[0], seq, e2[0], e2[1]))
            seq += 1
        else:  # :102-110
            value = e1[0] - e2[0]
            if value != 0:
                heapq.heappush(heap, (-value, seq, value, e1[1]))
                seq += 1
            heapq.heappush(heap, (-e2[0], seq, e2[0], C.add(e1[1], e2[1])))
            seq += 1
    while e1 is not None:  # :113-116
        result = C.add(result, C.mul(e1[1], e1[0]))
        e1 = poll()
    return result


def filtered_msm(C: Curve, scalars: Sequence[int], bases: Sequence):
    """VariableBaseMSM.java:736-770 (single-group restatement of the 0/1 filter
    in front of pippengerMSM)."""
    acc = C.zero
    conv_s, conv_b = [], []
    num_bits = 0
    for s, b in zip(scalars, bases):
        if s == 0:
            continue
        if s == 1:
            acc = C.add(acc, b)
        else:
            conv_s.append(s)
            conv_b.append(b)
            num_bits = max(num_bits, s.bit_length())
    if not conv_s:
        return acc
    return C.add(acc, pippenger_msm(C, conv_s, conv_b, num_bits))


# --------------------------------------------------------------------------
# Fixed-base MSM (algebra/msm/FixedBaseMSM.java)
# --------------------------------------------------------------------------
def fixed_base_window_size(num_scalars: int, table: Sequence[int]) -> int:
    """FixedBaseMSM.java:49-66."""
    if not table:
        return 17
    window = 1
    for i in range(len(table) - 1, -1, -1):
        if table[i] != 0 and num_scalars >= table[i]:
            window = i + 1
            break
    return window


def fixed_base_window_table(C: Curve, base, scalar_size: int, window_size: int):
    """FixedBaseMSM.java:71-99."""
    num_windows = scalar_size // window_size if scalar_size % window_size == 0 \
        else scalar_size // window_size + 1
    inner_limit = 1 << window_size
    if num_windows == 0:
        return [[C.zero]]
    table = []
    base_outer = base
    for _outer in range(num_windows):
        row = []
        base_inner = C.zero
        for _inner in range(inner_limit):
            row.append(base_inner)
            base_inner = C.add(base_inner, base_outer)
        table.append(row)
        for _w in range(window_size):
            base_outer = C.twice(base_outer)
    return table


def fixed_base_serial_msm(C: Curve, scalar_size: int, window_size: int, table, scalar: int):
    """FixedBaseMSM.java:141-167."""
    outerc = (scalar_size + window_size - 1) // window_size
    res = table[0][0]
    for outer in range(outerc):
        inner = (scalar >> (outer * window_size)) & ((1 << window_size) - 1)
        res = C.add(res, table[outer][inner])
    return res


def fixed_base_mul(C: Curve, base, scalar_size: int, window_size: int, scalar: int):
    """Same value as fixed_base_serial_msm without materialising the table:
    sum_w digit_w * 2^(w*windowSize) * base  (FixedBaseMSM.java:71-99,141-167).
    Only the first outerc windows of the scalar are used, as in the reference."""
    outerc = (scalar_size + window_size - 1) // window_size
    truncated = scalar & ((1 << (outerc * window_size)) - 1)
    return C.mul(base, truncated) if truncated != 1 else base


def field_batch_mul(xs: Sequence[int], b: int) -> List[int]:
    """FixedBaseMSM.java:753-785 / algebra_msm_FixedBaseMSM.cu:1241-1266:
    x_i * b mod r."""
    return [(x * b) % R for x in xs]


# --------------------------------------------------------------------------
# FFT over Fr (algebra/fft/FFTAuxiliary.java, SerialFFT.java)
# --------------------------------------------------------------------------
def fr_root_of_unity(order: int) -> int:
    """Fp.java:98-102: root^(floor(r / order)) mod r."""
    return pow(FR_ROOT, R // order, R)


def serial_radix2_fft(a: List[int], omega: int, p: int = R) -> None:
    """FFTAuxiliary.java:60-124 (in place)."""
    n = len(a)
    if n == 1:
        return
    logn = java_log2(n)
    assert n == (1 << logn)
    for k in range(n):  # :101-106
        rk = bitreverse(k, logn)
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    m = 1
    for _s in range(1, logn + 1):  # :108-123
        w_m = pow(omega, n // (2 * m), p)
        for k in range(0, n, 2 * m):
            w = 1
            for j in range(m):
                t = (w * a[k + j + m]) % p
                a[k + j + m] = (a[k + j] - t) % p
                a[k + j] = (a[k + j] + t) % p
                w = (w * w_m) % p
        m *= 2


def multiply_by_coset(a: List[int], g: int, p: int = R) -> None:
    """FFTAuxiliary.java:224-232."""
    coset = g
    for i in range(1, len(a)):
        a[i] = (a[i] * coset) % p
        coset = (coset * g) % p


def radix2_fft(a: List[int], p: int = R) -> None:
    """SerialFFT.java:75-78 with omega from SerialFFT.java:24-28."""
    serial_radix2_fft(a, fr_root_of_unity(len(a)), p)


def radix2_inverse_fft(a: List[int], p: int = R) -> None:
    """SerialFFT.java:86-95."""
    n = len(a)
    omega = fr_root_of_unity(n)
    serial_radix2_fft(a, pow(omega, -1, p), p)
    c = pow(n, -1, p)
    for i in range(n):
        a[i] = (a[i] * c) % p


def radix2_coset_fft(a: List[int], g: int, p: int = R) -> None:
    """SerialFFT.java:100-105."""
    multiply_by_coset(a, g, p)
    radix2_fft(a, p)


def radix2_coset_inverse_fft(a: List[int], g: int, p: int = R) -> None:
    """SerialFFT.java:111-115."""
    radix2_inverse_fft(a, p)
    multiply_by_coset(a, pow(g, -1, p), p)


def compute_z(t: int, domain_size: int, p: int = R) -> int:
    """SerialFFT.java:143-145: the vanishing polynomial of S at t."""
    return (pow(t, domain_size, p) - 1) % p


def divide_by_z_on_coset(a: List[int], coset: int, p: int = R) -> None:
    """SerialFFT.java:158-163."""
    zinv = pow(compute_z(coset, len(a), p), -1, p)
    for i in range(len(a)):
        a[i] = (a[i] * zinv) % p


def qap_witness_coefficients_h(a_eval: Sequence[int], b_eval: Sequence[int], c_eval: Sequence[int],
                               g: int = FR_MULT_GEN, p: int = R) -> List[int]:
    """R1CStoQAP.R1CStoQAPWitness from the evaluations on (R1CStoQAP.java:163-230): returns
    coefficientsH, domainSize + 1 entries."""
    A, B, C = list(a_eval), list(b_eval), list(c_eval)
    radix2_inverse_fft(A, p)        # :166-167
    radix2_inverse_fft(B, p)
    radix2_coset_fft(A, g, p)       # :173-174
    radix2_coset_fft(B, g, p)
    H = [(x * y) % p for x, y in zip(A, B)]   # :180-183
    radix2_inverse_fft(C, p)        # :201
    radix2_coset_fft(C, g, p)       # :207
    H = [(h - c) % p for h, c in zip(H, C)]   # :213-216
    divide_by_z_on_coset(H, g, p)   # :224
    radix2_coset_inverse_fft(H, g, p)  # :229
    H.append(0)                     # :230
    return H


def naive_dft(a: Sequence[int], omega: int, p: int = R) -> List[int]:
    """Polynomial evaluation at omega^i — what SerialFFTTest.java:168-190
    compares the FFT with."""
    n = len(a)
    out = []
    for i in range(n):
        x = pow(omega, i, p)
        acc = 0
        for coeff in reversed(a):
            acc = (acc * x + coeff) % p
        out.append(acc)
    return out


# --------------------------------------------------------------------------
# Wire codec (JNI byte formats)
# --------------------------------------------------------------------------
def to_le32(v: int) -> bytes:
    """VariableBaseMSM.java:121-131 bigIntegerToByteArrayHelperCGBN: 32-byte
    little-endian (value < 2^255)."""
    return int(v).to_bytes(32, "little")


def from_le64(b: bytes) -> int:
    """VariableBaseMSM.java:239-258: reverse 64 bytes, new BigInteger."""
    assert len(b) == 64
    return int.from_bytes(b, "little")


def from_be64(b: bytes) -> int:
    """FixedBaseMSM.java:233-241: 64-byte big-endian slice, new BigInteger."""
    assert len(b) == 64
    return int.from_bytes(b, "big")


def to_fft_bytes(v: int) -> bytes:
    """FFTAuxiliary.java:41-51: BigInteger.toByteArray() (two's complement,
    minimal) reversed into a buffer padded to a multiple of 4 bytes."""
    v = int(v)
    nbytes = v.bit_length() // 8 + 1  # BigInteger.toByteArray length for v >= 0
    padded = (nbytes + 3) // 4 * 4
    return v.to_bytes(padded, "little")


def g1_to_wire(P) -> bytes:
    """VariableBaseMSM.java:221-228 + BN254aG1.java:42-48: X|Y|Z, 32-B LE each."""
    return to_le32(P[0]) + to_le32(P[1]) + to_le32(P[2])


def g2_to_wire(P) -> bytes:
    """VariableBaseMSM.java:279-288 + BN254aG2.java:77-86:
    X.c0|X.c1|Y.c0|Y.c1|Z.c0|Z.c1, 32-B LE each."""
    return b"".join(to_le32(P[i][j]) for i in range(3) for j in range(2))


def g1_from_out_le(b: bytes):
    """VariableBaseMSM.java:239-258 (192 B, 64-B LE coords)."""
    return tuple(from_le64(b[64 * i:64 * (i + 1)]) for i in range(3))


def g2_from_out_le(b: bytes):
    """VariableBaseMSM.java:293-326 (384 B: Xa|Xb|Ya|Yb|Za|Zb, 64-B LE)."""
    v = [from_le64(b[64 * i:64 * (i + 1)]) for i in range(6)]
    return ((v[0], v[1]), (v[2], v[3]), (v[4], v[5]))


def g1_from_out_be(b: bytes):
    """FixedBaseMSM.java:233-246 (192 B, 64-B BE coords)."""
    return tuple(from_be64(b[64 * i:64 * (i + 1)]) for i in range(3))


def g2_from_out_be(b: bytes):
    """FixedBaseMSM.java:284-305 (384 B, 64-B BE coords)."""
    v = [from_be64(b[64 * i:64 * (i + 1)]) for i in range(6)]
    return ((v[0], v[1]), (v[2], v[3]), (v[4], v[5]))


def g1_out_le(P) -> bytes:
    """Expected JNI return bytes for an (affine-normalised) G1 point."""
    return b"".join(int(c).to_bytes(64, "little") for c in P)


def g2_out_le(P) -> bytes:
    return b"".join(int(P[i][j]).to_bytes(64, "little") for i in range(3) for j in range(2))


def g1_out_be(P) -> bytes:
    return b"".join(int(c).to_bytes(64, "big") for c in P)


def g2_out_be(P) -> bytes:
    return b"".join(int(P[i][j]).to_bytes(64, "big") for i in range(3) for j in range(2))
