"""Deterministic input generators shared by tests/golden/make_golden.py and the tests that
replay the golden fixtures (inputs of the larger cases are regenerated, their SHA-256 is
checked against the fixture)."""
import random

from oracle import bn254 as o
from oracle.javarand import JavaRandom

# (name, n, kind)
MSM_CASES = [
    ("n1_uniform", 1, "uniform"), ("n2_uniform", 2, "uniform"), ("n3_uniform", 3, "uniform"),
    ("n8_edge", 8, "edge"), ("n64_uniform", 64, "uniform"), ("n64_fp_random", 64, "fp_random"),
    ("n64_repeated_base", 64, "repeated"), ("n1023_uniform", 1023, "uniform"),
    ("n1024_fp_random", 1024, "fp_random"),
]


def _bases(C, n, seed, jacobian=False):
    """P_i = (k0 + i) * G by successive additions (fast), affine unless jacobian."""
    rng = random.Random(seed)
    k0 = rng.randrange(1, 1 << 60)
    step = C.mul(C.one, 1)
    P = C.mul(C.one, k0)
    out = []
    for _ in range(n):
        out.append(P if jacobian else C.to_affine(P))
        P = C.add(P, step)
    return out


def msm_inputs(C, n, kind):
    rng = random.Random(1000 * n + len(kind))
    if kind == "uniform":
        return [rng.randrange(o.R) for _ in range(n)], _bases(C, n, n)
    if kind == "fp_random":
        # Fp.random: new Random(seed).nextLong() mod r (Fp.java:72-80): 64-bit or r - 64-bit
        return [JavaRandom(10 + i).next_long() % o.R for i in range(n)], _bases(C, n, n + 1, jacobian=True)
    if kind == "repeated":
        # VariableBaseMSMProfiling.java:23-31: one base N times
        return [rng.randrange(o.R) for _ in range(n)], [_bases(C, 1, 7)[0]] * n
    if kind == "edge":
        b = _bases(C, n, 3)
        b[2] = C.zero
        b[5] = C.negate(b[4])
        s = [0, 1, o.R - 1, 12345, 77, 77, 2, (1 << 253) + 1]
        return s[:n], b
    raise ValueError(kind)


def fft_inputs(n):
    rng = random.Random(31 * n + 5)
    return [rng.randrange(o.R) for _ in range(n)]


def fixed_scalars(k):
    rng = random.Random(99)
    return [0, 1, o.R - 1, (1 << 253) + 7] + [rng.randrange(o.R) for _ in range(k - 4)]
