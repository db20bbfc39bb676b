"""Pure-Python (exact `int`) restatement of the reference's serial Groth16 setup and prove for BN254a
— the CALLER of the MSM / FFT hot path (BASELINE.json configs[4]).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Paths are relative to
/root/reference/src/main/java/.  Restated:

    profiler/generation/R1CSConstruction.java:28-110     serialConstruct (synthetic add / mul chain)
    relations/objects/LinearCombination.java:39-50       evaluate (index 0 counts as `one`)
    algebra/fft/FFTAuxiliary.java:250-302                serialRadix2LagrangeCoefficients
    reductions/r1cs_to_qap/R1CStoQAP.java:37-98          R1CStoQAPRelation (QAP instance at t)
    reductions/r1cs_to_qap/R1CStoQAP.java:125-237        R1CStoQAPWitness
    zk_proof_systems/zkSNARK/SerialSetup.java:32-192     generate (without the pairing of the verification key)
    zk_proof_systems/zkSNARK/SerialProver.java:26-119    prove

Every "random" field element of setup and prove is `new Random(config.seed()).nextLong()` reduced mod r
(Fp.java:72-80, Configuration.java:52: seed 10) — the SAME value every time, which is what the reference
does.  Group elements are kept as the Jacobian triples the Java objects hold; the proof is compared on
its affine-normalised form (BNG1.java:163-172).

Because every key element is a known multiple of the generators, the Groth16 verification equation can be
checked in the exponent without a pairing (`verify_in_the_exponent`): that is what pins this restatement
(the reference's own acceptance test is Verifier.verify == true, SerialzkSNARKTest.java:76-92).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

from . import bn254 as o
from .javarand import fp_random

R = o.R
SEED = 10  # configuration/Configuration.java:52


def fr_random(seed: int = SEED) -> int:
    """Fp.random (Fp.java:72-80) on BN254aFr with config.seed()."""
    return fp_random(seed, R)


def lowest_power_of_two(n: int) -> int:
    """common/MathUtils.java:20-41."""
    if n < 1:
        return 1
    r = 1
    while r < n:
        r <<= 1
    return r


# ---------------------------------------------------------------------------- R1CS
class R1CS:
    """relations/r1cs/R1CSRelation: constraints[i] = (A, B, C), each a list of (index, value) terms."""

    def __init__(self, constraints, num_inputs, num_auxiliary):
        self.constraints = constraints
        self.num_inputs = num_inputs
        self.num_auxiliary = num_auxiliary
        self.num_variables = num_inputs + num_auxiliary
        self.num_constraints = len(constraints)


def serial_construct(num_constraints: int, num_inputs: int, seed: int = SEED):
    """R1CSConstruction.serialConstruct (R1CSConstruction.java:28-110): returns (r1cs, primary, auxiliary)."""
    assert num_inputs <= num_constraints + 1
    num_auxiliary = 3 + num_constraints - num_inputs
    num_variables = num_inputs + num_auxiliary
    a = fr_random(seed)
    b = fr_random(seed)
    full = [1, a, b]
    constraints = []
    for i in range(num_constraints - 1):
        if i % 2 != 0:  # a * b = c
            A, B, C = [(i + 1, 1)], [(i + 2, 1)], [(i + 3, 1)]
            tmp = (a * b) % R
        else:           # a + b = c
            A, B, C = [(i + 1, 1), (i + 2, 1)], [(0, 1)], [(i + 3, 1)]
            tmp = (a + b) % R
        a, b = b, tmp
        full.append(tmp)
        constraints.append((A, B, C))
    A = [(i, 1) for i in range(1, num_variables - 1)]
    B = [(i, 1) for i in range(1, num_variables - 1)]
    res = 0
    for i in range(1, num_variables - 1):
        res = (res + full[i]) % R
    C = [(num_variables - 1, 1)]
    full.append((res * res) % R)
    constraints.append((A, B, C))
    r1cs = R1CS(constraints, num_inputs, num_auxiliary)
    assert r1cs.num_variables == len(full)
    return r1cs, full[:num_inputs], full[num_inputs:]


def lc_evaluate(terms, full: Sequence[int]) -> int:
    """LinearCombination.evaluate (LinearCombination.java:39-50): a term with index 0 contributes `one`
    whatever its coefficient."""
    result = 0
    for index, value in terms:
        result = (result + (1 if index == 0 else (full[index] * value) % R)) % R
    return result


def is_satisfied(r1cs: R1CS, primary, auxiliary) -> bool:
    """relations/r1cs/R1CSRelation.isSatisfied."""
    full = list(primary) + list(auxiliary)
    if full[0] != 1:
        return False
    for A, B, C in r1cs.constraints:
        if (lc_evaluate(A, full) * lc_evaluate(B, full) - lc_evaluate(C, full)) % R != 0:
            return False
    return True


# ---------------------------------------------------------------------------- R1CS -> QAP
def lagrange_coefficients(t: int, m: int) -> List[int]:
    """FFTAuxiliary.serialRadix2LagrangeCoefficients (FFTAuxiliary.java:250-302)."""
    if m == 1:
        return [1]
    assert m == 1 << o.java_log2(m)
    omega = o.fr_root_of_unity(m)
    coeffs = [0] * m
    if pow(t, m, R) == 1:
        omega_i = 1
        for i in range(m):
            if omega_i == t:
                coeffs[i] = 1
                return coeffs
            omega_i = (omega_i * omega) % R
    Z = (pow(t, m, R) - 1) % R
    l = (Z * pow(m, -1, R)) % R
    r = 1
    for i in range(m):
        coeffs[i] = (l * pow((t - r) % R, -1, R)) % R
        l = (l * omega) % R
        r = (r * omega) % R
    return coeffs


class QAPRelation:
    def __init__(self, At, Bt, Ct, Ht, Zt, t, num_inputs, num_variables, degree):
        self.At, self.Bt, self.Ct, self.Ht, self.Zt, self.t = At, Bt, Ct, Ht, Zt, t
        self.num_inputs, self.num_variables, self.degree = num_inputs, num_variables, degree


def r1cs_to_qap_relation(r1cs: R1CS, t: int) -> QAPRelation:
    """R1CStoQAP.R1CStoQAPRelation (R1CStoQAP.java:37-98)."""
    nc, ni, nv = r1cs.num_constraints, r1cs.num_inputs, r1cs.num_variables
    m = lowest_power_of_two(nc + ni)
    At, Bt, Ct = [0] * nv, [0] * nv, [0] * nv
    lag = lagrange_coefficients(t, m)
    for i in range(ni):
        At[i] = lag[nc + i]
    for i in range(nc):
        A, B, C = r1cs.constraints[i]
        li = lag[i]
        for index, value in A:
            At[index] = (At[index] + li * value) % R
        for index, value in B:
            Bt[index] = (Bt[index] + li * value) % R
        for index, value in C:
            Ct[index] = (Ct[index] + li * value) % R
    Ht = []
    ti = 1
    for _ in range(m + 1):
        Ht.append(ti)
        ti = (ti * t) % R
    Zt = o.compute_z(t, m)
    return QAPRelation(At, Bt, Ct, Ht, Zt, t, ni, nv, m)


def r1cs_to_qap_witness(r1cs: R1CS, primary, auxiliary):
    """R1CStoQAP.R1CStoQAPWitness (R1CStoQAP.java:125-237): returns (full assignment, coefficientsH with
    domainSize + 1 entries, domainSize) and the three evaluation vectors the transforms start from."""
    nc, ni = r1cs.num_constraints, r1cs.num_inputs
    m = lowest_power_of_two(nc + ni)
    full = list(primary) + list(auxiliary)
    A, B, C = [0] * m, [0] * m, [0] * m
    for i in range(ni):
        A[i + nc] = full[i]
    for i in range(nc):
        cA, cB, cC = r1cs.constraints[i]
        A[i] = (lc_evaluate(cA, full) + A[i]) % R
        B[i] = lc_evaluate(cB, full)
        C[i] = lc_evaluate(cC, full)
    H = o.qap_witness_coefficients_h(A, B, C)   # :163-230
    return full, H, m, (A, B, C)


# ---------------------------------------------------------------------------- setup
def _bit_size_g1(P) -> int:
    """BNG1.bitSize (BNG1.java:174-176) of the Jacobian triple the Java object holds."""
    return max(c.bit_length() for c in P)


def _bit_size_g2(P) -> int:
    """BNG2.bitSize (BNG2.java:179-181); Fp2.bitSize = max over the two components."""
    return max(max(c[0].bit_length(), c[1].bit_length()) for c in P)


def _num_windows(scalar_size: int, window_size: int) -> int:
    return scalar_size // window_size if scalar_size % window_size == 0 else scalar_size // window_size + 1


def _batch_msm(C, scalar_size, window_size, base, scalars):
    """FixedBaseMSM.batchMSM semantics = serialMSM over getWindowTable (FixedBaseMSM.java:71-99,141-167)."""
    table = o.fixed_base_window_table(C, base, scalar_size, window_size)
    return [o.fixed_base_serial_msm(C, scalar_size, window_size, table, s) for s in scalars]


class CRS:
    pass


def serial_setup(r1cs: R1CS, seed: int = SEED) -> CRS:
    """SerialSetup.generate (SerialSetup.java:32-192).  Returns the proving key, the G1/G2 part of the
    verification key (no pairing) and the secrets / key scalars (for checks in the exponent)."""
    t = alpha = beta = gamma = delta = fr_random(seed)   # :40-44: five draws of new Random(seed)
    inv_gamma = pow(gamma, -1, R)
    inv_delta = pow(delta, -1, R)
    qap = r1cs_to_qap_relation(r1cs, t)
    ni, nv = qap.num_inputs, qap.num_variables
    gammaABC = [((beta * qap.At[i] + alpha * qap.Bt[i] + qap.Ct[i]) * inv_gamma) % R for i in range(ni)]
    deltaABC = [((beta * qap.At[i] + alpha * qap.Bt[i] + qap.Ct[i]) * inv_delta) % R for i in range(ni, nv)]
    non_zero_at = sum(1 for x in qap.At if x != 0)
    non_zero_bt = sum(1 for x in qap.Bt if x != 0)
    # :91-112
    gen_g1 = o.G1.mul(o.G1.one, fr_random(seed))            # BNG1.random (BNG1.java:125-127)
    scalar_count_g1 = non_zero_at + non_zero_bt + nv
    scalar_size_g1 = _bit_size_g1(gen_g1)
    window_g1 = o.fixed_base_window_size(scalar_count_g1, o.G1_FIXED_BASE_WINDOW_TABLE)
    gen_g2 = o.G2.mul(o.G2.one, fr_random(seed))
    scalar_count_g2 = non_zero_bt
    scalar_size_g2 = _bit_size_g2(gen_g2)
    window_g2 = o.fixed_base_window_size(scalar_count_g2, o.G2_FIXED_BASE_WINDOW_TABLE)
    crs = CRS()
    crs.qap = qap
    crs.secrets = dict(t=t, alpha=alpha, beta=beta, gamma=gamma, delta=delta)
    crs.gen_g1, crs.gen_g2 = gen_g1, gen_g2
    crs.window_g1, crs.window_g2 = window_g1, window_g2
    crs.scalar_size_g1, crs.scalar_size_g2 = scalar_size_g1, scalar_size_g2
    # :117-121
    crs.alpha_g1 = o.G1.mul(gen_g1, alpha)
    crs.beta_g1 = o.G1.mul(gen_g1, beta)
    crs.beta_g2 = o.G2.mul(gen_g2, beta)
    crs.delta_g1 = o.G1.mul(gen_g1, delta)
    crs.delta_g2 = o.G2.mul(gen_g2, delta)
    # :123-151: the fixed-base batches.  All G1 batches share one window table.
    table_g1 = o.fixed_base_window_table(o.G1, gen_g1, scalar_size_g1, window_g1)
    table_g2 = o.fixed_base_window_table(o.G2, gen_g2, scalar_size_g2, window_g2)

    def b1(scalars):
        return [o.fixed_base_serial_msm(o.G1, scalar_size_g1, window_g1, table_g1, s) for s in scalars]

    def b2(scalars):
        return [o.fixed_base_serial_msm(o.G2, scalar_size_g2, window_g2, table_g2, s) for s in scalars]

    crs.delta_abc_scalars = deltaABC
    crs.delta_abc_g1 = b1(deltaABC)
    crs.query_a = b1(qap.At)
    crs.query_b = list(zip(b1(qap.Bt), b2(qap.Bt)))
    inv_delta_zt = (qap.Zt * pow(delta, -1, R)) % R
    crs.ht_scalars = [(h * inv_delta_zt) % R for h in qap.Ht]   # :146-149 (qap.Ht() is overwritten)
    crs.query_h = b1(crs.ht_scalars)
    # verification key, group part (:160-164)
    crs.gamma_g2 = o.G2.mul(gen_g2, gamma)
    crs.gamma_abc_scalars = gammaABC
    crs.gamma_abc_g1 = b1(gammaABC)
    crs.r1cs = r1cs
    return crs


# ---------------------------------------------------------------------------- prove
def serial_prove(crs: CRS, primary, auxiliary, seed: int = SEED):
    """SerialProver.prove (SerialProver.java:26-119).  Returns (A, B, C) = (G1, G2, G1) Jacobian triples
    and the intermediate values of the run."""
    r1cs = crs.r1cs
    full, H, m, _ = r1cs_to_qap_witness(r1cs, primary, auxiliary)
    r = fr_random(seed)   # :58-59
    s = fr_random(seed)
    G1, G2 = o.G1, o.G2
    rs_delta = G1.mul(crs.delta_g1, (r * s) % R)
    ni, nv = r1cs.num_inputs, r1cs.num_variables
    # :76-79  (VariableBaseMSM.serialMSM == the group element pippengerMSM computes)
    eval_at = G1.add(o.pippenger_msm(G1, list(primary), crs.query_a[:ni]),
                     o.pippenger_msm(G1, list(auxiliary), crs.query_a[ni:nv]))
    # :82-88
    bp1 = o.pippenger_msm(G1, list(primary), [q[0] for q in crs.query_b[:ni]])
    bp2 = o.pippenger_msm(G2, list(primary), [q[1] for q in crs.query_b[:ni]])
    bw1 = o.pippenger_msm(G1, list(auxiliary), [q[0] for q in crs.query_b[ni:nv]])
    bw2 = o.pippenger_msm(G2, list(auxiliary), [q[1] for q in crs.query_b[ni:nv]])
    eval_bt_g1 = G1.add(bp1, bw1)
    eval_bt_g2 = G2.add(bp2, bw2)
    # :91-93
    eval_ht_zt = o.pippenger_msm(G1, H, crs.query_h)
    # :98-102
    num_witness = nv - ni
    eval_abc = o.pippenger_msm(G1, list(auxiliary[:num_witness]), crs.delta_abc_g1[:num_witness])
    eval_abc = G1.add(eval_abc, eval_ht_zt)
    # :105-114
    A = G1.add(G1.add(crs.alpha_g1, eval_at), G1.mul(crs.delta_g1, r))
    B1 = G1.add(G1.add(crs.beta_g1, eval_bt_g1), G1.mul(crs.delta_g1, s))
    B2 = G2.add(G2.add(crs.beta_g2, eval_bt_g2), G2.mul(crs.delta_g2, s))
    C = G1.add(G1.add(G1.add(eval_abc, G1.mul(A, s)), G1.mul(B1, r)), G1.negate(rs_delta))
    info = dict(full=full, H=H, m=m, r=r, s=s, eval_at=eval_at, eval_bt_g1=eval_bt_g1, eval_bt_g2=eval_bt_g2,
                eval_ht_zt=eval_ht_zt, eval_abc=eval_abc, B1=B1)
    return (A, B2, C), info


# ---------------------------------------------------------------------------- checks in the exponent
def proof_scalars(crs: CRS, full: Sequence[int], H: Sequence[int], r: int, s: int) -> Tuple[int, int, int]:
    """Discrete logs (a, b, c) of the proof w.r.t. the setup's generators, from exact integer arithmetic:
    A = a * genG1, B = b * genG2, C = c * genG1 (SerialProver.java:105-114 with every key element replaced
    by its known scalar)."""
    q, sec = crs.qap, crs.secrets
    ni, nv = q.num_inputs, q.num_variables
    alpha, beta, delta = sec["alpha"], sec["beta"], sec["delta"]
    a = (alpha + sum(z * x for z, x in zip(full, q.At)) + r * delta) % R
    b = (beta + sum(z * x for z, x in zip(full, q.Bt)) + s * delta) % R
    habc = sum(z * x for z, x in zip(full[ni:nv], crs.delta_abc_scalars)) + sum(h * x for h, x in zip(H, crs.ht_scalars))
    c = (habc + a * s + b * r - r * s * delta) % R
    return a, b, c


def verify_in_the_exponent(crs: CRS, primary: Sequence[int], abc: Tuple[int, int, int]) -> bool:
    """Groth16 verification equation e(A, B) = e(alpha, beta) e(sum_i x_i gammaABC_i, gamma) e(C, delta)
    (zkSNARK/Verifier.java:24-59) with both sides taken to the exponent of e(genG1, genG2)."""
    a, b, c = abc
    sec = crs.secrets
    acc = sum(x * g for x, g in zip(primary, crs.gamma_abc_scalars)) % R
    return (a * b - sec["alpha"] * sec["beta"] - acc * sec["gamma"] - c * sec["delta"]) % R == 0
