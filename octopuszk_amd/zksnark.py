"""Host-side mirror of the reference's SERIAL Groth16 setup and prover for BN254a — the callers of the
MSM / FFT hot path (BASELINE.json configs[4]) — with every group / transform
operation on the GPU through the C ABI of libozk_hip.so and the keys resident in HBM between them.

    serial_construct        profiler/generation/R1CSConstruction.java:28-110   (synthetic R1CS + witness)
    r1cs_to_qap_relation    reductions/r1cs_to_qap/R1CStoQAP.java:37-98        (QAP instance at t; host)
    SerialSetup.generate    zk_proof_systems/zkSNARK/SerialSetup.java:32-192   (4 x batchMSM + doubleBatchMSM)
    SerialProver.prove      zk_proof_systems/zkSNARK/SerialProver.java:26-119  (witness map, 4 x serialMSM,
                                                                                2 x doubleMSM, assembly)

What runs where.  The reference keeps field elements as BigInteger objects on the JVM heap and crosses
the JNI for each MSM / batch; here the host side is Python ints (the image has no JDK) for exactly the
parts the Java does on the CPU in the SETUP (R1CS construction, Lagrange coefficients, the sparse accumulation
of A_i(t), B_i(t), C_i(t)), and device buffers for everything else:

  * setup: the five fixed-base batches write the proving key straight into the wire-in format of the
    variable-base MSM (ozk_fixed_batch_msm_compact_dev), so the key never leaves HBM;
  * prove: the assignment goes up once (32-byte elements), the constraint matrices sit in HBM as CSR and are
    evaluated there (ozk_r1cs_evaluate_dev), ozk_qap_witness_dev leaves coefficientsH in HBM, the
    MSMs run over bases prepared once per key (ozk_var_msm_prepare_dev) — G1 through a two-stage pipeline on
    two streams, G2 on a third — and the proof is assembled on the device (ozk_points_sum_dev and one
    3-term MSM, s A + r B1 - r s delta, on a fourth stream behind the long MSMs) from the MSM results.

Proof elements are returned in the wire-out format of the variable-base natives (affine-normalised,
64-byte little-endian coordinates).  There is no CPU fallback: without the HIP library nothing here works.
"""
import ctypes
import os
import time

import numpy as np
import torch

from . import lib as _lib
from .fft import FR, FR_MULT_GEN, root_of_unity
from .fixed_base_msm import G1_WINDOW_TABLE, G2_WINDOW_TABLE, get_window_size

SEED = 10  # configuration/Configuration.java:52
G1_ONE = (1, 2, 1)  # BN254aG1Parameters.java:23-24
G2_ONE = (  # BN254aG2Parameters.java:25-32
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
    (1, 0))


# ---------------------------------------------------------------------------- small helpers
def _java_random_next_long(seed: int) -> int:
    """new java.util.Random(seed).nextLong() (JDK API specification: 48-bit LCG, two signed 32-bit draws)."""
    mult, mask = 0x5DEECE66D, (1 << 48) - 1
    st = (seed ^ mult) & mask
    out = []
    for _ in range(2):
        st = (st * mult + 0xB) & mask
        v = st >> 16
        out.append(v - (1 << 32) if v >= 1 << 31 else v)
    v = ((out[0] << 32) + out[1]) & ((1 << 64) - 1)
    return v - (1 << 64) if v >= 1 << 63 else v


def fr_random(seed: int = SEED) -> int:
    """Fp.random (algebra/fields/Fp.java:72-80): new Fp(new Random(seed).nextLong()) — reduced mod r by the
    constructor, so every draw with the same seed is the same element."""
    return _java_random_next_long(seed) % FR


def lowest_power_of_two(n: int) -> int:
    """common/MathUtils.java:20-41."""
    r = 1
    while r < n:
        r <<= 1
    return r


def _le32(values) -> bytes:
    return b"".join(int(v).to_bytes(32, "little") for v in values)


def _upload(a: np.ndarray) -> torch.Tensor:
    """A host TEMPORARY to the device.  Large ones go through pinned memory: handed a pageable range, the HIP runtime
    registers it with the kernel driver, and when the freed range is later recycled and unmapped the driver evicts all
    GPU queues of the process for 10-30 ms — a stall that would land in some later proof."""
    a = np.ascontiguousarray(a)
    t = torch.from_numpy(a if a.flags.writeable else a.copy())
    return (t.pin_memory() if t.numel() * t.element_size() >= (1 << 20) else t).cuda()


def _dev_bytes(b: bytes) -> torch.Tensor:
    return _upload(np.frombuffer(b, dtype=np.uint8))


def _ptr(t):
    return int(t.data_ptr())


def _stream():
    return int(torch.cuda.current_stream().cuda_stream)


def g1_wire(P) -> bytes:
    return _le32(P)


def g2_wire(P) -> bytes:
    return _le32(P[i][j] for i in range(3) for j in range(2))


def wire_out_to_in(out: torch.Tensor, type_: int) -> torch.Tensor:
    """One point in the natives' return layout (64-byte LE coordinates, upper half zero) -> the 32-byte
    coordinates they take as input (VariableBaseMSM.java:221-228 vs :239-258)."""
    k = 3 if type_ == 1 else 6
    return out.view(k, 64)[:, :32].reshape(-1).contiguous()


# ---------------------------------------------------------------------------- R1CS (CSR arrays)
class LinearCombinations:
    """The A (or B, or C) side of all constraints: row i = terms ptr[i] .. ptr[i+1] of (index, value).
    value is None when every coefficient is `one` (the synthetic circuits)."""

    def __init__(self, ptr, index, value=None):
        self.ptr = np.asarray(ptr, dtype=np.int64)
        self.index = np.asarray(index, dtype=np.int64)
        self.value = value  # None, or a numpy object array of Python ints

    @property
    def rows(self):
        return len(self.ptr) - 1

    def row_of_term(self):
        return np.repeat(np.arange(self.rows, dtype=np.int64), np.diff(self.ptr))

    def evaluate(self, full_obj):
        """LinearCombination.evaluate (relations/objects/LinearCombination.java:39-50) for every row: a term
        with index 0 contributes `one` whatever its coefficient."""
        vals = full_obj[self.index]
        if self.value is not None:
            vals = vals * self.value
        vals[self.index == 0] = 1
        out = np.zeros(self.rows, dtype=object)
        nz = np.diff(self.ptr) > 0
        if nz.any():
            red = np.add.reduceat(vals, self.ptr[:-1][nz])
            out[nz] = red
        return out % FR


class R1CSRelation:
    def __init__(self, A, B, C, num_inputs, num_auxiliary):
        self.A, self.B, self.C = A, B, C
        self.num_inputs, self.num_auxiliary = num_inputs, num_auxiliary
        self.num_variables = num_inputs + num_auxiliary
        self.num_constraints = A.rows


def serial_construct(num_constraints: int, num_inputs: int, seed: int = SEED):
    """R1CSConstruction.serialConstruct (R1CSConstruction.java:28-110): the alternating a + b = c / a * b = c
    chain, closed by one constraint (sum x_i)^2 = x_last.  Returns (r1cs, primary, auxiliary) with the
    assignments as lists of ints."""
    assert num_inputs <= num_constraints + 1
    nc = num_constraints
    num_auxiliary = 3 + nc - num_inputs
    nv = num_inputs + num_auxiliary
    a = fr_random(seed)
    b = fr_random(seed)
    full = [1, a, b]
    for i in range(nc - 1):
        tmp = (a * b) % FR if i % 2 else (a + b) % FR
        a, b = b, tmp
        full.append(tmp)
    res = sum(full[1:nv - 1]) % FR
    full.append(res * res % FR)
    i = np.arange(nc - 1, dtype=np.int64)
    even = (i % 2) == 0
    tail = np.arange(1, nv - 1, dtype=np.int64)
    # A: [i+1, i+2] on even rows (a + b), [i+1] on odd rows (a * b); last row: all of 1 .. nv-2
    a_cnt = np.where(even, 2, 1)
    a_ptr = np.concatenate(([0], np.cumsum(a_cnt)))
    a_idx = np.empty(int(a_ptr[-1]), dtype=np.int64)
    a_idx[a_ptr[:-1]] = i + 1
    a_idx[a_ptr[:-1][even] + 1] = i[even] + 2
    A = LinearCombinations(np.concatenate((a_ptr, [a_ptr[-1] + len(tail)])), np.concatenate((a_idx, tail)))
    # B: [0] on even rows, [i+2] on odd rows; last row as A
    b_idx = np.where(even, 0, i + 2)
    B = LinearCombinations(np.concatenate((np.arange(nc, dtype=np.int64), [nc - 1 + len(tail)])),
                           np.concatenate((b_idx, tail)))
    # C: [i+3]; last row [nv-1]
    C = LinearCombinations(np.arange(nc + 1, dtype=np.int64), np.concatenate((i + 3, [nv - 1])))
    r1cs = R1CSRelation(A, B, C, num_inputs, num_auxiliary)
    assert len(full) == nv
    return r1cs, full[:num_inputs], full[num_inputs:]


def constraint_evaluations(r1cs: R1CSRelation, full):
    """The three vectors R1CStoQAP.R1CStoQAPWitness fills before its transforms (R1CStoQAP.java:143-160,
    195-199): evaluations of A, B, C on the domain, with the extra constraints input_i * 0 = 0."""
    nc, ni = r1cs.num_constraints, r1cs.num_inputs
    m = lowest_power_of_two(nc + ni)
    z = np.array(full, dtype=object)
    ev = []
    for k, lc in enumerate((r1cs.A, r1cs.B, r1cs.C)):
        v = np.zeros(m, dtype=object)
        v[:nc] = lc.evaluate(z)
        if k == 0:
            v[nc:nc + ni] = z[:ni]
        ev.append(v)
    return ev, m


def assignment_bytes(full) -> np.ndarray:
    """The assignment as the natives take scalars: 32-byte little-endian elements (the marshalling the Java does
    per MSM with bigIntegerToByteArrayHelperCGBN, VariableBaseMSM.java:121-131,221-228)."""
    return np.frombuffer(_le32(full), dtype=np.uint8).copy()


class R1CSDevice:
    """The three constraint matrices resident in HBM as CSR (u32 row offsets / variable indices, optional 32-byte
    coefficients), with the rows R1CStoQAPWitness adds: A gets `input_i * 0 = 0` rows behind the constraints
    (R1CStoQAP.java:149-151) and all three are padded with empty rows to the domain size, so that
    ozk_r1cs_evaluate_dev leaves exactly the vectors the transforms start from."""

    def __init__(self, r1cs: R1CSRelation):
        nc, ni = r1cs.num_constraints, r1cs.num_inputs
        self.m = m = lowest_power_of_two(nc + ni)
        self.mats = []
        for k, lc in enumerate((r1cs.A, r1cs.B, r1cs.C)):
            ptr, idx, val = lc.ptr, lc.index, lc.value
            if k == 0:   # A[nc + i] = z_i
                ptr = np.concatenate((ptr, ptr[-1] + 1 + np.arange(ni, dtype=np.int64)))
                idx = np.concatenate((idx, np.arange(ni, dtype=np.int64)))
                if val is not None:
                    val = np.concatenate((val, np.ones(ni, dtype=object)))
            ptr = np.concatenate((ptr, np.full(m + 1 - len(ptr), ptr[-1], dtype=np.int64)))
            assert len(ptr) == m + 1 and ptr[-1] == len(idx) < 1 << 32
            long_rows = np.nonzero(np.diff(ptr) > 64)[0].astype(np.uint32)
            d = dict(ptr=_upload(ptr.astype(np.uint32).view(np.uint8)),
                     idx=_upload(idx.astype(np.uint32).view(np.uint8)),
                     coeff=None if val is None else _dev_bytes(_le32(int(v) % FR for v in val)),
                     long=_upload(long_rows.view(np.uint8)) if len(long_rows) else None,
                     n_long=len(long_rows))
            self.mats.append(d)
        self.out = [torch.empty(m * 32, dtype=torch.uint8, device="cuda") for _ in range(3)]
        self.ws_bytes = int(_lib.load().ozk_r1cs_evaluate_workspace_bytes(max(d["n_long"] for d in self.mats)))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device="cuda")

    def evaluate(self, d_full):
        """d_full: the assignment in HBM (num_variables x 32 B).  Asynchronous on the current stream; returns the
        three evaluation vectors (m x 32 B each)."""
        L = _lib.load()
        for d, out in zip(self.mats, self.out):
            _lib.check(L.ozk_r1cs_evaluate_dev(_ptr(d["ptr"]), _ptr(d["idx"]), _ptr(d["coeff"]) if d["coeff"] is not None else None,
                                               _ptr(d_full), self.m, _ptr(d["long"]) if d["long"] is not None else None,
                                               d["n_long"], _ptr(out), _ptr(self.ws), self.ws_bytes, _stream()))
        return self.out


def is_satisfied(r1cs: R1CSRelation, primary, auxiliary) -> bool:
    ev, _ = constraint_evaluations(r1cs, list(primary) + list(auxiliary))
    nc = r1cs.num_constraints
    return bool(np.all((ev[0][:nc] * ev[1][:nc] - ev[2][:nc]) % FR == 0))


# ---------------------------------------------------------------------------- QAP instance at t (host)
def _batch_inverse(xs):
    """Montgomery's trick over Python ints (the Java calls BigInteger.modInverse once per element)."""
    n = len(xs)
    pref = [1] * (n + 1)
    acc = 1
    for i, x in enumerate(xs):
        acc = acc * x % FR
        pref[i + 1] = acc
    inv = pow(acc, -1, FR)
    out = [0] * n
    for i in range(n - 1, -1, -1):
        out[i] = inv * pref[i] % FR
        inv = inv * xs[i] % FR
    return out


def lagrange_coefficients(t: int, m: int):
    """FFTAuxiliary.serialRadix2LagrangeCoefficients (FFTAuxiliary.java:250-302)."""
    if m == 1:
        return [1]
    omega = root_of_unity(m)
    if pow(t, m, FR) == 1:   # t is a domain element: one coefficient is 1
        out, w = [0] * m, 1
        for i in range(m):
            if w == t:
                out[i] = 1
                return out
            w = w * omega % FR
    Z = (pow(t, m, FR) - 1) % FR
    l = Z * pow(m, -1, FR) % FR
    ls, ds, r = [], [], 1
    for _ in range(m):
        ls.append(l)
        ds.append((t - r) % FR)
        l = l * omega % FR
        r = r * omega % FR
    inv = _batch_inverse(ds)
    return [a * b % FR for a, b in zip(ls, inv)]


class QAPRelation:
    pass


def r1cs_to_qap_relation(r1cs: R1CSRelation, t: int) -> QAPRelation:
    """R1CStoQAP.R1CStoQAPRelation (R1CStoQAP.java:37-98)."""
    nc, ni, nv = r1cs.num_constraints, r1cs.num_inputs, r1cs.num_variables
    m = lowest_power_of_two(nc + ni)
    lag = np.array(lagrange_coefficients(t, m), dtype=object)
    q = QAPRelation()
    out = []
    for k, lc in enumerate((r1cs.A, r1cs.B, r1cs.C)):
        acc = np.zeros(nv, dtype=object)
        if k == 0:
            acc[:ni] = lag[nc:nc + ni]
        contrib = lag[lc.row_of_term()]
        if lc.value is not None:
            contrib = contrib * lc.value
        np.add.at(acc, lc.index, contrib)
        out.append([int(x) for x in (acc % FR)])
    q.At, q.Bt, q.Ct = out
    ht, ti = [], 1
    for _ in range(m + 1):
        ht.append(ti)
        ti = ti * t % FR
    q.Ht = ht
    q.Zt = (pow(t, m, FR) - 1) % FR   # SerialFFT.computeZ (SerialFFT.java:140-142)
    q.t, q.num_inputs, q.num_variables, q.degree = t, ni, nv, m
    return q


# ---------------------------------------------------------------------------- QAP instance at t (device)
def _le32_one(v: int):
    return ctypes.create_string_buffer(int(v % FR).to_bytes(32, "little"), 32)


def _ints_from_dev(t: torch.Tensor):
    raw = bytes(t.cpu().numpy())
    return [int.from_bytes(raw[k:k + 32], "little") for k in range(0, len(raw), 32)]


class R1CSTransposedDevice:
    """The constraint matrices TRANSPOSED, resident in HBM as CSR: row j = the terms (constraint i, coefficient) of
    variable j, with the `input_i * 0 = 0` rows R1CStoQAPRelation adds to A (R1CStoQAP.java:52-55: At[i] gets the
    Lagrange coefficient of constraint numConstraints + i).  Built once per R1CS on the host (a stable sort of the
    terms by variable); At / Bt / Ct at any t are then one sparse product each with the Lagrange vector."""

    def __init__(self, r1cs: R1CSRelation):
        nc, ni, nv = r1cs.num_constraints, r1cs.num_inputs, r1cs.num_variables
        self.nv = nv
        self.mats = []
        for k, lc in enumerate((r1cs.A, r1cs.B, r1cs.C)):
            rows, cols, val = lc.row_of_term(), lc.index, lc.value
            if k == 0:
                rows = np.concatenate((rows, nc + np.arange(ni, dtype=np.int64)))
                cols = np.concatenate((cols, np.arange(ni, dtype=np.int64)))
                if val is not None:
                    val = np.concatenate((val, np.ones(ni, dtype=object)))
            order = np.argsort(cols, kind="stable")
            ptr = np.concatenate(([0], np.cumsum(np.bincount(cols, minlength=nv))))
            idx = rows[order]
            assert len(ptr) == nv + 1 and ptr[-1] == len(idx) < 1 << 32
            long_rows = np.nonzero(np.diff(ptr) > 64)[0].astype(np.uint32)
            self.mats.append(dict(
                ptr=_upload(ptr.astype(np.uint32).view(np.uint8)),
                idx=_upload(idx.astype(np.uint32).view(np.uint8)),
                coeff=None if val is None else _dev_bytes(_le32(int(v) % FR for v in val[order])),
                long=_upload(long_rows.view(np.uint8)) if len(long_rows) else None,
                n_long=len(long_rows)))
        self.ws_bytes = int(_lib.load().ozk_r1cs_evaluate_workspace_bytes(max(d["n_long"] for d in self.mats)))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device="cuda")


class QAPRelationDevice:
    """R1CStoQAP.R1CStoQAPRelation (R1CStoQAP.java:37-98) with At, Bt, Ct, Ht left in HBM (n x 32-byte LE values);
    the integer lists the host version offers (At, Bt, Ct, Ht) are downloaded on first use (tests)."""

    def __init__(self):
        self._cache = {}

    def _ints(self, name):
        if name not in self._cache:
            self._cache[name] = _ints_from_dev(getattr(self, "d_" + name))
        return self._cache[name]

    At = property(lambda self: self._ints("At"))
    Bt = property(lambda self: self._ints("Bt"))
    Ct = property(lambda self: self._ints("Ct"))
    Ht = property(lambda self: self._ints("Ht"))


def r1cs_to_qap_relation_dev(r1cs: R1CSRelation, t: int, transposed: R1CSTransposedDevice = None) -> QAPRelationDevice:
    """The QAP instance at t on the device: Lagrange coefficients (ozk_qap_lagrange_dev, FFTAuxiliary.java:250-302),
    three sparse products over the transposed matrices (ozk_sparse_mat_vec_dev), the powers of t
    (ozk_fr_powers_dev).  Falls back to the host version when t lies in the domain (t^m = 1: the reference's
    indicator branch, FFTAuxiliary.java:272-283; probability m / r for a random t)."""
    L = _lib.load()
    nc, ni, nv = r1cs.num_constraints, r1cs.num_inputs, r1cs.num_variables
    m = lowest_power_of_two(nc + ni)
    if m < 2 or pow(t, m, FR) == 1:
        return r1cs_to_qap_relation(r1cs, t)
    T = transposed if transposed is not None else R1CSTransposedDevice(r1cs)
    q = QAPRelationDevice()
    st = _stream()
    d_lag = torch.empty(m * 32, dtype=torch.uint8, device="cuda")
    d_zt = torch.empty(32, dtype=torch.uint8, device="cuda")
    wsb = int(L.ozk_qap_lagrange_workspace_bytes(m))
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    tb, ob = _le32_one(t), _le32_one(root_of_unity(m))
    _lib.check(L.ozk_qap_lagrange_dev(ctypes.cast(tb, ctypes.c_void_p), ctypes.cast(ob, ctypes.c_void_p), m, _ptr(d_lag),
                                      _ptr(d_zt), _ptr(ws), wsb, st))
    outs = []
    for d in T.mats:
        out = torch.empty(nv * 32, dtype=torch.uint8, device="cuda")
        _lib.check(L.ozk_sparse_mat_vec_dev(_ptr(d["ptr"]), _ptr(d["idx"]), _ptr(d["coeff"]) if d["coeff"] is not None else None,
                                            _ptr(d_lag), nv, _ptr(d["long"]) if d["long"] is not None else None,
                                            d["n_long"], _ptr(out), _ptr(T.ws), T.ws_bytes, st))
        outs.append(out)
    q.d_At, q.d_Bt, q.d_Ct = outs
    q.d_Ht = torch.empty((m + 1) * 32, dtype=torch.uint8, device="cuda")
    pwb = int(L.ozk_fr_powers_workspace_bytes(m + 1))
    pws = torch.empty(pwb, dtype=torch.uint8, device="cuda")
    one = _le32_one(1)
    _lib.check(L.ozk_fr_powers_dev(ctypes.cast(tb, ctypes.c_void_p), ctypes.cast(one, ctypes.c_void_p), m + 1, _ptr(q.d_Ht),
                                   _ptr(pws), pwb, st))
    torch.cuda.current_stream().synchronize()       # the host buffers and workspaces above die here
    q.Zt = int.from_bytes(bytes(d_zt.cpu().numpy()), "little")
    q.d_lagrange = d_lag
    q.t, q.num_inputs, q.num_variables, q.degree = t, ni, nv, m
    return q


# ---------------------------------------------------------------------------- fixed-base batches on the device
def _num_windows(scalar_size, window_size):
    return scalar_size // window_size if scalar_size % window_size == 0 else scalar_size // window_size + 1


def batch_msm_dev(scalar_size: int, window_size: int, base_wire: bytes, scalars, type_: int) -> torch.Tensor:
    """FixedBaseMSM.batchMSM (FixedBaseMSM.java:186-315) with the result left in HBM in the variable-base
    wire-in format: uint8 tensor n x 96 (G1) / n x 192 (G2)."""
    L = _lib.load()
    on_device = isinstance(scalars, torch.Tensor)   # n x 32-byte LE values already in HBM, or a list of ints
    n = scalars.numel() // 32 if on_device else len(scalars)
    outerc = (scalar_size + window_size - 1) // window_size   # FixedBaseMSM.java:212
    if outerc * window_size < 254:
        raise _lib.OzkError("window plan covers %d bits of a 254-bit scalar" % (outerc * window_size))
    d_base = _dev_bytes(base_wire)
    d_sc = scalars if on_device else _dev_bytes(_le32(scalars))
    out = torch.empty(n * (96 if type_ == 1 else 192), dtype=torch.uint8, device="cuda")
    wsb = int(L.ozk_fixed_batch_msm_workspace_bytes(outerc, window_size, n, type_))
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    # (per-call window table: the five batches of ONE setup differ in size, hence in the table window the library
    # chooses, so the table cache of ozk_fixed_batch_msm_base_dev would only add its allocations here — 22 -> 85 ms)
    _lib.check(L.ozk_fixed_batch_msm_compact_dev(outerc, window_size, n, _ptr(d_base), _ptr(d_sc), type_, _ptr(out),
                                                 _ptr(ws), wsb, _stream()))
    torch.cuda.current_stream().synchronize()   # ws / d_sc die here
    return out


def _bit_size(wire: bytes) -> int:
    """BNG1.bitSize / BNG2.bitSize (BNG1.java:174-176): the longest coordinate."""
    return max(int.from_bytes(wire[k:k + 32], "little").bit_length() for k in range(0, len(wire), 32))


class _LazyScalars(dict):
    """the scalars behind the key elements, as lists of ints; device-resident ones are downloaded on first use"""

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if isinstance(v, torch.Tensor):
            v = _ints_from_dev(v)
            dict.__setitem__(self, k, v)
        return v


class ProvingKey:
    """zk_proof_systems/zkSNARK/objects/ProvingKey.java, every group element resident in HBM in the
    variable-base wire-in format."""


class CRS:
    pass


def serial_setup_generate(r1cs: R1CSRelation, seed: int = SEED, log=None) -> CRS:
    """SerialSetup.generate (SerialSetup.java:32-192) without the pairing of the verification key."""
    tm = {}
    t0 = time.perf_counter()
    t = alpha = beta = gamma = delta = fr_random(seed)          # :40-44
    inv_gamma, inv_delta = pow(gamma, -1, FR), pow(delta, -1, FR)
    on_device = os.environ.get("OZK_SETUP_HOST", "0") != "1"
    if on_device:
        # the transposed matrices are a property of the R1CS (built once, like the CSR arrays of R1CSDevice), not of t
        if getattr(r1cs, "_transposed_dev", None) is None:
            r1cs._transposed_dev = R1CSTransposedDevice(r1cs)
            torch.cuda.synchronize()
        tm["r1cs_transpose_once_host_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        qap = r1cs_to_qap_relation_dev(r1cs, t, r1cs._transposed_dev)                          # :50
    else:
        qap = r1cs_to_qap_relation(r1cs, t)
    on_device = isinstance(qap, QAPRelationDevice)
    if on_device:
        torch.cuda.synchronize()
    tm["qap_relation_%s_s" % ("device" if on_device else "host")] = time.perf_counter() - t0
    ni, nv = qap.num_inputs, qap.num_variables
    if on_device:
        # (beta At + alpha Bt + Ct) / gamma for the inputs, / delta for the rest (:61-74), and the non-zero counts
        # behind the window sizes (:76-88), without the values leaving HBM
        L = _lib.load()
        k3 = torch.empty(96, dtype=torch.uint8, device="cuda")
        d_gamma_abc = torch.empty(ni * 32, dtype=torch.uint8, device="cuda")
        d_delta_abc = torch.empty((nv - ni) * 32, dtype=torch.uint8, device="cuda")
        kb, ka = _le32_one(beta), _le32_one(alpha)
        for lo, cnt, kk, out in ((0, ni, inv_gamma, d_gamma_abc), (ni, nv - ni, inv_delta, d_delta_abc)):
            if cnt > 0:
                kkb = _le32_one(kk)
                _lib.check(L.ozk_fr_lincomb3_dev(_ptr(qap.d_At) + 32 * lo, _ptr(qap.d_Bt) + 32 * lo, _ptr(qap.d_Ct) + 32 * lo,
                                                 cnt, ctypes.cast(kb, ctypes.c_void_p), ctypes.cast(ka, ctypes.c_void_p),
                                                 ctypes.cast(kkb, ctypes.c_void_p), _ptr(out), _ptr(k3), _stream()))
                torch.cuda.current_stream().synchronize()
        gamma_abc, delta_abc = d_gamma_abc, d_delta_abc
        non_zero_at = int((qap.d_At.view(nv, 32) != 0).any(dim=1).sum().item())
        non_zero_bt = int((qap.d_Bt.view(nv, 32) != 0).any(dim=1).sum().item())
    else:
        abc = [(beta * a + alpha * b + c) % FR for a, b, c in zip(qap.At, qap.Bt, qap.Ct)]
        gamma_abc = [x * inv_gamma % FR for x in abc[:ni]]          # :61-66
        delta_abc = [x * inv_delta % FR for x in abc[ni:]]          # :69-74
        non_zero_at = sum(1 for x in qap.At if x)                   # :76-88
        non_zero_bt = sum(1 for x in qap.Bt if x)
    # :91-112 generators = one * random, window sizes from the per-curve tables
    rnd = fr_random(seed)
    gen_g1 = bytes(batch_msm_dev(254, 16, g1_wire(G1_ONE), [rnd], 1).cpu().numpy())
    gen_g2 = bytes(batch_msm_dev(254, 16, g2_wire(G2_ONE), [rnd], 2).cpu().numpy())
    scalar_size_g1, scalar_size_g2 = _bit_size(gen_g1), _bit_size(gen_g2)
    window_g1 = get_window_size(non_zero_at + non_zero_bt + nv, G1_WINDOW_TABLE)
    window_g2 = get_window_size(non_zero_bt, G2_WINDOW_TABLE)
    t1 = time.perf_counter()

    def b1(scalars):
        return batch_msm_dev(scalar_size_g1, window_g1, gen_g1, scalars, 1)

    def b2(scalars):
        return batch_msm_dev(scalar_size_g2, window_g2, gen_g2, scalars, 2)

    pk = ProvingKey()
    k1 = b1([alpha, beta, delta])                               # :117-121
    pk.alpha_g1, pk.beta_g1, pk.delta_g1 = k1[:96], k1[96:192], k1[192:288]
    k2 = b2([beta, delta, gamma])
    pk.beta_g2, pk.delta_g2, gamma_g2 = k2[:192], k2[192:384], k2[384:576]
    pk.delta_abc_g1 = b1(delta_abc)                             # :123-126
    pk.query_a = b1(qap.d_At if on_device else qap.At)          # :128-131
    pk.query_b_g1 = b1(qap.d_Bt if on_device else qap.Bt)       # :133-144 doubleBatchMSM: G1 and G2 over Bt
    pk.query_b_g2 = b2(qap.d_Bt if on_device else qap.Bt)
    inv_delta_zt = qap.Zt * inv_delta % FR                      # :146-151
    if on_device:
        L = _lib.load()
        ht_scalars = torch.empty((qap.degree + 1) * 32, dtype=torch.uint8, device="cuda")
        pwb = int(L.ozk_fr_powers_workspace_bytes(qap.degree + 1))
        pws = torch.empty(pwb, dtype=torch.uint8, device="cuda")
        tb, kb2 = _le32_one(t), _le32_one(inv_delta_zt)
        _lib.check(L.ozk_fr_powers_dev(ctypes.cast(tb, ctypes.c_void_p), ctypes.cast(kb2, ctypes.c_void_p), qap.degree + 1,
                                       _ptr(ht_scalars), _ptr(pws), pwb, _stream()))
        torch.cuda.current_stream().synchronize()
    else:
        ht_scalars = [h * inv_delta_zt % FR for h in qap.Ht]
    pk.query_h = b1(ht_scalars)
    pk.r1cs = r1cs
    crs = CRS()
    crs.proving_key = pk
    crs.gamma_g2 = gamma_g2                                     # :160-164
    crs.gamma_abc_g1 = b1(gamma_abc)
    tm["fixed_base_gpu_s"] = time.perf_counter() - t1
    # kept for checks in the exponent (tests): the scalars behind every key element
    crs.qap = qap
    crs.secrets = dict(t=t, alpha=alpha, beta=beta, gamma=gamma, delta=delta, generator=rnd)
    crs.scalars = _LazyScalars(delta_abc=delta_abc, gamma_abc=gamma_abc, ht=ht_scalars)
    crs.gen_g1, crs.gen_g2 = gen_g1, gen_g2
    crs.window_g1, crs.window_g2 = window_g1, window_g2
    crs.scalar_size_g1, crs.scalar_size_g2 = scalar_size_g1, scalar_size_g2
    crs.timing = tm
    if log:
        log("setup: QAP instance (%s) %.3f s, fixed-base batches (GPU, incl. marshalling) %.2f s"
            % ("device" if on_device else "host", tm.get("qap_relation_device_s", tm.get("qap_relation_host_s")),
               tm["fixed_base_gpu_s"]))
    return crs


# ---------------------------------------------------------------------------- prover
class _G1Pipeline:
    """Several G1 MSMs of different lengths in flight over ONE workspace: heads back to back on the caller's
    stream, each tail on a side stream out of its own tail buffer (device.VarMsmPipeline generalised to a
    length per submission)."""

    def __init__(self, sizes, depth=2):
        L = _lib.load()
        self.depth = depth
        self.ws_bytes = max(int(L.ozk_var_msm_head_workspace_bytes(n, 1)) for n in sizes)
        self.tail_bytes = max(int(L.ozk_var_msm_tail_bytes(n, 1)) for n in sizes)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device="cuda")
        self.tails = [torch.empty(self.tail_bytes, dtype=torch.uint8, device="cuda") for _ in range(depth)]
        self.full_ws_bytes = max(int(L.ozk_var_msm_workspace_bytes(n, 1)) for n in sizes)
        self.full_ws = torch.empty(self.full_ws_bytes, dtype=torch.uint8, device="cuda")   # for submit(last=True)
        self.side = torch.cuda.Stream()
        self.head_done = [torch.cuda.Event() for _ in range(depth)]
        self.tail_done = [torch.cuda.Event() for _ in range(depth)]
        self.levels_done = []
        for _ in range(depth):
            ev = ctypes.c_void_p()
            _lib.check(L.ozk_order_event_create(ctypes.byref(ev)))
            self.levels_done.append(ev)
        self.count = 0

    def submit(self, d_prepared, d_scalars, n, out, last=False):
        """out: uint8[192] tensor that receives the result (valid once the returned event has fired).
        last=True: nothing follows this MSM in the pipeline, so most of its bucket accumulation and all of its tail
        run with the chip to themselves — the single-call entry point then does better than head + tail: its level-1
        launch is a whole number of rounds of the chip and its tail has the latency shape (a 2^20-constraint proof's H
        MSM: 2.49 + 0.22 + 1.1 ms -> see DESIGN.md section 8)."""
        L = _lib.load()
        slot = self.count % self.depth
        main = torch.cuda.current_stream()
        if self.count >= self.depth:
            main.wait_event(self.tail_done[slot])
        if last and os.environ.get("OZK_PROVER_LAST_LONE", "1") != "0":
            _lib.check(L.ozk_var_msm_prepared_dev(_ptr(d_prepared), _ptr(d_scalars), n, 1, _ptr(out), _ptr(self.full_ws),
                                                  self.full_ws_bytes, int(main.cuda_stream)))
            self.tail_done[slot].record(main)
            self.count += 1
            return self.tail_done[slot]
        prev = self.levels_done[(self.count - 1) % self.depth] if self.count else None
        _lib.check(L.ozk_var_msm_head_prepared_dev(_ptr(d_prepared), _ptr(d_scalars), n, 1, _ptr(self.ws), self.ws_bytes,
                                                   _ptr(self.tails[slot]), self.tail_bytes, int(main.cuda_stream), prev))
        self.head_done[slot].record(main)
        self.side.wait_event(self.head_done[slot])
        # throughput shape of the window sums: the proof keeps the vector ALU busy from start to end, so the additions
        # the serial levels save are worth more than the dependent additions they add (include/ozk.h)
        _lib.check(L.ozk_var_msm_tail_mode_dev(n, 1, _ptr(self.tails[slot]), self.tail_bytes, _ptr(out),
                                               int(self.side.cuda_stream), self.levels_done[slot], 1))
        self.tail_done[slot].record(self.side)
        self.count += 1
        return self.tail_done[slot]

    def close(self):
        if self.levels_done:
            L = _lib.load()
            torch.cuda.synchronize()
            for ev in self.levels_done:
                L.ozk_order_event_destroy(ev)
            self.levels_done = []


class _G1Pipeline3:
    """The three-stage form (device.VarMsmPipeline3 generalised to a length per submission): the SORT of an MSM on the
    caller's stream, its bucket ACCUMULATION on a second, its TAIL on a third / fourth — so the sort of MSM k + 1 runs
    beside the accumulation of MSM k instead of after it.  Same interface as _G1Pipeline."""

    def __init__(self, sizes, depth=4, tail_streams=2):
        L = _lib.load()
        self.depth = depth
        sb = swb = awb = 0
        for n in sizes:
            a, b, c = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
            _lib.check(L.ozk_var_msm_stage_bytes(n, 1, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
            sb, swb, awb = max(sb, a.value), max(swb, b.value), max(awb, c.value)
        self.sorted_bytes, self.sort_ws_bytes, self.accum_ws_bytes = sb, swb, awb
        self.tail_bytes = max(int(L.ozk_var_msm_tail_bytes(n, 1)) for n in sizes)
        buf = lambda b: torch.empty(b, dtype=torch.uint8, device="cuda")
        self.sorted = [buf(sb) for _ in range(2)]
        self.sort_ws, self.accum_ws = buf(swb), buf(awb)
        self.tails = [buf(self.tail_bytes) for _ in range(depth)]
        self.full_ws_bytes = max(int(L.ozk_var_msm_workspace_bytes(n, 1)) for n in sizes)
        self.full_ws = buf(self.full_ws_bytes)
        self.acc = torch.cuda.Stream()
        self.tail_st = [torch.cuda.Stream() for _ in range(tail_streams)]
        ev = lambda k: [torch.cuda.Event() for _ in range(k)]
        self.sort_done, self.accum_done, self.tail_done = ev(2), ev(2), ev(depth)
        self.count = 0

    def submit(self, d_prepared, d_scalars, n, out, last=False):
        L = _lib.load()
        k = self.count
        s, slot = k % 2, k % self.depth
        main = torch.cuda.current_stream()
        if last and os.environ.get("OZK_PROVER_LAST_LONE", "1") != "0":
            # the whole MSM on the accumulate stream, behind the accumulations already queued (see _G1Pipeline.submit)
            ready = torch.cuda.Event()
            ready.record(main)
            self.acc.wait_event(ready)
            if k >= self.depth:
                self.acc.wait_event(self.tail_done[slot])
            _lib.check(L.ozk_var_msm_prepared_dev(_ptr(d_prepared), _ptr(d_scalars), n, 1, _ptr(out), _ptr(self.full_ws),
                                                  self.full_ws_bytes, int(self.acc.cuda_stream)))
            self.tail_done[slot].record(self.acc)
            self.count += 1
            return self.tail_done[slot]
        if k >= 2:
            main.wait_event(self.accum_done[s])
        _lib.check(L.ozk_var_msm_sort_prepared_dev(_ptr(d_prepared), _ptr(d_scalars), n, 1, _ptr(self.sorted[s]),
                                                   self.sorted_bytes, _ptr(self.sort_ws), self.sort_ws_bytes,
                                                   int(main.cuda_stream)))
        self.sort_done[s].record(main)
        self.acc.wait_event(self.sort_done[s])
        if k >= self.depth:
            self.acc.wait_event(self.tail_done[slot])
        _lib.check(L.ozk_var_msm_accum_prepared_dev(_ptr(d_prepared), n, 1, _ptr(self.sorted[s]), self.sorted_bytes,
                                                    _ptr(self.accum_ws), self.accum_ws_bytes, _ptr(self.tails[slot]),
                                                    self.tail_bytes, int(self.acc.cuda_stream)))
        self.accum_done[s].record(self.acc)
        T = self.tail_st[k % len(self.tail_st)]
        T.wait_event(self.accum_done[s])
        _lib.check(L.ozk_var_msm_tail_mode_dev(n, 1, _ptr(self.tails[slot]), self.tail_bytes, _ptr(out),
                                               int(T.cuda_stream), None, 1))
        self.tail_done[slot].record(T)
        self.count += 1
        return self.tail_done[slot]

    def close(self):
        pass


class Proof:
    """zk_proof_systems/zkSNARK/objects/Proof.java: gA (G1), gB (G2), gC (G1) — wire-out bytes."""

    def __init__(self, a, b, c):
        self.g_a, self.g_b, self.g_c = a, b, c


class SerialProver:
    """SerialProver.prove (SerialProver.java:26-119) over a proving key resident in HBM.  Construct once per
    key (prepares the bases), call prove() per witness."""

    def __init__(self, pk: ProvingKey):
        L = _lib.load()
        self.pk = pk
        r1cs = pk.r1cs
        self.ni, self.nv = r1cs.num_inputs, r1cs.num_variables
        self.nw = self.nv - self.ni
        self.m = lowest_power_of_two(r1cs.num_constraints + self.ni)
        ni, nw, m = self.ni, self.nw, self.m
        assert pk.query_h.numel() == (m + 1) * 96 and pk.query_a.numel() == self.nv * 96

        def prep(d_bases, n, type_):
            nb = int(L.ozk_var_msm_prepared_bytes(n, type_))
            out = torch.empty(nb, dtype=torch.uint8, device="cuda")
            _lib.check(L.ozk_var_msm_prepare_dev(_ptr(d_bases), n, type_, _ptr(out), nb, _stream()))
            return out

        # A = alpha + sum z_i A_i(t) + r delta (SerialProver.java:76-79,105): the Java sums a primary-input MSM,
        # an auxiliary-input MSM, alpha and r delta; here ONE MSM over query A ++ [alphaG1, deltaG1] with scalars
        # z ++ [1, r] — the same group element, hence the same affine bytes, without the two short MSMs (each of
        # which costs a whole latency-bound tail) and four additions.  B likewise with beta, delta and s (:82-88,108-110).
        cat = torch.cat
        self.qa = prep(cat((pk.query_a, pk.alpha_g1, pk.delta_g1)), self.nv + 2, 1)
        self.qb1 = prep(cat((pk.query_b_g1, pk.beta_g1, pk.delta_g1)), self.nv + 2, 1)
        self.qb2 = prep(cat((pk.query_b_g2, pk.beta_g2, pk.delta_g2)), self.nv + 2, 2)
        self.qh = prep(pk.query_h, m + 1, 1)
        self.dabc = prep(pk.delta_abc_g1, nw, 1)
        torch.cuda.synchronize()
        # three-stage by default, one tail stream (a 2^20-constraint proof: 15.6-15.7 ms against 15.9 with the two-stage
        # pipeline of round 2, OZK_PROVER_PIPE3=0; two tail streams measure the same as one)
        three = os.environ.get("OZK_PROVER_PIPE3", "1") == "1"
        ts = int(os.environ.get("OZK_PROVER_TAIL_STREAMS", "1"))
        self.pipe = _G1Pipeline3([self.nv + 2, m + 1, nw], tail_streams=ts) if three else _G1Pipeline([self.nv + 2, m + 1, nw])
        self.g2_ws_bytes = int(L.ozk_var_msm_head_workspace_bytes(self.nv + 2, 2))
        self.g2_ws = torch.empty(self.g2_ws_bytes, dtype=torch.uint8, device="cuda")
        self.g2_tail_bytes = int(L.ozk_var_msm_tail_bytes(self.nv + 2, 2))
        self.g2_tail = torch.empty(self.g2_tail_bytes, dtype=torch.uint8, device="cuda")
        self.s_g2 = torch.cuda.Stream()
        # witness map + C's share: dispatched ahead of the MSMs that do not depend on them (the H MSM waits for the map)
        self.s_fin = torch.cuda.Stream(priority=-1)
        self.fin_ws_bytes = int(L.ozk_var_msm_workspace_bytes(3, 1))
        self.fin_ws = torch.empty(self.fin_ws_bytes, dtype=torch.uint8, device="cuda")
        self.q_ws_bytes = int(L.ozk_qap_witness_workspace_bytes(m))
        self.q_ws = torch.empty(self.q_ws_bytes, dtype=torch.uint8, device="cuda")
        self.d_h = torch.empty((m + 1) * 32, dtype=torch.uint8, device="cuda")
        # results: G1 MSM outputs (192 B each) and G2 outputs (384 B)
        self.o1 = torch.zeros(5, 192, dtype=torch.uint8, device="cuda")   # A, B1, deltaABC, H, C's share
        self.o2 = torch.zeros(1, 384, dtype=torch.uint8, device="cuda")   # B
        self.r1cs_dev = R1CSDevice(r1cs)   # the constraint matrices, uploaded once per key
        self.omega = ctypes.create_string_buffer(root_of_unity(m).to_bytes(32, "little"), 32)
        self.g = ctypes.create_string_buffer(FR_MULT_GEN.to_bytes(32, "little"), 32)

    def close(self):
        self.pipe.close()

    def prove(self, primary, auxiliary, seed: int = SEED, timing=None, full_bytes=None) -> Proof:
        """`full_bytes` (optional): the assignment primary ++ auxiliary already marshalled (assignment_bytes) —
        what a caller that keeps its witness as bytes hands over; otherwise it is marshalled here."""
        L = _lib.load()
        pk, ni, nw, m = self.pk, self.ni, self.nw, self.m
        T = {}
        t0 = time.perf_counter()
        if full_bytes is None:
            full_bytes = assignment_bytes(list(primary) + list(auxiliary))
        assert full_bytes.size == self.nv * 32
        T["marshal_assignment_host_ms"] = (time.perf_counter() - t0) * 1e3
        r = fr_random(seed)                                      # SerialProver.java:58-59
        s = fr_random(seed)
        t1 = time.perf_counter()
        d_full = torch.from_numpy(full_bytes).cuda()
        tails = _dev_bytes(_le32([1, r, 1, s, s, r, (FR - r * s % FR) % FR]))
        d_aux = d_full[ni * 32:]
        d_full_r = torch.cat((d_full, tails[:64]))               # z ++ [1, r]
        d_full_s = torch.cat((d_full, tails[64:128]))            # z ++ [1, s]
        d_fin_sc = tails[128:]                                   # [s, r, -rs]
        torch.cuda.synchronize()
        T["upload_ms"] = (time.perf_counter() - t1) * 1e3
        t2 = time.perf_counter()
        main = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(main)
        o1, o2 = self.o1, self.o2
        nv = self.nv
        # B in G2 (doubleMSM, SerialProver.java:82-88): own stream, nothing to wait for but the uploads.  Issued
        # first: its tail is the longest latency-bound chain of the proof and hides under the G1 accumulations.
        self.s_g2.wait_event(ready)
        with torch.cuda.stream(self.s_g2):
            _lib.check(L.ozk_var_msm_head_prepared_dev(_ptr(self.qb2), _ptr(d_full_s), nv + 2, 2, _ptr(self.g2_ws),
                                                       self.g2_ws_bytes, _ptr(self.g2_tail), self.g2_tail_bytes, _stream(),
                                                       None))
            _lib.check(L.ozk_var_msm_tail_mode_dev(nv + 2, 2, _ptr(self.g2_tail), self.g2_tail_bytes, _ptr(o2[0]),
                                                   _stream(), None, 1))
            g2_done = torch.cuda.Event()
            g2_done.record(self.s_g2)
        # witness map (SerialProver.java:36-41): constraint evaluations (R1CStoQAP.java:143-160,195-199) and the
        # seven transforms on the device; coefficientsH stay in HBM.  Own stream: only the H MSM waits for it, the
        # three MSMs over the assignment run beside it.
        self.s_fin.wait_event(ready)
        with torch.cuda.stream(self.s_fin):
            d_ev = self.r1cs_dev.evaluate(d_full)
            _lib.check(L.ozk_qap_witness_dev(_ptr(d_ev[0]), _ptr(d_ev[1]), _ptr(d_ev[2]), m, ctypes.cast(self.omega, ctypes.c_void_p),
                                             ctypes.cast(self.g, ctypes.c_void_p), _ptr(self.d_h), _ptr(self.q_ws),
                                             self.q_ws_bytes, _stream()))
            h_ready = torch.cuda.Event()
            h_ready.record(self.s_fin)
        p = self.pipe
        ev_a = p.submit(self.qa, d_full_r, nv + 2, o1[0])        # :76-79,105  A
        ev_b = p.submit(self.qb1, d_full_s, nv + 2, o1[1])       # :82-88,108-110  B in G1
        # A and B1 are complete once their tails are: their share of C — s A + r B1 - r s delta (:114) — is a 3-term
        # MSM that runs on its own stream while the long MSMs still occupy the pipeline.  (The waits are queued NOW:
        # the pipeline re-records these per-slot events for the next two submissions.)
        self.s_fin.wait_event(ev_a)
        self.s_fin.wait_event(ev_b)
        ev_l = p.submit(self.dabc, d_aux, nw, o1[2])             # :98-101 deltaABC
        main.wait_event(h_ready)
        ev_h = p.submit(self.qh, self.d_h, m + 1, o1[3], last=True)   # :91-93 query H
        with torch.cuda.stream(self.s_fin):
            fin_bases = torch.cat((wire_out_to_in(o1[0], 1), wire_out_to_in(o1[1], 1), pk.delta_g1))
            _lib.check(L.ozk_var_msm_dev(_ptr(fin_bases), _ptr(d_fin_sc), 3, 1, _ptr(o1[4]), _ptr(self.fin_ws),
                                         self.fin_ws_bytes, _stream()))
            fin_done = torch.cuda.Event()
            fin_done.record(self.s_fin)
        main.wait_event(ev_l)
        main.wait_event(ev_h)
        main.wait_event(fin_done)
        # C = evaluationABC + H(t)Z(t)/delta + (s A + r B1 - r s delta)   (:102,:114)
        c_out = torch.zeros(192, dtype=torch.uint8, device="cuda")
        _lib.check(L.ozk_points_sum_dev(_ptr(o1[2:5]), 3, 1, _ptr(c_out), int(main.cuda_stream)))
        main.wait_event(g2_done)
        torch.cuda.synchronize()
        T["gpu_ms"] = (time.perf_counter() - t2) * 1e3
        proof = Proof(bytes(o1[0].cpu().numpy()), bytes(o2[0].cpu().numpy()), bytes(c_out.cpu().numpy()))
        self._keep = (d_full, tails, d_full_r, d_full_s, fin_bases, d_ev)
        if timing is not None:
            timing.update(T)
        return proof

    def coefficients_h(self):
        """coefficientsH of the last prove() (m + 1 ints), for checks."""
        raw = bytes(self.d_h.cpu().numpy())
        return [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(self.m + 1)]
