"""Host-side mirror of the reference's algebra.msm.FixedBaseMSM JNI path
(FixedBaseMSM.java:49-66 getWindowSize, :186-315 batchMSM, :321-443 batchMSMPartition, :489-602 doubleBatchMSM,
:753-785 batchFieldMSMPartition, :788-852 batchFilterFieldMSMPartition): same parameters, same chunking, same byte
formats; the natives are the C ABI of libozk_hip.so."""
import ctypes

from . import lib as _lib
from .variable_base_msm import big_integer_to_byte_array_cgbn, marshal_scalars

G1_ITERATION_BATCH = 1 << 23       # FixedBaseMSM.java:200, :333
G2_ITERATION_BATCH = 1 << 22       # FixedBaseMSM.java:257, :390
DOUBLE_ITERATION_BATCH = 1 << 21   # FixedBaseMSM.java:511
FR_MODULUS = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # BN254aFrParameters

G1_WINDOW_TABLE = [1, 5, 11, 32, 55, 162, 360, 815, 2373, 6978, 7122, 0, 57818, 0, 169679,
                   439759, 936073, 0, 4666555, 7580404, 0, 34552892]  # BN254aG1Parameters.java:25-50
G2_WINDOW_TABLE = [1, 5, 10, 25, 59, 154, 334, 743, 2034, 4988, 8888, 26271, 39768, 106276,
                   141703, 462423, 926872, 0, 4873049, 5706708, 0, 31673815]  # BN254aG2Parameters.java:33-58


def get_window_size(num_scalars: int, table) -> int:
    """FixedBaseMSM.getWindowSize (FixedBaseMSM.java:49-66)."""
    if not table:
        return 17
    window = 1
    for i in range(len(table) - 1, -1, -1):
        if table[i] != 0 and num_scalars >= table[i]:
            window = i + 1
            break
    return window


def _vp(b):
    return ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p)


def batch_msm_native_helper(outerc, window_size, out_len, inner_len, batch_size, scalar_size, base: bytes,
                            scalars: bytes, bn_type: int, task_id: int) -> bytes:
    """JNI native of FixedBaseMSM.java:102-109."""
    L = _lib.load()
    out = ctypes.create_string_buffer(batch_size * (192 if bn_type == 1 else 384))
    _lib.check(L.ozk_fixed_batch_msm_host(outerc, window_size, out_len, inner_len, batch_size, scalar_size,
                                          _vp(base), _vp(scalars), bn_type, task_id,
                                          ctypes.cast(out, ctypes.c_void_p)))
    return out.raw


def double_batch_msm_native_helper(outerc1, window_size1, outerc2, window_size2, out_len1, inner_len1, out_len2,
                                   inner_len2, batch_size, base1: bytes, base2: bytes, scalars: bytes,
                                   task_id: int) -> bytes:
    """JNI native of FixedBaseMSM.java:473-485."""
    L = _lib.load()
    out = ctypes.create_string_buffer(batch_size * 576)
    _lib.check(L.ozk_fixed_double_batch_msm_host(outerc1, window_size1, outerc2, window_size2, out_len1, inner_len1,
                                                 out_len2, inner_len2, batch_size, _vp(base1), _vp(base2),
                                                 _vp(scalars), task_id, ctypes.cast(out, ctypes.c_void_p)))
    return out.raw


def field_batch_msm_native_helper(data: bytes, batch_size: int, task_id: int) -> bytes:
    """JNI native of FixedBaseMSM.java:747-749: (n+1) x 32 B in, n x 64 B big-endian out."""
    L = _lib.load()
    out = ctypes.create_string_buffer(batch_size * 64)
    _lib.check(L.ozk_field_batch_mul_host(_vp(data), batch_size, task_id, ctypes.cast(out, ctypes.c_void_p)))
    return out.raw


def _be64(b: bytes) -> int:
    """FixedBaseMSM.java:233-241: new BigInteger(64-byte big-endian slice)."""
    return int.from_bytes(b, "big")


def _base_wire(base, is_g1):
    if is_g1:
        return b"".join(big_integer_to_byte_array_cgbn(c) for c in base)
    return b"".join(big_integer_to_byte_array_cgbn(base[i][j]) for i in range(3) for j in range(2))


def _points_be(raw, count, is_g1):
    """count points of 3 (G1) / 6 (G2) 64-byte big-endian values (FixedBaseMSM.java:233-246, :286-305)"""
    out = []
    per = 192 if is_g1 else 384
    for i in range(count):
        v = [_be64(raw[per * i + 64 * k: per * i + 64 * (k + 1)]) for k in range(per // 64)]
        out.append(tuple(v) if is_g1 else ((v[0], v[1]), (v[2], v[3]), (v[4], v[5])))
    return out


def batch_msm(scalar_size, window_size, base, scalars, is_g1=True, task_id=0):
    """FixedBaseMSM.batchMSM (FixedBaseMSM.java:186-315): base * s_i for every scalar, as Jacobian integer points;
    chunks of 2^23 (G1) / 2^22 (G2) scalars per native call, the window table rebuilt by each call."""
    outerc = (scalar_size + window_size - 1) // window_size
    num_windows = scalar_size // window_size if scalar_size % window_size == 0 else scalar_size // window_size + 1
    step = G1_ITERATION_BATCH if is_g1 else G2_ITERATION_BATCH
    bw = _base_wire(base, is_g1)
    out = []
    for it in range(0, len(scalars), step):
        sc = scalars[it:it + step]
        raw = batch_msm_native_helper(outerc, window_size, num_windows, 1 << window_size, len(sc), scalar_size,
                                      bw, marshal_scalars(sc), 1 if is_g1 else 2, task_id)
        out.extend(_points_be(raw, len(sc), is_g1))
    return out


def batch_msm_partition(scalar_size, window_size, out_size, in_size, base, indexed_scalars, is_g1=True, task_id=0):
    """FixedBaseMSM.batchMSMPartition (FixedBaseMSM.java:321-443): a partition's list of (index, scalar) tuples ->
    list of (index, point); the same chunking as batchMSM, the partition's taskID handed to the native."""
    outerc = (scalar_size + window_size - 1) // window_size
    step = G1_ITERATION_BATCH if is_g1 else G2_ITERATION_BATCH
    bw = _base_wire(base, is_g1)
    out = []
    for it in range(0, len(indexed_scalars), step):
        part = indexed_scalars[it:it + step]
        raw = batch_msm_native_helper(outerc, window_size, out_size, in_size, len(part), scalar_size, bw,
                                      marshal_scalars([p[1] for p in part]), 1 if is_g1 else 2, task_id)
        pts = _points_be(raw, len(part), is_g1)
        out.extend((part[i][0], pts[i]) for i in range(len(part)))
    return out


def double_batch_msm(out_size1, in_size1, out_size2, in_size2, scalar_size1, window_size1, base_g1, scalar_size2,
                     window_size2, base_g2, scalars):
    """FixedBaseMSM.doubleBatchMSM (FixedBaseMSM.java:489-602): (base_g1 * s_i, base_g2 * s_i) for every scalar.
    Chunks of 2^21 scalars; per element the native returns NINE 64-byte big-endian values —
    G1 (X, Y, Z) then G2 (Xa, Xb, Ya, Yb, Za, Zb) (FixedBaseMSM.java:557-591, algebra_msm_FixedBaseMSM.cu:1479-1482)."""
    outerc1 = (scalar_size1 + window_size1 - 1) // window_size1
    outerc2 = (scalar_size2 + window_size2 - 1) // window_size2
    b1, b2 = _base_wire(base_g1, True), _base_wire(base_g2, False)
    out = []
    for it in range(0, len(scalars), DOUBLE_ITERATION_BATCH):
        sc = scalars[it:it + DOUBLE_ITERATION_BATCH]
        raw = double_batch_msm_native_helper(outerc1, window_size1, outerc2, window_size2, out_size1, in_size1,
                                             out_size2, in_size2, len(sc), b1, b2, marshal_scalars(sc), 0)
        for i in range(len(sc)):
            v = [_be64(raw[(9 * i + k) * 64:(9 * i + k + 1) * 64]) for k in range(9)]
            out.append(((v[0], v[1], v[2]), ((v[3], v[4]), (v[5], v[6]), (v[7], v[8]))))
    return out


def batch_field_msm_partition(base_element, indexed_scalars, task_id=0):
    """FixedBaseMSM.batchFieldMSMPartition (FixedBaseMSM.java:753-785): x_i * base in Fr for a partition's list of
    (index, x_i).  The native's input is the n scalars FOLLOWED BY THE BASE as element n (:765); its output n 64-byte
    big-endian values, which setBigInteger reduces mod r (bn254a/BN254aFields.java:52-54)."""
    data = marshal_scalars([p[1] for p in indexed_scalars]) + big_integer_to_byte_array_cgbn(base_element)
    n = len(indexed_scalars)
    raw = field_batch_msm_native_helper(data, n, task_id)
    return [(indexed_scalars[i][0], _be64(raw[64 * i:64 * (i + 1)]) % FR_MODULUS) for i in range(n)]


def batch_filter_field_msm_partition(inverse_delta, inverse_gamma, num_inputs, indexed_scalars, type_, task_id=0):
    """FixedBaseMSM.batchFilterFieldMSMPartition (FixedBaseMSM.java:788-852): type_ 0 keeps the elements with
    index < num_inputs and multiplies them by inverse_gamma, type_ 1 the others by inverse_delta; one native call
    over the kept elements with the multiplier appended."""
    keep = [p for p in indexed_scalars if (p[0] < num_inputs) == (type_ == 0)]
    mult = inverse_gamma if type_ == 0 else inverse_delta
    if not keep:
        return []
    return batch_field_msm_partition(mult, keep, task_id)


