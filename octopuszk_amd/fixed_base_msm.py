"""Host-side mirror of the reference's algebra.msm.FixedBaseMSM JNI path
(FixedBaseMSM.java:49-66 getWindowSize, :186-315 batchMSM, :489-602 doubleBatchMSM,
:753-785 batchFieldMSMPartition): same parameters, same byte formats; the natives are
the C ABI of libozk_hip.so."""
import ctypes

from . import lib as _lib
from .variable_base_msm import big_integer_to_byte_array_cgbn, marshal_scalars

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


def batch_msm(scalar_size, window_size, base, scalars, is_g1=True, task_id=0):
    """FixedBaseMSM.batchMSM (FixedBaseMSM.java:186-315): list of Jacobian integer points."""
    outerc = (scalar_size + window_size - 1) // window_size
    num_windows = scalar_size // window_size if scalar_size % window_size == 0 else scalar_size // window_size + 1
    if is_g1:
        bw = b"".join(big_integer_to_byte_array_cgbn(c) for c in base)
    else:
        bw = b"".join(big_integer_to_byte_array_cgbn(base[i][j]) for i in range(3) for j in range(2))
    raw = batch_msm_native_helper(outerc, window_size, num_windows, 1 << window_size, len(scalars), scalar_size,
                                  bw, marshal_scalars(scalars), 1 if is_g1 else 2, task_id)
    out = []
    per = 192 if is_g1 else 384
    for i in range(len(scalars)):
        v = [_be64(raw[per * i + 64 * k: per * i + 64 * (k + 1)]) for k in range(per // 64)]
        out.append(tuple(v) if is_g1 else ((v[0], v[1]), (v[2], v[3]), (v[4], v[5])))
    return out
