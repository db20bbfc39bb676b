"""Host-side mirror of the reference's algebra.msm.VariableBaseMSM for the JNI path
(VariableBaseMSM.java:199-338 serialMSM, :484-610 doubleMSM): same chunking, same byte
marshalling, the native call replaced by the C ABI of libozk_hip.so.

Points are Jacobian integer triples (X, Y, Z) for G1 and ((x0,x1),(y0,y1),(z0,z1)) for
G2, scalars are Python ints — the Python analogue of the Java BigInteger objects.
The Java prover binds the same C ABI through the JNI shims.
"""
import ctypes

from . import lib as _lib

G1_ITERATION_BATCH = 1 << 23  # VariableBaseMSM.java:211
G2_ITERATION_BATCH = 1 << 22  # VariableBaseMSM.java:268
DOUBLE_ITERATION_BATCH = 1 << 21  # VariableBaseMSM.java:494


def big_integer_to_byte_array_cgbn(v: int) -> bytes:
    """VariableBaseMSM.java:121-131: 32-byte little-endian."""
    return int(v).to_bytes(32, "little")


def _le64(b: bytes) -> int:
    """VariableBaseMSM.java:239-258: reverse the 64 bytes, new BigInteger."""
    return int.from_bytes(b, "little")


def marshal_g1(bases) -> bytes:
    return b"".join(big_integer_to_byte_array_cgbn(c) for P in bases for c in P)


def marshal_g2(bases) -> bytes:
    return b"".join(big_integer_to_byte_array_cgbn(P[i][j]) for P in bases for i in range(3) for j in range(2))


def marshal_scalars(scalars) -> bytes:
    return b"".join(big_integer_to_byte_array_cgbn(s) for s in scalars)


def variable_base_serial_msm_native_helper(bases_xyz: bytes, scalars: bytes, batch_size: int, type_: int,
                                           task_id: int) -> bytes:
    """The JNI native of VariableBaseMSM.java:193-197, on the C ABI."""
    L = _lib.load()
    out = ctypes.create_string_buffer(192 if type_ == 1 else 384)
    b = ctypes.c_char_p(bases_xyz)
    s = ctypes.c_char_p(scalars)
    _lib.check(L.ozk_var_msm_host(ctypes.cast(b, ctypes.c_void_p), ctypes.cast(s, ctypes.c_void_p),
                                  batch_size, type_, task_id, ctypes.cast(out, ctypes.c_void_p)))
    return out.raw


def variable_base_double_msm_native_helper(bases1: bytes, bases2: bytes, scalars: bytes, batch_size: int,
                                           task_id: int) -> bytes:
    """The JNI native of VariableBaseMSM.java:473-478."""
    L = _lib.load()
    out = ctypes.create_string_buffer(576)
    _lib.check(L.ozk_var_double_msm_host(ctypes.cast(ctypes.c_char_p(bases1), ctypes.c_void_p),
                                         ctypes.cast(ctypes.c_char_p(bases2), ctypes.c_void_p),
                                         ctypes.cast(ctypes.c_char_p(scalars), ctypes.c_void_p),
                                         batch_size, task_id, ctypes.cast(out, ctypes.c_void_p)))
    return out.raw


def unmarshal_g1(res: bytes):
    return tuple(_le64(res[64 * i:64 * (i + 1)]) for i in range(3))


def unmarshal_g2(res: bytes):
    v = [_le64(res[64 * i:64 * (i + 1)]) for i in range(6)]
    return ((v[0], v[1]), (v[2], v[3]), (v[4], v[5]))


def serial_msm(scalars, bases, group_add, group_zero, is_g1=True, task_id=0):
    """VariableBaseMSM.serialMSM (VariableBaseMSM.java:199-338): chunk, marshal, native call,
    un-marshal, sum the chunk results with the group's own add (`group_add`, as the Java
    code does with BNG1.add at :261-265)."""
    assert len(bases) == len(scalars)
    step = G1_ITERATION_BATCH if is_g1 else G2_ITERATION_BATCH
    results = []
    for it in range(0, len(scalars), step):
        sc = scalars[it:it + step]
        bs = bases[it:it + step]
        raw = variable_base_serial_msm_native_helper(
            marshal_g1(bs) if is_g1 else marshal_g2(bs), marshal_scalars(sc), len(sc), 1 if is_g1 else 2,
            task_id)
        results.append(unmarshal_g1(raw) if is_g1 else unmarshal_g2(raw))
    final = group_zero
    for g in results:
        final = group_add(final, g)
    return final


class PreparedBases:
    """Bases kept on the GPU across MSMs (include/ozk.h "prepared bases").  The
    reference marshals and uploads the proving-key slice for every call (VariableBaseMSM.java:224-227);
    with a handle only the scalars travel.  Same bytes out as variable_base_serial_msm_native_helper."""

    def __init__(self, bases_xyz: bytes, batch_size: int, type_: int, task_id: int = 0):
        L = _lib.load()
        self._h = ctypes.c_void_p()
        self.n, self.type = batch_size, type_
        _lib.check(L.ozk_bases_create_host(ctypes.cast(ctypes.c_char_p(bases_xyz), ctypes.c_void_p), batch_size, type_,
                                           task_id, ctypes.byref(self._h)))

    def msm(self, scalars: bytes) -> bytes:
        L = _lib.load()
        out = ctypes.create_string_buffer(192 if self.type == 1 else 384)
        _lib.check(L.ozk_var_msm_bases_host(self._h, ctypes.cast(ctypes.c_char_p(scalars), ctypes.c_void_p),
                                            len(scalars) // 32, ctypes.cast(out, ctypes.c_void_p)))
        return out.raw

    def close(self):
        if self._h:
            _lib.load().ozk_bases_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        import sys
        if sys is None or sys.is_finalizing():   # no HIP calls during interpreter teardown
            return
        try:
            self.close()
        except Exception:
            pass
