"""Host-side mirror of the witness half of reductions.r1cs_to_qap.R1CStoQAP for BN254 Fr:
R1CStoQAPWitness (R1CStoQAP.java:139-237) from the constraint evaluations on.  The arithmetic is
the C ABI of libozk_hip.so (`ozk_qap_witness_host`): seven transforms and the pointwise stages on
the GPU; nothing here computes."""
import ctypes

from . import lib as _lib
from .fft import FR, FR_MULT_GEN, root_of_unity


def _le32(v: int) -> bytes:
    return int(v % FR).to_bytes(32, "little")


def qap_witness_native_helper(a_eval: bytes, b_eval: bytes, c_eval: bytes, m: int, omega: bytes, g: bytes,
                              task_id: int) -> bytes:
    """The optional native INTEGRATION.md §5 describes: three m x 32 B LE evaluation vectors in,
    (m + 1) x 32 B LE coefficients of H out."""
    L = _lib.load()
    out = ctypes.create_string_buffer(32 * (m + 1))

    def vp(b):
        return ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p)

    _lib.check(L.ozk_qap_witness_host(vp(a_eval), vp(b_eval), vp(c_eval), m, vp(omega), vp(g), task_id,
                                      ctypes.cast(out, ctypes.c_void_p)))
    return out.raw


def coefficients_h(a_eval, b_eval, c_eval, g: int = FR_MULT_GEN, task_id: int = 0):
    """R1CStoQAP.java:163-230: `a_eval[i]`, `b_eval[i]`, `c_eval[i]` are the evaluations of the
    polynomials A, B, C on the i-th domain point (the lists the Java fills at :143-160,195-199,
    including the extra input_i * 0 = 0 constraints); returns coefficientsH (domainSize + 1 ints)."""
    m = len(a_eval)
    assert m >= 2 and m & (m - 1) == 0 and len(b_eval) == m and len(c_eval) == m
    raw = qap_witness_native_helper(b"".join(map(_le32, a_eval)), b"".join(map(_le32, b_eval)),
                                    b"".join(map(_le32, c_eval)), m, _le32(root_of_unity(m)), _le32(g), task_id)
    return [int.from_bytes(raw[32 * i:32 * (i + 1)], "little") for i in range(m + 1)]
