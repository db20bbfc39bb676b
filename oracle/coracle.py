"""ctypes wrapper of oracle/ozk_oracle.c (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py)."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "ozk_oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libozk_oracle.so")
_lib = None


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if force or not os.path.exists(LIB) or os.path.getmtime(SRC) > os.path.getmtime(LIB):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", LIB, SRC, "-lm"])
    return LIB


def load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _buf(b):
    return ctypes.cast(ctypes.c_char_p(bytes(b)), ctypes.c_void_p)


def pippenger_g1(bases: bytes, scalars: bytes, n: int, num_bits: int = 254) -> bytes:
    out = ctypes.create_string_buffer(192)
    rc = load().oracle_pippenger_g1(_buf(bases), _buf(scalars), n, num_bits, out)
    assert rc == 0
    return out.raw


def naive_g1(bases: bytes, scalars: bytes, n: int) -> bytes:
    out = ctypes.create_string_buffer(192)
    assert load().oracle_naive_g1(_buf(bases), _buf(scalars), n, out) == 0
    return out.raw


def fixed_base_g1(base: bytes, scalars: bytes, n: int, outerc: int, window: int) -> bytes:
    out = ctypes.create_string_buffer(192 * n)
    assert load().oracle_fixed_base_g1(_buf(base), _buf(scalars), n, outerc, window, out) == 0
    return out.raw


def fft_fr(data: bytes, n: int, omega: bytes) -> bytes:
    out = ctypes.create_string_buffer(64 * n)
    assert load().oracle_fft_fr(_buf(data), n, _buf(omega), out) == 0
    return out.raw


def field_batch_mul(data: bytes, n: int) -> bytes:
    out = ctypes.create_string_buffer(64 * n)
    assert load().oracle_field_batch_mul(_buf(data), n, out) == 0
    return out.raw


def field_op(field: int, op: int, a: int, b: int) -> int:
    out = ctypes.create_string_buffer(32)
    load().oracle_field_op(field, op, _buf(a.to_bytes(32, "little")), _buf(b.to_bytes(32, "little")), out)
    return int.from_bytes(out.raw, "little")
