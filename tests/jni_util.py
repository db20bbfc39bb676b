"""ctypes driver of tests/native/mock_jvm.c (mock JNIEnv) for the JNI shim tests."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "native", "mock_jvm.c")
LIB = os.path.join(HERE, "native", "_mockjvm.so")
PKG = os.path.join(ROOT, "octopuszk_amd")
SHIM_VAR = os.path.join(PKG, "libAlgebraMSMVariableBaseMSM.so")
SHIM_FIXED = os.path.join(PKG, "libAlgebraMSMFixedBaseMSM.so")
SHIM_FFT = os.path.join(PKG, "libAlgebraFFTAuxiliary.so")


class JavaException(Exception):
    pass


def mock():
    deps = [SRC, os.path.join(ROOT, "include", "ozk_jni.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call(["gcc", "-std=c11", "-O2", "-shared", "-fPIC", "-o", LIB, SRC, "-ldl"])
    return ctypes.CDLL(LIB)


def _finish(rc, out, err):
    if rc < 0:
        raise JavaException(err.value.decode())
    return out.raw[:rc]


def var_msm(bases, scalars, n, type_, task=0):
    m = mock()
    out, err = ctypes.create_string_buffer(384), ctypes.create_string_buffer(1024)
    m.mock_var_msm.restype = ctypes.c_long
    rc = m.mock_var_msm(SHIM_VAR.encode(), bases, ctypes.c_long(len(bases)), scalars, ctypes.c_long(len(scalars)),
                        n, type_, task, out, ctypes.c_long(384), err)
    return _finish(rc, out, err)


def var_double_msm(b1, b2, scalars, n, task=0):
    m = mock()
    out, err = ctypes.create_string_buffer(576), ctypes.create_string_buffer(1024)
    m.mock_var_double_msm.restype = ctypes.c_long
    rc = m.mock_var_double_msm(SHIM_VAR.encode(), b1, ctypes.c_long(len(b1)), b2, ctypes.c_long(len(b2)), scalars,
                               ctypes.c_long(len(scalars)), n, task, out, ctypes.c_long(576), err)
    return _finish(rc, out, err)


def fixed_batch(outerc, ws, n, scalar_size, base, scalars, bn, task=0):
    m = mock()
    cap = n * 384
    out, err = ctypes.create_string_buffer(cap), ctypes.create_string_buffer(1024)
    m.mock_fixed_batch.restype = ctypes.c_long
    rc = m.mock_fixed_batch(SHIM_FIXED.encode(), outerc, ws, outerc, 1 << ws, n, scalar_size, base,
                            ctypes.c_long(len(base)), scalars, ctypes.c_long(len(scalars)), bn, task, out,
                            ctypes.c_long(cap), err)
    return _finish(rc, out, err)


def fixed_double_batch(oc1, ws1, oc2, ws2, n, b1, b2, scalars, task=0):
    m = mock()
    cap = n * 576
    out, err = ctypes.create_string_buffer(cap), ctypes.create_string_buffer(1024)
    m.mock_fixed_double_batch.restype = ctypes.c_long
    rc = m.mock_fixed_double_batch(SHIM_FIXED.encode(), oc1, ws1, oc2, ws2, n, b1, ctypes.c_long(len(b1)), b2,
                                   ctypes.c_long(len(b2)), scalars, ctypes.c_long(len(scalars)), task, out,
                                   ctypes.c_long(cap), err)
    return _finish(rc, out, err)


def field_mul(data, n, task=0):
    m = mock()
    out, err = ctypes.create_string_buffer(n * 64), ctypes.create_string_buffer(1024)
    m.mock_field_mul.restype = ctypes.c_long
    rc = m.mock_field_mul(SHIM_FIXED.encode(), data, ctypes.c_long(len(data)), n, task, out, ctypes.c_long(n * 64), err)
    return _finish(rc, out, err)


def fft(elements, omega, task=0):
    """elements: list of bytes (the List<byte[]>). Returns (result bytes, local refs deleted)."""
    m = mock()
    n = len(elements)
    data = b"".join(elements)
    offs, lens, o = [], [], 0
    for e in elements:
        offs.append(o)
        lens.append(len(e))
        o += len(e)
    out, err = ctypes.create_string_buffer(max(n, 1) * 64), ctypes.create_string_buffer(1024)
    refs = ctypes.c_int(0)
    m.mock_fft.restype = ctypes.c_long
    rc = m.mock_fft(SHIM_FFT.encode(), data, (ctypes.c_long * max(n, 1))(*offs), (ctypes.c_int * max(n, 1))(*lens), n,
                    omega, ctypes.c_long(len(omega)), task, out, ctypes.c_long(max(n, 1) * 64), err,
                    ctypes.byref(refs))
    return _finish(rc, out, err), refs.value


def qap_witness(a, b, c, m, omega, g, task=0):
    mk = mock()
    cap = 32 * (m + 1)
    out, err = ctypes.create_string_buffer(cap), ctypes.create_string_buffer(1024)
    mk.mock_qap_witness.restype = ctypes.c_long
    rc = mk.mock_qap_witness(SHIM_FFT.encode(), a, b, c, ctypes.c_long(len(a)), m, omega, ctypes.c_long(len(omega)),
                             g, ctypes.c_long(len(g)), task, out, ctypes.c_long(cap), err)
    return _finish(rc, out, err)


def prepared_msm(bases, scalars, n, type_, task=0, reps=2):
    m = mock()
    out, err = ctypes.create_string_buffer(384), ctypes.create_string_buffer(1024)
    m.mock_prepared_msm.restype = ctypes.c_long
    rc = m.mock_prepared_msm(SHIM_VAR.encode(), bases, ctypes.c_long(len(bases)), scalars, ctypes.c_long(len(scalars)),
                             n, type_, task, reps, out, ctypes.c_long(384), err)
    return _finish(rc, out, err)


def fixed_batch_compact(outerc, ws, n, base, scalars, bn, task=0):
    m = mock()
    cap = n * 192
    out, err = ctypes.create_string_buffer(cap), ctypes.create_string_buffer(1024)
    m.mock_fixed_batch_compact.restype = ctypes.c_long
    rc = m.mock_fixed_batch_compact(SHIM_FIXED.encode(), outerc, ws, n, base, ctypes.c_long(len(base)), scalars,
                                    ctypes.c_long(len(scalars)), bn, task, out, ctypes.c_long(cap), err)
    return _finish(rc, out, err)


def fft_flat(data, n, omega, task=0):
    m = mock()
    cap = max(n, 1) * 32
    out, err = ctypes.create_string_buffer(cap), ctypes.create_string_buffer(1024)
    m.mock_fft_flat.restype = ctypes.c_long
    rc = m.mock_fft_flat(SHIM_FFT.encode(), data, ctypes.c_long(len(data)), n, omega, ctypes.c_long(len(omega)), task,
                         out, ctypes.c_long(cap), err)
    return _finish(rc, out, err)
