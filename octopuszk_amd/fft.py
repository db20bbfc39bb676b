"""Host-side mirror of the reference's algebra.fft.SerialFFT / FFTAuxiliary JNI path
(FFTAuxiliary.java:41-51 marshal, :60-124 serialRadix2FFT, :224-232 multiplyByCoset;
SerialFFT.java:24-28 omega, :75-115 wrappers).  The native is the C ABI of libozk_hip.so;
the coset / inverse wrappers do the same element-wise host arithmetic the Java wrappers do."""
import ctypes

from . import lib as _lib

FR = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # BN254aFrParameters.java:33
FR_ROOT = 19103219067921713944291392827692070036145651957329286315305642004821462161904  # :34
FR_MULT_GEN = 5  # :35


def big_integer_to_byte_array_cgbn(v: int) -> bytes:
    """FFTAuxiliary.java:41-51: toByteArray() reversed, padded to a multiple of 4 bytes."""
    v = int(v)
    nbytes = v.bit_length() // 8 + 1
    return v.to_bytes((nbytes + 3) // 4 * 4, "little")


def root_of_unity(order: int) -> int:
    """Fp.rootOfUnity (Fp.java:98-102)."""
    return pow(FR_ROOT, FR // order, FR)


def serial_radix2_fft_native_helper(inputs, omega: bytes, task_id: int) -> bytes:
    """JNI native of FFTAuxiliary.java:53-55: `inputs` is the List<byte[]> (each LE, length a
    multiple of 4, <= 32).  The JNI shim flattens the list into n x 32 B exactly like this."""
    n = len(inputs)
    flat = b"".join(bytes(b).ljust(32, b"\x00") for b in inputs)
    L = _lib.load()
    out = ctypes.create_string_buffer(64 * n)
    _lib.check(L.ozk_fft_host(ctypes.cast(ctypes.c_char_p(flat), ctypes.c_void_p), n,
                              ctypes.cast(ctypes.c_char_p(bytes(omega).ljust(32, b"\x00")), ctypes.c_void_p),
                              task_id, ctypes.cast(out, ctypes.c_void_p)))
    return out.raw


def serial_radix2_fft(values, omega: int, task_id: int = 0):
    """FFTAuxiliary.serialRadix2FFT (FFTAuxiliary.java:60-124) through the native path
    (the commented call site at :72-97): returns the transformed list of ints."""
    n = len(values)
    if n == 1:
        return list(values)
    raw = serial_radix2_fft_native_helper([big_integer_to_byte_array_cgbn(v) for v in values],
                                          big_integer_to_byte_array_cgbn(omega), task_id)
    return [int.from_bytes(raw[64 * i:64 * (i + 1)], "little") for i in range(n)]


class SerialFFT:
    """SerialFFT.java:17-115 for BN254 Fr."""

    def __init__(self, domain_size: int):
        assert domain_size & (domain_size - 1) == 0
        self.domain_size = domain_size
        self.omega = root_of_unity(domain_size)  # SerialFFT.java:24-28

    def radix2_fft(self, a, task_id=0):  # :75-78
        return serial_radix2_fft(a, self.omega, task_id)

    def radix2_inverse_fft(self, a, task_id=0):  # :86-95
        out = serial_radix2_fft(a, pow(self.omega, -1, FR), task_id)
        c = pow(self.domain_size, -1, FR)
        return [x * c % FR for x in out]

    def radix2_coset_fft(self, a, g, task_id=0):  # :100-105 + FFTAuxiliary.java:224-232
        coset, b = g, list(a)
        for i in range(1, len(b)):
            b[i] = b[i] * coset % FR
            coset = coset * g % FR
        return self.radix2_fft(b, task_id)

    def radix2_coset_inverse_fft(self, a, g, task_id=0):  # :111-115
        b = self.radix2_inverse_fft(a, task_id)
        gi = pow(g, -1, FR)
        coset = gi
        for i in range(1, len(b)):
            b[i] = b[i] * coset % FR
            coset = coset * gi % FR
        return b
