"""MI355X-native BN254 MSM / FFT back end behind the OctopusZK (DIZK) JNI surface.

Layout: csrc/ (HIP kernels + C ABI + JNI shims), lib.py (ctypes binding),
variable_base_msm.py / fixed_base_msm.py / fft.py (host-side mirrors of the reference's
algebra.msm.VariableBaseMSM, algebra.msm.FixedBaseMSM and algebra.fft.SerialFFT for this
path: same names, argument meaning and byte formats)."""
from . import lib  # noqa: F401

__all__ = ["lib"]
