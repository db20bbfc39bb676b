"""MI355X-native BN254 MSM / FFT back end behind the OctopusZK (DIZK) JNI surface.

Layout: csrc/ (HIP kernels + C ABI + JNI shims), lib.py (ctypes binding),
variable_base_msm.py / fixed_base_msm.py / fft.py / r1cs_to_qap.py (host-side mirrors of the
reference's algebra.msm.VariableBaseMSM, algebra.msm.FixedBaseMSM, algebra.fft.SerialFFT and the
witness half of reductions.r1cs_to_qap.R1CStoQAP for this path: same names, argument meaning and
byte formats), device.py (device-resident entry points on torch tensors: workspaces, the two-stream
pipeline, prepared bases), distributed.py (multi-GPU composition), build.py."""
from . import lib  # noqa: F401

__all__ = ["lib"]
