"""Parity at BASELINE.json's FULL sizes for the entry points round 1 only covered at small sizes
(VERDICT r1 "weak" 1): G2 variable-base MSM at 2^20, the double MSM at 2^18, the whole 2^22 FFT against the
C oracle (SHA-256 of every output byte), fixed-base G1 (window 17) and G2 at 2^20."""
import ctypes
import hashlib

import numpy as np
import pytest

from oracle import bn254 as o
from oracle import coracle

pytestmark = pytest.mark.gpu


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _fixed_base_compact(n, type_, ks_bytes, window=16):
    """k_i * generator for the n 32-byte scalars `ks_bytes`, in the variable-base wire-in format, on the device."""
    import torch
    from octopuszk_amd import lib
    L = lib.load()
    outerc = (254 + window - 1) // window
    gen = o.g1_to_wire(o.G1.one) if type_ == 1 else o.g2_to_wire(o.G2.one)
    base = torch.from_numpy(np.frombuffer(gen, dtype=np.uint8).copy()).cuda()
    d_k = torch.from_numpy(ks_bytes.reshape(-1)).cuda()
    out = torch.empty(n * (96 if type_ == 1 else 192), dtype=torch.uint8, device="cuda")
    wsb = int(L.ozk_fixed_batch_msm_workspace_bytes(outerc, window, n, type_))
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    lib.check(L.ozk_fixed_batch_msm_compact_dev(outerc, window, n, _p(base), _p(d_k), type_, _p(out), _p(ws), wsb,
                                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return out


def _rand_scalars(n, seed, bits64=False):
    rng = np.random.default_rng(seed)
    s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    if bits64:
        s[:, 8:] = 0
    else:
        s[:, 31] &= 0x1F
    return s


def _dot_mod_r(a, b):
    """sum a_i b_i mod r for two (n, 32) little-endian byte arrays, in exact integers."""
    acc = 0
    for x, y in zip(a, b):
        acc += int.from_bytes(x.tobytes(), "little") * int.from_bytes(y.tobytes(), "little")
    return acc % o.R


def test_g2_var_msm_2p20_discrete_log_identity():
    """sum s_i (k_i G2) = (sum s_i k_i mod r) G2 at the size of the prover's B query."""
    import torch
    from octopuszk_amd import device as dev
    n = 1 << 20
    ks, sc = _rand_scalars(n, 201, bits64=True), _rand_scalars(n, 202)
    bases = _fixed_base_compact(n, 2, ks)
    ws = dev.VarMsmWorkspace(n, 2)
    out = ws.run(bases, torch.from_numpy(sc.reshape(-1)).cuda())
    torch.cuda.synchronize()
    want = o.G2.to_affine(o.G2.mul(o.G2.one, _dot_mod_r(sc, ks)))
    assert bytes(out.cpu().numpy()) == o.g2_out_le(want)


def test_double_msm_2p18_host_entry():
    """variableBaseDoubleMSMNativeHelper's C entry (host buffers, overlapped G1 || G2) at 2^18."""
    from octopuszk_amd import variable_base_msm as vb
    n = 1 << 18
    k1, k2, sc = _rand_scalars(n, 211, bits64=True), _rand_scalars(n, 212, bits64=True), _rand_scalars(n, 213)
    b1 = bytes(_fixed_base_compact(n, 1, k1).cpu().numpy())
    b2 = bytes(_fixed_base_compact(n, 2, k2).cpu().numpy())
    raw = vb.variable_base_double_msm_native_helper(b1, b2, sc.tobytes(), n, 0)
    assert raw[:192] == o.g1_out_le(o.G1.to_affine(o.G1.mul(o.G1.one, _dot_mod_r(sc, k1))))
    assert raw[192:] == o.g2_out_le(o.G2.to_affine(o.G2.mul(o.G2.one, _dot_mod_r(sc, k2))))


def test_fft_2p22_every_byte_vs_c_oracle():
    """BASELINE.json configs[2]: the whole 2^22 transform, compared through SHA-256 of all 256 MiB of output."""
    from octopuszk_amd import lib
    L = lib.load()
    n = 1 << 22
    a = _rand_scalars(n, 221)
    a[0] = 0
    a[1] = np.frombuffer((o.R - 1).to_bytes(32, "little"), dtype=np.uint8)
    w = o.to_le32(o.fr_root_of_unity(n))
    out = ctypes.create_string_buffer(64 * n)
    lib.check(L.ozk_fft_host(a.ctypes.data_as(ctypes.c_void_p), n, ctypes.cast(ctypes.c_char_p(w), ctypes.c_void_p), 0,
                             ctypes.cast(out, ctypes.c_void_p)))
    want = coracle.fft_fr(a.tobytes(), n, w)
    assert hashlib.sha256(out.raw).hexdigest() == hashlib.sha256(want).hexdigest()
    # the compact form carries the same values in 32-byte elements
    out32 = ctypes.create_string_buffer(32 * n)
    lib.check(L.ozk_fft_compact_host(a.ctypes.data_as(ctypes.c_void_p), n, ctypes.cast(ctypes.c_char_p(w), ctypes.c_void_p),
                                     0, ctypes.cast(out32, ctypes.c_void_p)))
    assert out32.raw == np.frombuffer(want, dtype=np.uint8).reshape(n, 64)[:, :32].tobytes()


@pytest.mark.parametrize("type_", [1, 2])
def test_fixed_base_2p20_window_17_sampled(type_):
    """batchMSMNativeHelper's C entry with the reference's own window (17 at 2^20 scalars,
    BN254aG1Parameters.java:25-50): sampled results against the oracle, plus a structural check of all of them
    (Z = 1, 64-byte big-endian coordinates) and the compact form's bytes."""
    from octopuszk_amd import fixed_base_msm as fb, lib
    L = lib.load()
    C = o.G1 if type_ == 1 else o.G2
    n, w = 1 << 20, 17
    outerc = (254 + w - 1) // w
    sc = _rand_scalars(n, 230 + type_)
    sc[0] = 0
    sc[1] = np.frombuffer((o.R - 1).to_bytes(32, "little"), dtype=np.uint8)
    base = C.mul(C.one, 0xDEADBEEFCAFE)     # Jacobian base, Z != 1
    bw = o.g1_to_wire(base) if type_ == 1 else o.g2_to_wire(base)
    raw = fb.batch_msm_native_helper(outerc, w, outerc, 1 << w, n, 254, bw, sc.tobytes(), type_, 0)
    per, k = (192, 3) if type_ == 1 else (384, 6)
    arr = np.frombuffer(raw, dtype=np.uint8).reshape(n, k, 64)
    assert not arr[:, :, :32].any()                       # upper half of every big-endian coordinate
    from_be = o.g1_from_out_be if type_ == 1 else o.g2_from_out_be
    for i in (0, 1, 2, 3, 77777, n // 2, n - 1):
        s = int.from_bytes(sc[i].tobytes(), "little")
        assert from_be(raw[per * i:per * (i + 1)]) == C.to_affine(o.fixed_base_mul(C, base, 254, w, s)), i
    compact = ctypes.create_string_buffer(n * per // 2)
    lib.check(L.ozk_fixed_batch_msm_compact_host(outerc, w, n, ctypes.cast(ctypes.c_char_p(bw), ctypes.c_void_p),
                                                 sc.ctypes.data_as(ctypes.c_void_p), type_, 0,
                                                 ctypes.cast(compact, ctypes.c_void_p)))
    # compact = the same values, little-endian, 32 bytes each
    assert compact.raw == np.ascontiguousarray(arr[:, :, :31:-1]).tobytes()
