"""GPU parity tests of the radix-2 FFT over Fr through the C ABI (bit-exact)."""
import random

import pytest

from oracle import bn254 as o
from oracle import coracle
import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", gu.load("fft_fr.json")["cases"], ids=lambda c: "n%d%s" % (c["n"], "_kat" if "kat" in c else ""))
def test_fft_golden(case):
    from octopuszk_amd import lib
    import ctypes
    L = lib.load()
    data = gu.fft_case_wire(case)
    n = case["n"]
    out = ctypes.create_string_buffer(64 * n)
    lib.check(L.ozk_fft_host(ctypes.cast(ctypes.c_char_p(data), ctypes.c_void_p), n,
                             ctypes.cast(ctypes.c_char_p(bytes.fromhex(case["omega"])), ctypes.c_void_p), 0,
                             ctypes.cast(out, ctypes.c_void_p)))
    gu.fft_check(case, out.raw)


@pytest.mark.parametrize("logn", [1, 2, 5, 9, 10, 11, 13, 16, 17])
def test_fft_vs_c_oracle(logn):
    from octopuszk_amd import fft
    n = 1 << logn
    rng = random.Random(logn)
    a = [rng.randrange(o.R) for _ in range(n)]
    a[0], a[1] = 0, o.R - 1
    w = o.fr_root_of_unity(n)
    got = fft.serial_radix2_fft(a, w)
    want_raw = coracle.fft_fr(b"".join(o.to_le32(x) for x in a), n, o.to_le32(w))
    want = [int.from_bytes(want_raw[64 * i:64 * (i + 1)], "little") for i in range(n)]
    assert got == want


def test_fft_wrappers_roundtrip_and_kat():
    from octopuszk_amd import fft
    f = fft.SerialFFT(4)
    assert f.radix2_fft([2, 5, 3, 8]) == o.naive_dft([2, 5, 3, 8], o.fr_root_of_unity(4))  # SerialFFTTest.java:168-190
    rng = random.Random(1)
    n = 4096
    a = [rng.randrange(o.R) for _ in range(n)]
    f = fft.SerialFFT(n)
    assert f.radix2_inverse_fft(f.radix2_fft(a)) == a
    assert f.radix2_coset_inverse_fft(f.radix2_coset_fft(a, fft.FR_MULT_GEN), fft.FR_MULT_GEN) == a
    b = list(a)
    o.radix2_coset_fft(b, o.FR_MULT_GEN)
    assert f.radix2_coset_fft(a, fft.FR_MULT_GEN) == b


def test_fft_large_properties():
    """BASELINE config 3 size (2^22): size-independent checks — linearity and FFT(delta_k)[i] = omega^(ik),
    plus a spot check of a few outputs against the direct sum."""
    import ctypes
    import numpy as np
    from octopuszk_amd import lib
    L = lib.load()
    logn = 22
    n = 1 << logn
    w = o.fr_root_of_unity(n)

    def run(buf):
        out = ctypes.create_string_buffer(64 * n)
        lib.check(L.ozk_fft_host(buf.ctypes.data_as(ctypes.c_void_p), n,
                                 ctypes.cast(ctypes.c_char_p(o.to_le32(w)), ctypes.c_void_p), 0,
                                 ctypes.cast(out, ctypes.c_void_p)))
        return np.frombuffer(out.raw, dtype=np.uint8).reshape(n, 64)

    def val(row):
        return int.from_bytes(row.tobytes(), "little")

    delta = np.zeros((n, 32), dtype=np.uint8)
    k = 123457
    delta[k, 0] = 1
    fd = run(delta)
    for i in (0, 1, 2, 77777, n - 1):
        assert val(fd[i]) == pow(w, i * k, o.R)
    rng = np.random.default_rng(4)
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    a[:, 31] &= 0x1F
    fa = run(a)
    # a + delta_k (no carry: bump a different byte position safely by constructing in ints for row k)
    a2 = a.copy()
    vk = (int.from_bytes(a[k].tobytes(), "little") + 1) % o.R
    a2[k] = np.frombuffer(vk.to_bytes(32, "little"), dtype=np.uint8)
    fa2 = run(a2)
    for i in (0, 5, 4242, n // 2, n - 3):
        assert val(fa2[i]) == (val(fa[i]) + pow(w, i * k, o.R)) % o.R
    # out[0] = sum of inputs
    s = 0
    for row in a:
        s += int.from_bytes(row.tobytes(), "little")
    assert val(fa[0]) == s % o.R


@pytest.mark.parametrize("knobs", [{}, {"OZK_FFT_PLAN": "0"}, {"OZK_FFT_MAXK": "6"}, {"OZK_FFT_MAXK": "10"},
                                   {"OZK_FFT_KS": "6,8,6"}, {"OZK_FFT_KS": "5,5,5,5"}, {"OZK_FFT_KS": "9,4,7"}],
                         ids=lambda k: "_".join("%s%s" % (a[8:], b) for a, b in k.items()) or "default")
def test_fft_every_pass_plan_gives_the_oracle_bytes(knobs, monkeypatch):
    """The stages are spread over passes by a plan (default: every pass after the first even — 2^20 = 8 + 6 + 6; the
    even split of rounds 1-2 with OZK_FFT_PLAN=0 runs the odd-stage kernels of the later passes; OZK_FFT_MAXK caps a
    pass; OZK_FFT_KS gives the split by hand — used for 2^20 only, ignored where it does not sum to log2 n).  Every
    plan, every kernel specialisation (first / later pass, odd / even, workspace / result output) must give the C
    oracle's bytes; the sizes cover the default plans [6,4] [8,4] [8,6] [7,8] [7,6,4] [8,6,4] [7,6,6] [8,6,6]."""
    import ctypes
    import numpy as np
    from octopuszk_amd import lib
    L = lib.load()
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    L.ozk_tuning_reload()
    try:
        for logn in (10, 12, 14, 15, 17, 18, 19, 20):
            n = 1 << logn
            a = np.random.default_rng(logn).integers(0, 256, size=(n, 32), dtype=np.uint8)
            a[:, 31] &= 0x1F
            w = o.to_le32(o.fr_root_of_unity(n))
            out = ctypes.create_string_buffer(64 * n)
            lib.check(L.ozk_fft_host(a.ctypes.data_as(ctypes.c_void_p), n, ctypes.cast(ctypes.c_char_p(w), ctypes.c_void_p), 0,
                                     ctypes.cast(out, ctypes.c_void_p)))
            assert out.raw == coracle.fft_fr(a.tobytes(), n, w), (knobs, logn)
    finally:
        for k in knobs:
            monkeypatch.delenv(k)
        L.ozk_tuning_reload()


def test_fft_plan_cache_many_domains_from_many_threads():
    """the plan cache handed out raw pointers into four process-wide slots.  Nine domains (more than the
    four plans a device keeps) cycled from six host threads at once — every call must return what the same call
    returns alone; the plans in use are pinned while their caller enqueues its kernels (csrc/fft.hip PlanPin)."""
    import threading
    from octopuszk_amd import fft as F, lib
    rng = random.Random(99)
    domains = []
    for logn in (3, 5, 6, 8, 9, 10, 11, 12, 13):
        n = 1 << logn
        vals = [rng.randrange(F.FR) for _ in range(n)]
        domains.append((vals, F.root_of_unity(n)))
    alone = [F.serial_radix2_fft(v, w) for v, w in domains]
    assert alone[0] == o.naive_dft(domains[0][0], domains[0][1]) and alone[3] == o.naive_dft(domains[3][0], domains[3][1])
    errors = []

    def worker(t):
        try:
            r = random.Random(t)
            for it in range(40):
                k = r.randrange(len(domains))
                if F.serial_radix2_fft(domains[k][0], domains[k][1], task_id=t) != alone[k]:
                    errors.append((t, it, k))
        except Exception as e:   # noqa: BLE001
            errors.append((t, repr(e)))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors[:5]
    lib.load().ozk_host_cache_release()          # releases the unpinned plans and stops the copy helpers ...
    assert F.serial_radix2_fft(domains[4][0], domains[4][1]) == alone[4]   # ... and the next call starts over
