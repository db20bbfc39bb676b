"""GPU parity of the three-stage schedule (sort | accumulate | tail on their own streams, device.VarMsmPipeline3)
and of the THROUGHPUT shape of the window sums its tails use (serial levels, csrc/msm_var_driver.cuh tail_shape):
same bytes as the single-call path and as the oracle, on uniform, skewed, tiny and multi-tile inputs."""
import random

import pytest

from oracle import bn254 as o

pytestmark = pytest.mark.gpu


def _scalars(n, seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x1F
    return sc.reshape(-1)


def _dlog_expected(sc_bytes, ks):
    acc = 0
    for i, k in enumerate(ks):
        acc += int.from_bytes(bytes(sc_bytes[32 * i:32 * i + 32]), "little") * k
    return o.g1_out_le(o.G1.to_affine(o.G1.mul(o.G1.one, acc % o.R)))


@pytest.mark.parametrize("n", [1, 5, 300, 1 << 12, (1 << 15) + 77, 1 << 18])
@pytest.mark.parametrize("prepared", [False, True])
def test_three_stage_pipeline_equals_single_calls(n, prepared):
    import torch
    from octopuszk_amd import device as dev
    bases = dev.gen_g1_bases(n, seed=40 + n % 7)
    inputs = [torch.from_numpy(_scalars(n, 50 + i)).cuda() for i in range(7)]
    ws = dev.VarMsmWorkspace(n, 1)
    serial = []
    for d_sc in inputs:
        out = ws.run(bases, d_sc)
        torch.cuda.synchronize()
        serial.append(bytes(out.cpu().numpy()))
    pipe = dev.VarMsmPipeline3(n, 1, depth=3, tail_streams=2)
    assert pipe.depth == 4          # rounded up so that a result slot keeps its tail stream
    b = pipe.prepare(bases) if prepared else bases
    got, pending = [], []
    for d_sc in inputs:
        pending.append(pipe.submit(b, d_sc, prepared=prepared))
        if len(pending) > 2:        # results are taken two submissions late, on the tail's own stream
            t = pending.pop(0)
            with torch.cuda.stream(pipe.stream_of(t)):
                r = pipe.result(t).clone()
            got.append(r)
    for t in pending:
        with torch.cuda.stream(pipe.stream_of(t)):
            got.append(pipe.result(t).clone())
    torch.cuda.synchronize()
    assert [bytes(g.cpu().numpy()) for g in got] == serial
    assert len(set(serial)) == (7 if n > 1 else len(set(serial)))


def test_three_stage_pipeline_vs_oracle_small():
    import numpy as np
    import torch
    from octopuszk_amd import device as dev
    rng = random.Random(77)
    n = 97
    G = o.G1
    bases = [G.to_affine(G.mul(G.one, rng.randrange(1, 1 << 64))) for _ in range(n)]
    bases[3] = G.zero
    bases[10] = G.negate(bases[11])
    d_bases = torch.from_numpy(np.frombuffer(b"".join(o.g1_to_wire(b) for b in bases), dtype=np.uint8).copy()).cuda()
    pipe = dev.VarMsmPipeline3(n, 1)
    tickets, wants = [], []
    for r in range(4):
        scalars = [rng.randrange(o.R) for _ in range(n)]
        scalars[0], scalars[1], scalars[2] = 0, 1, o.R - 1
        scalars[10] = scalars[11] = 4242
        sc = np.frombuffer(b"".join(s.to_bytes(32, "little") for s in scalars), dtype=np.uint8).copy()
        # (the last MSM of the burst announces itself: latency-shaped tail, same bytes)
        tickets.append((pipe.submit(d_bases, torch.from_numpy(sc).cuda(), last=(r == 3)), sc))
        wants.append(o.g1_out_le(G.to_affine(o.naive_msm(G, scalars, bases))))
    torch.cuda.synchronize()
    got = [bytes(pipe.outs[t % pipe.depth].cpu().numpy()) for t, _ in tickets]
    assert got == wants


@pytest.mark.parametrize("n", [7, 300, 1 << 12])
def test_three_stage_pipeline_g2_equals_single_calls(n):
    """the same schedule over G2 (LDS-resident accumulator, Fq2 group law) against the single-call path"""
    import numpy as np
    import torch
    from octopuszk_amd import device as dev
    rng = random.Random(300 + n)
    G = o.G2
    pts = [G.to_affine(G.mul(G.one, rng.randrange(1, 1 << 64))) for _ in range(min(n, 48))]
    bases = [pts[i % len(pts)] for i in range(n)]
    d_bases = torch.from_numpy(np.frombuffer(b"".join(o.g2_to_wire(b) for b in bases), dtype=np.uint8).copy()).cuda()
    inputs = [torch.from_numpy(_scalars(n, 70 + i)).cuda() for i in range(5)]
    ws = dev.VarMsmWorkspace(n, 2)
    serial = []
    for d_sc in inputs:
        out = ws.run(d_bases, d_sc)
        torch.cuda.synchronize()
        serial.append(bytes(out.cpu().numpy()))
    pipe = dev.VarMsmPipeline3(n, 2)
    ts = [pipe.submit(d_bases, d_sc) for d_sc in inputs[:4]]
    torch.cuda.synchronize()
    assert [bytes(pipe.outs[t % pipe.depth].cpu().numpy()) for t in ts] == serial[:4]
    want = o.g2_out_le(G.to_affine(o.naive_msm(G, [int.from_bytes(bytes(inputs[0].cpu().numpy()[32 * i:32 * i + 32]), "little")
                                                   for i in range(n)], bases)))
    assert serial[0] == want


@pytest.mark.parametrize("logn", [20, 21])
def test_three_stage_full_size_discrete_log_identity(logn):
    """2^20 (the bench workload) and 2^21 (bins of several register tiles in k_sort2): sum s_i (k_i G) =
    (sum s_i k_i) G with the right-hand side from exact integers."""
    import torch
    from octopuszk_amd import device as dev
    n = 1 << logn
    bases = dev.gen_g1_bases(n, seed=5)
    ks = dev.gen_base_logs(n, 5)
    pipe = dev.VarMsmPipeline3(n, 1)
    scs = [_scalars(n, 900 + i) for i in range(3)]
    ts = [pipe.submit(bases, torch.from_numpy(sc).cuda()) for sc in scs]
    torch.cuda.synchronize()
    for t, sc in zip(ts, scs):
        assert bytes(pipe.outs[t % pipe.depth].cpu().numpy()) == _dlog_expected(sc, ks)


def test_three_stage_profiler_shaped_full_size():
    """One repeated base, 64-bit / r - 64-bit scalars (VariableBaseMSMProfiling.java:19-31): giant sort bins and
    buckets cut into thousands of pieces, through the staged entry points and the throughput tail."""
    import numpy as np
    import torch
    from octopuszk_amd import device as dev
    n = 1 << 20
    rng = np.random.default_rng(10)
    lows = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
    neg = rng.integers(0, 2, size=n).astype(bool)
    vals = [(o.R - int(v)) if ng else int(v) for v, ng in zip(lows, neg)]
    sc = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint8).copy()
    base = o.G1.to_affine(o.G1.mul(o.G1.one, 987654321))
    bases = np.frombuffer(o.g1_to_wire(base) * n, dtype=np.uint8).copy()
    d_bases, d_scalars = torch.from_numpy(bases).cuda(), torch.from_numpy(sc).cuda()
    pipe = dev.VarMsmPipeline3(n, 1)
    ts = [pipe.submit(d_bases, d_scalars) for _ in range(3)]
    torch.cuda.synchronize()
    want = o.g1_out_le(o.G1.to_affine(o.G1.mul(base, sum(vals) % o.R)))
    for t in ts:
        assert bytes(pipe.outs[t % pipe.depth].cpu().numpy()) == want


@pytest.mark.parametrize("type_", [1, 2])
@pytest.mark.parametrize("mode", ["0", "1"])
def test_tail_shapes_give_the_same_bytes(type_, mode, monkeypatch):
    """OZK_MSM_TAIL_MODE forces the latency (0) / throughput (1) shape of the window sums on every entry point:
    the single-call path must return the oracle's bytes under both."""
    from octopuszk_amd import lib, variable_base_msm as vb
    G = o.G1 if type_ == 1 else o.G2
    rng = random.Random(60 + type_)
    n = 1500 if type_ == 1 else 200
    pts = [G.to_affine(G.mul(G.one, rng.randrange(1, 1 << 64))) for _ in range(24)]
    bases = (pts * (n // 24 + 1))[:n]
    scalars = [rng.randrange(o.R) for _ in range(n)]
    want = G.to_affine(o.pippenger_msm(G, scalars, bases))
    L = lib.load()
    monkeypatch.setenv("OZK_MSM_TAIL_MODE", mode)
    L.ozk_tuning_reload()
    try:
        if type_ == 1:
            raw = vb.variable_base_serial_msm_native_helper(vb.marshal_g1(bases), vb.marshal_scalars(scalars), n, 1, 0)
            assert raw == o.g1_out_le(want)
        else:
            raw = vb.variable_base_serial_msm_native_helper(vb.marshal_g2(bases), vb.marshal_scalars(scalars), n, 2, 0)
            assert raw == o.g2_out_le(want)
    finally:
        monkeypatch.delenv("OZK_MSM_TAIL_MODE")
        L.ozk_tuning_reload()


def test_device_clock_timing_agrees_with_hip_events():
    """ozk_prof_enable(2) (the kernel's waves stamp the device clock) against ozk_prof_enable(1) (HIP events on the
    dispatch) on a lone MSM stream, where neither perturbs anything: same kernel, same duration within 8 % (the
    two runs are not the same launches; 0 - 6.5 % apart on the boxes of the pool)."""
    import ctypes
    import torch
    from octopuszk_amd import device as dev, lib
    L = lib.load()
    n = 1 << 18
    bases = dev.gen_g1_bases(n, seed=8)
    d_sc = torch.from_numpy(_scalars(n, 3)).cuda()
    ws = dev.VarMsmWorkspace(n, 1)
    means, clocks = [], []
    try:
        for mode in (2, 1, 2, 1):
            ws.run(bases, d_sc)
            torch.cuda.synchronize()
            lib.check(L.ozk_prof_enable(mode))
            for _ in range(8):
                ws.run(bases, d_sc)
            torch.cuda.synchronize()
            st, k = (ctypes.c_double * 4)(), ctypes.c_int()
            lib.check(L.ozk_prof_dominant_kernel_stats(st, ctypes.byref(k)))
            if mode == 2:   # the shader clock the kernel stamped itself (round 4: what bench.py reports instead of sysfs)
                c4, ck = (ctypes.c_double * 4)(), ctypes.c_int()
                lib.check(L.ozk_prof_dominant_kernel_clock_mhz(c4, ctypes.byref(ck)))
                assert ck.value == 8
                clocks.append(list(c4))
            assert k.value == 8 and st[2] > 0
            means.append(st[1])
    finally:
        lib.check(L.ozk_prof_enable(0))   # (a failure in here must not leave profiling enabled for the tests that follow)
    for c4 in clocks:   # mean, median, min, max: an MI355X runs its shaders between 1 and 2.5 GHz under load
        assert 1000.0 < c4[2] <= c4[1] <= c4[3] < 2600.0, c4
    clock, events = (means[0] + means[2]) / 2, (means[1] + means[3]) / 2
    assert abs(clock - events) / events < 0.08, (clock, events)


@pytest.mark.parametrize("split", [False, True])
def test_cu_partitioned_and_split_schedules_give_the_same_bytes(split):
    """VarMsmPipeline3(tail_cus=32): tail streams on 32 compute units, the accumulate stream on the others
    (ozk_stream_create_cu_range); split_accum: level 1 and the rest of the accumulate stage on different streams
    (ozk_var_msm_accum_part_dev).  Same bytes as the single-call path; the partitioned form refuses the null stream."""
    import torch
    from octopuszk_amd import device as dev, lib
    L = lib.load()
    assert L.ozk_device_cu_count() >= 64
    n = (1 << 15) + 77
    bases = dev.gen_g1_bases(n, seed=44)
    inputs = [torch.from_numpy(_scalars(n, 150 + i)).cuda() for i in range(7)]
    ws = dev.VarMsmWorkspace(n, 1)
    serial = []
    for d_sc in inputs:
        out = ws.run(bases, d_sc)
        torch.cuda.synchronize()
        serial.append(bytes(out.cpu().numpy()))
    pipe = dev.VarMsmPipeline3(n, 1, depth=4, tail_streams=2, tail_cus=32, split_accum=split)
    with pytest.raises(RuntimeError, match="null stream"):
        pipe.submit(bases, inputs[0])
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ts = [pipe.submit(bases, d_sc, last=(i == 6)) for i, d_sc in enumerate(inputs[:4])]
        torch.cuda.synchronize()
        got = [bytes(pipe.outs[t % pipe.depth].cpu().numpy()) for t in ts]
        ts = [pipe.submit(bases, d_sc) for d_sc in inputs[4:]]
        torch.cuda.synchronize()
        got += [bytes(pipe.outs[t % pipe.depth].cpu().numpy()) for t in ts]
    assert got == serial
    pipe.close()
    with pytest.raises(ValueError):
        dev.VarMsmPipeline3(n, 1, tail_cus=100000)


def test_split_accumulate_stage_without_partition():
    import torch
    from octopuszk_amd import device as dev
    n = 5000
    bases = dev.gen_g1_bases(n, seed=45)
    inputs = [torch.from_numpy(_scalars(n, 170 + i)).cuda() for i in range(6)]
    ws = dev.VarMsmWorkspace(n, 1)
    serial = []
    for d_sc in inputs:
        out = ws.run(bases, d_sc)
        torch.cuda.synchronize()
        serial.append(bytes(out.cpu().numpy()))
    pipe = dev.VarMsmPipeline3(n, 1, depth=4, split_accum=True)
    b = pipe.prepare(bases)
    got = []
    for half in (inputs[:3], inputs[3:]):
        ts = [pipe.submit(b, d_sc, prepared=True) for d_sc in half]
        torch.cuda.synchronize()
        got += [bytes(pipe.outs[t % pipe.depth].cpu().numpy()) for t in ts]
    assert got == serial


def test_split_accumulate_stage_g2():
    """the two-part accumulate stage over G2 (level 1 with its accumulators in LDS) against the single-call path"""
    import numpy as np
    import torch
    from octopuszk_amd import device as dev
    rng = random.Random(9)
    n, G = 700, o.G2
    pts = [G.to_affine(G.mul(G.one, rng.randrange(1, 1 << 64))) for _ in range(32)]
    d_bases = torch.from_numpy(np.frombuffer(b"".join(o.g2_to_wire(pts[i % 32]) for i in range(n)), dtype=np.uint8).copy()).cuda()
    inputs = [torch.from_numpy(_scalars(n, 190 + i)).cuda() for i in range(5)]
    ws = dev.VarMsmWorkspace(n, 2)
    serial = []
    for d_sc in inputs:
        out = ws.run(d_bases, d_sc)
        torch.cuda.synchronize()
        serial.append(bytes(out.cpu().numpy()))
    pipe = dev.VarMsmPipeline3(n, 2, depth=4, split_accum=True)
    got = []
    for d_sc in inputs:
        t = pipe.submit(d_bases, d_sc)
        got.append(bytes(pipe.result(t).cpu().numpy()))
    assert got == serial
