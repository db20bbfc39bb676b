"""The chunking variants of the reference's Java callers, mirrored in octopuszk_amd/variable_base_msm.py and
fixed_base_msm.py (VERDICT r2 "missing" 1): serialMSMPartition / doubleMSM / doubleMSMPartition
(VariableBaseMSM.java:341-469, 484-612, 614-770), batchMSM's chunk loop / batchMSMPartition / doubleBatchMSM with its
9 x 64-byte records / batchFieldMSMPartition with the base appended as element n / batchFilterFieldMSMPartition
(FixedBaseMSM.java:186-443, 489-602, 753-852).  Chunk sizes are lowered so that a few hundred elements make several
native calls; the chunk results must add up to the oracle's single answer."""
import random

import pytest

from oracle import bn254 as o

pytestmark = pytest.mark.gpu


def _pts(C, n, rng, affine=True):
    out = []
    for _ in range(n):
        P = C.mul(C.one, rng.randrange(1, 1 << 64))
        out.append(C.to_affine(P) if affine else P)
    return out


@pytest.mark.parametrize("is_g1", [True, False])
def test_serial_msm_partition_chunks_and_task_id(is_g1, monkeypatch):
    from octopuszk_amd import variable_base_msm as vb
    C = o.G1 if is_g1 else o.G2
    rng = random.Random(21 + is_g1)
    n = 150 if is_g1 else 70
    bases = _pts(C, n, rng, affine=False)          # un-normalised Jacobian inputs, as reference-produced keys
    sc = [rng.randrange(o.R) for _ in range(n)]
    monkeypatch.setattr(vb, "G1_PARTITION_ITERATION_BATCH", 64)
    monkeypatch.setattr(vb, "G2_PARTITION_ITERATION_BATCH", 32)
    got = vb.serial_msm_partition(list(zip(sc, bases)), C.add, C.zero, is_g1=is_g1, task_id=7)   # 7 % 1 device = 0
    assert C.equals(got, o.naive_msm(C, sc, bases))


def test_double_msm_and_partition_chunks(monkeypatch):
    from octopuszk_amd import variable_base_msm as vb
    rng = random.Random(31)
    n = 90
    b1, b2 = _pts(o.G1, n, rng), _pts(o.G2, n, rng)
    sc = [rng.randrange(o.R) for _ in range(n)]
    sc[0], sc[1] = 0, o.R - 1
    monkeypatch.setattr(vb, "DOUBLE_ITERATION_BATCH", 32)      # 3 native calls
    w1, w2 = o.naive_msm(o.G1, sc, b1), o.naive_msm(o.G2, sc, b2)
    g1, g2 = vb.double_msm(sc, list(zip(b1, b2)), o.G1.add, o.G1.zero, o.G2.add, o.G2.zero)
    assert o.G1.equals(g1, w1) and o.G2.equals(g2, w2)
    g1, g2 = vb.double_msm_partition(list(zip(sc, zip(b1, b2))), o.G1.add, o.G1.zero, o.G2.add, o.G2.zero, task_id=3)
    assert o.G1.equals(g1, w1) and o.G2.equals(g2, w2)
    # one chunk == the raw native on the whole input
    monkeypatch.setattr(vb, "DOUBLE_ITERATION_BATCH", 1 << 21)
    h1, h2 = vb.double_msm(sc, list(zip(b1, b2)), o.G1.add, o.G1.zero, o.G2.add, o.G2.zero)
    assert o.G1.equals(h1, w1) and o.G2.equals(h2, w2)


@pytest.mark.parametrize("is_g1", [True, False])
def test_batch_msm_chunk_loop_and_partition(is_g1, monkeypatch):
    from octopuszk_amd import fixed_base_msm as fb
    C = o.G1 if is_g1 else o.G2
    rng = random.Random(41 + is_g1)
    n, window = (100, 5) if is_g1 else (40, 4)
    base = C.mul(C.one, rng.randrange(1, o.R))
    scalars = [rng.randrange(o.R) for _ in range(n)]
    monkeypatch.setattr(fb, "G1_ITERATION_BATCH", 32)
    monkeypatch.setattr(fb, "G2_ITERATION_BATCH", 16)
    want = [C.to_affine(C.mul(base, s)) for s in scalars]
    assert fb.batch_msm(254, window, base, scalars, is_g1=is_g1) == want
    idx = [(1000 + 3 * i, s) for i, s in enumerate(scalars)]
    num_windows = (254 + window - 1) // window
    got = fb.batch_msm_partition(254, window, num_windows, 1 << window, base, idx, is_g1=is_g1, task_id=5)
    assert [g[0] for g in got] == [p[0] for p in idx] and [g[1] for g in got] == want


def test_double_batch_msm_nine_value_records(monkeypatch):
    from octopuszk_amd import fixed_base_msm as fb
    rng = random.Random(51)
    n = 70
    base1 = o.G1.mul(o.G1.one, rng.randrange(1, o.R))
    base2 = o.G2.mul(o.G2.one, rng.randrange(1, o.R))
    scalars = [rng.randrange(o.R) for _ in range(n)]
    scalars[0], scalars[1], scalars[2] = 0, 1, o.R - 1
    monkeypatch.setattr(fb, "DOUBLE_ITERATION_BATCH", 32)
    w1, w2 = 6, 5
    nw1, nw2 = (254 + w1 - 1) // w1, (254 + w2 - 1) // w2
    got = fb.double_batch_msm(nw1, 1 << w1, nw2, 1 << w2, 254, w1, base1, 254, w2, base2, scalars)
    assert len(got) == n
    for s, (p1, p2) in zip(scalars, got):
        assert p1 == o.G1.to_affine(o.G1.mul(base1, s))
        assert p2 == o.G2.to_affine(o.G2.mul(base2, s))


def test_batch_field_msm_partition_and_filter():
    from octopuszk_amd import fixed_base_msm as fb
    rng = random.Random(61)
    n, num_inputs = 300, 37
    b = rng.randrange(1, o.R)
    xs = [(i, rng.randrange(o.R)) for i in range(n)]
    xs[0] = (0, 0)
    xs[1] = (1, o.R - 1)
    rng.shuffle(xs)                                  # a partition's elements come in no particular order
    assert fb.batch_field_msm_partition(b, xs, task_id=2) == [(i, x * b % o.R) for i, x in xs]
    inv_d, inv_g = rng.randrange(1, o.R), rng.randrange(1, o.R)
    gam = fb.batch_filter_field_msm_partition(inv_d, inv_g, num_inputs, xs, 0, task_id=1)
    dlt = fb.batch_filter_field_msm_partition(inv_d, inv_g, num_inputs, xs, 1, task_id=1)
    assert gam == [(i, x * inv_g % o.R) for i, x in xs if i < num_inputs]
    assert dlt == [(i, x * inv_d % o.R) for i, x in xs if i >= num_inputs]
    assert len(gam) + len(dlt) == n
