"""world_size-2 gloo test of the multi-GPU composition (index-range sharding + all-gather of
partials + sum): the per-rank HIP kernels are replaced by the oracle so the N > 1 control and
communication path runs on CPU."""
import os
import random
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import bn254 as o
from oracle import coracle


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, bases_wire, scalars_wire, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from octopuszk_amd import distributed as D
    lo, hi = D.shard_range(n, rank, world)

    def local_partial():
        raw = coracle.pippenger_g1(bases_wire[96 * lo:96 * hi], scalars_wire[32 * lo:32 * hi], hi - lo)
        return torch.frombuffer(bytearray(raw), dtype=torch.uint8)

    def sum_fn(gathered, k, type_):
        acc = o.G1.zero
        g = bytes(gathered.numpy())
        for i in range(k):
            acc = o.G1.add(acc, o.g1_from_out_le(g[192 * i:192 * (i + 1)]))
        return torch.frombuffer(bytearray(o.g1_out_le(o.G1.to_affine(acc))), dtype=torch.uint8)

    out = D.distributed_var_msm(local_partial, sum_fn, 1)
    q.put((rank, bytes(out.numpy())))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_cover_everything():
    from octopuszk_amd import distributed as D
    for n in (1, 7, 8, 1 << 20, (1 << 24) + 3):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_sharded_msm_matches_single():
    rng = random.Random(5)
    n = 101
    bases = [o.G1.to_affine(o.G1.mul(o.G1.one, rng.randrange(1, 1 << 40))) for _ in range(n)]
    scalars = [rng.randrange(o.R) for _ in range(n)]
    bw = b"".join(o.g1_to_wire(P) for P in bases)
    sw = b"".join(o.to_le32(s) for s in scalars)
    want = coracle.pippenger_g1(bw, sw, n)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, bw, sw, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == want and got[1] == want


def _worker_double(rank, world, port, n, pts1, pts2, scalars, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from octopuszk_amd import distributed as D
    lo, hi = D.shard_range(n, rank, world)
    sc, p1, p2 = scalars[lo:hi], pts1[lo:hi], pts2[lo:hi]

    def local_partial():   # what ozk_var_double_msm_host returns for the slice: G1 (192) || G2 (384)
        raw = (o.g1_out_le(o.G1.to_affine(o.pippenger_msm(o.G1, sc, p1))) +
               o.g2_out_le(o.G2.to_affine(o.pippenger_msm(o.G2, sc, p2))))
        return torch.frombuffer(bytearray(raw), dtype=torch.uint8)

    def sum_fn(gathered, k, type_):
        G, size = (o.G1, 192) if type_ == 1 else (o.G2, 384)
        dec = o.g1_from_out_le if type_ == 1 else o.g2_from_out_le
        enc = o.g1_out_le if type_ == 1 else o.g2_out_le
        g = bytes(gathered.numpy())
        acc = G.zero
        for i in range(k):
            acc = G.add(acc, dec(g[size * i:size * (i + 1)]))
        return torch.frombuffer(bytearray(enc(G.to_affine(acc))), dtype=torch.uint8)

    out = D.distributed_var_double_msm(local_partial, sum_fn)
    q.put((rank, bytes(out.numpy())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_double_msm_over_gloo_matches_single(world):
    """VariableBaseMSM.distributedDoubleMSM (VariableBaseMSM.java:805-818) as octopuszk_amd.distributed composes it:
    per-rank double MSM over an index-range slice, ONE all-gather of the 576-byte G1 || G2 partials, two point sums."""
    rng = random.Random(11)
    n = 23
    p1 = [o.G1.to_affine(o.G1.mul(o.G1.one, rng.randrange(1, 1 << 40))) for _ in range(n)]
    p2 = [o.G2.to_affine(o.G2.mul(o.G2.one, rng.randrange(1, 1 << 40))) for _ in range(n)]
    p1[2], p2[5] = o.G1.zero, o.G2.zero
    scalars = [rng.randrange(o.R) for _ in range(n)]
    want = (o.g1_out_le(o.G1.to_affine(o.naive_msm(o.G1, scalars, p1))) +
            o.g2_out_le(o.G2.to_affine(o.naive_msm(o.G2, scalars, p2))))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_double, args=(r, world, port, n, p1, p2, scalars, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(got[r] == want for r in range(world))
