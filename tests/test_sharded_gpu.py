"""V10 on hardware (VERDICT r1 "missing" 2): the index-range-sharded variable-base MSM with the REAL per-rank HIP
pipeline — two ranks sharing cuda:0, exchange over gloo (RCCL refuses two ranks on one device) — against the C
oracle on the WHOLE input.  VariableBaseMSM.distributedMSM semantics (VariableBaseMSM.java:775-786):
mapPartitions(serialMSMPartition) -> reduce(GroupT::add)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import bn254 as o
from oracle import coracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, type_, bases_wire, scalars_wire, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from octopuszk_amd import device as dev
    from octopuszk_amd import distributed as D
    pt = 96 if type_ == 1 else 192
    lo, hi = D.shard_range(n, rank, world)
    d_b = torch.from_numpy(np.frombuffer(bases_wire[pt * lo:pt * hi], dtype=np.uint8).copy()).cuda()
    d_s = torch.from_numpy(np.frombuffer(scalars_wire[32 * lo:32 * hi], dtype=np.uint8).copy()).cuda()
    ws = dev.VarMsmWorkspace(hi - lo, type_)

    def gather_cpu(partial, group=None):   # gloo moves host tensors
        out = [torch.empty(partial.numel(), dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(out, partial.cpu().contiguous())
        return torch.cat(out).cuda()

    D.all_gather_partials = gather_cpu
    out = D.gpu_var_msm(ws, d_b, d_s)         # HIP MSM on the slice + all-gather + HIP point sum
    torch.cuda.synchronize()
    q.put((rank, bytes(out.cpu().numpy())))
    dist.barrier()
    dist.destroy_process_group()


def _persist_failure(tag, **kw):
    """Everything a failure of a concurrent host call needs, written before the assertion fires (the one
    failure of round 3 left only its test's name): gpurun_out/sharded_failure_<tag>.json"""
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    rec = {k: (v.hex() if isinstance(v, (bytes, bytearray)) else v) for k, v in kw.items()}
    with open(os.path.join(d, "sharded_failure_%s.json" % tag), "w") as f:
        json.dump(rec, f, indent=1)


def _sharded_case(type_, n, shards, tag):
    """ozk_var_msm_sharded_host in BOTH exchange forms (RCCL all-gather of the partials: ncclCommInitAll over the one
    device of this box; and through the host, OZK_SHARD_RCCL=0) against ozk_var_msm_host and the oracle.  Returns
    nothing; on any difference or error return the codes, messages, all byte strings and the library's per-call wait
    accounting are persisted first."""
    import ctypes
    from octopuszk_amd import lib
    L = lib.load()
    rng = np.random.default_rng(n + shards)
    G = o.G1 if type_ == 1 else o.G2
    to_wire = o.g1_to_wire if type_ == 1 else o.g2_to_wire
    pts = [G.to_affine(G.mul(G.one, int(k))) for k in rng.integers(1, 1 << 62, size=16)]
    pts[3] = G.zero
    bw = np.frombuffer(b"".join(to_wire(pts[i % 16]) for i in range(n)), dtype=np.uint8)
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x1F
    ob = 192 if type_ == 1 else 384
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    res = {}

    def call(name, fn, *args):
        out = np.zeros(ob, dtype=np.uint8)
        rc = fn(*args, vp(out))
        st = (ctypes.c_double * 10)()
        L.ozk_host_call_stats(st)
        res[name] = dict(rc=rc, err=(L.ozk_last_error() or b"").decode(errors="replace") if rc else "", out=bytes(out),
                         host_call_stats=list(st), exchange=L.ozk_shard_last_exchange())

    call("sharded_rccl", L.ozk_var_msm_sharded_host, vp(bw), vp(sc), n, type_, shards)
    os.environ["OZK_SHARD_RCCL"] = "0"
    lib.check(L.ozk_tuning_reload())
    try:
        call("sharded_host", L.ozk_var_msm_sharded_host, vp(bw), vp(sc), n, type_, shards)
    finally:
        del os.environ["OZK_SHARD_RCCL"]
        lib.check(L.ozk_tuning_reload())
    call("single", L.ozk_var_msm_host, vp(bw), vp(sc), n, type_, 0)
    if n <= 6000:
        scal = [int.from_bytes(sc[i].tobytes(), "little") for i in range(n)]
        want = G.to_affine(o.pippenger_msm(G, scal, [pts[i % 16] for i in range(n)]))
        want = o.g1_out_le(want) if type_ == 1 else o.g2_out_le(want)
    elif type_ == 1:
        want = coracle.pippenger_g1(bw.tobytes(), sc.tobytes(), n)
    else:
        want = res["single"]["out"]
    ok = all(r["rc"] == 0 and r["out"] == want for r in res.values())
    k = min(shards, n)
    if k > 1:
        ok = ok and res["sharded_rccl"]["exchange"] == 1 and res["sharded_host"]["exchange"] == 0
    if not ok:
        _persist_failure(tag, type=type_, n=n, shards=shards, want=want,
                         **{name: {kk: (vv.hex() if isinstance(vv, bytes) else vv) for kk, vv in r.items()} for name, r in res.items()})
    for name, r in res.items():
        assert r["rc"] == 0, "%s returned %d: %s" % (name, r["rc"], r["err"])
    # (which side is wrong, should they ever differ)
    assert res["single"]["out"] == want, "single call differs from the oracle"
    assert res["sharded_host"]["out"] == want, "sharded call (host exchange) differs from the oracle (the single call agrees)"
    assert res["sharded_rccl"]["out"] == want, "sharded call (RCCL exchange) differs from the oracle (the single call agrees)"
    if k > 1:
        assert res["sharded_rccl"]["exchange"] == 1, "the RCCL form did not run (librccl missing, or ncclCommInitAll failed)"
        assert res["sharded_host"]["exchange"] == 0


@pytest.mark.parametrize("type_,n,min_log", [(1, (1 << 19) + 37, 18), (1, 70001, 14), (2, 9001, 11), (1, 4099, 12)])
def test_host_entry_sliced_upload_equals_single_slice(type_, n, min_log, monkeypatch):
    """ozk_var_msm_host cuts a large call into index-range slices (each sorted and accumulated into its own bucket
    array behind its own upload; the arrays are added and ONE tail finishes): the bytes must be those of the
    unsliced call, and for G1 the C oracle's over the whole input.  Repeated bases, so that the same point meets
    itself when bucket arrays are added; the last slice is padded."""
    import ctypes
    from octopuszk_amd import lib
    L = lib.load()
    rng = np.random.default_rng(n)
    G = o.G1 if type_ == 1 else o.G2
    to_wire = o.g1_to_wire if type_ == 1 else o.g2_to_wire
    pts = [G.to_affine(G.mul(G.one, int(k))) for k in rng.integers(1, 1 << 62, size=32)]
    pts[5] = G.zero
    wire = [np.frombuffer(to_wire(p), dtype=np.uint8) for p in pts]
    bw = np.ascontiguousarray(np.stack(wire)[rng.integers(0, 32, size=n)]).reshape(-1)
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x1F
    sc[: n // 3, 2:] = 0          # a third of the scalars tiny: whole windows of some slices stay empty
    ob = 192 if type_ == 1 else 384
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    outs = {}
    try:
        monkeypatch.setenv("OZK_HOST_SLICE_MIN_LOG", str(min_log))
        for k in (8, 1, 3):
            monkeypatch.setenv("OZK_HOST_SLICES", str(k))
            lib.check(L.ozk_tuning_reload())
            got = np.zeros(ob, dtype=np.uint8)
            lib.check(L.ozk_var_msm_host(vp(bw), vp(sc), n, type_, 0, vp(got)))
            outs[k] = bytes(got)
    finally:
        monkeypatch.delenv("OZK_HOST_SLICES", raising=False)
        monkeypatch.delenv("OZK_HOST_SLICE_MIN_LOG", raising=False)
        lib.check(L.ozk_tuning_reload())
    assert outs[8] == outs[1] == outs[3]
    if type_ == 1:
        assert outs[8] == coracle.pippenger_g1(bw.tobytes(), sc.tobytes(), n)


@pytest.mark.parametrize("n,min_log", [(9001, 11), (70001, 14)])
def test_double_msm_host_sliced_equals_two_single_calls(n, min_log, monkeypatch):
    """ozk_var_double_msm_host above the slicing threshold: the scalars go up once, G1 and G2 bases slice by slice,
    one tail per curve — the 576 bytes must be the G1 and the G2 result of the unsliced single calls."""
    import ctypes
    from octopuszk_amd import lib
    L = lib.load()
    rng = np.random.default_rng(n)
    p1 = [o.G1.to_affine(o.G1.mul(o.G1.one, int(k))) for k in rng.integers(1, 1 << 62, size=16)]
    p2 = [o.G2.to_affine(o.G2.mul(o.G2.one, int(k))) for k in rng.integers(1, 1 << 62, size=16)]
    p1[3], p2[7] = o.G1.zero, o.G2.zero
    pick = rng.integers(0, 16, size=n)
    b1 = np.ascontiguousarray(np.stack([np.frombuffer(o.g1_to_wire(p), dtype=np.uint8) for p in p1])[pick]).reshape(-1)
    b2 = np.ascontiguousarray(np.stack([np.frombuffer(o.g2_to_wire(p), dtype=np.uint8) for p in p2])[pick]).reshape(-1)
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x1F
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    try:
        monkeypatch.setenv("OZK_HOST_SLICES", "1")
        lib.check(L.ozk_tuning_reload())
        one1, one2, dbl1 = np.zeros(192, dtype=np.uint8), np.zeros(384, dtype=np.uint8), np.zeros(576, dtype=np.uint8)
        lib.check(L.ozk_var_msm_host(vp(b1), vp(sc), n, 1, 0, vp(one1)))
        lib.check(L.ozk_var_msm_host(vp(b2), vp(sc), n, 2, 0, vp(one2)))
        lib.check(L.ozk_var_double_msm_host(vp(b1), vp(b2), vp(sc), n, 0, vp(dbl1)))
        monkeypatch.setenv("OZK_HOST_SLICES", "4")
        monkeypatch.setenv("OZK_HOST_SLICE_MIN_LOG", str(min_log))
        lib.check(L.ozk_tuning_reload())
        dbl4 = np.zeros(576, dtype=np.uint8)
        lib.check(L.ozk_var_double_msm_host(vp(b1), vp(b2), vp(sc), n, 0, vp(dbl4)))
    finally:
        monkeypatch.delenv("OZK_HOST_SLICES", raising=False)
        monkeypatch.delenv("OZK_HOST_SLICE_MIN_LOG", raising=False)
        lib.check(L.ozk_tuning_reload())
    assert bytes(dbl1) == bytes(one1) + bytes(one2)
    assert bytes(dbl4) == bytes(dbl1)


@pytest.mark.parametrize("n,world", [(4096, 2), (5001, 3)])
def test_sharded_g1_msm_equals_oracle_on_the_whole_input(n, world):
    rng = np.random.default_rng(n)
    G = o.G1
    pts = [G.to_affine(G.mul(G.one, int(k))) for k in rng.integers(1, 1 << 62, size=64)]
    bw = b"".join(o.g1_to_wire(pts[i % 64]) for i in range(n))
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x1F
    sw = sc.tobytes()
    want = coracle.pippenger_g1(bw, sw, n)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, 1, bw, sw, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in range(world):
        assert got[r] == want, r


def test_bench_two_ranks_on_one_gpu_result_is_the_global_msm():
    """bench.py's N = 2 path on one GPU (OZK_BENCH_REHEARSAL=1: gloo instead of RCCL): the contract line, and
    the printed result equals (sum over BOTH ranks' inputs of s_i k_i) G — bench.py's bases are k_i G with
    k_i = splitmix64(seed + i), so the expected point is exact integer arithmetic plus one scalar multiplication."""
    from octopuszk_amd import device as dev
    logn = 12
    env = dict(os.environ, OZK_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4",
           "--warmup", "1", "--logn", str(logn)]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["unit"] == "Mscalar-mul/s"
    assert d["value"] > 0 and d["cpu_baseline"] is None and d["roofline"]["bound"] == "hbm"
    assert "x2" in d["config"]["parallelism"]
    n = 1 << logn
    sys.path.insert(0, ROOT)
    import bench
    acc = 0
    for rank in range(2):
        ks = dev.gen_base_logs(n, bench.base_seed(rank))
        sc = bench.rand_scalars(n, bench.scalar_seed(rank)).reshape(n, 32)
        acc += sum(k * int.from_bytes(s.tobytes(), "little") for k, s in zip(ks, sc))
    want = o.g1_out_le(o.G1.to_affine(o.G1.mul(o.G1.one, acc % o.R)))
    assert bytes.fromhex(d["config"]["result_hex"]) == want


def _gpu_children(n_children=3):
    """n child processes that each run one small MSM on cuda:0 and exit — what the multi-process tests above leave
    behind them — returning once every child has been reaped"""
    code = ("import numpy as np, torch, sys; sys.path.insert(0, %r); from octopuszk_amd import device as dev; "
            "b = dev.gen_g1_bases(3000, seed=3); s = torch.from_numpy(np.random.default_rng(1).integers(0, 256, 3000 * 32, "
            "dtype=np.uint8)).cuda(); ws = dev.VarMsmWorkspace(3000, 1); o = ws.run(b, s); torch.cuda.synchronize()" % ROOT)
    procs = [subprocess.Popen([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
             for _ in range(n_children)]
    for p in procs:
        _, err = p.communicate(timeout=300)
        assert p.returncode == 0, err.decode()[-2000:]


def test_in_process_sharded_call_right_after_gpu_children_exit():
    """The condition of round 3's one unexplained failure, made deterministic: three child processes use
    the GPU and exit, and the FIRST thing this process does is the in-process sharded call — three concurrent
    1667-pair MSMs on one device, two of them on contexts created at that moment.  Everything is persisted on
    failure (_sharded_case)."""
    from octopuszk_amd import lib
    L = lib.load()
    L.ozk_host_cache_release()     # the three slices create their contexts now, as the first in-process test did
    _gpu_children(3)
    _sharded_case(1, 5001, 3, "after_children")


@pytest.mark.parametrize("type_,n,shards", [(1, 5001, 3), (1, 4, 8), (2, 301, 2), (1, 70000, 4)])
def test_in_process_sharded_entry_equals_single_call(type_, n, shards):
    """ozk_var_msm_sharded_host (what the JNI native routes large calls to with OZK_SHARD=1): the slices run
    concurrently from their own host threads — here all on device 0 — and the summed result, by either exchange
    form, must be the single-call bytes, which are the oracle's."""
    _sharded_case(type_, n, shards, "%d_%d_%d" % (type_, n, shards))


@pytest.mark.parametrize("n,shards", [(3001, 3), (40000, 2)])
def test_in_process_sharded_double_msm_equals_the_single_double_call(n, shards):
    """ozk_var_double_msm_sharded_host (VariableBaseMSM.distributedDoubleMSM, VariableBaseMSM.java:805-818): 576 bytes
    = the G1 and the G2 result of the unsharded double call, by both exchange forms."""
    import ctypes
    from octopuszk_amd import lib
    L = lib.load()
    rng = np.random.default_rng(n)
    p1 = [o.G1.to_affine(o.G1.mul(o.G1.one, int(k))) for k in rng.integers(1, 1 << 62, size=16)]
    p2 = [o.G2.to_affine(o.G2.mul(o.G2.one, int(k))) for k in rng.integers(1, 1 << 62, size=16)]
    p1[3], p2[7] = o.G1.zero, o.G2.zero
    pick = rng.integers(0, 16, size=n)
    b1 = np.ascontiguousarray(np.stack([np.frombuffer(o.g1_to_wire(p), dtype=np.uint8) for p in p1])[pick]).reshape(-1)
    b2 = np.ascontiguousarray(np.stack([np.frombuffer(o.g2_to_wire(p), dtype=np.uint8) for p in p2])[pick]).reshape(-1)
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x1F
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    one = np.zeros(576, dtype=np.uint8)
    lib.check(L.ozk_var_double_msm_host(vp(b1), vp(b2), vp(sc), n, 0, vp(one)))
    if n <= 6000:
        scal = [int.from_bytes(sc[i].tobytes(), "little") for i in range(n)]
        w1 = o.g1_out_le(o.G1.to_affine(o.pippenger_msm(o.G1, scal, [p1[i] for i in pick])))
        w2 = o.g2_out_le(o.G2.to_affine(o.pippenger_msm(o.G2, scal, [p2[i] for i in pick])))
        assert bytes(one) == w1 + w2
    rccl = np.zeros(576, dtype=np.uint8)
    lib.check(L.ozk_var_double_msm_sharded_host(vp(b1), vp(b2), vp(sc), n, shards, vp(rccl)))
    assert L.ozk_shard_last_exchange() == 1
    os.environ["OZK_SHARD_RCCL"] = "0"
    lib.check(L.ozk_tuning_reload())
    try:
        host = np.zeros(576, dtype=np.uint8)
        lib.check(L.ozk_var_double_msm_sharded_host(vp(b1), vp(b2), vp(sc), n, shards, vp(host)))
        assert L.ozk_shard_last_exchange() == 0
    finally:
        del os.environ["OZK_SHARD_RCCL"]
        lib.check(L.ozk_tuning_reload())
    assert bytes(host) == bytes(one)
    assert bytes(rccl) == bytes(one)
