#!/usr/bin/env python3
"""bench.py — BN254 G1 VariableBaseMSM throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one complete variable-base MSM of 2^20 (scalar, base) pairs per GPU, from the
reference's JNI wire bytes already resident in HBM (n x 96 B Jacobian bases + n x 32 B
scalars) to the 192-byte affine-normalised result.  With N GPUs every rank owns its own
2^20-pair range of one N*2^20 MSM (weak scaling), the ranks all-gather their 192-byte
partials over RCCL and each adds them up with the HIP point-sum kernel — the
Spark `mapPartitions -> reduce(add)` of VariableBaseMSM.java:777-783.

Steps overlap: by default three MSMs are in flight as sort | bucket accumulation | tail on their own streams
(device.VarMsmPipeline3; `--schedule pipeline` is round 2's head | tail form, `--schedule streams` issues complete
MSMs round-robin on independent streams; see --help).  Prints ONE JSON line (rank 0).  `value` = N * 2^20 * K / wall-time / 1e6 Mscalar-mul/s,
max wall-time over ranks, inputs resident in HBM when the timed region starts.
`roofline` is for the dominant kernel (level-1 bucket accumulation): algorithmic bytes
(128 B per scalar-mul) / its duration inside the timed region, against the 8 TB/s HBM peak.  The
duration comes from the device clock the kernel's own waves stamp (`kernel_ms`); the same K steps run once more with
HIP start / stop events on every level-1 dispatch (`kernel_ms_hip_events`: an event-carrying dispatch slows the
three-stage schedule, so it stays out of the timed region — that pass runs FIRST, in front of the W warm-up steps, so
that the timed steps see the clock the chip settles at under this load and not its ramp from idle), and five lone MSMs
after the timed region give the kernel alone (`kernel_ms_alone`).  `roofline.shader_clock_mhz` is the shader clock the
level-1 launches ran at, stamped by the kernel itself.
`cpu_baseline` times the C oracle (oracle/ozk_oracle.c, a single-thread C port of the
reference's serial Java pippengerMSM) on the SAME 2^20 inputs on one host core — rank 0,
N = 1 only — and the bench asserts the GPU bytes equal the CPU bytes.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOGN = 20
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
ALG_BYTES_PER_MUL = 128  # 32 B scalar + 96 B base, SURVEY.md §8(d)


def rand_scalars(n, seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    b = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    b[:, 31] &= 0x1F  # uniform in [0, 2^253) subset of [0, r)
    return b.reshape(-1)


def java_baseline(bases, sc_host, n, result_bytes):
    """when the box has a JDK, time bench/java/SerialPippenger.java — the reference's serial
    pippengerMSM restated over java.math.BigInteger — on one core, on a bounded sample (2^16 pairs, about 20 s),
    and check its point against the GPU's when the sample is the whole workload.  None without a JDK."""
    import shutil
    import subprocess
    import tempfile
    if not (shutil.which("javac") and shutil.which("java")):
        return None
    ns = min(n, 1 << 16)
    src = os.path.join(ROOT, "bench", "java", "SerialPippenger.java")
    try:
        with tempfile.TemporaryDirectory() as td:
            subprocess.check_call(["javac", "-d", td, src], timeout=120)
            inp = os.path.join(td, "in.bin")
            with open(inp, "wb") as f:
                f.write(bytes(bases[:ns * 96].cpu().numpy()))
                f.write(bytes(sc_host[:ns * 32]))
            out = subprocess.check_output(["java", "-cp", td, "-Xmx8g", "SerialPippenger", inp, str(ns)], timeout=600)
        secs, x, y, z = out.decode().split()
        if ns == n:
            got = int(x, 16).to_bytes(64, "little") + int(y, 16).to_bytes(64, "little") + int(z).to_bytes(64, "little")
            if got != result_bytes:
                # (this file has never met a JDK in the build container: a mismatch is reported, and the C port,
                # which IS checked against the oracle's golden vectors, stays the baseline)
                sys.stderr.write("bench: bench/java/SerialPippenger.java disagrees with the GPU result; ignoring it\n")
                return None
        return {"value": round(ns / float(secs) / 1e6, 6), "unit": "Mscalar-mul/s", "cores": 1, "kind": "port",
                "sample": "bench/java/SerialPippenger.java (VariableBaseMSM.pippengerMSM restated over java.math.BigInteger, "
                          "one thread) on the first 2^%d pairs of the workload, %.1f s" % (ns.bit_length() - 1, float(secs))}
    except (subprocess.SubprocessError, OSError, ValueError):
        return None


def clock_stats(L):
    """shader clock (MHz) the level-1 launches recorded since the last ozk_prof_enable(2) ran at, stamped by the kernel
    itself: shader-clock ticks / constant-rate ticks inside the kernel (VERDICT r3 "next" 7: the sysfs value read after
    the run was noise — 94 / 106 / 540 / 1713 / 2400 MHz for the same workload)"""
    from octopuszk_amd import lib as ozk
    c4, cl = (ctypes.c_double * 4)(), ctypes.c_int()
    ozk.check(L.ozk_prof_dominant_kernel_clock_mhz(c4, ctypes.byref(cl)))
    if cl.value == 0:
        return None
    return {"mean": round(c4[0], 1), "median": round(c4[1], 1), "min": round(c4[2], 1), "max": round(c4[3], 1),
            "launches": cl.value}


def base_seed(rank):
    return 2 + (rank << 32)


def scalar_seed(rank):
    return 1 + rank


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--logn", type=int, default=LOGN)
    ap.add_argument("--total-logn", type=int, default=None,
                    help="STRONG scaling: one MSM of 2^TOTAL_LOGN pairs sharded over the ranks (BASELINE.json configs[3]: "
                         "24 over 8 GPUs = 2^21 per rank); the default is weak scaling, 2^LOGN pairs per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-streams-leg", action="store_true", help="skip the secondary three-streams throughput figure")
    ap.add_argument("--timed-only", action="store_true",
                    help="nothing but warm-up + the timed steps (no single-MSM latency loop, no three-streams leg): the form "
                         "profiled with rocprofv3, so that its per-kernel averages are those of the timed schedule")
    ap.add_argument("--prepared", action="store_true",
                    help="bases prepared once outside the timed region (ozk_var_msm_prepare_dev): NOT the "
                         "BASELINE.json workload, whose every MSM starts from the JNI wire bytes")
    ap.add_argument("--in-flight", type=int, default=None,
                    help="MSMs in flight (default: 3 for --schedule streams, 2 for pipeline; 1 = strictly serial steps)")
    ap.add_argument("--tail-streams", type=int, default=2, help="pipeline3: streams the tails alternate on")
    ap.add_argument("--tail-cus", type=int, default=int(os.environ.get("OZK_BENCH_TAIL_CUS", "0")),
                    help="pipeline3: confine the tail streams to this many compute units and the accumulate stream to "
                         "the others (device.VarMsmPipeline3(tail_cus=...)); the steps then run on a side stream, "
                         "because CU-masked streams synchronise with the null stream.  0 = no partition")
    ap.add_argument("--schedule", choices=["streams", "pipeline", "pipeline3"], default=os.environ.get("OZK_BENCH_SCHEDULE", "pipeline3"),
                    help="pipeline3 (default): sort of MSM k+1 | bucket accumulation of MSM k | tail of MSM k-1 on their own "
                         "streams (device.VarMsmPipeline3); "
                         "pipeline (rounds 1-2): one head stream + one tail stream with the launch-order hint "
                         "(device.VarMsmPipeline) — the level-1 kernel runs alone, so its HIP-event duration is the "
                         "kernel's own; streams: complete MSMs issued round-robin on independent streams, the way "
                         "concurrent prover threads drive the JNI — 5-8 %% more throughput at 3 in flight (590-620 "
                         "Mscalar-mul/s; the default run reports it as a secondary figure), but the overlapping level-1 "
                         "kernels stretch each other's duration, which would distort `roofline`")
    args = ap.parse_args()
    if args.in_flight is None:
        args.in_flight = 2 if args.schedule == "pipeline" else 3

    import torch
    import torch.distributed as dist
    from octopuszk_amd import device as dev
    from octopuszk_amd import lib as ozk

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs the torch.distributed.run launcher (WORLD_SIZE=%d)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # OZK_BENCH_REHEARSAL=1 (tests only): all ranks share cuda:0 and exchange over gloo, so that the
    # N > 1 control path — sharding, all-gather of the partials, HIP point sum, max-over-ranks timing —
    # can be exercised on a one-GPU box.  RCCL refuses two ranks on one device; the numbers mean nothing.
    rehearsal = os.environ.get("OZK_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # OZK_BENCH_FORCE_COLLECTIVE=1 (tests only): take the N > 1 code path — init_process_group("nccl"), the RCCL
    # all-gather of the partial and the HIP point sum inside every step — with a world of one, so that it runs on
    # hardware on a one-GPU box (the sum of one affine point is that point)
    force_coll = os.environ.get("OZK_BENCH_FORCE_COLLECTIVE", "0") == "1" and "RANK" in os.environ
    if world > 1 or force_coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    L = ozk.load()
    n = 1 << args.logn
    strong = args.total_logn is not None
    if strong:
        if (1 << args.total_logn) % world:
            raise SystemExit("--total-logn: 2^%d pairs do not divide over %d ranks" % (args.total_logn, world))
        n = (1 << args.total_logn) // world
    # inputs: rank r owns pairs [r*n, (r+1)*n) of the global MSM (distinct seeds per rank)
    # Order of the set-up (round 4): everything that leaves the device idle — the scalars (host), the allocations, the
    # calibration of the device clock (two reads 25 ms apart, ozk_prof_enable) — comes FIRST, the generation of the
    # synthetic bases (k_i G by 64-step double-and-add: ~70 ms of vector-ALU work) LAST, right in front of the warm-up.
    # The chip raises its shader clock over tens of milliseconds of load: with the calibration's idle period in front
    # of a 5-step (8 ms) warm-up, the 20 timed steps of the driver's command ran at 2.02-2.08 GHz against 2.24 GHz for
    # the same schedule over 100 steps (roofline.shader_clock_mhz, profiles/r04_bench_*.json) — 10 % of clock that has
    # nothing to do with the kernels.
    sc_host = rand_scalars(n, scalar_seed(rank))
    scalars = torch.from_numpy(sc_host).cuda()
    if args.schedule == "pipeline3" and args.in_flight >= 2:
        pipe = dev.VarMsmPipeline3(n, 1, depth=args.in_flight, tail_streams=args.tail_streams, tail_cus=args.tail_cus)
    else:
        pipe = dev.VarMsmPipeline(n, 1, depth=max(1, args.in_flight))
    ozk.check(L.ozk_prof_enable(2))   # (calibrates and allocates; enabled again, cleared, in front of the timed steps)
    ozk.check(L.ozk_prof_enable(0))
    bases = dev.gen_g1_bases(n, seed=base_seed(rank))
    msm_bases = pipe.prepare(bases) if args.prepared else bases
    wb, wn = ctypes.c_int32(), ctypes.c_int32()
    ozk.check(L.ozk_var_msm_plan(n, ctypes.byref(wb), ctypes.byref(wn)))
    glv = int(L.ozk_var_msm_glv(n))
    adds = n * (2 if glv else 1) * wn.value  # bucket additions of the level-1 kernel (upper bound: zero digits skipped)

    from octopuszk_amd import distributed as ozk_dist

    def finish(ticket):
        # the step's result; for N > 1: RCCL all-gather of the 192-B partials + HIP point sum.
        # Issued on the pipeline's side stream (behind the tail it consumes), so the single-lane
        # point sum does not sit between two heads on the main stream.
        three = isinstance(pipe, dev.VarMsmPipeline3)
        solo = world == 1 and not force_coll
        if solo and not three:
            return pipe.result(ticket)
        # three-stage schedule: the caller's stream is the SORT stream and must not wait for a tail, so the result
        # is taken on the stream that ran this ticket's tail (the final barrier synchronises the device)
        side = pipe.stream_of(ticket) if three else pipe.side
        with torch.cuda.stream(side):
            if solo:
                return pipe.result(ticket)
            out = ozk_dist.distributed_var_msm(lambda: pipe.result(ticket), dev.points_sum, 1, always_collective=force_coll)
            done = torch.cuda.Event()
            done.record(side)
        finish.last_event = done
        return out

    finish.last_event = None

    S = max(1, args.in_flight)
    wss = [dev.VarMsmWorkspace(n, 1) for _ in range(S)] if args.schedule == "streams" else []
    sts = [torch.cuda.Stream() for _ in range(S)] if args.schedule == "streams" else []
    issued = [0]

    def run_steps_streams(k):
        """k complete MSMs, round-robin over S independent streams (each MSM: head and tail on its stream)."""
        res = None
        for _ in range(k):
            i = issued[0] % S
            issued[0] += 1
            with torch.cuda.stream(sts[i]):
                res = ozk_dist.distributed_var_msm(lambda: wss[i].run(msm_bases, scalars, prepared=args.prepared),
                                                   dev.points_sum, 1, always_collective=force_coll)
        return res

    partitioned = isinstance(pipe, dev.VarMsmPipeline3) and pipe.tail_cus > 0
    sort_stream = torch.cuda.Stream() if partitioned else None

    def run_steps(k):
        """k complete MSMs; step i's tail overlaps step i+1's head (args.in_flight > 1)."""
        if args.schedule == "streams":
            return run_steps_streams(k)
        if partitioned:   # (the sort stage on a stream of this process's own: see --tail-cus)
            with torch.cuda.stream(sort_stream):
                return run_steps_on_current_stream(k)
        return run_steps_on_current_stream(k)

    def run_steps_on_current_stream(k):
        res, prev = None, None
        three = isinstance(pipe, dev.VarMsmPipeline3) and os.environ.get("OZK_BENCH_LAST_LATENCY", "1") != "0"
        for i in range(k):
            # (the last MSM of a burst of k: nothing follows it, so its tail takes the latency shape — VarMsmPipeline3)
            t = (pipe.submit(msm_bases, scalars, prepared=args.prepared, last=(i == k - 1)) if three
                 else pipe.submit(msm_bases, scalars, prepared=args.prepared))
            if pipe.depth == 1:
                res = finish(t)
                continue
            if prev is not None:
                res = finish(prev)
            prev = t
        return finish(prev) if prev is not None else res

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The HIP-event cross-check pass (the contract's measurement of the dominant kernel: K steps of the same schedule
    # with start / stop events on every level-1 dispatch, which slow the schedule and therefore stay out of the timed
    # region) runs FIRST, then the W warm-up steps, then the K timed steps.  Round 4 moved it in front: the chip raises
    # its shader clock over the first ~40 ms of this load — 2.03-2.08 GHz during 20 timed steps that follow a 5-step
    # warm-up from idle, 2.20 after 25 steps, 2.24-2.26 after 60 (roofline.shader_clock_mhz; same box: 607-615 / 641 /
    # 647-658 Mscalar-mul/s, profiles/r04_clock_ramp.txt) — and a run of 25 steps measured the ramp, not the kernels.
    ev_stats, ev_launches, ev_ms_per_step, res_ev_bytes = None, 0, None, None
    if not args.timed_only:
        ozk.check(L.ozk_prof_enable(1))
        run_steps(1)   # (first use of the event pool)
        barrier()
        ozk.check(L.ozk_prof_enable(1))
        e0 = time.perf_counter()
        res_ev = run_steps(args.steps)
        barrier()
        e1 = time.perf_counter()
        es, el = (ctypes.c_double * 4)(), ctypes.c_int()
        ozk.check(L.ozk_prof_dominant_kernel_stats(es, ctypes.byref(el)))
        ozk.check(L.ozk_prof_enable(0))
        ev_stats, ev_launches, ev_ms_per_step = [float(x) for x in es], el.value, (e1 - e0) / args.steps * 1e3
        res_ev_bytes = bytes(res_ev.cpu().numpy())
    # (the device clock was calibrated above, before the bases were generated: this enable only clears the records)
    ozk.check(L.ozk_prof_enable(2))
    res = run_steps(max(1, args.warmup))
    barrier()
    # level-1 kernel duration per launch, from the device clock stamped by the kernel's own waves (ozk_prof_enable(2))
    ozk.check(L.ozk_prof_enable(2))
    t0 = time.perf_counter()
    res = run_steps(args.steps)
    barrier()
    t1 = time.perf_counter()
    kstats, launches = (ctypes.c_double * 4)(), ctypes.c_int()
    ozk.check(L.ozk_prof_dominant_kernel_stats(kstats, ctypes.byref(launches)))
    clock_timed = clock_stats(L)
    ozk.check(L.ozk_prof_enable(0))
    avg_ms = ctypes.c_double(kstats[0])
    result_bytes = bytes(res.cpu().numpy())
    if res_ev_bytes is not None and res_ev_bytes != result_bytes:
        raise SystemExit("bench: the HIP-event pass returned a different point")
    # latency of ONE MSM with nothing else in flight (not part of `value`), and the level-1 kernel's duration in that
    # situation: inside the timed schedule the kernel shares the vector ALU with the next MSM's sort and the previous
    # MSMs' tails, so its duration there says how the chip was shared, its duration alone what the kernel costs
    lat = []
    alone = None
    clock_alone = None
    if not args.timed_only:
        # (through the single-call entry point ozk_var_msm_dev, the way one caller with one MSM uses the library: its
        # tail takes the latency shape; a lone MSM pushed through the pipeline object would get the throughput-shaped
        # tail, 0.2-0.3 ms slower when nothing runs beside it)
        lone = dev.VarMsmWorkspace(n, 1)
        lone.run(msm_bases, scalars, prepared=args.prepared)
        ozk.check(L.ozk_prof_enable(2))
        for _ in range(5):
            torch.cuda.synchronize()
            l0 = time.perf_counter()
            lone_out = lone.run(msm_bases, scalars, prepared=args.prepared)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - l0)
        if world == 1 and bytes(lone_out.cpu().numpy()) != result_bytes:
            raise SystemExit("bench: the single-call entry point returned a different point")
        a4, al = (ctypes.c_double * 4)(), ctypes.c_int()
        ozk.check(L.ozk_prof_dominant_kernel_stats(a4, ctypes.byref(al)))
        clock_alone = clock_stats(L)
        ozk.check(L.ozk_prof_enable(0))
        alone = {"mean": round(a4[0], 4), "median": round(a4[1], 4), "min": round(a4[2], 4), "max": round(a4[3], 4),
                 "launches": al.value, "source": "device clock, five lone MSMs after the timed region"}
    single_ms = sorted(lat)[len(lat) // 2] * 1e3 if lat else 0.0
    # Secondary figure (never `value`): the same MSMs issued as three free-running streams — how concurrent
    # prover threads drive the library.  Higher throughput (everything but the bucket accumulation of one MSM runs
    # inside the accumulation of the others), but the co-running kernels stretch each accumulation launch, so the
    # default schedule, which keeps that kernel's duration clean for `roofline`, stays the timed one.
    streams3 = None
    if world == 1 and args.schedule == "pipeline" and args.in_flight > 1 and not (args.no_streams_leg or args.timed_only):
        ws3 = [dev.VarMsmWorkspace(n, 1) for _ in range(3)]
        # the runtime multiplexes streams onto 4 hardware queues (GPU_MAX_HW_QUEUES): two streams on one queue run
        # in order, so reuse the two streams this process already has rather than add three to them
        st3 = [torch.cuda.current_stream(), pipe.side, torch.cuda.Stream()]
        k3 = max(60, args.steps)           # (fill and drain of three streams cost ~2 MSM times)
        for rep in range(2):            # first round: warm-up
            torch.cuda.synchronize()
            s0 = time.perf_counter()
            for i in range(k3):
                with torch.cuda.stream(st3[i % 3]):
                    ws3[i % 3].run(msm_bases, scalars, prepared=args.prepared)
            torch.cuda.synchronize()
            s1 = time.perf_counter()
        streams3 = round(n * k3 / (s1 - s0) / 1e6, 3)
        del ws3

    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        value = world * n * args.steps / elapsed / 1e6
        alg_bytes = ALG_BYTES_PER_MUL * n
        k_ms = avg_ms.value
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        # PMC counters cannot be collected inside this run (rocprofv3 owns them): `traffic` is the per-launch
        # figure of the last counter collection (tools/collect_profiles_r02.sh), with the commit it was taken on
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("k_segreduce_level1_hbm_bytes_per_launch")
                traffic_src = "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, commit %s)" % tj.get("collected_on_commit", "?")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                    "kernel": "k_segreduce<G1Cfg,true> (level-1 bucket accumulation)",
                    "kernel_avg_ms": round(k_ms, 4), "launches_timed": launches.value,
                    "kernel_ms": {"mean": round(kstats[0], 4), "median": round(kstats[1], 4), "min": round(kstats[2], 4),
                                  "max": round(kstats[3], 4),
                                  "source": "device clock stamped by the kernel's waves inside the timed region (first wave "
                                            "start to last wave end)"},
                    "kernel_ms_hip_events": None if ev_stats is None else {
                        "mean": round(ev_stats[0], 4), "median": round(ev_stats[1], 4), "min": round(ev_stats[2], 4),
                        "max": round(ev_stats[3], 4), "launches": ev_launches,
                        "ms_per_step_of_that_pass": round(ev_ms_per_step, 4),
                        "source": "HIP start/stop events on the dispatch, a separate pass of the same steps in front of the warm-up (an event-carrying "
                                  "dispatch slows the three-stage schedule, hence not inside the timed region)"},
                    "kernel_ms_alone": alone,
                    "shader_clock_mhz": {"in_schedule": clock_timed, "alone": clock_alone,
                                         "source": "shader-clock / constant-rate ticks stamped inside the level-1 kernel"},
                    "algorithmic_bytes_per_launch": alg_bytes,
                    # the bound that actually holds: v_mad_u64_u32 issue.  10 Montgomery
                    # multiplications per XYZZ mixed addition x points x windows, against the 179 G mulmod/s
                    # this chip sustains on back-to-back multiplications in the shipped form (per-column
                    # v_mad_u64_u32 chains, 8 waves per SIMD: profiles/r02_ubench_mont.txt, variant E)
                    "alu": {"achieved": round(10.0 * adds / (k_ms * 1e-3) / 1e9, 1) if k_ms > 0 else 0.0,
                            "peak": 179.0, "unit": "G mulmod/s",
                            "frac": round(10.0 * adds / (k_ms * 1e-3) / 1e9 / 179.0, 3) if k_ms > 0 else 0.0},
                    "note": "integer-ALU-bound: 10 Fq mulmod per XYZZ mixed add x %d points x %d windows%s; see DESIGN.md"
                            % (n * (2 if glv else 1), wn.value, " (GLV: 2n half-length scalars)" if glv else "")}
        if alone and alone["median"] > 0:
            # the same ratio for the kernel running alone: what the kernel itself reaches of the multiplier's rate
            ach = 10.0 * adds / (alone["median"] * 1e-3) / 1e9
            roofline["alu_alone"] = {"achieved": round(ach, 1), "peak": 179.0, "unit": "G mulmod/s", "frac": round(ach / 179.0, 3)}
            roofline["frac_alone"] = round(alg_bytes / (alone["median"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import coracle  # checker / baseline only
            # bounded sample (~13 s of CPU work): the whole workload up to 2^20 pairs — then the GPU bytes must
            # equal the CPU's —, else its first 2^20 pairs (timing only)
            ns = min(n, 1 << 20)
            bases_host = bytes(bases[:ns * 96].cpu().numpy())
            sc_bytes = bytes(sc_host[:ns * 32])
            c0 = time.perf_counter()
            cpu_out = coracle.pippenger_g1(bases_host, sc_bytes, ns)
            c1 = time.perf_counter()
            if ns == n and cpu_out != result_bytes:
                raise SystemExit("PARITY FAILURE: GPU result differs from the CPU oracle on the bench inputs")
            cpu = {"value": round(ns / (c1 - c0) / 1e6, 5), "unit": "Mscalar-mul/s", "cores": 1, "kind": "port",
                   "sample": ("the full 2^%d-pair workload, same inputs, C port of VariableBaseMSM.pippengerMSM "
                              "(c=14, 254 bits), %.1f s; result bytes equal the GPU's" % (args.logn, c1 - c0)) if ns == n else
                             ("the first 2^20 of the %d pairs, C port of VariableBaseMSM.pippengerMSM, %.1f s" % (n, c1 - c0))}
            # with a JDK on the box the reported baseline is the BigInteger restatement (closer to the reference's
            # arithmetic than the 64-bit Montgomery C port, which stays as the parity checker above)
            jb = java_baseline(bases, sc_host, n, result_bytes)
            if jb is not None:
                jb["c_port_value"] = cpu["value"]
                cpu = jb
        line = {"metric": "BN254 G1 VariableBaseMSM Mscalar-mul/s at 2^%d" % (args.total_logn if strong else args.logn),
                "value": round(value, 3),
                "unit": "Mscalar-mul/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
                "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                "config": {"workload": ("VariableBaseMSM BN254 G1 2^%d random scalars/bases in total, index-range sharded "
                                        "over the ranks (BASELINE.json configs[3])" % args.total_logn) if strong else
                                       ("VariableBaseMSM BN254 G1 2^%d random scalars/bases per GPU, bit-exact vs "
                                        "the serial CPU path (BASELINE.json configs[1])" % args.logn),
                           "n_per_gpu": n, "window_bits": wb.value, "windows": wn.value, "glv": bool(glv),
                           "prepared_bases": bool(args.prepared),
                           "msms_in_flight": max(1, args.in_flight), "schedule": args.schedule,
                           "cu_partition": ({"tail": pipe.tail_cus, "accumulate": int(L.ozk_device_cu_count()) - pipe.tail_cus}
                                            if partitioned else None),
                           "single_msm_latency_ms": round(single_ms, 3),
                           "three_streams_Mscalar_mul_s": streams3,
                           "result_hex": result_bytes.hex(),
                           "parallelism": "index-range shard x%d, RCCL all-gather of 192-B partials + HIP point sum" % world,
                           "collective_backend": (dist.get_backend() if dist.is_initialized() else None)},
                "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(line), flush=True)
    if world > 1 or force_coll:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
