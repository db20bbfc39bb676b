#!/usr/bin/env python3
"""Throughput of consecutive 2^logn G1 MSMs under one schedule, without bench.py's extras.
usage: sched_probe.py [--sched p3|p2] [--reps N] [--depth D] [--tail-streams T] [--prof] [--logn L] [--prepared]
                      [--own-sort-stream]"""
import argparse, ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev, lib as ozk
ap = argparse.ArgumentParser()
ap.add_argument("--sched", default="p3")
ap.add_argument("--reps", type=int, default=100)
ap.add_argument("--depth", type=int, default=3)
ap.add_argument("--tail-streams", type=int, default=2)
ap.add_argument("--logn", type=int, default=20)
ap.add_argument("--prof", type=int, default=0)
ap.add_argument("--prepared", action="store_true")
ap.add_argument("--own-sort-stream", action="store_true")
ap.add_argument("--idle-ms", type=float, default=None, help="with --fit: sleep this long, run 5 warm-up MSMs, then time (the driver's bench form after an idle device)")
ap.add_argument("--finish", action="store_true", help="with --fit: take every result on its tail stream as bench.py does")
ap.add_argument("--gc-off", action="store_true")
ap.add_argument("--ramp", action="store_true", help="after 300 ms of idle device: 16 back-to-back bursts of 10 MSMs, each timed, with the shader clock read after each")
ap.add_argument("--fit", action="store_true", help="time 10/20/40/100/200 MSMs (three times each) in one process: total = c0 + K * s")
ap.add_argument("--sort-prio", type=int, default=None, help="run the sort stage on a new stream of this priority (-1 = high)")
ap.add_argument("--acc-prio", type=int, default=None)
ap.add_argument("--tail-prio", type=int, default=None)
ap.add_argument("--tail-cus", type=int, default=0, help="tail streams created with a CU mask of this many bits (hipExtStreamCreateWithCUMask)")
ap.add_argument("--acc-mask", default="full", help="full | comp (the accumulate stream gets the complement of the tail's mask)")
ap.add_argument("--sort-mask", default="full", help="full | tail (sort stage on a stream with the tail's mask) | comp")
ap.add_argument("--class-tail-cus", type=int, default=0, help="VarMsmPipeline3(tail_cus=N): the shipped form of --tail-cus N --acc-mask comp (runs on a side stream)")
ap.add_argument("--rest-stream", default="tail", help="with OZK_P3_SPLIT_ACCUM=1: tail | own (an unmasked stream) | comp (a stream with the accumulate mask)")
ap.add_argument("--mask-layout", default="low", help="low: bits [0, N) | high: bits [256 - N, 256)")
a = ap.parse_args()
L = ozk.load()
n = 1 << a.logn
bases = dev.gen_g1_bases(n, seed=2)
sc = np.random.default_rng(10).integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
d_sc = torch.from_numpy(sc.reshape(-1)).cuda()
pipe = dev.VarMsmPipeline3(n, 1, depth=a.depth, tail_streams=a.tail_streams, tail_cus=a.class_tail_cus) if a.sched == "p3" else dev.VarMsmPipeline(n, 1, depth=2)
b = pipe.prepare(bases) if a.prepared else bases
st = torch.cuda.Stream() if (a.own_sort_stream or a.class_tail_cus) else torch.cuda.current_stream()
if a.sort_prio is not None:
    st = torch.cuda.Stream(priority=a.sort_prio)
if a.acc_prio is not None and a.sched == "p3":
    pipe.acc = torch.cuda.Stream(priority=a.acc_prio)
if a.tail_prio is not None and a.sched == "p3":
    pipe.tail_st = [torch.cuda.Stream(priority=a.tail_prio) for _ in pipe.tail_st]
    pipe.side = pipe.tail_st[0]
if a.tail_cus and a.sched == "p3":
    hip = ctypes.CDLL("libamdhip64.so")
    def masked(bits):
        words = [0] * 8
        for i in bits: words[i // 32] |= 1 << (i % 32)
        h = ctypes.c_void_p(); arr = (ctypes.c_uint32 * 8)(*words)
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, arr)
        assert rc == 0, rc
        return torch.cuda.ExternalStream(h.value)
    tb = list(range(a.tail_cus)) if a.mask_layout == "low" else list(range(256 - a.tail_cus, 256))
    cb = [i for i in range(256) if i not in tb]
    pipe.tail_st = [masked(tb) for _ in pipe.tail_st]
    pipe.side = pipe.tail_st[0]
    if a.acc_mask == "comp": pipe.acc = masked(cb)
    if a.rest_stream == "comp": pipe.rest_st = masked(cb)
    if a.sort_mask == "tail": st = masked(tb)
    elif a.sort_mask == "comp": st = masked(cb)
if a.rest_stream == "own" and a.sched == "p3": pipe.rest_st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(6): t = pipe.submit(b, d_sc, prepared=a.prepared)
    torch.cuda.synchronize()
    if a.gc_off:
        import gc; gc.disable()
    if a.ramp:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import sclk_mhz
        import glob, re
        def clk(name):
            for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_" + name)):
                try:
                    for line in open(f):
                        if "*" in line and not line.startswith("S"):
                            m = re.search(r"(\d+)\s*Mhz", line, re.I)
                            if m and int(m.group(1)) > 0: return m.group(1)
                except OSError:
                    pass
            return "?"
        for rnd in range(3):
            torch.cuda.synchronize(); time.sleep(0.3); out = []
            for burst in range(16):
                t0 = time.perf_counter()
                for _ in range(10): pipe.submit(b, d_sc, prepared=a.prepared)
                torch.cuda.synchronize(); out.append("%.2f@%s/m%s/f%s/soc%s" % ((time.perf_counter() - t0) * 1e3, sclk_mhz(), clk("mclk"), clk("fclk"), clk("socclk")))
            print("ms per burst of 10 @ sclk/mclk/fclk/socclk MHz:", " ".join(out), flush=True)
        sys.exit(0)
    if a.fit:
        if a.prof: ozk.check(L.ozk_prof_enable(a.prof))
        pts = []
        for k in (10, 20, 40, 100, 200) * 3:
            torch.cuda.synchronize()
            if a.idle_ms is not None:
                time.sleep(a.idle_ms / 1e3)
                for _ in range(5): pipe.submit(b, d_sc, prepared=a.prepared)
                torch.cuda.synchronize()
            t0 = time.perf_counter(); hs = []
            for _ in range(k):
                h0 = time.perf_counter(); t = pipe.submit(b, d_sc, prepared=a.prepared)
                if a.finish and t > 0:
                    with torch.cuda.stream(pipe.stream_of(t - 1)): pipe.result(t - 1)
                hs.append(time.perf_counter() - h0)
            t_issued = time.perf_counter() - t0
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            pts.append((k, dt * 1e3))
            print("K=%3d total %.2f ms (%.3f per MSM); host: issue of all %.2f ms, submit median %.0f us, max %.0f us, first %.0f us"
                  % (k, dt * 1e3, dt * 1e3 / k, t_issued * 1e3, sorted(hs)[k // 2] * 1e6, max(hs) * 1e6, hs[0] * 1e6), flush=True)
        A = np.array([[1.0, k] for k, _ in pts]); y = np.array([t for _, t in pts])
        c0, sl = np.linalg.lstsq(A, y, rcond=None)[0]
        print("fit: total = %.2f ms + K * %.3f ms" % (c0, sl)); sys.exit(0)
    if a.prof: ozk.check(L.ozk_prof_enable(a.prof))
    t0 = time.perf_counter()
    for _ in range(a.reps): t = pipe.submit(b, d_sc, prepared=a.prepared)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
msg = ("class tail_cus=%d " % a.class_tail_cus if a.class_tail_cus else "") + ("tail_cus=%d acc=%s sort=%s %s " % (a.tail_cus, a.acc_mask, a.sort_mask, a.mask_layout) if a.tail_cus else "") + "%s depth=%d ts=%d prof=%d: %.1f Mscalar-mul/s (%.3f ms per MSM)" % (a.sched, a.depth, a.tail_streams, a.prof, a.reps * n / dt / 1e6, dt / a.reps * 1e3)
if a.prof:
    s4 = (ctypes.c_double * 4)(); k = ctypes.c_int()
    ozk.check(L.ozk_prof_dominant_kernel_stats(s4, ctypes.byref(k)))
    msg += "  level-1 ms mean/med/min/max = %.3f/%.3f/%.3f/%.3f (%d)" % (s4[0], s4[1], s4[2], s4[3], k.value)
print(msg, flush=True)
