#!/usr/bin/env python3
"""Wall time of the JNI-shaped (`*_host`) entry points as a Java caller sees them: inputs in PAGEABLE host
memory that the runtime has never seen (a fresh buffer per call), result back in host memory.  Prints, per
entry point, the first call (cold: context creation, arena growth) and min / median of the following ones, next
to the bytes moved and the device-resident time of the same operation.  Output is committed under profiles/."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev  # noqa: E402
from octopuszk_amd import lib as ozk  # noqa: E402
from oracle import bn254 as o  # noqa: E402

# the Python collector's pauses land inside whichever call happens to be timed (10-40 ms, tools/host_jitter.py:
# the library itself never showed them); they are the harness's, not the entry point's
import gc  # noqa: E402
gc.disable()

L = ozk.load()


def to_device(a):
    """numpy -> device through PINNED memory.  torch.from_numpy(x).cuda() on a large pageable temporary makes the HIP
    runtime register that range with the kernel driver; once the temporary is freed and the range recycled by the
    fresh input buffers of later calls, every unmap of it evicts all GPU queues of the process for 10-30 ms — the
    alternating 6 / 25 ms calls this script used to show for the double MSM.  --pageable-uploads
    brings them back."""
    t = torch.from_numpy(np.ascontiguousarray(a))
    if "--pageable-uploads" in sys.argv:
        return t.cuda()
    return t.pin_memory().cuda()


def from_device(t):
    """device -> numpy through PINNED memory, for the same reason (round 4: t.cpu() into a large pageable temporary
    registers that range just as an upload does — the collection of a280974 showed the alternating 6 / 17-40 ms calls
    again for `--only=double --only=fixed_batch_msm_host`, whose set-up downloads the generated bases that way)"""
    if "--pageable-uploads" in sys.argv:
        return t.cpu().numpy()
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t)
    torch.cuda.synchronize()
    return np.array(h.numpy(), copy=True)


def vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def scalars(n, seed):
    b = np.random.default_rng(seed).integers(0, 256, size=(n, 32), dtype=np.uint8)
    b[:, 31] &= 0x1F
    return b


def fresh(a):
    """a new pageable copy (what GetPrimitiveArrayCritical hands over changes from call to call)"""
    return np.array(a, copy=True)


ONLY = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--only=")]
ALL_STATS = "--all-stats" in sys.argv
REPS = [int(a.split("=", 1)[1]) for a in sys.argv[1:] if a.startswith("--reps=")]   # --reps=N: N calls of every entry


def throttled_us():
    """microseconds this container's cgroup has spent THROTTLED by its CPU quota so far (cgroup v2 cpu.stat
    throttled_usec, v1 throttled_time in ns), or None"""
    for path, key, div in (("/sys/fs/cgroup/cpu.stat", "throttled_usec", 1), ("/sys/fs/cgroup/cpu/cpu.stat", "throttled_time", 1000),
                           ("/sys/fs/cgroup/cpu,cpuacct/cpu.stat", "throttled_time", 1000)):
        try:
            for line in open(path):
                k, v = line.split()
                if k == key:
                    return int(v) // div
        except (OSError, ValueError):
            pass
    return None


def run(name, fn, make_inputs, reps, moved_mib, dev_ms=None):
    if ONLY and not any(o_ in name for o_ in ONLY):
        return
    import resource
    times, stats, faults = [], [], []
    if REPS:
        reps = REPS[0]
    for _ in range(reps + 1):
        args = make_inputs()
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        th0 = throttled_us()
        t0 = time.perf_counter()
        fn(*args)
        times.append((time.perf_counter() - t0) * 1e3)
        th1 = throttled_us()
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        faults.append((ru1.ru_minflt - ru0.ru_minflt, ru1.ru_nvcsw - ru0.ru_nvcsw, ru1.ru_nivcsw - ru0.ru_nivcsw,
                       -1.0 if th0 is None or th1 is None else (th1 - th0) / 1e3,
                       (ru1.ru_utime + ru1.ru_stime - ru0.ru_utime - ru0.ru_stime) * 1e3))
        st = (ctypes.c_double * 10)()
        L.ozk_host_call_stats(st)
        stats.append(list(st))
    if ALL_STATS:   # every call: wall, the library's own account of its waits, page faults / context switches of the process
        f = ("acquire", "reserve", "stage_wait", "memcpy_in", "memcpy_out", "enqueue", "sync")
        for i, (t, st_, fl) in enumerate(zip(times, stats, faults)):
            print("    call %d: %6.2f ms (library %6.2f)  " % (i, t, st_[9]) + "  ".join("%s %.2f" % (n_, v) for n_, v in zip(f, st_))
                  + "  | minor faults %d  ctx switches %d + %d  cgroup throttled %.1f ms  process CPU %.1f ms" % fl, flush=True)
    warm = sorted(times[1:])
    extra = "" if dev_ms is None else "  | device-resident %.2f ms" % dev_ms
    print("%-46s first %8.2f ms | min %7.2f  median %7.2f ms | %5.0f MiB over PCIe%s   all: %s"
          % (name, times[0], warm[0], warm[len(warm) // 2], moved_mib, extra, " ".join("%.1f" % t for t in times[1:])), flush=True)
    k = max(range(1, len(times)), key=lambda i: times[i])
    if times[k] > 1.5 * warm[0]:   # where the slowest call's time went, as the library saw it
        f = ("acquire", "reserve", "stage_wait", "memcpy_in", "memcpy_out", "enqueue", "sync")
        print("    slowest call %.1f ms, inside the library %.1f: " % (times[k], stats[k][9])
              + "  ".join("%s %.2f" % (n, v) for n, v in zip(f, stats[k])), flush=True)


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    logn = int(pos[0]) if pos else 20
    n = 1 << logn
    print("device:", torch.cuda.get_device_name(0), " n = 2^%d" % logn, flush=True)
    g1 = from_device(dev.gen_g1_bases(n, seed=2))
    sc = scalars(n, 1)
    out = np.zeros(576, dtype=np.uint8)
    # device-resident reference
    d_b, d_s = to_device(g1.copy()), to_device(sc.reshape(-1))
    ws = dev.VarMsmWorkspace(n, 1)
    ws.run(d_b, d_s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ws.run(d_b, d_s)
    torch.cuda.synchronize()
    dev_ms = (time.perf_counter() - t0) / 5 * 1e3
    want = bytes(ws.out.cpu().numpy())

    def var_g1(b, s):
        ozk.check(L.ozk_var_msm_host(vp(b), vp(s), n, 1, 0, vp(out)))
        assert bytes(out[:192]) == want

    run("ozk_var_msm_host G1", var_g1, lambda: (fresh(g1), fresh(sc)), 6, n * 128 / 2**20, dev_ms)
    # prepared bases: scalars only
    h = ctypes.c_void_p()
    ozk.check(L.ozk_bases_create_host(vp(g1), n, 1, 0, ctypes.byref(h)))

    def var_prep(s):
        ozk.check(L.ozk_var_msm_bases_host(h, vp(s), n, vp(out)))
        assert bytes(out[:192]) == want

    run("ozk_var_msm_bases_host G1 (prepared)", var_prep, lambda: (fresh(sc),), 6, n * 32 / 2**20, dev_ms)
    ozk.check(L.ozk_bases_destroy(h))
    # double MSM at n / 4 (the Java chunk is 2^21; G2 bases 192 B each)
    m = n // 4
    ks = scalars(m, 9)
    ks[:, 8:] = 0
    st = int(torch.cuda.current_stream().cuda_stream)
    base2 = to_device(np.frombuffer(o.g2_to_wire(o.G2.one), dtype=np.uint8).copy())
    g2d = torch.empty(m * 192, dtype=torch.uint8, device="cuda")
    wsb = int(L.ozk_fixed_batch_msm_workspace_bytes(16, 16, m, 2))
    wsf = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    d_ks = to_device(ks.reshape(-1))
    ozk.check(L.ozk_fixed_batch_msm_compact_dev(16, 16, m, int(base2.data_ptr()), int(d_ks.data_ptr()),
                                                2, int(g2d.data_ptr()), int(wsf.data_ptr()), wsb, st))
    torch.cuda.synchronize()
    g2 = from_device(g2d)
    del wsf, g2d

    def dbl(b1, b2, s):
        ozk.check(L.ozk_var_double_msm_host(vp(b1), vp(b2), vp(s), m, 0, vp(out)))

    run("ozk_var_double_msm_host n/4", dbl, lambda: (fresh(g1[:m * 96]), fresh(g2), fresh(sc[:m])), 4, m * 320 / 2**20)
    # fixed base G1 / G2, window 17
    for bn, name in ((1, "G1"), (2, "G2")):
        per = 192 if bn == 1 else 384
        C = o.G1 if bn == 1 else o.G2
        bw = np.frombuffer(o.g1_to_wire(C.one) if bn == 1 else o.g2_to_wire(C.one), dtype=np.uint8)
        fout = np.zeros(n * per, dtype=np.uint8)

        def fb(s):
            ozk.check(L.ozk_fixed_batch_msm_host(15, 17, 15, 1 << 17, n, 254, vp(bw), vp(s), bn, 0, vp(fout)))

        run("ozk_fixed_batch_msm_host %s w=17" % name, fb, lambda: (fresh(sc),), 4, n * (32 + per) / 2**20)
        cout = np.zeros(n * per // 2, dtype=np.uint8)

        def fbc(s):
            ozk.check(L.ozk_fixed_batch_msm_compact_host(15, 17, n, vp(bw), vp(s), bn, 0, vp(cout)))

        run("ozk_fixed_batch_msm_compact_host %s" % name, fbc, lambda: (fresh(sc),), 4, n * (32 + per // 2) / 2**20)
        del fout, cout
    # FFT 4n (2^22 at the default), reference format and compact
    nf = 4 * n
    a = scalars(nf, 3)
    om = np.frombuffer(o.to_le32(o.fr_root_of_unity(nf)), dtype=np.uint8)
    f64, f32 = np.zeros(nf * 64, dtype=np.uint8), np.zeros(nf * 32, dtype=np.uint8)
    run("ozk_fft_host n=2^%d" % (logn + 2), lambda x: ozk.check(L.ozk_fft_host(vp(x), nf, vp(om), 0, vp(f64))),
        lambda: (fresh(a),), 4, nf * 96 / 2**20)
    run("ozk_fft_compact_host n=2^%d" % (logn + 2), lambda x: ozk.check(L.ozk_fft_compact_host(vp(x), nf, vp(om), 0, vp(f32))),
        lambda: (fresh(a),), 4, nf * 64 / 2**20)
    # witness map at 2n
    mq = 2 * n
    ev = [scalars(mq, 20 + k) for k in range(3)]
    gq = np.frombuffer(o.to_le32(o.FR_MULT_GEN), dtype=np.uint8)
    omq = np.frombuffer(o.to_le32(o.fr_root_of_unity(mq)), dtype=np.uint8)
    hq = np.zeros((mq + 1) * 32, dtype=np.uint8)
    run("ozk_qap_witness_host m=2^%d" % (logn + 1),
        lambda x, y, z: ozk.check(L.ozk_qap_witness_host(vp(x), vp(y), vp(z), mq, vp(omq), vp(gq), 0, vp(hq))),
        lambda: tuple(fresh(e) for e in ev), 4, mq * 128 / 2**20)


if __name__ == "__main__":
    main()
