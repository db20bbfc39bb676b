"""Aggregate throughput of T host threads, each issuing complete 2^20 G1 MSMs (ozk_var_msm_dev) on its own
stream — the shape of the reference's concurrent Spark task threads.  mt_throughput.py <threads> [<reps>]"""
import os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev
T = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n = 1 << 20
bases = dev.gen_g1_bases(n, seed=2)
scs = []
for t in range(T):
    sc = np.random.default_rng(10 + t).integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
    scs.append(torch.from_numpy(sc.reshape(-1)).cuda())
wss = [dev.VarMsmWorkspace(n, 1) for _ in range(T)]
streams = [torch.cuda.Stream() for _ in range(T)]
def work(t, k):
    with torch.cuda.stream(streams[t]):
        for _ in range(k):
            wss[t].run(bases, scs[t])
for t in range(T): work(t, 2)
torch.cuda.synchronize()
ths = [threading.Thread(target=work, args=(t, reps)) for t in range(T)]
t0 = time.perf_counter()
for th in ths: th.start()
for th in ths: th.join()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("threads %d: %.1f Mscalar-mul/s aggregate (%.3f ms per MSM per thread)" % (T, T * reps * n / dt / 1e6, dt / reps * 1e3), flush=True)
