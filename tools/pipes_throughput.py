"""P ordered two-stream pipelines (VarMsmPipeline), MSMs issued round-robin.  pipes_throughput.py <P> [<reps>]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import device as dev
P = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
n = 1 << 20
bases = dev.gen_g1_bases(n, seed=2)
sc = np.random.default_rng(10).integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
d_sc = torch.from_numpy(sc.reshape(-1)).cuda()
pipes = [dev.VarMsmPipeline(n, 1, depth=2) for _ in range(P)]
mains = [torch.cuda.Stream() for _ in range(P)]
def run(k):
    for i in range(k):
        with torch.cuda.stream(mains[i % P]):
            pipes[i % P].submit(bases, d_sc)
run(4 * P); torch.cuda.synchronize()
t0 = time.perf_counter(); run(reps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("pipelines %d: %.1f Mscalar-mul/s (%.3f ms per MSM)" % (P, reps * n / dt / 1e6, dt / reps * 1e3), flush=True)
