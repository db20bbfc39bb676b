#!/usr/bin/env python3
"""How long the first operation after T ms of idle device takes: a pinned 8 MiB host-to-device copy (DMA engine) and a
small kernel, each synchronised.  Ten samples per T, median / max."""
import time, torch
h = torch.empty(8 << 20, dtype=torch.uint8).pin_memory()
d = torch.empty(8 << 20, dtype=torch.uint8, device="cuda")
x = torch.zeros(1 << 16, device="cuda")
s = torch.cuda.Stream()
def copy():
    with torch.cuda.stream(s):
        d.copy_(h, non_blocking=True)
    s.synchronize()
def kern():
    with torch.cuda.stream(s):
        x.add_(1.0)
    s.synchronize()
for name, op in (("pinned 8 MiB H2D copy", copy), ("small kernel", kern)):
    for _ in range(5): op()
    for T in (0, 1, 3, 5, 8, 10, 12, 15, 20, 30, 50, 100, 300):
        ts = []
        for _ in range(10):
            time.sleep(T / 1e3)
            t0 = time.perf_counter(); op(); ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        print("%-22s after %3d ms idle: median %.2f ms  max %.2f ms   all: %s" % (name, T, ts[5], ts[-1], " ".join("%.1f" % t for t in ts)), flush=True)
