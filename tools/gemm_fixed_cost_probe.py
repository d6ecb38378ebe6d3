import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from haet_amd import ops
dev = torch.device("cuda", 0)
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - a) / n * 1e6
W = torch.randn(256, 256, device=dev) * 0.05
w = ops.SplitLinear(W, torch.zeros(256, device=dev))
sp = ops.SliceParams(torch.randn(64, 32, device=dev) * 0.3, torch.zeros(64, device=dev), torch.randn(32, device=dev) * 0.1, 0.05, 0.5, 1e-6)
for rows in (16384, 32768, 65536, 131072, 262144, 1048576):
    x = torch.randn(1, rows, 256, device=dev)
    a = t(lambda: ops.linear(x, w))
    b = t(lambda: ops.project_split(x, w, sp, 8, 32)) if ops.project_split_supported(256, 256, rows) else float("nan")
    print(f"rows={rows}: linear {a:.1f} us, project_split {b:.1f} us", flush=True)
