import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haet_amd import ops
dev = torch.device("cuda", 0)
N = 1_000_000
x = torch.randn(N, 256, device=dev)
W = torch.randn(256, 256, device=dev) / 16; b = torch.randn(256, device=dev)
W2 = torch.randn(512, 256, device=dev) / 16; b2 = torch.randn(512, device=dev)
lw, lb = torch.ones(256, device=dev), torch.zeros(256, device=dev)
w = ops.SplitLinear(W, b); w2 = ops.SplitLinear(W2, b2)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
print("linear 256->256        ", t(lambda: ops.linear(x, w)))
print("linear 256->256 + ln   ", t(lambda: ops.linear(x, w, ln=(lw, lb, 1e-5))))
print("linear 256->512 gelu   ", t(lambda: ops.linear(x, w2, gelu=True)))
print("linear 256->512 gelu+ln", t(lambda: ops.linear(x, w2, gelu=True, ln=(lw, lb, 1e-5))))
print("layernorm              ", t(lambda: ops.layernorm(x, lw, lb, 1e-5)))
