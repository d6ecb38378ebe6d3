import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haet_amd import ops
dev = torch.device("cuda", 0)
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
rows, K, O = 262144, 6912, 256
x = torch.randn(rows, K, device=dev)
W = torch.randn(O, K, device=dev) / K ** 0.5
w = ops.SplitLinear(W, torch.zeros(O, device=dev))
ms = t(lambda: ops.linear(x, w))
print(f"plain GEMM {rows} x {K} x {O}: {ms:.2f} ms = {3 * 2 * rows * K * O / ms / 1e9:.0f} TFLOP/s bf16")
del x
g = 64
xc = torch.randn(1, g ** 3, 256, device=dev)
conv = torch.nn.Conv3d(256, 256, 3, 1, 1).to(dev)
wc = ops.SplitConv(conv.weight, conv.bias)
ms = t(lambda: ops.conv_project(xc, wc, (g, g, g)))
print(f"conv {g}^3 x 256 -> 256: {ms:.2f} ms = {3 * 2 * g**3 * 6912 * 256 / ms / 1e9:.0f} TFLOP/s bf16")
