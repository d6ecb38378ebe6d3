import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import torch.nn.functional as F
from haet_amd import ops
dev = torch.device("cuda", 0)
H, M = 8, 64
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - a) / n * 1e6
for B, N in [(4, 11271), (1, 45084), (1, 200000), (1, 1000000)]:
    torch.manual_seed(0)
    res = {}
    for C in (16, 32):
        x = torch.randn(B, N, H * 16, device=dev)
        Ws, bs, wt = torch.randn(M, 16, device=dev) * 0.3, torch.randn(M, device=dev) * 0.1, torch.randn(16, device=dev) * 0.1
        if C == 32:
            x, Ws, wt = ops.pad_heads(x, H, 16, 32), F.pad(Ws, (0, 16)), F.pad(wt, (0, 16))
        sp = ops.SliceParams(Ws, bs, wt, 0.05, 0.5, 1e-6)
        parts = ops.slice_partials(x, sp, H, C, M)
        cm, cs, S = ops.merge_records(parts, B, H, C, M, True)
        S2 = S * 1.1
        g = torch.randn_like(x)
        dS = torch.randn_like(S); A = torch.randn(B * H * M, device=dev)
        res[C] = (t(lambda: ops.slice_partials(x, sp, H, C, M)), t(lambda: ops.deslice(x, sp, cm, cs, S2, H, C, M)),
                  t(lambda: ops.deslice_bwd_tokens(x, sp, cm, cs, g, H, C, M)), t(lambda: ops.slice_bwd_points(x, sp, cm, cs, g, S2, dS, A, H, C, M)))
    print(f"B={B} N={N}: us (slice, deslice, bwd_tokens, bwd_points)  C=16 VALU {[round(v) for v in res[16]]}   C=32 padded {[round(v) for v in res[32]]}", flush=True)
