import time, torch
dev = torch.device("cuda", 0)
conv = torch.nn.Conv3d(256, 256, 3, 1, 1).to(dev)
x = torch.randn(1, 256, 96, 96, 96, device=dev)
y = conv(x)
g = torch.randn_like(y)
def t(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
print("MIOpen fwd        ", t(lambda: conv(x)))
print("MIOpen bwd weight ", t(lambda: torch.autograd.grad(conv(x), [conv.weight], g)))
xr = x.clone().requires_grad_(True)
print("MIOpen bwd data+fwd", t(lambda: torch.autograd.grad(conv(xr), [xr], g)))
