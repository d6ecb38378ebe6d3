import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haet_amd.PhysicsAttention import Physics_Attention_Irregular_Mesh
dev = torch.device("cuda", 0)
import gc; gc.collect(); gc.freeze()  # (a full collection of the import-time heap is a ~75 ms stall: tools/stall_probe.py)
for (N, dim, H, C, M) in [(100_000, 256, 4, 64, 32), (100_000, 256, 8, 32, 32), (1_000_000, 256, 4, 64, 32)]:
    torch.manual_seed(0)
    mod = Physics_Attention_Irregular_Mesh(dim=dim, heads=H, dim_head=C, slice_num=M, rotate=45).eval().to(dev)
    x = torch.randn(1, N, dim, device=dev)
    with torch.no_grad():
        for _ in range(3): mod(x)
        mod.stage_events = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): mod(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    st = {}
    for n, a, b in mod.stage_events: st[n] = st.get(n, 0) + a.elapsed_time(b) / 10
    print(f"N={N} H={H} C={C} M={M}: {1e3*dt:.3f} ms  " + ", ".join(f"{k} {v:.3f}" for k, v in st.items()), flush=True)
