"""Forward-time / peak-memory sweep of HAETransolver_Irregular_Mesh over the point count -- the measurement the
reference ships as benchmarks/00-VelocityMemMocker/velocity-mem-checker.py (model configuration :34-44, protocol
:46-76, sweep :82, OOM handling :102-112), on this package's model.  `python bench.py --sweep` prints it."""
from __future__ import annotations

import time
from typing import Dict, Iterable, List, Optional

import torch

SWEEP_POINTS = (1_000, 10_000, 100_000, 1_000_000, 2_000_000, 3_000_000)  # velocity-mem-checker.py:82


def build_model(space_dim=3, fun_dim=1, n_hidden=256, device="cuda"):
    """The sweep's model (velocity-mem-checker.py:34-44): one layer, 8 heads, mlp_ratio 2, 4 outputs, 32 slices."""
    from .. import HAETransolver_Irregular_Mesh

    model = HAETransolver_Irregular_Mesh(space_dim=space_dim, fun_dim=fun_dim, n_hidden=n_hidden, n_layers=1, n_head=8,
                                         mlp_ratio=2, out_dim=4, slice_num=32, unified_pos=0)
    return model.to(device).eval()


def benchmark_model(num_points: int, space_dim=3, fun_dim=1, n_hidden=256, autocast=True, warmup=3, timed=1,
                    model=None, device="cuda") -> Dict[str, float]:
    """One size of the sweep: `warmup` untimed forwards, `timed` forwards between device synchronisations (the
    reference times ONE; more average out launch jitter), then one forward under a reset peak-memory counter.
    Inputs are uniform random coordinates / function values, batch 1 (velocity-mem-checker.py:30-31)."""
    x = torch.rand(1, num_points, space_dim, device=device)
    fx = torch.rand(1, num_points, fun_dim, device=device)
    model = build_model(space_dim, fun_dim, n_hidden, device) if model is None else model

    def fwd():
        with torch.no_grad(), torch.autocast(device_type="cuda", enabled=autocast):
            return model(x, fx)

    for _ in range(warmup):
        fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(timed):
        out = fwd()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / timed
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    out = fwd()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated()
    return {"points": int(num_points), "seconds": dt, "points_per_s": num_points / dt,
            "peak_memory_mb": peak / 1024 ** 2, "resident_before_mb": base / 1024 ** 2,
            "finite": bool(torch.isfinite(out.float()).all()), "autocast": bool(autocast)}


def sweep(points: Optional[Iterable[int]] = None, autocast=True, timed=5, log=print, **kw) -> List[Dict[str, float]]:
    """The whole sweep; an out-of-memory size ends it (as velocity-mem-checker.py:102-107 does) and is reported."""
    rows = []
    for n in (SWEEP_POINTS if points is None else points):
        torch.cuda.empty_cache()
        try:
            r = benchmark_model(int(n), autocast=autocast, timed=timed, **kw)
        except RuntimeError as exc:  # torch.cuda.OutOfMemoryError is a RuntimeError
            if "out of memory" in str(exc).lower():
                rows.append({"points": int(n), "error": "out of memory"})
                log(f"ran out of memory with {n} points")
                break
            raise
        rows.append(r)
        log(f"{n:>9d} points: {1e3 * r['seconds']:9.3f} ms  {r['points_per_s']:.3e} points/s  "
            f"peak {r['peak_memory_mb']:9.1f} MB")
    return rows
