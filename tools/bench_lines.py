#!/usr/bin/env python3
"""Print ms/step and stage times of bench JSON lines: tools/bench_lines.py gpurun_out/b*.json"""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as exc:  # noqa: BLE001
        print(f, "unreadable:", exc)
        continue
    r = d.get("roofline", {})
    print(f"{f}: {d['ms_per_step']:.4f} ms  {d['value']:.3e} pts/s  path_frac={r.get('path_frac', 0):.3f}  "
          + " ".join(f"{k}={v:.3f}" for k, v in d["stage_ms"].items()))
