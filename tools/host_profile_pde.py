#!/usr/bin/env python3
"""Host-side profile of one PDE-benchmark configuration's training step (tools/pde_configs_probe.py, PDE_ONLY): cProfile around
the attention module's forward AND around its autograd backward -- which runs on autograd's device thread, where a profiler
enabled on the main thread sees nothing.  Sorted by self time; the step is host-bound at these sizes."""
import cProfile
import io
import os
import pstats
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PDE_ONLY", "airfoil M=64")
import torch

from haet_amd import _autograd

prof_b, prof_f = cProfile.Profile(), cProfile.Profile()
F = _autograd.PhysicsAttentionFunction
orig_b, orig_f = F.backward, F.forward


def backward(ctx, *a):
    prof_b.enable()
    try:
        return orig_b(ctx, *a)
    finally:
        prof_b.disable()


def forward(ctx, *a):
    prof_f.enable()
    try:
        return orig_f(ctx, *a)
    finally:
        prof_f.disable()


F.backward, F.forward = staticmethod(backward), staticmethod(forward)
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pde_configs_probe.py"), run_name="__main__")
for name, pr in (("attention forward (train mode)", prof_f), ("attention backward", prof_b)):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(38)
    print("=" * 30, name)
    print("\n".join(l[:170] for l in s.getvalue().splitlines() if l.strip())[:9000])
