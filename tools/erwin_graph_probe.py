#!/usr/bin/env python3
"""Which part of the Erwin torch restatement survives HIP-graph capture (diagnostic for _autograd._erwin_stage)."""
import faulthandler
import os
import sys

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HAET_ERWIN_GRAPH"] = "1"
import torch

from bench import build_module
from haet_amd import _autograd, ops

which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
dev = torch.device("cuda", 0)
mod = build_module(dev)
mod.train(which.endswith("train"))
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = torch.randn(NC * 64, 32, device=dev)
w = _autograd._ErwinTokens(mod, NC)
if len(sys.argv) > 3:  # eager forward + backward of the whole module first
    xx = torch.randn(NC // 8, 3000, 256, device=dev, requires_grad=True)
    if "hip" in sys.argv[3]:
        os.environ.pop("HAET_ERWIN_GRAPH", None)
        mod(xx).sum().backward()
    else:
        _autograd.attention_forward(mod, xx).sum().backward()
    torch.cuda.synchronize()
    print("eager pre-run ok", flush=True)
w.training = mod.erwin.training
print("mode", which, flush=True)
if which.startswith("tree"):
    pos = torch.rand(512, 3, device=dev)
    ops.balltree_build(pos, 8, 64, mod.erwin.strides, mod.erwin.ball_sizes, 45.0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = ops.balltree_build(pos, 8, 64, mod.erwin.strides, mod.erwin.ball_sizes, 45.0)
        idx = torch.argsort(out[0].long())
    g.replay()
    torch.cuda.synchronize()
    print("tree ok", idx[:4].tolist(), flush=True)
elif which.startswith("fwd"):
    with torch.no_grad():
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                y = w(S)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        print("warm-up ok", flush=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = w(S)
        print("capture ok", flush=True)
        g.replay()
        torch.cuda.synchronize()
        print("fwd ok", float(y.abs().sum()), flush=True)
else:
    sample = torch.randn_like(S).requires_grad_(True)
    gw = torch.cuda.make_graphed_callables(w, (sample,), allow_unused_input=True)
    print("graphed", flush=True)
    x = S.clone().requires_grad_(True)
    y = gw(x)
    y.sum().backward()
    torch.cuda.synchronize()
    print("fwd+bwd ok", float(x.grad.abs().sum()), flush=True)
