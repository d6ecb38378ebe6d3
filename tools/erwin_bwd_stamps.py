#!/usr/bin/env python3
"""Diagnostic: cycle stamps of haet_erwin_bwd (stamped build: make -C haet_amd/csrc clean all EXTRA=-DHAET_STAMPS on the GPU
box).  Phases of the first block the kernel processes (the decoder's last block) on cloud 0, and per-op totals."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haet_amd import _lib
from haet_amd.PhysicsAttention import Physics_Attention_Irregular_Mesh

dev = torch.device("cuda:0")
torch.manual_seed(0)
mod = Physics_Attention_Irregular_Mesh(256, heads=8, dim_head=32, slice_num=64, rotate=45).to(dev).train()
x = torch.randn(1, 200_000, 256, device=dev, requires_grad=True)
for _ in range(3):
    mod.zero_grad(set_to_none=True)
    mod(x).sum().backward()
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 64)()
assert lib.haet_debug_read_bwd_stamps(buf) == 0
t = list(buf)
names = {1: "tape + rel_pos + norm1", 2: "qkv gemm", 3: "attention fwd", 4: "proj + x_mid + norm2", 5: "SwiGLU backward (all chunks)",
         6: "rms_bwd 2", 7: "proj wgrad + d ao", 8: "attention bwd", 9: "qkv wgrad + dn1 + pe", 10: "rms_bwd 1"}
for i in range(1, 11):
    print(f"  {names[i]:32s} {t[i] - t[i - 1]:9d} cycles")
print("first SwiGLU chunk: h1/h2 gemms, g, dW3, dg gemm, dh, dW1+dW2, dn2 gemms:", [t[i + 1] - t[i] for i in range(20, 27)])
ops = [t[48 - k] for k in range(8)]
print("per op (reverse order: dec blocks, unpool, bottleneck blocks, pool, enc blocks):", [ops[k + 1] - ops[k] for k in range(7)], "block total", t[10] - t[0])

print("inside eb_phase (phase A of the first chunk, thread 0): entry->barrier+job", t[53] - t[52], "uniform copies", t[54] - t[53],
      "decode + weights requested", t[59] - t[54], "bias / accumulator init (waits for the loads)", t[55] - t[59], "first group requested", t[56] - t[55],
      "k-loop done", t[60] - t[56], "stores + loop end", t[57] - t[60], "db + exit", t[58] - t[57])
if os.environ.get("STAMP_PHASE_C"):
    print("phase C (dW3 tile on wave 0 | dg tiles): wgrad tile: gather + MFMA loop", t[62] - t[61], "; from loop end to the phase's tile loop end (stores)", t[57] - t[62],
          "; entry -> first tile", t[61] - t[52])
