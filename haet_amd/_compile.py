"""torch.compile integration (the reference's scripts wrap the model in
`torch.compile(model, mode="max-autotune" | "reduce-overhead")`, exp_airfoil.py:137, exp_plas.py:177).

The HIP path is driven by ctypes calls that Dynamo cannot trace.  Instead of leaving it `torch.compiler.disable`d
-- a graph break around every attention layer and every point-wise fast-path layer, which also keeps
"reduce-overhead" from capturing the forward in one HIP graph -- each nn.Module of this package whose forward is
such a path is exposed to the compiler as ONE opaque operator, `haet::module_forward(x, handle, out_features)`,
registered with `torch.library.custom_op` together with a fake (meta) implementation.  Under `torch.compile` with
gradients disabled the whole model then traces into a single graph whose only non-ATen nodes are these operators, and
the kernels they launch on the current stream are captured like any other (the reference needs
`@torch.compiler.disable` for its host-side tree builder, models/components/balltree/balltree.pyx:598; here the tree is
built on the device inside the operator).

The operator takes the module by an integer handle: parameters stay inside the module (packed / split operands are
cached there), the compiler sees a function of the activations only.  With gradients enabled the modules keep their
eager autograd path (a graph break, as before).  NOTE (inherent to HIP-graph replay): a captured graph replays the
launches recorded at capture time, so weight-derived operands re-packed by Python code (the Erwin blob, bf16 planes)
are NOT refreshed by a replay -- re-capture after changing weights in a "reduce-overhead" deployment.
"""
from __future__ import annotations

import itertools
import weakref

import torch

_registry = weakref.WeakValueDictionary()
_ids = itertools.count(1)


def register(module) -> int:
    """Give `module` a handle for haet::module_forward."""
    h = next(_ids)
    _registry[h] = module
    module.__dict__["_haet_handle"] = h
    return h


class OpaqueForwardMixin:
    """For nn.Modules exposed through haet::module_forward: the integer handle lives in the instance __dict__ (Dynamo
    reads it as a guarded constant); a copy.deepcopy / unpickled module re-registers itself so that it never dispatches
    to the module it was copied from."""

    def __setstate__(self, state):
        super().__setstate__(state)
        register(self)
        self._after_copy()

    def _after_copy(self):
        pass


@torch.compiler.assume_constant_result
def primed_handle(handle: int) -> int:
    """Called by the traced forwards around their handle: Dynamo evaluates it for real at TRACE time, i.e. before
    Inductor's HIP-graph warm-up and capture.  It runs the module once on a small dummy input so that every
    weight-derived operand the module caches (bf16 planes, the packed Erwin blob) is allocated now, in ordinary
    memory -- allocated lazily inside the first compiled call they would live in the graph's private pool and
    outlive it ("live storage not accounted for as an output of cudagraph trees")."""
    mod = _registry.get(handle)
    if mod is not None:
        try:
            with torch.no_grad():
                mod._prime()
        except Exception:  # noqa: BLE001 -- priming is an optimisation of the first call, never a failure of compile
            pass
    return handle


@torch.library.custom_op("haet::module_forward", mutates_args=(), device_types="cuda")
def module_forward(x: torch.Tensor, handle: int, out_features: int) -> torch.Tensor:
    mod = _registry.get(handle)
    if mod is None:
        raise RuntimeError(f"haet::module_forward: module handle {handle} is gone (the module was deleted)")
    with torch.no_grad():
        y = mod._opaque_forward(x)
    return y.to(x.dtype).contiguous()


@module_forward.register_fake
def _module_forward_fake(x, handle, out_features):
    return x.new_empty(x.shape[:-1] + (out_features,))


def use_operator(x: torch.Tensor) -> bool:
    """Inside a Dynamo trace, no gradient wanted, device tensor: route through the opaque operator."""
    return torch.compiler.is_compiling() and not torch.is_grad_enabled() and x.is_cuda
