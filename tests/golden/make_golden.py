#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference and `make -C oracle ref`);
the fixtures it writes (*.npz: inputs, weights, expected outputs) are committed,
this script is the record of how they were made.  Nothing here ships.

How the reference is imported:
  * models/components/balltree is the compiled Cython extension built by
    oracle/Makefile (`ref` target) from the sources where they lie.
  * third-party packages the image lacks are replaced by in-process stand-ins
    *for import only*:  timm.models.layers.trunc_normal_ -> torch.nn.init's;
    torch_cluster / torch_scatter -> stubs that raise if called (unreachable:
    embed=False, mp_steps=0);  flash_attn.flash_attn_qkvpacked_func -> exact
    softmax(QK^T*scale)V (flash-attention's published contract).
  * precision: BallMSA.forward casts its weights with `.to(torch.float16)`
    (attention.py:43-56), which cannot run on fp32 CPU inputs.  The attention
    module's global name `torch` is replaced by a proxy whose `float16` IS
    `torch.float32`, so the reference's own forward statements run unmodified,
    in fp32 ("oracle mode A" of SURVEY.md 8c).
"""
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("HAET_REFERENCE", "/root/reference")


def _install_import_stand_ins():
    timm = types.ModuleType("timm")
    timm_models = types.ModuleType("timm.models")
    timm_layers = types.ModuleType("timm.models.layers")
    timm_layers.trunc_normal_ = torch.nn.init.trunc_normal_
    timm.models, timm_models.layers = timm_models, timm_layers
    sys.modules.update({"timm": timm, "timm.models": timm_models, "timm.models.layers": timm_layers})

    fa = types.ModuleType("flash_attn")

    def flash_attn_qkvpacked_func(qkv, dropout_p=0.0, softmax_scale=None, causal=False):
        assert dropout_p == 0.0 and not causal
        q, k, v = (t.permute(0, 2, 1, 3) for t in qkv.unbind(2))  # [nb, h, n, hd]
        return F.scaled_dot_product_attention(q, k, v, scale=softmax_scale).permute(0, 2, 1, 3)

    fa.flash_attn_qkvpacked_func = flash_attn_qkvpacked_func
    sys.modules["flash_attn"] = fa

    # torch_scatter / torch_cluster (unpinned in the reference's requirements.txt files) are only reached by the
    # stand-alone ErwinFlash fixtures (embed=True, mp_steps>0: SURVEY.md 8f-3).  Stand-ins restate their published
    # contracts:  scatter_mean(src, index, dim=0, dim_size) = per-index mean of the rows, zero rows where no index
    # points;  radius_graph(x, r, batch, loop, max_num_neighbors=32, flow='source_to_target') = torch_cluster's
    # brute-force device kernel: for every centre i (ascending) the first 32 points j (ascending, same cloud)
    # with |x_j - x_i|^2 < r^2, returned as [[j...], [i...]].
    def scatter_mean(src, index, dim=0, out=None, dim_size=None):
        assert dim == 0 and out is None
        n = int(dim_size) if dim_size is not None else int(index.max()) + 1
        acc = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype).index_add_(0, index, src)
        cnt = torch.zeros(n, dtype=src.dtype).index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
        return acc / cnt.clamp(min=1).view(-1, *([1] * (src.dim() - 1)))

    def radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32, flow="source_to_target", num_workers=1):
        assert flow == "source_to_target"
        n = x.shape[0]
        batch = torch.zeros(n, dtype=torch.long) if batch is None else batch
        d2 = ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)  # fp32, like the device kernel
        ok = (d2 < r * r) & (batch[:, None] == batch[None, :])
        if not loop:
            ok &= ~torch.eye(n, dtype=torch.bool)
        ok &= ok.long().cumsum(dim=1) <= max_num_neighbors  # first 32 neighbours by index
        i, j = ok.nonzero(as_tuple=True)  # row-major: centre i ascending, neighbour j ascending
        return torch.stack([j, i], dim=0)

    tc = types.ModuleType("torch_cluster")
    tc.radius_graph = radius_graph
    ts = types.ModuleType("torch_scatter")
    ts.scatter_mean = scatter_mean
    sys.modules.update({"torch_cluster": tc, "torch_scatter": ts})


def import_reference():
    _install_import_stand_ins()
    sys.path[:0] = [os.path.join(ROOT, "oracle", "_ref"), REF]
    import models  # noqa: the reference package
    from models.components.erwinflash.components import attention as ref_attention

    class _TorchFp32Proxy:
        float16 = torch.float32

        def __getattr__(self, name):
            return getattr(torch, name)

    ref_attention.torch = _TorchFp32Proxy()
    return models


# --------------------------------------------------------------------------- capture helpers
def capture_tree(models):
    """Wrap build_balltree_with_rotations where ErwinTransformer looks it up and keep a
    copy of what it returned (the caller pops the rot list empty, erwin_flash.py:184,187)."""
    from models.components.erwinflash import erwin_flash

    store = {}
    orig = erwin_flash.build_balltree_with_rotations

    def wrapped(*a, **k):
        ti, tm, rots = orig(*a, **k)
        store["tree_idx"], store["tree_mask"] = ti.clone(), tm.clone()
        for i, r in enumerate(rots):
            if r is not None:
                store[f"rot{i}"] = r.clone()
        return ti, tm, rots

    erwin_flash.build_balltree_with_rotations = wrapped
    return store


def run_attention(mod, x, tree_store):
    cap = {}
    hooks = []

    def hk(name, idx=None):
        def f(m, inp, out):
            cap[name] = out.detach().clone()
        return f

    hooks.append(mod.in_project_x.register_forward_hook(hk("x_lin_raw")))
    hooks.append(mod.erwin.register_forward_hook(hk("processed_flat")))
    hooks.append(mod.erwin.register_forward_pre_hook(
        lambda m, inp: cap.update(eidetic_flat=inp[0].detach().clone(), pos=inp[1].detach().clone(),
                                  batch_idx=inp[2].detach().clone())))
    n_enc = len(mod.erwin.encoder)
    for i, layer in enumerate(mod.erwin.encoder):
        for j, blk in enumerate(layer.blocks):
            hooks.append(blk.register_forward_hook(hk(f"erwin.enc{i}.block{j}")))
        hooks.append(layer.pool.register_forward_hook(
            lambda m, inp, out, i=i: cap.update({f"erwin.enc{i}.pool": out.x.detach().clone()})))
    for j, blk in enumerate(mod.erwin.bottleneck.blocks):
        hooks.append(blk.register_forward_hook(hk(f"erwin.bott.block{j}")))
    for k, layer in enumerate(mod.erwin.decoder):
        hooks.append(layer.unpool.register_forward_hook(
            lambda m, inp, out, k=k: cap.update({f"erwin.dec{k}.unpool": out.x.detach().clone()})))
        for j, blk in enumerate(layer.blocks):
            hooks.append(blk.register_forward_hook(hk(f"erwin.dec{k}.block{j}")))
    tree_store.clear()
    out = mod(x)
    for h in hooks:
        h.remove()
    cap["out"] = out.detach().clone()
    cap.update({"erwin." + k: v for k, v in tree_store.items()})
    # rotated blocks are hooked in ROTATED order (layers.py:86-88): un-permute to tree order
    for key in list(cap):
        if key.endswith("block1"):
            lvl = key.split(".")[1]
            rk = "erwin.rot1" if lvl == "bott" else "erwin.rot0"
            if rk in cap:
                cap[key] = cap[key][torch.argsort(cap[rk])]
    return cap


def amplify(mod, gen, scale=10.0):
    """Make every Erwin op numerically visible: scale the Erwin
    matrices, give every bias / norm weight / BN statistic a non-trivial value."""
    with torch.no_grad():
        for name, p in mod.named_parameters():
            if name.startswith("erwin.") and p.dim() == 2:
                p.mul_(scale)
            elif name.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=gen))
            elif name.startswith("erwin.") and "norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=gen))
        for name, b in mod.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=gen))
            elif name.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=gen))


def model_init_rule(mod):
    """Model._init_weights (HAETransolver_Irregular_Mesh.py:279-295) applied to a bare module."""
    import torch.nn as nn

    for m in mod.modules():
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, (nn.LayerNorm, nn.BatchNorm1d)):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)


def to_np(d):
    out = {}
    for k, v in d.items():
        v = v.detach().cpu() if torch.is_tensor(v) else torch.as_tensor(v)
        out[k] = v.numpy()
    return out


ATTN_CASES = [
    # name, kind, ctor kwargs, N / grid, weights, train
    ("irr_small_default", "irregular", dict(dim=64, heads=4, dim_head=16, slice_num=8, rotate=45), 512, "default", False),
    ("irr_small_amp", "irregular", dict(dim=64, heads=4, dim_head=16, slice_num=16, rotate=45), 777, "amp", False),
    ("irr_small_amp_train", "irregular", dict(dim=64, heads=4, dim_head=16, slice_num=16, rotate=45), 600, "amp", True),
    ("irr_c2like_amp", "irregular", dict(dim=128, heads=4, dim_head=64, slice_num=32, rotate=45), 1500, "amp", False),
    ("irr_c3like_model", "irregular", dict(dim=256, heads=8, dim_head=32, slice_num=64, rotate=45), 1024, "model", False),
    ("irr_c3like_amp", "irregular", dict(dim=256, heads=8, dim_head=32, slice_num=64, rotate=45), 1024, "amp", False),
    ("irr_m128_amp", "irregular", dict(dim=64, heads=8, dim_head=32, slice_num=128, rotate=45), 640, "amp", False),
    ("irr_norot_amp", "irregular", dict(dim=64, heads=4, dim_head=16, slice_num=16, rotate=0), 300, "amp", False),
    ("irr_b2_amp", "irregular", dict(dim=64, heads=4, dim_head=16, slice_num=16, rotate=45), 400, "amp", False),
    ("irr_nonpow2_amp", "irregular", dict(dim=64, heads=4, dim_head=16, slice_num=12, rotate=45, ball_sizes=[16, 8]), 333, "amp", False),
    ("s2d_c1like_amp", "structured2d", dict(dim=64, heads=4, dim_head=16, slice_num=8, H=16, W=16, rotate=45), (16, 16), "amp", False),
    ("s3d_c5like_amp", "structured3d", dict(dim=64, heads=4, dim_head=16, slice_num=16, H=8, W=8, D=8, rotate=45), (8, 8, 8), "amp", False),
    # round 2 -- stride-1 "no pooling" level (pooling.py:26-27): the pooled node IS the node, widths must agree
    ("irr_stride1_amp", "irregular", dict(dim=64, heads=4, dim_head=16, slice_num=16, rotate=45, strides=[1],
                                          c_hidden=[16, 16], ball_sizes=[16, 8]), 500, "amp", False),
    ("irr_stride1_amp_train", "irregular", dict(dim=64, heads=4, dim_head=16, slice_num=16, rotate=45, strides=[1],
                                                c_hidden=[16, 16], ball_sizes=[16, 8]), 450, "amp", True),
    ("irr_b2_amp_train", "irregular", dict(dim=64, heads=4, dim_head=16, slice_num=16, rotate=45), 350, "amp", True),
    ("irr_c3like_amp_train", "irregular", dict(dim=256, heads=8, dim_head=32, slice_num=64, rotate=45), 900, "amp", True),
    # round 4 -- the reference's own velocity-mem-checker shape (benchmarks/00-VelocityMemMocker/velocity-mem-checker.py:34-44:
    # n_hidden 256, n_head 8, slice_num 32): 32-token clouds whose coarse level has 16 tokens per cloud (the fragment kernel's
    # padded tile, the MFMA ball attention on a partial tile)
    ("irr_memchk_amp", "irregular", dict(dim=256, heads=8, dim_head=32, slice_num=32, rotate=45), 800, "amp", False),
    ("irr_memchk_b2_amp", "irregular", dict(dim=256, heads=8, dim_head=32, slice_num=32, rotate=45), 300, "amp", False),
    # round 4 -- the widths of the reference's PDE-benchmark scripts (benchmarks/04-PDE-Solving-StandardBenchmark/scripts/*.sh:
    # n_hidden 128, 8 heads -> dim_head 16; slice_num 64 / 128): dim_head 16 with >= 32 slices runs the split-bf16 passes padded
    # to 32 channels per head, 128-token x 16-channel clouds the wave-cooperative tree builder
    ("irr_pde_amp", "irregular", dict(dim=128, heads=8, dim_head=16, slice_num=64, rotate=45), 1100, "amp", False),
    ("irr_pde_m128_amp", "irregular", dict(dim=128, heads=8, dim_head=16, slice_num=128, rotate=45), 700, "amp", False),
    ("s2d_pde_amp", "structured2d", dict(dim=128, heads=8, dim_head=16, slice_num=64, H=24, W=20, rotate=45), (24, 20), "amp", False),
]

# BASELINE configs C1 and C5 at their REAL widths (dim 256 -> 8 heads x 32: inside the HIP convolution's envelope),
# C1 at its real grid, C5 on a grid the CPU can check.  "big" fixtures: inputs and the convolution weight are
# int8 * 2^-k (exact in fp32, 1 byte per value in the file), per-point outputs are kept for a row sample only.
BIG_CASES = [
    ("c1_real", "structured2d", dict(dim=256, heads=8, dim_head=32, slice_num=8, H=64, W=64, rotate=45), (64, 64), 4096.0),
    ("c5_real", "structured3d", dict(dim=256, heads=8, dim_head=32, slice_num=128, H=8, W=16, D=16, rotate=45), (8, 16, 16), 8192.0),
]


def gen_attention_fixtures(models, tree_store, only=None):
    from models.PhysicsAttention import (Physics_Attention_Irregular_Mesh,
                                         Physics_Attention_Structured_Mesh_2D,
                                         Physics_Attention_Structured_Mesh_3D)

    ctor = {"irregular": Physics_Attention_Irregular_Mesh,
            "structured2d": Physics_Attention_Structured_Mesh_2D,
            "structured3d": Physics_Attention_Structured_Mesh_3D}
    for ci, (name, kind, kw, n, weights, train) in enumerate(ATTN_CASES):
        if only and ("attn_" + name) not in only:
            continue
        torch.manual_seed(100 + ci)
        gen = torch.Generator().manual_seed(1000 + ci)
        mod = ctor[kind](**kw)
        if weights == "model":
            model_init_rule(mod)
        elif weights == "amp":
            amplify(mod, gen)
        mod.train(train)
        N = n if isinstance(n, int) else int(np.prod(n))
        B = 2 if "b2" in name else 1
        # fp16-representable inputs keep the fixture small and exactly reproducible
        x = torch.randn(B, N, kw["dim"], generator=gen).half().float()
        sd = {k: v.clone() for k, v in mod.state_dict().items()}  # BEFORE forward (BN stats move in train)
        with torch.no_grad():
            cap = run_attention(mod, x, tree_store)
        meta = dict(kind=kind, kw=kw, grid=(n if not isinstance(n, int) else ()), train=train, B=B, N=N)
        arrays = {"x": x.half().numpy()}
        arrays.update({"sd/" + k: v for k, v in to_np(sd).items()})
        keep_rows = 48  # per-point tensors: keep a strided sample of rows + the full final output
        rows = torch.linspace(0, N - 1, keep_rows).long()
        arrays["rows"] = rows.numpy()
        arrays["exp/x_lin_rows"] = _xlin_rows(cap["x_lin_raw"], kind, B, N).index_select(1, rows).numpy()
        for k, v in cap.items():
            if k in ("x_lin_raw",):
                continue
            arrays["exp/" + k] = v.numpy()
        arrays["meta"] = np.frombuffer(repr(meta).encode(), dtype=np.uint8)
        path = os.path.join(HERE, f"attn_{name}.npz")
        np.savez_compressed(path, **arrays)
        print(f"{path}: {os.path.getsize(path)/1e6:.2f} MB  out_norm={cap['out'].norm():.4f}")


def gen_big_fixtures(models, tree_store, only=None):
    from models.PhysicsAttention import Physics_Attention_Structured_Mesh_2D, Physics_Attention_Structured_Mesh_3D

    ctor = {"structured2d": Physics_Attention_Structured_Mesh_2D, "structured3d": Physics_Attention_Structured_Mesh_3D}
    for ci, (name, kind, kw, grid, wscale) in enumerate(BIG_CASES):
        if only and ("big_" + name) not in only:
            continue
        torch.manual_seed(300 + ci)
        gen = torch.Generator().manual_seed(3000 + ci)
        mod = ctor[kind](**kw)
        amplify(mod, gen)
        N = int(np.prod(grid))
        xq = torch.clamp(torch.round(torch.randn(1, N, kw["dim"], generator=gen) * 32.0), -127, 127).to(torch.int8)
        x = xq.float() / 32.0
        with torch.no_grad():  # the convolution weight on an int8 grid (its bias and everything else stay fp32)
            wq = torch.clamp(torch.round(mod.in_project_x.weight * wscale), -127, 127).to(torch.int8)
            mod.in_project_x.weight.copy_(wq.float() / wscale)
        mod.eval()
        sd = {k: v.clone() for k, v in mod.state_dict().items()}
        with torch.no_grad():
            cap = run_attention(mod, x, tree_store)
        meta = dict(kind=kind, kw=kw, grid=grid, train=False, B=1, N=N)
        rows = torch.linspace(0, N - 1, 384).long()
        arrays = {"x_i8": xq.numpy(), "x_scale": np.float32(32.0), "w_i8": wq.numpy(), "w_scale": np.float32(wscale),
                  "rows": rows.numpy()}
        arrays.update({"sd/" + k: v for k, v in to_np(sd).items() if k != "in_project_x.weight"})
        arrays["exp/x_lin_rows"] = _xlin_rows(cap["x_lin_raw"], kind, 1, N).index_select(1, rows).numpy()
        arrays["exp/out_rows"] = cap["out"].index_select(1, rows).numpy()
        arrays["exp/out_norm"] = np.float64(cap["out"].double().norm())
        for k, v in cap.items():
            if k in ("x_lin_raw", "out"):
                continue
            arrays["exp/" + k] = v.numpy()
        arrays["meta"] = np.frombuffer(repr(meta).encode(), dtype=np.uint8)
        path = os.path.join(HERE, f"big_{name}.npz")
        np.savez_compressed(path, **arrays)
        print(f"{path}: {os.path.getsize(path)/1e6:.2f} MB  out_norm={cap['out'].norm():.4f}")


GRAD_CASES = ["irr_small_amp", "irr_small_amp_train", "irr_b2_amp", "irr_b2_amp_train", "irr_norot_amp",
              "s2d_c1like_amp", "s3d_c5like_amp", "irr_stride1_amp", "irr_stride1_amp_train", "irr_nonpow2_amp",
              "irr_pde_amp", "irr_pde_m128_amp", "s2d_pde_amp"]  # (round 4: the PDE scripts' width -- the padded dim_head-16 backward)


def gen_grad_fixtures(models, only=None):
    """Gradients of the REFERENCE module (its own autograd) w.r.t. the input and every parameter for
    loss = <out, gout>, on the inputs / weights of an existing attention fixture (read back from its .npz)."""
    import ast

    from models.PhysicsAttention import (Physics_Attention_Irregular_Mesh, Physics_Attention_Structured_Mesh_2D,
                                         Physics_Attention_Structured_Mesh_3D)

    ctor = {"irregular": Physics_Attention_Irregular_Mesh, "structured2d": Physics_Attention_Structured_Mesh_2D,
            "structured3d": Physics_Attention_Structured_Mesh_3D}
    for ci, name in enumerate(GRAD_CASES):
        if only and ("grad_" + name) not in only:
            continue
        z = np.load(os.path.join(HERE, f"attn_{name}.npz"))
        meta = ast.literal_eval(bytes(z["meta"]).decode())
        mod = ctor[meta["kind"]](**meta["kw"])
        mod.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")})
        mod.train(meta["train"])
        x = torch.from_numpy(z["x"].astype(np.float32)).requires_grad_(True)
        gen = torch.Generator().manual_seed(4000 + ci)
        gout = torch.randn(x.shape[0], x.shape[1], meta["kw"]["dim"], generator=gen).half().float()
        out = mod(x)
        assert np.allclose(out.detach().numpy(), z["exp/out"], rtol=0, atol=1e-5 * float(np.abs(z["exp/out"]).max()))
        (out * gout).sum().backward()
        arrays = {"gout": gout.half().numpy(), "gx": x.grad.numpy()}
        for pn, p in mod.named_parameters():
            if p.grad is not None:
                arrays["g/" + pn] = p.grad.numpy()
            else:
                arrays["nograd/" + pn] = np.zeros(0, dtype=np.float32)
        path = os.path.join(HERE, f"grad_{name}.npz")
        np.savez_compressed(path, **arrays)
        print(f"{path}: {os.path.getsize(path)/1e6:.2f} MB  |gx|={x.grad.norm():.4f}")


def _xlin_rows(raw, kind, B, N):
    if kind == "irregular":
        return raw
    if kind == "structured2d":
        return raw.permute(0, 2, 3, 1).reshape(B, N, -1)
    return raw.permute(0, 2, 3, 4, 1).reshape(B, N, -1)


# Stand-alone ErwinFlash on raw point clouds (benchmarks/02.1-Car-Design-ShapeNetCar/main.py:50-99):
# embed=True, mp_steps>0, clouds of unequal size that are not powers of two (padded tree, tree_mask).
EF_CASES = [
    # name, ctor kwargs, cloud sizes, edges ("radius" r | "given" r), train
    ("car_like", dict(c_in=7, c_hidden=[128, 128], ball_sizes=[64, 32], enc_num_heads=[8, 8], enc_depths=[4, 4],
                      dec_num_heads=[8], dec_depths=[4], strides=[2], rotate=45, decode=True, mlp_ratio=4,
                      dimensionality=3, mp_steps=3), [700], ("given", 0.15), False),
    ("two_clouds", dict(c_in=5, c_hidden=[64, 128], ball_sizes=[32, 16], enc_num_heads=[4, 8], enc_depths=[2, 2],
                        dec_num_heads=[4], dec_depths=[2], strides=[2], rotate=45, decode=True, mlp_ratio=2,
                        dimensionality=3, mp_steps=2), [300, 500], ("radius", 0.2), False),
    ("two_clouds_train", dict(c_in=5, c_hidden=[64, 128], ball_sizes=[32, 16], enc_num_heads=[4, 8], enc_depths=[2, 2],
                              dec_num_heads=[4], dec_depths=[2], strides=[2], rotate=45, decode=True, mlp_ratio=2,
                              dimensionality=3, mp_steps=2), [260, 410], ("radius", 0.2), True),
    ("planar_norot", dict(c_in=4, c_hidden=[32, 64, 64], ball_sizes=[32, 16, 8], enc_num_heads=[2, 4, 4],
                          enc_depths=[2, 2, 2], dec_num_heads=[2, 4], dec_depths=[1, 2], strides=[2, 2], rotate=0,
                          decode=True, mlp_ratio=4, dimensionality=2, mp_steps=1), [512, 200], ("radius", 0.1), False),
    ("no_mpnn", dict(c_in=6, c_hidden=[64, 64], ball_sizes=[64, 16], enc_num_heads=[4, 4], enc_depths=[2, 2],
                     dec_num_heads=[4], dec_depths=[2], strides=[4], rotate=30, decode=True, mlp_ratio=4,
                     dimensionality=3, mp_steps=0), [1000], ("none", 0.0), False),
]


def amplify_erwinflash(mod, gen, scale=6.0):
    with torch.no_grad():
        for name, p in mod.named_parameters():
            if p.dim() == 2 and not name.startswith("embed."):
                p.mul_(scale)
            elif p.dim() == 2:
                p.mul_(3.0)
            elif name.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=gen))
            elif "norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=gen))
        for name, b in mod.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=gen))
            elif name.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=gen))


def gen_erwinflash_fixtures(models, tree_store, only=None):
    import torch_cluster
    from models.components.erwinflash import ErwinTransformer

    for ci, (name, kw, sizes, (edges, r), train) in enumerate(EF_CASES):
        if only and ("ef_" + name) not in only:
            continue
        torch.manual_seed(500 + ci)
        gen = torch.Generator().manual_seed(5000 + ci)
        mod = ErwinTransformer(**kw)
        amplify_erwinflash(mod, gen)
        mod.train(train)
        q8 = {}
        if name == "car_like":  # 3.3M matrix entries: kept on an int8 * 2^-k grid, one byte each in the file
            with torch.no_grad():
                for pn, p in mod.named_parameters():
                    if p.dim() == 2:
                        k = math.floor(math.log2(127.0 / float(p.abs().max())))
                        q = torch.clamp(torch.round(p * 2.0 ** k), -127, 127).to(torch.int8)
                        p.copy_(q.float() * 2.0 ** -k)
                        q8[pn] = (q.numpy(), np.float32(2.0 ** -k))
        n, d = sum(sizes), kw["dimensionality"]
        pos = torch.rand(n, d, generator=gen)  # full fp32 mantissas: no equal coordinates
        feat = torch.randn(n, kw["c_in"], generator=gen).half().float()
        batch_idx = torch.cat([torch.full((s,), i, dtype=torch.long) for i, s in enumerate(sizes)])
        edge_index = torch_cluster.radius_graph(pos, r, batch=batch_idx, loop=True) if edges != "none" else None
        sd = {k: v.clone() for k, v in mod.state_dict().items()}
        cap, hooks = {}, []
        if kw.get("embed", True):
            hooks.append(mod.embed.register_forward_hook(lambda m, i, o: cap.update(embed=o.detach().clone())))
        tree_store.clear()
        with torch.no_grad():
            if edges == "given":
                out = mod(feat, pos, batch_idx, edge_index=edge_index)
            else:
                out = mod(feat, pos, batch_idx, radius=r)
        for h in hooks:
            h.remove()
        arrays = {"pos": pos.numpy(), "feat": feat.half().numpy(), "batch_idx": batch_idx.numpy(), "exp/out": out.numpy()}
        if edge_index is not None:
            arrays["edge_index"] = edge_index.numpy().astype(np.int32)
        arrays.update({"exp/" + k: v.numpy() for k, v in cap.items()})
        arrays.update({"exp/" + k: v.numpy() for k, v in tree_store.items()})
        arrays.update({"sd/" + k: v for k, v in to_np(sd).items() if k not in q8})
        for k, (q, sc) in q8.items():
            arrays["sdq/" + k], arrays["sdscale/" + k] = q, sc
        if train:
            arrays.update({"sd_after/" + k: v for k, v in to_np(mod.state_dict()).items() if "running" in k or "tracked" in k})
        meta = dict(kw=kw, sizes=sizes, edges=edges, radius=r, train=train)
        arrays["meta"] = np.frombuffer(repr(meta).encode(), dtype=np.uint8)
        path = os.path.join(HERE, f"ef_{name}.npz")
        np.savez_compressed(path, **arrays)
        print(f"{path}: {os.path.getsize(path)/1e6:.2f} MB  out_norm={out.norm():.4f}  edges="
              f"{0 if edge_index is None else edge_index.shape[1]}")


EF_GRAD_CASES = ["two_clouds", "two_clouds_train", "planar_norot"]


def gen_erwinflash_grad_fixtures(models, only=None):
    """Gradients of the REFERENCE ErwinTransformer (its own autograd) w.r.t. the node features and every parameter
    for loss = <out, gout>, on the inputs / weights of an existing ef_* fixture (read back from its .npz)."""
    import ast

    from models.components.erwinflash import ErwinTransformer

    for ci, name in enumerate(EF_GRAD_CASES):
        if only and ("efgrad_" + name) not in only:
            continue
        z = np.load(os.path.join(HERE, f"ef_{name}.npz"))
        meta = ast.literal_eval(bytes(z["meta"]).decode())
        mod = ErwinTransformer(**meta["kw"])
        mod.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")})
        mod.train(meta["train"])
        feat = torch.from_numpy(z["feat"].astype(np.float32)).requires_grad_(True)
        pos, batch_idx = torch.from_numpy(z["pos"]), torch.from_numpy(z["batch_idx"])
        if meta["edges"] == "given":
            out = mod(feat, pos, batch_idx, edge_index=torch.from_numpy(z["edge_index"].astype(np.int64)))
        else:
            out = mod(feat, pos, batch_idx, radius=meta["radius"])
        assert np.allclose(out.detach().numpy(), z["exp/out"], rtol=0, atol=1e-5 * float(np.abs(z["exp/out"]).max()))
        gen = torch.Generator().manual_seed(6000 + ci)
        gout = torch.randn(out.shape, generator=gen).half().float()
        (out * gout).sum().backward()
        arrays = {"gout": gout.half().numpy(), "gfeat": feat.grad.numpy()}
        for pn, p in mod.named_parameters():
            if p.grad is not None:
                arrays["g/" + pn] = p.grad.numpy()
            else:
                arrays["nograd/" + pn] = np.zeros(0, dtype=np.float32)
        path = os.path.join(HERE, f"efgrad_{name}.npz")
        np.savez_compressed(path, **arrays)
        print(f"{path}: {os.path.getsize(path)/1e6:.2f} MB  |gfeat|={feat.grad.norm():.4f}")


MODEL_CASES = [
    ("irr", "HAETransolver_Irregular_Mesh", dict(space_dim=3, fun_dim=4, n_hidden=64, n_layers=2, n_head=4, mlp_ratio=2, out_dim=4, slice_num=16), 500, True),
    ("irr_nofx", "HAETransolver_Irregular_Mesh", dict(space_dim=3, fun_dim=0, n_hidden=64, n_layers=1, n_head=4, mlp_ratio=1, out_dim=1, slice_num=8), 300, False),
    ("s2d", "HAETransolver_Structured_Mesh_2D", dict(space_dim=2, fun_dim=1, n_hidden=64, n_layers=1, n_head=8, out_dim=1, slice_num=8, H=16, W=16), 256, True),
    ("s3d", "HAETransolver_Structured_Mesh_3D", dict(space_dim=3, fun_dim=1, n_hidden=64, n_layers=1, n_head=4, out_dim=1, slice_num=16, H=8, W=8, D=8), 512, True),
]


def gen_model_fixtures(models):
    for ci, (name, cls, kw, N, with_fx) in enumerate(MODEL_CASES):
        torch.manual_seed(200 + ci)
        gen = torch.Generator().manual_seed(2000 + ci)
        model = getattr(models, cls)(**kw).eval()
        for blk in model.blocks:
            amplify(blk.Attn, gen)
        x = torch.rand(1, N, kw["space_dim"], generator=gen).half().float()
        fx = torch.rand(1, N, kw["fun_dim"], generator=gen).half().float() if with_fx else None
        with torch.no_grad():
            out = model(x, fx)
        arrays = {"x": x.numpy(), "out": out.numpy()}
        if fx is not None:
            arrays["fx"] = fx.numpy()
        arrays.update({"sd/" + k: v for k, v in to_np(model.state_dict()).items()})
        arrays["meta"] = np.frombuffer(repr(dict(cls=cls, kw=kw, N=N)).encode(), dtype=np.uint8)
        path = os.path.join(HERE, f"model_{name}.npz")
        np.savez_compressed(path, **arrays)
        print(f"{path}: {os.path.getsize(path)/1e6:.2f} MB  out_norm={out.norm():.4f}")


def gen_tree_fixtures():
    """Tree-only fixtures straight from the reference's Python entry point
    (balltree.build_balltree_with_rotations, balltree.pyx:598-662)."""
    import balltree as ref_bt  # oracle/_ref/balltree*.so

    rng = np.random.default_rng(7)
    cases = []
    for M in (2, 5, 6, 8, 12, 16, 32, 64, 128):
        for d in (2, 3):
            for kind in ("rand", "ties", "grid", "line"):
                for rep in range(6):
                    nb = int(rng.integers(1, 5))
                    T = nb * M
                    if kind == "rand":
                        p = rng.random((T, d))
                    elif kind == "ties":
                        p = rng.integers(0, 3, (T, d)).astype(np.float64)
                    elif kind == "grid":
                        p = np.round(rng.random((T, d)) * 4) / 4
                    else:
                        p = rng.random((T, d))
                        p[:, 1:] = 0.25
                    cases.append((M, d, nb, p.astype(np.float32)))
    out = {}
    for i, (M, d, nb, p) in enumerate(cases):
        bidx = torch.arange(nb).repeat_interleave(M)
        pow2 = (M & (M - 1)) == 0
        ball_sizes = [min(32, M), max(1, min(16, M // 2))]
        angle = 45.0 if pow2 else 0.0
        ti, tm, rots = ref_bt.build_balltree_with_rotations(torch.from_numpy(p), bidx, [2], ball_sizes, angle)
        out[f"{i}/pos"] = p
        out[f"{i}/M"] = np.int64(M)
        out[f"{i}/nb"] = np.int64(nb)
        out[f"{i}/angle"] = np.float64(angle)
        out[f"{i}/tree_idx"] = ti.numpy()
        out[f"{i}/tree_mask"] = tm.numpy()
        for j, r in enumerate(rots):
            if r is not None:
                out[f"{i}/rot{j}"] = r.numpy()
    out["count"] = np.int64(len(cases))
    path = os.path.join(HERE, "tree_cases.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {len(cases)} cases, {os.path.getsize(path)/1e6:.2f} MB")


if __name__ == "__main__":
    # no arguments: regenerate everything; otherwise only the named fixture files (without .npz), e.g.
    #   make_golden.py attn_irr_stride1_amp big_c1_real grad_irr_small_amp
    # (an .npz is a zip: rewriting an unchanged fixture changes its bytes through the member timestamps)
    only = set(sys.argv[1:])
    torch.set_num_threads(8)
    models = import_reference()
    store = capture_tree(models)
    if not only:
        gen_tree_fixtures()
        gen_model_fixtures(models)
    gen_attention_fixtures(models, store, only)
    gen_big_fixtures(models, store, only)
    gen_grad_fixtures(models, only)
    gen_erwinflash_fixtures(models, store, only)
    gen_erwinflash_grad_fixtures(models, only)
