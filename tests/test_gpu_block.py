"""GPU: the point-wise layers around the attention on HIP kernels (haet_amd/_fast.py) --
LayerNorm, the narrow first Linear of the preprocess MLP, the GELU epilogue of the MFMA GEMM -- against fp64
torch, and the whole model with and without them."""
import pytest
import torch

from oracle import haet_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,D", [(1, 256), (1000, 256), (70001, 256), (33, 64), (257, 128), (100, 1024), (64, 36), (5, 4)])
def test_layernorm_vs_fp64(rows, D, hip_device):
    from haet_amd import ops

    g = torch.Generator().manual_seed(rows + D)
    x = torch.randn(rows, D, generator=g) * 3 + 1.5
    w, b = torch.randn(D, generator=g), torch.randn(D, generator=g)
    ref = torch.nn.functional.layer_norm(x.double(), (D,), w.double(), b.double(), 1e-5)
    out = ops.layernorm(x.to(hip_device), w.to(hip_device), b.to(hip_device), 1e-5)
    assert O.rel_l2(out.cpu().double(), ref) < 2e-6


@pytest.mark.parametrize("rows,K,Oc,gelu", [(1, 7, 512, True), (1000, 7, 512, True), (50001, 3, 256, False), (33, 16, 64, True),
                                           (10, 1, 4, False)])
def test_narrow_linear_vs_fp64(rows, K, Oc, gelu, hip_device):
    from haet_amd import ops

    g = torch.Generator().manual_seed(rows + K)
    x, W, b = torch.randn(rows, K, generator=g), torch.randn(Oc, K, generator=g), torch.randn(Oc, generator=g)
    ref = x.double() @ W.double().t() + b.double()
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    out = ops.linear_narrow(x.to(hip_device), W.to(hip_device), b.to(hip_device), gelu)
    assert O.rel_l2(out.cpu().double(), ref) < 2e-6


@pytest.mark.parametrize("rows,K,Oc", [(1000, 256, 512), (70000, 256, 256), (513, 512, 256), (64, 128, 256), (100, 64, 128),
                                       (66000, 512, 256)])
def test_linear_with_gelu_epilogue_vs_fp64(rows, K, Oc, hip_device):
    from haet_amd import ops

    g = torch.Generator().manual_seed(rows + K)
    x, W, b = torch.randn(rows, K, generator=g), torch.randn(Oc, K, generator=g) / K ** 0.5, torch.randn(Oc, generator=g)
    ref = torch.nn.functional.gelu(x.double() @ W.double().t() + b.double())
    w = ops.SplitLinear(W.to(hip_device), b.to(hip_device))
    out = ops.linear(x.to(hip_device), w, gelu=True)
    assert O.rel_l2(out.cpu().double(), ref) < 2e-5


@pytest.mark.parametrize("family", ["irregular", "structured2d"])
def test_model_fast_block_matches_stock_layers(family, hip_device, monkeypatch):
    """The same model forward with the LayerNorm / MLP layers on the HIP kernels and on the stock modules."""
    import haet_amd

    torch.manual_seed(1)
    if family == "irregular":
        model = haet_amd.HAETransolver_Irregular_Mesh(space_dim=3, fun_dim=4, n_hidden=256, n_layers=2, n_head=8,
                                                      mlp_ratio=2, out_dim=4, slice_num=64)
        args = (torch.randn(1, 3000, 7), None)
    else:
        model = haet_amd.HAETransolver_Structured_Mesh_2D(space_dim=2, fun_dim=1, n_hidden=128, n_layers=2, n_head=4,
                                                          mlp_ratio=1, out_dim=1, slice_num=16, H=24, W=20)
        args = (torch.rand(2, 480, 2), torch.rand(2, 480, 1))
    model = model.eval().to(hip_device)
    args = tuple(None if a is None else a.to(hip_device) for a in args)
    with torch.no_grad():
        fast = model(*args)
        monkeypatch.setenv("HAET_NO_BLOCK_FAST", "1")
        stock = model(*args)
    assert torch.isfinite(fast).all()
    assert O.rel_l2(fast.cpu(), stock.cpu()) < 2e-5


@pytest.mark.parametrize("rows,D", [(1, 256), (1000, 256), (50001, 256), (33, 64), (100, 1024), (64, 36)])
def test_layernorm_backward_vs_fp64(rows, D, hip_device):
    from haet_amd import ops

    g = torch.Generator().manual_seed(rows + D)
    x = (torch.randn(rows, D, generator=g) * 2 + 0.5)
    w, b, gy = torch.randn(D, generator=g), torch.randn(D, generator=g), torch.randn(rows, D, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    torch.nn.functional.layer_norm(xr, (D,), wr, br, 1e-5).backward(gy.double())
    dx, dg, db = ops.layernorm_bwd(x.to(hip_device), w.to(hip_device), gy.to(hip_device), 1e-5)
    assert O.rel_l2(dx.cpu().double(), xr.grad) < 5e-6
    assert O.rel_l2(dg.cpu().double(), wr.grad) < 5e-6
    assert O.rel_l2(db.cpu().double(), br.grad) < 5e-6


def test_model_training_step_fast_layers_match_stock(hip_device, monkeypatch):
    """loss.backward() through the whole model with the LayerNorm / Linear layers as HIP autograd nodes against
    the same step on the stock torch layers: loss, and every parameter gradient."""
    import haet_amd

    torch.manual_seed(2)
    model = haet_amd.HAETransolver_Irregular_Mesh(space_dim=3, fun_dim=4, n_hidden=256, n_layers=2, n_head=8,
                                                  mlp_ratio=2, out_dim=4, slice_num=64).to(hip_device).train()
    x = torch.randn(1, 2500, 7, device=hip_device)
    y = torch.randn(1, 2500, 4, device=hip_device)
    state = {k: v.clone() for k, v in model.state_dict().items()}

    def step():
        model.load_state_dict(state)
        model.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(model(x, None), y)
        loss.backward()
        return float(loss.detach()), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    loss_fast, g_fast = step()
    monkeypatch.setenv("HAET_NO_BLOCK_FAST", "1")
    loss_ref, g_ref = step()
    assert abs(loss_fast - loss_ref) < 1e-5 * max(1.0, abs(loss_ref))
    assert set(g_fast) == set(g_ref)
    scale = max(float(g.norm()) for g in g_ref.values())
    for name, gr in g_ref.items():
        err = float((g_fast[name] - gr).norm()) / max(float(gr.norm()), 1e-4 * scale)
        assert err < 2e-3, (name, err)


@pytest.mark.parametrize("rows,D", [(1, 256), (4097, 256), (77, 128), (10, 1024)])
def test_add_layernorm_vs_fp64(rows, D, hip_device):
    from haet_amd import ops

    g = torch.Generator().manual_seed(rows + D)
    x, r = torch.randn(rows, D, generator=g), torch.randn(rows, D, generator=g) * 2
    w, b = torch.randn(D, generator=g), torch.randn(D, generator=g)
    s, out = ops.add_layernorm(x.to(hip_device), r.to(hip_device), w.to(hip_device), b.to(hip_device), 1e-5)
    assert torch.equal(s.cpu(), x + r)
    ref = torch.nn.functional.layer_norm((x + r).double(), (D,), w.double(), b.double(), 1e-5)
    assert O.rel_l2(out.cpu().double(), ref) < 2e-6


@pytest.mark.parametrize("rows,K,Oc,gelu", [(1000, 256, 512, True), (70001, 256, 256, False), (64, 128, 256, True), (4099, 128, 512, False)])
def test_layernorm_fused_into_the_gemm_loader_vs_fp64(rows, K, Oc, gelu, hip_device):
    from haet_amd import ops

    g = torch.Generator().manual_seed(rows + K)
    x = torch.randn(rows, K, generator=g) * 2 + 0.7
    lw, lb = torch.randn(K, generator=g), torch.randn(K, generator=g)
    W, b = torch.randn(Oc, K, generator=g) / K ** 0.5, torch.randn(Oc, generator=g)
    ref = torch.nn.functional.layer_norm(x.double(), (K,), lw.double(), lb.double(), 1e-5) @ W.double().t() + b.double()
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    assert ops.linear_ln_supported(K, Oc, rows)
    w = ops.SplitLinear(W.to(hip_device), b.to(hip_device))
    out = ops.linear(x.to(hip_device), w, gelu=gelu, ln=(lw.to(hip_device), lb.to(hip_device), 1e-5))
    assert O.rel_l2(out.cpu().double(), ref) < 2e-5


def test_pre_norm_in_the_projection_matches_separate_layernorm(hip_device):
    """Attn(x, pre_norm=ln) (LayerNorm inside the projection GEMM's loader) against Attn(ln(x))."""
    from haet_amd.PhysicsAttention import Physics_Attention_Irregular_Mesh

    torch.manual_seed(4)
    mod = Physics_Attention_Irregular_Mesh(dim=256, heads=8, dim_head=32, slice_num=64, rotate=45).eval().to(hip_device)
    ln = torch.nn.LayerNorm(256).to(hip_device)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5)
        ln.bias.normal_(0, 0.3)
        x = torch.randn(1, 5000, 256, device=hip_device) * 1.7 + 0.2
        assert mod.fuses_pre_norm(x, ln)
        fused = mod(x, pre_norm=ln)
        ref = mod(ln(x))
    assert O.rel_l2(fused.cpu(), ref.cpu()) < 2e-5


@pytest.mark.parametrize("kind", ["2d", "3d"])
def test_pre_norm_in_the_convolution_presplit_matches_separate_layernorm(kind, hip_device, monkeypatch):
    """Structured variants (round 4): Attn(x, pre_norm=ln) with the LayerNorm inside the pre-split pass of the LDS-DMA
    convolution (haet_conv_presplit) against Attn(ln(x)), on grids large enough for that kernel (>= 192 row tiles); and the
    same module on the streaming convolution (HAET_CONV_V1=1)."""
    from haet_amd.PhysicsAttention import Physics_Attention_Structured_Mesh_2D, Physics_Attention_Structured_Mesh_3D

    torch.manual_seed(5)
    if kind == "2d":
        mod = Physics_Attention_Structured_Mesh_2D(dim=256, heads=8, dim_head=32, slice_num=32, H=240, W=230, rotate=45)
        N = 240 * 230
    else:
        mod = Physics_Attention_Structured_Mesh_3D(dim=256, heads=8, dim_head=32, slice_num=64, H=40, W=38, D=36, rotate=45)
        N = 40 * 38 * 36
    mod = mod.eval().to(hip_device)
    ln = torch.nn.LayerNorm(256).to(hip_device)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5)
        ln.bias.normal_(0, 0.3)
        x = torch.randn(1, N, 256, device=hip_device) * 1.7 + 0.2
        assert mod.fuses_pre_norm(x, ln)
        fused = mod(x, pre_norm=ln)
        ref = mod(ln(x))
        monkeypatch.setenv("HAET_CONV_V1", "1")
        assert not mod.fuses_pre_norm(x, ln)
        old = mod(x, pre_norm=ln)
    assert O.rel_l2(fused.cpu(), ref.cpu()) < 2e-5
    assert O.rel_l2(fused.cpu(), old.cpu()) < 2e-5


@pytest.mark.parametrize("mode", ["reduce-overhead", "max-autotune-no-cudagraphs"])
def test_model_under_torch_compile(mode, hip_device):
    """The reference's scripts wrap the model in torch.compile (exp_airfoil.py:137 "max-autotune", exp_plas.py:177
    "reduce-overhead").  Without gradients the model traces into ONE graph -- fullgraph=True raises on any graph break
    -- whose HIP stages are haet::module_forward operators (custom_op + fake impl, haet_amd/_compile.py); with
    "reduce-overhead" the whole forward, device-side ball tree included, replays as one captured HIP graph."""
    import haet_amd

    torch.manual_seed(3)
    model = haet_amd.HAETransolver_Irregular_Mesh(space_dim=3, fun_dim=4, n_hidden=256, n_layers=2, n_head=8,
                                                  mlp_ratio=2, out_dim=4, slice_num=64).eval().to(hip_device)
    x = torch.randn(1, 3000, 7, device=hip_device)
    with torch.no_grad():
        ref = model(x, None)
        compiled = torch.compile(model, mode=mode, fullgraph=True)
        outs = [compiled(x, None).clone() for _ in range(4)]  # warm-up, capture, replays
        x2 = torch.randn(1, 3000, 7, device=hip_device)
        ref2 = model(x2, None)
        out_x2 = compiled(x2, None).clone()  # same shapes, new data: a replay with refreshed inputs
    for o in outs:
        assert O.rel_l2(o.cpu(), ref.cpu()) < 1e-5
    assert O.rel_l2(out_x2.cpu(), ref2.cpu()) < 1e-5


def test_compiled_graph_holds_opaque_operators_only(hip_device):
    """What Dynamo sees of a 2-layer model: one graph, the preprocess operator + one operator per block."""
    import haet_amd

    torch.manual_seed(4)
    model = haet_amd.HAETransolver_Irregular_Mesh(space_dim=3, fun_dim=0, n_hidden=64, n_layers=2, n_head=4, out_dim=2,
                                                  slice_num=16).eval().to(hip_device)
    x = torch.randn(1, 1000, 3, device=hip_device)
    with torch.no_grad():
        ex = torch._dynamo.explain(model)(x, None)
    assert ex.graph_break_count == 0, ex.break_reasons
    names = [str(n.target) for g in ex.graphs for n in g.graph.nodes if n.op == "call_function"]
    assert sum("haet.module_forward" in n for n in names) == 3, names
    # an attention module compiled on its own is one operator too
    att = model.blocks[0].Attn
    with torch.no_grad():
        ex = torch._dynamo.explain(att)(torch.randn(1, 1000, 64, device=hip_device))
    assert ex.graph_break_count == 0 and ex.graph_count == 1


def test_training_step_under_torch_compile(hip_device):
    import haet_amd

    torch.manual_seed(5)
    model = haet_amd.HAETransolver_Irregular_Mesh(space_dim=3, fun_dim=4, n_hidden=256, n_layers=1, n_head=8,
                                                  mlp_ratio=2, out_dim=4, slice_num=64).to(hip_device).train()
    x = torch.randn(1, 2000, 7, device=hip_device)
    y = torch.randn(1, 2000, 4, device=hip_device)
    state = {k: v.clone() for k, v in model.state_dict().items()}

    def step(m):
        model.load_state_dict(state)
        model.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(m(x, None), y)
        loss.backward()
        return float(loss.detach()), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    loss_ref, g_ref = step(model)
    loss_c, g_c = step(torch.compile(model))
    assert abs(loss_c - loss_ref) < 1e-5 * max(1.0, abs(loss_ref))
    scale = max(float(g.norm()) for g in g_ref.values())
    for name, gr in g_ref.items():
        assert float((g_c[name] - gr).norm()) / max(float(gr.norm()), 1e-4 * scale) < 2e-3, name


@pytest.mark.parametrize("rows,od,with_res", [(1, 4, True), (5000, 4, True), (777, 1, False), (300, 16, True), (64, 7, False)])
def test_layernorm_narrow_linear_vs_fp64(rows, od, with_res, hip_device):
    from haet_amd import ops

    g = torch.Generator().manual_seed(rows + od)
    x, r = torch.randn(rows, 256, generator=g) * 1.5, torch.randn(rows, 256, generator=g)
    lw, lb = torch.randn(256, generator=g), torch.randn(256, generator=g)
    W, b = torch.randn(od, 256, generator=g) / 16, torch.randn(od, generator=g)
    s = x.double() + (r.double() if with_res else 0)
    ref = torch.nn.functional.layer_norm(s, (256,), lw.double(), lb.double(), 1e-5) @ W.double().t() + b.double()
    out = ops.ln_linear_small(x.to(hip_device), r.to(hip_device) if with_res else None, lw.to(hip_device), lb.to(hip_device),
                              1e-5, W.to(hip_device), b.to(hip_device))
    assert out.shape == (rows, od)
    assert O.rel_l2(out.cpu().double(), ref) < 2e-6


def test_reference_sweep_harness_small_sizes(hip_device):
    """The velocity-mem-checker sweep (haet_amd.harness.sweep) at its two smallest sizes, with and without autocast."""
    from haet_amd import harness

    for ac in (True, False):
        rows = harness.sweep(points=(1000, 10000), autocast=ac, timed=2, log=lambda m: None)
        assert [r["points"] for r in rows] == [1000, 10000]
        assert all(r["finite"] and r["seconds"] > 0 and r["peak_memory_mb"] > 0 for r in rows)


def test_car_evaluation_on_cached_sample_format(tmp_path, hip_device):
    """Cached-ShapeNetCar-format samples -> model -> the car benchmark's metrics, end to end on the device."""
    import numpy as np

    import haet_amd
    from haet_amd import harness
    from tests.test_harness_cpu import _write_samples

    names = _write_samples(str(tmp_path), [3000, 2500])
    samples, coef = harness.load_samples(str(tmp_path), names, norm=True)
    torch.manual_seed(0)
    model = haet_amd.HAETransolver_Irregular_Mesh(space_dim=7, fun_dim=0, n_hidden=64, n_layers=1, n_head=4, out_dim=4,
                                                  slice_num=16).to(hip_device).eval()
    ev = harness.CarEvaluation(coef)
    with torch.no_grad():
        for i, smp in enumerate(samples):
            x, fx = harness.model_inputs(smp, hip_device)
            out = model(x, fx)[0]
            ev.add(out, smp.y, smp.surf, seconds=0.01, cd_true=0.3 + 0.1 * i, cd_pred=0.31 + 0.1 * i)
    s = ev.summary()
    assert s["samples"] == 2 and np.isfinite(s["rel_l2_press"]) and np.isfinite(s["rel_l2_velo"]) and abs(s["rho_d"] - 1) < 1e-12


def test_eval_blob_follows_running_statistics_updated_by_a_train_mode_launch(hip_device):
    """A train-mode launch updates the BatchNorm running statistics through raw pointers (no `_version` bump): the
    eval-mode weight blob folded from the OLD statistics must not be reused afterwards (ErwinTransformer._stats_epoch).
    eval -> train-mode forward under no_grad (no optimizer step: the parameters' versions stay) -> eval, against a fresh
    module loaded from the state_dict."""
    from haet_amd.PhysicsAttention import Physics_Attention_Irregular_Mesh

    torch.manual_seed(5)
    kw = dict(dim=256, heads=8, dim_head=32, slice_num=64, rotate=45)
    mod = Physics_Attention_Irregular_Mesh(**kw)
    with torch.no_grad():
        for p in mod.erwin.parameters():
            if p.dim() >= 2:
                p.mul_(5.0)
    mod = mod.to(hip_device).eval()
    x = torch.randn(1, 3000, 256, device=hip_device)
    with torch.no_grad():
        y0 = mod(x)
        mod.train()
        for _ in range(3):
            mod(torch.randn(1, 3000, 256, device=hip_device) * 2.0 + 0.5)  # (running statistics move)
        mod.eval()
        y1 = mod(x)
        fresh = Physics_Attention_Irregular_Mesh(**kw).to(hip_device).eval()
        fresh.load_state_dict(mod.state_dict())
        y_ref = fresh(x)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    assert rel(y1, y_ref) < 1e-6
    assert rel(y1, y0) > 1e-5  # (the statistics did change the output: the test would pass vacuously otherwise)


def test_slice_parameter_scalars_of_all_live_modules_are_read_together(hip_device):
    """The Ada-Temp bias is a scalar kernel argument (a device -> host read when the weights changed): the first module
    that finds its parameters changed reads the scalars of every live module on its device in one copy
    (_base._refresh_slice_params) -- once per optimizer step, not once per layer."""
    from haet_amd.PhysicsAttention import Physics_Attention_Irregular_Mesh

    mods = [Physics_Attention_Irregular_Mesh(64, heads=4, dim_head=16, slice_num=8).to(hip_device) for _ in range(3)]
    with torch.no_grad():
        for i, m in enumerate(mods):
            m.ada_temp_linear.bias.fill_(0.1 * (i + 1))
    sp0 = mods[0]._slice_params()
    assert abs(sp0.bt - 0.1) < 1e-7
    for i, m in enumerate(mods[1:], 2):  # filled by the first module's refresh, with their own values
        assert m._slice_cache is not None and m._slice_cache[0] == m._slice_key() and abs(m._slice_cache[1].bt - 0.1 * i) < 1e-7
    cached = mods[2]._slice_cache[1]
    with torch.no_grad():
        mods[1].ada_temp_linear.bias.add_(1.0)   # an "optimizer step" on one module
    assert mods[2]._slice_params() is cached      # unchanged modules keep their cache
    assert abs(mods[1]._slice_params().bt - 1.2) < 1e-6
    import copy
    clone = copy.deepcopy(mods[0])                # (never ran __init__: registers itself at its first refresh)
    with torch.no_grad():
        clone.ada_temp_linear.bias.fill_(0.7)
    assert abs(clone._slice_params().bt - 0.7) < 1e-7 and abs(mods[0]._slice_params().bt - 0.1) < 1e-7


@pytest.mark.parametrize("O,K", [(256, 256), (128, 96), (512, 128), (48, 40)])
def test_split_linear_planes_carry_their_tile_ordered_copy(O, K, hip_device):
    """haet_split_bf16_tiled (ABI 9): behind each [O][K] bf16 plane its tile-ordered copy -- 16-B pieces (8 consecutive k of
    one row) at ((row / 16) * (K / 8) + k / 8) * 16 + row % 16 -- which the W-stationary kernels fetch; hi = RNE(w),
    hi + lo = w to 2^-16."""
    from haet_amd import ops

    torch.manual_seed(O + K)
    W = torch.randn(O, K)
    w = ops.SplitLinear(W.to(hip_device), None)
    assert w.tiled and w.hi.data_ptr() == w.hi_all.data_ptr() and w.hi_all.numel() == 2 * O * K
    for plane_all, want in ((w.hi_all, W.bfloat16()), (w.lo_all, (W - W.bfloat16().float()).bfloat16())):
        plane = plane_all[:O * K].view(torch.bfloat16).view(O, K).cpu()
        assert torch.equal(plane, want)
        tiled = plane_all[O * K:].view(torch.bfloat16).view(O // 16, K // 8, 16, 8).cpu()
        assert torch.equal(tiled.permute(0, 2, 1, 3).reshape(O, K), plane)


def test_structured_small_grid_model_replays_as_a_hip_graph(hip_device):
    """exp_plas.py:177 compiles its model with mode="reduce-overhead": the forward replays as one captured HIP graph.  The
    small-grid convolution splits its k-tiles over workgroups that ADD into a zeroed output; zeroed by hipMemsetAsync, that
    output came back as garbage from the second replay on (two layers, n_hidden 128: `tools/graph_replay_check.py`) -- the
    zeroing is a kernel now (csrc/common.h zero_async).  Five calls: eager warm-up, capture, three replays."""
    import haet_amd

    torch.manual_seed(0)
    model = haet_amd.HAETransolver_Structured_Mesh_2D(space_dim=2, n_layers=2, n_hidden=128, n_head=8, fun_dim=0, out_dim=1,
                                                      slice_num=64, H=61, W=31).eval().to(hip_device)
    x = torch.rand(2, 61 * 31, 2, device=hip_device)
    with torch.no_grad():
        ref = model(x, None)
        compiled = torch.compile(model, mode="reduce-overhead")
        outs = [compiled(x, None).clone() for _ in range(5)]
    for i, o in enumerate(outs):
        assert bool(torch.isfinite(o).all()), i
        assert O.rel_l2(o.cpu(), ref.cpu()) < 1e-4, i  # (float atomics: the summation order varies, ~1e-5)
