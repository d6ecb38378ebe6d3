"""CPU: pin the oracle (oracle/) against the golden vectors generated from the
reference itself (tests/golden/make_golden.py) -- SURVEY.md 8c.  fp32 vs fp32 on
the same CPU ops, so the tolerance is tight (1e-5 relative L2, a few ulps of
reassociation inside BLAS), and integer outputs (trees) are bit-exact."""
import numpy as np
import pytest
import torch

from oracle import balltree as obt
from oracle import haet_oracle as O
from tests import golden_util as G

TOL = 1e-5


def test_tree_fixtures_bit_exact():
    z = np.load(G.GOLDEN + "/tree_cases.npz")
    n = int(z["count"])
    assert n >= 400
    for i in range(n):
        pos = torch.from_numpy(z[f"{i}/pos"])
        M, nb, angle = int(z[f"{i}/M"]), int(z[f"{i}/nb"]), float(z[f"{i}/angle"])
        bidx = torch.arange(nb).repeat_interleave(M)
        ball_sizes = [min(32, M), max(1, min(16, M // 2))]
        ti, tm, rots = O.build_balltree_with_rotations(pos, bidx, [2], ball_sizes, angle)
        assert np.array_equal(ti.numpy(), z[f"{i}/tree_idx"]), i
        assert np.array_equal(tm.numpy(), z[f"{i}/tree_mask"]), i
        for j, r in enumerate(rots):
            if r is None:
                assert f"{i}/rot{j}" not in z.files
            else:
                assert np.array_equal(r.numpy(), z[f"{i}/rot{j}"]), (i, j)


@pytest.mark.skipif(not obt.ref_available(), reason="oracle/_ref not built (make -C oracle ref)")
def test_tree_oracle_vs_compiled_reference_header():
    rng = np.random.default_rng(11)
    for trial in range(2000):
        M = int(rng.choice([2, 3, 5, 6, 7, 8, 13, 16, 32, 33, 64, 100, 128]))
        d = int(rng.choice([2, 3]))
        nb = int(rng.integers(1, 4))
        kind = trial % 4
        if kind == 0:
            data = rng.random((nb * M, d))
        elif kind == 1:
            data = rng.integers(0, 4, (nb * M, d)).astype(float)
        elif kind == 2:
            data = np.round(rng.random((nb * M, d)), 1)
        else:
            data = rng.random((nb * M, d))
            data[:, 1:] = 0.5
        bidx = np.repeat(np.arange(nb), M)
        a, b = obt.build_balltree(data, bidx), obt.ref_build_balltree(data, bidx)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        if M & (M - 1) == 0:
            for tl in range(5):
                assert np.array_equal(obt.partition_balltree(data, bidx, tl),
                                      obt.ref_partition_balltree(data, bidx, tl))


@pytest.mark.parametrize("name", G.attn_fixture_names())
def test_attention_oracle_matches_reference(name):
    x, sd, meta, exp, rows = G.load_attn(name)
    cfg = G.oracle_cfg_from_meta(meta)
    trace = {}
    with torch.no_grad():
        out = O.physics_attention_forward(x, sd, cfg, training=meta["train"], trace=trace)
    # integer stage first: identical trees, otherwise nothing below is comparable
    for k in ("tree_idx", "tree_mask", "rot0", "rot1"):
        if "erwin." + k in exp:
            assert torch.equal(trace["erwin." + k], exp["erwin." + k]), k
    assert O.rel_l2(trace["x_lin"].index_select(1, rows), exp["x_lin_rows"]) < TOL
    assert O.rel_l2(trace["eidetic"].reshape(exp["eidetic_flat"].shape), exp["eidetic_flat"]) < TOL
    assert O.rel_l2(trace["pos"], exp["pos"]) < TOL
    assert torch.equal(trace["batch_idx"], exp["batch_idx"])
    for k, v in exp.items():
        if k.startswith("erwin.") and ("block" in k or "pool" in k):
            assert O.rel_l2(trace[k], v) < 5 * TOL, k
    assert O.rel_l2(trace["processed"].reshape(exp["processed_flat"].shape), exp["processed_flat"]) < 5 * TOL
    assert O.rel_l2(out, exp["out"]) < 5 * TOL


@pytest.mark.parametrize("name", G.model_fixture_names())
def test_model_oracle_matches_reference(name):
    x, fx, sd, meta, ref_out = G.load_model(name)
    kw = meta["kw"]
    kind = {"HAETransolver_Irregular_Mesh": "irregular", "HAETransolver_Structured_Mesh_2D": "structured2d",
            "HAETransolver_Structured_Mesh_3D": "structured3d"}[meta["cls"]]
    grid = {"irregular": (), "structured2d": (kw.get("H"), kw.get("W")),
            "structured3d": (kw.get("H"), kw.get("W"), kw.get("D"))}[kind]
    nh = kw.get("n_head", 8)
    cfg = O.AttnCfg(dim=kw["n_hidden"], heads=nh, dim_head=kw["n_hidden"] // nh,
                    slice_num=kw.get("slice_num", 32), kind=kind, grid=grid,
                    rotate=kw.get("rotate", 45), mlp_ratio=kw.get("mlp_ratio", 1))
    with torch.no_grad():
        out = O.model_forward(x, fx, sd, cfg, kw.get("n_layers", 5))
    assert O.rel_l2(out, ref_out) < 5 * TOL


@pytest.mark.parametrize("name", G.big_fixture_names())
def test_big_config_oracle_matches_reference(name):
    """BASELINE configs C1 (2-D 64 x 64, M = 8) and C5's widths (3-D, M = 128) at dim 256 / 8 heads x 32."""
    x, sd, meta, exp, rows = G.load_big(name)
    cfg = G.oracle_cfg_from_meta(meta)
    trace = {}
    with torch.no_grad():
        out = O.physics_attention_forward(x, sd, cfg, trace=trace)
    for k in ("tree_idx", "tree_mask", "rot0", "rot1"):
        assert torch.equal(trace["erwin." + k], exp["erwin." + k]), k
    assert O.rel_l2(trace["x_lin"].index_select(1, rows), exp["x_lin_rows"]) < TOL
    assert O.rel_l2(trace["eidetic"].reshape(exp["eidetic_flat"].shape), exp["eidetic_flat"]) < TOL
    assert O.rel_l2(trace["processed"].reshape(exp["processed_flat"].shape), exp["processed_flat"]) < 5 * TOL
    assert O.rel_l2(out.index_select(1, rows), exp["out_rows"]) < 5 * TOL
    assert abs(float(out.double().norm()) / float(exp["out_norm"]) - 1.0) < 5 * TOL


@pytest.mark.parametrize("name", G.grad_fixture_names())
def test_oracle_gradients_match_reference_autograd(name):
    """Autograd through the oracle against the REFERENCE module's own autograd (tests/golden/grad_*.npz): input
    and every parameter.  Pins, among the rest, that token positions carry no gradient (attention.py:35,
    pooling.py:29,59 are no_grad in the reference) and which parameters get none at all (sigma_att)."""
    x, sd, meta, exp, rows = G.load_attn(name)
    gout, gx_ref, g_ref, nograd = G.load_grad(name)
    cfg = G.oracle_cfg_from_meta(meta)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
    full = dict(sd)
    full.update(leaves)
    xr = x.clone().requires_grad_(True)
    out = O.physics_attention_forward(xr, full, cfg, training=meta["train"])
    names = [k for k in leaves if k in g_ref]
    got = torch.autograd.grad((out * gout).sum(), [xr] + [leaves[k] for k in names], allow_unused=True)
    assert O.rel_l2(got[0], gx_ref) < 2e-4
    scale = max(float(g.norm()) for g in g_ref.values())
    for k, g in zip(names, got[1:]):
        assert g is not None, k
        err = float((g - g_ref[k]).norm()) / max(float(g_ref[k].norm()), 1e-4 * scale)
        assert err < 5e-4, (k, err)
    assert set(nograd) == {k for k in leaves if k not in g_ref}, nograd


@pytest.mark.parametrize("name", G.ef_fixture_names())
def test_standalone_erwinflash_oracle_matches_reference(name):
    """Stand-alone ErwinFlash on raw clouds: embedding + MPNN + padded trees of unequal clouds, against
    the imported reference's output; the radius graph against the edges the fixture was generated with."""
    f = G.load_ef(name)
    kw, meta, exp = f["meta"]["kw"], f["meta"], f["exp"]
    trace = {}
    out = O.erwinflash_forward(f["feat"], f["pos"], f["batch_idx"], f["sd"], G.ef_cfg(meta), kw["mp_steps"],
                               edge_index=f["edge_index"] if meta["edges"] == "given" else None, radius=meta["radius"],
                               training=meta["train"], trace=trace)
    assert torch.equal(trace["tree_idx"], exp["tree_idx"]) and torch.equal(trace["tree_mask"], exp["tree_mask"])
    if f["edge_index"] is not None:
        assert torch.equal(trace["edge_index"], f["edge_index"])
        assert torch.equal(O.radius_graph(f["pos"], meta["radius"], f["batch_idx"]), f["edge_index"])
    if "embed" in exp:
        assert O.rel_l2(trace["embed"], exp["embed"]) < 1e-6
    assert O.rel_l2(out, exp["out"]) < 1e-5


@pytest.mark.parametrize("name", ["two_clouds", "two_clouds_train", "planar_norot"])
def test_standalone_erwinflash_oracle_gradients_match_reference_autograd(name):
    """Autograd through the oracle's stand-alone ErwinFlash against the REFERENCE ErwinTransformer's own autograd
    (tests/golden/efgrad_*.npz): node features and every parameter (sigma_att gets none).  Parameters whose true
    gradient is zero (a bias in front of a train-mode BatchNorm) are held to the fp32 noise of the largest gradient."""
    f, gr = G.load_ef(name), G.load_ef_grad(name)
    meta, kw = f["meta"], f["meta"]["kw"]
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and "running_" not in k) for k, v in f["sd"].items()}
    feat = f["feat"].clone().requires_grad_(True)
    out = O.erwinflash_forward(feat, f["pos"], f["batch_idx"], sd, G.ef_cfg(meta), kw["mp_steps"],
                               edge_index=f["edge_index"] if meta["edges"] == "given" else None, radius=meta["radius"],
                               training=meta["train"])
    (out * gr["gout"]).sum().backward()
    assert O.rel_l2(feat.grad, gr["gfeat"]) < 2e-5
    scale = max(float(g.norm()) for g in gr["g"].values())
    for k, g in gr["g"].items():
        err = float((sd[k].grad - g).norm()) / max(float(g.norm()), 1e-3 * scale)
        assert err < 5e-5, (k, err)
    assert set(gr["nograd"]) == {k for k, v in sd.items() if v.requires_grad and v.grad is None}
