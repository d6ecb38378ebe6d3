"""CPU: host-side mirror of the reference interface -- constructor surface, state_dict keys
(checked against the state_dicts the reference itself produced, stored in the fixtures),
loud failure without the HIP path, weight packing."""
import pytest
import torch

import haet_amd
from haet_amd.PhysicsAttention import (Physics_Attention_Irregular_Mesh,
                                       Physics_Attention_Structured_Mesh_2D,
                                       Physics_Attention_Structured_Mesh_3D)
from haet_amd._lib import HaetError
from tests import golden_util as G

CTOR = {"irregular": Physics_Attention_Irregular_Mesh, "structured2d": Physics_Attention_Structured_Mesh_2D,
        "structured3d": Physics_Attention_Structured_Mesh_3D}


@pytest.mark.parametrize("name", G.attn_fixture_names())
def test_attention_state_dict_matches_reference(name):
    x, sd, meta, exp, rows = G.load_attn(name)
    mod = CTOR[meta["kind"]](**meta["kw"])
    mod.load_state_dict(sd, strict=True)  # identical keys and shapes


@pytest.mark.parametrize("name", G.model_fixture_names())
def test_model_state_dict_matches_reference(name):
    x, fx, sd, meta, out = G.load_model(name)
    model = getattr(haet_amd, meta["cls"])(**meta["kw"])
    model.load_state_dict(sd, strict=True)


def test_default_erwin_hyperparameters():
    m = Physics_Attention_Irregular_Mesh(256, heads=8, dim_head=32, slice_num=64, rotate=45)
    assert m.erwin.ball_sizes == [32, 16] and m.erwin.strides == [2]
    assert m.erwin.encoder[0].num_heads == 4 and m.erwin.bottleneck.num_heads == 8
    assert m.erwin.c_hidden == [32, 64] and m.dimensionality == 3
    assert Physics_Attention_Structured_Mesh_2D(64, heads=4, dim_head=16, H=8, W=8).dimensionality == 2


def test_no_cpu_fallback():
    m = Physics_Attention_Irregular_Mesh(64, heads=4, dim_head=16, slice_num=8).eval()
    with torch.no_grad(), pytest.raises(HaetError):
        m(torch.randn(1, 100, 64))


def test_no_cpu_fallback_with_gradients_or_in_train_mode_either():
    m = Physics_Attention_Irregular_Mesh(64, heads=4, dim_head=16, slice_num=8)
    with pytest.raises(HaetError):
        m.eval()(torch.randn(1, 100, 64))          # gradients enabled: still no CPU path
    with pytest.raises(HaetError):
        m.train()(torch.randn(1, 100, 64))


def test_embedding_variants_are_out_of_scope():
    with pytest.raises(NotImplementedError):
        Physics_Attention_Irregular_Mesh(64, heads=4, dim_head=16, slice_num=8, embed=True)


def test_pack_erwin_layout():
    from haet_amd.components.erwinflash.pack import pack_erwin

    m = Physics_Attention_Irregular_Mesh(256, heads=8, dim_head=32, slice_num=64, rotate=45).eval()
    blob, desc = pack_erwin(m.erwin)
    assert desc.n_levels == 2 and desc.d == 3 and desc.rotate == 1 and desc.decode == 1
    e0, e1, d0 = desc.enc[0], desc.enc[1], desc.dec[0]
    assert (e0.c, e0.heads, e0.ball, e0.depth, e0.hidden) == (32, 4, 32, 2, 128)
    assert (e1.c, e1.heads, e1.ball, e1.depth, e1.hidden) == (64, 8, 16, 2, 256)
    assert (d0.c, d0.heads, d0.ball, d0.depth) == (32, 4, 32, 2)
    # blob starts with a copy of the descriptor
    import ctypes
    from haet_amd.components.erwinflash.pack import HEADER_FLOATS
    assert bytes(blob[:HEADER_FLOATS].numpy().tobytes()[:ctypes.sizeof(desc)]) == bytes(desc)
    # Linear: [out][in~32] bf16 hi plane, then lo plane; hi + lo reproduces the fp32 weight to ~2^-16
    blk = m.erwin.encoder[0].blocks[1]
    w = e0.blocks[1]
    planes = blob[w.qkv_w:w.qkv_w + 96 * 32].view(torch.bfloat16).view(2, 96, 32).float()
    assert torch.allclose(planes[0] + planes[1], blk.BMSA.qkv.weight, rtol=2e-5, atol=1e-9)
    assert w.qkv_w % 4 == 0 and w.w3_w % 4 == 0
    pool = m.erwin.encoder[0].pool  # K = 70 -> padded to 96
    pw = blob[desc.pool[0].w:desc.pool[0].w + 64 * 96].view(torch.bfloat16).view(2, 64, 96).float()
    assert torch.allclose(pw[0, :, :70] + pw[1, :, :70], pool.proj.weight, rtol=2e-5, atol=1e-9)
    assert (pw[:, :, 70:] == 0).all()
    assert torch.equal(blob[w.pe_w:w.pe_w + 3 * 32].view(3, 32), blk.BMSA.pe_proj.weight.t())
    from haet_amd import _lib
    import ctypes
    assert 0 < _lib.load().haet_erwin_lds_bytes(ctypes.byref(desc), 64) < 160 * 1024


@pytest.mark.parametrize("kw", [
    dict(dim=64, heads=4, dim_head=16, slice_num=16, rotate=45),
    dict(dim=256, heads=8, dim_head=32, slice_num=64, rotate=45),
    dict(dim=128, heads=4, dim_head=64, slice_num=32, rotate=0),
    dict(dim=64, heads=4, dim_head=16, slice_num=12, rotate=45, ball_sizes=[16, 8]),
    dict(dim=64, heads=4, dim_head=16, slice_num=16, rotate=45, strides=[1], c_hidden=[16, 16], ball_sizes=[16, 8]),
])
def test_pack_plans_equal_the_direct_packers_bit_for_bit(kw):
    """What the modules pack with since round 4 -- the blob layouts as gather plans (pack.py ErwinPackPlan /
    ErwinF32PackPlan: ~8 launches per repack instead of ~400, the training step of the reference's PDE configurations was
    host-bound on the repack) -- against the direct packers: same bits, same descriptor / layout / gradient map, in both
    modes, and again after the weights and the BatchNorm statistics changed (the plan is reused, the blob follows)."""
    from haet_amd.components.erwinflash import pack

    torch.manual_seed(1)
    er = Physics_Attention_Irregular_Mesh(**kw).erwin
    plans = {}
    for round_ in range(2):
        with torch.no_grad():
            for b in er.buffers():
                if b.dtype.is_floating_point:
                    b.uniform_(0.5, 1.5)
            for q in er.parameters():
                q.add_(torch.randn_like(q) * 0.01)
        for mode in (True, False):
            er.train(mode)
            blob, desc = pack.pack_erwin(er)
            got, gdesc = er.packed()
            assert torch.equal(blob.view(torch.int32), got.view(torch.int32))
            assert bytes(desc) == bytes(gdesc)
            plan = er.__dict__["_plans_hit"][1][("fwd", mode)]
            assert plans.setdefault(mode, plan) is plan  # (same parameter objects: the layout is not recomputed)
        b32, lay, pmap = pack.pack_erwin_f32(er)
        g32, glay, gpmap = er.packed_f32()
        assert torch.equal(b32.view(torch.int32), g32.view(torch.int32))
        assert bytes(lay) == bytes(glay) and [(id(q), o) for q, o in pmap] == [(id(q), o) for q, o in gpmap]


def test_rot_targets_use_total_leaf_count():
    from haet_amd.ops import rot_targets, tree_slots

    assert rot_targets(512, [32, 16]) == [4, 5]  # C3: 8 clouds x 64 slots (balltree.pyx:643 quirk)
    assert tree_slots(64) == 64 and tree_slots(12) == 16 and tree_slots(5) == 8 and tree_slots(2) == 2


def test_weight_version_key_sees_every_kind_of_update():
    """The packed Erwin weights are rebuilt when a parameter changes; the key walks a CACHED tensor
    list, so every way a tensor can change or be replaced has to invalidate it."""
    m = Physics_Attention_Irregular_Mesh(64, heads=4, dim_head=16, slice_num=8).eval()
    er = m.erwin
    k0 = er._pack_key()
    assert er._pack_key() == k0
    # in-place update (optimizer step, copy_): same tensor object, new version
    with torch.no_grad():
        er.encoder[0].blocks[0].norm1.weight.mul_(1.5)
    k1 = er._pack_key()
    assert k1 != k0
    # load_state_dict on an ANCESTOR with assign=True replaces the tensor objects
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.load_state_dict(sd, assign=True)
    k2 = er._pack_key()
    assert k2 != k1 and {p for p, _ in k2} != {p for p, _ in k1}
    # a dtype/device move goes through _apply
    m.double()
    k3 = er._pack_key()
    assert k3 != k2
    m.float()
    # a parameter (or buffer) re-assigned by hand on a SUBMODULE is seen at the very next call: torch's global
    # registration hooks bump the cache generation (haet_amd/_lib.py)
    er._pack_key()
    er.encoder[0].blocks[0].norm1.weight = torch.nn.Parameter(torch.ones(16))
    new_ptr = er.encoder[0].blocks[0].norm1.weight.data_ptr()
    assert any(p == new_ptr for p, _ in er._pack_key())
    er.encoder[0].pool.norm.running_mean = torch.zeros(32) + 0.25
    new_ptr = er.encoder[0].pool.norm.running_mean.data_ptr()
    assert any(p == new_ptr for p, _ in er._pack_key())
    # in-place writes through .data change neither _version nor the pointer: invalidate_caches() is the documented way
    k4 = er._pack_key()
    er.encoder[0].blocks[0].norm2.weight.data.mul_(2.0)
    assert er._pack_key() == k4
    m.invalidate_caches()
    assert er._packed is None and m._slice_cache is None


def test_grad_check_cache_follows_requires_grad():
    m = Physics_Attention_Irregular_Mesh(64, heads=4, dim_head=16, slice_num=8).eval()
    assert m._any_requires_grad()
    for p in m.parameters():
        p.requires_grad_(False)
    assert not m._any_requires_grad()
    m.to_out[0].weight.requires_grad_(True)
    assert m._any_requires_grad()


def test_modules_round_trip_through_pickle_and_deepcopy():
    """torch.save(module) / copy.deepcopy (spawn-based multiprocessing passes models this way): the hooks are
    module-level functions, per-process caches (ctypes descriptors, device scratch, weak references) are dropped and
    re-made, the copy gets its own haet::module_forward handle and the same state_dict."""
    import copy
    import io

    import haet_amd
    from haet_amd.HAETransolver_Irregular_Mesh import Model

    m = haet_amd.PhysicsAttention.Physics_Attention_Irregular_Mesh(64, heads=4, dim_head=16, slice_num=16, rotate=45)
    blob, desc = m.erwin.packed()              # fill the caches that hold ctypes objects
    m.erwin._train_state(desc, 4, torch.device("cpu"))
    mm = Model(space_dim=3, n_layers=1, n_hidden=64, n_head=4, fun_dim=0, out_dim=1, slice_num=16)
    for mod in (m, mm):
        buf = io.BytesIO()
        torch.save(mod, buf)
        buf.seek(0)
        for twin in (torch.load(buf, weights_only=False), copy.deepcopy(mod)):
            if mod is m:
                assert twin._haet_handle != mod._haet_handle
            else:
                assert twin._pre_handles != mod._pre_handles and twin.blocks[0]._haet_handle != mod.blocks[0]._haet_handle
            a, b = mod.state_dict(), twin.state_dict()
            assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
    assert copy.deepcopy(m).erwin._packed is None


def test_grid_barrier_timeout_word_surfaces_as_error():
    """A train-mode launch whose grid barrier gave up sets the workspace's sticky time-out word (include/haet_hip.h,
    residency contract); the host raises at its next check and re-arms the word."""
    import haet_amd
    from haet_amd import _lib, ops

    m = haet_amd.PhysicsAttention.Physics_Attention_Irregular_Mesh(64, heads=4, dim_head=16, slice_num=16, rotate=45)
    _, desc = m.erwin.packed()
    st = ops.ErwinTrainState(m.erwin, desc, 4, torch.device("cpu"))
    st.check_barrier()                                   # clean
    st.workspace.view(torch.int32)[0] = 12               # an arrival count is not an error
    st.check_barrier()
    st.workspace.view(torch.int32)[1] = 1                # what erwin_grid_barrier / bwd_grid_barrier store on time-out
    with pytest.raises(_lib.HaetError, match="grid barrier"):
        st.check_barrier()
    st.check_barrier()                                   # re-armed
    assert 1 <= _lib.load().haet_resident_workgroup_limit() <= 4096


def test_structured_3d_model_defaults_follow_reference():
    """HAETransolver_Structured_Mesh_3D.py:243: gradient checkpointing is off by default (with it on, every block's
    forward runs twice per training step and train-mode BatchNorm statistics advance twice)."""
    from haet_amd.HAETransolver_Structured_Mesh_3D import Model

    assert Model(n_layers=1, n_hidden=64, n_head=4, H=4, W=4, D=4, slice_num=8).use_checkpoint is False

