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
