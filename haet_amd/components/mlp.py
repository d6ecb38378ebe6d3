"""Generic MLP used by the model wrappers (mirrors models/components/mlp.py:23-96).  Stock PyTorch-ROCm
modules; with gradients disabled the GELU variant runs on the HIP GEMM (haet_amd/_fast.py)."""
import torch.nn as nn

from .. import _fast

ACTIVATION = {
    "gelu": nn.GELU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid, "relu": nn.ReLU,
    "leaky_relu": nn.LeakyReLU(0.1), "softplus": nn.Softplus, "ELU": nn.ELU, "silu": nn.SiLU,
}


class MLP(nn.Module):
    def __init__(self, n_input, n_hidden, n_output, n_layers=1, act="gelu", res=True):
        super().__init__()
        if act not in ACTIVATION:
            raise NotImplementedError
        act = ACTIVATION[act]
        self.n_input, self.n_hidden, self.n_output = n_input, n_hidden, n_output
        self.n_layers, self.res = n_layers, res
        self.linear_pre = nn.Sequential(nn.Linear(n_input, n_hidden), act())
        self.linear_post = nn.Linear(n_hidden, n_output)
        self.linears = nn.ModuleList(
            [nn.Sequential(nn.Linear(n_hidden, n_hidden), act()) for _ in range(n_layers)])

    def forward(self, x):
        y = _fast.mlp(self, x)  # inference fast path (None: does not apply)
        if y is not None:
            return y
        x = self.linear_pre(x)
        for i in range(self.n_layers):
            x = self.linears[i](x) + x if self.res else self.linears[i](x)
        return self.linear_post(x)
