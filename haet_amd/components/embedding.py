"""Sinusoidal time embedding of the model wrappers (API surface only; stock PyTorch, outside the
accelerated path).  Same function as models/components/embedding.py:69-99 of the reference: output
[N, dim] = [cos(t * f_k) | sin(t * f_k)] with f_k = max_period^(-k / half), k < half = dim // 2, and one
zero column when dim is odd."""
import math

import torch


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    """`repeat_only` is accepted for signature compatibility; the reference ignores it as well."""
    n_freq = dim // 2
    exponent = torch.arange(n_freq, dtype=torch.float32, device=timesteps.device) * (-math.log(max_period) / n_freq)
    phase = timesteps.float().unsqueeze(-1) * exponent.exp()  # [..., n_freq]; T is [B] or [B, 1] (exp_plas.py:210-211)
    out = phase.new_zeros(phase.shape[:-1] + (dim,))
    out[..., :n_freq] = phase.cos()
    out[..., n_freq:2 * n_freq] = phase.sin()
    return out
