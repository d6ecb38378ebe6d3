"""Preprocessed ShapeNetCar samples -- the on-disk format the reference's car benchmarks cache and train from
(benchmarks/02-Car-Design-ShapeNetCar/dataset/dataset.py:155-159: one directory per sample holding x.npy [N, 7] =
position(3) | signed distance(1) | normal(3), y.npy [N, 4] = velocity(3) | pressure(1), pos.npy [N, 3], surf.npy [N]
(1 on the car surface), edge_index.npy [2, E]) and the normalisation applied to them (:243-287: per-channel mean and
standard deviation pooled over ALL points of the training samples, `(v - mean) / (std + 1e-8)`).

Only the cached format is read; producing it from the raw VTK files (dataset.py:161-236) needs vtk and stays with the
reference's preprocessing script."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

FILES = ("x", "y", "pos", "surf", "edge_index")


@dataclass
class CarSample:
    x: torch.Tensor           # [N, 7] position | sdf | normal
    y: torch.Tensor           # [N, 4] velocity | pressure
    pos: torch.Tensor         # [N, 3]
    surf: torch.Tensor        # [N] bool, True on the car surface
    edge_index: torch.Tensor  # [2, E] int64
    name: str = ""


def load_sample(path: str, mmap: bool = False) -> CarSample:
    """One cached sample directory (dataset.py:155-159)."""
    arr = {k: np.load(os.path.join(path, k + ".npy"), mmap_mode="r" if mmap else None) for k in FILES}
    return CarSample(x=torch.as_tensor(np.asarray(arr["x"])), y=torch.as_tensor(np.asarray(arr["y"])),
                     pos=torch.as_tensor(np.asarray(arr["pos"])), surf=torch.as_tensor(np.asarray(arr["surf"])).bool(),
                     edge_index=torch.as_tensor(np.asarray(arr["edge_index"])).long(), name=os.path.basename(path.rstrip("/")))


def normalisation(samples: Sequence[CarSample]) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """(mean_in, std_in, mean_out, std_out) over all points of `samples`: what dataset.py:243-278 accumulates sample by
    sample (running means, then running mean squared deviations) -- the pooled population statistics."""
    n = sum(s.x.shape[0] for s in samples)
    mean_in = sum(s.x.double().sum(0) for s in samples) / n
    mean_out = sum(s.y.double().sum(0) for s in samples) / n
    var_in = sum(((s.x.double() - mean_in) ** 2).sum(0) for s in samples) / n
    var_out = sum(((s.y.double() - mean_out) ** 2).sum(0) for s in samples) / n
    return mean_in.numpy(), var_in.sqrt().numpy(), mean_out.numpy(), var_out.sqrt().numpy()


def load_samples(root: str, names: Sequence[str], norm: bool = False, coef_norm=None):
    """`get_datalist(..., preprocessed=True)` (dataset.py:143-289): the samples under `root` that exist, normalised
    with `coef_norm` when given, or with their own pooled statistics when `norm` (then returns (samples, coef_norm)),
    features as float32."""
    out: List[CarSample] = []
    for nm in names:
        p = os.path.join(root, nm)
        if os.path.exists(p):
            out.append(load_sample(p))
    own = norm and coef_norm is None
    if own:
        coef_norm = normalisation(out)
    if coef_norm is not None:
        mi, si, mo, so = (torch.as_tensor(np.asarray(c)) for c in coef_norm)
        for s in out:
            s.x = ((s.x - mi) / (si + 1e-8)).float()
            s.y = ((s.y - mo) / (so + 1e-8)).float()
    return (out, coef_norm) if own else out


def model_inputs(sample: CarSample, device="cuda"):
    """(x [1, N, 7], None): how the car benchmark feeds HAETransolver_Irregular_Mesh(space_dim=7, fun_dim=0) -- the
    whole feature row as coordinates, no separate function values (models/Transolver.py of that benchmark)."""
    return sample.x.float().unsqueeze(0).to(device), None
