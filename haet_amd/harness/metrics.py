"""Evaluation metrics of the car benchmarks (benchmarks/02-Car-Design-ShapeNetCar/main_evaluation.py:262-351):
per-sample relative L2 of the surface pressure and of the volume velocity after de-normalisation, their RMSEs,
Spearman's rank correlation of predicted against true drag coefficients and the mean relative coefficient error.
(The drag coefficient itself integrates pressure and friction over the VTK surface mesh -- `cal_coefficient` of the
reference's drag_coefficient.py, which needs the raw meshes -- so coefficients are passed in.)"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch


def rel_l2(pred: torch.Tensor, target: torch.Tensor) -> float:
    """||pred - target||_2 / ||target||_2 (main_evaluation.py:311-312; utils/testloss.py:31-45 of benchmark 04)."""
    return float(torch.linalg.vector_norm(pred.double() - target.double()) / torch.linalg.vector_norm(target.double()))


def spearman_rho(a, b) -> float:
    """Spearman rank correlation with average ranks for ties (scipy.stats.spearmanr, main_evaluation.py:343)."""
    def ranks(v):
        v = np.asarray(v, dtype=np.float64)
        order = np.argsort(v, kind="mergesort")
        r = np.empty(len(v))
        r[order] = np.arange(1, len(v) + 1)
        for val in np.unique(v):  # ties share the mean of their ranks
            m = v == val
            if m.sum() > 1:
                r[m] = r[m].mean()
        return r

    ra, rb = ranks(a), ranks(b)
    ra, rb = ra - ra.mean(), rb - rb.mean()
    return float((ra * rb).sum() / np.sqrt((ra ** 2).sum() * (rb ** 2).sum()))


class CarEvaluation:
    """Accumulates the per-sample numbers of main_evaluation.py:275-351 and reports their summary."""

    def __init__(self, coef_norm=None):
        self.mean = None if coef_norm is None else torch.as_tensor(np.asarray(coef_norm[2])).double()
        self.std = None if coef_norm is None else torch.as_tensor(np.asarray(coef_norm[3])).double()
        self.l2_press: List[float] = []
        self.l2_velo: List[float] = []
        self.mse_press: List[float] = []
        self.mse_velo: List[np.ndarray] = []
        self.times: List[float] = []
        self.cd_true: List[float] = []
        self.cd_pred: List[float] = []

    def add(self, out: torch.Tensor, target: torch.Tensor, surf: torch.Tensor, seconds: Optional[float] = None,
            cd_true: Optional[float] = None, cd_pred: Optional[float] = None):
        """out / target [N, 4] = velocity(3) | pressure(1), NORMALISED as the model sees them; surf [N] bool."""
        out, target = out.detach().double().cpu(), target.detach().double().cpu()
        surf = surf.cpu().bool()
        den = (lambda v, sl: v * self.std[sl] + self.mean[sl]) if self.mean is not None else (lambda v, sl: v)
        self.l2_press.append(rel_l2(den(out[surf, -1], -1), den(target[surf, -1], -1)))
        self.l2_velo.append(rel_l2(den(out[~surf, :-1], slice(None, -1)), den(target[~surf, :-1], slice(None, -1))))
        self.mse_press.append(float(((out[surf, -1] - target[surf, -1]) ** 2).mean()))
        self.mse_velo.append(((out[~surf, :-1] - target[~surf, :-1]) ** 2).mean(dim=0).numpy())
        if seconds is not None:
            self.times.append(seconds)
        if cd_true is not None and cd_pred is not None:
            self.cd_true.append(float(cd_true))
            self.cd_pred.append(float(cd_pred))

    def summary(self) -> dict:
        rmse_p = float(np.sqrt(np.mean(self.mse_press)))
        rmse_v = np.sqrt(np.mean(np.stack(self.mse_velo), axis=0))
        if self.std is not None:  # back to physical units (main_evaluation.py:351-353)
            rmse_p *= float(self.std[-1])
            rmse_v = rmse_v * self.std[:-1].numpy()
        res = {"rel_l2_press": float(np.mean(self.l2_press)), "rel_l2_velo": float(np.mean(self.l2_velo)),
               "rmse_press": rmse_p, "rmse_velo": rmse_v.tolist(), "rmse_velo_all": float(np.sqrt(np.mean(rmse_v ** 2))),
               "samples": len(self.l2_press)}
        if self.times:
            res["mean_inference_s"] = float(np.mean(self.times))
        if len(self.cd_true) > 1:
            t, p = np.asarray(self.cd_true), np.asarray(self.cd_pred)
            res["rho_d"] = spearman_rho(t, p)
            res["c_d_error"] = float(np.mean(np.abs(p - t) / t))
        return res
