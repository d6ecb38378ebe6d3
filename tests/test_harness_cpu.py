"""CPU: the harness adjacency -- the cached-ShapeNetCar sample format and its normalisation against
the reference's streaming formulas (benchmarks/02-Car-Design-ShapeNetCar/dataset/dataset.py:243-287, restated here in
numpy exactly as written there), and the evaluation metrics against scipy / hand-computed values."""
import os

import numpy as np
import pytest
import torch

from haet_amd import harness as Hn


def _write_samples(root, sizes, seed=0):
    rng = np.random.default_rng(seed)
    names = []
    for i, n in enumerate(sizes):
        d = os.path.join(root, f"param{i}", f"s{i}")
        os.makedirs(d)
        np.save(os.path.join(d, "x.npy"), rng.normal(size=(n, 7)) * (1 + i) + i)
        np.save(os.path.join(d, "y.npy"), rng.normal(size=(n, 4)) * 3 - i)
        np.save(os.path.join(d, "pos.npy"), rng.random((n, 3)))
        np.save(os.path.join(d, "surf.npy"), (rng.random(n) < 0.3).astype(np.float64))
        np.save(os.path.join(d, "edge_index.npy"), rng.integers(0, n, (2, 5 * n)))
        names.append(f"param{i}/s{i}")
    return names


def _reference_streaming_stats(xs, ys):
    """dataset.py:243-278 as written: running means over the samples, then running mean squared deviations."""
    for k, (init, target) in enumerate(zip(xs, ys)):
        if k == 0:
            old_length = init.shape[0]
            mean_in, mean_out = init.mean(axis=0), target.mean(axis=0)
        else:
            new_length = old_length + init.shape[0]
            mean_in = mean_in + (init.sum(axis=0) - init.shape[0] * mean_in) / new_length
            mean_out = mean_out + (target.sum(axis=0) - init.shape[0] * mean_out) / new_length
            old_length = new_length
    for k, (x, y) in enumerate(zip(xs, ys)):
        if k == 0:
            old_length = x.shape[0]
            std_in = ((x - mean_in) ** 2).sum(axis=0) / old_length
            std_out = ((y - mean_out) ** 2).sum(axis=0) / old_length
        else:
            new_length = old_length + x.shape[0]
            std_in = std_in + (((x - mean_in) ** 2).sum(axis=0) - x.shape[0] * std_in) / new_length
            std_out = std_out + (((y - mean_out) ** 2).sum(axis=0) - x.shape[0] * std_out) / new_length
            old_length = new_length
    return mean_in, np.sqrt(std_in), mean_out, np.sqrt(std_out)


def test_sample_format_and_normalisation(tmp_path):
    names = _write_samples(str(tmp_path), [50, 31, 77])
    raw = Hn.load_samples(str(tmp_path), names + ["missing/one"])
    assert len(raw) == 3 and raw[1].x.shape == (31, 7) and raw[1].surf.dtype == torch.bool and raw[1].edge_index.shape[0] == 2
    samples, coef = Hn.load_samples(str(tmp_path), names, norm=True)
    ref = _reference_streaming_stats([s.x.numpy() for s in raw], [s.y.numpy() for s in raw])
    for a, b in zip(coef, ref):
        assert np.allclose(a, b, rtol=1e-10, atol=1e-12)
    allx = torch.cat([s.x for s in samples])
    assert samples[0].x.dtype == torch.float32
    assert torch.allclose(allx.mean(0), torch.zeros(7), atol=1e-5) and torch.allclose(allx.std(0, unbiased=False), torch.ones(7), atol=1e-4)
    again = Hn.load_samples(str(tmp_path), names, coef_norm=coef)  # a validation split uses the training statistics
    assert torch.equal(again[2].y, samples[2].y)


def test_metrics_match_scipy_and_hand_values():
    import scipy.stats

    rng = np.random.default_rng(1)
    a, b = rng.normal(size=40), rng.normal(size=40)
    b[:5] = b[5:10]  # ties
    assert abs(Hn.spearman_rho(a, b) - scipy.stats.spearmanr(a, b)[0]) < 1e-12
    assert abs(Hn.rel_l2(torch.tensor([3.0, 4.0]), torch.tensor([0.0, 4.0])) - 0.75) < 1e-12
    coef = (np.zeros(7), np.ones(7), np.array([1.0, 2.0, 3.0, 4.0]), np.array([2.0, 2.0, 2.0, 5.0]))
    ev = Hn.CarEvaluation(coef)
    out, tgt = torch.zeros(6, 4), torch.ones(6, 4)
    surf = torch.tensor([1, 1, 0, 0, 0, 0]).bool()
    ev.add(out, tgt, surf, seconds=0.1, cd_true=0.30, cd_pred=0.33)
    ev.add(out, tgt, surf, seconds=0.3, cd_true=0.40, cd_pred=0.36)
    s = ev.summary()
    # pressure: pred = 4, true = 9 on both surface points -> |5| / |9|
    assert abs(s["rel_l2_press"] - 5.0 / 9.0) < 1e-12
    # velocity: pred = (1,2,3), true = (3,4,5) -> sqrt(12) / sqrt(50)
    assert abs(s["rel_l2_velo"] - (12.0 / 50.0) ** 0.5) < 1e-12
    assert abs(s["rmse_press"] - 5.0) < 1e-12 and np.allclose(s["rmse_velo"], [2.0, 2.0, 2.0])
    assert abs(s["mean_inference_s"] - 0.2) < 1e-12 and abs(s["rho_d"] - 1.0) < 1e-12
    assert abs(s["c_d_error"] - 0.1) < 1e-12


def test_sweep_sizes_are_the_references():
    assert Hn.SWEEP_POINTS == (1000, 10000, 100000, 1000000, 2000000, 3000000)  # velocity-mem-checker.py:82
