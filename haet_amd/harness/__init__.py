"""Harness adjacency of the hot path: the reference's own forward-time / peak-memory sweep
(benchmarks/00-VelocityMemMocker/velocity-mem-checker.py), the preprocessed-ShapeNetCar sample format its car
benchmarks train on (benchmarks/02-Car-Design-ShapeNetCar/dataset/dataset.py:155-159,243-287) and the evaluation
metrics they report (main_evaluation.py:262-351) -- so that the published numbers can be
re-measured on this package's modules with real meshes.  No VTK, torch_geometric, wandb: the raw-VTK preprocessing,
the training loop and the visualisation stay with the reference's scripts."""
from .metrics import CarEvaluation, rel_l2, spearman_rho  # noqa: F401
from .shapenetcar import CarSample, load_sample, load_samples, model_inputs, normalisation  # noqa: F401
from .sweep import SWEEP_POINTS, benchmark_model, sweep  # noqa: F401
