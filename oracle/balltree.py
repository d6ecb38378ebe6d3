"""TEST INFRASTRUCTURE ONLY -- ctypes front-ends for the CPU ball-tree oracles.

* ``build_balltree`` / ``partition_balltree`` drive ``liboracle_balltree.so``
  (our plain-C restatement, oracle/balltree_oracle.c).
* ``ref_build_balltree`` / ``ref_partition_balltree`` drive
  ``oracle/_ref/libballtree_ref.so`` -- the reference's own ``balltree.h``
  compiled where it lies (oracle/Makefile ``ref``) -- when that file exists.

Both follow the reference's Python-visible contract
(models/components/balltree/balltree.pyx:424-574): ``data`` is [T, d] (any float
dtype, converted to float64 exactly as balltree.pyx:542-543 does), ``batch_idx``
is a sorted int vector starting at 0; results are int64 / bool numpy arrays.

Nothing under haet_amd/ may import this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

_i64p = ctypes.POINTER(ctypes.c_int64)
_f64p = ctypes.POINTER(ctypes.c_double)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build_oracle_lib(force: bool = False) -> str:
    """Compile oracle/balltree_oracle.c with gcc (seconds) if missing/stale."""
    src = os.path.join(_HERE, "balltree_oracle.c")
    so = os.path.join(_HERE, "liboracle_balltree.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(
            ["gcc", "-O2", "-std=c99", "-shared", "-fPIC", src, "-o", so, "-lm"]
        )
    return so


def _lib():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(build_oracle_lib())
        lib.oracle_tree_depth.restype = ctypes.c_int64
        lib.oracle_tree_depth.argtypes = [ctypes.c_int64]
        lib.oracle_build_balltree.restype = ctypes.c_int
        lib.oracle_build_balltree.argtypes = [_f64p, _i64p, ctypes.c_int64, ctypes.c_int64, _i64p, _u8p]
        lib.oracle_partition_balltree.restype = ctypes.c_int
        lib.oracle_partition_balltree.argtypes = [_f64p, _i64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _i64p]
        _LIB = lib
    return _LIB


def ref_available() -> bool:
    return os.path.exists(os.path.join(_HERE, "_ref", "libballtree_ref.so"))


def _ref():
    global _REF
    if _REF is None:
        lib = ctypes.CDLL(os.path.join(_HERE, "_ref", "libballtree_ref.so"))
        lib.ref_build_tree.restype = None
        lib.ref_build_tree.argtypes = [_f64p, _i64p, _i64p, _u8p, ctypes.c_long, ctypes.c_long, ctypes.c_long]
        lib.ref_partition_ball_tree.restype = None
        lib.ref_partition_ball_tree.argtypes = [_f64p, _i64p, _i64p, ctypes.c_long, ctypes.c_long, ctypes.c_long]
        _REF = lib
    return _REF


def offsets_from_batch_idx(batch_idx: np.ndarray) -> np.ndarray:
    """balltree.pyx:462-470: cloud boundaries = positions where batch_idx changes."""
    batch_idx = np.asarray(batch_idx)
    if batch_idx.ndim != 1 or len(batch_idx) == 0:
        raise ValueError("batch_idx must be a non-empty 1D array")
    if batch_idx[0] != 0:
        raise ValueError("Batch indices must start from 0")
    n = len(batch_idx)
    change = np.where(batch_idx[1:] != batch_idx[:-1])[0] + 1
    nb = int(batch_idx[-1]) + 1
    off = np.zeros(nb + 1, dtype=np.int64)
    off[1:nb] = change
    off[nb] = n
    return off


def tree_depth(n: int) -> int:
    return int(_lib().oracle_tree_depth(int(n)))


def _prep(data, batch_idx):
    data = np.ascontiguousarray(np.asarray(data, dtype=np.float64))
    if data.ndim != 2:
        raise ValueError("data must be [T, d]")
    off = offsets_from_batch_idx(batch_idx)
    if len(batch_idx) != data.shape[0]:
        raise ValueError("batch_idx length must match number of data points")
    return data, off


def build_balltree(data, batch_idx):
    data, off = _prep(data, batch_idx)
    nb = len(off) - 1
    total = sum(2 << tree_depth(int(off[b + 1] - off[b])) for b in range(nb))
    out = np.empty(total, dtype=np.int64)
    mask = np.empty(total, dtype=np.uint8)
    rc = _lib().oracle_build_balltree(
        data.ctypes.data_as(_f64p), off.ctypes.data_as(_i64p), nb, data.shape[1],
        out.ctypes.data_as(_i64p), mask.ctypes.data_as(_u8p))
    if rc != 0:
        raise RuntimeError(f"oracle_build_balltree failed: {rc}")
    return out, mask.astype(bool)


def partition_balltree(data, batch_idx, target_level: int):
    data, off = _prep(data, batch_idx)
    out = np.empty(data.shape[0], dtype=np.int64)
    rc = _lib().oracle_partition_balltree(
        data.ctypes.data_as(_f64p), off.ctypes.data_as(_i64p), len(off) - 1, data.shape[1],
        int(target_level), out.ctypes.data_as(_i64p))
    if rc != 0:
        raise RuntimeError(f"oracle_partition_balltree failed: {rc}")
    return out


# ---- the compiled reference (balltree.h) behind the same driver logic ---------
def ref_build_balltree(data, batch_idx):
    data, off = _prep(data, batch_idx)
    nf = data.shape[1]
    outs, masks = [], []
    for b in range(len(off) - 1):
        s, e = int(off[b]), int(off[b + 1])
        n = e - s
        depth = tree_depth(n)
        slots = 2 << depth
        idx = np.arange(n, dtype=np.int64)
        out = np.empty(slots, dtype=np.int64)
        mask = np.empty(slots, dtype=np.uint8)
        sub = np.ascontiguousarray(data[s:e])
        _ref().ref_build_tree(sub.ctypes.data_as(_f64p), idx.ctypes.data_as(_i64p),
                              out.ctypes.data_as(_i64p), mask.ctypes.data_as(_u8p), n, nf, depth)
        outs.append(out + s)
        masks.append(mask.astype(bool))
    return np.concatenate(outs), np.concatenate(masks)


def ref_partition_balltree(data, batch_idx, target_level: int):
    data, off = _prep(data, batch_idx)
    nf = data.shape[1]
    outs = []
    for b in range(len(off) - 1):
        s, e = int(off[b]), int(off[b + 1])
        n = e - s
        idx = np.arange(n, dtype=np.int64)
        out = np.empty(n, dtype=np.int64)
        sub = np.ascontiguousarray(data[s:e])
        _ref().ref_partition_ball_tree(sub.ctypes.data_as(_f64p), idx.ctypes.data_as(_i64p),
                                       out.ctypes.data_as(_i64p), n, nf, int(target_level))
        outs.append(out + s)
    return np.concatenate(outs)
