"""Helpers to read the committed golden fixtures (tests/golden/*.npz, made by
tests/golden/make_golden.py from the reference itself)."""
import ast
import glob
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def attn_fixture_names():
    return sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN, "attn_*.npz")))


def model_fixture_names():
    return sorted(os.path.basename(p)[6:-4] for p in glob.glob(os.path.join(GOLDEN, "model_*.npz")))


def big_fixture_names():
    return sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLDEN, "big_*.npz")))


def grad_fixture_names():
    return sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN, "grad_*.npz")))


def load_big(name):
    """BASELINE configs C1 / C5 at their real widths: x and the convolution weight are int8 * 2^-k (exact in
    fp32); per-point expectations (`x_lin_rows`, `out_rows`) cover the sampled `rows` only."""
    z = np.load(os.path.join(GOLDEN, f"big_{name}.npz"))
    meta = _meta(z)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    sd["in_project_x.weight"] = torch.from_numpy(z["w_i8"].astype(np.float32) / float(z["w_scale"]))
    exp = {k[4:]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith("exp/")}
    x = torch.from_numpy(z["x_i8"].astype(np.float32) / float(z["x_scale"]))
    return x, sd, meta, exp, torch.from_numpy(z["rows"])


def load_grad(name):
    """Gradients of the reference module for loss = <out, gout> on the inputs of attn_<name>.npz:
    (gout, gx, {param: grad}, [params the reference gives no gradient])."""
    z = np.load(os.path.join(GOLDEN, f"grad_{name}.npz"))
    grads = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("g/")}
    nograd = [k[7:] for k in z.files if k.startswith("nograd/")]
    return torch.from_numpy(z["gout"].astype(np.float32)), torch.from_numpy(z["gx"]), grads, nograd


def ef_fixture_names():
    return sorted(f[3:-4] for f in os.listdir(GOLDEN) if f.startswith("ef_") and f.endswith(".npz"))


def load_ef(name):
    """Stand-alone ErwinFlash fixture: dict with feat, pos, batch_idx, edge_index (or None), sd,
    sd_after (train: BatchNorm statistics after the forward), meta, exp."""
    z = np.load(os.path.join(GOLDEN, f"ef_{name}.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    for k in z.files:
        if k.startswith("sdq/"):  # matrices stored on an int8 * 2^-k grid
            sd[k[4:]] = torch.from_numpy(z[k].astype(np.float32) * float(z["sdscale/" + k[4:]]))
    return dict(feat=torch.from_numpy(z["feat"].astype(np.float32)), pos=torch.from_numpy(z["pos"]),
                batch_idx=torch.from_numpy(z["batch_idx"]),
                edge_index=torch.from_numpy(z["edge_index"].astype(np.int64)) if "edge_index" in z.files else None,
                sd=sd, sd_after={k[9:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_after/")},
                meta=_meta(z), exp={k[4:]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith("exp/")})


def load_ef_grad(name):
    """Gradients of the REFERENCE ErwinTransformer for loss = <out, gout> on fixture ef_<name> (make_golden.py
    gen_erwinflash_grad_fixtures): dict gout, gfeat, g {parameter name: gradient}, nograd [names]."""
    z = np.load(os.path.join(GOLDEN, f"efgrad_{name}.npz"))
    return dict(gout=torch.from_numpy(z["gout"].astype(np.float32)), gfeat=torch.from_numpy(z["gfeat"]),
                g={k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("g/")},
                nograd=[k[7:] for k in z.files if k.startswith("nograd/")])


def _meta(z):
    return ast.literal_eval(bytes(z["meta"]).decode())


def load_attn(name):
    z = np.load(os.path.join(GOLDEN, f"attn_{name}.npz"))
    meta = _meta(z)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    exp = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("exp/")}
    x = torch.from_numpy(z["x"].astype(np.float32))
    rows = torch.from_numpy(z["rows"])
    return x, sd, meta, exp, rows


def load_model(name):
    z = np.load(os.path.join(GOLDEN, f"model_{name}.npz"))
    meta = _meta(z)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    x = torch.from_numpy(z["x"])
    fx = torch.from_numpy(z["fx"]) if "fx" in z.files else None
    return x, fx, sd, meta, torch.from_numpy(z["out"])


def oracle_cfg_from_meta(meta):
    """AttnCfg of the oracle from the reference ctor kwargs stored in a fixture."""
    from oracle.haet_oracle import AttnCfg

    kw = dict(meta["kw"])
    kind = meta["kind"]
    grid = tuple(meta["grid"]) if meta["grid"] else ()
    defaults = {"irregular": dict(heads=8, dim_head=64, slice_num=64),
                "structured2d": dict(heads=8, dim_head=64, slice_num=64),
                "structured3d": dict(heads=8, dim_head=64, slice_num=32)}[kind]
    cfg = AttnCfg(dim=kw["dim"], heads=kw.get("heads", defaults["heads"]),
                  dim_head=kw.get("dim_head", defaults["dim_head"]),
                  slice_num=kw.get("slice_num", defaults["slice_num"]),
                  kind=kind, grid=grid, kernel=kw.get("kernel", 3),
                  rotate=kw.get("rotate", 1), mlp_ratio=kw.get("mlp_ratio", 4))
    for k in ("c_hidden", "ball_sizes", "enc_num_heads", "enc_depths", "dec_num_heads", "dec_depths", "strides"):
        if kw.get(k) is not None:
            setattr(cfg.erwin, k, kw[k])
    return cfg


def ef_cfg(meta):
    from oracle import haet_oracle as O

    kw = meta["kw"]
    return O.ErwinCfg(**{k: kw[k] for k in ("c_in", "c_hidden", "ball_sizes", "enc_num_heads", "enc_depths", "dec_num_heads",
                                            "dec_depths", "strides", "rotate", "decode", "mlp_ratio", "dimensionality")})
