"""GPU (-m gpu): the point-sharded module path end to end with TWO processes on one GPU
(gloo rendezvous; RCCL needs one device per rank, which the 8-GPU bench has).  Each rank owns
half of the points of a golden fixture, runs the HIP path with shard_group set and must
reproduce its half of the REFERENCE output."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from haet_amd.sharding import shard_bounds, shard_points
    from oracle import haet_oracle as O
    from tests import golden_util as G
    from tests.test_modules_cpu import CTOR

    dev = torch.device("cuda:0")
    errs = {}
    for name in ("irr_c3like_amp", "irr_small_amp", "irr_b2_amp"):
        x, sd, meta, exp, rows = G.load_attn(name)
        mod = CTOR[meta["kind"]](**meta["kw"])
        mod.load_state_dict(sd)
        mod = shard_points(mod.to(dev).eval())
        a, b = shard_bounds(x.shape[1], rank, world)
        with torch.no_grad():
            out = mod(x[:, a:b].contiguous().to(dev))
        errs[name] = O.rel_l2(out.cpu(), exp["out"][:, a:b])
    torch.save(errs, os.path.join(tmp, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_point_sharding_matches_reference(tmp_path, hip_device):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        for name, err in torch.load(os.path.join(tmp_path, f"r{r}.pt")).items():
            assert err < 1e-3, (r, name, err)
            assert err < 2e-4, (r, name, err)
