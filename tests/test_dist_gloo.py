"""The N > 1 path on CPU: world_size-2 gloo process groups exercising the sharding and the single
end-of-run gather (the GPU kernels are replaced by the oracle here -- this tests the plumbing)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mallorn_astrophysics_amd import synth
from mallorn_astrophysics_amd.dist import gather_rows, shard_bounds, shard_csr


def test_shard_bounds_balanced_and_contiguous():
    lc = synth.make_lightcurves(101, seed=8)
    for world in (1, 2, 3, 8):
        b = shard_bounds(lc["offsets"], world)
        assert b[0] == 0 and b[-1] == 101 and np.all(np.diff(b) >= 0) and len(b) == world + 1
        pts = [lc["offsets"][b[r + 1]] - lc["offsets"][b[r]] for r in range(world)]
        assert max(pts) - min(pts) <= 2 * np.diff(lc["offsets"]).max()
        rebuilt = np.concatenate([shard_csr(lc, r, world)[0]["flux"] for r in range(world)])
        assert np.array_equal(rebuilt, lc["flux"])
    # more ranks than objects: empty shards are legal
    small = synth.make_lightcurves(2, seed=1)
    b = shard_bounds(small["offsets"], 4)
    assert b[-1] == 2 and len(b) == 5


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    lc = synth.make_lightcurves(37, seed=21)
    bounds = shard_bounds(lc["offsets"], world)
    sub, sub_z, (lo, hi) = shard_csr(lc, rank, world, lc["z"])
    local = torch.from_numpy(oracle.extract("stat", sub, sub_z))
    assert local.shape[0] == hi - lo
    full = gather_rows(local, 37, bounds)
    if rank == 0:
        np.save(os.path.join(tmp, "full.npy"), full.numpy())
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_matches_single_process(tmp_path, world):
    import oracle
    port = 29511 + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "full.npy")
    lc = synth.make_lightcurves(37, seed=21)
    ref = oracle.extract("stat", lc, lc["z"])
    assert np.array_equal(np.nan_to_num(got, nan=-7.0), np.nan_to_num(ref, nan=-7.0))
