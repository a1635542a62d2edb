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
        b = shard_bounds(lc["offsets"], world, sets=["stat", "bazin"])          # no GP: cost = points
        assert b[0] == 0 and b[-1] == 101 and np.all(np.diff(b) >= 0) and len(b) == world + 1
        pts = [lc["offsets"][b[r + 1]] - lc["offsets"][b[r]] for r in range(world)]
        assert max(pts) - min(pts) <= 2 * np.diff(lc["offsets"]).max()
        rebuilt = np.concatenate([shard_csr(lc, r, world)[0]["flux"] for r in range(world)])
        assert np.array_equal(rebuilt, lc["flux"])
    # more ranks than objects: empty shards are legal
    small = synth.make_lightcurves(2, seed=1)
    b = shard_bounds(small["offsets"], 4)
    assert b[-1] == 2 and len(b) == 5


def test_shards_of_config4_are_cost_balanced():
    """with the GP on, the cost of an object is a N + b N^3, and the slowest of the 8 ranks sets the
    time of config 4 (10,178 objects over 8 GPUs).  The predicted cost of the heaviest shard must be within 5 % of
    the mean; balancing by point count alone is visibly worse on the same survey."""
    from mallorn_astrophysics_amd.dist import object_costs
    n = np.clip(np.rint(np.random.default_rng(10178).lognormal(np.log(120.0), 0.5, 10178)), 12, 500).astype(np.int64)
    offsets = np.concatenate([[0], np.cumsum(n)])
    assert np.array_equal(offsets, synth.make_lightcurves(10178, seed=10178)["offsets"])   # the C2-C4 survey
    cost = object_costs(offsets, ["stat", "bazin", "gp2d"])

    def imbalance(b):
        per = np.array([cost[b[r]:b[r + 1]].sum() for r in range(8)])
        return per.max() / per.mean()

    b = shard_bounds(offsets, 8, sets=["stat", "bazin", "gp2d"])
    assert b[0] == 0 and b[-1] == 10178 and np.all(np.diff(b) > 0)
    assert imbalance(b) <= 1.05, imbalance(b)
    by_points = shard_bounds(offsets, 8, cost=np.diff(offsets).astype(float))
    assert imbalance(b) <= imbalance(by_points)
    # the default (sets=None) is the full v34a/v55 workload, GP included
    assert np.array_equal(shard_bounds(offsets, 8), b)


def test_int_mask_is_accepted_like_a_set_list():
    """`extract_sharded` forwards the caller's `sets` to the cost model: an int mask (what mask_of / DeviceBatch.run
    accept) must shard exactly like the list of names it stands for."""
    from mallorn_astrophysics_amd.engine import mask_of
    lc = synth.make_lightcurves(300, seed=5)
    for names in (["stat"], ["stat", "gp2d"], ["bazin", "gp1d"], ["stat", "bazin", "powerlaw", "tde", "color", "shape", "physics", "gp2d"]):
        m = mask_of(names)
        assert np.array_equal(shard_bounds(lc["offsets"], 4, m), shard_bounds(lc["offsets"], 4, names)), names
        assert np.array_equal(shard_bounds(lc["offsets"], 4, np.int64(m)), shard_bounds(lc["offsets"], 4, names))
        sub_m = shard_csr(lc, 2, 4, lc["z"], m)
        sub_n = shard_csr(lc, 2, 4, lc["z"], names)
        assert sub_m[2] == sub_n[2] and np.array_equal(sub_m[0]["flux"], sub_n[0]["flux"])


def test_shards_of_the_config5_survey_are_cost_balanced():
    """Config 5 as bench.py --gpus 8 draws it: ONE survey of 8 x 125,000 objects (block b = seed 1000000 + b) cut into
    8 cost-balanced contiguous shards.  The predicted cost of the heaviest shard stays within 5 % of the mean, and a
    shard differs from its block by a fraction of a percent (so a rank generates at most two blocks)."""
    from mallorn_astrophysics_amd.dist import object_costs
    n_all = np.concatenate([synth.lengths(125000, seed=1000000 + b) for b in range(8)])
    offsets = np.concatenate([[0], np.cumsum(n_all)]).astype(np.int64)
    sets = ["stat", "bazin", "powerlaw", "tde", "color", "shape", "physics", "gp2d"]
    b = shard_bounds(offsets, 8, sets)
    cost = object_costs(offsets, sets)
    per = np.array([cost[b[r]:b[r + 1]].sum() for r in range(8)])
    assert per.max() / per.mean() <= 1.05, per / per.mean()
    assert np.abs(np.diff(b) - 125000).max() <= 2500, np.diff(b)


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    lc = synth.make_lightcurves(37, seed=21)
    sets = ["stat"] if world == 2 else 1          # an int mask is as good as the names (world 3)
    bounds = shard_bounds(lc["offsets"], world, sets)
    sub, sub_z, (lo, hi) = shard_csr(lc, rank, world, lc["z"], sets)
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


def _status_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lc = synth.make_lightcurves(23, seed=4)
    bounds = shard_bounds(lc["offsets"], world, ["bazin"])
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    # status words travel like the feature rows: int32 blocks padded to the largest shard (with zeros, not NaN)
    local = torch.arange(lo * 12, hi * 12, dtype=torch.int32).reshape(hi - lo, 12)
    full = gather_rows(local, 23, bounds)
    if rank == 0:
        np.save(os.path.join(tmp, "status.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_gather_of_int32_status_words(tmp_path):
    mp.spawn(_status_worker, args=(3, 29537, str(tmp_path)), nprocs=3, join=True)
    got = np.load(tmp_path / "status.npy")
    assert got.dtype == np.int32 and np.array_equal(got, np.arange(23 * 12, dtype=np.int32).reshape(23, 12))
