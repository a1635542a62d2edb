"""bench.py host logic that needs no GPU: the sharded survey every rank draws from, and the launcher guard."""
import json
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT
from mallorn_astrophysics_amd import synth

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_shards_of_all_ranks_rebuild_the_survey():
    objects, seed = 300, 77
    sets = ["stat", "bazin", "gp2d"]
    for world in (1, 2, 3):
        survey = synth.concat([synth.make_lightcurves(objects, seed=seed + b) for b in range(world)])
        flux, n_tot = [], 0
        for rank in range(world):
            lc, bounds, n_local = bench.local_shard(objects, seed, rank, world, sets)
            assert bounds[0] == 0 and bounds[-1] == objects * world
            assert n_local == bounds[rank + 1] - bounds[rank] == len(lc["offsets"]) - 1
            flux.append(lc["flux"])
            n_tot += n_local
        assert n_tot == objects * world
        assert np.array_equal(np.concatenate(flux), survey["flux"])
    # one rank: exactly make_lightcurves(objects, seed)
    lc, _, _ = bench.local_shard(objects, seed, 0, 1, sets)
    assert np.array_equal(lc["t"], synth.make_lightcurves(objects, seed=seed)["t"])


def test_gpus_2_without_a_launcher_never_reports_a_one_gpu_result():
    """`python bench.py --gpus 2` starts its own ranks (torch.distributed.run, child process).  On this GPU-less box
    the ranks fail: the call must exit non-zero and must not print a result line at all -- in particular not an
    n_gpus = 1 line dressed up as success."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--objects", "40", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode == 0:                       # a box with >= 2 GPUs: the line must say 2
        assert lines and json.loads(lines[-1])["n_gpus"] == 2
    else:
        assert not lines, r.stdout
    assert "launching" in r.stderr
