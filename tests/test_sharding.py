"""Multi-GPU path on CPU: tile arithmetic, and the FrameSharder step (render own tiles -> gather -> assemble) under
torch.distributed with the gloo backend and world_size 2, against a single-process render of the same frame."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from helpers import vra  # noqa: F401
from voxel_rs_amd import hip, sharding

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("w,h,world", [(1920, 1080, 8), (200, 120, 3), (33, 31, 2), (64, 64, 5)])
def test_tile_arithmetic_matches_the_library(w, h, world):
    tx, ty = sharding.tile_grid(w, h)
    ids = [sharding.local_tile_ids(w, h, r, world) for r in range(world)]
    assert sorted(t for l in ids for t in l) == list(range(tx * ty))  # every tile exactly once
    for r in range(world):
        assert len(ids[r]) == hip.local_tile_count(w, h, r, world)  # vx_local_tile_count
    assert max(map(len, ids)) - min(map(len, ids)) <= 1  # interleaving balances the tile counts


def test_extract_then_assemble_is_identity():
    rng = np.random.default_rng(0)
    for (w, h, world) in ((200, 120, 3), (96, 64, 2), (50, 70, 4)):
        img = rng.random((h, w, 4), dtype=np.float32)
        n_max = max(len(sharding.local_tile_ids(w, h, r, world)) for r in range(world))
        gathered = np.zeros((world, n_max, 32, 32, 4), dtype=np.float32)
        for r in range(world):
            t = sharding.extract_tiles(img, r, world)
            gathered[r, :len(t)] = t
        assert np.array_equal(sharding.assemble_tiles(gathered, w, h), img)


@pytest.mark.parametrize("world,group,frames,pixel_format", [(2, 1, 1, "rgba32f"), (3, 1, 2, "rgba32f"), (2, 2, 5, "rgba32f"), (2, 1, 3, "rgba8")])
def test_frame_sharder_with_gloo(tmp_path, world, group, frames, pixel_format):
    out = tmp_path / "result.txt"
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + world + os.getpid() % 200), str(ROOT / "tests" / "dist_worker.py"), str(out), "200", "120", str(group), str(frames), pixel_format]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    same, w_, n_max, _ = out.read_text().split()
    assert same == "1" and int(w_) == world and int(n_max) == -(-28 // world)
