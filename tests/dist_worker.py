"""Worker of tests/test_sharding.py: one rank of a world_size-N gloo job that runs the SAME FrameSharder bench.py uses,
with the CPU oracle standing in for the GPU renderer (the oracle is the checker; this is a test)."""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import SVO_TYPES, orc, vra  # noqa: E402
from voxel_rs_amd import scenes, sharding  # noqa: E402


def main():
    out_path, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    w_ = vra.World(SVO_TYPES["esvo"])
    st = w_.build_heightfield(7, threads=1)
    scene = orc.OracleScene(SVO_TYPES["esvo"], w_.frame(), scenes.synthetic_materials().view(orc.MATERIAL_DTYPE), scenes.synthetic_textures(), 6)
    u = orc.Uniforms.from_buffer_copy(bytes(scenes.bench_camera(7, st["h_max"], w, h)))

    def render_tiles(tiles):
        # render only this rank's tiles (rect by rect) into the compact list, like a sharded vx_render does
        tx, _ = sharding.tile_grid(w, h)
        tiles.zero_()
        for k, t in enumerate(sharding.local_tile_ids(w, h, rank, world)):
            x0, y0 = (t % tx) * 32, (t // tx) * 32
            img, _ = scene.render(u, w, h, rect=(x0, y0, min(x0 + 32, w), min(y0 + 32, h)), want_hits=False, threads=1)
            blk = img[y0:y0 + 32, x0:x0 + 32]
            tiles[k, :blk.shape[0], :blk.shape[1]] = torch.from_numpy(np.ascontiguousarray(blk))

    def assemble(gathered, image):
        image.copy_(torch.from_numpy(sharding.assemble_tiles(gathered.numpy(), w, h)))

    fs = sharding.FrameSharder(w, h, rank, world, dist, "cpu", render_tiles, assemble)
    image = fs.step()
    dist.barrier()
    if rank == 0:
        full, _ = scene.render(u, w, h, want_hits=False, threads=2)
        same = np.array_equal(np.nan_to_num(image.numpy(), nan=-7.0), np.nan_to_num(full, nan=-7.0))
        Path(out_path).write_text(f"{int(same)} {world} {fs.n_max} {int(np.isfinite(full).all())}\n")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
