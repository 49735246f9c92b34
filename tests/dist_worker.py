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

    pixel_format = sys.argv[6] if len(sys.argv) > 6 else "rgba32f"

    def to_format(img):
        """rgba8: Framebuffer::as_image's bytes (clamp to [0,1], round to 255 steps, NaN -> 0), rows left in place"""
        if pixel_format == "rgba32f":
            return img
        return (np.clip(np.nan_to_num(img, nan=0.0), 0.0, 1.0) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)

    def render_tiles(tiles):
        # render only this rank's tiles (rect by rect) into the compact list, like a sharded vx_render does
        tx, _ = sharding.tile_grid(w, h)
        tiles.zero_()
        for k, t in enumerate(sharding.local_tile_ids(w, h, rank, world)):
            x0, y0 = (t % tx) * 32, (t // tx) * 32
            img, _ = scene.render(u, w, h, rect=(x0, y0, min(x0 + 32, w), min(y0 + 32, h)), want_hits=False, threads=1)
            blk = to_format(img[y0:y0 + 32, x0:x0 + 32])
            tiles[k, :blk.shape[0], :blk.shape[1]] = torch.from_numpy(np.ascontiguousarray(blk))

    def assemble(gathered, image):
        image.copy_(torch.from_numpy(sharding.assemble_tiles(gathered.numpy(), w, h)))

    # argv[4] = frames per gather, argv[5] = frames to run (every frame has its own ambient light, so that a frame delivered
    # in another frame's place is noticed)
    group = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    n_frames = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    calls = {"before": [], "exchange": []}
    fs = sharding.FrameSharder(w, h, rank, world, dist, "cpu", render_tiles, assemble, buffers=2 * group, group=group,
                               before_render=lambda g: calls["before"].append(g), after_exchange=lambda g: calls["exchange"].append(g),
                               pixel_format=pixel_format)
    delivered = []
    for f in range(n_frames):
        u.ambient = 0.3 + 0.05 * f
        image = fs.step()
        if rank == 0 and image is not None:
            # a full group came back: its frames are in fs.images[0..group)
            delivered += [im.numpy().copy() for im in fs.images]
    image = fs.flush()
    if rank == 0 and n_frames % group:
        delivered += [im.numpy().copy() for im in fs.images[:n_frames % group]]
    dist.barrier()
    assert calls["exchange"] == [k % 2 for k in range(-(-n_frames // group))], calls
    assert calls["before"] == [(f // group) % 2 for f in range(n_frames)], calls
    if rank == 0:
        same = len(delivered) == n_frames
        for f in range(n_frames):
            u.ambient = 0.3 + 0.05 * f
            full, _ = scene.render(u, w, h, want_hits=False, threads=2)
            if pixel_format == "rgba8":
                same = same and delivered[f].dtype == np.uint8 and np.array_equal(delivered[f], to_format(full))
            else:
                same = same and np.array_equal(np.nan_to_num(delivered[f], nan=-7.0), np.nan_to_num(full, nan=-7.0))
        last = image.numpy()
        same = same and (np.array_equal(last, delivered[-1]) if pixel_format == "rgba8" else
                         np.array_equal(np.nan_to_num(last, nan=-7.0), np.nan_to_num(delivered[-1], nan=-7.0)))
        Path(out_path).write_text(f"{int(same)} {world} {fs.n_max} {int(np.isfinite(full).all())}\n")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
