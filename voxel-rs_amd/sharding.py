"""Screen-tile sharding of one frame over the GPUs of a node.

Rays are independent and the SVO is replicated, so the only exchange step is the gather of finished tiles:
32x32-pixel tiles, numbered row-major from the bottom-left, tile t belongs to rank t % world (interleaved, so sky-only
and silhouette-heavy regions spread evenly). A rank renders its tiles into a COMPACT list (local tile k = global tile
k*world + rank); rank 0 gathers the lists (RCCL: every peer sends over its own xGMI link) and scatters them into the image.

`FrameSharder` is the one implementation of that step; bench.py runs it on GPUs over NCCL/RCCL, the CPU tests run it with
gloo and world_size 2 with an oracle-backed renderer.
"""
import numpy as np

TILE = 32


def tile_grid(width, height):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def local_tile_ids(width, height, rank, world):
    tx, ty = tile_grid(width, height)
    return list(range(rank, tx * ty, world))


def extract_tiles(image, rank, world):
    """Compact tile list of `rank` cut from a full [h][w][c] image (pixels outside the image are zero): what a sharded
    vx_render writes."""
    h, w, c = image.shape
    tx, _ = tile_grid(w, h)
    ids = local_tile_ids(w, h, rank, world)
    out = np.zeros((len(ids), TILE, TILE, c), dtype=image.dtype)
    for k, t in enumerate(ids):
        x0, y0 = (t % tx) * TILE, (t // tx) * TILE
        blk = image[y0:y0 + TILE, x0:x0 + TILE]
        out[k, :blk.shape[0], :blk.shape[1]] = blk
    return out


def assemble_tiles(gathered, width, height):
    """numpy counterpart of vx_assemble_tiles: gathered is [world][n_max][32][32][c]."""
    world = gathered.shape[0]
    tx, _ = tile_grid(width, height)
    out = np.zeros((height, width, gathered.shape[-1]), dtype=gathered.dtype)
    for rank in range(world):
        for k, t in enumerate(local_tile_ids(width, height, rank, world)):
            x0, y0 = (t % tx) * TILE, (t // tx) * TILE
            hh, ww = min(TILE, height - y0), min(TILE, width - x0)
            out[y0:y0 + hh, x0:x0 + ww] = gathered[rank, k, :hh, :ww]
    return out


class FrameSharder:
    """One frame = render own tiles -> gather to rank 0 -> assemble.

    render_tiles(tiles_tensor): fills this rank's compact tile list (asynchronously on the renderer's stream is fine).
    assemble(gathered_tensor, image_tensor): rank 0 only.
    Stream-ordering hooks (GPU: the renderer has its own stream, the collective runs on torch's; no-ops on CPU):
      before_render -- the renderer must not overwrite a tile list the previous collective using it is still sending;
      before_gather -- the collective waits for the render;  after_gather -- the assembly waits for the collective.
    Tile lists and gather buffers are double-buffered so that frame k+1's render overlaps frame k's gather/assembly.
    """

    def __init__(self, width, height, rank, world, dist, device, render_tiles, assemble, before_render=None, before_gather=None,
                 after_gather=None, buffers=2):
        import torch

        self.width, self.height, self.rank, self.world, self.dist = width, height, rank, world, dist
        self.render_tiles, self.assemble = render_tiles, assemble
        noop = lambda: None  # noqa: E731
        self.before_render, self.before_gather, self.after_gather = before_render or noop, before_gather or noop, after_gather or noop
        self.n_local = len(local_tile_ids(width, height, rank, world))
        self.n_max = max(len(local_tile_ids(width, height, r, world)) for r in range(world))
        self.tiles = [torch.zeros((self.n_max, TILE, TILE, 4), dtype=torch.float32, device=device) for _ in range(buffers)]
        self.gathered = [torch.zeros((world, self.n_max, TILE, TILE, 4), dtype=torch.float32, device=device) if rank == 0 else None
                         for _ in range(buffers)]
        self.image = torch.zeros((height, width, 4), dtype=torch.float32, device=device) if rank == 0 else None
        if str(device).startswith("cuda"):
            torch.cuda.synchronize()  # the zero fills ran on torch's stream; a renderer with its own streams must not race them
        self.frame = 0

    def step(self):
        b = self.frame % len(self.tiles)
        self.frame += 1
        tiles, gathered = self.tiles[b], self.gathered[b]
        self.before_render()
        self.render_tiles(tiles)
        self.before_gather()
        # the one exchange step of the path: finished tiles -> rank 0
        self.dist.gather(tiles, list(gathered.unbind(0)) if self.rank == 0 else None, dst=0)
        if self.rank == 0:
            self.after_gather()
            self.assemble(gathered, self.image)
        return self.image
