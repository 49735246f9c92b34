"""Screen-tile sharding of one frame over the GPUs of a node.

Rays are independent and the SVO is replicated, so the only exchange step is the gather of finished tiles:
32x32-pixel tiles taken in Morton order of their (x, y); the tile at place j of that order belongs to rank j % world
(any run of consecutive places is a compact patch of the screen, so every rank gets an even sample of every region:
sky, silhouettes, near ground). A rank renders its tiles into a COMPACT list (local tile k = the tile at place
k*world + rank); rank 0 gathers the lists (RCCL: every peer sends over its own xGMI link) and scatters them into the image.

`FrameSharder` is the one implementation of that step; bench.py runs it on GPUs over NCCL/RCCL, the CPU tests run it with
gloo and world_size 2 with an oracle-backed renderer.
"""
import numpy as np

TILE = 32


def tile_grid(width, height):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def _spread(v):
    v &= 0xFFFF
    v = (v | (v << 8)) & 0x00FF00FF
    v = (v | (v << 4)) & 0x0F0F0F0F
    v = (v | (v << 2)) & 0x33333333
    return (v | (v << 1)) & 0x55555555


def tile_order(width, height):
    """Row-major tile ids (ty * tiles_x + tx, from the bottom-left) in Morton order of (tx, ty): vx_tile_order."""
    tx, ty = tile_grid(width, height)
    return sorted(range(tx * ty), key=lambda t: (_spread(t % tx) | (_spread(t // tx) << 1), t))


def local_tile_ids(width, height, rank, world):
    return tile_order(width, height)[rank::world]


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
    assemble(gathered_view, image_tensor): rank 0 only; gathered_view is [world][n_max][32][32][4], possibly strided over ranks.

    `buffers` frames are in flight (one tile list each), so that frame k+1's render overlaps frame k's gather/assembly.
    `group` consecutive frames share ONE collective (their tile lists are adjacent in memory): the gather's fixed cost -- launch,
    rendezvous, one message per peer -- is paid once per group, which is what matters when a rank's share of a frame takes
    about as long as a small collective does (8 GPUs at 1080p). step() then returns None for the frames that only rendered,
    and flush() exchanges a group that is not full yet (call it before waiting for the last frames).

    `gather(tiles, gathered)` replaces torch.distributed's gather for the exchange (same layout: rank r's lists at gathered[r]).

    Stream-ordering hooks (GPU: the renderer has its own streams, the collective runs on torch's; no-ops on CPU):
      before_render(g)  -- the renderer must not overwrite group g's tile lists while the collective that last used them is
                           still sending;
      after_render()    -- the collective's stream has to wait for the render just issued;
      after_exchange(g) -- group g's gather (and, on rank 0, assembly) has been issued.
    """

    def __init__(self, width, height, rank, world, dist, device, render_tiles, assemble, before_render=None, after_render=None,
                 after_exchange=None, buffers=2, group=1, gather=None, pixel_format="rgba32f", fused=None):
        import torch

        # rgba32f: the reference's framebuffer; rgba8: what Framebuffer::as_image reads back from it (vx_format) -- a quarter of the
        # bytes on the links, which is what the exchange is bound by: rank 0 takes in (world - 1) / world of every frame
        if pixel_format not in ("rgba32f", "rgba8"):
            raise ValueError("pixel_format must be rgba32f or rgba8")
        dtype = torch.float32 if pixel_format == "rgba32f" else torch.uint8
        self.pixel_format, self.dtype = pixel_format, dtype

        if group < 1 or buffers < group or buffers % group:
            raise ValueError("buffers must be a multiple of group")
        self.width, self.height, self.rank, self.world, self.dist = width, height, rank, world, dist
        self.render_tiles, self.assemble = render_tiles, assemble
        self.before_render = before_render or (lambda g: None)
        self.after_render = after_render or (lambda: None)
        self.after_exchange = after_exchange or (lambda g: None)
        self.group = group
        # gather(tiles, gathered_or_None): the exchange step. Default: torch.distributed's gather (what the CPU tests run, on gloo);
        # bench.py hands in the library's own (vx_gather_tiles: RCCL send/receive owned by the render context)
        self.gather = gather
        # fused(g, tiles, gathered_or_None, image_or_None): the whole frame -- wait for the exchange that last read buffer g's list, render, gather,
        # assemble on rank 0 -- as ONE call of the renderer (vx_render_gather), for frames that are exchanged one by one (group 1): a rank's
        # share of a frame is tens of microseconds of GPU time at eight ranks, and so are four separate calls from a Python loop
        self.fused = fused if group == 1 else None
        self.n_local = len(local_tile_ids(width, height, rank, world))
        self.n_max = max(len(local_tile_ids(width, height, r, world)) for r in range(world))
        n_groups = buffers // group
        # [group][frame in group][tile]...: the lists of one group are one contiguous message
        self.tiles = [torch.zeros((group, self.n_max, TILE, TILE, 4), dtype=dtype, device=device) for _ in range(n_groups)]
        self.gathered = [torch.zeros((world, group, self.n_max, TILE, TILE, 4), dtype=dtype, device=device) if rank == 0 else None
                         for _ in range(n_groups)]
        if gather is not None and rank == 0:
            # the root renders straight into its own place in the gathered buffer: its share needs no copy (vx_gather_tiles)
            self.tiles = [self.gathered[i][0] for i in range(n_groups)]
        self.images = [torch.zeros((height, width, 4), dtype=dtype, device=device) for _ in range(group)] if rank == 0 else None
        if str(device).startswith("cuda"):
            torch.cuda.synchronize()  # the zero fills ran on torch's stream; a renderer with its own streams must not race them
        self.frame = 0  # frames rendered
        self._g = 0     # the group being filled ...
        self._slot = 0  # ... and how many of its frames are rendered
        self._last = None
        self.last_gathered = None

    @property
    def image(self):
        """rank 0: the most recently assembled frame"""
        return self.images[self._last] if self.rank == 0 and self._last is not None else None

    def step(self):
        if self.fused is not None:
            g = self._g
            self.fused(g, self.tiles[g][0], self.gathered[g], self.images[0] if self.rank == 0 else None)
            self.frame += 1
            if self.rank == 0:
                self._last = 0
                self.last_gathered = self.gathered[g][:, 0]
            self._g = (g + 1) % len(self.tiles)
            return self.image
        g, j = self._g, self._slot
        self.before_render(g)
        self.render_tiles(self.tiles[g][j])
        self.after_render()
        self.frame += 1
        self._slot += 1
        if self._slot < self.group:
            return None
        return self._exchange()

    def flush(self):
        """Exchanges the frames of a group that is not full yet (same count on every rank: they step together); the next frame
        starts a new group."""
        return self._exchange() if self._slot else self.image

    def _exchange(self):
        g, count = self._g, self._slot
        tiles, gathered = self.tiles[g][:count], self.gathered[g]
        # the one exchange step of the path: finished tiles -> rank 0
        if self.gather is not None:
            self.gather(self.tiles[g], gathered)  # (whole groups: a group that is not full yet travels with its unused lists)
        else:
            self.dist.gather(tiles, [gathered[r, :count] for r in range(self.world)] if self.rank == 0 else None, dst=0)
        if self.rank == 0:
            for j in range(count):
                self.assemble(gathered[:, j], self.images[j])
            self._last = count - 1
            self.last_gathered = gathered[:, count - 1]  # [world][n_max]...: the newest frame's tile lists as they arrived
        self.after_exchange(g)
        self._g, self._slot = (g + 1) % len(self.tiles), 0
        return self.image
