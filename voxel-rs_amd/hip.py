"""ctypes binding of libvoxelhip.so (include/voxel_hip.h) shaped like the reference's `graphics::Svo`
(src/graphics/svo.rs:56-256): new / update / render / raycast / get_stats.

This is harness code. It adds nothing to the path: every method is one or two C-ABI calls, and every failure
of the native library raises -- there is no fallback of any kind.
"""
import ctypes as C
import math

import numpy as np

from .build import lib_path, share_hip_runtime_with_torch

VX_OK = 0
VX_MEM_HOST, VX_MEM_DEVICE = 0, 1
VX_FORMAT_RGBA32F, VX_FORMAT_RGBA8 = 0, 1
VX_COMM_ID_BYTES = 128
TILE = 32

MATERIAL_DTYPE = np.dtype([("specular_pow", "<f4"), ("specular_strength", "<f4"), ("tex_top", "<i4"), ("tex_side", "<i4"), ("tex_bottom", "<i4"),
                           ("tex_top_normal", "<i4"), ("tex_side_normal", "<i4"), ("tex_bottom_normal", "<i4")])
HIT_DTYPE = np.dtype([("t", "<f4"), ("value", "<u4"), ("face_id", "<i4"), ("flags", "<u4"), ("pos", "<f4", 3), ("lod", "<f4"), ("uv", "<f4", 2),
                      ("shadow_t", "<f4"), ("steps", "<u4")])
PICKER_TASK_DTYPE = np.dtype([("max_dst", "<f4"), ("_p0", "<f4", 3), ("pos", "<f4", 3), ("_p1", "<f4"), ("dir", "<f4", 3), ("_p2", "<f4")])
PICKER_RESULT_DTYPE = np.dtype([("dst", "<f4"), ("inside_voxel", "<u4"), ("_p0", "<f4", 2), ("pos", "<f4", 3), ("_p1", "<f4"), ("normal", "<f4", 3),
                                ("_p2", "<f4")])
FRAME_DTYPE = np.dtype([("t_min", "<f4"), ("ptr", "<u4"), ("idx", "<u4"), ("parent_octant_idx", "<u4"), ("scale", "<i4"), ("is_child", "<i4"),
                        ("is_leaf", "<i4"), ("crossed_boundary", "<i4"), ("next_ptr", "<u4")])
assert HIT_DTYPE.itemsize == 48 and PICKER_TASK_DTYPE.itemsize == 48 and PICKER_RESULT_DTYPE.itemsize == 48 and FRAME_DTYPE.itemsize == 36

COUNTER_FIELDS = ["rays", "iterations", "pushes", "leaf_tests", "leaf_tests_trilinear", "boundaries", "csvo_header_bytes", "csvo_pointer_bytes",
                  "pixels", "lit_pixels", "shadow_rays", "wave_steps", "services", "refills", "tail_wave_steps", "tail_iterations"]


class Uniforms(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("fovy", C.c_float), ("aspect", C.c_float), ("ambient", C.c_float), ("light_dir", C.c_float * 3),
                ("cam_pos", C.c_float * 3), ("render_shadows", C.c_int32), ("shadow_distance", C.c_float), ("highlight_pos", C.c_float * 3)]


class Range(C.Structure):
    _fields_ = [("start", C.c_uint64), ("length", C.c_uint64)]


class Target(C.Structure):
    _fields_ = [("rgba32f", C.c_void_p), ("hits", C.c_void_p), ("memory", C.c_int32), ("tile_rank", C.c_uint32), ("tile_count", C.c_uint32),
                ("format", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("t", C.c_float), ("value", C.c_uint32), ("face_id", C.c_int32), ("pos", C.c_float * 3), ("uv", C.c_float * 2),
                ("color", C.c_float * 4), ("lod", C.c_float), ("inside_voxel", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("used_bytes", C.c_uint64), ("capacity_bytes", C.c_uint64), ("depth", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_FIELDS]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_FIELDS}


# every symbol include/voxel_hip.h declares: (restype, argtypes)
_vp, _u32, _u64, _sz, _int = C.c_void_p, C.c_uint32, C.c_uint64, C.c_size_t, C.c_int
SYMBOLS = {
    "vx_create": (_int, [_int, _sz, _int, C.POINTER(_vp)]),
    "vx_destroy": (None, [_vp]),
    "vx_set_materials": (_int, [_vp, _vp, _u32]),
    "vx_set_textures": (_int, [_vp, _vp, _u32, _u32, _u32, _u32]),
    "vx_staging_ptr": (_vp, [_vp]),
    "vx_capacity": (_sz, [_vp]),
    "vx_commit": (_int, [_vp, _u32, _vp, _u32, _u64]),
    "vx_commit_all": (_int, [_vp, _u32, _u64]),
    "vx_set_commit_mode": (_int, [_vp, _int]),
    "vx_commit_wait": (_int, [_vp]),
    "vx_get_stats": (_int, [_vp, C.POINTER(Stats)]),
    "vx_render": (_int, [_vp, C.POINTER(Uniforms), _u32, _u32, C.POINTER(Target)]),
    "vx_raycast": (_int, [_vp, _vp, _u32, _vp]),
    "vx_debug_trace": (_int, [_vp, C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 3), C.c_float, _int, C.POINTER(Result), _vp, _u32, C.POINTER(_u32)]),
    "vx_sync": (_int, [_vp]),
    "vx_set_frames_in_flight": (_int, [_vp, _int]),
    "vx_wait_event": (_int, [_vp, _vp]),
    "vx_stream_wait_render": (_int, [_vp, _vp]),
    "vx_traversal_image": (_u64, [_int, _vp, _u64, _int, _vp, _u64]),
    "vx_traversal_image_with_origin": (_u64, [_int, _vp, _u64, _int, _vp, _u64, _vp, _u64]),
    "vx_arena_capacity": (_sz, [_vp]),
    "vx_resolve_2x2": (_int, [_vp, _vp, _u32, _u32, _vp, _vp]),
    "vx_assemble_tiles": (_int, [_vp, _vp, _u64, _u32, _u32, _u32, _vp]),
    "vx_assemble_tiles_on": (_int, [_vp, _vp, _u64, _u32, _u32, _u32, _vp, _vp]),
    "vx_assemble_tiles_format": (_int, [_vp, _vp, _u64, _u32, _u32, _u32, _vp, _int, _vp]),
    "vx_tile_order": (_u32, [_u32, _u32, _vp, _u32]),
    "vx_present_begin": (_int, [_vp, C.POINTER(Uniforms), _u32, _u32, _int, C.POINTER(_int)]),
    "vx_present_wait": (_int, [_vp, _int, C.POINTER(_vp), C.POINTER(_sz)]),
    "vx_comm_library": (_int, [C.c_char_p]),
    "vx_comm_unique_id": (_int, [_vp, _sz]),
    "vx_comm_init": (_int, [_vp, _int, _int, _vp]),
    "vx_comm_destroy": (_int, [_vp]),
    "vx_comm_info": (_int, [_vp, C.POINTER(_int), C.POINTER(_int)]),
    "vx_set_comm_headroom": (_int, [_vp, _int]),
    "vx_gather_tiles": (_int, [_vp, _vp, _u64, _vp, _int, C.POINTER(_int)]),
    "vx_wait_gather": (_int, [_vp, _int]),
    "vx_render_gather": (_int, [_vp, C.POINTER(Uniforms), _u32, _u32, C.POINTER(Target), _u64, _vp, _int, _vp, _int, C.POINTER(_int)]),
    "vx_comm_stream": (_vp, [_vp]),
    "vx_local_tile_count": (_u32, [_u32, _u32, _u32, _u32]),
    "vx_render_counters": (_int, [_vp, C.POINTER(Uniforms), _u32, _u32, _u32, _u32, C.POINTER(Counters)]),
    "vx_excursion_counters": (_int, [_vp, C.POINTER(_u64 * 4), _int]),
    "vx_image_info": (_int, [_vp, C.POINTER(_u64 * 4)]),
    "vx_debug_knobs": (_int, [_vp, C.POINTER(_u32 * 8)]),
    "vx_timeline_read": (_u32, [_vp, _vp, _u32]),
    "vx_gather_query": (_int, [_vp, _int]),
    "vx_comm_profile_read": (_int, [_vp, C.POINTER(C.c_double), C.POINTER(_u32)]),
    "vx_clock_probe": (_int, [_vp, _u32, C.POINTER(C.c_double)]),
    "vx_profile_enable": (_int, [_vp, _int]),
    "vx_profile_read": (_int, [_vp, C.POINTER(C.c_double), C.POINTER(_u32)]),
    "vx_stream": (_vp, [_vp]),
    "vx_device": (_int, [_vp]),
    "vx_last_error": (C.c_char_p, []),
    "vx_version": (C.c_char_p, []),
}

_lib = None


class VoxelHipError(RuntimeError):
    pass


def lib():
    """Loads libvoxelhip.so; raises if it (or any declared symbol) is missing -- never falls back."""
    global _lib
    if _lib is None:
        share_hip_runtime_with_torch()
        L = C.CDLL(str(lib_path("libvoxelhip.so")))
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _check(rc):
    if rc != VX_OK:
        raise VoxelHipError(f"libvoxelhip error {rc}: {lib().vx_last_error().decode()}")


def make_uniforms(view, fovy, aspect, ambient, light_dir, cam_pos, render_shadows, shadow_distance, highlight_pos=None):
    u = Uniforms()
    u.view = (C.c_float * 16)(*[float(x) for x in view])
    u.fovy, u.aspect, u.ambient = float(fovy), float(aspect), float(ambient)
    u.light_dir = (C.c_float * 3)(*[float(x) for x in light_dir])
    u.cam_pos = (C.c_float * 3)(*[float(x) for x in cam_pos])
    u.render_shadows = int(render_shadows)
    u.shadow_distance = float(shadow_distance)
    hp = highlight_pos if highlight_pos is not None else (math.nan,) * 3  # NaN when nothing is selected (svo.rs:211)
    u.highlight_pos = (C.c_float * 3)(*[float(x) for x in hp])
    return u


def local_tile_count(width, height, rank, count):
    return lib().vx_local_tile_count(width, height, rank, count)


def tile_order(width, height):
    """vx_tile_order: row-major tile ids in the Morton order the ranks share out."""
    n = lib().vx_tile_order(width, height, None, 0)
    out = np.zeros(n, dtype=np.uint32)
    lib().vx_tile_order(width, height, out.ctypes.data_as(_vp), n)
    return out


def comm_library(path):
    """The RCCL build to open instead of librccl.so.1 (vx_comm_library); before the first communicator."""
    _check(lib().vx_comm_library(str(path).encode()))


def comm_unique_id():
    """A fresh RCCL unique id (bytes) for vx_comm_init; made on ONE rank and handed to the others by the caller."""
    buf = C.create_string_buffer(VX_COMM_ID_BYTES)
    _check(lib().vx_comm_unique_id(buf, VX_COMM_ID_BYTES))
    return buf.raw


def traversal_image(svo_type, world_frame_words, used_bytes, layout=0, with_origin=False):
    """vx_traversal_image: world frame (uint32 words: scale, header, arena...) -> traversal image (uint32 words);
    layout 0 = ESVO frame (walkable by any ESVO traversal), 1 = what the renderer walks -- octants of one {pointer | value, masks} entry per existing
    child, addressed in 8-byte units, through a buffer resource --, 2 = the same bytes walked through a 64-bit pointer (images beyond 4 GiB).
    with_origin: (image, image) -- since round 6 the origin of a CSVO world's voxel-parent octant is a unit of the image itself, in front of the octant's
    values; what used to be a table of its own IS the image (a walk reads unit `lo` of it)."""
    f = np.ascontiguousarray(world_frame_words, dtype=np.uint32)
    n = lib().vx_traversal_image(svo_type, f.ctypes.data_as(_vp), used_bytes, layout, None, 0)
    if n == 0:
        raise ValueError("this world frame cannot be imaged")
    out = np.zeros(n, dtype=np.uint32)
    lib().vx_traversal_image(svo_type, f.ctypes.data_as(_vp), used_bytes, layout, out.ctypes.data_as(_vp), n)
    return (out, out) if with_origin else out


class Svo:
    """`graphics::Svo` over the C ABI."""

    def __init__(self, svo_type, capacity_bytes, device=0):
        self._h = _vp()
        self.svo_type = svo_type
        _check(lib().vx_create(svo_type, capacity_bytes, device, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            lib().vx_destroy(self._h)
            self._h = None

    __del__ = close

    # -- resources ---------------------------------------------------------------------------------------
    def set_materials(self, materials):
        m = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
        _check(lib().vx_set_materials(self._h, m.ctypes.data_as(_vp), m.size))

    def set_textures(self, base_rgba8, mip_levels):
        """base_rgba8: uint8 [layers][h][w][4], row 0 = bottom (already flipped like TextureArrayBuilder does)."""
        t = np.ascontiguousarray(base_rgba8, dtype=np.uint8)
        layers, h, w, c = t.shape
        assert c == 4
        _check(lib().vx_set_textures(self._h, t.ctypes.data_as(_vp), w, h, layers, mip_levels))

    # -- Svo::update (svo.rs:171-189) ---------------------------------------------------------------------
    def update(self, world):
        ranges = world.updated_ranges()
        staging = lib().vx_staging_ptr(self._h)
        world.write_changes_to(staging + 4, lib().vx_arena_capacity(self._h), True)
        arr = (Range * max(len(ranges), 1))()
        for i, (s, n) in enumerate(ranges):
            arr[i].start, arr[i].length = s, n
        _check(lib().vx_commit(self._h, world.depth, C.cast(arr, _vp), len(ranges), world.size_in_bytes))

    def update_full(self, world):
        """First upload into a fresh buffer: WorldSvo::write_to (whole arena) instead of the dirty ranges, which a
        WorldSvo only tracks for ONE target buffer (they are cleared by the first write_changes_to)."""
        if not hasattr(world, "write_frame_to"):
            self.upload_frame(world.frame(pad_words=0), world.depth)
            return
        # straight into the staging buffer, as graphics::Svo::update writes into its mapped buffer (svo.rs:171-189): no copy on the way
        try:
            wrote = world.write_frame_to(lib().vx_staging_ptr(self._h), lib().vx_capacity(self._h))
        except ValueError as e:
            raise VoxelHipError(str(e)) from None
        header = 20 if self.svo_type == 1 else 4
        _check(lib().vx_commit_all(self._h, world.depth, max(wrote - 4 - header, 0)))

    def upload_frame(self, frame_words, depth):
        """Copies a complete mapped-buffer image ([f32 scale][header][arena]) into staging and commits all of it."""
        raw = np.ascontiguousarray(frame_words).view(np.uint8)
        cap = lib().vx_capacity(self._h)
        if raw.size > cap:
            raise VoxelHipError("frame larger than the world buffer")
        C.memmove(lib().vx_staging_ptr(self._h), raw.ctypes.data, raw.size)
        header = 20 if self.svo_type == 1 else 4
        _check(lib().vx_commit_all(self._h, depth, max(raw.size - 4 - header, 0)))

    def set_commit_mode(self, pipelined):
        """VX_COMMIT_PIPELINED: update() only posts the commit; a worker thread of the context does the rest."""
        _check(lib().vx_set_commit_mode(self._h, 1 if pipelined else 0))

    def commit_wait(self):
        _check(lib().vx_commit_wait(self._h))

    def get_stats(self):
        s = Stats()
        _check(lib().vx_get_stats(self._h, C.byref(s)))
        return dict(used_bytes=int(s.used_bytes), capacity_bytes=int(s.capacity_bytes), depth=int(s.depth))

    # -- Svo::render (svo.rs:196-229) ---------------------------------------------------------------------
    def render(self, uniforms, width, height, want_hits=False, tile_rank=0, tile_count=1, fmt=VX_FORMAT_RGBA32F):
        """Returns (image [h][w][4] -- float32, row 0 = bottom; or uint8 (fmt RGBA8), row 0 = top --, hits or None); tile-sharded
        calls return compact tile lists."""
        dt = np.uint8 if fmt == VX_FORMAT_RGBA8 else np.float32
        if tile_count > 1:
            n = local_tile_count(width, height, tile_rank, tile_count)
            img = np.zeros((n, TILE, TILE, 4), dtype=dt)
            hits = np.zeros((n, TILE, TILE), dtype=HIT_DTYPE) if want_hits else None
        else:
            img = np.zeros((height, width, 4), dtype=dt)
            hits = np.zeros((height, width), dtype=HIT_DTYPE) if want_hits else None
        t = Target(img.ctypes.data, hits.ctypes.data if want_hits else None, VX_MEM_HOST, tile_rank, tile_count, fmt)
        _check(lib().vx_render(self._h, C.byref(uniforms), width, height, C.byref(t)))
        return img, hits

    def render_device(self, uniforms, width, height, out_ptr, hits_ptr=None, tile_rank=0, tile_count=1, fmt=VX_FORMAT_RGBA32F):
        """Asynchronous render into device memory (e.g. a torch tensor's data_ptr()); pair with sync()."""
        t = Target(out_ptr, hits_ptr, VX_MEM_DEVICE, tile_rank, tile_count, fmt)
        _check(lib().vx_render(self._h, C.byref(uniforms), width, height, C.byref(t)))

    # -- pipelined presentation -------------------------------------------------------------------------------
    def present_begin(self, uniforms, width, height, fmt=VX_FORMAT_RGBA8):
        slot = _int(0)
        _check(lib().vx_present_begin(self._h, C.byref(uniforms), width, height, fmt, C.byref(slot)))
        return slot.value

    def present_wait(self, slot, width, height, fmt=VX_FORMAT_RGBA8):
        """The slot's image as a numpy VIEW of the library's pinned ring (copy it to keep it beyond three more frames)."""
        ptr, n = _vp(), _sz(0)
        _check(lib().vx_present_wait(self._h, slot, C.byref(ptr), C.byref(n)))
        dt = np.uint8 if fmt == VX_FORMAT_RGBA8 else np.float32
        buf = (C.c_char * n.value).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dt).reshape(height, width, 4)

    # -- multi-GPU ------------------------------------------------------------------------------------------------
    def comm_init(self, nranks, rank, unique_id):
        _check(lib().vx_comm_init(self._h, nranks, rank, C.c_char_p(unique_id)))

    def comm_destroy(self):
        _check(lib().vx_comm_destroy(self._h))

    def set_comm_headroom(self, waves_per_cu):
        """Wave slots per CU left to RCCL's workgroups by a context whose communicator has more than one rank (vx_set_comm_headroom)."""
        _check(lib().vx_set_comm_headroom(self._h, int(waves_per_cu)))

    def gather_tiles(self, tiles_ptr, bytes_per_rank, gathered_ptr, root=0):
        ticket = _int(-1)
        _check(lib().vx_gather_tiles(self._h, tiles_ptr, bytes_per_rank, gathered_ptr, root, C.byref(ticket)))
        return ticket.value

    def render_gather(self, uniforms, width, height, tiles_ptr, bytes_per_rank, gathered_ptr, image_ptr, wait_ticket=-1, tile_rank=0, tile_count=1, fmt=VX_FORMAT_RGBA32F, root=0):
        """One sharded frame in one call (vx_render_gather): wait for the exchange that last read the list, render this rank's tiles, gather them to
        `root`, and assemble on the root. Returns the exchange's ticket."""
        t = Target(tiles_ptr, None, VX_MEM_DEVICE, tile_rank, tile_count, fmt)
        ticket = _int(-1)
        _check(lib().vx_render_gather(self._h, C.byref(uniforms), width, height, C.byref(t), bytes_per_rank, gathered_ptr, root, image_ptr, wait_ticket, C.byref(ticket)))
        return ticket.value

    def wait_gather(self, ticket):
        _check(lib().vx_wait_gather(self._h, ticket))

    def gather_query(self, ticket):
        """1 = that gather (and an assembly issued behind it on the communicator's stream) has finished, 0 = not yet, -1 = error. Never blocks."""
        return int(lib().vx_gather_query(self._h, ticket))

    def comm_profile_read(self):
        ms, n = C.c_double(0), _u32(0)
        _check(lib().vx_comm_profile_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    @property
    def comm_stream(self):
        return lib().vx_comm_stream(self._h)

    def assemble_tiles_format(self, tiles_ptr, stride_pixels, tile_count, width, height, out_ptr, fmt, stream):
        _check(lib().vx_assemble_tiles_format(self._h, tiles_ptr, stride_pixels, tile_count, width, height, out_ptr, fmt, _vp(stream or 0)))

    def render_counters(self, uniforms, width, height, tile_rank=0, tile_count=1):
        c = Counters()
        _check(lib().vx_render_counters(self._h, C.byref(uniforms), width, height, tile_rank, tile_count, C.byref(c)))
        return c.as_dict()

    # -- Svo::raycast (svo.rs:233-255) --------------------------------------------------------------------
    def raycast(self, tasks):
        tasks = np.ascontiguousarray(tasks, dtype=PICKER_TASK_DTYPE)
        out = np.zeros(tasks.size, dtype=PICKER_RESULT_DTYPE)
        _check(lib().vx_raycast(self._h, tasks.ctypes.data_as(_vp), tasks.size, out.ctypes.data_as(_vp)))
        return out

    def debug_trace(self, pos, direction, max_dst, cast_translucent, max_frames=100):
        p = (C.c_float * 3)(*[float(x) for x in pos])
        d = (C.c_float * 3)(*[float(x) for x in direction])
        res = Result()
        frames = np.zeros(max(max_frames, 1), dtype=FRAME_DTYPE)
        n = _u32(0)
        _check(lib().vx_debug_trace(self._h, C.byref(p), C.byref(d), max_dst, int(cast_translucent), C.byref(res), frames.ctypes.data_as(_vp),
                                    max_frames, C.byref(n)))
        return res, frames[:min(n.value, max_frames)], n.value

    def assemble_tiles(self, tiles_ptr, stride_floats, tile_count, width, height, out_ptr, stream=None):
        """stream = a raw hipStream_t to launch on (default: the context's own stream)."""
        if stream is None:
            _check(lib().vx_assemble_tiles(self._h, tiles_ptr, stride_floats, tile_count, width, height, out_ptr))
        else:
            _check(lib().vx_assemble_tiles_on(self._h, tiles_ptr, stride_floats, tile_count, width, height, out_ptr, _vp(stream)))

    def resolve_2x2(self, src_ptr, width, height, dst_ptr, stream=None):
        """Box-filters a (2*width x 2*height) device image down to width x height on the raw hipStream_t `stream`."""
        _check(lib().vx_resolve_2x2(self._h, src_ptr, width, height, dst_ptr, _vp(stream or 0)))

    def sync(self):
        _check(lib().vx_sync(self._h))

    def set_frames_in_flight(self, frames):
        _check(lib().vx_set_frames_in_flight(self._h, frames))

    def wait_event(self, hip_event):
        """The next render waits for this raw hipEvent_t (e.g. torch.cuda.Event.cuda_event)."""
        _check(lib().vx_wait_event(self._h, _vp(hip_event)))

    def stream_wait_render(self, stream):
        """Makes the raw hipStream_t `stream` wait for the most recently issued render."""
        _check(lib().vx_stream_wait_render(self._h, _vp(stream)))

    def timeline(self):
        """Per wave of the last launch: [start, queue empty, exit] in 10 ns ticks, sub-tiles taken | phases | ticks in them, then the
        wave's life and its traversal loop in shader cycles and the loop's trips (voxel_hip.h: vx_timeline_read; needs VX_TIMELINE=1)."""
        out = np.zeros((8192, 8), dtype=np.uint64)
        n = lib().vx_timeline_read(self._h, out.ctypes.data_as(_vp), 8192)
        return out[:n]

    def knobs(self):
        """vx_debug_knobs: the scheduling knobs the context runs with."""
        out = (_u32 * 8)()
        _check(lib().vx_debug_knobs(self._h, C.byref(out)))
        return dict(zip(("refill_min", "service_min", "waves_per_cu", "queue_stripe", "tile_strip", "hot_first", "comm_headroom", "measurement_build"), [int(v) for v in out]))

    def image_info(self):
        out = (_u64 * 4)()
        _check(lib().vx_image_info(self._h, C.byref(out)))
        return {"layout": int(out[0]), "image_bytes": int(out[1]), "origin_bytes": int(out[2]), "chunks": int(out[3])}

    def excursion_counters(self, reset=True, stop=False):
        """Reads the counters of the walks inside voxels. reset=True zeroes them and (re)starts the counting -- it is off until first asked for, it
        costs frame time --; stop=True zeroes them and switches it off again."""
        out = (_u64 * 4)()
        _check(lib().vx_excursion_counters(self._h, C.byref(out), 2 if stop else int(reset)))
        return {"rays": int(out[0]), "started_over": int(out[1]), "service_phases": int(out[2]), "iterations_on_bytes": int(out[3])}

    def clock_probe(self, microseconds=200):
        """The shader clock (MHz) the device runs at during the call (vx_clock_probe); callable from a second thread while frames render."""
        mhz = C.c_double(0)
        _check(lib().vx_clock_probe(self._h, microseconds, C.byref(mhz)))
        return mhz.value

    def profile_enable(self, on=True):
        _check(lib().vx_profile_enable(self._h, int(on)))

    def profile_read(self):
        ms, n = C.c_double(0), _u32(0)
        _check(lib().vx_profile_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    @property
    def stream(self):
        return lib().vx_stream(self._h)
