"""ctypes binding of libvoxelhost.so: chunks and world SVOs (the reference's `world::chunk` / `world::hds`).

Mirrors the Rust call sequence the reference's tests use (src/graphics/svo_shader_tests.rs:78-115):

    chunk = Chunk(0, 0, 0, lod=5); chunk.set_block(31, 0, 0, 1); chunk.compact()
    world = World(SVO_ESVO); world.set_chunk((0, 0, 0), chunk); world.serialize()
    frame = world.frame()        # bytes of the mapped world buffer: [f32 2^-depth][preamble/root_ptr][arena]
"""
import ctypes as C

import numpy as np

from .build import lib_path, share_hip_runtime_with_torch

SVO_ESVO = 1  # SvoType::Esvo, shader define "1" (src/graphics/svo.rs:35)
SVO_CSVO = 2  # SvoType::Csvo, shader define "2" (src/graphics/svo.rs:36)

_lib = None


def lib():
    global _lib
    if _lib is None:
        share_hip_runtime_with_torch()  # libvoxelhost links libvoxelhip, which links the HIP runtime
        L = C.CDLL(str(lib_path("libvoxelhost.so")))
        vp, u32, i32, u64, sz = C.c_void_p, C.c_uint32, C.c_int32, C.c_uint64, C.c_size_t
        sig = {
            "vxh_chunk_new": (vp, [i32, i32, i32, u32]),
            "vxh_chunk_free": (None, [vp]),
            "vxh_chunk_set_block": (None, [vp, u32, u32, u32, u32]),
            "vxh_chunk_get_block": (u32, [vp, u32, u32, u32]),
            "vxh_chunk_compact": (None, [vp]),
            "vxh_chunk_set_lod": (None, [vp, u32]),
            "vxh_chunk_fill_dense": (None, [vp, vp]),
            "vxh_chunk_pos_hash": (u64, [i32, i32, i32]),
            "vxh_world_new": (vp, [C.c_int]),
            "vxh_world_free": (None, [vp]),
            "vxh_world_set_chunk": (C.c_int, [vp, u32, u32, u32, vp, C.c_int]),
            "vxh_world_serialize": (None, [vp]),
            "vxh_world_depth": (u32, [vp]),
            "vxh_world_size_in_bytes": (sz, [vp]),
            "vxh_world_header_bytes": (sz, [vp]),
            "vxh_world_write_to": (sz, [vp, vp]),
            "vxh_world_write_changes_to": (C.c_int, [vp, vp, sz, C.c_int]),
            "vxh_world_updated_ranges": (sz, [vp, vp, sz]),
            "vxh_world_frame": (sz, [vp, vp, sz]),
            "vxh_scene_build_heightfield": (u64, [vp, u32, u32, u32, vp, vp]),
            "vxh_picker_serialize": (u32, [vp, u32, vp, u32, vp, u32]),
            "vxh_picker_deserialize": (None, [vp, u32, vp, u32, vp, vp, vp]),
            "vxh_stream_new": (vp, [C.c_int, u32, u32, u32, C.c_int32, C.c_int32]),
            "vxh_stream_free": (None, [vp]),
            "vxh_stream_set_no_lod": (None, [vp, C.c_int]),
            "vxh_stream_move_to": (u64, [vp, C.c_float, C.c_float, C.c_float]),
            "vxh_stream_move_to_view": (u64, [vp, C.c_float, C.c_float, C.c_float, vp]),
            "vxh_stream_pump": (C.c_int, [vp, vp, u32, vp]),
            "vxh_stream_pump_mode": (C.c_int, [vp, vp, u32, C.c_int, vp]),
            "vxh_stream_frame": (sz, [vp, vp, sz]),
            "vxh_stream_mirror_image": (C.c_int, [vp, C.c_uint64, C.c_int]),
            "vxh_stream_image": (C.c_uint64, [vp, vp, C.c_uint64]),
            "vxh_oct64_same_tree": (C.c_int, [vp, C.c_uint64, vp, C.c_uint64]),
            "vxh_stream_to_svo": (None, [vp, vp, vp]),
            "vxh_stream_resident_chunks": (u64, [vp]),
            "vxh_physics_step_many": (C.c_int64, [vp, C.c_float, u32, vp, u32]),
            "vxh_physics_update": (None, [C.c_float, vp, vp, u32]),
            "vxh_reference_render_test": (C.c_int, [C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_double), C.c_char_p, sz]),
            "vxh_mapper_raycast_test": (C.c_int, [C.c_int, u32, vp, vp, u32, vp, vp, u32, C.c_float, vp, vp, C.c_char_p, sz]),
            "vxh_scene_height": (u32, [u32, u32, u32, u32]),
            "vxh_scene_hash32": (u32, [u32, u32, u32, u32]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


class Chunk:
    """32^3 voxel chunk (src/world/chunk.rs:94-131). Storage starts pre-expanded to depth 5."""

    def __init__(self, x=0, y=0, z=0, lod=5):
        self._h = lib().vxh_chunk_new(x, y, z, lod)
        if not self._h:
            raise MemoryError("vxh_chunk_new failed")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().vxh_chunk_free(self._h)
            self._h = None

    def set_block(self, x, y, z, block):
        lib().vxh_chunk_set_block(self._h, x, y, z, block)

    def get_block(self, x, y, z):
        return lib().vxh_chunk_get_block(self._h, x, y, z)

    def compact(self):
        lib().vxh_chunk_compact(self._h)

    def set_lod(self, lod):
        lib().vxh_chunk_set_lod(self._h, lod)

    def fill_dense(self, ids):
        """ids: uint32 array of shape (32, 32, 32) indexed [z][y][x]; 0 = empty (Chunk::fill_with)."""
        a = np.ascontiguousarray(ids, dtype=np.uint32)
        assert a.size == 32 * 32 * 32
        lib().vxh_chunk_fill_dense(self._h, a.ctypes.data_as(C.c_void_p))

    def apply_blocks(self, blocks):
        """blocks: list of [x, y, z, id] or {"box": [[x0,x1],[y0,y1],[z0,z1]], "id": id} (golden fixture format)."""
        for b in blocks:
            if isinstance(b, dict):
                (x0, x1), (y0, y1), (z0, z1) = b["box"]
                for x in range(x0, x1):
                    for z in range(z0, z1):
                        for y in range(y0, y1):
                            self.set_block(x, y, z, b["id"])
            else:
                self.set_block(*b)


class World:
    """World-level SVO (`Esvo<SerializedChunk>` / `Csvo`, src/world/hds/common.rs:3-15)."""

    def __init__(self, svo_type):
        self.svo_type = svo_type
        self._h = lib().vxh_world_new(svo_type)
        if not self._h:
            raise ValueError(f"bad svo_type {svo_type}")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().vxh_world_free(self._h)
            self._h = None

    def set_chunk(self, svo_pos, chunk, serialize=True):
        lib().vxh_world_set_chunk(self._h, svo_pos[0], svo_pos[1], svo_pos[2], chunk._h, int(serialize))

    def serialize(self):
        lib().vxh_world_serialize(self._h)

    @property
    def depth(self):
        return lib().vxh_world_depth(self._h)

    @property
    def size_in_bytes(self):
        return lib().vxh_world_size_in_bytes(self._h)

    @property
    def header_bytes(self):
        return lib().vxh_world_header_bytes(self._h)

    def write_to(self):
        buf = np.zeros(self.header_bytes + self.size_in_bytes, dtype=np.uint8)
        n = lib().vxh_world_write_to(self._h, buf.ctypes.data_as(C.c_void_p))
        return buf[:n]

    def write_changes_to(self, dst_ptr, dst_len, reset=True):
        """Writes header + dirty ranges at `dst_ptr` (an address), like WorldSvo::write_changes_to."""
        rc = lib().vxh_world_write_changes_to(self._h, C.c_void_p(dst_ptr), dst_len, int(reset))
        if rc != 0:
            raise RuntimeError("dst is not large enough")  # the reference asserts (esvo.rs:328)

    def updated_ranges(self):
        n = lib().vxh_world_updated_ranges(self._h, None, 0)
        out = np.zeros((max(n, 1), 2), dtype=np.uint64)
        lib().vxh_world_updated_ranges(self._h, out.ctypes.data_as(C.c_void_p), n)
        return [(int(a), int(b)) for a, b in out[:n]]

    def frame(self, pad_words=4):
        """[f32 2^-depth][header][arena] as uint32 words (+ zero padding: CSVO's read_uint touches word+1)."""
        need = lib().vxh_world_frame(self._h, None, 0)
        words = (need + 3) // 4 + pad_words
        buf = np.zeros(words, dtype=np.uint32)
        lib().vxh_world_frame(self._h, buf.ctypes.data_as(C.c_void_p), words * 4)
        return buf

    def write_frame_to(self, ptr, capacity):
        """frame() written at `ptr` (room for `capacity` bytes: a context's staging buffer); returns the bytes written."""
        need = lib().vxh_world_frame(self._h, None, 0)
        if need > capacity:
            raise ValueError("frame larger than the buffer: %d > %d" % (need, capacity))
        return int(lib().vxh_world_frame(self._h, C.c_void_p(ptr), capacity))

    def build_heightfield(self, depth, seed=0x5EED0001, threads=0):
        """Seeded synthetic scene; returns dict(chunks, leaves, h_max)."""
        import os

        leaves, hmax = C.c_uint64(0), C.c_uint32(0)
        threads = threads or min(32, os.cpu_count() or 1)
        chunks = lib().vxh_scene_build_heightfield(self._h, depth, seed, threads, C.byref(leaves), C.byref(hmax))
        return dict(chunks=int(chunks), leaves=int(leaves.value), h_max=int(hmax.value))


def scene_height(depth, seed, x, z):
    return lib().vxh_scene_height(depth, seed, x, z)


PICKER_TASK_DTYPE = np.dtype([("max_dst", "<f4"), ("_p0", "<f4", 3), ("pos", "<f4", 3), ("_p1", "<f4"), ("dir", "<f4", 3), ("_p2", "<f4")])
PICKER_RESULT_DTYPE = np.dtype([("dst", "<f4"), ("inside_voxel", "<u4"), ("_p0", "<f4", 2), ("pos", "<f4", 3), ("_p1", "<f4"), ("normal", "<f4", 3),
                                ("_p2", "<f4")])


def _batch_arrays(rays, aabbs):
    r = np.array([list(x["pos"]) + list(x["dir"]) + [x["max_dst"]] for x in rays], dtype=np.float32).reshape(-1, 7)
    a = np.array([list(x["pos"]) + list(x["offset"]) + list(x["extents"]) for x in aabbs], dtype=np.float32).reshape(-1, 9)
    return np.ascontiguousarray(r), np.ascontiguousarray(a)


def picker_serialize(rays, aabbs):
    """PickerBatch::serialize_tasks (src/graphics/svo_picker.rs:63-80)."""
    r, a = _batch_arrays(rays, aabbs)
    n = lib().vxh_picker_serialize(r.ctypes.data_as(C.c_void_p), len(rays), a.ctypes.data_as(C.c_void_p), len(aabbs), None, 0)
    out = np.zeros(n, dtype=PICKER_TASK_DTYPE)
    lib().vxh_picker_serialize(r.ctypes.data_as(C.c_void_p), len(rays), a.ctypes.data_as(C.c_void_p), len(aabbs), out.ctypes.data_as(C.c_void_p), n)
    return out


def picker_deserialize(rays, aabbs, results):
    """PickerBatch::deserialize_results (svo_picker.rs:84-104): (rays n x 8 floats, aabbs n x 6 floats)."""
    r, a = _batch_arrays(rays, aabbs)
    res = np.ascontiguousarray(results, dtype=PICKER_RESULT_DTYPE)
    out_r = np.zeros((len(rays), 8), dtype=np.float32)
    out_a = np.zeros((len(aabbs), 6), dtype=np.float32)
    lib().vxh_picker_deserialize(r.ctypes.data_as(C.c_void_p), len(rays), a.ctypes.data_as(C.c_void_p), len(aabbs), res.ctypes.data_as(C.c_void_p),
                                 out_r.ctypes.data_as(C.c_void_p), out_a.ctypes.data_as(C.c_void_p))
    return out_r, out_a


class WorldStreamer:
    """Chunk loader -> generated heightfield chunks at their LOD -> SVO leaves -> dirty ranges -> vx_commit
    (csrc/host/stream.hpp; src/systems/chunkloader.rs + src/systems/worldsvo.rs:133-196)."""

    PUMP_FIELDS = ("events", "loads", "unloads", "lod_changes", "ranges", "bytes", "arena_bytes", "pending", "build_us", "apply_us", "commit_us")

    def __init__(self, svo_type, scene_depth, radius, start_y, end_y, seed=0x5EED0001, no_lod=False):
        self._h = lib().vxh_stream_new(svo_type, scene_depth, seed, radius, start_y, end_y)
        if not self._h:
            raise ValueError("bad streamer parameters")
        if no_lod:  # (--no-lod of the reference: every chunk at full detail, world.rs:141,151)
            lib().vxh_stream_set_no_lod(self._h, 1)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().vxh_stream_free(self._h)
            self._h = None

    def move_to(self, x, y, z, forward=None, up=(0.0, 1.0, 0.0), fov_y_deg=72.0, aspect=16.0 / 9.0, near=0.01, far=1024.0):
        """Queues the chunk events of the target moving to this world position; returns how many. With `forward` (the camera's
        view direction) the events are ordered like the reference orders them (src/gamelogic/world.rs:233-262): chunks in the
        view frustum first, the rest from front to back; without, nearest first (the chunk loader's own order)."""
        if forward is None:
            return int(lib().vxh_stream_move_to(self._h, x, y, z))
        view = (C.c_float * 10)(*forward, *up, fov_y_deg, aspect, near, far)
        return int(lib().vxh_stream_move_to_view(self._h, x, y, z, view))

    def pump(self, svo_handle, max_events=400, wait=True):
        """Applies up to max_events queued events and commits the dirty ranges to the vx context (needs a GPU). Chunks are built
        by background workers from the moment move_to queues their events; wait=False applies only those that are finished (a
        frame loop: the rest arrive with later frames), wait=True waits for them (the same events per call on any machine)."""
        out = (C.c_uint64 * 11)()
        if lib().vxh_stream_pump_mode(self._h, svo_handle, max_events, int(wait), out) != 0:
            raise RuntimeError("stream pump failed (capacity exceeded or HIP error)")
        return dict(zip(self.PUMP_FIELDS, (int(v) for v in out)))

    def frame(self, pad_words=4):
        """The whole current world as one frame (what a full upload would send), uint32 words."""
        need = lib().vxh_stream_frame(self._h, None, 0)
        buf = np.zeros((need + 3) // 4 + pad_words, dtype=np.uint32)
        lib().vxh_stream_frame(self._h, buf.ctypes.data_as(C.c_void_p), buf.size * 4)
        return buf

    def mirror_image(self, capacity_bytes, layout=1):
        """Tests: pump(None) from now on applies its dirty ranges to a host mirror like vx_commit does to the staging buffer
        and keeps a traversal image of it up to date incrementally (what a context does next to its device world buffer)."""
        if lib().vxh_stream_mirror_image(self._h, capacity_bytes, layout) != 0:
            raise ValueError("bad mirror parameters")

    def image(self):
        """The incrementally maintained traversal image (uint32 words)."""
        n = lib().vxh_stream_image(self._h, None, 0)
        if n == 0:
            raise ValueError("no image (mirror_image() not called, nothing committed yet, or the world cannot be imaged)")
        out = np.zeros(n, dtype=np.uint32)
        lib().vxh_stream_image(self._h, out.ctypes.data_as(C.c_void_p), n)
        return out

    def to_svo(self, world_pos):
        w = (C.c_float * 3)(*world_pos)
        s = (C.c_float * 3)()
        lib().vxh_stream_to_svo(self._h, w, s)
        return tuple(s)

    @property
    def resident_chunks(self):
        return int(lib().vxh_stream_resident_chunks(self._h))


def oct64_same_tree(a, b):
    """Do two layout-1 traversal images hold the same tree (wherever their octants are placed)?"""
    a, b = np.ascontiguousarray(a, dtype=np.uint32), np.ascontiguousarray(b, dtype=np.uint32)
    r = lib().vxh_oct64_same_tree(a.ctypes.data_as(C.c_void_p), a.size, b.ctypes.data_as(C.c_void_p), b.size)
    if r < 0:
        raise ValueError("an image pointer is out of range")
    return bool(r)


ENTITY_FLOATS = 17  # position, velocity, aabb offset, aabb extents, wall_clip, flying, gravity, max_fall_velocity, is_grounded


def make_entities(positions, extents=(0.8, 1.8, 0.8), offset=(-0.4, 0.0, -0.4), gravity=60.0, max_fall_velocity=100.0):
    """Entity records for physics_step_many (src/systems/physics.rs:10-75)."""
    e = np.zeros((len(positions), ENTITY_FLOATS), dtype=np.float32)
    e[:, 0:3] = positions
    e[:, 6:9] = offset
    e[:, 9:12] = extents
    e[:, 14] = gravity
    e[:, 15] = max_fall_velocity
    return e


def physics_step_many(svo_handle, delta_time, steps, entities):
    """Physics::step_many (physics.rs:122-136) `steps` times over vx_raycast of the given context (needs a GPU).
    Updates `entities` in place; returns the number of picker tasks cast."""
    e = np.ascontiguousarray(entities, dtype=np.float32)
    n = lib().vxh_physics_step_many(svo_handle, delta_time, steps, e.ctypes.data_as(C.c_void_p), len(e))
    if n < 0:
        raise RuntimeError("vxh_physics_step_many failed")
    entities[...] = e
    return int(n)


def physics_update(delta_time, entities, aabb_results):
    """Physics::update_entity (physics.rs:139-170) with given AabbResults (n x 6: neg, pos)."""
    e = np.ascontiguousarray(entities, dtype=np.float32)
    r = np.ascontiguousarray(aabb_results, dtype=np.float32)
    lib().vxh_physics_update(delta_time, e.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), len(e))
    entities[...] = e


def reference_render_test(svo_type, texture_dir, expected_png, actual_png_out=""):
    """src/graphics/svo.rs:342-399 through the C++ mirror of graphics::Svo (needs a GPU). Returns the diff fraction."""
    diff = C.c_double(0)
    err = C.create_string_buffer(512)
    rc = lib().vxh_reference_render_test(svo_type, str(texture_dir).encode(), str(expected_png).encode(), str(actual_png_out).encode(), C.byref(diff), err, 512)
    if rc != 0:
        raise RuntimeError(err.value.decode())
    return diff.value


def mapper_raycast_test(svo_type, render_distance, chunks, centers, xz, y0):
    """worldsvo::Svo end to end (needs a GPU). chunks: [(cx, cy, cz, floor_height)], centers: two ChunkPos. Returns two (n, 2) arrays {dst, pos.y}."""
    cp = np.ascontiguousarray([c[:3] for c in chunks], dtype=np.int32)
    fh = np.ascontiguousarray([c[3] for c in chunks], dtype=np.uint32)
    cen = np.ascontiguousarray(centers, dtype=np.int32).reshape(6)
    pts = np.ascontiguousarray(xz, dtype=np.float32).reshape(-1, 2)
    a = np.zeros((len(pts), 2), dtype=np.float32)
    b = np.zeros((len(pts), 2), dtype=np.float32)
    err = C.create_string_buffer(512)
    rc = lib().vxh_mapper_raycast_test(svo_type, render_distance, cp.ctypes.data_as(C.c_void_p), fh.ctypes.data_as(C.c_void_p), len(chunks),
                                       cen.ctypes.data_as(C.c_void_p), pts.ctypes.data_as(C.c_void_p), len(pts), y0, a.ctypes.data_as(C.c_void_p),
                                       b.ctypes.data_as(C.c_void_p), err, 512)
    if rc != 0:
        raise RuntimeError(err.value.decode())
    return a, b
