"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

ctypes binding of oracle/_build/liboracle.so (the C restatement of the reference's GLSL ray path, see
svo_oracle.h). Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
SO = HERE / "_build" / "liboracle.so"


class Material(C.Structure):
    _fields_ = [("specular_pow", C.c_float), ("specular_strength", C.c_float), ("tex_top", C.c_int32), ("tex_side", C.c_int32),
                ("tex_bottom", C.c_int32), ("tex_top_normal", C.c_int32), ("tex_side_normal", C.c_int32), ("tex_bottom_normal", C.c_int32)]


MATERIAL_DTYPE = np.dtype([("specular_pow", "<f4"), ("specular_strength", "<f4"), ("tex_top", "<i4"), ("tex_side", "<i4"), ("tex_bottom", "<i4"),
                           ("tex_top_normal", "<i4"), ("tex_side_normal", "<i4"), ("tex_bottom_normal", "<i4")])


class Textures(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("layers", C.c_uint32), ("levels", C.c_uint32), ("level", C.c_void_p * 16)]


class Scene(C.Structure):
    _fields_ = [("svo_type", C.c_int), ("world", C.c_void_p), ("world_words", C.c_size_t), ("materials", C.c_void_p), ("n_materials", C.c_uint32),
                ("tex", Textures)]


class Result(C.Structure):
    _fields_ = [("t", C.c_float), ("value", C.c_uint32), ("face_id", C.c_int32), ("pos", C.c_float * 3), ("uv", C.c_float * 2),
                ("color", C.c_float * 4), ("lod", C.c_float), ("inside_voxel", C.c_int32)]


FRAME_DTYPE = np.dtype([("t_min", "<f4"), ("ptr", "<u4"), ("idx", "<u4"), ("parent_octant_idx", "<u4"), ("scale", "<i4"), ("is_child", "<i4"),
                        ("is_leaf", "<i4"), ("crossed_boundary", "<i4"), ("next_ptr", "<u4")])

COUNTER_FIELDS = ["rays", "iterations", "pushes", "leaf_tests", "leaf_tests_trilinear", "boundaries", "csvo_header_bytes", "csvo_pointer_bytes"]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_FIELDS]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_FIELDS}


class Uniforms(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("fovy", C.c_float), ("aspect", C.c_float), ("ambient", C.c_float), ("light_dir", C.c_float * 3),
                ("cam_pos", C.c_float * 3), ("render_shadows", C.c_int32), ("shadow_distance", C.c_float), ("highlight_pos", C.c_float * 3)]


HIT_DTYPE = np.dtype([("t", "<f4"), ("value", "<u4"), ("face_id", "<i4"), ("flags", "<u4"), ("pos", "<f4", 3), ("lod", "<f4"), ("uv", "<f4", 2),
                      ("shadow_t", "<f4"), ("steps", "<u4")])
PICKER_TASK_DTYPE = np.dtype([("max_dst", "<f4"), ("_p0", "<f4", 3), ("pos", "<f4", 3), ("_p1", "<f4"), ("dir", "<f4", 3), ("_p2", "<f4")])
PICKER_RESULT_DTYPE = np.dtype([("dst", "<f4"), ("inside_voxel", "<u4"), ("_p0", "<f4", 2), ("pos", "<f4", 3), ("_p1", "<f4"), ("normal", "<f4", 3),
                                ("_p2", "<f4")])
assert HIT_DTYPE.itemsize == 48 and PICKER_TASK_DTYPE.itemsize == 48 and PICKER_RESULT_DTYPE.itemsize == 48 and FRAME_DTYPE.itemsize == 36

_lib = None


def build():
    r = subprocess.run(["make", "-C", str(HERE), "all"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout)


def lib():
    global _lib
    if _lib is None:
        if not SO.exists():
            build()
        L = C.CDLL(str(SO))
        vp = C.c_void_p
        L.or_intersect.restype = None
        L.or_intersect.argtypes = [C.POINTER(Scene), C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 3), C.c_float, C.c_int, C.POINTER(Result), vp,
                                   C.c_int, C.POINTER(C.c_int), vp]
        L.or_texture_lod.restype = None
        L.or_texture_lod.argtypes = [C.POINTER(Textures), C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float * 4)]
        L.or_build_mips.restype = C.c_size_t
        L.or_build_mips.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]
        L.or_render.restype = None
        L.or_render.argtypes = [C.POINTER(Scene), C.POINTER(Uniforms)] + [C.c_uint32] * 6 + [vp, vp, vp, C.c_int]
        L.or_primary_ray.restype = None
        L.or_primary_ray.argtypes = [C.POINTER(Uniforms)] + [C.c_uint32] * 4 + [C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 3)]
        L.or_picker.restype = None
        L.or_picker.argtypes = [C.POINTER(Scene), vp, C.c_uint32, vp, C.c_int]
        L.or_csvo_read_uint.restype = C.c_uint32
        L.or_csvo_read_uint.argtypes = [vp, C.c_size_t, C.c_uint32]
        L.or_csvo_read_next_ptr.restype = C.c_uint32
        L.or_csvo_read_next_ptr.argtypes = [vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_int)]
        L.or_csvo_read_leaf.restype = C.c_uint32
        L.or_csvo_read_leaf.argtypes = [vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.or_max_threads.restype = C.c_int
        _lib = L
    return _lib


def mip_chain(base, levels):
    """base: uint8 [layers][h][w][4] (row 0 = bottom). Returns the list of level arrays (level 0 = base)."""
    base = np.ascontiguousarray(base, dtype=np.uint8)
    layers, h, w, _ = base.shape
    out = [base]
    if levels > 1:
        total = sum(layers * max(h >> l, 1) * max(w >> l, 1) * 4 for l in range(1, levels))
        buf = np.zeros(total, dtype=np.uint8)
        n = lib().or_build_mips(base.ctypes.data_as(C.c_void_p), w, h, layers, levels, buf.ctypes.data_as(C.c_void_p))
        assert n == total
        off = 0
        for l in range(1, levels):
            hl, wl = max(h >> l, 1), max(w >> l, 1)
            out.append(buf[off:off + layers * hl * wl * 4].reshape(layers, hl, wl, 4))
            off += layers * hl * wl * 4
    return out


class OracleScene:
    """Keeps the numpy buffers alive behind an `or_scene`."""

    def __init__(self, svo_type, world_words, materials, tex_base, mip_levels):
        self.world = np.ascontiguousarray(world_words, dtype=np.uint32)
        self.materials = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
        self.levels = mip_chain(tex_base, mip_levels)
        s = Scene()
        s.svo_type = svo_type
        s.world = self.world.ctypes.data
        s.world_words = self.world.size
        s.materials = self.materials.ctypes.data
        s.n_materials = self.materials.size
        layers, h, w, _ = self.levels[0].shape
        s.tex.width, s.tex.height, s.tex.layers, s.tex.levels = w, h, layers, len(self.levels)
        for i, lv in enumerate(self.levels):
            s.tex.level[i] = lv.ctypes.data
        self.c = s

    def intersect(self, pos, direction, max_dst, cast_translucent, max_frames=0, counters=None):
        ro = (C.c_float * 3)(*pos)
        rd = (C.c_float * 3)(*direction)
        res = Result()
        frames = np.zeros(max(max_frames, 1), dtype=FRAME_DTYPE)
        n = C.c_int(0)
        lib().or_intersect(C.byref(self.c), C.byref(ro), C.byref(rd), max_dst, int(cast_translucent), C.byref(res),
                           frames.ctypes.data_as(C.c_void_p) if max_frames else None, max_frames, C.byref(n),
                           C.byref(counters) if counters is not None else None)
        return res, frames[:min(n.value, max_frames)], n.value

    def render(self, uniforms, w, h, rect=None, want_hits=True, counters=None, threads=0):
        x0, y0, x1, y1 = rect or (0, 0, w, h)
        img = np.zeros((h, w, 4), dtype=np.float32)
        hits = np.zeros((h, w), dtype=HIT_DTYPE) if want_hits else None
        lib().or_render(C.byref(self.c), C.byref(uniforms), w, h, x0, y0, x1, y1, img.ctypes.data_as(C.c_void_p),
                        hits.ctypes.data_as(C.c_void_p) if want_hits else None, C.byref(counters) if counters is not None else None,
                        threads or lib().or_max_threads())
        return img, hits

    def picker(self, tasks, threads=1):
        tasks = np.ascontiguousarray(tasks, dtype=PICKER_TASK_DTYPE)
        out = np.zeros(tasks.size, dtype=PICKER_RESULT_DTYPE)
        lib().or_picker(C.byref(self.c), tasks.ctypes.data_as(C.c_void_p), tasks.size, out.ctypes.data_as(C.c_void_p), threads)
        return out


def normalize(v):
    """cgmath `Vector3::normalize` as the reference's test harness applies it (svo_shader_tests.rs:246), in fp32."""
    v = np.asarray(v, dtype=np.float32)
    mag = np.sqrt(np.float32(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), dtype=np.float32)
    return (v / mag).astype(np.float32)


def look_to_rh_inverse(eye, fwd, up):
    """u_view of src/graphics/svo.rs:197: inverse of cgmath 0.18 `Matrix4::look_to_rh(eye, dir, up)`.

    look_to_rh builds rows (s, u, -f) with translation (-eye.s, -eye.u, eye.f); for that orthonormal basis
    the inverse has columns [s, u, -f, eye]. Returned column-major as 16 floats (fp32)."""
    f = normalize(fwd)
    s = normalize(np.cross(f, np.asarray(up, dtype=np.float32)).astype(np.float32))
    u = np.cross(s, f).astype(np.float32)
    m = np.zeros(16, dtype=np.float32)
    m[0:3], m[4:7], m[8:11], m[12:15] = s, u, -f, np.asarray(eye, dtype=np.float32)
    m[15] = 1.0
    return m


def make_uniforms(view, fovy, aspect, ambient, light_dir, cam_pos, render_shadows, shadow_distance, highlight_pos=None):
    u = Uniforms()
    u.view = (C.c_float * 16)(*[float(x) for x in view])
    u.fovy, u.aspect, u.ambient = float(fovy), float(aspect), float(ambient)
    u.light_dir = (C.c_float * 3)(*[float(x) for x in light_dir])
    u.cam_pos = (C.c_float * 3)(*[float(x) for x in cam_pos])
    u.render_shadows = int(render_shadows)
    u.shadow_distance = float(shadow_distance)
    hp = highlight_pos if highlight_pos is not None else (float("nan"),) * 3  # src/graphics/svo.rs:211
    u.highlight_pos = (C.c_float * 3)(*[float(x) for x in hp])
    return u
