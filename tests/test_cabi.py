"""The C-ABI library loads and exports every symbol include/voxel_hip.h declares (no GPU needed: no compute calls)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np

from helpers import vra  # noqa: F401  (loads the package alias)

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "voxel_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vx_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_surface():
    names = declared_symbols()
    for required in ("vx_create", "vx_destroy", "vx_set_materials", "vx_set_textures", "vx_staging_ptr", "vx_commit", "vx_render", "vx_raycast",
                     "vx_get_stats", "vx_sync", "vx_last_error"):
        assert required in names


def test_library_exports_every_declared_symbol():
    from voxel_rs_amd import hip

    L = C.CDLL(str(ROOT / "voxel-rs_amd" / "lib" / "libvoxelhip.so"))
    for name in declared_symbols():
        assert hasattr(L, name), f"{name} declared in voxel_hip.h but not exported"
    # the Python binding covers the same set, so a renamed symbol cannot silently drop out of the tests
    assert sorted(hip.SYMBOLS) == declared_symbols()
    assert hip.lib().vx_version().startswith(b"voxel-hip")


def test_struct_layouts_match_the_reference_std430_layouts():
    from voxel_rs_amd import hip

    # PickerTask / PickerResult: 48 bytes, vec3 at 16 and 32 (assets/shaders/picker.glsl:9-27)
    assert hip.PICKER_TASK_DTYPE.fields["pos"][1] == 16 and hip.PICKER_TASK_DTYPE.fields["dir"][1] == 32
    assert hip.PICKER_RESULT_DTYPE.fields["inside_voxel"][1] == 4 and hip.PICKER_RESULT_DTYPE.fields["pos"][1] == 16
    assert hip.PICKER_RESULT_DTYPE.fields["normal"][1] == 32
    assert hip.MATERIAL_DTYPE.itemsize == 32  # MaterialInstance (svo_registry.rs:29-40)
    assert C.sizeof(hip.Uniforms) == 4 * (16 + 3 + 3 + 3 + 1 + 1 + 3)
    assert hip.FRAME_DTYPE.itemsize == 36  # StackFrame (svo.test.glsl:23-33)


def test_create_without_gpu_fails_loudly():
    """There is no CPU fallback: without a HIP device vx_create reports VX_ERR_NO_DEVICE."""
    import torch
    from voxel_rs_amd import hip

    if torch.cuda.device_count() > 0:
        return  # on the GPU box the parity tests exercise vx_create
    h = C.c_void_p()
    rc = hip.lib().vx_create(1, 1 << 20, 0, C.byref(h))
    assert rc == 2 and not h.value
    assert b"no HIP device" in hip.lib().vx_last_error()


def test_local_tile_count():
    from voxel_rs_amd import hip

    assert hip.local_tile_count(1920, 1080, 0, 1) == 60 * 34
    assert sum(hip.local_tile_count(1920, 1080, r, 8) for r in range(8)) == 60 * 34
    assert hip.local_tile_count(200, 120, 2, 3) == (7 * 4 - 2 + 2) // 3
    assert np.isfinite(1.0)


def test_tile_order_is_a_morton_sequence_shared_with_the_host_side():
    """vx_tile_order (what vx_render and vx_assemble_tiles use) = voxel_rs_amd.sharding.tile_order (what the host-side sharding
    helpers and the gloo tests use); consecutive places are neighbours on the screen (Z-order)."""
    from voxel_rs_amd import hip, sharding

    for w, h in ((1920, 1080), (3840, 2160), (7680, 4320), (200, 120), (33, 31), (64, 64), (31, 2000)):
        order = hip.tile_order(w, h)
        assert list(order) == sharding.tile_order(w, h)
        assert sorted(order) == list(range(len(order)))
    order = hip.tile_order(128, 128)  # 4 x 4 tiles: the textbook Z
    assert list(order) == [0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15]
    # eight ranks at 1080p: every rank's tiles are spread over the whole screen (no rank owns a band)
    tx = 60
    for r in range(8):
        mine = hip.tile_order(1920, 1080)[r::8]
        assert (mine % tx).min() < 8 and (mine % tx).max() > 51 and (mine // tx).min() < 4 and (mine // tx).max() > 29


def test_comm_entry_points_reject_bad_arguments():
    """The RCCL side of the ABI (vx_comm_*, vx_gather_tiles, vx_wait_gather) exists and fails cleanly without a context, a
    communicator or a GPU -- no call reaches RCCL here."""
    from voxel_rs_amd import hip

    L = hip.lib()
    assert L.vx_comm_unique_id(None, 128) == 1 and b"128" in L.vx_last_error()
    small = C.create_string_buffer(16)
    assert L.vx_comm_unique_id(small, 16) == 1
    ident = C.create_string_buffer(hip.VX_COMM_ID_BYTES)
    assert L.vx_comm_init(None, 2, 0, ident) == 1
    assert L.vx_comm_destroy(None) == 1
    assert L.vx_gather_tiles(None, None, 0, None, 0, None) == 1
    assert L.vx_wait_gather(None, 0) == 1
    assert L.vx_comm_stream(None) is None
    n, r = C.c_int(-1), C.c_int(-1)
    assert L.vx_comm_info(None, C.byref(n), C.byref(r)) == 1
    assert L.vx_present_wait(None, 0, None, None) == 1
    assert L.vx_set_comm_headroom(None, 2) == 1 and L.vx_debug_knobs(None, None) == 1
