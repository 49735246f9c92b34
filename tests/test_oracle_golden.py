"""Pins the CPU oracle (oracle/svo_oracle.c) to the reference's own golden vectors.

Every case replays a #[test] of /root/reference/src/graphics/svo_shader_tests.rs (ESVO :286-754, CSVO :756-1224)
through the C restatement: per-iteration traversal frames are compared exactly (ints) / to 1e-5 (t_min), results
with the tolerance the reference's assert macros state. Data: tests/golden/svo_shader_tests.json.
"""
import numpy as np
import pytest

from helpers import oracle_scene, orc

FMTS = ["esvo", "csvo"]
EPS = 1e-5  # assert_float_eq! default (src/graphics/macros.rs:103-114)


def f32(x):
    """The reference's expected values are Rust f32 literals: compare against their fp32 rounding."""
    return np.asarray(x, dtype=np.float32).astype(np.float64)


def check_result(res, exp, min_tol=0.0, what=""):
    assert abs(res.t - float(f32(exp["t"]))) <= max(exp["t_tol"], min_tol) + 1e-12, (what, "t", res.t, exp["t"])
    assert res.value == exp["value"], (what, "value", res.value)
    assert res.face_id == exp["face_id"], (what, "face", res.face_id)
    np.testing.assert_allclose(list(res.pos), f32(exp["pos"]), rtol=0, atol=max(exp["pos_tol"], min_tol) + 1e-12, err_msg=f"{what} pos")
    np.testing.assert_allclose(list(res.uv), f32(exp["uv"]), rtol=0, atol=max(exp["uv_tol"], min_tol) + 1e-12, err_msg=f"{what} uv")
    np.testing.assert_allclose(list(res.color), f32(exp["color"]), rtol=0, atol=max(exp["color_tol"], min_tol) + 1e-12, err_msg=f"{what} color")
    assert bool(res.inside_voxel) == exp["inside_voxel"], (what, "inside_voxel")


def check_frames(frames, n, expected, fmt):
    assert n == len(expected), f"frame count {n} != {len(expected)}"
    for i, (got, exp) in enumerate(zip(frames, expected)):
        assert abs(float(got["t_min"]) - float(f32(exp["t_min"]))) < EPS, (i, got, exp)
        for k in ("ptr", "idx", "parent_octant_idx", "scale", "is_child", "is_leaf", "crossed_boundary"):
            assert int(got[k]) == exp[k], (i, k, got, exp)
        if fmt == "csvo":
            # the boundary-crossing frame reports the absolute pointer with its flag bit cleared (0 for the first chunk)
            assert int(got["next_ptr"]) == exp["next_ptr"], (i, "next_ptr", got, exp)


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("case", ["shader_svo_traversal", "check_at_higher_coordinates"])
def test_traversal_frames(golden, fmt, case):
    g = golden["formats"][fmt][case]
    scene, _ = oracle_scene(golden, fmt, g["svo_pos"], g["blocks"])
    ray = g["ray"]
    res, frames, n = scene.intersect(ray["pos"], orc.normalize(ray["dir"]), ray["max_dst"], ray["cast_translucent"], max_frames=100)
    check_frames(frames, n, g["frames"], fmt)
    check_result(res, g["result"], what=case)


@pytest.mark.parametrize("fmt", FMTS)
def test_cast_inside_outside_all_axes(golden, fmt):
    g = golden["formats"][fmt]["cast_inside_outside_all_axes"]
    scene, _ = oracle_scene(golden, fmt, g["svo_pos"], g["blocks"])
    for case in g["cases"]:
        d = orc.normalize(case["dir"])
        res, _, _ = scene.intersect(case["pos"], d, g["max_dst"], g["cast_translucent"])
        check_result(res, case["expected"], min_tol=EPS, what=case["name"] + " inside")
        # the same ray from one unit further back: t grows by one, everything else stays (:482-487)
        pos = (np.asarray(case["pos"], dtype=np.float32) - d).astype(np.float32)
        exp = dict(case["expected"], t=case["expected"]["t"] + 1.0)
        res, _, _ = scene.intersect(pos, d, g["max_dst"], g["cast_translucent"])
        check_result(res, exp, min_tol=EPS, what=case["name"] + " outside")


@pytest.mark.parametrize("fmt", FMTS)
def test_uv_coords_on_all_sides(golden, fmt):
    g = golden["formats"][fmt]["uv_coords_on_all_sides"]
    scene, _ = oracle_scene(golden, fmt, g["svo_pos"], g["blocks"])
    for i, case in enumerate(g["cases"]):
        res, _, _ = scene.intersect(case["pos"], orc.normalize(case["dir"]), g["max_dst"], g["cast_translucent"])
        np.testing.assert_allclose(list(res.uv), f32(case["expected_uv"]), rtol=0, atol=EPS, err_msg=f"case {i} uv")
        np.testing.assert_allclose(list(res.color), f32(case["expected_color"]), rtol=0, atol=EPS, err_msg=f"case {i} color")


@pytest.mark.parametrize("fmt", FMTS)
def test_casting_against_translucent_leafs(golden, fmt):
    g = golden["formats"][fmt]["casting_against_translucent_leafs"]
    scene, _ = oracle_scene(golden, fmt, g["svo_pos"], g["blocks"])
    for case in g["cases"]:
        res, _, _ = scene.intersect(case["pos"], orc.normalize(case["dir"]), g["max_dst"], case["cast_translucent"])
        check_result(res, case["expected"], what=case["name"])


@pytest.mark.parametrize("fmt", FMTS)
def test_detect_inside_leaf_voxel(golden, fmt):
    g = golden["formats"][fmt]["detect_inside_leaf_voxel"]
    scene, _ = oracle_scene(golden, fmt, g["svo_pos"], g["blocks"])
    for case in g["cases"]:
        res, _, _ = scene.intersect(case["pos"], orc.normalize(case["dir"]), g["max_dst"], g["cast_translucent"])
        check_result(res, case["expected"], what=case["name"])


@pytest.mark.parametrize("fmt", FMTS)
def test_picker_end_to_end(golden, fmt):
    """src/graphics/svo.rs:402-449: three picker rays against two blocks (chunk NOT compacted there)."""
    g = golden["picker_raycast"]
    scene, _ = oracle_scene(golden, fmt, [0, 0, 0], g["blocks"], compact=g["compact_chunk"])
    tasks = np.zeros(len(g["rays"]), dtype=orc.PICKER_TASK_DTYPE)
    for i, r in enumerate(g["rays"]):
        tasks[i]["max_dst"], tasks[i]["pos"], tasks[i]["dir"] = r["max_dst"], r["pos"], r["dir"]
    out = scene.picker(tasks)
    for got, exp in zip(out, g["expected"]):
        assert abs(float(got["dst"]) - exp["dst"]) < g["tol"]
        assert bool(got["inside_voxel"]) == exp["inside_voxel"]
        np.testing.assert_allclose(got["pos"], exp["pos"], rtol=0, atol=g["tol"])
        np.testing.assert_array_equal(got["normal"], np.asarray(exp["normal"], dtype=np.float32))


@pytest.mark.parametrize("fmt", FMTS)
def test_golden_results_bit_exact(golden, fmt):
    """Stronger than the reference's own 1e-5: with the oracle's explicit FMA placement (svo_oracle.c header) the
    printed golden floats of cast_inside_outside_all_axes and both traversal traces are reproduced bit for bit."""
    g = golden["formats"][fmt]["cast_inside_outside_all_axes"]
    scene, _ = oracle_scene(golden, fmt, g["svo_pos"], g["blocks"])
    for case in g["cases"]:
        res, _, _ = scene.intersect(case["pos"], orc.normalize(case["dir"]), g["max_dst"], g["cast_translucent"])
        e = case["expected"]
        assert np.float32(res.t) == np.float32(e["t"]), case["name"]
        assert [np.float32(x) for x in res.pos] == [np.float32(x) for x in e["pos"]], case["name"]
        assert [np.float32(x) for x in res.uv] == [np.float32(x) for x in e["uv"]], case["name"]
    for name in ("shader_svo_traversal", "check_at_higher_coordinates"):
        g = golden["formats"][fmt][name]
        scene, _ = oracle_scene(golden, fmt, g["svo_pos"], g["blocks"])
        ray = g["ray"]
        res, frames, n = scene.intersect(ray["pos"], orc.normalize(ray["dir"]), ray["max_dst"], ray["cast_translucent"], max_frames=100)
        e = g["result"]
        assert np.float32(res.t) == np.float32(e["t"]), name
        assert [np.float32(x) for x in res.pos] == [np.float32(x) for x in e["pos"]], name
        assert [np.float32(x) for x in res.uv] == [np.float32(x) for x in e["uv"]], name
        assert [np.float32(f["t_min"]) for f in frames] == [np.float32(f["t_min"]) for f in g["frames"]], name
