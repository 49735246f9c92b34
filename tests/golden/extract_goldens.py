#!/usr/bin/env python3
"""Extracts the golden vectors of the reference's own GPU tests into tests/golden/svo_shader_tests.json.

Source of the data: /root/reference/src/graphics/svo_shader_tests.rs (traversal frames, ray results, test
worlds, test textures/materials) and /root/reference/src/graphics/svo.rs:402-449 (picker end-to-end).
Only VALUES are taken over (expected frames/results and the inputs that produce them); run once in the
build container -- /root/reference does not exist on the GPU box, the JSON is what travels.

    python tests/golden/extract_goldens.py
"""
import json
import re
from pathlib import Path

REF = Path("/root/reference/src/graphics/svo_shader_tests.rs")
OUT = Path(__file__).with_name("svo_shader_tests.json")

FRAME_RE = re.compile(
    r"StackFrame \{ t_min: (?P<t>assert_float_eq!\([^)]*\)|[^,]+), ptr: (?P<ptr>\d+), idx: (?P<idx>\d+), parent_octant_idx: (?P<p>\d+), scale: (?P<scale>\d+), "
    r"is_child: AlignedBool\((?P<c>\d)\), is_leaf: AlignedBool\((?P<l>\d)\), crossed_boundary: AlignedBool\((?P<b>\d)\), next_ptr: (?P<n>\d+) \}"
)


def parse_t(txt):
    m = re.match(r"assert_float_eq!\(.+, ([0-9.e+-]+)\)$", txt.strip())
    return float(m.group(1)) if m else float(txt)


def frames_of(block):
    out = []
    for m in FRAME_RE.finditer(block):
        out.append(dict(t_min=parse_t(m["t"]), ptr=int(m["ptr"]), idx=int(m["idx"]), parent_octant_idx=int(m["p"]), scale=int(m["scale"]),
                        is_child=int(m["c"]), is_leaf=int(m["l"]), crossed_boundary=int(m["b"]), next_ptr=int(m["n"])))
    return out


def nums(txt):
    return [float(x) for x in re.findall(r"-?\d+\.?\d*(?:e-?\d+)?", txt)]


def result_of(block):
    """Parses one `OctreeResult { ... }` literal (values possibly wrapped in assert_*_eq! macros)."""
    def field(name, pattern):
        m = re.search(name + r":\s*" + pattern, block)
        assert m, (name, block)
        return m
    t_txt = field("t", r"([^\n]+),\n").group(1)
    m = re.match(r"assert_float_eq!\([^,]+, ([0-9.e+-]+)(?:, ([0-9.e+-]+))?\)", t_txt.strip())
    t, t_tol = (float(m.group(1)), float(m.group(2)) if m.group(2) else 1e-5) if m else (float(t_txt), 0.0)
    value = int(field("value", r"(\d+)").group(1))
    face = int(field("face_id", r"(\d+)").group(1))

    def vec(name, n):
        m = re.search(name + r":\s*([^\n]+)\n", block)
        txt = m.group(1)
        tol = 0.0
        mm = re.match(r"assert_vec\d_eq!\([^,]+,\s*\w+::new\(([^)]*)\)(?:,\s*([0-9.e+-]+))?\)", txt.strip())
        if mm:
            vals = nums(mm.group(1))
            tol = float(mm.group(2)) if mm.group(2) else 1e-5
        else:
            vals = nums(re.search(r"new\(([^)]*)\)", txt).group(1))
        assert len(vals) == n, (name, txt)
        return vals, tol
    pos, pos_tol = vec("pos", 3)
    uv, uv_tol = vec("uv", 2)
    color, color_tol = vec("color", 4)
    inside = "true" in field("inside_voxel", r"AlignedBool::from\((\w+)\)").group(1)
    return dict(t=t, t_tol=t_tol, value=value, face_id=face, pos=pos, pos_tol=pos_tol, uv=uv, uv_tol=uv_tol, color=color,
                color_tol=color_tol, inside_voxel=inside)


def section(src, start_marker, end_marker):
    a = src.index(start_marker)
    b = src.index(end_marker, a)
    return src[a:b]


def main():
    src = REF.read_text()
    esvo = section(src, "mod esvo_tests {", "mod csvo_tests {")
    csvo = section(src, "mod csvo_tests {", "mod esvo_benchmarks {")

    # test fixtures shared by every case: svo_shader_tests.rs:117-202 (4x4 RGBA8 textures, rows listed top to bottom;
    # TextureArrayBuilder::add_rgba8 flips them on load) and the 5 materials built from them
    tex_block = section(src, "fn create_test_materials()", "let material_buffer = Buffer::new")
    textures = {}
    for m in re.finditer(r'\.add_rgba8\("(\w+)", 4, 4, vec!\[(.*?)\]\)\?', tex_block, re.S):
        body = re.sub(r"//[^\n]*", "", m.group(2))  # drop line comments ("255 * 0.2 = 51")
        vals = [int(x) for x in re.findall(r"\d+", body.replace("/**/", ""))]
        assert len(vals) == 64, (m.group(1), len(vals))
        textures[m.group(1)] = vals
    tex_order = ["full", "coords", "transparent_1", "transparent_2"]
    assert list(textures) == tex_order
    materials = [dict(specular_pow=0.0, specular_strength=0.0, tex_top=-1, tex_side=-1, tex_bottom=-1, tex_top_normal=-1, tex_side_normal=-1,
                      tex_bottom_normal=-1)]
    for i, _ in enumerate(tex_order):
        materials.append(dict(specular_pow=0.0, specular_strength=0.0, tex_top=i, tex_side=i, tex_bottom=i, tex_top_normal=-1,
                              tex_side_normal=-1, tex_bottom_normal=-1))

    out = dict(source="src/graphics/svo_shader_tests.rs", float_eps=1e-5,
               textures=dict(width=4, height=4, mip_levels=1, order=tex_order, rgba8_rows_top_to_bottom=textures), materials=materials,
               formats={})

    floor5 = [dict(box=[[0, 32], [0, 5], [0, 32]], id=1)]  # :709-715, `for x in 0..32, z in 0..32, y in 0..5`
    for name, blk in (("esvo", esvo), ("csvo", csvo)):
        fmt = {}
        # shader_svo_traversal (:293-334 / :763-804)
        s = section(blk, "fn shader_svo_traversal()", "fn cast_inside_outside_all_axes()")
        fmt["shader_svo_traversal"] = dict(svo_pos=[0, 0, 0], blocks=[[31, 0, 0, 1]], ray=dict(pos=[0.0, 0.5, 0.5], dir=[1.0, 0.0, 0.0], max_dst=32.0,
                                           cast_translucent=False), frames=frames_of(s), result=result_of(section(s, "assert_eq!(buffer_out.result, OctreeResult {", "});")))
        assert len(fmt["shader_svo_traversal"]["frames"]) == 11

        # check_at_higher_coordinates (:707-753 / :1177-1223)
        s = blk[blk.index("fn check_at_higher_coordinates()"):]
        fmt["check_at_higher_coordinates"] = dict(svo_pos=[15, 15, 15], blocks=floor5, ray=dict(pos=[484.9203, 485.95938, 493.8467], dir=[0.0, -1.0, 0.0],
                                                  max_dst=10.0, cast_translucent=False), frames=frames_of(s),
                                                  result=result_of(section(s, "assert_eq!(buffer_out.result, OctreeResult {", "});")))
        assert len(fmt["check_at_higher_coordinates"]["frames"]) == 10

        # cast_inside_outside_all_axes (:340-489 / :810-959): each case also runs from one unit further back
        s = section(blk, "fn cast_inside_outside_all_axes()", "fn uv_coords_on_all_sides()")
        cases = []
        for m in re.finditer(r'name: "([^"]+)",\s*pos: Point3::new\(([^)]*)\),\s*dir: Vector3::new\(([^)]*)\),\s*expected: OctreeResult \{(.*?)\n\s*\},\n', s, re.S):
            cases.append(dict(name=m.group(1), pos=nums(m.group(2)), dir=nums(m.group(3)), expected=result_of("OctreeResult {" + m.group(4) + "\n}")))
        assert len(cases) == 8, len(cases)
        fmt["cast_inside_outside_all_axes"] = dict(svo_pos=[0, 0, 0], blocks=[[30, 0, 0, 1], [0, 30, 0, 1], [0, 0, 30, 1], [30, 30, 30, 1]], max_dst=100.0,
                                                   cast_translucent=False, cases=cases)

        # uv_coords_on_all_sides (:495-604 / :965-1074)
        s = section(blk, "fn uv_coords_on_all_sides()", "fn casting_against_translucent_leafs()")
        cases = []
        for m in re.finditer(r"pos: Point3::new\(([^)]*)\),\s*dir: Vector3::new\(([^)]*)\),\s*expected_uv: Point2::new\(([^)]*)\),\s*expected_color: Vector4::new\(([^)]*)\)", s):
            cases.append(dict(pos=nums(m.group(1)), dir=nums(m.group(2)), expected_uv=nums(m.group(3)), expected_color=nums(m.group(4))))
        assert len(cases) == 14, len(cases)
        fmt["uv_coords_on_all_sides"] = dict(svo_pos=[0, 0, 0], blocks=[[0, 0, 0, 2]], max_dst=32.0, cast_translucent=False, cases=cases)

        # casting_against_translucent_leafs (:609-658 / :1079-1128)
        s = section(blk, "fn casting_against_translucent_leafs()", "fn detect_inside_leaf_voxel()")
        rs = [result_of(m.group(0)) for m in re.finditer(r"OctreeResult \{.*?\n\s*\}, \"", s, re.S)]
        assert len(rs) == 3
        d = [0.75 - 0.25, 0.5 - 0.5, 1.0 - (-0.1)]
        fmt["casting_against_translucent_leafs"] = dict(
            svo_pos=[0, 0, 0], blocks=[[0, 0, 0, 3], [0, 0, 1, 3], [5, 0, 0, 3], [5, 0, 1, 4]], max_dst=32.0,
            cases=[dict(name="do not cast translucent", pos=[0.25, 0.5, -0.1], dir=d, cast_translucent=False, expected=rs[0]),
                   dict(name="cast translucent with adjacent identical", pos=[0.25, 0.5, -0.1], dir=d, cast_translucent=True, expected=rs[1]),
                   dict(name="cast translucent with adjacent different", pos=[5.25, 0.5, -0.1], dir=d, cast_translucent=True, expected=rs[2])])

        # detect_inside_leaf_voxel (:663-701 / :1133-1171)
        s = section(blk, "fn detect_inside_leaf_voxel()", "fn check_at_higher_coordinates()")
        rs = [result_of(m.group(0)) for m in re.finditer(r"OctreeResult \{.*?\n\s*\}, \"", s, re.S)]
        assert len(rs) == 2
        fmt["detect_inside_leaf_voxel"] = dict(svo_pos=[0, 0, 0], blocks=[[0, 0, 0, 1]], max_dst=32.0, cast_translucent=False,
                                               cases=[dict(name="inside block", pos=[0.5, 0.5, 0.5], dir=[1.0, 0.0, 0.0], expected=rs[0]),
                                                      dict(name="outside block", pos=[-0.5, 0.5, 0.5], dir=[1.0, 0.0, 0.0], expected=rs[1])])
        out["formats"][name] = fmt

    # the result tables of the two formats are identical in the reference; keep that as a checked property
    for k in ("cast_inside_outside_all_axes", "uv_coords_on_all_sides", "casting_against_translucent_leafs", "detect_inside_leaf_voxel"):
        assert out["formats"]["esvo"][k] == out["formats"]["csvo"][k], k

    # picker end-to-end, src/graphics/svo.rs:402-449 (both formats, tolerance 1e-4)
    out["picker_raycast"] = dict(
        source="src/graphics/svo.rs:402-449", blocks=[[0, 0, 0, 1], [1, 0, 0, 1]], tol=1e-4, compact_chunk=False,
        rays=[dict(pos=[0.5, 1.5, 0.5], dir=[0.0, -1.0, 0.0], max_dst=1.0), dict(pos=[0.5, 0.5, 0.5], dir=[1.0, 0.0, 0.0], max_dst=1.0),
              dict(pos=[0.5, 0.5, -2.0], dir=[0.0, 0.0, 1.0], max_dst=1.0)],
        expected=[dict(dst=0.5, inside_voxel=False, pos=[0.5, 1.0, 0.5], normal=[0.0, 1.0, 0.0]),
                  dict(dst=0.5, inside_voxel=True, pos=[1.0, 0.5, 0.5], normal=[-1.0, 0.0, 0.0]),
                  dict(dst=-1.0, inside_voxel=False, pos=[0.0, 0.0, 0.0], normal=[0.0, 0.0, 0.0])])

    # picker batch (de)serialisation tables, src/graphics/svo_picker.rs:311-418 and :422-536
    psrc = Path("/root/reference/src/graphics/svo_picker.rs").read_text()
    ser = section(psrc, "fn picker_batch_serialization()", "fn picker_batch_deserialization()")
    tasks = [dict(max_dst=float(m.group(1)), pos=nums(m.group(2)), dir=nums(m.group(3))) for m in re.finditer(
        r"PickerTask \{ max_dst: ([0-9.]+), pos: AlignedPoint3\(Point3::new\(([^)]*)\)\), dir: AlignedVec3\(Vector3::new\(([^)]*)\)\) \}", ser)]
    tasks = tasks[1:]  # the first literal is the buffer's default fill value, not an expected task
    assert len(tasks) == 80, len(tasks)
    des = psrc[psrc.index("fn picker_batch_deserialization()"):]
    results = [dict(dst=float(m.group(1)), inside_voxel=m.group(2) == "true", pos=nums(m.group(3)), normal=nums(m.group(4))) for m in re.finditer(
        r"PickerResult \{ dst: (-?[0-9.]+), inside_voxel: (\w+), pos: AlignedPoint3\(Point3::new\(([^)]*)\)\), normal: AlignedVec3\(Vector3::new\(([^)]*)\)\) \}", des)]
    assert len(results) == 80, len(results)
    batch = dict(aabbs=[dict(pos=[0.5, 0.0, 0.5], offset=[-0.5, 0.0, -0.5], extents=[1.0, 1.0, 1.0]),
                        dict(pos=[0.0, 0.0, 0.0], offset=[0.0, 0.0, 0.0], extents=[1.5, 1.5, 1.5])])
    out["picker_batch"] = dict(
        source="src/graphics/svo_picker.rs:311-536",
        serialization=dict(rays=[dict(pos=[1.0, 0.0, 1.0], dir=[0.0, 1.0, 0.0], max_dst=20.0), dict(pos=[2.0, 0.0, 2.0], dir=[1.0, 0.0, 0.0], max_dst=40.0)],
                           aabbs=batch["aabbs"], expected_tasks=tasks),
        deserialization=dict(rays=[dict(pos=[0.0, 0.0, 0.0], dir=[-1.0, 0.0, 0.0], max_dst=20.0), dict(pos=[0.0, 0.0, 0.0], dir=[1.0, 0.0, 0.0], max_dst=20.0)],
                             aabbs=batch["aabbs"], results=results,
                             expected_rays=[dict(dst=-1.0, inside_voxel=False, pos=[0.0, 0.0, 0.0], normal=[0.0, 0.0, 0.0]),
                                            dict(dst=10.0, inside_voxel=True, pos=[-1.0, 0.0, 0.0], normal=[10.0, 0.0, 0.0])],
                             expected_aabbs=[dict(neg=[8.0, 7.0, 8.0], pos=[2.0, 4.0, 1.0]), dict(neg=[9.0, 8.0, 7.0], pos=[1.0, 4.0, 3.0])]))

    OUT.write_text(json.dumps(out, indent=1))
    print("wrote", OUT, OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
