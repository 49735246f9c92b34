"""bench.py's pieces that need no GPU: the byte models of SURVEY.md 8(d) on hand-made counters, the rule that stored PMC counters are quoted only for the
library sources they were measured on, what the box grants, the defaults the driver relies on."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

COUNTERS = {"rays": 10, "iterations": 300, "pushes": 120, "leaf_tests": 7, "leaf_tests_trilinear": 2, "boundaries": 3, "csvo_header_bytes": 450, "csvo_pointer_bytes": 160,
            "pixels": 8, "lit_pixels": 5}


def test_byte_models_are_the_surveys():
    # ESVO: 4 B descriptor per iteration, 4 B pointer per PUSH, per leaf test pointer + value (8) + material row (32) + 4 B (nearest) or 32 B (trilinear) of texels
    esvo = 4 * 300 + 4 * 120 + 7 * (8 + 32) + 5 * 4 + 2 * 32
    # CSVO: header and pointer bytes as counted, per leaf test u16 + 8 mask bytes + u32 + material row, texels, 5 B per chunk boundary
    csvo = 450 + 160 + 7 * (2 + 8 + 4 + 32) + 5 * 4 + 2 * 32 + 5 * 3
    shade = 16 * 8 + 5 * (32 + 4)  # per pixel the RGBA32F store, per lit pixel a material row and one normal-map texel
    assert bench.algorithmic_bytes("esvo", COUNTERS) == esvo + shade
    assert bench.algorithmic_bytes("csvo", COUNTERS) == csvo + shade
    assert bench.image_model_bytes(COUNTERS) == 8 * 120 + 7 * (4 + 32) + 5 * 4 + 2 * 32 + shade


def test_stored_counters_are_quoted_for_their_sources_only(tmp_path, monkeypatch):
    import _pkg

    (tmp_path / "profiles" / "round5").mkdir(parents=True)
    (tmp_path / "profiles" / "round5" / "traffic.json").write_text(json.dumps({"commit": "abc1234", "csrc_sha16": "0123456789abcdef", "csvo": {"bytes_per_launch": 42},
                                                                                "C4": {"csvo": {"bytes_per_launch": 43}}}))
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    monkeypatch.setattr(_pkg, "csrc_hash", lambda: "0123456789abcdef")
    row, source, note = bench.stored_counters("csvo")
    assert row == {"bytes_per_launch": 42} and "round5/traffic.json @ abc1234" in source and note is None
    assert bench.stored_counters("csvo", "C4")[0] == {"bytes_per_launch": 43} and bench.stored_counters("esvo", "C4")[0] is None
    monkeypatch.setattr(_pkg, "csrc_hash", lambda: "fedcba9876543210")
    row, source, note = bench.stored_counters("csvo")
    assert row is None and "other library sources" in note and "fedcba9876543210" in note
    # a counter file that names no hash (round 4's) is never quoted
    (tmp_path / "profiles" / "round5" / "traffic.json").write_text(json.dumps({"commit": "abc1234", "csvo": {"bytes_per_launch": 42}}))
    assert bench.stored_counters("csvo")[0] is None


def test_the_committed_counters_belong_to_the_committed_sources():
    """profiles/round6/traffic.json was measured on the library sources in the tree: bench.py will quote it (a kernel change without a new profile
    pass fails here instead of silently dropping `roofline.traffic` from the record). It holds the headline's counters (C3) and the depth-14 frame's (C4)."""
    from _pkg import csrc_hash

    t = json.loads((ROOT / "profiles" / "round6" / "traffic.json").read_text())
    assert t["csrc_sha16"] == csrc_hash(), "re-run profiles/round6/run_profiles.sh on the GPU box and commit its traffic.json"
    for fmt in ("csvo", "esvo"):
        row, _, note = bench.stored_counters(fmt)
        assert note is None and row["bytes_per_launch"] > 0 and row["SQ_INSTS_VALU"] > 0 and row["read_bytes"] > 0
        row4, _, note4 = bench.stored_counters(fmt, "C4")
        assert note4 is None and row4["bytes_per_launch"] > row["bytes_per_launch"] and row4["SQ_INSTS_VALU"] > row["SQ_INSTS_VALU"] and row4["TCC_MISS"] > 0


def test_granted_cpus_and_defaults():
    g = bench.granted_cpus()
    assert g["logical_cpus"] >= 1 and (g["affinity"] is None or 1 <= g["affinity"] <= g["logical_cpus"])
    assert g["cgroup_cpu_max"] is None or g["cgroup_cpu_max"] > 0
    a = bench.parse_args([])
    assert (a.gpus, a.format, a.depth, a.width, a.height) == (1, "csvo", 12, 1920, 1080) and a.sustained_seconds >= 3.0 and a.dist_backend == "nccl"
    u = bench.moving_uniforms(_scenes(), 12, 1000, 1920, 1080, 7)
    assert u.render_shadows == 1 and u.shadow_distance > 1e30


def _scenes():
    from _pkg import load_package

    load_package()
    from voxel_rs_amd import scenes

    return scenes
