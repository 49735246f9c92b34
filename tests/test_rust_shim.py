"""N4: the Rust a maintainer drops into the reference -- integration/rust/voxel_hip_sys.rs (raw FFI), svo_hip.rs (the
public surface of graphics::Svo, /root/reference/src/graphics/svo.rs:56-256, over it) and worldsvo_updated_ranges.diff
(WorldSvo::updated_ranges, /root/reference/src/world/hds/common.rs:3-15) -- cannot be compiled here (no rustc), so it is checked
against include/voxel_hip.h by parsing both: every constant, every #[repr(C)] struct (field names, order, widths, offsets, size) and
every extern "C" function (name, arity, argument order, names, integer widths, pointer constness, return type) must agree; every call
svo_hip.rs makes must pass the declared number of arguments; the patch must apply to the reference (checked when /root/reference is
present: the build container)."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "voxel_hip.h"
SYS = ROOT / "integration" / "rust" / "voxel_hip_sys.rs"
WRAPPER = ROOT / "integration" / "rust" / "svo_hip.rs"
PATCH = ROOT / "integration" / "rust" / "worldsvo_updated_ranges.diff"
REFERENCE = Path("/root/reference")

# C scalar -> (Rust spellings that are the same type on x86-64 Linux, size, alignment)
C_SCALARS = {
    "float": ({"f32"}, 4, 4), "double": ({"f64"}, 8, 8), "uint8_t": ({"u8"}, 1, 1), "int32_t": ({"i32", "c_int"}, 4, 4), "uint32_t": ({"u32"}, 4, 4),
    "uint64_t": ({"u64"}, 8, 8), "int": ({"c_int", "i32"}, 4, 4), "size_t": ({"usize"}, 8, 8), "char": ({"c_char"}, 1, 1), "void": ({"c_void"}, 0, 1),
}
# reference types the Rust side passes where the header has its own mirror of the same std430 / repr(C) layout
RUST_ALIASES = {"PickerTask": "vx_picker_task", "PickerResult": "vx_picker_result", "MaterialInstance": "vx_material"}
# declared by the header for the test and measurement harness only: graphics::Svo has no use for them (everything else must be bound)
NOT_BOUND = {"vx_debug_trace", "vx_render_counters", "vx_profile_enable", "vx_profile_read", "vx_timeline_read", "vx_excursion_counters",
             "vx_traversal_image", "vx_traversal_image_with_origin", "vx_assemble_tiles_on", "vx_comm_profile_read", "vx_clock_probe", "vx_debug_knobs"}


def strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def strip_rust_comments(text):
    return re.sub(r"//[^\n]*", " ", text)


def split_top_level(s, sep=","):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out]


# ---- the C side ------------------------------------------------------------------------------------------------------

def c_type(decl_type, array):
    """Normal form of a C type: (constness of each pointee from the outermost pointer inwards, base, pointer depth, array length or
    None). The header only writes `const` in front of the base type, i.e. on the innermost pointee (`const void** pixels`)."""
    t = decl_type.replace("struct ", "").strip()
    const = bool(re.search(r"\bconst\b", t))
    assert not re.search(r"\*\s*const", t), "a const pointer level: teach c_type about it"
    t = re.sub(r"\bconst\b", "", t).strip()
    ptr = t.count("*")
    base = t.replace("*", "").strip()
    return ((False,) * (ptr - 1) + (const,) if ptr else (), base, ptr, array)


def parse_c_decl(decl):
    """`const float* tiles` / `float pos[3]` / `vx_context** out` -> (name, normal-form type)."""
    m = re.match(r"^(.*?)([A-Za-z_]\w*)\s*(\[\s*(\d+)\s*\])?$", decl.strip())
    assert m, decl
    return m.group(2), c_type(m.group(1), int(m.group(4)) if m.group(4) else None)


def parse_header():
    text = strip_c_comments(HEADER.read_text())
    consts = {k: int(v, 0) for k, v in re.findall(r"^#define\s+(VX_\w+)\s+(\d+)\s*$", text, flags=re.M)}
    for body in re.findall(r"typedef\s+enum\s+\w+\s*\{(.*?)\}", text, flags=re.S):
        for k, v in re.findall(r"(VX_\w+)\s*=\s*(-?\d+)", body):
            consts[k] = int(v)
    structs = {}
    for body, name in re.findall(r"typedef\s+struct\s+\w+\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for stmt in body.split(";"):
            stmt = " ".join(stmt.split())
            if not stmt:
                continue
            # `float fovy, aspect` / `float max_dst, _pad0[3]`: one type, several declarators
            first = re.match(r"^((?:const\s+)?(?:struct\s+)?\w+\s*\**)\s*(.*)$", stmt)
            base, rest = first.group(1), first.group(2)
            for d in split_top_level(rest):
                stars = re.match(r"^(\**)\s*(.*)$", d)
                fields.append(parse_c_decl(base + stars.group(1) + " " + stars.group(2)))
        structs[name] = fields
    funcs = {}
    protos = re.sub(r"typedef\s+(struct|enum)\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    protos = re.sub(r"^\s*#[^\n]*$", " ", protos, flags=re.M)  # preprocessor lines
    protos = re.sub(r'extern\s+"C"\s*\{', " ", protos)
    for ret, name, args in re.findall(r"([\w\s\*]+?)\b(vx_\w+)\s*\(([^()]*)\)\s*;", protos):
        ret = " ".join(ret.split())
        if ret.startswith("typedef"):
            continue
        params = [] if args.strip() in ("", "void") else [parse_c_decl(a) for a in split_top_level(args)]
        funcs[name] = (c_type(ret, None), params)
    return consts, structs, funcs


# ---- the Rust side ---------------------------------------------------------------------------------------------------

def rust_type(t):
    """Normal form of a Rust type, comparable with c_type()'s: (const?, base, pointer depth, array length or None)."""
    t = t.strip()
    const, ptr = (), 0
    while True:
        m = re.match(r"^\*(const|mut)\s+(.*)$", t)
        if not m:
            break
        const += (m.group(1) == "const",)  # (`*mut *const c_void`: the outer pointee -- a pointer -- is mutable, the inner one const)
        ptr += 1
        t = m.group(2).strip()
    arr = None
    m = re.match(r"^\[\s*([\w:]+)\s*;\s*(\d+)\s*\]$", t)
    if m:
        t, arr = m.group(1), int(m.group(2))
    return (const, t.split("::")[-1], ptr, arr)


def parse_rust():
    text = strip_rust_comments(SYS.read_text())
    consts = {k: (ty, int(v)) for k, ty, v in re.findall(r"pub const (VX_\w+)\s*:\s*(\w+)\s*=\s*(-?\d+)\s*;", text)}
    structs = {}
    for attrs, name, body in re.findall(r"((?:#\[[^\]]*\]\s*)+)pub struct (\w+)\s*\{(.*?)\}", text, flags=re.S):
        fields = [(n, rust_type(t)) for n, t in re.findall(r"(?:pub\s+)?(\w+)\s*:\s*([^,\n]+?)\s*,", body)]
        structs[name] = ("repr(C)" in attrs, fields)
    ext = re.search(r'extern "C"\s*\{(.*?)\n\}', text, flags=re.S).group(1)
    funcs = {}
    for name, args, ret in re.findall(r"pub fn (\w+)\s*\(([^()]*)\)\s*(?:->\s*([^;]+?))?\s*;", ext):
        params = [(a.split(":", 1)[0].strip(), rust_type(a.split(":", 1)[1])) for a in split_top_level(args)]
        funcs[name] = (rust_type(ret) if ret else ((), "c_void", 0, None), params)
    return consts, structs, funcs


def same_type(c, r, c_structs):
    """A C type and a Rust type name the same ABI type."""
    c_const, c_base, c_ptr, c_arr = c
    r_const, r_base, r_ptr, r_arr = r
    r_base = RUST_ALIASES.get(r_base, r_base)
    if c_arr is not None and r_arr is None and r_ptr == c_ptr + 1:
        # a C array PARAMETER (`uint64_t out[4]`) is a pointer; Rust spells it `*mut [u64; 4]`-- handled by the caller
        return False
    if (c_ptr, c_arr) != (r_ptr, r_arr):
        return False
    if c_ptr and c_const != r_const:
        return False
    if c_base in C_SCALARS:
        return r_base in C_SCALARS[c_base][0]
    return c_base == r_base and (c_base in c_structs or c_base == "vx_context")


def layout(fields, c_structs, rust=False):
    """[(name, offset, size)] and the struct's (size, alignment) under the x86-64 SysV rules both languages follow for repr(C)."""
    def scalar(base):
        if rust:
            for c_name, (spellings, size, align) in C_SCALARS.items():
                if base in spellings:
                    return size, align
            base = RUST_ALIASES.get(base, base)
        elif base in C_SCALARS:
            return C_SCALARS[base][1:]
        sub = layout(c_structs[base], c_structs)
        return sub[1]
    out, at, max_align = [], 0, 1
    for name, (_, base, ptr, arr) in fields:
        size, align = (8, 8) if ptr else scalar(base)
        total = size * (arr or 1)
        at = (at + align - 1) // align * align
        out.append((name, at, total))
        at += total
        max_align = max(max_align, align)
    return out, ((at + max_align - 1) // max_align * max_align, max_align)


# ---- the checks ------------------------------------------------------------------------------------------------------

def test_constants_agree():
    c_consts, _, _ = parse_header()
    r_consts, _, _ = parse_rust()
    assert r_consts, "no constants parsed from voxel_hip_sys.rs"
    for name, (ty, value) in r_consts.items():
        assert name in c_consts, f"{name} is not in voxel_hip.h"
        assert c_consts[name] == value, f"{name}: header {c_consts[name]}, Rust {value}"
        assert ty in ("c_int", "i32", "usize", "u32"), (name, ty)
    # everything graphics::Svo needs to name: status codes, memory kinds, formats, node formats
    for name in c_consts:
        if name.startswith(("VX_ERR_", "VX_MEM_", "VX_FORMAT_", "VX_SVO_")) or name in ("VX_OK", "VX_COMM_ID_BYTES"):
            assert name in r_consts, f"{name} of voxel_hip.h has no Rust constant"


def test_structs_agree():
    _, c_structs, _ = parse_header()
    _, r_structs, _ = parse_rust()
    checked = 0
    for name, (repr_c, r_fields) in r_structs.items():
        if name == "vx_context":  # opaque on both sides
            assert repr_c
            continue
        assert name in c_structs, f"struct {name} is not in voxel_hip.h"
        assert repr_c, f"struct {name} is not #[repr(C)]"
        c_fields = c_structs[name]
        assert [n for n, _ in r_fields] == [n for n, _ in c_fields], f"{name}: field names / order differ: {r_fields} vs {c_fields}"
        for (n, ct), (_, rt) in zip(c_fields, r_fields):
            assert same_type(ct, rt, c_structs), f"{name}.{n}: header {ct}, Rust {rt}"
        c_lay, r_lay = layout(c_fields, c_structs), layout(r_fields, c_structs, rust=True)
        assert c_lay == r_lay, f"{name}: layouts differ: {c_lay} vs {r_lay}"
        checked += 1
    assert checked >= 5
    for name in ("vx_range", "vx_uniforms", "vx_hit", "vx_stats", "vx_target"):
        assert name in r_structs, f"{name} has no Rust mirror"
    # the sizes the Rust file asserts at compile time are the header's
    sizes = {n: layout(f, c_structs)[1][0] for n, f in c_structs.items()}
    text = SYS.read_text()
    for name, expr in re.findall(r"size_of::<(\w+)>\(\)\s*==\s*([\d\s+*()]+?)\s*(?:\)\s*;|&&)", text):
        c_name = RUST_ALIASES.get(name, name)
        assert sizes[c_name] == eval(expr, {"__builtins__": {}}), f"size_of::<{name}>() asserted as {expr}, the header's {c_name} has {sizes[c_name]} bytes"
    assert sizes["vx_material"] == 32 and sizes["vx_picker_task"] == 48 and sizes["vx_picker_result"] == 48 and sizes["vx_hit"] == 48


def test_functions_agree():
    _, c_structs, c_funcs = parse_header()
    _, _, r_funcs = parse_rust()
    assert len(r_funcs) >= 30
    for name, (r_ret, r_params) in r_funcs.items():
        assert name in c_funcs, f"{name} is declared in Rust but not in voxel_hip.h"
        c_ret, c_params = c_funcs[name]
        assert len(c_params) == len(r_params), f"{name}: {len(c_params)} parameters in the header, {len(r_params)} in Rust"
        assert [n for n, _ in c_params] == [n for n, _ in r_params], f"{name}: parameter names / order differ: {c_params} vs {r_params}"
        for (n, ct), (_, rt) in zip(c_params, r_params):
            if ct[3] is not None:  # `uint64_t out[4]` decays to a pointer; Rust: `*mut [u64; 4]`
                assert rt[2] == ct[2] + 1 and rt[3] == ct[3] and same_type(((), ct[1], 0, None), ((), rt[1], 0, None), c_structs), f"{name}({n})"
                continue
            assert same_type(ct, rt, c_structs), f"{name}({n}): header {ct}, Rust {rt}"
        if c_ret == ((), "void", 0, None):
            assert r_ret == ((), "c_void", 0, None), f"{name}: returns nothing in the header"
        else:
            assert same_type(c_ret, r_ret, c_structs), f"{name}: return type {c_ret} vs {r_ret}"
    missing = set(c_funcs) - set(r_funcs) - NOT_BOUND
    assert not missing, f"declared by voxel_hip.h, neither bound in voxel_hip_sys.rs nor listed as harness-only: {sorted(missing)}"
    assert not (NOT_BOUND - set(c_funcs)), "NOT_BOUND names something the header no longer declares"
    # the header's order, as the Rust file promises
    order = [n for n in re.findall(r"\b(vx_\w+)\s*\(", strip_c_comments(HEADER.read_text())) if n in r_funcs]
    seen = []
    for n in order:
        if n not in seen:
            seen.append(n)
    assert set(seen) == set(r_funcs)


def test_wrapper_calls_pass_the_declared_arguments():
    """svo_hip.rs: every call of a bound function has as many arguments as the header declares, and every graphics::Svo method of
    the reference (svo.rs:109-255: new, reload_resources, update, get_stats, render, raycast, Drop) is there."""
    _, _, c_funcs = parse_header()
    text = strip_rust_comments(WRAPPER.read_text())
    calls = 0
    for m in re.finditer(r"\b(vx_[a-z0-9_]+)\s*\(", text):
        name = m.group(1)
        if name not in c_funcs:
            continue
        depth, i = 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        args = split_top_level(text[m.end():i - 1])
        assert len(args) == len(c_funcs[name][1]), f"svo_hip.rs calls {name} with {len(args)} arguments, the header declares {len(c_funcs[name][1])}"
        calls += 1
    assert calls >= 12
    for method in ("pub fn new", "pub fn update", "pub fn get_stats", "pub fn render", "pub fn raycast", "impl Drop for"):
        assert method in text, f"svo_hip.rs lacks `{method}`"
    for used in ("vx_create", "vx_destroy", "vx_set_materials", "vx_set_textures", "vx_staging_ptr", "vx_commit", "vx_render", "vx_raycast"):
        assert re.search(r"\b" + used + r"\s*\(", text), f"svo_hip.rs never calls {used}"


@pytest.mark.skipif(not REFERENCE.is_dir() or shutil.which("patch") is None, reason="needs /root/reference (the build container) and patch(1)")
def test_updated_ranges_patch_applies_to_the_reference(tmp_path):
    """The patch names four files of src/world/hds; they are copied to a scratch tree (the reference is read-only) and the patch is
    applied there for real, then the result is looked at: WorldSvo gained updated_ranges and both serializers implement it."""
    files = sorted(set(re.findall(r"^\+\+\+ b/(\S+)", PATCH.read_text(), flags=re.M)))
    assert files and all(f.startswith("src/world/hds/") for f in files)
    for f in files:
        (tmp_path / f).parent.mkdir(parents=True, exist_ok=True)
        shutil.copy(REFERENCE / f, tmp_path / f)
    r = subprocess.run(["patch", "-p1", "--forward", "--fuzz=0", "-i", str(PATCH)], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert "fn updated_ranges(&self) -> Vec<Range>;" in (tmp_path / "src/world/hds/common.rs").read_text()
    for f in ("src/world/hds/esvo.rs", "src/world/hds/csvo.rs"):
        assert "fn updated_ranges(&self) -> Vec<Range>" in (tmp_path / f).read_text(), f
