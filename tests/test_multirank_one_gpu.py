"""The N > 1 branch of the library's exchange and of bench.py, EXECUTED on the one GPU there is: two and three ranks share device 0, with
tests/stub_rccl (HIP IPC + flag kernels between processes) standing in for RCCL, which refuses two ranks on one device. Not a test of RCCL: a test of
voxel-rs_amd/csrc/hip/comm.cpp's grouped receive loop, its tickets, the assembly's ordering, the root-renders-in-place offsets, vx_comm_init on
several ranks, the wave slots a context with a communicator leaves free (VX_COMM_HEADROOM 4 and 0), and of bench.py's Sharded runner with its watchdog.
The design they implement: SURVEY.md 8(e); the reference has one GL context (src/graphics/svo.rs:196-229)."""
import json
import os
import signal
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
STUB = ROOT / "tests" / "_build" / "stub_rccl" / "librccl_stub.so"


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(world, argv, extra_env=None, timeout=420):
    """`world` processes, all on GPU 0 (LOCAL_RANK 0), rendezvous on 127.0.0.1; returns [(returncode, stdout, stderr)] by rank. A run that does not end in
    time is killed by process group (never by pattern)."""
    assert STUB.exists(), f"{STUB} is missing: make -C tests/stub_rccl (also done by __graft_entry__.build())"
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True))
    out = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                try:
                    os.killpg(q.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
            o, e = p.communicate()
            e += "\n[killed: timeout]"
        out.append((p.returncode, o, e))
    return out


@pytest.mark.parametrize("world,fmt,gather_format,headroom,group,calls", [
    (2, "csvo", "rgba8", "4", 1, "one"), (2, "esvo", "rgba32f", "0", 1, "separate"), (3, "csvo", "rgba32f", "4", 1, "one"), (3, "esvo", "rgba8", "0", 2, "separate"),
], ids=["2ranks-csvo-rgba8-headroom4", "2ranks-esvo-rgba32f-headroom0-separate-calls", "3ranks-csvo-rgba32f-headroom4", "3ranks-esvo-rgba8-headroom0-group2"])
def test_ranks_sharing_one_gpu_gather_the_whole_frame(tmp_path, world, fmt, gather_format, headroom, group, calls):
    """Thirty frames of a moving camera through the library's exchange with `world` ranks -- a frame per call (vx_render_gather) or as wait / render /
    gather / assemble --: every frame rank 0 assembles is, byte for byte, the frame rendered whole."""
    out = tmp_path / "result.json"
    res = launch(world, [str(ROOT / "tests" / "multirank_worker.py"), str(out), fmt, gather_format, "30", str(group), calls], {"VX_COMM_HEADROOM": headroom})
    for r, (rc, _, err) in enumerate(res):
        assert rc == 0, f"rank {r} failed:\n{err[-3000:]}"
    d = json.loads(out.read_text())
    assert d["identical"] and d["world"] == world and d["checked"] == -(-30 // group), d
    assert d["frames_in_flight"] >= 2  # (max(3, N), rounded down to whole groups)


def bench_ranks(world, extra):
    argv = [str(ROOT / "bench.py"), "--gpus", str(world), "--steps", "10", "--warmup", "2", "--repeats", "3", "--sustained-seconds", "0.2", "--no-cpu-baseline", "--depth", "9",
            "--width", "640", "--height", "360", "--dist-backend", "gloo", "--comm-library", str(STUB), "--textures", "procedural"] + extra
    res = launch(world, argv)
    for r, (rc, _, err) in enumerate(res):
        assert rc == 0, f"rank {r} failed:\n{err[-3000:]}"
    lines = [ln for ln in res[0][1].splitlines() if ln.startswith("{")]
    assert lines, res[0][1][-2000:] + res[0][2][-2000:]
    return json.loads(lines[-1])


def test_bench_with_two_ranks_on_one_gpu():
    """bench.py itself as the driver launches it for N = 2 (one process per rank, its whole Sharded runner, the barriers, the max over ranks), the library's
    exchange carrying every frame: the JSON line says so, rank 0's last frame is the whole render, both ranks' rays add up to the frame's."""
    d = bench_ranks(2, ["--gather", "library"])
    c = d["config"]
    assert d["n_gpus"] == 2 and c["rccl_ranks"] == 2 and c["gather"].startswith("vx_gather_tiles") and "gather_note" not in c
    assert c["sharded_frame_identical_to_whole_render"] is True
    assert len(c["per_rank"]) == 2 and all(r["rays_per_block"] > 0 for r in c["per_rank"])
    assert d["value"] > 0 and d["sustained"]["frames"] >= 100
    # the self-calibration of a first real N > 1 run (functional here: the ranks share one GPU): comm headroom {0, 2, 4} x frames per gather {1, 2}, the
    # same choice on every rank, and the timed region ran with it
    cal = c["calibration"]
    assert len(cal["trials"]) == 6 and {(t["comm_headroom"], t["frames_per_gather"]) for t in cal["trials"]} == {(h, g) for h in (0, 2, 4) for g in (1, 2)}
    assert cal["chosen"] in cal["trials"] and cal["chosen"]["ms_per_step"] == min(t["ms_per_step"] for t in cal["trials"])
    assert d["roofline"]["frames_per_gather"] == cal["chosen"]["frames_per_gather"] and c["comm_headroom"] == cal["chosen"]["comm_headroom"]


def test_bench_with_eight_ranks_on_one_gpu():
    """... and for N = 8, the size of the node the scaling run uses: eight processes, eight frames in flight in groups of two (pinned here; the calibration
    then tries the three headrooms only), seven receives per exchange on rank 0."""
    d = bench_ranks(8, ["--gather", "library", "--gather-group", "2"])
    c = d["config"]
    assert d["n_gpus"] == 8 and c["rccl_ranks"] == 8 and c["gather"].startswith("vx_gather_tiles") and "gather_note" not in c
    assert c["sharded_frame_identical_to_whole_render"] is True
    assert len(c["per_rank"]) == 8 and all(r["rays_per_block"] > 0 for r in c["per_rank"])
    assert d["roofline"]["frames_in_flight"] == 8 and d["roofline"]["frames_per_gather"] == 2
    assert [t["comm_headroom"] for t in c["calibration"]["trials"]] == [0, 2, 4] and all(t["frames_per_gather"] == 2 for t in c["calibration"]["trials"])


def test_bench_watchdog_when_a_peer_never_joins():
    """A peer that never joins the exchange: rank 0's receive waits on the device, vx_gather_query stays at 'not yet', the watchdog's deadline passes, every
    rank switches IN THE SAME PROCESS to torch.distributed's gather, whose frames are checked the same way -- the run ends, with a line that says what ran."""
    d = bench_ranks(2, ["--gather", "auto", "--gather-timeout", "3", "--simulate-absent-peer", "--no-calibration"])
    c = d["config"]
    assert c["gather"].startswith("torch.distributed.gather") and "failed its first frames" in c["gather_note"]
    assert c["sharded_frame_identical_to_whole_render"] is True
