"""Worker of tests/test_multirank_one_gpu.py: one RANK of an N-rank run whose ranks all share GPU 0.

Runs the multi-rank code path bench.py runs on a node -- bench.Sharded: tile-list renders with max(3, N) frames in flight, the library's own exchange
(vx_comm_init, vx_gather_tiles: the grouped receive loop on rank 0, a send on the others), the tickets that keep a render out of a list the exchange
still reads, rank 0's assembly on the communicator's stream -- with tests/stub_rccl standing in for RCCL (which refuses two ranks on one device) and
gloo carrying the unique id and the barriers. Rank 0 compares EVERY assembled frame of a moving camera with the same frame rendered whole.

    RANK=r LOCAL_RANK=0 WORLD_SIZE=N MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/multirank_worker.py <out.json> <format> <gather-format> <frames> <group> [separate]
"""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    out_path, fmt_name, gather_format, n_frames, group = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    import torch
    import torch.distributed as dist

    import bench
    from _pkg import load_package

    vra = load_package()
    from voxel_rs_amd import hip, scenes

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    stub = ROOT / "tests" / "_build" / "stub_rccl" / "librccl_stub.so"
    hip.comm_library(stub)
    args = bench.parse_args(["--gpus", str(world), "--steps", str(n_frames), "--depth", "9", "--width", "640", "--height", "360", "--format", fmt_name,
                             "--gather", "library", "--gather-format", gather_format, "--gather-group", str(group), "--gather-timeout", "60",
                             "--dist-backend", "gloo", "--textures", "procedural"] + (["--separate-calls"] if len(sys.argv) > 6 and sys.argv[6] == "separate" else []))
    args.ctl_device = "cpu"
    wl = bench.Workload(args, vra, hip, scenes, rank, world, 0)
    run = bench.Sharded(args, wl, torch, dist, hip, rank, world)  # (its first frames are already checked against the whole render, on every rank's verdict)
    assert run.gather_used == "library", run.gather_note
    nranks, myrank = hip.C.c_int(0), hip.C.c_int(0)
    hip.lib().vx_comm_info(wl.svo._h, hip.C.byref(nranks), hip.C.byref(myrank))
    assert (nranks.value, myrank.value) == (world, rank)
    identical, checked = True, 0
    run.i = 0
    for f in range(n_frames):
        run.step()
        if (f + 1) % group == 0 or f + 1 == n_frames:
            run.flush()
            if rank == 0:
                wl.svo.sync()
                torch.cuda.synchronize()
                # the newest frame of the group that came back, against the whole frame of its view rendered on this GPU
                identical = identical and run.frame_is_whole()
                checked += 1
    run.flush()
    wl.svo.sync()
    torch.cuda.synchronize()
    dist.barrier()
    ms, gathers = wl.svo.comm_profile_read()
    if rank == 0:
        Path(out_path).write_text(json.dumps({"identical": bool(identical), "checked": checked, "world": world, "frames_in_flight": run.frames,
                                              "n_max": run.sharder.n_max, "headroom": os.environ.get("VX_COMM_HEADROOM", "default")}))
    wl.svo.comm_destroy()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
