#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + shadow) at 1920x1080 on a synthetic depth-12 SVO (BASELINE.json).

One "step" = one frame: vx_render of this rank's screen tiles (primary ray, shading, one shadow ray per lit pixel,
sky) plus, for N > 1, the RCCL gather of the finished tiles to rank 0 and their assembly into the image.

    python bench.py                      # 1 GPU, C3 workload (configs[2] of BASELINE.json)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line. `roofline.achieved` = algorithmic bytes per launch (step counters of the instrumented
kernel variant x the byte model of DESIGN.md) / the render kernel's average duration, measured with HIP events on the
stream it is launched on. `cpu_baseline` = the C oracle (restatement of the reference's GLSL; the reference has no CPU
raycast) timed on this box's host cores over a bounded sample of the same frame.
"""
import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.3 TB/s achievable)


def algorithmic_bytes(fmt, c):
    """SURVEY.md §8(d) byte model, per frame, from the step counters."""
    nearest = c["leaf_tests"] - c["leaf_tests_trilinear"]
    if fmt == "esvo":
        # 4 B descriptor per iteration, 4 B child pointer per PUSH, per leaf test: pointer + value (8) + material row (32) + texel(s)
        trav = 4 * c["iterations"] + 4 * c["pushes"] + c["leaf_tests"] * (8 + 32) + nearest * 4 + c["leaf_tests_trilinear"] * 32
    else:
        # node header bytes per iteration, pointer bytes per PUSH, per leaf test: u16 material offset + 8 mask bytes + u32 material
        # + material row + texel(s); 5 B chunk frame header per boundary crossing
        trav = (c["csvo_header_bytes"] + c["csvo_pointer_bytes"] + c["leaf_tests"] * (2 + 8 + 4 + 32) + nearest * 4 + c["leaf_tests_trilinear"] * 32
                + 5 * c["boundaries"])
    # per pixel: 16 B RGBA32F store; per lit pixel: 32 B material row + one normal-map sample (counted as one 4 B texel)
    shade = 16 * c["pixels"] + c["lit_pixels"] * (32 + 4)
    return trav + shade


def measured_traffic(fmt):
    """HBM-side bytes per launch of the render kernel on this workload: a STORED artifact -- the committed rocprofv3 --pmc passes
    of this same command (profiles/round3/profile_r3.sh -> profiles/round3/traffic.json, which names the commit it was measured at); PMC
    counters cannot be read from inside this run. Returns (bytes, source) or (None, None)."""
    for rnd in ("round3", "round2", "round1"):
        try:
            t = json.loads((ROOT / "profiles" / rnd / "traffic.json").read_text())
            return t[fmt]["bytes_per_launch"], f"profiles/{rnd}/traffic.json" + (f" @ {t['commit']}" if "commit" in t else "")
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def issue_model(fmt):
    """What bounds the kernel in practice: instruction issue. STORED artifacts: the render kernel's instruction counts per launch from the
    committed --pmc passes (traffic.json: SQ_INSTS_VALU / SALU / VMEM_RD / LDS / SMEM) and the measured cost of an instruction with four
    waves on a SIMD (issue_model.json: profiles/tools/valu_issue.hip). Returns a dict or None."""
    try:
        t = json.loads((ROOT / "profiles" / "round3" / "traffic.json").read_text())
        m = json.loads((ROOT / "profiles" / "round3" / "issue_model.json").read_text())
        r = t[fmt]
        insts = sum(float(r.get(k) or 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"))
        clock = float(m["clock_mhz"][fmt])
        bound_ms = insts * float(m["simd_cycles_per_instruction"]) / float(m["simds"]) / (clock * 1e3)
        return {"bound": "instruction issue (every SIMD issuing its waves' instructions back to back)", "instructions_per_launch": int(insts),
                "valu_per_launch": int(r["SQ_INSTS_VALU"]), "salu_per_launch": int(r["SQ_INSTS_SALU"]), "valu_lane_utilisation": r.get("valu_lane_utilisation"),
                "simd_cycles_per_instruction": m["simd_cycles_per_instruction"], "simds": m["simds"], "clock_mhz_in_kernel": clock,
                "issue_bound_ms": round(bound_ms, 4), "source": f"profiles/round3/traffic.json @ {t.get('commit', '?')}, profiles/round3/issue_model.json"}
    except (OSError, KeyError, ValueError, TypeError):
        return None


def image_model_bytes(c):
    """What the kernel that is TIMED fetches by the same accounting: it walks the traversal image, whichever format the world is
    in -- one 8-byte entry per PUSH, a 4-byte value + the 32-byte material row + texel(s) per leaf test, nothing per iteration --
    plus the same per-pixel terms."""
    nearest = c["leaf_tests"] - c["leaf_tests_trilinear"]
    return 8 * c["pushes"] + c["leaf_tests"] * (4 + 32) + nearest * 4 + c["leaf_tests_trilinear"] * 32 + 16 * c["pixels"] + c["lit_pixels"] * (32 + 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--repeats", type=int, default=25, help="the timed block of --steps frames is run this many times; the MEDIAN block is reported")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--format", choices=["esvo", "csvo"], default="csvo", help="node format; csvo is the reference's default build feature")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames the renderer keeps in flight (1..8); default: the library's own (2) on one GPU, max(3, N) when the frame is sharded over N "
                         "(a sharded frame's next render waits for the exchange and rank 0's assembly of the frame before last on its stream)")
    ap.add_argument("--gather-group", type=int, default=0, help="sharded: frames per gather (default 1)")
    ap.add_argument("--gather", choices=["auto", "library", "torch"], default="auto",
                    help="sharded: the exchange step -- library: vx_gather_tiles (RCCL send/receive owned by the render context); torch: torch.distributed.gather; "
                         "auto: the library's, checked on its first frames (watchdog + the assembled frame against the whole render), torch's if that fails")
    ap.add_argument("--gather-format", choices=["rgba8", "rgba32f"], default="rgba8",
                    help="sharded: pixel format of the tile lists that travel and of rank 0's image (rgba8 = Framebuffer::as_image's bytes: a quarter of the link time)")
    ap.add_argument("--gather-timeout", type=float, default=30.0, help="sharded: seconds the first exchange may take before it is declared hung")
    ap.add_argument("--simulate-gather-failure", action="store_true", help="testing: make the library's exchange fail, to exercise the fall-back")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sd500", action="store_true", help="skip the second line (the same frame with the game's shadow_distance = 500): profiler runs, whose per-kernel averages it would dilute")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the N > 1 code path (tile lists, RCCL gather, assembly) even with one rank; needs a torch.distributed.run launch")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-clock target for the cpu_baseline sample (all host cores)")
    ap.add_argument("--textures", choices=["assets", "procedural"], default="assets",
                    help="assets: the reference's own 64x64 textures (tests/golden/textures) for the blocks the terrain uses")
    args = ap.parse_args()

    import numpy as np
    import torch

    from _pkg import load_package

    vra = load_package()
    from voxel_rs_amd import hip, scenes

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if rank != 0:
        os.dup2(2, 1)  # only rank 0 owns stdout (the one JSON line): whatever a library prints on the other ranks goes to stderr
    if world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    sharded = world_size > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist

        if "RANK" not in os.environ:  # --force-sharded started as a plain script: a one-rank group of its own
            import socket

            with socket.socket() as probe:
                probe.bind(("127.0.0.1", 0))
                port = probe.getsockname()[1]
            os.environ.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # nccl == RCCL on ROCm

    fmt = vra.SVO_ESVO if args.format == "esvo" else vra.SVO_CSVO
    W, H = args.width, args.height

    # ---- scene: replicated per GPU (every rank builds and uploads the same serialized SVO) ------------------------------
    t0 = time.time()
    world = vra.World(fmt)
    st = world.build_heightfield(args.depth)
    build_s = time.time() - t0
    tex = scenes.asset_textures(ROOT / "tests" / "golden" / "textures") if args.textures == "assets" else scenes.synthetic_textures()
    mats = scenes.synthetic_materials()
    svo = hip.Svo(fmt, world.size_in_bytes + (16 << 20), device=local_rank)
    svo.set_materials(mats)
    svo.set_textures(tex, 6)
    t0 = time.time()
    svo.update(world)
    upload_s = time.time() - t0
    # every primary hit casts its shadow ray ("primary + 1 shadow ray" of configs[2]); the game's default cut-off of
    # 500 blocks would cast almost none from this altitude
    uniforms = scenes.bench_camera(args.depth, st["h_max"], W, H, shadow_distance=3.0e38, render_shadows=True)

    # ---- work per frame: deterministic for a fixed scene/camera -----------------------------------------------------------
    counters = svo.render_counters(uniforms, W, H, rank, world_size)
    my_rays = counters["rays"]
    my_bytes = algorithmic_bytes(args.format, counters)

    gather_used, gather_note, comm_hung = None, None, False
    if sharded:
        from voxel_rs_amd.sharding import FrameSharder

        # What travels: the tiles as RGBA8 (Framebuffer::as_image's bytes, vx_format) by default. The exchange is bound by the links
        # into rank 0 -- (N - 1) / N of every frame, one xGMI link per peer: a 1080p RGBA32F frame is 33 MB, i.e. 16.6 / 8.3 / 4.2 MB
        # per link at N = 2 / 4 / 8, which at ~77 GB/s a direction is 1.3 x the time the ranks take to RENDER their shares; RGBA8 is
        # a quarter of that, and a quarter of rank 0's assembly pass.
        vx_fmt = hip.VX_FORMAT_RGBA8 if args.gather_format == "rgba8" else hip.VX_FORMAT_RGBA32F
        # frames in flight: the smaller a rank's share of the frame, the longer its tail relative to its body (a frame cannot
        # finish before its longest ray) and the more frames it takes to keep the device full
        FRAMES = min(8, max(1, args.frames_in_flight)) if args.frames_in_flight else min(8, max(3, world_size))
        # frames per collective (1: every frame is gathered as soon as it is rendered; more: fewer, larger messages)
        GROUP = args.gather_group if args.gather_group else 1
        GROUP = max(1, min(GROUP, FRAMES))
        FRAMES -= FRAMES % GROUP
        svo.set_frames_in_flight(FRAMES)
        gather_events = []  # torch path: (start, stop) event pairs around the collective, on torch's stream

        def render_tiles(tiles):
            svo.render_device(uniforms, W, H, tiles.data_ptr(), tile_rank=rank, tile_count=world_size, fmt=vx_fmt)

        def build_sharder(library_gather):
            """The whole N > 1 path with one of the two exchanges: the library's own (vx_gather_tiles: grouped ncclSend / ncclRecv on the
            render context's communicator and stream) or torch.distributed.gather (nccl backend = RCCL as well)."""
            exchange_stream = svo.comm_stream if library_gather else torch.cuda.current_stream().cuda_stream
            exchange_done = [torch.cuda.Event() for _ in range(FRAMES // GROUP)]
            recorded = [False] * (FRAMES // GROUP)
            tickets = {}

            def assemble(gathered, image):
                # on the exchange's stream: ordered after the gather by construction, and the render streams stay free for the
                # next frame's tiles. `gathered` is [rank][tile]... with the ranks a whole group of frames apart.
                svo.assemble_tiles_format(gathered.data_ptr(), gathered.stride(0) // 4, world_size, W, H, image.data_ptr(), vx_fmt, exchange_stream)

            def before_render(g):
                # a render into a tile list waits for whatever still reads it: the gather AND rank 0's assembly (the ticket covers both)
                if library_gather:
                    if g in tickets:
                        svo.wait_gather(tickets[g])
                elif recorded[g]:
                    svo.wait_event(exchange_done[g].cuda_event)

            def after_render():
                if not library_gather:
                    svo.stream_wait_render(torch.cuda.current_stream().cuda_stream)

            def after_exchange(g):
                if library_gather:
                    if rank == 0:
                        tickets[g] = state["last_ticket"]  # (re-recorded behind the assembly by vx_assemble_tiles_format)
                else:
                    exchange_done[g].record(torch.cuda.current_stream())
                    recorded[g] = True

            state = {"last_ticket": None}

            def gather(tiles, gathered):
                g = sh._g  # (the group being exchanged)
                if library_gather:
                    if args.simulate_gather_failure:
                        raise RuntimeError("simulated failure of the library's gather (--simulate-gather-failure)")
                    state["last_ticket"] = tickets[g] = svo.gather_tiles(tiles.data_ptr(), tiles.numel() * tiles.element_size(),
                                                                        gathered.data_ptr() if rank == 0 else None, root=0)
                else:
                    timed = svo_profile["on"]
                    if timed:
                        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                        ev[0].record()
                    dist.gather(tiles, [gathered[r] for r in range(world_size)] if rank == 0 else None, dst=0)
                    if timed:
                        ev[1].record()
                        gather_events.append(ev)

            sh = FrameSharder(W, H, rank, world_size, dist, "cuda", render_tiles, assemble, before_render=before_render, after_render=after_render,
                              after_exchange=after_exchange, buffers=FRAMES, group=GROUP, gather=gather, pixel_format=args.gather_format)
            if not library_gather:
                # (torch's gather wants its own send buffer: the root's list is not rendered in place)
                sh.tiles = [torch.zeros((GROUP, sh.n_max, 32, 32, 4), dtype=sh.dtype, device="cuda") for _ in range(FRAMES // GROUP)]
                torch.cuda.synchronize()
            sh.query = (lambda: svo.gather_query(state["last_ticket"]) if state["last_ticket"] is not None else 1) if library_gather else None
            return sh

        svo_profile = {"on": False}

        def whole_frame():
            img = torch.zeros((H, W, 4), dtype=torch.uint8 if vx_fmt == hip.VX_FORMAT_RGBA8 else torch.float32, device="cuda")
            torch.cuda.synchronize()
            svo.render_device(uniforms, W, H, img.data_ptr(), fmt=vx_fmt)
            svo.sync()
            return img

        def frame_is_whole(sh):
            """rank 0: what the path delivered, against the same frame rendered whole on this GPU"""
            whole = whole_frame()
            if world_size > 1:
                got = sh.image
            else:
                # one rank (--force-sharded): tile_count 1 means "the whole frame, row-major" to vx_render, so this rank's "tile list" is the
                # frame itself and the assembly kernel (which runs for its cost) scatters something that is no tile list -- the check is
                # on what the gather delivered. (An RGBA8 frame is stored top row first, a tile list bottom row first: same here, the
                # "list" IS the frame.)
                got = sh.last_gathered[0].reshape(-1)[:H * W * 4].view(H, W, 4)
            a, b = whole.contiguous().view(torch.uint8), got.contiguous().view(torch.uint8)
            return bool(torch.equal(a, b))

        def first_frames_ok(sh, seconds):
            """GROUP frames through the whole path, watched: the exchange must come back within `seconds` on every rank (a collective a
            peer never joins would otherwise hang the benchmark) and rank 0's assembled frame must be the whole render's."""
            ok = 1
            try:
                for _ in range(GROUP):
                    sh.step()
                sh.flush()
                if sh.query is not None:
                    deadline = time.time() + seconds
                    while True:
                        q = sh.query()
                        if q != 0:
                            ok = 1 if q == 1 else 0
                            break
                        if time.time() > deadline:
                            ok = 0
                            break
                        time.sleep(0.002)
                if ok:
                    svo.sync()
                    torch.cuda.synchronize()
                    if rank == 0:
                        ok = 1 if frame_is_whole(sh) else 0
            except Exception as e:  # an error code from the library (VX_ERR_HIP with RCCL's message), or the simulated one
                print(f"[bench rank {rank}] exchange failed: {e}", file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(int(flag.item()))

        want_library = args.gather in ("auto", "library")
        sharder = None
        if want_library:
            try:
                # the render context owns the RCCL communicator the tiles travel over (vx_comm_init); torch.distributed only carries the
                # id to the ranks and the barriers / statistics of this script
                uid = [hip.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                svo.comm_init(world_size, rank, uid[0])
                sharder = build_sharder(True)
                init_ok = 1
            except Exception as e:
                print(f"[bench rank {rank}] vx_comm_init failed: {e}", file=sys.stderr)
                init_ok = 0
            flag = torch.tensor([init_ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) and first_frames_ok(sharder, args.gather_timeout):
                gather_used = "library"
            elif args.gather == "library":
                raise SystemExit("--gather library: the library's exchange failed its first frames (see stderr); --gather auto falls back to torch.distributed")
            else:
                # never re-exec a process that has touched the GPU: the other exchange, in this process. The library's communicator is left
                # alone (destroying one with a collective in flight can block).
                gather_note = "the library's vx_gather_tiles failed its first frames on this node; fell back in-process"
                comm_hung = True
                sharder = None
        if sharder is None:
            sharder = build_sharder(False)
            gather_used = "torch"
            if not first_frames_ok(sharder, args.gather_timeout):
                gather_note = (gather_note + "; " if gather_note else "") + "torch.distributed.gather's first frames did not reproduce the whole render"
        step = sharder.step
        flush = sharder.flush
    else:
        # frames in flight (the library rotates over that many streams; its default is 2): one image per frame in flight
        FRAMES = min(8, max(1, args.frames_in_flight)) if args.frames_in_flight else 2
        svo.set_frames_in_flight(FRAMES)
        images = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(FRAMES)]
        torch.cuda.synchronize()  # zero fills run on torch's stream, the renderer on its own
        state = {"i": 0}

        def step():
            svo.render_device(uniforms, W, H, images[state["i"] % FRAMES].data_ptr())
            state["i"] += 1

        def flush():
            pass

    def barrier():
        flush()  # (sharded: a group of frames that has not been exchanged yet)
        svo.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # The timed block -- exactly --steps frames between two barriers -- is run --repeats times back to back and the MEDIAN block is
    # what is reported: 50 frames are 25 ms of GPU time, too little for one sample to stand on.
    svo.profile_enable(True)
    if sharded:
        svo_profile["on"] = True
    blocks, enqueue = [], []
    for _ in range(max(args.repeats, 1)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        enqueue.append(time.perf_counter() - t0)  # host time to issue the steps (the loop is asynchronous)
        barrier()
        blocks.append(time.perf_counter() - t0)
    kernel_ms, launches = svo.profile_read()
    # the exchanges of the timed region: time on the exchange's stream from the first send / receive to the last (the wait for the
    # slowest peer included)
    gather_ms, gathers = 0.0, 0
    if sharded:
        svo_profile["on"] = False
        if gather_used == "library":
            gather_ms, gathers = svo.comm_profile_read()
        else:
            torch.cuda.synchronize()
            gather_ms, gathers = sum(a.elapsed_time(b) for a, b in gather_events), len(gather_events)
            gather_events.clear()
    svo.profile_enable(False)
    enqueue_s = sorted(enqueue)[len(enqueue) // 2]

    # One frame at a time (outside the timed region): with several frames in flight a kernel's HIP-event span includes the time it
    # shares the device with its neighbours, so the kernel's OWN duration -- what the roofline fraction is defined on -- is measured
    # with the device to itself, twice:
    #  (a) the timed region's own launch policy: the frame streams, every frame waited for before the next is issued;
    #  (b) the library's one-frame-at-a-time mode (set_frames_in_flight(1): the context's own stream, where it hands out a view's
    #      sub-tiles most expensive first -- the table is made from the frame before last -- so that the longest rays start early):
    #      what a consumer that presents every frame runs, and the same command rocprofv3 is run on for profiles/ (VX_FRAMES_IN_FLIGHT=1).
    barrier()
    svo.profile_enable(True)
    for _ in range(20):
        step()
        flush()
        svo.sync()
    barrier()
    exclusive_ms, exclusive_launches = svo.profile_read()
    if sharded and gather_used == "library":
        svo.comm_profile_read()
    svo.profile_enable(False)
    kernel_exclusive_frame_stream_ms = exclusive_ms / max(exclusive_launches, 1)
    kernel_exclusive_ms = kernel_exclusive_frame_stream_ms
    if not sharded:
        svo.set_frames_in_flight(1)
        for _ in range(4):
            step()
        barrier()
        svo.profile_enable(True)
        for _ in range(20):
            step()
        barrier()
        exclusive_ms, exclusive_launches = svo.profile_read()
        svo.profile_enable(False)
        svo.set_frames_in_flight(FRAMES)
        kernel_exclusive_ms = exclusive_ms / max(exclusive_launches, 1)

    # The game's own shadow cut-off (500 blocks, src/gamelogic/world.rs:105-108; SURVEY.md 8d asks for both): the same frame with
    # shadow_distance = 500 -- from this altitude few or no hits are that near, so it is close to a primary-rays-only frame. One GPU only.
    sd500 = None
    if not sharded and not args.no_sd500:
        u500 = scenes.bench_camera(args.depth, st["h_max"], W, H, shadow_distance=500.0, render_shadows=True)
        rays500 = svo.render_counters(u500, W, H, 0, 1)["rays"]
        for _ in range(args.warmup):
            svo.render_device(u500, W, H, images[0].data_ptr())
        svo.sync()
        t500 = []
        for _ in range(5):
            svo.sync()
            t0 = time.perf_counter()
            for i in range(args.steps):
                svo.render_device(u500, W, H, images[i % FRAMES].data_ptr())
            svo.sync()
            t500.append(time.perf_counter() - t0)
        ms500 = sorted(t500)[2] / args.steps * 1e3
        sd500 = {"shadow_distance": 500.0, "rays_per_frame": int(rays500), "ms_per_step": round(ms500, 4), "value": round(rays500 / (ms500 * 1e-3) / 1e6, 3),
                 "unit": "Mrays/s", "note": "the timed frame casts a shadow ray from every primary hit (shadow_distance = inf); this is the game's default cut-off"}

    times = torch.tensor(blocks, dtype=torch.float64, device="cuda")
    stats = torch.tensor([float(my_rays), float(my_bytes), kernel_ms / max(launches, 1), kernel_exclusive_ms, gather_ms / max(gathers, 1)],
                         dtype=torch.float64, device="cuda")
    per_rank = None
    if dist is not None:
        dist.all_reduce(times, op=dist.ReduceOp.MAX)  # per block: the slowest rank
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        total_rays = float(sm[0])
        every = [torch.zeros_like(stats) for _ in range(world_size)]
        dist.all_gather(every, stats)
        per_rank = [{"rank": r, "rays": int(e[0]), "kernel_span_ms_in_flight": round(float(e[2]), 4), "kernel_exclusive_ms": round(float(e[3]), 4),
                     "exchange_ms": round(float(e[4]), 4)} for r, e in enumerate(every)]
    else:
        total_rays = float(my_rays)
    block_s = sorted(float(t) for t in times)
    elapsed = block_s[len(block_s) // 2]
    # sharded: the frame rank 0 assembled last against the same frame rendered whole on this GPU (outside the timed region)
    sharded_frame_identical = None
    if sharded and rank == 0:
        sharded_frame_identical = frame_is_whole(sharder)
    if rank != 0:
        if sharded and gather_used == "library":
            svo.comm_destroy()
        if dist is not None:
            dist.destroy_process_group()
        if comm_hung:
            os._exit(0)  # (a communicator with a collective that never completed: its teardown can block)
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = total_rays / (ms_per_step * 1e-3) / 1e6  # Mrays/s, whole job
    kernel_avg_ms = kernel_ms / max(launches, 1)
    achieved = my_bytes / (kernel_exclusive_ms * 1e-3) / 1e9 if kernel_exclusive_ms > 0 else 0.0
    traffic, traffic_source = measured_traffic(args.format) if (W, H, args.depth, world_size) == (1920, 1080, 12, 1) else (None, None)
    roofline = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                "traffic": traffic, "traffic_source": traffic_source, "kernel": "render_persistent",
                # achieved = algorithmic bytes per launch / the kernel's own duration: one frame at a time on one stream, HIP events around
                # each launch (20 launches after the timed region); the same figure as rocprofv3's average with VX_FRAMES_IN_FLIGHT=1
                "kernel_exclusive_ms": round(kernel_exclusive_ms, 4),
                # per-launch event span inside the timed region: with frames in flight the spans overlap (span x launches > elapsed)
                "kernel_span_ms_in_flight": round(kernel_avg_ms, 4), "launches": launches,
                "algorithmic_bytes_per_launch": int(my_bytes), "bytes_per_ray": round(my_bytes / max(my_rays, 1), 2),
                "byte_model": "the reference's own fetches (SURVEY.md 8d), counted by the instrumented kernel on the world's own bytes",
                "image_model_bytes_per_launch": int(image_model_bytes(counters)),
                "frames_in_flight": FRAMES, **({"frames_per_gather": GROUP} if sharded else {}),
                "kernel_exclusive_mode": ("one frame at a time on the context's own stream (set_frames_in_flight(1)): sub-tiles handed out most expensive first"
                                          if not sharded else "every frame waited for before the next is issued; the timed region's streams and launch policy"),
                # the same with the timed region's own launch policy (frame streams, screen order), every frame waited for
                "kernel_exclusive_ms_timed_policy": round(kernel_exclusive_frame_stream_ms, 4),
                "timed_region_mode": f"{FRAMES} frames in flight (ms_per_step); kernel_span_ms_in_flight is a launch's own event span there",
                # the timed frames are one view, rendered again and again: what the library keeps between frames of such a view
                "temporal_reuse": ("scheduling only, identical pixels: a still view's frames are rendered in sorted passes (64 pixels of a 16x16 block put "
                                   "together by what they cost in earlier frames; VX_SORTED=0: 8 % slower), the one-frame-at-a-time pass also hands work "
                                   "out most expensive first; under a moving camera neither applies: profiles/round3/pass_aq"),
                # what the device sustains over the median timed block: bytes x frames / elapsed
                "sustained_GBps": round(my_bytes * args.steps / max(elapsed, 1e-9) / 1e9, 3)}
    # The HBM byte model is what the contract asks for, but this kernel's working set is cache resident and it is bound by instruction
    # issue: the second, practical bound, with the fraction of it the kernel reaches one frame at a time and in the timed mode.
    issue = issue_model(args.format) if (W, H, args.depth, world_size) == (1920, 1080, 12, 1) else None
    if issue:
        issue["frac_of_bound_one_frame_at_a_time"] = round(issue["issue_bound_ms"] / kernel_exclusive_ms, 4) if kernel_exclusive_ms > 0 else None
        issue["frac_of_bound_timed_mode"] = round(issue["issue_bound_ms"] / (elapsed / args.steps * 1e3), 4)
        roofline["issue"] = issue

    cpu = None
    if not args.no_cpu_baseline and world_size == 1:  # rank 0 at N = 1 only: a baseline of the workload, not of the scaling run
        from oracle import oracle as orc  # the checker, timed here as the CPU baseline ("port": the reference has no CPU raycast)

        scene = orc.OracleScene(fmt, world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
        ou = orc.Uniforms.from_buffer_copy(bytes(uniforms))
        cores = orc.lib().or_max_threads()
        # warm up the thread pool and the page cache on a thin band, then time one band of a tenth of the frame to decide
        # between whole frames and bands (a frame this size takes well under a second on a server CPU)
        scene.render(ou, W, H, rect=(0, H // 2, W, H // 2 + 8), want_hits=False, counters=orc.Counters(), threads=cores)
        t0 = time.perf_counter()
        scene.render(ou, W, H, rect=(0, H // 2 - H // 20, W, H // 2 + H // 20), want_hits=False, counters=orc.Counters(), threads=cores)
        frame_estimate_s = 10.0 * (time.perf_counter() - t0)
        cc = orc.Counters()
        if frame_estimate_s <= args.cpu_seconds:
            # repeat the whole frame until about cpu_seconds of wall time on all host cores have been timed
            reps = 0
            t0 = time.perf_counter()
            while reps < 400 and (reps == 0 or time.perf_counter() - t0 < args.cpu_seconds):
                scene.render(ou, W, H, want_hits=False, counters=cc, threads=cores)
                reps += 1
            cpu_s = time.perf_counter() - t0
            sample = f"{reps} x the whole {W}x{H} frame"
        else:
            bands = 8
            band_h = max(int(H * args.cpu_seconds / frame_estimate_s) // bands, 1)
            t0 = time.perf_counter()
            for b in range(bands):
                y0 = int((b + 0.5) * H / bands) - band_h // 2
                scene.render(ou, W, H, rect=(0, max(y0, 0), W, min(y0 + band_h, H)), want_hits=False, counters=cc, threads=cores)
            cpu_s = time.perf_counter() - t0
            sample = f"{bands} bands x {band_h} rows of the {W}x{H} frame"
        # and on ONE thread ("1 thread, and all host cores"): bands of rows spread over the frame, about 4 s of work
        c1 = orc.Counters()
        scene.render(ou, W, H, rect=(0, H // 2, W, H // 2 + 2), want_hits=False, counters=c1, threads=1)  # (warm)
        t0 = time.perf_counter()
        scene.render(ou, W, H, rect=(0, H // 2, W, H // 2 + 16), want_hits=False, counters=c1, threads=1)
        per_row_s = (time.perf_counter() - t0) / 16
        rows = max(2, min(H // 8, int(4.0 / max(per_row_s, 1e-6)) // 8))
        c1 = orc.Counters()
        t0 = time.perf_counter()
        for b in range(8):
            y0 = int((b + 0.5) * H / 8) - rows // 2
            scene.render(ou, W, H, rect=(0, max(y0, 0), W, min(y0 + rows, H)), want_hits=False, counters=c1, threads=1)
        one_s = time.perf_counter() - t0
        cpu = {"value": round(cc.rays / cpu_s / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
               "sample": f"{sample}: {cc.rays} rays in {cpu_s:.2f} s (C restatement of the GLSL path, OpenMP; the reference has no CPU raycast)",
               "single_thread": {"value": round(c1.rays / one_s / 1e6, 4), "unit": "Mrays/s", "cores": 1,
                                 "sample": f"8 bands x {rows} rows of the {W}x{H} frame: {c1.rays} rays in {one_s:.2f} s"}}

    out = {
        "metric": "Mrays/sec (primary+shadow) at 1920x1080, depth-12 SVO; achieved HBM GB/s",
        "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
        "repeats": len(block_s), "block_ms_min_median_max": [round(block_s[0] * 1e3, 3), round(elapsed * 1e3, 3), round(block_s[-1] * 1e3, 3)],
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"C3: {W}x{H} primary + 1 shadow ray per lit pixel, textured + normal-mapped shading, depth-{args.depth} SVO "
                               f"({args.format.upper()} nodes), 1 frame per step", "svo_format": args.format, "svo_bytes": world.size_in_bytes,
                   "leaves": st["leaves"], "chunks": st["chunks"], "textures": args.textures, "rays_per_frame": int(total_rays), "primary_rays": W * H,
                   "parallelism": f"screen tiles (32x32, Morton order, round-robin) over {world_size} GPU(s), SVO replicated, RCCL gather to rank 0",
                   **({"rccl_ranks": world_size, "gather": "vx_gather_tiles (grouped ncclSend/ncclRecv on the render context's own communicator)" if gather_used == "library"
                       else "torch.distributed.gather (nccl backend)", "gather_requested": args.gather, **({"gather_note": gather_note} if gather_note else {}),
                       "gather_format": args.gather_format, "bytes_gathered_per_frame": int((world_size - 1) * sharder.n_max * 1024 * (4 if args.gather_format == "rgba8" else 16)),
                       "exchange_ms_per_gather_rank0": round(gather_ms / max(gathers, 1), 4), "per_rank": per_rank} if sharded else {}),
                   **({"sharded_frame_identical_to_whole_render": sharded_frame_identical} if sharded else {}),
                   "scene_build_s": round(build_s, 2), "upload_s": round(upload_s, 3),
                   "host_issue_ms_per_step": round(enqueue_s / args.steps * 1e3, 4)},
        "roofline": roofline, "cpu_baseline": cpu, **({"shadow_distance_500": sd500} if sd500 else {}),
    }
    # the JSON line is the LAST thing on stdout: tear the communicators down first (RCCL prints a banner through C stdio, which
    # is flushed at exit otherwise) and flush C's buffers before Python's
    if sharded and gather_used == "library":
        svo.comm_destroy()
    if dist is not None:
        dist.destroy_process_group()
    import ctypes

    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    print(json.dumps(out), flush=True)
    if comm_hung:
        os._exit(0)


if __name__ == "__main__":
    main()
