#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + shadow) at 1920x1080 on a synthetic depth-12 SVO (BASELINE.json).

One "step" = one frame of a camera that MOVES EVERY FRAME, as the reference's frame loop moves it (src/gamelogic/game.rs:111-148): the
view turns by a quarter of a degree and walks a twelfth of a block per frame. A frame = vx_render of this rank's screen tiles (primary
ray, shading, one shadow ray per lit pixel, sky) plus, for N > 1, the RCCL gather of the finished tiles to rank 0 and their assembly.

    python bench.py                      # 1 GPU, C3 workload (configs[2] of BASELINE.json)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line. Two measurements of the same frames stand behind `value`: the MEDIAN of --repeats timed blocks of exactly --steps
frames each (a block is a few milliseconds), and one SUSTAINED block of --sustained-seconds of the same path (the reference's own harness
samples for 20 s, benchmark-ingame.py:37); `value` is the burst median unless the sustained figure is more than 2 % below it, then the sustained
one (`value_source` says which). `roofline.achieved` = algorithmic bytes per launch (step counters of the instrumented kernel variant x the
byte model of DESIGN.md, averaged over the K views) / the render kernel's average duration, measured with HIP events on the stream it is launched
on. `roofline.issue.clock_mhz_in_kernel` is sampled in THIS run (vx_clock_probe, while the sustained block renders). `configs` = the other
BASELINE configurations (C2, C4 static and streamed, C5) on this GPU; `forced_sharded` = the N > 1 code path on this one GPU (a child process).
`cpu_baseline` = the C oracle (restatement of the reference's GLSL; the reference has no CPU raycast) timed on this box's host cores over
whole frames of the first view at 1 / 8 / 32 / all threads, with what the box grants this process (affinity, cgroup quota) stated.
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
PROFILE_ROUNDS = ("round6", "round5", "round4", "round3", "round2", "round1")


# ---- byte models ---------------------------------------------------------------------------------------------------


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


def image_model_bytes(c):
    """What the kernel that is TIMED fetches by the same accounting: it walks the traversal image, whichever format the world is
    in -- one 8-byte entry per PUSH, a 4-byte value + the 32-byte material row + texel(s) per leaf test, nothing per iteration --
    plus the same per-pixel terms."""
    nearest = c["leaf_tests"] - c["leaf_tests_trilinear"]
    return 8 * c["pushes"] + c["leaf_tests"] * (4 + 32) + nearest * 4 + c["leaf_tests_trilinear"] * 32 + 16 * c["pixels"] + c["lit_pixels"] * (32 + 4)


def stored_counters(fmt, config=None):
    """The committed rocprofv3 --pmc passes of this same command (profiles/roundN/profile*.sh -> profiles/roundN/traffic.json): PMC counters cannot
    be read from inside this run, so they are a STORED artifact -- and only quoted when they were measured on the library sources this run uses:
    traffic.json names the hash of voxel-rs_amd/csrc/hip (`csrc_sha16`, _pkg.csrc_hash), and any other hash makes `traffic` and `issue` null with a
    note instead of stale numbers. `config`: a key of the file other than the headline's ("C4": the depth-14 frames). Returns (row or None, source, note)."""
    from _pkg import csrc_hash

    here = csrc_hash()
    for rnd in PROFILE_ROUNDS:
        try:
            t = json.loads((ROOT / "profiles" / rnd / "traffic.json").read_text())
        except (OSError, ValueError):
            continue
        src = f"profiles/{rnd}/traffic.json" + (f" @ {t['commit']}" if "commit" in t else "")
        if t.get("csrc_sha16") != here:
            return None, src, (f"{src} was measured on other library sources (csrc_sha16 {t.get('csrc_sha16', 'not recorded')}, this run {here}): "
                               "not quoted; re-run the profile passes (profiles/round6/run_profiles.sh)")
        return (t.get(config) or {}).get(fmt) if config else t.get(fmt), src, None
    return None, None, "no stored counter file"


def issue_model(row, source, clock_mhz, clock_source):
    """What bounds the kernel in practice: instruction issue. The render kernel's instruction counts per launch from the stored --pmc passes
    (SQ_INSTS_VALU / SALU / VMEM_RD / LDS / SMEM) x the measured cost of an instruction with four waves on a SIMD (profiles/round3/issue_model.json:
    profiles/tools/valu_issue.hip) / 1024 SIMDs / the shader clock measured in this run."""
    try:
        m = json.loads((ROOT / "profiles" / "round3" / "issue_model.json").read_text())
        insts = sum(float(row.get(k) or 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"))
        bound_ms = insts * float(m["simd_cycles_per_instruction"]) / float(m["simds"]) / (clock_mhz * 1e3)
        return {"bound": "instruction issue (every SIMD issuing its waves' instructions back to back)", "instructions_per_launch": int(insts),
                "valu_per_launch": int(row["SQ_INSTS_VALU"]), "salu_per_launch": int(row["SQ_INSTS_SALU"]), "valu_lane_utilisation": row.get("valu_lane_utilisation"),
                "wait_share_of_wave_cycles": (round(row["SQ_WAIT_ANY"] / row["SQ_WAVE_CYCLES"], 3) if row.get("SQ_WAIT_ANY") and row.get("SQ_WAVE_CYCLES") else None),
                "simd_cycles_per_instruction": m["simd_cycles_per_instruction"], "simds": m["simds"], "clock_mhz_in_kernel": round(clock_mhz, 1),
                "clock_source": clock_source, "issue_bound_ms": round(bound_ms, 4), "source": f"{source}, profiles/round3/issue_model.json"}
    except (OSError, KeyError, ValueError, TypeError):
        return None


# ---- the workload -----------------------------------------------------------------------------------------------------------------


def moving_uniforms(scenes, depth, h_max, W, H, i, shadows=True):
    """Frame i of the camera's path: the §8d view, turned by a quarter of a degree a frame about the vertical and walked forward at 5
    blocks a second (at 60 frames a second). Every primary hit casts its shadow ray ("primary + 1 shadow ray" of configs[2]): the game's
    default cut-off of 500 blocks would cast almost none from this altitude (the `shadow_distance_500` object)."""
    n = float(1 << depth)
    a = math.radians(0.25 * i)
    fwd = (0.6 * math.cos(a) - 0.7 * math.sin(a), -0.35, 0.6 * math.sin(a) + 0.7 * math.cos(a))
    eye = (0.5 * n + 0.05 * i, h_max + 0.05 * n, 0.5 * n + 0.066 * i)
    return scenes.render_params_to_uniforms(eye, fwd, (0.0, 1.0, 0.0), math.radians(72.0), W / H, 0.3, (-1.0, -1.0, -1.0), shadows, 3.0e38)


class Workload:
    """The scene (replicated per GPU: every rank builds and uploads the same serialized SVO), the camera's path and the work per frame of this rank."""

    def __init__(self, args, vra, hip, scenes, rank, world_size, local_rank):
        self.args, self.rank, self.world_size = args, rank, world_size
        self.fmt = vra.SVO_ESVO if args.format == "esvo" else vra.SVO_CSVO
        W, H = args.width, args.height
        t0 = time.time()
        self.world = vra.World(self.fmt)
        self.st = self.world.build_heightfield(args.depth)
        self.build_s = time.time() - t0
        self.tex = scenes.asset_textures(ROOT / "tests" / "golden" / "textures") if args.textures == "assets" else scenes.synthetic_textures()
        self.mats = scenes.synthetic_materials()
        self.svo = hip.Svo(self.fmt, self.world.size_in_bytes + (16 << 20), device=local_rank)
        self.svo.set_materials(self.mats)
        self.svo.set_textures(self.tex, 6)
        t0 = time.time()
        self.svo.update(self.world)
        self.upload_s = time.time() - t0
        # the K views of a timed block (every block walks the same path: its first view follows its last one, a cut)
        self.path = [moving_uniforms(scenes, args.depth, self.st["h_max"], W, H, i) for i in range(args.steps)]
        self.still = scenes.bench_camera(args.depth, self.st["h_max"], W, H, shadow_distance=3.0e38, render_shadows=True)
        # work per frame: deterministic for a fixed scene / camera, counted by the instrumented kernel for this rank's tiles
        self.counters = [self.svo.render_counters(u, W, H, rank, world_size) for u in self.path]
        self.rays_per_block = sum(c["rays"] for c in self.counters)
        self.iterations_per_block = sum(c["iterations"] for c in self.counters)
        self.bytes_per_frame = sum(algorithmic_bytes(args.format, c) for c in self.counters) / len(self.counters)
        self.image_bytes_per_frame = sum(image_model_bytes(c) for c in self.counters) / len(self.counters)


class SingleGpu:
    """Frames into device memory, FRAMES in flight (the library rotates over that many streams; its default is 2): one image per frame in flight."""

    sharded = False

    def __init__(self, args, wl, torch):
        self.wl, self.svo, self.W, self.H = wl, wl.svo, args.width, args.height
        self.frames = min(8, max(1, args.frames_in_flight)) if args.frames_in_flight else 2
        self.svo.set_frames_in_flight(self.frames)
        self.images = [torch.zeros((self.H, self.W, 4), dtype=torch.float32, device="cuda") for _ in range(self.frames)]
        torch.cuda.synchronize()  # zero fills run on torch's stream, the renderer on its own
        self.i = 0
        self.view = None  # None: the path; else one view again and again
        self.profile_on = False

    def step(self):
        u = self.view if self.view is not None else self.wl.path[self.i % len(self.wl.path)]
        self.svo.render_device(u, self.W, self.H, self.images[self.i % self.frames].data_ptr())
        self.i += 1

    def flush(self):
        pass


class Sharded:
    """The N > 1 path: every rank renders its tiles of the frame into a compact list, the lists are gathered to rank 0 and assembled.
    What travels: the tiles as RGBA8 (Framebuffer::as_image's bytes) by default -- the exchange is bound by the links into rank 0, (N - 1) / N of
    every frame, one xGMI link per peer: a 1080p RGBA32F frame is 33 MB, i.e. 16.6 / 8.3 / 4.2 MB per link at N = 2 / 4 / 8, which at ~77 GB/s a
    direction is 1.3 x the time the ranks take to RENDER their shares; RGBA8 is a quarter of that, and of rank 0's assembly pass.
    The exchange: the library's own (vx_gather_tiles: grouped ncclSend / ncclRecv on the render context's communicator and stream), checked on
    its first frames -- a watchdog on vx_gather_query (a collective a peer never joins must not hang the run) and rank 0's assembled frame
    against the whole frame rendered on its own GPU --, or torch.distributed.gather (nccl backend = RCCL as well) if that fails, in the
    same process (never re-exec a process that has touched the GPU)."""

    sharded = True

    def __init__(self, args, wl, torch, dist, hip, rank, world_size):
        from voxel_rs_amd.sharding import FrameSharder

        self.args, self.wl, self.svo, self.torch, self.dist, self.hip = args, wl, wl.svo, torch, dist, hip
        self.rank, self.world_size, self.W, self.H = rank, world_size, args.width, args.height
        self.FrameSharder = FrameSharder
        self.vx_fmt = hip.VX_FORMAT_RGBA8 if args.gather_format == "rgba8" else hip.VX_FORMAT_RGBA32F
        # frames in flight: the smaller a rank's share of the frame, the longer its tail relative to its body (a frame cannot
        # finish before its longest ray) and the more frames it takes to keep the device full
        frames = min(8, max(1, args.frames_in_flight)) if args.frames_in_flight else min(8, max(3, world_size))
        # frames per collective (1: every frame is gathered as soon as it is rendered; more: fewer, larger messages). From six ranks on two by
        # default: a rank's share of a 1080p frame is then 0.03-0.04 ms of GPU time, what one exchange costs its host thread (0.03 ms measured on
        # one rank: profiles/round5/pass_a) and its communicator stream -- by analysis only: no run with more than one GPU has been possible
        self.group = max(1, min(args.gather_group if args.gather_group else (2 if world_size >= 6 else 1), frames))
        self.frames = frames - frames % self.group
        # (the library's frame streams -- kernels that can run side by side -- are a choice of their own: a list that is still being exchanged needs a
        # buffer, not a stream)
        self.streams = min(8, max(1, args.streams)) if args.streams else self.frames
        self.svo.set_frames_in_flight(self.streams)
        self.i = 0
        self.view = None
        self.last_view = None
        self.profile_on = False
        self.gather_events = []  # torch path: (start, stop) event pairs around the collective, on torch's stream
        self.gather_used, self.gather_note, self.comm_hung = None, None, False
        self.sharder = None
        self._choose_exchange()

    # -- one frame
    def _render_tiles(self, tiles):
        u = self.view if self.view is not None else self.wl.path[self.i % len(self.wl.path)]
        self.last_view = u
        self.svo.render_device(u, self.W, self.H, tiles.data_ptr(), tile_rank=self.rank, tile_count=self.world_size, fmt=self.vx_fmt)
        self.i += 1

    def step(self):
        self.sharder.step()

    def flush(self):
        self.sharder.flush()

    # -- the two exchanges
    def _build(self, library_gather):
        torch, svo, dist, rank, world_size, args = self.torch, self.svo, self.dist, self.rank, self.world_size, self.args
        FRAMES, GROUP, W, H, vx_fmt = self.frames, self.group, self.W, self.H, self.vx_fmt
        exchange_stream = svo.comm_stream if library_gather else torch.cuda.current_stream().cuda_stream
        exchange_done = [torch.cuda.Event() for _ in range(FRAMES // GROUP)]
        recorded = [False] * (FRAMES // GROUP)
        tickets = {}
        state = {"last_ticket": None}

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

        def gather(tiles, gathered):
            g = sh._g  # (the group being exchanged)
            if library_gather:
                if args.simulate_gather_failure:
                    raise RuntimeError("simulated failure of the library's gather (--simulate-gather-failure)")
                if args.simulate_absent_peer and rank != 0:
                    return  # (this rank never joins the exchange: rank 0's receive waits, and its watchdog has to notice)
                state["last_ticket"] = tickets[g] = svo.gather_tiles(tiles.data_ptr(), tiles.numel() * tiles.element_size(),
                                                                    gathered.data_ptr() if rank == 0 else None, root=0)
            else:
                if self.profile_on:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                if args.dist_backend == "gloo":
                    # (the tests' control plane, several ranks on one GPU: gloo moves host memory)
                    host_tiles = tiles.cpu()
                    parts = [torch.empty_like(host_tiles) for _ in range(world_size)] if rank == 0 else None
                    dist.gather(host_tiles, parts, dst=0)
                    if rank == 0:
                        for r in range(world_size):
                            gathered[r].copy_(parts[r])
                else:
                    dist.gather(tiles, [gathered[r] for r in range(world_size)] if rank == 0 else None, dst=0)
                if self.profile_on:
                    ev[1].record()
                    self.gather_events.append(ev)

        def fused(g, tiles, gathered, image):
            # the library's exchange, frame by frame: one call per frame (vx_render_gather) instead of four
            u = self.view if self.view is not None else self.wl.path[self.i % len(self.wl.path)]
            self.last_view = u
            self.i += 1
            state["last_ticket"] = tickets[g] = svo.render_gather(u, W, H, tiles.data_ptr(), tiles.numel() * tiles.element_size(), gathered.data_ptr() if rank == 0 else None,
                                                                  image.data_ptr() if rank == 0 else None, tickets.get(g, -1), rank, world_size, vx_fmt)

        plain_library = library_gather and not (args.simulate_gather_failure or args.simulate_absent_peer or args.separate_calls)
        sh = self.FrameSharder(W, H, rank, world_size, dist, "cuda", self._render_tiles, assemble, before_render=before_render, after_render=after_render,
                               after_exchange=after_exchange, buffers=FRAMES, group=GROUP, gather=gather, pixel_format=args.gather_format,
                               fused=fused if plain_library else None)
        if not library_gather:
            # (torch's gather wants its own send buffer: the root's list is not rendered in place)
            sh.tiles = [torch.zeros((GROUP, sh.n_max, 32, 32, 4), dtype=sh.dtype, device="cuda") for _ in range(FRAMES // GROUP)]
            torch.cuda.synchronize()
        sh.query = (lambda: svo.gather_query(state["last_ticket"]) if state["last_ticket"] is not None else 1) if library_gather else None
        return sh

    def whole_frame(self, u):
        torch = self.torch
        img = torch.zeros((self.H, self.W, 4), dtype=torch.uint8 if self.vx_fmt == self.hip.VX_FORMAT_RGBA8 else torch.float32, device="cuda")
        torch.cuda.synchronize()
        self.svo.render_device(u, self.W, self.H, img.data_ptr(), fmt=self.vx_fmt)
        self.svo.sync()
        return img

    def frame_is_whole(self):
        """rank 0: what the path delivered last, against the same frame rendered whole on this GPU"""
        sh = self.sharder
        whole = self.whole_frame(self.last_view)
        if self.world_size > 1:
            got = sh.image
        else:
            # one rank (--force-sharded): tile_count 1 means "the whole frame, row-major" to vx_render, so this rank's "tile list" is the
            # frame itself and the assembly kernel (which runs for its cost) scatters something that is no tile list -- the check is
            # on what the gather delivered. (An RGBA8 frame is stored top row first, a tile list bottom row first: same here, the
            # "list" IS the frame.)
            got = sh.last_gathered[0].reshape(-1)[:self.H * self.W * 4].view(self.H, self.W, 4)
        a, b = whole.contiguous().view(self.torch.uint8), got.contiguous().view(self.torch.uint8)
        return bool(self.torch.equal(a, b))

    def _first_frames_ok(self, seconds):
        """GROUP frames through the whole path, watched: the exchange must come back within `seconds` on every rank and rank 0's assembled frame must
        be the whole render's."""
        sh, torch = self.sharder, self.torch
        ok = 1
        try:
            for _ in range(self.group):
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
                self.svo.sync()
                torch.cuda.synchronize()
                if self.rank == 0:
                    ok = 1 if self.frame_is_whole() else 0
        except Exception as e:  # an error code from the library (VX_ERR_HIP with RCCL's message), or the simulated one
            print(f"[bench rank {self.rank}] exchange failed: {e}", file=sys.stderr)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=self.args.ctl_device)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    def _choose_exchange(self):
        args, torch, dist = self.args, self.torch, self.dist
        if args.gather in ("auto", "library"):
            try:
                # the render context owns the RCCL communicator the tiles travel over (vx_comm_init); torch.distributed only carries the
                # id to the ranks and the barriers / statistics of this script
                uid = [self.hip.comm_unique_id() if self.rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                self.svo.comm_init(self.world_size, self.rank, uid[0])
                self.sharder = self._build(True)
                init_ok = 1
            except Exception as e:
                print(f"[bench rank {self.rank}] vx_comm_init failed: {e}", file=sys.stderr)
                init_ok = 0
            flag = torch.tensor([init_ok], dtype=torch.int32, device=args.ctl_device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) and self._first_frames_ok(args.gather_timeout):
                self.gather_used = "library"
                return
            if args.gather == "library":
                raise SystemExit("--gather library: the library's exchange failed its first frames (see stderr); --gather auto falls back to torch.distributed")
            # the other exchange, in this process. The library's communicator is left alone (destroying one with a collective in flight can block).
            self.gather_note = "the library's vx_gather_tiles failed its first frames on this node; fell back in-process"
            self.comm_hung = True
        self.sharder = self._build(False)
        self.gather_used = "torch"
        if not self._first_frames_ok(args.gather_timeout):
            self.gather_note = (self.gather_note + "; " if self.gather_note else "") + "torch.distributed.gather's first frames did not reproduce the whole render"

    def regroup(self, group):
        """the same exchange with `group` frames per gather: new tile lists and tickets (the caller has drained the device)"""
        frames = min(8, max(1, self.args.frames_in_flight)) if self.args.frames_in_flight else min(8, max(3, self.world_size))
        self.group = max(1, min(group, frames))
        self.frames = frames - frames % self.group
        self.sharder = self._build(self.gather_used == "library")
        self.i = 0

    def exchange_profile(self):
        """(ms summed, exchanges) of the timed region: time on the exchange's stream from the first send / receive to the last"""
        if self.gather_used == "library":
            return self.svo.comm_profile_read()
        self.torch.cuda.synchronize()
        out = sum(a.elapsed_time(b) for a, b in self.gather_events), len(self.gather_events)
        self.gather_events.clear()
        return out


# ---- measurements -----------------------------------------------------------------------------------------------------------------


def granted_cpus():
    """What this process may run on: the affinity mask, the cgroup's CPU quota (v2 cpu.max, v1 cfs quota / period), the machine's logical CPUs."""
    out = {"logical_cpus": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None, "cgroup_cpu_max": None}
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        out["cgroup_cpu_max"] = None if quota == "max" else round(float(quota) / float(period), 2)
    except (OSError, ValueError):
        try:
            quota = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
            period = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            out["cgroup_cpu_max"] = round(quota / period, 2) if quota > 0 else None
        except (OSError, ValueError):
            pass
    try:
        out["loadavg_1min"] = round(os.getloadavg()[0], 1)
    except OSError:
        pass
    return out


def cpu_baseline(args, wl, orc):
    """The C oracle on this box's host cores: whole frames of the path's first view (the unit of its OpenMP loop is a 32x32 tile, handed out one
    at a time) at 1, 8, 32 and all threads -- about 3 s each, --cpu-seconds for the last -- so that the line says how the restatement scales on
    what this process is GRANTED. `value` / `cores` = the best point of the sweep and the threads it ran on."""
    W, H = args.width, args.height
    scene = orc.OracleScene(wl.fmt, wl.world.frame(), wl.mats.view(orc.MATERIAL_DTYPE), wl.tex, 6)
    ou = orc.Uniforms.from_buffer_copy(bytes(wl.path[0]))
    granted = granted_cpus()
    omp_threads = orc.lib().or_max_threads()
    every = omp_threads
    quota = int(granted["cgroup_cpu_max"]) if granted.get("cgroup_cpu_max") else 0  # (a thread per granted CPU, where a quota says how many those are)
    points = sorted({t for t in (1, 8, quota, 32, every) if 1 <= t <= every})
    scene.render(ou, W, H, rect=(0, H // 2, W, H // 2 + 32), want_hits=False, counters=orc.Counters(), threads=every)  # (thread pool, page cache)
    sweep = []
    for t in points:
        seconds = args.cpu_seconds if t == every else min(3.0, args.cpu_seconds)
        cc = orc.Counters()
        frames = 0
        t0 = time.perf_counter()
        while frames < 400 and (frames == 0 or time.perf_counter() - t0 < seconds):
            scene.render(ou, W, H, want_hits=False, counters=cc, threads=t)
            frames += 1
        dt = time.perf_counter() - t0
        sweep.append({"threads": t, "value": round(cc.rays / dt / 1e6, 4), "frames": frames, "seconds": round(dt, 2), "rays": int(cc.rays)})
    best = max(sweep, key=lambda r: r["value"])
    one = sweep[0]
    return {"value": best["value"], "unit": "Mrays/s", "cores": best["threads"], "kind": "port",
            "sample": f"{best['frames']} x the whole {W}x{H} frame (the path's first view): {best['rays']} rays in {best['seconds']:.2f} s on {best['threads']} threads "
                      "(C restatement of the GLSL path, OpenMP over 32x32 tiles; the reference has no CPU raycast)",
            "granted": granted, "omp_max_threads": omp_threads, "sweep": sweep, "speedup_best_over_one_thread": round(best["value"] / max(one["value"], 1e-9), 2),
            "single_thread": {"value": one["value"], "unit": "Mrays/s", "cores": 1, "sample": f"{one['frames']} x the whole frame: {one['rays']} rays in {one['seconds']:.2f} s"}}


# ---- the other BASELINE configurations on this GPU -----------------------------------------------------------------------------------


def other_configs(args, vra, hip, scenes, torch, formats, clock=None):
    """C2 (1080p primary rays, depth 10), C4 (4K primary + shadow on the full-detail depth-14 terrain: static, and streamed by the chunk loader),
    C5 (7680x4320 + 2x2 resolve on that terrain: one rank's share of eight, and the whole frame on this one GPU): the same moving camera, frames into
    device memory with the library's two frames in flight, the median of five blocks. Rays and iterations from the instrumented kernel (first view)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("stream_bench", ROOT / "profiles" / "stream_bench.py")
    stream_bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(stream_bench)
    out = {}
    mats, tex = scenes.synthetic_materials(), scenes.asset_textures(ROOT / "tests" / "golden" / "textures")
    stream = torch.cuda.current_stream().cuda_stream

    def timed(svo, frame, steps, repeats=5):
        for i in range(4):
            frame(i)
        svo.sync()
        torch.cuda.synchronize()
        blocks = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            for i in range(steps):
                frame(i)
            svo.sync()
            torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t0) / steps * 1e3)
        return sorted(blocks)[len(blocks) // 2]

    def line(ms, c, extra=None):
        return {"ms_per_step": round(ms, 4), "value": round(c["rays"] / ms / 1e3, 1), "unit": "Mrays/s", "rays_per_step": int(c["rays"]),
                "iterations_per_s": round(c["iterations"] / ms * 1e3, 0), **(extra or {})}

    for fmt_name in formats:
        fmt = vra.SVO_ESVO if fmt_name == "esvo" else vra.SVO_CSVO
        for depth in (10, 14):
            t0 = time.perf_counter()
            world = vra.World(fmt)
            st = world.build_heightfield(depth)
            build_s = time.perf_counter() - t0
            svo = hip.Svo(fmt, world.size_in_bytes + (16 << 20))
            svo.set_materials(mats)
            svo.set_textures(tex, 6)
            t0 = time.perf_counter()
            svo.update(world)
            svo.sync()
            commit_s = time.perf_counter() - t0
            about = {"svo_bytes": world.size_in_bytes, "leaves": st["leaves"], "scene_build_s": round(build_s, 2), "first_commit_s": round(commit_s, 2)}
            if depth == 10:
                W, H = 1920, 1080
                path = [moving_uniforms(scenes, depth, st["h_max"], W, H, i, shadows=False) for i in range(20)]
                imgs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
                torch.cuda.synchronize()
                ms = timed(svo, lambda i: svo.render_device(path[i % 20], W, H, imgs[i % 2].data_ptr()), 200)
                out.setdefault("C2", {})[fmt_name] = line(ms, svo.render_counters(path[0], W, H), {"workload": "1920x1080 primary rays only, depth-10 SVO", **about})
            else:
                w, h = 3840, 2160
                path = [moving_uniforms(scenes, depth, st["h_max"], w, h, i) for i in range(20)]
                imgs = [torch.zeros((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
                torch.cuda.synchronize()
                ms = timed(svo, lambda i: svo.render_device(path[i % 20], w, h, imgs[i % 2].data_ptr()), 40)
                c4 = svo.render_counters(path[0], w, h)
                # the depth-14 frame's own roofline block, as for the headline: the kernel alone (one frame at a time, HIP events around each launch), the
                # algorithmic bytes of the first view, the stored counters of profiles/round6/c4_counters.sh (quoted for these library sources only)
                svo.set_frames_in_flight(1)
                for i in range(4):
                    svo.render_device(path[i % 20], w, h, imgs[0].data_ptr())
                svo.sync()
                svo.profile_enable(True)
                for i in range(12):
                    svo.render_device(path[i % 20], w, h, imgs[0].data_ptr())
                    svo.sync()
                kms, launches = svo.profile_read()
                svo.profile_enable(False)
                svo.set_frames_in_flight(2)
                excl = kms / max(launches, 1)
                alg = algorithmic_bytes(fmt_name, c4)
                stored, src, note = stored_counters(fmt_name, "C4")
                roof = {"bound": "hbm", "achieved": round(alg / (excl * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(alg / (excl * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "traffic": stored.get("bytes_per_launch") if stored else None,
                        "traffic_source": src, **({"traffic_note": note} if note else {}), "kernel": "render_persistent",
                        "kernel_exclusive_ms": round(excl, 4), "algorithmic_bytes_per_launch": int(alg), "bytes_per_ray": round(alg / max(c4["rays"], 1), 2)}
                if stored and clock:
                    issue = issue_model(stored, src, clock["mhz_median"], "this run: " + clock["source"])
                    if issue:
                        issue["frac_of_bound_one_frame_at_a_time"] = round(issue["issue_bound_ms"] / excl, 4)
                        issue["frac_of_bound_timed_mode"] = round(issue["issue_bound_ms"] / ms, 4)
                        roof["issue"] = issue
                        roof.update({"issue_bound_ms": issue["issue_bound_ms"], "issue_frac_timed_mode": issue["frac_of_bound_timed_mode"],
                                     "wait_share_of_wave_cycles": issue["wait_share_of_wave_cycles"]})
                    if stored.get("TCC_HIT") and stored.get("TCC_MISS"):
                        roof["l2_hit_rate"] = round(stored["TCC_HIT"] / (stored["TCC_HIT"] + stored["TCC_MISS"]), 3)
                out.setdefault("C4_static", {})[fmt_name] = line(ms, c4, {"workload": "3840x2160 primary + shadow, static full-detail depth-14 terrain", **about,
                                                                       "image": svo.image_info(), "roofline": roof})
                del imgs
                # C5: 2x2 supersamples of the 4K frame. One rank's share of eight (its tile list; the resolve of the assembled frame is rank 0's), and
                # the whole supersampled frame + the resolve on this one GPU
                W, H = 2 * w, 2 * h
                u5 = moving_uniforms(scenes, depth, st["h_max"], W, H, 0)
                per = hip.local_tile_count(W, H, 0, 8)
                lists = [torch.zeros((per, 32, 32, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
                torch.cuda.synchronize()
                ms = timed(svo, lambda i: svo.render_device(u5, W, H, lists[i % 2].data_ptr(), tile_rank=0, tile_count=8), 40)
                out.setdefault("C5_rank_share", {})[fmt_name] = line(ms, svo.render_counters(u5, W, H, 0, 8),
                                                                     {"workload": "rank 0's tiles (one in eight, Morton round-robin) of the 7680x4320 supersampled frame, depth-14 terrain"})
                del lists
                big = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
                small = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
                torch.cuda.synchronize()

                def whole(i):
                    svo.render_device(u5, W, H, big[i % 2].data_ptr())
                    svo.stream_wait_render(stream)
                    svo.resolve_2x2(big[i % 2].data_ptr(), w, h, small.data_ptr(), stream=stream)

                ms = timed(svo, whole, 12, 3)
                out.setdefault("C5_whole_on_one_gpu", {})[fmt_name] = line(ms, svo.render_counters(u5, W, H), {"workload": "7680x4320 + 2x2 resolve to 3840x2160, depth-14 terrain, this GPU alone"})
                del big, small
                try:  # every rank's share of the supersampled frame, rendered here in turn (a projection: scale_model)
                    out.setdefault("C5_scale_model", {})[fmt_name] = scale_model(svo, hip, torch, [u5], W, H, ms, frames_per_share=10)
                except Exception as e:
                    out.setdefault("C5_scale_model", {})[fmt_name] = {"error": f"{type(e).__name__}: {e}"[:200]}
            svo.close()
            del world, svo
        # C4 as the game produces it: the chunk loader streams the depth-14 terrain in around the camera (radius 40, <= 400 events a commit,
        # pipelined commits) while frames render: the kernel in mid-stream and settled, and what the frame loop's thread pays per step
        r = stream_bench.run(stream_bench.parse_args(["--format", fmt_name, "--scene-depth", "14", "--radius", "40", "--width", "3840", "--height", "2160", "--frames", "80"]))
        out.setdefault("C4_streamed", {})[fmt_name] = {
            "ms_per_step": r["kernel_ms_streaming_median"], "value": r["Mrays_per_s_settled"], "unit": "Mrays/s", "rays_per_step": r["rays_last_frame"],
            "iterations_per_s": round(r["iterations_last_frame"] / max(r["kernel_ms_settled_median"], 1e-9) * 1e3, 0),
            "ms_per_step_settled": r["kernel_ms_settled_median"], "host_ms_per_step_median": r["host_ms_per_step_median"], "host_ms_per_step_max": r["host_ms_per_step_max"],
            "apply_ms_median": r["apply_ms_median"], "commit_ms_median": r["commit_ms_median"], "resident_chunks": r["resident_chunks"],
            "initial_fill_s": r["initial_fill"]["seconds"], "workload": r["workload"]}
    return out


def scale_model(svo, hip, torch, views, W, H, whole_ms, frames_per_share=30):
    """A PROJECTION, not a measurement, of the N-GPU frame from this one GPU: every one of the N tile shares (N = 2, 4, 8: the tiles in Morton order, round-robin)
    of the frame rendered here as the rank that owns it would render it (its RGBA8 tile list, max(3, N) frames in flight, the camera moving), rank 0's assembly
    of N gathered lists timed on its own, and the link time of a share into rank 0 at 77 GB/s per xGMI direction. projected frame = max(the dearest other
    rank's share, rank 0's share + its assembly, the link time); projected_speedup = this GPU's whole frame / that. It says how EVEN the shares are
    (share_ms_max against share_ms_mean) and what a perfect overlap would give; no N > 1 run stands behind it."""
    out = {}
    stream = torch.cuda.current_stream().cuda_stream

    def timed(frame, steps):
        for i in range(4):
            frame(i)
        svo.sync()
        torch.cuda.synchronize()
        blocks, issue = [], []
        for _ in range(3):
            t0 = time.perf_counter()
            for i in range(steps):
                frame(i)
            issue.append((time.perf_counter() - t0) / steps * 1e3)
            svo.sync()
            torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t0) / steps * 1e3)
        return sorted(blocks)[1], sorted(issue)[1]

    for n in (2, 4, 8):
        svo.set_frames_in_flight(min(8, max(3, n)))
        per = max(hip.local_tile_count(W, H, r, n) for r in range(n))
        lists = [torch.zeros((per, 32, 32, 4), dtype=torch.uint8, device="cuda") for _ in range(min(8, max(3, n)))]
        torch.cuda.synchronize()
        shares, issues = [], []
        for r in range(n):
            ms, issue = timed(lambda i, r=r: svo.render_device(views[i % len(views)], W, H, lists[i % len(lists)].data_ptr(), tile_rank=r, tile_count=n, fmt=hip.VX_FORMAT_RGBA8),
                              frames_per_share)
            shares.append(ms)
            issues.append(issue)
        gathered = torch.zeros((n, per, 32, 32, 4), dtype=torch.uint8, device="cuda")
        image = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        asm, _ = timed(lambda i: svo.assemble_tiles_format(gathered.data_ptr(), gathered.stride(0) // 4, n, W, H, image.data_ptr(), hip.VX_FORMAT_RGBA8, stream), frames_per_share)
        link_ms = per * 1024 * 4 / 77e9 * 1e3
        frame_ms = max(max(shares[1:]), shares[0] + asm, link_ms)
        out[str(n)] = {"share_ms_max": round(max(shares), 4), "share_ms_mean": round(sum(shares) / n, 4), "share_ms_min": round(min(shares), 4),
                       "host_issue_ms": round(max(issues), 4), "assembly_ms": round(asm, 4), "link_ms": round(link_ms, 4),
                       "projected_frame_ms": round(frame_ms, 4), "projected_speedup": round(whole_ms / frame_ms, 2)}
        del lists, gathered, image
    svo.set_frames_in_flight(2)
    out["whole_frame_ms_on_this_gpu"] = round(whole_ms, 4)
    out["note"] = "a projection from one GPU (every rank's share rendered here in turn); no scaling curve was measured"
    return out


def forced_sharded_child(args):
    """The N > 1 code path (tile lists, the library's RCCL gather on a one-rank communicator, rank 0's assembly) on this one GPU, in a child process of
    its own (a fresh context and process group; this process has finished its GPU work and waits): `bench.py --force-sharded` at the same sizes."""
    import subprocess

    cmd = [sys.executable, str(Path(__file__).resolve()), "--force-sharded", "--no-cpu-baseline", "--no-extras", "--sustained-seconds", "0", "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--repeats", "9", "--depth", str(args.depth), "--width", str(args.width), "--height", str(args.height), "--format", args.format,
           "--textures", args.textures]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        d = json.loads(line)
        return {"value": d["value"], "unit": "Mrays/s", "ms_per_step": d["ms_per_step"], "identical": d["config"].get("sharded_frame_identical_to_whole_render"),
                "gather": d["config"].get("gather"), "gather_format": d["config"].get("gather_format"), "frames_in_flight": d["roofline"].get("frames_in_flight"),
                "host_issue_ms_per_step": d["config"].get("host_issue_ms_per_step")}
    except Exception as e:  # the record says so instead of failing the headline
        return {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}


def picker_latency(wl, hip, np):
    """vx_raycast is synchronous per call, like the reference's fence wait (svo.rs:248-249), and the game's physics calls it at 250 Hz
    (src/gamelogic/game.rs:90,130-134): a batch of 80 tasks (the reference's own: one entity's AABB fan, svo_picker.rs:311-536) and C1's 65,536."""
    out = {}
    n = float(1 << wl.args.depth)
    rng = np.random.default_rng(1)
    for name, count in (("reference_batch_80", 80), ("c1_65536", 65536)):
        tasks = np.zeros(count, dtype=hip.PICKER_TASK_DTYPE)
        tasks["pos"] = (np.float32([0.5 * n, wl.st["h_max"] + 4.0, 0.5 * n]) + rng.uniform(-8, 8, size=(count, 3))).astype(np.float32)
        d = rng.normal(size=(count, 3))
        tasks["dir"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        tasks["max_dst"] = 32.0
        for _ in range(5):
            wl.svo.raycast(tasks)
        reps = 200 if count < 1000 else 20
        t0 = time.perf_counter()
        for _ in range(reps):
            wl.svo.raycast(tasks)
        us = (time.perf_counter() - t0) / reps * 1e6
        out[name] = {"rays": count, "us_per_call": round(us, 2), "Mrays_s": round(count / us, 3)}
    return out


def parse_args(argv=None):
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
    ap.add_argument("--gather-group", type=int, default=0, help="sharded: frames per gather (default 1; 2 from six ranks on)")
    ap.add_argument("--no-calibration", action="store_true",
                    help="sharded, more than one rank: skip the self-calibration (two warm-up blocks at comm headroom {0, 2, 4} x frames per gather {1, 2}, the best kept "
                         "on every rank; --gather-group pins the frames per gather)")
    ap.add_argument("--streams", type=int, default=0, help="sharded: the library's frame streams (default: as many as frames in flight, i.e. tile-list buffers)")
    ap.add_argument("--gather", choices=["auto", "library", "torch"], default="auto",
                    help="sharded: the exchange step -- library: vx_gather_tiles (RCCL send/receive owned by the render context); torch: torch.distributed.gather; "
                         "auto: the library's, checked on its first frames (watchdog + the assembled frame against the whole render), torch's if that fails")
    ap.add_argument("--gather-format", choices=["rgba8", "rgba32f"], default="rgba8",
                    help="sharded: pixel format of the tile lists that travel and of rank 0's image (rgba8 = Framebuffer::as_image's bytes: a quarter of the link time)")
    ap.add_argument("--gather-timeout", type=float, default=30.0, help="sharded: seconds the first exchange may take before it is declared hung")
    ap.add_argument("--simulate-gather-failure", action="store_true", help="testing: make the library's exchange fail, to exercise the fall-back")
    ap.add_argument("--separate-calls", action="store_true", help="sharded, the library's exchange: wait / render / gather / assemble as four calls per frame instead of vx_render_gather's one (A/B)")
    ap.add_argument("--simulate-absent-peer", action="store_true", help="testing: the ranks other than 0 never join the library's exchange (the watchdog must end the wait)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed's backend for this script's barriers and statistics (gloo: the tests, where several ranks share one GPU and RCCL proper cannot run)")
    ap.add_argument("--comm-library", default="", help="the RCCL build the render context opens (vx_comm_library); the tests name their stand-in (tests/stub_rccl)")
    ap.add_argument("--sustained-seconds", type=float, default=3.0,
                    help="after the median-of-blocks headline: one block of at least this many seconds of the same path (0 = none); it becomes `value` if more than 2 %% below the burst median")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` object (C2, C4 static / streamed, C5 on this GPU: about 40 s, most of it the depth-14 terrain's build)")
    ap.add_argument("--config-formats", default="csvo,esvo", help="node formats the `configs` object is measured for")
    ap.add_argument("--no-forced-sharded", action="store_true", help="skip the `forced_sharded` object (the N > 1 code path on this one GPU, a child process)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary blocks (still view, shadow_distance 500, picker, configs, forced_sharded): profiler runs, whose per-kernel averages they would dilute")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the N > 1 code path (tile lists, RCCL gather, assembly) even with one rank; needs a torch.distributed.run launch")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="wall-clock target for the cpu_baseline sample on all threads (the sweep's other points: 3 s each)")
    ap.add_argument("--textures", choices=["assets", "procedural"], default="assets",
                    help="assets: the reference's own 64x64 textures (tests/golden/textures) for the blocks the terrain uses")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    # `forced_sharded` first, while this process holds nothing of the GPU: measured after its own work -- with the parent's idle context still
    # mapped (a dozen streams = hardware queues) -- the child's frame streams were time-sliced and the figure was 7-10 % below what the same
    # command does on its own.
    args.forced_sharded_result = None
    if (int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.force_sharded and not args.no_extras and not args.no_forced_sharded):
        args.forced_sharded_result = forced_sharded_child(args)
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
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo")
    args.ctl_device = "cuda" if args.dist_backend == "nccl" else "cpu"
    if args.comm_library:
        hip.comm_library(args.comm_library)

    W, H = args.width, args.height
    wl = Workload(args, vra, hip, scenes, rank, world_size, local_rank)
    svo = wl.svo
    run = Sharded(args, wl, torch, dist, hip, rank, world_size) if sharded else SingleGpu(args, wl, torch)

    def barrier():
        run.flush()  # (sharded: a group of frames that has not been exchanged yet)
        svo.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_blocks(repeats):
        """--steps frames between two barriers, `repeats` times back to back: block times and host issue times"""
        blocks, enqueue = [], []
        for _ in range(max(repeats, 1)):
            barrier()
            run.i = 0  # every block walks the path from its first view
            t0 = time.perf_counter()
            for _ in range(args.steps):
                run.step()
            enqueue.append(time.perf_counter() - t0)  # host time to issue the steps (the loop is asynchronous)
            barrier()
            blocks.append(time.perf_counter() - t0)
        return blocks, enqueue

    for _ in range(args.warmup):
        run.step()
    barrier()
    # Self-calibration (more than one rank): what RCCL's kernels need beside persistent render waves (the wave slots per CU a context leaves them:
    # vx_set_comm_headroom) and whether two frames per gather pay could only be guessed without a multi-GPU run, so the first such run measures them:
    # two untimed blocks at every setting, the block times reduced to the slowest rank's, the best setting kept by every rank. Outside the timed region.
    calibration = None
    if sharded and world_size > 1 and not args.no_calibration:
        trials = []
        groups = [args.gather_group] if args.gather_group else [1, 2]
        for group in groups:
            barrier()
            run.regroup(group)
            for headroom in (0, 2, 4):
                svo.set_comm_headroom(headroom)
                for _ in range(max(run.frames, 4)):  # (the first frames of a setting: new buffers, another grid)
                    run.step()
                b, _ = timed_blocks(2)
                t = torch.tensor([min(b)], dtype=torch.float64, device=args.ctl_device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                trials.append({"comm_headroom": headroom, "frames_per_gather": run.group, "ms_per_step": round(float(t[0]) / args.steps * 1e3, 4)})
        best = min(trials, key=lambda r: r["ms_per_step"])  # (the reduced times are the same on every rank: so is the choice)
        barrier()
        run.regroup(best["frames_per_gather"])
        svo.set_comm_headroom(best["comm_headroom"])
        for _ in range(args.warmup):
            run.step()
        barrier()
        calibration = {"chosen": best, "trials": trials, "note": "two untimed blocks per setting, slowest rank's time; outside the timed region"}
    # The timed block -- exactly --steps frames between two barriers -- is run --repeats times back to back and the MEDIAN block is
    # what is reported: 50 frames are 25 ms of GPU time, too little for one sample to stand on.
    svo.profile_enable(True)
    run.profile_on = True
    blocks, enqueue = timed_blocks(args.repeats)
    kernel_ms, launches = svo.profile_read()
    gather_ms, gathers = run.exchange_profile() if sharded else (0.0, 0)
    run.profile_on = False
    svo.profile_enable(False)
    enqueue_s = sorted(enqueue)[len(enqueue) // 2]

    # The sustained block: the same path, again and again, for --sustained-seconds without a pause -- what the device holds over seconds (clock,
    # power, temperature), which a 5 ms block cannot say. The host stays a few hundred frames ahead at most: every 100 frames an event is
    # recorded behind the newest render (vx_stream_wait_render on torch's stream), and the host waits for the event four marks back and notes
    # when it passed -- frames completed against time, from which the first and the last second's rates are read. Meanwhile a second thread
    # samples the shader clock (vx_clock_probe: a one-lane kernel beside the render kernels).
    sustained, clock = None, None
    if args.sustained_seconds > 0:
        import threading

        probes, stop = [], threading.Event()

        def sample_clock():
            while not stop.is_set():
                try:
                    probes.append(svo.clock_probe(200))
                except Exception:  # (the record then carries no clock)
                    return
                stop.wait(0.02)

        ts = torch.cuda.current_stream()
        MARK = 100
        # (how many frames: fixed before the block from the burst rate, the same number on every rank -- a sharded frame is a collective)
        burst = torch.tensor([sorted(blocks)[len(blocks) // 2] / args.steps], dtype=torch.float64, device=args.ctl_device)
        if dist is not None:
            dist.all_reduce(burst, op=dist.ReduceOp.MAX)
        frames_target = max(MARK, int(math.ceil(args.sustained_seconds * 1.02 / max(float(burst[0]), 1e-6) / MARK)) * MARK)
        barrier()
        run.i = 0
        sampler = threading.Thread(target=sample_clock, daemon=True)
        sampler.start()
        marks, passed, frames_issued = [], [], 0
        t0 = time.perf_counter()
        while frames_issued < frames_target:
            for _ in range(MARK):
                run.step()
            frames_issued += MARK
            svo.stream_wait_render(ts.cuda_stream)
            ev = torch.cuda.Event()
            ev.record(ts)
            marks.append((frames_issued, ev))
            while len(marks) > 4:
                f, e = marks.pop(0)
                e.synchronize()
                passed.append((f, time.perf_counter() - t0))
        for f, e in marks:
            e.synchronize()
            passed.append((f, time.perf_counter() - t0))
        barrier()
        total_s = time.perf_counter() - t0
        stop.set()
        sampler.join(timeout=5.0)
        if dist is not None:  # (the slowest rank's time stands)
            slowest = torch.tensor([total_s], dtype=torch.float64, device=args.ctl_device)
            dist.all_reduce(slowest, op=dist.ReduceOp.MAX)
            total_s = float(slowest[0])

        def rate_between(a, b):  # frames per second between two moments of the block, from the marks
            inside = [(f, t) for f, t in passed if a <= t <= b]
            return (inside[-1][0] - inside[0][0]) / (inside[-1][1] - inside[0][1]) if len(inside) >= 2 and inside[-1][1] > inside[0][1] else None

        sustained = {"seconds": round(total_s, 3), "frames": frames_issued, "ms_per_step": round(total_s / frames_issued * 1e3, 4),
                     "frames_per_s_first_second": rate_between(0.0, 1.0), "frames_per_s_last_second": rate_between(total_s - 1.0, total_s)}
        if probes:
            ps = sorted(probes)
            clock = {"mhz_median": round(ps[len(ps) // 2], 1), "mhz_min": round(ps[0], 1), "mhz_max": round(ps[-1], 1), "samples": len(ps),
                     "source": "vx_clock_probe (s_memtime against the 100 MHz counter over 200 us, a wave of its own) sampled every 20 ms while the sustained block's kernels run"}

    # One frame at a time (outside the timed region): with several frames in flight a kernel's HIP-event span includes the time it
    # shares the device with its neighbours, so the kernel's OWN duration -- what the roofline fraction is defined on -- is measured
    # with the device to itself, over the path's views, twice:
    #  (a) the timed region's own launch policy: the frame streams, every frame waited for before the next is issued;
    #  (b) the library's one-frame-at-a-time mode (set_frames_in_flight(1): the context's own stream): what a consumer that presents
    #      every frame runs, and the same command rocprofv3 is run on for profiles/ (--frames-in-flight 1).
    def exclusive(n):
        """(the kernel's own event span, the host's wall clock) per frame, every frame waited for before the next is issued -- the wall clock is what
        a consumer that presents every frame sees (the reference's loop: svo.rs:220-228), order kernel, launch and wait included"""
        barrier()
        run.i = 0
        svo.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(n):
            run.step()
            run.flush()
            svo.sync()
        wall_ms = (time.perf_counter() - t0) / n * 1e3
        barrier()
        ms, k = svo.profile_read()
        if sharded and run.gather_used == "library":
            svo.comm_profile_read()
        svo.profile_enable(False)
        return ms / max(k, 1), wall_ms

    kernel_exclusive_frame_stream_ms, one_frame_wall_ms = exclusive(min(args.steps, 25))
    kernel_exclusive_ms = kernel_exclusive_frame_stream_ms
    if not sharded:
        svo.set_frames_in_flight(1)
        for _ in range(4):
            run.step()
        kernel_exclusive_ms, one_frame_wall_ms = exclusive(min(args.steps, 25))
        svo.set_frames_in_flight(run.frames)

    # Secondary blocks (one GPU): the camera standing still (round 3's headline: the same view again and again), the game's own shadow
    # cut-off (500 blocks, src/gamelogic/world.rs:105-108: from this altitude few or no hits are that near, so it is close to a
    # primary-rays-only frame; SURVEY.md 8d asks for both), the picker's latency.
    still, sd500, picker = None, None, None
    if not sharded and not args.no_extras:
        def view_block(u, repeats=9):
            rays = svo.render_counters(u, W, H, 0, 1)["rays"]
            run.view = u
            for _ in range(args.warmup):
                run.step()
            b, _ = timed_blocks(repeats)
            run.view = None
            ms = sorted(b)[len(b) // 2] / args.steps * 1e3
            return {"rays_per_frame": int(rays), "ms_per_step": round(ms, 4), "value": round(rays / (ms * 1e-3) / 1e6, 3), "unit": "Mrays/s"}

        still = view_block(wl.still)
        still["note"] = "the §8d view, again and again (round 3's headline; its sorted passes -- a still-view-only gain -- are gone: profiles/round4/pass_k)"
        sd500 = view_block(scenes.bench_camera(args.depth, wl.st["h_max"], W, H, shadow_distance=500.0, render_shadows=True), 5)
        sd500.update({"shadow_distance": 500.0, "note": "the timed frames cast a shadow ray from every primary hit (shadow_distance = inf); this is the game's default cut-off, a still view"})
        picker = picker_latency(wl, hip, np)
    scale = None
    if not sharded and not args.no_extras and rank == 0:
        try:
            scale = scale_model(svo, hip, torch, wl.path, W, H, sorted(blocks)[len(blocks) // 2] / args.steps * 1e3)
        except Exception as e:
            scale = {"error": f"{type(e).__name__}: {e}"[:200]}
        svo.set_frames_in_flight(run.frames)
    configs, forced = None, None
    if not sharded and not args.no_extras and rank == 0:
        if not args.no_configs:
            try:
                configs = other_configs(args, vra, hip, scenes, torch, [f for f in args.config_formats.split(",") if f in ("csvo", "esvo")], clock=clock)
            except Exception as e:  # (a box without the memory for the depth-14 terrain: the headline still stands)
                configs = {"error": f"{type(e).__name__}: {e}"[:300]}
        forced = args.forced_sharded_result  # (measured before this process touched the GPU: see main())

    times = torch.tensor(blocks, dtype=torch.float64, device=args.ctl_device)
    stats = torch.tensor([float(wl.rays_per_block), float(wl.bytes_per_frame), kernel_ms / max(launches, 1), kernel_exclusive_ms, gather_ms / max(gathers, 1),
                          float(wl.iterations_per_block)],
                         dtype=torch.float64, device=args.ctl_device)
    per_rank = None
    if dist is not None:
        dist.all_reduce(times, op=dist.ReduceOp.MAX)  # per block: the slowest rank
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        total_rays_per_block = float(sm[0])
        total_iterations_per_block = float(sm[5])
        every = [torch.zeros_like(stats) for _ in range(world_size)]
        dist.all_gather(every, stats)
        per_rank = [{"rank": r, "rays_per_block": int(e[0]), "kernel_span_ms_in_flight": round(float(e[2]), 4), "kernel_exclusive_ms": round(float(e[3]), 4),
                     "exchange_ms": round(float(e[4]), 4)} for r, e in enumerate(every)]
    else:
        total_rays_per_block = float(wl.rays_per_block)
        total_iterations_per_block = float(wl.iterations_per_block)
    block_s = sorted(float(t) for t in times)
    elapsed = block_s[len(block_s) // 2]
    # sharded: the frame rank 0 assembled last against the same frame rendered whole on this GPU (outside the timed region)
    sharded_frame_identical = run.frame_is_whole() if sharded and rank == 0 else None
    if rank != 0:
        if sharded and run.gather_used == "library":
            svo.comm_destroy()
        if dist is not None:
            dist.destroy_process_group()
        if sharded and run.comm_hung:
            os._exit(0)  # (a communicator with a collective that never completed: its teardown can block)
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = total_rays_per_block / elapsed / 1e6  # Mrays/s, whole job
    timed_region_s = sum(block_s) + (sustained["seconds"] if sustained else 0.0)  # every timed block and the sustained one: what `value` stands on
    burst = {"value": round(value, 3), "ms_per_step": round(ms_per_step, 4)}
    value_source = f"median of {len(block_s)} timed blocks of {args.steps} frames"
    if sustained:
        # the same path's views in the same order: rays per frame as in the blocks. The sustained figure is the headline if it is more than 2 % lower.
        rays_per_frame_all = total_rays_per_block / args.steps
        sustained["value"] = round(rays_per_frame_all / (sustained["ms_per_step"] * 1e-3) / 1e6, 3)
        for k in ("first", "last"):
            fps = sustained.pop(f"frames_per_s_{k}_second")
            sustained[f"block_{k}_s_value"] = round(rays_per_frame_all * fps / 1e6, 3) if fps else None
        sustained["vs_burst_median"] = round(sustained["value"] / value, 4)
        if sustained["value"] < 0.98 * value:
            value, ms_per_step = sustained["value"], sustained["ms_per_step"]
            value_source = f"the sustained block ({sustained['seconds']} s, {sustained['frames']} frames): more than 2 % below the burst median, so it is the headline"
    kernel_avg_ms = kernel_ms / max(launches, 1)
    my_bytes = wl.bytes_per_frame
    achieved = my_bytes / (kernel_exclusive_ms * 1e-3) / 1e9 if kernel_exclusive_ms > 0 else 0.0
    reference_shape = (W, H, args.depth, world_size) == (1920, 1080, 12, 1)
    stored, traffic_source, stored_note = stored_counters(args.format) if reference_shape else (None, None, "not the reference shape (1920x1080, depth 12, one GPU)")
    traffic = stored.get("bytes_per_launch") if stored else None
    roofline = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                "traffic": traffic, "traffic_source": traffic_source, **({"traffic_note": stored_note} if stored_note else {}), "kernel": "render_persistent",
                # achieved = algorithmic bytes per launch / the kernel's own duration: one frame at a time on one stream, HIP events bracketing
                # each launch (what rocprofv3's kernel duration is 6-8 % below: the bracket includes the launch's own start and end)
                "kernel_exclusive_ms": round(kernel_exclusive_ms, 4), "kernel_exclusive_ms_is": "HIP-event bracket around each launch",
                # per-launch event span inside the timed region: with frames in flight the spans overlap (span x launches > elapsed)
                "kernel_span_ms_in_flight": round(kernel_avg_ms, 4), "launches": launches,
                "algorithmic_bytes_per_launch": int(my_bytes), "bytes_per_ray": round(my_bytes * args.steps / max(wl.rays_per_block, 1), 2),
                "byte_model": "the reference's own fetches (SURVEY.md 8d), counted by the instrumented kernel on the world's own bytes, mean of the path's views",
                "image_model_bytes_per_launch": int(wl.image_bytes_per_frame),
                "frames_in_flight": run.frames, **({"frames_per_gather": run.group} if sharded else {}),
                "kernel_exclusive_mode": ("one frame at a time on the context's own stream (set_frames_in_flight(1))"
                                          if not sharded else "every frame waited for before the next is issued; the timed region's streams and launch policy"),
                # the same with the timed region's own launch policy (frame streams), every frame waited for
                "kernel_exclusive_ms_timed_policy": round(kernel_exclusive_frame_stream_ms, 4),
                "timed_region_mode": f"{run.frames} frames in flight (ms_per_step); kernel_span_ms_in_flight is a launch's own event span there",
                "temporal_reuse": "none: every frame is a new view, and nothing a frame leaves behind is used by the next",
                # what the device sustains over the median timed block: bytes x frames / elapsed
                "sustained_GBps": round(my_bytes * args.steps / max(elapsed, 1e-9) / 1e9, 3)}
    # The HBM byte model is what the contract asks for, but this kernel's working set is cache resident and it is bound by instruction
    # issue: the second, practical bound, with the fraction of it the kernel reaches one frame at a time and in the timed mode.
    issue = issue_model(stored, traffic_source, clock["mhz_median"], "this run: " + clock["source"]) if (stored and clock) else None
    if clock:
        roofline["clock"] = clock
    if issue:
        issue["frac_of_bound_one_frame_at_a_time"] = round(issue["issue_bound_ms"] / kernel_exclusive_ms, 4) if kernel_exclusive_ms > 0 else None
        # (against the ms_per_step this line reports: the sustained block's where that is the headline)
        issue["frac_of_bound_timed_mode"] = round(issue["issue_bound_ms"] / ms_per_step, 4)
        roofline["issue"] = issue
    elif reference_shape:
        roofline["issue_note"] = (stored_note if not stored else
                                  "no shader-clock sample in this run (--sustained-seconds 0: the clock is probed while the sustained block renders): the issue bound is not quoted")
    # the same as flat scalars (a record that keeps only the scalars of `roofline` still carries them)
    roofline.update({"clock_mhz_median": clock["mhz_median"] if clock else None, "issue_bound_ms": issue["issue_bound_ms"] if issue else None,
                     "issue_frac_one_frame_at_a_time": issue["frac_of_bound_one_frame_at_a_time"] if issue else None,
                     "issue_frac_timed_mode": issue["frac_of_bound_timed_mode"] if issue else None,
                     "instructions_per_launch": issue["instructions_per_launch"] if issue else None,
                     "wait_share_of_wave_cycles": issue["wait_share_of_wave_cycles"] if issue else None,
                     "valu_lane_utilisation": issue["valu_lane_utilisation"] if issue else None})

    cpu = None
    if not args.no_cpu_baseline and world_size == 1:  # rank 0 at N = 1 only: a baseline of the workload, not of the scaling run
        from oracle import oracle as orc  # the checker, timed here as the CPU baseline ("port": the reference has no CPU raycast)

        cpu = cpu_baseline(args, wl, orc)

    rays_per_frame = total_rays_per_block / args.steps
    if cpu:  # flat scalars beside the nested sweep
        cpu["granted_cpus"] = cpu["granted"].get("cgroup_cpu_max") or cpu["granted"].get("affinity")
        cpu["one_thread_value"] = cpu["single_thread"]["value"]
    # the other configurations' frames as flat scalars of `config` (the nested `configs` object holds the whole lines)
    flat_configs = {}
    if configs and "error" not in configs:
        for name, short in (("C2", "c2"), ("C4_static", "c4_static"), ("C4_streamed", "c4_streamed"), ("C5_rank_share", "c5_rank_share"), ("C5_whole_on_one_gpu", "c5_whole")):
            for f, r in (configs.get(name) or {}).items():
                flat_configs[f"{short}_{f}_ms"] = r.get("ms_per_step")
                if name == "C4_static" and isinstance(r.get("roofline"), dict):
                    flat_configs[f"c4_static_{f}_roofline_frac"] = r["roofline"].get("frac")
                    flat_configs[f"c4_static_{f}_kernel_exclusive_ms"] = r["roofline"].get("kernel_exclusive_ms")
                    flat_configs[f"c4_static_{f}_wait_share"] = r["roofline"].get("wait_share_of_wave_cycles")
    out = {
        "metric": "Mrays/sec (primary+shadow) at 1920x1080, depth-12 SVO; achieved HBM GB/s",
        "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
        "repeats": len(block_s), "block_ms_min_median_max": [round(block_s[0] * 1e3, 3), round(elapsed * 1e3, 3), round(block_s[-1] * 1e3, 3)],
        "ms_per_step": round(ms_per_step, 4), "value_source": value_source, "burst": burst, **({"sustained": sustained} if sustained else {}),
        # a scene-independent rate: iterations of the traversal loop (the instrumented kernel's count for this path's views) per second
        "iterations_per_s": round(total_iterations_per_block / args.steps / (ms_per_step * 1e-3), 0),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"C3: {W}x{H} primary + 1 shadow ray per lit pixel, textured + normal-mapped shading, depth-{args.depth} SVO "
                               f"({args.format.upper()} nodes), 1 frame per step, the camera moves every frame (0.25 degrees, 0.083 blocks)",
                   "svo_format": args.format, "svo_bytes": wl.world.size_in_bytes,
                   "leaves": wl.st["leaves"], "chunks": wl.st["chunks"], "textures": args.textures, "rays_per_frame": int(rays_per_frame), "primary_rays": W * H,
                   "camera": "a new view every frame: the §8d view turned by 0.25 degrees per frame about the vertical and walked 0.083 blocks per frame (5 blocks a second at 60 Hz)",
                   "parallelism": f"screen tiles (32x32, Morton order, round-robin) over {world_size} GPU(s), SVO replicated, RCCL gather to rank 0",
                   **({"rccl_ranks": world_size, "gather": "vx_gather_tiles (grouped ncclSend/ncclRecv on the render context's own communicator)" if run.gather_used == "library"
                       else "torch.distributed.gather (nccl backend)", "gather_requested": args.gather, **({"gather_note": run.gather_note} if run.gather_note else {}),
                       "gather_format": args.gather_format, "bytes_gathered_per_frame": int((world_size - 1) * run.sharder.n_max * 1024 * (4 if args.gather_format == "rgba8" else 16)),
                       "exchange_ms_per_gather_rank0": round(gather_ms / max(gathers, 1), 4), "per_rank": per_rank,
                       "sharded_frame_identical_to_whole_render": sharded_frame_identical,
                       **({"calibration": calibration, "comm_headroom": calibration["chosen"]["comm_headroom"]} if calibration else {})} if sharded else {}),
                   "scene_build_s": round(wl.build_s, 2), "upload_s": round(wl.upload_s, 3),
                   "host_issue_ms_per_step": round(enqueue_s / args.steps * 1e3, 4),
                   # flat copies of what the nested objects say (`sustained`, `burst`, `configs`, `forced_sharded`)
                   "burst_value": burst["value"], "burst_ms_per_step": burst["ms_per_step"],
                   "sustained_value": sustained["value"] if sustained else None, "sustained_ms_per_step": sustained["ms_per_step"] if sustained else None,
                   "sustained_seconds": sustained["seconds"] if sustained else None, "sustained_frames": sustained["frames"] if sustained else None,
                   # every frame waited for before the next is issued, the host's wall clock per frame (kernel_exclusive_ms is the kernel's event span there)
                   "one_frame_at_a_time_wall_ms": round(one_frame_wall_ms, 4),
                   "forced_sharded_ms": forced.get("ms_per_step") if forced else None,
                   "forced_sharded_vs_plain": round(forced["value"] / burst["value"], 4) if forced and forced.get("value") else None,
                   **flat_configs,
                   **({f"projected_speedup_{n}_gpus": scale[n]["projected_speedup"] for n in ("2", "4", "8")} if scale and "error" not in scale else {}),
                   **({"share_ms_max_over_mean_8_gpus": round(scale["8"]["share_ms_max"] / max(scale["8"]["share_ms_mean"], 1e-9), 3)} if scale and "error" not in scale else {}),
                   "timed_region_s": round(timed_region_s, 3)},
        "timed_region_s": round(timed_region_s, 3),
        "roofline": roofline, "cpu_baseline": cpu, **({"still_view": still} if still else {}), **({"shadow_distance_500": sd500} if sd500 else {}),
        **({"picker": picker} if picker else {}), **({"scale_model": scale} if scale else {}), **({"configs": configs} if configs else {}),
        **({"forced_sharded": forced} if forced else {}),
    }
    if forced and forced.get("value"):
        forced["vs_plain_burst"] = round(forced["value"] / burst["value"], 4)
    # the JSON line is the LAST thing on stdout: tear the communicators down first (RCCL prints a banner through C stdio, which
    # is flushed at exit otherwise) and flush C's buffers before Python's
    if sharded and run.gather_used == "library":
        svo.comm_destroy()
    if dist is not None:
        dist.destroy_process_group()
    import ctypes

    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    print(json.dumps(out), flush=True)
    if sharded and run.comm_hung:
        os._exit(0)


if __name__ == "__main__":
    main()
