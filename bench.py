#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the SDF ray-marching hot path on MI355X.

A step = one frame of the BASELINE metric workload as the reference's RayMarchingCallback issues it
(renderer.rs:196-255): prepare (uniform write + command-buffer rewrite) and paint (the draw) --
1920x1080, 32-node graph (G32), 256 max steps, still orbit camera of SURVEY 8(d).  The image stays
resident in HBM for `value`; the PCIe-inclusive rate of the same frames is reported beside it
(`end_to_end`), never as `value`.

  --gpus 1   whole frames on one GPU, four frames in flight (one context / stream / image each).
  --gpus N   north-star layout: every frame is tiled over the N GPUs in interleaved 16-row strips
             (rm_draw_strips), one process per GPU, no collective on the data path; `value` counts
             whole frames with each GPU's strips resident in its HBM, `end_to_end` adds the final
             host-side gather (rm_gather_strips: every rank copies its strips into one shared-memory
             frame; the only inter-process traffic is a completion counter), and `frames_sharded` is
             the other partition (whole frames per rank: BASELINE config 5's, selected as the
             headline by --camera orbit).

`python bench.py --gpus N` starts its own N rank processes (before anything touches the GPU) unless
it already runs under a launcher (WORLD_SIZE set, e.g. torch.distributed.run).  Rank 0 prints ONE
JSON line.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_VECTOR_PEAK_TFLOPS = 157.3  # ditto
BYTES_PER_PIXEL = 16             # one RGBA32F store per pixel: SURVEY 8(d) algorithmic bytes
SALU_CYCLES_PER_INST = 4.33       # measured: tools/ubench_salu.hip
VALU_CYCLES_PER_WAVE_INST = 2.0  # MI355X_MICROARCH.md: a wave64 VALU instruction occupies a SIMD-32 for 2 cycles


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--scene", default="g32")
    p.add_argument("--max-iter", type=int, default=256)
    p.add_argument("--kernel", type=int, default=0, help="rm_kernel enum (0 = default tuned kernel)")
    p.add_argument("--refill-min", type=int, default=0, help="idle lanes that trigger a refill (0 = library default)")
    p.add_argument("--waves-per-tile", type=int, default=0, help="1/2/4/8 waves share a tile (0 = default)")
    p.add_argument("--specialize", type=int, default=2, choices=[0, 1, 2],
                   help="structure specialisation of the march kernel (hipRTC): 0 interpreter kernel only, 1 compile in the "
                        "background, 2 compile when the scene is uploaded (default: the scene is static, so the one-off "
                        "compilation happens before the warm-up, like the reference's own shader compilation)")
    p.add_argument("--prune", type=int, default=-1, choices=[-1, 0, 1, 2],
                   help="far-primitive pruning in the specialised kernel (exact): 0 off, 1 on, 2 for programs with >= 12 primitives; "
                        "-1 (default) leaves the library default (2)")
    p.add_argument("--no-cull", action="store_true", help="A/B: disable the exact miss-ray culling")
    p.add_argument("--balance", type=int, default=-1, help="A/B: RM_OPT_BALANCE value (0 raster ... 3 last frame's durations)")
    p.add_argument("--camera", choices=["still", "orbit"], default="still",
                   help="still: the metric's camera; orbit: frame f of a 1024-frame orbit (BASELINE config 5)")
    p.add_argument("--mode", choices=["auto", "frames", "tile"], default="auto",
                   help="tile: every frame is tiled over the ranks in interleaved 16-row strips (north-star layout; auto for "
                        "--gpus > 1 with the still camera); frames: every rank renders whole frames (auto for one GPU and for "
                        "the orbit batch)")
    p.add_argument("--dist-backend", default="nccl", help="process-group backend for the timing barrier (nccl = RCCL)")
    p.add_argument("--all-ranks-on-device0", action="store_true",
                   help="rehearsal on a one-GPU box: every rank uses GPU 0 (use with --dist-backend gloo)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-legs", "--no-ab", dest="no_legs", action="store_true",
                   help="only the headline measurement (skips the serial / orbit / interpreter / end-to-end legs)")
    p.add_argument("--frames-in-flight", type=int, default=4,
                   help="frames drawn concurrently, each on its own context / stream / output buffer (1 = strictly serial)")
    p.add_argument("--cpu-sample-div", type=int, default=1, help="CPU baseline renders W/div x H/div")
    p.add_argument("--launch-timeout", type=float, default=900.0, help="--gpus N self-launch: seconds before the ranks are stopped")
    return p.parse_args(argv)


def host_cores():
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def launch_ranks(args, argv):
    """--gpus N without a launcher: start the N ranks ourselves.  This process never initialises the GPU."""
    from ray_marching_amd import launch
    rc, out = launch.run_ranks(args.gpus, launch.python_argv(os.path.abspath(__file__), argv), timeout=args.launch_timeout)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    elif out.strip():
        sys.stderr.write(out)
    if rc == 0 and not lines:
        sys.stderr.write("bench.py: rank 0 printed no result line\n")
        rc = 1
    return rc


def oracle_frame(args, scene_words, u_bytes, W, H, cores, want_counters=False):
    """The oracle's render of one frame from the very 144 uniform bytes the GPU drew with."""
    from oracle import cbind
    cbind.build()
    cc, words = scene_words
    u = cbind.Uniforms.from_buffer_copy(u_bytes)
    return cbind.render(u, (0.01, 100.0, args.max_iter), cc, words, W, H, threads=cores, want_counters=want_counters)


def cpu_baseline(args, scene_words, u_bytes, W, H):
    """The oracle (a CPU port of the reference shader) timed on the host cores, on a bounded sample: the same scene / camera /
    limits at 1/div^2 of the pixels (Mpx/s is resolution-normalised).  The image is kept: it is what `parity` compares the
    GPU's frames with."""
    cores = host_cores()
    oracle_frame(args, scene_words, u_bytes, 64, 36, cores)     # warm the threads / caches (any frame will do)
    t0 = time.perf_counter()
    img, cnt = oracle_frame(args, scene_words, u_bytes, W, H, cores, want_counters=True)
    dt = time.perf_counter() - t0
    return {"value": W * H / dt / 1e6, "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": "%dx%d render (1/%d of the pixels) of the same scene, camera and limits, %.1f s wall"
                      % (W, H, args.cpu_sample_div ** 2, dt),
            "label": "CPU restatement of the reference shader (oracle/rm_oracle.c), not wgpu"}, cnt, img


def compare_frames(gpu, ref):
    """(max |gpu - ref|, pixels with any differing BIT) of two RGBA32F frames.  The contract is bit-exactness; north_star's
    tolerance is 1e-4 per channel."""
    import numpy as np
    a, b = np.ascontiguousarray(gpu, dtype=np.float32), np.ascontiguousarray(ref, dtype=np.float32)
    if a.shape != b.shape:
        return float("inf"), int(max(a.size, b.size) // 4)
    differ = (a.view(np.uint32) != b.view(np.uint32)).any(axis=-1)
    n = int(differ.sum())
    if n == 0:
        return 0.0, 0
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    d = np.where(np.isnan(d), np.inf, d)        # a NaN on one side only is as wrong as it gets
    return float(d[differ].max()), n


def run_rank(args):
    import torch
    import torch.distributed as dist
    from ray_marching_amd import _ffi, camera, csg, renderer, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if args.all_ranks_on_device0:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d wants GPU %d, %d visible (one-GPU rehearsal: --all-ranks-on-device0 "
                         "--dist-backend gloo)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    # The process group carries the timing barrier and a MAX reduction, nothing else.  Default group: gloo; RCCL ("nccl") is
    # probed on top BEFORE any other GPU work and dropped by all ranks together if any of them cannot bring it up
    # (launch.init_timing_group) -- a timing barrier must not be what loses a multi-GPU run.
    tgroup, barrier_backend, red_device, group_world, fallback_reason = None, None, "cpu", 1, None
    if world > 1:
        from ray_marching_amd import launch
        tgroup, barrier_backend, red_device, group_world, fallback_reason = launch.init_timing_group(args.dist_backend, local_rank)
        if group_world != world:
            raise SystemExit("bench.py: the process group reports %d ranks, WORLD_SIZE is %d" % (group_world, world))
        if fallback_reason and rank == 0:
            sys.stderr.write("bench.py: RCCL barrier group unavailable (%s); using gloo\n" % fallback_reason)

    W, H, K = args.width, args.height, args.steps
    mode = args.mode
    if mode == "auto":
        mode = "tile" if (world > 1 and args.camera == "still") else "frames"
    tile = mode == "tile"
    F = max(1, args.frames_in_flight)
    SR = shard.DEFAULT_STRIP_ROWS
    node = csg.scene(args.scene)
    cc, words = csg.serialize(node)

    def make_context():
        r = renderer.RayMarchingResources(local_rank)
        r.set_option(_ffi.RM_OPT_KERNEL, args.kernel)
        r.set_option(_ffi.RM_OPT_SPECIALIZE, args.specialize)
        if args.prune >= 0:
            r.set_option(_ffi.RM_OPT_PRUNE, args.prune)
        if args.refill_min:
            r.set_option(_ffi.RM_OPT_REFILL_MIN, args.refill_min)
        if args.waves_per_tile:
            r.set_option(_ffi.RM_OPT_WAVES_PER_TILE, args.waves_per_tile)
        if args.no_cull:
            r.set_option(_ffi.RM_OPT_CULL, 0)
        if args.balance >= 0:
            r.set_option(_ffi.RM_OPT_BALANCE, args.balance)
        r.set_limits(renderer.RayMarchLimits(0.01, 100.0, args.max_iter))
        if len(words) > 255:
            r.resize_command_buffer(4 * (len(words) + 1 + 63) // 64 * 64)
        r.set_program(cc, words)
        return r

    ctxs = [make_context() for _ in range(F)]
    res = ctxs[0]
    streams = [torch.cuda.Stream() for _ in range(F)]   # kernels, events and syncs of frame f all use stream f % F
    torch.cuda.set_stream(streams[0])

    still_ctl = camera.OrbitCameraController.new([0.0, 0.0, 0.0], 5.0)
    still_ctl.update(camera.Orbit([35.0, -25.0]))
    orbit_ctl = camera.OrbitCameraController.new([0.0, 0.0, 0.0], 5.0)

    def uniforms(step, cam, frame_stride=1, frame_offset=0):
        if cam == "orbit":                                  # config 5: frame f of a 1024-frame orbit
            f = (step * frame_stride + frame_offset) % 1024
            orbit_ctl.set_angles(2.0 * 3.141592653589793 * f / 1024.0, -0.25, 5.0)
            return renderer.prepare_uniforms((float(W), float(H)), orbit_ctl.camera())
        return renderer.prepare_uniforms((float(W), float(H)), still_ctl.camera())

    total = args.warmup + K
    my_rows = shard.strip_row_count(H, SR, rank, world)
    full = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(F)]          # whole frames
    strips = [torch.empty((max(my_rows, 1), W, 4), dtype=torch.float32, device="cuda") for _ in range(F)] if world > 1 else full

    def barrier():
        if world > 1:
            dist.barrier(group=tgroup)

    def sync_all():
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()

    def max_over_ranks(values):
        if world == 1:
            return [float(v) for v in values]
        t = torch.tensor(list(values), dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=tgroup)
        return [float(v) for v in t]

    def prepare(c, u):
        """RayMarchingCallback::prepare (renderer.rs:196-242): the uniform write and the rewrite of the command buffer the
        reference does every frame (identical bytes: recognised by a memcmp, nothing is decoded or copied again)."""
        c.set_uniforms(u)
        c.set_program(cc, words)

    def draw_full(i):
        ctxs[i].draw_device(W, H, full[i].data_ptr(), stream=streams[i].cuda_stream)

    def draw_strips(i):
        if my_rows:
            ctxs[i].draw_strips_device(W, H, SR, rank, world, strips[i].data_ptr(), stream=streams[i].cuda_stream)

    def timed(n_ctx, draw, unis, after_issue=None, finish=None):
        """W warm-up + K timed steps over the first n_ctx contexts, bracketed by barrier + synchronize on both sides.
        Returns (seconds [max over ranks], mean per-draw ms from HIP events on the launch stream, mean march-kernel ms
        from the library's own events)."""
        for s in range(args.warmup):
            prepare(ctxs[s % n_ctx], unis[s])
            draw(s % n_ctx)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        for c in ctxs[:n_ctx]:
            c.set_option(_ffi.RM_OPT_TIMING, 1)     # library-side HIP events around the march kernel itself
        sync_all()
        gc_was_on = gc.isenabled()
        gc.disable()                                 # a collection in the middle of 12 ms of timed steps would be a large share of them
        t0 = time.perf_counter()
        for k in range(K):
            i = k % n_ctx
            prepare(ctxs[i], unis[args.warmup + k])
            ev[k][0].record(streams[i])
            draw(i)                                  # paint(): the kernels, on this frame's stream
            ev[k][1].record(streams[i])
            if after_issue:
                after_issue(k, i)
        if finish:
            finish()
        torch.cuda.synchronize()
        barrier()
        seconds = time.perf_counter() - t0
        if gc_was_on:
            gc.enable()
        per_draw = sum(a.elapsed_time(b) for a, b in ev) / max(1, K)
        kms = [c.info(_ffi.RM_INFO_KERNEL_MS) for c in ctxs[:n_ctx]]
        for c in ctxs[:n_ctx]:
            c.set_option(_ffi.RM_OPT_TIMING, 0)
        seconds, per_draw, kms = max_over_ranks([seconds, per_draw, sum(kms) / len(kms)])
        return seconds, per_draw, kms

    def last_uniforms(us, i, n_ctx):
        """Uniforms of the last draw timed() made into buffer i (timed step k draws into k % n_ctx with us[warmup + k], warm-up
        step s into s % n_ctx with us[s]; before that every context drew once with us[0])."""
        ks = [k for k in range(K) if k % n_ctx == i]
        if ks:
            return us[args.warmup + ks[-1]]
        ws = [w for w in range(args.warmup) if w % n_ctx == i]
        return us[ws[-1]] if ws else us[0]

    cam_stride, cam_offset = (world, rank) if not tile else (1, 0)    # frames mode: rank r renders frames r, r + world, ...
    unis = [uniforms(s, args.camera, cam_stride, cam_offset) for s in range(total)]
    # set-up, not a step: every context draws once so that its scratch buffers exist and the scene's kernel is compiled
    # and loaded before anything is timed, however small --warmup is
    headline_draw = draw_strips if tile else draw_full
    for i in range(F):
        prepare(ctxs[i], unis[0])
        headline_draw(i)
    torch.cuda.synchronize()

    # ---- headline: image(s) resident in HBM ------------------------------------------------------------------
    elapsed, draw_ms, kernel_ms = timed(F, headline_draw, unis)
    specialized = bool(res.info(_ffi.RM_INFO_SPECIALIZED))                    # what the LAST timed launch ran
    jit_ms = res.info(_ffi.RM_INFO_JIT_COMPILE_MS)
    jit_from_cache = bool(res.info(_ffi.RM_INFO_JIT_FROM_CACHE))
    # what the compiler takes for this structure when the on-disk cache of compiled kernels does not have it (the first time a
    # structure is drawn on a machine): the same source through hipRTC with the cache switched off, untimed set-up
    jit_cold_ms = None
    if rank == 0 and specialized:
        old_dir = os.environ.get("RM_JIT_CACHE_DIR")
        os.environ["RM_JIT_CACHE_DIR"] = "off"
        try:
            rc_, jit_cold_ms, _, _ = renderer.jit_compile(cc, words, 4, prune=bool(res.info(_ffi.RM_INFO_PRUNED)))
            if rc_ != _ffi.RM_OK:
                jit_cold_ms = None
        finally:
            if old_dir is None:
                del os.environ["RM_JIT_CACHE_DIR"]
            else:
                os.environ["RM_JIT_CACHE_DIR"] = old_dir
    last = (strips if tile else full)[(K - 1) % F]
    checksum = float(last[..., :3].double().sum().item())                     # touches the result: nothing was skipped
    # every in-flight buffer as the timed loop left it, with the uniforms of the last draw into it: `parity` (below) compares
    # ALL of them with the oracle's render of the same uniforms
    snaps = []
    if world == 1 and not tile and not args.no_cpu_baseline:
        for i in range(F):
            snaps.append((bytes(last_uniforms(unis, i, F)), full[i].cpu().numpy()))
    frames_per_step = 1 if tile else world
    value = W * H * K * frames_per_step / elapsed / 1e6

    legs = {}
    exit_code = 0
    orbit_snap = None
    useful_occ = None
    if not args.no_legs:
        # strictly serial loop (one frame in flight): what ONE frame costs, and the only honest source of per-launch
        # durations (with several frames in flight an event-bracketed draw also contains time queued behind the others)
        s_el, s_draw, s_k = timed(1, headline_draw, unis)
        legs["one_frame_in_flight"] = {"value": W * H * K * frames_per_step / s_el / 1e6, "unit": "Mpixels/s",
                                       "draw_ms": s_draw, "kernel_ms": s_k}
        # USEFUL lane occupancy of the march kernel (one untimed draw with per-wave counters, rm_read_wave_stats): lanes that
        # carry a live ray (or a waiting hit, in a tap phase) over 64 x the map_scene iterations of all waves.  A wave that
        # marches its 64 rays in step keeps its exec mask full while the lanes whose rays ended early do dead work: the
        # exec-mask figure of the PMC pass does not see that, this one does.
        try:
            import numpy as np
            res.set_option(_ffi.RM_OPT_WAVE_STATS, 1)
            prepare(res, unis[-1])
            headline_draw(0)
            torch.cuda.synchronize()
            st = res.wave_stats()
            res.set_option(_ffi.RM_OPT_WAVE_STATS, 0)
            iters = int((st[:, 2] & np.uint64(0xFFFFFFFF)).sum())
            live = int((st[:, 3] & np.uint64(0xFFFFFFFF)).sum())
            useful_occ = live / (64.0 * iters) if iters else None
        except Exception as e:      # diagnostics only
            sys.stderr.write("bench.py: wave statistics unavailable: %s\n" % e)
            res.set_option(_ffi.RM_OPT_WAVE_STATS, 0)
        # end to end: the image lands in host memory.  N = 1: D2H into one pinned frame; N > 1: the north-star gather --
        # every rank copies its strips to their rows of ONE shared-memory frame (two frame slots, frame k -> slot k % 2,
        # so frame k + 1 renders while frame k is copied), a per-rank completion counter says when a frame is whole.
        if args.camera == "still" or tile:
            D = 2 if F >= 2 else 1
            name = "rm_bench_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid() if world > 1 else os.getpid())
            def host_barrier():     # the default group is gloo (host side), whatever carries the timing barrier
                if world > 1:
                    dist.barrier()

            shared = shard.SharedImage(name, W, H, slots=D).open(rank, world, host_barrier)
            shared.register()
            done = [torch.cuda.Event() for _ in range(D)]
            state = {"base": 0}

            def deliver(k):      # frame k of this leg has left this rank's GPU; wait until it is whole on the host
                done[k % D].synchronize()
                shared.mark_done(rank, state["base"] + k + 1)
                shared.wait_all(state["base"] + k + 1)

            def after_issue(k, i):
                if my_rows:
                    ctxs[i].gather_strips(W, H, SR, rank, world, strips[i].data_ptr(), shared.slot_address(k),
                                          stream=streams[i].cuda_stream)
                done[k % D].record(streams[i])
                if k >= D - 1:
                    deliver(k - (D - 1))

            def finish():
                for k in range(max(0, K - (D - 1)), K):
                    deliver(k)

            # set-up, not a step: the first copy into each slot of the freshly registered frame pays a one-off mapping cost
            # (~6 ms per slot at 1080p)
            for i in range(D):
                prepare(ctxs[i], unis[0])
                (draw_strips if world > 1 else draw_full)(i)
                if my_rows:
                    ctxs[i].gather_strips(W, H, SR, rank, world, strips[i].data_ptr(), shared.slot_address(i), stream=streams[i].cuda_stream)
            torch.cuda.synchronize()
            host_barrier()
            e_el, _, _ = timed(D, draw_strips if world > 1 else draw_full, unis, after_issue, finish)
            state["base"] += K
            host_barrier()
            e2e = {"value": W * H * K / e_el / 1e6, "unit": "Mpixels/s", "ms_per_frame": e_el / K * 1e3,
                   "what": ("frame tiled over %d GPUs + host-side gather: each rank's strips D2H straight into their rows of one "
                            "page-locked POSIX shared-memory frame (rm_gather_strips), completion counter per rank, two frames "
                            "in flight" % world) if world > 1
                   else "draw + D2H of the frame into page-locked host memory (rm_gather_strips), two frames in flight"}
            if rank == 0:
                # the gathered frame must be the frame one GPU renders, byte for byte (tiling invariance)
                prepare(ctxs[0], unis[-1])
                draw_full(0)
                torch.cuda.synchronize()
                e2e["gathered_frame_identical_to_one_gpu_render"] = bool(
                    (torch.from_numpy(shared.slot((K - 1) % D)) == full[0].cpu()).all().item()) if args.camera == "still" else None
            legs["end_to_end"] = e2e
            shared.close(host_barrier)
        if world > 1 and tile:
            # the other partition of north_star (BASELINE config 5): whole frames per rank, no gather
            funis = [uniforms(s, args.camera, world, rank) for s in range(total)]
            for i in range(F):
                prepare(ctxs[i], funis[0])
                draw_full(i)
            torch.cuda.synchronize()
            f_el, _, _ = timed(F, draw_full, funis)
            legs["frames_sharded"] = {"value": W * H * K * world / f_el / 1e6, "unit": "Mpixels/s", "scaling": "weak",
                                      "what": "every rank renders whole frames (frame f -> rank f %% %d), %d in flight per rank" % (world, F)}
        if world == 1 and args.camera == "still":
            # an orbiting camera: every frame a new view, so the temporal tile order (RM_OPT_BALANCE = 3) predicts from a
            # neighbouring frame instead of the identical one
            ounis = [uniforms(s, "orbit") for s in range(total)]
            o_el, _, _ = timed(F, draw_full, ounis)
            legs["orbit_camera"] = {"value": W * H * K / o_el / 1e6, "unit": "Mpixels/s",
                                    "what": "same loop, frame f of a 1024-frame orbit (yaw 2 pi f / 1024, pitch -0.25, r 5)"}
            if not args.no_cpu_baseline:     # one sampled frame of the leg for `parity`: the last one drawn
                i_last = (K - 1) % F
                orbit_snap = (bytes(last_uniforms(ounis, i_last, F)), full[i_last].cpu().numpy(), (args.warmup + K - 1) % 1024)
        if world == 1 and specialized:
            # A/B: the interpreter kernel, i.e. the design north_star spells out (node array staged in LDS and INTERPRETED)
            res.set_option(_ffi.RM_OPT_SPECIALIZE, 0)
            a_el, _, a_k = timed(1, draw_full, unis)
            loop = int(res.info(_ffi.RM_INFO_INTERPRETER_LOOP))
            legs["ab_interpreter_kernel"] = {
                "kernel": "rm_render_v5 (interpreter: LDS-staged records; %s), one frame in flight"
                          % ["general record loop, accumulator machine", "stack-free chain loop",
                             "stack-free chain loop over the records wave-level culling names", "tree loop, one dispatch per record",
                             "tree loop over the records wave-level culling leaves",
                             "record machine over the units of a blending chain wave-level culling names"][min(loop, 5)],
                "value": W * H * K / a_el / 1e6, "unit": "Mpixels/s", "kernel_ms": a_k,
                "same_image": float(full[0][..., :3].double().sum().item()) == checksum or args.camera != "still"}
            res.set_option(_ffi.RM_OPT_SPECIALIZE, args.specialize)

    if rank == 0:
        serial = legs.get("one_frame_in_flight")
        # per-launch figures come from the serial leg; without it (--no-legs) from the headline loop, labelled as such
        launch_draw_ms = serial["draw_ms"] if serial else draw_ms
        launch_kernel_ms = serial["kernel_ms"] if serial else kernel_ms
        rows_here = my_rows if tile else H
        ach = BYTES_PER_PIXEL * W * rows_here / (launch_draw_ms * 1e-3) / 1e9
        traffic = None
        if world == 1:
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    traffic = json.load(f).get("%s_%dx%d_%d" % (args.scene, W, H, args.max_iter))
            except (OSError, ValueError):
                traffic = None
        if tile:
            sharding = ("every frame tiled over %d ranks in interleaved %d-row strips (rm_draw_strips: all of a rank's strips in one "
                        "launch), strips resident in each GPU's HBM; no collective; `end_to_end` adds the host-side gather into one "
                        "shared-memory frame" % (world, SR))
        else:
            sharding = ("frames over ranks (frame f -> rank f %% %d), no collective" % world) if world > 1 else "single GPU"
        line = {
            "metric": "Mpixels/s at 1920x1080, 256-step march, 32-node SDF",
            "value": value, "unit": "Mpixels/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
            "scaling": None if world == 1 else ("strong" if tile else "weak"),
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%dx%d, %s (%d commands / %d words), %d max steps, 16 rays/px, RGBA32F out"
                                   % (W, H, args.scene, cc, len(words), args.max_iter),
                       "step": "prepare (uniform write + command-buffer rewrite, renderer.rs:213-239) + draw, image resident in HBM",
                       "camera": args.camera, "kernel": args.kernel, "specialized_kernel": specialized,
                       "jit_compile_ms": jit_ms, "jit_from_disk_cache": jit_from_cache, "jit_cold_compile_ms": jit_cold_ms,
                       "frames_in_flight": F, "sharding": sharding,
                       "barrier_backend": barrier_backend, "barrier_group_world_size": group_world if world > 1 else None,
                       "barrier_fallback_reason": fallback_reason},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": ("rm_render_v5_spec (hipRTC-compiled for this scene's structure)" if specialized else "rm_render_v5")
                                   + ", the dominant (march) kernel",
                         "kernel_ms": launch_kernel_ms, "draw_ms": launch_draw_ms,
                         "timing": ("one frame in flight: HIP events on the launch stream; kernel_ms = the march kernel alone, draw_ms = "
                                    "the draw's three launches (pre-pass, sort, march), which `achieved` divides this GPU's %d x %d x 16 "
                                    "bytes by" % (W, rows_here)) if serial else "headline loop (frames overlap: upper bounds)",
                         "note": "algorithmic bytes = 16 B/pixel (one RGBA32F store); the kernel is FP32-VALU bound by ~3 orders of "
                                 "magnitude, see `compute`"},
            "checksum_rgb": checksum,
        }
        if world == 1 and not args.no_legs:
            # the measured HBM-write ceiling of this GPU, same run: a fill kernel with 16 B/lane stores over 1 GiB
            # (SURVEY 8(d)); the declared peak above stays the spec figure
            try:
                fill = res.measure_write_bandwidth(1 << 30, 10)
                line["roofline"]["measured_write_GBps"] = fill
                line["roofline"]["frac_of_measured"] = ach / fill if fill > 0 else None
            except Exception as e:      # diagnostics only
                line["roofline"]["measured_write_GBps"] = None
                line["roofline"]["measured_write_error"] = str(e)
        line.update(legs)
        parity_ok = True
        if world == 1 and not args.no_cpu_baseline:
            import numpy as np
            div = max(1, args.cpu_sample_div)
            cw, ch = W // div, H // div
            sample_u = bytes(renderer.prepare_uniforms((float(cw), float(ch)), still_ctl.camera()))
            base, cnt, ref_img = cpu_baseline(args, (cc, words), sample_u, cw, ch)
            line["cpu_baseline"] = base
            # ---- parity: the frames this run TIMED against the oracle's render of the same uniforms, bit for bit ----
            cores = base["cores"]
            refs = {sample_u: ref_img} if div == 1 else {}
            worst, differing, checked = 0.0, 0, 0
            if div == 1:
                for ub, img in snaps:                       # every in-flight buffer of the headline loop
                    if ub not in refs:
                        refs[ub] = oracle_frame(args, (cc, words), ub, W, H, cores)
                    d, n = compare_frames(img, refs[ub])
                    worst, differing, checked = max(worst, d), differing + n, checked + 1
                what = "every in-flight buffer of the headline loop (%d x %dx%d) as the timed region left it" % (len(snaps), W, H)
            else:                                           # bounded sample: the reduced frame the CPU baseline rendered
                res.set_option(_ffi.RM_OPT_SPECIALIZE, args.specialize)
                res.set_uniforms(_ffi.Uniforms.from_buffer_copy(sample_u))
                d, n = compare_frames(res.draw(cw, ch), ref_img)
                worst, differing, checked = d, n, 1
                what = "one %dx%d frame of the same scene and camera (--cpu-sample-div %d: the timed frames are not compared)" % (cw, ch, div)
            line["parity"] = {"vs": "oracle", "max_abs_diff": worst, "pixels_differing": differing, "frames_checked": checked,
                              "what": what, "tolerance": "bit-exact (north_star allows 1e-4 per channel)",
                              "oracle": "oracle/rm_oracle.c, parity unpinned against a real wgpu render (DESIGN 3)"}
            parity_ok = differing == 0
            if orbit_snap is not None and "orbit_camera" in line and div == 1:
                ub, img, f = orbit_snap
                d, n = compare_frames(img, oracle_frame(args, (cc, words), ub, W, H, cores))
                line["orbit_camera"]["parity"] = {"vs": "oracle", "max_abs_diff": d, "pixels_differing": n, "frames_checked": 1,
                                                  "what": "the last frame the leg drew (frame %d of the orbit), %dx%d" % (f, W, H)}
                parity_ok = parity_ok and n == 0
            if args.camera == "still":
                # oracle counters -> algorithmic map_scene evaluations of the full-size frame
                scale = (W * H) / float(cw * ch)
                evals = (cnt["march_steps"] + cnt["normal_taps"]) * scale
                line["compute"] = {"map_scene_evals_per_frame": evals,
                                   "evals_per_s": evals / (launch_kernel_ms * 1e-3),
                                   "note": "evaluations counted by the oracle on the CPU sample, scaled by pixel count"}
                line["compute"].update(valu_view(args, res, launch_kernel_ms, evals, words, useful_occ))
        print(json.dumps(line), flush=True)
        if not parity_ok:
            sys.stderr.write("bench.py: the GPU frames differ from the oracle's: %s\n" % json.dumps(line.get("parity")))
            exit_code = 1
    for c in ctxs:
        c.close()
    if world > 1:
        dist.destroy_process_group()
    return exit_code


def valu_view(args, res, march_ms, evals, words, useful_occ=None):
    """The honest bound: FP32 vector issue.  Instruction counts and the clock come from the committed PMC
    pass of this configuration (profiles/pmc_traffic.json), the duration from this run (march kernel, one frame in
    flight).  Floor: a wave64 VALU instruction occupies its SIMD-32 for 2 cycles (MI355X_MICROARCH.md), transcendental
    ones for 8 (profiles/r02_ubench_valu_issue_cycles.txt)."""
    from ray_marching_amd import _ffi
    out = {}
    # algorithmic flops of the reference's map_scene for this program (10 / sphere, 22 / box, 1 / union,
    # 2 / subtraction), times the evaluations the REFERENCE would make (the GPU skips those of rays it proves to miss)
    f_prog, i = 0, 0
    per_op = {0: (10, 4), 1: (22, 6), 100: (1, 0), 101: (2, 0)}
    while i < len(words):
        fl, n = per_op.get(int(words[i]), (0, 0))
        f_prog += fl
        i += 1 + n
    out["flops_per_eval_reference"] = f_prog
    # what the REFERENCE's evaluation count would amount to in this kernel's time: NOT a rate the chip reaches (it can exceed
    # the vector peak) -- the exact miss tests and the far-primitive pruning skip most of those evaluations
    out["reference_equivalent_TFLOPs"] = evals * f_prog / (march_ms * 1e-3) / 1e12
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pmc = json.load(fh).get("%s_%dx%d_%d_valu" % (args.scene, args.width, args.height, args.max_iter))
    except (OSError, ValueError):
        pmc = None
    if pmc:
        simds = 4.0 * res.info(_ffi.RM_INFO_CU_COUNT)
        insts = pmc["SQ_INSTS_VALU"]
        trans = pmc.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
        clock_ghz = pmc.get("clock_ghz", 2.4)
        flop_insts = pmc["SQ_INSTS_VALU_ADD_F32"] + pmc["SQ_INSTS_VALU_MUL_F32"] + 2.0 * pmc["SQ_INSTS_VALU_FMA_F32"] + trans
        cycles_per_inst = march_ms * 1e-3 * clock_ghz * 1e9 * simds / insts
        t_cyc = pmc.get("trans_issue_cycles", 8.0)
        floor_cycles = (VALU_CYCLES_PER_WAVE_INST * (insts - trans) + t_cyc * trans) / insts
        meas = pmc.get("measured_simple_issue_cycles")
        meas_floor = (meas * (insts - trans) + t_cyc * trans) / insts if meas else None
        out.update({
            "bound": "fp32 valu issue",
            "valu_insts_per_frame": insts,
            "clock_ghz": clock_ghz,
            "cycles_per_valu_inst_per_simd": cycles_per_inst,
            "issue_floor_cycles_per_inst": floor_cycles,
            "frac_of_valu_issue": floor_cycles / cycles_per_inst,
            "measured_issue_cycles_per_inst": meas_floor,
            "frac_of_measured_valu_issue": meas_floor / cycles_per_inst if meas_floor else None,
            "lds_pipe_busy": pmc.get("SQ_ACTIVE_INST_LDS", 0.0) * 4.0 / (simds / 4.0) / (march_ms * 1e-3 * clock_ghz * 1e9) if pmc.get("SQ_ACTIVE_INST_LDS") else None,
            # the other issue bound: a CU has ONE scalar unit for its four SIMDs -- a scalar instruction costs its SIMD 4.3 cycles
            # (profiles/r03_ubench_scalar_issue_cycles.txt), and scalar and vector instructions of different waves issue side by side
            "salu_insts_per_frame": pmc.get("SQ_INSTS_SALU"),
            "scalar_issue_cycles_per_inst": SALU_CYCLES_PER_INST,
            "scalar_unit_share_of_kernel_time": (pmc["SQ_INSTS_SALU"] * SALU_CYCLES_PER_INST / simds) / (march_ms * 1e-3 * clock_ghz * 1e9) if pmc.get("SQ_INSTS_SALU") else None,
            "vector_issue_share_of_kernel_time": (meas_floor if meas_floor else floor_cycles) / cycles_per_inst,
            # lanes that carry a live ray (this run's wave counters), not lanes whose exec bit is set (the PMC figure, kept
            # beside it): a wave marching 64 rays in step executes the lanes whose rays ended early as dead work
            "useful_lane_occupancy": useful_occ,
            "exec_mask_lane_occupancy": pmc["lane_occupancy"],
            "executed_fp32_TFLOPs": flop_insts * 64.0 * (useful_occ if useful_occ else pmc["lane_occupancy"]) / (march_ms * 1e-3) / 1e12,
            "exec_mask_fp32_TFLOPs": flop_insts * 64.0 * pmc["lane_occupancy"] / (march_ms * 1e-3) / 1e12,
            "peak_fp32_vector_TFLOPs": FP32_VECTOR_PEAK_TFLOPS,
            "valu_note": "instruction counts and clock (GRBM_GUI_ACTIVE / 8 / duration): committed rocprofv3 PMC pass of this configuration (" +
                         pmc.get("source", "profiles/") + "); floor: 2 cycles per wave64 VALU instruction on a SIMD-32 (MI355X_MICROARCH.md), "
                         "transcendentals 8; `measured`: what back-to-back v_add / v_mul / v_fma reach on this chip, 2.3-2.4 cycles "
                         "(profiles/r02_ubench_valu_issue_cycles.txt); lds_pipe_busy = SQ_ACTIVE_INST_LDS (quad-cycles) per CU over the kernel's "
                         "cycles; the 157.3 TFLOP/s peak assumes packed FMAs, the arithmetic contract forbids fusing (DESIGN 2)",
        })
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
