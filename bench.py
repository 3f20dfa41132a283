#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the SDF ray-marching hot path on MI355X.

A step = one full frame (prepare's uniform write + the draw) of the BASELINE metric
workload: 1920x1080, 32-node graph (G32), 256 max steps, still orbit camera of SURVEY 8(d).
The output image stays resident in HBM.  N > 1 (launched by torch.distributed.run, one rank
per GPU): frames are sharded over ranks with no data-path collective ("weak" scaling: every
rank renders its own K frames); only the timing barrier / max-over-ranks uses the process group.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_VECTOR_PEAK_TFLOPS = 157.3  # ditto
BYTES_PER_PIXEL = 16             # one RGBA32F store per pixel: SURVEY 8(d) algorithmic bytes


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--scene", default="g32")
    p.add_argument("--max-iter", type=int, default=256)
    p.add_argument("--kernel", type=int, default=0, help="rm_kernel enum (0 = default tuned kernel)")
    p.add_argument("--refill-min", type=int, default=0, help="raypool refill threshold (0 = library default)")
    p.add_argument("--waves-per-tile", type=int, default=0, help="v3 kernels: 1/2/4/8 waves share a tile (0 = default)")
    p.add_argument("--specialize", type=int, default=2, choices=[0, 1, 2],
                   help="structure specialisation of the march kernel (hipRTC): 0 interpreter kernel only, 1 compile in the "
                        "background, 2 compile when the scene is uploaded (default: the scene is static, so the one-off "
                        "compilation happens before the warm-up, like the reference's own shader compilation)")
    p.add_argument("--prune", type=int, default=-1, choices=[-1, 0, 1, 2],
                   help="far-primitive pruning in the specialised kernel (exact): 0 off, 1 on, 2 for programs with >= 12 primitives; "
                        "-1 (default) leaves the library default (2)")
    p.add_argument("--no-cull", action="store_true", help="A/B: disable the exact miss-ray culling")
    p.add_argument("--no-balance", action="store_true", help="A/B: disable the heaviest-tile-first pre-pass")
    p.add_argument("--balance", type=int, default=-1, help="A/B: RM_OPT_BALANCE value (0 raster, 1 fullest tiles first, 2 silhouette tiles first)")
    p.add_argument("--camera", choices=["still", "orbit"], default="still")
    p.add_argument("--mode", choices=["frames", "tile"], default="frames",
                   help="frames: every rank renders whole frames (weak scaling, default); tile: ONE frame per step "
                        "is tiled over the ranks in interleaved 16-row strips (strong scaling, north-star layout)")
    p.add_argument("--dist-backend", default="nccl", help="process-group backend for the timing barrier (nccl = RCCL)")
    p.add_argument("--all-ranks-on-device0", action="store_true",
                   help="rehearsal on a one-GPU box: every rank uses GPU 0 (use with --dist-backend gloo)")
    p.add_argument("--gather", nargs="?", const="gloo", default=None, choices=["gloo", "shm"],
                   help="tile mode: include the host-side gather in the timed region: gloo = gather to rank 0 over the "
                        "process group; shm = every rank copies its strips into one shared-memory image (never RCCL)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-ab", action="store_true", help="skip the serial and interpreter-kernel legs that follow the timed region")
    p.add_argument("--frames-in-flight", type=int, default=4,
                   help="frames drawn concurrently, each on its own context / stream / output buffer (1 = strictly serial)")
    p.add_argument("--cpu-sample-div", type=int, default=1, help="CPU baseline renders W/div x H/div")
    return p.parse_args()


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


def cpu_baseline(args, scene_words, cam_events):
    """The oracle (a CPU port of the reference shader) timed on the host cores, on a bounded
    sample: the same scene/camera/limits at 1/div^2 of the pixels (Mpx/s is resolution-normalised)."""
    from oracle import cbind
    cbind.build()
    cc, words = scene_words
    W, H = args.width // args.cpu_sample_div, args.height // args.cpu_sample_div
    cores = host_cores()
    u, *_ = cbind.orbit_uniforms((float(W), float(H)), events=cam_events)
    lim = (0.01, 100.0, args.max_iter)
    cbind.render(u, lim, cc, words, 64, 36, threads=cores)     # warm the threads / caches
    t0 = time.perf_counter()
    _, cnt = cbind.render(u, lim, cc, words, W, H, threads=cores, want_counters=True)
    dt = time.perf_counter() - t0
    return {"value": W * H / dt / 1e6, "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": "%dx%d render (1/%d of the pixels) of the same scene, camera and limits, %.1f s wall"
                      % (W, H, args.cpu_sample_div ** 2, dt),
            "label": "CPU restatement of the reference shader (oracle/rm_oracle.c), not wgpu"}, cnt, (W, H)


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from ray_marching_amd import _ffi, camera, csg, renderer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if args.all_ranks_on_device0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.dist_backend)
    assert args.gpus == world, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)

    W, H = args.width, args.height
    from ray_marching_amd import shard
    tile = args.mode == "tile"
    # Frames in flight: frame f is drawn by context f % F on stream f % F into buffer f % F (a renderer with F frames in flight).
    # A frame's draw ends with a tail in which its last tiles drain and most of the chip idles; the next frame's
    # launches fill it.  F = 1 is the strictly serial loop.  (Tile mode gathers every frame: serial by nature.)
    F = 1 if tile else max(1, args.frames_in_flight)
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
        if args.no_balance:
            r.set_option(_ffi.RM_OPT_BALANCE, 0)
        r.set_limits(renderer.RayMarchLimits(0.01, 100.0, args.max_iter))
        if len(words) > 255:
            r.resize_command_buffer(4 * (len(words) + 1 + 63) // 64 * 64)
        r.set_program(cc, words)
        return r

    ctxs = [make_context() for _ in range(F)]
    res = ctxs[0]

    still_events = [(1, 35.0, -25.0)]                      # Orbit([35,-25]): yaw 0.35, pitch -0.25
    ctl = camera.OrbitCameraController.new([0.0, 0.0, 0.0], 5.0)
    ctl.update(camera.Orbit([35.0, -25.0]))

    def uniforms_for(step):
        if args.camera == "orbit":                          # config 5: frame f of a 1024-frame orbit
            f = (step * world + rank) % 1024
            ctl.set_angles(2.0 * 3.141592653589793 * f / 1024.0, -0.25, 5.0)
        return renderer.prepare_uniforms((float(W), float(H)), ctl.camera())

    my_rows = shard.strip_row_count(H, shard.DEFAULT_STRIP_ROWS, rank, world) if tile else H
    outs = [torch.empty((max(my_rows, 1), W, 4), dtype=torch.float32, device="cuda") for _ in range(F)]
    out = outs[0]
    gloo = dist.new_group(backend="gloo") if (tile and args.gather and world > 1) else None
    shared = pinned = None
    if tile and args.gather == "shm":
        def host_barrier():
            dist.barrier(group=gloo)
        shared = shard.SharedImage("rm_bench_%s" % os.environ.get("MASTER_PORT", "0"), W, H).open(
            rank, world, host_barrier if world > 1 else (lambda: None))
        pinned = torch.empty((max(my_rows, 1), W, 4), dtype=torch.float32).pin_memory()
    streams = [torch.cuda.Stream() for _ in range(F)]   # kernels, events and syncs of frame f all use stream f % F
    stream = streams[0]
    torch.cuda.set_stream(stream)
    sptr = stream.cuda_stream

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def draw(i=0):
        if tile:
            res.draw_strips_device(W, H, shard.DEFAULT_STRIP_ROWS, rank, world, out.data_ptr(), stream=sptr)
            if args.gather == "shm":      # D2H into pinned memory, then this rank's strips go to their place in the shared frame
                with torch.cuda.stream(stream):
                    pinned.copy_(out, non_blocking=True)
                stream.synchronize()
                shared.put_strips(pinned[:my_rows].numpy(), rank, world)
                if world > 1:
                    dist.barrier(group=gloo)
            elif args.gather:             # final host-side gather (D2H + gloo), never RCCL
                stream.synchronize()
                shard.gather_image(out[:my_rows].cpu().numpy(), W, H, rank, world, group=gloo)
        else:
            ctxs[i].draw_device(W, H, outs[i].data_ptr(), stream=streams[i].cuda_stream)

    total = args.warmup + args.steps
    unis = [uniforms_for(s) for s in range(total)]
    # set-up, not a step: every context draws once so that its scratch buffers exist and the scene's kernel is compiled
    # and loaded before anything is timed, however small --warmup is
    for i in range(F):
        ctxs[i].set_uniforms(unis[0])
        draw(i)
    torch.cuda.synchronize()

    def timed_run(n_ctx):
        """W warm-up + K timed steps over the first n_ctx contexts.  Returns (seconds, mean per-draw ms from events,
        mean march-kernel ms from the library's own events)."""
        for s in range(args.warmup):
            ctxs[s % n_ctx].set_uniforms(unis[s])
            draw(s % n_ctx)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for c in ctxs[:n_ctx]:
            c.set_option(_ffi.RM_OPT_TIMING, 1)     # library-side HIP events around the march kernel itself
        sync_all()
        t0 = time.perf_counter()
        for k in range(args.steps):
            i = k % n_ctx
            ctxs[i].set_uniforms(unis[args.warmup + k])        # prepare(): uniform write
            ev[k][0].record(streams[i])
            draw(i)                                             # paint(): the kernels, on this frame's stream
            ev[k][1].record(streams[i])
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        seconds = time.perf_counter() - t0
        per_draw = sum(a.elapsed_time(b) for a, b in ev) / max(1, args.steps)    # all launches of a draw
        kms = [c.info(_ffi.RM_INFO_KERNEL_MS) for c in ctxs[:n_ctx]]             # the dominant (march) kernel alone
        for c in ctxs[:n_ctx]:
            c.set_option(_ffi.RM_OPT_TIMING, 0)
        return seconds, per_draw, sum(kms) / len(kms)

    elapsed, draw_ms, kernel_ms = timed_run(F)
    specialized = bool(res.info(_ffi.RM_INFO_SPECIALIZED))                    # what the LAST timed launch ran
    jit_ms = res.info(_ffi.RM_INFO_JIT_COMPILE_MS)
    if world > 1:
        t = torch.tensor([elapsed, kernel_ms, draw_ms], dtype=torch.float64,
                         device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms, draw_ms = float(t[0]), float(t[1]), float(t[2])

    checksum = float(outs[(args.steps - 1) % F][..., :3].double().sum().item())    # touches the result: nothing was skipped

    # Legs after the timed region (N = 1 only), reported beside the headline number:
    #   one_frame_in_flight     the strictly serial loop (what an interactive frame costs end to end)
    #   ab_interpreter_kernel   the same, through the interpreter kernel, i.e. the design north_star spells out
    #                           (node array staged in LDS and INTERPRETED)
    serial = ab = None
    if world == 1 and not tile and not args.no_ab:
        if F > 1:
            s_el, s_draw, s_k = timed_run(1)
            serial = {"value": W * H * args.steps / s_el / 1e6, "unit": "Mpixels/s", "draw_ms": s_draw, "kernel_ms": s_k,
                      "same_image": float(outs[0][..., :3].double().sum().item()) == checksum or args.camera != "still"}
        if specialized:
            res.set_option(_ffi.RM_OPT_SPECIALIZE, 0)
            a_el, a_draw, a_k = timed_run(1)
            ab = {"kernel": "rm_render_v5 (interpreter: LDS-staged records, accumulator machine), one frame in flight",
                  "value": W * H * args.steps / a_el / 1e6, "unit": "Mpixels/s", "kernel_ms": a_k,
                  "same_image": float(outs[0][..., :3].double().sum().item()) == checksum or args.camera != "still"}
            res.set_option(_ffi.RM_OPT_SPECIALIZE, args.specialize)
    if rank == 0:
        pixels = W * H * args.steps * (1 if tile else world)
        value = pixels / elapsed / 1e6
        # whole-frame algorithmic bytes over the whole draw (pre-pass + sort + march kernel launches)
        # launch duration of one frame's three kernels: with several frames in flight the event-bracketed draw also
        # contains the time spent queued behind the other frame, so the serial leg's figure is the one that applies
        draw_ms_launch = serial["draw_ms"] if serial is not None else draw_ms
        ach = BYTES_PER_PIXEL * W * (my_rows if tile else H) / (draw_ms_launch * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):
            try:
                with open(tfile) as f:
                    traffic = json.load(f).get("%s_%dx%d_%d" % (args.scene, W, H, args.max_iter))
            except Exception:
                traffic = None
        line = {
            "metric": "Mpixels/s at 1920x1080, 256-step march, 32-node SDF",
            "value": value, "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if tile else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%dx%d, %s (%d commands / %d words), %d max steps, 16 rays/px, RGBA32F out"
                                   % (W, H, args.scene, cc, len(words), args.max_iter),
                       "camera": args.camera, "kernel": args.kernel,
                       "specialized_kernel": specialized, "jit_compile_ms": jit_ms, "frames_in_flight": F,
                       "sharding": ("one frame tiled over ranks in interleaved 16-row strips%s, no collective"
                                    % ((" + host gather (%s)" % args.gather) if args.gather else "")) if tile
                       else ("frames over ranks, no collective" if world > 1 else "single GPU")},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": ("rm_render_v5_spec (hipRTC-compiled for this scene's structure)" if specialized else "rm_render_v5")
                                   + " = the dominant (march) kernel: kernel_ms; `achieved` divides the frame's "
                                   "bytes by draw_ms, the draw's three launches (pre-pass, sort, march)",
                         "kernel_ms": kernel_ms, "draw_ms": draw_ms,
                         "draw_ms_one_frame_in_flight": draw_ms_launch,
                         "frames_in_flight_note": ("%d frames in flight: a frame's event-bracketed draw_ms includes queueing behind "
                                                   "the other frame and exceeds ms_per_step; `achieved` uses the launch duration "
                                                   "of the same three kernels measured without overlap in this run "
                                                   "(one_frame_in_flight), when that leg ran" % F) if F > 1 else None,
                         "note": "algorithmic bytes = 16 B/pixel (one RGBA32F store); the kernel is FP32-VALU "
                                 "bound by ~3 orders of magnitude, see `compute`"},
            "checksum_rgb": checksum,
        }
        if world == 1:
            # the measured HBM-write ceiling of this GPU, same run: a fill kernel with 16 B/lane stores over 1 GiB
            # (SURVEY 8(d)); the declared peak above stays the spec figure
            try:
                fill = res.measure_write_bandwidth(1 << 30, 10)
                line["roofline"]["measured_write_GBps"] = fill
                line["roofline"]["frac_of_measured"] = ach / fill if fill > 0 else None
            except Exception as e:      # diagnostics only
                line["roofline"]["measured_write_GBps"] = None
                line["roofline"]["measured_write_error"] = str(e)
        if serial is not None:
            line["one_frame_in_flight"] = serial
        if ab is not None:
            line["ab_interpreter_kernel"] = ab
        if world == 1 and not args.no_cpu_baseline:
            base, cnt, (cw, ch) = cpu_baseline(args, (cc, words), still_events)
            line["cpu_baseline"] = base
            if args.camera == "still":
                # oracle counters -> algorithmic map_scene evaluations of the full-size frame
                scale = (W * H) / float(cw * ch)
                evals = (cnt["march_steps"] + cnt["normal_taps"]) * scale
                line["compute"] = {"map_scene_evals_per_frame": evals,
                                   "evals_per_s": evals / (kernel_ms * 1e-3),
                                   "note": "evaluations counted by the oracle on the CPU sample, scaled by pixel count"}
                line["compute"].update(valu_view(args, res, serial["kernel_ms"] if serial else kernel_ms, evals, words))
        print(json.dumps(line), flush=True)
    for c in ctxs:
        c.close()
    if shared is not None:
        shared.close((lambda: dist.barrier(group=gloo)) if world > 1 else None)
    if world > 1:
        dist.destroy_process_group()


def valu_view(args, res, march_ms, evals, words):
    """The honest bound: FP32 vector issue.  Instruction counts come from the committed PMC pass of this
    configuration (profiles/pmc_traffic.json), the duration from this run (march kernel, no other frame in flight)."""
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
    out["algorithmic_TFLOPs"] = evals * f_prog / (march_ms * 1e-3) / 1e12
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pmc = json.load(fh).get("%s_%dx%d_%d_valu" % (args.scene, args.width, args.height, args.max_iter))
    except (OSError, ValueError):
        pmc = None
    if pmc:
        simds = 4.0 * res.info(_ffi.RM_INFO_CU_COUNT)
        insts = pmc["SQ_INSTS_VALU"]
        flop_insts = pmc["SQ_INSTS_VALU_ADD_F32"] + pmc["SQ_INSTS_VALU_MUL_F32"] + 2.0 * pmc["SQ_INSTS_VALU_FMA_F32"] + pmc["SQ_INSTS_VALU_TRANS_F32"]
        ns_per_inst = march_ms * 1e6 * simds / insts
        out.update({
            "bound": "fp32 valu issue",
            "valu_insts_per_frame": insts,
            "ns_per_valu_inst_per_simd": ns_per_inst,
            "fastest_valu_issue_ns_per_simd": pmc["fastest_valu_issue_ns_per_simd"],
            "frac_of_valu_issue": pmc["fastest_valu_issue_ns_per_simd"] / ns_per_inst,
            "executed_fp32_TFLOPs": flop_insts * 64.0 * pmc["lane_occupancy"] / (march_ms * 1e-3) / 1e12,
            "peak_fp32_vector_TFLOPs": 157.3,
            "valu_note": "instruction counts: committed rocprofv3 PMC pass of this configuration; issue floor: fastest class of "
                         "profiles/r01_ubench_valu_issue_rates.txt (v_add / v_mul / v_fma; min / max / cmp / transcendental ops issue "
                         "slower); the 157.3 TFLOP/s peak assumes packed FMAs, the contract forbids fusing (DESIGN 2)",
        })
    return out


if __name__ == "__main__":
    main()
