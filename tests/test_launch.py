"""The rank launcher behind `bench.py --gpus N` (ray-marching_amd/launch.py): one fresh process per rank with the
environment torch.distributed.run would set, rank 0's stdout captured, failures propagated.  CPU (gloo); the two tests
marked gpu run bench.py itself, as one command, at N = 1 and with two ranks sharing one GPU."""
import json
import os
import subprocess
import sys
import textwrap
import time

import pytest

from ray_marching_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ranks_form_a_process_group_and_rank0_reports(tmp_path):
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent("""
        import json, os
        import torch
        import torch.distributed as dist
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["LOCAL_RANK"]) == rank
        dist.init_process_group(backend="gloo")
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        print("noise from rank %d" % rank)
        if rank == 0:
            print(json.dumps({"world": world, "max": float(t[0])}))
        dist.destroy_process_group()
    """))
    rc, out = launch.run_ranks(2, [sys.executable, str(script)], timeout=120)
    assert rc == 0
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert json.loads(lines[-1]) == {"world": 2, "max": 2.0}
    assert "noise from rank 1" not in out          # only rank 0's stdout is the result channel


def test_a_failing_rank_stops_the_others(tmp_path):
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent("""
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(600)      # would wait for a rendezvous that can no longer happen
    """))
    t0 = time.monotonic()
    rc, _ = launch.run_ranks(3, [sys.executable, str(script)], timeout=120, echo_stderr=False)
    assert rc == 7
    assert time.monotonic() - t0 < 60


def test_output_of_a_failing_rank0_is_kept_and_children_are_reaped(tmp_path):
    """Rank 0 prints its diagnostics and exits non-zero while another rank still sleeps: everything rank 0 wrote comes back
    (communicate() keeps what it read across its timeouts; a later stdout.read() would not), and no child is left behind."""
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent("""
        import os, sys, time
        if os.environ["RANK"] == "0":
            print("first line of rank 0", flush=True)
            time.sleep(1.0)            # several communicate() timeouts pass in the parent
            print("last words of rank 0", flush=True)
            sys.exit(3)
        time.sleep(600)
    """))
    rc, out = launch.run_ranks(2, [sys.executable, str(script)], timeout=120, echo_stderr=False)
    assert rc == 3
    assert "first line of rank 0" in out and "last words of rank 0" in out
    # nothing of ours is left defunct or running
    kids = subprocess.run(["ps", "--ppid", str(os.getpid()), "-o", "pid=,stat=,args="], capture_output=True, text=True).stdout
    assert "rank.py" not in kids, kids


def test_timing_group_falls_back_to_gloo_on_every_rank(tmp_path):
    """bench.py's barrier group (launch.init_timing_group): asked for RCCL where RCCL cannot come up (no GPU here; on a GPU box
    the test forces the failure), every rank ends up on the same gloo group, in the same process, and the group works."""
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys
        sys.path.insert(0, %r)
        import torch
        import torch.distributed as dist
        from ray_marching_amd import launch
        rank = int(os.environ["RANK"])
        if rank == 1:
            os.environ["RM_BENCH_FORCE_NCCL_FAILURE"] = "1"      # one rank failing is enough: all must agree
        group, backend, device, world, why = launch.init_timing_group("nccl", 0, timeout_s=60)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        dist.barrier(group=group)
        if rank == 0:
            print(json.dumps({"backend": backend, "world": world, "max": float(t[0]), "why": why, "group_is_default": group is None}))
        dist.destroy_process_group()
    """ % ROOT))
    rc, out = launch.run_ranks(2, [sys.executable, str(script)], timeout=240)
    assert rc == 0, out
    got = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert got["backend"] == "gloo" and got["world"] == 2 and got["max"] == 2.0 and got["group_is_default"] is True
    assert got["why"]


def test_timing_group_gloo_when_asked(tmp_path):
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys
        sys.path.insert(0, %r)
        import torch.distributed as dist
        from ray_marching_amd import launch
        group, backend, device, world, why = launch.init_timing_group("gloo", 0)
        dist.barrier(group=group)
        if os.environ["RANK"] == "0":
            print(json.dumps([backend, device, world, why]))
        dist.destroy_process_group()
    """ % ROOT))
    rc, out = launch.run_ranks(2, [sys.executable, str(script)], timeout=240)
    assert rc == 0
    assert json.loads([l for l in out.splitlines() if l.startswith("[")][-1]) == ["gloo", "cpu", 2, None]


def test_timeout_is_reported(tmp_path):
    script = tmp_path / "rank.py"
    script.write_text("import time\ntime.sleep(600)\n")
    rc, _ = launch.run_ranks(2, [sys.executable, str(script)], timeout=2, echo_stderr=False)
    assert rc == 124


def test_rank_env_keeps_dmabuf_ipc_setting():
    env = launch.rank_env(3, 8, 29500, base={"HSA_ENABLE_IPC_MODE_LEGACY": "0", "PATH": "/bin"})
    assert env["RANK"] == "3" and env["LOCAL_RANK"] == "3" and env["WORLD_SIZE"] == "8"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29500"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_self_launches_and_fails_loudly_without_a_gpu():
    """`bench.py --gpus 2` with no launcher around it starts its own ranks; on a GPU-less box every rank refuses to run
    (there is no CPU fallback) and the parent reports the failure instead of hanging or printing a number."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is visible")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "needs a GPU" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_bench_under_a_launcher_checks_world_size():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=3" in p.stderr


def test_bench_mode_defaults():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse(["--gpus", "8"])
    assert a.mode == "auto" and a.camera == "still" and a.frames_in_flight == 4
    assert bench.parse([]).gpus == 1


@pytest.mark.gpu
def test_bench_two_ranks_tile_one_frame_and_gather_it():
    """`bench.py --gpus 2` as ONE command on a one-GPU box (both ranks on GPU 0, gloo for the timing barrier): the
    north-star layout -- frame tiled over the ranks in interleaved strips, host-side gather into one shared-memory frame --
    end to end, and the gathered frame is the single-GPU frame byte for byte."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--all-ranks-on-device0", "--dist-backend", "gloo",
                        "--steps", "6", "--warmup", "2", "--width", "640", "--height", "360"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == 6 and line["value"] > 0
    assert "interleaved 16-row strips" in line["config"]["sharding"] and line["config"]["specialized_kernel"] is True
    assert line["end_to_end"]["gathered_frame_identical_to_one_gpu_render"] is True
    assert line["end_to_end"]["value"] > 0 and line["frames_sharded"]["scaling"] == "weak"
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1 and line["roofline"]["kernel_ms"] > 0
    assert line["vs_baseline"] is None and line["unit"] == "Mpixels/s"
    assert line["config"]["barrier_backend"] == "gloo" and line["config"]["barrier_group_world_size"] == 2


@pytest.mark.gpu
def test_bench_one_gpu_line_has_every_contract_key():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--cpu-sample-div", "8"],
                       capture_output=True, text=True, timeout=600,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["scaling"] is None and line["dtype"] == "f32" and line["data"] == "synthetic"
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in line["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in line["cpu_baseline"], key
    assert line["cpu_baseline"]["kind"] == "port"
    assert line["one_frame_in_flight"]["kernel_ms"] <= line["one_frame_in_flight"]["draw_ms"]
    assert line["ab_interpreter_kernel"]["same_image"] is True and line["end_to_end"]["gathered_frame_identical_to_one_gpu_render"] is True
    # the line certifies its own frames against the oracle (with --cpu-sample-div: one reduced frame; the default run compares
    # every in-flight buffer of the timed loop)
    assert line["parity"]["vs"] == "oracle" and line["parity"]["pixels_differing"] == 0 and line["parity"]["max_abs_diff"] == 0.0
    assert line["parity"]["frames_checked"] >= 1
    assert 0.3 < line["compute"]["useful_lane_occupancy"] <= 1.0
