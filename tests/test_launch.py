"""The rank launcher behind `bench.py --gpus N` (ray-marching_amd/launch.py): one fresh process per rank with the
environment torch.distributed.run would set, rank 0's stdout captured, failures propagated.  CPU only (gloo)."""
import json
import os
import subprocess
import sys
import textwrap
import time

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
