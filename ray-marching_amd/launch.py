"""One process per GPU, started from a parent that never touches the GPU.

`bench.py --gpus N` (and anything else that wants N ranks on one node) calls run_ranks(): it
starts N fresh Python processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
in their environment -- the variables torch.distributed.run would set -- captures rank 0's
stdout and waits for all of them.  The parent makes no HIP or torch.cuda call before or after
(a process that has initialised the GPU must not exec or fork GPU children on this pool), and
children are only ever signalled by the exact PIDs started here.

There is no data-path collective in this renderer (tiles and frames are independent, SURVEY.md
8(e)); the process group the ranks form only carries the timing barrier and a MAX reduction.
"""
import os
import socket
import subprocess
import sys
import time


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update({
        "RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
        "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
        # dmabuf IPC only on this pool (RCCL / shared device memory between processes fail without it)
        "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
    })
    return env


def run_ranks(world, argv, timeout=None, env=None, echo_stderr=True):
    """Start `world` copies of `argv` (a full command line), rank r with RANK=LOCAL_RANK=r.
    Returns (exit_code, rank0_stdout).  exit_code is the first non-zero code of any rank, else 0;
    when one rank fails the others are terminated (by PID) instead of waiting for a rendezvous
    that can no longer complete."""
    assert world >= 1
    port = free_port()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen(
            list(argv), env=rank_env(r, world, port, env),
            stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
            stderr=None if echo_stderr else subprocess.DEVNULL, text=True))
    deadline = None if timeout is None else time.monotonic() + timeout
    rc = 0
    out0 = ""
    try:
        # rank 0's pipe is drained by communicate(); the others write nothing we keep
        pending = set(range(world))
        while pending:
            for r in sorted(pending):
                p = procs[r]
                try:
                    if r == 0:
                        o, _ = p.communicate(timeout=0.2)
                        out0 = o or ""
                    else:
                        p.wait(timeout=0.05)
                except subprocess.TimeoutExpired:
                    continue
                pending.discard(r)
                if p.returncode != 0 and rc == 0:
                    rc = p.returncode
            if rc != 0 and pending:
                break
            if deadline is not None and time.monotonic() > deadline and pending:
                rc = 124
                break
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs[1:]:
            if p.poll() is None:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()            # reap it: a killed child stays defunct until someone waits for it
        # Rank 0 wrote to a pipe: communicate() keeps what it has read across a timeout, so only another communicate() returns
        # ALL of rank 0's output (stdout.read() would return what arrived after the last timeout -- a failed rank's
        # diagnostics would be truncated or lost).
        if procs[0].stdout is not None and not out0:
            try:
                out0 = procs[0].communicate(timeout=10)[0] or ""
            except subprocess.TimeoutExpired:
                procs[0].kill()
                try:
                    out0 = procs[0].communicate(timeout=10)[0] or ""
                except (subprocess.TimeoutExpired, OSError, ValueError):
                    pass
            except (OSError, ValueError):
                pass
        if procs[0].poll() is None:
            procs[0].kill()
            procs[0].wait()
    return rc, out0


def python_argv(script, args):
    return [sys.executable, script] + list(args)


def init_timing_group(backend="nccl", local_rank=0, timeout_s=120.0):
    """The process group that carries bench.py's timing barrier and its MAX reduction (nothing else: tiles and frames are
    independent, there is no collective on the data path).  Call it before any other GPU work.

    The DEFAULT group is always gloo (TCP between the ranks of one node: it has nothing to do with the GPUs and cannot be
    lost to them).  With backend == "nccl" (RCCL on ROCm) a second group is created on top and probed with one tiny
    all-reduce; the ranks then agree over gloo whether EVERY rank got through.  If any did not -- RCCL missing, communicator
    set-up failing on this node, no GPU -- all ranks keep gloo, in the same process (never a re-exec).
    Returns (group or None for the default group, backend name actually used, device for the reduction tensors, world size
    the group reports, reason for a fallback or None)."""
    import datetime

    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=max(60.0, timeout_s)))
    world = dist.get_world_size()
    if backend != "nccl":
        return None, "gloo", "cpu", world, None
    ok, why, group = 1, None, None
    try:
        if os.environ.get("RM_BENCH_FORCE_NCCL_FAILURE"):
            raise RuntimeError("RM_BENCH_FORCE_NCCL_FAILURE is set")
        if not (dist.is_nccl_available() and torch.cuda.is_available()):
            raise RuntimeError("RCCL or a GPU is not available in this process")
        group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=timeout_s))
        t = torch.ones(1, device=torch.device("cuda", local_rank))
        dist.all_reduce(t, group=group)
        torch.cuda.synchronize()
        if int(t.item()) != world:
            raise RuntimeError("probe all-reduce returned %r, expected %d" % (t.item(), world))
    except Exception as e:  # noqa: BLE001 -- whatever went wrong, the barrier must not cost the run
        ok, why = 0, "%s: %s" % (type(e).__name__, e)
    flag = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)    # over gloo: every rank learns whether every rank got RCCL up
    if int(flag.item()) == 1:
        return group, "nccl", "cuda", dist.get_world_size(group), None
    if group is not None and ok:
        why = "another rank could not bring RCCL up"
    return None, "gloo", "cpu", world, why or "another rank could not bring RCCL up"
