"""Supervision of self-launched ranks (dgppo_amd/launch.py, used by `bench.py --gpus N` and `train.py --gpus N`): a dead rank
must stop the whole job within seconds with a diagnosable message — never leave the survivors waiting in a collective."""
import io
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _script(tmp_path, body: str) -> str:
    p = tmp_path / "rank_prog.py"
    p.write_text("import os, sys, time\nsys.path.insert(0, %r)\n" % ROOT + textwrap.dedent(body))
    return str(p)


def test_two_ranks_rendezvous_through_the_file_store_and_pass_the_allreduce_selfcheck(tmp_path):
    from dgppo_amd import launch
    prog = _script(tmp_path, """
        import torch
        from dgppo_amd import dist as D
        rank, world = D.init_control_plane(timeout_s=60)
        assert os.environ.get("DGPPO_RDZV_FILE")
        allreduce, close = D.make_allreduce(world, "gloo")
        D.selfcheck_allreduce(allreduce, rank, world, torch.device("cpu"))
        print("progress from rank", rank, file=sys.stderr, flush=True)
        D.barrier(world)
        if rank == 0:
            print('{"ok": true}', flush=True)
        D.shutdown(world)
    """)
    out, err = io.StringIO(), io.StringIO()
    rc = launch.spawn_ranks(prog, [], 2, str(tmp_path / "logs"), stall_seconds=120, out=out, err=err)
    assert rc == 0, err.getvalue()
    assert out.getvalue().strip() == '{"ok": true}'
    assert "progress from rank 0" in err.getvalue()                     # rank 0's stderr is relayed
    assert "progress from rank 1" in open(tmp_path / "logs" / "rank1.err").read()


def test_a_dead_rank_stops_the_job_within_seconds(tmp_path):
    from dgppo_amd import launch
    prog = _script(tmp_path, """
        from dgppo_amd import dist as D
        rank, world = D.init_control_plane(timeout_s=600)
        if rank == 1:
            print("rank 1 about to die", file=sys.stderr, flush=True)
            os._exit(3)
        D.barrier(world)            # rank 0 would wait here for the gloo timeout
        time.sleep(600)
    """)
    out, err = io.StringIO(), io.StringIO()
    t0 = time.time()
    rc = launch.spawn_ranks(prog, [], 2, str(tmp_path / "logs"), stall_seconds=300, out=out, err=err)
    dt = time.time() - t0
    assert rc == 1
    assert dt < 60, f"the launcher took {dt:.0f} s to notice a dead rank"
    msg = err.getvalue()
    assert "(1, 3)" in msg and "rank 1 about to die" in msg, msg


def test_a_silent_hang_trips_the_stall_deadline(tmp_path):
    from dgppo_amd import launch
    prog = _script(tmp_path, """
        print("started", file=sys.stderr, flush=True)
        time.sleep(600)
    """)
    out, err = io.StringIO(), io.StringIO()
    t0 = time.time()
    rc = launch.spawn_ranks(prog, [], 2, str(tmp_path / "logs"), stall_seconds=3, out=out, err=err)
    assert rc == 1 and time.time() - t0 < 30
    assert "no rank wrote anything" in err.getvalue()


def test_bench_py_multi_gpu_launch_fails_fast_and_loudly_without_gpus(tmp_path):
    """here (no GPU) every rank of `bench.py --gpus 2` refuses to start; the parent must return non-zero quickly and say why"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: covered by the -m gpu fault-injection test")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert time.time() - t0 < 120
    assert "[launch] FAILED" in r.stderr and "GPU(s) visible" in r.stderr, r.stderr[-2000:]
