"""bench.py --gpus N starts its own ranks (zkvm_amd/launch.py): the parent spawns N
fresh processes with the torchrun environment, relays rank 0's line, and fails if any rank fails -- also when the others
would wait for the dead one for ever.  CPU only: the children here are small scripts, and bench.py itself is run where it
must refuse (no GPU)."""
import io
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

RANK_SCRIPT = r"""
import json, os, sys
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
import torch, torch.distributed as dist
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
objs = [None] * world
dist.all_gather_object(objs, {"rank": rank, "pid": os.getpid()})
dist.barrier()
dist.destroy_process_group()
print(json.dumps({"rank": rank, "max": t.item(), "ranks": [o["rank"] for o in objs]}))
"""


def test_ranks_rendezvous_over_gloo_and_only_rank_zero_is_relayed():
    from zkvm_amd.launch import spawn_ranks
    out, err = io.StringIO(), io.StringIO()
    rc, codes = spawn_ranks([sys.executable, "-c", RANK_SCRIPT], 3, out=out, err=err, timeout=240)
    assert rc == 0 and codes == [0, 0, 0], err.getvalue()
    # (gloo prints its connection chatter on standard output; bench.py sends fd 1 to standard error for that reason)
    lines = [l for l in out.getvalue().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec == {"rank": 0, "max": 3.0, "ranks": [0, 1, 2]}
    other = [l for l in err.getvalue().splitlines() if l.startswith("[rank ") and "{" in l]
    assert sorted(l.split("]")[0] for l in other) == ["[rank 1", "[rank 2"]


def test_a_failing_rank_fails_the_launch_and_the_waiting_ranks_are_ended():
    """rank 1 dies with code 7 while rank 0 and rank 2 wait (as they would in a collective): the launch returns 7 after the
    grace period, nobody is left behind."""
    from zkvm_amd.launch import spawn_ranks
    script = ("import os, sys, time\n"
              "r = int(os.environ['RANK'])\n"
              "print('pid', os.getpid(), flush=True)\n"
              "if r == 1:\n    sys.exit(7)\n"
              "time.sleep(600)\n")
    out, err = io.StringIO(), io.StringIO()
    t0 = time.monotonic()
    rc, codes = spawn_ranks([sys.executable, "-c", script], 3, out=out, err=err, grace=1.0)
    dt = time.monotonic() - t0
    assert rc == 7 and codes[1] == 7 and codes[0] != 0 and codes[2] != 0
    assert dt < 30
    pids = [int(l.split()[-1]) for l in (out.getvalue() + err.getvalue()).splitlines() if "pid" in l]
    assert len(pids) == 3
    for pid in pids:                                                     # every process the launch started is gone
        with pytest.raises(ProcessLookupError):
            os.kill(pid, 0)


def test_a_rank_killed_by_a_signal_and_a_timeout_are_failures_too():
    from zkvm_amd.launch import spawn_ranks
    script = ("import os, signal, time\n"
              "if os.environ['RANK'] == '0':\n    os.kill(os.getpid(), signal.SIGKILL)\n"
              "time.sleep(600)\n")
    rc, codes = spawn_ranks([sys.executable, "-c", script], 2, out=io.StringIO(), err=io.StringIO(), grace=0.5)
    assert rc == 128 + 9 and codes[0] == -9
    rc, codes = spawn_ranks([sys.executable, "-c", "import time; time.sleep(600)"], 2, out=io.StringIO(), err=io.StringIO(), timeout=1.0)
    assert rc == 124 and all(c != 0 for c in codes)


def test_bench_gpus_2_without_a_launcher_spawns_ranks_and_fails_when_they_do():
    """The driver's N > 1 command read by the pattern of its N = 1 command (`python3 bench.py --gpus N ...`, no torchrun): on a
    box without a GPU both ranks refuse ("needs a GPU"), and so must the parent -- non-zero, nothing on standard output."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the ranks would run")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert r.stderr.count("needs a GPU") == 2 and "rank exit codes [1, 1]" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("try_rccl", [False, True])
def test_bench_gpus_2_on_one_gpu_prints_one_line_and_says_how_the_bitmaps_travelled(try_rccl):
    """`python3 bench.py --gpus 2` as the driver would type it, both ranks made to share the one GPU of the test box
    (ZKGPU_BENCH_SHARE_GPU=1: a rehearsal of the N > 1 code path, not a measurement): ONE JSON line on standard output, two
    ranks in it, every step's bitmap exchanged and checked on both.  With ZKGPU_BENCH_TRY_RCCL=1 the ranks try to bring RCCL
    up all the same -- it refuses two ranks on one device -- and must then agree to exchange over gloo and say why in the
    line, instead of dying: the shape of a node on which RCCL cannot start."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["ZKGPU_BENCH_SHARE_GPU"] = "1"
    if try_rccl:
        env["ZKGPU_BENCH_TRY_RCCL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--lean"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["value"] > 0
    assert len(d["config"]["per_rank"]) == 2
    ex = d["config"]["exchange"]
    if try_rccl:
        assert ex.startswith("gloo -- RCCL could not be brought up on rank") and d["config"].get("rccl") is None, ex
    else:
        assert ex == "gloo (ranks share one GPU)", ex
