"""bench.py --gpus N starts its own ranks (zkvm_amd/launch.py): the parent spawns N
fresh processes with the torchrun environment, relays rank 0's line, and fails if any rank fails -- also when the others
would wait for the dead one for ever; and the communicator's bring-up at N > 1 is bounded (zkvm_amd/bringup.py).  The CPU
tests' children are small scripts, and bench.py itself is run where it must refuse (no GPU); the `gpu` tests run bench.py
--gpus 2 on the one GPU of the test box, also with RCCL refusing and with RCCL stalling."""
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


# ---- the bounded bring-up of the communicator (zkvm_amd/bringup.py) -----------------------------------
# CPU: the probe's children are stand-in programs (the real child is `python -m zkvm_amd.bringup`, run in the GPU tests).
FAKE_CHILD = r"""
import sys, time
mode, rank, world, uid = sys.argv[1], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
if uid == "root":
    if mode == "no-uid":
        time.sleep(600)
    print("uid " + "ab" * 128, flush=True)
if mode == "ok":
    print("ok", flush=True)
elif mode == "stall-1" and rank == 1:
    time.sleep(600)
elif mode == "stall-1":
    print("ok", flush=True)
elif mode == "refuse":
    print("ZkGpuError: ncclCommInitRank: invalid usage", flush=True); sys.exit(1)
elif mode == "slow-exit":
    # the id and the answer leave in ONE write on rank 0, then a teardown that never ends (ncclCommDestroy hanging)
    sys.stdout.write("ok\n"); sys.stdout.flush(); time.sleep(600)
elif mode == "not-ok":
    print("the collective gave back a book", flush=True); sys.exit(0)        # ends in the letters "ok", is not the answer
"""

PROBE_RANK = r"""
import json, os, sys, time
sys.path.insert(0, %r)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
import torch, torch.distributed as dist
dist.init_process_group("gloo", rank=rank, world_size=world)
def bcast(b, n):
    t = torch.frombuffer(bytearray(b if rank == 0 else bytes(n)), dtype=torch.uint8)
    dist.broadcast(t, src=0)
    return bytes(t.numpy().tobytes())
def gather(o):
    out = [None] * world
    dist.all_gather_object(out, o)
    return out
from zkvm_amd import bringup
mode, timeout = os.environ["FAKE_MODE"], float(os.environ["FAKE_TIMEOUT"])
t0 = time.monotonic()
errs = bringup.probe(rank, world, 0, timeout, bcast, gather,
                     command=lambda dev, r, w, uid: [sys.executable, "-c", os.environ["FAKE_CHILD"], mode, str(dev), str(r), str(w), uid])
dist.barrier()
dist.destroy_process_group()
print(json.dumps({"rank": rank, "errs": errs, "s": time.monotonic() - t0}))
""" % ROOT


def _gone(pid):
    """no such process, or a dead one that its new parent (the container's init) has not reaped yet"""
    try:
        with open("/proc/%d/stat" % pid) as f:
            return f.read().rsplit(")", 1)[1].split()[0] == "Z"
    except (FileNotFoundError, ProcessLookupError):
        return True


def _probe_world(mode, timeout, world=3):
    from zkvm_amd.launch import spawn_ranks
    env = dict(os.environ, FAKE_MODE=mode, FAKE_TIMEOUT=str(timeout), FAKE_CHILD=FAKE_CHILD)
    out, err = io.StringIO(), io.StringIO()
    rc, codes = spawn_ranks([sys.executable, "-c", PROBE_RANK], world, env=env, out=out, err=err, timeout=240)
    assert rc == 0 and codes == [0] * world, err.getvalue()[-2000:]
    return json.loads([l for l in out.getvalue().splitlines() if l.startswith("{")][0])


def test_probe_children_all_succeed():
    rec = _probe_world("ok", 60)
    assert rec["errs"] == [None, None, None]


def test_probe_with_one_stalled_child_ends_in_time_on_every_rank_and_names_the_rank():
    """rank 1's child never answers (what ncclCommInitRank does when a peer cannot be reached): every rank is back from the
    probe after the timeout -- nobody waits for the stalled one -- with the SAME per-rank verdicts, the child is killed."""
    rec = _probe_world("stall-1", 4)
    assert rec["errs"][0] is None and rec["errs"][2] is None
    assert rec["errs"][1].startswith("stalled: no answer") and rec["s"] < 30


def test_probe_child_that_refuses_or_never_makes_an_id():
    rec = _probe_world("refuse", 30, world=2)
    assert all(e and "invalid usage" in e for e in rec["errs"]), rec
    rec = _probe_world("no-uid", 3, world=2)
    assert "did not produce a unique id" in rec["errs"][0] and "no unique id" in rec["errs"][1] and rec["s"] < 60


def test_probe_answer_is_a_whole_line_and_a_slow_teardown_after_it_is_not_a_stall():
    """(1) a child that has answered "ok" and then hangs in its teardown is a success on every rank, rank 0
    included -- whose first read may hold the id and the answer together -- and the probe is back long before the timeout;
    (2) text that merely ENDS in the letters "ok" is not the answer."""
    rec = _probe_world("slow-exit", 20, world=2)
    assert rec["errs"] == [None, None] and rec["s"] < 15, rec
    rec = _probe_world("not-ok", 20, world=2)
    assert all(e and "exited with code 0" in e for e in rec["errs"]), rec


def test_watchdog_ends_a_rank_that_stalls_and_the_launcher_ends_the_others():
    """The in-process bound: rank 1 'stalls in ncclCommInitRank' (sleeps) under a 1.5 s watchdog -> it says why and leaves
    with code 3; the launcher gives the waiting ranks their grace, ends them, returns 3; nobody is left.  A watchdog that is
    cancelled in time does nothing."""
    from zkvm_amd.bringup import Watchdog
    from zkvm_amd.launch import spawn_ranks
    fired = []
    with Watchdog(0.3, "nothing", _exit=fired.append):
        pass
    time.sleep(0.6)
    assert fired == []
    w = Watchdog(0.2, "a stand-in", code=5, rank=7, _exit=fired.append).start()
    time.sleep(0.8)
    assert fired == [5]
    w.cancel()
    script = ("import os, sys, time\nsys.path.insert(0, %r)\n"
              "from zkvm_amd.bringup import Watchdog\n"
              "r = int(os.environ['RANK'])\nprint('pid', os.getpid(), flush=True)\n"
              "with Watchdog(1.5 if r == 1 else 600, 'zkgpu_comm_create', rank=r):\n    time.sleep(600)\n" % ROOT)
    out, err = io.StringIO(), io.StringIO()
    t0 = time.monotonic()
    rc, codes = spawn_ranks([sys.executable, "-c", script], 3, out=out, err=err, grace=1.0)
    assert rc == 3 and codes[1] == 3 and time.monotonic() - t0 < 30
    pids = [int(l.split()[-1]) for l in (out.getvalue() + err.getvalue()).splitlines() if "pid" in l]
    assert len(pids) == 3
    for pid in pids:
        with pytest.raises(ProcessLookupError):
            os.kill(pid, 0)


def test_a_terminated_launcher_takes_its_ranks_and_their_children_with_it():
    """`timeout 600 python bench.py --gpus 8` sends SIGTERM to the PARENT; the ranks (and a helper a rank
    started) must not stay behind holding GPUs.  Every rank is the leader of its own session and is ended by process group."""
    parent = ("import sys\nsys.path.insert(0, %r)\nfrom zkvm_amd.launch import spawn_ranks\n"
              "child = 'import os, subprocess, sys, time\\n"
              "h = subprocess.Popen([sys.executable, \"-c\", \"import time; time.sleep(600)\"])\\n"
              "print(\"pid\", os.getpid(), h.pid, flush=True)\\ntime.sleep(600)\\n'\n"
              "rc, codes = spawn_ranks([sys.executable, '-c', child], 2)\nsys.exit(rc)\n" % ROOT)
    p = subprocess.Popen([sys.executable, "-c", parent], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    pids, t0 = [], time.monotonic()
    lines = []
    while len(pids) < 4 and time.monotonic() - t0 < 60:       # rank 0's line arrives on stdout, rank 1's on stderr: read both
        import select
        r, _, _ = select.select([p.stdout, p.stderr], [], [], 1.0)
        for s in r:
            l = s.readline()
            lines.append(l)
            if "pid" in l:
                pids += [int(x) for x in l.split("pid")[1].split()]
    assert len(pids) == 4, lines
    p.terminate()
    time.sleep(0.3)
    try:
        p.terminate()                     # a second SIGTERM, as `timeout` and drivers send: it must not abort the clean-up
    except ProcessLookupError:
        pass
    assert p.wait(timeout=30) == 143
    time.sleep(0.5)
    for pid in pids:
        assert _gone(pid), pid


@pytest.mark.gpu
@pytest.mark.parametrize("probe", [True, False])
def test_bench_gpus_2_when_the_communicator_stalls_instead_of_refusing(probe):
    """The rehearsal VERDICT r04 asked for: RCCL's bring-up STALLS on rank 1 (ZKGPU_TEST_COMM_STALL=init:1 -- the library's
    test hook makes ncclCommInitRank of that rank sleep for ever; rank 0 then waits in RCCL's own bootstrap).  With the probe
    (the default) the stall happens in child processes, is bounded, and the run falls back to gloo and prints its ONE line
    naming the stalled rank; with ZKGPU_BENCH_COMM_PROBE=0 the stall hits the ranks themselves: the watchdog ends them with
    code 3 and the reason on standard error, the launcher returns non-zero within its grace, nothing on standard output --
    and in neither case is a process left behind."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(ZKGPU_BENCH_SHARE_GPU="1", ZKGPU_BENCH_TRY_RCCL="1", ZKGPU_TEST_HOOKS="1", ZKGPU_TEST_COMM_STALL="init:1")
    if not probe:
        env["ZKGPU_BENCH_COMM_PROBE"] = "0"
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--lean",
                        "--comm-timeout", "25"], env=env, capture_output=True, text=True, timeout=900)
    dt = time.monotonic() - t0
    if probe:
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, r.stdout
        d = json.loads(lines[0])
        ex = d["config"]["exchange"]
        assert ex.startswith("gloo -- RCCL could not be brought up on rank") and "stalled" in ex, ex
        assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["bringup"]["probe_s"] < 60
    else:
        assert r.returncode == 3, (r.returncode, r.stderr[-3000:])
        assert r.stdout.strip() == ""
        assert "did not return within 25 s" in r.stderr and "leaving with exit code 3" in r.stderr
        assert dt < 400


@pytest.mark.gpu
def test_bench_under_torch_distributed_run_as_the_driver_launches_it():
    """The driver's N > 1 command verbatim -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` -- with both ranks sharing the test box's one GPU
    (ZKGPU_BENCH_SHARE_GPU=1, ZKGPU_BENCH_TRY_RCCL=1: RCCL is tried in the probe children, refuses two ranks on one device,
    and the ranks agree on gloo): ONE JSON line on standard output from rank 0, two ranks in it, the fallback named."""
    from zkvm_amd.launch import free_port
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(ZKGPU_BENCH_SHARE_GPU="1", ZKGPU_BENCH_TRY_RCCL="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--lean"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["value"] > 0 and len(d["config"]["per_rank"]) == 2
    assert d["config"]["exchange"].startswith("gloo -- RCCL could not be brought up on rank"), d["config"]["exchange"]
    assert d["config"]["bringup"]["probe_s"] < 120
