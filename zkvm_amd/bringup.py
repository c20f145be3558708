"""Bounded bring-up of the RCCL communicator of a rank.

`ncclCommInitRank` is a collective that a process cannot cancel from inside: when one rank of a node never reaches it
(or RCCL's bootstrap between the ranks stalls) the others sit in it for ever, and a bench that meets its first 8-GPU
node that way would die at the driver's limit without a line.  Two bounds, both with the reason on standard error:

  * `probe(...)`: every rank tries the communicator FIRST IN A CHILD PROCESS (this module run as a program: a fresh
    interpreter that loads libzkgpu, makes a context on the rank's device, `zkgpu_comm_create` + one real all-gather,
    prints "ok").  The rank waits for its child for at most `timeout` seconds and kills it (by PID) when it is late.
    A child that stalls or fails costs the rank nothing: the ranks then agree -- over the control plane the caller
    hands in -- that RCCL cannot be brought up here, and the caller falls back (bench.py: the bitmaps travel over gloo
    and the line says so and why).  The unique id of the probe is made by rank 0's child, so not even
    `ncclGetUniqueId` runs in a process that has to survive.
  * `Watchdog`: around the in-process bring-up that follows a successful probe (and around anything else that must not
    wait for ever): if it is not cancelled within `seconds` the process prints why and leaves with `os._exit(code)` --
    never an exec: this process has initialised the GPU.  The launcher (zkvm_amd.launch / torch.distributed.run) then ends
    the other ranks and returns non-zero.

Only the standard library at import time; the child imports zkvm_amd.native (ctypes), never torch.
"""
from __future__ import annotations

import os
import select
import subprocess
import sys
import threading
import time
from typing import Callable, List, Optional

PROBE_PAYLOAD = b"zkgp"


class Watchdog:
    """`with Watchdog(120, "ncclCommInitRank + first all-gather"):` -- the block must end within the time or the
    process exits with `code` (default 3) after saying why on standard error."""

    def __init__(self, seconds: float, what: str, code: int = 3, rank: Optional[int] = None, _exit: Callable[[int], None] = os._exit):
        self.seconds, self.what, self.code, self.rank = float(seconds), what, int(code), rank
        self._exit = _exit
        self._done = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True, name="zkgpu-bringup-watchdog")

    def _run(self):
        if self._done.wait(self.seconds):
            return
        who = "" if self.rank is None else "rank %d: " % self.rank
        sys.stderr.write("[bringup] %s%s did not return within %.0f s -- leaving with exit code %d (a stalled collective "
                         "cannot be cancelled from inside the process)\n" % (who, self.what, self.seconds, self.code))
        sys.stderr.flush()
        self._exit(self.code)

    def start(self) -> "Watchdog":
        self._thread.start()
        return self

    def cancel(self) -> None:
        self._done.set()

    def __enter__(self) -> "Watchdog":
        return self.start()

    def __exit__(self, *exc) -> None:
        self.cancel()


def child_command(device: int, rank: int, world: int, uid_hex: str) -> List[str]:
    return [sys.executable, "-m", "zkvm_amd.bringup", str(device), str(rank), str(world), uid_hex]


def _read_line(stream, deadline: float) -> Optional[str]:
    """one line from a pipe, or None when the deadline passes first (the child may be stalled before printing)"""
    buf = b""
    fd = stream.fileno()
    while True:
        left = deadline - time.monotonic()
        if left <= 0:
            return None
        r, _, _ = select.select([fd], [], [], min(left, 0.5))
        if not r:
            continue
        c = os.read(fd, 4096)
        if not c:
            return buf.decode(errors="replace") if buf else ""
        buf += c
        if b"\n" in buf:
            return buf.decode(errors="replace")


def _end(p: subprocess.Popen) -> None:
    if p.poll() is None:
        p.kill()                     # (the exact PID this call started)
    try:
        p.wait(timeout=10)
    except Exception:                # noqa: BLE001
        pass


def probe(rank: int, world: int, device: int, timeout: float, broadcast: Callable[[bytes, int], bytes],
          gather: Callable[[object], list], command: Callable[[int, int, int, str], List[str]] = child_command,
          env: Optional[dict] = None) -> List[Optional[str]]:
    """Collective over the caller's control plane (`broadcast(bytes_on_rank0, n) -> bytes`, `gather(obj) -> [obj per rank]`).
    -> per rank: None (its child brought the communicator up and gathered) or the reason it did not."""
    t0 = time.monotonic()
    deadline = t0 + timeout
    child: Optional[subprocess.Popen] = None
    err: Optional[str] = None
    uid = b"\0" * 128
    penv = dict(os.environ if env is None else env)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    penv["PYTHONPATH"] = root + (os.pathsep + penv["PYTHONPATH"] if penv.get("PYTHONPATH") else "")
    tail = ""
    if rank == 0:
        try:
            child = subprocess.Popen(command(device, 0, world, "root"), stdout=subprocess.PIPE, stdin=subprocess.DEVNULL, env=penv)
            line = _read_line(child.stdout, min(deadline, t0 + max(30.0, timeout / 2)))
            if line is None:
                err = "the probe child did not produce a unique id (ncclGetUniqueId stalled?)"
            elif not line.startswith("uid ") or len(line.split()[1]) != 256:
                err = "the probe child failed before the unique id: %s" % (line.strip()[:300] or "no output")
            else:
                uid = bytes.fromhex(line.split()[1])
                tail = line.split("\n", 1)[1] if "\n" in line else ""
        except Exception as e:       # noqa: BLE001
            err = "%s: %s" % (type(e).__name__, str(e)[:300])
    uid = broadcast(uid, 128)
    have_uid = any(uid)
    if rank != 0 and have_uid:
        try:
            child = subprocess.Popen(command(device, rank, world, uid.hex()), stdout=subprocess.PIPE, stdin=subprocess.DEVNULL, env=penv)
        except Exception as e:       # noqa: BLE001
            err = "%s: %s" % (type(e).__name__, str(e)[:300])
    def answered(text: str) -> bool:
        # a whole line "ok" (not any text that happens to END in those letters: ADVICE r05)
        return any(l.strip() == "ok" for l in text.splitlines())

    if child is not None and err is None:
        while True:
            # checked BEFORE the next read: rank 0's first read may have brought "uid ...\nok\n" together, and waiting for more
            # output then means waiting for the child to exit -- a slow teardown would read as a stall
            if answered(tail):
                err = None
                try:
                    child.wait(timeout=max(1.0, min(3.0, deadline - time.monotonic())))
                except subprocess.TimeoutExpired:
                    pass             # (it answered; whatever it still does at exit is not the bench's problem: ended below)
                break
            line = _read_line(child.stdout, deadline)
            if line is None:
                err = "stalled: no answer from zkgpu_comm_create + the first all-gather within %.0f s (child killed)" % timeout
                break
            if line == "":
                code = child.wait()
                err = "probe child exited with code %d: %s" % (code, tail.strip()[-300:] or "no output")
                break
            tail += line
    elif err is None and not have_uid:
        err = "no unique id (rank 0's probe failed)"
    if child is not None:
        _end(child)
    return gather(err)


def _child_main(argv: List[str]) -> int:
    device, rank, world, uid_hex = int(argv[0]), int(argv[1]), int(argv[2]), argv[3]
    from zkvm_amd.native import Comm, Context
    if uid_hex == "root":
        uid = Comm.unique_id()
        sys.stdout.write("uid %s\n" % uid.hex())
        sys.stdout.flush()
    else:
        uid = bytes.fromhex(uid_hex)
    ctx = Context(device)
    comm = Comm(ctx, rank, world, uid)
    got = comm.allgather(PROBE_PAYLOAD + rank.to_bytes(4, "little"))
    want = b"".join(PROBE_PAYLOAD + r.to_bytes(4, "little") for r in range(world))
    if got != want:
        sys.stdout.write("the first all-gather returned the wrong bytes\n")
        return 1
    sys.stdout.write("ok\n")
    sys.stdout.flush()
    comm.close()
    ctx.close()
    return 0


if __name__ == "__main__":
    try:
        code = _child_main(sys.argv[1:])
    except Exception as e:           # noqa: BLE001
        sys.stdout.write("%s: %s\n" % (type(e).__name__, str(e)[:300]))
        code = 1
    sys.stdout.flush()
    os._exit(code)                   # (no interpreter teardown behind a communicator that may be half made)
