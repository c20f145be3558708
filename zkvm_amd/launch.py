"""One process per GPU, started by the program itself (bench.py --gpus N without torchrun).

The parent NEVER touches the GPU: it must spawn its ranks before anything in it initialises HIP (a process that has
done so must not be replaced or forked into workers on this pool), so this module imports nothing but the standard
library and `spawn_ranks` is called before torch is imported.  Every rank is a FRESH interpreter running the same
command with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set -- exactly the environment
`python -m torch.distributed.run --nproc-per-node N` would give it, so a rank cannot tell the two launchers apart.

Contract:
  * rank 0's standard output is relayed (that is where the ONE JSON line of the bench is printed); the other ranks'
    standard output goes to the parent's standard error; standard error is inherited;
  * the return code is 0 only if EVERY rank exited 0: the first rank that fails decides the code, and the ranks that are
    still running then -- they would wait for the dead one in their next collective for ever -- get `grace` seconds to
    finish by themselves, a SIGTERM, and two seconds later a SIGKILL (by PID: processes this call started, nothing else);
  * `timeout` bounds the whole run the same way (code 124);
  * however the call is left -- KeyboardInterrupt, a SIGTERM to the parent (code 143), an exception -- the ranks it
    started are ended on the way out (SIGTERM, then SIGKILL, to each rank's own process group: every rank is started
    as the leader of a new session, so helpers a rank started go with it).
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import threading
import time
from typing import Dict, List, Optional, Sequence, Tuple


def free_port() -> int:
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_environment(base: Dict[str, str], rank: int, world: int, port: int) -> Dict[str, str]:
    env = dict(base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ZKGPU_LAUNCHED_BY="zkvm_amd.launch")
    # the host driver of this pool supports dmabuf IPC only (RCCL between processes fails without it)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def spawn_ranks(command: Sequence[str], world: int, env: Optional[Dict[str, str]] = None, grace: float = 15.0,
                timeout: Optional[float] = None, out=None, err=None) -> Tuple[int, List[int]]:
    """Runs `command` as `world` processes.  -> (return code, per-rank exit codes).  `out` / `err`: where rank 0's
    standard output / the other ranks' standard output is written (text streams; default sys.stdout / sys.stderr)."""
    if world < 1:
        raise ValueError("world must be at least 1")
    out = out if out is not None else sys.stdout
    err = err if err is not None else sys.stderr
    base = dict(os.environ if env is None else env)
    port = free_port()
    procs: List[subprocess.Popen] = []
    pumps: List[threading.Thread] = []

    def pump(stream, sink, prefix):
        for line in iter(stream.readline, ""):
            sink.write(prefix + line)
            sink.flush()
        stream.close()

    def end_all(sig):
        """the processes THIS call started (and, each being the leader of a session of its own, whatever helpers they
        started: a rank's probe child, a profiler's wrapper) -- by PID / process group, never by pattern"""
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig)
                except (ProcessLookupError, PermissionError):
                    try:
                        p.send_signal(sig)
                    except ProcessLookupError:
                        pass

    codes: List[Optional[int]] = [None] * world
    first_bad: Optional[int] = None
    # a SIGTERM to the parent (`timeout 600 python bench.py --gpus 8`, the driver's limit) must not leave the ranks behind,
    # holding the GPUs and waiting in a collective: it becomes an exception here and the `finally` below ends them
    # (the handler only SETS A FLAG that the poll loop reads -- an exception raised from a signal handler lands at
    # whatever bytecode is running, and a second SIGTERM, which `timeout` and drivers do send, used to land inside the clean-up
    # and abort it with the ranks still alive; during the clean-up further SIGTERMs are ignored)
    class _Terminated(BaseException):
        pass

    term_seen = threading.Event()

    def on_term(signum, frame):
        term_seen.set()

    old_term = None
    if threading.current_thread() is threading.main_thread():
        try:
            old_term = signal.signal(signal.SIGTERM, on_term)
        except (ValueError, OSError):
            old_term = None
    try:
        for r in range(world):
            if term_seen.is_set():
                raise _Terminated()
            p = subprocess.Popen(list(command), env=rank_environment(base, r, world, port), stdout=subprocess.PIPE,
                                 stdin=subprocess.DEVNULL, text=True, bufsize=1, start_new_session=True)
            procs.append(p)
            t = threading.Thread(target=pump, args=(p.stdout, out if r == 0 else err, "" if r == 0 else "[rank %d] " % r), daemon=True)
            t.start()
            pumps.append(t)

        t_start = time.monotonic()
        t_bad: Optional[float] = None
        sent_term = False
        while True:
            if term_seen.is_set():
                raise _Terminated()
            running = 0
            for r, p in enumerate(procs):
                if codes[r] is None:
                    c = p.poll()
                    if c is None:
                        running += 1
                        continue
                    codes[r] = c
                    if c != 0 and first_bad is None:
                        first_bad = c if c > 0 else 128 + (-c)          # killed by a signal: the shell's convention
                        t_bad = time.monotonic()
                        err.write("[launch] rank %d exited with code %d; the other ranks have %.0f s to finish\n" % (r, c, grace))
                        err.flush()
            if running == 0:
                break
            now = time.monotonic()
            if timeout is not None and first_bad is None and now - t_start > timeout:
                first_bad, t_bad, grace = 124, now, 0.0
                err.write("[launch] the run exceeded %.0f s\n" % timeout)
                err.flush()
            if t_bad is not None:
                if not sent_term and now - t_bad > grace:
                    end_all(signal.SIGTERM)
                    sent_term = True
                    t_bad = now
                elif sent_term and now - t_bad > 2.0:
                    end_all(signal.SIGKILL)
            time.sleep(0.05)
    except _Terminated:
        err.write("[launch] terminated; ending the ranks\n")
        err.flush()
        first_bad = first_bad or 143
    finally:
        # whatever way this is left (an exception above, KeyboardInterrupt, SIGTERM, a failed Popen): nobody stays behind
        if old_term is not None:
            try:
                signal.signal(signal.SIGTERM, signal.SIG_IGN)           # (a second SIGTERM must not interrupt the clean-up)
            except (ValueError, OSError):
                pass
        if any(p.poll() is None for p in procs):
            end_all(signal.SIGTERM)
            t_end = time.monotonic() + 2.0
            while time.monotonic() < t_end and any(p.poll() is None for p in procs):
                time.sleep(0.05)
            end_all(signal.SIGKILL)
            for p in procs:
                try:
                    p.wait(timeout=5.0)
                except Exception:                                       # noqa: BLE001
                    pass
        if old_term is not None:
            signal.signal(signal.SIGTERM, old_term)
    for r, p in enumerate(procs):
        if codes[r] is None:
            c = p.poll()
            codes[r] = c if c is not None else -9
    for t in pumps:
        t.join(timeout=5.0)
    return (first_bad or 0), [int(c) for c in codes]      # type: ignore[arg-type]
