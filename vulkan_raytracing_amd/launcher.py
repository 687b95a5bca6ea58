"""Start the N ranks of a multi-GPU run from a plain shell (no torchrun): `python3 bench.py --gpus N`.

The reference binds one device (physicalDeviceHandleList[0], src/main.cpp:928); here one process drives one GPU, so an
N-GPU run is N fresh processes.  The parent that calls spawn_ranks() must not have touched the GPU (no HIP call, no
torch.cuda call): replacing or forking a process that has initialised the device is not allowed on this pool, so the ranks
are started as NEW child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment
(what `python -m torch.distributed.run` would set), rank 0's stdout (the single JSON line) is relayed, and the parent exits
with the first non-zero child status, ending the other ranks when one fails.

Nothing in this module imports torch or loads a HIP library.
"""
import os
import signal
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


def rank_env(rank, n, port, base=None):
    """environment of rank `rank` of an n-rank, one-node job (the variables torch.distributed's env:// rendezvous reads)"""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it across processes)
    return env


def spawn_ranks(n, command, out=None, err=None, poll_s=0.05, timeout_s=None, base_env=None):
    """Run `command` (argv list) n times as ranks 0..n-1 and wait for them.  Rank 0's stdout is relayed to `out` (default
    sys.stdout) when the job ends, the other ranks' stdout goes to `err` (default sys.stderr) as it comes.  Returns the exit
    status of the job: 0 when every rank returned 0, otherwise the first non-zero status seen (a signal -s becomes 128 + s,
    a timeout 124); the remaining ranks are terminated (SIGTERM, then SIGKILL after 10 s) — by PID, never by pattern."""
    if n < 1:
        raise ValueError("need at least one rank")
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    port = free_port()
    procs = []
    try:
        err_fd = err.fileno()
    except (AttributeError, OSError, ValueError):
        err_fd = None
    for r in range(n):
        procs.append(subprocess.Popen(list(command), env=rank_env(r, n, port, base_env),
                                      stdout=subprocess.PIPE if r == 0 else (err_fd if err_fd is not None else subprocess.DEVNULL),
                                      stderr=err_fd, start_new_session=False))
    status = 0
    failed = None        # (rank, status) of the first rank that failed
    t0 = time.monotonic()
    rank0_out = b""
    # rank 0's pipe is drained by communicate() at the end; its JSON line is far smaller than a pipe buffer, but a chatty
    # rank 0 must not block on a full pipe, so poll with a non-blocking read
    os.set_blocking(procs[0].stdout.fileno(), False)
    live = set(range(n))
    while live:
        try:
            chunk = procs[0].stdout.read()
            if chunk:
                rank0_out += chunk
        except (BlockingIOError, ValueError):
            pass
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and status == 0:
                status = 128 - rc if rc < 0 else rc
                failed = (r, rc)
        if status != 0 or (timeout_s is not None and time.monotonic() - t0 > timeout_s):
            if status == 0:
                status = 124
            break
        if live:
            time.sleep(poll_s)
    if live:   # a rank failed (or the job timed out): end the others, exactly the processes started above
        for r in live:
            try:
                procs[r].send_signal(signal.SIGTERM)
            except ProcessLookupError:
                pass
        t1 = time.monotonic()
        while any(procs[r].poll() is None for r in live) and time.monotonic() - t1 < 10.0:
            time.sleep(poll_s)
        for r in live:
            if procs[r].poll() is None:
                procs[r].kill()
            procs[r].wait()
    try:
        os.set_blocking(procs[0].stdout.fileno(), True)
        rest = procs[0].stdout.read()
        if rest:
            rank0_out += rest
    except (OSError, ValueError):
        pass
    procs[0].stdout.close()
    text = rank0_out.decode(errors="replace")
    if status != 0:
        # a failed or timed-out job has no result: what rank 0 printed so far is NOT relayed as one (a consumer that parses stdout
        # without looking at the exit status would take a partial line for the answer); the cause goes to `err` in one line
        why = "timed out after %.0f s" % timeout_s if failed is None else "rank %d exited with status %d" % failed
        err.write("[launcher] %d-rank job failed: %s (exit status %d); %d byte(s) of rank 0's stdout withheld\n" % (n, why, status, len(text)))
        err.flush()
        return status
    if text:
        out.write(text)
        out.flush()
    return status
