"""Start one process per GPU on this node without an external launcher.

`python bench.py --gpus N` (and the crank-out driver with `--gpus N`) call spawn_ranks() when no
launcher has set WORLD_SIZE: the parent starts N fresh children -- the same command line, with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment -- BEFORE anything
in the parent has touched the GPU (the parent never imports torch), waits for them and exits with
the worst of their exit codes.  Children are ordinary child processes (no exec from a process that
has initialised the GPU).  Rank 0 inherits stdout, so its JSON line is the parent's output.
"""
import os
import socket
import subprocess
import sys


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launched_by_launcher():
    """True inside a rank process (torch.distributed.run or spawn_ranks set the variables)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def spawn_ranks(n, argv=None, extra_env=None, timeout=None):
    """Run `python argv` as n rank processes; return the worst exit code."""
    argv = list(sys.argv if argv is None else argv)
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    # the host driver only supports dmabuf IPC (RCCL needs it)
                    "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        if extra_env:
            env.update(extra_env)
        # ranks other than 0 keep stderr, drop stdout: the contract is ONE line on stdout
        out = None if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable] + argv, env=env, stdout=out))
    # Poll ALL children: whichever rank dies first (a bad device, a failed init), the others would sit in
    # init_process_group or a barrier until the collective's own timeout (10-30 minutes) -- they are ended as soon as
    # any child has exited non-zero.  `timeout` is one deadline for the whole run, not per child.
    import time
    worst = 0
    deadline = None if timeout is None else time.monotonic() + timeout
    live = list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0 and worst == 0:
                worst = rc if rc > 0 else 1
                for q in live:
                    q.terminate()
        if not live:
            break
        if deadline is not None and time.monotonic() > deadline:
            worst = 124
            for q in live:
                q.kill()
            for q in live:
                q.wait()
            break
        time.sleep(0.05)
    return worst
