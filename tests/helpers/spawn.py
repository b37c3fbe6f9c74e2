"""Start R ranks of a script as fresh child processes (one process per rank, torchrun's environment contract) and wait.
Used by tests/conftest.py (before the pytest process touches the GPU) and mirrored by bench.py --gpus N."""
import os
import socket
import subprocess
import sys


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_ranks(script_args, world, extra_env=None, timeout=900, log_dir=None):
    """Returns (return codes, log paths).  Ranks inherit the environment plus RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*."""
    port = free_port()
    procs, logs = [], []
    for r in range(world):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env or {})
        log = os.path.join(log_dir, "rank%d.log" % r) if log_dir else os.devnull
        logs.append(log)
        f = open(log, "w")
        procs.append((subprocess.Popen([sys.executable] + list(script_args), env=env, stdout=f, stderr=subprocess.STDOUT), f))
    rcs = []
    for p, f in procs:
        try:
            rcs.append(p.wait(timeout=timeout))
        except subprocess.TimeoutExpired:
            p.kill()
            rcs.append(-9)
        f.close()
    return rcs, logs
