"""Start one process per GPU from a GPU-free parent.

``bench.py --gpus N`` (and ``python -m melissa_amd.train --gpus N``) must work without an external launcher: the
parent - which has not touched the GPU (no HIP call, no ``torch.cuda.is_available()``) - starts N fresh children of
the same script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relays rank 0's stdout and
returns the worst exit code.  Nothing here imports torch: it has to stay cheap and GPU-free.

The reference has no launcher (it is single-process apart from tianshou's SubprocVectorEnv, l_dgn.py:137-146); this is
the process model SURVEY.md 8(e) prescribes: env shards are independent, one rank per GPU.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus() -> int:
    """Number of GPUs this process may use, without initialising the HIP runtime (``device_count`` only reads the
    topology; it honours HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES)."""
    import torch
    return int(torch.cuda.device_count())


def _pump(stream, sink, prefix="", other=None):
    """``other``: where rank 0's non-JSON stdout lines go (library chatter such as gloo's connection banner must not
    break the one-JSON-line contract of the relayed stdout)."""
    for line in iter(stream.readline, ""):
        if other is not None and not line.lstrip().startswith("{"):
            other.write("[rank 0] " + line)
            other.flush()
            continue
        sink.write(prefix + line)
        sink.flush()
    stream.close()


def spawn_ranks(argv, n_ranks: int, python: str | None = None, extra_env: dict | None = None,
                poll_s: float = 0.2, stdout=None, stderr=None) -> int:
    """Run ``python argv...`` as ``n_ranks`` processes (rank r: RANK = LOCAL_RANK = r) and wait for them.

    Rank 0's JSON lines are relayed to ``stdout`` unchanged (the bench's one JSON line; anything else it prints goes to
    ``stderr``); the other ranks' stdout and every
    rank's stderr go to ``stderr`` with a ``[rank r]`` prefix.  If a rank exits non-zero the others are terminated
    (exactly the PIDs started here).  Returns the largest exit code (signals count as 128 + signo)."""
    stdout = stdout or sys.stdout
    stderr = stderr or sys.stderr
    port = free_port()
    procs, pumps = [], []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n_ranks) // n_ranks)))
        if extra_env:
            env.update(extra_env)
        p = subprocess.Popen([python or sys.executable, *argv], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             text=True, bufsize=1)
        procs.append(p)
        out_sink, out_prefix = (stdout, "") if r == 0 else (stderr, f"[rank {r}] ")
        for stream, sink, prefix, other in ((p.stdout, out_sink, out_prefix, stderr if r == 0 else None),
                                            (p.stderr, stderr, f"[rank {r}] ", None)):
            t = threading.Thread(target=_pump, args=(stream, sink, prefix, other), daemon=True)
            t.start()
            pumps.append(t)
    worst = 0
    live = set(range(n_ranks))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            rc = 128 - rc if rc < 0 else rc
            worst = max(worst, rc)
            if rc != 0:                                    # one rank failed: the others would hang in a collective
                for q in sorted(live):
                    procs[q].terminate()
        time.sleep(poll_s)
    for t in pumps:
        t.join(timeout=5)
    return worst


def maybe_spawn(script: str, argv, n_ranks: int, check_devices: bool = True) -> int | None:
    """The launcher entry of a script with a ``--gpus N`` flag.  Returns None when this process IS a rank (or N == 1)
    and should carry on; otherwise runs the ranks and returns their exit code (the caller exits with it).
    Exits with code 2 when the environment contradicts ``--gpus``."""
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is not None:
        if int(world_env) != n_ranks:
            sys.stderr.write(f"--gpus {n_ranks} contradicts WORLD_SIZE={world_env}: refusing to report a wrong n_gpus\n")
            raise SystemExit(2)
        return None
    if n_ranks <= 1:
        return None
    if check_devices:
        have = visible_gpus()
        if have < n_ranks:
            sys.stderr.write(f"--gpus {n_ranks} but only {have} GPU(s) visible\n")
            raise SystemExit(2)
    return spawn_ranks([script, *argv], n_ranks)
