"""Start one process per GPU from a GPU-free parent.

``bench.py --gpus N`` (and ``python -m melissa_amd.train --gpus N``) must work without an external launcher: the
parent - which has not touched the GPU (no HIP call, no ``torch.cuda.is_available()``) - starts N fresh children of
the same script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relays rank 0's stdout and
returns the worst exit code.  The parent's code path imports neither torch nor any HIP binding (tests/test_launch.py runs it
with both made un-importable): devices are counted from the KFD topology in sysfs.

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


KFD_NODES = "/sys/class/kfd/kfd/topology/nodes"


def _kfd_gpus(nodes_dir: str = KFD_NODES, dev_dir: str = "/dev/dri") -> int:
    """GPUs of the KFD topology this process can open: nodes with SIMDs (CPU nodes have ``simd_count 0``) whose render node
    exists and is readable + writable (a container that is given one GPU sees the other nodes in sysfs but not their device
    files).  Plain file reads: no HIP / ROCr / torch call."""
    try:
        names = sorted(os.listdir(nodes_dir), key=lambda x: int(x) if x.isdigit() else 1 << 30)
    except OSError:
        return 0
    count = 0
    for name in names:
        props = {}
        try:
            with open(os.path.join(nodes_dir, name, "properties")) as f:
                for line in f:
                    parts = line.split()
                    if len(parts) == 2:
                        props[parts[0]] = parts[1]
        except OSError:
            continue                                      # a node this process may not read is a node it may not use
        if int(props.get("simd_count", "0")) <= 0:
            continue
        minor = int(props.get("drm_render_minor", "-1"))
        if minor >= 0 and not os.access(os.path.join(dev_dir, f"renderD{minor}"), os.R_OK | os.W_OK):
            continue
        count += 1
    return count


def _apply_visible_list(available: int, value: str | None) -> int:
    """HIP / ROCr semantics of a ``*_VISIBLE_DEVICES`` list: entries are taken in order up to the first invalid one; an
    entry is an index into the ``available`` devices or a ``GPU-<uuid>`` (counted as one device)."""
    if value is None:
        return available
    count = 0
    seen = set()
    for item in value.split(","):
        item = item.strip()
        if not item:
            break
        if item.isdigit():
            if int(item) >= available or int(item) in seen:
                break
            seen.add(int(item))
        elif not item.upper().startswith("GPU-"):
            break
        count += 1
    return min(count, available)


def visible_gpus(env: dict | None = None, nodes_dir: str = KFD_NODES, dev_dir: str = "/dev/dri") -> int:
    """Number of GPUs this process may use, WITHOUT touching a GPU API: the KFD topology in sysfs, filtered by
    ROCR_VISIBLE_DEVICES (applied by the ROCr runtime) and then HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (applied by HIP to
    what ROCr left).  ``torch.cuda.device_count()`` is GPU-free only while its amdsmi path works and falls back to
    ``hipGetDeviceCount`` - which initialises the runtime in the launcher parent - otherwise; the ranks re-check with it
    themselves (``check_rank_device``), where initialising the runtime is what they are about to do anyway."""
    env = os.environ if env is None else env
    nodes_dir, dev_dir = env.get("MEL_KFD_NODES", nodes_dir), env.get("MEL_DRI_DIR", dev_dir)      # (tests point these at a fake tree)
    n = _kfd_gpus(nodes_dir, dev_dir)
    n = _apply_visible_list(n, env.get("ROCR_VISIBLE_DEVICES"))
    hip = env.get("HIP_VISIBLE_DEVICES", env.get("CUDA_VISIBLE_DEVICES"))
    return _apply_visible_list(n, hip)


def check_rank_device() -> None:
    """Called by a RANK (never by the launcher parent) before it binds its device: exit 2 when LOCAL_RANK names a GPU this
    process cannot see - the sysfs count of the parent is a pre-flight check, this is the authoritative one."""
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    have = int(torch.cuda.device_count())
    if local >= have:
        sys.stderr.write(f"rank with LOCAL_RANK={local} but only {have} GPU(s) visible to it\n")
        raise SystemExit(2)


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
    (exactly the PIDs started here).  Returns the largest exit code of the ranks that ended by themselves (signals count as
    128 + signo; the SIGTERM this function sends to the survivors of a failed run does not)."""
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
    stopped = set()                                        # ranks THIS function terminated: their SIGTERM is not a result
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            rc = 128 - rc if rc < 0 else rc
            if r not in stopped or rc not in (0, 143):
                worst = max(worst, rc)
            if rc != 0:                                    # one rank failed: the others would hang in a collective
                for q in sorted(live):
                    procs[q].terminate()
                    stopped.add(q)
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
