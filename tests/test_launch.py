"""The GPU-free rank launcher behind ``bench.py --gpus N`` (melissa_amd/launch.py), with a dummy target."""
import io
import json
import os
import subprocess
import sys
import textwrap

import pytest

from melissa_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _script(tmp_path, body):
    path = tmp_path / "target.py"
    path.write_text(textwrap.dedent(body))
    return str(path)


def test_spawn_ranks_sets_env_and_relays_rank0(tmp_path):
    script = _script(tmp_path, """
        import json, os, sys
        r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1"
        import torch.distributed as dist                  # the ranks really rendezvous (gloo, CPU)
        dist.init_process_group("gloo", rank=r, world_size=w)
        import torch
        t = torch.tensor([float(r + 1)])
        dist.all_reduce(t)
        print(json.dumps({"rank": r, "world": w, "sum": float(t), "argv": sys.argv[1:]}))
        print("noise on stderr", file=sys.stderr)
        dist.destroy_process_group()
    """)
    out, err = io.StringIO(), io.StringIO()
    rc = launch.spawn_ranks([script, "--steps", "3"], 3, stdout=out, stderr=err)
    assert rc == 0
    lines = [l for l in out.getvalue().splitlines() if l.strip()]
    assert len(lines) == 1                                   # only rank 0 reaches stdout
    rec = json.loads(lines[0])
    assert rec == {"rank": 0, "world": 3, "sum": 6.0, "argv": ["--steps", "3"]}
    assert "[rank 1] " in err.getvalue() and "[rank 2] " in err.getvalue()


def test_failed_rank_stops_the_others(tmp_path):
    script = _script(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(60)                                      # would hang in a collective
    """)
    rc = launch.spawn_ranks([script], 2, stdout=io.StringIO(), stderr=io.StringIO())
    assert rc != 0


def test_maybe_spawn_is_a_noop_for_a_rank_or_one_gpu(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert launch.maybe_spawn("x.py", [], 1) is None
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert launch.maybe_spawn("x.py", [], 4) is None
    with pytest.raises(SystemExit) as e:                     # WORLD_SIZE contradicts --gpus: refuse, never mis-report
        launch.maybe_spawn("x.py", [], 8)
    assert e.value.code == 2


def _fake_kfd(tmp_path, gpus, cpus=2, openable=None):
    """A KFD topology tree like /sys/class/kfd/kfd/topology/nodes with ``cpus`` CPU nodes and ``gpus`` GPU nodes, plus a /dev/dri
    stand-in holding the render nodes of the GPUs in ``openable`` (default: all)."""
    nodes, dev = tmp_path / "nodes", tmp_path / "dri"
    dev.mkdir()
    for i in range(cpus + gpus):
        d = nodes / str(i)
        d.mkdir(parents=True)
        gpu = i >= cpus
        minor = 128 + (i - cpus) if gpu else -1
        (d / "properties").write_text(f"cpu_cores_count {0 if gpu else 64}\nsimd_count {1024 if gpu else 0}\n"
                                      f"gfx_target_version {90500 if gpu else 0}\ndrm_render_minor {minor}\n")
        if gpu and (openable is None or (i - cpus) in openable):
            (dev / f"renderD{minor}").write_text("")
    return str(nodes), str(dev)


def test_visible_gpus_reads_sysfs_and_honours_the_visibility_lists(tmp_path):
    nodes, dev = _fake_kfd(tmp_path, gpus=8)
    count = lambda env: launch.visible_gpus(env, nodes, dev)
    assert count({}) == 8
    assert count({"HIP_VISIBLE_DEVICES": "0,1,2"}) == 3
    assert count({"CUDA_VISIBLE_DEVICES": "3"}) == 1
    assert count({"HIP_VISIBLE_DEVICES": "0,1", "CUDA_VISIBLE_DEVICES": "0,1,2,3"}) == 2          # HIP_ wins over CUDA_
    assert count({"ROCR_VISIBLE_DEVICES": "0,1,2,3", "HIP_VISIBLE_DEVICES": "1,3"}) == 2          # HIP indexes what ROCr left
    assert count({"ROCR_VISIBLE_DEVICES": "0,1", "HIP_VISIBLE_DEVICES": "0,1,2"}) == 2            # index 2 is invalid: list ends
    assert count({"HIP_VISIBLE_DEVICES": "0,9,1"}) == 1                                            # ... at the first invalid entry
    assert count({"HIP_VISIBLE_DEVICES": ""}) == 0
    assert count({"ROCR_VISIBLE_DEVICES": "GPU-abcdef0123456789,GPU-0123"}) == 2
    assert launch.visible_gpus({}, str(tmp_path / "absent"), dev) == 0                               # no KFD at all (this container)


def test_visible_gpus_counts_only_devices_this_process_can_open(tmp_path):
    """A container that is handed one GPU of an 8-GPU host still sees 8 nodes in sysfs; only the render nodes it was given count."""
    nodes, dev = _fake_kfd(tmp_path, gpus=8, openable={2})
    assert launch.visible_gpus({}, nodes, dev) == 1


PARENT_GUARD = """
import builtins, os
_real = builtins.__import__
def guarded(name, *a, **k):
    if name.split(".")[0] == "torch":
        raise AssertionError("launcher parent imported " + name)
    return _real(name, *a, **k)
def no_library(*a, **k):
    raise AssertionError("launcher parent imported a shared library through ctypes")
if "RANK" not in os.environ:            # the PARENT: torch (any HIP entry point) and loading a library through ctypes are off limits
    builtins.__import__ = guarded
    import ctypes
    ctypes.CDLL = no_library
"""


@pytest.mark.parametrize("target", ["bench", "train"])
def test_launcher_parent_never_touches_torch_or_the_library(tmp_path, target):
    """The process that starts the ranks must not initialise the GPU - on this pool a fork + exec out of a GPU-initialised
    process takes the machine down.  Run the real ``bench.py --gpus 2`` / ``python -m melissa_amd.train --gpus 2`` with a
    sitecustomize that makes ``import torch`` and ``ctypes.CDLL`` raise in any process without RANK: the parent must get as far
    as starting two ranks (which then fail on their own - there is no GPU here - or are refused by the device count)."""
    (tmp_path / "sitecustomize.py").write_text(PARENT_GUARD)
    nodes, dev = _fake_kfd(tmp_path, gpus=2)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HIP_VISIBLE_DEVICES",
                                                            "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES")}
    env["PYTHONPATH"] = os.pathsep.join([str(tmp_path), ROOT, env.get("PYTHONPATH", "")])
    env["MEL_KFD_NODES"], env["MEL_DRI_DIR"] = nodes, dev          # (test hook: where the launcher reads the topology)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
           "--no-extra-legs"] if target == "bench" else \
          [sys.executable, "-m", "melissa_amd.train", "--gpus", "2", "--updates", "1"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert "launcher parent imported" not in res.stderr, res.stderr[-2000:]
    # both ranks were started (their stderr is relayed with a prefix) and left with the "no GPU visible to this rank" code
    assert "[rank 0]" in res.stderr and "[rank 1]" in res.stderr, res.stderr[-2000:]
    assert res.returncode == 2 and "GPU(s) visible to it" in res.stderr


def test_bench_refuses_more_gpus_than_visible(tmp_path):
    """python bench.py --gpus 4096 exits with code 2 instead of printing a 1-GPU line (more ranks than any host has GPUs: the
    refusal does not depend on the machine the test runs on)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4096", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 2 and "GPU(s) visible" in res.stderr and res.stdout.strip() == ""
