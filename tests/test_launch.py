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


def test_bench_refuses_more_gpus_than_visible():
    """python bench.py --gpus 8 on a box without 8 GPUs exits non-zero instead of printing a 1-GPU line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 2 and "GPU(s) visible" in res.stderr and res.stdout.strip() == ""
