"""Two-rank rehearsal of the training loop's DEFAULT multi-rank form on a one-GPU box (run it straight from the shell / gpurun:
it must start from a process that has not touched the GPU - this pool forbids starting programs out of a GPU-initialised
process, which is also why this is a script and not a ``-m gpu`` pytest case: by the time a test runs, pytest's process has
initialised the runtime).

``python -m melissa_amd.train --gpus 2`` starts two ranks from the GPU-free launcher parent (melissa_amd/launch.py); every update
is graph A (sample .. backward, gradients packed) | EAGER all-reduce of the flat gradient | graph B (unpack, Adam) - the collective
never enters a capture.  Here both ranks use cuda:0 and gloo carries the collective (RCCL needs one GPU per rank).  Checks: the
updates really come from the graphs, the replicas stay bit-identical, no env error flag.  python tools/rehearse_two_ranks.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for model in ("hl_dgn", "l_dgn"):
        cmd = [sys.executable, "-m", "melissa_amd.train", "--gpus", "2", "--backend", "gloo", "--model", model, "--nodes", "20",
               "--envs", "64", "--updates", "12"]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert res.returncode == 0, res.stderr[-3000:]
        lines = [l for l in res.stdout.splitlines() if l.lstrip().startswith("{")]
        assert len(lines) == 1, res.stdout
        out = json.loads(lines[0])
        print(lines[0])
        assert out["world"] == 2 and out["updates_from_hip_graphs"] is True and out["warmup_updates"] == 2, out
        assert out["replicas_identical"] is True and out["errors"] == 0 and out["loss_last"] == out["loss_last"], out
    print("two-rank rehearsal ok: captured updates with an eager collective between the graphs, replicas identical")


if __name__ == "__main__":
    main()
