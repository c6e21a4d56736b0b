"""Tuning aid: in-kernel cycle breakdown of env_round_kernel over a few round steps of the bench workload.
Build with MEL_HIPCC_FLAGS="-DMEL_ENV_PROF"."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from melissa_amd import _lib
net, venv, loop = bench.build_workload(torch.device("cuda", 0), 0, 1024, 50, "l_dgn", "round", False, 1)
lib = _lib.load()
fn = lib.mel_debug_env_prof
fn.argtypes = [C.c_void_p]
buf = (C.c_ulonglong * 9)()
loop.run(20)
torch.cuda.synchronize()
fn(buf)
loop.run(40)
torch.cuda.synchronize()
fn(buf)
v = list(buf)
w = max(v[7], 1)
print("wavefronts sampled", v[7], "loop iterations per env round", v[8] / w)
names = ["state load", "round loop", "  env_step with the world step", "  other env_step calls", "  env_observe", "  episode end (log + reset)", "state store"]
for i, name in enumerate(names):
    print(f"{name:34s} {v[i] / w:9.0f} cycles per env round")
fw = lib.mel_debug_world_prof
fw.argtypes = [C.c_void_p]
wb = (C.c_ulonglong * 4)()
fw(wb)
loop.run(20)
torch.cuda.synchronize()
fw(wb)
c = max(wb[3], 1)
print(f"world step (sample of {wb[3]} calls): relay + scripted {wb[0] / c:.0f}, move + all-pairs edges {wb[1] / c:.0f}, two-hop masks {wb[2] / c:.0f} cycles")
