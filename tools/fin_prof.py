"""Tuning aid: cycle breakdown of head_finish_kernel (wave 0 of the busy workgroups).  Build with MEL_HIPCC_FLAGS="-DMEL_FIN_PROF"."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from melissa_amd import _lib
net, venv, loop = bench.build_workload(torch.device("cuda", 0), 0, 1024, 50, "l_dgn", "round", False, 1)
lib = _lib.load()
fn = lib.mel_debug_fin_prof
fn.argtypes = [C.c_void_p]
buf = (C.c_ulonglong * 5)()
loop.run(20)
torch.cuda.synchronize()
fn(buf)
loop.run(40)
torch.cuda.synchronize()
fn(buf)
v = list(buf)
w = max(v[3], 1)
print("workgroups counted", v[3], "per launch", v[3] / 40)
print(f"kernel start -> row count known      {v[4] / w:8.0f} cycles")
print(f"kernel start -> h0 in LDS (barrier 1) {v[0] / w:8.0f}")
print(f"kernel start -> hidden layer 1 done   {v[1] / w:8.0f}")
print(f"kernel start -> end                   {v[2] / w:8.0f}")
