"""Tuning aid: in-kernel cycle breakdown of gat_attend_rows_kernel.  Build with MEL_HIPCC_FLAGS="-DMEL_ATT_PROF=<mode>"
(mode 0 = conv1 attention, 2 = conv2 attention)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from melissa_amd import _lib
net, venv, loop = bench.build_workload(torch.device("cuda", 0), 0, 1024, 50, "l_dgn", "round", False, 1)
lib = _lib.load()
fn = lib.mel_debug_att_prof
fn.argtypes = [C.c_void_p]
buf = (C.c_ulonglong * 8)()
loop.run(20)
fn(buf)
N = 20
loop.run(N)
fn(buf)
v = list(buf)
waves, rows = max(v[5], 1), max(v[6], 1)
print(f"waves per launch {v[5] / N:.0f}, rows per launch {v[6] / N:.0f}, rows per wave {rows / waves:.2f}")
for i, name in enumerate(["prologue (row count, att, bias)", "descriptor load", "attend_target", "stores"]):
    print(f"{name:32s} {v[i] / waves:10.0f} cycles per wave  ({100.0 * v[i] / max(v[4], 1):5.1f} % of wave time)")
print(f"{'whole wave':32s} {v[4] / waves:10.0f} cycles")
