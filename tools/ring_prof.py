"""Tuning aid: in-kernel cycle breakdown of gemm_f32_ring_kernel over a few round steps.
Build with MEL_HIPCC_FLAGS="-DMEL_RING_PROF=3" (the heads' first layer, the one launch the ring kernel serves)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from melissa_amd import _lib
net, venv, loop = bench.build_workload(torch.device("cuda", 0), 0, 1024, 50, "l_dgn", "round", False, 1)
lib = _lib.load()
fn = lib.mel_debug_ring_prof
fn.argtypes = [C.c_void_p]
buf = (C.c_ulonglong * 8)()
loop.run(20)
fn(buf)
N = 20
loop.run(N)
fn(buf)
v = list(buf)
wgs = v[4]
print("workgroups counted", wgs, "per step", wgs / N)
names = ["consumer MFMA section", "consumer step barrier", "consumer epilogue", "consumer whole kernel", "-", "loader issue", "loader wait_landed", "loader barrier"]
for i in (0, 1, 2, 3, 5, 6, 7):
    print(f"{names[i]:26s} {v[i] / wgs:12.0f} ticks per workgroup-launch  ({100.0 * v[i] / max(v[3], 1):5.1f} % of consumer kernel time)")
