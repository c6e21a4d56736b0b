"""Tuning aid: in-kernel cycle breakdown of gemm_f32_persistent_kernel (the default GEMM of the round step).
Build with MEL_HIPCC_FLAGS="-DMEL_GEMM_PROF=<tag>" (tag 1 = conv1, 2 = conv2, 3 = heads' second layer)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from melissa_amd import _lib
net, venv, loop = bench.build_workload(torch.device("cuda", 0), 0, 1024, 50, "l_dgn", "round", False, 1)
lib = _lib.load()
fn = lib.mel_debug_gemm_prof
fn.argtypes = [C.c_void_p]
buf = (C.c_ulonglong * 8)()
loop.run(20)
fn(buf)
N = 20
loop.run(N)
fn(buf)
v = list(buf)
wgs = max(v[6], 1)
print("workgroups counted", v[6], "per step", v[6] / N)
names = ["prefetch issue + next-tile setup", "LDS fragment reads + MFMA chain", "wait prefetch + fill LDS stage", "step barrier", "epilogue", "whole kernel"]
for i, name in enumerate(names):
    print(f"{name:34s} {v[i] / wgs:12.0f} cycles per workgroup-launch  ({100.0 * v[i] / max(v[5], 1):5.1f} %)")
