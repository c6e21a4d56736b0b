"""Tuning aid: in-kernel cycle breakdown of gemm_split_big_kernel.  Build with
MEL_HIPCC_FLAGS="-DMEL_GEMM_PROF=99 -DMEL_SPLIT_PROF", then: python tools/split_prof.py M N K"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from melissa_amd import _lib
M, N, K = [int(x) for x in sys.argv[1:4]]
lib = _lib.load()
fn = lib.mel_debug_gemm_prof
fn.argtypes = [C.c_void_p]
buf = (C.c_ulonglong * 8)()
fine = (C.c_ulonglong * 16)()
fn2 = lib.mel_debug_split_prof
fn2.argtypes = [C.c_void_p]
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
Y = torch.empty(M, N, device="cuda")
scratch = torch.empty(6 * N * K + 256, dtype=torch.uint8, device="cuda")
def run(i):
    _lib.check(lib.mel_gemm_f32_split(A.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), N, M, N, K, 0, 2 + (100 if i else 0), 0,
                                      scratch.data_ptr(), scratch.numel(), _lib.current_stream_ptr()))
for i in range(3):
    run(i)
fn(buf)
fn2(fine)
R = 5
for i in range(R):
    run(1)
fn(buf)
v = list(buf)
wgs, steps = max(v[6], 1), max(v[7], 1)
print(f"M={M} N={N} K={K}: workgroups per launch {v[6] / R:.0f}, K steps per workgroup {steps / wgs:.1f}, kernel {v[5] / wgs:.0f} cycles per workgroup")
names = ["stream bookkeeping (+ next item)", "reads + MFMAs + fill + prefetch", "LDS writes landing", "step barrier", "epilogue"]
for i, name in enumerate(names):
    print(f"  {name:36s} {v[i] / steps:9.0f} cycles per K step  ({100.0 * v[i] / max(v[5], 1):5.1f} % of the kernel)")
fn2(fine)
f = list(fine)
labels = ["12 fragment reads issued", "MFMA group 0 issued (waits for the fragments)", "fill A rows 0 (vmcnt wait + split + 3 ds_write)",
          "MFMA group 1", "fill A rows 1", "MFMA group 2", "fill W (3 ds_write_b128)", "MFMA group 3", "2 A loads issued",
          "MFMA group 4", "3 W loads issued", "MFMA group 5", "(end)"]
print("  issue-time stamps inside the MFMA phase, cycles per K step:")
for k in range(13):
    print(f"    {labels[k]:52s} {f[k] / steps:8.0f}")
