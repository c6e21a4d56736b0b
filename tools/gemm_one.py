"""Run one GEMM shape/tile a few times (for rocprofv3 --pmc passes): python tools/gemm_one.py M N K tile reps [split]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from melissa_amd import _lib
M, N, K, tile, reps = [int(x) for x in sys.argv[1:6]]
split = len(sys.argv) > 6 and sys.argv[6] == "split"
lib = _lib.load()
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
Y = torch.empty(M, N, device="cuda")
scratch = torch.empty(6 * N * K + 256, dtype=torch.uint8, device="cuda")
for i in range(reps):
    if split:
        lib.mel_gemm_f32_split(A.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), N, M, N, K, 0, tile + (100 if i else 0), 0,
                               scratch.data_ptr(), scratch.numel(), _lib.current_stream_ptr())
    else:
        lib.mel_gemm_f32(A.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), N, M, N, K, 0, tile, _lib.current_stream_ptr())
torch.cuda.synchronize()
