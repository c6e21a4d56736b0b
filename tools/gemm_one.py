"""Run one GEMM shape/tile a few times (for rocprofv3 --pmc passes)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from melissa_amd import _lib
M, N, K, tile, reps = [int(x) for x in sys.argv[1:6]]
lib = _lib.load()
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
Y = torch.empty(M, N, device="cuda")
for _ in range(reps):
    lib.mel_gemm_f32(A.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), N, M, N, K, 0, tile, _lib.current_stream_ptr())
torch.cuda.synchronize()
