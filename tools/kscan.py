import sys, os, torch
sys.path.insert(0, "/root/repo")
from melissa_amd import _lib
lib = _lib.load()
for (M,N,K) in [(16384,512,128),(16384,512,512),(16384,512,2048),(16384,512,8192),(65536,512,512),(65536,512,4096)]:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    Y = torch.empty(M, N, device="cuda")
    for tile in (1,2):
        for _ in range(3):
            lib.mel_gemm_f32(A.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), N, M, N, K, 0, tile, _lib.current_stream_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.mel_gemm_f32(A.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), N, M, N, K, 0, tile, _lib.current_stream_ptr())
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1)/5*1e3
        print(f"M={M} N={N} K={K} tile{tile}: {us:8.1f} us {2.0*M*N*K/us/1e6:6.1f} TF", flush=True)
