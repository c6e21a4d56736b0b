"""Tuning aid: does the row stride of the operands (K * 4 bytes) matter to the 64 x 64 GEMM kernels?  Times the
heads' first-layer shape (M = 4820, N = 256) at K around 1152 and the conv2 shape around K = 512."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from melissa_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda")
torch.manual_seed(0)
for M, N, Ks in ((4820, 256, (1152,)), (1024, 256, (1152,)), (15468, 512, (512,)), (16131, 512, (128,)), (51200, 1024, (128,))):
    for K in Ks:
        for pad in (0,):
            A = torch.randn(M, K + pad, device=dev)
            W = torch.randn(N, K, device=dev) / K ** 0.5
            b = torch.randn(N, device=dev)
            Y = torch.empty(M, N, device=dev)
            line = f"M={M} N={N} K={K} lda={K + pad}:"
            for t in (11, 32, 33):
                call = lambda: lib.mel_gemm_f32(A.data_ptr(), K + pad, W.data_ptr(), b.data_ptr(), Y.data_ptr(), N, M, N, K, 0, t,
                                                _lib.current_stream_ptr())
                _lib.check(call())
                err = (Y - torch.addmm(b, A[:, :K], W.t())).abs().max().item()
                ts = []
                for _ in range(20):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(4):
                        call()
                    e1.record()
                    e1.synchronize()
                    ts.append(e0.elapsed_time(e1) / 4 * 1e3)
                ts.sort()
                line += f"  tile{t}: {ts[len(ts) // 2]:6.1f} us {2.0 * M * N * K / ts[len(ts) // 2] / 1e6:6.1f} TF (err {err:.0e})"
            print(line, flush=True)
