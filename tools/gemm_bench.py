"""Microbenchmark of the fp32-MFMA row GEMM at the shapes of the L-DGN step (interleaved rounds in one
process, torch events on the launch stream).  python tools/gemm_bench.py [--rounds 30]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from melissa_amd import _lib  # noqa: E402

SHAPES = [("r_conv2_lin", 10653, 512, 512), ("r_conv1_lin", 16131, 512, 128), ("r_conv1_lin_r", 10653, 512, 128),
          ("r_conv2_lin_r", 4820, 512, 512), ("r_head0", 4820, 256, 1152), ("r_head1", 4820, 128, 128),
          ("conv2_lin", 6630, 512, 512), ("conv1_lin", 12962, 512, 128), ("conv1_lin_r", 6630, 512, 128),
          ("conv2_lin_r", 1024, 512, 512), ("head0", 1024, 256, 1152), ("head1", 1024, 128, 128),
          ("encoder", 12962, 128, 128), ("hl_conv1", 51200, 1024, 128), ("big", 65536, 512, 512)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=30)
    ap.add_argument("--tiles", default="1,2")
    ap.add_argument("--bf16", action="store_true", help="the bf16 feature-path GEMM (mel_gemm_bf16; tiles 1 / 2)")
    ap.add_argument("--split", action="store_true", help="fp32 via split bf16 operands (mel_gemm_f32_split; tiles 1 / 2, weights pre-split)")
    ap.add_argument("--ksplit", type=int, default=0, help="with --split: split-K chunks of the 128 x 128 kernel where K allows")
    ap.add_argument("--lda-pad", type=int, default=0, help="extra floats per A row (row stride K + pad)")
    ap.add_argument("--shapes", default="", help="comma-separated shape names (default: all)")
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda")
    tiles = [int(t) for t in args.tiles.split(",")]
    torch.manual_seed(0)
    for name, M, N, K in SHAPES:
        if args.shapes and name not in args.shapes.split(","):
            continue
        lda = K + args.lda_pad
        A = torch.randn(M, lda, device=dev)[:, :K]
        W = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        Y = torch.empty(M, N, device=dev)
        if args.bf16:
            A, W, Y = A.to(torch.bfloat16), W.to(torch.bfloat16), Y.to(torch.bfloat16)
            ref = torch.addmm(b, A.float(), W.float().t())
            gemm = lambda t: lib.mel_gemm_bf16(A.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), N, M, N, K, 0, 0, t,
                                               _lib.current_stream_ptr())
        elif args.split:
            ref = torch.addmm(b.double(), A.double(), W.double().t())
            ks = args.ksplit if args.ksplit > 1 and (K // 16) % args.ksplit == 0 and K // 16 // args.ksplit >= 4 else 0
            scratch = torch.empty(6 * N * K + 256 + 4 * max(ks, 1) * M * N, dtype=torch.uint8, device=dev)
            first = [True]
            def gemm(t):
                tt = t if first[0] else t + 100
                first[0] = False
                return lib.mel_gemm_f32_split(A.data_ptr(), lda, W.data_ptr(), b.data_ptr(), Y.data_ptr(), N, M, N, K, 0, tt,
                                              ks if t == 2 else 0, scratch.data_ptr(), scratch.numel(), _lib.current_stream_ptr())
        else:
            ref = torch.addmm(b, A, W.t())
            gemm = lambda t: lib.mel_gemm_f32(A.data_ptr(), lda, W.data_ptr(), b.data_ptr(), Y.data_ptr(), N, M, N, K, 0, t,
                                              _lib.current_stream_ptr())
        res = {}
        for t in tiles:
            _lib.check(gemm(t))
            err = (Y.to(ref.dtype) - ref).abs().max().item()
            res[t] = [err, []]
        for _ in range(args.rounds):
            for t in tiles:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    gemm(t)
                e1.record()
                e1.synchronize()
                res[t][1].append(e0.elapsed_time(e1) / 4 * 1e3)
        flops = 2.0 * M * N * K
        line = f"{name:12s} M={M:6d} N={N:5d} K={K:5d} {flops/1e9:7.2f} GF |"
        for t in tiles:
            ts = sorted(res[t][1])
            med = ts[len(ts) // 2]
            line += f" tile{t}: {med:7.1f} us {flops/med/1e6:6.1f} TF (min {ts[0]:6.1f}) err {res[t][0]:.1e} |"
        print(line, flush=True)


if __name__ == "__main__":
    main()
