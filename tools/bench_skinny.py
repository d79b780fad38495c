"""Skinny (M <= 64) GEMM microbench at the decode / text-prefill shapes: default (cross-workgroup K split through the
workspace) vs a workspace too small for it (split inside the workgroup only).  us per launch and weight GB/s.
    python3 tools/bench_skinny.py [M ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g2vlm_amd import hip  # noqa: E402


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # rotate through several weight copies so the stream is from HBM, not from the 256 MB infinity cache
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


if __name__ == "__main__":
    Ms = [int(v) for v in sys.argv[1:]] or [2, 8, 16, 40]
    hip.lib()
    torch.manual_seed(0)
    shapes = [("qkv", 2048, 1536, hip.EPI_BF16), ("o", 1536, 1536, hip.EPI_RES_F32), ("gateup", 17920, 1536, hip.EPI_SWIGLU),
              ("down", 1536, 8960, hip.EPI_RES_F32), ("lm_head", 151936, 1536, hip.EPI_BF16)]
    tiny = torch.zeros(64, dtype=torch.int32, device="cuda")
    print(f"{'shape':8s} {'M':>3s} {'MB':>7s}   split-ws us  GB/s    in-wg us  GB/s")
    for name, N, K, epi in shapes:
        ncopy = max(2, min(24, int(600e6 / (2 * N * K))))
        ws_ = [(torch.randn((N, K), device="cuda") * K ** -0.5).bfloat16() for _ in range(ncopy)]
        for M in Ms:
            x = torch.randn((M, K), device="cuda").bfloat16()
            n_out = N // 2 if epi == hip.EPI_SWIGLU else N
            out = torch.zeros((M, n_out), dtype=torch.float32 if epi == hip.EPI_RES_F32 else torch.bfloat16, device="cuda")
            res = out if epi == hip.EPI_RES_F32 else None
            r = []
            for ws in (None, tiny):
                us = timeit(lambda i=0: hip.linear(x, ws_[i % ncopy], None, epi, out=out, res=res, ws=ws))
                r.append(us)
            mb = 2 * N * K / 1e6
            print(f"{name:8s} {M:3d} {mb:7.1f}   {r[0]:9.1f} {mb / r[0] * 1e3:6.0f}   {r[1]:9.1f} {mb / r[1] * 1e3:6.0f}")
