"""Achieved TFLOP/s of hip.linear on the Linear shapes of one und-expert layer at a given row count:
    python3 tools/gemm_shapes.py [M ...]      (default: 731 = one ViT image's prefill, 5848 = eight, 10968 = the C3 geo prefill)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attn_small_q import timeit  # noqa: E402
from g2vlm_amd import hip  # noqa: E402
from g2vlm_amd.weights import interleave_gate_up  # noqa: E402

if __name__ == "__main__":
    Ms = [int(v) for v in sys.argv[1:]] or [731, 5848, 10968]
    hip.lib()
    torch.manual_seed(0)
    H, F = 1536, 8960
    r = lambda *s: (torch.randn(s, device="cuda") * 0.05).bfloat16()  # noqa: E731
    wqkv, wo, wgu, wd = r(2048, H), r(H, H), interleave_gate_up(r(F, H), r(F, H)), r(H, F)
    for M in Ms:
        x, act, xf = r(M, H), r(M, F), torch.randn((M, H), device="cuda")
        ao = r(M, H)
        cases = [("qkv  N 2048  K 1536", lambda: hip.linear(x, wqkv, None), 2 * M * 2048 * H),
                 ("o    N 1536  K 1536 +res", lambda: hip.linear(ao, wo, None, hip.EPI_RES_F32, out=xf, res=xf), 2 * M * H * H),
                 ("g/u  N 17920 K 1536 swiglu", lambda: hip.linear(x, wgu, None, hip.EPI_SWIGLU), 2 * M * 2 * F * H),
                 ("down N 1536  K 8960 +res", lambda: hip.linear(act, wd, None, hip.EPI_RES_F32, out=xf, res=xf), 2 * M * H * F)]
        tot = 0.0
        for name, fn, fl in cases:
            us = timeit(fn)
            tot += us
            print(f"M {M:6d}  {name:28s} {us:8.1f} us  {fl / us / 1e6:7.0f} TF/s")
        print(f"M {M:6d}  layer Linears {tot:8.1f} us")
