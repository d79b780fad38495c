"""Tile height A/B of the 8-phase GEMM on the C3 Linear shapes (two groups: 10 952 geo rows + 16 und rows, as the MoT prefill
launches them):   python3 tools/gemm_heights.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attn_small_q import timeit  # noqa: E402
from g2vlm_amd import hip  # noqa: E402

if __name__ == "__main__":
    hip.lib()
    torch.manual_seed(0)
    r = lambda *s: (torch.randn(s, device="cuda") * 0.05).bfloat16()  # noqa: E731
    shapes = [("qkv", 10968, 2048, 1536, hip.EPI_BF16), ("o", 10968, 1536, 1536, hip.EPI_RES_F32), ("gate/up", 10968, 17920, 1536, hip.EPI_SWIGLU),
              ("down", 10968, 1536, 8960, hip.EPI_RES_F32), ("dino qkv", 10992, 3072, 1024, hip.EPI_BF16), ("dino dense", 10992, 1024, 1024, hip.EPI_RES_F32),
              ("dino fc1", 10992, 4096, 1024, hip.EPI_GELU), ("dino fc2", 10992, 1024, 4096, hip.EPI_RES_F32),
              ("dec qkv", 10952, 4608, 1536, hip.EPI_BF16), ("dec fc1", 10952, 6144, 1536, hip.EPI_GELU), ("dec fc2", 10952, 1536, 6144, hip.EPI_RES_F32)]
    hs = (("auto", 0), ("288", hip.P8_H288), ("256", hip.P8_H256), ("224", hip.P8_H224), ("192", hip.P8_H192), ("160", hip.P8_H160), ("128", hip.P8_H128))
    for name, M, N, K, epi in shapes:
        x, w = r(M, K), r(N, K)
        res = torch.randn((M, N), device="cuda") if epi == hip.EPI_RES_F32 else None
        out = torch.empty((M, N // 2 if epi == hip.EPI_SWIGLU else N), dtype=torch.float32 if epi == hip.EPI_RES_F32 else torch.bfloat16, device="cuda")
        best = {hn: 1e30 for hn, _ in hs}
        for _ in range(4):                                   # interleaved rounds, best of each: clock drift hits all forms alike
            for hn, fl in hs:
                best[hn] = min(best[hn], timeit(lambda: hip.linear(x, w, None, epi, out=out, res=res, flags=hip.FORCE_8P | fl), reps=8))
        lo = min(v for k_, v in best.items() if k_ != "auto")
        print(f"{name:10s} M {M} N {N:5d} K {K:4d}: " + " | ".join(f"{hn} {us:7.1f}{'*' if us == lo else ' '}" for hn, us in best.items()))
