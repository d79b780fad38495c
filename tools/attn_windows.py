"""Per-view window attention of the DINO encoder (head_dim 64, 16 heads) and the Pi3 decoders (head_dim 96, 16 heads; self
and cross windows) at the C3 shape: kernel time by query-tile height.
    python3 tools/attn_windows.py [views] [P]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g2vlm_amd import hip  # noqa: E402
from attn_small_q import timeit  # noqa: E402

if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 1369
    hip.lib()
    torch.manual_seed(0)
    cases = [("dino (H1 windows of P over N (P + 5) rows)", 64, N * (P + 5), [(i * P, P, i * P, P, False) for i in range(N)]),
             ("decoder self", 96, N * P, [(v * P, P, v * P, P, False) for v in range(N)]),
             ("decoder cross (all views -> view 0)", 96, N * P, [(v * P, P, 0, P, False) for v in range(N)])]
    for name, D, T, wins in cases:
        H = 16
        q = torch.randn((T, H * D), device="cuda").bfloat16()
        k = torch.randn((T, H * D), device="cuda").bfloat16()
        v = torch.randn((T, H * D), device="cuda").bfloat16()
        o = torch.zeros_like(q)
        fl = 4.0 * N * P * P * H * D
        for rows in (128, 256):
            plan = hip.make_attn_plan(wins, H, "cuda", tile_rows=rows)
            us = timeit(lambda: hip.flash_attn(q, k, v, o, plan, H, H, D))
            print(f"{name:48s} D {D:3d} tile_rows {rows}: blocks {plan.n_blocks:4d}  {us:7.1f} us  {fl / us / 1e6:6.0f} TF/s")
