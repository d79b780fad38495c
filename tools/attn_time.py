"""Time the prefill attention at a named shape (cases of tools/attn_only.py), interleaving nothing else:
    python3 tools/attn_time.py [mot|c4|c4rank|dino|dec|vit] [reps]"""
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
    cases = {"mot": (10968, 10976, 12, 2, 128, 1, 256), "c4": (43872, 43880, 12, 2, 128, 1, 256), "c4rank": (5484, 43880, 12, 2, 128, 1, 256),
             "dino": (10952, 10952, 16, 16, 64, 8, 256), "dec": (10952, 10952, 16, 16, 96, 8, 256), "vit": (2916, 2916, 16, 16, 80, 1, 256)}
    for what in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["mot"]):
        Lq, Lk, Hq, Hkv, D, nwin, rows = cases[what]
        q = torch.randn((Lq, Hq * D), device="cuda").bfloat16()
        k = torch.randn((Lk, Hkv * D), device="cuda").bfloat16()
        v = torch.randn((Lk, Hkv * D), device="cuda").bfloat16()
        o = torch.empty_like(q)
        wl = Lq // nwin
        wins = [(i * wl, wl, i * wl if nwin > 1 else 0, wl if nwin > 1 else Lk, False) for i in range(nwin)]
        plan = hip.make_attn_plan(wins, Hq, "cuda", tile_rows=rows)
        best = min(timeit(lambda: hip.flash_attn(q, k, v, o, plan, Hq, Hkv, D), reps=20) for _ in range(3))
        fl = 4.0 * Lq * (Lk if nwin == 1 else wl) * Hq * D
        print(f"{what:7s} {best:9.1f} us  {fl / best / 1e6:7.0f} TF/s  checksum {float(o.float().abs().mean()):.6f}")
