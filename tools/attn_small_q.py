"""Prefill attention with few query rows against a long cache (the 731-token ViT prefill of chat_with_recon, a long text
prompt): kernel time by query-tile height and persistent block count.
    python3 tools/attn_small_q.py [Lq] [Lk]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g2vlm_amd import hip  # noqa: E402


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


if __name__ == "__main__":
    Lq = int(sys.argv[1]) if len(sys.argv) > 1 else 731
    Lk = int(sys.argv[2]) if len(sys.argv) > 2 else 15000
    Hq, Hkv, D = 12, 2, 128
    hip.lib()
    torch.manual_seed(0)
    q = torch.randn((Lq, Hq * D), device="cuda").bfloat16()
    k = torch.randn((Lk, Hkv * D), device="cuda").bfloat16()
    v = torch.randn((Lk, Hkv * D), device="cuda").bfloat16()
    o = torch.empty_like(q)
    fl = 4.0 * Lq * Lk * Hq * D
    for rows in (256, 128):
        for mb in (None, 512, 1024):
            kw = {} if mb is None else {"max_blocks": mb}
            try:
                plan = hip.make_attn_plan([(0, Lq, 0, Lk, False)], Hq, "cuda", tile_rows=rows, **kw)
            except TypeError:
                continue
            us = timeit(lambda: hip.flash_attn(q, k, v, o, plan, Hq, Hkv, D))
            print(f"Lq {Lq} Lk {Lk} tile_rows {rows} max_blocks {mb}: blocks {plan.n_blocks} splits {plan.n_split}  {us:7.1f} us  {fl / us / 1e6:6.0f} TF/s")
