"""Sustained (not burst) GEMM rate on random data: each form is launched back to back for `secs` seconds and timed over the last
third (events), so the chip's power management has settled (MI355X guide, 'DVFS give-back' item 6: >= 2 s).  Yardstick only.
    python3 tools/gemm_sustained.py [secs]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g2vlm_amd import hip  # noqa: E402


def sustained(fn, secs):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    while time.time() - t0 < secs * 2 / 3:                  # settle
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); n += 20
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(20, n // 2)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


if __name__ == "__main__":
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    hip.lib()
    torch.manual_seed(0)
    for M, N, K in ((8192, 8192, 8192), (10968, 17920, 1536), (10968, 1536, 8960)):
        x = (torch.randn((M, K), device="cuda") * 0.05).bfloat16()
        w = (torch.randn((N, K), device="cuda") * 0.05).bfloat16()
        out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        wt = w.t()
        fl = 2.0 * M * N * K / 1e6
        forms = (("eight-wave", lambda: hip.linear(x, w, None, out=out, flags=hip.FORCE_8P | hip.P8_EIGHT_WAVES)),
                 ("four-wave", lambda: hip.linear(x, w, None, out=out, flags=hip.FORCE_8P | hip.P8_FOUR_WAVES)),
                 ("torch.matmul", lambda: torch.matmul(x, wt, out=out)))
        line = []
        for name, fn in forms:
            us = sustained(fn, secs)
            line.append(f"{name} {us:8.1f} us {fl / us:6.0f} TF/s")
        print(f"M {M:6d} N {N:6d} K {K:6d} sustained {secs:.0f} s: " + " | ".join(line), flush=True)
