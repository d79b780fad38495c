"""hip.linear on plain GEMMs by main-loop form (the guide's 8-phase template is quoted at 4096^3 / 8192^3: 1.32-1.47 PF):
default (software-pipelined, one barrier per phase), two-barrier, two-barrier with staggered wave rows.
    python3 tools/gemm_square.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attn_small_q import timeit  # noqa: E402
from g2vlm_amd import hip  # noqa: E402

FORMS = (("pipelined", 32768), ("two-barrier", 1024), ("staggered", 131072), ("four-wave", 65536))

if __name__ == "__main__":
    hip.lib()
    torch.manual_seed(0)
    for M, N, K in ((8192, 8192, 8192), (4096, 4096, 4096), (10968, 17920, 1536), (10968, 1536, 8960), (10968, 1536, 1536), (10968, 2048, 1536)):
        x = (torch.randn((M, K), device="cuda") * 0.05).bfloat16()
        w = (torch.randn((N, K), device="cuda") * 0.05).bfloat16()
        ref = None
        for name, fl in FORMS:
            out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
            us = timeit(lambda: hip.linear(x, w, None, out=out, flags=fl), reps=10)
            if ref is None:
                ref = out.clone()
            same = bool(torch.equal(out, ref))
            print(f"M {M:6d} N {N:6d} K {K:6d} {name:12s}: {us:9.1f} us  {2.0 * M * N * K / us / 1e6:7.0f} TF/s  same bits as pipelined: {same}")
