"""Yardstick only (never the product path): hip.linear (four-wave form; eight-wave staggered form of round 2) beside torch's bf16 matmul (hipBLASLt / rocBLAS, whatever
this image's PyTorch dispatches to) on the C3 Linear shapes and two squares, same inputs, same process, interleaved.
    python3 tools/gemm_yardstick.py"""
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
    for M, N, K in ((8192, 8192, 8192), (4096, 4096, 4096), (10968, 17920, 1536), (10968, 1536, 8960), (10968, 1536, 1536), (10968, 2048, 1536),
                    (10992, 3072, 768), (10992, 768, 3072), (10952, 4096, 1024)):
        x = (torch.randn((M, K), device="cuda") * 0.05).bfloat16()
        w = (torch.randn((N, K), device="cuda") * 0.05).bfloat16()
        out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        out2 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        out3 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        wt = w.t()
        res = []
        for rep in range(2):
            a = timeit(lambda: hip.linear(x, w, None, out=out, flags=hip.FORCE_8P | hip.P8_EIGHT_WAVES), reps=10)
            b = timeit(lambda: torch.matmul(x, wt, out=out2), reps=10)
            c = timeit(lambda: hip.linear(x, w, None, out=out3, flags=hip.FORCE_8P | hip.P8_FOUR_WAVES), reps=10)
            res.append((a, b, c))
        a, b, c = min(r[0] for r in res), min(r[1] for r in res), min(r[2] for r in res)
        fl = 2.0 * M * N * K / 1e6
        rel = float((out.double() - out2.double()).norm() / out2.double().norm())
        print(f"M {M:6d} N {N:6d} K {K:6d}: eight-wave {a:8.1f} us {fl / a:6.0f} TF/s | torch.matmul {b:8.1f} us {fl / b:6.0f} TF/s | four-wave {c:8.1f} us {fl / c:6.0f} TF/s same bits {bool(torch.equal(out, out3))} | rel diff {rel:.1e}", flush=True)
