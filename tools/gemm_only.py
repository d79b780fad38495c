"""Launch one GEMM shape of the C3 path a few times with a chosen main-loop form: the target of rocprofv3 PMC passes.
    python3 tools/gemm_only.py <gu|down|o|qkv|sq8k> <staggered|pipelined|two_barrier> [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g2vlm_amd import hip  # noqa: E402
from g2vlm_amd.weights import interleave_gate_up  # noqa: E402

if __name__ == "__main__":
    what, form = sys.argv[1], sys.argv[2]
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    fl = {"four_wave": hip.P8_FOUR_WAVES, "staggered": hip.P8_EIGHT_WAVES, "pipelined": hip.P8_PIPELINED, "two_barrier": hip.P8_TWO_BARRIER}[form] | hip.FORCE_8P
    hip.lib()
    torch.manual_seed(0)
    r = lambda *s: (torch.randn(s, device="cuda") * 0.05).bfloat16()  # noqa: E731
    M, H, F = 10968, 1536, 8960
    if what == "gu":
        x, w = r(M, H), interleave_gate_up(r(F, H), r(F, H))
        run = lambda: hip.linear(x, w, None, hip.EPI_SWIGLU, flags=fl)  # noqa: E731
    elif what == "down":
        x, w, res = r(M, F), r(H, F), torch.randn((M, H), device="cuda")
        run = lambda: hip.linear(x, w, None, hip.EPI_RES_F32, out=res, res=res, flags=fl)  # noqa: E731
    elif what == "o":
        x, w, res = r(M, H), r(H, H), torch.randn((M, H), device="cuda")
        run = lambda: hip.linear(x, w, None, hip.EPI_RES_F32, out=res, res=res, flags=fl)  # noqa: E731
    elif what == "qkv":
        x, w = r(M, H), r(2048, H)
        run = lambda: hip.linear(x, w, None, flags=fl)  # noqa: E731
    else:
        x, w = r(8192, 8192), r(8192, 8192)
        run = lambda: hip.linear(x, w, None, flags=fl)  # noqa: E731
    for _ in range(reps):
        out = run()
    torch.cuda.synchronize()
    print("ok", what, form, float(out.float().abs().mean()))
