"""Launch ONE GEMM shape of the C3 workload a few times (target of rocprofv3 --pmc passes, see tools/attn_only.py).
    python3 tools/gemm_only.py [gateup|down|qkv|fc1] [flags] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g2vlm_amd import hip  # noqa: E402

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "gateup"
    flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    hip.lib()
    torch.manual_seed(0)
    M, N, K, epi = {"gateup": (10968, 17920, 1536, hip.EPI_SWIGLU), "down": (10968, 1536, 8960, hip.EPI_RES_F32),
                    "qkv": (10968, 2048, 1536, hip.EPI_BF16), "fc1": (10952, 6144, 1536, hip.EPI_GELU)}[what]
    x = (torch.randn((M, K), device="cuda") * 0.5).bfloat16()
    w = (torch.randn((N, K), device="cuda") * 0.5).bfloat16()
    n_out = N // 2 if epi == hip.EPI_SWIGLU else N
    out = torch.empty((M, n_out), dtype=torch.float32 if epi == hip.EPI_RES_F32 else torch.bfloat16, device="cuda")
    res = out if epi == hip.EPI_RES_F32 else None
    for _ in range(reps):
        hip.linear(x, w, None, epi, out=out, res=res, flags=flags)
    torch.cuda.synchronize()
    print("ok", what, flags, reps)
