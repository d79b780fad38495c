"""Run ONE GEMM shape a few times (for rocprofv3 --pmc passes).  usage: one_gemm.py M N K small|big [epi]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g2vlm_amd import hip
M, N, K = (int(v) for v in sys.argv[1:4])
flags = hip.FORCE_SMALL_TILE if sys.argv[4] == "small" else 0
epi = int(sys.argv[5]) if len(sys.argv) > 5 else hip.EPI_BF16
x = (torch.randn((M, K), device="cuda") * 0.5).bfloat16(); w = (torch.randn((N, K), device="cuda") * 0.5).bfloat16()
out = torch.empty((M, N // 2 if epi == hip.EPI_SWIGLU else N), dtype=torch.bfloat16, device="cuda")
for _ in range(5):
    hip.linear(x, w, None, epi, out=out, flags=flags)
torch.cuda.synchronize()
