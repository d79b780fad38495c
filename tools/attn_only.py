"""Launch the MoT prefill attention (and nothing else of ours) at the C3 shape a few times: the target of the
rocprofv3 PMC passes.  Usage on the GPU box (one counter group per pass, each its own process):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o p -- python3 tools/attn_only.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o p -- python3 tools/attn_only.py
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d ... -- python3 tools/attn_only.py
    python3 tools/pmc_summary.py gpurun_out/pmc_* > profiles/rNN_attn_pmc.md

K/V are unit-RMS rows like the cache content after qk-norm (K) and a projection (V); Q likewise.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g2vlm_amd import hip  # noqa: E402

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "mot"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    hip.lib()
    torch.manual_seed(0)
    # mot = C3; c4 = the 32-view scene unsharded; c4rank = one of C4's 8 ranks (its 4 views' queries against all 43 880 keys)
    cases = {"mot": (10968, 10976, 12, 2, 128, 1, 256), "c4": (43872, 43880, 12, 2, 128, 1, 256), "c4rank": (5484, 43880, 12, 2, 128, 1, 256),
             "dino": (10952, 10952, 16, 16, 64, 8, 128),
             "dec": (10952, 10952, 16, 16, 96, 8, 128), "vit": (2916, 2916, 16, 16, 80, 1, 256)}
    Lq, Lk, Hq, Hkv, D, nwin, rows = cases[what]
    q = torch.randn((Lq, Hq * D), device="cuda").bfloat16()
    k = torch.randn((Lk, Hkv * D), device="cuda").bfloat16()
    v = torch.randn((Lk, Hkv * D), device="cuda").bfloat16()
    o = torch.empty_like(q)
    wl = Lq // nwin
    wins = [(i * wl, wl, i * wl if nwin > 1 else 0, wl if nwin > 1 else Lk, False) for i in range(nwin)]
    plan = hip.make_attn_plan(wins, Hq, "cuda", tile_rows=rows)
    for _ in range(reps):
        hip.flash_attn(q, k, v, o, plan, Hq, Hkv, D)
    torch.cuda.synchronize()
    print("ok", what, reps, float(o.float().abs().mean()))
