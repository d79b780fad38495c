"""Cumulative cycle profile of one KV tile of flash_fwd64_kernel: a fixed stamp at the tile's entry and exit and ONE movable
stamp whose position (after the reference check, after each of the 8 QK^T slices, after each of the 16 P.V slices, after the
DMA wait) is chosen per launch - 26 launches, two s_memtime per tile each, so the schedule is barely perturbed.
Diagnostic build -DEXP_STAMPS -DEXP_PSTAMPS (g2vlm_amd/lib/exp/lib_pstamps.so; the branches of the movable stamp slow the tile by ~35 %: read the profile relatively).    python tools/attn64_profile.py [mot|c4rank]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g2vlm_amd import build  # noqa: E402

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "g2vlm_amd", "lib", "exp", "lib_pstamps.so")
if not os.path.exists(out):
    build.build(extra_flags=["-DEXP_STAMPS", "-DEXP_PSTAMPS"], out=out)
os.environ["G2V_LIB_PATH"] = out
import torch  # noqa: E402

from g2vlm_amd import hip  # noqa: E402

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "mot"
    Lq, Lk = {"mot": (10968, 10976), "c4rank": (5484, 43880)}[what]
    Hq, Hkv, D = 12, 2, 128
    torch.manual_seed(0)
    q = torch.randn((Lq, Hq * D), device="cuda").bfloat16()
    k = torch.randn((Lk, Hkv * D), device="cuda").bfloat16()
    v = torch.randn((Lk, Hkv * D), device="cuda").bfloat16()
    o = torch.empty_like(q)
    plan = hip.make_attn_plan([(0, Lq, 0, Lk, False)], Hq, "cuda", tile_rows=256)
    lib = hip.lib()
    for n, sig in (("g2v_debug_attn_stamps", [C.c_void_p]), ("g2v_debug_attn_stamp_pos", [C.c_int]), ("g2v_debug_attn_form", [C.c_int])):
        getattr(lib, n).argtypes, getattr(lib, n).restype = sig, C.c_int
    lib.g2v_debug_attn_form(1)
    NT, NP = 24, 8
    buf = torch.zeros(3 * 2 * NT * NP, dtype=torch.int64, device="cuda")
    for _ in range(5):
        hip.flash_attn(q, k, v, o, plan, Hq, Hkv, D)
    names = ["reference check"] + [f"QK slice {i}" for i in range(8)] + [f"PV slice {i}" for i in range(16)] + ["sums + vmcnt wait"]
    prev = 0.0
    for pos in range(26):
        buf.zero_()
        lib.g2v_debug_attn_stamp_pos(pos)
        lib.g2v_debug_attn_stamps(buf.data_ptr())
        hip.flash_attn(q, k, v, o, plan, Hq, Hkv, D)
        torch.cuda.synchronize()
        lib.g2v_debug_attn_stamps(None)
        t = buf.view(3, 2, NT, NP).cpu()[0, 0]                 # workgroup 0, wave 0
        cum = (t[:, 1] - t[:, 0]).float().median().item()
        tile = (t[:, 2] - t[:, 0]).float().median().item()
        period = (t[1:, 0] - t[:-1, 0]).float().median().item()
        print(f"{names[pos]:18s} cumulative {cum:7.0f}  (+{cum - prev:5.0f})   tile entry->exit {tile:6.0f}  period {period:6.0f}")
        prev = cum
