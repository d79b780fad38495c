"""Where a KV tile of the MoT prefill attention spends its cycles: s_memtime stamps of waves 0 and 4 of three workgroups
(diagnostic build -DEXP_STAMPS of csrc/attn.hip; the product library has no stamp).  Stamp points per tile:
0 entry, 1 after the DMA issue, 2 after row max / rescale decision, 3 after the QK^T + exp slices, 4 after P.V, 5 after the
DMA wait, 6 after the barrier.

    python tools/attn_stamps.py [mot|c4rank]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g2vlm_amd import build  # noqa: E402

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "g2vlm_amd", "lib", "exp", "lib_stamps.so")
if True:
    build.build(extra_flags=["-DEXP_STAMPS"] + [f for f in os.environ.get("G2V_EXTRA_FLAGS", "").split() if f], out=out)
os.environ["G2V_LIB_PATH"] = out
import ctypes as C  # noqa: E402

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
    NT, NP = 24, 8
    buf = torch.zeros(3 * 2 * NT * NP, dtype=torch.int64, device="cuda")
    fn = hip.lib().g2v_debug_attn_stamps
    fn.argtypes, fn.restype = [C.c_void_p], C.c_int
    ff = hip.lib().g2v_debug_attn_form
    ff.argtypes, ff.restype = [C.c_int], C.c_int
    ff(int(os.environ.get("G2V_ATTN_FORM", "1")))
    for _ in range(5):
        hip.flash_attn(q, k, v, o, plan, Hq, Hkv, D)
    torch.cuda.synchronize()
    fn(buf.data_ptr())
    hip.flash_attn(q, k, v, o, plan, Hq, Hkv, D)
    torch.cuda.synchronize()
    fn(None)
    t = buf.view(3, 2, NT, NP).cpu()
    form = int(os.environ.get("G2V_ATTN_FORM", "1"))
    names = ["dma issue", "kfrag+rowmax+rescale", "QK^T(next)+exp slices", "P.V (+exp b=1)", "dma wait", "barrier"] if form == 0 else \
        ["reference check", "phase 1: QK^T(next) + exp", "phase 2: P.V + exp + max + DMA", "sums + vmcnt wait", "barrier", "-"]
    for b in range(3):
        for wv in range(2):
            x = t[b, wv]
            if int(x[0, 0]) == 0:
                continue
            npt = 6 if form == 0 else 5
            d = (x[:, 1:npt + 1] - x[:, 0:npt]).float()
            per_tile = (x[1:, 0] - x[:-1, 0]).float()
            print(f"block {b} wave {4 * wv}: cycles per tile median {per_tile.median():.0f} (min {per_tile.min():.0f} max {per_tile.max():.0f}); "
                  f"in-tile sum {d.sum(1).median():.0f}")
            for i, n in enumerate(names[:npt]):
                print(f"    {n:26s} median {d[:, i].median():7.0f}  min {d[:, i].min():7.0f}  max {d[:, i].max():7.0f}")
            gap = (x[1:, 0] - x[:-1, npt]).float()
            print(f"    {'tile seam (6 -> next 0)':26s} median {gap.median():7.0f}")
