"""A/B of the two forms of the MoT prefill attention (head dim 128, 256-row items): form 1 = 4 waves x 64 rows, one wave per
SIMD (flash_fwd64_kernel), form 0 = 8 waves x 32 rows (flash_fwd_kernel<128, 8>).  Time, agreement between the forms, agreement
with an fp64 softmax on a row sample, and run-to-run bit identity (a hand-spaced hazard that is too short shows as flicker).

    python tools/attn_ab.py [mot,c4rank,causal,vitpre]
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attn_small_q import timeit  # noqa: E402
from g2vlm_amd import hip  # noqa: E402


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


if __name__ == "__main__":
    lib = hip.lib()
    form = lib.g2v_debug_attn_form
    form.argtypes, form.restype = [C.c_int], C.c_int
    torch.manual_seed(0)
    # (Lq, Lk, causal, Q scale): vitpre = a 731-row image prefill against a long cache; causal = a long text prompt
    cases = {"mot": (10968, 10976, False, 1.0), "c4rank": (5484, 43880, False, 1.0), "causal": (4000, 6000, True, 1.0), "vitpre": (2924, 17000, False, 1.0),
             "peaked": (10968, 10976, False, 6.0)}
    Hq, Hkv, D = 12, 2, 128
    for what in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["mot", "c4rank", "causal", "vitpre", "peaked"]):
        Lq, Lk, causal, qs = cases[what]
        q = (torch.randn((Lq, Hq * D), device="cuda") * qs).bfloat16()
        k = torch.randn((Lk, Hkv * D), device="cuda").bfloat16()
        v = torch.randn((Lk, Hkv * D), device="cuda").bfloat16()
        plan = hip.make_attn_plan([(0, Lq, 0, Lk, causal)], Hq, "cuda", tile_rows=256)
        outs, times = {}, {}
        for f in (0, 1):
            form(f)
            o = torch.zeros_like(q)
            hip.flash_attn(q, k, v, o, plan, Hq, Hkv, D)
            torch.cuda.synchronize()
            first = o.clone()
            flick = 0
            for _ in range(20):
                o2 = torch.zeros_like(q)
                hip.flash_attn(q, k, v, o2, plan, Hq, Hkv, D)
                flick += int(not torch.equal(o2, first))
            times[f] = min(timeit(lambda: hip.flash_attn(q, k, v, o, plan, Hq, Hkv, D), reps=20) for _ in range(3))
            outs[f] = (first, flick)
        form(1)
        # fp64 reference on a sample of rows of head 0 and head 7
        rows = torch.randint(0, Lq, (64,), device="cuda")
        errs = {}
        for f in (0, 1):
            e = []
            for h in (0, 7):
                kvh = h // (Hq // Hkv)
                qq = q[rows, h * D:(h + 1) * D].double()
                s = qq @ k[:, kvh * D:(kvh + 1) * D].double().T * D ** -0.5
                if causal:
                    lim = rows + (Lk - Lq)
                    s = s.masked_fill(torch.arange(Lk, device="cuda")[None, :] > lim[:, None], float("-inf"))
                ref = torch.softmax(s, -1) @ v[:, kvh * D:(kvh + 1) * D].double()
                e.append(rel(outs[f][0][rows, h * D:(h + 1) * D], ref))
            errs[f] = max(e)
        fl = 4.0 * Lq * Lk * Hq * D * (0.5 + 0.5 * (Lk - Lq) / Lk if causal else 1.0)
        print(f"{what:7s} form0 {times[0]:8.1f} us ({fl / times[0] / 1e6:5.0f} TF/s)  form1 {times[1]:8.1f} us ({fl / times[1] / 1e6:5.0f} TF/s)  x{times[0] / times[1]:.3f}  "
              f"form1 vs form0 rel {rel(outs[1][0], outs[0][0]):.2e}  vs fp64: form0 {errs[0]:.2e} form1 {errs[1]:.2e}  flicker {outs[0][1]}/{outs[1][1]} of 20")
