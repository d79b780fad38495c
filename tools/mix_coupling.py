"""Does the form of one GEMM change the duration of an UNRELATED kernel that runs next to it?  gate/up (eight-wave / four-wave) and the
MoT attention launch alternate for `secs` seconds (they share no data), then per-kernel durations are taken with events over the last
third; beside them the same kernels in homogeneous loops.      python3 tools/mix_coupling.py [secs]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g2vlm_amd import hip  # noqa: E402
from g2vlm_amd.weights import interleave_gate_up  # noqa: E402


def run(fns, secs):
    """fns: list of callables launched round-robin; returns mean us per callable over the last third."""
    for f in fns:
        f()
    torch.cuda.synchronize()
    t0 = time.time()
    while time.time() - t0 < secs * 2 / 3:
        for _ in range(10):
            for f in fns:
                f()
        torch.cuda.synchronize()
    reps = 200
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)] for _ in fns]
    for r in range(reps):
        for i, f in enumerate(fns):
            ev[i][r][0].record(); f(); ev[i][r][1].record()
    torch.cuda.synchronize()
    return [sum(a.elapsed_time(b) for a, b in e) * 1e3 / reps for e in ev]


if __name__ == "__main__":
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    hip.lib()
    torch.manual_seed(0)
    r = lambda *s: (torch.randn(s, device="cuda") * 0.05).bfloat16()  # noqa: E731
    M, H, F = 10968, 1536, 8960
    x, w = r(M, H), interleave_gate_up(r(F, H), r(F, H))
    act = torch.empty((M, F), dtype=torch.bfloat16, device="cuda")
    gu8 = lambda: hip.linear(x, w, None, hip.EPI_SWIGLU, out=act, flags=hip.FORCE_8P | hip.P8_EIGHT_WAVES)  # noqa: E731
    gu4 = lambda: hip.linear(x, w, None, hip.EPI_SWIGLU, out=act, flags=hip.FORCE_8P | hip.P8_FOUR_WAVES)   # noqa: E731
    Hq, Hkv, D, Lq, Lk = 12, 2, 128, 10968, 10976
    q = torch.randn((Lq, Hq * D), device="cuda").bfloat16()
    k = torch.randn((Lk, Hkv * D), device="cuda").bfloat16()
    v = torch.randn((Lk, Hkv * D), device="cuda").bfloat16()
    o = torch.empty_like(q)
    plan = hip.make_attn_plan([(0, Lq, 0, Lk, False)], Hq, "cuda", tile_rows=256)
    att = lambda: hip.flash_attn(q, k, v, o, plan, Hq, Hkv, D)  # noqa: E731
    solo = {n: run([f], secs)[0] for n, f in (("gate/up eight-wave", gu8), ("gate/up four-wave", gu4), ("attention", att))}
    for n, t in solo.items():
        print(f"homogeneous loop  {n:20s} {t:8.1f} us", flush=True)
    for n, f in (("eight-wave", gu8), ("four-wave", gu4)):
        g, a = run([f, att], secs)
        print(f"alternating       gate/up {n:11s} {g:8.1f} us | attention {a:8.1f} us | pair {g + a:8.1f} us (sum of the homogeneous loops "
              f"{solo['gate/up ' + n] + solo['attention']:8.1f})", flush=True)
