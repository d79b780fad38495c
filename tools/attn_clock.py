"""In-kernel clock of the MoT prefill attention (MI355X guide, 'DVFS give-back' item 6): wave 0 of every workgroup stamps
s_memtime (shader cycles) and s_memrealtime (100 MHz) at kernel entry and exit, after >= 2 s of back-to-back launches on random
data.  Prints, for both kernel forms, the median clock, the kernel's length in cycles and per KV tile, and the wall time.
Diagnostic build -DEXP_STAMPS.     python tools/attn_clock.py [mot|c4rank]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g2vlm_amd import build  # noqa: E402

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "g2vlm_amd", "lib", "exp", "lib_stamps.so")
if not os.path.exists(out):
    build.build(extra_flags=["-DEXP_STAMPS"], out=out)
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
    for n, sig in (("g2v_debug_attn_clock", [C.c_void_p]), ("g2v_debug_attn_form", [C.c_int])):
        getattr(lib, n).argtypes, getattr(lib, n).restype = sig, C.c_int
    tiles_per_wg = (Lq + 255) // 256 * Hq * ((Lk + 63) // 64) / plan.n_blocks
    for form in (0, 1, 0, 1):
        lib.g2v_debug_attn_form(form)
        t0 = time.time()
        while time.time() - t0 < 2.0:
            for _ in range(50):
                hip.flash_attn(q, k, v, o, plan, Hq, Hkv, D)
            torch.cuda.synchronize()
        buf = torch.zeros(2 * plan.n_blocks, dtype=torch.int64, device="cuda")
        lib.g2v_debug_attn_clock(buf.data_ptr())
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        hip.flash_attn(q, k, v, o, plan, Hq, Hkv, D)
        ev1.record()
        torch.cuda.synchronize()
        lib.g2v_debug_attn_clock(None)
        t = buf.view(-1, 2).cpu().double()
        clk = (t[:, 0] / t[:, 1] * 100).median().item()
        cyc = t[:, 0].median().item()
        print(f"form {form}: in-kernel clock {clk:6.0f} MHz, kernel {cyc:9.0f} cycles = {cyc / tiles_per_wg:6.0f} per KV tile "
              f"({tiles_per_wg:.0f} tiles per workgroup), longest workgroup {t[:, 1].max().item() / 100:7.1f} us, launch {ev0.elapsed_time(ev1) * 1e3:7.1f} us")
    lib.g2v_debug_attn_form(1)
