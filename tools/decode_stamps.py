"""Where do the decode kernels spend their cycles?  Diagnostic build of the library with in-kernel stamps (-DG2V_STAMPS):
lane 0 of every wave records s_memtime at the phase boundaries of gemv_pg_kernel / decode_attn_pg_kernel and s_memrealtime
(100 MHz) at the kernel's start and end.  Prints, per kernel, the median over waves of each phase in shader cycles, the clock
the chip held and the span first-start -> last-end in microseconds.  Never quote this build's run time: read the SHARES.

    python tools/decode_stamps.py            # builds g2vlm_amd/lib/libg2vlm_stamps.so on first use (hipcc, CPU-side)
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "g2vlm_amd", "lib", "libg2vlm_stamps.so")


def build():
    from g2vlm_amd import build as b
    if not os.path.exists(LIB) or any(os.path.getmtime(os.path.join(b.CSRC, f)) > os.path.getmtime(LIB) for f in os.listdir(b.CSRC)):
        b.build(extra_flags=["-DG2V_STAMPS"], out=LIB)
    return LIB


if __name__ == "__main__" and "--build-only" in sys.argv:
    print(build())
    sys.exit(0)

os.environ["G2V_LIB_PATH"] = build() if not os.path.exists(LIB) else LIB
import torch  # noqa: E402

from g2vlm_amd import hip  # noqa: E402
from g2vlm_amd.weights import interleave_gate_up  # noqa: E402


def report(name, buf, n_waves, labels):
    t = buf[:n_waves * 12].view(n_waves, 12).cpu().to(torch.int64)
    live = (t[:, :len(labels) + 1] > 0).all(dim=1) & (t[:, 11] > 0)
    t = t[live]
    cyc = (t[:, len(labels)] - t[:, 0]).float()
    rt = (t[:, 11] - t[:, 10]).float() * 10.0                      # ns
    clock = float((cyc / rt).median())                              # GHz
    span_us = float(t[:, 11].max() - t[:, 10].min()) * 0.01
    parts = []
    for i, lb in enumerate(labels):
        d = (t[:, i + 1] - t[:, i]).float()
        parts.append(f"{lb} {float(d.median()):.0f} (max {float(d.max()):.0f})")
    print(f"{name}: waves {int(live.sum())}, clock {clock:.2f} GHz, wave life median {float(cyc.median()):.0f} cyc = {float(cyc.median()) / clock / 1e3:.2f} us, "
          f"kernel span {span_us:.2f} us\n    " + " | ".join(parts))


def main():
    dev = torch.device("cuda", 0)
    lib = hip.lib()
    lib.g2v_debug_stamps.argtypes, lib.g2v_debug_stamps.restype = [C.c_void_p], C.c_int
    dbg = torch.zeros(4096 * 12, dtype=torch.int64, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(0)
    r = lambda *s, sc=0.02: torch.randn(s, generator=g, device=dev) * sc  # noqa: E731
    H, Hq, Hkv, Fd, cap, Lk = 1536, 12, 2, 8960, 12288, 10976
    NL = 6                                                                # distinct layers: every launch reads cold weights
    ws = [dict(qkv=r((Hq + 2 * Hkv) * 128, H).bfloat16(), qkvb=r((Hq + 2 * Hkv) * 128).bfloat16(), o=r(H, Hq * 128).bfloat16(),
               gu=interleave_gate_up(r(Fd, H), r(Fd, H)).bfloat16(), down=r(H, Fd).bfloat16(),
               k=torch.randn((cap, Hkv, 128), generator=g, device=dev).bfloat16(), v=torch.randn((cap, Hkv, 128), generator=g, device=dev).bfloat16())
          for _ in range(NL)]
    x = r(H, sc=1.0)
    ln = torch.ones(H, device=dev)
    qn = torch.ones(128, device=dev)
    qkv = torch.empty((1, (Hq + 2 * Hkv) * 128), dtype=torch.bfloat16, device=dev)
    ao = torch.empty((1, Hq * 128), dtype=torch.bfloat16, device=dev)
    act = torch.empty(Fd, dtype=torch.bfloat16, device=dev)
    cos = torch.ones((1, 128), device=dev); sin = torch.zeros((1, 128), device=dev)
    ld = torch.tensor([Lk], dtype=torch.int32, device=dev)
    ws2 = torch.empty(hip.decode_attn_pg_workspace(Hq, Hkv, 1) // 4, dtype=torch.float32, device=dev)
    trash = torch.empty(512 << 20, dtype=torch.uint8, device=dev)

    def run(fn, name, n_waves, labels):
        for w in ws[:-1]:                                                 # warm code paths, leave the last layer's weights cold
            fn(w)
        trash.fill_(1)                                                    # flush L2 / Infinity Cache
        torch.cuda.synchronize()
        dbg.zero_()
        lib.g2v_debug_stamps(C.c_void_p(dbg.data_ptr()))
        fn(ws[-1])
        torch.cuda.synchronize()
        lib.g2v_debug_stamps(None)
        report(name, dbg, n_waves, labels)

    gl = ["issue", "x ready+norm", "weights+dot", "reduce", "store"]
    run(lambda w: hip.gemv_pg(x, w["qkv"], norm_w=ln, eps=1e-6, bias=w["qkvb"], out=qkv.view(-1)), "qkv gemv (N 2048, K 1536, fused norm)", 2048, gl)
    run(lambda w: hip.gemv_pg(ao.view(-1), w["o"], res=x), "o gemv (N 1536, K 1536, residual)", 2048, gl)
    run(lambda w: hip.gemv_pg(x, w["gu"], norm_w=ln, eps=1e-6, out=act, act=True), "gate/up gemv (N 17920, K 1536, norm + SwiGLU)", 2048, gl)
    run(lambda w: hip.gemv_pg(act, w["down"], res=x), "down gemv (N 1536, K 8960, residual)", 2048, gl)
    al = ["issue loads", "q/k norm", "batch 1: scores, softmax, P.V", "further batches", "result -> LDS", "barrier", "merge+store"]
    run(lambda w: hip.decode_attn_pg(qkv, qn, qn, 1e-6, 1, cos, sin, w["k"], w["v"], ao, ld, cap, cap, Hq, Hkv, 128 ** -0.5, ws2),
        "attention pg (Lk 10976, cap 12288)", 1024, al)


if __name__ == "__main__":
    main()
