"""Energy per FLOP by MFMA shape (tools/mfma_shape_bench.hip): bare 32x32x16 and 16x16x32 bf16 loops with the same FLOPs per launch
(about one MoT attention launch), alone for `secs` seconds and alternating with the eight-wave gate/up GEMM launch.  In the mix the time
of a pair of MFMA-bound kernels is their energy over the power the chip holds (tools/mix_coupling.py): the pair times rank the
shapes by energy.      python tools/mfma_shape_bench.py [secs] [--build-only]"""
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC, LIB = os.path.join(HERE, "mfma_shape_bench.hip"), os.path.join(HERE, "mfma_shape_bench.so")
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def build():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", SRC, "-o", LIB], check=True)
    return LIB


if __name__ == "__main__":
    build()
    if "--build-only" in sys.argv:
        sys.exit(0)
    import torch
    from g2vlm_amd import hip
    from g2vlm_amd.weights import interleave_gate_up
    from mix_coupling import run
    secs = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 3.0
    lib = C.CDLL(LIB)
    lib.mfma_shape_bench.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    hip.lib()
    torch.manual_seed(0)
    rnd = torch.randn((256 * 256 * 8, 8), device="cuda").bfloat16()
    out = torch.empty(256 * 256, dtype=torch.float32, device="cuda")
    iters = 2750                                              # 256 x 4 x 2750 x 2 x 64 x 64 x 32 = 7.38e11 FLOP, one C3 attention launch
    fl = 256 * 4 * iters * 2.0 * 64 * 64 * 32
    st = torch.cuda.current_stream().cuda_stream
    k32 = lambda: lib.mfma_shape_bench(0, rnd.data_ptr(), out.data_ptr(), iters, st)    # noqa: E731
    k16 = lambda: lib.mfma_shape_bench(1, rnd.data_ptr(), out.data_ptr(), iters, st)    # noqa: E731
    r = lambda *s: (torch.randn(s, device="cuda") * 0.05).bfloat16()  # noqa: E731
    M, H, F = 10968, 1536, 8960
    x, w = r(M, H), interleave_gate_up(r(F, H), r(F, H))
    act = torch.empty((M, F), dtype=torch.bfloat16, device="cuda")
    gu = lambda: hip.linear(x, w, None, hip.EPI_SWIGLU, out=act, flags=hip.FORCE_8P | hip.P8_EIGHT_WAVES)  # noqa: E731
    solo = {}
    for n, f in (("32x32x16", k32), ("16x16x32", k16), ("gate/up", gu)):
        solo[n] = run([f], secs)[0]
        print(f"homogeneous loop  {n:9s} {solo[n]:8.1f} us" + (f"  {fl / solo[n] / 1e6:6.0f} TF/s" if n != "gate/up" else ""), flush=True)
    for n, f in (("32x32x16", k32), ("16x16x32", k16)):
        g, a = run([gu, f], secs)
        print(f"alternating       gate/up {g:8.1f} us | {n} loop {a:8.1f} us | pair {g + a:8.1f} us", flush=True)
