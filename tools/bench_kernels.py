"""Micro-benchmarks of the hot kernels at the C3 shapes (HIP events on the launch stream, interleaved rounds in one
process).  Usage on the GPU box: python tools/bench_kernels.py [gemm] [attn] [norm]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g2vlm_amd import hip  # noqa: E402


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return min(ts)


def rnd(*s):
    return (torch.randn(s, device="cuda") * 0.5).bfloat16()


def gemm_cases():
    M = int(os.environ.get("G2V_BENCH_M", "10968"))
    shapes = [("mot.qkv", M, 2048, 1536, hip.EPI_BF16), ("mot.o", M, 1536, 1536, hip.EPI_RES_F32),
              ("mot.gateup", M, 17920, 1536, hip.EPI_SWIGLU), ("mot.down", M, 1536, 8960, hip.EPI_RES_F32),
              ("dino.qkv", 10992, 3072, 1024, hip.EPI_BF16), ("dino.fc1", 10992, 4096, 1024, hip.EPI_GELU),
              ("dino.fc2", 10992, 1024, 4096, hip.EPI_RES_F32), ("dec.qkv", 10952, 4608, 1536, hip.EPI_BF16),
              ("dec.fc1", 10952, 6144, 1536, hip.EPI_GELU), ("dec.fc2", 10952, 1536, 6144, hip.EPI_RES_F32)]
    if "G2V_BENCH_M" in os.environ:
        shapes = shapes[:4]
    print(f"{'gemm':12s} {'M':>6s} {'N':>6s} {'K':>5s}  small(ms) TF/s   8p-256(ms) TF/s   8p-288   8p-224   8p-192   8p-160  8p-auto  big  default(ms) TF/s")
    for name, M_, N, K, epi in shapes:
        x, w = rnd(M_, K), rnd(N, K)
        n_out = N // 2 if epi == hip.EPI_SWIGLU else N
        out = torch.empty((M_, n_out), dtype=torch.float32 if epi == hip.EPI_RES_F32 else torch.bfloat16, device="cuda")
        res = out if epi == hip.EPI_RES_F32 else None
        fl = 2.0 * M_ * N * K
        r = []
        for flags in (hip.FORCE_SMALL_TILE, hip.FORCE_8P | 512, hip.FORCE_8P | 4096, hip.FORCE_8P | 8192, hip.FORCE_8P | 128, hip.FORCE_8P | 16384, hip.FORCE_8P, hip.FORCE_BIG_TILE, 0):
            ms = timeit(lambda: hip.linear(x, w, None, epi, out=out, res=res, flags=flags))
            r.append((ms, fl / ms / 1e9))
        print(f"{name:12s} {M_:6d} {N:6d} {K:5d}  " + "  ".join(f"{ms:7.3f} {tf:5.0f}" for ms, tf in r))


def attn_cases():
    print(f"{'attn':12s} {'Lq':>6s} {'Lk':>6s} {'H':>3s} {'D':>4s}   ms    TF/s")
    for name, Lq, Lk, Hq, Hkv, D, nwin in [("mot", 10968, 10976, 12, 2, 128, 1), ("dino", 10952, 10952, 16, 16, 64, 8),
                                            ("dec", 10952, 10952, 16, 16, 96, 8), ("vit", 2916, 2916, 16, 16, 80, 1)]:
        q, k, v = rnd(Lq, Hq * D), rnd(Lk, Hkv * D), rnd(Lk, Hkv * D)
        o = torch.empty_like(q)
        wl = Lq // nwin
        wins = [(i * wl, wl, i * wl if nwin > 1 else 0, wl if nwin > 1 else Lk, False) for i in range(nwin)]
        fl = 4.0 * sum(w_[1] * w_[3] for w_ in wins) * Hq * D
        line = f"{name:12s} {Lq:6d} {Lk:6d} {Hq:3d} {D:4d}"
        for tr in (128, 256):
            plan = hip.make_attn_plan(wins, Hq, "cuda", tile_rows=tr)
            ms = timeit(lambda: hip.flash_attn(q, k, v, o, plan, Hq, Hkv, D))
            line += f"   [{tr} rows] {ms:7.3f} ms {fl / ms / 1e9:6.0f} TF/s (split {plan.n_split})"
        print(line)


def norm_cases():
    x = torch.randn((10968, 1536), device="cuda")
    w = torch.ones(1536, device="cuda")
    o = torch.empty((10968, 1536), dtype=torch.bfloat16, device="cuda")
    ms = timeit(lambda: hip.rmsnorm(x, w, w, 100, 1e-6, out=o))
    print(f"rmsnorm 10968x1536: {ms*1e3:.1f} us  {(x.numel()*4 + o.numel()*2)/ms/1e6:.0f} GB/s")
    x2 = torch.randn((10992, 1024), device="cuda"); w2 = torch.ones(1024, device="cuda")
    o2 = torch.empty((10992, 1024), dtype=torch.bfloat16, device="cuda")
    ms = timeit(lambda: hip.layernorm(x2, w2, w2, 1e-6, out=o2))
    print(f"layernorm 10992x1024: {ms*1e3:.1f} us  {(x2.numel()*4 + o2.numel()*2)/ms/1e6:.0f} GB/s")


if __name__ == "__main__":
    hip.lib()
    what = sys.argv[1:] or ["gemm", "attn", "norm"]
    if "gemm" in what:
        gemm_cases()
    if "attn" in what:
        attn_cases()
    if "norm" in what:
        norm_cases()
