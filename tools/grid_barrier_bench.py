"""Cost of a grid-wide barrier between 256 co-resident workgroups on MI355X (tools/grid_barrier_bench.hip).

    python tools/grid_barrier_bench.py [--build-only]

Prints, per mode, the time per barrier from HIP events around the whole launch and from the in-kernel cycle counter.
The number decides whether a persistent whole-step decode kernel (phases separated by grid barriers instead of kernel
boundaries) can beat the captured graph of per-phase kernels.
"""
import ctypes as C
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC, LIB = os.path.join(HERE, "grid_barrier_bench.hip"), os.path.join(HERE, "grid_barrier_bench.so")


def build():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", SRC, "-o", LIB], check=True)
    return LIB


def main():
    build()
    if "--build-only" in sys.argv:
        return
    import torch
    lib = C.CDLL(LIB)
    P, I = C.c_void_p, C.c_int
    lib.grid_barrier_bench.argtypes = [I, I, I, P, P, P, P, P, P]
    lib.grid_barrier_bench.restype = I
    dev = torch.device("cuda", 0)
    nb, iters = 256, 2001
    out = {}
    for mode, name in ((1, "bare atomic + poll"), (0, "with release/acquire fences"), (2, "fences + 16 KB exchange per block"),
                       (3, "two-level (8 groups of 32) with fences"), (4, "two-level, no fences"),
                       (5, "two-level, sc1 exchange of 16 KB per block, no fences"), (6, "per-block flags, sc1 exchange, no fences"),
                       (7, "8 group counters polled together, sc1 exchange, no fences")):
        best = None
        for rep in range(3):
            ctr = torch.zeros(64, dtype=torch.int32, device=dev)
            grp = torch.zeros(1024, dtype=torch.int32, device=dev)
            xch = torch.zeros(2 * nb * 512, dtype=torch.float32, device=dev)
            cyc = torch.zeros(nb, dtype=torch.int64, device=dev)
            err = torch.zeros(1, dtype=torch.int32, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            rc = lib.grid_barrier_bench(mode, nb, iters, ctr.data_ptr(), grp.data_ptr(), xch.data_ptr(), cyc.data_ptr(), err.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream)
            e1.record()
            torch.cuda.synchronize()
            assert rc == 0
            if int(err[0]):
                print(f"mode {mode}: a spin gave up (blocks not co-resident?)")
                break
            us = e0.elapsed_time(e1) * 1e3 / iters
            cy = float(cyc.double().median()) / (iters - 1)
            if best is None or us < best[0]:
                best = (us, cy)
        if best:
            out[name] = dict(us_per_barrier_events=round(best[0], 3), counter_ticks_per_barrier=round(best[1], 1))
            print(f"{name:44s} {best[0]:7.3f} us / barrier (events)   {best[1]:9.1f} counter ticks")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
