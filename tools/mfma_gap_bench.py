"""Cycles per v_mfma_f32_32x32x16_bf16 with N filler instructions in its gap, one wave per SIMD, every CU busy
(tools/mfma_gap_bench.hip): the premise of the 4 x 64 attention kernel, measured for ITS operand forms.

    python tools/mfma_gap_bench.py [--build-only]
"""
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC, LIB = os.path.join(HERE, "mfma_gap_bench.hip"), os.path.join(HERE, "mfma_gap_bench.so")


def build():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", SRC, "-o", LIB], check=True)
    return LIB


if __name__ == "__main__":
    build()
    if "--build-only" in sys.argv:
        sys.exit(0)
    import torch
    lib = C.CDLL(LIB)
    lib.mfma_gap_bench.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    out = torch.empty(256 * 256, dtype=torch.float32, device="cuda")
    cyc = torch.zeros(256, dtype=torch.int64, device="cuda")
    iters = 2000
    modes = {0: "acc in AGPR, A/B VGPR (asm)", 1: "acc in VGPR, B from AGPR (asm)", 2: "builtin"}
    kinds = {0: "v_fma_f32", 1: "v_exp_f32", 2: "fma->exp pair", 3: "ds_read_b128"}
    for fk in (0, 1, 2, 3):
        for mode in (0, 1, 2):
            row = []
            for nf in range(0, 9):
                if fk == 2 and nf > 4:
                    break
                for _ in range(2):
                    assert lib.mfma_gap_bench(mode, nf, fk, out.data_ptr(), cyc.data_ptr(), iters, None) == 0
                torch.cuda.synchronize()
                row.append(float(cyc.double().median()) / (iters * 4))
            print(f"{kinds[fk]:14s} {modes[mode]:32s} cycles / MFMA by fillers 0..: " + " ".join(f"{v:6.1f}" for v in row))
