"""Build libg2vlm_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the built
.so travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libg2vlm_hip.so")
SOURCES = ["gemm.hip", "gemm_big.hip", "gemm_8p.hip", "gemm_skinny.hip", "attn.hip", "norm_rope.hip", "misc.hip", "decode.hip"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "g2vlm_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), out=None):
    """Compile every HIP source into one shared library.  Returns the library path.
    extra_flags / out: experiment builds (tools/attn_variants.sh), never the shipped library."""
    if out is not None:
        return _compile(list(extra_flags), out, verbose)
    if not force and not _stale():
        return LIB
    return _compile([], LIB, verbose)


def _compile(extra_flags, LIB, verbose):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *extra_flags,
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, *srcs, "-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
