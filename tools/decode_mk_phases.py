"""Where the one-launch decode step (csrc/decode_mk.hip) spends its time: workgroup 0's 100 MHz stamps at every barrier.

    python tools/decode_mk_phases.py [--kv 10976] [--layers 28]
"""
import argparse
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from decode_bench import FakeWeights  # noqa: E402
from g2vlm_amd import hip  # noqa: E402
from g2vlm_amd.engine import Engine, KVCache  # noqa: E402
from g2vlm_amd.synthetic import REAL_DIMS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kv", type=int, default=10976)
    ap.add_argument("--layers", type=int, default=28)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    dims = copy.deepcopy(REAL_DIMS)
    dims["llm"]["layers"] = a.layers
    L = dims["llm"]
    eng = Engine(FakeWeights(dims, dev, a.layers), dims)
    eng.decode_gen = 3
    cache = KVCache(a.layers, L["kv_heads"], dev, capacity=a.kv + 256)
    cache.length = a.kv
    buf = torch.zeros(4 * 6 * a.layers + 8, dtype=torch.int64, device=dev)
    hip.lib().g2v_debug_mk_stamps(buf.data_ptr())               # before the capture: the graph freezes the kernel's arguments
    st = eng.decode_begin(cache, 5, a.kv, 64, use_graph=True)
    for _ in range(5):
        eng.decode_step(st)
    torch.cuda.synchronize()
    t = buf.cpu().tolist()
    hip.lib().g2v_debug_mk_stamps(None)
    names = ["qkv", "attn", "combine", "o", "gate/up", "down"]
    nb = 6 * a.layers
    # per barrier e (1-based): t[4e-3] work issued, t[4e-2] workgroup done, t[4e-1] arrival acknowledged, t[4e] released
    seg = {n: [0.0] * 4 for n in names}
    for e in range(1, nb + 1):
        prev = t[4 * (e - 1)]
        n = names[(e - 1) % 6]
        cur = [t[4 * e - 3], t[4 * e - 2], t[4 * e - 1], t[4 * e]]
        last = prev
        for i, v in enumerate(cur):
            seg[n][i] += (v - last) / 100.0
            last = v
    print("per layer, workgroup 0, us: [work issued by wave 0, rest of the workgroup + stores acknowledged, arrival atomic(s), wait for release]")
    for n in names:
        print(f"  {n:8s}", [round(v / a.layers, 2) for v in seg[n]], "total", round(sum(seg[n]) / a.layers, 2))
    print("layers total %.1f us, lm_head %.1f us, step %.1f us" % ((t[4 * nb] - t[0]) / 100.0, (t[4 * nb + 1] - t[4 * nb]) / 100.0,
                                                                   (t[4 * nb + 1] - t[0]) / 100.0))


if __name__ == "__main__":
    main()
